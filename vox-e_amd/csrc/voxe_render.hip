// voxe_render.hip -- fused volumetric render kernels for gfx950 (MI355X), forward + backward.
//
// One kernel replaces the reference's sampler -> point processor -> accumulator chain
// (thre3d_atom/rendering/volumetric/render_interface.py:140-171) and never materialises the
// [rays x samples] temporaries; the backward kernel replaces autograd through that chain.
//
// HBM layout: the reference keeps densities [X,Y,Z,1] and features [X,Y,Z,F] as two tensors.  The
// kernels work on a packed array-of-structs copy [X,Y,Z,C], C = F+1, channel order
// (f_0..f_{F-1}, pre(d*scale)) built by pack_grid_kernel (one streaming pass), so a trilinear
// corner is ONE aligned 16-byte load for SH-0 (float4) / 8-byte load for the attention grid, and
// the two z-neighbours of a corner pair are contiguous (32 B).  Gradients are accumulated in the
// same packed layout and split back (with the pre-activation chain rule) by unpack_grad_kernel.
#include <stdint.h>
#include <stdlib.h>

#include <math.h>

#include "voxe_device.hpp"
#include "voxe_launch.hpp"
#include "voxe_render_common.hpp"

namespace voxe {

// ------------------------------------------------------------------------------------------------
// pack / unpack (HBM streaming; voxels.py:303-305 pre-activation is applied per voxel here, exactly
// like the reference applies it to the whole grid before interpolating)
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void pack_grid_kernel(const float* __restrict__ dens,
                                                        const float* __restrict__ feat,
                                                        float* __restrict__ packed, long long nvox,
                                                        float scale, int pre_act) {
  constexpr int F = C - 1;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvox; i += stride) {
    float v[C];
#pragma unroll
    for (int f = 0; f < F; ++f) v[f] = feat[i * F + f];
    v[F] = pre_activate(pre_act, dens[i], scale);
    if constexpr (C == 4) {
      reinterpret_cast<float4*>(packed)[i] = make_float4(v[0], v[1], v[2], v[3]);
    } else if constexpr (C == 2) {
      reinterpret_cast<float2*>(packed)[i] = make_float2(v[0], v[1]);
    } else {
#pragma unroll
      for (int f = 0; f < C; ++f) packed[i * C + f] = v[f];
    }
  }
}

template <int C>
__global__ __launch_bounds__(256) void unpack_grad_kernel(const float* __restrict__ gpacked,
                                                          const float* __restrict__ dens,
                                                          float* __restrict__ d_dens,
                                                          float* __restrict__ d_feat, long long nvox,
                                                          float scale, int pre_act, int accumulate,
                                                          int bricked, int Y, int Z) {
  constexpr int F = C - 1;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvox; i += stride) {
    // source position: the voxel itself, or its slot in the 2x2x2-bricked gradient buffer of the scatter backward
    long long si = i;
    if (bricked) {
      const int z = (int)(i % Z), y = (int)((i / Z) % Y), x = (int)(i / ((long long)Y * Z));
      si = brick_slot(x, y, z, Y, Z);
    }
    float v[C];
    if constexpr (C == 4) {
      const float4 t = reinterpret_cast<const float4*>(gpacked)[si];
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else if constexpr (C == 2) {
      const float2 t = reinterpret_cast<const float2*>(gpacked)[si];
      v[0] = t.x; v[1] = t.y;
    } else {
#pragma unroll
      for (int f = 0; f < C; ++f) v[f] = gpacked[si * C + f];
    }
    if (d_feat) {
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const long long j = i * F + f;
        d_feat[j] = accumulate ? d_feat[j] + v[f] : v[f];
      }
    }
    if (d_dens) {
      const float gval = v[F] * pre_activate_grad(pre_act, dens[i], scale);
      d_dens[i] = accumulate ? d_dens[i] + gval : gval;
    }
  }
}

// Wide texels (13 / 28 / 49 channels): one thread per ELEMENT of the packed array, so that consecutive lanes touch
// consecutive floats of both layouts (a thread per voxel strides by C floats: 64 cache lines per wave instruction).
template <int C>
__global__ __launch_bounds__(256) void pack_grid_wide_kernel(const float* __restrict__ dens, const float* __restrict__ feat,
                                                             float* __restrict__ packed, long long nvox, float scale,
                                                             int pre_act) {
  constexpr int F = C - 1;
  // (32-bit index math: validate() bounds nvox * C below 2^31)
  const unsigned n = (unsigned)(nvox * C), stride = gridDim.x * blockDim.x;
  for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
    const unsigned i = e / C, ch = e - i * C;
    packed[e] = ch < F ? feat[i * F + ch] : pre_activate(pre_act, dens[i], scale);
  }
}

template <int C>
__global__ __launch_bounds__(256) void unpack_grad_wide_kernel(const float* __restrict__ gpacked,
                                                               const float* __restrict__ dens, float* __restrict__ d_dens,
                                                               float* __restrict__ d_feat, long long nvox, float scale,
                                                               int pre_act, int accumulate, int bricked, int Y, int Z) {
  constexpr int F = C - 1;
  const unsigned n = (unsigned)(nvox * C), stride = gridDim.x * blockDim.x;
  for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
    const unsigned i = e / C, ch = e - i * C;
    long long si = i;
    if (bricked) {
      const int z = (int)(i % (unsigned)Z), y = (int)((i / (unsigned)Z) % (unsigned)Y), x = (int)(i / ((unsigned)Y * (unsigned)Z));
      si = brick_slot(x, y, z, Y, Z);
    }
    const float v = gpacked[si * C + ch];
    if (ch < F) {
      if (d_feat) d_feat[i * F + ch] = accumulate ? d_feat[i * F + ch] + v : v;
    } else if (d_dens) {
      const float gval = v * pre_activate_grad(pre_act, dens[i], scale);
      d_dens[i] = accumulate ? d_dens[i] + gval : gval;
    }
  }
}

// r04: the same two conversions with 16-byte accesses on BOTH layouts.  A block takes chunks of kWideChunk voxels through LDS:
// the chunk is contiguous in either layout (64 x F floats of `feat`, 64 x C floats of `packed`; both a multiple of 16 bytes),
// only the interleave differs, and that happens on the 4-byte LDS side.  The one-element-per-thread kernels above moved
// 2.3 - 2.8 TB/s at 49 channels (4 bytes per lane and an integer division per element).
// Whole chunks only; the launcher sends the tail (nvox % 64 voxels) and bricked gradient buffers through the kernels above.
#ifndef VOXE_WIDE16
#define VOXE_WIDE16 1
#endif
constexpr int kWideChunk = 64;
template <int C>
__global__ __launch_bounds__(256) void pack_grid_wide16_kernel(const float* __restrict__ dens, const float* __restrict__ feat,
                                                               float* __restrict__ packed, long long nchunks, float scale,
                                                               int pre_act) {
  constexpr int F = C - 1, V = kWideChunk;
  __shared__ float4 buf4[V * C / 4];
  float* const buf = reinterpret_cast<float*>(buf4);
  const int tid = threadIdx.x;
  for (long long ck = blockIdx.x; ck < nchunks; ck += gridDim.x) {
    const float4* __restrict__ f4 = reinterpret_cast<const float4*>(feat + ck * (V * F));
    for (int q = tid; q < V * F / 4; q += 256) {
      const float4 t = f4[q];
      const float e[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = 4 * q + u, i = idx / F;
        buf[i * C + (idx - i * F)] = e[u];
      }
    }
    if (tid < V) buf[tid * C + F] = pre_activate(pre_act, dens[ck * V + tid], scale);
    __syncthreads();
    float4* __restrict__ p4 = reinterpret_cast<float4*>(packed + ck * (V * C));
    for (int q = tid; q < V * C / 4; q += 256) p4[q] = buf4[q];
    __syncthreads();
  }
}

template <int C>
__global__ __launch_bounds__(256) void unpack_grad_wide16_kernel(const float* __restrict__ gpacked,
                                                                 const float* __restrict__ dens, float* __restrict__ d_dens,
                                                                 float* __restrict__ d_feat, long long nchunks, float scale,
                                                                 int pre_act, int accumulate) {
  constexpr int F = C - 1, V = kWideChunk;
  __shared__ float4 buf4[V * C / 4];
  float* const buf = reinterpret_cast<float*>(buf4);
  const int tid = threadIdx.x;
  for (long long ck = blockIdx.x; ck < nchunks; ck += gridDim.x) {
    const float4* __restrict__ g4 = reinterpret_cast<const float4*>(gpacked + ck * (V * C));
    for (int q = tid; q < V * C / 4; q += 256) buf4[q] = g4[q];
    __syncthreads();
    if (d_feat) {
      float4* __restrict__ o4 = reinterpret_cast<float4*>(d_feat + ck * (V * F));
      for (int q = tid; q < V * F / 4; q += 256) {
        float e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = 4 * q + u, i = idx / F;
          e[u] = buf[i * C + (idx - i * F)];
        }
        float4 t = make_float4(e[0], e[1], e[2], e[3]);
        if (accumulate) {
          const float4 old = o4[q];
          t = make_float4(old.x + t.x, old.y + t.y, old.z + t.z, old.w + t.w);
        }
        o4[q] = t;
      }
    }
    if (d_dens && tid < V) {
      const long long i = ck * V + tid;
      const float gval = buf[tid * C + F] * pre_activate_grad(pre_act, dens[i], scale);
      d_dens[i] = accumulate ? d_dens[i] + gval : gval;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Fused optimiser step: un-pack the gradient (+ chain rule of the density pre-activation), Adam on both parameter
// tensors, re-pack the updated grid, clear the gradient -- one streaming pass instead of four
// (unpack 131 MB + Adam 459 MB + pack 131 MB + memset 65 MB -> 590 MB at 160^3).  Arithmetic of unpack_grad_kernel and
// adam_kernel (voxe_grid_ops.hip), operation for operation.
// ------------------------------------------------------------------------------------------------
struct AdamHyper {
  float step_size, bc2_sqrt, beta1, beta2, eps;
};
#ifndef VOXE_GA_BLOCKS
#define VOXE_GA_BLOCKS 16384
#endif
__device__ __forceinline__ float adam_update(float p, float gi, float& m, float& v, const AdamHyper& h) {
  const float mi = m + (gi - m) * (1.0f - h.beta1);
  const float vi = v * h.beta2 + ((1.0f - h.beta2) * gi) * gi;
  const float denom = sqrtf(vi) / h.bc2_sqrt + h.eps;
  m = mi;
  v = vi;
  return p - h.step_size * (mi / denom);
}

// feature-correlation gradient of one voxel (feature_correlation_kernel's expressions, voxe_grid_ops.hip; sds_trainer.py:526-534):
// p / r = the voxel's features / reference features, fk = 2 x weight
template <int F>
__device__ __forceinline__ void featcorr_term(const float (&p)[F], const float (&r)[F], float fk, float (&out)[F]) {
  float sg[F], D = 0.0f;
#pragma unroll
  for (int f = 0; f < F; ++f) { sg[f] = 1.0f / (1.0f + expf(-p[f])); D += sg[f] - 1.0f / (1.0f + expf(-r[f])); }
#pragma unroll
  for (int f = 0; f < F; ++f) out[f] = (fk * D) * ((1.0f - sg[f]) * sg[f]);
}

// density-correlation gradient of one voxel (dcl_grad_kernel's expression; stats = mean a, mean b, k1, k2 with the weight folded in)
// r06: VOXE_DREG_L2 / L1 (sds_trainer.py:494-503): (a - b) (2 w / n)  /  sign(a - b) (w / n), no statistics
__device__ __forceinline__ float dcl_term(const DclTerm& t, float a, long long i) {
  if (t.kind != VOXE_DREG_CORRELATION) {
    const float d = a - t.b[i];
    return t.kind == VOXE_DREG_L2 ? d * t.k : (d > 0.0f ? t.k : (d < 0.0f ? -t.k : 0.0f));
  }
  const float ma = (float)t.stats[0], mb = (float)t.stats[1], k1 = (float)t.stats[2], k2 = (float)t.stats[3];
  const float A = a - ma, B = t.b[i] - mb;
  return A * k2 - B * k1;
}

template <int C>
__global__ __launch_bounds__(256) void grid_adam_kernel(float* __restrict__ gpacked, float* __restrict__ dens,
                                                        float* __restrict__ feat, const float* __restrict__ extra_d,
                                                        const float* __restrict__ extra_f, float* __restrict__ m_d,
                                                        float* __restrict__ v_d, float* __restrict__ m_f,
                                                        float* __restrict__ v_f, float* __restrict__ packed,
                                                        long long vox_begin, long long vox_end, float scale,
                                                        int pre_act, int bricked, int Y, int Z, AdamHyper h_d, AdamHyper h_f,
                                                        DclTerm dcl) {
  constexpr int F = C - 1;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = vox_begin + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < vox_end; i += stride) {
    long long si = i;
    if (bricked) {
      const int z = (int)(i % Z), y = (int)((i / Z) % Y), x = (int)(i / ((long long)Y * Z));
      si = brick_slot(x, y, z, Y, Z);
    }
    float g[C];
    if constexpr (C == 4) {
      const float4 t = reinterpret_cast<const float4*>(gpacked)[si];
      g[0] = t.x; g[1] = t.y; g[2] = t.z; g[3] = t.w;
      reinterpret_cast<float4*>(gpacked)[si] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    } else if constexpr (C == 2) {
      const float2 t = reinterpret_cast<const float2*>(gpacked)[si];
      g[0] = t.x; g[1] = t.y;
      reinterpret_cast<float2*>(gpacked)[si] = make_float2(0.0f, 0.0f);
    } else {
#pragma unroll
      for (int f = 0; f < C; ++f) { g[f] = gpacked[si * C + f]; gpacked[si * C + f] = 0.0f; }
    }
    float out[C];
    float fg[F];
#pragma unroll
    for (int f = 0; f < F; ++f) fg[f] = 0.0f;
    if (dcl.fref && m_f) {   // (the term is evaluated on the parameters the step starts from, like every gradient)
      float pf[F], rf[F];
#pragma unroll
      for (int f = 0; f < F; ++f) { pf[f] = feat[i * F + f]; rf[f] = dcl.fref[i * F + f]; }
      featcorr_term<F>(pf, rf, dcl.fk, fg);
    }
#pragma unroll
    for (int f = 0; f < F; ++f) {
      const long long j = i * F + f;
      float p = feat[j];
      if (m_f) {
        float gi = extra_f ? g[f] + extra_f[j] : g[f];
        if (dcl.fref) gi += fg[f];
        float m = m_f[j], v = v_f[j];
        p = adam_update(p, gi, m, v, h_f);
        feat[j] = p; m_f[j] = m; v_f[j] = v;
      }
      out[f] = p;
    }
    float d = dens[i];
    if (m_d) {
      const float gd = g[F] * pre_activate_grad(pre_act, d, scale);
      float gi = extra_d ? gd + extra_d[i] : gd;
      if (dcl.b) gi += dcl_term(dcl, d, i);
      float m = m_d[i], v = v_d[i];
      d = adam_update(d, gi, m, v, h_d);
      dens[i] = d; m_d[i] = m; v_d[i] = v;
    }
    out[F] = pre_activate(pre_act, d, scale);
    if constexpr (C == 4) {
      reinterpret_cast<float4*>(packed)[i] = make_float4(out[0], out[1], out[2], out[3]);
    } else if constexpr (C == 2) {
      reinterpret_cast<float2*>(packed)[i] = make_float2(out[0], out[1]);
    } else {
#pragma unroll
      for (int f = 0; f < C; ++f) packed[i * C + f] = out[f];
    }
  }
}

#ifndef VOXE_GA_FLIP
#define VOXE_GA_FLIP 1
#endif
// C == 4, linear gradient layout, every global access a fully coalesced wave instruction (r03): a wave takes 256 consecutive
// voxels; lane l owns voxels base + l + 64 k (k = 0..3), so the float4 streams (packed gradient, packed grid) and the per-voxel
// scalars (density, its moments) are contiguous across the lanes of every instruction; the [N,3] streams (features, their
// moments) are read as three contiguous float4 instructions per wave and handed to their voxels' lanes through a wave-private
// LDS buffer (stride-3 reads: conflict free), and written back the same way.  Measured against the alternatives (same arithmetic
// per element): one voxel per thread (nine 4-byte accesses of stride 12 for the [N,3] streams) 0.130 ms; four voxels per thread with
// 16-byte chunks at 48 / 64-byte lane strides 0.110 - 0.118 ms and 20 % more bytes written than the streams hold (partial lines,
// PMC WRITE_SIZE 394 MB); this kernel 0.085 ms, 328 MB written.  The sweep direction alternates with the step (the tail of one
// step's sweep is the head of the next in the Infinity Cache: -5 % at 100x100, +-0 at 400x400); non-temporal loads / stores of
// parameters and moments were 25 % slower (the moments live partly in the Infinity Cache from step to step).
#ifndef VOXE_GA_V5
#define VOXE_GA_V5 1
#endif
__device__ __forceinline__ void wave_lds_order() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// FEATCORR (r06): with the feature-correlation term compiled in, the kernel needs 170 registers -- 2 waves per SIMD instead of the 3
// that 150 leave -- and the plain step ran 103 instead of 86 us (rocprofv3 averages r05 -> r06): the term is its own instantiation.
template <bool FEATCORR>
__global__ __launch_bounds__(256) void grid_adam_v5_kernel(float4* __restrict__ gpacked, float* __restrict__ dens,
                                                           float4* __restrict__ feat, const float* __restrict__ extra_d,
                                                           const float4* __restrict__ extra_f, float* __restrict__ m_d,
                                                           float* __restrict__ v_d, float4* __restrict__ m_f,
                                                           float4* __restrict__ v_f, float4* __restrict__ packed,
                                                           long long vox_begin, int nchunks, float scale, int pre_act,
                                                           AdamHyper h_d, AdamHyper h_f, int flip, DclTerm dcl) {
  __shared__ float4 lds4[4][3][192];   // [wave][buffer][256 voxels x 3 floats]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nwaves = gridDim.x * 4;
  for (int c0 = blockIdx.x * 4 + wave; c0 < nchunks; c0 += nwaves) {
    const int c = flip ? nchunks - 1 - c0 : c0;
    const long long vb = vox_begin + (long long)c * 256;      // first voxel of the chunk (a multiple of 4)
    const long long f4 = vb * 3 / 4;                           // first float4 of the chunk in an [N,3] stream
    float4 g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = gpacked[vb + k * 64 + lane];
    // [N,3] streams -> LDS (three coalesced float4 instructions each)
    float4 (*buf)[192] = lds4[wave];
#pragma unroll
    for (int k = 0; k < 3; ++k) buf[0][k * 64 + lane] = feat[f4 + k * 64 + lane];
    if (m_f) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { buf[1][k * 64 + lane] = m_f[f4 + k * 64 + lane]; buf[2][k * 64 + lane] = v_f[f4 + k * 64 + lane]; }
    }
    float d[4], md[4], vd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      d[k] = dens[vb + k * 64 + lane];
      if (m_d) { md[k] = m_d[vb + k * 64 + lane]; vd[k] = v_d[vb + k * 64 + lane]; }
    }
    wave_lds_order();
    float p[12], m[12], v[12];
    const float* b0 = reinterpret_cast<const float*>(buf[0]);
    const float* b1 = reinterpret_cast<const float*>(buf[1]);
    const float* b2 = reinterpret_cast<const float*>(buf[2]);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        const int i = (k * 64 + lane) * 3 + f;
        p[k * 3 + f] = b0[i];
        if (m_f) { m[k * 3 + f] = b1[i]; v[k * 3 + f] = b2[i]; }
      }
    wave_lds_order();
    float fg[FEATCORR ? 12 : 1];
    if constexpr (FEATCORR) if (m_f && dcl.fref) {   // feature-correlation term (r06; weight 0 by default): the reference features through buffer 0
      const float4* __restrict__ fr4 = reinterpret_cast<const float4*>(dcl.fref);
#pragma unroll
      for (int k = 0; k < 3; ++k) buf[0][k * 64 + lane] = fr4[f4 + k * 64 + lane];
      wave_lds_order();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float pk[3] = {p[k * 3], p[k * 3 + 1], p[k * 3 + 2]};
        const float rk[3] = {b0[(k * 64 + lane) * 3], b0[(k * 64 + lane) * 3 + 1], b0[(k * 64 + lane) * 3 + 2]};
        float o[3];
        featcorr_term<3>(pk, rk, dcl.fk, o);
        fg[k * 3] = o[0]; fg[k * 3 + 1] = o[1]; fg[k * 3 + 2] = o[2];
      }
      wave_lds_order();
    }
    if (m_f) {
      if (extra_f) {   // regulariser gradient of the features (rare path): through buffer 0
#pragma unroll
        for (int k = 0; k < 3; ++k) buf[0][k * 64 + lane] = extra_f[f4 + k * 64 + lane];
        wave_lds_order();
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int f = 0; f < 3; ++f) {
          const float gj = f == 0 ? g[k].x : (f == 1 ? g[k].y : g[k].z);
          float gi = extra_f ? gj + b0[(k * 64 + lane) * 3 + f] : gj;
          if constexpr (FEATCORR) { if (dcl.fref) gi += fg[k * 3 + f]; }
          p[k * 3 + f] = adam_update(p[k * 3 + f], gi, m[k * 3 + f], v[k * 3 + f], h_f);
        }
      wave_lds_order();
      // back through LDS: every lane writes its 12 values of each stream, the wave stores three float4 instructions
      float* w0 = reinterpret_cast<float*>(buf[0]);
      float* w1 = reinterpret_cast<float*>(buf[1]);
      float* w2 = reinterpret_cast<float*>(buf[2]);
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int f = 0; f < 3; ++f) {
          const int i = (k * 64 + lane) * 3 + f;
          w0[i] = p[k * 3 + f]; w1[i] = m[k * 3 + f]; w2[i] = v[k * 3 + f];
        }
      wave_lds_order();
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        feat[f4 + k * 64 + lane] = buf[0][k * 64 + lane];
        m_f[f4 + k * 64 + lane] = buf[1][k * 64 + lane];
        v_f[f4 + k * 64 + lane] = buf[2][k * 64 + lane];
      }
      wave_lds_order();   // (the buffers are reused by the next chunk)
    }
    if (m_d) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gd = g[k].w * pre_activate_grad(pre_act, d[k], scale);
        float gi = extra_d ? gd + extra_d[vb + k * 64 + lane] : gd;
        if (dcl.b) gi += dcl_term(dcl, d[k], vb + k * 64 + lane);
        d[k] = adam_update(d[k], gi, md[k], vd[k], h_d);
        dens[vb + k * 64 + lane] = d[k];
        m_d[vb + k * 64 + lane] = md[k];
        v_d[vb + k * 64 + lane] = vd[k];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      gpacked[vb + k * 64 + lane] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      packed[vb + k * 64 + lane] = make_float4(p[k * 3], p[k * 3 + 1], p[k * 3 + 2], pre_activate(pre_act, d[k], scale));
    }
  }
}

// the same step with one thread per ELEMENT of the packed arrays (wide texels, see pack_grid_wide_kernel)
template <int C>
__global__ __launch_bounds__(256) void grid_adam_wide_kernel(float* __restrict__ gpacked, float* __restrict__ dens,
                                                             float* __restrict__ feat, const float* __restrict__ extra_d,
                                                             const float* __restrict__ extra_f, float* __restrict__ m_d,
                                                             float* __restrict__ v_d, float* __restrict__ m_f,
                                                             float* __restrict__ v_f, float* __restrict__ packed,
                                                             long long vox_begin, long long vox_end, float scale,
                                                             int pre_act, int bricked, int Y, int Z, AdamHyper h_d, AdamHyper h_f,
                                                             DclTerm dcl) {
  constexpr int F = C - 1;
  const unsigned e_end = (unsigned)(vox_end * C), stride = gridDim.x * blockDim.x;
  for (unsigned e = (unsigned)(vox_begin * C) + blockIdx.x * blockDim.x + threadIdx.x; e < e_end; e += stride) {
    const unsigned i = e / C, ch = e - i * C;
    long long si = i;
    if (bricked) {
      const int z = (int)(i % (unsigned)Z), y = (int)((i / (unsigned)Z) % (unsigned)Y), x = (int)(i / ((unsigned)Y * (unsigned)Z));
      si = brick_slot(x, y, z, Y, Z);
    }
    const float g = gpacked[si * C + ch];
    gpacked[si * C + ch] = 0.0f;
    if (ch < F) {
      const unsigned j = i * F + ch;
      float p = feat[j];
      if (m_f) {
        const float gi = extra_f ? g + extra_f[j] : g;
        float m = m_f[j], v = v_f[j];
        p = adam_update(p, gi, m, v, h_f);
        feat[j] = p; m_f[j] = m; v_f[j] = v;
      }
      packed[e] = p;
    } else {
      float d = dens[i];
      if (m_d) {
        const float gd = g * pre_activate_grad(pre_act, d, scale);
        float gi = extra_d ? gd + extra_d[i] : gd;
        if (dcl.b) gi += dcl_term(dcl, d, i);
        float m = m_d[i], v = v_d[i];
        d = adam_update(d, gi, m, v, h_d);
        dens[i] = d; m_d[i] = m; v_d[i] = v;
      }
      packed[e] = pre_activate(pre_act, d, scale);
    }
  }
}

// r04: the wide-texel step with 16-byte accesses (the scheme of pack_grid_wide16_kernel): a block takes chunks of kWideChunk voxels;
// the chunk's packed gradient goes through LDS, the feature-side arrays (feat, exp_avg, exp_avg_sq, extra gradient: [voxel][F], contiguous
// per chunk) move as float4, the updated texels are assembled in the same LDS buffer and leave as float4.  adam_update() per element,
// operation for operation what grid_adam_wide_kernel does (9 four-byte accesses per element there: 1.5 ms for a 28-channel 160^3 grid).
template <int C>
__global__ __launch_bounds__(256) void grid_adam_wide16_kernel(float* __restrict__ gpacked, float* __restrict__ dens,
                                                               float* __restrict__ feat, const float* __restrict__ extra_d,
                                                               const float* __restrict__ extra_f, float* __restrict__ m_d,
                                                               float* __restrict__ v_d, float* __restrict__ m_f,
                                                               float* __restrict__ v_f, float* __restrict__ packed,
                                                               long long vox_begin, long long nchunks, float scale, int pre_act,
                                                               AdamHyper h_d, AdamHyper h_f, DclTerm dcl) {
  constexpr int F = C - 1, V = kWideChunk;
  __shared__ float4 buf4[V * C / 4];
  float* const buf = reinterpret_cast<float*>(buf4);
  const int tid = threadIdx.x;
  for (long long ck = blockIdx.x; ck < nchunks; ck += gridDim.x) {
    const long long v0 = vox_begin + ck * V;
    float4* __restrict__ g4 = reinterpret_cast<float4*>(gpacked + v0 * C);
    for (int q = tid; q < V * C / 4; q += 256) {
      buf4[q] = g4[q];
      g4[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    __syncthreads();
    float4* __restrict__ f4 = reinterpret_cast<float4*>(feat + v0 * F);
    for (int q = tid; q < V * F / 4; q += 256) {
      const float4 pf = f4[q];
      float p[4] = {pf.x, pf.y, pf.z, pf.w};
      int slot[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = 4 * q + u, i = idx / F;
        slot[u] = i * C + (idx - i * F);
      }
      if (m_f) {
        float4* __restrict__ m4 = reinterpret_cast<float4*>(m_f + v0 * F);
        float4* __restrict__ v4 = reinterpret_cast<float4*>(v_f + v0 * F);
        const float4 mm = m4[q], vv = v4[q];
        float m[4] = {mm.x, mm.y, mm.z, mm.w}, v[4] = {vv.x, vv.y, vv.z, vv.w};
        float ex[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (extra_f) {
          const float4 e4 = reinterpret_cast<const float4*>(extra_f + v0 * F)[q];
          ex[0] = e4.x; ex[1] = e4.y; ex[2] = e4.z; ex[3] = e4.w;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float g = buf[slot[u]];
          const float gi = extra_f ? g + ex[u] : g;
          p[u] = adam_update(p[u], gi, m[u], v[u], h_f);
        }
        f4[q] = make_float4(p[0], p[1], p[2], p[3]);
        m4[q] = make_float4(m[0], m[1], m[2], m[3]);
        v4[q] = make_float4(v[0], v[1], v[2], v[3]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) buf[slot[u]] = p[u];    // (every slot is read above by the thread that overwrites it)
    }
    if (tid < V) {
      const long long i = v0 + tid;
      float d = dens[i];
      if (m_d) {
        const float gd = buf[tid * C + F] * pre_activate_grad(pre_act, d, scale);
        float gi = extra_d ? gd + extra_d[i] : gd;
        if (dcl.b) gi += dcl_term(dcl, d, i);
        float m = m_d[i], v = v_d[i];
        d = adam_update(d, gi, m, v, h_d);
        dens[i] = d; m_d[i] = m; v_d[i] = v;
      }
      buf[tid * C + F] = pre_activate(pre_act, d, scale);
    }
    __syncthreads();
    float4* __restrict__ p4 = reinterpret_cast<float4*>(packed + v0 * C);
    for (int q = tid; q < V * C / 4; q += 256) p4[q] = buf4[q];
    __syncthreads();
  }
}

template <int C>
static void launch_grid_adam_t(const VoxeGridDesc* gd, bool bricked, int x_begin, int x_end, float* gpacked, const float* extra_d,
                               const float* extra_f, float* m_d, float* v_d, float* m_f, float* v_f, AdamHyper h_d,
                               AdamHyper h_f, float* packed_out, hipStream_t st, int flip, DclTerm dcl) {
  const long long plane = (long long)gd->Y * gd->Z, nvox = (x_end - x_begin) * plane;
  if constexpr (C > 4) {
    long long vb = x_begin * plane;
    const long long ve = x_end * plane;
    auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (VOXE_WIDE16 && !bricked && vb % 4 == 0 && ve - vb >= kWideChunk && a16(gpacked) && a16(gd->features) && a16(m_f) && a16(v_f) &&
        a16(extra_f) && a16(packed_out)) {
      const long long nchunks = (ve - vb) / kWideChunk;
      grid_adam_wide16_kernel<C><<<(int)(nchunks < 4096 ? nchunks : 4096), 256, 0, st>>>(
          gpacked, const_cast<float*>(gd->densities), const_cast<float*>(gd->features), extra_d, extra_f, m_d, v_d, m_f, v_f,
          packed_out, vb, nchunks, gd->density_scale, gd->density_pre_act, h_d, h_f, dcl);
      vb += nchunks * kWideChunk;
    }
    if (vb < ve) {   // bricked gradient buffers, unaligned slabs, the last (ve - vb) % 64 voxels
      const long long n = (ve - vb) * C;
      const int nbw = (int)((n + 255) / 256 < 4 * VOXE_GA_BLOCKS ? (n + 255) / 256 : 4 * VOXE_GA_BLOCKS);
      grid_adam_wide_kernel<C><<<nbw, 256, 0, st>>>(gpacked, const_cast<float*>(gd->densities), const_cast<float*>(gd->features),
                                                    extra_d, extra_f, m_d, v_d, m_f, v_f, packed_out, vb, ve, gd->density_scale,
                                                    gd->density_pre_act, bricked ? 1 : 0, gd->Y, gd->Z, h_d, h_f, dcl);
    }
    return;
  }
  if constexpr (C == 4 && VOXE_GA_V5) {
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const long long vb = x_begin * plane, ve = x_end * plane;
    if (!bricked && vb % 4 == 0 && al16(gpacked) && al16(gd->densities) && al16(gd->features) && al16(extra_d) &&
        al16(extra_f) && al16(m_d) && al16(v_d) && al16(m_f) && al16(v_f) && al16(packed_out) && al16(dcl.fref)) {
      if ((ve - vb) >= 256) {
        const long long nchunks = (ve - vb) / 256, tail = vb + nchunks * 256;
        const int nb5 = (int)((nchunks + 3) / 4 < VOXE_GA_BLOCKS ? (nchunks + 3) / 4 : VOXE_GA_BLOCKS);
#define VOXE_GA5(FC)                                                                                                              \
        grid_adam_v5_kernel<FC><<<nb5, 256, 0, st>>>(                                                                             \
            reinterpret_cast<float4*>(gpacked), const_cast<float*>(gd->densities),                                                \
            reinterpret_cast<float4*>(const_cast<float*>(gd->features)), extra_d, reinterpret_cast<const float4*>(extra_f), m_d, v_d, \
            reinterpret_cast<float4*>(m_f), reinterpret_cast<float4*>(v_f), reinterpret_cast<float4*>(packed_out), vb, (int)nchunks, \
            gd->density_scale, gd->density_pre_act, h_d, h_f, VOXE_GA_FLIP ? flip : 0, dcl)
        if (dcl.fref && m_f) VOXE_GA5(true); else VOXE_GA5(false);
#undef VOXE_GA5
        if (tail < ve)   // fewer than 256 voxels left: the per-voxel kernel
          grid_adam_kernel<C><<<1, 256, 0, st>>>(gpacked, const_cast<float*>(gd->densities), const_cast<float*>(gd->features), extra_d,
                                                 extra_f, m_d, v_d, m_f, v_f, packed_out, tail, ve, gd->density_scale,
                                                 gd->density_pre_act, 0, gd->Y, gd->Z, h_d, h_f, dcl);
        return;
      }
    }
  }
  const int nb = (int)((nvox + 255) / 256 < VOXE_GA_BLOCKS ? (nvox + 255) / 256 : VOXE_GA_BLOCKS);
  grid_adam_kernel<C><<<nb, 256, 0, st>>>(gpacked, const_cast<float*>(gd->densities), const_cast<float*>(gd->features),
                                          extra_d, extra_f, m_d, v_d, m_f, v_f, packed_out, x_begin * plane, x_end * plane, gd->density_scale,
                                          gd->density_pre_act, bricked ? 1 : 0, gd->Y, gd->Z, h_d, h_f, dcl);
}

bool launch_grid_adam(const VoxeGridDesc* gd, bool bricked, int x_begin, int x_end, float* gpacked, const float* extra_d, const float* extra_f,
                      float* m_d, float* v_d, float* m_f, float* v_f, float lr, float beta1, float beta2, float eps,
                      long long step_d, long long step_f, float* packed_out, hipStream_t st, DclTerm dcl) {
  // torch.optim.Adam keeps one step counter PER PARAMETER: the two tensors' bias corrections may differ (a tensor that
  // skipped a step, e.g. a regulariser-only iteration on the densities)
  auto hyper = [&](long long step) {
    const double bc1 = 1.0 - pow((double)beta1, (double)step);   // same host arithmetic as launch_adam()
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    return AdamHyper{(float)((double)lr / bc1), (float)sqrt(bc2), beta1, beta2, eps};
  };
  const AdamHyper h_d = hyper(step_d), h_f = hyper(step_f);
  const int flip = (int)(step_d & 1);   // alternate sweep direction: the tail of one step's sweep is the head of the next (Infinity Cache)
  switch (gd->F + 1) {
    case 2: launch_grid_adam_t<2>(gd, bricked, x_begin, x_end, gpacked, extra_d, extra_f, m_d, v_d, m_f, v_f, h_d, h_f, packed_out, st, flip, dcl); return true;
    case 4: launch_grid_adam_t<4>(gd, bricked, x_begin, x_end, gpacked, extra_d, extra_f, m_d, v_d, m_f, v_f, h_d, h_f, packed_out, st, flip, dcl); return true;
    case 13: launch_grid_adam_t<13>(gd, bricked, x_begin, x_end, gpacked, extra_d, extra_f, m_d, v_d, m_f, v_f, h_d, h_f, packed_out, st, flip, dcl); return true;
    case 28: launch_grid_adam_t<28>(gd, bricked, x_begin, x_end, gpacked, extra_d, extra_f, m_d, v_d, m_f, v_f, h_d, h_f, packed_out, st, flip, dcl); return true;
    case 49: launch_grid_adam_t<49>(gd, bricked, x_begin, x_end, gpacked, extra_d, extra_f, m_d, v_d, m_f, v_f, h_d, h_f, packed_out, st, flip, dcl); return true;
  }
  return false;
}

// ------------------------------------------------------------------------------------------------
// Forward
// ------------------------------------------------------------------------------------------------
template <int COUT, int NCM, int NCU>
__global__ __launch_bounds__(256) void render_fwd_kernel(DevGrid g, DevCfg c,
                                                         const float* __restrict__ packed,
                                                         const float* __restrict__ rays_o,
                                                         const float* __restrict__ rays_d,
                                                         const float* __restrict__ jitter,
                                                         float* __restrict__ colour,
                                                         float* __restrict__ depth,
                                                         float* __restrict__ acc,
                                                         float* __restrict__ disparity,
                                                         float* __restrict__ ray_state) {
  long long r;
  if (!map_ray(c, r)) return;
  RayCtx<COUT, NCM, NCU> rc;
  rc.init(g, c, r, rays_o, rays_d, jitter);

  float csum[COUT];
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) csum[ch] = 0.0f;
  float asum = 0.0f, dsum = 0.0f, T = 1.0f;

  // depth-segment states for the segmented backward: state BEFORE sample b * kSegLen
  const int nbound = ray_state ? num_segments(c.S, c.seg_len) - 1 : 0;
  int nextb = 1;
  auto save_state = [&](int b) {
    constexpr int NC = COUT + 3;
    ray_state[ray_state_index(b, 0, NC, c.R, r)] = T;
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) ray_state[ray_state_index(b, 1 + ch, NC, c.R, r)] = csum[ch];
    ray_state[ray_state_index(b, 1 + COUT, NC, c.R, r)] = asum;
    ray_state[ray_state_index(b, 2 + COUT, NC, c.R, r)] = dsum;
  };

  if (rc.k_lo <= rc.k_hi) {
    float z_next = rc.dg.z(rc.k_lo);
    for (int k = rc.k_lo; k <= rc.k_hi; ++k) {
      while (nextb <= nbound && nextb * c.seg_len <= k) { save_state(nextb); ++nextb; }
      const float z = z_next;
      const bool last = (k == c.S - 1);
      if (!last) z_next = rc.dg.z(k + 1);
      float p[3];
      rc.point(z, p);
      Footprint fp;
      footprint(g, p, fp);
      if (!fp.inside) continue;  // sigma = 0 -> alpha = 0 -> w = 0, T unchanged (process.py:83)
      Cell cell;
      make_cell_fast(g, fp, cell);
      float v, rad[COUT];
      gather<COUT, NCM, NCU>(g, packed, cell, rc.basis, v, rad);
      const float sigma = post_activate(g.post_act, v);
      // accumulate.py:49-55,63-67
      const float dl = last ? kInfinity : (z_next - z);
      const float delta = dl * rc.dnorm;
      const float e = fast_exp(-(sigma * delta));
      const float alpha = 1.0f - e;
      const float om = 1.0f - alpha;
      const float w = alpha * T;
      T = T * om;
#pragma unroll
      for (int ch = 0; ch < COUT; ++ch) csum[ch] = csum[ch] + sigmoidf(rad[ch]) * w;
      asum = asum + w;
      dsum = dsum + z * w;
    }
  }
  while (nextb <= nbound) { save_state(nextb); ++nextb; }  // boundaries behind the last sample: final state
  // the backward wants SUFFIX sums (what lies at and behind the boundary), see render_fwd_combine_kernel: final - prefix
  // (this single-march forward only runs with early termination or without the segment buffer; the segmented forward
  // sums its suffixes back to front instead)
  for (int b = 1; b <= nbound; ++b) {
    constexpr int NC = COUT + 3;
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) {
      const long long i = ray_state_index(b, 1 + ch, NC, c.R, r);
      ray_state[i] = csum[ch] - ray_state[i];
    }
    const long long ia = ray_state_index(b, 1 + COUT, NC, c.R, r), id = ray_state_index(b, 2 + COUT, NC, c.R, r);
    ray_state[ia] = asum - ray_state[ia];
    ray_state[id] = dsum - ray_state[id];
  }
  // accumulate.py:77-88
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) {
    float col = csum[ch];
    if (c.white) {
      float bk = 1.0f - asum;
      if (c.attn) bk = bk * 0.0f;
      col = col + bk;
    }
    if (colour) colour[r * COUT + ch] = col;
  }
  if (depth) depth[r] = dsum;
  if (acc) acc[r] = asum;
  if (disparity) {
    const float q = dsum / asum;
    const float m = (q != q) ? q : (q > kZeroPlus ? q : kZeroPlus);  // torch.maximum keeps NaN
    disparity[r] = 1.0f / m;
  }
}

// ------------------------------------------------------------------------------------------------
// Depth-segmented forward: one 400x400 image is only ~2500 waves, too few to hide the gather latency on
// 1024 SIMDs, and one thread marching all S samples is a long dependent chain.  Compositing is associative:
// segment s (samples [s*kSegLen, (s+1)*kSegLen)) computes, with a LOCAL transmittance starting at 1,
//   Tseg = prod (1 - alpha_k),  csum = sum col_k alpha_k Tloc_k,  asum = sum alpha_k Tloc_k,  dsum = sum z_k alpha_k Tloc_k
// in its own thread (8x more waves), and render_fwd_combine_kernel folds the segments front to back:
//   T_start(s) = prod_{s' < s} Tseg(s'),  colour = sum_s T_start(s) csum(s), ...
// The combine pass also emits the per-ray segment-start states the segmented backward consumes.
// (The forward integrates every sample whatever term_eps says: the switch only truncates gradients, see voxe.h.)
// Tried and measured as nulls (r03, 400x400: 0.246 ms either way): two samples per loop trip with both gathers (16 texel
// loads) in flight before either sample's arithmetic -- the kernel is not short of memory-level parallelism inside a wave;
// fewer registers for more resident waves (launch bounds 6 / 8: 0.284 / 0.438 ms, spills).  PMC: VALU issue 0.47, 3.4 of 5
// possible waves per SIMD resident on average (12 k non-empty one-wave blocks = 2.3 rounds), waves wait 64 % of their time.
// segbuf layout: [segment][component][ray], components (Tseg, csum[COUT], asum, dsum).
// ------------------------------------------------------------------------------------------------
template <int COUT, int NCM, int NCU>
// one-wave blocks (an 8x8 pixel tile each): 2-3 % faster than 256-thread blocks at 400x400, 33 % at 100x100 (finer
// scheduling granularity; the kernel has no block-level cooperation)
#ifndef VOXE_FWD_LB
#define VOXE_FWD_LB 1
#endif
__global__ __launch_bounds__(64, VOXE_FWD_LB) void render_fwd_seg_kernel(DevGrid g, DevCfg c, int fseg,
                                                             const float* __restrict__ packed,
                                                             const float* __restrict__ rays_o,
                                                             const float* __restrict__ rays_d,
                                                             const float* __restrict__ jitter,
                                                             float* __restrict__ segbuf,
                                                             float4* __restrict__ sample_fwd) {
  constexpr int NC = COUT + 3;
  const int nseg = num_segments(c.S, c.seg_len);
  // one thread = `fseg` consecutive depth segments of one ray (fseg = 1: finest split, used for small images)
  const int ncoarse = (nseg + fseg - 1) / fseg;
  // Block order: SEGMENT-MAJOR -- all ray blocks at coarse segment 0, then all at segment 1, ...  The first and last
  // depth segments lie mostly outside the AABB and retire at once; if the segments of one ray block sat on consecutive
  // blocks, "empty" and "full" blocks would alternate with period ncoarse, which aliases with the placement of block b
  // on XCD b % 8 and round-robin on its 32 CUs (see render_bwd_tile_kernel).
  const int nrb = gridDim.x / ncoarse;  // ray blocks (a multiple of 8)
  const int cseg = blockIdx.x / nrb, rb = blockIdx.x - cseg * nrb;
  long long r;
  long long tile_lane = -1;   // (tile * nseg) * seg_len * 64 + lane: base of this lane's slots in sample_fwd (image order only)
  {  // one wave = one 8x8 pixel tile (image order) or 64 consecutive rays
    const int lane = threadIdx.x;
    if (c.image_width > 0) {
      const int W = c.image_width;
      const int ntx = (W + 7) >> 3, nty = (int)tile_rows_total(c, 8);
      const int t = logical_tile_of(c, rb, nrb, ntx, nty);
      if (t < 0) return;
      tile_lane = (long long)t * nseg * c.seg_len * 64 + lane;
      const int ty = t / ntx, tx = t - ty * ntx;
      if (!tile_pixel_ray(c, ty, lane >> 3, (tx << 3) + (lane & 7), 8, r)) return;
    } else {
      const int nt = (int)((c.R + 63) / 64);
      const int t = logical_tile_of(c, rb, nrb, 1, nt);
      if (t < 0) return;
      r = (long long)t * 64 + lane;
      if (r >= c.R) return;
    }
  }
  RayCtx<COUT, NCM, NCU> rc;
  rc.init(g, c, r, rays_o, rays_d, jitter);
  const int s_end = min(nseg, (cseg + 1) * fseg);
  float z_next = 0.0f;
  int z_for = -1;  // sample index z_next belongs to
  for (int seg = cseg * fseg; seg < s_end; ++seg) {
    const int ks = seg * c.seg_len, ke = min(c.S, ks + c.seg_len) - 1;
    const int k_lo = max(rc.k_lo, ks), k_hi = min(rc.k_hi, ke);
    float csum[COUT];
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) csum[ch] = 0.0f;
    float asum = 0.0f, dsum = 0.0f, T = 1.0f;
    if (k_lo <= k_hi) {
      if (z_for != k_lo) z_next = rc.dg.z(k_lo);
      for (int k = k_lo; k <= k_hi; ++k) {
        const float z = z_next;
        const bool last = (k == c.S - 1);
        if (!last) { z_next = rc.dg.z(k + 1); z_for = k + 1; }
        float p[3];
        rc.point(z, p);
        Footprint fp;
        footprint(g, p, fp);
        if (!fp.inside) continue;
        Cell cell;
        make_cell_fast(g, fp, cell);
        float v, rad[COUT];
        gather<COUT, NCM, NCU>(g, packed, cell, rc.basis, v, rad);
        if constexpr (COUT == 3 && NCU > 1) {
          // slot of (tile, segment, sample k, lane) -- render_bwd_tile_kernel's src_base + k * 64
          if (sample_fwd && tile_lane >= 0)
            sample_fwd[tile_lane + ((long long)seg * c.seg_len + (k - ks)) * 64] = make_float4(rad[0], rad[1], rad[2], v);
        }
        const float sigma = post_activate(g.post_act, v);
        const float dl = last ? kInfinity : (z_next - z);
        const float delta = dl * rc.dnorm;
        const float e = fast_exp(-(sigma * delta));
        const float alpha = 1.0f - e;
        const float om = 1.0f - alpha;
        const float w = alpha * T;
        T = T * om;
#pragma unroll
        for (int ch = 0; ch < COUT; ++ch) csum[ch] = fmaf(sigmoidf(rad[ch]), w, csum[ch]);
        asum = asum + w;
        dsum = fmaf(z, w, dsum);
      }
    }
    const long long base = (long long)seg * NC;
    segbuf[(base + 0) * c.R + r] = T;
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) segbuf[(base + 1 + ch) * c.R + r] = csum[ch];
    segbuf[(base + 1 + COUT) * c.R + r] = asum;
    segbuf[(base + 2 + COUT) * c.R + r] = dsum;
  }
}

template <int COUT>
__global__ __launch_bounds__(256) void render_fwd_combine_kernel(DevCfg c, const float* __restrict__ segbuf,
                                                                 float* __restrict__ colour,
                                                                 float* __restrict__ depth,
                                                                 float* __restrict__ acc,
                                                                 float* __restrict__ disparity,
                                                                 float* __restrict__ ray_state) {
  constexpr int NC = COUT + 3;
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= c.R) return;
  const int nseg = num_segments(c.S, c.seg_len);
  float csum[COUT];
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) csum[ch] = 0.0f;
  float asum = 0.0f, dsum = 0.0f, T = 1.0f;
  for (int s = 0; s < nseg; ++s) {
    if (ray_state && s > 0) ray_state[ray_state_index(s, 0, NC, c.R, r)] = T;   // transmittance BEFORE segment s == boundary s
    const long long base = (long long)s * NC;
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) csum[ch] = fmaf(T, segbuf[(base + 1 + ch) * c.R + r], csum[ch]);
    asum = fmaf(T, segbuf[(base + 1 + COUT) * c.R + r], asum);
    dsum = fmaf(T, segbuf[(base + 2 + COUT) * c.R + r], dsum);
    T = T * segbuf[(base + 0) * c.R + r];
  }
  if (ray_state) {
    // State of boundary s for the backward: transmittance before it and the SUFFIX sums -- what the samples at and
    // behind the boundary contribute to (csum, asum, dsum) -- summed BACK TO FRONT (faint far segments first).  The
    // backward needs sum_{j > k} dL/dw_j w_j; taking it as (whole ray) - (prefix) loses the samples deep inside a dense
    // medium to cancellation (relative error 1e-7 / T_k); suffix sums keep it relative to the suffix itself, like the
    // reference's reverse cumsum (torch's cumprod backward; rendering/volumetric/accumulate.py:63-67).
    float sc[COUT], sa = 0.0f, sd = 0.0f;
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) sc[ch] = 0.0f;
    for (int s = nseg - 1; s > 0; --s) {
      const float Ts = ray_state[ray_state_index(s, 0, NC, c.R, r)];   // (written by this thread above)
      const long long base = (long long)s * NC;
#pragma unroll
      for (int ch = 0; ch < COUT; ++ch) {
        sc[ch] = fmaf(Ts, segbuf[(base + 1 + ch) * c.R + r], sc[ch]);
        ray_state[ray_state_index(s, 1 + ch, NC, c.R, r)] = sc[ch];
      }
      sa = fmaf(Ts, segbuf[(base + 1 + COUT) * c.R + r], sa);
      sd = fmaf(Ts, segbuf[(base + 2 + COUT) * c.R + r], sd);
      ray_state[ray_state_index(s, 1 + COUT, NC, c.R, r)] = sa;
      ray_state[ray_state_index(s, 2 + COUT, NC, c.R, r)] = sd;
    }
  }
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) {
    float col = csum[ch];
    if (c.white) {
      float bk = 1.0f - asum;
      if (c.attn) bk = bk * 0.0f;
      col = col + bk;
    }
    if (colour) colour[r * COUT + ch] = col;
  }
  if (depth) depth[r] = dsum;
  if (acc) acc[r] = asum;
  if (disparity) {
    const float q = dsum / asum;
    const float m = (q != q) ? q : (q > kZeroPlus ? q : kZeroPlus);
    disparity[r] = 1.0f / m;
  }
}

// r06: the same pass for at most MAXSEG depth segments with every partial in registers.  The kernel above re-reads, in its
// back-to-front loop, the transmittances it has just stored (a store -> load round trip through memory per boundary, which the
// compiler may not reorder: ~13 us for ANY number of rays, launch-latency scale work turned into seven dependent memory trips --
// kernel timeline, profiles/r06_bench_trace.txt).  Here all loads are issued up front (independent), the boundary transmittances
// stay in registers, and the stores are fire-and-forget: the same products and sums in the same order, bit-identical outputs.
template <int COUT, int MAXSEG>
__global__ __launch_bounds__(256) void render_fwd_combine_reg_kernel(DevCfg c, const float* __restrict__ segbuf,
                                                                     float* __restrict__ colour, float* __restrict__ depth,
                                                                     float* __restrict__ acc, float* __restrict__ disparity,
                                                                     float* __restrict__ ray_state) {
  constexpr int NC = COUT + 3;
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= c.R) return;
  const int nseg = num_segments(c.S, c.seg_len);     // <= MAXSEG (host)
  float part[MAXSEG][NC];
#pragma unroll
  for (int s = 0; s < MAXSEG; ++s)
#pragma unroll
    for (int q = 0; q < NC; ++q) part[s][q] = (s < nseg) ? segbuf[((long long)s * NC + q) * c.R + r] : 0.0f;
  float csum[COUT];
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) csum[ch] = 0.0f;
  float asum = 0.0f, dsum = 0.0f, T = 1.0f;
  float Tb[MAXSEG];                                   // transmittance BEFORE segment s == boundary s
#pragma unroll
  for (int s = 0; s < MAXSEG; ++s) {
    Tb[s] = T;
    if (s < nseg) {
#pragma unroll
      for (int ch = 0; ch < COUT; ++ch) csum[ch] = fmaf(T, part[s][1 + ch], csum[ch]);
      asum = fmaf(T, part[s][1 + COUT], asum);
      dsum = fmaf(T, part[s][2 + COUT], dsum);
      T = T * part[s][0];
    }
  }
  if (ray_state) {
    float sc[COUT], sa = 0.0f, sd = 0.0f;
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) sc[ch] = 0.0f;
#pragma unroll
    for (int s = MAXSEG - 1; s > 0; --s) {
      if (s < nseg) {
        const float Ts = Tb[s];
        ray_state[ray_state_index(s, 0, NC, c.R, r)] = Ts;
#pragma unroll
        for (int ch = 0; ch < COUT; ++ch) {
          sc[ch] = fmaf(Ts, part[s][1 + ch], sc[ch]);
          ray_state[ray_state_index(s, 1 + ch, NC, c.R, r)] = sc[ch];
        }
        sa = fmaf(Ts, part[s][1 + COUT], sa);
        sd = fmaf(Ts, part[s][2 + COUT], sd);
        ray_state[ray_state_index(s, 1 + COUT, NC, c.R, r)] = sa;
        ray_state[ray_state_index(s, 2 + COUT, NC, c.R, r)] = sd;
      }
    }
  }
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) {
    float col = csum[ch];
    if (c.white) {
      float bk = 1.0f - asum;
      if (c.attn) bk = bk * 0.0f;
      col = col + bk;
    }
    if (colour) colour[r * COUT + ch] = col;
  }
  if (depth) depth[r] = dsum;
  if (acc) acc[r] = asum;
  if (disparity) {
    const float q = dsum / asum;
    const float m = (q != q) ? q : (q > kZeroPlus ? q : kZeroPlus);
    disparity[r] = 1.0f / m;
  }
}

// ------------------------------------------------------------------------------------------------
// Backward: recompute the march, turn (d_colour, d_depth, d_acc) into per-sample gradients with the
// prefix/total form of the suffix sum, scatter-add to the packed gradient grid.
//   dL/dw_k    = sum_c g_c col_kc - [white & !attn] sum_c g_c + g_depth z_k + g_acc
//   dL/dsig_k  = delta_k e_k (T_k dL/dw_k - (sum_{j>k} dL/dw_j w_j) / om_k)
//   dL/drad_kc = w_k g_c col_kc (1 - col_kc)
// ------------------------------------------------------------------------------------------------
template <int COUT, int NCM, int NCU, bool WANT_D, bool WANT_F>
__global__ __launch_bounds__(256) void render_bwd_kernel(
    DevGrid g, DevCfg c, const float* __restrict__ packed, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, const float* __restrict__ jitter,
    const float* __restrict__ colour, const float* __restrict__ depth,
    const float* __restrict__ acc, const float* __restrict__ d_colour,
    const float* __restrict__ d_depth, const float* __restrict__ d_acc,
    float* __restrict__ gpacked) {
  constexpr int C = COUT * NCM + 1;
  long long r;
  if (!map_ray(c, r)) return;
  RayCtx<COUT, NCM, NCU> rc;
  rc.init(g, c, r, rays_o, rays_d, jitter);
  if (rc.k_lo > rc.k_hi) return;

  float gc[COUT], gsum = 0.0f;
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) { gc[ch] = d_colour[r * COUT + ch]; gsum += gc[ch]; }
  const float gdep = d_depth ? d_depth[r] : 0.0f;
  const float gacc = d_acc ? d_acc[r] : 0.0f;
  const bool white = c.white && !c.attn;
  const float asum = acc[r];
  // total = sum_j dL/dw_j w_j from the forward outputs
  float total = gdep * depth[r] + gacc * asum;
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) {
    const float csum = white ? colour[r * COUT + ch] - (1.0f - asum) : colour[r * COUT + ch];
    total += gc[ch] * csum;
  }
  if (white) total -= gsum * asum;

  float prefix = 0.0f, T = 1.0f;
  float z_next = rc.dg.z(rc.k_lo);
  for (int k = rc.k_lo; k <= rc.k_hi; ++k) {
    const float z = z_next;
    const bool last = (k == c.S - 1);
    if (!last) z_next = rc.dg.z(k + 1);
    float p[3];
    rc.point(z, p);
    Footprint fp;
    footprint(g, p, fp);
    if (!fp.inside) continue;
    Cell cell;
    make_cell_fast(g, fp, cell);
    float v, rad[COUT];
    gather<COUT, NCM, NCU>(g, packed, cell, rc.basis, v, rad);
    float sigma, dpost;
    post_activate_vg(g.post_act, v, sigma, dpost);
    const float dl = last ? kInfinity : (z_next - z);
    const float delta = dl * rc.dnorm;
    const float e = fast_exp(-(sigma * delta));
    const float alpha = 1.0f - e;
    const float om = 1.0f - alpha;
    const float w = alpha * T;

    float col[COUT], dldw = gdep * z + gacc;
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) { col[ch] = sigmoidf(rad[ch]); dldw += gc[ch] * col[ch]; }
    if (white) dldw -= gsum;
    prefix += dldw * w;
    const float suffix = last ? 0.0f : (total - prefix);
    const float tail = (om > 0.0f) ? suffix / om : 0.0f;
    const float dsig = (delta * e) * (T * dldw - tail);
    const float dv = dsig * dpost;
    float drad[COUT];
    bool any = (dv != 0.0f);
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) {
      drad[ch] = (w * gc[ch]) * (col[ch] * (1.0f - col[ch]));
      any = any || (drad[ch] != 0.0f);
    }
    T = T * om;

    if (any) {  // adding exact zeros is skipped
      const CellAddr ad = cell_addr(g, cell);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const float wg = (cell.w[0][kk & 1] * cell.w[1][(kk >> 1) & 1]) * cell.w[2][kk >> 2];
        if (wg != 0.0f) {
          float* __restrict__ dst =
              gpacked + (long long)(ad.base + (kk & 1) * ad.sx + ((kk >> 1) & 1) * ad.sy + (kk >> 2) * ad.sz) * C;
          if constexpr (WANT_F) {
#pragma unroll
            for (int ch = 0; ch < COUT; ++ch)
#pragma unroll
              for (int j = 0; j < NCU; ++j)
                atomicAdd(dst + ch * NCM + j, (drad[ch] * rc.basis[j]) * wg);
          }
          if constexpr (WANT_D) atomicAdd(dst + (C - 1), dv * wg);
        }
      }
    }
    if (c.term_eps > 0.0f && T < c.term_eps) break;
  }
}

// ------------------------------------------------------------------------------------------------
// Probe: per-sample index / mask / depth / sigma / radiance (test hook, shares all device code)
// ------------------------------------------------------------------------------------------------
template <int COUT, int NCM, int NCU>
__global__ __launch_bounds__(256) void sample_probe_kernel(
    DevGrid g, DevCfg c, const float* __restrict__ packed, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, const float* __restrict__ jitter, int32_t* __restrict__ idx,
    uint8_t* __restrict__ inside, float* __restrict__ zvals, float* __restrict__ sigma,
    float* __restrict__ radv) {
  long long r;
  if (!map_ray(c, r)) return;
  RayCtx<COUT, NCM, NCU> rc;
  rc.init(g, c, r, rays_o, rays_d, jitter);
  for (int k = 0; k < c.S; ++k) {
    const float z = rc.dg.z(k);
    float p[3];
    rc.point(z, p);
    Footprint fp;
    footprint(g, p, fp);
    const long long i = r * c.S + k;
    if (idx) { idx[3 * i + 0] = fp.i0[0]; idx[3 * i + 1] = fp.i0[1]; idx[3 * i + 2] = fp.i0[2]; }
    if (inside) inside[i] = fp.inside ? 1 : 0;
    if (zvals) zvals[i] = z;
    float v = 0.0f, rad[COUT];
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) rad[ch] = -kInfinity;  // process.py:80-82
    float sg = 0.0f;
    // the renderer only evaluates samples of [k_lo, k_hi]; the probe reports the same decision
    if (fp.inside && k >= rc.k_lo && k <= rc.k_hi) {
      Cell cell;
      make_cell(g, fp, cell);
      gather<COUT, NCM, NCU>(g, packed, cell, rc.basis, v, rad);
      sg = post_activate(g.post_act, v);
    }
    if (sigma) sigma[i] = sg;
    if (radv) {
#pragma unroll
      for (int ch = 0; ch < COUT; ++ch) radv[i * COUT + ch] = rad[ch];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Point query: VoxelGrid.forward / forward_attn (voxels.py:287-406), un-masked, + its backward.
// One thread per point; same footprint / cell code as the renderer (indices bit-exact by construction).
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void query_fwd_kernel(DevGrid g, const float* __restrict__ packed,
                                                        const float* __restrict__ points, long long N,
                                                        float* __restrict__ out) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float p[3] = {points[3 * n], points[3 * n + 1], points[3 * n + 2]};
  Footprint fp;
  footprint(g, p, fp);
  Cell cell;
  make_cell(g, fp, cell);
  const CellAddr ad = cell_addr(g, cell);
  float acc[C];
#pragma unroll
  for (int ch = 0; ch < C; ++ch) acc[ch] = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float w = (cell.w[0][k & 1] * cell.w[1][(k >> 1) & 1]) * cell.w[2][k >> 2];
    const float* __restrict__ src = packed + (long long)(ad.base + (k & 1) * ad.sx + ((k >> 1) & 1) * ad.sy + (k >> 2) * ad.sz) * C;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) acc[ch] = fmaf(src[ch], w, acc[ch]);
  }
  acc[C - 1] = post_activate(g.post_act, acc[C - 1]);
#pragma unroll
  for (int ch = 0; ch < C; ++ch) out[n * C + ch] = acc[ch];
}

template <int C>
__global__ __launch_bounds__(256) void query_bwd_kernel(DevGrid g, const float* __restrict__ packed,
                                                        const float* __restrict__ points, long long N,
                                                        const float* __restrict__ d_out,
                                                        float* __restrict__ gpacked, int want_d, int want_f) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float p[3] = {points[3 * n], points[3 * n + 1], points[3 * n + 2]};
  Footprint fp;
  footprint(g, p, fp);
  Cell cell;
  make_cell(g, fp, cell);
  const CellAddr ad = cell_addr(g, cell);
  float v = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float w = (cell.w[0][k & 1] * cell.w[1][(k >> 1) & 1]) * cell.w[2][k >> 2];
    v = fmaf(packed[(long long)(ad.base + (k & 1) * ad.sx + ((k >> 1) & 1) * ad.sy + (k >> 2) * ad.sz) * C + (C - 1)], w, v);
  }
  float value, dpost;
  post_activate_vg(g.post_act, v, value, dpost);
  const float dv = d_out[n * C + (C - 1)] * dpost;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float w = (cell.w[0][k & 1] * cell.w[1][(k >> 1) & 1]) * cell.w[2][k >> 2];
    if (w != 0.0f) {
      float* __restrict__ dst = gpacked + (long long)(ad.base + (k & 1) * ad.sx + ((k >> 1) & 1) * ad.sy + (k >> 2) * ad.sz) * C;
      if (want_f) {
#pragma unroll
        for (int ch = 0; ch < C - 1; ++ch) atomicAdd(dst + ch, d_out[n * C + ch] * w);
      }
      if (want_d) atomicAdd(dst + (C - 1), dv * w);
    }
  }
}

template <int C>
static void launch_query_t(const DevGrid& g, const float* packed, const float* points, long long N, float* out,
                           const float* d_out, float* gpacked, bool want_d, bool want_f, hipStream_t st) {
  const int nb = (int)((N + 255) / 256);
  if (out) query_fwd_kernel<C><<<nb, 256, 0, st>>>(g, packed, points, N, out);
  else query_bwd_kernel<C><<<nb, 256, 0, st>>>(g, packed, points, N, d_out, gpacked, want_d, want_f);
}

void launch_query(const DevGrid& g, int C, const float* packed, const float* points, long long N, float* out,
                  const float* d_out, float* gpacked, bool want_d, bool want_f, hipStream_t st) {
  switch (C) {
    case 2: launch_query_t<2>(g, packed, points, N, out, d_out, gpacked, want_d, want_f, st); break;
    case 4: launch_query_t<4>(g, packed, points, N, out, d_out, gpacked, want_d, want_f, st); break;
    case 13: launch_query_t<13>(g, packed, points, N, out, d_out, gpacked, want_d, want_f, st); break;
    case 28: launch_query_t<28>(g, packed, points, N, out, d_out, gpacked, want_d, want_f, st); break;
    case 49: launch_query_t<49>(g, packed, points, N, out, d_out, gpacked, want_d, want_f, st); break;
  }
}

// ------------------------------------------------------------------------------------------------
// Host-side launchers (called from voxe_api.hip)
// ------------------------------------------------------------------------------------------------
static inline int blocks_for(const DevCfg& c) {
  if (c.image_width > 0) {
    return blocks_for_tiles(c.map_mode, (c.image_width + 15) / 16, tile_rows_total(c, 16));
  }
  return blocks_for_tiles(c.map_mode, 1, (c.R + 255) / 256);
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int C>
static void launch_pack(const VoxeGridDesc* gd, float* packed, hipStream_t st) {
  const long long nvox = (long long)gd->X * gd->Y * gd->Z;
  if constexpr (C > 4) {
    constexpr int F = C - 1;
    long long done = 0;   // voxels converted by the 16-byte kernel (whole chunks; needs 16-byte aligned tensors)
    if (VOXE_WIDE16 && al16(gd->features) && al16(packed) && nvox >= kWideChunk) {
      const long long nchunks = nvox / kWideChunk;
      pack_grid_wide16_kernel<C><<<(int)(nchunks < 4096 ? nchunks : 4096), 256, 0, st>>>(
          gd->densities, gd->features, packed, nchunks, gd->density_scale, gd->density_pre_act);
      done = nchunks * kWideChunk;
    }
    if (done < nvox) {
      const long long n = (nvox - done) * C;
      const int nbw = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
      pack_grid_wide_kernel<C><<<nbw, 256, 0, st>>>(gd->densities + done, gd->features + done * F, packed + done * C,
                                                    nvox - done, gd->density_scale, gd->density_pre_act);
    }
    return;
  }
  const int nb = (int)((nvox + 255) / 256 < 4096 ? (nvox + 255) / 256 : 4096);
  pack_grid_kernel<C><<<nb, 256, 0, st>>>(gd->densities, gd->features, packed, nvox,
                                          gd->density_scale, gd->density_pre_act);
}

template <int C>
static void launch_unpack(const VoxeGridDesc* gd, const float* gpacked, float* d_dens, float* d_feat,
                          int accumulate, int bricked, hipStream_t st) {
  const long long nvox = (long long)gd->X * gd->Y * gd->Z;
  if constexpr (C > 4) {
    constexpr int F = C - 1;
    long long done = 0;
    if (VOXE_WIDE16 && !bricked && al16(gpacked) && (!d_feat || al16(d_feat)) && nvox >= kWideChunk) {
      const long long nchunks = nvox / kWideChunk;
      unpack_grad_wide16_kernel<C><<<(int)(nchunks < 4096 ? nchunks : 4096), 256, 0, st>>>(
          gpacked, gd->densities, d_dens, d_feat, nchunks, gd->density_scale, gd->density_pre_act, accumulate);
      done = nchunks * kWideChunk;
    }
    if (done < nvox) {   // (the tail: not bricked here -- a bricked buffer takes the element kernel whole)
      const long long n = (nvox - done) * C;
      const int nbw = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
      unpack_grad_wide_kernel<C><<<nbw, 256, 0, st>>>(gpacked + done * C, gd->densities + done, d_dens ? d_dens + done : nullptr,
                                                      d_feat ? d_feat + done * F : nullptr, nvox - done, gd->density_scale,
                                                      gd->density_pre_act, accumulate, done ? 0 : bricked, gd->Y, gd->Z);
    }
    return;
  }
  const int nb = (int)((nvox + 255) / 256 < 4096 ? (nvox + 255) / 256 : 4096);
  unpack_grad_kernel<C><<<nb, 256, 0, st>>>(gpacked, gd->densities, d_dens, d_feat, nvox,
                                            gd->density_scale, gd->density_pre_act, accumulate, bricked, gd->Y, gd->Z);
}

template <int COUT, int NCM, int NCU>
static void launch_fwd_t(const DevGrid& g, const HostCfg& c, const FwdArgs& a, hipStream_t st) {
  const int nseg = num_segments(c.S, c.seg_len);
  // The march of every ray block is split into depth segments handled by different blocks (a 100x100 image alone
  // is 40 blocks; at 400x400 the finer split hides the gather latency better): with the segment-major block order
  // one 32-sample segment per task is fastest at every image size (400x400, mean of three cameras: fseg 1 / 2 / 4 / 8
  // = 0.273 / 0.286 / 0.291 / 0.326 ms).
  // (term_eps never changes a forward -- since r03 it only truncates the BACKWARD: samples behind a transmittance below
  //  it receive no gradient -- so the segmented forward runs whatever its value)
  if (a.segbuf && nseg > 1) {
    const int fseg = c.disp.fwd_segments_per_thread > 0 ? c.disp.fwd_segments_per_thread : 1;
    const int ncoarse = (nseg + fseg - 1) / fseg;
    const int nrb64 = c.image_width > 0
                          ? blocks_for_tiles(c.map_mode, (c.image_width + 7) / 8, tile_rows_total(c, 8))
                          : blocks_for_tiles(c.map_mode, 1, (c.R + 63) / 64);
    if (NCU == 1 && fseg == 1 && fwd_tile4_supported(g, c, a, COUT, NCM))
      launch_fwd_tile4(g, c, a, st);     // r05: the lean tile-ordered forward (voxe_render_tile4.hip)
    else if (NCU == 1 && fseg == 1 && fwd_tile_supported(g, c, COUT, NCM))
      launch_fwd_tile(g, c, a, st);      // texels of a tile staged in LDS (voxe_render_tile.hip)
    else if (NCU > 1 && fseg == 1 && fwd_tilew_supported(g, c, a, COUT, NCM, NCU))
      launch_fwd_tilew(g, c, NCM, a, st);   // r06: whole view-dependent texels of a tile staged in LDS (voxe_render_tilew.hip)
    else
      render_fwd_seg_kernel<COUT, NCM, NCU><<<nrb64 * ncoarse, 64, 0, st>>>(
          g, c, fseg, a.packed, a.rays_o, a.rays_d, a.jitter, a.segbuf, reinterpret_cast<float4*>(a.sample_fwd));
#ifndef VOXE_COMBINE_REG
#define VOXE_COMBINE_REG 1
#endif
    const int cb = (int)((c.R + 255) / 256);
    if (VOXE_COMBINE_REG && nseg <= 8)
      render_fwd_combine_reg_kernel<COUT, 8><<<cb, 256, 0, st>>>(c, a.segbuf, a.colour, a.depth, a.acc, a.disparity, a.ray_state);
    else if (VOXE_COMBINE_REG && nseg <= 16)
      render_fwd_combine_reg_kernel<COUT, 16><<<cb, 256, 0, st>>>(c, a.segbuf, a.colour, a.depth, a.acc, a.disparity, a.ray_state);
    else
      render_fwd_combine_kernel<COUT><<<cb, 256, 0, st>>>(c, a.segbuf, a.colour, a.depth, a.acc, a.disparity, a.ray_state);
    return;
  }
  render_fwd_kernel<COUT, NCM, NCU><<<blocks_for(c), 256, 0, st>>>(
      g, c, a.packed, a.rays_o, a.rays_d, a.jitter, a.colour, a.depth, a.acc, a.disparity, a.ray_state);
}

template <int COUT, int NCM, int NCU>
static void launch_bwd_t(const DevGrid& g, const DevCfg& c, const BwdArgs& a, hipStream_t st) {
  const int nb = blocks_for(c);
#define VOXE_BWD(WD, WF)                                                                         \
  render_bwd_kernel<COUT, NCM, NCU, WD, WF><<<nb, 256, 0, st>>>(                                 \
      g, c, a.packed, a.rays_o, a.rays_d, a.jitter, a.colour, a.depth, a.acc, a.d_colour,        \
      a.d_depth, a.d_acc, a.gpacked)
  if (a.want_d && a.want_f) VOXE_BWD(true, true);
  else if (a.want_d) VOXE_BWD(true, false);
  else VOXE_BWD(false, true);
#undef VOXE_BWD
}

template <int COUT, int NCM, int NCU>
static void launch_probe_t(const DevGrid& g, const DevCfg& c, const ProbeArgs& a, hipStream_t st) {
  sample_probe_kernel<COUT, NCM, NCU><<<blocks_for(c), 256, 0, st>>>(
      g, c, a.packed, a.rays_o, a.rays_d, a.jitter, a.idx, a.inside, a.zvals, a.sigma, a.rad);
}

// variant dispatch on (feature kind, SH degree, diffuse)
#define VOXE_DISPATCH(FN, attn, deg, diffuse, ...)                        \
  do {                                                                    \
    if (attn) { FN<1, 1, 1>(__VA_ARGS__); }                               \
    else if ((deg) == 0) { FN<3, 1, 1>(__VA_ARGS__); }                    \
    else if ((deg) == 1) { if (diffuse) FN<3, 4, 1>(__VA_ARGS__); else FN<3, 4, 4>(__VA_ARGS__); }    \
    else if ((deg) == 2) { if (diffuse) FN<3, 9, 1>(__VA_ARGS__); else FN<3, 9, 9>(__VA_ARGS__); }    \
    else { if (diffuse) FN<3, 16, 1>(__VA_ARGS__); else FN<3, 16, 16>(__VA_ARGS__); }                 \
  } while (0)

void launch_pack_any(const VoxeGridDesc* gd, float* packed, hipStream_t st) {
  switch (gd->F + 1) {
    case 2: launch_pack<2>(gd, packed, st); break;
    case 4: launch_pack<4>(gd, packed, st); break;
    case 13: launch_pack<13>(gd, packed, st); break;
    case 28: launch_pack<28>(gd, packed, st); break;
    case 49: launch_pack<49>(gd, packed, st); break;
  }
}
void launch_unpack_any(const VoxeGridDesc* gd, const float* gpacked, float* d_dens, float* d_feat,
                       int accumulate, int bricked, hipStream_t st) {
  switch (gd->F + 1) {
    case 2: launch_unpack<2>(gd, gpacked, d_dens, d_feat, accumulate, bricked, st); break;
    case 4: launch_unpack<4>(gd, gpacked, d_dens, d_feat, accumulate, bricked, st); break;
    case 13: launch_unpack<13>(gd, gpacked, d_dens, d_feat, accumulate, bricked, st); break;
    case 28: launch_unpack<28>(gd, gpacked, d_dens, d_feat, accumulate, bricked, st); break;
    case 49: launch_unpack<49>(gd, gpacked, d_dens, d_feat, accumulate, bricked, st); break;
  }
}
void launch_fwd(const DevGrid& g, const HostCfg& c, int deg, int diffuse, const FwdArgs& a,
                hipStream_t st) {
  VOXE_DISPATCH(launch_fwd_t, c.attn, deg, diffuse, g, c, a, st);
}
void launch_bwd(const DevGrid& g, const DevCfg& c, int deg, int diffuse, const BwdArgs& a,
                hipStream_t st) {
  VOXE_DISPATCH(launch_bwd_t, c.attn, deg, diffuse, g, c, a, st);
}
void launch_probe(const DevGrid& g, const DevCfg& c, int deg, int diffuse, const ProbeArgs& a,
                  hipStream_t st) {
  VOXE_DISPATCH(launch_probe_t, c.attn, deg, diffuse, g, c, a, st);
}

}  // namespace voxe
