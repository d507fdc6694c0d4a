// voxe_device.hpp -- device-side building blocks of the gfx950 voxel-grid renderer.
//
// Everything here restates, per sample, the arithmetic of the reference hot path
// (thre3d_atom/rendering/volumetric/{sample,process,accumulate}.py, thre3d_atom/thre3d_reprs/voxels.py)
// in float32 with the reference's operation order.  This translation unit is compiled with
// -ffp-contract=off: an FMA appears only where fmaf() is written, so the voxel-index math
// (p = o + d*z -> n = p*scale + bias -> u = ((n+1)*N-1)/2 -> floor) is bit-identical to the oracle.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/voxe.h"

namespace voxe {

constexpr float kZeroPlus = 1e-10f;   // thre3d_atom/utils/constants.py:8
constexpr float kInfinity = 1e10f;    // thre3d_atom/utils/constants.py:9
constexpr float kC0 = 0.28209479177387814f;  // spherical_harmonics.py:33

// Launch-constant description of the grid and the render config, passed by value (kernarg / SGPRs).
struct DevGrid {
  int X, Y, Z;
  float lo[3], hi[3], scale[3], bias[3];
  float density_scale;
  int pre_act, post_act;
};

struct DevCfg {
  int S;
  float near, far;
  int perturb, lindisp, aabb_clip, white, attn;
  float term_eps;
  uint32_t key0, key1, ctr3;  // Philox key / 4th counter word
  int image_width;            // 0 = linear ray order
  int map_mode;               // block -> tile mapping: 0 XCD bands, 1 linear, 2 tile rows interleaved over XCDs
  long long R;
};

// ------------------------------------------------------------------------------------------------
// Philox4x32-10: in-kernel jitter stream (same definition as oracle/voxe_cpu.c)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]);
    const uint32_t lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]);
    const uint32_t lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0;
    const uint32_t n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// ------------------------------------------------------------------------------------------------
// Activations (voxels.py:303-320)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pre_activate(int act, float raw, float scale) {
  const float v = raw * scale;
  return act == VOXE_ACT_ABS ? fabsf(v) : v;
}
__device__ __forceinline__ float pre_activate_grad(int act, float raw, float scale) {
  const float v = raw * scale;
  float s = 1.0f;
  if (act == VOXE_ACT_ABS) s = (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f);
  return s * scale;
}
// torch.nn.Softplus(beta=1, threshold=20) | ReLU | Identity
__device__ __forceinline__ float post_activate(int act, float v) {
  if (act == VOXE_ACT_SOFTPLUS) return v > 20.0f ? v : log1pf(expf(v));
  if (act == VOXE_ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}
__device__ __forceinline__ float post_activate_grad(int act, float v) {
  if (act == VOXE_ACT_SOFTPLUS) {
    if (v > 20.0f) return 1.0f;
    const float z = expf(v);
    return z / (z + 1.0f);
  }
  if (act == VOXE_ACT_RELU) return v > 0.f ? 1.f : 0.f;
  return 1.0f;
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// ------------------------------------------------------------------------------------------------
// Sample depths: sample_uniform_points_on_rays (sample.py:15-68) as a rolling generator.
// z(k) for k = 0..S-1 in order; keeps the window needed by the stratified jitter (mids) and by
// delta_k = z_{k+1} - z_k (accumulate.py:49).
// ------------------------------------------------------------------------------------------------
struct DepthGen {
  float near, far, step;
  int S, half;
  bool lindisp, perturb;
  const float* jit;  // this ray's row of the jitter tensor or nullptr
  uint32_t k0, k1, c0, c1, c3;
  uint32_t r0, r1, r2, r3;  // cached Philox block (scalars: a register array indexed by k&3 would spill)
  int rnd_block;

  // torch.linspace(0,1,S)[k]  (sample.py:44)
  __device__ __forceinline__ float tval(int k) const {
    if (S == 1) return 0.0f;
    if (k < half) return step * (float)k;
    return fmaf(-step, (float)(S - 1 - k), 1.0f);
  }
  // un-jittered depth (sample.py:48-54)
  __device__ __forceinline__ float zlin(int k) const {
    const float t = tval(k);
    if (lindisp) {
      const float a = (1.0f / (near + kZeroPlus)) * (1.0f - t);
      const float b = (1.0f / far) * t;
      return 1.0f / (a + b);
    }
    const float a = near * (1.0f - t);
    const float b = far * t;
    return a + b;
  }
  __device__ __forceinline__ float uniform(int k) {
    if (jit) return jit[k];
    const int blk = k >> 2;
    if (blk != rnd_block) {
      uint32_t c[4] = {c0, c1, (uint32_t)blk, c3};
      philox4x32_10(c, k0, k1);
      r0 = c[0]; r1 = c[1]; r2 = c[2]; r3 = c[3];
      rnd_block = blk;
    }
    const uint32_t lo = (k & 1) ? r1 : r0, hi = (k & 1) ? r3 : r2;
    const uint32_t x = (k & 2) ? hi : lo;
    return (float)(x >> 8) * (1.0f / 16777216.0f);
  }
  // final depth of sample k (sample.py:57-64)
  __device__ __forceinline__ float z(int k) {
    const float zk = zlin(k);
    if (!perturb) return zk;
    const float lower = (k == 0) ? zk : 0.5f * (zk + zlin(k - 1));
    const float upper = (k == S - 1) ? zk : 0.5f * (zlin(k + 1) + zk);
    const float u = uniform(k);
    const float span = upper - lower;
    return lower + span * u;
  }
};

// _ray_aabb_intersection (sample.py:71-184): per-ray (near, far)
__device__ __forceinline__ void ray_aabb_bounds(const DevGrid& g, const float (&o)[3],
                                                const float (&d)[3], float& near, float& far) {
  float fmin_ = 0.f, fmax_ = 0.f;
  bool hit = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float den = d[a] + kZeroPlus;
    float t0 = (g.lo[a] - o[a]) / den;
    float t1 = (g.hi[a] - o[a]) / den;
    if (t0 > t1) { const float t = t0; t0 = t1; t1 = t; }
    if (a == 0) {
      fmin_ = t0; fmax_ = t1;
    } else {
      if (fmin_ > t1 || t0 > fmax_) hit = false;
      if (t0 > fmin_) fmin_ = t0;
      if (t1 < fmax_) fmax_ = t1;
    }
  }
  if (!hit) { fmin_ = near; fmax_ = far; }
  if (fmin_ < 0.0f) fmin_ = 0.0f;
  if (fmax_ < 0.0f) fmax_ = 0.0f;
  near = fmin_; far = fmax_;
}

// Conservative range [k_lo, k_hi] of samples that can lie strictly inside the AABB; samples
// outside the AABB contribute exactly nothing (sigma = 0 => alpha = 0, process.py:83-84), so
// skipping them changes no output bit.  The exact strict test is still applied per sample.
__device__ __forceinline__ void inside_range(const DevGrid& g, const DevCfg& c, const DepthGen& dg,
                                             const float (&o)[3], const float (&d)[3], int& k_lo,
                                             int& k_hi) {
  k_lo = 0; k_hi = c.S - 1;
  float t_in = -3.0e38f, t_out = 3.0e38f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (d[a] != 0.0f) {
      const float inv = 1.0f / d[a];
      float t0 = (g.lo[a] - o[a]) * inv, t1 = (g.hi[a] - o[a]) * inv;
      if (t0 > t1) { const float t = t0; t0 = t1; t1 = t; }
      t_in = fmaxf(t_in, t0);
      t_out = fminf(t_out, t1);
    } else if (!(o[a] > g.lo[a] && o[a] < g.hi[a])) {
      k_lo = 1; k_hi = 0;  // parallel to the slab and outside it: no sample can be inside
      return;
    }
  }
  if (!(t_in <= t_out)) {  // miss (or NaN): nothing inside / be safe on NaN
    if (t_in > t_out) { k_lo = 1; k_hi = 0; }
    return;
  }
  const float zn = dg.near, zf = dg.far;
  if (!(zf > zn) || c.S < 2) return;  // degenerate bounds: keep the full range
  float f_in, f_out;  // fractional sample index of entry / exit
  if (dg.lindisp) {
    // 1/z is affine in t: t = (1/zn - 1/z) / (1/zn - 1/zf)
    const float in = 1.0f / (zn + kZeroPlus), ifar = 1.0f / zf;
    const float den = in - ifar;
    if (!(den > 0.f) || !(t_in > 0.f)) { f_in = -1.0f; } else { f_in = (in - 1.0f / t_in) / den; }
    if (!(den > 0.f) || !(t_out > 0.f)) { k_lo = 1; k_hi = 0; return; }
    f_out = (in - 1.0f / t_out) / den;
  } else {
    const float inv = 1.0f / (zf - zn);
    f_in = (t_in - zn) * inv;
    f_out = (t_out - zn) * inv;
  }
  const float sm1 = (float)(c.S - 1);
  // +-2 samples of slack cover the jitter (half a step) and all rounding of this estimate
  float a = floorf(f_in * sm1) - 2.0f, b = ceilf(f_out * sm1) + 2.0f;
  if (!(a == a) || !(b == b)) return;
  a = fmaxf(a, 0.0f); b = fminf(b, sm1);
  if (a > b) { k_lo = 1; k_hi = 0; return; }
  k_lo = (int)a; k_hi = (int)b;
}

// ------------------------------------------------------------------------------------------------
// Trilinear footprint of one sample: VoxelGrid._normalize_points (voxels.py:225-234) + ATen
// grid_sampler_3d(align_corners=False, zeros padding) index/weight math + test_inside_volume
// (voxels.py:263-285).
// ------------------------------------------------------------------------------------------------
struct Footprint {
  int i0[3];        // floor index of the low corner (may be -1 / N-1 at the faces)
  float w[3][2];    // axis weights, [a][0] for i0, [a][1] for i0+1
  bool inside;
};

__device__ __forceinline__ void footprint(const DevGrid& g, const float (&p)[3], Footprint& f) {
  const int N[3] = {g.X, g.Y, g.Z};
  f.inside = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float n = p[a] * g.scale[a];
    n = n + g.bias[a];
    float u = n + 1.0f;
    u = u * (float)N[a];
    u = u - 1.0f;
    u = u * 0.5f;  // == u / 2 exactly
    const float fl = floorf(u);
    f.i0[a] = (int)fl;
    f.w[a][0] = (fl + 1.0f) - u;
    f.w[a][1] = u - fl;
    f.inside = f.inside && (p[a] > g.lo[a]) && (p[a] < g.hi[a]);
  }
}

// Clamped corner addressing: corner (cx,cy,cz) -> voxel index + weight (0 for out-of-range corners,
// which ATen skips).  Order of the 8 corners = ATen's tnw,tne,tsw,tse,bnw,bne,bsw,bse.
struct Corners {
  int vox[8];
  float wgt[8];
};

__device__ __forceinline__ void corners(const DevGrid& g, const Footprint& f, Corners& c) {
  int ix[2], iy[2], iz[2];
  float wx[2], wy[2], wz[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int x = f.i0[0] + s, y = f.i0[1] + s, z = f.i0[2] + s;
    const bool vx = (x >= 0) && (x < g.X), vy = (y >= 0) && (y < g.Y), vz = (z >= 0) && (z < g.Z);
    ix[s] = min(max(x, 0), g.X - 1);
    iy[s] = min(max(y, 0), g.Y - 1);
    iz[s] = min(max(z, 0), g.Z - 1);
    wx[s] = vx ? f.w[0][s] : 0.0f;
    wy[s] = vy ? f.w[1][s] : 0.0f;
    wz[s] = vz ? f.w[2][s] : 0.0f;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int cx = k & 1, cy = (k >> 1) & 1, cz = k >> 2;
    c.vox[k] = (ix[cx] * g.Y + iy[cy]) * g.Z + iz[cz];
    c.wgt[k] = (wx[cx] * wy[cy]) * wz[cz];
  }
}

// SH basis for a unit view direction (spherical_harmonics.py:86-116). NC = (deg+1)^2 used.
template <int NC>
__device__ __forceinline__ void sh_basis(const float (&v)[3], float (&b)[NC]) {
  b[0] = kC0;
  if constexpr (NC > 1) {
    const float x = v[0], y = v[1], z = v[2];
    constexpr float C1 = 0.4886025119029199f;
    b[1] = -(C1 * y);
    b[2] = C1 * z;
    b[3] = -(C1 * x);
    if constexpr (NC > 4) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      b[4] = 1.0925484305920792f * xy;
      b[5] = -1.0925484305920792f * yz;
      b[6] = 0.31539156525252005f * (2.0f * zz - xx - yy);
      b[7] = -1.0925484305920792f * xz;
      b[8] = 0.5462742152960396f * (xx - yy);
      if constexpr (NC > 9) {
        b[9] = (-0.5900435899266435f * y) * (3.f * xx - yy);
        b[10] = (2.890611442640554f * xy) * z;
        b[11] = (-0.4570457994644658f * y) * (4.f * zz - xx - yy);
        b[12] = (0.3731763325901154f * z) * (2.f * zz - 3.f * xx - 3.f * yy);
        b[13] = (-0.4570457994644658f * x) * (4.f * zz - xx - yy);
        b[14] = (1.445305721320277f * z) * (xx - yy);
        b[15] = (-0.5900435899266435f * x) * (xx - 3.f * yy);
      }
    }
  }
}

// XCC (XCD) id of the executing wave: HW_REG_XCC_ID (id 20), bits [3:0].
__device__ __forceinline__ int xcc_id() {
  return (int)(__builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)));
}

}  // namespace voxe
