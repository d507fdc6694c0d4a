// voxe_device.hpp -- device-side building blocks of the gfx950 voxel-grid renderer.
//
// Everything here restates, per sample, the arithmetic of the reference hot path
// (thre3d_atom/rendering/volumetric/{sample,process,accumulate}.py, thre3d_atom/thre3d_reprs/voxels.py)
// in float32 with the reference's operation order.  This translation unit is compiled with
// -ffp-contract=off: an FMA appears only where fmaf() is written, so the voxel-index math
// (p = o + d*z -> n = p*scale + bias -> u = ((n+1)*N-1)/2 -> floor) is bit-identical to the oracle.
#pragma once

#include <hip/hip_runtime.h>
#include <limits.h>
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
  uint32_t key0, key1;        // jitter stream keys (see jitter_base())
  int image_width;            // 0 = linear ray order
  int image_height;           // rows of ONE image (image_width > 0); the launch holds R / (H * W) images back to back
  int map_mode;               // block -> tile mapping: 0 XCD bands, 1 linear, 2 tile rows interleaved over XCDs
  long long R;
  int linear_grad;            // the scatter backward writes the linear gradient layout (VoxeRenderCfg::linear_grad)
  int seg_len;                // samples per depth segment of the segmented kernels (seg_len_for(R))
  // paired launch (voxe_recon_step on SH-0 grids, space-binned route only): rays [pair_R, 2 pair_R) of the launch are rays
  // [0, pair_R) of the ray arrays again, drawn with the second jitter stream (key0b, key1b); 0 = off
  long long pair_R;
  uint32_t key0b, key1b;
};

// ------------------------------------------------------------------------------------------------
// In-kernel jitter stream: a counter-based 32-bit hash (same definition as oracle/voxe_cpu.c).
//   key0 = seed_lo ^ (offset_lo * 0x9E3779B1), key1 = seed_hi ^ offset_hi ^ 0x7F4A7C15
//   base(ray)  = mix32(mix32(ray_lo ^ key0) + (ray_hi ^ key1))
//   u(ray, k)  = (mix32(base + k * 0x9E3779B9) >> 8) * 2^-24          in [0, 1)
// mix32 is the "lowbias32" integer finaliser (2 multiplies, 3 xor-shifts): ~10 VALU per sample instead of
// ~25 for Philox4x32-10, which matters because the render kernels are instruction-issue bound.
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// Slot of voxel (x, y, z) in a 2x2x2-BRICKED buffer: the 8 voxels of a brick are contiguous (8 x 16 B = one 128 B line
// with 4 channels), so the 2x2x2 footprint of a sample touches (1.5)^3 = 3.4 lines on average instead of 4.5 in the
// linear [X,Y,Z] order.  Used for the gradient buffer of the scatter backward (requests, not bytes, bound it).
__host__ __device__ inline long long brick_slot(int x, int y, int z, int Y, int Z) {
  const long long by = (Y + 1) >> 1, bz = (Z + 1) >> 1;
  return ((((long long)(x >> 1) * by + (y >> 1)) * bz + (z >> 1)) << 3) | (long long)(((x & 1) << 2) | ((y & 1) << 1) | (z & 1));
}

// Keyed pseudo-random permutation of [0, n): 4-round Feistel network on b = 2 * half bits (2^b >= n) with mix32 round
// functions, restricted to [0, n) by cycle walking (a bijection of [0, 2^b) stays a bijection of [0, n) when out-of-range
// values are fed through again).  Same definition in oracle/voxe_cpu.c.
__host__ __device__ inline uint32_t feistel_permute(uint32_t x, uint32_t n, int half, uint32_t key0, uint32_t key1) {
  const uint32_t mask = (1u << half) - 1u;
  do {
    uint32_t l = x >> half, r = x & mask;
    for (int round = 0; round < 4; ++round) {
      const uint32_t f = mix32(r ^ (round & 1 ? key1 : key0) ^ ((uint32_t)round * 0x9E3779B9u)) & mask;
      const uint32_t t = l ^ f;
      l = r;
      r = t;
    }
    x = (l << half) | r;
  } while (x >= n);
  return x;
}

__host__ __device__ inline uint32_t jitter_base(uint32_t key0, uint32_t key1, long long ray) {
  const uint32_t lo = (uint32_t)ray, hi = (uint32_t)((unsigned long long)ray >> 32);
  return mix32(mix32(lo ^ key0) + (hi ^ key1));
}
__host__ __device__ inline float jitter_uniform(uint32_t base, int k) {
  return (float)(mix32(base + (uint32_t)k * 0x9E3779B9u) >> 8) * (1.0f / 16777216.0f);
}

// ------------------------------------------------------------------------------------------------
// Fast transcendental helpers (v_exp_f32 / v_log_f32 / v_rcp_f32 are 1-ulp hardware ops).  Only used
// for VALUES (alpha, softplus, sigmoid); the voxel-index arithmetic never goes through them.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// exp(x) = 2^(x log2 e): one multiply + v_exp_f32.  The rounding of the product costs a relative error of
// |x| * 6e-8 in the result, i.e. <= 1e-6 for |x| <= 16 (beyond that exp(-|x|) < 1e-7 and only its order of
// magnitude matters to alpha / sigmoid / softplus).  A compensated two-float product was measured to buy no
// accuracy that survives the float32 compositing and costs 4 more instructions per call on an issue-bound kernel.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
// log1p(t) for t in [0, 1]
__device__ __forceinline__ float fast_log1p01(float t) {
  constexpr float kLn2 = 0.693147182464599609375f;
  const float big = __builtin_amdgcn_logf(1.0f + t) * kLn2;
  const float small = fmaf(t, -0.5f * t, t);  // t - t^2/2, exact to float for t < 2^-12
  return t < 0.000244140625f ? small : big;
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
// torch.nn.Softplus(beta=1, threshold=20) | ReLU | Identity, value and derivative in one go.
// softplus(v) = max(v, 0) + log1p(exp(-|v|)) (== v above the threshold 20 in float32, like torch);
// softplus'(v) = sigmoid(v) (torch: z / (z + 1), z = exp(v)).
__device__ __forceinline__ void post_activate_vg(int act, float v, float& value, float& grad) {
  if (act == VOXE_ACT_SOFTPLUS) {
    const float t = fast_exp(-fabsf(v));
    const float rc = fast_rcp(1.0f + t);
    value = fmaxf(v, 0.0f) + fast_log1p01(t);
    grad = v >= 0.0f ? rc : t * rc;
  } else if (act == VOXE_ACT_RELU) {
    value = v > 0.f ? v : 0.f;
    grad = v > 0.f ? 1.f : 0.f;
  } else {
    value = v;
    grad = 1.0f;
  }
}
__device__ __forceinline__ float post_activate(int act, float v) {
  float value, grad;
  post_activate_vg(act, v, value, grad);
  return value;
}
// sigmoid(x) = 1 / (1 + exp(-x)) evaluated as written: exp2 saturates to +inf for x < -88 (-> rcp(inf) = 0, the true value is
// below 1e-38) and to 0 for x > 88 (-> 1); relative error <= 2 ulp everywhere else.  r04: the sign-split form
// (exp(-|x|), select, extra multiply) cost 3 more VALU instructions per call, three calls per sample in kernels that are VALU
// bound since the banked LDS window.
__device__ __forceinline__ float sigmoidf(float x) { return fast_rcp(1.0f + fast_exp(-x)); }

// ------------------------------------------------------------------------------------------------
// Sample depths: sample_uniform_points_on_rays (sample.py:15-68) as a rolling generator.
// z(k) for k = 0..S-1 in order; keeps the window needed by the stratified jitter (mids) and by
// delta_k = z_{k+1} - z_k (accumulate.py:49).
// ------------------------------------------------------------------------------------------------
struct DepthGen {
  float near, far, step;
  int S, half;
  bool lindisp, perturb;
  const float* jit;   // this ray's row of the jitter tensor or nullptr
  uint32_t base;      // jitter_base() of this ray
  int kc;             // centre of the cached window (zm, z0, zp) = zlin(kc-1 .. kc+1); INT_MIN = empty
  float zm, z0, zp;

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
  __device__ __forceinline__ float uniform(int k) const { return jit ? jit[k] : jitter_uniform(base, k); }
  // final depth of sample k (sample.py:57-64).  Sequential calls (k, k+1, ...) cost one zlin() each.
  __device__ __forceinline__ float z(int k) {
    if (!perturb) return zlin(k);
    if (k == kc + 1) { zm = z0; z0 = zp; }
    else { zm = (k > 0) ? zlin(k - 1) : 0.0f; z0 = zlin(k); }
    zp = (k < S - 1) ? zlin(k + 1) : 0.0f;
    kc = k;
    const float lower = (k == 0) ? z0 : 0.5f * (z0 + zm);
    const float upper = (k == S - 1) ? z0 : 0.5f * (zp + z0);
    const float span = upper - lower;
    return lower + span * uniform(k);
  }
};

// Stratum table of one depth segment (r04).  Without AABB clipping every ray of a launch has the same (near, far), so the
// stratum (lower_k, span_k) of sample k -- DepthGen::z()'s three zlin() values, two mid-points and the difference, ~17 VALU
// instructions per sample in kernels that are VALU-issue bound -- is the same for all rays: the lanes of a block tabulate the
// strata of their segment ONCE, with DepthGen's own expressions (identical floats), and a sample's depth becomes one LDS
// read + `lower + span * u`.  Entry j of the table belongs to sample ks + j.
__device__ __forceinline__ float2 depth_stratum(const DepthGen& dg, int k) {
  const float z0 = dg.zlin(k);
  if (!dg.perturb) return make_float2(z0, 0.0f);
  const float zm = (k > 0) ? dg.zlin(k - 1) : 0.0f;
  const float zp = (k < dg.S - 1) ? dg.zlin(k + 1) : 0.0f;
  const float lower = (k == 0) ? z0 : 0.5f * (z0 + zm);
  const float upper = (k == dg.S - 1) ? z0 : 0.5f * (zp + z0);
  return make_float2(lower, upper - lower);
}
template <bool TAB>
struct SegDepth {
  const float2* tab;   // LDS
  int ks;
  __device__ __forceinline__ float z(DepthGen& dg, int k) const {
    if constexpr (TAB) {
      const float2 t = tab[k - ks];
      if (!dg.perturb) return t.x;
      return t.x + t.y * dg.uniform(k);
    } else {
      return dg.z(k);
    }
  }
};

// The same table for kernels whose blocks march many rays (space-binned route): strata of samples k0 .. k0 + n - 1 built by all
// threads of the block from the launch's (near, far); `on` is block-uniform (false: per-ray AABB bounds, or S beyond the table).
struct BlockStrata {
  const float2* tab;
  int k0;
  bool on;
  __device__ __forceinline__ float z(DepthGen& dg, int k) const {
    if (on) {
      const float2 t = tab[k - k0];
      if (!dg.perturb) return t.x;
      return t.x + t.y * dg.uniform(k);
    }
    return dg.z(k);
  }
};
__device__ __forceinline__ BlockStrata build_block_strata(float2* tab, int capacity, const DevCfg& c, int k0, int n, int tid,
                                                          int nthreads) {
  BlockStrata b{tab, k0, !c.aabb_clip && n <= capacity};
  if (b.on) {
    DepthGen dg;   // (what RayCtx::init / SegRay::init set without AABB clipping)
    dg.near = c.near; dg.far = c.far;
    dg.lindisp = c.lindisp != 0;
    dg.S = c.S; dg.half = c.S >> 1;
    dg.step = 1.0f / (float)(c.S - 1);
    dg.perturb = c.perturb != 0;
    dg.jit = nullptr; dg.base = 0u; dg.kc = INT_MIN; dg.zm = dg.z0 = dg.zp = 0.0f;
    for (int i = tid; i < n; i += nthreads)
      if (k0 + i < c.S) tab[i] = depth_stratum(dg, k0 + i);
  }
  return b;   // (the caller's next __syncthreads() publishes the table)
}

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

// Cell: the footprint with ATen's zero-padding rule folded into the weights.  Corners that fall outside
// the grid (index -1 or N at the faces) contribute nothing in grid_sampler_3d; instead of testing each of
// the 8 corners, the low corner is shifted into [0, N-2] per axis and the weight of the out-of-range side
// is set to 0 (and moved to the other side), so all 8 corners i + {0,1} are always addressable:
//   i0 == -1  : corners (-1, 0), weights (w0, w1)  ->  corners (0, 1),     weights (w1, 0)
//   i0 == N-1 : corners (N-1, N), weights (w0, w1) ->  corners (N-2, N-1), weights (0, w0)
// (axes of size 1 use stride 0 for the "+1" corner, whose weight is always 0.)
struct Cell {
  int i[3];
  float w[3][2];
};

__device__ __forceinline__ void make_cell(const DevGrid& g, const Footprint& f, Cell& c) {
  const int N[3] = {g.X, g.Y, g.Z};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int i0 = f.i0[a];
    const float w0 = f.w[a][0], w1 = f.w[a][1];
    const bool lo = i0 < 0, hi = i0 >= N[a] - 1, wide = N[a] > 1;
    const float w0_at_top = (i0 == N[a] - 1) ? w0 : 0.0f;
    c.i[a] = lo ? 0 : (hi ? max(N[a] - 2, 0) : i0);
    c.w[a][0] = lo ? ((i0 == -1) ? w1 : 0.0f) : (hi ? (wide ? 0.0f : w0_at_top) : w0);
    c.w[a][1] = lo ? 0.0f : (hi ? (wide ? w0_at_top : 0.0f) : w1);
  }
}

// Interior test: the low corner and the +1 corner are in range on every axis (no zero padding involved).
__device__ __forceinline__ bool cell_is_interior(const DevGrid& g, const Footprint& f) {
  return ((unsigned)f.i0[0] < (unsigned)(g.X - 1)) && ((unsigned)f.i0[1] < (unsigned)(g.Y - 1)) &&
         ((unsigned)f.i0[2] < (unsigned)(g.Z - 1));
}
// make_cell with a wave-uniform shortcut: away from the faces (almost every sample) the footprint IS the cell.
__device__ __forceinline__ void make_cell_fast(const DevGrid& g, const Footprint& f, Cell& c) {
  if (__builtin_amdgcn_ballot_w64(!cell_is_interior(g, f)) == 0ull) {  // all active lanes interior
#pragma unroll
    for (int a = 0; a < 3; ++a) { c.i[a] = f.i0[a]; c.w[a][0] = f.w[a][0]; c.w[a][1] = f.w[a][1]; }
  } else {
    make_cell(g, f, c);
  }
}

// linear voxel index of the cell's low corner and the strides of the +1 corners
// (unsigned: indices are non-negative, so 64-bit addresses need no sign extension; 24-bit multiplies are full
// rate on CDNA while 32-bit integer multiplies are not -- grid dims are far below 2^24, the voxel count below 2^31)
struct CellAddr {
  unsigned base, sx, sy, sz;
};
__device__ __forceinline__ CellAddr cell_addr(const DevGrid& g, const Cell& c) {
  CellAddr a;
  // operands of the 24-bit multiplies: ix, Y < 2^24 and ix*Y + iy < X*Y < 2^24 (validated on the host)
  a.base = __umul24(__umul24((unsigned)c.i[0], (unsigned)g.Y) + (unsigned)c.i[1], (unsigned)g.Z) + (unsigned)c.i[2];
  a.sx = g.X > 1 ? g.Y * g.Z : 0;
  a.sy = g.Y > 1 ? g.Z : 0;
  a.sz = g.Z > 1 ? 1 : 0;
  return a;
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
