// voxe_render_common.hpp -- per-ray device code shared by the render kernels (forward, the two
// backward variants and the probe): thread->ray mapping, ray context, trilinear gather.
#pragma once

#include "voxe_device.hpp"

namespace voxe {

// ------------------------------------------------------------------------------------------------
// Thread -> ray mapping.
//   * XCD-aware: hardware places block b on XCD b % 8 and, inside an XCD, round-robin on its 32 CUs (observed,
//     not contractual; only speed depends on it).  Default: tile = block index, i.e. neighbouring tiles on
//     different XCDs.  The grid lives in the 256 MB Infinity Cache whatever the XCD, so balancing the eight XCDs
//     (tiles near the image centre cross more of the volume) pays more than keeping each XCD's private 4 MiB L2 on
//     a compact band of the frustum: the banded map is 5-11 % slower per kernel at 400x400 (measured, three cameras).
//   * image_width > 0: a 256-thread block is a 16x16 pixel tile, each wave an 8x8 sub-tile, so the
//     64 lanes of a wave touch a ~4x4x2 voxel neighbourhood per step (coalesced 16 B texel reads).
// ------------------------------------------------------------------------------------------------
// Per-ray depth-segment states.  The image-ordered backward splits every ray's march into segments of
// kSegLen samples that are processed by different waves; the forward saves, at every segment boundary
// k = b * kSegLen (b = 1 .. nseg-1), the transmittance T BEFORE sample k and the SUFFIX sums (csum[COUT], asum, dsum) of
// the samples k, k+1, ... (summed back to front: render_fwd_combine_kernel).  Layout [boundary-1][component][ray]
// (component-major: coalesced per ray run).
constexpr int kSegLen = VOXE_SEGMENT_SAMPLES;
// Small launches leave the chip under-filled, and what a wave then costs is its dependent chain of samples (~7 us each):
// half-length segments double the waves (64x64: backward 0.26 -> 0.15 ms, 100x100: 0.26 -> 0.22 ms); at 400x400 the
// chip is full either way and 16 / 32 are equal, 8 is 8 % slower (more state traffic, more partial windows).
#ifndef VOXE_SEG16_MAX_RAYS
#define VOXE_SEG16_MAX_RAYS 20000
#endif
#ifndef VOXE_SEG8_MAX_RAYS
#define VOXE_SEG8_MAX_RAYS 0     // (8-sample segments for the smallest launches: measured, no gain -- see DESIGN.md 4.9)
#endif
__host__ __device__ inline int seg_len_for(long long R) {
  if (R <= VOXE_SEG8_MAX_RAYS && kSegLen > 8) return 8;
  return (R <= VOXE_SEG16_MAX_RAYS && kSegLen > 16) ? 16 : kSegLen;
}
__host__ __device__ inline int num_segments(int S, int seg_len) { return (S + seg_len - 1) / seg_len; }
__device__ __forceinline__ long long ray_state_index(int boundary, int comp, int ncomp, long long R, long long r) {
  return ((long long)(boundary - 1) * ncomp + comp) * R + r;
}

// logical tile index of this block.  nt = number of tiles, ntx = tiles per image row (1 for linear ray order).
//   mode 0: interleaved -- tile = block index (neighbouring tiles on different XCDs; default)
//   mode 1: XCD bands   -- XCD x (blocks b % 8 == x) walks tiles [x*nt/8, (x+1)*nt/8): compulsory L2 traffic only
//   mode 2: row interleave -- XCD x walks tile rows x, x+8, x+16, ...: balances the XCDs when the work per
//           row varies (image centre vs borders) at the price of every XCD touching the whole frustum
// Tiles >= nt (padding of the launch) are reported as -1.
__device__ __forceinline__ int logical_tile_of(const DevCfg& c, int b, int nblocks, int ntx, int nty) {
  if (c.map_mode == 0) return b < ntx * nty ? b : -1;
  const int x = b & 7, slot = b >> 3;
  if (c.map_mode == 2) {
    const int row = x + 8 * (slot / ntx), col = slot % ntx;
    return row < nty ? row * ntx + col : -1;
  }
  const int per = nblocks >> 3;  // host launches a multiple of 8 blocks
  const int t = x * per + slot;
  return t < ntx * nty ? t : -1;
}
__device__ __forceinline__ int logical_tile(const DevCfg& c, int ntx, int nty) {
  return logical_tile_of(c, blockIdx.x, gridDim.x, ntx, nty);
}

// number of blocks to launch for nt = ntx * nty tiles under the mapping mode
static inline int blocks_for_tiles(int map_mode, long long ntx, long long nty) {
  if (map_mode == 0) return (int)((ntx * nty + 7) / 8 * 8);
  if (map_mode == 2) return (int)(8 * ((nty + 7) / 8) * ntx);
  return (int)((ntx * nty + 7) / 8 * 8);
}

// Image-ordered launches hold K = R / (H * W) images back to back (K = 1 unless VoxeRenderCfg::image_height says
// otherwise).  Pixel tiles of `side` x `side` pixels are laid out per image -- a tile never straddles two cameras --
// and numbered row-major over a virtual image of K * ceil(H / side) tile rows.
__host__ __device__ inline long long tile_rows_total(const DevCfg& c, int side) {
  const long long per = (c.image_height + side - 1) / side;
  const long long nimg = c.image_height > 0 ? c.R / ((long long)c.image_width * c.image_height) : 0;
  return per * (nimg > 0 ? nimg : 1);
}
// ray of pixel (px, row `py_in_tile` of tile row `ty`): false when the pixel is outside its image
__device__ __forceinline__ bool tile_pixel_ray(const DevCfg& c, int ty, int py_in_tile, int px, int side, long long& r) {
  const int per = (c.image_height + side - 1) / side;       // tile rows of one image
  const int img = ty / per, tyi = ty - img * per;
  const int py = tyi * side + py_in_tile;
  r = ((long long)img * c.image_height + py) * c.image_width + px;
  return px < c.image_width && py < c.image_height;
}

// b / nblocks: this block's index among the `nblocks` ray blocks of the launch (a kernel that runs several
// blocks per ray block, e.g. one per depth segment, passes blockIdx.x / n and gridDim.x / n).
__device__ __forceinline__ bool map_ray_block(const DevCfg& c, int b, int nblocks, long long& r) {
  const int tid = threadIdx.x;
  if (c.image_width > 0) {
    const int W = c.image_width;
    const int ntx = (W + 15) >> 4, nty = (int)tile_rows_total(c, 16);
    const int logical = logical_tile_of(c, b, nblocks, ntx, nty);
    if (logical < 0) return false;
    const int ty = logical / ntx, tx = logical - ty * ntx;
    const int wave = tid >> 6, lane = tid & 63;
    const int px = (tx << 4) + ((wave & 1) << 3) + (lane & 7);
    return tile_pixel_ray(c, ty, ((wave >> 1) << 3) + (lane >> 3), px, 16, r);
  }
  const int nt = (int)((c.R + 255) / 256);
  const int logical = logical_tile_of(c, b, nblocks, 1, nt);
  if (logical < 0) return false;
  r = (long long)logical * 256 + tid;
  return r < c.R;
}
__device__ __forceinline__ bool map_ray(const DevCfg& c, long long& r) {
  return map_ray_block(c, blockIdx.x, gridDim.x, r);
}

// ------------------------------------------------------------------------------------------------
// Per-ray state shared by forward / backward / probe
// ------------------------------------------------------------------------------------------------
template <int COUT, int NCM, int NCU>
struct RayCtx {
  static constexpr int C = COUT * NCM + 1;
  float o[3], d[3], dnorm;
  float basis[NCU];
  DepthGen dg;
  int k_lo, k_hi;

  __device__ __forceinline__ void init(const DevGrid& g, const DevCfg& c, long long r,
                                       const float* __restrict__ rays_o,
                                       const float* __restrict__ rays_d,
                                       const float* __restrict__ jitter) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { o[a] = rays_o[3 * r + a]; d[a] = rays_d[3 * r + a]; }
    // rays.directions.norm(dim=-1)  (accumulate.py:55, process.py:53)
    dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if constexpr (NCU > 1) {
      const float v[3] = {d[0] / dnorm, d[1] / dnorm, d[2] / dnorm};
      sh_basis<NCU>(v, basis);
    } else {
      basis[0] = kC0;
    }
    dg.near = c.near; dg.far = c.far;
    dg.lindisp = c.lindisp != 0;
    if (c.aabb_clip) {  // sample.py:187-202 (linear_disparity is not forwarded there)
      ray_aabb_bounds(g, o, d, dg.near, dg.far);
      dg.lindisp = false;
    }
    dg.S = c.S; dg.half = c.S >> 1;
    dg.step = 1.0f / (float)(c.S - 1);
    dg.perturb = c.perturb != 0;
    dg.jit = jitter ? jitter + r * c.S : nullptr;
    dg.base = jitter_base(c.key0, c.key1, r);
    dg.kc = INT_MIN;
    inside_range(g, c, dg, o, d, k_lo, k_hi);
  }

  __device__ __forceinline__ void point(float z, float (&p)[3]) const {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float dz = d[a] * z;  // sample.py:67: o + d * z (two roundings)
      p[a] = o[a] + dz;
    }
  }
};

// The 4-channel texel path of gather<>() in two halves, so that a kernel can put the loads of TWO samples in flight before
// it interpolates either (render_fwd_seg_kernel): the 8 corner texels (16 bytes each) and their interpolation --
// (r, g) and (b, sigma) pairs through v_pk_fma_f32 (2 FMAs per issue slot), corner order k = x + 2 y + 4 z.
__device__ __forceinline__ void load_texels4(const float* __restrict__ packed, const CellAddr& ad, float4 (&t)[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k)
    t[k] = reinterpret_cast<const float4*>(packed)[ad.base + (k & 1) * ad.sx + ((k >> 1) & 1) * ad.sy + (k >> 2) * ad.sz];
}
__device__ __forceinline__ void interp_texels4(const float4 (&t)[8], const Cell& cell, float& f0, float& f1, float& f2, float& v) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  float wxy[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) wxy[k] = cell.w[0][k & 1] * cell.w[1][k >> 1];
  v2f rg = {0.0f, 0.0f}, bs = {0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float w = wxy[k & 3] * cell.w[2][k >> 2];
    const v2f ww = {w, w};
    const v2f a = {t[k].x, t[k].y}, b = {t[k].z, t[k].w};
    rg = __builtin_elementwise_fma(a, ww, rg);
    bs = __builtin_elementwise_fma(b, ww, bs);
  }
  f0 = rg.x; f1 = rg.y; f2 = bs.x; v = bs.y;
}

// Interpolate density + features at the 8 corners of a Cell and evaluate
// v = interp(pre-activated density), rad_c = sum_j basis_j * interp(coef_cj)   (process.py:45-78, voxels.py:307-332).
// Values are interpolated with FMAs (only the index math has to round like the reference).
template <int COUT, int NCM, int NCU>
__device__ __forceinline__ void gather(const DevGrid& g, const float* __restrict__ packed, const Cell& cell,
                                       const float (&basis)[NCU], float& v, float (&rad)[COUT]) {
  constexpr int C = COUT * NCM + 1;
  const CellAddr ad = cell_addr(g, cell);
  float wxy[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) wxy[k] = cell.w[0][k & 1] * cell.w[1][k >> 1];
  float f[COUT * NCU];
#pragma unroll
  for (int i = 0; i < COUT * NCU; ++i) f[i] = 0.0f;
  v = 0.0f;
  if constexpr (C == 4) {
    float4 t[8];
    load_texels4(packed, ad, t);
    interp_texels4(t, cell, f[0], f[1], f[2], v);
  } else if constexpr (C == 2) {
    float2 t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      t[k] = reinterpret_cast<const float2*>(packed)[ad.base + (k & 1) * ad.sx + ((k >> 1) & 1) * ad.sy + (k >> 2) * ad.sz];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float w = wxy[k & 3] * cell.w[2][k >> 2];
      f[0] = fmaf(t[k].x, w, f[0]);
      v = fmaf(t[k].y, w, v);
    }
  } else {
    // View-dependent texels (13 / 28 / 49 channels): contract every corner with the SH basis first and interpolate the
    // COUT resulting values -- the same sum as interpolating COUT * NCU coefficients and contracting afterwards
    // (process.py:45-78), re-associated so that COUT accumulators stay live instead of COUT * NCU (48 at degree 3:
    // the kernels spilled 0.5 - 1.3 KB per lane before)
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) rad[ch] = 0.0f;
    // corners in flight at once: all 8 would be 8 x C loaded values (392 registers at degree 3)
    constexpr int kCornersInFlight = C > 28 ? 1 : (C > 13 ? 2 : 8);
#pragma unroll kCornersInFlight
    for (int k = 0; k < 8; ++k) {
      // (selects, not array indexing: k is a run-time value in the partially unrolled loop)
      const float wx = (k & 1) ? cell.w[0][1] : cell.w[0][0], wy = (k & 2) ? cell.w[1][1] : cell.w[1][0];
      const float w = (wx * wy) * ((k & 4) ? cell.w[2][1] : cell.w[2][0]);
      const float* __restrict__ src =
          packed + (long long)(ad.base + (k & 1) * ad.sx + ((k >> 1) & 1) * ad.sy + (k >> 2) * ad.sz) * C;
#pragma unroll
      for (int ch = 0; ch < COUT; ++ch) {
        float r = basis[0] * src[ch * NCM];
#pragma unroll
        for (int j = 1; j < NCU; ++j) r = fmaf(basis[j], src[ch * NCM + j], r);
        rad[ch] = fmaf(r, w, rad[ch]);
      }
      v = fmaf(src[C - 1], w, v);
    }
    return;
  }
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) {
    float r = basis[0] * f[ch * NCU];
#pragma unroll
    for (int j = 1; j < NCU; ++j) r = fmaf(basis[j], f[ch * NCU + j], r);
    rad[ch] = r;
  }
}

}  // namespace voxe
