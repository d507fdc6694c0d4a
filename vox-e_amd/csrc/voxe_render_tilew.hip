// voxe_render_tilew.hip -- forward of view-dependent grids (SH degree 1 - 3: 13 / 28 / 49-channel texels) for image-ordered
// rays with the tile's TEXELS staged in LDS (r06).
//
// Reference semantics: thre3d_atom/rendering/volumetric/utils/spherical_harmonics.py:87-116 (basis), process.py:45-67
// (radiance = sum_j basis_j coef_cj per corner-interpolated texel), accumulate.py:49-84 (compositing).
//
// render_fwd_seg_kernel<3, NCM, NCU> gathers the 8 corner texels of every sample from L1 / L2: 416 - 1568 bytes per lane and
// sample in 4-byte pieces at 52 - 196-byte strides, and the 64 rays of an 8x8-pixel tile ask for the same ~36 texels of a
// layer pair ~14 times.  Here one wave (tile x depth segment) keeps the sheared, ray-aligned window of the SH-0 window forward
// (voxe_render_tile.hip: fwd_window_march) -- a ring of layers along the march axis x 8 x 8 lateral voxels -- but of WHOLE
// texels: a layer is 64 x C floats (3.3 / 7 / 12.25 KB), copied once when the march reaches it
//   * march along x or y: a layer is eight z-runs of 8 texels = eight contiguous 8 C-float rows: 8 lanes per row, C (or C / 4
//     float4) strided loads per lane and layer, one base address per lane;
//   * march along z: 64 separate texels, one per lane;
// and the corner fetches are LDS reads (ds_read_b128 for 112-byte texels, dword pairs otherwise).  The contraction is
// gather<3, NCM, NCU>()'s, operation for operation (corner order x + 2 y + 4 z; per corner basis . coef per colour, then the
// corner weight): outputs and the per-sample (rad, v) the two-phase backward reads are BIT-IDENTICAL to render_fwd_seg_kernel
// (tests/test_hip_r06.py::test_wide_window_forward_*).  Samples whose footprint is not in the window and tiles that do not
// fit it take the global gather: results never depend on the window.
//
// Occupancy is set by LDS (ring x layer bytes per one-wave block): a ring as deep as the SH-0 window's (6 layers, the tile's
// whole extent along the march axis) leaves 3 blocks per CU at 112-byte texels -- less than one wave per SIMD, and that version
// was SLOWER than the gathers.  So the lanes of a tile do not march in step here: a lane whose footprint lies ahead of the
// resident layers WAITS for the slower lanes, the ring is 3 - 4 layers whatever the tile's obliqueness, and 5 - 11 blocks fit
// a CU (SH-2 400x400: 1.57 -> 0.86 ms; ring 6 / 5 / 4 / 3 / 2: 1.53 / 1.14 / 0.99 / 0.86 / 1.06 ms; 0.81 with the bank layout below).
#include "voxe_launch.hpp"
#include "voxe_render_common.hpp"
#include "voxe_tile_window.hpp"

#include <type_traits>

namespace voxe {
namespace {

#ifndef VOXE_FWDW_RING_13
#define VOXE_FWDW_RING_13 4
#endif
#ifndef VOXE_FWDW_RING_28
#define VOXE_FWDW_RING_28 3
#endif
#ifndef VOXE_FWDW_RING_49
#define VOXE_FWDW_RING_49 3
#endif
#ifndef VOXE_FWDW_PARITY
#define VOXE_FWDW_PARITY 1
#endif
#ifndef VOXE_FWDW_PARITY_ROWS
#define VOXE_FWDW_PARITY_ROWS 0
#endif
#ifndef VOXE_FWDW_LB
#define VOXE_FWDW_LB 3
#endif
#ifndef VOXE_FWDW_CORNERS   // corners whose LDS reads are in flight together (0: per texel width)
#define VOXE_FWDW_CORNERS 0
#endif
constexpr int kOrgW = 64;   // layers tabulated per block: keys key0 .. key0 + 63 (<= 1.7 layers per sample x 32 samples + ring)

template <int C>
struct WideTex {
  static constexpr bool kVec = (C % 4 == 0);            // 16-byte aligned texels (SH-2: 112 bytes): float4 copies and reads
  static constexpr int kW = kVec ? 4 : 1;               // floats per copy element
  static constexpr int kIter = C / kW;                  // copy elements per lane and layer
  static constexpr int kRing = C <= 13 ? VOXE_FWDW_RING_13 : (C <= 28 ? VOXE_FWDW_RING_28 : VOXE_FWDW_RING_49);
  static constexpr int kLayer = 64 * C;                 // floats per layer
  static constexpr int kCorners = VOXE_FWDW_CORNERS ? VOXE_FWDW_CORNERS : (C > 28 ? 1 : (C > 13 ? 4 : 8));
  typedef typename std::conditional<kVec, float4, float>::type Elem;
};

template <int C>
__device__ __forceinline__ typename WideTex<C>::Elem wide_zero() {
  if constexpr (WideTex<C>::kVec) return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  else return 0.0f;
}

// ds_read_b128 serves a wave in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- and only
// lanes of one group can conflict (banks (a / 4) mod 64: 16 slots of four banks; a 112-byte texel stride puts texel index i on
// slot 7 i mod 16).  So for 16-byte texel pieces (SH-2)
//   * a lane group is a 4 x 4-PIXEL quadrant of the tile (pix_of_lane), whose footprints span at most 4 x 4 texels of a layer;
//   * a layer stores its 8 x 8 texels in 4 x 4 blocks (texel_slot: i mod 16 = 4 (a mod 4) + b mod 4), so ANY 4 x 4 texels are
//     16 different slots.
// With lanes in pixel rows and texels in rows of 8, 74 % of the LDS cycles were bank conflicts (profiles/r06_sh_window.txt).
// 4-byte reads (SH-1 / 3: odd strides, banks mod 32, groups of 32 lanes = 4 x 8 pixels) keep rows: i mod 32 = 8 (a mod 4) + b.
__host__ __device__ constexpr int popc32(unsigned x) { int n = 0; for (int i = 0; i < 32; ++i) n += (x >> i) & 1u; return n; }
__host__ __device__ constexpr int pix_of_lane_c(int l) {
  const unsigned mask0 = 0x0FF0F00Fu;
  const int l5 = l & 31;
  const bool member = (mask0 >> l5) & 1u;
  const int i16 = popc32((member ? mask0 : ~mask0) & ((1u << l5) - 1u));
  const int g = (l >> 5) * 2 + (member ? 0 : 1);
  return (((g >> 1) * 4 + (i16 >> 2)) << 3) + (g & 1) * 4 + (i16 & 3);
}
__host__ __device__ constexpr int lane_of_pix_c(int p) { for (int l = 0; l < 64; ++l) if (pix_of_lane_c(l) == p) return l; return -1; }
__device__ __forceinline__ int pix_of_lane(int l) {
  const unsigned mask0 = 0x0FF0F00Fu;
  const int l5 = l & 31;
  const bool member = (mask0 >> l5) & 1u;
  const int i16 = __popc((member ? mask0 : ~mask0) & ((1u << l5) - 1u));
  const int g = (l >> 5) * 2 + (member ? 0 : 1);
  return (((g >> 1) * 4 + (i16 >> 2)) << 3) + (g & 1) * 4 + (i16 & 3);
}
// position of lateral texel (a, b) inside its layer
template <bool VEC>
__host__ __device__ constexpr int texel_slot(int a, int b) {
  return VEC ? ((a >> 2) << 5) | ((b >> 2) << 4) | ((a & 3) << 2) | (b & 3) : a * 8 + b;
}

// The march of one (tile, depth segment), march axis M a compile-time constant.
template <int M, bool STRATA, int NCM, int NCU>
__device__ __forceinline__ void fwd_wide_march(const DevGrid& g, const DevCfg& c, const float* __restrict__ packed,
                                               RayCtx<3, NCM, NCU>& rc, const int lane, const bool has, const int k_lo,
                                               const int k_hi, const int ref,
                                               float* __restrict__ tex, int4* __restrict__ org, const SegDepth<STRATA> sd,
                                               float (&csum)[3], float& asum, float& dsum, float& T,
                                               float4* __restrict__ sample_out, const int ks, const int strict) {
  constexpr int COUT = 3, C = COUT * NCM + 1;
  typedef WideTex<C> WT;
  typedef typename WT::Elem Elem;
  constexpr int RING = WT::kRing, W = WT::kW, NIT = WT::kIter;
  constexpr int U = (M == 0) ? 1 : 0, V = (M == 2) ? 1 : 2;   // lateral axes (v = z whenever m != z)
  constexpr int kCtr = Lat<8>::kCentre;
  // ---- window geometry from the reference ray (as in fwd_window_march) ----
  const int N[3] = {g.X, g.Y, g.Z};
  float U0[3], DU[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float ro = readlane_f32(rc.o[a], ref), rd = readlane_f32(rc.d[a], ref);
    const float half = 0.5f * (float)N[a];
    U0[a] = ((ro * g.scale[a] + g.bias[a]) + 1.0f) * half - 0.5f;
    DU[a] = rd * g.scale[a] * half;
  }
  const int sgn = (DU[M] < 0.0f) ? -1 : 1;
  const int neg = sgn < 0 ? 1 : 0;
  const float inv = (DU[M] != 0.0f) ? 1.0f / DU[M] : 0.0f;
  const float Bu = DU[U] * inv, Au = U0[U] - Bu * U0[M];
  const float Bv = DU[V] * inv, Av = U0[V] - Bv * U0[M];
  const int sx = g.Y * g.Z, sy = g.Z;
  const int stride_m = (M == 0) ? sx : ((M == 1) ? sy : 1);
  const int stride_u = (U == 0) ? sx : sy;
  const int stride_v = (V == 1) ? sy : 1;
  const long long total = (long long)g.X * g.Y * g.Z * C;   // floats of the packed grid
  auto minkey = [&](int pm) { return sgn > 0 ? pm : -(pm + 1); };
  const int Nm2 = N[M] - 2;
  // lower key of a footprint's two layers, from the index make_cell() will use (the low corner clamped into [0, N - 2])
  auto key_of = [&](const Footprint& f) { return minkey(min(max(f.i0[M], 0), Nm2)); };

  float z_cur = 0.0f;
  Footprint fp_cur;
  fp_cur.inside = false;
#pragma unroll
  for (int a = 0; a < 3; ++a) { fp_cur.i0[a] = 0; fp_cur.w[a][0] = fp_cur.w[a][1] = 0.0f; }
  int first_key = INT_MAX;
  if (has) {
    z_cur = sd.z(rc.dg, k_lo);
    float p[3];
    rc.point(z_cur, p);
    footprint(g, p, fp_cur);
    first_key = key_of(fp_cur);
  }
  const int key0 = wave_min_i32(first_key);
  // per layer: lateral origin (u, v), voxel offset of the origin, float offset of the layer's ring slot
  {
    const int im = sgn * (key0 + lane);
    const int ou = (int)floorf(Au + Bu * (float)im) - kCtr, ov = (int)floorf(Av + Bv * (float)im) - kCtr;
    org[lane] = make_int4(ou, ov, im * stride_m + ou * stride_u + ov * stride_v, (lane % RING) * WT::kLayer);
  }
  __syncthreads();
  // ---- the copy of a layer: lane -> (row, piece) for x / y marches, lane -> texel for z marches ----
  const int la = lane >> 3, lb8 = lane & 7;
  const int src_lane = (M == 2) ? (la * stride_u + lb8 * stride_v) * C : la * stride_u * C + lb8 * W;   // floats from the layer origin
  constexpr int kStep = (M == 2) ? W : 8 * W;   // floats between a lane's consecutive pieces (source side)
  // VOXE_FWDW_PARITY: the lanes of a quadrant are spread over two adjacent layers (jitter; lanes that waited), and the ring
  // slots of two layers are whole bank periods apart: the same (a, b) of both layers would share its banks.  Layers of odd key
  // store texel (a, b) at the position of (a ^ 2, b ^ 2) -- compact footprints in adjacent layers then meet on different slots
  // (the slot index's bits 3 and 1 flip: ^ 10).
  constexpr int kParXor = !VOXE_FWDW_PARITY ? 0 : (WT::kVec ? 10 : VOXE_FWDW_PARITY_ROWS);   // (rows of 8: a ^ 2 = index ^ 16)
  int dst_slot[NIT], dst_part[NIT];             // lane constants: texel position in the layer, float offset inside the texel
#pragma unroll
  for (int t = 0; t < NIT; ++t) {
    if constexpr (M == 2) {
      dst_slot[t] = texel_slot<WT::kVec>(la, lb8); dst_part[t] = t * W;
    } else if constexpr (WT::kVec) {
      const int piece = lb8 + 8 * t, b = piece / (C / 4), part = piece - b * (C / 4);   // (row of 8 texels = 2 C float4)
      dst_slot[t] = texel_slot<WT::kVec>(la, b); dst_part[t] = part * 4;
    } else {
      dst_slot[t] = la * 8; dst_part[t] = lb8 + 8 * t;   // (a row is contiguous: texel b = piece / C follows from the offset)
    }
  }
  const long long span = (long long)(7 * stride_u + 7 * stride_v + 1) * C;
  auto fetch_layer = [&](int key, Elem (&buf)[NIT]) {
    const int idx = key - key0;                       // wave-uniform
#pragma unroll
    for (int t = 0; t < NIT; ++t) buf[t] = wide_zero<C>();
    if (idx < kOrgW) {
      const int vb = __builtin_amdgcn_readfirstlane(org[idx].z);
      const long long lo = (long long)vb * C;
      const float* __restrict__ src = packed + lo + src_lane;
      if (lo >= 0 && lo + span <= total) {            // the whole 8 x 8 layer lies inside the allocation (wave-uniform)
#pragma unroll
        for (int t = 0; t < NIT; ++t) buf[t] = *reinterpret_cast<const Elem*>(src + t * kStep);
      } else {
        // grid border: a piece outside the allocation is skipped.  Pieces inside it are read whatever voxel they belong to --
        // the address of an in-grid voxel is linear in (m, u, v), so every texel a footprint can ask for gets its own data,
        // and the others are never read
#pragma unroll
        for (int t = 0; t < NIT; ++t) {
          const long long e = lo + src_lane + t * kStep;
          if (e >= 0 && e + W <= total) buf[t] = *reinterpret_cast<const Elem*>(src + t * kStep);
        }
      }
    }
  };
  auto slot_of = [&](int key) { return ((key - key0) % RING) * WT::kLayer; };   // wave-uniform; == the table's .w inside the table
  auto store_layer = [&](int key, const Elem (&buf)[NIT]) {
    float* __restrict__ dst = tex + slot_of(key);
    const int px = (key & 1) ? kParXor : 0;       // wave-uniform
#pragma unroll
    for (int t = 0; t < NIT; ++t) *reinterpret_cast<Elem*>(dst + (dst_slot[t] ^ px) * C + dst_part[t]) = buf[t];
  };
  int base = key0;            // layers base .. base + RING - 1 are in LDS
  Elem pend[NIT];
  for (int i = 0; i < RING; ++i) {
    fetch_layer(base + i, pend);
    store_layer(base + i, pend);
  }
  __syncthreads();
  // The lanes of the tile do NOT march in step: a lane whose sample lies ahead of the resident layers waits for the slower
  // lanes (`base` is the layer of the slowest one, so that lane is always served), and the ring only has to hold the two layers of
  // a footprint plus what is in flight -- not the tile's whole extent along the march axis (a 6-layer ring of 112-byte texels
  // left room for 3 one-wave blocks per CU, and tiles more oblique than 4.5 layers went ray by ray).  Every ray still
  // composites its own samples in order: the arithmetic per ray is render_fwd_seg_kernel's.
  int k_cur = k_lo;
  bool left = has;
  while (__ballot(left) != 0ull) {
    if (left) {
      const int kl = key_of(fp_cur);                   // lower key of the footprint's two layers; the other is kl + 1
      const bool ahead = (kl - base) >= (RING - 1);
      if (!fp_cur.inside || !ahead) {
        const int k = k_cur;
        const float z = z_cur;
        const Footprint fp = fp_cur;
        const bool last = (k == c.S - 1);
        float z_next = z;
        if (!last) {
          z_next = sd.z(rc.dg, k + 1);
          float pn[3];
          rc.point(z_next, pn);
          footprint(g, pn, fp_cur);
          z_cur = z_next;
        }
        ++k_cur;
        left = k_cur <= k_hi;
        if (fp.inside) {
          Cell cell;
          make_cell_fast(g, fp, cell);
          const int pu = cell.i[U], pv = cell.i[V];
          const int il = kl - key0;
          const int ilc = min(max(il, 0), kOrgW - 2);
          const int4 ol = org[ilc], oh = org[ilc + 1];
          const int4 o0 = sgn > 0 ? ol : oh, o1 = sgn > 0 ? oh : ol;   // origins / ring slots of layers pm, pm + 1
          const int a0 = pu - o0.x, b0 = pv - o0.y, a1 = pu - o1.x, b1 = pv - o1.y;
          // (kl >= base by construction -- base is the minimum over the lanes' current keys -- asserted here so that a ray running
          //  against the tile's march direction could only ever fall back to the gather, never read a recycled layer)
          const bool fits = (kl >= base) && (il == ilc) && ((unsigned)a0 < 7u) && ((unsigned)b0 < 7u) && ((unsigned)a1 < 7u) && ((unsigned)b1 < 7u);
          float v, rad[COUT];
          if (fits) {
            // corner texel (layer dm, lateral du, dv): float offset in the ring
            int coff[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int dm = (q >> M) & 1, du = (q >> U) & 1, dv = (q >> V) & 1;   // compile-time after unrolling
              const int aa = (dm ? a1 : a0) + du, bb = (dm ? b1 : b0) + dv;
              coff[q] = (dm ? o1.w : o0.w) + (texel_slot<WT::kVec>(aa, bb) ^ (((kl + dm + neg) & 1) ? kParXor : 0)) * C;   // (key of layer pm + dm: kl + dm, or kl + 1 - dm when keys run against the index)
            }
            // gather<3, NCM, NCU>()'s view-dependent branch with the corner texels read from the window
#pragma unroll
            for (int ch = 0; ch < COUT; ++ch) rad[ch] = 0.0f;
            v = 0.0f;
#pragma unroll
            for (int q0 = 0; q0 < 8; q0 += WT::kCorners) {
              Elem s[WT::kCorners][NIT];
#pragma unroll
              for (int qq = 0; qq < WT::kCorners; ++qq) {
                const float* __restrict__ src = tex + coff[q0 + qq];
#pragma unroll
                for (int t = 0; t < NIT; ++t) s[qq][t] = *reinterpret_cast<const Elem*>(src + t * W);
              }
#pragma unroll
              for (int qq = 0; qq < WT::kCorners; ++qq) {
                const int q = q0 + qq;
                const float wq = (cell.w[0][q & 1] * cell.w[1][(q >> 1) & 1]) * cell.w[2][q >> 2];
                const float* sv = reinterpret_cast<const float*>(&s[qq][0]);   // (constant indices after unrolling: registers)
#pragma unroll
                for (int ch = 0; ch < COUT; ++ch) {
                  float r = rc.basis[0] * sv[ch * NCM];
#pragma unroll
                  for (int j = 1; j < NCU; ++j) r = fmaf(rc.basis[j], sv[ch * NCM + j], r);
                  rad[ch] = fmaf(r, wq, rad[ch]);
                }
                v = fmaf(sv[C - 1], wq, v);
              }
            }
          } else {
            gather<3, NCM, NCU>(g, packed, cell, rc.basis, v, rad);
            if (strict) v = __int_as_float(0x7fc00000);   // test aid (VoxeDispatch::fwd_window = 2): mark what the window did not serve
          }
          if (sample_out) sample_out[(long long)(k - ks) * 64] = make_float4(rad[0], rad[1], rad[2], v);
          const float sigma = post_activate(g.post_act, v);
          const float dl = last ? kInfinity : (z_next - z);
          const float delta = dl * rc.dnorm;
          const float e = fast_exp(-(sigma * delta));
          const float alpha = 1.0f - e;
          const float om = 1.0f - alpha;
          const float wgt = alpha * T;
          T = T * om;
#pragma unroll
          for (int ch = 0; ch < COUT; ++ch) csum[ch] = fmaf(sigmoidf(rad[ch]), wgt, csum[ch]);
          asum = asum + wgt;
          dsum = fmaf(z, wgt, dsum);
        }
      }
    }
    // ---- slide the window: its first layer is the one the slowest lane needs next ----
    const int newbase = wave_min_i32(left ? key_of(fp_cur) : INT_MAX);
    if (newbase > base && newbase != INT_MAX) {   // wave-uniform
      __syncthreads();
      // (measured, profiles/r06_sh_window.txt: requesting the next layer a sample early -- stores deferred as in the SH-0 window
      //  forward, or a speculative fetch of layer base + RING -- costs more in registers / usable layers than the wait it saves:
      //  +5 ... +28 %; the other waves of the CU cover the copy)
      for (int key = max(base + RING, newbase); key < newbase + RING; ++key) {
        fetch_layer(key, pend);
        store_layer(key, pend);
      }
      __syncthreads();
      base = newbase;
    }
  }
}

template <int NCM, int NCU>
__global__ __launch_bounds__(64, (3 * NCM + 1) <= 13 ? VOXE_FWDW_LB : ((3 * NCM + 1) <= 28 ? 2 : 1)) void render_fwd_tilew_kernel(DevGrid g, DevCfg c, const float* __restrict__ packed,
                                                                            const float* __restrict__ rays_o,
                                                                            const float* __restrict__ rays_d,
                                                                            const float* __restrict__ jitter,
                                                                            float* __restrict__ segbuf,
                                                                            float4* __restrict__ sample_fwd, const float fit_lat,
                                                                            const float fit_m, const float zdom, const float max_adv, const int strict) {
  constexpr int COUT = 3, NC = COUT + 3, C = COUT * NCM + 1;
  typedef WideTex<C> WT;
  __shared__ float4 tex4[WT::kRing * WT::kLayer / 4];
  __shared__ int4 org[kOrgW];
  __shared__ float2 strat[64];
  float* tex = reinterpret_cast<float*>(tex4);
  const int lane = threadIdx.x;
  const int nseg = num_segments(c.S, c.seg_len);
  const int nrb = gridDim.x / nseg;                    // tile slots (segment-major block order, like render_fwd_seg_kernel)
  const int seg = blockIdx.x / nrb, rb = blockIdx.x - seg * nrb;
  const int W = c.image_width;
  const int ntx = (W + 7) >> 3, nty = (int)tile_rows_total(c, 8);
  const int tile = logical_tile_of(c, rb, nrb, ntx, nty);
  if (tile < 0) return;
  const int ty = tile / ntx, tx = tile - ty * ntx;
  long long r_px;
  // pixel of this lane, row-major in the tile: 16-byte reads want a lane group = a 4 x 4-pixel quadrant (see pix_of_lane)
  const int pl = WT::kVec ? pix_of_lane(lane) : lane;
  constexpr int L0 = WT::kVec ? lane_of_pix_c(0) : 0, L1 = WT::kVec ? lane_of_pix_c(1) : 1, L8 = WT::kVec ? lane_of_pix_c(8) : 8,
                L27 = WT::kVec ? lane_of_pix_c(27) : 27;   // lanes of pixels (0,0), (1,0), (0,1), (3,3)
  const bool alive = tile_pixel_ray(c, ty, pl >> 3, (tx << 3) + (pl & 7), 8, r_px);
  const long long r = alive ? r_px : 0;
  RayCtx<COUT, NCM, NCU> rc;
  rc.init(g, c, r, rays_o, rays_d, jitter);
  const int ks = seg * c.seg_len, ke = min(c.S, ks + c.seg_len) - 1;
  const int k_lo = max(rc.k_lo, ks);
  const int k_hi = alive ? min(rc.k_hi, ke) : k_lo - 1;
  const bool has = k_lo <= k_hi;
  const int kmin = wave_min_i32(has ? k_lo : INT_MAX);
  const int kmax = wave_max_i32(has ? k_hi : -1);
  // slots of (tile, segment, sample, lane) -- render_fwd_seg_kernel's tile_lane + (seg * seg_len + (k - ks)) * 64
  float4* __restrict__ sample_out =
      sample_fwd ? sample_fwd + ((long long)tile * nseg + seg) * c.seg_len * 64 + pl : nullptr;
  float csum[COUT] = {0.0f, 0.0f, 0.0f};
  float asum = 0.0f, dsum = 0.0f, T = 1.0f;
  // per tile (wave-uniform): through the window along axis m, or ray by ray -- render_fwd_tile_kernel's decision
  int m = -1;
  int ref = 0;
  if (kmin <= kmax) {
    const unsigned long long hm = __ballot(has);
    ref = ((hm >> L27) & 1ull) ? L27 : (__ffsll((long long)hm) - 1);
    const unsigned long long am = __ballot(alive);
    if ((am >> L1 & 1ull) && (am >> L8 & 1ull)) {
      const int N[3] = {g.X, g.Y, g.Z};
      const float zref = readlane_f32(rc.dg.zlin(ke), L0);
      float ad[3], e3[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float sc = g.scale[a] * 0.5f * (float)N[a];
        const float da = readlane_f32(rc.d[a], L0);
        ad[a] = fabsf(readlane_f32(rc.d[a], ref) * sc);
        e3[a] = 7.0f * (fabsf((readlane_f32(rc.d[a], L1) - da) * sc * zref) + fabsf((readlane_f32(rc.d[a], L8) - da) * sc * zref));
      }
      const int mxy = ad[0] >= ad[1] ? 0 : 1;
      const int mm = (ad[2] >= fabsf(zdom) * ad[mxy]) ? 2 : mxy;
      float lat = 0.0f, alongm = 0.0f;
#pragma unroll
      for (int a = 0; a < 3; ++a) { if (a == mm) alongm = e3[a]; else lat = fmaxf(lat, e3[a]); }
      const float adv = ad[mm] * fabsf(readlane_f32(rc.dg.zlin(ke) - rc.dg.zlin(ke > 0 ? ke - 1 : 0), ref));
      if (lat <= fit_lat && alongm <= fit_m && adv <= max_adv && (mm != 2 || zdom < 0.0f)) m = mm;
    }
  }
  const bool use_strata = VOXE_TILE_STRATA && !c.aabb_clip && (ke + 1 - ks) < 64;
  if (use_strata && lane <= ke + 1 - ks && ks + lane < c.S) strat[lane] = depth_stratum(rc.dg, ks + lane);
  __syncthreads();
  auto march_rays = [&](auto strata_tag) {   // ray by ray (the loop of render_fwd_seg_kernel)
    const SegDepth<decltype(strata_tag)::value> sd{strat, ks};
    if (has) {
      float z_next = sd.z(rc.dg, k_lo);
      for (int k = k_lo; k <= k_hi; ++k) {
        const float z = z_next;
        const bool last = (k == c.S - 1);
        if (!last) z_next = sd.z(rc.dg, k + 1);
        float p[3];
        rc.point(z, p);
        Footprint fp;
        footprint(g, p, fp);
        if (!fp.inside) continue;
        Cell cell;
        make_cell_fast(g, fp, cell);
        float v, rad[COUT];
        gather<COUT, NCM, NCU>(g, packed, cell, rc.basis, v, rad);
        if (sample_out) sample_out[(long long)(k - ks) * 64] = make_float4(rad[0], rad[1], rad[2], v);
        const float sigma = post_activate(g.post_act, v);
        const float dl = last ? kInfinity : (z_next - z);
        const float delta = dl * rc.dnorm;
        const float e = fast_exp(-(sigma * delta));
        const float alpha = 1.0f - e;
        const float om = 1.0f - alpha;
        const float wgt = alpha * T;
        T = T * om;
#pragma unroll
        for (int ch = 0; ch < COUT; ++ch) csum[ch] = fmaf(sigmoidf(rad[ch]), wgt, csum[ch]);
        asum = asum + wgt;
        dsum = fmaf(z, wgt, dsum);
      }
    }
  };
  auto march_tile = [&](auto strata_tag) {
    constexpr bool ST = decltype(strata_tag)::value;
    const SegDepth<ST> sd{strat, ks};
    if (m < 0) march_rays(strata_tag);
    else if (m == 0) fwd_wide_march<0, ST, NCM, NCU>(g, c, packed, rc, lane, has, k_lo, k_hi, ref, tex, org, sd, csum, asum, dsum, T, sample_out, ks, strict);
    else if (m == 1) fwd_wide_march<1, ST, NCM, NCU>(g, c, packed, rc, lane, has, k_lo, k_hi, ref, tex, org, sd, csum, asum, dsum, T, sample_out, ks, strict);
    else fwd_wide_march<2, ST, NCM, NCU>(g, c, packed, rc, lane, has, k_lo, k_hi, ref, tex, org, sd, csum, asum, dsum, T, sample_out, ks, strict);
  };
  if (use_strata) march_tile(std::true_type{});
  else march_tile(std::false_type{});
  if (!alive) return;
  if (strict && m < 0 && has) asum = __int_as_float(0x7fc00000);   // test aid: this tile marched ray by ray
  const long long base = (long long)seg * NC;
  segbuf[(base + 0) * c.R + r] = T;
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) segbuf[(base + 1 + ch) * c.R + r] = csum[ch];
  segbuf[(base + 1 + COUT) * c.R + r] = asum;
  segbuf[(base + 2 + COUT) * c.R + r] = dsum;
}

}  // namespace

// Full view-dependent evaluation (NCU == NCM > 1) of an image-ordered render on a grid with no degenerate axis; VoxeDispatch::
// fwd_window = -1 switches the window off (ray-ordered forward), like the SH-0 window forward.
bool fwd_tilew_supported(const DevGrid& g, const HostCfg& c, const FwdArgs& a, int cout, int ncm, int ncu) {
  if (cout != 3 || ncm <= 1 || ncu != ncm || c.image_width <= 0) return false;
  if ((3 * ncm + 1) % 4 == 0 && (reinterpret_cast<uintptr_t>(a.packed) & 15) != 0) return false;   // 112-byte texels move as float4
  if (g.X < 2 || g.Y < 2 || g.Z < 2) return false;
  if ((long long)g.X * g.Y * g.Z * (3 * ncm + 1) >= (1ll << 31)) return false;   // float offsets inside a layer are 32-bit
  return c.disp.fwd_window >= 0;
}
void launch_fwd_tilew(const DevGrid& g, const HostCfg& c, int ncm, const FwdArgs& a, hipStream_t st) {
  const int nseg = num_segments(c.S, c.seg_len);
  const int nb = blocks_for_tiles(c.map_mode, (c.image_width + 7) / 8, tile_rows_total(c, 8)) * nseg;
  // z-dominant tiles march along z through the window as well (a layer normal to z is 64 separate texels of 52 - 196 bytes: the
  // copy pays from 52-byte texels on, unlike the 16-byte texels of SH-0; profiles/r06_sh_window.txt: camera 26 -35 ... -40 %)
  const float zdom = disp_or(c.disp.fwd_zdom, -1.0f), max_adv = disp_or(c.disp.fwd_max_adv, 1.7f);
  const float fit_lat = disp_or(c.disp.fwd_fit_lat, 5.5f);
  float4* sf = reinterpret_cast<float4*>(a.sample_fwd);
#define VOXE_FWDW(NCM)                                                                                                        \
  render_fwd_tilew_kernel<NCM, NCM><<<nb, 64, 0, st>>>(g, c, a.packed, a.rays_o, a.rays_d, a.jitter, a.segbuf, sf, fit_lat,   \
                                                       disp_or(c.disp.fwd_fit_m, 12.0f),  \
                                                       zdom, max_adv, c.disp.fwd_window == 2 ? 1 : 0)
  if (ncm == 4) VOXE_FWDW(4);
  else if (ncm == 9) VOXE_FWDW(9);
  else VOXE_FWDW(16);
#undef VOXE_FWDW
}

}  // namespace voxe
