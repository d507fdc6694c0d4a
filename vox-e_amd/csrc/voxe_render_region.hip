// voxe_render_region.hip -- SPACE-BINNED render (forward + backward) for rays that share no voxels with their neighbours
// in the launch: random training batches of the reconstruction loop (modules/trainers.py:288-351), low-resolution /
// multi-view images whose pixels are more than a voxel apart.
//
// Why: such rays still meet -- 32768 rays x 144 in-AABB samples x 8 corners = 38 M texel fetches and 38 M gradient
// deposits land on 4 M voxels -- but not inside a wave.  The ray-ordered kernels then pay for it twice: the forward
// gather pulls ~4.5 cache lines per sample through L2 (5x slower per ray than an image-ordered render), and the backward
// has nothing to combine in LDS, so it is a scatter bound by the ~20 G atomic cache-line requests/s of the memory side
// (profiles/r01_microbench_atomics.md).  Here the work is binned by SPACE instead of by ray:
//
//   1. region_seg_kernel      index math only (depths, footprints; no gather): every ray is cut into SEGMENTS of consecutive
//                             samples whose 2x2x2 footprint starts in the same 8x8x8-cell REGION of the grid; each
//                             (ray, depth segment) lane owns its slots of the segment table -- no global cursor -- and
//                             takes the segment's rank inside its region from one returning atomic per segment;
//   2. region_scan / fill     counting sort of the segments by region (no host round trip);
//   3. region_fwd_kernel      one block per region: the region's 9x9x9 texels are staged in LDS ONCE (coalesced rows),
//                             lanes = segments, trilinear gathers come from LDS; every segment composites with a LOCAL
//                             transmittance starting at 1 (compositing is associative: the depth-segmented forward of
//                             voxe_render.hip does the same with fixed 32-sample segments);
//   4. region_fold_kernel     one thread per (ray, depth segment) folds that lane's segments, the lanes of a ray meet in LDS
//                             -> colour / depth / acc / disparity, and for every segment the transmittance in front of
//                             it and the SUFFIX sums behind it (Horner, back to front) for the backward;
//   5. region_bwd_kernel      one block per region again: texels in LDS, lanes = segments starting from their saved
//                             state; the 8 corners x C channels of every sample go into a 9x9x9-voxel LDS window of
//                             doubles (ds_add_f64, as in the tile kernel); ONE dense flush per region.
//
// Global memory sees each region's texels once per pass and ~160 atomic requests per region instead of ~3.6 per sample.
// A lane whose (ray, depth segment) needs more segment slots than it owns (a ray zig-zagging through region corners)
// puts the rest of its samples into the GENERIC bin: same kernels, texels from global memory, global atomics.
// Per-sample math: identical to render_fwd_seg_kernel / render_bwd_packed_scatter_kernel (same device functions).
// Reference: thre3d_atom/rendering/volumetric/{sample,process,accumulate}.py, thre3d_reprs/voxels.py (+ autograd).
#include <limits.h>
#include <stdlib.h>

#include "voxe_device.hpp"
#include "voxe_launch.hpp"
#include "voxe_render_common.hpp"

namespace voxe {

// region edges in cells (low-corner indices) along x / y / z; a window is one voxel wider per axis
#ifndef VOXE_REGION_BX
#define VOXE_REGION_BX 8
#endif
#ifndef VOXE_REGION_BY
#define VOXE_REGION_BY 8
#endif
#ifndef VOXE_REGION_BZ
#define VOXE_REGION_BZ 8
#endif
constexpr int kRBX = VOXE_REGION_BX, kRBY = VOXE_REGION_BY, kRBZ = VOXE_REGION_BZ;
constexpr int kRWX = kRBX + 1, kRWY = kRBY + 1, kRWZ = kRBZ + 1;
constexpr int kRWin = kRWX * kRWY * kRWZ;   // 729 voxels for 8 x 8 x 8 cells
constexpr int kRPlane = kRWin + 7;          // channel plane of the gradient window (doubles), padded off the bank period
// r04: parity-class banked gradient window of the 4-channel backward (the map of voxe_render_tile.hip's WinMap in three
// lateral-free dimensions): index (doubles) = 32 (((x >> 1) * kPY + (y >> 1)) * kPZ + (z >> 1)) + 8 ch + 4 (x & 1) + 2 (y & 1)
// + (z & 1) for window voxel (x, y, z) in [0, 9)^3.  The lanes pick the ORDER of a sample's 8 corners so that instruction
// (cc, j) of lane L adds to parity class cc ^ (lane bits 1..3) and channel (j + lane bits 0, 4) & 3: the lanes of a half-wave
// hit different bank pairs whatever the rays do (segments of unrelated rays meet in a region: r02 / r03 measured 0.84 of the
// launch in LDS cycles, a third of them conflicts).  9 is odd: the map spends 10^3 / 9^3 of the r03 window (32 KB for 23.5).
#ifndef VOXE_REGION_PCB
#define VOXE_REGION_PCB 1
#endif
constexpr int kPX = (kRWX + 1) / 2, kPY = (kRWY + 1) / 2, kPZ = (kRWZ + 1) / 2;
constexpr int kPcbDoubles = 32 * kPX * kPY * kPZ;
#ifndef VOXE_REGION_PCB_MIN_RAYS
#define VOXE_REGION_PCB_MIN_RAYS 65536
#endif
constexpr long long kBankedMinRays = VOXE_REGION_PCB_MIN_RAYS;   // launches from this many rays on take the banked window (see launch_bwd_region_t)
__device__ __forceinline__ int pcb_index(int x, int y, int z, int ch) {
  return 32 * (((x >> 1) * kPY + (y >> 1)) * kPZ + (z >> 1)) + (ch << 3) + ((x & 1) << 2) + ((y & 1) << 1) + (z & 1);
}
#ifndef VOXE_REGION_CHUNK
#define VOXE_REGION_CHUNK 16       // longest segment (samples): bounds the lane divergence of the region kernels
#endif
#ifndef VOXE_REGION_BLOCK
#define VOXE_REGION_BLOCK 256      // threads of a region block (its waves share the LDS windows)
#endif
#ifndef VOXE_REGION_SLOTS
#define VOXE_REGION_SLOTS 16
#endif
constexpr int kSlotsPerLane = VOXE_REGION_SLOTS;  // segment slots of one (ray, depth segment)
#ifndef VOXE_REGION_SH3
#define VOXE_REGION_SH3 1          // SH degree 3 (49-channel texels) on this route too: the staging kernels take 151.6 KB of LDS
#endif
#ifndef VOXE_REGION_STAGE_FWD_NCU
#define VOXE_REGION_STAGE_FWD_NCU 1    // view-dependent grids: stage a region's whole texels in LDS up to this many coefficients per colour
                                       // (r04, 32 400 random rays, staged / from L2: forward SH-1 0.49 / 0.47, SH-2 1.01 / 0.71, SH-3 2.78 / 1.13 ms;
                                       // backward SH-1 1.10 / 1.16, SH-2 2.49 / 2.27, SH-3 5.41 / 3.74 -- 81.6 / 151.6 KB leave ONE block per CU)
#endif
#ifndef VOXE_REGION_STAGE_BWD_NCU
#define VOXE_REGION_STAGE_BWD_NCU 4
#endif
#ifndef VOXE_REGION_STRATA
#define VOXE_REGION_STRATA 512      // depth strata tabulated per block (BlockStrata, voxe_device.hpp) for S up to this; 0: off
#endif
#ifndef VOXE_REGION_FWDVAL
#define VOXE_REGION_FWDVAL 1         // view-dependent grids: the forward keeps (rad, v) per sample for the backward's source pass
#endif
#ifndef VOXE_REGION_SH_WC
#define VOXE_REGION_SH_WC 7        // gradient channels per deposit pass of a view-dependent grid (window: WC x 5.9 KB of LDS).
                                   // Swept (32 400 random rays, backward ms, SH-1 / SH-2): 4 -> 1.21 / 2.56, 7 -> 1.10 / 2.48 (13 = 2 / 4
                                   // passes instead of 4 / 7: fewer footprint re-computations), 13 -> 1.22 / 3.27 (77 KB: 2 blocks per CU)
#endif
#ifndef VOXE_REGION_BWD_TEX
#define VOXE_REGION_BWD_TEX 2      // backward: stage the region's texels in LDS too: 1 always | 0 never (gather them from L1 / L2) |
                                   // 2 (r05) by launch: the channel-strided texels of wide grids (NCM > 1) always, packed 16- / 8-byte
                                   // texels only from VOXE_REGION_TEX_MIN_RAYS rays on.  Below, they come from L1 / L2: without the
                                   // 11.7 KB the banked kernel holds 36 KB of LDS and, asked to fit 128 registers, runs 4 blocks per CU
                                   // instead of 3 (backward, staged -> from L2: 32 761 rays 0.218 -> 0.212 ms, 65 536 0.278 -> 0.263,
                                   // the reconstruction batch 0.323 -> 0.296; 160 000 rays 0.485 -> 0.513: regions that full re-use a
                                   // staged texel often enough to pay for the residency)
#endif
#ifndef VOXE_REGION_TEX_MIN_RAYS
#define VOXE_REGION_TEX_MIN_RAYS 110000
#endif
constexpr unsigned kNoRegion = 0xFFFFFFFFu;
// Segments of a region are grouped by LENGTH class (longest first): the lanes of a wave then run similar trip counts
// instead of all waiting for the longest segment among 64 random ones (mean length ~5, cap 16).
constexpr int kLenClasses = 4;
__host__ __device__ inline int len_class(int len) { return len >= 9 ? 0 : (len >= 5 ? 1 : (len >= 3 ? 2 : 3)); }
constexpr unsigned kRegionMask = 0x00FFFFFFu;   // slot_region = region | class << 24

__host__ __device__ inline int regions_along(int N, int edge) { return ((N > 1 ? N - 1 : 1) + edge - 1) / edge; }

// Slot numbering: slot j of lane L is slot_of(L, j) = j * nlanes + L (j-major: threads that walk the lanes -- fill, fold --
// touch consecutive records; the region kernels scatter / gather by slot id).
struct BinScratch {
  unsigned* slot_region;  // [nslots] region | length class << 24 of every USED segment slot (region nreg: the generic bin);
                          // slot j of a lane is used iff j < lane_n
  unsigned* slot_pos;     // [nslots] rank of the segment inside its region (returned by the counting atomic)
  uint2* slot_seg;        // [nslots] (ray, k0 | k1 << 16)
  uint4* sorted;          // [nslots] (ray, k0 | k1 << 16, slot, -) grouped by region
  float4* part;           // [nslots][2] by SLOT: forward partials of the segment (Tseg, csum[0..2] | asum, dsum)
  float4* state;          // [nslots][2] by SLOT, for the backward: (T in front of the segment, U_c[0..2] | U_a, U_d) with U
                          // the suffix sums of (csum, asum, dsum) from the segment's first sample on, RELATIVE to T
  unsigned* lane_n;       // [R * nseg] segments of every (ray, depth segment) lane
  unsigned* count;        // [(nreg + 1) * 4 + 1] segments per (region, length class) (+ generic bin; last entry stays 0)
  unsigned* start;        // [(nreg + 1) * 4 + 1] exclusive scan of `count`: first position in `sorted`; a region's segments are
                          // start[region * 4] .. start[(region + 1) * 4]
  unsigned long long* blockhist;  // [NB][nreg + 1] region_seg_lds_kernel's per-block counters (4 x 16 bits per region); null: global ranks
  uint4* blockoff;        // [NB][nreg + 1] per class: segments of the region in the blocks before this one
};

__device__ __forceinline__ size_t slot_of(size_t lane, unsigned j, long long nlanes) { return (size_t)j * (size_t)nlanes + lane; }
// Lane numbering: (ray r, depth segment s) is lane s * R + r (r03; r02: r * nseg + s).  region_seg_kernel's waves hold 64 rays of
// ONE depth segment: segment-major ids make their table writes contiguous runs (8 rays of an image tile = one 32-byte sector
// of a 4-byte table) instead of one 4-byte word per sector (PMC: 293 MB written for 40 MB of records); the fold kernel's
// threads (8 rays x nseg segments per wave) still read whole 256-byte runs of the 32-byte segment records.
__device__ __forceinline__ long long lane_of(long long r, int seg, long long R) { return (long long)seg * R + r; }

// ---- ray context of a segment lane: origin, direction, depth generator (no sample range, no SH basis) ----------------------
// LDS stride (floats) of a full view-dependent texel: 13 -> 16, 28 -> 32 (16-byte aligned rows for ds_read_b128)
__host__ __device__ constexpr int tex_stride(int cm) { return (cm + 3) / 4 * 4; }

template <int NCU>
__device__ __forceinline__ void ray_basis(const float (&d)[3], float dnorm, float (&basis)[NCU]) {
  if constexpr (NCU > 1) {
    const float v[3] = {d[0] / dnorm, d[1] / dnorm, d[2] / dnorm};   // (as RayCtx::init)
    sh_basis<NCU>(v, basis);
  } else {
    basis[0] = kC0;
  }
}

// paired launch (DevCfg::pair_R): the ray-array index and jitter keys of launch ray r
__device__ __forceinline__ long long pair_source(const DevCfg& c, long long r, uint32_t& k0, uint32_t& k1) {
  k0 = c.key0; k1 = c.key1;
  if (c.pair_R > 0 && r >= c.pair_R) { k0 = c.key0b; k1 = c.key1b; return r - c.pair_R; }
  return r;
}

struct SegRay {
  float o[3], d[3], dnorm;
  DepthGen dg;
  __device__ __forceinline__ void init(const DevGrid& g, const DevCfg& c, long long r_launch, const float* __restrict__ rays_o,
                                       const float* __restrict__ rays_d, const float* __restrict__ jitter) {
    uint32_t k0, k1;
    const long long r = pair_source(c, r_launch, k0, k1);
#pragma unroll
    for (int a = 0; a < 3; ++a) { o[a] = rays_o[3 * r + a]; d[a] = rays_d[3 * r + a]; }
    dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    dg.near = c.near; dg.far = c.far;
    dg.lindisp = c.lindisp != 0;
    if (c.aabb_clip) { ray_aabb_bounds(g, o, d, dg.near, dg.far); dg.lindisp = false; }
    dg.S = c.S; dg.half = c.S >> 1;
    dg.step = 1.0f / (float)(c.S - 1);
    dg.perturb = c.perturb != 0;
    dg.jit = jitter ? jitter + r * c.S : nullptr;
    dg.base = jitter_base(k0, k1, r);
    dg.kc = INT_MIN;
  }
  __device__ __forceinline__ void point(float z, float (&p)[3]) const {
#pragma unroll
    for (int a = 0; a < 3; ++a) { const float dz = d[a] * z; p[a] = o[a] + dz; }
  }
};

// ---- pass 1: segments (index math only) ------------------------------------------------------------------------------------
// One UNIT of the pass = one wave = 64 rays (an 8x8 pixel tile of an image-ordered launch) x one depth segment; `vb` numbers the
// units as the blocks of the r02 launch did (segment-major), `rank(region, class)` hands out the segment's rank.
//   * region_seg_kernel       (r02)  one unit per 64-thread block; rank = one RETURNING global atomic per segment on the
//                             (region, class) counters.  The memory side retires ~20 G atomic requests/s whatever the scope
//                             (profiles/r01_microbench_atomics.md): 1.9 M segments of a reconstruction batch = 139 us, of which the
//                             index math is 53 (profiles/r05_recon_experiments.txt).  Kept for grids whose region table does not
//                             fit in LDS (above ~184^3).
//   * region_seg_lds_kernel   (r05)  1024-thread blocks, each wave walks a run of units; the block ranks its segments in an LDS
//                             table -- one 64-bit word per region, four 16-bit class counters, ds_add_rtn_u64 -- and leaves the
//                             table in `blockhist[block][region]`; region_colscan_kernel turns the NB tables into per-block
//                             offsets and the (region, class) totals.  No global atomic anywhere in the pass.
template <typename Rank>
__device__ __forceinline__ void seg_unit(const DevGrid& g, const DevCfg& c, const float* __restrict__ rays_o,
                                         const float* __restrict__ rays_d, const float* __restrict__ jitter, const BinScratch& bs,
                                         const int nreg, const int vb, const int nvb, const int lane, const BlockStrata& bst,
                                         Rank&& rank) {
  const int nseg = num_segments(c.S, c.seg_len);
  const int nrb = nvb / nseg;
  const int seg = vb / nrb;
  // image-ordered launches (sparse / multi-view images): a wave is an 8x8 pixel tile, so its lanes cross the same regions
  // at about the same samples; unordered rays: 64 consecutive rays
  long long r;
  if (c.image_width > 0) {
    const int ntx = (c.image_width + 7) >> 3, nty = (int)tile_rows_total(c, 8);
    const int t = logical_tile_of(c, vb - seg * nrb, nrb, ntx, nty);
    if (t < 0) return;
    const int ty = t / ntx, tx = t - ty * ntx;
    if (!tile_pixel_ray(c, ty, lane >> 3, (tx << 3) + (lane & 7), 8, r)) return;
  } else {
    const int nt = (int)((c.R + 63) / 64);
    const int logical = logical_tile_of(c, vb - seg * nrb, nrb, 1, nt);
    if (logical < 0) return;
    r = (long long)logical * 64 + lane;
    if (r >= c.R) return;
  }
  RayCtx<3, 1, 1> rc;   // (origin, direction, depth generator, conservative in-AABB sample range)
  {
    DevCfg cj = c;        // (paired launch: the second half's jitter keys, the first half's rays)
    const long long rs = pair_source(c, r, cj.key0, cj.key1);
    rc.init(g, cj, rs, rays_o, rays_d, jitter);
  }
  const int ks = seg * c.seg_len, ke = min(c.S, ks + c.seg_len) - 1;
  const int k_lo = max(rc.k_lo, ks), k_hi = min(rc.k_hi, ke);
  const long long nlanes = c.R * nseg, lane_id = lane_of(r, seg, c.R);
  if (k_lo > k_hi) { bs.lane_n[lane_id] = 0u; return; }   // (every lane writes its count: no memset of the table)
  const int nry = regions_along(g.Y, kRBY), nrz = regions_along(g.Z, kRBZ);
  int nslots = 0;
  unsigned cur = kNoRegion;
  int k0 = 0, k_prev = 0;
  auto emit = [&](int k_end) {
    if (cur == kNoRegion) return;
    const unsigned cls = (unsigned)len_class(k_end - k0 + 1);
    const size_t sl = slot_of((size_t)lane_id, (unsigned)nslots, nlanes);
    bs.slot_region[sl] = cur | (cls << 24);
    bs.slot_seg[sl] = make_uint2((unsigned)r, (unsigned)k0 | ((unsigned)k_end << 16));
    bs.slot_pos[sl] = rank(cur, cls);
    ++nslots;
  };
  for (int k = k_lo; k <= k_hi; ++k) {
    const float z = bst.z(rc.dg, k);
    float p[3];
    rc.point(z, p);
    Footprint fp;
    footprint(g, p, fp);
    if (!fp.inside) continue;          // contributes nothing (process.py:83): not part of any segment
    Cell cell;
    make_cell_fast(g, fp, cell);
    unsigned region = (unsigned)(((cell.i[0] / kRBX) * nry + cell.i[1] / kRBY) * nrz + cell.i[2] / kRBZ);
    if (cur == (unsigned)nreg) region = cur;              // the generic bin keeps the rest of this lane's samples
    if (region != cur || (cur != (unsigned)nreg && k - k0 + 1 > VOXE_REGION_CHUNK)) {
      emit(k_prev);
      cur = (nslots == kSlotsPerLane - 1) ? (unsigned)nreg : region;   // last slot of the lane: generic bin
      k0 = k;
    }
    k_prev = k;
  }
  emit(k_prev);
  bs.lane_n[lane_id] = (unsigned)nslots;
}

__global__ __launch_bounds__(64) void region_seg_kernel(DevGrid g, DevCfg c, const float* __restrict__ rays_o,
                                                        const float* __restrict__ rays_d, const float* __restrict__ jitter,
                                                        BinScratch bs, const int nreg) {
  const int lane = threadIdx.x;
  const int nseg = num_segments(c.S, c.seg_len);
  const int seg = blockIdx.x / (gridDim.x / nseg);
  // depth strata of this block's segment, tabulated once (before any lane leaves)
  __shared__ float2 strat[64];
  const int ks_blk = seg * c.seg_len;
  const BlockStrata bst = build_block_strata(strat, VOXE_REGION_STRATA ? 64 : 0, c, ks_blk, min(c.S, ks_blk + c.seg_len) - ks_blk, lane, 64);
  __syncthreads();
  const bool coherent = c.image_width > 0;
  seg_unit(g, c, rays_o, rays_d, jitter, bs, nreg, (int)blockIdx.x, (int)gridDim.x, lane, bst, [&](unsigned cur, unsigned cls) {
    const unsigned key = cur * kLenClasses + cls;
    unsigned pos;
    if (coherent) {
      // wave-aggregated ranking: the lanes that close a segment in the same loop trip group by counter; one returning
      // atomic per group (its first lane) instead of one per lane
      unsigned long long todo = __ballot(1);
      int leader = lane;
      unsigned rank = 0, size = 1;
      while (todo) {
        const int l = __ffsll((long long)todo) - 1;
        const unsigned key_l = (unsigned)__builtin_amdgcn_readlane((int)key, l);
        const unsigned long long m = __ballot(key == key_l) & todo;
        if (key == key_l) {
          leader = l;
          rank = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
          size = (unsigned)__popcll(m);
        }
        todo &= ~m;
      }
      unsigned base = 0;
      if (lane == leader) base = atomicAdd(bs.count + key, size);
      base = (unsigned)__shfl((int)base, leader, 64);
      pos = base + rank;
    } else {
      pos = atomicAdd(bs.count + key, 1u);   // rank inside (region, class)
    }
    return pos;
  });
}

// r05: block-local ranking in LDS.  slot_pos = rank inside (block, region, class) | block << 16; the fill pass adds
// blockoff[block][region].class and start[region, class].  A block never sees 65 536 segments (the host bounds its lanes).
constexpr int kSegLdsThreads = 1024, kSegLdsWaves = kSegLdsThreads / 64;
__global__ __launch_bounds__(kSegLdsThreads) void region_seg_lds_kernel(DevGrid g, DevCfg c, const float* __restrict__ rays_o,
                                                                        const float* __restrict__ rays_d,
                                                                        const float* __restrict__ jitter, BinScratch bs,
                                                                        const int nreg, const int nvb, const int units_per_wave) {
  extern __shared__ unsigned long long seg_hist[];   // [nreg + 1]: class c of a region in bits 16 c .. 16 c + 15
  __shared__ float2 strat_all[kSegLdsWaves][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i <= nreg; i += kSegLdsThreads) seg_hist[i] = 0ull;
  __syncthreads();
  const int nseg = num_segments(c.S, c.seg_len);
  const int nrb = nvb / nseg;
  const int u0 = ((int)blockIdx.x * kSegLdsWaves + wave) * units_per_wave;
  int seg_tab = -1;
  float2* const strat = strat_all[wave];
  BlockStrata bst{strat, 0, false};
  for (int u = u0; u < min(nvb, u0 + units_per_wave); ++u) {
    const int seg = u / nrb;
    if (seg != seg_tab) {   // (wave-uniform; the wave's lanes have re-converged here)
      seg_tab = seg;
      const int ks_blk = seg * c.seg_len;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      bst = build_block_strata(strat, VOXE_REGION_STRATA ? 64 : 0, c, ks_blk, min(c.S, ks_blk + c.seg_len) - ks_blk, lane, 64);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    seg_unit(g, c, rays_o, rays_d, jitter, bs, nreg, u, nvb, lane, bst, [&](unsigned cur, unsigned cls) {
      const unsigned long long old = atomicAdd(&seg_hist[cur], 1ull << (16 * cls));
      return ((unsigned)(old >> (16 * cls)) & 0xFFFFu) | ((unsigned)blockIdx.x << 16);
    });
  }
  __syncthreads();
  unsigned long long* __restrict__ out = bs.blockhist + (size_t)blockIdx.x * (size_t)(nreg + 1);
  for (int i = tid; i <= nreg; i += kSegLdsThreads) out[i] = seg_hist[i];
}

// Per-block offsets and totals from the NB block tables: thread = (region, chunk of NB / 16 consecutive blocks); the 16 chunk
// sums of a region meet in LDS.  blockoff[b][region] = segments of (region, class) in blocks < b; count[region * 4 + class] = total.
constexpr int kColRegions = 64, kColChunks = 16;
__global__ __launch_bounds__(kColRegions * kColChunks) void region_colscan_kernel(BinScratch bs, const int nreg, const int nb) {
  __shared__ uint4 chunk_sum[kColChunks][kColRegions];
  const int rl = threadIdx.x % kColRegions, q = threadIdx.x / kColRegions;
  const int region = (int)blockIdx.x * kColRegions + rl;
  const int per = nb / kColChunks;               // (the host launches a multiple of 16 blocks)
  const bool live = region <= nreg;
  const size_t stride = (size_t)(nreg + 1);
  const unsigned long long* __restrict__ h = bs.blockhist + (size_t)(q * per) * stride + (size_t)region;
  uint4 sum = make_uint4(0u, 0u, 0u, 0u);
  if (live) {
#pragma unroll 4
    for (int i = 0; i < per; ++i) {
      const unsigned long long v = h[(size_t)i * stride];
      sum.x += (unsigned)v & 0xFFFFu; sum.y += (unsigned)(v >> 16) & 0xFFFFu;
      sum.z += (unsigned)(v >> 32) & 0xFFFFu; sum.w += (unsigned)(v >> 48);
    }
  }
  chunk_sum[q][rl] = sum;
  __syncthreads();
  if (!live) return;
  uint4 run = make_uint4(0u, 0u, 0u, 0u);
  for (int j = 0; j < q; ++j) { const uint4 t = chunk_sum[j][rl]; run.x += t.x; run.y += t.y; run.z += t.z; run.w += t.w; }
  uint4* __restrict__ o = bs.blockoff + (size_t)(q * per) * stride + (size_t)region;
#pragma unroll 4
  for (int i = 0; i < per; ++i) {
    const unsigned long long v = h[(size_t)i * stride];    // (second read: L2)
    o[(size_t)i * stride] = run;
    run.x += (unsigned)v & 0xFFFFu; run.y += (unsigned)(v >> 16) & 0xFFFFu;
    run.z += (unsigned)(v >> 32) & 0xFFFFu; run.w += (unsigned)(v >> 48);
  }
  if (q == kColChunks - 1) reinterpret_cast<uint4*>(bs.count)[region] = run;   // (kLenClasses == 4)
}

// ---- pass 2: counting sort of the segments by region ---------------------------------------------------------------------
__global__ __launch_bounds__(1024) void region_scan_kernel(const unsigned* __restrict__ count, unsigned* __restrict__ start,
                                                           int n) {
  // exclusive prefix sum of `count` in ONE block (n = a few thousand .. 131 k counters).  Every thread owns a run of
  // 4 * per4 consecutive counters and moves them as 16-byte vectors with all loads in flight at once (a scalar loop over
  // the run costs one dependent memory round trip per element: 50 us at 32 k counters, measured); the 1024 run totals are
  // scanned in LDS.  Both arrays are 256-byte aligned and padded to a multiple of 4 entries by the host.
  __shared__ unsigned partial[1024];
  const int tid = threadIdx.x;
  const int per4 = ((n + 1023) / 1024 + 3) / 4;          // uint4 vectors per thread
  const int lo = tid * per4 * 4;
  const uint4* __restrict__ c4 = reinterpret_cast<const uint4*>(count);
  uint4* __restrict__ s4 = reinterpret_cast<uint4*>(start);
  unsigned sum = 0;
#pragma unroll 8
  for (int q = 0; q < per4; ++q) {
    if (lo + q * 4 < n) {                                 // (entries past n inside the last vector are padding: 0)
      const uint4 v = c4[(lo >> 2) + q];
      sum += v.x + v.y + v.z + v.w;
    }
  }
  partial[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const unsigned t = tid >= off ? partial[tid - off] : 0u;
    __syncthreads();
    partial[tid] += t;
    __syncthreads();
  }
  unsigned run = tid > 0 ? partial[tid - 1] : 0u;
#pragma unroll 8
  for (int q = 0; q < per4; ++q) {
    if (lo + q * 4 < n) {
      const uint4 v = c4[(lo >> 2) + q];                  // (second read: L2-resident)
      uint4 o;
      o.x = run; run += v.x;
      o.y = run; run += v.y;
      o.z = run; run += v.z;
      o.w = run; run += v.w;
      s4[(lo >> 2) + q] = o;
    }
  }
}

__global__ __launch_bounds__(256) void region_fill_kernel(BinScratch bs, long long nlanes, const int nreg, const int lds_ranks) {
  // one thread per (ray, depth segment) lane: its USED slots only (mean ~4 of 16).  Three rounds of independent loads (slot
  // records; start + block offset; -) instead of a dependent chain per slot: 47 -> 2x us on a reconstruction batch (r05).
  const long long lane = (long long)blockIdx.x * 256 + threadIdx.x;
  if (lane >= nlanes) return;
  const unsigned n = bs.lane_n[lane];
  unsigned rcv[kSlotsPerLane], posv[kSlotsPerLane];
  uint2 sgv[kSlotsPerLane];
#pragma unroll
  for (int j = 0; j < kSlotsPerLane; ++j) {
    if ((unsigned)j < n) {
      const size_t sl = slot_of((size_t)lane, (unsigned)j, nlanes);
      rcv[j] = bs.slot_region[sl];
      posv[j] = bs.slot_pos[sl];
      sgv[j] = bs.slot_seg[sl];
    }
  }
  unsigned base[kSlotsPerLane];
#pragma unroll
  for (int j = 0; j < kSlotsPerLane; ++j) {
    if ((unsigned)j < n) {
      const unsigned region = rcv[j] & kRegionMask, cls = rcv[j] >> 24;
      base[j] = bs.start[region * kLenClasses + cls];
      if (lds_ranks) {   // (slot_pos = rank inside the block | block << 16: region_seg_lds_kernel)
        const unsigned* __restrict__ off = reinterpret_cast<const unsigned*>(bs.blockoff + (size_t)(posv[j] >> 16) * (size_t)(nreg + 1) + region);
        base[j] += off[cls];
        posv[j] &= 0xFFFFu;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < kSlotsPerLane; ++j) {
    if ((unsigned)j < n) {
      const size_t sl = slot_of((size_t)lane, (unsigned)j, nlanes);
      bs.sorted[base[j] + posv[j]] = make_uint4(sgv[j].x, sgv[j].y, (unsigned)sl, 0u);
    }
  }
}

// ---- the region's texels in LDS ----------------------------------------------------------------------------------------------
// tex[voxel of the 9x9x9 window][C] floats, C = COUT + 1: (coefficient 0 of every colour, pre-activated density) -- all a
// single-group render (SH-0 / diffuse / attention) reads.  Voxels beyond the grid's far faces are zero (their weights are).
template <int COUT, int NCM>
__device__ __forceinline__ void load_window(const DevGrid& g, const float* __restrict__ packed, float* __restrict__ tex,
                                            int ox, int oy, int oz, int tid) {
  constexpr int C = COUT + 1, CM = COUT * NCM + 1;
  for (int v = tid; v < kRWin; v += VOXE_REGION_BLOCK) {
    const int x = ox + v / (kRWY * kRWZ), y = oy + (v / kRWZ) % kRWY, z = oz + v % kRWZ;
    const bool in = x < g.X && y < g.Y && z < g.Z;
    const long long vox = ((long long)x * g.Y + y) * g.Z + z;
    if constexpr (CM == 4) {
      const float4 t = in ? reinterpret_cast<const float4*>(packed)[vox] : make_float4(0.f, 0.f, 0.f, 0.f);
      reinterpret_cast<float4*>(tex)[v] = t;
    } else {
#pragma unroll
      for (int ch = 0; ch < C; ++ch) tex[v * C + ch] = in ? packed[vox * CM + (ch == COUT ? CM - 1 : ch * NCM)] : 0.0f;
    }
  }
}

// trilinear gather from the LDS window: same products / FMA order as gather<>() of voxe_render_common.hpp for C == 4 / 2
template <int COUT>
__device__ __forceinline__ void gather_lds(const float* __restrict__ tex, int idx0, const Cell& cell, float& v,
                                           float (&rad)[COUT]) {
  constexpr int C = COUT + 1;
  float wxy[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) wxy[k] = cell.w[0][k & 1] * cell.w[1][k >> 1];
  if constexpr (C == 4) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f rg = {0.0f, 0.0f}, bs = {0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float4 t = reinterpret_cast<const float4*>(tex)[idx0 + (k & 1) * (kRWY * kRWZ) + ((k >> 1) & 1) * kRWZ + (k >> 2)];
      const float w = wxy[k & 3] * cell.w[2][k >> 2];
      const v2f ww = {w, w};
      const v2f a = {t.x, t.y}, b = {t.z, t.w};
      rg = __builtin_elementwise_fma(a, ww, rg);
      bs = __builtin_elementwise_fma(b, ww, bs);
    }
    rad[0] = kC0 * rg.x; rad[1] = kC0 * rg.y; rad[2] = kC0 * bs.x; v = bs.y;
  } else {
    float f = 0.0f;
    v = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float2 t = reinterpret_cast<const float2*>(tex)[idx0 + (k & 1) * (kRWY * kRWZ) + ((k >> 1) & 1) * kRWZ + (k >> 2)];
      const float w = wxy[k & 3] * cell.w[2][k >> 2];
      f = fmaf(t.x, w, f);
      v = fmaf(t.y, w, v);
    }
    rad[0] = kC0 * f;
  }
}

// View-dependent grids (SH degree 1 / 2: 13 / 28-channel texels): the WHOLE texel of every window voxel in LDS, rows of
// CP = tex_stride(CM) floats; a corner is contracted with the ray's basis first (gather<>() of voxe_render_common.hpp:
// same products, same order), so COUT accumulators stay live.
template <int CM>
__device__ __forceinline__ void load_window_full(const DevGrid& g, const float* __restrict__ packed, float* __restrict__ tex,
                                                 int ox, int oy, int oz, int tid) {
  constexpr int CP = tex_stride(CM);
  for (int e = tid; e < kRWin * CM; e += VOXE_REGION_BLOCK) {
    const int v = e / CM, ch = e - v * CM;
    const int x = ox + v / (kRWY * kRWZ), y = oy + (v / kRWZ) % kRWY, z = oz + v % kRWZ;
    const bool in = x < g.X && y < g.Y && z < g.Z;
    tex[v * CP + ch] = in ? packed[(((long long)x * g.Y + y) * g.Z + z) * CM + ch] : 0.0f;
  }
}
template <int COUT, int NCM, int NCU>
__device__ __forceinline__ void gather_lds_sh(const float* __restrict__ tex, int idx0, const Cell& cell,
                                              const float (&basis)[NCU], float& v, float (&rad)[COUT]) {
  constexpr int CM = COUT * NCM + 1, CP = tex_stride(CM);
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) rad[ch] = 0.0f;
  v = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float w = (cell.w[0][k & 1] * cell.w[1][(k >> 1) & 1]) * cell.w[2][k >> 2];
    const float4* __restrict__ t4 =
        reinterpret_cast<const float4*>(tex + (idx0 + (k & 1) * (kRWY * kRWZ) + ((k >> 1) & 1) * kRWZ + (k >> 2)) * CP);
    float t[CP];
#pragma unroll
    for (int q = 0; q < CP / 4; ++q) { const float4 x = t4[q]; t[4 * q] = x.x; t[4 * q + 1] = x.y; t[4 * q + 2] = x.z; t[4 * q + 3] = x.w; }
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) {
      float r = basis[0] * t[ch * NCM];
#pragma unroll
      for (int j = 1; j < NCU; ++j) r = fmaf(basis[j], t[ch * NCM + j], r);
      rad[ch] = fmaf(r, w, rad[ch]);
    }
    v = fmaf(t[CM - 1], w, v);
  }
}

struct RegionBlock {
  int ox, oy, oz;   // window origin (voxels)
  bool generic;     // the generic bin: texels from global memory, global atomics
};
constexpr int kGenericBlocks = 256;   // blocks that share the generic bin (no LDS involved: any number can work on it)
__device__ __forceinline__ RegionBlock region_block(const DevGrid& g, unsigned region, int nreg) {
  RegionBlock b;
  b.generic = region >= (unsigned)nreg;
  const int nry = regions_along(g.Y, kRBY), nrz = regions_along(g.Z, kRBZ);
  b.oz = (int)(region % (unsigned)nrz) * kRBZ;
  b.oy = (int)((region / (unsigned)nrz) % (unsigned)nry) * kRBY;
  b.ox = (int)(region / (unsigned)(nrz * nry)) * kRBX;
  return b;
}

// ---- pass 3: forward, one block per region -----------------------------------------------------------------------------------
template <int COUT, int NCM, int NCU>
// (r05: 4-channel texels need 91 registers = 5 waves per SIMD; asked to fit 80 they run 6 -- 133 -> 123 us on the reconstruction batch;
//  7 is equal, 8 spills: 160 us.  View-dependent instantiations keep their budget.)
__global__ __launch_bounds__(VOXE_REGION_BLOCK, NCU == 1 ? 6 : 1) void region_fwd_kernel(DevGrid g, DevCfg c, const float* __restrict__ packed,
                                                                       const float* __restrict__ rays_o,
                                                                       const float* __restrict__ rays_d,
                                                                       const float* __restrict__ jitter, BinScratch bs,
                                                                       const int nreg, const int stage,
                                                                       float4* __restrict__ fwdval) {
  constexpr int C = COUT + 1;
  // single-group renders (SH-0 / diffuse / attention) stage (coefficient 0 of every colour, density); view-dependent ones
  // the whole texel (dynamic LDS: 46.6 KB at degree 1, 81.6 KB at degree 2)
  __shared__ float tex_small[NCU == 1 ? kRWin * C : 1];
  extern __shared__ float4 tex_dyn[];
  float* const tex = NCU == 1 ? tex_small : reinterpret_cast<float*>(tex_dyn);
  const int tid = threadIdx.x;
  const unsigned region = min(blockIdx.x, (unsigned)nreg);     // blocks nreg .. nreg + kGenericBlocks - 1: the generic bin
  const unsigned first = bs.start[region * kLenClasses];
  const unsigned n = bs.start[(region + 1) * kLenClasses] - first;
  if (n == 0) return;                       // block-uniform
  const RegionBlock rb = region_block(g, blockIdx.x, nreg);
  // stage == 0 (degree-3 texels, launch_fwd_region_t): the region's segments march together but fetch their texels from L1 / L2
  const bool from_global = rb.generic || stage == 0;
  if (!from_global) {
    if constexpr (NCU == 1) load_window<COUT, NCM>(g, packed, tex, rb.ox, rb.oy, rb.oz, tid);
    else load_window_full<COUT * NCM + 1>(g, packed, tex, rb.ox, rb.oy, rb.oz, tid);
  }
  __shared__ float2 strat[VOXE_REGION_STRATA ? VOXE_REGION_STRATA : 1];
  const BlockStrata bst = build_block_strata(strat, VOXE_REGION_STRATA, c, 0, c.S, tid, VOXE_REGION_BLOCK);
  __syncthreads();
  const unsigned i_begin = rb.generic ? (blockIdx.x - (unsigned)nreg) * VOXE_REGION_BLOCK + tid : tid;
  const unsigned i_step = rb.generic ? kGenericBlocks * VOXE_REGION_BLOCK : VOXE_REGION_BLOCK;
  for (unsigned i = i_begin; i < n; i += i_step) {
    const uint4 rec = bs.sorted[first + i];
    const long long r = rec.x;
    const int k0 = (int)(rec.y & 0xFFFFu), k1 = (int)(rec.y >> 16);
    SegRay ray;
    ray.init(g, c, r, rays_o, rays_d, jitter);
    float basis[NCU];
    ray_basis<NCU>(ray.d, ray.dnorm, basis);
    float csum[3] = {0.0f, 0.0f, 0.0f};
    float asum = 0.0f, dsum = 0.0f, T = 1.0f;
    float z_next = bst.z(ray.dg, k0);
    for (int k = k0; k <= k1; ++k) {
      const float z = z_next;
      const bool last = (k == c.S - 1);
      if (!last) z_next = bst.z(ray.dg, k + 1);
      float p[3];
      ray.point(z, p);
      Footprint fp;
      footprint(g, p, fp);
      if (!fp.inside) continue;
      Cell cell;
      make_cell_fast(g, fp, cell);
      float v, rad[COUT];
      if (from_global) {
        if (!rb.generic) {   // (a segment's samples outside its region belong to another segment, as below)
          const int lx = cell.i[0] - rb.ox, ly = cell.i[1] - rb.oy, lz = cell.i[2] - rb.oz;
          if ((unsigned)lx >= (unsigned)kRBX || (unsigned)ly >= (unsigned)kRBY || (unsigned)lz >= (unsigned)kRBZ) continue;
        }
        gather<COUT, NCM, NCU>(g, packed, cell, basis, v, rad);
      } else {
        const int lx = cell.i[0] - rb.ox, ly = cell.i[1] - rb.oy, lz = cell.i[2] - rb.oz;
        if ((unsigned)lx >= (unsigned)kRBX || (unsigned)ly >= (unsigned)kRBY || (unsigned)lz >= (unsigned)kRBZ) continue;
        if constexpr (NCU == 1) gather_lds<COUT>(tex, (lx * kRWY + ly) * kRWZ + lz, cell, v, rad);
        else gather_lds_sh<COUT, NCM, NCU>(tex, (lx * kRWY + ly) * kRWZ + lz, cell, basis, v, rad);
      }
      if constexpr (NCU > 1 && COUT == 3) {
        if (fwdval) fwdval[r * c.S + k] = make_float4(rad[0], rad[1], rad[2], v);   // (for region_bwd_src_kernel)
      }
      const float sigma = post_activate(g.post_act, v);
      const float dl = last ? kInfinity : (z_next - z);
      const float delta = dl * ray.dnorm;
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
    const size_t pi = rec.z;
    bs.part[2 * pi] = make_float4(T, csum[0], csum[1], csum[2]);
    bs.part[2 * pi + 1] = make_float4(asum, dsum, 0.0f, 0.0f);
  }
}

// ---- pass 4: fold -----------------------------------------------------------------------------------------------------------
// One thread per (ray, depth segment) lane; the nseg lanes of a ray are consecutive threads of one block and meet in LDS.
// Compositing is a Horner scheme read back to front: with (T_j, P_j) the transmittance and partial sums of segment j,
//   U_j = P_j + T_j * U_{j+1}            (sums of everything from segment j on, RELATIVE to the transmittance in front of j)
// -- only products and sums of same-signed terms, small (far) contributions first.  The backward needs exactly this suffix
// (sum_{i >= segment} dL/dw_i w_i = T * <g, U>); the r02 formulation took it as (whole ray) - (prefix), which loses the
// samples deep inside a dense medium to cancellation.  U of a ray's first segment is its output.
//   1. every lane loads its <= 16 segment partials ONCE (predicated, all loads in flight), Horner over them with
//      U_end = 0 -> the lane's own (T_lane, U_lane) into LDS;
//   2. back to front over the ray's later lanes (LDS): U behind the lane; front to back over the earlier ones: T in front;
//   3. per segment: T in front of it (running product) and U_j = U_j(own) + (prod_{i >= j} T_i) * U_behind  -> `state`.
template <int COUT>
__global__ __launch_bounds__(256) void region_fold_kernel(DevCfg c, BinScratch bs, float* __restrict__ colour,
                                                          float* __restrict__ depth, float* __restrict__ acc,
                                                          float* __restrict__ disparity, const int rays_per_block) {
  __shared__ float lds[256 * 6];     // per lane: T_lane, U_lane c0 c1 c2 a d
  const int nseg = num_segments(c.S, c.seg_len);
  const long long nlanes = c.R * nseg;
  const int lr = threadIdx.x / nseg, s = threadIdx.x - lr * nseg;          // ray of the block, depth segment
  const long long r = (long long)blockIdx.x * rays_per_block + lr;
  const bool live = lr < rays_per_block && r < c.R;
  const long long lane = live ? lane_of(r, s, c.R) : 0;
  const unsigned n = live ? bs.lane_n[lane] : 0u;
  float4 pa[kSlotsPerLane];
  float2 pb[kSlotsPerLane];
  // slots no lane of the wave uses (mean use: 4 of 16) are skipped with a wave-uniform test: they are the identity of the
  // fold (T = 1, P = 0), so every loop below may pass over them
  unsigned long long used[kSlotsPerLane];
#pragma unroll
  for (int j = 0; j < kSlotsPerLane; ++j) {
    pa[j] = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
    pb[j] = make_float2(0.0f, 0.0f);
    used[j] = __ballot((unsigned)j < n);
    if ((unsigned)j < n) {
      const size_t sl = slot_of((size_t)lane, (unsigned)j, nlanes);
      pa[j] = bs.part[2 * sl];
      const float4 t = bs.part[2 * sl + 1];
      pb[j] = make_float2(t.x, t.y);
    }
  }
  // own Horner, back to front (unused slots are the identity: T = 1, P = 0)
  float Tl = 1.0f, U[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int j = kSlotsPerLane - 1; j >= 0; --j) {
    if (used[j] == 0ull) continue;
    U[0] = fmaf(pa[j].x, U[0], pa[j].y);
    U[1] = fmaf(pa[j].x, U[1], pa[j].z);
    U[2] = fmaf(pa[j].x, U[2], pa[j].w);
    U[3] = fmaf(pa[j].x, U[3], pb[j].x);
    U[4] = fmaf(pa[j].x, U[4], pb[j].y);
    Tl = Tl * pa[j].x;
  }
  float* me = lds + threadIdx.x * 6;
  me[0] = Tl;
#pragma unroll
  for (int q = 0; q < 5; ++q) me[1 + q] = U[q];
  __syncthreads();
  if (!live) return;
  // behind the lane: Horner over the ray's later lanes, back to front
  float B[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  for (int q = nseg - 1; q > s; --q) {
    const float* o = lds + (threadIdx.x - s + q) * 6;
#pragma unroll
    for (int t = 0; t < 5; ++t) B[t] = fmaf(o[0], B[t], o[1 + t]);
  }
  // in front of the lane: product of the earlier lanes' transmittances
  float T = 1.0f;
  for (int q = 0; q < s; ++q) T = T * lds[(threadIdx.x - s + q) * 6];
  // per segment: state = (T in front, U from the segment on); U_j = own_j + (prod_{i >= j} T_i) * B, again back to front
  float Tfront[kSlotsPerLane];
  {
    float t = T;
#pragma unroll
    for (int j = 0; j < kSlotsPerLane; ++j) { Tfront[j] = t; if (used[j] != 0ull) t = t * pa[j].x; }
  }
  float V[5] = {B[0], B[1], B[2], B[3], B[4]};
#pragma unroll
  for (int j = kSlotsPerLane - 1; j >= 0; --j) {
    if (used[j] == 0ull) continue;
    V[0] = fmaf(pa[j].x, V[0], pa[j].y);
    V[1] = fmaf(pa[j].x, V[1], pa[j].z);
    V[2] = fmaf(pa[j].x, V[2], pa[j].w);
    V[3] = fmaf(pa[j].x, V[3], pb[j].x);
    V[4] = fmaf(pa[j].x, V[4], pb[j].y);
    if ((unsigned)j < n) {
      const size_t sl = slot_of((size_t)lane, (unsigned)j, nlanes);
      bs.state[2 * sl] = make_float4(Tfront[j], V[0], V[1], V[2]);
      bs.state[2 * sl + 1] = make_float4(V[3], V[4], 0.0f, 0.0f);
    }
  }
  if (s != 0) return;
  // the ray's first lane holds the whole ray: V = (csum, asum, dsum)   (accumulate.py:77-88)
  const float asum = V[3], dsum = V[4];
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) {
    float col = V[ch];
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

// ---- pass 5: backward, one block per region ------------------------------------------------------------------------------------
template <int COUT, int NCM, bool BANKED, bool TEXLDS>
__global__ __launch_bounds__(VOXE_REGION_BLOCK, TEXLDS ? 1 : 4) void region_bwd_kernel(
    DevGrid g, DevCfg c, const float* __restrict__ packed, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
    const float* __restrict__ jitter, const float* __restrict__ colour, const float* __restrict__ depth,
    const float* __restrict__ acc, const float* __restrict__ d_colour, const float* __restrict__ d_depth,
    const float* __restrict__ d_acc, float* __restrict__ gpacked, const int want_d, const int want_f, BinScratch bs,
    const int nreg) {
  constexpr int C = COUT + 1, CM = COUT * NCM + 1;
  constexpr bool kTexLds = TEXLDS;
  constexpr bool kPcb = BANKED && C == 4;
  constexpr int kWinDoubles = kPcb ? kPcbDoubles : C * kRPlane;
  __shared__ float tex[kTexLds ? kRWin * C : 1];
  __shared__ double win[kWinDoubles];
  const int tid = threadIdx.x;
  const unsigned region = min(blockIdx.x, (unsigned)nreg);     // blocks nreg .. nreg + kGenericBlocks - 1: the generic bin
  const unsigned first = bs.start[region * kLenClasses];
  const unsigned n = bs.start[(region + 1) * kLenClasses] - first;
  if (n == 0) return;                       // block-uniform
  const RegionBlock rb = region_block(g, blockIdx.x, nreg);
  if (!rb.generic) {
    if constexpr (kTexLds) load_window<COUT, NCM>(g, packed, tex, rb.ox, rb.oy, rb.oz, tid);
    for (int i = tid; i < kWinDoubles; i += VOXE_REGION_BLOCK) win[i] = 0.0;
  }
  // lane constants of the parity-class deposit
  const int lane = tid & 63;
  const int hx = (lane >> 1) & 1, hy = (lane >> 2) & 1, hz = (lane >> 3) & 1, crot = (lane & 1) | ((lane >> 3) & 2);
  __shared__ float2 strat[VOXE_REGION_STRATA ? VOXE_REGION_STRATA : 1];
  const BlockStrata bst = build_block_strata(strat, VOXE_REGION_STRATA, c, 0, c.S, tid, VOXE_REGION_BLOCK);
  __syncthreads();
  const float basis0[1] = {kC0};
  const bool white = c.white && !c.attn;
  const unsigned i_begin = rb.generic ? (blockIdx.x - (unsigned)nreg) * VOXE_REGION_BLOCK + tid : tid;
  const unsigned i_step = rb.generic ? kGenericBlocks * VOXE_REGION_BLOCK : VOXE_REGION_BLOCK;
  for (unsigned i = i_begin; i < n; i += i_step) {
    const uint4 rec = bs.sorted[first + i];
    const long long r = rec.x;
    const int k0 = (int)(rec.y & 0xFFFFu), k1 = (int)(rec.y >> 16);
    SegRay ray;
    ray.init(g, c, r, rays_o, rays_d, jitter);
    // transmittance in front of the segment and the suffix sums from it on, relative to it (region_fold_kernel); the
    // per-ray constants of the backward (render_bwd_kernel)
    const size_t pi = rec.z;
    const float4 sa = bs.state[2 * pi], sb = bs.state[2 * pi + 1];
    float T = sa.x;
    const float suf_c[3] = {sa.y, sa.z, sa.w};
    const float suf_a = sb.x, suf_d = sb.y;
    float gc[COUT], gsum = 0.0f;
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) { gc[ch] = d_colour[r * COUT + ch]; gsum += gc[ch]; }
    const float gdep = d_depth ? d_depth[r] : 0.0f;
    const float gacc = d_acc ? d_acc[r] : 0.0f;
    // suffix0 = sum_{i >= first sample of the segment} dL/dw_i w_i = T * <upstream gradient, U>
    float suffix0 = gdep * suf_d + gacc * suf_a;
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) suffix0 += gc[ch] * suf_c[ch];
    if (white) suffix0 -= gsum * suf_a;
    suffix0 *= T;
    float run = 0.0f;   // sum of dL/dw_i w_i over the samples of this segment up to and including the current one

    float z_next = bst.z(ray.dg, k0);
    for (int k = k0; k <= k1; ++k) {
      const float z = z_next;
      const bool last = (k == c.S - 1);
      if (!last) z_next = bst.z(ray.dg, k + 1);
      float p[3];
      ray.point(z, p);
      Footprint fp;
      footprint(g, p, fp);
      if (!fp.inside) continue;
      Cell cell;
      make_cell_fast(g, fp, cell);
      float v, rad[COUT];
      int idx0 = 0;
      if (rb.generic) {
        gather<COUT, NCM, 1>(g, packed, cell, basis0, v, rad);
      } else {
        const int lx = cell.i[0] - rb.ox, ly = cell.i[1] - rb.oy, lz = cell.i[2] - rb.oz;
        if ((unsigned)lx >= (unsigned)kRBX || (unsigned)ly >= (unsigned)kRBY || (unsigned)lz >= (unsigned)kRBZ) continue;
        idx0 = (lx * kRWY + ly) * kRWZ + lz;
        if constexpr (kTexLds) gather_lds<COUT>(tex, idx0, cell, v, rad);
        else gather<COUT, NCM, 1>(g, packed, cell, basis0, v, rad);
      }
      float sigma, dpost;
      post_activate_vg(g.post_act, v, sigma, dpost);
      const float dl = last ? kInfinity : (z_next - z);
      const float delta = dl * ray.dnorm;
      const float e = fast_exp(-(sigma * delta));
      const float alpha = 1.0f - e;
      const float om = 1.0f - alpha;
      const float wk = alpha * T;
      float col[COUT], dldw = fmaf(gdep, z, gacc);
#pragma unroll
      for (int ch = 0; ch < COUT; ++ch) { col[ch] = sigmoidf(rad[ch]); dldw = fmaf(gc[ch], col[ch], dldw); }
      if (white) dldw -= gsum;
      run = fmaf(dldw, wk, run);
      const float suffix = last ? 0.0f : (suffix0 - run);
      const float tail = (om > 0.0f) ? suffix * fast_rcp(om) : 0.0f;
      const float dsig = (delta * e) * fmaf(T, dldw, -tail);
      float gch[C];
      bool any = false;
#pragma unroll
      for (int ch = 0; ch < COUT; ++ch) {
        gch[ch] = want_f ? ((wk * gc[ch]) * (col[ch] * (1.0f - col[ch]))) * kC0 : 0.0f;
        any = any || (gch[ch] != 0.0f);
      }
      gch[COUT] = want_d ? dsig * dpost : 0.0f;
      any = any || (gch[COUT] != 0.0f);
      T = T * om;
      if (!any) continue;
#if defined(VOXE_REGION_EXP) && (VOXE_REGION_EXP & 4)
      if (gch[0] != 12345.0f) continue;   // timing experiment: no deposit at all
#endif
      if (rb.generic) {
        const CellAddr ad = cell_addr(g, cell);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float w = (cell.w[0][j & 1] * cell.w[1][(j >> 1) & 1]) * cell.w[2][j >> 2];
          if (w == 0.0f) continue;
          float* __restrict__ texel =
              gpacked + (long long)(ad.base + (j & 1) * ad.sx + ((j >> 1) & 1) * ad.sy + (j >> 2) * ad.sz) * CM;
#pragma unroll
          for (int ch = 0; ch < C; ++ch)
            if (gch[ch] != 0.0f) atomicAdd(texel + (ch == COUT ? CM - 1 : ch * NCM), gch[ch] * w);
        }
      } else if constexpr (kPcb) {
        // per axis: the corner whose window coordinate has parity h goes where the instruction's axis bit is 0
        const int lx = cell.i[0] - rb.ox, ly = cell.i[1] - rb.oy, lz = cell.i[2] - rb.oz;
        auto split = [](int l, int h, float w0, float w1, int stride, int lo, int& t0, int& t1, float& x0, float& x1) {
          const int d0 = (l ^ h) & 1;
          const int c0 = l + d0, c1 = l + 1 - d0;
          t0 = (c0 >> 1) * stride + (h << lo);
          t1 = (c1 >> 1) * stride + ((1 - h) << lo);
          x0 = d0 ? w1 : w0;
          x1 = d0 ? w0 : w1;
        };
        int tx[2], ty[2], tz[2];
        float wx[2], wy[2], wz[2];
        split(lx, hx, cell.w[0][0], cell.w[0][1], 32 * kPY * kPZ, 2, tx[0], tx[1], wx[0], wx[1]);
        split(ly, hy, cell.w[1][0], cell.w[1][1], 32 * kPZ, 1, ty[0], ty[1], wy[0], wy[1]);
        split(lz, hz, cell.w[2][0], cell.w[2][1], 32, 0, tz[0], tz[1], wz[0], wz[1]);
        const bool c1 = crot & 1, c2 = crot & 2;
        const float q0 = c1 ? gch[1] : gch[0], q1 = c1 ? gch[2] : gch[1], q2 = c1 ? gch[3] : gch[2], q3 = c1 ? gch[0] : gch[3];
        const float gr[4] = {c2 ? q2 : q0, c2 ? q3 : q1, c2 ? q0 : q2, c2 ? q1 : q3};   // gr[j] = gch[(j + crot) & 3]
        int choff[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) choff[j] = ((j + crot) & 3) << 3;
        const int txy[4] = {tx[0] + ty[0], tx[1] + ty[0], tx[0] + ty[1], tx[1] + ty[1]};
        const float wxy[4] = {wx[0] * wy[0], wx[1] * wy[0], wx[0] * wy[1], wx[1] * wy[1]};
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
          const float w = wxy[cc & 3] * wz[cc >> 2];
          const int idx = txy[cc & 3] + tz[cc >> 2];
          // (a frozen tensor's channels deposit exact zeros: skipped by the flush)
#pragma unroll
          for (int ch = 0; ch < 4; ++ch)
#if defined(VOXE_REGION_EXP) && (VOXE_REGION_EXP & 2)
            if (gr[ch] * w == 12345.0f && idx == 77)   // timing experiment: products and addresses formed, no LDS add
#endif
            __hip_atomic_fetch_add(&win[choff[ch] + idx], (double)(gr[ch] * w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float w = (cell.w[0][j & 1] * cell.w[1][(j >> 1) & 1]) * cell.w[2][j >> 2];
          const int idx = idx0 + (j & 1) * (kRWY * kRWZ) + ((j >> 1) & 1) * kRWZ + (j >> 2);
#pragma unroll
          for (int ch = 0; ch < C; ++ch) {
            if ((ch < COUT) ? want_f : want_d)
              __hip_atomic_fetch_add(&win[ch * kRPlane + idx], (double)(gch[ch] * w), __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
  }
  if (rb.generic) return;
  __syncthreads();
  // flush: lanes = (voxel, channel), channel fastest -> 16-byte dense global atomics along the z-runs of the window
  for (int e = tid; e < kRWin * C; e += VOXE_REGION_BLOCK) {
    const int vl = e / C, ch = e - vl * C;
    const double val = kPcb ? win[pcb_index(vl / (kRWY * kRWZ), (vl / kRWZ) % kRWY, vl % kRWZ, ch)] : win[ch * kRPlane + vl];
    if (val == 0.0) continue;
    const int x = rb.ox + vl / (kRWY * kRWZ), y = rb.oy + (vl / kRWZ) % kRWY, zz = rb.oz + vl % kRWZ;
    if (x >= g.X || y >= g.Y || zz >= g.Z) continue;   // (weight-0 corners of size-1 axes / beyond the grid's far faces)
    const long long vox = ((long long)x * g.Y + y) * g.Z + zz;
    atomicAdd(gpacked + vox * CM + (ch == COUT ? CM - 1 : ch * NCM), (float)val);
  }
}

// ---- pass 5', view-dependent grids (SH degree 1 / 2): two-phase backward ---------------------------------------------------------
// d rad_c / d coef_cj = basis_j(ray) is a per-ray constant, so a sample has only FOUR gradient sources (d rad_0..2, d v),
// exactly like SH-0, and every gradient channel is one of them times a constant.
//   phase 1  region_bwd_src_kernel   one block per region, the region's WHOLE texels in LDS: the march of
//            region_bwd_kernel up to the 4 sources of every sample -> src[ray * S + k] (16 bytes per sample);
//   phase 2  region_bwd_dep_kernel   one block per (region, group of 7 gradient channels): footprints only (no gather),
//            sources x basis -> the same 9x9x9 double window as SH-0, one dense flush per (region, group).
template <int NCM, int NCU>
__global__ __launch_bounds__(VOXE_REGION_BLOCK) void region_bwd_src_kernel(
    DevGrid g, DevCfg c, const float* __restrict__ packed, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
    const float* __restrict__ jitter, const float* __restrict__ d_colour, const float* __restrict__ d_depth,
    const float* __restrict__ d_acc, const int want_d, const int want_f, BinScratch bs, const int nreg,
    float4* __restrict__ src, const int stage, const float4* __restrict__ fwdval) {
  constexpr int COUT = 3;
  extern __shared__ float4 tex_dyn[];
  float* const tex = reinterpret_cast<float*>(tex_dyn);
  const int tid = threadIdx.x;
  const unsigned region = min(blockIdx.x, (unsigned)nreg);
  const unsigned first = bs.start[region * kLenClasses];
  const unsigned n = bs.start[(region + 1) * kLenClasses] - first;
  if (n == 0) return;
  const RegionBlock rb = region_block(g, blockIdx.x, nreg);
  // fwdval: the forward of the same rays left (rad, v) of every sample -- no texels needed at all
  const bool from_global = rb.generic || stage == 0 || fwdval != nullptr;   // (see region_fwd_kernel)
  if (!from_global) load_window_full<COUT * NCM + 1>(g, packed, tex, rb.ox, rb.oy, rb.oz, tid);
  __shared__ float2 strat[VOXE_REGION_STRATA ? VOXE_REGION_STRATA : 1];
  const BlockStrata bst = build_block_strata(strat, VOXE_REGION_STRATA, c, 0, c.S, tid, VOXE_REGION_BLOCK);
  __syncthreads();
  const bool white = c.white && !c.attn;
  const unsigned i_begin = rb.generic ? (blockIdx.x - (unsigned)nreg) * VOXE_REGION_BLOCK + tid : tid;
  const unsigned i_step = rb.generic ? kGenericBlocks * VOXE_REGION_BLOCK : VOXE_REGION_BLOCK;
  for (unsigned i = i_begin; i < n; i += i_step) {
    const uint4 rec = bs.sorted[first + i];
    const long long r = rec.x;
    const int k0 = (int)(rec.y & 0xFFFFu), k1 = (int)(rec.y >> 16);
    SegRay ray;
    ray.init(g, c, r, rays_o, rays_d, jitter);
    float basis[NCU];
    ray_basis<NCU>(ray.d, ray.dnorm, basis);
    const size_t pi = rec.z;
    const float4 sa = bs.state[2 * pi], sb = bs.state[2 * pi + 1];
    float T = sa.x;
    float gc[COUT], gsum = 0.0f;
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) { gc[ch] = d_colour[r * COUT + ch]; gsum += gc[ch]; }
    const float gdep = d_depth ? d_depth[r] : 0.0f;
    const float gacc = d_acc ? d_acc[r] : 0.0f;
    float suffix0 = gdep * sb.y + gacc * sb.x + gc[0] * sa.y + gc[1] * sa.z + gc[2] * sa.w;
    if (white) suffix0 -= gsum * sb.x;
    suffix0 *= T;
    float run = 0.0f;
    float z_next = bst.z(ray.dg, k0);
    for (int k = k0; k <= k1; ++k) {
      const float z = z_next;
      const bool last = (k == c.S - 1);
      if (!last) z_next = bst.z(ray.dg, k + 1);
      float p[3];
      ray.point(z, p);
      Footprint fp;
      footprint(g, p, fp);
      if (!fp.inside) continue;
      Cell cell;
      make_cell_fast(g, fp, cell);
      float v, rad[COUT];
      if (from_global) {
        if (!rb.generic) {
          const int lx = cell.i[0] - rb.ox, ly = cell.i[1] - rb.oy, lz = cell.i[2] - rb.oz;
          if ((unsigned)lx >= (unsigned)kRBX || (unsigned)ly >= (unsigned)kRBY || (unsigned)lz >= (unsigned)kRBZ) continue;
        }
        if (fwdval) {
          const float4 f4 = fwdval[r * c.S + k];
          rad[0] = f4.x; rad[1] = f4.y; rad[2] = f4.z; v = f4.w;
        } else {
          gather<COUT, NCM, NCU>(g, packed, cell, basis, v, rad);
        }
      } else {
        const int lx = cell.i[0] - rb.ox, ly = cell.i[1] - rb.oy, lz = cell.i[2] - rb.oz;
        if ((unsigned)lx >= (unsigned)kRBX || (unsigned)ly >= (unsigned)kRBY || (unsigned)lz >= (unsigned)kRBZ) continue;
        gather_lds_sh<COUT, NCM, NCU>(tex, (lx * kRWY + ly) * kRWZ + lz, cell, basis, v, rad);
      }
      float sigma, dpost;
      post_activate_vg(g.post_act, v, sigma, dpost);
      const float dl = last ? kInfinity : (z_next - z);
      const float delta = dl * ray.dnorm;
      const float e = fast_exp(-(sigma * delta));
      const float alpha = 1.0f - e;
      const float om = 1.0f - alpha;
      const float wk = alpha * T;
      float col[COUT], dldw = fmaf(gdep, z, gacc);
#pragma unroll
      for (int ch = 0; ch < COUT; ++ch) { col[ch] = sigmoidf(rad[ch]); dldw = fmaf(gc[ch], col[ch], dldw); }
      if (white) dldw -= gsum;
      run = fmaf(dldw, wk, run);
      const float suffix = last ? 0.0f : (suffix0 - run);
      const float tail = (om > 0.0f) ? suffix * fast_rcp(om) : 0.0f;
      const float dsig = (delta * e) * fmaf(T, dldw, -tail);
      float4 o4;
      o4.x = want_f ? (wk * gc[0]) * (col[0] * (1.0f - col[0])) : 0.0f;
      o4.y = want_f ? (wk * gc[1]) * (col[1] * (1.0f - col[1])) : 0.0f;
      o4.z = want_f ? (wk * gc[2]) * (col[2] * (1.0f - col[2])) : 0.0f;
      o4.w = want_d ? dsig * dpost : 0.0f;
      src[r * c.S + k] = o4;
      T = T * om;
    }
  }
}

template <int NCM, int NCU>
__global__ __launch_bounds__(VOXE_REGION_BLOCK) void region_bwd_dep_kernel(
    DevGrid g, DevCfg c, const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ jitter,
    float* __restrict__ gpacked, BinScratch bs, const int nreg, const float4* __restrict__ src, const int grp_begin) {
  constexpr int COUT = 3, CM = COUT * NCM + 1, NG = COUT * NCU + 1, WC = VOXE_REGION_SH_WC;
  __shared__ double win[WC * kRPlane];
  const int tid = threadIdx.x;
  const unsigned region = min(blockIdx.x, (unsigned)nreg);
  const unsigned first = bs.start[region * kLenClasses];
  const unsigned n = bs.start[(region + 1) * kLenClasses] - first;
  if (n == 0) return;
  const RegionBlock rb = region_block(g, blockIdx.x, nreg);
  const int grp = grp_begin + (int)blockIdx.y;
  // window channel s <-> gradient channel q = grp * 4 + s: coefficient j of colour ch (texel channel ch * NCM + j, factor
  // basis_j of the ray) or, last, the density (texel channel CM - 1, factor 1); -1: unused slot of the last group
  int chsel[WC], jsel[WC], memch[WC];
#pragma unroll
  for (int sidx = 0; sidx < WC; ++sidx) {
    const int q = grp * WC + sidx;
    if (q >= NG) { chsel[sidx] = -1; jsel[sidx] = 0; memch[sidx] = 0; }
    else if (q == NG - 1) { chsel[sidx] = COUT; jsel[sidx] = 0; memch[sidx] = CM - 1; }
    else { chsel[sidx] = q / NCU; jsel[sidx] = q - chsel[sidx] * NCU; memch[sidx] = chsel[sidx] * NCM + jsel[sidx]; }
  }
  if (!rb.generic)
    for (int i = tid; i < WC * kRPlane; i += VOXE_REGION_BLOCK) win[i] = 0.0;
  __shared__ float2 strat[VOXE_REGION_STRATA ? VOXE_REGION_STRATA : 1];
  const BlockStrata bst = build_block_strata(strat, VOXE_REGION_STRATA, c, 0, c.S, tid, VOXE_REGION_BLOCK);
  __syncthreads();
  const unsigned i_begin = rb.generic ? (blockIdx.x - (unsigned)nreg) * VOXE_REGION_BLOCK + tid : tid;
  const unsigned i_step = rb.generic ? kGenericBlocks * VOXE_REGION_BLOCK : VOXE_REGION_BLOCK;
  for (unsigned i = i_begin; i < n; i += i_step) {
    const uint4 rec = bs.sorted[first + i];
    const long long r = rec.x;
    const int k0 = (int)(rec.y & 0xFFFFu), k1 = (int)(rec.y >> 16);
    SegRay ray;
    ray.init(g, c, r, rays_o, rays_d, jitter);
    float basis[NCU];
    ray_basis<NCU>(ray.d, ray.dnorm, basis);
    float mult[WC];
#pragma unroll
    for (int sidx = 0; sidx < WC; ++sidx) {
      float b = basis[0];
#pragma unroll
      for (int t = 1; t < NCU; ++t) b = (jsel[sidx] == t) ? basis[t] : b;
      mult[sidx] = chsel[sidx] < 0 ? 0.0f : (chsel[sidx] == COUT ? 1.0f : b);
    }
    for (int k = k0; k <= k1; ++k) {
      const float z = bst.z(ray.dg, k);
      float p[3];
      ray.point(z, p);
      Footprint fp;
      footprint(g, p, fp);
      if (!fp.inside) continue;
      Cell cell;
      make_cell_fast(g, fp, cell);
      int idx0 = 0;
      if (!rb.generic) {
        const int lx = cell.i[0] - rb.ox, ly = cell.i[1] - rb.oy, lz = cell.i[2] - rb.oz;
        if ((unsigned)lx >= (unsigned)kRBX || (unsigned)ly >= (unsigned)kRBY || (unsigned)lz >= (unsigned)kRBZ) continue;
        idx0 = (lx * kRWY + ly) * kRWZ + lz;
      }
      const float4 s4 = src[r * c.S + k];
      float gch[WC];
      bool any = false;
#pragma unroll
      for (int sidx = 0; sidx < WC; ++sidx) {
        const float x = chsel[sidx] == 0 ? s4.x : (chsel[sidx] == 1 ? s4.y : (chsel[sidx] == 2 ? s4.z : s4.w));
        gch[sidx] = x * mult[sidx];
        any = any || (gch[sidx] != 0.0f);
      }
      if (!any) continue;
      if (rb.generic) {
        const CellAddr ad = cell_addr(g, cell);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float w = (cell.w[0][j & 1] * cell.w[1][(j >> 1) & 1]) * cell.w[2][j >> 2];
          if (w == 0.0f) continue;
          float* __restrict__ texel =
              gpacked + (long long)(ad.base + (j & 1) * ad.sx + ((j >> 1) & 1) * ad.sy + (j >> 2) * ad.sz) * CM;
#pragma unroll
          for (int sidx = 0; sidx < WC; ++sidx)
            if (gch[sidx] != 0.0f) atomicAdd(texel + memch[sidx], gch[sidx] * w);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float w = (cell.w[0][j & 1] * cell.w[1][(j >> 1) & 1]) * cell.w[2][j >> 2];
          const int idx = idx0 + (j & 1) * (kRWY * kRWZ) + ((j >> 1) & 1) * kRWZ + (j >> 2);
#pragma unroll
          for (int sidx = 0; sidx < WC; ++sidx)
            if (chsel[sidx] >= 0)   // (block-uniform: the last group of 13 / 28 channels holds 1 / 4 used slots)
              __hip_atomic_fetch_add(&win[sidx * kRPlane + idx], (double)(gch[sidx] * w), __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
  }
  if (rb.generic) return;
  __syncthreads();
  for (int e = tid; e < kRWin * WC; e += VOXE_REGION_BLOCK) {
    const int vl = e / WC, sidx = e - vl * WC;
    const double val = win[sidx * kRPlane + vl];
    if (val == 0.0) continue;
    const int x = rb.ox + vl / (kRWY * kRWZ), y = rb.oy + (vl / kRWZ) % kRWY, zz = rb.oz + vl % kRWZ;
    if (x >= g.X || y >= g.Y || zz >= g.Z) continue;
    const long long vox = ((long long)x * g.Y + y) * g.Z + zz;
    int mc = memch[0];
#pragma unroll
    for (int t = 1; t < WC; ++t) mc = (sidx == t) ? memch[t] : mc;
    atomicAdd(gpacked + vox * CM + mc, (float)val);
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
bool region_bwd_supported(const DevGrid& g, const HostCfg& c, int deg, int diffuse, bool tiled) {
  const long long min_rays = disp_region_min_rays(c.disp);   // (< 0: the route is switched off)
  if (min_rays < 0 || c.R < min_rays || c.term_eps > 0.0f) return false;
  // SH-0 / diffuse / attention: one channel group.  SH degree 1 - 3: whole texels in LDS + two-phase backward (degree 3:
  // 729 texels x 52 floats = 151.6 KB of the CU's 160 KB -- one block per CU, still 2.4x the generic scatter; r04)
  if (VOXE_REGION_SH3 == 0 && !(c.attn || deg <= 2 || diffuse)) return false;
  if (num_segments(c.S, c.seg_len) > 256) return false;       // (the lanes of a ray meet in one block's LDS: region_fold_kernel)
  if (c.R > (1ll << 19) || c.S >= 65536) return false;         // segment records hold 32-bit rays / 16-bit sample indices;
                                                               // the tables take ~12 KB per ray (S = 256): capped at 512 k rays (6 GB) per launch
  if (!tiled) return true;                                     // unordered rays, images below the tile threshold
  // image-ordered launches: only when the pixels are clearly more than a voxel apart (nothing to combine inside a wave:
  // 100x100 cameras on a 160^3 grid).  The pixel spacing is not known on the host; for a camera that frames the volume
  // it is ~ grid side / image width.
  const float ratio = disp_region_image_ratio(c.disp);
  const int side = g.X > g.Y ? (g.X > g.Z ? g.X : g.Z) : (g.Y > g.Z ? g.Y : g.Z);
  return (float)side >= ratio * (float)c.image_width;
}

static inline size_t up256(size_t x) { return (x + 255) / 256 * 256; }
struct RegionLayout { size_t slot_region, slot_pos, slot_seg, sorted, part, state, lane_n, counters, src, fwdval, blockhist, blockoff, total; long long nslots, nlanes; int nreg, seg_blocks; };
// blocks of region_seg_lds_kernel for a launch of `nlanes` (ray, depth segment) lanes (a multiple of 16: region_colscan_kernel): one
// unit per wave while that takes at most one block per CU, two from there on; 0: the region table does not fit in LDS
constexpr size_t kSegLdsMaxBytes = 128 * 1024;
static int seg_lds_blocks(long long nlanes, int nreg) {
  if ((size_t)(nreg + 1) * sizeof(unsigned long long) > kSegLdsMaxBytes) return 0;
  auto up16 = [](long long x) { return (x + 15) / 16 * 16; };
  long long nb = up16((nlanes + 1023) / 1024);             // one unit (64 lanes) per wave ...
  if (nb > 256) { nb = up16((nlanes + 2047) / 2048); if (nb < 256) nb = 256; }   // ... two once that takes more than a block per CU
  return (int)(nb < 16 ? 16 : (nb > 1024 ? 1024 : nb));
}
static RegionLayout region_layout(int X, int Y, int Z, long long R, int S, bool full_sh = false) {
  RegionLayout l;
  const int nseg = num_segments(S, seg_len_for(R));
  l.nslots = R * nseg * kSlotsPerLane;
  l.nlanes = R * nseg;
  l.nreg = regions_along(X, kRBX) * regions_along(Y, kRBY) * regions_along(Z, kRBZ);
  size_t off = 0;
  l.slot_region = off; off += up256((size_t)l.nslots * sizeof(unsigned));
  l.slot_pos = off; off += up256((size_t)l.nslots * sizeof(unsigned));
  l.slot_seg = off; off += up256((size_t)l.nslots * sizeof(uint2));
  l.sorted = off; off += up256((size_t)l.nslots * sizeof(uint4));
  l.part = off; off += up256((size_t)l.nslots * 2 * sizeof(float4));
  l.state = off; off += up256((size_t)l.nslots * 2 * sizeof(float4));
  l.lane_n = off; off += up256((size_t)l.nlanes * sizeof(unsigned));
  l.counters = off; off += 2 * up256(((size_t)(l.nreg + 1) * kLenClasses + 4) * sizeof(unsigned));   // count | start
  // view-dependent grids: the 4 gradient sources of every sample between the two phases of the backward
  l.src = off; off += full_sh ? up256((size_t)R * (size_t)S * sizeof(float4)) : 0;
  // ... and the forward's (rad_0..2, v) of every sample, same indexing: the source pass reads them instead of gathering again (r04)
  l.fwdval = off; off += full_sh ? up256((size_t)R * (size_t)S * sizeof(float4)) : 0;
  // r05: block tables of the LDS-ranked segment pass
  l.seg_blocks = seg_lds_blocks(l.nlanes, l.nreg);
  l.blockhist = off; off += up256((size_t)l.seg_blocks * (size_t)(l.nreg + 1) * sizeof(unsigned long long));
  l.blockoff = off; off += up256((size_t)l.seg_blocks * (size_t)(l.nreg + 1) * sizeof(uint4));
  l.total = off;
  return l;
}
size_t region_scratch_bytes(int X, int Y, int Z, long long R, int S, bool full_sh) {
  if (R <= 0 || S <= 0) return 0;
  return region_layout(X, Y, Z, R, S, full_sh).total;
}
// The segment TABLES (everything the binning passes write: slots, sorted list, counters, block tables) may live in another
// buffer than the per-segment partials / states: voxe_recon_prefetch bins the NEXT iteration's batch into a second set of tables
// while this iteration's backward still reads the first (tl_region_bins, set by voxe_recon_step around its render calls).
thread_local const RegionBins* tl_region_bins = nullptr;
static BinScratch bin_scratch(const RegionLayout& l, void* scratch, void* tables = nullptr) {
  char* base = (char*)scratch;
  char* tb = tables ? (char*)tables : ((tl_region_bins && tl_region_bins->tables) ? (char*)tl_region_bins->tables : base);
  BinScratch bs;
  bs.slot_region = (unsigned*)(tb + l.slot_region);
  bs.slot_pos = (unsigned*)(tb + l.slot_pos);
  bs.slot_seg = (uint2*)(tb + l.slot_seg);
  bs.sorted = (uint4*)(tb + l.sorted);
  bs.part = (float4*)(base + l.part);
  bs.state = (float4*)(base + l.state);
  bs.lane_n = (unsigned*)(tb + l.lane_n);
  bs.count = (unsigned*)(tb + l.counters);
  bs.start = bs.count + up256(((size_t)(l.nreg + 1) * kLenClasses + 4) * sizeof(unsigned)) / sizeof(unsigned);
  bs.blockhist = l.seg_blocks ? (unsigned long long*)(tb + l.blockhist) : nullptr;
  bs.blockoff = (uint4*)(tb + l.blockoff);
  return bs;
}

void region_debug_layout(int X, int Y, int Z, long long R, int S, long long out[16]) {
  const RegionLayout l = region_layout(X, Y, Z, R, S);   // (the tables come first: same offsets with or without the source buffer)
  const size_t start_off = l.counters + up256(((size_t)(l.nreg + 1) * kLenClasses + 4) * sizeof(unsigned));
  const long long v[16] = {(long long)l.slot_region, (long long)l.slot_pos, (long long)l.slot_seg, (long long)l.sorted,
                           (long long)l.lane_n, (long long)l.counters, (long long)start_off, l.nslots, l.nlanes, l.nreg,
                           kSlotsPerLane, kRBX, kRBY, kRBZ, kLenClasses, VOXE_REGION_CHUNK};
  for (int i = 0; i < 16; ++i) out[i] = v[i];
}

// dynamic LDS of the kernels that stage whole texels (above 64 KB the function attribute has to allow it)
template <typename K>
static size_t full_tex_lds(K kernel, int cm) {
  const size_t bytes = (size_t)kRWin * tex_stride(cm) * sizeof(float);
  if (bytes > 64 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return bytes;
}

// the binning passes (index math only: they read the rays and the jitter streams, never the grid): segments -> counting sort by
// (region, length class).  Writes the TABLES of `bs` only.
static void launch_bin_region_passes(const DevGrid& g, const DevCfg& c, const float* rays_o, const float* rays_d, const float* jitter,
                                     const RegionLayout& l, const BinScratch& bs, hipStream_t st, bool lds_ok, int phase = 3) {
  // phase: 1 = clear the counters | 2 = everything behind that | 3 = both
  if (phase & 1) (void)hipMemsetAsync(bs.count, 0, 2 * up256(((size_t)(l.nreg + 1) * kLenClasses + 4) * sizeof(unsigned)), st);
  if (!(phase & 2)) return;
  const int nseg = num_segments(c.S, c.seg_len);
  const int nb = (c.image_width > 0 ? blocks_for_tiles(c.map_mode, (c.image_width + 7) / 8, tile_rows_total(c, 8))
                                    : blocks_for_tiles(c.map_mode, 1, (c.R + 63) / 64)) * nseg;
  // segment ranks: block-local in LDS (r05) when the region table fits and no block can see 65 536 segments of one counter
  // (<= 4 units of 64 lanes per wave x 16 waves x 15 slots); one returning global atomic per segment otherwise
  const int units_per_wave = l.seg_blocks ? (nb + l.seg_blocks * kSegLdsWaves - 1) / (l.seg_blocks * kSegLdsWaves) : 0;
  const bool lds_ranks = l.seg_blocks > 0 && units_per_wave <= 4 && lds_ok;
  if (lds_ranks) {
    const size_t hist_bytes = (size_t)(l.nreg + 1) * sizeof(unsigned long long);
    if (hist_bytes > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)region_seg_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hist_bytes);
    region_seg_lds_kernel<<<l.seg_blocks, kSegLdsThreads, hist_bytes, st>>>(g, c, rays_o, rays_d, jitter, bs, l.nreg, nb, units_per_wave);
    region_colscan_kernel<<<(l.nreg + 1 + kColRegions - 1) / kColRegions, kColRegions * kColChunks, 0, st>>>(bs, l.nreg, l.seg_blocks);
  } else {
    region_seg_kernel<<<nb, 64, 0, st>>>(g, c, rays_o, rays_d, jitter, bs, l.nreg);
  }
  region_scan_kernel<<<1, 1024, 0, st>>>(bs.count, bs.start, (l.nreg + 1) * kLenClasses + 1);
  region_fill_kernel<<<(int)((l.nlanes + 255) / 256), 256, 0, st>>>(bs, l.nlanes, l.nreg, lds_ranks ? 1 : 0);
}
// ... on their own, into the tables at `tables` (a region scratch, or a buffer of the same size): what launch_fwd_region would do
// first for these rays (voxe_recon_prefetch; the table offsets do not depend on the grid's SH degree)
void launch_bin_region(const DevGrid& g, const HostCfg& c, const float* rays_o, const float* rays_d, const float* jitter, void* tables,
                       hipStream_t st, int phase) {
  const RegionLayout l = region_layout(g.X, g.Y, g.Z, c.R, c.S, false);
  const BinScratch bs = bin_scratch(l, tables, tables);
  launch_bin_region_passes(g, c, rays_o, rays_d, jitter, l, bs, st, disp_region_lds_ranks(c.disp), phase);
}

template <int COUT, int NCM, int NCU>
static void launch_fwd_region_t(const DevGrid& g, const DevCfg& c, const FwdArgs& a, void* scratch, hipStream_t st, bool lds_ok) {
  const RegionLayout l = region_layout(g.X, g.Y, g.Z, c.R, c.S, NCU > 1);
  const BinScratch bs = bin_scratch(l, scratch);
  if (!(tl_region_bins && tl_region_bins->prebinned))     // (the tables already hold these rays' segments)
    launch_bin_region_passes(g, c, a.rays_o, a.rays_d, a.jitter, l, bs, st, lds_ok);
  const int nseg = num_segments(c.S, c.seg_len);
  // degree 3: staging the 151.6 KB of a region's texels leaves one block (4 waves) per CU
  const int stage = (NCU > VOXE_REGION_STAGE_FWD_NCU) ? 0 : 1;
  const size_t lds = (NCU > 1 && stage) ? full_tex_lds(region_fwd_kernel<COUT, NCM, NCU>, COUT * NCM + 1) : 0;
  float4* fwdval = (NCU > 1 && a.keep_samples && VOXE_REGION_FWDVAL) ? (float4*)((char*)scratch + l.fwdval) : nullptr;
  region_fwd_kernel<COUT, NCM, NCU><<<l.nreg + kGenericBlocks, VOXE_REGION_BLOCK, lds, st>>>(g, c, a.packed, a.rays_o, a.rays_d, a.jitter, bs, l.nreg, stage, fwdval);
  if (tl_region_bins && tl_region_bins->after_fwd) (void)hipEventRecord(tl_region_bins->after_fwd, st);
  const int rays_per_block = 256 / nseg;        // (nseg <= 256: region_bwd_supported)
  region_fold_kernel<COUT><<<(int)((c.R + rays_per_block - 1) / rays_per_block), 256, 0, st>>>(
      c, bs, a.colour, a.depth, a.acc, a.disparity, rays_per_block);
}

template <int COUT, int NCM, int NCU>
static void launch_bwd_region_t(const DevGrid& g, const DevCfg& c, const BwdArgs& a, void* scratch, hipStream_t st) {
  const RegionLayout l = region_layout(g.X, g.Y, g.Z, c.R, c.S, NCU > 1);
  const BinScratch bs = bin_scratch(l, scratch);
  if constexpr (NCU == 1) {
    // The parity-class banked window (conflict free, +75 VALU per sample, 32 KB instead of 23.5: 3 blocks per CU instead of 4)
    // pays when the regions are FULL: 160 000 unordered rays 0.79 -> 0.51 ms, 80 000 (8 cameras of 100x100) 0.373 -> 0.357,
    // but 32 768 (a reconstruction batch: ~110 segments per region, under two waves) 0.246 -> 0.287 per render
    // (profiles/r04_ab_lds_layout.txt).  VOXE_REGION_PCB = 0 builds without it.
    // texels staged in LDS or gathered from L1 / L2 (VOXE_REGION_BWD_TEX)
    const bool stage = VOXE_REGION_BWD_TEX == 1 || (VOXE_REGION_BWD_TEX == 2 && (NCM > 1 || c.R >= (long long)VOXE_REGION_TEX_MIN_RAYS));
    const bool banked = VOXE_REGION_PCB && COUT == 3 && c.R >= kBankedMinRays;
#define VOXE_RBWD(B, T)                                                                                                          \
    region_bwd_kernel<COUT, NCM, B, T><<<l.nreg + kGenericBlocks, VOXE_REGION_BLOCK, 0, st>>>(                                   \
        g, c, a.packed, a.rays_o, a.rays_d, a.jitter, a.colour, a.depth, a.acc, a.d_colour, a.d_depth, a.d_acc, a.gpacked,       \
        a.want_d ? 1 : 0, a.want_f ? 1 : 0, bs, l.nreg)
    if constexpr (NCM > 1) {      // (wide grids: always staged -- the un-staged instantiations are not built)
      if (banked) VOXE_RBWD(true, true); else VOXE_RBWD(false, true);
    } else {
      if (banked && stage) VOXE_RBWD(true, true);
      else if (banked) VOXE_RBWD(true, false);
      else if (stage) VOXE_RBWD(false, true);
      else VOXE_RBWD(false, false);
    }
#undef VOXE_RBWD
  } else {
    float4* src = (float4*)((char*)scratch + l.src);
    // the forward that filled these tables (voxe_render_fwd on this workspace, or the backward's own re-march) kept its samples
    const float4* fwdval = VOXE_REGION_FWDVAL ? (const float4*)((char*)scratch + l.fwdval) : nullptr;
    const int stage = (NCU > VOXE_REGION_STAGE_BWD_NCU || fwdval) ? 0 : 1;
    const size_t lds = stage ? full_tex_lds(region_bwd_src_kernel<NCM, NCU>, COUT * NCM + 1) : 0;
    region_bwd_src_kernel<NCM, NCU><<<l.nreg + kGenericBlocks, VOXE_REGION_BLOCK, lds, st>>>(
        g, c, a.packed, a.rays_o, a.rays_d, a.jitter, a.d_colour, a.d_depth, a.d_acc, a.want_d ? 1 : 0, a.want_f ? 1 : 0, bs,
        l.nreg, src, stage, fwdval);
    // channel groups of VOXE_REGION_SH_WC: all of them for a feature gradient, only the one holding the density channel (the last) otherwise
    constexpr int NGRP = (COUT * NCU + 1 + VOXE_REGION_SH_WC - 1) / VOXE_REGION_SH_WC;
    const int grp_begin = a.want_f ? 0 : NGRP - 1, ngrp = a.want_f ? NGRP : 1;
    region_bwd_dep_kernel<NCM, NCU><<<dim3((unsigned)(l.nreg + kGenericBlocks), (unsigned)ngrp), VOXE_REGION_BLOCK, 0, st>>>(
        g, c, a.rays_o, a.rays_d, a.jitter, a.gpacked, bs, l.nreg, src, grp_begin);
  }
}

#define VOXE_REGION_DISPATCH(FN, ...)                                                    \
  do {                                                                                   \
    if (c.attn) FN<1, 1, 1>(__VA_ARGS__);                                                \
    else if (deg == 0) FN<3, 1, 1>(__VA_ARGS__);                                         \
    else if (deg == 1) { if (diffuse) FN<3, 4, 1>(__VA_ARGS__); else FN<3, 4, 4>(__VA_ARGS__); }   \
    else if (deg == 2) { if (diffuse) FN<3, 9, 1>(__VA_ARGS__); else FN<3, 9, 9>(__VA_ARGS__); }   \
    else { if (diffuse || VOXE_REGION_SH3 == 0) FN<3, 16, 1>(__VA_ARGS__); else FN<3, 16, 16>(__VA_ARGS__); }   \
  } while (0)

// forward of the space-binned path: fills the segment tables + per-segment states in `scratch` (the backward reuses them
// when the caller says they belong to this call: VoxeRenderCfg::ray_state_valid) and the outputs that are not null
void launch_fwd_region(const DevGrid& g, const HostCfg& c, int deg, int diffuse, const FwdArgs& a, void* scratch,
                       hipStream_t st) {
  VOXE_REGION_DISPATCH(launch_fwd_region_t, g, c, a, scratch, st, disp_region_lds_ranks(c.disp));
}
void launch_bwd_region(const DevGrid& g, const DevCfg& c, int deg, int diffuse, const BwdArgs& a, void* scratch,
                       hipStream_t st) {
  VOXE_REGION_DISPATCH(launch_bwd_region_t, g, c, a, scratch, st);
}

}  // namespace voxe
