// voxe_render_region.hip -- backward render for rays that share no voxels with their NEIGHBOURS IN THE LAUNCH (random
// training batches of the reconstruction loop, modules/trainers.py:288-351; low-resolution images whose pixels are more
// than a voxel apart): the deposit is binned by SPACE instead of by ray.
//
// Why: such rays still meet -- 32768 rays x 144 in-AABB samples x 8 corners = 38 M deposits land on 4 M voxels -- but
// not inside a wave, so the LDS window of render_bwd_tile_kernel has nothing to combine and the line-dense scatter
// (render_bwd_packed_scatter_kernel) stays bound by the ~20 G atomic cache-line requests/s of the memory side
// (profiles/r01_microbench_atomics.md: 3.6 requests per sample).  Here
//
//   1. render_bwd_src_kernel      one lane = one ray (x depth segment, like the scatter kernel): the full march (gather,
//                                 compositing, gradient math) WITHOUT any deposit; it stores the 4 gradient sources of
//                                 every sample (d rad_0..2 x C0, d v: 16 B) and cuts the ray into SEGMENTS of consecutive
//                                 samples whose 2x2x2 footprint starts in the same 8x8x8-cell REGION of the grid;
//   2. region_scan / region_fill  counting sort of the segments by region (no host round trip, no global cursor:
//                                 every (ray, depth segment) owns its slots of the segment table);
//   3. render_bwd_region_kernel   one block per region: lanes = segments; footprints are recomputed (index math only, no
//                                 gather), sources loaded, and the 8 corners x C channels go into a 9x9x9-voxel LDS
//                                 window of doubles (ds_add_f64, as in the tile kernel); ONE dense flush per region.
//
// Global atomics drop from ~3.6 requests per sample to ~160 per region (x 8000 regions at 160^3), the deposit runs at
// LDS speed, and the gathers of pass 1 are the only incoherent memory traffic left.
// Per-sample math: identical to render_bwd_kernel / render_bwd_packed_scatter_kernel (same device functions).
// Reference: autograd through thre3d_atom/rendering/volumetric/{sample,process,accumulate}.py, thre3d_reprs/voxels.py.
#include <limits.h>
#include <stdlib.h>

#include "voxe_device.hpp"
#include "voxe_launch.hpp"
#include "voxe_render_common.hpp"

namespace voxe {

constexpr int kRB = 8;             // region edge in cells (low-corner indices)
constexpr int kRW = kRB + 1;       // window edge in voxels
constexpr int kRWin = kRW * kRW * kRW;      // 729 voxels
constexpr int kRPlane = kRWin + 7;          // channel plane (doubles), padded off the bank period
#ifndef VOXE_REGION_CHUNK
#define VOXE_REGION_CHUNK 16       // longest segment (samples).  Swept on MI355X (recon batch, backward of an iteration): 4 / 6 / 8 / 16 -> 1.36 / 1.33 / 1.31 / 1.31 ms
#endif
#ifndef VOXE_REGION_BLOCK
#define VOXE_REGION_BLOCK 256      // threads of a region block (its waves share the LDS window): 64 / 128 / 256 -> 1.46 / 1.33 / 1.28 ms
#endif
constexpr int kSlotsPerLane = 16;  // segment slots of one (ray, depth segment); beyond: direct global atomics (rare)
constexpr unsigned kNoRegion = 0xFFFFFFFFu;

__host__ __device__ inline int regions_along(int N) { return ((N > 1 ? N - 1 : 1) + kRB - 1) / kRB; }

struct RegionScratch {
  float4* src;            // [R * S]   gradient sources per sample (features already x C0, density last used slot)
  unsigned* slot_region;  // [R * nseg * 16] region of every segment slot (kNoRegion: empty)
  uint2* slot_seg;        // [R * nseg * 16] (ray, k0 | k1 << 16)
  uint2* sorted;          // [R * nseg * 16] segments grouped by region
  unsigned* count;        // [nreg] segments per region
  unsigned* start;        // [nreg] first position of the region in `sorted`
  unsigned* fill;         // [nreg] cursor of region_fill_kernel
};

// ---- pass 1: march, sources, segments ------------------------------------------------------------------------------------
template <int COUT, int NCM>
__global__ __launch_bounds__(64) void render_bwd_src_kernel(
    DevGrid g, DevCfg c, const float* __restrict__ packed, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, const float* __restrict__ jitter, const float* __restrict__ colour,
    const float* __restrict__ depth, const float* __restrict__ acc, const float* __restrict__ d_colour,
    const float* __restrict__ d_depth, const float* __restrict__ d_acc, const float* __restrict__ ray_state,
    float* __restrict__ gpacked, const int want_d, const int want_f, RegionScratch rs) {
  constexpr int C = COUT + 1;
  constexpr int CM = COUT * NCM + 1;
  const int lane = threadIdx.x;
  const int nt = (int)((c.R + 63) / 64);
  const int nseg = num_segments(c.S, c.seg_len);
  const int nrb = gridDim.x / nseg;
  const int seg = blockIdx.x / nrb;
  const int logical = logical_tile_of(c, blockIdx.x - seg * nrb, nrb, 1, nt);
  if (logical < 0) return;
  const long long r0 = (long long)logical * 64 + lane;
  if (r0 >= c.R) return;
  const long long r = r0;

  RayCtx<COUT, NCM, 1> rc;
  rc.init(g, c, r, rays_o, rays_d, jitter);
  const int ks = seg * c.seg_len, ke = min(c.S, ks + c.seg_len) - 1;
  const int k_lo = max(rc.k_lo, ks), k_hi = min(rc.k_hi, ke);
  if (k_lo > k_hi) return;
  float T = 1.0f, pre_c[COUT], pre_a = 0.0f, pre_d = 0.0f;
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) pre_c[ch] = 0.0f;
  if (seg > 0) {
    constexpr int NC = COUT + 3;
    T = ray_state[ray_state_index(seg, 0, NC, c.R, r)];
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) pre_c[ch] = ray_state[ray_state_index(seg, 1 + ch, NC, c.R, r)];
    pre_a = ray_state[ray_state_index(seg, 1 + COUT, NC, c.R, r)];
    pre_d = ray_state[ray_state_index(seg, 2 + COUT, NC, c.R, r)];
  }
  float gc[COUT], gsum = 0.0f;
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) { gc[ch] = d_colour[r * COUT + ch]; gsum += gc[ch]; }
  const float gdep = d_depth ? d_depth[r] : 0.0f;
  const float gacc = d_acc ? d_acc[r] : 0.0f;
  const bool white = c.white && !c.attn;
  const float asum = acc[r];
  float total = gdep * depth[r] + gacc * asum;
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) {
    const float csum = white ? colour[r * COUT + ch] - (1.0f - asum) : colour[r * COUT + ch];
    total += gc[ch] * csum;
  }
  if (white) total -= gsum * asum;
  float prefix = gdep * pre_d + gacc * pre_a;
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) prefix += gc[ch] * pre_c[ch];
  if (white) prefix -= gsum * pre_a;

  const int nry = regions_along(g.Y), nrz = regions_along(g.Z);
  const long long slot0 = (r * nseg + seg) * kSlotsPerLane;
  int nslots = 0;
  unsigned cur_region = kNoRegion;
  int seg_k0 = 0;
  // close the open segment [seg_k0, k_end] (if any) into the next slot of this (ray, depth segment)
  auto emit = [&](int k_end) {
    if (cur_region == kNoRegion) return;
    rs.slot_region[slot0 + nslots] = cur_region;
    rs.slot_seg[slot0 + nslots] = make_uint2((unsigned)r, (unsigned)seg_k0 | ((unsigned)k_end << 16));
    atomicAdd(rs.count + cur_region, 1u);
    ++nslots;
  };
  float z_next = rc.dg.z(k_lo);
  for (int k = k_lo; k <= k_hi; ++k) {
    const float z = z_next;
    const bool last = (k == c.S - 1);
    if (!last) z_next = rc.dg.z(k + 1);
    float p[3];
    rc.point(z, p);
    Footprint fp;
    footprint(g, p, fp);
    float gch[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (fp.inside) {
      Cell cell;
      make_cell_fast(g, fp, cell);
      float v, rad[COUT];
      gather<COUT, NCM, 1>(g, packed, cell, rc.basis, v, rad);
      float sigma, dpost;
      post_activate_vg(g.post_act, v, sigma, dpost);
      const float dl = last ? kInfinity : (z_next - z);
      const float delta = dl * rc.dnorm;
      const float e = fast_exp(-(sigma * delta));
      const float alpha = 1.0f - e;
      const float om = 1.0f - alpha;
      const float wk = alpha * T;
      float col[COUT], dldw = fmaf(gdep, z, gacc);
#pragma unroll
      for (int ch = 0; ch < COUT; ++ch) { col[ch] = sigmoidf(rad[ch]); dldw = fmaf(gc[ch], col[ch], dldw); }
      if (white) dldw -= gsum;
      prefix = fmaf(dldw, wk, prefix);
      const float suffix = last ? 0.0f : (total - prefix);
      const float tail = (om > 0.0f) ? suffix * fast_rcp(om) : 0.0f;
      const float dsig = (delta * e) * fmaf(T, dldw, -tail);
      bool any = false;
#pragma unroll
      for (int ch = 0; ch < COUT; ++ch) {
        gch[ch] = want_f ? ((wk * gc[ch]) * (col[ch] * (1.0f - col[ch]))) * kC0 : 0.0f;
        any = any || (gch[ch] != 0.0f);
      }
      gch[COUT] = want_d ? dsig * dpost : 0.0f;
      any = any || (gch[COUT] != 0.0f);
      T = T * om;
      if (any) {
        const unsigned region =
            (unsigned)(((cell.i[0] / kRB) * nry + cell.i[1] / kRB) * nrz + cell.i[2] / kRB);
        const bool full = (k - seg_k0 + 1 > VOXE_REGION_CHUNK);
        if (region != cur_region || full) {
          emit(k - 1);
          if (nslots < kSlotsPerLane) {
            cur_region = region;
            seg_k0 = k;
          } else {
            // (segment table of this lane exhausted -- a ray that zig-zags through region corners: deposit directly)
            cur_region = kNoRegion;
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
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) gch[ch] = 0.0f;   // (nothing left for the deposit kernel)
          }
        }
      }
    }
    rs.src[r * c.S + k] = make_float4(gch[0], gch[1], gch[2], gch[3]);
  }
  emit(k_hi);
}

// ---- pass 2: counting sort of the segments by region ---------------------------------------------------------------------
__global__ __launch_bounds__(1024) void region_scan_kernel(const unsigned* __restrict__ count, unsigned* __restrict__ start,
                                                           int nreg) {
  // exclusive prefix sum of `count` (one block; nreg is a few thousand .. 32768)
  __shared__ unsigned partial[1024];
  const int tid = threadIdx.x;
  const int per = (nreg + 1023) / 1024;
  const int lo = tid * per, hi = min(nreg, lo + per);
  unsigned sum = 0;
  for (int i = lo; i < hi; ++i) sum += count[i];
  partial[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const unsigned v = tid >= off ? partial[tid - off] : 0u;
    __syncthreads();
    partial[tid] += v;
    __syncthreads();
  }
  unsigned run = tid > 0 ? partial[tid - 1] : 0u;
  for (int i = lo; i < hi; ++i) { start[i] = run; run += count[i]; }
}

__global__ __launch_bounds__(256) void region_fill_kernel(RegionScratch rs, long long nslots) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nslots) return;
  const unsigned region = rs.slot_region[i];
  if (region == kNoRegion) return;
  const unsigned pos = rs.start[region] + atomicAdd(rs.fill + region, 1u);
  rs.sorted[pos] = rs.slot_seg[i];
}

// ---- pass 3: one block per region, LDS window, one dense flush -------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(VOXE_REGION_BLOCK) void render_bwd_region_kernel(DevGrid g, DevCfg c, const float* __restrict__ rays_o,
                                                                const float* __restrict__ rays_d,
                                                                const float* __restrict__ jitter,
                                                                float* __restrict__ gpacked, const int cout, const int ncm,
                                                                const int chmask, RegionScratch rs) {
  __shared__ double win[C * kRPlane];
  const int tid = threadIdx.x;
  const unsigned region = blockIdx.x;
  const unsigned n = rs.count[region];
  if (n == 0) return;                       // block-uniform
  const unsigned first = rs.start[region];
  for (int i = tid; i < C * kRPlane; i += VOXE_REGION_BLOCK) win[i] = 0.0;
  const int nry = regions_along(g.Y), nrz = regions_along(g.Z);
  const int rz = (int)(region % (unsigned)nrz), ry = (int)((region / (unsigned)nrz) % (unsigned)nry);
  const int rx = (int)(region / (unsigned)(nrz * nry));
  const int ox = rx * kRB, oy = ry * kRB, oz = rz * kRB;   // window origin (voxels)
  __syncthreads();

  for (unsigned base = 0; base < n; base += VOXE_REGION_BLOCK) {
    const unsigned i = base + tid;
    if (i < n) {
      const uint2 sg = rs.sorted[first + i];
      const long long r = sg.x;
      const int k0 = (int)(sg.y & 0xFFFFu), k1 = (int)(sg.y >> 16);
      // ray context: origin, direction and the depth generator (no bounds, no SH basis: index math only)
      float o[3], d[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) { o[a] = rays_o[3 * r + a]; d[a] = rays_d[3 * r + a]; }
      DepthGen dg;
      dg.near = c.near; dg.far = c.far;
      dg.lindisp = c.lindisp != 0;
      if (c.aabb_clip) { ray_aabb_bounds(g, o, d, dg.near, dg.far); dg.lindisp = false; }
      dg.S = c.S; dg.half = c.S >> 1;
      dg.step = 1.0f / (float)(c.S - 1);
      dg.perturb = c.perturb != 0;
      dg.jit = jitter ? jitter + r * c.S : nullptr;
      dg.base = jitter_base(c.key0, c.key1, r);
      dg.kc = INT_MIN;
      for (int k = k0; k <= k1; ++k) {
        const float4 s4 = rs.src[r * c.S + k];
        const float z = dg.z(k);
        const float s[4] = {s4.x, s4.y, s4.z, s4.w};
        bool any = false;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) any = any || (s[ch] != 0.0f);
        if (!any) continue;
        float p[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float dz = d[a] * z; p[a] = o[a] + dz; }
        Footprint fp;
        footprint(g, p, fp);
        Cell cell;
        make_cell(g, fp, cell);
        const int lx = cell.i[0] - ox, ly = cell.i[1] - oy, lz = cell.i[2] - oz;
        if ((unsigned)lx >= (unsigned)kRB || (unsigned)ly >= (unsigned)kRB || (unsigned)lz >= (unsigned)kRB) continue;
        const int idx0 = (lx * kRW + ly) * kRW + lz;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float w = (cell.w[0][j & 1] * cell.w[1][(j >> 1) & 1]) * cell.w[2][j >> 2];
          const int idx = idx0 + (j & 1) * (kRW * kRW) + ((j >> 1) & 1) * kRW + (j >> 2);
#pragma unroll
          for (int ch = 0; ch < C; ++ch) {
            if ((chmask >> ch) & 1)
              __hip_atomic_fetch_add(&win[ch * kRPlane + idx], (double)(s[ch] * w), __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
  }
  __syncthreads();
  // flush: lanes = (voxel, channel), channel fastest -> 16-byte dense global atomics along the z-runs of the window
  const int CM = cout * ncm + 1;
  for (int e = tid; e < kRWin * C; e += VOXE_REGION_BLOCK) {
    const int vl = e / C, ch = e - vl * C;
    const double val = win[ch * kRPlane + vl];
    if (val == 0.0) continue;
    const int x = ox + vl / (kRW * kRW), y = oy + (vl / kRW) % kRW, zz = oz + vl % kRW;
    if (x >= g.X || y >= g.Y || zz >= g.Z) continue;   // (weight-0 corners of size-1 axes / the grid's far faces)
    const long long vox = ((long long)x * g.Y + y) * g.Z + zz;
    atomicAdd(gpacked + vox * CM + (ch == cout ? CM - 1 : ch * ncm), (float)val);
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
static long long region_min_rays() {
  const char* e = getenv("VOXE_REGION_MIN_RAYS");   // read per launch (tests / A-B runs flip it); < 0 disables the path
  return e ? atoll(e) : 16384ll;
}

bool region_bwd_supported(const DevGrid& g, const DevCfg& c, int deg, int diffuse, bool tiled) {
  const long long min_rays = region_min_rays();
  if (min_rays < 0 || c.R < min_rays || c.term_eps > 0.0f) return false;
  if (!(c.attn || deg == 0 || diffuse)) return false;          // one channel group (SH-0 / diffuse / attention)
  if (c.R >= (1ll << 32) || c.S >= 65536) return false;        // segment records: 32-bit ray, 16-bit sample indices
  if (!tiled) return true;                                     // unordered rays, image rows below the tile threshold
  // image-ordered launches: only when the pixels are clearly more than a voxel apart (nothing to combine inside a wave:
  // 100x100 cameras on a 160^3 grid).  The pixel spacing is not known on the host; for a camera that frames the volume
  // it is ~ grid side / image width.
  const char* e = getenv("VOXE_REGION_IMAGE_RATIO");
  const float ratio = e ? (float)atof(e) : 1.3f;
  const int side = g.X > g.Y ? (g.X > g.Z ? g.X : g.Z) : (g.Y > g.Z ? g.Y : g.Z);
  return (float)side >= ratio * (float)c.image_width;
}

static inline size_t up256(size_t x) { return (x + 255) / 256 * 256; }
struct RegionLayout { size_t src, slot_region, slot_seg, sorted, counters, total; long long nslots; int nreg; };
static RegionLayout region_layout(int X, int Y, int Z, long long R, int S) {
  RegionLayout l;
  const int nseg = num_segments(S, seg_len_for(R));
  l.nslots = R * nseg * kSlotsPerLane;
  l.nreg = regions_along(X) * regions_along(Y) * regions_along(Z);
  size_t off = 0;
  l.src = off; off += up256((size_t)R * S * sizeof(float4));
  l.slot_region = off; off += up256((size_t)l.nslots * sizeof(unsigned));
  l.slot_seg = off; off += up256((size_t)l.nslots * sizeof(uint2));
  l.sorted = off; off += up256((size_t)l.nslots * sizeof(uint2));
  l.counters = off; off += up256((size_t)3 * l.nreg * sizeof(unsigned));
  l.total = off;
  return l;
}
size_t region_scratch_bytes(int X, int Y, int Z, long long R, int S) {
  if (R <= 0 || S <= 0) return 0;
  return region_layout(X, Y, Z, R, S).total;
}

template <int COUT, int NCM>
static void launch_bwd_region_t(const DevGrid& g, const DevCfg& c, const BwdArgs& a, void* scratch, hipStream_t st) {
  const RegionLayout l = region_layout(g.X, g.Y, g.Z, c.R, c.S);
  char* base = (char*)scratch;
  RegionScratch rs;
  rs.src = (float4*)(base + l.src);
  rs.slot_region = (unsigned*)(base + l.slot_region);
  rs.slot_seg = (uint2*)(base + l.slot_seg);
  rs.sorted = (uint2*)(base + l.sorted);
  rs.count = (unsigned*)(base + l.counters);
  rs.start = rs.count + l.nreg;
  rs.fill = rs.start + l.nreg;
  (void)hipMemsetAsync(rs.slot_region, 0xFF, (size_t)l.nslots * sizeof(unsigned), st);
  (void)hipMemsetAsync(rs.count, 0, (size_t)3 * l.nreg * sizeof(unsigned), st);
  const int nseg = num_segments(c.S, c.seg_len);
  const int nb = blocks_for_tiles(c.map_mode, 1, (c.R + 63) / 64) * nseg;
  render_bwd_src_kernel<COUT, NCM><<<nb, 64, 0, st>>>(g, c, a.packed, a.rays_o, a.rays_d, a.jitter, a.colour, a.depth, a.acc,
                                                      a.d_colour, a.d_depth, a.d_acc, a.ray_state, a.gpacked,
                                                      a.want_d ? 1 : 0, a.want_f ? 1 : 0, rs);
  region_scan_kernel<<<1, 1024, 0, st>>>(rs.count, rs.start, l.nreg);
  region_fill_kernel<<<(int)((l.nslots + 255) / 256), 256, 0, st>>>(rs, l.nslots);
  constexpr int C = COUT + 1;
  const int chmask = (a.want_f ? ((1 << COUT) - 1) : 0) | (a.want_d ? (1 << COUT) : 0);
  render_bwd_region_kernel<C><<<l.nreg, VOXE_REGION_BLOCK, 0, st>>>(g, c, a.rays_o, a.rays_d, a.jitter, a.gpacked, COUT, NCM, chmask, rs);
}

void launch_bwd_region(const DevGrid& g, const DevCfg& c, int deg, int diffuse, const BwdArgs& a, void* scratch,
                       hipStream_t st) {
  (void)diffuse;
  if (c.attn) launch_bwd_region_t<1, 1>(g, c, a, scratch, st);
  else if (deg == 0) launch_bwd_region_t<3, 1>(g, c, a, scratch, st);
  else if (deg == 1) launch_bwd_region_t<3, 4>(g, c, a, scratch, st);
  else if (deg == 2) launch_bwd_region_t<3, 9>(g, c, a, scratch, st);
  else launch_bwd_region_t<3, 16>(g, c, a, scratch, st);
}

}  // namespace voxe
