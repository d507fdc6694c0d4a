// voxe_render_tile.hip -- backward render kernel for image-ordered rays: LDS gradient window.
//
// Why: the memory side of MI355X retires ~20 G atomic cache-line requests/s chip-wide
// (profiles/r01_microbench_atomics.md); scattering every trilinear corner with its own global
// atomicAdd (render_bwd_kernel, 735 M per 400x400 image) is bound by exactly that (33.7 ms).
// Neighbouring rays hit the same voxels (~2.6 rays per voxel width), so the adds are combined on chip:
//
//   * one WAVE (64 lanes) = one 8x8 pixel tile, one lane per ray, all lanes march in lock step over the
//     sample index k (the per-ray math is identical to render_bwd_kernel);
//   * the tile's gradient is accumulated in a per-wave LDS window that slides along the tile's dominant
//     march axis m: a ring of 6 voxel layers x (8 x 8) lateral voxels x C channels, stored as DOUBLE
//     and updated with ds_add_f64 (~10 clk per wave instruction; ds_add_f32 is a 193-clk serial path);
//     the lateral origin of every layer follows the tile's reference ray (a sheared, ray-aligned box),
//     so 8x8 suffices for any view direction;
//   * lanes of one cell rotate through the 8 corners ((c + rot(lane)) & 7) so that the lanes of a wave
//     instruction rarely target the same LDS address;
//   * a layer is flushed once no lane can touch it again (wave-min of the lanes' next cell layer): 16
//     voxels x 4 channels per global atomic instruction, i.e. 16-byte-dense requests (z-runs share lines);
//   * anything that falls outside the window goes straight to a global float atomic, so correctness
//     never depends on the window heuristics.
//
// Gradient formulas: see render_bwd_kernel (voxe_render.hip).  Reference: autograd through
// thre3d_atom/rendering/volumetric/{sample,process,accumulate}.py and thre3d_reprs/voxels.py.
#include <limits.h>
#include <stdlib.h>

#include <type_traits>

#include "voxe_device.hpp"
#include "voxe_launch.hpp"
#include "voxe_render_common.hpp"
#include "voxe_tile_window.hpp"

namespace voxe {

// WC = channels held by the window, CM = channels of a packed texel, memch = texel channel of THIS lane's window
// channel (lane % WC)
// DET (deterministic mode, test / race-check only): the window holds 64-bit FIXED-POINT sums (integer adds are
// associative: any order gives the same bits) and is flushed with 64-bit integer atomics into `gdet`.
template <int WC, int KL, bool DET = false>
__device__ __forceinline__ void flush_layer(double* __restrict__ win, float* __restrict__ gpacked,
                                            const Window& w, int key, int lane, int CM, int memch,
                                            unsigned long long* __restrict__ gdet = nullptr) {
  constexpr int C = WC;
  constexpr int kLat = KL, kLayerSlots = Lat<KL>::kLayerSlots;
  constexpr int kPerInstr = 64 / C;  // voxels per wave instruction (C == 4 -> 16, C == 2 -> 32)
  constexpr int NJ = (kLayerSlots + kPerInstr - 1) / kPerInstr;
  const int im = w.sgn * key;
  // r03: the layer's origin voxel is wave-uniform -- scalar registers, 64-bit scalar arithmetic -- and a lane adds its lateral
  // offset with 24-bit multiplies (full rate; 32 / 64-bit integer multiplies are quarter rate: three 64-bit ones per slot took
  // ~15 % of the kernel's VALU time); the NJ window reads of a layer are issued together (one LDS round trip instead of NJ)
  const int offu = __builtin_amdgcn_readfirstlane(w.off_u(im)), offv = __builtin_amdgcn_readfirstlane(w.off_v(im));
  const long long vox0 = (long long)im * w.stride_m + (long long)offu * w.stride_u + (long long)offv * w.stride_v;
  const int ch = lane % C;
  int idx[NJ];
  double val[NJ];
  // lateral cell (a, b) of this lane in instruction j (b is the z-run).  Parity-class banked window: the 16 voxels of an
  // instruction are two a-rows x eight b as ever, but dealt to the lanes as 2 x 2 blocks, so that 16 consecutive lanes see all
  // four (a, b) parity classes -- a layer has ONE slot parity, so 2-way bank conflicts are the floor of a single-layer flush
  // (rows of eight b per 32 lanes: 4-way).
  auto cell_of = [&](int j) {
    if constexpr (WinMap<KL, WC>::kPcb && KL == 8) { const int q = lane >> 2; return (2 * j + (q & 1)) * kLat + (q >> 1); }
    else return j * kPerInstr + lane / C;
  };
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int ab = cell_of(j);
    idx[j] = WinMap<KL, WC>::at(ring_slot(key), key, ab / kLat, ab % kLat, ch);
    const bool live = (kLayerSlots % kPerInstr == 0) || ab < kLayerSlots;
    val[j] = live ? win[idx[j]] : 0.0;        // (DET: the same 64 bits, read as a double only to be tested against zero)
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int ab = cell_of(j);
    if constexpr (DET) {
      const unsigned long long q = (unsigned long long)__double_as_longlong(val[j]);
      if (q != 0ull) {
        reinterpret_cast<unsigned long long*>(win)[idx[j]] = 0ull;
        const long long vox = vox0 + (long long)(__umul24((unsigned)(ab / kLat), (unsigned)w.stride_u) + __umul24((unsigned)(ab % kLat), (unsigned)w.stride_v));
        atomicAdd(gdet + vox * CM + memch, q);
      }
      continue;
    }
    if (val[j] != 0.0) {
      win[idx[j]] = 0.0;
      const long long vox = vox0 + (long long)(__umul24((unsigned)(ab / kLat), (unsigned)w.stride_u) + __umul24((unsigned)(ab % kLat), (unsigned)w.stride_v));
      atomicAdd(gpacked + vox * CM + memch, (float)val[j]);
    }
  }
}

// View-dependent grids (SH degree 1-3: NCU = 4 / 9 / 16 coefficients per colour): the window still holds 4 channels;
// the COUT * NCU + 1 gradient channels are split into GROUPS of 4 that run as sibling blocks (like the tile parts), each
// re-marching the segment (full gather: the colour needs every coefficient) and depositing its own 4 channels --
// d rad_c / d coef_cj = basis_j of the ray, a per-lane constant of the pass.  Per-voxel atomics stay combined in LDS;
// the cost is ngroups x the march instead of (8 corners x channels) global atomics per sample (135 ms -> see DESIGN.md).
//
// With a scratch buffer (BwdArgs::sample_src, 16 bytes per sample) the groups do not even re-march: MODE 1 marches ONCE
// (full gather, all the math) and stores the 4 per-sample gradient sources (d rad_0..2, d v); MODE 2 (one block per
// group) recomputes only the footprints, loads the sources and deposits its 4 channels.  MODE 0 does both in one kernel
// (every single-group render; view-dependent grids without the scratch buffer).
// fixed-point image of a float contribution (DET): the value was pre-multiplied by a power of two (exact), so the
// rounding to an integer is the only step that differs from the float path; two's-complement wrap makes signed sums work
__device__ __forceinline__ unsigned long long det_quant(float x) { return (unsigned long long)__double2ll_rn((double)x); }

template <int COUT, int NCM, int NCU, bool WANT_D, bool WANT_F, int MODE, int KL, bool DET = false>
// launch bounds swept: (64, 3) best; 2 and 4..6 are 6-9 % slower (register budget vs the LDS-bound residency)
#ifndef VOXE_TILE_LB
#define VOXE_TILE_LB 3
#endif
#ifndef VOXE_TILE_WIDE_FROM
#define VOXE_TILE_WIDE_FROM 9     // windows at least this wide get the relaxed register budget (2 waves per SIMD)
#endif
// (attention grids with frozen densities -- the refinement loop -- hold a 2-channel window of 6 - 10 KB and need 126 - 130
//  registers: asked to fit 128, they run 4 waves per SIMD; r05)
#ifndef VOXE_TILE_SRC_LB
#define VOXE_TILE_SRC_LB 3        // waves per SIMD the source pass (MODE 1) of view-dependent grids is built for: it holds no window (8 B of
                                  // LDS), so registers alone bound its residency -- built for 2 it took 177 / 178 of them at SH-1 / SH-3, ten
                                  // above what 3 waves leave; asked to fit 168 it spills 32 / 24 B and the SH-1 / 2 / 3 backward is
                                  // 1.91 / 2.97 / 5.29 -> 1.82 / 2.96 / 5.17 ms (400x400, 160^3)
#endif
__global__ __launch_bounds__(64, (COUT == 1 && NCU == 1 && !WANT_D && !DET) ? 4
                                 : ((NCU > 1 && MODE == 1 && KL < VOXE_TILE_WIDE_FROM) ? VOXE_TILE_SRC_LB
                                 : (((NCU > 1 && MODE != 2) || KL >= VOXE_TILE_WIDE_FROM) ? 2 : VOXE_TILE_LB))) void render_bwd_tile_kernel(
    DevGrid g, DevCfg c, const float* __restrict__ packed, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, const float* __restrict__ jitter,
    const float* __restrict__ colour, const float* __restrict__ depth,
    const float* __restrict__ acc, const float* __restrict__ d_colour,
    const float* __restrict__ d_depth, const float* __restrict__ d_acc,
    const float* __restrict__ ray_state, float* __restrict__ gpacked, const int qsplit, const int grp_begin,
    const int ngrp, float4* __restrict__ sample_src, unsigned long long* __restrict__ gdet = nullptr,
    float* __restrict__ det_scale = nullptr, const int det_phase = 0, const float fit_m = 5.5f,
    const float fit_lat_arg = 0.0f, const float4* __restrict__ sample_fwd = nullptr) {
  // DET: det_phase 0 measures max |contribution| per channel class into det_scale[0..1] (features, density) as
  // float bits (atomicMax: order independent); det_phase 1 deposits with the power-of-two scales det_scale[2..3].
  static_assert(!DET || MODE == 0, "the deterministic mode is the single-kernel backward");
  constexpr int CM = COUT * NCM + 1;          // channels of a packed texel
  constexpr int NG = COUT * NCU + 1;          // channels that receive a gradient (the used coefficients + density)
  constexpr int C = NG < 4 ? NG : 4;          // channels held by the LDS window
  constexpr int NGRP = (NG + C - 1) / C;      // channel groups (1 for SH-0 / diffuse / attention renders)
  static_assert(MODE == 0 || (COUT == 3 && NGRP > 1), "the two-phase backward is for view-dependent grids");
  constexpr int kLat = KL, kLayerSlots = Lat<KL>::kLayerSlots, kPlane = Lat<KL>::kPlane;
  constexpr int kWinDoubles = MODE == 1 ? 1 : WinMap<KL, C>::kDoubles;   // (the source pass has no window)
  __shared__ double win[kWinDoubles];
  const int lane = threadIdx.x;
  for (int i = lane; i < kWinDoubles; i += 64) win[i] = 0.0;

  // ---- tile -> ray (XCD-banded like map_ray) ----------------------------------------------------
  const int W = c.image_width;
  const int ntx = (W + 7) >> 3, nty = (int)tile_rows_total(c, 8);   // (all cameras of a multi-view launch)
  // a block = (pixel tile, depth segment[, part]).  qsplit == 4 (small images that leave the chip under-filled): the
  // parts of a tile run as sibling blocks instead of one after the other.
  // Block order: (part, SEGMENT)-major, tile minor -- all tiles at depth segment 0, then all at segment 1, ...  The
  // segments in front of / behind the volume are empty and retire at once; with the segments of a tile on consecutive
  // blocks instead, "empty" and "full" blocks alternate with period nseg, which aliases with the round-robin placement
  // on the 32 CUs of an XCD whenever nseg divides 32 (measured: kSegLen 32 -> 0.87 ms, 24 -> 0.68 ms for the same
  // work before this ordering).  Tiles are spread over the XCDs by logical_tile_of() (default: tile t on XCD t % 8).
  const int nseg = num_segments(c.S, c.seg_len);
  const int ntp = gridDim.x / (nseg * qsplit * ((NGRP == 1 || MODE == 1) ? 1 : ngrp));  // tile slots (a multiple of 8 >= ntx * nty)
  int part = blockIdx.x / ntp;
  int grp = 0;                                   // channel group of this block (outermost: group-major block order)
  if constexpr (NGRP > 1 && MODE != 1) {
    const int g_local = part / (nseg * qsplit);
    part -= g_local * nseg * qsplit;
    grp = grp_begin + g_local;
  }
  const int quad = part / nseg, seg = part - quad * nseg;
  const int tile = logical_tile_of(c, blockIdx.x % ntp, ntp, ntx, nty);
  if (tile < 0) return;  // launch padding (wave-uniform)
  const int ks = seg * c.seg_len, ke = min(c.S, ks + c.seg_len) - 1;  // samples of this segment
  // two-phase backward: slot of (tile, segment, sample k, lane) in the source buffer
  const long long src_base = ((long long)tile * nseg + seg) * c.seg_len * 64 + lane - (long long)ks * 64;
  const int ty = tile / ntx, tx = tile - ty * ntx;
  long long r_px;
  bool alive = tile_pixel_ray(c, ty, lane >> 3, (tx << 3) + (lane & 7), 8, r_px);
  long long r = alive ? r_px : 0;

  RayCtx<COUT, NCM, NCU> rc;
  rc.init(g, c, r, rays_o, rays_d, jitter);
  // Orientation of the tile in the wave (r03, wave-uniform): consecutive lanes (lane & 7) should step along the window's
  // lateral axis u (the stride-8 index of a layer), lane >> 3 along v.  With lanes running along the pixel rows that holds for
  // views whose image x axis maps to u; for the others (e.g. cameras above the volume whose image x is world y) the same
  // instruction count cost 19 % more LDS cycles in bank conflicts (PMC: SQ_LDS_BANK_CONFLICT 105 M vs 80 M per launch): those
  // tiles take their lanes down the pixel COLUMNS instead (profiles/r03_ab_orientation.txt).
  // (the parity-class banked window is conflict free whatever the lane order: no re-orientation there)
  // (decided by the deposit's window type alone: the two phases of the view-dependent backward must map lanes alike)
  if constexpr (!WinMap<KL, (COUT * NCU + 1 < 4 ? COUT * NCU + 1 : 4)>::kPcb) {
    const unsigned long long am0 = __ballot(alive);
    if ((am0 & 1ull) && (am0 >> 1 & 1ull) && (am0 >> 8 & 1ull)) {
      float d0[3], dx[3], dy[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float sc = g.scale[a] * (float)(a == 0 ? g.X : (a == 1 ? g.Y : g.Z));
        const float da = readlane_f32(rc.d[a], 0);
        d0[a] = fabsf(da * sc);
        dx[a] = fabsf((readlane_f32(rc.d[a], 1) - da) * sc);
        dy[a] = fabsf((readlane_f32(rc.d[a], 8) - da) * sc);
      }
      const int m0 = (d0[0] >= d0[1] && d0[0] >= d0[2]) ? 0 : ((d0[1] >= d0[2]) ? 1 : 2);
      const int u0 = (m0 == 0) ? 1 : 0;
      const float ex_u = u0 == 0 ? dx[0] : dx[1], ey_u = u0 == 0 ? dy[0] : dy[1];
      if (ex_u * VOXE_TILE_ORIENT_K < ey_u) {   // image y moves along u more than image x does: lanes down the columns
        alive = tile_pixel_ray(c, ty, lane & 7, (tx << 3) + (lane >> 3), 8, r_px);
        r = alive ? r_px : 0;
        rc.init(g, c, r, rays_o, rays_d, jitter);
      }
    }
  }

  // ---- pixel footprint: does the 8x8 tile fit the 8x8 lateral LDS window? ------------------------------
  // Lower-resolution images have pixels farther apart; a tile whose footprint exceeds the window is processed as two
  // 8x4 / 4x8 halves (32 lanes each) or, failing that, four 4x4 quadrants (16 lanes each) in consecutive passes over
  // the same window, so the gradient is still combined in LDS instead of falling back to one global atomic per
  // corner and channel.  (Wave-uniform decision.)
  // split: 0 = whole tile, 1 = left / right halves, 2 = top / bottom halves (2 passes of 32 lanes), 3 = quadrants
  int split = 0;
  {
    const unsigned long long am = __ballot(alive);
    if ((am >> 1 & 1ull) && (am >> 8 & 1ull)) {
      const int N[3] = {g.X, g.Y, g.Z};
      const float zref = readlane_f32(rc.dg.zlin(ke), 0);
      // Extent of the tile against the window (r03: the march axis has its own bound).  The window is a ring of kRing layers
      // along the march axis m; laterally it is KL wide.  A pass fits when (i) its lateral extent leaves room for the 2-wide
      // footprints (KL - 2.5) and (ii) the lanes' spread ALONG m at one sample index fits the ring next to the two layers of a
      // footprint and the jitter: `fit_m` layers.  r01 / r02 held all three axes against KL - 2.5: oblique views overflowed the
      // 6-layer ring -- 5 - 11 % of their samples took the 32 global atomics of the fallback at 128 - 266 pixels (0.78 vs
      // 0.31 ms for the same 200x200 image from two cameras).  Splitting an oblique tile costs blocks instead; which is cheaper
      // depends on how full the chip is, so the host picks fit_m by launch size (launch_bwd_tile_t; profiles/r03_ab_fit_m.txt).
      float d0[3], ex3[3], ey3[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float s = g.scale[a] * 0.5f * (float)N[a];
        const float da = readlane_f32(rc.d[a], 0);
        d0[a] = fabsf(da * s);
        ex3[a] = fabsf((readlane_f32(rc.d[a], 1) - da) * s * zref);
        ey3[a] = fabsf((readlane_f32(rc.d[a], 8) - da) * s * zref);
      }
      const int m = (d0[0] >= d0[1] && d0[0] >= d0[2]) ? 0 : ((d0[1] >= d0[2]) ? 1 : 2);
      const float fit_lat = fit_lat_arg > 0.0f ? fit_lat_arg : (float)KL - 2.5f;   // lateral extent (voxels) a pass may have: 5.5 for the 8-wide window
      auto fits_pass = [&](float wx, float wy) {            // a pass of (wx + 1) x (wy + 1) pixels
        float lat = 0.0f, alongm = 0.0f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const float e = wx * ex3[a] + wy * ey3[a];
          if (a == m) alongm = e; else lat = fmaxf(lat, e);
        }
        return lat <= fit_lat && alongm <= fit_m;
      };
      if (!fits_pass(7.0f, 7.0f)) {
        const bool hx = fits_pass(3.0f, 7.0f), hy = fits_pass(7.0f, 3.0f);
        const float sx = ex3[0] + ex3[1] + ex3[2], sy = ey3[0] + ey3[1] + ey3[2];
        if (hx && hy) split = (sx >= sy) ? 1 : 2;           // halve along the pixel axis that spreads the tile more
        else split = hx ? 1 : (hy ? 2 : 3);
      }
    }
  }
  // r04: the strata of this depth segment, tabulated once per block (SegDepth; not with per-ray AABB bounds)
  constexpr bool kStrata = (VOXE_TILE_STRATA_BWD && CM == 4 && MODE == 0 && !DET) || (VOXE_TILE_STRATA_SH && MODE != 0);
  __shared__ float2 strat[kStrata ? 64 : 1];
  bool use_strata = false;
  if constexpr (kStrata) {
    use_strata = !c.aabb_clip && (ke + 1 - ks) < 64;
    if (use_strata && lane <= ke + 1 - ks && ks + lane < c.S) strat[lane] = depth_stratum(rc.dg, ks + lane);
    __syncthreads();
  }
  auto run_pass = [&](const bool alive_q, const int centre_lane, const int centre_lane2) {
    rc.dg.kc = INT_MIN;                        // fresh rolling depth window for this pass
    const int k_lo = max(rc.k_lo, ks);          // this lane's samples inside the segment
    int k_hi = alive_q ? min(rc.k_hi, ke) : k_lo - 1;
    bool has = k_lo <= k_hi;
    // state at the segment start (transmittance + partial sums of the forward), see save_state()
    float T = 1.0f, suf_c[COUT], suf_a = 0.0f, suf_d = 0.0f;   // (suffix sums of the samples from the segment start on)
  #pragma unroll
    for (int ch = 0; ch < COUT; ++ch) suf_c[ch] = 0.0f;
    if (has && seg > 0) {
      constexpr int NC = COUT + 3;
      T = ray_state[ray_state_index(seg, 0, NC, c.R, r)];
  #pragma unroll
      for (int ch = 0; ch < COUT; ++ch) suf_c[ch] = ray_state[ray_state_index(seg, 1 + ch, NC, c.R, r)];
      suf_a = ray_state[ray_state_index(seg, 1 + COUT, NC, c.R, r)];
      suf_d = ray_state[ray_state_index(seg, 2 + COUT, NC, c.R, r)];
      if (c.term_eps > 0.0f && T < c.term_eps) { has = false; k_hi = k_lo - 1; }  // gradient truncation: nothing behind T < term_eps receives a gradient
    }
    const int kmin = wave_min_i32(has ? k_lo : INT_MAX);
    const int kmax = wave_max_i32(has ? k_hi : -1);
    if (kmin > kmax) return;  // wave-uniform: no ray of this pass meets the volume in this segment

    // ---- window geometry from a reference ray (the lane next to the tile centre, if it has samples) ----
    Window w;
    w.ctr = Lat<KL>::kCentre;
    {
      const unsigned long long hm = __ballot(has);
      const int ref = ((hm >> centre_lane) & 1ull) ? centre_lane : (__ffsll((long long)hm) - 1);
      // r03: the window follows the CENTRE of the pass -- the mean of the two pixels around it (the centre lies between pixels) --
      // when both have samples: a quarter voxel more slack per lateral axis than following pixel (3, 3); the model of
      // tools/sim/lds_conflicts.py has 25 - 40 % fewer wave-samples with a lane outside the window for diagonal views
      const int ref2 = (VOXE_TILE_CENTRE2 && ref == centre_lane && ((hm >> centre_lane2) & 1ull)) ? centre_lane2 : ref;
      const int N[3] = {g.X, g.Y, g.Z};
      float U0[3], DU[3];
  #pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float ro = readlane_f32(rc.o[a], ref), rd = 0.5f * (readlane_f32(rc.d[a], ref) + readlane_f32(rc.d[a], ref2));
        const float half = 0.5f * (float)N[a];
        U0[a] = ((ro * g.scale[a] + g.bias[a]) + 1.0f) * half - 0.5f;
        DU[a] = rd * g.scale[a] * half;
      }
      const float ax = fabsf(DU[0]), ay = fabsf(DU[1]), az = fabsf(DU[2]);
      w.m = (ax >= ay && ax >= az) ? 0 : ((ay >= az) ? 1 : 2);
      w.u = (w.m == 0) ? 1 : 0;
      w.v = (w.m == 2) ? 1 : 2;
      const float DUm = (w.m == 0) ? DU[0] : ((w.m == 1) ? DU[1] : DU[2]);
      const float U0m = (w.m == 0) ? U0[0] : ((w.m == 1) ? U0[1] : U0[2]);
      const float DUu = (w.u == 0) ? DU[0] : DU[1], U0u = (w.u == 0) ? U0[0] : U0[1];
      const float DUv = (w.v == 1) ? DU[1] : DU[2], U0v = (w.v == 1) ? U0[1] : U0[2];
      w.sgn = (DUm < 0.0f) ? -1 : 1;
      const float inv = (DUm != 0.0f) ? 1.0f / DUm : 0.0f;
      w.Bu = DUu * inv; w.Au = U0u - w.Bu * U0m;
      w.Bv = DUv * inv; w.Av = U0v - w.Bv * U0m;
      const int sx = g.Y * g.Z, sy = g.Z;
      w.stride_m = (w.m == 0) ? sx : ((w.m == 1) ? sy : 1);
      w.stride_u = (w.u == 0) ? sx : sy;
      w.stride_v = (w.v == 1) ? sy : 1;
    }
    // lowest layer key a sample with low-corner index pm can write (layers pm and pm + 1)
    auto minkey = [&](int pm) { return w.sgn > 0 ? pm : -(pm + 1); };
    auto pick = [&](const int (&t)[3], int axis) { return axis == 0 ? t[0] : (axis == 1 ? t[1] : t[2]); };
    // r04: the march of the pass with the window's axes as COMPILE-TIME constants (MA = 0 / 1 / 2; -1: run time).  The deposit
    // is VALU bound since the banked window (LDS issue 0.39, VALU issue 0.83), and the run-time axis picks of the cell's
    // (march, u, v) indices / weights were ~16 v_cndmask per sample on wave-uniform masks that live in spilled SGPRs
    // (v_readlane + s_nop each): like fwd_window_march<M>.  Only the 4-channel texel kernels are instantiated three times.
    auto march = [&](auto axis_tag, auto strata_tag) {
    constexpr int MA = decltype(axis_tag)::value;
    const SegDepth<decltype(strata_tag)::value> sd{strat, ks};
    constexpr int UA = (MA == 0) ? 1 : 0, VA = (MA == 2) ? 1 : 2;
    auto pick_m = [&](const int (&t)[3]) { if constexpr (MA >= 0) return t[MA]; else return pick(t, w.m); };

    // ---- per-ray constants of the backward (see render_bwd_kernel) -------------------------------
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
    // suffix0 = sum_{j >= segment start} dL/dw_j w_j: the whole ray (from the forward outputs) for the first segment, the
    // saved suffix sums (back-to-front sums of the forward: no cancellation against the part in front) otherwise
    float suffix0 = total;
    if (seg > 0) {
      suffix0 = gdep * suf_d + gacc * suf_a;
#pragma unroll
      for (int ch = 0; ch < COUT; ++ch) suffix0 += gc[ch] * suf_c[ch];
      if (white) suffix0 -= gsum * suf_a;
    }
    float run = 0.0f;   // sum of dL/dw_j w_j over the samples of this segment up to and including the current one

    // window channel s of this block = gradient channel q = grp * C + s: coefficient j of colour ch = q / NCU (texel
    // channel ch * NCM + j, factor basis_j of this ray) or, last, the density (texel channel CM - 1, factor 1).
    // NGRP == 1: grp == 0 and everything below is a compile-time constant.
    int chsel[C], memch[C];     // wave-uniform: which of (d rad_0 .. d rad_{COUT-1}, d v) feeds the slot; -1 = unused
    float mult[C];
  #pragma unroll
    for (int s = 0; s < C; ++s) {
      const int q = grp * C + s;
      if (q >= NG) { chsel[s] = -1; memch[s] = 0; mult[s] = 0.0f; }
      else if (q == NG - 1) { chsel[s] = COUT; memch[s] = CM - 1; mult[s] = 1.0f; }
      else {
        const int ch = q / NCU, j = q - ch * NCU;
        chsel[s] = ch; memch[s] = ch * NCM + j;
        float b = rc.basis[0];
  #pragma unroll
        for (int t = 1; t < NCU; ++t) b = (j == t) ? rc.basis[t] : b;
        mult[s] = b;
      }
    }
    float dscale[C];            // DET: power-of-two scale of every window channel; det_max: largest |contribution| seen
    float det_max_f = 0.0f, det_max_d = 0.0f;
  #pragma unroll
    for (int s = 0; s < C; ++s) dscale[s] = 1.0f;
    if constexpr (DET) {
      if (det_phase == 1) {
  #pragma unroll
        for (int s = 0; s < C; ++s) dscale[s] = (chsel[s] == COUT) ? det_scale[3] : det_scale[2];
      }
    }
    int my_memch = memch[0];    // texel channel of window channel lane % C (flush)
  #pragma unroll
    for (int s = 1; s < C; ++s) my_memch = (lane % C == s) ? memch[s] : my_memch;

    // first sample of every ray (rolling: z_cur / fp_cur always describe sample max(k, k_lo))
    float z_cur = 0.0f;
    Footprint fp_cur;
    fp_cur.inside = false;
  #pragma unroll
    for (int a = 0; a < 3; ++a) { fp_cur.i0[a] = 0; fp_cur.w[a][0] = fp_cur.w[a][1] = 0.0f; }
    bool dead = false;          // early ray termination reached (term_eps > 0)
    int first_key = INT_MAX;
    if (has) {
      z_cur = sd.z(rc.dg, k_lo);
      float p[3];
      rc.point(z_cur, p);
      footprint(g, p, fp_cur);
      first_key = minkey(pick_m(fp_cur.i0));
    }
    w.base = wave_min_i32(first_key);
    __syncthreads();  // window zeroed

    const int rot = VOXE_TILE_LANEROT(lane) & 7;  // per-lane corner permutation: neighbours in the tile differ
    for (int k = kmin; k <= kmax; ++k) {
      const bool on = has && (k >= k_lo) && (k <= k_hi);
      if (on) {
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
        if (fp.inside) {
          Cell cell;
          make_cell_fast(g, fp, cell);
          // gradient w.r.t. (rad_0 .. rad_{COUT-1}, v) of this sample
          float gsrc[COUT + 1];
          if constexpr (MODE == 2) {
            const float4 t4 = sample_src[src_base + (long long)k * 64];
            gsrc[0] = t4.x; gsrc[1] = t4.y; gsrc[2] = t4.z; gsrc[3] = t4.w;
          } else {
  #pragma unroll
            for (int ch = 0; ch <= COUT; ++ch) gsrc[ch] = 0.0f;
            if (!dead) {
              float v = 0.0f, rad[COUT];
              bool from_fwd = false;
              if constexpr (MODE == 1 && COUT == 3) {
                // r04: the forward of the same rays left (rad, v) of this sample in the source buffer's layout: no gather
                if (sample_fwd) {
                  const float4 f4 = sample_fwd[src_base + (long long)k * 64];
                  rad[0] = f4.x; rad[1] = f4.y; rad[2] = f4.z; v = f4.w;
                  from_fwd = true;
                }
              }
              if (!from_fwd) gather<COUT, NCM, NCU>(g, packed, cell, rc.basis, v, rad);
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
              run = fmaf(dldw, wk, run);
              const float suffix = last ? 0.0f : (suffix0 - run);
              const float tail = (om > 0.0f) ? suffix * fast_rcp(om) : 0.0f;
              const float dsig = (delta * e) * fmaf(T, dldw, -tail);
  #pragma unroll
              for (int ch = 0; ch < COUT; ++ch) gsrc[ch] = WANT_F ? (wk * gc[ch]) * (col[ch] * (1.0f - col[ch])) : 0.0f;
              gsrc[COUT] = WANT_D ? dsig * dpost : 0.0f;
              T = T * om;
            }
            if constexpr (MODE == 1) sample_src[src_base + (long long)k * 64] = make_float4(gsrc[0], gsrc[1], gsrc[2], gsrc[3]);
          }
          // per window channel: x basis_j (SH-0: C0) resp. x 1
          float gch[C];
          bool any = false;
  #pragma unroll
          for (int s = 0; s < C; ++s) {
            float x = 0.0f;
  #pragma unroll
            for (int t = 0; t <= COUT; ++t) x = (chsel[s] == t) ? gsrc[t] : x;
            gch[s] = (chsel[s] == COUT) ? x : x * mult[s];
            any = any || (gch[s] != 0.0f);
          }
          if constexpr (MODE == 1) any = false;   // the source pass deposits nothing
          if constexpr (DET) {
            if (det_phase == 0) {   // measuring pass: weights are <= 1, so |gch| bounds every contribution
  #pragma unroll
              for (int s = 0; s < C; ++s) {
                if (chsel[s] == COUT) det_max_d = fmaxf(det_max_d, fabsf(gch[s]));
                else det_max_f = fmaxf(det_max_f, fabsf(gch[s]));
              }
              any = false;
            } else {
  #pragma unroll
              for (int s = 0; s < C; ++s) gch[s] *= dscale[s];   // exact (power of two)
            }
          }

          if (MODE != 1 && any) {
#if VOXE_TILE_SLOTINC
            const int base_slot = ring_slot(w.base);   // wave-uniform (scalar unit)
#endif
            // the cell in (march, lateral u, lateral v) order; all 8 corners are in range (make_cell)
            int pm, pu, pv;
            float wm[2], wu[2], wv[2];
            if constexpr (MA >= 0) {
              pm = cell.i[MA]; pu = cell.i[UA]; pv = cell.i[VA];
  #pragma unroll
              for (int s = 0; s < 2; ++s) { wm[s] = cell.w[MA][s]; wu[s] = cell.w[UA][s]; wv[s] = cell.w[VA][s]; }
            } else {
              pm = pick(cell.i, w.m); pu = pick(cell.i, w.u); pv = pick(cell.i, w.v);
  #pragma unroll
              for (int s = 0; s < 2; ++s) {
                wm[s] = (w.m == 0) ? cell.w[0][s] : ((w.m == 1) ? cell.w[1][s] : cell.w[2][s]);
                wu[s] = (w.u == 0) ? cell.w[0][s] : cell.w[1][s];
                wv[s] = (w.v == 1) ? cell.w[1][s] : cell.w[2][s];
              }
            }
            // per layer (cm = 0, 1): ring slot, lateral position of corner (cu, cv) = (0, 0), window test
            int lofs[2], ab0[2], la[2], lb[2], lsl[2];
            bool fits = true;
  #pragma unroll
            for (int s = 0; s < 2; ++s) {
              const int im = pm + s, key = w.sgn * im;
              const int a0 = pu - w.off_u(im), b0 = pv - w.off_v(im);
              const int dk = key - w.base;
              fits = fits && ((unsigned)dk < (unsigned)kRing) && ((unsigned)a0 < (unsigned)(kLat - 1)) &&
                     ((unsigned)b0 < (unsigned)(kLat - 1));
#if VOXE_TILE_SLOTINC
              // ring slot of a live layer = slot of the window base (wave-uniform) + its distance, wrapped once: no
              // per-lane modulo (keys outside the ring give garbage here and take the `!fits` path, which recomputes)
              int sl = base_slot + dk;
              sl = sl >= kRing ? sl - kRing : sl;
#else
              const int sl = ring_slot(key);
#endif
              lofs[s] = sl * kLayerSlots;
              ab0[s] = a0 * kLat + b0 + (KL == 8 ? VOXE_TILE_ROT * sl : 0);  // + the per-layer rotation of Lat::pos()
              la[s] = a0; lb[s] = b0; lsl[s] = sl;
            }
            // (density-only / feature-only SH-0 backwards deposit their zero channels too: exact zeros, skipped by the flush)
            constexpr bool kPcbPath = WinMap<KL, C>::kPcb;
            if (kPcbPath && fits) {
              // Parity-class banked deposit (see WinMap): instruction (cc, j) of this lane goes to the corner whose window
              // voxel has parity class cc ^ (hm, hu, hv) -- lane bits 2, 3, 4 -- and to channel (j + lane bits 0..1) & 3.
              // Instructions with the slot-parity bit 0 ("A") take the layer whose ring slot has parity hm, the others
              // ("B") the other layer; per layer and lateral axis the corner with parity h is used where the axis bit is 0.
              // (The two layers of a footprint have their own lateral origins, hence their own a / b parities.)
              if constexpr (kPcbPath) {
                // lane bits -> (channel rotation, parity class): bit 0 and bit 4 rotate the channel, bits 1..3 pick the class.
                // Any 16 consecutive lanes then differ in index bits 0..3 (bank pair modulo 16: channel bit 0 + class) and any
                // 32 in bits 0..4 -- conflict free whether the LDS serves a 64-bit atomic in groups of 16 or of 32 lanes
                // (the first assignment, class from bits 2..4 and channel from bits 0..1, left 64 M of the 81 M conflict
                // cycles: 16 lanes shared 8 bank pairs).
                const int hm = (lane >> 1) & 1, hu = (lane >> 2) & 1, hv = (lane >> 3) & 1;
                const bool rm = ((lsl[0] ^ hm) & 1) != 0;
                const int sA = rm ? lsl[1] : lsl[0], sB = rm ? lsl[0] : lsl[1];
                const int aA = rm ? la[1] : la[0], aB = rm ? la[0] : la[1];
                const int bA = rm ? lb[1] : lb[0], bB = rm ? lb[0] : lb[1];
                const float wmA = rm ? wm[1] : wm[0], wmB = rm ? wm[0] : wm[1];
                // (t0, x0): index term / weight of the corner with parity h along one lateral axis, (t1, x1): the other corner
                auto split = [](int a, int h, float w0, float w1, int stride, int lo, int& t0, int& t1, float& x0, float& x1) {
                  const int d0 = (a ^ h) & 1;
                  const int c0 = a + d0, c1 = a + 1 - d0;
                  t0 = (c0 >> 1) * stride + (h << lo);
                  t1 = (c1 >> 1) * stride + ((1 - h) << lo);
                  x0 = d0 ? w1 : w0;
                  x1 = d0 ? w0 : w1;
                };
                int gA[2], gB[2], hA[2], hB[2];
                float wuA[2], wuB[2], wvA[2], wvB[2];
                split(aA, hu, wu[0], wu[1], WinMap<KL, C>::kSA, 1, gA[0], gA[1], wuA[0], wuA[1]);
                split(aB, hu, wu[0], wu[1], WinMap<KL, C>::kSA, 1, gB[0], gB[1], wuB[0], wuB[1]);
                split(bA, hv, wv[0], wv[1], WinMap<KL, C>::kSB, 0, hA[0], hA[1], wvA[0], wvA[1]);
                split(bB, hv, wv[0], wv[1], WinMap<KL, C>::kSB, 0, hB[0], hB[1], wvB[0], wvB[1]);
                const int FA = (sA >> 1) * WinMap<KL, C>::kSS + (hm << 2), FB = (sB >> 1) * WinMap<KL, C>::kSS + ((1 - hm) << 2);   // (sA & 1 == hm by the choice of A)
                const int fgA[2] = {FA + gA[0], FA + gA[1]}, fgB[2] = {FB + gB[0], FB + gB[1]};
                const float wmuA[2] = {wmA * wuA[0], wmA * wuA[1]}, wmuB[2] = {wmB * wuB[0], wmB * wuB[1]};
                const int crot = (lane & 1) | ((lane >> 3) & 2);
                const bool c1 = crot & 1, c2 = crot & 2;
                const float q0 = c1 ? gch[1] : gch[0], q1 = c1 ? gch[2] : gch[1], q2 = c1 ? gch[3] : gch[2], q3 = c1 ? gch[0] : gch[3];
                const float gr[4] = {c2 ? q2 : q0, c2 ? q3 : q1, c2 ? q0 : q2, c2 ? q1 : q3};   // gr[j] = gch[(j + crot) & 3]
                int choff[4];
  #pragma unroll
                for (int j = 0; j < 4; ++j) choff[j] = ((j + crot) & 3) << 3;
  #pragma unroll
                for (int cc = 0; cc < 8; ++cc) {
                  const int bm = cc & 1, bu = (cc >> 1) & 1, bv = cc >> 2;  // compile-time bits of this instruction
                  const float wgt = (bm ? wmuB[bu] : wmuA[bu]) * (bm ? wvB[bv] : wvA[bv]);
                  const int idx = (bm ? fgB[bu] : fgA[bu]) + (bm ? hB[bv] : hA[bv]);
  #pragma unroll
                  for (int ch = 0; ch < 4; ++ch) {
                    if constexpr (DET)
                      __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(win) + (choff[ch] + idx),
                                             det_quant(gr[ch] * wgt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else
                      __hip_atomic_fetch_add(&win[choff[ch] + idx], (double)(gr[ch] * wgt), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_WORKGROUP);
                  }
                }
              }
            } else if (fits) {  // common case: the whole 2x2x2 footprint is inside the LDS window
              // Lanes permute the corner order (corner index XOR rot, rot = 3 per-lane bits) AND the channel order
              // (crot): the lanes of one wave instruction then spread over 8 corners x C channel planes, so lanes that
              // share a voxel rarely hit the same LDS address / bank in the same instruction.  With an XOR the
              // corner bits of instruction cc are (constant bit) ^ (lane bit): every operand pair is swapped ONCE per
              // sample ("A" = value used where the constant bit is 0, "B" where it is 1) and the unrolled loop below
              // contains no selects at all.
              constexpr bool kAllCh = (WANT_D && WANT_F) || NGRP > 1;   // (several groups: slots are not channels)
              const bool r0 = rot & 1, r1 = rot & 2, r2 = rot & 4;
              const float wmA = r0 ? wm[1] : wm[0], wmB = r0 ? wm[0] : wm[1];
              const float wuA = r1 ? wu[1] : wu[0], wuB = r1 ? wu[0] : wu[1];
              const float wvA = r2 ? wv[1] : wv[0], wvB = r2 ? wv[0] : wv[1];
              const int lofA = r0 ? lofs[1] : lofs[0], lofB = r0 ? lofs[0] : lofs[1];
              const int abA = r0 ? ab0[1] : ab0[0], abB = r0 ? ab0[0] : ab0[1];
              const int uA = r1 ? kLat : 0, uB = kLat - uA, vA = r2 ? 1 : 0, vB = 1 - vA;
              const float wmu[4] = {wmA * wuA, wmB * wuA, wmA * wuB, wmB * wuB};       // [cm + 2 cu]
              const int abu[4] = {abA + uA, abB + uA, abA + uB, abB + uB};
              float gr[C];
              int poff[C];
              if constexpr (kAllCh && C == 4) {
                const int crot = VOXE_TILE_CROT(lane) & 3;
                const bool c1 = crot & 1, c2 = crot & 2;
                // gr[j] = gch[(j + crot) & 3], poff[j] = plane offset of that channel
                const float a0 = c1 ? gch[1] : gch[0], a1 = c1 ? gch[2] : gch[1], a2 = c1 ? gch[3] : gch[2], a3 = c1 ? gch[0] : gch[3];
                gr[0] = c2 ? a2 : a0; gr[1] = c2 ? a3 : a1; gr[2] = c2 ? a0 : a2; gr[3] = c2 ? a1 : a3;
  #pragma unroll
                for (int j = 0; j < 4; ++j) poff[j] = ((j + crot) & 3) * kPlane;
              } else {
  #pragma unroll
                for (int j = 0; j < C; ++j) { gr[j] = gch[j]; poff[j] = j * kPlane; }
              }
#if VOXE_TILE_F64MUL
              double grd[C];   // products in double: 4 + 8 conversions per sample instead of 32 (one per product)
  #pragma unroll
              for (int ch = 0; ch < C; ++ch) grd[ch] = (double)gr[ch];
#endif
  #pragma unroll
              for (int cc = 0; cc < 8; ++cc) {
                const int bm = cc & 1, bu = (cc >> 1) & 1, bv = cc >> 2;  // compile-time bits of this instruction
                const float wgt = wmu[bm + 2 * bu] * (bv ? wvB : wvA);
                const int idx = (bm ? lofB : lofA) + Lat<KL>::wrap(abu[bm + 2 * bu] + (bv ? vB : vA));
#if VOXE_TILE_F64MUL
                const double wgtd = (double)wgt;
#endif
  #pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                  if (kAllCh || (ch < COUT && WANT_F) || (ch == COUT && WANT_D)) {
#if VOXE_TILE_F64MUL
                    if constexpr (!DET) {
                      __hip_atomic_fetch_add(&win[poff[ch] + idx], grd[ch] * wgtd, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_WORKGROUP);
                      continue;
                    }
#endif
                    if constexpr (DET)
                      __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(win) + (poff[ch] + idx),
                                             det_quant(gr[ch] * wgt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else
                      __hip_atomic_fetch_add(&win[poff[ch] + idx], (double)(gr[ch] * wgt), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_WORKGROUP);
                  }
                }
              }
            } else {  // some corner outside the window: per-corner test, global scatter for the outsiders (rare)
  #pragma unroll
              for (int cc = 0; cc < 8; ++cc) {
                const int cm = cc & 1, cu = (cc >> 1) & 1, cv = cc >> 2;
                const float wgt = (wm[cm] * wu[cu]) * wv[cv];
                if (wgt != 0.0f) {
                  const int im = pm + cm, iu = pu + cu, iv = pv + cv;
                  const int key = w.sgn * im;
                  const int a = iu - w.off_u(im), b = iv - w.off_v(im);
                  const bool inwin = ((unsigned)(key - w.base) < (unsigned)kRing) && ((unsigned)a < (unsigned)kLat) &&
                                     ((unsigned)b < (unsigned)kLat);
                  if (inwin) {
                    const int idx = WinMap<KL, C>::at(ring_slot(key), key, a, b, 0);
                    constexpr int kChStep = WinMap<KL, C>::kPcb ? 8 : kPlane;
  #pragma unroll
                    for (int ch = 0; ch < C; ++ch) {
                      if (NGRP > 1 || (ch < COUT && WANT_F) || (ch == COUT && WANT_D)) {
                        if constexpr (DET)
                          __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(win) + (ch * kChStep + idx),
                                                 det_quant(gch[ch] * wgt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        else
                          __hip_atomic_fetch_add(&win[ch * kChStep + idx], (double)(gch[ch] * wgt), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_WORKGROUP);
                      }
                    }
                  } else {
                    const long long vox = (long long)im * w.stride_m + (long long)iu * w.stride_u + (long long)iv * w.stride_v;
  #pragma unroll
                    for (int ch = 0; ch < C; ++ch) {
                      if (NGRP > 1 ? (chsel[ch] >= 0) : ((ch < COUT && WANT_F) || (ch == COUT && WANT_D))) {
                        if constexpr (DET) atomicAdd(gdet + vox * CM + memch[ch], det_quant(gch[ch] * wgt));
                        else atomicAdd(gpacked + vox * CM + memch[ch], gch[ch] * wgt);
                      }
                    }
                  }
                }
              }
            }
          }
          if constexpr (MODE == 0) { if (c.term_eps > 0.0f && T < c.term_eps) k_hi = k; }
          // (two-phase: the source pass keeps writing zeros so that the deposit pass reads defined values)
          if constexpr (MODE == 1) { if (c.term_eps > 0.0f && T < c.term_eps) dead = true; }
        }
      }
      // ---- slide the window: flush every layer no lane can reach any more ---------------------------
      if constexpr (MODE == 1) continue;
      int lb = INT_MAX;
      if (has) {
        if (k + 1 < k_lo) lb = first_key;
        else if (k + 1 <= k_hi) lb = minkey(pick_m(fp_cur.i0));
      }
      const int newbase = wave_min_i32(lb);
      // VOXE_TILE_SLIDE_MIN > 1: let the window lag -- flush only once that many layers can go at once (every flush
      // drains the wave's LDS queue: fewer, larger flushes; the ring must hold the extra layers)
      if (newbase >= w.base + VOXE_TILE_SLIDE_MIN) {  // wave-uniform
        __syncthreads();
        const long long adv = (long long)newbase - (long long)w.base;
        const int nflush = adv < kRing ? (int)adv : kRing;
        for (int i = 0; i < nflush; ++i) flush_layer<C, KL, DET>(win, gpacked, w, w.base + i, lane, CM, my_memch, gdet);
        w.base = newbase;
        __syncthreads();
      }
    }
    __syncthreads();
    if constexpr (MODE != 1) {
      if (w.base != INT_MAX) {
        for (int i = 0; i < kRing; ++i) flush_layer<C, KL, DET>(win, gpacked, w, w.base + i, lane, CM, my_memch, gdet);
      }
    }
    if constexpr (DET) {
      if (det_phase == 0) {   // non-negative floats order like their bit patterns: atomicMax on the bits
        const unsigned mf = (unsigned)wave_max_i32((int)__float_as_uint(det_max_f));
        const unsigned md = (unsigned)wave_max_i32((int)__float_as_uint(det_max_d));
        if (lane == 0) {
          atomicMax(reinterpret_cast<unsigned*>(det_scale), mf);
          atomicMax(reinterpret_cast<unsigned*>(det_scale) + 1, md);
        }
      }
    }
    };   // march
    auto march_axis = [&](auto strata_tag) {
      if constexpr (VOXE_TILE_AXIS_TEMPLATE && !DET && ((CM == 4 && MODE == 0) || (VOXE_TILE_AXIS_TEMPLATE_SH && MODE == 2))) {
        if (w.m == 0) march(std::integral_constant<int, 0>{}, strata_tag);
        else if (w.m == 1) march(std::integral_constant<int, 1>{}, strata_tag);
        else march(std::integral_constant<int, 2>{}, strata_tag);
      } else {
        march(std::integral_constant<int, -1>{}, strata_tag);
      }
    };
    if constexpr (kStrata) {
      if (use_strata) march_axis(std::true_type{});
      else march_axis(std::false_type{});
    } else {
      march_axis(std::false_type{});
    }
  };
  // lanes and reference lane of part q under the chosen split
  auto in_part = [&](int q) {
    const int hx = (lane >> 2) & 1, hy = (lane >> 5) & 1;
    return split == 0 ? true : (split == 1 ? hx == q : (split == 2 ? hy == q : hx + 2 * hy == q));
  };
  auto centre_of = [&](int q) {
    return split == 0 ? 27 : (split == 1 ? 26 + 4 * q : (split == 2 ? 19 + 32 * q : 18 + 4 * (q & 1) + 32 * (q >> 1)));
  };
  auto centre2_of = [&](int q) {   // the pixel diagonally across the part's centre from centre_of(q)
    return split == 0 ? 36 : (split == 1 ? 33 + 4 * q : (split == 2 ? 12 + 32 * q : 9 + 4 * (q & 1) + 32 * (q >> 1)));
  };
  const int nparts = split == 0 ? 1 : (split == 3 ? 4 : 2);
  // qsplit == 4: the parts of a tile run as sibling blocks (siblings without a part have nothing to do); otherwise one
  // block walks its parts.  ONE call site: the pass is ~3.5 k instructions and would otherwise be inlined four times.
  const int q_begin = qsplit == 4 ? quad : 0;
  const int q_end = qsplit == 4 ? min(quad + 1, nparts) : nparts;
  for (int q = q_begin; q < q_end; ++q) {
    run_pass(alive && in_part(q), centre_of(q), centre2_of(q));
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Forward with the tile's texels staged in LDS (SH-0 grids, image-ordered rays; r02, rebuilt in r03).
//
// The ray-ordered forward (render_fwd_seg_kernel) fetches the 8 corner texels of every sample from L1 / L2: at 400x400 on
// 160^3 a texel is requested ~13 times by the 64 rays of a tile within one 32-sample segment, and those requests -- not
// the ~170 VALU instructions of a sample -- pace the kernel (VALU issue 47 % busy; with the gathers replaced by one
// shared read the kernel takes 0.143 instead of 0.247 ms, profiles/r03_ab_fwd_window.txt).  Here one wave (8x8-pixel tile, depth
// segment) keeps the same sheared, ray-aligned window as the backward -- a ring of kTexRing layers along the march axis x
// 8 x 8 lateral voxels, but of TEXELS (float4: 6 KB) -- loads every layer ONCE with one coalesced 1 KB read when the march
// reaches it, and serves the corner fetches with ds_read_b128.  Samples whose 2x2x2 footprint is not inside the window
// (oblique tile borders, pixels more than ~0.7 voxel apart) take the global gather: results never depend on the window.
// Interpolation: the same products and FMA order as gather<3,1,1>() -- bit-identical outputs to render_fwd_seg_kernel.
// ------------------------------------------------------------------------------------------------------------------------
#ifndef VOXE_FWD_TILE_DEFER
#define VOXE_FWD_TILE_DEFER 1
#endif
#ifndef VOXE_FWD_TILE_LB
#define VOXE_FWD_TILE_LB 4
#endif
#ifndef VOXE_FWD_TILE_RING
#define VOXE_FWD_TILE_RING 6
#endif
#ifndef VOXE_FWD_TILE_TABLE
#define VOXE_FWD_TILE_TABLE 64
#endif
// ring of 6 layers (6 KB) + the per-layer table (1 KB): measured 0 - 2 % faster than 8 layers + 128 entries (9.5 KB) on 12 camera /
// size pairs.  The kernel is occupancy sensitive (3 KB more LDS per block: +13 %), but its 121 VGPRs hold it at 4 waves per SIMD
// whatever the LDS says; forcing 96 registers spills 51 of them (0.24 -> 0.36 ms).
constexpr int kTexRing = VOXE_FWD_TILE_RING;     // layers of the texel ring; the ring slot of a layer comes from the per-layer table
constexpr int kOrgTable = VOXE_FWD_TILE_TABLE;  // layers tabulated per block: keys key0 .. key0 + 63 (a window tile advances <= 1.7 layers per sample: 32 samples + ring < 64)

// The march of one (tile, depth segment) with the march axis M as a COMPILE-TIME constant (r03): the corner -> (layer, lateral
// offset) mapping, the texel strides and the picks of the cell's (m, u, v) indices fold into immediates.  r02's kernel took
// the axes from registers: 360 VALU instructions per wave-sample against 170 of the ray-ordered forward, and was VALU bound.
template <int M, bool STRATA>
__device__ __forceinline__ void fwd_window_march(const DevGrid& g, const DevCfg& c, const float* __restrict__ packed,
                                                 RayCtx<3, 1, 1>& rc, const int lane, const bool has, const int k_lo,
                                                 const int k_hi, const int kmin, const int kmax, const int ref,
                                                 float4* __restrict__ tex, int4* __restrict__ org,
                                                 const SegDepth<STRATA> sd, float (&csum)[3],
                                                 float& asum, float& dsum, float& T) {
  constexpr int COUT = 3;
  constexpr int U = (M == 0) ? 1 : 0, V = (M == 2) ? 1 : 2;   // lateral axes (v = z whenever m != z: coalesced layer reads)
  constexpr int kCtr = Lat<8>::kCentre;
  // ---- window geometry from the reference ray (as in render_bwd_tile_kernel) ----
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
  const float inv = (DU[M] != 0.0f) ? 1.0f / DU[M] : 0.0f;
  const float Bu = DU[U] * inv, Au = U0[U] - Bu * U0[M];
  const float Bv = DU[V] * inv, Av = U0[V] - Bv * U0[M];
  const int sx = g.Y * g.Z, sy = g.Z;
  const int stride_m = (M == 0) ? sx : ((M == 1) ? sy : 1);
  const int stride_u = (U == 0) ? sx : sy;
  const int stride_v = (V == 1) ? sy : 1;
  const int Nm = N[M], Nu = N[U], Nv = N[V];
  auto minkey = [&](int pm) { return sgn > 0 ? pm : -(pm + 1); };

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
    first_key = minkey(fp_cur.i0[M]);
  }
  const int key0 = wave_min_i32(first_key);
  // lateral origin + voxel offset of that origin for every layer this block can meet, once: lane i -> layers key0 + i, key0 + 64 + i
#pragma unroll
  for (int j = 0; j < kOrgTable / 64; ++j) {
    const int im = sgn * (key0 + j * 64 + lane);
    const int ou = (int)floorf(Au + Bu * (float)im) - kCtr, ov = (int)floorf(Av + Bv * (float)im) - kCtr;
    // (origin u, origin v, voxel offset of the origin -- grids below 2^31 voxels, used only when in range --, ring slot x 64)
    org[j * 64 + lane] = make_int4(ou, ov, im * stride_m + ou * stride_u + ov * stride_v, ((j * 64 + lane) % kTexRing) * 64);
  }
  __syncthreads();
  const int la = lane >> 3, lb8 = lane & 7;
  const int lane_off = la * stride_u + lb8 * stride_v;
  // one coalesced read per layer: lane (a, b) fetches voxel (im, off_u + a, off_v + b)
  auto fetch_layer = [&](int key) -> float4 {
    const int idx = key - key0;                       // wave-uniform
    float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (idx < kOrgTable) {
      const int4 o = org[idx];
      const int vb = o.z;
      const int im = sgn * key;
      if ((unsigned)im < (unsigned)Nm && (unsigned)(o.x + la) < (unsigned)Nu && (unsigned)(o.y + lb8) < (unsigned)Nv)
        t = reinterpret_cast<const float4*>(packed)[vb + lane_off];
    }
    return t;
  };
  auto slot64 = [&](int key) { return ((key - key0) % kTexRing) * 64; };   // wave-uniform (scalar unit); == the table's .w
  auto load_layer = [&](int key) { tex[slot64(key) + lane] = fetch_layer(key); };
  int base = key0;
  int avail = kTexRing;      // layers base .. base + avail - 1 are in LDS (the newest ones of a slide land one sample later)
  float4 pend[2];
  int pend_key = 0, npend = 0;   // wave-uniform
#pragma unroll
  for (int i = 0; i < kTexRing; ++i) load_layer(base + i);
  __syncthreads();
  for (int k = kmin; k <= kmax; ++k) {
    const bool on = has && (k >= k_lo) && (k <= k_hi);
    if (on) {
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
      if (fp.inside) {
        Cell cell;
        make_cell_fast(g, fp, cell);
        const int pm = cell.i[M], pu = cell.i[U], pv = cell.i[V];
        const int kl = minkey(pm);                       // lower key of the footprint's two layers; the other is kl + 1
        const int il = kl - key0;
        const int ilc = min(max(il, 0), kOrgTable - 2);
        const int4 ol = org[ilc], oh = org[ilc + 1];
        const int4 o0 = sgn > 0 ? ol : oh, o1 = sgn > 0 ? oh : ol;   // origins / ring slots of layers pm, pm + 1
        const int a0 = pu - o0.x, b0 = pv - o0.y, a1 = pu - o1.x, b1 = pv - o1.y;
        const bool fits = ((unsigned)(kl - base) < (unsigned)(avail - 1)) && (il == ilc) && ((unsigned)a0 < 7u) &&
                          ((unsigned)b0 < 7u) && ((unsigned)a1 < 7u) && ((unsigned)b1 < 7u);
        float v, rad[COUT];
        if (fits) {
          const float4* __restrict__ t0 = tex + (o0.w + a0 * 8 + b0);
          const float4* __restrict__ t1 = tex + (o1.w + a1 * 8 + b1);
          // gather<3,1,1>() with the eight texels read from the window: corner q = (x + (q & 1), y + ((q >> 1) & 1), z + (q >> 2));
          // the same products and FMA order
          typedef float v2f __attribute__((ext_vector_type(2)));
          float wxy[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) wxy[q] = cell.w[0][q & 1] * cell.w[1][q >> 1];
          float4 t[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int dm = (q >> M) & 1, du = (q >> U) & 1, dv = (q >> V) & 1;   // compile-time after unrolling
            t[q] = (dm ? t1 : t0)[du * 8 + dv];
          }
          v2f rg = {0.0f, 0.0f}, bs = {0.0f, 0.0f};
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float wq = wxy[q & 3] * cell.w[2][q >> 2];
            const v2f ww = {wq, wq};
            const v2f a = {t[q].x, t[q].y}, b = {t[q].z, t[q].w};
            rg = __builtin_elementwise_fma(a, ww, rg);
            bs = __builtin_elementwise_fma(b, ww, bs);
          }
          rad[0] = rc.basis[0] * rg.x; rad[1] = rc.basis[0] * rg.y; rad[2] = rc.basis[0] * bs.x; v = bs.y;
        } else {
          gather<3, 1, 1>(g, packed, cell, rc.basis, v, rad);
        }
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
    // ---- slide the window: bring in the layers the march reaches next ----
    int lb = INT_MAX;
    if (has) {
      if (k + 1 < k_lo) lb = first_key;
      else if (k + 1 <= k_hi) lb = minkey(fp_cur.i0[M]);
    }
    const int newbase = wave_min_i32(lb);
    // the layers fetched at the previous slide have had this sample's time to arrive: into the ring now
    if (npend > 0) {                              // wave-uniform
      __syncthreads();
      tex[slot64(pend_key) + lane] = pend[0];
      if (npend > 1) tex[slot64(pend_key + 1) + lane] = pend[1];
      npend = 0;
      avail = kTexRing;
      __syncthreads();
    }
    if (newbase > base && newbase != INT_MAX) {   // wave-uniform
      const int from = max(base + kTexRing, newbase), need = newbase + kTexRing - from;
      if (need <= 2 && VOXE_FWD_TILE_DEFER) {     // usual case: fetch now, store after the next sample (its footprints are behind these layers)
        pend_key = from;
        npend = need;
        pend[0] = fetch_layer(from);
        if (need > 1) pend[1] = fetch_layer(from + 1);
        avail = from - newbase;
      } else {
        __syncthreads();
        for (int key = from; key < newbase + kTexRing; ++key) load_layer(key);
        __syncthreads();
      }
      base = newbase;
    }
  }
}

__global__ __launch_bounds__(64, VOXE_FWD_TILE_LB) void render_fwd_tile_kernel(DevGrid g, DevCfg c, const float* __restrict__ packed,
                                                                const float* __restrict__ rays_o,
                                                                const float* __restrict__ rays_d,
                                                                const float* __restrict__ jitter,
                                                                float* __restrict__ segbuf, const float fit_lat,
                                                                const float fit_m, const float zdom, const float max_adv) {
  constexpr int COUT = 3, NC = COUT + 3;
  __shared__ float4 tex[kTexRing * 64];
  __shared__ int4 org[kOrgTable];
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
  const bool alive = tile_pixel_ray(c, ty, lane >> 3, (tx << 3) + (lane & 7), 8, r_px);
  const long long r = alive ? r_px : 0;
  RayCtx<3, 1, 1> rc;
  rc.init(g, c, r, rays_o, rays_d, jitter);
  const int ks = seg * c.seg_len, ke = min(c.S, ks + c.seg_len) - 1;
  const int k_lo = max(rc.k_lo, ks);
  const int k_hi = alive ? min(rc.k_hi, ke) : k_lo - 1;
  const bool has = k_lo <= k_hi;
  const int kmin = wave_min_i32(has ? k_lo : INT_MAX);
  const int kmax = wave_max_i32(has ? k_hi : -1);
  float csum[COUT] = {0.0f, 0.0f, 0.0f};
  float asum = 0.0f, dsum = 0.0f, T = 1.0f;
  // Per tile (wave-uniform): march through the texel window, or ray by ray (the loop of render_fwd_seg_kernel)?
  //  * march axis m = the dominant axis of the reference ray -- when that is x or y.  Tiles whose rays run mostly along z march
  //    ray by ray: a layer normal to z is 64 texels in 64 different cache lines (z runs fastest in memory; a layer normal to
  //    x or y is eight 128-byte runs), and marching such a view along x or y instead shears the window by more than a voxel
  //    per layer -- measured (profiles/r03_ab_fwd_window.txt): z-dominant views gain nothing either way, 256^3 / 800x800 loses 9 %;
  //  * the tile has to fit the window -- the measure of the backward's split decision: lateral extent of the 8x8 pixels at the
  //    far end of the segment against the 8-wide layers, spread along m against the ring.  Tiles that do not fit (coarse
  //    images, very oblique tiles) would send most lanes through the global gather AND pay the window.
  int m = -1;   // -1: ray by ray
  int ref = 0;
  if (kmin <= kmax) {
    const unsigned long long hm = __ballot(has);
    ref = ((hm >> 27) & 1ull) ? 27 : (__ffsll((long long)hm) - 1);
    const unsigned long long am = __ballot(alive);
    if ((am >> 1 & 1ull) && (am >> 8 & 1ull)) {
      const int N[3] = {g.X, g.Y, g.Z};
      const float zref = readlane_f32(rc.dg.zlin(ke), 0);
      float ad[3], e3[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float sc = g.scale[a] * 0.5f * (float)N[a];
        const float da = readlane_f32(rc.d[a], 0);
        ad[a] = fabsf(readlane_f32(rc.d[a], ref) * sc);
        e3[a] = 7.0f * (fabsf((readlane_f32(rc.d[a], 1) - da) * sc * zref) + fabsf((readlane_f32(rc.d[a], 8) - da) * sc * zref));
      }
      const int mxy = ad[0] >= ad[1] ? 0 : 1;
      const int mm = (ad[2] >= fabsf(zdom) * ad[mxy]) ? 2 : mxy;
      float lat = 0.0f, alongm = 0.0f;
#pragma unroll
      for (int a = 0; a < 3; ++a) { if (a == mm) alongm = e3[a]; else lat = fmaxf(lat, e3[a]); }
      // layers the march advances per sample (coarse sampling: every sample would wait for 2 - 3 new layers -- S = 128 on 160^3
      // 1.5 - 1.9 layers per sample: 3 - 6 % slower through the window; S = 192: 1.0 - 1.3, 3 - 8 % faster; S = 512 on 256^3: 19 % faster;
      // 256^3 / 800x800 / S = 256 advances 1.55 and is still 9 % faster -- finer pixels, more rays per texel: the bound is 1.7)
      const float adv = ad[mm] * fabsf(readlane_f32(rc.dg.zlin(ke) - rc.dg.zlin(ke > 0 ? ke - 1 : 0), ref));
      if (lat <= fit_lat && alongm <= fit_m && adv <= max_adv && (mm != 2 || zdom < 0.0f)) m = mm;   // (zdom < 0: experiment, march along z too)
    }
  }
  // r04: the strata of this depth segment, tabulated once per block (SegDepth; not with per-ray AABB bounds)
  __shared__ float2 strat[64];
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
        gather<3, 1, 1>(g, packed, cell, rc.basis, v, rad);
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
    else if (m == 0) fwd_window_march<0, ST>(g, c, packed, rc, lane, has, k_lo, k_hi, kmin, kmax, ref, tex, org, sd, csum, asum, dsum, T);
    else if (m == 1) fwd_window_march<1, ST>(g, c, packed, rc, lane, has, k_lo, k_hi, kmin, kmax, ref, tex, org, sd, csum, asum, dsum, T);
    else fwd_window_march<2, ST>(g, c, packed, rc, lane, has, k_lo, k_hi, kmin, kmax, ref, tex, org, sd, csum, asum, dsum, T);
  };
  if (use_strata) march_tile(std::true_type{});
  else march_tile(std::false_type{});
  if (!alive) return;
  const long long base = (long long)seg * NC;
  segbuf[(base + 0) * c.R + r] = T;
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) segbuf[(base + 1 + ch) * c.R + r] = csum[ch];
  segbuf[(base + 1 + COUT) * c.R + r] = asum;
  segbuf[(base + 2 + COUT) * c.R + r] = dsum;
}

// The window forward is the default for image-ordered SH-0 renders since r03 (VoxeDispatch::fwd_window = -1 switches it off):
// r02's version (march axis in registers, 360 VALU instructions per wave-sample) was 12 % slower than the ray-ordered forward;
// with the march axis as a template parameter, layer origins / voxel offsets tabulated once per block, the stores of a slide
// deferred by one sample and a per-tile choice between window and ray-by-ray march it is 8 - 20 % faster for views that run
// along x or y (profiles/r03_ab_fwd_window.txt) and equal otherwise.  Outputs are bit-identical either way
// (tests/test_hip_configs.py::test_lds_staged_forward_*).
bool fwd_tile_supported(const DevGrid& g, const HostCfg& c, int cout, int ncm) {
  (void)g;
  if (cout != 3 || ncm != 1 || c.image_width <= 0) return false;
  return c.disp.fwd_window >= 0;
}
void launch_fwd_tile(const DevGrid& g, const HostCfg& c, const FwdArgs& a, hipStream_t st) {
  const int nseg = num_segments(c.S, c.seg_len);
  const int nb = blocks_for_tiles(c.map_mode, (c.image_width + 7) / 8, tile_rows_total(c, 8)) * nseg;
  const float fit_lat = disp_or(c.disp.fwd_fit_lat, 5.5f), fit_m = disp_or(c.disp.fwd_fit_m, 4.5f);
  const float zdom = disp_or(c.disp.fwd_zdom, 1.0f), max_adv = disp_or(c.disp.fwd_max_adv, 1.7f);
  render_fwd_tile_kernel<<<nb, 64, 0, st>>>(g, c, a.packed, a.rays_o, a.rays_d, a.jitter, a.segbuf, fit_lat, fit_m, zdom, max_adv);
}

// Images of a few thousand rays leave the chip empty whatever the kernel and usually have pixels far apart (little to
// combine in LDS): the depth-segmented line-dense scatter is faster there (64x64: 0.18 vs 0.40 ms; 100x100: 0.40 vs 0.31 ms).
// VoxeDispatch::tile_min_rays overrides the threshold (parity tests ask for -1 so that small images exercise this kernel).
bool tile_bwd_supported(const HostCfg& c, int deg) {
  (void)deg;   // every SH degree: view-dependent grids run their gradient channels as groups of 4 (sibling blocks)
  return c.image_width > 0 && c.R >= disp_tile_min_rays(c.disp);
}

// ---- deterministic mode: scales from the measured maxima, fixed-point -> float ---------------------------------------
__global__ void det_scale_kernel(float* __restrict__ det_scale) {
  // det_scale[0..1]: max |contribution| (features, density); -> det_scale[2..3]: 2^(38 - e) with max < 2^e, so that a
  // scaled contribution is below 2^38 and 2^24 of them still fit the 64-bit sum
  for (int i = 0; i < 2; ++i) {
    int e = 0;
    const float m = det_scale[i];
    if (m > 0.0f && m < 3.0e38f) (void)frexpf(m, &e);
    det_scale[2 + i] = ldexpf(1.0f, 38 - e);
  }
}
__global__ __launch_bounds__(256) void det_finalize_kernel(unsigned long long* __restrict__ gdet, float* __restrict__ gpacked,
                                                           long long n, int CM, const float* __restrict__ det_scale) {
  const double inv_f = 1.0 / (double)det_scale[2], inv_d = 1.0 / (double)det_scale[3];   // exact (powers of two)
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const long long q = (long long)gdet[i];
    if (q != 0) {
      gpacked[i] += (float)((double)q * ((int)(i % CM) == CM - 1 ? inv_d : inv_f));
      gdet[i] = 0ull;   // ready for the next call
    }
  }
}
size_t det_bytes(long long nvox, int C) { return (((size_t)nvox * C * 8 + 255) / 256 + 1) * 256; }
bool det_bwd_supported(const DevCfg& c, int deg, int diffuse) {
  return c.image_width > 0 && (c.attn || deg == 0 || diffuse);   // one channel group, image-ordered rays
}

// grid side / image width from which on the backward takes the 10-wide lateral window (pixels ~0.6 voxel apart and more)
constexpr float kWideWindowRatio = 0.58f;

template <int COUT, int NCM, int NCU>
static void launch_bwd_tile_t(const DevGrid& g, const HostCfg& c, const BwdArgs& a, hipStream_t st) {
  const long long ntx8 = (c.image_width + 7) / 8, nty8 = tile_rows_total(c, 8);
  // The parts (halves / quadrants) of a tile run as sibling blocks instead of consecutive passes while the launch is
  // small enough for the extra blocks to pay off (LDS bounds residency at 9 blocks per CU, 2304 on the chip; siblings
  // of a tile that fits the window whole retire at once); measured cross-over on MI355X with the segment-major block
  // order: 15 % better at 266x266 (9248 tile-segments), equal at 320x320 (12800), 3 % worse at 400x400, 8 % at 800x800
  const int env_q = c.disp.tile_qsplit;
  constexpr int NG = COUT * NCU + 1, WC = NG < 4 ? NG : 4, NGRP = (NG + WC - 1) / WC;
  // channel groups: all of them for a feature gradient; only the one holding the density channel (the last) otherwise
  const int grp_begin = a.want_f ? 0 : NGRP - 1, ngrp = a.want_f ? NGRP : 1;
  const long long tiles = ntx8 * nty8 * num_segments(c.S, c.seg_len) * ngrp;
  const int qsplit = env_q ? (env_q == 4 ? 4 : 1) : (tiles <= 11000 ? 4 : 1);
  const int nb = blocks_for_tiles(c.map_mode, ntx8, nty8) * num_segments(c.S, c.seg_len) * qsplit * ngrp;
  // Bound on a pass's spread along the march axis (layers; the kernel's split decision).  5.5 = the r02 behaviour (never split
  // because of the ring).  Swept over five cameras (tools/ab_cam_lib.sh, profiles/r03_ab_fit_m.txt): while the parts of a tile
  // run as sibling blocks on an under-filled chip, splitting oblique tiles beats their ring overflows by up to 1.7x (4.0
  // best at 266 px, 4.5 - 5.0 at 128 - 200 px in the 10-wide window); mid-size launches 4.5; a full chip (400x400: 20 000
  // tile-segments) prefers the overflow of ~1 % of the samples to 1.5x the blocks.
  const float env_fit_m = c.disp.tile_fit_m;
  const long long tile_segs = ntx8 * nty8 * num_segments(c.S, c.seg_len);
  const int side_for_kl = g.X > g.Y ? (g.X > g.Z ? g.X : g.Z) : (g.Y > g.Z ? g.Y : g.Z);
  const bool wide = (float)side_for_kl >= kWideWindowRatio * (float)c.image_width;
  const float env_fit_lat = c.disp.tile_fit_lat;
  const float fit_lat = env_fit_lat;   // (0: the kernel's default, KL - 2.5 voxels)
  const float fit_m = env_fit_m > 0.0f ? env_fit_m : (qsplit == 4 ? (wide ? 4.5f : 4.0f) : (tile_segs <= 16000 ? 4.5f : 5.5f));
#define VOXE_TBWD(WD, WF, MODE, KL, NB, GB, NGR)                                                 \
  render_bwd_tile_kernel<COUT, NCM, NCU, WD, WF, MODE, KL><<<NB, 64, 0, st>>>(                    \
      g, c, a.packed, a.rays_o, a.rays_d, a.jitter, a.colour, a.depth, a.acc, a.d_colour,         \
      a.d_depth, a.d_acc, a.ray_state, a.gpacked, qsplit, GB, NGR, reinterpret_cast<float4*>(a.sample_src), nullptr, nullptr, 0, fit_m, fit_lat, \
      reinterpret_cast<const float4*>(a.sample_fwd))
  if constexpr (NGRP == 1) {
    if (a.gdet) {   // deterministic mode: measure the maxima, derive the scales, deposit in fixed point, convert
      const long long n = (long long)g.X * g.Y * g.Z * (COUT * NCM + 1);
      (void)hipMemsetAsync(a.det_scale, 0, 4 * sizeof(float), st);
      for (int phase = 0; phase < 2; ++phase) {
        render_bwd_tile_kernel<COUT, NCM, NCU, true, true, 0, 8, true><<<nb, 64, 0, st>>>(
            g, c, a.packed, a.rays_o, a.rays_d, a.jitter, a.colour, a.depth, a.acc, a.d_colour, a.d_depth, a.d_acc,
            a.ray_state, a.gpacked, qsplit, 0, 1, nullptr, a.gdet, a.det_scale, phase, fit_m, fit_lat);
        if (phase == 0) det_scale_kernel<<<1, 1, 0, st>>>(a.det_scale);
      }
      det_finalize_kernel<<<4096, 256, 0, st>>>(a.gdet, a.gpacked, n, COUT * NCM + 1, a.det_scale);
      return;
    }
  }
  if constexpr (NGRP > 1) {
    if (a.sample_src && a.want_f) {   // two-phase: march once for the per-sample sources, then one deposit block per group
      if (a.want_d) VOXE_TBWD(true, true, 1, 8, nb / ngrp, 0, 1);
      else VOXE_TBWD(false, true, 1, 8, nb / ngrp, 0, 1);
      if (NCM == NCU && tile4_dep_supported(g, c, a, NCU, COUT * NCM + 1))   // r05: the lean kernel's deposit passes
        launch_bwd_tile4_dep(g, c, a, NCU, nb, qsplit, ngrp, fit_m, fit_lat, st);
      else
        VOXE_TBWD(true, true, 2, 8, nb, 0, ngrp);
      return;
    }
    if (a.want_d && a.want_f) VOXE_TBWD(true, true, 0, 8, nb, grp_begin, ngrp);
    else if (a.want_d) VOXE_TBWD(true, false, 0, 8, nb, grp_begin, ngrp);
    else VOXE_TBWD(false, true, 0, 8, nb, grp_begin, ngrp);
  } else {
    // lateral window edge: pixels that are far apart (in voxels) need a wider window to keep larger parts of a tile in one
    // pass.  The pixel spacing is not known on the host; for a camera that frames the volume it is ~ grid side / image
    // width.  Measured on MI355X (160^3, backward ms, window 8 / 10 / 12): 400 px 0.58 / 0.78 / 1.06, 266 px 0.38 / 0.40 /
    // 0.55, 200 px 0.36 / 0.31 / 0.36, 100 px 0.30 / 0.26 / 0.26 -- the wider window costs residency (19.4 KB of LDS
    // per block) and only pays once most tiles of the 8-wide window would run as halves or quadrants.
    const int env_kl = c.disp.tile_kl;                 // 8 | 10 overrides the choice
    const int side = g.X > g.Y ? (g.X > g.Z ? g.X : g.Z) : (g.Y > g.Z ? g.Y : g.Z);
    // r04 (parity-class banked windows; profiles/r04_small_image_sweep.txt): 266 px on 160^3 (ratio 0.60) now prefers the
    // 10-wide window too (backward 0.353 -> 0.344 ms); 400 px (0.40) keeps 8
    const int kl = env_kl ? env_kl : ((float)side >= kWideWindowRatio * (float)c.image_width ? 10 : 8);
#define VOXE_TBWD_KL(KL)                                                       \
    do {                                                                       \
      if (a.want_d && a.want_f) VOXE_TBWD(true, true, 0, KL, nb, 0, 1);         \
      else if (a.want_d) VOXE_TBWD(true, false, 0, KL, nb, 0, 1);               \
      else VOXE_TBWD(false, true, 0, KL, nb, 0, 1);                             \
    } while (0)
    if constexpr (COUT == 3 && NCM == 1 && NCU == 1) {
      if (tile4_bwd_supported(g, c, a, kl)) {   // r05: the lean kernel (voxe_render_tile4.hip), same launch geometry
        launch_bwd_tile4(g, c, a, kl, nb, qsplit, fit_m, fit_lat, st);
        return;
      }
    }
    if (kl == 10) VOXE_TBWD_KL(10);
#ifdef VOXE_TILE_KL_EXPERIMENTS   // (A/B builds only: every extra window width is three more ~1 k-line kernels per grid kind)
    else if (kl == 9 && COUT == 3 && NCM == 1 && a.want_d && a.want_f) VOXE_TBWD(true, true, 0, 9, nb, 0, 1);
    else if (kl == 12 && COUT == 3 && NCM == 1 && a.want_d && a.want_f) VOXE_TBWD(true, true, 0, 12, nb, 0, 1);
    else if (kl == 16 && COUT == 3 && NCM == 1 && a.want_d && a.want_f) VOXE_TBWD(true, true, 0, 16, nb, 0, 1);
#endif
    else VOXE_TBWD_KL(8);
#undef VOXE_TBWD_KL
  }
#undef VOXE_TBWD
}

size_t tile_src_bytes(long long R, int W, int H1, int S, int deg, int diffuse, int attn) {
  if (W <= 0 || R <= 0 || deg <= 0 || diffuse || attn) return 0;   // single-group renders do not use it
  const long long H = H1 > 0 ? H1 : R / W, nimg = R / (H * W);
  const long long tiles = ((W + 7) / 8) * ((H + 7) / 8) * (nimg > 0 ? nimg : 1);
  const int seg = seg_len_for(R);
  const size_t bytes = (size_t)tiles * num_segments(S, seg) * seg * 64 * sizeof(float4);
  return bytes <= ((size_t)4 << 30) ? bytes : 0;   // above 4 GB the single-kernel groups run instead
}

void launch_bwd_tile(const DevGrid& g, const HostCfg& c, int deg, int diffuse, const BwdArgs& a, hipStream_t st) {
  if (c.attn) launch_bwd_tile_t<1, 1, 1>(g, c, a, st);
  else if (deg == 0) launch_bwd_tile_t<3, 1, 1>(g, c, a, st);
  else if (deg == 1) { if (diffuse) launch_bwd_tile_t<3, 4, 1>(g, c, a, st); else launch_bwd_tile_t<3, 4, 4>(g, c, a, st); }
  else if (deg == 2) { if (diffuse) launch_bwd_tile_t<3, 9, 1>(g, c, a, st); else launch_bwd_tile_t<3, 9, 9>(g, c, a, st); }
  else { if (diffuse) launch_bwd_tile_t<3, 16, 1>(g, c, a, st); else launch_bwd_tile_t<3, 16, 16>(g, c, a, st); }
}

}  // namespace voxe
