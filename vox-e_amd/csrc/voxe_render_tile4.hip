// voxe_render_tile4.hip -- the LDS-window backward of image-ordered SH-0 renders, rebuilt around its instruction count (r05).
//
// Same algorithm, same window and same results as render_bwd_tile_kernel<3, 1, 1, ., ., 0, KL> (voxe_render_tile.hip: one wave =
// one 8x8-pixel tile x one depth segment, marching in lock step; the tile's gradient is combined in a sliding ring of voxel
// layers held as doubles in LDS -- parity-class banked, WinMap -- and flushed with 16-byte-dense float atomics).  That kernel is
// VALU-issue bound (PMC, profiles/r04_pmc_summary.json: ~590 VALU instructions per wave-sample, 0.85 of the launch), and a third
// of those instructions were bookkeeping the general kernel cannot shed: 106 SGPRs wanted of 102 (v_readlane / v_writelane /
// s_nop spill traffic in the loop), 64-bit address arithmetic, per-sample float math for the lateral origin of two window
// layers, a per-lane modulo for the ring slot, run-time channel-group tables of the view-dependent variants.  This file holds
// ONE configuration -- 4-channel texels, one channel group, float atomics, in-kernel jitter, launch-wide (near, far) -- written
// so that what the loop touches is small:
//   * depth strata (DepthGen's lower / span of sample k) live in two VGPRs, lane l = sample ks + l; all lanes are at the same k,
//     so a sample's stratum is two v_readlane into SGPRs -- no zlin() chain per sample, no LDS;
//   * the window's per-layer data (lateral origin, ring-slot term of the LDS address) is tabulated once per pass, 8 bytes per
//     layer key, 64 keys (512 B: window + table = 12 800 B = ten 1 280-byte LDS granules, 12 one-wave blocks per CU as before);
//   * texel gathers and the flush's atomics address memory as (SGPR base + 32-bit VGPR offset + immediate): the z-neighbour of a
//     corner is an immediate, no 64-bit vector arithmetic anywhere (grids below 2 GiB of packed texels; the host checks);
//   * the flush reads AND clears a window voxel with one ds_wrxchg_rtn_b64, its per-lane LDS / voxel offsets are constants of the
//     pass, the layer's origin voxel is folded into the scalar base pointer: 2 VALU instructions per 16 voxels x 4 channels;
//   * the deposit products are formed in double from 4 + 8 converted operands (32 v_mul_f64, exact) instead of 32 float products
//     + 32 conversions; the parity-class corner order costs ~36 integer instructions per sample (layer roles from the parity of
//     the cell's march index, so the table is read per role and nothing but the two march weights is swapped);
//   * samples whose footprint is not inside the window go straight to global float atomics (rare; results never depend on it).
// Everything else (footprint / cell / interpolation / activations / gradient formulas, block order, tile split, window
// geometry, depth-segment states) is the shared device code of voxe_device.hpp / voxe_render_common.hpp, so sample positions,
// voxel indices and masks stay bit-identical to the oracle.  Reference arithmetic: thre3d_atom/rendering/volumetric/
// accumulate.py:49-84 (alpha, transmittance, weights), process.py:45-84, voxels.py:287-332 (trilinear sample), sample.py:44-67.
#include <limits.h>

#include <type_traits>

#include "voxe_device.hpp"
#include "voxe_launch.hpp"
#include "voxe_render_common.hpp"
#include "voxe_tile_window.hpp"

namespace voxe {

#ifndef VOXE_TILE4_LB
#define VOXE_TILE4_LB 3
#endif
#ifndef VOXE_T4_EXP
#define VOXE_T4_EXP 0     // timing experiments, WRONG RESULTS by construction (tools/variants.py): 1 every lane gathers the same texels |
                          // 2 no LDS adds (products kept) | 4 no flush | 8 flush without the global atomics | 16 no deposit at all
#endif
#ifndef VOXE_T4_SKEW
#define VOXE_T4_SKEW 0    // r06: per-lane sample shift of oblique tiles (bwd4_march, LK).  1 = every oblique tile in ONE skewed pass: measured, NOT
                          // shipped -- the extra march variant pushes the whole kernel over its register budget (profiles/r06_phases_skew.txt);
                          // 2 = only inside the phased passes of tiles that split for their march extent (no new march variant)
#endif
#ifndef VOXE_T4_CHJ
#define VOXE_T4_CHJ 1     // r06: b-parity bit folded into the b term per sample (4 lane constants instead of 8)
#endif
#ifndef VOXE_T4_PHASE_MARGIN
#define VOXE_T4_PHASE_MARGIN 0.0f   // layers added to fit_m in the "do this tile's parts leave room for sample phases" test
#endif
#ifndef VOXE_T4_REMAT
#define VOXE_T4_REMAT 0   // r06 experiment: the deposit's lane constants re-derived from the lane index per sample instead of living in ~14 VGPRs
#endif
#ifndef VOXE_T4_DEBUG
#define VOXE_T4_DEBUG 0   // debugging builds (tools/variants.py): 1 every sample through the per-corner path | 2 ... and no corner in the window
#endif

struct Tile4Args {
  const float *packed, *rays_o, *rays_d, *colour, *depth, *acc, *d_colour, *d_depth, *d_acc, *ray_state;
  float* gpacked;
  int qsplit;
  float fit_m, fit_lat;
  int want_d, want_f;
  int phases;             // VoxeDispatch::tile_phases >= 0: parts of a split tile run 2 / 4 sample phases per ray (r06)
  const double* segsum;   // PREC kernels: the forward's segment-local sums in double (FwdArgs::segsum_d)
  // deposit passes of view-dependent grids (DEP kernels): the per-sample gradient sources of the source pass, channel groups of
  // the launch (group g = texel channels 4 g .. 4 g + 3), gradient channels in all, voxels of the grid.  `gpacked` is then the
  // GROUP-PLANAR staging gradient [group][voxel][4]: a group's flush is as line-dense as the SH-0 kernel's (a z-run of 8 voxels =
  // one 128-byte line; flushed straight into 112-byte texels the same run is 7 lines = 7 atomic requests, and the memory side
  // retires requests -- 0.45 instead of 0.31 ms per pass); planar_to_packed_kernel adds the planes into the packed gradient.
  const float4* sample_src;
  int ngrp, ng;
  long long nvox;
};

// what a deposit pass multiplies the sample's sources with (this lane's ray, this block's channel group):
// window channel s = A * mA[s] + B * mB[s] with A / B two of (d rad_0, d rad_1, d rad_2, d v) -- a group of four consecutive
// gradient channels spans at most two colours, or a colour and the density
struct DepCtx {
  float mA[4], mB[4];
  int cA, cB;            // wave-uniform: source index of A / B
  long long src_base;    // slot of (tile, segment, sample 0, lane) in the source buffer (render_bwd_tile_kernel's layout)
};

constexpr int kTabKeys = 64;     // layer keys tabulated per pass (re-based when the window has moved 32 layers)

// byte strides of the parity-class banked window (WinMap<KL, 4>): index (doubles) = kSS (slot >> 1) + 4 (slot & 1) + kSA (a >> 1)
// + 2 (a & 1) + kSB (b >> 1) + (b & 1) + 8 ch
template <int KL>
struct Pcb {
  static_assert(WinMap<KL, 4>::kPcb, "the lean kernel needs the parity-class banked window");
  static constexpr int SB = WinMap<KL, 4>::kSB * 8, SA = WinMap<KL, 4>::kSA * 8, SS = WinMap<KL, 4>::kSS * 8;
  static __device__ __forceinline__ int mterm(int slot) { return (slot >> 1) * SS + ((slot & 1) << 5); }
  // byte term of lateral coordinate x (already made even: x & ~1) along a / b
  static __device__ __forceinline__ int aterm(int x_even) { return (x_even >> 1) * SA; }
  static __device__ __forceinline__ int bterm(int x_even) { return (x_even >> 1) * SB; }
};

// wave-uniform geometry of a pass (not used inside the sample loop: the loop reads the table)
struct Geo4 {
  float Au, Bu, Av, Bv;
  int sgn;
};

__device__ __forceinline__ int rfl(int x) { return __builtin_amdgcn_readfirstlane(x); }
// a * b + c on the 24-bit multiplier (full rate; the compiler turns __umul24(a, b) + c into the quarter-rate v_mad_u64_u32);
// b wave-uniform
__device__ __forceinline__ unsigned mad24(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
  return r;
}
// a wave-uniform GLOBAL byte pointer the compiler has to keep in SGPRs (so that p + zext(32-bit lane offset) becomes the
// `saddr + voffset` form of the global instructions instead of 64-bit vector arithmetic).  The explicit address space matters:
// an integer that went through a loop-carried variable comes back as a generic pointer, i.e. FLAT instructions, whose
// out-of-order lgkmcnt forces s_waitcnt lgkmcnt(0) in front of every LDS result of the loop.
typedef __attribute__((address_space(1))) char gchar;
typedef __attribute__((address_space(1))) float gfloat;
__device__ __forceinline__ gchar* scalar_ptr(unsigned long long v) {
  // (readfirstlane: a no-op for a value the compiler already holds in SGPRs; where its uniformity analysis gave up on a
  //  loop-carried scalar it is what moves the value there)
  unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  asm volatile("" : "+s"(lo), "+s"(hi));
  return reinterpret_cast<gchar*>(((unsigned long long)hi << 32) | lo);
}
// float atomic add (no return) at scalar base + 32-bit lane offset.  The compiler has to see the instruction (an inline-asm
// atomic is invisible to its wait-count pass, which then waits for MORE than the loads it means).  The lane offset is made
// opaque at every use: once its zero extension is hoisted out of the loop as a 64-bit pair the instruction selector falls
// back to v_lshl_add_u64 + a 64-bit vector address instead of `saddr + voffset`.
__device__ __forceinline__ void global_add_f32(gchar* base, unsigned voff, float x) {
  asm volatile("" : "+v"(voff));
  __hip_atomic_fetch_add(reinterpret_cast<gfloat*>(base + (size_t)voff), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wave-wide integer min with the DPP operand folded into v_min_i32 (wave_min_i32 of voxe_tile_window.hpp compiles to
// v_mov_b32_dpp + v_min_i32 + s_nop per step); all 64 lanes active
__device__ __forceinline__ int wave_min_dpp(int v) {
  asm volatile(
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return min(min(a, b), min(c, d));
}

// Orientation of a tile in its wave (wave-uniform): do the 8 consecutive lanes of a row run along the image rows (false) or down
// the image columns (true)?  The texel gathers are bound by the texture-address / L1 path (~45 clk per divergent 16-byte wave
// load; profiles/r05_lean_experiments.txt), and what that path sees is how many cache lines a quad / a row of lanes touches: z
// runs fastest in memory, so consecutive lanes should step along the image axis whose world step has the larger z share (8 pixels
// x ~0.4 voxel along z = one 128-byte line instead of eight).  Same forward, bit for bit (results are per ray); the backward's
// parity-class deposit is conflict free for ANY lane -> pixel map.  d0 / dx / dy: directions of pixels (0, 0), (1, 0), (0, 1).
__device__ __forceinline__ bool tile_lanes_down_columns(const DevGrid& g, const float (&d0)[3], const float (&dx)[3], const float (&dy)[3]) {
  float ex[3], ey[3];
  const int N[3] = {g.X, g.Y, g.Z};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float s = g.scale[a] * (float)N[a];
    ex[a] = (dx[a] - d0[a]) * s;
    ey[a] = (dy[a] - d0[a]) * s;
  }
  const float lat_x = ex[0] * ex[0] + ex[1] * ex[1], lat_y = ey[0] * ey[0] + ey[1] * ey[1];
  // the z share of the column step beats the row step's:  ey.z^2 / lat_y > ex.z^2 / lat_x
  return (ey[2] * ey[2]) * lat_x > (ex[2] * ex[2]) * lat_y;
}

// One pass of one (tile, depth segment) with the march axis MA as a compile-time constant.
// PREC (VoxeDispatch::precise_grad): the suffix sum behind sample k, sum_{j > k} dL/dw_j w_j, is what remains of the segment's
// LOCAL sums (double, from the forward: the same float products, subtracted here in the same order -- exact to 1e-16 of the
// segment's sum, whatever the transmittance does inside the segment) scaled by the transmittance at the segment start, plus
// the saved suffix of the NEXT boundary (summed back to front by the combine pass: accurate relative to itself).  The default
// path takes (suffix at the segment start) - (float running sum): relative error 1e-7 / (T_k / T_start), i.e. the density
// gradients of samples behind a dense stretch of the same segment are off by 1e-5 (median, profiles/r05_band_probe.txt).
// DEP (deposit pass of a view-dependent grid, r05): no gather, no activations, no ray state -- the sample's four gradient sources
// come from the source pass's buffer (16 bytes per sample, coalesced), times this ray's basis values for the block's channel
// group; the window, its flush and the march are the SH-0 kernel's with the texel stride of the wide grid (`a.tex_bytes`).
// NP (r06): SAMPLE PHASES per ray.  A tile whose footprint outgrows the window runs as 2 or 4 parts of 32 / 16 rays; with NP = 1
// the other lanes of those passes idle (every VALU / LDS instruction of the pass costs what a full wave's costs).  With NP = 2 / 4
// the lanes outside the part take the other phases of the part's rays: lane (ray, phase p) is at sample kb + p while the wave is
// at kb, kb advances by NP.  Everything per sample is per lane as before (depth, footprint, gather, activation, deposit); what
// chains the samples of a ray -- the transmittance T and the running sum of dL/dw w -- is carried per ray as the value IN
// FRONT OF the wave's NP samples, and the NP lanes of a ray exchange (1 - alpha) and dL/dw w through the LDS crossbar
// (ds_bpermute: xor masks xm1 / xm2 of the part's lane bits) and apply them in sample order -- the products and sums of the
// one-sample-per-iteration march, in the same order.
// LK (r06, "skew"): per-LANE sample index.  The lanes of a wave need not be at the same sample of their rays: nothing couples
// different rays but the window, and the window wants the lanes in the same LAYERS.  A tile seen obliquely puts its rays several
// layers apart at equal sample index (the reason such tiles ran as two passes of 32 lanes: "alongm > fit_m"); with a per-lane
// shift `sh` -- the layers a ray is ahead of the pass's reference ray, rounded to samples -- lane l works on sample kb + koff,
// koff = phase - sh, and the wave's footprint along the march axis is that of ONE ray (+ rounding) whatever the view.  The
// strata then come through the LDS crossbar (the stratum index differs from lane to lane: no v_readlane).
template <int MA, int KL, bool PREC, bool DEP, int NP = 1, bool LK = (NP > 1)>
__device__ __forceinline__ void bwd4_march(const DevGrid& g, const DevCfg& c, const Tile4Args& a, RayCtx<3, 1, 1>& rc,
                                           double* __restrict__ win, int2* __restrict__ tab, const int lane, const long long r,
                                           bool has, const int k_lo, int k_hi, const int kmin, const int kmax, const int seg,
                                           const int ks, const Geo4 geo, const float strat_lo, const float strat_sp,
                                           const DepCtx& dep, const int phase = 0, const int xm1 = 0, const int xm2 = 0, const int koff = 0) {
  static_assert(!LK || (!PREC && !DEP), "per-lane sample indices (phases, skew): the SH-0 float kernel only");
  static_assert(NP == 1 || LK, "sample phases need the per-lane sample index");
  constexpr int UA = (MA == 0) ? 1 : 0, VA = (MA == 2) ? 1 : 2;
  constexpr int COUT = 3;
  constexpr int kCtr = Lat<KL>::kCentre;
  typedef Pcb<KL> P;
  static_assert(!(DEP && PREC), "deposit passes carry no suffix sums");
  constexpr unsigned TB = 16u;   // bytes of a texel of the gradient the window is flushed into (DEP: the group's plane, see DepCtx)

  // ---- per-ray constants of the backward (render_bwd_kernel, voxe_render.hip) ----------------------------------------
  float gc[COUT] = {0.0f, 0.0f, 0.0f}, gsum = 0.0f;
  if constexpr (!DEP) {
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) { gc[ch] = a.d_colour[r * COUT + ch]; gsum += gc[ch]; }
  }
  const float gdep = (!DEP && a.d_depth) ? a.d_depth[r] : 0.0f;
  const float gacc = (!DEP && a.d_acc) ? a.d_acc[r] : 0.0f;
  const bool white = c.white != 0;
  float T = 1.0f, suffix0 = 0.0f;
  if constexpr (!DEP) {
    const float asum = a.acc[r];
    float total = gdep * a.depth[r] + gacc * asum;
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) {
      const float csum = white ? a.colour[r * COUT + ch] - (1.0f - asum) : a.colour[r * COUT + ch];
      total += gc[ch] * csum;
    }
    if (white) total -= gsum * asum;
    suffix0 = total;
    if (seg > 0) {   // suffix sums saved by the forward's combine pass at this segment boundary (no cancellation)
      constexpr int NC = COUT + 3;
      float suf_c[COUT];
      T = a.ray_state[ray_state_index(seg, 0, NC, c.R, r)];
#pragma unroll
      for (int ch = 0; ch < COUT; ++ch) suf_c[ch] = a.ray_state[ray_state_index(seg, 1 + ch, NC, c.R, r)];
      const float suf_a = a.ray_state[ray_state_index(seg, 1 + COUT, NC, c.R, r)];
      const float suf_d = a.ray_state[ray_state_index(seg, 2 + COUT, NC, c.R, r)];
      suffix0 = gdep * suf_d + gacc * suf_a;
#pragma unroll
      for (int ch = 0; ch < COUT; ++ch) suffix0 += gc[ch] * suf_c[ch];
      if (white) suffix0 -= gsum * suf_a;
    }
  }
  float suffix_next = 0.0f, T_start = 1.0f, Tl = 1.0f;
  double rem = 0.0;     // what is left of the segment's local sum of dL/dw_j w_j (double; ONE value: five would not fit the registers)
  if constexpr (PREC) {
    constexpr int NC = COUT + 3;
    const int nseg_p = num_segments(c.S, c.seg_len);
    T_start = T;
    if (seg + 1 < nseg_p) {
      suffix_next = gdep * a.ray_state[ray_state_index(seg + 1, 2 + COUT, NC, c.R, r)] +
                    gacc * a.ray_state[ray_state_index(seg + 1, 1 + COUT, NC, c.R, r)];
#pragma unroll
      for (int ch = 0; ch < COUT; ++ch) suffix_next += gc[ch] * a.ray_state[ray_state_index(seg + 1, 1 + ch, NC, c.R, r)];
      if (white) suffix_next -= gsum * a.ray_state[ray_state_index(seg + 1, 1 + COUT, NC, c.R, r)];
    }
    // the segment's sum of dL/dw_j w_j (local weights) from the forward's component sums, in double; the march subtracts
    // dL/dw_j w_j sample by sample with dL/dw_j ALSO formed in double from the same float inputs: what remains is exact to
    // 1e-16 of the component sums
    const double* const ss = a.segsum + (long long)seg * 5 * c.R + r;
    const double sa = ss[3 * c.R];
    rem = (double)gdep * ss[4 * c.R] + ((double)gacc - (double)(white ? gsum : 0.0f)) * sa;
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) rem = fma((double)gc[ch], ss[(long long)ch * c.R], rem);
    // (the suffix behind the segment rides along in local units: suffix_k = T_start (rem_k + suffix_next / T_start))
    rem += T_start > 0.0f ? (double)suffix_next / (double)T_start : 0.0;
  }
  const float gsumw = white ? gsum : 0.0f;                      // (x - 0 == x: the white-background term without a branch)
  float gcf[COUT];                                               // d rad_c = (w g_c) (col (1 - col)) C0, C0 folded into g_c's copy
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) gcf[ch] = a.want_f ? gc[ch] : 0.0f;
  const float dmask = a.want_d ? 1.0f : 0.0f;
  float run = 0.0f;

  // ---- lane constants of the parity-class banked deposit (WinMap) ----------------------------------------------------
  // instruction (cc, j) of this lane goes to the corner whose window voxel has parity class cc ^ (hm, hu, hv) and to channel
  // (j + crot) & 3: the 32 lanes of a half-wave hit 32 different bank pairs in every instruction
  const int hm = (lane >> 1) & 1, hu = (lane >> 2) & 1, hv = (lane >> 3) & 1;
  const int crot = (lane & 1) | ((lane >> 3) & 2);
  const int k0u = 1 - hu, k1u = hu, k0v = 1 - hv, k1v = hv;      // the corner with parity h is (x + 1 - h) & ~1 | h
  const int KU0 = hu << 4, KU1 = (1 - hu) << 4;                  // byte offset of the a-parity bit
#if VOXE_T4_CHJ
  const int KV0 = hv << 3, KV1 = (1 - hv) << 3;                  // ... of the b-parity bit (r06: folded into the b term per sample --
                                                                 // four adds -- instead of eight lane constants b-parity + channel)
  int CHJ[4];                                                    // channel of instruction j
#pragma unroll
  for (int j = 0; j < 4; ++j) CHJ[j] = ((j + crot) & 3) << 6;
#define VOXE_CH(bv, j) CHJ[j]
#else
  const int KV0 = 0, KV1 = 0;
  int CH[2][4];                                                  // b-parity bit + channel of instruction j
#pragma unroll
  for (int cv = 0; cv < 2; ++cv)
#pragma unroll
    for (int j = 0; j < 4; ++j) CH[cv][j] = ((hv ^ cv) << 3) + (((j + crot) & 3) << 6);
#define VOXE_CH(bv, j) CH[bv][j]
#endif
  const bool c1 = crot & 1, c2 = crot & 2;

  // ---- window geometry -> per-layer table -----------------------------------------------------------------------------
  const int sgn = geo.sgn;
  const int smask = sgn < 0 ? -1 : 0;                            // minkey(pm) = pm ^ smask  (sgn < 0: -(pm + 1) = ~pm)
  const int sx = g.Y * g.Z, sy = g.Z;
  const int stride_m = (MA == 0) ? sx : ((MA == 1) ? sy : 1);
  const int stride_u = (UA == 0) ? sx : sy;
  const int stride_v = (VA == 1) ? sy : 1;
  auto build_tab = [&](int key0) {
    const int key = key0 + lane, im = sgn * key;
    const int ou = (int)floorf(geo.Au + geo.Bu * (float)im) - kCtr, ov = (int)floorf(geo.Av + geo.Bv * (float)im) - kCtr;
    tab[lane] = make_int2((ou & 0xffff) | (ov << 16), P::mterm(ring_slot(key)));
  };

  // depth of sample k: stratum (lower, span) of the segment's table (wave-uniform: every lane is at the same k) + this ray's jitter
  auto depth_of = [&](int k) {
    const int j = k - ks;                                       // wave-uniform
    const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(strat_lo), j));
    const float sp = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(strat_sp), j));
    const float su = sp * jitter_uniform(rc.dg.base, k);
    return lo + su;
  };
  // (LK) depth of this LANE's sample k: the stratum comes through the LDS crossbar (lane j of the wave holds sample ks + j's);
  // call with all 64 lanes active
  auto depth_lane = [&](int k) {
    const int j = (k - ks) & 63;
    const float lo = __shfl(strat_lo, j, 64), sp = __shfl(strat_sp, j, 64);
    const float su = sp * jitter_uniform(rc.dg.base, k);
    return lo + su;
  };
  // first sample of every ray (rolling: z_cur / fp always describe sample max(k, k_lo)); NP > 1: of every lane -- the first
  // sample >= k_lo of this lane's phase (samples kmin + phase, kmin + phase + NP, ...)
  int kf = k_lo;
  int kb0 = kmin, kb1 = kmax;          // range of the wave's loop variable
  if constexpr (LK) {
    // lane l is at sample kb + koff: the loop starts where the first lane has a sample and ends behind the last one's
    kb0 = wave_min_i32(has ? k_lo - koff : INT_MAX);
    kb1 = wave_max_i32(has ? k_hi - koff : -(1 << 30));
    const int m = (kb0 + koff - k_lo) % NP;      // this lane's samples: k = kb0 + koff (mod NP)
    kf = k_lo + (m < 0 ? m + NP : m);
    if (kf > k_hi) has = false;
  }
  float z_cur = 0.0f;
  Footprint fp;
  fp.inside = false;
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) { fp.i0[ax] = 0; fp.w[ax][0] = fp.w[ax][1] = 0.0f; }
  int nextkey = INT_MAX;      // lowest layer key this lane's NEXT sample can write (INT_MAX: no sample left)
  if (has) {   // (the first sample index differs from lane to lane: its stratum is evaluated per lane, DepthGen's expressions)
    const float2 st0 = depth_stratum(rc.dg, kf);
    const float su0 = st0.y * jitter_uniform(rc.dg.base, kf);
    z_cur = st0.x + su0;
    float p[3];
    rc.point(z_cur, p);
    footprint(g, p, fp);
    nextkey = fp.i0[MA] ^ smask;
  }
  int base = wave_min_i32(nextkey);      // lowest live layer key
  int key0 = base;                       // key of table entry 0
  build_tab(key0);
  __syncthreads();                       // window zeroed, table written

  // flush of one layer: 16 voxels x 4 channels per instruction group.  KL == 8: KL / 2 groups of two a-rows each, dealt to the
  // lanes as 2 x 2 blocks (a layer has ONE slot parity: 2-way bank conflicts are the floor); the LDS / voxel offsets of group j
  // are those of group 0 plus j times a constant (an immediate / a scalar add).  Other widths: group j = lateral cells
  // 16 j .. 16 j + 15 in row-major order, per-lane offsets tabulated per pass (the wide window runs at 2 waves per SIMD: the
  // registers are there).
  const int q4 = lane >> 2, ch4 = lane & 3;
  constexpr int NJ = (KL == 8) ? KL / 2 : (KL * KL + 15) / 16;
  int fl_lds[KL == 8 ? 1 : NJ];
  unsigned fl_vox[KL == 8 ? 1 : NJ];
  if constexpr (KL == 8) {
    const int fa = q4 & 1, fb = q4 >> 1;                           // lateral cell (2 j + fa, fb) of this lane in group j
    fl_lds[0] = ((fa << 1) + (fb >> 1) * (P::SB / 8) + (fb & 1) + (ch4 << 3)) * 8;
    fl_vox[0] = (unsigned)(fa * stride_u + fb * stride_v) * TB + (unsigned)(ch4 * 4);
  } else {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int ab = j * 16 + q4, fa = ab / KL, fb = ab - fa * KL;
      const bool cell_live = ab < KL * KL;
      fl_lds[j] = cell_live ? ((fa >> 1) * P::SA + ((fa & 1) << 4) + (fb >> 1) * P::SB + ((fb & 1) << 3) + (ch4 << 6)) : -1;
      fl_vox[j] = (unsigned)(fa * stride_u + fb * stride_v) * TB + (unsigned)(ch4 * 4);
    }
  }
  const unsigned long long gaddr = reinterpret_cast<unsigned long long>(a.gpacked);
  const long long sm16 = (long long)stride_m * TB, su16 = (long long)stride_u * TB, sv16 = (long long)stride_v * TB;   // (bytes per step)
  // The flush of a layer is split in two: flush_issue() reads-and-clears the layer (ds_wrxchg_rtn_b64) and keeps the returned
  // values pending in registers, flush_consume() -- one iteration later, after the next sample's texels are in -- converts them
  // and issues the global atomics (vmcnt counts in order: loads behind an atomic wait for it).
  //
  // Tiles that march along z (kPair): a layer is an (x, y) plane, every one of its voxels sits in its own cache line, and the
  // memory side retires atomic REQUESTS (lines per instruction; profiles/r01_microbench_atomics.md) -- the flush of such a tile
  // costs 4 - 8x the requests of an x / y march, whose 8-voxel b-runs are whole lines (camera 12: 0.135 of 0.57 ms, timing
  // experiment).  Two consecutive layers are z-neighbours, i.e. the SAME lines: the pending layer therefore waits for the next
  // one, layers with an odd key are read with the two b-halves of the lane map swapped, and one atomic instruction carries
  // (8 voxels) x (2 layers) x 4 channels -- 32 contiguous bytes per voxel pair, half the requests.
#ifndef VOXE_T4_ZPAIR
#define VOXE_T4_ZPAIR 1
#endif
  constexpr bool kPair = VOXE_T4_ZPAIR && MA == 2 && KL == 8;
  unsigned long long pend[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) pend[j] = 0ull;
  unsigned long long pend_vb = 0ull;     // scalar: global byte address of the pending layer's origin voxel
  int npend = 0;                         // wave-uniform: pending layers (0 / 1; kPair: up to 2)
  // (kPair) the second pending layer, the keys' parities, this lane's voxel offsets under both lane maps
  unsigned long long pendB[kPair ? NJ : 1];
  unsigned long long pendB_vb = 0ull;
  int oddA = 0, oddB = 0;
  const bool lane_hi = lane >= 32;
  const unsigned step4 = (unsigned)(4 * stride_v) * TB;
  const unsigned vox_b03 = kPair ? fl_vox[0] - (lane_hi ? step4 : 0u) : 0u;   // this lane's voxel with b reduced to 0..3
  auto flush_issue = [&](int key) __attribute__((always_inline)) {      // key wave-uniform
    if (VOXE_T4_EXP & 4) return;
    const int2 e = tab[key - key0];      // (broadcast read)
    const int ex = rfl(e.x), mt = rfl(e.y);
    const int ou = (int)(short)(ex & 0xffff), ov = ex >> 16;
    const int im = sgn * key;
    const unsigned long long vb = gaddr + (unsigned long long)((long long)im * sm16 + (long long)ou * su16 + (long long)ov * sv16);   // scalar
    char* const wb = reinterpret_cast<char*>(win);
    if constexpr (kPair) {
      const int odd = key & 1;
      const int lo = (fl_lds[0] ^ (odd ? 2 * P::SB : 0)) + mt;    // odd keys: b <-> b ^ 4  ((b >> 1) * SB: bit 1 of b >> 1)
      if (npend == 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          pend[j] = __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(wb + lo) + j * (P::SA / 8), 0ull, __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_WORKGROUP);
        pend_vb = vb; oddA = odd;
      } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          pendB[j] = __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(wb + lo) + j * (P::SA / 8), 0ull, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
        pendB_vb = vb; oddB = odd;
      }
      ++npend;
      return;
    }
    pend_vb = vb;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if constexpr (KL == 8) {
        pend[j] = __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(wb + (fl_lds[0] + mt)) + j * (P::SA / 8), 0ull,
                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        pend[j] = 0ull;
        if (fl_lds[j] >= 0)
          pend[j] = __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(wb + (fl_lds[j] + mt)), 0ull, __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    npend = 1;
  };
  // one pending layer on its own (kPair: under the lane map its key's parity chose)
  auto consume_single = [&](const unsigned long long (&pv)[NJ], unsigned long long vb, int odd) __attribute__((always_inline)) {
    const unsigned vox = kPair ? vox_b03 + ((lane_hi != (odd != 0)) ? step4 : 0u) : fl_vox[0];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const double val = __longlong_as_double((long long)pv[j]);
      if (!(VOXE_T4_EXP & 8) && val != 0.0) {
        if constexpr (KL == 8)
          global_add_f32(scalar_ptr(vb + (unsigned long long)((long long)j * 2ll * su16)), vox, (float)val);   // scalar base of group j
        else
          global_add_f32(scalar_ptr(vb), fl_vox[j], (float)val);
      }
    }
  };
  // `force`: the march is over (or a third layer is on its way): whatever is pending goes out
  auto flush_consume = [&](bool force = false) __attribute__((always_inline)) {
    if (npend == 0) return;              // wave-uniform
    if constexpr (kPair) {
      if (npend == 1) {
        if (!force) return;              // wait for the z-neighbour layer
        consume_single(pend, pend_vb, oddA);
        npend = 0;
        return;
      }
      if (oddA == oddB) {                // (not neighbours: the window jumped)
        consume_single(pend, pend_vb, oddA);
        consume_single(pendB, pendB_vb, oddB);
      } else {
        // lanes with m set carry layer A in the instruction of b 0..3 and layer B in the instruction of b 4..7 (the others
        // the other way round): A's lanes of b 0..3 are the low half when its key is even, the high half when it is odd
        const bool m = lane_hi == (oddA != 0);
        const unsigned long long vmin = pend_vb < pendB_vb ? pend_vb : pendB_vb;
        const unsigned dA = (unsigned)(pend_vb - vmin), dB = (unsigned)(pendB_vb - vmin);     // scalar, small
        const unsigned off1 = vox_b03 + (m ? dA : dB), off2 = vox_b03 + step4 + (m ? dB : dA);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const float fA = (float)__longlong_as_double((long long)pend[j]), fB = (float)__longlong_as_double((long long)pendB[j]);
          const float v1 = m ? fA : fB, v2 = m ? fB : fA;
          gchar* const gb = scalar_ptr(vmin + (unsigned long long)((long long)j * 2ll * su16));
          if (!(VOXE_T4_EXP & 8) && v1 != 0.0f) global_add_f32(gb, off1, v1);
          if (!(VOXE_T4_EXP & 8) && v2 != 0.0f) global_add_f32(gb, off2, v2);
        }
      }
      npend = 0;
      return;
    }
    consume_single(pend, pend_vb, 0);
    npend = 0;
  };

  const unsigned sxb = g.X > 1 ? (unsigned)(g.Y * g.Z) * 16u : 0u, syb = g.Y > 1 ? (unsigned)g.Z * 16u : 0u;
  const unsigned szb = g.Z > 1 ? 16u : 0u;
  const unsigned sxi = (unsigned)(g.Y * g.Z) * 16u, syi = (unsigned)g.Z * 16u;   // strides of the cell's low-corner index
  const char* const pbytes = reinterpret_cast<const char*>(a.packed);
  char* const gbytes = reinterpret_cast<char*>(a.gpacked);
  const int Sm1 = c.S - 1;
  const float term_eps = c.term_eps;
  const int nk0 = -key0;   // (re-based with the table)
  int nkey0 = nk0;

  for (int kb = kb0; kb <= kb1; kb += NP) {
    const int k = LK ? kb + koff : kb;
    const bool on = has && (k >= k_lo) && (k <= k_hi);
    const bool live = on && fp.inside;
    float z_n1 = 0.0f, z_nP = 0.0f;      // (NP > 1) depths of this lane's samples k + 1 (the interval) and k + NP (its next sample)
    if constexpr (LK) {
      z_n1 = depth_lane(k + 1);
      z_nP = (NP > 1) ? depth_lane(k + NP) : z_n1;
    }
    // ---- phase 1: the cell of this sample, its 8 texel loads and the two window-table reads go out -- in STRAIGHT-LINE code,
    // executed by every lane (lanes without a sample gather texel 0 and read table entry 0): the compiler's wait-count pass is
    // not path sensitive, loads issued under one `if` and consumed under a later one leave it with "possibly outstanding" at
    // the loop's back edge, which it resolves with s_waitcnt vmcnt(0) in front of the next iteration's loads -- and that wait
    // also covers the flush's global atomics.
    if (live) {
      // the footprint with the zero-padding rule folded in (make_cell), IN PLACE: away from the faces (almost every sample;
      // wave-uniform test) the footprint is the cell, and nothing of it is read again before the next footprint() overwrites it
      if (__builtin_amdgcn_ballot_w64(!cell_is_interior(g, fp)) != 0ull) {
        Cell cf;
        make_cell(g, fp, cf);
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) { fp.i0[ax] = cf.i[ax]; fp.w[ax][0] = cf.w[ax][0]; fp.w[ax][1] = cf.w[ax][1]; }
      }
    }
    Cell cell;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) { cell.i[ax] = fp.i0[ax]; cell.w[ax][0] = fp.w[ax][0]; cell.w[ax][1] = fp.w[ax][1]; }
    // gather: 8 texels as (scalar base + 32-bit offset); the z-neighbour is the immediate
    float4 t[DEP ? 1 : 8];
    if constexpr (DEP) {
      // the sample's sources: lanes without a sample read slot 0 of the block's own rows (always inside the buffer)
      const long long so = live ? dep.src_base + (long long)k * 64 : dep.src_base + (long long)ks * 64;
      t[0] = a.sample_src[so];
    } else {
      unsigned off0 = mad24((unsigned)cell.i[0], sxi, mad24((unsigned)cell.i[1], syi, (unsigned)cell.i[2] << 4));
      if (!live || (VOXE_T4_EXP & 1)) off0 = 0u;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const unsigned o = off0 + ((q & 1) ? sxb : 0u) + ((q & 2) ? syb : 0u);
        t[q] = *reinterpret_cast<const float4*>(pbytes + ((size_t)o + ((q & 4) ? szb : 0u)));
      }
    }
    // layer roles of the deposit: "A" = the layer whose ring slot has parity hm (slot parity == key parity == parity of the
    // march index: the ring depth is even), "B" the other one; their table entries
    const int tm = (cell.i[MA] ^ hm) & 1;
    const int imA = cell.i[MA] + tm, imB = cell.i[MA] + 1 - tm;
    const int relA = ((imA ^ smask) - smask) + nkey0, relB = ((imB ^ smask) - smask) + nkey0;   // sgn * im - key0
    const int2 eA = tab[relA & (kTabKeys - 1)], eB = tab[relB & (kTabKeys - 1)];
    float f0, f1, f2, v;
    if constexpr (DEP) { f0 = t[0].x; f1 = t[0].y; f2 = t[0].z; v = t[0].w; }
    else interp_texels4(t, cell, f0, f1, f2, v);
    asm volatile("" ::"v"(f0), "v"(f1), "v"(f2), "v"(v));   // (consumed HERE, by every lane: see phase 1)
    // ---- the layer flushed at the end of the previous iteration: its values have long arrived.  The global atomics go out
    // AFTER this sample's texels are in: vmcnt counts in order, so loads behind an atomic would wait for it, and atomics issued
    // between the loads and their use make the wait-count pass wait for the atomics as well (they sit in conditional blocks).
    flush_consume();
    // the deposit of one sample (this lane's cell, table entries and gradient channels) into the LDS window; a lambda so that the
    // one-sample march and the phased march (NP > 1) share the text
    auto do_deposit = [&](const float (&gch)[4]) __attribute__((always_inline)) {
#if VOXE_T4_REMAT
      int ln = lane;
      asm volatile("" : "+v"(ln));     // (opaque: the derivations below must not be hoisted out of the sample loop)
      const int hu = (ln >> 2) & 1, hv = (ln >> 3) & 1;
      const int crot = (ln & 1) | ((ln >> 3) & 2);
      const int k0u = 1 - hu, k1u = hu, k0v = 1 - hv, k1v = hv;
      const int KU0 = hu << 4, KU1 = (1 - hu) << 4;
      const int KV0 = hv << 3, KV1 = (1 - hv) << 3;
      int CHJ[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) CHJ[j] = ((j + crot) & 3) << 6;
      const bool c1 = crot & 1, c2 = crot & 2;
#endif
      // ---- the cell in (march, lateral u, lateral v) order ---------------------------------------------------------
      const int pm = cell.i[MA], pu = cell.i[UA], pv = cell.i[VA];
      const float wm0 = cell.w[MA][0], wm1 = cell.w[MA][1];
      const float wu0 = cell.w[UA][0], wu1 = cell.w[UA][1], wv0 = cell.w[VA][0], wv1 = cell.w[VA][1];
      // (the table entries stay packed until HERE: left alone the compiler derives the eight window coordinates right behind
      //  the LDS reads of phase 1 and carries them -- eight registers for two -- across the whole per-sample math)
      int eAx = eA.x, eBx = eB.x;
      asm volatile("" : "+v"(eAx), "+v"(eBx));
      const int ouA = (int)(short)(eAx & 0xffff), ovA = eAx >> 16, ouB = (int)(short)(eBx & 0xffff), ovB = eBx >> 16;
      const int PU0 = pu + k0u, PU1 = pu + k1u, PV0 = pv + k0v, PV1 = pv + k1v;
      const int xA0 = PU0 - ouA, xA1 = PU1 - ouA, yA0 = PV0 - ovA, yA1 = PV1 - ovA;   // {a, a + 1}, {b, b + 1} of layer A
      const int xB0 = PU0 - ouB, xB1 = PU1 - ouB, yB0 = PV0 - ovB, yB1 = PV1 - ovB;
      // window test: both layers inside the ring, every lateral coordinate inside [0, KL)
      const int klrel = min(relA, relB);
      bool fits = (unsigned)(klrel - (base + nkey0)) < (unsigned)(kRing - 1);   // (base - key0 <= 32: both entries are tabulated)
      if constexpr (KL == 8) fits = fits && ((unsigned)(xA0 | xA1 | yA0 | yA1 | xB0 | xB1 | yB0 | yB1) < 8u);
      else fits = fits && (max(max(max((unsigned)xA0, (unsigned)xA1), max((unsigned)yA0, (unsigned)yA1)),
                               max(max((unsigned)xB0, (unsigned)xB1), max((unsigned)yB0, (unsigned)yB1))) < (unsigned)KL);
      if (VOXE_T4_DEBUG & 3) fits = false;
      if (fits) {
        // weights in role order: x0 = weight of the corner with parity h (the low corner iff its coordinate has parity h)
        const float wmA = tm ? wm1 : wm0, wmB = tm ? wm0 : wm1;
        auto pick2 = [](int x1, float w0, float w1, float& o0, float& o1) {   // x1 = coordinate + h: odd <=> the HIGH corner has parity h
          const bool hi = x1 & 1;
          o0 = hi ? w1 : w0;
          o1 = hi ? w0 : w1;
        };
        float wuA[2], wuB[2], wvA[2], wvB[2];
        pick2(xA1, wu0, wu1, wuA[0], wuA[1]);
        pick2(xB1, wu0, wu1, wuB[0], wuB[1]);
        pick2(yA1, wv0, wv1, wvA[0], wvA[1]);
        pick2(yB1, wv0, wv1, wvB[0], wvB[1]);
        const float wmuA[2] = {wmA * wuA[0], wmA * wuA[1]}, wmuB[2] = {wmB * wuB[0], wmB * wuB[1]};
        // byte addresses: slot term (table) + a term + b term + parity bits + channel
        const int MA0 = eA.y + KU0, MA1 = eA.y + KU1, MB0 = eB.y + KU0, MB1 = eB.y + KU1;
        const int muA[2] = {P::aterm(xA0 & ~1) + MA0, P::aterm(xA1 & ~1) + MA1};
        const int muB[2] = {P::aterm(xB0 & ~1) + MB0, P::aterm(xB1 & ~1) + MB1};
        const int bvA[2] = {P::bterm(yA0 & ~1) + KV0, P::bterm(yA1 & ~1) + KV1}, bvB[2] = {P::bterm(yB0 & ~1) + KV0, P::bterm(yB1 & ~1) + KV1};
        // gr[j] = gch[(j + crot) & 3], as doubles
        const float q0 = c1 ? gch[1] : gch[0], q1 = c1 ? gch[2] : gch[1], q2 = c1 ? gch[3] : gch[2], q3 = c1 ? gch[0] : gch[3];
        const double gr[4] = {(double)(c2 ? q2 : q0), (double)(c2 ? q3 : q1), (double)(c2 ? q0 : q2), (double)(c2 ? q1 : q3)};
        char* const wb = reinterpret_cast<char*>(win);
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
          const int bm = cc & 1, bu = (cc >> 1) & 1, bv = cc >> 2;   // compile-time bits of this instruction
          const double wgt = (double)((bm ? wmuB[bu] : wmuA[bu]) * (bm ? wvB[bv] : wvA[bv]));
          const int idx = (bm ? muB[bu] : muA[bu]) + (bm ? bvB[bv] : bvA[bv]);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (VOXE_T4_EXP & 2) { const double pr = gr[j] * wgt; const int ad = idx + VOXE_CH(bv, j); asm volatile("" ::"v"(pr), "v"(ad)); continue; }
            __hip_atomic_fetch_add(reinterpret_cast<double*>(wb + (idx + VOXE_CH(bv, j))), gr[j] * wgt, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      } else {
        // Some corner outside the window (oblique tile borders, ring overflow, grid faces): per corner, in natural order --
        // inside the window the LDS add (bank conflicts do not matter here), else a global float atomic.
        const int brel = base + nkey0;
        const int2 e0 = tm ? eB : eA, e1 = tm ? eA : eB;                    // table entries of layers pm, pm + 1
        const int rel0 = tm ? relB : relA, rel1 = tm ? relA : relB;
        const unsigned vo = (unsigned)(pm * stride_m + pu * stride_u + pv * stride_v) * TB;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
          const int cm = cc & 1, cu = (cc >> 1) & 1, cv = cc >> 2;
          const float wgt = ((cm ? wm1 : wm0) * (cu ? wu1 : wu0)) * (cv ? wv1 : wv0);
          if (wgt != 0.0f) {
            const int2 es = cm ? e1 : e0;
            const int aa = pu + cu - (int)(short)(es.x & 0xffff), bb = pv + cv - (es.x >> 16);
            const bool inwin = !(VOXE_T4_DEBUG & 2) && ((unsigned)((cm ? rel1 : rel0) - brel) < (unsigned)kRing) &&
                               ((unsigned)aa < (unsigned)KL) && ((unsigned)bb < (unsigned)KL);
            if (inwin) {
              double* const wp = reinterpret_cast<double*>(reinterpret_cast<char*>(win) +
                                                           (es.y + (aa >> 1) * P::SA + ((aa & 1) << 4) + (bb >> 1) * P::SB + ((bb & 1) << 3)));
#pragma unroll
              for (int ch = 0; ch < 4; ++ch)
                __hip_atomic_fetch_add(wp + ch * 8, (double)(gch[ch] * wgt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
              float* const gp = reinterpret_cast<float*>(gbytes + (size_t)(vo + (unsigned)(cm * stride_m + cu * stride_u + cv * stride_v) * TB));
#pragma unroll
              for (int ch = 0; ch < 4; ++ch)
                atomicAdd(gp + ch, gch[ch] * wgt);
            }
          }
        }
      }
    };
    if constexpr (NP > 1) {
      // ---- stage A (lanes with a live sample): everything of the sample that does not need the transmittance ---------------
      const float z = z_cur;
      const bool last = (k == Sm1);
      const bool act = on && fp.inside;
      float om = 1.0f, alpha = 0.0f, dldw = 0.0f, de = 0.0f, dpost = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
      if (act) {
        const float rad[COUT] = {kC0 * f0, kC0 * f1, kC0 * f2};
        float sigma;
        post_activate_vg(g.post_act, v, sigma, dpost);
        const float dl = last ? kInfinity : (z_n1 - z);
        const float delta = dl * rc.dnorm;
        const float e = fast_exp(-(sigma * delta));
        alpha = 1.0f - e;
        om = 1.0f - alpha;
        de = delta * e;
        c0 = sigmoidf(rad[0]); c1 = sigmoidf(rad[1]); c2 = sigmoidf(rad[2]);
        dldw = fmaf(gdep, z, gacc);
        dldw = fmaf(gc[0], c0, dldw); dldw = fmaf(gc[1], c1, dldw); dldw = fmaf(gc[2], c2, dldw);
        dldw -= gsumw;
      }
      // ---- stage B (all 64 lanes): the NP samples of a ray, in sample order ------------------------------------------------
      // value of phase q of this lane's ray: own for q == phase, else the partner lane's (phase ^ q: xor of the part's lane bits)
      auto of_phase = [&](float own, float (&out)[NP]) __attribute__((always_inline)) {
        if constexpr (NP == 1) { out[0] = own; return; }
        const float s1 = __shfl_xor(own, xm1, 64);
        float s2 = 0.0f, s3 = 0.0f;
        if constexpr (NP == 4) { s2 = __shfl_xor(own, xm2, 64); s3 = __shfl_xor(own, xm1 ^ xm2, 64); }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          const int x = phase ^ q;
          out[q] = (x == 0) ? own : ((x == 1) ? s1 : ((x == 2) ? s2 : s3));
        }
      };
      float omv[NP], dlv[NP], wkv[NP];
      of_phase(om, omv);
      float Tk = T, Tn = T;
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        if (q < phase) Tk = Tk * omv[q];
        Tn = Tn * omv[q];
      }
      const float wk = alpha * Tk;
      // the running sum of dL/dw w in sample order, one fused multiply-add per sample like the one-sample march (lanes without a
      // live sample contribute dL/dw = 0: fma(0, w, run) = run)
      of_phase(dldw, dlv);
      of_phase(wk, wkv);
      float run_k = run, run_n = run;
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        run_n = fmaf(dlv[q], wkv[q], run_n);
        if (q == phase) run_k = run_n;
      }
      T = Tn;
      run = run_n;
      // ---- stage C: the sample's gradient channels and their deposit ------------------------------------------------------
      if (act) {
        const float suffix = last ? 0.0f : (suffix0 - run_k);
        const float tail = (om > 0.0f) ? suffix * fast_rcp(om) : 0.0f;
        const float dsig = de * fmaf(Tk, dldw, -tail);
        float gch[4];
        gch[0] = ((wk * gcf[0]) * (c0 * (1.0f - c0))) * kC0;
        gch[1] = ((wk * gcf[1]) * (c1 * (1.0f - c1))) * kC0;
        gch[2] = ((wk * gcf[2]) * (c2 * (1.0f - c2))) * kC0;
        gch[3] = (dsig * dpost) * dmask;
        if (!(VOXE_T4_EXP & 16) && (wk != 0.0f || gch[3] != 0.0f)) do_deposit(gch);
      }
      if (term_eps > 0.0f && Tn < term_eps && k <= k_hi) k_hi = k;   // gradient truncation (not in the reference): per ray, behind the wave's NP samples
      // ---- this lane's next sample: k + NP --------------------------------------------------------------------------------
      if (on) {
        nextkey = INT_MAX;
        if (k + NP <= k_hi) {
          float pn[3];
          rc.point(z_nP, pn);
          footprint(g, pn, fp);
          z_cur = z_nP;
          nextkey = fp.i0[MA] ^ smask;
        }
      }
    } else {
    if (on) {
      const float z = z_cur;
      const bool last = (k == Sm1);        // wave-uniform (LK: per lane)
      const float z_next = last ? z : (LK ? z_n1 : depth_of(k + 1));
      if (fp.inside) {
        float gch[4];
        bool deposit;
        if constexpr (DEP) {
          const float sA = dep.cA == 0 ? f0 : (dep.cA == 1 ? f1 : (dep.cA == 2 ? f2 : v));
          const float sB = dep.cB == 0 ? f0 : (dep.cB == 1 ? f1 : (dep.cB == 2 ? f2 : v));
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) gch[s4] = fmaf(sB, dep.mB[s4], sA * dep.mA[s4]);   // (one of the two products is an exact zero)
          deposit = gch[0] != 0.0f || gch[1] != 0.0f || gch[2] != 0.0f || gch[3] != 0.0f;
        } else {
        const float rad[COUT] = {kC0 * f0, kC0 * f1, kC0 * f2};
        float sigma, dpost;
        post_activate_vg(g.post_act, v, sigma, dpost);
        const float dl = last ? kInfinity : (z_next - z);
        const float delta = dl * rc.dnorm;
        const float e = fast_exp(-(sigma * delta));
        const float alpha = 1.0f - e;
        const float om = 1.0f - alpha;
        float col[COUT], dldw, wk, suffix, Tk;
#pragma unroll
        for (int ch = 0; ch < COUT; ++ch) col[ch] = sigmoidf(rad[ch]);
        if constexpr (PREC) {
          const float wl = alpha * Tl;                       // the forward's local weight, bit for bit
          double dd = fma((double)gdep, (double)z, (double)gacc);
#pragma unroll
          for (int ch = 0; ch < COUT; ++ch) dd = fma((double)gc[ch], (double)col[ch], dd);
          dd -= (double)gsumw;
          rem = fma(-dd, (double)wl, rem);
          dldw = (float)dd;
          suffix = last ? 0.0f : T_start * (float)rem;
          Tk = T_start * Tl;
          wk = T_start * wl;
          Tl = Tl * om;
        } else {
          dldw = fmaf(gdep, z, gacc);
#pragma unroll
          for (int ch = 0; ch < COUT; ++ch) dldw = fmaf(gc[ch], col[ch], dldw);
          dldw -= gsumw;
          Tk = T;
          wk = alpha * T;
          run = fmaf(dldw, wk, run);
          suffix = last ? 0.0f : (suffix0 - run);
        }
        const float tail = (om > 0.0f) ? suffix * fast_rcp(om) : 0.0f;
        const float dsig = (delta * e) * fmaf(Tk, dldw, -tail);
#pragma unroll
        for (int ch = 0; ch < COUT; ++ch) gch[ch] = ((wk * gcf[ch]) * (col[ch] * (1.0f - col[ch]))) * kC0;
        gch[3] = (dsig * dpost) * dmask;
        if constexpr (PREC) { if (term_eps > 0.0f && Tk * om < term_eps) k_hi = k; }
        else { T = T * om; if (term_eps > 0.0f && T < term_eps) k_hi = k; }   // gradient truncation (not in the reference)
        deposit = wk != 0.0f || gch[3] != 0.0f;
        }

        if (!(VOXE_T4_EXP & 16) && deposit) do_deposit(gch);
      }
      // ---- the next sample's footprint (after the deposit: nothing of it is live across the 32 LDS adds) -------------------
      nextkey = INT_MAX;
      if (!last && k < k_hi) {
        float pn[3];
        rc.point(z_next, pn);
        footprint(g, pn, fp);
        z_cur = z_next;
        nextkey = fp.i0[MA] ^ smask;
      }
    }
    }   // (NP == 1)
    // ---- slide the window: flush every layer no lane can reach any more -------------------------------------------------
    // (one compare per iteration: does ANY lane's next sample still reach layer `base`?)
    if (__ballot(nextkey <= base) == 0ull) {   // wave-uniform
      int n = 0;
      do {
        if (!kPair || npend == 2) flush_consume(true);   // (room for the layer about to be read)
        flush_issue(base + n);
        if (VOXE_T4_ZPAIR == 2 && kPair && npend == 2) flush_consume(true);   // (experiment: the pair goes out at once, pendB is not loop carried)
        ++n;
      } while (n < kRing && __ballot(nextkey <= base + n) == 0ull);
      base = (n == kRing) ? wave_min_dpp(nextkey) : base + n;
      if (base != INT_MAX && base + nkey0 > kTabKeys / 2) {   // re-base the table (everything below `base` is flushed)
        key0 = base;
        nkey0 = -key0;
        build_tab(key0);
      }
    }
  }
  flush_consume(true);
  if (base != INT_MAX) {
    for (int i = 0; i < kRing; ++i) {
      flush_issue(base + i);
      if (!kPair || npend == 2) flush_consume(true);
    }
    flush_consume(true);
  }
}

// NCU_DEP > 0: the deposit pass of a view-dependent grid with NCU_DEP coefficients per colour (MODE 2 of render_bwd_tile_kernel:
// same block order -- channel group outermost --, same lanes -> pixels, same source-buffer slots as its source pass)
template <int KL, bool PREC, int NCU_DEP = 0>
#ifndef VOXE_TILE4_LB_PREC
#define VOXE_TILE4_LB_PREC 2   // the precise kernels at 3 waves per SIMD spill inside the sample loop (0.64 vs 0.50 ms on the bench camera)
#endif
__global__ __launch_bounds__(64, KL >= 9 ? 2 : (PREC ? VOXE_TILE4_LB_PREC : VOXE_TILE4_LB)) void render_bwd_tile4_kernel(const DevGrid g, const DevCfg c, const Tile4Args a_in) {
  constexpr bool DEP = NCU_DEP > 0;
  __shared__ double win[WinMap<KL, 4>::kDoubles];
  __shared__ int2 tab[kTabKeys];
  const int lane = threadIdx.x;
  for (int i = lane; i < WinMap<KL, 4>::kDoubles; i += 64) win[i] = 0.0;

  // ---- block -> ([channel group,] pixel tile, depth segment[, part]): the block order of render_bwd_tile_kernel ----------
  const int W = c.image_width;
  const int ntx = (W + 7) >> 3, nty = (int)tile_rows_total(c, 8);
  const int nseg = num_segments(c.S, c.seg_len);
  const int ntp = gridDim.x / (nseg * a_in.qsplit * (DEP ? a_in.ngrp : 1));
  int part = blockIdx.x / ntp;
  int grp = 0;
  if constexpr (DEP) { grp = part / (nseg * a_in.qsplit); part -= grp * nseg * a_in.qsplit; }
  Tile4Args a = a_in;
  if constexpr (DEP) a.gpacked = a_in.gpacked + (long long)grp * a_in.nvox * 4;   // the group's plane of the staging gradient
  const int quad = part / nseg, seg = part - quad * nseg;
  const int tile = logical_tile_of(c, blockIdx.x % ntp, ntp, ntx, nty);
  if (tile < 0) return;  // launch padding (wave-uniform)
  const int ks = seg * c.seg_len, ke = min(c.S, ks + c.seg_len) - 1;
  const int ty = tile / ntx, tx = tile - ty * ntx;
  long long r_px;
  bool alive = tile_pixel_ray(c, ty, lane >> 3, (tx << 3) + (lane & 7), 8, r_px);
  long long r = alive ? r_px : 0;

  RayCtx<3, 1, 1> rc;
  rc.init(g, c, r, a.rays_o, a.rays_d, nullptr);
#ifndef VOXE_T4_ORIENT
#define VOXE_T4_ORIENT 1   // 0: lanes always along the pixel rows
#endif
  bool columns = false;           // lanes run down the image columns (wave-uniform)
  if (VOXE_T4_ORIENT && !DEP) {   // lanes along the image rows or down the columns (tile_lanes_down_columns): fewer cache lines per gather
                                  // (a deposit pass gathers nothing, and its lanes must sit where the source pass's did)
    const unsigned long long am0 = __ballot(alive);
    if ((am0 & 1ull) && (am0 >> 1 & 1ull) && (am0 >> 8 & 1ull)) {
      const float d0[3] = {readlane_f32(rc.d[0], 0), readlane_f32(rc.d[1], 0), readlane_f32(rc.d[2], 0)};
      const float dx[3] = {readlane_f32(rc.d[0], 1), readlane_f32(rc.d[1], 1), readlane_f32(rc.d[2], 1)};
      const float dy[3] = {readlane_f32(rc.d[0], 8), readlane_f32(rc.d[1], 8), readlane_f32(rc.d[2], 8)};
      if (tile_lanes_down_columns(g, d0, dx, dy)) {   // wave-uniform
        columns = true;
        alive = tile_pixel_ray(c, ty, lane & 7, (tx << 3) + (lane >> 3), 8, r_px);
        r = alive ? r_px : 0;
        rc.init(g, c, r, a.rays_o, a.rays_d, nullptr);
      }
    }
  }

  // ---- does the 8x8 tile fit the lateral window?  (the split decision of render_bwd_tile_kernel) -------------------------
  int split = 0;
  bool phases_fit = false;     // (r06) the parts of this tile leave room in the ring for 2 / 4 consecutive samples of a ray
  bool phases_skew = false;    // (r06) ... only when their lanes are shifted by the samples a ray is ahead of the part's reference ray
  {
    const unsigned long long am = __ballot(alive);
    if ((am >> 1 & 1ull) && (am >> 8 & 1ull)) {
      const int N[3] = {g.X, g.Y, g.Z};
      const float zref = readlane_f32(rc.dg.zlin(ke), 0);
      float d0[3], ex3[3], ey3[3];
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        const float s = g.scale[ax] * 0.5f * (float)N[ax];
        const float da = readlane_f32(rc.d[ax], 0);
        d0[ax] = fabsf(da * s);
        ex3[ax] = fabsf((readlane_f32(rc.d[ax], 1) - da) * s * zref);
        ey3[ax] = fabsf((readlane_f32(rc.d[ax], 8) - da) * s * zref);
      }
      const int m = (d0[0] >= d0[1] && d0[0] >= d0[2]) ? 0 : ((d0[1] >= d0[2]) ? 1 : 2);
      const float fit_lat = a.fit_lat > 0.0f ? a.fit_lat : (float)KL - 2.5f;
      // r06: with the per-lane sample shift of the skewed march (bwd4_march, LK) a pass's extent along the march axis is no
      // constraint any more; what counts is its lateral extent at equal LAYER: sliding a ray back by its lead along m moves it
      // sideways by lead x |d_lat / d_m|
      const bool skew_ok = VOXE_T4_SKEW == 1 && !PREC && !DEP && a.phases >= 0;
      auto fits_pass = [&](float wx, float wy) {
        float e[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) e[ax] = wx * ex3[ax] + wy * ey3[ax];
        const float alongm = (m == 0) ? e[0] : ((m == 1) ? e[1] : e[2]);
        const float dm = (m == 0) ? d0[0] : ((m == 1) ? d0[1] : d0[2]);
        float lat = 0.0f;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax)
          if (ax != m) lat = fmaxf(lat, skew_ok ? e[ax] + alongm * (d0[ax] / dm) : e[ax]);
        return lat <= fit_lat && (skew_ok || alongm <= a.fit_m);
      };
      if (!fits_pass(7.0f, 7.0f)) {
        const bool hx = fits_pass(3.0f, 7.0f), hy = fits_pass(7.0f, 3.0f);
        const float sx = ex3[0] + ex3[1] + ex3[2], sy = ey3[0] + ey3[1] + ey3[2];
        if (hx && hy) split = (sx >= sy) ? 1 : 2;
        else split = hx ? 1 : (hy ? 2 : 3);
        // sample phases (bwd4_march, NP) put NP consecutive samples of a ray into one wave iteration: the part's extent along the
        // march axis grows by (NP - 1) x the layers a sample advances.  Only parts that still fit the ring run phased -- the others
        // would send their samples down the per-corner global-atomic path (100x100, oblique views: 0.25 -> 0.34 ms, r06l)
        const float pe = ((split == 2) ? 7.0f : 3.0f) * ((m == 0) ? ex3[0] : ((m == 1) ? ex3[1] : ex3[2])) +
                         ((split == 1) ? 7.0f : 3.0f) * ((m == 0) ? ey3[0] : ((m == 1) ? ey3[1] : ey3[2]));
        const float dzs = fabsf(zref - readlane_f32(rc.dg.zlin(max(ke - 1, 0)), 0));
        const float per_sample = ((m == 0) ? d0[0] : ((m == 1) ? d0[1] : d0[2])) * dzs;
        const float more = (float)((split == 3) ? 3 : 1) * per_sample;
        phases_fit = pe + more <= a.fit_m + VOXE_T4_PHASE_MARGIN;
        if (VOXE_T4_SKEW == 2 && !phases_fit) {
          // a part that splits for its extent ALONG the march axis: with every lane shifted by its ray's lead (bwd4_march: koff) the
          // part occupies one ray's layers (+ 1 for the rounding of the shift) -- at the price of its lateral extent at equal layer,
          // lead x |d_lat / d_m| wider, which has to fit the window as well
          const float wx = (split == 2) ? 7.0f : 3.0f, wy = (split == 1) ? 7.0f : 3.0f;
          const float dmm = (m == 0) ? d0[0] : ((m == 1) ? d0[1] : d0[2]);
          float lat = 0.0f;
#pragma unroll
          for (int ax = 0; ax < 3; ++ax)
            if (ax != m) lat = fmaxf(lat, wx * ex3[ax] + wy * ey3[ax] + pe * (d0[ax] / dmm));
          phases_skew = lat <= fit_lat && 1.0f + more <= a.fit_m + VOXE_T4_PHASE_MARGIN;
          phases_fit = phases_skew;
        }
      }
    }
  }
  // the strata of this depth segment: lane l holds (lower, span) of sample ks + l (DepthGen's own expressions)
  float strat_lo = 0.0f, strat_sp = 0.0f;
  if (ks + lane < c.S && lane <= ke + 1 - ks) {
    const float2 st = depth_stratum(rc.dg, ks + lane);
    strat_lo = st.x; strat_sp = st.y;
  }
  DepCtx dep;
  dep.cA = dep.cB = 0; dep.src_base = 0;
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) dep.mA[s4] = dep.mB[s4] = 0.0f;
  if constexpr (DEP) {
    // window channel s4 = gradient channel q = 4 grp + s4: coefficient q % NCU of colour q / NCU (factor: that basis value of
    // this ray), the density last (factor 1), nothing beyond (render_bwd_tile_kernel's chsel / mult tables)
    float basis[NCU_DEP];
    {
      const float vdir[3] = {rc.d[0] / rc.dnorm, rc.d[1] / rc.dnorm, rc.d[2] / rc.dnorm};   // (as RayCtx::init)
      sh_basis<NCU_DEP>(vdir, basis);
    }
    const int q0 = 4 * grp;
    dep.cA = q0 >= a.ng - 1 ? 3 : q0 / NCU_DEP;
    dep.cB = dep.cA;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int q = q0 + s4;
      if (q >= a.ng) continue;
      const int src = q == a.ng - 1 ? 3 : q / NCU_DEP;
      float m = 1.0f;
      if (src != 3) {
        const int j = q - src * NCU_DEP;
        m = basis[0];
#pragma unroll
        for (int t = 1; t < NCU_DEP; ++t) m = (j == t) ? basis[t] : m;
      }
      if (src == dep.cA) dep.mA[s4] = m;
      else { dep.cB = src; dep.mB[s4] = m; }
    }
    dep.src_base = ((long long)tile * nseg + seg) * c.seg_len * 64 + lane - (long long)ks * 64;
  }

  // PH (r06): sample phases per ray of this pass (1 | 2 | 4); phase / xm1 / xm2: this lane's phase and the xor masks between the
  // lanes of one ray (bwd4_march)
  auto run_pass = [&](const bool alive_q, const int centre_lane, const int centre_lane2, auto ph_tag, const int phase, const int xm1, const int xm2) {
    constexpr int PH = decltype(ph_tag)::value;
    const int k_lo = max(rc.k_lo, ks);
    int k_hi = alive_q ? min(rc.k_hi, ke) : k_lo - 1;
    bool has = k_lo <= k_hi;
    if (has && seg > 0 && c.term_eps > 0.0f) {   // gradient truncation: nothing behind T < term_eps receives a gradient
      if (a.ray_state[ray_state_index(seg, 0, 6, c.R, r)] < c.term_eps) { has = false; k_hi = k_lo - 1; }
    }
    const int kmin = wave_min_i32(has ? k_lo : INT_MAX);
    const int kmax = wave_max_i32(has ? k_hi : -1);
    if (kmin > kmax) return;  // wave-uniform: no ray of this pass meets the volume in this segment
    // ---- window geometry from the reference ray (render_bwd_tile_kernel) ----
    Geo4 geo;
    int m, ref_lane;
    {
      const unsigned long long hmk = __ballot(has);
      const int ref = ((hmk >> centre_lane) & 1ull) ? centre_lane : (__ffsll((long long)hmk) - 1);
      ref_lane = ref;
      const int ref2 = (VOXE_TILE_CENTRE2 && ref == centre_lane && ((hmk >> centre_lane2) & 1ull)) ? centre_lane2 : ref;
      const int N[3] = {g.X, g.Y, g.Z};
      float U0[3], DU[3];
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        const float ro = readlane_f32(rc.o[ax], ref), rd = 0.5f * (readlane_f32(rc.d[ax], ref) + readlane_f32(rc.d[ax], ref2));
        const float half = 0.5f * (float)N[ax];
        U0[ax] = ((ro * g.scale[ax] + g.bias[ax]) + 1.0f) * half - 0.5f;
        DU[ax] = rd * g.scale[ax] * half;
      }
      const float ax_ = fabsf(DU[0]), ay_ = fabsf(DU[1]), az_ = fabsf(DU[2]);
      m = (ax_ >= ay_ && ax_ >= az_) ? 0 : ((ay_ >= az_) ? 1 : 2);
      const int u = (m == 0) ? 1 : 0, v = (m == 2) ? 1 : 2;
      const float DUm = (m == 0) ? DU[0] : ((m == 1) ? DU[1] : DU[2]);
      const float U0m = (m == 0) ? U0[0] : ((m == 1) ? U0[1] : U0[2]);
      const float DUu = (u == 0) ? DU[0] : DU[1], U0u = (u == 0) ? U0[0] : U0[1];
      const float DUv = (v == 1) ? DU[1] : DU[2], U0v = (v == 1) ? U0[1] : U0[2];
      geo.sgn = (DUm < 0.0f) ? -1 : 1;
      const float inv = (DUm != 0.0f) ? 1.0f / DUm : 0.0f;
      geo.Bu = DUu * inv; geo.Au = U0u - geo.Bu * U0m;
      geo.Bv = DUv * inv; geo.Av = U0v - geo.Bv * U0m;
    }
    // ---- skew (r06): how many samples is this lane's ray ahead of the reference ray along the march axis?  Position along m of
    // a ray at depth z (voxel units): ((o + d z) scale + bias + 1) N / 2 - 1 / 2; evaluated at the middle of the segment; one
    // sample advances the reference ray by |DU_m| dz layers (dz: the launch's sample spacing at that depth)
    int sh = 0;
    if constexpr ((VOXE_T4_SKEW == 1 || (VOXE_T4_SKEW == 2 && PH > 1)) && !PREC && !DEP) {
      if (a.phases >= 0 && has && (VOXE_T4_SKEW == 1 || phases_skew)) {
        const int kmid = (kmin + kmax) >> 1;
        const float zmid = readlane_f32(rc.dg.zlin(kmid), ref_lane), zmid1 = readlane_f32(rc.dg.zlin(min(kmid + 1, c.S - 1)), ref_lane);
        const float om_ = (m == 0) ? rc.o[0] : ((m == 1) ? rc.o[1] : rc.o[2]), dm_ = (m == 0) ? rc.d[0] : ((m == 1) ? rc.d[1] : rc.d[2]);
        const float sc_ = (m == 0) ? g.scale[0] : ((m == 1) ? g.scale[1] : g.scale[2]);
        const float half_ = 0.5f * (float)((m == 0) ? g.X : ((m == 1) ? g.Y : g.Z));
        const float pos = (om_ + dm_ * zmid) * sc_ * half_;                   // (+ constants that cancel in the difference)
        const float pos_ref = readlane_f32(pos, ref_lane);
        const float per_sample = fabsf(readlane_f32(dm_, ref_lane) * sc_ * half_ * (zmid1 - zmid));
        if (per_sample > 1e-6f) {
          const float ahead = (float)geo.sgn * (pos - pos_ref) / per_sample;
          sh = (int)rintf(fminf(fmaxf(ahead, -12.0f), 12.0f));
        }
      }
    }
    const bool skew = __ballot(sh != 0) != 0ull;     // wave-uniform
#define VOXE_T4_MARCH(AX)                                                                                                           \
    do {                                                                                                                            \
      if constexpr (PH > 1)                                                                                                         \
        bwd4_march<AX, KL, PREC, DEP, PH, true>(g, c, a, rc, win, tab, lane, r, has, k_lo, k_hi, kmin, kmax, seg, ks, geo, strat_lo, \
                                                strat_sp, dep, phase, xm1, xm2, phase - sh);                                   \
      else if constexpr (VOXE_T4_SKEW == 1 && !PREC && !DEP) {                                                                           \
        if (skew)                                                                                                                   \
          bwd4_march<AX, KL, PREC, DEP, 1, true>(g, c, a, rc, win, tab, lane, r, has, k_lo, k_hi, kmin, kmax, seg, ks, geo, strat_lo, \
                                                 strat_sp, dep, 0, 0, 0, -sh);                                                      \
        else                                                                                                                        \
          bwd4_march<AX, KL, PREC, DEP>(g, c, a, rc, win, tab, lane, r, has, k_lo, k_hi, kmin, kmax, seg, ks, geo, strat_lo, strat_sp, dep); \
      } else                                                                                                                        \
        bwd4_march<AX, KL, PREC, DEP>(g, c, a, rc, win, tab, lane, r, has, k_lo, k_hi, kmin, kmax, seg, ks, geo, strat_lo, strat_sp, dep); \
    } while (0)
    if (m == 0) VOXE_T4_MARCH(0);
    else if (m == 1) VOXE_T4_MARCH(1);
    else VOXE_T4_MARCH(2);
#undef VOXE_T4_MARCH
  };
  auto in_part = [&](int q) {
    const int hx = (lane >> 2) & 1, hy = (lane >> 5) & 1;
    return split == 0 ? true : (split == 1 ? hx == q : (split == 2 ? hy == q : hx + 2 * hy == q));
  };
  auto centre_of = [&](int q) {
    return split == 0 ? 27 : (split == 1 ? 26 + 4 * q : (split == 2 ? 19 + 32 * q : 18 + 4 * (q & 1) + 32 * (q >> 1)));
  };
  auto centre2_of = [&](int q) {
    return split == 0 ? 36 : (split == 1 ? 33 + 4 * q : (split == 2 ? 12 + 32 * q : 9 + 4 * (q & 1) + 32 * (q >> 1)));
  };
  const int nparts = split == 0 ? 1 : (split == 3 ? 4 : 2);
  const int q_begin = a.qsplit == 4 ? quad : 0;
  const int q_end = a.qsplit == 4 ? min(quad + 1, nparts) : nparts;
  // r06: the parts of a split tile with 2 / 4 sample phases per ray -- the lanes outside part q take the other phases of the
  // part's rays (lane bits 2 / 5 say which part a lane's own pixel is in: xor-ing them with q gives the phase, forcing them to q
  // the lane whose ray this lane works on)
#ifndef VOXE_T4_PHASES_KL8
#define VOXE_T4_PHASES_KL8 0      // 1: sample phases compiled into the 8-wide kernel too.  They are not: the phased march variants in the same
                                  // kernel cost the ONE-sample march of the 8-wide window 8 % (400x400, same box, backward 0.448 -> 0.410 ms on
                                  // camera 3, 0.504 -> 0.466 over the 20 views: code size / register allocation of the shared prologue), and at
                                  // the image sizes that take the 8-wide window the parts of a split tile split for their march extent, which
                                  // phases do not help (profiles/r06_phases_kl8.txt).  The 10-wide kernel (images below ~0.58 x grid side pixels
                                  // per voxel: 100 .. 266 px on 160^3): see VOXE_T4_PHASES_KL10.
#endif
#ifndef VOXE_T4_PHASES_KL10
#define VOXE_T4_PHASES_KL10 0     // ... nor into the 10-wide one: over the 20 views the backward is 0.255 / 0.341 / 0.393 ms with them and
                                  // 0.248 / 0.343 / 0.387 ms without at 100 / 200 / 266 px (same box) -- what the phases win on the parts that
                                  // fit the ring, the one-sample march of all the other passes loses to the shared kernel.  The march variants
                                  // stay in the source (tests build them: tools/variants.py -DVOXE_T4_PHASES_KL10=1) as a measured alternative.
#endif
  constexpr bool kPhases = !PREC && !DEP && (KL >= 9 ? VOXE_T4_PHASES_KL10 != 0 : VOXE_T4_PHASES_KL8 != 0);
  if constexpr (kPhases) {
   if (split != 0 && a.phases >= 0 && phases_fit) {
    const int bx = (lane >> 2) & 1, by = (lane >> 5) & 1;
    for (int q = q_begin; q < q_end; ++q) {
      int phase, rl, xm1, xm2 = 0;
      if (split == 1) { phase = bx ^ q; rl = (lane & ~4) | (q << 2); xm1 = 4; }
      else if (split == 2) { phase = by ^ q; rl = (lane & ~32) | (q << 5); xm1 = 32; }
      else { phase = (bx ^ (q & 1)) | ((by ^ (q >> 1)) << 1); rl = (lane & ~36) | ((q & 1) << 2) | ((q >> 1) << 5); xm1 = 4; xm2 = 32; }
      const bool alive_q = columns ? tile_pixel_ray(c, ty, rl & 7, (tx << 3) + (rl >> 3), 8, r_px)
                                   : tile_pixel_ray(c, ty, rl >> 3, (tx << 3) + (rl & 7), 8, r_px);
      r = alive_q ? r_px : 0;
      rc.init(g, c, r, a.rays_o, a.rays_d, nullptr);
      if (split == 3) run_pass(alive_q, centre_of(q), centre2_of(q), std::integral_constant<int, 4>(), phase, xm1, xm2);
      else run_pass(alive_q, centre_of(q), centre2_of(q), std::integral_constant<int, 2>(), phase, xm1, xm2);
      __syncthreads();
    }
    return;
   }
  }
  for (int q = q_begin; q < q_end; ++q) {
    run_pass(alive && in_part(q), centre_of(q), centre2_of(q), std::integral_constant<int, 1>(), 0, 0, 0);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// The forward of the same renders, built the same way (r05): one wave = (8x8-pixel tile, depth segment), all lanes at the same
// sample index, strata from two VGPRs through v_readlane, gathers as scalar base + 32-bit offset, loads and interpolation in
// straight-line code.  No LDS: the timing experiments on the backward (profiles/r05_lean_experiments.txt) price its march +
// per-sample math at 0.13 ms with ~200 VALU instructions per wave-sample and the divergent gathers at 0.02 ms, against 0.21 ms
// of render_fwd_tile_kernel (LDS texel window: 230 VALU instructions per wave-sample, 4 waves per SIMD by its LDS).  Outputs
// are the per-segment partials (T, csum, asum, dsum) of render_fwd_seg_kernel, bit for bit: the same expressions in the same
// order (interp_texels4, post_activate, fast_exp, sigmoidf; depth = stratum lower + span * jitter as SegDepth<true>).
// ------------------------------------------------------------------------------------------------------------------------
#ifndef VOXE_FWD4_LB
#define VOXE_FWD4_LB 4
#endif
// PREC (VoxeDispatch::precise_grad): besides the float partials (unchanged, bit for bit) the segment-LOCAL sums of w col_c, w and
// z w are accumulated in double over the exact products and stored per (segment, component, ray): the backward subtracts the
// same products in the same order and gets the in-segment suffix sums without cancellation (see bwd4_march).
// COUT = 3: SH-0 colour grids (4-channel texels); COUT = 1: attention grids (2-channel texels: attention value, density)
template <int COUT, bool PREC>
__global__ __launch_bounds__(64, VOXE_FWD4_LB) void render_fwd_tile4_kernel(const DevGrid g, const DevCfg c,
                                                                            const float* __restrict__ packed,
                                                                            const float* __restrict__ rays_o,
                                                                            const float* __restrict__ rays_d,
                                                                            float* __restrict__ segbuf,
                                                                            double* __restrict__ segsum) {
  constexpr int NC = COUT + 3;
  static_assert(!PREC || COUT == 3, "precise_grad: SH-0 colour renders");
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
#ifndef VOXE_F4_EXP
#define VOXE_F4_EXP 0   // timing experiments: 1 lanes always down the pixel columns | 4 always along the rows | 2 (wrong results) the four z-neighbour loads dropped
#endif
  bool alive = tile_pixel_ray(c, ty, lane >> 3, (tx << 3) + (lane & 7), 8, r_px);
  long long r = alive ? r_px : 0;
  RayCtx<COUT, 1, 1> rc;
  rc.init(g, c, r, rays_o, rays_d, nullptr);
  {
    const unsigned long long am = __ballot(alive);
    bool cols = (VOXE_F4_EXP & 1) != 0;
    if (!(VOXE_F4_EXP & 5) && (am & 1ull) && (am >> 1 & 1ull) && (am >> 8 & 1ull)) {
      const float d0[3] = {readlane_f32(rc.d[0], 0), readlane_f32(rc.d[1], 0), readlane_f32(rc.d[2], 0)};
      const float dx[3] = {readlane_f32(rc.d[0], 1), readlane_f32(rc.d[1], 1), readlane_f32(rc.d[2], 1)};
      const float dy[3] = {readlane_f32(rc.d[0], 8), readlane_f32(rc.d[1], 8), readlane_f32(rc.d[2], 8)};
      cols = tile_lanes_down_columns(g, d0, dx, dy);
    }
    if (cols) {   // wave-uniform
      alive = tile_pixel_ray(c, ty, lane & 7, (tx << 3) + (lane >> 3), 8, r_px);
      r = alive ? r_px : 0;
      rc.init(g, c, r, rays_o, rays_d, nullptr);
    }
  }
  const int ks = seg * c.seg_len, ke = min(c.S, ks + c.seg_len) - 1;
  const int k_lo = max(rc.k_lo, ks);
  const int k_hi = alive ? min(rc.k_hi, ke) : k_lo - 1;
  const bool has = k_lo <= k_hi;
  const int kmin = wave_min_dpp(has ? k_lo : INT_MAX);
  const int kmax = -wave_min_dpp(has ? -k_hi : INT_MAX);
  float csum[COUT];
  double csum_d[COUT];
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) { csum[ch] = 0.0f; csum_d[ch] = 0.0; }
  float asum = 0.0f, dsum = 0.0f, T = 1.0f;
  double asum_d = 0.0, dsum_d = 0.0;
  if (kmin <= kmax) {   // wave-uniform
    // the strata of this depth segment: lane l holds (lower, span) of sample ks + l (DepthGen's own expressions)
    float strat_lo = 0.0f, strat_sp = 0.0f;
    if (ks + lane < c.S && lane <= ke + 1 - ks) {
      const float2 st = depth_stratum(rc.dg, ks + lane);
      strat_lo = st.x; strat_sp = st.y;
    }
    float z_next = 0.0f;
    if (has) {   // (the first sample index differs from lane to lane: its stratum is evaluated per lane)
      const float2 st = depth_stratum(rc.dg, k_lo);
      const float su = st.y * jitter_uniform(rc.dg.base, k_lo);
      z_next = st.x + su;
    }
    constexpr unsigned TB = (COUT + 1) * 4;   // bytes of a packed texel
    const unsigned sxb = g.X > 1 ? (unsigned)(g.Y * g.Z) * TB : 0u, syb = g.Y > 1 ? (unsigned)g.Z * TB : 0u;
    const unsigned szb = g.Z > 1 ? TB : 0u;
    const unsigned sxi = (unsigned)(g.Y * g.Z) * TB, syi = (unsigned)g.Z * TB;
    const char* const pbytes = reinterpret_cast<const char*>(packed);
    const int Sm1 = c.S - 1;
    for (int k = kmin; k <= kmax; ++k) {
      const bool on = has && (k >= k_lo) && (k <= k_hi);
      const bool last = (k == Sm1);        // wave-uniform
      const float z = z_next;
      if (on && !last) {
        const int j = k + 1 - ks;          // wave-uniform
        const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(strat_lo), j));
        const float sp = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(strat_sp), j));
        const float su = sp * jitter_uniform(rc.dg.base, k + 1);
        z_next = lo + su;
      }
      float p[3];
      rc.point(z, p);
      Footprint fp;
      footprint(g, p, fp);
      const bool live = on && fp.inside;
      if (__builtin_amdgcn_ballot_w64(live && !cell_is_interior(g, fp)) != 0ull) {   // (rare: a sample next to a grid face)
        Cell cf;
        make_cell(g, fp, cf);
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) { fp.i0[ax] = cf.i[ax]; fp.w[ax][0] = cf.w[ax][0]; fp.w[ax][1] = cf.w[ax][1]; }
      }
      Cell cell;
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) { cell.i[ax] = fp.i0[ax]; cell.w[ax][0] = fp.w[ax][0]; cell.w[ax][1] = fp.w[ax][1]; }
      unsigned off0 = mad24((unsigned)cell.i[0], sxi, mad24((unsigned)cell.i[1], syi, (unsigned)cell.i[2] * TB));
      if (!live) off0 = 0u;
      float fch[COUT], v;
      if constexpr (COUT == 3) {
        float4 t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const unsigned o = off0 + ((q & 1) ? sxb : 0u) + ((q & 2) ? syb : 0u);
          if ((VOXE_F4_EXP & 2) && (q & 4)) { t[q] = t[q - 4]; continue; }
          t[q] = *reinterpret_cast<const float4*>(pbytes + ((size_t)o + ((q & 4) ? szb : 0u)));
        }
        interp_texels4(t, cell, fch[0], fch[1], fch[2], v);
        asm volatile("" ::"v"(fch[0]), "v"(fch[1]), "v"(fch[2]), "v"(v));   // (loads consumed by every lane, in straight-line code: see the backward)
      } else {
        // 2-channel texels: the products and FMA order of gather<1, 1, 1>() (render_fwd_seg_kernel)
        float2 t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const unsigned o = off0 + ((q & 1) ? sxb : 0u) + ((q & 2) ? syb : 0u);
          t[q] = *reinterpret_cast<const float2*>(pbytes + ((size_t)o + ((q & 4) ? szb : 0u)));
        }
        float wxy[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) wxy[q] = cell.w[0][q & 1] * cell.w[1][q >> 1];
        fch[0] = 0.0f; v = 0.0f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float w = wxy[q & 3] * cell.w[2][q >> 2];
          fch[0] = fmaf(t[q].x, w, fch[0]);
          v = fmaf(t[q].y, w, v);
        }
        asm volatile("" ::"v"(fch[0]), "v"(v));
      }
      if (live) {
        float rad[COUT];
#pragma unroll
        for (int ch = 0; ch < COUT; ++ch) rad[ch] = kC0 * fch[ch];
        const float sigma = post_activate(g.post_act, v);
        const float dl = last ? kInfinity : (z_next - z);
        const float delta = dl * rc.dnorm;
        const float e = fast_exp(-(sigma * delta));
        const float alpha = 1.0f - e;
        const float om = 1.0f - alpha;
        const float w = alpha * T;
        T = T * om;
#pragma unroll
        for (int ch = 0; ch < COUT; ++ch) {
          const float col = sigmoidf(rad[ch]);
          csum[ch] = fmaf(col, w, csum[ch]);
          if constexpr (PREC) csum_d[ch] = fma((double)col, (double)w, csum_d[ch]);
        }
        asum = asum + w;
        dsum = fmaf(z, w, dsum);
        if constexpr (PREC) { asum_d += (double)w; dsum_d = fma((double)z, (double)w, dsum_d); }
      }
    }
  }
  if (!alive) return;
  if constexpr (PREC) {
    const long long pb = (long long)seg * 5;
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) segsum[(pb + ch) * c.R + r] = csum_d[ch];   // (COUT == 3)
    segsum[(pb + 3) * c.R + r] = asum_d;
    segsum[(pb + 4) * c.R + r] = dsum_d;
  }
  const long long base = (long long)seg * NC;
  segbuf[(base + 0) * c.R + r] = T;
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) segbuf[(base + 1 + ch) * c.R + r] = csum[ch];
  segbuf[(base + 1 + COUT) * c.R + r] = asum;
  segbuf[(base + 2 + COUT) * c.R + r] = dsum;
}

// ------------------------------------------------------------------------------------------------------------------------
// r06: the lean forward WITH the tile's texels in an LDS window.  The lean forward above is paced by its divergent 16-byte
// gathers (texture-address path: SQ_WAIT_INST_ANY 48 % of its wave cycles, and the whole view dependence of the forward --
// profiles/r06_pmc_by_camera.txt; with four of the eight loads dropped it runs in 0.147 instead of 0.182 ms,
// profiles/r06_fwd_window4.txt).  r03's window forward (voxe_render_tile.hip) removed the gathers but paid 230 VALU
// instructions per wave-sample and 121 registers; this is the same window -- a ring of layers along the march axis x 8 x 8
// lateral texels, sheared along the tile's reference ray, one coalesced 1 KB copy per layer -- inside the lean kernel's loop:
// strata through v_readlane, every lane reads its 8 corners with ds_read_b128 at two base addresses + immediates, and only
// when some lane's footprint is outside the window (ballot) do those lanes gather from global memory.  Tiles that do not fit
// the window (coarse pixels, z-dominant views by default) run the lean loop unchanged (WIN = false).  The interpolation is
// interp_texels4() on the same eight texels: outputs bit-identical to render_fwd_seg_kernel.
// ------------------------------------------------------------------------------------------------------------------------
#ifndef VOXE_FWD4W_RING
#define VOXE_FWD4W_RING 6
#endif
#ifndef VOXE_FWD4W_LB
#define VOXE_FWD4W_LB 4
#endif
#ifndef VOXE_FWD4W_PREFETCH
#define VOXE_FWD4W_PREFETCH 1
#endif
#ifndef VOXE_FWD4W_ZDOM
#define VOXE_FWD4W_ZDOM 1.0f   // > 0: z-dominant tiles run the lean loop; < 0: along z through the window
#endif
constexpr int kF4Ring = VOXE_FWD4W_RING;   // layers of the texel ring (6 KB)
constexpr int kF4Table = 64;               // layers tabulated per block: keys key0 .. key0 + 63

template <int M, bool WIN, bool PREC>
__device__ __forceinline__ void fwd4w_march(const DevGrid& g, const DevCfg& c, const float* __restrict__ packed, RayCtx<3, 1, 1>& rc,
                                            const int lane, const bool has, const int k_lo, const int k_hi, const int kmin,
                                            const int kmax, const int ks, const int ke, const int ref, float4* __restrict__ tex,
                                            int4* __restrict__ org, float (&csum)[3], double (&csum_d)[3], float& asum, float& dsum,
                                            float& T, double& asum_d, double& dsum_d) {
  constexpr int COUT = 3;
  constexpr int U = (M == 0) ? 1 : 0, V = (M == 2) ? 1 : 2;   // lateral axes (v = z whenever m != z: a layer is eight 128-byte runs)
  constexpr int kCtr = Lat<8>::kCentre;
  // the strata of this depth segment: lane l holds (lower, span) of sample ks + l (DepthGen's own expressions)
  float strat_lo = 0.0f, strat_sp = 0.0f;
  if (ks + lane < c.S && lane <= ke + 1 - ks) {
    const float2 st = depth_stratum(rc.dg, ks + lane);
    strat_lo = st.x; strat_sp = st.y;
  }
  float z_next = 0.0f;
  if (has) {   // (the first sample index differs from lane to lane: its stratum is evaluated per lane)
    const float2 st = depth_stratum(rc.dg, k_lo);
    const float su = st.y * jitter_uniform(rc.dg.base, k_lo);
    z_next = st.x + su;
  }
  constexpr unsigned TB = 16;   // bytes of a packed texel
  const unsigned sxb = g.X > 1 ? (unsigned)(g.Y * g.Z) * TB : 0u, syb = g.Y > 1 ? (unsigned)g.Z * TB : 0u;
  const unsigned szb = g.Z > 1 ? TB : 0u;
  const unsigned sxi = (unsigned)(g.Y * g.Z) * TB, syi = (unsigned)g.Z * TB;
  const char* const pbytes = reinterpret_cast<const char*>(packed);
  const int Sm1 = c.S - 1;
  // ---- window geometry from the reference ray, per-layer table, first layers (as fwd_window_march) ----
  int sgn = 1, key0 = 0, base = 0, up0 = 0;
  const int N[3] = {g.X, g.Y, g.Z};
  const int Nm2 = N[M] - 2;
  const int la = lane >> 3, lb8 = lane & 7;
  const int sx = g.Y * g.Z, sy = g.Z;
  const int stride_u = (U == 0) ? sx : sy, stride_v = (V == 1) ? sy : 1;
  const int lane_off = la * stride_u + lb8 * stride_v;
  auto minkey = [&](int pm) { return sgn > 0 ? pm : -(pm + 1); };
  // one coalesced read per layer: lane (a, b) fetches voxel (im, origin u + a, origin v + b)
  auto fetch_layer = [&](int key) -> float4 {
    const int idx = key - key0;      // wave-uniform
    float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (idx < kF4Table) {
      const int4 o = org[idx];
      const int im = sgn * key;
      if ((unsigned)im < (unsigned)N[M] && (unsigned)(o.x + la) < (unsigned)N[U] && (unsigned)(o.y + lb8) < (unsigned)N[V])
        t = reinterpret_cast<const float4*>(packed)[o.z + lane_off];
    }
    return t;
  };
  auto store_layer = [&](int key, const float4 t) { tex[((key - key0) % kF4Ring) * 64 + lane] = t; };
  auto load_layer = [&](int key) { store_layer(key, fetch_layer(key)); };
  float4 pend = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // layer base + ring, requested one slide early (VOXE_FWD4W_PREFETCH)
  bool have_pend = false;                               // wave-uniform
  if constexpr (WIN) {
    float U0[3], DU[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float ro = readlane_f32(rc.o[a], ref), rd = readlane_f32(rc.d[a], ref);
      const float half = 0.5f * (float)N[a];
      U0[a] = ((ro * g.scale[a] + g.bias[a]) + 1.0f) * half - 0.5f;
      DU[a] = rd * g.scale[a] * half;
    }
    sgn = (DU[M] < 0.0f) ? -1 : 1;
    const float inv = (DU[M] != 0.0f) ? 1.0f / DU[M] : 0.0f;
    const float Bu = DU[U] * inv, Au = U0[U] - Bu * U0[M];
    const float Bv = DU[V] * inv, Av = U0[V] - Bv * U0[M];
    const int stride_m = (M == 0) ? sx : ((M == 1) ? sy : 1);
    int first_key = INT_MAX;
    if (has) {
      float p0[3];
      rc.point(z_next, p0);
      Footprint f0;
      footprint(g, p0, f0);
      first_key = minkey(min(max(f0.i0[M], 0), Nm2));   // (the index make_cell() uses)
    }
    key0 = wave_min_dpp(first_key);
    up0 = sgn > 0 ? 0 : 1;
    {
      const int im = sgn * (key0 + lane);
      const int ou = (int)floorf(Au + Bu * (float)im) - kCtr, ov = (int)floorf(Av + Bv * (float)im) - kCtr;
      // (origin u, origin v, voxel offset of the origin -- used only when in range --, ring slot x 64)
      org[lane] = make_int4(ou, ov, im * stride_m + ou * stride_u + ov * stride_v, (lane % kF4Ring) * 64);
    }
    __syncthreads();
    base = key0;
#pragma unroll
    for (int i = 0; i < kF4Ring; ++i) load_layer(base + i);
    __syncthreads();
  }
  for (int k = kmin; k <= kmax; ++k) {
    const bool on = has && (k >= k_lo) && (k <= k_hi);
    const bool last = (k == Sm1);        // wave-uniform
    const float z = z_next;
    if (on && !last) {
      const int j = k + 1 - ks;          // wave-uniform
      const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(strat_lo), j));
      const float sp = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(strat_sp), j));
      const float su = sp * jitter_uniform(rc.dg.base, k + 1);
      z_next = lo + su;
    }
    float p[3];
    rc.point(z, p);
    Footprint fp;
    footprint(g, p, fp);
    const bool live = on && fp.inside;
    if (__builtin_amdgcn_ballot_w64(live && !cell_is_interior(g, fp)) != 0ull) {   // (rare: a sample next to a grid face)
      Cell cf;
      make_cell(g, fp, cf);
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) { fp.i0[ax] = cf.i[ax]; fp.w[ax][0] = cf.w[ax][0]; fp.w[ax][1] = cf.w[ax][1]; }
    }
    Cell cell;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) { cell.i[ax] = fp.i0[ax]; cell.w[ax][0] = fp.w[ax][0]; cell.w[ax][1] = fp.w[ax][1]; }
    float4 t[8];
    bool need = live;   // lanes whose corners come from global memory
    if constexpr (WIN) {
      const int kl = minkey(cell.i[M]);                  // lower key of the footprint's two layers; the other is kl + 1
      const int nb = wave_min_dpp(live ? kl : INT_MAX);  // the window starts at the layer the tile's rearmost live sample needs
      if (nb != INT_MAX && nb > base) {                  // wave-uniform: bring in the layers the march has reached
        for (int key = max(base + kF4Ring, nb); key < nb + kF4Ring; ++key) {
          if (have_pend && key == base + kF4Ring) store_layer(key, pend);
          else load_layer(key);
        }
        base = nb;
        if (VOXE_FWD4W_PREFETCH) { pend = fetch_layer(base + kF4Ring); have_pend = true; }   // travels while the samples of this layer are interpolated
        __syncthreads();
      }
      const int il = kl - key0;
      const int ilc = (int)min((unsigned)il, (unsigned)(kF4Table - 2));   // (il >= 0 for live lanes: key0 is the minimum of the first keys)
      const int4 o0 = org[ilc + up0], o1 = org[ilc + 1 - up0];   // origins / ring slots of layers pm, pm + 1 (keys kl, kl + 1 or the reverse)
      const int pu = cell.i[U], pv = cell.i[V];
      const unsigned a0 = (unsigned)(pu - o0.x), b0 = (unsigned)(pv - o0.y), a1 = (unsigned)(pu - o1.x), b1 = (unsigned)(pv - o1.y);
      // inside the ring, inside the table, laterally inside both layers (one unsigned maximum)
      const bool fits = live && ((unsigned)(kl - base) < (unsigned)(kF4Ring - 1)) && ((unsigned)il < (unsigned)(kF4Table - 1)) &&
                        (max(max(a0, b0), max(a1, b1)) < 7u);
      // (addresses of lanes that do not fit stay inside the ring: their reads are overwritten by the gather below or never used)
      constexpr unsigned kLast = kF4Ring * 64 - 10;
      const float4* __restrict__ t0 = tex + min((unsigned)o0.w + a0 * 8u + b0, kLast);
      const float4* __restrict__ t1 = tex + min((unsigned)o1.w + a1 * 8u + b1, kLast);
#pragma unroll
      for (int q = 0; q < 8; ++q) {   // corner q = (x + (q & 1), y + ((q >> 1) & 1), z + (q >> 2))
        const int dm = (q >> M) & 1, du = (q >> U) & 1, dv = (q >> V) & 1;   // compile-time after unrolling
        t[q] = (dm ? t1 : t0)[du * 8 + dv];
      }
      need = live && !fits;
    }
    if (!WIN || __builtin_amdgcn_ballot_w64(need) != 0ull) {
      unsigned off0 = mad24((unsigned)cell.i[0], sxi, mad24((unsigned)cell.i[1], syi, (unsigned)cell.i[2] * TB));
      if constexpr (WIN) {
        if (need) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const unsigned o = off0 + ((q & 1) ? sxb : 0u) + ((q & 2) ? syb : 0u);
            t[q] = *reinterpret_cast<const float4*>(pbytes + ((size_t)o + ((q & 4) ? szb : 0u)));
          }
        }
      } else {
        if (!live) off0 = 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const unsigned o = off0 + ((q & 1) ? sxb : 0u) + ((q & 2) ? syb : 0u);
          t[q] = *reinterpret_cast<const float4*>(pbytes + ((size_t)o + ((q & 4) ? szb : 0u)));
        }
      }
    }
    float fch[COUT], v;
    interp_texels4(t, cell, fch[0], fch[1], fch[2], v);
    asm volatile("" ::"v"(fch[0]), "v"(fch[1]), "v"(fch[2]), "v"(v));   // (reads consumed by every lane, in straight-line code)
    if (live) {
      float rad[COUT];
#pragma unroll
      for (int ch = 0; ch < COUT; ++ch) rad[ch] = kC0 * fch[ch];
      const float sigma = post_activate(g.post_act, v);
      const float dl = last ? kInfinity : (z_next - z);
      const float delta = dl * rc.dnorm;
      const float e = fast_exp(-(sigma * delta));
      const float alpha = 1.0f - e;
      const float om = 1.0f - alpha;
      const float w = alpha * T;
      T = T * om;
#pragma unroll
      for (int ch = 0; ch < COUT; ++ch) {
        const float col = sigmoidf(rad[ch]);
        csum[ch] = fmaf(col, w, csum[ch]);
        if constexpr (PREC) csum_d[ch] = fma((double)col, (double)w, csum_d[ch]);
      }
      asum = asum + w;
      dsum = fmaf(z, w, dsum);
      if constexpr (PREC) { asum_d += (double)w; dsum_d = fma((double)z, (double)w, dsum_d); }
    }
  }
}

template <bool PREC>
__global__ __launch_bounds__(64, VOXE_FWD4W_LB) void render_fwd_tile4w_kernel(const DevGrid g, const DevCfg c,
                                                                              const float* __restrict__ packed,
                                                                              const float* __restrict__ rays_o,
                                                                              const float* __restrict__ rays_d,
                                                                              float* __restrict__ segbuf, double* __restrict__ segsum,
                                                                              const float fit_lat, const float fit_m, const float zdom,
                                                                              const float max_adv) {
  constexpr int COUT = 3, NC = COUT + 3;
  __shared__ float4 tex[kF4Ring * 64];
  __shared__ int4 org[kF4Table];
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
  bool alive = tile_pixel_ray(c, ty, lane >> 3, (tx << 3) + (lane & 7), 8, r_px);
  long long r = alive ? r_px : 0;
  RayCtx<COUT, 1, 1> rc;
  rc.init(g, c, r, rays_o, rays_d, nullptr);
  const int ks = seg * c.seg_len, ke = min(c.S, ks + c.seg_len) - 1;
  // ---- per tile (wave-uniform): through the window along axis m, or the lean loop (render_fwd_tile_kernel's decision) ----
  int m = -1, ref = 0;
  {
    const int k_lo0 = max(rc.k_lo, ks), k_hi0 = alive ? min(rc.k_hi, ke) : k_lo0 - 1;
    const unsigned long long hm = __ballot(k_lo0 <= k_hi0), am = __ballot(alive);
    if (hm != 0ull && (am & 1ull) && (am >> 1 & 1ull) && (am >> 8 & 1ull)) {
      ref = ((hm >> 27) & 1ull) ? 27 : (__ffsll((long long)hm) - 1);
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
      const float adv = ad[mm] * fabsf(readlane_f32(rc.dg.zlin(ke) - rc.dg.zlin(ke > 0 ? ke - 1 : 0), ref));
      if (lat <= fit_lat && alongm <= fit_m && adv <= max_adv && (mm != 2 || zdom < 0.0f) && g.X > 1 && g.Y > 1 && g.Z > 1) m = mm;
    }
    if (m < 0 && (am & 1ull) && (am >> 1 & 1ull) && (am >> 8 & 1ull)) {   // gathers from global memory: the lane orientation that touches fewer lines
      const float d0[3] = {readlane_f32(rc.d[0], 0), readlane_f32(rc.d[1], 0), readlane_f32(rc.d[2], 0)};
      const float dx[3] = {readlane_f32(rc.d[0], 1), readlane_f32(rc.d[1], 1), readlane_f32(rc.d[2], 1)};
      const float dy[3] = {readlane_f32(rc.d[0], 8), readlane_f32(rc.d[1], 8), readlane_f32(rc.d[2], 8)};
      if (tile_lanes_down_columns(g, d0, dx, dy)) {   // wave-uniform
        alive = tile_pixel_ray(c, ty, lane & 7, (tx << 3) + (lane >> 3), 8, r_px);
        r = alive ? r_px : 0;
        rc.init(g, c, r, rays_o, rays_d, nullptr);
      }
    }
  }
  const int k_lo = max(rc.k_lo, ks);
  const int k_hi = alive ? min(rc.k_hi, ke) : k_lo - 1;
  const bool has = k_lo <= k_hi;
  const int kmin = wave_min_dpp(has ? k_lo : INT_MAX);
  const int kmax = -wave_min_dpp(has ? -k_hi : INT_MAX);
  float csum[COUT];
  double csum_d[COUT];
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) { csum[ch] = 0.0f; csum_d[ch] = 0.0; }
  float asum = 0.0f, dsum = 0.0f, T = 1.0f;
  double asum_d = 0.0, dsum_d = 0.0;
  if (kmin <= kmax) {   // wave-uniform
    if (m == 0) fwd4w_march<0, true, PREC>(g, c, packed, rc, lane, has, k_lo, k_hi, kmin, kmax, ks, ke, ref, tex, org, csum, csum_d, asum, dsum, T, asum_d, dsum_d);
    else if (m == 1) fwd4w_march<1, true, PREC>(g, c, packed, rc, lane, has, k_lo, k_hi, kmin, kmax, ks, ke, ref, tex, org, csum, csum_d, asum, dsum, T, asum_d, dsum_d);
    else if (m == 2) fwd4w_march<2, true, PREC>(g, c, packed, rc, lane, has, k_lo, k_hi, kmin, kmax, ks, ke, ref, tex, org, csum, csum_d, asum, dsum, T, asum_d, dsum_d);
    else fwd4w_march<0, false, PREC>(g, c, packed, rc, lane, has, k_lo, k_hi, kmin, kmax, ks, ke, ref, tex, org, csum, csum_d, asum, dsum, T, asum_d, dsum_d);
  }
  if (!alive) return;
  if constexpr (PREC) {
    const long long pb = (long long)seg * 5;
#pragma unroll
    for (int ch = 0; ch < COUT; ++ch) segsum[(pb + ch) * c.R + r] = csum_d[ch];
    segsum[(pb + 3) * c.R + r] = asum_d;
    segsum[(pb + 4) * c.R + r] = dsum_d;
  }
  const long long base = (long long)seg * NC;
  segbuf[(base + 0) * c.R + r] = T;
#pragma unroll
  for (int ch = 0; ch < COUT; ++ch) segbuf[(base + 1 + ch) * c.R + r] = csum[ch];
  segbuf[(base + 1 + COUT) * c.R + r] = asum;
  segbuf[(base + 2 + COUT) * c.R + r] = dsum;
}

// the lean forward takes what the lean backward takes (the window width does not matter to it)
bool fwd_tile4_supported(const DevGrid& g, const HostCfg& c, const FwdArgs& a, int cout, int ncm) {
  if (c.disp.tile_lean < 0 || ncm != 1 || !((cout == 3 && !c.attn) || (cout == 1 && c.attn))) return false;
  if (a.jitter || c.aabb_clip || c.image_width <= 0 || !a.segbuf) return false;
  if (c.seg_len + 1 > 64) return false;
  const long long bytes = (long long)g.X * g.Y * g.Z * 16;
  return bytes < (1ll << 31) && (long long)g.Y * g.Z * 16 < (1 << 24) && g.X < (1 << 24);
}
void launch_fwd_tile4(const DevGrid& g, const HostCfg& c, const FwdArgs& a, hipStream_t st) {
  const int nseg = num_segments(c.S, c.seg_len);
  const int nb = blocks_for_tiles(c.map_mode, (c.image_width + 7) / 8, tile_rows_total(c, 8)) * nseg;
  if (c.attn) render_fwd_tile4_kernel<1, false><<<nb, 64, 0, st>>>(g, c, a.packed, a.rays_o, a.rays_d, a.segbuf, nullptr);
  else if (c.disp.fwd_window >= 0 && (reinterpret_cast<uintptr_t>(a.packed) & 15) == 0) {   // r06: corners from an LDS window of the tile's texels
    const float fit_lat = disp_or(c.disp.fwd_fit_lat, 5.5f), fit_m = disp_or(c.disp.fwd_fit_m, (float)kF4Ring - 1.5f);
    const float zdom = disp_or(c.disp.fwd_zdom, VOXE_FWD4W_ZDOM), max_adv = disp_or(c.disp.fwd_max_adv, 1.7f);
    if (a.segsum_d) render_fwd_tile4w_kernel<true><<<nb, 64, 0, st>>>(g, c, a.packed, a.rays_o, a.rays_d, a.segbuf, a.segsum_d, fit_lat, fit_m, zdom, max_adv);
    else render_fwd_tile4w_kernel<false><<<nb, 64, 0, st>>>(g, c, a.packed, a.rays_o, a.rays_d, a.segbuf, nullptr, fit_lat, fit_m, zdom, max_adv);
  }
  else if (a.segsum_d) render_fwd_tile4_kernel<3, true><<<nb, 64, 0, st>>>(g, c, a.packed, a.rays_o, a.rays_d, a.segbuf, a.segsum_d);
  else render_fwd_tile4_kernel<3, false><<<nb, 64, 0, st>>>(g, c, a.packed, a.rays_o, a.rays_d, a.segbuf, nullptr);
}

// The lean kernel takes: SH-0 grids (4-channel texels), the 8-wide parity-class banked window, float atomics, in-kernel jitter
// (no caller-supplied uniforms), launch-wide (near, far) (no per-ray AABB bounds), depth segments that fit the strata
// registers, packed grids below 2 GiB (32-bit byte offsets).  VoxeDispatch::tile_lean = -1 switches it off (A/B, parity tests).
bool tile4_bwd_supported(const DevGrid& g, const HostCfg& c, const BwdArgs& a, int kl) {
  if (c.disp.tile_lean < 0) return false;
  if ((kl != 8 && kl != 10) || a.gdet || a.jitter || c.aabb_clip || c.attn || c.image_width <= 0) return false;
  if (c.seg_len + 1 > 64) return false;
  const long long bytes = (long long)g.X * g.Y * g.Z * 16;
  return bytes < (1ll << 31) && (long long)g.Y * g.Z * 16 < (1 << 24) && g.X < (1 << 24);   // (mad24 operands)
}

// ---- group-planar staging gradient -> packed gradient (accumulating) -------------------------------------------------------------
// chunks of 64 voxels through LDS: 16-byte reads of every group's plane, 16-byte read-modify-writes of 64 whole texels
__global__ __launch_bounds__(256) void planar_to_packed_kernel(const float4* __restrict__ planar, float* __restrict__ gpacked,
                                                               const long long nvox, const int cm, const int ngrp) {
  constexpr int V = 64;
  extern __shared__ float4 p2p_buf4[];
  float* const buf = reinterpret_cast<float*>(p2p_buf4);
  const int tid = threadIdx.x;
  const long long nchunks = nvox / V;
  for (long long ck = blockIdx.x; ck < nchunks; ck += gridDim.x) {
    for (int q = tid; q < V * ngrp; q += 256) {
      const int gi = q / V, i = q - gi * V;
      const float4 t = planar[(long long)gi * nvox + ck * V + i];
      const float e[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (4 * gi + u < cm) buf[i * cm + 4 * gi + u] = e[u];
    }
    __syncthreads();
    float4* __restrict__ g4 = reinterpret_cast<float4*>(gpacked + ck * V * cm);
    for (int q = tid; q < V * cm / 4; q += 256) {
      const float4 o = g4[q], t = p2p_buf4[q];
      g4[q] = make_float4(o.x + t.x, o.y + t.y, o.z + t.z, o.w + t.w);
    }
    __syncthreads();
  }
  if (blockIdx.x == 0) {   // the voxels behind the last whole chunk
    const float* const pl = reinterpret_cast<const float*>(planar);
    for (long long e = nchunks * V * cm + tid; e < nvox * cm; e += 256) {
      const long long vox = e / cm;
      const int ch = (int)(e - vox * cm);
      gpacked[e] += pl[((long long)(ch >> 2) * nvox + vox) * 4 + (ch & 3)];
    }
  }
}

// deposit passes of the two-phase backward of view-dependent grids (SH degree 1 - 3, not diffuse): the lean kernel, flushing
// into the group-planar staging gradient (BwdArgs::grad_planar); same conditions as the SH-0 kernel
bool tile4_dep_supported(const DevGrid& g, const HostCfg& c, const BwdArgs& a, int ncu, int cm) {
  if (c.disp.tile_lean < 0 || !(ncu == 4 || ncu == 9 || ncu == 16) || cm != 3 * ncu + 1) return false;
  if (a.gdet || a.jitter || c.aabb_clip || c.attn || c.image_width <= 0 || !a.sample_src || !a.grad_planar) return false;
  if (c.seg_len + 1 > 64) return false;
  const long long bytes = (long long)g.X * g.Y * g.Z * 16;
  return bytes < (1ll << 31) && (long long)g.Y * g.Z * 16 < (1 << 24) && g.X < (1 << 24);
}
size_t tile_planar_bytes(long long nvox, int cm) { return (size_t)nvox * 16 * (size_t)((cm + 3) / 4); }
void launch_bwd_tile4_dep(const DevGrid& g, const HostCfg& c, const BwdArgs& a, int ncu, int nb, int qsplit, int ngrp, float fit_m,
                          float fit_lat, hipStream_t st) {
  Tile4Args t;
  t.packed = a.packed; t.rays_o = a.rays_o; t.rays_d = a.rays_d; t.colour = a.colour; t.depth = a.depth; t.acc = a.acc;
  t.d_colour = a.d_colour; t.d_depth = a.d_depth; t.d_acc = a.d_acc; t.ray_state = a.ray_state; t.gpacked = a.grad_planar;
  t.qsplit = qsplit; t.fit_m = fit_m; t.fit_lat = fit_lat; t.want_d = 1; t.want_f = 1;
  t.segsum = nullptr; t.phases = -1;
  t.sample_src = reinterpret_cast<const float4*>(a.sample_src);
  const int cm = 3 * ncu + 1;
  t.ngrp = ngrp; t.ng = cm; t.nvox = (long long)g.X * g.Y * g.Z;
  (void)hipMemsetAsync(a.grad_planar, 0, tile_planar_bytes(t.nvox, cm), st);
  if (ncu == 4) render_bwd_tile4_kernel<8, false, 4><<<nb, 64, 0, st>>>(g, c, t);
  else if (ncu == 9) render_bwd_tile4_kernel<8, false, 9><<<nb, 64, 0, st>>>(g, c, t);
  else render_bwd_tile4_kernel<8, false, 16><<<nb, 64, 0, st>>>(g, c, t);
  const long long nchunks = t.nvox / 64;
  planar_to_packed_kernel<<<(int)(nchunks < 8192 ? (nchunks > 0 ? nchunks : 1) : 8192), 256, (size_t)64 * cm * sizeof(float), st>>>(
      reinterpret_cast<const float4*>(a.grad_planar), a.gpacked, t.nvox, cm, ngrp);
}

void launch_bwd_tile4(const DevGrid& g, const HostCfg& c, const BwdArgs& a, int kl, int nb, int qsplit, float fit_m, float fit_lat,
                      hipStream_t st) {
  Tile4Args t;
  t.sample_src = nullptr; t.ngrp = 1; t.ng = 4; t.nvox = 0;
  t.packed = a.packed; t.rays_o = a.rays_o; t.rays_d = a.rays_d; t.colour = a.colour; t.depth = a.depth; t.acc = a.acc;
  t.d_colour = a.d_colour; t.d_depth = a.d_depth; t.d_acc = a.d_acc; t.ray_state = a.ray_state; t.gpacked = a.gpacked;
  t.qsplit = qsplit; t.fit_m = fit_m; t.fit_lat = fit_lat; t.want_d = a.want_d ? 1 : 0; t.want_f = a.want_f ? 1 : 0;
  t.segsum = a.segsum_d; t.phases = c.disp.tile_phases;
  if (a.segsum_d) {   // VoxeDispatch::precise_grad
    if (kl == 10) render_bwd_tile4_kernel<10, true><<<nb, 64, 0, st>>>(g, c, t);
    else render_bwd_tile4_kernel<8, true><<<nb, 64, 0, st>>>(g, c, t);
  } else if (kl == 10) render_bwd_tile4_kernel<10, false><<<nb, 64, 0, st>>>(g, c, t);
  else render_bwd_tile4_kernel<8, false><<<nb, 64, 0, st>>>(g, c, t);
}

}  // namespace voxe
