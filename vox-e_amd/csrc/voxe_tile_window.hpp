// voxe_tile_window.hpp -- the sliding LDS gradient window of the image-ordered backward kernels: ring depth, lateral
// extent, the parity-class banked address map (WinMap) and the wave-level helpers both tile kernel files use
// (voxe_render_tile.hip: every grid kind; voxe_render_tile4.hip: the lean SH-0 kernel).
#pragma once

#include <limits.h>

#include "voxe_device.hpp"

namespace voxe {

#ifndef VOXE_TILE_ORIENT_K
#define VOXE_TILE_ORIENT_K 1.0f   // (1e30f: always along the pixel rows, the r02 mapping; 0: always down the columns)
#endif
#ifndef VOXE_TILE_CENTRE2
#define VOXE_TILE_CENTRE2 1
#endif
#ifndef VOXE_TILE_RING
#define VOXE_TILE_RING 6
#endif
// swept over 15 cameras: 6 layers (12.4 KB: 12 one-wave blocks per CU, the same bound as the 168 VGPRs) is 8 % faster
// than 8 (9 blocks per CU) for every camera; 5 is worse for oblique views, 4 overflows the window for them (2.1x
// slower), 16 halves the residency (1.9x slower)
constexpr int kRing = VOXE_TILE_RING;    // live layers along the march axis
// ring position of a layer key (keys can be negative)
__device__ __forceinline__ int ring_slot(int key) {
  if constexpr ((VOXE_TILE_RING & (VOXE_TILE_RING - 1)) == 0) return key & (VOXE_TILE_RING - 1);
  const int m = key % VOXE_TILE_RING;
  return m < 0 ? m + VOXE_TILE_RING : m;
}
// Lateral window edge KL (voxels): 8 for images whose pixels are at most ~0.5 voxel apart (a whole 8x8-pixel tile then
// spans < 5.5 voxels); 10 for coarse images (about one pixel per voxel or fewer), where an 8-wide window would force
// most tiles into 16-lane quadrants.  The price is LDS: 12.4 / 19.4 KB per one-wave block (12 did not pay: 27.8 KB).
// channel planes are padded by a few doubles so the C channels of one voxel, read together by the
// flush, sit in different bank groups; the plane offset folds into the ds instruction's immediate
#ifndef VOXE_TILE_PAD
#define VOXE_TILE_PAD 6
#endif
#ifndef VOXE_TILE_ROT
#define VOXE_TILE_ROT 25
#endif
#ifndef VOXE_TILE_LANEROT
#define VOXE_TILE_LANEROT(lane) (((lane) & 7) + 3 * ((lane) >> 3))
#endif
#ifndef VOXE_TILE_CROT
#define VOXE_TILE_CROT(lane) (lane)
#endif
#ifndef VOXE_TILE_SLOTINC
#define VOXE_TILE_SLOTINC 0
#endif
#ifndef VOXE_TILE_F64MUL
#define VOXE_TILE_F64MUL 0
#endif
#ifndef VOXE_TILE_SLIDE_MIN
#define VOXE_TILE_SLIDE_MIN 1
#endif
// LDS mapping constants, swept on hardware with tools/variants.py + tools/ab_variants.sh (three cameras): channel
// rotation by lane & 3 instead of (lane >> 1) & 3: backward -6.5 %; layer rotation 9 / 17 / 25 ~ equal, 21 +0.4 %;
// plane padding 6 ~ 10 < 2 < 4 << 8 (+35 %: bank aliasing)
template <int KL>
struct Lat {
  static constexpr int kLat = KL;
  static constexpr int kLayerSlots = KL * KL;               // 64 for KL = 8
  static constexpr int kPlane = kRing * KL * KL + VOXE_TILE_PAD;   // padding swept on hardware (KL = 8; 0: 1.05 ms, 8: 0.98, 16: 1.05, 6: 0.953 per 400x400 backward)
  static constexpr int kCentre = KL / 2 - 1;                // lateral cells below the reference ray
  // storage position of lateral cell ab inside its layer: KL = 8 rotates it per layer (64 slots = one bank period, so
  // the same (a, b) of neighbouring layers would share banks); 100 / 144-slot layers are staggered by their size
  static __device__ __forceinline__ int pos(int key, int ab) {
    if constexpr (KL == 8) return (ab + VOXE_TILE_ROT * ring_slot(key)) & 63;
    return ab;
  }
  static __device__ __forceinline__ int rot_of(int key) { return KL == 8 ? VOXE_TILE_ROT * ring_slot(key) : 0; }
  static __device__ __forceinline__ int wrap(int x) { return KL == 8 ? (x & 63) : x; }
};

// r04: the window's address map.  Even KL with four channels: a PARITY-CLASS BANKED layout -- the three parity bits of a
// window voxel (ring slot, lateral a, lateral b) are the low bits of its position and the channel sits right above them,
//   index (doubles) = 32 ((KL/2)^2 (slot >> 1) + (KL/2) (a >> 1) + (b >> 1)) + 8 ch + 4 (slot & 1) + 2 (a & 1) + (b & 1),
// so the LDS bank pair of a double is (ch, slot parity, a parity, b parity): 32 combinations = the 32 bank pairs.  The eight
// corners of a cell have eight different parity classes, so the deposit can choose, per lane and sample, the corner ORDER
// such that instruction (cc, j) of lane L goes to class cc ^ (lane bits 1..3) and channel (j + lane bits 0, 4) & 3: the 32
// lanes of a half-wave hit 32 DIFFERENT bank pairs in every one of the 32 deposit instructions -- conflict free BY
// CONSTRUCTION, whatever the view direction (r01 - r03 rotated corners / channels / layers by lane constants, which left 31 %
// of the LDS cycles to bank conflicts on the axis-aligned bench camera and more on diagonal views).  Other window widths and
// channel counts keep the r03 map (channel planes, per-layer rotation).
#ifndef VOXE_TILE_PCB
#define VOXE_TILE_PCB 1
#endif
#ifndef VOXE_TILE_STRATA
#define VOXE_TILE_STRATA 1   // stratum table per depth segment (voxe_device.hpp: SegDepth) in the window forward: -2 .. -5 % forward time
#endif
#ifndef VOXE_TILE_STRATA_SH
#define VOXE_TILE_STRATA_SH 0    // ... and in the two passes of the view-dependent backward
#endif
#ifndef VOXE_TILE_STRATA_BWD
#define VOXE_TILE_STRATA_BWD 0   // ... and in the SH-0 tile backward: measured equal (0.478 ms either way; 80 B more scratch), off
#endif
#ifndef VOXE_TILE_AXIS_TEMPLATE_SH
#define VOXE_TILE_AXIS_TEMPLATE_SH 0   // ... and the deposit passes of view-dependent grids (MODE 2)
#endif
#ifndef VOXE_TILE_AXIS_TEMPLATE
#define VOXE_TILE_AXIS_TEMPLATE 1   // the backward's march instantiated per window axis (4-channel texel kernels)
#endif
template <int KL, int C>
struct WinMap {
  static constexpr bool kPcb = VOXE_TILE_PCB && (KL % 2 == 0) && C == 4 && (VOXE_TILE_RING % 2 == 0);
  // strides (doubles) of a pair of b, of a, of ring slots: 32 doubles = one (channel, parity class) block per voxel octet
  static constexpr int kSB = 32, kSA = 32 * (KL / 2), kSS = 32 * (KL / 2) * (KL / 2);
  static constexpr int kDoubles = kPcb ? (VOXE_TILE_RING / 2) * kSS : C * Lat<KL>::kPlane;
  // double index of window voxel (ring slot of layer `key`, lateral a, b), window channel ch
  static __device__ __forceinline__ int at(int slot, int key, int a, int b, int ch) {
    if constexpr (kPcb) {
      (void)key;
      return (slot >> 1) * kSS + ((slot & 1) << 2) + (a >> 1) * kSA + ((a & 1) << 1) + (b >> 1) * kSB + (b & 1) + (ch << 3);
    } else {
      return ch * Lat<KL>::kPlane + slot * Lat<KL>::kLayerSlots + Lat<KL>::pos(key, a * KL + b);
    }
  }
};

// wave-wide integer min / max, result wave-uniform (DPP inside rows of 16, readlane across rows).
// Must be called with all 64 lanes active.
__device__ __forceinline__ int wave_min_i32(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));  // row_half_mirror
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));  // row_mirror
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return min(min(a, b), min(c, d));
}
__device__ __forceinline__ int wave_max_i32(int v) { return -wave_min_i32(-v); }

__device__ __forceinline__ float readlane_f32(float x, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane));
}

// wave-uniform geometry of the sliding window
struct Window {
  int m, u, v;          // march axis and the two lateral axes (v = z whenever m != z)
  int sgn;              // +1: keys grow with the voxel index along m, -1: they shrink
  float Au, Bu, Av, Bv; // lateral position of the reference ray as a function of the m index
  int stride_m, stride_u, stride_v;
  int base;             // lowest live layer key
  int ctr;              // lateral cells below the reference ray (Lat<KL>::kCentre)

  __device__ __forceinline__ int off_u(int im) const { return (int)floorf(Au + Bu * (float)im) - ctr; }
  __device__ __forceinline__ int off_v(int im) const { return (int)floorf(Av + Bv * (float)im) - ctr; }
};

}  // namespace voxe
