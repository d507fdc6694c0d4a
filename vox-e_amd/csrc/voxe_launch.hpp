// voxe_launch.hpp -- host-side launcher interface between voxe_api.hip and the kernel files.
#pragma once

#include <hip/hip_runtime.h>

#include "voxe_device.hpp"

namespace voxe {

struct FwdArgs {
  const float *packed, *rays_o, *rays_d, *jitter;
  float *colour, *depth, *acc, *disparity;
  float* ray_state;  // per-ray depth-segment states (nullable): see ray_state_index()
  float* segbuf;     // per-ray per-segment partial results of the segmented forward (nullable)
  // view-dependent grids, image order (r04): (rad_0..2, v) of every sample in the layout of the two-phase tile backward's source
  // buffer (tile, depth segment, sample, lane) -- its source pass then needs no gather (nullable; tile_src_bytes())
  float* sample_fwd = nullptr;
  // space-binned route, view-dependent grids: keep (rad, v) of every sample in the route's own scratch for the backward's source
  // pass (false: inference, VoxeRenderCfg::ray_state_valid = -1)
  bool keep_samples = true;
  // VoxeDispatch::precise_grad (lean tile kernels): per (segment, component, ray) the segment-LOCAL sums (csum[3], asum, dsum) of
  // the forward accumulated in double over the exact products (nullable)
  double* segsum_d = nullptr;
};
struct BwdArgs {
  const float *packed, *rays_o, *rays_d, *jitter, *colour, *depth, *acc, *d_colour, *d_depth, *d_acc;
  float* gpacked;
  bool want_d, want_f;
  const float* ray_state;  // states written by the forward for the same rays (tile backward only)
  float* sample_src = nullptr;   // scratch of the two-phase tile backward of view-dependent grids (tile_src_bytes())
  const float* sample_fwd = nullptr;   // the forward's per-sample (rad, v) for the SAME rays (FwdArgs::sample_fwd), or null
  float* grad_planar = nullptr;        // group-planar staging gradient of the lean deposit passes (tile_planar_bytes()), or null
  // deterministic mode (VoxeRenderCfg::deterministic): 64-bit fixed-point gradient [voxels * C] + 4 floats
  // (max |contribution| of features / density as float bits, then their power-of-two scales); see det_bytes()
  unsigned long long* gdet = nullptr;
  float* det_scale = nullptr;
  const double* segsum_d = nullptr;   // FwdArgs::segsum_d of the forward of the SAME rays (VoxeDispatch::precise_grad), or null
};
// Launch-constant device config + the dispatch decisions of THIS call (VoxeRenderCfg::dispatch, NULL = all defaults): kernels
// take the DevCfg base by value, host-side predicates and launchers read `disp` through the accessors below (0 = default).
struct HostCfg : DevCfg {
  VoxeDispatch disp;
};
inline long long disp_tile_min_rays(const VoxeDispatch& d) { return d.tile_min_rays == 0 ? 8192ll : (d.tile_min_rays < 0 ? 0ll : (long long)d.tile_min_rays); }
inline long long disp_region_min_rays(const VoxeDispatch& d) { return d.region_min_rays == 0 ? 16384ll : (long long)d.region_min_rays; }   // (< 0: route off)
inline bool disp_region_lds_ranks(const VoxeDispatch& d) { return d.region_lds_ranks >= 0; }
inline float disp_region_image_ratio(const VoxeDispatch& d) { return d.region_image_ratio == 0.0f ? 1.3f : (d.region_image_ratio < 0.0f ? 0.0f : d.region_image_ratio); }
inline float disp_or(float v, float dflt) { return v == 0.0f ? dflt : v; }

struct ProbeArgs {
  const float *packed, *rays_o, *rays_d, *jitter;
  int32_t* idx;
  uint8_t* inside;
  float *zvals, *sigma, *rad;
};


// voxe_render.hip
void launch_pack_any(const VoxeGridDesc* gd, float* packed, hipStream_t st);
void launch_unpack_any(const VoxeGridDesc* gd, const float* gpacked, float* d_dens, float* d_feat,
                       int accumulate, int bricked, hipStream_t st);
// fused un-pack + Adam + re-pack (false: channel count without a kernel)
// density-correlation term evaluated inside the grid step: reference densities b, the finalized moment statistics of
// launch_dcl_moments (mean a, mean b, k1, k2 with the weight folded in), see dcl_grad_kernel
// r06: the other regulariser kinds of the edit (sds_trainer.py:494-505: l2_mode / l1_mode; :526-534: feature correlation) are
// per-voxel terms of the same pass: `kind` (VOXE_DREG_*) says what `b` is compared with and how, `fref` / `fk` carry the
// feature term (reference features [N, F], 2 x weight)
struct DclTerm {
  const float* b = nullptr;
  const double* stats = nullptr;
  int kind = 0;
  float k = 0.0f;          // VOXE_DREG_L2: 2 weight / n;  VOXE_DREG_L1: weight / n
  const float* fref = nullptr;
  float fk = 0.0f;
};
bool launch_grid_adam(const VoxeGridDesc* gd, bool bricked, int x_begin, int x_end, float* gpacked, const float* extra_d, const float* extra_f,
                      float* m_d, float* v_d, float* m_f, float* v_f, float lr, float beta1, float beta2, float eps,
                      long long step_d, long long step_f, float* packed_out, hipStream_t st, DclTerm dcl = DclTerm());
// moments + finalize of the density-correlation loss (no gradient kernel): stats = (mean a, mean b, k1 * scale, k2 * scale)
// in `scratch` (returned pointer), loss value to loss_out (nullable)
const double* launch_dcl_moments(const float* a, const float* b, long long n, float grad_scale, float* loss_out, void* scratch,
                                 hipStream_t st);
void launch_fwd(const DevGrid& g, const HostCfg& c, int deg, int diffuse, const FwdArgs& a,
                hipStream_t st);
void launch_bwd(const DevGrid& g, const DevCfg& c, int deg, int diffuse, const BwdArgs& a,
                hipStream_t st);
void launch_probe(const DevGrid& g, const DevCfg& c, int deg, int diffuse, const ProbeArgs& a,
                  hipStream_t st);

// point query (out != nullptr: forward; else backward into gpacked)
void launch_query(const DevGrid& g, int C, const float* packed, const float* points, long long N, float* out,
                  const float* d_out, float* gpacked, bool want_d, bool want_f, hipStream_t st);

// voxe_render_tile.hip: LDS-window backward for image-ordered SH-0 / attention renders
bool tile_bwd_supported(const HostCfg& c, int deg);
// bytes of BwdArgs::sample_src for an image of R rays, width W, S samples (0: that render does not use it)
size_t tile_src_bytes(long long R, int W, int H, int S, int deg, int diffuse, int attn);
void launch_bwd_tile(const DevGrid& g, const HostCfg& c, int deg, int diffuse, const BwdArgs& a, hipStream_t st);
// voxe_render_tile4.hip: the lean LDS-window backward of SH-0 image-ordered renders (8-wide window); launch_bwd_tile hands it
// the launch geometry it computed (blocks, sibling parts, fit bounds)
bool tile4_bwd_supported(const DevGrid& g, const HostCfg& c, const BwdArgs& a, int kl);
void launch_bwd_tile4(const DevGrid& g, const HostCfg& c, const BwdArgs& a, int kl, int nb, int qsplit, float fit_m, float fit_lat,
                      hipStream_t st);
// ... and the forward built the same way (no LDS; per-segment partials into a.segbuf, then the ordinary combine pass)
size_t tile_planar_bytes(long long nvox, int cm);
bool tile4_dep_supported(const DevGrid& g, const HostCfg& c, const BwdArgs& a, int ncu, int cm);
void launch_bwd_tile4_dep(const DevGrid& g, const HostCfg& c, const BwdArgs& a, int ncu, int nb, int qsplit, int ngrp, float fit_m,
                          float fit_lat, hipStream_t st);
bool fwd_tile4_supported(const DevGrid& g, const HostCfg& c, const FwdArgs& a, int cout, int ncm);
void launch_fwd_tile4(const DevGrid& g, const HostCfg& c, const FwdArgs& a, hipStream_t st);
// LDS-staged forward for SH-0 image-ordered renders: writes the per-segment partials into a.segbuf (the caller then runs
// the ordinary combine pass)
bool fwd_tile_supported(const DevGrid& g, const HostCfg& c, int cout, int ncm);
void launch_fwd_tile(const DevGrid& g, const HostCfg& c, const FwdArgs& a, hipStream_t st);
// ... and for view-dependent grids (SH degree 1 - 3, full evaluation): whole texels staged (voxe_render_tilew.hip)
bool fwd_tilew_supported(const DevGrid& g, const HostCfg& c, const FwdArgs& a, int cout, int ncm, int ncu);
void launch_fwd_tilew(const DevGrid& g, const HostCfg& c, int ncm, const FwdArgs& a, hipStream_t st);
// deterministic (ordered-accumulation) backward: single-group image-ordered renders only
bool det_bwd_supported(const DevCfg& c, int deg, int diffuse);
size_t det_bytes(long long nvox, int C);   // [fixed-point gradient | 4 floats], 256-byte aligned parts

// voxe_render_scatter.hip: line-dense scatter backward for unordered rays (SH-0 / attention)
bool packed_scatter_supported(int deg);
void launch_bwd_packed_scatter(const DevGrid& g, const DevCfg& c, int deg, int diffuse, const BwdArgs& a, hipStream_t st);

// voxe_render_region.hip: space-binned render (segments of rays grouped by 8x8x8-cell region; texels and the gradient
// window of a region live in LDS)
bool region_bwd_supported(const DevGrid& g, const HostCfg& c, int deg, int diffuse, bool tiled);
size_t region_scratch_bytes(int X, int Y, int Z, long long R, int S, bool full_sh);   // full_sh: SH degree 1 / 2, not diffuse
void launch_fwd_region(const DevGrid& g, const HostCfg& c, int deg, int diffuse, const FwdArgs& a, void* scratch,
                       hipStream_t st);   // segment tables + forward; leaves the per-segment states in `scratch`
void launch_bwd_region(const DevGrid& g, const DevCfg& c, int deg, int diffuse, const BwdArgs& a, void* scratch,
                       hipStream_t st);   // needs the tables / states of launch_fwd_region for the same rays
// r06: the binning passes of launch_fwd_region on their own, into the tables at `tables` (a buffer of region_scratch_bytes), and
// the override the region launches read while it is set: tables elsewhere than in `scratch` / already filled for these rays
struct RegionBins { void* tables; int prebinned; hipEvent_t after_fwd; };   // after_fwd (or null): recorded between the region forward and the fold pass
extern thread_local const RegionBins* tl_region_bins;
void launch_bin_region(const DevGrid& g, const HostCfg& c, const float* rays_o, const float* rays_d, const float* jitter, void* tables,
                       hipStream_t st, int phase = 3);   // phase 1: clear the counters | 2: the passes behind that | 3: both
// byte offsets of the segment tables inside the region scratch + their dimensions (test aid: voxe_region_debug_layout)
void region_debug_layout(int X, int Y, int Z, long long R, int S, long long out[16]);

// voxe_grid_ops.hip
void launch_cast_rays(int H, int W, float focal, const float* rot, const float* trans, float* rays_o,
                      float* rays_d, hipStream_t st);
void launch_cast_rays_indexed(int H, int W, float focal, const float* poses, int K, const long long* flat_index,
                              long long B, float* rays_o, float* rays_d, hipStream_t st);
void launch_random_subset(long long n, long long count, unsigned long long seed, unsigned long long rng_offset,
                          long long* out, hipStream_t st);
size_t dcl_scratch_bytes(long long n);
// l2 / l1 density regulariser and the feature-correlation regulariser on their own (value + gradient; see voxe.h)
void launch_density_diff(const float* a, const float* b, long long n, int kind, float grad_scale, float* loss_out, float loss_norm,
                         float* d_a, int accumulate, void* scratch, hipStream_t st);
void launch_feature_correlation(const float* f, const float* r, long long nvox, int F, float grad_scale, float* loss_out, float* d_f,
                                int accumulate, void* scratch, hipStream_t st);
void launch_dcl(const float* a, const float* b, long long n, float grad_scale, float* loss_out,
                float* d_a, int accumulate, void* scratch, hipStream_t st);
size_t tv_scratch_bytes(int X, int Y, int Z, int C);
void launch_tv(const float* grid, int X, int Y, int Z, int C, float grad_scale, float* loss_out,
               float* d_grid, int accumulate, void* scratch, hipStream_t st);
void launch_adam(float* param, const float* grad, float* m, float* v, long long n, float lr,
                 float beta1, float beta2, float eps, long long step, hipStream_t st);
void launch_upsample(const float* src, int X, int Y, int Z, int C, float* dst, int X2, int Y2, int Z2,
                     hipStream_t st);
void launch_disparity_bwd(const float* depth, const float* acc, const float* d_disp, const float* d_depth_in,
                          const float* d_acc_in, float* d_depth_out, float* d_acc_out, long long R, hipStream_t st);
void launch_gather_pixels(const float* images, const long long* image_rows, const long long* subset, long long B, int per,
                          int num_images, float* out, hipStream_t st);
size_t l1_scratch_bytes();
void launch_l1_loss_grad(const float* a, const float* b, long long n, float* d_a, float* out2, void* scratch, hipStream_t st);
size_t attn_l1_scratch_bytes();
void launch_attn_masked_l1(const float* render, const float* map, long long n, float* d_render, float* loss_out, void* scratch,
                           hipStream_t st);
void launch_l1_loss_grad_n(const float* a, const float* b, long long n, int renders, float* d_a, float* out2, void* scratch,
                           hipStream_t st);   // (scratch: l1_scratch_bytes() covers two renders)
void launch_recon_batch(long long B, unsigned long long seed, unsigned long long rng_offset, int H, int W, float focal, int K,
                        const float* poses, const float* images, const long long* image_rows, int num_images, long long* subset,
                        float* rays_o, float* rays_d, float* target, hipStream_t st);
double run_clock_probe(int spin, hipStream_t st);   // sustained shader clock in Hz (blocking; 0 on failure)


// voxe_refine.hip: graph construction / minimum cut / connected components of the refinement stage
void launch_graph_build(const float* dens, const float* feat, int X, int Y, int Z, int F, float sigma,
                        int dilate_yz, uint8_t* node_mask, int32_t* cap, hipStream_t stream);
size_t graphcut_scratch_bytes(int X, int Y, int Z);
hipError_t run_graphcut(const uint8_t* node_mask, const int8_t* terminal, int32_t* cap, int X, int Y, int Z,
                        uint8_t* segment, int64_t* flow, void* scratch, hipStream_t stream);
size_t cc_scratch_bytes(int X, int Y, int Z, int k);
void launch_cc_largest_k(const uint8_t* mask, int X, int Y, int Z, int k, int32_t* labels, int32_t* ncomp,
                         void* scratch, hipStream_t stream);

}  // namespace voxe
