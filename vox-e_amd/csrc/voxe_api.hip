// voxe_api.hip -- the extern "C" boundary of libvoxe_hip.so (see include/voxe.h).
// Validation + argument marshalling only; kernels live in voxe_render.hip / voxe_grid_ops.hip.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <unordered_map>

#include "../../include/voxe.h"
#include "voxe_launch.hpp"
#include "voxe_render_common.hpp"

using namespace voxe;

namespace {

// The dispatch of a call: VoxeRenderCfg::dispatch, or the shipped defaults (all fields 0) when it is NULL.  Nothing on the
// render path reads the environment (ABI v7).
const VoxeDispatch kDefaultDispatch = {};
const VoxeDispatch& disp_of(const VoxeRenderCfg* c) { return (c && c->dispatch) ? *c->dispatch : kDefaultDispatch; }
// bwd_mode 1 forces the plain global-atomic backward, 2 the line-dense scatter backward (A/B measurements, debugging)
bool force_scatter_bwd(const VoxeDispatch& d) { return d.bwd_mode == 1; }
bool force_no_tile_bwd(const VoxeDispatch& d) { return d.bwd_mode != 0; }
// tile_two_phase -1: view-dependent grids run the single-kernel channel groups even when the workspace has room for the
// per-sample sources
bool two_phase_disabled(const VoxeDispatch& d) { return d.tile_two_phase < 0; }

// block -> tile mapping (see logical_tile_of()).  Default: image-ordered rays are interleaved over the XCDs (load
// balance wins: neighbouring pixels share their voxels inside a wave anyway); rays in arbitrary order run in bands
// (a batch kept in memory order then gives every XCD's L2 a compact part of the volume: forward -22 % at 32768
// random rays).  VoxeDispatch::tile_map = 1 interleave | 2 band | 3 rows overrides both (A/B runs).
int tile_map_mode(int image_width, const VoxeDispatch& d) {
  if (d.tile_map >= 1 && d.tile_map <= 3) return d.tile_map - 1;
  return image_width > 0 ? 0 : 1;
}

struct Variant {
  int cout, ncoef_mem, C;
  bool attn;
};

int validate(const VoxeGridDesc* g, const VoxeRenderCfg* c, int64_t R, Variant* v) {
  if (!g || !c) return VOXE_ERR_NULL_POINTER;
  if (!g->densities || !g->features) return VOXE_ERR_NULL_POINTER;
  if (g->X <= 0 || g->Y <= 0 || g->Z <= 0 || g->F <= 0 || R < 0 || c->num_samples <= 0)
    return VOXE_ERR_BAD_SHAPE;
  if ((long long)g->X * g->Y * g->Z * (g->F + 1) >= (1LL << 31)) return VOXE_ERR_BAD_SHAPE;
  if ((long long)g->X * g->Y >= (1LL << 24) || (long long)g->Y * g->Z >= (1LL << 24) || g->Z >= (1 << 24)) return VOXE_ERR_BAD_SHAPE;  // 24-bit index multiplies (cell addresses; strides in the window flush)
  if (g->feature_kind == VOXE_FEAT_ATTN) {
    if (g->F != 1) return VOXE_ERR_BAD_SHAPE;
    v->cout = 1; v->ncoef_mem = 1; v->attn = true;
  } else if (g->feature_kind == VOXE_FEAT_SH) {
    if (c->sh_degree < 0 || c->sh_degree > 3) return VOXE_ERR_UNSUPPORTED;
    const int nc = (c->sh_degree + 1) * (c->sh_degree + 1);
    if (g->F != 3 * nc) return VOXE_ERR_BAD_SHAPE;
    v->cout = 3; v->ncoef_mem = nc; v->attn = false;
  } else {
    return VOXE_ERR_UNSUPPORTED;
  }
  v->C = g->F + 1;
  if (g->density_pre_act != VOXE_ACT_IDENTITY && g->density_pre_act != VOXE_ACT_ABS)
    return VOXE_ERR_UNSUPPORTED;
  if (g->density_post_act != VOXE_ACT_IDENTITY && g->density_post_act != VOXE_ACT_RELU &&
      g->density_post_act != VOXE_ACT_SOFTPLUS)
    return VOXE_ERR_UNSUPPORTED;
  if (c->image_width < 0 || (c->image_width > 0 && R % c->image_width != 0))
    return VOXE_ERR_BAD_SHAPE;
  if (c->image_height < 0 || (c->image_height > 0 && (c->image_width <= 0 || R % ((int64_t)c->image_height * c->image_width) != 0)))
    return VOXE_ERR_BAD_SHAPE;
  return VOXE_OK;
}

// voxe_recon_step's paired render (two renders of the same batch with independent jitter streams in ONE launch of 2 B rays):
// set around its calls of the render entry points; the space-binned kernels read DevCfg::pair_R
struct PairSpec { int64_t rays; uint64_t rng_offset; };
thread_local const PairSpec* tl_pair = nullptr;
struct PairScope {
  explicit PairScope(const PairSpec* p) { tl_pair = p; }
  ~PairScope() { tl_pair = nullptr; }
};

void make_dev(const VoxeGridDesc* g, const VoxeRenderCfg* c, int64_t R, const Variant& v, DevGrid* dg,
              HostCfg* dc) {
  dc->disp = disp_of(c);
  dc->pair_R = 0; dc->key0b = dc->key1b = 0;
  if (tl_pair) {
    dc->pair_R = tl_pair->rays;
    dc->key0b = (uint32_t)c->seed ^ ((uint32_t)tl_pair->rng_offset * 0x9E3779B1u);
    dc->key1b = (uint32_t)(c->seed >> 32) ^ (uint32_t)(tl_pair->rng_offset >> 32) ^ 0x7F4A7C15u;
  }
  dg->X = g->X; dg->Y = g->Y; dg->Z = g->Z;
  for (int a = 0; a < 3; ++a) {
    dg->lo[a] = g->aabb_lo[a]; dg->hi[a] = g->aabb_hi[a];
    dg->scale[a] = g->norm_scale[a]; dg->bias[a] = g->norm_bias[a];
  }
  dg->density_scale = g->density_scale;
  dg->pre_act = g->density_pre_act; dg->post_act = g->density_post_act;
  dc->S = c->num_samples;
  dc->near = c->near; dc->far = c->far;
  dc->perturb = c->perturb; dc->lindisp = c->linear_disparity; dc->aabb_clip = c->aabb_clip;
  dc->white = c->white_bkgd; dc->attn = v.attn ? 1 : 0;
  dc->term_eps = c->term_eps;
  dc->key0 = (uint32_t)c->seed ^ ((uint32_t)c->rng_offset * 0x9E3779B1u);
  dc->key1 = (uint32_t)(c->seed >> 32) ^ (uint32_t)(c->rng_offset >> 32) ^ 0x7F4A7C15u;
  dc->image_width = c->image_width;
  dc->image_height = c->image_width > 0 ? (c->image_height > 0 ? c->image_height : (int)(R / c->image_width)) : 0;
  dc->map_mode = tile_map_mode(c->image_width, dc->disp);
  dc->R = R;
  dc->linear_grad = c->linear_grad;
  dc->seg_len = seg_len_for(R);
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// workspace = [ packed grid | packed gradient | per-ray depth-segment states ]
struct WsLayout {
  size_t packed_off, grad_off, state_off, seg_off, prec_off, src_off, fwdval_off, planar_off, det_off, region_off, fwd_total, total, total_with_src;
  bool region;   // the space-binned backward applies to (grid, cfg, R): its scratch is part of the workspace
};
WsLayout ws_layout(const VoxeGridDesc* g, const VoxeRenderCfg* c, int64_t R) {
  const size_t nvox = (size_t)g->X * g->Y * g->Z;
  const size_t bytes = align_up(nvox * (size_t)(g->F + 1) * sizeof(float), 256);
  const int cout = g->feature_kind == VOXE_FEAT_ATTN ? 1 : 3;
  const int nseg = c ? num_segments(c->num_samples, seg_len_for(R)) : 1;
  const size_t state = align_up((size_t)(nseg - 1) * (size_t)(cout + 3) * (size_t)(R > 0 ? R : 0) * sizeof(float), 256);
  // the gradient region also fits the 2x2x2-bricked layout of the scatter backward (dims rounded up to even)
  const size_t bvox = (size_t)((g->X + 1) / 2) * ((g->Y + 1) / 2) * ((g->Z + 1) / 2) * 8;
  const size_t gbytes = align_up(bvox * (size_t)(g->F + 1) * sizeof(float), 256);
  WsLayout l;
  l.packed_off = 0;
  l.grad_off = bytes;
  l.state_off = bytes + gbytes;
  // partial results of the depth-segmented forward: nseg x (cout + 3) floats per ray
  const size_t seg = align_up((size_t)nseg * (size_t)(cout + 3) * (size_t)(R > 0 ? R : 0) * sizeof(float), 256);
  l.seg_off = bytes + gbytes + state;
  l.fwd_total = bytes;  // the forward alone needs only the packed grid (states / segments are used when they fit)
  // VoxeDispatch::precise_grad: segment-local sums of the forward in double, nseg x 5 doubles per ray (lean tile kernels)
  const bool precise = c && disp_of(c).precise_grad > 0 && c->image_width > 0 && cout == 3;
  const size_t prec = precise ? align_up((size_t)nseg * 5 * (size_t)(R > 0 ? R : 0) * sizeof(double), 256) : 0;
  l.prec_off = bytes + gbytes + state + seg;
  l.total = bytes + gbytes + state + seg + prec;
  // per-sample gradient sources of the two-phase backward of view-dependent grids (optional: without it the channel
  // groups re-march the segment)
  l.src_off = l.total;
  const size_t srcb = c ? align_up(tile_src_bytes(R, c->image_width, c->image_height, c->num_samples, c->sh_degree, c->render_diffuse,
                                                  g->feature_kind == VOXE_FEAT_ATTN), 256) : 0;
  // ... and, behind the sources, the forward's per-sample (rad, v) in the same layout (r04: the source pass reads them instead of
  // gathering the wide texels again)
  l.fwdval_off = l.total + srcb;
  // ... and the group-planar staging gradient of the lean deposit passes (r05: [group][voxel][4], 16 bytes per voxel and group)
  l.planar_off = l.total + 2 * srcb;
  const size_t planarb = srcb ? align_up(tile_planar_bytes((long long)nvox, g->F + 1), 256) : 0;
  l.total_with_src = l.total + 2 * srcb + planarb;
  // deterministic mode: the fixed-point gradient and its scales behind everything else
  l.det_off = l.total_with_src;
  if (c && c->deterministic) l.total_with_src += det_bytes((long long)nvox, g->F + 1);
  // space-binned backward (voxe_render_region.hip): per-sample sources + segment tables
  l.region_off = l.total_with_src;
  l.region = false;
  if (c && R > 0 && !c->deterministic && !force_no_tile_bwd(disp_of(c))) {
    Variant v;
    if (validate(g, c, R, &v) == VOXE_OK) {
      DevGrid dg; HostCfg dc;
      make_dev(g, c, R, v, &dg, &dc);
      const bool tiled = tile_bwd_supported(dc, c->sh_degree);
      l.region = region_bwd_supported(dg, dc, c->sh_degree, c->render_diffuse, tiled);
      if (l.region)
        l.total_with_src += region_scratch_bytes(g->X, g->Y, g->Z, R, c->num_samples,
                                                 g->feature_kind != VOXE_FEAT_ATTN && c->sh_degree > 0 && !c->render_diffuse);
    }
  }
  return l;
}

int finish() { return hipGetLastError() == hipSuccess ? VOXE_OK : VOXE_ERR_LAUNCH; }

// VoxeDispatch::precise_grad: the double segment sums exist only when the forward of these rays is
// render_fwd_tile4_kernel<3, true> -- launch_fwd_t's condition, restated ONCE here for the forward, the re-march and the backward:
// a depth-segmented march (nseg > 1), one segment per task, the lean forward's own conditions.  Everywhere else (S <= one segment,
// fwd_segments_per_thread > 1, caller jitter, AABB clip, ...) the backward runs its default suffix arithmetic.
bool precise_sums_apply(const WsLayout& l, const DevGrid& dg, const HostCfg& dc, const VoxeRenderCfg* cfg, const float* jitter) {
  if (!(l.total > l.prec_off) || cfg->sh_degree != 0 || dc.attn) return false;
  if (num_segments(cfg->num_samples, dc.seg_len) <= 1 || dc.disp.fwd_segments_per_thread > 1) return false;
  FwdArgs probe{};
  probe.jitter = jitter;
  probe.segbuf = reinterpret_cast<float*>(1);   // (the segment partials live in the same workspace tier as the sums)
  return fwd_tile4_supported(dg, dc, probe, 3, 1);
}

// ---- what the last forward left in a workspace --------------------------------------------------------------------
// VoxeRenderCfg::ray_state_valid = 1 is a CLAIM by the caller ("this workspace still holds what voxe_render_fwd wrote for exactly
// these rays / cfg / jitter").  The library checks the claim instead of trusting it: every forward leaves a host-side record of
// what it rendered into the workspace it was given (keyed by the workspace address; calls on one workspace are ordered by the
// caller's stream, the record by the order of the calls), and a backward whose claim does not match the record re-marches the rays
// exactly as if it had been called with 0.  In particular a forward with ray_state_valid = -1 (inference: per-sample values not
// kept) followed by a backward with 1 is served by a re-march, not by stale per-sample values.  The record is a cache of facts
// about buffers, not dispatch state: losing it (more than kMaxStamps workspaces) only costs a re-march.
struct FwdStamp {
  const void *densities, *features, *rays_o, *rays_d, *jitter;
  int64_t R, pair_rays;
  uint64_t seed, rng_offset, pair_offset, wsbytes;
  int32_t X, Y, Z, F, feature_kind, pre_act, post_act, S, perturb, lindisp, clip, white, deg, diffuse, width, height, det, kept;
  float near_, far_, density_scale, lo[3], hi[3];
  // the dispatch fields (VoxeDispatch, field by field: the struct's padding is the caller's)
  int64_t d_tile_min_rays, d_region_min_rays;
  int32_t d_bwd_mode, d_tile_map, d_two_phase, d_qsplit, d_kl, d_fwd_window, d_fwd_spt, d_lean, d_precise, d_lds_ranks, d_phases;
  float d_fit_m, d_fit_lat, d_ffit_lat, d_ffit_m, d_zdom, d_max_adv, d_image_ratio;
};
FwdStamp make_stamp(const VoxeGridDesc* g, const VoxeRenderCfg* c, const float* rays_o, const float* rays_d, int64_t R,
                    const float* jitter, size_t wsbytes) {
  FwdStamp k;
  memset(&k, 0, sizeof(k));   // (padding included: stamps are compared with memcmp)
  const int64_t pair_rays = tl_pair ? tl_pair->rays : 0;
  const uint64_t pair_offset = tl_pair ? tl_pair->rng_offset : 0;
  k.densities = g->densities; k.features = g->features; k.rays_o = rays_o; k.rays_d = rays_d; k.jitter = jitter;
  k.R = R; k.pair_rays = pair_rays; k.seed = c->seed; k.rng_offset = c->rng_offset; k.pair_offset = pair_offset; k.wsbytes = wsbytes;
  k.X = g->X; k.Y = g->Y; k.Z = g->Z; k.F = g->F; k.feature_kind = g->feature_kind; k.pre_act = g->density_pre_act;
  k.post_act = g->density_post_act; k.S = c->num_samples; k.perturb = c->perturb; k.lindisp = c->linear_disparity;
  k.clip = c->aabb_clip; k.white = c->white_bkgd; k.deg = c->sh_degree; k.diffuse = c->render_diffuse; k.width = c->image_width;
  k.height = c->image_height; k.det = c->deterministic;
  k.near_ = c->near; k.far_ = c->far; k.density_scale = g->density_scale;
  for (int a = 0; a < 3; ++a) { k.lo[a] = g->aabb_lo[a]; k.hi[a] = g->aabb_hi[a]; }
  const VoxeDispatch& d = disp_of(c);
  k.d_tile_min_rays = d.tile_min_rays; k.d_region_min_rays = d.region_min_rays; k.d_bwd_mode = d.bwd_mode; k.d_tile_map = d.tile_map;
  k.d_two_phase = d.tile_two_phase; k.d_qsplit = d.tile_qsplit; k.d_kl = d.tile_kl; k.d_fwd_window = d.fwd_window;
  k.d_fwd_spt = d.fwd_segments_per_thread; k.d_lean = d.tile_lean; k.d_precise = d.precise_grad; k.d_lds_ranks = d.region_lds_ranks; k.d_phases = d.tile_phases;
  k.d_fit_m = d.tile_fit_m; k.d_fit_lat = d.tile_fit_lat; k.d_ffit_lat = d.fwd_fit_lat; k.d_ffit_m = d.fwd_fit_m; k.d_zdom = d.fwd_zdom;
  k.d_max_adv = d.fwd_max_adv; k.d_image_ratio = d.region_image_ratio;
  return k;
}
constexpr size_t kMaxStamps = 256;
std::mutex g_stamp_mu;
std::unordered_map<const void*, FwdStamp> g_stamps;
void record_stamp(const void* workspace, const FwdStamp& k) {
  std::lock_guard<std::mutex> lock(g_stamp_mu);
  if (g_stamps.size() >= kMaxStamps && g_stamps.find(workspace) == g_stamps.end()) g_stamps.clear();
  g_stamps[workspace] = k;
}
void forget_stamp(const void* workspace) {
  std::lock_guard<std::mutex> lock(g_stamp_mu);
  g_stamps.erase(workspace);
}
// the tensor at `param` is about to be rewritten in place: no workspace's record of a forward over it describes the grid any more
void forget_stamps_of(const void* param) {
  if (!param) return;
  std::lock_guard<std::mutex> lock(g_stamp_mu);
  for (auto it = g_stamps.begin(); it != g_stamps.end();)
    it = (it->second.densities == param || it->second.features == param) ? g_stamps.erase(it) : std::next(it);
}
// does `workspace` hold the forward of exactly this render, per-sample values included?
bool stamp_matches(const void* workspace, FwdStamp k) {
  k.kept = 1;
  std::lock_guard<std::mutex> lock(g_stamp_mu);
  const auto it = g_stamps.find(workspace);
  return it != g_stamps.end() && memcmp(&it->second, &k, sizeof(k)) == 0;
}

// ---- per-phase timing (voxe_profile_*) ----------------------------------------------------------
enum Phase { PH_PACK = 0, PH_FWD, PH_MEMSET, PH_BWD, PH_UNPACK, PH_COUNT };
struct Profiler {
  static constexpr int kMax = 512;
  bool on = false, created = false;
  hipEvent_t ev[kMax][2];
  int phase[kMax];
  int count = 0, dropped = 0;
} g_prof;

struct PhaseTimer {  // records start/stop events around one phase when profiling is enabled
  int slot = -1;
  hipStream_t st;
  PhaseTimer(Phase ph, hipStream_t s) : st(s) {
    if (!g_prof.on) return;
    if (g_prof.count >= Profiler::kMax) { ++g_prof.dropped; return; }
    slot = g_prof.count++;
    g_prof.phase[slot] = ph;
    (void)hipEventRecord(g_prof.ev[slot][0], st);
  }
  ~PhaseTimer() {
    if (slot >= 0) (void)hipEventRecord(g_prof.ev[slot][1], st);
  }
};

}  // namespace

extern "C" {

int voxe_abi_version(void) { return VOXE_ABI_VERSION; }

const char* voxe_strerror(int status) {
  switch (status) {
    case VOXE_OK: return "ok";
    case VOXE_ERR_NULL_POINTER: return "null pointer";
    case VOXE_ERR_BAD_SHAPE: return "bad shape";
    case VOXE_ERR_UNSUPPORTED: return "unsupported activation / mode";
    case VOXE_ERR_WORKSPACE: return "workspace missing or too small";
    case VOXE_ERR_LAUNCH: return "kernel launch failed";
    case VOXE_ERR_NO_DEVICE: return "no gfx950 HIP device";
    default: return "unknown voxe status";
  }
}

int voxe_device_check(char* name, size_t name_len) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return VOXE_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return VOXE_ERR_NO_DEVICE;
  if (name && name_len) {
    strncpy(name, prop.gcnArchName, name_len - 1);
    name[name_len - 1] = 0;
  }
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? VOXE_OK : VOXE_ERR_NO_DEVICE;
}

int voxe_profile_enable(int32_t on) {
  if (on && !g_prof.created) {
    for (int i = 0; i < Profiler::kMax; ++i)
      for (int j = 0; j < 2; ++j)
        if (hipEventCreate(&g_prof.ev[i][j]) != hipSuccess) return VOXE_ERR_LAUNCH;
    g_prof.created = true;
  }
  g_prof.on = on != 0;
  g_prof.count = 0;
  g_prof.dropped = 0;
  return VOXE_OK;
}

int voxe_profile_read(VoxeProfile* out) {
  if (!out) return VOXE_ERR_NULL_POINTER;
  memset(out, 0, sizeof(*out));
  double* ms[PH_COUNT] = {&out->ms_pack, &out->ms_fwd, &out->ms_memset, &out->ms_bwd, &out->ms_unpack};
  int32_t* n[PH_COUNT] = {&out->n_pack, &out->n_fwd, &out->n_memset, &out->n_bwd, &out->n_unpack};
  for (int i = 0; i < g_prof.count; ++i) {
    if (hipEventSynchronize(g_prof.ev[i][1]) != hipSuccess) return VOXE_ERR_LAUNCH;
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof.ev[i][0], g_prof.ev[i][1]) != hipSuccess) return VOXE_ERR_LAUNCH;
    *ms[g_prof.phase[i]] += (double)t;
    *n[g_prof.phase[i]] += 1;
  }
  out->n_dropped = g_prof.dropped;
  g_prof.count = 0;
  g_prof.dropped = 0;
  return VOXE_OK;
}

int voxe_cast_rays(int32_t H, int32_t W, float focal, const float* rot, const float* trans,
                   float* rays_o, float* rays_d, void* stream) {
  if (!rot || !trans || !rays_o || !rays_d) return VOXE_ERR_NULL_POINTER;
  if (H <= 0 || W <= 0) return VOXE_ERR_BAD_SHAPE;
  launch_cast_rays(H, W, focal, rot, trans, rays_o, rays_d, (hipStream_t)stream);
  return finish();
}

int voxe_cast_rays_indexed(int32_t H, int32_t W, float focal, const float* poses, int32_t K,
                           const int64_t* flat_index, int64_t B, float* rays_o, float* rays_d, void* stream) {
  if (!poses || !flat_index || !rays_o || !rays_d) return VOXE_ERR_NULL_POINTER;
  if (H <= 0 || W <= 0 || K <= 0 || B < 0) return VOXE_ERR_BAD_SHAPE;
  if (B == 0) return VOXE_OK;
  launch_cast_rays_indexed(H, W, focal, poses, K, (const long long*)flat_index, B, rays_o, rays_d,
                           (hipStream_t)stream);
  return finish();
}

int voxe_random_subset(int64_t n, int64_t count, uint64_t seed, uint64_t rng_offset, int64_t* out, void* stream) {
  if (!out) return VOXE_ERR_NULL_POINTER;
  if (n <= 0 || n > (1ll << 31) || count < 0 || count > n) return VOXE_ERR_BAD_SHAPE;
  if (count == 0) return VOXE_OK;
  launch_random_subset(n, count, seed, rng_offset, (long long*)out, (hipStream_t)stream);
  return finish();
}

size_t voxe_workspace_bytes(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg, int64_t R) {
  if (!grid || grid->X <= 0 || grid->Y <= 0 || grid->Z <= 0 || grid->F <= 0) return 0;
  const WsLayout l = ws_layout(grid, cfg, R);
  // ray_state_valid = -1 (inference: no backward of these rays follows): the packed grid and the segmented forward's states /
  // partials -- none of the backward's per-sample sources, staging gradient or binning scratch (gigabytes at large R)
  if (cfg && cfg->ray_state_valid < 0) return l.total;
  return l.total_with_src;
}

int voxe_render_fwd(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg, const float* rays_o,
                    const float* rays_d, int64_t R, const float* jitter, float* colour, float* depth,
                    float* acc, float* disparity, void* workspace, size_t workspace_bytes,
                    void* stream) {
  Variant v;
  const int st = validate(grid, cfg, R, &v);
  if (st) return st;
  if (R > 0 && (!rays_o || !rays_d || !colour)) return VOXE_ERR_NULL_POINTER;
  const WsLayout l = ws_layout(grid, cfg, R);
  if (!workspace || workspace_bytes < l.fwd_total) return VOXE_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  float* packed = (float*)((char*)workspace + l.packed_off);
  if (!cfg->reuse_packed_grid) { PhaseTimer t(PH_PACK, s); launch_pack_any(grid, packed, s); }
  if (R == 0) return finish();
  {
    // what this workspace holds from now on (a backward's ray_state_valid = 1 is checked against it)
    FwdStamp k = make_stamp(grid, cfg, rays_o, rays_d, R, jitter, workspace_bytes);
    k.kept = cfg->ray_state_valid >= 0 ? 1 : 0;
    record_stamp(workspace, k);
  }
  DevGrid dg; HostCfg dc;
  make_dev(grid, cfg, R, v, &dg, &dc);
  // depth-segment states for the segmented backward, when that backward applies and the workspace holds them
  float* state = nullptr;
  const bool tiled = tile_bwd_supported(dc, cfg->sh_degree) && !force_no_tile_bwd(dc.disp);
  const bool packed_bwd = !tiled && packed_scatter_supported(cfg->sh_degree) && !force_scatter_bwd(dc.disp);
  if ((tiled || packed_bwd) && workspace_bytes >= l.total) state = (float*)((char*)workspace + l.state_off);
  float* segbuf = workspace_bytes >= l.total ? (float*)((char*)workspace + l.seg_off) : nullptr;
  FwdArgs a{packed, rays_o, rays_d, jitter, colour, depth, acc, disparity, state, segbuf};
  a.keep_samples = cfg->ray_state_valid >= 0;
  if (segbuf && cfg->ray_state_valid >= 0 && precise_sums_apply(l, dg, dc, cfg, jitter)) a.segsum_d = (double*)((char*)workspace + l.prec_off);
  if (cfg->ray_state_valid >= 0 && tiled && l.fwdval_off > l.src_off && workspace_bytes >= l.total_with_src && !two_phase_disabled(dc.disp))
    a.sample_fwd = (float*)((char*)workspace + l.fwdval_off);   // (what render_bwd_common's two-phase backward will read)
  if (l.region && workspace_bytes >= l.total_with_src) {
    // space-binned path (unordered / sparse rays): forward through the region kernels; the segment tables and per-segment
    // states stay in the workspace for the backward of the same call (cfg->ray_state_valid)
    PhaseTimer t(PH_FWD, s);
    launch_fwd_region(dg, dc, cfg->sh_degree, cfg->render_diffuse, a, (char*)workspace + l.region_off, s);
    return finish();
  }
  { PhaseTimer t(PH_FWD, s); launch_fwd(dg, dc, cfg->sh_degree, cfg->render_diffuse, a, s); }
  return finish();
}

}  // extern "C" (reopened below)

namespace {
// memset (optional) + backward kernel into the workspace's packed gradient region; *bricked = layout of that region
int render_bwd_common(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg, const Variant& v, const float* rays_o,
                      const float* rays_d, int64_t R, const float* jitter, const float* colour, const float* depth,
                      const float* acc, const float* d_colour, const float* d_depth, const float* d_acc, bool want_d,
                      bool want_f, bool zero_first, bool* bricked, void* workspace, size_t workspace_bytes,
                      hipStream_t s, void* grad_workspace = nullptr, size_t grad_workspace_bytes = 0) {
  const WsLayout l = ws_layout(grid, cfg, R);
  if (!workspace || workspace_bytes < l.total) return VOXE_ERR_WORKSPACE;
  if (grad_workspace && grad_workspace_bytes < l.state_off) return VOXE_ERR_WORKSPACE;   // (packed grid + gradient region)
  float* packed = (float*)((char*)workspace + l.packed_off);
  float* gpacked = (float*)((char*)(grad_workspace ? grad_workspace : workspace) + l.grad_off);
  float* state = (float*)((char*)workspace + l.state_off);
  if (!cfg->reuse_packed_grid) { PhaseTimer t(PH_PACK, s); launch_pack_any(grid, packed, s); }
  if (zero_first) {
    PhaseTimer t(PH_MEMSET, s);
    if (hipMemsetAsync(gpacked, 0, l.state_off - l.grad_off, s) != hipSuccess) return VOXE_ERR_LAUNCH;
  }
  *bricked = false;
  if (R > 0) {
    DevGrid dg; HostCfg dc;
    make_dev(grid, cfg, R, v, &dg, &dc);
    const bool tiled = tile_bwd_supported(dc, cfg->sh_degree) && !force_no_tile_bwd(dc.disp);
    const bool packed_bwd = !tiled && packed_scatter_supported(cfg->sh_degree) && !force_scatter_bwd(dc.disp);
    BwdArgs a{packed, rays_o, rays_d, jitter, colour, depth, acc, d_colour, d_depth, d_acc, gpacked,
              want_d, want_f, (tiled || packed_bwd) ? state : nullptr};
    const bool precise = precise_sums_apply(l, dg, dc, cfg, jitter);
    if (precise) a.segsum_d = (const double*)((char*)workspace + l.prec_off);
    const bool two_phase = tiled && l.fwdval_off > l.src_off && workspace_bytes >= l.total_with_src && !two_phase_disabled(dc.disp);
    if (two_phase) {
      a.sample_src = (float*)((char*)workspace + l.src_off);
      // the forward of these rays (voxe_render_fwd with this workspace, or the re-march below) wrote its per-sample values when it
      // ran the depth-segmented kernel: launch_fwd_t's condition
      if (num_segments(cfg->num_samples, dc.seg_len) > 1) a.sample_fwd = (const float*)((char*)workspace + l.fwdval_off);
      a.grad_planar = (float*)((char*)workspace + l.planar_off);
    }
    if (cfg->deterministic) {
      if (!det_bwd_supported(dc, cfg->sh_degree, cfg->render_diffuse)) return VOXE_ERR_UNSUPPORTED;
      if (workspace_bytes < l.total_with_src) return VOXE_ERR_WORKSPACE;
      a.gdet = (unsigned long long*)((char*)workspace + l.det_off);
      a.det_scale = (float*)((char*)workspace + l.det_off + det_bytes((long long)grid->X * grid->Y * grid->Z, grid->F + 1) - 256);
      a.ray_state = state;
    }
    const bool det = cfg->deterministic != 0;
    const bool region = l.region && !det && workspace_bytes >= l.total_with_src;
    // the caller's claim that the workspace holds this render's forward is only believed when the forward's record says so
    const FwdStamp stamp = make_stamp(grid, cfg, rays_o, rays_d, R, jitter, workspace_bytes);
    const bool states_valid = cfg->ray_state_valid > 0 && stamp_matches(workspace, stamp);
    if (!states_valid) {
      // (the re-march below writes everything a backward reads: from here on the workspace holds THIS render)
      FwdStamp k = stamp;
      k.kept = 1;
      record_stamp(workspace, k);
    }
    if (region && !states_valid) {
      // the caller's workspace does not hold this call's segment tables / states: rebuild them (no outputs)
      PhaseTimer t(PH_FWD, s);
      FwdArgs f{packed, rays_o, rays_d, jitter, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
      launch_fwd_region(dg, dc, cfg->sh_degree, cfg->render_diffuse, f, (char*)workspace + l.region_off, s);
    } else if ((tiled || packed_bwd || det) && !states_valid) {
      // the caller's workspace does not hold this call's forward states: re-march to rebuild them
      PhaseTimer t(PH_FWD, s);
      FwdArgs f{packed, rays_o, rays_d, jitter, nullptr, nullptr, nullptr, nullptr, state,
                (float*)((char*)workspace + l.seg_off)};
      if (two_phase) f.sample_fwd = (float*)((char*)workspace + l.fwdval_off);
      if (precise) f.segsum_d = (double*)((char*)workspace + l.prec_off);
      launch_fwd(dg, dc, cfg->sh_degree, cfg->render_diffuse, f, s);
    }
    PhaseTimer t(PH_BWD, s);
    if (det) {
      // (the fixed-point region must start cleared: the finalize pass leaves it so, the first use clears it here)
      if (hipMemsetAsync(a.gdet, 0, det_bytes((long long)grid->X * grid->Y * grid->Z, grid->F + 1), s) != hipSuccess)
        return VOXE_ERR_LAUNCH;
      launch_bwd_tile(dg, dc, cfg->sh_degree, cfg->render_diffuse, a, s);
    } else if (region) {
      launch_bwd_region(dg, dc, cfg->sh_degree, cfg->render_diffuse, a, (char*)workspace + l.region_off, s);
    } else if (tiled)
      launch_bwd_tile(dg, dc, cfg->sh_degree, cfg->render_diffuse, a, s);
    else if (packed_bwd) {
      launch_bwd_packed_scatter(dg, dc, cfg->sh_degree, cfg->render_diffuse, a, s);
      *bricked = !cfg->linear_grad;  // that kernel accumulates into the bricked layout unless asked otherwise
    } else
      launch_bwd(dg, dc, cfg->sh_degree, cfg->render_diffuse, a, s);
  }
  return VOXE_OK;
}
}  // namespace

extern "C" {

int voxe_render_bwd(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg, const float* rays_o,
                    const float* rays_d, int64_t R, const float* jitter, const float* colour,
                    const float* depth, const float* acc, const float* d_colour, const float* d_depth,
                    const float* d_acc, float* d_densities, float* d_features, int32_t accumulate,
                    void* workspace, size_t workspace_bytes, void* stream) {
  Variant v;
  const int st = validate(grid, cfg, R, &v);
  if (st) return st;
  if (R > 0 && (!rays_o || !rays_d || !colour || !depth || !acc || !d_colour))
    return VOXE_ERR_NULL_POINTER;
  if (!d_densities && !d_features) return VOXE_OK;
  hipStream_t s = (hipStream_t)stream;
  bool bricked = false;
  const int rc = render_bwd_common(grid, cfg, v, rays_o, rays_d, R, jitter, colour, depth, acc, d_colour, d_depth, d_acc,
                                   d_densities != nullptr, d_features != nullptr, true, &bricked, workspace,
                                   workspace_bytes, s);
  if (rc) return rc;
  {
    const WsLayout l = ws_layout(grid, cfg, R);
    PhaseTimer t(PH_UNPACK, s);
    launch_unpack_any(grid, (const float*)((char*)workspace + l.grad_off), d_densities, d_features, accumulate,
                      bricked ? 1 : 0, s);
  }
  return finish();
}

int voxe_render_bwd_acc(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg, const float* rays_o, const float* rays_d,
                        int64_t R, const float* jitter, const float* colour, const float* depth, const float* acc,
                        const float* d_colour, const float* d_depth, const float* d_acc, int32_t want_densities,
                        int32_t want_features, int32_t zero_first, int32_t* grad_layout, void* workspace,
                        size_t workspace_bytes, void* stream) {
  Variant v;
  const int st = validate(grid, cfg, R, &v);
  if (st) return st;
  if (!grad_layout || (R > 0 && (!rays_o || !rays_d || !colour || !depth || !acc || !d_colour)))
    return VOXE_ERR_NULL_POINTER;
  bool bricked = false;
  const int rc = render_bwd_common(grid, cfg, v, rays_o, rays_d, R, jitter, colour, depth, acc, d_colour, d_depth, d_acc,
                                   want_densities != 0, want_features != 0, zero_first != 0, &bricked, workspace,
                                   workspace_bytes, (hipStream_t)stream);
  if (rc) return rc;
  *grad_layout = R > 0 ? (bricked ? VOXE_GRAD_BRICKED : VOXE_GRAD_LINEAR) : VOXE_GRAD_ANY;
  return finish();
}

int voxe_render_bwd_acc_into(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg, const float* rays_o, const float* rays_d,
                             int64_t R, const float* jitter, const float* colour, const float* depth, const float* acc,
                             const float* d_colour, const float* d_depth, const float* d_acc, int32_t want_densities,
                             int32_t want_features, int32_t zero_first, int32_t* grad_layout, void* workspace,
                             size_t workspace_bytes, void* grad_workspace, size_t grad_workspace_bytes, void* stream) {
  Variant v;
  const int st = validate(grid, cfg, R, &v);
  if (st) return st;
  if (!grad_layout || (R > 0 && (!rays_o || !rays_d || !colour || !depth || !acc || !d_colour)))
    return VOXE_ERR_NULL_POINTER;
  bool bricked = false;
  const int rc = render_bwd_common(grid, cfg, v, rays_o, rays_d, R, jitter, colour, depth, acc, d_colour, d_depth, d_acc,
                                   want_densities != 0, want_features != 0, zero_first != 0, &bricked, workspace,
                                   workspace_bytes, (hipStream_t)stream, grad_workspace, grad_workspace_bytes);
  if (rc) return rc;
  *grad_layout = R > 0 ? (bricked ? VOXE_GRAD_BRICKED : VOXE_GRAD_LINEAR) : VOXE_GRAD_ANY;
  return finish();
}

int voxe_render_bwd_layout(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg, int64_t R) {
  Variant v;
  const int st = validate(grid, cfg, R, &v);
  if (st) return st < -1 ? st : VOXE_ERR_BAD_SHAPE;
  if (R == 0) return VOXE_GRAD_ANY;
  DevGrid dg; HostCfg dc;
  make_dev(grid, cfg, R, v, &dg, &dc);
  if (cfg->deterministic) return VOXE_GRAD_LINEAR;
  const bool tiled = tile_bwd_supported(dc, cfg->sh_degree) && !force_no_tile_bwd(dc.disp);
  const bool packed_bwd = !tiled && packed_scatter_supported(cfg->sh_degree) && !force_scatter_bwd(dc.disp);
  if (ws_layout(grid, cfg, R).region) return VOXE_GRAD_LINEAR;   // (given the workspace voxe_workspace_bytes asks for)
  return (packed_bwd && !cfg->linear_grad) ? VOXE_GRAD_BRICKED : VOXE_GRAD_LINEAR;
}

int voxe_render_route(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg, int64_t R) {
  Variant v;
  const int st = validate(grid, cfg, R, &v);
  if (st) return st < -1 ? st : VOXE_ERR_BAD_SHAPE;
  if (R == 0) return VOXE_ROUTE_NONE;
  DevGrid dg; HostCfg dc;
  make_dev(grid, cfg, R, v, &dg, &dc);
  if (cfg->deterministic) return VOXE_ROUTE_DETERMINISTIC;
  if (ws_layout(grid, cfg, R).region) return VOXE_ROUTE_REGION;
  if (tile_bwd_supported(dc, cfg->sh_degree) && !force_no_tile_bwd(dc.disp)) return VOXE_ROUTE_TILE;
  if (packed_scatter_supported(cfg->sh_degree) && !force_scatter_bwd(dc.disp)) return VOXE_ROUTE_PACKED_SCATTER;
  return VOXE_ROUTE_SCATTER;
}

int voxe_disparity_bwd(const float* depth, const float* acc, const float* d_disparity, const float* d_depth_in,
                       const float* d_acc_in, float* d_depth_out, float* d_acc_out, int64_t R, void* stream) {
  if (R < 0) return VOXE_ERR_BAD_SHAPE;
  if (R > 0 && (!depth || !acc || !d_disparity || !d_depth_out || !d_acc_out)) return VOXE_ERR_NULL_POINTER;
  launch_disparity_bwd(depth, acc, d_disparity, d_depth_in, d_acc_in, d_depth_out, d_acc_out, R, (hipStream_t)stream);
  return finish();
}

namespace {
#ifndef VOXE_RECON_PAIRED
#define VOXE_RECON_PAIRED 1
#endif
inline size_t up256b(size_t x) { return (x + 255) / 256 * 256; }
struct ReconLayout { size_t subset, rays_o, rays_d, target, out[2], d_colour[2], partial, out2, d_colour2, batch2, total; };
ReconLayout recon_layout(int64_t B) {
  ReconLayout l;
  size_t off = 0;
  const size_t b = (size_t)(B > 0 ? B : 0);
  l.subset = off; off += up256b(b * sizeof(int64_t));
  l.rays_o = off; off += up256b(b * 3 * sizeof(float));
  l.rays_d = off; off += up256b(b * 3 * sizeof(float));
  l.target = off; off += up256b(b * 3 * sizeof(float));
  l.batch2 = off;   // (the same four buffers again, behind everything else: the batch voxe_recon_prefetch assembles ahead of its iteration)
  for (int i = 0; i < 2; ++i) { l.out[i] = off; off += up256b(b * 5 * sizeof(float)); }        // colour [B,3] | depth [B] | acc [B]
  for (int i = 0; i < 2; ++i) { l.d_colour[i] = off; off += up256b(b * 3 * sizeof(float)); }
  l.partial = off; off += up256b(l1_scratch_bytes());
  // paired render: colour [2B,3] | depth [2B] | acc [2B], d_colour [2B,3]
  l.out2 = off; off += up256b(2 * b * 5 * sizeof(float));
  l.d_colour2 = off; off += up256b(2 * b * 3 * sizeof(float));
  const size_t batch_bytes = l.batch2;
  l.batch2 = off; off += batch_bytes;
  l.total = off;
  return l;
}

// ---- voxe_recon_prefetch: the batch and the segment tables of the NEXT iteration, assembled on a side stream ---------------------
// Batch assembly and binning read the cameras, the images and the jitter streams -- never the grid -- so iteration i + 1's can run
// while iteration i's backward and grid step occupy the memory system (binning is index arithmetic; the grid step is a pure
// HBM stream).  Two sets of tables / batch buffers alternate: set 0 = (region scratch of `workspace`, front of `scratch`), set 1 =
// (the same offsets of `workspace2`, ReconLayout::batch2).  A prefetch is a HINT recorded per workspace: the step that follows
// uses it iff the arguments that decide the batch and the tables are the ones it was made for (ReconKey, compared field by
// field); otherwise it waits for the side stream and assembles its own, exactly as without the hint.
struct ReconKey {
  FwdStamp fwd;                  // grid geometry, cfg, dispatch, jitter streams, ray buffers of the paired render
  const void *poses, *image_rows, *images, *workspace2, *scratch;
  uint64_t subset_offset, ws2bytes, scbytes;
  int64_t batch;
  int32_t H, W, K, num_images;
  float focal;
};
struct ReconPending { bool valid = false; int slot = 0; ReconKey key; hipEvent_t done = nullptr; };
struct ReconSide {
  hipStream_t side = nullptr;
  hipEvent_t fork0 = nullptr;    // on the caller's stream, at the start of a step: from here on the idle set of tables / batch buffers may be rewritten
  hipEvent_t fork = nullptr;     // ... behind the step's forward (fold pass): from here on the segment pass of the hint may run
  bool forked = false;
  int last_slot = 0;             // set the last step of this workspace rendered from
  ReconPending pend;
};
std::mutex g_recon_mu;
std::unordered_map<const void*, ReconSide> g_recon;     // by workspace
constexpr size_t kMaxReconSides = 16;
int64_t g_recon_stats[3] = {0, 0, 0};                   // hints issued | taken by the following step | dropped by it (voxe_recon_prefetch_stats)
#ifndef VOXE_RECON_SIDE_LOW
#define VOXE_RECON_SIDE_LOW 1
#endif
#ifndef VOXE_RECON_FORK
#define VOXE_RECON_FORK 1        // where a step lets the prefetch of its successor start: 0 at its own start | 1 behind its forward (fold) |
                                 // 2 batch assembly + clearing of the counters at its start, the segment pass behind its forward |
                                 // 3 like 2, the segment pass already behind the region forward (beside the fold pass).
                                 // Measured (160^3, 2 x 32768 rays, ms per iteration of the trainer's loop, which the host paced at the
                                 // time; profiles/r06_recon_prefetch.txt): no hint 0.805, 1 -> 0.725, 0 -> 0.79, 2 -> 0.78, 3 -> 0.79.
                                 // (A device-paced loop gains ~1 % from schedule 1 and loses with the others: device work is conserved.)  The backward's blocks hold every SIMD's registers and 144 of 160 KB
                                 // of LDS: launched behind it (1) the segment pass (1024-thread blocks, 72 KB) waits until the backward
                                 // drains and then shares the machine with the grid step -- an HBM stream that leaves the CUs idle;
                                 // launched in front of it (2) the segment pass runs at once but the column scan behind it starves and the
                                 // backward loses a quarter of its slots for as long.  Stream priority moves nothing (hi / lo equal).
#endif
ReconKey recon_key(const VoxeGridDesc* grid, const VoxeRenderCfg* pc, const VoxeRenderCfg* cfg, const VoxeReconStep* rs, const float* rays_o,
                   const float* rays_d, size_t workspace_bytes, const void* workspace2, size_t workspace2_bytes, const void* scratch,
                   size_t scratch_bytes) {
  ReconKey k;
  memset(&k, 0, sizeof(k));
  k.fwd = make_stamp(grid, pc, rays_o, rays_d, 2 * rs->batch, nullptr, workspace_bytes);
  k.poses = rs->poses; k.image_rows = rs->image_rows; k.images = rs->images; k.workspace2 = workspace2; k.scratch = scratch;
  k.subset_offset = cfg->rng_offset; k.ws2bytes = workspace2_bytes; k.scbytes = scratch_bytes;
  k.batch = rs->batch; k.H = rs->H; k.W = rs->W; k.K = rs->K; k.num_images = rs->num_images; k.focal = rs->focal;
  return k;
}
// the paired render's cfg / jitter pair of an iteration whose caller passed `cfg`
VoxeRenderCfg recon_pair_cfg(const VoxeRenderCfg* cfg) {
  VoxeRenderCfg pc = *cfg;
  pc.ray_state_valid = 0;
  pc.rng_offset = cfg->rng_offset + 1;
  pc.render_diffuse = 0;
  pc.linear_grad = 1;
  pc.reuse_packed_grid = 0;      // (not part of any key; the step sets it itself)
  return pc;
}
struct ReconBatchPtrs { int64_t* subset; float *rays_o, *rays_d, *target; };
ReconBatchPtrs recon_batch_ptrs(const ReconLayout& l, void* scratch, int slot) {
  char* sc = (char*)scratch + (slot ? l.batch2 : 0);
  return ReconBatchPtrs{(int64_t*)(sc + l.subset), (float*)(sc + l.rays_o), (float*)(sc + l.rays_d), (float*)(sc + l.target)};
}
int recon_check(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg, const VoxeReconStep* rs) {
  if (!grid || !cfg || !rs || !rs->poses || !rs->images || !rs->losses) return VOXE_ERR_NULL_POINTER;
  if (grid->feature_kind != VOXE_FEAT_SH) return VOXE_ERR_UNSUPPORTED;
  if (rs->H <= 0 || rs->W <= 0 || rs->K <= 0 || rs->batch <= 0 || rs->batch > (int64_t)rs->K * rs->H * rs->W)
    return VOXE_ERR_BAD_SHAPE;
  if (rs->num_images <= 0 || (!rs->image_rows && rs->K > rs->num_images)) return VOXE_ERR_BAD_SHAPE;
  if (cfg->image_width != 0 || cfg->image_height != 0 || cfg->deterministic) return VOXE_ERR_UNSUPPORTED;   // a random batch has no image order
  if ((int64_t)rs->K * rs->H * rs->W > (1ll << 31)) return VOXE_ERR_BAD_SHAPE;
  return VOXE_OK;
}
}  // namespace

size_t voxe_recon_scratch_bytes(int64_t batch) { return batch > 0 ? recon_layout(batch).total : 0; }

int voxe_recon_prefetch(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg, const VoxeReconStep* rs, void* workspace,
                        size_t workspace_bytes, void* workspace2, size_t workspace2_bytes, void* scratch, size_t scratch_bytes,
                        void* stream) {
  const int chk = recon_check(grid, cfg, rs);
  if (chk) return chk;
  // only the paired render of SH-0 grids has tables worth assembling ahead; everything else: the hint is dropped
  if (!(rs->diffuse_regularisation && cfg->sh_degree == 0 && VOXE_RECON_PAIRED) || !workspace || !workspace2 || !scratch) return VOXE_OK;
  const int64_t B = rs->batch;
  const ReconLayout l = recon_layout(B);
  if (scratch_bytes < l.total) return VOXE_ERR_WORKSPACE;
  const VoxeRenderCfg pc = recon_pair_cfg(cfg);
  const PairSpec pair{B, cfg->rng_offset + 2};
  PairScope scope(&pair);
  const WsLayout wl = ws_layout(grid, &pc, 2 * B);
  if (!wl.region || workspace_bytes < wl.total_with_src || workspace2_bytes < wl.total_with_src) return VOXE_OK;
  Variant v;
  const int st = validate(grid, &pc, 2 * B, &v);
  if (st) return st;
  hipStream_t s = (hipStream_t)stream;
  std::lock_guard<std::mutex> lock(g_recon_mu);
  if (g_recon.size() >= kMaxReconSides && g_recon.find(workspace) == g_recon.end()) {
    // (workspaces come and go with their trainers: records without a hint in flight give their stream and events back)
    for (auto it = g_recon.begin(); it != g_recon.end();) {
      ReconSide& o = it->second;
      if (o.pend.valid) { ++it; continue; }          // (a hint in flight: its events are still waited for)
      if (o.side) { (void)hipStreamSynchronize(o.side); (void)hipStreamDestroy(o.side); }
      if (o.fork) (void)hipEventDestroy(o.fork);
      if (o.fork0) (void)hipEventDestroy(o.fork0);
      if (o.pend.done) (void)hipEventDestroy(o.pend.done);
      it = g_recon.erase(it);
    }
  }
  ReconSide& rsd = g_recon[workspace];
  if (!rsd.side) {
    // (lowest priority: the side stream's kernels fill what the iteration's own kernels leave idle, they do not queue in front of them)
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (hipStreamCreateWithPriority(&rsd.side, hipStreamNonBlocking, VOXE_RECON_SIDE_LOW ? prio_lo : prio_hi) != hipSuccess) return VOXE_ERR_LAUNCH;
    if (hipEventCreateWithFlags(&rsd.fork, hipEventDisableTiming) != hipSuccess) return VOXE_ERR_LAUNCH;
    if (hipEventCreateWithFlags(&rsd.fork0, hipEventDisableTiming) != hipSuccess) return VOXE_ERR_LAUNCH;
    if (hipEventCreateWithFlags(&rsd.pend.done, hipEventDisableTiming) != hipSuccess) return VOXE_ERR_LAUNCH;
  }
  // (a hint that was never consumed: its tables are overwritten below -- in stream order on the side stream)
  const int slot = 1 - rsd.last_slot;
  if (!rsd.forked) {     // no step has run on this workspace yet (or the last one took another route): behind whatever is on `stream`
    (void)hipEventRecord(rsd.fork0, s);
    (void)hipEventRecord(rsd.fork, s);
  }
  const bool early = VOXE_RECON_FORK == 2 || VOXE_RECON_FORK == 3 || VOXE_RECON_FORK == 0;
  (void)hipStreamWaitEvent(rsd.side, early ? rsd.fork0 : rsd.fork, 0);
  const ReconBatchPtrs bp = recon_batch_ptrs(l, scratch, slot);
  launch_recon_batch(B, cfg->seed, cfg->rng_offset, rs->H, rs->W, rs->focal, rs->K, rs->poses, rs->images,
                     (const long long*)rs->image_rows, rs->num_images, (long long*)bp.subset, bp.rays_o, bp.rays_d, bp.target, rsd.side);
  DevGrid dg; HostCfg dc;
  make_dev(grid, &pc, 2 * B, v, &dg, &dc);
  void* const tables = (char*)(slot ? workspace2 : workspace) + wl.region_off;
  launch_bin_region(dg, dc, bp.rays_o, bp.rays_d, nullptr, tables, rsd.side, /*phase=*/1);      // counters cleared
  if (VOXE_RECON_FORK == 2 || VOXE_RECON_FORK == 3) (void)hipStreamWaitEvent(rsd.side, rsd.fork, 0);
  launch_bin_region(dg, dc, bp.rays_o, bp.rays_d, nullptr, tables, rsd.side, /*phase=*/2);      // segments -> sorted tables
  (void)hipEventRecord(rsd.pend.done, rsd.side);
  rsd.pend.valid = true;
  rsd.pend.slot = slot;
  ++g_recon_stats[0];
  rsd.pend.key = recon_key(grid, &pc, cfg, rs, bp.rays_o, bp.rays_d, workspace_bytes, workspace2, workspace2_bytes, scratch, scratch_bytes);
  return finish();
}

int voxe_recon_prefetch_stats(int64_t* out) {
  if (!out) return VOXE_ERR_NULL_POINTER;
  std::lock_guard<std::mutex> lock(g_recon_mu);
  for (int i = 0; i < 3; ++i) out[i] = g_recon_stats[i];
  return VOXE_OK;
}

int voxe_recon_step(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg, const VoxeReconStep* rs, void* workspace,
                    size_t workspace_bytes, void* workspace2, size_t workspace2_bytes, void* scratch, size_t scratch_bytes,
                    void* stream) {
  const int chk = recon_check(grid, cfg, rs);
  if (chk) return chk;
  const int nrender = rs->diffuse_regularisation ? 2 : 1;
  if (nrender == 2 && !workspace2) return VOXE_ERR_WORKSPACE;
  const ReconLayout l = recon_layout(rs->batch);
  if (!scratch || scratch_bytes < l.total) return VOXE_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  char* sc = (char*)scratch;
  const int64_t B = rs->batch;
  int st = VOXE_OK;
  // a prefetch pending on this workspace: the step waits for the side stream whether or not it takes the tables (they may be
  // the set this call is about to write), and takes them iff they were made for exactly this call
  bool have_side = false, pending = false;
  int pend_slot = 0;
  ReconKey pend_key;
  {
    std::lock_guard<std::mutex> lock(g_recon_mu);
    const auto it = g_recon.find(workspace);
    if (it != g_recon.end()) {
      have_side = true;
      if (it->second.pend.valid) {
        pending = true; pend_slot = it->second.pend.slot; pend_key = it->second.pend.key;
        (void)hipStreamWaitEvent(s, it->second.pend.done, 0);
        it->second.pend.valid = false;
      }
      it->second.last_slot = 0;
      it->second.forked = false;    // (until the paired route below says from where on the idle tables may be rewritten: a hint given
                                    //  behind a step that took another route is ordered behind everything on the stream)
    }
  }
  // SH-0 grids: the diffuse render is the specular kernel with another jitter stream -- both renders as ONE launch of 2 B
  // rays on the space-binned route (one binning pass, one forward, one backward; no second packed grid).  32 768-ray
  // batches leave the chip half empty and the binning / scan / fold passes are launch-latency sized: r04, 160^3,
  // 1.15 -> see profiles/r04_recon_bench.txt.  Needs `workspace` sized for 2 B rays (voxe_workspace_bytes(grid, cfg, 2 B)).
  if (nrender == 2 && cfg->sh_degree == 0 && VOXE_RECON_PAIRED) {
    VoxeRenderCfg pc = recon_pair_cfg(cfg);
    pc.reuse_packed_grid = cfg->reuse_packed_grid;
    const PairSpec pair{B, cfg->rng_offset + 2};
    PairScope scope(&pair);
    const WsLayout wl = ws_layout(grid, &pc, 2 * B);
    if (wl.region && workspace_bytes >= wl.total_with_src) {
      int slot = 0;
      if (pending) {
        const ReconBatchPtrs pp = recon_batch_ptrs(l, scratch, pend_slot);
        const ReconKey k = recon_key(grid, &pc, cfg, rs, pp.rays_o, pp.rays_d, workspace_bytes, workspace2, workspace2_bytes, scratch, scratch_bytes);
        if (memcmp(&k, &pend_key, sizeof(k)) == 0) slot = pend_slot; else pending = false;
        std::lock_guard<std::mutex> lock(g_recon_mu);
        ++g_recon_stats[pending ? 1 : 2];
      }
      const bool prebinned = pending;
      const ReconBatchPtrs bp = recon_batch_ptrs(l, scratch, slot);
      if (have_side) {
        std::lock_guard<std::mutex> lock(g_recon_mu);
        const auto it = g_recon.find(workspace);      // (another thread's voxe_recon_prefetch may have recycled the record since)
        if (it == g_recon.end() || !it->second.side) have_side = false;
        else {
          (void)hipEventRecord(it->second.fork0, s);
          if (VOXE_RECON_FORK == 0) { (void)hipEventRecord(it->second.fork, s); it->second.forked = true; }
        }
      }
      // batch assembly: keyed subset of the K * H * W pixels -> rays + target pixels (one launch; same streams as
      // voxe_random_subset, voxe_cast_rays_indexed and the pixel gather)
      if (!prebinned)
        launch_recon_batch(B, cfg->seed, cfg->rng_offset, rs->H, rs->W, rs->focal, rs->K, rs->poses, rs->images,
                           (const long long*)rs->image_rows, rs->num_images, (long long*)bp.subset, bp.rays_o, bp.rays_d, bp.target, s);
      hipEvent_t mid = nullptr;
      if (have_side && VOXE_RECON_FORK == 3) {
        std::lock_guard<std::mutex> lock(g_recon_mu);
        const auto it = g_recon.find(workspace);
        if (it != g_recon.end()) mid = it->second.fork;
      }
      const RegionBins bins{slot ? (void*)((char*)workspace2 + wl.region_off) : nullptr, prebinned ? 1 : 0, mid};
      struct BinsScope { explicit BinsScope(const RegionBins* b) { tl_region_bins = b; } ~BinsScope() { tl_region_bins = nullptr; } } bscope(&bins);
      float* colour = (float*)(sc + l.out2);
      float *depth = colour + 6 * B, *acc = depth + 2 * B;
      float* d_colour = (float*)(sc + l.d_colour2);
      st = voxe_render_fwd(grid, &pc, bp.rays_o, bp.rays_d, 2 * B, nullptr, colour, depth, acc, nullptr, workspace, workspace_bytes, stream);
      if (st) return st;
      if (have_side) {
        std::lock_guard<std::mutex> lock(g_recon_mu);
        const auto it = g_recon.find(workspace);
        if (it != g_recon.end() && it->second.side) {
          ReconSide& rsd = it->second;
          rsd.last_slot = slot;
          if (VOXE_RECON_FORK == 3) rsd.forked = true;      // (recorded between the region forward and the fold pass: RegionBins::after_fwd)
          else if (VOXE_RECON_FORK != 0) { (void)hipEventRecord(rsd.fork, s); rsd.forked = true; }
        }
      }
      launch_l1_loss_grad_n(colour, bp.target, 3 * B, 2, d_colour, rs->losses, sc + l.partial, s);   // both renders, one launch pair
      pc.ray_state_valid = 1;
      pc.reuse_packed_grid = 1;
      int32_t layout = VOXE_GRAD_ANY;
      st = voxe_render_bwd_acc_into(grid, &pc, bp.rays_o, bp.rays_d, 2 * B, nullptr, colour, depth, acc, d_colour, nullptr, nullptr,
                                    rs->exp_avg_densities != nullptr, rs->exp_avg_features != nullptr,
                                    rs->zero_gradient_first ? 1 : 0, &layout, workspace, workspace_bytes, nullptr, 0, stream);
      if (st) return st;
      if (layout != VOXE_GRAD_LINEAR) return VOXE_ERR_UNSUPPORTED;
      return voxe_grid_adam_step(grid, VOXE_GRAD_LINEAR, 0, grid->X, nullptr, nullptr, rs->exp_avg_densities,
                                 rs->exp_avg_sq_densities, rs->exp_avg_features, rs->exp_avg_sq_features, rs->lr, rs->beta1,
                                 rs->beta2, rs->eps, rs->step_densities, rs->step_features, nullptr, workspace, workspace_bytes,
                                 stream);
    }
  }
  int64_t* subset = (int64_t*)(sc + l.subset);
  float* rays_o = (float*)(sc + l.rays_o);
  float* rays_d = (float*)(sc + l.rays_d);
  float* target = (float*)(sc + l.target);
  launch_recon_batch(B, cfg->seed, cfg->rng_offset, rs->H, rs->W, rs->focal, rs->K, rs->poses, rs->images,
                     (const long long*)rs->image_rows, rs->num_images, (long long*)subset, rays_o, rays_d, target, s);
  VoxeRenderCfg rc[2] = {*cfg, *cfg};
  void* ws[2] = {workspace, workspace2};
  size_t wsb[2] = {workspace_bytes, workspace2_bytes};
  for (int i = 0; i < nrender; ++i) {
    rc[i].rng_offset = cfg->rng_offset + 1 + (uint64_t)i;
    rc[i].ray_state_valid = 0;
    rc[i].render_diffuse = i == 1 ? 1 : cfg->render_diffuse;
    rc[i].linear_grad = 1;                         // both renders write ONE gradient layout, whatever kernel runs
    if (i == 1) rc[i].reuse_packed_grid = 0;       // the second workspace still holds the previous iteration's grid
    float* colour = (float*)(sc + l.out[i]);
    float *depth = colour + 3 * B, *acc = depth + B;
    st = voxe_render_fwd(grid, &rc[i], rays_o, rays_d, B, nullptr, colour, depth, acc, nullptr, ws[i], wsb[i], stream);
    if (st) return st;
    launch_l1_loss_grad(colour, target, 3 * B, (float*)(sc + l.d_colour[i]), rs->losses + 2 * i, sc + l.partial, s);
  }
  for (int i = 0; i < nrender; ++i) {
    float* colour = (float*)(sc + l.out[i]);
    float *depth = colour + 3 * B, *acc = depth + B;
    rc[i].ray_state_valid = 1;                     // the forward above left this render's states in ws[i]
    rc[i].reuse_packed_grid = 1;
    int32_t layout = VOXE_GRAD_ANY;
    st = voxe_render_bwd_acc_into(grid, &rc[i], rays_o, rays_d, B, nullptr, colour, depth, acc, (const float*)(sc + l.d_colour[i]),
                                  nullptr, nullptr, rs->exp_avg_densities != nullptr, rs->exp_avg_features != nullptr,
                                  /*zero_first=*/(i == 0 && rs->zero_gradient_first) ? 1 : 0, &layout, ws[i], wsb[i], i == 0 ? nullptr : workspace,
                                  i == 0 ? 0 : workspace_bytes, stream);
    if (st) return st;
    if (layout != VOXE_GRAD_LINEAR) return VOXE_ERR_UNSUPPORTED;
  }
  return voxe_grid_adam_step(grid, VOXE_GRAD_LINEAR, 0, grid->X, nullptr, nullptr, rs->exp_avg_densities,
                             rs->exp_avg_sq_densities, rs->exp_avg_features, rs->exp_avg_sq_features, rs->lr, rs->beta1,
                             rs->beta2, rs->eps, rs->step_densities, rs->step_features, nullptr, workspace, workspace_bytes, stream);
}

namespace {
struct RefineLayout { size_t out, d_render, tv_grad, tv_scratch, l1_scratch, total; };
RefineLayout refine_layout(const VoxeGridDesc* g, int64_t R) {
  RefineLayout l;
  size_t off = 0;
  const size_t r = (size_t)(R > 0 ? R : 0);
  l.out = off; off += up256b(r * 4 * sizeof(float));          // attn [R] | depth [R] | acc [R] | (spare)
  l.d_render = off; off += up256b(r * sizeof(float));
  l.tv_grad = off; off += up256b((size_t)g->X * g->Y * g->Z * sizeof(float));
  l.tv_scratch = off; off += up256b(tv_scratch_bytes(g->X, g->Y, g->Z, 1));
  l.l1_scratch = off; off += up256b(attn_l1_scratch_bytes());
  l.total = off;
  return l;
}
}  // namespace

size_t voxe_attn_masked_l1_scratch_bytes(void) { return attn_l1_scratch_bytes(); }

int voxe_attn_masked_l1(const float* render, const float* attn_map, int64_t n, float* d_render, float* loss_out, void* scratch,
                        size_t scratch_bytes, void* stream) {
  if (n < 0) return VOXE_ERR_BAD_SHAPE;
  if (n == 0) return VOXE_OK;
  if (!render || !attn_map || !d_render) return VOXE_ERR_NULL_POINTER;
  if (!scratch || scratch_bytes < attn_l1_scratch_bytes()) return VOXE_ERR_WORKSPACE;
  launch_attn_masked_l1(render, attn_map, n, d_render, loss_out, scratch, (hipStream_t)stream);
  return finish();
}

size_t voxe_attn_refine_scratch_bytes(const VoxeGridDesc* grid, int64_t R) {
  if (!grid || grid->X <= 0 || grid->Y <= 0 || grid->Z <= 0 || R <= 0) return 0;
  return refine_layout(grid, R).total;
}

int voxe_attn_refine_step(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg, const VoxeAttnRefineStep* rs, const float* rays_o,
                          const float* rays_d, int64_t R, void* workspace, size_t workspace_bytes, void* scratch,
                          size_t scratch_bytes, void* stream) {
  if (!grid || !cfg || !rs || !rays_o || !rays_d || !rs->attn_map || !rs->exp_avg || !rs->exp_avg_sq) return VOXE_ERR_NULL_POINTER;
  if (grid->feature_kind != VOXE_FEAT_ATTN) return VOXE_ERR_UNSUPPORTED;
  if (R <= 0) return VOXE_ERR_BAD_SHAPE;
  if (cfg->deterministic) return VOXE_ERR_UNSUPPORTED;
  const RefineLayout l = refine_layout(grid, R);
  if (!scratch || scratch_bytes < l.total) return VOXE_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  char* sc = (char*)scratch;
  float* attn = rs->attn_render ? rs->attn_render : (float*)(sc + l.out);
  float *depth = (float*)(sc + l.out) + R, *acc = depth + R;
  float* d_render = (float*)(sc + l.d_render);
  float* tv_grad = (float*)(sc + l.tv_grad);
  VoxeRenderCfg rc = *cfg;
  rc.ray_state_valid = 0;
  int st = voxe_render_fwd(grid, &rc, rays_o, rays_d, R, nullptr, attn, depth, acc, nullptr, workspace, workspace_bytes, stream);
  if (st) return st;
  launch_attn_masked_l1(attn, rs->attn_map, R, d_render, rs->losses, sc + l.l1_scratch, s);
  rc.ray_state_valid = 1;
  rc.reuse_packed_grid = 1;
  int32_t layout = VOXE_GRAD_ANY;
  st = voxe_render_bwd_acc_into(grid, &rc, rays_o, rays_d, R, nullptr, attn, depth, acc, d_render, nullptr, nullptr,
                                /*want_densities=*/0, /*want_features=*/1, rs->zero_gradient_first ? 1 : 0, &layout, workspace,
                                workspace_bytes, nullptr, 0, stream);
  if (st) return st;
  // the TV stencil reads the parameters BEFORE the step overwrites them: its gradient goes through the step's extra-gradient input
  const float* extra = nullptr;
  if (rs->tv_weight != 0.0f || (rs->losses && rs->tv_loss_always)) {
    launch_tv(grid->features, grid->X, grid->Y, grid->Z, 1, rs->tv_weight, rs->losses ? rs->losses + 1 : nullptr,
              rs->tv_weight != 0.0f ? tv_grad : nullptr, 0, sc + l.tv_scratch, s);
    if (rs->tv_weight != 0.0f) extra = tv_grad;
  }
  return voxe_grid_adam_step(grid, layout, 0, grid->X, nullptr, extra, nullptr, nullptr, rs->exp_avg, rs->exp_avg_sq, rs->lr,
                             rs->beta1, rs->beta2, rs->eps, rs->step, rs->step, nullptr, workspace, workspace_bytes, stream);
}

int voxe_clock_probe(int32_t spin, double* shader_hz, void* stream) {
  if (!shader_hz) return VOXE_ERR_NULL_POINTER;
  *shader_hz = run_clock_probe(spin, (hipStream_t)stream);
  return *shader_hz > 0.0 ? finish() : VOXE_ERR_LAUNCH;
}

int voxe_region_debug_layout(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg, int64_t R, int64_t out[17]) {
  if (!out) return VOXE_ERR_NULL_POINTER;
  Variant v;
  const int st = validate(grid, cfg, R, &v);
  if (st) return st;
  const WsLayout l = ws_layout(grid, cfg, R);
  if (!l.region) return VOXE_ERR_UNSUPPORTED;   // (grid, cfg, R) does not take the space-binned route
  long long t[16];
  region_debug_layout(grid->X, grid->Y, grid->Z, R, cfg->num_samples, t);
  out[0] = (int64_t)l.region_off;
  for (int i = 0; i < 16; ++i) out[1 + i] = t[i];
  return VOXE_OK;
}

size_t voxe_workspace_grad_offset(const VoxeGridDesc* grid) {
  if (!grid || grid->X <= 0 || grid->Y <= 0 || grid->Z <= 0 || grid->F <= 0) return 0;
  return ws_layout(grid, nullptr, 0).grad_off;
}
size_t voxe_workspace_grad_bytes(const VoxeGridDesc* grid) {
  if (!grid || grid->X <= 0 || grid->Y <= 0 || grid->Z <= 0 || grid->F <= 0) return 0;
  const WsLayout l = ws_layout(grid, nullptr, 0);
  return l.state_off - l.grad_off;
}

int voxe_grid_adam_step(const VoxeGridDesc* grid, int32_t grad_layout, int32_t x_begin, int32_t x_end,
                        const float* extra_d_densities,
                        const float* extra_d_features, float* exp_avg_d, float* exp_avg_sq_d, float* exp_avg_f,
                        float* exp_avg_sq_f, float lr, float beta1, float beta2, float eps, int64_t step,
                        int64_t step_features, const VoxeGridRegularisers* reg, void* workspace, size_t workspace_bytes,
                        void* stream) {
  if (!grid || !grid->densities || !grid->features) return VOXE_ERR_NULL_POINTER;
  if (grid->X <= 0 || grid->Y <= 0 || grid->Z <= 0 || grid->F <= 0 || step < 1 || step_features < 0) return VOXE_ERR_BAD_SHAPE;
  if ((long long)grid->X * grid->Y * grid->Z * (grid->F + 1) >= (1LL << 31)) return VOXE_ERR_BAD_SHAPE;
  if (x_begin < 0 || x_end > grid->X || x_begin > x_end) return VOXE_ERR_BAD_SHAPE;
  if (grad_layout == VOXE_GRAD_BRICKED && (x_begin & 1) && x_begin != x_end) return VOXE_ERR_BAD_SHAPE;
  if ((exp_avg_d == nullptr) != (exp_avg_sq_d == nullptr) || (exp_avg_f == nullptr) != (exp_avg_sq_f == nullptr))
    return VOXE_ERR_NULL_POINTER;
  if (grad_layout != VOXE_GRAD_LINEAR && grad_layout != VOXE_GRAD_BRICKED && grad_layout != VOXE_GRAD_ANY)
    return VOXE_ERR_UNSUPPORTED;
  const WsLayout l = ws_layout(grid, nullptr, 0);
  if (!workspace || workspace_bytes < l.state_off) return VOXE_ERR_WORKSPACE;
  DclTerm dcl;
  const long long nvox_all = (long long)grid->X * grid->Y * grid->Z;
  if (reg && reg->dcl_reference && reg->density_kind == VOXE_DREG_CORRELATION) {
    // moments over the WHOLE grid of the current parameters; the gradient is evaluated inside the step
    const long long n = nvox_all;
    if (x_begin != 0 || x_end != grid->X || !exp_avg_d) return VOXE_ERR_UNSUPPORTED;
    if (!reg->scratch || reg->scratch_bytes < dcl_scratch_bytes(n)) return VOXE_ERR_WORKSPACE;
    dcl.b = reg->dcl_reference;
    dcl.stats = launch_dcl_moments(grid->densities, reg->dcl_reference, n, reg->dcl_weight, reg->dcl_loss, reg->scratch,
                                   (hipStream_t)stream);
  } else if (reg && reg->dcl_reference) {
    // l2_mode / l1_mode (sds_trainer.py:494-503): per-voxel terms, no statistics; a reduction only for the logged value
    if (reg->density_kind != VOXE_DREG_L2 && reg->density_kind != VOXE_DREG_L1) return VOXE_ERR_UNSUPPORTED;
    if (!exp_avg_d) return VOXE_ERR_UNSUPPORTED;
    const long long plane = (long long)grid->Y * grid->Z;
    if (reg->dcl_loss) {
      if (!reg->scratch || reg->scratch_bytes < dcl_scratch_bytes(nvox_all)) return VOXE_ERR_WORKSPACE;
      if (x_begin == x_end && hipMemsetAsync(reg->dcl_loss, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) return VOXE_ERR_LAUNCH;
      if (x_begin < x_end)
        launch_density_diff(grid->densities + x_begin * plane, reg->dcl_reference + x_begin * plane, (x_end - x_begin) * plane,
                            reg->density_kind, 0.0f, reg->dcl_loss, (float)(1.0 / (double)nvox_all), nullptr, 0, reg->scratch,
                            (hipStream_t)stream);
    }
    dcl.b = reg->dcl_reference;
    dcl.kind = reg->density_kind;
    dcl.k = (float)((reg->density_kind == VOXE_DREG_L2 ? 2.0 : 1.0) * (double)reg->dcl_weight / (double)nvox_all);
  }
  if (reg && reg->feat_reference) {
    // _feature_correlation_loss (sds_trainer.py:526-534): per voxel inside the step (texels of up to 4 channels)
    if (grid->F > 3 || !exp_avg_f) return VOXE_ERR_UNSUPPORTED;
    const long long plane = (long long)grid->Y * grid->Z;
    if (reg->feat_loss) {
      if (!reg->scratch || reg->scratch_bytes < dcl_scratch_bytes(nvox_all)) return VOXE_ERR_WORKSPACE;
      if (x_begin == x_end && hipMemsetAsync(reg->feat_loss, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) return VOXE_ERR_LAUNCH;
      if (x_begin < x_end)
        launch_feature_correlation(grid->features + x_begin * plane * grid->F, reg->feat_reference + x_begin * plane * grid->F,
                                   (x_end - x_begin) * plane, grid->F, 0.0f, reg->feat_loss, nullptr, 0, reg->scratch,
                                   (hipStream_t)stream);
    }
    dcl.fref = reg->feat_reference;
    dcl.fk = 2.0f * reg->feat_weight;
  }
  if (x_begin == x_end) return VOXE_OK;
  // the parameters move: the per-ray states a forward left in this workspace no longer describe the grid -- a backward that still
  // claims ray_state_valid = 1 afterwards is served by a re-march
  forget_stamp(workspace);
  forget_stamps_of(grid->densities);   // (sibling workspaces / a grad_workspace's owner rendered the same tensors)
  forget_stamps_of(grid->features);
  if (!launch_grid_adam(grid, grad_layout == VOXE_GRAD_BRICKED, x_begin, x_end, (float*)((char*)workspace + l.grad_off),
                        extra_d_densities, extra_d_features, exp_avg_d, exp_avg_sq_d, exp_avg_f, exp_avg_sq_f, lr, beta1,
                        beta2, eps, step, step_features > 0 ? step_features : step,
                        (float*)((char*)workspace + l.packed_off), (hipStream_t)stream, dcl))
    return VOXE_ERR_UNSUPPORTED;
  return finish();
}

int voxe_sample_probe(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg, const float* rays_o,
                      const float* rays_d, int64_t R, const float* jitter, int32_t* idx,
                      uint8_t* inside, float* zvals, float* sigma, float* rad, void* workspace,
                      size_t workspace_bytes, void* stream) {
  Variant v;
  const int st = validate(grid, cfg, R, &v);
  if (st) return st;
  if (R > 0 && (!rays_o || !rays_d)) return VOXE_ERR_NULL_POINTER;
  const WsLayout l = ws_layout(grid, cfg, R);
  if (!workspace || workspace_bytes < l.fwd_total) return VOXE_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  float* packed = (float*)((char*)workspace + l.packed_off);
  if (!cfg->reuse_packed_grid) launch_pack_any(grid, packed, s);
  if (R == 0) return finish();
  DevGrid dg; HostCfg dc;
  make_dev(grid, cfg, R, v, &dg, &dc);
  ProbeArgs a{packed, rays_o, rays_d, jitter, idx, inside, zvals, sigma, rad};
  launch_probe(dg, dc, cfg->sh_degree, cfg->render_diffuse, a, s);
  return finish();
}

namespace {
int validate_grid_only(const VoxeGridDesc* g) {
  if (!g || !g->densities || !g->features) return VOXE_ERR_NULL_POINTER;
  if (g->X <= 0 || g->Y <= 0 || g->Z <= 0 || g->F <= 0) return VOXE_ERR_BAD_SHAPE;
  const int C = g->F + 1;
  if (C != 2 && C != 4 && C != 13 && C != 28 && C != 49) return VOXE_ERR_BAD_SHAPE;
  if ((long long)g->X * g->Y * g->Z * C >= (1LL << 31)) return VOXE_ERR_BAD_SHAPE;
  if ((long long)g->X * g->Y >= (1LL << 24) || (long long)g->Y * g->Z >= (1LL << 24) || g->Z >= (1 << 24)) return VOXE_ERR_BAD_SHAPE;
  if (g->density_pre_act != VOXE_ACT_IDENTITY && g->density_pre_act != VOXE_ACT_ABS) return VOXE_ERR_UNSUPPORTED;
  if (g->density_post_act != VOXE_ACT_IDENTITY && g->density_post_act != VOXE_ACT_RELU &&
      g->density_post_act != VOXE_ACT_SOFTPLUS)
    return VOXE_ERR_UNSUPPORTED;
  return VOXE_OK;
}
void grid_to_dev(const VoxeGridDesc* g, DevGrid* dg) {
  dg->X = g->X; dg->Y = g->Y; dg->Z = g->Z;
  for (int a = 0; a < 3; ++a) {
    dg->lo[a] = g->aabb_lo[a]; dg->hi[a] = g->aabb_hi[a];
    dg->scale[a] = g->norm_scale[a]; dg->bias[a] = g->norm_bias[a];
  }
  dg->density_scale = g->density_scale;
  dg->pre_act = g->density_pre_act; dg->post_act = g->density_post_act;
}
}  // namespace

int voxe_query_fwd(const VoxeGridDesc* grid, const float* points, int64_t N, float* out,
                   int32_t reuse_packed_grid, void* workspace, size_t workspace_bytes, void* stream) {
  const int st = validate_grid_only(grid);
  if (st) return st;
  if (N < 0) return VOXE_ERR_BAD_SHAPE;
  if (N > 0 && (!points || !out)) return VOXE_ERR_NULL_POINTER;
  const WsLayout l = ws_layout(grid, nullptr, 0);
  if (!workspace || workspace_bytes < l.fwd_total) return VOXE_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  float* packed = (float*)((char*)workspace + l.packed_off);
  if (!reuse_packed_grid) launch_pack_any(grid, packed, s);
  if (N == 0) return finish();
  DevGrid dg;
  grid_to_dev(grid, &dg);
  launch_query(dg, grid->F + 1, packed, points, N, out, nullptr, nullptr, false, false, s);
  return finish();
}

int voxe_query_bwd(const VoxeGridDesc* grid, const float* points, int64_t N, const float* d_out,
                   float* d_densities, float* d_features, int32_t accumulate, int32_t reuse_packed_grid,
                   void* workspace, size_t workspace_bytes, void* stream) {
  const int st = validate_grid_only(grid);
  if (st) return st;
  if (N < 0) return VOXE_ERR_BAD_SHAPE;
  if (N > 0 && (!points || !d_out)) return VOXE_ERR_NULL_POINTER;
  if (!d_densities && !d_features) return VOXE_OK;
  const WsLayout l = ws_layout(grid, nullptr, 0);
  if (!workspace || workspace_bytes < l.total) return VOXE_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  float* packed = (float*)((char*)workspace + l.packed_off);
  float* gpacked = (float*)((char*)workspace + l.grad_off);
  if (!reuse_packed_grid) launch_pack_any(grid, packed, s);
  if (hipMemsetAsync(gpacked, 0, l.state_off - l.grad_off, s) != hipSuccess) return VOXE_ERR_LAUNCH;
  if (N > 0) {
    DevGrid dg;
    grid_to_dev(grid, &dg);
    launch_query(dg, grid->F + 1, packed, points, N, nullptr, d_out, gpacked, d_densities != nullptr,
                 d_features != nullptr, s);
  }
  launch_unpack_any(grid, gpacked, d_densities, d_features, accumulate, 0, s);
  return finish();
}

size_t voxe_dcl_scratch_bytes(int64_t n) { return dcl_scratch_bytes(n); }

int voxe_dcl_fwd_bwd(const float* a, const float* b, int64_t n, float grad_scale, float* loss_out,
                     float* d_a, int32_t accumulate, void* scratch, size_t scratch_bytes,
                     void* stream) {
  if (!a || !b || !loss_out) return VOXE_ERR_NULL_POINTER;
  if (n <= 0) return VOXE_ERR_BAD_SHAPE;
  if (!scratch || scratch_bytes < dcl_scratch_bytes(n)) return VOXE_ERR_WORKSPACE;
  launch_dcl(a, b, n, grad_scale, loss_out, d_a, accumulate, scratch, (hipStream_t)stream);
  return finish();
}

int voxe_density_diff_fwd_bwd(const float* a, const float* b, int64_t n, int32_t kind, float grad_scale, float* loss_out,
                              float* d_a, int32_t accumulate, void* scratch, size_t scratch_bytes, void* stream) {
  if (!a || !b) return VOXE_ERR_NULL_POINTER;
  if (n <= 0) return VOXE_ERR_BAD_SHAPE;
  if (kind != VOXE_DREG_L2 && kind != VOXE_DREG_L1) return VOXE_ERR_UNSUPPORTED;
  if (loss_out && (!scratch || scratch_bytes < dcl_scratch_bytes(n))) return VOXE_ERR_WORKSPACE;
  if (!loss_out && !d_a) return VOXE_OK;
  launch_density_diff(a, b, n, kind, grad_scale, loss_out, (float)(1.0 / (double)n), d_a, accumulate, scratch, (hipStream_t)stream);
  return finish();
}

int voxe_feature_correlation_fwd_bwd(const float* f, const float* r, int64_t nvox, int32_t F, float grad_scale, float* loss_out,
                                     float* d_f, int32_t accumulate, void* scratch, size_t scratch_bytes, void* stream) {
  if (!f || !r) return VOXE_ERR_NULL_POINTER;
  if (nvox <= 0 || F < 1 || F > 64) return VOXE_ERR_BAD_SHAPE;
  if (loss_out && (!scratch || scratch_bytes < dcl_scratch_bytes(nvox))) return VOXE_ERR_WORKSPACE;
  if (!loss_out && !d_f) return VOXE_OK;
  launch_feature_correlation(f, r, nvox, F, grad_scale, loss_out, d_f, accumulate, scratch, (hipStream_t)stream);
  return finish();
}

size_t voxe_tv_scratch_bytes(int32_t X, int32_t Y, int32_t Z, int32_t C) {
  return tv_scratch_bytes(X, Y, Z, C);
}

int voxe_tv_fwd_bwd(const float* grid, int32_t X, int32_t Y, int32_t Z, int32_t C, float grad_scale,
                    float* loss_out, float* d_grid, int32_t accumulate, void* scratch,
                    size_t scratch_bytes, void* stream) {
  if (!grid || !loss_out) return VOXE_ERR_NULL_POINTER;
  if (X <= 0 || Y <= 0 || Z <= 0 || C <= 0) return VOXE_ERR_BAD_SHAPE;
  if ((long long)Y * Z * C >= (1ll << 31) - 256) return VOXE_ERR_BAD_SHAPE;   // (32-bit index arithmetic inside an x-plane)
  if (!scratch || scratch_bytes < tv_scratch_bytes(X, Y, Z, C)) return VOXE_ERR_WORKSPACE;
  launch_tv(grid, X, Y, Z, C, grad_scale, loss_out, d_grid, accumulate, scratch, (hipStream_t)stream);
  return finish();
}

int voxe_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   float lr, float beta1, float beta2, float eps, int64_t step, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq) return VOXE_ERR_NULL_POINTER;
  if (n < 0 || step < 1) return VOXE_ERR_BAD_SHAPE;
  forget_stamps_of(param);   // (a later ray_state_valid = 1 over this tensor re-marches)
  launch_adam(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, (hipStream_t)stream);
  return finish();
}

int voxe_upsample_trilinear(const float* src, int32_t X, int32_t Y, int32_t Z, int32_t C, float* dst,
                            int32_t X2, int32_t Y2, int32_t Z2, void* stream) {
  if (!src || !dst) return VOXE_ERR_NULL_POINTER;
  if (X <= 0 || Y <= 0 || Z <= 0 || C <= 0 || X2 <= 0 || Y2 <= 0 || Z2 <= 0) return VOXE_ERR_BAD_SHAPE;
  if (X2 > 65535 || (long long)Y2 * Z2 * C >= (1ll << 31) - 256) return VOXE_ERR_BAD_SHAPE;   // (grid.y / 32-bit plane index)
  launch_upsample(src, X, Y, Z, C, dst, X2, Y2, Z2, (hipStream_t)stream);
  return finish();
}

// ---- refinement stage ------------------------------------------------------------------------------
static bool refine_dims_ok(int32_t X, int32_t Y, int32_t Z) {
  return X > 0 && Y > 0 && Z > 0 && (long long)X * Y * Z < (1ll << 30);
}

int voxe_graph_build(const float* density_grid, const float* feature_grid, int32_t X, int32_t Y, int32_t Z,
                     int32_t F, float sigma, int32_t dilate_yz, uint8_t* node_mask, int32_t* cap, void* stream) {
  if (!density_grid || !feature_grid || !node_mask || !cap) return VOXE_ERR_NULL_POINTER;
  if (!refine_dims_ok(X, Y, Z) || F <= 0 || !(sigma > 0.0f)) return VOXE_ERR_BAD_SHAPE;
  launch_graph_build(density_grid, feature_grid, X, Y, Z, F, sigma, dilate_yz ? 1 : 0, node_mask, cap,
                     (hipStream_t)stream);
  return finish();
}

size_t voxe_graphcut_scratch_bytes(int32_t X, int32_t Y, int32_t Z) {
  return refine_dims_ok(X, Y, Z) ? graphcut_scratch_bytes(X, Y, Z) : 0;
}

int voxe_graphcut(const uint8_t* node_mask, const int8_t* terminal, int32_t* cap, int32_t X, int32_t Y, int32_t Z,
                  uint8_t* segment, int64_t* flow, void* scratch, size_t scratch_bytes, void* stream) {
  if (!node_mask || !terminal || !cap || !segment || !flow) return VOXE_ERR_NULL_POINTER;
  if (!refine_dims_ok(X, Y, Z)) return VOXE_ERR_BAD_SHAPE;
  if (!scratch || scratch_bytes < graphcut_scratch_bytes(X, Y, Z)) return VOXE_ERR_WORKSPACE;
  if (run_graphcut(node_mask, terminal, cap, X, Y, Z, segment, flow, scratch, (hipStream_t)stream) != hipSuccess)
    return VOXE_ERR_LAUNCH;
  return finish();
}

size_t voxe_cc_scratch_bytes(int32_t X, int32_t Y, int32_t Z, int32_t k) {
  return refine_dims_ok(X, Y, Z) && k >= 0 ? cc_scratch_bytes(X, Y, Z, k) : 0;
}

int voxe_cc_largest_k(const uint8_t* mask, int32_t X, int32_t Y, int32_t Z, int32_t k, int32_t* labels,
                      int32_t* num_components, void* scratch, size_t scratch_bytes, void* stream) {
  if (!mask || !labels || !num_components) return VOXE_ERR_NULL_POINTER;
  if (!refine_dims_ok(X, Y, Z) || k < 0 || k > 4096) return VOXE_ERR_BAD_SHAPE;
  if (!scratch || scratch_bytes < cc_scratch_bytes(X, Y, Z, k)) return VOXE_ERR_WORKSPACE;
  launch_cc_largest_k(mask, X, Y, Z, k, labels, num_components, scratch, (hipStream_t)stream);
  return finish();
}

}  // extern "C"
