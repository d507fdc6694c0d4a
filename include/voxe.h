/*
 * voxe.h -- C ABI of the MI355X-native voxel-grid volumetric renderer (libvoxe_hip.so)
 *           and of its CPU twin, the test oracle (oracle/libvoxe_oracle.so, voxe_cpu_*).
 *
 * The reference (TAU-VAILab/Vox-E, thre3d_atom) has NO native ABI: its hot path is a chain of
 * ATen ops reached through autograd.  This header is the boundary a maintainer would bind from
 * Python (ctypes stub in INTEGRATION.md); each entry point names the reference code it replaces
 * (paths relative to the reference repository root).
 *
 * Conventions
 *   - plain C, no torch / HIP types: `stream` is a hipStream_t passed as void* (NULL = default
 *     stream); every pointer of a voxe_* function is a DEVICE pointer on the current HIP device,
 *     every pointer of a voxe_cpu_* function is a HOST pointer;
 *   - all buffers are caller-owned, contiguous, float32 unless stated; nothing is allocated inside;
 *   - every function returns 0 on success or a negative VoxeStatus; nothing throws;
 *   - functions are asynchronous w.r.t. the host (they enqueue on `stream`) and thread-safe for
 *     distinct (stream, workspace) pairs.
 *
 * Grid memory layout (reference: thre3d_atom/thre3d_reprs/voxels.py:46-136)
 *   densities [X,Y,Z,1], features [X,Y,Z,F]  (C innermost, Z fastest spatial axis);
 *   element (ix,iy,iz,c) lives at ((ix*Y + iy)*Z + iz)*C + c.
 */
#ifndef VOXE_H_
#define VOXE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VOXE_ABI_VERSION 12

typedef enum VoxeStatus {
  VOXE_OK = 0,
  VOXE_ERR_NULL_POINTER = -1,
  VOXE_ERR_BAD_SHAPE = -2,       /* non-positive grid dims / R / S, F not 3*(deg+1)^2 ...        */
  VOXE_ERR_UNSUPPORTED = -3,     /* activation / mode the HIP path does not implement            */
  VOXE_ERR_WORKSPACE = -4,       /* workspace NULL or smaller than voxe_workspace_bytes()        */
  VOXE_ERR_LAUNCH = -5,          /* hipGetLastError() after a launch was not hipSuccess          */
  VOXE_ERR_NO_DEVICE = -6        /* no HIP device / wrong architecture                           */
} VoxeStatus;

/* density activations (voxels.py:55-57; CLI table train_sh_based_voxel_grid_with_posed_images.py:177-200) */
typedef enum VoxeAct {
  VOXE_ACT_IDENTITY = 0,
  VOXE_ACT_ABS = 1,        /* torch.abs            (pre-activation of the non-softplus field)     */
  VOXE_ACT_RELU = 2,       /* torch.nn.ReLU        (post)                                         */
  VOXE_ACT_SOFTPLUS = 3    /* torch.nn.Softplus(beta=1, threshold=20) (post; the CLI default)     */
} VoxeAct;

/* What the "feature" channels are. */
typedef enum VoxeFeatureKind {
  VOXE_FEAT_SH = 0,    /* RGB spherical-harmonic coefficients, F = 3*(deg+1)^2; Cout = 3
                          (process.py:20-96, accumulate.py:31-113)                                */
  VOXE_FEAT_ATTN = 1   /* one attention channel, F = 1; Cout = 1; background multiplied by 0
                          (voxels.py:344-406, process.py:98-174, accumulate.py:115-198)           */
} VoxeFeatureKind;

typedef struct VoxeGridDesc {
  const float* densities;   /* [X,Y,Z,1]  VoxelGrid._densities (or .orig_densities)               */
  const float* features;    /* [X,Y,Z,F]  VoxelGrid._features  (or .attn for VOXE_FEAT_ATTN)      */
  int32_t X, Y, Z, F;
  float aabb_lo[3];         /* float32(aabb.{x,y,z}_range[0])  voxels.py:196-223                  */
  float aabb_hi[3];         /* float32(aabb.{x,y,z}_range[1])                                     */
  float norm_scale[3];      /* adjust_dynamic_range(slack=True): np.float32(2)/(f32(hi)-f32(lo))  */
  float norm_bias[3];       /*   np.float32(-1) - f32(lo)*scale   (imaging_utils.py:57-63)        */
  float density_scale;      /* expected_density_scale (voxels.py:303-305)                         */
  int32_t density_pre_act;  /* VoxeAct: IDENTITY | ABS                                            */
  int32_t density_post_act; /* VoxeAct: IDENTITY | RELU | SOFTPLUS                                */
  int32_t feature_kind;     /* VoxeFeatureKind                                                    */
} VoxeGridDesc;

/* Which kernels render a call, and with which tuning parameters (ABI v7).  Every field: 0 = the shipped default, so a
 * zero-initialised struct -- or VoxeRenderCfg::dispatch == NULL -- is the shipped dispatch.  The struct is read on every call
 * from the caller's memory: there is no process-global dispatch state in the library and nothing reads the environment on
 * the render path (the Python binding resolves its VOXE_* environment switches ONCE into one of these, voxe_hip/dispatch.py).
 * Forward and backward of one render must be given the same values (the per-ray states a forward leaves in the workspace
 * belong to one route: voxe_render_route).  Results never depend on these fields beyond float summation order. */
typedef struct VoxeDispatch {
  int32_t bwd_mode;            /* 0 auto | 1 plain global-atomic scatter (the A/B baseline) | 2 line-dense scatter            */
  int32_t tile_map;            /* block -> pixel-tile map: 0 auto | 1 interleaved over the XCDs | 2 XCD bands | 3 tile rows    */
  int64_t tile_min_rays;       /* image-ordered renders below this many rays take the line-dense scatter backward instead of
                                  the LDS-window one: 0 = 8192 | > 0 explicit | -1 no minimum                                 */
  int32_t tile_two_phase;      /* view-dependent grids: 0 two-phase backward when the workspace holds the per-sample sources |
                                  -1 always the single-kernel channel groups                                                  */
  int32_t tile_qsplit;         /* parts of a tile that does not fit the window: 0 auto (by launch size) | 1 consecutive passes
                                  of one block | 4 sibling blocks                                                             */
  int32_t tile_kl;             /* lateral edge of the backward's LDS window (voxels): 0 auto | 8 | 10                          */
  float tile_fit_m;            /* a pass fits the window when its spread along the march axis is below this many layers:
                                  0 = by launch size (4.0 ... 5.5)                                                            */
  float tile_fit_lat;          /* ... and its lateral extent below this many voxels: 0 = window edge - 2.5                     */
  int32_t fwd_window;          /* LDS texel window of the image-ordered forward (SH-0: 16-byte texels; r06: SH 1 - 3, whole texels):
                                  0 on | -1 off (ray-ordered forward) | 2 test aid (SH 1 - 3): samples the window did not serve
                                  render as NaN                                                                               */
  float fwd_fit_lat, fwd_fit_m;/* window forward: fit bounds of a tile (0 = 5.5 voxels / 4.5 layers; SH 1 - 3: 5.5 / 12 -- its
                                  lanes wait for a layer instead of leaving the window)                                       */
  float fwd_zdom;              /* a tile is z-dominant when |d_z| >= |this| x max(|d_x|, |d_y|); > 0: such tiles march ray by ray,
                                  < 0: along z through the window.  0 = 1.0 for SH-0, -1.0 for SH 1 - 3 (4.4)    */
  float fwd_max_adv;           /* layers a window tile may advance per sample: 0 = 1.7                                         */
  int32_t fwd_segments_per_thread; /* depth segments one thread of the ray-ordered forward walks: 0 = 1                        */
  int64_t region_min_rays;     /* space-binned route for unordered / sparse rays from this many rays on: 0 = 16384 | > 0
                                  explicit | -1 route off                                                                     */
  float region_image_ratio;    /* image-ordered launches take the space-binned route when grid side >= this x image width:
                                  0 = 1.3 | < 0: every image-ordered launch the route accepts                                 */
  int32_t tile_lean;           /* ABI v8.  SH-0 image-ordered backward: 0 = the lean LDS-window kernel (voxe_render_tile4.hip)
                                  wherever it applies | -1 always the general kernel (A/B runs, parity tests)                 */
  int32_t precise_grad;        /* ABI v8.  0 = float running sums inside a depth segment (density gradients ~1e-5 median
                                  relative error against a double-precision backward) | 1 = the suffix sums of the
                                  image-ordered SH-0 backward are carried in double (~2e-6, a few percent slower).  Forward
                                  and backward of one render must agree on it (the forward saves the segment states).       */
  int32_t region_lds_ranks;    /* ABI v9.  Space-binned route, segment pass: 0 = segments are ranked per block in an LDS table
                                  (no global atomic; grids up to ~200^3) | -1 one returning global atomic per segment (r02)   */
  int32_t tile_phases;         /* ABI v11.  SH-0 image-ordered backward, tiles that do not fit the window and run as 2 / 4 parts:
                                  0 = the lanes outside a part take the other SAMPLE PHASES of the part's rays (32 rays x 2
                                  consecutive samples, 16 rays x 4: every wave instruction works on 64 lanes) | -1 = one
                                  sample per ray and iteration, the other lanes idle (r05).  Read only by a library built with
                                  -DVOXE_T4_PHASES_KL8=1 / -DVOXE_T4_PHASES_KL10=1: the shipped kernels carry no phased march
                                  (inside one kernel it cost the one-sample march 8 % at 400x400 and won nothing over the
                                  views at 100 .. 266 px: csrc/voxe_render_tile4.hip, profiles/r06_phases_kl8.txt).          */
} VoxeDispatch;

typedef struct VoxeRenderCfg {
  int32_t num_samples;        /* S   SHVoxGridRenderConfig.num_samples_per_ray (renderers.py:32)  */
  float near, far;            /* CameraBounds (sample.py:38-41)                                   */
  int32_t perturb;            /* stratified jitter (sample.py:55-64). jitter==NULL -> in-kernel
                                 counter hash keyed by (seed, rng_offset, ray, sample)             */
  int32_t linear_disparity;   /* sample.py:48-51                                                  */
  int32_t aabb_clip;          /* optimized_sampling: per-ray bounds from the ray/AABB slab test
                                 (sample.py:71-202)                                               */
  int32_t white_bkgd;         /* accumulate.py:77-81 (for VOXE_FEAT_ATTN the term is *0.0, :166)  */
  int32_t sh_degree;          /* 0..3 ; F == 3*(sh_degree+1)^2 for VOXE_FEAT_SH                   */
  int32_t render_diffuse;     /* use only the degree-0 coefficient (process.py:59-63)             */
  float term_eps;             /* gradient truncation (NOT in the reference; 0 = off).  The forward always integrates all S
                                 samples exactly like the reference; with term_eps > 0 the BACKWARD stops marching a ray once
                                 its transmittance is below term_eps: samples behind that point get no gradient (their
                                 contributions scale with T < term_eps), the samples in front keep their EXACT gradient
                                 (the suffix sums come from the full forward).  -20 % backward time on surface-like scenes. */
  uint64_t seed, rng_offset;  /* in-kernel jitter stream                                          */
  int32_t reuse_packed_grid;  /* 1: workspace already holds this grid packed by a previous call
                                 on the same workspace (grid values unchanged)                    */
  int32_t image_width;        /* 0 = unknown. >0: rays are a row-major H x W image (ray r is
                                 pixel (r / W, r % W)); lets the kernels use 2-D pixel tiles      */
  int32_t image_height;       /* 0 = one image (H = R / image_width).  >0 with image_width > 0: the rays are K =
                                 R / (image_height * image_width) row-major images of the same size, one after
                                 the other (ray r = (camera * H + y) * W + x): ONE launch renders a multi-view
                                 batch; pixel tiles never straddle two cameras.  R must be a multiple of H * W.  */
  int32_t deterministic;      /* backward only, test / race-check mode (SURVEY 8(b) `deterministic`).  1: the gradient
                                 is accumulated in 64-bit FIXED POINT (integer adds are associative, so the result
                                 does not depend on the order lanes, waves and blocks meet on a voxel): two calls on
                                 the same inputs return identical bits.  Costs an extra measuring pass (per-call
                                 power-of-two scales from max |contribution|) and 8 bytes per gradient value of
                                 workspace (voxe_workspace_bytes accounts for it).  Image-ordered rays
                                 (image_width > 0), SH degree 0 / render_diffuse / attention grids; other
                                 configurations return VOXE_ERR_UNSUPPORTED.  0: float atomics (default).          */
  int32_t linear_grad;        /* voxe_render_bwd_acc(_into) only.  1: the gradient region is written in
                                 VOXE_GRAD_LINEAR whatever backward kernel runs (the line-dense scatter of small
                                 unordered batches otherwise prefers VOXE_GRAD_BRICKED, -7 % on that kernel): lets a
                                 caller accumulate ANY mix of renders into one optimiser step.                     */
  int32_t ray_state_valid;    /* backward only. 1: `workspace` still holds the per-ray depth-segment
                                 states (transmittance + partial sums every VOXE_SEGMENT_SAMPLES
                                 samples) written by voxe_render_fwd for EXACTLY these rays / cfg /
                                 jitter, so the depth-segmented backward starts from them.
                                 0: the backward first re-marches the rays to rebuild them.
                                 voxe_render_fwd reads it too: -1 = no backward of these rays will follow
                                 (inference): the forward skips what only that backward would read (the per-sample
                                 values of view-dependent image-ordered renders, r04), and
                                 voxe_workspace_bytes() leaves out the backward's scratch.  The 1 is a CLAIM the
                                 library checks (r05): every forward records, per workspace address, what it
                                 rendered there (grid, rays, jitter, cfg, dispatch, whether the per-sample values
                                 were kept); a backward whose claim does not match the record -- after a forward
                                 with -1, a forward of other rays in between, or a voxe_grid_adam_step on this
                                 workspace (the parameters moved) -- re-marches as if given 0.  voxe_grid_adam_step
                                 and voxe_adam_step also drop the record of EVERY workspace whose forward read the
                                 tensors they rewrite (r06).  What the record compares is identity -- pointers,
                                 shapes, AABB, cfg, dispatch -- never CONTENTS: a caller that rewrites the grid, the
                                 rays or the jitter in place by other means (its own kernels, a copy into the same
                                 buffer) must pass 0.                                                            */
  const VoxeDispatch* dispatch; /* HOST pointer, NULL = the shipped dispatch; read during the call only (ABI v7)      */
} VoxeRenderCfg;

/* depth-segment length of the segmented kernels (samples per segment); launches of at most 20000 rays use 16: they
 * leave the chip under-filled, and shorter dependent chains then matter more than the extra boundary states     */
#ifndef VOXE_SEGMENT_SAMPLES
#define VOXE_SEGMENT_SAMPLES 32
#endif

/* ------------------------------------------------------------------------------------------------
 * Library / device
 * ---------------------------------------------------------------------------------------------- */
int voxe_abi_version(void);
const char* voxe_strerror(int status);
/* 0 when a gfx950 device is current; VOXE_ERR_NO_DEVICE otherwise. Fills name (may be NULL). */
int voxe_device_check(char* name, size_t name_len);

/* ------------------------------------------------------------------------------------------------
 * Per-phase device timing (measurement hook used by bench.py; no reference counterpart).
 * While enabled, every render call brackets its phases with hipEvents ON THE CALLER'S STREAM
 * (pack, forward kernel, gradient memset, backward kernel, unpack).  voxe_profile_read() waits for
 * the recorded events and returns the summed milliseconds / launch counts since voxe_profile_enable(1).
 * At most 512 phase records are kept between enable and read; later ones are dropped (n_dropped).
 * ---------------------------------------------------------------------------------------------- */
typedef struct VoxeProfile {
  double ms_pack, ms_fwd, ms_memset, ms_bwd, ms_unpack;
  int32_t n_pack, n_fwd, n_memset, n_bwd, n_unpack, n_dropped;
} VoxeProfile;
int voxe_profile_enable(int32_t on);
int voxe_profile_read(VoxeProfile* out);

/* ------------------------------------------------------------------------------------------------
 * Ray casting -- rendering/volumetric/utils/misc.py:12-50 (cast_rays)
 *   rot[9] row-major 3x3, trans[3] : HOST pointers (12 floats, copied by value into the launch)
 *   rays_o, rays_d : [H*W,3]
 * ---------------------------------------------------------------------------------------------- */
int voxe_cast_rays(int32_t H, int32_t W, float focal, const float* rot, const float* trans,
                   float* rays_o, float* rays_d, void* stream);

/* Rays of SELECTED pixels of K cameras sharing (H, W, focal): what the reconstruction trainer keeps of
 *   cast_rays x K -> collate_rays -> randperm subset   (modules/trainers.py:290-313, misc.py:60-70,126-138)
 * without materialising the K full images of rays.  poses: DEVICE [K,3,4] (rotation | translation);
 * flat_index: DEVICE int64 [B], value = (camera * H + y) * W + x;  rays_o, rays_d: [B,3].  Arithmetic per pixel
 * identical to voxe_cast_rays (bit-exact same rays).                                                       */
int voxe_cast_rays_indexed(int32_t H, int32_t W, float focal, const float* poses, int32_t K,
                           const int64_t* flat_index, int64_t B, float* rays_o, float* rays_d, void* stream);

/* A uniformly random SUBSET of `count` distinct indices of [0, n), in random order: what `torch.randperm(n)[:count]`
 * is used for in the ray-batch samplers (rendering/volumetric/utils/misc.py:126-138) without permuting all n
 * (n = 1.28 M pixels for a 32768-ray batch).  out[i] = P(i), P a keyed 4-round Feistel permutation of [0, 2^b)
 * (b = even number of bits >= log2 n) restricted to [0, n) by cycle walking: distinct by construction, reproducible
 * from (seed, rng_offset); integer work, bit-exact between device and oracle.  n <= 2^31, count <= n.       */
int voxe_random_subset(int64_t n, int64_t count, uint64_t seed, uint64_t rng_offset, int64_t* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Volumetric render, forward -- replaces the whole chain
 *   render_sh_voxel_grid(_attn)            thre3d_reprs/renderers.py:50-163
 *   -> sample_uniform_points_on_rays       rendering/volumetric/sample.py:15-68 (+ :71-202)
 *   -> process_points_with_sh_voxel_grid   rendering/volumetric/process.py:20-174
 *      -> VoxelGrid.forward(_attn)         thre3d_reprs/voxels.py:287-406
 *      -> evaluate_spherical_harmonics     rendering/volumetric/utils/spherical_harmonics.py:64-132
 *   -> accumulate_radiance_density_on_rays rendering/volumetric/accumulate.py:31-198
 *
 *   rays_o, rays_d [R,3]; jitter [R,S] uniforms in [0,1) or NULL;
 *   outputs colour [R,Cout] (Cout = 3 SH / 1 attn), depth [R], acc [R], disparity [R]
 *   (any of depth/acc/disparity may be NULL).
 * ---------------------------------------------------------------------------------------------- */
/* Workspace: [packed grid | packed gradient | per-ray depth-segment states | segment partials | per-sample gradient
 * sources].  voxe_workspace_bytes() is the size that lets every kernel take its fast route; the forward needs only the
 * packed grid, and the backward needs everything but the last region (16 B per sample of image-ordered renders of SH
 * degree >= 1, capped at 4 GB): without it their gradient-channel groups re-march the rays instead of sharing one march.
 * What the optional regions cost, so that a caller can size its pools:
 *   - view-dependent grids, image-ordered renders: besides the 2 x 16 B per sample above, the GROUP-PLANAR staging gradient of
 *     the lean deposit passes, voxels x 16 B x ceil((F + 1) / 4) (262 MB for SH-1 at 160^3, ~850 MB for SH-3), cleared and
 *     folded into the packed gradient on every such backward whatever the image size;
 *   - unordered / sparse rays (space-binned route): segment tables + per-segment states, and for grids whose region table fits
 *     LDS (up to ~200^3) the per-block rank tables of the segment pass, blocks x (regions + 1) x 24 B (blocks <= 1024: ~48 MB
 *     at 160^3 for a 32 768-ray batch, ~120 MB at 160 000 rays), allocated whether or not VoxeDispatch::region_lds_ranks
 *     takes that pass;
 *   - VoxeDispatch::precise_grad: 5 doubles per (ray, depth segment); cfg->deterministic: 8 B per gradient value.
 * cfg->ray_state_valid = -1 (inference) returns the size without any of these. */
size_t voxe_workspace_bytes(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg, int64_t R);

int voxe_render_fwd(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg,
                    const float* rays_o, const float* rays_d, int64_t R, const float* jitter,
                    float* colour, float* depth, float* acc, float* disparity,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Volumetric render, backward -- replaces autograd through the chain above
 * (loss.backward(): modules/trainers.py:350, modules/sds_trainer.py:332,
 *  modules/attn_grid_trainer.py:372,376).
 *
 *   colour/depth/acc: the forward outputs for the same inputs (same jitter / seed);
 *   d_colour [R,Cout]; d_depth [R] or NULL; d_acc [R] or NULL   (upstream gradients)
 *   d_densities [X,Y,Z,1] or NULL (skip), d_features [X,Y,Z,F] or NULL (skip);
 *   accumulate != 0: += into the outputs, else overwrite.
 * ---------------------------------------------------------------------------------------------- */
int voxe_render_bwd(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg,
                    const float* rays_o, const float* rays_d, int64_t R, const float* jitter,
                    const float* colour, const float* depth, const float* acc,
                    const float* d_colour, const float* d_depth, const float* d_acc,
                    float* d_densities, float* d_features, int32_t accumulate,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Point query -- VoxelGrid.forward / forward_attn  thre3d_reprs/voxels.py:287-406
 *   points [N,3] world coordinates  ->  out [N,F+1] = (trilinear features f_0..f_{F-1}, post(trilinear pre(d*scale)))
 *   NOT masked by the AABB (the reference masks later, process.py:80-84): zero padding outside the grid.
 *   Backward: d_out [N,F+1] -> d_densities / d_features (either may be NULL), accumulate as above.
 *   workspace: >= voxe_workspace_bytes(grid, NULL, 0); cfg-less, so packed-grid reuse is an explicit flag.
 * ---------------------------------------------------------------------------------------------- */
int voxe_query_fwd(const VoxeGridDesc* grid, const float* points, int64_t N, float* out,
                   int32_t reuse_packed_grid, void* workspace, size_t workspace_bytes, void* stream);
int voxe_query_bwd(const VoxeGridDesc* grid, const float* points, int64_t N, const float* d_out,
                   float* d_densities, float* d_features, int32_t accumulate, int32_t reuse_packed_grid,
                   void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Per-sample probe (test hook for the bit-exact index-math contract): for ray r, sample k writes
 *   idx  [R,S,3] int32  floor() voxel index of the low trilinear corner (may be -1 or N-1.. out of range)
 *   inside [R,S] uint8  strict AABB test (voxels.py:263-285)
 *   zvals [R,S] float   sample depths (sample.py:46-64)
 *   sigma [R,S] float   post-activated, masked density; rad [R,S,Cout] masked raw radiance (process.py:80-84;
 *                       -1e10 outside)
 * any output may be NULL.
 * ---------------------------------------------------------------------------------------------- */
int voxe_sample_probe(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg,
                      const float* rays_o, const float* rays_d, int64_t R, const float* jitter,
                      int32_t* idx, uint8_t* inside, float* zvals, float* sigma, float* rad,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Whole-grid passes of the SDS / reconstruction step (HBM-streaming)
 * ---------------------------------------------------------------------------------------------- */
/* _density_correlation_loss  modules/sds_trainer.py:507-524
 *   loss = 1 - mean((a-mean a)(b-mean b)) / (sqrt(var a * var b) + 1e-7)
 *   a = sds densities (differentiated), b = regular densities (constant); n elements
 *   loss_out: 1 float; d_a: n floats (d loss / d a * grad_scale), accumulate as above; d_a may be NULL.
 *   scratch: >= voxe_dcl_scratch_bytes(n) bytes.                                                  */
size_t voxe_dcl_scratch_bytes(int64_t n);
int voxe_dcl_fwd_bwd(const float* a, const float* b, int64_t n, float grad_scale,
                     float* loss_out, float* d_a, int32_t accumulate,
                     void* scratch, size_t scratch_bytes, void* stream);

/* density_correlation_loss_fn's l2_mode / l1_mode  modules/sds_trainer.py:494-503 (+ autograd)
 *   VOXE_DREG_L2: loss = mean((a - b)^2)  (torch mse_loss),  d_a = grad_scale * 2 (a - b) / n
 *   VOXE_DREG_L1: loss = mean(|a - b|)    (torch l1_loss),   d_a = grad_scale * sign(a - b) / n   (sign(0) = 0)
 *   a = sds densities (differentiated), b = regular densities (constant); loss_out: 1 float or NULL; d_a: n floats or NULL
 *   (accumulate as above); scratch: >= voxe_dcl_scratch_bytes(n) bytes (only read when loss_out != NULL).
 *   VOXE_DREG_CORRELATION names the default (_density_correlation_loss, voxe_dcl_fwd_bwd) in VoxeGridRegularisers.      */
enum { VOXE_DREG_CORRELATION = 0, VOXE_DREG_L2 = 1, VOXE_DREG_L1 = 2 };
int voxe_density_diff_fwd_bwd(const float* a, const float* b, int64_t n, int32_t kind, float grad_scale,
                              float* loss_out, float* d_a, int32_t accumulate,
                              void* scratch, size_t scratch_bytes, void* stream);

/* _feature_correlation_loss  modules/sds_trainer.py:526-534 (+ autograd): f = sds features [nvox, F] (differentiated),
 *   r = regular features (constant):  D_v = sum_c (sigmoid(f_vc) - sigmoid(r_vc));  loss = sum_v D_v^2;
 *   d_f[v, c] = grad_scale * 2 D_v sigmoid(f_vc) (1 - sigmoid(f_vc)).  loss_out: 1 float or NULL; d_f: nvox * F floats or NULL
 *   (accumulate as above); scratch: >= voxe_dcl_scratch_bytes(nvox) bytes.  1 <= F <= 64.                                */
int voxe_feature_correlation_fwd_bwd(const float* f, const float* r, int64_t nvox, int32_t F, float grad_scale,
                                     float* loss_out, float* d_f, int32_t accumulate,
                                     void* scratch, size_t scratch_bytes, void* stream);

/* _tv_loss_on_grid  modules/sds_trainer.py:563-567 : grid [X,Y,Z,C]
 *   loss = (mean|diff_x| + mean|diff_y| + mean|diff_z|)/3 ; d_grid accumulates grad_scale * dloss/dgrid */
size_t voxe_tv_scratch_bytes(int32_t X, int32_t Y, int32_t Z, int32_t C);
int voxe_tv_fwd_bwd(const float* grid, int32_t X, int32_t Y, int32_t Z, int32_t C, float grad_scale,
                    float* loss_out, float* d_grid, int32_t accumulate,
                    void* scratch, size_t scratch_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused optimiser step of a voxel grid: gradient un-pack + torch.optim.Adam + re-pack in ONE streaming pass
 *   (modules/sds_trainer.py:200-203,332-333; modules/trainers.py:247-255,350-351).
 *
 * voxe_render_bwd_acc  == voxe_render_bwd that LEAVES the gradient in the workspace (kernel layout) instead of
 *   splitting it into the two API tensors.  zero_first != 0 clears the gradient region first, 0 accumulates on top
 *   (several renders per optimiser step).  *grad_layout (HOST int) receives VOXE_GRAD_LINEAR / VOXE_GRAD_BRICKED
 *   (which backward kernel ran; VOXE_GRAD_ANY for R == 0); every render accumulated into one step must report the
 *   same layout.
 * voxe_grid_adam_step  grid->densities / grid->features are the PARAMETERS and are updated in place (the const
 *   of the descriptor is cast away); gradient = the workspace's packed gradient (chain rule of the density
 *   pre-activation applied like voxe_render_bwd does) + optional extra gradients in API layout (regularisers);
 *   exp_avg / exp_avg_sq: Adam state per tensor, NULL pair = that tensor is frozen.  Afterwards the workspace holds the
 *   NEW grid packed (pass reuse_packed_grid = 1 to the next render) and a ZEROED gradient region (pass zero_first = 0).
 *   [x_begin, x_end) restricts the step to the voxels of those x-planes (0, X = the whole grid; x_begin must be even
 *   for VOXE_GRAD_BRICKED unless it is 0 ... X): an optimiser sharded over GPUs updates only its slab -- a slab is one
 *   contiguous byte range of every tensor involved, the workspace's packed grid and gradient regions included
 *   (voxe_workspace_grad_offset + x_begin * Y * Z * (F + 1) * 4 in the linear layout).
 *   Arithmetic identical to voxe_render_bwd + voxe_adam_step, bit for bit.                                        */
enum { VOXE_GRAD_ANY = -1, VOXE_GRAD_LINEAR = 0, VOXE_GRAD_BRICKED = 1 };
int voxe_render_bwd_acc(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg,
                        const float* rays_o, const float* rays_d, int64_t R, const float* jitter,
                        const float* colour, const float* depth, const float* acc,
                        const float* d_colour, const float* d_depth, const float* d_acc,
                        int32_t want_densities, int32_t want_features, int32_t zero_first, int32_t* grad_layout,
                        void* workspace, size_t workspace_bytes, void* stream);
/* The same with the gradient region of ANOTHER workspace of the same grid as the destination (grad_workspace; NULL =
 * `workspace` itself): two renders of one optimiser step that run in different workspaces -- each keeps its own
 * packed grid and per-ray states, e.g. the specular and the diffuse render of a reconstruction iteration
 * (modules/trainers.py:316,333) -- then sum into ONE gradient region that voxe_grid_adam_step consumes.            */
int voxe_render_bwd_acc_into(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg,
                             const float* rays_o, const float* rays_d, int64_t R, const float* jitter,
                             const float* colour, const float* depth, const float* acc,
                             const float* d_colour, const float* d_depth, const float* d_acc,
                             int32_t want_densities, int32_t want_features, int32_t zero_first, int32_t* grad_layout,
                             void* workspace, size_t workspace_bytes, void* grad_workspace, size_t grad_workspace_bytes,
                             void* stream);
/* layout the backward of (grid, cfg, R) would write: VOXE_GRAD_LINEAR / VOXE_GRAD_BRICKED (VOXE_GRAD_ANY for R == 0), or
 * a negative VOXE_ERR_* below -1 for invalid arguments -- lets a caller that accumulates several renders into one step
 * check that they agree BEFORE running one                                                                          */
int voxe_render_bwd_layout(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg, int64_t R);
size_t voxe_workspace_grad_offset(const VoxeGridDesc* grid);  /* byte offset / size of the gradient region, e.g. */
size_t voxe_workspace_grad_bytes(const VoxeGridDesc* grid);   /* for the multi-GPU all-reduce between the two calls */
/* Whole-grid regularisers evaluated INSIDE the grid step (modules/sds_trainer.py:305-334): the SDS edit's
 * default regulariser is the density-correlation loss between the edited and the original densities (weight 200,
 * edit_pretrained_relu_field.py:171).  Its gradient is elementwise in the parameters the step streams anyway, so only the
 * moment reduction stays a launch of its own (two small kernels); the separate gradient kernel, the [X,Y,Z,1] gradient
 * buffer and its round trip through HBM are gone.  dcl_reference == NULL: no term.  Requires the whole grid
 * (x_begin = 0, x_end = X: the moments are over all voxels).  Same arithmetic as voxe_dcl_fwd_bwd with
 * grad_scale = dcl_weight (the weight enters the two gradient constants in double instead of multiplying the finished float
 * gradient: results agree to float rounding, not bit for bit).  The total-variation terms (default weight 0) stay
 * separate passes: their stencil reads neighbours the in-place update of this very pass is overwriting. */
typedef struct VoxeGridRegularisers {
  const float* dcl_reference;   /* [X,Y,Z,1] densities of the pretrained field (constant), device; NULL = no DCL term      */
  float dcl_weight;             /* density_correlation_weight (x any schedule / 1 / world factor)                         */
  float* dcl_loss;              /* device float[1] or NULL: receives 1 - corr (unweighted, what the trainer logs)         */
  void* scratch;                /* >= voxe_dcl_scratch_bytes(X * Y * Z) bytes of device memory                            */
  size_t scratch_bytes;
  /* ABI v11: every regulariser the edit's CLI can switch on is a term of the fused step (no torch op, no gradient tensor)   */
  int32_t density_kind;         /* VOXE_DREG_*: what (dcl_reference, dcl_weight, dcl_loss) mean.  L2 / L1 are per-voxel
                                   terms with NO reduction (a reduction runs only when dcl_loss != NULL) and, unlike the
                                   correlation, also work on a slab (x_begin, x_end) -- the loss value then is the slab's
                                   share: sum over the slab / (X Y Z)                                                      */
  const float* feat_reference;  /* [X,Y,Z,F] features of the pretrained field (constant), device; NULL = no term:
                                   _feature_correlation_loss (sds_trainer.py:526-534) evaluated per voxel inside the step
                                   on the parameters the step starts from.  SH-0 / attention grids (F <= 3);
                                   VOXE_ERR_UNSUPPORTED for wider texels (use voxe_feature_correlation_fwd_bwd +
                                   extra_d_features there)                                                                */
  float feat_weight;            /* feature_correlation_weight (x any 1 / world factor)                                    */
  float* feat_loss;             /* device float[1] or NULL: the unweighted loss value (a reduction of its own)            */
} VoxeGridRegularisers;
int voxe_grid_adam_step(const VoxeGridDesc* grid, int32_t grad_layout, int32_t x_begin, int32_t x_end,
                        const float* extra_d_densities, const float* extra_d_features,
                        float* exp_avg_densities, float* exp_avg_sq_densities,
                        float* exp_avg_features, float* exp_avg_sq_features,
                        float lr, float beta1, float beta2, float eps, int64_t step, int64_t step_features,
                        const VoxeGridRegularisers* regularisers /* NULL = none */,
                        void* workspace, size_t workspace_bytes, void* stream);
/*   `step` / `step_features`: the 1-based Adam step counts of the densities and of the features (torch.optim.Adam keeps
 *   one counter per parameter; step_features = 0 means "the same as step").                                          */

/* Which kernels render (grid, cfg, R).  The per-ray states a forward leaves in the workspace belong to ONE route, so a
 * caller that sets cfg->ray_state_valid must know that forward and backward resolve to the same one (the choice also
 * depends on process-level tuning switches); negative VOXE_ERR_* below -1 for invalid arguments.                      */
enum { VOXE_ROUTE_NONE = -1, VOXE_ROUTE_SCATTER = 0, VOXE_ROUTE_TILE = 1, VOXE_ROUTE_PACKED_SCATTER = 2,
       VOXE_ROUTE_REGION = 3, VOXE_ROUTE_DETERMINISTIC = 4 };
int voxe_render_route(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg, int64_t R);

/* disparity = 1 / max(1e-10, depth / acc)   rendering/volumetric/accumulate.py:85-88
 *   chain rule of an upstream d_disparity [R] into d_depth / d_acc (what autograd does through those three tensor ops):
 *   d_depth_out = d_depth_in + dq / acc,  d_acc_out = d_acc_in - dq * depth / acc^2,  dq = -d_disparity / q^2 where
 *   q = depth / acc > 1e-10, 0 elsewhere (and wherever a term is NaN / infinite: rays that miss the volume).
 *   d_*_in may be NULL (= 0); the outputs may alias the inputs.                                                       */
int voxe_disparity_bwd(const float* depth, const float* acc, const float* d_disparity, const float* d_depth_in,
                       const float* d_acc_in, float* d_depth_out, float* d_acc_out, int64_t R, void* stream);

/* ------------------------------------------------------------------------------------------------
 * One iteration of the reconstruction loop in ONE call   modules/trainers.py:288-351
 *   (cast the rays of a random pixel batch over K cached cameras -> render specular [-> render diffuse] -> L1 loss(es)
 *   against the target pixels -> backward -> Adam): the same kernels the separate entry points launch, enqueued back to
 *   back on `stream` with no host work in between -- at 32768 rays the host side of ~45 framework-level launches was
 *   one third of the iteration.
 *   batch   : voxe_random_subset(K*H*W, batch, cfg->seed, cfg->rng_offset) picks flat (camera, y, x) pixels (a uniformly
 *             random set of distinct pixels, like the reference's randperm subset); rays by voxe_cast_rays_indexed; target
 *             pixels gathered from images [N,3,H,W] (row image_rows[camera], or camera itself when image_rows == NULL);
 *   renders : cfg as given with the jitter stream (seed, rng_offset + 1) for the specular render and, when
 *             diffuse_regularisation != 0, a second render with render_diffuse = 1 and the stream (seed, rng_offset + 2)
 *             (trainers.py:316,333); each runs in its own workspace (per-ray states), both gradients sum into the FIRST
 *             workspace's gradient region (voxe_render_bwd_acc_into);
 *   loss    : mean |colour - target| per render (torch.nn.functional.l1_loss), summed; losses[0..3] (DEVICE floats) receive
 *             L1 specular, MSE specular (the trainer logs it as PSNR), L1 diffuse, MSE diffuse;
 *   update  : voxe_grid_adam_step over the whole grid with the given Adam state / hyper-parameters; afterwards
 *             `workspace` holds the updated grid packed (pass cfg->reuse_packed_grid = 1 next time) and a cleared gradient.
 *   scratch : voxe_recon_scratch_bytes(batch) bytes of device memory (rays, targets, outputs, upstream gradients).
 *   SH-0 grids with diffuse_regularisation: when `workspace` holds voxe_workspace_bytes(grid, cfg, 2 * batch) bytes, both
 *   renders run as ONE launch of 2 * batch rays (same rays, jitter streams offset + 1 / + 2 as below; one binning pass,
 *   one forward, one backward) and `workspace2` is not touched; a smaller `workspace` takes the two-render path.
 *   Requires an SH grid (3 colour channels).  Arithmetic identical to the composition of the separate calls.        */
typedef struct {
  int32_t H, W;
  float focal;
  const float* poses;           /* [K,3,4] camera-to-world (rotation | translation), device                    */
  const int64_t* image_rows;    /* [K] rows of `images` of the K cameras, device; NULL = 0 .. K-1              */
  const float* images;          /* [N,3,H,W] device                                                            */
  int32_t num_images;           /* N: every entry of image_rows (or K itself when image_rows == NULL) must be below it;
                                   a row outside [0, N) reads nothing and makes that pixel's target NaN (ABI v7)        */
  int32_t K;
  int64_t batch;                /* rays per iteration                                                          */
  int32_t diffuse_regularisation;
  float lr, beta1, beta2, eps;
  int64_t step_densities, step_features;   /* 1-based Adam steps (torch counts per parameter)                  */
  float *exp_avg_densities, *exp_avg_sq_densities, *exp_avg_features, *exp_avg_sq_features;
  float* losses;                /* [4] device                                                                  */
  int32_t zero_gradient_first;  /* != 0: clear the gradient region of `workspace` first (a workspace no fused step has
                                   left cleared yet)                                                           */
} VoxeReconStep;
size_t voxe_recon_scratch_bytes(int64_t batch);
int voxe_recon_step(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg, const VoxeReconStep* step,
                    void* workspace, size_t workspace_bytes, void* workspace2, size_t workspace2_bytes,
                    void* scratch, size_t scratch_bytes, void* stream);
/* ABI v12.  The NEXT iteration's batch, assembled ahead of its call -- a hint, never a requirement.
 *   Batch assembly (subset, rays, target pixels) and the space-binning passes of the paired SH-0 render read the cameras,
 *   the images and the jitter streams, never the grid: modules/trainers.py:288-312 of iteration i + 1 does not depend on
 *   optimizer.step() of iteration i.  Call this right after voxe_recon_step(i) with the arguments voxe_recon_step(i + 1) WILL
 *   be called with (same pointers and sizes; only cfg->rng_offset, step->poses / image_rows and the Adam step counters
 *   normally differ -- the Adam fields, `losses` and cfg->reuse_packed_grid / ray_state_valid are not looked at): the library
 *   enqueues that work on a stream of its own, ordered behind iteration i's forward, where it overlaps iteration i's
 *   backward and grid step (what that is worth: ~1 % of a device-paced loop, and all the time a slow host would otherwise
 *   leave the device idle -- DESIGN 4.5).  The following voxe_recon_step waits for it and skips its own
 *   batch assembly / binning iff every argument that decides them equals the announced one; otherwise it waits and proceeds
 *   exactly as if no hint had been given.  Results do not depend on whether, or with what, this was called.
 *   The tables of the hint live in the region scratch of `workspace2` (which must hold voxe_workspace_bytes(grid, cfg,
 *   2 * batch) bytes like `workspace`) and alternate with those of `workspace`; `scratch` holds both batches
 *   (voxe_recon_scratch_bytes).  ORDER: the side stream is ordered behind the forward of the voxe_recon_step that was
 *   enqueued last on this workspace -- not behind this call -- so next_step->poses / image_rows / images must be complete on
 *   `stream` BEFORE that step was enqueued (draw the next iteration's cameras first, then call the step, then this), and
 *   stay valid and unchanged until the step that consumes the hint has been enqueued.  Freeing or reusing `workspace2` / `scratch` while a hint is in flight needs a device
 *   synchronisation first (the side stream is the library's own).  Returns VOXE_OK without doing anything when the iteration
 *   would not take the paired route (view-dependent grid, no diffuse regularisation, workspaces too small).            */
int voxe_recon_prefetch(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg, const VoxeReconStep* next_step,
                        void* workspace, size_t workspace_bytes, void* workspace2, size_t workspace2_bytes,
                        void* scratch, size_t scratch_bytes, void* stream);
/* (test aid) out[3]: hints issued | taken by the step that followed | dropped by it (arguments differed), process-wide */
int voxe_recon_prefetch_stats(int64_t* out);

/* ------------------------------------------------------------------------------------------------
 * One attention grid's share of an iteration of the refinement loop in ONE call (ABI v10)
 *   modules/attn_grid_trainer.py:335-378 -- per grid (edit / object): render_rays_attn -> calc_loss_on_attn_grid
 *   (modules/refinement_functions.py:42-77) + attn_tv_weight * _tv_loss_on_grid (:659-663) -> backward -> Adam step of the
 *   attention tensor.  The cross-attention map of the UNet (`attn_map`, the boundary to the networks) is an INPUT: the
 *   reference computes it before the attention renders, so everything behind it is grid work and runs here back to back on
 *   `stream`: voxe_render_fwd -> masked L1 + its gradient (one block) -> voxe_render_bwd_acc (attention channel only: the
 *   densities are frozen) -> voxe_tv_fwd_bwd -> voxe_grid_adam_step.  Arithmetic identical to the composition of those calls.
 *   grid     : VOXE_FEAT_ATTN; grid->features is the attention tensor [X,Y,Z,1], updated in place;
 *   rays     : [R,3] device, image order when cfg->image_width > 0 (R = H * W); jitter stream (cfg->seed, cfg->rng_offset);
 *   losses   : [2] device floats (or NULL): masked L1, TV (unweighted);
 *   scratch  : voxe_attn_refine_scratch_bytes(grid, R) bytes of device memory (outputs, upstream gradient, TV gradient).
 *   Afterwards `workspace` holds the updated grid packed and a cleared gradient region, like voxe_grid_adam_step.        */
/* calc_loss_on_attn_grid  modules/refinement_functions.py:42-77 (+ autograd), on its own:
 *   mask = render > 0;  *loss_out = sum(|render - map| * mask) / sum(mask);  d_render = ((1 / sum(mask)) * mask) * sign(render - map)
 *   render, map, d_render: [n] device floats; loss_out: device float or NULL; scratch: voxe_attn_masked_l1_scratch_bytes() bytes. */
size_t voxe_attn_masked_l1_scratch_bytes(void);
int voxe_attn_masked_l1(const float* render, const float* attn_map, int64_t n, float* d_render, float* loss_out,
                        void* scratch, size_t scratch_bytes, void* stream);

typedef struct {
  const float* attn_map;        /* [R] device, row-major like the rays                                          */
  float tv_weight;              /* attn_tv_weight; 0: no TV gradient (and no TV pass unless tv_loss_always)      */
  int32_t tv_loss_always;       /* != 0: losses[1] is evaluated even when tv_weight == 0                         */
  float lr, beta1, beta2, eps;
  int64_t step;                 /* 1-based Adam step of the attention tensor                                    */
  float *exp_avg, *exp_avg_sq;  /* [X,Y,Z,1] Adam state of the attention tensor                                 */
  float* losses;                /* [2] device or NULL                                                           */
  float* attn_render;           /* [R] device or NULL: receives the rendered attention image (logging)          */
  int32_t zero_gradient_first;  /* != 0: clear the gradient region of `workspace` first                         */
} VoxeAttnRefineStep;
size_t voxe_attn_refine_scratch_bytes(const VoxeGridDesc* grid, int64_t R);
int voxe_attn_refine_step(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg, const VoxeAttnRefineStep* step,
                          const float* rays_o, const float* rays_d, int64_t R, void* workspace, size_t workspace_bytes,
                          void* scratch, size_t scratch_bytes, void* stream);

/* Measurement aids (bench.py, tests): not part of the reference's interface.
 * voxe_clock_probe        sustained shader clock in Hz: a chip-filling VALU + LDS kernel on `stream` reads the shader-clock
 *                         and the constant reference-clock counters around its loop (blocking; `spin` iterations per
 *                         thread, <= 0: default, ~0.3 ms of device time);
 * voxe_region_debug_layout byte offsets (from the start of the workspace) and dimensions of the segment tables of the
 *                         space-binned route: out[0] = offset of the region scratch, out[1..16] = {slot_region, slot_pos,
 *                         slot_seg, sorted, lane_n, count, start} offsets inside it, nslots, nlanes, nreg, slots per
 *                         lane, region edge x / y / z (cells), length classes, longest segment; VOXE_ERR_UNSUPPORTED when
 *                         (grid, cfg, R) does not take that route.                                                    */
int voxe_clock_probe(int32_t spin, double* shader_hz, void* stream);
int voxe_region_debug_layout(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg, int64_t R, int64_t out[17]);

/* torch.optim.Adam(betas, eps, weight_decay=0, amsgrad=False) single-tensor step
 *   modules/sds_trainer.py:200-203, modules/trainers.py:247-255; `step` is the 1-based step count. */
int voxe_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   float lr, float beta1, float beta2, float eps, int64_t step, void* stream);

/* scale_voxel_grid_with_required_output_size  thre3d_reprs/voxels.py:409-447
 *   F.interpolate(mode="trilinear", align_corners=False, recompute_scale_factor=False)
 *   src [X,Y,Z,C] -> dst [X2,Y2,Z2,C]                                                            */
int voxe_upsample_trilinear(const float* src, int32_t X, int32_t Y, int32_t Z, int32_t C,
                            float* dst, int32_t X2, int32_t Y2, int32_t Z2, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Refinement stage: 3-D grid graph cut and connected components.
 * Integer / byte work on whole grids; results are bit-exact between libvoxe_hip.so and the oracle.
 * ---------------------------------------------------------------------------------------------- */

/* capacity quantum: an n-link of affinity exp(-l2/sigma) == 1 gets VOXE_GRAPH_CAP_ONE units per contribution */
#define VOXE_GRAPH_CAP_ONE (1 << 28)
/* direction order of the six n-link capacity planes */
enum { VOXE_DIR_XP = 0, VOXE_DIR_XM = 1, VOXE_DIR_YP = 2, VOXE_DIR_YM = 3, VOXE_DIR_ZP = 4, VOXE_DIR_ZM = 5 };

/* Graph construction of build_graph   modules/refinement_functions.py:182-287
 *   density_grid [X,Y,Z], feature_grid [X,Y,Z,F] (= sigmoid(_features), :378; pooled grids in the
 *   down-sampled branch :189-196).
 *   nodes      : dilate_yz = 1 -> `MaxPool3d(3, 1, 1)(densities[X,Y,Z,1]) > 0` (:186,:200): a 4-D input makes X the
 *                channel axis, so the dilation runs over the Y-Z plane only;  0 -> `density_grid > 0` (:194);
 *   n-links    : node i adds an edge (w, w) to each of its 6 neighbours n with density_grid[n] > 0 (:261-287),
 *                w = K * exp(-l2(feature_i - feature_n) / sigma); a pair of nodes therefore carries
 *                w * ([density_i > 0] + [density_n > 0]) in both directions.  The reference's bounds test (:264-266)
 *                compares every coordinate of n with x, y AND z, i.e. with min(X,Y,Z): on non-cubic grids a voxel
 *                with a coordinate >= min(X,Y,Z) is never visited as a neighbour (reproduced);  K scales every capacity alike
 *                and, with infinite seed t-links only (:252-257), cannot change the cut: capacities are stored as
 *                integers  llrint(exp(-l2/sigma) * VOXE_GRAPH_CAP_ONE) * multiplicity.
 *   node_mask u8 [X,Y,Z];  cap int32 [6,X,Y,Z] (VOXE_DIR_* planes; 0 where there is no edge).            */
int voxe_graph_build(const float* density_grid, const float* feature_grid,
                     int32_t X, int32_t Y, int32_t Z, int32_t F, float sigma, int32_t dilate_yz,
                     uint8_t* node_mask, int32_t* cap, void* stream);

/* g.maxflow() + g.get_segment()   modules/refinement_functions.py:289-294   (PyMaxflow 1.x, Boykov-Kolmogorov)
 *   terminal int8 [X,Y,Z]: +1 = add_tedge(inf, 0) ("edit" seed, source), -1 = add_tedge(0, inf) ("object" seed,
 *   sink), 0 = no t-link.  cap is consumed (holds the residual capacities on return); every capacity must be
 *   <= 2^29 so that a residual (forward + reverse capacity of a pair) stays below 2^31.
 *   segment u8 [X,Y,Z]: 255 = not a node, 1 = the node can still reach a sink seed in the residual graph of a
 *   maximum flow (BK's sink tree, get_segment == 1), 0 = otherwise (get_segment == 0, "edit").  That set is the
 *   same for every maximum flow, so any exact solver yields the same labels.
 *   flow int64[1]: value of the maximum flow in capacity units.
 *   BLOCKING: synchronises `stream` (the iteration count is data dependent).                             */
size_t voxe_graphcut_scratch_bytes(int32_t X, int32_t Y, int32_t Z);
int voxe_graphcut(const uint8_t* node_mask, const int8_t* terminal, int32_t* cap,
                  int32_t X, int32_t Y, int32_t Z, uint8_t* segment, int64_t* flow,
                  void* scratch, size_t scratch_bytes, void* stream);

/* cc3d.largest_k(mask, k, connectivity=26, delta=0)   edit_pretrained_relu_field.py:384-389,411-416
 *   mask u8 [X,Y,Z] (non-zero = foreground).  labels int32 [X,Y,Z]: 0 = background or a component that is not
 *   among the k largest; the M = min(k, N) largest components are numbered 1..M in ascending size (the largest
 *   is M; equal sizes: the component whose first voxel in memory order comes first counts as larger).
 *   num_components int32[1] = N (all 26-connected components).                                            */
size_t voxe_cc_scratch_bytes(int32_t X, int32_t Y, int32_t Z, int32_t k);
int voxe_cc_largest_k(const uint8_t* mask, int32_t X, int32_t Y, int32_t Z, int32_t k,
                      int32_t* labels, int32_t* num_components,
                      void* scratch, size_t scratch_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * CPU twin == the oracle (oracle/voxe_cpu.c). Same semantics, HOST pointers, no stream/workspace.
 * TEST INFRASTRUCTURE ONLY: never linked into libvoxe_hip.so, never called by the product path.
 * ---------------------------------------------------------------------------------------------- */
int voxe_cpu_cast_rays(int32_t H, int32_t W, float focal, const float* rot, const float* trans,
                       float* rays_o, float* rays_d);
int voxe_cpu_cast_rays_indexed(int32_t H, int32_t W, float focal, const float* poses, int32_t K,
                               const int64_t* flat_index, int64_t B, float* rays_o, float* rays_d);
int voxe_cpu_random_subset(int64_t n, int64_t count, uint64_t seed, uint64_t rng_offset, int64_t* out);
int voxe_cpu_render_fwd(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg,
                        const float* rays_o, const float* rays_d, int64_t R, const float* jitter,
                        float* colour, float* depth, float* acc, float* disparity);
int voxe_cpu_render_bwd(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg,
                        const float* rays_o, const float* rays_d, int64_t R, const float* jitter,
                        const float* d_colour, const float* d_depth, const float* d_acc,
                        float* d_densities, float* d_features, int32_t accumulate);
int voxe_cpu_query_fwd(const VoxeGridDesc* grid, const float* points, int64_t N, float* out);
int voxe_cpu_query_bwd(const VoxeGridDesc* grid, const float* points, int64_t N, const float* d_out,
                       float* d_densities, float* d_features, int32_t accumulate);
int voxe_cpu_sample_probe(const VoxeGridDesc* grid, const VoxeRenderCfg* cfg,
                          const float* rays_o, const float* rays_d, int64_t R, const float* jitter,
                          int32_t* idx, uint8_t* inside, float* zvals, float* sigma, float* rad);
int voxe_cpu_dcl_fwd_bwd(const float* a, const float* b, int64_t n, float grad_scale,
                         float* loss_out, float* d_a, int32_t accumulate);
int voxe_cpu_density_diff_fwd_bwd(const float* a, const float* b, int64_t n, int32_t kind, float grad_scale,
                                  float* loss_out, float* d_a, int32_t accumulate);
int voxe_cpu_feature_correlation_fwd_bwd(const float* f, const float* r, int64_t nvox, int32_t F, float grad_scale,
                                         float* loss_out, float* d_f, int32_t accumulate);
int voxe_cpu_tv_fwd_bwd(const float* grid, int32_t X, int32_t Y, int32_t Z, int32_t C,
                        float grad_scale, float* loss_out, float* d_grid, int32_t accumulate);
int voxe_cpu_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                       int64_t n, float lr, float beta1, float beta2, float eps, int64_t step);
int voxe_cpu_upsample_trilinear(const float* src, int32_t X, int32_t Y, int32_t Z, int32_t C,
                                float* dst, int32_t X2, int32_t Y2, int32_t Z2);
int voxe_cpu_graph_build(const float* density_grid, const float* feature_grid,
                         int32_t X, int32_t Y, int32_t Z, int32_t F, float sigma, int32_t dilate_yz,
                         uint8_t* node_mask, int32_t* cap);
int voxe_cpu_graphcut(const uint8_t* node_mask, const int8_t* terminal, int32_t* cap,
                      int32_t X, int32_t Y, int32_t Z, uint8_t* segment, int64_t* flow);
int voxe_cpu_cc_largest_k(const uint8_t* mask, int32_t X, int32_t Y, int32_t Z, int32_t k,
                          int32_t* labels, int32_t* num_components);
/* number of OpenMP threads the oracle will use (1 when built without OpenMP) */
int voxe_cpu_num_threads(void);
/* jitter value the HIP kernels draw for (seed, offset, ray, sample): lets tests replay the stream */
float voxe_cpu_jitter_uniform(uint64_t seed, uint64_t rng_offset, int64_t ray, int32_t sample);

#ifdef __cplusplus
}
#endif
#endif /* VOXE_H_ */
