/* A consumer of include/voxe.h written in C (compiled with gcc, no C++, no Python, no torch): what a maintainer of a C / cgo / JNI
 * host would write.  Renders a small SH-0 grid forward and backward through the C ABI with device memory from the HIP runtime's C
 * API and writes inputs-independent outputs to a binary file that tests/test_c_abi_consumer_gpu.py compares with the Python
 * binding's results for the same inputs (same kernels: bit-identical forward).
 *   usage: voxe_c_consumer <inputs.bin> <outputs.bin>
 *   inputs.bin  : int32 X, Y, Z, H, W, S; float focal; float rot[9], trans[3]; float densities[X*Y*Z], features[X*Y*Z*3],
 *                 d_colour[H*W*3]
 *   outputs.bin : float colour[H*W*3], depth[H*W], acc[H*W], d_densities[X*Y*Z], d_features[X*Y*Z*3]                        */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "voxe.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_VOXE(x) do { int s_ = (x); if (s_ != VOXE_OK) { fprintf(stderr, "voxe: %s at %s:%d\n", voxe_strerror(s_), __FILE__, __LINE__); return 3; } } while (0)

static float* to_device(const float* host, size_t n) {
  float* d = NULL;
  if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess) return NULL;
  if (host && hipMemcpy(d, host, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return NULL;
  return d;
}

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s inputs.bin outputs.bin\n", argv[0]); return 1; }
  if (voxe_abi_version() != VOXE_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
  char name[64];
  CHECK_VOXE(voxe_device_check(name, sizeof(name)));
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  int32_t dims[6];
  float cam[13];
  if (fread(dims, sizeof(int32_t), 6, f) != 6 || fread(cam, sizeof(float), 13, f) != 13) return 1;
  const int X = dims[0], Y = dims[1], Z = dims[2], H = dims[3], W = dims[4], S = dims[5];
  const size_t nvox = (size_t)X * Y * Z, R = (size_t)H * W;
  float* h_dens = (float*)malloc(nvox * sizeof(float));
  float* h_feat = (float*)malloc(nvox * 3 * sizeof(float));
  float* h_gc = (float*)malloc(R * 3 * sizeof(float));
  if (fread(h_dens, sizeof(float), nvox, f) != nvox || fread(h_feat, sizeof(float), nvox * 3, f) != nvox * 3 ||
      fread(h_gc, sizeof(float), R * 3, f) != R * 3) return 1;
  fclose(f);

  float *dens = to_device(h_dens, nvox), *feat = to_device(h_feat, nvox * 3), *gc = to_device(h_gc, R * 3);
  float *rays_o = to_device(NULL, R * 3), *rays_d = to_device(NULL, R * 3);
  float *colour = to_device(NULL, R * 3), *depth = to_device(NULL, R), *acc = to_device(NULL, R);
  float *d_dens = to_device(NULL, nvox), *d_feat = to_device(NULL, nvox * 3);
  if (!dens || !feat || !gc || !rays_o || !rays_d || !colour || !depth || !acc || !d_dens || !d_feat) return 2;

  /* VoxelGrid(voxel_size = 3 / N, softplus densities, expected_density_scale 10): AABB [-1.5, 1.5]^3 */
  VoxeGridDesc g;
  memset(&g, 0, sizeof(g));
  g.densities = dens; g.features = feat;
  g.X = X; g.Y = Y; g.Z = Z; g.F = 3;
  for (int a = 0; a < 3; ++a) {
    g.aabb_lo[a] = -1.5f; g.aabb_hi[a] = 1.5f;
    g.norm_scale[a] = 2.0f / (g.aabb_hi[a] - g.aabb_lo[a]);        /* adjust_dynamic_range(slack=True), imaging_utils.py:57-63 */
    g.norm_bias[a] = -1.0f - g.aabb_lo[a] * g.norm_scale[a];
  }
  g.density_scale = 10.0f;
  g.density_pre_act = VOXE_ACT_IDENTITY;
  g.density_post_act = VOXE_ACT_SOFTPLUS;
  g.feature_kind = VOXE_FEAT_SH;

  VoxeRenderCfg c;
  memset(&c, 0, sizeof(c));                                         /* dispatch = NULL: the shipped kernels */
  c.num_samples = S; c.near = 1.8f; c.far = 6.6f;
  c.perturb = 1; c.white_bkgd = 1; c.sh_degree = 0;
  c.seed = 7; c.rng_offset = 11;
  c.image_width = W;

  CHECK_VOXE(voxe_cast_rays(H, W, cam[0], cam + 1, cam + 10, rays_o, rays_d, NULL));
  const size_t wsb = voxe_workspace_bytes(&g, &c, (int64_t)R);
  void* ws = NULL;
  CHECK_HIP(hipMalloc(&ws, wsb));
  CHECK_VOXE(voxe_render_fwd(&g, &c, rays_o, rays_d, (int64_t)R, NULL, colour, depth, acc, NULL, ws, wsb, NULL));
  c.reuse_packed_grid = 1;
  c.ray_state_valid = 1;          /* the workspace holds this render's forward: the library checks the claim */
  CHECK_VOXE(voxe_render_bwd(&g, &c, rays_o, rays_d, (int64_t)R, NULL, colour, depth, acc, gc, NULL, NULL, d_dens, d_feat, 0, ws, wsb, NULL));
  CHECK_HIP(hipDeviceSynchronize());

  FILE* o = fopen(argv[2], "wb");
  if (!o) return 1;
  const struct { float* p; size_t n; } outs[5] = {{colour, R * 3}, {depth, R}, {acc, R}, {d_dens, nvox}, {d_feat, nvox * 3}};
  for (int i = 0; i < 5; ++i) {
    float* h = (float*)malloc(outs[i].n * sizeof(float));
    CHECK_HIP(hipMemcpy(h, outs[i].p, outs[i].n * sizeof(float), hipMemcpyDeviceToHost));
    fwrite(h, sizeof(float), outs[i].n, o);
    free(h);
  }
  fclose(o);
  printf("voxe C consumer ok on %s: %zu rays, workspace %zu bytes\n", name, R, wsb);
  return 0;
}
