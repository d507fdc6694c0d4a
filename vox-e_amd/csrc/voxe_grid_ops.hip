// voxe_grid_ops.hip -- ray casting and the whole-grid (HBM-streaming) passes of an SDS /
// reconstruction step: density-correlation loss, total-variation loss, Adam, trilinear upsampling.
#include "voxe_device.hpp"
#include "voxe_launch.hpp"

namespace voxe {

// ------------------------------------------------------------------------------------------------
// cast_rays -- rendering/volumetric/utils/misc.py:12-50
// ------------------------------------------------------------------------------------------------
struct Pose {
  float rot[9];
  float trans[3];
};

__global__ __launch_bounds__(256) void cast_rays_kernel(int H, int W, float focal, Pose pose,
                                                        float* __restrict__ rays_o,
                                                        float* __restrict__ rays_d) {
  const long long n = (long long)H * W;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int py = (int)(i / W), px = (int)(i - (long long)py * W);
  const float x = (float)px + 0.5f, y = (float)py + 0.5f;  // linspace(0.5, W-0.5, W) (misc.py:29-33)
  const float dx = (x - (float)W * 0.5f) / focal;            // misc.py:39-45
  const float dy = -(y - (float)H * 0.5f) / focal;
  const float dz = -1.0f;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    rays_d[3 * i + r] = pose.rot[3 * r + 0] * dx + pose.rot[3 * r + 1] * dy + pose.rot[3 * r + 2] * dz;
    rays_o[3 * i + r] = pose.trans[r];
  }
}

void launch_cast_rays(int H, int W, float focal, const float* rot, const float* trans, float* rays_o,
                      float* rays_d, hipStream_t st) {
  Pose p;
  for (int i = 0; i < 9; ++i) p.rot[i] = rot[i];
  for (int i = 0; i < 3; ++i) p.trans[i] = trans[i];
  const long long n = (long long)H * W;
  cast_rays_kernel<<<(int)((n + 255) / 256), 256, 0, st>>>(H, W, focal, p, rays_o, rays_d);
}

__device__ __forceinline__ void cast_indexed_ray(int H, int W, float focal, int K, const float* __restrict__ poses, long long f,
                                                 long long i, float* __restrict__ rays_o, float* __restrict__ rays_d) {
  const long long per = (long long)H * W;
  long long cam = f / per;
  const long long rem = f - cam * per;
  cam = cam < 0 ? 0 : (cam >= K ? K - 1 : cam);  // the API validates the range; never read out of bounds
  const int py = (int)(rem / W), px = (int)(rem - (long long)py * W);
  const float* pose = poses + cam * 12;            // [3,4] = rotation | translation
  const float x = (float)px + 0.5f, y = (float)py + 0.5f;
  const float dx = (x - (float)W * 0.5f) / focal;
  const float dy = -(y - (float)H * 0.5f) / focal;
  const float dz = -1.0f;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    rays_d[3 * i + r] = pose[4 * r + 0] * dx + pose[4 * r + 1] * dy + pose[4 * r + 2] * dz;
    rays_o[3 * i + r] = pose[4 * r + 3];
  }
}
__global__ __launch_bounds__(256) void cast_rays_indexed_kernel(int H, int W, float focal, int K,
                                                                const float* __restrict__ poses,
                                                                const long long* __restrict__ flat_index, long long B,
                                                                float* __restrict__ rays_o, float* __restrict__ rays_d) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  cast_indexed_ray(H, W, focal, K, poses, flat_index[i], i, rays_o, rays_d);
}

void launch_cast_rays_indexed(int H, int W, float focal, const float* poses, int K, const long long* flat_index,
                              long long B, float* rays_o, float* rays_d, hipStream_t st) {
  cast_rays_indexed_kernel<<<(int)((B + 255) / 256), 256, 0, st>>>(H, W, focal, K, poses, flat_index, B, rays_o,
                                                                    rays_d);
}

__global__ __launch_bounds__(256) void random_subset_kernel(uint32_t n, long long count, int half, uint32_t key0,
                                                           uint32_t key1, long long* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = (long long)feistel_permute((uint32_t)i, n, half, key0, key1);
}

struct SubsetKeys { int half; uint32_t key0, key1; };
static SubsetKeys subset_keys(long long n, unsigned long long seed, unsigned long long rng_offset) {
  int bits = 2;
  while ((1ull << bits) < (unsigned long long)n) bits += 2;
  return {bits / 2, mix32((uint32_t)seed ^ ((uint32_t)rng_offset * 0x9E3779B1u)),
          mix32((uint32_t)(seed >> 32) ^ (uint32_t)(rng_offset >> 32) ^ 0x7F4A7C15u)};
}
void launch_random_subset(long long n, long long count, unsigned long long seed, unsigned long long rng_offset,
                          long long* out, hipStream_t st) {
  const SubsetKeys k = subset_keys(n, seed, rng_offset);
  random_subset_kernel<<<(int)((count + 255) / 256), 256, 0, st>>>((uint32_t)n, count, k.half, k.key0, k.key1, out);
}

// ------------------------------------------------------------------------------------------------
// block reduction helper (double): wave shuffle -> LDS -> lane 0
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* out /* NV doubles, global */) {
  __shared__ double sm[NV][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const double s = wave_sum(v[i]);
    if (lane == 0) sm[i][wave] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    const int i = threadIdx.x;
    out[i] = (sm[i][0] + sm[i][1]) + (sm[i][2] + sm[i][3]);
  }
}

// ------------------------------------------------------------------------------------------------
// _density_correlation_loss -- modules/sds_trainer.py:507-524 (+ autograd). Three launches:
// moments (1 read of a and b), finalize (stats + loss), gradient (1 read of a and b, 1 write).
// Deterministic: per-block partials, fixed-order final sum.
// ------------------------------------------------------------------------------------------------
constexpr int kRedBlocks = 1024;

__global__ __launch_bounds__(256) void dcl_moments_kernel(const float* __restrict__ a,
                                                          const float* __restrict__ b, long long n,
                                                          double* __restrict__ partial) {
  double s[5] = {0, 0, 0, 0, 0};
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double x = a[i], y = b[i];
    s[0] += x; s[1] += y; s[2] += x * x; s[3] += y * y; s[4] += x * y;
  }
  block_sum<5>(s, partial + (long long)blockIdx.x * 5);
}

// stats: [0]=mean a, [1]=mean b, [2]=k1, [3]=k2
__global__ __launch_bounds__(256) void dcl_finalize_kernel(const double* __restrict__ partial,
                                                           int nblocks, long long n,
                                                           double* __restrict__ stats,
                                                           float* __restrict__ loss_out, double fold_scale = 1.0) {
  double s[5] = {0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x)
#pragma unroll
    for (int j = 0; j < 5; ++j) s[j] += partial[(long long)i * 5 + j];
  __shared__ double tot[5];
  block_sum<5>(s, tot);
  __syncthreads();
  if (threadIdx.x == 0) {
    const double dn = (double)n;
    const double ma = tot[0] / dn, mb = tot[1] / dn;
    const double va = tot[2] / dn - ma * ma, vb = tot[3] / dn - mb * mb;
    const double cov = tot[4] / dn - ma * mb;
    const double eps = 0.0000001;
    const double D = sqrt(fmax(va * vb, 0.0));
    if (loss_out) *loss_out = (float)(1.0 - cov / (D + eps));
    stats[0] = ma; stats[1] = mb;
    stats[2] = fold_scale / (dn * (D + eps));
    stats[3] = (D > 0.0) ? fold_scale * cov * vb / (dn * D * (D + eps) * (D + eps)) : 0.0;
  }
}

__global__ __launch_bounds__(256) void dcl_grad_kernel(const float* __restrict__ a,
                                                       const float* __restrict__ b, long long n,
                                                       const double* __restrict__ stats,
                                                       float grad_scale, float* __restrict__ d_a,
                                                       int accumulate) {
  const float ma = (float)stats[0], mb = (float)stats[1];
  const float k1 = (float)(stats[2] * (double)grad_scale), k2 = (float)(stats[3] * (double)grad_scale);
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float A = a[i] - ma, B = b[i] - mb;
    const float gv = A * k2 - B * k1;
    d_a[i] = accumulate ? d_a[i] + gv : gv;
  }
}

size_t dcl_scratch_bytes(long long) { return sizeof(double) * (kRedBlocks * 5 + 8); }

const double* launch_dcl_moments(const float* a, const float* b, long long n, float grad_scale, float* loss_out, void* scratch,
                                 hipStream_t st) {
  double* partial = (double*)scratch;
  double* stats = partial + kRedBlocks * 5;
  const int nb = (int)((n + 255) / 256 < kRedBlocks ? (n + 255) / 256 : kRedBlocks);
  dcl_moments_kernel<<<nb, 256, 0, st>>>(a, b, n, partial);
  dcl_finalize_kernel<<<1, 256, 0, st>>>(partial, nb, n, stats, loss_out, (double)grad_scale);
  return stats;
}

void launch_dcl(const float* a, const float* b, long long n, float grad_scale, float* loss_out,
                float* d_a, int accumulate, void* scratch, hipStream_t st) {
  double* partial = (double*)scratch;
  double* stats = partial + kRedBlocks * 5;
  const int nb = (int)((n + 255) / 256 < kRedBlocks ? (n + 255) / 256 : kRedBlocks);
  dcl_moments_kernel<<<nb, 256, 0, st>>>(a, b, n, partial);
  dcl_finalize_kernel<<<1, 256, 0, st>>>(partial, nb, n, stats, loss_out);
  if (d_a) {
    const int nbg = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    dcl_grad_kernel<<<nbg, 256, 0, st>>>(a, b, n, stats, grad_scale, d_a, accumulate);
  }
}

// ------------------------------------------------------------------------------------------------
// density_correlation_loss_fn, l2_mode / l1_mode -- modules/sds_trainer.py:494-503 (+ autograd): torch mse_loss / l1_loss of
// the two density grids (mean reduction).  The gradient needs no reduction: (a - b) * (2 w / n)  /  sign(a - b) * (w / n).
// _feature_correlation_loss -- modules/sds_trainer.py:526-534 (+ autograd): per voxel D = sum_c (sigmoid(f_c) - sigmoid(r_c)),
// loss = sum_v D^2, d f_c = 2 w D sigmoid(f_c) (1 - sigmoid(f_c)).  Values: per-block double partials, fixed-order final sum.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoid_ref(float x) { return 1.0f / (1.0f + expf(-x)); }   // (torch.sigmoid; not the render's fast path)

__global__ __launch_bounds__(256) void density_diff_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                                           int kind, float k, double* __restrict__ partial,
                                                           float* __restrict__ d_a, int accumulate) {
  double s[1] = {0.0};
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float d = a[i] - b[i];
    if (partial) s[0] += kind == VOXE_DREG_L2 ? (double)d * (double)d : (double)fabsf(d);
    if (d_a) {
      const float gv = kind == VOXE_DREG_L2 ? d * k : (d > 0.0f ? k : (d < 0.0f ? -k : 0.0f));
      d_a[i] = accumulate ? d_a[i] + gv : gv;
    }
  }
  if (partial) block_sum<1>(s, partial + blockIdx.x);
}

// loss = (sum of the partials) * norm
__global__ __launch_bounds__(256) void sum_finalize_kernel(const double* __restrict__ partial, int nblocks, double norm,
                                                           float* __restrict__ loss_out) {
  double s[1] = {0.0};
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x) s[0] += partial[i];
  __shared__ double tot[1];
  block_sum<1>(s, tot);
  __syncthreads();
  if (threadIdx.x == 0) *loss_out = (float)(tot[0] * norm);
}

// loss_norm: 1 / (elements of the mean) -- the whole grid's count even when [a, a + n) is a slab of it
void launch_density_diff(const float* a, const float* b, long long n, int kind, float grad_scale, float* loss_out, float loss_norm,
                         float* d_a, int accumulate, void* scratch, hipStream_t st) {
  double* partial = loss_out ? (double*)scratch : nullptr;
  const int nb = (int)((n + 255) / 256 < kRedBlocks ? (n + 255) / 256 : kRedBlocks);
  const float k = (float)((kind == VOXE_DREG_L2 ? 2.0 : 1.0) * (double)grad_scale * (double)loss_norm);
  density_diff_kernel<<<nb, 256, 0, st>>>(a, b, n, kind, k, partial, d_a, accumulate);
  if (loss_out) sum_finalize_kernel<<<1, 256, 0, st>>>(partial, nb, (double)loss_norm, loss_out);
}

__global__ __launch_bounds__(256) void feature_correlation_kernel(const float* __restrict__ f, const float* __restrict__ r,
                                                                  long long nvox, int F, float k2, double* __restrict__ partial,
                                                                  float* __restrict__ d_f, int accumulate) {
  double s[1] = {0.0};
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += stride) {
    float D = 0.0f;
    for (int c = 0; c < F; ++c) D += sigmoid_ref(f[v * F + c]) - sigmoid_ref(r[v * F + c]);
    if (partial) s[0] += (double)D * (double)D;
    if (d_f) {
      for (int c = 0; c < F; ++c) {
        const float sc = sigmoid_ref(f[v * F + c]);
        const float gv = (k2 * D) * ((1.0f - sc) * sc);
        d_f[v * F + c] = accumulate ? d_f[v * F + c] + gv : gv;
      }
    }
  }
  if (partial) block_sum<1>(s, partial + blockIdx.x);
}

void launch_feature_correlation(const float* f, const float* r, long long nvox, int F, float grad_scale, float* loss_out, float* d_f,
                                int accumulate, void* scratch, hipStream_t st) {
  double* partial = loss_out ? (double*)scratch : nullptr;
  const int nb = (int)((nvox + 255) / 256 < kRedBlocks ? (nvox + 255) / 256 : kRedBlocks);
  feature_correlation_kernel<<<nb, 256, 0, st>>>(f, r, nvox, F, 2.0f * grad_scale, partial, d_f, accumulate);
  if (loss_out) sum_finalize_kernel<<<1, 256, 0, st>>>(partial, nb, 1.0, loss_out);
}

// ------------------------------------------------------------------------------------------------
// _tv_loss_on_grid -- modules/sds_trainer.py:563-567 (+ autograd), gather form (no atomics):
// every element reads its 6 axis neighbours once.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sgnf(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

// Thread mapping: blockIdx.y walks x-planes, blockIdx.x / threadIdx.x the Y * Z * C elements of a plane (contiguous in
// memory: coalesced), both with a stride, so that (x, y, z, channel) come from ONE 32-bit division per element (two for
// channel counts other than 1 / 3) instead of 64-bit div / mod chains (which made this streaming pass instruction bound:
// 0.7 TB/s at 160^3 before, r03).  Same per-element arithmetic as before; the double partial sums only change their order.
template <int CT>   // CT = channel count known at compile time (1, 3) or 0: run-time C
__global__ __launch_bounds__(256) void tv_kernel(const float* __restrict__ grid, int X, int Y, int Z,
                                                 int C_rt, float gx, float gy, float gz,
                                                 double* __restrict__ partial,
                                                 float* __restrict__ d_grid, int accumulate) {
  const unsigned C = CT > 0 ? (unsigned)CT : (unsigned)C_rt;
  const unsigned sy = (unsigned)Z * C, plane = (unsigned)Y * sy;   // elements of a y-row / an x-plane (< 2^31: validated on the host)
  double s[3] = {0, 0, 0};
  for (unsigned x = blockIdx.y; x < (unsigned)X; x += gridDim.y) {
    const long long base = (long long)x * plane;
    for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < plane; e += gridDim.x * 256u) {
      const unsigned y = e / sy, zc = e - y * sy, z = zc / C;
      const long long i = base + e;
      const float v = grid[i];
      float g = 0.0f;
      if (x + 1 < (unsigned)X) { const float df = grid[i + plane] - v; s[0] += fabsf(df); g -= sgnf(df) * gx; }
      if (x > 0) { const float df = v - grid[i - plane]; g += sgnf(df) * gx; }
      if (y + 1 < (unsigned)Y) { const float df = grid[i + sy] - v; s[1] += fabsf(df); g -= sgnf(df) * gy; }
      if (y > 0) { const float df = v - grid[i - sy]; g += sgnf(df) * gy; }
      if (z + 1 < (unsigned)Z) { const float df = grid[i + C] - v; s[2] += fabsf(df); g -= sgnf(df) * gz; }
      if (z > 0) { const float df = v - grid[i - C]; g += sgnf(df) * gz; }
      if (d_grid) d_grid[i] = accumulate ? d_grid[i] + g : g;
    }
  }
  block_sum<3>(s, partial + (long long)(blockIdx.y * gridDim.x + blockIdx.x) * 3);
}

// The same pass with FOUR consecutive elements per thread (rows of Z * C floats that are a multiple of 4: 16-byte aligned
// vectors that never straddle a y-row): 7 vector loads per 4 elements (the element vector, its x / y neighbours, the next
// and the previous vector of the row for the z neighbours) instead of 28 scalar ones.  C = 1 (densities, attention grids)
// and C = 3 (SH-0 features); everything else takes tv_kernel.  Same per-element arithmetic.
template <int C>
__global__ __launch_bounds__(256) void tv_kernel_v4(const float* __restrict__ grid, int X, int Y, int Z, float gx, float gy,
                                                    float gz, double* __restrict__ partial, float* __restrict__ d_grid,
                                                    int accumulate) {
  static_assert(C == 1 || C == 3, "vector TV pass: 1 or 3 channels");
  const unsigned sy = (unsigned)Z * C, plane = (unsigned)Y * sy, sy4 = sy >> 2, plane4 = plane >> 2;
  double s[3] = {0, 0, 0};
  for (unsigned x = blockIdx.y; x < (unsigned)X; x += gridDim.y) {
    const float4* __restrict__ base4 = reinterpret_cast<const float4*>(grid + (long long)x * plane);
    for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < plane4; q += gridDim.x * 256u) {
      const unsigned y = q / sy4, zq = q - y * sy4;            // zq: vector index inside the row
      const float4 c4 = base4[q];
      const float v[4] = {c4.x, c4.y, c4.z, c4.w};
      float xp[4] = {0, 0, 0, 0}, xm[4] = {0, 0, 0, 0}, yp[4] = {0, 0, 0, 0}, ym[4] = {0, 0, 0, 0};
      const bool hxp = x + 1 < (unsigned)X, hxm = x > 0, hyp = y + 1 < (unsigned)Y, hym = y > 0;
      if (hxp) { const float4 t = base4[q + plane4]; xp[0] = t.x; xp[1] = t.y; xp[2] = t.z; xp[3] = t.w; }
      if (hxm) { const float4 t = (base4 - plane4)[q]; xm[0] = t.x; xm[1] = t.y; xm[2] = t.z; xm[3] = t.w; }
      if (hyp) { const float4 t = base4[q + sy4]; yp[0] = t.x; yp[1] = t.y; yp[2] = t.z; yp[3] = t.w; }
      if (hym) { const float4 t = base4[q - sy4]; ym[0] = t.x; ym[1] = t.y; ym[2] = t.z; ym[3] = t.w; }
      float4 nx = make_float4(0, 0, 0, 0), pv = make_float4(0, 0, 0, 0);
      if (zq + 1 < sy4) nx = base4[q + 1];
      if (zq > 0) pv = base4[q - 1];
      // element j's z neighbours are j +- C floats away
      float zp[4], zm[4];
      if (C == 1) {
        zp[0] = v[1]; zp[1] = v[2]; zp[2] = v[3]; zp[3] = nx.x;
        zm[0] = pv.w; zm[1] = v[0]; zm[2] = v[1]; zm[3] = v[2];
      } else {
        zp[0] = v[3]; zp[1] = nx.x; zp[2] = nx.y; zp[3] = nx.z;
        zm[0] = pv.y; zm[1] = pv.z; zm[2] = pv.w; zm[3] = v[0];
      }
      float g[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned z = (zq * 4u + (unsigned)j) / (unsigned)C;
        float gj = 0.0f;
        if (hxp) { const float df = xp[j] - v[j]; s[0] += fabsf(df); gj -= sgnf(df) * gx; }
        if (hxm) { const float df = v[j] - xm[j]; gj += sgnf(df) * gx; }
        if (hyp) { const float df = yp[j] - v[j]; s[1] += fabsf(df); gj -= sgnf(df) * gy; }
        if (hym) { const float df = v[j] - ym[j]; gj += sgnf(df) * gy; }
        if (z + 1 < (unsigned)Z) { const float df = zp[j] - v[j]; s[2] += fabsf(df); gj -= sgnf(df) * gz; }
        if (z > 0) { const float df = v[j] - zm[j]; gj += sgnf(df) * gz; }
        g[j] = gj;
      }
      if (d_grid) {
        float4* __restrict__ out4 = reinterpret_cast<float4*>(d_grid + (long long)x * plane) + q;
        float4 o = make_float4(g[0], g[1], g[2], g[3]);
        if (accumulate) { const float4 old = *out4; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
        *out4 = o;
      }
    }
  }
  block_sum<3>(s, partial + (long long)(blockIdx.y * gridDim.x + blockIdx.x) * 3);
}

__global__ __launch_bounds__(256) void tv_finalize_kernel(const double* __restrict__ partial,
                                                          int nblocks, double cx, double cy, double cz,
                                                          float* __restrict__ loss_out) {
  double s[3] = {0, 0, 0};
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x)
#pragma unroll
    for (int j = 0; j < 3; ++j) s[j] += partial[(long long)i * 3 + j];
  __shared__ double tot[3];
  block_sum<3>(s, tot);
  __syncthreads();
  // mean over an empty diff (dim of size 1) is NaN in torch
  if (threadIdx.x == 0) *loss_out = (float)((tot[0] / cx + tot[1] / cy + tot[2] / cz) / 3.0);
}

// (r03: 16384 blocks instead of 1024 -- with 47 dependent-latency iterations per thread the pass ran at 0.9 TB/s; the
//  partial sums stay per block and are folded in a fixed order: the loss is bit-reproducible)
constexpr int kTvBlocks = 8192;
size_t tv_scratch_bytes(int, int, int, int) { return sizeof(double) * (kTvBlocks * 3 + 8); }

void launch_tv(const float* grid, int X, int Y, int Z, int C, float grad_scale, float* loss_out,
               float* d_grid, int accumulate, void* scratch, hipStream_t st) {
  const long long n = (long long)X * Y * Z * C;
  const double cx = (double)(X - 1) * Y * Z * C, cy = (double)X * (Y - 1) * Z * C,
               cz = (double)X * Y * (Z - 1) * C;
  const float gx = cx > 0 ? (float)((double)grad_scale / (3.0 * cx)) : 0.f;
  const float gy = cy > 0 ? (float)((double)grad_scale / (3.0 * cy)) : 0.f;
  const float gz = cz > 0 ? (float)((double)grad_scale / (3.0 * cz)) : 0.f;
  double* partial = (double*)scratch;
  (void)n;
  // at most kTvBlocks blocks (the scratch holds their partial sums): bx along a plane, by over the planes
  const long long plane = (long long)Y * Z * C;
  const int bx = (int)((plane + 255) / 256 < 128 ? (plane + 255) / 256 : 128);
  const int by = X < kTvBlocks / bx ? X : kTvBlocks / bx;
  if ((C == 1 || C == 3) && ((long long)Z * C) % 4 == 0 && ((uintptr_t)grid % 16) == 0 && (!d_grid || ((uintptr_t)d_grid % 16) == 0)) {
    const long long plane4 = plane / 4;
    const int vx = (int)((plane4 + 255) / 256 < 64 ? (plane4 + 255) / 256 : 64);
    const int vy = X < kTvBlocks / vx ? X : kTvBlocks / vx;
    if (C == 1) tv_kernel_v4<1><<<dim3(vx, vy), 256, 0, st>>>(grid, X, Y, Z, gx, gy, gz, partial, d_grid, accumulate);
    else tv_kernel_v4<3><<<dim3(vx, vy), 256, 0, st>>>(grid, X, Y, Z, gx, gy, gz, partial, d_grid, accumulate);
    tv_finalize_kernel<<<1, 256, 0, st>>>(partial, vx * vy, cx, cy, cz, loss_out);
    return;
  }
  const dim3 gridsz(bx, by);
  if (C == 1) tv_kernel<1><<<gridsz, 256, 0, st>>>(grid, X, Y, Z, C, gx, gy, gz, partial, d_grid, accumulate);
  else if (C == 3) tv_kernel<3><<<gridsz, 256, 0, st>>>(grid, X, Y, Z, C, gx, gy, gz, partial, d_grid, accumulate);
  else tv_kernel<0><<<gridsz, 256, 0, st>>>(grid, X, Y, Z, C, gx, gy, gz, partial, d_grid, accumulate);
  tv_finalize_kernel<<<1, 256, 0, st>>>(partial, bx * by, cx, cy, cz, loss_out);
}

// ------------------------------------------------------------------------------------------------
// torch.optim.Adam (single tensor, no weight decay / amsgrad): one read-modify-write stream over
// param, exp_avg, exp_avg_sq + one read of grad  (7 * n * 4 bytes).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   long long n, float step_size, float bc2_sqrt,
                                                   float beta1, float beta2, float eps) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const float omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * omb1;              // exp_avg.lerp_(grad, 1-beta1)
    const float vi = v[i] * beta2 + (omb2 * gi) * gi;        // mul_(beta2).addcmul_(g, g, 1-beta2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);                  // addcdiv_(exp_avg, denom, -step_size)
    m[i] = mi;
    v[i] = vi;
  }
}

void launch_adam(float* param, const float* grad, float* m, float* v, long long n, float lr,
                 float beta1, float beta2, float eps, long long step, hipStream_t st) {
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  if (n == 0) return;
  const int nb = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  adam_kernel<<<nb, 256, 0, st>>>(param, grad, m, v, n, step_size, bc2_sqrt, beta1, beta2, eps);
}

// ------------------------------------------------------------------------------------------------
// scale_voxel_grid_with_required_output_size -- voxels.py:409-447 (ATen upsample_trilinear3d,
// align_corners=False, scale = in/out)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void up_axis(int out_i, int in_n, int out_n, int& i0, int& i1, float& l0,
                                        float& l1) {
  const float scale = (float)in_n / (float)out_n;
  float src = scale * ((float)out_i + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  int a = (int)src;
  if (a > in_n - 1) a = in_n - 1;
  i0 = a;
  i1 = a + ((a < in_n - 1) ? 1 : 0);
  l1 = src - (float)a;
  l0 = 1.0f - l1;
}

// Thread mapping: blockIdx.y = output x-plane, blockIdx.x / threadIdx.x = the Y2 * Z2 voxels of the plane; one thread per
// output VOXEL computes the three axis interpolations once and loops over the C channels (contiguous per corner) -- the
// index math (float ops + 64-bit div / mod per ELEMENT before r03) was what bounded this pass, not its bytes.
template <int CT>   // channel count known at compile time (1, 3) or 0: run-time C
__global__ __launch_bounds__(256) void upsample_kernel(const float* __restrict__ src, int X, int Y,
                                                       int Z, int C_rt, float* __restrict__ dst, int X2,
                                                       int Y2, int Z2) {
  const int C = CT > 0 ? CT : C_rt;
  const unsigned plane2 = (unsigned)Y2 * (unsigned)Z2;
  const unsigned e = blockIdx.x * 256u + threadIdx.x;
  if (e >= plane2) return;
  const int x = blockIdx.y;
  const int y = (int)(e / (unsigned)Z2), z = (int)(e - (unsigned)y * (unsigned)Z2);
  int x0, x1, y0, y1, z0, z1;
  float lx0, lx1, ly0, ly1, lz0, lz1;
  up_axis(x, X, X2, x0, x1, lx0, lx1);
  up_axis(y, Y, Y2, y0, y1, ly0, ly1);
  up_axis(z, Z, Z2, z0, z1, lz0, lz1);
  auto at = [&](int ix, int iy, int iz) { return src + (((long long)ix * Y + iy) * Z + iz) * C; };
  const float* __restrict__ p000 = at(x0, y0, z0); const float* __restrict__ p001 = at(x0, y0, z1);
  const float* __restrict__ p010 = at(x0, y1, z0); const float* __restrict__ p011 = at(x0, y1, z1);
  const float* __restrict__ p100 = at(x1, y0, z0); const float* __restrict__ p101 = at(x1, y0, z1);
  const float* __restrict__ p110 = at(x1, y1, z0); const float* __restrict__ p111 = at(x1, y1, z1);
  float* __restrict__ out = dst + ((long long)x * plane2 + e) * C;
  auto channel = [&](int ch) {
    // (the same expression as before, operand for operand)
    out[ch] = lx0 * (ly0 * (lz0 * p000[ch] + lz1 * p001[ch]) + ly1 * (lz0 * p010[ch] + lz1 * p011[ch])) +
              lx1 * (ly0 * (lz0 * p100[ch] + lz1 * p101[ch]) + ly1 * (lz0 * p110[ch] + lz1 * p111[ch]));
  };
  if constexpr (CT > 0) {
#pragma unroll
    for (int ch = 0; ch < CT; ++ch) channel(ch);
  } else {
    for (int ch = 0; ch < C; ++ch) channel(ch);   // (run-time channel count: nothing to unroll)
  }
}

void launch_upsample(const float* src, int X, int Y, int Z, int C, float* dst, int X2, int Y2, int Z2,
                     hipStream_t st) {
  const long long plane2 = (long long)Y2 * Z2;
  const dim3 gridsz((unsigned)((plane2 + 255) / 256), (unsigned)X2);
  if (C == 1) upsample_kernel<1><<<gridsz, 256, 0, st>>>(src, X, Y, Z, C, dst, X2, Y2, Z2);
  else if (C == 3) upsample_kernel<3><<<gridsz, 256, 0, st>>>(src, X, Y, Z, C, dst, X2, Y2, Z2);
  else upsample_kernel<0><<<gridsz, 256, 0, st>>>(src, X, Y, Z, C, dst, X2, Y2, Z2);
}

// ------------------------------------------------------------------------------------------------
// disparity = 1 / max(1e-10, depth / acc)  (accumulate.py:85-88): chain rule of an upstream d_disparity into d_depth and
// d_acc (what autograd does through the reference's three tensor ops), one thread per ray.  Rays whose quotient is below
// the clamp, NaN (acc == 0: the reference's disparity is NaN there) or infinite receive no gradient.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float finite_or_zero(float x) { return (x - x == 0.0f) ? x : 0.0f; }   // NaN / +-inf -> 0

__global__ __launch_bounds__(256) void disparity_bwd_kernel(const float* __restrict__ depth, const float* __restrict__ acc,
                                                            const float* __restrict__ d_disp,
                                                            const float* __restrict__ d_depth_in,
                                                            const float* __restrict__ d_acc_in, float* __restrict__ d_depth_out,
                                                            float* __restrict__ d_acc_out, long long R) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float dep = depth[r], a = acc[r];
  const float q = dep / a;
  const float live = q > 1e-10f ? 1.0f : 0.0f;
  const float dq = finite_or_zero((-d_disp[r] / (q * q)) * live);
  const float gd = finite_or_zero(dq / a);
  const float ga = finite_or_zero((-dq * dep) / (a * a));
  d_depth_out[r] = d_depth_in ? d_depth_in[r] + gd : gd;
  d_acc_out[r] = d_acc_in ? d_acc_in[r] + ga : ga;
}

void launch_disparity_bwd(const float* depth, const float* acc, const float* d_disp, const float* d_depth_in,
                          const float* d_acc_in, float* d_depth_out, float* d_acc_out, long long R, hipStream_t st) {
  if (R <= 0) return;
  disparity_bwd_kernel<<<(int)((R + 255) / 256), 256, 0, st>>>(depth, acc, d_disp, d_depth_in, d_acc_in, d_depth_out,
                                                               d_acc_out, R);
}

// ------------------------------------------------------------------------------------------------
// Pieces of the fused reconstruction iteration (voxe_recon_step; modules/trainers.py:288-351): target pixels of a random
// batch gathered straight from the image stack, and torch.nn.functional.l1_loss (+ its gradient, + the MSE the trainer
// logs as PSNR) in one pass.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gather_pixel(const float* __restrict__ images, const long long* __restrict__ image_rows,
                                             long long f, long long i, int per, int num_images, float* __restrict__ out) {
  // images [N, 3, H, W]; f: flat (camera, y, x) index over the K cached cameras; out [B, 3]
  const long long cam = f / per, rem = f - cam * per;
  const long long row = image_rows ? image_rows[cam] : cam;
  if (row < 0 || row >= num_images) {   // a caller's bug: never read outside `images`; the NaN target shows up in the loss
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) out[i * 3 + ch] = __int_as_float(0x7fc00000);
    return;
  }
  const float* __restrict__ src = images + row * 3 * (long long)per + rem;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) out[i * 3 + ch] = src[(long long)ch * per];
}
__global__ __launch_bounds__(256) void gather_pixels_kernel(const float* __restrict__ images, const long long* __restrict__ image_rows,
                                                            const long long* __restrict__ subset, long long B, int per,
                                                            int num_images, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  gather_pixel(images, image_rows, subset[i], i, per, num_images, out);
}
// batch assembly of a reconstruction iteration in ONE launch (r04): entry i of the keyed permutation (random_subset_kernel) ->
// its ray (cast_rays_indexed_kernel) and its target pixel (gather_pixels_kernel); three launches of ~5 us each before
__global__ __launch_bounds__(256) void recon_batch_kernel(uint32_t n, long long B, int half, uint32_t key0, uint32_t key1, int H,
                                                          int W, float focal, int K, const float* __restrict__ poses,
                                                          const float* __restrict__ images,
                                                          const long long* __restrict__ image_rows, int num_images,
                                                          long long* __restrict__ subset, float* __restrict__ rays_o,
                                                          float* __restrict__ rays_d, float* __restrict__ target) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const long long f = (long long)feistel_permute((uint32_t)i, n, half, key0, key1);
  subset[i] = f;
  cast_indexed_ray(H, W, focal, K, poses, f, i, rays_o, rays_d);
  gather_pixel(images, image_rows, f, i, H * W, num_images, target);
}

// mean |a - b| over n elements and its gradient w.r.t. a (sign(a - b) / n: torch's l1_loss backward, sign(0) = 0), plus
// the mean squared difference; per-block partial sums in double, folded in a fixed order by l1_finalize_kernel
__global__ __launch_bounds__(256) void l1_loss_grad_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                                           float inv_n, float* __restrict__ d_a, double* __restrict__ partial) {
  // blockIdx.y: render (the renders' a / d_a lie back to back, n elements each, and share the target b)
  a += (long long)blockIdx.y * n;
  d_a += (long long)blockIdx.y * n;
  partial += (long long)blockIdx.y * (kRedBlocks * 2 + 8);
  double s[2] = {0, 0};
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float df = a[i] - b[i];
    s[0] += fabsf(df);
    s[1] += (double)df * (double)df;
    d_a[i] = sgnf(df) * inv_n;
  }
  block_sum<2>(s, partial + (long long)blockIdx.x * 2);
}
__global__ __launch_bounds__(256) void l1_finalize_kernel(const double* __restrict__ partial, int nblocks, double n,
                                                          float* __restrict__ out2) {
  partial += (long long)blockIdx.x * (kRedBlocks * 2 + 8);   // blockIdx.x: render
  out2 += 2 * blockIdx.x;
  double s[2] = {0, 0};
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x) { s[0] += partial[2 * i]; s[1] += partial[2 * i + 1]; }
  __shared__ double tot[2];
  block_sum<2>(s, tot);
  __syncthreads();
  if (threadIdx.x == 0) { out2[0] = (float)(tot[0] / n); out2[1] = (float)(tot[1] / n); }
}

void launch_gather_pixels(const float* images, const long long* image_rows, const long long* subset, long long B, int per,
                          int num_images, float* out, hipStream_t st) {
  if (B <= 0) return;
  gather_pixels_kernel<<<(int)((B + 255) / 256), 256, 0, st>>>(images, image_rows, subset, B, per, num_images, out);
}
void launch_recon_batch(long long B, unsigned long long seed, unsigned long long rng_offset, int H, int W, float focal, int K,
                        const float* poses, const float* images, const long long* image_rows, int num_images, long long* subset,
                        float* rays_o, float* rays_d, float* target, hipStream_t st) {
  if (B <= 0) return;
  const long long n = (long long)K * H * W;
  const SubsetKeys k = subset_keys(n, seed, rng_offset);
  recon_batch_kernel<<<(int)((B + 255) / 256), 256, 0, st>>>((uint32_t)n, B, k.half, k.key0, k.key1, H, W, focal, K, poses, images,
                                                             image_rows, num_images, subset, rays_o, rays_d, target);
}
size_t l1_scratch_bytes() { return 2 * sizeof(double) * (kRedBlocks * 2 + 8); }   // (two renders per launch: launch_l1_loss_grad2)
void launch_l1_loss_grad(const float* a, const float* b, long long n, float* d_a, float* out2, void* scratch, hipStream_t st) {
  double* partial = (double*)scratch;
  const int nb = (int)((n + 255) / 256 < kRedBlocks ? (n + 255) / 256 : kRedBlocks);
  l1_loss_grad_kernel<<<nb, 256, 0, st>>>(a, b, n, (float)(1.0 / (double)n), d_a, partial);
  l1_finalize_kernel<<<1, 256, 0, st>>>(partial, nb, (double)n, out2);
}

// `renders` renders of n elements each, back to back in a / d_a, against the same target; out2[2 * r] = (L1, MSE) of render r
void launch_l1_loss_grad_n(const float* a, const float* b, long long n, int renders, float* d_a, float* out2, void* scratch,
                           hipStream_t st) {
  double* partial = (double*)scratch;
  const int nb = (int)((n + 255) / 256 < kRedBlocks ? (n + 255) / 256 : kRedBlocks);
  l1_loss_grad_kernel<<<dim3((unsigned)nb, (unsigned)renders), 256, 0, st>>>(a, b, n, (float)(1.0 / (double)n), d_a, partial);
  l1_finalize_kernel<<<renders, 256, 0, st>>>(partial, nb, (double)n, out2);
}

// ------------------------------------------------------------------------------------------------
// calc_loss_on_attn_grid -- modules/refinement_functions.py:42-77 (+ autograd):
//   mask = render > 0;  loss = sum(|render - map| * mask) / sum(mask)
//   d loss / d render = ((1 / sum(mask)) * mask) * sign(render - map)      (div, sum, mul, abs backward in that order; sign(0) = 0)
// Two launches: per-block (count, sum) partials in a fixed order, then every block of the gradient pass folds the (few) partials
// itself -- no finalize launch in between.  sum(mask) is an exact integer; the loss sum is carried in double and rounded once.
// (A single block doing both passes measured 69 us at 266x266: 70 dependent round trips per thread.)
// ------------------------------------------------------------------------------------------------
constexpr int kAttnL1Blocks = 128;
__global__ __launch_bounds__(256) void attn_masked_l1_partial_kernel(const float* __restrict__ render, const float* __restrict__ map,
                                                                      long long n, double* __restrict__ partial) {
  double s[2] = {0.0, 0.0};     // sum |render - map| over the mask, count of the mask
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float r = render[i];
    if (r > 0.0f) { s[0] += (double)fabsf(r - map[i]); s[1] += 1.0; }
  }
  block_sum<2>(s, partial + (long long)blockIdx.x * 2);
}
__global__ __launch_bounds__(256) void attn_masked_l1_grad_kernel(const float* __restrict__ render, const float* __restrict__ map,
                                                                   long long n, const double* __restrict__ partial, int nblocks,
                                                                   float* __restrict__ d_render, float* __restrict__ loss_out) {
  __shared__ float sm_inv;
  if (threadIdx.x < 64) {       // one wave folds the partials (nblocks <= 128), the same order in every block
    double ts = 0.0, tc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 64) { ts += partial[2 * i]; tc += partial[2 * i + 1]; }
    ts = wave_sum(ts);
    tc = wave_sum(tc);
    if (threadIdx.x == 0) {
      const float msum = (float)tc;                                        // mask.sum(): float32, exact below 2^24 pixels
      if (loss_out && blockIdx.x == 0) *loss_out = (float)ts / msum;        // diff_masked.sum() / mask.sum()  (0 / 0 = NaN, like the reference)
      sm_inv = 1.0f / msum;
    }
  }
  __syncthreads();
  const float inv = sm_inv;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float r = render[i];
    const float m = r > 0.0f ? 1.0f : 0.0f;
    d_render[i] = (inv * m) * sgnf(r - map[i]);
  }
}
size_t attn_l1_scratch_bytes() { return sizeof(double) * 2 * kAttnL1Blocks; }
void launch_attn_masked_l1(const float* render, const float* map, long long n, float* d_render, float* loss_out, void* scratch,
                           hipStream_t st) {
  if (n <= 0) return;
  const int nb = (int)((n + 255) / 256 < kAttnL1Blocks ? (n + 255) / 256 : kAttnL1Blocks);
  double* partial = (double*)scratch;
  attn_masked_l1_partial_kernel<<<nb, 256, 0, st>>>(render, map, n, partial);
  const int ng = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  attn_masked_l1_grad_kernel<<<ng, 256, 0, st>>>(render, map, n, partial, nb, d_render, loss_out);
}

// ------------------------------------------------------------------------------------------------
// Sustained shader clock (measurement aid of bench.py: the issue-rate ceilings of the render kernels are quoted in shader
// clocks, so the clock has to be MEASURED under load, not assumed).  s_memtime counts shader clocks, s_memrealtime a
// constant reference clock (hipDeviceAttributeWallClockRate, 100 MHz); every block of a chip-filling launch that keeps
// the VALU and LDS pipes busy (packed FMAs + LDS double adds: the mix of the render backward) for ~`spin` iterations
// reports both deltas; the host sums them (per-block quantisation of the 100 MHz counter averages out).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long* __restrict__ out, int spin) {
  __shared__ double acc[256];
  acc[threadIdx.x] = 0.0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f x[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = v2f{1.0f + 1e-3f * threadIdx.x, 0.5f + i};
  for (int it = 0; it < spin; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = __builtin_elementwise_fma(x[i], v2f{1.0001f, 0.9999f}, v2f{1e-3f, 1e-4f});
    if ((it & 7) == 0)
      __hip_atomic_fetch_add(&acc[(threadIdx.x * 5 + it) & 255], (double)x[0].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += x[i].x + x[i].y;
  __syncthreads();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = t1 - t0;
    out[2 * blockIdx.x + 1] = r1 - r0;
    if (s == -1.0f && acc[0] == -1.0) out[0] = 0;   // (keeps the arithmetic alive)
  }
}

// blocking: launches on `st`, waits, returns the clock in Hz (0 on failure).  ~0.3 ms of device time with the default spin.
double run_clock_probe(int spin, hipStream_t st) {
  static unsigned long long* dbuf = nullptr;
  constexpr int kBlocks = 256 * 4;   // 4 blocks of 4 waves per CU: every SIMD holds 4 waves
  if (!dbuf && hipMalloc(&dbuf, kBlocks * 2 * sizeof(unsigned long long)) != hipSuccess) return 0.0;
  clock_probe_kernel<<<kBlocks, 256, 0, st>>>(dbuf, spin > 0 ? spin : 4000);
  static unsigned long long host[kBlocks * 2];
  if (hipMemcpyAsync(host, dbuf, sizeof(host), hipMemcpyDeviceToHost, st) != hipSuccess) return 0.0;
  if (hipStreamSynchronize(st) != hipSuccess) return 0.0;
  double shader = 0.0, wall = 0.0;
  for (int i = 0; i < kBlocks; ++i) { shader += (double)host[2 * i]; wall += (double)host[2 * i + 1]; }
  int dev = 0, wall_khz = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0.0;
  if (hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || wall_khz <= 0) return 0.0;
  return wall > 0.0 ? shader / wall * (double)wall_khz * 1e3 : 0.0;
}

}  // namespace voxe
