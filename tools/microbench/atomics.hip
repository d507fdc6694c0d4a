// Microbenchmark: float atomic-add throughput on MI355X for the access patterns of the render
// backward (scatter of trilinear gradients).  Build: hipcc --offload-arch=gfx950 -O3 atomics.hip -o atomics
//   scope:   agent (atomicAdd, sc1, performed memory-side) vs workgroup (no sc1, performed in the XCD's L2)
//   pattern: distinct (lane i -> element i), dup4/dup16 (4/16 lanes per element), neigh (64 lanes over
//            a 4x4x2 voxel neighbourhood x 4 channels like an 8x8 pixel tile), random
// Also: ds_add_f32 throughput with the same duplication.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ int xcc_id() { return (int)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); }

enum Pattern { DISTINCT = 0, DUP4, DUP16, NEIGH, RANDOM };

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// n_elems: buffer size in floats (power of two). iters: atomics per thread.
template <int SCOPE /*0 agent, 1 workgroup+perXCD*/, int PAT>
__global__ __launch_bounds__(256) void atom_kernel(float* buf, unsigned mask, int iters, size_t xcd_stride) {
  const unsigned gtid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned lane = threadIdx.x & 63;
  const unsigned wave = gtid >> 6;
  float* b = buf;
  if (SCOPE == 1) b = buf + (size_t)xcc_id() * xcd_stride;
  for (int it = 0; it < iters; ++it) {
    unsigned idx;
    if (PAT == DISTINCT) idx = (wave * 64 + lane + it * 9973u * 64u);
    else if (PAT == DUP4) idx = (wave * 16 + (lane >> 2) + it * 9973u * 16u);
    else if (PAT == DUP16) idx = (wave * 4 + (lane >> 4) + it * 9973u * 4u);
    else if (PAT == NEIGH) {
      // 8x8 pixel tile: voxel (lane&7)/2, (lane>>3)/2, (it&1); 160^3 x 4ch grid; z fastest
      const unsigned vx = (wave * 3u + it / 2u) % 150u + ((lane & 7) >> 1), vy = (wave * 7u) % 150u + ((lane >> 3) >> 1), vz = (it * 1u) % 150u;
      idx = ((vx * 160u + vy) * 160u + vz) * 4u + (it & 3);
    } else idx = hash32(gtid * 131u + it);
    idx &= mask;
    if (SCOPE == 0) atomicAdd(&b[idx], 1.0f);
    else __hip_atomic_fetch_add(&b[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

template <int DUP>
__global__ __launch_bounds__(256) void lds_kernel(float* out, int iters) {
  __shared__ float s[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) s[i] = 0.f;
  __syncthreads();
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int it = 0; it < iters; ++it) {
    unsigned idx = (wave * 2048 + ((lane / DUP) * 4 + (it & 3)) + (it >> 2) * 37) & 8191;
    __hip_atomic_fetch_add(&s[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  float acc = 0; for (int i = threadIdx.x; i < 8192; i += 256) acc += s[i];
  if (acc == -1.f) out[0] = acc;
}

template <typename F> float time_ms(F f, int reps = 3) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
  return best;
}

int main() {
  const size_t n = 1u << 24;  // 16M floats = 64 MB (the 160^3 x 4 grid is 16.4M floats)
  float* buf; CK(hipMalloc(&buf, n * 4 * 8));  // 8 per-XCD copies
  CK(hipMemset(buf, 0, n * 4 * 8));
  const int blocks = 256 * 8, iters = 256;
  const double total = (double)blocks * 256 * iters;
  const char* names[] = {"distinct", "dup4", "dup16", "neigh", "random"};
#define RUN(SC, PAT) { float ms = time_ms([&] { atom_kernel<SC, PAT><<<blocks, 256>>>(buf, (unsigned)(n - 1), iters, n); }); \
    printf("scope=%s pattern=%-8s : %8.3f ms  %8.2f G atomics/s\n", SC ? "wg+perXCD" : "agent    ", names[PAT], ms, total / ms * 1e-6); }
  RUN(0, DISTINCT) RUN(0, DUP4) RUN(0, DUP16) RUN(0, NEIGH) RUN(0, RANDOM)
  RUN(1, DISTINCT) RUN(1, DUP4) RUN(1, DUP16) RUN(1, NEIGH) RUN(1, RANDOM)
  // correctness of the per-XCD scheme: sum over the 8 copies must equal the number of atomics
  CK(hipMemset(buf, 0, n * 4 * 8));
  atom_kernel<1, DUP16><<<blocks, 256>>>(buf, (unsigned)(n - 1), iters, n); CK(hipDeviceSynchronize());
  { std::vector<float> h(n * 8); CK(hipMemcpy(h.data(), buf, n * 4 * 8, hipMemcpyDeviceToHost)); double s = 0; double per[8] = {0};
    for (int x = 0; x < 8; ++x) for (size_t i = 0; i < n; ++i) { s += h[x * n + i]; per[x] += h[x * n + i]; }
    printf("per-XCD wg-scope sum = %.0f expected %.0f (%s)\n  per xcd:", s, total, s == total ? "OK" : "LOST UPDATES");
    for (int x = 0; x < 8; ++x) printf(" %.0f", per[x]); printf("\n"); }
  // what happens if all XCDs share ONE buffer with workgroup scope (expected: lost updates)
  CK(hipMemset(buf, 0, n * 4 * 8));
  atom_kernel<1, DUP16><<<blocks, 256>>>(buf, (unsigned)(n - 1), iters, 0); CK(hipDeviceSynchronize());
  { std::vector<float> h(n); CK(hipMemcpy(h.data(), buf, n * 4, hipMemcpyDeviceToHost)); double s = 0; for (size_t i = 0; i < n; ++i) s += h[i];
    printf("shared buffer wg-scope sum = %.0f expected %.0f (%s)\n", s, total, s == total ? "no loss" : "LOST UPDATES as expected"); }
  float* o; CK(hipMalloc(&o, 4));
#define RUNL(D) { const int li = 4096; float ms = time_ms([&] { lds_kernel<D><<<blocks, 256>>>(o, li); }); \
    printf("LDS ds_add dup=%-2d : %8.3f ms  %8.2f G atomics/s (%.2f per clk per CU @2.4GHz)\n", D, ms, (double)blocks * 256 * li / ms * 1e-6, (double)blocks * 256 * li / ms * 1e-6 / 256 / 2.4); }
  RUNL(1) RUNL(2) RUNL(4) RUNL(16)
  return 0;
}
