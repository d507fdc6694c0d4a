// Microbenchmark 4: LDS atomic add throughput by operand type, in the access pattern of the tile backward's deposit
// (channel-planar window, slot index varying per lane, optional duplicates = lanes sharing an address).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/atomics4.hip -o tools/microbench/atomics4 && ./atomics4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename T, int DUP, int ACTIVE>
__global__ __launch_bounds__(64) void lds_kernel(float* out, int iters) {
  __shared__ T s[4 * 390];
  for (int i = threadIdx.x; i < 4 * 390; i += 64) s[i] = T(0);
  __syncthreads();
  const unsigned lane = threadIdx.x & 63;
  if (lane < ACTIVE) {
    for (int it = 0; it < iters; it += 4) {
      const unsigned slot = ((lane / DUP) * 3 + it * 5) % 384;
#pragma unroll
      for (int ch = 0; ch < 4; ++ch)
        __hip_atomic_fetch_add(&s[((ch + lane) & 3) * 390 + slot], T(1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  double acc = 0;
  for (int i = threadIdx.x; i < 4 * 390; i += 64) acc += (double)s[i];
  if (acc == -1.0) out[0] = (float)acc;
}

template <typename F> float time_ms(F f, int reps = 3) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
  return best;
}

int main() {
  float* o; CK(hipMalloc(&o, 4));
  const int blocks = 256 * 12, li = 8192;  // 12 one-wave blocks per CU, like the tile backward
#define RUNL(T, D, A) { float ms = time_ms([&] { lds_kernel<T, D, A><<<blocks, 64>>>(o, li); }); \
    double instr = (double)blocks * li / ms * 1e-6; /* G wave-instr/s */ \
    printf("ds_add %-18s dup=%d active=%2d: %8.3f ms  %6.2f clk per wave-instr per CU\n", #T, D, A, ms, 2.4 / (instr / 256)); }
  RUNL(double, 1, 64) RUNL(double, 2, 64) RUNL(double, 4, 64) RUNL(double, 1, 32) RUNL(double, 1, 16)
  RUNL(unsigned long long, 1, 64) RUNL(unsigned long long, 2, 64) RUNL(unsigned long long, 4, 64) RUNL(unsigned long long, 1, 16)
  RUNL(unsigned int, 1, 64) RUNL(unsigned int, 4, 64)
  RUNL(float, 1, 64) RUNL(float, 4, 64) RUNL(float, 1, 16)
  return 0;
}
