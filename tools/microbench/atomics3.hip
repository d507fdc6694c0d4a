// Microbenchmark 3: LDS atomic rates by type (f32 / u32 / u64 / f64), with duplication and with a
// reduced number of active lanes; plain LDS read-modify-write for reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename T, int DUP, int ACTIVE>
__global__ __launch_bounds__(64) void lds_atomic(T* out, int iters) {
  __shared__ T s[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) s[i] = T(0);
  __syncthreads();
  const unsigned lane = threadIdx.x & 63;
  if (lane < ACTIVE) {
    for (int it = 0; it < iters; it += 4) {
      const unsigned slot = ((lane / DUP) + it * 5) & 511;
#pragma unroll
      for (int ch = 0; ch < 4; ++ch)
        __hip_atomic_fetch_add(&s[ch * 512 + slot], T(1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  T acc = 0; for (int i = threadIdx.x; i < 2048; i += 64) acc += s[i];
  if (acc == T(-1)) out[0] = acc;
}

template <int DUP>
__global__ __launch_bounds__(64) void lds_rmw(float* out, int iters) {  // NOT race free; rate reference
  __shared__ float s[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) s[i] = 0.f;
  __syncthreads();
  const unsigned lane = threadIdx.x & 63;
  for (int it = 0; it < iters; it += 4) {
    const unsigned slot = ((lane / DUP) + it * 5) & 511;
    float v[4];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) v[ch] = s[ch * 512 + slot];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) s[ch * 512 + slot] = v[ch] + 1.0f;
  }
  __syncthreads();
  float acc = 0; for (int i = threadIdx.x; i < 2048; i += 64) acc += s[i];
  if (acc == -1.f) out[0] = acc;
}

__global__ __launch_bounds__(64) void glob_u64(unsigned long long* buf, unsigned mask, int iters, int pat) {
  const unsigned gtid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, wave = gtid >> 6;
  for (int it = 0; it < iters; ++it) {
    unsigned h = wave * 977u + it; h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16;
    unsigned idx = pat == 0 ? h * 64u + lane : (h * 131u + lane * 7919u);
    atomicAdd(&buf[idx & mask], 1ull);
  }
}

template <typename F> float time_ms(F f, int reps = 3) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
  return best;
}

int main() {
  void* o; CK(hipMalloc(&o, 64));
  const int blocks = 256 * 16, li = 4096;
#define RUNA(T, D, A) { float ms = time_ms([&] { lds_atomic<T, D, A><<<blocks, 64>>>((T*)o, li); }); \
    double rate = (double)blocks * A * li / ms * 1e-6; \
    printf("LDS atomic add %-18s dup=%-2d active=%2d: %8.3f ms %9.2f G lane-ops/s, %6.1f clk per wave-instr per CU\n", #T, D, A, ms, rate, (double)ms * 1e-3 * 2.4e9 / ((double)blocks / 256 * li)); }
  RUNA(float, 1, 64) RUNA(float, 4, 64) RUNA(float, 1, 16) RUNA(float, 1, 32)
  RUNA(unsigned, 1, 64) RUNA(unsigned, 4, 64) RUNA(unsigned, 16, 64)
  RUNA(unsigned long long, 1, 64) RUNA(unsigned long long, 4, 64) RUNA(unsigned long long, 16, 64)
  RUNA(double, 1, 64) RUNA(double, 4, 64)
  RUNA(int, 1, 64)
#define RUNR(D) { float ms = time_ms([&] { lds_rmw<D><<<blocks, 64>>>((float*)o, li); }); \
    printf("LDS plain read+add+write dup=%-2d: %8.3f ms, %6.1f clk per (4 reads + 4 writes) per CU\n", D, ms, (double)ms * 1e-3 * 2.4e9 / ((double)blocks / 256 * li / 4)); }
  RUNR(1) RUNR(4)
  const size_t n = 1u << 23; unsigned long long* gb; CK(hipMalloc(&gb, n * 8)); CK(hipMemset(gb, 0, n * 8));
  for (int pat = 0; pat < 2; ++pat) { const int gbk = 4096, gi = 128; float ms = time_ms([&] { glob_u64<<<gbk, 64>>>(gb, (unsigned)(n - 1), gi, pat); });
    printf("global atomicAdd u64 %s: %8.3f ms %8.2f G lane-atomics/s\n", pat == 0 ? "distinct(8 lines)" : "random", ms, (double)gbk * 64 * gi / ms * 1e-6); }
  return 0;
}
