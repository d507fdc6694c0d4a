// Microbenchmark 2: (a) is the ~20 G requests/s float-atomic limit per CU or chip-wide? (vary #blocks)
// (b) flush-like patterns: 4 dwords/line x 16 lines per instruction, z-runs; (c) ds_add_f32 rate, stride 1,
// with same-address duplication typical of an 8x8 pixel tile.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

enum { P_RANDOM = 0, P_VOX4, P_ZRUN, P_DISTINCT };
__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int PAT>
__global__ __launch_bounds__(64) void atom_kernel(float* buf, unsigned mask, int iters) {
  const unsigned gtid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, wave = gtid >> 6;
  for (int it = 0; it < iters; ++it) {
    unsigned idx;
    if (PAT == P_RANDOM) idx = hash32(gtid * 131u + it);
    else if (PAT == P_VOX4) {  // 16 voxels (far apart: stride 160*4 floats) x 4 channels
      idx = hash32(wave * 977u + it) + (lane >> 2) * 640u + (lane & 3);
    } else if (PAT == P_ZRUN) {  // 2 runs of 8 consecutive voxels x 4 channels
      idx = (hash32(wave * 977u + it) & ~31u) + (lane >> 5) * 102400u + (lane & 31);
    } else idx = hash32(wave * 977u + it) * 64u + lane;
    atomicAdd(&buf[idx & mask], 1.0f);
  }
}

template <int DUP, bool SOA>
__global__ __launch_bounds__(64) void lds_kernel(float* out, int iters) {
  __shared__ float s[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) s[i] = 0.f;
  __syncthreads();
  const unsigned lane = threadIdx.x & 63;
  for (int it = 0; it < iters; it += 4) {
    const unsigned slot = ((lane / DUP) + it * 5) & 511;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      const unsigned idx = SOA ? ch * 512 + slot : slot * 4 + ch;
      __hip_atomic_fetch_add(&s[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  float acc = 0; for (int i = threadIdx.x; i < 2048; i += 64) acc += s[i];
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
  const size_t n = 1u << 24;
  float* buf; CK(hipMalloc(&buf, n * 4)); CK(hipMemset(buf, 0, n * 4));
  const char* names[] = {"random(64 req)", "vox4 (16 req x4dw)", "zrun (4 lines x16dw)", "distinct(4 lines)"};
  for (int blocks : {256, 1024, 4096, 16384}) {
    const int iters = 128;
    const double total = (double)blocks * 64 * iters;
#define RUN(PAT) { float ms = time_ms([&] { atom_kernel<PAT><<<blocks, 64>>>(buf, (unsigned)(n - 1), iters); }); \
    printf("waves=%5d pattern=%-22s: %8.3f ms %8.2f G lane-atomics/s %7.2f G wave-instr/s\n", blocks, names[PAT], ms, total / ms * 1e-6, total / 64 / ms * 1e-6); }
    RUN(P_RANDOM) RUN(P_VOX4) RUN(P_ZRUN) RUN(P_DISTINCT)
  }
  float* o; CK(hipMalloc(&o, 4));
  const int blocks = 256 * 16, li = 8192;
#define RUNL(D, S) { float ms = time_ms([&] { lds_kernel<D, S><<<blocks, 64>>>(o, li); }); \
    double rate = (double)blocks * 64 * li / ms * 1e-6; \
    printf("LDS ds_add_f32 dup=%-2d %s: %8.3f ms %9.2f G/s = %.2f lanes/clk/CU (%.1f clk per wave-instr)\n", D, S ? "SoA" : "AoS", ms, rate, rate / 256 / 2.4, 64.0 / (rate / 256 / 2.4)); }
  RUNL(1, true) RUNL(2, true) RUNL(4, true) RUNL(8, true) RUNL(1, false) RUNL(4, false)
  return 0;
}
