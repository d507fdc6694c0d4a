// Microbenchmark 5 (r03): what could make the tile backward's deposit cheaper?
//   (a) ds_add_f32 was measured as a 193-clk serial path (atomics.hip / atomics4.hip).  Is that the FP32 denormal mode?
//       Same kernel with MODE.FP_DENORM[1:0] (single precision) set to "flush" by s_setreg at kernel start.
//   (b) masked deposits (the "pending footprint" idea: only the lanes whose ray left a corner add): ds_add_f64 with a
//       RANDOM half / quarter of the lanes active, instead of the contiguous low lanes of atomics4.hip.
//   (c) issue cost of the VALU instructions around the add: v_cvt_f64_f32, v_mul_f64, v_mul_f32, v_exp_f32 -- clk per
//       wave instruction per SIMD (a full-rate f32 instruction is 4).
//   (d) the sustained shader clock: s_memtime (shader clock) against s_memrealtime (constant 100 MHz) inside a kernel.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/microbench/atomics5.hip -o tools/microbench/atomics5
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// s_setreg operand: hwreg(HW_REG_MODE = 1, offset 4, size 2) = MODE.FP_DENORM[1:0], the single-precision denormal mode
// (0 = flush sources and results, 3 = keep; [3:2] is the f64 / f16 mode and stays as the compiler set it)
constexpr int kDenormF32 = 1 | (4 << 6) | ((2 - 1) << 11);
static double g_clk_ghz = 2.4;   // replaced by the measured clock in main()

// MASK: 0 = all lanes, 1 = random half (per-lane hash bit), 2 = random quarter, 3 = lanes < 32, 4 = even lanes
template <typename T, int DUP, int MASK, bool FLUSH_DENORM>
__global__ __launch_bounds__(64) void lds_kernel(float* out, int iters) {
  __shared__ T s[4 * 390];
  if (FLUSH_DENORM) __builtin_amdgcn_s_setreg(kDenormF32, 0);
  for (int i = threadIdx.x; i < 4 * 390; i += 64) s[i] = T(0);
  __syncthreads();
  const unsigned lane = threadIdx.x & 63;
  const unsigned h = (lane * 2654435761u) >> 16;
  bool act = true;
  if (MASK == 1) act = (h & 1) != 0;
  if (MASK == 2) act = (h & 3) == 0;
  if (MASK == 3) act = lane < 32;
  if (MASK == 4) act = (lane & 1) == 0;
  if (act) {
    for (int it = 0; it < iters; it += 4) {
      const unsigned slot = ((lane / DUP) * 3 + it * 5) % 384;
#pragma unroll
      for (int ch = 0; ch < 4; ++ch)
        __hip_atomic_fetch_add(&s[((ch + lane) & 3) * 390 + slot], T(1.5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  double acc = 0;
  for (int i = threadIdx.x; i < 4 * 390; i += 64) acc += (double)s[i];
  if (acc == -1.0) out[0] = (float)acc;
}

// the same pattern but with the deposit's arithmetic in front of every add: value = (float product) -> T
template <typename T, bool FLUSH_DENORM>
__global__ __launch_bounds__(64) void lds_deposit_kernel(float* out, int iters, float g0) {
  __shared__ T s[4 * 390];
  if (FLUSH_DENORM) __builtin_amdgcn_s_setreg(kDenormF32, 0);
  for (int i = threadIdx.x; i < 4 * 390; i += 64) s[i] = T(0);
  __syncthreads();
  const unsigned lane = threadIdx.x & 63;
  float w = 1.0f + 1e-3f * lane;
  for (int it = 0; it < iters; it += 4) {
    const unsigned slot = (lane * 3 + it * 5) % 384;
    w = w * 1.0001f;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch)
      __hip_atomic_fetch_add(&s[((ch + lane) & 3) * 390 + slot], (T)((g0 + ch) * w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  double acc = 0;
  for (int i = threadIdx.x; i < 4 * 390; i += 64) acc += (double)s[i];
  if (acc == -1.0) out[0] = (float)acc;
}

// VALU issue cost: OP 0 = v_mul_f32, 1 = v_cvt_f64_f32 (+ the f32 op that feeds it), 2 = v_mul_f64, 3 = v_exp_f32,
// 4 = v_fma_f64, 5 = v_pk_mul_f32, 6 = v_rcp_f32.  8 independent chains per lane, 4 waves per SIMD.
template <int OP>
__global__ __launch_bounds__(256) void valu_kernel(float* out, int iters, float seed) {
  float x[8];
  double d[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = seed + i + threadIdx.x * 1e-3f; d[i] = x[i]; }
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = v2f{x[i], x[i] + 1.0f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) x[i] = x[i] * 1.0001f;
      if (OP == 1) { d[i] = (double)x[i]; asm volatile("" : "+v"(d[i])); x[i] = __builtin_bit_cast(float, (int)__builtin_bit_cast(long long, d[i]) | 0x3f800000); }
      if (OP == 2) d[i] = d[i] * 1.0001;
      if (OP == 3) x[i] = __builtin_amdgcn_exp2f(x[i]) ;
      if (OP == 4) d[i] = __builtin_fma(d[i], 1.0001, 0.5);
      if (OP == 5) p[i] = p[i] * v2f{1.0001f, 0.9999f};
      if (OP == 6) x[i] = __builtin_amdgcn_rcpf(x[i]);
    }
  }
  float acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += x[i] + (float)d[i] + p[i].x + p[i].y;
  if (acc == -1.2345f) out[0] = acc;
}

__global__ void clock_kernel(unsigned long long* out, int spin) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
  if (x == -1.0f) out[0] = 0;
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
  // ---- (d) clock first: everything below is reported in clocks of the MEASURED shader clock
  {
    unsigned long long* c; CK(hipMalloc(&c, 256 * 16));
    unsigned long long h[512];
    for (int rep = 0; rep < 3; ++rep) {
      clock_kernel<<<256, 64>>>(c, 4000000);
      CK(hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost));
      double st = 0, sr = 0;
      for (int i = 0; i < 256; ++i) { st += (double)h[2 * i]; sr += (double)h[2 * i + 1]; }
      int wall = 0; CK(hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0));
      int smclk = 0; CK(hipDeviceGetAttribute(&smclk, hipDeviceAttributeClockRate, 0));
      printf("clock: s_memtime / s_memrealtime = %.4f  (wall clock rate %d kHz, attribute clock rate %d kHz) -> if s_memtime counts shader clocks: %.3f GHz\n",
             st / sr, wall, smclk, st / sr * wall * 1e-6);
      const double ghz = st / sr * wall * 1e-6;
      if (ghz > 0.8 && ghz < 3.2) g_clk_ghz = ghz;
    }
    // cross-check: a dependent FMA chain of known length (4000000 x 1 v_fma, 4 clk issue each ... latency-bound per wave: ~4-8 clk)
    float ms = time_ms([&] { clock_kernel<<<256, 64>>>(c, 4000000); });
    printf("clock: 4e6 dependent v_fma_f32 in %.3f ms -> %.2f ns per FMA (one wave per SIMD)\n", ms, ms * 1e6 / 4e6);
  }
  const int blocks = 256 * 12, li = 8192;  // 12 one-wave blocks per CU, like the tile backward
#define RUNL(T, D, M, F) { float ms = time_ms([&] { lds_kernel<T, D, M, F><<<blocks, 64>>>(o, li); }); \
    double instr = (double)blocks * li / ms * 1e-6; \
    printf("ds_add %-18s dup=%d mask=%d flush_denorm=%d: %8.3f ms  %6.2f clk per wave-instr per CU\n", #T, D, M, (int)F, ms, g_clk_ghz / (instr / 256)); }
  RUNL(double, 1, 0, false) RUNL(double, 1, 1, false) RUNL(double, 1, 2, false) RUNL(double, 1, 3, false) RUNL(double, 1, 4, false)
  RUNL(double, 2, 0, false) RUNL(double, 2, 1, false) RUNL(double, 4, 0, false) RUNL(double, 4, 1, false)
  RUNL(float, 1, 0, false) RUNL(float, 1, 0, true) RUNL(float, 4, 0, true) RUNL(float, 1, 1, true)
  RUNL(double, 1, 0, true) RUNL(unsigned int, 1, 0, false)
#define RUND(T, F) { float ms = time_ms([&] { lds_deposit_kernel<T, F><<<blocks, 64>>>(o, li, 0.5f); }); \
    double instr = (double)blocks * li / ms * 1e-6; \
    printf("deposit (mul [+cvt] + ds_add) %-8s flush_denorm=%d: %8.3f ms  %6.2f clk per add per CU\n", #T, (int)F, ms, g_clk_ghz / (instr / 256)); }
  RUND(double, false) RUND(float, true) RUND(float, false)
  const int vb = 256 * 4, vi = 20000;      // 4 blocks of 256 threads per CU = 4 waves per SIMD
#define RUNV(OP, NAME) { float ms = time_ms([&] { valu_kernel<OP><<<vb, 256>>>(o, vi, 1.0f); }); \
    double per_simd = (double)vb * 4 * vi * 8 / 1024.0; /* wave instructions per SIMD */ \
    printf("valu %-28s: %8.3f ms  %6.2f clk per wave instruction per SIMD\n", NAME, ms, ms * 1e-3 * g_clk_ghz * 1e9 / per_simd); }
  RUNV(0, "v_mul_f32") RUNV(5, "v_pk_mul_f32") RUNV(1, "v_cvt_f64_f32 + v_or_b32") RUNV(2, "v_mul_f64") RUNV(4, "v_fma_f64")
  RUNV(3, "v_exp_f32") RUNV(6, "v_rcp_f32")
  return 0;
}
