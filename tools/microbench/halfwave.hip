// Does a wave64 VALU instruction cost less when one 32-lane half of EXEC is empty?  (gfx950: SIMD-32, a wave64 instruction
// issues over two passes.)   hipcc --offload-arch=gfx950 -O3 halfwave.hip -o halfwave && ./halfwave
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(64) void k(float* out, int iters, int mode) {
  const int lane = threadIdx.x;
  float a = lane * 0.001f, b = 1.0001f, c = 0.5f, d = 0.25f;
  bool on = mode == 0 ? true : (mode == 1 ? lane < 32 : (mode == 2 ? (lane & 1) == 0 : (mode == 3 ? lane >= 32 : ((lane >> 2) & 1) == 0)));
  if (on) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) { a = fmaf(a, b, c); c = fmaf(c, b, d); d = fmaf(d, b, a); b = fmaf(b, 0.99999f, 1e-7f); }
    }
  }
  out[(blockIdx.x % 4096) * 64 + lane] = a + b + c + d;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"all 64 lanes", "lanes 0-31", "even lanes", "lanes 32-63", "groups of 4 (split = 1 pattern)"};
  for (int mode = 0; mode < 5; ++mode) {
    k<<<4096, 64>>>(out, 200, mode); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<4096 * 3, 64>>>(out, 2000, mode); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %.3f ms\n", names[mode], ms);
  }
  return 0;
}
