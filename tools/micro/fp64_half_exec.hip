// Microbenchmark (diagnostic): does a wave64 FP64 FMA with only 32 active lanes issue faster on gfx950,
// and how do two such waves share one SIMD?   hipcc --offload-arch=gfx950 -O3 -o /tmp/fp64_half tools/micro/fp64_half_exec.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE> __global__ void __launch_bounds__(64) k(double* out, int iters, double a) {
  const int lane = threadIdx.x;
  double x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3, x4 = lane + 4, x5 = lane + 5, x6 = lane + 6, x7 = lane + 7;
  if (MODE == 1 && lane >= 32) { out[blockIdx.x * 64 + lane] = 0; return; }   // half of the wave idles
  if (MODE == 2 && lane >= 16) { out[blockIdx.x * 64 + lane] = 0; return; }
#pragma unroll 32
  for (int i = 0; i < iters; i++) {
    x0 = fma(x0, a, 1.0); x1 = fma(x1, a, 1.0); x2 = fma(x2, a, 1.0); x3 = fma(x3, a, 1.0);
    x4 = fma(x4, a, 1.0); x5 = fma(x5, a, 1.0); x6 = fma(x6, a, 1.0); x7 = fma(x7, a, 1.0);
  }
  out[blockIdx.x * 64 + lane] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int MODE> float run(int blocks, int iters, double* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 0.999);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 0.999);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  double* d; hipMalloc(&d, 8192 * 64 * 8);
  const int iters = 20000;   // 160k FMAs per lane
  for (int blocks : {1024, 2048, 4096}) {
    float f = run<0>(blocks, iters, d), h = run<1>(blocks, iters, d), q = run<2>(blocks, iters, d);
    printf("blocks %4d (waves/SIMD %.0f): full exec %.3f ms, 32 lanes %.3f ms, 16 lanes %.3f ms  -> cycles/FMA at 2.4 GHz: %.2f / %.2f / %.2f\n", blocks, blocks / 1024.0,
           f, h, q, f * 1e-3 * 2.4e9 / (iters * 8.0) / (blocks / 1024.0), h * 1e-3 * 2.4e9 / (iters * 8.0) / (blocks / 1024.0), q * 1e-3 * 2.4e9 / (iters * 8.0) / (blocks / 1024.0));
  }
  return 0;
}
