// Microbenchmark (diagnostic): a 50/50 mix of v_fma_f64 and cheap 32-bit VALU (v_xor / v_add_u32) per wavefront,
// at 1, 2 and 4 wavefronts per SIMD: do the cheap instructions hide under the FP64 pipe of the other wavefront?
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MIX> __global__ void __launch_bounds__(64) k(double* out, int iters, double a) {
  const int lane = threadIdx.x;
  double x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
  unsigned u0 = lane, u1 = lane * 3, u2 = lane * 5, u3 = lane * 7;
  for (int i = 0; i < iters; i++) {
    x0 = fma(x0, a, 1.0); if (MIX) { u0 = (u0 ^ u1) + 0x9e3779b9u; }
    x1 = fma(x1, a, 1.0); if (MIX) { u1 = (u1 ^ u2) + 0x7f4a7c15u; }
    x2 = fma(x2, a, 1.0); if (MIX) { u2 = (u2 ^ u3) + 0x85ebca6bu; }
    x3 = fma(x3, a, 1.0); if (MIX) { u3 = (u3 ^ u0) + 0xc2b2ae35u; }
  }
  out[blockIdx.x * 64 + lane] = x0 + x1 + x2 + x3 + (double)(u0 ^ u1 ^ u2 ^ u3);
}
template <int MIX> float run(int blocks, int iters, double* d) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MIX>, dim3(blocks), dim3(64), 0, 0, d, iters, 0.999);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MIX>, dim3(blocks), dim3(64), 0, 0, d, iters, 0.999);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  double* d; (void)hipMalloc(&d, 8192 * 64 * 8);
  const int iters = 40000;   // 160k FMAs (+ 320k cheap VALU in the mixed kernel) per lane
  for (int blocks : {1024, 2048, 4096}) {
    float f = run<0>(blocks, iters, d), m = run<1>(blocks, iters, d);
    printf("waves/SIMD %d: FP64 only %.3f ms, FP64 + 2 cheap VALU each %.3f ms  (per wave-level FMA at 2.4 GHz: %.2f / %.2f cycles)\n", blocks / 1024, f, m,
           f * 1e-3 * 2.4e9 / (iters * 4.0) / (blocks / 1024.0), m * 1e-3 * 2.4e9 / (iters * 4.0) / (blocks / 1024.0));
  }
  return 0;
}
