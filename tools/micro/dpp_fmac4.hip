// Fourth microbenchmark: is the cost of an inline-asm block of FP64 ops a per-BLOCK matter or a per-LOOP-ITERATION one?
// R blocks of 12 fused ops per loop iteration, R = 1, 2, 4, 8 (same total work), three accumulators, lone wavefront per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define FD(A, X, Y) "v_fmac_f64_dpp %" #A ", %" #X ", %" #Y " row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
#define INS "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11])
template <int R, bool STMT> __global__ void __launch_bounds__(64) k(double* out, int iters, double a) {
  const int lane = threadIdx.x;
  double x[12], c0 = lane, c1 = lane + 1, c2 = lane + 2;
#pragma unroll
  for (int i = 0; i < 12; i++) x[i] = a + 1e-3 * (lane + i);
  asm volatile("s_nop 4" ::: "memory");
  for (int it = 0; it < iters / R; it++) {
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (!STMT)
        asm volatile(FD(0, 3, 3) FD(1, 4, 4) FD(2, 5, 5) FD(0, 6, 6) FD(1, 7, 7) FD(2, 8, 8) FD(0, 9, 9) FD(1, 10, 10) FD(2, 11, 11) FD(0, 12, 12) FD(1, 13, 13) FD(2, 14, 14)
                     : "+v"(c0), "+v"(c1), "+v"(c2) : INS);
      else {
#pragma unroll
        for (int i = 0; i < 12; i += 3) {
          asm volatile("v_fmac_f64_dpp %0, %1, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(c0) : "v"(x[i]));
          asm volatile("v_fmac_f64_dpp %0, %1, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(c1) : "v"(x[i + 1]));
          asm volatile("v_fmac_f64_dpp %0, %1, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(c2) : "v"(x[i + 2]));
        }
      }
    }
  }
  out[blockIdx.x * 64 + lane] = c0 + c1 + c2;
}
template <int R, bool STMT> float run(int iters, double* d) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; w++) hipLaunchKernelGGL((k<R, STMT>), dim3(1024), dim3(64), 0, 0, d, iters, 0.999);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<R, STMT>), dim3(1024), dim3(64), 0, 0, d, iters, 0.999);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  double* d; (void)hipMalloc(&d, 4096 * 64 * 8);
  const int iters = 16384;
  float b[4] = {run<1, false>(iters, d), run<2, false>(iters, d), run<4, false>(iters, d), run<8, false>(iters, d)};
  float s[4] = {run<1, true>(iters, d), run<2, true>(iters, d), run<4, true>(iters, d), run<8, true>(iters, d)};
  const int R[4] = {1, 2, 4, 8};
  for (int i = 0; i < 4; i++)
    printf("%d x 12 fused ops per loop iteration: asm blocks %.2f cycles per op, single-op statements %.2f (2.4 GHz)\n", R[i],
           b[i] * 1e-3 * 2.4e9 / (iters * 12.0), s[i] * 1e-3 * 2.4e9 / (iters * 12.0));
  return 0;
}
