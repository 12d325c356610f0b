// Follow-up microbenchmark: what limits back-to-back v_fmac_f64_dpp?  12 fused ops per loop iteration, lone wavefront per SIMD.
//   0  one asm block, 3 accumulators (distance 3)         1  one asm block, 6 accumulators (distance 6)
//   2  one asm block, 12 accumulators (independent)       3  one asm block, 3 accumulators, s_nop 0 after every op
//   4  one asm block, 3 accumulators, s_nop 1 after every op
//   5  one asm block, 3 accumulators, a plain v_fmac_f64 (independent) after every fused op
//   6  12 separate asm statements, 3 accumulators         7  plain v_fmac_f64 only, 3 accumulators (distance 3)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define FD(A, X) "v_fmac_f64_dpp %" #A ", %" #X ", %" #X " row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
#define N0 "s_nop 0\n\t"
#define N1 "s_nop 1\n\t"
template <int MODE> __global__ void __launch_bounds__(64) k(double* out, int iters, double a) {
  const int lane = threadIdx.x;
  double x[12], c[12];
#pragma unroll
  for (int i = 0; i < 12; i++) { x[i] = a + 1e-3 * (lane + i); c[i] = lane + i; }
  double p0 = 1.0, p1 = 2.0, p2 = 3.0;
  asm volatile("s_nop 4" ::: "memory");
  for (int it = 0; it < iters; it++) {
    if (MODE == 0)
      asm volatile(FD(0, 3) FD(1, 4) FD(2, 5) FD(0, 6) FD(1, 7) FD(2, 8) FD(0, 9) FD(1, 10) FD(2, 11) FD(0, 12) FD(1, 13) FD(2, 14)
                   : "+v"(c[0]), "+v"(c[1]), "+v"(c[2])
                   : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]));
    else if (MODE == 1)
      asm volatile(FD(0, 6) FD(1, 7) FD(2, 8) FD(3, 9) FD(4, 10) FD(5, 11) FD(0, 12) FD(1, 13) FD(2, 14) FD(3, 15) FD(4, 16) FD(5, 17)
                   : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5])
                   : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]));
    else if (MODE == 2)
      asm volatile(FD(0, 12) FD(1, 13) FD(2, 14) FD(3, 15) FD(4, 16) FD(5, 17) FD(6, 18) FD(7, 19) FD(8, 20) FD(9, 21) FD(10, 22) FD(11, 23)
                   : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11])
                   : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]));
    else if (MODE == 3)
      asm volatile(FD(0, 3) N0 FD(1, 4) N0 FD(2, 5) N0 FD(0, 6) N0 FD(1, 7) N0 FD(2, 8) N0 FD(0, 9) N0 FD(1, 10) N0 FD(2, 11) N0 FD(0, 12) N0 FD(1, 13) N0 FD(2, 14) N0
                   : "+v"(c[0]), "+v"(c[1]), "+v"(c[2])
                   : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]));
    else if (MODE == 4)
      asm volatile(FD(0, 3) N1 FD(1, 4) N1 FD(2, 5) N1 FD(0, 6) N1 FD(1, 7) N1 FD(2, 8) N1 FD(0, 9) N1 FD(1, 10) N1 FD(2, 11) N1 FD(0, 12) N1 FD(1, 13) N1 FD(2, 14) N1
                   : "+v"(c[0]), "+v"(c[1]), "+v"(c[2])
                   : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]));
    else if (MODE == 5) {
#define PF(P, X) "v_fmac_f64 %" #P ", %" #X ", %" #X "\n\t"
      asm volatile(FD(0, 6) PF(3, 6) FD(1, 7) PF(4, 7) FD(2, 8) PF(5, 8) FD(0, 9) PF(3, 9) FD(1, 10) PF(4, 10) FD(2, 11) PF(5, 11)
                   FD(0, 12) PF(3, 12) FD(1, 13) PF(4, 13) FD(2, 14) PF(5, 14) FD(0, 15) PF(3, 15) FD(1, 16) PF(4, 16) FD(2, 17) PF(5, 17)
                   : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(p0), "+v"(p1), "+v"(p2)
                   : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]));
    } else if (MODE == 6) {
#pragma unroll
      for (int i = 0; i < 12; i += 3) {
        asm volatile("v_fmac_f64_dpp %0, %1, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(c[0]) : "v"(x[i]));
        asm volatile("v_fmac_f64_dpp %0, %1, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(c[1]) : "v"(x[i + 1]));
        asm volatile("v_fmac_f64_dpp %0, %1, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(c[2]) : "v"(x[i + 2]));
      }
    } else {
#define PG(A, X) "v_fmac_f64 %" #A ", %" #X ", %" #X "\n\t"
      asm volatile(PG(0, 3) PG(1, 4) PG(2, 5) PG(0, 6) PG(1, 7) PG(2, 8) PG(0, 9) PG(1, 10) PG(2, 11) PG(0, 12) PG(1, 13) PG(2, 14)
                   : "+v"(c[0]), "+v"(c[1]), "+v"(c[2])
                   : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]));
    }
  }
  double s = p0 + p1 + p2;
#pragma unroll
  for (int i = 0; i < 12; i++) s += c[i];
  out[blockIdx.x * 64 + lane] = s;
}
template <int MODE> float run(int blocks, int iters, double* d) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 0.999);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 0.999);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  double* d; (void)hipMalloc(&d, 4096 * 64 * 8);
  const int iters = 20000;
  const char* names[8] = {"block, 3 accumulators", "block, 6 accumulators", "block, 12 accumulators (independent)", "block, 3 acc, s_nop 0 between",
                          "block, 3 acc, s_nop 1 between", "block, 3 acc, plain v_fmac_f64 between (2 ops per slot)", "12 asm statements, 3 acc", "plain v_fmac_f64 block, 3 acc"};
  for (int blocks : {1024, 2048}) {
    float ms[8] = {run<0>(blocks, iters, d), run<1>(blocks, iters, d), run<2>(blocks, iters, d), run<3>(blocks, iters, d),
                   run<4>(blocks, iters, d), run<5>(blocks, iters, d), run<6>(blocks, iters, d), run<7>(blocks, iters, d)};
    for (int m = 0; m < 8; m++)
      printf("waves/SIMD %d  %-58s %.3f ms -> %.2f cycles per fused op slot (2.4 GHz)\n", blocks / 1024, names[m], ms[m],
             ms[m] * 1e-3 * 2.4e9 / (iters * 12.0) / (blocks / 1024.0));
  }
  return 0;
}
