// Third microbenchmark: same-register operands (src0 == src1) vs distinct registers, fused DPP vs plain, asm block vs statements.
// 12 FP64 accumulate operations per loop iteration, three accumulators, lone wavefront per SIMD (1024 x 64 threads).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define FD(A, X, Y) "v_fmac_f64_dpp %" #A ", %" #X ", %" #Y " row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
#define PF(A, X, Y) "v_fmac_f64 %" #A ", %" #X ", %" #Y "\n\t"
#define INS "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]), "v"(y)
template <int MODE> __global__ void __launch_bounds__(64) k(double* out, int iters, double a) {
  const int lane = threadIdx.x;
  double x[12], c0 = lane, c1 = lane + 1, c2 = lane + 2, y = a * 0.5 + lane * 1e-4;
#pragma unroll
  for (int i = 0; i < 12; i++) x[i] = a + 1e-3 * (lane + i);
  asm volatile("s_nop 4" ::: "memory");
  for (int it = 0; it < iters; it++) {
    if (MODE == 0)        // fused, block, same register twice (the kernel's dot_bc form)
      asm volatile(FD(0, 3, 3) FD(1, 4, 4) FD(2, 5, 5) FD(0, 6, 6) FD(1, 7, 7) FD(2, 8, 8) FD(0, 9, 9) FD(1, 10, 10) FD(2, 11, 11) FD(0, 12, 12) FD(1, 13, 13) FD(2, 14, 14)
                   : "+v"(c0), "+v"(c1), "+v"(c2) : INS);
    else if (MODE == 1)   // fused, block, distinct registers (x[i] broadcast times a common y: the update form)
      asm volatile(FD(0, 3, 15) FD(1, 4, 15) FD(2, 5, 15) FD(0, 6, 15) FD(1, 7, 15) FD(2, 8, 15) FD(0, 9, 15) FD(1, 10, 15) FD(2, 11, 15) FD(0, 12, 15) FD(1, 13, 15) FD(2, 14, 15)
                   : "+v"(c0), "+v"(c1), "+v"(c2) : INS);
    else if (MODE == 2)   // fused, block, distinct registers x[i] * x[i+1]
      asm volatile(FD(0, 3, 4) FD(1, 4, 5) FD(2, 5, 6) FD(0, 6, 7) FD(1, 7, 8) FD(2, 8, 9) FD(0, 9, 10) FD(1, 10, 11) FD(2, 11, 12) FD(0, 12, 13) FD(1, 13, 14) FD(2, 14, 3)
                   : "+v"(c0), "+v"(c1), "+v"(c2) : INS);
    else if (MODE == 3)   // plain, block, same register twice
      asm volatile(PF(0, 3, 3) PF(1, 4, 4) PF(2, 5, 5) PF(0, 6, 6) PF(1, 7, 7) PF(2, 8, 8) PF(0, 9, 9) PF(1, 10, 10) PF(2, 11, 11) PF(0, 12, 12) PF(1, 13, 13) PF(2, 14, 14)
                   : "+v"(c0), "+v"(c1), "+v"(c2) : INS);
    else if (MODE == 4)   // plain, block, distinct registers
      asm volatile(PF(0, 3, 4) PF(1, 4, 5) PF(2, 5, 6) PF(0, 6, 7) PF(1, 7, 8) PF(2, 8, 9) PF(0, 9, 10) PF(1, 10, 11) PF(2, 11, 12) PF(0, 12, 13) PF(1, 13, 14) PF(2, 14, 3)
                   : "+v"(c0), "+v"(c1), "+v"(c2) : INS);
    else if (MODE == 5) { // plain, compiler-generated, distinct registers
#pragma unroll
      for (int i = 0; i < 12; i += 3) { c0 = fma(x[i], x[(i + 1) % 12], c0); c1 = fma(x[i + 1], x[(i + 2) % 12], c1); c2 = fma(x[i + 2], x[(i + 3) % 12], c2); }
    } else if (MODE == 6) { // plain, compiler-generated, same register twice
#pragma unroll
      for (int i = 0; i < 12; i += 3) { c0 = fma(x[i], x[i], c0); c1 = fma(x[i + 1], x[i + 1], c1); c2 = fma(x[i + 2], x[i + 2], c2); }
    } else {              // fused, 12 statements, distinct registers
#pragma unroll
      for (int i = 0; i < 12; i += 3) {
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(c0) : "v"(x[i]), "v"(y));
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(c1) : "v"(x[i + 1]), "v"(y));
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(c2) : "v"(x[i + 2]), "v"(y));
      }
    }
  }
  out[blockIdx.x * 64 + lane] = c0 + c1 + c2;
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
  const char* names[8] = {"fused, block, src0 == src1", "fused, block, x[i] * y", "fused, block, x[i] * x[i+1]", "plain, block, src0 == src1",
                          "plain, block, x[i] * x[i+1]", "plain, compiler, x[i] * x[i+1]", "plain, compiler, x[i] * x[i]", "fused, 12 statements, x[i] * y"};
  for (int blocks : {1024, 2048}) {
    float ms[8] = {run<0>(blocks, iters, d), run<1>(blocks, iters, d), run<2>(blocks, iters, d), run<3>(blocks, iters, d),
                   run<4>(blocks, iters, d), run<5>(blocks, iters, d), run<6>(blocks, iters, d), run<7>(blocks, iters, d)};
    for (int m = 0; m < 8; m++)
      printf("waves/SIMD %d  %-36s %.3f ms -> %.2f cycles per op (2.4 GHz)\n", blocks / 1024, names[m], ms[m], ms[m] * 1e-3 * 2.4e9 / (iters * 12.0) / (blocks / 1024.0));
  }
  return 0;
}
