// Microbenchmark (diagnostic): what acc += bcast16(x, lane) * y costs on gfx950 for a lone wavefront per SIMD, in the
// forms the 16-lane tick kernel could use:
//   0  plain v_fma_f64, no broadcast (the issue floor of an FP64 chain; 6 independent accumulators)
//   1  compiler form: v_mov_b64_dpp row_newbcast + v_fma_f64 (what `acc += dpp(x) * y` compiles to: the mov is never folded)
//   2  fused v_fmac_f64_dpp, ONE inline-asm statement per op (the hazard recogniser adds wait states between them)
//   3  fused v_fmac_f64_dpp, blocks of 12 ops in one asm statement, three accumulators (the kernel's dot_bc / rows3_bc form)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/dpp_fmac tools/micro/dpp_fmac.hip && /tmp/dpp_fmac
// Cycle counts assume 2.4 GHz (the clock under load is somewhat lower: the true counts are a few per cent smaller);
// s_memtime-free: HIP events around a long loop, 1024 workgroups of 64 threads = one wavefront per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ double mov_bc(double x) {   // the compiler's own v_mov_b64_dpp row_newbcast:5 (as HexDev::dpp in wbc_kernels.hip)
  return __builtin_amdgcn_update_dpp(0.0, x, 0x155, 0xF, 0xF, true);
}

template <int MODE> __global__ void __launch_bounds__(64) k(double* out, int iters, double a) {
  const int lane = threadIdx.x;
  double x[12], acc0 = lane, acc1 = lane + 1, acc2 = lane + 2, acc3 = lane + 3, acc4 = lane + 4, acc5 = lane + 5;
#pragma unroll
  for (int i = 0; i < 12; i++) x[i] = a + 1e-3 * (lane + i);
  asm volatile("s_nop 4" ::: "memory");
#pragma unroll 4
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 12; i += 6) {
        acc0 = fma(x[i], x[i + 1], acc0); acc1 = fma(x[i + 1], x[i + 2], acc1); acc2 = fma(x[i + 2], x[i + 3], acc2);
        acc3 = fma(x[i + 3], x[i + 4], acc3); acc4 = fma(x[i + 4], x[i + 5], acc4); acc5 = fma(x[i + 5], x[i], acc5);
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 12; i += 3) {
        acc0 = fma(mov_bc(x[i]), x[i], acc0); acc1 = fma(mov_bc(x[i + 1]), x[i + 1], acc1); acc2 = fma(mov_bc(x[i + 2]), x[i + 2], acc2);
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 12; i += 3) {
        asm volatile("v_fmac_f64_dpp %0, %1, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc0) : "v"(x[i]));
        asm volatile("v_fmac_f64_dpp %0, %1, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc1) : "v"(x[i + 1]));
        asm volatile("v_fmac_f64_dpp %0, %1, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc2) : "v"(x[i + 2]));
      }
    } else {
#define FD(A, X) "v_fmac_f64_dpp %" #A ", %" #X ", %" #X " row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
      asm volatile(FD(0, 3) FD(1, 4) FD(2, 5) FD(0, 6) FD(1, 7) FD(2, 8) FD(0, 9) FD(1, 10) FD(2, 11) FD(0, 12) FD(1, 13) FD(2, 14)
                   : "+v"(acc0), "+v"(acc1), "+v"(acc2)
                   : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]));
    }
  }
  out[blockIdx.x * 64 + lane] = acc0 + acc1 + acc2 + acc3 + acc4 + acc5;
}

template <int MODE> float run(int blocks, int iters, double* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 0.999);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 0.999);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  double* d; hipMalloc(&d, 4096 * 64 * 8);
  const int iters = 20000;   // 12 accumulate operations per iteration
  const char* names[4] = {"plain v_fma_f64 (no broadcast)", "v_mov_b64_dpp + v_fma_f64 (compiler form)", "v_fmac_f64_dpp, one asm per op", "v_fmac_f64_dpp, 12 ops per asm block"};
  for (int blocks : {1024, 2048}) {
    float ms[4] = {run<0>(blocks, iters, d), run<1>(blocks, iters, d), run<2>(blocks, iters, d), run<3>(blocks, iters, d)};
    for (int m = 0; m < 4; m++)
      printf("waves/SIMD %d  %-44s %.3f ms  -> %.2f cycles per accumulate (2.4 GHz)\n", blocks / 1024, names[m], ms[m],
             ms[m] * 1e-3 * 2.4e9 / (iters * 12.0) / (blocks / 1024.0));
  }
  return 0;
}
