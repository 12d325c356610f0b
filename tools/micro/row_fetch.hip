// Microbenchmark (diagnostic): the generic trip's "fetch" -- every lane of a 16-lane row needs the 14 doubles (12-slot image, value, norm) held by ONE lane of
// its row, a different lane per row and per trip.  The kernel does it with 28 ds_bpermute_b32 (LDS crossbar, no LDS memory).  How long do those take, and is
// a round trip through LDS memory (the owning lane writes 14 doubles, the row reads them back as 7 x ds_read_b128) any faster?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/row_fetch tools/micro/row_fetch.hip && /tmp/row_fetch
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int NW = 14;
template <int MODE> __global__ void __launch_bounds__(64) k(double* out, unsigned long long* cyc, int iters, int seed) {
  __shared__ __attribute__((aligned(16))) double slot[4][NW + 2];
  const int lane = threadIdx.x, h = lane & 15, row = lane >> 4;
  double x[NW], acc = 0.0;
#pragma unroll
  for (int i = 0; i < NW; i++) x[i] = lane * 0.5 + i;
  int pl = (seed + row * 5) & 15;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    double d[NW];
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < NW; i++) d[i] = __shfl(x[i], (lane & 48) | pl, 64);
    } else {
      if (h == pl) {
#pragma unroll
        for (int i = 0; i < NW; i++) slot[row][i] = x[i];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < NW; i++) d[i] = slot[row][i];
      __builtin_amdgcn_wave_barrier();
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NW; i++) { s += d[i]; x[i] = x[i] * 0.999 + d[i] * 1e-3; }
    acc += s;
    pl = (pl * 5 + 1 + (int)(s > 1e300)) & 15;      // the next trip's lane depends on this trip's data, as in the kernel
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + lane] = acc;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> double run(int blocks, int iters, double* d, unsigned long long* c) {
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, c, iters, 3);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, c, iters, 3);
  hipDeviceSynchronize();
  static unsigned long long hc[8192];
  hipMemcpy(hc, c, blocks * 8, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < blocks; i++) s += hc[i];
  return s / blocks / iters;
}
int main() {
  double* d; unsigned long long* c; hipMalloc(&d, 8192 * 64 * 8); hipMalloc(&c, 8192 * 8);
  const int iters = 2000;
  for (int blocks : {256, 1024, 2048}) {
    const double a = run<0>(blocks, iters, d, c), b = run<1>(blocks, iters, d, c);
    printf("blocks %4d: per trip (incl. 14 adds + 14 FMAs + the lane update): 28 x ds_bpermute_b32 %.0f s_memtime ticks | LDS write + read-back %.0f ticks\n", blocks, a, b);
  }
  return 0;
}
