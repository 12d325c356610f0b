// Microbenchmark (diagnostic): what rocprofv3's FETCH_SIZE reports for THIS kernel's access shapes on gfx950.
// MI355X_MICROARCH.md: FETCH_SIZE reports exactly half the bytes of a wide (16 B/lane) coalesced streaming read; other widths are
// uncalibrated.  The tick kernel reads 8 B per lane: (a) the model replica, 64 consecutive doubles per wavefront instruction;
// (b) the inputs, 16 row segments of 4 consecutive doubles (32 B) per instruction, rows `ld` doubles apart.
//   hipcc --offload-arch=gfx950 -O3 -o build_variants/fetch_calib tools/micro/fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fc -o run --output-format csv -- build_variants/fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(64) read_coalesced8(const double* a, double* out, size_t n) {   // 8 B/lane, 512 B per instruction
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < n; i += (size_t)gridDim.x * 64) s += a[i];
  if (s == 1.2345e300) out[0] = s;
}
__global__ void __launch_bounds__(64) read_segments32(const double* a, double* out, int ld, int nrows) {   // the input staging shape
  // block b reads robots 4b..4b+3 of every row: lane t -> row (t >> 2) + 16 j, robot 4b + (t & 3)
  double s = 0.0;
  for (int j = 0; j * 16 < nrows; j++) {
    const int row = j * 16 + (threadIdx.x >> 2);
    if (row < nrows) s += a[(size_t)row * ld + (size_t)blockIdx.x * 4 + (threadIdx.x & 3)];
  }
  if (s == 1.2345e300) out[0] = s;
}
__global__ void __launch_bounds__(64) read_dwordx4(const double2* a, double* out, size_t n) {   // 16 B/lane reference shape
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < n; i += (size_t)gridDim.x * 64) { double2 v = a[i]; s += v.x + v.y; }
  if (s == 1.2345e300) out[0] = s;
}
int main() {
  const size_t bytes = 512ull << 20;   // 512 MiB: beyond L2 and the Infinity Cache
  double *a, *out;
  hipMalloc(&a, bytes); hipMalloc(&out, 8); hipMemset(a, 0, bytes);
  const size_t n = bytes / 8;
  hipLaunchKernelGGL(read_coalesced8, dim3(4096), dim3(64), 0, 0, a, out, n);
  hipLaunchKernelGGL(read_dwordx4, dim3(4096), dim3(64), 0, 0, (const double2*)a, out, n / 2);
  const int ld = 1 << 20, nrows = 64;   // 64 rows x 1 Mi robots x 8 B = 512 MiB, every byte read once, in 32-byte segments
  hipLaunchKernelGGL(read_segments32, dim3(ld / 4), dim3(64), 0, 0, a, out, ld, nrows);
  hipDeviceSynchronize();
  printf("each kernel reads %zu bytes exactly once\n", bytes);
  return 0;
}
