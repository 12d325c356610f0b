// Microbenchmark (diagnostic): north_star's literal mapping -- ONE WAVEFRONT PER ROBOT -- measured on the largest
// distributed phase of the tick, the 30-row Householder append (csrc/wbc_hex.hpp: hex_qr_append; reference: the QP solve of
// controllers/inverse_dynamics_controller.py:223), against the product's 16-lanes-per-robot mapping.
//   (i)  product mapping: 16 lanes (one DPP row) per robot, one COLUMN per lane (12 + right-hand side), 30 rows in registers, 4 robots
//        per wavefront, N / 4 wavefronts (one per SIMD at N = 4096) -- the product's own hex_qr_append, included from csrc/.
//   (ii) 64 lanes per robot: lane = 16 g + c, DPP row g holds rows 8g .. 8g+7 of column c (13 of 16 lanes of a row carry a column),
//        the factor R replicated on the four rows.  A pivot step is 8 fused broadcast-FMAs for the partial dots, an ALL-REDUCE over the
//        four DPP rows (gfx950's v_permlane16_swap / v_permlane32_swap: 2 x (2 swaps + 1 add)), the same scalar chain (norm, reciprocal,
//        scale) and 8 fused updates.  120 VGPRs as compiled: up to 4 wavefronts per SIMD; N wavefronts.
// Both produce R[12][13]; they are checked against each other.  Timed: REPS appends back to back per wavefront from register copies
// (no memory traffic in the loop), N robots, HIP events.    hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/wpr tools/micro/wave_per_robot_qr.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../quadruped_drake_amd/csrc/wbc_hex.hpp"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int P = 30, NC = 13;   // appended rows, columns (12 variables + right-hand side)

// ---- (i) the product's communication policy, the members hex_qr_append uses (csrc/wbc_kernels.hip: HexDev)
struct Hex16 {
  template <int CTRL> static __device__ __forceinline__ double dpp(double x) { return __builtin_amdgcn_update_dpp(0.0, x, CTRL, 0xF, 0xF, true); }
  __device__ __forceinline__ double bcast16(double x, int src) const {
    switch (src) {
      case 0: return dpp<0x150>(x);   case 1: return dpp<0x151>(x);   case 2: return dpp<0x152>(x);   case 3: return dpp<0x153>(x);
      case 4: return dpp<0x154>(x);   case 5: return dpp<0x155>(x);   case 6: return dpp<0x156>(x);   case 7: return dpp<0x157>(x);
      case 8: return dpp<0x158>(x);   case 9: return dpp<0x159>(x);   case 10: return dpp<0x15A>(x);  case 11: return dpp<0x15B>(x);
      case 12: return dpp<0x15C>(x);  case 13: return dpp<0x15D>(x);  case 14: return dpp<0x15E>(x);  default: return dpp<0x15F>(x);
    }
  }
  template <int SRC> __device__ __forceinline__ double fma_bc(double acc, double x, double y) const {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y), "n"(SRC));
    return acc;
  }
#define WBC_FD(A, X) "v_fmac_f64_dpp %" #A ", %" #X ", %" #X " row_newbcast:%[ln] row_mask:0xf bank_mask:0xf\n\t"
  template <int SRC, int N> __device__ __forceinline__ void dot_bc(double& ta, double& tb, double& tc, const double* a) const {
    static_assert(N == 15 || N == 9 || N == 6 || N == 3, "chunk sizes");
    if constexpr (N == 15)
      asm volatile(WBC_FD(0, 3) WBC_FD(1, 4) WBC_FD(2, 5) WBC_FD(0, 6) WBC_FD(1, 7) WBC_FD(2, 8) WBC_FD(0, 9) WBC_FD(1, 10) WBC_FD(2, 11) WBC_FD(0, 12) WBC_FD(1, 13) WBC_FD(2, 14) WBC_FD(0, 15) WBC_FD(1, 16) WBC_FD(2, 17)
                   : "+v"(ta), "+v"(tb), "+v"(tc) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), [ln] "n"(SRC));
    else if constexpr (N == 9)
      asm volatile(WBC_FD(0, 3) WBC_FD(1, 4) WBC_FD(2, 5) WBC_FD(0, 6) WBC_FD(1, 7) WBC_FD(2, 8) WBC_FD(0, 9) WBC_FD(1, 10) WBC_FD(2, 11)
                   : "+v"(ta), "+v"(tb), "+v"(tc) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), [ln] "n"(SRC));
    else if constexpr (N == 6)
      asm volatile(WBC_FD(0, 3) WBC_FD(1, 4) WBC_FD(2, 5) WBC_FD(0, 6) WBC_FD(1, 7) WBC_FD(2, 8)
                   : "+v"(ta), "+v"(tb), "+v"(tc) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), [ln] "n"(SRC));
    else
      asm volatile(WBC_FD(0, 3) WBC_FD(1, 4) WBC_FD(2, 5)
                   : "+v"(ta), "+v"(tb), "+v"(tc) : "v"(a[0]), "v"(a[1]), "v"(a[2]), [ln] "n"(SRC));
  }
  template <int N> static __device__ __forceinline__ void dpp_fence(double* a) {
    if constexpr (N >= 6) {
      asm volatile("s_nop 4" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]));
      dpp_fence<N - 6>(a + 6);
    } else if constexpr (N == 5) { asm volatile("s_nop 4" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4])); }
    else if constexpr (N == 4) { asm volatile("s_nop 4" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])); }
    else if constexpr (N == 3) { asm volatile("s_nop 4" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2])); }
    else if constexpr (N == 2) { asm volatile("s_nop 4" : "+v"(a[0]), "+v"(a[1])); }
    else if constexpr (N == 1) { asm volatile("s_nop 1" : "+v"(a[0])); }
  }
};

// column c of robot r: A[r][row][c], R0 = diag(eps) with a right-hand side column
template <int WPE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
qr16_kernel(const double* __restrict__ A, int n, int reps, double* __restrict__ Rout, unsigned long long* __restrict__ cyc) {
  const int h = threadIdx.x & 15, rob = blockIdx.x * 4 + (threadIdx.x >> 4);
  if (rob >= n) return;
  // lane -> column: the product's owner lanes (sub-lane 3 of leg 0 = the right-hand side; the other sub-lane-3 lanes mirror it)
  const int col = (h & 3) == 3 ? 12 : 3 * (h >> 2) + (h & 3);
  double a0[P];
  for (int i = 0; i < P; i++) a0[i] = A[((size_t)rob * P + i) * NC + col];
  Hex16 qo;
  double chk = 0.0, Rcol[12];
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int rp = 0; rp < reps; rp++) {
    double Acol[P];
    for (int i = 0; i < P; i++) Acol[i] = a0[i] + chk * 1e-300;      // (depends on the previous repetition: nothing is hoisted)
    for (int k = 0; k < 12; k++) Rcol[k] = (col == wbc::hex_piv(k)) ? 1e-4 : 0.0;   // slot k = variable hex_piv(k) (the product's graded pivot order)
    wbc::hex_qr_append<Hex16, P, 12>(qo, Rcol, Acol);
    chk += Rcol[11];
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((h & 3) != 3 || h == 3)
    for (int k = 0; k < 12; k++) Rout[((size_t)rob * 12 + k) * NC + col] = Rcol[k] + chk * 1e-300;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// ---- (ii) one wavefront per robot
__device__ __forceinline__ double allreduce4(double x) {
  unsigned lo = (unsigned)__double_as_longlong(x), hi = (unsigned)(__double_as_longlong(x) >> 32);
  {
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double u = __longlong_as_double(((long long)b[0] << 32) | a[0]), v = __longlong_as_double(((long long)b[1] << 32) | a[1]);
    x = u + v;
  }
  lo = (unsigned)__double_as_longlong(x); hi = (unsigned)(__double_as_longlong(x) >> 32);
  {
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const double u = __longlong_as_double(((long long)b[0] << 32) | a[0]), v = __longlong_as_double(((long long)b[1] << 32) | a[1]);
    x = u + v;
  }
  return x;
}
#define WBC_F8(A, X) "v_fmac_f64_dpp %" #A ", %" #X ", %" #X " row_newbcast:%[ln] row_mask:0xf bank_mask:0xf\n\t"
template <int SRC> __device__ __forceinline__ void dot8(double& ta, double& tb, const double* a) {
  asm volatile(WBC_F8(0, 2) WBC_F8(1, 3) WBC_F8(0, 4) WBC_F8(1, 5) WBC_F8(0, 6) WBC_F8(1, 7) WBC_F8(0, 8) WBC_F8(1, 9)
               : "+v"(ta), "+v"(tb) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), [ln] "n"(SRC));
}
template <int WPE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
qr64_kernel(const double* __restrict__ A, int n, int reps, double* __restrict__ Rout, unsigned long long* __restrict__ cyc) {
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15, rob = blockIdx.x;
  if (rob >= n) return;
  double a0[8];
  for (int i = 0; i < 8; i++) {
    const int row = 8 * g + i;
    a0[i] = (row < P && c < NC) ? A[((size_t)rob * P + row) * NC + c] : 0.0;
  }
  Hex16 qo;
  double chk = 0.0, Rc[12];
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int rp = 0; rp < reps; rp++) {
    double Ac[8];
    for (int i = 0; i < 8; i++) Ac[i] = a0[i] + chk * 1e-300;
    for (int k = 0; k < 12; k++) Rc[k] = (c == wbc::hex_piv(k)) ? 1e-4 : 0.0;
    Hex16::dpp_fence<8>(Ac);
    wbc::static_for<12>([&](auto K) {
      constexpr int k = K, pv = wbc::hex_piv(k);           // the pivot column's lane within every DPP row
      double ta = 0.0, tb = 0.0;
      dot8<pv>(ta, tb, Ac);                                 // partial dots of this row group: sum_i A[i][k] A[i][c]
      const double t = allreduce4(ta + tb);                // over the four row groups, result on every lane of column c
      const double s2 = qo.bcast16(t, pv), rkk = qo.bcast16(Rc[k], pv);
      const double nrm = wbc::fast_sqrt(rkk * rkk + s2);
      const double alpha = (rkk > 0.0) ? -nrm : nrm;
      const double v0 = rkk - alpha;
      const double beta = (s2 > 0.0) ? wbc::fast_rcp(nrm * (nrm + fabs(rkk))) : 0.0;
      const double ns = -((v0 * Rc[k] + t) * beta);
      Rc[k] += ns * v0;
      wbc::static_for<8>([&](auto I) { Ac[I] = qo.template fma_bc<pv>(Ac[I], Ac[I], ns); });
    });
    chk += Rc[11];
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (g == 0 && c < NC)
    for (int k = 0; k < 12; k++) Rout[((size_t)rob * 12 + k) * NC + c] = Rc[k] + chk * 1e-300;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class F> static double time_ms(F launch) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch();   // warm
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 200;
  std::vector<int> sizes = {4096, 32768};
  const int nmax = 32768;
  std::vector<double> hA((size_t)nmax * P * NC);
  srand(7);
  for (auto& x : hA) x = (double)rand() / RAND_MAX - 0.5;
  // two scales, like the tick's rows: the last 12 rows are eps-sized
  for (int r = 0; r < nmax; r++)
    for (int i = 18; i < P; i++)
      for (int c = 0; c < NC; c++) hA[((size_t)r * P + i) * NC + c] *= 1e-4;
  double *dA, *dR16, *dR64;
  unsigned long long* dC;
  CHECK(hipMalloc(&dA, hA.size() * 8)); CHECK(hipMalloc(&dR16, (size_t)nmax * 12 * NC * 8)); CHECK(hipMalloc(&dR64, (size_t)nmax * 12 * NC * 8));
  CHECK(hipMalloc(&dC, (size_t)nmax * 8));
  CHECK(hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice));
  // correctness: one append each, |R| compared (row signs may differ)
  {
    const int n = 4096;
    hipLaunchKernelGGL(qr16_kernel<1>, dim3(n / 4), dim3(64), 0, 0, dA, n, 1, dR16, dC);
    hipLaunchKernelGGL(qr64_kernel<4>, dim3(n), dim3(64), 0, 0, dA, n, 1, dR64, dC);
    CHECK(hipDeviceSynchronize());
    std::vector<double> r16((size_t)n * 12 * NC), r64((size_t)n * 12 * NC);
    CHECK(hipMemcpy(r16.data(), dR16, r16.size() * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(r64.data(), dR64, r64.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int r = 0; r < n; r++)
      for (int k = 0; k < 12; k++)
      {
        double rown = 0.0;
        for (int c = 0; c < NC; c++) rown = fmax(rown, fabs(r16[((size_t)r * 12 + k) * NC + c]));
        for (int c = 0; c < NC; c++) {
          bool later = (c == 12);                                     // entry (row k, column c) of R is meaningful when column c is pivoted at or after step k
          for (int kk = k; kk < 12; kk++) later = later || (wbc::hex_piv(kk) == c);
          if (!later) continue;
          const double a = fabs(r16[((size_t)r * 12 + k) * NC + c]), b = fabs(r64[((size_t)r * 12 + k) * NC + c]);
          const double d = fabs(a - b) / rown;
          if (d > worst) worst = d;
        }
      }
    printf("check: max | |R16| - |R64| | / max|row| over %d robots = %.2e\n", n, worst);
  }
  printf("%d appends of %d rows x %d columns per robot, back to back, from registers\n", reps, P, NC);
  printf("%-46s %8s %12s %14s %16s\n", "mapping", "N", "kernel ms", "ns / append", "cycles / wave-append");
  auto report = [&](const char* name, int n, double ms, int nwaves) {
    std::vector<unsigned long long> cy(nwaves);
    CHECK(hipMemcpy(cy.data(), dC, (size_t)nwaves * 8, hipMemcpyDeviceToHost));
    double s = 0; for (auto x : cy) s += (double)x;
    printf("%-46s %8d %12.3f %14.2f %16.0f\n", name, n, ms, ms * 1e6 / ((double)n * reps), s / nwaves / reps);
  };
  for (int n : sizes) {
    report("16 lanes / robot, 1 wavefront / SIMD (product)", n, time_ms([&] { hipLaunchKernelGGL(qr16_kernel<1>, dim3(n / 4), dim3(64), 0, 0, dA, n, reps, dR16, dC); }), n / 4);
    report("16 lanes / robot, 2 wavefronts / SIMD", n, time_ms([&] { hipLaunchKernelGGL(qr16_kernel<2>, dim3(n / 4), dim3(64), 0, 0, dA, n, reps, dR16, dC); }), n / 4);
    report("64 lanes / robot, 1 wavefront / SIMD", n, time_ms([&] { hipLaunchKernelGGL(qr64_kernel<1>, dim3(n), dim3(64), 0, 0, dA, n, reps, dR64, dC); }), n);
    report("64 lanes / robot, 2 wavefronts / SIMD", n, time_ms([&] { hipLaunchKernelGGL(qr64_kernel<2>, dim3(n), dim3(64), 0, 0, dA, n, reps, dR64, dC); }), n);
    report("64 lanes / robot, 4 wavefronts / SIMD", n, time_ms([&] { hipLaunchKernelGGL(qr64_kernel<4>, dim3(n), dim3(64), 0, 0, dA, n, reps, dR64, dC); }), n);
  }
  return 0;
}
