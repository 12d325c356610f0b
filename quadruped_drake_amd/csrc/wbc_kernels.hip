// wbc_kernels.hip -- HIP kernels (gfx950) + the C ABI of include/wbc.h.
//
// Layout in HBM: struct-of-arrays, batch index fastest (row r of robot i at base[r*ld + i]).  The whole tick
// (FK -> CRBA/RNEA -> reduced QP assembly -> QR -> active set -> torques) is ONE fused launch, so the only HBM
// traffic is the 864 algorithmic bytes per tick.  One product kernel family: wbc_hex_kernel, 16 lanes (one DPP row)
// per robot, one QP column per lane (wbc_hex.hpp).  The round-1 lane-per-robot and quad-per-robot mappings lost at
// every batch size (profiles/r01/hex_sweep.md) and are retired; wbc_tick.hpp remains as the shared per-leg math and
// as the host-instantiated scalar form (tests, flop counting).
//
// There is no CPU path in this file: if HIP fails the entry points return an error.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>

#include "../../include/wbc.h"
#include "../../include/wbc_extras.h"
#ifdef WBC_STAMPS
__device__ unsigned long long g_wbc_stamps[16 * 4096];
#endif
#include "wbc_model.hpp"
#include "wbc_tick.hpp"
#include "wbc_hex.hpp"
#include "wbc_traj_dev.hpp"
#include "wbc_device_guard.hpp"

namespace {

thread_local char g_err[512] = "";

int fail(const char* what, hipError_t e) {
  snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
  return -2;
}
int misuse(const char* what) {
  snprintf(g_err, sizeof g_err, "%s", what);
  return -1;
}
#define HIP_TRY(x)                                  \
  do {                                              \
    hipError_t e_ = (x);                            \
    if (e_ != hipSuccess) return fail(#x, e_);      \
  } while (0)

struct StatsDev {
  double ticks, status_nonzero, iters_sum, tau_abs_sum;
  unsigned long long tau_abs_max_bits;
  double err_sum;
  double mask_count[16];
};

constexpr int BLOCK = 64;
// The end-of-rollout statistics are accumulated with fire-and-forget atomics into one of STAT_SLOTS
// per-block slots: 22 atomics per wavefront on ONE address set serialise in L2 (measured: +50 us at
// 1024 wavefronts, profiles/r02); spread over slots they pipeline.  wbc_stats_get reduces the slots.
constexpr int STAT_SLOTS = 2048;
static_assert(sizeof(StatsDev) == 22 * 8, "StatsDev is 22 eight-byte words (wbc_stats_reduce_kernel)");

// The slots are stored WORD-major (word w of slot s at words[w * STAT_SLOTS + s]): the reduction reads each word's 2048
// slots as one coalesced stream (the slot-major layout cost 11 us per wbc_stats_get, profiles/r02/hex_kernel_stats.csv).
__device__ __forceinline__ void stat_add(StatsDev* stats, unsigned block, int st, int iters, double tau_sum, double tau_max, double err, unsigned mk) {
  double* w = reinterpret_cast<double*>(stats) + (block & (STAT_SLOTS - 1));
  if (st != 0) atomicAdd(w + 1 * STAT_SLOTS, 1.0);      // (word 0, `ticks`, is not accumulated: it is the sum of the 16 mask bins, wbc_stats_get)
  atomicAdd(w + 2 * STAT_SLOTS, (double)iters);
  atomicAdd(w + 3 * STAT_SLOTS, tau_sum);
  atomicMax(reinterpret_cast<unsigned long long*>(w + 4 * STAT_SLOTS), (unsigned long long)__double_as_longlong(tau_max));
  atomicAdd(w + 5 * STAT_SLOTS, err);
  atomicAdd(w + (6 + mk) * STAT_SLOTS, 1.0);
}
// The tick kernels (the launches whose HBM traffic is reported) do it per wavefront, not per robot (round 4; rounds 1-3 and the
// persistent rollout kernels, which have no register to spare: the helper above, 7 atomics from each robot's lead lane): the four robots' contributions (each replicated on its robot's 16-lane
// row) are folded with the two wavefront-broadcast DPP steps (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3:
// lane 63 holds the total) and ONE lane issues the atomics: 6 lane-atomics per wavefront instead of 28.  Every atomic costs a
// 32-byte write in HBM (WRITE_SIZE 1296 B per wavefront against 528 B of outputs); what time it takes does not matter
// (HBM is at 2 % of its peak), the byte count is reported in roofline.traffic.
__device__ __forceinline__ double wave4_sum(double x) {
  x += __builtin_amdgcn_update_dpp(0.0, x, 0x142, 0xA, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0.0, x, 0x143, 0xC, 0xF, false);
  return x;
}
__device__ __forceinline__ double wave4_max(double x) {   // x >= 0, never NaN (compare + select: fmax() of a DPP result breaks this compiler's -save-temps bitcode round trip)
  const double a = __builtin_amdgcn_update_dpp(0.0, x, 0x142, 0xA, 0xF, false);
  x = (a > x) ? a : x;
  const double b = __builtin_amdgcn_update_dpp(0.0, x, 0x143, 0xC, 0xF, false);
  return (b > x) ? b : x;
}
__device__ __forceinline__ int wave4_sum(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
  return x;
}
__device__ __forceinline__ void stat_add_wave(StatsDev* stats, unsigned block, bool live, int st, int iters, double tau_sum, double tau_max,
                                              double err, unsigned mk) {
  const int n = wave4_sum(live ? 1 : 0), bad = wave4_sum((live && st != 0) ? 1 : 0), it = wave4_sum(live ? iters : 0);
  const double ts = wave4_sum(live ? tau_sum : 0.0), tm = wave4_max(live ? tau_max : 0.0), er = wave4_sum(live ? err : 0.0);
  (void)n;   // `ticks` is not accumulated: every live robot lands in exactly one mask bin, wbc_stats_get adds the 16 bins up
  double* w = reinterpret_cast<double*>(stats) + (block & (STAT_SLOTS - 1));
  if ((threadIdx.x & 63) == 63) {
    if (bad) atomicAdd(w + 1 * STAT_SLOTS, (double)bad);
    atomicAdd(w + 2 * STAT_SLOTS, (double)it);
    atomicAdd(w + 3 * STAT_SLOTS, ts);
    atomicMax(reinterpret_cast<unsigned long long*>(w + 4 * STAT_SLOTS), (unsigned long long)__double_as_longlong(tm));
    atomicAdd(w + 5 * STAT_SLOTS, er);
  }
  // contact masks: one lane-atomic per robot (one instruction for the four lead lanes; one atomic per DISTINCT mask of the wavefront
  // was measured: +1 % time for 30 KB less traffic, profiles/r04)
  if (live && (threadIdx.x & 15) == 0) atomicAdd(w + (6 + mk) * STAT_SLOTS, 1.0);
}

__device__ __forceinline__ double wave_sum(double x) {
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
__device__ __forceinline__ double wave_max(double x) {
  for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o, 64));
  return x;
}

constexpr int NIN = 19 + 18 + 54;    // input rows per robot (q, v, targets)
constexpr int MODEL_PAD_WORDS = 320;  // ModelC padded to 2.5 KB (multiple of 64 8-byte words)
// Replicas of the 2.5 KB model table in HBM, read by workgroup b % MODEL_REPLICAS: ONE copy is a hot line for a thousand wavefronts that start together (round 1: +18 us in the
// first phase), one per workgroup is 640 KB of HBM reads per launch for 3 MB of inputs.  32 = four per XCD under the round-robin dispatch: same launch time as 256 (22.6 us; N = 32768
// 150.6 against 151.8), HBM bytes per launch 5.33 -> 4.79 MB (profiles/r05/model_replicas.md).
constexpr int MODEL_REPLICAS = 32;
static_assert(sizeof(wbc::ModelC) <= MODEL_PAD_WORDS * 8 && MODEL_PAD_WORDS % 64 == 0, "model padding");

// ---------------------------------------------------------------- v4: 16 lanes (one DPP row) per robot
// Static-lane broadcasts are ONE v_mov_b64_dpp row_newbcast (gfx90a+ DP-ALU DPP); reductions over the
// four legs are row_ror:8 then row_ror:4 (after the first stage the data is 8-periodic, so the
// rotation by 4 pairs lane h with h^4: the butterfly is commutative-symmetric and every lane ends
// with bit-identical sums); in-leg traffic is quad_perm.
struct HexDev {
  int h;
  __device__ HexDev() : h(threadIdx.x & 15) {}
  __device__ __forceinline__ int lane() const { return h; }
  template <int CTRL> static __device__ __forceinline__ double dpp(double x) {
    return __builtin_amdgcn_update_dpp(0.0, x, CTRL, 0xF, 0xF, true);
  }
  template <int CTRL> static __device__ __forceinline__ int dppi(int x) {
    return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true);
  }
  __device__ __forceinline__ double bcast16(double x, int src) const {
    switch (src) {
      case 0: return dpp<0x150>(x);   case 1: return dpp<0x151>(x);   case 2: return dpp<0x152>(x);   case 3: return dpp<0x153>(x);
      case 4: return dpp<0x154>(x);   case 5: return dpp<0x155>(x);   case 6: return dpp<0x156>(x);   case 7: return dpp<0x157>(x);
      case 8: return dpp<0x158>(x);   case 9: return dpp<0x159>(x);   case 10: return dpp<0x15A>(x);  case 11: return dpp<0x15B>(x);
      case 12: return dpp<0x15C>(x);  case 13: return dpp<0x15D>(x);  case 14: return dpp<0x15E>(x);  default: return dpp<0x15F>(x);
    }
  }
  // acc + bcast16(x, src) * y as ONE instruction: v_fmac_f64_dpp with row_newbcast (the one DPP control the DP ALU
  // takes; the compiler never folds v_mov_b64_dpp into its consumer, profiles/r02/dpp_fmac.md).  The compiler's hazard
  // recogniser does not see a DPP read inside inline asm, so the "VALU write -> DPP read of the same VGPR needs 2 wait
  // states" rule is kept by construction: the asms are volatile (they stay in program order among themselves) and
  // every array that is DPP-read this way passes through dpp_fence() first (all its producers before the fence's
  // s_nop, all DPP reads after); a value these asms PRODUCE is fenced before compiler-generated DPP code reads it.
  template <int SRC> __device__ __forceinline__ double fma_bc(double acc, double x, double y) const {
    static_assert(SRC >= 0 && SRC < 16, "row_newbcast lane");
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y), "n"(SRC));
    return acc;
  }
  // Blocks of fused ops as ONE asm statement.  The hazard recogniser counts an inline-asm statement as zero wait
  // states and assumes its result may be read by the next one, so a run of single-instruction asms that accumulate
  // gets an s_nop per round (one per three ops with three accumulators, profiles/r02/dpp_fmac.md); inside one
  // statement the hardware's own interlocks order the dependent v_fmac_f64 (as in compiler-generated chains).
#define WBC_FD(A, X) "v_fmac_f64_dpp %" #A ", %" #X ", %" #X " row_newbcast:%[ln] row_mask:0xf bank_mask:0xf\n\t"
  // (ta, tb, tc) += sum over i of bcast16(a[i], SRC) * a[i], term i on accumulator i % 3
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
    else if constexpr (N == 3)
      asm volatile(WBC_FD(0, 3) WBC_FD(1, 4) WBC_FD(2, 5)
                   : "+v"(ta), "+v"(tb), "+v"(tc) : "v"(a[0]), "v"(a[1]), "v"(a[2]), [ln] "n"(SRC));
  }
#undef WBC_FD
#define WBC_FP(A, X, Y, L) "v_fmac_f64_dpp %" #A ", %" #X ", %" #Y " row_newbcast:%[" #L "] row_mask:0xf bank_mask:0xf\n\t"
  // three rows of an all-pairs block: d_j += sum over k < 6 of bcast16(x[k], Lj) * y[k]
  template <int L0, int L1, int L2> __device__ __forceinline__ void rows3_bc(double& d0, double& d1, double& d2, const double* x, const double* y) const {
    asm volatile(WBC_FP(0, 3, 9, l0) WBC_FP(1, 3, 9, l1) WBC_FP(2, 3, 9, l2) WBC_FP(0, 4, 10, l0) WBC_FP(1, 4, 10, l1) WBC_FP(2, 4, 10, l2) WBC_FP(0, 5, 11, l0) WBC_FP(1, 5, 11, l1) WBC_FP(2, 5, 11, l2) WBC_FP(0, 6, 12, l0) WBC_FP(1, 6, 12, l1) WBC_FP(2, 6, 12, l2) WBC_FP(0, 7, 13, l0) WBC_FP(1, 7, 13, l1) WBC_FP(2, 7, 13, l2) WBC_FP(0, 8, 14, l0) WBC_FP(1, 8, 14, l1) WBC_FP(2, 8, 14, l2)
                 : "+v"(d0), "+v"(d1), "+v"(d2)
                 : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]),
                   [l0] "n"(L0), [l1] "n"(L1), [l2] "n"(L2));
  }
  // the same with a seventh term (the rows' constant rides along: x[6] = the row's constant, y[6] = 1 on the right-hand-side
  // lanes and 0 on the column lanes -- no separate broadcast + select of the constant)
  template <int L0, int L1, int L2> __device__ __forceinline__ void rows3_bc7(double& d0, double& d1, double& d2, const double* x, const double* y) const {
    asm volatile(WBC_FP(0, 3, 10, l0) WBC_FP(1, 3, 10, l1) WBC_FP(2, 3, 10, l2) WBC_FP(0, 4, 11, l0) WBC_FP(1, 4, 11, l1) WBC_FP(2, 4, 11, l2) WBC_FP(0, 5, 12, l0) WBC_FP(1, 5, 12, l1) WBC_FP(2, 5, 12, l2) WBC_FP(0, 6, 13, l0) WBC_FP(1, 6, 13, l1) WBC_FP(2, 6, 13, l2) WBC_FP(0, 7, 14, l0) WBC_FP(1, 7, 14, l1) WBC_FP(2, 7, 14, l2) WBC_FP(0, 8, 15, l0) WBC_FP(1, 8, 15, l1) WBC_FP(2, 8, 15, l2) WBC_FP(0, 9, 16, l0) WBC_FP(1, 9, 16, l1) WBC_FP(2, 9, 16, l2)
                 : "+v"(d0), "+v"(d1), "+v"(d2)
                 : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]),
                   [l0] "n"(L0), [l1] "n"(L1), [l2] "n"(L2));
  }
#undef WBC_FP
  // s_nop 4 = 5 wait states: covers the DPP-source rule (2) and "VALU wrote EXEC" (5)
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
  __device__ __forceinline__ double leg_bcast(double x, int s0) const {
    switch (s0) {
      case 0: return dpp<0x00>(x);
      case 1: return dpp<0x55>(x);
      case 2: return dpp<0xAA>(x);
      default: return dpp<0xFF>(x);
    }
  }
  __device__ __forceinline__ double leg_pairs(double x) const { return dpp<0x50>(x); }  // quad_perm [0,0,1,1]
  // dynamic (robot-uniform) source lane within the row: ds_bpermute (LDS crossbar, no LDS memory)
  __device__ __forceinline__ double bcast16d(double x, int src) const { return __shfl(x, (threadIdx.x & 48) | src, 64); }
  __device__ __forceinline__ int bcast16d_i(int x, int src) const { return __shfl(x, (threadIdx.x & 48) | src, 64); }
  __device__ __forceinline__ double leg_sum(double x) const {
    x += dpp<0xB1>(x);
    x += dpp<0x4E>(x);
    return x;
  }
  __device__ __forceinline__ double legs_sum(double x) const {
    x += dpp<0x128>(x);  // row_ror:8
    x += dpp<0x124>(x);  // row_ror:4
    return x;
  }
  __device__ __forceinline__ double sum16(double x) const { return legs_sum(leg_sum(x)); }
  // Sum over the four legs of N values that are REPLICATED on the four sub-lanes of their leg, result on all 16 lanes:
  // ((x[lane 0] + x[lane 4]) + x[lane 8]) + x[lane 12] as one v_mov_b64_dpp + three fused broadcast-FMAs with 1.0 (4
  // instructions instead of the 6 of the row_ror butterfly; the DP ALU takes no other DPP control).  One hazard fence
  // for the whole batch, the N chains are independent.
  template <int N> __device__ __forceinline__ void legs_sum_n(double* x) const {
    dpp_fence<N>(x);
    const double one = 1.0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      double acc;
      asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                   "v_fmac_f64_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                   "v_fmac_f64_dpp %0, %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                   "v_fmac_f64_dpp %0, %1, %2 row_newbcast:12 row_mask:0xf bank_mask:0xf"
                   : "=&v"(acc) : "v"(x[i]), "v"(one));
      x[i] = acc;
    }
  }
  // v_min_f64 / v_max_f64 through asm: fmin()/fmax() first canonicalise both operands (a v_max_f64 x, x, x each), which
  // only matters for signalling NaNs -- the keys here are never NaN (HEX_NONE is finite)
  static __device__ __forceinline__ double vmin(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
  static __device__ __forceinline__ double vmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
  __device__ __forceinline__ double min16(double x) const {
    x = vmin(x, dpp<0xB1>(x));
    x = vmin(x, dpp<0x4E>(x));
    x = vmin(x, dpp<0x128>(x));
    x = vmin(x, dpp<0x124>(x));
    return x;
  }
  __device__ __forceinline__ double max16(double x) const {
    x = vmax(x, dpp<0xB1>(x));
    x = vmax(x, dpp<0x4E>(x));
    x = vmax(x, dpp<0x128>(x));
    x = vmax(x, dpp<0x124>(x));
    return x;
  }
  __device__ __forceinline__ bool any16(bool b) const {
    int x = b ? 1 : 0;
    x |= dppi<0xB1>(x);
    x |= dppi<0x4E>(x);
    x |= dppi<0x128>(x);
    x |= dppi<0x124>(x);
    return x != 0;
  }
  __device__ __forceinline__ bool wave_all(bool b) const { return __all(b); }
  __device__ __forceinline__ bool wave_any(bool b) const { return __any(b); }
  __device__ __forceinline__ int wave_max_int(int x) const {   // maximum over the wavefront of a value in 0..15, wave-uniform
    int m = 0;
    bool cand = true;        // lanes whose value still agrees with the maximum's leading bits
#pragma unroll
    for (int b = 3; b >= 0; b--) {
      const bool bit = (x >> b) & 1;
      if (__any(cand && bit)) { m |= 1 << b; cand = cand && bit; }
    }
    return __builtin_amdgcn_readfirstlane(m);
  }
};

constexpr int HEX_BLOCK = 64;              // threads per workgroup of the 16-lane kernel: one wavefront (256 = one model copy per four wavefronts measured
                                           // the same at N = 4096 and 3.5 % slower at N = 32768: a CU slot frees only when all four wavefronts are done)
constexpr int HROBOTS = HEX_BLOCK / 16;    // robots per workgroup
// robot-level park in LDS (wbc_hex.hpp): reads go through a laundered pointer so that the compiler
// cannot forward the stored values (i.e. keep them in registers / spill them) yet the loads stay
// ordinary, schedulable LDS loads
struct ParkLds {
  double* a;
  double* la;   // lane-private slots: element i of this lane at la[i * HEX_BLOCK] (consecutive lanes = consecutive banks)
  int z;  // an opaque zero: reads go through a[z + i], which the compiler can neither forward from the stores
          // nor demote to a generic (flat) access -- laundering the POINTER loses the LDS address space and
          // turns every read into a flat_load with vmcnt waits (measured, profiles/r01/hex_cuts.md)
  __device__ __forceinline__ ParkLds(double* p, double* lp) : a(p), la(lp), z(0) { asm volatile("" : "+v"(z)); }
  __device__ __forceinline__ void put(int i, double v) { a[i] = v; }
  __device__ __forceinline__ double get(int i) const { return a[z + i]; }
  __device__ __forceinline__ void lput(int i, double v) { la[i * HEX_BLOCK] = v; }
  __device__ __forceinline__ double lget(int i) const { return la[z + i * HEX_BLOCK]; }
};

// XCD-aware workgroup -> robots mapping.  A 64-thread workgroup stages 4 robots = one 32-byte segment of every input row, and the
// hardware deals workgroups round-robin over the 8 XCDs (each with its own L2), so with the identity mapping the four segments of
// a 128-byte line are fetched by four different XCDs: FETCH_SIZE measured 4x the algorithmic input bytes (calibrated on the same
// access shape: tools/micro/fetch_calib.hip, profiles/r02/fetch_calib.md).  Here workgroups b, b+8, b+16, b+24 -- the same XCD --
// take the four segments of one line.  Placement is only a speed / traffic matter; any mapping is correct.
__device__ __forceinline__ int hex_effective_block(int b, int nblocks) {
  const int full = nblocks & ~31;                 // groups of 32 workgroups = 8 XCDs x 4 segments; the ragged tail keeps the identity
  if (b >= full) return b;
  const int g = b >> 5, x = b & 7, y = (b >> 3) & 3;
  return (g << 5) + (x << 2) + y;
}

// One wavefront per SIMD by design: 446 registers (two per SIMD by amdgpu_waves_per_eu(2): 996 B/lane of scratch, 2.3x slower, profiles/r02)
template <int KIND, bool TB = false>
__global__ void __launch_bounds__(HEX_BLOCK) __attribute__((amdgpu_waves_per_eu(1)))
wbc_hex_kernel(const wbc::ModelC* __restrict__ mp, const wbc::ParamsX* __restrict__ pp, int n, int ld,
               const double* __restrict__ q, const double* __restrict__ v, const double* __restrict__ tg,
               const uint8_t* __restrict__ mask, const double* __restrict__ mu,
               const double* __restrict__ ms, double* __restrict__ tau, double* __restrict__ met,
               int32_t* __restrict__ status, StatsDev* __restrict__ stats, double* __restrict__ vdot) {
  constexpr int PER_LANE = (NIN * HROBOTS + HEX_BLOCK - 1) / HEX_BLOCK;   // 6 input words per lane
  constexpr int MPER = (MODEL_PAD_WORDS + HEX_BLOCK - 1) / HEX_BLOCK;
  __shared__ double mbuf[MPER * HEX_BLOCK];
  __shared__ double inbuf[PER_LANE * HEX_BLOCK];                      // 91 rows x 4 robots, padded to whole lanes
  __shared__ double parkbuf[HROBOTS * wbc::PK_N];
  __shared__ double lanebuf[wbc::LP_N * HEX_BLOCK];
#ifdef WBC_PAD   // diagnostic: shift the kernel body by WBC_PAD instructions (code-placement sensitivity, profiles/r05/code_placement.md)
  asm volatile(".rept %0\n\ts_nop 0\n\t.endr" :: "n"(WBC_PAD));
#endif
  WBC_STAMP(0);
#ifdef WBC_STAMPS   // where this wavefront runs: HW_ID (cu / simd / se) and XCC_ID
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_wbc_stamps[(size_t)blockIdx.x * 16 + 8] = hw;
    g_wbc_stamps[(size_t)blockIdx.x * 16 + 9] = xcc;
  }
#endif
  const int slot = threadIdx.x >> 4;
  const int eb = hex_effective_block(blockIdx.x, gridDim.x);
  const int i = eb * HROBOTS + slot;
  const bool live = i < n;
  const int ii = live ? i : (n - 1);
  // Prologue: EVERY global load of the tick is issued here, unconditionally (clamped indices, no divergent
  // guards), before the single wait: model table (per-block replica: concurrent blocks hit different lines / channels), the 91 x 4 input
  // words, and the per-robot mask / mu / mass scale -- one memory round trip instead of two (profiles/r01/hex_cuts.md).
  const double* msrc = reinterpret_cast<const double*>(mp) + (size_t)(blockIdx.x % MODEL_REPLICAS) * MODEL_PAD_WORDS;
  double t[MPER], tmp[PER_LANE];
#pragma unroll
  for (int j = 0; j < MPER; j++) t[j] = msrc[min(j * HEX_BLOCK + (int)threadIdx.x, MODEL_PAD_WORDS - 1)];
  {
    const int r0 = eb * HROBOTS;
    // Word j * 64 + t of the staged block is row 16 j + (t >> 2) of robot slot t & 3: the slot (hence the robot) is a per-lane constant and the row of
    // load j is 16 j + tr, so which array a load reads is known at compile time except for two of the six (rows 16 - 31: q | v, rows 32 - 47: v | targets).
    // Offsets are 32-bit (wbc_step rejects ld > WBC_MAX_LD = 2^23, so 54 rows x ld x 8 bytes stay below 4 GB): one multiply-add per address instead of the
    // two 64-bit multiply-adds, the selects between three base pointers and the 64-bit additions of the generic form (192 -> ~120 instructions before the
    // last load is issued; stamps: profiles/r05/prologue.md).
    static_assert(HEX_BLOCK == 64 && NIN == 91 && HROBOTS == 4 && PER_LANE == 6, "prologue specialised to 91 rows x 4 robots per wavefront");
    const unsigned tr = threadIdx.x >> 2;
    const unsigned rob = (unsigned)min(r0 + (int)(threadIdx.x & 3), n - 1), uld = (unsigned)ld;
    tmp[0] = q[tr * uld + rob];
    {
      const bool lo = tr < 3u;                                   // rows 16 - 18: q, rows 19 - 31: v rows 0 - 12
      const double* b1 = lo ? q : v;
      tmp[1] = b1[(lo ? 16u + tr : tr - 3u) * uld + rob];
    }
    {
      const bool lo = tr < 5u;                                   // rows 32 - 36: v rows 13 - 17, rows 37 - 47: targets rows 0 - 10
      const double* b2 = lo ? v : tg;
      tmp[2] = b2[(lo ? 13u + tr : tr - 5u) * uld + rob];
    }
    tmp[3] = tg[(11u + tr) * uld + rob];
    tmp[4] = tg[(27u + tr) * uld + rob];
    tmp[5] = tg[min(43u + tr, 53u) * uld + rob];                 // rows 80 - 90, the lanes beyond re-read the last row
  }
  WBC_STAMP(1);   // all loads issued
  const unsigned mk = mask[ii] & 0xF;   // (bits 4..7 are not read)
  // per-instance friction coefficient / mass scale, or the handle's mu / 1.0 read from the parameter block: one unconditional load each (the pointer
  // is selected, not the load skipped: two taken branches and a dependent scalar load less in every tick without domain randomisation)
  const double mu_in = *(mu ? mu + ii : &pp->mu), ms_in = *(ms ? ms + ii : &pp->one);
#pragma unroll
  for (int j = 0; j < MPER; j++) mbuf[j * HEX_BLOCK + threadIdx.x] = t[j];
#pragma unroll
  for (int j = 0; j < PER_LANE; j++) inbuf[j * HEX_BLOCK + threadIdx.x] = tmp[j];
  WBC_STAMP(2);   // LDS writes issued (loads landed)
  __syncthreads();
  WBC_STAMP(3);
  const wbc::ModelC& m = *reinterpret_cast<const wbc::ModelC*>(mbuf);
  const wbc::ParamsX& P = *pp;
  HexDev qo;
  auto in = [&](int r) -> double { return inbuf[r * HROBOTS + slot]; };
  double tsum = 0.0, tmax = 0.0, errv = 0.0;
  auto ot = [&](int k, double x) {
    if (live) tau[(size_t)k * ld + ii] = x;
    tsum += fabs(x);
    tmax = fmax(tmax, fabs(x));
  };
  const bool lead = qo.h == 0;
  auto om = [&](int k, double x) {   // rows 0..3: metrics (lead lane), rows 4..9: base accelerations (lead), 10..21: own joint
    if (k >= 4) {
      if (live && vdot && (k >= 10 || lead)) vdot[(size_t)(k - 4) * ld + ii] = x;
      return;
    }
    if (live && lead && met) met[(size_t)k * ld + ii] = x;
    if (k == 1) errv = x;
  };
  const double mui = mu_in;
  const double msi = ms_in;
  int iters = 0;
#ifdef WBC_FORCE_SCRATCH   // diagnostic: give the kernel a private segment without changing its math
  volatile double junk[WBC_FORCE_SCRATCH];
  for (int k = 0; k < WBC_FORCE_SCRATCH; k++) junk[k] = inbuf[k];
#endif
  ParkLds park(parkbuf + slot * wbc::PK_N, lanebuf + threadIdx.x);
  const int st = wbc::hex_tick<HexDev, KIND, TB>(m, P, qo, in, mk, mui, msi, park, ot, om, &iters);
#ifdef WBC_FORCE_SCRATCH
  if (junk[threadIdx.x % WBC_FORCE_SCRATCH] == 1.2345e300) iters++;
#endif
  WBC_STAMP(4);   // tick done
  if (live && lead && status) status[ii] = st;
  if (stats) {
    // per-robot reductions on the DPP row, the four robots folded on the wavefront, then fire-and-forget atomics from ONE lane
    // into the block's slot (no waits at the tail of the kernel)
    const double ts = qo.sum16(tsum), tm = qo.max16(tmax);
    stat_add_wave(stats, blockIdx.x, live, st, iters, ts, tm, errv, mk);
  }
  WBC_STAMP(5);
}

// ---------------------------------------------------------------- persistent closed loop
// `steps` ticks of  lookup(time) -> tick -> semi-implicit Euler -> time += dt  inside ONE launch: every wavefront
// runs its four robots' whole rollout on its own, the state (q, v) lives in the LDS input buffer between ticks.
// No launch boundaries, no start-up latency per tick, and -- because the wavefronts never wait for each other --
// the per-tick active-set tail of the slowest robot averages out over the rollout instead of ending every launch.
// Arithmetic is identical to the launch-per-stage path (wbc_integrate_kernel, traj_lookup_kernel): the two agree
// bit for bit (tests/test_rollout.py).
template <int KIND, bool TB>
__global__ void __launch_bounds__(HEX_BLOCK)
wbc_hex_rollout_kernel(const wbc::ModelC* __restrict__ mp, const wbc::ParamsX* __restrict__ pp, int n, int ld, int steps,
                       double dt, wbc::TrajDev T, double* __restrict__ q, double* __restrict__ v, double* __restrict__ time,
                       double* __restrict__ tg, uint8_t* __restrict__ mask, const double* __restrict__ mu,
                       const double* __restrict__ ms, double* __restrict__ tau, double* __restrict__ met,
                       int32_t* __restrict__ status, StatsDev* __restrict__ stats, double* __restrict__ vdot, int warm) {
  constexpr int PER_LANE = (NIN * HROBOTS + HEX_BLOCK - 1) / HEX_BLOCK;
  constexpr int MPER = (MODEL_PAD_WORDS + HEX_BLOCK - 1) / HEX_BLOCK;
  constexpr int NST = 37;   // state rows q (19) + v (18)
  __shared__ double mbuf[MPER * HEX_BLOCK];
  __shared__ double inbuf[PER_LANE * HEX_BLOCK];
  __shared__ double parkbuf[HROBOTS * wbc::PK_N];
  __shared__ double lanebuf[wbc::LP_N * HEX_BLOCK];
  __shared__ double vdbuf[HROBOTS * 18];
  __shared__ double outbuf[HROBOTS * 18];
  __shared__ double robuf[HROBOTS * 4];     // per robot: time, lookup hint, mu, mass scale (loop-carried, kept out of registers)   // last tick: tau (12), metrics (4), status, mask -- written to HBM once, after the loop
  const int slot = threadIdx.x >> 4;
  const int i = hex_effective_block(blockIdx.x, gridDim.x) * HROBOTS + slot;
  const bool live = i < n;
  const int ii = live ? i : (n - 1);
  HexDev qo;
  const int h = qo.h;
  const bool lead = h == 0;
  {
    const double* msrc = reinterpret_cast<const double*>(mp) + (size_t)(blockIdx.x % MODEL_REPLICAS) * MODEL_PAD_WORDS;
#pragma unroll
    for (int j = 0; j < MPER; j++) mbuf[j * HEX_BLOCK + threadIdx.x] = msrc[min(j * HEX_BLOCK + (int)threadIdx.x, MODEL_PAD_WORDS - 1)];
    // own robot's state rows: lane h takes rows h, h+16, h+32 (< 37)
#pragma unroll
    for (int r = h; r < NST; r += 16)
      inbuf[r * HROBOTS + slot] = (r < 19) ? q[(size_t)r * ld + ii] : v[(size_t)(r - 19) * ld + ii];
    for (int r = h; r < 18; r += 16) vdbuf[slot * 18 + r] = 0.0;   // vdot is an OUTPUT: never read from the caller
  }
  if (lead) {
    robuf[slot * 4 + 0] = time[ii];
    robuf[slot * 4 + 1] = (double)(T.K / 2);
    robuf[slot * 4 + 2] = mu ? mu[ii] : pp->mu;
    robuf[slot * 4 + 3] = ms ? ms[ii] : 1.0;
  }
  __syncthreads();
  auto in = [&](int r) -> double { return inbuf[r * HROBOTS + slot]; };
  ParkLds park(parkbuf + slot * wbc::PK_N, lanebuf + threadIdx.x);
  // Warm start (wbc_set_warm_start; wbc_hex.hpp: hex_gi<..., WARM>): was this lane's friction row active when the robot's previous tick ended?  The robot
  // stays on this wavefront for the whole rollout, so last tick's active set is one bit per lane that never leaves the chip.  `warm` = 0 (the default):
  // the bit is never set and every pick is the cold start's -- the rollout then equals the launch-per-stage loop bit for bit.
  bool seed = false;
  for (int step = 0; step < steps; step++) {
    // The model table and the parameters are loop-invariant, and the compiler would hoist ~130 doubles of them out
    // of the step loop into registers (the kernel then spills): an opaque zero offset per iteration keeps those
    // reads where the tick uses them.
    int zoff = 0;
    asm volatile("" : "+v"(zoff));
    // the same for everything derived from the lane id (dozens of per-lane predicate masks would otherwise live in
    // SGPR pairs across the whole loop body)
    qo.h = h + zoff;
    const wbc::ModelC& m = *reinterpret_cast<const wbc::ModelC*>(mbuf + zoff);
    const wbc::ParamsX* ppi = pp;
    asm volatile("" : "+s"(ppi));
    const wbc::ParamsX& P = *ppi;
    const double tnow = robuf[slot * 4 + 0], mui = robuf[slot * 4 + 2], msi = robuf[slot * 4 + 3];
    const int hint = (int)robuf[slot * 4 + 1];
    // ---- targets and contact mask of this tick (planners/towr.py:92-148)
    const int c = wbc::traj_index(T, tnow, hint);
    const double* src = c < 0 ? T.standing : T.table + (size_t)c * 54;
    const unsigned mk = (c < 0 ? T.standing_mask : T.masks[c]) & 0xF;
#pragma unroll
    for (int e = h; e < 54; e += 16) {
      inbuf[(NST + e) * HROBOTS + slot] = src[e];
    }
    __syncthreads();
    // ---- the tick
    double tsum = 0.0, tmax = 0.0, errv = 0.0;
    auto ot = [&](int k, double x) {
      outbuf[slot * 18 + k] = x;
      tsum += fabs(x);
      tmax = fmax(tmax, fabs(x));
    };
    auto om = [&](int k, double x) {
      if (k >= 4) {
        if (k >= 10 || lead) vdbuf[slot * 18 + (k - 4)] = x;
        return;
      }
      if (lead) outbuf[slot * 18 + 12 + k] = x;
      if (k == 1) errv = x;
    };
    int iters = 0;
    bool sd = seed;
    const int st = wbc::hex_tick<HexDev, KIND, TB, true>(m, P, qo, in, mk, mui, msi, park, ot, om, &iters, &sd);
    seed = (warm != 0) && (st == wbc::ST_OK) && sd;     // (a tick that was not solved leaves nothing worth starting from)
    if (lead) { outbuf[slot * 18 + 16] = (double)st; outbuf[slot * 18 + 17] = (double)mk; }
    if (stats) {
      const double ts = qo.sum16(tsum), tm = qo.max16(tmax);
      if (live && lead) {
        stat_add(stats, blockIdx.x, st, iters, ts, tm, errv, mk);
      }
    }
    __syncthreads();
    // ---- forward step, same arithmetic as wbc_integrate_kernel: v+ = v + dt vd; orientation by exp(dt/2 w+); p, joints by v+
#pragma unroll
    for (int r = h; r < 18; r += 16) {
      const double vn = inbuf[(19 + r) * HROBOTS + slot] + dt * vdbuf[slot * 18 + r];
      inbuf[(19 + r) * HROBOTS + slot] = vn;
      if (r >= 3) inbuf[(4 + (r - 3)) * HROBOTS + slot] += dt * vn;   // q rows 4..18 <- v rows 3..17
    }
    __syncthreads();
    if (lead) {
      const double w0 = in(19), w1 = in(20), w2 = in(21);
      const double wn = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
      const double ang = 0.5 * wn * dt;
      double dw = 1.0, dx = 0.0, dy = 0.0, dz = 0.0;
      if (wn > 0.0) {
        const double sc = sin(ang) / wn;
        dw = cos(ang); dx = sc * w0; dy = sc * w1; dz = sc * w2;
      }
      const double w1q = in(0), x1 = in(1), y1 = in(2), z1 = in(3);
      double qw = dw * w1q - dx * x1 - dy * y1 - dz * z1;
      double qx = dw * x1 + dx * w1q + dy * z1 - dz * y1;
      double qy = dw * y1 - dx * z1 + dy * w1q + dz * x1;
      double qz = dw * z1 + dx * y1 - dy * x1 + dz * w1q;
      const double inv = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
      inbuf[0 * HROBOTS + slot] = qw * inv; inbuf[1 * HROBOTS + slot] = qx * inv;
      inbuf[2 * HROBOTS + slot] = qy * inv; inbuf[3 * HROBOTS + slot] = qz * inv;
    }
    if (lead) {
      robuf[slot * 4 + 0] = tnow + dt;
      if (c >= 0) robuf[slot * 4 + 1] = (double)c;
    }
    __syncthreads();
  }
  // ---- final state and the last tick's outputs back to HBM
  if (live) {
#pragma unroll
    for (int r = h; r < NST; r += 16) {
      const double x = inbuf[r * HROBOTS + slot];
      if (r < 19) q[(size_t)r * ld + ii] = x; else v[(size_t)(r - 19) * ld + ii] = x;
    }
    for (int e = h; e < 54; e += 16) tg[(size_t)e * ld + ii] = inbuf[(NST + e) * HROBOTS + slot];
    for (int r = h; r < 18; r += 16) vdot[(size_t)r * ld + ii] = vdbuf[slot * 18 + r];
    if (h < 12) tau[(size_t)h * ld + ii] = outbuf[slot * 18 + h];
    else if (met) met[(size_t)(h - 12) * ld + ii] = outbuf[slot * 18 + h];
    if (lead) {
      time[ii] = robuf[slot * 4 + 0];
      mask[ii] = (uint8_t)(int)outbuf[slot * 18 + 17];
      if (status) status[ii] = (int32_t)outbuf[slot * 18 + 16];
    }
  }
}

// ---------------------------------------------------------------- forward step
// Semi-implicit Euler in the reference's coordinates: v = [w_WB (world); v_WBo (world); qd].
__global__ void __launch_bounds__(64) wbc_integrate_kernel(int n, int ld, double dt, double* __restrict__ q, double* __restrict__ v,
                                     const double* __restrict__ vd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // EVERY load first: the rows of q are read and written through one pointer with a run-time stride, so the compiler cannot move a load of q above a
  // store to q -- written row by row the kernel was three dependent trips to memory (v | quaternion | the 15 other rows of q); this way it is one
  double vn[18], qq[19];
#pragma unroll
  for (int r = 0; r < 18; r++) vn[r] = v[(size_t)r * ld + i];
#pragma unroll
  for (int r = 0; r < 19; r++) qq[r] = q[(size_t)r * ld + i];
#pragma unroll
  for (int r = 0; r < 18; r++) vn[r] = vn[r] + dt * vd[(size_t)r * ld + i];
  // orientation: q+ = exp(dt/2 w) (x) q, w in the world frame
  const double wn = sqrt(vn[0] * vn[0] + vn[1] * vn[1] + vn[2] * vn[2]);
  const double ang = 0.5 * wn * dt;
  double dw = 1.0, dx = 0.0, dy = 0.0, dz = 0.0;
  if (wn > 0.0) {
    const double sc = sin(ang) / wn;
    dw = cos(ang); dx = sc * vn[0]; dy = sc * vn[1]; dz = sc * vn[2];
  }
  const double w1 = qq[0], x1 = qq[1], y1 = qq[2], z1 = qq[3];
  double qw = dw * w1 - dx * x1 - dy * y1 - dz * z1;
  double qx = dw * x1 + dx * w1 + dy * z1 - dz * y1;
  double qy = dw * y1 - dx * z1 + dy * w1 + dz * x1;
  double qz = dw * z1 + dx * y1 - dy * x1 + dz * w1;
  const double inv = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
  qq[0] = qw * inv; qq[1] = qx * inv; qq[2] = qy * inv; qq[3] = qz * inv;
#pragma unroll
  for (int r = 0; r < 3; r++) qq[4 + r] += dt * vn[3 + r];
#pragma unroll
  for (int r = 0; r < 12; r++) qq[7 + r] += dt * vn[6 + r];
#pragma unroll
  for (int r = 0; r < 18; r++) v[(size_t)r * ld + i] = vn[r];
#pragma unroll
  for (int r = 0; r < 19; r++) q[(size_t)r * ld + i] = qq[r];
}

// out <- sum / max over all slots: one 256-thread block per statistics word, coalesced reads (word-major slots)
__global__ void __launch_bounds__(256) wbc_stats_reduce_kernel(const StatsDev* __restrict__ st, StatsDev* __restrict__ out) {
  __shared__ double part[4];
  const int w = blockIdx.x;  // word 4 is the max (bit pattern of a non-negative double)
  const double* src = reinterpret_cast<const double*>(st) + (size_t)w * STAT_SLOTS;
  double acc = 0.0;
  for (int sl = threadIdx.x; sl < STAT_SLOTS; sl += 256) {
    const double x = src[sl];
    acc = (w == 4) ? fmax(acc, x) : acc + x;
  }
  acc = (w == 4) ? wave_max(acc) : wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double r = (w == 4) ? fmax(fmax(part[0], part[1]), fmax(part[2], part[3])) : (part[0] + part[1]) + (part[2] + part[3]);
    reinterpret_cast<double*>(out)[w] = r;
  }
}

__global__ void wbc_advance_time_kernel(int n, double dt, double* __restrict__ time) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) time[i] += dt;
}

}  // namespace

struct wbc_handle_s {
  int kind, max_batch, device, variant;
  int last_variant;  // kernel used by the most recent launch (1 lane, 2 quad, 3 hex)
  bool torque_box;
  uint32_t flags;
  hipStream_t stream;
  bool own_stream;
  bool warm_start;   // wbc_set_warm_start: wbc_rollout seeds every tick's active set with the previous tick's
  wbc::ModelC* d_model;
  wbc::ParamsX* d_params;
  StatsDev* d_stats;
  StatsDev* h_stats;  // pinned, device-mapped: the reduce kernel writes the totals straight into host memory
  hipEvent_t ev0, ev1;
  int timed_steps;  // launches between ev0 and ev1 of the last wbc_time_steps call (0: none yet)
  std::vector<hipEvent_t> evs;  // per-launch events of wbc_time_steps_each (grown on demand, outside any timed region)
  // staging buffers for WBC_HOST_PTRS
  double* d_vdot;  // optional [18][ld] output of the generalized accelerations (wbc_set_vdot_output)
  double *s_q, *s_v, *s_tg, *s_mu, *s_ms, *s_tau, *s_met;
  uint8_t* s_mask;
  int32_t* s_status;
  // WBC_HOST_PTRS, small batches (n <= ZC_MAX: the LeafSystem adapter's one robot): ONE pinned, device-mapped block.  The tick reads
  // its inputs and writes its outputs straight through it (728 + 132 bytes per robot over PCIe inside the kernel): no copy calls at
  // all -- each of the nine hipMemcpy*Async calls of the staged path costs the host ~10 us, the kernel itself ~15 us.  The outputs
  // reach the caller's arrays in wbc_sync (or at the next wbc_step), which is when the ABI promises them.
  double* z_host;
  double* z_dev;
  struct { bool active; int n, ld; double* tau; double* met; int32_t* status; } zpend;
};
constexpr int ZC_MAX = 64;
// block layout, in doubles, leading dimension n: q 19 | v 18 | targets 54 | mu | mass scale | tau 12 | metrics 4 ; then at fixed offsets status (int32) and mask (bytes)
constexpr size_t ZC_STATUS_OFF = 109 * ZC_MAX, ZC_MASK_OFF = ZC_STATUS_OFF + ZC_MAX / 2, ZC_DOUBLES = ZC_MASK_OFF + ZC_MAX / 8;

static int zc_finish(wbc_handle_s* h) {   // the pending small-batch tick: wait, hand the outputs over
  if (!h->zpend.active) return 0;
  HIP_TRY(hipStreamSynchronize(h->stream));
  const int n = h->zpend.n, ld = h->zpend.ld;
  const double* z = h->z_host;
  for (int r = 0; r < 12; r++) memcpy(h->zpend.tau + (size_t)r * ld, z + (size_t)(93 + r) * n, (size_t)n * 8);
  if (h->zpend.met) for (int r = 0; r < 4; r++) memcpy(h->zpend.met + (size_t)r * ld, z + (size_t)(105 + r) * n, (size_t)n * 8);
  if (h->zpend.status) memcpy(h->zpend.status, z + ZC_STATUS_OFF, (size_t)n * 4);
  h->zpend.active = false;
  return 0;
}

extern "C" int wbc_traj_raw_(wbc_traj t, wbc::TrajDev* out);   // wbc_traj.hip (internal)
// internal: wbc_traj.hip reports its failures through the same thread-local buffer wbc_last_error() returns
extern "C" void wbc_set_error_(const char* msg) { snprintf(g_err, sizeof g_err, "%s", msg ? msg : ""); }

extern "C" {

const char* wbc_last_error(void) { return g_err; }
int wbc_version(void) { return 100; }

int wbc_params_default(int kind, wbc_params* out) {
  if (!out || kind < WBC_KIND_ID || kind > WBC_KIND_CLF) return misuse("wbc_params_default: bad argument");
  static_assert(sizeof(wbc_params) == sizeof(wbc::ParamsC), "params layout");
  wbc::params_default(kind, reinterpret_cast<wbc::ParamsC*>(out));
  return 0;
}

int wbc_create(const wbc_model* model, int kind, const wbc_params* params, int max_batch, int device,
               uint32_t flags, wbc_handle* out) {
  if (!model || !out) return misuse("wbc_create: null argument");
  if (kind < WBC_KIND_ID || kind > WBC_KIND_CLF) return misuse("wbc_create: kind must be WBC_KIND_ID, _MPTC, _PC or _CLF");
  if (max_batch <= 0 || max_batch > WBC_MAX_LD) return misuse("wbc_create: max_batch must be in 1 .. WBC_MAX_LD (2^23)");
  wbc::ModelC m;
  if (wbc::model_from_flat(model->flat, &m)) return misuse("wbc_create: joint axes must be axis-aligned");
  if (!wbc::model_axes_are_xyy(&m))
    return misuse("wbc_create: unsupported kinematic tree -- the kernels are specialised to legs with the abduction joint "
                  "about +-x and the hip and knee joints about +-y (Mini Cheetah, ANYmal)");
  bool seen_q[12] = {0}, seen_a[12] = {0};
  for (int i = 0; i < 12; i++) {
    int a = model->q_perm[i], b = model->act_perm[i];
    if (a < 0 || a >= 12 || b < 0 || b >= 12 || seen_q[a] || seen_a[b]) return misuse("wbc_create: q_perm/act_perm must be permutations of 0..11");
    seen_q[a] = seen_a[b] = true;
  }
  {
    int qp[12], ap[12];
    for (int i = 0; i < 12; i++) { qp[i] = model->q_perm[i]; ap[i] = model->act_perm[i]; }
    wbc::model_set_perms(&m, qp, ap);
  }
  wbc::ParamsC P;
  wbc::params_default(kind, &P);
  if (params) memcpy(&P, params, sizeof P);
  if (!(P.mu > 0) || !(P.eps2 > 0) || !(P.w_body > 0) || !(P.w_foot > 0) || !(P.tau_max > 0))
    return misuse("wbc_create: mu, eps2, w_body, w_foot and tau_max must be positive");
  WBC_ON_DEVICE(device, fail);
  wbc_handle h = new wbc_handle_s();
  h->kind = kind; h->max_batch = max_batch; h->device = device; h->flags = flags;
  h->variant = 0;
  h->torque_box = P.tau_max < 1e300;
  // every allocation below is released by wbc_destroy on failure (null members are skipped)
  auto build = [&]() -> int {
    HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
    HIP_TRY(hipMalloc(&h->d_model, (size_t)MODEL_REPLICAS * MODEL_PAD_WORDS * 8));
    HIP_TRY(hipMalloc(&h->d_params, sizeof(wbc::ParamsX)));
    HIP_TRY(hipMalloc(&h->d_stats, sizeof(StatsDev) * (STAT_SLOTS + 1)));
    {
      // MODEL_REPLICAS padded copies: concurrent workgroups read different lines / channels
      char* rep = new char[(size_t)MODEL_REPLICAS * MODEL_PAD_WORDS * 8]();
      for (int r = 0; r < MODEL_REPLICAS; r++) memcpy(rep + (size_t)r * MODEL_PAD_WORDS * 8, &m, sizeof m);
      hipError_t e = hipMemcpy(h->d_model, rep, (size_t)MODEL_REPLICAS * MODEL_PAD_WORDS * 8, hipMemcpyHostToDevice);
      delete[] rep;
      if (e != hipSuccess) return fail("hipMemcpy(model replicas)", e);
    }
    {
      wbc::ParamsX PX;
      wbc::params_derive(P, &PX);
      HIP_TRY(hipMemcpy(h->d_params, &PX, sizeof PX, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemset(h->d_stats, 0, sizeof(StatsDev) * (STAT_SLOTS + 1)));
    HIP_TRY(hipHostMalloc(&h->h_stats, sizeof(StatsDev), hipHostMallocMapped));
    HIP_TRY(hipEventCreate(&h->ev0));
    HIP_TRY(hipEventCreate(&h->ev1));
    if (flags & WBC_HOST_PTRS) {
      size_t nb = (size_t)max_batch;
      HIP_TRY(hipMalloc(&h->s_q, 19 * nb * 8)); HIP_TRY(hipMalloc(&h->s_v, 18 * nb * 8));
      HIP_TRY(hipMalloc(&h->s_tg, 54 * nb * 8)); HIP_TRY(hipMalloc(&h->s_mu, nb * 8));
      HIP_TRY(hipMalloc(&h->s_ms, nb * 8)); HIP_TRY(hipMalloc(&h->s_tau, 12 * nb * 8));
      HIP_TRY(hipMalloc(&h->s_met, 4 * nb * 8)); HIP_TRY(hipMalloc(&h->s_mask, nb));
      HIP_TRY(hipMalloc(&h->s_status, nb * 4));
      HIP_TRY(hipHostMalloc(&h->z_host, ZC_DOUBLES * 8, hipHostMallocMapped));
      HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->z_dev), h->z_host, 0));
    }
    return 0;
  };
  const int rc = build();
  if (rc) { wbc_destroy(h); return rc; }   // g_err keeps the message of the failed call
  *out = h;
  return 0;
}

int wbc_destroy(wbc_handle h) {
  if (!h) return 0;
  wbc::DeviceGuard device_guard_(h->device);   // (a failure here leaves nothing to report to: the frees below are best effort)
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  h->zpend.active = false;   // a small-batch host-pointer tick that was never collected is ABANDONED: destroy writes nothing into caller-owned arrays
                             // (a caller on an error path may already have freed them); wbc_sync before wbc_destroy delivers it
  void* bufs[] = {h->d_model, h->d_params, h->d_stats, h->s_q, h->s_v, h->s_tg, h->s_mu, h->s_ms,
                  h->s_tau, h->s_met, h->s_mask, h->s_status};
  for (void* b : bufs) if (b) (void)hipFree(b);
  if (h->h_stats) (void)hipHostFree(h->h_stats);
  if (h->z_host) (void)hipHostFree(h->z_host);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  for (hipEvent_t e : h->evs) (void)hipEventDestroy(e);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

int wbc_set_stream(wbc_handle h, void* hip_stream) {
  if (!h) return misuse("wbc_set_stream: null handle");
  if (h->zpend.active) {                 // a small-batch host-pointer tick still in flight on the old stream: collect it first
    const int rc = zc_finish(h);
    if (rc) return rc;
  }
  if (h->own_stream) {
    (void)hipStreamSynchronize(h->stream);
    (void)hipStreamDestroy(h->stream);
  }
  h->stream = (hipStream_t)hip_stream;
  h->own_stream = false;
  return 0;
}

// One product kernel family (16 lanes per robot) for every law, batch size and option; the variant number (3) is
// kept in the ABI for compatibility with round-1 callers.
static int pick_variant(const wbc_handle_s*, int) { return 3; }

static int launch(wbc_handle h, int n, int ld, const double* q, const double* v, const double* tg,
                  const uint8_t* mask, const double* mu, const double* ms, double* tau, double* met,
                  int32_t* status) {
  h->last_variant = 3;
  StatsDev* d_stats = h->d_stats;
  dim3 grid((n + HROBOTS - 1) / HROBOTS);
#ifdef WBC_DEV_ONLY   // diagnostic builds (tools/build_cuts.sh): ONE law instantiated, seconds to compile
  if (h->kind != WBC_DEV_ONLY || h->torque_box) return misuse("this diagnostic build carries one law only (WBC_DEV_ONLY)");
  hipLaunchKernelGGL((wbc_hex_kernel<WBC_DEV_ONLY, false>), grid, dim3(HEX_BLOCK), 0, h->stream, h->d_model, h->d_params, n, ld, q, v,
                     tg, mask, mu, ms, tau, met, status, d_stats, h->d_vdot);
  HIP_TRY(hipGetLastError());
  return 0;
#else
#define WBC_HEX_ARGS grid, dim3(HEX_BLOCK), 0, h->stream, h->d_model, h->d_params, n, ld, q, v, tg, mask, mu, ms, tau, met, status, d_stats, h->d_vdot
  if (h->torque_box) {   // second constraint slot per lane: |tau_j| <= tau_max (wbc_hex.hpp)
    switch (h->kind) {
      case WBC_KIND_ID: hipLaunchKernelGGL((wbc_hex_kernel<wbc::KIND_ID, true>), WBC_HEX_ARGS); break;
      case WBC_KIND_MPTC: hipLaunchKernelGGL((wbc_hex_kernel<wbc::KIND_MPTC, true>), WBC_HEX_ARGS); break;
      case WBC_KIND_PC: hipLaunchKernelGGL((wbc_hex_kernel<wbc::KIND_PC, true>), WBC_HEX_ARGS); break;
      default: hipLaunchKernelGGL((wbc_hex_kernel<wbc::KIND_CLF, true>), WBC_HEX_ARGS);
    }
  } else {
    switch (h->kind) {
      case WBC_KIND_ID: hipLaunchKernelGGL((wbc_hex_kernel<wbc::KIND_ID, false>), WBC_HEX_ARGS); break;
      case WBC_KIND_MPTC: hipLaunchKernelGGL((wbc_hex_kernel<wbc::KIND_MPTC, false>), WBC_HEX_ARGS); break;
      case WBC_KIND_PC: hipLaunchKernelGGL((wbc_hex_kernel<wbc::KIND_PC, false>), WBC_HEX_ARGS); break;
      default: hipLaunchKernelGGL((wbc_hex_kernel<wbc::KIND_CLF, false>), WBC_HEX_ARGS);
    }
  }
#undef WBC_HEX_ARGS
  HIP_TRY(hipGetLastError());
  return 0;
#endif
}

static int check_step_args(wbc_handle h, int n, int ld, const void* q, const void* v, const void* tg,
                           const void* mask, const void* tau) {
  if (!h) return misuse("wbc_step: null handle");
  if (n < 0 || n > h->max_batch) return misuse("wbc_step: n out of range (0..max_batch)");
  if (n > 0 && ld < n) return misuse("wbc_step: ld must be >= n");
  if (ld > WBC_MAX_LD) return misuse("wbc_step: ld exceeds WBC_MAX_LD (2^23 instances: the kernels address a row block with 32-bit offsets)");
  if (n > 0 && (!q || !v || !tg || !mask || !tau)) return misuse("wbc_step: q, v, targets, contact_mask and tau are required");
  return 0;
}

int wbc_step(wbc_handle h, int n, int ld, const double* q, const double* v, const double* targets,
             const uint8_t* contact_mask, const double* mu, const double* mass_scale, double* tau,
             double* metrics, int32_t* status) {
  int rc = check_step_args(h, n, ld, q, v, targets, contact_mask, tau);
  if (rc) return rc;
  if (n == 0) return 0;
  WBC_ON_DEVICE(h->device, fail);
  if (!(h->flags & WBC_HOST_PTRS))
    return launch(h, n, ld, q, v, targets, contact_mask, mu, mass_scale, tau, metrics, status);
  rc = zc_finish(h);                    // a small-batch tick whose outputs were never collected: collect them first
  if (rc) return rc;
  if (n <= ZC_MAX) {
    // small batch: inputs written into the mapped block, the kernel works through it, outputs handed over at wbc_sync
    double* z = h->z_host;
    for (int r = 0; r < 19; r++) memcpy(z + (size_t)r * n, q + (size_t)r * ld, (size_t)n * 8);
    for (int r = 0; r < 18; r++) memcpy(z + (size_t)(19 + r) * n, v + (size_t)r * ld, (size_t)n * 8);
    for (int r = 0; r < 54; r++) memcpy(z + (size_t)(37 + r) * n, targets + (size_t)r * ld, (size_t)n * 8);
    if (mu) memcpy(z + (size_t)91 * n, mu, (size_t)n * 8);
    if (mass_scale) memcpy(z + (size_t)92 * n, mass_scale, (size_t)n * 8);
    memcpy(z + ZC_MASK_OFF, contact_mask, (size_t)n);
    double* d = h->z_dev;
    rc = launch(h, n, n, d, d + (size_t)19 * n, d + (size_t)37 * n, reinterpret_cast<const uint8_t*>(d + ZC_MASK_OFF),
                mu ? d + (size_t)91 * n : nullptr, mass_scale ? d + (size_t)92 * n : nullptr, d + (size_t)93 * n,
                metrics ? d + (size_t)105 * n : nullptr, status ? reinterpret_cast<int32_t*>(d + ZC_STATUS_OFF) : nullptr);
    if (rc) return rc;
    h->zpend = {true, n, ld, tau, metrics, status};
    return 0;
  }
  // host pointers: stage rows through the handle's device buffers (leading dimension n on the device)
  hipStream_t s = h->stream;
  HIP_TRY(hipMemcpy2DAsync(h->s_q, (size_t)n * 8, q, (size_t)ld * 8, (size_t)n * 8, 19, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpy2DAsync(h->s_v, (size_t)n * 8, v, (size_t)ld * 8, (size_t)n * 8, 18, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpy2DAsync(h->s_tg, (size_t)n * 8, targets, (size_t)ld * 8, (size_t)n * 8, 54, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(h->s_mask, contact_mask, n, hipMemcpyHostToDevice, s));
  if (mu) HIP_TRY(hipMemcpyAsync(h->s_mu, mu, (size_t)n * 8, hipMemcpyHostToDevice, s));
  if (mass_scale) HIP_TRY(hipMemcpyAsync(h->s_ms, mass_scale, (size_t)n * 8, hipMemcpyHostToDevice, s));
  rc = launch(h, n, n, h->s_q, h->s_v, h->s_tg, h->s_mask, mu ? h->s_mu : nullptr, mass_scale ? h->s_ms : nullptr,
              h->s_tau, metrics ? h->s_met : nullptr, status ? h->s_status : nullptr);
  if (rc) return rc;
  HIP_TRY(hipMemcpy2DAsync(tau, (size_t)ld * 8, h->s_tau, (size_t)n * 8, (size_t)n * 8, 12, hipMemcpyDeviceToHost, s));
  if (metrics) HIP_TRY(hipMemcpy2DAsync(metrics, (size_t)ld * 8, h->s_met, (size_t)n * 8, (size_t)n * 8, 4, hipMemcpyDeviceToHost, s));
  if (status) HIP_TRY(hipMemcpyAsync(status, h->s_status, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  return 0;
}

int wbc_sync(wbc_handle h) {
  if (!h) return misuse("wbc_sync: null handle");
  WBC_ON_DEVICE(h->device, fail);
  HIP_TRY(hipStreamSynchronize(h->stream));
  return zc_finish(h);
}

int wbc_time_steps(wbc_handle h, int steps, int n, int ld, const double* q, const double* v,
                   const double* targets, const uint8_t* contact_mask, const double* mu,
                   const double* mass_scale, double* tau, double* metrics, int32_t* status,
                   float* ms_per_step) {
  int rc = check_step_args(h, n, ld, q, v, targets, contact_mask, tau);
  if (rc) return rc;
  if (steps <= 0) return misuse("wbc_time_steps: steps must be positive");
  if (h->flags & WBC_HOST_PTRS) return misuse("wbc_time_steps: needs a WBC_DEVICE_PTRS handle (inputs resident in HBM)");
  WBC_ON_DEVICE(h->device, fail);
  HIP_TRY(hipEventRecord(h->ev0, h->stream));
  for (int s = 0; s < steps; s++) {
    rc = launch(h, n, ld, q, v, targets, contact_mask, mu, mass_scale, tau, metrics, status);
    if (rc) return rc;
  }
  HIP_TRY(hipEventRecord(h->ev1, h->stream));
  h->timed_steps = steps;
  if (!ms_per_step) return 0;   // asynchronous form: wbc_time_steps_result() reads the events later
  return wbc_time_steps_result(h, ms_per_step);
}

int wbc_time_steps_result(wbc_handle h, float* ms_per_step) {
  if (!h || !ms_per_step) return misuse("wbc_time_steps_result: null argument");
  if (h->timed_steps <= 0) return misuse("wbc_time_steps_result: no wbc_time_steps call to report");
  WBC_ON_DEVICE(h->device, fail);
  HIP_TRY(hipEventSynchronize(h->ev1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *ms_per_step = ms / h->timed_steps;
  return 0;
}

int wbc_time_steps_each(wbc_handle h, int steps, int n, int ld, const double* q, const double* v,
                        const double* targets, const uint8_t* contact_mask, const double* mu,
                        const double* mass_scale, double* tau, double* metrics, int32_t* status,
                        float* ms_each) {
  int rc = check_step_args(h, n, ld, q, v, targets, contact_mask, tau);
  if (rc) return rc;
  if (steps <= 0 || !ms_each) return misuse("wbc_time_steps_each: steps must be positive and ms_each non-null");
  if (h->flags & WBC_HOST_PTRS) return misuse("wbc_time_steps_each: needs a WBC_DEVICE_PTRS handle (inputs resident in HBM)");
  WBC_ON_DEVICE(h->device, fail);
  while ((int)h->evs.size() < steps + 1) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    h->evs.push_back(e);
  }
  HIP_TRY(hipEventRecord(h->evs[0], h->stream));
  for (int s = 0; s < steps; s++) {
    rc = launch(h, n, ld, q, v, targets, contact_mask, mu, mass_scale, tau, metrics, status);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(h->evs[s + 1], h->stream));
  }
  HIP_TRY(hipEventSynchronize(h->evs[steps]));
  for (int s = 0; s < steps; s++) HIP_TRY(hipEventElapsedTime(&ms_each[s], h->evs[s], h->evs[s + 1]));
  return 0;
}

int wbc_stats_get(wbc_handle h, wbc_stats* out) {
  if (!h || !out) return misuse("wbc_stats_get: null argument");
  WBC_ON_DEVICE(h->device, fail);
  StatsDev s;
  StatsDev* dst = nullptr;
  HIP_TRY(hipHostGetDevicePointer((void**)&dst, h->h_stats, 0));
  // stream order: the reduction runs after every launch queued so far -- no wait before it
  hipLaunchKernelGGL(wbc_stats_reduce_kernel, dim3(sizeof(StatsDev) / 8), dim3(256), 0, h->stream, h->d_stats, dst);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));   // the only wait: the totals are already in host memory
  memcpy(&s, h->h_stats, sizeof s);
  out->ticks = 0.0;   // every stepped instance landed in exactly one contact-mask bin: no atomic is spent on the total
  for (int k = 0; k < 16; k++) out->ticks += s.mask_count[k];
  out->status_nonzero = s.status_nonzero; out->iters_sum = s.iters_sum;
  out->tau_abs_sum = s.tau_abs_sum; out->err_sum = s.err_sum;
  double mx; memcpy(&mx, &s.tau_abs_max_bits, 8);
  out->tau_abs_max = mx;
  for (int k = 0; k < 16; k++) out->mask_count[k] = s.mask_count[k];
  return 0;
}

int wbc_stats_pack(wbc_handle h, double* out22) {
  if (!h || !out22) return misuse("wbc_stats_pack: null argument");
  static_assert(sizeof(wbc_stats) == WBC_NSTAT * sizeof(double), "wbc_stats is WBC_NSTAT doubles in declaration order");
  wbc_stats s;
  const int rc = wbc_stats_get(h, &s);
  if (rc) return rc;
  memcpy(out22, &s, sizeof s);
  return 0;
}

int wbc_stats_reduce(const double* gathered, int world, wbc_stats* out) {
  if (!gathered || !out || world < 1) return misuse("wbc_stats_reduce: null argument or world < 1");
  double acc[WBC_NSTAT];
  for (int k = 0; k < WBC_NSTAT; k++) acc[k] = gathered[k];
  constexpr int MAX_FIELD = offsetof(wbc_stats, tau_abs_max) / sizeof(double);
  for (int r = 1; r < world; r++) {
    const double* g = gathered + (size_t)r * WBC_NSTAT;
    for (int k = 0; k < WBC_NSTAT; k++) acc[k] = (k == MAX_FIELD) ? (g[k] > acc[k] ? g[k] : acc[k]) : acc[k] + g[k];
  }
  memcpy(out, acc, sizeof acc);
  return 0;
}

int wbc_stats_reset(wbc_handle h) {
  if (!h) return misuse("wbc_stats_reset: null handle");
  WBC_ON_DEVICE(h->device, fail);
  HIP_TRY(hipMemsetAsync(h->d_stats, 0, sizeof(StatsDev) * (STAT_SLOTS + 1), h->stream));
  return 0;
}

#ifdef WBC_STAMPS
// diagnostic build only: copies the phase stamps of the last launch (16 per block) to the host
int wbc_debug_stamps(unsigned long long* out, int nblocks) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wbc_stamps), sizeof(unsigned long long) * 16 * nblocks));
  return 0;
}
#endif

int wbc_set_vdot_output(wbc_handle h, double* vdot) {
  if (!h) return misuse("wbc_set_vdot_output: null handle");
  if (vdot && (h->flags & WBC_HOST_PTRS)) return misuse("wbc_set_vdot_output: needs a WBC_DEVICE_PTRS handle");
  h->d_vdot = vdot;
  return 0;
}

int wbc_integrate(wbc_handle h, int n, int ld, double dt, double* q, double* v, const double* vdot) {
  if (!h) return misuse("wbc_integrate: null handle");
  if (n < 0 || (n > 0 && (ld < n || !q || !v || !vdot))) return misuse("wbc_integrate: bad arguments");
  if (h->flags & WBC_HOST_PTRS) return misuse("wbc_integrate: needs a WBC_DEVICE_PTRS handle");
  if (n == 0) return 0;
  WBC_ON_DEVICE(h->device, fail);
  hipLaunchKernelGGL(wbc_integrate_kernel, dim3((n + 63) / 64), dim3(64), 0, h->stream, n, ld, dt, q, v, vdot);   // (64-thread workgroups: 64 compute units at N = 4096)
  HIP_TRY(hipGetLastError());
  return 0;
}

int wbc_rollout(wbc_handle h, wbc_traj traj, int steps, double dt, int n, int ld, double* q, double* v,
                double* time, double* targets, uint8_t* contact_mask, const double* mu,
                const double* mass_scale, double* tau, double* metrics, int32_t* status, double* vdot) {
  if (!h || !traj) return misuse("wbc_rollout: null handle");
  if (steps < 0 || !time || !vdot) return misuse("wbc_rollout: steps >= 0, time and vdot are required");
  int rc = check_step_args(h, n, ld, q, v, targets, contact_mask, tau);
  if (rc) return rc;
  if (h->flags & WBC_HOST_PTRS) return misuse("wbc_rollout: needs a WBC_DEVICE_PTRS handle");
  if (steps == 0 || n == 0) return 0;
  WBC_ON_DEVICE(h->device, fail);
#ifdef WBC_DEV_ONLY
  return misuse("wbc_rollout: not part of a WBC_DEV_ONLY diagnostic build");
#else
  {
    // the whole rollout is ONE persistent launch per <= 1024 ticks (wbc_hex_rollout_kernel)
    wbc::TrajDev T;
    if (wbc_traj_raw_(traj, &T)) return misuse("wbc_rollout: bad trajectory handle");
    h->last_variant = 3;
    StatsDev* d_stats = h->d_stats;
    dim3 grid((n + HROBOTS - 1) / HROBOTS);
#define WBC_RO_ARGS grid, dim3(HEX_BLOCK), 0, h->stream, h->d_model, h->d_params, n, ld, steps, dt, T, q, v, time, targets, \
                    contact_mask, mu, mass_scale, tau, metrics, status, d_stats, vdot, (h->warm_start ? 1 : 0)
#define WBC_RO_KIND(TBV)                                                                                   \
    switch (h->kind) {                                                                                     \
      case WBC_KIND_ID: hipLaunchKernelGGL((wbc_hex_rollout_kernel<wbc::KIND_ID, TBV>), WBC_RO_ARGS); break;     \
      case WBC_KIND_MPTC: hipLaunchKernelGGL((wbc_hex_rollout_kernel<wbc::KIND_MPTC, TBV>), WBC_RO_ARGS); break; \
      case WBC_KIND_PC: hipLaunchKernelGGL((wbc_hex_rollout_kernel<wbc::KIND_PC, TBV>), WBC_RO_ARGS); break;     \
      default: hipLaunchKernelGGL((wbc_hex_rollout_kernel<wbc::KIND_CLF, TBV>), WBC_RO_ARGS);                    \
    }
    // at most 1024 ticks per launch (~25 ms): long rollouts stay a sequence of bounded kernels
    const int total = steps;
    for (int done_steps = 0; done_steps < total; done_steps += 1024) {
      steps = (total - done_steps < 1024) ? total - done_steps : 1024;
      if (h->torque_box) { WBC_RO_KIND(true) } else { WBC_RO_KIND(false) }
    }
#undef WBC_RO_KIND
#undef WBC_RO_ARGS
    HIP_TRY(hipGetLastError());
    return 0;
  }
#endif
}

int wbc_set_warm_start(wbc_handle h, int on) {
  if (!h) return misuse("wbc_set_warm_start: null handle");
  h->warm_start = on != 0;
  return 0;
}

int wbc_set_variant(wbc_handle h, int variant) {
  if (!h) return misuse("wbc_set_variant: null handle");
  if (variant != 0 && variant != 3)
    return misuse("wbc_set_variant: 0 = auto or 3 = 16 lanes per robot (the lane- and quad-per-robot kernels of round 1 are retired)");
  h->variant = variant;
  return 0;
}

int wbc_variant_for(wbc_handle h, int n) {
  if (!h || n < 0) return misuse("wbc_variant_for: bad argument");
  return pick_variant(h, n);
}

int wbc_kernel_info(wbc_handle h, int* num_vgpr, int* scratch_bytes, int* lds_bytes, int* block_threads) {
  if (!h) return misuse("wbc_kernel_info: null handle");
  WBC_ON_DEVICE(h->device, fail);
  hipFuncAttributes a;
  const void* fn;
#ifdef WBC_DEV_ONLY
  fn = (const void*)wbc_hex_kernel<WBC_DEV_ONLY, false>;
#else
  if (h->torque_box)
    fn = h->kind == WBC_KIND_ID ? (const void*)wbc_hex_kernel<wbc::KIND_ID, true>
       : h->kind == WBC_KIND_MPTC ? (const void*)wbc_hex_kernel<wbc::KIND_MPTC, true>
       : h->kind == WBC_KIND_PC ? (const void*)wbc_hex_kernel<wbc::KIND_PC, true> : (const void*)wbc_hex_kernel<wbc::KIND_CLF, true>;
  else
    fn = h->kind == WBC_KIND_ID ? (const void*)wbc_hex_kernel<wbc::KIND_ID>
       : h->kind == WBC_KIND_MPTC ? (const void*)wbc_hex_kernel<wbc::KIND_MPTC>
       : h->kind == WBC_KIND_PC ? (const void*)wbc_hex_kernel<wbc::KIND_PC> : (const void*)wbc_hex_kernel<wbc::KIND_CLF>;
#endif
  HIP_TRY(hipFuncGetAttributes(&a, fn));
  if (num_vgpr) *num_vgpr = a.numRegs;
  if (scratch_bytes) *scratch_bytes = (int)a.localSizeBytes;
  if (lds_bytes) *lds_bytes = (int)a.sharedSizeBytes;
  if (block_threads) *block_threads = HEX_BLOCK;
  return 0;
}

int wbc_rollout_kernel_info(wbc_handle h, int* num_vgpr, int* scratch_bytes, int* lds_bytes, int* block_threads) {
  if (!h) return misuse("wbc_rollout_kernel_info: null handle");
  WBC_ON_DEVICE(h->device, fail);
  hipFuncAttributes a;
  const void* fn;
#ifdef WBC_DEV_ONLY
  (void)fn; (void)a;
  return misuse("wbc_rollout_kernel_info: not part of a WBC_DEV_ONLY diagnostic build");
#else
  if (h->torque_box)
    fn = h->kind == WBC_KIND_ID ? (const void*)wbc_hex_rollout_kernel<wbc::KIND_ID, true>
       : h->kind == WBC_KIND_MPTC ? (const void*)wbc_hex_rollout_kernel<wbc::KIND_MPTC, true>
       : h->kind == WBC_KIND_PC ? (const void*)wbc_hex_rollout_kernel<wbc::KIND_PC, true> : (const void*)wbc_hex_rollout_kernel<wbc::KIND_CLF, true>;
  else
    fn = h->kind == WBC_KIND_ID ? (const void*)wbc_hex_rollout_kernel<wbc::KIND_ID, false>
       : h->kind == WBC_KIND_MPTC ? (const void*)wbc_hex_rollout_kernel<wbc::KIND_MPTC, false>
       : h->kind == WBC_KIND_PC ? (const void*)wbc_hex_rollout_kernel<wbc::KIND_PC, false> : (const void*)wbc_hex_rollout_kernel<wbc::KIND_CLF, false>;
#endif
  HIP_TRY(hipFuncGetAttributes(&a, fn));
  if (num_vgpr) *num_vgpr = a.numRegs;
  if (scratch_bytes) *scratch_bytes = (int)a.localSizeBytes;
  if (lds_bytes) *lds_bytes = (int)a.sharedSizeBytes;
  if (block_threads) *block_threads = HEX_BLOCK;
  return 0;
}

}  // extern "C"
