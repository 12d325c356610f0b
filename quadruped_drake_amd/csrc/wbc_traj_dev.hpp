// wbc_traj_dev.hpp -- device-side pieces shared by the stand-alone target lookup (wbc_traj.hip) and the
// persistent closed-loop kernel (wbc_kernels.hip): the stored trunk trajectory as plain device pointers,
// the nearest-sample index of planners/towr.py:92-106, and the semi-implicit Euler step of the rollout.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define WBC_TRAJ_HD __host__ __device__ inline
#else
#define WBC_TRAJ_HD inline
#endif
#include <math.h>
#include <stdint.h>

namespace wbc {

struct TrajDev {
  int K;
  double wait_time;
  const double* ts;        // [K] non-decreasing
  const double* table;     // [K][54]
  const uint8_t* masks;    // [K]
  const double* standing;  // [54]
  uint8_t standing_mask;
};

// Index of the sample np.abs(timestamps - (t - wait_time)).argmin() would return (first index on ties and among
// equal timestamps), or -1 for the standing targets (t < wait_time, or an empty trajectory).
// `hint`: where to start looking (any value; the previous tick's index makes the search O(1) in a rollout).
WBC_TRAJ_HD int traj_index(const TrajDev& T, double t, int hint) {
  if (t < T.wait_time || T.K == 0) return -1;
  t -= T.wait_time;
  const double* ts = T.ts;
  const int K = T.K;
  // bracket the first index with ts[idx] >= t by galloping from the hint, then bisect
  int lo, hi;
  int h = hint < 0 ? 0 : (hint >= K ? K - 1 : hint);
  if (ts[h] < t) {
    int step = 1;
    lo = h + 1;
    hi = K;
    while (lo < K) {
      const int probe = (h + step < K) ? h + step : K - 1;
      if (ts[probe] < t) { lo = probe + 1; if (probe == K - 1) break; step <<= 1; }
      else { hi = probe; break; }
    }
    if (lo > hi) lo = hi;
  } else {
    int step = 1;
    hi = h;
    lo = 0;
    while (hi > 0) {
      const int probe = (h - step > 0) ? h - step : 0;
      if (ts[probe] >= t) { hi = probe; if (probe == 0) break; step <<= 1; }
      else { lo = probe + 1; break; }
    }
  }
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (ts[mid] < t) lo = mid + 1; else hi = mid;
  }
  int c;
  if (lo == 0) c = 0;
  else if (lo == K) c = K - 1;
  else c = (fabs(ts[lo - 1] - t) <= fabs(ts[lo] - t)) ? lo - 1 : lo;
  while (c > 0 && ts[c - 1] == ts[c]) c--;
  return c;
}

}  // namespace wbc
