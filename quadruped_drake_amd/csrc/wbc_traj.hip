// wbc_traj.hip -- callers of the hot path: trunk_state_t wire decode, the device-side
// nearest-timestamp target lookup of planners/towr.py:92-148, and the robot-side wire format of the use_lcm path
// (robot_state_control_lcmt: batched unpack / pack on the device).  Byte/index work: bit-exact against the
// reference's own encoders (tests/golden/trunk_state_*, robot_state_*).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/wbc.h"
#include "../../include/wbc_extras.h"
#include "wbc_traj_dev.hpp"
#include "wbc_device_guard.hpp"

// the one thread-local message buffer behind wbc_last_error() (defined in wbc_kernels.hip)
extern "C" void wbc_set_error_(const char* msg);

namespace {
int tfail(const char* what, hipError_t e) {
  char b[256];
  snprintf(b, sizeof b, "%s: %s", what, hipGetErrorString(e));
  wbc_set_error_(b);
  return -2;
}
int tmisuse(const char* what) { wbc_set_error_(what); return -1; }
#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return tfail(#x, e_); } while (0)

// lcm_types/trunklcm/trunk_state_t.py:123-133: hash = rotl1(0xbd03c56c9649d0b6)
constexpr uint64_t kBase = 0xbd03c56c9649d0b6ull;
constexpr uint64_t kFingerprint = (kBase << 1) + (kBase >> 63);

inline double be_double(const uint8_t* p) {
  uint64_t u = 0;
  for (int i = 0; i < 8; i++) u = (u << 8) | p[i];
  double d;
  memcpy(&d, &u, 8);
  return d;
}
inline const uint8_t* rd3(const uint8_t* p, double* out) {
  for (int i = 0; i < 3; i++) out[i] = be_double(p + 8 * i);
  return p + 24;
}

// One thread per robot: nearest sample (wbc_traj_dev.hpp), then a 54-double gather.
// The search is a chain of DEPENDENT loads (each a trip to L2 or HBM, ~1 us when the table is cold): from the middle of a 5001-sample table it is 13 of
// them and the whole kernel 13.5 us.  traj_index starts from any hint and gallops, so the thread first guesses where its time falls if the samples were
// evenly spaced (two loads every thread shares: the first and the last timestamp); a stored trajectory is evenly spaced or nearly so (planners/towr.py
// streams TOWR's fixed 1 ms grid), the guess is then the answer or next to it, and an unevenly spaced table only costs the gallop it always cost.  The hint
// changes how the index is found, never which (round 6).
__global__ void traj_lookup_kernel(int n, int ld, wbc::TrajDev T, const double* __restrict__ time,
                                   double* __restrict__ targets, uint8_t* __restrict__ contact_mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double ti = time[i];
  int hint = T.K / 2;
  if (T.K > 1) {
    const double t0 = T.ts[0], t1 = T.ts[T.K - 1], x = (ti - T.wait_time - t0) * (double)(T.K - 1);
    // (a NaN or out-of-range guess is clamped inside traj_index; a zero span keeps the middle)
    if (t1 > t0 && x > 0.0) { const double g = x / (t1 - t0); hint = (g < (double)(T.K - 1)) ? (int)(g + 0.5) : T.K - 1; }
    else if (t1 > t0) hint = 0;
  }
  const int c = wbc::traj_index(T, ti, hint);
  const double* src = c < 0 ? T.standing : T.table + (size_t)c * 54;
  const uint8_t mk = c < 0 ? T.standing_mask : T.masks[c];
  for (int r = 0; r < 54; r++) targets[(size_t)r * ld + i] = src[r];
  contact_mask[i] = mk;
}

// ---- robot_state_control_lcmt (lcm_types/cheetahlcm/robot_state_control_lcmt.py:62-72): hash = rotl1(0xbe14089c923ad667)
constexpr uint64_t kRsBase = 0xbe14089c923ad667ull;
constexpr uint64_t kRsFingerprint = (kRsBase << 1) + (kRsBase >> 63);
constexpr int kRsWords = WBC_ROBOT_STATE_BYTES / 4;   // 51: 2 fingerprint words + 19 + 18 + 12 floats

__host__ __device__ inline uint32_t bswap32(uint32_t x) {
  return (x >> 24) | ((x >> 8) & 0xff00u) | ((x << 8) & 0xff0000u) | (x << 24);
}

// thread (i, r): field r (0..36 = q then v) of message i -> SoA row r, column i
__global__ void robot_states_unpack_kernel(int n, int ld, const uint32_t* __restrict__ w, double* __restrict__ q,
                                           double* __restrict__ v, uint8_t* __restrict__ ok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (i >= n) return;
  const uint32_t* m = w + (size_t)i * kRsWords;
  const bool good = bswap32(m[0]) == (uint32_t)(kRsFingerprint >> 32) && bswap32(m[1]) == (uint32_t)kRsFingerprint;
  if (r == 0 && ok) ok[i] = good ? 1 : 0;
  if (!good) return;
  const uint32_t bits = bswap32(m[2 + r]);
  const double x = (double)__uint_as_float(bits);
  if (r < 19) q[(size_t)r * ld + i] = x;
  else v[(size_t)(r - 19) * ld + i] = x;
}

struct RsSrc { int k[12]; };   // message.tau[jd] = tau[k[jd]]

// thread (i, word): word 0/1 fingerprint, 2..38 zeros (q, v), 39..50 torques in the plant's joint order
__global__ void robot_controls_pack_kernel(int n, int ld, const double* __restrict__ tau, RsSrc src, uint32_t* __restrict__ w) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (i >= n) return;
  uint32_t out = 0u;
  if (c == 0) out = (uint32_t)(kRsFingerprint >> 32);
  else if (c == 1) out = (uint32_t)kRsFingerprint;
  else if (c >= 39) out = __float_as_uint((float)tau[(size_t)src.k[c - 39] * ld + i]);   // round to nearest even, like struct.pack('>f')
  w[(size_t)i * kRsWords + c] = bswap32(out);
}

// ---- BasicController.ControlLaw (controllers/basic_controller.py:322-352): joint-space PD, thread (robot i, actuator k)
struct PdArgs { int jd[12]; double qn[12]; };   // plant joint index driven by actuator k; nominal angle of that joint
__global__ void pd_step_kernel(int n, int ld, const double* __restrict__ q, const double* __restrict__ v, PdArgs a, double kp,
                               double kd, double u_max, double* __restrict__ tau) {
#pragma clang fp contract(off)   // the reference's numpy rounds each product and the difference separately: no fused multiply-add here
  const int i = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (i >= n) return;
  const int j = a.jd[k];
  const double qe = q[(size_t)(7 + j) * ld + i] - a.qn[k];
  // -Kp@q_err - Kd@qd_err, then np.clip
  const double p1 = kp * qe, p2 = kd * v[(size_t)(6 + j) * ld + i];
  double u = -p1 - p2;
  u = fmin(fmax(u, -u_max), u_max);
  tau[(size_t)k * ld + i] = u;
}
}  // namespace

struct wbc_traj_s {
  int device, K;
  double wait_time;
  uint8_t standing_mask;
  double *d_ts, *d_table, *d_standing;
  uint8_t* d_masks;
};

extern "C" {

int wbc_trunk_state_decode(const uint8_t* buf, size_t len, wbc_trunk_state* out) {
  if (!buf || !out || len < WBC_TRUNK_STATE_BYTES) return -1;
  uint64_t fp = 0;
  for (int i = 0; i < 8; i++) fp = (fp << 8) | buf[i];
  if (fp != kFingerprint) return -3;
  const uint8_t* p = buf + 8;
  out->timestamp = be_double(p); p += 8;
  out->finished = (*p++) != 0;
  p = rd3(p, out->base_p); p = rd3(p, out->base_pd); p = rd3(p, out->base_pdd);
  p = rd3(p, out->base_rpy); p = rd3(p, out->base_rpyd); p = rd3(p, out->base_rpydd);
  for (int f = 0; f < 4; f++) p = rd3(p, out->foot_p[f]);
  for (int f = 0; f < 4; f++) p = rd3(p, out->foot_pd[f]);
  for (int f = 0; f < 4; f++) p = rd3(p, out->foot_pdd[f]);
  for (int f = 0; f < 4; f++) out->contact[f] = (*p++) != 0;
  for (int f = 0; f < 4; f++) p = rd3(p, out->foot_f[f]);
  return (p - buf) == WBC_TRUNK_STATE_BYTES ? 0 : -1;
}

int wbc_robot_state_decode(const uint8_t* buf, size_t len, wbc_robot_state* out) {
  if (!buf || !out || len < WBC_ROBOT_STATE_BYTES) return -1;
  uint64_t fp = 0;
  for (int i = 0; i < 8; i++) fp = (fp << 8) | buf[i];
  if (fp != kRsFingerprint) return -3;
  float* dst = out->q;   // q[19], v[18], tau[12] are contiguous floats
  static_assert(sizeof(wbc_robot_state) == 49 * sizeof(float), "robot_state layout");
  for (int k = 0; k < 49; k++) {
    const uint8_t* p = buf + 8 + 4 * k;
    const uint32_t u = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
    memcpy(dst + k, &u, 4);
  }
  return 0;
}

int wbc_robot_state_encode(const wbc_robot_state* in, uint8_t* buf, size_t cap) {
  if (!in || !buf || cap < WBC_ROBOT_STATE_BYTES) return -1;
  for (int i = 0; i < 8; i++) buf[i] = (uint8_t)(kRsFingerprint >> (56 - 8 * i));
  const float* src = in->q;
  for (int k = 0; k < 49; k++) {
    uint32_t u;
    memcpy(&u, src + k, 4);
    uint8_t* p = buf + 8 + 4 * k;
    p[0] = (uint8_t)(u >> 24); p[1] = (uint8_t)(u >> 16); p[2] = (uint8_t)(u >> 8); p[3] = (uint8_t)u;
  }
  return WBC_ROBOT_STATE_BYTES;
}

int wbc_robot_states_unpack(int device, void* hip_stream, int n, int ld, const uint8_t* msgs, double* q, double* v,
                            uint8_t* ok) {
  if (n < 0 || ld < n || (n > 0 && (!msgs || !q || !v))) return tmisuse("wbc_robot_states_unpack: bad argument");
  if (((uintptr_t)msgs & 3u) != 0) return tmisuse("wbc_robot_states_unpack: msgs must be 4-byte aligned");
  if (n == 0) return 0;
  WBC_ON_DEVICE(device, tfail);
  hipLaunchKernelGGL(robot_states_unpack_kernel, dim3((n + 255) / 256, 37), dim3(256), 0, (hipStream_t)hip_stream, n, ld,
                     reinterpret_cast<const uint32_t*>(msgs), q, v, ok);
  HIP_TRY(hipGetLastError());
  return 0;
}

int wbc_robot_controls_pack(int device, void* hip_stream, int n, int ld, const double* tau, const int* q_perm,
                            const int* act_perm, uint8_t* msgs) {
  if (n < 0 || ld < n || (n > 0 && (!tau || !msgs))) return tmisuse("wbc_robot_controls_pack: bad argument");
  if (((uintptr_t)msgs & 3u) != 0) return tmisuse("wbc_robot_controls_pack: msgs must be 4-byte aligned");
  RsSrc src;
  bool seen[12] = {false};
  for (int k = 0; k < 12; k++) {
    const int j = act_perm ? act_perm[k] : k;                       // canonical joint driven by actuator k
    if (j < 0 || j >= 12) return tmisuse("wbc_robot_controls_pack: act_perm must be a permutation of 0..11");
    const int jd = q_perm ? q_perm[j] : j;                          // its index in the plant's joint order
    if (jd < 0 || jd >= 12 || seen[jd]) return tmisuse("wbc_robot_controls_pack: q_perm / act_perm must be permutations of 0..11");
    seen[jd] = true;
    src.k[jd] = k;
  }
  if (n == 0) return 0;
  WBC_ON_DEVICE(device, tfail);
  hipLaunchKernelGGL(robot_controls_pack_kernel, dim3((n + 255) / 256, kRsWords), dim3(256), 0, (hipStream_t)hip_stream, n, ld,
                     tau, src, reinterpret_cast<uint32_t*>(msgs));
  HIP_TRY(hipGetLastError());
  return 0;
}

int wbc_pd_step(int device, void* hip_stream, int n, int ld, const double* q, const double* v, const double* q_nom19,
                double kp, double kd, double u_max, const int* q_perm, const int* act_perm, double* tau) {
  if (n < 0 || ld < n || (n > 0 && (!q || !v || !tau))) return tmisuse("wbc_pd_step: bad argument");
  if (!(u_max >= 0.0)) return tmisuse("wbc_pd_step: u_max must be non-negative");
  static const double kNominal[3] = {0.0, -0.8, 1.6};   // basic_controller.py:333-340
  PdArgs a;
  bool seen[12] = {false};
  for (int k = 0; k < 12; k++) {
    const int j = act_perm ? act_perm[k] : k;
    if (j < 0 || j >= 12) return tmisuse("wbc_pd_step: act_perm must be a permutation of 0..11");
    const int jd = q_perm ? q_perm[j] : j;
    if (jd < 0 || jd >= 12 || seen[jd]) return tmisuse("wbc_pd_step: q_perm / act_perm must be permutations of 0..11");
    seen[jd] = true;
    a.jd[k] = jd;
    a.qn[k] = q_nom19 ? q_nom19[7 + jd] : kNominal[jd % 3];
  }
  if (n == 0) return 0;
  WBC_ON_DEVICE(device, tfail);
  hipLaunchKernelGGL(pd_step_kernel, dim3((n + 255) / 256, 12), dim3(256), 0, (hipStream_t)hip_stream, n, ld, q, v, a, kp, kd, u_max, tau);
  HIP_TRY(hipGetLastError());
  return 0;
}

int wbc_trunk_state_to_targets(const wbc_trunk_state* s, double* t, uint8_t* contact_mask) {
  if (!s || !t || !contact_mask) return -1;
  const double* body[6] = {s->base_p, s->base_pd, s->base_pdd, s->base_rpy, s->base_rpyd, s->base_rpydd};
  for (int k = 0; k < 6; k++) for (int i = 0; i < 3; i++) t[3 * k + i] = body[k][i];
  uint8_t m = 0;
  for (int f = 0; f < 4; f++) {
    for (int i = 0; i < 3; i++) {
      t[18 + 9 * f + i] = s->foot_p[f][i];
      t[21 + 9 * f + i] = s->foot_pd[f][i];
      t[24 + 9 * f + i] = s->foot_pdd[f][i];
    }
    if (s->contact[f]) m |= (uint8_t)(1u << f);
  }
  *contact_mask = m;
  return 0;
}

int wbc_traj_create(int device, int K, const double* timestamps, const double* targets, const uint8_t* masks,
                    const double* standing_targets54, uint8_t standing_mask, double wait_time, wbc_traj* out) {
  if (!out || K < 0 || (K > 0 && (!timestamps || !targets || !masks)) || !standing_targets54)
    return tmisuse("wbc_traj_create: bad argument");
  for (int i = 1; i < K; i++)
    if (!(timestamps[i] >= timestamps[i - 1])) return tmisuse("wbc_traj_create: timestamps must be non-decreasing");
  WBC_ON_DEVICE(device, tfail);
  wbc_traj t = new wbc_traj_s();
  memset(t, 0, sizeof *t);
  t->device = device; t->K = K; t->wait_time = wait_time; t->standing_mask = standing_mask;
  const size_t kk = K > 0 ? K : 1;
  // any failure below releases what was already allocated (hipFree(nullptr) is a no-op)
  auto build = [&]() -> int {
    HIP_TRY(hipMalloc(&t->d_ts, kk * 8));
    HIP_TRY(hipMalloc(&t->d_table, kk * 54 * 8));
    HIP_TRY(hipMalloc(&t->d_masks, kk));
    HIP_TRY(hipMalloc(&t->d_standing, 54 * 8));
    if (K > 0) {
      HIP_TRY(hipMemcpy(t->d_ts, timestamps, (size_t)K * 8, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(t->d_table, targets, (size_t)K * 54 * 8, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(t->d_masks, masks, (size_t)K, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(t->d_standing, standing_targets54, 54 * 8, hipMemcpyHostToDevice));
    return 0;
  };
  const int rc = build();
  if (rc) { wbc_traj_destroy(t); return rc; }
  *out = t;
  return 0;
}

int wbc_traj_destroy(wbc_traj t) {
  if (!t) return 0;
  wbc::DeviceGuard device_guard_(t->device);
  (void)hipFree(t->d_ts); (void)hipFree(t->d_table); (void)hipFree(t->d_masks); (void)hipFree(t->d_standing);
  delete t;
  return 0;
}

int wbc_traj_lookup(wbc_traj t, void* hip_stream, int n, int ld, const double* time, double* targets,
                    uint8_t* contact_mask) {
  if (!t || n < 0 || (n > 0 && (ld < n || !time || !targets || !contact_mask))) return tmisuse("wbc_traj_lookup: bad argument");
  if (n == 0) return 0;
  WBC_ON_DEVICE(t->device, tfail);
  wbc::TrajDev T{t->K, t->wait_time, t->d_ts, t->d_table, t->d_masks, t->d_standing, t->standing_mask};
  // 64-thread workgroups: a batch of 4096 robots spreads over 64 compute units instead of 16 (the kernel is a latency chain, not a throughput problem)
  hipLaunchKernelGGL(traj_lookup_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)hip_stream, n, ld, T, time,
                     targets, contact_mask);
  HIP_TRY(hipGetLastError());
  return 0;
}

// internal (not in include/wbc.h): the trajectory's device pointers for the persistent rollout kernel
int wbc_traj_raw_(wbc_traj t, wbc::TrajDev* out) {
  if (!t || !out) return -1;
  *out = wbc::TrajDev{t->K, t->wait_time, t->d_ts, t->d_table, t->d_masks, t->d_standing, t->standing_mask};
  return 0;
}

}  // extern "C"
