// wbc_tick.hpp -- shared per-robot math of the whole-body-QP tick (product code): scalar helpers, the model / parameter
// PODs, per-leg forward kinematics, Newton-Euler and composite-inertia passes in world-aligned coordinates about body-frame
// origins, the RPY handling, status codes.  The product kernel that uses it is the 16-lanes-per-robot tick of wbc_hex.hpp.
// Templated on the scalar so that tools/ can instantiate it on the host (tests, flop counting); the shipped C-ABI library
// (include/wbc.h) only ever runs it on the device.  (The retired one-lane-per-robot tick lives in tools/wbc_scalar_tick.hpp.)
//
// Reference path (what this replaces, per tick):
//   controllers/basic_controller.py:101-115,173-220,246-269   M, Cv, tau_g, foot/body J, Jdv, Jd
//   controllers/inverse_dynamics_controller.py:103-234        ID QP
//   controllers/mptc_controller.py:125-310                    MPTC QP
//
// Formulation (all laws): the 18 dynamics rows and 3 nc contact rows are eliminated exactly into 12 reduced variables
// with only the 4 nc friction rows left as inequalities (DESIGN.md "Reduced QP" derives it).
// The QP is solved by a Goldfarb-Idnani dual active set on a QR factor of the square-root
// form, with the same 1/2*eps2*|[tau; f]|^2 tie-break as the oracle.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define WBC_HD __host__ __device__ __forceinline__
#define WBC_HDN __host__ __device__ __noinline__
#else
#define WBC_HD inline
#define WBC_HDN
#endif

// Diagnostic builds only (-DWBC_STAMPS): shader-clock stamps at phase boundaries, written by lane 0
// of each block to a side buffer that nothing else reads.  Expands to nothing in the product build.
#if defined(WBC_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#ifdef WBC_STAMPS_GI   // slots 10..15 carry the active set's accumulated section times instead of the tick's phase stamps
#define WBC_STAMP_OK(i) ((i) < 10)
#else
#define WBC_STAMP_OK(i) true
#endif
#define WBC_STAMP(i)                                                                                  \
  do {                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    unsigned long long t_;                                                                            \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                        \
    if (WBC_STAMP_OK(i) && threadIdx.x == 0) g_wbc_stamps[(size_t)blockIdx.x * 16 + (i)] = t_;       \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  } while (0)
#else
#define WBC_STAMP(i) do { } while (0)
#endif
// Diagnostic builds only (-DWBC_STAMPS -DWBC_STAMPS_GI): shader cycles of the sections of the active set's generic trips,
// accumulated over a tick's trips (section i = code between timer i-1 and timer i) -> stamp slots 10 + i
#if defined(WBC_STAMPS_GI) && defined(__HIP_DEVICE_COMPILE__)
#define WBC_GI_TIMERS unsigned long long gt_prev_ = 0, gt_acc_[6] = {0, 0, 0, 0, 0, 0}
#define WBC_GI_T0()                                                                                   \
  do {                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(gt_prev_)::"memory");                  \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  } while (0)
#define WBC_GI_T(i)                                                                                   \
  do {                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    unsigned long long t_;                                                                            \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                        \
    gt_acc_[i] += t_ - gt_prev_;                                                                      \
    gt_prev_ = t_;                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  } while (0)
#define WBC_GI_TEND()                                                                                 \
  do {                                                                                                \
    if (threadIdx.x == 0)                                                                             \
      for (int i_ = 0; i_ < 6; i_++) g_wbc_stamps[(size_t)blockIdx.x * 16 + 10 + i_] = gt_acc_[i_];  \
  } while (0)
#else
#define WBC_GI_TIMERS do { } while (0)
#define WBC_GI_T0() do { } while (0)
#define WBC_GI_T(i) do { } while (0)
#define WBC_GI_TEND() do { } while (0)
#endif

namespace wbc {

// 1/x and sqrt(x) for well-scaled positive-magnitude arguments (pivots, norms): hardware seed + Newton steps
// instead of the IEEE division / sqrt expansions (~15 instructions each with scaling and fix-up, which
// only matter for denormal or huge inputs).  Accurate to ~1 ulp; 0, inf and NaN propagate to inf/NaN
// results that every caller discards through a select.  Host: the exact operations.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
}
__device__ __forceinline__ double fast_sqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);          // ~2^-26 relative
  double g = x * y, hh = 0.5 * y;               // g ~ sqrt(x), hh ~ 1/(2 sqrt(x))
  double e = __builtin_fma(-hh, g, 0.5);
  g = __builtin_fma(g, e, g);
  hh = __builtin_fma(hh, e, hh);
  e = __builtin_fma(-g, g, x);                  // residual
  g = __builtin_fma(e, hh, g);
  return (x > 0.0) ? g : 0.0;
}
// sqrt(x) and 1/sqrt(x) of a positive x from ONE seed: the refinement of the root carries h ~ 1/(2 sqrt(x)) along anyway
__device__ __forceinline__ void fast_sqrt_rsq(double x, double& root, double& rs) {
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y, hh = 0.5 * y;
  double e = __builtin_fma(-hh, g, 0.5);
  g = __builtin_fma(g, e, g);
  hh = __builtin_fma(hh, e, hh);
  e = __builtin_fma(-g, g, x);
  g = __builtin_fma(e, hh, g);
  e = __builtin_fma(-hh, g, 0.5);          // one more step for h: its error was still 2^-50
  hh = __builtin_fma(hh, e, hh);
  root = (x > 0.0) ? g : 0.0;
  rs = hh + hh;
}
#else
WBC_HD double fast_rcp(double x) { return 1.0 / x; }
WBC_HD double fast_sqrt(double x) { return sqrt(x); }
WBC_HD void fast_sqrt_rsq(double x, double& root, double& rs) { root = sqrt(x); rs = 1.0 / root; }
#endif

// one range reduction for both (the f64 sin/cos are long software routines on the GPU)
WBC_HD void wbc_sincos(double x, double& s, double& c) {
#if defined(__HIP_DEVICE_COMPILE__)
  sincos(x, &s, &c);
#else
  s = sin(x); c = cos(x);
#endif
}
template <class T> WBC_HD void wbc_sincos(const T& x, T& s, T& c) { s = sin(x); c = cos(x); }

// Straight knee: the reduced formulation inverts every leg's 3x3 foot Jacobian, |det J_leg| ~ 0.04 |sin(knee)|, which the
// reference's full QP does not need.  Measured (profiles/r02/singular_envelope.md): torques agree with the dense oracle to
// 1e-8 down to a knee angle of 1e-8 rad, and for a SWING leg the oracle's solution at the exactly straight knee is the
// limit of those (6e-7), so a swing leg of the ID / CLF laws with an (exactly or nearly) straight knee is evaluated AT
// |sin(knee)| = 1e-8 -- a joint-angle change far below any encoder's resolution -- instead of being reported as singular.
// Not done for a CONTACT leg (there the full QP's solution is discontinuous at the singular point: the contact rows lose
// rank; status 2 stays) nor for MPTC / PC (Lambda = (J M^-1 J')^-1 is singular in the reference itself).
template <class T> WBC_HD void knee_clamp(T& sn, const T& cs) {
  if (cs > T(0.0) && sn < T(1e-8) && sn > T(-1e-8)) sn = (sn < T(0.0)) ? T(-1e-8) : T(1e-8);
}

enum { KIND_ID = 0, KIND_MPTC = 1, KIND_PC = 2, KIND_CLF = 3 };
// PC = MPTC + the passivity row Vdot <= 0 (pc_controller.py); CLF = ID-type cost with LQR feedback, a
// slack delta (13th reduced variable) and the CLF row (clf_controller.py) -- lane kernel only.
enum { ST_OK = 0, ST_ITER = 1, ST_SINGULAR = 2, ST_ILLCOND = 3 };
// MPTC / PC: the law itself inverts J M^-1 J' (mptc_controller.py:237-238), which loses rank with a straight knee.  Measured
// (profiles/r02/singular_envelope.md): below |sin(knee)| ~ 1e-5 kernel and dense restatement disagree by 20 % ... 300 % with
// both solvers reporting success -- there is no trustworthy answer.  A tick of these laws with |sin(knee)| < 1e-4 on any leg
// is therefore solved as usual (torques and accelerations written) but REPORTED: status 3 "ill-conditioned".
constexpr double KNEE_ILLCOND = 1e-4;
// Malformed instances (include/wbc.h): the tick reports -- status 2, zero torques and accelerations -- what it cannot answer with finite numbers, where the
// reference would assert (inverse_dynamics_controller.py:224) or hand NaN on.  The test is on the OUTPUT side (hex_tick's last stage): every value the
// law reads reaches the torques through arithmetic, so one look at them finds a NaN / inf input, and an overflow on the way, at a dozen instructions.
// !(|x| < inf) is true for NaN and for +-inf.
WBC_HD bool not_finite(double x) { return !(fabs(x) < __builtin_huge_val()); }

struct LinkC {
  double off[3];  // joint origin in the parent link frame
  int axis;       // 0,1,2
  double sgn;     // +-1
  double mass;
  double mc[3];   // mass * com (first moment, link frame)
  double I[6];    // about the link origin, link frame: xx yy zz xy xz yz
  double axv[3];  // sgn * e_axis: the same joint axis as a vector (select-free kinematics, leg_fk_vec)
};
struct ModelC {
  double base_mass, base_mc[3], base_I[6];
  LinkC link[4][3];
  double foot_off[4][3];
  double gravity;
  int q_perm[12];    // canonical joint j is read from input row 7 + q_perm[j] / 6 + q_perm[j]
  int act_perm[12];  // output row k (actuator k) = canonical joint act_perm[k]
  int act_inv[12];   // canonical joint j is written to output row act_inv[j]
};
struct ParamsC {
  double Kp_body_p, Kd_body_p, Kp_body_rpy, Kd_body_rpy, Kp_foot, Kd_foot;
  double w_body, w_foot, mu, Kd_contact, tau_max, eps2;
};
// The kernel's view: the public POD plus the three square roots every robot of every tick would otherwise form from it
// (correctly rounded on the host as on the device: the same bits, ~60 instructions per tick less).
struct ParamsX : ParamsC {
  double sq_eps, sq_w_body, sq_w_foot;
  double one;   // 1.0: what a kernel reads in place of a per-instance mass scale that was not given (a load from here instead of a branch around the load)
};
inline void params_derive(const ParamsC& p, ParamsX* x) {
  static_cast<ParamsC&>(*x) = p;
  x->sq_eps = sqrt(p.eps2); x->sq_w_body = sqrt(p.w_body); x->sq_w_foot = sqrt(p.w_foot);
  x->one = 1.0;
}

// ---------------------------------------------------------------- tiny vector helpers
template <class T> WBC_HD void cross(const T* a, const T* b, T* c) {
  T x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  c[0] = x; c[1] = y; c[2] = z;
}
template <class T> WBC_HD T dot(const T* a, const T* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
// y = S x for symmetric S stored [xx yy zz xy xz yz]
template <class T> WBC_HD void symv(const T* S, const T* x, T* y) {
  T a = S[0] * x[0] + S[3] * x[1] + S[4] * x[2];
  T b = S[3] * x[0] + S[1] * x[1] + S[5] * x[2];
  T c = S[4] * x[0] + S[5] * x[1] + S[2] * x[2];
  y[0] = a; y[1] = b; y[2] = c;
}
template <class T> WBC_HD void mm3(const T* A, const T* B, T* C);
// y = R x, R row-major 3x3
template <class T> WBC_HD void rotv(const T* R, const T* x, T* y) {
  T a = R[0] * x[0] + R[1] * x[1] + R[2] * x[2];
  T b = R[3] * x[0] + R[4] * x[1] + R[5] * x[2];
  T c = R[6] * x[0] + R[7] * x[1] + R[8] * x[2];
  y[0] = a; y[1] = b; y[2] = c;
}
// Iw = R I R' for symmetric I (6) -> symmetric (6)
template <class T> WBC_HD void rot_inertia(const T* R, const double* I6, T* out) {
  T I[9] = {T(I6[0]), T(I6[3]), T(I6[4]), T(I6[3]), T(I6[1]), T(I6[5]), T(I6[4]), T(I6[5]), T(I6[2])};
  T A[9];  // A = R I
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A[3 * i + j] = R[3 * i] * I[j] + R[3 * i + 1] * I[3 + j] + R[3 * i + 2] * I[6 + j];
  out[0] = A[0] * R[0] + A[1] * R[1] + A[2] * R[2];
  out[1] = A[3] * R[3] + A[4] * R[4] + A[5] * R[5];
  out[2] = A[6] * R[6] + A[7] * R[7] + A[8] * R[8];
  out[3] = A[0] * R[3] + A[1] * R[4] + A[2] * R[5];
  out[4] = A[0] * R[6] + A[1] * R[7] + A[2] * R[8];
  out[5] = A[3] * R[6] + A[4] * R[7] + A[5] * R[8];
}
// Shift a composite inertia (mass m, first moment h, inertia I about its origin O1) to a new
// origin O0 with O1 = O0 + r, accumulating into (M, H, J).
template <class T> WBC_HD void shift_add(const T& m, const T* h, const T* I, const T* r, T& M, T* H, T* J) {
  T hr = h[0] * r[0] + h[1] * r[1] + h[2] * r[2];
  T rr = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  T d = T(2.0) * hr + m * rr;
  M = M + m;
  for (int i = 0; i < 3; i++) H[i] = H[i] + h[i] + m * r[i];
  J[0] = J[0] + I[0] + d - (T(2.0) * h[0] * r[0] + m * r[0] * r[0]);
  J[1] = J[1] + I[1] + d - (T(2.0) * h[1] * r[1] + m * r[1] * r[1]);
  J[2] = J[2] + I[2] + d - (T(2.0) * h[2] * r[2] + m * r[2] * r[2]);
  J[3] = J[3] + I[3] - (h[0] * r[1] + r[0] * h[1] + m * r[0] * r[1]);
  J[4] = J[4] + I[4] - (h[0] * r[2] + r[0] * h[2] + m * r[0] * r[2]);
  J[5] = J[5] + I[5] - (h[1] * r[2] + r[1] * h[2] + m * r[1] * r[2]);
}

// ---------------------------------------------------------------- per-leg data
// Per-leg kinematics cache, world-aligned, positions relative to the base origin.  Accessed only
// through the accessors below so that the storage can be registers (LegKin) or a lane-strided LDS
// arena (LegKinLds in wbc_kernels.hip): r = link origins, ax = joint axes, mcw = first moments
// m*c, Iw = inertia about the link origin (xx yy zz xy xz yz), rf = foot position.
enum { KIN_R = 0, KIN_AX = 9, KIN_MCW = 18, KIN_IW = 27, KIN_RF = 45, KIN_N = 48 };
template <class T> struct LegKin {
  T d[KIN_N];
  WBC_HD T& r(int k, int i) { return d[KIN_R + 3 * k + i]; }
  WBC_HD T& ax(int k, int i) { return d[KIN_AX + 3 * k + i]; }
  WBC_HD T& mcw(int k, int i) { return d[KIN_MCW + 3 * k + i]; }
  WBC_HD T& Iw(int k, int i) { return d[KIN_IW + 6 * k + i]; }
  WBC_HD T& rf(int i) { return d[KIN_RF + i]; }
  WBC_HD const T& r(int k, int i) const { return d[KIN_R + 3 * k + i]; }
  WBC_HD const T& ax(int k, int i) const { return d[KIN_AX + 3 * k + i]; }
  WBC_HD const T& mcw(int k, int i) const { return d[KIN_MCW + 3 * k + i]; }
  WBC_HD const T& Iw(int k, int i) const { return d[KIN_IW + 6 * k + i]; }
  WBC_HD const T& rf(int i) const { return d[KIN_RF + i]; }
};
template <class T> struct LegDyn {
  T Jl[9];    // d(foot velocity)/d(own joint rates), row-major 3x3
  T Ji[9];    // Jl^-1
  T Jdv[3];   // bias acceleration of the foot
  T pd[3];    // foot velocity (world)
  T rd[3];    // foot velocity relative to the base origin (for Jd)
  T Jd[9];    // time derivative of Jl (MPTC, swing legs)
  T Mbl[18];  // 6x3: rows [angular(3); linear(3)] of the base, columns own joints
  T Mll[6];   // symmetric 3x3
  T hl[3];    // bias + gravity on own joints
};

// Forward kinematics of one leg from cached sines/cosines.
template <class T, class KinT>
WBC_HD void leg_fk(const ModelC& m, int l, const T* R0, const T* sn, const T* cs, KinT& K) {
  T R[9];
  for (int i = 0; i < 9; i++) R[i] = R0[i];
  T p[3] = {T(0.0), T(0.0), T(0.0)};
  for (int k = 0; k < 3; k++) {
    const LinkC& L = m.link[l][k];
    T off[3] = {T(L.off[0]), T(L.off[1]), T(L.off[2])}, t[3];
    rotv(R, off, t);
    for (int i = 0; i < 3; i++) { p[i] = p[i] + t[i]; K.r(k, i) = p[i]; }
    int a = L.axis, b = (a + 1) % 3, c = (a + 2) % 3;
    T s = T(L.sgn) * sn[k], co = cs[k];
    for (int i = 0; i < 3; i++) {
      K.ax(k, i) = T(L.sgn) * R[3 * i + a];
      // R <- R * Rot(axis a, angle): columns b, c mix
      T cb = R[3 * i + b], cc = R[3 * i + c];
      R[3 * i + b] = cb * co + cc * s;
      R[3 * i + c] = cc * co - cb * s;
    }
    T mc[3] = {T(L.mc[0]), T(L.mc[1]), T(L.mc[2])}, mcw[3], Iw[6];
    rotv(R, mc, mcw);
    rot_inertia(R, L.I, Iw);
    for (int i = 0; i < 3; i++) K.mcw(k, i) = mcw[i];
    for (int i = 0; i < 6; i++) K.Iw(k, i) = Iw[i];
  }
  T fo[3] = {T(m.foot_off[l][0]), T(m.foot_off[l][1]), T(m.foot_off[l][2])}, t[3];
  rotv(R, fo, t);
  for (int i = 0; i < 3; i++) K.rf(i) = p[i] + t[i];
}

// The same forward kinematics without register-array indexing by the (run-time) axis number: the joint
// rotation is applied as a 3x3 product with Rl = diag(a_i^2 + c (1 - a_i^2)) + s [a]x  (a = +-e_axis,
// so the diagonal is exactly 1 or c).  leg_fk's R[3*i + axis] accesses compile to select trees on the
// GPU (~800 v_cndmask per leg, profiles/r02); this form is ~120 plain FMAs.
template <class T, class KinT>
WBC_HD void leg_fk_vec(const ModelC& m, int l, const T* R0, const T* sn, const T* cs, KinT& K) {
  T R[9];
  for (int i = 0; i < 9; i++) R[i] = R0[i];
  T p[3] = {T(0.0), T(0.0), T(0.0)};
  for (int k = 0; k < 3; k++) {
    const LinkC& L = m.link[l][k];
    T off[3] = {T(L.off[0]), T(L.off[1]), T(L.off[2])}, t[3];
    rotv(R, off, t);
    for (int i = 0; i < 3; i++) { p[i] = p[i] + t[i]; K.r(k, i) = p[i]; }
    const T a0 = T(L.axv[0]), a1 = T(L.axv[1]), a2 = T(L.axv[2]);
    for (int i = 0; i < 3; i++) K.ax(k, i) = R[3 * i] * a0 + R[3 * i + 1] * a1 + R[3 * i + 2] * a2;
    const T s = sn[k], c = cs[k];
    const T q0 = a0 * a0, q1 = a1 * a1, q2 = a2 * a2;
    T Rl[9];
    Rl[0] = q0 + c * (T(1.0) - q0); Rl[4] = q1 + c * (T(1.0) - q1); Rl[8] = q2 + c * (T(1.0) - q2);
    Rl[1] = T(0.0) - s * a2; Rl[2] = s * a1;
    Rl[3] = s * a2;          Rl[5] = T(0.0) - s * a0;
    Rl[6] = T(0.0) - s * a1; Rl[7] = s * a0;
    T Rn[9];
    mm3(R, Rl, Rn);
    for (int i = 0; i < 9; i++) R[i] = Rn[i];
    T mc[3] = {T(L.mc[0]), T(L.mc[1]), T(L.mc[2])}, mcw[3], Iw[6];
    rotv(R, mc, mcw);
    rot_inertia(R, L.I, Iw);
    for (int i = 0; i < 3; i++) K.mcw(k, i) = mcw[i];
    for (int i = 0; i < 6; i++) K.Iw(k, i) = Iw[i];
  }
  T fo[3] = {T(m.foot_off[l][0]), T(m.foot_off[l][1]), T(m.foot_off[l][2])}, t[3];
  rotv(R, fo, t);
  for (int i = 0; i < 3; i++) K.rf(i) = p[i] + t[i];
}

// Forward kinematics specialised to the joint-axis pattern of both supported trees -- abduction about +-x, hip and
// knee about +-y (Mini Cheetah: +x -y -y, ANYmal: +x +y +y; checked by wbc_create).  The joint rotation then mixes
// just two columns of R (12 flops instead of a 3x3 product) and the joint axis is +- a column of R, with
// compile-time column numbers: no select trees (leg_fk) and no generic products (leg_fk_vec).  Same results as
// leg_fk to rounding.
template <class T, class KinT>
WBC_HD void leg_fk_xyy(const ModelC& m, int l, const T* R0, const T* sn, const T* cs, KinT& K) {
  T R[9];
  for (int i = 0; i < 9; i++) R[i] = R0[i];
  T p[3] = {T(0.0), T(0.0), T(0.0)};
  for (int k = 0; k < 3; k++) {
    const LinkC& L = m.link[l][k];
    T off[3] = {T(L.off[0]), T(L.off[1]), T(L.off[2])}, t[3];
    rotv(R, off, t);
    for (int i = 0; i < 3; i++) { p[i] = p[i] + t[i]; K.r(k, i) = p[i]; }
    const T sg = T(L.sgn), s = sg * sn[k], c = cs[k];
    const int a = (k == 0) ? 0 : 1, b = (k == 0) ? 1 : 2, cc = (k == 0) ? 2 : 0;   // axis column and the two it mixes: (a, b, cc) cyclic
    for (int i = 0; i < 3; i++) {
      K.ax(k, i) = sg * R[3 * i + a];
      const T cb = R[3 * i + b], ccv = R[3 * i + cc];
      R[3 * i + b] = cb * c + ccv * s;
      R[3 * i + cc] = ccv * c - cb * s;
    }
    T mc[3] = {T(L.mc[0]), T(L.mc[1]), T(L.mc[2])}, mcw[3], Iw[6];
    rotv(R, mc, mcw);
    rot_inertia(R, L.I, Iw);
    for (int i = 0; i < 3; i++) K.mcw(k, i) = mcw[i];
    for (int i = 0; i < 6; i++) K.Iw(k, i) = Iw[i];
  }
  T fo[3] = {T(m.foot_off[l][0]), T(m.foot_off[l][1]), T(m.foot_off[l][2])}, t[3];
  rotv(R, fo, t);
  for (int i = 0; i < 3; i++) K.rf(i) = p[i] + t[i];
}

// Newton-Euler bias pass for one leg (vd = 0): joint torques hl and the leg's reaction wrench
// (Nb about the base origin, Fb).  w0 = base angular velocity, qd = own joint rates, gz = gravity.
// Optionally also returns the foot bias acceleration / velocities / Jd columns.
// mass3 = the three link masses of this leg, loaded ONCE by the caller: a per-lane indexed model
// load inside every pass is a full global-memory round trip at one wavefront per SIMD.
template <class T, bool KINEXTRA, class KinT>
WBC_HD void leg_rnea(const T* mass3, const KinT& K, const T* w0, const T* qd, T gz, T* hl, T* Nb,
                     T* Fb, LegDyn<T>* D) {
  T w[3] = {w0[0], w0[1], w0[2]};
  T al[3] = {T(0.0), T(0.0), T(0.0)};
  T a[3] = {T(0.0), T(0.0), T(0.0)};   // origin acceleration minus base-origin acceleration (=0)
  T vo[3] = {T(0.0), T(0.0), T(0.0)};  // origin velocity relative to the base origin velocity
  T F[3][3], N[3][3];
  T rk[3][3], axk[3][3];
  T rfv[3] = {K.rf(0), K.rf(1), K.rf(2)};
  for (int k = 0; k < 3; k++) {
    for (int i = 0; i < 3; i++) { rk[k][i] = K.r(k, i); axk[k][i] = K.ax(k, i); }
    T r[3];
    for (int i = 0; i < 3; i++) r[i] = rk[k][i] - (k ? rk[k - 1][i] : T(0.0));
    T wxr[3], t[3], alxr[3], wxa[3];
    cross(w, r, wxr);
    cross(w, wxr, t);
    cross(al, r, alxr);
    cross(w, axk[k], wxa);
    for (int i = 0; i < 3; i++) {
      a[i] = a[i] + alxr[i] + t[i];
      vo[i] = vo[i] + wxr[i];
      al[i] = al[i] + wxa[i] * qd[k];
    }
    if (KINEXTRA) {
      // Jd column k = (w_parent x a_k) x (rf - r_k) + a_k x (vf - v_k); the a_k x vf part is added below
      T d[3] = {rfv[0] - rk[k][0], rfv[1] - rk[k][1], rfv[2] - rk[k][2]}, c1[3], c2[3];
      cross(wxa, d, c1);
      cross(axk[k], vo, c2);
      for (int i = 0; i < 3; i++) D->Jd[3 * i + k] = c1[i] - c2[i];
    }
    for (int i = 0; i < 3; i++) w[i] = w[i] + axk[k][i] * qd[k];
    // body wrench about the link origin
    T mcw[3] = {K.mcw(k, 0), K.mcw(k, 1), K.mcw(k, 2)};
    T Iw[6] = {K.Iw(k, 0), K.Iw(k, 1), K.Iw(k, 2), K.Iw(k, 3), K.Iw(k, 4), K.Iw(k, 5)};
    T ag[3] = {a[0], a[1], a[2] + gz};
    T t1[3], t2[3], t3[3], Iw_w[3], Ial[3];
    cross(al, mcw, t1);
    cross(w, mcw, t2);
    cross(w, t2, t2);
    symv(Iw, w, Iw_w);
    symv(Iw, al, Ial);
    cross(w, Iw_w, t3);
    T mk = mass3[k], t4[3];
    cross(mcw, ag, t4);
    for (int i = 0; i < 3; i++) {
      F[k][i] = mk * ag[i] + t1[i] + t2[i];
      N[k][i] = Ial[i] + t3[i] + t4[i];
    }
  }
  if (KINEXTRA) {
    // foot: d = rf - r_shank
    T d[3] = {rfv[0] - rk[2][0], rfv[1] - rk[2][1], rfv[2] - rk[2][2]};
    T wxd[3], t[3], alxd[3];
    cross(w, d, wxd);
    cross(w, wxd, t);
    cross(al, d, alxd);
    for (int i = 0; i < 3; i++) {
      D->Jdv[i] = a[i] + alxd[i] + t[i];
      D->rd[i] = vo[i] + wxd[i];
    }
    for (int k = 0; k < 3; k++) {
      T c2[3];
      cross(axk[k], D->rd, c2);
      for (int i = 0; i < 3; i++) D->Jd[3 * i + k] = D->Jd[3 * i + k] + c2[i];
    }
  }
  // inward pass
  for (int k = 2; k >= 0; k--) {
    hl[k] = dot(axk[k], N[k]);
    T r[3], rxF[3];
    for (int i = 0; i < 3; i++) r[i] = rk[k][i] - (k ? rk[k - 1][i] : T(0.0));
    cross(r, F[k], rxF);
    if (k > 0)
      for (int i = 0; i < 3; i++) { F[k - 1][i] = F[k - 1][i] + F[k][i]; N[k - 1][i] = N[k - 1][i] + N[k][i] + rxF[i]; }
    else
      for (int i = 0; i < 3; i++) { Fb[i] = F[0][i]; Nb[i] = N[0][i] + rxF[i]; }
  }
}

// Composite-rigid-body pass for one leg: Mbl, Mll and the leg's composite inertia at the base origin.
template <class T, class KinT>
WBC_HD void leg_crba(const T* mass3, const KinT& K, LegDyn<T>& D, T& Mc, T* Hc, T* Ic) {
  T cm = T(0.0), ch[3] = {T(0.0), T(0.0), T(0.0)}, cI[6] = {T(0.0), T(0.0), T(0.0), T(0.0), T(0.0), T(0.0)};
  T rk[3][3], axk[3][3];
  for (int k = 0; k < 3; k++)
    for (int i = 0; i < 3; i++) { rk[k][i] = K.r(k, i); axk[k][i] = K.ax(k, i); }
  for (int k = 2; k >= 0; k--) {
    // composite of links k..2 about origin k
    T zero[3] = {T(0.0), T(0.0), T(0.0)};
    T nm = T(0.0), nh[3] = {T(0.0), T(0.0), T(0.0)}, nI[6] = {T(0.0), T(0.0), T(0.0), T(0.0), T(0.0), T(0.0)};
    if (k < 2) {
      T r[3] = {rk[k + 1][0] - rk[k][0], rk[k + 1][1] - rk[k][1], rk[k + 1][2] - rk[k][2]};
      shift_add(cm, ch, cI, r, nm, nh, nI);
    }
    T mk = mass3[k];
    T mcw[3] = {K.mcw(k, 0), K.mcw(k, 1), K.mcw(k, 2)};
    T Iw[6] = {K.Iw(k, 0), K.Iw(k, 1), K.Iw(k, 2), K.Iw(k, 3), K.Iw(k, 4), K.Iw(k, 5)};
    shift_add(mk, mcw, Iw, zero, nm, nh, nI);
    cm = nm;
    for (int i = 0; i < 3; i++) ch[i] = nh[i];
    for (int i = 0; i < 6; i++) cI[i] = nI[i];
    // unit joint acceleration of joint k: n = I a, f = a x h  (about origin k)
    T n[3], f[3];
    symv(cI, axk[k], n);
    cross(axk[k], ch, f);
    // diagonal and ancestors within the leg
    T nk[3] = {n[0], n[1], n[2]};
    for (int j = k; j >= 0; j--) {
      if (j < k) {
        T r[3] = {rk[j + 1][0] - rk[j][0], rk[j + 1][1] - rk[j][1], rk[j + 1][2] - rk[j][2]}, rxf[3];
        cross(r, f, rxf);
        for (int i = 0; i < 3; i++) nk[i] = nk[i] + rxf[i];
      }
      T v = dot(axk[j], nk);
      // symmetric 3x3 index of (j,k), j<=k
      int idx = (j == k) ? j : (j == 0 ? (k == 1 ? 3 : 4) : 5);
      D.Mll[idx] = v;
    }
    T rxf[3];
    cross(rk[0], f, rxf);
    for (int i = 0; i < 3; i++) {
      D.Mbl[(i)*3 + k] = nk[i] + rxf[i];
      D.Mbl[(3 + i) * 3 + k] = f[i];
    }
  }
  shift_add(cm, ch, cI, rk[0], Mc, Hc, Ic);
}

// 3x3 inverse; returns |det| relative measure
template <class T> WBC_HD T inv3(const T* A, T* B) {
  T c0 = A[4] * A[8] - A[5] * A[7], c1 = A[5] * A[6] - A[3] * A[8], c2 = A[3] * A[7] - A[4] * A[6];
  T det = A[0] * c0 + A[1] * c1 + A[2] * c2;
  T id = T(1.0) / det;
  B[0] = c0 * id; B[1] = (A[2] * A[7] - A[1] * A[8]) * id; B[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  B[3] = c1 * id; B[4] = (A[0] * A[8] - A[2] * A[6]) * id; B[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  B[6] = c2 * id; B[7] = (A[1] * A[6] - A[0] * A[7]) * id; B[8] = (A[0] * A[4] - A[1] * A[3]) * id;
  return det;
}
// the same with the hardware-seeded reciprocal (kernels; ~10 instructions fewer than the IEEE division)
WBC_HD double inv3_fast(const double* A, double* B) {
  const double c0 = A[4] * A[8] - A[5] * A[7], c1 = A[5] * A[6] - A[3] * A[8], c2 = A[3] * A[7] - A[4] * A[6];
  const double det = A[0] * c0 + A[1] * c1 + A[2] * c2;
  const double id = fast_rcp(det);
  B[0] = c0 * id; B[1] = (A[2] * A[7] - A[1] * A[8]) * id; B[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  B[3] = c1 * id; B[4] = (A[0] * A[8] - A[2] * A[6]) * id; B[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  B[6] = c2 * id; B[7] = (A[1] * A[6] - A[0] * A[7]) * id; B[8] = (A[0] * A[4] - A[1] * A[3]) * id;
  return det;
}
template <class T> WBC_HD void sym_to_full(const T* S, T* A) {
  A[0] = S[0]; A[1] = S[3]; A[2] = S[4]; A[3] = S[3]; A[4] = S[1]; A[5] = S[5]; A[6] = S[4]; A[7] = S[5]; A[8] = S[2];
}
// C(3x3) = A(3x3) B(3x3)
template <class T> WBC_HD void mm3(const T* A, const T* B, T* C) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

template <class T> WBC_HD T wabs(const T& x) { return x < T(0.0) ? T(0.0) - x : x; }

enum { NZ = 12, MAXC = 42, PC_ROW = 40, CLF_ROW = 41 };   // reduced variables; (scalar form: constraint-table sizes and the ids of its dense rows)

}  // namespace wbc
