// wbc_quad.hpp -- one control tick computed cooperatively by a QUAD of lanes (product math, v2).
//
// Mapping (DESIGN.md "Kernel v2"): 4 consecutive lanes of a wavefront own one robot, lane l owns
// leg l of [LF RF LH RH].  The quadruped's arrowhead structure makes this the natural split:
//   * FK / RNEA / CRBA of a leg touch only that leg's 3 links          -> lane-local
//   * everything on the base (6x6 composite inertia, bias wrench, G_b, Lambda_bb) is a sum of
//     four leg contributions                                            -> 4-lane DPP butterfly
//   * the reduced QP has exactly 3 variables per leg (z_l = f_l or foot acceleration), so lane l
//     owns columns 3l..3l+2 of every z-space matrix (QR factor) and rows 3l..3l+2 of J = R^-1
//     (Goldfarb-Idnani: its Givens/Householder updates act on columns => local per row).
// Per-lane state is ~1/4 of the lane-per-robot kernel (no 14 KB/lane scratch) and a batch of N
// robots gives N/16 wavefronts instead of N/64.
//
// `Q` is the quad-communication policy: QuadDev (DPP quad_perm, wbc_kernels.hip) on the GPU,
// a 4-thread emulation in tools/host_tick.cpp for CPU-side validation.  Replicated values are
// bit-identical on the four lanes by construction (butterfly sums are commutative-symmetric),
// so quad-uniform branches stay uniform.
#pragma once
#include "wbc_tick.hpp"

namespace wbc {

// Active-set factor shared by the quad (LDS on the device; every lane performs the same
// replicated writes, so no barrier is needed inside a wavefront).
struct QuadShared {
  double Rq[NZ][NZ];
  double u[NZ + 1];
  int A[NZ];
  int pad_;
};

// Distributed Householder append: fold P dense rows into the upper-triangular factor.
// Lane l holds columns 3l..3l+2: R[12][3], A[P][3]; the right-hand-side column is replicated.
// Columns left of the pivot need no predication: their sub-diagonal part of R is zero and the
// owner zeroes its pivot column of A after each step, so the reflection is a no-op on them.
template <class Q, int P>
WBC_HD void quad_qr_append(Q& qo, int l, double (*R)[3], double* rhsR, double (*A)[3], double* rhsA) {
#pragma unroll
  for (int k = 0; k < NZ; k++) {
    const int owner = k / 3, kk = k % 3;
    double col[P];
    double s2 = 0.0;
#pragma unroll
    for (int i = 0; i < P; i++) {
      col[i] = qo.bcast_s(A[i][kk], owner);
      s2 += col[i] * col[i];
    }
    const double rkk = qo.bcast_s(R[k][kk], owner);
    if (!(s2 > 0.0)) continue;
    const double nrm = sqrt(rkk * rkk + s2);
    const double alpha = (rkk > 0.0) ? -nrm : nrm;
    const double v0 = rkk - alpha;
    const double beta = 1.0 / (nrm * (nrm + fabs(rkk)));  // = 2 / (s2 + v0^2)
    const bool own = (l == owner);
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
      double s = v0 * R[k][jj];
#pragma unroll
      for (int i = 0; i < P; i++) s += col[i] * A[i][jj];
      s *= beta;
      R[k][jj] -= s * v0;
#pragma unroll
      for (int i = 0; i < P; i++) A[i][jj] -= s * col[i];
    }
    R[k][kk] = own ? alpha : R[k][kk];
#pragma unroll
    for (int i = 0; i < P; i++) A[i][kk] = own ? 0.0 : A[i][kk];
    double s = v0 * rhsR[k];
#pragma unroll
    for (int i = 0; i < P; i++) s += col[i] * rhsA[i];
    s *= beta;
    rhsR[k] -= s * v0;
#pragma unroll
    for (int i = 0; i < P; i++) rhsA[i] -= s * col[i];
  }
}

// Dual part of one active-set step for an active set of size q <= QB: r = Rq^-1 d1, the
// blocking multiplier (ldrop, t1 = t1n/t1d) -- replicated on the quad.  Specialised by a bound
// on q because the wavefront's instruction count is the latency: typical trot ticks have q <= 3.
template <int QB>
WBC_HD void gi_dual(const QuadShared& sh, int q, const double* d, double* r, int& ldrop, double& t1n, double& t1d) {
#pragma unroll
  for (int k = NZ - 1; k >= QB; k--) r[k] = 0.0;
#pragma unroll
  for (int k = QB - 1; k >= 0; k--) {
    double s = d[k];
#pragma unroll
    for (int j = k + 1; j < QB; j++) s -= ((j < q) ? sh.Rq[k][j] : 0.0) * r[j];
    r[k] = (k < q) ? s * sh.Rq[k][k] : 0.0;  // the diagonal slot holds the RECIPROCAL pivot
  }
  ldrop = -1; t1n = 0.0; t1d = 1.0;
#pragma unroll
  for (int k = 0; k < QB; k++) {
    if (k < q && r[k] > 0.0) {
      const double uk = sh.u[k];
      if (ldrop < 0 || uk * t1d < t1n * r[k]) { t1n = uk; t1d = r[k]; ldrop = k; }
    }
  }
}

// Goldfarb-Idnani on the friction rows.  Jr = rows 3l..3l+2 of J (J J' = H^-1), zl = own 3
// entries of z.  Constraint index p = 4*leg + row.
//
// SIMT shape: ONE flat loop; in every trip each unfinished robot of the wavefront performs exactly
// one step of the algorithm (pick the most violated row if it has none in hand, then one
// primal/dual step ending in an add or a drop).  The textbook nested loops make the 16 robots of a
// wavefront diverge into different loop bodies and serialise (measured 6 us per step); here the
// trip count is the maximum over the robots, not the sum.
// Optional PC row (pc_controller.py): Vdot(z) = sum_l vrow_l . z_l + vc <= 0, index 16; every lane
// holds its own three coefficients, normalised by pc_inv = 1/|vrow| (0 disables the row).
template <class Q>
WBC_HD int quad_gi(Q& qo, int l, bool ct, double (*Jr)[NZ], double* zl, double mu_n, double inv_s, QuadShared& sh,
                   int* iters_out, const double* vrow_l = nullptr, double vc = 0.0, double pc_inv = 0.0) {
  const bool pc = pc_inv > 0.0;
  int q = 0, iters = 0, status = ST_OK;
  unsigned active = 0u;
  const int maxit = 200;
  bool done = false, need_pick = true;
  int p = -1;
  double sp = 0.0;
  double npl[3] = {0.0, 0.0, 0.0};
  for (int trip = 0; trip < maxit; trip++) {
    if (!done && need_pick) {
      double zinf = fmax(fabs(zl[0]), fmax(fabs(zl[1]), fabs(zl[2])));
      zinf = qo.max(zinf);
      const double tol = 1e-13 * (1.0 + zinf);
      sp = -tol;
      p = -1;
      if (ct) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const bool act = (active >> (4 * l + r)) & 1u;
          const double zc = (r >> 1) ? zl[1] : zl[0];
          const double s = ((r & 1) ? inv_s : -inv_s) * zc + mu_n * zl[2];
          if (!act && s < sp) { sp = s; p = 4 * l + r; }
        }
      }
      qo.argmin(sp, p);
      if (pc && !((active >> 16) & 1u)) {
        const double s = -(qo.sum(vrow_l[0] * zl[0] + vrow_l[1] * zl[1] + vrow_l[2] * zl[2]) + vc) * pc_inv;
        if (s < sp) { sp = s; p = 16; }
      }
      if (p < 0) {
        done = true;
      } else {
        if (p == 16) {
          npl[0] = -vrow_l[0] * pc_inv; npl[1] = -vrow_l[1] * pc_inv; npl[2] = -vrow_l[2] * pc_inv;
        } else {
          const int owner = p >> 2, rr = p & 3;
          const double sg = (rr & 1) ? inv_s : -inv_s;
          const bool mine = (l == owner);
          npl[0] = (mine && !(rr >> 1)) ? sg : 0.0;
          npl[1] = (mine && (rr >> 1)) ? sg : 0.0;
          npl[2] = mine ? mu_n : 0.0;
        }
        sh.u[q] = 0.0;
        need_pick = false;
      }
    }
    if (qo.wave_all(done)) break;
    if (done) continue;
    iters++;
    double d[NZ], dn = 0.0, d2n = 0.0;
#pragma unroll
    for (int k = 0; k < NZ; k++) {
      d[k] = qo.sum(Jr[0][k] * npl[0] + Jr[1][k] * npl[1] + Jr[2][k] * npl[2]);
      dn += d[k] * d[k];
      if (k >= q) d2n += d[k] * d[k];
    }
    double zd[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < NZ; k++) {
      const double dk = (k >= q) ? d[k] : 0.0;
      zd[0] += Jr[0][k] * dk; zd[1] += Jr[1][k] * dk; zd[2] += Jr[2][k] * dk;
    }
    // r = Rq^-1 d1 and the blocking multiplier, by a wave-uniform bound on the active-set size
    double r[NZ];
    int ldrop;
    double t1n, t1d;
    const int qmax = qo.wave_max_int(q);
    if (qmax <= 3) gi_dual<3>(sh, q, d, r, ldrop, t1n, t1d);
    else if (qmax <= 6) gi_dual<6>(sh, q, d, r, ldrop, t1n, t1d);
    else gi_dual<NZ>(sh, q, d, r, ldrop, t1n, t1d);
    const bool have_t1 = ldrop >= 0;
    const double t1 = have_t1 ? t1n / t1d : 0.0;
    const bool dependent = !(d2n > 1e-22 * dn) || q == NZ;
    const double znp = qo.sum(zd[0] * npl[0] + zd[1] * npl[1] + zd[2] * npl[2]);
    const double t2 = dependent ? 0.0 : -sp / znp;
    if (dependent && !have_t1) { status = ST_SINGULAR; done = true; continue; }
    const bool full = !dependent && (!have_t1 || !(t1 < t2));
    const double t = full ? t2 : t1;
#pragma unroll
    for (int k = 0; k < NZ; k++)
      if (k < q) sh.u[k] -= t * r[k];
    sh.u[q] += t;
    if (!dependent) { zl[0] += t * zd[0]; zl[1] += t * zd[1]; zl[2] += t * zd[2]; }
    if (full) {
      // one Householder reflection H on d[q:] (H d2 = alpha e1); J2 <- J2 H on the own rows
      double dq = 0.0;
#pragma unroll
      for (int k = 0; k < NZ; k++) dq = (k == q) ? d[k] : dq;
      const double nrm = sqrt(d2n);
      const double alpha = (dq > 0.0) ? -nrm : nrm;
      const double vq = dq - alpha;
      const double vv = d2n - dq * dq + vq * vq;
      if (vv > 0.0) {
        const double beta = 2.0 / vv;
        double w0 = 0.0, w1 = 0.0, w2 = 0.0;
        double hv[NZ];
#pragma unroll
        for (int k = 0; k < NZ; k++) {
          hv[k] = (k < q) ? 0.0 : ((k == q) ? vq : d[k]);
          w0 += Jr[0][k] * hv[k]; w1 += Jr[1][k] * hv[k]; w2 += Jr[2][k] * hv[k];
        }
        w0 *= beta; w1 *= beta; w2 *= beta;
#pragma unroll
        for (int k = 0; k < NZ; k++) {
          Jr[0][k] -= w0 * hv[k]; Jr[1][k] -= w1 * hv[k]; Jr[2][k] -= w2 * hv[k];
        }
      }
#pragma unroll
      for (int k = 0; k < NZ; k++)
        if (k < q) sh.Rq[k][q] = d[k];
      sh.Rq[q][q] = 1.0 / alpha;
      sh.A[q] = p;
      active |= (1u << p);
      q++;
      need_pick = true;
      continue;
    }
    // partial / pure dual step: drop active constraint ldrop
    active &= ~(1u << sh.A[ldrop]);
    for (int j = ldrop; j < q - 1; j++) {
      sh.A[j] = sh.A[j + 1];
      sh.u[j] = sh.u[j + 1];
      for (int k = 0; k <= j + 1; k++) sh.Rq[k][j] = sh.Rq[k][j + 1];
    }
    sh.u[q - 1] = sh.u[q];
    q--;
    sh.u[q + 1] = 0.0;
    for (int j = ldrop; j < q; j++) {
      // After the shift, column j (>= ldrop) holds the old column j+1 verbatim: rows 0..j are plain
      // values (including the new "diagonal" (j,j)) and row j+1 is the old pivot, stored as its
      // reciprocal.  Rotate rows j, j+1 to kill that sub-diagonal entry.
      const double a = sh.Rq[j][j], bb = 1.0 / sh.Rq[j + 1][j];
      const double h = sqrt(a * a + bb * bb), c = a / h, sn = bb / h;
      sh.Rq[j][j] = 1.0 / h;
      for (int k = j + 1; k < q; k++) {
        const double x = sh.Rq[j][k], y = sh.Rq[j + 1][k];
        sh.Rq[j][k] = c * x + sn * y;
        sh.Rq[j + 1][k] = c * y - sn * x;
      }
#pragma unroll
      for (int jj = 0; jj < NZ - 1; jj++) {
        if (jj != j) continue;
#pragma unroll
        for (int i = 0; i < 3; i++) {
          const double x = Jr[i][jj], y = Jr[i][jj + 1];
          Jr[i][jj] = c * x + sn * y;
          Jr[i][jj + 1] = c * y - sn * x;
        }
      }
    }
    if (!dependent) sp = qo.sum(npl[0] * zl[0] + npl[1] * zl[1] + npl[2] * zl[2]) - ((p == 16) ? vc * pc_inv : 0.0);
  }
  *iters_out = iters;
  if (!done && status == ST_OK) status = ST_ITER;
  return status;
}

// 6x6 solve with partial pivoting for NR right-hand sides held as extra columns: Ab = [G | rhs].
// Fully unrolled (register-resident); returns min|pivot| / max|pivot|.
template <int NR> WBC_HD double solve6(double (*Ab)[6 + NR]) {
  double pmin = 0.0, pmax = 0.0;
#pragma unroll
  for (int c = 0; c < 6; c++) {
    int p = c;
    double best = fabs(Ab[c][c]);
#pragma unroll
    for (int r = c + 1; r < 6; r++) {
      const double a = fabs(Ab[r][c]);
      if (a > best) { best = a; p = r; }
    }
#pragma unroll
    for (int r = c + 1; r < 6; r++) {
      if (r != p) continue;
#pragma unroll
      for (int j = c; j < 6 + NR; j++) { const double t = Ab[c][j]; Ab[c][j] = Ab[r][j]; Ab[r][j] = t; }
    }
    if (c == 0 || best < pmin) pmin = best;
    if (best > pmax) pmax = best;
    const double id = 1.0 / Ab[c][c];
#pragma unroll
    for (int r = c + 1; r < 6; r++) {
      const double f = Ab[r][c] * id;
#pragma unroll
      for (int j = c + 1; j < 6 + NR; j++) Ab[r][j] -= f * Ab[c][j];
    }
  }
#pragma unroll
  for (int c = 5; c >= 0; c--) {
    const double id = 1.0 / Ab[c][c];
#pragma unroll
    for (int n = 0; n < NR; n++) {
      double s = Ab[c][6 + n];
#pragma unroll
      for (int j = c + 1; j < 6; j++) s -= Ab[c][j] * Ab[j][6 + n];
      Ab[c][6 + n] = s * id;
    }
  }
  return pmin / pmax;
}

// Diagnostic builds only (-DWBC_CUT=k): return after phase k, folding the live values into the
// output so the truncated phases are not dead-code-eliminated.  Timing such builds as whole
// kernels gives a phase attribution that the optimiser cannot blur (unlike in-kernel stamps).
#ifdef WBC_CUT
#define WBC_CUT_AT(k, expr)                                                        \
  if (WBC_CUT == (k)) {                                                            \
    double sink_ = (expr);                                                         \
    for (int k_ = 0; k_ < 3; k_++) out_tau(m.act_inv[3 * l + k_], sink_);          \
    out_met(0, sink_); out_met(1, sink_); out_met(2, 0.0); out_met(3, 0.0);        \
    *iters_out = 0;                                                                \
    return ST_OK;                                                                  \
  }
#else
#define WBC_CUT_AT(k, expr)
#endif

// The tick.  Every lane of the quad calls this with its own Q (lane id = leg).
// out_tau(row, x): lane l writes the output rows of its own three joints;  out_met: lane 0 only.
//
// The order of the phases is chosen for register pressure (the kernel is VGPR-bound): the leg
// kinematics die right after the last Newton-Euler pass, the torque-map columns are built just
// before the level-2 rows that consume them, and torques are recovered from (a_b, z_l) locally.
// Cold per-lane storage ("stage"): values written in the leg phase and only needed again much later
// (torque-map build, MPTC assembly, output) are parked in LDS instead of occupying registers for
// the whole tick.  Slots in doubles.
enum { ST_Y = 0, ST_PM = 18, ST_JL = 27, ST_JI = 36, ST_JD = 45, ST_MLL = 54, ST_B = 60, ST_AB0 = 78, ST_N = 84 };

// Read of a quad-shared (replicated-write) LDS value.  A volatile read (forcing a real LDS load)
// was measured 10 % SLOWER than letting the compiler forward the stored value (profiles/r01).
#define WBC_SH_GET(x) (x)
template <class T> struct StageReg {  // host / register-resident fallback
  T d[ST_N];
  WBC_HD void put(int i, T v) { d[i] = v; }
  WBC_HD T get(int i) const { return d[i]; }
};

template <class Q, int KIND, class KinT, class StT, class In, class OutTau, class OutMet>
WBC_HD int quad_tick(const ModelC& m, const ParamsC& P, Q& qo, In in, unsigned mask, double mu, double mass_scale,
                     KinT& K, StT& st, QuadShared& sh, OutTau out_tau, OutMet out_met, int* iters_out) {
  const int l = qo.lane();
  const bool ct = (mask >> l) & 1u;
  int status = ST_OK;
  WBC_STAMP(0);
  WBC_CUT_AT(11, in(0) + in(20) + in(40) + in(37 + 18 + 9 * l) + mu + mass_scale)
  // ---------------- state (replicated on the 4 lanes)
  double R0[9];
  {
    const double qw = in(0), qx = in(1), qy = in(2), qz = in(3);
    const double s = 2.0 / (qw * qw + qx * qx + qy * qy + qz * qz);
    R0[0] = 1.0 - s * (qy * qy + qz * qz); R0[1] = s * (qx * qy - qw * qz); R0[2] = s * (qx * qz + qw * qy);
    R0[3] = s * (qx * qy + qw * qz); R0[4] = 1.0 - s * (qx * qx + qz * qz); R0[5] = s * (qy * qz - qw * qx);
    R0[6] = s * (qx * qz - qw * qy); R0[7] = s * (qy * qz + qw * qx); R0[8] = 1.0 - s * (qx * qx + qy * qy);
  }
  const double p0[3] = {in(4), in(5), in(6)};
  const double w0[3] = {in(19), in(20), in(21)};
  const double v0[3] = {in(22), in(23), in(24)};
  const double gz = m.gravity;
  const double bm = m.base_mass * mass_scale;
  double bmc[3], bI[6];
  {
    const double t[3] = {m.base_mc[0] * mass_scale, m.base_mc[1] * mass_scale, m.base_mc[2] * mass_scale};
    rotv(R0, t, bmc);
    rot_inertia(R0, m.base_I, bI);
    for (int i = 0; i < 6; i++) bI[i] *= mass_scale;
  }
  // task-space errors of the body (both laws)
  double rpy[3], E[9], rpyd[3];
  {
    rpy[0] = atan2(R0[7], R0[8]);
    rpy[1] = atan2(-R0[6], sqrt(R0[0] * R0[0] + R0[3] * R0[3]));
    rpy[2] = atan2(R0[3], R0[0]);
    // sin/cos of pitch and yaw straight from R0 = Rz(y) Ry(p) Rx(r): no trig calls needed
    const double cp = sqrt(R0[0] * R0[0] + R0[3] * R0[3]), sp = -R0[6];
    const double icp = 1.0 / cp;
    const double cy = R0[0] * icp, sy = R0[3] * icp;
    E[0] = cp * cy; E[1] = -sy; E[2] = 0.0; E[3] = cp * sy; E[4] = cy; E[5] = 0.0; E[6] = -sp; E[7] = 0.0; E[8] = 1.0;
    const double Ei[9] = {cy * icp, sy * icp, 0.0, -sy, cy, 0.0, cy * sp * icp, sy * sp * icp, 1.0};
    rotv(Ei, w0, rpyd);
  }
  double xt_b[6], xdt_b[6], xdd_b[6], ades[6];
  {
    double tg_pb[3], tg_pdb[3], tg_pddb[3], tg_rpy[3], tg_rpyd[3], tg_rpydd[3];
    for (int i = 0; i < 3; i++) {
      tg_pb[i] = in(37 + i); tg_pdb[i] = in(40 + i); tg_pddb[i] = in(43 + i);
      tg_rpy[i] = in(46 + i); tg_rpyd[i] = in(49 + i); tg_rpydd[i] = in(52 + i);
    }
    for (int i = 0; i < 3; i++) { xt_b[i] = rpy[i] - tg_rpy[i]; xt_b[3 + i] = p0[i] - tg_pb[i]; }
    if (KIND == KIND_ID) {
      double rpydd_des[3], od[3];
      for (int i = 0; i < 3; i++) {
        ades[3 + i] = tg_pddb[i] - P.Kp_body_p * xt_b[3 + i] - P.Kd_body_p * (v0[i] - tg_pdb[i]);
        rpydd_des[i] = tg_rpydd[i] - P.Kp_body_rpy * xt_b[i] - P.Kd_body_rpy * (rpyd[i] - tg_rpyd[i]);
      }
      rotv(E, rpydd_des, od);
      for (int i = 0; i < 3; i++) { ades[i] = od[i]; xdt_b[i] = 0.0; xdt_b[3 + i] = 0.0; xdd_b[i] = 0.0; xdd_b[3 + i] = 0.0; }
    } else {
      double om_rt[3], xdn[3], xddn[3];
      rotv(E, rpyd, om_rt);      // literal E(rpy) * rpyd round trip of mptc_controller.py:245
      rotv(E, tg_rpyd, xdn);
      rotv(E, tg_rpydd, xddn);
      for (int i = 0; i < 3; i++) {
        xdt_b[i] = om_rt[i] - xdn[i];
        xdt_b[3 + i] = v0[i] - tg_pdb[i];
        xdd_b[i] = xddn[i];
        xdd_b[3 + i] = tg_pddb[i];
        ades[i] = 0.0; ades[3 + i] = 0.0;
      }
    }
  }
  WBC_CUT_AT(12, xt_b[0] + xt_b[4] + xdt_b[1] + xdt_b[5] + xdd_b[2] + ades[1] + bI[3] + bmc[1] + R0[5] + rpyd[0] + E[3])
  WBC_STAMP(1);
  // ---------------- own leg: FK, Newton-Euler bias pass, composite inertia, foot Jacobian
  double rf[3], Jdv[3], pd[3], rd[3], hl[3];
  double X[18];
  double hbN[6];                 // own leg's reaction wrench at the base origin
  double t0l[3];                 // own rows of t0 without the Y ab0 part: hl + ct Pm bc
  double lm = 0.0, lh[3] = {0.0, 0.0, 0.0}, lI[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  double Cb_leg[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, Cl[3] = {0.0, 0.0, 0.0}, xi[3] = {0.0, 0.0, 0.0};
  double xt_s[3], xdt_s[3], xdd_s[3];
  {
    LegDyn<double> D;
    double qd[3];
    double Jl[9], Ji[9], Y[18], Pm[9];
    const double mass3[3] = {m.link[l][0].mass, m.link[l][1].mass, m.link[l][2].mass};
    {
      double sn[3], cs[3];
      for (int k = 0; k < 3; k++) {
        const int row = m.q_perm[3 * l + k];
        const double th = in(7 + row);
        wbc_sincos(th, sn[k], cs[k]);
        qd[k] = in(25 + row);
      }
      WBC_CUT_AT(13, sn[0] + sn[1] + sn[2] + cs[0] + cs[1] + cs[2] + qd[1] + xt_b[0] + xdt_b[4] + xdd_b[2] + ades[1] + bI[3] + R0[5])
      leg_fk_vec(m, l, R0, sn, cs, K);
    }
    WBC_CUT_AT(7, K.rf(0) + K.r(1, 1) + K.Iw(2, 3) + K.mcw(0, 2) + K.ax(2, 0) + xt_b[0] + xdt_b[4] + xdd_b[2] + ades[1] + qd[0])
    WBC_STAMP(2);
    {
      const KinT& Kc = K;
      leg_rnea<double, true>(mass3, Kc, w0, qd, gz, hl, hbN, hbN + 3, &D);
      leg_crba(mass3, Kc, D, lm, lh, lI);
    }
    WBC_STAMP(3);
    for (int i = 0; i < 3; i++) { rf[i] = K.rf(i); Jdv[i] = D.Jdv[i]; rd[i] = D.rd[i]; pd[i] = v0[i] + D.rd[i]; }
    for (int i = 0; i < 9; i++) st.put(ST_JD + i, D.Jd[i]);
    for (int k = 0; k < 3; k++) {
      const double d[3] = {rf[0] - K.r(k, 0), rf[1] - K.r(k, 1), rf[2] - K.r(k, 2)};
      const double axv[3] = {K.ax(k, 0), K.ax(k, 1), K.ax(k, 2)};
      double c[3];
      cross(axv, d, c);
      for (int i = 0; i < 3; i++) Jl[3 * i + k] = c[i];
    }
    const double det = inv3(Jl, Ji);
    if (qo.any(!(fabs(det) > 1e-12))) status = ST_SINGULAR;
    double Mf[9];
    sym_to_full(D.Mll, Mf);
    mm3(Mf, Ji, Pm);  // Pm = Mll Ji
    WBC_CUT_AT(8, Pm[0] + Pm[8] + Ji[4] + Jl[2] + D.Mbl[7] + D.Jd[3] + Jdv[1] + rd[2] + hl[0] + hbN[4] + lm + lh[0] + lI[5] + xt_b[0] + xdt_b[4] + xdd_b[2] + ades[1])
    // own foot targets / errors (zero for a contact leg)
    for (int i = 0; i < 3; i++) {
      const double pf = p0[i] + rf[i];
      const double tp = in(37 + 18 + 9 * l + i), tpd = in(37 + 21 + 9 * l + i), tpdd = in(37 + 24 + 9 * l + i);
      xt_s[i] = ct ? 0.0 : pf - tp;
      xdt_s[i] = ct ? 0.0 : pd[i] - tpd;
      xdd_s[i] = ct ? 0.0 : tpdd;
    }
    if (KIND != KIND_ID) {
      // xi = Jbar xd_tilde on the own joints.  Contact leg: xi = -Mll^-1 (Y xdt_b) - Ji Jfb xdt_b
      // with Y xdt_b = Mbl' xdt_b - Pm (Jfb xdt_b), so neither Y nor Mll^-1 Y is needed yet.
      double Mli[9];
      inv3(Mf, Mli);
      double t[3], jfb[3];
      cross(xdt_b, rf, t);
      for (int i = 0; i < 3; i++) jfb[i] = xdt_b[3 + i] + t[i];
      if (ct) {
        double yx[3];
        for (int i = 0; i < 3; i++) {
          double s = 0.0;
          for (int j = 0; j < 6; j++) s += D.Mbl[3 * j + i] * xdt_b[j];
          yx[i] = s - (Pm[3 * i] * jfb[0] + Pm[3 * i + 1] * jfb[1] + Pm[3 * i + 2] * jfb[2]);
        }
        for (int i = 0; i < 3; i++)
          xi[i] = -(Mli[3 * i] * yx[0] + Mli[3 * i + 1] * yx[1] + Mli[3 * i + 2] * yx[2]) -
                  (Ji[3 * i] * jfb[0] + Ji[3 * i + 1] * jfb[1] + Ji[3 * i + 2] * jfb[2]);
      } else {
        const double y[3] = {xdt_s[0] - jfb[0], xdt_s[1] - jfb[1], xdt_s[2] - jfb[2]};
        rotv(Ji, y, xi);
      }
      WBC_CUT_AT(9, xi[0] + xi[1] + xi[2] + Pm[0] + Ji[4] + Jl[2] + D.Mbl[7] + D.Jd[3] + Jdv[1] + rd[2] + hl[0] + hbN[4] + lm + lh[0] + lI[5] + xt_b[0] + xdd_b[2])
      // C xi = 1/4 [h(v + xi) - h(v - xi)]  (two more Newton-Euler passes over the cached kinematics)
      const KinT& Kc = K;
      for (int sgi = 0; sgi < 2; sgi++) {
        const double sg = sgi ? -0.25 : 0.25, s1 = sgi ? -1.0 : 1.0;
        const double wv[3] = {w0[0] + s1 * xdt_b[0], w0[1] + s1 * xdt_b[1], w0[2] + s1 * xdt_b[2]};
        const double qv[3] = {qd[0] + s1 * xi[0], qd[1] + s1 * xi[1], qd[2] + s1 * xi[2]};
        double hl2[3], Nb[3], Fb[3];
        leg_rnea<double, false>(mass3, Kc, wv, qv, 0.0, hl2, Nb, Fb, (LegDyn<double>*)nullptr);
        for (int i = 0; i < 3; i++) { Cb_leg[i] += sg * Nb[i]; Cb_leg[3 + i] += sg * Fb[i]; Cl[i] += sg * hl2[i]; }
      }
    }
    WBC_CUT_AT(10, Cb_leg[0] + Cb_leg[4] + Cl[1] + xi[0] + Pm[0] + Ji[4] + Jl[2] + D.Mbl[7] + D.Jd[3] + Jdv[1] + rd[2] + hl[0] + hbN[4] + lm + lh[0] + lI[5] + xt_b[0] + xdd_b[2])
    WBC_STAMP(4);
    // X = Mbl Ji, Y = Mbl' - Pm Jfb
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 3; j++)
        X[3 * i + j] = D.Mbl[3 * i] * Ji[j] + D.Mbl[3 * i + 1] * Ji[3 + j] + D.Mbl[3 * i + 2] * Ji[6 + j];
    for (int i = 0; i < 3; i++) {
      const double a0 = Pm[3 * i], a1 = Pm[3 * i + 1], a2 = Pm[3 * i + 2];
      Y[6 * i + 0] = D.Mbl[0 * 3 + i] + (a1 * rf[2] - a2 * rf[1]);
      Y[6 * i + 1] = D.Mbl[1 * 3 + i] + (a2 * rf[0] - a0 * rf[2]);
      Y[6 * i + 2] = D.Mbl[2 * 3 + i] + (a0 * rf[1] - a1 * rf[0]);
      Y[6 * i + 3] = D.Mbl[3 * 3 + i] - a0;
      Y[6 * i + 4] = D.Mbl[4 * 3 + i] - a1;
      Y[6 * i + 5] = D.Mbl[5 * 3 + i] - a2;
    }
    for (int i = 0; i < 18; i++) st.put(ST_Y + i, Y[i]);
    for (int i = 0; i < 9; i++) { st.put(ST_PM + i, Pm[i]); st.put(ST_JL + i, Jl[i]); st.put(ST_JI + i, Ji[i]); }
    for (int i = 0; i < 6; i++) st.put(ST_MLL + i, D.Mll[i]);
    for (int i = 0; i < 3; i++) {
      const double b0 = ct ? (-P.Kd_contact * pd[0] - Jdv[0]) : 0.0, b1 = ct ? (-P.Kd_contact * pd[1] - Jdv[1]) : 0.0,
                   b2 = ct ? (-P.Kd_contact * pd[2] - Jdv[2]) : 0.0;
      t0l[i] = hl[i] + (Pm[3 * i] * b0 + Pm[3 * i + 1] * b1 + Pm[3 * i + 2] * b2);
    }
  }  // D dead; K (the kinematics arena) is not read again: its storage may alias `sh`
  const double bc[3] = {ct ? (-P.Kd_contact * pd[0] - Jdv[0]) : 0.0, ct ? (-P.Kd_contact * pd[1] - Jdv[1]) : 0.0,
                        ct ? (-P.Kd_contact * pd[2] - Jdv[2]) : 0.0};
  WBC_CUT_AT(1, X[0] + X[7] + X[17] + hbN[0] + hbN[5] + lm + lh[1] + lI[3] + Cb_leg[2] + Cl[1] + xi[0] + t0l[0] + t0l[2] + st.get(ST_Y + 3) + st.get(ST_JL + 4) + st.get(ST_JD + 2) + xt_s[0] + xdt_s[1])
  WBC_STAMP(5);
  // ---------------- base: bias wrench, composite inertia -> Gs = G_b, kv
  double Gs[6][6], kv[6];
  {
    double hb[6];
    {
      double t2[3], t3[3], Iw_w[3], t4[3];
      const double g3[3] = {0.0, 0.0, gz};
      cross(w0, bmc, t2);
      cross(w0, t2, t2);
      symv(bI, w0, Iw_w);
      cross(w0, Iw_w, t3);
      cross(bmc, g3, t4);
      for (int i = 0; i < 3; i++) { hb[i] = t3[i] + t4[i]; hb[3 + i] = bm * g3[i] + t2[i]; }
    }
    for (int i = 0; i < 6; i++)
      kv[i] = hb[i] + qo.sum(hbN[i] + (ct ? (X[3 * i] * bc[0] + X[3 * i + 1] * bc[1] + X[3 * i + 2] * bc[2]) : 0.0));
    const double Mc = bm + qo.sum(lm);
    double Hc[3], Ic[6];
    for (int i = 0; i < 3; i++) Hc[i] = bmc[i] + qo.sum(lh[i]);
    for (int i = 0; i < 6; i++) Ic[i] = bI[i] + qo.sum(lI[i]);
    double Mbb[6][6];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) Mbb[i][j] = 0.0;
    Mbb[0][0] = Ic[0]; Mbb[1][1] = Ic[1]; Mbb[2][2] = Ic[2];
    Mbb[0][1] = Mbb[1][0] = Ic[3]; Mbb[0][2] = Mbb[2][0] = Ic[4]; Mbb[1][2] = Mbb[2][1] = Ic[5];
    Mbb[0][4] = -Hc[2]; Mbb[0][5] = Hc[1];
    Mbb[1][3] = Hc[2];  Mbb[1][5] = -Hc[0];
    Mbb[2][3] = -Hc[1]; Mbb[2][4] = Hc[0];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Mbb[3 + j][i] = Mbb[i][3 + j];
    Mbb[3][3] = Mc; Mbb[4][4] = Mc; Mbb[5][5] = Mc;
    for (int i = 0; i < 6; i++) {
      const double a0 = X[3 * i], a1 = X[3 * i + 1], a2 = X[3 * i + 2];
      Gs[i][0] = Mbb[i][0] + qo.sum(a1 * rf[2] - a2 * rf[1]);
      Gs[i][1] = Mbb[i][1] + qo.sum(a2 * rf[0] - a0 * rf[2]);
      Gs[i][2] = Mbb[i][2] + qo.sum(a0 * rf[1] - a1 * rf[0]);
      Gs[i][3] = Mbb[i][3] - qo.sum(a0);
      Gs[i][4] = Mbb[i][4] - qo.sum(a1);
      Gs[i][5] = Mbb[i][5] - qo.sum(a2);
    }
  }
  WBC_STAMP(6);
  // own columns of B and ab0:  G_b [B_l | ab0] = [W_l or -X_l | -k]
  double B[6][3], ab0[6];
  {
    double Ab[6][10];
    for (int i = 0; i < 6; i++) {
      for (int j = 0; j < 6; j++) Ab[i][j] = Gs[i][j];
      Ab[i][9] = -kv[i];
    }
    for (int j = 0; j < 3; j++) {
      const double e[3] = {j == 0 ? 1.0 : 0.0, j == 1 ? 1.0 : 0.0, j == 2 ? 1.0 : 0.0};
      double c[3];
      cross(rf, e, c);
      for (int i = 0; i < 3; i++) {
        Ab[i][6 + j] = ct ? c[i] : -X[3 * i + j];
        Ab[3 + i][6 + j] = ct ? e[i] : -X[3 * (3 + i) + j];
      }
    }
    const double rc = solve6<4>(Ab);
    if (!(rc > 1e-12)) status = ST_SINGULAR;
    for (int i = 0; i < 6; i++) { B[i][0] = Ab[i][6]; B[i][1] = Ab[i][7]; B[i][2] = Ab[i][8]; ab0[i] = Ab[i][9]; }
    for (int i = 0; i < 6; i++) {
      st.put(ST_AB0 + i, ab0[i]);
      for (int j = 0; j < 3; j++) st.put(ST_B + 3 * i + j, B[i][j]);
    }
  }

  WBC_CUT_AT(2, B[0][0] + B[3][1] + B[5][2] + ab0[0] + ab0[5] + Gs[2][3] + Cb_leg[2] + Cl[1] + xi[0] + t0l[1] + st.get(ST_Y + 3))
  WBC_STAMP(7);
  // ---------------- level-1 rows
  const double eps = sqrt(P.eps2);
  const double sw_b = sqrt(P.w_body), sw_f = sqrt(P.w_foot);
  double Rc[NZ][3], rhsR[NZ];
  double met_err = 0.0;
  for (int i = 0; i < 6; i++) met_err += xt_b[i] * xt_b[i];
  met_err += qo.sum(xt_s[0] * xt_s[0] + xt_s[1] * xt_s[1] + xt_s[2] * xt_s[2]);
  double met_V = 0.0, met_Vdot = 0.0;
  double vrow[3] = {0.0, 0.0, 0.0}, vconst = 0.0;  // Vdot += vconst + sum_l vrow_l . z_l
  {
    // diagonal rows: swing leg sqrt(w_foot) (ID) / 0 (MPTC); contact leg eps
    double dval[3], drhs[3];
    for (int i = 0; i < 3; i++) {
      if (ct) { dval[i] = eps; drhs[i] = 0.0; }
      else if (KIND == KIND_ID) {
        const double des = xdd_s[i] - P.Kp_foot * xt_s[i] - P.Kd_foot * xdt_s[i];
        dval[i] = sw_f; drhs[i] = sw_f * (des - Jdv[i]);
      } else { dval[i] = 0.0; drhs[i] = 0.0; }
    }
#pragma unroll
    for (int k = 0; k < NZ; k++)
#pragma unroll
      for (int jj = 0; jj < 3; jj++) Rc[k][jj] = (k == 3 * l + jj) ? dval[jj] : 0.0;
#pragma unroll
    for (int lp = 0; lp < 4; lp++)
#pragma unroll
      for (int i = 0; i < 3; i++) rhsR[3 * lp + i] = qo.bcast_s(drhs[i], lp);
  }
  if (KIND == KIND_ID) {
    double blk[6][3], brhs[6];
    for (int i = 0; i < 6; i++) {
      for (int j = 0; j < 3; j++) blk[i][j] = sw_b * B[i][j];
      brhs[i] = sw_b * (ades[i] - ab0[i]);
    }
    quad_qr_append<Q, 6>(qo, l, Rc, rhsR, blk, brhs);
  } else {
    // ---- MPTC in task coordinates (derivation: wbc_tick.hpp / DESIGN.md)
    // Register diet: Lambda_bb (a replicated 6x6) is never materialised.  Each of its rows is
    // formed from six quad sums and consumed at once by the two Lambda*vector products and by the
    // body least-squares row, then dropped.
    double Y[18], Ji[9], MiY[18], Mt_bl[18], Mt_ll[9];
    for (int i = 0; i < 18; i++) Y[i] = st.get(ST_Y + i);
    for (int i = 0; i < 9; i++) Ji[i] = st.get(ST_JI + i);
    {
      double Ms[6], Mf[9], Mli[9];
      for (int i = 0; i < 6; i++) Ms[i] = st.get(ST_MLL + i);
      sym_to_full(Ms, Mf);
      inv3(Mf, Mli);
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 6; j++) MiY[6 * i + j] = Mli[3 * i] * Y[j] + Mli[3 * i + 1] * Y[6 + j] + Mli[3 * i + 2] * Y[12 + j];
    }
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 6; j++) Mt_bl[3 * j + i] = Ji[i] * Y[j] + Ji[3 + i] * Y[6 + j] + Ji[6 + i] * Y[12 + j];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        Mt_ll[3 * i + j] = Ji[i] * st.get(ST_PM + j) + Ji[3 + i] * st.get(ST_PM + 3 + j) + Ji[6 + i] * st.get(ST_PM + 6 + j);
    // Lambda (J Minv C xi): base rows replicated, own swing rows local
    double LJ_b[6], LJ_s[3];
    {
      double Cb_base[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      for (int sgi = 0; sgi < 2; sgi++) {
        const double sg = sgi ? -0.25 : 0.25, s1 = sgi ? -1.0 : 1.0;
        const double wv[3] = {w0[0] + s1 * xdt_b[0], w0[1] + s1 * xdt_b[1], w0[2] + s1 * xdt_b[2]};
        double t2[3], t3[3], Iw_w[3];
        cross(wv, bmc, t2);
        cross(wv, t2, t2);
        symv(bI, wv, Iw_w);
        cross(wv, Iw_w, t3);
        for (int i = 0; i < 3; i++) { Cb_base[i] += sg * t3[i]; Cb_base[3 + i] += sg * t2[i]; }
      }
      double gl[3], c[3];
      for (int i = 0; i < 3; i++) gl[i] = Ji[i] * Cl[0] + Ji[3 + i] * Cl[1] + Ji[6 + i] * Cl[2];
      cross(rf, gl, c);
      for (int j = 0; j < 6; j++) {
        double loc = Cb_leg[j] - ((j < 3) ? c[j] : gl[j - 3]);
        if (ct) loc -= MiY[j] * Cl[0] + MiY[6 + j] * Cl[1] + MiY[12 + j] * Cl[2];
        LJ_b[j] = Cb_base[j] + qo.sum(loc);
      }
      for (int i = 0; i < 3; i++) LJ_s[i] = ct ? 0.0 : gl[i];
    }
    double s1_s[3];
    {
      double t[3];
      cross(xdt_b, rd, t);
      for (int i = 0; i < 3; i++) {
        const double jx = t[i] + st.get(ST_JD + 3 * i) * xi[0] + st.get(ST_JD + 3 * i + 1) * xi[1] + st.get(ST_JD + 3 * i + 2) * xi[2];
        s1_s[i] = ct ? 0.0 : xdd_s[i] - Jdv[i] + jx;
      }
    }
    // swing parts of the two Lambda*vector products (independent of Lambda_bb)
    double Ls_s[3], Lx_s[3];
    for (int i = 0; i < 3; i++) {
      double a = Mt_ll[3 * i] * s1_s[0] + Mt_ll[3 * i + 1] * s1_s[1] + Mt_ll[3 * i + 2] * s1_s[2];
      double b = Mt_ll[3 * i] * xdt_s[0] + Mt_ll[3 * i + 1] * xdt_s[1] + Mt_ll[3 * i + 2] * xdt_s[2];
      for (int j = 0; j < 6; j++) { a += Mt_bl[3 * j + i] * xdd_b[j]; b += Mt_bl[3 * j + i] * xdt_b[j]; }
      Ls_s[i] = ct ? 0.0 : a;
      Lx_s[i] = ct ? 0.0 : b;
    }
    double c1_s[3];
    {
      double lv = 0.0, lvd = 0.0;
      for (int i = 0; i < 3; i++) {
        c1_s[i] = LJ_s[i] - Ls_s[i] + P.Kp_foot * xt_s[i] + P.Kd_foot * xdt_s[i];
        lv += 0.5 * P.Kp_foot * xt_s[i] * xt_s[i] + 0.5 * xdt_s[i] * Lx_s[i];
        lvd += -P.Kd_foot * xdt_s[i] * xdt_s[i] + xdt_s[i] * c1_s[i];
      }
      met_V += qo.sum(lv);
      met_Vdot += qo.sum(lvd);
    }
    double blk[12][3], brhs[12];
    {
      double A[18];  // Ji Jfb
      for (int i = 0; i < 3; i++) {
        const double a0 = Ji[3 * i], a1 = Ji[3 * i + 1], a2 = Ji[3 * i + 2];
        A[6 * i + 0] = -(a1 * rf[2] - a2 * rf[1]);
        A[6 * i + 1] = -(a2 * rf[0] - a0 * rf[2]);
        A[6 * i + 2] = -(a0 * rf[1] - a1 * rf[0]);
        A[6 * i + 3] = a0; A[6 * i + 4] = a1; A[6 * i + 5] = a2;
      }
      for (int j = 0; j < 3; j++) vrow[j] = ct ? 0.0 : Lx_s[j];
#pragma unroll
      for (int i = 0; i < 6; i++) {
        double Lrow[6];
#pragma unroll
        for (int j = 0; j < 6; j++) {
          double c = A[i] * Y[j] + A[6 + i] * Y[6 + j] + A[12 + i] * Y[12 + j];
          if (ct) c += Y[i] * MiY[j] + Y[6 + i] * MiY[6 + j] + Y[12 + i] * MiY[12 + j];
          Lrow[j] = Gs[i][j] - qo.sum(c);
        }
        double ls = 0.0, lx = 0.0, la = 0.0;
#pragma unroll
        for (int j = 0; j < 6; j++) { ls += Lrow[j] * xdd_b[j]; lx += Lrow[j] * xdt_b[j]; la += Lrow[j] * ab0[j]; }
        ls += qo.sum(ct ? 0.0 : Mt_bl[3 * i] * s1_s[0] + Mt_bl[3 * i + 1] * s1_s[1] + Mt_bl[3 * i + 2] * s1_s[2]);
        lx += qo.sum(ct ? 0.0 : Mt_bl[3 * i] * xdt_s[0] + Mt_bl[3 * i + 1] * xdt_s[1] + Mt_bl[3 * i + 2] * xdt_s[2]);
        const double kp = (i < 3) ? P.Kp_body_rpy : P.Kp_body_p, kd = (i < 3) ? P.Kd_body_rpy : P.Kd_body_p;
        const double c1 = LJ_b[i] - ls + kp * xt_b[i] + kd * xdt_b[i];
        met_V += 0.5 * kp * xt_b[i] * xt_b[i] + 0.5 * xdt_b[i] * lx;
        met_Vdot += -kd * xdt_b[i] * xdt_b[i] + xdt_b[i] * c1;
        vconst += lx * ab0[i];
#pragma unroll
        for (int j = 0; j < 3; j++) {
          double sj = 0.0;
#pragma unroll
          for (int k = 0; k < 6; k++) sj += Lrow[k] * B[k][j];
          if (!ct) sj += Mt_bl[3 * i + j];
          blk[i][j] = sw_b * sj;
          vrow[j] += lx * B[i][j];
        }
        brhs[i] = -sw_b * (c1 + la);
      }
    }
    WBC_STAMP(8);
    // rows of sqrt(W) Lambda [B; Sel]: 6 body rows (above) + 3 rows per swing leg.  The first two
    // swing legs share one 12-row Householder append with the body rows (the trot case);
    // further swing legs (nc < 2) go through a second 6-row append.
    int sw0 = -1, sw1 = -1, sw2 = -1, sw3 = -1, ns = 0;
#pragma unroll
    for (int lp = 0; lp < 4; lp++)
      if (!((mask >> lp) & 1u)) {
        if (ns == 0) sw0 = lp; else if (ns == 1) sw1 = lp; else if (ns == 2) sw2 = lp; else sw3 = lp;
        ns++;
      }
    auto swing_rows = [&](int lp, double (*b3)[3], double* r3) {
      // lp is quad-uniform; lp < 0 -> zero rows
      if (lp < 0) {
        for (int i = 0; i < 3; i++) { b3[i][0] = b3[i][1] = b3[i][2] = 0.0; r3[i] = 0.0; }
        return;
      }
      double Mp[18], c1p[3];
#pragma unroll
      for (int i = 0; i < 18; i++) Mp[i] = qo.bcast_d(Mt_bl[i], lp);
#pragma unroll
      for (int i = 0; i < 3; i++) c1p[i] = qo.bcast_d(c1_s[i], lp);
      for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
          double sj = 0.0;
          for (int k = 0; k < 6; k++) sj += Mp[3 * k + i] * B[k][j];
          if (lp == l) sj += Mt_ll[3 * i + j];
          b3[i][j] = sw_f * sj;
        }
        double sj = c1p[i];
        for (int k = 0; k < 6; k++) sj += Mp[3 * k + i] * ab0[k];
        r3[i] = -sw_f * sj;
      }
    };
    swing_rows(sw0, blk + 6, brhs + 6);
    swing_rows(sw1, blk + 9, brhs + 9);
    quad_qr_append<Q, 12>(qo, l, Rc, rhsR, blk, brhs);
    if (ns > 2) {
      double blk2[6][3], brhs2[6];
      swing_rows(sw2, blk2, brhs2);
      swing_rows(sw3, blk2 + 3, brhs2 + 3);
      quad_qr_append<Q, 6>(qo, l, Rc, rhsR, blk2, brhs2);
    }
  }
  WBC_CUT_AT(3, Rc[0][0] + Rc[5][1] + Rc[11][2] + rhsR[0] + rhsR[7] + rhsR[11] + met_V + met_Vdot + vrow[1] + vconst + B[2][2] + ab0[1] + t0l[1])
  WBC_STAMP(9);
  // ---------------- level-2 rows eps (T z + t0): own COLUMNS of the torque map, built here
  //   T[3l'+i][3l+j] = (Y_l' B_l)[i][j] + delta_{l l'} D_l[i][j] ;  t0 replicated
  {
    double blk[12][3], brhs[12];
    double B2[6][3], ab2[6];
    for (int i = 0; i < 6; i++) {
      ab2[i] = st.get(ST_AB0 + i);
      for (int j = 0; j < 3; j++) B2[i][j] = st.get(ST_B + 3 * i + j);
    }
    double Y[18], Dl[9];  // Dl = own diagonal block D_l: -Jl' (contact) or Pm (swing)
    for (int i = 0; i < 18; i++) Y[i] = st.get(ST_Y + i);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Dl[3 * i + j] = ct ? -st.get(ST_JL + 3 * j + i) : st.get(ST_PM + 3 * i + j);
#pragma unroll
    for (int lp = 0; lp < 4; lp++) {
      double Yp[18];
#pragma unroll
      for (int i = 0; i < 18; i++) Yp[i] = qo.bcast_s(Y[i], lp);
#pragma unroll
      for (int i = 0; i < 3; i++) {
        double t0r = qo.bcast_s(t0l[i], lp);
#pragma unroll
        for (int k = 0; k < 6; k++) t0r += Yp[6 * i + k] * ab2[k];
        brhs[3 * lp + i] = -eps * t0r;
#pragma unroll
        for (int j = 0; j < 3; j++) {
          double s = 0.0;
#pragma unroll
          for (int k = 0; k < 6; k++) s += Yp[6 * i + k] * B2[k][j];
          if (lp == l) s += Dl[3 * i + j];
          blk[3 * lp + i][j] = eps * s;
        }
      }
    }
    quad_qr_append<Q, 12>(qo, l, Rc, rhsR, blk, brhs);
  }
  WBC_CUT_AT(4, Rc[0][0] + Rc[5][1] + Rc[11][2] + Rc[3][0] + rhsR[0] + rhsR[7] + rhsR[11] + met_V + met_Vdot + vrow[1] + vconst + B[2][2] + ab0[1] + t0l[1])
  WBC_STAMP(10);
  // ---------------- gather R, unconstrained minimiser, own rows of J = R^-1
  double zl[3], Jr[3][NZ];
  {
    double Rf[NZ][NZ];
#pragma unroll
    for (int lp = 0; lp < 4; lp++)
#pragma unroll
      for (int jj = 0; jj < 3; jj++)
#pragma unroll
        for (int k = 0; k < NZ; k++)
          if (k <= 3 * lp + jj) Rf[k][3 * lp + jj] = qo.bcast_s(Rc[k][jj], lp);
    double invd[NZ];
    double rmax = 0.0, rmin = 0.0;
#pragma unroll
    for (int i = 0; i < NZ; i++) {
      const double a = fabs(Rf[i][i]);
      if (i == 0 || a > rmax) rmax = a;
      if (i == 0 || a < rmin) rmin = a;
      invd[i] = 1.0 / Rf[i][i];
    }
    if (!(rmin > 1e-13 * rmax)) status = ST_SINGULAR;
    if (status == ST_SINGULAR) {
      for (int k = 0; k < 3; k++) out_tau(m.act_inv[3 * l + k], 0.0);
      out_met(0, 0.0); out_met(1, met_err); out_met(2, 0.0); out_met(3, 0.0);
      *iters_out = 0;
      return status;
    }
    double z[NZ];
#pragma unroll
    for (int k = NZ - 1; k >= 0; k--) {
      double s = rhsR[k];
#pragma unroll
      for (int j = k + 1; j < NZ; j++) s -= Rf[k][j] * z[j];
      z[k] = s * invd[k];
    }
#pragma unroll
    for (int k = 0; k < NZ; k++)
#pragma unroll
      for (int jj = 0; jj < 3; jj++)
        if (k == 3 * l + jj) zl[jj] = z[k];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int row = 3 * l + i;
#pragma unroll
      for (int c = 0; c < NZ; c++) {
        double s = (c == row) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < c; k++) s -= Jr[i][k] * Rf[k][c];
        Jr[i][c] = s * invd[c];
      }
    }
  }
  WBC_CUT_AT(5, zl[0] + zl[1] + zl[2] + Jr[0][0] + Jr[1][5] + Jr[2][11] + Jr[0][7] + met_V + met_Vdot + vrow[1] + vconst + B[2][2] + ab0[1] + t0l[1])
  WBC_STAMP(11);
  // ---------------- friction rows
  int iters = 0;
  {
    const double s = sqrt(1.0 + mu * mu);
    int st;
    if (KIND == KIND_PC) {
      // pc_controller.py:14-40,229-237: Vdot <= delta <= 0 with a cost-free delta  <=>  Vdot(z) <= 0
      const double n2 = qo.sum(vrow[0] * vrow[0] + vrow[1] * vrow[1] + vrow[2] * vrow[2]);
      st = quad_gi(qo, l, ct, Jr, zl, mu / s, 1.0 / s, sh, &iters, vrow, vconst + met_Vdot, (n2 > 0.0) ? 1.0 / sqrt(n2) : 0.0);
    } else {
      st = quad_gi(qo, l, ct, Jr, zl, mu / s, 1.0 / s, sh, &iters);
    }
    if (st != ST_OK) status = st;
  }
  *iters_out = iters;
  WBC_CUT_AT(6, zl[0] + zl[1] + zl[2] + (double)iters + met_V + met_Vdot + vrow[1] + vconst + B[2][2] + ab0[1] + t0l[1])
  WBC_STAMP(12);
  // ---------------- outputs: a_b = ab0 + sum_l B_l z_l ;  tau_l = Y_l a_b + D_l z_l + t0l
  {
    double ab[6];
    for (int i = 0; i < 6; i++)
      ab[i] = st.get(ST_AB0 + i) + qo.sum(st.get(ST_B + 3 * i) * zl[0] + st.get(ST_B + 3 * i + 1) * zl[1] + st.get(ST_B + 3 * i + 2) * zl[2]);
    for (int i = 0; i < 3; i++) {
      double s = t0l[i];
      for (int k = 0; k < 6; k++) s += st.get(ST_Y + 6 * i + k) * ab[k];
      for (int j = 0; j < 3; j++) s += (ct ? -st.get(ST_JL + 3 * j + i) : st.get(ST_PM + 3 * i + j)) * zl[j];
      out_tau(m.act_inv[3 * l + i], (status == ST_SINGULAR) ? 0.0 : s);
    }
    // generalized accelerations of the QP solution (rows 4..21 of out_met; see wbc_tick.hpp)
    for (int i = 0; i < 6; i++) out_met(4 + i, ab[i]);
    double t[3], y[3];
    cross(ab, rf, t);
    for (int i = 0; i < 3; i++) y[i] = (ct ? bc[i] : zl[i]) - (ab[3 + i] + t[i]);
    for (int k = 0; k < 3; k++)
      out_met(4 + 6 + m.q_perm[3 * l + k], st.get(ST_JI + 3 * k) * y[0] + st.get(ST_JI + 3 * k + 1) * y[1] + st.get(ST_JI + 3 * k + 2) * y[2]);
  }
  double res = 0.0;
  if (ct) res = fmax(fabs(zl[0]) - mu * zl[2], fabs(zl[1]) - mu * zl[2]);
  res = fmax(0.0, qo.max(res));
  WBC_STAMP(13);
  if (KIND != KIND_ID) {
    met_Vdot += vconst + qo.sum(vrow[0] * zl[0] + vrow[1] * zl[1] + vrow[2] * zl[2]);
    out_met(0, met_V); out_met(1, met_err); out_met(2, 0.0); out_met(3, met_Vdot);
  } else {
    out_met(0, 0.0); out_met(1, met_err); out_met(2, res); out_met(3, 0.0);
  }
  return status;
}

}  // namespace wbc
