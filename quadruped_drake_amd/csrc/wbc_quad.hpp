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
    const double beta = 2.0 / (s2 + v0 * v0);
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
      const bool upd = (3 * l + jj) > k;
      double s = v0 * R[k][jj];
#pragma unroll
      for (int i = 0; i < P; i++) s += col[i] * A[i][jj];
      s *= beta;
      if (upd) {
        R[k][jj] -= s * v0;
#pragma unroll
        for (int i = 0; i < P; i++) A[i][jj] -= s * col[i];
      }
    }
    if (l == owner) R[k][kk] = alpha;
    double s = v0 * rhsR[k];
#pragma unroll
    for (int i = 0; i < P; i++) s += col[i] * rhsA[i];
    s *= beta;
    rhsR[k] -= s * v0;
#pragma unroll
    for (int i = 0; i < P; i++) rhsA[i] -= s * col[i];
  }
}

// Goldfarb-Idnani on the friction rows.  Jr = rows 3l..3l+2 of J (J J' = H^-1), zl = own 3
// entries of z.  Constraint index p = 4*leg + row.
template <class Q>
WBC_HD int quad_gi(Q& qo, int l, bool ct, double (*Jr)[NZ], double* zl, double mu_n, double inv_s, QuadShared& sh,
                   int* iters_out) {
  int q = 0, iters = 0;
  unsigned active = 0u;
  const int maxit = 200;
  for (;;) {
    double zinf = fmax(fabs(zl[0]), fmax(fabs(zl[1]), fabs(zl[2])));
    zinf = qo.max(zinf);
    const double tol = 1e-13 * (1.0 + zinf);
    double sp = -tol;
    int p = -1;
    if (ct) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if ((active >> (4 * l + r)) & 1u) continue;
        const double zc = (r >> 1) ? zl[1] : zl[0];
        const double s = ((r & 1) ? inv_s : -inv_s) * zc + mu_n * zl[2];
        if (s < sp) { sp = s; p = 4 * l + r; }
      }
    }
    qo.argmin(sp, p);
    if (p < 0) { *iters_out = iters; return ST_OK; }
    const int owner = p >> 2, rr = p & 3;
    double npl[3] = {0.0, 0.0, 0.0};
    if (l == owner) {
      const double sg = (rr & 1) ? inv_s : -inv_s;
      npl[0] = (rr >> 1) ? 0.0 : sg;
      npl[1] = (rr >> 1) ? sg : 0.0;
      npl[2] = mu_n;
    }
    sh.u[q] = 0.0;
    for (;;) {
      if (++iters > maxit) { *iters_out = iters; return ST_ITER; }
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
      // r = Rq^-1 d1 (replicated); entries >= q are zero
      double r[NZ];
#pragma unroll
      for (int k = NZ - 1; k >= 0; k--) {
        double s = d[k];
#pragma unroll
        for (int j = k + 1; j < NZ; j++) s -= ((j < q) ? sh.Rq[k][j] : 0.0) * r[j];
        r[k] = (k < q) ? s / sh.Rq[k][k] : 0.0;
      }
      int ldrop = -1;
      double t1n = 0.0, t1d = 1.0;  // t1 = t1n / t1d, compared by cross-multiplication
#pragma unroll
      for (int k = 0; k < NZ; k++) {
        if (k < q && r[k] > 0.0) {
          const double uk = sh.u[k];
          if (ldrop < 0 || uk * t1d < t1n * r[k]) { t1n = uk; t1d = r[k]; ldrop = k; }
        }
      }
      const bool have_t1 = ldrop >= 0;
      const double t1 = have_t1 ? t1n / t1d : 0.0;
      const bool dependent = !(d2n > 1e-22 * dn) || q == NZ;
      double t2 = 0.0;
      if (!dependent) {
        const double znp = qo.sum(zd[0] * npl[0] + zd[1] * npl[1] + zd[2] * npl[2]);
        t2 = -sp / znp;
      }
      if (dependent && !have_t1) { *iters_out = iters; return ST_SINGULAR; }
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
        sh.Rq[q][q] = alpha;
        sh.A[q] = p;
        active |= (1u << p);
        q++;
        break;
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
        const double a = sh.Rq[j][j], bb = sh.Rq[j + 1][j];
        if (bb == 0.0) continue;
        const double h = sqrt(a * a + bb * bb), c = a / h, s = bb / h;
        for (int k = j; k < q; k++) {
          const double x = sh.Rq[j][k], y = sh.Rq[j + 1][k];
          sh.Rq[j][k] = c * x + s * y;
          sh.Rq[j + 1][k] = c * y - s * x;
        }
#pragma unroll
        for (int jj = 0; jj < NZ - 1; jj++) {
          if (jj != j) continue;
#pragma unroll
          for (int i = 0; i < 3; i++) {
            const double x = Jr[i][jj], y = Jr[i][jj + 1];
            Jr[i][jj] = c * x + s * y;
            Jr[i][jj + 1] = c * y - s * x;
          }
        }
      }
      if (!dependent) sp = qo.sum(npl[0] * zl[0] + npl[1] * zl[1] + npl[2] * zl[2]);
    }
  }
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

// The tick.  Every lane of the quad calls this with its own Q (lane id = leg).
// out_tau(k, x): called by lane l for actuator rows k = 3l..3l+2;  out_met: lane 0 only.
template <class Q, int KIND, class In, class OutTau, class OutMet>
WBC_HD int quad_tick(const ModelC& m, const ParamsC& P, Q& qo, In in, unsigned mask, double mu, double mass_scale,
                     QuadShared& sh, OutTau out_tau, OutMet out_met, int* iters_out) {
  const int l = qo.lane();
  const bool ct = (mask >> l) & 1u;
  int status = ST_OK;
  // ---------------- state (replicated on the 4 lanes)
  const double qw = in(0), qx = in(1), qy = in(2), qz = in(3);
  double R0[9];
  {
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
  // ---------------- own leg (lane-local)
  LegKin<double> K;
  LegDyn<double> D;
  double qd[3];
  {
    double sn[3], cs[3];
    for (int k = 0; k < 3; k++) {
      const int row = m.q_perm[3 * l + k];
      const double th = in(7 + row);
      sn[k] = sin(th); cs[k] = cos(th);
      qd[k] = in(25 + row);
    }
    leg_fk(m, l, R0, sn, cs, K);
  }
  double lm = 0.0, lh[3] = {0.0, 0.0, 0.0}, lI[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  {
    double Nb[3], Fb[3];
    leg_rnea<double, true>(m, l, K, w0, qd, gz, D.hl, Nb, Fb, &D);
    for (int i = 0; i < 3; i++) { hb[i] += qo.sum(Nb[i]); hb[3 + i] += qo.sum(Fb[i]); }
    leg_crba(m, l, K, D, lm, lh, lI);
    for (int k = 0; k < 3; k++) {
      const double d[3] = {K.rf[0] - K.r[k][0], K.rf[1] - K.r[k][1], K.rf[2] - K.r[k][2]};
      double c[3];
      cross(K.ax[k], d, c);
      for (int i = 0; i < 3; i++) D.Jl[3 * i + k] = c[i];
    }
    const double det = inv3(D.Jl, D.Ji);
    if (qo.any(!(fabs(det) > 1e-12))) status = ST_SINGULAR;
    for (int i = 0; i < 3; i++) D.pd[i] = v0[i] + D.rd[i];
  }
  const double Mc = bm + qo.sum(lm);
  double Hc[3], Ic[6];
  for (int i = 0; i < 3; i++) Hc[i] = bmc[i] + qo.sum(lh[i]);
  for (int i = 0; i < 6; i++) Ic[i] = bI[i] + qo.sum(lI[i]);

  // ---------------- per-leg reductions
  const double* r = K.rf;
  double X[18], Pm[9], Y[18], bc[3];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 3; j++)
      X[3 * i + j] = D.Mbl[3 * i] * D.Ji[j] + D.Mbl[3 * i + 1] * D.Ji[3 + j] + D.Mbl[3 * i + 2] * D.Ji[6 + j];
  {
    double Mf[9];
    sym_to_full(D.Mll, Mf);
    mm3(Mf, D.Ji, Pm);
  }
  for (int i = 0; i < 3; i++) {
    const double a0 = Pm[3 * i], a1 = Pm[3 * i + 1], a2 = Pm[3 * i + 2];
    Y[6 * i + 0] = D.Mbl[0 * 3 + i] + (a1 * r[2] - a2 * r[1]);
    Y[6 * i + 1] = D.Mbl[1 * 3 + i] + (a2 * r[0] - a0 * r[2]);
    Y[6 * i + 2] = D.Mbl[2 * 3 + i] + (a0 * r[1] - a1 * r[0]);
    Y[6 * i + 3] = D.Mbl[3 * 3 + i] - a0;
    Y[6 * i + 4] = D.Mbl[4 * 3 + i] - a1;
    Y[6 * i + 5] = D.Mbl[5 * 3 + i] - a2;
    bc[i] = ct ? (-P.Kd_contact * D.pd[i] - D.Jdv[i]) : 0.0;
  }
  // G_b = Mbb + sum_l [X [r]x, -X] ;  k = hb + sum_ct X bc
  double Gs[6][6];
  {
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
      Gs[i][0] = Mbb[i][0] + qo.sum(a1 * r[2] - a2 * r[1]);
      Gs[i][1] = Mbb[i][1] + qo.sum(a2 * r[0] - a0 * r[2]);
      Gs[i][2] = Mbb[i][2] + qo.sum(a0 * r[1] - a1 * r[0]);
      Gs[i][3] = Mbb[i][3] - qo.sum(a0);
      Gs[i][4] = Mbb[i][4] - qo.sum(a1);
      Gs[i][5] = Mbb[i][5] - qo.sum(a2);
    }
  }
  double kv[6];
  for (int i = 0; i < 6; i++)
    kv[i] = hb[i] + qo.sum(ct ? (X[3 * i] * bc[0] + X[3 * i + 1] * bc[1] + X[3 * i + 2] * bc[2]) : 0.0);
  // own columns of B and ab0:  G_b [B_l | ab0] = [W_l or -X_l | -k]
  double B[6][3], ab0[6];
  {
    double Ab[6][10];
    for (int i = 0; i < 6; i++) {
      for (int j = 0; j < 6; j++) Ab[i][j] = Gs[i][j];
      Ab[i][9] = -kv[i];
    }
    for (int j = 0; j < 3; j++) {
      double col[6];
      if (ct) {
        const double e[3] = {j == 0 ? 1.0 : 0.0, j == 1 ? 1.0 : 0.0, j == 2 ? 1.0 : 0.0};
        double c[3];
        cross(r, e, c);
        col[0] = c[0]; col[1] = c[1]; col[2] = c[2]; col[3] = e[0]; col[4] = e[1]; col[5] = e[2];
      } else {
        for (int i = 0; i < 6; i++) col[i] = -X[3 * i + j];
      }
      for (int i = 0; i < 6; i++) Ab[i][6 + j] = col[i];
    }
    const double rc = solve6<4>(Ab);
    if (!(rc > 1e-12)) status = ST_SINGULAR;
    for (int i = 0; i < 6; i++) { B[i][0] = Ab[i][6]; B[i][1] = Ab[i][7]; B[i][2] = Ab[i][8]; ab0[i] = Ab[i][9]; }
  }
  // torque map, own COLUMNS: Tc[l'][i][j] = (Y_l' B_l)[i][j] + delta_{l l'} D_l[i][j];  t0 replicated
  double Tc[4][3][3], t0[NZ];
  {
    double t0l[3];
    for (int i = 0; i < 3; i++) {
      double s = D.hl[i];
      for (int j = 0; j < 6; j++) s += Y[6 * i + j] * ab0[j];
      if (ct) s += Pm[3 * i] * bc[0] + Pm[3 * i + 1] * bc[1] + Pm[3 * i + 2] * bc[2];
      t0l[i] = s;
    }
#pragma unroll
    for (int lp = 0; lp < 4; lp++) {
      double Yp[18];
#pragma unroll
      for (int i = 0; i < 18; i++) Yp[i] = qo.bcast_s(Y[i], lp);
#pragma unroll
      for (int i = 0; i < 3; i++) {
        t0[3 * lp + i] = qo.bcast_s(t0l[i], lp);
#pragma unroll
        for (int j = 0; j < 3; j++) {
          double s = 0.0;
#pragma unroll
          for (int k = 0; k < 6; k++) s += Yp[6 * i + k] * B[k][j];
          if (lp == l) s += ct ? -D.Jl[3 * j + i] : Pm[3 * i + j];
          Tc[lp][i][j] = s;
        }
      }
    }
  }

  // ---------------- level-1 rows
  const double eps = sqrt(P.eps2);
  double Rc[NZ][3], rhsR[NZ];
  const double sw_b = sqrt(P.w_body), sw_f = sqrt(P.w_foot);
  double rpy[3], E[9], Ei[9];
  {
    rpy[0] = atan2(R0[7], R0[8]);
    rpy[1] = atan2(-R0[6], sqrt(R0[0] * R0[0] + R0[3] * R0[3]));
    rpy[2] = atan2(R0[3], R0[0]);
    const double sp = sin(rpy[1]), cp = cos(rpy[1]), sy = sin(rpy[2]), cy = cos(rpy[2]);
    E[0] = cp * cy; E[1] = -sy; E[2] = 0.0; E[3] = cp * sy; E[4] = cy; E[5] = 0.0; E[6] = -sp; E[7] = 0.0; E[8] = 1.0;
    const double icp = 1.0 / cp;
    Ei[0] = cy * icp; Ei[1] = sy * icp; Ei[2] = 0.0; Ei[3] = -sy; Ei[4] = cy; Ei[5] = 0.0;
    Ei[6] = cy * sp * icp; Ei[7] = sy * sp * icp; Ei[8] = 1.0;
  }
  double rpyd[3];
  rotv(Ei, w0, rpyd);
  double tg_pb[3], tg_pdb[3], tg_pddb[3], tg_rpy[3], tg_rpyd[3], tg_rpydd[3];
  for (int i = 0; i < 3; i++) {
    tg_pb[i] = in(37 + i); tg_pdb[i] = in(40 + i); tg_pddb[i] = in(43 + i);
    tg_rpy[i] = in(46 + i); tg_rpyd[i] = in(49 + i); tg_rpydd[i] = in(52 + i);
  }
  // own foot targets / errors (zero for a contact leg)
  double xt_s[3], xdt_s[3], xdd_s[3];
  for (int i = 0; i < 3; i++) {
    const double pf = p0[i] + K.rf[i];
    const double tp = in(37 + 18 + 9 * l + i), tpd = in(37 + 21 + 9 * l + i), tpdd = in(37 + 24 + 9 * l + i);
    xt_s[i] = ct ? 0.0 : pf - tp;
    xdt_s[i] = ct ? 0.0 : D.pd[i] - tpd;
    xdd_s[i] = ct ? 0.0 : tpdd;
  }
  double xt_b[6], xdt_b[6];
  for (int i = 0; i < 3; i++) { xt_b[i] = rpy[i] - tg_rpy[i]; xt_b[3 + i] = p0[i] - tg_pb[i]; }
  double met_err = 0.0;
  for (int i = 0; i < 6; i++) met_err += xt_b[i] * xt_b[i];
  met_err += qo.sum(xt_s[0] * xt_s[0] + xt_s[1] * xt_s[1] + xt_s[2] * xt_s[2]);
  double met_V = 0.0, met_Vdot = 0.0;
  double vrow[3] = {0.0, 0.0, 0.0}, vconst = 0.0;  // Vdot += vconst + sum_l vrow_l . z_l

  // diagonal rows: swing leg sqrt(w_foot) (ID) / 0 (MPTC); contact leg eps
  double dval[3], drhs[3];
  for (int i = 0; i < 3; i++) {
    if (ct) { dval[i] = eps; drhs[i] = 0.0; }
    else if (KIND == KIND_ID) {
      const double des = xdd_s[i] - P.Kp_foot * xt_s[i] - P.Kd_foot * xdt_s[i];
      dval[i] = sw_f; drhs[i] = sw_f * (des - D.Jdv[i]);
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

  double blk[6][3], brhs[6];
  if (KIND == KIND_ID) {
    double rpydd_des[3], od[3], ades[6];
    for (int i = 0; i < 3; i++) {
      ades[3 + i] = tg_pddb[i] - P.Kp_body_p * (p0[i] - tg_pb[i]) - P.Kd_body_p * (v0[i] - tg_pdb[i]);
      rpydd_des[i] = tg_rpydd[i] - P.Kp_body_rpy * (rpy[i] - tg_rpy[i]) - P.Kd_body_rpy * (rpyd[i] - tg_rpyd[i]);
    }
    rotv(E, rpydd_des, od);
    for (int i = 0; i < 3; i++) ades[i] = od[i];
    for (int i = 0; i < 6; i++) {
      for (int j = 0; j < 3; j++) blk[i][j] = sw_b * B[i][j];
      brhs[i] = sw_b * (ades[i] - ab0[i]);
    }
    quad_qr_append<Q, 6>(qo, l, Rc, rhsR, blk, brhs);
  } else {
    // ---- MPTC in task coordinates (see wbc_tick.hpp for the derivation)
    double A[18];  // Ji Jfb
    for (int i = 0; i < 3; i++) {
      const double a0 = D.Ji[3 * i], a1 = D.Ji[3 * i + 1], a2 = D.Ji[3 * i + 2];
      A[6 * i + 0] = -(a1 * r[2] - a2 * r[1]);
      A[6 * i + 1] = -(a2 * r[0] - a0 * r[2]);
      A[6 * i + 2] = -(a0 * r[1] - a1 * r[0]);
      A[6 * i + 3] = a0; A[6 * i + 4] = a1; A[6 * i + 5] = a2;
    }
    double Mt_bl[18], Mt_ll[9], Mli[9], MiY[18];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 6; j++) Mt_bl[3 * j + i] = D.Ji[i] * Y[j] + D.Ji[3 + i] * Y[6 + j] + D.Ji[6 + i] * Y[12 + j];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Mt_ll[3 * i + j] = D.Ji[i] * Pm[j] + D.Ji[3 + i] * Pm[3 + j] + D.Ji[6 + i] * Pm[6 + j];
    {
      double Mf[9];
      sym_to_full(D.Mll, Mf);
      inv3(Mf, Mli);
    }
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 6; j++) MiY[6 * i + j] = Mli[3 * i] * Y[j] + Mli[3 * i + 1] * Y[6 + j] + Mli[3 * i + 2] * Y[12 + j];
    double Lbb[6][6];
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) {
        double c = A[i] * Y[j] + A[6 + i] * Y[6 + j] + A[12 + i] * Y[12 + j];
        if (ct) c += Y[i] * MiY[j] + Y[6 + i] * MiY[6 + j] + Y[12 + i] * MiY[12 + j];
        Lbb[i][j] = Gs[i][j] - qo.sum(c);
      }
    double om_rt[3], xdn[3], xddn[3], xdd_b[6];
    rotv(E, rpyd, om_rt);
    rotv(E, tg_rpyd, xdn);
    rotv(E, tg_rpydd, xddn);
    for (int i = 0; i < 3; i++) {
      xdt_b[i] = om_rt[i] - xdn[i];
      xdt_b[3 + i] = v0[i] - tg_pdb[i];
      xdd_b[i] = xddn[i];
      xdd_b[3 + i] = tg_pddb[i];
    }
    // xi (own joints)
    double xi[3];
    {
      double t[3], jfb[3];
      cross(xdt_b, r, t);
      for (int i = 0; i < 3; i++) jfb[i] = xdt_b[3 + i] + t[i];
      if (ct) {
        for (int i = 0; i < 3; i++) {
          double s = 0.0;
          for (int j = 0; j < 6; j++) s += MiY[6 * i + j] * xdt_b[j];
          xi[i] = -s - (D.Ji[3 * i] * jfb[0] + D.Ji[3 * i + 1] * jfb[1] + D.Ji[3 * i + 2] * jfb[2]);
        }
      } else {
        const double y[3] = {xdt_s[0] - jfb[0], xdt_s[1] - jfb[1], xdt_s[2] - jfb[2]};
        rotv(D.Ji, y, xi);
      }
    }
    // C xi = 1/4 [h(v + xi) - h(v - xi)]
    double Cb_base[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, Cb_leg[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, Cl[3] = {0.0, 0.0, 0.0};
    for (int sgi = 0; sgi < 2; sgi++) {
      const double sg = sgi ? -0.25 : 0.25;
      const double wv[3] = {w0[0] + (sgi ? -1.0 : 1.0) * xdt_b[0], w0[1] + (sgi ? -1.0 : 1.0) * xdt_b[1],
                            w0[2] + (sgi ? -1.0 : 1.0) * xdt_b[2]};
      double t2[3], t3[3], Iw_w[3];
      cross(wv, bmc, t2);
      cross(wv, t2, t2);
      symv(bI, wv, Iw_w);
      cross(wv, Iw_w, t3);
      for (int i = 0; i < 3; i++) { Cb_base[i] += sg * t3[i]; Cb_base[3 + i] += sg * t2[i]; }
      const double qv[3] = {qd[0] + (sgi ? -1.0 : 1.0) * xi[0], qd[1] + (sgi ? -1.0 : 1.0) * xi[1],
                            qd[2] + (sgi ? -1.0 : 1.0) * xi[2]};
      double hl2[3], Nb[3], Fb[3];
      leg_rnea<double, false>(m, l, K, wv, qv, 0.0, hl2, Nb, Fb, (LegDyn<double>*)nullptr);
      for (int i = 0; i < 3; i++) { Cb_leg[i] += sg * Nb[i]; Cb_leg[3 + i] += sg * Fb[i]; Cl[i] += sg * hl2[i]; }
    }
    // Lambda (J Minv C xi): base rows replicated, own swing rows local
    double LJ_b[6], LJ_s[3];
    {
      double gl[3], c[3];
      for (int i = 0; i < 3; i++) gl[i] = D.Ji[i] * Cl[0] + D.Ji[3 + i] * Cl[1] + D.Ji[6 + i] * Cl[2];
      cross(r, gl, c);
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
      cross(xdt_b, D.rd, t);
      for (int i = 0; i < 3; i++) {
        const double jx = t[i] + D.Jd[3 * i] * xi[0] + D.Jd[3 * i + 1] * xi[1] + D.Jd[3 * i + 2] * xi[2];
        s1_s[i] = ct ? 0.0 : xdd_s[i] - D.Jdv[i] + jx;
      }
    }
    auto lam_mul = [&](const double* yb, const double* ys, double* ob, double* os) {
      for (int i = 0; i < 6; i++) {
        double s = 0.0;
        for (int j = 0; j < 6; j++) s += Lbb[i][j] * yb[j];
        const double loc = ct ? 0.0 : Mt_bl[3 * i] * ys[0] + Mt_bl[3 * i + 1] * ys[1] + Mt_bl[3 * i + 2] * ys[2];
        ob[i] = s + qo.sum(loc);
      }
      for (int i = 0; i < 3; i++) {
        double s = Mt_ll[3 * i] * ys[0] + Mt_ll[3 * i + 1] * ys[1] + Mt_ll[3 * i + 2] * ys[2];
        for (int j = 0; j < 6; j++) s += Mt_bl[3 * j + i] * yb[j];
        os[i] = ct ? 0.0 : s;
      }
    };
    double Ls_b[6], Ls_s[3], c1_b[6], c1_s[3];
    lam_mul(xdd_b, s1_s, Ls_b, Ls_s);
    for (int i = 0; i < 6; i++) {
      const double kp = (i < 3) ? P.Kp_body_rpy : P.Kp_body_p, kd = (i < 3) ? P.Kd_body_rpy : P.Kd_body_p;
      c1_b[i] = LJ_b[i] - Ls_b[i] + kp * xt_b[i] + kd * xdt_b[i];
      met_V += 0.5 * kp * xt_b[i] * xt_b[i];
      met_Vdot += -kd * xdt_b[i] * xdt_b[i] + xdt_b[i] * c1_b[i];
    }
    {
      double lv = 0.0, lvd = 0.0;
      for (int i = 0; i < 3; i++) {
        c1_s[i] = LJ_s[i] - Ls_s[i] + P.Kp_foot * xt_s[i] + P.Kd_foot * xdt_s[i];
        lv += 0.5 * P.Kp_foot * xt_s[i] * xt_s[i];
        lvd += -P.Kd_foot * xdt_s[i] * xdt_s[i] + xdt_s[i] * c1_s[i];
      }
      double Lx_b[6], Lx_s[3];
      lam_mul(xdt_b, xdt_s, Lx_b, Lx_s);
      for (int i = 0; i < 6; i++) met_V += 0.5 * xdt_b[i] * Lx_b[i];
      for (int i = 0; i < 3; i++) lv += 0.5 * xdt_s[i] * Lx_s[i];
      met_V += qo.sum(lv);
      met_Vdot += qo.sum(lvd);
      for (int j = 0; j < 3; j++) {
        double s = 0.0;
        for (int k = 0; k < 6; k++) s += Lx_b[k] * B[k][j];
        vrow[j] = s + (ct ? 0.0 : Lx_s[j]);
      }
      for (int k = 0; k < 6; k++) vconst += Lx_b[k] * ab0[k];
    }
    // body rows of sqrt(W) Lambda [B; Sel]
    for (int i = 0; i < 6; i++) {
      for (int j = 0; j < 3; j++) {
        double s = 0.0;
        for (int k = 0; k < 6; k++) s += Lbb[i][k] * B[k][j];
        if (!ct) s += Mt_bl[3 * i + j];
        blk[i][j] = sw_b * s;
      }
      double s = c1_b[i];
      for (int k = 0; k < 6; k++) s += Lbb[i][k] * ab0[k];
      brhs[i] = -sw_b * s;
    }
    quad_qr_append<Q, 6>(qo, l, Rc, rhsR, blk, brhs);
    // swing rows: 3 per swing leg lp
#pragma unroll
    for (int lp = 0; lp < 4; lp++) {
      if ((mask >> lp) & 1u) continue;  // quad-uniform
      double Mp[18], c1p[3];
#pragma unroll
      for (int i = 0; i < 18; i++) Mp[i] = qo.bcast_s(Mt_bl[i], lp);
#pragma unroll
      for (int i = 0; i < 3; i++) c1p[i] = qo.bcast_s(c1_s[i], lp);
      double b3[3][3], r3[3];
      for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
          double s = 0.0;
          for (int k = 0; k < 6; k++) s += Mp[3 * k + i] * B[k][j];
          if (lp == l) s += Mt_ll[3 * i + j];
          b3[i][j] = sw_f * s;
        }
        double s = c1p[i];
        for (int k = 0; k < 6; k++) s += Mp[3 * k + i] * ab0[k];
        r3[i] = -sw_f * s;
      }
      quad_qr_append<Q, 3>(qo, l, Rc, rhsR, b3, r3);
    }
  }
  // ---------------- level-2 rows eps (T z + t0)
#pragma unroll
  for (int h = 0; h < 2; h++) {
#pragma unroll
    for (int i = 0; i < 6; i++) {
      const int row = 6 * h + i;
#pragma unroll
      for (int j = 0; j < 3; j++) blk[i][j] = eps * Tc[row / 3][row % 3][j];
      brhs[i] = -eps * t0[row];
    }
    quad_qr_append<Q, 6>(qo, l, Rc, rhsR, blk, brhs);
  }
  // ---------------- gather R, unconstrained minimiser, own rows of J = R^-1
  double Rf[NZ][NZ];
#pragma unroll
  for (int lp = 0; lp < 4; lp++)
#pragma unroll
    for (int jj = 0; jj < 3; jj++)
#pragma unroll
      for (int k = 0; k < NZ; k++)
        if (k <= 3 * lp + jj) Rf[k][3 * lp + jj] = qo.bcast_s(Rc[k][jj], lp);
  double invd[NZ];
  {
    double rmax = 0.0, rmin = 0.0;
#pragma unroll
    for (int i = 0; i < NZ; i++) {
      const double a = fabs(Rf[i][i]);
      if (i == 0 || a > rmax) rmax = a;
      if (i == 0 || a < rmin) rmin = a;
      invd[i] = 1.0 / Rf[i][i];
    }
    if (!(rmin > 1e-13 * rmax)) status = ST_SINGULAR;
  }
  if (status == ST_SINGULAR) {
    for (int k = 0; k < 3; k++) out_tau(3 * l + k, 0.0);
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
  double zl[3];
#pragma unroll
  for (int k = 0; k < NZ; k++)
#pragma unroll
    for (int jj = 0; jj < 3; jj++)
      if (k == 3 * l + jj) zl[jj] = z[k];
  double Jr[3][NZ];
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
  // ---------------- friction rows
  int iters = 0;
  {
    const double s = sqrt(1.0 + mu * mu);
    const int st = quad_gi(qo, l, ct, Jr, zl, mu / s, 1.0 / s, sh, &iters);
    if (st != ST_OK) status = st;
  }
  *iters_out = iters;
  // ---------------- outputs
  double tauc[NZ];
#pragma unroll
  for (int lp = 0; lp < 4; lp++)
#pragma unroll
    for (int i = 0; i < 3; i++)
      tauc[3 * lp + i] = t0[3 * lp + i] + qo.sum(Tc[lp][i][0] * zl[0] + Tc[lp][i][1] * zl[1] + Tc[lp][i][2] * zl[2]);
  for (int k = 0; k < 3; k++) {
    const int idx = m.act_perm[3 * l + k];
    double v = 0.0;
#pragma unroll
    for (int c = 0; c < NZ; c++) v = (c == idx) ? tauc[c] : v;
    out_tau(3 * l + k, (status == ST_SINGULAR) ? 0.0 : v);
  }
  double res = 0.0;
  if (ct) res = fmax(fabs(zl[0]) - mu * zl[2], fabs(zl[1]) - mu * zl[2]);
  res = fmax(0.0, qo.max(res));
  if (KIND == KIND_MPTC) {
    met_Vdot += vconst + qo.sum(vrow[0] * zl[0] + vrow[1] * zl[1] + vrow[2] * zl[2]);
    out_met(0, met_V); out_met(1, met_err); out_met(2, 0.0); out_met(3, met_Vdot);
  } else {
    out_met(0, 0.0); out_met(1, met_err); out_met(2, res); out_met(3, 0.0);
  }
  return status;
}

}  // namespace wbc
