// wbc_scalar_tick.hpp -- TEST / ANALYSIS code, not product code (moved out of the product translation unit in round 3).
// The retired one-lane-per-robot form of the tick (round 1's first kernel): dense 12-variable QR + Goldfarb-Idnani and the
// whole tick written for ONE scalar type T.  It is instantiated on the host only -- with `double` as an independent second
// derivation of the reduced formulation for the tests, and with an operation-counting scalar for the frozen flops/tick
// figure (tools/host_tick.cpp).  The product kernels (csrc/wbc_hex.hpp) share only the per-leg math of csrc/wbc_tick.hpp.
// Known limit: with a knee angle below ~0.1 rad this form loses agreement with the oracle on some states (the 16-lane
// product form agrees down to 1e-8 rad, profiles/r03/singular_envelope.md); the tests use it on the regular configurations.
#pragma once
#include "../quadruped_drake_amd/csrc/wbc_tick.hpp"

namespace wbc {

// ---------------------------------------------------------------- QR + Goldfarb-Idnani (n = 12)

// Fold a block of `p` dense rows A[p][13] (12 coefficients + rhs) into the upper-triangular
// factor R[12][13] by Householder reflections on [R_kk; A_0k..A_pk].
template <class T, int NV> WBC_HD void qr_append(T (*R)[NV + 1], T (*A)[NV + 1], int p) {
  for (int k = 0; k < NV; k++) {
    T s2 = T(0.0);
    for (int i = 0; i < p; i++) s2 = s2 + A[i][k] * A[i][k];
    if (!(s2 > T(0.0))) continue;
    T rkk = R[k][k];
    T nrm = sqrt(rkk * rkk + s2);
    T alpha = (rkk > T(0.0)) ? T(0.0) - nrm : nrm;
    T v0 = rkk - alpha;            // Householder vector [v0; A[:,k]]
    T beta = T(1.0) / (s2 + v0 * v0) * T(2.0);
    R[k][k] = alpha;
    for (int j = k + 1; j <= NV; j++) {
      T s = v0 * R[k][j];
      for (int i = 0; i < p; i++) s = s + A[i][k] * A[i][j];
      s = s * beta;
      R[k][j] = R[k][j] - s * v0;
      for (int i = 0; i < p; i++) A[i][j] = A[i][j] - s * A[i][k];
    }
  }
}

// Constraint i (0..15: friction row i%4 of leg i/4; 16..39: torque box) as a unit normal over z
// and offset: n.z >= b.
template <class T> struct QpCons {
  T fr[4][3];         // per contact-slot... (unused entries zero)
  T mu_n, inv_s;      // mu / sqrt(1+mu^2), 1/sqrt(1+mu^2)
  const T* Trow;      // torque map rows [12][tstride]: coefficients then t0 (torque box; nullable)
  int tstride;
  T tau_max;
  T tnorm[12];
  unsigned mask;
  const T* pcrow;  // PC law: Vdot(z) = pcrow[0..11].z + pcrow[12] <= 0 (nullable)
  T pc_inv;        // 1 / |pcrow[0..11]|
  const T* clfrow; // CLF law: clfrow[0..12].[z; delta] <= clfrow[13], already normalised (nullable)
};

template <class T, int NV> WBC_HD void cons_normal(const QpCons<T>& C, int i, T* n, T& b) {
  for (int k = 0; k < NV; k++) n[k] = T(0.0);
  if (i < 16) {
    int l = i >> 2, r = i & 3;
    // rows: +fx - mu fz <= 0, -fx - mu fz <= 0, +fy - mu fz <= 0, -fy - mu fz <= 0  ->  n.z >= 0
    int comp = r >> 1;
    T sg = (r & 1) ? C.inv_s : T(0.0) - C.inv_s;
    n[3 * l + comp] = sg;
    n[3 * l + 2] = C.mu_n;
    b = T(0.0);
  } else if (i == PC_ROW) {
    for (int k = 0; k < NZ; k++) n[k] = T(0.0) - C.pcrow[k] * C.pc_inv;
    b = C.pcrow[NZ] * C.pc_inv;
  } else if (i == CLF_ROW) {
    for (int k = 0; k < NV; k++) n[k] = T(0.0) - C.clfrow[k];
    b = T(0.0) - C.clfrow[NV];
  } else {
    int j = (i - 16) >> 1;
    T sg = ((i - 16) & 1) ? T(1.0) : T(-1.0);   // even: tau_j <= tau_max -> -T_j z >= t0_j - tau_max
    T inv = T(1.0) / C.tnorm[j];
    const T* row = C.Trow + j * C.tstride;
    for (int k = 0; k < NZ; k++) n[k] = sg * row[k] * inv;
    b = (T(0.0) - sg * row[C.tstride - 1] - C.tau_max) * inv;
  }
}

// Goldfarb-Idnani.  J = R^-1 (12x12), z = unconstrained minimiser.  `elig` = bitmask of
// constraints that exist.  Returns status, iteration count in *iters.
template <class T, int NV>
WBC_HD int gi_solve(T (*J)[NV], T* z, const QpCons<T>& C, unsigned long long elig, int* iters_out) {
  int A[NV], q = 0;
  unsigned long long active = 0ull;
  T u[NV + 1], Rq[NV][NV], d[NV], zd[NV], r[NV], np[NV];
  int iters = 0;
  const int maxit = 200;
  for (;;) {
    T zinf = T(0.0);
    for (int i = 0; i < NV; i++) { T a = wabs(z[i]); if (a > zinf) zinf = a; }
    T tol = T(1e-13) * (T(1.0) + zinf);
    int p = -1;
    T sp = T(0.0) - tol, bp = T(0.0);
    for (int i = 0; i < MAXC; i++) {
      if (!((elig >> i) & 1ull) || ((active >> i) & 1ull)) continue;
      T n[NV], b;
      cons_normal<T, NV>(C, i, n, b);
      T s = T(0.0) - b;
      for (int k = 0; k < NV; k++) s = s + n[k] * z[k];
      if (s < sp) { sp = s; p = i; }
    }
    if (p < 0) { *iters_out = iters; return ST_OK; }
    cons_normal<T, NV>(C, p, np, bp);
    u[q] = T(0.0);
    for (;;) {
      if (++iters > maxit) { *iters_out = iters; return ST_ITER; }
      T dn = T(0.0), d2n = T(0.0);
      for (int k = 0; k < NV; k++) {
        T s = T(0.0);
        for (int i = 0; i < NV; i++) s = s + J[i][k] * np[i];
        d[k] = s;
        dn = dn + s * s;
        if (k >= q) d2n = d2n + s * s;
      }
      for (int i = 0; i < NV; i++) {
        T s = T(0.0);
        for (int k = q; k < NV; k++) s = s + J[i][k] * d[k];
        zd[i] = s;
      }
      for (int k = q - 1; k >= 0; k--) {
        T s = d[k];
        for (int j = k + 1; j < q; j++) s = s - Rq[k][j] * r[j];
        r[k] = s / Rq[k][k];
      }
      int l = -1;
      bool have_t1 = false;
      T t1 = T(0.0);
      for (int k = 0; k < q; k++)
        if (r[k] > T(0.0)) {
          T c = u[k] / r[k];
          if (!have_t1 || c < t1) { t1 = c; l = k; have_t1 = true; }
        }
      bool dependent = !(d2n > T(1e-22) * dn) || q == NV;
      T t2 = T(0.0);
      if (!dependent) {
        T znp = T(0.0);
        for (int i = 0; i < NV; i++) znp = znp + zd[i] * np[i];
        t2 = (T(0.0) - sp) / znp;
      }
      if (dependent && !have_t1) { *iters_out = iters; return ST_SINGULAR; }
      bool full = !dependent && (!have_t1 || !(t1 < t2));
      T t = full ? t2 : t1;
      for (int k = 0; k < q; k++) u[k] = u[k] - t * r[k];
      u[q] = u[q] + t;
      if (!dependent)
        for (int i = 0; i < NV; i++) z[i] = z[i] + t * zd[i];
      if (full) {
        for (int j = NV - 1; j > q; j--) {
          T a = d[j - 1], bb = d[j];
          if (bb == T(0.0)) continue;
          T h = sqrt(a * a + bb * bb), c = a / h, s = bb / h;
          d[j - 1] = h; d[j] = T(0.0);
          for (int i = 0; i < NV; i++) {
            T x = J[i][j - 1], y = J[i][j];
            J[i][j - 1] = c * x + s * y;
            J[i][j] = c * y - s * x;
          }
        }
        for (int k = 0; k <= q; k++) Rq[k][q] = d[k];
        A[q] = p;
        active |= (1ull << p);
        q++;
        break;
      }
      active &= ~(1ull << A[l]);
      for (int j = l; j < q - 1; j++) {
        A[j] = A[j + 1];
        u[j] = u[j + 1];
        for (int k = 0; k <= j + 1; k++) Rq[k][j] = Rq[k][j + 1];
      }
      u[q - 1] = u[q];
      q--;
      u[q + 1] = T(0.0);
      for (int j = l; j < q; j++) {
        T a = Rq[j][j], bb = Rq[j + 1][j];
        if (bb == T(0.0)) continue;
        T h = sqrt(a * a + bb * bb), c = a / h, s = bb / h;
        for (int k = j; k < q; k++) {
          T x = Rq[j][k], y = Rq[j + 1][k];
          Rq[j][k] = c * x + s * y;
          Rq[j + 1][k] = c * y - s * x;
        }
        for (int i = 0; i < NV; i++) {
          T x = J[i][j], y = J[i][j + 1];
          J[i][j] = c * x + s * y;
          J[i][j + 1] = c * y - s * x;
        }
      }
      if (!dependent) {
        sp = T(0.0) - bp;
        for (int k = 0; k < NV; k++) sp = sp + np[k] * z[k];
      }
    }
  }
}

// 6x6 LU with partial pivoting, in place; piv[6].  Returns smallest |pivot| / largest |pivot|.
template <class T> WBC_HD T lu6(T (*A)[6], int* piv) {
  T pmin = T(0.0), pmax = T(0.0);
  for (int c = 0; c < 6; c++) {
    int p = c;
    T best = wabs(A[c][c]);
    for (int r = c + 1; r < 6; r++) { T a = wabs(A[r][c]); if (a > best) { best = a; p = r; } }
    piv[c] = p;
    if (p != c)
      for (int j = 0; j < 6; j++) { T t = A[c][j]; A[c][j] = A[p][j]; A[p][j] = t; }
    if (c == 0 || best < pmin) pmin = best;
    if (best > pmax) pmax = best;
    T id = T(1.0) / A[c][c];
    for (int r = c + 1; r < 6; r++) {
      T f = A[r][c] * id;
      A[r][c] = f;
      for (int j = c + 1; j < 6; j++) A[r][j] = A[r][j] - f * A[c][j];
    }
  }
  return pmin / pmax;
}
template <class T> WBC_HD void lu6_solve(const T (*A)[6], const int* piv, T* b) {
  for (int c = 0; c < 6; c++) {
    if (piv[c] != c) { T t = b[c]; b[c] = b[piv[c]]; b[piv[c]] = t; }
    for (int r = c + 1; r < 6; r++) b[r] = b[r] - A[r][c] * b[c];
  }
  for (int c = 5; c >= 0; c--) {
    T s = b[c];
    for (int j = c + 1; j < 6; j++) s = s - A[c][j] * b[j];
    b[c] = s / A[c][c];
  }
}

// ---------------------------------------------------------------- the tick
// Accessors: in(i) returns input row i of this robot (q rows 0..18, v rows 19..36, targets rows
// 37..90); outputs are written through out_tau(k, value) / out_metric(k, value).
template <class T, int KIND, class In, class OutTau, class OutMet>
WBC_HD int tick(const ModelC& m, const ParamsC& P, In in, unsigned mask, T mu, T mass_scale, OutTau out_tau,
                OutMet out_met, int* iters_out) {
  constexpr int NV = (KIND == KIND_CLF) ? NZ + 1 : NZ;  // reduced variables: z (12) [+ delta for CLF]
  int status = ST_OK;
  bool illc = false;   // MPTC / PC: a nearly straight knee (status 3)
  // ---- state
  T qw = in(0), qx = in(1), qy = in(2), qz = in(3);
  T R0[9];
  {
    T s = T(2.0) / (qw * qw + qx * qx + qy * qy + qz * qz);
    R0[0] = T(1.0) - s * (qy * qy + qz * qz); R0[1] = s * (qx * qy - qw * qz); R0[2] = s * (qx * qz + qw * qy);
    R0[3] = s * (qx * qy + qw * qz); R0[4] = T(1.0) - s * (qx * qx + qz * qz); R0[5] = s * (qy * qz - qw * qx);
    R0[6] = s * (qx * qz - qw * qy); R0[7] = s * (qy * qz + qw * qx); R0[8] = T(1.0) - s * (qx * qx + qy * qy);
  }
  T p0[3] = {in(4), in(5), in(6)};
  T w0[3] = {in(19), in(20), in(21)};
  T v0[3] = {in(22), in(23), in(24)};
  T gz = T(m.gravity);
  int nc = 0;
  for (int l = 0; l < 4; l++) nc += (mask >> l) & 1;

  LegKin<T> K[4];
  LegDyn<T> D[4];
  T qd[4][3];
  // ---- base body
  T bm = T(m.base_mass) * mass_scale;
  T bmc_l[3] = {T(m.base_mc[0]) * mass_scale, T(m.base_mc[1]) * mass_scale, T(m.base_mc[2]) * mass_scale};
  T bmc[3], bI[6];
  rotv(R0, bmc_l, bmc);
  rot_inertia(R0, m.base_I, bI);
  for (int i = 0; i < 6; i++) bI[i] = bI[i] * mass_scale;
  // composite inertia of the whole robot at the base origin -> Mbb; base wrench -> hb
  T Mc = bm, Hc[3] = {bmc[0], bmc[1], bmc[2]}, Ic[6] = {bI[0], bI[1], bI[2], bI[3], bI[4], bI[5]};
  T hb[6];
  {
    T t2[3], t3[3], Iw_w[3], g3[3] = {T(0.0), T(0.0), gz}, t4[3];
    cross(w0, bmc, t2);
    cross(w0, t2, t2);
    symv(bI, w0, Iw_w);
    cross(w0, Iw_w, t3);
    cross(bmc, g3, t4);
    for (int i = 0; i < 3; i++) { hb[i] = t3[i] + t4[i]; hb[3 + i] = bm * g3[i] + t2[i]; }
  }
  for (int l = 0; l < 4; l++) {
    T sn[3], cs[3];
    for (int k = 0; k < 3; k++) {
      int row = m.q_perm[3 * l + k];
      T th = in(7 + row);
      wbc_sincos(th, sn[k], cs[k]);
      qd[l][k] = in(25 + row);
    }
    if ((KIND == KIND_ID || KIND == KIND_CLF) && !((mask >> l) & 1u)) knee_clamp(sn[2], cs[2]);   // swing legs of the ID-type laws
    if ((KIND == KIND_MPTC || KIND == KIND_PC) && wabs(sn[2]) < T(KNEE_ILLCOND)) illc = true;
    leg_fk(m, l, R0, sn, cs, K[l]);
    T Nb[3], Fb[3];
    const T mass3[3] = {T(m.link[l][0].mass), T(m.link[l][1].mass), T(m.link[l][2].mass)};
    leg_rnea<T, true>(mass3, K[l], w0, qd[l], gz, D[l].hl, Nb, Fb, &D[l]);
    for (int i = 0; i < 3; i++) { hb[i] = hb[i] + Nb[i]; hb[3 + i] = hb[3 + i] + Fb[i]; }
    leg_crba(mass3, K[l], D[l], Mc, Hc, Ic);
    // foot Jacobian block wrt own joints
    for (int k = 0; k < 3; k++) {
      T d[3] = {K[l].rf(0) - K[l].r(k, 0), K[l].rf(1) - K[l].r(k, 1), K[l].rf(2) - K[l].r(k, 2)}, c[3];
      T axv[3] = {K[l].ax(k, 0), K[l].ax(k, 1), K[l].ax(k, 2)};
      cross(axv, d, c);
      for (int i = 0; i < 3; i++) D[l].Jl[3 * i + k] = c[i];
    }
    T det = inv3(D[l].Jl, D[l].Ji);
    if (!(wabs(det) > T(1e-12))) status = ST_SINGULAR;
    for (int i = 0; i < 3; i++) D[l].pd[i] = v0[i] + D[l].rd[i];
  }
  // Mbb (6x6): [[Ic, [Hc]x], [[Hc]x', Mc 1]]
  T Gb[6][6];
  {
    T Mbb[6][6];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) Mbb[i][j] = T(0.0);
    Mbb[0][0] = Ic[0]; Mbb[1][1] = Ic[1]; Mbb[2][2] = Ic[2];
    Mbb[0][1] = Mbb[1][0] = Ic[3]; Mbb[0][2] = Mbb[2][0] = Ic[4]; Mbb[1][2] = Mbb[2][1] = Ic[5];
    Mbb[0][4] = T(0.0) - Hc[2]; Mbb[0][5] = Hc[1];
    Mbb[1][3] = Hc[2];          Mbb[1][5] = T(0.0) - Hc[0];
    Mbb[2][3] = T(0.0) - Hc[1]; Mbb[2][4] = Hc[0];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Mbb[3 + j][i] = Mbb[i][3 + j];
    Mbb[3][3] = Mc; Mbb[4][4] = Mc; Mbb[5][5] = Mc;
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) Gb[i][j] = Mbb[i][j];
  }
  // ---- task-space quantities shared by both laws (reference :156-197 / :156-257)
  T rpy[3], E[9], Ei[9];
  {
    rpy[0] = atan2(R0[7], R0[8]);
    rpy[1] = atan2(T(0.0) - R0[6], sqrt(R0[0] * R0[0] + R0[3] * R0[3]));
    rpy[2] = atan2(R0[3], R0[0]);
    T sp = sin(rpy[1]), cp = cos(rpy[1]), sy = sin(rpy[2]), cy = cos(rpy[2]);
    E[0] = cp * cy; E[1] = T(0.0) - sy; E[2] = T(0.0);
    E[3] = cp * sy; E[4] = cy;          E[5] = T(0.0);
    E[6] = T(0.0) - sp; E[7] = T(0.0);  E[8] = T(1.0);
    T icp = T(1.0) / cp;
    Ei[0] = cy * icp; Ei[1] = sy * icp; Ei[2] = T(0.0);
    Ei[3] = T(0.0) - sy; Ei[4] = cy; Ei[5] = T(0.0);
    Ei[6] = cy * sp * icp; Ei[7] = sy * sp * icp; Ei[8] = T(1.0);
  }
  T rpyd[3];
  rotv(Ei, w0, rpyd);
  // targets
  T tg_pb[3], tg_pdb[3], tg_pddb[3], tg_rpy[3], tg_rpyd[3], tg_rpydd[3];
  for (int i = 0; i < 3; i++) {
    tg_pb[i] = in(37 + i); tg_pdb[i] = in(40 + i); tg_pddb[i] = in(43 + i);
    tg_rpy[i] = in(46 + i); tg_rpyd[i] = in(49 + i); tg_rpydd[i] = in(52 + i);
  }

  // ---- per-leg reductions: X_l = Mbl Ji, P_l = Mll Ji, Y_l = Mbl' - P_l Jfb_l, G_b, k
  T X[4][18], Pm[4][9], Y[4][18], bc[4][3];
  T kvec[6];
  for (int i = 0; i < 6; i++) kvec[i] = hb[i];
  for (int l = 0; l < 4; l++) {
    const T* r = &K[l].rf(0);
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 3; j++)
        X[l][3 * i + j] = D[l].Mbl[3 * i] * D[l].Ji[j] + D[l].Mbl[3 * i + 1] * D[l].Ji[3 + j] + D[l].Mbl[3 * i + 2] * D[l].Ji[6 + j];
    T Mf[9];
    sym_to_full(D[l].Mll, Mf);
    mm3(Mf, D[l].Ji, Pm[l]);
    // Jfb = [-[r]x, 1]:  (A Jfb) = [ -A [r]x , A ];  -A[r]x column j = (A (e_j x r))... use rows: (A[r]x)_{ij}
    // [r]x = [[0,-r2,r1],[r2,0,-r0],[-r1,r0,0]]
    for (int i = 0; i < 6; i++) {
      T a0 = X[l][3 * i], a1 = X[l][3 * i + 1], a2 = X[l][3 * i + 2];
      // (X [r]x)_i = [a1 r2 - a2 r1, a2 r0 - a0 r2, a0 r1 - a1 r0]
      Gb[i][0] = Gb[i][0] + (a1 * r[2] - a2 * r[1]);
      Gb[i][1] = Gb[i][1] + (a2 * r[0] - a0 * r[2]);
      Gb[i][2] = Gb[i][2] + (a0 * r[1] - a1 * r[0]);
      Gb[i][3] = Gb[i][3] - a0; Gb[i][4] = Gb[i][4] - a1; Gb[i][5] = Gb[i][5] - a2;
    }
    for (int i = 0; i < 3; i++) {
      T a0 = Pm[l][3 * i], a1 = Pm[l][3 * i + 1], a2 = Pm[l][3 * i + 2];
      Y[l][6 * i + 0] = D[l].Mbl[0 * 3 + i] + (a1 * r[2] - a2 * r[1]);
      Y[l][6 * i + 1] = D[l].Mbl[1 * 3 + i] + (a2 * r[0] - a0 * r[2]);
      Y[l][6 * i + 2] = D[l].Mbl[2 * 3 + i] + (a0 * r[1] - a1 * r[0]);
      Y[l][6 * i + 3] = D[l].Mbl[3 * 3 + i] - a0;
      Y[l][6 * i + 4] = D[l].Mbl[4 * 3 + i] - a1;
      Y[l][6 * i + 5] = D[l].Mbl[5 * 3 + i] - a2;
    }
    bool ct = (mask >> l) & 1;
    for (int i = 0; i < 3; i++) bc[l][i] = ct ? (T(0.0) - T(P.Kd_contact) * D[l].pd[i] - D[l].Jdv[i]) : T(0.0);
    if (ct)
      for (int i = 0; i < 6; i++)
        kvec[i] = kvec[i] + X[l][3 * i] * bc[l][0] + X[l][3 * i + 1] * bc[l][1] + X[l][3 * i + 2] * bc[l][2];
  }
  // ---- a_b = sum_l B_l z_l + ab0 with  G_b a_b = sum_ct W_l z_l - sum_sw X_l z_l - k
  int piv[6];
  T Gs[6][6];
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) Gs[i][j] = Gb[i][j];
  T rc = lu6(Gb, piv);
  if (!(rc > T(1e-12))) status = ST_SINGULAR;
  T B[6][NZ], ab0[6];
  for (int i = 0; i < 6; i++) ab0[i] = T(0.0) - kvec[i];
  lu6_solve(Gb, piv, ab0);
  for (int l = 0; l < 4; l++) {
    bool ct = (mask >> l) & 1;
    const T* r = &K[l].rf(0);
    for (int j = 0; j < 3; j++) {
      T col[6];
      if (ct) {  // W_l e_j = [r x e_j; e_j]
        T e[3] = {T(j == 0 ? 1.0 : 0.0), T(j == 1 ? 1.0 : 0.0), T(j == 2 ? 1.0 : 0.0)}, c[3];
        cross(r, e, c);
        col[0] = c[0]; col[1] = c[1]; col[2] = c[2]; col[3] = e[0]; col[4] = e[1]; col[5] = e[2];
      } else {
        for (int i = 0; i < 6; i++) col[i] = T(0.0) - X[l][3 * i + j];
      }
      lu6_solve(Gb, piv, col);
      for (int i = 0; i < 6; i++) B[i][3 * l + j] = col[i];
    }
  }
  // ---- torque map tau = Tm z + t0 (canonical joint order), rows 3l..3l+2
  T Tm[NZ][NV + 1];
  for (int i = 0; i < NZ; i++) for (int c = NZ; c < NV; c++) Tm[i][c] = T(0.0);
  for (int l = 0; l < 4; l++) {
    bool ct = (mask >> l) & 1;
    for (int i = 0; i < 3; i++) {
      for (int c = 0; c < NZ; c++) {
        T s = T(0.0);
        for (int j = 0; j < 6; j++) s = s + Y[l][6 * i + j] * B[j][c];
        Tm[3 * l + i][c] = s;
      }
      T s = D[l].hl[i];
      for (int j = 0; j < 6; j++) s = s + Y[l][6 * i + j] * ab0[j];
      if (ct) s = s + Pm[l][3 * i] * bc[l][0] + Pm[l][3 * i + 1] * bc[l][1] + Pm[l][3 * i + 2] * bc[l][2];
      Tm[3 * l + i][NV] = s;
      for (int j = 0; j < 3; j++)
        Tm[3 * l + i][3 * l + j] = Tm[3 * l + i][3 * l + j] + (ct ? (T(0.0) - D[l].Jl[3 * j + i]) : Pm[l][3 * i + j]);
    }
  }

  // ---- level-1 rows into the QR factor.  R starts as the diagonal rows:
  //      swing leg: sqrt(w_foot) (z_l - target)  [ID only];  contact leg: eps f_l
  T eps = sqrt(T(P.eps2));
  T Rf[NV][NV + 1];
  for (int i = 0; i < NV; i++) for (int j = 0; j <= NV; j++) Rf[i][j] = T(0.0);
  T blk[6][NV + 1];
  for (int i = 0; i < 6; i++) for (int c = NZ; c < NV; c++) blk[i][c] = T(0.0);
  T clfrow[NZ + 3];
  T clf_c0 = T(0.0), clf_gb[6], clf_gs[4][3];  // CLF logging: Vdot = clf_c0 + clf_gb.(a_b) + sum clf_gs.z_sw
  T met_V = T(0.0), met_err = T(0.0), met_Vdot = T(0.0);
  // task errors (needed by both laws for logging)
  T xt_b[6], xdt_b[6];
  for (int i = 0; i < 3; i++) { xt_b[i] = rpy[i] - tg_rpy[i]; xt_b[3 + i] = p0[i] - tg_pb[i]; }
  for (int i = 0; i < 6; i++) met_err = met_err + xt_b[i] * xt_b[i];

  if (KIND == KIND_ID) {
    T sw_b = sqrt(T(P.w_body)), sw_f = sqrt(T(P.w_foot));
    // desired body acceleration (:187-195)
    T rpydd_des[3], od[3], ades[6];
    for (int i = 0; i < 3; i++) {
      ades[3 + i] = tg_pddb[i] - T(P.Kp_body_p) * (p0[i] - tg_pb[i]) - T(P.Kd_body_p) * (v0[i] - tg_pdb[i]);
      rpydd_des[i] = tg_rpydd[i] - T(P.Kp_body_rpy) * (rpy[i] - tg_rpy[i]) - T(P.Kd_body_rpy) * (rpyd[i] - tg_rpyd[i]);
    }
    rotv(E, rpydd_des, od);
    for (int i = 0; i < 3; i++) ades[i] = od[i];
    for (int l = 0; l < 4; l++) {
      bool ct = (mask >> l) & 1;
      for (int i = 0; i < 3; i++) {
        if (ct) { Rf[3 * l + i][3 * l + i] = eps; continue; }
        T pf = p0[i] + K[l].rf(i);
        T tp = in(37 + 18 + 9 * l + i), tpd = in(37 + 21 + 9 * l + i), tpdd = in(37 + 24 + 9 * l + i);
        T des = tpdd - T(P.Kp_foot) * (pf - tp) - T(P.Kd_foot) * (D[l].pd[i] - tpd);
        Rf[3 * l + i][3 * l + i] = sw_f;
        Rf[3 * l + i][NV] = sw_f * (des - D[l].Jdv[i]);
        met_err = met_err + (pf - tp) * (pf - tp);
      }
    }
    for (int i = 0; i < 6; i++) {
      for (int c = 0; c < NZ; c++) blk[i][c] = sw_b * B[i][c];
      blk[i][NV] = sw_b * (ades[i] - ab0[i]);
    }
    qr_append<T, NV>(Rf, blk, 6);
  } else if (KIND == KIND_CLF) {
    // ---------------- CLF-QP (clf_controller.py:48-234) in task coordinates, weights 1 on every task row
    const T Qp_b = T(5000.0), Qd_b = T(200.0), Qp_f = T(200.0), Qd_f = T(20.0), rr = T(1.0), w_delta = T(1000.0);  // :65-73
    T pb11, pb12, pb22, pf11, pf12, pf22;  // closed-form CARE per task dimension (:187)
    pb12 = sqrt(Qp_b * rr); pb22 = sqrt(rr * (Qd_b + T(2.0) * pb12)); pb11 = pb12 * pb22 / rr;
    pf12 = sqrt(Qp_f * rr); pf22 = sqrt(rr * (Qd_f + T(2.0) * pf12)); pf11 = pf12 * pf22 / rr;
    T om_rt[3], xdn[3], xddn[3], xdd_b[6];
    rotv(E, rpyd, om_rt);
    rotv(E, tg_rpyd, xdn);
    rotv(E, tg_rpydd, xddn);
    for (int i = 0; i < 3; i++) {
      xdt_b[i] = om_rt[i] - xdn[i];
      xdt_b[3 + i] = v0[i] - tg_pdb[i];
      xdd_b[i] = xddn[i];
      xdd_b[3 + i] = tg_pddb[i];
    }
    bool any_swing = false;
    T V = T(0.0), ePFe = T(0.0), ub = T(0.0), gb_ab0 = T(0.0), row2 = T(1.0);  // row2 = |coefficients|^2 (delta: 1)
    for (int c = 0; c < NV; c++) clfrow[c] = T(0.0);
    for (int i = 0; i < 6; i++) {
      T pg = pb12 * xt_b[i] + pb22 * xdt_b[i];
      T gt = T(2.0) * pg;
      clf_gb[i] = gt;
      V = V + pb11 * xt_b[i] * xt_b[i] + T(2.0) * pb12 * xt_b[i] * xdt_b[i] + pb22 * xdt_b[i] * xdt_b[i];
      ePFe = ePFe + pb11 * xt_b[i] * xdt_b[i] + pb12 * xdt_b[i] * xdt_b[i];
      ub = ub + gt * xdd_b[i];                         // -gt (Jdv - xdd_nom), Jdv_body = 0
      T ystar = xdd_b[i] - pg / rr - gt;               // xdd_des - Jdv - gt
      for (int c = 0; c < NZ; c++) { blk[i][c] = B[i][c]; clfrow[c] = clfrow[c] + gt * B[i][c]; }
      blk[i][NV] = ystar - ab0[i];
      gb_ab0 = gb_ab0 + gt * ab0[i];
    }
    for (int l = 0; l < 4; l++) {
      bool ct = (mask >> l) & 1;
      for (int i = 0; i < 3; i++) {
        clf_gs[l][i] = T(0.0);
        if (ct) { Rf[3 * l + i][3 * l + i] = eps; continue; }
        any_swing = true;
        T pf = p0[i] + K[l].rf(i);
        T xt = pf - in(37 + 18 + 9 * l + i), xdt = D[l].pd[i] - in(37 + 21 + 9 * l + i), xddn_s = in(37 + 24 + 9 * l + i);
        T pg = pf12 * xt + pf22 * xdt, gt = T(2.0) * pg;
        clf_gs[l][i] = gt;
        V = V + pf11 * xt * xt + T(2.0) * pf12 * xt * xdt + pf22 * xdt * xdt;
        ePFe = ePFe + pf11 * xt * xdt + pf12 * xdt * xdt;
        ub = ub - gt * (D[l].Jdv[i] - xddn_s);
        Rf[3 * l + i][3 * l + i] = T(1.0);
        Rf[3 * l + i][NV] = xddn_s - pg / rr - D[l].Jdv[i] - gt;
        clfrow[3 * l + i] = clfrow[3 * l + i] + gt;
        met_err = met_err + xt * xt;
      }
    }
    Rf[NZ][NZ] = sqrt(T(2.0) * w_delta);                // w_delta delta^2 = 1/2 (sqrt(2 w) delta)^2   (:206)
    // gamma = min eig(Q) / max eig(P) over the task dimensions that exist (:188)
    T hb = T(0.5) * (pb11 + pb22), db = T(0.5) * (pb11 - pb22), evb = hb + sqrt(db * db + pb12 * pb12);
    T hf = T(0.5) * (pf11 + pf22), df = T(0.5) * (pf11 - pf22), evf = hf + sqrt(df * df + pf12 * pf12);
    T qmin = any_swing ? Qd_f : Qd_b, pmax = (any_swing && evf > evb) ? evf : evb;
    T gamma = qmin / pmax;
    ub = ub - gamma * V - T(2.0) * ePFe - gb_ab0;       // gt.(B z + ab0) + gt_s.z_sw - delta <= ub'
    clfrow[NZ] = T(-1.0);
    for (int c = 0; c < NZ; c++) row2 = row2 + clfrow[c] * clfrow[c];
    T inv = T(1.0) / sqrt(row2);
    for (int c = 0; c < NV; c++) clfrow[c] = clfrow[c] * inv;
    clfrow[NV] = ub * inv;
    met_V = V;
    clf_c0 = T(2.0) * ePFe;
    for (int i = 0; i < 6; i++) clf_c0 = clf_c0 - clf_gb[i] * xdd_b[i];
    for (int l = 0; l < 4; l++)
      if (!((mask >> l) & 1))
        for (int i = 0; i < 3; i++) clf_c0 = clf_c0 + clf_gs[l][i] * (D[l].Jdv[i] - in(37 + 24 + 9 * l + i));
    qr_append<T, NV>(Rf, blk, 6);
  } else {
    // ---------------- MPTC (mptc_controller.py:227-292), in task coordinates
    // Task inertia (arrowhead):  Mt_bb = Gs - sum_l (Ji Jfb)' Y_l ; Mt_bl = (Ji' Y_l)' ; Mt_ll = Ji' P_l
    T Mt_bb[6][6], Mt_bl[4][18], Mt_ll[4][9], Mli[4][9];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) Mt_bb[i][j] = Gs[i][j];
    for (int l = 0; l < 4; l++) {
      const T* r = &K[l].rf(0);
      // A_l = Ji Jfb (3x6) = [ -Ji [r]x , Ji ]
      T A[18];
      for (int i = 0; i < 3; i++) {
        T a0 = D[l].Ji[3 * i], a1 = D[l].Ji[3 * i + 1], a2 = D[l].Ji[3 * i + 2];
        A[6 * i + 0] = T(0.0) - (a1 * r[2] - a2 * r[1]);
        A[6 * i + 1] = T(0.0) - (a2 * r[0] - a0 * r[2]);
        A[6 * i + 2] = T(0.0) - (a0 * r[1] - a1 * r[0]);
        A[6 * i + 3] = a0; A[6 * i + 4] = a1; A[6 * i + 5] = a2;
      }
      for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++)
          Mt_bb[i][j] = Mt_bb[i][j] - (A[i] * Y[l][j] + A[6 + i] * Y[l][6 + j] + A[12 + i] * Y[l][12 + j]);
      // Mt_lb = Ji' Y_l (3x6) ; store Mt_bl[l] as 6x3 = (Mt_lb)'
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 6; j++)
          Mt_bl[l][3 * j + i] = D[l].Ji[i] * Y[l][j] + D[l].Ji[3 + i] * Y[l][6 + j] + D[l].Ji[6 + i] * Y[l][12 + j];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
          Mt_ll[l][3 * i + j] = D[l].Ji[i] * Pm[l][j] + D[l].Ji[3 + i] * Pm[l][3 + j] + D[l].Ji[6 + i] * Pm[l][6 + j];
      T Mf[9];
      sym_to_full(D[l].Mll, Mf);
      inv3(Mf, Mli[l]);
    }
    // Lambda_bb = Mt_bb - sum_ct Y_l' Mll^-1 Y_l  (contact legs eliminated)
    T Lbb[6][6];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) Lbb[i][j] = Mt_bb[i][j];
    T MiY[4][18];  // Mll^-1 Y_l (3x6)
    for (int l = 0; l < 4; l++) {
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 6; j++)
          MiY[l][6 * i + j] = Mli[l][3 * i] * Y[l][j] + Mli[l][3 * i + 1] * Y[l][6 + j] + Mli[l][3 * i + 2] * Y[l][12 + j];
      if ((mask >> l) & 1)
        for (int i = 0; i < 6; i++)
          for (int j = 0; j < 6; j++)
            Lbb[i][j] = Lbb[i][j] - (Y[l][i] * MiY[l][j] + Y[l][6 + i] * MiY[l][6 + j] + Y[l][12 + i] * MiY[l][12 + j]);
    }
    // task errors: body (:243-257; xd uses E*rpyd = omega round trip, kept literal)
    T om_rt[3], xdn[3], xddn[3];
    rotv(E, rpyd, om_rt);
    rotv(E, tg_rpyd, xdn);
    rotv(E, tg_rpydd, xddn);
    T xdd_b[6];
    for (int i = 0; i < 3; i++) {
      xdt_b[i] = om_rt[i] - xdn[i];
      xdt_b[3 + i] = v0[i] - tg_pdb[i];
      xdd_b[i] = xddn[i];
      xdd_b[3 + i] = tg_pddb[i];
    }
    T xt_s[4][3], xdt_s[4][3], xdd_s[4][3];
    for (int l = 0; l < 4; l++)
      for (int i = 0; i < 3; i++) {
        bool ct = (mask >> l) & 1;
        T pf = p0[i] + K[l].rf(i);
        xt_s[l][i] = ct ? T(0.0) : pf - in(37 + 18 + 9 * l + i);
        xdt_s[l][i] = ct ? T(0.0) : D[l].pd[i] - in(37 + 21 + 9 * l + i);
        xdd_s[l][i] = ct ? T(0.0) : in(37 + 24 + 9 * l + i);
        met_err = met_err + xt_s[l][i] * xt_s[l][i];
      }
    // xi = Jbar xd_tilde in generalized coordinates: xi_b = xdt_b; foot task velocities:
    //   swing: xdt_s ; contact: y_c = -Jl Mll^-1 Y_l xdt_b   ->  xi_l = Ji (y_l - Jfb xdt_b)
    T xi_l[4][3];
    for (int l = 0; l < 4; l++) {
      const T* r = &K[l].rf(0);
      T jfb[3], t[3];
      cross(xdt_b, r, t);  // omega x r
      for (int i = 0; i < 3; i++) jfb[i] = xdt_b[3 + i] + t[i];
      if ((mask >> l) & 1) {
        for (int i = 0; i < 3; i++) {
          T s = T(0.0);
          for (int j = 0; j < 6; j++) s = s + MiY[l][6 * i + j] * xdt_b[j];
          T s2 = D[l].Ji[3 * i] * jfb[0] + D[l].Ji[3 * i + 1] * jfb[1] + D[l].Ji[3 * i + 2] * jfb[2];
          xi_l[l][i] = T(0.0) - s - s2;
        }
      } else {
        T y[3] = {xdt_s[l][0] - jfb[0], xdt_s[l][1] - jfb[1], xdt_s[l][2] - jfb[2]};
        rotv(D[l].Ji, y, xi_l[l]);
      }
    }
    // C xi = 1/4 [h(v + xi) - h(v - xi)]  (bias is a quadratic form; gravity cancels)
    T Cxi_b[6] = {T(0.0), T(0.0), T(0.0), T(0.0), T(0.0), T(0.0)}, Cxi_l[4][3];
    for (int sgi = 0; sgi < 2; sgi++) {
      T sg = sgi ? T(-1.0) : T(1.0);
      T wv[3] = {w0[0] + sg * xdt_b[0], w0[1] + sg * xdt_b[1], w0[2] + sg * xdt_b[2]};
      T t2[3], t3[3], Iw_w[3];
      cross(wv, bmc, t2);
      cross(wv, t2, t2);
      symv(bI, wv, Iw_w);
      cross(wv, Iw_w, t3);
      for (int i = 0; i < 3; i++) { Cxi_b[i] = Cxi_b[i] + sg * T(0.25) * t3[i]; Cxi_b[3 + i] = Cxi_b[3 + i] + sg * T(0.25) * t2[i]; }
      for (int l = 0; l < 4; l++) {
        T qv[3] = {qd[l][0] + sg * xi_l[l][0], qd[l][1] + sg * xi_l[l][1], qd[l][2] + sg * xi_l[l][2]};
        T hl2[3], Nb[3], Fb[3];
        const T mass3[3] = {T(m.link[l][0].mass), T(m.link[l][1].mass), T(m.link[l][2].mass)};
        leg_rnea<T, false>(mass3, K[l], wv, qv, T(0.0), hl2, Nb, Fb, (LegDyn<T>*)nullptr);
        for (int i = 0; i < 3; i++) {
          Cxi_b[i] = Cxi_b[i] + sg * T(0.25) * Nb[i];
          Cxi_b[3 + i] = Cxi_b[3 + i] + sg * T(0.25) * Fb[i];
          Cxi_l[l][i] = (sgi ? Cxi_l[l][i] : T(0.0)) + sg * T(0.25) * hl2[i];
        }
      }
    }
    // g = Ybar' (C xi):  g_b = w_b - sum_l A_l' w_l ; g_l = Ji' w_l.
    // Lambda (J Minv C xi) = g_t - [sum_ct Y_l' Mll^-1 w_l ; 0]
    T LJMC_b[6], LJMC_s[4][3];
    for (int i = 0; i < 6; i++) LJMC_b[i] = Cxi_b[i];
    for (int l = 0; l < 4; l++) {
      const T* r = &K[l].rf(0);
      T gl[3];  // Ji' w_l
      for (int i = 0; i < 3; i++) gl[i] = D[l].Ji[i] * Cxi_l[l][0] + D[l].Ji[3 + i] * Cxi_l[l][1] + D[l].Ji[6 + i] * Cxi_l[l][2];
      // A_l' w_l = Jfb' gl = [r x gl ; gl]
      T c[3];
      cross(r, gl, c);
      for (int i = 0; i < 3; i++) { LJMC_b[i] = LJMC_b[i] - c[i]; LJMC_b[3 + i] = LJMC_b[3 + i] - gl[i]; }
      if ((mask >> l) & 1) {
        for (int j = 0; j < 6; j++)
          LJMC_b[j] = LJMC_b[j] - (MiY[l][j] * Cxi_l[l][0] + MiY[l][6 + j] * Cxi_l[l][1] + MiY[l][12 + j] * Cxi_l[l][2]);
        for (int i = 0; i < 3; i++) LJMC_s[l][i] = T(0.0);
      } else {
        for (int i = 0; i < 3; i++) LJMC_s[l][i] = gl[i];
      }
    }
    // s1 = xdd_nom - Jd v + Jd xi   (task space; body rows of Jd are zero)
    T s1_b[6], s1_s[4][3];
    for (int i = 0; i < 6; i++) s1_b[i] = xdd_b[i];
    for (int l = 0; l < 4; l++) {
      if ((mask >> l) & 1) { for (int i = 0; i < 3; i++) s1_s[l][i] = T(0.0); continue; }
      // Jd xi = xi_w x rd + Jd_l xi_l
      T t[3];
      cross(xdt_b, D[l].rd, t);
      for (int i = 0; i < 3; i++) {
        T jx = t[i] + D[l].Jd[3 * i] * xi_l[l][0] + D[l].Jd[3 * i + 1] * xi_l[l][1] + D[l].Jd[3 * i + 2] * xi_l[l][2];
        s1_s[l][i] = xdd_s[l][i] - D[l].Jdv[i] + jx;
      }
    }
    // Lambda * (vector in task space), Lambda = [[Lbb, Mt_b,sw],[Mt_sw,b, Mt_ll]]
    auto lam_mul = [&](const T* yb, const T (*ys)[3], T* ob, T (*os)[3]) {
      for (int i = 0; i < 6; i++) {
        T s = T(0.0);
        for (int j = 0; j < 6; j++) s = s + Lbb[i][j] * yb[j];
        ob[i] = s;
      }
      for (int l = 0; l < 4; l++) {
        if ((mask >> l) & 1) { os[l][0] = os[l][1] = os[l][2] = T(0.0); continue; }
        for (int i = 0; i < 6; i++)
          ob[i] = ob[i] + Mt_bl[l][3 * i] * ys[l][0] + Mt_bl[l][3 * i + 1] * ys[l][1] + Mt_bl[l][3 * i + 2] * ys[l][2];
        for (int i = 0; i < 3; i++) {
          T s = Mt_ll[l][3 * i] * ys[l][0] + Mt_ll[l][3 * i + 1] * ys[l][1] + Mt_ll[l][3 * i + 2] * ys[l][2];
          for (int j = 0; j < 6; j++) s = s + Mt_bl[l][3 * j + i] * yb[j];
          os[l][i] = s;
        }
      }
    };
    // c1 = -Lambda s1 + Lambda J Minv C xi + Kp xt + Kd xdt ;  residual r1(z) = Lambda y_t(z) + c1
    T Ls_b[6], Ls_s[4][3];
    lam_mul(s1_b, s1_s, Ls_b, Ls_s);
    T c1_b[6], c1_s[4][3];
    for (int i = 0; i < 6; i++) {
      T kp = (i < 3) ? T(P.Kp_body_rpy) : T(P.Kp_body_p), kd = (i < 3) ? T(P.Kd_body_rpy) : T(P.Kd_body_p);
      c1_b[i] = LJMC_b[i] - Ls_b[i] + kp * xt_b[i] + kd * xdt_b[i];
      met_V = met_V + T(0.5) * kp * xt_b[i] * xt_b[i];
      met_Vdot = met_Vdot - kd * xdt_b[i] * xdt_b[i];
    }
    for (int l = 0; l < 4; l++)
      for (int i = 0; i < 3; i++) {
        c1_s[l][i] = LJMC_s[l][i] - Ls_s[l][i] + T(P.Kp_foot) * xt_s[l][i] + T(P.Kd_foot) * xdt_s[l][i];
        met_V = met_V + T(0.5) * T(P.Kp_foot) * xt_s[l][i] * xt_s[l][i];
        met_Vdot = met_Vdot - T(P.Kd_foot) * xdt_s[l][i] * xdt_s[l][i];
      }
    {
      T Lx_b[6], Lx_s[4][3];
      lam_mul(xdt_b, xdt_s, Lx_b, Lx_s);
      for (int i = 0; i < 6; i++) met_V = met_V + T(0.5) * xdt_b[i] * Lx_b[i];
      for (int l = 0; l < 4; l++) for (int i = 0; i < 3; i++) met_V = met_V + T(0.5) * xdt_s[l][i] * Lx_s[l][i];
    }
    // level-1 rows: sqrt(W) (Lambda [B z + ab0; z_sw] + c1); contact legs get the eps f rows
    T sw_b = sqrt(T(P.w_body)), sw_f = sqrt(T(P.w_foot));
    for (int l = 0; l < 4; l++)
      if ((mask >> l) & 1)
        for (int i = 0; i < 3; i++) Rf[3 * l + i][3 * l + i] = eps;
    // body rows
    for (int i = 0; i < 6; i++) {
      for (int c = 0; c < NZ; c++) {
        T s = T(0.0);
        for (int j = 0; j < 6; j++) s = s + Lbb[i][j] * B[j][c];
        blk[i][c] = sw_b * s;
      }
      T s = c1_b[i];
      for (int j = 0; j < 6; j++) s = s + Lbb[i][j] * ab0[j];
      blk[i][NV] = T(0.0) - sw_b * s;
      for (int l = 0; l < 4; l++)
        if (!((mask >> l) & 1))
          for (int j = 0; j < 3; j++) blk[i][3 * l + j] = blk[i][3 * l + j] + sw_b * Mt_bl[l][3 * i + j];
    }
    qr_append<T, NV>(Rf, blk, 6);
    // swing rows (3 per swing leg), appended in blocks of 3
    for (int l = 0; l < 4; l++) {
      if ((mask >> l) & 1) continue;
      for (int i = 0; i < 3; i++) {
        for (int c = 0; c < NZ; c++) {
          T s = T(0.0);
          for (int j = 0; j < 6; j++) s = s + Mt_bl[l][3 * j + i] * B[j][c];
          blk[i][c] = sw_f * s;
        }
        T s = c1_s[l][i];
        for (int j = 0; j < 6; j++) s = s + Mt_bl[l][3 * j + i] * ab0[j];
        blk[i][NV] = T(0.0) - sw_f * s;
        for (int j = 0; j < 3; j++) blk[i][3 * l + j] = blk[i][3 * l + j] + sw_f * Mt_ll[l][3 * i + j];
      }
      qr_append<T, NV>(Rf, blk, 3);
    }
    // keep c1 and Lambda pieces for Vdot: Vdot = xdt' r1 - xdt' Kd xdt, r1 = Lambda y_t + c1
    // evaluated after the solve; stash what is needed in blk-independent storage
    // (recomputed below from z through lam_mul)
    // -> store c1 dot xdt now:
    for (int i = 0; i < 6; i++) met_Vdot = met_Vdot + xdt_b[i] * c1_b[i];
    for (int l = 0; l < 4; l++) for (int i = 0; i < 3; i++) met_Vdot = met_Vdot + xdt_s[l][i] * c1_s[l][i];
    // and Lambda xdt (symmetric) so that xdt' Lambda y_t = (Lambda xdt)' y_t
    {
      T Lx_b[6], Lx_s[4][3];
      lam_mul(xdt_b, xdt_s, Lx_b, Lx_s);
      // fold into a 12-vector over z plus constant:  y_t = [B z + ab0 ; z_sw]
      // Vdot += Lx_b'(B z + ab0) + sum_sw Lx_s' z_l  -> coefficients kept in blk[0]
      for (int c = 0; c < NZ; c++) {
        T s = T(0.0);
        for (int j = 0; j < 6; j++) s = s + Lx_b[j] * B[j][c];
        blk[0][c] = s;
      }
      T s = T(0.0);
      for (int j = 0; j < 6; j++) s = s + Lx_b[j] * ab0[j];
      blk[0][NV] = s;
      for (int l = 0; l < 4; l++)
        if (!((mask >> l) & 1))
          for (int i = 0; i < 3; i++) blk[0][3 * l + i] = blk[0][3 * l + i] + Lx_s[l][i];
    }
  }
  T vdot_row[NZ + 1];
  for (int c = 0; c < NZ; c++) vdot_row[c] = (KIND == KIND_MPTC || KIND == KIND_PC) ? blk[0][c] : T(0.0);
  vdot_row[NZ] = (KIND == KIND_MPTC || KIND == KIND_PC) ? blk[0][NV] : T(0.0);

  // ---- level-2 rows: eps (Tm z + t0)
  for (int h = 0; h < 2; h++) {
    for (int i = 0; i < 6; i++) {
      for (int c = 0; c < NZ; c++) blk[i][c] = eps * Tm[6 * h + i][c];
      for (int c = NZ; c < NV; c++) blk[i][c] = T(0.0);
      blk[i][NV] = T(0.0) - eps * Tm[6 * h + i][NV];
    }
    qr_append<T, NV>(Rf, blk, 6);
  }
  // ---- unconstrained minimiser and J = R^-1
  T z[NV], Jm[NV][NV];
  {
    T rmax = T(0.0), rmin = T(0.0);
    for (int i = 0; i < NV; i++) {
      T a = wabs(Rf[i][i]);
      if (i == 0 || a > rmax) rmax = a;
      if (i == 0 || a < rmin) rmin = a;
    }
    if (!(rmin > T(1e-13) * rmax)) status = ST_SINGULAR;
  }
  if (status == ST_SINGULAR) {
    for (int k = 0; k < 12; k++) out_tau(k, T(0.0));
    out_met(0, T(0.0)); out_met(1, met_err); out_met(2, T(0.0)); out_met(3, T(0.0));
    for (int k = 0; k < 18; k++) out_met(4 + k, T(0.0));   // defined accelerations with the zero torques
    *iters_out = 0;
    return status;
  }
  for (int k = NV - 1; k >= 0; k--) {
    T s = Rf[k][NV];
    for (int j = k + 1; j < NV; j++) s = s - Rf[k][j] * z[j];
    z[k] = s / Rf[k][k];
  }
  for (int c = 0; c < NV; c++)
    for (int k = NV - 1; k >= 0; k--) {
      if (k > c) { Jm[k][c] = T(0.0); continue; }
      T s = (k == c) ? T(1.0) : T(0.0);
      for (int j = k + 1; j <= c; j++) s = s - Rf[k][j] * Jm[j][c];
      Jm[k][c] = s / Rf[k][k];
    }
  // ---- inequalities
  QpCons<T> C;
  {
    T s = sqrt(T(1.0) + mu * mu);
    C.inv_s = T(1.0) / s;
    C.mu_n = mu * C.inv_s;
    C.Trow = &Tm[0][0];
    C.tstride = NV + 1;
    C.clfrow = nullptr;
    C.tau_max = T(P.tau_max);
    C.mask = mask;
    C.pcrow = nullptr;
    C.pc_inv = T(0.0);
  }
  unsigned long long elig = 0ull;
  for (int l = 0; l < 4; l++)
    if ((mask >> l) & 1) elig |= (0xFull << (4 * l));
  if (P.tau_max < 1e300) {
    for (int j = 0; j < 12; j++) {
      T s = T(0.0);
      for (int k = 0; k < NZ; k++) s = s + Tm[j][k] * Tm[j][k];
      C.tnorm[j] = sqrt(s);
      if (s > T(0.0)) elig |= (3ull << (16 + 2 * j));
    }
  }
  T pcrow[NZ + 1];
  if (KIND == KIND_PC) {
    // pc_controller.py:14-40,229-237: Vdot <= delta <= 0 with a cost-free delta  <=>  Vdot <= 0
    T s = T(0.0);
    for (int c = 0; c < NZ; c++) { pcrow[c] = vdot_row[c]; s = s + vdot_row[c] * vdot_row[c]; }
    pcrow[NZ] = vdot_row[NZ] + met_Vdot;
    if (s > T(0.0)) {
      C.pcrow = pcrow;
      C.pc_inv = T(1.0) / sqrt(s);
      elig |= (1ull << PC_ROW);
    }
  }
  if (KIND == KIND_CLF) {
    C.clfrow = clfrow;
    elig |= (1ull << CLF_ROW);
  }
  int iters = 0;
  int st = gi_solve<T, NV>(Jm, z, C, elig, &iters);
  *iters_out = iters;
  if (st != ST_OK) status = st;
  if (status == ST_OK && illc) status = ST_ILLCOND;
  // ---- outputs
  T tauc[12];
  for (int i = 0; i < 12; i++) {
    T s = Tm[i][NV];
    for (int c = 0; c < NZ; c++) s = s + Tm[i][c] * z[c];
    tauc[i] = s;
  }
  for (int k = 0; k < 12; k++) out_tau(k, (status == ST_SINGULAR) ? T(0.0) : tauc[m.act_perm[k]]);
  {
    // generalized accelerations of the QP solution, rows 4..21 of out_met (optional consumer: the
    // forward step of the closed-loop rollout): vd_b = a_b, vd_l = Ji (a_foot_l - Jfb_l a_b)
    T ab[6];
    for (int i = 0; i < 6; i++) {
      T sab = ab0[i];
      for (int c = 0; c < NZ; c++) sab = sab + B[i][c] * z[c];
      ab[i] = sab;
      out_met(4 + i, (status == ST_SINGULAR) ? T(0.0) : sab);
    }
    for (int l = 0; l < 4; l++) {
      bool ctl = (mask >> l) & 1;
      T rr[3] = {K[l].rf(0), K[l].rf(1), K[l].rf(2)}, t[3], y[3], vd[3];
      cross(ab, rr, t);
      for (int i = 0; i < 3; i++) y[i] = (ctl ? bc[l][i] : z[3 * l + i]) - (ab[3 + i] + t[i]);
      rotv(D[l].Ji, y, vd);
      for (int k = 0; k < 3; k++) out_met(4 + 6 + m.q_perm[3 * l + k], (status == ST_SINGULAR) ? T(0.0) : vd[k]);
    }
  }
  // primal residual: worst friction / torque-box violation
  T res = T(0.0);
  for (int l = 0; l < 4; l++)
    if ((mask >> l) & 1) {
      T fx = z[3 * l], fy = z[3 * l + 1], fz = z[3 * l + 2];
      T a = wabs(fx) - mu * fz, b = wabs(fy) - mu * fz;
      if (a > res) res = a;
      if (b > res) res = b;
    }
  if (KIND == KIND_CLF) {
    // Vdot = 2 eta'PF eta + 2 eta'PG (J vd + Jdv - xdd_nom)   (clf_controller.py:230)
    T s = clf_c0;
    for (int i = 0; i < 6; i++) {
      T ab = ab0[i];
      for (int c = 0; c < NZ; c++) ab = ab + B[i][c] * z[c];
      s = s + clf_gb[i] * ab;
    }
    for (int l = 0; l < 4; l++)
      for (int i = 0; i < 3; i++) s = s + clf_gs[l][i] * z[3 * l + i];
    out_met(0, met_V); out_met(1, met_err); out_met(2, T(0.0)); out_met(3, s);
  } else if (KIND != KIND_ID) {
    T s = vdot_row[NZ];
    for (int c = 0; c < NZ; c++) s = s + vdot_row[c] * z[c];
    met_Vdot = met_Vdot + s;
    out_met(0, met_V); out_met(1, met_err); out_met(2, T(0.0)); out_met(3, met_Vdot);
  } else {
    out_met(0, T(0.0)); out_met(1, met_err); out_met(2, res); out_met(3, T(0.0));
  }
  return status;
}

}  // namespace wbc
