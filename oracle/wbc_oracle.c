/*
 * wbc_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See wbc_oracle.h.
 *
 * Every function cites the reference file:line it restates.  The rigid-body terms restate
 * what the Drake calls at those lines are documented to return (Drake itself is absent:
 * "parity unpinned" at that boundary, see header).  Deliberately dense and literal: M is built from 18
 * inverse-dynamics passes like CalcMassMatrixViaInverseDynamics, the QP is assembled in
 * the reference's 30+3nc variables, matrices are inverted where the reference inverts.
 */
#include "wbc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ small dense helpers */
/* Which status the task-space laws report on (nearly) straight knees: 1 (default) = the product's reporting convention of
 * include/wbc.h mirrored (status 2 on an exactly singular foot Jacobian, status 3 on |sin(knee)| < 1e-4), so that checker and
 * checked can be compared tick by tick; 0 = what the dense restatement itself returns there (its own solver's status, nothing
 * mirrored) -- the independent view, tests/test_oracle_qp.py::test_raw_status_is_the_dense_solver_s_own. */
static int g_product_status = 1;
void orc_set_status_convention(int product) { g_product_status = product ? 1 : 0; }

static void cross3(const double* a, const double* b, double* c) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  c[0] = x; c[1] = y; c[2] = z;
}
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void mat3_vec(const double* R, const double* x, double* y) {
  double a = R[0] * x[0] + R[1] * x[1] + R[2] * x[2];
  double b = R[3] * x[0] + R[4] * x[1] + R[5] * x[2];
  double c = R[6] * x[0] + R[7] * x[1] + R[8] * x[2];
  y[0] = a; y[1] = b; y[2] = c;
}
static void mat3_mul(const double* A, const double* B, double* C) {
  double T[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(C, T, sizeof T);
}
/* C(m x n) = A(m x k) B(k x n) */
static void mm(int m, int k, int n, const double* A, const double* B, double* C) {
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) {
      double s = 0;
      for (int l = 0; l < k; l++) s += A[i * k + l] * B[l * n + j];
      C[i * n + j] = s;
    }
}
/* C(m x n) = A(m x k) B(n x k)^T */
static void mmt(int m, int k, int n, const double* A, const double* B, double* C) {
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) {
      double s = 0;
      for (int l = 0; l < k; l++) s += A[i * k + l] * B[j * k + l];
      C[i * n + j] = s;
    }
}
/* C(m x n) = A(k x m)^T B(k x n) */
static void mtm(int m, int k, int n, const double* A, const double* B, double* C) {
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) {
      double s = 0;
      for (int l = 0; l < k; l++) s += A[l * m + i] * B[l * n + j];
      C[i * n + j] = s;
    }
}
static void mv(int m, int n, const double* A, const double* x, double* y) {
  for (int i = 0; i < m; i++) {
    double s = 0;
    for (int j = 0; j < n; j++) s += A[i * n + j] * x[j];
    y[i] = s;
  }
}
static void mtv(int m, int n, const double* A, const double* x, double* y) { /* y(n) = A^T x(m) */
  for (int j = 0; j < n; j++) {
    double s = 0;
    for (int i = 0; i < m; i++) s += A[i * n + j] * x[i];
    y[j] = s;
  }
}
/* In-place inverse by Gauss-Jordan with partial pivoting (np.linalg.inv restatement:
 * mptc_controller.py:237-238).  Returns 0 on success. */
static int mat_inv(int n, double* A) {
  double W[18 * 36]; /* n <= 18 on this path; no heap traffic in the hot loop */
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      W[i * 2 * n + j] = A[i * n + j];
      W[i * 2 * n + n + j] = (i == j);
    }
  for (int c = 0; c < n; c++) {
    int piv = c;
    for (int r = c + 1; r < n; r++)
      if (fabs(W[r * 2 * n + c]) > fabs(W[piv * 2 * n + c])) piv = r;
    if (fabs(W[piv * 2 * n + c]) < 1e-300) return 1;
    if (piv != c)
      for (int j = 0; j < 2 * n; j++) { double t = W[c * 2 * n + j]; W[c * 2 * n + j] = W[piv * 2 * n + j]; W[piv * 2 * n + j] = t; }
    double d = 1.0 / W[c * 2 * n + c];
    for (int j = 0; j < 2 * n; j++) W[c * 2 * n + j] *= d;
    for (int r = 0; r < n; r++)
      if (r != c) {
        double f = W[r * 2 * n + c];
        if (f != 0)
          for (int j = 0; j < 2 * n; j++) W[r * 2 * n + j] -= f * W[c * 2 * n + j];
      }
  }
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) A[i * n + j] = W[i * 2 * n + n + j];
  return 0;
}

/* ------------------------------------------------------------------ model */
void orc_model_from_flat(const double* f, orc_model* m) {
  int k = 0;
  m->base_mass = f[k++];
  for (int i = 0; i < 3; i++) m->base_com[i] = f[k++];
  for (int i = 0; i < 6; i++) m->base_I[i] = f[k++];
  for (int l = 0; l < 4; l++)
    for (int j = 0; j < 3; j++) {
      orc_link* L = &m->link[l][j];
      for (int i = 0; i < 3; i++) L->off[i] = f[k++];
      for (int i = 0; i < 3; i++) L->axis[i] = f[k++];
      L->mass = f[k++];
      for (int i = 0; i < 3; i++) L->com[i] = f[k++];
      for (int i = 0; i < 6; i++) L->I[i] = f[k++];
    }
  for (int l = 0; l < 4; l++)
    for (int i = 0; i < 3; i++) m->foot_off[l][i] = f[k++];
  m->gravity = f[k++];
  for (int i = 0; i < 12; i++) m->act_perm[i] = i;
}

void orc_params_id_default(orc_params* p) { /* inverse_dynamics_controller.py:117-127 */
  p->Kp_body_p = 500.0; p->Kd_body_p = 50.0;
  p->Kp_body_rpy = 500.0; p->Kd_body_rpy = 50.0;
  p->Kp_foot = 100.0; p->Kd_foot = 20.0;
  p->w_body = 10.0; p->w_foot = 1.0;
  p->mu = 0.7; p->Kd_contact = 100.0;
  p->tau_max = INFINITY;
  p->tiebreak_eps2 = 1e-8;
}
void orc_params_mptc_default(orc_params* p) { /* mptc_controller.py:143-153 */
  p->Kp_body_p = 100.0; p->Kd_body_p = 10.0;
  p->Kp_body_rpy = 100.0; p->Kd_body_rpy = 10.0;
  p->Kp_foot = 200.0; p->Kd_foot = 20.0;
  p->w_body = 10.0; p->w_foot = 1.0;
  p->mu = 0.7; p->Kd_contact = 100.0;
  p->tau_max = INFINITY;
  p->tiebreak_eps2 = 1e-8;
}

/* ------------------------------------------------------------------ kinematics
 * Body 0 = floating base, body 1+3l+k = link k of leg l.  Everything is expressed in the
 * world frame W, velocities/accelerations are those of each body-frame ORIGIN (Drake's
 * spatial-velocity convention V_WB = [w_WB; v_WBo]). */
#define NB 13
typedef struct {
  double R[NB][9], p[NB][3], ax[NB][3];
  double w[NB][3], vo[NB][3];   /* angular velocity, origin velocity */
  double al[NB][3], ao[NB][3];  /* angular acceleration, origin acceleration */
} orc_kin;

static int parent_of(int b) { return (b == 0) ? -1 : (((b - 1) % 3 == 0) ? 0 : b - 1); }
static const orc_link* link_of(const orc_model* m, int b) { return &m->link[(b - 1) / 3][(b - 1) % 3]; }

static void quat_to_R(const double* q, double* R) {
  /* Drake RotationMatrix(Eigen::Quaternion): scales by 2/|q|^2, so a non-unit q is tolerated */
  double w = q[0], x = q[1], y = q[2], z = q[3];
  double s = 2.0 / (w * w + x * x + y * y + z * z);
  R[0] = 1 - s * (y * y + z * z); R[1] = s * (x * y - w * z);     R[2] = s * (x * z + w * y);
  R[3] = s * (x * y + w * z);     R[4] = 1 - s * (x * x + z * z); R[5] = s * (y * z - w * x);
  R[6] = s * (x * z - w * y);     R[7] = s * (y * z + w * x);     R[8] = 1 - s * (x * x + y * y);
}
static void axis_angle_R(const double* a, double th, double* R) {
  double c = cos(th), s = sin(th), t = 1 - c;
  R[0] = c + a[0] * a[0] * t;        R[1] = a[0] * a[1] * t - a[2] * s; R[2] = a[0] * a[2] * t + a[1] * s;
  R[3] = a[1] * a[0] * t + a[2] * s; R[4] = c + a[1] * a[1] * t;        R[5] = a[1] * a[2] * t - a[0] * s;
  R[6] = a[2] * a[0] * t - a[1] * s; R[7] = a[2] * a[1] * t + a[0] * s; R[8] = c + a[2] * a[2] * t;
}

/* positions, velocities and accelerations for given (q, v, vd) */
static void orc_forward(const orc_model* m, const double* q, const double* v, const double* vd, orc_kin* K) {
  quat_to_R(q, K->R[0]);
  for (int i = 0; i < 3; i++) {
    K->p[0][i] = q[4 + i];
    K->w[0][i] = v[i]; K->vo[0][i] = v[3 + i];
    K->al[0][i] = vd[i]; K->ao[0][i] = vd[3 + i];
    K->ax[0][i] = 0;
  }
  for (int b = 1; b < NB; b++) {
    int P = parent_of(b);
    const orc_link* L = link_of(m, b);
    double th = q[7 + b - 1], thd = v[6 + b - 1], thdd = vd[6 + b - 1];
    double r[3], Rj[9], t[3], t2[3];
    mat3_vec(K->R[P], L->off, r);
    for (int i = 0; i < 3; i++) K->p[b][i] = K->p[P][i] + r[i];
    axis_angle_R(L->axis, th, Rj);
    mat3_mul(K->R[P], Rj, K->R[b]);
    mat3_vec(K->R[P], L->axis, K->ax[b]);
    /* velocity */
    cross3(K->w[P], r, t);
    for (int i = 0; i < 3; i++) {
      K->w[b][i] = K->w[P][i] + K->ax[b][i] * thd;
      K->vo[b][i] = K->vo[P][i] + t[i];
    }
    /* acceleration: al = al_P + (w_P x a) thd + a thdd ; ao = ao_P + al_P x r + w_P x (w_P x r) */
    double wxa[3];
    cross3(K->w[P], K->ax[b], wxa);
    cross3(K->al[P], r, t);
    cross3(K->w[P], r, t2);
    cross3(K->w[P], t2, t2);
    for (int i = 0; i < 3; i++) {
      K->al[b][i] = K->al[P][i] + wxa[i] * thd + K->ax[b][i] * thdd;
      K->ao[b][i] = K->ao[P][i] + t[i] + t2[i];
    }
  }
}

static void body_inertia(const orc_model* m, int b, double* mass, const double** com, const double** I6) {
  if (b == 0) { *mass = m->base_mass; *com = m->base_com; *I6 = m->base_I; }
  else { const orc_link* L = link_of(m, b); *mass = L->mass; *com = L->com; *I6 = L->I; }
}

/* Newton-Euler inverse dynamics in world coordinates about body origins. */
void orc_inverse_dynamics(const orc_model* m, const double* q, const double* v, const double* vd,
                          int with_gravity, double* tau) {
  orc_kin K;
  orc_forward(m, q, v, vd, &K);
  double F[NB][3], N[NB][3];
  double gvec[3] = {0, 0, with_gravity ? -m->gravity : 0.0};
  for (int b = 0; b < NB; b++) {
    double mass; const double *com, *I6;
    body_inertia(m, b, &mass, &com, &I6);
    double c[3], Ib[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]}, Iw[9], Rt[9];
    mat3_vec(K.R[b], com, c);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rt[3 * i + j] = K.R[b][3 * j + i];
    mat3_mul(K.R[b], Ib, Iw);
    mat3_mul(Iw, Rt, Iw);
    double alxc[3], wxc[3], wxwxc[3], Ial[3], Iw_w[3], wxIw[3], cxa[3], cxg[3];
    cross3(K.al[b], c, alxc);
    cross3(K.w[b], c, wxc);
    cross3(K.w[b], wxc, wxwxc);
    mat3_vec(Iw, K.al[b], Ial);
    mat3_vec(Iw, K.w[b], Iw_w);
    cross3(K.w[b], Iw_w, wxIw);
    double a_minus_g[3] = {K.ao[b][0] - gvec[0], K.ao[b][1] - gvec[1], K.ao[b][2] - gvec[2]};
    cross3(c, a_minus_g, cxa);
    (void)cxg;
    for (int i = 0; i < 3; i++) {
      F[b][i] = mass * (a_minus_g[i] + alxc[i] + wxwxc[i]);
      N[b][i] = Ial[i] + wxIw[i] + mass * cxa[i];
    }
  }
  for (int b = NB - 1; b >= 1; b--) {
    int P = parent_of(b);
    double r[3], rxF[3];
    for (int i = 0; i < 3; i++) r[i] = K.p[b][i] - K.p[P][i];
    cross3(r, F[b], rxF);
    tau[6 + b - 1] = dot3(K.ax[b], N[b]);
    for (int i = 0; i < 3; i++) { F[P][i] += F[b][i]; N[P][i] += N[b][i] + rxF[i]; }
  }
  for (int i = 0; i < 3; i++) { tau[i] = N[0][i]; tau[3 + i] = F[0][i]; }
}

/* basic_controller.py:101-115 */
void orc_calc_dynamics(const orc_model* m, const double* q, const double* v, double* M, double* Cv,
                       double* tau_g) {
  double zero[ORC_NV] = {0}, e[ORC_NV], col[ORC_NV];
  /* :110 CalcMassMatrixViaInverseDynamics: column j = ID(q, v=0, vd=e_j) without gravity */
  for (int j = 0; j < ORC_NV; j++) {
    memset(e, 0, sizeof e);
    e[j] = 1.0;
    orc_inverse_dynamics(m, q, zero, e, 0, col);
    for (int i = 0; i < ORC_NV; i++) M[i * ORC_NV + j] = col[i];
  }
  /* :111 CalcBiasTerm: C(q,v)v, no gravity */
  orc_inverse_dynamics(m, q, v, zero, 0, Cv);
  /* :112 tau_g = -CalcGravityGeneralizedForces: the gravity term on the LEFT-hand side */
  orc_inverse_dynamics(m, q, zero, zero, 1, tau_g);
}

/* basic_controller.py:117-132: C = 1/2 d(Cv)/dv.  Cv is a homogeneous quadratic form in v, so
 * d(Cv)/dv e_j = b(v+e_j) - b(v) - b(e_j) exactly (what the AutoDiffXd pass computes). */
void orc_coriolis_matrix(const orc_model* m, const double* q, const double* v, double* C) {
  double zero[ORC_NV] = {0}, b0[ORC_NV], bj[ORC_NV], bvj[ORC_NV], e[ORC_NV], ve[ORC_NV];
  orc_inverse_dynamics(m, q, v, zero, 0, b0);
  for (int j = 0; j < ORC_NV; j++) {
    memset(e, 0, sizeof e);
    e[j] = 1.0;
    memcpy(ve, v, sizeof ve);
    ve[j] += 1.0;
    orc_inverse_dynamics(m, q, e, zero, 0, bj);
    orc_inverse_dynamics(m, q, ve, zero, 0, bvj);
    for (int i = 0; i < ORC_NV; i++) C[i * ORC_NV + j] = 0.5 * (bvj[i] - b0[i] - bj[i]);
  }
}

/* basic_controller.py:173-196: CalcPointsPositions, CalcJacobianTranslationalVelocity (wrt v,
 * world/world), CalcBiasTranslationalAcceleration for the foot frame origin. */
void orc_foot_quantities(const orc_model* m, const double* q, const double* v, int foot, double* p,
                         double* J, double* Jdv) {
  orc_kin K;
  double zero[ORC_NV] = {0};
  orc_forward(m, q, v, zero, &K);
  int sh = 1 + 3 * foot + 2;
  double d[3];
  mat3_vec(K.R[sh], m->foot_off[foot], d);
  for (int i = 0; i < 3; i++) p[i] = K.p[sh][i] + d[i];
  memset(J, 0, sizeof(double) * 3 * ORC_NV);
  double r[3] = {p[0] - K.p[0][0], p[1] - K.p[0][1], p[2] - K.p[0][2]};
  /* v_f = v_0 + w_0 x r + sum_k (a_k x (p_f - p_k)) thd_k */
  J[0 * 18 + 1] = r[2];  J[0 * 18 + 2] = -r[1];
  J[1 * 18 + 0] = -r[2]; J[1 * 18 + 2] = r[0];
  J[2 * 18 + 0] = r[1];  J[2 * 18 + 1] = -r[0];
  J[0 * 18 + 3] = 1; J[1 * 18 + 4] = 1; J[2 * 18 + 5] = 1;
  for (int k = 0; k < 3; k++) {
    int b = 1 + 3 * foot + k;
    double rk[3] = {p[0] - K.p[b][0], p[1] - K.p[b][1], p[2] - K.p[b][2]}, col[3];
    cross3(K.ax[b], rk, col);
    for (int i = 0; i < 3; i++) J[i * 18 + 6 + 3 * foot + k] = col[i];
  }
  /* bias acceleration = acceleration of the point with vd = 0 */
  double t[3], t2[3];
  cross3(K.al[sh], d, t);
  cross3(K.w[sh], d, t2);
  cross3(K.w[sh], t2, t2);
  for (int i = 0; i < 3; i++) Jdv[i] = K.ao[sh][i] + t[i] + t2[i];
}

/* basic_controller.py:198-220 + helpers.py:5-33: Jd = (dJ/dq) N(q) v = dJ/dt along the flow. */
void orc_foot_jacobian_dot(const orc_model* m, const double* q, const double* v, int foot, double* Jd) {
  orc_kin K;
  double zero[ORC_NV] = {0};
  orc_forward(m, q, v, zero, &K);
  int sh = 1 + 3 * foot + 2;
  double d[3], pf[3], vf[3], t[3];
  mat3_vec(K.R[sh], m->foot_off[foot], d);
  cross3(K.w[sh], d, t);
  for (int i = 0; i < 3; i++) { pf[i] = K.p[sh][i] + d[i]; vf[i] = K.vo[sh][i] + t[i]; }
  memset(Jd, 0, sizeof(double) * 3 * ORC_NV);
  double rd[3] = {vf[0] - K.vo[0][0], vf[1] - K.vo[0][1], vf[2] - K.vo[0][2]};
  Jd[0 * 18 + 1] = rd[2];  Jd[0 * 18 + 2] = -rd[1];
  Jd[1 * 18 + 0] = -rd[2]; Jd[1 * 18 + 2] = rd[0];
  Jd[2 * 18 + 0] = rd[1];  Jd[2 * 18 + 1] = -rd[0];
  for (int k = 0; k < 3; k++) {
    int b = 1 + 3 * foot + k, P = parent_of(b);
    double rk[3], rkd[3], ad[3], c1[3], c2[3];
    for (int i = 0; i < 3; i++) { rk[i] = pf[i] - K.p[b][i]; rkd[i] = vf[i] - K.vo[b][i]; }
    cross3(K.w[P], K.ax[b], ad); /* axis is fixed in the parent: d/dt a = w_P x a */
    cross3(ad, rk, c1);
    cross3(K.ax[b], rkd, c2);
    for (int i = 0; i < 3; i++) Jd[i * 18 + 6 + 3 * foot + k] = c1[i] + c2[i];
  }
}

/* basic_controller.py:246-269 for the floating body frame (the task frame of both laws). */
void orc_body_quantities(const orc_model* m, const double* q, const double* v, double* R, double* p,
                         double* J, double* Jdv) {
  (void)m; (void)v;
  quat_to_R(q, R);
  for (int i = 0; i < 3; i++) p[i] = q[4 + i];
  memset(J, 0, sizeof(double) * 6 * ORC_NV);
  for (int i = 0; i < 6; i++) J[i * 18 + i] = 1.0; /* V_WB = [w; v] = v[0:6] */
  for (int i = 0; i < 6; i++) Jdv[i] = 0.0;
}

void orc_rpy_from_R(const double* R, double* rpy) {
  /* R = Rz(y) Ry(p) Rx(r) */
  rpy[0] = atan2(R[7], R[8]);
  rpy[1] = atan2(-R[6], sqrt(R[0] * R[0] + R[3] * R[3]));
  rpy[2] = atan2(R[3], R[0]);
}
void orc_rpy_E(const double* rpy, double* E) {
  double sp = sin(rpy[1]), cp = cos(rpy[1]), sy = sin(rpy[2]), cy = cos(rpy[2]);
  E[0] = cp * cy; E[1] = -sy; E[2] = 0;
  E[3] = cp * sy; E[4] = cy;  E[5] = 0;
  E[6] = -sp;     E[7] = 0;   E[8] = 1;
}
static void rpy_Einv(const double* rpy, double* Ei) {
  /* CalcRpyDtFromAngularVelocityInParent */
  double sp = sin(rpy[1]), cp = cos(rpy[1]), sy = sin(rpy[2]), cy = cos(rpy[2]);
  Ei[0] = cy / cp;      Ei[1] = sy / cp;      Ei[2] = 0;
  Ei[3] = -sy;          Ei[4] = cy;           Ei[5] = 0;
  Ei[6] = cy * sp / cp; Ei[7] = sy * sp / cp; Ei[8] = 1;
}

/* ------------------------------------------------------------------ dense QP */
/* Householder QR of A (m x n, m >= n), in place: R in the upper triangle; Q (m x m) explicit. */
static void householder_qr(int m, int n, double* A, double* Q) {
  for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Q[i * m + j] = (i == j);
  double w[64]; /* m <= 62 here */
  for (int k = 0; k < n && k < m - 1; k++) {
    double nrm = 0;
    for (int i = k; i < m; i++) nrm += A[i * n + k] * A[i * n + k];
    nrm = sqrt(nrm);
    if (nrm == 0) continue;
    double alpha = (A[k * n + k] > 0) ? -nrm : nrm;
    for (int i = 0; i < m; i++) w[i] = 0;
    for (int i = k; i < m; i++) w[i] = A[i * n + k];
    w[k] -= alpha;
    double wn = 0;
    for (int i = k; i < m; i++) wn += w[i] * w[i];
    if (wn == 0) continue;
    for (int j = k; j < n; j++) {
      double s = 0;
      for (int i = k; i < m; i++) s += w[i] * A[i * n + j];
      s *= 2.0 / wn;
      for (int i = k; i < m; i++) A[i * n + j] -= s * w[i];
    }
    for (int i = 0; i < m; i++) { /* Q <- Q H */
      double s = 0;
      for (int l = k; l < m; l++) s += Q[i * m + l] * w[l];
      s *= 2.0 / wn;
      for (int l = k; l < m; l++) Q[i * m + l] -= s * w[l];
    }
  }
}

/* Goldfarb-Idnani dual active set.
 * min 1/2 z'Hz + g'z  s.t.  N z >= b, with H = R'R given through J = R^-1 (n x n) and the
 * unconstrained minimiser z.  N rows must have unit norm.  Returns status. */
#define GI_MAXN 16
#define GI_MAXM 48
static int gi_solve(int n, double* J, double* z, int m, const double* N, const double* b, int* iters_out) {
  int A[GI_MAXN], q = 0, active[GI_MAXM];
  double u[GI_MAXN + 1], Rq[GI_MAXN * GI_MAXN], d[GI_MAXN], zd[GI_MAXN], r[GI_MAXN];
  for (int i = 0; i < m; i++) active[i] = 0;
  int iters = 0, maxit = 10 * (m + n) + 20;
  for (;;) {
    /* step 1: most violated inactive constraint */
    double zinf = 0;
    for (int i = 0; i < n; i++) if (fabs(z[i]) > zinf) zinf = fabs(z[i]);
    double tol = 1e-13 * (1.0 + zinf);
    int p = -1; double sp = -tol;
    for (int i = 0; i < m; i++) {
      if (active[i]) continue;
      double s = -b[i];
      for (int k = 0; k < n; k++) s += N[i * n + k] * z[k];
      if (s < sp) { sp = s; p = i; }
    }
    if (p < 0) { *iters_out = iters; return 0; }
    const double* np = N + p * n;
    u[q] = 0;
    for (;;) { /* step 2 */
      if (++iters > maxit) { *iters_out = iters; return 1; }
      double dn = 0, d2n = 0;
      for (int k = 0; k < n; k++) {
        double s = 0;
        for (int i = 0; i < n; i++) s += J[i * n + k] * np[i];
        d[k] = s; dn += s * s;
        if (k >= q) d2n += s * s;
      }
      for (int i = 0; i < n; i++) {
        double s = 0;
        for (int k = q; k < n; k++) s += J[i * n + k] * d[k];
        zd[i] = s;
      }
      for (int k = q - 1; k >= 0; k--) { /* r = Rq^-1 d1 */
        double s = d[k];
        for (int j = k + 1; j < q; j++) s -= Rq[k * GI_MAXN + j] * r[j];
        r[k] = s / Rq[k * GI_MAXN + k];
      }
      int l = -1; double t1 = INFINITY;
      for (int k = 0; k < q; k++)
        if (r[k] > 0 && u[k] / r[k] < t1) { t1 = u[k] / r[k]; l = k; }
      int dependent = (d2n <= 1e-22 * dn) || q == n;
      double t2 = INFINITY;
      if (!dependent) {
        double znp = 0;
        for (int i = 0; i < n; i++) znp += zd[i] * np[i];
        t2 = -sp / znp;
      }
      double t = (t1 < t2) ? t1 : t2;
      if (!(t < INFINITY)) { *iters_out = iters; return 2; }
      for (int k = 0; k < q; k++) u[k] -= t * r[k];
      u[q] += t;
      if (!dependent) {
        for (int i = 0; i < n; i++) z[i] += t * zd[i];
      }
      if (!dependent && t == t2) {
        /* full step: add p.  Givens from the bottom so that d[q+1..] -> 0 */
        for (int j = n - 1; j > q; j--) {
          double a = d[j - 1], bb = d[j];
          if (bb == 0) continue;
          double h = hypot(a, bb), c = a / h, s = bb / h;
          d[j - 1] = h; d[j] = 0;
          for (int i = 0; i < n; i++) {
            double x = J[i * n + j - 1], y = J[i * n + j];
            J[i * n + j - 1] = c * x + s * y;
            J[i * n + j] = -s * x + c * y;
          }
        }
        for (int k = 0; k <= q; k++) Rq[k * GI_MAXN + q] = d[k];
        A[q] = p; active[p] = 1; q++;
        break;
      }
      /* partial (or pure dual) step: drop active constraint l */
      active[A[l]] = 0;
      for (int j = l; j < q - 1; j++) {
        A[j] = A[j + 1]; u[j] = u[j + 1];
        for (int k = 0; k <= j + 1; k++) Rq[k * GI_MAXN + j] = Rq[k * GI_MAXN + j + 1];
      }
      u[q - 1] = u[q];
      q--;
      u[q + 1] = 0;
      for (int j = l; j < q; j++) { /* re-triangularise rows j, j+1 */
        double a = Rq[j * GI_MAXN + j], bb = Rq[(j + 1) * GI_MAXN + j];
        if (bb == 0) continue;
        double h = hypot(a, bb), c = a / h, s = bb / h;
        for (int k = j; k < q; k++) {
          double x = Rq[j * GI_MAXN + k], y = Rq[(j + 1) * GI_MAXN + k];
          Rq[j * GI_MAXN + k] = c * x + s * y;
          Rq[(j + 1) * GI_MAXN + k] = -s * x + c * y;
        }
        for (int i = 0; i < n; i++) {
          double x = J[i * n + j], y = J[i * n + j + 1];
          J[i * n + j] = c * x + s * y;
          J[i * n + j + 1] = -s * x + c * y;
        }
      }
      if (!dependent) {
        sp = -b[p];
        for (int k = 0; k < n; k++) sp += np[k] * z[k];
      }
    }
  }
}

int orc_qp_solve(int n, int mls, const double* Als, const double* bls, double eps2, const double* dreg,
                 int me, const double* Aeq, const double* beq, int mi, const double* Ain,
                 const double* bin, double* x, int* iters, double* primal_res) {
  int nz = n - me, status = 0;
  *iters = 0;
  /* fixed workspaces (n <= 42, me <= 30, reduced rows <= 60): the hot loop never touches the heap */
  double At[ORC_NMAX * 30], Q[ORC_NMAX * ORC_NMAX], xp[ORC_NMAX];
  if (n > ORC_NMAX || me > 30) return 2;
  memset(xp, 0, sizeof xp);
  /* 1. null-space basis of the equalities: Aeq' = Q [R1; 0] */
  for (int i = 0; i < me; i++) for (int j = 0; j < n; j++) At[j * me + i] = Aeq[i * n + j];
  if (me > 0) householder_qr(n, me, At, Q);
  else for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) Q[i * n + j] = (i == j);
  if (me > 0) {
    double dmax = 0;
    for (int i = 0; i < me; i++) if (fabs(At[i * me + i]) > dmax) dmax = fabs(At[i * me + i]);
    double y[48];
    for (int i = 0; i < me; i++) { /* R1' y = beq, forward substitution */
      if (fabs(At[i * me + i]) <= 1e-11 * dmax) { status = 2; goto done; }
      double s = beq[i];
      for (int k = 0; k < i; k++) s -= At[k * me + i] * y[k];
      y[i] = s / At[i * me + i];
    }
    for (int i = 0; i < n; i++) {
      double s = 0;
      for (int k = 0; k < me; k++) s += Q[i * n + k] * y[k];
      xp[i] = s;
    }
  }
  if (nz > GI_MAXN || mi > GI_MAXM) { status = 2; goto done; }
  {
    /* 2. reduced least squares: rows [Als; sqrt(eps2*dreg_i) e_i'] Z, rhs [bls; 0] - rows*xp */
    int nreg = 0;
    for (int i = 0; i < n; i++) if (dreg[i] > 0) nreg++;
    int mB = mls + nreg;
    if (mB < nz) { status = 2; goto done; }
    double B[62 * GI_MAXN], rhs[62], QB[62 * 62];
    if (mB > 62) { status = 2; goto done; }
    memset(B, 0, sizeof B); memset(rhs, 0, sizeof rhs);
    for (int i = 0; i < mls; i++) {
      double s = bls[i];
      for (int j = 0; j < n; j++) s -= Als[i * n + j] * xp[j];
      rhs[i] = s;
      for (int k = 0; k < nz; k++) {
        double t = 0;
        for (int j = 0; j < n; j++) t += Als[i * n + j] * Q[j * n + me + k];
        B[i * nz + k] = t;
      }
    }
    int row = mls;
    for (int i = 0; i < n; i++) {
      if (!(dreg[i] > 0)) continue;
      double w = sqrt(eps2 * dreg[i]);
      rhs[row] = -w * xp[i];
      for (int k = 0; k < nz; k++) B[row * nz + k] = w * Q[i * n + me + k];
      row++;
    }
    householder_qr(mB, nz, B, QB);
    double J[GI_MAXN * GI_MAXN], z[GI_MAXN], y2[GI_MAXN];
    double rmax = 0;
    for (int i = 0; i < nz; i++) if (fabs(B[i * nz + i]) > rmax) rmax = fabs(B[i * nz + i]);
    for (int i = 0; i < nz; i++) if (fabs(B[i * nz + i]) <= 1e-13 * rmax) status = 2;
    if (status == 0) {
      for (int k = 0; k < nz; k++) { /* y2 = (QB' rhs)[0:nz] */
        double s = 0;
        for (int i = 0; i < mB; i++) s += QB[i * mB + k] * rhs[i];
        y2[k] = s;
      }
      for (int k = nz - 1; k >= 0; k--) { /* R z = y2 */
        double s = y2[k];
        for (int j = k + 1; j < nz; j++) s -= B[k * nz + j] * z[j];
        z[k] = s / B[k * nz + k];
      }
      for (int c = 0; c < nz; c++) { /* J = R^-1 by back substitution on unit vectors */
        for (int k = nz - 1; k >= 0; k--) {
          double s = (k == c) ? 1.0 : 0.0;
          for (int j = k + 1; j < nz; j++) s -= B[k * nz + j] * J[j * nz + c];
          J[k * nz + c] = s / B[k * nz + k];
        }
      }
      /* 3. inequalities in z:  -(Ain Z) z >= -(bin - Ain xp), rows normalised */
      double Nn[GI_MAXM * GI_MAXN], bb[GI_MAXM];
      for (int i = 0; i < mi; i++) {
        double s = bin[i], nn = 0;
        for (int j = 0; j < n; j++) s -= Ain[i * n + j] * xp[j];
        for (int k = 0; k < nz; k++) {
          double t = 0;
          for (int j = 0; j < n; j++) t += Ain[i * n + j] * Q[j * n + me + k];
          Nn[i * nz + k] = -t; nn += t * t;
        }
        nn = sqrt(nn);
        if (nn > 0) { for (int k = 0; k < nz; k++) Nn[i * nz + k] /= nn; bb[i] = -s / nn; }
        else bb[i] = (s >= 0) ? -1.0 : 1.0; /* 0 >= -s */
      }
      status = gi_solve(nz, J, z, mi, Nn, bb, iters);
      for (int i = 0; i < n; i++) {
        double s = xp[i];
        for (int k = 0; k < nz; k++) s += Q[i * n + me + k] * z[k];
        x[i] = s;
      }
    }
  }
done:
  if (status == 2 && primal_res) *primal_res = INFINITY;
  if (status != 2 && primal_res) {
    double res = 0;
    for (int i = 0; i < me; i++) {
      double s = -beq[i];
      for (int j = 0; j < n; j++) s += Aeq[i * n + j] * x[j];
      if (fabs(s) > res) res = fabs(s);
    }
    for (int i = 0; i < mi; i++) {
      double s = -bin[i];
      for (int j = 0; j < n; j++) s += Ain[i * n + j] * x[j];
      if (s > res) res = s;
    }
    *primal_res = res;
  }
  return status;
}

/* ------------------------------------------------------------------ the two control laws */
typedef struct {
  int nc, ns, cidx[4], sidx[4];
  double M[18 * 18], Cv[18], tau_g[18], S[12 * 18];
  double R_body[9], p_body[3], J_body[6 * 18], Jdv_body[6];
  double rpy[3], E[9], Einv[9], omega[3], pd_body[3], rpyd[3];
  double p_feet[4][3], J_feet[4][3 * 18], Jdv_feet[4][3], pd_feet[4][3];
  const double *p_body_nom, *pd_body_nom, *pdd_body_nom, *rpy_nom, *rpyd_nom, *rpydd_nom;
  const double *p_f_nom[4], *pd_f_nom[4], *pdd_f_nom[4];
} tick_common;

/* Shared head of both ControlLaw bodies: inverse_dynamics_controller.py:130-185 ==
 * mptc_controller.py:156-214 */
static void tick_head(const orc_model* m, const double* q, const double* v, const double* tg,
                      const int* contact, tick_common* T) {
  orc_calc_dynamics(m, q, v, T->M, T->Cv, T->tau_g);
  memset(T->S, 0, sizeof T->S);
  for (int k = 0; k < 12; k++) T->S[k * 18 + 6 + m->act_perm[k]] = 1.0; /* MakeActuationMatrix().T */
  T->nc = T->ns = 0;
  for (int i = 0; i < 4; i++) {
    if (contact[i]) T->cidx[T->nc++] = i; else T->sidx[T->ns++] = i;
  }
  T->p_body_nom = tg; T->pd_body_nom = tg + 3; T->pdd_body_nom = tg + 6;
  T->rpy_nom = tg + 9; T->rpyd_nom = tg + 12; T->rpydd_nom = tg + 15;
  for (int i = 0; i < 4; i++) {
    T->p_f_nom[i] = tg + 18 + 9 * i; T->pd_f_nom[i] = tg + 21 + 9 * i; T->pdd_f_nom[i] = tg + 24 + 9 * i;
  }
  orc_body_quantities(m, q, v, T->R_body, T->p_body, T->J_body, T->Jdv_body);
  double Jv[6];
  mv(6, 18, T->J_body, v, Jv);
  for (int i = 0; i < 3; i++) { T->omega[i] = Jv[i]; T->pd_body[i] = Jv[3 + i]; }
  orc_rpy_from_R(T->R_body, T->rpy);
  orc_rpy_E(T->rpy, T->E);
  rpy_Einv(T->rpy, T->Einv);
  mat3_vec(T->Einv, T->omega, T->rpyd);
  for (int i = 0; i < 4; i++) {
    orc_foot_quantities(m, q, v, i, T->p_feet[i], T->J_feet[i], T->Jdv_feet[i]);
    mv(3, 18, T->J_feet[i], v, T->pd_feet[i]);
  }
}

/* Constraint builders shared verbatim by both laws (inverse_dynamics_controller.py:48-101 ==
 * mptc_controller.py:70-123).  x = [vd(18); tau(12); f_1..f_nc]. */
static void build_constraints(const tick_common* T, const orc_params* p, const double* v, orc_qp* qp, int n) {
  int nc = T->nc;
  qp->n = n; qp->nc = nc;
  memset(qp->Aeq, 0, sizeof qp->Aeq); memset(qp->beq, 0, sizeof qp->beq);
  memset(qp->Ain, 0, sizeof qp->Ain); memset(qp->bin, 0, sizeof qp->bin);
  /* AddDynamicsConstraint: [M, -S', -J_c'] x = -Cv - tau_g */
  for (int i = 0; i < 18; i++) {
    for (int j = 0; j < 18; j++) qp->Aeq[i * n + j] = T->M[i * 18 + j];
    for (int k = 0; k < 12; k++) qp->Aeq[i * n + 18 + k] = -T->S[k * 18 + i];
    for (int c = 0; c < nc; c++)
      for (int a = 0; a < 3; a++) qp->Aeq[i * n + 30 + 3 * c + a] = -T->J_feet[T->cidx[c]][a * 18 + i];
    qp->beq[i] = -T->Cv[i] - T->tau_g[i];
  }
  qp->me = 18; qp->mi = 0;
  if (nc > 0) {
    /* AddFrictionPyramidConstraint */
    const double Ai[4][3] = {{1, 0, -p->mu}, {-1, 0, -p->mu}, {0, 1, -p->mu}, {0, -1, -p->mu}};
    for (int c = 0; c < nc; c++)
      for (int r = 0; r < 4; r++) {
        for (int a = 0; a < 3; a++) qp->Ain[(4 * c + r) * n + 30 + 3 * c + a] = Ai[r][a];
        qp->bin[4 * c + r] = 0.0;
      }
    qp->mi = 4 * nc;
    /* AddContactConstraint: J_c vd = -Kd J_c v - Jdv_c */
    for (int c = 0; c < nc; c++) {
      int f = T->cidx[c];
      for (int a = 0; a < 3; a++) {
        int row = 18 + 3 * c + a;
        for (int j = 0; j < 18; j++) qp->Aeq[row * n + j] = T->J_feet[f][a * 18 + j];
        qp->beq[row] = -p->Kd_contact * T->pd_feet[f][a] - T->Jdv_feet[f][a];
      }
    }
    qp->me = 18 + 3 * nc;
  }
  (void)v;
  /* optional torque box (NOT in the reference; tau_max = +inf reproduces it) */
  if (isfinite(p->tau_max)) {
    for (int k = 0; k < 12; k++) {
      qp->Ain[(qp->mi) * n + 18 + k] = 1.0;  qp->bin[qp->mi] = p->tau_max; qp->mi++;
      qp->Ain[(qp->mi) * n + 18 + k] = -1.0; qp->bin[qp->mi] = p->tau_max; qp->mi++;
    }
  }
}

static int solve_and_extract(const orc_params* p, orc_qp* qp, double* tau) {
  double dreg[ORC_NMAX];
  for (int i = 0; i < qp->n; i++) dreg[i] = (i >= 18) ? 1.0 : 0.0; /* tie-break on [tau; f] (and delta) */
  qp->status = orc_qp_solve(qp->n, qp->mls, qp->Als, qp->bls, p->tiebreak_eps2, dreg, qp->me, qp->Aeq,
                            qp->beq, qp->mi, qp->Ain, qp->bin, qp->x, &qp->iters, &qp->primal_res);
  for (int k = 0; k < 12; k++) tau[k] = (qp->status == 2) ? 0.0 : qp->x[18 + k];
  return qp->status;
}

/* inverse_dynamics_controller.py:103-234 */
int orc_id_control_law(const orc_model* m, const orc_params* p, const double* q, const double* v,
                       const double* targets, const int* contact, double* tau, double* metrics,
                       orc_qp* qp_out) {
  tick_common Ts, *T = &Ts;
  orc_qp qps, *qp = qp_out ? qp_out : &qps;
  tick_head(m, q, v, targets, contact, T);
  int nc = T->nc, ns = T->ns, n = 30 + 3 * nc;
  /* :187-197 desired task-space accelerations */
  double pdd_body_des[3], rpydd_des[3], omegad_des[3], vd_body_des[6], pdd_s_des[4][3];
  for (int i = 0; i < 3; i++) {
    pdd_body_des[i] = T->pdd_body_nom[i] - p->Kp_body_p * (T->p_body[i] - T->p_body_nom[i]) -
                      p->Kd_body_p * (T->pd_body[i] - T->pd_body_nom[i]);
    rpydd_des[i] = T->rpydd_nom[i] - p->Kp_body_rpy * (T->rpy[i] - T->rpy_nom[i]) -
                   p->Kd_body_rpy * (T->rpyd[i] - T->rpyd_nom[i]);
  }
  mat3_vec(T->E, rpydd_des, omegad_des); /* :192 no Edot term, as in the reference */
  for (int i = 0; i < 3; i++) { vd_body_des[i] = omegad_des[i]; vd_body_des[3 + i] = pdd_body_des[i]; }
  for (int s = 0; s < ns; s++) {
    int f = T->sidx[s];
    for (int i = 0; i < 3; i++)
      pdd_s_des[s][i] = T->pdd_f_nom[f][i] - p->Kp_foot * (T->p_feet[f][i] - T->p_f_nom[f][i]) -
                        p->Kd_foot * (T->pd_feet[f][i] - T->pd_f_nom[f][i]);
  }
  /* :199-211 costs: AddJacobianTypeCost  Q = w J'J, c = w J'(Jd_qd - xdd_des) on vd */
  memset(qp->Q, 0, sizeof qp->Q); memset(qp->c, 0, sizeof qp->c);
  memset(qp->Als, 0, sizeof qp->Als); memset(qp->bls, 0, sizeof qp->bls);
  int row = 0;
  for (int t = 0; t < 1 + ns; t++) {
    int k = (t == 0) ? 6 : 3;
    const double* J = (t == 0) ? T->J_body : T->J_feet[T->sidx[t - 1]];
    const double* Jdqd = (t == 0) ? T->Jdv_body : T->Jdv_feet[T->sidx[t - 1]];
    const double* des = (t == 0) ? vd_body_des : pdd_s_des[t - 1];
    double w = (t == 0) ? p->w_body : p->w_foot, sw = sqrt(w);
    for (int i = 0; i < 18; i++) {
      for (int j = 0; j < 18; j++) {
        double s = 0;
        for (int a = 0; a < k; a++) s += J[a * 18 + i] * J[a * 18 + j];
        qp->Q[i * n + j] += w * s;
      }
      double s = 0;
      for (int a = 0; a < k; a++) s += J[a * 18 + i] * (Jdqd[a] - des[a]);
      qp->c[i] += w * s;
    }
    for (int a = 0; a < k; a++) {
      for (int j = 0; j < 18; j++) qp->Als[row * n + j] = sw * J[a * 18 + j];
      qp->bls[row] = sw * (des[a] - Jdqd[a]);
      row++;
    }
  }
  qp->mls = row;
  build_constraints(T, p, v, qp, n); /* :213-221 */
  int status = solve_and_extract(p, qp, tau); /* :223-225 */
  /* :227-232 logging */
  double err = 0;
  for (int i = 0; i < 3; i++) {
    double a = T->rpy[i] - T->rpy_nom[i], b = T->p_body[i] - T->p_body_nom[i];
    err += a * a + b * b;
  }
  for (int s = 0; s < ns; s++)
    for (int i = 0; i < 3; i++) {
      double a = T->p_feet[T->sidx[s]][i] - T->p_f_nom[T->sidx[s]][i];
      err += a * a;
    }
  if (metrics) { metrics[0] = 0; metrics[1] = err; metrics[2] = qp->primal_res; metrics[3] = 0; }
  return status;
}

/* mptc_controller.py:125-310; with pc != 0 additionally pc_controller.py:14-40,202,229-237 */
static int mptc_like_control_law(const orc_model* m, const orc_params* p, const double* q, const double* v,
                                 const double* targets, const int* contact, double* tau, double* metrics,
                                 orc_qp* qp_out, int pc) {
  tick_common Ts, *T = &Ts;
  orc_qp qps, *qp = qp_out ? qp_out : &qps;
  tick_head(m, q, v, targets, contact, T);
  int nc = T->nc, ns = T->ns, n = 30 + 3 * nc + (pc ? 1 : 0), mt = 6 + 3 * ns, nf = 3 * ns;
  double C[18 * 18];
  orc_coriolis_matrix(m, q, v, C); /* :158 */
  /* :227-235 stacked task Jacobian J (mt x 18), Jd */
  double J[18 * 18], Jd[18 * 18];
  memset(J, 0, sizeof J); memset(Jd, 0, sizeof Jd);
  memcpy(J, T->J_body, sizeof(double) * 6 * 18); /* Jd_body = 0 (:186) */
  for (int s = 0; s < ns; s++) {
    memcpy(J + (6 + 3 * s) * 18, T->J_feet[T->sidx[s]], sizeof(double) * 3 * 18);
    orc_foot_jacobian_dot(m, q, v, T->sidx[s], Jd + (6 + 3 * s) * 18); /* :216-225 */
  }
  /* :237-240 */
  double Minv[18 * 18], JMi[18 * 18], Lam[18 * 18], Jbar[18 * 18], Qm[18 * 18], tmp[18 * 18];
  memcpy(Minv, T->M, sizeof Minv);
  int bad = mat_inv(18, Minv);
  mm(mt, 18, 18, J, Minv, JMi);          /* J Minv */
  mmt(mt, 18, mt, JMi, J, Lam);          /* J Minv J' */
  bad |= mat_inv(mt, Lam);               /* Lambda */
  /* (only under orc_set_status_convention(1), the default; 0 = the dense restatement's own status, nothing mirrored)
   * the product's status convention (include/wbc.h), mirrored: a leg whose 3x3 foot Jacobian block is singular to rounding
   * (|det| <= 1e-12: an exactly straight knee) makes J Minv J' singular up to rounding noise -- mat_inv above may or may not
   * notice -- and is reported as status 2 with zero torques, not solved */
  for (int l = 0; l < 4 && g_product_status; l++) {
    const double* B = T->J_feet[l] + 6 + 3 * l;   /* rows 18 apart */
    double det = B[0] * (B[19] * B[38] - B[20] * B[37]) - B[1] * (B[18] * B[38] - B[20] * B[36]) + B[2] * (B[18] * B[37] - B[19] * B[36]);
    if (!(fabs(det) > 1e-12)) bad = 1;
  }
  mtm(18, mt, mt, JMi, Lam, Jbar);       /* Jbar = Minv J' Lambda  (18 x mt); Minv symmetric */
  mm(mt, 18, 18, JMi, C, Qm);            /* Q = J Minv C - Jd */
  for (int i = 0; i < mt * 18; i++) Qm[i] -= Jd[i];
  /* :242-257 task-space states and errors */
  double x[18], xd[18], x_nom[18], xd_nom[18], xdd_nom[18], xt[18], xdt[18], t3[3];
  for (int i = 0; i < 3; i++) { x[i] = T->rpy[i]; x[3 + i] = T->p_body[i]; x_nom[i] = T->rpy_nom[i]; x_nom[3 + i] = T->p_body_nom[i]; }
  mat3_vec(T->E, T->rpyd, t3);      for (int i = 0; i < 3; i++) { xd[i] = t3[i]; xd[3 + i] = T->pd_body[i]; }
  mat3_vec(T->E, T->rpyd_nom, t3);  for (int i = 0; i < 3; i++) { xd_nom[i] = t3[i]; xd_nom[3 + i] = T->pd_body_nom[i]; }
  mat3_vec(T->E, T->rpydd_nom, t3); for (int i = 0; i < 3; i++) { xdd_nom[i] = t3[i]; xdd_nom[3 + i] = T->pdd_body_nom[i]; }
  for (int s = 0; s < ns; s++) {
    int f = T->sidx[s];
    for (int i = 0; i < 3; i++) {
      x[6 + 3 * s + i] = T->p_feet[f][i];  x_nom[6 + 3 * s + i] = T->p_f_nom[f][i];
      xd[6 + 3 * s + i] = T->pd_feet[f][i]; xd_nom[6 + 3 * s + i] = T->pd_f_nom[f][i];
      xdd_nom[6 + 3 * s + i] = T->pdd_f_nom[f][i];
    }
  }
  for (int i = 0; i < mt; i++) { xt[i] = x[i] - x_nom[i]; xdt[i] = xd[i] - xd_nom[i]; }
  /* :259-268 Kp, Kd, W (all diagonal) */
  double Kp[18], Kd[18], W[18];
  for (int i = 0; i < mt; i++) {
    Kp[i] = (i < 3) ? p->Kp_body_rpy : (i < 6 ? p->Kp_body_p : p->Kp_foot);
    Kd[i] = (i < 3) ? p->Kd_body_rpy : (i < 6 ? p->Kd_body_p : p->Kd_foot);
    W[i] = (i < 6) ? p->w_body : p->w_foot;
  }
  (void)nf;
  /* :272 f_des = Lam xdd_nom + Lam Q (v - Jbar xd_tilde) + Jbar' tau_g - Kp x_tilde - Kd xd_tilde */
  double f_des[18], a18[18], b18[18], c18[18];
  mv(18, mt, Jbar, xdt, a18);
  for (int i = 0; i < 18; i++) a18[i] = v[i] - a18[i];
  mv(mt, 18, Qm, a18, b18);
  for (int i = 0; i < mt; i++) b18[i] += xdd_nom[i];
  mv(mt, mt, Lam, b18, f_des);
  mtv(18, mt, Jbar, T->tau_g, c18);
  for (int i = 0; i < mt; i++) f_des[i] += c18[i] - Kp[i] * xt[i] - Kd[i] * xdt[i];
  /* :274-282 AddTaskForceCost: U = [S', Jc'], G = Jbar' U, Q = G' W G, c = -G' W f_des on [tau; f] */
  int nu = 12 + 3 * nc;
  double U[18 * 24], G[18 * 24];
  memset(U, 0, sizeof U);
  for (int i = 0; i < 18; i++) {
    for (int k = 0; k < 12; k++) U[i * nu + k] = T->S[k * 18 + i];
    for (int c = 0; c < nc; c++)
      for (int a = 0; a < 3; a++) U[i * nu + 12 + 3 * c + a] = T->J_feet[T->cidx[c]][a * 18 + i];
  }
  mtm(mt, 18, nu, Jbar, U, G);
  memset(qp->Q, 0, sizeof qp->Q); memset(qp->c, 0, sizeof qp->c);
  memset(qp->Als, 0, sizeof qp->Als); memset(qp->bls, 0, sizeof qp->bls);
  for (int i = 0; i < nu; i++) {
    for (int j = 0; j < nu; j++) {
      double s = 0;
      for (int a = 0; a < mt; a++) s += G[a * nu + i] * W[a] * G[a * nu + j];
      qp->Q[(18 + i) * n + 18 + j] = s;
    }
    double s = 0;
    for (int a = 0; a < mt; a++) s += G[a * nu + i] * W[a] * f_des[a];
    qp->c[18 + i] = -s;
  }
  for (int a = 0; a < mt; a++) {
    double sw = sqrt(W[a]);
    for (int j = 0; j < nu; j++) qp->Als[a * n + 18 + j] = sw * G[a * nu + j];
    qp->bls[a] = sw * f_des[a];
  }
  qp->mls = mt;
  build_constraints(T, p, v, qp, n); /* :284-292 */
  if (pc) {
    /* pc_controller.py:14-40 AddVdotConstraint: xd_tilde' Jbar' U [tau; f] - delta <= ub ;  :233-237 delta <= 0 */
    double a18[18], b18[18], c18[18], d18[18], ub = 0;
    int r = qp->mi;
    for (int j = 0; j < nu; j++) {
      double sj = 0;
      for (int a = 0; a < mt; a++) sj += xdt[a] * G[a * nu + j];
      qp->Ain[r * n + 18 + j] = sj;
    }
    qp->Ain[r * n + n - 1] = -1.0;
    mtv(18, mt, Jbar, T->tau_g, a18);          /* Jbar' tau_g */
    mv(18, mt, Jbar, xdt, b18);                /* Jbar xd_tilde - v */
    for (int i = 0; i < 18; i++) b18[i] -= v[i];
    mv(mt, 18, Qm, b18, c18);
    mv(mt, mt, Lam, c18, d18);                 /* Lam Q (Jbar xd_tilde - v) */
    mv(mt, mt, Lam, xdd_nom, c18);             /* Lam xdd_nom */
    for (int a = 0; a < mt; a++) ub += xdt[a] * (a18[a] - d18[a] + c18[a] - Kp[a] * xt[a]);
    qp->bin[r] = ub;
    qp->Ain[(r + 1) * n + n - 1] = 1.0;
    qp->bin[r + 1] = 0.0;
    qp->mi += 2;
  }
  int status = bad ? 2 : solve_and_extract(p, qp, tau); /* :294-296 */
  if (bad) { for (int k = 0; k < 12; k++) tau[k] = 0; qp->status = 2; }
  /* the product's status convention (include/wbc.h), mirrored so that the checker and the checked agree on it: a solved tick
   * with a nearly straight knee (|sin| < 1e-4: inv(J Minv J') above has lost its digits) is reported as 3 */
  if (status == 0 && g_product_status)
    for (int l = 0; l < 4; l++)
      if (fabs(sin(q[7 + 3 * l + 2])) < 1e-4) status = 3;
  if (status == 3) qp->status = 3;
  /* :298-308 logging (pc_controller.py:240-252 is identical) */
  if (metrics) {
    double V = 0, err = 0, Vdot = 0;
    mv(mt, mt, Lam, xdt, a18);
    for (int i = 0; i < mt; i++) { V += 0.5 * xdt[i] * a18[i] + 0.5 * Kp[i] * xt[i] * xt[i]; err += xt[i] * xt[i]; }
    double u18[18], fvec[18], fg[18], d18[18], e18[18];
    mv(18, nu, U, qp->x + 18, u18);       /* u = S' tau + Jc' fc */
    mtv(18, mt, Jbar, u18, fvec);          /* f = Jbar' u */
    mtv(18, mt, Jbar, T->tau_g, fg);
    mv(18, mt, Jbar, xdt, d18);
    for (int i = 0; i < 18; i++) d18[i] -= v[i];
    mv(mt, 18, Qm, d18, e18);
    mv(mt, mt, Lam, e18, d18);             /* Lam Q (Jbar xdt - v) */
    mv(mt, mt, Lam, xdd_nom, e18);
    for (int i = 0; i < mt; i++) Vdot += xdt[i] * (fvec[i] - fg[i] + d18[i] - e18[i] + Kp[i] * xt[i]);
    metrics[0] = V; metrics[1] = err; metrics[2] = 0.0; metrics[3] = Vdot;
    if (status == 2) { metrics[3] = 0; }
  }
  (void)tmp;
  return status;
}

int orc_mptc_control_law(const orc_model* m, const orc_params* p, const double* q, const double* v,
                         const double* targets, const int* contact, double* tau, double* metrics,
                         orc_qp* qp_out) {
  return mptc_like_control_law(m, p, q, v, targets, contact, tau, metrics, qp_out, 0);
}
int orc_pc_control_law(const orc_model* m, const orc_params* p, const double* q, const double* v,
                       const double* targets, const int* contact, double* tau, double* metrics,
                       orc_qp* qp_out) {
  return mptc_like_control_law(m, p, q, v, targets, contact, tau, metrics, qp_out, 1);
}


/* ------------------------------------------------------------------ CLF law */
void orc_clf_care(double qp, double qd, double r, double* p11, double* p12, double* p22) {
  /* F = [[0 1],[0 0]], G = [0;1]:  0 = F'P + PF - P G R^-1 G' P + Q  ->
   *   p12^2 / r = qp ;  2 p12 - p22^2 / r + qd = 0 ;  p11 = p12 p22 / r            */
  *p12 = sqrt(qp * r);
  *p22 = sqrt(r * (qd + 2.0 * *p12));
  *p11 = *p12 * *p22 / r;
}

/* clf_controller.py:48-234 */
int orc_clf_control_law(const orc_model* m, const orc_params* p, const double* q, const double* v,
                        const double* targets, const int* contact, double* tau, double* metrics,
                        orc_qp* qp_out) {
  /* :65-73 tuning literals */
  const double Q_body_p = 5000.0, Q_body_pd = 200.0, Q_body_rpy = 5000.0, Q_body_rpyd = 200.0;
  const double Q_foot_p = 200.0, Q_foot_pd = 20.0, rr = 1.0, w_delta = 1000.0;
  tick_common Ts, *T = &Ts;
  orc_qp qps, *qp = qp_out ? qp_out : &qps;
  tick_head(m, q, v, targets, contact, T);
  int nc = T->nc, ns = T->ns, n = 31 + 3 * nc, mt = 6 + 3 * ns;
  /* :131-137 J, Jdv */
  double J[18 * 18], Jdv[18];
  memset(J, 0, sizeof J);
  memcpy(J, T->J_body, sizeof(double) * 6 * 18);
  for (int i = 0; i < 6; i++) Jdv[i] = T->Jdv_body[i];
  for (int s = 0; s < ns; s++) {
    memcpy(J + (6 + 3 * s) * 18, T->J_feet[T->sidx[s]], sizeof(double) * 3 * 18);
    for (int i = 0; i < 3; i++) Jdv[6 + 3 * s + i] = T->Jdv_feet[T->sidx[s]][i];
  }
  /* :139-157 task-space states and errors (same construction as the MPTC law) */
  double x[18], xd[18], x_nom[18], xd_nom[18], xdd_nom[18], xt[18], xdt[18], t3[3];
  for (int i = 0; i < 3; i++) { x[i] = T->rpy[i]; x[3 + i] = T->p_body[i]; x_nom[i] = T->rpy_nom[i]; x_nom[3 + i] = T->p_body_nom[i]; }
  mat3_vec(T->E, T->rpyd, t3);      for (int i = 0; i < 3; i++) { xd[i] = t3[i]; xd[3 + i] = T->pd_body[i]; }
  mat3_vec(T->E, T->rpyd_nom, t3);  for (int i = 0; i < 3; i++) { xd_nom[i] = t3[i]; xd_nom[3 + i] = T->pd_body_nom[i]; }
  mat3_vec(T->E, T->rpydd_nom, t3); for (int i = 0; i < 3; i++) { xdd_nom[i] = t3[i]; xdd_nom[3 + i] = T->pdd_body_nom[i]; }
  for (int s = 0; s < ns; s++) {
    int f = T->sidx[s];
    for (int i = 0; i < 3; i++) {
      x[6 + 3 * s + i] = T->p_feet[f][i];  x_nom[6 + 3 * s + i] = T->p_f_nom[f][i];
      xd[6 + 3 * s + i] = T->pd_feet[f][i]; xd_nom[6 + 3 * s + i] = T->pd_f_nom[f][i];
      xdd_nom[6 + 3 * s + i] = T->pdd_f_nom[f][i];
    }
  }
  for (int i = 0; i < mt; i++) { xt[i] = x[i] - x_nom[i]; xdt[i] = xd[i] - xd_nom[i]; }
  /* :159-189 Q, R, CARE -> P (block diagonal per task dimension), gamma */
  double P11[18], P12[18], P22[18], qmin = 1e300, pmax = 0;
  for (int i = 0; i < mt; i++) {
    double qp_i = (i < 3) ? Q_body_rpy : (i < 6 ? Q_body_p : Q_foot_p);
    double qd_i = (i < 3) ? Q_body_rpyd : (i < 6 ? Q_body_pd : Q_foot_pd);
    orc_clf_care(qp_i, qd_i, rr, &P11[i], &P12[i], &P22[i]);
    if (qp_i < qmin) qmin = qp_i;
    if (qd_i < qmin) qmin = qd_i;
    double h = 0.5 * (P11[i] + P22[i]), d = 0.5 * (P11[i] - P22[i]);
    double ev = h + sqrt(d * d + P12[i] * P12[i]);
    if (ev > pmax) pmax = ev;
  }
  double gamma = qmin / pmax; /* :188 */
  /* :199 xdd_des = xdd_nom - R^-1 G'P eta ;  :15-25 a = 2 eta'PG J */
  double xdd_des[18], gt[18], V = 0, ePFe = 0;
  for (int i = 0; i < mt; i++) {
    double pg = P12[i] * xt[i] + P22[i] * xdt[i];
    xdd_des[i] = xdd_nom[i] - pg / rr;
    gt[i] = 2.0 * pg;
    V += P11[i] * xt[i] * xt[i] + 2.0 * P12[i] * xt[i] * xdt[i] + P22[i] * xdt[i] * xdt[i];
    ePFe += P11[i] * xt[i] * xdt[i] + P12[i] * xdt[i] * xdt[i];
  }
  memset(qp->Q, 0, sizeof qp->Q); memset(qp->c, 0, sizeof qp->c);
  memset(qp->Als, 0, sizeof qp->Als); memset(qp->bls, 0, sizeof qp->bls);
  /* :200 AddJacobianTypeCost(J, vd, Jdv, xdd_des, 1.0) ; :203 AddVdotCost ; :206 w_delta delta^2 */
  for (int i = 0; i < 18; i++) {
    for (int j = 0; j < 18; j++) {
      double s = 0;
      for (int a = 0; a < mt; a++) s += J[a * 18 + i] * J[a * 18 + j];
      qp->Q[i * n + j] = s;
    }
    double s = 0;
    for (int a = 0; a < mt; a++) s += J[a * 18 + i] * (Jdv[a] - xdd_des[a] + gt[a]);
    qp->c[i] = s;
  }
  qp->Q[(n - 1) * n + n - 1] = 2.0 * w_delta;
  /* square-root form: 1/2 |J vd - (xdd_des - Jdv - gt)|^2 + 1/2 (sqrt(2 w) delta)^2 */
  for (int a = 0; a < mt; a++) {
    for (int j = 0; j < 18; j++) qp->Als[a * n + j] = J[a * 18 + j];
    qp->bls[a] = xdd_des[a] - Jdv[a] - gt[a];
  }
  if (mt < 18) {
    qp->Als[mt * n + n - 1] = sqrt(2.0 * w_delta);
    qp->bls[mt] = 0.0;
    qp->mls = mt + 1;
  } else {
    /* nc = 0: all 18 task rows are in use; fold the delta row into the tie-break weights below */
    qp->mls = mt;
  }
  build_constraints(T, p, v, qp, n); /* :212-221 */
  {
    /* :27-45, :209 AddVdotConstraint: 2 eta'PG J vd - delta <= -gamma V - 2 eta'PF eta - 2 eta'PG (Jdv - xdd_nom) */
    int r = qp->mi;
    double ub = -gamma * V - 2.0 * ePFe;
    for (int j = 0; j < 18; j++) {
      double s = 0;
      for (int a = 0; a < mt; a++) s += gt[a] * J[a * 18 + j];
      qp->Ain[r * n + j] = s;
    }
    for (int a = 0; a < mt; a++) ub -= gt[a] * (Jdv[a] - xdd_nom[a]);
    qp->Ain[r * n + n - 1] = -1.0;
    qp->bin[r] = ub;
    qp->mi += 1;
  }
  double dreg[ORC_NMAX];
  for (int i = 0; i < n; i++) dreg[i] = (i >= 18 && i < n - 1) ? 1.0 : 0.0; /* tie-break on [tau; f] only */
  if (mt == 18) dreg[n - 1] = 2.0 * w_delta / p->tiebreak_eps2;             /* exact delta cost when no LS row is left */
  qp->status = orc_qp_solve(n, qp->mls, qp->Als, qp->bls, p->tiebreak_eps2, dreg, qp->me, qp->Aeq, qp->beq, qp->mi,
                            qp->Ain, qp->bin, qp->x, &qp->iters, &qp->primal_res);
  for (int k = 0; k < 12; k++) tau[k] = (qp->status == 2) ? 0.0 : qp->x[18 + k];
  if (metrics) {
    double err = 0, Vdot = 2.0 * ePFe, Jvd[18];
    mv(mt, 18, J, qp->x, Jvd);
    for (int i = 0; i < mt; i++) { err += xt[i] * xt[i]; Vdot += gt[i] * (Jvd[i] + Jdv[i] - xdd_nom[i]); }
    metrics[0] = V; metrics[1] = err; metrics[2] = 0.0; metrics[3] = (qp->status == 2) ? 0.0 : Vdot;
  }
  return qp->status;
}

/* ------------------------------------------------------------------ batched driver */
/* The product's convention for instances that cannot be answered with finite numbers (include/wbc.h "Malformed instances"), MIRRORED here like status 3
 * so that checker and checked can be compared instance by instance (g_product_status; the reference itself asserts, inverse_dynamics_controller.py:224,
 * or hands NaN on): a value that is not a finite number in anything the law READS -- q, v, the body targets, the targets of the SWING feet (the
 * reference indexes p_feet_nom[swing_feet], inverse_dynamics_controller.py:152-154: a contact foot's targets are never read) --, a mu or mass scale
 * that is not a positive finite number, or finite inputs whose torques or metrics overflow -> status 2, zero torques, zero metrics. */
static int not_finite(double x) { return !(fabs(x) < HUGE_VAL); }
static int malformed_instance(const double* q, const double* v, const double* tg, const int* ct, double mu, double ms) {
  for (int k = 0; k < 19; k++) if (not_finite(q[k])) return 1;
  for (int k = 0; k < 18; k++) if (not_finite(v[k])) return 1;
  for (int k = 0; k < 18; k++) if (not_finite(tg[k])) return 1;
  for (int f = 0; f < 4; f++)
    if (!ct[f]) for (int k = 0; k < 9; k++) if (not_finite(tg[18 + 9 * f + k])) return 1;
  if (!(mu > 0.0) || not_finite(mu) || !(ms > 0.0) || not_finite(ms)) return 1;
  double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  return !(n2 > 0.0) || not_finite(n2);   /* a quaternion without a direction: 2 / |q|^2 is inf or 0 x inf */
}
static int one_instance(const orc_model* m, const orc_params* p, int kind, const double* qi, const double* vi, const double* tg, const int* ct,
                        double mu_i, double ms_i, double* ti, double* mi) {
  double tgs[54];
  if (g_product_status) {
    if (malformed_instance(qi, vi, tg, ct, mu_i, ms_i)) {
      for (int k = 0; k < 12; k++) ti[k] = 0.0;
      for (int k = 0; k < 4; k++) mi[k] = 0.0;
      return 2;
    }
    /* what the law does not read may hold anything (the restatement below multiplies some of it by a zero weight, which a NaN survives) */
    memcpy(tgs, tg, sizeof tgs);
    for (int f = 0; f < 4; f++)
      if (ct[f]) for (int k = 0; k < 9; k++) if (not_finite(tgs[18 + 9 * f + k])) tgs[18 + 9 * f + k] = 0.0;
    tg = tgs;
  }
  int st = (kind == 0) ? orc_id_control_law(m, p, qi, vi, tg, ct, ti, mi, NULL)
           : (kind == 1) ? orc_mptc_control_law(m, p, qi, vi, tg, ct, ti, mi, NULL)
           : (kind == 2) ? orc_pc_control_law(m, p, qi, vi, tg, ct, ti, mi, NULL)
                         : orc_clf_control_law(m, p, qi, vi, tg, ct, ti, mi, NULL);
  if (g_product_status) {
    int bad = 0;
    for (int k = 0; k < 12; k++) bad |= not_finite(ti[k]);
    bad |= not_finite(mi[0] + mi[1] + mi[3]);
    if (bad) st = 2;
    if (st == 2) {   /* the product reports zeros throughout for an instance it could not answer */
      for (int k = 0; k < 12; k++) ti[k] = 0.0;
      for (int k = 0; k < 4; k++) mi[k] = 0.0;
    }
  }
  return st;
}
int orc_step_batch(const orc_model* m, const orc_params* p, int kind, int n, int stride, const double* q,
                   const double* v, const double* targets, const unsigned char* mask, const double* mu,
                   const double* mass_scale, double* tau, double* metrics, int* status, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 8)
  for (int i = 0; i < n; i++) {
    double qi[19], vi[18], tg[54], ti[12], mi[4];
    int ct[4];
    for (int k = 0; k < 19; k++) qi[k] = q[(size_t)k * stride + i];
    for (int k = 0; k < 18; k++) vi[k] = v[(size_t)k * stride + i];
    for (int k = 0; k < 54; k++) tg[k] = targets[(size_t)k * stride + i];
    for (int k = 0; k < 4; k++) ct[k] = (mask[i] >> k) & 1;
    orc_model ml = *m;
    orc_params pl = *p;
    if (mu) pl.mu = mu[i];
    if (mass_scale) {
      ml.base_mass *= mass_scale[i];
      for (int k = 0; k < 6; k++) ml.base_I[k] *= mass_scale[i];
    }
    int st = one_instance(&ml, &pl, kind, qi, vi, tg, ct, pl.mu, mass_scale ? mass_scale[i] : 1.0, ti, mi);
    for (int k = 0; k < 12; k++) tau[(size_t)k * stride + i] = ti[k];
    if (metrics) for (int k = 0; k < 4; k++) metrics[(size_t)k * stride + i] = mi[k];
    if (status) status[i] = st;
  }
  return 0;
}

/* Timing driver for bench.py's cpu_baseline leg: `reps` passes over the n instances inside ONE
 * OpenMP parallel region (the outputs of the last pass are kept). */
int orc_bench_batch(const orc_model* m, const orc_params* p, int kind, int n, int stride, const double* q,
                    const double* v, const double* targets, const unsigned char* mask, const double* mu,
                    const double* mass_scale, double* tau, int* status, int nthreads, int reps) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
  long total = (long)n * reps;
#pragma omp parallel for schedule(dynamic, 8)
  for (long j = 0; j < total; j++) {
    int i = (int)(j % n);
    double qi[19], vi[18], tg[54], ti[12], mi[4];
    int ct[4];
    for (int k = 0; k < 19; k++) qi[k] = q[(size_t)k * stride + i];
    for (int k = 0; k < 18; k++) vi[k] = v[(size_t)k * stride + i];
    for (int k = 0; k < 54; k++) tg[k] = targets[(size_t)k * stride + i];
    for (int k = 0; k < 4; k++) ct[k] = (mask[i] >> k) & 1;
    orc_model ml = *m;
    orc_params pl = *p;
    if (mu) pl.mu = mu[i];
    if (mass_scale) {
      ml.base_mass *= mass_scale[i];
      for (int k = 0; k < 6; k++) ml.base_I[k] *= mass_scale[i];
    }
    int st = one_instance(&ml, &pl, kind, qi, vi, tg, ct, pl.mu, mass_scale ? mass_scale[i] : 1.0, ti, mi);
    if (j >= total - n) {
      for (int k = 0; k < 12; k++) tau[(size_t)k * stride + i] = ti[k];
      if (status) status[i] = st;
    }
  }
  return 0;
}
