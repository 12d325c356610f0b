/*
 * wbc_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, double-precision restatement of the per-control-tick hot path of
 * vincekurtz/quadruped_drake:
 *   controllers/basic_controller.py:89-132,173-269
 *   controllers/inverse_dynamics_controller.py:19-234
 *   controllers/mptc_controller.py:20-310
 *   helpers.py:5-33
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it.
 *
 * PARITY UNPINNED AT THE DRAKE / OSQP BOUNDARY: the rigid-body numbers and the QP solve of
 * the reference live in Drake MultibodyPlant and OSQP (both un-vendored, unpinned, absent
 * here) and the reference ships no golden vectors for this path.  What IS pinned by the
 * reference itself: everything its own Python defines above that boundary -- targets,
 * gains, RPY handling, Lambda / Jbar / Q / f_des, the Coriolis-matrix and Jdot definitions,
 * QP assembly, logging -- against outputs of the reference's controller modules EXECUTED in
 * this container over stand-ins for the plant and the solver
 * (tests/golden/make_reference_law_golden.py, tests/test_reference_law.py: torques within
 * 9e-7, all four laws, every contact mask, both robots).  Below the boundary this oracle
 * restates Drake's *documented* semantics and is backed by (1) analytic known answers,
 * (2) an independent energy-based numpy derivation and (3) scipy QP solves -- see tests/
 * and DESIGN.md.
 *
 * Conventions (Drake's, as the reference relies on them):
 *   q = [qw qx qy qz | x y z | 12 joints]      (simulate.py:171-176)
 *   v = [w_WB (world) | v_WBo (world) | 12 joint rates]   (mptc_controller.py:190,194)
 *   joints are in canonical leg-major order [LF RF LH RH] x [abduct/HAA, hip/HFE, knee/KFE]
 *   M vd + Cv + tau_g = S' tau + sum_j J_cj' f_j              (basic_controller.py:106)
 * Matrices are row-major.
 */
#ifndef WBC_ORACLE_H
#define WBC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NQ 19
#define ORC_NV 18
#define ORC_NU 12
#define ORC_MODEL_FLAT 215
#define ORC_NTARGET 54

typedef struct {
  double off[3], axis[3], mass, com[3], I[6]; /* I = [xx yy zz xy xz yz] about link origin */
} orc_link;

typedef struct {
  double base_mass, base_com[3], base_I[6];
  orc_link link[4][3];
  double foot_off[4][3];
  double gravity;
  int act_perm[12]; /* actuator k drives canonical joint act_perm[k] */
} orc_model;

/* Gains/weights: literals of the two ControlLaw bodies. */
typedef struct {
  double Kp_body_p, Kd_body_p, Kp_body_rpy, Kd_body_rpy, Kp_foot, Kd_foot;
  double w_body, w_foot;
  double mu;          /* 0.7: inverse_dynamics_controller.py:19, mptc_controller.py:20 */
  double Kd_contact;  /* 100: inverse_dynamics_controller.py:93 */
  double tau_max;     /* +inf = reference behaviour (no torque rows) */
  double tiebreak_eps2; /* weight of the 1/2*eps2*|[tau;f]|^2 tie-break; see DESIGN.md */
} orc_params;

void orc_set_status_convention(int product);   /* 1 (default): include/wbc.h's status 2 / 3 on straight knees mirrored; 0: the dense solver's own */
void orc_model_from_flat(const double* flat215, orc_model* m);
void orc_params_id_default(orc_params* p);   /* inverse_dynamics_controller.py:117-127 */
void orc_params_mptc_default(orc_params* p); /* mptc_controller.py:143-153 */

/* basic_controller.py:101-115  (M: 18x18, Cv: 18, tau_g: 18 with the reference's sign) */
void orc_calc_dynamics(const orc_model* m, const double* q, const double* v, double* M, double* Cv,
                       double* tau_g);
/* generic inverse dynamics tau = M vd + Cv + grav*tau_g (what Drake's CalcInverseDynamics does) */
void orc_inverse_dynamics(const orc_model* m, const double* q, const double* v, const double* vd,
                          int with_gravity, double* tau);
/* basic_controller.py:117-132: C = 1/2 d(Cv)/dv (18x18) */
void orc_coriolis_matrix(const orc_model* m, const double* q, const double* v, double* C);
/* basic_controller.py:173-196 for foot in [LF RF LH RH]: p(3), J(3x18), Jdv(3) */
void orc_foot_quantities(const orc_model* m, const double* q, const double* v, int foot, double* p,
                         double* J, double* Jdv);
/* basic_controller.py:198-220: Jd (3x18) */
void orc_foot_jacobian_dot(const orc_model* m, const double* q, const double* v, int foot, double* Jd);
/* basic_controller.py:246-269 for the floating body frame: R(3x3), p(3), J(6x18), Jdv(6) */
void orc_body_quantities(const orc_model* m, const double* q, const double* v, double* R, double* p,
                         double* J, double* Jdv);
/* RollPitchYaw helpers */
void orc_rpy_from_R(const double* R, double* rpy);
void orc_rpy_E(const double* rpy, double* E /*3x3: omega = E rpyDt*/);

/* Literal QP data of one tick (n = 30 + 3 nc).  Filled by the control laws for inspection. */
#define ORC_NMAX 43 /* 18 + 12 + 12 (+ 1 slack delta of the PC law) */
#define ORC_MIN 42  /* 16 friction + 24 torque box + 2 PC rows */
typedef struct {
  int n, nc, me, mi, mls;
  double Q[ORC_NMAX * ORC_NMAX], c[ORC_NMAX]; /* 1/2 x'Qx + c'x, exactly as the reference adds them */
  double Aeq[30 * ORC_NMAX], beq[30];         /* dynamics rows then contact rows */
  double Ain[ORC_MIN * ORC_NMAX], bin[ORC_MIN]; /* Ain x <= bin: friction (+ PC rows, + optional torque box) */
  double Als[18 * ORC_NMAX], bls[18];         /* square-root form of the cost: 1/2|Als x - bls|^2 = cost + const */
  double x[ORC_NMAX];                         /* solution */
  int iters, status;
  double primal_res;
} orc_qp;

/* One control tick.  targets[54]: body p,pd,pdd,rpy,rpyd,rpydd (18) then per foot
 * [LF RF LH RH]: p,pd,pdd (9 each)  -- planners/simple.py:45-85.
 * contact[4] in [LF RF LH RH].  tau[12] in actuator order.  metrics[4] = [V, err, res, Vdot].
 * Returns status: 0 optimal, 1 iteration cap, 2 numerically infeasible/singular. */
int orc_id_control_law(const orc_model* m, const orc_params* p, const double* q, const double* v,
                       const double* targets, const int* contact, double* tau, double* metrics,
                       orc_qp* qp_out /* nullable */);
int orc_mptc_control_law(const orc_model* m, const orc_params* p, const double* q, const double* v,
                         const double* targets, const int* contact, double* tau, double* metrics,
                         orc_qp* qp_out /* nullable */);
/* controllers/pc_controller.py:44-255: MPTC + slack delta, rows Vdot <= delta and delta <= 0
 * (x = [vd; tau; f; delta], n = 31 + 3 nc).  delta carries no cost in the reference (:207-217). */
int orc_pc_control_law(const orc_model* m, const orc_params* p, const double* q, const double* v,
                       const double* targets, const int* contact, double* tau, double* metrics,
                       orc_qp* qp_out /* nullable */);

/* controllers/clf_controller.py:48-234: CLF-QP inverse dynamics.  x = [vd; tau; f; delta].
 * Gains are the literals of :65-73 (Q_body_p 5000, Q_body_pd 200, Q_foot_p 200, Q_foot_pd 20,
 * r 1, w_delta 1000); mu / Kd_contact / tiebreak_eps2 come from `p`.  The CARE of :187 has the
 * closed form of a double integrator with diagonal Q, R (checked against scipy in the tests).
 * metrics = [V, err, 0, Vdot] (:227-230). */
int orc_clf_control_law(const orc_model* m, const orc_params* p, const double* q, const double* v,
                        const double* targets, const int* contact, double* tau, double* metrics,
                        orc_qp* qp_out /* nullable */);
/* closed-form CARE blocks for one task dimension: P = [[p11 p12],[p12 p22]] */
void orc_clf_care(double qp, double qd, double r, double* p11, double* p12, double* p22);

/* Generic dense QP used by both laws:
 *   min 1/2|Als x - bls|^2 + 1/2 eps2 sum_i dreg[i] x_i^2   s.t. Aeq x = beq, Ain x <= bin
 * (null-space elimination of the equalities, QR of the stacked square-root form,
 *  Goldfarb-Idnani dual active set).  Returns status as above. */
int orc_qp_solve(int n, int mls, const double* Als, const double* bls, double eps2, const double* dreg,
                 int me, const double* Aeq, const double* beq, int mi, const double* Ain,
                 const double* bin, double* x, int* iters, double* primal_res);

/* Batched driver over SoA arrays (batch index fastest), OpenMP over instances when built
 * with -fopenmp.  kind: 0 = ID, 1 = MPTC, 2 = PC, 3 = CLF.  mask bit i = foot i in contact.
 * mu / mass_scale may be NULL.  Used by tests and by bench.py's cpu_baseline leg. */
int orc_step_batch(const orc_model* m, const orc_params* p, int kind, int n, int stride, const double* q,
                   const double* v, const double* targets, const unsigned char* mask, const double* mu,
                   const double* mass_scale, double* tau, double* metrics, int* status, int nthreads);

/* Timing driver (bench.py cpu_baseline): `reps` passes inside one OpenMP parallel region. */
int orc_bench_batch(const orc_model* m, const orc_params* p, int kind, int n, int stride, const double* q,
                    const double* v, const double* targets, const unsigned char* mask, const double* mu,
                    const double* mass_scale, double* tau, int* status, int nthreads, int reps);

#ifdef __cplusplus
}
#endif
#endif
