/*
 * wbc.h -- C ABI of the MI355X batched whole-body-QP controller (libwbc_hip.so).
 *
 * Drop-in boundary for ONE hot path of vincekurtz/quadruped_drake: the per-control-tick
 * whole-body QP of IDController / MPTCController.  One `wbc_step` call replaces N calls of
 *
 *     BasicController.DoSetControlTorques -> self.ControlLaw(context, q, v)
 *         controllers/basic_controller.py:286-320
 *         controllers/inverse_dynamics_controller.py:103-234   (kind = WBC_KIND_ID)
 *         controllers/mptc_controller.py:125-310               (kind = WBC_KIND_MPTC)
 *         controllers/pc_controller.py:44-255                  (kind = WBC_KIND_PC)
 *         controllers/clf_controller.py:48-234                 (kind = WBC_KIND_CLF)
 *
 * i.e. everything between the reference's `quad_state` / `trunk_input` input ports and its
 * `quad_torques` / `output_metrics` output ports (basic_controller.py:33-50,
 * inverse_dynamics_controller.py:14-16).  Plain pointers and sizes only; no torch types.
 *
 * Data layout: struct-of-arrays, batch index fastest.  Row r of robot i is at base[r*ld + i].
 *   q        [19][ld]  qw qx qy qz | x y z | 12 joint angles       (simulate.py:171-176)
 *   v        [18][ld]  w_WB (world) | v_WBo (world) | 12 joint rates (mptc_controller.py:190,194)
 *   targets  [54][ld]  body p, pd, pdd, rpy, rpyd, rpydd (18 rows), then for each foot
 *                      [LF RF LH RH]: p, pd, pdd (9 rows)          (planners/simple.py:45-85)
 *   contact_mask [n]   bit i set = foot i of [LF RF LH RH] in contact (`contact_states`)
 *   tau      [12][ld]  joint torques in ACTUATOR order             (basic_controller.py:37-40,320)
 *   metrics  [4][ld]   V, err, res, Vdot                            (basic_controller.py:47-50,283)
 *   status   [n]       0 optimal, 1 iteration cap, 2 singular / infeasible (tau and the accelerations are 0 then),
 *                      3 ill-conditioned (MPTC / PC only: |sin(knee)| < 1e-4 on some leg, where the law's own
 *                      inv(J M^-1 J') (mptc_controller.py:237-238) loses its digits; tau, metrics and accelerations ARE
 *                      written, but neither this library nor a dense restatement can vouch for them)
 *                      (the reference asserts result.is_success(): inverse_dynamics_controller.py:224)
 * Joint rows of q/v are mapped through model.q_perm (canonical joint j is read from joint row
 * q_perm[j]); torque row k is canonical joint act_perm[k]  (basic_controller.py:310-313).
 *
 * Malformed instances: the reference asserts on a failed solve and hands NaN on through numpy; a batch cannot.  An instance is
 * reported with status 2 -- zero torques, zero accelerations, zero metrics -- when the
 * tick cannot answer it with finite numbers:
 *   - a value that is not a finite number (NaN, +-inf) in anything the law READS: q, v, the body targets, the targets of the SWING
 *     feet (a contact foot's target rows are not read -- the reference indexes p_feet_nom[swing_feet],
 *     inverse_dynamics_controller.py:152-154 -- and may hold anything), mu, mass_scale;
 *   - mu or mass_scale not a positive finite number;
 *   - a quaternion without a direction (zero, or so small / large that |q|^2 under- or overflows; any other quaternion stands for
 *     its normalised self: Drake's RotationMatrix(quaternion) scales by 2/|q|^2, and so do the kernels);
 *   - finite inputs whose torques or metrics overflow on the way (targets of 1e200 ...).
 * No output of a tick is ever NaN or inf.  The other instances of the batch -- including the three that share its wavefront -- are
 * computed exactly as if it were not there (bit-identical; tests/test_robustness_gpu.py), and in a closed-loop rollout such an
 * instance is reported on every tick for which the condition holds (its state is integrated with zero accelerations meanwhile)
 * without touching its neighbours.  contact_mask: only bits 0..3 are read.
 *
 * Error convention: every function returns 0 on success, <0 on API misuse or a HIP error;
 * nothing throws.  wbc_last_error() returns a thread-local message.
 * Threading: a handle is not thread-safe; distinct handles are independent (one per GPU).
 * Current device: every call runs on its handle's device and leaves the calling thread's current HIP device as it found it
 * (hipGetDevice before == after), so handles on several GPUs can be driven from one thread and torch's current device is not moved.
 * wbc_step is asynchronous on the handle's stream; wbc_sync blocks.  No allocation in wbc_step.
 */
#ifndef WBC_H
#define WBC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WBC_KIND_ID 0
#define WBC_KIND_MPTC 1
#define WBC_KIND_PC 2 /* controllers/pc_controller.py:44-255: MPTC + passivity row Vdot <= 0 */
#define WBC_KIND_CLF 3 /* controllers/clf_controller.py:48-234: CLF-QP (13 reduced variables: z and the slack delta) */

#define WBC_MODEL_FLAT 215
#define WBC_NQ 19
#define WBC_NV 18
#define WBC_NTARGET 54
#define WBC_NU 12
#define WBC_NMETRIC 4
#define WBC_MAX_LD 8388608 /* largest leading dimension (and batch) of one call: 2^23 instances (7 GB of tick I/O); wbc_create / wbc_step reject more */

/* wbc_create flags */
#define WBC_DEVICE_PTRS 0u /* wbc_step receives device pointers (default) */
#define WBC_HOST_PTRS 1u   /* wbc_step receives host pointers; staged through handle-owned buffers (n <= 64: one pinned,
                              device-mapped block the kernel reads and writes directly, no copy calls).  The OUTPUT arrays of a
                              wbc_step must stay valid until the outputs have been delivered, and only the library delivers
                              them: wbc_sync, or the next wbc_step / wbc_set_stream on the handle (each first collects a
                              pending result).  Waiting on the stream or on an event of one's own does NOT: for n <= 64 the
                              copy into the caller's arrays is host code that runs inside those calls.  wbc_destroy waits
                              for the device and delivers nothing itself -- but what that means differs by batch size:
                              n <= 64: an uncollected result is ABANDONED (the caller's arrays are not written);
                              n > 64: the device-to-host copies into the caller's arrays were queued by wbc_step itself and
                              COMPLETE while wbc_destroy waits.  So: the output arrays of the last wbc_step must stay
                              allocated until wbc_sync or wbc_destroy has returned, whatever the batch size */

/* Kinematic tree + inertias, as produced by tools/compile_model.py from the reference's URDFs
 * (models/mini_cheetah/mini_cheetah_mesh.urdf, models/anymal_b_simple_description/urdf/anymal_drake.urdf):
 * flat = base{mass, com[3], I[6]}, 4 legs x 3 links {off[3], axis[3], mass, com[3], I[6]},
 * 4 x foot_off[3], gravity.  Inertias about the link origin, order xx yy zz xy xz yz. */
typedef struct {
  double flat[WBC_MODEL_FLAT];
  int32_t q_perm[12];
  int32_t act_perm[12];
} wbc_model;

/* Gains, weights and friction: literals of the two ControlLaw bodies
 * (inverse_dynamics_controller.py:19,93,117-127; mptc_controller.py:20,115,143-153).
 * tau_max = +inf reproduces the reference (it has no torque rows); a finite value adds the box |tau_j| <= tau_max;
 * eps2 is the weight of the 1/2*eps2*|[tau; f]|^2 tie-break. */
typedef struct {
  double Kp_body_p, Kd_body_p, Kp_body_rpy, Kd_body_rpy, Kp_foot, Kd_foot;
  double w_body, w_foot, mu, Kd_contact, tau_max, eps2;
} wbc_params;

typedef struct wbc_handle_s* wbc_handle;
typedef struct wbc_traj_s* wbc_traj; /* stored trunk trajectory, see the end of this header */

/* End-of-rollout statistics accumulated on the device by wbc_step (one small vector per GPU;
 * reduced across GPUs with ONE RCCL all-gather of wbc_stats_pack vectors and wbc_stats_reduce, below). */
typedef struct {
  double ticks;          /* instances stepped */
  double status_nonzero; /* instances with status != 0 */
  double iters_sum;      /* active-set iterations */
  double tau_abs_sum;    /* sum |tau| over all joints */
  double tau_abs_max;    /* max |tau| */
  double err_sum;        /* sum of metrics[1] */
  double mask_count[16]; /* instances per contact mask */
} wbc_stats;

const char* wbc_last_error(void);
int wbc_version(void);

/* Defaults of the reference for `kind`. */
int wbc_params_default(int kind, wbc_params* out);

/* Creates a controller for up to max_batch instances on HIP device `device`.
 * params may be NULL (reference defaults).  Owns all scratch buffers and one stream. */
int wbc_create(const wbc_model* model, int kind, const wbc_params* params, int max_batch, int device,
               uint32_t flags, wbc_handle* out);
int wbc_destroy(wbc_handle h);

/* Use an externally owned hipStream_t (e.g. torch's current stream) instead of the handle's. */
int wbc_set_stream(wbc_handle h, void* hip_stream);

/* One control tick for n <= max_batch instances.  mu / mass_scale (per-instance friction
 * coefficient and trunk mass/inertia scale; domain randomisation, no reference counterpart),
 * metrics and status may be NULL. */
int wbc_step(wbc_handle h, int n, int ld, const double* q, const double* v, const double* targets,
             const uint8_t* contact_mask, const double* mu, const double* mass_scale, double* tau,
             double* metrics, int32_t* status);

int wbc_sync(wbc_handle h);

/* Runs `steps` back-to-back wbc_step launches bracketed by HIP events on the handle's stream
 * and returns the average milliseconds per launch (device time).  Blocks -- unless ms_per_step is
 * NULL: then the launches and both events are only queued, and wbc_time_steps_result() reports the
 * time once the caller has waited for the stream anyway (e.g. through wbc_stats_get). */
int wbc_time_steps(wbc_handle h, int steps, int n, int ld, const double* q, const double* v,
                   const double* targets, const uint8_t* contact_mask, const double* mu,
                   const double* mass_scale, double* tau, double* metrics, int32_t* status,
                   float* ms_per_step);

int wbc_time_steps_result(wbc_handle h, float* ms_per_step);

/* The same `steps` launches with one HIP event between every two of them: ms_each[s] = device time of launch s
 * (for the median / p10 / p90 of SURVEY 8d's protocol; the events add ~1 us between launches, so the
 * average of wbc_time_steps stays the headline figure).  Blocks. */
int wbc_time_steps_each(wbc_handle h, int steps, int n, int ld, const double* q, const double* v,
                        const double* targets, const uint8_t* contact_mask, const double* mu,
                        const double* mass_scale, double* tau, double* metrics, int32_t* status,
                        float* ms_each);

/* Statistics since the last reset (blocks until the stream is idle). */
int wbc_stats_get(wbc_handle h, wbc_stats* out);
int wbc_stats_reset(wbc_handle h);
/* The one exchange between GPUs (north_star: "RCCL only to gather end-of-rollout statistics"), for a caller that brings its own
 * collective: wbc_stats_pack = wbc_stats_get as a flat vector of WBC_NSTAT doubles in the field order of the wbc_stats struct; it blocks like wbc_stats_get --
 * the send buffer of ONE ncclAllGather(sendbuf, recvbuf, WBC_NSTAT, ncclDouble, comm, stream) per rank; wbc_stats_reduce folds the
 * `world` gathered vectors (rank-major, world * WBC_NSTAT doubles, host memory) into one wbc_stats: every field is summed except
 * tau_abs_max, which is the maximum.  Pure host arithmetic, no handle, no GPU; world >= 1. */
#define WBC_NSTAT 22
int wbc_stats_pack(wbc_handle h, double* out22);
int wbc_stats_reduce(const double* gathered, int world, wbc_stats* out);

/* Closed-loop rollouts.  If set (device pointer, [18][ld], same ld as wbc_step),
 * every wbc_step also writes the generalized accelerations vd of its QP solution (rows in the
 * order of v: w_WB-dot, v_WBo-dot, 12 joint accelerations in the caller's joint order).  NULL disables. */
int wbc_set_vdot_output(wbc_handle h, double* vdot);

/* Semi-implicit Euler on the reference's state conventions, in place on device arrays:
 *   v+ = v + dt vd ;  p+ = p + dt v_lin+ ;  quat+ = exp(dt/2 w+) (x) quat (world-frame w), renormalised ;
 *   joints+ = joints + dt qd+.   A simplified stand-in for MultibodyPlant(time_step=dt) (simulate.py:38):
 * contacts are whatever the QP's contact rows imply, there is no collision solver.  Asynchronous. */
int wbc_integrate(wbc_handle h, int n, int ld, double dt, double* q, double* v, const double* vdot);

/* `steps` closed-loop ticks entirely on the device: targets/contact from `traj` at time[i]
 * (wbc_traj_lookup), wbc_step, wbc_integrate, time += dt.  q, v, time, targets, contact_mask, tau,
 * vdot are device buffers ([..][ld]); metrics/status may be NULL.  Statistics accumulate in the
 * handle (wbc_stats_get).  Asynchronous on the handle's stream.  On the 16-lane kernel the whole rollout is
 * one persistent launch (state kept on chip between ticks), bit-identical to the launch-per-stage loop;
 * on return q, v, time hold the final state and targets / contact_mask / tau / metrics / status / vdot the
 * last tick's values. */
int wbc_rollout(wbc_handle h, wbc_traj traj, int steps, double dt, int n, int ld, double* q, double* v,
                double* time, double* targets, uint8_t* contact_mask, const double* mu,
                const double* mass_scale, double* tau, double* metrics, int32_t* status, double* vdot);

/* Warm-started active set inside wbc_rollout (default off).  The reference builds and solves a fresh QP every tick
 * (inverse_dynamics_controller.py:200: a new MathematicalProgram).  A rollout keeps each robot on chip between its ticks, so the friction rows that
 * were active when a robot's previous tick ended are free to remember; with on != 0 the active-set method adds those rows first (those of them that
 * are violated: the method stays the same Goldfarb-Idnani iteration with another choice among the violated rows).  On closed loops that sit on their
 * friction limits this rebuilds the set in as many steps as it has rows instead of 10 - 13 (profiles/r06/warm_start.md).  The QP is strictly convex,
 * so the solution does not depend on that choice: outputs agree with the cold start to rounding (1e-9 relative over thousands of ticks), NOT bit for
 * bit -- which is why it is an option: with the default (off) wbc_rollout stays bit-identical to the launch-per-stage loop.  A rollout call starts
 * cold in either case (the memory does not outlive the launch); wbc_step is not affected. */
int wbc_set_warm_start(wbc_handle h, int on);

/* Kernel variant: 0 = auto (default) or 3 = 16 lanes (one DPP row) per robot -- the one product kernel family for
 * every law, batch size and option.  (1 = lane-per-robot and 2 = quad-per-robot were round-1 mappings that lost at
 * every batch size and are retired: the call rejects them.) */
int wbc_set_variant(wbc_handle h, int variant);
/* The variant a wbc_step of n instances would run (always 3). */
int wbc_variant_for(wbc_handle h, int n);

/* Kernel resource report for the variant of the most recent launch (or of max_batch before any):
 * registers, scratch bytes/lane, LDS bytes. */
int wbc_kernel_info(wbc_handle h, int* num_vgpr, int* scratch_bytes, int* lds_bytes, int* block_threads);
/* The same for the persistent closed-loop kernel wbc_rollout launches for this handle (its own register allocation: the tick
 * inlined into the step loop). */
int wbc_rollout_kernel_info(wbc_handle h, int* num_vgpr, int* scratch_bytes, int* lds_bytes, int* block_threads);

/* ------------------------------------------------------------------------------------------
 * Callers of the path: trunk-trajectory wire format and target lookup.
 *
 * wbc_trunk_state mirrors lcm_types/trunk_state_t.lcm:3-49; the wire format is the 549-byte
 * big-endian LCM encoding of lcm_types/trunklcm/trunk_state_t.py:50-121 (8-byte fingerprint, then
 * the fields in declaration order).  Feet are [lf rf lh rh].  Host-side, no GPU needed. */
#define WBC_TRUNK_STATE_BYTES 549
typedef struct {
  double timestamp;
  uint8_t finished;
  double base_p[3], base_pd[3], base_pdd[3], base_rpy[3], base_rpyd[3], base_rpydd[3];
  double foot_p[4][3], foot_pd[4][3], foot_pdd[4][3];
  uint8_t contact[4];
  double foot_f[4][3];
} wbc_trunk_state;

/* 0 on success; -1 short buffer, -3 fingerprint mismatch ("Decode error", trunk_state_t.py:87-88). */
int wbc_trunk_state_decode(const uint8_t* buf, size_t len, wbc_trunk_state* out);
/* planners/towr.py:111-148: message -> the 54 target rows + contact mask of wbc_step. */
int wbc_trunk_state_to_targets(const wbc_trunk_state* s, double* targets54, uint8_t* contact_mask);

/* A stored trunk trajectory on the device and the per-tick lookup of planners/towr.py:92-106:
 * t < wait_time -> the standing targets (planners/simple.py:39-85), otherwise the sample whose
 * timestamp is nearest to t - wait_time (first index on ties: numpy argmin).  `timestamps` must be
 * non-decreasing; `targets` is [K][54] (one row per sample), `masks` [K]. */
int wbc_traj_create(int device, int K, const double* timestamps, const double* targets, const uint8_t* masks,
                    const double* standing_targets54, uint8_t standing_mask, double wait_time, wbc_traj* out);
int wbc_traj_destroy(wbc_traj t);
/* time[n] (device): per-instance simulation time.  Writes targets[54][ld] and contact_mask[n]
 * (device) on `hip_stream` (NULL = default stream).  Asynchronous. */
int wbc_traj_lookup(wbc_traj t, void* hip_stream, int n, int ld, const double* time, double* targets,
                    uint8_t* contact_mask);

#ifdef __cplusplus
}
#endif
#endif
