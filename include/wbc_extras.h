/*
 * wbc_extras.h -- exports of libwbc_hip.so that are NOT part of the controller interface of include/wbc.h:
 * callers and data formats either side of the hot path that earlier rounds widened into (the robot-side wire format of the
 * reference's use_lcm loop, its joint-space PD law).  Kept working and tested, frozen since round 3; a maintainer who binds the
 * whole-body-QP path needs include/wbc.h only.
 */
#ifndef WBC_EXTRAS_H
#define WBC_EXTRAS_H

#include "wbc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * The robot-side wire format of the reference's use_lcm path (controllers/basic_controller.py:52-61,79-87,289-314):
 * `robot_state_control_lcmt` (lcm_types/robot_state_control_lcmt.lcm: float q[19], v[18], tau[12]) -- 204 bytes,
 * 8-byte fingerprint rotl1(0xbe14089c923ad667) then 49 big-endian IEEE floats
 * (lcm_types/cheetahlcm/robot_state_control_lcmt.py:28-79).  States arrive on "robot_current_state" in the plant's
 * own joint order (exactly the rows wbc_step takes with q_perm); torques leave on "robot_control_input" as
 * (S'u)[-12:] -- the actuator-order torques re-indexed to the plant's joint order (basic_controller.py:310-313),
 * q and v of that message left zero. */
#define WBC_ROBOT_STATE_BYTES 204
typedef struct {
  float q[19], v[18], tau[12];
} wbc_robot_state;
/* host, one message: 0 on success; -1 short buffer / null; -3 fingerprint mismatch ("Decode error", :51-52) */
int wbc_robot_state_decode(const uint8_t* buf, size_t len, wbc_robot_state* out);
/* host, one message: bytes written (204) or -1 */
int wbc_robot_state_encode(const wbc_robot_state* in, uint8_t* buf, size_t cap);
/* device, batched: n messages back to back in `msgs` (device, 4-byte aligned) -> q[19][ld], v[18][ld] (float -> double
 * is exact); ok[i] (nullable) = 1 decoded / 0 fingerprint mismatch (that robot's columns are left untouched).
 * Asynchronous on `hip_stream`. */
int wbc_robot_states_unpack(int device, void* hip_stream, int n, int ld, const uint8_t* msgs, double* q, double* v,
                            uint8_t* ok);
/* device, batched: tau[12][ld] in ACTUATOR order (what wbc_step wrote) -> n "robot_control_input" messages.
 * q_perm / act_perm: host int[12], the handle's model permutations (NULL = identity):
 * message.tau[q_perm[act_perm[k]]] = (float) tau[k].  Asynchronous on `hip_stream`. */
int wbc_robot_controls_pack(int device, void* hip_stream, int n, int ld, const double* tau, const int* q_perm,
                            const int* act_perm, uint8_t* msgs);

/* ------------------------------------------------------------------------------------------
 * The reference's joint-space PD law (control method "B": BasicController.ControlLaw, controllers/basic_controller.py:322-352),
 * batched and stateless: tau_v = -Kp N+(q)(q - q_nom) - Kd v, u = clip(S tau_v, -u_max, u_max).  S selects the twelve joint rows,
 * where N+ is the identity, so u[k] = clip(-(kp (q_j - q_nom_j)) - kd v_j) with j = the plant index of the joint actuator k drives
 * (j = q_perm[act_perm[k]]); evaluated without fused multiply-add, i.e. bit for bit what the reference's numpy computes.
 * q[19][ld], v[18][ld], tau[12][ld] (actuator order) are device pointers; q_nom19 is a HOST array in the plant's own joint order
 * (NULL = the reference's literal: base at (0, 0, 0.3), every leg (0, -0.8, 1.6)); the reference's gains are kp 30, kd 1.5, u_max 150.
 * q_perm / act_perm: host int[12], NULL = identity.  Asynchronous on `hip_stream`. */
int wbc_pd_step(int device, void* hip_stream, int n, int ld, const double* q, const double* v, const double* q_nom19,
                double kp, double kd, double u_max, const int* q_perm, const int* act_perm, double* tau);

#ifdef __cplusplus
}
#endif
#endif
