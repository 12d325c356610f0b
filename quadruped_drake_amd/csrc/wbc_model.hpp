// wbc_model.hpp -- flat 215-double model table (tools/compile_model.py) -> wbc::ModelC.
#pragma once
#include <math.h>
#include "wbc_tick.hpp"

namespace wbc {

enum { MODEL_FLAT = 215 };

// Returns 0 on success, <0 if a joint axis is not axis-aligned.
inline int model_from_flat(const double* f, ModelC* m) {
  int k = 0;
  m->base_mass = f[k++];
  double c[3];
  for (int i = 0; i < 3; i++) c[i] = f[k++];
  for (int i = 0; i < 3; i++) m->base_mc[i] = m->base_mass * c[i];
  for (int i = 0; i < 6; i++) m->base_I[i] = f[k++];
  for (int l = 0; l < 4; l++)
    for (int j = 0; j < 3; j++) {
      LinkC& L = m->link[l][j];
      for (int i = 0; i < 3; i++) L.off[i] = f[k++];
      double a[3];
      for (int i = 0; i < 3; i++) a[i] = f[k++];
      int ax = -1;
      for (int i = 0; i < 3; i++)
        if (fabs(fabs(a[i]) - 1.0) < 1e-12) ax = i;
      if (ax < 0) return -1;
      L.axis = ax;
      L.sgn = a[ax] > 0 ? 1.0 : -1.0;
      for (int i = 0; i < 3; i++) L.axv[i] = (i == ax) ? L.sgn : 0.0;
      L.mass = f[k++];
      for (int i = 0; i < 3; i++) L.mc[i] = L.mass * f[k++];
      for (int i = 0; i < 6; i++) L.I[i] = f[k++];
    }
  for (int l = 0; l < 4; l++)
    for (int i = 0; i < 3; i++) m->foot_off[l][i] = f[k++];
  m->gravity = f[k++];
  for (int i = 0; i < 12; i++) { m->q_perm[i] = i; m->act_perm[i] = i; m->act_inv[i] = i; }
  return 0;
}

// the kernels' kinematics are specialised to this joint-axis pattern (leg_fk_xyy): abduction about +-x, hip and knee about +-y
inline bool model_axes_are_xyy(const ModelC* m) {
  for (int l = 0; l < 4; l++)
    if (m->link[l][0].axis != 0 || m->link[l][1].axis != 1 || m->link[l][2].axis != 1) return false;
  return true;
}

inline void model_set_perms(ModelC* m, const int* q_perm, const int* act_perm) {
  for (int i = 0; i < 12; i++) {
    if (q_perm) m->q_perm[i] = q_perm[i];
    if (act_perm) m->act_perm[i] = act_perm[i];
  }
  for (int k = 0; k < 12; k++) m->act_inv[m->act_perm[k]] = k;
}

inline void params_default(int kind, ParamsC* p) {
  // inverse_dynamics_controller.py:117-127 / mptc_controller.py:143-153; mu :19 / :20; Kd :93 / :115
  if (kind == KIND_ID || kind == KIND_CLF) {  // CLF inherits IDController; its own gains are literals in the law
    p->Kp_body_p = 500.0; p->Kd_body_p = 50.0; p->Kp_body_rpy = 500.0; p->Kd_body_rpy = 50.0;
    p->Kp_foot = 100.0; p->Kd_foot = 20.0;
  } else {
    p->Kp_body_p = 100.0; p->Kd_body_p = 10.0; p->Kp_body_rpy = 100.0; p->Kd_body_rpy = 10.0;
    p->Kp_foot = 200.0; p->Kd_foot = 20.0;
  }
  p->w_body = 10.0; p->w_foot = 1.0; p->mu = 0.7; p->Kd_contact = 100.0;
  p->tau_max = INFINITY; p->eps2 = 1e-8;
}

}  // namespace wbc
