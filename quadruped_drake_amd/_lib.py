"""ctypes binding of libwbc_hip.so (include/wbc.h).  No fallback: if the HIP library is missing or a
call fails, this raises -- the product has no CPU path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WBC_HIP_LIB", os.path.join(_HERE, "libwbc_hip.so"))  # override = A/B kernel builds

c_double_p = C.POINTER(C.c_double)
c_u8_p = C.POINTER(C.c_uint8)
c_i32_p = C.POINTER(C.c_int32)

KIND_ID, KIND_MPTC, KIND_PC, KIND_CLF = 0, 1, 2, 3
DEVICE_PTRS, HOST_PTRS = 0, 1

# every symbol include/wbc.h (the controller interface) and include/wbc_extras.h (frozen out-of-scope exports) declare
SYMBOLS = ["wbc_last_error", "wbc_version", "wbc_params_default", "wbc_create", "wbc_destroy", "wbc_set_stream",
           "wbc_step", "wbc_sync", "wbc_time_steps", "wbc_time_steps_result", "wbc_time_steps_each", "wbc_stats_get", "wbc_stats_reset", "wbc_stats_pack", "wbc_stats_reduce", "wbc_set_variant", "wbc_set_warm_start", "wbc_set_vdot_output", "wbc_integrate", "wbc_rollout",
           "wbc_kernel_info", "wbc_rollout_kernel_info", "wbc_variant_for", "wbc_trunk_state_decode", "wbc_trunk_state_to_targets", "wbc_traj_create",
           "wbc_traj_destroy", "wbc_traj_lookup", "wbc_robot_state_decode", "wbc_robot_state_encode",
           "wbc_robot_states_unpack", "wbc_robot_controls_pack", "wbc_pd_step"]


class WbcModel(C.Structure):
    _fields_ = [("flat", C.c_double * 215), ("q_perm", C.c_int32 * 12), ("act_perm", C.c_int32 * 12)]


class WbcParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in
                ("Kp_body_p", "Kd_body_p", "Kp_body_rpy", "Kd_body_rpy", "Kp_foot", "Kd_foot",
                 "w_body", "w_foot", "mu", "Kd_contact", "tau_max", "eps2")]


class WbcStats(C.Structure):
    _fields_ = [("ticks", C.c_double), ("status_nonzero", C.c_double), ("iters_sum", C.c_double),
                ("tau_abs_sum", C.c_double), ("tau_abs_max", C.c_double), ("err_sum", C.c_double),
                ("mask_count", C.c_double * 16)]


class WbcTrunkState(C.Structure):
    _fields_ = [("timestamp", C.c_double), ("finished", C.c_uint8),
                ("base_p", C.c_double * 3), ("base_pd", C.c_double * 3), ("base_pdd", C.c_double * 3),
                ("base_rpy", C.c_double * 3), ("base_rpyd", C.c_double * 3), ("base_rpydd", C.c_double * 3),
                ("foot_p", (C.c_double * 3) * 4), ("foot_pd", (C.c_double * 3) * 4), ("foot_pdd", (C.c_double * 3) * 4),
                ("contact", C.c_uint8 * 4), ("foot_f", (C.c_double * 3) * 4)]


class WbcRobotState(C.Structure):
    _fields_ = [("q", C.c_float * 19), ("v", C.c_float * 18), ("tau", C.c_float * 12)]


class WbcError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WbcError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)" % LIB_PATH)
        # ONE HIP runtime per process: libwbc_hip.so is linked against libamdhip64.so by soname and PyTorch-ROCm ships its own copy.
        # Loaded after torch, the library binds to the copy torch already mapped; loaded BEFORE it (build() then smoke() in one
        # process) the system copy comes in first, torch then maps its own, and the second runtime finds no device.  Device memory
        # and streams are torch's here anyway, so torch goes first whenever it is installed.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        l.wbc_last_error.restype = C.c_char_p
        l.wbc_version.restype = C.c_int
        l.wbc_params_default.argtypes = [C.c_int, C.POINTER(WbcParams)]
        l.wbc_create.argtypes = [C.POINTER(WbcModel), C.c_int, C.POINTER(WbcParams), C.c_int, C.c_int, C.c_uint32,
                                 C.POINTER(C.c_void_p)]
        l.wbc_destroy.argtypes = [C.c_void_p]
        l.wbc_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        step_args = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 9
        l.wbc_step.argtypes = step_args
        l.wbc_sync.argtypes = [C.c_void_p]
        l.wbc_time_steps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 9 + [C.POINTER(C.c_float)]
        l.wbc_time_steps_result.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        l.wbc_time_steps_each.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 9 + [C.POINTER(C.c_float)]
        l.wbc_stats_get.argtypes = [C.c_void_p, C.POINTER(WbcStats)]
        l.wbc_stats_reset.argtypes = [C.c_void_p]
        ab_build = "WBC_HIP_LIB" in os.environ and not hasattr(l, "wbc_stats_pack")   # an older kernel build under A/B timing (tools/qt.py --lib)
        if not ab_build:
            l.wbc_stats_pack.argtypes = [C.c_void_p, c_double_p]
            l.wbc_stats_reduce.argtypes = [c_double_p, C.c_int, C.POINTER(WbcStats)]
        l.wbc_set_variant.argtypes = [C.c_void_p, C.c_int]
        l.wbc_set_warm_start.argtypes = [C.c_void_p, C.c_int]
        l.wbc_variant_for.argtypes = [C.c_void_p, C.c_int]
        l.wbc_set_vdot_output.argtypes = [C.c_void_p, C.c_void_p]
        l.wbc_integrate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        l.wbc_rollout.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int] + [C.c_void_p] * 11
        l.wbc_kernel_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
        l.wbc_trunk_state_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(WbcTrunkState)]
        l.wbc_trunk_state_to_targets.argtypes = [C.POINTER(WbcTrunkState), c_double_p, c_u8_p]
        l.wbc_traj_create.argtypes = [C.c_int, C.c_int, c_double_p, c_double_p, c_u8_p, c_double_p, C.c_uint8, C.c_double,
                                      C.POINTER(C.c_void_p)]
        l.wbc_traj_destroy.argtypes = [C.c_void_p]
        l.wbc_traj_lookup.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        l.wbc_robot_state_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(WbcRobotState)]
        l.wbc_robot_state_encode.argtypes = [C.POINTER(WbcRobotState), C.c_char_p, C.c_size_t]
        l.wbc_robot_states_unpack.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        l.wbc_robot_controls_pack.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                              C.c_void_p]
        l.wbc_pd_step.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, c_double_p, C.c_double, C.c_double,
                                  C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]
        for s in SYMBOLS:
            if not (ab_build and s in ("wbc_stats_pack", "wbc_stats_reduce")):
                getattr(l, s)
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise WbcError("libwbc_hip: rc=%d: %s" % (rc, lib().wbc_last_error().decode()))
