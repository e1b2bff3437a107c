"""ctypes binding of liblmpc_hip.so -- the same entry points a Julia `ccall` would bind
(include/lmpc_hip.h; Julia stub in INTEGRATION.md).

There is no CPU fallback anywhere in this package: if the shared library is missing the import
fails loudly, and if no HIP device is present every setup/solve call raises `LmpcError`.
"""
from __future__ import annotations

import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# LMPC_HIP_LIB: another build of the same library (A/B of compiler flags); default = the in-tree build
LIB_PATH = os.environ.get("LMPC_HIP_LIB") or os.path.join(_PKG, "lib", "liblmpc_hip.so")

LMPC_OK = 1
ERR_NAMES = {-1: "INFEASIBLE", -5: "NONCONVEX", -6: "OVERDETERMINED", -100: "BADARG",
             -101: "NOGPU", -102: "HIP", -103: "UNSUPPORTED"}

# every symbol include/lmpc_hip.h declares (tests check the library exports all of them)
SYMBOLS = (
    "lmpc_abi_version", "lmpc_default_settings", "lmpc_setup", "lmpc_setup_ldp", "lmpc_transform",
    "lmpc_get_ldp", "lmpc_get_dims", "lmpc_active_words", "lmpc_set_settings",
    "lmpc_solve_batch", "lmpc_solve_batch_device", "lmpc_solve_batches_device", "lmpc_solve_one",
    "lmpc_default_settings_f32", "lmpc_solve_batch_f32", "lmpc_solve_batch_f32_device", "lmpc_simulate",
    "lmpc_simulate_device", "lmpc_simulate_f32", "lmpc_simulate_f32_device", "lmpc_form_parameter_device", "lmpc_simulate_ref_device", "lmpc_kernel_name",
    "lmpc_set_parameter_layout", "lmpc_compute_control", "lmpc_compute_control_device",
    "lmpc_set_observer", "lmpc_predict_state", "lmpc_correct_state", "lmpc_predict_state_device",
    "lmpc_correct_state_device", "lmpc_compute_control_observer_device",
    "lmpc_profile", "lmpc_profile_read", "lmpc_set_option", "lmpc_free", "lmpc_last_error",
    "lmpc_setup_multi", "lmpc_multi_devices", "lmpc_multi_handle", "lmpc_multi_partition",
    "lmpc_solve_batch_multi", "lmpc_solve_batch_multi_device", "lmpc_multi_last_error", "lmpc_free_multi",
    "lmpc_pin_host", "lmpc_unpin_host", "lmpc_release_scratch", "lmpc_check",
    "lmpc_distinct_active_sets_device", "lmpc_distinct_active_sets_overflowed", "lmpc_wave_stats",
    "lmpc_setup_ex", "lmpc_is_avi", "lmpc_get_avi", "lmpc_transform_avi", "lmpc_multi_set_option", "lmpc_discover_regions_device", "lmpc_reserve", "lmpc_get_prox",
)


class LmpcError(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        super().__init__(f"lmpc error {code} ({ERR_NAMES.get(code, '?')}): {msg}")


class Settings(ctypes.Structure):
    """`lmpc_settings`; defaults are DAQP's as documented in reference docs/src/manual/solver.md:49-56."""
    _fields_ = [("primal_tol", ctypes.c_double), ("dual_tol", ctypes.c_double),
                ("zero_tol", ctypes.c_double), ("progress_tol", ctypes.c_double),
                ("fval_bound", ctypes.c_double), ("rho_soft", ctypes.c_double),
                ("cycle_tol", ctypes.c_int32), ("iter_limit", ctypes.c_int32),
                ("eps_prox", ctypes.c_double), ("eta_prox", ctypes.c_double)]


class Block(ctypes.Structure):
    """`lmpc_block`: one block of theta cut from a (w x T, column-major) trajectory on the device."""
    _fields_ = [("src", ctypes.c_void_p), ("stride", ctypes.c_int64), ("w", ctypes.c_int32),
                ("T", ctypes.c_int32), ("k0", ctypes.c_int32), ("H", ctypes.c_int32)]


class ParamLayout(ctypes.Structure):
    """`lmpc_param_layout`: the N_STATE ... N_AFFINE_PARAMETER constants of the generated header
    (reference codegen.jl:154-165) and, for reference condensation, N_PREVIEW_HORIZON + traj2setpoint."""
    _fields_ = [("n_state", ctypes.c_int32), ("n_reference", ctypes.c_int32), ("n_disturbance", ctypes.c_int32),
                ("n_control_prev", ctypes.c_int32), ("n_affine_parameter", ctypes.c_int32),
                ("n_preview_horizon", ctypes.c_int32), ("traj2setpoint", ctypes.c_void_p)]


class Observer(ctypes.Structure):
    """`lmpc_observer`: the generated observer's arrays (reference src/observer.jl:124-141)."""
    _fields_ = [("n_state", ctypes.c_int32), ("n_control", ctypes.c_int32), ("n_disturbance", ctypes.c_int32),
                ("n_measurement", ctypes.c_int32), ("plant_dynamics", ctypes.c_void_p),
                ("measurement_function", ctypes.c_void_p), ("k_transpose", ctypes.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C linearmpc.jl_amd/csrc)")
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    L.lmpc_abi_version.restype = i32
    L.lmpc_default_settings.argtypes = [ctypes.POINTER(Settings)]
    L.lmpc_default_settings.restype = None
    L.lmpc_setup.argtypes = [ctypes.POINTER(vp)] + [i32] * 5 + [vp] * 9 + [i32, vp, i32]
    L.lmpc_setup.restype = i32
    L.lmpc_setup_ex.argtypes = [ctypes.POINTER(vp)] + [i32] * 5 + [vp] * 9 + [i32, vp, vp, i32, i32, i32]
    L.lmpc_setup_ex.restype = i32
    L.lmpc_transform_avi.argtypes = [i32] * 5 + [vp] * 9 + [i32] + [vp] * 9
    L.lmpc_transform_avi.restype = i32
    L.lmpc_is_avi.argtypes = [vp]
    L.lmpc_is_avi.restype = i32
    L.lmpc_get_avi.argtypes = [vp, vp, vp]
    L.lmpc_get_avi.restype = i32
    L.lmpc_setup_ldp.argtypes = [ctypes.POINTER(vp)] + [i32] * 5 + [vp] * 9 + [i32]
    L.lmpc_setup_ldp.restype = i32
    L.lmpc_transform.argtypes = [i32] * 5 + [vp] * 9 + [i32] + [vp] * 7
    L.lmpc_transform.restype = i32
    L.lmpc_get_ldp.argtypes = [vp] * 9
    L.lmpc_get_ldp.restype = i32
    L.lmpc_get_dims.argtypes = [vp, ctypes.POINTER(ctypes.c_int32 * 6)]
    L.lmpc_get_dims.restype = i32
    L.lmpc_active_words.argtypes = [vp]
    L.lmpc_active_words.restype = i32
    L.lmpc_set_settings.argtypes = [vp, ctypes.POINTER(Settings)]
    L.lmpc_set_settings.restype = i32
    L.lmpc_solve_batch.argtypes = [vp, i64] + [vp] * 6
    L.lmpc_solve_batch.restype = i32
    L.lmpc_solve_batch_device.argtypes = [vp, i64] + [vp] * 7
    L.lmpc_solve_batch_device.restype = i32
    if hasattr(L, "lmpc_solve_batches_device"):
        L.lmpc_solve_batches_device.argtypes = [vp, i32, i64, vp, vp, vp, vp]
        L.lmpc_solve_batches_device.restype = i32
    L.lmpc_default_settings_f32.argtypes = [ctypes.POINTER(Settings)]
    L.lmpc_default_settings_f32.restype = None
    L.lmpc_solve_batch_f32.argtypes = [vp, i64] + [vp] * 6
    L.lmpc_solve_batch_f32.restype = i32
    L.lmpc_solve_batch_f32_device.argtypes = [vp, i64] + [vp] * 7
    L.lmpc_solve_batch_f32_device.restype = i32
    L.lmpc_solve_one.argtypes = [vp, vp, vp]
    L.lmpc_solve_one.restype = i32
    L.lmpc_simulate.argtypes = [vp, i64] + [i32] * 4 + [vp] * 8 + [i32]
    L.lmpc_simulate.restype = i32
    L.lmpc_simulate_device.argtypes = [vp, i64] + [i32] * 4 + [vp] * 8 + [i32, vp]
    L.lmpc_simulate_device.restype = i32
    L.lmpc_simulate_f32.argtypes = [vp, i64] + [i32] * 4 + [vp] * 8 + [i32]
    L.lmpc_simulate_f32.restype = i32
    L.lmpc_simulate_f32_device.argtypes = [vp, i64] + [i32] * 4 + [vp] * 8 + [i32, vp]
    L.lmpc_simulate_f32_device.restype = i32
    bp = ctypes.POINTER(Block)
    L.lmpc_form_parameter_device.argtypes = [vp, i64, vp, vp, i32, bp, bp, vp, i32, bp, vp]
    L.lmpc_form_parameter_device.restype = i32
    L.lmpc_simulate_ref_device.argtypes = [vp, i64, i32, i32, bp, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp]
    L.lmpc_simulate_ref_device.restype = i32
    L.lmpc_set_parameter_layout.argtypes = [vp, ctypes.POINTER(ParamLayout)]
    L.lmpc_set_parameter_layout.restype = i32
    L.lmpc_compute_control.argtypes = [vp, i64] + [vp] * 6 + [i32]
    L.lmpc_compute_control.restype = i32
    L.lmpc_compute_control_device.argtypes = [vp, i64] + [vp] * 6 + [i32, vp]
    L.lmpc_compute_control_device.restype = i32
    L.lmpc_set_observer.argtypes = [vp, ctypes.POINTER(Observer)]
    L.lmpc_set_observer.restype = i32
    for fn in (L.lmpc_predict_state, L.lmpc_correct_state):
        fn.argtypes = [vp, i64, vp, vp, vp]
        fn.restype = i32
    for fn in (L.lmpc_predict_state_device, L.lmpc_correct_state_device):
        fn.argtypes = [vp, i64, vp, vp, vp, vp]
        fn.restype = i32
    L.lmpc_compute_control_observer_device.argtypes = [vp, i64, vp, vp, i32, vp, vp, vp, vp, i32, vp]
    L.lmpc_compute_control_observer_device.restype = i32
    L.lmpc_kernel_name.argtypes = [vp]
    L.lmpc_kernel_name.restype = ctypes.c_char_p
    L.lmpc_profile.argtypes = [vp, i32]
    L.lmpc_profile.restype = i32
    L.lmpc_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_double * 3)]
    L.lmpc_profile_read.restype = i32
    L.lmpc_set_option.argtypes = [vp, ctypes.c_char_p, i32]
    L.lmpc_set_option.restype = i32
    L.lmpc_free.argtypes = [vp]
    L.lmpc_free.restype = None
    L.lmpc_setup_multi.argtypes = [ctypes.POINTER(vp)] + [i32] * 5 + [vp] * 9 + [i32, vp, vp, i32]
    L.lmpc_setup_multi.restype = i32
    L.lmpc_multi_devices.argtypes = [vp]
    L.lmpc_multi_devices.restype = i32
    L.lmpc_multi_handle.argtypes = [vp, i32]
    L.lmpc_multi_handle.restype = vp
    L.lmpc_multi_partition.argtypes = [i64, i32, ctypes.POINTER(i64)]
    L.lmpc_multi_partition.restype = None
    L.lmpc_solve_batch_multi.argtypes = [vp, i64] + [vp] * 6
    L.lmpc_solve_batch_multi.restype = i32
    L.lmpc_solve_batch_multi_device.argtypes = [vp] + [vp] * 6
    L.lmpc_solve_batch_multi_device.restype = i32
    L.lmpc_multi_last_error.argtypes = [vp]
    L.lmpc_multi_last_error.restype = ctypes.c_char_p
    L.lmpc_free_multi.argtypes = [vp]
    L.lmpc_free_multi.restype = None
    L.lmpc_release_scratch.argtypes = [vp]
    L.lmpc_release_scratch.restype = i32
    if hasattr(L, "lmpc_distinct_active_sets_device"):
        L.lmpc_distinct_active_sets_device.argtypes = [vp, ctypes.c_int64, vp, vp, i32, vp, vp, vp, vp, vp]
        L.lmpc_distinct_active_sets_device.restype = i32
        L.lmpc_distinct_active_sets_overflowed.argtypes = [vp, vp]
        L.lmpc_distinct_active_sets_overflowed.restype = i32
    if hasattr(L, "lmpc_wave_stats"):
        L.lmpc_wave_stats.argtypes = [vp, vp]
        L.lmpc_wave_stats.restype = i32
    if hasattr(L, "lmpc_check"):              # (an older build selected with LMPC_HIP_LIB for an A/B lacks it)
        L.lmpc_check.argtypes = [vp]
        L.lmpc_check.restype = i32
    L.lmpc_pin_host.argtypes = [vp, ctypes.c_size_t]
    L.lmpc_pin_host.restype = i32
    L.lmpc_unpin_host.argtypes = [vp]
    L.lmpc_unpin_host.restype = i32
    L.lmpc_last_error.argtypes = [vp]
    L.lmpc_last_error.restype = ctypes.c_char_p
    _lib = L
    return L


def default_settings() -> Settings:
    s = Settings()
    lib().lmpc_default_settings(ctypes.byref(s))
    return s


def default_settings_f32() -> Settings:
    s = Settings()
    lib().lmpc_default_settings_f32(ctypes.byref(s))
    return s


def last_error(handle=None) -> str:
    return (lib().lmpc_last_error(handle) or b"").decode()


def check(rc, handle=None):
    if rc != LMPC_OK:
        raise LmpcError(rc, last_error(handle))
