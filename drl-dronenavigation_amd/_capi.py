"""ctypes binding of libdronenav.so (include/dronenav.h).  No torch types cross this boundary.

There is no CPU fallback: if the library is missing the import of the product fails loudly
(`DroneNavLibraryError`) and tells the user how to build it.
"""
import ctypes as C
import os

from . import build as _build

MAX_WAYPOINTS = 64
OBS_DIM = 13
ACT_DIM = 4
ABI_VERSION = 9
GROUND_CONTACT_AUTO = 2

DN_OK = 0
STATUS_NAMES = {0: "DN_OK", -1: "DN_ERR_INVALID_ARGUMENT", -2: "DN_ERR_HIP", -3: "DN_ERR_OUT_OF_MEMORY",
                -4: "DN_ERR_NO_DEVICE", -5: "DN_ERR_BAD_STATE"}


class DroneNavError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class DroneNavLibraryError(ImportError):
    pass


class DnConfig(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int64), ("device_id", C.c_int32), ("num_waypoints", C.c_int32),
        ("waypoints", C.c_double * (MAX_WAYPOINTS * 3)), ("spawn", C.c_double * 3), ("aviary_dim", C.c_double * 6),
        ("threshold", C.c_double), ("max_steps", C.c_int32), ("circle", C.c_int32), ("cylinder", C.c_int32),
        ("include_distance", C.c_int32), ("normalize_actions", C.c_int32), ("normalize_obs", C.c_int32),
        ("ground_contact", C.c_int32), ("compute_f32", C.c_int32), ("act_noise_sigma", C.c_float),
        ("obs_noise_sigma", C.c_float), ("seed", C.c_uint64), ("env_id_offset", C.c_int64),
        ("clip_rew", C.c_int32), ("norm_rew", C.c_int32), ("physics", C.c_int32), ("action_type", C.c_int32),
        ("random_spawn", C.c_int32), ("zero_damping", C.c_int32),
    ]


class DnEnvState(C.Structure):
    _fields_ = [
        ("pos", C.c_float * 3), ("quat", C.c_float * 4), ("vel", C.c_float * 3), ("ang_v", C.c_float * 3),
        ("prev_vel", C.c_float * 3), ("prev_ang_v", C.c_float * 3), ("cur_pos", C.c_float * 3),
        ("d", C.c_float), ("d_prev", C.c_float), ("idx", C.c_int32), ("steps", C.c_int32), ("just_found", C.c_int32),
        ("ep_ret", C.c_float), ("ep_len", C.c_int32),
        ("rms_mean", C.c_double * OBS_DIM), ("rms_var", C.c_double * OBS_DIM), ("rms_count", C.c_double),
        ("rr_returns", C.c_double), ("rr_mean", C.c_double), ("rr_var", C.c_double), ("rr_count", C.c_double),
        ("last_rpm", C.c_float * 4), ("pid", C.c_double * 9), ("ep_ret_lo", C.c_float),
    ]


class DnMlpNet(C.Structure):
    _fields_ = [("w1", C.c_void_p), ("w2", C.c_void_p), ("w3", C.c_void_p), ("wh", C.c_void_p),
                ("b1", C.c_void_p), ("b2", C.c_void_p), ("b3", C.c_void_p), ("bh", C.c_void_p),
                ("out", C.c_void_p), ("out_dim", C.c_int32), ("grade", C.c_int32), ("arch", C.c_int32),
                ("reserved_", C.c_int32)]


class DnStats(C.Structure):
    _fields_ = [("env_steps", C.c_int64), ("episodes", C.c_int64), ("truncated", C.c_int64), ("completed", C.c_int64),
                ("sum_ep_len", C.c_int64), ("sum_found_targets", C.c_int64), ("sum_ep_return", C.c_double)]


# every entry point declared in include/dronenav.h: name -> (restype, argtypes)
_VP, _I32, _I64 = C.c_void_p, C.c_int32, C.c_int64
PROTOTYPES = {
    "dn_abi_version": (_I32, []),
    "dn_last_error": (C.c_char_p, []),
    "dn_device_count": (_I32, []),
    "dn_config_default": (None, [C.POINTER(DnConfig)]),
    "dn_create": (_I32, [C.POINTER(DnConfig), C.POINTER(_VP)]),
    "dn_destroy": (_I32, [_VP]),
    "dn_num_envs": (_I64, [_VP]),
    "dn_get_config": (_I32, [_VP, C.POINTER(DnConfig)]),
    "dn_get_num_cus": (_I32, [_VP]),
    "dn_get_exact_flags": (_I32, [_VP]),
    "dn_resolve_ground_contact": (_I32, [C.POINTER(DnConfig)]),
    "dn_reset": (_I32, [_VP, _VP, _VP]),
    "dn_step": (_I32, [_VP] * 12),
    "dn_step_many": (_I32, [_VP, _I64] + [_VP] * 11),
    "dn_eval_kinematics": (_I32, [_VP] * 11),
    "dn_compact_done": (_I32, [_VP, _I64, _VP, _VP, _I32, _VP]),
    "dn_pack_done": (_I32, [_VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I32, _VP]),
    "dn_stream_copy": (_I32, [_VP, _VP, _I64, _I32, _VP]),
    "dn_get_state": (_I32, [_VP, _VP, _I64]),
    "dn_set_state": (_I32, [_VP, _VP, _I64]),
    "dn_get_stats": (_I32, [_VP, C.POINTER(DnStats), _VP]),
    "dn_reset_stats": (_I32, [_VP, _VP]),
    "dn_get_kernel_waves": (_I32, [_VP, _I32]),
    "dn_get_step_count": (_I32, [_VP, C.POINTER(C.c_uint64)]),
    "dn_set_step_count": (_I32, [_VP, C.c_uint64]),
    "dn_preprocess_action": (_I32, [_VP, _I64, _I32, _VP, _VP, _VP, _I32, _VP]),
    "dn_policy_sample": (_I32, [_VP, _VP, C.POINTER(C.c_float), C.c_uint64, _I32, _VP, _VP, _VP, _VP]),
    "dn_squashed_sample": (_I32, [_VP, _VP, C.c_uint64, _I32, _VP, _VP, _VP]),
    "dn_step_squashed": (_I32, [_VP, _VP, C.c_uint64, _I32] + [_VP] * 12),
    "dn_add_bootstrap": (_I32, [_VP, _VP, _VP, C.c_double, _I64, _I32, _VP]),
    "dn_step_sampled": (_I32, [_VP, _VP, C.POINTER(C.c_float), C.c_uint64, _I32] + [_VP] * 12),
    "dn_mlp_forward": (_I32, [_VP, _I32, _VP, _VP, _I64, _I32, _I32, _VP]),
    "dn_mlp_step_sampled": (_I32, [_VP, _VP, _I32, _VP, _I32, C.POINTER(C.c_float), C.c_uint64, _I32] + [_VP] * 12),
    "dn_gae": (_I32, [_VP] * 5 + [_I64, _I64, C.c_double, C.c_double, _VP, _VP, _I32, _VP]),
    "dn_set_launch_events": (_I32, [_VP, _VP, _VP]),
    "dn_state_bytes": (_I64, [_I64, _I32]),
}

_lib = None


def _truthy(name):
    """An on / off environment switch, spelt as dn_create accepts it (anything else is left for dn_create to refuse)."""
    return os.environ.get(name, "").strip().lower() in ("1", "true", "on", "yes")


def library_path():
    # DN_LIB_PATH: an alternative build of the same ABI (A/B measurements of kernel variants).  DN_EXACT_NORM=1: the build with the
    # normaliser's float64 output stage (libdronenav_exact.so, include/dronenav.h DN_EXACT_FLAG_NORM).  Default = the in-tree library.
    if os.environ.get("DN_LIB_PATH"):
        return os.environ["DN_LIB_PATH"]
    return _build.LIB_PATH_EXACT if _truthy("DN_EXACT_NORM") else _build.LIB_PATH


def load():
    """Load libdronenav.so (built in-tree by build.build_library / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise DroneNavLibraryError(
            f"{path} is missing: run `python __graft_entry__.py` (or drl-dronenavigation_amd/build.py) to compile "
            "the HIP kernels with hipcc.  This package has no CPU fallback.")
    # PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's).  The device pointers
    # handed across the C ABI come from torch's allocator, so both must share ONE HIP runtime instance:
    # import torch first, and the dynamic loader resolves our NEEDED libamdhip64.so.7 to the copy torch
    # already mapped.  (Pure-C users link against the system runtime as usual.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        lib = C.CDLL(path)
    except OSError as exc:
        raise DroneNavLibraryError(f"cannot load {path}: {exc}") from exc
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise DroneNavLibraryError(f"{path} does not export {name}") from exc
        fn.restype = res
        fn.argtypes = args
    if lib.dn_abi_version() != ABI_VERSION:
        raise DroneNavLibraryError(f"{path}: ABI {lib.dn_abi_version()} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(status):
    if status != DN_OK:
        raise DroneNavError(status, load().dn_last_error().decode("utf-8", "replace"))
