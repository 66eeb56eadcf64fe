"""ctypes binding of the CPU oracle (oracle/dn_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, tests/golden/gen_golden.py,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product package never
imports this module (tests/test_layout.py enforces that).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DN_ORACLE_LIB=liboracle_asan.so: the AddressSanitizer / UBSan build of the same restatement (tests/test_oracle_asan.py runs
# golden-fixture replays under it in a child process with libasan preloaded)
_LIB_NAME = os.environ.get("DN_ORACLE_LIB", "liboracle.so")
_LIB_PATH = os.path.join(_HERE, _LIB_NAME)
MAX_WAYPOINTS = 64
OBS_DIM = 13


class OrcConfig(C.Structure):
    _fields_ = [
        ("num_waypoints", C.c_int32),
        ("waypoints", C.c_double * (MAX_WAYPOINTS * 3)),
        ("spawn", C.c_double * 3),
        ("dim", C.c_double * 6),
        ("threshold", C.c_double),
        ("max_steps", C.c_int32),
        ("circle", C.c_int32),
        ("cylinder", C.c_int32),
        ("include_distance", C.c_int32),
        ("normalize_actions", C.c_int32),
        ("normalize_obs", C.c_int32),
        ("ground_contact", C.c_int32),
        ("f32_state", C.c_int32),
        ("act_noise_sigma", C.c_float),
        ("obs_noise_sigma", C.c_float),
        ("seed", C.c_uint64),
        ("env_id_offset", C.c_int64),
        ("clip_rew", C.c_int32),
        ("norm_rew", C.c_int32),
        ("physics", C.c_int32),
        ("action_type", C.c_int32),
        ("random_spawn", C.c_int32),
        ("zero_damping", C.c_int32),
    ]


class OrcEnv(C.Structure):
    _fields_ = [
        ("pos", C.c_double * 3), ("quat", C.c_double * 4), ("vel", C.c_double * 3), ("ang_v", C.c_double * 3),
        ("rpy", C.c_double * 3),
        ("cur_pos", C.c_double * 3),
        ("cur_vel", C.c_double * 3), ("cur_ang_v", C.c_double * 3),
        ("prev_vel", C.c_double * 3), ("prev_ang_v", C.c_double * 3),
        ("d", C.c_double), ("d_prev", C.c_double),
        ("idx", C.c_int32), ("just_found", C.c_int32), ("is_done", C.c_int32), ("steps", C.c_int32),
        ("ep_ret", C.c_double), ("ep_len", C.c_int32),
        ("rms_mean", C.c_double * OBS_DIM), ("rms_var", C.c_double * OBS_DIM), ("rms_count", C.c_double),
        ("step_count", C.c_uint64),
        ("rr_returns", C.c_double), ("rr_mean", C.c_double), ("rr_var", C.c_double), ("rr_count", C.c_double),
        ("last_clipped_action", C.c_double * 4),
        ("pid", C.c_double * 9),
        ("gid", C.c_uint64), ("spawn_pt", C.c_double * 3), ("spawn_ready", C.c_int32),
    ]


class OrcStepOut(C.Structure):
    _fields_ = [("obs", C.c_float * OBS_DIM), ("reward", C.c_double),
                ("terminated", C.c_int32), ("truncated", C.c_int32), ("found_targets", C.c_int32)]


# numpy structured view of OrcEnv (same memory layout) for vectorised access from tests
ENV_DTYPE = np.dtype([
    ("pos", "f8", 3), ("quat", "f8", 4), ("vel", "f8", 3), ("ang_v", "f8", 3), ("rpy", "f8", 3),
    ("cur_pos", "f8", 3), ("cur_vel", "f8", 3), ("cur_ang_v", "f8", 3), ("prev_vel", "f8", 3), ("prev_ang_v", "f8", 3),
    ("d", "f8"), ("d_prev", "f8"),
    ("idx", "i4"), ("just_found", "i4"), ("is_done", "i4"), ("steps", "i4"),
    ("ep_ret", "f8"), ("ep_len", "i4"),
    ("rms_mean", "f8", OBS_DIM), ("rms_var", "f8", OBS_DIM), ("rms_count", "f8"),
    ("step_count", "u8"),
    ("rr_returns", "f8"), ("rr_mean", "f8"), ("rr_var", "f8"), ("rr_count", "f8"),
    ("last_clipped_action", "f8", 4),
    ("pid", "f8", 9),
    ("gid", "u8"), ("spawn_pt", "f8", 3), ("spawn_ready", "i4"),
], align=True)


def build(force=False):
    """Compile liboracle.so with the committed Makefile (gcc, seconds)."""
    src = os.path.join(_HERE, "dn_oracle.c")
    hdr = os.path.join(_HERE, "dn_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, _LIB_NAME] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    fp, dp, u8p, i32p = (C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_int32))
    cfgp, envp = C.POINTER(OrcConfig), C.POINTER(OrcEnv)
    L.orc_constants.argtypes = [dp]
    L.orc_action_bounds.argtypes = [fp, fp]
    L.orc_rescale_action.argtypes = [fp, fp]
    L.orc_preprocess_action.argtypes = [fp, fp]
    L.orc_rotor_forces.argtypes = [fp, fp, fp]
    L.orc_bullet_step.argtypes = [dp, dp, dp, dp, dp, C.c_double]
    L.orc_bullet_step_ex.argtypes = [dp, dp, dp, dp, dp, C.c_double, dp]
    L.orc_bullet_step_damp.argtypes = [dp, dp, dp, dp, dp, C.c_double, dp, C.c_double]
    L.orc_rpm_action.argtypes = [fp, dp, dp, dp]
    L.orc_ground_effect.argtypes = [dp, dp, dp, dp, C.c_int, dp]
    L.orc_drag.argtypes = [dp, dp, dp, C.c_int, dp]
    L.orc_euler_from_quat.argtypes = [dp, dp]
    L.orc_pid_control.argtypes = [C.c_int32, dp, dp, dp, fp, dp, dp]
    L.orc_point_around_line.argtypes = [dp, dp, C.c_double, dp, C.c_double, dp, dp]
    L.orc_random_spawn.argtypes = [cfgp, C.c_uint64, C.c_uint64, dp]
    L.orc_env_construct.argtypes = [cfgp, envp]
    L.orc_env_reset.argtypes = [cfgp, envp, fp]
    L.orc_env_step.argtypes = [cfgp, envp, fp, C.POINTER(OrcStepOut)]
    L.orc_compute_obs.argtypes = [cfgp, envp, fp]
    L.orc_compute_reward.argtypes = [cfgp, envp]
    L.orc_compute_reward.restype = C.c_double
    for name in ("orc_compute_terminated", "orc_compute_truncated", "orc_has_collision"):
        getattr(L, name).argtypes = [cfgp, envp]
        getattr(L, name).restype = C.c_int32
    L.orc_post_step.argtypes = [cfgp, envp]
    L.orc_normalize_obs.argtypes = [envp, fp, dp]
    L.orc_reward_wrappers.argtypes = [cfgp, envp, C.c_double, C.c_int32]
    L.orc_reward_wrappers.restype = C.c_double
    L.orc_vec_create.argtypes = [cfgp, C.c_void_p, C.c_int64]
    L.orc_vec_refresh_rpy.argtypes = [C.c_void_p, C.c_int64]
    L.orc_vec_reset.argtypes = [cfgp, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
    L.orc_vec_step.argtypes = [cfgp, C.c_void_p, C.c_int64] + [C.c_void_p] * 10 + [C.c_int]
    L.orc_gae.argtypes = [C.c_void_p] * 5 + [C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    L.orc_philox4x32.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)]
    L.orc_noise4.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, fp]
    L.orc_noise4_many.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, C.c_uint64, C.c_uint32, fp]
    L.orc_noise4_many.restype = None
    for name in ("orc_sizeof_env", "orc_sizeof_config", "orc_max_threads"):
        getattr(L, name).restype = C.c_int32
    assert L.orc_sizeof_env() == C.sizeof(OrcEnv) == ENV_DTYPE.itemsize, \
        (L.orc_sizeof_env(), C.sizeof(OrcEnv), ENV_DTYPE.itemsize)
    assert L.orc_sizeof_config() == C.sizeof(OrcConfig)
    _lib = L
    return L


def make_config(waypoints, spawn, dim, *, threshold=0.3, max_steps=4096, circle=False, cylinder=True,
                include_distance=True, normalize_actions=True, normalize_obs=False, ground_contact=False,
                f32_state=False, act_noise_sigma=0.0, obs_noise_sigma=0.0, seed=0, env_id_offset=0, clip_rew=False,
                norm_rew=False, physics=0, action_type=0, random_spawn=False, zero_damping=False):
    wp = np.asarray(waypoints, dtype=np.float64).reshape(-1, 3)
    assert 1 <= len(wp) <= MAX_WAYPOINTS
    cfg = OrcConfig()
    cfg.num_waypoints = len(wp)
    for i, v in enumerate(wp.ravel()):
        cfg.waypoints[i] = v
    for i, v in enumerate(np.asarray(spawn, dtype=np.float64).ravel()[:3]):
        cfg.spawn[i] = v
    for i, v in enumerate(np.asarray(dim, dtype=np.float64).ravel()[:6]):
        cfg.dim[i] = v
    cfg.threshold = threshold
    cfg.max_steps = int(max_steps)
    cfg.circle, cfg.cylinder = int(circle), int(cylinder)
    cfg.include_distance, cfg.normalize_actions = int(include_distance), int(normalize_actions)
    cfg.normalize_obs, cfg.ground_contact, cfg.f32_state = int(normalize_obs), int(ground_contact), int(f32_state)
    cfg.act_noise_sigma, cfg.obs_noise_sigma = float(act_noise_sigma), float(obs_noise_sigma)
    cfg.seed, cfg.env_id_offset = int(seed), int(env_id_offset)
    cfg.clip_rew, cfg.norm_rew = int(clip_rew), int(norm_rew)
    cfg.physics, cfg.action_type = int(physics), int(action_type)
    cfg.random_spawn = int(random_spawn)
    cfg.zero_damping = int(zero_damping)
    return cfg


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleVecEnv:
    """N oracle envs stepped the way SubprocVecEnv + Monitor + NormalizeObservation would."""

    def __init__(self, cfg, num_envs, threads=1):
        self.L = lib()
        self.cfg = cfg
        self.n = int(num_envs)
        self.threads = int(threads)
        self.envs = np.zeros(self.n, dtype=ENV_DTYPE)
        self.L.orc_vec_create(C.byref(cfg), _p(self.envs), self.n)

    def reset(self):
        obs = np.empty((self.n, OBS_DIM), np.float32)
        self.L.orc_vec_reset(C.byref(self.cfg), _p(self.envs), self.n, _p(obs), self.threads)
        return obs

    def refresh_rpy(self):
        """After overwriting envs["quat"] (teacher forcing): recompute the cached rpy the force terms read."""
        self.L.orc_vec_refresh_rpy(_p(self.envs), self.n)

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.n, 4)
        out = dict(
            obs=np.empty((self.n, OBS_DIM), np.float32), reward=np.empty(self.n, np.float32),
            done=np.empty(self.n, np.uint8), truncated=np.empty(self.n, np.uint8),
            found_targets=np.empty(self.n, np.int32), terminal_obs=np.zeros((self.n, OBS_DIM), np.float32),
            ep_ret=np.zeros(self.n, np.float32), ep_len=np.zeros(self.n, np.int32),
            terminated=np.empty(self.n, np.uint8))
        self.L.orc_vec_step(C.byref(self.cfg), _p(self.envs), self.n, _p(a), _p(out["obs"]), _p(out["reward"]),
                            _p(out["done"]), _p(out["truncated"]), _p(out["found_targets"]),
                            _p(out["terminal_obs"]), _p(out["ep_ret"]), _p(out["ep_len"]), _p(out["terminated"]),
                            self.threads)
        return out


def gae(rewards, values, dones, last_values, last_dones, gamma, lam):
    T, N = rewards.shape
    r = np.ascontiguousarray(rewards, np.float32)
    v = np.ascontiguousarray(values, np.float32)
    d = np.ascontiguousarray(dones, np.uint8)
    lv = np.ascontiguousarray(last_values, np.float32)
    ld = np.ascontiguousarray(last_dones, np.uint8)
    adv = np.empty((T, N), np.float32)
    ret = np.empty((T, N), np.float32)
    lib().orc_gae(_p(r), _p(v), _p(d), _p(lv), _p(ld), T, N, gamma, lam, _p(adv), _p(ret))
    return adv, ret
