"""drl-dronenavigation_amd -- MI355X-native vectorised drone-navigation RL environment.

The hot path of eRGiBi/DRL-DroneNavigation (PBDroneEnv.step -> BaseAviary.step -> PyBullet,
behind SB3's SubprocVecEnv) as hand-written HIP kernels for gfx950 behind a C ABI
(include/dronenav.h, csrc/), plus the thin Python host mirror of the reference's VecEnv surface.
Importing this package needs the compiled library (there is no CPU fallback); build it with
`python __graft_entry__.py` or `drl-dronenavigation_amd/build.py`.
"""
from . import _capi, build, tracks  # noqa: F401
from ._capi import DroneNavError, DroneNavLibraryError  # noqa: F401
from .tracks import Track  # noqa: F401

__all__ = ["DroneVecEnv", "Track", "tracks", "gae", "DroneNavError", "DroneNavLibraryError", "make_config",
           "RolloutCollector", "ShardPlan", "all_gather_rollout", "preprocess_action", "stream_copy", "MlpActorCritic", "SacActor", "FusedSacActor"]


def __getattr__(name):
    # vec_env imports torch; keep `import drl_dronenavigation_amd` light for tools that only build.
    import importlib
    if name in ("DroneVecEnv", "gae", "make_config", "vec_env", "preprocess_action", "stream_copy"):
        vec_env = importlib.import_module(__name__ + ".vec_env")
        return vec_env if name == "vec_env" else getattr(vec_env, name)
    if name in ("collector", "RolloutCollector", "ShardPlan", "all_gather_rollout", "ReplayBuffer", "RingReplayBuffer", "OffPolicyCollector",
                "FusedRolloutCollector"):
        collector = importlib.import_module(__name__ + ".collector")
        return collector if name == "collector" else getattr(collector, name)
    if name in ("policy_mfma", "FusedMlpPolicy", "FusedSacActor"):
        pm = importlib.import_module(__name__ + ".policy_mfma")
        return pm if name == "policy_mfma" else getattr(pm, name)
    if name == "metrics":
        return importlib.import_module(__name__ + ".metrics")
    if name in ("policy", "MlpActorCritic", "SacActor"):
        policy = importlib.import_module(__name__ + ".policy")
        return policy if name == "policy" else getattr(policy, name)
    raise AttributeError(name)
