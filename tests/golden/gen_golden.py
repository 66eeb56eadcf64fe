#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference's own Python.

Runs only in the build container (needs /root/reference; the GPU box never sees it).
    python tests/golden/gen_golden.py
The fixtures are plain data (inputs + the reference's outputs); no reference source is copied.

What is pinned: A1 A2 A3 (action chain, float32 bit patterns and the
force/torque values handed to the Bullet C-API), A6 (observation packing), A7 A8 A9 (reward,
termination, truncation, post-step/reset bookkeeping incl. quirks Q1-Q5), A10
(normalize.NormalizeObservation), A12 (tracks), N1 (the GAE lines of cleanRLPPO.py).
What is NOT pinned: A4/A5's Bullet half -- the fake pybullet integrates with the oracle's own
restatement (tests/golden/_ref_stubs.py), so closed-loop fixtures validate everything AROUND
the integrator, not the integrator.

SB3's SubprocVecEnv worker and Monitor are not in the tree either; `RefVec` below restates
their auto-reset/episode-statistics semantics from memory [3P-recall] around the real
reference env + the real reference normalize.NormalizeObservation.
"""
import os
import sys
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import oracle as O          # noqa: E402
import _ref_stubs                        # noqa: E402

pb = _ref_stubs.install(O.lib())
os.chdir(REF)                            # the reference opens "Sol/resources/..." relative to cwd
sys.path.insert(0, REF)
import io                                # noqa: E402
import contextlib                        # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    from Sol.Model.Environments.PBDroneEnv import PBDroneEnv      # noqa: E402
    from Sol.Model.Environments import normalize as ref_normalize  # noqa: E402
    from Sol.Utilities import Waypoints                            # noqa: E402
    from Sol.PyBullet.enums import ActionType                      # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def make_ref_env(targets, spawn, dim, circle, max_steps=4096, threshold=0.3):
    """PBDroneEnv exactly as PBDroneSimulator.make_env builds it (PBDroneSimulator.py:154-171)."""
    return quiet(PBDroneEnv, target_points=targets, threshold=threshold, discount=0.999, max_steps=max_steps,
                 act=ActionType.THRUST, gui=False, initial_xyzs=spawn, save_folder=None, aviary_dim=dim,
                 random_spawn=False, cylinder=True, circle=circle, include_distance=True,
                 normalize_actions=True, collect_rollouts=False)


def track_circle(n):
    wp, spawn, dim = Waypoints.circle(radius=1, num_points=n, height=1)
    targets = [np.array(w) for w in wp]
    targets.pop(0)                       # PBDroneSimulator.py:129-130 (circle tracks drop the first point)
    return targets, spawn, dim, True


def track_reaching():
    wp, spawn, dim = Waypoints.reaching()
    return [np.array(w) for w in wp], spawn, dim, False


def internals(env):
    e = env
    return dict(pos=e.pos[0].copy(), quat=e.quat[0].copy(), rpy=e.rpy[0].copy(), vel=e.vel[0].copy(),
                ang_v=e.ang_v[0].copy(), cur_pos=np.array(e._current_position, dtype=np.float64).copy(),
                cur_vel=np.array(e.current_vel, dtype=np.float64), cur_ang_v=np.array(e.current_ang_v, dtype=np.float64),
                prev_vel=np.array(e.prev_vel, dtype=np.float64), prev_ang_v=np.array(e.prev_ang_v, dtype=np.float64),
                d=float(e._distance_to_target), d_prev=float(e._prev_distance_to_target),
                idx=int(e._current_target_index), just_found=int(e.just_found), is_done=int(e._is_done),
                steps=int(e._steps))


INT_KEYS = ("pos", "quat", "rpy", "vel", "ang_v", "cur_pos", "cur_vel", "cur_ang_v", "prev_vel", "prev_ang_v",
            "d", "d_prev", "idx", "just_found", "is_done", "steps")


class RefVec:
    """[3P-recall] SubprocVecEnv worker + Monitor around the real reference env (+ real normaliser)."""

    def __init__(self, mk, n, normalize_obs):
        self.raw, self.envs = [], []
        for i in range(n):
            env = mk()
            quiet(env.reset, seed=i)                         # make_env: env.reset(seed=seed+rank)
            self.raw.append(env)
            self.envs.append(ref_normalize.NormalizeObservation(env) if normalize_obs else env)
        self.rewards = [[] for _ in range(n)]

    def reset(self):
        out = []
        for i, env in enumerate(self.envs):
            obs, _ = quiet(env.reset)
            self.rewards[i] = []
            out.append(np.asarray(obs))
        return np.stack(out)

    def step(self, actions):
        n = len(self.envs)
        res = dict(obs=np.zeros((n, 13), np.float64), reward=np.zeros(n, np.float64), done=np.zeros(n, np.uint8),
                   truncated=np.zeros(n, np.uint8), terminated=np.zeros(n, np.uint8),
                   found_targets=np.zeros(n, np.int32), terminal_obs=np.zeros((n, 13), np.float64),
                   ep_ret=np.zeros(n, np.float64), ep_len=np.zeros(n, np.int32))
        for i, env in enumerate(self.envs):
            obs, r, term, trunc, info = quiet(env.step, actions[i])
            self.rewards[i].append(float(r))                                 # Monitor.step
            done = bool(term or trunc)
            res["reward"][i] = r
            res["done"][i] = done
            res["terminated"][i] = bool(term)
            res["truncated"][i] = bool(trunc and not term)                   # info["TimeLimit.truncated"]
            res["found_targets"][i] = info["found_targets"]
            if done:
                res["ep_ret"][i] = sum(self.rewards[i])                      # Monitor: ep_rew = sum(self.rewards)
                res["ep_len"][i] = len(self.rewards[i])
                res["terminal_obs"][i] = obs                                 # info["terminal_observation"]
                obs, _ = quiet(env.reset)
                self.rewards[i] = []
            res["obs"][i] = obs
        return res


# --------------------------------------------------------------------------------------
def gen_constants_and_actions(out):
    targets, spawn, dim, circle = track_circle(4)
    env = make_ref_env(targets, spawn, dim, circle)
    consts = dict(M=env.M, L=env.L, KF=env.KF, KM=env.KM, IXX=env.J[0, 0], IYY=env.J[1, 1], IZZ=env.J[2, 2],
                  PWM2RPM_SCALE=env.PWM2RPM_SCALE, PWM2RPM_CONST=env.PWM2RPM_CONST, MIN_PWM=env.MIN_PWM,
                  MAX_PWM=env.MAX_PWM, G=env.G, GRAVITY=env.GRAVITY, HOVER_RPM=env.HOVER_RPM, MAX_RPM=env.MAX_RPM,
                  PYB_TIMESTEP=env.PYB_TIMESTEP, PYB_STEPS_PER_CTRL=env.PYB_STEPS_PER_CTRL,
                  max_target_dist=env._max_target_dist)
    a_low, a_high = env.physical_action_bounds
    rng = np.random.default_rng(7)
    acts = np.concatenate([
        np.linspace(-1, 1, 2001), np.arange(0.0895, 0.0976, 1e-5), rng.uniform(-1, 1, 4000),
        0.0922 + 0.003 * rng.standard_normal(4000), [0.0, 1.0, -1.0, 0.092227, 1e-30, -1e-30, 5.0, -5.0],
    ]).astype(np.float32)
    acts = acts[: (len(acts) // 4) * 4].reshape(-1, 4)
    resc = np.stack([env.rescale_action(a) for a in acts])
    rpm = np.stack([env._preprocessAction(r) for r in resc])
    assert resc.dtype == np.float32 and rpm.dtype == np.float32
    forces = np.zeros((len(acts), 4))
    zt = np.zeros(len(acts))
    cl = pb._clients[env.CLIENT]
    kinds = set()
    for k, r in enumerate(rpm):
        cl["applied"].clear()
        env._physics(r, 0)
        for kind, link, tname, val in cl["applied"]:
            kinds.add(tname)
            if kind == "F":
                forces[k, link] = val
            else:
                zt[k] = val
    out["actions"] = dict(actions=acts, rescaled=resc, rpm=rpm, forces=forces, z_torque=zt,
                          a_low=np.float32(a_low[0]), a_high=np.float32(a_high[0]),
                          api_scalar_types=np.array(sorted(kinds)),
                          **{"const_" + k: np.float64(v) for k, v in consts.items()})


def gen_tracks(out):
    d = {}
    for name, fn in (("circle4", lambda: Waypoints.circle(1, 4, 1)), ("circle6", lambda: Waypoints.circle(1, 6, 1)),
                     ("reaching", Waypoints.reaching), ("up", Waypoints.up),
                     ("half_up_forward", Waypoints.half_up_forward), ("up_circle", Waypoints.up_circle),
                     ("up_sharp_back_turn", Waypoints.up_sharp_back_turn)):
        wp, spawn, dim = fn()
        d[name + "_waypoints"] = np.array(wp, dtype=np.float64)
        d[name + "_spawn"] = np.array(spawn, dtype=np.float64).reshape(-1)
        d[name + "_dim"] = np.array(dim, dtype=np.float64)
    out["tracks"] = d


def gen_obs_pack(out):
    """A6: random kinematic states -> _computeObs (teacher-forced through the fake Bullet getters)."""
    rng = np.random.default_rng(11)
    res = {}
    for tname, mk in (("circle", lambda: track_circle(4)), ("race", track_reaching)):
        targets, spawn, dim, circle = mk()
        env = make_ref_env(targets, spawn, dim, circle)
        cl = pb._clients[env.CLIENT]
        n = 600
        pos = rng.uniform(-5, 5, (n, 3))
        quat = rng.standard_normal((n, 4))
        quat /= np.linalg.norm(quat, axis=1, keepdims=True)
        vel = rng.standard_normal((n, 3)) * np.array([4, 4, 2])
        ang = rng.standard_normal((n, 3)) * 5
        ang[::50] = 0.0                                       # zero angular velocity stays zero
        vel[1::50] = [3.0, -3.0, 1.0]                         # clip edges
        quat[2::50] = [0, np.sin(np.pi / 4), 0, np.cos(np.pi / 4)]   # gimbal-lock branch of the Euler conversion
        quat[3::50] = [0, -np.sin(np.pi / 4), 0, np.cos(np.pi / 4)]
        dist = rng.uniform(0, 6, n)
        obs = np.zeros((n, 13), np.float32)
        rpy = np.zeros((n, 3))
        for k in range(n):
            cl["pos"], cl["quat"], cl["vel"], cl["ang_v"] = pos[k].copy(), quat[k].copy(), vel[k].copy(), ang[k].copy()
            env._updateAndStoreKinematicInformation()
            env._distance_to_target = np.float64(dist[k])
            o = env._computeObs()
            assert o.dtype == np.float32
            obs[k] = o
            rpy[k] = env.rpy[0]
        res.update({tname + "_pos": pos, tname + "_quat": quat, tname + "_vel": vel, tname + "_ang_v": ang,
                    tname + "_dist": dist, tname + "_obs": obs, tname + "_rpy": rpy})
    out["obs_pack"] = res


def run_closed_loop(mk_track, n, T, action_fn, normalize_obs, max_steps, seed):
    targets, spawn, dim, circle = mk_track()
    vec = RefVec(lambda: make_ref_env(targets, spawn, dim, circle, max_steps=max_steps), n, normalize_obs)
    rng = np.random.default_rng(seed)
    rec = dict(reset_obs=vec.reset().astype(np.float64))
    keys = ("obs", "reward", "done", "truncated", "terminated", "found_targets", "terminal_obs", "ep_ret", "ep_len")
    steps = {k: [] for k in keys}
    ints = {k: [] for k in INT_KEYS}
    actions = []
    for t in range(T):
        a = action_fn(rng, t, n).astype(np.float32)
        actions.append(a)
        r = vec.step(a)
        for k in keys:
            steps[k].append(r[k])
        cur = [internals(e) for e in vec.raw]
        for k in INT_KEYS:
            ints[k].append(np.stack([np.asarray(c[k]) for c in cur]))
    rec["actions"] = np.stack(actions)
    for k in keys:
        rec[k] = np.stack(steps[k])
    for k in INT_KEYS:
        rec["int_" + k] = np.stack(ints[k])
    rec["waypoints"] = np.array(targets, dtype=np.float64)
    rec["spawn"] = np.array(spawn, dtype=np.float64).reshape(-1)
    rec["dim"] = np.array(dim, dtype=np.float64)
    rec["circle"] = np.int32(circle)
    rec["max_steps"] = np.int32(max_steps)
    rec["normalize_obs"] = np.int32(normalize_obs)
    if normalize_obs:
        rec["rms_mean"] = np.stack([e.obs_rms.mean for e in vec.envs])
        rec["rms_var"] = np.stack([e.obs_rms.var for e in vec.envs])
        rec["rms_count"] = np.array([e.obs_rms.count for e in vec.envs])
    return rec


def a_uniform(rng, t, n):
    return rng.uniform(-1, 1, (n, 4))


def a_hover(rng, t, n):
    return 0.0922 + 0.003 * rng.standard_normal((n, 4))


def a_mixed(rng, t, n):
    a = 0.0922 + 0.002 * rng.standard_normal((n, 4))
    a[: n // 2] += 0.0004 * np.sin(t / 15.0 + np.arange(n // 2))[:, None] * np.array([1, -1, -1, 1])
    return a


def gen_closed_loop(out):
    out["traj_circle_uniform"] = run_closed_loop(lambda: track_circle(4), 8, 300, a_uniform, False, 4096, 1)
    out["traj_circle_hover"] = run_closed_loop(lambda: track_circle(4), 8, 400, a_hover, False, 150, 2)
    out["traj_race_uniform"] = run_closed_loop(track_reaching, 8, 300, a_uniform, False, 4096, 3)
    out["traj_race_mixed_norm"] = run_closed_loop(track_reaching, 8, 400, a_mixed, True, 120, 4)
    out["traj_circle6_norm"] = run_closed_loop(lambda: track_circle(6), 4, 300, a_hover, True, 4096, 5)


# --------------------------------------------------------------------------------------
def scripted_path(targets, spawn, rng, per_seg, lateral, drift, n_laps=1, arc=False):
    """Positions that follow spawn -> wp0 -> wp1 ... with lateral noise and an optional growing drift
    (along the unit circle at z=1 when `arc`, so a circle track's torus corridor is respected)."""
    pts = [np.array(spawn, dtype=np.float64).reshape(3)] + [np.array(t, dtype=np.float64) for t in targets]
    pos = []
    k = 0
    for _ in range(n_laps):
        for a, b in zip(pts[:-1], pts[1:]):
            for s in range(per_seg):
                u = (s + 1) / per_seg
                p = a + (b - a) * u
                if arc:
                    th0, th1 = np.arctan2(a[1], a[0]), np.arctan2(b[1], b[0])
                    th = th0 + ((th1 - th0) % (2 * np.pi)) * u
                    p = np.array([np.cos(th), np.sin(th), 1.0])
                p = p + lateral * rng.standard_normal(3) + drift * k * np.array([0.3, -0.2, 0.1])
                pos.append(p)
                k += 1
    return np.array(pos)


def run_scripted(mk_track, per_seg, lateral, drift, max_steps, seed, n_after=40):
    """Teacher-forced: the fake Bullet is frozen and its state is written before every step, so the
    reference's reward/termination/bookkeeping code sees a chosen kinematic sequence."""
    targets, spawn, dim, circle = mk_track()
    env = make_ref_env(targets, spawn, dim, circle, max_steps=max_steps)
    cl = pb._clients[env.CLIENT]
    quiet(env.reset, seed=0)
    quiet(env.reset)
    rng = np.random.default_rng(seed)
    path = scripted_path(targets, spawn, rng, per_seg, lateral, drift, arc=bool(circle))
    path = np.concatenate([path, path[-1] + np.cumsum(0.02 * rng.standard_normal((n_after, 3)), axis=0)])
    T = len(path)
    vel = np.gradient(path, 1 / 240.0, axis=0) * rng.uniform(0.5, 1.5, (T, 1))
    heading = np.arctan2(vel[:, 1], vel[:, 0]) + 0.15 * rng.standard_normal(T)
    pitch = 0.2 * rng.standard_normal(T)
    roll = 0.2 * rng.standard_normal(T)
    ang = rng.standard_normal((T, 3)) * rng.choice([0.05, 2.0, 40.0], (T, 1))
    vel[rng.random(T) < 0.1] *= 40.0                          # velocity jumps -> smoothness penalties
    quat = np.array([pb.getQuaternionFromEuler([roll[t], pitch[t], heading[t]]) for t in range(T)])
    rec = {k: [] for k in ("obs", "reward", "terminated", "truncated", "found_targets", "reset_obs")}
    ints = {k: [] for k in INT_KEYS}
    cl["frozen"] = True
    for t in range(T):
        cl["pos"], cl["quat"], cl["vel"], cl["ang_v"] = path[t].copy(), quat[t].copy(), vel[t].copy(), ang[t].copy()
        # freeze: write the state AFTER the (no-op) stepSimulation by patching it in place
        obs, r, term, trunc, info = quiet(env.step, np.full(4, 0.0922, np.float32))
        rec["obs"].append(obs)
        rec["reward"].append(float(r))
        rec["terminated"].append(bool(term))
        rec["truncated"].append(bool(trunc))
        rec["found_targets"].append(info["found_targets"])
        if term or trunc:
            o, _ = quiet(env.reset)
            rec["reset_obs"].append(o)
        else:
            rec["reset_obs"].append(np.zeros(13, np.float32))
        cur = internals(env)
        for k in INT_KEYS:
            ints[k].append(np.asarray(cur[k]))
    res = dict(pos=path, quat=quat, vel=vel, ang_v=ang, waypoints=np.array(targets, dtype=np.float64),
               spawn=np.array(spawn, dtype=np.float64).reshape(-1), dim=np.array(dim, dtype=np.float64),
               circle=np.int32(circle), max_steps=np.int32(max_steps))
    for k, v in rec.items():
        res[k] = np.array(v)
    for k, v in ints.items():
        res["int_" + k] = np.stack(v)
    return res


def gen_scripted(out):
    out["script_circle_follow"] = run_scripted(lambda: track_circle(4), 40, 0.03, 0.0, 4096, 21)
    out["script_circle_drift"] = run_scripted(lambda: track_circle(4), 30, 0.05, 0.004, 4096, 22)
    out["script_circle_trunc"] = run_scripted(lambda: track_circle(4), 40, 0.02, 0.0, 37, 23)
    out["script_race_follow"] = run_scripted(track_reaching, 60, 0.04, 0.0, 4096, 24)
    out["script_race_drift"] = run_scripted(track_reaching, 40, 0.08, 0.001, 4096, 25)
    out["script_race_trunc"] = run_scripted(track_reaching, 50, 0.03, 0.0, 61, 26)
    up = lambda: ([np.array(w) for w in Waypoints.up()[0]], np.array([Waypoints.up()[1]]), Waypoints.up()[2], False)  # noqa: E731
    out["script_up_follow"] = run_scripted(up, 30, 0.02, 0.0, 4096, 27)


def gen_normalize(out):
    """A10 alone: a 300-long observation stream through normalize.NormalizeObservation.normalize."""
    class _E:
        observation_space = _ref_stubs.Box(-np.ones(13), np.ones(13), dtype=np.float32)
    w = ref_normalize.NormalizeObservation(_E())
    rng = np.random.default_rng(31)
    x = (rng.standard_normal((300, 13)) * rng.uniform(0.01, 3, 13) + rng.uniform(-1, 1, 13)).astype(np.float32)
    x[:, 4] = 0.0
    y = np.stack([w.normalize(np.array([o]))[0] for o in x])
    out["normalize"] = dict(x=x, y=y, mean=w.obs_rms.mean, var=w.obs_rms.var, count=np.float64(w.obs_rms.count))


def gen_reward_wrappers(out):
    """N4: make_env's optional reward wrappers (PBDroneSimulator.py:191-194): the reference's NormalizeReward (its
    normalize.py copy of gym's) around a scripted env, with and without the clip lambda in front of it."""
    rng = np.random.default_rng(41)
    T = 400
    rews = np.where(rng.random(T) < 0.1, rng.choice([-10.0, 8.0, 3.0, 40.0, -25.0], T), rng.normal(0.05, 0.4, T))
    dones = rng.random(T) < 0.04

    class _E:
        def __init__(self):
            self.t = 0

        def step(self, action):
            r, d = float(rews[self.t]), bool(dones[self.t])
            self.t += 1
            return None, r, d, False, {}

    clip = lambda r: np.clip(r, -10, 10)                     # noqa: E731  (the lambda of PBDroneSimulator.py:192)
    res = {}
    for tag, do_clip in (("norm", False), ("clip_norm", True)):
        env = _E()
        if do_clip:
            inner = env.step
            env.step = lambda a, inner=inner: (lambda o, r, d, tr, i: (o, clip(r), d, tr, i))(*inner(a))
        w = ref_normalize.NormalizeReward(env)
        ys = np.array([float(w.step(None)[1]) for _ in range(T)])
        res[tag] = ys
        res[tag + "_mean"], res[tag + "_var"] = np.float64(w.return_rms.mean), np.float64(w.return_rms.var)
        res[tag + "_count"], res[tag + "_returns"] = np.float64(w.return_rms.count), np.float64(w.returns[0])
    out["reward_wrappers"] = dict(rewards=rews, dones=dones.astype(np.uint8), **res)


def gen_extra_physics(out):
    """N4: the force terms of Physics.PYB_GND / PYB_DRAG (BaseAviary._groundEffect / _drag) and ActionType.RPM
    (BaseSingleAgentAviary._preprocessAction + BaseAviary._physics), called on the reference's own methods with the
    kinematic caches set by hand.  Pins the numpy halves (dtypes, operation order); p.getLinkStates and
    p.getMatrixFromQuaternion are the stub's [3P-recall]."""
    from Sol.PyBullet.BaseSingleAgentAviary import BaseSingleAgentAviary
    targets, spawn, dim, circle = track_circle(4)
    env = make_ref_env(targets, spawn, dim, circle)
    cl = pb._clients[env.CLIENT]
    rng = np.random.default_rng(11)
    n = 600
    pos = rng.uniform([-2, -2, 0.0], [2, 2, 2], (n, 3))
    pos[:150, 2] = rng.uniform(0.0, 0.08, 150)              # near the ground: clip at GND_EFF_H_CLIP and large effects
    ang = rng.uniform(-1, 1, (n, 3)) * np.array([2.5, 1.7, np.pi])
    ang[:100] *= 0.2
    quat = np.array([pb.getQuaternionFromEuler(a) for a in ang])
    vel = rng.uniform(-3, 3, (n, 3))
    acts = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
    acts[::3] = (0.0922 + 0.003 * rng.standard_normal((len(acts[::3]), 4))).astype(np.float32)
    rpm32 = np.stack([env._preprocessAction(env.rescale_action(a)) for a in acts])
    assert rpm32.dtype == np.float32
    env.ACT_TYPE = ActionType.RPM
    rpm64 = np.stack([BaseSingleAgentAviary._preprocessAction(env, a) for a in acts])
    env.ACT_TYPE = ActionType.THRUST
    assert rpm64.dtype == np.float64
    res = {}
    for tag, rpms in (("f32", rpm32), ("f64", rpm64)):
        gnd = np.zeros((n, 4))
        drag = np.zeros((n, 3))
        forces = np.zeros((n, 4))
        zt = np.zeros(n)
        rpy = np.zeros((n, 3))
        for k in range(n):
            cl["pos"][:] = pos[k]; cl["quat"][:] = quat[k]; cl["vel"][:] = vel[k]
            quiet(env._updateAndStoreKinematicInformation)
            rpy[k] = env.rpy[0]
            cl["applied"].clear()
            env._groundEffect(rpms[k], 0)
            for kind, link, tname, val in cl["applied"]:
                assert kind == "F" and tname == "float64"
                gnd[k, link] = val
            cl["applied"].clear()
            env._drag(rpms[(k + 1) % n], 0)
            (kind, link, tname, val), = cl["applied"]
            assert kind == "D"
            drag[k] = val
            cl["applied"].clear()
            env._physics(rpms[k], 0)
            for kind, link, tname, val in cl["applied"]:
                if kind == "F":
                    forces[k, link] = val
                else:
                    zt[k] = val
            cl["forces"][:] = 0
        res.update({f"gnd_{tag}": gnd, f"drag_{tag}": drag, f"forces_{tag}": forces, f"z_torque_{tag}": zt, f"rpm_{tag}": rpms})
    out["extra_physics"] = dict(pos=pos, quat=quat, vel=vel, rpy=rpy, actions=acts,
                                GND_EFF_H_CLIP=np.float64(env.GND_EFF_H_CLIP), HOVER_RPM=np.float64(env.HOVER_RPM), **res)


def gen_pid_control(out):
    """N4: ActionType.PID / VEL / ONE_D_RPM / ONE_D_PID (BaseSingleAgentAviary._preprocessAction, BaseSingleAgentAviary.py:176-222)
    with the DSLPIDControl loop (Sol/PyBullet/DSLPIDControl.py) -- the reference's own methods, called unbound on a
    PBDroneEnv whose ACT_TYPE is switched (PBDroneEnv's own _preprocessAction override never reaches them).  One
    controller per action type lives through the whole sequence (the reference never calls ctrl.reset() after
    construction), fed a random walk of kinematic states through the fake Bullet getters."""
    from Sol.PyBullet.BaseSingleAgentAviary import BaseSingleAgentAviary
    from Sol.PyBullet.DSLPIDControl import DSLPIDControl
    from Sol.PyBullet.enums import DroneModel
    targets, spawn, dim, circle = track_circle(4)
    env = make_ref_env(targets, spawn, dim, circle)
    cl = pb._clients[env.CLIENT]
    rng = np.random.default_rng(23)
    T = 400
    pos = np.array([1.0, 0.0, 1.0]) + np.cumsum(0.01 * rng.standard_normal((T, 3)), axis=0)
    ang = np.cumsum(0.03 * rng.standard_normal((T, 3)), axis=0)
    ang[T // 2:T // 2 + 20, 1] = np.pi / 2 - 1e-7             # a stretch inside getEulerFromQuaternion's gimbal-lock branch
    quat = np.array([pb.getQuaternionFromEuler(a) for a in ang])
    vel = 0.5 * rng.standard_normal((T, 3))
    ang_v = rng.standard_normal((T, 3))
    acts = rng.uniform(-1, 1, (T, 4)).astype(np.float32)
    acts[::7, :3] = 0.0                                        # VEL: zero direction vector
    acts[::11] *= 3.0                                          # PID: destinations farther than one step away
    res = dict(pos=pos, quat=quat, vel=vel, ang_v=ang_v, actions=acts, SPEED_LIMIT=np.float64(0.03 * env.MAX_SPEED_KMH * (1000 / 3600)),
               CTRL_TIMESTEP=np.float64(env.CTRL_TIMESTEP))
    env.SPEED_LIMIT = float(res["SPEED_LIMIT"])
    for name, at, adim in (("pid", ActionType.PID, 3), ("vel", ActionType.VEL, 4), ("one_d_rpm", ActionType.ONE_D_RPM, 1),
                           ("one_d_pid", ActionType.ONE_D_PID, 1)):
        env.ACT_TYPE = at
        env.ctrl = quiet(DSLPIDControl, drone_model=DroneModel.CF2X)
        rpm = np.zeros((T, 4))
        ipos, lrpy, irpy = np.zeros((T, 3)), np.zeros((T, 3)), np.zeros((T, 3))
        for t in range(T):
            cl["pos"], cl["quat"], cl["vel"], cl["ang_v"] = pos[t].copy(), quat[t].copy(), vel[t].copy(), ang_v[t].copy()
            quiet(env._updateAndStoreKinematicInformation)
            r = quiet(BaseSingleAgentAviary._preprocessAction, env, acts[t, :adim].copy())
            rpm[t] = np.asarray(r, dtype=np.float64).reshape(4)
            if at != ActionType.ONE_D_RPM:
                ipos[t], lrpy[t], irpy[t] = env.ctrl.integral_pos_e, env.ctrl.last_rpy, env.ctrl.integral_rpy_e
        res.update({f"{name}_rpm": rpm, f"{name}_integral_pos_e": ipos, f"{name}_last_rpy": lrpy, f"{name}_integral_rpy_e": irpy})
    env.ACT_TYPE = ActionType.THRUST
    out["pid_control"] = res


def gen_random_spawn(out):
    """N4: PositionGenerator.generate_random_point_around_line (Sol/Utilities/position_generator.py:121-152), the geometry of the
    dormant random-spawn block (PBDroneEnv.py:622-627).  Its draws come from `random` / `np.random`; here they are supplied
    (the two modules' functions are patched for the duration of the call), so the fixture pins the arithmetic: interpolation,
    the perpendicular through a cross product, the offset, the clip to the aviary bounds."""
    import random
    from unittest import mock
    from Sol.Utilities.position_generator import PositionGenerator
    rng = np.random.default_rng(31)
    n = 400
    bounds = np.array([-2.0, -2.0, 0.0, 2.0, 2.0, 2.0])
    gen = PositionGenerator(bounds, 0.1)                      # PBDroneEnv.py:168-169
    frm = rng.uniform([-2, -2, 0], [2, 2, 2], (n, 3))
    to = rng.uniform([-2, -2, 0], [2, 2, 2], (n, 3))
    frm[:40] = np.clip(frm[:40] * 1.2, bounds[:3], bounds[3:])   # some on the faces of the box: the clip binds
    t = rng.uniform(0, 1, n)
    rv = rng.standard_normal((n, 3))
    u = rng.uniform(0, 1, n)                                  # random.uniform(a, b) = a + (b - a) * random()
    pts = np.zeros((n, 3))
    for k in range(n):
        draws = iter([t[k], -0.1 + (0.1 - -0.1) * u[k]])
        with mock.patch.object(random, "uniform", lambda a, b: next(draws)), \
                mock.patch.object(np.random, "randn", lambda *shape: rv[k].copy()):
            pts[k] = gen.generate_random_point_around_line(frm[k], to[k])
    out["random_spawn"] = dict(frm=frm, to=to, t=t, rv=rv, u=u, bounds=bounds, max_distance=np.float64(0.1), points=pts)


def gen_dead_dynamics(out):
    """Row A4, structure only: the reference's own explicit rigid-body model, BaseAviary._dynamics + _integrateQ
    (BaseAviary.py:899-973).  It is dead code (Physics.DYN is never selected and `self.TIMESTEP` is undefined, :944), differs
    from what Bullet simulates in three declared ways (no damping; arm L / sqrt(2) instead of the loaded URDF's 0.028; the y
    signs of the safegym URDF's prop layout) and is NOT Bullet -- but it is the one statement of the
    semi-implicit step, the thrust direction, the yaw-torque convention and the gyroscopic term that the reference itself
    owns.  Called unbound on a PBDroneEnv with TIMESTEP set; the fake Bullet records what it hands to
    resetBasePositionAndOrientation / resetBaseVelocity."""
    from Sol.PyBullet.BaseAviary import BaseAviary
    targets, spawn, dim, circle = track_circle(4)
    env = make_ref_env(targets, spawn, dim, circle)
    env.TIMESTEP = env.PYB_TIMESTEP
    env.rpy_rates = np.zeros((1, 3))                           # _housekeeping creates it only for Physics.DYN (BaseAviary.py:547-548)
    rng = np.random.default_rng(41)
    n = 300
    pos = rng.uniform(-2, 2, (n, 3)) + [0, 0, 3]
    quat = rng.standard_normal((n, 4))
    quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    vel = rng.normal(0, 2.0, (n, 3))
    rates = rng.normal(0, 8.0, (n, 3))                        # body-frame angular velocity (rpy_rates)
    rates[:3] = 0.0                                           # _integrateQ's |omega| = 0 branch
    rpm = rng.uniform(9440.3, 21666.4, (n, 4))
    got = {k: np.zeros((n, d)) for k, d in (("pos", 3), ("quat", 4), ("vel", 3), ("ang_v_world", 3), ("rates", 3))}
    captured = {}
    orig_pose, orig_vel = pb.resetBasePositionAndOrientation, pb.resetBaseVelocity
    pb.resetBasePositionAndOrientation = lambda bid, p_, q_, **kw: captured.update(pos=np.array(p_, float), quat=np.array(q_, float))
    pb.resetBaseVelocity = lambda bid, v_, w_, **kw: captured.update(vel=np.array(v_, float), ang=np.array(w_, float))
    try:
        for k in range(n):
            env.pos[0], env.quat[0], env.vel[0], env.rpy_rates[0] = pos[k], quat[k], vel[k], rates[k]
            BaseAviary._dynamics(env, rpm[k].copy(), 0)
            got["pos"][k], got["quat"][k], got["vel"][k], got["ang_v_world"][k] = captured["pos"], captured["quat"], captured["vel"], captured["ang"]
            got["rates"][k] = env.rpy_rates[0]
        # second set: thrust COMMANDS through the reference's own action chain (PBDroneEnv._preprocessAction -> float32 rpm ->
        # _dynamics), in the pattern (a, b, a, b): no roll / pitch torque in either prop layout, so this model and the body
        # Bullet simulates coincide once the damping is off -- a direct target for the oracle and the HIP step with
        # dn_config.zero_damping (the changeDynamics line the reference keeps commented out, BaseAviary.py:571-573).
        # Inputs are float32-representable (the HIP state's type).
        m = 256
        f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)      # noqa: E731
        pos2 = f32(rng.uniform(-2, 2, (m, 3)) + [0, 0, 3])
        q2 = rng.standard_normal((m, 4)); q2 /= np.linalg.norm(q2, axis=1, keepdims=True); q2 = f32(q2)
        vel2, rates2 = f32(rng.normal(0, 2.0, (m, 3))), rng.normal(0, 6.0, (m, 3))
        ab = rng.uniform(0.03, 0.14, (m, 2)).astype(np.float32)
        thrust = np.stack([ab[:, 0], ab[:, 1], ab[:, 0], ab[:, 1]], axis=1)
        rpm2 = np.zeros((m, 4), np.float32)
        got2 = {k: np.zeros((m, d)) for k, d in (("pos", 3), ("quat", 4), ("vel", 3), ("ang_v_world", 3))}
        w_in = np.zeros((m, 3))
        for k in range(m):
            R = np.array(pb.getMatrixFromQuaternion(q2[k])).reshape(3, 3)
            w_in[k] = f32(R @ rates2[k])                          # world angular velocity as the float32 state holds it
            rates2[k] = R.T @ w_in[k]
            rpm2[k] = env._preprocessAction(thrust[k].copy())
            assert rpm2[k].dtype == np.float32
            env.pos[0], env.quat[0], env.vel[0], env.rpy_rates[0] = pos2[k], q2[k], vel2[k], rates2[k]
            BaseAviary._dynamics(env, rpm2[k].copy(), 0)
            got2["pos"][k], got2["quat"][k], got2["vel"][k], got2["ang_v_world"][k] = captured["pos"], captured["quat"], captured["vel"], captured["ang"]
    finally:
        pb.resetBasePositionAndOrientation, pb.resetBaseVelocity = orig_pose, orig_vel
    out["dead_dynamics"] = dict(pos=pos, quat=quat, vel=vel, rates=rates, rpm=rpm, L=np.float64(env.L), KF=np.float64(env.KF),
                                KM=np.float64(env.KM), **{"out_" + k: v for k, v in got.items()},
                                chain_pos=pos2, chain_quat=q2, chain_vel=vel2, chain_ang_v=w_in, chain_thrust=thrust, chain_rpm=rpm2,
                                **{"chain_out_" + k: v for k, v in got2.items()})


def gen_gae(out):
    """N1: execute the reference's GAE lines (cleanRLPPO.py:234-248) on random buffers."""
    import torch
    src = open(os.path.join(REF, "Sol/Model/Algorithms/cleanRLPPO.py")).read().splitlines()
    block = textwrap.dedent("\n".join(src[233:248]))          # lines 234..248, 1-based
    assert "lastgaelam" in block and "returns = advantages + values" in block
    rng = np.random.default_rng(41)
    T, N = 33, 17

    class _Args:
        num_steps, gamma, gae_lambda = T, 0.99, 0.95

    class _Agent:
        def __init__(self, v):
            self.v = v

        def get_value(self, obs):
            return self.v

    rewards = torch.tensor(rng.standard_normal((T, N)).astype(np.float32))
    values = torch.tensor(rng.standard_normal((T, N)).astype(np.float32))
    dones = torch.tensor((rng.random((T, N)) < 0.1).astype(np.float32))
    next_done = torch.tensor((rng.random(N) < 0.1).astype(np.float32))
    next_value = torch.tensor(rng.standard_normal(N).astype(np.float32))
    ns = dict(torch=torch, args=_Args, agent=_Agent(next_value), next_obs=None, rewards=rewards, values=values,
              dones=dones, next_done=next_done, device="cpu")
    exec(block, ns)
    out["gae"] = dict(rewards=rewards.numpy(), values=values.numpy(), dones=dones.numpy().astype(np.uint8),
                      next_done=next_done.numpy().astype(np.uint8), next_value=next_value.numpy(),
                      gamma=np.float64(0.99), gae_lambda=np.float64(0.95),
                      advantages=ns["advantages"].numpy(), returns=ns["returns"].numpy())


def gen_rollout_dump(out):
    """N3: the reference's rollout text dump (PBDroneEnv.collect_rollout, PBDroneEnv.py:811-821): its own method,
    called on a minimal stand-in for `self`, over observation/reward pairs incl. awkward float32 values."""
    import tempfile
    import threading
    import types
    rng = np.random.default_rng(7)
    obs = rng.uniform(-1, 1, (40, 13)).astype(np.float32)
    obs[0] = np.float32([0, -0.0, 1, -1, 0.5, 1e-7, -1e-7, 1e-20, 3.4e38, 0.1, 1 / 3, 2 / 3, 1e-45])
    obs[1, :4] = np.float32([0.0899400, 0.3535533845424652, 123456.789, 1e10])
    rewards = [np.float64(x) for x in rng.uniform(-10, 8, 38)] + [-10.0, 8.0]     # np.float64 and Python floats
    rewards[2] = np.float64(np.float32(-0.11290731))                              # a float32-valued reward
    with tempfile.TemporaryDirectory() as d:
        stub = types.SimpleNamespace(rollout_path=os.path.join(d, "rollouts.txt"), lock=threading.Lock())
        for o, r in zip(obs, rewards):
            PBDroneEnv.collect_rollout(stub, o, r)
        text = open(stub.rollout_path, "rb").read()
    out["rollout_dump"] = dict(obs=obs, rewards=np.array([float(r) for r in rewards], dtype=np.float64),
                               text=np.frombuffer(text, dtype=np.uint8))


def main():
    out = {}
    only = set(sys.argv[1:])
    gens = dict(actions=gen_constants_and_actions, tracks=gen_tracks, obs_pack=gen_obs_pack, closed_loop=gen_closed_loop,
                scripted=gen_scripted, normalize=gen_normalize, gae=gen_gae, rollout_dump=gen_rollout_dump,
                reward_wrappers=gen_reward_wrappers, extra_physics=gen_extra_physics, pid_control=gen_pid_control, random_spawn=gen_random_spawn,
                dead_dynamics=gen_dead_dynamics)
    for key, fn in gens.items():          # `python gen_golden.py rollout_dump` regenerates only that group
        if not only or key in only:
            fn(out)
    for name, d in out.items():
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **d)
        print(f"{name:28s} {os.path.getsize(path) / 1024:8.1f} KiB")


if __name__ == "__main__":
    main()
