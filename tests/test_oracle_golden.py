"""The CPU oracle against the golden vectors captured from the reference's own Python
(tests/golden/gen_golden.py).  CPU only.  Tolerances: float32 action chain bit-exact;
float64 quantities 1e-12 relative (numpy's BLAS-backed norm/dot may round differently from a
plain C sum by an ulp); flags, indices and step counters exact.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O

FP = C.POINTER(C.c_float)
DP = C.POINTER(C.c_double)


def cfg_from(g, **kw):
    return O.make_config(g["waypoints"], g["spawn"], g["dim"], circle=bool(g["circle"]),
                         max_steps=int(g["max_steps"]), ground_contact=False, **kw)


def test_constants(golden):
    g = golden("actions")
    k = np.zeros(16)
    O.lib().orc_constants(k.ctypes.data_as(DP))
    names = ["M", "KF", "KM", "IXX", "IYY", "IZZ", "PWM2RPM_SCALE", "PWM2RPM_CONST", "MIN_PWM", "MAX_PWM", "G",
             "PYB_TIMESTEP", "GRAVITY", "HOVER_RPM"]
    for i, n in enumerate(names):
        assert k[i] == float(g["const_" + n]), n
    assert int(g["const_PYB_STEPS_PER_CTRL"]) == 1
    lo, hi = C.c_float(), C.c_float()
    O.lib().orc_action_bounds(C.byref(lo), C.byref(hi))
    assert np.float32(lo.value) == g["a_low"] and np.float32(hi.value) == g["a_high"]
    # the values that reach the Bullet C-API are numpy float32 scalars
    assert list(g["api_scalar_types"]) == ["float32"]


def test_action_chain_bit_exact(golden):
    g = golden("actions")
    L = O.lib()
    acts = g["actions"]
    resc = np.zeros_like(acts)
    rpm = np.zeros_like(acts)
    forces = np.zeros(acts.shape, np.float32)
    zt = np.zeros(len(acts), np.float32)
    for i in range(len(acts)):
        L.orc_rescale_action(acts[i].ctypes.data_as(FP), resc[i].ctypes.data_as(FP))
        L.orc_preprocess_action(resc[i].ctypes.data_as(FP), rpm[i].ctypes.data_as(FP))
        z = C.c_float()
        L.orc_rotor_forces(rpm[i].ctypes.data_as(FP), forces[i].ctypes.data_as(FP), C.byref(z))
        zt[i] = z.value
    assert np.array_equal(resc.view(np.uint32), g["rescaled"].view(np.uint32))
    assert np.array_equal(rpm.view(np.uint32), g["rpm"].view(np.uint32))
    assert np.array_equal(forces.astype(np.float64), g["forces"])
    assert np.array_equal(zt.astype(np.float64), g["z_torque"])
    # regime check quoted in SURVEY 8(a) A1: only a in [0.08994, 0.09717] is unsaturated
    unsat = (g["rpm"] > 9440.31) & (g["rpm"] < 21666.44)
    a = acts[unsat]
    assert a.min() > 0.0899 and a.max() < 0.0972


@pytest.mark.parametrize("track", ["circle", "race"])
def test_obs_packing(golden, track):
    g = golden("obs_pack")
    gt = golden("traj_circle_uniform" if track == "circle" else "traj_race_uniform")
    cfg = cfg_from(gt)
    L = O.lib()
    e = O.OrcEnv()
    n = len(g[track + "_pos"])
    bad = 0
    for k in range(n):
        e.pos[:] = g[track + "_pos"][k]
        e.quat[:] = g[track + "_quat"][k]
        e.vel[:] = g[track + "_vel"][k]
        e.ang_v[:] = g[track + "_ang_v"][k]
        e.d = g[track + "_dist"][k]
        rpy = np.zeros(3)
        L.orc_euler_from_quat(np.ascontiguousarray(g[track + "_quat"][k]).ctypes.data_as(DP), rpy.ctypes.data_as(DP))
        e.rpy[:] = rpy
        obs = np.zeros(13, np.float32)
        L.orc_compute_obs(C.byref(cfg), C.byref(e), obs.ctypes.data_as(FP))
        ref = g[track + "_obs"][k]
        if not np.array_equal(obs.view(np.uint32), ref.view(np.uint32)):
            bad += 1
            np.testing.assert_allclose(obs, ref, rtol=2e-7, atol=1e-30)
    assert bad <= n // 100, bad          # float64 norm rounding can move a float32 by one ulp, rarely


CLOSED = ["traj_circle_uniform", "traj_circle_hover", "traj_race_uniform", "traj_race_mixed_norm", "traj_circle6_norm"]


@pytest.mark.parametrize("name", CLOSED)
def test_closed_loop_vec_env(golden, name):
    """Whole VecEnv trajectories (auto-reset, Monitor, optional obs normaliser) incl. Q1-Q5."""
    g = golden(name)
    T, n = g["actions"].shape[:2]
    cfg = cfg_from(g, normalize_obs=bool(g["normalize_obs"]))
    v = O.OracleVecEnv(cfg, n)
    obs0 = v.reset()
    np.testing.assert_allclose(obs0, g["reset_obs"], rtol=1e-6, atol=1e-7)
    n_done = 0
    for t in range(T):
        o = v.step(g["actions"][t])
        for k in ("done", "truncated", "terminated", "found_targets"):
            assert np.array_equal(o[k], g[k][t]), (name, t, k)
        dn = o["done"].astype(bool)
        n_done += dn.sum()
        assert np.array_equal(o["ep_len"][dn], g["ep_len"][t][dn])
        np.testing.assert_allclose(o["reward"], g["reward"][t], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(o["obs"], g["obs"][t], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(o["terminal_obs"][dn], g["terminal_obs"][t][dn], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(o["ep_ret"][dn], g["ep_ret"][t][dn], rtol=1e-5, atol=1e-5)
        e = v.envs
        for k in ("pos", "quat", "rpy", "vel", "ang_v", "cur_pos", "cur_vel", "cur_ang_v", "prev_vel", "prev_ang_v",
                  "d", "d_prev"):
            np.testing.assert_allclose(e[k], g["int_" + k][t], rtol=1e-11, atol=1e-13, err_msg=f"{name} t={t} {k}")
        for k in ("idx", "just_found", "is_done", "steps"):
            assert np.array_equal(e[k], g["int_" + k][t]), (name, t, k)
    assert n_done > 0
    if bool(g["normalize_obs"]):
        np.testing.assert_allclose(v.envs["rms_mean"], g["rms_mean"], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(v.envs["rms_var"], g["rms_var"], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(v.envs["rms_count"], g["rms_count"], rtol=1e-15)


SCRIPTED = ["script_circle_follow", "script_circle_drift", "script_circle_trunc", "script_race_follow",
            "script_race_drift", "script_race_trunc", "script_up_follow"]


@pytest.mark.parametrize("name", SCRIPTED)
def test_scripted_teacher_forced(golden, name):
    """Reward / termination / truncation / bookkeeping on chosen kinematic sequences (gate passes,
    final gate +200/25, corridor and box exits -10, truncation at max_steps, reset quirks)."""
    g = golden(name)
    L = O.lib()
    cfg = cfg_from(g)
    e = O.OrcEnv()
    obs = np.zeros(13, np.float32)
    L.orc_env_construct(C.byref(cfg), C.byref(e))
    L.orc_env_reset(C.byref(cfg), C.byref(e), obs.ctypes.data_as(FP))
    L.orc_env_reset(C.byref(cfg), C.byref(e), obs.ctypes.data_as(FP))
    T = len(g["pos"])
    seen = dict(found=0, done200=0, crash=0, trunc=0)
    for t in range(T):
        e.pos[:] = g["pos"][t]
        e.quat[:] = g["quat"][t]
        e.vel[:] = g["vel"][t]
        e.ang_v[:] = g["ang_v"][t]
        rpy = np.zeros(3)
        L.orc_euler_from_quat(np.ascontiguousarray(g["quat"][t]).ctypes.data_as(DP), rpy.ctypes.data_as(DP))
        e.rpy[:] = rpy
        L.orc_compute_obs(C.byref(cfg), C.byref(e), obs.ctypes.data_as(FP))
        idx_before = e.idx
        r = L.orc_compute_reward(C.byref(cfg), C.byref(e))
        term = L.orc_compute_terminated(C.byref(cfg), C.byref(e))
        trunc = L.orc_compute_truncated(C.byref(cfg), C.byref(e))
        found = e.idx
        if not term:
            L.orc_post_step(C.byref(cfg), C.byref(e))
        assert term == int(g["terminated"][t]), (name, t)
        assert trunc == int(g["truncated"][t]), (name, t)
        assert found == int(g["found_targets"][t]), (name, t)
        np.testing.assert_allclose(r, g["reward"][t], rtol=1e-9, atol=1e-11, err_msg=f"{name} t={t}")
        np.testing.assert_allclose(obs, g["obs"][t], rtol=2e-7, atol=1e-30)
        seen["found"] += found > idx_before
        seen["done200"] += (r == 8.0)
        seen["crash"] += (r == -10.0)
        seen["trunc"] += bool(trunc and not term)
        if term or trunc:
            L.orc_env_reset(C.byref(cfg), C.byref(e), obs.ctypes.data_as(FP))
            np.testing.assert_allclose(obs, g["reset_obs"][t], rtol=2e-7, atol=1e-30)
        for k in ("cur_pos", "cur_vel", "cur_ang_v", "prev_vel", "prev_ang_v"):
            np.testing.assert_allclose(np.array(getattr(e, k)), g["int_" + k][t], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(e.d, g["int_d"][t], rtol=1e-12)
        np.testing.assert_allclose(e.d_prev, g["int_d_prev"][t], rtol=1e-12)
        for k in ("idx", "just_found", "is_done", "steps"):
            assert getattr(e, k) == int(g["int_" + k][t]), (name, t, k)
    assert seen["found"] > 0
    if name.endswith("follow") and "up" not in name:
        assert seen["done200"] >= 1, seen
    if name.endswith("drift"):
        assert seen["crash"] >= 1, seen
    if name.endswith("trunc"):
        assert seen["trunc"] >= 1, seen


def test_normalize_observation(golden):
    g = golden("normalize")
    L = O.lib()
    e = O.OrcEnv()
    for i in range(13):
        e.rms_mean[i], e.rms_var[i] = 0.0, 1.0
    e.rms_count = 1e-4
    y = np.zeros(13)
    for x, ref in zip(g["x"], g["y"]):
        L.orc_normalize_obs(C.byref(e), np.ascontiguousarray(x).ctypes.data_as(FP), y.ctypes.data_as(DP))
        np.testing.assert_allclose(y, ref, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(np.array(e.rms_mean), g["mean"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(np.array(e.rms_var), g["var"], rtol=1e-12, atol=1e-15)
    assert e.rms_count == float(g["count"])


def test_gae(golden):
    g = golden("gae")
    adv, ret = O.gae(g["rewards"], g["values"], g["dones"], g["next_value"], g["next_done"],
                     float(g["gamma"]), float(g["gae_lambda"]))
    assert np.array_equal(adv.view(np.uint32), g["advantages"].view(np.uint32))
    assert np.array_equal(ret.view(np.uint32), g["returns"].view(np.uint32))


def test_reward_wrappers_match_reference(golden):
    """make_env's optional reward wrappers (--clip_rew, --norm_rew; PBDroneSimulator.py:191-194): the oracle against the
    reference's own NormalizeReward (normalize.py:100-147) driven over a scripted reward / done stream."""
    g = golden("reward_wrappers")
    L = O.lib()
    for tag, clip in (("norm", False), ("clip_norm", True)):
        cfg = O.make_config(np.zeros((1, 3)), np.zeros(3), np.array([-1, -1, 0, 1, 1, 1.0]), clip_rew=clip, norm_rew=True)
        env = O.OracleVecEnv(cfg, 1)
        e = env.envs.ctypes.data_as(C.POINTER(O.OrcEnv))
        ys = np.array([L.orc_reward_wrappers(C.byref(cfg), e, float(r), int(d)) for r, d in zip(g["rewards"], g["dones"])])
        np.testing.assert_allclose(ys, g[tag], rtol=1e-12, atol=1e-14)
        assert abs(env.envs["rr_mean"][0] - float(g[tag + "_mean"])) < 1e-12
        assert abs(env.envs["rr_var"][0] - float(g[tag + "_var"])) <= 1e-12 * float(g[tag + "_var"])
        assert env.envs["rr_count"][0] == float(g[tag + "_count"])
        assert abs(env.envs["rr_returns"][0] - float(g[tag + "_returns"])) < 1e-12


def test_extra_physics_terms_match_reference(golden):
    """N4: BaseAviary._groundEffect / _drag and the ActionType.RPM chain, as the reference's own methods compute them
    (extra_physics.npz).  The RPM chain is bit-exact; the force terms 1e-12 (np.dot / matmul in the stubbed Bullet
    getters round an ulp differently from a plain C sum)."""
    g = golden("extra_physics")
    L = O.lib()
    n = len(g["pos"])
    P = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(DP)   # noqa: E731
    for tag, is_f32 in (("f32", 1), ("f64", 0)):
        rpm = g["rpm_" + tag].astype(np.float64)
        hit = 0
        for k in range(n):
            out = np.zeros(4)
            L.orc_ground_effect(P(g["pos"][k]), P(g["quat"][k]), P(g["rpy"][k]), P(rpm[k]), is_f32, out.ctypes.data_as(DP))
            ref = g["gnd_" + tag][k]
            np.testing.assert_allclose(out, ref, rtol=1e-12, atol=0)
            hit += bool(ref.any())
            d3 = np.zeros(3)
            L.orc_drag(P(g["quat"][k]), P(g["vel"][k]), P(rpm[(k + 1) % n]), is_f32, d3.ctypes.data_as(DP))
            np.testing.assert_allclose(d3, g["drag_" + tag][k], rtol=1e-12, atol=0)
        assert 0.3 * n < hit < n            # both sides of the |roll|, |pitch| < pi/2 test are covered
    for k in range(n):
        a = np.ascontiguousarray(g["actions"][k])
        rpm, f, zt = np.zeros(4), np.zeros(4), C.c_double()
        L.orc_rpm_action(a.ctypes.data_as(FP), rpm.ctypes.data_as(DP), f.ctypes.data_as(DP), C.byref(zt))
        assert np.array_equal(rpm, g["rpm_f64"][k]) and np.array_equal(f, g["forces_f64"][k])
        assert zt.value == g["z_torque_f64"][k]
    # a ground-effect force is a second force on the same prop link: the env step adds it to the rotor thrust
    wp = np.array([[0.0, 1.0, 1.0], [-1.0, 0.0, 1.0]])
    dim = np.array([-2.0, -2.0, 0.0, 2.0, 2.0, 2.0])
    for physics, act in ((0, 0), (1, 0), (2, 0), (4, 0), (4, 1)):
        cfg = O.make_config(wp, np.array([1.0, 0.0, 0.1]), dim, circle=False, cylinder=False, ground_contact=False,
                            physics=physics, action_type=act, normalize_actions=act == 0)
        ve = O.OracleVecEnv(cfg, 1)
        ve.reset()
        a = np.full((1, 4), 0.0925 if act == 0 else 0.2, np.float32)
        for _ in range(3):
            assert not ve.step(a)["done"][0]
        envs = ve.envs
        assert np.all(envs["last_clipped_action"][0] > 9000.0)
        if physics == 0 and act == 0:
            base = envs["pos"][0].copy()
        elif act == 0:
            assert not np.array_equal(envs["pos"][0], base)         # the extra terms act
            np.testing.assert_allclose(envs["pos"][0], base, atol=1e-4)


def test_pid_family_action_types_match_reference(golden):
    """N4: ActionType.PID / VEL / ONE_D_RPM / ONE_D_PID -- BaseSingleAgentAviary._preprocessAction with the DSLPIDControl loop,
    as the reference's own methods compute them over a 400-step sequence with one persistent controller per action type
    (pid_control.npz, incl. a stretch inside getEulerFromQuaternion's gimbal-lock branch).  PID / ONE_D_PID agree to 1e-13,
    ONE_D_RPM exactly, VEL to 1e-7 (numpy's float32 norm of the direction vector sums in BLAS order)."""
    g = golden("pid_control")
    L = O.lib()
    assert float(g["SPEED_LIMIT"]) == 0.25 and abs(float(g["CTRL_TIMESTEP"]) - 1 / 240) < 1e-18
    P = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(DP)   # noqa: E731
    for name, code, tol in (("pid", 2, 1e-13), ("vel", 3, 1e-7), ("one_d_rpm", 4, 0.0), ("one_d_pid", 5, 1e-13)):
        st = np.zeros(9)
        unsat = 0
        for t in range(len(g["pos"])):
            rpm = np.zeros(4)
            a = np.ascontiguousarray(g["actions"][t])
            L.orc_pid_control(code, P(g["pos"][t]), P(g["quat"][t]), P(g["vel"][t]), a.ctypes.data_as(FP), st.ctypes.data_as(DP),
                              rpm.ctypes.data_as(DP))
            ref = g[name + "_rpm"][t]
            np.testing.assert_allclose(rpm, ref, rtol=tol, atol=0, err_msg=f"{name} t={t}")
            unsat += int(((ref > 9441) & (ref < 21666)).any())
            if code != 4:
                np.testing.assert_allclose(st[:3], g[name + "_integral_pos_e"][t], rtol=1e-12, atol=1e-15)
                np.testing.assert_allclose(st[3:6], g[name + "_last_rpy"][t], rtol=1e-12, atol=1e-15)
                np.testing.assert_allclose(st[6:], g[name + "_integral_rpy_e"][t], rtol=1e-6 if code == 3 else 1e-12, atol=1e-9 if code == 3 else 1e-12)
        assert unsat > 30, (name, unsat)            # the mixer is not pinned at the PWM limits all the time
    # the env step takes these action types: a hovering ONE_D_RPM drone, a PID drone flying towards a point
    wp = np.array([[0.0, 1.0, 1.0], [-1.0, 0.0, 1.0]])
    dim = np.array([-2.0, -2.0, 0.0, 2.0, 2.0, 2.0])
    for act, a in ((4, [0.0, 9, 9, 9]), (2, [1.0, 0.0, 1.2, 0]), (3, [1, 0, 0, 0.5]), (5, [0.3, 0, 0, 0])):
        cfg = O.make_config(wp, np.array([1.0, 0.0, 1.0]), dim, circle=False, cylinder=False, action_type=act, normalize_actions=False)
        ve = O.OracleVecEnv(cfg, 1)
        ve.reset()
        for _ in range(120):
            out = ve.step(np.array([a], np.float32))
            assert not out["done"][0]
        p = ve.envs["pos"][0]
        assert np.all(np.abs(p - [1.0, 0.0, 1.0]) < 0.5), (act, p)       # controlled flight, no run-away
        if act == 2:
            assert p[2] > 1.0                                            # climbing towards z = 1.2
        if act != 4:
            assert np.any(ve.envs["pid"][0] != 0.0)


def test_random_spawn_geometry_matches_reference(golden):
    """N4: PositionGenerator.generate_random_point_around_line (position_generator.py:121-152) with the draws supplied
    (random_spawn.npz): interpolation along the line, perpendicular offset through the cross product, clip to the aviary."""
    g = golden("random_spawn")
    L = O.lib()
    P = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(DP)   # noqa: E731
    assert float(g["max_distance"]) == 0.1
    clipped = 0
    for k in range(len(g["t"])):
        out = np.zeros(3)
        L.orc_point_around_line(P(g["frm"][k]), P(g["to"][k]), float(g["t"][k]), P(g["rv"][k]), -0.1 + (0.1 - -0.1) * float(g["u"][k]),
                                P(g["bounds"]), out.ctypes.data_as(DP))
        np.testing.assert_allclose(out, g["points"][k], rtol=1e-13, atol=1e-15)
        clipped += bool(((out == g["bounds"][:3]) | (out == g["bounds"][3:])).any())
    assert clipped >= 1
    # the Philox-keyed draw: inside the aviary, within max_distance of the line through two DISTINCT gates, reproducible
    wp = np.array([[0.0, 1.0, 1.0], [-1.0, 0.0, 1.0], [0.0, -1.0, 1.2], [1.0, 0.0, 0.8]])
    cfg = O.make_config(wp, np.array([1.0, 0.0, 1.0]), g["bounds"], circle=False, random_spawn=True, seed=7)
    seen = set()
    for env_id in range(300):
        a, b = np.zeros(3), np.zeros(3)
        L.orc_random_spawn(C.byref(cfg), env_id, 5, a.ctypes.data_as(DP))
        L.orc_random_spawn(C.byref(cfg), env_id, 5, b.ctypes.data_as(DP))
        assert np.array_equal(a, b) and np.all(a >= g["bounds"][:3]) and np.all(a <= g["bounds"][3:])
        dist = min(np.linalg.norm(np.cross(q - p_, a - p_)) / np.linalg.norm(q - p_) for i, p_ in enumerate(wp) for j, q in enumerate(wp) if i != j)
        assert dist <= 0.1 + 1e-12
        seen.add(tuple(np.round(a, 6)))
    assert len(seen) == 300
