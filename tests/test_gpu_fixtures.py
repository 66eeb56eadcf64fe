"""Reference-generated fixtures replayed on the HIP path DIRECTLY (through dn_eval_kinematics, C ABI), not via the
oracle: rows A5-A9 of SURVEY.md section 8(a).

tests/golden/obs_pack.npz and script_*.npz were produced by importing the reference's own Python
(tests/golden/gen_golden.py): `_computeObs` on random kinematic states incl. gimbal-lock attitudes, clip edges and zero
angular velocity (PBDroneEnv.py:296-398), and `PBDroneEnv.step` / `reset` driven over scripted kinematic sequences --
gate passes, the last gate (+200/25), corridor and box exits (-10), truncation at max_steps, the reset quirks Q1-Q5
(PBDroneEnv.py:171-223, :444-607, :609-665, :678-786).  dn_eval_kinematics runs the same device functions as dn_step
(attitude_phase, observe_phase, rules_phase, report_phase) on a GIVEN post-physics state.

Bars: flags, waypoint index, step counters exact; observation 1e-5; reward 1e-5 relative + 1e-4 absolute (the state's
distances are float32: one ulp of d enters the reward 120x, PBDroneEnv.py:556); stored distances 1e-6.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _pkg():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: the HIP path has no CPU fallback")
    import drl_dronenavigation_amd as pkg
    return pkg


def _kin(pos, quat, vel, ang_v):
    k = np.concatenate([np.atleast_2d(pos), np.atleast_2d(quat), np.atleast_2d(vel), np.atleast_2d(ang_v)], axis=1)
    return torch.from_numpy(np.ascontiguousarray(k, dtype=np.float64)).to("cuda:0")


@pytest.mark.parametrize("track", ["circle", "race"])
def test_obs_pack_fixture_on_gpu(golden, track):
    """A5/A6: 600 random kinematic states per track -> the observation the reference's _computeObs produced."""
    pkg = _pkg()
    from drl_dronenavigation_amd import tracks
    g = golden("obs_pack")
    pos, quat, vel, ang_v = (g[f"{track}_{k}"] for k in ("pos", "quat", "vel", "ang_v"))
    dist, want = g[track + "_dist"], g[track + "_obs"]
    n = len(pos)
    sarg = -2.0 * (quat[:, 0] * quat[:, 2] - quat[:, 3] * quat[:, 1])
    locked = np.abs(sarg) >= 0.99999
    assert locked.sum() >= 20 and (sarg[locked] > 0).any() and (sarg[locked] < 0).any(), "the fixture must hold both gimbal-lock branches"
    assert (np.abs(ang_v).sum(1) == 0).sum() >= 10 and (np.abs(vel[:, 0]) > 3).any() and (np.abs(vel[:, 2]) > 1).any()
    t = tracks.circle(1, 4, 1) if track == "circle" else tracks.reaching()
    env = pkg.DroneVecEnv(t, n, normalize_obs=False, ground_contact=False, device="cuda:0")
    env.reset_tensor()
    st = env.get_state()
    st["d"] = dist.astype(np.float32)                         # _distance_to_target as the fixture set it (observation column 12)
    env.set_state(st)
    obs, rew, done, info = env.eval_kinematics_tensor(_kin(pos, quat, vel, ang_v))
    torch.cuda.synchronize()
    dn = done.cpu().numpy().astype(bool)
    got = np.where(dn[:, None], info["terminal_obs"].cpu().numpy(), obs.cpu().numpy())    # a finished drone's step observation
    assert dn.sum() > n // 2                                                               # is its terminal_observation
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    assert err.max() <= 1e-5, (err.max(), np.unravel_index(err.argmax(), err.shape))
    # the gimbal-lock rows: roll column exactly 0, pitch column +-1/2
    np.testing.assert_array_equal(got[locked, 3], 0.0)
    np.testing.assert_allclose(got[locked, 4], np.sign(sarg[locked]) * 0.5, atol=1e-7)
    print(f"obs_pack/{track}: max |obs err| = {err.max():.2e}, {locked.sum()} gimbal-lock rows, {dn.sum()} finished")
    env.close()


SCRIPTED = ["script_circle_follow", "script_circle_drift", "script_circle_trunc", "script_race_follow",
            "script_race_drift", "script_race_trunc", "script_up_follow"]


@pytest.mark.parametrize("name", SCRIPTED)
def test_scripted_fixture_on_gpu(golden, name):
    """A7-A9: the reference's reward / termination / truncation / bookkeeping over a scripted kinematic sequence, step by
    step, with the internal variables compared after every step."""
    pkg = _pkg()
    g = golden(name)
    env = pkg.DroneVecEnv(None, 1, target_points=g["waypoints"], initial_xyzs=g["spawn"], aviary_dim=g["dim"],
                          circle=bool(g["circle"]), max_steps=int(g["max_steps"]), normalize_obs=False,
                          ground_contact=False, device="cuda:0")
    env.reset_tensor()                                        # the script was recorded after reset(seed=0); reset()
    T = len(g["pos"])
    seen = dict(found=0, done200=0, crash=0, trunc=0)
    ep_len = 0
    ep_ret = 0.0
    idx_before = 0
    max_obs = max_rew = 0.0
    for t in range(T):
        obs, rew, done, info = env.eval_kinematics_tensor(_kin(g["pos"][t], g["quat"][t], g["vel"][t], g["ang_v"][t]))
        obs, rew, dn = obs.cpu().numpy()[0].copy(), float(rew.cpu().numpy()[0]), bool(done.cpu().numpy()[0])
        tl = bool(info["truncated"].cpu().numpy()[0])
        found = int(info["found_targets"].cpu().numpy()[0])
        term, trunc = bool(g["terminated"][t]), bool(g["truncated"][t])
        assert dn == (term or trunc), (name, t, "done")
        assert tl == (trunc and not term), (name, t, "TimeLimit.truncated")
        assert found == int(g["found_targets"][t]), (name, t, "found_targets")
        np.testing.assert_allclose(rew, g["reward"][t], rtol=1e-5, atol=1e-4, err_msg=f"{name} t={t}: reward")
        max_rew = max(max_rew, abs(rew - float(g["reward"][t])))
        ep_len += 1
        ep_ret += float(g["reward"][t])
        step_obs = info["terminal_obs"].cpu().numpy()[0] if dn else obs
        np.testing.assert_allclose(step_obs, g["obs"][t], rtol=0, atol=1e-5, err_msg=f"{name} t={t}: obs")
        max_obs = max(max_obs, np.abs(step_obs - g["obs"][t]).max())
        if dn:
            np.testing.assert_allclose(obs, g["reset_obs"][t], rtol=0, atol=1e-5, err_msg=f"{name} t={t}: reset obs (Q2)")
            assert int(info["ep_length"].cpu().numpy()[0]) == ep_len, (name, t, "episode l")
            np.testing.assert_allclose(info["ep_return"].cpu().numpy()[0], ep_ret, rtol=1e-5, atol=1e-3)
            ep_len, ep_ret = 0, 0.0
        seen["found"] += found > idx_before
        seen["done200"] += abs(float(g["reward"][t]) - 8.0) < 1e-12
        seen["crash"] += float(g["reward"][t]) == -10.0
        seen["trunc"] += tl
        st = env.get_state()[0]
        for k in ("idx", "steps", "just_found"):
            assert int(st[k]) == int(g["int_" + k][t]), (name, t, k)
        np.testing.assert_allclose(st["d"], g["int_d"][t], rtol=1e-6, atol=1e-6, err_msg=f"{name} t={t}: d")
        np.testing.assert_allclose(st["d_prev"], g["int_d_prev"][t], rtol=1e-6, atol=1e-6, err_msg=f"{name} t={t}: d_prev")
        np.testing.assert_allclose(st["cur_pos"], g["int_cur_pos"][t], rtol=1e-6, atol=1e-6, err_msg=f"{name} t={t}: _current_position (Q3)")
        np.testing.assert_allclose(st["prev_vel"], g["int_prev_vel"][t], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(st["prev_ang_v"], g["int_prev_ang_v"][t], rtol=1e-6, atol=1e-6)
        # current_vel / current_ang_v (quirk Q4) are the body velocities of the state
        np.testing.assert_allclose(st["vel"], g["int_cur_vel"][t], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(st["ang_v"], g["int_cur_ang_v"][t], rtol=1e-6, atol=1e-6)
        idx_before = int(st["idx"])
    assert seen["found"] > 0
    if name.endswith("follow") and "up" not in name:
        assert seen["done200"] >= 1, seen
    if name.endswith("drift"):
        assert seen["crash"] >= 1, seen
    if name.endswith("trunc"):
        assert seen["trunc"] >= 1, seen
    print(f"{name}: {T} steps, {seen}, max |obs err| = {max_obs:.2e}, max |reward err| = {max_rew:.2e}")
    env.close()
