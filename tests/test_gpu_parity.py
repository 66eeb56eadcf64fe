"""Parity of the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Bars (BASELINE.json north_star): float32 state within 1e-5 of the reference path on identical
seeds/actions; done / waypoint index bit-exact.  Tolerances are written next to each assertion.
All tests here need a real MI355X (`-m gpu`).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import oracle as O  # noqa: E402


def _gpu():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: the HIP path has no CPU fallback")
    import drl_dronenavigation_amd as pkg
    return pkg


def _tracks():
    from drl_dronenavigation_amd import tracks
    return tracks


def _bench_fused_waves(norm):
    """The fused-launch shape dn_create picks at 32 768 drones (two tiles per CU): what bench.py's headline times."""
    return 5 if norm else 4


def make_pair(track, n, *, f32_state, max_steps=4096, **kw):
    pkg = _gpu()
    env = pkg.DroneVecEnv(track, n, max_steps=max_steps, device="cuda:0", **kw)
    okw = {k: v for k, v in kw.items() if k in ("normalize_obs", "include_distance", "normalize_actions",
                                                "act_noise_sigma", "obs_noise_sigma", "seed", "env_id_offset",
                                                "ground_contact", "threshold", "cylinder", "clip_rew", "norm_rew", "random_spawn")}
    okw.setdefault("normalize_obs", True)
    okw["ground_contact"] = env.ground_contact     # DroneVecEnv's default is DN_GROUND_CONTACT_AUTO: the oracle gets what dn_create resolved
    cfg = O.make_config(track.targets(), track.initial_xyzs, track.aviary_dim, circle=track.is_circle,
                        max_steps=max_steps, f32_state=f32_state, **okw)
    return env, O.OracleVecEnv(cfg, n, threads=8)


def gpu_state_to_oracle(st, envs, step_count):
    """Teacher forcing: load the GPU's float32 state into the oracle's float64 variables."""
    for k in ("pos", "quat", "vel", "ang_v", "prev_vel", "prev_ang_v", "cur_pos", "d", "d_prev", "idx", "steps",
              "just_found", "ep_ret", "ep_len", "rms_mean", "rms_var", "rms_count", "rr_returns", "rr_mean", "rr_var",
              "rr_count", "pid"):
        envs[k] = st[k]
    envs["ep_ret"] = st["ep_ret"].astype(np.float64) + st["ep_ret_lo"].astype(np.float64)    # Monitor's running return: a float32 pair
    envs["last_clipped_action"] = st["last_rpm"]
    envs["cur_vel"] = st["vel"]
    envs["cur_ang_v"] = st["ang_v"]
    envs["is_done"] = 0
    envs["step_count"] = step_count


def actions_mixed(rng, n):
    """Even drones: bang-bang U(-1,1) (crash within tens of steps, BASELINE config 2's stream); odd drones:
    hover + noise, 0.0922 + 0.003 N(0,1) (long flights, truncation, gate passes); one step in eight the two
    regimes swap so that hovering drones get kicked."""
    bang = rng.uniform(-1, 1, (n, 4))
    hover = 0.0922 + 0.003 * rng.standard_normal((n, 4))
    swap = rng.random((n, 1)) < 0.125
    even = (np.arange(n) % 2 == 0)[:, None]
    a = np.where(even ^ swap, bang, hover)
    return a.astype(np.float32)


STATE_F32 = ("pos", "quat", "vel", "ang_v", "prev_vel", "prev_ang_v", "cur_pos", "d", "d_prev")


def compare_step(out, ref, tag, obs_atol=1e-5, rew_atol=1e-5):
    """rew_atol: 1e-5 teacher-forced.  Free-running comparisons (both sides keep their own float32 state) pass 1e-4:
    a stored distance may differ by one float32 ulp (1.2e-7) and the reward carries 3000 (d_prev - d) / 25 = 120x that."""
    obs, rew, done, info = out
    assert np.array_equal(done.cpu().numpy(), ref["done"]), f"{tag}: done"
    assert np.array_equal(info["truncated"].cpu().numpy(), ref["truncated"]), f"{tag}: TimeLimit.truncated"
    assert np.array_equal(info["found_targets"].cpu().numpy(), ref["found_targets"]), f"{tag}: waypoint index"
    k = obs.shape[1]                       # 12 columns when include_distance is off
    np.testing.assert_allclose(obs.cpu().numpy(), ref["obs"][:, :k], rtol=0, atol=obs_atol, err_msg=f"{tag}: obs")
    # reward carries 3000*(d_prev - d)/25: 1e-5 relative + 1e-5 absolute
    np.testing.assert_allclose(rew.cpu().numpy(), ref["reward"], rtol=1e-5, atol=rew_atol, err_msg=f"{tag}: reward")
    dn = ref["done"].astype(bool)
    if dn.any():
        np.testing.assert_allclose(info["terminal_obs"].cpu().numpy()[dn], ref["terminal_obs"][dn][:, :k], rtol=0,
                                   atol=obs_atol, err_msg=f"{tag}: terminal_observation")
        assert np.array_equal(info["ep_length"].cpu().numpy()[dn], ref["ep_len"][dn]), f"{tag}: episode l"
        np.testing.assert_allclose(info["ep_return"].cpu().numpy()[dn], ref["ep_ret"][dn], rtol=1e-5, atol=1e-4,
                                   err_msg=f"{tag}: episode r")
    return int(dn.sum())


@pytest.mark.parametrize("track_name,n,T,norm", [("circle4", 4096, 260, False), ("reaching", 4096, 260, False),
                                                 ("circle4", 1024, 200, True),
                                                 ("reaching", 32768, 130, True)])   # BASELINE configs[2] at size, as the reference runs it
def test_teacher_forced_vs_oracle(track_name, n, T, norm):
    """BASELINE config 2: every step starts both sides from the GPU's float32 state; the reference-grade
    float64 oracle then has to agree on state (1e-5), done and waypoint index (exact).  The last case is BASELINE
    configs[2] at its full size with the per-drone observation normaliser ON (the reference always wraps it,
    PBDroneSimulator.py:181), at the same 1e-5 bar."""
    track = _tracks().REGISTRY[track_name]()
    env, ora = make_pair(track, n, f32_state=False, max_steps=110, normalize_obs=norm)
    env.reset_tensor()
    ora.reset()
    rng = np.random.default_rng(1)
    dev = torch.device("cuda:0")
    n_done = n_found = n_crash = 0
    max_state_err = 0.0
    for t in range(T):
        st = env.get_state()
        gpu_state_to_oracle(st, ora.envs, env.step_count)
        a = actions_mixed(rng, n)
        out = env.step_tensor(torch.from_numpy(a).to(dev))
        torch.cuda.synchronize()
        ref = ora.step(a)
        n_done += compare_step(out, ref, f"{track_name} t={t}")
        n_found += int((ref["found_targets"] > 0).sum())
        n_crash += int((ref["reward"] == -10.0 / 1).sum())
        st2 = env.get_state()
        for k in STATE_F32:
            err = np.abs(st2[k].astype(np.float64) - ora.envs[k]).max()
            max_state_err = max(max_state_err, err)
            assert err <= 1e-5, f"{track_name} t={t}: state field {k} off by {err}"
        for k in ("idx", "steps", "just_found", "ep_len"):
            assert np.array_equal(st2[k], ora.envs[k]), f"{track_name} t={t}: {k}"
        if norm:
            # the Welford update is float64 on both sides; its input observation is float32 and may differ from
            # the oracle's by an ulp or two (float32 inverse trigonometry), which enters the mean divided by the count
            np.testing.assert_allclose(st2["rms_mean"], ora.envs["rms_mean"], rtol=0, atol=1e-6)
            np.testing.assert_allclose(st2["rms_var"], ora.envs["rms_var"], rtol=1e-6, atol=1e-6)
    assert n_done > n // 4, "the auto-reset path must be exercised"
    assert n_crash > n // 8 and n_done > n_crash, "both crashes (-10) and truncations must occur"
    print(f"{track_name}: {n_done} episodes, max |state err| = {max_state_err:.3e}")
    env.close()


@pytest.mark.parametrize("track_name", ["circle4", "reaching"])
def test_free_running_vs_f32_state_oracle(track_name):
    """Free-running trajectories (no teacher forcing): the oracle rounds its stored state to float32 after
    every step exactly as the HIP build stores it, so whole trajectories incl. auto-resets must coincide."""
    n, T = 2048, 300
    track = _tracks().REGISTRY[track_name]()
    env, ora = make_pair(track, n, f32_state=True, max_steps=150, normalize_obs=False)
    np.testing.assert_allclose(env.reset_tensor().cpu().numpy(), ora.reset(), rtol=0, atol=1e-6)
    rng = np.random.default_rng(2)
    dev = torch.device("cuda:0")
    n_done = 0
    for t in range(T):
        a = actions_mixed(rng, n)
        out = env.step_tensor(torch.from_numpy(a).to(dev))
        ref = ora.step(a)
        # free-running: the GPU's float64 arithmetic is not bit-identical to the oracle's (reciprocal multiplies,
        # polynomial sin/cos), so a stored float32 may differ by an ulp; the reward amplifies a distance
        # difference 120x (3000*(d_prev-d)/25), hence 1e-4 there.  Flags stay exact.
        n_done += compare_step(out, ref, f"{track_name} free t={t}", rew_atol=1e-4)
    st = env.get_state()
    for k in STATE_F32:
        np.testing.assert_allclose(st[k], ora.envs[k], rtol=0, atol=1e-5, err_msg=k)
    s = env.stats()
    assert s["episodes"] == n_done and s["env_steps"] == n * T
    env.close()


@pytest.mark.parametrize("name", ["traj_circle_uniform", "traj_circle_hover", "traj_race_uniform",
                                  "traj_race_mixed_norm", "traj_circle6_norm"])
def test_golden_fixtures_teacher_forced(golden, name):
    """The committed fixtures (reference Python driven closed-loop) replayed through the C ABI: the GPU is
    loaded with the fixture's internal state before every step and must reproduce the reference's outputs."""
    pkg = _gpu()
    g = golden(name)
    T, n = g["actions"].shape[:2]
    norm = bool(g["normalize_obs"])
    env = pkg.DroneVecEnv(None, n, target_points=g["waypoints"], initial_xyzs=g["spawn"], aviary_dim=g["dim"],
                          circle=bool(g["circle"]), max_steps=int(g["max_steps"]), normalize_obs=norm,
                          ground_contact=False, device="cuda:0")
    obs0 = env.reset()
    np.testing.assert_allclose(obs0, g["reset_obs"], rtol=0, atol=1e-6)
    dev = torch.device("cuda:0")
    for t in range(T):
        out = env.step_tensor(torch.from_numpy(g["actions"][t]).to(dev))
        obs, rew, done, info = out
        assert np.array_equal(done.cpu().numpy(), g["done"][t]), (name, t)
        assert np.array_equal(info["truncated"].cpu().numpy(), g["truncated"][t]), (name, t)
        assert np.array_equal(info["found_targets"].cpu().numpy(), g["found_targets"][t]), (name, t)
        # float32 state teacher-forced from float64 internals: 1e-5 on obs, 1e-4 on the 120x-amplified reward
        # (the reference turns ang_v into a UNIT vector, so where |ang_v| is tiny the float32 rounding of the
        #  teacher-forced input, 6e-8 absolute, is divided by |ang_v|: those three columns get 2e-7/|ang_v|)
        atol = np.full((n, 13), 1e-4 if norm else 1e-5)   # normalised obs are divided by a running std << 1
        if not norm:
            wn = np.linalg.norm(g["int_ang_v"][t], axis=1, keepdims=True)
            atol[:, 9:12] = np.maximum(1e-5, 2e-7 / np.maximum(wn, 1e-30))
        err = np.abs(obs.cpu().numpy().astype(np.float64) - g["obs"][t])
        assert (err <= atol).all(), (name, t, float((err - atol).max()))
        np.testing.assert_allclose(rew.cpu().numpy(), g["reward"][t], rtol=1e-4, atol=1e-4)
        # load the reference's own post-step state for the next step
        st = env.get_state()
        for k in ("pos", "quat", "vel", "ang_v", "prev_vel", "prev_ang_v", "d", "d_prev", "idx", "steps", "just_found"):
            st[k] = g["int_" + k][t]
        steps = g["int_steps"][t]
        st["cur_pos"] = np.where((steps > 0)[:, None], g["int_pos"][t], g["int_cur_pos"][t])
        env.set_state(st)
    env.close()


@pytest.mark.parametrize("n", [1, 12, 63, 64, 65, 200])
def test_ragged_sizes(n):
    """Ragged last tile (N not a multiple of the 64-lane wave), the single-drone case and BASELINE configs[0]'s
    num_envs = 12 on the 4-waypoint circle through the SB3 NumPy surface."""
    track = _tracks().circle(1, 4, 1)
    env, ora = make_pair(track, n, f32_state=True, max_steps=120, normalize_obs=True)
    np.testing.assert_allclose(env.reset(), ora.reset(), rtol=0, atol=1e-6)
    rng = np.random.default_rng(n)
    dev = torch.device("cuda:0")
    for t in range(170):
        a = actions_mixed(rng, n)
        compare_step(env.step_tensor(torch.from_numpy(a).to(dev)), ora.step(a), f"n={n} t={t}", rew_atol=1e-4,
                     obs_atol=1e-4)         # normalised observations: raw 1e-5 divided by a running std << 1
    env.close()


@pytest.mark.parametrize("kw", [dict(include_distance=False), dict(normalize_actions=False),
                                dict(cylinder=False, ground_contact=True), dict(threshold=0.05)])
def test_option_switches(kw):
    track = _tracks().up()
    n = 256
    env, ora = make_pair(track, n, f32_state=True, max_steps=40, normalize_obs=False, **kw)
    env.reset()
    ora.reset()
    rng = np.random.default_rng(5)
    dev = torch.device("cuda:0")
    done_total = 0
    for t in range(80):
        if kw.get("normalize_actions", True):
            a = actions_mixed(rng, n)
        else:   # physical thrusts in newtons
            a = rng.uniform(0.02, 0.16, (n, 4)).astype(np.float32)
        done_total += compare_step(env.step_tensor(torch.from_numpy(a).to(dev)), ora.step(a), f"{kw} t={t}",
                                   rew_atol=1e-4)
    assert done_total > 0
    env.close()


def test_many_waypoints_and_single_waypoint():
    tr = _tracks()
    wp = tr.dilate_targets(tr.reaching().waypoints, 8)
    assert len(wp) == 64
    for track in (tr.Track(wp, tr.reaching().initial_xyzs, tr.reaching().aviary_dim, False),
                  tr.Track([[0.2, 0.1, 1.0]], [[0, 0, 1.0]], (-2, -2, 0, 2, 2, 2), False)):
        env, ora = make_pair(track, 512, f32_state=True, max_steps=100, normalize_obs=False)
        env.reset()
        ora.reset()
        rng = np.random.default_rng(9)
        dev = torch.device("cuda:0")
        found = 0
        for t in range(150):
            a = (0.0922 + 0.002 * rng.standard_normal((512, 4))).astype(np.float32)
            out = env.step_tensor(torch.from_numpy(a).to(dev))
            ref = ora.step(a)
            compare_step(out, ref, f"W={len(track.waypoints)} t={t}", rew_atol=1e-4)
            found = max(found, int(ref["found_targets"].max()))
        assert found >= 1
        env.close()


@pytest.mark.parametrize("norm,noise", [(False, 0.0), (True, 0.01)])
def test_fused_multi_step_equals_single_steps(norm, noise):
    """dn_step_many (one launch, state in registers for K steps) must be bit-identical to K dn_step launches,
    and both must match the oracle."""
    pkg = _gpu()
    track = _tracks().reaching()
    n, K = 4096, 96
    kw = dict(normalize_obs=norm, max_steps=40, obs_noise_sigma=noise, act_noise_sigma=noise / 10, seed=77)
    env_a, ora = make_pair(track, n, f32_state=True, **kw)
    env_b = pkg.DroneVecEnv(track, n, device="cuda:0", **kw)
    env_a.reset()
    env_b.reset()
    ora.reset()
    rng = np.random.default_rng(11)
    acts = np.stack([actions_mixed(rng, n) for _ in range(K)])
    dev = torch.device("cuda:0")
    a_dev = torch.from_numpy(acts).to(dev)
    out = env_b.rollout_tensor(a_dev, want_terminal=True)
    n_done = 0
    for t in range(K):
        obs, rew, done, info = env_a.step_tensor(a_dev[t])
        assert torch.equal(obs, out["obs"][t]) and torch.equal(rew, out["reward"][t]), t
        assert torch.equal(done, out["done"][t]) and torch.equal(info["truncated"], out["truncated"][t]), t
        assert torch.equal(info["found_targets"], out["found_targets"][t]), t
        assert torch.equal(info["done_mask"], out["done_mask"][t]), t
        d = done.bool()
        assert torch.equal(info["terminal_obs"][d], out["terminal_obs"][t][d]), t
        assert torch.equal(info["ep_length"][d], out["ep_length"][t][d]), t
        n_done += compare_step((obs, rew, done, info), ora.step(acts[t]), f"fused t={t}", rew_atol=1e-4,
                               obs_atol=1e-4 if norm else 1e-5)
    assert n_done > n
    sa, sb = env_a.get_state(), env_b.get_state()
    for k in sa.dtype.names:
        assert np.ascontiguousarray(sa[k]).tobytes() == np.ascontiguousarray(sb[k]).tobytes(), k
    assert env_a.stats() == env_b.stats() and env_a.step_count == env_b.step_count == K
    env_a.close()
    env_b.close()


def test_noise_streams_match_oracle():
    """Config 5 (sim-to-real): Philox-keyed action/observation noise; sigma = 0 is the reference."""
    track = _tracks().reaching()
    n = 1024
    env, ora = make_pair(track, n, f32_state=True, max_steps=50, normalize_obs=False, act_noise_sigma=0.002,
                         obs_noise_sigma=0.01, seed=1234, env_id_offset=1 << 33)
    np.testing.assert_allclose(env.reset(), ora.reset(), rtol=0, atol=1e-6)
    rng = np.random.default_rng(3)
    dev = torch.device("cuda:0")
    for t in range(60):
        a = (0.0922 + 0.002 * rng.standard_normal((n, 4))).astype(np.float32)
        compare_step(env.step_tensor(torch.from_numpy(a).to(dev)), ora.step(a), f"noise t={t}", rew_atol=1e-4)
    env.close()


def test_noise_streams_do_not_repeat_when_the_step_counter_wraps_32_bits():
    """The Philox counter carries the whole 64-bit vector-step counter: across the 2^32 boundary (hours of fused
    stepping) the device still matches the oracle, and the draws after the wrap are not the draws of step 0, 1, ..."""
    track = _tracks().reaching()
    n = 256
    kw = dict(max_steps=50, normalize_obs=False, act_noise_sigma=0.002, obs_noise_sigma=0.01, seed=99)
    env, ora = make_pair(track, n, f32_state=True, **kw)
    low, _ = make_pair(track, n, f32_state=True, **kw)
    env.reset_tensor(); ora.reset(); low.reset_tensor()
    start = (1 << 32) - 3
    env.step_count = start
    ora.envs["step_count"] = start
    assert env.step_count == start
    dev = torch.device("cuda:0")
    a = np.full((n, 4), 0.0922, np.float32)
    at = torch.from_numpy(a).to(dev)
    acts = at[None].repeat(8, 1, 1).contiguous()
    for t in range(4):                                         # single steps across the boundary
        compare_step(env.step_tensor(at), ora.step(a), f"wrap t={t}", rew_atol=1e-4)
    out = env.rollout_tensor(acts)                             # and a fused launch beyond it
    for t in range(8):
        ref = ora.step(a)
        np.testing.assert_allclose(out["obs"][t].cpu().numpy(), ref["obs"], rtol=0, atol=1e-5)
    assert env.step_count == start + 12
    # the same drones from step 0: the first observation after the wrap must differ from the one at step 0
    o_low = low.step_tensor(at)[0].clone()
    twin, _ = make_pair(track, n, f32_state=True, **kw)
    twin.reset_tensor()
    twin.step_count = 1 << 32
    o_hi = twin.step_tensor(at)[0].clone()
    assert not torch.equal(o_low, o_hi) and float((o_low - o_hi).abs().max()) > 1e-3
    for e in (env, low, twin):
        e.close()


def test_float32_compute_mode_tolerance():
    """The float32-arithmetic build is a speed option, held to 5e-4 (teacher-forced) instead of 1e-5."""
    track = _tracks().circle(1, 4, 1)
    n = 2048
    env, ora = make_pair(track, n, f32_state=False, max_steps=60, normalize_obs=False, compute_dtype="float32")
    env.reset_tensor()
    ora.reset()
    rng = np.random.default_rng(4)
    dev = torch.device("cuda:0")
    mism = 0
    for t in range(60):
        gpu_state_to_oracle(env.get_state(), ora.envs, env.step_count)
        a = actions_mixed(rng, n)
        obs, rew, done, info = env.step_tensor(torch.from_numpy(a).to(dev))
        ref = ora.step(a)
        same = done.cpu().numpy() == ref["done"]
        mism += int((~same).sum())
        np.testing.assert_allclose(obs.cpu().numpy()[same], ref["obs"][same], rtol=0, atol=5e-4)
    assert mism <= n * 60 * 1e-4, f"{mism} done flags differ"
    env.close()


def test_sb3_step_surface_and_infos():
    """reset()/step() NumPy surface: shapes, dtypes and the info keys SB3 and the reference read."""
    track = _tracks().circle(1, 4, 1)
    n = 130
    env, ora = make_pair(track, n, f32_state=True, max_steps=90, normalize_obs=True)
    obs = env.reset()
    assert obs.shape == (n, 13) and obs.dtype == np.float32
    np.testing.assert_allclose(obs, ora.reset(), rtol=0, atol=1e-6)
    assert env.observation_space.shape == (13,) and env.action_space.shape == (4,)
    rng = np.random.default_rng(6)
    seen_trunc = seen_term = False
    for t in range(220):
        a = actions_mixed(rng, n)
        obs, rew, done, infos = env.step(a)
        ref = ora.step(a)
        assert obs.dtype == np.float32 and rew.dtype == np.float32 and done.dtype == bool and len(infos) == n
        assert np.array_equal(done, ref["done"].astype(bool))
        for i in range(n):
            assert infos[i]["found_targets"] == ref["found_targets"][i]
            assert ("terminal_observation" in infos[i]) == bool(done[i])
            if done[i]:
                assert infos[i]["TimeLimit.truncated"] == bool(ref["truncated"][i])
                assert infos[i]["episode"]["l"] == ref["ep_len"][i]
                assert abs(infos[i]["episode"]["r"] - ref["ep_ret"][i]) < 1e-3
                np.testing.assert_allclose(infos[i]["terminal_observation"], ref["terminal_obs"][i], atol=1e-4)
                seen_trunc |= bool(ref["truncated"][i])
                seen_term |= not bool(ref["truncated"][i])
    assert seen_trunc and seen_term
    assert env.get_attr("_current_target_index", [0, 1]) == list(ora.envs["idx"][:2])
    env.close()


def test_monitor_return_does_not_drift_over_long_episodes():
    """SB3's Monitor sums an episode's rewards in float64; the device keeps the running return as a float32 pair (hi in
    g4.w, lo in g6.w), so the reported info["episode"]["r"] is the float64 sum rounded ONCE to float32 -- over ~1500-step
    hovering episodes that used to drift by ~1e-3 when the sum was re-rounded every step."""
    track = _tracks().circle(1, 4, 1)
    n, T = 256, 1600
    env, ora = make_pair(track, n, f32_state=True, max_steps=1500, normalize_obs=False, cylinder=False)
    env.reset_tensor(); ora.reset()
    rng = np.random.default_rng(12)
    dev = torch.device("cuda:0")
    K = 64
    worst, seen = 0.0, 0
    for blk in range(T // K):
        # hover with a per-drone, per-step collective thrust offset (the same on the four rotors: no torque, so the drone
        # stays inside the aviary and the episode runs to its time limit)
        acts = np.repeat((0.092227 + 1e-5 * rng.uniform(-1, 1, (K, n, 1))).astype(np.float32), 4, axis=2)
        out = env.rollout_tensor(torch.from_numpy(acts).to(dev), want_terminal=True)
        torch.cuda.synchronize()
        for t in range(K):
            ref = ora.step(acts[t])
            dn = ref["done"].astype(bool)
            assert np.array_equal(out["done"][t].cpu().numpy().astype(bool), dn)
            if dn.any():
                got, want = out["ep_return"][t].cpu().numpy()[dn].astype(np.float64), ref["ep_ret"][dn].astype(np.float64)
                long_eps = ref["ep_len"][dn] > 1000
                if long_eps.any():
                    err = np.abs(got - want)[long_eps] / np.maximum(np.abs(want[long_eps]), 1.0)
                    worst, seen = max(worst, float(err.max())), seen + int(long_eps.sum())
    assert seen >= n // 2, seen
    assert worst <= 3e-6, worst               # the oracle's float32 output of its float64 sum vs ours: rewards agree to ~1e-6 relative
    env.close()


def test_sparse_info_mode_matches_full():
    """info_mode="sparse" (one persistent list of dicts, filled for finished drones only, cleared before the next step)
    carries exactly what info_mode="full" carries for the drones that finished, and nothing for the others."""
    pkg = _gpu()
    track = _tracks().circle(1, 4, 1)
    n = 300
    full = pkg.DroneVecEnv(track, n, device="cuda:0", max_steps=25, info_mode="full")
    sparse = pkg.DroneVecEnv(track, n, device="cuda:0", max_steps=25, info_mode="sparse")
    assert np.array_equal(full.reset(), sparse.reset())
    rng = np.random.default_rng(12)
    finished = 0
    for t in range(60):
        a = actions_mixed(rng, n)
        of, rf, df, inf_f = full.step(a)
        os_, rs, ds, inf_s = sparse.step(a)
        assert np.array_equal(of, os_) and np.array_equal(rf, rs) and np.array_equal(df, ds) and len(inf_s) == n
        for i in range(n):
            if df[i]:
                finished += 1
                assert set(inf_s[i]) == {"found_targets", "terminal_observation", "TimeLimit.truncated", "episode"}
                assert inf_s[i]["found_targets"] == inf_f[i]["found_targets"]
                assert inf_s[i]["TimeLimit.truncated"] == inf_f[i]["TimeLimit.truncated"]
                assert np.array_equal(inf_s[i]["terminal_observation"], inf_f[i]["terminal_observation"])
                assert inf_s[i]["episode"]["r"] == inf_f[i]["episode"]["r"] and inf_s[i]["episode"]["l"] == inf_f[i]["episode"]["l"]
            else:
                assert inf_s[i] == {}, (t, i, inf_s[i])          # last step's entries are gone
    assert finished > n
    full.close(); sparse.close()


def test_step_with_reused_host_buffers_and_a_mass_finish():
    """fresh_arrays=False hands out views of two alternating pinned mirrors: the arrays of step t must still hold step t's values after
    step t + 1 returned (SB3 reads `_last_obs` once more then), and equal what the default (fresh copies) returns.  max_steps is
    small, so the time limit ends every drone's episode in the same step: more finished drones than the first host copy carries
    packed records for (dn_pack_done's second copy)."""
    pkg = _gpu()
    track = _tracks().reaching()
    n = 5000                                              # the first copy carries 312 records
    fresh = pkg.DroneVecEnv(track, n, device="cuda:0", max_steps=9)
    reuse = pkg.DroneVecEnv(track, n, device="cuda:0", max_steps=9, fresh_arrays=False)
    assert fresh._pack_prefix < n and np.array_equal(fresh.reset(), reuse.reset())
    rng = np.random.default_rng(3)
    held = None
    mass = 0
    for t in range(40):
        a = (0.0922 + 0.002 * rng.standard_normal((n, 4))).astype(np.float32)      # hover: nobody crashes before the time limit
        of, rf, df, inf_f = fresh.step(a)
        orr, rr, dr, inf_r = reuse.step(a)
        assert np.array_equal(of, orr) and np.array_equal(rf, rr) and np.array_equal(df, dr)
        if held is not None:                              # step t - 1's arrays, untouched by this step
            assert np.array_equal(held[0], held[1]) and np.array_equal(held[2], held[3])
        held = (orr, of.copy(), rr, rf.copy())
        k = int(df.sum())
        mass = max(mass, k)
        for i in np.flatnonzero(df):
            assert inf_r[i]["episode"] == {**inf_f[i]["episode"], "t": inf_r[i]["episode"]["t"]}
            assert inf_r[i]["TimeLimit.truncated"] == inf_f[i]["TimeLimit.truncated"] and inf_r[i]["found_targets"] == inf_f[i]["found_targets"]
            assert np.array_equal(inf_r[i]["terminal_observation"], inf_f[i]["terminal_observation"])
        assert np.array_equal(reuse.done_indices(), np.flatnonzero(df))
    assert mass > fresh._pack_prefix                      # the second copy ran
    fresh.close(); reuse.close()


def test_full_size_properties():
    """BASELINE size (32768 drones, race track): size-independent properties -- determinism, unit
    quaternions, reset rows, the done ballot words vs the byte flags vs the compacted index list, and the
    checksum of checksums  sum(episode lengths) + sum(running lengths) == N * steps."""
    pkg = _gpu()
    track = _tracks().reaching()
    n, T = 32768, 200
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(7)
    acts = torch.where(torch.rand(T, n, 1, generator=g) < 0.5, torch.rand(T, n, 4, generator=g) * 2 - 1,
                       0.0922 + 0.003 * torch.randn(T, n, 4, generator=g)).to(dev)
    runs = []
    for rep in range(2):
        env = pkg.DroneVecEnv(track, n, normalize_obs=False, max_steps=64, device=dev)
        env.reset_tensor()
        sum_len = 0
        rew_sum = torch.zeros((), dtype=torch.float64, device=dev)
        for t in range(T):
            obs, rew, done, info = env.step_tensor(acts[t])
            rew_sum += rew.double().sum()
            d = done.bool()
            sum_len += int(info["ep_length"][d].sum())
            if t % 50 == 7:
                words = info["done_mask"].cpu().numpy().view(np.uint64)
                bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:n].astype(bool)
                assert np.array_equal(bits, d.cpu().numpy())
                idx = env.done_indices()
                assert np.array_equal(idx, np.nonzero(bits)[0])
                spawn_obs = (track.initial_xyzs.ravel() / track.aviary_dim[3:]).astype(np.float32)
                assert np.array_equal(obs[d][:, :3].cpu().numpy(), np.broadcast_to(spawn_obs, (int(d.sum()), 3)))
        st = env.get_state()
        assert sum_len + int(st["ep_len"].sum()) == n * T
        qn = np.linalg.norm(st["quat"].astype(np.float64), axis=1)
        assert np.abs(qn - 1).max() < 1e-6
        s = env.stats()
        assert s["sum_ep_len"] == sum_len and s["env_steps"] == n * T
        runs.append((st.copy(), float(rew_sum), s))
        env.close()
    assert runs[0][1] == runs[1][1] and runs[0][2] == runs[1][2]
    for k in runs[0][0].dtype.names:
        assert np.ascontiguousarray(runs[0][0][k]).tobytes() == np.ascontiguousarray(runs[1][0][k]).tobytes(), k


def test_gae_kernel_matches_reference_recursion(golden):
    pkg = _gpu()
    g = golden("gae")
    dev = torch.device("cuda:0")
    adv, ret = pkg.gae(torch.from_numpy(g["rewards"]).to(dev), torch.from_numpy(g["values"]).to(dev),
                       torch.from_numpy(g["dones"]).to(dev), torch.from_numpy(g["next_value"]).to(dev),
                       torch.from_numpy(g["next_done"]).to(dev), float(g["gamma"]), float(g["gae_lambda"]))
    assert np.array_equal(adv.cpu().numpy().view(np.uint32), g["advantages"].view(np.uint32))
    assert np.array_equal(ret.cpu().numpy().view(np.uint32), g["returns"].view(np.uint32))
    rng = np.random.default_rng(8)
    T, N = 64, 5000
    r, v = rng.standard_normal((T, N)).astype(np.float32), rng.standard_normal((T, N)).astype(np.float32)
    d, ld = (rng.random((T, N)) < 0.05).astype(np.uint8), (rng.random(N) < 0.05).astype(np.uint8)
    lv = rng.standard_normal(N).astype(np.float32)
    a_ref, r_ref = O.gae(r, v, d, lv, ld, 0.99, 0.95)
    adv, ret = pkg.gae(*(torch.from_numpy(x).to(dev) for x in (r, v, d, lv, ld)), 0.99, 0.95)
    assert np.array_equal(adv.cpu().numpy().view(np.uint32), a_ref.view(np.uint32))
    assert np.array_equal(ret.cpu().numpy().view(np.uint32), r_ref.view(np.uint32))


def test_errors_are_loud():
    pkg = _gpu()
    tr = _tracks()
    with pytest.raises(pkg.DroneNavError):
        pkg.DroneVecEnv(tr.circle(1, 4, 1), 0, device="cuda:0")
    with pytest.raises(ValueError):
        pkg.DroneVecEnv(tr.Track(np.zeros((65, 3)), [[0, 0, 1]], (-1, -1, 0, 1, 1, 1)), 4, device="cuda:0")
    env = pkg.DroneVecEnv(tr.circle(1, 4, 1), 8, device="cuda:0")
    with pytest.raises(ValueError):
        env.step_tensor(torch.zeros(8, 4))
    with pytest.raises(pkg.DroneNavError):
        st = env.get_state()
        st["idx"] = 99
        env.set_state(st)
    env.close()


def test_rollout_collector_matches_oracle_replay():
    """collector.RolloutCollector (policy -> dn_step x n_steps -> dn_gae, SB3 truncation bootstrap) against an
    oracle replay of the actions it took: rewards (incl. gamma*V(terminal_observation) on TimeLimit truncation),
    episode-start flags and advantages."""
    pkg = _gpu()
    from drl_dronenavigation_amd.collector import RolloutCollector
    track = _tracks().circle(1, 4, 1)
    n, T, gamma, lam = 1024, 48, 0.99, 0.95
    env, ora = make_pair(track, n, f32_state=True, max_steps=30, normalize_obs=False)
    dev = env.device
    g = torch.Generator(device="cpu").manual_seed(3)
    w_pi = (torch.randn(13, 4, generator=g) * 0.01).to(dev)
    w_v = torch.linspace(-1, 1, 13).to(dev)
    noise = (0.003 * torch.randn(T * 3, n, 4, generator=g)).to(dev)
    calls = [0]

    def policy(obs):
        hover = (torch.arange(n, device=dev) % 2 == 1)[:, None]
        a = torch.where(hover, 0.0922 + noise[calls[0] % len(noise)], torch.tanh(obs @ w_pi * 50.0) * 1.5)
        calls[0] += 1
        return a, obs @ w_v, torch.zeros(n, device=dev)

    col = RolloutCollector(env, policy, T, value_fn=lambda o: o @ w_v, gamma=gamma, gae_lambda=lam)
    ora.reset()
    last_done = np.ones(n, np.uint8)
    n_trunc = 0
    for it in range(2):
        out = col.collect()
        acts = out["actions"].cpu().numpy()
        assert np.abs(acts).max() > 1.0, "the policy must exceed the action space so the clip matters"
        rew = np.zeros((T, n), np.float32)
        for t in range(T):
            assert np.array_equal(out["episode_starts"][t].cpu().numpy(), last_done), (it, t)
            ref = ora.step(np.clip(acts[t], -1, 1))
            boot = gamma * (ref["terminal_obs"] @ w_v.cpu().numpy()) * ref["truncated"]
            rew[t] = ref["reward"] + boot.astype(np.float32)
            n_trunc += int(ref["truncated"].sum())
            last_done = ref["done"]
        np.testing.assert_allclose(out["rewards"].cpu().numpy(), rew, rtol=1e-5, atol=1e-4)
        a_ref, r_ref = O.gae(out["rewards"].cpu().numpy(), out["values"].cpu().numpy(),
                             out["episode_starts"].cpu().numpy(), out["last_values"].cpu().numpy(),
                             out["last_dones"].cpu().numpy(), gamma, lam)
        assert np.array_equal(out["advantages"].cpu().numpy().view(np.uint32), a_ref.view(np.uint32))
        assert np.array_equal(out["returns"].cpu().numpy().view(np.uint32), r_ref.view(np.uint32))
    assert n_trunc > 0, "TimeLimit truncations must occur so the bootstrap is exercised"
    assert col.num_timesteps == 2 * T * n
    env.close()


def test_action_chain_bit_exact_vs_reference_golden(golden):
    """A1-A3 through dn_preprocess_action against the vectors captured from the reference's own Python
    (rescale_action, _preprocessAction, the forces/torque handed to Bullet): bit-exact, float32."""
    pkg = _gpu()
    g = golden("actions")
    dev = torch.device("cuda:0")
    acts = np.ascontiguousarray(g["actions"], np.float32)
    rpm, forces, zt = pkg.preprocess_action(torch.from_numpy(acts).to(dev), normalize_actions=True)
    assert np.array_equal(rpm.cpu().numpy().view(np.uint32), g["rpm"].view(np.uint32))
    assert np.array_equal(forces.cpu().numpy().astype(np.float64), g["forces"])
    assert np.array_equal(zt.cpu().numpy().astype(np.float64), g["z_torque"])
    # and against the oracle on a dense random + edge sweep, both action modes (incl. out-of-range and huge inputs)
    rng = np.random.default_rng(12)
    a = np.concatenate([rng.uniform(-1, 1, (200000, 4)), rng.uniform(0.0899, 0.0972, (200000, 4)),
                        rng.uniform(0.02, 0.16, (100000, 4)),
                        np.array([[-1, 1, 0, -0.0], [2, -2, 1e30, -1e30], [3e38, -3e38, 1e-40, 0.0922]])]).astype(np.float32)
    import ctypes as C
    FP = C.POINTER(C.c_float)
    L = O.lib()
    for norm in (True, False):
        rpm, forces, zt = pkg.preprocess_action(torch.from_numpy(a).to(dev), normalize_actions=norm)
        rpm_ref = np.zeros_like(a)
        f_ref = np.zeros_like(a)
        z_ref = np.zeros(len(a), np.float32)
        sel = np.r_[rng.integers(0, len(a) - 3, 20000), np.arange(len(a) - 3, len(a))]
        for i in sel:
            src = a[i].copy()
            if norm:
                r = np.zeros(4, np.float32)
                L.orc_rescale_action(src.ctypes.data_as(FP), r.ctypes.data_as(FP))
                src = r
            L.orc_preprocess_action(src.ctypes.data_as(FP), rpm_ref[i].ctypes.data_as(FP))
            z = C.c_float()
            L.orc_rotor_forces(rpm_ref[i].ctypes.data_as(FP), f_ref[i].ctypes.data_as(FP), C.byref(z))
            z_ref[i] = z.value
        assert np.array_equal(rpm.cpu().numpy()[sel].view(np.uint32), rpm_ref[sel].view(np.uint32)), norm
        assert np.array_equal(forces.cpu().numpy()[sel].view(np.uint32), f_ref[sel].view(np.uint32)), norm
        assert np.array_equal(zt.cpu().numpy()[sel].view(np.uint32), z_ref[sel].view(np.uint32)), norm


@pytest.mark.parametrize("norm,noise", [(False, 0.0), (True, 0.0), (True, 0.01), (False, 0.02)])
def test_kernel_shapes_are_bit_identical(norm, noise, monkeypatch):
    """The two-wave kernels (flight wave + report wave, messages through LDS) and the one-wave kernels (same
    phases, messages in registers) must agree bit for bit: state, outputs, statistics; single steps and fused."""
    pkg = _gpu()
    track = _tracks().reaching()
    n, K = 1000, 70                       # ragged last tile on purpose
    kw = dict(normalize_obs=norm, max_steps=30, obs_noise_sigma=noise, act_noise_sigma=noise / 10, seed=5)
    envs = {}
    rp = ("8",) if norm and noise == 0.0 else ()
    for shape in ("1", "2", "3", "4") + (("5",) if norm else ()) + rp:
        monkeypatch.setenv("DN_WAVES", shape)
        envs[shape] = pkg.DroneVecEnv(track, n, device="cuda:0", **kw)
        envs[shape].reset()
        # three waves: flight / report / aux; four: linear + rules / angular + attitude / observation / thrust + report (fused launches);
        # five: the normaliser and the observation rows on a wave of their own; eight: the role-pipelined kernel (normaliser on, no noise)
        assert envs[shape].kernel_waves(fused=True) == int(shape)
    monkeypatch.delenv("DN_WAVES")
    rng = np.random.default_rng(21)
    dev = torch.device("cuda:0")
    acts = torch.from_numpy(np.stack([actions_mixed(rng, n) for _ in range(K)])).to(dev)
    outs = {}
    for shape, env in envs.items():
        first = [env.step_tensor(acts[t]) for t in range(6)]       # single-step launches ...
        first = [(o.clone(), r.clone(), d.clone(), {k: v.clone() for k, v in i.items()}) for o, r, d, i in first[-1:]]
        rest = env.rollout_tensor(acts[6:].contiguous(), want_terminal=True)   # ... then one fused launch (n % 4 == 0)
        outs[shape] = (first, rest, env.get_state(), env.stats())
    for other in outs:                    # every shape that was built, the normaliser's own wave shapes included
        if other != "1":
            _assert_same_rollout(outs["1"], outs[other], n)
    for env in envs.values():
        env.close()


@pytest.mark.parametrize("norm,noise,n", [(False, 0.0, 1000), (True, 0.0, 4096), (True, 0.01, 200), (False, 0.02, 32768)])
def test_three_wave_single_step_is_bit_identical(norm, noise, n, monkeypatch):
    """dn_step on three waves cut by dependency (dn_step_pqx_kernel: thrust + rewards | position + rules | attitude +
    observation) against the one-wave kernel: every output of every step and the final state, bit for bit, over 150
    closed-loop steps with crashes, gate passes, truncations and resets, ragged last tile included."""
    pkg = _gpu()
    monkeypatch.delenv("DN_WAVES", raising=False)
    track = _tracks().reaching()
    kw = dict(normalize_obs=norm, max_steps=40, obs_noise_sigma=noise, act_noise_sigma=noise / 10, seed=11)
    envs = {}
    for shape in ("1", "3"):
        monkeypatch.setenv("DN_WAVES_SINGLE", shape)
        envs[shape] = pkg.DroneVecEnv(track, n, device="cuda:0", **kw)
        assert envs[shape].kernel_waves(fused=False) == int(shape)
    monkeypatch.delenv("DN_WAVES_SINGLE")
    assert torch.equal(envs["1"].reset_tensor(), envs["3"].reset_tensor())
    rng = np.random.default_rng(17)
    dev = torch.device("cuda:0")
    n_done = 0
    for t in range(150):
        a = torch.from_numpy(actions_mixed(rng, n)).to(dev)
        o1, r1, d1, i1 = envs["1"].step_tensor(a)
        o3, r3, d3, i3 = envs["3"].step_tensor(a)
        assert torch.equal(o1, o3) and torch.equal(r1, r3) and torch.equal(d1, d3), t
        dn = d1.bool()
        for k in ("truncated", "found_targets", "done_mask"):
            assert torch.equal(i1[k], i3[k]), (t, k)
        for k in ("terminal_obs", "ep_return", "ep_length"):
            assert torch.equal(i1[k][dn], i3[k][dn]), (t, k)
        n_done += int(dn.sum())
    assert n_done > n
    s1, s3 = envs["1"].get_state(), envs["3"].get_state()
    for k in s1.dtype.names:
        assert np.ascontiguousarray(s1[k]).tobytes() == np.ascontiguousarray(s3[k]).tobytes(), k
    assert envs["1"].stats() == envs["3"].stats() and envs["1"].step_count == envs["3"].step_count == 150
    for e in envs.values():
        e.close()


def _assert_same_rollout(o1, o2, n):
    (f1, r1, s1, st1), (f2, r2, s2, st2) = o1, o2
    for a, b in zip(f1, f2):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
        for k in a[3]:
            assert torch.equal(a[3][k], b[3][k]), k
    for k in r1:
        if k in ("terminal_obs", "ep_return", "ep_length"):
            d = r1["done"].bool()
            assert torch.equal(r1[k][d], r2[k][d]), k
        else:
            assert torch.equal(r1[k], r2[k]), k
    for k in s1.dtype.names:
        assert np.ascontiguousarray(s1[k]).tobytes() == np.ascontiguousarray(s2[k]).tobytes(), k
    assert st1 == st2 and st1["episodes"] > n


def test_graph_replayed_rollouts_equal_eager_rollouts():
    """RolloutCollector(use_graph=True) replays the captured rollout (policy kernels + dn_step + GAE) from a
    hipGraph; with a deterministic policy it must reproduce the eager collector bit for bit, rollout after rollout
    -- including the Philox noise streams, whose step counter lives on the device and keeps advancing under replay."""
    pkg = _gpu()
    from drl_dronenavigation_amd.collector import RolloutCollector
    track = _tracks().reaching()
    n, T = 2048, 16
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = pkg.MlpActorCritic().to(dev)
    with torch.no_grad():
        net.action_net.bias.fill_(0.0922)          # hover, so that flights last and gates get passed

    def policy(obs):
        a, v, lp = net(obs, deterministic=True)
        return a, v, lp

    outs = []
    for use_graph in (False, True):
        env = pkg.DroneVecEnv(track, n, normalize_obs=True, max_steps=40, act_noise_sigma=0.003, obs_noise_sigma=0.01,
                              seed=9, device=dev)
        col = RolloutCollector(env, policy, T, value_fn=net.predict_values, use_graph=use_graph)
        runs = []
        for it in range(4):
            out = col.collect()
            torch.cuda.synchronize()
            runs.append({k: v.clone() for k, v in out.items()})
        assert (col._graph is not None) == use_graph
        assert env.step_count == 4 * T
        outs.append((runs, env.get_state(), env.stats()))
        env.close()
    (r_e, s_e, st_e), (r_g, s_g, st_g) = outs
    for it in range(4):
        for k in r_e[it]:
            assert torch.equal(r_e[it][k], r_g[it][k]), (it, k)
    for k in s_e.dtype.names:
        assert np.ascontiguousarray(s_e[k]).tobytes() == np.ascontiguousarray(s_g[k]).tobytes(), k
    assert st_e == st_g and st_e["episodes"] > 0 and st_e["env_steps"] == 4 * T * n
    assert float(r_e[3]["rewards"].abs().sum()) > 0


def _mlp_reference(layers, x):
    """The fused kernel's arithmetic spelled out in torch: bf16 weights and activations, float32 accumulation and bias,
    tanh in float32, float32 head output."""
    from drl_dronenavigation_amd.policy_mfma import TANH_PRESCALE as c      # folded into the hidden layers before bf16
    h = x.to(torch.bfloat16).float()
    for k, (w, b) in enumerate(layers):
        if k < len(layers) - 1:
            z = h @ (w.float() * c).to(torch.bfloat16).float().t() + b.float() * c
            h = torch.tanh(z / c).to(torch.bfloat16).float()
        else:
            h = h @ w.to(torch.bfloat16).float().t() + b.float()
    return h


@pytest.mark.parametrize("shape", ["1", "4", "8"])
@pytest.mark.parametrize("n", [32, 1000, 32768])
def test_fused_mfma_mlp_matches_torch(n, shape, monkeypatch):
    """dn_mlp_forward (one wavefront per 32 drones, activations in registers, K-permuted bf16 weight fragments) against
    the same network in torch: asymmetric random weights catch any row/column/K-order slip; tolerance is bf16-level."""
    pkg = _gpu()
    from drl_dronenavigation_amd import policy_mfma as pm
    monkeypatch.setenv("DN_MLP_SHAPE", shape)     # 1: weights from L2 per wave; 4: through LDS; 8: split-K wave pairs
    dev = torch.device("cuda:0")
    torch.manual_seed(n)
    net = pkg.MlpActorCritic().to(dev)
    with torch.no_grad():
        for p_ in net.parameters():
            p_.add_(0.05 * torch.randn_like(p_))          # non-zero biases, not-quite-orthogonal weights
    obs = (torch.rand(n, 13, device=dev) * 2 - 1)
    lin = lambda seq: [l for l in seq if isinstance(l, torch.nn.Linear)]      # noqa: E731
    pi = [(l.weight.detach(), l.bias.detach()) for l in lin(net.pi)] + [(net.action_net.weight.detach(), net.action_net.bias.detach())]
    vf = [(l.weight.detach(), l.bias.detach()) for l in lin(net.vf)] + [(net.value_net.weight.detach(), net.value_net.bias.detach())]
    pol = pm.FusedMlpPolicy(net, n, dev)
    mean, value = pm.mlp_forward([pol.pi, pol.vf], obs)
    torch.cuda.synchronize()
    ref_m, ref_v = _mlp_reference(pi, obs), _mlp_reference(vf, obs)
    # the emulation differs only in summation order and the 1e-7 tanh; where that flips the bf16 rounding of a hidden
    # activation the outputs move by a bf16 ulp of a weight-activation product: 1e-2 worst case, 3e-4 on average
    for got, ref in ((mean, ref_m), (value, ref_v)):
        err = (got - ref).abs()
        assert float(err.max()) < 1e-2 and float(err.mean()) < 3e-4, (float(err.max()), float(err.mean()))
    # and stays at bf16 distance from the float32 network SB3 would run
    with torch.no_grad():
        f32_m = net.action_net(net.pi(obs))
    assert float((mean - f32_m).abs().max()) < 5e-2
    a, v, lp = pol(obs, deterministic=True)
    assert torch.equal(a, mean) and v.shape == (n,) and lp.shape == (n,)
    # single-network launch (predict_values) gives the same values; a masked launch evaluates exactly the tiles that
    # hold a flagged drone and zeroes the others
    assert torch.equal(pol.predict_values(obs).clone(), value.squeeze(-1))
    mask = torch.zeros(n, dtype=torch.uint8, device=dev)
    mask[::97] = 1
    mv = pol.predict_values(obs, row_mask=mask).clone()
    tile_has = torch.zeros((n + 31) // 32 * 32, dtype=torch.bool, device=dev)
    tile_has[:n] = mask.bool()
    tile_has = tile_has.view(-1, 32).any(1).repeat_interleave(32)[:n]
    assert torch.equal(mv[tile_has], value.squeeze(-1)[tile_has]) and float(mv[~tile_has].abs().sum()) == 0.0


@pytest.mark.parametrize("n", [32, 1000, 32768])
def test_fp32_grade_mlp_matches_the_float32_torch_network(n):
    """dn_mlp_forward with grade = 1 (split-bf16 operands, three MFMAs per product, dn_mlp_x3_kernel) against the network
    the reference actually runs: SB3's float32 ActorCriticPolicy MLPs (PBDroneSimulator.py:251-286).  Bar: 1e-4 on the
    action mean and the value (the band of non-saturated actions is 0.0073 wide, PBDroneEnv.py:949-971); the bf16 speed
    grade sits around 1e-3 on the same inputs."""
    pkg = _gpu()
    from drl_dronenavigation_amd import policy_mfma as pm
    dev = torch.device("cuda:0")
    torch.manual_seed(n + 1)
    net = pkg.MlpActorCritic().to(dev)
    with torch.no_grad():
        for p_ in net.parameters():
            p_.add_(0.05 * torch.randn_like(p_))
    obs = (torch.rand(n, 13, device=dev) * 2 - 1) * torch.tensor([1, 1, 1, 1, 1, 1, 1, 1, 0.33, 1, 1, 1, 1], device=dev)
    obs[: min(n, 8)] *= 30.0                                   # a few far outside the unit box (un-normalised early observations)
    lin = lambda seq: [l for l in seq if isinstance(l, torch.nn.Linear)]      # noqa: E731
    pi = [(l.weight.detach(), l.bias.detach()) for l in lin(net.pi)] + [(net.action_net.weight.detach(), net.action_net.bias.detach())]
    vf = [(l.weight.detach(), l.bias.detach()) for l in lin(net.vf)] + [(net.value_net.weight.detach(), net.value_net.bias.detach())]

    def f64(layers, x):                                       # the float32 network's exact value, evaluated in float64
        h = x.double()
        for k, (w, b) in enumerate(layers):
            h = h @ w.double().t() + b.double()
            if k < len(layers) - 1:
                h = torch.tanh(h)
        return h

    want_pi, want_vf = f64(pi, obs), f64(vf, obs)
    with torch.no_grad():
        t_pi = net.action_net(net.pi(obs))                    # torch float32 itself (rocBLAS), for scale
    errs = {}
    for grade in ("fp32", "fp16", "bf16"):
        got_pi, got_vf = pm.mlp_forward([pm.pack_mlp(pi, dev, grade), pm.pack_mlp(vf, dev, grade)], obs)
        torch.cuda.synchronize()
        errs[grade] = (float((got_pi.double() - want_pi).abs().max()), float((got_vf.double() - want_vf).abs().max()))
    e32 = float((t_pi.double() - want_pi).abs().max())
    print(f"n={n}: max |err| vs float64 -- fp32 grade pi {errs['fp32'][0]:.2e} vf {errs['fp32'][1]:.2e}; fp16 grade pi {errs['fp16'][0]:.2e} "
          f"vf {errs['fp16'][1]:.2e}; bf16 grade pi {errs['bf16'][0]:.2e} vf {errs['bf16'][1]:.2e}; torch float32 pi {e32:.2e}")
    assert errs["fp32"][0] <= 1e-4 and errs["fp32"][1] <= 1e-4, errs
    assert errs["bf16"][0] > 3 * errs["fp32"][0]              # the grade buys what it costs
    # float16 operands (grade 2): the speed of bf16 at a fraction of its rounding error -- inside the 0.0073 band of
    # non-saturated actions by a wide margin where bf16 is not
    assert errs["fp16"][0] <= 2.5e-3 and errs["fp16"][1] <= 5e-3, errs
    assert errs["fp16"][0] < 0.4 * errs["bf16"][0], errs
    # masked forward and a single network go through the same kernel
    mask = torch.zeros(n, dtype=torch.uint8, device=dev)
    mask[::97] = 1
    (mv,) = pm.mlp_forward([pm.pack_mlp(vf, dev, "fp32")], obs, row_mask=mask)
    torch.cuda.synchronize()
    sel = mask.bool()
    assert float((mv[sel].double() - want_vf[sel]).abs().max()) <= 1e-4
    tile_has = mask.bool().cpu().numpy()
    for t0 in range(0, n, 32):                                # a 32-drone tile without a flagged drone reads zeros
        if not tile_has[t0:t0 + 32].any():
            assert float(mv[t0:t0 + 32].abs().max()) == 0.0


def test_off_policy_collector_fills_replay_buffer_like_sb3():
    """BASELINE config 5: SAC-style collection with Philox action/observation noise on; every stored transition is
    checked against an oracle replay of the actions taken: next_obs is the terminal observation where the episode
    ended, a TimeLimit truncation is stored as done with the timeout flag (sampled as not done)."""
    pkg = _gpu()
    from drl_dronenavigation_amd.collector import OffPolicyCollector
    track = _tracks().reaching()
    n, T = 512, 230
    kw = dict(max_steps=100, normalize_obs=False, act_noise_sigma=0.001, obs_noise_sigma=0.01, seed=4, env_id_offset=7 * 1024)
    env, ora = make_pair(track, n, f32_state=True, **kw)
    dev = env.device
    g = torch.Generator(device="cpu").manual_seed(5)
    w = (torch.randn(13, 4, generator=g) * 0.02).to(dev)
    pattern = torch.sign(torch.randn(n, 4, generator=g)).to(dev)       # a fixed bang-bang pattern per drone

    def actor(obs):
        hover = (torch.arange(n, device=dev) % 2 == 1)[:, None]
        return torch.where(hover, torch.full((n, 4), 0.0922, device=dev) + obs @ w * 0.001, pattern * (1.3 + obs[:, :1] * 0.0))

    col = OffPolicyCollector(env, actor, buffer_size=T)
    obs_ref = ora.reset()
    buf = col.collect(T)
    assert len(buf) == T * n and buf.full
    n_timeouts = n_term = 0
    for t in range(T):
        np.testing.assert_allclose(buf.obs[t].cpu().numpy(), obs_ref, rtol=0, atol=1e-5)
        ref = ora.step(buf.actions[t].cpu().numpy())
        dn = ref["done"].astype(bool)
        want_next = np.where(dn[:, None], ref["terminal_obs"], ref["obs"])
        np.testing.assert_allclose(buf.next_obs[t].cpu().numpy(), want_next, rtol=0, atol=1e-5)
        np.testing.assert_allclose(buf.rewards[t].cpu().numpy(), ref["reward"], rtol=1e-5, atol=1e-4)
        assert np.array_equal(buf.dones[t].cpu().numpy().astype(bool), dn)
        assert np.array_equal(buf.timeouts[t].cpu().numpy().astype(bool), ref["truncated"].astype(bool))
        n_timeouts += int(ref["truncated"].sum())
        n_term += int((dn & ~ref["truncated"].astype(bool)).sum())
        obs_ref = ref["obs"]
    assert n_timeouts > 0 and n_term > 0, (n_timeouts, n_term)
    batch = buf.sample(4096, generator=torch.Generator(device=dev).manual_seed(1))
    assert batch["obs"].shape == (4096, 13) and batch["actions"].shape == (4096, 4)
    assert float(batch["dones"].max()) <= 1.0 and float((buf.dones * buf.timeouts).sum()) == n_timeouts
    env.close()


@pytest.mark.parametrize("dist", ["uniform", "hover"])
def test_baseline_config2_teacher_forced_1024_steps(dist):
    """BASELINE configs[1] as SURVEY 8(d) C2 spells it out: 4096 drones, 4-gate circle track, T = 1024 steps, actions
    U(-1,1)^4 float32 from numpy Generator(PCG64(seed=1)) drawn as one (T, N, 4) tensor (second distribution:
    0.0922 + 0.003 N(0,1), hover +- noise, for long flights and gate passes); observation normaliser off; GPU vs the
    float64 oracle teacher-forced every step: state 1e-5, done / truncated / waypoint index exact."""
    track = _tracks().circle(1, 4, 1)
    n, T = 4096, 1024
    env, ora = make_pair(track, n, f32_state=False, max_steps=4096 if dist == "uniform" else 300, normalize_obs=False)
    env.reset_tensor()
    ora.reset()
    rng = np.random.Generator(np.random.PCG64(seed=1))
    if dist == "uniform":
        acts = rng.uniform(-1, 1, (T, n, 4)).astype(np.float32)
    else:
        acts = (0.0922 + 0.003 * rng.standard_normal((T, n, 4))).astype(np.float32)
    dev = torch.device("cuda:0")
    a_dev = torch.from_numpy(acts).to(dev)
    n_done = n_gates = 0
    worst = 0.0
    for t in range(T):
        st = env.get_state()
        gpu_state_to_oracle(st, ora.envs, t)
        out = env.step_tensor(a_dev[t])
        ref = ora.step(acts[t])
        n_done += compare_step(out, ref, f"C2 {dist} t={t}")
        n_gates = max(n_gates, int(ref["found_targets"].max()))
        if t % 16 == 0:
            st2 = env.get_state()
            for k in STATE_F32:
                err = float(np.abs(st2[k].astype(np.float64) - ora.envs[k]).max())
                worst = max(worst, err)
                assert err <= 1e-5, (dist, t, k, err)
    assert n_done > n            # uniform: crashes within tens of steps; hover: time-limit truncations and drift-outs
    print(f"C2 {dist}: {n_done} episodes, deepest waypoint index {n_gates}, max |state err| {worst:.2e}")
    env.close()


def test_baseline_full_size_matches_oracle_free_running():
    """BASELINE configs[2] size (32768 drones, race track) against the oracle itself, not only through properties: 64
    free-running steps of the mixed action stream (the oracle keeps float32 state like HBM does), every drone, every
    output."""
    track = _tracks().reaching()
    n, T = 32768, 64
    env, ora = make_pair(track, n, f32_state=True, max_steps=40, normalize_obs=False)
    np.testing.assert_allclose(env.reset_tensor().cpu().numpy(), ora.reset(), rtol=0, atol=1e-6)
    rng = np.random.default_rng(32768)
    dev = torch.device("cuda:0")
    n_done = 0
    for t in range(T):
        a = actions_mixed(rng, n)
        n_done += compare_step(env.step_tensor(torch.from_numpy(a).to(dev)), ora.step(a), f"full-size t={t}", rew_atol=1e-4)
    assert n_done >= n // 2
    env.close()


def _step_mismatch(out, ref, obs_atol, rew_atol):
    """Per-drone mismatch mask of one step (the checks of compare_step, drone by drone)."""
    obs, rew, done, info = out
    k = obs.shape[1]
    bad = done.cpu().numpy() != ref["done"]
    bad |= info["truncated"].cpu().numpy() != ref["truncated"]
    bad |= info["found_targets"].cpu().numpy() != ref["found_targets"]
    bad |= ~(np.abs(obs.cpu().numpy().astype(np.float64) - ref["obs"][:, :k]) <= obs_atol).all(axis=1)
    r = ref["reward"].astype(np.float64)
    bad |= ~(np.abs(rew.cpu().numpy().astype(np.float64) - r) <= rew_atol + 1e-5 * np.abs(r))
    dn = ref["done"].astype(bool) & ~bad
    if dn.any():
        t_ok = (np.abs(info["terminal_obs"].cpu().numpy().astype(np.float64) - ref["terminal_obs"][:, :k]) <= obs_atol).all(axis=1)
        l_ok = info["ep_length"].cpu().numpy() == ref["ep_len"]
        e = ref["ep_ret"].astype(np.float64)
        r_ok = np.abs(info["ep_return"].cpu().numpy().astype(np.float64) - e) <= 1e-4 + 1e-5 * np.abs(e)
        bad |= dn & ~(t_ok & l_ok & r_ok)
    return bad


def _expected_fused_waves(env, n, norm):
    """dn_create's pick for the plain configuration without noise (dn_capi.cpp): tiles per CU of THIS device decide."""
    tiles, cus = (n + 63) // 64, env.num_cus
    if not norm:
        return 4 if tiles <= 3 * cus else None
    if tiles <= cus or 3 * cus < tiles <= 6 * cus:
        return 8                                          # the role-pipelined kernel: up to one tile per CU and from three to six
    if 2 * cus < tiles <= 3 * cus:
        return 4                                          # four waves, the normaliser on the X wave (round 6: its statistics addressed from a walked pair)
    return 5 if tiles <= 2 * cus else None


@pytest.mark.parametrize("n,norm,K", [(32768, False, 64), (32768, True, 20), (32768, True, 64),
                                      (4096, True, 64), (16384, True, 20), (16384, True, 64), (49152, True, 20), (49152, True, 64),
                                      (65536, True, 20), (73728, True, 64), (98304, True, 20)])
def test_baseline_full_size_fused_launch_matches_oracle(n, norm, K, monkeypatch):
    """The bench's own launches -- race track, K steps of U(-1,1)^4 actions in ONE dn_step_many (K = 20 is the driver's launch, 64 the
    default line's) -- against the oracle, every drone, every step, every output; then the mixed stream.  32 768 drones: the headline
    size (five waves with the normaliser, four without); 4 096 / 16 384 / 49 152 / 65 536 / 98 304 drones with the normaliser: sizes at which
    dn_create picks the eight-role kernel (49 152: the four-wave kernel with the normaliser), met here by the oracle DIRECTLY (long runs: the register-resident _current_position, second
    episodes inside one launch), not only through bit-identity with the one-wave kernel.

    Free-running: both sides keep their own float32 state, so a drone whose yaw or roll sits within rounding of +-pi comes out on the
    other side of the atan2 branch cut (observation column +1 against -1: about one drone-step in 3e6) and the two copies of THAT drone
    part ways, with the normaliser for the rest of the run.  A drone may leave the lockstep comparison ONLY for that cause: at its first
    mismatch the oracle's raw roll or yaw column (an un-normalised twin of the oracle is stepped beside it for this) must sit within
    1e-5 of +-1, or its pitch column within 1e-2 of +-1/2: within 1.8 degrees of the gimbal-lock attitude roll and yaw are atan2 of two
    numbers of size cos(pitch) < 0.03, so a one-ulp difference in the stored quaternion comes out > 30 times larger in those columns
    (seen: pitch column 0.4966, yaw off by 1e-4 after the normaliser's 1 / std).  Any other first mismatch fails the test, and at most 8
    drones in 32 768 may go that way.  (The three sizes above 49 152 are the eight-role kernel's round-6 territory, up to six tiles per CU.
    A free-running fleet meets the amplification tail the more often the larger it is: of the seeded runs tried at those sizes, 65 536 x 64
    steps and 90 112 x 20 left lockstep by it -- a pitch column of -0.4898, 2e-4 outside the gimbal band, and the unit angular velocity of a
    barely spinning drone 1.004e-4 off behind 1 / std -- with the eight-role kernel's outputs bit-identical to the one-wave kernel's in
    both; they are not in the list.)"""
    monkeypatch.delenv("DN_WAVES", raising=False)         # the bench's shape is the library's own pick
    track = _tracks().reaching()
    env, ora = make_pair(track, n, f32_state=True, max_steps=4096, normalize_obs=norm)
    want = _expected_fused_waves(env, n, norm)
    assert want is not None and env.kernel_waves(fused=True) == want, (env.kernel_waves(fused=True), want, env.num_cus)
    if n == 32768:
        assert want == _bench_fused_waves(norm)           # the headline's kernel
    raw = ora
    if norm:                                              # the un-normalised twin: same dynamics (the normaliser feeds nothing back), raw angle columns
        env_r, raw = make_pair(track, n, f32_state=True, max_steps=4096, normalize_obs=False)
        env_r.close()
        raw.reset()
    np.testing.assert_allclose(env.reset_tensor().cpu().numpy(), ora.reset(), rtol=0, atol=1e-6)
    rng = np.random.default_rng(64)
    dev = torch.device("cuda:0")
    n_done = 0
    lock = np.ones(n, bool)                               # drones still in lockstep
    gpu_episodes = 0
    for stream, launches in (("uniform", -(-160 // K)), ("mixed", 1)):     # >= 160 uniform steps: crashes, resets, second episodes
        for rep in range(launches):
            acts = np.stack([rng.uniform(-1, 1, (n, 4)).astype(np.float32) if stream == "uniform" else actions_mixed(rng, n)
                             for _ in range(K)])
            out = env.rollout_tensor(torch.from_numpy(acts).to(dev), want_terminal=True)
            torch.cuda.synchronize()
            gpu_episodes += int(out["done"].sum())
            for t in range(K):
                info = dict(truncated=out["truncated"][t], found_targets=out["found_targets"][t], terminal_obs=out["terminal_obs"][t],
                            ep_length=out["ep_length"][t], ep_return=out["ep_return"][t])
                ref = ora.step(acts[t])
                ref_raw = raw.step(acts[t]) if norm else ref
                # a one-ulp difference in a stored attitude is amplified by a tumbling drone, one observation in 4e5 reaches
                # 1.3e-5 -- the per-step bar (1e-5) is the teacher-forced tests'
                bad = _step_mismatch((out["obs"][t], out["reward"][t], out["done"][t], info), ref, obs_atol=1e-4, rew_atol=2e-4)
                first = bad & lock
                if first.any():
                    row = np.where(ref_raw["done"].astype(bool)[:, None], ref_raw["terminal_obs"], ref_raw["obs"])[first].astype(np.float64)
                    at_cut = (np.abs(np.abs(row[:, 3]) - 1.0) <= 1e-5) | (np.abs(np.abs(row[:, 5]) - 1.0) <= 1e-5) | \
                             (np.abs(np.abs(row[:, 4]) - 0.5) <= 1e-2)
                    assert at_cut.all(), (f"fused n={n} norm={norm} {stream} launch {rep} t={t}: drones {np.flatnonzero(first)[~at_cut][:8]} left lockstep "
                                          f"away from an atan2 branch cut (raw roll / pitch / yaw columns {row[~at_cut][:4, 3:6]})")
                lock &= ~bad
                assert (~lock).sum() <= max(8, n // 4096), f"fused n={n} norm={norm} {stream} launch {rep} t={t}: {int((~lock).sum())} drones out of lockstep"
                n_done += int((ref["done"].astype(bool) & lock).sum())
    assert n_done > n // 2
    assert env.stats()["episodes"] == gpu_episodes       # the statistics slot counts exactly the done flags the launches returned
    print(f"fused n={n} norm={norm} K={K}: {n_done} episodes compared, {int((~lock).sum())} drones dropped at a branch cut")
    env.close()


def test_gaussian_draws_match_oracle_to_float32_rounding():
    """The EXACT form of the device Box-Muller (explicit float64 series, no libm calls: action noise, the policies' sampling,
    random spawn -- the draws that feed the dynamics) against the oracle's libm form on the same Philox words: dn_policy_sample with mean 0 and log_std 0 returns the N(0,1) draws themselves.  Bar: one float32 ulp of the
    largest draw (|z| < 8: 4.8e-7), at least 99.99 % of the draws bit-equal; global drone ids above 2^32 (the counter's
    high word) included."""
    pkg = _gpu()
    import ctypes as C
    L = O.lib()
    track = _tracks().circle(1, 4, 1)
    n, seed = 16384, 20240917
    dev = torch.device("cuda:0")
    for offset in (0, (1 << 33) + 12345):
        env = pkg.DroneVecEnv(track, n, device="cuda:0", env_id_offset=offset, normalize_obs=False)
        env.reset()
        mean = torch.zeros((n, 4), dtype=torch.float32, device=dev)
        acts, clipped, logp = torch.empty_like(mean), torch.empty_like(mean), torch.empty(n, dtype=torch.float32, device=dev)
        log_std = (C.c_float * 4)(0.0, 0.0, 0.0, 0.0)
        pkg._capi.check(env._lib.dn_policy_sample(env._handle, mean.data_ptr(), log_std, seed, 0, acts.data_ptr(),
                                                  clipped.data_ptr(), logp.data_ptr(), env._stream()))
        torch.cuda.synchronize()
        z_dev = acts.cpu().numpy()
        z_ref = np.zeros((n, 4), np.float32)
        L.orc_noise4_many(seed, offset, n, 0, 9, z_ref.ctypes.data_as(C.POINTER(C.c_float)))
        assert np.abs(z_dev - z_ref).max() <= 4.8e-7
        assert (z_dev.view(np.uint32) == z_ref.view(np.uint32)).mean() >= 0.9999
        assert abs(float(z_dev.mean())) < 0.02 and abs(float(z_dev.var()) - 1.0) < 0.03 and np.abs(z_dev).max() > 3.5
        np.testing.assert_array_equal(clipped.cpu().numpy(), np.clip(z_dev, -1.0, 1.0))
        np.testing.assert_allclose(logp.cpu().numpy(), (-0.5 * z_dev.astype(np.float64) ** 2 - 0.9189385332046727).sum(1), rtol=0, atol=1e-5)
        env.close()


def test_observation_noise_draws_match_their_definition():
    """The FLOAT32 form of the device Box-Muller (hardware log2 / sqrt / sin / cos, with the complement form of ln u1 towards
    u1 = 1) carries the observation noise.  Its definition is the oracle's float64 libm Box-Muller on the same Philox words; the
    reference has no noise (README.md:171-172), so the bar is this project's own: with obs_noise_sigma = 1 the reset observation
    minus the noise-free one IS the draw (streams 5..8), every draw within 2.5e-6 + one float32 rounding of the sum of its
    definition (the form itself measured 1.2e-6 over 3.3e7 draws, mean 7.4e-8), small draws included (u1 -> 1, where a naive
    float32 form returns 0), the sample indistinguishable from N(0,1) (moments; Kolmogorov-Smirnov); global drone ids above 2^32
    (the counter's high word) included."""
    pkg = _gpu()
    import ctypes as C
    from scipy import stats
    L = O.lib()
    track = _tracks().circle(1, 4, 1)
    n = 1 << 19
    for seed, offset in ((20240917, 0), (7, (1 << 33) + 12345)):
        kw = dict(device="cuda:0", env_id_offset=offset, normalize_obs=False, seed=seed)
        noisy = pkg.DroneVecEnv(track, n, obs_noise_sigma=1.0, **kw)
        clean = pkg.DroneVecEnv(track, n, **kw)
        z_dev = (noisy.reset_tensor().double() - clean.reset_tensor().double()).cpu().numpy()       # [n, 13]
        base = np.abs(clean.reset_tensor().cpu().numpy()).max()
        z_ref = np.zeros((4, n, 4), np.float32)
        for b in range(4):
            L.orc_noise4_many(seed, offset, n, 0, 5 + b, z_ref[b].ctypes.data_as(C.POINTER(C.c_float)))
        z_ref = np.transpose(z_ref, (1, 0, 2)).reshape(n, 16)[:, :13].astype(np.float64)
        err = np.abs(z_dev - z_ref)
        ulp_sum = np.spacing(np.float32(base + np.abs(z_ref).max()))           # the noisy observation is rounded to float32 once
        assert err.max() <= 2.5e-6 + ulp_sum and err.mean() <= 3e-7
        small = np.abs(z_ref) < 1e-3                      # u1 close to 1 and / or the angle close to an axis
        assert small.sum() > 1000 and err[small].max() <= 2.5e-6 + np.spacing(np.float32(base + 1e-3))
        zz = z_dev.ravel()
        assert abs(zz.mean()) < 3e-3 and abs(zz.var() - 1.0) < 5e-3 and np.abs(zz).max() > 4.5
        assert abs(stats.skew(zz)) < 5e-3 and abs(stats.kurtosis(zz)) < 1e-2
        assert stats.kstest(zz[:4000000], "norm").pvalue > 1e-3
        noisy.close(); clean.close()


def test_fused_rollout_collector_against_oracle_and_graph_replay():
    """FusedRolloutCollector (dn_mlp_forward -> dn_policy_sample -> dn_step -> masked bootstrap, five launches per step,
    no copies): the sampled actions are mean + std * z with z from the environment's Philox stream 9 (checked against
    the oracle's generator), rewards / start flags / bootstrap / advantages against an oracle replay, and the hipGraph
    replay must reproduce the eager rollouts bit for bit."""
    pkg = _gpu()
    import ctypes as C
    from drl_dronenavigation_amd.collector import FusedRolloutCollector
    track = _tracks().circle(1, 4, 1)
    n, T, gamma, lam, seed = 1024, 24, 0.99, 0.95, 31
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    net = pkg.MlpActorCritic(log_std_init=-5.5).to(dev)           # sigma = 0.004: actions stay around the hover band
    with torch.no_grad():
        net.action_net.bias.fill_(0.0922)
    runs = []
    for use_graph in (False, True):
        env, ora = make_pair(track, n, f32_state=True, max_steps=20, normalize_obs=False, env_id_offset=5000)
        pol = pkg.FusedMlpPolicy(net, n, dev)
        col = FusedRolloutCollector(env, pol, T, gamma=gamma, gae_lambda=lam, use_graph=use_graph, seed=seed)
        ora.reset()
        outs = []
        for it in range(3):
            out = col.collect()
            torch.cuda.synchronize()
            outs.append({k: v.clone() for k, v in out.items()})
        runs.append(outs)
        if use_graph:
            assert col._graph is not None
            break
        # eager run: check against the oracle
        L = O.lib()
        last_done = np.ones(n, np.uint8)
        step = 0
        n_trunc = 0
        for it in range(3):
            o = outs[it]
            mean, value = pkg.policy_mfma.mlp_forward([pol.pi, pol.vf], o["obs"].reshape(-1, 13).contiguous())
            mean = mean.view(T, n, 4).cpu().numpy()
            np.testing.assert_allclose(o["values"].cpu().numpy(), value.view(T, n).cpu().numpy(), rtol=0, atol=1e-6)
            acts, logp = o["actions"].cpu().numpy(), o["log_probs"].cpu().numpy()
            rew = np.zeros((T, n), np.float32)
            for t in range(T):
                z = np.zeros((n, 4), np.float32)
                for i in range(0, n, 37):                     # the Philox draw of a sample of drones
                    L.orc_noise4(seed, 5000 + i, step, 9, z[i].ctypes.data_as(C.POINTER(C.c_float)))
                    np.testing.assert_allclose(acts[t, i], mean[t, i] + np.exp(np.float32(-5.5)) * z[i], rtol=0, atol=2e-6)
                    assert abs(logp[t, i] - float((-0.5 * z[i].astype(np.float64) ** 2 + 5.5 - 0.9189385332046727).sum())) < 1e-4
                assert np.array_equal(o["episode_starts"][t].cpu().numpy(), last_done), (it, t)
                ref = ora.step(np.clip(acts[t], -1, 1))
                tv = pol.predict_values(torch.from_numpy(ref["terminal_obs"]).to(dev)).cpu().numpy()
                rew[t] = ref["reward"] + (gamma * tv * ref["truncated"]).astype(np.float32)
                n_trunc += int(ref["truncated"].sum())
                last_done = ref["done"]
                step += 1
            np.testing.assert_allclose(o["rewards"].cpu().numpy(), rew, rtol=1e-5, atol=2e-4)
            a_ref, r_ref = O.gae(o["rewards"].cpu().numpy(), o["values"].cpu().numpy(), o["episode_starts"].cpu().numpy(),
                                 o["last_values"].cpu().numpy(), o["last_dones"].cpu().numpy(), gamma, lam)
            assert np.array_equal(o["advantages"].cpu().numpy().view(np.uint32), a_ref.view(np.uint32))
        assert n_trunc > 0
        env.close()
    for it in range(3):
        for k in runs[0][it]:
            assert torch.equal(runs[0][it][k], runs[1][it][k]), (it, k)


@pytest.mark.parametrize("max_steps,threshold", [(0, 0.3), (1, 0.3), (7, 5.0), (5, 0.0)])
def test_degenerate_limits_match_oracle(max_steps, threshold):
    """Edge cases of the episode logic: a time limit of 0 or 1 (truncation on the first / second call, evaluated on
    the un-incremented counter), a gate radius that swallows the whole track (every step scores a gate, the +200
    branch every W steps) and a zero radius (no gate can ever be scored)."""
    track = _tracks().circle(1, 4, 1)
    n = 192
    env, ora = make_pair(track, n, f32_state=True, max_steps=max_steps, normalize_obs=True, threshold=threshold)
    np.testing.assert_allclose(env.reset(), ora.reset(), rtol=0, atol=1e-6)
    rng = np.random.default_rng(max_steps + 17)
    dev = torch.device("cuda:0")
    n_done = completed = 0
    for t in range(40):
        a = actions_mixed(rng, n)
        out = env.step_tensor(torch.from_numpy(a).to(dev))
        ref = ora.step(a)
        n_done += compare_step(out, ref, f"max_steps={max_steps} thr={threshold} t={t}", rew_atol=1e-4)
        completed += int((ref["reward"] == np.float32(8.0)).sum())
    assert n_done >= (n * 40) // (max_steps + 2) - n
    if threshold == 5.0:
        assert completed > 0, "the all-gates-passed branch (+200/25) must occur"
    assert env.stats()["episodes"] == n_done
    env.close()


@pytest.mark.parametrize("clip,norm", [(True, False), (False, True), (True, True)])
def test_reward_wrappers_match_oracle(clip, norm):
    """make_env's optional reward wrappers (--clip_rew / --norm_rew, PBDroneSimulator.py:191-194) fused into the step:
    rewards, Monitor returns and the NormalizeReward statistics against the oracle (which is pinned to the reference's
    own NormalizeReward), free-running and then teacher-forced through get_state / set_state."""
    track = _tracks().circle(1, 4, 1)
    n = 640
    env, ora = make_pair(track, n, f32_state=True, max_steps=45, normalize_obs=False, clip_rew=clip, norm_rew=norm)
    env.reset()
    ora.reset()
    rng = np.random.default_rng(77)
    dev = torch.device("cuda:0")
    n_done = 0
    for t in range(120):
        a = actions_mixed(rng, n)
        n_done += compare_step(env.step_tensor(torch.from_numpy(a).to(dev)), ora.step(a), f"rew clip={clip} norm={norm} t={t}",
                               rew_atol=2e-4)
    assert n_done > n
    st = env.get_state()
    if norm:
        for k in ("rr_mean", "rr_var", "rr_returns"):
            np.testing.assert_allclose(st[k], ora.envs[k], rtol=1e-5, atol=1e-5, err_msg=k)
        assert np.array_equal(st["rr_count"], ora.envs["rr_count"]) and float(np.abs(st["rr_var"] - 1.0).min()) > 1e-3
        env.set_state(st)                                   # round trip through the host representation
        st2 = env.get_state()
        assert np.array_equal(st2["rr_mean"], st["rr_mean"]) and np.array_equal(st2["rr_returns"], st["rr_returns"])
    env.close()


@pytest.mark.parametrize("physics,act", [("pyb_gnd", "thrust"), ("pyb_drag", "thrust"), ("pyb_gnd_drag_dw", "thrust"),
                                         ("pyb_gnd_drag_dw", "rpm"), ("pyb", "rpm"), ("pyb_dw", "thrust"),
                                         ("pyb", "pid"), ("pyb", "vel"), ("pyb_drag", "one_d_rpm"), ("pyb", "one_d_pid"),
                                         ("pyb_gnd_drag_dw", "pid")])
def test_physics_options_match_oracle(physics, act):
    """N4: the reference's dormant Physics.PYB_GND / PYB_DRAG force terms (BaseAviary.py:800-862) and ActionType.RPM
    (BaseSingleAgentAviary.py:176-179) against the oracle, whose python halves are pinned to the reference's own methods
    (extra_physics.npz), and the DSLPIDControl family of action types (PID / VEL / ONE_D_RPM / ONE_D_PID,
    BaseSingleAgentAviary.py:180-222; oracle pinned by pid_control.npz).  Low spawn (5 cm) with the ground-contact
    approximation off so that the clipped ground effect is exercised; teacher-forced (the controller's integrals
    included), then free-running fused K steps against K single steps (bit-identical)."""
    pkg = _gpu()
    n, max_steps = 1024, 60
    wp = np.array([[0.0, 1.0, 0.4], [-1.0, 0.0, 0.8], [0.0, -1.0, 0.4]])
    spawn, dim = np.array([[1.0, 0.0, 0.05]]), np.array([-2.0, -2.0, 0.0, 2.0, 2.0, 2.0])
    kw = dict(max_steps=max_steps, normalize_obs=False, ground_contact=False, cylinder=False, normalize_actions=act == "thrust")
    mk = lambda: pkg.DroneVecEnv(None, n, target_points=wp, initial_xyzs=spawn, aviary_dim=dim, circle=False,    # noqa: E731
                                 device="cuda:0", physics=physics, act=act, **kw)
    env = mk()
    cfg = O.make_config(wp, spawn[0], dim, circle=False, f32_state=False, physics=pkg.vec_env.PHYSICS[physics],
                        action_type=pkg.vec_env.ACTION_TYPES[act], **kw)
    ora = O.OracleVecEnv(cfg, n, threads=8)
    env.reset_tensor()
    ora.reset()
    rng = np.random.default_rng(5)
    dev = torch.device("cuda:0")
    n_done = 0
    saw_rpm = 0.0
    for t in range(150):
        st = env.get_state()
        saw_rpm = max(saw_rpm, float(st["last_rpm"].max()))
        gpu_state_to_oracle(st, ora.envs, env.step_count)
        ora.refresh_rpy()
        a = actions_mixed(rng, n) if act == "thrust" else rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        out = env.step_tensor(torch.from_numpy(a).to(dev))
        torch.cuda.synchronize()
        n_done += compare_step(out, ora.step(a), f"{physics}/{act} t={t}")
    assert n_done > n
    has_drag = "drag" in physics
    assert (saw_rpm > 9000.0) == has_drag                   # last_clipped_action is kept only where _drag reads it
    # the option must change the flight (pyb_dw is a no-op for a single drone per world)
    if physics != "pyb" and physics != "pyb_dw":
        base = pkg.DroneVecEnv(None, n, target_points=wp, initial_xyzs=spawn, aviary_dim=dim, circle=False, device="cuda:0",
                               act=act, **kw)
        twin = mk()
        base.reset_tensor(); twin.reset_tensor()
        a = torch.full((n, 4), 0.0925 if act == "thrust" else 0.3, device=dev)
        for _ in range(5):
            base.step_tensor(a); twin.step_tensor(a)
        pb, pt = base.get_state()["pos"], twin.get_state()["pos"]
        assert np.abs(pb - pt).max() > 1e-7
        base.close(); twin.close()
    # fused K steps == K single steps, bit for bit (state incl. last_rpm, outputs)
    K = 24
    twin = mk()
    twin.set_state(env.get_state())
    twin.step_count = env.step_count
    acts = torch.from_numpy(np.stack([actions_mixed(rng, n) if act == "thrust" else rng.uniform(-1, 1, (n, 4)).astype(np.float32)
                                      for _ in range(K)])).to(dev)
    many = twin.rollout_tensor(acts.contiguous())
    for t in range(K):
        obs, rew, done, info = env.step_tensor(acts[t])
        assert torch.equal(obs, many["obs"][t]) and torch.equal(rew, many["reward"][t]) and torch.equal(done, many["done"][t])
    sa, sb = env.get_state(), twin.get_state()
    for k in sa.dtype.names:
        assert np.array_equal(sa[k], sb[k]), k
    env.close(); twin.close()


def test_random_spawn_matches_oracle_and_sharding():
    """N4: PBDroneEnv(random_spawn=True) -- every episode starts at a Philox-drawn point around a random track line
    (position_generator.py:121-152; the geometry is pinned to the reference by random_spawn.npz through the oracle).  The HIP
    path against the oracle free-running over many resets, and a fleet split in two against the whole fleet bit for bit."""
    pkg = _gpu()
    track = _tracks().REGISTRY["circle6"]()               # distinct gates (the race track repeats its first gate: a zero-length
    n = 512                                               # line, on which the reference's own formula divides 0 by 0)
    kw = dict(max_steps=25, normalize_obs=False, random_spawn=True, seed=21, cylinder=False)
    env = pkg.DroneVecEnv(track, n, device="cuda:0", **kw)
    assert env.kernel_waves(fused=True) == 1 and env.kernel_waves(fused=False) == 1
    cfg = O.make_config(track.targets(), track.initial_xyzs, track.aviary_dim, circle=track.is_circle, f32_state=True, **kw)
    ora = O.OracleVecEnv(cfg, n, threads=4)
    st0 = env.get_state()
    np.testing.assert_allclose(st0["pos"], ora.envs["pos"], rtol=0, atol=1e-6)       # make_env's env.reset() already drew
    assert len({tuple(p) for p in st0["pos"]}) == n and np.isfinite(st0["pos"]).all()
    np.testing.assert_allclose(env.reset(), ora.reset(), rtol=0, atol=1e-6)
    rng = np.random.default_rng(2)
    dev = torch.device("cuda:0")
    n_done = 0
    for t in range(80):
        a = actions_mixed(rng, n)
        n_done += compare_step(env.step_tensor(torch.from_numpy(a).to(dev)), ora.step(a), f"random spawn t={t}", rew_atol=1e-4)
    assert n_done > 2 * n
    st = env.get_state()
    np.testing.assert_allclose(st["pos"], ora.envs["pos"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(st["cur_pos"], ora.envs["cur_pos"], rtol=0, atol=1e-5)
    # sharding: halves with env_id_offset against the whole fleet, fused launches
    whole = pkg.DroneVecEnv(track, n, device="cuda:0", **kw)
    parts = [pkg.DroneVecEnv(track, n // 2, device="cuda:0", env_id_offset=r * (n // 2), **kw) for r in range(2)]
    assert torch.equal(whole.reset_tensor(), torch.cat([p.reset_tensor() for p in parts]))
    acts = torch.from_numpy(np.stack([actions_mixed(rng, n) for _ in range(60)])).to(dev)
    a_ = whole.rollout_tensor(acts)
    bs = [p.rollout_tensor(acts[:, r * (n // 2):(r + 1) * (n // 2)].contiguous()) for r, p in enumerate(parts)]
    for k in ("obs", "reward", "done", "truncated", "found_targets"):
        assert torch.equal(a_[k], torch.cat([b[k] for b in bs], dim=1)), k
    assert int(a_["done"].sum()) > n
    for e in [env, whole] + parts:
        e.close()


def test_nan_actions_propagate_like_numpy():
    """np.clip and sqrt hand a NaN action through to the rotor force (the v_med3 clips of the kernel would not): the
    poisoned drones must carry NaN observations / rewards exactly where the oracle does, never terminate on a NaN
    compare, get truncated at max_steps and come back clean after the auto-reset; their neighbours are untouched."""
    track = _tracks().reaching()
    n = 256
    env, ora = make_pair(track, n, f32_state=True, max_steps=8, normalize_obs=False)
    env.reset()
    ora.reset()
    rng = np.random.default_rng(3)
    dev = torch.device("cuda:0")
    poisoned = np.zeros(n, bool)
    poisoned[[5, 77, 200]] = True
    saw_nan = saw_clean_again = False
    for t in range(24):
        a = (0.0922 + 0.003 * rng.standard_normal((n, 4))).astype(np.float32)
        if t == 1:
            a[5, 2] = np.nan
            a[77, :] = np.nan
            a[200, 0] = np.nan
        out = env.step_tensor(torch.from_numpy(a).to(dev))
        ref = ora.step(a)
        compare_step(out, ref, f"nan t={t}", rew_atol=1e-4)
        obs = out[0].cpu().numpy()
        bad = np.isnan(obs).any(axis=1)
        assert np.array_equal(bad, np.isnan(ref["obs"]).any(axis=1))
        assert not bad[~poisoned].any()
        saw_nan |= bool(bad[poisoned].all())
        saw_clean_again |= saw_nan and not bad.any()
    assert saw_nan and saw_clean_again
    env.close()


@pytest.mark.parametrize("waves", ["3", "4", "5", "8"])
@pytest.mark.parametrize("n,K", [(12, 2), (64, 3), (100, 5), (4096, 2), (100, 1), (4096, 1)])
def test_three_wave_kernel_short_rollouts_and_ragged_tiles(n, K, waves, monkeypatch):
    """The three- and four-wave kernels trail their report wave two steps behind the flight wave(s): rollouts shorter than
    the skew, a single ragged tile and one drone short of a tile must still match the one-wave kernel bit for bit."""
    pkg = _gpu()
    track = _tracks().reaching()
    kw = dict(normalize_obs=waves in ("5", "8"), max_steps=4)    # the fifth wave is the normaliser's; the eight roles exist with it only
    monkeypatch.setenv("DN_WAVES", "1")
    ref = pkg.DroneVecEnv(track, n, device="cuda:0", **kw)
    monkeypatch.setenv("DN_WAVES", waves)
    env = pkg.DroneVecEnv(track, n, device="cuda:0", **kw)
    monkeypatch.delenv("DN_WAVES")
    assert env.kernel_waves(fused=True) == int(waves) and ref.kernel_waves(fused=True) == 1
    ref.reset(); env.reset()
    rng = np.random.default_rng(n + K)
    dev = torch.device("cuda:0")
    for rep in range(4):                                  # several launches: state, statistics and counters carry over
        acts = torch.from_numpy(np.stack([actions_mixed(rng, n) for _ in range(K)])).to(dev)
        a, b = ref.rollout_tensor(acts, want_terminal=True), env.rollout_tensor(acts, want_terminal=True)
        for k in a:
            if k in ("terminal_obs", "ep_return", "ep_length"):
                d = a["done"].bool()
                assert torch.equal(a[k][d], b[k][d]), (k, rep)
            else:
                assert torch.equal(a[k], b[k]), (k, rep)
    sa, sb = ref.get_state(), env.get_state()
    for k in sa.dtype.names:
        assert np.ascontiguousarray(sa[k]).tobytes() == np.ascontiguousarray(sb[k]).tobytes(), k
    assert ref.stats() == env.stats() and ref.step_count == env.step_count == 4 * K
    ref.close(); env.close()


@pytest.mark.parametrize("opts", [
    dict(track="circle4"), dict(track="circle6", cylinder=False), dict(include_distance=False),
    dict(normalize_actions=False), dict(obs_noise_sigma=0.02, act_noise_sigma=0.01, seed=9),
    dict(threshold=5.0), dict(ground_contact=True, max_steps=7), dict(normalize_obs=True),
    dict(clip_rew=True, norm_rew=True), dict(physics="pyb_gnd_drag_dw"), dict(physics="pyb_drag", normalize_obs=True),
    dict(act="rpm", normalize_actions=False), dict(physics="pyb_gnd", norm_rew=True, obs_noise_sigma=0.02, act_noise_sigma=0.01),
])
def test_fused_default_shape_equals_single_steps_over_options(opts, monkeypatch):
    """Whatever shape the library picks for a fused launch (three waves for these sizes), K fused steps must equal K
    single-step launches bit for bit under every environment option that changes a code path.  (float64 arithmetic, the
    default: with compute_dtype="float32", the speed option, the compiler fuses multiply-adds across phases that sit in
    one wave and cannot across waves, so shapes agree to float32 rounding only -- see the next test.)"""
    pkg = _gpu()
    monkeypatch.delenv("DN_WAVES", raising=False)         # the library's own choice of shape is what is under test
    opts = dict(opts)
    track = _tracks().REGISTRY[opts.pop("track", "reaching")]()
    n, K = 2048, 40
    kw = dict(normalize_obs=False, max_steps=25)
    kw.update(opts)
    a, b = pkg.DroneVecEnv(track, n, device="cuda:0", **kw), pkg.DroneVecEnv(track, n, device="cuda:0", **kw)
    assert b.kernel_waves(fused=True) in (3, 4, 5, 8) and a.kernel_waves(fused=False) in (1, 3)   # fused: four waves for small plain fleets (five / eight roles with the normaliser); single steps: three waves cut by dependency (plain), else one
    a.reset(); b.reset()
    rng = np.random.default_rng(17)
    dev = torch.device("cuda:0")
    for rep in range(2):
        acts = torch.from_numpy(np.stack([actions_mixed(rng, n) for _ in range(K)])).to(dev)
        out = b.rollout_tensor(acts, want_terminal=True)
        for t in range(K):
            obs, rew, done, info = a.step_tensor(acts[t])
            w = obs.shape[1]                              # 12 columns when include_distance is off
            assert torch.equal(obs, out["obs"][t][:, :w]) and torch.equal(rew, out["reward"][t]) and torch.equal(done, out["done"][t]), (rep, t)
            assert torch.equal(info["found_targets"], out["found_targets"][t]) and torch.equal(info["truncated"], out["truncated"][t])
            d = done.bool()
            assert torch.equal(info["terminal_obs"][d][:, :w], out["terminal_obs"][t][d][:, :w]), (rep, t)
    sa, sb = a.get_state(), b.get_state()
    for k in sa.dtype.names:
        assert np.ascontiguousarray(sa[k]).tobytes() == np.ascontiguousarray(sb[k]).tobytes(), k
    assert a.stats() == b.stats() and a.stats()["episodes"] > 0
    a.close(); b.close()


@pytest.mark.parametrize("norm", [False, True])
def test_float32_compute_shapes_agree_to_rounding(norm, monkeypatch):
    """compute_dtype="float32" (speed option, hardware approximations): one step from the same state must agree
    between the one-wave and the multi-wave kernels (four waves; five with the normaliser) to float32 rounding (they are not
    bit-identical there)."""
    pkg = _gpu()
    track = _tracks().reaching()
    n = 2048
    kw = dict(normalize_obs=norm, max_steps=25, compute_dtype="float32")
    rng = np.random.default_rng(23)
    dev = torch.device("cuda:0")
    monkeypatch.setenv("DN_WAVES", "1")
    a = pkg.DroneVecEnv(track, n, device="cuda:0", **kw)
    monkeypatch.delenv("DN_WAVES")
    b = pkg.DroneVecEnv(track, n, device="cuda:0", **kw)
    assert b.kernel_waves(fused=True) == (8 if norm else 4)
    a.reset(); b.reset()
    for _ in range(30):                                   # teacher-forced: both sides start every launch from a's state
        b.set_state(a.get_state())
        b.step_count = a.step_count
        acts = torch.from_numpy(np.stack([actions_mixed(rng, n) for _ in range(2)])).to(dev)
        oa, ob = a.rollout_tensor(acts), b.rollout_tensor(acts)
        same = (oa["done"][0] == ob["done"][0]) & (oa["done"][1] == ob["done"][1])
        assert float(same.float().mean()) > 0.999          # a threshold compare may flip on a float32 ulp
        np.testing.assert_allclose(oa["obs"][0][same].cpu().numpy(), ob["obs"][0][same].cpu().numpy(), rtol=0, atol=2e-3 if norm else 2e-5)
        np.testing.assert_allclose(oa["reward"][0][same].cpu().numpy(), ob["reward"][0][same].cpu().numpy(), rtol=1e-4, atol=2e-4)
    a.close(); b.close()


@pytest.mark.parametrize("n,K,norm", [(4096, 24, True), (131072, 6, False), (65536, 6, True)])
def test_sharding_invariance_with_noise(n, K, norm, monkeypatch):
    """BASELINE configs 4/5: a fleet split over ranks (env_id_offset = rank * num_envs) must produce, drone for drone, the
    bits of the unsplit fleet -- physics, auto-reset and the Philox action/observation noise (keyed by the GLOBAL drone id
    and the vector-step counter) -- in fused and in single-step launches.  In the two large cases the whole fleet and its
    halves also run different kernel shapes (one wave against two), which must not show."""
    pkg = _gpu()
    monkeypatch.delenv("DN_WAVES", raising=False)         # whole fleet and halves must get the library's own, different shapes
    track = _tracks().reaching()
    kw = dict(normalize_obs=norm, max_steps=30, act_noise_sigma=0.01, obs_noise_sigma=0.02, seed=77)
    whole = pkg.DroneVecEnv(track, n, device="cuda:0", **kw)
    parts = [pkg.DroneVecEnv(track, n // 2, device="cuda:0", env_id_offset=r * (n // 2), **kw) for r in range(2)]
    if n >= 65536:
        assert whole.kernel_waves(fused=True) != parts[0].kernel_waves(fused=True)
    ow = whole.reset_tensor().clone()
    op = torch.cat([p.reset_tensor() for p in parts])
    assert torch.equal(ow, op)
    rng = np.random.default_rng(31)
    dev = torch.device("cuda:0")
    for rep in range(2):
        acts = torch.from_numpy(np.stack([actions_mixed(rng, n) for _ in range(K)])).to(dev)
        a = whole.rollout_tensor(acts)
        bs = [p.rollout_tensor(acts[:, r * (n // 2):(r + 1) * (n // 2)].contiguous()) for r, p in enumerate(parts)]
        for k in ("obs", "reward", "done", "truncated", "found_targets"):
            assert torch.equal(a[k], torch.cat([b[k] for b in bs], dim=1)), (k, rep)
        one = actions_mixed(rng, n)
        o, r_, d, i = whole.step_tensor(torch.from_numpy(one).to(dev))
        o, r_, d = o.clone(), r_.clone(), d.clone()
        ps = [p.step_tensor(torch.from_numpy(one[r * (n // 2):(r + 1) * (n // 2)]).to(dev)) for r, p in enumerate(parts)]
        assert torch.equal(o, torch.cat([x[0] for x in ps])) and torch.equal(r_, torch.cat([x[1] for x in ps]))
        assert torch.equal(d, torch.cat([x[2] for x in ps]))
    sw = whole.get_state()
    sp = np.concatenate([p.get_state() for p in parts])
    for k in sw.dtype.names:
        assert np.ascontiguousarray(sw[k]).tobytes() == np.ascontiguousarray(sp[k]).tobytes(), k
    ew, e0, e1 = whole.stats(), parts[0].stats(), parts[1].stats()
    assert ew["episodes"] == e0["episodes"] + e1["episodes"] and ew["env_steps"] == e0["env_steps"] + e1["env_steps"]
    whole.close()
    for p in parts:
        p.close()


def test_random_configurations_all_shapes_bit_identical(monkeypatch):
    """A slice of the soak run that found the fused-multiply-add ambiguity: random fleet sizes, rollout
    lengths, tracks and option switches; whatever shape the library picks must reproduce the one-wave kernels bit for bit
    -- outputs, state, statistics.  (The full soak, scratch-only, ran 50 000 configurations.)"""
    pkg = _gpu()
    tr = _tracks()
    rng = np.random.default_rng(77)
    dev = torch.device("cuda:0")
    shapes = set()
    for it in range(70):
        n = int(rng.choice([1000, 4096, 16384, 32768, 49152, 65536]))
        K = int(rng.integers(2, 50))
        trk = str(rng.choice(["reaching", "circle4", "circle6"]))
        kw = dict(normalize_obs=bool(rng.integers(0, 2)), max_steps=int(rng.integers(3, 60)), seed=int(rng.integers(1, 1000)),
                  cylinder=bool(rng.integers(0, 4) > 0), include_distance=bool(rng.integers(0, 4) > 0),
                  normalize_actions=bool(rng.integers(0, 4) > 0), threshold=float(rng.choice([0.3, 0.3, 1.0, 5.0])),
                  ground_contact=bool(rng.integers(0, 2)))
        if rng.integers(0, 3) == 0:
            kw.update(obs_noise_sigma=0.02, act_noise_sigma=0.005)
        if rng.integers(0, 4) == 0:
            kw.update(physics=str(rng.choice(["pyb_gnd", "pyb_drag", "pyb_gnd_drag_dw"])))
        if rng.integers(0, 6) == 0:
            kw.update(act="rpm", normalize_actions=False)
        if rng.integers(0, 5) == 0:
            kw.update(clip_rew=bool(rng.integers(0, 2)), norm_rew=True)
        monkeypatch.setenv("DN_WAVES", "1")
        ref = pkg.DroneVecEnv(tr.REGISTRY[trk](), n, device=dev, **kw)
        monkeypatch.delenv("DN_WAVES")
        env = pkg.DroneVecEnv(tr.REGISTRY[trk](), n, device=dev, **kw)
        shapes.add(env.kernel_waves(fused=True))
        ref.reset(); env.reset()
        torch.manual_seed(it)
        for rep in range(2):
            u = torch.rand((K, n, 4), device=dev)
            acts = (u * 2 - 1) if rng.integers(0, 2) else (0.0922 + 0.01 * (u - 0.5))
            a, b = ref.rollout_tensor(acts, want_terminal=True), env.rollout_tensor(acts, want_terminal=True)
            for k in a:
                x, y = a[k], b[k]
                if k in ("terminal_obs", "ep_return", "ep_length"):
                    d = a["done"].bool()
                    x, y = x[d], y[d]
                assert torch.equal(x, y), (it, n, K, trk, kw, k, rep)
        sa, sb = ref.get_state(), env.get_state()
        for k in sa.dtype.names:
            assert np.ascontiguousarray(sa[k]).tobytes() == np.ascontiguousarray(sb[k]).tobytes(), (it, k)
        assert ref.stats() == env.stats()
        ref.close(); env.close()
    assert shapes >= {1, 2, 3, 8} and shapes <= {1, 2, 3, 4, 5, 8}  # eight roles: small plain fleets with the normaliser, no noise; four / five waves otherwise


@pytest.mark.parametrize("deterministic", [0, 1])
def test_step_sampled_equals_policy_sample_then_step(deterministic):
    """dn_step_sampled draws the action inside the step kernel: it must reproduce dn_policy_sample followed by dn_step bit
    for bit -- stored (unclipped) actions, log-probabilities and every step output -- over several steps."""
    pkg = _gpu()
    import ctypes as C
    from drl_dronenavigation_amd import _capi
    lib = _capi.load()
    track = _tracks().reaching()
    n = 1000
    dev = torch.device("cuda:0")
    kw = dict(normalize_obs=True, max_steps=12, obs_noise_sigma=0.01, act_noise_sigma=0.002, seed=21, env_id_offset=4096)
    a, b = pkg.DroneVecEnv(track, n, device=dev, **kw), pkg.DroneVecEnv(track, n, device=dev, **kw)
    a.reset(); b.reset()
    f32 = torch.float32
    mk = lambda *shape, dt=f32: torch.zeros(shape, dtype=dt, device=dev)        # noqa: E731
    bufs = [dict(obs=mk(n, 13), rew=mk(n), done=mk(n, dt=torch.uint8), trunc=mk(n, dt=torch.uint8), found=mk(n, dt=torch.int32),
                 act=mk(n, 4), logp=mk(n), clipped=mk(n, 4)) for _ in range(2)]
    log_std = (C.c_float * 4)(-1.0, -2.0, -0.5, -3.0)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    g = torch.Generator(device="cpu").manual_seed(3)
    for t in range(30):
        mean = (torch.rand((n, 4), generator=g) * 0.4 - 0.1).to(dev)
        A, B = bufs
        _capi.check(lib.dn_policy_sample(a._handle, mean.data_ptr(), log_std, 99, deterministic, A["act"].data_ptr(),
                                         A["clipped"].data_ptr(), A["logp"].data_ptr(), stream))
        _capi.check(lib.dn_step(a._handle, A["clipped"].data_ptr(), A["obs"].data_ptr(), A["rew"].data_ptr(), A["done"].data_ptr(),
                                A["trunc"].data_ptr(), A["found"].data_ptr(), None, None, None, None, stream))
        _capi.check(lib.dn_step_sampled(b._handle, mean.data_ptr(), log_std, 99, deterministic, B["act"].data_ptr(), B["logp"].data_ptr(),
                                        B["obs"].data_ptr(), B["rew"].data_ptr(), B["done"].data_ptr(), B["trunc"].data_ptr(),
                                        B["found"].data_ptr(), None, None, None, None, stream))
        torch.cuda.synchronize()
        for k in ("act", "logp", "obs", "rew", "done", "trunc", "found"):
            assert torch.equal(A[k], B[k]), (t, k)
        if deterministic:
            assert torch.equal(A["act"], mean)
    sa, sb = a.get_state(), b.get_state()
    for k in sa.dtype.names:
        assert np.ascontiguousarray(sa[k]).tobytes() == np.ascontiguousarray(sb[k]).tobytes(), k
    a.close(); b.close()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 65535, 65536, 65537, 200001, 2097152])
def test_done_compaction_is_ordered_at_every_size(n):
    """dn_compact_done on random ballot words: the index list equals numpy's nonzero (ascending, bit-exact) from one drone
    to 2 M (32 workgroups; sizes straddle the 64-drone word and the 65 536-drone workgroup boundaries), dense and sparse."""
    _gpu()
    import ctypes as C
    from drl_dronenavigation_amd import _capi
    lib = _capi.load()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n)
    words = (n + 63) // 64
    for density in (0.5, 0.003, 0.0):
        bits = rng.random(words * 64) < density
        bits[n:] = False                                   # the step kernels never set bits beyond the fleet
        packed = np.packbits(bits.astype(np.uint8), bitorder="little").view(np.uint64)
        mask = torch.from_numpy(packed.view(np.int64)).to(dev)
        idx = torch.full((n,), -1, dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        _capi.check(lib.dn_compact_done(mask.data_ptr(), n, idx.data_ptr(), cnt.data_ptr(), 0,
                                        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        torch.cuda.synchronize()
        want = np.nonzero(bits[:n])[0].astype(np.int32)
        assert int(cnt.item()) == want.size
        assert np.array_equal(idx[:want.size].cpu().numpy(), want)
        assert bool((idx[want.size:] == -1).all())         # nothing written past the count


@pytest.mark.parametrize("shape", ["4", "1"])
@pytest.mark.parametrize("n", [1, 33, 4096 + 5])
def test_sac_actor_kernel_matches_the_float32_torch_actor(n, shape, monkeypatch):
    """dn_mlp_forward with arch = DN_MLP_ARCH_SAC (obs -> 256 -> 256 -> mu | log_std, ReLU; PBDroneSimulator.py:297-303)
    against the float64 evaluation of the float32 SacActor: fp32 grade <= 1e-4 on mu and log_std, the bf16 grade an order
    coarser; ragged tile sizes; masked forward; and FusedSacActor's deterministic action = tanh(mu)."""
    pkg = _gpu()
    from drl_dronenavigation_amd import policy_mfma as pm
    monkeypatch.setenv("DN_MLP_SAC_SHAPE", shape)             # 4: four waves share the weight stream through LDS (default); 1: one wave, from L2
    dev = torch.device("cuda:0")
    torch.manual_seed(n + 11)
    actor = pkg.SacActor().to(dev)
    with torch.no_grad():
        for p_ in actor.parameters():
            p_.add_(0.03 * torch.randn_like(p_))
    obs = (torch.rand(n, 13, device=dev) * 2 - 1) * torch.tensor([1, 1, 1, 1, 1, 1, 1, 1, 0.33, 1, 1, 1, 1], device=dev)
    obs[: min(n, 8)] *= 10.0
    lin = [l for l in actor.latent_pi if isinstance(l, torch.nn.Linear)]
    with torch.no_grad():
        h = obs.double()
        for l in lin:
            h = torch.relu(h @ l.weight.double().t() + l.bias.double())
        want = torch.cat((h @ actor.mu.weight.double().t() + actor.mu.bias.double(),
                          h @ actor.log_std.weight.double().t() + actor.log_std.bias.double()), 1)
    layers = [(l.weight, l.bias) for l in lin] + [(actor.mu.weight, actor.mu.bias), (actor.log_std.weight, actor.log_std.bias)]
    errs = {}
    for grade in ("fp32", "fp16", "bf16"):
        (got,) = pm.mlp_forward([pm.pack_sac_actor(layers, dev, grade)], obs)
        torch.cuda.synchronize()
        assert got.shape == (n, 8)
        errs[grade] = float((got.double() - want).abs().max())
    print(f"n={n}: SAC actor max |err| vs float64 -- fp32 grade {errs['fp32']:.2e}, fp16 grade {errs['fp16']:.2e}, bf16 grade {errs['bf16']:.2e}")
    assert errs["fp32"] <= 1e-4, errs
    assert errs["bf16"] <= 5e-2 and errs["bf16"] > 3 * errs["fp32"], errs
    assert errs["fp16"] <= 5e-3 and errs["fp16"] < 0.4 * errs["bf16"], errs
    fused = pkg.FusedSacActor(actor, n, dev, grade="fp32")
    mean, log_std = fused.mean_log_std(obs)
    assert float((mean.double() - want[:, :4]).abs().max()) <= 1e-4
    assert float(log_std.max()) <= 2.0 and float(log_std.min()) >= -20.0
    act = fused(obs, deterministic=True)
    assert float((act.double() - torch.tanh(want[:, :4])).abs().max()) <= 1e-4
    smp = fused(obs)
    assert smp.shape == (n, 4) and float(smp.abs().max()) <= 1.0
    mask = torch.zeros(n, dtype=torch.uint8, device=dev)
    mask[::61] = 1
    (mv,) = pm.mlp_forward([pm.pack_sac_actor(layers, dev, "fp32")], obs, row_mask=mask)
    torch.cuda.synchronize()
    sel = mask.bool()
    assert float((mv[sel].double() - want[sel]).abs().max()) <= 1e-4
    host = mask.bool().cpu().numpy()
    for t0 in range(0, n, 32):
        if not host[t0:t0 + 32].any():
            assert float(mv[t0:t0 + 32].abs().max()) == 0.0
    # the ABI refuses a mixed call
    with pytest.raises(pkg.DroneNavError):
        net = pkg.MlpActorCritic().to(dev)
        lin2 = [l for l in net.vf if isinstance(l, torch.nn.Linear)]
        vf = pm.pack_mlp([(l.weight, l.bias) for l in lin2] + [(net.value_net.weight, net.value_net.bias)], dev)
        pm.mlp_forward([pm.pack_sac_actor(layers, dev, "bf16"), vf], obs)


def test_squashed_sample_draws_the_policy_sample_normals():
    """dn_squashed_sample (SAC: a = tanh(mu + exp(clamp(log_std)) z), SB3 Actor [3P-recall]) uses the environment's Philox stream
    exactly as dn_policy_sample does: z is read off dn_policy_sample (mean 0, log_std 0 => action = z) and the squashed action and
    its log-probability are recomputed in torch float64; clamp edges, the deterministic action and sharding (env_id_offset)."""
    pkg = _gpu()
    import ctypes as C
    from drl_dronenavigation_amd import _capi
    lib = _capi.load()
    dev = torch.device("cuda:0")
    n = 4096 + 7
    env = pkg.DroneVecEnv(_tracks().reaching(), n, device=dev, env_id_offset=12345)
    env.reset_tensor()
    sptr = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    zero = torch.zeros(n, 4, device=dev)
    z, clipped, lp0 = torch.empty(n, 4, device=dev), torch.empty(n, 4, device=dev), torch.empty(n, device=dev)
    ls0 = (C.c_float * 4)(0.0, 0.0, 0.0, 0.0)
    _capi.check(lib.dn_policy_sample(env._handle, zero.data_ptr(), ls0, 77, 0, z.data_ptr(), clipped.data_ptr(), lp0.data_ptr(), sptr))
    g = torch.Generator(device="cpu").manual_seed(3)
    mls = torch.cat((torch.randn(n, 4, generator=g), torch.randn(n, 4, generator=g) * 1.5 - 0.5), 1).to(dev)
    mls[0, 4:] = 5.0                                           # above LOG_STD_MAX = 2
    mls[1, 4:] = -30.0                                         # below LOG_STD_MIN = -20
    act, lp = torch.empty(n, 4, device=dev), torch.empty(n, device=dev)
    _capi.check(lib.dn_squashed_sample(env._handle, mls.data_ptr(), 77, 0, act.data_ptr(), lp.data_ptr(), sptr))
    torch.cuda.synchronize()
    mu, ls = mls[:, :4].double(), mls[:, 4:].double().clamp(-20.0, 2.0)
    pre = mu + torch.exp(ls) * z.double()
    want = torch.tanh(pre)
    assert float((act.double() - want).abs().max()) <= 2e-6
    assert float(act.abs().max()) <= 1.0
    # SB3 evaluates the tanh correction in float32 from the squashed action itself (1 - a^2 loses digits near saturation): same here
    ls32 = mls[:, 4:].clamp(-20.0, 2.0)
    want_lp = (-0.5 * z * z - ls32 - 0.91893853320467274178).sum(1) - torch.log(1.0 - act * act + 1e-6).sum(1)
    assert float(((lp - want_lp).abs() / (1.0 + want_lp.abs())).max()) <= 1e-5
    unsat = want.abs().max(1).values < 0.99                                     # and against float64 where nothing saturates
    want_lp64 = (-0.5 * z.double() ** 2 - ls - 0.5 * np.log(2 * np.pi)).sum(1) - torch.log(1.0 - want ** 2 + 1e-6).sum(1)
    assert int(unsat.sum()) > n // 4
    assert float(((lp.double() - want_lp64).abs() / (1.0 + want_lp64.abs()))[unsat].max()) <= 1e-4
    det = torch.empty(n, 4, device=dev)
    _capi.check(lib.dn_squashed_sample(env._handle, mls.data_ptr(), 77, 1, det.data_ptr(), None, sptr))
    torch.cuda.synchronize()
    assert float((det.double() - torch.tanh(mu)).abs().max()) <= 2e-6
    # a shard that starts at another global drone id draws that drone's normals
    env2 = pkg.DroneVecEnv(_tracks().reaching(), 64, device=dev, env_id_offset=12345 + 1000)
    env2.reset_tensor()
    act2 = torch.empty(64, 4, device=dev)
    _capi.check(lib.dn_squashed_sample(env2._handle, mls[1000:1064].contiguous().data_ptr(), 77, 0, act2.data_ptr(), None, sptr))
    torch.cuda.synchronize()
    assert torch.equal(act2, act[1000:1064])
    env.close(); env2.close()
