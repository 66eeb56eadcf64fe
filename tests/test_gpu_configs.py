"""BASELINE configs 4 and 5 at their full per-GPU size on ONE MI355X, and the RCCL code path of bench.py with one rank.

  config 4: num_envs = 131072 = 4 x 32768, PPO rollout with an all-gather of advantages
  config 5: num_envs = 262144 = 8 x 32768, SAC collection with action / observation noise
The driver owns the multi-GPU runs; what one box can settle is (a) that the per-rank shards of the full fleet compute,
drone for drone, the bits of the unsplit fleet (env_id_offset keys the Philox streams by GLOBAL drone id),
(b) that a full-size shard's off-policy collection matches the oracle on sampled drone ranges, and (c) that
`bench.py --force-dist --ppo-sharded` initialises RCCL, runs its barriers and collectives in order and prints ONE JSON
line (a one-rank process group: every collective call of the N > 1 path executes, over a trivial group)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import oracle as O  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pkg():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: the HIP path has no CPU fallback")
    import drl_dronenavigation_amd as pkg
    return pkg


def _mixed(rng, n):
    bang = rng.uniform(-1, 1, (n, 4))
    hover = 0.0922 + 0.003 * rng.standard_normal((n, 4))
    return np.where((np.arange(n) % 2 == 0)[:, None], bang, hover).astype(np.float32)


def test_config5_eight_shards_of_32768_equal_the_whole_fleet_with_noise(monkeypatch):
    """262144 drones with action and observation noise: eight 32768-drone shards (what the eight ranks own) against
    the unsplit fleet, fused and single-step launches, bit for bit, final state included."""
    pkg = _pkg()
    from drl_dronenavigation_amd import tracks
    monkeypatch.delenv("DN_WAVES", raising=False)
    n, R, K = 262144, 8, 4
    m = n // R
    track = tracks.reaching()
    kw = dict(normalize_obs=False, max_steps=12, act_noise_sigma=0.01, obs_noise_sigma=0.02, seed=2026)
    whole = pkg.DroneVecEnv(track, n, device="cuda:0", **kw)
    parts = [pkg.DroneVecEnv(track, m, device="cuda:0", env_id_offset=r * m, **kw) for r in range(R)]
    assert torch.equal(whole.reset_tensor(), torch.cat([p.reset_tensor() for p in parts]))
    rng = np.random.default_rng(5)
    dev = torch.device("cuda:0")
    n_done = 0
    for rep in range(3):
        acts = torch.from_numpy(np.stack([_mixed(rng, n) for _ in range(K)])).to(dev)
        a = whole.rollout_tensor(acts)
        bs = [p.rollout_tensor(acts[:, r * m:(r + 1) * m].contiguous()) for r, p in enumerate(parts)]
        for k in ("obs", "reward", "done", "truncated", "found_targets"):
            assert torch.equal(a[k], torch.cat([b[k] for b in bs], dim=1)), (k, rep)
        n_done += int(a["done"].sum())
        one = torch.from_numpy(_mixed(rng, n)).to(dev)
        o, r_, d, _ = whole.step_tensor(one)
        o, r_, d = o.clone(), r_.clone(), d.clone()
        ps = [p.step_tensor(one[r * m:(r + 1) * m].contiguous()) for r, p in enumerate(parts)]
        assert torch.equal(o, torch.cat([x[0] for x in ps])) and torch.equal(r_, torch.cat([x[1] for x in ps]))
        assert torch.equal(d, torch.cat([x[2] for x in ps]))
    assert n_done >= n // 2                                    # max_steps = 12: the auto-reset path ran fleet-wide
    sw = whole.get_state()
    sp = np.concatenate([p.get_state() for p in parts])
    for k in sw.dtype.names:
        assert np.ascontiguousarray(sw[k]).tobytes() == np.ascontiguousarray(sp[k]).tobytes(), k
    ew, es = whole.stats(), [p.stats() for p in parts]
    assert ew["episodes"] == sum(e["episodes"] for e in es) and ew["env_steps"] == sum(e["env_steps"] for e in es)
    whole.close()
    for p in parts:
        p.close()


def test_config5_off_policy_collection_on_a_full_size_shard_matches_the_oracle():
    """Rank 3 of 8 in config 5 (32768 drones, env_id_offset = 3 x 32768, Philox action / observation noise): the
    OffPolicyCollector's replay buffer against the oracle on two sampled drone ranges (the oracle's drones are
    independent worlds keyed by global drone id, so a contiguous sub-range can be replayed on its own)."""
    pkg = _pkg()
    from drl_dronenavigation_amd import tracks
    from drl_dronenavigation_amd.collector import OffPolicyCollector
    track = tracks.reaching()
    n, T, rank = 32768, 48, 3
    kw = dict(max_steps=30, normalize_obs=False, act_noise_sigma=0.002, obs_noise_sigma=0.01, seed=9)
    env = pkg.DroneVecEnv(track, n, device="cuda:0", env_id_offset=rank * n, **kw)
    dev = env.device
    g = torch.Generator(device="cpu").manual_seed(5)
    pattern = torch.sign(torch.randn(n, 4, generator=g)).to(dev)
    w = (torch.randn(13, 4, generator=g) * 0.02).to(dev)
    hover = (torch.arange(n, device=dev) % 2 == 1)[:, None]

    def actor(obs):
        return torch.where(hover, torch.full((n, 4), 0.0922, device=dev) + obs @ w * 0.001, pattern * 1.3)

    col = OffPolicyCollector(env, actor, buffer_size=T)
    buf = col.collect(T)
    assert len(buf) == T * n and buf.full
    seen_done = seen_timeout = 0
    for lo in (1000, n - 256):
        sl = slice(lo, lo + 256)
        cfg = O.make_config(track.targets(), track.initial_xyzs, track.aviary_dim, circle=False, f32_state=True,
                            env_id_offset=rank * n + lo, **kw)
        ora = O.OracleVecEnv(cfg, 256, threads=4)
        obs_ref = ora.reset()
        for t in range(T):
            np.testing.assert_allclose(buf.obs[t, sl].cpu().numpy(), obs_ref, rtol=0, atol=1e-5, err_msg=f"obs t={t} lo={lo}")
            ref = ora.step(buf.actions[t, sl].cpu().numpy())
            dn = ref["done"].astype(bool)
            want_next = np.where(dn[:, None], ref["terminal_obs"], ref["obs"])
            np.testing.assert_allclose(buf.next_obs[t, sl].cpu().numpy(), want_next, rtol=0, atol=1e-5)
            np.testing.assert_allclose(buf.rewards[t, sl].cpu().numpy(), ref["reward"], rtol=1e-5, atol=1e-4)
            assert np.array_equal(buf.dones[t, sl].cpu().numpy().astype(bool), dn)
            assert np.array_equal(buf.timeouts[t, sl].cpu().numpy().astype(bool), ref["truncated"].astype(bool))
            seen_done += int(dn.sum())
            seen_timeout += int(ref["truncated"].sum())
            obs_ref = ref["obs"]
    assert seen_done > 256 and seen_timeout > 0
    env.close()


def test_bench_rccl_path_with_one_rank():
    """`bench.py --force-dist --ppo-sharded` in a child process: RCCL init with device_id, the contract's barriers, the
    per-rollout all-gather of advantages / returns inside FusedRolloutCollector(gather=True), the flush-and-barrier
    before the JSON line, destroy_process_group -- the N > 1 code path of the bench on the driver's box, over a one-rank
    group.  The scaling curve itself is the driver's to measure (SCALE_rNN.json)."""
    _pkg()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "64", "--warmup", "8", "--force-dist",
           "--ppo-sharded", "--no-ppo-rollout", "--no-cpu-baseline"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["metric"] == "env_steps_per_sec" and line["n_gpus"] == 1 and line["steps"] == 64 and line["scaling"] == "weak"
    assert line["value"] > 1e8 and 0 < line["roofline"]["frac"] <= 1.0
    assert line["roofline"]["traffic"] and 0 < line["roofline"]["traffic_frac"] <= 1.0
    sh = line["ppo_rollout_sharded"]
    assert sh["global_num_envs"] == 32768 and sh["value"] > 1e7 and sh["all_gather_bytes_sent_per_rank_per_rollout"] == 2 * 32 * 32768 * 4


def test_config5_collection_with_the_fused_sac_actor():
    """BASELINE config 5's loop as the reference's SAC agent runs it -- actor network in the loop (PBDroneSimulator.py:297-338):
    FusedSacActor (MFMA kernel, fp32 grade) -> tanh-squashed actions -> dn_step with action / observation noise ->
    ReplayBuffer.  The transitions are replayed through the oracle from the stored actions on a sampled range of drones."""
    pkg = _pkg()
    from drl_dronenavigation_amd import tracks
    from drl_dronenavigation_amd.collector import OffPolicyCollector
    track = tracks.reaching()
    n, T, rank = 32768, 24, 5
    kw = dict(max_steps=20, normalize_obs=False, act_noise_sigma=0.002, obs_noise_sigma=0.01, seed=3)
    env = pkg.DroneVecEnv(track, n, device="cuda:0", env_id_offset=rank * n, **kw)
    torch.manual_seed(0)
    actor = pkg.SacActor().to(env.device)
    fused = pkg.FusedSacActor(actor, n, env.device, grade="fp32")
    col = OffPolicyCollector(env, fused, buffer_size=T)
    buf = col.collect(T)
    assert len(buf) == T * n and buf.full
    assert float(buf.actions.abs().max()) <= 1.0 and float(buf.actions.std()) > 0.3      # squashed Gaussian, not a constant
    # the stored action of a deterministic call is tanh(mu) of the float32 actor on the stored observation
    with torch.no_grad():
        mu, _ = actor.mean_log_std(buf.obs[T - 1])
    assert float((fused(buf.obs[T - 1], deterministic=True) - torch.tanh(mu)).abs().max()) <= 2e-4
    lo = 7777
    sl = slice(lo, lo + 128)
    cfg = O.make_config(track.targets(), track.initial_xyzs, track.aviary_dim, circle=False, f32_state=True,
                        env_id_offset=rank * n + lo, **kw)
    ora = O.OracleVecEnv(cfg, 128, threads=4)
    obs_ref = ora.reset()
    done_seen = 0
    for t in range(T):
        np.testing.assert_allclose(buf.obs[t, sl].cpu().numpy(), obs_ref, rtol=0, atol=1e-5, err_msg=f"obs t={t}")
        ref = ora.step(buf.actions[t, sl].cpu().numpy())
        dn = ref["done"].astype(bool)
        np.testing.assert_allclose(buf.next_obs[t, sl].cpu().numpy(), np.where(dn[:, None], ref["terminal_obs"], ref["obs"]), rtol=0, atol=1e-5)
        np.testing.assert_allclose(buf.rewards[t, sl].cpu().numpy(), ref["reward"], rtol=1e-5, atol=1e-4)
        assert np.array_equal(buf.dones[t, sl].cpu().numpy().astype(bool), dn)
        done_seen += int(dn.sum())
        obs_ref = ref["obs"]
    assert done_seen > 0
    env.close()


def test_off_policy_cycle_under_hipgraph_equals_the_eager_loop():
    """OffPolicyCollector.collect_cycle(): the third pass (a hipGraph replay) leaves the replay buffer bit-identical to three
    eager passes of the same loop -- actor kernel, dn_squashed_sample (Philox draw keyed by the device-side step counter),
    dn_step with Philox action / observation noise, every output written in place into the ring."""
    pkg = _pkg()
    from drl_dronenavigation_amd import tracks
    from drl_dronenavigation_amd.collector import OffPolicyCollector
    track = tracks.reaching()
    n, T = 4096, 16
    kw = dict(max_steps=12, normalize_obs=True, act_noise_sigma=0.002, obs_noise_sigma=0.01, seed=4)
    torch.manual_seed(2)
    actor = pkg.SacActor().to("cuda:0")
    bufs = []
    for mode in ("eager", "graph"):
        env = pkg.DroneVecEnv(track, n, device="cuda:0", **kw)
        fused = pkg.FusedSacActor(actor, n, env.device, grade="fp32")
        col = OffPolicyCollector(env, fused, buffer_size=T, seed=11)    # the direct loop: three launches per step, Philox draws
        for _ in range(3):
            if mode == "eager":
                col.collect(T)
            else:
                col.collect_cycle()
        torch.cuda.synchronize()
        assert col.num_timesteps == 3 * T * n and col.buffer.full and col.buffer.pos == 0
        assert col.direct
        bufs.append({k: getattr(col.buffer, k).clone() for k in ("obs", "next_obs", "actions", "rewards", "dones", "timeouts")})
        if mode == "graph":
            assert col._graph is not None
        env.close()
    for k in bufs[0]:
        assert torch.equal(bufs[0][k], bufs[1][k]), k
    assert float(bufs[0]["dones"].sum()) >= n                                  # episodes ended and restarted inside the cycle
    assert float(bufs[0]["actions"].std()) > 0.3                               # sampled, not the deterministic mean


def test_ring_replay_buffer_across_wraps_matches_the_oracle():
    """The direct SAC loop writes in place into RingReplayBuffer; after 2.5 passes over a 10-slot ring every stored transition
    (the newest 10 steps, wrapped) must still read as SB3's (obs, next_obs, action, reward, done, timeout): replayed through
    the oracle from the stored actions -- the row that carries the newest observation over the wrap is the delicate one."""
    pkg = _pkg()
    from drl_dronenavigation_amd import tracks
    from drl_dronenavigation_amd.collector import OffPolicyCollector, RingReplayBuffer
    track = tracks.reaching()
    n, T, steps = 256, 10, 25
    kw = dict(max_steps=7, normalize_obs=False, act_noise_sigma=0.0, obs_noise_sigma=0.0, seed=1)
    env = pkg.DroneVecEnv(track, n, device="cuda:0", **kw)
    torch.manual_seed(5)
    actor = pkg.SacActor().to(env.device)
    col = OffPolicyCollector(env, pkg.FusedSacActor(actor, n, env.device, grade="fp32"), buffer_size=T, seed=2)
    assert isinstance(col.buffer, RingReplayBuffer)
    hist = []                                                      # every action the loop took, in order
    for _ in range(steps):
        p = col.buffer.pos
        col.collect(1)
        hist.append(col.buffer.actions[p].cpu().numpy().copy())
    buf = col.buffer
    assert buf.full and buf.pos == steps % T and len(buf) == (T - 1) * n      # mid-cycle: the slot being replaced is not whole
    assert buf.valid_slots() == [t for t in range(T) if t != buf.pos]
    slots, _ = buf._sample_slots(4096)
    assert buf.pos not in set(slots.cpu().tolist()) and set(slots.cpu().tolist()) == set(buf.valid_slots())
    cfg = O.make_config(track.targets(), track.initial_xyzs, track.aviary_dim, circle=False, f32_state=True, **kw)
    ora = O.OracleVecEnv(cfg, n, threads=4)
    obs_ref = ora.reset()
    obs_b, next_b, rew_b = buf.obs.cpu().numpy(), buf.next_obs.cpu().numpy(), buf.rewards.cpu().numpy()
    done_b, to_b = buf.dones.cpu().numpy().astype(bool), buf.timeouts.cpu().numpy().astype(bool)
    checked = 0
    for t in range(steps):
        ref = ora.step(hist[t])
        if t >= steps - T and t % T in buf.valid_slots():          # still in the ring, at slot t % T
            s_ = t % T
            dn = ref["done"].astype(bool)
            np.testing.assert_allclose(obs_b[s_], obs_ref, rtol=0, atol=1e-5, err_msg=f"obs of step {t} (slot {s_})")
            np.testing.assert_allclose(next_b[s_], np.where(dn[:, None], ref["terminal_obs"], ref["obs"]), rtol=0, atol=1e-5)
            np.testing.assert_allclose(rew_b[s_], ref["reward"], rtol=1e-5, atol=1e-4)
            assert np.array_equal(done_b[s_], dn) and np.array_equal(to_b[s_], ref["truncated"].astype(bool))
            checked += 1
        obs_ref = ref["obs"]
    assert checked == T - 1
    batch = buf.sample(512)
    assert batch["obs"].shape == (512, 13) and batch["next_obs"].shape == (512, 13) and batch["dones"].max() <= 1.0
    env.close()


def test_step_squashed_equals_squashed_sample_then_step():
    """dn_step_squashed (the draw inside the step kernel) against dn_squashed_sample + dn_step: the two collection loops leave
    bit-identical replay rings -- one-wave and three-wave single-step kernels, with and without noise / normaliser."""
    pkg = _pkg()
    from drl_dronenavigation_amd import tracks
    from drl_dronenavigation_amd.collector import OffPolicyCollector
    track = tracks.reaching()
    torch.manual_seed(8)
    actor = pkg.SacActor().to("cuda:0")
    for n, kw in ((4096, dict(max_steps=9, normalize_obs=True, act_noise_sigma=0.002, obs_noise_sigma=0.01, seed=4)),
                  (2 * 65536 + 64, dict(max_steps=9, normalize_obs=False, seed=4))):          # the large fleet takes the one-wave kernel
        rings = []
        for fused_sample in (True, False):
            env = pkg.DroneVecEnv(track, n, device="cuda:0", **kw)
            fused = pkg.FusedSacActor(actor, n, env.device, grade="bf16")
            col = OffPolicyCollector(env, fused, buffer_size=12, seed=21, fused_sample=fused_sample)
            assert col._fused_sample == fused_sample
            col.collect(12)
            torch.cuda.synchronize()
            b = col.buffer
            rings.append([t.clone() for t in (b.obs_ring, b.terminal_obs, b.actions, b.rewards, b.done_flags, b.timeout_flags)])
            env.close()
        for a, b in zip(*rings):
            assert torch.equal(a, b)
        assert float(rings[0][4].sum()) > 0
