"""Round 5: the measurement hook of ABI 8 (dn_set_launch_events), the copy / pickle behaviour of sparse infos and the lazily
allocated host mirrors of the NumPy surface.  All through the C ABI / DroneVecEnv on a real MI355X (`-m gpu`)."""
import copy
import ctypes as C
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _gpu():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: the HIP path has no CPU fallback")
    import drl_dronenavigation_amd as pkg
    return pkg


def test_launch_events_time_the_kernel_alone_and_change_nothing():
    """dn_set_launch_events: the two hipEvents ride on the step kernel's own dispatch (one shot).  Their elapsed time is the kernel's
    duration: positive, below what a pair of marker events around the same call sees, and the launch's results are those of an
    un-instrumented twin, bit for bit.  A second launch without re-arming leaves the events untouched."""
    pkg = _gpu()
    from drl_dronenavigation_amd import tracks
    dev = torch.device("cuda:0")
    n, K = 32768, 20
    envs = [pkg.DroneVecEnv(tracks.reaching(), n, normalize_obs=True, seed=3, device=dev) for _ in range(2)]
    for e in envs:
        e.reset_tensor()
    torch.manual_seed(1)
    acts = torch.rand((K, n, 4), device=dev) * 2 - 1
    lib = pkg._capi.load()
    stream = torch.cuda.current_stream(dev)
    k0, k1, m0, m1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    for ev in (k0, k1, m0, m1):
        ev.record(stream)                                  # a torch event creates its hipEvent on first use
    torch.cuda.synchronize()
    for _ in range(3):                                     # warm both
        a = envs[0].rollout_tensor(acts)
        b = envs[1].rollout_tensor(acts)
    torch.cuda.synchronize()
    pkg._capi.check(lib.dn_set_launch_events(envs[0]._handle, C.c_void_p(k0.cuda_event), C.c_void_p(k1.cuda_event)))
    m0.record(stream)
    a = envs[0].rollout_tensor(acts)
    m1.record(stream)
    b = envs[1].rollout_tensor(acts)
    torch.cuda.synchronize()
    kernel_us, marker_us = k0.elapsed_time(k1) * 1e3, m0.elapsed_time(m1) * 1e3
    assert 5.0 < kernel_us < marker_us, (kernel_us, marker_us)         # a 20-step launch of 32 768 drones is ~36 us of kernel
    for key in a:
        assert torch.equal(a[key], b[key]), key
    # one shot: the next launch does not touch the events
    a2 = envs[0].rollout_tensor(acts)
    torch.cuda.synchronize()
    assert abs(k0.elapsed_time(k1) * 1e3 - kernel_us) < 1e-6
    # the single-step launch takes the hook as well
    pkg._capi.check(lib.dn_set_launch_events(envs[0]._handle, C.c_void_p(k0.cuda_event), C.c_void_p(k1.cuda_event)))
    envs[0].step_tensor(acts[0])
    torch.cuda.synchronize()
    assert 1.0 < k0.elapsed_time(k1) * 1e3 < kernel_us
    assert lib.dn_set_launch_events(None, None, None) != 0            # loud on a NULL handle
    for e in envs:
        e.close()


def test_sparse_infos_snapshot_on_copy_deepcopy_and_pickle():
    """an info of `info_mode="sparse"` is a live view (its answers follow the next step; it references the whole
    DroneVecEnv).  copy(), copy.copy(), copy.deepcopy() and pickle must give a plain dict with every key of the step, detached."""
    pkg = _gpu()
    from drl_dronenavigation_amd import tracks
    n = 256
    env = pkg.DroneVecEnv(tracks.reaching(), n, max_steps=5, device="cuda:0")           # sparse is the default
    ref = pkg.DroneVecEnv(tracks.reaching(), n, max_steps=5, device="cuda:0", info_mode="full")
    env.reset(); ref.reset()
    rng = np.random.default_rng(0)
    kept = []
    for t in range(8):
        a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        _, _, done, infos = env.step(a)
        _, _, done_f, infos_f = ref.step(a)
        assert np.array_equal(done, done_f)
        snaps = [infos[0].copy(), copy.copy(infos[1]), copy.deepcopy(infos[2]), pickle.loads(pickle.dumps(infos[3]))]
        for i, sdict in enumerate(snaps):
            assert type(sdict) is dict and "found_targets" in sdict and "TimeLimit.truncated" in sdict, (t, i, sdict)
            full = infos_f[i]
            assert sdict["found_targets"] == full["found_targets"] and sdict["TimeLimit.truncated"] == full["TimeLimit.truncated"]
            assert ("episode" in sdict) == ("episode" in full) == bool(done[i])
            if done[i]:
                np.testing.assert_array_equal(sdict["terminal_observation"], full["terminal_observation"])
                assert sdict["episode"]["l"] == full["episode"]["l"]
        kept.append((t, [dict(s) for s in snaps], [dict(found_targets=f["found_targets"]) for f in infos_f[:4]]))
    for t, snaps, fulls in kept:                          # the snapshots did not move with later steps
        for sdict, full in zip(snaps, fulls):
            assert sdict["found_targets"] == full["found_targets"], t
    assert any("episode" in s for _, snaps, _ in kept for s in snaps)     # max_steps = 5: episodes did end
    env.close(); ref.close()


def test_tensor_api_allocates_no_host_mirrors():
    """the pinned host mirrors belong to the NumPy surface; a tensor-only user (bench.py's 2 M-drone legs, the collectors)
    must not pay for them.  They appear with the first step()."""
    pkg = _gpu()
    from drl_dronenavigation_amd import tracks
    n = 4096
    env = pkg.DroneVecEnv(tracks.reaching(), n, device="cuda:0")
    env.reset_tensor()
    acts = torch.rand((4, n, 4), device="cuda:0") * 2 - 1
    env.step_tensor(acts[0])
    env.rollout_tensor(acts)
    torch.cuda.synchronize()
    assert env._mirrors is None
    env.reset()                                            # the NumPy reset does not need them either
    assert env._mirrors is None
    obs, rew, done, infos = env.step(acts[1].cpu().numpy())
    assert env._mirrors is not None and len(env._mirrors) == 1 and obs.shape == (n, 13) and len(infos) == n
    env.close()


def test_exact_observation_noise_switch(monkeypatch):
    """DN_EXACT_OBS_NOISE=1 (read by dn_create) draws the observation noise in the exact float64 Box-Muller form the
    dynamics-feeding draws use, for runs that must replay on another GPU generation or against the CPU definition.  With sigma = 1 the noisy
    reset observation is float32(clean + z): against the oracle's libm definition of z on the same Philox words at least 99.9 % of the
    values are bit-equal and none is more than one float32 ulp off; the default (hardware float32 transcendentals) is not that close; and
    the three-wave single step (draws shared out over the waves and, for reset observations, across the lanes) stays bit-identical to
    the one-wave kernel under the switch."""
    pkg = _gpu()
    from drl_dronenavigation_amd import tracks
    from oracle import oracle as O
    L = O.lib()
    track = tracks.circle(1, 4, 1)
    n, seed = 1 << 17, 99
    kw = dict(device="cuda:0", normalize_obs=False, seed=seed)
    clean = pkg.DroneVecEnv(track, n, **kw)
    o_clean = clean.reset_tensor().cpu().numpy()
    z_ref = np.zeros((4, n, 4), np.float32)
    for b in range(4):
        L.orc_noise4_many(seed, 0, n, 0, 5 + b, z_ref[b].ctypes.data_as(C.POINTER(C.c_float)))
    z_ref = np.transpose(z_ref, (1, 0, 2)).reshape(n, 16)[:, :13]
    o_ref = (o_clean + z_ref).astype(np.float32)           # float32 + float32, rounded once: what add_obs_noise computes
    equal = {}
    for exact in ("1", "0"):
        monkeypatch.setenv("DN_EXACT_OBS_NOISE", exact)
        noisy = pkg.DroneVecEnv(track, n, obs_noise_sigma=1.0, **kw)
        o_dev = noisy.reset_tensor().cpu().numpy()
        equal[exact] = float((o_dev.view(np.uint32) == o_ref.view(np.uint32)).mean())
        if exact == "1":
            ulps = np.abs(o_dev.view(np.int32).astype(np.int64) - o_ref.view(np.int32).astype(np.int64))
            assert ulps.max() <= 1, int(ulps.max())
        noisy.close()
    assert equal["1"] >= 0.999 and equal["0"] < 0.9, equal
    clean.close()
    # shapes under the switch: single step on three waves against one wave, noise + normaliser, short episodes
    monkeypatch.setenv("DN_EXACT_OBS_NOISE", "1")
    envs = {}
    for shape in ("1", "3"):
        monkeypatch.setenv("DN_WAVES_SINGLE", shape)
        envs[shape] = pkg.DroneVecEnv(tracks.reaching(), 4096, max_steps=6, normalize_obs=True, obs_noise_sigma=0.02, act_noise_sigma=0.005,
                                      seed=5, device="cuda:0")
        envs[shape].reset_tensor()
    monkeypatch.delenv("DN_WAVES_SINGLE")
    monkeypatch.delenv("DN_EXACT_OBS_NOISE")
    assert envs["3"].kernel_waves(fused=False) == 3 and envs["1"].kernel_waves(fused=False) == 1
    torch.manual_seed(2)
    for t in range(20):
        a = torch.rand((4096, 4), device="cuda:0") * 2 - 1
        x, y = envs["1"].step_tensor(a), envs["3"].step_tensor(a)
        for k in range(3):
            assert torch.equal(x[k], y[k]), (t, k)
        d = x[2].bool()
        assert torch.equal(x[3]["terminal_obs"][d], y[3]["terminal_obs"][d]), t
    for e in envs.values():
        e.close()
