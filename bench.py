#!/usr/bin/env python3
"""bench.py -- env-steps/s of the HIP drone-navigation environment on N MI355X of one node.

Contract (see the task brief): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver
launches it under torch.distributed.run (one rank per GPU, RCCL).  A "step" is one pass of the hot path
(dn_step: one kernel launch advancing every drone by one 240 Hz control step, auto-reset included) over
one batch of synthetic actions that is already resident in HBM.  W untimed steps, then EXACTLY K timed
steps bracketed by barrier + synchronize, MAX over ranks, one JSON line from rank 0.

Workload = BASELINE.json configs[2]: 32 768 drones per GPU on the 8-gate race track
(`Waypoints.reaching()`), norm_rew off, positional observation; actions U(-1,1)^4 float32 from a seeded
generator (the bang-bang regime that exercises the auto-reset path constantly).
The path shards embarrassingly (no data-path collective): weak scaling, 32 768 drones per rank.

Extra objects on the JSON line:
  roofline     -- HBM roofline of the kernel the timed region launches.  ALGORITHMIC bytes over the average
                  launch duration measured with HIP events on the launch stream over the timed region; peak 8 TB/s
                  (MI355X_MICROARCH.md).  Per drone a single-step launch (dn_step) moves 288 B (+432 B with the obs
                  normaliser); a fused K-step launch (dn_step_many) keeps the state in registers, so its algorithmic bytes
                  are 78 B per step (action 16 + outputs 62) + 210 B of state once per launch (+432 B of statistics once):
                  pricing a fused launch at 288 B per step would count bytes it never moves.  `traffic` = HBM bytes per
                  launch from the committed rocprofv3 --pmc passes (profiles/hbm_traffic.json; exact key, or the linear
                  model fitted to those passes for other K), `traffic_frac` the same fraction with measured bytes.
  single_step / normalize_obs_on -- the closed-loop launch (one dn_step per policy step) and the configuration the
                  reference actually runs (NormalizeObservation always on, PBDroneSimulator.py:181), each timed with HIP
                  events over its own >= 25 ms region, with kernel name, us, algorithmic and counter fractions.
  cpu_baseline -- the CPU oracle (a port; the reference's PyBullet path cannot run here) timed on this
                  box's host cores on a bounded sample of the same workload, rank 0, N = 1 only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 288          # state R+W 104, bookkeeping R+W 106, action 16, outputs 62
ALGO_BYTES_IO_PER_STEP = 78            # of which per step whatever the launch shape: action 16 + outputs 62
ALGO_BYTES_STATE = 210                 # and once per LAUNCH: state R+W 104 + bookkeeping R+W 106 (registers hold it between fused steps)
ALGO_BYTES_NORMALISER = 432            # per-drone NormalizeObservation statistics, 27 float64 R+W (once per launch)
HBM_PEAK_GBPS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PREROLL_SECONDS = 0.5                  # untimed fused stepping before the warm-up: the GPU clock needs ~50 ms of load to settle


def algo_bytes_per_launch(n, steps_per_launch, norm):
    """Algorithmic HBM bytes of one launch over n drones."""
    return n * (ALGO_BYTES_IO_PER_STEP * steps_per_launch + ALGO_BYTES_STATE + (ALGO_BYTES_NORMALISER if norm else 0))


def kernel_name(waves, dtype, norm, fused):
    r, nm = ("double" if dtype == "float64" else "float"), ("true" if norm else "false")
    if waves == 8 and fused:
        return f"dn_step_many_rp8_kernel<{r}>"
    if waves == 5 and fused:
        return f"dn_step_many_5w_kernel<{r}, false>"
    if waves == 4 and fused:
        return f"dn_step_many_4w_kernel<{r}, {nm}, false>"
    if waves >= 3 and fused:
        return f"dn_step_many_3w_kernel<{r}, {nm}, false, false>"
    if waves >= 3:
        return f"dn_step_pqx_kernel<{r}, {nm}, false, false>"
    return f"dn_step_many_{waves}w_kernel<{r}, {nm}, false, {'false' if fused else 'true'}, false, false>"


def traffic_per_launch(track, n, dtype, norm, fused_steps, waves):
    """HBM bytes per launch from the committed PMC passes: the exact key if that launch was profiled, else the linear
    model bytes = n (a K + b) fitted to the profiled fused launches of the same configuration (state once per launch,
    I/O per step).  Returns (bytes or None, source string)."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        tj = json.load(open(path))
    except Exception:  # noqa: BLE001
        return None, "profiles/hbm_traffic.json missing"
    base = f"{track}_{n}_{dtype}" + ("_norm" if norm else "")
    key = base + ("_single" if fused_steps == 0 else f"_fused{fused_steps}") + f"_{waves}w"
    if key in tj:
        return tj[key]["bytes_per_launch"], f"pmc:{key}"
    if fused_steps:
        model = tj.get("_fused_model", {}).get(base)
        if model:
            return int(n * (model["per_step"] * fused_steps + model["per_launch"])), f"model:{base} ({model['per_step']} B x K + {model['per_launch']} B per drone, fitted to {model['fitted_on']})"
    return None, f"no PMC pass for {key}"


def issue_evidence(kernel):
    """The committed instruction-mix counters of the kernel that was actually timed (profiles/instmix.json), or None."""
    try:
        ev = json.load(open(os.path.join(ROOT, "profiles", "instmix.json"))).get(kernel)
    except Exception:  # noqa: BLE001
        return None
    return dict(ev, kernel=kernel) if ev else None


SHADER_CLOCK_GHZ_ASSUMED = 2.14         # fallback only: 74 504 cycles (s_memtime) in 34.84 us of the 100 MHz wall clock, stamped build of the five-wave
                                        # kernel in round 5.  The committed counter passes carry their own clock (GRBM_GUI_ACTIVE / kernel duration).


def valu_bound(kernel, n, steps_per_launch, us_per_step, num_cus=256, regime=None):
    """The bound that governs the fused step at 32 768 drones, from the committed SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU pass of the
    kernel that was timed (profiles/instmix.json, re-keyed every round): instructions per 64-drone tile-step x ALU cycles per
    instruction x tiles per CU / 4 SIMDs / shader clock = the time a step would take if the vector ALUs never idled.  The shader
    clock is the one of the counter pass itself (GRBM_GUI_ACTIVE cycles / the kernel's duration in that pass) when the pass has it;
    otherwise an assumed figure, labelled as such.  `regime`: a named sub-entry of the kernel's record (e.g. "hover_band_k20")."""
    ev = issue_evidence(kernel)
    if not ev or "alu_cycles_per_valu_instruction" not in ev:
        return None
    if regime:
        src = ev.get(regime)
        if not src:
            return None
    else:
        src = ev.get("driver_launch_k20") if steps_per_launch == 20 and ev.get("driver_launch_k20") else ev
    insts = src["valu_instructions_per_64_drone_step"]
    cyc = src["alu_cycles_per_valu_instruction"]
    clock = src.get("shader_clock_ghz")
    clock_src = src.get("shader_clock_source", "GRBM_GUI_ACTIVE / kernel duration of the counter pass") if clock else "ASSUMED (round 5's stamped build); no GRBM_GUI_ACTIVE in the committed pass"
    clock = clock or SHADER_CLOCK_GHZ_ASSUMED
    tiles_per_cu = ((n + 63) // 64) / num_cus
    floor_us = tiles_per_cu * insts * cyc / 4.0 / (clock * 1e3)
    return {"valu_insts_per_tile_step": insts, "valu_cycles_per_inst": cyc, "tiles_per_cu": round(tiles_per_cu, 3),
            "shader_clock_ghz": clock, "shader_clock_source": clock_src,
            "valu_floor_us_per_step": round(floor_us, 4), "valu_frac": round(floor_us / us_per_step, 4),
            "counters_from": src.get("source") or ev.get("source"),
            "what": "vector-ALU floor = tiles per CU x instructions per tile-step x ALU cycles per instruction / 4 SIMDs / shader clock; "
                    "valu_frac = floor / measured time per step (the fraction of the step during which the vector ALUs are busy)"}


# csrc/dn_action_sat.h: outside (ACT_SAT_LO, ACT_SAT_HI) the rescaled action is clipped to a thrust bound (rotor_force_sat's fast path)
ACT_SAT_LO, ACT_SAT_HI = 0.0899437964, 0.0971653908


def fast_path_hit_rate(torch, acts):
    """Fraction of (64-drone tile, rotor, step) groups of the resident action batches in which EVERY lane is saturated, i.e. in which the
    wave skips the float32 thrust chain (rotor_force_sat).  acts: [A, n, 4] float32 on the device."""
    a_, n_ = acts.shape[0], acts.shape[1]
    m = n_ // 64 * 64
    sat = (acts[:, :m] <= ACT_SAT_LO) | (acts[:, :m] >= ACT_SAT_HI)
    return float(sat.view(a_, m // 64, 64, 4).all(dim=2).float().mean().item())


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 0.13 s of timed work -- the GPU clock needs ~50 ms of load to settle (20 000 steps = 26 ms read 4 % slower)
    ap.add_argument("--steps", type=int, default=100000)
    ap.add_argument("--warmup", type=int, default=10000)
    ap.add_argument("--num-envs", type=int, default=32768, help="drones per GPU")
    ap.add_argument("--track", default="reaching")
    ap.add_argument("--compute-dtype", default="float64", choices=["float64", "float32"])
    ap.add_argument("--normalize-obs", dest="normalize_obs", action="store_true", default=True,
                    help="per-drone NormalizeObservation fused in (+432 B of statistics per drone and launch): the DEFAULT, because the "
                         "reference wraps every env in it (PBDroneSimulator.py:181)")
    ap.add_argument("--no-normalize-obs", dest="normalize_obs", action="store_false",
                    help="headline without the normaliser (otherwise reported as the `normalize_obs_off` sub-leg)")
    ap.add_argument("--mode", default="many", choices=["many", "single", "graph"],
                    help="many: one dn_step_many call; single: K python-level dn_step calls; graph: hipGraph replay")
    ap.add_argument("--action-batches", type=int, default=64, help="distinct resident action batches cycled")
    ap.add_argument("--actions", default="uniform", choices=["uniform", "hover"],
                    help="action distribution of the timed region (SURVEY 8(d) C2): uniform = U(-1,1)^4, BASELINE's stream (99.6 %% of actions saturate a "
                         "thrust bound; episodes of ~130 steps); hover = 0.0922 + 0.003 N(0,1), the un-saturated band a trained policy lives in (long "
                         "flights, every wave evaluates the float32 thrust chain).  The default run times uniform as `value` and hover as the "
                         "`hover_band` leg; --actions hover makes hover the timed region (the counter passes of that regime)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ppo-rollout", action="store_true", help="skip the policy-in-the-loop rollout measurement")
    ap.add_argument("--ppo-sharded", dest="ppo_sharded", action="store_true", default=None,
                    help="BASELINE configs[3]: also time the policy-in-the-loop rollout on every rank with the per-rollout RCCL "
                         "all-gather of advantages/returns.  On by default whenever WORLD_SIZE > 1 (the one collective the design "
                         "has must be seen by the scaling run); a watchdog ends the process with a non-zero code if it hangs")
    ap.add_argument("--no-ppo-sharded", dest="ppo_sharded", action="store_false")
    ap.add_argument("--sharded-timeout", type=float, default=240.0, help="seconds the sharded leg may take before the watchdog exits(3)")
    ap.add_argument("--profile-lite", action="store_true",
                    help="for rocprofv3 --pmc passes (every dispatch costs tens of ms there): no pre-roll, a few launches per leg, "
                         "no NumPy-surface legs; timings of such a run mean nothing, only the counters do")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group even for one rank (exercises the N > 1 code path on a 1-GPU box)")
    return ap.parse_args()


def cpu_baseline(track, num_envs, max_steps, seconds, normalize_obs=True):
    """Time the CPU oracle (test infrastructure used here only as the reported baseline) on all host cores, in the HEADLINE's
    configuration: the per-drone NormalizeObservation on or off as the timed GPU line has it (the reference always wraps it,
    PBDroneSimulator.py:181).  The other setting is reported beside it (`other_normaliser_setting`)."""
    import numpy as np
    from oracle import oracle as O
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        usable = os.cpu_count() or 1
    cfg = O.make_config(track.targets(), track.initial_xyzs, track.aviary_dim, circle=track.is_circle,
                        max_steps=max_steps, normalize_obs=normalize_obs)
    cfg_other = O.make_config(track.targets(), track.initial_xyzs, track.aviary_dim, circle=track.is_circle,
                              max_steps=max_steps, normalize_obs=not normalize_obs)
    rng = np.random.default_rng(1)
    acts = rng.uniform(-1, 1, (16, num_envs, 4)).astype(np.float32)

    def rate(threads, budget, cfg=cfg):
        env = O.OracleVecEnv(cfg, num_envs, threads=threads)
        env.reset()
        env.step(acts[0])
        t0 = time.perf_counter()
        env.step(acts[1])
        per = max(time.perf_counter() - t0, 1e-6)
        k = max(2, int(budget / per))
        t0 = time.perf_counter()
        done = 0
        for t in range(k):
            env.step(acts[t % len(acts)])
            done += 1
            if done >= 2 and time.perf_counter() - t0 > 1.5 * budget:      # a team that slowed down after its first step: stay inside the budget
                break
        dt = time.perf_counter() - t0
        return dict(value=num_envs * done / dt, steps=done, seconds=dt, threads=threads)

    # The visible CPU count can exceed what the container may actually use (a cgroup quota makes a 256-thread
    # OpenMP team slower than one thread), so probe a ladder of team sizes briefly and time the fastest one.
    ladder = sorted({t for t in (1, 2, 4, 8, 16, 32, 64, 128, 256, usable) if t <= usable})
    probe = {t: rate(t, 0.04 * seconds)["value"] for t in ladder}
    # a short probe can flatter an oversubscribed team (128 threads: 186 M in the probe, 15 M sustained on a shared host): the two best of
    # the ladder are both timed for a quarter of the budget and the better SUSTAINED rate is the baseline
    ranked = sorted(probe, key=probe.get, reverse=True)
    one = rate(1, 0.2 * seconds)
    cands = [rate(t, 0.25 * seconds) if t != 1 else one for t in ranked[:2]]
    top = max(cands, key=lambda d: d["value"])
    other = rate(top["threads"], 0.1 * seconds, cfg_other)
    return {"value": round(top["value"], 1), "unit": "env-steps/s", "cores": top["threads"],
            "kind": "port",
            "sample": f"{num_envs} drones x {top['steps']} vector steps ({top['seconds']:.1f} s) of the same "
                      f"workload (race track, U(-1,1)^4 actions, per-drone NormalizeObservation {'ON' if normalize_obs else 'OFF'} as in the "
                      f"timed GPU line) through oracle/dn_oracle.c (OpenMP team of {top['threads']}, float64); "
                      f"{usable} CPUs visible to the process",
            "normalize_obs": bool(normalize_obs),
            "other_normaliser_setting": {"normalize_obs": (not normalize_obs), "value": round(other["value"], 1), "cores": other["threads"],
                                         "steps": other["steps"]},
            "single_thread_value": round(one["value"], 1),
            "thread_ladder": {str(t): round(v, 1) for t, v in probe.items()}}

MFMA_PEAK_FLOPS = 2.5e15               # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak (the 5 PF headline includes 2:1 sparsity)
PPO_MACS_MFMA = 16 * 512 + 512 * 512 + 512 * 256 + 256 * 32      # per drone and network as the MFMA tiles see it (K, M padded to 16 / 32)
PPO_MACS = 13 * 512 + 512 * 512 + 512 * 256                      # + 256 * out_dim: the network's own multiply-adds
SAC_MACS_MFMA = 16 * 256 + 256 * 256 + 256 * 32
SAC_MACS = 13 * 256 + 256 * 256 + 256 * 8


def time_launches(torch, dev, fn, reps=200, warm=20):
    """Average duration of one launch of `fn`: `reps` launches captured ONCE into a hipGraph on a side stream, the graph replayed between two
    HIP events -- device time per launch, dispatch gap included, host launch rate excluded (a Python loop around a ctypes call issues one
    launch per ~10-18 us, longer than the small kernels themselves).  `fn` must enqueue on torch's CURRENT stream."""
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(warm):
            fn()
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(reps):
                fn()
        graph.replay()
        side.synchronize()
        best = float("inf")
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            graph.replay()
            e1.record(side)
            side.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    torch.cuda.current_stream(dev).wait_stream(side)
    return best


def mlp_kernel_leg(torch, dev, name, fn, n, macs_mfma, macs, nets, passes):
    us = time_launches(torch, dev, fn)
    mfma_flop = 2.0 * macs_mfma * n * nets * passes
    return {"kernel": name, "avg_us": round(us, 2), "mfma_flop": mfma_flop, "useful_flop": 2.0 * macs * n * nets,
            "mfma_tflops": round(mfma_flop / (us * 1e-6) / 1e12, 1), "mfma_frac": round(mfma_flop / (us * 1e-6) / MFMA_PEAK_FLOPS, 4),
            "useful_tflops": round(2.0 * macs * n * nets / (us * 1e-6) / 1e12, 1),
            "passes": passes, "what": "200 launches of this kernel alone replayed from one hipGraph between two HIP events; mfma_frac = MFMA flop issued "
                                      "(padded tiles; x3 for the split-bf16 float32 grade) / time / 2.5 PFLOP/s dense peak"}


def ppo_rollout(pkg, track, n, max_steps, dev, rank):
    """BASELINE configs[2] as the learner sees it: SB3-PPO-shaped rollout collection with the policy in the loop --
    n_steps x (MLP 13-512-512-256 actor + critic, Gaussian sample, clip, dn_step, truncation bootstrap) + GAE, every
    buffer on the GPU, the rollout replayed from a hipGraph.  Reported beside the headline, never as `value`."""
    import torch
    from drl_dronenavigation_amd.collector import FusedRolloutCollector, RolloutCollector
    n_steps = 32                                   # re-parameterised from 4096 (4096 x 32768 does not fit)
    torch.manual_seed(1 + rank)
    net = pkg.MlpActorCritic().to(dev)
    res = {}
    fused = pkg.FusedMlpPolicy(net, n, dev)
    fused32 = pkg.FusedMlpPolicy(net, n, dev, grade="fp32")
    fused16 = pkg.FusedMlpPolicy(net, n, dev, grade="fp16")
    for label, use_graph, trunk in (("eager", False, None), ("graph", True, None), ("graph_bf16", True, torch.bfloat16),
                                    ("graph_mfma", True, "mfma"), ("eager_mfma", False, "mfma"),
                                    ("fused_eager", False, "fused"), ("fused_graph", True, "fused"),
                                    ("fused_graph_fp16", True, "fused16"), ("fused_graph_fp32", True, "fused32"),
                                    ("one_launch_graph", True, "one"), ("one_launch_graph_fp32", True, "one32")):
        net.trunk_dtype = trunk if trunk not in ("mfma", "fused", "fused32", "fused16", "one", "one32") else None
        env = pkg.DroneVecEnv(track, n, max_steps=max_steps, normalize_obs=True, env_id_offset=rank * n, device=dev)
        if trunk in ("one", "one32"):
            col = FusedRolloutCollector(env, fused if trunk == "one" else fused32, n_steps, use_graph=use_graph, seed=1 + rank, one_launch=True)
        elif trunk == "fused32":
            col = FusedRolloutCollector(env, fused32, n_steps, use_graph=use_graph, seed=1 + rank)
        elif trunk == "fused16":
            col = FusedRolloutCollector(env, fused16, n_steps, use_graph=use_graph, seed=1 + rank)
        elif trunk == "fused":
            col = FusedRolloutCollector(env, fused, n_steps, use_graph=use_graph, seed=1 + rank)
        elif trunk == "mfma":
            col = RolloutCollector(env, fused, n_steps, value_fn=fused.predict_values, use_graph=use_graph)
        else:
            col = RolloutCollector(env, net, n_steps, value_fn=net.predict_values, use_graph=use_graph)
        for _ in range(3):
            col.collect()
        torch.cuda.synchronize(dev)
        reps, best = 10, 0.0                                # a window is 20-70 ms; the best of three (a host hiccup of a few ms would
        for _ in range(3):                                  # otherwise halve a 12 ms reading)
            t0 = time.perf_counter()
            for _ in range(reps):
                col.collect()
            torch.cuda.synchronize(dev)
            best = max(best, n * n_steps * reps / (time.perf_counter() - t0))
        res[label] = best
        env.close()
    # the kernels of the fused loop on their own (what rocprofv3 --kernel-trace --stats shows for the same launches: profiles/r03_*)
    from drl_dronenavigation_amd.policy_mfma import mlp_forward
    kernels = {}
    try:
        env = pkg.DroneVecEnv(track, n, max_steps=max_steps, normalize_obs=True, env_id_offset=rank * n, device=dev)
        obs = env.reset_tensor().clone()
        mean, val = torch.zeros((n, 4), device=dev), torch.zeros((n, 1), device=dev)
        shape = os.environ.get("DN_MLP_SHAPE", "4")        # the library's default: four waves per workgroup sharing the weight stream
        kn = {"8": "dn_mlp_pair_kernel<%s, NoTail>", "1": "dn_mlp_kernel"}.get(shape, "dn_mlp_lds_kernel<%s>")
        for gname, pol, kname, passes in (("bf16", fused, kn % "false" if "%" in kn else kn, 1), ("fp16", fused16, (kn if "%" in kn else "dn_mlp_lds_kernel<%s>") % "true", 1),
                                          ("fp32", fused32, "dn_mlp_x3_kernel<NoTail>", 3)):
            kernels["mlp_" + gname] = mlp_kernel_leg(torch, dev, kname, lambda pol=pol: mlp_forward([pol.pi, pol.vf], obs, [mean, val]),
                                                     n, PPO_MACS_MFMA, PPO_MACS + 256 * 2.5, 2, passes)
        lib, h = pkg._capi.load(), env._handle
        bufs = dict(act=torch.zeros((n, 4), device=dev), logp=torch.zeros(n, device=dev), obs=torch.zeros((n, 13), device=dev),
                    rew=torch.zeros(n, device=dev), done=torch.zeros(n, dtype=torch.uint8, device=dev),
                    trunc=torch.zeros(n, dtype=torch.uint8, device=dev), found=torch.zeros(n, dtype=torch.int32, device=dev),
                    term=torch.zeros((n, 13), device=dev))
        log_std = (C.c_float * 4)(0.0, 0.0, 0.0, 0.0)

        def step_sampled():
            pkg._capi.check(lib.dn_step_sampled(h, mean.data_ptr(), log_std, 1, 0, bufs["act"].data_ptr(), bufs["logp"].data_ptr(),
                                                bufs["obs"].data_ptr(), bufs["rew"].data_ptr(), bufs["done"].data_ptr(),
                                                bufs["trunc"].data_ptr(), bufs["found"].data_ptr(), bufs["term"].data_ptr(), None, None, None,
                                                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        us = time_launches(torch, dev, step_sampled)
        kernels["step_sampled"] = {"kernel": f"dn_step_pqx_kernel<double, true, false, true> ({env.kernel_waves(fused=False)} waves per tile)"
                                             if env.kernel_waves(fused=False) == 3 else "dn_step_many_1w_kernel<double, true, false, true, false, true>",
                                   "avg_us": round(us, 2), "what": "Gaussian draw + clip + log-probability + the control step, obs normaliser on; "
                                                                   "200 launches replayed from one hipGraph"}
        env.close()
    except Exception as exc:  # noqa: BLE001
        kernels["error"] = f"{type(exc).__name__}: {exc}"
    return {"value": round(res["fused_graph_fp32"], 1), "unit": "env-steps/s", "kernels": kernels,
            "policy": "the reference's float32 networks (PBDroneSimulator.py:251-286) as fused MFMA MLPs at float32 grade (dn_mlp_forward grade 1: "
                      "split-bf16 operands, three MFMAs per product; <= 1e-4 on the action mean against the float32 torch network) + dn_step_sampled "
                      "(Gaussian sample + step): two launches per step, the truncation bootstrap as one masked critic pass per rollout, hipGraph replay",
            "value_bf16_grade": round(res["fused_graph"], 1),
            "policy_bf16_grade": "the same loop with bf16 weights / activations, float32 accumulate (narrower arithmetic than the reference's networks: "
                                 "~9e-3 on the action mean): a named variant, never `value`",
            "value_fp16_grade": round(res["fused_graph_fp16"], 1),
            "policy_fp16_grade": "the same loop with float16 weights and activations (dn_mlp_forward grade 2): the bf16 grade's speed at an eighth "
                                 "of its rounding error (~1e-3 on the action mean against the float32 torch network)",
            "value_fp32_grade": round(res["fused_graph_fp32"], 1),
            "variants": {"torch fp32 eager": round(res["eager"], 1), "torch fp32 hipGraph": round(res["graph"], 1),
                         "torch bf16 trunks hipGraph": round(res["graph_bf16"], 1),
                         "fused MFMA policy, torch glue, eager": round(res["eager_mfma"], 1),
                         "fused MFMA policy, torch glue, hipGraph": round(res["graph_mfma"], 1),
                         "fused collector eager": round(res["fused_eager"], 1),
                         "fused collector hipGraph": round(res["fused_graph"], 1),
                         "fused collector hipGraph, fp16-grade networks": round(res["fused_graph_fp16"], 1),
                         "fused collector hipGraph, fp32-grade networks": round(res["fused_graph_fp32"], 1),
                         "ONE launch per step (dn_mlp_step_sampled), hipGraph, bf16-grade networks": round(res["one_launch_graph"], 1),
                         "ONE launch per step (dn_mlp_step_sampled), hipGraph, fp32-grade networks": round(res["one_launch_graph_fp32"], 1)},
            "n_steps": n_steps, "num_envs": n,
            "what": "policy-in-the-loop rollout: MLP 13-512-512-256 (pi, vf; Tanh) + Gaussian sample + dn_step + "
                    "V(terminal_obs) bootstrap per step, dn_gae per rollout, per-drone obs normaliser on"}


def sac_collect(pkg, track, n, max_steps, dev, rank):
    """BASELINE configs[4] per GPU as the reference's SAC agent collects (PBDroneSimulator.py:297-338): actor network
    13-256-256 -> (mu, log_std) in the loop, squashed-Gaussian sample, dn_step with action / observation noise, every
    transition into a device-resident replay buffer.  Reported beside the headline, never as `value`."""
    import torch
    from drl_dronenavigation_amd.collector import OffPolicyCollector
    torch.manual_seed(7 + rank)
    actor = pkg.SacActor().to(dev)
    res = {}
    steps = 64
    for label, grade, graph in (("torch fp32 actor, eager", None, False), ("torch fp32 actor, hipGraph", None, True),
                                ("fused MFMA actor bf16 grade, two launches per step, hipGraph", "bf16", True),
                                ("fused MFMA actor fp16 grade, two launches per step, hipGraph", "fp16", True),
                                ("fused MFMA actor fp32 grade, two launches per step, eager", "fp32", False),
                                ("fused MFMA actor fp32 grade, two launches per step, hipGraph", "fp32", True)):
        env = pkg.DroneVecEnv(track, n, max_steps=max_steps, normalize_obs=True, act_noise_sigma=0.002, obs_noise_sigma=0.01,
                              seed=1, env_id_offset=rank * n, device=dev)
        pol = actor if grade is None else pkg.FusedSacActor(actor, n, dev, grade=grade)
        col = OffPolicyCollector(env, pol, buffer_size=steps)
        run = col.collect_cycle if graph else (lambda: col.collect(steps))      # noqa: B023
        for _ in range(3):
            run()
        torch.cuda.synchronize(dev)
        reps, best = 10, 0.0                                # best of three ~20-60 ms windows
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                run()
            torch.cuda.synchronize(dev)
            best = max(best, n * steps * reps / (time.perf_counter() - t0))
        res[label] = round(best, 1)
        env.close()
    obs = torch.rand(n, 13, device=dev)
    kern, kernels = {}, {}
    for grade, kname, passes in (("bf16", "dn_mlp_sac_lds_kernel<false, false>", 1), ("fp16", "dn_mlp_sac_lds_kernel<false, true>", 1),
                                 ("fp32", "dn_mlp_sac_lds_kernel<true, false>", 3)):
        fa = pkg.FusedSacActor(actor, n, dev, grade=grade)
        kernels["actor_" + grade] = mlp_kernel_leg(torch, dev, kname, lambda fa=fa: fa.mean_log_std(obs), n, SAC_MACS_MFMA, SAC_MACS, 1, passes)
        kern[grade] = kernels["actor_" + grade]["avg_us"]
    try:
        env = pkg.DroneVecEnv(track, n, max_steps=max_steps, normalize_obs=True, act_noise_sigma=0.002, obs_noise_sigma=0.01,
                              seed=1, env_id_offset=rank * n, device=dev)
        env.reset_tensor()
        lib, h = pkg._capi.load(), env._handle
        mls = torch.zeros((n, 8), device=dev)
        b = dict(act=torch.zeros((n, 4), device=dev), obs=torch.zeros((n, 13), device=dev), rew=torch.zeros(n, device=dev),
                 done=torch.zeros(n, dtype=torch.uint8, device=dev), trunc=torch.zeros(n, dtype=torch.uint8, device=dev),
                 found=torch.zeros(n, dtype=torch.int32, device=dev), term=torch.zeros((n, 13), device=dev))

        def step_squashed():
            pkg._capi.check(lib.dn_step_squashed(h, mls.data_ptr(), 1, 0, b["act"].data_ptr(), None, b["obs"].data_ptr(), b["rew"].data_ptr(),
                                                 b["done"].data_ptr(), b["trunc"].data_ptr(), b["found"].data_ptr(), b["term"].data_ptr(),
                                                 None, None, None, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        kernels["step_squashed"] = {"kernel": "dn_step_pqx_kernel<double, true, true, true>" if env.kernel_waves(fused=False) == 3
                                    else "dn_step_many_1w_kernel<double, true, true, true, false, true>",
                                    "avg_us": round(time_launches(torch, dev, step_squashed), 2),
                                    "what": "clamp + Philox draw + tanh + the control step with action / observation noise and the obs normaliser"}
        env.close()
    except Exception as exc:  # noqa: BLE001
        kernels["error"] = f"{type(exc).__name__}: {exc}"
    return {"value": res["fused MFMA actor fp32 grade, two launches per step, hipGraph"], "unit": "env-steps/s", "variants": res, "num_envs": n, "steps": steps,
            "actor_forward_us_python_loop": kern, "kernels": kernels,
            "what": "SAC collection loop of config 5 on one shard: dn_mlp_forward (actor 13-256-256 ReLU -> mu | log_std) and dn_step_squashed "
                    "(clamp, Philox draw, tanh inside the step kernel; Philox action + observation noise, per-drone obs normaliser), every output "
                    "written in place into the replay ring; `value` = float32-grade actor, the whole ring-buffer cycle replayed from a hipGraph"}


def ppo_rollout_sharded(pkg, track, n, max_steps, dev, rank, world, dist):
    """BASELINE configs[3] (131072 drones over 4 GPUs): FusedRolloutCollector per rank on its shard + one all-gather of
    the packed advantages/returns per rollout over RCCL; aggregate env-steps/s with the collective inside the timed
    region (MAX over ranks)."""
    import torch
    from drl_dronenavigation_amd.collector import FusedRolloutCollector
    n_steps = 32
    torch.manual_seed(1)                                   # same weights on every rank
    net = pkg.MlpActorCritic().to(dev)
    fused = pkg.FusedMlpPolicy(net, n, dev)
    env = pkg.DroneVecEnv(track, n, max_steps=max_steps, normalize_obs=True, env_id_offset=rank * n, device=dev)
    col = FusedRolloutCollector(env, fused, n_steps, gather=True, seed=1)
    for _ in range(3):
        out = col.collect()
    torch.cuda.synchronize(dev)
    dist.barrier(device_ids=[dev.index])
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        out = col.collect()
    torch.cuda.synchronize(dev)
    tw = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(tw, op=dist.ReduceOp.MAX)
    assert tuple(out["advantages_global"].shape) == (n_steps, world, n)      # strided views of the static receive buffer
    env.close()
    return {"value": round(n * world * n_steps * reps / float(tw[0]), 1), "unit": "env-steps/s", "n_steps": n_steps,
            "global_num_envs": n * world, "all_gather_bytes_sent_per_rank_per_rollout": 2 * n_steps * n * 4,
            "rccl_world_size": dist.get_world_size(),
            "what": "FusedRolloutCollector(gather=True) on every rank: policy in the loop + one RCCL all-gather of the packed "
                    "advantages/returns per rollout"}


def main():
    args = parse()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("DN_BENCH_SPIN", "1") == "1":
        # The contract's timed region ends in a host-side synchronize; with the driver's 20 steps it is one 40 us launch, and the
        # default blocking wait adds a thread wake-up of the same order.  Ask HIP to spin in its waits instead (a host-side wait
        # policy, hipDeviceScheduleSpin; it must be set before the device is initialised).
        try:
            hip = C.CDLL("libamdhip64.so")
            hip.hipSetDevice(C.c_int(local_rank))              # the flag belongs to the current device: this rank's own
            hip.hipSetDeviceFlags(C.c_uint(1))
        except OSError:
            pass
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1 or args.force_dist:
        # the GPU boxes export NCCL_DEBUG=VERSION: RCCL then prints its banner / warnings on STDOUT from every rank (C stdio,
        # flushed at exit, i.e. after the JSON line).  The bench's stdout is one JSON line: RCCL's log goes to stderr.
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # one line per rank on stderr BEFORE the first collective: a mis-launched scaling run (N ranks on one device, a rank that sees
        # fewer devices than WORLD_SIZE) is then diagnosable from the driver's tail even if the rendezvous hangs
        print(f"bench.py: rank {rank}/{world} local_rank {local_rank} -> cuda:{local_rank} of {torch.cuda.device_count()} visible device(s), "
              f"rccl_world_size {world}, MASTER {os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}", file=sys.stderr, flush=True)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import drl_dronenavigation_amd as pkg
    from drl_dronenavigation_amd import tracks
    track = tracks.REGISTRY[args.track]()
    n, K, W = args.num_envs, args.steps, args.warmup
    max_steps = 4096                                      # --max_env_steps default, parameter_manager.py:25
    env = pkg.DroneVecEnv(track, n, max_steps=max_steps, normalize_obs=args.normalize_obs,
                          compute_dtype=args.compute_dtype, env_id_offset=rank * n, device=dev)
    env.reset_tensor()

    # synthetic actions resident in HBM before the timed region: U(-1,1)^4 float32, seeded per global drone id
    A = max(1, min(args.action_batches, K))
    g = torch.Generator(device="cpu").manual_seed(1 + rank)

    def draw_actions(kind, gen):
        if kind == "hover":                                # hover is a = 0.092227
            return (0.0922 + 0.003 * torch.randn((A, n, 4), generator=gen, dtype=torch.float32)).to(dev)
        return (torch.rand((A, n, 4), generator=gen, dtype=torch.float32) * 2 - 1).to(dev)
    acts = draw_actions(args.actions, g)
    lib = pkg._capi.load()
    stream = torch.cuda.current_stream(dev)
    sptr = C.c_void_p(stream.cuda_stream)
    h = env._handle
    o = dict(obs=torch.empty((A, n, 13), dtype=torch.float32, device=dev), reward=torch.empty((A, n), device=dev),
             done=torch.empty((A, n), dtype=torch.uint8, device=dev), trunc=torch.empty((A, n), dtype=torch.uint8, device=dev),
             found=torch.empty((A, n), dtype=torch.int32, device=dev))

    step_many = lib.dn_step_many                          # addresses taken once: the driver's timed region is ONE launch, every
    many_tail = (acts.data_ptr(), o["obs"].data_ptr(), o["reward"].data_ptr(), o["done"].data_ptr(), o["trunc"].data_ptr(),
                 o["found"].data_ptr(), None, None, None, None, sptr)           # microsecond of Python in front of it counts

    def run_many(k):
        """k steps as ceil(k/A) dn_step_many calls over the A resident action batches."""
        done = 0
        while done < k:
            c = min(A, k - done)
            rc = step_many(h, c, *many_tail)
            if rc:
                pkg._capi.check(rc)
            done += c

    ptrs = [(acts[j].data_ptr(), o["obs"][j].data_ptr(), o["reward"][j].data_ptr(), o["done"][j].data_ptr(),
             o["trunc"][j].data_ptr(), o["found"][j].data_ptr()) for j in range(A)]

    def run_single(k):
        for t in range(k):
            p = ptrs[t % A]
            rc = lib.dn_step(h, p[0], p[1], p[2], p[3], p[4], p[5], None, None, None, None, sptr)
            if rc:
                pkg._capi.check(rc)

    def build_graph():
        """hipGraph of A single-step launches (dn_step x A), captured once: what a policy-in-the-loop step costs on
        the GPU timeline when the host's per-launch overhead (ctypes + hipLaunchKernel, ~7 us) is taken out."""
        side = torch.cuda.Stream(dev)
        side.wait_stream(stream)
        with torch.cuda.stream(side):                      # warm the kernels on the capture stream
            sp = C.c_void_p(side.cuda_stream)
            for p_ in ptrs[:2]:
                pkg._capi.check(lib.dn_step(h, p_[0], p_[1], p_[2], p_[3], p_[4], p_[5], None, None, None, None, sp))
        stream.wait_stream(side)
        torch.cuda.synchronize(dev)
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_, stream=side):
            sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            for p_ in ptrs:
                pkg._capi.check(lib.dn_step(h, p_[0], p_[1], p_[2], p_[3], p_[4], p_[5], None, None, None, None, sp))
        return g_

    graph = None

    def run_graph(k):
        nonlocal graph
        if graph is None:
            graph = build_graph()
        for _ in range((k + A - 1) // A):
            graph.replay()

    if args.mode == "graph":
        if K % A or W % A:
            raise SystemExit(f"--mode graph needs --steps and --warmup to be multiples of {A}")
        run_graph(A)
    run = {"many": run_many, "single": run_single, "graph": run_graph}[args.mode]

    def barrier():
        if dist is not None:
            dist.barrier(device_ids=[local_rank])

    def timed(fn, k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0.record(stream)
        fn(k)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) * 1e3 / k           # us per step on the GPU timeline

    # untimed pre-roll, independent of --warmup: the clock of an idle GPU takes ~50 ms of load to settle, and the
    # driver's default (--steps 20 --warmup 5) is a 40 us timed region
    t_pre = time.perf_counter()
    pre_steps = 0
    while time.perf_counter() - t_pre < (0.0 if args.profile_lite else PREROLL_SECONDS):
        run_many(64 * A)
        torch.cuda.synchronize(dev)
        pre_steps += 64 * A
    run(W)
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)                                     # a torch event creates its hipEvent on first use (~30 us, measured):
    e1.record(stream)                                     # not inside a timed region that is one 31 us launch
    torch.cuda.synchronize(dev)
    # The roofline's HIP events bracket the timed region on the launch stream (two marker packets; measured: they cost the wall clock
    # nothing, 13.05-13.43 against 13.15-13.53 G env-steps/s without them).  Attaching them to the kernel's own dispatch INSIDE the region
    # (dn_set_launch_events -> hipExtLaunchKernelGGL) was tried and costs the host 6-15 us of launch path (8.2-10.7 G): that form times one
    # more launch after the region instead (`kernel_only` below).
    e0.record(stream)                                     # the stream is idle: it completes at once
    t0 = time.perf_counter()
    ta = t0
    run(K)
    tb = time.perf_counter()
    e1.record(stream)
    tc = time.perf_counter()
    torch.cuda.synchronize(dev)
    td = time.perf_counter()
    wall = td - t0                                        # this rank's K steps, started together (barrier above); MAX over ranks below
    barrier()                                             # the closing bracket: a rank's clock stops when ITS work is done, not after
    torch.cuda.synchronize(dev)                           # an RCCL barrier (tens of us, the size of the driver's whole 20-step region)
    if os.environ.get("DN_BENCH_DEBUG"):
        print(f"timed region: rec0 {(ta - t0) * 1e6:.1f} run {(tb - ta) * 1e6:.1f} rec1 {(tc - tb) * 1e6:.1f} sync {(td - tc) * 1e6:.1f} us", file=sys.stderr)
    gpu_ms = e0.elapsed_time(e1)                          # HIP events on the launch stream, timed region only
    def kernel_only(handle, fn, k):
        """The same k-step launch with its own dispatch's begin / end stamps (dn_set_launch_events -> hipExtLaunchKernelGGL): the kernel's
        duration as rocprofv3 --kernel-trace reports it (marker events also see the ~1 us between a marker and the kernel).  Median of 5."""
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0.record(stream); k1.record(stream)
        torch.cuda.synchronize(dev)
        reps_ = []
        for _ in range(5):
            pkg._capi.check(lib.dn_set_launch_events(handle, C.c_void_p(k0.cuda_event), C.c_void_p(k1.cuda_event)))
            fn(k)
            torch.cuda.synchronize(dev)
            reps_.append(k0.elapsed_time(k1) * 1e3)
        return sorted(reps_)[len(reps_) // 2]

    kernel_only_us = None
    if args.mode == "many" and K <= A and not args.profile_lite:      # (not in the counter passes: these launches take another dispatch path)
        kernel_only_us = kernel_only(h, run, K)                       # OUTSIDE the region the wall clock brackets
    if dist is not None:
        tw = torch.tensor([wall, gpu_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall, gpu_ms = float(tw[0]), float(tw[1])
    st = env.stats()
    waves = env.kernel_waves(fused=args.mode == "many")
    waves_single = env.kernel_waves(fused=False)

    def leg(label, fn, steps, steps_per_launch, norm, wv):
        """One launch shape timed on its own (HIP events over >= 25 ms of launches), with both byte accountings."""
        fn(max(A, 64))
        us = timed(fn, steps)
        launch_us = us * max(1, steps_per_launch)
        algo = algo_bytes_per_launch(n, max(1, steps_per_launch), norm)
        traffic, src = traffic_per_launch(args.track, n, args.compute_dtype, norm, steps_per_launch if steps_per_launch > 1 else 0, wv)
        d = {"kernel": kernel_name(wv, args.compute_dtype, norm, steps_per_launch > 1), "us_per_vector_step": round(us, 4),
             "avg_launch_us": round(launch_us, 4), "vector_steps_per_launch": max(1, steps_per_launch),
             "value": round(n * world / (us * 1e-6), 1), "unit": "env-steps/s", "timed_vector_steps": steps,
             "algorithmic_bytes_per_launch": algo,
             "frac": round(algo / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5),
             "traffic": traffic, "traffic_source": src,
             "traffic_frac": (round(traffic / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5) if traffic else None)}
        return d

    # the other launch shapes, outside the timed region: "single" = one dn_step launch per step from Python (what a
    # policy-in-the-loop VecEnv.step_tensor() costs incl. host launch gaps), "graph" = the same launches replayed from a
    # hipGraph (GPU timeline only), "many" = the fused K-step kernel
    k_single = (8192 if not args.profile_lite else 2 * A) // A * A if A <= 8192 else A
    k_many = max(A, (32768 if not args.profile_lite else 4 * A) // A * A)
    others = {}
    for m_ in ("many", "single", "graph"):
        if world > 1 and m_ == "graph":
            continue
        fn = {"many": run_many, "single": run_single, "graph": run_graph}[m_]
        spl = A if m_ == "many" else 1
        wv = env.kernel_waves(fused=m_ == "many")
        d = leg(m_, fn, k_single if m_ != "many" else k_many, spl, args.normalize_obs, wv)
        others[m_] = d
    single_step = dict(others["graph"] if "graph" in others else others["single"])
    single_step["launched_from"] = "hipGraph replay of dn_step launches" if "graph" in others else "python loop of dn_step calls"
    single_step["python_loop_us_per_vector_step"] = others["single"]["us_per_vector_step"]

    # the same workload with the normaliser switched the OTHER way (the headline has it on, as the reference does,
    # PBDroneSimulator.py:181; the sub-leg shows what the bare step costs)
    other_norm = not args.normalize_obs
    norm_leg = None
    if world == 1:
        try:
            env_n = pkg.DroneVecEnv(track, n, max_steps=max_steps, normalize_obs=other_norm, compute_dtype=args.compute_dtype,
                                    env_id_offset=rank * n, device=dev)
            env_n.reset_tensor()
            hn = env_n._handle

            def many_n(k):
                done = 0
                while done < k:
                    c = min(A, k - done)
                    pkg._capi.check(lib.dn_step_many(hn, c, acts.data_ptr(), o["obs"].data_ptr(), o["reward"].data_ptr(),
                                                     o["done"].data_ptr(), o["trunc"].data_ptr(), o["found"].data_ptr(),
                                                     None, None, None, None, sptr))
                    done += c

            def single_n(k):
                for t in range(k):
                    p = ptrs[t % A]
                    rc = lib.dn_step(hn, p[0], p[1], p[2], p[3], p[4], p[5], None, None, None, None, sptr)
                    if rc:
                        pkg._capi.check(rc)

            norm_leg = {"fused": leg("many", many_n, k_many, A, other_norm, env_n.kernel_waves(fused=True)),
                        "single_step": leg("single", single_n, k_single, 1, other_norm, env_n.kernel_waves(fused=False)),
                        "what": ("same workload WITHOUT the per-drone NormalizeObservation (the bare step; the reference never runs it "
                                 "this way)" if not other_norm else
                                 "same workload with the per-drone NormalizeObservation fused in (the reference always wraps it, "
                                 "PBDroneSimulator.py:181): +432 B of statistics per drone and launch")}
            env_n.close()
        except Exception as exc:  # noqa: BLE001
            norm_leg = {"error": f"{type(exc).__name__}: {exc}"}

    # SURVEY 8(d) C2's second action distribution, the same launch: a = 0.0922 + 0.003 N(0,1), the un-saturated band around hover.  The
    # headline's U(-1,1) saturates 99.6 % of actions, which rotor_force_sat turns into two selects per rotor on 79 % of (tile, rotor)
    # evaluations; here every wave evaluates the float32 chain and episodes are long (no reset pass on most tile-steps) -- the regime a
    # trained policy lives in, timed so that the fast path cannot be read as tuning to the benchmark.
    hover_leg = None
    if world == 1 and not args.profile_lite and args.mode == "many":
        try:
            other_kind = "hover" if args.actions == "uniform" else "uniform"
            env_h = pkg.DroneVecEnv(track, n, max_steps=max_steps, normalize_obs=args.normalize_obs, compute_dtype=args.compute_dtype,
                                    env_id_offset=rank * n, device=dev)
            env_h.reset_tensor()
            hh = env_h._handle
            acts_h = draw_actions(other_kind, torch.Generator(device="cpu").manual_seed(101 + rank))

            def many_h(k):
                done = 0
                while done < k:
                    c = min(A, k - done)
                    pkg._capi.check(lib.dn_step_many(hh, c, acts_h.data_ptr(), o["obs"].data_ptr(), o["reward"].data_ptr(),
                                                     o["done"].data_ptr(), o["trunc"].data_ptr(), o["found"].data_ptr(),
                                                     None, None, None, None, sptr))
                    done += c
            t_h = time.perf_counter()
            while time.perf_counter() - t_h < 0.25:            # into the regime's steady state (episodes of up to max_steps steps)
                many_h(64 * A)
                torch.cuda.synchronize(dev)
            wv_h = env_h.kernel_waves(fused=True)
            hover_leg = leg("many", many_h, k_many, A, args.normalize_obs, wv_h)
            ko_h = kernel_only(hh, many_h, A)
            st_h = env_h.stats()
            hover_leg.update({
                "actions": ("0.0922 + 0.003 N(0,1) (hover band, SURVEY 8(d) C2)" if other_kind == "hover" else "U(-1,1)^4"),
                "fast_path_hit_rate": round(fast_path_hit_rate(torch, acts_h), 4),
                "episodes_finished": st_h["episodes"], "env_steps": st_h.get("env_steps"),
                "kernel_only_launch_us": round(ko_h, 3), "kernel_only_us_per_vector_step": round(ko_h / A, 4),
                "kernel_only_frac": round(algo_bytes_per_launch(n, A, args.normalize_obs) / (ko_h * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5),
                "valu": valu_bound(kernel_name(wv_h, args.compute_dtype, args.normalize_obs, True), n, A, ko_h / A, env.num_cus,
                                   regime=("hover_band_k20" if other_kind == "hover" else None)),
                "what": "the headline's launch (same kernel, same K, same fleet size, normaliser as in the headline) under the OTHER action "
                        "distribution of SURVEY 8(d) C2; us_per_vector_step = HIP events around back-to-back launches (dispatch gaps included), "
                        "kernel_only_* = the kernel's own duration (dn_set_launch_events, median of 5)"})
            env_h.close()
            del acts_h
        except Exception as exc:  # noqa: BLE001
            hover_leg = {"error": f"{type(exc).__name__}: {exc}"}

    # a measured stream-copy ceiling of THIS box beside the nominal 8 TB/s (a device-to-device copy of 1 GiB: read + write)
    copy_ceiling = None
    if world == 1 and not args.profile_lite:
        try:
            src = torch.empty(1 << 28, dtype=torch.float32, device=dev).fill_(1.0)
            dst = torch.empty_like(src)

            def time_copy(fn):
                for _ in range(3):
                    fn()
                e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(dev)
                e0_.record(stream)
                for _ in range(20):
                    fn()
                e1_.record(stream)
                torch.cuda.synchronize(dev)
                return 2.0 * src.numel() * 4 * 20 / (e0_.elapsed_time(e1_) * 1e-3) / 1e9
            gbps = time_copy(lambda: pkg.stream_copy(dst, src))
            gbps_torch = time_copy(lambda: dst.copy_(src))
            copy_ceiling = {"GBps": round(gbps, 1), "frac_of_nominal": round(gbps / HBM_PEAK_GBPS, 4),
                            "what": "dn_stream_copy: hand-written float4 copy kernel (one 16-byte load + store per lane, one lane per 16 bytes: the fastest of "
                                    "the forms swept in profiles/r04_copy_sweep.txt) over 1 GiB float32 -- 1 GiB read + 1 GiB written per copy, 20 copies "
                                    "between two HIP events",
                            "torch_copy_GBps": round(gbps_torch, 1),
                            "guide_float4_copy_GBps": 6290,
                            "guide_source": "/opt/skills/guides/MI355X_MICROARCH.md (float4 copy, 6.29 TB/s)"}
            del src, dst
        except Exception as exc:  # noqa: BLE001
            copy_ceiling = {"error": f"{type(exc).__name__}: {exc}"}

    # where HBM IS the bound: the same step kernel over a fleet that fills the chip many times over (one wave per 64 drones,
    # 32768 workgroups), one control step per launch -- the regime the byte model of SURVEY 8(d) describes
    def hbm_bound_leg(norm_l):
        nl = 2097152
        torch.cuda.empty_cache()                              # 1.7 GB of buffers for this leg: from a compact allocator state, not from the holes the earlier legs left
        env_l = pkg.DroneVecEnv(track, nl, max_steps=max_steps, normalize_obs=norm_l, compute_dtype=args.compute_dtype, device=dev)
        env_l.reset_tensor()
        gl = torch.Generator(device="cpu").manual_seed(7)
        acts_l = (torch.rand((2, nl, 4), generator=gl, dtype=torch.float32) * 2 - 1).to(dev)
        for t in range(20):
            env_l.step_tensor(acts_l[t & 1])
        e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0_.record(stream)
        reps_l = 100
        for t in range(reps_l):
            env_l.step_tensor(acts_l[t & 1])
        e1_.record(stream)
        torch.cuda.synchronize(dev)
        us_python = e0_.elapsed_time(e1_) * 1e3 / reps_l      # eager Python loop: the host's launch gaps are inside (side key only)
        flip = [0]

        def one_step():
            flip[0] ^= 1
            env_l.step_tensor(acts_l[flip[0]])
        us_l = time_launches(torch, dev, one_step, reps=100, warm=4)   # the DEVICE timeline: 100 dn_step launches replayed from one hipGraph
        bytes_l = (ALGO_BYTES_PER_ENV_STEP + (ALGO_BYTES_NORMALISER if norm_l else 0)) * nl
        wv_l = env_l.kernel_waves(fused=False)
        traffic_l, src_l = traffic_per_launch(args.track, nl, args.compute_dtype, norm_l, 0, wv_l)
        d = {"num_envs": nl, "kernel": kernel_name(wv_l, args.compute_dtype, norm_l, False),
             "us_per_vector_step": round(us_l, 3), "value": round(nl / (us_l * 1e-6), 1), "unit": "env-steps/s",
             "algorithmic_bytes_per_launch": bytes_l, "achieved_GBps": round(bytes_l / (us_l * 1e-6) / 1e9, 1),
             "frac": round(bytes_l / (us_l * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5),
             "frac_of_copy_ceiling": (round(bytes_l / (us_l * 1e-6) / 1e9 / copy_ceiling["GBps"], 4)
                                      if copy_ceiling and "GBps" in copy_ceiling else None),
             "timed_with": "100 dn_step launches captured once into a hipGraph and replayed between two HIP events (device timeline; best of 3)",
             "python_loop_us_per_vector_step": round(us_python, 3),
             "traffic": traffic_l, "traffic_source": src_l,
             "traffic_ratio": (round(traffic_l / bytes_l, 4) if traffic_l else None),
             "what": ("2 097 152 drones, one control step per launch, per-drone NormalizeObservation ON as the reference runs it "
                      "(PBDroneSimulator.py:181): 288 + 432 = 720 B per drone-step, the regime in which the step IS bandwidth bound"
                      if norm_l else
                      "2 097 152 drones, one control step per launch, normaliser off (288 B per drone-step; the reference never runs it this way)")}
        env_l.close()
        del acts_l
        return d

    large = large_norm = None
    if world == 1 and not args.profile_lite:
        for norm_l in (True, False):
            try:
                d_l = hbm_bound_leg(norm_l)
            except Exception as exc:  # noqa: BLE001
                d_l = {"error": f"{type(exc).__name__}: {exc}"}
            if norm_l:
                large_norm = d_l
            else:
                large = d_l

    # the fused launch on larger single-GPU fleets (BASELINE configs[3] / [4] hold 131 072 / 262 144 drones in all: on one GPU they are one fleet,
    # sharded they are the headline's 32 768 per GPU): dn_create's own shape pick per size, 64-step launches replayed from one hipGraph
    fleet_sweep = None
    if world == 1 and not args.profile_lite and args.mode == "many":
        fleet_sweep = {"what": "us per vector step of the fused launch (64 steps per launch, 20 launches replayed from one hipGraph, best of 3) at larger "
                               "fleets on ONE GPU, the shape dn_create picks at each size; workload as in the headline", "sizes": {}}
        for n_f in (65536, 131072, 262144):
            try:
                torch.cuda.empty_cache()
                env_f = pkg.DroneVecEnv(track, n_f, max_steps=max_steps, normalize_obs=args.normalize_obs, compute_dtype=args.compute_dtype, device=dev)
                env_f.reset_tensor()
                gf = torch.Generator(device="cpu").manual_seed(11)
                acts_f = (torch.rand((64, n_f, 4), generator=gf, dtype=torch.float32) * 2 - 1).to(dev)
                out_f = env_f.rollout_tensor(acts_f)
                for _ in range(4):                                # into a mixed fleet state (episodes ending in every launch)
                    env_f.rollout_tensor(acts_f, out=out_f)
                us_f = time_launches(torch, dev, lambda: env_f.rollout_tensor(acts_f, out=out_f), reps=20, warm=2) / 64.0
                wv_f = env_f.kernel_waves(fused=True)
                fleet_sweep["sizes"][str(n_f)] = {"kernel": kernel_name(wv_f, args.compute_dtype, args.normalize_obs, True), "waves_per_64_drones": wv_f,
                                                  "us_per_vector_step": round(us_f, 4), "value": round(n_f / (us_f * 1e-6), 1), "unit": "env-steps/s"}
                env_f.close()
                del acts_f, out_f
            except Exception as exc:  # noqa: BLE001
                fleet_sweep["sizes"][str(n_f)] = {"error": f"{type(exc).__name__}: {exc}"}

    # the SB3 NumPy surface (PCIe-inclusive: H2D actions, D2H obs/reward/done/found, N info dicts built in Python);
    # host bound, reported for the record only
    if world == 1 and not args.profile_lite:
        import numpy as np
        a_np = acts[0].cpu().numpy()
        for mode_, fresh_ in (("full", True), ("sparse", True), ("sparse", False)):
            try:
                env_s = pkg.DroneVecEnv(track, n, max_steps=max_steps, normalize_obs=args.normalize_obs, compute_dtype=args.compute_dtype,
                                        env_id_offset=rank * n, device=dev, info_mode=mode_, fresh_arrays=fresh_)
                env_s.reset()
                for _ in range(40):                              # into the steady state: episodes ending every step
                    env_s.step(a_np)
                best_ = 1e30
                for _ in range(3):
                    t0_ = time.perf_counter()
                    reps_ = 20 if mode_ == "sparse" else 5
                    for _ in range(reps_):
                        env_s.step(a_np)
                    best_ = min(best_, (time.perf_counter() - t0_) * 1e6 / reps_)
                env_s.close()
                others["sb3_numpy_step_infos_" + mode_ + ("" if fresh_ else "_reused_host_buffers")] = {
                    "us_per_vector_step": round(best_, 1), "value": round(n / (best_ * 1e-6), 1)}
            except Exception as exc:  # noqa: BLE001
                others["sb3_numpy_step_infos_" + mode_ + ("" if fresh_ else "_reused_host_buffers")] = {"error": f"{type(exc).__name__}: {exc}"}

    line = None
    if rank == 0:
        value = n * world * K / wall
        step_us = gpu_ms * 1e3 / K
        steps_per_launch = A if args.mode == "many" else 1
        launches = (K + steps_per_launch - 1) // steps_per_launch
        launch_us = gpu_ms * 1e3 / launches
        algo_launch = algo_bytes_per_launch(n, steps_per_launch, args.normalize_obs)
        achieved = algo_launch / (launch_us * 1e-6) / 1e9
        wv = waves if args.mode == "many" else waves_single
        traffic, tsrc = traffic_per_launch(args.track, n, args.compute_dtype, args.normalize_obs,
                                           steps_per_launch if args.mode == "many" else 0, wv)
        kname = kernel_name(wv, args.compute_dtype, args.normalize_obs, args.mode == "many")
        valu = (valu_bound(kname, n, steps_per_launch, (kernel_only_us / K) if kernel_only_us else step_us, env.num_cus,
                           regime=("hover_band_k20" if args.actions == "hover" else None)) if args.mode == "many" else None)
        ln_ok = isinstance(large_norm, dict) and "frac" in large_norm
        l_ok = isinstance(large, dict) and "frac" in large
        hv_ok = isinstance(hover_leg, dict) and "kernel_only_us_per_vector_step" in hover_leg
        # Every fraction DESIGN.md section 5 quotes, as SCALAR fields of the roofline object (nested objects do not survive every
        # reader of this line): the bound that governs the headline, the kernel's own duration, the other launch shape, the other action
        # regime, and the regime in which HBM IS the bound -- each with the place its inputs come from.
        flat = {
            "valu_frac": valu and valu["valu_frac"], "valu_floor_us_per_step": valu and valu["valu_floor_us_per_step"],
            "valu_insts_per_tile_step": valu and valu["valu_insts_per_tile_step"], "valu_cycles_per_inst": valu and valu["valu_cycles_per_inst"],
            "valu_shader_clock_ghz": valu and valu["shader_clock_ghz"],
            "valu_source": valu and f"{valu['counters_from']}; clock: {valu['shader_clock_source']}",
            "governing_bound": ("valu" if valu and valu["valu_frac"] > achieved / HBM_PEAK_GBPS else "hbm"),
            "kernel_only_us": kernel_only_us and round(kernel_only_us, 3),
            "kernel_only_frac": kernel_only_us and round(algo_launch / (kernel_only_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5),
            "kernel_only_source": "dn_set_launch_events on 5 launches right after the timed region (median); rocprofv3 twin: profiles/r06_a_bench32k_k20.txt",
            "single_step_us": single_step.get("us_per_vector_step"), "single_step_frac": single_step.get("frac"),
            "single_step_source": "single_step leg of this line (dn_step launches replayed from a hipGraph, 288 + 432 B per drone)",
            "fast_path_hit_rate": round(fast_path_hit_rate(torch, acts), 4),
            "hover_us_per_step": hover_leg["kernel_only_us_per_vector_step"] if hv_ok else None,
            "hover_frac": hover_leg["kernel_only_frac"] if hv_ok else None,
            "hover_fast_path_hit_rate": hover_leg["fast_path_hit_rate"] if hv_ok else None,
            "hover_valu_insts_per_tile_step": (hover_leg["valu"] or {}).get("valu_insts_per_tile_step") if hv_ok else None,
            "hover_valu_frac": (hover_leg["valu"] or {}).get("valu_frac") if hv_ok else None,
            "hover_source": "hover_band leg of this line (same launch, actions 0.0922 + 0.003 N(0,1)); counters: profiles/instmix.json hover_band_k20",
            "hbm_bound_num_envs": 2097152 if ln_ok else None,
            "hbm_bound_us": large_norm["us_per_vector_step"] if ln_ok else None,
            "hbm_bound_frac": large_norm["frac"] if ln_ok else None,
            "hbm_bound_frac_of_copy": large_norm["frac_of_copy_ceiling"] if ln_ok else None,
            "hbm_bound_traffic_ratio": large_norm.get("traffic_ratio") if ln_ok else None,
            "hbm_bound_nonorm_us": large["us_per_vector_step"] if l_ok else None,
            "hbm_bound_nonorm_frac": large["frac"] if l_ok else None,
            "hbm_bound_nonorm_frac_of_copy": large["frac_of_copy_ceiling"] if l_ok else None,
            "hbm_copy_GBps": (copy_ceiling or {}).get("GBps"),
            "hbm_bound_source": "hbm_bound_fleet_norm / hbm_bound_fleet / hbm_copy_ceiling legs of this line (2 097 152 drones, one dn_step per launch, "
                                "100 launches replayed from one hipGraph; copy kernel measured in the same process)",
        }
        line = {
            "metric": "env_steps_per_sec", "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": round(wall * 1e3 / K, 6), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if args.compute_dtype == "float64" else "f32", "data": "synthetic",
            "config": {"workload": f"{n} drones/GPU, 8-gate race track (Waypoints.reaching), "
                                   f"{'U(-1,1)^4' if args.actions == 'uniform' else '0.0922 + 0.003 N(0,1) hover-band'} actions resident in HBM, "
                                   f"norm_rew off, obs normaliser {'on' if args.normalize_obs else 'off'}, auto-reset on",
                       "num_envs_per_gpu": n, "global_num_envs": n * world, "track": args.track,
                       "state_dtype": "f32", "launch_mode": args.mode, "parallelism": f"env-shard x{world} (no data-path collective)",
                       "episodes_finished_rank0": st["episodes"],
                       "rccl_world_size": (dist.get_world_size() if dist is not None else 1),
                       "ground_contact": ("on" if env.ground_contact else
                                          "resolved off by DN_GROUND_CONTACT_AUTO: on this track every point low enough to touch the floor is "
                                          "already outside the corridor of every segment, so the contact term of PBDroneEnv.py:699 cannot fire"),
                       "preroll": {"seconds": PREROLL_SECONDS, "vector_steps": pre_steps,
                                   "what": "untimed fused stepping before --warmup so that the timed region runs at a settled clock"}},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": tsrc,
                         "traffic_frac": (round(traffic / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5) if traffic else None),
                         **flat,
                         "kernel": kname,
                         "waves_per_64_drones": wv,
                         "env_steps_per_launch": n * steps_per_launch, "vector_steps_per_launch": steps_per_launch,
                         "algorithmic_bytes_per_launch": algo_launch,
                         "algorithmic_bytes_model": (f"{ALGO_BYTES_IO_PER_STEP} B x {steps_per_launch} steps + {ALGO_BYTES_STATE} B state"
                                                     + (f" + {ALGO_BYTES_NORMALISER} B statistics" if args.normalize_obs else "")
                                                     + " per drone and launch (SURVEY 8(d) split into per-step I/O and per-launch state)"),
                         "avg_launch_us": round(launch_us, 4), "us_per_vector_step": round(step_us, 4),
                         "timed_with": "HIP events recorded on the launch stream around the timed region (marker to kernel gaps included)",
                         "kernel_only": ({"launch_us": round(kernel_only_us, 3),
                                          "frac": round(algo_launch / (kernel_only_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5),
                                          "what": "median of 5 identical launches right after the timed region, each timed by HIP events attached to "
                                                  "its own dispatch (dn_set_launch_events -> hipExtLaunchKernelGGL): the kernel's duration as "
                                                  "rocprofv3 --kernel-trace reports it"} if kernel_only_us else None),
                         "note": "at 32768 drones neither launch shape is bandwidth bound: with two tiles per CU the fused launch is bound by the vector ALUs "
                                 "(`valu_*`: vector instructions per tile-step x ALU cycles each, and a SIMD with 2.5 waves delivers 70-80 % of its "
                                 "float64 rate: profiles/r06_notes.md, profiles/r05_valu_throughput.txt), with one tile per CU by the role with the longest "
                                 "instruction stream; the single-step launch is bound by load + launch latency (DESIGN.md 4, profiles/r03_pqx_stamps.txt)",
                         "valu": valu,
                         "issue_bound_evidence": issue_evidence(kname),
                         # the regime in which HBM IS the bound, beside the headline's own fraction (full legs: hbm_bound_fleet[_norm] below)
                         "hbm_bound_regime": ({"num_envs": 2097152, "timed_with": "100 dn_step launches replayed from one hipGraph",
                                               "normaliser_on": {k_: large_norm.get(k_) for k_ in ("us_per_vector_step", "achieved_GBps", "frac", "frac_of_copy_ceiling", "traffic_ratio")},
                                               "normaliser_off": {k_: large.get(k_) for k_ in ("us_per_vector_step", "achieved_GBps", "frac", "frac_of_copy_ceiling", "traffic_ratio")},
                                               "copy_ceiling_GBps": (copy_ceiling or {}).get("GBps")}
                                              if isinstance(large_norm, dict) and isinstance(large, dict) and "frac" in large_norm and "frac" in large else None)},
            "single_step": single_step,
            "hover_band": hover_leg,
            ("normalize_obs_on" if other_norm else "normalize_obs_off"): norm_leg,
            "hbm_bound_fleet_norm": large_norm,
            "hbm_bound_fleet": large,
            "hbm_copy_ceiling": copy_ceiling,
            "other_launch_shapes": others,
            "fleet_sweep": fleet_sweep,
        }
        # the two side legs must never cost the headline line: a failure is reported in place of the numbers
        if world == 1 and not args.no_ppo_rollout:
            try:
                line["ppo_rollout"] = ppo_rollout(pkg, track, n, max_steps, dev, rank)
            except Exception as exc:  # noqa: BLE001
                line["ppo_rollout"] = {"error": f"{type(exc).__name__}: {exc}"}
            try:
                line["sac_collect"] = sac_collect(pkg, track, n, max_steps, dev, rank)
            except Exception as exc:  # noqa: BLE001
                line["sac_collect"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(track, n, max_steps, args.cpu_baseline_seconds, args.normalize_obs)
            except Exception as exc:  # noqa: BLE001
                line["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}
    # BASELINE configs[3]: the one collective of the design (RCCL all-gather of advantages / returns per rollout), timed on every rank
    # whenever WORLD_SIZE > 1.  It runs AFTER the headline line is assembled: an exception is reported inside the line, and if the
    # collective hangs a watchdog thread ends THIS process with a non-zero code (never a re-exec) -- rank 0 printing the headline it
    # already holds first, so that a scaling run does not lose its measurement to the optional leg.
    want_sharded = (world > 1) if args.ppo_sharded is None else args.ppo_sharded
    if want_sharded and dist is not None:
        import threading

        def _overrun():
            print(f"bench.py: rank {rank}: the sharded PPO leg (RCCL all-gather) did not finish within {args.sharded_timeout:.0f} s; "
                  "exiting with code 3", file=sys.stderr, flush=True)
            if rank == 0 and line is not None:
                line["ppo_rollout_sharded"] = {"error": f"timeout after {args.sharded_timeout:.0f} s (watchdog; exit code 3)"}
                print(json.dumps(line), flush=True)
            os._exit(3)
        dog = threading.Timer(args.sharded_timeout, _overrun)
        dog.daemon = True
        dog.start()
        try:
            sharded = ppo_rollout_sharded(pkg, track, n, max_steps, dev, rank, world, dist)
        except Exception as exc:  # noqa: BLE001
            sharded = {"error": f"{type(exc).__name__}: {exc}"}
        finally:
            dog.cancel()
        if rank == 0:
            line["ppo_rollout_sharded"] = sharded
    env.close()
    if dist is not None:
        # RCCL writes its NCCL_DEBUG=VERSION banner to STDOUT through C stdio (buffered on a pipe, flushed at exit): push every
        # rank's out now, then let rank 0 print the one JSON line after everybody has done so
        import ctypes
        ctypes.CDLL(None).fflush(None)
        dist.barrier(device_ids=[local_rank])
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()
        import ctypes
        ctypes.CDLL(None).fflush(None)


if __name__ == "__main__":
    main()
