#!/usr/bin/env python3
"""Device time per launch of every policy kernel, hipGraph-replayed (bench.py's `time_launches`): python profiles/time_mlp.py [num_envs] [reps] [ppo_fp32,sac_bf16,...]
PPO pair of networks (13-512-512-256, actor + critic in one launch) at the three grades and the SAC actor (13-256-256-8) at the three grades.
A/B of library variants: run it once per variant with DN_LIB_PATH set, interleaved (scratch/r6/ab_mlp.sh); weights are seeded, so the outputs'
checksums printed beside the times must agree between variants that claim identical arithmetic."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import drl_dronenavigation_amd as pkg  # noqa: E402
from drl_dronenavigation_amd import policy_mfma as pm  # noqa: E402

spec = importlib.util.spec_from_file_location("bench_for_time_mlp", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None          # e.g. ppo_fp32,sac_fp32
torch.manual_seed(7)
net = pkg.MlpActorCritic().to(dev)
actor = pkg.SacActor().to(dev)
obs = torch.randn(n, 13, device=dev).clamp_(-5, 5)
mean, val = torch.zeros((n, 4), device=dev), torch.zeros((n, 1), device=dev)
out = []
for grade, passes in (("bf16", 1), ("fp16", 1), ("fp32", 3)):
    if only and "ppo_" + grade not in only:
        continue
    pol = pm.FusedMlpPolicy(net, n, dev, grade=grade)
    us = bench.time_launches(torch, dev, lambda: pm.mlp_forward([pol.pi, pol.vf], obs, [mean, val]), reps=reps)
    flop = 2.0 * bench.PPO_MACS_MFMA * n * 2 * passes
    out.append(f"ppo_{grade} {us:7.2f} us (mfma_frac {flop / (us * 1e-6) / bench.MFMA_PEAK_FLOPS:.3f}) sum {float(mean.double().sum()):+.6f} {float(val.double().sum()):+.6f}")
for grade, passes in (("bf16", 1), ("fp16", 1), ("fp32", 3)):
    if only and "sac_" + grade not in only:
        continue
    fa = pm.FusedSacActor(actor, n, dev, grade=grade)
    us = bench.time_launches(torch, dev, lambda: fa.mean_log_std(obs), reps=reps)
    m, s = fa.mean_log_std(obs)
    flop = 2.0 * bench.SAC_MACS_MFMA * n * passes
    out.append(f"sac_{grade} {us:7.2f} us (mfma_frac {flop / (us * 1e-6) / bench.MFMA_PEAK_FLOPS:.3f}) sum {float(m.double().sum()):+.6f} {float(s.double().sum()):+.6f}")
print(f"n={n} reps={reps} lib={os.environ.get('DN_LIB_PATH', 'tree')}")
print("\n".join(out))
