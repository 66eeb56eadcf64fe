#!/usr/bin/env python3
"""Times the fused MFMA policy/value MLP (dn_mlp_forward) against torch: python profiles/bench_mlp.py [num_envs]
DN_MLP_SHAPE=1 selects the one-wave-per-workgroup kernel (weights straight from L2), default 4 (weights through LDS)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import drl_dronenavigation_amd as pkg  # noqa: E402
from drl_dronenavigation_amd import policy_mfma as pm  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
net = pkg.MlpActorCritic().to(dev)
pol = pm.FusedMlpPolicy(net, n, dev)
obs = torch.rand(n, 13, device=dev)


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


flop = 2 * (13 * 512 + 512 * 512 + 512 * 256 + 256 * 4) * n
t2 = timeit(lambda: pm.mlp_forward([pol.pi, pol.vf], obs, [pol._mean, pol._value]))
t1 = timeit(lambda: pm.mlp_forward([pol.vf], obs, [pol._value]))
print(f"n={n}: fused pi+vf {t2:.1f} us ({2 * flop / t2 / 1e6:.0f} TFLOP/s), vf only {t1:.1f} us ({flop / t1 / 1e6:.0f} TFLOP/s)")
for grade, mult in (("fp16", 1), ("fp32", 3)):                      # the other grades of the PPO networks
    pg = pm.FusedMlpPolicy(net, n, dev, grade=grade)
    tg = timeit(lambda: pm.mlp_forward([pg.pi, pg.vf], obs, [pg._mean, pg._value]))
    print(f"n={n}: grade {grade} pi+vf {tg:.1f} us ({2 * mult * flop / tg / 1e6:.0f} TFLOP/s of 16-bit MFMA work)")
actor = pkg.SacActor().to(dev)                                       # the SAC actor, 13-256-256-8
flop_sac = 2 * (13 * 256 + 256 * 256 + 256 * 8) * n
for grade, mult in (("bf16", 1), ("fp16", 1), ("fp32", 3)):
    fa = pm.FusedSacActor(actor, n, dev, grade=grade)
    ts = timeit(lambda: fa.mean_log_std(obs))
    print(f"n={n}: SAC actor grade {grade} {ts:.1f} us ({mult * flop_sac / ts / 1e6:.0f} TFLOP/s of 16-bit MFMA work)")
with torch.no_grad():
    tt = timeit(lambda: (net.action_net(net.pi(obs)), net.value_net(net.vf(obs))))
    net.trunk_dtype = torch.bfloat16
    tb = timeit(lambda: net(obs, deterministic=True))
print(f"torch fp32 pi+vf {tt:.1f} us; torch bf16-trunk forward {tb:.1f} us")
