# N single-step launches of a fleet for profiling: python3 run_single.py n norm launches
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import drl_dronenavigation_amd as pkg
from drl_dronenavigation_amd import tracks
n = int(sys.argv[1]); norm = bool(int(sys.argv[2])); L = int(sys.argv[3])
dev = torch.device("cuda:0")
env = pkg.DroneVecEnv(tracks.REGISTRY["reaching"](), n, max_steps=4096, normalize_obs=norm, seed=1, device=dev)
env.reset_tensor()
acts = torch.rand((2, n, 4), device=dev) * 2 - 1
for t in range(L): env.step_tensor(acts[t & 1])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for t in range(L): env.step_tensor(acts[t & 1])
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / L
b = 720 if norm else 288
print(f"n={n} norm={norm} waves={env.kernel_waves(fused=False)}: {us:.2f} us per step, {b * n / us / 1e3:.0f} GB/s algorithmic = {b * n / us / 1e3 / 8000:.3f} of 8 TB/s")
env.close()
