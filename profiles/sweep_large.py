# us per vector step, one-wave kernel vs eight-role kernel, normaliser on: python3 profiles/sweep_large.py
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import drl_dronenavigation_amd as pkg
from drl_dronenavigation_amd import tracks
dev = torch.device("cuda:0")
def time_fused(env, acts, launches=20, reps=4):
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        out = env.rollout_tensor(acts)
        for _ in range(3): env.rollout_tensor(acts, out=out)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(launches): env.rollout_tensor(acts, out=out)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / launches)
    return best
for n in [int(x) for x in (sys.argv[1:] or "40960 49152 57344 65536 81920 98304 114688 131072 196608 262144".split())]:
    row = []
    for K in (20, 64):
        acts = torch.rand((K, n, 4), device=dev) * 2 - 1
        for w in ("1", "5", "8"):
            os.environ["DN_WAVES"] = w
            env = pkg.DroneVecEnv(tracks.reaching(), n, normalize_obs=True, device=dev)
            env.reset_tensor()
            for _ in range(3): env.rollout_tensor(acts)          # into a mixed fleet state
            row.append(f"K={K} w={env.kernel_waves(fused=True)}: {time_fused(env, acts) / K:.3f}")
            env.close()
    print(f"n={n} tiles/CU={n / 64 / 256:.2f}  " + "  ".join(row), flush=True)
