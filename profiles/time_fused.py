# us per vector step of the fused launch (K = 64 and 20), hipGraph replay of 20 launches, best of 6:
#   python3 profiles/time_fused.py <drones> <normaliser 0|1> [uniform|hover]   (DN_LIB_PATH / DN_WAVES from the environment)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import drl_dronenavigation_amd as pkg
from drl_dronenavigation_amd import tracks
n = int(sys.argv[1]); norm = bool(int(sys.argv[2])); dist = sys.argv[3] if len(sys.argv) > 3 else "uniform"
dev = torch.device("cuda:0")
env = pkg.DroneVecEnv(tracks.REGISTRY["reaching"](), n, max_steps=4096, normalize_obs=norm, seed=1, device=dev)
env.reset_tensor()
for K in (64, 20):
    acts = (torch.rand((K, n, 4), device=dev) * 2 - 1) if dist == "uniform" else (0.0922 + 0.003 * torch.randn((K, n, 4), device=dev))
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        out = env.rollout_tensor(acts)
        for _ in range(5): env.rollout_tensor(acts, out=out)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20): env.rollout_tensor(acts, out=out)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 20 / K)
    print(f"n={n} norm={norm} actions={dist} waves={env.kernel_waves(fused=True)} K={K}: {best:.3f} us per vector step (graph of 20 launches)")
env.close()
