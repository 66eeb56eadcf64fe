# fused-launch sweep: us per vector step for fleets x shapes (DN_WAVES), K = 64, graph of 10 launches, best of 5
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import drl_dronenavigation_amd as pkg
from drl_dronenavigation_amd import tracks
dev = torch.device("cuda:0")
norm = bool(int(sys.argv[1]))
shapes = sys.argv[2].split(",")
sizes = [int(x) for x in sys.argv[3].split(",")]
K = int(sys.argv[4]) if len(sys.argv) > 4 else 64
for n in sizes:
    row = []
    for w in shapes:
        os.environ["DN_WAVES"] = w
        env = pkg.DroneVecEnv(tracks.REGISTRY["reaching"](), n, max_steps=4096, normalize_obs=norm, seed=1, device=dev)
        env.reset_tensor()
        acts = torch.rand((K, n, 4), device=dev) * 2 - 1
        s = torch.cuda.Stream(dev)
        with torch.cuda.stream(s):
            out = env.rollout_tensor(acts)
            for _ in range(5): env.rollout_tensor(acts, out=out)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(10): env.rollout_tensor(acts, out=out)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 10 / K)
        row.append(f"{env.kernel_waves(fused=True)}w {best:.3f}")
        env.close()
        del g
    print(f"n={n:6d} norm={int(norm)} K={K}: " + " | ".join(row), flush=True)
