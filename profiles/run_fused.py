# N fused launches for profiling: python3 run_fused.py n norm K launches   (DN_WAVES from the environment)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import drl_dronenavigation_amd as pkg
from drl_dronenavigation_amd import tracks
n = int(sys.argv[1]); norm = bool(int(sys.argv[2])); K = int(sys.argv[3]); L = int(sys.argv[4])
dev = torch.device("cuda:0")
env = pkg.DroneVecEnv(tracks.REGISTRY["reaching"](), n, max_steps=4096, normalize_obs=norm, seed=1, device=dev)
env.reset_tensor()
acts = torch.rand((K, n, 4), device=dev) * 2 - 1
out = env.rollout_tensor(acts)
for _ in range(L): env.rollout_tensor(acts, out=out)
torch.cuda.synchronize()
print("waves", env.kernel_waves(fused=True))
env.close()
