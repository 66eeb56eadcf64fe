# The bandwidth-bound regime for profiling: L single-step launches of a 2 M-drone fleet and L stream copies of 1 GiB in one process
#   python3 profiles/run_hbm_bound.py [n] [norm 0|1] [launches] [warm-up steps]      (DN_LIB_PATH / DN_WAVES_SINGLE from the environment)
# warm-up steps: 4 = a freshly reset fleet (hardly an episode ends inside the timed launches); ~300 = the steady state bench.py's leg times,
# where a third of the tiles carry a finished drone (second normaliser pass, reset) in every step
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import drl_dronenavigation_amd as pkg
from drl_dronenavigation_amd import tracks
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2097152
norm = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
L = int(sys.argv[3]) if len(sys.argv) > 3 else 10
W = int(sys.argv[4]) if len(sys.argv) > 4 else 4
WT = os.environ.get('WANT_TERMINAL', '1') != '0'      # 0: no terminal_obs / ep_return / ep_length outputs (scattered rows of finished drones)
dev = torch.device("cuda:0")
env = pkg.DroneVecEnv(tracks.REGISTRY["reaching"](), n, max_steps=4096, normalize_obs=norm, seed=1, device=dev)
env.reset_tensor()
acts = torch.rand((2, n, 4), device=dev) * 2 - 1
src = torch.empty(1 << 28, dtype=torch.float32, device=dev).fill_(1.0)
dst = torch.empty_like(src)
for t in range(W):
    env.step_tensor(acts[t & 1])
    if t < 4: pkg.stream_copy(dst, src)
torch.cuda.synchronize()
def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(L): fn(t)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / L
us = timed(lambda t: env.step_tensor(acts[t & 1], want_terminal=WT))
uc = timed(lambda t: pkg.stream_copy(dst, src))
b = 720 if norm else 288
print(f"n={n} norm={norm} warm={W} terminal={int(WT)} waves={env.kernel_waves(fused=False)}: {us:.2f} us per step (eager), {b * n / us / 1e3:.0f} GB/s algorithmic = {b * n / us / 1e3 / 8000:.3f} of 8 TB/s; "
      f"copy {2 * src.numel() * 4 / uc / 1e3:.0f} GB/s; step / copy = {b * n / us / (2 * src.numel() * 4 / uc):.3f}")
env.close()
