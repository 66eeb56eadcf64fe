"""Single-step launch (dn_step) sweep: us per step over fleet sizes, plain and with the observation normaliser, for the kernel shape
forced by DN_WAVES_SINGLE=1|3 (one wave per 64 drones | three waves cut by dependency).  AB_GRAPH=1 replays the launches from a
hipGraph (device timeline only); otherwise a ctypes loop of dn_step calls (host launch cost included).  -> profiles/r02_sweep_single.txt"""
import sys, os, ctypes as C, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
tag = os.environ.get("DN_WAVES_SINGLE", "auto")
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4096, 16384, 32768, 49152, 65536, 98304, 131072]
from drl_dronenavigation_amd import _capi
import drl_dronenavigation_amd as pkg
from drl_dronenavigation_amd import tracks
lib = _capi.load()
dev = torch.device("cuda:0")
for n in sizes:
  for norm in (False, True):
    env = pkg.DroneVecEnv(tracks.reaching(), n, normalize_obs=norm, device="cuda:0")
    env.reset_tensor()
    g = torch.Generator(device="cpu").manual_seed(1)
    A = 64
    acts = (torch.rand((A, n, 4), generator=g) * 2 - 1).cuda()
    o = dict(obs=torch.empty((A, n, 13), device=dev), reward=torch.empty((A, n), device=dev), done=torch.empty((A, n), dtype=torch.uint8, device=dev),
             trunc=torch.empty((A, n), dtype=torch.uint8, device=dev), found=torch.empty((A, n), dtype=torch.int32, device=dev))
    ptrs = [(acts[j].data_ptr(), o["obs"][j].data_ptr(), o["reward"][j].data_ptr(), o["done"][j].data_ptr(), o["trunc"][j].data_ptr(), o["found"][j].data_ptr()) for j in range(A)]
    sptr = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    h = env._handle
    def run(k):
        for t in range(k):
            p = ptrs[t % A]
            lib.dn_step(h, p[0], p[1], p[2], p[3], p[4], p[5], None, None, None, None, sptr)
    if os.environ.get("AB_GRAPH"):
        side = torch.cuda.Stream(dev); side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            sp = C.c_void_p(side.cuda_stream)
            for p_ in ptrs[:2]: lib.dn_step(h, p_[0], p_[1], p_[2], p_[3], p_[4], p_[5], None, None, None, None, sp)
        torch.cuda.current_stream(dev).wait_stream(side); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            for p_ in ptrs: lib.dn_step(h, p_[0], p_[1], p_[2], p_[3], p_[4], p_[5], None, None, None, None, sp)
        def run(k):
            for _ in range(k // A): gr.replay()
    run(30000 // A * A); torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(8192); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 8192)
    print(tag, n, "norm" if norm else "plain", "waves", env.kernel_waves(fused=False), f"{best:.3f} us/step", flush=True)
    env.close()
