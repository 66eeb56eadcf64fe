# per-role timeline of one tile of the three-wave single step (timing build: scratch/r3/build_kvariant.sh <name> -DDN_PQX_STAMP=<tile>)
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import drl_dronenavigation_amd as pkg
from drl_dronenavigation_amd import tracks
dev = torch.device("cuda:0")
n = 32768
raw = C.CDLL(pkg._capi.library_path())
buf = (C.c_longlong * 48)()
names = ["start", "->B1", "B1", "->B2", "B2", "->B3", "B3", "end"]
for mode in sys.argv[1:] or ["sac", "ppo", "plain_norm", "plain"]:
    kw = dict(act_noise_sigma=0.002, obs_noise_sigma=0.01) if mode == "sac" else {}
    env = pkg.DroneVecEnv(tracks.REGISTRY["reaching"](), n, max_steps=4096, normalize_obs=mode != "plain", seed=1, device=dev, **kw)
    env.reset_tensor()
    lib, h = pkg._capi.load(), env._handle
    mls = torch.randn((n, 8), device=dev) * 0.3
    mean = torch.randn((n, 4), device=dev) * 0.3
    lstd = torch.zeros(4, device=dev)
    acts = torch.rand((n, 4), device=dev) * 2 - 1
    b = dict(lp=torch.zeros(n, device=dev), act=torch.zeros((n, 4), device=dev), obs=torch.zeros((n, 13), device=dev), rew=torch.zeros(n, device=dev), done=torch.zeros(n, dtype=torch.uint8, device=dev),
             trunc=torch.zeros(n, dtype=torch.uint8, device=dev), found=torch.zeros(n, dtype=torch.int32, device=dev), term=torch.zeros((n, 13), device=dev))
    fp = lambda t: C.cast(t.data_ptr(), C.POINTER(C.c_float))
    def go(sp):
        if mode == "sac":
            pkg._capi.check(lib.dn_step_squashed(h, mls.data_ptr(), 1, 0, b["act"].data_ptr(), None, b["obs"].data_ptr(), b["rew"].data_ptr(), b["done"].data_ptr(),
                                                 b["trunc"].data_ptr(), b["found"].data_ptr(), b["term"].data_ptr(), None, None, None, sp))
        elif mode == "ppo":
            pkg._capi.check(lib.dn_step_sampled(h, fp(mean), fp(lstd), 1, 0, fp(b["act"]), fp(b["lp"]), fp(b["obs"]), fp(b["rew"]), b["done"].data_ptr(),
                                                b["trunc"].data_ptr(), b["found"].data_ptr(), fp(b["term"]), None, None, None, sp))
        else:
            pkg._capi.check(lib.dn_step(h, fp(acts), fp(b["obs"]), fp(b["rew"]), b["done"].data_ptr(), b["trunc"].data_ptr(), b["found"].data_ptr(), fp(b["term"]),
                                        None, None, None, sp))
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        sp = C.c_void_p(s.cuda_stream)
        for _ in range(3): go(sp)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(40): go(sp)
    torch.cuda.synchronize()
    for rep in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
        if rep < 4: continue
        assert raw.dn_debug_pqx_stamps(buf) == 0
        v = [[buf[r * 16 + k] for k in range(8)] for r in range(3)]
        wall = [(buf[r * 16 + 9] - buf[r * 16 + 8]) * 10 for r in range(3)]     # ns (100 MHz counter)
        print("    wall ns per role:", wall, " => clock", [round((v[r][7] - v[r][0]) / max(wall[r], 1), 2) for r in range(3)], "GHz")
        t0 = min(r[0] for r in v)
        print(f"{mode}: {e0.elapsed_time(e1) * 1e3 / 200:.2f} us per launch (graph replay); last launch, tile {os.environ.get('TILE', '?')}, cycles from the tile's first mark:")
        for r, nm in enumerate("PQX"):
            print("   ", nm, " ".join(f"{names[k]}={v[r][k] - t0}" for k in range(8)))
    env.close()
