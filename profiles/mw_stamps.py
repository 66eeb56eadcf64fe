# Per-role timeline of one tile of the fused multi-wave step (timing build: -DDN_MW_STAMP=<tile> on dn_kernels_mw.hip, selected with
# DN_LIB_PATH): for each role the busy cycles (barrier release -> next barrier arrival) and the wait at the barrier, iterations 8..55.
#   DN_LIB_PATH=scratch/r4/lib_st300.so python profiles/mw_stamps.py [n_drones] [norm 0|1] [K]
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import drl_dronenavigation_amd as pkg
from drl_dronenavigation_amd import tracks

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
norm = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
K = int(sys.argv[3]) if len(sys.argv) > 3 else 64
NR = int(os.environ.get("MW_ROLES", "8"))
dev = torch.device("cuda:0")
raw = C.CDLL(pkg._capi.library_path())
buf = (C.c_longlong * (8 * 48 * 2))()
env = pkg.DroneVecEnv(tracks.REGISTRY["reaching"](), n, max_steps=4096, normalize_obs=norm, seed=1, device=dev)
env.reset_tensor()
print("waves", env.kernel_waves(fused=True))
acts = torch.rand((K, n, 4), device=dev) * 2 - 1
for _ in range(20):
    env.rollout_tensor(acts)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    env.rollout_tensor(acts)
e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) * 1e3 / 50 / K:.3f} us per vector step (eager launches of {K} steps)")
assert raw.dn_debug_mw_stamps(buf) == 0
v = np.array(list(buf), dtype=np.int64).reshape(8, 48, 2)
names = os.environ.get("MW_NAMES", "LAQXN___")
for r in range(NR):
    if v[r].max() == 0:
        continue
    arrive, release = v[r, :, 0], v[r, :, 1]
    busy = arrive[1:] - release[:-1]
    wait = release - arrive
    period = release[1:] - release[:-1]
    print(f"role {r} {names[r]}: busy mean {busy.mean():7.0f} (min {busy.min()}, max {busy.max()})  barrier wait mean {wait.mean():6.0f}  period {period.mean():7.0f}")
rel = v[:NR, :, 1]
ok = rel.max(axis=1) > 0
print("release skew between roles (cycles, mean over iterations):", (rel[ok].max(axis=0) - rel[ok].min(axis=0)).mean())
arr = v[:NR, :, 0][ok]
last = arr.argmax(axis=0)
print("last role to arrive, histogram:", {names[int(np.flatnonzero(ok)[k])]: int((last == k).sum()) for k in range(ok.sum())})
if hasattr(raw, "dn_debug_mw_edges"):
    eb = (C.c_longlong * 64)()
    assert raw.dn_debug_mw_edges(eb) == 0
    e = np.array(list(eb), dtype=np.int64).reshape(8, 8)
    print(f"last launch, per role: cycles entry -> barrier P -> exit, and the role's wall time in the kernel (100 MHz counter)")
    for r in range(NR):
        if e[r, 0] == 0:
            continue
        hw = int(e[r, 6])                                   # HW_REG_HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
        print(f"  role {r} {names[r]}: prologue {e[r, 1] - e[r, 0]:6d}  loop {e[r, 2] - e[r, 1] if e[r, 2] else -1:7d}  epilogue {e[r, 3] - e[r, 2] if e[r, 2] else -1:6d}  (loop+epilogue {e[r, 3] - e[r, 1]:7d})  wall {(e[r, 5] - e[r, 4]) * 10} ns   "
              f"HW_ID {hw:#010x}: simd {(hw >> 4) & 3} wave slot {hw & 15} cu {(hw >> 8) & 15} sh {(hw >> 12) & 1} se {(hw >> 13) & 7}")
    w0, w1 = e[:NR, 4][e[:NR, 4] > 0].min(), e[:NR, 5].max()
    print(f"  tile wall time in the kernel {(w1 - w0) * 10} ns for {K} steps = {(w1 - w0) * 10 / K / 1000:.3f} us per step; eager launch-to-launch above")
env.close()
