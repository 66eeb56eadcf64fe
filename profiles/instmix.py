"""Per-kernel averages of the counters collected by profiles/instmix.sh:  python profiles/instmix.py <dir> [K [N]]

With K (vector steps per fused launch) and N (drones) the per-launch averages are also reduced to what profiles/instmix.json keys:
vector instructions per 64-drone tile-step, ALU cycles per vector instruction, the float64 / conversion / transcendental mix and the
kernel's duration in the pass -- printed as one JSON object."""
import glob
import json
import os
import sqlite3
import sys

d = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
vals, durs = {}, {}
for db in sorted(glob.glob(os.path.join(d, "p*", "p*_results.db"))):
    con = sqlite3.connect(db)
    try:
        ctrs = set()
        for name, ctr, avg, n in con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                             "where kernel_name like '%dn_step_%' group by kernel_name, counter_name"):
            vals.setdefault(name, {})[ctr] = (avg, n)
            ctrs.add(ctr)
        try:      # the kernel's duration in THIS pass (ns): the clock below pairs it with the cycle counter of the same dispatches
            for name, avg, n in con.execute("select name, avg(duration), count(*) from kernels where name like '%dn_step_%' group by name"):
                for ctr in ctrs:
                    durs.setdefault(name, {})[ctr] = (avg, n)
        except sqlite3.Error:
            pass
    finally:
        con.close()
derived = {}
for name, c in vals.items():
    print(f"\n## {name}")
    for ctr in sorted(c):
        extra = ""
        if name in durs and ctr in durs[name]:
            extra = f"   kernel avg {durs[name][ctr][0] / 1e3:9.3f} us in this pass"
        print(f"  {ctr:28s} {c[ctr][0]:16.0f} per launch   (x{c[ctr][1]} launches){extra}")
    if K and "SQ_INSTS_VALU" in c:
        tile_steps = (N + 63) // 64 * K
        g = lambda k_: c[k_][0] if k_ in c else None      # noqa: E731
        e = {"tile_steps_per_launch": tile_steps, "SQ_INSTS_VALU": round(g("SQ_INSTS_VALU")),
             "valu_instructions_per_64_drone_step": round(g("SQ_INSTS_VALU") / tile_steps, 1)}
        if g("SQ_ACTIVE_INST_VALU") is not None:
            e["SQ_ACTIVE_INST_VALU_quad_cycles"] = round(g("SQ_ACTIVE_INST_VALU"))
            e["alu_cycles_per_valu_instruction"] = round(4.0 * g("SQ_ACTIVE_INST_VALU") / g("SQ_INSTS_VALU"), 3)
        for key, ctr in (("salu_instructions_per_64_drone_step", "SQ_INSTS_SALU"), ("lds_instructions_per_64_drone_step", "SQ_INSTS_LDS"),
                         ("branches_per_64_drone_step", "SQ_INSTS_BRANCH"), ("conversions_per_64_drone_step", "SQ_INSTS_VALU_CVT"),
                         ("f64_trans_per_64_drone_step", "SQ_INSTS_VALU_TRANS_F64"), ("f32_trans_per_64_drone_step", "SQ_INSTS_VALU_TRANS_F32"),
                         ("int32_per_64_drone_step", "SQ_INSTS_VALU_INT32")):
            if g(ctr) is not None:
                e[key] = round(g(ctr) / tile_steps, 1)
        if all(g(k_) is not None for k_ in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64")):
            e["f64_fma_mul_add_per_64_drone_step"] = [round(g(k_) / tile_steps, 1) for k_ in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64")]
        if name in durs and "SQ_INSTS_VALU" in durs[name]:
            e["kernel_avg_us_in_the_SQ_INSTS_VALU_pass"] = round(durs[name]["SQ_INSTS_VALU"][0] / 1e3, 3)
        # (no shader clock from GRBM_GUI_ACTIVE: its per-dispatch value differed 7x between two identical passes of round 6 -- it is
        # aggregated over an instance count / window this tool does not control.  The clock comes from the stamped build: mw_stamps.py.)
        if g("SQ_WAIT_INST_ANY") is not None and g("SQ_WAVE_CYCLES") is not None:
            e["wave_cycles_waiting_for_issue"] = round(g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 3)
        if g("SQ_WAIT_ANY") is not None and g("SQ_WAVE_CYCLES") is not None:
            e["wave_cycles_waiting_any"] = round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 3)
        derived[name] = e
if derived:
    print("\n## derived (profiles/instmix.json entries; K = %d, N = %d)" % (K, N))
    print(json.dumps(derived, indent=1))
