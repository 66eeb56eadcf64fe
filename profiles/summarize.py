#!/usr/bin/env python3
"""Turn the rocprofv3 (ROCm 7.2, rocpd sqlite) outputs written by profiles/prof.sh into a small text summary
that can be committed:  python profiles/summarize.py gpurun_out/prof_<tag> > profiles/<name>.txt

Reads  <dir>/stats/stats_results.db   (--kernel-trace --stats)
       <dir>/fetch/fetch_results.db   (--kernel-trace --pmc FETCH_SIZE)   separate pass
       <dir>/write/write_results.db   (--kernel-trace --pmc WRITE_SIZE)   separate pass
HBM bytes follow MI355X_MICROARCH.md section HBM: both counters are in KB; on gfx950 FETCH_SIZE reports half
the bytes of a wide (16 B/lane) coalesced read stream, so the read side is doubled; WRITE_SIZE is calibrated
on dn_fill4_kernel (N x 16 B written, exact)."""
import glob
import os
import sqlite3
import sys


def q(path, sql):
    db = sqlite3.connect(path)
    try:
        return list(db.execute(sql))
    finally:
        db.close()


def main(d):
    out = []
    for log in sorted(glob.glob(os.path.join(d, "bench_*.log"))):
        for line in open(log):
            if line.startswith("{\"metric\""):
                out.append(f"# {os.path.basename(log)}: {line.strip()}")
    st = os.path.join(d, "stats", "stats_results.db")
    if os.path.exists(st):
        out.append("\n## rocprofv3 --kernel-trace --stats (durations in us)")
        out.append(f"{'calls':>8} {'total_us':>12} {'avg_us':>10} {'pct':>7}  kernel")
        for name, calls, total, avg, pct in q(st, "select name,total_calls,total_duration,average,percentage from top_kernels limit 12"):
            out.append(f"{calls:8d} {total:12.1f} {avg:10.3f} {pct:7.2f}  {name}")
        rows = q(st, "select name, count(*), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), "
                     "max(lds_size), max(grid_x), max(workgroup_x) from kernels where name like '%dn_%' group by name")
        out.append("\n## dn_* kernels: calls, avg/min/max ns, VGPR, SGPR, LDS bytes, grid, workgroup")
        for r in rows:
            out.append("  " + " | ".join(str(x) for x in r))
    vals = {}
    for tag, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        p = os.path.join(d, tag, f"{tag}_results.db")
        if not os.path.exists(p):
            continue
        out.append(f"\n## rocprofv3 --kernel-trace --pmc {ctr} (separate pass; KB per dispatch, raw)")
        for name, avg, n in q(p, f"select kernel_name, avg(value), count(*) from counters_collection where counter_name='{ctr}' "
                                 "and kernel_name like '%dn_%' group by kernel_name"):
            out.append(f"  {avg:14.2f} KB  x{n:6d}  {name}")
            vals[(ctr, name)] = avg
    steps = [k[1] for k in vals if "dn_step_" in k[1]]
    for name in sorted(set(steps)):
        f, w = vals.get(("FETCH_SIZE", name)), vals.get(("WRITE_SIZE", name))
        if f is not None and w is not None:
            hbm = (2 * f + w) * 1024
            out.append(f"\n## HBM traffic per launch, {name}")
            out.append(f"  read  = 2 x FETCH_SIZE = {2 * f * 1024 / 1e6:.3f} MB   (gfx950 wide-load correction x2)")
            out.append(f"  write =     WRITE_SIZE = {w * 1024 / 1e6:.3f} MB")
            out.append(f"  total = {hbm / 1e6:.3f} MB per launch")
    out.append("\n(dn_step_many_Nw_kernel<R, NORM, NOISE, ONE, XOPT>: N = waves per 64 drones; ONE = true is the single-step launch "
               "dn_step, ONE = false the fused K-step launch dn_step_many; XOPT = the rarely used options (reward wrappers, N4 physics terms, RPM actions) compiled in)")
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1])
