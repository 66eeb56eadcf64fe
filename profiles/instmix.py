"""Per-kernel averages of the counters collected by profiles/instmix.sh:  python profiles/instmix.py <dir>"""
import glob, os, sqlite3, sys

d = sys.argv[1]
vals = {}
for db in sorted(glob.glob(os.path.join(d, "p*", "p*_results.db"))):
    con = sqlite3.connect(db)
    try:
        for name, ctr, avg, n in con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                             "where kernel_name like '%dn_step_many%' group by kernel_name, counter_name"):
            vals.setdefault(name, {})[ctr] = (avg, n)
    finally:
        con.close()
for name, c in vals.items():
    print(f"\n## {name}")
    for ctr in sorted(c):
        print(f"  {ctr:28s} {c[ctr][0]:16.0f} per launch   (x{c[ctr][1]} launches)")
