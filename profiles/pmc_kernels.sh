#!/bin/bash
# Per-kernel PMC averages of any python command (run on the GPU box through gpurun):
#   profiles/pmc_kernels.sh <tag> "<CTR1 CTR2>" ["<CTR3 CTR4>" ...] -- <script.py> [args...]
# One rocprofv3 --kernel-trace --pmc pass per quoted counter group (never combined with other trace domains); prints and
# writes gpurun_out/pmc_<tag>.txt: counter, kernel, average per dispatch, dispatches.
set -eu
TAG=${1:?usage: profiles/pmc_kernels.sh <tag> "<counters>" ... -- script.py [args]}; shift
GROUPS_=()
while [ "$#" -gt 0 ] && [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
[ "$#" -gt 1 ] || { echo "missing -- <script.py>"; exit 2; }
shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT="$ROOT/gpurun_out/pmc_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
SCRIPT="$1"; shift
cd /tmp
i=0
FAILED=""
for grp in "${GROUPS_[@]}"; do
    i=$((i+1))
    # (a pass that aborts inside rocprofv3 can hang in its signal handler for the rest of the call: bounded)
    timeout -k 10 ${DN_PMC_PASS_TIMEOUT:-600} rocprofv3 --kernel-trace --pmc $grp -d "$OUT/p$i" -o p$i -- python3 "$ROOT/$SCRIPT" "$@" > "$OUT/run_p$i.log" 2>&1 || FAILED="$FAILED p$i($grp)"
done
cd "$ROOT"
python3 - "$OUT" > "$ROOT/gpurun_out/pmc_$TAG.txt" <<'PY'
import glob, sqlite3, sys
rows = {}
for p in sorted(glob.glob(sys.argv[1] + "/p*/**/*.db", recursive=True)):
    db = sqlite3.connect(p)
    try:
        for k, c, avg, n in db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                       "where kernel_name like '%dn_%' group by kernel_name, counter_name"):
            rows[(k, c)] = (avg, n)
    except Exception as e:  # noqa: BLE001
        print("#", p, e)
    db.close()
for k in sorted({k for k, _ in rows}):
    print("##", k)
    for (kk, c), (avg, n) in sorted(rows.items()):
        if kk == k:
            print(f"  {c:32s} {avg:18.1f} per dispatch   (x{n})")
PY
if [ -n "$FAILED" ]; then
    echo "# FAILED passes (logs kept under gpurun_out/pmc_$TAG/): $FAILED -- this summary is incomplete" >> "$ROOT/gpurun_out/pmc_$TAG.txt"
    find "$OUT" -name "*.db" -delete
else
    rm -rf "$OUT"
fi
cat "$ROOT/gpurun_out/pmc_$TAG.txt"
[ -z "$FAILED" ] || exit 1
