#!/bin/bash
# rocprofv3 recipe behind the summaries under profiles/ (run on the GPU box through gpurun):
#   profiles/prof.sh <tag> [bench.py arguments...]
# Three separate passes of the same bench command (kernel trace only; --pmc is never combined with other trace domains):
#   stats  rocprofv3 --kernel-trace --stats
#   fetch  rocprofv3 --kernel-trace --pmc FETCH_SIZE
#   write  rocprofv3 --kernel-trace --pmc WRITE_SIZE
# then profiles/summarize.py turns the databases into gpurun_out/prof_<tag>.txt (copy the ones to keep into profiles/).
set -eu
TAG=${1:?usage: profiles/prof.sh <tag> [bench args]}; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT="$ROOT/gpurun_out/prof_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
FAILED=""
pass() {   # pass <name> <rocprofv3 arguments...>: a failed pass is recorded in the summary and fails the script (its log is kept)
    local name=$1; shift
    if ! "$@" > "$OUT/bench_$name.log" 2>&1; then FAILED="$FAILED $name"; fi
}
pass stats rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" --no-cpu-baseline "$@"
if [ "${DN_PROF_PMC:-1}" = "1" ]; then
    pass fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-ppo-rollout --profile-lite "$@"
    pass write rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o write -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-ppo-rollout --profile-lite "$@"
fi
cd "$ROOT"
python3 profiles/summarize.py "$OUT" > "$ROOT/gpurun_out/prof_$TAG.txt"
for f in $FAILED; do echo "# pass $f FAILED (see gpurun_out/prof_$TAG/bench_$f.log): this summary is incomplete" >> "$ROOT/gpurun_out/prof_$TAG.txt"; done
find "$OUT" -name "*.db" -size +20M -delete        # the raw databases exceed what gpurun copies back
tail -n 60 "$ROOT/gpurun_out/prof_$TAG.txt"
[ -z "$FAILED" ] || { echo "profiles/prof.sh: failed passes:$FAILED" >&2; exit 1; }
