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
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/bench_stats.log" 2>&1 || true
if [ "${DN_PROF_PMC:-1}" = "1" ]; then
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-ppo-rollout --profile-lite "$@" > "$OUT/bench_fetch.log" 2>&1 || true
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o write -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-ppo-rollout --profile-lite "$@" > "$OUT/bench_write.log" 2>&1 || true
fi
cd "$ROOT"
python3 profiles/summarize.py "$OUT" > "$ROOT/gpurun_out/prof_$TAG.txt"
find "$OUT" -name "*.db" -size +20M -delete        # the raw databases exceed what gpurun copies back
tail -n 60 "$ROOT/gpurun_out/prof_$TAG.txt"
