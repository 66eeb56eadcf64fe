#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun):   profiles/prof.sh <tag> <bench.py args...>
# Three separate rocprofv3 passes of the same bench command -- kernel trace + stats, then one PMC counter per pass
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc is never combined with other trace domains):
#   gpurun_out/prof_<tag>/{stats,fetch,write}/...   ->   python profiles/summarize.py gpurun_out/prof_<tag> > profiles/<name>.txt
TAG=$1; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python3 $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench_stats.log 2>&1
# the counter passes run the bench with --profile-lite: under --pmc every dispatch costs tens of milliseconds, and the counters of a
# kernel do not depend on how many times it is launched
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python3 $ROOT/bench.py --no-cpu-baseline --profile-lite "$@" > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write -- python3 $ROOT/bench.py --no-cpu-baseline --profile-lite "$@" > $OUT/bench_write.log 2>&1
ls $OUT/*/ | head -20
