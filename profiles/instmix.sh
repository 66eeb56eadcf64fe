#!/bin/bash
# Instruction-mix counters of the fused step kernel (run on the GPU box through gpurun):   profiles/instmix.sh <tag>
# One rocprofv3 pass per counter pair (kernel trace only; --pmc is never combined with other trace domains), the bench in
# --profile-lite mode; profiles/instmix.py <dir> prints per-launch and (DN_IM_K=<steps per launch> DN_IM_N=<drones>) per 64-drone-step values.
set -eu
TAG=${1:?usage: profiles/instmix.sh <tag> [bench args]}; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT="$ROOT/gpurun_out/instmix_$TAG"
mkdir -p "$OUT"
cd /tmp
i=0
FAILED=""
for pair in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
            "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT" \
            "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $pair -d "$OUT/p$i" -o p$i -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-ppo-rollout --profile-lite "$@" > "$OUT/bench_p$i.log" 2>&1 || FAILED="$FAILED p$i($pair)"
done
cd "$ROOT"
python3 profiles/instmix.py "$OUT" ${DN_IM_K:-0} ${DN_IM_N:-32768} > "$ROOT/gpurun_out/instmix_$TAG.txt"
if [ -n "$FAILED" ]; then
    echo "# FAILED passes (logs kept under gpurun_out/instmix_$TAG/): $FAILED -- this summary is incomplete" >> "$ROOT/gpurun_out/instmix_$TAG.txt"
    find "$OUT" -name "*.db" -delete
else
    rm -rf "$OUT"      # the raw databases exceed what gpurun copies back (TAG is non-empty: set -u / the usage check above)
fi
cat "$ROOT/gpurun_out/instmix_$TAG.txt"
[ -z "$FAILED" ] || exit 1
