#!/bin/bash
# profiles/pmc_fused.sh <tag> <DN_WAVES> <n> <norm>: instruction-mix counters of the fused kernel (K = 64, 30 launches), one pass per counter pair
set -u
TAG=$1; export DN_WAVES=$2; N=$3; NORM=$4
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT="$ROOT/gpurun_out/pmc_$TAG"
mkdir -p "$OUT"
cd /tmp
i=0
FAILED=""
for pair in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
            "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_WR" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F32"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $pair -d "$OUT/p$i" -o p$i -- python3 "$ROOT/profiles/run_fused.py" $N $NORM 64 30 > "$OUT/log_p$i.txt" 2>&1 || FAILED="$FAILED p$i($pair)"
done
cd "$ROOT"
mkdir -p "$ROOT/gpurun_out/r4"
python3 profiles/instmix.py "$OUT" > "$ROOT/gpurun_out/r4/pmc_$TAG.txt"
if [ -n "$FAILED" ]; then
    echo "# FAILED passes (logs kept under gpurun_out/pmc_$TAG/): $FAILED -- this summary is incomplete" >> "$ROOT/gpurun_out/r4/pmc_$TAG.txt"
    find "$OUT" -name "*.db" -delete
else
    rm -rf "$OUT"
fi
cat "$ROOT/gpurun_out/r4/pmc_$TAG.txt"
[ -z "$FAILED" ] || exit 1
