#!/bin/bash
# Where the 2 M-drone single step spends its time, counter by counter, beside the copy kernel of the same process (run on the GPU box):
#   profiles/pmc_hbm_bound.sh <tag> [n] [norm]      -> gpurun_out/pmc_<tag>.txt
set -eu
TAG=${1:?usage: profiles/pmc_hbm_bound.sh <tag> [n] [norm]}; N=${2:-2097152}; NORM=${3:-1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
exec "$ROOT/profiles/pmc_kernels.sh" "$TAG" \
  "SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
  "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" \
  "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
  "TCC_TAG_STALL_sum TCC_BUSY_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_CYCLE_sum TCC_REQ_sum" \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "GRBM_GUI_ACTIVE GRBM_EA_BUSY" \
  -- profiles/run_hbm_bound.py "$N" "$NORM" 8
