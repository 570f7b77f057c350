#!/bin/bash
# Deeper PMC diagnostics of the traversal kernels (GPU box, through gpurun). One rocprofv3 run per counter group.
set -u
TAG=${1:-diag}
SPP=${2:-16}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --spp $SPP --steps 1 --warmup 0 --cpu-seconds 0"
run() { # name counters...
  local name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" -d "$OUT/$name" -o $name -- $BENCH > "$OUT/$name.log" 2>&1 || { echo "$name pass failed"; tail -5 "$OUT/$name.log"; exit 1; }
}
run sqa SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
run sqb SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY
run sqc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_IFETCH
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
du -sh "$OUT"
