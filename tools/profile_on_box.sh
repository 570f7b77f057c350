#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + the PMC passes the roofline numbers come from.
# Every pass is its own rocprofv3 run with its own time limit; outputs land under gpurun_out/prof_<tag>/.
set -u
TAG=${1:-r1}
ARGS=${2:---steps 2 --warmup 1}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py $ARGS --cpu-seconds 0"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.log" 2>&1 || { echo "trace pass failed"; tail -5 "$OUT/trace.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o fetch -- $BENCH > "$OUT/pmc_fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$OUT/pmc_fetch.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o write -- $BENCH > "$OUT/pmc_write.log" 2>&1 || { echo "write pass failed"; tail -5 "$OUT/pmc_write.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d "$OUT/pmc_sq" -o sq -- $BENCH > "$OUT/pmc_sq.log" 2>&1 || { echo "sq pass failed"; tail -5 "$OUT/pmc_sq.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE -d "$OUT/pmc_cache" -o cache -- $BENCH > "$OUT/pmc_cache.log" 2>&1 || { echo "cache pass failed"; tail -5 "$OUT/pmc_cache.log"; exit 1; }
find "$OUT" -name '*.csv' | head -40
du -sh "$OUT"
# the record-fetch probe under the same cache counters: how many L1->L2 requests one scattered 64-byte record costs, and
# the rates it reaches un-profiled (the roofs bench.py prices the traversal kernels against)
timeout -s KILL 120 $ROOT/tools/micro/l1_access_probe > "$OUT/record_fetch_probe.txt" 2>&1 || echo "probe failed"
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d "$OUT/pmc_probe" -o probe -- $ROOT/tools/micro/l1_access_probe > "$OUT/pmc_probe.log" 2>&1 || echo "probe counter pass failed"
