#!/bin/bash
# GPU box: the launches that trace a bounce's visibility queries and the next bounce's rays together (used for parts of a
# frame) against the separate launches: same frames, then the partition tests and the strong-scaling probe both ways.
set -e
timeout -k 10 200 python tools/fused_debug.py > gpurun_out/fused_debug.txt 2>&1
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_multi_gpu_gloo.py -m gpu -x -q -k "partition or gather or config3 or counters" --timeout 200 --timeout-method thread > gpurun_out/t_part.log 2>&1
timeout -k 10 200 python tools/part_overhead_probe.py > gpurun_out/part_fused.txt 2>&1
RT_NO_FUSED_TRACE=1 timeout -k 10 200 python tools/part_overhead_probe.py > gpurun_out/part_unfused.txt 2>&1
