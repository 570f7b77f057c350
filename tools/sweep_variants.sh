#!/bin/bash
# GPU box: run the throughput probe with the in-tree library and with every variant under variants/ (kernel experiments).
export PROBE_BOUNCES=${PROBE_BOUNCES:-4}
echo "in-tree" >> gpurun_out/sweep.log; timeout -k 10 100 python tools/lane_use_probe.py >> gpurun_out/sweep.log 2>&1 || exit 1
shopt -s nullglob
for f in variants/lib_*.so; do echo "$f" >> gpurun_out/sweep.log; RT_AMD_DEVICE_LIB=$f timeout -k 10 100 python tools/lane_use_probe.py >> gpurun_out/sweep.log 2>&1 || exit 1; done
