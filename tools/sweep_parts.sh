#!/bin/bash
# GPU box: tools/part_overhead_probe.py with the in-tree library and with every variant under variants/
echo "in-tree" >> gpurun_out/sweep_parts.log; timeout -k 10 200 python tools/part_overhead_probe.py >> gpurun_out/sweep_parts.log 2>&1 || exit 1
shopt -s nullglob
for f in variants/lib_*.so; do echo "$f" >> gpurun_out/sweep_parts.log; RT_AMD_DEVICE_LIB=$f timeout -k 10 200 python tools/part_overhead_probe.py >> gpurun_out/sweep_parts.log 2>&1 || exit 1; done
