#!/bin/bash
# GPU box: frames of the in-tree library and of variants/lib_old.so on the same scenes (sha1 per frame), then the timing probe of both.
set -e
timeout -k 10 300 python tools/diff_vs_oracle.py > gpurun_out/ab_new.txt 2>&1
RT_AMD_DEVICE_LIB=variants/lib_old.so timeout -k 10 300 python tools/diff_vs_oracle.py > gpurun_out/ab_old.txt 2>&1
export PROBE_BOUNCES=4
timeout -k 10 100 python tools/lane_use_probe.py > gpurun_out/ab_probe_new.txt 2>&1
RT_AMD_DEVICE_LIB=variants/lib_old.so timeout -k 10 100 python tools/lane_use_probe.py > gpurun_out/ab_probe_old.txt 2>&1
