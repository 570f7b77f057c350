#!/bin/bash
# GPU box: replay one scene of tools/fuzz_parity.py in detail with the in-tree library and with every variants/lib_*.so
SEED=$1; SCENE=$2
timeout -k 10 200 python tools/fuzz_parity.py --seed $SEED --only $SCENE > gpurun_out/replay_intree.txt 2>&1 || exit 1
shopt -s nullglob
for f in variants/lib_*.so; do n=$(basename $f .so); RT_AMD_DEVICE_LIB=$f timeout -k 10 200 python tools/fuzz_parity.py --seed $SEED --only $SCENE > gpurun_out/replay_$n.txt 2>&1 || exit 1; done
