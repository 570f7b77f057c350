#!/bin/bash
# Builds kernel-experiment variants of the HIP library into variants/lib_<name>.so (they travel to the GPU box with the
# tree; tools/sweep_variants.sh times them). Usage: tools/build_variants.sh name1="-DX=1 -DY=2" name2="..."
cd "$(dirname "$0")/.." || exit 1
mkdir -p variants
pids=()
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  ( /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize $flags -Iinclude -Iraytracer_amd/csrc \
      -shared raytracer_amd/csrc/rt_device.hip raytracer_amd/csrc/rt_bvh_build.hip -o variants/lib_$name.so 2> variants/$name.log || echo "build of $name failed" ) &
  pids+=($!)
  if (( ${#pids[@]} >= 6 )); then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
done
wait
ls -la variants/*.so
