#!/bin/bash
# Register, spill and LDS figures of the traversal kernels for a set of -D flags (build container; compiles to ISA only).
# Usage: tools/kernel_regs.sh "-DRT_PRIM_PAIR=1" [kernel-name-pattern]
cd "$(dirname "$0")/.." || exit 1
out=$(mktemp /tmp/rt_regs_XXXX.s)
/opt/rocm/bin/hipcc -std=c++17 -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize $1 -Iinclude -Iraytracer_amd/csrc \
  -S --cuda-device-only raytracer_amd/csrc/rt_device.hip -o "$out" 2>/dev/null || { echo "compile failed"; exit 1; }
pat=${2:-'k_extendILb0|k_shadowILb0|k_shadow_then_extendILb0|k_shade'}
awk -v pat="$pat" '/^  - \.agpr_count/{blk=""} {blk=blk"\n"$0} /\.wavefront_size/{ if (blk ~ pat) { n=split(blk, L, "\n"); for(i=1;i<=n;i++) if (L[i] ~ /\.name:|vgpr_count|sgpr_count|spill_count|group_segment_fixed|private_segment_fixed/) printf "%s ", L[i]; print "" } blk="" }' "$out" | sed 's/  */ /g'
echo "asm: $out"
