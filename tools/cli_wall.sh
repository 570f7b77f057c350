#!/bin/bash
# GPU box: wall time of the whole command line on the bench scene (scene files -> PNG), host BVH and device BVH
set -e
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from tools import gen_scenes
os.makedirs("/tmp/cli_scene", exist_ok=True)
gen_scenes.scene_readme("/tmp/cli_scene", "readme", size=1080, spp=256, grid=(512, 512), texsize=512)
PY
cd /tmp/cli_scene
for flags in "" "--device-bvh"; do
  for run in first second; do
    t0=$(date +%s%N)
    timeout -k 10 200 $GRAFT_REPO_ROOT/raytracer_amd/raytracer --stats $flags readme.sdml
    t1=$(date +%s%N)
    echo "wall $(( (t1 - t0) / 1000000 )) ms ($flags, $run run)"
  done
done
ls -la /tmp/cli_scene/*.png | head
