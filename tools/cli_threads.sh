#!/bin/bash
# GPU box: host phases of the command line by host thread count (bench scene)
set -e
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from tools import gen_scenes
os.makedirs("/tmp/cli_scene", exist_ok=True)
gen_scenes.scene_readme("/tmp/cli_scene", "readme", size=1080, spp=8, grid=(512, 512), texsize=512)
PY
cd /tmp/cli_scene
for t in 1 2 4 8 16; do echo "-t$t"; timeout -k 10 200 $GRAFT_REPO_ROOT/raytracer_amd/raytracer -t$t --stats readme.sdml | grep -E "construction|Device scene"; done
