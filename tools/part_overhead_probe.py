#!/usr/bin/env python3
"""GPU box: wall time of rt_render for one rank's share of the frame (1/N of the tiles) against the device time the
library reports, i.e. the host-side overhead per call that limits strong scaling."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi  # noqa: E402
from tools import gen_scenes  # noqa: E402

d = "/tmp/rt_probe_scene256"
os.makedirs(d, exist_ok=True)
if not os.path.exists(d + "/readme.sdml"):
    gen_scenes.scene_readme(d, "readme", size=1080, spp=256, grid=(512, 512), texsize=512)
hs = capi.HostScene.load(d + "/readme.sdml").build()
dev = capi.device_scene(hs.desc)
for n in (1, 2, 4, 8):
    dev.render(part_index=0, part_count=n, download=False)
    best = None
    for _ in range(2):
        t = time.perf_counter()
        _, st = dev.render(part_index=0, part_count=n, download=False)
        wall = (time.perf_counter() - t) * 1e3
        if best is None or wall < best[0]:
            best = (wall, st)
    wall, st = best
    rays = st.raysClosest + st.raysAny
    print(f"1/{n} of the frame: wall {wall:8.2f} ms, device {st.msTotal:8.2f} ms, host overhead {wall - st.msTotal:6.2f} ms, "
          f"{rays / wall / 1e3:7.1f} Mrays/s per GPU, x{n} = {n * rays / wall / 1e3:8.1f} | raygen {st.msRaygen:.2f} extend {st.msExtend:.2f} "
          f"shade {st.msShade:.2f} shadow {st.msShadow:.2f} direct {st.msDirect:.2f} accumulate {st.msAccumulate:.2f} ms", flush=True)
