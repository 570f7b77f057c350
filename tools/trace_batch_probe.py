#!/usr/bin/env python3
"""GPU box: rt_trace_closest / rt_trace_any on a batch of 4 M incoherent rays in pageable host memory (bench scene): wall
time of the call against the kernel's own time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi
from tools import gen_scenes
d = "/tmp/rt_probe_scene"
os.makedirs(d, exist_ok=True)
if not os.path.exists(d + "/readme.sdml"):
    gen_scenes.scene_readme(d, "readme", size=1080, spp=8, grid=(512, 512), texsize=512)
hs = capi.HostScene.load(d + "/readme.sdml").build()
dev = capi.device_scene(hs.desc)
rng = np.random.default_rng(1)
n = 4_000_000
o = (rng.normal(size=(n, 3)) * np.array([4, 2.4, 4]) + np.array([0, 0.3, -3])).astype(np.float32)
dd = rng.normal(size=(n, 3)).astype(np.float32)
for name, fn in (("rt_trace_closest", dev.trace_closest), ("rt_trace_any", dev.trace_any)):
    fn(o, dd)
    t = time.perf_counter(); r = fn(o, dd); wall = (time.perf_counter() - t) * 1e3
    print(f"{name}: {n} rays, wall {wall:.2f} ms, kernel {dev.last_trace_kernel_ms():.2f} ms")
