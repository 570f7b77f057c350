#!/usr/bin/env python3
"""Would two passes in flight at once pay? (GPU box.) The bench frame rendered by ONE scene, against the same frame split
in two tile partitions rendered at the same time by two scenes on the same GPU (each has its own stream and path
buffers), so that one partition's streaming kernels can run beside the other's traversal kernels."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi  # noqa: E402
from tools import gen_scenes  # noqa: E402

d = "/tmp/rt_probe_scene"
os.makedirs(d, exist_ok=True)
if not os.path.exists(d + "/readme.sdml"):
    gen_scenes.scene_readme(d, "readme", size=1080, spp=8, grid=(512, 512), texsize=512)
hs = capi.HostScene.load(d + "/readme.sdml").build()
o = hs.desc.contents.options
spp = int(os.environ.get("PROBE_SPP", "256"))
opt = capi.RtOptions(o.width, o.height, o.bias, o.exposure, o.maxBounces, spp, o.fisheye, o.focus, o.lens)
parts = int(os.environ.get("PROBE_PARTS", "2"))
share = int(os.environ.get("PROBE_SHARE", "1"))  # the work is 1/share of the frame (one rank's tiles of `share` GPUs), split in `parts`
one = capi.device_scene(hs.desc)
one.set_view(options=opt)
many = [capi.device_scene(hs.desc) for _ in range(parts)]
for s in many:
    s.set_view(options=opt)


def timed(fn, reps=3):
    fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
    return best * 1e3


def together():
    th = [threading.Thread(target=lambda k=k: many[k].render(download=False, part_index=k * share, part_count=parts * share)) for k in range(parts)]
    for t in th: t.start()
    for t in th: t.join()


def in_turn():
    for k in range(parts):
        many[k].render(download=False, part_index=k * share, part_count=parts * share)


print(f"{spp} spp, 1/{share} of the frame: one scene {timed(lambda: one.render(download=False, part_index=0, part_count=share)):.1f} ms; {parts} partitions in turn {timed(in_turn):.1f} ms; "
      f"{parts} partitions at the same time {timed(together):.1f} ms")
