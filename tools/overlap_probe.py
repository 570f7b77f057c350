#!/usr/bin/env python3
"""GPU box: does the GPU gain from two render streams at once? Two scene handles (own stream, own path buffers) render
half of the samples each from two host threads; wall time against one handle rendering all of them."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi  # noqa: E402
from tools import gen_scenes  # noqa: E402

d = "/tmp/rt_probe_scene256"
os.makedirs(d, exist_ok=True)
if not os.path.exists(d + "/readme.sdml"):
    gen_scenes.scene_readme(d, "readme", size=1080, spp=256, grid=(512, 512), texsize=512)
hs = capi.HostScene.load(d + "/readme.sdml").build()
o = hs.desc.contents.options
devs = [capi.device_scene(hs.desc) for _ in range(2)]
full = capi.RtOptions(o.width, o.height, o.bias, o.exposure, o.maxBounces, 256, o.fisheye, o.focus, o.lens)
half = capi.RtOptions(o.width, o.height, o.bias, o.exposure, o.maxBounces, 128, o.fisheye, o.focus, o.lens)
devs[0].set_view(options=full)
devs[0].render(download=False)
t = time.perf_counter(); devs[0].render(download=False); one = (time.perf_counter() - t) * 1e3
for dv in devs:
    dv.set_view(options=half)
    dv.render(download=False)


def work(dv):
    dv.render(download=False)


best = 1e9
for _ in range(3):
    th = [threading.Thread(target=work, args=(dv,)) for dv in devs]
    t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    best = min(best, (time.perf_counter() - t) * 1e3)
t = time.perf_counter(); devs[0].render(download=False); devs[1].render(download=False); seq = (time.perf_counter() - t) * 1e3
print(f"one stream, 256 spp: {one:.1f} ms | two streams x 128 spp concurrently: {best:.1f} ms | the same two one after the other: {seq:.1f} ms")
