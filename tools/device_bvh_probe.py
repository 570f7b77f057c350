#!/usr/bin/env python3
"""GPU box: the bench scene (and optionally the 10 M-triangle scene) through the host-built SAH tree, the tree clustered on
the GPU and the linear BVH built on the GPU: build times, frame times at 64 spp, primary-hit t bits against the host tree."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi  # noqa: E402
from tools import gen_scenes  # noqa: E402

d = "/tmp/rt_probe_scene"
os.makedirs(d, exist_ok=True)
if not os.path.exists(d + "/readme.sdml"):
    gen_scenes.scene_readme(d, "readme", size=1080, spp=8, grid=(512, 512), texsize=512)
spp = int(os.environ.get("PROBE_SPP", "64"))


def frame_ms(dev, o):
    dev.set_view(options=capi.RtOptions(o.width, o.height, o.bias, o.exposure, o.maxBounces, spp, o.fisheye, o.focus, o.lens))
    dev.render(download=False)
    best = None
    for _ in range(3):
        _, st = dev.render(download=False)
        best = st if best is None or st.msTotal < best.msTotal else best
    _, c = dev.render(download=False, collect_counters=True)
    best.note = (f"steps per ray: extension {c.innerExtend / c.raysExtend:.2f} inner + {c.primExtend / c.raysExtend:.2f} objects, "
                 f"visibility {c.innerShadow / max(1, c.raysShadow):.2f} + {c.primShadow / max(1, c.raysShadow):.2f}")
    return best


hs = capi.HostScene.load(d + "/readme.sdml").build()
o = hs.desc.contents.options
host = capi.device_scene(hs.desc)
ids0, t0 = host.trace_primary()
st = frame_ms(host, o)
print(f"host SAH tree:        frame {st.msTotal:8.2f} ms (extend {st.msExtend:.2f}, shadow {st.msShadow:.2f}) {st.note}", flush=True)
raw = capi.HostScene.load(d + "/readme.sdml").flatten_without_bvh()
for kind in os.environ.get("PROBE_TREES", "ploc:2048,ploc:1,lbvh").split(","):
    os.environ["RT_DEVICE_BVH"] = kind.split(":")[0]
    if ":" in kind:
        os.environ["RT_PLOC_TOP"] = kind.split(":")[1]  # clusters left when the clustering hands over to the top-down top (1 = clustering to the root)
    dev = capi.device_scene(raw.desc)
    ids1, t1 = dev.trace_primary()
    same = bool(np.array_equal(t0.view(np.uint32), t1.view(np.uint32)))
    st = frame_ms(dev, o)
    print(f"device tree ({kind}): build {dev.bvh_build_ms():6.2f} ms, frame {st.msTotal:8.2f} ms (extend {st.msExtend:.2f}, shadow {st.msShadow:.2f}), "
          f"primary t bits equal: {same}, ids differing {int((ids0 != ids1).sum())} | {st.note}", flush=True)
    dev.close()
