#!/usr/bin/env python3
"""Lane-utilisation probe (GPU box): counters of the traversal kernels for maxBounces = 1..4 on the bench scene."""
import hashlib
import os
import sys
import time

os.environ.setdefault("RT_NO_FUSED_TRACE", "1")  # one kernel family per launch, so that the per-family columns mean what they say
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi  # noqa: E402
from tools import gen_scenes  # noqa: E402

d = "/tmp/rt_probe_scene"
os.makedirs(d, exist_ok=True)
if not os.path.exists(d + "/readme.sdml"):
    gen_scenes.scene_readme(d, "readme", size=1080, spp=8, grid=(512, 512), texsize=512)
hs = capi.HostScene.load(d + "/readme.sdml").build()
dev = capi.device_scene(hs.desc)
o = hs.desc.contents.options
for mb in [int(x) for x in os.environ.get('PROBE_BOUNCES', '1,2,4').split(',')]:
    dev.set_view(options=capi.RtOptions(o.width, o.height, o.bias, o.exposure, mb, int(os.environ.get("PROBE_SPP", "64")), o.fisheye, o.focus, o.lens))
    img, _ = dev.render()
    sha = hashlib.sha1(img.tobytes()).hexdigest()[:12]  # variants of the kernels must leave the frame as it is
    _, st = dev.render(download=False)
    _, c = dev.render(download=False, collect_counters=True)
    print(f"maxBounces={mb}: extend {st.raysExtend / st.msExtend / 1e3:.0f} Mrays/s ({st.msExtend:.2f} ms), shadow "
          f"{st.raysShadow / st.msShadow / 1e3:.0f} Mrays/s ({st.msShadow:.2f} ms), shade {st.msShade:.2f} ms, direct {st.msDirect:.2f} ms, resolve {st.msResolve:.2f} ms, total {st.msTotal:.2f} ms | lane use extend inner "
          f"{c.laneUseInnerExtend:.3f} prim {c.laneUsePrimExtend:.3f}, shadow inner {c.laneUseInnerShadow:.3f} prim {c.laneUsePrimShadow:.3f} | "
          f"inner/ray ext {c.innerExtend / c.raysExtend:.1f} sh {c.innerShadow / c.raysShadow:.1f}, prim/ray ext {c.primExtend / c.raysExtend:.1f} sh {c.primShadow / c.raysShadow:.1f} | frame {sha}"
          + (f" | continuing INNER steps {(c.planeTests - 3 * (c.raysExtend + c.raysShadow + c.raysPlaneAnswered)) / c.innerExpansions:.3f} of all" if os.environ.get("PROBE_CONT") else ""))
