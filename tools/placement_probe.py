#!/usr/bin/env python3
"""GPU box: does the physical placement of the path-state arena change the streaming kernels? One process creates the
device scene several times in a row (optionally holding a dummy allocation while the arena is made) and times a frame."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi  # noqa: E402
from tools import gen_scenes  # noqa: E402

d = "/tmp/rt_probe_scene"
os.makedirs(d, exist_ok=True)
if not os.path.exists(d + "/readme.sdml"):
    gen_scenes.scene_readme(d, "readme", size=1080, spp=8, grid=(512, 512), texsize=512)
hs = capi.HostScene.load(d + "/readme.sdml").build()
o = hs.desc.contents.options
hip = ctypes.CDLL("libamdhip64.so")
for trial in range(6):
    dummy = ctypes.c_void_p()
    hold = int(os.environ.get("HOLD_GB", "0")) if trial % 2 == 1 else 0
    if hold:
        assert hip.hipMalloc(ctypes.byref(dummy), ctypes.c_size_t(hold << 30)) == 0
    dev = capi.device_scene(hs.desc)
    dev.set_view(options=capi.RtOptions(o.width, o.height, o.bias, o.exposure, 4, 256, o.fisheye, o.focus, o.lens))
    dev.render(download=False)
    if hold:
        hip.hipFree(dummy)
    _, st = dev.render(download=False)
    print(f"trial {trial} hold {hold} GiB: shade {st.msShade:.2f} direct {st.msDirect:.2f} extend {st.msExtend:.2f} shadow {st.msShadow:.2f} total {st.msTotal:.2f}", flush=True)
    dev.close()
