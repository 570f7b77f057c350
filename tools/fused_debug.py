#!/usr/bin/env python3
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi
from tools import gen_scenes
with tempfile.TemporaryDirectory() as tmp:
    for name in ("mesh1k_gi", "readme_small", "materials"):
        gen_scenes.SCENES[name](tmp)
        hs = capi.HostScene.load(os.path.join(tmp, f"{name}.sdml")).build()
        dev = capi.device_scene(hs.desc)
        o = hs.desc.contents.options
        for mb in (1, 2, 3, 4):
            dev.set_view(options=capi.RtOptions(o.width, o.height, o.bias, o.exposure, mb, o.numRays, o.fisheye, o.focus, o.lens))
            os.environ["RT_NO_FUSED_TRACE"] = "1"
            a, sa = dev.render(seed=3, part_index=0, part_count=2, tile=(32, 32))
            del os.environ["RT_NO_FUSED_TRACE"]
            b, sb = dev.render(seed=3, part_index=0, part_count=2, tile=(32, 32))
            d = np.abs(a - b).max(-1)
            print(name, "maxBounces", mb, "pixels differing", int((d > 0).sum()), "max", float(d.max()), "rays", sa.raysClosest + sa.raysAny, sb.raysClosest + sb.raysAny, "extend", sa.raysExtend, sb.raysExtend, flush=True)
        dev.close()
