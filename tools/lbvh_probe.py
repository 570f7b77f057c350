#!/usr/bin/env python3
"""Device-built BVH against the host-built (reference) tree on the GPU box: primary hits, random rays, frame, timings."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi  # noqa: E402
from tools import gen_scenes  # noqa: E402


def rays(n, seed):
    rng = np.random.default_rng(seed)
    o = (rng.normal(size=(n, 3)) * np.array([4, 2.4, 4]) + np.array([0, 0.3, -3])).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d *= (1.0 + np.abs(rng.normal(size=(n, 1)))).astype(np.float32)  # |d| >= 1
    return o, d.astype(np.float32)


d = "/tmp/rt_lbvh_scenes"
os.makedirs(d, exist_ok=True)
for name in sys.argv[1:] or ["mesh1k", "materials", "mesh70k", "readme"]:
    ext = "txt" if name.startswith("spheres_txt") else "sdml"
    if not os.path.exists(f"{d}/{name}.{ext}"):
        gen_scenes.SCENES[name](d)
    t0 = time.time()
    hs = capi.HostScene.load(f"{d}/{name}.{ext}").build()
    t_host = time.time() - t0
    ref = capi.device_scene(hs.desc)
    t0 = time.time()
    lb = capi.device_scene(hs.desc, build_on_device=True)
    t_create = time.time() - t0
    ids0, t0_ = ref.trace_primary()
    ids1, t1_ = lb.trace_primary()
    tb = t0_.view(np.uint32) != t1_.view(np.uint32)
    idd = ids0 != ids1
    o, dd = rays(200_000, 3)
    a = ref.trace_closest(o, dd); b = lb.trace_closest(o, dd)
    rt = a[1].view(np.uint32) != b[1].view(np.uint32)
    occ = ref.trace_any(o, dd) != lb.trace_any(o, dd)
    ref.render(download=False); lb.render(download=False)
    _, s0 = ref.render(download=False)
    _, s1 = lb.render(download=False)
    print(f"{name}: {hs.desc.contents.numPrims} objects | device build {lb.bvh_build_ms():.2f} ms (scene create {t_create:.2f} s; host parse+build {t_host:.2f} s) | "
          f"primary: t differs {int(tb.sum())}, id differs {int(idd.sum())} (with equal t {int((idd & ~tb).sum())}) of {tb.size} | random rays: t differs {int(rt.sum())}, "
          f"id differs {int((a[0] != b[0]).sum())}, any differs {int(occ.sum())} of {rt.size} | frame {s0.msTotal:.1f} ms (reference tree) vs {s1.msTotal:.1f} ms (device tree)", flush=True)
    ref.close(); lb.close()
