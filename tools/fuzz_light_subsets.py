#!/usr/bin/env python3
"""GPU box: one scene of tools/fuzz_parity.py (--seed S --only K) rendered with each of its lights alone, HIP library
against the CPU restatement: which light a frame difference comes from."""
import argparse
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import capi  # noqa: E402
from tests.oracle_binding import oracle_scene  # noqa: E402

spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tools", "fuzz_parity.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, required=True)
ap.add_argument("--only", type=int, required=True)
args = ap.parse_args()

real_add_light = capi.HostScene.add_light
keep = None
calls = [0]


def add_light(self, *a, **kw):
    i = calls[0]; calls[0] += 1
    if keep is None or i in keep:
        real_add_light(self, *a, **kw)


capi.HostScene.add_light = add_light


def scene(subset):
    global keep
    rng = np.random.default_rng(args.seed)
    for k in range(args.only + 1):
        keep = subset if k == args.only else set()
        calls[0] = 0
        hs = fz.random_scene(rng, k)
        fz.random_rays(rng, 20000)
    return hs, calls[0]


hs, nl = scene(None)
print("lights in the scene:", nl, flush=True)
for subset in [None] + [{i} for i in range(nl)]:
    hs, _ = scene(subset)
    o = hs.desc.contents.options
    dev, orc = capi.device_scene(hs.desc), oracle_scene(hs.desc, threads=8)
    for mb in (1, o.maxBounces):
        opt = capi.RtOptions(o.width, o.height, o.bias, o.exposure, mb, o.numRays, o.fisheye, o.focus, o.lens)
        dev.set_view(options=opt); orc.set_view(options=opt)
        img, st = dev.render(seed=args.only); ref, ost = orc.render(seed=args.only)
        diff = np.abs(img[..., :3] - ref[..., :3]).max(-1)
        print(f"lights kept {subset}: maxBounces {mb}: {int((diff > 1e-6).sum())} pixels differ (max {diff.max():.3g}); skipped gpu {st.raysSkipped} plane-answered {st.raysPlaneAnswered}; "
              f"rays gpu {st.raysClosest + st.raysAny + st.raysSkipped} cpu {ost.raysClosest + ost.raysAny}", flush=True)
    dev.close(); orc.close()
