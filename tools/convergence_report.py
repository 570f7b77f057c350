#!/usr/bin/env python3
"""GPU box: convergence of the README-scene stand-in without denoising. Frames at 1, 4, 16, 64, 256 (and 1024) samples per
pixel against a 4096-spp frame of the same scene: PSNR / RMSE, frame time, Mrays/s. Independent seeds per level.

    python tools/convergence_report.py [--size 540] [--grid 256] [--ref-spp 4096]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi  # noqa: E402
from tools import gen_scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=540)
ap.add_argument("--grid", type=int, default=256)
ap.add_argument("--ref-spp", type=int, default=4096)
args = ap.parse_args()
d = f"/tmp/rt_conv_{args.size}_{args.grid}"
os.makedirs(d, exist_ok=True)
gen_scenes.scene_readme(d, "readme", size=args.size, spp=16, grid=(args.grid, args.grid), texsize=256)
hs = capi.HostScene.load(d + "/readme.sdml").build()
dev = capi.device_scene(hs.desc)
o = hs.desc.contents.options


def frame(spp, seed, **kw):
    dev.set_view(options=capi.RtOptions(o.width, o.height, o.bias, o.exposure, o.maxBounces, spp, o.fisheye, o.focus, o.lens))
    img, st = dev.render(seed=seed, **kw)
    return img[..., :3].astype(np.float64), st


ref, st = frame(args.ref_spp, 1000)
rows = []
for spp in (1, 4, 16, 64, 256, 1024):
    if spp >= args.ref_spp:
        break
    img, st = frame(spp, 7 + spp)
    mse = float(((img - ref) ** 2).mean())
    rows.append({"spp": spp, "rmse": round(mse ** 0.5, 5), "psnr_db": round(10 * np.log10(1.0 / mse), 2), "ms": round(st.msTotal, 2),
                 "mrays_per_s": round((st.raysClosest + st.raysAny) / st.msTotal / 1e3, 1)})
# adaptive sampling (up to 1024 samples per pixel, stop per pixel by the standard error of its luminance): what a given
# error costs in samples compared with the uniform levels above
adaptive = []
for thr in (0.1, 0.05, 0.02, 0.01):
    img, st = frame(min(1024, args.ref_spp // 2), 99, adaptive_threshold=thr, adaptive_batch=16)
    mse = float(((img - ref) ** 2).mean())
    adaptive.append({"threshold": thr, "mean_spp": round(st.cameraSamples / (o.width * o.height), 1), "rmse": round(mse ** 0.5, 5),
                     "psnr_db": round(10 * np.log10(1.0 / mse), 2), "ms": round(st.msTotal, 2), "passes": st.passes})
# progressive rendering: the estimate after 16, 64, 256 samples of ONE 256-sample frame built up by windows
prog = []
first = 0
for upto in (16, 64, 256):
    dev.set_view(options=capi.RtOptions(o.width, o.height, o.bias, o.exposure, o.maxBounces, 256, o.fisheye, o.focus, o.lens))
    img, st = dev.render(seed=5, first_sample=first, sample_count=upto - first, keep_accumulator=first > 0)
    mse = float(((img[..., :3].astype(np.float64) - ref) ** 2).mean())
    prog.append({"samples_so_far": upto, "rmse": round(mse ** 0.5, 5), "ms_this_window": round(st.msTotal, 2)})
    first = upto
print(json.dumps({"adaptive": adaptive, "progressive": prog}))
print(json.dumps({"scene": f"README stand-in {o.width}x{o.height}, {hs.desc.contents.numPrims} objects", "reference_spp": args.ref_spp,
                  "levels": rows}))
