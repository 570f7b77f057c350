#!/usr/bin/env python3
"""The 10 M-triangle stress configuration on one GPU: a procedural ~10 M-triangle mesh (built in memory through the host C API, no OBJ text),
3840x2160, multi-bounce GI with a point and an environment light.

    python tools/config5_stress.py [--tris-side 2237] [--width 3840 --height 2160] [--spp 4] [--check-pixels 200000]

Prints one JSON line: build times, Mrays/s, per-kernel milliseconds, and the parity checks it ran against the CPU
restatement on the same scene description (primary hits of a pixel band and incoherent queries, bit-exact)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi  # noqa: E402
from tools import gen_scenes  # noqa: E402


def build_scene(side, width, height, spp, bounces):
    hs = capi.HostScene.new(width, height)
    plastic = hs.add_material(capi.RT_MAT_PLASTIC, Kd=0.8, Ks=0.2, eta=1.5, Kr=0.2, roughness=0.2)
    # several displaced spheres (incoherent secondary rays between them) totalling ~2*side*side triangles
    parts = [(1.00, (0.0, 0.0, -4.0), 2.6, (7, 5, 31, 3)), (0.35, (-2.6, 1.2, -3.2), 1.3, (5, 9, 23, 7)),
             (0.35, (2.5, -1.0, -3.0), 1.2, (3, 4, 11, 2)), (0.30, (2.2, 1.6, -5.0), 1.4, (9, 6, 17, 5))]
    norm = sum(p[0] for p in parts)
    tris = 0
    for frac, center, scale, bumps in parts:
        n = max(8, int(side * (frac / norm) ** 0.5))
        v, nn, f = gen_scenes.blob_mesh(n, n, bumps=bumps, amp=(0.12, 0.04))
        hs.add_mesh(v, nn, f, center=center, scale=scale, color=(0.8, 0.7, 0.6, 1), material=plastic)
        tris += f.shape[0]
    hs.add_plane((0, 1, 0), 3.0, color=(0.6, 0.6, 0.6, 1))
    hs.add_light(capi.RT_LIGHT_POINT, color=(8, 8, 8, 1), v=(2, 3, 1))
    hs.add_light(capi.RT_LIGHT_ENVIRONMENT, color=(0.1, 0.1, 0.12, 1), radius=20.0)
    hs.set_camera()
    hs.set_options(width, height, maxBounces=bounces, numRays=spp)
    return hs, tris


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris-side", type=int, default=2237, help="total triangles ~ 2*side*side (2237 -> ~10 M)")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--check-rows", type=int, default=64, help="rows of primary hits compared with the CPU restatement")
    ap.add_argument("--check-rays", type=int, default=100000)
    args = ap.parse_args()

    t0 = time.time()
    hs, tris = build_scene(args.tris_side, args.width, args.height, args.spp, args.bounces)
    t_mesh = time.time() - t0
    t0 = time.time(); hs.build(); t_bvh = time.time() - t0
    d = hs.desc.contents
    t0 = time.time(); dev = capi.device_scene(hs.desc); t_upload = time.time() - t0
    dev.render(download=False)
    _, st = dev.render(download=False)
    rays = st.raysClosest + st.raysAny
    out = {"triangles": tris, "bvh_nodes": int(d.numNodes), "width": args.width, "height": args.height, "spp": args.spp,
           "maxBounces": args.bounces, "seconds": {"mesh": round(t_mesh, 1), "bvh_build_host": round(t_bvh, 1), "upload": round(t_upload, 2)},
           "ms_frame": round(st.msTotal, 2), "mrays_per_s": round(rays / st.msTotal / 1e3, 1), "rays": int(rays),
           "rays_skipped": int(st.raysSkipped),
           "kernel_ms": {k: round(getattr(st, k), 2) for k in ("msRaygen", "msExtend", "msShade", "msShadow", "msDirect", "msAccumulate")}}
    if args.check_rows > 0:
        from tests.oracle_binding import oracle_scene
        orc = oracle_scene(hs.desc, threads=min(16, len(os.sched_getaffinity(0))))
        y0 = args.height // 2 - args.check_rows // 2
        gi, gt = dev.trace_primary(0, y0, args.width, args.check_rows)
        oi, ot = orc.trace_primary(0, y0, args.width, args.check_rows)
        rng = np.random.default_rng(3)
        o = (rng.normal(size=(args.check_rays, 3)) * 3 + np.array([0, 0, -4])).astype(np.float32)
        dr = rng.normal(size=(args.check_rays, 3)).astype(np.float32)
        g = dev.trace_closest(o, dr); r = orc.trace_closest(o, dr)
        out["parity"] = {
            "primary_pixels": int(gi.size), "primary_ids_equal": bool(np.array_equal(gi, oi)),
            "primary_t_bits_equal": bool(np.array_equal(gt.view(np.uint32), ot.view(np.uint32))),
            "primary_hits": int((gi >= 0).sum()),
            "random_rays": args.check_rays,
            "random_equal": bool(all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(g, r))),
            "random_any_equal": bool(np.array_equal(dev.trace_any(o, dr), orc.trace_any(o, dr)))}
        orc.close()
    # the same scene with the tree built on the GPU (RtSceneDesc.nodes = NULL): build time, frame time, t bits of the primary hits
    t0 = time.time()
    lb = capi.device_scene(hs.desc, build_on_device=True)
    t_create = time.time() - t0
    lb.render(download=False)
    _, sl = lb.render(download=False)
    y0 = args.height // 2 - max(1, args.check_rows) // 2
    gi, gt = dev.trace_primary(0, y0, args.width, max(1, args.check_rows))
    li, lt = lb.trace_primary(0, y0, args.width, max(1, args.check_rows))
    out["device_bvh"] = {"build_ms": round(lb.bvh_build_ms(), 2), "scene_create_s": round(t_create, 2), "ms_frame": round(sl.msTotal, 2),
                         "primary_t_bits_equal_to_host_tree": bool(np.array_equal(gt.view(np.uint32), lt.view(np.uint32))),
                         "primary_ids_differing": int((gi != li).sum())}
    lb.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
