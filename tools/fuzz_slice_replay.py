#!/usr/bin/env python3
"""GPU box: one scene of the fixed fuzz slice of the test suite (tests/test_gpu_fuzz_slice.py: seed of the stream, scene
number) again, in detail: the frame from an ordinary launch, from a counting launch (which never splits visibility
queries) and from the CPU restatement, and where they differ.

    python tools/fuzz_slice_replay.py 101 572"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi  # noqa: E402
from tests.oracle_binding import oracle_scene  # noqa: E402
from tools import fuzz_parity  # noqa: E402

seed, want = int(sys.argv[1]), int(sys.argv[2])
rays_per_scene = int(sys.argv[3]) if len(sys.argv) > 3 else 6000
rng = np.random.default_rng(seed)
for k in range(want + 1):
    hs, rays_of, family = fuzz_parity.scene_of_the_stream(rng, k)
    o, d = rays_of(rng, rays_per_scene)
dev, orc = capi.device_scene(hs.desc), oracle_scene(hs.desc, threads=8)
dsc = hs.desc.contents
print(f"scene {want} of stream {seed}: family {family}, prims {dsc.numPrims} planes {dsc.numPlanes} lights {dsc.numLights}, "
      f"{dsc.options.width}x{dsc.options.height} spp {dsc.options.numRays} bounces {dsc.options.maxBounces}")
for i in range(dsc.numLights):
    l = dsc.lights[i]
    print(f"  light {i}: type {l.type} color {list(l.color)} v {list(l.v)}")
for i in range(dsc.numPlanes):
    p = dsc.planes[i]
    print(f"  plane {i}: normal {list(p.normal)} point {list(p.point)} material {p.material}")
img, st = dev.render(seed=want)
cnt, stc = dev.render(seed=want, collect_counters=True)
ref, ost = orc.render(seed=want)
print("ordinary launch == counting launch:", bool(np.array_equal(img.view(np.uint32), cnt.view(np.uint32))))
for name, a in (("ordinary", img), ("counting", cnt)):
    off = np.abs(a[..., :3] - ref[..., :3]).max(-1) > 1e-3
    print(f"{name} vs CPU: {int(off.sum())} pixels off by more than 1e-3, alpha equal {bool(np.array_equal(a[..., 3], ref[..., 3]))}; "
          f"rays gpu {st.raysClosest + st.raysAny + st.raysSkipped} cpu {ost.raysClosest + ost.raysAny}")
    ys, xs = np.nonzero(off)
    for y, x in list(zip(ys, xs))[:8]:
        print(f"   pixel ({x},{y}): gpu {a[y, x].tolist()} cpu {ref[y, x].tolist()}")

# ---- the same scene with one sample through every pixel centre and direct light only: which point-light queries differ?
f32 = np.float32
o_ = dsc.options
opt = capi.RtOptions(o_.width, o_.height, o_.bias, o_.exposure, 1, 1, o_.fisheye, o_.focus, o_.lens)
dev.set_view(options=opt); orc.set_view(options=opt)
img1, _ = dev.render(seed=want)
ref1, _ = orc.render(seed=want)
off = np.abs(img1[..., :3] - ref1[..., :3]).max(-1) > 1e-3
print(f"one centre sample, one bounce: {int(off.sum())} pixels differ")
if o_.fisheye == 0 and o_.focus <= 0 and off.any():
    cam = dsc.camera
    eye, fwd, right, up = (np.array(list(v), f32) for v in (cam.eye, cam.forward, cam.right, cam.up))
    W, H = o_.width, o_.height
    ys, xs = np.nonzero(off)
    sx = ((f32(2) * xs.astype(f32) - f32(W)) / f32(max(W, H))).astype(f32)
    sy = ((f32(H) - f32(2) * ys.astype(f32)) / f32(max(W, H))).astype(f32)
    d0 = (fwd[None, :] + sx[:, None] * right[None, :] + sy[:, None] * up[None, :]).astype(f32)
    o0 = np.repeat(eye[None, :], len(xs), 0)
    ids, t, pt, nm = dev.trace_closest(o0, d0)
    ids_c, t_c, pt_c, nm_c = orc.trace_closest(o0, d0)
    print("   primary hits equal on both:", bool(np.array_equal(ids, ids_c) and np.array_equal(pt.view(np.uint32), pt_c.view(np.uint32))))
    nd0 = d0 / np.sqrt((d0 * d0).sum(1, dtype=f32))[:, None]
    facing = (-(nd0 * nm).sum(1)) < 0
    outn = np.where(facing[:, None], -nm, nm).astype(f32)
    p1 = (pt + f32(o_.bias) * outn).astype(f32)
    for li in range(dsc.numLights):
        l = dsc.lights[li]
        if l.type != capi.RT_LIGHT_POINT:
            continue
        ld = (np.array(list(l.v), f32)[None, :] - p1).astype(f32)
        D = np.sqrt((ld * ld).sum(1, dtype=f32))
        hi, ht, hp, _ = dev.trace_closest(p1, ld)
        ci, ct, cp, _ = orc.trace_closest(p1, ld)
        same = np.array_equal(hi, ci) and np.array_equal(hp.view(np.uint32), cp.view(np.uint32))
        dist = np.sqrt(((p1 - hp) ** 2).sum(1, dtype=f32))
        shadow = (hi >= 0) & (dist < D)
        for j in range(min(10, len(xs))):
            print(f"   pixel ({xs[j]},{ys[j]}): gpu {img1[ys[j], xs[j], :3].tolist()} cpu {ref1[ys[j], xs[j], :3].tolist()} | primary hit {ids[j]} at t {t[j]:.6g} "
                  f"point {pt[j].tolist()} | light {li}: closest {hi[j]} t {ht[j]:.6g} |point - hit| {dist[j]:.6g} bulb {D[j]:.6g} -> in shadow {bool(shadow[j])} "
                  f"(closest-hit queries equal on both: {same})")
