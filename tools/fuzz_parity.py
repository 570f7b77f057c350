#!/usr/bin/env python3
"""GPU box: randomised parity. Random scenes (triangle soups with degenerate and huge triangles, spheres, planes, point /
distant / environment lights, random materials; meshes on coarse grids around near bulbs; clusters in front of a plane
millions of units away), random cameras, and rays with extreme directions (zero components, tiny and huge lengths,
origins inside boxes and far away): closest and any-hit queries must equal the CPU restatement bit for bit; rendered
frames must have the same alpha and at most 1 % of their pixels off by more than 1e-3.

    python tools/fuzz_parity.py [--seconds 120] [--seed 1]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi  # noqa: E402
from tests.oracle_binding import oracle_scene  # noqa: E402


def random_scene(rng, k):
    w, h = int(rng.integers(8, 97)), int(rng.integers(8, 97))
    hs = capi.HostScene.new(w, h)
    mats = [hs.add_material(capi.RT_MAT_DEFAULT)]
    for t in (capi.RT_MAT_GLASS, capi.RT_MAT_MIRROR, capi.RT_MAT_PLASTIC, capi.RT_MAT_METAL):
        mats.append(hs.add_material(t, Kd=float(rng.uniform(0.2, 1)), Ks=float(rng.uniform(0, 1)), eta=float(rng.uniform(1.1, 2.5)),
                                    Kr=float(rng.uniform(0.2, 1)), Kt=float(rng.uniform(0.2, 1)), Ka=float(rng.uniform(0.5, 4)),
                                    roughness=float(rng.uniform(0, 0.5))))
    # smooth textures (a last-bit difference in the texture coordinate then moves the colour by little) on some scenes
    texs = []
    if k % 5 == 0:
        for _ in range(2):
            th, tw = int(rng.integers(2, 40)), int(rng.integers(2, 40))
            yy, xx = np.mgrid[0:th, 0:tw]
            img = np.stack([0.5 + 0.4 * np.sin(xx / tw * 6 + rng.uniform(0, 6)), 0.5 + 0.4 * np.cos(yy / th * 5), 0.3 + 0.2 * np.sin((xx + yy) / 9.0),
                            np.full(xx.shape, float(rng.choice([1.0, 1.0, 0.5, 2.0])))], -1)
            texs.append(hs.add_texture(img))
    pick_tex = lambda: int(rng.choice(texs)) if texs and rng.random() < 0.6 else -1
    ntri = int(rng.choice([0, 1, 3, 50, 700, 5000]))
    if ntri:
        nv = max(3, ntri // 2)
        scale = float(rng.choice([0.05, 1.0, 30.0]))
        v = rng.normal(size=(nv, 3)) * scale
        if k % 3 == 0:
            v = np.round(v * 4) / 4          # many coplanar / collinear points, exact ties in t
        f = rng.integers(0, nv, size=(ntri, 3))
        f[::11, 1] = f[::11, 0]              # degenerate triangles
        hs.add_mesh(v, rng.normal(size=(nv, 3)), f, center=tuple(rng.normal(size=3) * 0.5 + np.array([0, 0, -3])),
                    scale=float(rng.uniform(0.5, 4)), color=tuple(rng.uniform(0, 1, size=3)) + (1.0,), material=int(rng.choice(mats)))
    for _ in range(int(rng.choice([0, 1, 5, 40]))):
        hs.add_sphere(tuple(rng.normal(size=3) * 1.5 + np.array([0, 0, -3])), float(rng.uniform(0.05, 1.2)),
                      color=tuple(rng.uniform(0, 1, size=3)) + (1.0,), material=int(rng.choice(mats)), texture=pick_tex())
    for _ in range(int(rng.choice([0, 1, 2]))):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        hs.add_plane(tuple(n), float(rng.uniform(1, 4)), color=tuple(rng.uniform(0, 1, size=3)) + (1.0,), material=int(rng.choice(mats)),
                     tex_top_left=tuple(rng.normal(size=3)), tex_zoom=float(rng.uniform(1, 30)), tex_shift=tuple(rng.normal(size=3)),
                     texture=pick_tex())
    kinds = rng.choice([capi.RT_LIGHT_POINT, capi.RT_LIGHT_DISTANT, capi.RT_LIGHT_ENVIRONMENT], size=int(rng.integers(0, 4)))
    for t in kinds:
        la = float(rng.choice([1.0, 1.0, 0.25, 3.0]))  # alpha of the light's colour: it reaches the pixel's alpha through max()
        if t == capi.RT_LIGHT_POINT:  # some bulbs closer than 1 unit to the geometry (|d| < 1 visibility rays)
            hs.add_light(t, color=tuple(rng.uniform(0, 6, size=3)) + (la,), v=tuple(rng.normal(size=3) * 0.7 + np.array([0, 0, -3])))
        elif t == capi.RT_LIGHT_DISTANT:
            hs.add_light(t, color=tuple(rng.uniform(0, 1, size=3)) + (la,), v=tuple(rng.normal(size=3)))
        else:
            hs.add_light(t, color=tuple(rng.uniform(0, 0.5, size=3)) + (la,), radius=float(rng.uniform(5, 20)), texture=pick_tex())
    fwd = rng.normal(size=3) * 0.2 + np.array([0, 0, -1])
    hs.set_camera(eye=tuple(rng.normal(size=3) * 0.3), forward=tuple(fwd), up=(0, 1, 0))
    mode = k % 4
    hs.set_options(w, h, maxBounces=int(rng.integers(1, 6)), numRays=int(rng.integers(1, 5)),
                   fisheye=1 if mode == 1 else 0, focus=2.5 if mode == 2 else -1.0, lens=0.2 if mode == 2 else 0.0,
                   exposure=1.5 if mode == 3 else -1.0)
    return hs.build()


def random_rays(rng, n):
    o = rng.normal(size=(n, 3)) * np.array([3, 2, 3]) + np.array([0, 0, -3])
    o[::7] *= 40
    d = rng.normal(size=(n, 3)) * np.exp(rng.normal(size=(n, 1)) * 3)
    for c in range(3):
        d[c::13, c] = 0.0
    d[5::17] = np.round(d[5::17])          # axis-aligned and diagonal directions, zero vectors
    return o.astype(np.float32), d.astype(np.float32)


def flat_box_scene(rng, k):
    """Meshes whose vertices lie on a quarter-unit grid, most triangles flat in an axis-aligned plane (their leaves have
    boxes of zero thickness), around point lights that sit a hair off a grid plane and closer than one unit to the
    geometry: visibility rays from a flat triangle toward such a bulb run almost parallel to the flat boxes of its
    neighbours, where the slab test's entry distance is worst conditioned (a huge reciprocal times a rounded difference),
    and their un-normalised direction is shorter than 1, where the reference's box culling is over-aggressive
    (src/acceleration/BVH.cpp:302-305 compares a distance in units of |d| with t in units of 1)."""
    w, h = int(rng.integers(12, 65)), int(rng.integers(12, 65))
    hs = capi.HostScene.new(w, h)
    mats = [hs.add_material(capi.RT_MAT_DEFAULT),
            hs.add_material(capi.RT_MAT_PLASTIC, Kd=0.7, Kr=0.3, eta=1.5, roughness=0.2),
            hs.add_material(capi.RT_MAT_MIRROR, Kr=0.9, eta=1.5)]
    ntri = int(rng.choice([40, 300, 1500, 4000]))
    cell = float(rng.choice([0.25, 0.25, 0.5, 0.125]))
    ext = int(rng.choice([3, 6, 10]))
    base = rng.integers(-ext, ext + 1, size=(ntri, 1, 3)).astype(float)
    tri = base + rng.integers(-2, 3, size=(ntri, 3, 3))
    flat_axis = rng.integers(0, 3, size=ntri)
    is_flat = rng.random(ntri) < 0.8
    for a in range(3):  # flat in one axis: the three vertices share that coordinate
        sel = is_flat & (flat_axis == a)
        tri[sel, :, a] = base[sel, 0, a][:, None]
    tri *= cell
    v = tri.reshape(-1, 3)
    f = np.arange(ntri * 3).reshape(ntri, 3)
    center = np.array([0.0, 0.0, -3.0])
    hs.add_mesh(v, rng.normal(size=v.shape), f, center=tuple(center), scale=float(rng.choice([1.0, 1.0, 2.0])),
                color=tuple(rng.uniform(0.2, 1, size=3)) + (1.0,), material=int(rng.choice(mats)))
    for _ in range(int(rng.choice([0, 0, 2, 8]))):
        hs.add_sphere(tuple(rng.normal(size=3) + center), float(rng.uniform(0.05, 0.6)),
                      color=tuple(rng.uniform(0, 1, size=3)) + (1.0,), material=int(rng.choice(mats)))
    if rng.random() < 0.4:
        n = np.zeros(3); n[int(rng.integers(0, 3))] = float(rng.choice([-1.0, 1.0]))
        hs.add_plane(tuple(n), float(rng.uniform(1, 4)), color=tuple(rng.uniform(0, 1, size=3)) + (1.0,), material=mats[0])
    for _ in range(int(rng.integers(1, 4))):
        # a bulb inside the mesh, a hair off a grid plane in one or two axes
        g = rng.integers(-ext, ext + 1, size=3).astype(float) * cell * 0.5
        off = rng.choice([0.0, 1e-6, 1e-4, 3e-3, 0.02], size=3) * rng.choice([-1.0, 1.0], size=3)
        hs.add_light(capi.RT_LIGHT_POINT, color=tuple(rng.uniform(1, 6, size=3)) + (1.0,), v=tuple(g + off + center))
    if rng.random() < 0.3:
        hs.add_light(capi.RT_LIGHT_DISTANT, color=tuple(rng.uniform(0, 1, size=3)) + (1.0,), v=tuple(np.round(rng.normal(size=3) * 2) / 2 + [0, 1e-3, 0]))
    hs.set_camera(eye=tuple(rng.normal(size=3) * 0.3), forward=tuple(rng.normal(size=3) * 0.2 + np.array([0, 0, -1])), up=(0, 1, 0))
    hs.set_options(w, h, maxBounces=int(rng.integers(1, 5)), numRays=int(rng.integers(1, 4)))
    return hs.build()


def far_plane_scene(rng, k):
    """A cluster of spheres and triangles, a point light beside it, and a plane 1e4 .. 2e8 units behind it. Every path that
    reaches the plane asks for the bulb from there: the origin's ulp is up to several units, so the slab test of every box
    near the bulb is decided by rounding and the query wanders through much of the tree, over a distance (the bulb's)
    that dwarfs the root box. Those are the queries the library splits over idle lanes at the end of a launch; their
    answers must still be the reference loop's. The bulb is bright enough to light the plane from that far; distant lights
    add any-hit queries from the same points."""
    w, h = int(rng.integers(12, 65)), int(rng.integers(12, 65))
    hs = capi.HostScene.new(w, h)
    mats = [hs.add_material(capi.RT_MAT_DEFAULT),
            hs.add_material(capi.RT_MAT_PLASTIC, Kd=0.7, Kr=0.3, eta=1.5, roughness=0.2),
            hs.add_material(capi.RT_MAT_MIRROR, Kr=0.9, eta=1.5)]
    center = np.array([0.0, 0.0, -4.0])
    ntri = int(rng.choice([0, 60, 800, 3000]))
    if ntri:
        nv = max(3, ntri // 2)
        hs.add_mesh(rng.normal(size=(nv, 3)) * 0.6, rng.normal(size=(nv, 3)), rng.integers(0, nv, size=(ntri, 3)), center=tuple(center),
                    scale=float(rng.uniform(0.5, 2)), color=tuple(rng.uniform(0.2, 1, size=3)) + (1.0,), material=int(rng.choice(mats)))
    for _ in range(int(rng.choice([3, 30, 300]))):
        hs.add_sphere(tuple(rng.normal(size=3) * np.array([1.5, 1.0, 1.0]) + center), float(rng.uniform(0.03, 0.3)),
                      color=tuple(rng.uniform(0, 1, size=3)) + (1.0,), material=int(rng.choice(mats)))
    dist = float(10.0 ** rng.uniform(4, 8.3))
    n = np.array([0.0, 0.0, 1.0]) + rng.normal(size=3) * float(rng.choice([0.0, 0.05, 0.3]))
    n /= np.linalg.norm(n)
    hs.add_plane(tuple(n), dist, color=tuple(rng.uniform(0.3, 1, size=3)) + (1.0,), material=int(rng.choice(mats[:2])))
    if rng.random() < 0.3:  # a floor under the cluster as well
        hs.add_plane((0.0, 1.0, 0.0), float(rng.uniform(1.5, 4)), color=tuple(rng.uniform(0, 1, size=3)) + (1.0,), material=mats[0])
    for _ in range(int(rng.integers(1, 3))):
        hs.add_light(capi.RT_LIGHT_POINT, color=tuple(rng.uniform(0.05, 0.6, size=3) * dist * dist) + (1.0,),
                     v=tuple(rng.normal(size=3) * np.array([1.5, 1.5, 1.0]) + center + np.array([0, 0, 1.0])))
    if rng.random() < 0.5:
        hs.add_light(capi.RT_LIGHT_DISTANT, color=tuple(rng.uniform(0, 1, size=3)) + (1.0,), v=tuple(rng.normal(size=3) + np.array([0, 0, 1.0])))
    hs.set_camera(eye=tuple(rng.normal(size=3) * 0.3), forward=tuple(rng.normal(size=3) * 0.15 + np.array([0, 0, -1])), up=(0, 1, 0))
    # Direct light only, and no environment light: a direction drawn with sinf / cosf differs in its last bit between the
    # device's and glibc's functions, and from millions of units away a last bit is a different hit back at the cluster;
    # with a bulb this bright one such path turns a pixel from black to white. Without drawn directions no transcendental
    # function is involved and the frames are expected to agree everywhere.
    hs.set_options(w, h, maxBounces=1, numRays=int(rng.integers(1, 4)))
    return hs.build()


def scene_of_the_stream(rng, k):
    """Scene k of a campaign and the rays traced through it: three in five are random soups, one in five comes from the
    flat-box family, one in five from the far-plane family."""
    if k % 5 == 4:
        return flat_box_scene(rng, k), near_parallel_rays, "flat"
    if k % 5 == 2:
        return far_plane_scene(rng, k), random_rays, "far"
    return random_scene(rng, k), random_rays, "soup"


def near_parallel_rays(rng, n, cell=0.25):
    """Rays that start on (or a hair off) grid planes and run almost parallel to them, many of them shorter than 1."""
    o = np.round((rng.normal(size=(n, 3)) * 1.5) / cell) * cell + np.array([0, 0, -3])
    o += rng.choice([0.0, 1e-4, -1e-4, 1e-6], size=(n, 3))
    d = rng.normal(size=(n, 3))
    ax = rng.integers(0, 3, size=n)
    d[np.arange(n), ax] *= rng.choice([0.0, 1e-7, 1e-5, 1e-3], size=n)
    d *= np.exp(rng.normal(size=(n, 1)) * 1.5 - 1.0)
    return o.astype(np.float32), d.astype(np.float32)


def same_bits(a, b):
    """Bit-equal, except that any NaN equals any NaN: x86 and gfx950 give NaNs of different sign and payload (a
    zero-length shading normal divided by its length, for instance)."""
    if a.dtype.kind != "f":
        return np.array_equal(a, b)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def detail(hs, dev, orc, o, d):
    dsc = hs.desc.contents
    print(f"scene: prims {dsc.numPrims} nodes {dsc.numNodes} planes {dsc.numPlanes}")
    names = ("id", "t", "point", "normal")
    g, r = dev.trace_closest(o, d), orc.trace_closest(o, d)
    for nm, a, b in zip(names, g, r):
        neq = (a.view(np.uint32) != b.view(np.uint32)) & ~((a != a) & (b != b)) if a.dtype.kind == 'f' else a != b
        rows = np.unique(np.nonzero(neq)[0])
        print(f"closest {nm}: {rows.size} rays differ")
        for i in rows[:5]:
            print(f"   ray {i}: o {o[i].tolist()} d {d[i].tolist()} gpu id {g[0][i]} t {g[1][i]!r} cpu id {r[0][i]} t {r[1][i]!r} "
                  f"gpu {nm} {a[i].tolist() if a.ndim > 1 else a[i]} cpu {b[i].tolist() if b.ndim > 1 else b[i]}")
    ga, ra = dev.trace_any(o, d), orc.trace_any(o, d)
    rows = np.nonzero(ga != ra)[0]
    print(f"any: {rows.size} rays differ")
    for i in rows[:5]:
        print(f"   ray {i}: o {o[i].tolist()} d {d[i].tolist()} gpu {ga[i]} cpu {ra[i]} closest gpu id {g[0][i]} t {g[1][i]!r}")
    gi, gt = dev.trace_primary(); oi, ot = orc.trace_primary()
    print(f"primary: {(gi != oi).sum()} ids, {(gt.view(np.uint32) != ot.view(np.uint32)).sum()} t differ")
    # the frame: which pixels differ, and from which path length on (a path that branched the other way on a last-bit
    # difference of a transcendental shows as agreement up to some number of bounces and a jump after it)
    op = dsc.options
    for mb in range(1, op.maxBounces + 1):
        v = capi.RtOptions(op.width, op.height, op.bias, op.exposure, mb, op.numRays, op.fisheye, op.focus, op.lens)
        dev.set_view(options=v); orc.set_view(options=v)
        img, st = dev.render(seed=int(os.environ.get("FUZZ_FRAME_SEED", "0"))); ref, ost = orc.render(seed=int(os.environ.get("FUZZ_FRAME_SEED", "0")))
        diff = np.abs(img[..., :3] - ref[..., :3]).max(-1)
        ys, xs = np.nonzero(diff > 1e-6)
        print(f"maxBounces {mb}: {ys.size} pixels differ by > 1e-6 (max {diff.max():.3g}), rays gpu {st.raysClosest + st.raysAny + st.raysSkipped} cpu {ost.raysClosest + ost.raysAny}")
        for y, x in list(zip(ys, xs))[:6]:
            print(f"   pixel ({x},{y}): gpu {img[y, x].tolist()} cpu {ref[y, x].tolist()}")


def check_scene(hs, dev, orc, o, d, frame_seed):
    """One scene on the HIP library and on the CPU restatement: rays (o, d) as closest and as any-hit queries, the primary
    rays, and the scene's frame. Returns {"ok", "same_counts", "rays", "what"}: ok = every query bit-equal (ids, bits of
    t, point, normal; visibility answers), alpha equal, at most 1 % of the pixels (or 2) off by more than 1e-3."""
    g, r = dev.trace_closest(o, d), orc.trace_closest(o, d)
    ok = all(same_bits(a, b) for a, b in zip(g, r))
    ok = ok and np.array_equal(dev.trace_any(o, d), orc.trace_any(o, d))
    gi, gt = dev.trace_primary(); oi, ot = orc.trace_primary()
    ok = ok and np.array_equal(gi, oi) and same_bits(gt, ot)
    img, st = dev.render(seed=frame_seed); ref, ost = orc.render(seed=frame_seed)
    finite = np.isfinite(ref).all() and np.isfinite(img).all()
    mse = float(((img[..., :3] - ref[..., :3]) ** 2).mean()) if finite else 0.0
    psnr = 200.0 if mse == 0 else 10 * np.log10(1.0 / mse)
    same_counts = st.raysClosest + st.raysAny + st.raysSkipped == ost.raysClosest + ost.raysAny
    # A last-bit difference between the device's and glibc's sinf / cosf / acosf can send ONE path of a frame the other
    # way at a branch (lobe choice, Fresnel test): a handful of rays more or less and a few pixels off. That is
    # counted apart ("diverged"); a mismatch is anything more: queries not bit-equal, alpha differing, or more
    # than 1 % of the pixels off.
    npx = img.shape[0] * img.shape[1]
    off = int((np.abs(img[..., :3] - ref[..., :3]).max(-1) > 1e-3).sum()) if finite else npx
    alpha_ok = bool(np.array_equal(img[..., 3], ref[..., 3]))
    ok_img = alpha_ok and off <= max(2, npx // 100)
    d_ = hs.desc.contents
    o_ = d_.options
    what = (f"prims {d_.numPrims} planes {d_.numPlanes} lights {d_.numLights} queries_ok {ok} alpha_ok {alpha_ok} psnr {psnr:.1f} "
            f"counts {same_counts} | {o_.width}x{o_.height} spp {o_.numRays} bounces {o_.maxBounces} pixels>1e-3: {off} rays gpu "
            f"{st.raysClosest + st.raysAny + st.raysSkipped} cpu {ost.raysClosest + ost.raysAny}")
    return {"ok": bool(ok and ok_img), "same_counts": bool(same_counts), "what": what, "off": off,
            "rays": 2 * len(o) + gi.size + st.raysClosest + st.raysAny}


def fuzz_device_bvh(args):
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    k = bad = rays = built = skipped = 0
    while time.time() - t0 < args.seconds:
        hs = random_scene(rng, k)
        n = 20000
        o = (rng.normal(size=(n, 3)) * np.array([3, 2, 3]) + np.array([0, 0, -3])).astype(np.float32)
        d = rng.normal(size=(n, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        d = (d * (1.0 + np.abs(rng.normal(size=(n, 1))))).astype(np.float32)
        if hs.desc.contents.numPrims == 0:
            k += 1
            continue
        dev = capi.device_scene(hs.desc)
        # Zero-area triangles get infinite edge vectors from the reference's set-up (reciprocals of a zero cross product):
        # every barycentric comparison is then one with NaN and rejects nothing, so such a triangle is hit wherever a ray
        # meets its PLANE — far outside its box — provided the traversal reaches its leaf at all, which depends on the
        # boxes of the tree. Rays that either tree answers with such an object are tree-dependent and left out.
        dsc = hs.desc.contents
        rec = np.ctypeslib.as_array(dsc.primHit, shape=(dsc.numPrims, 12))
        unbounded = np.ctypeslib.as_array(dsc.primOrigIndex, shape=(dsc.numPrims,))[~np.isfinite(rec).all(1)].copy()
        hs.flatten_without_bvh()
        lb = capi.device_scene(hs.desc)
        a, b = dev.trace_closest(o, d), lb.trace_closest(o, d)
        ha, hb = dev.trace_any(o, d), lb.trace_any(o, d)
        keep = ~(np.isin(a[0], unbounded) | np.isin(b[0], unbounded))
        skipped += int((~keep).sum())
        # a visibility answer cannot say which object it found: with such triangles in the scene it is only checked where
        # the closest hit is a bounded object (then both trees must say "something is there")
        found = keep & (a[0] >= 0)
        ok = same_bits(a[1][keep], b[1][keep]) and bool(ha[found].all()) and bool(hb[found].all()) and \
            (unbounded.size > 0 or np.array_equal(ha, hb))
        # which of several objects hit at exactly the same t is reported is the tree's choice; point and normal follow it
        swapped = float((a[0] != b[0]).mean())
        rays += 2 * n; built += 1
        if not ok:
            bad += 1
            diff = np.nonzero((a[1].view(np.uint32) != b[1].view(np.uint32)) & keep)[0]
            print(f"MISMATCH scene {k}: prims {hs.desc.contents.numPrims} t differs on {diff.size} rays, first {diff[:3].tolist()} "
                  f"host {a[1][diff[:3]].tolist()} device {b[1][diff[:3]].tolist()} ids swapped {swapped:.4f}", flush=True)
        dev.close(); lb.close()
        k += 1
        if k % 500 == 0:
            print(f"{k} scenes, {built} trees built on the GPU, {rays / 1e6:.1f} M rays, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {k} scenes, {built} trees built on the GPU, {rays / 1e6:.1f} M rays, {bad} mismatches; {skipped} rays left out "
          f"(answered with a zero-area triangle, whose hits depend on the tree)")
    return 1 if bad else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, default=-1, help="replay the random stream up to this scene and report on it in detail")
    ap.add_argument("--device-bvh", action="store_true",
                    help="instead: the tree built on the GPU against the host-built one on every scene (rays of unit length or longer, "
                         "for which the answer does not depend on the visit order): bits of t and visibility answers equal")
    args = ap.parse_args()
    if args.device_bvh:
        return fuzz_device_bvh(args)
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    k = bad = rays = diverged = 0
    while time.time() - t0 < args.seconds:
        hs, rays_of, _ = scene_of_the_stream(rng, k)
        o, d = rays_of(rng, 20000)
        if args.only >= 0 and k < args.only:
            k += 1
            continue
        dev, orc = capi.device_scene(hs.desc), oracle_scene(hs.desc, threads=8)
        if args.only >= 0:
            detail(hs, dev, orc, o, d)
            return 0
        res = check_scene(hs, dev, orc, o, d, k)
        diverged += int(res["ok"] and not res["same_counts"])
        rays += res["rays"]
        if not res["ok"]:
            bad += 1
            print(f"MISMATCH scene {k}: {res['what']}", flush=True)
        dev.close(); orc.close()
        k += 1
        if k % 500 == 0:
            print(f"{k} scenes, {rays / 1e6:.1f} M rays, {bad} mismatches, {diverged} frames with a diverged path, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {k} scenes, {rays / 1e6:.1f} M rays, {bad} mismatches, {diverged} frames with a diverged path")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
