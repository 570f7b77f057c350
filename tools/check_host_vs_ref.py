#!/usr/bin/env python3
"""Bit-compare what this repository's host library builds from a scene file (records, BVH nodes, object
permutation) with what the compiled reference builds from the same file (oracle/_ref/ref_harness).
    python tools/check_host_vs_ref.py scene.sdml [...]"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi  # noqa: E402
from tests import refdump  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def same(a, b):
    """Bit-equal; two NaNs are equal whatever their sign and payload (which operand order a compiler picked for a
    commutative operation decides that, and nothing reads it)."""
    a = np.ascontiguousarray(a, dtype=np.float32); b = np.ascontiguousarray(b, dtype=np.float32)
    return a.shape == b.shape and bool(np.all((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))))


def check(path, with_bvh=True):
    """with_bvh=False: parsed records only (for files on which the reference's own BVH build dies: src/acceleration/BVH.cpp:93-99
    indexes its bins with an undefined value when a node's centroids coincide on an axis)."""
    tmp = tempfile.mkdtemp()
    refdump.run_harness("scene", path, tmp + "/scene.bin")
    ref = refdump.read_scene(tmp + "/scene.bin")
    hs = capi.HostScene.load(path).build()
    if with_bvh:
        refdump.run_harness("bvh", path, tmp + "/bvh.bin")
        rnodes, rperm = refdump.read_bvh(tmp + "/bvh.bin")
    else:
        dd = hs.desc.contents
        rperm = np.ctypeslib.as_array(dd.primOrigIndex, shape=(dd.numPrims,)).copy()
        rnodes = hs.nodes() if dd.numNodes else np.zeros((0, 8), np.int32)
    d = hs.desc.contents
    errs = []
    late_fields = path.endswith(".txt") and any(line.split()[:1] in (["roughness"], ["ior"]) for line in open(path, errors="replace"))
    o = d.options
    for k in ("width", "height", "maxBounces", "numRays", "fisheye"):
        if getattr(o, k) != ref[k]:
            errs.append(f"option {k}: {getattr(o, k)} != {ref[k]}")
    for k in ("bias", "exposure", "focus", "lens"):
        if not same(getattr(o, k), ref[k]):
            errs.append(f"option {k}: {getattr(o, k)} != {ref[k]}")
    for k in ("eye", "forward", "right", "up"):
        if not same(list(getattr(d.camera, k)), ref[k]):
            errs.append(f"camera {k}: {list(getattr(d.camera, k))} != {ref[k]}")
    n = d.numPrims
    if n != len(ref["objects"]):
        errs.append(f"numPrims {n} != {len(ref['objects'])}")
    else:
        orig = np.ctypeslib.as_array(d.primOrigIndex, shape=(n,)) if n else np.zeros(0, np.int32)
        if not np.array_equal(orig, rperm):
            errs.append(f"BVH permutation differs at {np.flatnonzero(orig != rperm)[:5]}")
        if n:
            hit = np.ctypeslib.as_array(d.primHit, shape=(n, 12)); nrm = np.ctypeslib.as_array(d.primNormals, shape=(n, 9))
            col = np.ctypeslib.as_array(d.primColor, shape=(n, 4)); kind = np.ctypeslib.as_array(d.primKind, shape=(n,))
            mat = np.ctypeslib.as_array(d.primMaterial, shape=(n,)); tex = np.ctypeslib.as_array(d.primTexture, shape=(n,))
            bad = 0
            for i in range(n):
                r = ref["objects"][orig[i]]
                if kind[i] != r["kind"]:
                    bad += 1; continue
                if r["kind"] == 0:
                    ok = same(hit[i], np.concatenate([r["p1"], r["normal"], r["e1"], r["e2"]])) and \
                        same(nrm[i], np.concatenate([r["n1"], r["n2"], r["n3"]]))
                else:
                    ok = same(hit[i, :4], np.concatenate([r["center"], [r["r"]]]))
                ok = ok and same(col[i], r["color"])
                m = d.materials[mat[i]]
                ok = ok and m.type == r["material"]["type"]
                if r["material"]["numLobes"] > 0:
                    mine, theirs = [m.Kd, m.Ks, m.eta, m.Kr, m.Kt, m.Ka, m.roughness], list(r["material"]["params"])
                    if late_fields:
                        # `roughness` / `ior` lines of a txt scene assign Material::roughness / eta after the lobes were
                        # built (src/parser/parser.cpp:268-270, 288-290) and nothing reads the fields afterwards; the
                        # host library keeps the values the lobes were built from
                        mine, theirs = mine[:2] + mine[3:6], theirs[:2] + theirs[3:6]
                    ok = ok and same(mine, theirs)
                ok = ok and ((tex[i] >= 0) == (r["texture"] is not None))
                if r["texture"] is not None and tex[i] >= 0:
                    t = d.textures[tex[i]]
                    ok = ok and (t.width, t.height) == r["texture"][:2]
                    data = np.ctypeslib.as_array(t.rgba, shape=(t.width * t.height * 4,))
                    if t.width * t.height <= 1 << 16:
                        ok = ok and refdump.fnv1a(data.tobytes()) == r["texture"][2]
                bad += not ok
            if bad:
                errs.append(f"{bad} of {n} object records differ")
    nodes = hs.nodes() if d.numNodes else np.zeros((0, 8), np.int32)
    if nodes.shape != rnodes.shape:
        errs.append(f"node count {nodes.shape[0]} != {rnodes.shape[0]}")
    elif not np.array_equal(nodes, rnodes):
        errs.append(f"{int((nodes != rnodes).any(1).sum())} BVH nodes differ")
    if d.numPlanes != len(ref["planes"]):
        errs.append("plane count")
    else:
        for k in range(d.numPlanes):
            p, r = d.planes[k], ref["planes"][k]
            ok = same(list(p.normal), r["normal"]) and same(list(p.point), r["point"]) and same(list(p.color), r["color"]) \
                and (p.texture < 0 or (same(list(p.texTopLeft), r["texTopLeft"]) and same(p.texZoom, r["texZoom"])
                                       and same(list(p.texShift), r["texShift"]))) \
                and d.materials[p.material].type == r["material"]["type"] and ((p.texture >= 0) == (r["texture"] is not None))
            if not ok:
                errs.append(f"plane {k} differs")
    if d.numLights != len(ref["lights"]):
        errs.append("light count")
    else:
        for k in range(d.numLights):
            l, r = d.lights[k], ref["lights"][k]
            ok = l.type == r["type"] and same(list(l.color), r["color"]) and same(list(l.v), r["v"])
            if l.type == 2:
                ok = ok and same(l.radius, r["radius"]) and ((l.texture >= 0) == (r["texture"] is not None))
            if not ok:
                errs.append(f"light {k} differs: {list(l.v)} vs {r['v']}")
    print(("OK   " if not errs else "FAIL ") + os.path.basename(path), f"prims={n} nodes={d.numNodes}", *errs, sep="  ")
    return not errs


if __name__ == "__main__":
    ok = all([check(p) for p in sys.argv[1:]])
    sys.exit(0 if ok else 1)
