"""Readers for the raw dumps oracle/_ref/ref_harness writes (see oracle/ref_harness.cpp) and a helper to run it.
Only usable where the harness binary exists (this container); fixtures under tests/golden/ are made from it."""
import os
import struct
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(REPO, "oracle", "_ref", "ref_harness")


def have_harness():
    return os.path.exists(HARNESS)


def run_harness(mode, src, out, *extra, cwd=None):
    r = subprocess.run([HARNESS, mode, src, out, *map(str, extra)], cwd=cwd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"ref_harness {mode} {src}: exit {r.returncode}\n{r.stdout}\n{r.stderr}")
    return r.stdout


class _Reader:
    def __init__(self, path):
        self.b = open(path, "rb").read()
        self.o = 0

    def i32(self):
        v = struct.unpack_from("<i", self.b, self.o)[0]; self.o += 4; return v

    def u32(self):
        v = struct.unpack_from("<I", self.b, self.o)[0]; self.o += 4; return v

    def f32(self, n=1):
        v = np.frombuffer(self.b, dtype="<f4", count=n, offset=self.o).copy(); self.o += 4 * n
        return v if n > 1 else v[0]

    def tex(self):
        w, h = self.i32(), self.i32()
        if w == 0 and h == 0:
            return None
        return (w, h, self.u32())


def read_scene(path):
    r = _Reader(path)
    s = {"width": r.i32(), "height": r.i32(), "bias": r.f32(), "exposure": r.f32(), "maxBounces": r.i32(),
         "numRays": r.i32(), "fisheye": r.i32(), "focus": r.f32(), "lens": r.f32()}
    s["eye"], s["forward"], s["right"], s["up"] = r.f32(3), r.f32(3), r.f32(3), r.f32(3)

    def material():
        v = r.f32(7)
        return {"params": v, "type": r.i32(), "numLobes": r.i32()}

    objs = []
    for _ in range(r.i32()):
        kind = r.i32()
        o = {"kind": kind}
        if kind == 0:
            for k in ("p1", "normal", "e1", "e2", "n1", "n2", "n3", "t1", "t2", "t3"):
                o[k] = r.f32(3)
        elif kind == 1:
            o["center"] = r.f32(3); o["r"] = r.f32()
        o["aabbMin"], o["aabbMax"], o["centroid"] = r.f32(3), r.f32(3), r.f32(3)
        o["color"] = r.f32(4)
        o["material"] = material()
        o["texture"] = r.tex()
        objs.append(o)
    s["objects"] = objs
    planes = []
    for _ in range(r.i32()):
        p = {"normal": r.f32(3), "point": r.f32(3), "texTopLeft": r.f32(3), "texZoom": r.f32(), "texShift": r.f32(3),
             "color": r.f32(4)}
        p["material"] = material()
        p["texture"] = r.tex()
        planes.append(p)
    s["planes"] = planes
    lights = []
    for _ in range(r.i32()):
        t = r.i32()
        l = {"type": t, "color": r.f32(4), "v": r.f32(3)}
        if t == 2:
            l["radius"] = r.f32(); l["texture"] = r.tex()
        lights.append(l)
    s["lights"] = lights
    return s


def read_bvh(path):
    r = _Reader(path)
    n, m = r.i32(), r.i32()
    nodes = np.frombuffer(r.b, dtype="<i4", count=n * 8, offset=r.o).reshape(n, 8).copy(); r.o += n * 32
    perm = np.frombuffer(r.b, dtype="<i4", count=m, offset=r.o).copy()
    return nodes, perm


def read_hits(path):
    r = _Reader(path)
    w, h, nobj, npl = r.i32(), r.i32(), r.i32(), r.i32()
    a = np.frombuffer(r.b, dtype="<i4", count=w * h * 3, offset=r.o).reshape(h, w, 3)
    return {"width": w, "height": h, "numObjects": nobj, "numPlanes": npl, "orig": a[..., 0].copy(),
            "post": a[..., 1].copy(), "tbits": a[..., 2].copy().view(np.uint32)}


def read_render(path):
    r = _Reader(path)
    w, h = r.i32(), r.i32()
    img = np.frombuffer(r.b, dtype="<f4", count=w * h * 4, offset=r.o).reshape(h, w, 4).copy(); r.o += w * h * 16
    rgba8 = np.frombuffer(r.b, dtype=np.uint8, count=w * h * 4, offset=r.o).reshape(h, w, 4).copy()
    return img, rgba8


def read_tex(path):
    r = _Reader(path)
    w, h = r.i32(), r.i32()
    return np.frombuffer(r.b, dtype="<f4", count=w * h * 4, offset=r.o).reshape(h, w, 4).copy()


def read_trace(path):
    a = np.frombuffer(open(path, "rb").read(), dtype="<i4").reshape(-1, 8)
    return {"id": a[:, 0].copy(), "tbits": a[:, 1].copy().view(np.uint32), "point": a[:, 2:5].copy().view(np.float32),
            "normal": a[:, 5:8].copy().view(np.float32)}


def fnv1a(data: bytes) -> int:
    h = 2166136261
    # vectorised FNV is awkward; the textures hashed here are small
    for b in data:
        h = ((h ^ b) * 16777619) & 0xFFFFFFFF
    return h
