"""A scene description (RtSceneDesc) kept as plain arrays in an .npz fixture, and back: the scenes the reference ships
under example_scenes/ travel to the GPU box this way — flattened by this repository's host parser where the reference's
files exist (tools/make_golden.py bundled, which also checks the records against the compiled reference), rebuilt here
into the C structure without touching a scene file. TEST-SIDE ONLY."""
import ctypes as C

import numpy as np

from raytracer_amd import capi

_ARRAYS = (("primKind", np.int32, 1), ("primHit", np.float32, 12), ("primNormals", np.float32, 9), ("primColor", np.float32, 4),
           ("primMaterial", np.int32, 1), ("primTexture", np.int32, 1), ("primOrigIndex", np.int32, 1),
           ("primAabbMin", np.float32, 3), ("primAabbMax", np.float32, 3))
_TABLES = (("planes", capi.RtPlane, "numPlanes"), ("materials", capi.RtMaterial, "numMaterials"), ("lights", capi.RtLight, "numLights"))


def arrays_of(desc_ptr):
    """dict of numpy arrays holding everything behind an RtSceneDesc without textures or texture coordinates"""
    d = desc_ptr.contents
    assert d.numTextures == 0 and not d.primTexCoords, "fixture scenes carry no textures"
    n = d.numPrims
    out = {"nodes": np.ctypeslib.as_array(C.cast(d.nodes, C.POINTER(C.c_int32)), shape=(d.numNodes, 8)).copy() if d.numNodes else np.zeros((0, 8), np.int32)}
    for name, dtype, width in _ARRAYS:
        p = getattr(d, name)
        out[name] = np.ctypeslib.as_array(p, shape=(n * width,)).copy() if n else np.zeros(0, dtype)
    for name, ctype, count in _TABLES:
        k = getattr(d, count)
        out[name] = np.frombuffer(C.string_at(getattr(d, name), k * C.sizeof(ctype)), np.uint8).copy() if k else np.zeros(0, np.uint8)
    out["camera"] = np.frombuffer(C.string_at(C.addressof(d.camera), C.sizeof(capi.RtCamera)), np.uint8).copy()
    out["options"] = np.frombuffer(C.string_at(C.addressof(d.options), C.sizeof(capi.RtOptions)), np.uint8).copy()
    return out


class DescFromArrays:
    """`.desc` is a pointer to an RtSceneDesc over the arrays (which this object keeps alive)."""

    def __init__(self, z, width=None, height=None):
        self._keep = {}
        d = capi.RtSceneDesc()
        nodes = np.ascontiguousarray(z["nodes"], np.int32)
        self._keep["nodes"] = nodes
        d.numNodes = nodes.shape[0]
        d.nodes = C.cast(nodes.ctypes.data, C.POINTER(capi.RtBvhNode)) if nodes.shape[0] else None
        d.numPrims = int(np.asarray(z["primKind"]).shape[0])
        for name, dtype, _ in _ARRAYS:
            a = np.ascontiguousarray(z[name], dtype)
            self._keep[name] = a
            ctype = C.c_int32 if dtype == np.int32 else C.c_float
            setattr(d, name, C.cast(a.ctypes.data, C.POINTER(ctype)) if a.size else None)
        for name, ctype, count in _TABLES:
            raw = np.ascontiguousarray(z[name], np.uint8)
            self._keep[name] = raw
            setattr(d, count, raw.size // C.sizeof(ctype))
            setattr(d, name, C.cast(raw.ctypes.data, C.POINTER(ctype)) if raw.size else None)
        C.memmove(C.addressof(d.camera), np.ascontiguousarray(z["camera"], np.uint8).ctypes.data, C.sizeof(capi.RtCamera))
        C.memmove(C.addressof(d.options), np.ascontiguousarray(z["options"], np.uint8).ctypes.data, C.sizeof(capi.RtOptions))
        if width:
            d.options.width, d.options.height = int(width), int(height)
        self._d = d
        self.desc = C.pointer(d)
        self.desc._owner = self
