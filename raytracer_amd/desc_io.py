"""A flattened scene (RtSceneDesc: BVH nodes, object records, planes, materials, lights, textures, camera, options) as a
directory of plain .npy files, and back.

What it is for: several processes on one host render tiles of the same frame (one process per GPU). The scene files are
parsed and the BVH is built ONCE, by one process with all host threads; the others map the finished arrays instead of
repeating 8 x (parse + build) on the same cores. The arrays are exactly what `rt_scene_create` reads, so a scene loaded
from the directory gives the same frames, bit for bit, as the scene it was saved from.

The reference builds its scene once per process too (src/main.cpp:27-32); it has one process.
"""
import ctypes as C
import json
import os

import numpy as np

from . import capi

# name, numpy dtype, values per object
_PRIM_ARRAYS = (("primKind", np.int32, 1), ("primHit", np.float32, 12), ("primNormals", np.float32, 9),
                ("primTexCoords", np.float32, 9), ("primColor", np.float32, 4), ("primMaterial", np.int32, 1),
                ("primTexture", np.int32, 1), ("primOrigIndex", np.int32, 1), ("primAabbMin", np.float32, 3),
                ("primAabbMax", np.float32, 3))
_TABLES = (("planes", capi.RtPlane, "numPlanes"), ("materials", capi.RtMaterial, "numMaterials"), ("lights", capi.RtLight, "numLights"))


def save_desc(desc_ptr, directory):
    """Writes everything behind the description into `directory` (created if needed). The last file written is
    `desc.json`, so a reader that finds it finds a complete directory."""
    d = desc_ptr.contents
    os.makedirs(directory, exist_ok=True)
    n = d.numPrims
    meta = {"numPrims": int(n), "numNodes": int(d.numNodes), "arrays": [], "textures": []}

    def put(name, arr):
        np.save(os.path.join(directory, name + ".npy"), arr)
        meta["arrays"].append(name)

    if d.numNodes and d.nodes:
        put("nodes", np.ctypeslib.as_array(C.cast(d.nodes, C.POINTER(C.c_int32)), shape=(d.numNodes, 8)))
    for name, dtype, width in _PRIM_ARRAYS:
        p = getattr(d, name)
        if n and p:
            put(name, np.ctypeslib.as_array(p, shape=(n * width,)))
    for name, ctype, count in _TABLES:
        k = getattr(d, count)
        if k:
            put(name, np.frombuffer(C.string_at(getattr(d, name), k * C.sizeof(ctype)), np.uint8))
    for i in range(d.numTextures):
        t = d.textures[i]
        meta["textures"].append([int(t.width), int(t.height)])
        if t.width > 0 and t.height > 0:
            put(f"texture{i}", np.ctypeslib.as_array(t.rgba, shape=(t.height * t.width * 4,)))
    put("camera", np.frombuffer(C.string_at(C.addressof(d.camera), C.sizeof(capi.RtCamera)), np.uint8))
    put("options", np.frombuffer(C.string_at(C.addressof(d.options), C.sizeof(capi.RtOptions)), np.uint8))
    tmp = os.path.join(directory, "desc.json.tmp")
    with open(tmp, "w") as f:
        json.dump(meta, f)
    os.replace(tmp, os.path.join(directory, "desc.json"))


class LoadedDesc:
    """`.desc` points to an RtSceneDesc over arrays mapped from a directory written by save_desc (kept alive here)."""

    def __init__(self, directory):
        with open(os.path.join(directory, "desc.json")) as f:
            meta = json.load(f)
        have = set(meta["arrays"])
        self._keep = {}

        def get(name, dtype):
            a = np.load(os.path.join(directory, name + ".npy"), mmap_mode="r")
            a = np.ascontiguousarray(a, dtype)  # a read-only map of the right type is passed on as it is
            self._keep[name] = a
            return a

        def ptr(a, ctype):
            return C.cast(a.ctypes.data, C.POINTER(ctype))

        d = capi.RtSceneDesc()
        d.numPrims = meta["numPrims"]
        d.numNodes = meta["numNodes"] if "nodes" in have else 0
        d.nodes = ptr(get("nodes", np.int32), capi.RtBvhNode) if "nodes" in have else None
        for name, dtype, _ in _PRIM_ARRAYS:
            ctype = C.c_int32 if dtype == np.int32 else C.c_float
            setattr(d, name, ptr(get(name, dtype), ctype) if name in have else None)
        for name, ctype, count in _TABLES:
            if name in have:
                raw = get(name, np.uint8)
                setattr(d, count, raw.size // C.sizeof(ctype))
                setattr(d, name, ptr(raw, ctype))
            else:
                setattr(d, count, 0)
                setattr(d, name, None)
        texs = (capi.RtTexture * max(1, len(meta["textures"])))()
        for i, (w, h) in enumerate(meta["textures"]):
            texs[i].width, texs[i].height = w, h
            texs[i].rgba = ptr(get(f"texture{i}", np.float32), C.c_float) if w > 0 and h > 0 else None
        self._keep["textures"] = texs
        d.numTextures = len(meta["textures"])
        d.textures = C.cast(texs, C.POINTER(capi.RtTexture)) if meta["textures"] else None
        C.memmove(C.addressof(d.camera), get("camera", np.uint8).ctypes.data, C.sizeof(capi.RtCamera))
        C.memmove(C.addressof(d.options), get("options", np.uint8).ctypes.data, C.sizeof(capi.RtOptions))
        self._d = d
        self.desc = C.pointer(d)
        self.desc._owner = self


def load_desc(directory):
    return LoadedDesc(directory)
