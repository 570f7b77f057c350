"""ctypes mirror of include/rt_amd.h and include/rt_host.h.

Two shared libraries live next to this file after `make` (or `__graft_entry__.build()`):
  librt_host.so        host scene library (parsers, scene records, BVH build, PNG pixels)
  libraytracer_amd.so  the HIP render loop for gfx950

Nothing here falls back to a CPU implementation: if libraraytracer_amd.so is missing or fails to load,
`load_device_library()` raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

RT_OK = 0
RT_PRIM_TRIANGLE, RT_PRIM_SPHERE = 0, 1
RT_MAT_DEFAULT, RT_MAT_GLASS, RT_MAT_MIRROR, RT_MAT_PLASTIC, RT_MAT_METAL = range(5)
RT_LIGHT_DISTANT, RT_LIGHT_POINT, RT_LIGHT_ENVIRONMENT = range(3)
RT_RNG_COUNTER, RT_RNG_REF_MT19937 = 0, 1

_f3 = C.c_float * 3
_f4 = C.c_float * 4


class RtBvhNode(C.Structure):
    _fields_ = [("aabbMin", _f3), ("aabbMax", _f3), ("rightOrStart", C.c_int32), ("numObjects", C.c_int32)]


class RtMaterial(C.Structure):
    _fields_ = [("type", C.c_int32), ("Kd", C.c_float), ("Ks", C.c_float), ("eta", C.c_float), ("Kr", C.c_float),
                ("Kt", C.c_float), ("Ka", C.c_float), ("roughness", C.c_float)]


class RtPlane(C.Structure):
    _fields_ = [("normal", _f3), ("point", _f3), ("texTopLeft", _f3), ("texZoom", C.c_float), ("texShift", _f3),
                ("color", _f4), ("material", C.c_int32), ("texture", C.c_int32)]


class RtLight(C.Structure):
    _fields_ = [("type", C.c_int32), ("color", _f4), ("v", _f3), ("radius", C.c_float), ("texture", C.c_int32)]


class RtTexture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgba", C.POINTER(C.c_float))]


class RtCamera(C.Structure):
    _fields_ = [("eye", _f3), ("forward", _f3), ("right", _f3), ("up", _f3)]


class RtOptions(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("bias", C.c_float), ("exposure", C.c_float),
                ("maxBounces", C.c_int32), ("numRays", C.c_int32), ("fisheye", C.c_int32), ("focus", C.c_float),
                ("lens", C.c_float)]


class RtSceneDesc(C.Structure):
    _fields_ = [
        ("nodes", C.POINTER(RtBvhNode)), ("numNodes", C.c_int32),
        ("numPrims", C.c_int32),
        ("primKind", C.POINTER(C.c_int32)), ("primHit", C.POINTER(C.c_float)), ("primNormals", C.POINTER(C.c_float)),
        ("primTexCoords", C.POINTER(C.c_float)), ("primColor", C.POINTER(C.c_float)),
        ("primMaterial", C.POINTER(C.c_int32)), ("primTexture", C.POINTER(C.c_int32)),
        ("primOrigIndex", C.POINTER(C.c_int32)),
        ("numPlanes", C.c_int32), ("planes", C.POINTER(RtPlane)),
        ("numMaterials", C.c_int32), ("materials", C.POINTER(RtMaterial)),
        ("numLights", C.c_int32), ("lights", C.POINTER(RtLight)),
        ("numTextures", C.c_int32), ("textures", C.POINTER(RtTexture)),
        ("camera", RtCamera), ("options", RtOptions),
        ("primAabbMin", C.POINTER(C.c_float)), ("primAabbMax", C.POINTER(C.c_float)),
    ]


class RtRenderParams(C.Structure):
    _fields_ = [("rngMode", C.c_int32), ("seed", C.c_uint32), ("partIndex", C.c_int32), ("partCount", C.c_int32),
                ("tileW", C.c_int32), ("tileH", C.c_int32), ("samplesPerPass", C.c_int32),
                ("collectCounters", C.c_int32), ("skipExposure", C.c_int32),
                ("firstSample", C.c_int32), ("sampleCount", C.c_int32), ("keepAccumulator", C.c_int32),
                ("adaptiveThreshold", C.c_float), ("adaptiveBatch", C.c_int32), ("adaptiveMinSamples", C.c_int32)]


class RtStats(C.Structure):
    _fields_ = [("cameraSamples", C.c_uint64), ("raysClosest", C.c_uint64), ("raysAny", C.c_uint64),
                ("innerExpansions", C.c_uint64), ("primTests", C.c_uint64), ("planeTests", C.c_uint64),
                ("raysExtend", C.c_uint64), ("raysShadow", C.c_uint64), ("innerExtend", C.c_uint64),
                ("primExtend", C.c_uint64), ("innerShadow", C.c_uint64), ("primShadow", C.c_uint64),
                ("msTotal", C.c_double), ("msRaygen", C.c_double), ("msExtend", C.c_double), ("msShade", C.c_double),
                ("msShadow", C.c_double), ("msDirect", C.c_double), ("msAccumulate", C.c_double),
                ("msResolve", C.c_double), ("launchesExtend", C.c_uint32), ("launchesShadow", C.c_uint32),
                ("passes", C.c_uint32), ("laneUseInnerExtend", C.c_double), ("laneUsePrimExtend", C.c_double),
                ("laneUseInnerShadow", C.c_double), ("laneUsePrimShadow", C.c_double), ("raysSkipped", C.c_uint64),
                ("raysPlaneAnswered", C.c_uint64), ("launchesFused", C.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


# ------------------------------------------------------------------------------------------ host library
_host = None


def load_host_library():
    global _host
    if _host is not None:
        return _host
    path = os.path.join(_HERE, "librt_host.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `make host` (or __graft_entry__.build())")
    L = C.CDLL(path)
    vp, i32, f32, cp = C.c_void_p, C.c_int32, C.c_float, C.c_char_p
    L.rth_last_error.restype = cp
    L.rth_scene_load.restype = vp; L.rth_scene_load.argtypes = [cp]
    L.rth_scene_new.restype = vp; L.rth_scene_new.argtypes = [i32, i32]
    L.rth_scene_free.argtypes = [vp]
    L.rth_scene_add_material.restype = i32; L.rth_scene_add_material.argtypes = [vp, C.POINTER(RtMaterial)]
    L.rth_scene_add_texture.restype = i32; L.rth_scene_add_texture.argtypes = [vp, i32, i32, C.POINTER(f32)]
    L.rth_scene_add_texture_file.restype = i32; L.rth_scene_add_texture_file.argtypes = [vp, cp]
    L.rth_scene_add_sphere.argtypes = [vp, _f3, f32, _f4, i32, i32]
    L.rth_scene_add_plane.argtypes = [vp, _f3, f32, _f4, i32, _f3, f32, _f3, i32, i32]
    L.rth_scene_add_mesh.argtypes = [vp, C.c_int64, C.POINTER(f32), C.POINTER(f32), C.c_int64, C.POINTER(i32), _f3, f32,
                                     _f4, i32]
    L.rth_scene_add_light.argtypes = [vp, C.POINTER(RtLight)]
    L.rth_scene_set_camera.argtypes = [vp, _f3, _f3, _f3]
    L.rth_scene_set_options.argtypes = [vp, C.POINTER(RtOptions)]
    L.rth_scene_build.argtypes = [vp]
    L.rth_scene_flatten_without_bvh.argtypes = [vp]
    L.rth_scene_desc.restype = C.POINTER(RtSceneDesc); L.rth_scene_desc.argtypes = [vp]
    L.rth_scene_filename.restype = cp; L.rth_scene_filename.argtypes = [vp]
    L.rth_scene_log_count.restype = i32; L.rth_scene_log_count.argtypes = [vp]
    L.rth_scene_log_line.restype = cp; L.rth_scene_log_line.argtypes = [vp, i32]
    L.rth_png_read_linear.argtypes = [cp, C.POINTER(i32), C.POINTER(i32), C.POINTER(C.POINTER(f32))]
    L.rth_png_read_rgba8.argtypes = [cp, C.POINTER(i32), C.POINTER(i32), C.POINTER(C.POINTER(C.c_uint8))]
    L.rth_png_write_rgba8.argtypes = [cp, i32, i32, C.POINTER(C.c_uint8)]
    L.rth_to_srgb8.argtypes = [C.POINTER(f32), C.c_int64, C.POINTER(C.c_uint8)]
    L.rth_free.argtypes = [vp]
    L.rth_set_threads.argtypes = [i32]; L.rth_set_threads.restype = None
    L.rth_get_threads.restype = i32
    _host = L
    return L


def set_host_threads(n):
    """Host worker threads for parsing, triangle set-up and the BVH build (`-t N`); n <= 0 = default."""
    load_host_library().rth_set_threads(int(n))


def host_threads():
    return load_host_library().rth_get_threads()


class HostScene:
    """Owner of an RthScene: load or construct, build the BVH, expose the RtSceneDesc."""

    def __init__(self, handle):
        self._L = load_host_library()
        self._h = handle
        self._keep = []

    @classmethod
    def load(cls, path):
        L = load_host_library()
        h = L.rth_scene_load(os.fsencode(path))
        if not h:
            raise RuntimeError("rth_scene_load: " + L.rth_last_error().decode())
        return cls(h)

    @classmethod
    def new(cls, width, height):
        L = load_host_library()
        return cls(L.rth_scene_new(width, height))

    def close(self):
        if self._h:
            self._L.rth_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != RT_OK:
            raise RuntimeError(f"{what}: {self._L.rth_last_error().decode()}")

    def add_material(self, type=RT_MAT_DEFAULT, Kd=1.0, Ks=0.0, eta=1.0, Kr=1.0, Kt=0.0, Ka=0.0, roughness=0.0):
        m = RtMaterial(type, Kd, Ks, eta, Kr, Kt, Ka, roughness)
        idx = self._L.rth_scene_add_material(self._h, C.byref(m))
        if idx < 0:
            raise RuntimeError("rth_scene_add_material: " + self._L.rth_last_error().decode())
        return idx

    def add_named_material(self, name):
        presets = {  # src/scene/Material.h:57-64 of the reference
            "default": dict(type=RT_MAT_DEFAULT),
            "glass": dict(type=RT_MAT_GLASS, Kd=0.0, Ks=1.0, eta=1.5, Kr=1.0, Kt=1.0, Ka=0.0, roughness=0.0),
            "plastic": dict(type=RT_MAT_PLASTIC, Kd=0.5, Ks=0.5, eta=1.3, Kr=1.0, Kt=0.0, Ka=0.0, roughness=0.1),
            "copper": dict(type=RT_MAT_METAL, Kd=0.0, Ks=1.0, eta=0.23883, Kr=0.9553, Kt=0.0, Ka=3.415658, roughness=0.01),
            "gold": dict(type=RT_MAT_METAL, Kd=0.0, Ks=1.0, eta=0.18104, Kr=0.99, Kt=0.0, Ka=3.068099, roughness=0.01),
            "mirror": dict(type=RT_MAT_MIRROR, Kd=0.0, Ks=1.0, eta=0.0, Kr=0.9, Kt=0.0, Ka=0.0, roughness=0.0),
        }
        return self.add_material(**presets[name])

    def add_texture(self, rgba):
        a = np.ascontiguousarray(rgba, dtype=np.float32)
        assert a.ndim == 3 and a.shape[2] == 4
        idx = self._L.rth_scene_add_texture(self._h, a.shape[1], a.shape[0], _fp(a))
        if idx < 0:
            raise RuntimeError("rth_scene_add_texture: " + self._L.rth_last_error().decode())
        return idx

    def add_sphere(self, center, r, color=(1, 1, 1, 1), material=0, texture=-1):
        self._check(self._L.rth_scene_add_sphere(self._h, _f3(*center), r, _f4(*color), material, texture), "add_sphere")

    def add_plane(self, normal, D, color=(1, 1, 1, 1), material=0, tex_top_left=(0, 0, 0), tex_zoom=1.0,
                  tex_shift=(0, 0, 0), texture=-1, flip_toward_camera=True):
        self._check(self._L.rth_scene_add_plane(self._h, _f3(*normal), D, _f4(*color), material, _f3(*tex_top_left),
                                                tex_zoom, _f3(*tex_shift), texture, int(flip_toward_camera)), "add_plane")

    def add_mesh(self, vertices, normals, indices, center=(0, 0, 0), scale=1.0, color=(1, 1, 1, 1), material=0):
        v = np.ascontiguousarray(vertices, dtype=np.float32)
        n = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32)
        i = np.ascontiguousarray(indices, dtype=np.int32)
        self._check(self._L.rth_scene_add_mesh(self._h, v.shape[0], _fp(v), None if n is None else _fp(n), i.shape[0],
                                               _ip(i), _f3(*center), scale, _f4(*color), material), "add_mesh")

    def add_light(self, type, color=(1, 1, 1, 1), v=(0, 0, 0), radius=0.0, texture=-1):
        l = RtLight(type, _f4(*color), _f3(*v), radius, texture)
        self._check(self._L.rth_scene_add_light(self._h, C.byref(l)), "add_light")

    def set_camera(self, eye=(0, 0, 0), forward=(0, 0, -1), up=(0, 1, 0)):
        self._check(self._L.rth_scene_set_camera(self._h, _f3(*eye), _f3(*forward), _f3(*up)), "set_camera")

    def set_options(self, width, height, bias=1e-4, exposure=-1.0, maxBounces=4, numRays=1, fisheye=0, focus=-1.0, lens=0.0):
        o = RtOptions(width, height, bias, exposure, maxBounces, numRays, fisheye, focus, lens)
        self._check(self._L.rth_scene_set_options(self._h, C.byref(o)), "set_options")

    def build(self):
        self._check(self._L.rth_scene_build(self._h), "rth_scene_build")
        return self

    def flatten_without_bvh(self):
        """The scene description without a host-built tree (the device library then builds its own BVH)."""
        self._check(self._L.rth_scene_flatten_without_bvh(self._h), "rth_scene_flatten_without_bvh")
        return self

    @property
    def desc(self):
        d = self._L.rth_scene_desc(self._h)
        if not d:
            raise RuntimeError("rth_scene_desc: " + self._L.rth_last_error().decode())
        d._owner = self  # the arrays behind the description live in this scene
        return d

    @property
    def filename(self):
        return self._L.rth_scene_filename(self._h).decode()

    @property
    def log(self):
        return [self._L.rth_scene_log_line(self._h, i).decode() for i in range(self._L.rth_scene_log_count(self._h))]

    # numpy views of the flattened arrays (valid while the scene is alive and not rebuilt)
    def nodes(self):
        d = self.desc.contents
        return np.ctypeslib.as_array(C.cast(d.nodes, C.POINTER(C.c_int32)), shape=(d.numNodes, 8)).copy()

    def prim_orig_index(self):
        d = self.desc.contents
        return np.ctypeslib.as_array(d.primOrigIndex, shape=(d.numPrims,)).copy()


def png_read_linear(path):
    L = load_host_library()
    w, h, p = C.c_int32(), C.c_int32(), C.POINTER(C.c_float)()
    if L.rth_png_read_linear(os.fsencode(path), C.byref(w), C.byref(h), C.byref(p)) != RT_OK:
        raise RuntimeError("rth_png_read_linear: " + L.rth_last_error().decode())
    out = np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
    L.rth_free(p)
    return out


def png_read_rgba8(path):
    L = load_host_library()
    w, h, p = C.c_int32(), C.c_int32(), C.POINTER(C.c_uint8)()
    if L.rth_png_read_rgba8(os.fsencode(path), C.byref(w), C.byref(h), C.byref(p)) != RT_OK:
        raise RuntimeError("rth_png_read_rgba8: " + L.rth_last_error().decode())
    out = np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
    L.rth_free(p)
    return out


def png_write_rgba8(path, rgba8):
    L = load_host_library()
    a = np.ascontiguousarray(rgba8, dtype=np.uint8)
    if L.rth_png_write_rgba8(os.fsencode(path), a.shape[1], a.shape[0], a.ctypes.data_as(C.POINTER(C.c_uint8))) != RT_OK:
        raise RuntimeError("rth_png_write_rgba8: " + L.rth_last_error().decode())


def to_srgb8(rgba):
    L = load_host_library()
    a = np.ascontiguousarray(rgba, dtype=np.float32)
    out = np.empty(a.shape, dtype=np.uint8)
    L.rth_to_srgb8(_fp(a), a.size // 4, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


# ------------------------------------------------------------------------------------------ render-loop ABI
def bind_render_abi(L, prefix):
    """Declare the rt_amd.h entry points on a loaded library. `prefix` is "rt_" for the HIP library; the
    test-only CPU restatement exports the same functions as "rto_"."""
    vp, i32, i64, f32p = C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_float)

    def fn(name):
        return getattr(L, prefix + name)

    fn("scene_create").restype = vp; fn("scene_create").argtypes = [C.POINTER(RtSceneDesc), C.c_int]
    fn("scene_destroy").argtypes = [vp]
    fn("scene_set_view").argtypes = [vp, C.POINTER(RtCamera), C.POINTER(RtOptions)]
    fn("trace_primary").argtypes = [vp, i32, i32, i32, i32, C.POINTER(i32), f32p]
    fn("trace_closest").argtypes = [vp, i64, f32p, f32p, C.POINTER(i32), f32p, f32p, f32p]
    fn("trace_any").argtypes = [vp, i64, f32p, f32p, C.POINTER(C.c_uint8)]
    fn("render").argtypes = [vp, C.POINTER(RtRenderParams), f32p, C.POINTER(RtStats)]
    fn("framebuffer_device").restype = vp; fn("framebuffer_device").argtypes = [vp]
    fn("resolve_rgba8").argtypes = [vp, C.POINTER(C.c_uint8)]
    fn("framebuffer_add").argtypes = [vp, vp]
    fn("framebuffer_gather").argtypes = [vp, C.POINTER(vp), i32]
    if hasattr(L, prefix + "trace_primary_static"):  # HIP library only
        fn("trace_primary_static").argtypes = [vp, i32, i32, i32, i32, C.POINTER(i32), f32p]
        fn("trace_primary_static_max_pixels").restype = i64
    if hasattr(L, prefix + "last_trace_kernel_ms"):  # HIP library only
        fn("last_trace_kernel_ms").restype = C.c_float; fn("last_trace_kernel_ms").argtypes = [vp]
        fn("scene_bvh_build_ms").restype = C.c_float; fn("scene_bvh_build_ms").argtypes = [vp]
    fn("framebuffer_read").argtypes = [vp, f32p]
    fn("device_count").restype = C.c_int
    fn("last_error").restype = C.c_char_p
    fn("version").restype = C.c_char_p
    return L


class RenderScene:
    """A scene uploaded through the render-loop ABI (HIP library, or any library exporting the same entry points)."""

    def __init__(self, lib, prefix, desc, device=0):
        self._L, self._p = lib, prefix
        self._f = lambda name: getattr(lib, prefix + name)
        self.width = desc.contents.options.width
        self.height = desc.contents.options.height
        self.num_prims = desc.contents.numPrims
        self._h = self._f("scene_create")(desc, device)
        if not self._h:
            raise RuntimeError(prefix + "scene_create: " + self._f("last_error")().decode())

    def close(self):
        if self._h:
            self._f("scene_destroy")(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != RT_OK:
            raise RuntimeError(f"{self._p}{what}: {self._f('last_error')().decode()}")

    def set_view(self, camera=None, options=None):
        self._check(self._f("scene_set_view")(self._h, None if camera is None else C.byref(camera),
                                              None if options is None else C.byref(options)), "scene_set_view")
        if options is not None:
            self.width, self.height = options.width, options.height

    def trace_primary(self, x0=0, y0=0, w=None, h=None):
        w = self.width if w is None else w
        h = self.height if h is None else h
        ids = np.empty((h, w), dtype=np.int32)
        t = np.empty((h, w), dtype=np.float32)
        self._check(self._f("trace_primary")(self._h, x0, y0, w, h, _ip(ids), _fp(t)), "trace_primary")
        return ids, t

    def trace_primary_static(self, x0=0, y0=0, w=None, h=None):
        """rt_trace_primary through the entry point that allocates nothing (static device scratch, bounded size)."""
        w = self.width if w is None else w
        h = self.height if h is None else h
        ids = np.empty((h, w), dtype=np.int32)
        t = np.empty((h, w), dtype=np.float32)
        self._check(self._f("trace_primary_static")(self._h, x0, y0, w, h, _ip(ids), _fp(t)), "trace_primary_static")
        return ids, t

    def trace_closest(self, origins, directions):
        o = np.ascontiguousarray(origins, dtype=np.float32); d = np.ascontiguousarray(directions, dtype=np.float32)
        n = o.shape[0]
        ids = np.empty(n, dtype=np.int32); t = np.empty(n, dtype=np.float32)
        pt = np.empty((n, 3), dtype=np.float32); nm = np.empty((n, 3), dtype=np.float32)
        self._check(self._f("trace_closest")(self._h, n, _fp(o), _fp(d), _ip(ids), _fp(t), _fp(pt), _fp(nm)), "trace_closest")
        return ids, t, pt, nm

    def trace_any(self, origins, directions):
        o = np.ascontiguousarray(origins, dtype=np.float32); d = np.ascontiguousarray(directions, dtype=np.float32)
        n = o.shape[0]
        occ = np.empty(n, dtype=np.uint8)
        self._check(self._f("trace_any")(self._h, n, _fp(o), _fp(d), occ.ctypes.data_as(C.POINTER(C.c_uint8))), "trace_any")
        return occ

    def render(self, rng_mode=RT_RNG_COUNTER, seed=56, part_index=0, part_count=1, tile=(32, 32), samples_per_pass=0,
               collect_counters=False, skip_exposure=False, download=True, out_ptr=None, first_sample=0, sample_count=0,
               keep_accumulator=False, adaptive_threshold=0.0, adaptive_batch=0, adaptive_min_samples=0):
        """out_ptr: optional raw pointer (host or device memory of width*height*4 floats) that receives the frame
        instead of a fresh numpy array. first_sample / sample_count / keep_accumulator: progressive rendering;
        adaptive_*: adaptive sampling (see RtRenderParams in include/rt_amd.h)."""
        p = RtRenderParams(rng_mode, seed, part_index, part_count, tile[0], tile[1], samples_per_pass,
                           int(collect_counters), int(skip_exposure), first_sample, sample_count, int(keep_accumulator),
                           adaptive_threshold, adaptive_batch, adaptive_min_samples)
        st = RtStats()
        out = None
        if out_ptr is not None:
            dst = C.cast(C.c_void_p(out_ptr), C.POINTER(C.c_float))
        elif download:
            out = np.empty((self.height, self.width, 4), dtype=np.float32)
            dst = _fp(out)
        else:
            dst = None
        self._check(self._f("render")(self._h, C.byref(p), dst, C.byref(st)), "render")
        return out, st

    def framebuffer_device(self):
        return self._f("framebuffer_device")(self._h)

    def framebuffer_add(self, other):
        """Adds `other`'s last frame to this scene's (device to device): the gather of disjoint tile partitions."""
        self._check(self._f("framebuffer_add")(self._h, other._h), "framebuffer_add")

    def framebuffer_gather(self, others):
        """Adds the last frames of `others` (disjoint tile partitions) to this scene's in one call."""
        arr = (C.c_void_p * len(others))(*[o._h for o in others])
        self._check(self._f("framebuffer_gather")(self._h, arr, len(others)), "framebuffer_gather")

    def bvh_build_ms(self):
        return float(self._f("scene_bvh_build_ms")(self._h))

    def last_trace_kernel_ms(self):
        return float(self._f("last_trace_kernel_ms")(self._h))

    def framebuffer_read(self):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(self._f("framebuffer_read")(self._h, _fp(out)), "framebuffer_read")
        return out

    def resolve_rgba8(self):
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        self._check(self._f("resolve_rgba8")(self._h, out.ctypes.data_as(C.POINTER(C.c_uint8))), "resolve_rgba8")
        return out


_device = None


def load_device_library():
    """The HIP library. Raises if it is missing or cannot be loaded: there is no CPU fallback."""
    global _device
    if _device is not None:
        return _device
    path = os.environ.get("RT_AMD_DEVICE_LIB") or os.path.join(_HERE, "libraytracer_amd.so")  # override: kernel experiments
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `make hip` (or __graft_entry__.build()); there is no CPU fallback")
    _device = bind_render_abi(C.CDLL(path), "rt_")
    return _device


def device_scene(desc, device=0, build_on_device=False):
    """Uploads the scene. build_on_device: hand it over WITHOUT the host-built tree (nodes = NULL); the library then builds
    its own BVH on the GPU from the object boxes (RtSceneDesc.primAabbMin / primAabbMax)."""
    if build_on_device:
        d = RtSceneDesc()
        C.pointer(d)[0] = desc.contents
        d.nodes = None
        d.numNodes = 0
        desc = C.pointer(d)
    return RenderScene(load_device_library(), "rt_", desc, device)
