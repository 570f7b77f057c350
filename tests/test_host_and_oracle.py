"""CPU-side tests: the host scene library and the CPU restatement (oracle) against fixtures generated from the
compiled reference (tests/golden/*.npz, made by tools/make_golden.py), plus ABI export checks. No GPU needed."""
import ctypes as C
import hashlib
import os
import re
import subprocess

import numpy as np
import pytest

from tests.conftest import REPO, SMALL_SCENES, psnr
from raytracer_amd import capi
from tests.oracle_binding import oracle_scene


def sha1(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", SMALL_SCENES + ["mesh70k", "readme_mid"])
def test_host_bvh_matches_reference(name, scene_path, golden):
    """Same node array and same object permutation as BVH::BVH (src/acceleration/BVH.cpp:161-182) on the same file."""
    g = golden(name)
    hs = capi.HostScene.load(scene_path(name)).build()
    d = hs.desc.contents
    assert d.numNodes == int(g["num_nodes"]) and d.numPrims == int(g["num_prims"])
    assert sha1(hs.nodes()) == str(g["bvh_nodes_sha1"])
    assert sha1(hs.prim_orig_index()) == str(g["bvh_perm_sha1"])
    assert f"BVH created with {d.numNodes} nodes on {d.numPrims} objects." in hs.log


@pytest.mark.parametrize("name", SMALL_SCENES)
def test_oracle_primary_hits_bit_exact(name, scene_path, golden):
    """Hit ids and the bits of t equal what Scene::findClosestObject returns in the reference, for every pixel."""
    g = golden(name)
    osc = oracle_scene(capi.HostScene.load(scene_path(name)).build().desc)
    ids, t = osc.trace_primary()
    assert np.array_equal(ids, g["hit_ids"])
    assert np.array_equal(t.view(np.uint32), g["hit_tbits"])


def test_oracle_primary_hits_config2_hash(scene_path, golden):
    """Config 2 (1080x1080, ~70k triangles): all 1,166,400 ids and t bit patterns, by hash."""
    g = golden("mesh70k")
    osc = oracle_scene(capi.HostScene.load(scene_path("mesh70k")).build().desc)
    ids, t = osc.trace_primary()
    assert int((ids >= 0).sum()) == int(g["num_hits"])
    assert sha1(ids) == str(g["hit_ids_sha1"])
    assert sha1(t.view(np.uint32)) == str(g["hit_tbits_sha1"])


def test_full_size_benchmark_scene_host_and_oracle(scene_path, golden):
    """The benchmark scene at full size (~1.05 M objects, 665 k nodes, 1080x1080): the host's parse + BVH build equal
    the reference's node array and object order, and the oracle's primary hits equal the reference's, by hash."""
    g = golden("readme")
    hs = capi.HostScene.load(scene_path("readme")).build()
    d = hs.desc.contents
    assert (d.numNodes, d.numPrims) == (int(g["num_nodes"]), int(g["num_prims"]))
    assert sha1(hs.nodes()) == str(g["bvh_nodes_sha1"])
    assert sha1(hs.prim_orig_index()) == str(g["bvh_perm_sha1"])
    osc = oracle_scene(hs.desc)
    ids, t = osc.trace_primary()
    assert int((ids >= 0).sum()) == int(g["num_hits"])
    assert sha1(ids) == str(g["hit_ids_sha1"]) and sha1(t.view(np.uint32)) == str(g["hit_tbits_sha1"])
    osc.close()


@pytest.mark.parametrize("name", SMALL_SCENES)
def test_oracle_render_replays_reference_t1(name, scene_path, golden):
    """ref-mt19937 mode: the float framebuffer of the reference at -t1, bit for bit, and the PNG bytes it would save."""
    g = golden(name)
    osc = oracle_scene(capi.HostScene.load(scene_path(name)).build().desc)
    img, st = osc.render(rng_mode=capi.RT_RNG_REF_MT19937)
    assert np.array_equal(img.view(np.uint32), g["render_f32"].view(np.uint32))
    assert np.array_equal(osc.resolve_rgba8(), g["render_rgba8"])
    assert st.raysClosest > 0


@pytest.mark.parametrize("name", ["spiral", "tenthousand"])
def test_oracle_on_the_scenes_the_reference_ships(name, golden):
    """example_scenes/scene_files/spiral.txt (5000 spheres, two suns) and tenthousand.txt (10 000 spheres, depth of field,
    16 bounces) of the reference, carried as arrays (tests/desc_fixture.py): primary hits at the scenes' own 800 x 800
    equal the compiled reference's (ids and bits of t, by hash and on a band of rows), and the -t1 frame at a reduced size
    equals the reference's float frame and PNG bytes."""
    from tests.desc_fixture import DescFromArrays
    g = golden("bundled_" + name)
    w, h = int(g["width"]), int(g["height"])
    full = DescFromArrays(g)
    osc = oracle_scene(full.desc)
    ids, t = osc.trace_primary()
    assert sha1(ids) == str(g["hit_ids_sha1"]) and sha1(t.view(np.uint32)) == str(g["hit_tbits_sha1"])
    r0, r1 = (int(x) for x in g["band_rows"])
    assert np.array_equal(ids.reshape(h, w)[r0:r1].ravel(), g["band_ids"])
    assert np.array_equal(t.view(np.uint32).reshape(h, w)[r0:r1].ravel(), g["band_tbits"])
    assert int((ids >= 0).sum()) == int(g["num_hits"])
    osc.close()
    side = int(g["small_side"])
    small = DescFromArrays(g, side, side)
    osc = oracle_scene(small.desc)
    ids, t = osc.trace_primary()
    assert np.array_equal(ids.ravel(), g["small_hit_ids"]) and np.array_equal(t.view(np.uint32).ravel(), g["small_hit_tbits"])
    img, _ = osc.render(rng_mode=capi.RT_RNG_REF_MT19937)
    assert np.array_equal(img.view(np.uint32), g["small_render_f32"].view(np.uint32))
    assert np.array_equal(osc.resolve_rgba8(), g["small_render_rgba8"])
    osc.close()


def test_oracle_counter_mode_partition_invariant(scene_path):
    """Counter RNG: the image does not depend on the tile partition or the thread count."""
    hs = capi.HostScene.load(scene_path("materials")).build()
    full, _ = oracle_scene(hs.desc, threads=3).render(seed=7)
    acc = np.zeros_like(full)
    for part in range(3):
        img, _ = oracle_scene(hs.desc, threads=1 + part).render(seed=7, part_index=part, part_count=3, tile=(16, 8))
        assert np.all((acc == 0) | (img == 0))
        acc += img
    assert np.array_equal(acc.view(np.uint32), full.view(np.uint32))


def test_oracle_counter_mode_converges_to_reference_stream(scene_path, golden):
    """Different random streams, same estimator: the counter-mode image is close to the -t1 image (loose bound;
    it only guards against a broken generator)."""
    g = golden("config1")
    img, _ = oracle_scene(capi.HostScene.load(scene_path("config1")).build().desc).render(seed=1)
    assert psnr(img[..., :3], g["render_f32"][..., :3]) > 20.0


def test_png_decoder_matches_reference_on_every_png_kind(tmp_path, golden):
    """All colour types and bit depths, plain and Adam7-interlaced, the five row filters, palettes, tRNS, several IDAT
    chunks: what PNG::readFromFile leaves in memory in the reference (hashes from the compiled reference)."""
    from tools import png_variants
    g = golden("png_variants")
    names = png_variants.write_all(str(tmp_path))
    assert names == [str(n) for n in g["names"]]
    for n, want in zip(names, g["decoded"]):
        img = capi.png_read_linear(str(tmp_path / n))
        assert f"{img.shape[1]}x{img.shape[0]}:{sha1(img)}" == str(want), n


def test_png_pixels_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    rgba8 = rng.integers(0, 256, size=(37, 53, 4), dtype=np.uint8)
    p = str(tmp_path / "t.png")
    capi.png_write_rgba8(p, rgba8)
    assert np.array_equal(capi.png_read_rgba8(p), rgba8)
    lin = capi.png_read_linear(p)
    # PNG::readFromFile: byte/255 then sRGB->linear on r,g,b (src/image/PNG.cpp:137-144, 14-18)
    c = rgba8[..., :3].astype(np.float32) / np.float32(255)
    cd = c.astype(np.float64)
    ref = np.where(cd < 0.04045, cd / 12.92, ((cd + 0.055) / 1.055) ** 2.4).astype(np.float32)
    assert np.allclose(lin[..., :3], ref, rtol=0, atol=1e-7)
    assert np.array_equal(lin[..., 3], rgba8[..., 3].astype(np.float32) / np.float32(255))
    # linear -> sRGB8 (PNG::saveToFile) inverts it exactly on 8-bit data
    assert np.array_equal(capi.to_srgb8(lin), rgba8)


def test_png_written_in_parallel_pieces_is_one_valid_stream(tmp_path):
    """The encoder deflates the image in independent 512 KiB pieces on the host threads and joins them into one zlib
    stream: an image of several pieces (and sizes around the piece boundary) must decode — with Python's zlib, and
    with the library's own decoder — to the bytes that went in, with every chunk CRC right."""
    import struct
    import zlib
    rng = np.random.default_rng(5)
    for w, h in ((700, 600), (1, 1), (363, 361), (131071, 1), (512, 256)):
        rgba8 = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
        rgba8[:, : w // 2] = 9  # one half compresses, the other does not
        p = str(tmp_path / f"p{w}x{h}.png")
        capi.png_write_rgba8(p, rgba8)
        data = open(p, "rb").read()
        assert data[:8] == b"\x89PNG\r\n\x1a\n"
        pos, idat, kinds = 8, b"", []
        while pos < len(data):
            n, = struct.unpack(">I", data[pos:pos + 4])
            kind, body = data[pos + 4:pos + 8], data[pos + 8:pos + 8 + n]
            assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(kind + body) & 0xFFFFFFFF
            kinds.append(kind)
            if kind == b"IDAT":
                idat += body
            pos += 12 + n
        assert kinds[0] == b"IHDR" and kinds[-1] == b"IEND"
        rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w * 4 + 1)
        assert (rows[:, 0] == 0).all() and np.array_equal(rows[:, 1:].reshape(h, w, 4), rgba8)
        assert np.array_equal(capi.png_read_rgba8(p), rgba8)


def _png_with_ihdr(path, width, height, depth, ctype, compression=0, flt=0, interlace=0):
    import struct
    import zlib

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)
    ihdr = struct.pack(">IIBBBBB", width & 0xFFFFFFFF, height & 0xFFFFFFFF, depth, ctype, compression, flt, interlace)
    body = zlib.compress(b"\0" * 64)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", body) + chunk(b"IEND", b""))


def test_malformed_png_headers_are_errors_not_crashes(tmp_path):
    """Bit depths outside the PNG table (0 would divide by zero, 3 or 32 would shift out of range), unknown colour
    types, non-zero compression/filter methods and absurd sizes: a decoder error like LodePNG's, and a scene that
    names such a texture carries on with an empty image like Scene::loadTexture (src/scene/raytracer.h) does."""
    bad = [dict(depth=0, ctype=0), dict(depth=3, ctype=0), dict(depth=32, ctype=2), dict(depth=4, ctype=2), dict(depth=16, ctype=3),
           dict(depth=2, ctype=6), dict(depth=8, ctype=5), dict(depth=8, ctype=6, compression=1), dict(depth=8, ctype=6, flt=1),
           dict(depth=8, ctype=6, interlace=2)]
    for i, kw in enumerate(bad):
        p = str(tmp_path / f"bad{i}.png")
        _png_with_ihdr(p, 4, 4, **kw)
        with pytest.raises(RuntimeError, match="PNG"):
            capi.png_read_linear(p)
        with pytest.raises(RuntimeError, match="PNG"):
            capi.png_read_rgba8(p)
    for w, h in ((0, 4), (4, 0), (0x7FFFFFFF, 0x7FFFFFFF), (0xFFFFFFFF, 1)):
        p = str(tmp_path / "size.png")
        _png_with_ihdr(p, w, h, 8, 6)
        with pytest.raises(RuntimeError, match="PNG"):
            capi.png_read_linear(p)
    p = str(tmp_path / "bad0.png")
    hs = capi.HostScene.new(8, 8)
    L = capi.load_host_library()
    idx = L.rth_scene_add_texture_file(hs._h, p.encode())
    assert idx >= 0  # an empty texture, logged, not an exception across the C boundary
    assert any("PNG" in line for line in hs.log)


def test_scene_errors(tmp_path):
    with pytest.raises(RuntimeError, match="Couldn't open file"):
        capi.HostScene.load(str(tmp_path / "missing.sdml"))
    p = tmp_path / "x.foo"; p.write_text("png 1 1 a.png\n")
    with pytest.raises(RuntimeError, match="Unrecognized file format"):
        capi.HostScene.load(str(p))
    p = tmp_path / "bad.txt"; p.write_text("jpg 4 4 a.png\n")
    with pytest.raises(RuntimeError, match="Expected PNG image type"):
        capi.HostScene.load(str(p))
    p = tmp_path / "nowidth.sdml"; p.write_text("<Scene options={height: 4; filename: a.png}>\n</Scene>\n")
    with pytest.raises(RuntimeError, match="width not provided"):
        capi.HostScene.load(str(p))
    # a material without lobes is refused (the reference would index bdfs[-1])
    p = tmp_path / "reflective.txt"; p.write_text("png 4 4 a.png\nreflective\nsphere 0 0 -2 1\n")
    with pytest.raises(RuntimeError, match="no BSDF lobes"):
        capi.HostScene.load(str(p)).build()


@pytest.mark.parametrize("body,needle", [
    ("v 0 0 0\nv 1 0\nv 0 1 0\nf 1 2 3\n", "malformed"),          # a coordinate is missing
    ("v 0 0 0\nv 1 x 0\nv 0 1 0\nf 1 2 3\n", "malformed"),        # not a number
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 q\n", "malformed"),        # not an index
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 1 2\n", "malformed"),        # index 0 (the reference's at(-1) throws)
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf -1 -2 -3\n", "malformed"),     # negative indices are not supported
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1//1 2//2 3//3\n", "malformed"),  # normals named but none given
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1//1 2 3\nvn 0 0 1\n", "malformed"),  # first corner has a normal, the others do not
    ("v 1e-60 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n", "malformed"),    # std::stof reports underflow as out of range
])
def test_obj_loader_rejects_what_the_reference_aborts_on(tmp_path, body, needle):
    """Where the reference dies on an uncaught std::stof / std::stoi / at() exception (src/parser/parser.cpp:47-142),
    the host library returns an error."""
    (tmp_path / "m.obj").write_text(body)
    (tmp_path / "s.sdml").write_text(
        "<Scene options={width: 8; height: 8; filename: o.png}>\n  <Camera options={}/>\n"
        f"  <Wavefront path=\"{tmp_path / 'm.obj'}\" options={{center: (0, 0, -3); scale: 1}}>\n  </Wavefront>\n</Scene>\n")
    with pytest.raises(RuntimeError, match=needle):
        capi.HostScene.load(str(tmp_path / "s.sdml"))


def test_obj_loader_on_a_file_of_many_pieces_reports_what_a_top_to_bottom_scan_would(tmp_path):
    """The OBJ text is scanned in pieces by the host threads. A file long enough for several pieces gives the same mesh
    with any thread count, and with errors in several pieces the one a sequential two-pass reader meets first is
    reported: any `v` error before any `f` error, and of two errors in the same pass the one higher up in the file."""
    n = 40000  # ~1.5 MB of text: more than twenty pieces of 64 KiB
    rng = np.random.default_rng(2)
    pts = rng.normal(size=(n, 3))
    lines = [f"v {x:.6f} {y:.6f} {z:.6f}" for x, y, z in pts]
    faces = [f"f {i + 1} {(i + 7) % n + 1} {(i + 13) % n + 1}" for i in range(n)]
    sdml = ("<Scene options={width: 8; height: 8; filename: o.png}>\n  <Camera options={}/>\n"
            f"  <Wavefront path=\"{tmp_path / 'm.obj'}\" options={{center: (0, 0, -3); scale: 1}}>\n  </Wavefront>\n</Scene>\n")
    (tmp_path / "s.sdml").write_text(sdml)

    def load(vlines, flines, threads):
        (tmp_path / "m.obj").write_text("\n".join(vlines + flines) + "\n")
        capi.set_host_threads(threads)
        try:
            return capi.HostScene.load(str(tmp_path / "s.sdml"))
        finally:
            capi.set_host_threads(0)

    hashes = []
    for threads in (1, 3, 16):
        hs = load(lines, faces, threads).build()
        d = hs.desc.contents
        assert d.numPrims == n
        hashes.append((sha1(np.ctypeslib.as_array(d.primHit, shape=(d.numPrims * 12,)).copy()), sha1(hs.nodes())))
    assert hashes[0] == hashes[1] == hashes[2]
    # a face error near the top, a vertex error near the bottom: the vertex scan comes first ("stof", not a range check)
    bad_v = list(lines); bad_v[n - 5] = "v 1 x 0"
    bad_f = list(faces); bad_f[3] = "f 0 1 2"
    with pytest.raises(RuntimeError, match="stof"):
        load(bad_v, bad_f, 16)
    # two vertex errors: a missing coordinate high up (range check), not a number far below (stof)
    bad_v = list(lines); bad_v[10] = "v 1 2"; bad_v[n - 5] = "v 1 x 0"
    with pytest.raises(RuntimeError, match="range_check"):
        load(bad_v, faces, 16)
    bad_v = list(lines); bad_v[10] = "v 1 x 0"; bad_v[n - 5] = "v 1 2"
    with pytest.raises(RuntimeError, match="stof"):
        load(bad_v, faces, 16)


def test_host_thread_count_never_changes_the_result(scene_path):
    """Parsing, triangle set-up and the BVH build give the same bytes with 1 and with many host threads."""
    out = []
    for n in (1, 7):
        capi.set_host_threads(n)
        assert capi.host_threads() == n
        hs = capi.HostScene.load(scene_path("readme_mid")).build()
        d = hs.desc.contents
        hit = np.ctypeslib.as_array(d.primHit, shape=(d.numPrims * 12,)).copy()
        out.append((sha1(hs.nodes()), sha1(hs.prim_orig_index()), sha1(hit)))
    capi.set_host_threads(0)
    assert out[0] == out[1]


def test_degenerate_mesh_builds_without_undefined_behaviour():
    """A mesh whose vertices coincide: the OBJ scaling divides by a zero extent (src/parser/parser.cpp:94), every
    triangle is NaN, and the reference's BVH build converts a NaN to a bin index (undefined: an out-of-bounds write on
    x86). Here the build is defined, the objects stay in the tree and no ray hits them."""
    hs = capi.HostScene.new(16, 16)
    m = hs.add_material(capi.RT_MAT_DEFAULT)
    hs.add_mesh(np.zeros((3, 3)), None, np.array([[0, 1, 2]] * 9), material=m)
    for i in range(6):
        hs.add_sphere((i - 2.5, 0, -4), 0.4, material=m)
    hs.add_light(capi.RT_LIGHT_POINT, color=(5, 5, 5, 1), v=(0, 3, 0))
    hs.set_camera(); hs.set_options(16, 16, numRays=2)
    hs.build()
    d = hs.desc.contents
    assert d.numPrims == 15 and d.numNodes >= 1
    osc = oracle_scene(hs.desc)
    ids, t = osc.trace_primary()
    assert (ids >= 0).any() and not np.isin(ids[ids >= 0], np.arange(9)).any()  # only the spheres (inserted after the mesh) are hit
    img, _ = osc.render(seed=1)
    assert np.isfinite(img).all()
    osc.close()


def test_empty_scene_and_planes_only(tmp_path):
    p = tmp_path / "empty.txt"; p.write_text("png 8 8 a.png\n")
    hs = capi.HostScene.load(str(p)).build()
    assert hs.desc.contents.numPrims == 0 and hs.desc.contents.numNodes == 0
    osc = oracle_scene(hs.desc)
    ids, t = osc.trace_primary()
    assert np.all(ids == -1) and np.all(np.isinf(t))
    img, _ = osc.render()
    assert np.all(img == 0)  # no light, no hit: transparent black everywhere
    p = tmp_path / "plane.txt"; p.write_text("png 8 8 a.png\nsun 0 1 0\nplane 0 1 0 1\n")
    osc = oracle_scene(capi.HostScene.load(str(p)).build().desc)
    ids, _ = osc.trace_primary()
    assert set(np.unique(ids)) == {-1, 0}  # plane k is reported as numPrims + k


def _declared(header):
    text = open(os.path.join(REPO, "include", header)).read()
    return sorted(set(re.findall(r"\b(rth?_[a-z0-9_]+)\s*\(", text)))


def test_oracle_progressive_and_adaptive_sampling(scene_path):
    """The extended sample loop (RtRenderParams.firstSample / sampleCount / keepAccumulator and adaptive*) in the CPU
    restatement: sample windows added up give the frame of one call bit for bit; the estimate after a partial window is
    the frame of a scene with that many samples except for alpha's denominator; adaptive sampling that may not stop before the
    last sample changes nothing, and with a real threshold takes fewer samples and stays close to the full frame."""
    hs = capi.HostScene.load(scene_path("mesh1k_gi")).build()
    orc = oracle_scene(hs.desc)
    full, st = orc.render(seed=7)
    n = hs.desc.contents.options.numRays
    assert n >= 4
    a, sa = orc.render(seed=7, first_sample=0, sample_count=n // 2)
    b, sb = orc.render(seed=7, first_sample=n // 2, keep_accumulator=True)
    assert sa.cameraSamples + sb.cameraSamples == st.cameraSamples
    assert np.array_equal(b.view(np.uint32), full.view(np.uint32))
    assert not np.array_equal(a.view(np.uint32), full.view(np.uint32))
    with pytest.raises(RuntimeError):
        orc.render(seed=7, first_sample=n)
    same, ss = orc.render(seed=7, adaptive_threshold=0.5, adaptive_batch=2, adaptive_min_samples=n)  # no pixel may stop before the end
    assert ss.cameraSamples == st.cameraSamples and np.array_equal(same.view(np.uint32), full.view(np.uint32))
    few, sf = orc.render(seed=7, adaptive_threshold=0.2, adaptive_batch=2)
    assert sf.cameraSamples < st.cameraSamples
    assert psnr(few[..., :3], full[..., :3]) >= 20.0
    orc.close()


REF_BINDING = os.path.join(REPO, "oracle", "_ref", "ref_binding")


@pytest.mark.skipif(not os.path.exists(REF_BINDING), reason="oracle/_ref/ref_binding is built only where the reference sources exist (make -C oracle ref)")
@pytest.mark.parametrize("name", ["config1", "materials", "mesh1k_gi", "spheres_txt_dof", "readme_small"])
def test_reference_side_binding_reproduces_the_reference(name, scene_path):
    """oracle/ref_binding.cpp is the binding of include/rt_amd.h a maintainer of the reference would add, compiled against
    the unmodified reference: it fills RtSceneDesc from the reference's own Scene members and calls the ABI. Through the CPU
    restatement it must give the reference's own primary hits (Scene::findClosestObject) and, replaying the reference's
    random stream, the frame of Scene::render(1), bit for bit."""
    path = scene_path(name)
    r = subprocess.run([REF_BINDING, os.path.basename(path), os.path.join(REPO, "oracle", "librt_oracle.so"), "rto_", "render"],
                       cwd=os.path.dirname(path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "BINDING OK" in r.stdout, r.stdout[-600:] + r.stderr[-600:]
    assert ", 0 mismatches" in r.stdout and ", 0 differ" in r.stdout


def test_libraries_export_every_declared_symbol():
    """Both shipped libraries load without a GPU and export exactly what include/*.h declares."""
    host = C.CDLL(os.path.join(REPO, "raytracer_amd", "librt_host.so"))
    for sym in _declared("rt_host.h"):
        assert hasattr(host, sym), sym
    dev = capi.load_device_library()
    names = [s for s in _declared("rt_amd.h") if s.startswith("rt_")]
    assert "rt_render" in names and "rt_trace_primary" in names
    for sym in names:
        assert hasattr(dev, sym), sym
    assert b"gfx950" in dev.rt_version()
    # the oracle mirrors the same entry points under rto_
    from tests.oracle_binding import load_oracle_library
    orc = load_oracle_library()
    # entry points about the device itself (kernel timing, the static-scratch variant) have no CPU mirror
    device_only = {"rt_last_trace_kernel_ms", "rt_trace_primary_static", "rt_trace_primary_static_max_pixels", "rt_scene_bvh_build_ms"}
    for sym in names:
        if sym not in device_only:
            assert hasattr(orc, "rto_" + sym[3:]), sym


def test_device_library_does_not_link_the_oracle():
    out = subprocess.run(["ldd", os.path.join(REPO, "raytracer_amd", "libraytracer_amd.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out and "amdhip64" in out


def test_philox_known_answers_and_stream_injectivity():
    """The counter-mode generator is Philox4x32-10: the three known-answer vectors of the Random123 distribution, and
    no two camera samples of a 1080 x 1080 x 256 frame share a stream (the counter (pixel, sample, block) is a
    bijection's input, so equal outputs need equal counters): checked on every sample of 4096 pixels spread over the
    frame and on 2^20 random (pixel, sample) pairs."""
    from tests.oracle_binding import load_oracle_library
    L = load_oracle_library()
    u32 = C.c_uint32

    def block(key, ctr):
        out = (u32 * 4)()
        L.rto_philox4x32_10((u32 * 2)(*key), (u32 * 4)(*ctr), out)
        return list(out)
    assert block([0, 0], [0, 0, 0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert block([0xffffffff] * 2, [0xffffffff] * 4) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert block([0xa4093822, 0x299f31d0], [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    # the first block of a stream, as 4 x 24-bit draws, for many (pixel, sample) pairs: all different
    L.rto_counter_draws.argtypes = [u32, u32, u32, C.c_int32, C.POINTER(C.c_float)]
    rng = np.random.default_rng(1)
    npix = 1080 * 1080
    pixels = np.concatenate([np.repeat(np.linspace(0, npix - 1, 4096).astype(np.int64), 256), rng.integers(0, npix, 1 << 20)])
    samples = np.concatenate([np.tile(np.arange(256), 4096), rng.integers(0, 256, 1 << 20)])
    pairs = np.unique(np.stack([pixels, samples], 1), axis=0)
    # vectorised restatement of the block function (same constants) so that two million streams take a second
    def philox(c0, c1, c2, c3, k0, k1):
        c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) for x in (c0, c1, c2, c3))
        m32 = np.uint64(0xffffffff)
        for _ in range(10):
            p0 = np.uint64(0xD2511F53) * c0; p1 = np.uint64(0xCD9E8D57) * c2
            n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0); n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
            c0, c1, c2, c3 = n0, p1 & m32, n2, p0 & m32
            k0 = (k0 + 0x9E3779B9) & 0xffffffff; k1 = (k1 + 0xBB67AE85) & 0xffffffff
        return np.stack([c0, c1, c2, c3], 1).astype(np.uint32)
    words = philox(pairs[:, 0], pairs[:, 1], np.zeros(len(pairs)), np.zeros(len(pairs)), 56, 0)
    draws = (words >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)
    for i in (0, 1, len(pairs) // 2, len(pairs) - 1):  # the vectorised form is the oracle's generator
        got = (C.c_float * 4)()
        L.rto_counter_draws(56, int(pairs[i, 0]), int(pairs[i, 1]), 4, got)
        assert list(got) == list(draws[i])
    assert len(np.unique(words, axis=0)) == len(pairs), "two camera samples share a random stream"
    assert len(np.unique(draws, axis=0)) == len(pairs)


def test_host_parallel_loops_cover_every_index_and_hand_exceptions_to_the_caller(tmp_path):
    """The loops every parallel step of the host library runs on (raytracer_amd/host/rth_threads.h): each index exactly
    once at 1, 2, 8 and 16 threads, chunk numbers in data order, and an exception thrown on a worker thread arrives on
    the calling thread (instead of ending the process) once all workers have stopped."""
    exe = str(tmp_path / "threads_check")
    libdir = os.path.join(REPO, "raytracer_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-I" + os.path.join(libdir, "host"),
                    os.path.join(REPO, "tests", "cpp", "threads_check.cpp"), "-L" + libdir, "-lrt_host",
                    "-Wl,-rpath," + libdir, "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_clustering_steps_of_the_gpu_bvh_build_on_the_cpu(tmp_path):
    """raytracer_amd/csrc/rt_ploc.h holds the per-element steps of the GPU BVH build (partner search, merge, counts, places);
    tests/cpp/ploc_check.cpp drives the same functions from plain loops under the address and undefined-behaviour
    sanitizers: one object, two, five, thousands of identical boxes, random soups, boxes growing geometrically, NaN and
    infinite boxes. Every object gets one place, every record one index, boxes hold their children's, every round merges."""
    exe = str(tmp_path / "ploc_check")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                    "-I" + os.path.join(REPO, "raytracer_amd", "csrc"), os.path.join(REPO, "tests", "cpp", "ploc_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]


def test_flattened_scene_survives_the_trip_through_files(tmp_path, scene_path):
    """raytracer_amd/desc_io.py: the arrays one process flattens are what another maps — same records, so the CPU
    restatement renders the same frame and finds the same hits from either description (textures, texture coordinates
    and planes included)."""
    from raytracer_amd import desc_io
    from tests.oracle_binding import oracle_scene
    hs = capi.HostScene.load(scene_path("readme_small")).build()
    desc_io.save_desc(hs.desc, str(tmp_path / "flat"))
    back = desc_io.load_desc(str(tmp_path / "flat"))
    a, b = hs.desc.contents, back.desc.contents
    assert (a.numPrims, a.numNodes, a.numPlanes, a.numLights, a.numMaterials, a.numTextures) == \
           (b.numPrims, b.numNodes, b.numPlanes, b.numLights, b.numMaterials, b.numTextures)
    assert a.numTextures > 0 and bool(a.primTexCoords) == bool(b.primTexCoords)
    one, two = oracle_scene(hs.desc, threads=2), oracle_scene(back.desc, threads=2)
    ids1, t1 = one.trace_primary(); ids2, t2 = two.trace_primary()
    assert np.array_equal(ids1, ids2) and np.array_equal(t1.view(np.uint32), t2.view(np.uint32))
    f1, _ = one.render(seed=5); f2, _ = two.render(seed=5)
    assert np.array_equal(np.asarray(f1).view(np.uint32), np.asarray(f2).view(np.uint32))
    one.close(); two.close()


def test_redchair_tree_is_buildable_and_consistent(golden):
    """redchair.txt of the reference (fisheye, flat `trif` quads) as arrays: the reference's own BVH build dies on it
    (src/acceleration/BVH.cpp:93-99, undefined bin index when a node's centroids coincide on an axis), so nothing pins the
    tree; the host library's tree for it must at least be a tree over all objects whose boxes hold their children's, and
    the CPU restatement must find, through it, exactly the hits a loop over all triangles finds for rays of unit length
    or longer (for those the reference's culling is conservative)."""
    from tests.desc_fixture import DescFromArrays
    g = golden("bundled_redchair")
    nodes = np.asarray(g["nodes"]).reshape(-1, 8)
    boxes = nodes[:, :6].view(np.float32)
    n = int(np.asarray(g["primKind"]).shape[0])
    covered = np.zeros(n, bool)
    for i, row in enumerate(nodes):
        if row[7] > 0:
            assert not covered[row[6]:row[6] + row[7]].any()
            covered[row[6]:row[6] + row[7]] = True
        else:
            for c in (i + 1, int(row[6])):
                assert np.all(boxes[c, :3] >= boxes[i, :3]) and np.all(boxes[c, 3:] <= boxes[i, 3:])
    assert covered.all()
    full = DescFromArrays(g, width=96, height=96)
    orc = oracle_scene(full.desc, threads=4)
    flat = DescFromArrays({**{k: g[k] for k in g.files}, "nodes": np.array([[0, 0, 0, 0, 0, 0, 0, n]], np.int32)}, width=96, height=96)
    fn = np.asarray(flat._keep["nodes"]).view(np.float32)
    fn[0, :3] = boxes[0, :3]; fn[0, 3:6] = boxes[0, 3:]
    brute = oracle_scene(flat.desc, threads=4)  # one leaf holding every object: no box decides anything
    rng = np.random.default_rng(3)
    o = (rng.normal(size=(4000, 3)) * 2 + np.array([0, 1, -2])).astype(np.float32)
    d = rng.normal(size=(4000, 3)).astype(np.float32)
    d *= ((1.0 + rng.random((4000, 1))) / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)  # lengths 1 .. 2
    a, b = orc.trace_closest(o, d), brute.trace_closest(o, d)
    assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and (a[0] >= 0).sum() > 400
    orc.close(); brute.close()


def test_scene_stream_of_the_randomised_campaign_builds_and_renders_on_the_cpu():
    """The generator behind tools/fuzz_parity.py and the GPU suite's fixed slice, without a GPU: every family of the
    stream builds on the host, and the CPU restatement renders it to a finite frame. The far-plane family must really put
    lit geometry millions of units away (that is what makes its visibility queries wander through the tree), with direct
    light only."""
    from tools import fuzz_parity
    rng = np.random.default_rng(5)
    seen = {}
    for k in range(15):
        hs, rays_of, family = fuzz_parity.scene_of_the_stream(rng, k)
        o, d = rays_of(rng, 500)
        assert o.shape == d.shape == (500, 3) and o.dtype == d.dtype == np.float32
        seen[family] = seen.get(family, 0) + 1
        orc = oracle_scene(hs.desc, threads=4)
        try:
            ref, st = orc.render(seed=k)
            ids, t, pt, nm = orc.trace_closest(o, d)
        finally:
            orc.close()
        dsc = hs.desc.contents
        assert ref.shape == (dsc.options.height, dsc.options.width, 4)
        assert np.isfinite(ref).all(), f"scene {k} ({family})"
        if family == "far":
            assert dsc.options.maxBounces == 1 and dsc.numPlanes >= 1
            far = max(abs(float(v)) for v in dsc.planes[0].point)
            assert far > 5e3
            assert all(dsc.lights[i].type != capi.RT_LIGHT_ENVIRONMENT for i in range(dsc.numLights))
            assert (ref[..., :3].max(-1) > 0).mean() > 0.02  # the bulb reaches the plane
    assert seen == {"soup": 9, "far": 3, "flat": 3}
