"""The command-line front-end (raytracer_amd/raytracer): argument handling and exit codes of the reference's
src/main.cpp:8-37 without a GPU; one whole run (scene file -> PNG) on the GPU box."""
import os
import subprocess

import numpy as np
import pytest

from raytracer_amd import capi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(REPO, "raytracer_amd", "raytracer")


def run(*args, cwd=None):
    return subprocess.run([CLI, *args], cwd=cwd, capture_output=True, text=True, timeout=300)


def test_cli_argument_errors(tmp_path):
    assert os.path.exists(CLI), "run `make cli`"
    r = run()
    assert r.returncode == 1 and "usage" in r.stderr          # src/main.cpp:9-12
    r = run("--no-such-flag", "x.txt")
    assert r.returncode == 255 and "usage" in r.stderr        # `return -1`, src/main.cpp:23
    r = run(str(tmp_path / "missing.txt"))
    assert r.returncode == 1 and r.stderr.strip()             # unreadable scene, src/main.cpp:28-30
    r = run("-t")
    assert r.returncode == 1


@pytest.mark.gpu
def test_cli_renders_the_scene_file(scene_path, tmp_path):
    """`./raytracer -t4 file`: the reference's stdout lines, and a PNG whose pixels equal the library's own frame
    (seed 56, the default) after PNG::saveToFile's conversion."""
    src = scene_path("config1")
    r = run("-t4", "--stats", src, cwd=os.path.dirname(src))
    assert r.returncode == 0, r.stderr
    for needle in ("BVH created with", "Default render.", "Scene construction:", "BVH construction:", "Rendering:", "Mrays/s"):
        assert needle in r.stdout, r.stdout
    hs = capi.HostScene.load(src).build()
    out_png = os.path.join(os.path.dirname(src), hs.filename)
    assert os.path.exists(out_png)
    got = capi.png_read_rgba8(out_png)
    h, w = got.shape[:2]
    dev = capi.device_scene(hs.desc)
    frame, _ = dev.render(seed=56)
    want = capi.to_srgb8(frame).reshape(h, w, 4)
    assert np.array_equal(np.asarray(got).reshape(h, w, 4), want)
    # the device-side conversion (rt_resolve_rgba8) agrees with the host's
    assert np.array_equal(dev.resolve_rgba8().reshape(h, w, 4), want)
    dev.close()
    # --gpus 3 on this one GPU (--share-gpus): three tile partitions, device-to-device gather, the same PNG
    os.rename(out_png, out_png + ".one")
    r = run("--gpus", "3", "--share-gpus", "--tile", "16x8", src, cwd=os.path.dirname(src))
    assert r.returncode == 0, r.stderr
    assert np.array_equal(capi.png_read_rgba8(out_png), capi.png_read_rgba8(out_png + ".one"))
    r = run("--gpus", "64", src, cwd=os.path.dirname(src))
    assert r.returncode == 1 and "device" in r.stderr
    # --dump-hits: ids and t of the primary rays, same as rt_trace_primary
    dump = str(tmp_path / "hits.bin")
    r = run("--dump-hits", dump, src, cwd=os.path.dirname(src))
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(dump, dtype=np.uint32).reshape(-1, 2)
    dev = capi.device_scene(hs.desc)
    ids, t = dev.trace_primary()
    assert np.array_equal(raw[:, 0].view(np.int32), ids.reshape(-1)) and np.array_equal(raw[:, 1], t.reshape(-1).view(np.uint32))
    dev.close()
    # --device-bvh: no host tree, the GPU builds its own; same t for every primary ray
    dump2 = str(tmp_path / "hits_device_bvh.bin")
    r = run("--device-bvh", "--dump-hits", dump2, src, cwd=os.path.dirname(src))
    assert r.returncode == 0 and "BVH built on the GPU" in r.stdout, r.stdout + r.stderr
    raw2 = np.fromfile(dump2, dtype=np.uint32).reshape(-1, 2)
    assert np.array_equal(raw2[:, 1], raw[:, 1])
