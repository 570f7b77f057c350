"""A fixed slice of the randomised parity campaign (tools/fuzz_parity.py), inside the suite: random scenes with fixed
seeds, each checked against the CPU restatement through the three query kinds (closest, any, primary: bit-exact ids,
bits of t, points, normals, visibility answers) and one tiny frame (alpha exact, at most 1 % of the pixels off by more
than 1e-3: a single path may branch the other way on a last-bit difference of the device's sinf / cosf). One scene in
five comes from the family that once got through every hand-written test: meshes on a coarse grid with flat,
axis-aligned boxes around point lights closer than one unit (visibility rays with |d| < 1 that run almost parallel to the
boxes). Another one in five puts a cluster of objects and a bright bulb in front of a plane up to 2e8 units away: the
visibility queries from there are the ones the library splits over idle lanes at the end of a launch. The second test pins the min/max form of the slab test against the compare-and-select form of
src/acceleration/BVH.cpp:390-409 on the same rays."""
import os
import time

import numpy as np
import pytest

from raytracer_amd import capi
from tests.oracle_binding import oracle_scene
from tools import fuzz_parity

pytestmark = pytest.mark.gpu

SLICES = ((101, 700), (39, 700), (7, 600))  # (seed of the scene stream, scenes)
RAYS_PER_SCENE = 6000
WALL_BUDGET_S = 90.0  # a guard, not the plan: the slice takes about 25 s on the GPU box


@pytest.mark.parametrize("seed,count", SLICES)
def test_random_scenes_match_the_oracle(seed, count):
    rng = np.random.default_rng(seed)
    t0 = time.time()
    bad, diverged, rays, flat_scenes, far_scenes = [], 0, 0, 0, 0
    for k in range(count):
        hs, rays_of, family = fuzz_parity.scene_of_the_stream(rng, k)
        o, d = rays_of(rng, RAYS_PER_SCENE)
        dev, orc = capi.device_scene(hs.desc), oracle_scene(hs.desc, threads=8)
        try:
            res = fuzz_parity.check_scene(hs, dev, orc, o, d, k)
        finally:
            dev.close(); orc.close()
        flat_scenes += int(family == "flat")
        far_scenes += int(family == "far")
        rays += res["rays"]
        diverged += int(res["ok"] and not res["same_counts"])
        if not res["ok"]:
            bad.append(f"seed {seed} scene {k}: {res['what']}")
        assert time.time() - t0 < WALL_BUDGET_S, f"slice overran its budget after {k + 1} scenes"
    print(f"seed {seed}: {count} scenes ({flat_scenes} of the flat-box family, {far_scenes} of the far-plane family), {rays / 1e6:.1f} M rays, {diverged} frames with a diverged "
          f"path, {time.time() - t0:.1f} s")
    assert not bad, "\n".join(bad[:10])
    assert flat_scenes >= count // 5 - 1 and far_scenes >= count // 5 - 1
    # single diverged paths are expected in a few percent of the frames, not in most
    assert diverged <= count // 8


def _scene_for_slab_test(rng, kind):
    if kind == "flat":
        return fuzz_parity.flat_box_scene(rng, 4), fuzz_parity.near_parallel_rays(rng, 150_000)
    hs = fuzz_parity.random_scene(rng, 3 if kind == "soup_grid" else 1)
    return hs, fuzz_parity.random_rays(rng, 150_000)


@pytest.mark.parametrize("kind", ["flat", "soup", "soup_grid", "readme_small", "mesh70k"])
def test_minmax_slab_test_equals_compare_and_select_form(kind, scene_path):
    """The traversal kernels test boxes with v_min_f32 / v_max_f32 when a ray's reciprocal direction and origin and every
    box are finite (intersectAABBFinite / slabFinite) and with the reference's compare-and-select chain otherwise
    (src/acceleration/BVH.cpp:390-409: std::min / std::max, whose NaN behaviour depends on argument order). The same
    rays through a scene created with RT_EXACT_SLAB_TEST set (compare-and-select for every ray) and through a scene
    created without it must give the same bits: ids, t, points, normals, visibility answers, primary hits and a frame."""
    rng = np.random.default_rng(17)
    if kind in ("readme_small", "mesh70k"):
        hs = capi.HostScene.load(scene_path(kind)).build()
        o, d = fuzz_parity.random_rays(rng, 300_000)
        o2, d2 = fuzz_parity.near_parallel_rays(rng, 100_000)
        o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
    else:
        for _ in range(40):  # a scene of that family with a tree worth the name
            hs, (o, d) = _scene_for_slab_test(rng, kind)
            if hs.desc.contents.numNodes >= 64:
                break
        assert hs.desc.contents.numNodes >= 64
    fast = capi.device_scene(hs.desc)
    assert "RT_EXACT_SLAB_TEST" not in os.environ
    os.environ["RT_EXACT_SLAB_TEST"] = "1"
    try:
        exact = capi.device_scene(hs.desc)
    finally:
        del os.environ["RT_EXACT_SLAB_TEST"]
    try:
        for a, b in zip(fast.trace_closest(o, d), exact.trace_closest(o, d)):
            assert fuzz_parity.same_bits(a, b)
        assert np.array_equal(fast.trace_any(o, d), exact.trace_any(o, d))
        for a, b in zip(fast.trace_primary(), exact.trace_primary()):
            assert fuzz_parity.same_bits(a, b)
        img_f, st_f = fast.render(seed=5, collect_counters=True)
        img_e, st_e = exact.render(seed=5, collect_counters=True)
        assert fuzz_parity.same_bits(img_f, img_e)
        # the same boxes entered and the same objects tested, not only the same answers
        assert (st_f.innerExpansions, st_f.primTests) == (st_e.innerExpansions, st_e.primTests)
        assert st_f.innerExpansions > 0
    finally:
        fast.close(); exact.close()
