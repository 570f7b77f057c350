"""rt_trace_primary_static: the entry point that allocates nothing (its queue ticket and the device-side staging of
ids and t live in a static device array of the library). Checked through the C ABI against rt_trace_primary, bit for
bit, on two calls in a row, the second smaller than the first (the array keeps the first call's contents)."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import capi

pytestmark = pytest.mark.gpu


def _same(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


def test_static_entry_matches_allocating_entry_twice_second_smaller(scene_path):
    hs = capi.HostScene.load(scene_path("mesh70k")).build()
    dev = capi.device_scene(hs.desc)
    try:
        L = capi.load_device_library()
        assert L.rt_trace_primary_static_max_pixels() >= 1080 * 1080
        want_full = dev.trace_primary()
        got_full = dev.trace_primary_static()
        assert int((got_full[0] >= 0).sum()) > 100_000
        assert _same(got_full, want_full), "first call (1080x1080) differs from rt_trace_primary"
        # second call, smaller and offset: stale ids / t of the first call must not leak into the result
        rect = dict(x0=411, y0=300, w=157, h=93)
        want = dev.trace_primary(**rect)
        got = dev.trace_primary_static(**rect)
        assert _same(got, want), "second, smaller call differs from rt_trace_primary"
        assert _same(got, (want_full[0][300:393, 411:568], want_full[1][300:393, 411:568]))
        # a third call of the full size again
        assert _same(dev.trace_primary_static(), want_full)
        # sizes beyond the static buffer are refused with a message, not truncated
        big = capi.RtOptions(4096, 2048, 1e-4, -1.0, 1, 1, 0, -1.0, 0.0)
        dev.set_view(options=big)
        with pytest.raises(RuntimeError, match="exceed the static scratch"):
            dev.trace_primary_static()
        ids, t = dev.trace_primary_static(w=2048, h=2048)  # exactly the maximum
        wi, wt = dev.trace_primary(w=2048, h=2048)
        assert _same((ids, t), (wi, wt))
    finally:
        dev.close()


def _device_free_bytes():
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_static_entry_allocates_no_device_memory_on_a_fresh_scene(scene_path):
    """The static entry is the FIRST thing called on a fresh scene, before any rt_trace_primary has given the scene its
    own staging: if it fell back to the scene's grow-on-demand buffers, the GPU's free memory would drop. It must not move,
    on the first call or on a second, smaller one. (A first scene warms the runtime: loading the code object, which is what
    places the static array, and the runtime's own copy staging are not this entry point's doing.)"""
    warm = capi.device_scene(capi.HostScene.load(scene_path("mesh1k")).build().desc)
    warm.trace_primary_static()
    hs = capi.HostScene.load(scene_path("mesh70k")).build()
    dev = capi.device_scene(hs.desc)
    try:
        before = _device_free_bytes()
        first = dev.trace_primary_static()
        after_first = _device_free_bytes()
        rect = dict(x0=100, y0=200, w=301, h=77)
        second = dev.trace_primary_static(**rect)
        after_second = _device_free_bytes()
        assert before == after_first == after_second, (before, after_first, after_second)
        # only now the allocating entry point, as the checker
        want = dev.trace_primary()
        assert after_second - _device_free_bytes() >= 2 * 1080 * 1080 * 4, "rt_trace_primary is expected to allocate its staging"
        assert _same(first, want)
        assert _same(second, (want[0][200:277, 100:401], want[1][200:277, 100:401]))
    finally:
        dev.close(); warm.close()


def test_static_entry_on_a_second_scene_of_the_same_device(scene_path):
    """Every scene on a GPU shares that GPU's array; sequential calls from different scenes must not disturb each other."""
    a = capi.device_scene(capi.HostScene.load(scene_path("mesh1k")).build().desc)
    b = capi.device_scene(capi.HostScene.load(scene_path("spheres_txt")).build().desc)
    try:
        ra, rb = a.trace_primary(), b.trace_primary()
        assert _same(a.trace_primary_static(), ra)
        assert _same(b.trace_primary_static(), rb)
        assert _same(a.trace_primary_static(), ra)
    finally:
        a.close(); b.close()
