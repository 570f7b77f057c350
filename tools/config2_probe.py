#!/usr/bin/env python3
"""GPU box: the primary-rays-only configuration (1080x1080 primary rays on the ~70 k-triangle mesh): rt_trace_primary and the 1-spp render."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi  # noqa: E402
from tools import gen_scenes  # noqa: E402

with tempfile.TemporaryDirectory() as tmp:
    gen_scenes.SCENES["mesh70k"](tmp)
    hs = capi.HostScene.load(os.path.join(tmp, "mesh70k.sdml")).build()
dev = capi.device_scene(hs.desc)
dev.trace_primary(); dev.render(download=False)
t = time.perf_counter(); ids, tt = dev.trace_primary(); t1 = (time.perf_counter() - t) * 1e3
t = time.perf_counter(); _, st = dev.render(download=False); t2 = (time.perf_counter() - t) * 1e3
n = ids.size
print(f"rt_trace_primary: {t1:.2f} ms wall for {n} rays incl. the copy of ids and t to the host ({n / t1 / 1e3:.0f} Mrays/s)")
print(f"rt_render 1 spp, maxBounces 1: wall {t2:.2f} ms, device {st.msTotal:.2f} ms (extend {st.msExtend:.2f}, shade {st.msShade:.2f}, "
      f"shadow {st.msShadow:.2f}), {(st.raysClosest + st.raysAny) / st.msTotal / 1e3:.0f} Mrays/s")
