#!/usr/bin/env python3
"""Bit-compare the CPU restatement (oracle/librt_oracle.so) with the compiled reference (oracle/_ref/ref_harness):
primary hits (ids + t bits) and the -t1 float framebuffer in ref-mt19937 replay mode.
    python tools/check_oracle_vs_ref.py scene [...]"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi  # noqa: E402
from tests import refdump  # noqa: E402
from tests.oracle_binding import oracle_scene  # noqa: E402


def check(path, render=True):
    tmp = tempfile.mkdtemp()
    hs = capi.HostScene.load(path).build()
    osc = oracle_scene(hs.desc)
    msgs = []
    refdump.run_harness("hits", path, tmp + "/hits.bin")
    rh = refdump.read_hits(tmp + "/hits.bin")
    ids, t = osc.trace_primary()
    id_bad = int((ids != rh["orig"]).sum()); t_bad = int((t.view(np.uint32) != rh["tbits"]).sum())
    msgs.append(f"hits: id mismatches {id_bad}, t-bit mismatches {t_bad} of {ids.size} ({int((ids >= 0).sum())} hit)")
    ok = id_bad == 0 and t_bad == 0
    if render:
        refdump.run_harness("render", path, tmp + "/render.bin", 1, cwd=tmp)
        rimg, r8 = refdump.read_render(tmp + "/render.bin")
        img, st = osc.render(rng_mode=capi.RT_RNG_REF_MT19937)
        same = np.array_equal(img.view(np.uint32), rimg.view(np.uint32))
        nbad = int((img.view(np.uint32) != rimg.view(np.uint32)).any(-1).sum())
        o8 = osc.resolve_rgba8()
        msgs.append(f"render -t1: {'bit-identical' if same else f'{nbad} pixels differ, max abs diff {np.nanmax(np.abs(img - rimg)):.3g}'};"
                    f" rgba8 mismatches {int((o8 != r8).sum())}; rays={st.raysClosest + st.raysAny}")
        ok = ok and same and np.array_equal(o8, r8)
    print(("OK   " if ok else "FAIL ") + os.path.basename(path), *msgs, sep="  ")
    return ok


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    ok = all([check(p, render="--no-render" not in sys.argv) for p in args])
    sys.exit(0 if ok else 1)
