#!/usr/bin/env python3
"""GPU box: per scene, how far the HIP frame is from the CPU restatement's at the same counter-RNG streams."""
import os
import sys
import tempfile

import hashlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi  # noqa: E402
from tests.oracle_binding import oracle_scene  # noqa: E402
from tools import gen_scenes  # noqa: E402

names = sys.argv[1:] or gen_scenes.DEFAULT + ["readme_mid"]
with tempfile.TemporaryDirectory() as tmp:
    for name in names:
        gen_scenes.SCENES[name](tmp)
        ext = "txt" if name.startswith("spheres_txt") else "sdml"
        hs = capi.HostScene.load(os.path.join(tmp, f"{name}.{ext}")).build()
        dev, orc = capi.device_scene(hs.desc), oracle_scene(hs.desc)
        a, st = dev.render(seed=3)
        b, ost = orc.render(seed=3)
        diff = np.abs(a - b)
        mse = float(((a[..., :3] - b[..., :3]) ** 2).mean())
        psnr = float("inf") if mse == 0 else 10 * np.log10(1.0 / mse)
        print(f"{name:22s} pixels differing {int((diff.max(-1) > 0).sum()):7d} of {a.shape[0] * a.shape[1] if a.ndim == 3 else a.size // 4:8d}  max|d| {diff.max():.3e}  PSNR {psnr:7.1f} dB  "
              f"gpu frame sha1 {hashlib.sha1(a.tobytes()).hexdigest()[:12]}  rays gpu {st.raysClosest + st.raysAny + st.raysSkipped} cpu {ost.raysClosest + ost.raysAny}", flush=True)
        dev.close(); orc.close()
