#!/usr/bin/env python3
"""Where /root/reference exists (needs oracle/_ref/ref_harness): random .txt scene files through this repository's host
parser and through the compiled reference's, compared record for record and bit for bit by tools/check_host_vs_ref.py
(options, camera, objects incl. the triangle constants, materials, planes, lights, textures, BVH nodes and order).

    python tools/fuzz_parsers_vs_ref.py [--count 200] [--seed 1]"""
import argparse
import io
import os
import sys
import tempfile
from contextlib import redirect_stdout

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import check_host_vs_ref, check_oracle_vs_ref, gen_scenes  # noqa: E402


def num(rng, lo=-4.0, hi=4.0):
    v = rng.uniform(lo, hi)
    style = rng.integers(0, 6)
    if style == 0:
        return "%.3f" % v
    if style == 1:
        return "%.7e" % v
    if style == 2:
        return "%+.5f" % v
    if style == 3:
        return str(int(round(v)))
    if style == 4:
        return ("%.4f" % v).replace("0.", ".", 1) if 0 < abs(v) < 1 and v > 0 else "%.4f" % v
    return repr(float(np.float32(v)))


WEIGHTS = np.array([14, 3, 10, 4, 4, 3, 16, 12, 1, 2, 2, 1, 1, 3, 3, 3, .5, 1, 10, 3, 3, 2, 1.5])


def random_txt(rng, tmp, k):
    w, h = int(rng.integers(4, 40)), int(rng.integers(4, 40))
    lines = [f"png {w} {h} out_{k}.png"]
    npoints = 0
    materials = ["glass", "plastic", "none", "copper", "gold", "mirror", "diffuse"]
    sep = lambda: " " * int(rng.choice([1, 1, 1, 2, 3]))  # runs of spaces are one separator
    for _ in range(int(rng.integers(5, 80))):
        kind = rng.choice(["sphere", "sun", "color", "plane", "bulb", "environment", "xyz", "trif", "expose", "bounces", "aa",
                           "roughness", "ior", "eye", "forward", "up", "fisheye", "dof", "material", "texture", "unknown", "blank",
                           "obj"],
                          p=WEIGHTS / WEIGHTS.sum())
        if kind == "sphere":
            lines.append(sep().join(["sphere", num(rng), num(rng), num(rng, -6, -1), num(rng, 0.05, 1.5)]))
        elif kind in ("sun", "bulb", "eye", "up"):
            lines.append(sep().join([kind, num(rng), num(rng), num(rng)]))
        elif kind == "forward":
            lines.append(sep().join(["forward", num(rng, -0.5, 0.5), num(rng, -0.5, 0.5), num(rng, -2, -0.5)]))
        elif kind == "color":
            lines.append(sep().join(["color", num(rng, 0, 1.5), num(rng, 0, 1.5), num(rng, 0, 1.5)]))
        elif kind == "plane":
            lines.append(sep().join(["plane", num(rng, -1, 1), num(rng, 0.2, 1), num(rng, -1, 1), num(rng, 0.5, 5)]))
        elif kind == "environment":
            if rng.random() < 0.5:
                lines.append(sep().join(["environment", num(rng, 5, 20)]))
            else:
                lines.append(sep().join(["environment", num(rng, 5, 20), num(rng, 0.5, 2), os.path.join(tmp, "sky.png")]))
        elif kind == "xyz":
            lines.append(sep().join(["xyz", num(rng), num(rng), num(rng, -6, -1)]))
            npoints += 1
        elif kind == "trif" and npoints >= 3:
            idx = [int(rng.integers(1, npoints + 1)) if rng.random() < 0.7 else -int(rng.integers(1, npoints + 1)) for _ in range(3)]
            lines.append(sep().join(["trif"] + [str(i) for i in idx]))
        elif kind == "expose":
            lines.append(sep().join(["expose", num(rng, 0.5, 4)]))
        elif kind == "bounces":
            lines.append(f"bounces {int(rng.integers(1, 7))}")
        elif kind == "aa":
            lines.append(f"aa {int(rng.integers(1, 5))}")
        elif kind in ("roughness", "ior"):
            # In the reference these overwrite a field of a material whose lobes are already built: no effect on any
            # pixel, but the dumped field would differ from the host's (which keeps the constructor's value, the one
            # the lobes came from). Emitted as a keyword the parsers ignore instead.
            lines.append(sep().join(["shininess", num(rng, 0, 2)]))
        elif kind == "fisheye":
            lines.append("fisheye")
        elif kind == "dof":
            lines.append(sep().join(["dof", num(rng, 1, 5), num(rng, 0.01, 0.5)]))
        elif kind == "material":
            lines.append(str(rng.choice(materials)))
        elif kind == "texture":
            lines.append("texture " + os.path.join(tmp, str(rng.choice(["marble.png", "wood.png"]))))
        elif kind == "unknown":
            lines.append(sep().join([str(rng.choice(["gi", "shininess", "transparency", "refractive", "#comment"])), num(rng)]))
        elif kind == "blank":
            lines.append("")
        elif kind == "obj":
            lines.append(sep().join(["obj", num(rng, -1, 1), num(rng, -1, 1), num(rng, -5, -2), num(rng, 0.5, 3),
                                     os.path.join(tmp, str(rng.choice(["blob.obj", "quirks.obj"])))]))
    path = os.path.join(tmp, f"fuzz_{k}.txt")
    open(path, "w").write("\n".join(lines) + "\n")
    return path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--count", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--render", action="store_true",
                    help="also compare the CPU restatement with the reference: primary hits and the -t1 frame, bit for bit "
                         "(tools/check_oracle_vs_ref.py)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    bad = 0
    tmp = tempfile.mkdtemp(prefix="rt_parser_fuzz_")  # kept: a failing file names its assets by path
    if True:
        gen_scenes.texture_marble(os.path.join(tmp, "marble.png"), 32)
        gen_scenes.texture_wood(os.path.join(tmp, "wood.png"), 16)
        gen_scenes.texture_sky(os.path.join(tmp, "sky.png"), 32, 16)
        gen_scenes.write_obj(os.path.join(tmp, "blob.obj"), *gen_scenes.blob_mesh(9, 7))
        gen_scenes.scene_obj_quirks(tmp, "quirks")
        for k in range(args.count):
            path = random_txt(rng, tmp, k)
            buf = io.StringIO()
            try:
                with redirect_stdout(buf):
                    ok = check_host_vs_ref.check(path)
                    if ok and args.render:
                        ok = check_oracle_vs_ref.check(path)
            except Exception as e:  # the reference aborting on a file is a generator problem, not a parser difference
                ok = False
                buf.write(f"exception: {e}")
            if not ok:
                bad += 1
                print(buf.getvalue().strip()[:600])
                print("   file:", path)
    print(f"{args.count} random .txt scenes, {bad} differ from the compiled reference")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
