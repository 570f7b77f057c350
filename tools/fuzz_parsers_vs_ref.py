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


def vec(rng, lo=-4.0, hi=4.0, zlo=None, zhi=None):
    z = num(rng, lo, hi) if zlo is None else num(rng, zlo, zhi)
    sp = lambda: " " * int(rng.integers(0, 3))
    return f"({sp()}{num(rng, lo, hi)},{sp()}{num(rng, lo, hi)},{sp()}{z}{sp()})"


def random_sdml(rng, tmp, k):
    """A random .sdml scene: every tag and option the loader knows, in random order (lights anywhere: they are read
    last), named and ad-hoc materials, textures, OBJ includes, optional options left out at random."""
    w, h = int(rng.integers(4, 40)), int(rng.integers(4, 40))
    opts = [f"width: {w}", f"height: {h}", f"filename: out_{k}.png"]
    if rng.random() < 0.7: opts.append(f"numRays: {int(rng.integers(1, 5))}")
    if rng.random() < 0.7: opts.append(f"maxBounces: {int(rng.integers(1, 7))}")
    if rng.random() < 0.4: opts.append(f"bias: {num(rng, 1e-5, 1e-3)}")
    if rng.random() < 0.3: opts.append(f"exposure: {num(rng, 0.5, 4)}")
    mode = rng.integers(0, 4)
    if mode == 1: opts.append("fisheye: 1")
    if mode == 2: opts += [f"focus: {num(rng, 1, 5)}", f"lens: {int(rng.integers(0, 3))}"]
    rng.shuffle(opts)
    body = []
    if rng.random() < 0.8:
        cam = []
        if rng.random() < 0.7: cam.append("eye: " + vec(rng, -1, 1))
        if rng.random() < 0.7: cam.append("forward: " + vec(rng, -0.4, 0.4, -2, -0.5))
        if rng.random() < 0.5: cam.append("up: " + vec(rng, -1, 1))
        body.append("  <Camera options={%s}/>" % "; ".join(cam))
    names = ["glass", "plastic", "mirror", "copper penny", "metallic gold", "something else"]

    def inner():
        out = []
        r = rng.random()
        if r < 0.45:
            out.append(f'    <Material name="{rng.choice(names)}"/>')
        elif r < 0.65:
            t = str(rng.choice(["glass", "mirror", "plastic", "metal", "dialectric"]))
            fields = [f"type: {t}"]
            for key, lo, hi in (("Kd", 0, 1), ("Ks", 0, 1), ("eta", 1.1, 2.5), ("Kr", 0.1, 1), ("Kt", 0.1, 1), ("Ka", 0.5, 4), ("roughness", 0, 0.5)):
                if t != "dialectric" or rng.random() < 0.5:
                    fields.append(f"{key}: {num(rng, lo, hi)}")
            if t != "dialectric":
                out.append("    <Material options={%s}/>" % "; ".join(fields))
        if rng.random() < 0.3:
            out.append(f'    <Texture path="{os.path.join(tmp, str(rng.choice(["marble.png", "wood.png"])))}"/>')
        return out

    for _ in range(int(rng.integers(1, 25))):
        kind = rng.choice(["sphere", "triangle", "plane", "point", "distant", "environment", "wavefront"], p=[.3, .25, .1, .1, .08, .07, .1])
        col = ("; color: " + vec(rng, 0, 1.2)) if rng.random() < 0.6 else ""
        if kind == "sphere":
            o = [f"radius: {num(rng, 0.05, 1.2)}", "center: " + vec(rng, -3, 3, -6, -1)]
            rng.shuffle(o)
            body += ['  <Shape type="sphere" options={%s%s}>' % ("; ".join(o), col)] + inner() + ["  </Shape>"]
        elif kind == "triangle":
            o = [f"p{i}: " + vec(rng, -3, 3, -6, -1) for i in (1, 2, 3)]
            if rng.random() < 0.3: o += [f"t{i}: " + vec(rng, 0, 15) for i in (1, 2, 3)]
            body += ['  <Shape type="triangle" options={%s%s}>' % ("; ".join(o), col)] + inner() + ["  </Shape>"]
        elif kind == "plane":
            o = ["normal: " + vec(rng, -1, 1), f"D: {num(rng, 0.5, 5)}"]
            if rng.random() < 0.5: o.append(f"texture-zoom: {num(rng, 1, 50)}")
            if rng.random() < 0.4: o.append("texture-shift: " + vec(rng, -5, 5))
            if rng.random() < 0.3: o.append("top-left: " + vec(rng, -2, 2))
            body += ['  <Shape type="plane" options={%s%s}>' % ("; ".join(o), col)] + inner() + ["  </Shape>"]
        elif kind == "point":
            body.append('  <Light type="point" options={center: %s%s}/>' % (vec(rng, -3, 3), col))
        elif kind == "distant":
            body.append('  <Light type="distant" options={direction: %s%s}/>' % (vec(rng, -1, 1), col))
        elif kind == "environment":
            if rng.random() < 0.5:
                body.append('  <Light type="environment" options={radius: %s%s}/>' % (num(rng, 5, 20), col))
            else:
                body.append('  <Light type="environment" path="%s" options={radius: %s; scale: %s}/>' % (os.path.join(tmp, "sky.png"), num(rng, 5, 20), num(rng, 0.5, 2)))
        else:
            o = ["center: " + vec(rng, -1, 1, -5, -2), f"scale: {num(rng, 0.5, 3)}"]
            body += ['  <Wavefront path="%s" options={%s%s}>' % (os.path.join(tmp, str(rng.choice(["blob.obj", "quirks.obj"]))), "; ".join(o), col)]
            body += [l for l in inner() if "Texture" not in l] + ["  </Wavefront>"]
    path = os.path.join(tmp, f"fuzz_{k}.sdml")
    open(path, "w").write("<Scene options={%s}>\n" % "; ".join(opts) + "\n".join(body) + "\n</Scene>\n")
    return path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--format", choices=["txt", "sdml"], default="txt")
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
            path = random_txt(rng, tmp, k) if args.format == 'txt' else random_sdml(rng, tmp, k)
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
    print(f"{args.count} random .{args.format} scenes, {bad} differ from the compiled reference")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
