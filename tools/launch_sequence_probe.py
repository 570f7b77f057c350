#!/usr/bin/env python3
"""GPU box, under `rocprofv3 --kernel-trace`: one frame of the whole bench scene, then one frame of one rank's share of it
(1/N of the tiles), launch by launch. With `--summarise <dir>` (no GPU needed) it reads the trace back and sets the two
frames side by side: for every launch of the frame its duration on the whole frame, on the share, and the share's
duration times N — where a rank of N loses time against an N-th of the whole frame's.

    rocprofv3 --kernel-trace -d gpurun_out/seq -o seq -- python3 tools/launch_sequence_probe.py
    python3 tools/launch_sequence_probe.py --summarise gpurun_out/seq
"""
import glob
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHARE = int(os.environ.get("PROBE_SHARE", "8"))


def run():
    from raytracer_amd import capi
    from tools import gen_scenes
    d = "/tmp/rt_probe_scene256"
    os.makedirs(d, exist_ok=True)
    if not os.path.exists(d + "/readme.sdml"):
        gen_scenes.scene_readme(d, "readme", size=1080, spp=256, grid=(512, 512), texsize=512)
    scene = d + "/readme.sdml"
    if os.environ.get("PROBE_LIGHT"):  # keep one of the two lights: whose visibility queries make a launch's tail?
        keep = os.environ["PROBE_LIGHT"]
        text = "".join(ln for ln in open(scene) if "<Light" not in ln or f'type="{keep}"' in ln)
        scene = d + f"/readme_{keep}.sdml"
        open(scene, "w").write(text)
    hs = capi.HostScene.load(scene).build()
    dev = capi.device_scene(hs.desc)
    if os.environ.get("PROBE_BOUNCES"):  # does a long last launch belong to the last bounce, or to the fourth?
        import ctypes
        opt = capi.RtOptions()
        ctypes.memmove(ctypes.byref(opt), ctypes.byref(hs.desc.contents.options), ctypes.sizeof(opt))
        opt.maxBounces = int(os.environ["PROBE_BOUNCES"])
        dev.set_view(options=opt)
    # each frame twice: the first of a size allocates; the summary takes the last frame of each size
    for n in (1, 1, SHARE, SHARE):
        _, st = dev.render(part_index=int(os.environ.get("PROBE_PART", "0")) % n, part_count=n, seed=int(os.environ.get("PROBE_SEED", "56")), download=False)
        print(f"1/{n}: device {st.msTotal:.2f} ms", flush=True)
    dev.close()


def summarise(root):
    db = sorted(glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True))[0]
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = con.execute(f"select {name}, start, end from kernels order by start").fetchall()
    frames, cur = [], None
    for n, s, e in rows:
        m = re.search(r"(k_\w+)", n)
        if not m:
            continue
        k = m.group(1)
        if k == "k_raygen":
            cur = []
            frames.append(cur)
        if cur is not None:
            cur.append((k, s, e))
    # frames: whole, whole, share, share (a frame may take several passes, each starting with k_raygen: group by size instead)
    def passes_of(frames_):
        return [(k, (e - s) / 1e6) for f in frames_ for k, s, e in f]
    half = len(frames) // 2
    whole, share = frames[half // 2:half], frames[half + (len(frames) - half) // 2:]
    a, b = passes_of(whole), passes_of(share)
    span = lambda fr: (fr[-1][-1][2] - fr[0][0][1]) / 1e6
    print(f"whole frame: {len(a)} launches, {span(whole):.2f} ms from first start to last end, {sum(t for _, t in a):.2f} ms inside kernels")
    print(f"1/{SHARE} of it:  {len(b)} launches, {span(share):.2f} ms from first start to last end, {sum(t for _, t in b):.2f} ms inside kernels")
    if [k for k, _ in a] != [k for k, _ in b]:
        print("(the two frames launch different kernel sequences: listed separately)")
        for tag, seq in (("whole", a), (f"1/{SHARE}", b)):
            print(tag)
            for i, (k, t) in enumerate(seq):
                print(f"  {i:3d} {k:<24} {t:9.3f} ms")
        by = lambda seq: {k: sum(t for kk, t in seq if kk == k) for k in dict(seq)}
        A, B = by(a), by(b)
        print(f"{'kernel':<24} {'whole ms':>10} {'share ms':>10} {'share x N':>10}")
        for k in sorted(set(A) | set(B)):
            print(f"{k:<24} {A.get(k, 0):>10.3f} {B.get(k, 0):>10.3f} {SHARE * B.get(k, 0):>10.3f}")
        return
    print(f"{'#':>3} {'kernel':<24} {'whole ms':>10} {'share ms':>10} {'share x N':>10} {'lost ms':>8}")
    lost = 0.0
    for i, ((k, ta), (_, tb)) in enumerate(zip(a, b)):
        lost += tb - ta / SHARE
        print(f"{i:>3} {k:<24} {ta:>10.3f} {tb:>10.3f} {SHARE * tb:>10.3f} {tb - ta / SHARE:>8.3f}")
    print(f"lost against an N-th of the whole frame's launches: {lost:.2f} ms")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2])
    else:
        run()
