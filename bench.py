#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and frame wall-time of the README-scene stand-in (1080x1080, 256 camera samples per
pixel, ~1.05 M triangles + 2 textured spheres + 3 textured planes, point + environment light, maxBounces 4).

    python bench.py [--gpus N] [--steps K] [--warmup W]      (N > 1: starts the N ranks itself, as a child process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step is one whole frame through the C ABI (rt_render): scene and BVH already resident in HBM, the frame left in a
device buffer. With N > 1 every rank renders its interleaved 32x32 tiles of the SAME frame (total work fixed: strong
scaling) and every rank's own pixels are gathered to rank 0 over RCCL/xGMI (one gather per frame, 1/N of the frame per rank).
"ray" = one BVH traversal query (Scene::findClosestObject / findAnyObject in the reference): camera, bounce and shadow
rays alike. Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=256, help="camera samples per pixel (numRays)")
    ap.add_argument("--size", type=int, default=1080)
    ap.add_argument("--grid", type=int, default=512, help="cells per side of the largest mesh (512 -> ~1.05 M triangles)")
    ap.add_argument("--spp-pass", type=int, default=0)
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample; 0 disables it")
    ap.add_argument("--scene-dir", default=None)
    return ap.parse_args()


def scene_files(args, rank, barrier):
    """The scene as FILES the reference's own parsers read (sdml + OBJ with vn + PNG textures), generated once."""
    from tools import gen_scenes
    d = args.scene_dir or f"/tmp/rt_bench_scene_{args.size}_{args.spp}_{args.grid}"
    marker = os.path.join(d, "readme.sdml.done")
    if rank == 0 and not os.path.exists(marker):
        os.makedirs(d, exist_ok=True)
        gen_scenes.scene_readme(d, "readme", size=args.size, spp=args.spp, grid=(args.grid, args.grid),
                                texsize=512 if args.grid >= 256 else 128)
        open(marker, "w").write("ok\n")
    barrier()
    return os.path.join(d, "readme.sdml")


def reference_cpu_run(scene_path, spp, threads):
    """Times the compiled reference itself (oracle/_ref/raytracer_ref, built from the reference's own sources by
    `make -C oracle ref` and shipped with the tree) on a copy of the scene file with `spp` samples per pixel.
    Returns the seconds of its "Rendering" phase (Scene::render: what `value` measures on the GPU), or None."""
    import re
    import subprocess
    exe = os.path.join(REPO, "oracle", "_ref", "raytracer_ref")
    if not os.path.exists(exe):
        return None
    text = open(scene_path).read()
    text, n = re.subn(r"numRays:\s*\d+", f"numRays: {spp}", text, count=1)
    if n != 1:
        return None
    copy = os.path.join(os.path.dirname(scene_path), f"cpu_baseline_{spp}spp.sdml")
    open(copy, "w").write(text)
    try:
        r = subprocess.run([exe, f"-t{threads}", copy], cwd=os.path.dirname(scene_path), capture_output=True, text=True, timeout=600)
    except (OSError, subprocess.TimeoutExpired):
        return None
    m = re.search(r"Rendering: (\d+) min ([\d.]+) sec", r.stdout)
    if r.returncode != 0 or not m:
        return None
    return int(m.group(1)) * 60 + float(m.group(2))


def bytes_per_ray(inner, prims, rays):
    """Algorithmic bytes of one traversal query: 32 B ray in + 64 B per inner-node expansion (two child boxes) +
    48 B per primitive test + 16 B hit record out."""
    if rays == 0:
        return 0.0
    return 32.0 + 64.0 * inner / rays + 48.0 * prims / rays + 16.0


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: this process (which has not touched the GPU and
    never will) starts the N ranks with torch.distributed.run as a CHILD process, relays what they print and exits with
    their code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            raise SystemExit(launch_ranks(args))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("RT_BENCH_DRY_RUN"):
        # launcher rehearsal for boxes without a GPU: rendezvous over gloo, nothing rendered, nothing measured
        if world > 1:
            dist.init_process_group("gloo")
            seen = torch.ones(1)
            dist.all_reduce(seen)
            assert int(seen[0]) == world
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup}), flush=True)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render loop has no CPU fallback")
    # RT_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: ranks share the GPUs there are
    # and the gather runs over gloo on host tensors. The measured configuration is the default: nccl (= RCCL), one GPU per rank.
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    gpu = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(gpu)
    tdev = torch.device("cuda", gpu) if backend == "nccl" else torch.device("cpu")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=tdev)
        else:
            dist.init_process_group(backend)

    def barrier():
        if world > 1:
            dist.barrier()

    from raytracer_amd import capi, desc_io, multi_gpu

    # The scene is parsed and its BVH built ONCE per host, by rank 0 with all host threads; the other ranks map the
    # finished arrays (what rt_scene_create reads) instead of repeating the work on the same cores.
    t_setup = time.time()
    path = scene_files(args, rank, barrier)
    shared = os.path.join(os.path.dirname(path), f"flattened_w{world}")
    t_build = 0.0
    if rank == 0:
        t = time.time()
        hs = capi.HostScene.load(path).build()
        t_build = time.time() - t
        desc = hs.desc
        if world > 1:
            desc_io.save_desc(desc, shared)
    barrier()
    if rank != 0:
        hs = desc_io.load_desc(shared)
        desc = hs.desc
    d = desc.contents
    dev = capi.device_scene(desc, device=gpu)
    W, H = d.options.width, d.options.height
    out = torch.zeros((H, W, 4), dtype=torch.float32, device=tdev)
    torch.cuda.synchronize()  # the library writes `out` from its own stream
    setup_s = time.time() - t_setup

    gatherer = multi_gpu.TileGather(W, H, (args.tile, args.tile), rank, world, out.device)

    gather_s = []  # per timed frame: this rank's time inside the gather (the render call before it has synchronised)

    def step():
        st = multi_gpu.render_partition(dev, out, rank, world, tile=(args.tile, args.tile), seed=56, samples_per_pass=args.spp_pass)
        if world > 1:
            t = time.perf_counter()
            gatherer.gather(out)
            gather_s.append(time.perf_counter() - t)
        return st

    for _ in range(args.warmup):
        step()
    gather_s.clear()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = [step() for _ in range(args.steps)]
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0

    # A ray = one visibility / closest-hit query answered by intersecting scene geometry (the reference's
    # Scene::findClosestObject / findAnyObject call): queries that traversed the BVH plus queries a plane of the scene
    # settled before the BVH was needed. Queries skipped because their light contributes exactly nothing are NOT counted.
    rays_local = sum(s.raysClosest + s.raysAny + s.raysPlaneAnswered for s in stats)
    traversed_local = sum(s.raysClosest + s.raysAny for s in stats)
    samples_local = sum(s.cameraSamples for s in stats)
    skipped_local = sum(s.raysSkipped for s in stats)
    plane_local = sum(s.raysPlaneAnswered for s in stats)
    agg = torch.tensor([float(elapsed), float(rays_local), float(samples_local), float(traversed_local), float(skipped_local),
                        float(plane_local)], dtype=torch.float64, device=tdev)
    per_rank = None  # N > 1: what every rank measured, so that a slow run can be read rank by rank
    if world > 1:
        tmax = agg[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        sums = agg[1:].clone()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        elapsed_local = elapsed
        elapsed, rays_total, samples_total, traversed_total = float(tmax[0]), float(sums[0]), float(sums[1]), float(sums[2])
        skipped_total, plane_total = float(sums[3]), float(sums[4])
        keys = ("msTotal", "msExtend", "msShade", "msShadow", "msDirect")
        mine = torch.tensor([elapsed_local / args.steps * 1e3, sum(gather_s) / max(1, len(gather_s)) * 1e3, max(gather_s, default=0.0) * 1e3]
                            + [sum(getattr(s, k) for s in stats) / args.steps for k in keys], dtype=torch.float64, device=tdev)
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        table = torch.stack(everyone).cpu().numpy()  # [rank, column]
        cols = ("wall_ms_per_step", "gather_ms", "gather_ms_max") + keys
        per_rank = {c: [round(float(v), 3) for v in table[:, i]] for i, c in enumerate(cols)}
    else:
        rays_total, samples_total, traversed_total = float(rays_local), float(samples_local), float(traversed_local)
        skipped_total, plane_total = float(skipped_local), float(plane_local)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    # `value`: BVH traversal queries per second, one per BVH::findClosestObject call of the reference that this library also
    # answers by walking the tree (src/acceleration/BVH.cpp:283). Queries a plane settles before the tree is needed and
    # queries whose answer cannot matter are reported beside it, not in it.
    mrays = traversed_total / elapsed / 1e6

    # ---- roofline of the dominant kernel: per-ray work counted by the kernels themselves on a few samples per pixel,
    # kernel time from the HIP events the library records around each launch of the timed steps.
    import ctypes as C
    o = d.options
    few = capi.RtOptions(o.width, o.height, o.bias, o.exposure, o.maxBounces, min(4, o.numRays), o.fisheye, o.focus, o.lens)
    full = capi.RtOptions(o.width, o.height, o.bias, o.exposure, o.maxBounces, o.numRays, o.fisheye, o.focus, o.lens)
    dev.set_view(options=few)
    _, cst = dev.render(seed=56, part_index=rank, part_count=world, tile=(args.tile, args.tile), collect_counters=True, download=False)
    dev.set_view(options=full)
    b_ext = bytes_per_ray(cst.innerExtend, cst.primExtend, cst.raysExtend)
    b_sh = bytes_per_ray(cst.innerShadow, cst.primShadow, cst.raysShadow)
    ms_ext = sum(s.msExtend for s in stats); ms_sh = sum(s.msShadow for s in stats)
    rays_ext = sum(s.raysExtend for s in stats); rays_sh = sum(s.raysShadow for s in stats)
    launches_ext = sum(s.launchesExtend for s in stats); launches_sh = sum(s.launchesShadow for s in stats)
    fused = sum(s.launchesFused for s in stats)
    if fused:
        # short passes (a part of the frame per GPU): a bounce's visibility queries and the next bounce's rays share a launch
        later = max(0, rays_ext - sum(s.cameraSamples for s in stats))
        kname, kms, krays, klaunch = "k_shadow_then_extend", ms_sh, rays_sh + later, launches_sh
        kb = (b_sh * rays_sh + b_ext * later) / max(1, krays)
    elif ms_ext >= ms_sh:
        kname, kb, kms, krays, klaunch = "k_extend", b_ext, ms_ext, rays_ext, launches_ext
    else:
        kname, kb, kms, krays, klaunch = "k_shadow", b_sh, ms_sh, rays_sh, launches_sh
    # Live in this process: kernel time (HIP events the library records on its own stream around every launch of the timed
    # steps) and the per-ray work (counted by the kernels on a few samples per pixel). Hardware counters cannot be read
    # from inside the process: they come from the rocprofv3 passes of this same command (tools/profile_on_box.sh ->
    # tools/pmc_traffic.py -> profiles/pmc_traffic.json) and are replayed only for the profiled workload and only while
    # the kernel sources still hash to what was profiled.
    from tools import pmc_traffic
    prof = None
    tpath = os.path.join(REPO, "profiles", "pmc_traffic.json")
    default_workload = (args.spp, args.size, args.grid, args.spp_pass, args.tile, world) == (256, 1080, 512, 0, 32, 1)
    if default_workload and os.path.exists(tpath):
        try:
            pj = json.load(open(tpath))
            if pj.get("source_sha1") == pmc_traffic.source_sha1():
                prof = pj["kernels"][kname]
        except (KeyError, ValueError):
            prof = None
    inner, prims = (cst.innerExtend, cst.primExtend) if kname == "k_extend" else (cst.innerShadow, cst.primShadow)
    lu_i, lu_p = ((cst.laneUseInnerExtend, cst.laneUsePrimExtend) if kname == "k_extend" else (cst.laneUseInnerShadow, cst.laneUsePrimShadow))
    lane_use = (inner + prims) / (inner / lu_i + prims / lu_p) if lu_i > 0 and lu_p > 0 else None
    avg_ms = kms / max(1, klaunch)
    rays_per_launch = krays / max(1, klaunch)
    alg_bytes = rays_per_launch * kb                      # algorithmic bytes of one launch
    alg_gbs = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    L1_PEAK = 1.46  # L1 accesses per clock per CU with every access a hit: tools/micro/l1_access_probe.hip, profiles/r02_micro_probes.txt
    l1 = prof.get("l1_access_busy") if prof else None
    hbm_gbs = prof["hbm_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 if prof and avg_ms > 0 else None
    roofline = {
        "kernel": kname,
        # Algorithmic bytes (32 B ray + 64 B per node visit + 48 B per object test + 16 B hit) per launch over the
        # launch's duration, against the HBM peak. The traversal kernels re-read the upper tree from the L1 / L2 caches, so
        # this exceeds the HBM peak: it is cache traffic. What actually leaves the L2 is `traffic` (hardware counters).
        "bound": "latency (dependent scattered record fetches, 7 waves/SIMD)",
        "achieved": round(alg_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg_gbs / HBM_PEAK_GBS, 4),
        "algorithmic_vs_hbm_peak": round(alg_gbs / HBM_PEAK_GBS, 4), "served_by": "L1/L2",
        "algorithmic_bytes_per_launch": int(alg_bytes), "bytes_per_ray": round(kb, 1), "rays_per_launch": round(rays_per_launch, 1),
        "avg_launch_ms": round(avg_ms, 4),
        "traffic": prof["hbm_bytes_per_launch"] if prof else None,
        "hbm_measured": ({"achieved": round(hbm_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_gbs / HBM_PEAK_GBS, 4),
                          "traffic_over_algorithmic": round(prof["hbm_bytes_per_launch"] / alg_bytes, 4) if alg_bytes else None,
                          "counters": "FETCH_SIZE (as counted: scattered 64-byte reads are counted exactly, profiles/r02_micro_probes.txt) + WRITE_SIZE"}
                         if prof else None),
        # The hardware rate the kernel comes nearest to. It does NOT bind: fetching records by quads (3.8 accesses per clock)
        # gains 0-9 % at equal occupancy, while time follows the number of resident waves and every extra dependent round
        # trip per node step (7 / 6 / 5 / 4 waves per SIMD: 50.6 / 57.9 / 65.9 / 79.3 ms; one more dependent read per step: + 17.5 %).
        "nearest_roof": {"name": "L1 scattered-access rate", "achieved": l1, "peak": L1_PEAK,
                         "unit": "L1 accesses per clock per CU (TCP_TOTAL_CACHE_ACCESSES_sum / 256 CUs / GPU cycles)",
                         "frac": round(l1 / L1_PEAK, 4) if l1 else None, "binds": False,
                         "peak_source": "this repository's micro-probe (tools/micro/l1_access_probe.hip), not a figure of the hardware guide",
                         "ta_busy": prof.get("ta_busy") if prof else None},
        "lane_use": round(lane_use, 4) if lane_use else None,
        "lane_use_by_loop": {"extend_inner": round(cst.laneUseInnerExtend, 3), "extend_prim": round(cst.laneUsePrimExtend, 3),
                             "shadow_inner": round(cst.laneUseInnerShadow, 3), "shadow_prim": round(cst.laneUsePrimShadow, 3)},
        "valu": ({"busy": prof.get("valu_busy"), "unit": "SQ_ACTIVE_INST_VALU * 4 / 1024 SIMDs / GPU cycles",
                  "instr_per_ray": round(prof["valu_instr_per_launch"] / rays_per_launch, 1) if "valu_instr_per_launch" in prof else None}
                 if prof else None),
        "l2": ({"read_requests_per_launch": prof.get("l2_read_requests_per_launch"), "misses_per_launch": prof.get("l2_misses_per_launch"),
                "hit": prof.get("l2_hit")} if prof else None),
        "live": ["avg_launch_ms", "rays_per_launch", "bytes_per_ray", "algorithmic_bytes_per_launch", "achieved", "frac", "lane_use",
                 "lane_use_by_loop", "mrays_per_s_in_kernel", "other_kernel"],
        "replayed_from": ({"file": "profiles/pmc_traffic.json", "fields": ["traffic", "hbm_measured", "nearest_roof.achieved", "nearest_roof.ta_busy", "valu", "l2"],
                           "profiled_avg_launch_ms": round(prof.get("trace_avg_us", 0) / 1e3, 4), "source_sha1_matches": True}
                          if prof else "no counters for this workload / these kernel sources: rerun tools/profile_on_box.sh"),
        "mrays_per_s_in_kernel": round(krays / (kms * 1e-3) / 1e6, 1) if kms > 0 else 0.0,
        "other_kernel": {"kernel": "k_shadow" if kname == "k_extend" else "k_extend",
                         "bytes_per_ray": round(b_sh if kname == "k_extend" else b_ext, 1),
                         "ms_total": round(ms_sh if kname == "k_extend" else ms_ext, 2),
                         "rays": int(rays_sh if kname == "k_extend" else (rays_ext - later if fused else rays_ext))},
        "fused_launches": int(fused),
    }

    # ---- CPU baseline on this box's host cores, bounded sample: the compiled reference when its binary is here, else
    # the CPU restatement of its loop.
    cpu = None
    if args.cpu_seconds > 0 and world == 1:
        from tests.oracle_binding import oracle_scene
        cores = min(16, len(os.sched_getaffinity(0)))
        orc = oracle_scene(desc, threads=cores)
        one = capi.RtOptions(o.width, o.height, o.bias, o.exposure, o.maxBounces, 1, o.fisheye, o.focus, o.lens)
        orc.set_view(options=one)
        t = time.perf_counter(); _, ost = orc.render(seed=56); t1 = time.perf_counter() - t
        spp_cpu = int(max(1, min(16, args.cpu_seconds / max(t1, 1e-3))))
        if spp_cpu > 1:
            orc.set_view(options=capi.RtOptions(o.width, o.height, o.bias, o.exposure, o.maxBounces, spp_cpu, o.fisheye, o.focus, o.lens))
            t = time.perf_counter(); _, ost = orc.render(seed=56); t1 = time.perf_counter() - t
        cpu_rays = ost.raysClosest + ost.raysAny
        port = {"value": round(cpu_rays / t1 / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
                "sample": f"same scene and resolution, {spp_cpu} of {o.numRays} samples per pixel, {cpu_rays} rays in {t1:.1f} s "
                          f"(CPU restatement of the reference loop, {cores} threads)"}
        cpu = port
        # The reference itself, when its binary (compiled from its own sources into oracle/_ref/) came along: same scene
        # files, same sample count. Its random streams differ from the restatement's, its number of traversal queries
        # per camera sample does not (same estimator), so the restatement's count stands in for the one it does not report.
        t_ref = reference_cpu_run(path, spp_cpu, cores)
        if t_ref and t_ref > 0.5:
            cpu = {"value": round(cpu_rays / t_ref / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "reference",
                   "sample": f"the compiled reference (./raytracer -t{cores}) on the same scene files at {spp_cpu} of {o.numRays} samples "
                             f"per pixel: Rendering phase {t_ref:.2f} s for {cpu_rays} traversal queries (count taken from the CPU "
                             f"restatement at the same sample count; the reference does not report one)",
                   "port": port}
        orc.close()

    tris = int(d.numPrims)
    line = {
        "metric": "Mrays/s", "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",  # strong: one fixed frame for every N
        "config": {"workload": f"README-scene stand-in: {W}x{H}, {o.numRays} spp, {tris} BVH objects + {d.numPlanes} planes, "
                               f"point + environment light, maxBounces {o.maxBounces}",
                   "width": W, "height": H, "spp": o.numRays, "objects": tris, "bvh_nodes": int(d.numNodes),
                   "parallelism": f"image tiles {args.tile}x{args.tile} interleaved over {world} GPU(s), "
                                  + ("RCCL gather to rank 0" if backend == "nccl" else f"REHEARSAL: {backend} gather, ranks share GPUs")},
        "wall_time_s": round(ms_per_step / 1e3, 3),
        "camera_msamples_per_s": round(samples_total / elapsed / 1e6, 2),
        "value_definition": "BVH traversal queries per second (camera, bounce and visibility rays that walked the tree: one "
                            "BVH::findClosestObject call of the reference each); queries a plane settles before the tree is needed and "
                            "zero-contribution queries are counted apart",
        "mrays_bvh_traversed_per_s": round(traversed_total / elapsed / 1e6, 2),
        # the same frame by two wider definitions of a ray
        "mrays_geometry_answered_per_s": round(rays_total / elapsed / 1e6, 2),   # + visibility queries one of the scene's planes settles
        "mrays_reference_equivalent_per_s": round((traversed_total + skipped_total) / elapsed / 1e6, 2),  # every query the reference traces for this frame
        "rays_bvh_traversed_per_frame": int(traversed_total / args.steps),
        "rays_plane_answered_per_frame": int(plane_total / args.steps),
        # visibility queries the reference traces and this library answers without a BVH traversal (zero contribution, or a plane alone decides)
        "rays_not_traversed_per_frame": int(skipped_total / args.steps),
        "reference_queries_per_frame": int((traversed_total + skipped_total) / args.steps),
        "rays_per_camera_sample": round(traversed_total / max(1.0, samples_total), 3),
        "kernel_ms_per_frame": {k: round(sum(getattr(s, k) for s in stats) / args.steps, 2)
                                for k in ("msTotal", "msRaygen", "msExtend", "msShade", "msShadow", "msDirect", "msAccumulate", "msResolve")},
        "setup_s": round(setup_s, 1), "scene_build_s": round(t_build, 2),
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    if world > 1:
        # one line per rank would be easier to read and harder to parse: lists are indexed by rank
        line["multi_gpu"] = {"backend": backend, "world_size": dist.get_world_size(),
                             "gather_ms": round(sum(gather_s) / max(1, len(gather_s)) * 1e3, 3),   # rank 0, per frame
                             "gather_ms_max_over_ranks": max(per_rank["gather_ms_max"]),
                             "gather_bytes_per_rank": int(W * H * 16 / world),
                             "per_rank": per_rank,
                             "scene": "parsed and BVH built once, by rank 0; other ranks map the flattened arrays"}
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
