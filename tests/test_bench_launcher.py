"""bench.py --gpus N: without a launcher around it the script starts its N ranks itself (a child process running
torch.distributed.run); under a launcher it refuses a --gpus that contradicts WORLD_SIZE."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _run(args, env_extra, timeout=300):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks_and_rank0_prints_one_line():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"RT_BENCH_DRY_RUN": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1


def test_gpus_1_runs_in_process():
    r = _run(["--gpus", "1"], {"RT_BENCH_DRY_RUN": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["n_gpus"] == 1


def test_gpus_that_contradicts_the_launcher_is_refused():
    r = _run(["--gpus", "4"], {"RT_BENCH_DRY_RUN": "1", "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=2" in r.stderr


@pytest.mark.gpu
def test_gpus_2_rehearsal_on_one_gpu_reports_two_ranks(tmp_path):
    """Two ranks sharing the one GPU of the box, gather over gloo (RT_BENCH_BACKEND=gloo): the N > 1 code path of the
    benchmark end to end on a small frame."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "1", "--size", "256", "--spp", "4", "--grid", "96", "--cpu-seconds", "0",
              "--scene-dir", str(tmp_path)], {"RT_BENCH_BACKEND": "gloo"}, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["width"] == 256
    # `value` counts BVH traversal queries only; the wider definitions ride along
    assert line["value"] == line["mrays_bvh_traversed_per_s"] <= line["mrays_geometry_answered_per_s"]
    # what a slow multi-GPU run is read from: the gather's share and every rank's own times
    mg = line["multi_gpu"]
    assert mg["world_size"] == 2 and mg["backend"] == "gloo" and mg["gather_ms"] >= 0 and mg["gather_bytes_per_rank"] == 256 * 256 * 16 // 2
    for key in ("wall_ms_per_step", "gather_ms", "msTotal", "msExtend", "msShadow", "msShade", "msDirect"):
        assert len(mg["per_rank"][key]) == 2 and all(v >= 0 for v in mg["per_rank"][key]), key
    # the scene was built once: rank 0 reports the build, the flattened arrays are on disk for the other rank
    assert line["scene_build_s"] > 0 and os.path.exists(os.path.join(str(tmp_path), "flattened_w2", "desc.json"))
    rl = line["roofline"]
    assert rl["bound"].startswith("latency") and rl["served_by"] == "L1/L2" and "live" in rl and "replayed_from" in rl
