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
