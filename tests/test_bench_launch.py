"""bench.py --gpus N starts its own ranks: the launch, sharding and gather without GPU work (dry run, gloo) here;
the real two-rank run on one GPU in the gpu-marked test."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv, timeout=300):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, timeout=timeout,
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_self_launch_two_ranks_dry_run():
    r = _run({"STX_BENCH_DRYRUN": "1"}, "--gpus", "2", "--steps", "3", "--warmup", "0")
    assert r["n_gpus"] == 2 and r["world_size_seen"] == 2 and r["videos_in_order"]
    # videos 0..5 present once each: the mean of the video ids
    assert r["mean_texture_score"] == pytest.approx(2.5)


@pytest.mark.gpu
def test_two_ranks_share_one_gpu():
    r = _run({"STX_BENCH_BACKEND": "gloo", "STX_BENCH_DEVICE": "0"}, "--gpus", "2", "--steps", "2", "--warmup", "1",
             "--frames", "6", "--height", "270", "--width", "480", "--no-cpu-baseline")
    assert r["n_gpus"] == 2 and r["world_size_seen"] == 2 and r["backend"] == "gloo"
    assert r["value"] > 0 and 0.0 <= r["mean_texture_score"] <= 1.0
    one = _run({}, "--gpus", "1", "--steps", "4", "--warmup", "1", "--frames", "6", "--height", "270", "--width", "480",
               "--no-cpu-baseline")
    # the same four videos (ids 0..3) whether one rank scores them or two: same mean score
    assert one["mean_texture_score"] == pytest.approx(r["mean_texture_score"], rel=1e-12)
