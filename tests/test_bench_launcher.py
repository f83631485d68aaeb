"""bench.py's rank launcher on a box without GPUs: `--gpus N` must start N ranks (here over gloo, as a dry run) and
prove the group with an all-reduce of the rank numbers -- or fail loudly; it must never run fewer ranks than asked for."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


@pytest.mark.timeout(300)
def test_gpus_2_launches_two_verified_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run-backend", "gloo"], capture_output=True, text=True,
                       env=_env(), timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                  # rank 0 prints, once
    assert lines[0]["n_gpus"] == 2 and lines[0]["ranks_verified"] == 2 and lines[0]["dry_run"] is True


@pytest.mark.skipif(torch.cuda.is_available() and torch.cuda.device_count() >= 2, reason="needs a box with fewer than 2 GPUs")
def test_gpus_2_without_two_gpus_fails_loudly():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       env=_env(), timeout=120)
    assert r.returncode != 0
    assert "GPU(s) are visible" in r.stderr and not any(l.startswith("{") for l in r.stdout.splitlines())


def test_world_size_that_disagrees_with_gpus_is_an_error():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run-backend", "gloo"], capture_output=True, text=True,
                       env=_env(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
