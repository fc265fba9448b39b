"""`python bench.py --gpus N` starts its N ranks by itself (the reference spawns one process
per GPU from its launcher, onmt/bin/train_wgan_wh.py:67-97, and joins them in onmt/utils/distributed.py:20-32).
CPU rehearsal: the launcher with --dry-run-gloo and 2 ranks -- both must join the group."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True,
                          env=e, timeout=300)


def test_launcher_starts_two_ranks_that_join():
    r = _run("--gpus", "2", "--dry-run-gloo")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE line, rank 0's
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["joined"] == 2
    assert out["rccl"] == {"backend": "gloo", "world_size": 2}


def test_launcher_refuses_more_ranks_than_gpus():
    """No GPU in the build container: --gpus 2 must fail loudly instead of printing an n_gpus = 1 line."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box has the GPUs")
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "refusing" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_gpus_flag_must_match_the_job():
    r = _run("--gpus", "1", "--dry-run-gloo", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["joined"] == 1
