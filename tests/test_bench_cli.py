"""bench.py's launcher contract, the part that needs no GPU: `--gpus N` with fewer than N visible
devices must refuse to run (never a silent N = 1 run under an N-GPU label), whether or not a launcher set WORLD_SIZE."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES",
              "CUDA_VISIBLE_DEVICES"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                          timeout=300)


def _visible_gpus():
    import torch

    return torch.cuda.device_count()


def test_more_gpus_than_visible_is_refused():
    n = _visible_gpus() + 1
    r = _run(["--gpus", str(n), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "refusing to run" in (r.stderr + r.stdout)
    assert '"metric"' not in r.stdout  # no bench line under a wrong label


def test_refused_under_a_launcher_too():
    n = _visible_gpus() + 1
    r = _run(["--gpus", str(n), "--steps", "1", "--warmup", "0"],
             {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    assert r.returncode != 0
    assert "refusing to run" in (r.stderr + r.stdout)


def test_zero_gpus_is_an_error():
    r = _run(["--gpus", "0"])
    assert r.returncode != 0


def test_pinned_rank_without_a_device_is_refused():
    """A launcher that pins one device per rank (each rank sees 1 device of an N-GPU job) is trusted up to the rank count the
    path's all-reduce returns; a rank that sees NO device still refuses at once."""
    if _visible_gpus() != 0:
        import pytest

        pytest.skip("needs a host without GPUs")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"],
             {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29534", "HIP_VISIBLE_DEVICES": "0"})
    assert r.returncode != 0
    assert "refusing to run" in (r.stderr + r.stdout)
