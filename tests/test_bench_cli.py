"""bench.py launch contract (CPU): `--gpus N` must never report a smaller job under a
larger `n_gpus`."""

import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = {k: v for k, v in os.environ.items()
         if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=e,
                          capture_output=True, text=True, timeout=300)


def test_self_launch_refuses_when_devices_are_missing():
    """No launcher environment, --gpus 2, fewer than 2 HIP devices (this container has
    none): exit code 2 and a clear message, no JSON line."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box really has two devices")
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert out.returncode == 2, (out.returncode, out.stderr[-500:])
    assert "--gpus 2 requested but only" in out.stderr
    assert not out.stdout.strip().startswith("{")


def test_launcher_world_size_must_match_gpus():
    """Under a launcher (RANK / WORLD_SIZE set) a mismatch between --gpus and the ranks that
    were started is an error, not a warning."""
    out = _run(["--gpus", "4", "--steps", "1"], env={"RANK": "0", "WORLD_SIZE": "2",
                                                    "LOCAL_RANK": "0"})
    assert out.returncode == 2
    assert "--gpus 4 but the launcher started WORLD_SIZE=2" in out.stderr


def test_c3_and_default_configs_parse():
    out = _run(["--help"])
    assert out.returncode == 0
    for flag in ("--config", "--total-rays", "--exchange", "--no-ref-baselines"):
        assert flag in out.stdout
