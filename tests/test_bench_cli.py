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


def _json_line(out):
    import json
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (out.stdout[-800:], out.stderr[-800:])
    return json.loads(lines[0])


def test_two_rank_plumbing_check_reports_both_exchange_kinds():
    """`--plumbing-check`: the launch / sharding / exchange / JSON plumbing of the N > 1 path
    on CPU tensors with gloo and the host build of the kernel source (TEST ONLY -- the line
    says so).  One run at N = 2 shows the reduce-first exchange AND the literal all-gather of
    image-plane hits, rank 0 prints exactly one line."""
    import pytest
    from tests import _hostmath
    if not _hostmath.available():
        pytest.skip("hipcc (used as host C++ compiler) missing")
    out = _run(["--plumbing-check", "--gpus", "2", "--rays", "1500", "--steps", "2",
                "--warmup", "1"])
    assert out.returncode == 0, out.stderr[-800:]
    d = _json_line(out)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    assert d["data"].startswith("plumbing-check")
    assert d["config"]["rays_total"] == 3000 and d["config"]["mode"] == "gen"
    assert d["scaling"] == "weak" and d["higher_is_better"] is True
    ex = d["exchange"]
    assert "spot-moment" in ex["kind"] and "all-gather of image-plane hits" in ex["other"]["kind"]
    for k in ("ms_per_step_with", "ms_per_step_without", "exchange_ms_per_step",
              "bytes_per_rank_per_step"):
        assert k in ex
    assert ex["other"]["bytes_per_rank_per_step"] == 3 * 4 * 1500
    # the strong-scaled C3 configuration shards the total over the ranks
    out = _run(["--plumbing-check", "--gpus", "2", "--config", "c3", "--total-rays", "3001",
                "--steps", "1", "--warmup", "0", "--exchange", "gather"])
    assert out.returncode == 0, out.stderr[-800:]
    d = _json_line(out)
    assert d["scaling"] == "strong" and d["dtype"] == "f64"
    assert d["config"]["rays_total"] == 3001 and d["config"]["rays_per_gpu"] in (1500, 1501)
    assert "all-gather of image-plane hits" in d["exchange"]["kind"]
    assert "spot-moment" in d["exchange"]["other"]["kind"]


def test_single_rank_line_has_the_contract_keys():
    import pytest
    from tests import _hostmath
    if not _hostmath.available():
        pytest.skip("hipcc (used as host C++ compiler) missing")
    for mode in ("gen", "record", "last", "spot", "opd"):
        out = _run(["--plumbing-check", "--rays", "800", "--steps", "1", "--warmup", "0",
                    "--mode", mode])
        assert out.returncode == 0, (mode, out.stderr[-800:])
        d = _json_line(out)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                  "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                  "roofline", "cpu_baseline"):
            assert k in d, (mode, k)
        assert d["config"]["mode"] == mode and d["vs_baseline"] is None


def test_eight_rank_plumbing_check_shards_a_total_that_does_not_divide():
    """The launch form the driver uses for its scaling record, at the node's full width: 8 ranks
    (gloo, host build of the kernel source -- TEST ONLY), the strong-scaled C3 configuration
    with a total the ranks do not divide, the literal all-gather of image-plane hits; and the
    weak-scaled headline configuration with the reduce-first exchange.  Rank 0 prints exactly
    one line; every rank's shard and placement are in it."""
    import pytest
    from tests import _hostmath
    if not _hostmath.available():
        pytest.skip("hipcc (used as host C++ compiler) missing")
    out = _run(["--plumbing-check", "--gpus", "8", "--config", "c3", "--total-rays", "20003",
                "--steps", "2", "--warmup", "1", "--exchange", "gather"])
    assert out.returncode == 0, out.stderr[-1200:]
    d = _json_line(out)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["dtype"] == "f64"
    assert d["config"]["rays_total"] == 20003
    shards = d["config"]["rays_per_rank"]
    assert len(shards) == 8 and sum(shards) == 20003 and max(shards) - min(shards) <= 1
    assert shards == sorted(shards, reverse=True)          # contiguous shards, the long ones first
    assert "all-gather of image-plane hits" in d["exchange"]["kind"]
    assert len(d["roofline"]["record_placement_ranks"]) == 8
    out = _run(["--plumbing-check", "--gpus", "8", "--rays", "700", "--steps", "2",
                "--warmup", "1"])
    assert out.returncode == 0, out.stderr[-1200:]
    d = _json_line(out)
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["rays_total"] == 5600
    assert d["config"]["rays_per_rank"] == [700] * 8 and "spot-moment" in d["exchange"]["kind"]
