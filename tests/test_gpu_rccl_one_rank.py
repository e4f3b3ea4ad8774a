"""The N > 1 step on the device, as far as one GPU can show it: a ONE-rank RCCL process group
(`backend="nccl"` IS RCCL on ROCm) around `ShardedTracer` -- the collectives really run
(`distributed._exchanging`: whenever a process group exists, also a group of one; the worker
counts the calls that reach `torch.distributed`): the all-reduce of the fused spot's seven
doubles, the literal all-gather of hits, the two small all-reduces of `spot_statistics` -- on
the product's kernels, with the step's record block REUSED and placed.  The two-rank arithmetic
of the shards is the gloo tests' business (tests/test_distributed_cpu.py).

Part of the default `-m gpu` run again (round 5).  Round 4's "hang inside the suite" was the
spawned worker dying at import (see tests/_rccl_worker.py); the worker is now a script in its
own interpreter.
"""

import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
def test_sharded_field_step_on_a_one_rank_rccl_group(tmp_path):
    import torch

    # (inside the whole suite this process holds tens of GB of cached device memory: given back
    # first -- the worker brings up RCCL and places a record block on the same GPU)
    torch.cuda.empty_cache()
    result = tmp_path / "rccl_worker.json"
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    wait = float(os.environ.get("OPTILAND_TEST_RCCL_WAIT", "240"))
    try:
        proc = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_worker.py"),
                               str(_free_port()), str(result)], cwd=ROOT, env=env,
                              capture_output=True, text=True, timeout=wait)
    except subprocess.TimeoutExpired as exc:
        pytest.fail(f"the worker did not finish within {wait:.0f} s; stderr: "
                    f"{(exc.stderr or b'')[-2000:]!r}")
    assert result.exists(), f"worker exit code {proc.returncode}: {proc.stderr[-3000:]}"
    out = json.loads(result.read_text())
    assert "error" not in out, out
    assert proc.returncode == 0, proc.stderr[-3000:]
    assert out["backend"] == "nccl" and out["world"] == 1
    # the collectives RAN: 4 field steps x 2 + trace_spot x 2 + reduce x 2 all-reduces, the
    # gather's size exchange and the gather itself
    assert out["calls"]["all_reduce"] >= 12, out["calls"]
    assert out["calls"]["all_gather"] >= 1 and out["calls"]["all_gather_into_tensor"] >= 1
    assert out["same_block"] and out["equal"] and out["hits_ok"]
    assert out["got"][0] == out["want"][0] == out["fused"][0]
    assert out["reduce_count"] > 0
    np.testing.assert_allclose(out["got"][1:], out["want"][1:], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(out["fused"][1:], out["want"][1:], rtol=1e-5, atol=1e-6)
