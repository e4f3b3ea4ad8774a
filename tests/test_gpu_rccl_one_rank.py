"""The N > 1 step on the device, as far as one GPU can show it: a ONE-rank RCCL process group
(`backend="nccl"` IS RCCL on ROCm) around `ShardedTracer` -- the collectives really run
(all-gather of the 4 KB slot block, the literal gather of hits, the all-reduce of the fused
spot's seven doubles), on the product's kernels, with the step's record block REUSED and placed.
The two-rank arithmetic of the shards is the gloo tests' business (tests/test_distributed_cpu.py).
"""

import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

# Opt-in (OPTILAND_TEST_RCCL=1, `tools/gpu_rccl_one_rank.sh`): run BY ITSELF on a fresh box it
# passes in 8 s (round 4, profiles/r04_rccl_one_rank.txt); run as part of the whole `-m gpu`
# suite -- the parent pytest process then holds a HIP context and tens of GB of cached device
# memory while the spawned worker brings up RCCL on the same GPU -- the worker did not answer
# within 280 s (one observation, the last GPU minutes of the round: not diagnosed; the parent now
# empties its cache first and reports a dead worker at once).  Until it has been seen green
# inside the suite, the suite does not depend on it.
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("OPTILAND_TEST_RCCL") != "1",
                                 reason="opt-in: OPTILAND_TEST_RCCL=1 (see the module comment)")]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import optiland_amd.tracer as tr
        from optiland_amd import load_system
        from optiland_amd.distributed import ShardedTracer
        table = load_system("double_gauss")
        n = 1_000_003  # ragged; above the 256 MB below which a block is never placed (fp32: 416 MB)
        g = torch.Generator(device="cuda").manual_seed(3)
        r = torch.rand(n, generator=g, device="cuda").sqrt()
        th = 2 * np.pi * torch.rand(n, generator=g, device="cuda")
        px, py = (r * th.cos()).float(), (r * th.sin()).float()
        t = tr.HipRayTracer(table, "cuda:0", dtype=torch.float32)
        st = ShardedTracer(t)
        block = st.alloc_field_record(n)
        outs = [st.trace_field(0.0, 0.7, px, py, 0.5876, center=(0.0, 15.0), record=block)
                for _ in range(3)]
        res = outs[-1]["result"]
        same_block = res.record.data_ptr() == block.data_ptr()
        fresh = st.trace_field(0.0, 0.7, px, py, 0.5876, center=(0.0, 15.0))
        equal = bool(torch.equal(fresh["result"].record[:, :, :n].nan_to_num(),
                                 block[:, :, :n].nan_to_num()))
        x, y, i = (res.record[-1, k, :n].double() for k in (0, 1, 6))
        m = i > 0
        want = (int(m.sum()), float(x[m].mean()), float(y[m].mean()))
        spot = outs[-1]["spot"]
        fs = st.trace_spot(0.0, 0.7, px, py, 0.5876, center=(0.0, 15.0))
        gen = st.trace_generic(torch.zeros(1000, device="cuda"), torch.full((1000,), 0.7,
                                                                            device="cuda"),
                               px[:1000], py[:1000], 0.5876, exchange="gather")
        hits_ok = bool(torch.equal(gen["hits"][0].cpu().nan_to_num(),
                                   gen["rays"].x.cpu().nan_to_num()))
        q.put(dict(same_block=same_block, equal=equal, want=want,
                   got=(spot["count"], spot["centroid"][0], spot["centroid"][1]),
                   fused=(fs["count"], fs["centroid"][0], fs["centroid"][1]), hits_ok=hits_ok))
    except Exception as exc:  # noqa: BLE001 - reported by the parent
        q.put(dict(error=repr(exc)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_field_step_on_a_one_rank_rccl_group():
    import queue
    import time

    # (inside the whole suite this process holds tens of GB of cached device memory: given back
    # first -- the worker brings up RCCL and a 40 GiB placement arena on the same GPU)
    torch.cuda.empty_cache()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    out, t_end = None, time.time() + 280
    while out is None and time.time() < t_end:
        try:
            out = q.get(timeout=2)
        except queue.Empty:
            if not p.is_alive():  # died without a word (an abort inside the runtime): say so now
                pytest.fail(f"the worker exited with code {p.exitcode} before reporting")
    assert out is not None, "the worker did not answer within 280 s"
    p.join(60)
    assert "error" not in out, out
    assert p.exitcode == 0
    assert out["same_block"] and out["equal"] and out["hits_ok"]
    assert out["got"][0] == out["want"][0] == out["fused"][0]
    np.testing.assert_allclose(out["got"][1:], out["want"][1:], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(out["fused"][1:], out["want"][1:], rtol=1e-5, atol=1e-6)
