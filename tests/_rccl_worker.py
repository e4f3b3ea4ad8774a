"""The worker of tests/test_gpu_rccl_one_rank.py, run as a SCRIPT in its own interpreter
(`python tests/_rccl_worker.py <port> <result.json>`): a ONE-rank RCCL process group
(`backend="nccl"` IS RCCL on ROCm) around `ShardedTracer` on the product's kernels.

Why a script and not a `multiprocessing` spawn of a function of the test module: the spawned
child has to IMPORT `tests.test_gpu_rccl_one_rank` to unpickle its target, with the parent's
`sys.path` -- and inside the whole suite an earlier module has put the staged reference
(`oracle/_ref`, which has a `tests` package of its own) in front of the repository, so the
child died with `ModuleNotFoundError` before it said a word.  Round 4 saw that as "the worker
did not answer within 280 s" (the parent then had no liveness check) and made the test opt-in;
round 5 found it (profiles/r05_rccl_in_suite.txt).
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(port, out_path):
    # OPTILAND_RCCL_WORKER_LOG=<file>: phase stamps and, every 30 s, the Python stacks of every
    # thread of this worker -- what a hang would be diagnosed from
    log = os.environ.get("OPTILAND_RCCL_WORKER_LOG")
    stamp = lambda *_a: None  # noqa: E731
    if log:
        import faulthandler
        fh = open(log, "a", buffering=1)
        faulthandler.dump_traceback_later(30, repeat=True, file=fh)
        t00 = time.time()

        def stamp(what):
            fh.write(f"[{time.time() - t00:8.2f} s] {what}\n")
    stamp("worker started")
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    stamp(f"device set; free/total = {torch.cuda.mem_get_info(0)}")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    stamp("process group up")
    out = {}
    try:
        import optiland_amd.tracer as tr
        from optiland_amd import load_system
        from optiland_amd.distributed import ShardedTracer
        # the collectives must really run in a one-rank group: count the calls that reach RCCL
        calls = {"all_reduce": 0, "all_gather": 0, "all_gather_into_tensor": 0}
        for name in list(calls):
            orig = getattr(dist, name)

            def counted(*a, _orig=orig, _name=name, **k):
                calls[_name] += 1
                return _orig(*a, **k)
            setattr(dist, name, counted)
        table = load_system("double_gauss")
        n = 1_000_003  # ragged; above the 256 MB below which a block is never placed (fp32: 416 MB)
        g = torch.Generator(device="cuda").manual_seed(3)
        r = torch.rand(n, generator=g, device="cuda").sqrt()
        th = 2 * np.pi * torch.rand(n, generator=g, device="cuda")
        px, py = (r * th.cos()).float(), (r * th.sin()).float()
        t = tr.HipRayTracer(table, "cuda:0", dtype=torch.float32)
        st = ShardedTracer(t)
        stamp("tracer built")
        block = st.alloc_field_record(n)
        stamp("record block allocated")
        outs = [st.trace_field(0.0, 0.7, px, py, 0.5876, center=(0.0, 15.0), record=block)
                for _ in range(3)]
        stamp("three field steps done")
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
        stamp("generic step with gather done")
        hits_ok = bool(torch.equal(gen["hits"][0].cpu().nan_to_num(),
                                   gen["rays"].x.cpu().nan_to_num()))
        red = st.trace_generic(torch.zeros(1000, device="cuda"), torch.full((1000,), 0.7,
                                                                            device="cuda"),
                               px[:1000], py[:1000], 0.5876, exchange="reduce")
        out = dict(same_block=same_block, equal=equal, want=want,
                   got=(spot["count"], spot["centroid"][0], spot["centroid"][1]),
                   fused=(fs["count"], fs["centroid"][0], fs["centroid"][1]), hits_ok=hits_ok,
                   reduce_count=red["spot"]["count"], calls=calls,
                   world=dist.get_world_size(), backend=dist.get_backend())
    except Exception as exc:  # noqa: BLE001 - reported by the parent
        import traceback
        out = dict(error=repr(exc), traceback=traceback.format_exc())
    finally:
        with open(out_path, "w") as f:
            json.dump(out, f)
        stamp("destroying the process group")
        dist.destroy_process_group()
        stamp("worker done")


if __name__ == "__main__":
    main(int(sys.argv[1]), sys.argv[2])
