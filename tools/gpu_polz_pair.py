"""Round 6: the polarised Zernike fp32 generating launch (configuration C5) on TWO rays per lane.

(a) the two forms -- one ray per lane (OL_TUNE_RAYS_PER_THREAD = 1) and the pair (= 3) -- on the
    same inputs: record block, PRT planes, updated intensity and status compared bit for bit;
(b) arm against arm in one process: ROUNDS x (60 launches of each arm back to back, the mean of
    the last 30), arms alternating, the SAME record block (placed), HIP events around each launch.

    python tools/gpu_polz_pair.py            (cycles: tools/gpu_r06.sh polz_cycles)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from optiland_amd import _capi  # noqa: E402
from optiland_amd.engine import HipSystem  # noqa: E402

ROUNDS = int(os.environ.get("ROUNDS", "4"))
N = int(float(os.environ.get("RAYS", "1e7")))
dev = torch.device("cuda", 0)
ARMS = {"one": 1, "pair": 3}
STATE = {"is_polarized": False, "Ex": None, "Ey": None, "phase_x": None, "phase_y": None}


def bits(t):
    return t.contiguous().view(torch.int32 if t.dtype == torch.float32 else torch.int64)


def main():
    table, hy, _desc, wavelength = bench.load_workload("zernike_fresnel")
    wl = table.wavelength_index(wavelength)
    hip = HipSystem(table, dev)
    dtype = torch.float32

    def tune(v):
        rc = hip.lib.ol_set_tuning(_capi.TUNE_RAYS_PER_THREAD, v)
        assert rc == 0, rc

    # (a) bits, at a ragged and an even size, with and without the epilogue
    for n in (1000, 4098, 1_000_000):
        px, py = bench.make_pupil(n, dtype, 77, dev)
        out = {}
        for arm, v in ARMS.items():
            tune(v)
            for epi in (False, True):
                prt = torch.full((9, n), float("nan"), dtype=dtype, device=dev)
                res = hip.trace_generate(px, py, wl, field=(0.0, hy), record=True, prt=prt,
                                         update_intensity=STATE if epi else None)
                torch.cuda.synchronize()
                out[arm, epi] = (res.record[:, :, :n].clone(), prt.clone(),
                                 None if not epi else res.updated_intensity.clone(),
                                 int(res.status.item()) if torch.is_tensor(res.status) else res.status)
        for epi in (False, True):
            a, b = out["one", epi], out["pair", epi]
            same_rec = torch.equal(bits(a[0]), bits(b[0]))
            same_prt = torch.equal(bits(a[1]), bits(b[1]))
            same_upd = True if a[2] is None else torch.equal(bits(a[2]), bits(b[2]))
            worst = float((a[0].double() - b[0].double()).abs().nan_to_num().max())
            print(f"n={n:8d} epilogue={epi!s:5s} record bits equal {same_rec} (max |d| {worst:.3g})"
                  f"  prt {same_prt}  updated intensity {same_upd}  status {a[3]} / {b[3]}",
                  flush=True)
    tune(0)

    # (b) time
    n = N
    px, py = bench.make_pupil(n, dtype, 1234, dev)
    # the record block with two more rows: the nine PRT planes INSIDE the placed window (arms
    # "+prt") instead of in an ordinary allocation of their own
    S_ = hip.num_surfaces
    block, info = hip.alloc_record_placed(n, dtype, rows=S_ + 2)
    rec = block[:S_]
    prt_plain = torch.empty((9, n), dtype=dtype, device=dev)
    prt_placed = block[S_:].reshape(-1)[: 9 * n].view(9, n)
    print(f"rays {n}  placed {info.get('placed')}  block {tuple(block.shape)}")
    arms = {"one": (1, prt_plain), "pair": (3, prt_plain), "one+prt": (1, prt_placed),
            "pair+prt": (3, prt_placed)}
    acc = {a: [] for a in arms}
    for r in range(ROUNDS):
        order = list(arms) if r % 2 == 0 else list(arms)[::-1]
        for arm in order:
            tune(arms[arm][0])
            prt = arms[arm][1]
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                   for _ in range(60)]
            for a, b in evs:
                a.record()
                hip.trace_generate(px, py, wl, field=(0.0, hy), record=rec, prt=prt,
                                   zero_status=False, defer_status=True)
                b.record()
            torch.cuda.synchronize()
            ms = np.array([a.elapsed_time(b) for a, b in evs])
            acc[arm].append(ms[30:].mean())
            print(f"round {r} {arm:8s} first 10 {ms[:10].mean():.4f}  10-29 {ms[10:30].mean():.4f}"
                  f"  30-59 {ms[30:].mean():.4f} ms", flush=True)
    tune(0)
    moved = (4 * 8 + 9) * 4 * n + 2 * 4 * n
    print("library", os.environ.get("OPTILAND_HIP_LIBRARY", "product"))
    for arm in arms:
        m = float(np.mean(acc[arm]))
        print(f"{arm:8s} sustained {m:.4f} ms  = {moved / m / 1e9:.2f} TB/s of bytes moved "
              f"({moved / m / 1e9 / 8.0:.3f} of 8 TB/s)")
    hip.close()


if __name__ == "__main__":
    main()
