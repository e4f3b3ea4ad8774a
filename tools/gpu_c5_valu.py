"""Round 6: where the vector instructions of the C5 launch go, measured.

The polarised generating launch of configuration C5 (Zernike freeform + Fresnel coatings, 1e7
rays fp32 / fp64, every row recorded, write-only PRT) with the Zernike surface's iteration cap
set to 0, 1, 2, 3, 100 -- and with / without the update_intensity epilogue.  Run under
`rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES` (tools/gpu_r06.sh c5_valu groups the
dispatches by the VARIANT lines printed here, LAUNCHES per variant): the differences between
caps are the dynamic cost of one Newton evaluation and the number of evaluations a wave takes,
to be held against the static rows of tools/phase_costs.py."""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from optiland_amd import load_system, system as S  # noqa: E402
from optiland_amd.engine import HipSystem  # noqa: E402

LAUNCHES = 3
dev = torch.device("cuda", 0)
base = load_system("zernike_fresnel_fringe")
n = 10_000_000
STATE = {"is_polarized": False, "Ex": None, "Ey": None, "phase_x": None, "phase_y": None}


def variants():
    for cap in (0, 1, 2, 3, 100):
        t = copy.deepcopy(base)
        t.surfaces["max_iter"] = np.where(t.surfaces["geom_kind"] == S.GEOM_ZERNIKE, cap, 0)
        yield f"max_iter={cap}", t, None
    yield "max_iter=100 + update_intensity epilogue", copy.deepcopy(base), STATE
    # one surface at a time switched to record-only (its surface_step is skipped, its row is
    # still written): the dynamic cost of that surface INSIDE the real kernel -- spills, copies
    # at the joins of the geometry / coating branches and all
    for idx, what in ((1, "Zernike surface"), (2, "spherical surface"), (3, "image plane")):
        t = copy.deepcopy(base)
        t.surfaces["interaction"][idx] = S.INTERACT_RECORD_ONLY
        yield f"surface {idx} ({what}) record-only", t, None
    t = copy.deepcopy(base)
    t.surfaces["coating_kind"][:] = S.COAT_NONE      # identity Jones: triads + PRT update, no Fresnel
    yield "uncoated (polarised: identity Jones matrices)", t, None
    t = copy.deepcopy(base)
    t.surfaces["interaction"][1:] = S.INTERACT_RECORD_ONLY
    yield "every surface record-only (prologue + rows + PRT stores)", t, None


for dtype in (torch.float32, torch.float64):
    g = torch.Generator(device=dev).manual_seed(1)
    r = torch.rand(n, generator=g, device=dev).sqrt()
    th = 2 * np.pi * torch.rand(n, generator=g, device=dev)
    px, py = (r * th.cos()).to(dtype), (r * th.sin()).to(dtype)
    for name, t, state in variants():
        hip = HipSystem(t, dev)
        rec = hip.alloc_record(n, dtype)
        prt = torch.empty((9, n), dtype=dtype, device=dev)
        for _ in range(LAUNCHES):
            hip.trace_generate(px, py, 0, field=(0.0, 1.0), record=rec, prt=prt,
                               update_intensity=state, defer_status=True)
        torch.cuda.synchronize()
        print("VARIANT", str(dtype).split(".")[1], name, flush=True)
        hip.close()
        del rec, prt
    # the same system without polarisation (coatings stripped, no PRT): what ALL of the
    # polarised work costs -- another kernel (POLK = 0), same Newton family
    t = copy.deepcopy(base)
    t.surfaces["coating_kind"][:] = S.COAT_NONE
    t.polarization = None
    hip = HipSystem(t, dev)
    rec = hip.alloc_record(n, dtype)
    for _ in range(LAUNCHES):
        hip.trace_generate(px, py, 0, field=(0.0, 1.0), record=rec, defer_status=True)
    torch.cuda.synchronize()
    print("VARIANT", str(dtype).split(".")[1], "unpolarised (no PRT at all)", flush=True)
    hip.close()
    del rec
