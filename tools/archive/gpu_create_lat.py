import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optiland_amd import load_system
from optiland_amd.engine import HipSystem
torch.zeros(1, device="cuda:0")
for name in ("cooke_generic", "double_gauss", "zernike_fresnel_fringe"):
    table = load_system(name)
    for _ in range(3):
        HipSystem(table, "cuda:0").close()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        h = HipSystem(table, "cuda:0")
        h.close()
    torch.cuda.synchronize()
    print(f"{name}: create + destroy {(time.perf_counter() - t0) / 100 * 1e6:.0f} us")
    h = HipSystem(table, "cuda:0")
    if hasattr(h, "update"):
        for _ in range(5):
            h.update(table)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            h.update(table)
        host = (time.perf_counter() - t0) / 200 * 1e6
        torch.cuda.synchronize()
        print(f"{name}: ol_system_update (in place) {host:.0f} us of host time per call")
    h.close()
