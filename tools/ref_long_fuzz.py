"""CPU, build container only: the differential fuzz families of tests/test_reference_fuzz.py
over a long seed range (python tools/ref_long_fuzz.py LO HI).  Last run (round 1): seeds
0..400 of the five stand-alone families and 150..1000 of reference vs packer + oracle:
0 failures."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "refshim"), "/root/reference"]
import warnings
warnings.filterwarnings("ignore")
import optiland.backend as be
be.set_backend("numpy")
import importlib.util
spec = importlib.util.spec_from_file_location("rf", os.path.join(ROOT, "tests", "test_reference_fuzz.py"))
rf = importlib.util.module_from_spec(spec); spec.loader.exec_module(rf)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3 and sys.argv[3] == "kernel-source":
    # round 5: the SAME families with the product's engine class on the host build of the kernel
    # source (tests/_hostmath.py) in place of the oracle-backed stand-in: live reference vs kernel
    # arithmetic, lens by lens (the tolerances are the oracle run's)
    import tests._fake_engine as _fe
    from tests import _hostmath as _hm
    _cls = _hm.make_engine_class()
    _fe.OracleEngine = lambda table, device="cpu": _cls(table, device)
    print("# engine: host build of the kernel source")
for name, extra in (("test_random_reference_lens_equals_packer_plus_oracle", ()),
                    ("test_standalone_tracer_on_random_lenses", ()),
                    ("test_standalone_spot_diagram_on_random_lenses", ("chief_ray",)),
                    ("test_standalone_spot_diagram_on_random_lenses", ("centroid",)),
                    ("test_standalone_opd_on_random_lenses", ("chief_ray",)),
                    ("test_standalone_opd_on_random_lenses", ("centroid_sphere",)),
                    ("test_standalone_opd_on_random_lenses", ("best_fit_sphere",)),
                    ("test_standalone_encircled_energy_on_random_lenses", ()),
                    ("test_standalone_irradiance_on_random_lenses", ())):
    if len(sys.argv) > 4 and sys.argv[4] not in name:
        continue   # optional 4th argument: only the families whose name contains it
    fn = getattr(rf, name)
    bad, skipped = [], 0
    for seed in range(lo, hi):
        try:
            fn(be, seed, *extra)
        except BaseException as e:
            if type(e).__name__ == "Skipped":
                skipped += 1
                continue
            bad.append((seed, type(e).__name__, str(e)[:400].replace("\n", " ")))
    print(name, extra, "checked", hi - lo, "skipped", skipped, "failures", len(bad))
    for b in bad[:6]:
        print("   ", b)
