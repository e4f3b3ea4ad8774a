"""CPU, build container only (needs /root/reference): pack the random lenses of the
differential fuzz (tests/test_reference_fuzz.py, seeds LO..HI) into system tables under
fuzz_tables/ (untracked scratch that travels to the GPU box with the snapshot), for
tools/gpu_fuzz_tables.py to hold the KERNEL to the oracle on."""
import importlib.util
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "refshim"), "/root/reference"]
warnings.filterwarnings("ignore")
import optiland.backend as be  # noqa: E402

from optiland_amd.packer import pack_optic  # noqa: E402

be.set_backend("numpy")
spec = importlib.util.spec_from_file_location("rf", os.path.join(ROOT, "tests", "test_reference_fuzz.py"))
rf = importlib.util.module_from_spec(spec)
spec.loader.exec_module(rf)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
out = os.path.join(ROOT, "fuzz_tables")
os.makedirs(out, exist_ok=True)
for seed in range(lo, hi):
    lens, _ = rf.build_random_lens(seed, be)
    pack_optic(lens, wavelengths=[float(lens.primary_wavelength)], name=f"fuzz{seed}").save(
        os.path.join(out, f"fuzz_{seed:04d}.json"))
print("wrote", hi - lo, "tables to", out)
