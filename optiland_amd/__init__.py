"""optiland_amd -- MI355X-native batched sequential ray trace for Optiland.

The package holds only what the hot path needs: the HIP kernels + C ABI
(`csrc/`, built into `lib/liboptiland_hip.so`), the ctypes binding (`_capi`),
the device engine (`engine.HipSystem`), the surface-table format (`system`), the
packer that flattens a live reference `Optic` (`packer`) and the host-side mirror
of the reference's tracer interface (`tracer.HipRayTracer`, `integration`).
"""

from __future__ import annotations

import os

from .system import SystemTable

__version__ = "0.1.0"
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def available_systems():
    """Names of the packed sample systems shipped under optiland_amd/data/."""
    return sorted(os.path.splitext(f)[0] for f in os.listdir(_DATA) if f.endswith(".json"))


def load_system(name: str) -> SystemTable:
    """Load a packed sample system (exported from the reference by
    tools/make_golden.py): 'double_gauss', 'cooke_generic', 'rc_asphere',
    'zernike_fresnel_fringe'."""
    path = os.path.join(_DATA, f"{name}.json")
    if not os.path.exists(path):
        raise KeyError(f"unknown system {name!r}; have {available_systems()}")
    return SystemTable.load(path)
