"""`trace_generic` accepts what the reference accepts and raises what the reference raises
(raytrace/real_ray_tracer.py:120-194; VERDICT round 5, "what's weak" 1b / 1c).

Python lists / tuples die in `x >= -1` of `_validate_normalized_coordinates` (TypeError);
arrays of two or more dimensions pass `_validate_array_size` untouched and fail to broadcast
against the ray generator's flattened planes (ValueError) -- unless they are a single row.
Before round 6 the drop-in flattened both and returned rays.

The live drop-in hands such calls to the reference's own method (same exception by
construction, checked here against the reference's NumPy AND torch backends); the standalone
`HipRayTracer` raises the NumPy backend's type and text.
"""

import os
import sys

import numpy as np
import pytest

REF = os.environ.get("OPTILAND_REFERENCE", "/root/reference")
needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "optiland")),
                                     reason="reference package not present")


@pytest.fixture(scope="module")
def ref():
    shim = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refshim")
    sys.dont_write_bytecode = True
    added = [p for p in (shim, REF) if p not in sys.path]
    sys.path[:0] = added
    import optiland.backend as be
    yield be
    be.set_backend("numpy")
    for p in added:
        sys.path.remove(p)


@pytest.fixture
def host_engine(ref, monkeypatch):
    import optiland_amd.tracer as tr
    from tests import _hostmath as hm
    if not hm.available():
        pytest.skip("hipcc (used as host C++ compiler) missing")
    cls = hm.make_engine_class()
    monkeypatch.setattr(tr, "_make_engine", lambda table, device: cls(table, device))
    be = ref
    be.set_backend("torch")
    be.set_device("cpu")
    be.set_precision("float64")
    yield be
    be.set_backend("numpy")


def _cases(be):
    z33, h33 = np.zeros((3, 3)), np.full((3, 3), 0.5)
    return {
        "2-D pupil": lambda L: L.trace_generic(0.0, 0.0, be.array(z33), be.array(h33), 0.55),
        "2-D everything": lambda L: L.trace_generic(*(be.array(np.zeros((2, 3))),) * 3,
                                                    be.array(np.full((2, 3), 0.5)), 0.55),
        "3-D pupil": lambda L: L.trace_generic(0.0, 0.0, be.array(np.zeros((2, 2, 2))),
                                               be.array(np.zeros((2, 2, 2))), 0.55),
        "one row": lambda L: L.trace_generic(0.0, 0.0, be.array(np.zeros((1, 3))),
                                             be.array(np.full((1, 3), 0.5)), 0.55),
        "list pupil": lambda L: L.trace_generic(0.0, 0.0, [0.0, 0.1], [0.5, 0.5], 0.55),
        "list everything": lambda L: L.trace_generic([0.0, 0.0], [0.0, 0.0], [0.0, 0.1],
                                                     [0.5, 0.5], 0.55),
        "tuple pupil": lambda L: L.trace_generic(0.0, 0.0, (0.0, 0.1), (0.5, 0.5), 0.55),
        "tuple field": lambda L: L.trace_generic((0.0,), 0.0, be.array([0.0, 0.1]),
                                                 be.array([0.5, 0.5]), 0.55),
        "numpy scalar field": lambda L: L.trace_generic(np.float64(0.0), 0.0,
                                                        be.array([0.0, 0.1, 0.2]),
                                                        be.array([0.5, 0.5, 0.1]), 0.55),
        "0-d arrays": lambda L: L.trace_generic(0.0, 0.0, be.array(0.1), be.array(0.5), 0.55),
        "python ints": lambda L: L.trace_generic(0, 0, 0, 1, 0.55),
    }


def _outcome(be, call, lens):
    try:
        rays = call(lens)
    except Exception as exc:  # noqa: BLE001 - the outcome IS the exception
        return type(exc).__name__, str(exc)
    return "ok", np.asarray(be.to_numpy(rays.y), dtype=np.float64)


@needs_reference
@pytest.mark.parametrize("backend", ["numpy", "torch"])
def test_live_drop_in_raises_what_the_reference_raises(host_engine, backend):
    """Same exception type and text as the reference on the SAME backend (the drop-in hands the
    call to the reference's own method), same rays where the reference accepts the call."""
    be = host_engine
    from optiland.samples.objectives import CookeTriplet
    from optiland_amd import integration

    def set_backend():
        be.set_backend(backend)
        if backend == "torch":
            be.set_device("cpu")
            be.set_precision("float64")

    set_backend()
    want = {k: _outcome(be, f, CookeTriplet()) for k, f in _cases(be).items()}
    assert want["2-D pupil"][0] in ("ValueError", "RuntimeError")
    assert want["list pupil"][0] == "TypeError" and want["one row"][0] == "ok"
    be.set_backend("torch")
    be.set_device("cpu")
    be.set_precision("float64")
    lens = CookeTriplet()
    tracer = integration.install(lens, force=True)
    if backend == "numpy":
        # (the drop-in serves torch-backend optics: the NumPy column is the parity target for
        # the TYPES -- list / tuple TypeErrors are backend independent -- and for the rays)
        for k in ("list pupil", "list everything", "tuple pupil", "tuple field"):
            assert _outcome(be, _cases(be)[k], lens) == want[k], k
        for k in ("one row", "numpy scalar field", "0-d arrays", "python ints"):
            got = _outcome(be, _cases(be)[k], lens)
            assert got[0] == "ok" and np.allclose(got[1], want[k][1], rtol=0, atol=1e-12), k
        return
    for k, f in _cases(be).items():
        got = _outcome(be, f, lens)
        assert got[0] == want[k][0], (k, got, want[k])
        if got[0] == "ok":
            assert np.allclose(got[1], want[k][1], rtol=0, atol=1e-12), k
        else:
            assert got[1] == want[k][1], k
    assert tracer.last_path in ("hip", "reference")


@needs_reference
def test_standalone_tracer_raises_the_numpy_backends_exceptions(ref):
    """`HipRayTracer.trace_generic` (no reference in the process): the NumPy backend's type
    and text, decided before any engine work."""
    be = ref
    import torch
    from optiland.samples.objectives import CookeTriplet
    from optiland_amd.packer import pack_optic
    from optiland_amd.tracer import HipRayTracer

    be.set_backend("numpy")
    lens = CookeTriplet()
    table = pack_optic(lens)

    class _NoEngine:  # (the checks come first: an engine that cannot launch)
        device = torch.device("cpu")
        table = None

    front = HipRayTracer(table, dtype=torch.float64, engine=_NoEngine())
    for k in ("2-D pupil", "2-D everything", "3-D pupil", "list pupil", "list everything",
              "tuple pupil", "tuple field"):
        want = _outcome(be, _cases(be)[k], lens)
        assert want[0] in ("ValueError", "TypeError"), k
        got = _outcome(be, _cases(be)[k], front)
        assert got[0] == want[0], (k, got, want)
        if k != "3-D pupil":   # (a 3-D array fails elsewhere in the reference: type only)
            assert got[1] == want[1], (k, got, want)
