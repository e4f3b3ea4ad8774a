"""Drop-in integration with the LIVE reference (build container only).

Skipped where /root/reference does not exist (the GPU box).  The fused trace is
stood in for by the oracle-backed engine (tests/_fake_engine.py) so that the
HOST side of the drop-in -- packer, backend registration, `Optic.ray_tracer`
replacement, per-surface write-back, fall-back to the reference for unsupported
systems -- is exercised against the reference's own objects and consumers.
"""

import os
import sys

import numpy as np
import pytest

REF = os.environ.get("OPTILAND_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "optiland")),
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


@pytest.fixture(params=["oracle", "kernel-source"])
def hip_on_cpu(ref, monkeypatch, request):
    """torch backend, cpu, fp64 (what the reference's own conftest uses for torch).  Every
    test runs twice: the fused trace stood in for by the oracle-backed engine, and by the
    product's own `HipSystem` class on the host build of the kernel source
    (tests/_hostmath.make_engine_class) -- live reference -> drop-in -> engine -> C ABI ->
    `surface_math.h`, all on the CPU."""
    import optiland_amd.tracer as tr
    if request.param == "oracle":
        from tests._fake_engine import OracleEngine
        monkeypatch.setattr(tr, "_make_engine", lambda table, device: OracleEngine(table, device))
    else:
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


def _np(be, a):
    return np.asarray(be.to_numpy(a), dtype=np.float64)


def test_register_backend_adds_hip_entry(ref):
    from optiland_amd.integration import register_backend
    be = ref
    inst = register_backend()
    assert "hip" in be.list_available_backends()
    assert inst.name == "hip"
    be.set_backend("hip")
    assert be.get_backend() == "hip"
    x = be.array([1.0, 2.0])
    assert float(be.to_numpy(be.sqrt(x * x))[1]) == 2.0  # inherited contract works
    be.set_backend("numpy")


@pytest.mark.parametrize("sample", ["CookeTriplet", "DoubleGauss"])
def test_install_matches_reference_tracer(hip_on_cpu, sample):
    be = hip_on_cpu
    from optiland.samples import objectives
    from optiland_amd.integration import install
    lens_ref = getattr(objectives, sample)()
    lens_hip = getattr(objectives, sample)()
    tracer = install(lens_hip, force=True)
    w = 0.55 if sample == "CookeTriplet" else 0.5876
    r0 = lens_ref.trace(0.0, 0.7, w, 6, "hexapolar")
    r1 = lens_hip.trace(0.0, 0.7, w, 6, "hexapolar")
    assert tracer.last_path == "hip"
    for k in ("x", "y", "z", "L", "M", "N", "i", "opd", "L0", "M0", "N0"):
        np.testing.assert_allclose(_np(be, getattr(r1, k)), _np(be, getattr(r0, k)),
                                   rtol=1e-9, atol=1e-10, err_msg=k)
    for k in ("x", "y", "z", "L", "M", "N", "intensity", "opd"):
        a = _np(be, getattr(lens_hip.surfaces, k))
        b = _np(be, getattr(lens_ref.surfaces, k))
        assert a.shape == b.shape
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-9, err_msg=k)
    # trace_generic with arrays
    px, py = be.array([0.1, -0.3, 0.5]), be.array([0.2, 0.4, -0.6])
    g0 = lens_ref.trace_generic(0.0, 1.0, px, py, w)
    g1 = lens_hip.trace_generic(0.0, 1.0, px, py, w)
    np.testing.assert_allclose(_np(be, g1.y), _np(be, g0.y), rtol=1e-9, atol=1e-10)


def test_spot_diagram_goldens_through_the_drop_in(hip_on_cpu):
    """The reference's own end-to-end goldens (tests/test_analysis.py:76-102: Cooke
    triplet RMS / geometric spot radii, fields 0/14/20 deg, 3 wavelengths) computed
    by the reference's SpotDiagram on top of the replaced tracer."""
    be = hip_on_cpu
    from optiland import analysis
    from optiland.samples.objectives import CookeTriplet
    from optiland_amd.integration import install
    lens = CookeTriplet()  # the `cooke_triplet` fixture of tests/test_analysis.py:27-29
    tracer = install(lens, force=True)
    spot = analysis.SpotDiagram(lens)
    assert tracer.last_path == "hip"
    rms = spot.rms_spot_radius()
    geo = spot.geometric_spot_radius()
    want_rms = [[0.003791335461448, 0.004293689564257, 0.006195618755672],
                [0.01582480029344623, 0.016918412809703662, 0.019221165873836682],
                [0.013236232767092956, 0.012116688566406967, 0.013648684944411313]]
    want_geo = [[0.00597244087781, 0.00628645771124, 0.00931911440064],
                [0.03928464835617618, 0.04075295155639047, 0.04772194200606705],
                [0.018909146395329878, 0.022501847359635008, 0.036545592330568866]]
    for f in range(3):
        for w in range(3):
            np.testing.assert_allclose(float(be.to_numpy(rms[f][w])), want_rms[f][w], rtol=1e-5)
            np.testing.assert_allclose(float(be.to_numpy(geo[f][w])), want_geo[f][w], rtol=1e-5)


def test_polarized_system_through_the_drop_in(hip_on_cpu):
    """Parity target is the reference's NumPy backend (BASELINE.json north_star).
    NB: the reference's own torch backend disagrees with its NumPy backend on this
    polarised system (non-unitary PRT, e.g. p[2,2] = 1.021 vs 0.9716) -- so r0 is
    produced on the NumPy backend; see DESIGN.md "Reference findings"."""
    be = hip_on_cpu
    from optiland.rays import PolarizationState
    from optiland.samples.simple import AsphericSinglet
    from optiland_amd.integration import install

    def build():
        lens = AsphericSinglet()
        lens.surfaces.set_fresnel_coatings()
        lens.updater.set_polarization(PolarizationState(is_polarized=False))
        return lens
    be.set_backend("numpy")
    a = build()
    r0 = a.trace(0.0, 0.0, 0.587, 8, "uniform")
    i0, p0 = np.array(r0.i), np.array(r0.p)
    r0.update_intensity(PolarizationState(is_polarized=True, Ex=1, Ey=0, phase_x=0, phase_y=0))
    i0x = np.array(r0.i)
    be.set_backend("torch")
    b = build()
    tracer = install(b, force=True)
    r1 = b.trace(0.0, 0.0, 0.587, 8, "uniform")
    assert tracer.last_path == "hip"
    assert type(r1).__name__ == "PolarizedRays"
    np.testing.assert_allclose(_np(be, r1.i), i0, rtol=1e-8)
    # `p` in the reference's (N, 3, 3) complex layout is produced from the kernel's planes on
    # first read (integration._LazyPrt); copies taken before that read carry the planes
    import copy
    assert "p" not in vars(r1) and "_hip_prt" in vars(r1)
    twin, deep = copy.copy(r1), copy.deepcopy(r1)
    assert hasattr(r1, "p") and "p" in vars(r1) and "_hip_prt" not in vars(r1)
    assert r1.p.shape == (p0.shape[0], 3, 3) and r1.p.is_complex()
    np.testing.assert_allclose(be.to_numpy(r1.p).real, p0.real, rtol=1e-7, atol=1e-10)
    for other in (twin, deep):
        assert "p" not in vars(other)
        np.testing.assert_array_equal(be.to_numpy(other.p), be.to_numpy(r1.p))
    twin.p = twin.p * 2                                   # a write is a plain attribute write
    np.testing.assert_array_equal(be.to_numpy(twin.p), 2 * be.to_numpy(r1.p))
    del deep.p
    assert not hasattr(deep, "p")
    # the tutorial recipe: repeated update_intensity on the returned object (this
    # runs the REFERENCE's PolarizedRays.update_intensity on our p / _i0 / _L0..)
    r1.update_intensity(PolarizationState(is_polarized=True, Ex=1, Ey=0, phase_x=0, phase_y=0))
    np.testing.assert_allclose(_np(be, r1.i), i0x, rtol=1e-8)


def test_unsupported_system_falls_back_to_reference(hip_on_cpu):
    be = hip_on_cpu
    from optiland.samples.objectives import CookeTriplet
    from optiland_amd.integration import install
    lens = CookeTriplet()
    lens.surfaces[3].geometry.__class__ = type("ForbesQbfsGeometry", (lens.surfaces[3].geometry.__class__,), {})
    tracer = install(lens, force=True)
    r = lens.trace(0.0, 0.0, 0.55, 4, "hexapolar")
    assert tracer.last_path == "reference"
    assert _np(be, r.x).shape == (61,)


def test_not_intercepted_on_numpy_backend(ref):
    be = ref
    be.set_backend("numpy")
    from optiland.samples.objectives import CookeTriplet
    from optiland_amd.integration import install
    lens = CookeTriplet()
    tracer = install(lens)  # no force
    lens.trace(0.0, 0.0, 0.55, 4, "hexapolar")
    assert tracer.last_path == "reference"


def test_packer_matches_committed_tables(ref):
    """The JSON fixtures used on the GPU box are what the packer produces today."""
    be = ref
    be.set_backend("numpy")
    from optiland.samples.objectives import DoubleGauss
    from optiland_amd.packer import pack_optic
    from tests._util import load_case
    table, _ = load_case("double_gauss")
    fresh = pack_optic(DoubleGauss(), wavelengths=[0.5876])
    assert fresh.surfaces.tobytes() == table.surfaces.tobytes()
    np.testing.assert_array_equal(fresh.optics, table.optics)
    assert fresh.raygen == table.raygen


def test_enable_routes_every_optic_and_disable_restores(hip_on_cpu):
    be = hip_on_cpu
    from optiland.raytrace.real_ray_tracer import RealRayTracer
    from optiland.samples.objectives import CookeTriplet
    from optiland_amd import integration
    orig = RealRayTracer.trace
    be.set_backend("numpy")
    want = np.array(CookeTriplet().trace(0.0, 1.0, 0.55, 5, "hexapolar").y)
    be.set_backend("torch")
    integration.enable(force=True)
    try:
        lens = CookeTriplet()  # created AFTER enable(), never touched by install()
        r = lens.trace(0.0, 1.0, 0.55, 5, "hexapolar")
        comp = lens.ray_tracer._hip_companion
        assert comp.last_path == "hip"
        np.testing.assert_allclose(_np(be, r.y), want, rtol=1e-9, atol=1e-10)
        g = lens.trace_generic(0.0, 0.5, be.array([0.1, 0.2]), be.array([0.0, -0.3]), 0.55)
        assert comp.last_path == "hip" and _np(be, g.x).shape == (2,)
    finally:
        integration.disable()
    assert RealRayTracer.trace is orig
    lens = CookeTriplet()
    lens.trace(0.0, 1.0, 0.55, 5, "hexapolar")
    assert "_hip_companion" not in lens.ray_tracer.__dict__


def test_reference_suite_sweeps_the_hip_backend(tmp_path):
    """The reference's conftest parametrises every test over
    `be.list_available_backends()` (tests/conftest.py:8-23): registering "hip" before
    collection makes its own suite sweep the new backend.  Run its backend-contract
    tests and the hot-path unit tests that way (CPU, fp64).  One test is
    deselected: `test__process_input` checks `be.ndarray`, which the reference picks
    by backend NAME."""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dst = tmp_path / "tests"
    shutil.copytree(os.path.join(REF, "tests"), dst,
                    ignore=shutil.ignore_patterns("__pycache__", "zemax_files", "*.zmx"))
    for base_, dirs_, files_ in os.walk(dst):   # (a read-only reference tree keeps its modes)
        for name_ in dirs_ + files_:
            os.chmod(os.path.join(base_, name_),
                     os.stat(os.path.join(base_, name_)).st_mode | 0o200)
    conf = (dst / "conftest.py").read_text()
    conf = conf.replace(
        "import optiland.backend as be\n",
        "import optiland.backend as be\nimport sys\nsys.path.insert(0, %r)\n"
        "from optiland_amd.integration import register_backend\nregister_backend()\n" % root, 1)
    conf = conf.replace('if backend_name == "torch":', 'if backend_name in ("torch", "hip"):')
    (dst / "conftest.py").write_text(conf)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1",
               PYTHONPATH=os.pathsep.join([os.path.join(root, "tests", "refshim"), REF]))
    files = ["tests/test_backend_contract.py", "tests/test_rays.py",
             "tests/test_coordinate_system.py", "tests/test_physical_apertures.py",
             "tests/test_coatings.py", "tests/test_jones.py", "tests/test_standard_surface.py",
             "tests/test_surface_group.py", "tests/test_zernike.py"]
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider",
                          "-k", "hip and not test__process_input", *files],
                         cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    tail = out.stdout.strip().splitlines()[-1]
    assert out.returncode == 0, out.stdout[-3000:]
    assert " passed" in tail and "failed" not in tail, tail
    n = int(tail.split(" passed")[0].split()[-1])
    assert n > 300, tail


@pytest.mark.parametrize("engine", ["oracle", "kernel-source"])
def test_reference_consumers_run_on_the_replaced_tracer(tmp_path, engine):
    """The reference's OWN tests of the consumers of the path -- Optic.trace /
    trace_generic, spot diagrams, encircled energy, irradiance, ray fans, wavefront / OPD /
    Zernike fits -- executed with `integration.enable(force=True)`: every real-ray trace
    inside them goes through the drop-in tracer and their hard-coded expectations still hold.
    Behind the tracer: the oracle-backed stand-in, or the product's engine class on the host
    build of the kernel source through the real C ABI (tests/_hostmath.make_engine_class).
    Autograd tests are deselected: the drop-in must not intercept differentiable traces."""
    if engine == "kernel-source":
        from tests import _hostmath as hm
        if not hm.available():
            pytest.skip("hipcc (used as host C++ compiler) missing")
        hm.load()  # build once, here, not inside the reference's test session
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dst = tmp_path / "tests"
    shutil.copytree(os.path.join(REF, "tests"), dst,
                    ignore=shutil.ignore_patterns("__pycache__"))
    for base_, dirs_, files_ in os.walk(dst):   # (a read-only reference tree keeps its modes)
        for name_ in dirs_ + files_:
            os.chmod(os.path.join(base_, name_),
                     os.stat(os.path.join(base_, name_)).st_mode | 0o200)
    counter = tmp_path / "hip_engines.txt"
    conf = (dst / "conftest.py").read_text()
    conf = conf.replace(
        "import optiland.backend as be\n",
        "import optiland.backend as be\n"
        "import atexit, importlib.util, sys\n"
        "sys.path.insert(0, %r)\n"
        "_spec = importlib.util.spec_from_file_location('_ol_engine_mod', %r)\n"
        "_mod = importlib.util.module_from_spec(_spec)\n"
        "_spec.loader.exec_module(_mod)\n"
        "_cls = %s\n"
        "import optiland_amd.tracer as _tr\n"
        "_made = [0]\n"
        "def _mk(table, device):\n"
        "    _made[0] += 1\n"
        "    return _cls(table, device)\n"
        "_tr._make_engine = _mk\n"
        "atexit.register(lambda: open(%r, 'w').write(str(_made[0])))\n"
        "from optiland_amd import integration as _integ\n"
        "_integ.enable(force=True)\n"
        % (root,
           os.path.join(root, "tests", "_fake_engine.py" if engine == "oracle" else "_hostmath.py"),
           "_mod.OracleEngine" if engine == "oracle" else "_mod.make_engine_class()",
           str(counter)), 1)
    conf = conf.replace("be.grad_mode.enable()", "be.grad_mode.disable()")
    (dst / "conftest.py").write_text(conf)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1",
               PYTHONPATH=os.pathsep.join([os.path.join(root, "tests", "refshim"), REF]))
    files = ["tests/test_optic.py", "tests/test_analysis.py", "tests/test_wavefront.py"]
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider",
                          "-k", "torch and not autodiff", *files],
                         cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    tail = out.stdout.strip().splitlines()[-1]
    assert out.returncode == 0, out.stdout[-3000:]
    assert " passed" in tail and "failed" not in tail, tail
    assert int(tail.split(" passed")[0].split()[-1]) > 140, tail
    assert int(counter.read_text()) > 50  # the traces really went through the drop-in


def test_packer_memoises_paraxial_scalars_and_invalidates(ref, monkeypatch):
    """The reference's paraxial traces (EPL / EPD / XPL) are memoised per optic against
    a fingerprint of the first-order layout: repeated packs (one per field / wavelength
    of an analysis) call them once; any change that moves them recomputes.  (The path that
    asks the reference at all: since round 3 the packer computes these scalars itself from
    the packed table -- paraxial_host.py, tests/test_paraxial_host.py -- and only falls back
    here for systems that restatement does not cover.)"""
    be = ref
    be.set_backend("numpy")
    monkeypatch.setenv("OPTILAND_HIP_HOST_PARAXIAL", "0")
    from optiland.paraxial import Paraxial
    from optiland.samples.objectives import CookeTriplet
    from optiland_amd.packer import pack_optic
    calls = {"EPL": 0}
    orig = Paraxial.EPL

    def counted(self):
        calls["EPL"] += 1
        return orig(self)

    monkeypatch.setattr(Paraxial, "EPL", counted)
    lens = CookeTriplet()
    a = pack_optic(lens, wavelengths=[0.55])
    n0 = calls["EPL"]
    assert n0 >= 1
    for w in (0.48, 0.55, 0.65):
        b = pack_optic(lens, wavelengths=[w])
        assert b.raygen == a.raygen and b.fields == a.fields
    assert calls["EPL"] == n0                        # memo hit: no paraxial trace
    # the packed positions are SurfaceGroup.positions
    pos = np.asarray(be.to_numpy(lens.surfaces.positions), dtype=np.float64).reshape(-1)
    assert np.array_equal(a.surfaces["origin"][1:, 2], pos[1:])
    # a first-order change invalidates: radius, thickness (position), aperture value
    lens.surfaces[2].geometry.radius = float(lens.surfaces[2].geometry.radius) * 1.01
    c = pack_optic(lens, wavelengths=[0.55])
    assert calls["EPL"] > n0 and c.raygen["EPL"] != a.raygen["EPL"]
    n1 = calls["EPL"]
    lens.set_aperture(aperture_type="EPD", value=8.0)
    d = pack_optic(lens, wavelengths=[0.55])
    assert calls["EPL"] > n1 and d.raygen["EPD"] == pytest.approx(8.0)
    # against a fresh computation
    monkeypatch.setattr(Paraxial, "EPL", orig)
    assert d.raygen["EPL"] == pytest.approx(float(np.asarray(lens.paraxial.EPL()).reshape(-1)[0]))


@pytest.mark.parametrize("kind", ["uniform", "random", "cross", "ring", "line_y", "sobol"])
def test_trace_with_a_distribution_object(hip_on_cpu, kind):
    """`Optic.trace(..., distribution=<BaseDistribution>)`: the caller's own sampler object
    (reference optiland/distribution.py), points taken as they are, field x pupil order of
    real_ray_tracer.py:95-98; and every named sampler through the string form."""
    be = hip_on_cpu
    from optiland.distribution import create_distribution
    from optiland.samples.objectives import CookeTriplet
    from optiland_amd.integration import install
    lens_ref, lens_hip = CookeTriplet(), CookeTriplet()
    tracer = install(lens_hip, force=True)
    d = create_distribution(kind)
    d.generate_points(7)
    hx, hy = be.array([0.0, 0.0]), be.array([0.0, 1.0])
    r0 = lens_ref.trace(hx, hy, 0.55, None, d)
    r1 = lens_hip.trace(hx, hy, 0.55, None, d)
    assert tracer.last_path == "hip"
    for k in ("x", "y", "z", "L", "M", "N", "i", "opd"):
        np.testing.assert_allclose(_np(be, getattr(r1, k)), _np(be, getattr(r0, k)),
                                   rtol=1e-9, atol=1e-10, err_msg=k)
    if kind not in ("random", "sobol"):   # deterministic samplers: the string form too
        r2 = lens_hip.trace(0.0, 1.0, 0.55, 7, kind)
        r3 = lens_ref.trace(0.0, 1.0, 0.55, 7, kind)
        np.testing.assert_allclose(_np(be, r2.y), _np(be, r3.y), rtol=1e-9, atol=1e-10)


def test_trace_generic_input_matrix_matches_the_reference(hip_on_cpu):
    """Every input form the reference's torch backend accepts for `trace_generic` -- python
    scalars, ints, 0-d / one-element / n-element tensors in fp64 and fp32, scalar field with
    tensor pupil and the reverse, rays on the pupil rim -- gives the reference's result, and
    every range violation (field before pupil, NaN included) the reference's ValueError.
    Where the reference only fails by accident (size mismatch: a RuntimeError out of a
    tensor op; empty input: `stack expects a non-empty TensorList`; numpy arrays: TypeError)
    the drop-in is a superset: a ValueError naming the problem, an empty result, and the arrays
    accepted.  Python lists raise the reference's own TypeError."""
    import torch
    be = hip_on_cpu
    from optiland.samples.objectives import CookeTriplet
    from optiland_amd.integration import install
    ref_lens, hip_lens = CookeTriplet(), CookeTriplet()
    tracer = install(hip_lens, force=True)
    T = lambda v: torch.tensor(v, dtype=torch.float64)  # noqa: E731

    def run(lens, args):
        try:
            r = lens.trace_generic(*args, 0.55)
            return "ok", _np(be, r.y), _np(be, r.i)
        except Exception as e:  # noqa: BLE001
            return "err", type(e).__name__, str(e)
    same = [
        (0.0, 0.5, T([0.1, -0.2, 0.3]), T([0.3, 0.2, -0.1])),
        (0, 1, T([0.1, -0.2, 0.3]), T([0.3, 0.2, -0.1])),
        (T([0.0, 0.0, 0.1]), T([0.0, 0.5, 1.0]), T([0.1, -0.2, 0.3]), T([0.3, 0.2, -0.1])),
        (T(0.0), T(0.5), T(0.2), T(-0.3)),
        (T([0.0]), T([0.5]), T([0.2]), T([-0.3])),
        (0.0, 0.5, 0.2, -0.3),
        (T([0.0, 0.0]), T([0.0, 1.0]), 0.2, 0.1),
        (0.0, 0.5, torch.tensor([0.1, 0.2]), torch.tensor([0.3, -0.2])),
        (0.0, 1.0, T([1.0, -1.0, 0.0]), T([0.0, 0.0, 1.0])),
        (0.0, 0.5, T([0.1, 1.2]), T([0.1, 0.2])),                      # pupil range
        (0.0, 1.5, T([0.1, 0.2]), T([0.1, 0.2])),                      # field range
        (T([0.0, 0.0]), T([0.5, -1.5]), T([0.1, 0.2]), T([0.1, 0.2])),
        (0.0, 1.5, T([0.1, 1.2]), T([0.1, 0.2])),                      # both: field reported
        (0.0, 0.5, T([0.1, float("nan")]), T([0.1, 0.2])),
    ]
    for args in same:
        a, b = run(ref_lens, args), run(hip_lens, args)
        assert a[0] == b[0], (args, a, b)
        if a[0] == "ok":
            assert tracer.last_path == "hip"
            np.testing.assert_allclose(b[1], a[1], rtol=1e-6, atol=1e-9)
            np.testing.assert_allclose(b[2], a[2], rtol=1e-6, atol=1e-9)
        else:
            assert a[1:] == b[1:], (args, a, b)
    # a call that fails its range check leaves the optic's surfaces as the last good trace
    # left them (the reference validates before it touches anything; here the status word is
    # read back after the result objects were BUILT but before any Surface is written)
    run(hip_lens, same[0])
    held = [s_.x for s_ in hip_lens.surfaces.surfaces]
    assert run(hip_lens, (0.0, 0.5, T([0.1, 1.2, 0.0]), T([0.1, 0.2, 0.0])))[0] == "err"
    assert all(a_ is b_ for a_, b_ in zip(held, [s_.x for s_ in hip_lens.surfaces.surfaces]))
    # superset behaviour
    bad_size = run(hip_lens, (0.0, 0.5, T([0.1, 0.2]), T([0.1, 0.2, 0.3])))
    # (arrays of different lengths: NumPy's broadcast error text, the exception the reference's
    # NumPy backend ends in; its torch backend raises torch's RuntimeError for the same call)
    assert bad_size[:2] == ("err", "ValueError")
    assert bad_size[2] == "operands could not be broadcast together with shapes (2,) (3,) "
    both = run(hip_lens, (T([0.0, 0.1]), T([0.0, 0.1, 0.2]), T([0.1, 0.2]), 0.1))
    assert both[1] == "ValueError" and both[2].startswith("shape mismatch: objects cannot be")
    empty = run(hip_lens, (0.0, 0.5, T([]), T([])))
    assert empty[0] == "ok" and empty[1].size == 0
    # one-dimensional numpy arrays are taken (the NumPy backend's own input type) ...
    arr = run(hip_lens, (0.0, 0.5, np.array([0.1, -0.2]), np.array([0.3, 0.2])))
    want = run(ref_lens, (0.0, 0.5, T([0.1, -0.2]), T([0.3, 0.2])))
    np.testing.assert_allclose(arr[1], want[1], rtol=1e-9)
    # ... Python lists are NOT, like in the reference (round 6: rounds 2-5 flattened them; the
    # reference dies in `x >= -1`, tests/test_trace_generic_validation.py)
    lst = (0.0, 0.5, [0.1, -0.2], [0.3, 0.2])
    assert run(hip_lens, lst) == run(ref_lens, lst) and run(hip_lens, lst)[1] == "TypeError"


def test_tracer_stats_report_what_the_drop_in_holds(hip_on_cpu):
    """Round 6: `tracer.stats()` -- the path of the last call, packs, device tables alive, and
    the record pools' footprint (`placed_bytes`; 0 on the CPU engines of this suite)."""
    from optiland.samples.objectives import CookeTriplet
    from optiland_amd import integration
    lens = CookeTriplet()
    tracer = integration.install(lens, force=True)
    lens.trace(0.0, 0.7, 0.55, 3, "hexapolar")
    st = tracer.stats()
    assert st["last_path"] == "hip" and st["packs"] == 1 and st["engines"] == 1
    assert st["placed_bytes"] == 0 and st["pools"] == [] and st["pool_idle_s"] >= 0


def test_dropin_keeps_device_tables_per_wavelength(hip_on_cpu):
    """Alternating wavelengths (what SpotDiagram does per field) reuses the device tables
    instead of re-creating one per call; a change of the prescription makes a new one."""
    from optiland.samples.objectives import CookeTriplet
    from optiland_amd import integration
    lens = CookeTriplet()
    integration.install(lens, force=True)
    t = lens.ray_tracer
    for _ in range(3):
        for w in (0.48, 0.55, 0.65):
            lens.trace(0.0, 0.7, w, num_rays=3, distribution="hexapolar")
    assert len(t._hip_engines) == 3
    assert t.pack_count == 3   # the change detector spared the other six packs
    # no cached front pins its last record block (4 GB each at 1e7 rays): the block lives
    # only through the reference objects that view it
    assert all(f.surfaces._res is None for hit in t._hip_engines.values()
               for f in hit[2].values())
    engines = {id(hit[0]) for hit in t._hip_engines.values()}
    for w in (0.48, 0.55, 0.65):
        lens.trace(0.0, 0.0, w, num_rays=3, distribution="hexapolar")
    assert {id(hit[0]) for hit in t._hip_engines.values()} == engines
    assert t.pack_count == 3
    lens.surfaces[2].geometry.radius = float(lens.surfaces[2].geometry.radius) * 1.02
    lens.trace(0.0, 0.0, 0.55, num_rays=3, distribution="hexapolar")
    assert t.pack_count == 4
    if t.engine_updates:   # an engine with ol_system_update: that wavelength's table is
        assert len(t._hip_engines) == 3 and t.engine_updates == 1   # patched in place
    else:
        assert len(t._hip_engines) == 4


# ------------------------------------------------------------------------------------
# the SurfaceGroup.trace(rays, skip) seam (SURVEY.md 8b "covers direct callers")
# ------------------------------------------------------------------------------------
@pytest.fixture()
def sg_seam(hip_on_cpu):
    from optiland_amd import integration
    integration.enable(force=True)
    integration._SG.update(count=0, fallbacks=0)
    yield integration
    integration.disable()


def _clone_rays(be, rays):
    import copy
    out = copy.copy(rays)
    for k, v in vars(rays).items():
        if hasattr(v, "clone"):
            setattr(out, k, v.clone())
    return out


@pytest.mark.parametrize("skip", [0, 1, 3])
def test_surface_group_seam_matches_reference_loop(sg_seam, skip):
    """Caller-built RealRays through `optic.surfaces.trace(rays, skip)`: in-place ray
    state, L0/M0/N0, and every traced surface's record equal the reference's own loop;
    surfaces before `skip` stay reset (empty)."""
    import optiland.backend as be
    from optiland.samples.objectives import CookeTriplet
    lens = CookeTriplet()
    gen = lens.ray_tracer.ray_generator
    px, py = be.array(np.linspace(-0.8, 0.8, 41)), be.array(np.linspace(0.7, -0.5, 41))
    start = gen.generate_rays(be.zeros_like(px), be.ones_like(px) * 0.6, px, py, 0.55)
    if skip:  # walk the first surfaces with the reference so the start state is mid-system
        for s in lens.surfaces.surfaces[:skip]:
            s.trace(start)
    a, b = _clone_rays(be, start), _clone_rays(be, start)
    want = sg_seam._ORIGINALS["sg_trace"](lens.surfaces, a, skip)
    rec_want = {k: [_np(be, getattr(s, k)) for s in lens.surfaces.surfaces]
                for k in ("x", "y", "z", "L", "M", "N", "intensity", "opd")}
    got = lens.surfaces.trace(b, skip=skip)
    assert got is b and sg_seam._SG["count"] == 1 and sg_seam._SG["fallbacks"] == 0
    for k in ("x", "y", "z", "L", "M", "N", "i", "opd", "L0", "M0", "N0", "w"):
        np.testing.assert_allclose(_np(be, getattr(got, k)), _np(be, getattr(want, k)),
                                   rtol=1e-10, atol=1e-11, err_msg=k)
    for k, rows in rec_want.items():
        for s, (surf, w_) in enumerate(zip(lens.surfaces.surfaces, rows)):
            g_ = _np(be, getattr(surf, k))
            assert g_.shape == w_.shape, (k, s)
            np.testing.assert_allclose(g_, w_, rtol=1e-10, atol=1e-10, err_msg=f"{k}[{s}]")
    assert _np(be, lens.surfaces.x).shape == (len(lens.surfaces.surfaces) - skip, 41) or skip == 0


def test_l0_is_in_the_frame_of_a_tilted_last_surface(sg_seam):
    """the reference stores L0 / M0 / N0 inside refract() / reflect(), after
    localize(): for a TILTED last surface they are the pre-interaction cosines in that
    surface's own frame, not the previous surface's recorded (global) direction.  Both
    seams -- SurfaceGroup.trace with caller-built rays and Optic.trace_generic -- must
    reproduce that."""
    import optiland.backend as be
    from optiland import optic as optic_mod

    def build():
        lens = optic_mod.Optic(name="TiltedImage")
        lens.surfaces.add(index=0, radius=be.inf, thickness=be.inf)
        lens.surfaces.add(index=1, radius=40.0, thickness=6.0, material="N-BK7", is_stop=True)
        lens.surfaces.add(index=2, radius=-60.0, thickness=30.0)
        lens.surfaces.add(index=3, radius=be.inf, rx=0.6, ry=-0.2)   # tilted image plane
        lens.set_aperture(aperture_type="EPD", value=10)
        lens.fields.set_type(field_type="angle")
        lens.fields.add(y=0)
        lens.fields.add(y=2.0)
        lens.wavelengths.add(value=0.55, is_primary=True)
        return lens
    lens = build()
    px, py = be.array(np.linspace(-0.8, 0.8, 23)), be.array(np.linspace(0.6, -0.7, 23))
    gen = lens.ray_tracer.ray_generator
    start = gen.generate_rays(be.zeros_like(px), be.ones_like(px) * 0.7, px, py, 0.55)
    a, b = _clone_rays(be, start), _clone_rays(be, start)
    want = sg_seam._ORIGINALS["sg_trace"](lens.surfaces, a, 0)
    got = lens.surfaces.trace(b, skip=0)
    assert sg_seam._SG["count"] == 1
    for k in ("L0", "M0", "N0", "x", "L"):
        np.testing.assert_allclose(_np(be, getattr(got, k)), _np(be, getattr(want, k)),
                                   rtol=1e-10, atol=1e-12, err_msg=k)
    # the cosines really are rotated (the test would be vacuous on an untilted surface)
    assert np.abs(_np(be, got.M0) - _np(be, lens.surfaces.surfaces[2].M)).max() > 0.1
    g = lens.trace_generic(0.0, 0.7, px, py, 0.55)
    assert lens.ray_tracer._hip_companion.last_path == "hip"
    for k in ("L0", "M0", "N0"):
        np.testing.assert_allclose(_np(be, getattr(g, k)), _np(be, getattr(want, k)),
                                   rtol=1e-10, atol=1e-12, err_msg=k)


def test_enable_twice_updates_the_settings(hip_on_cpu):
    """a second enable() with another device / force is not ignored."""
    from optiland.samples.objectives import CookeTriplet
    from optiland_amd import integration
    integration.enable(force=False)
    try:
        lens = CookeTriplet()
        lens.trace(0.0, 0.0, 0.55, 3, "hexapolar")
        assert lens.ray_tracer._hip_companion.last_path == "reference"   # cpu, not forced
        integration.enable(force=True)
        lens.trace(0.0, 0.0, 0.55, 3, "hexapolar")
        assert lens.ray_tracer._hip_companion.last_path == "hip"
        assert integration._SG["force"] is True
    finally:
        integration.disable()


def test_surface_group_seam_advances_an_existing_prt(sg_seam):
    """PolarizedRays whose `p` is no longer the identity (traced half way by the
    reference) continue through surfaces[skip:] on the seam: p, i, and the record match
    the reference loop.

    The reference side runs on its NumPy backend (the parity target): on the torch
    backend `cross(k0, k0)` is not exactly zero (fused multiply-adds), so the
    `mag == 0` branch of PolarizedRays.get_local_basis (rays/polarized_rays.py:154-166)
    is never taken and every undeviated ray -- every ray at the image plane -- gets an
    s-vector made of rounding noise and a non-unitary PRT."""
    import torch
    import optiland.backend as be
    from optiland.rays import PolarizationState
    from optiland.samples.objectives import CookeTriplet

    def build():
        lens = CookeTriplet()
        lens.surfaces.set_fresnel_coatings()
        lens.updater.set_polarization(PolarizationState(is_polarized=False))
        return lens

    def start_rays(lens):
        px, py = be.array(np.linspace(-0.7, 0.7, 29)), be.array(np.linspace(0.6, -0.6, 29))
        return lens.ray_tracer.ray_generator.generate_rays(
            be.zeros_like(px), be.ones_like(px) * 0.5, px, py, 0.55)

    be.set_backend("numpy")
    lens_np = build()
    a = start_rays(lens_np)
    assert type(a).__name__ == "PolarizedRays"
    for s in lens_np.surfaces.surfaces[:4]:
        s.trace(a)
    assert float(abs(a.p.real[:, 0, 0] - 1).max()) > 1e-3   # not the identity any more
    snap = {k: np.array(v) for k, v in vars(a).items() if isinstance(v, np.ndarray)}
    want = lens_np.surfaces.trace(a, skip=4)                 # NumPy backend: seam declines
    assert sg_seam._SG["count"] == 0
    be.set_backend("torch")
    lens_t = build()
    b = start_rays(lens_t)
    for k, v in snap.items():
        setattr(b, k, torch.as_tensor(v))
    got = lens_t.surfaces.trace(b, skip=4)
    assert sg_seam._SG["count"] == 1
    for k in ("x", "y", "z", "L", "M", "N", "i", "opd"):
        np.testing.assert_allclose(_np(be, getattr(got, k)), np.asarray(getattr(want, k)),
                                   rtol=1e-10, atol=1e-11, err_msg=k)
    np.testing.assert_allclose(be.to_numpy(got.p), want.p, rtol=1e-9, atol=1e-10)
    mk = lambda: PolarizationState(is_polarized=True, Ex=1.0, Ey=0.0,  # noqa: E731
                                   phase_x=0.0, phase_y=0.0)
    got.update_intensity(mk())
    be.set_backend("numpy")
    want.update_intensity(mk())
    be.set_backend("torch")
    np.testing.assert_allclose(_np(be, got.i), np.asarray(want.i), rtol=1e-9, atol=1e-12)


def test_surface_group_seam_declines_what_it_must(sg_seam):
    """Paraxial rays, per-ray wavelengths, polarised coatings with plain RealRays and the
    NumPy backend all run the reference's loop untouched."""
    import optiland.backend as be
    from optiland.samples.objectives import CookeTriplet
    lens = CookeTriplet()
    gen = lens.ray_tracer.ray_generator
    px = be.array(np.linspace(-0.5, 0.5, 7))
    rays = gen.generate_rays(be.zeros_like(px), be.zeros_like(px), px, px, 0.55)
    rays.w = rays.w * be.array(np.linspace(0.9, 1.1, 7))      # per-ray wavelengths
    lens.surfaces.trace(rays)
    assert (sg_seam._SG["count"], sg_seam._SG["fallbacks"]) == (0, 1)
    lens.paraxial._ray_tracer.trace(0.0, 1.0, 0.55)              # ParaxialRays walk the same method
    assert sg_seam._SG["count"] == 0 and sg_seam._SG["fallbacks"] >= 2
    before = sg_seam._SG["fallbacks"]
    be.set_backend("numpy")
    lens2 = CookeTriplet()
    r2 = lens2.ray_tracer.ray_generator.generate_rays(0.0, 0.0, np.zeros(3), np.zeros(3), 0.55)
    lens2.surfaces.trace(r2)
    assert sg_seam._SG["count"] == 0 and sg_seam._SG["fallbacks"] == before + 1
    be.set_backend("torch")


def test_irradiance_with_user_rays_runs_on_the_seam(sg_seam):
    """IncoherentIrradiance(user_initial_rays=...) calls optic.surfaces.trace(rays)
    directly (analysis/irradiance.py:279): with the seam enabled that trace is the HIP
    launch, and the irradiance map equals the reference's."""
    import optiland.backend as be
    from optiland import analysis
    from optiland.physical_apertures import RectangularAperture
    from optiland.samples.objectives import CookeTriplet

    def build():
        lens = CookeTriplet()
        lens.surfaces.surfaces[-1].aperture = RectangularAperture(-2.0, 2.0, -2.0, 2.0)
        return lens
    lens = build()
    px, py = np.meshgrid(np.linspace(-0.9, 0.9, 31), np.linspace(-0.9, 0.9, 31))
    px, py = be.array(px.ravel()), be.array(py.ravel())
    rays = lens.ray_tracer.ray_generator.generate_rays(be.zeros_like(px), be.zeros_like(px),
                                                       px, py, 0.55)
    irr = analysis.IncoherentIrradiance(lens, res=(16, 16), user_initial_rays=_clone_rays(be, rays))
    assert sg_seam._SG["count"] >= 1
    got = _np(be, irr.data[0][0][0])
    sg_seam.disable()
    ref_irr = analysis.IncoherentIrradiance(build(), res=(16, 16),
                                            user_initial_rays=_clone_rays(be, rays))
    want = _np(be, ref_irr.data[0][0][0])
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-12)
    assert want.sum() > 0


def _grating_lens(be):
    """tests/test_grating.py:13-44 with a curved lens after the grating."""
    from optiland.optic import Optic
    lens = Optic()
    lens.surfaces.add(index=0, radius=be.inf, thickness=be.inf)
    lens.surfaces.add(index=1, radius=80.0, thickness=4, material="N-BK7")
    lens.surfaces.add(index=2, radius=-90.0, thickness=5)
    lens.surfaces.add(index=3, radius=be.inf, thickness=3, surface_type="grating", grating_order=-1,
                      grating_period=5.0, groove_orientation_angle=0.0, is_stop=True,
                      material="N-BK7")
    lens.surfaces.add(index=4, radius=-60.0, thickness=6)
    lens.surfaces.add(index=5, radius=50.0, thickness=3, material="SF6")
    lens.surfaces.add(index=6, radius=be.inf, thickness=30)
    lens.surfaces.add(index=7)
    lens.set_aperture(aperture_type="EPD", value=12)
    lens.fields.set_type(field_type="angle")
    lens.fields.add(y=0)
    lens.fields.add(y=5)
    lens.wavelengths.add(value=0.587, is_primary=True)
    lens.updater.update_paraxial()
    return lens


def _thin_lens_system(be):
    from optiland.optic import Optic
    lens = Optic()
    lens.surfaces.add(index=0, radius=be.inf, thickness=be.inf)
    lens.surfaces.add(index=1, radius=70.0, thickness=4, material="N-BK7", is_stop=True)
    lens.surfaces.add(index=2, radius=-120.0, thickness=6)
    lens.surfaces.add(index=3, surface_type="paraxial", f=80, thickness=5)
    lens.surfaces.add(index=4, radius=40.0, thickness=3, material="N-SF11")
    lens.surfaces.add(index=5, radius=-300.0, thickness=25)
    lens.surfaces.add(index=6)
    lens.set_aperture(aperture_type="EPD", value=10)
    lens.fields.set_type(field_type="angle")
    lens.fields.add(y=0)
    lens.fields.add(y=4)
    lens.wavelengths.add(value=0.55, is_primary=True)
    lens.updater.update_paraxial()
    return lens


@pytest.mark.parametrize("build", [_grating_lens, _thin_lens_system], ids=["grating", "thin_lens"])
def test_unsupported_surface_is_bridged_by_the_reference(sg_seam, build):
    """A grating / thin lens in the middle of a system is left to the reference's own
    Surface.trace; the runs of supported surfaces on either side are fused launches.
    Whole-trace results (Optic.trace -> reference ray generation -> patched
    SurfaceGroup.trace) equal the unpatched reference."""
    import optiland.backend as be
    lens = build(be)
    got = lens.trace(0.0, 1.0, lens.primary_wavelength, 8, "hexapolar")
    c = dict(sg_seam._SG)
    assert c["count"] == 1 and c["foreign"] >= 1 and c["fallbacks"] == 0
    rec_got = {k: _np(be, getattr(lens.surfaces, k)) for k in ("x", "y", "z", "L", "M", "N",
                                                                "intensity", "opd")}
    sg_seam.disable()
    lens2 = build(be)
    want = lens2.trace(0.0, 1.0, lens2.primary_wavelength, 8, "hexapolar")
    for k in ("x", "y", "z", "L", "M", "N", "i", "opd"):
        np.testing.assert_allclose(_np(be, getattr(got, k)), _np(be, getattr(want, k)),
                                   rtol=1e-9, atol=1e-10, err_msg=k)
    assert np.isfinite(_np(be, want.x)).mean() > 0.9
    for k, a in rec_got.items():
        b = _np(be, getattr(lens2.surfaces, k))
        assert a.shape == b.shape
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-9, err_msg=k)


def test_install_alone_also_bridges_on_its_own_surface_group(hip_on_cpu):
    """install(optic) patches that optic's SurfaceGroup instance only: its grating system
    runs fused around the grating, another optic's SurfaceGroup is untouched, and
    uninstall() removes the instance patch."""
    be = hip_on_cpu
    from optiland_amd import integration
    integration._SG.update(count=0, fallbacks=0, foreign=0)
    lens, other = _grating_lens(be), _grating_lens(be)
    tracer = integration.install(lens, force=True)
    got = lens.trace(0.0, 1.0, lens.primary_wavelength, 8, "hexapolar")
    assert tracer.last_path == "reference"            # ray generation + driver: reference
    assert integration._SG["count"] == 1 and integration._SG["foreign"] >= 1
    want = other.trace(0.0, 1.0, other.primary_wavelength, 8, "hexapolar")
    assert integration._SG["count"] == 1              # the other optic never entered the seam
    for k in ("x", "y", "z", "L", "M", "N", "i", "opd"):
        np.testing.assert_allclose(_np(be, getattr(got, k)), _np(be, getattr(want, k)),
                                   rtol=1e-9, atol=1e-10, err_msg=k)
    import copy
    clone = copy.deepcopy(lens)                        # engines cached on the group survive
    clone.trace(0.0, 0.5, clone.primary_wavelength, 6, "hexapolar")
    assert integration._SG["count"] == 2
    integration.uninstall(lens)
    lens.trace(0.0, 1.0, lens.primary_wavelength, 8, "hexapolar")
    assert integration._SG["count"] == 2 and "trace" not in lens.surfaces.__dict__


@pytest.mark.parametrize("seed", range(40))
def test_drop_in_on_random_lenses(hip_on_cpu, seed):
    """Random reference-built lenses (tests/test_reference_fuzz.py) through the drop-in
    tracer -- install(force=True) on the oracle-backed engine, torch backend -- against
    the reference's own NumPy-backend trace: `trace_generic` with per-ray pupil arrays and
    `trace` with a distribution and two field points, incl. vignetting factors, field
    types, apodization, polarised update_intensity and the per-surface records."""
    be = hip_on_cpu
    import importlib.util
    from optiland_amd.integration import install
    spec = importlib.util.spec_from_file_location(
        "_ref_fuzz", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_reference_fuzz.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)

    be.set_backend("numpy")
    lens_np, rng = fz.build_random_lens(seed, be)
    w = float(lens_np.primary_wavelength)
    n = 200
    r, th = np.sqrt(rng.random(n)) * 0.9, 2 * np.pi * rng.random(n)
    px, py = r * np.cos(th), r * np.sin(th)
    hx, hy = float(rng.uniform(-0.6, 0.6)), float(rng.uniform(-1, 1))
    fields = (np.array([0.0, hx]), np.array([0.0, hy]))
    try:
        with np.errstate(all="ignore"):
            g0 = lens_np.trace_generic(hx, hy, px, py, w)
            rec0 = {k: np.asarray(getattr(lens_np.surfaces, k), dtype=np.float64)
                    for k in ("x", "y", "z", "L", "M", "N", "intensity", "opd")}
            t0 = lens_np.trace(fields[0], fields[1], w, 4, "hexapolar")
    except ValueError:
        pytest.skip("reference raises a coordinate-range error for this lens")
    want_g = {k: np.asarray(getattr(g0, k), dtype=np.float64) for k in "xyzLMN"}
    want_g.update(i=np.asarray(g0.i, dtype=np.float64), opd=np.asarray(g0.opd, dtype=np.float64))
    want_t = {k: np.asarray(getattr(t0, k), dtype=np.float64) for k in ("x", "y", "L", "M", "i", "opd")}

    be.set_backend("torch")
    lens, _ = fz.build_random_lens(seed, be)
    tracer = install(lens, force=True)
    with np.errstate(all="ignore"):
        g1 = lens.trace_generic(hx, hy, be.array(px), be.array(py), w)
        assert tracer.last_path == "hip"
        rec1 = {k: _np(be, getattr(lens.surfaces, k)) for k in rec0}
        t1 = lens.trace(be.array(fields[0]), be.array(fields[1]), w, 4, "hexapolar")
        assert tracer.last_path == "hip"
    scale = max(1.0, float(np.nanmax(np.abs(rec0["z"][1:][np.isfinite(rec0["z"][1:])]))))

    def close(a, b, k):
        assert a.shape == b.shape, k
        assert np.array_equal(np.isnan(a), np.isnan(b)), f"{k}: NaN masks differ"
        # (k = "<what> <plane>": lengths are held to 1e-7 of the system size -- the reference
        # stops its Newton loop at 1e-6 mm, the kernel converges each ray further)
        tol = 1e-7 * (scale if k.split()[-1] in ("x", "y", "z", "opd") else 1.0)
        np.testing.assert_allclose(np.nan_to_num(a, posinf=0, neginf=0),
                                   np.nan_to_num(b, posinf=0, neginf=0), rtol=0, atol=tol,
                                   err_msg=f"seed {seed} {k}")
    for k, b in want_g.items():
        close(_np(be, getattr(g1, k)), b, "generic " + k)
    for k, b in rec0.items():
        a = rec1[k]
        if k in "xyz" and not np.isfinite(b[0]).all():
            a, b = a[1:], b[1:]
        close(a, b, "record " + k)
    for k, b in want_t.items():
        close(_np(be, getattr(t1, k)), b, "trace " + k)


@pytest.mark.parametrize("seed", range(24))
def test_bridging_on_random_lenses(sg_seam, seed):
    """Random reference-built lenses with a thin lens (paraxial surface) or a grating
    inserted at a random position: `Optic.trace` under enable() -- reference ray generation,
    fused runs around the unsupported surface, that surface (and, after a thin lens, the
    renormalising next one) on the reference -- equals the reference's NumPy-backend trace,
    final rays and every surface's record."""
    import importlib.util
    import optiland.backend as be
    spec = importlib.util.spec_from_file_location(
        "_ref_fuzz", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_reference_fuzz.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)

    def build():
        lens, rng = fz.build_random_lens(seed, be)
        n_s = len(lens.surfaces.surfaces)
        k = int(rng.integers(2, n_s - 1)) if n_s > 3 else n_s - 1
        if seed % 2:
            lens.surfaces.add(index=k, surface_type="paraxial", f=float(rng.uniform(60, 200)),
                              thickness=float(rng.uniform(1, 3)))
        else:
            lens.surfaces.add(index=k, radius=be.inf, thickness=float(rng.uniform(1, 3)),
                              surface_type="grating", grating_order=int(rng.choice([-1, 0, 1])),
                              grating_period=float(rng.uniform(20.0, 60.0)),
                              groove_orientation_angle=float(rng.uniform(0, 1.5)))
        return lens, rng

    be.set_backend("numpy")
    try:
        lens_np, rng = build()
    except Exception as e:  # the reference cannot insert there (e.g. inside a mirror fold)
        be.set_backend("torch")
        pytest.skip(f"reference cannot build this variant: {type(e).__name__}")
    if lens_np.polarization != "ignore":
        be.set_backend("torch")
        pytest.skip("polarised reference traces differ between its own backends (DESIGN 7)")
    w = float(lens_np.primary_wavelength)
    hx, hy = float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-1, 1))
    try:
        with np.errstate(all="ignore"):
            r0 = lens_np.trace(hx, hy, w, 5, "hexapolar")
            want = {k: np.asarray(getattr(r0, k), dtype=np.float64) for k in ("x", "y", "z", "L", "M", "N", "i", "opd")}
            rec0 = {k: np.asarray(getattr(lens_np.surfaces, k), dtype=np.float64)
                    for k in ("x", "y", "z", "L", "M", "N", "intensity", "opd")}
    except ValueError:
        be.set_backend("torch")
        pytest.skip("reference raises a coordinate-range error for this lens")
    be.set_backend("torch")
    lens, _ = build()
    sg_seam._SG.update(count=0, fallbacks=0, foreign=0)
    with np.errstate(all="ignore"):
        r1 = lens.trace(hx, hy, w, 5, "hexapolar")
    assert sg_seam._SG["count"] == 1 and sg_seam._SG["foreign"] >= 1, dict(sg_seam._SG)
    z = rec0["z"][1:]
    scale = max(1.0, float(np.abs(z[np.isfinite(z)]).max()) if np.isfinite(z).any() else 1.0)

    def close(a, b, k):
        assert a.shape == b.shape, k
        assert np.array_equal(np.isnan(a), np.isnan(b)), f"{k}: NaN masks differ"
        tol = 1e-7 * (scale if k[-1] in "xyzd" else 1.0)
        np.testing.assert_allclose(np.nan_to_num(a, posinf=0, neginf=0),
                                   np.nan_to_num(b, posinf=0, neginf=0), rtol=0, atol=tol,
                                   err_msg=f"seed {seed} {k}")
    for k, b in want.items():
        close(_np(be, getattr(r1, k)), b, "final " + k)
    for k, b in rec0.items():
        a = _np(be, getattr(lens.surfaces, k))
        if k in "xyz" and not np.isfinite(b[0]).all():
            a, b = a[1:], b[1:]
        close(a, b, "record " + k)


def test_size_mismatch_text_is_the_numpy_backends(ref):
    """Round 5 ("array-length mismatch -- same type, different text").  The text
    `HipRayTracer.trace_generic` raises for coordinate arrays of different lengths is the one
    the LIVE reference's NumPy backend ends in, for every combination of array / scalar
    arguments (which operation of the reference fails first decides it)."""
    import itertools

    from optiland.samples.objectives import CookeTriplet
    from optiland_amd.tracer import _size_mismatch_message
    be = ref
    be.set_backend("numpy")
    lens = CookeTriplet()
    for lens_ in ((3, 4, 5, 6), (6, 5, 4, 3), (4, 4, 5, 6), (4, 4, 4, 6), (4, 4, 6, 4), (2, 2, 7, 5),
                  (9, 9, 2, 9), (3, 3, 3, 3)):
        for mask in itertools.product([0, 1], repeat=4):
            args = [np.full(n, 0.1) if m else 0.1 for n, m in zip(lens_, mask)]
            try:
                lens.trace_generic(*args, 0.55)
                want = None
            except ValueError as e:
                want = str(e)
            got = _size_mismatch_message(*[n if m else None for n, m in zip(lens_, mask)])
            assert got == want, (lens_, mask, got, want)
    # a one-element array broadcasts like a scalar
    assert lens.trace_generic(np.array([0.0]), 0.1, np.full(5, 0.1), np.array([0.2]), 0.55) \
        .x.shape == (5,)


def test_a_ray_without_a_direction_comes_back_without_a_position(hip_on_cpu, request):
    """Round 5 (tools/seam_fuzz.py, conic-only polarised lenses, seed 564).  `Optic.trace` ends
    with `x += t L` by the last surface's thickness (real_ray_tracer.py:104-110) even when that
    is 0, and 0 * NaN is NaN: a ray that is totally reflected at the LAST surface -- a position,
    no direction -- is returned without a position, while `surfaces.x[-1]` keeps it.  The
    kernel reports such rays (OL_STATUS_NAN_DIRECTION) and the drop-in then gives the returned
    rays their own x, y, z."""
    if "oracle" in request.node.name:
        pytest.skip("the stand-in engine has no status word")
    import importlib.util
    be = hip_on_cpu
    from optiland_amd import integration
    spec = importlib.util.spec_from_file_location(
        "_ref_fuzz2", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_reference_fuzz.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    fz.KINDS = ["standard"]

    def run():
        lens, _ = fz.build_random_lens(564, be)
        r = lens.trace(0.3, -0.5, lens.primary_wavelength, 5, "hexapolar")
        return ({k: _np(be, getattr(r, k)) for k in ("x", "y", "z", "L", "M", "N", "i", "opd")},
                {k: _np(be, getattr(lens.surfaces, k))[-1] for k in ("x", "y", "z", "L")})

    be.set_backend("numpy")
    want, want_row = run()
    be.set_backend("torch")
    be.set_device("cpu")
    be.set_precision("float64")
    lost = np.isnan(want["x"]) & np.isfinite(want_row["x"])
    assert lost.sum() >= 1 and np.isnan(want_row["L"][lost]).all()      # the reference's quirk
    integration.enable(force=True)
    try:
        got, got_row = run()
    finally:
        integration.disable()
    for k in want:
        assert np.array_equal(np.isnan(got[k]), np.isnan(want[k])), k
        np.testing.assert_allclose(np.nan_to_num(got[k]), np.nan_to_num(want[k]), rtol=0, atol=1e-9)
    np.testing.assert_allclose(got_row["x"], want_row["x"], rtol=0, atol=1e-9, equal_nan=True)
    assert np.isfinite(got_row["x"][lost]).all()                        # the record keeps it


def test_polarised_bundle_with_unnormalised_directions_goes_through_the_kernel(hip_on_cpu):
    """Found in round 5 (tools/seam_fuzz.py, family `aimed`), served by the kernels since round
    6.  The reference's iterative / robust ray aimers hand out direction cosines with
    |k|^2 - 1 ~ 1e-3, and nothing renormalises them.  Its PRT algebra takes k as it comes
    (polarized_rays.py:136-202: the triads stop being orthonormal, every surface scales the
    matrix by |k0| |k1|); the kernel's rank-2 update equals that only for |k| = 1 -- 0.3 % of
    the returned intensity on this lens -- unless it is told (`OL_TRACE_NONUNIT_K`): then it
    works on the normalised directions with the p and k amplitudes scaled by |k0| |k1|, which
    is the same matrix.  `SurfaceGroup.trace` sets the flag for such a polarised bundle and
    leaves only the equal-index image surface (s = rounding noise of k0 x k1 in the reference)
    to the reference's own `Surface.trace`."""
    import importlib.util
    be = hip_on_cpu
    from optiland_amd import integration
    spec = importlib.util.spec_from_file_location(
        "_ref_fuzz3", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_reference_fuzz.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    fz.KINDS = ["standard"]

    def run(mode):
        lens, _ = fz.build_random_lens(5205, be)
        assert lens.polarization != "ignore"
        lens.ray_tracer.set_aiming(mode, 10, 1e-9)
        r = lens.trace(0.0, 0.7, lens.primary_wavelength, 3, "hexapolar")
        k2 = _np(be, lens.surfaces.L)[0] ** 2 + _np(be, lens.surfaces.M)[0] ** 2 \
            + _np(be, lens.surfaces.N)[0] ** 2
        out = {k: _np(be, getattr(r, k)) for k in ("x", "y", "z", "L", "M", "N", "i", "opd")}
        out["p"] = np.asarray(be.to_numpy(r.p)).real.reshape(-1)
        return out, k2

    for mode in ("iterative", "robust"):
        be.set_backend("numpy")
        want, k2 = run(mode)
        assert np.abs(k2 - 1.0).max() > 1e-5                 # the aimer's quirk
        be.set_backend("torch")
        be.set_device("cpu")
        be.set_precision("float64")
        integration.enable(force=True)
        served, declined = integration._SG["count"], integration._SG["fallbacks"]
        try:
            got, _ = run(mode)
        finally:
            integration.disable()
        # the kernels traced the bundle (the aimers' own traces to the stop are unpolarised
        # seam calls too: the counter moves by more than one), nothing fell back
        assert integration._SG["count"] > served and integration._SG["fallbacks"] == declined
        for k in want:
            np.testing.assert_allclose(got[k], want[k], rtol=0, atol=1e-10, equal_nan=True,
                                       err_msg=f"{mode} {k}")
