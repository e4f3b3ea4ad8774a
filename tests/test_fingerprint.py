"""The change detector that spares the drop-in a re-pack per trace
(optiland_amd/fingerprint.py): it must (a) be STABLE -- an untouched optic is packed
once, however often it is traced -- and (b) see EVERY mutation the reference's public
API (and plain attribute writes, in-place tensor writes included) can make to anything
the packer reads.  (b) is checked differentially: after each mutation the memoised
drop-in must return exactly what a drop-in with the memo switched off returns.

CPU, live reference (build container or staged oracle/_ref), oracle-backed engine.
"""

import importlib
import inspect

import numpy as np
import pytest

from tests import _live

pytestmark = pytest.mark.skipif(_live.reference_root() is None,
                                reason="reference package not present")


@pytest.fixture(params=["native", "python"])
def be(monkeypatch, request):
    """Every test runs on both walks: csrc/fptoken.c (built by build.build_fptoken) and the
    pure-Python definitions in fingerprint.py."""
    from optiland_amd import build, fingerprint
    if request.param == "native":
        build.build_fptoken()
        if fingerprint._NATIVE is None:
            fingerprint._NATIVE = fingerprint._load_native()
        if not fingerprint.use_native(True):
            pytest.skip("native token extension not available")
    else:
        fingerprint.use_native(False)
    import optiland_amd.tracer as tr
    from tests._fake_engine import OracleEngine
    monkeypatch.setattr(tr, "_make_engine", lambda table, device: OracleEngine(table, device))
    be = _live.import_reference()
    be.set_backend("torch")
    be.set_device("cpu")
    be.set_precision("float64")
    yield be
    be.set_backend("numpy")
    fingerprint.use_native(True)


def test_native_and_python_walks_build_identical_tokens():
    """The C walk (csrc/fptoken.c) and the Python one must produce the SAME token tree --
    element for element -- on every lens of optiland.samples (plain and with coatings,
    polarisation, apertures), so that either can validate a memo the other wrote."""
    from optiland_amd import build, fingerprint
    build.build_fptoken()
    if fingerprint._NATIVE is None:
        fingerprint._NATIVE = fingerprint._load_native()
    if fingerprint._NATIVE is None:
        pytest.skip("native token extension not available")
    be_ = _live.import_reference()
    from optiland import physical_apertures as pa
    from optiland.rays import PolarizationState
    checked = 0
    try:
        for backend in ("numpy", "torch"):
            be_.set_backend(backend)
            for cname, cls in _sample_classes():
                lens = cls()
                if checked % 3 == 0:
                    lens.surfaces.set_fresnel_coatings()
                    lens.updater.set_polarization(PolarizationState(is_polarized=False))
                if checked % 4 == 0 and len(lens.surfaces.surfaces) > 3:
                    lens.surfaces[2].aperture = pa.UnionAperture(
                        pa.RadialAperture(r_max=5.0), pa.RectangularAperture(-1, 1, -2, 2))
                w = 0.55
                fingerprint.use_native(True)
                a, _ = fingerprint.optic_token(lens, w)
                sa, _ = fingerprint.surfaces_token(lens.surfaces.surfaces, w)
                fingerprint.use_native(False)
                b, _ = fingerprint.optic_token(lens, w)
                sb, _ = fingerprint.surfaces_token(lens.surfaces.surfaces, w)
                assert a == b and sa == sb, (backend, cname)
                checked += 1
    finally:
        fingerprint.use_native(True)
        be_.set_backend("numpy")
    assert checked >= 50


def _snapshot(be, lens, w):
    r = lens.trace(0.0, 0.6, w, 4, "hexapolar")
    out = [np.asarray(be.to_numpy(getattr(r, k)), dtype=np.float64)
           for k in ("x", "y", "z", "L", "M", "N", "i", "opd")]
    g = lens.trace_generic(0.0, 1.0, 0.3, -0.2, w)
    out += [np.asarray(be.to_numpy(getattr(g, k)), dtype=np.float64) for k in ("x", "y", "i")]
    out.append(np.asarray(be.to_numpy(lens.surfaces.y), dtype=np.float64))
    if hasattr(r, "p"):
        out.append(np.asarray(be.to_numpy(r.p)).real)
    return out


def _sample_classes():
    from optiland import optic as optic_mod
    for m in ("eyepieces", "infrared", "lithography", "microscopes", "miscellaneous",
              "objectives", "simple", "telescopes"):
        mod = importlib.import_module("optiland.samples." + m)
        for cname, cls in inspect.getmembers(mod, inspect.isclass):
            if cls.__module__ == mod.__name__ and issubclass(cls, optic_mod.Optic):
                yield cname, cls


def test_untouched_optics_are_packed_once(be):
    """Every lens of optiland.samples: three rounds of trace / trace_generic at two
    wavelengths -> exactly one pack per wavelength (tokens are stable across traces)."""
    from optiland_amd import integration
    checked = 0
    for cname, cls in _sample_classes():
        lens = cls()
        t = integration.install(lens, force=True)
        ws = [float(be.to_numpy(w.value).reshape(-1)[0]) if hasattr(w.value, "shape")
              else float(w.value) for w in lens.wavelengths.wavelengths][:2]
        for _ in range(3):
            for w in ws:
                lens.trace(0.0, 0.5, w, 2, "hexapolar")
                lens.trace_generic(0.0, 0.0, 0.1, 0.1, w)
        if t.last_path != "hip":
            continue  # a sample the fused path leaves to the reference
        assert t.pack_count == len(ws), (cname, t.pack_count)
        # per wavelength: the first call packs, the one after a pack validates before it
        # launches (an optic that has just been packed may be one that is being edited);
        # every later call was launched speculatively and kept
        assert t.speculative_hits == 6 * len(ws) - 2 * len(ws) and t.speculative_misses == 0, \
            (cname, t.speculative_hits, t.speculative_misses)
        checked += 1
    assert checked >= 20


def _mutations(be):
    from optiland import physical_apertures
    from optiland.coatings import SimpleCoating
    from optiland.materials import IdealMaterial
    from optiland.rays import PolarizationState
    state = PolarizationState(is_polarized=True, Ex=1.0, Ey=0.3, phase_x=0.0, phase_y=0.4)

    def inplace(lens):
        r = lens.surfaces[2].geometry.radius
        if hasattr(r, "mul_"):
            r.mul_(1.01)  # in-place tensor write: same object, `_version` bumps
        else:
            lens.surfaces[2].geometry.radius = r * 1.01

    def ap_inplace(lens):
        lens.surfaces[2].aperture.r_max = 3.1

    def coeff_inplace(lens):
        c = lens.surfaces[1].geometry.coefficients
        c[0] = c[0] * 0 + 2e-5

    return [
        ("set_radius", lambda L: L.set_radius(25.0, 1)),
        ("set_conic", lambda L: L.set_conic(-0.3, 1)),
        ("set_thickness", lambda L: L.set_thickness(4.1, 2)),
        ("geometry.radius attribute", lambda L: setattr(L.surfaces[3].geometry, "radius",
                                                        be.array(-21.0))),
        ("in-place tensor write", inplace),
        ("set_index", lambda L: L.set_index(1.61, 1)),
        ("set_material", lambda L: L.set_material(IdealMaterial(n=1.55, k=1e-7), 3)),
        ("new aperture object", lambda L: setattr(L.surfaces[2], "aperture",
                                                  physical_apertures.RadialAperture(r_max=3.5))),
        ("aperture attribute in place", ap_inplace),
        ("decentre", lambda L: setattr(L.surfaces[3].geometry.cs, "x", be.array(0.05))),
        ("tilt", lambda L: setattr(L.surfaces[4].geometry.cs, "rx", be.array(0.01))),
        ("field value", lambda L: setattr(L.fields.fields[1], "y", 12.0)),
        ("vignetting factor", lambda L: setattr(L.fields.fields[2], "vy", 0.2)),
        ("system aperture", lambda L: L.set_aperture("EPD", 8.0)),
        ("coating", lambda L: setattr(L.surfaces[1].interaction_model, "coating",
                                      SimpleCoating(transmittance=0.9, reflectance=0.05))),
        ("fresnel coatings + polarization", lambda L: (L.surfaces.set_fresnel_coatings(),
                                                       L.updater.set_polarization(state))),
        ("polarization state in place", lambda L: setattr(L.polarization, "Ey", 0.8)),
        ("apodization", lambda L: L.set_apodization("GaussianApodization", sigma=0.7)),
        ("image thickness", lambda L: setattr(L.surfaces[-1], "thickness", 0.25)),
    ]


def test_every_mutation_is_seen(be):
    from optiland.samples.objectives import CookeTriplet
    from optiland_amd import fingerprint, integration
    memo_lens, plain_lens = CookeTriplet(), CookeTriplet()
    t_memo = integration.install(memo_lens, force=True)
    t_plain = integration.install(plain_lens, force=True)
    w = 0.55

    def plain_snapshot():
        fingerprint.ENABLED = False
        try:
            return _snapshot(be, plain_lens, w)
        finally:
            fingerprint.ENABLED = True

    a, b = _snapshot(be, memo_lens, w), plain_snapshot()
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    for name, mutate in _mutations(be):
        before, missed = t_memo.pack_count, t_memo.speculative_misses
        mutate(memo_lens)
        mutate(plain_lens)
        a, b = _snapshot(be, memo_lens, w), plain_snapshot()
        assert t_memo.last_path == t_plain.last_path, name
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y, err_msg=f"stale table after: {name}")
        assert t_memo.pack_count > before, f"mutation not detected: {name}"
        # the launch queued on the stale table before the check was dropped, not returned
        assert t_memo.speculative_misses == missed + 1, name
        again = t_memo.pack_count
        _snapshot(be, memo_lens, w)
        assert t_memo.pack_count == again, f"token unstable after: {name}"


def test_asphere_coefficient_written_in_place(be):
    from optiland.samples.simple import AsphericSinglet
    from optiland_amd import fingerprint, integration
    memo_lens, plain_lens = AsphericSinglet(), AsphericSinglet()
    t = integration.install(memo_lens, force=True)
    integration.install(plain_lens, force=True)
    w = 0.587
    _snapshot(be, memo_lens, w)
    for lens in (memo_lens, plain_lens):
        c = lens.surfaces[1].geometry.coefficients
        c[0] = c[0] * 1.5  # element write into the list / tensor the geometry holds
    n0 = t.pack_count
    a = _snapshot(be, memo_lens, w)
    fingerprint.ENABLED = False
    try:
        b = _snapshot(be, plain_lens, w)
    finally:
        fingerprint.ENABLED = True
    assert t.pack_count > n0
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


def test_unsupported_verdict_is_memoised_and_revisited(be):
    """A system the fused path refuses is not re-packed on every call either -- and is
    looked at again as soon as it changes."""
    from optiland.samples.objectives import CookeTriplet
    from optiland_amd import integration
    lens = CookeTriplet()
    geom = lens.surfaces[3].geometry
    orig_cls = geom.__class__
    geom.__class__ = type("ForbesQbfsGeometry", (orig_cls,), {})
    t = integration.install(lens, force=True)
    for _ in range(3):
        lens.trace(0.0, 0.0, 0.55, 3, "hexapolar")
        assert t.last_path == "reference"
    assert t.pack_count == 1
    geom.__class__ = orig_cls
    lens.trace(0.0, 0.0, 0.55, 3, "hexapolar")
    assert t.last_path == "hip" and t.pack_count == 2


# ------------------------------------------------------------------------------------------
# round 3: the attribute set comes from the packer itself
# ------------------------------------------------------------------------------------------
def _traced_reads(optic, fn):
    """Every (object, attribute) of the reference object graph under `optic` that `fn()`
    reads out of an instance `__dict__`.  The classes of all objects reachable from the optic
    get a recording `__getattribute__` for the duration of the call."""
    classes, seen = set(), set()

    def walk(o, depth):
        if id(o) in seen or depth > 8:
            return
        seen.add(id(o))
        if isinstance(o, (list, tuple)):
            for e in o:
                walk(e, depth + 1)
            return
        d = getattr(o, "__dict__", None)
        if d is None or not type(o).__module__.startswith("optiland"):
            return
        classes.add(type(o))
        for v in list(d.values()):
            walk(v, depth + 1)

    walk(optic, 0)
    reads, saved = [], {}
    for cls in classes:
        saved[cls] = cls.__dict__.get("__getattribute__")
        base = cls.__getattribute__

        def ga(self, name, _base=base):
            v = _base(self, name)
            d = object.__getattribute__(self, "__dict__")
            if name in d:
                reads.append((self, name))
            return v

        cls.__getattribute__ = ga
    try:
        fn()
    finally:
        for cls, old in saved.items():
            if old is None:
                del cls.__getattribute__
            else:
                cls.__getattribute__ = old
    uniq, out = set(), []
    for o, n in reads:
        if (id(o), n) not in uniq:
            uniq.add((id(o), n))
            out.append((o, n))
    return out


def _value_changes(obj, name):
    """(apply, undo) pairs that change the VALUE of obj.name the way user code can: rebinding
    for python numbers / bools, in-place writes for tensors, arrays and lists."""
    import torch
    v = obj.__dict__[name]
    out = []
    if isinstance(v, bool):
        out.append((lambda: setattr(obj, name, not v), lambda: setattr(obj, name, v)))
    elif isinstance(v, (int, float)) and not isinstance(v, bool):
        if isinstance(v, float) and not np.isfinite(v):
            out.append((lambda: setattr(obj, name, 123.0), lambda: setattr(obj, name, v)))
        else:
            out.append((lambda: setattr(obj, name, v + 1), lambda: setattr(obj, name, v)))
    elif isinstance(v, torch.Tensor) and v.numel() > 0 and v.dtype.is_floating_point:
        old = v.detach().reshape(-1)[0].clone()
        new = old + 0.125 if bool(torch.isfinite(old)) else torch.full_like(old, 123.0)
        def app(v=v, new=new):
            with torch.no_grad():
                v.view(-1)[0] = new
        def und(v=v, old=old):
            with torch.no_grad():
                v.view(-1)[0] = old
        out.append((app, und))
    elif isinstance(v, np.ndarray) and v.size > 0 and v.dtype.kind == "f":
        j = v.size // 2
        old = v.flat[j].copy()
        new = old + 0.125 if np.isfinite(old) else 123.0
        def app(v=v, j=j, new=new):
            v.flat[j] = new
        def und(v=v, j=j, old=old):
            v.flat[j] = old
        out.append((app, und))
    elif isinstance(v, list) and v and all(isinstance(e, (int, float)) for e in v):
        old = v[0]
        out.append((lambda: v.__setitem__(0, old + 0.125), lambda: v.__setitem__(0, old)))
    return out


def _systems(be_):
    from optiland import physical_apertures as pa
    from optiland.samples.objectives import DoubleGauss
    yield "double_gauss", DoubleGauss()
    yield "rc_asphere", _live.rc_asphere()
    yield "zernike_fresnel", _live.zernike_fresnel("elliptical")
    lens = _live.rc_asphere()
    th = np.linspace(0.0, 2 * np.pi, 20000, endpoint=False)
    lens.surfaces[2].aperture = pa.PolygonAperture(1300.0 * np.cos(th), 1300.0 * np.sin(th))
    lens.surfaces[3].aperture = pa.RadialAperture(r_max=200.0) - pa.RectangularAperture(
        -10, 10, -10, 10)
    yield "big_polygon_and_boolean", lens


@pytest.mark.parametrize("backend", ["torch", "numpy"])
@pytest.mark.parametrize("walk", ["native", "python"])
def test_every_attribute_the_packer_reads_is_seen_by_the_change_detector(backend, walk):
    """Property test: trace which instance attributes `packer.pack_optic` reads on a live
    optic, change each of them the way user code can (rebind / in-place write) and require
    the token to differ.  The list is derived from the packer at test time -- a new read in
    packer.py is covered without touching this file."""
    from optiland_amd import build, fingerprint as fp
    from optiland_amd.packer import pack_optic
    be_ = _live.import_reference()
    if walk == "native":
        build.build_fptoken()
        if fp._NATIVE is None:
            fp._NATIVE = fp._load_native()
        if not fp.use_native(True):
            pytest.skip("native token extension not available")
    else:
        fp.use_native(False)
    be_.set_backend(backend)
    if backend == "torch":
        be_.set_device("cpu")
        be_.set_precision("float64")
    try:
        checked = 0
        for label, lens in _systems(be_):
            w = float(lens.primary_wavelength)
            reads = _traced_reads(lens, lambda: pack_optic(lens, wavelengths=[w]))
            assert len(reads) > 50, label
            for obj, name in reads:
                if name in fp._SURFACE_SKIP or type(obj.__dict__[name]) is dict:
                    continue  # recorded arrays; caches
                for apply, undo in _value_changes(obj, name):
                    t0, _k0 = fp.optic_token(lens, w)
                    apply()
                    try:
                        t1, _k1 = fp.optic_token(lens, w)
                    finally:
                        undo()
                    assert t0 != t1, (f"{label}: a change of {type(obj).__name__}.{name} is "
                                      "invisible to optic_token")
                    checked += 1
        assert checked > 150
    finally:
        be_.set_backend("numpy")
        fp.use_native(True)


def test_in_place_edit_of_a_big_polygon_is_detected_and_param_data_writes_force_a_repack(be):
    """VERDICT r2 #5 / ADVICE: (a) a 20 000-vertex polygon aperture edited IN PLACE (numpy
    array beyond the by-value size); (b) `param.data.clamp_()` on a variable bound into the
    optic (optimization/optimizer/torch/base.py:90-94) -- no `_version` bump."""
    import torch
    from optiland import physical_apertures as pa
    from optiland_amd import fingerprint as fp
    lens = _live.rc_asphere()
    th = np.linspace(0.0, 2 * np.pi, 20000, endpoint=False)
    poly = pa.PolygonAperture(1300.0 * np.cos(th), 1300.0 * np.sin(th))
    lens.surfaces[2].aperture = poly
    w = float(lens.primary_wavelength)
    t0, _ = fp.optic_token(lens, w)
    assert fp.optic_token(lens, w)[0] == t0          # stable
    v = poly.vertices
    v = v if isinstance(v, np.ndarray) else None
    if v is not None:
        assert v.size > fp._BIG_ARRAY
        v[12345, 0] *= 0.5                            # in place: same object, same shape
        assert fp.optic_token(lens, w)[0] != t0
    # a Parameter bound into the prescription: never trusted
    p = torch.nn.Parameter(torch.tensor(-1.001152, dtype=torch.float64))
    lens.surfaces[2].geometry.k = p
    a, _ = fp.optic_token(lens, w)
    with torch.no_grad():
        p.data.clamp_(-1.0, 0.0)                      # the optimiser's own idiom
    b, _ = fp.optic_token(lens, w)
    assert a != b
    assert fp.optic_token(lens, w)[0] != b            # ... and never equal to itself either
