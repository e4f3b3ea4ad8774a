"""A cheap change-detector for live reference objects, so that the drop-in does not
re-pack an unchanged `Optic` on every trace.

`packer.pack_optic` reads ~700 backend scalars per call; on the `cuda` device each one is
a blocking device-to-host copy (2.8 ms per DoubleGauss pack on the MI355X box, three
times the whole 1e7-ray trace).  The reference lets callers mutate anything at any time
(`optic.set_radius`, optimisation variables, `surface.geometry.k = ...`), so the packed
table can only be reused if NOTHING the packer reads has changed.  `optic_token()` walks
exactly those objects and collects, without touching the device:

  * python / numpy scalars and strings  -> their value
  * torch tensors                       -> (id, `_version`): a new tensor object or any
                                           in-place write changes the token
  * tensors that REQUIRE GRAD           -> never equal to anything (a fresh serial number):
                                           the reference's optimisers write their variables
                                           through `param.data` (optimization/optimizer/torch/
                                           base.py:90-94, variable/torch.py:59, torch_backend.py
                                           :214), which leaves `_version` alone -- such a
                                           tensor anywhere in the prescription means "pack
                                           again", every time
  * numpy arrays up to 16 384 elements  -> their bytes
  * larger numpy arrays                 -> shape + a 64-bit digest of their CONTENTS (xxh3
                                           when the module is there, else zlib.adler32 over
                                           the buffer): an in-place edit of a 20 000-vertex
                                           polygon aperture is seen
  * lists / tuples                      -> element tokens
  * other objects                       -> (class name, id)  [identity]
  * dicts                               -> ignored (the reference keeps caches in them)

Tokens are compared with `==`; the objects whose ids appear in a token are kept alive
next to it (`keep`) so that an id cannot be recycled while the token is cached.  What this
still cannot see is a write through `.data` / `set_` of a tensor that does NOT require grad
(no `_version` bump; nothing in the reference does that to a prescription value);
`OPTILAND_HIP_PACK_CACHE=0` turns the memo off.  tests/test_fingerprint.py derives the set
of attributes the packer really reads from a traced `pack_optic` run and mutates each one.
"""

from __future__ import annotations

import os

import numpy as np
import torch

ENABLED = os.environ.get("OPTILAND_HIP_PACK_CACHE", "1") != "0"
_SCALARS = (float, int, bool, str, type(None), complex)
_BIG_ARRAY = 1 << 14

# recorded per-trace arrays and back-pointers: never part of the optical prescription
_SURFACE_SKIP = frozenset(("x", "y", "z", "u", "L", "M", "N", "intensity", "aoi", "opd",
                           "_listeners", "parent_surface"))


_Tensor = torch.Tensor
_NoneType = type(None)
_SERIAL = [0]

try:  # content digest of big arrays
    import xxhash as _xx

    def _digest(buf) -> int:
        return _xx.xxh3_64_intdigest(buf)
except Exception:  # noqa: BLE001 - optional accelerator
    import zlib as _zlib

    def _digest(buf) -> int:
        return _zlib.adler32(buf)


def _grad_tensor_token(v):
    """A tensor that requires grad: `.data` writes do not bump `_version`, so it is never
    trusted -- every token of it is unique."""
    _SERIAL[0] += 1
    return ("G", id(v), _SERIAL[0])


def _big_array_token(v):
    a = v if v.flags.c_contiguous else np.ascontiguousarray(v)
    return ("A", v.shape, str(v.dtype), _digest(memoryview(a).cast("B")))


def _tensor_token(v, keep):
    keep.append(v)
    if v.requires_grad:
        return _grad_tensor_token(v)
    return ("T", id(v), v._version)


def _tok_py(v, keep):
    t = type(v)
    if t is float or t is _Tensor or t is int or t is str or t is bool or t is _NoneType:
        if t is _Tensor:
            return _tensor_token(v, keep)
        return v
    if t is complex:
        return v
    if isinstance(v, _Tensor):  # Parameter and other subclasses
        return _tensor_token(v, keep)
    if isinstance(v, np.ndarray):
        if v.size <= _BIG_ARRAY:
            return ("A", v.shape, v.tobytes())
        keep.append(v)
        return _big_array_token(v)
    if isinstance(v, np.generic):
        return v.item()
    if t is list or t is tuple:
        return tuple([_tok_py(e, keep) for e in v])
    if t is dict:
        return None
    keep.append(v)
    return ("O", t.__name__, id(v))


def _dict_tokens_py(d, skip, keep):
    if skip is None:
        return [_tok_py(v, keep) for v in d.values()]
    return [(k, _tok_py(v, keep)) for k, v in d.items() if k not in skip]


def _load_native():
    """The same two functions from csrc/fptoken.c (`build.build_fptoken()`); None when the
    extension has not been built -- the walk is host bookkeeping, not the compute path, so a
    missing accelerator only costs time (OPTILAND_HIP_NATIVE_TOKEN=0 forces the Python one)."""
    if os.environ.get("OPTILAND_HIP_NATIVE_TOKEN", "1") == "0":
        return None
    import importlib.machinery
    import importlib.util
    from . import build as _build
    path = _build.fptoken_path()
    if not os.path.exists(path):
        return None
    try:
        loader = importlib.machinery.ExtensionFileLoader("_fptoken", path)
        spec = importlib.util.spec_from_loader("_fptoken", loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
        mod.configure(torch.Tensor, np.ndarray, np.generic, _BIG_ARRAY, _grad_tensor_token,
                      _big_array_token)
        return mod
    except Exception:  # noqa: BLE001 - stale / foreign binary: fall back to the Python walk
        return None


_NATIVE = _load_native()
_tok = _NATIVE.tok if _NATIVE is not None else _tok_py
_dict_tokens = _NATIVE.dict_tokens if _NATIVE is not None else _dict_tokens_py


def use_native(flag: bool) -> bool:
    """Switch between the native and the Python walk (tests); returns whether the native
    one is active afterwards."""
    global _tok, _dict_tokens, _obj, surface_token
    if flag and _NATIVE is not None:
        _tok, _dict_tokens, _obj = _NATIVE.tok, _NATIVE.dict_tokens, _NATIVE.obj
        surface_token = _surface_token_native
        return True
    _tok, _dict_tokens, _obj = _tok_py, _dict_tokens_py, _obj_py
    surface_token = _surface_token_py
    return False


def _obj_py(o, keep, memo, skip=None):
    """One level of `o.__dict__`: values tokenised in dict order, sub-objects by identity
    (an added / removed attribute changes the length of the list)."""
    if o is None:
        return None
    key = id(o)
    hit = memo.get(key)
    if hit is not None:
        return hit
    keep.append(o)
    d = getattr(o, "__dict__", None)
    if d is None:
        out = _tok(o, keep)
    else:
        out = (type(o).__name__, key, _dict_tokens(d, skip, keep))
    memo[key] = out
    return out


_obj = _NATIVE.obj if _NATIVE is not None else _obj_py


def _cs(cs, keep, memo):
    out = []
    while cs is not None:
        out.append(_obj(cs, keep, memo))
        cs = getattr(cs, "reference_cs", None)
    return tuple(out)


def _aperture(ap, keep, memo):
    if ap is None:
        return None
    a, b = getattr(ap, "a", None), getattr(ap, "b", None)
    sub = tuple(_aperture(c, keep, memo) for c in (a, b)
                if c is not None and hasattr(c, "__dict__"))
    return (_obj(ap, keep, memo), sub)


def _coating(c, keep, memo):
    if c is None:
        return None
    return (_obj(c, keep, memo),
            _obj(getattr(c, "jones", None) or getattr(c, "_jones", None), keep, memo))


def _material(m, keep, memo):
    if m is None:
        return None
    return (_obj(m, keep, memo), type(getattr(m, "propagation_model", None)).__name__)


def _surface_token_py(s, keep, memo):
    geom = s.geometry
    im = getattr(s, "interaction_model", None)
    return (
        type(s).__name__, id(s),
        _obj(geom, keep, memo), _cs(getattr(geom, "cs", None), keep, memo),
        _obj(getattr(geom, "zernike", None), keep, memo),
        _material(getattr(s, "material_pre", None), keep, memo),
        _material(getattr(s, "material_post", None), keep, memo),
        _aperture(getattr(s, "aperture", None), keep, memo),
        _obj(im, keep, memo, _SURFACE_SKIP),
        _coating(getattr(im, "coating", None), keep, memo),
        _tok(getattr(s, "thickness", None), keep), bool(getattr(s, "is_stop", False)),
    )


def _surface_token_native(s, keep, memo):
    return _NATIVE.surface_token(s, keep, memo, _SURFACE_SKIP)


surface_token = _surface_token_native if _NATIVE is not None else _surface_token_py


def _packing_options():
    from . import system as S
    return (bool(S.OPTIONS["reference_root"]), bool(S.OPTIONS["reference_newton"]))


def surfaces_token(surfaces, wavelength):
    """Token of a SurfaceGroup's surface list (what `packer.pack_surfaces` reads)."""
    keep, memo = [], {}
    return (float(wavelength), tuple(surface_token(s, keep, memo) for s in surfaces),
            _packing_options()), keep


def optic_token(optic, wavelength):
    """Token of everything `packer.pack_optic(optic, [wavelength])` reads."""
    keep, memo = [], {}
    fields = optic.fields
    pol = getattr(optic, "polarization", "ignore")  # "ignore" | PolarizationState
    tok = (
        float(wavelength),
        tuple(surface_token(s, keep, memo) for s in optic.surfaces.surfaces),
        _obj(optic.aperture, keep, memo),
        tuple(_obj(f, keep, memo) for f in fields.fields),
        type(getattr(fields, "field_definition", None)).__name__,
        tuple(_obj(w, keep, memo) for w in optic.wavelengths.wavelengths),
        _obj(getattr(optic, "apodization", None), keep, memo),
        pol if isinstance(pol, str) else _obj(pol, keep, memo),
        bool(getattr(optic, "obj_space_telecentric", False)),
        getattr(optic.ray_tracer, "ray_aiming_config", {}).get("mode", "paraxial"),
        _packing_options(),
    )
    return tok, keep
