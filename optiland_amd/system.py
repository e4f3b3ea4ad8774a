"""Packed description of a sequential optical system (the "surface table").

`SystemTable` is the host-side POD image of what `ol_system_create`
(include/optiland_hip.h) consumes: one `ol_surface_desc` per traced surface, a
flat coefficient buffer and per-(surface, wavelength) optical constants.  It is
what `SurfaceGroup.trace` walks in the reference
(optiland/surfaces/surface_group.py:245-257), flattened once instead of being
re-discovered through Python attribute access on every call.

The table can be produced from a live reference `Optic` (see
`optiland_amd.packer.pack_optic`) or loaded from a JSON fixture, which is how the
GPU box -- where the reference package does not exist -- gets its systems.
"""

from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass, field

import numpy as np

# ---- enums (mirror include/optiland_hip.h) ---------------------------------
GEOM_PLANE, GEOM_STANDARD, GEOM_EVEN_ASPHERE, GEOM_ZERNIKE = 0, 1, 2, 3
GEOM_ODD_ASPHERE, GEOM_POLYNOMIAL = 4, 5
GEOM_CHEBYSHEV, GEOM_BICONIC, GEOM_TOROIDAL = 6, 7, 8
INTERACT_RECORD_ONLY, INTERACT_REFRACT, INTERACT_REFLECT = 0, 1, 2
AP_NONE, AP_RADIAL, AP_OFFSET_RADIAL, AP_RECTANGULAR, AP_ELLIPTICAL = 0, 1, 2, 3, 4
AP_COMPOSITE = 5
AP_POLYGON = 6
AP_OP_UNION, AP_OP_INTERSECTION, AP_OP_DIFFERENCE = 10, 11, 12
COAT_NONE, COAT_SIMPLE, COAT_FRESNEL, COAT_POLARIZER, COAT_RETARDER = 0, 1, 2, 3, 4
SURF_ROTATED = 0x1
SURF_REFERENCE_ROOT = 0x2   # OL_SURF_REFERENCE_ROOT
SURF_REFERENCE_NEWTON = 0x4  # OL_SURF_REFERENCE_NEWTON
# packing options (process-wide; `integration.enable(reference_root=..., reference_newton=...)`
# sets them)
OPTIONS = {"reference_root": os.environ.get("OPTILAND_HIP_REFERENCE_ROOT", "0") == "1",
           "reference_newton": os.environ.get("OPTILAND_HIP_REFERENCE_NEWTON", "0") == "1"}

STATUS_ZERNIKE_RANGE = 0x1
STATUS_K_PARALLEL_X = 0x2
STATUS_CHEBYSHEV_RANGE = 0x4
FIELD_ANGLE, FIELD_OBJECT_HEIGHT, FIELD_PARAXIAL_IMAGE_HEIGHT = 0, 1, 2
(APOD_NONE, APOD_GAUSSIAN, APOD_COSINE_SQUARED, APOD_HANN, APOD_POLYNOMIAL, APOD_SUPER_GAUSSIAN,
 APOD_TUKEY) = range(7)
STATUS_FIELD_RANGE = 0x8
STATUS_PUPIL_RANGE = 0x10
STATUS_NAN_DIRECTION = 0x20  # informational: a ray ended with a position and no direction

TRACE_WRITE_RAYS = 0x1
TRACE_COMPACT = 0x2
TRACE_FEW_WAVES = 0x10  # the record block is an ordinary allocation (not a placed window)
TRACE_PRT_COMPLEX = 0x4
TRACE_PRT_IDENTITY = 0x8
TRACE_NONUNIT_K = 0x20  # OL_TRACE_NONUNIT_K: polarised bundle whose directions are not unit

GEOM_NAMES = {
    GEOM_PLANE: "plane",
    GEOM_STANDARD: "standard",
    GEOM_EVEN_ASPHERE: "even_asphere",
    GEOM_ZERNIKE: "zernike",
    GEOM_ODD_ASPHERE: "odd_asphere",
    GEOM_POLYNOMIAL: "polynomial",
    GEOM_CHEBYSHEV: "chebyshev",
    GEOM_BICONIC: "biconic",
    GEOM_TOROIDAL: "toroidal",
}

# numpy image of `ol_surface_desc`; align=True reproduces the C layout.
SURFACE_DESC_DTYPE = np.dtype(
    [
        ("geom_kind", np.int32),
        ("interaction", np.int32),
        ("aperture_kind", np.int32),
        ("coating_kind", np.int32),
        ("coeff_offset", np.int32),
        ("n_coeff", np.int32),
        ("max_iter", np.int32),
        ("flags", np.uint32),
        ("poly_cols", np.int32),
        ("reserved_", np.int32),
        ("radius", np.float64),
        ("conic", np.float64),
        ("tol", np.float64),
        ("norm_radius", np.float64),
        ("origin", np.float64, (3,)),
        ("rot", np.float64, (9,)),
        ("aperture", np.float64, (4,)),
        ("coat", np.float64, (2,)),
    ],
    align=True,
)

SURFACE_OPTICS_DTYPE = np.dtype(
    [("n1", np.float64), ("n2", np.float64), ("absorb", np.float64)], align=True
)

RAYGEN_DTYPE = np.dtype(
    [
        ("object_infinite", np.int32),
        ("field_kind", np.int32),
        ("EPL", np.float64),
        ("EPD", np.float64),
        ("max_field", np.float64),
        ("offset", np.float64),
        ("z_first", np.float64),
        ("tele_dz", np.float64),
        ("apod_a", np.float64),
        ("apod_b", np.float64),
        ("apod_kind", np.int32),
        ("reserved_", np.int32),
    ],
    align=True,
)


def _enc(v: float):
    """JSON-safe float (JSON has no inf/nan literals that every parser takes)."""
    v = float(v)
    if math.isinf(v):
        return "inf" if v > 0 else "-inf"
    if math.isnan(v):
        return "nan"
    return v


def _dec(v) -> float:
    return float(v)


@dataclass
class SystemTable:
    """POD surface table + the scalars ray generation needs."""

    surfaces: np.ndarray  # SURFACE_DESC_DTYPE, shape (S+1,)  (object surface first)
    coeffs: np.ndarray  # float64, flat
    optics: np.ndarray  # SURFACE_OPTICS_DTYPE, shape (S+1, W)
    wavelengths: np.ndarray  # float64 (W,) microns
    raygen: dict = field(default_factory=dict)  # EPL, EPD, max_field, offset, ...
    fields: list = field(default_factory=list)  # [(x, y, vx, vy), ...] field points
    polarization: dict | None = None  # None => "ignore"; else PolarizationState
    name: str = ""
    last_thickness: float = 0.0  # optic.surfaces[-1].thickness
    primary_wavelength: float | None = None  # optic.primary_wavelength (microns), if known

    # ------------------------------------------------------------------ info
    def reference_wavelength_index(self, wavelengths=None) -> int:
        """Index, within `wavelengths` (default: the table's), of the optic's primary
        wavelength, 0 if it is not among them -- the reference wavelength of the
        reference's analyses (analysis/spot_diagram/core.py:114-119).  Tables written
        before the primary wavelength was recorded fall back to the middle entry (true for
        every shipped sample)."""
        wl = [float(w) for w in (self.wavelengths if wavelengths is None else wavelengths)]
        if self.primary_wavelength is None:
            return len(wl) // 2
        p = float(self.primary_wavelength)
        return wl.index(p) if p in wl else 0

    @property
    def num_surfaces(self) -> int:
        """Number of table rows (object + traced surfaces)."""
        return int(self.surfaces.shape[0])

    @property
    def num_traced(self) -> int:
        """S: ray-surface intersections per ray (every non-object surface)."""
        return self.num_surfaces - 1

    @property
    def uses_polarization(self) -> bool:
        """Any polarization-dependent coating (SurfaceGroup.uses_polarization)."""
        return bool(np.any(self.surfaces["coating_kind"] >= COAT_FRESNEL))

    @property
    def needs_complex_prt(self) -> bool:
        """Retarder coatings have a complex Jones matrix (jones.py:331-393)."""
        return bool(np.any(self.surfaces["coating_kind"] == COAT_RETARDER))

    def reference_newton_surfaces(self, first: int = 0, last: int | None = None) -> list:
        """Traced Newton-Raphson surfaces of [first, last] that carry SURF_REFERENCE_NEWTON (the
        reference's batch-global stop rule, newton_raphson.py:137-166): their iteration count is
        a property of the batch (`HipSystem._newton_counts`)."""
        memo = self.__dict__.get("_ref_newton")
        if memo is None:
            s = self.surfaces
            memo = self.__dict__["_ref_newton"] = [
                int(i) for i in np.nonzero((s["flags"] & SURF_REFERENCE_NEWTON != 0)
                                           & (s["geom_kind"] != GEOM_PLANE)
                                           & (s["geom_kind"] != GEOM_STANDARD)
                                           & (s["interaction"] != INTERACT_RECORD_ONLY))[0]]
        if not memo:
            return memo
        last = self.num_surfaces - 1 if last is None else last
        return [i for i in memo if first <= i <= last]

    def wavelength_index(self, wavelength: float) -> int:
        """Index of `wavelength` in the table (exact match on the packed value)."""
        w = float(wavelength)
        # (a handful of values, looked up on every trace: a Python loop, not numpy --
        # np.isclose alone cost 15-30 us of a ~230 us small-trace call)
        for i, v in enumerate(self.wavelengths.tolist()):
            if abs(v - w) <= 1e-12:
                return i
        raise KeyError(
            f"wavelength {w} not packed in this SystemTable "
            f"(have {self.wavelengths.tolist()})"
        )

    # ------------------------------------------------------------------ json
    def to_json(self) -> str:
        surf = []
        for s in self.surfaces:
            surf.append(
                {
                    k: (
                        [_enc(x) for x in np.atleast_1d(s[k])]
                        if SURFACE_DESC_DTYPE[k].shape
                        else (
                            _enc(s[k])
                            if SURFACE_DESC_DTYPE[k].kind == "f"
                            else int(s[k])
                        )
                    )
                    for k in SURFACE_DESC_DTYPE.names
                    if k != "reserved_"
                }
            )
        doc = {
            "name": self.name,
            "surfaces": surf,
            "coeffs": [_enc(c) for c in self.coeffs],
            "wavelengths": [float(w) for w in self.wavelengths],
            "optics": [
                [[_enc(o["n1"]), _enc(o["n2"]), _enc(o["absorb"])] for o in row]
                for row in self.optics
            ],
            "raygen": {k: _enc(v) for k, v in self.raygen.items()},
            "fields": [[float(v) for v in f] for f in self.fields],
            "polarization": self.polarization,
            "last_thickness": _enc(self.last_thickness),
        }
        if self.primary_wavelength is not None:
            doc["primary_wavelength"] = float(self.primary_wavelength)
        return json.dumps(doc, indent=1)

    @classmethod
    def from_json(cls, text: str) -> "SystemTable":
        doc = json.loads(text)
        surfaces = np.zeros(len(doc["surfaces"]), dtype=SURFACE_DESC_DTYPE)
        for i, s in enumerate(doc["surfaces"]):
            for k, v in s.items():
                if isinstance(v, list):
                    surfaces[i][k] = [_dec(x) for x in v]
                elif SURFACE_DESC_DTYPE[k].kind == "f":
                    surfaces[i][k] = _dec(v)
                else:
                    surfaces[i][k] = int(v)
        optics = np.zeros(
            (len(doc["optics"]), len(doc["wavelengths"])), dtype=SURFACE_OPTICS_DTYPE
        )
        for i, row in enumerate(doc["optics"]):
            for j, (n1, n2, ab) in enumerate(row):
                optics[i, j] = (_dec(n1), _dec(n2), _dec(ab))
        return cls(
            surfaces=surfaces,
            coeffs=np.array([_dec(c) for c in doc["coeffs"]], dtype=np.float64),
            optics=optics,
            wavelengths=np.array(doc["wavelengths"], dtype=np.float64),
            raygen={k: _dec(v) for k, v in doc.get("raygen", {}).items()},
            fields=[tuple(f) for f in doc.get("fields", [])],
            polarization=doc.get("polarization"),
            name=doc.get("name", ""),
            last_thickness=_dec(doc.get("last_thickness", 0.0)),
            primary_wavelength=doc.get("primary_wavelength"),
        )

    def save(self, path) -> None:
        with open(path, "w") as f:
            f.write(self.to_json())

    @classmethod
    def load(cls, path) -> "SystemTable":
        with open(path) as f:
            return cls.from_json(f.read())
