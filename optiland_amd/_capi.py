"""ctypes binding of include/optiland_hip.h (the drop-in boundary).

There is deliberately NO fallback: if the HIP extension is missing or cannot be
loaded every entry point raises -- a silent CPU path would void parity claims.
"""

from __future__ import annotations

import ctypes as C
import os

from . import build as _build

_LIB = None


class HipExtensionError(RuntimeError):
    """The HIP shared library is missing / failed to load / returned an error."""


class SurfaceDesc(C.Structure):
    _fields_ = [
        ("geom_kind", C.c_int32),
        ("interaction", C.c_int32),
        ("aperture_kind", C.c_int32),
        ("coating_kind", C.c_int32),
        ("coeff_offset", C.c_int32),
        ("n_coeff", C.c_int32),
        ("max_iter", C.c_int32),
        ("flags", C.c_uint32),
        ("poly_cols", C.c_int32),
        ("reserved_", C.c_int32),
        ("radius", C.c_double),
        ("conic", C.c_double),
        ("tol", C.c_double),
        ("norm_radius", C.c_double),
        ("origin", C.c_double * 3),
        ("rot", C.c_double * 9),
        ("aperture", C.c_double * 4),
        ("coat", C.c_double * 2),
    ]


class SurfaceOptics(C.Structure):
    _fields_ = [("n1", C.c_double), ("n2", C.c_double), ("absorb", C.c_double)]


class RaygenParams(C.Structure):
    _fields_ = [
        ("object_infinite", C.c_int32),
        ("field_kind", C.c_int32),
        ("EPL", C.c_double),
        ("EPD", C.c_double),
        ("max_field", C.c_double),
        ("offset", C.c_double),
        ("z_first", C.c_double),
        ("tele_dz", C.c_double),
        ("apod_a", C.c_double),
        ("apod_b", C.c_double),
        ("apod_kind", C.c_int32),
        ("reserved_", C.c_int32),
    ]


class RaygenInputs(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("hx", "hy", "px", "py", "vx", "vy")] + \
               [(k, C.c_double) for k in ("hx0", "hy0", "vx0", "vy0")] + \
               [("flags", C.c_uint32), ("reserved_", C.c_uint32)]


RAYGEN_CHECK_FIELD, RAYGEN_CHECK_PUPIL, RAYGEN_PRESCALE_PUPIL = 0x1, 0x2, 0x4
SPOT_POLARIZED_OK, SPOT_HITS_LOCAL = 0x8, 0x10   # ol_trace_spot / ol_trace_spot_batch (ABI 10)


class TraceExtras(C.Structure):
    _fields_ = [("spot_slots", C.c_void_p), ("cx", C.c_double), ("cy", C.c_double),
                ("record_first_surface", C.c_int32), ("reserved_", C.c_int32),
                # ABI 7: update_intensity as an epilogue of ol_trace_generate
                ("update_intensity_state", C.c_void_p), ("updated_intensity", C.c_void_p),
                # ABI 11: reference-Newton ranges (OL_SURF_REFERENCE_NEWTON, ol_newton_count)
                ("newton_iterations", C.c_void_p), ("newton_count_surface", C.c_int32),
                ("reserved2_", C.c_int32)]


SPOT_SLOTS = 64


class WavefrontParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("xc", "yc", "zc", "R", "n_image", "opd_ref", "ux",
                                          "uy", "half_epd", "wavelength_um", "nx", "ny", "nz",
                                          "last_thickness", "last_absorb")]


SPOT_BATCH_MAX_CELLS = 32  # OL_SPOT_BATCH_MAX_CELLS


class SpotCell(C.Structure):
    """ol_spot_cell"""
    _fields_ = [("hx", C.c_double), ("hy", C.c_double), ("vx", C.c_double), ("vy", C.c_double),
                ("cx", C.c_double), ("cy", C.c_double), ("wavelength_index", C.c_int32),
                ("reserved_", C.c_int32), ("optics_of", C.c_void_p)]


class PolarizationStateC(C.Structure):
    _fields_ = [
        ("is_polarized", C.c_int32),
        ("reserved_", C.c_int32),
        ("Ex", C.c_double),
        ("Ey", C.c_double),
        ("phase_x", C.c_double),
        ("phase_y", C.c_double),
    ]


EXPORTS = (
    "ol_abi_version",
    "ol_last_error",
    "ol_system_create",
    "ol_system_destroy",
    "ol_system_num_surfaces",
    "ol_trace",
    "ol_generate_rays",
    "ol_polarized_intensity",
    "ol_spot_moments",
    "ol_spot_max_r2",
    "ol_set_tuning",
    "ol_wavefront_opd",
    "ol_trace_spot",
    "ol_trace_ex",
    "ol_radial_energy",
    "ol_irradiance",
    "ol_trace_opd",
    "ol_pupil_fill",
    "ol_trace_generate",
    "ol_system_update",
    "ol_stream_fill",
    "ol_math_probe",
    "ol_pupil_points",
    "ol_wavefront_reference",
    "ol_trace_opd_dev",
    "ol_wavefront_fit",
    "ol_wavefront_opd_fitted",
    "ol_trace_spot_batch",
    "ol_newton_count",
    "ol_arena_alloc",
    "ol_arena_free",
)

F32, F64 = 0, 1
TUNE_RAYS_PER_THREAD, TUNE_COMPACT, TUNE_FIT_GRID, TUNE_RECORD_WG_CAP = 0, 1, 2, 3
ABI_VERSION = 11
OPD_MOMENTS = 12  # kOpdMoments / ol_trace_opd
WAVEFRONT_REFERENCE_DOUBLES = 16  # OL_WAVEFRONT_REFERENCE_DOUBLES
WAVEFRONT_FIT_WORKSPACE_DOUBLES = 32832  # OL_WAVEFRONT_FIT_WORKSPACE_DOUBLES
FIT_CENTROID, FIT_BEST_FIT = 0, 1
FIT_NO_VALID, FIT_TOO_FEW, FIT_NO_ALIVE, FIT_SINGULAR = 1, 2, 4, 8
FIT_STD_DDOF1, FIT_PISTON_SKIPS_NAN = 1, 2  # the torch backend's flavour: both


def library_path() -> str:
    return os.environ.get("OPTILAND_HIP_LIBRARY", _build.library_path())


def load():
    """Load liboptiland_hip.so (once).  Raises HipExtensionError when absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise HipExtensionError(
            f"HIP extension not built: {path} is missing. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "optiland_amd has no CPU fallback."
        )
    try:
        lib = C.CDLL(path)
    except OSError as exc:  # e.g. libamdhip64 missing
        raise HipExtensionError(f"cannot load {path}: {exc}") from exc
    if hasattr(lib, "ol_hostmath_harness"):
        # tests/hostmath builds the kernel arithmetic for the host as a checker; it is not
        # an implementation of this package and must never be picked up as one
        raise HipExtensionError(f"{path} is the host-math TEST harness, not the HIP extension")
    bind(lib, path)
    _LIB = lib
    return lib


def bind(lib, path: str = "?"):
    """Declare the prototypes of include/optiland_hip.h on a loaded library and check its
    ABI version."""
    vp, i32, i64, u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32
    lib.ol_abi_version.restype = i32
    lib.ol_abi_version.argtypes = []
    lib.ol_last_error.restype = C.c_char_p
    lib.ol_last_error.argtypes = []
    lib.ol_system_create.restype = C.c_int
    lib.ol_system_create.argtypes = [vp, i32, vp, i32, vp, i32, C.POINTER(vp)]
    lib.ol_system_destroy.restype = None
    lib.ol_system_destroy.argtypes = [vp]
    lib.ol_system_num_surfaces.restype = i32
    lib.ol_system_num_surfaces.argtypes = [vp]
    lib.ol_trace.restype = C.c_int
    lib.ol_trace.argtypes = [vp, C.c_int, i64, C.POINTER(vp), i32, vp, i64, vp, i32, i32, u32,
                             vp, vp]
    lib.ol_trace_ex.restype = C.c_int
    lib.ol_trace_ex.argtypes = [vp, C.c_int, i64, C.POINTER(vp), i32, vp, i64, vp, i32, i32, u32,
                                vp, vp, vp]
    lib.ol_generate_rays.restype = C.c_int
    lib.ol_generate_rays.argtypes = [vp, C.c_int, i64, vp, C.POINTER(vp), vp, vp]
    lib.ol_polarized_intensity.restype = C.c_int
    lib.ol_polarized_intensity.argtypes = [C.c_int, i64, vp, i32, C.POINTER(vp), vp, vp, vp, vp,
                                           vp]
    lib.ol_spot_moments.restype = C.c_int
    lib.ol_spot_moments.argtypes = [C.c_int, i64, vp, vp, vp, vp, vp]
    lib.ol_spot_max_r2.restype = C.c_int
    lib.ol_spot_max_r2.argtypes = [C.c_int, i64, vp, vp, vp, C.c_double, C.c_double, vp, vp]
    lib.ol_wavefront_opd.restype = C.c_int
    lib.ol_wavefront_opd.argtypes = [vp, C.c_int, i64, C.POINTER(vp), vp, vp, vp, vp, vp]
    lib.ol_trace_spot.restype = C.c_int
    lib.ol_trace_spot.argtypes = [vp, C.c_int, i64, vp, vp, C.c_double, C.c_double, i32,
                                  C.POINTER(vp), vp, vp, vp]
    lib.ol_irradiance.restype = C.c_int
    lib.ol_irradiance.argtypes = [C.c_int, i64, vp, vp, vp, vp, i32, vp, i32, vp, vp]
    lib.ol_radial_energy.restype = C.c_int
    lib.ol_radial_energy.argtypes = [C.c_int, i64, vp, vp, vp, C.c_double, C.c_double, vp, i32,
                                     vp, vp]
    lib.ol_set_tuning.restype = C.c_int
    lib.ol_set_tuning.argtypes = [i32, i32]
    lib.ol_trace_opd.restype = C.c_int
    lib.ol_trace_opd.argtypes = [vp, i32, i64, vp, vp, vp, i32, vp, vp, C.POINTER(vp), vp, vp, vp]
    lib.ol_pupil_fill.restype = C.c_int
    lib.ol_pupil_fill.argtypes = [i32, i64, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]
    have = lib.ol_abi_version()
    if have == 5 and os.environ.get("OPTILAND_HIP_ALLOW_ABI5") == "1":
        # A/B runs against the round-2 library (tools/gpu_r06.sh cycles_ab / ab6): ABI 5 lacks
        # ol_trace_generate and the record_first_surface extra; the engine takes the
        # two-launch path when the symbol is missing
        return lib
    if have != ABI_VERSION:
        raise HipExtensionError(
            f"{path}: ABI version {have} != expected {ABI_VERSION}; rebuild"
        )
    # (entry points newer than ABI 5 are bound BELOW the version check: a stale library then
    # says "ABI version X != expected" instead of a ctypes "undefined symbol")
    lib.ol_wavefront_reference.restype = C.c_int
    lib.ol_wavefront_reference.argtypes = [vp, i32, vp, vp, vp, C.c_double, i32, i32, vp, vp, vp,
                                           vp]
    lib.ol_trace_opd_dev.restype = C.c_int
    lib.ol_trace_opd_dev.argtypes = [vp, i32, i64, vp, vp, vp, i32, vp, vp, C.POINTER(vp), vp,
                                     vp, vp]
    lib.ol_wavefront_fit.restype = C.c_int
    lib.ol_wavefront_fit.argtypes = [i32, vp, C.c_double, C.c_uint32, i32, i64, C.POINTER(vp), vp, vp,
                                     vp, vp, vp, vp]
    lib.ol_wavefront_opd_fitted.restype = C.c_int
    lib.ol_wavefront_opd_fitted.argtypes = [i64, C.POINTER(vp), vp, vp, vp, vp, C.POINTER(vp), vp]
    lib.ol_system_update.restype = C.c_int
    lib.ol_system_update.argtypes = [vp, vp, i32, vp, i32, vp, i32, vp]
    lib.ol_trace_generate.restype = C.c_int
    lib.ol_trace_generate.argtypes = [vp, C.c_int, i64, vp, vp, i32, vp, i64, C.POINTER(vp), vp,
                                      u32, vp, vp, vp]
    lib.ol_trace_spot_batch.restype = C.c_int
    lib.ol_trace_spot_batch.argtypes = [vp, C.c_int, i64, vp, vp, i32, vp, vp, i64, vp, vp, vp]
    lib.ol_newton_count.restype = C.c_int
    lib.ol_newton_count.argtypes = [vp, C.c_int, i64, C.POINTER(vp), i32, i32, i32, vp, i32, vp]
    lib.ol_arena_alloc.restype = C.c_int
    lib.ol_arena_alloc.argtypes = [i64, C.POINTER(vp)]
    lib.ol_arena_free.restype = C.c_int
    lib.ol_arena_free.argtypes = [vp]
    lib.ol_pupil_points.restype = C.c_int
    lib.ol_pupil_points.argtypes = [i32, i32, C.c_int, i64, vp, vp, vp, vp, vp]
    lib.ol_math_probe.restype = C.c_int
    lib.ol_math_probe.argtypes = [i32, C.c_int, i64, vp, vp, vp, vp]
    lib.ol_stream_fill.restype = C.c_int
    lib.ol_stream_fill.argtypes = [vp, i64, i32, i32, u32, vp]
    return lib


def check(rc: int, what: str, lib=None) -> None:
    if rc != 0:
        msg = (lib or load()).ol_last_error().decode("utf-8", "replace")
        if "Polarization must be set" in msg or msg.startswith("Normalized "):
            # same exception type/text as rays/ray_generator.py:89-94 and
            # raytrace/real_ray_tracer.py:170-173
            raise ValueError(msg)
        raise HipExtensionError(f"{what} failed (code {rc}): {msg}")
