"""Ray containers returned by the HIP tracer (device tensors, struct-of-arrays).

Attribute-for-attribute mirror of the reference's containers so code written
against `Optic.trace()` keeps working: `RealRays` (optiland/rays/real_rays.py:47-89:
x, y, z, L, M, N, i, w, opd, L0, M0, N0, is_normalized) and `PolarizedRays`
(optiland/rays/polarized_rays.py:47-54: p, _i0, _L0, _M0, _N0, update_intensity).
The arrays are torch tensors on the GPU; nothing here computes on the CPU.
"""

from __future__ import annotations

import torch


class RealRays:
    """Final ray state after a trace (views into the recorded block when present)."""

    def __init__(self, x, y, z, L, M, N, intensity, wavelength, opd=None):
        self.x, self.y, self.z = x, y, z
        self.L, self.M, self.N = L, M, N
        self.i = intensity
        self._w = wavelength  # tensor, or a float materialised on first use
        self.opd = opd if opd is not None else torch.zeros_like(x)
        self.L0 = None
        self.M0 = None
        self.N0 = None
        self.is_normalized = True

    @property
    def w(self):
        """Per-ray wavelength array (real_rays.py:68); built on first access."""
        if not isinstance(self._w, torch.Tensor):
            self._w = torch.full_like(self.x, float(self._w))
        return self._w

    @w.setter
    def w(self, value):
        self._w = value

    def __len__(self):
        return int(self.x.numel())

    def planes(self):
        return [self.x, self.y, self.z, self.L, self.M, self.N, self.i, self.opd]


class PolarizedRays(RealRays):
    """Rays carrying the 3x3 polarisation ray-tracing matrix.

    On device the matrix is stored as planes: nine REAL ones (9, N) for uncoated /
    Fresnel / simple / polarizer surfaces (the imaginary part is
    identically zero unless the ray already went NaN through total internal
    reflection), eighteen (real then imaginary) when the system holds a retarder.
    `.p` materialises the reference's (N, 3, 3) complex layout on demand.
    """

    def __init__(self, x, y, z, L, M, N, intensity, wavelength, opd=None, *, engine=None,
                 prt=None, i0=None, k_init=None):
        super().__init__(x, y, z, L, M, N, intensity, wavelength, opd)
        self._engine = engine
        self._prt = prt
        self._i0 = i0
        self._L0, self._M0, self._N0 = k_init if k_init is not None else (None, None, None)

    @property
    def p(self) -> torch.Tensor:
        return prt_to_complex(self._prt)

    def update_intensity(self, state) -> None:
        """rays/polarized_rays.py:122-133 on device (`state`: dict, reference
        PolarizationState, or None for unpolarised)."""
        self.i = self._engine.polarized_intensity(
            self._prt, (self._L0, self._M0, self._N0), self._i0, _state_dict(state))


def new_prt(n: int, dtype, device, complex_: bool) -> torch.Tensor:
    """Identity PRT planes: (9, n) real, or (18, n) real + imaginary."""
    prt = torch.zeros((18 if complex_ else 9, n), dtype=dtype, device=device)
    prt[0].fill_(1), prt[4].fill_(1), prt[8].fill_(1)
    return prt


def prt_to_complex(prt: torch.Tensor) -> torch.Tensor:
    """(9|18, n) device planes -> the reference's (n, 3, 3) complex layout."""
    n = prt.shape[1]
    real = prt[:9].t().reshape(n, 3, 3)
    cdtype = torch.complex64 if real.dtype == torch.float32 else torch.complex128
    if prt.shape[0] == 18:
        return torch.complex(real.contiguous(), prt[9:].t().reshape(n, 3, 3).contiguous())
    out = real.to(cdtype)
    # a NaN real part means the reference's complex entry is NaN+NaNj
    nanmask = torch.isnan(real)
    if nanmask.any():
        out[nanmask] = complex(float("nan"), float("nan"))
    return out


def _state_dict(state):
    if state is None:
        return None
    if isinstance(state, dict):
        return state
    if not getattr(state, "is_polarized", False):
        return {"is_polarized": False}
    f = lambda v: float(v.item()) if hasattr(v, "item") else float(v)  # noqa: E731
    return {"is_polarized": True, "Ex": f(state.Ex), "Ey": f(state.Ey),
            "phase_x": f(state.phase_x), "phase_y": f(state.phase_y)}
