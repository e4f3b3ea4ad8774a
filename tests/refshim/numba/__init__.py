"""Identity stub for `numba` (absent in this image).

Test infrastructure only: lets the *reference* package import in this
container so golden vectors can be generated from it.  Nothing on the traced
hot path reaches a jitted function (numba is only used by BSDF scatter,
Huygens PSF and NURBS in the reference).
"""


def _identity_decorator(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


njit = jit = vectorize = guvectorize = _identity_decorator
prange = range
