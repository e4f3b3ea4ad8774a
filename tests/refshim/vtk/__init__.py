"""Empty stub for `vtk` (absent in this image); import-time placeholder only."""


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return _Dummy()

    def __call__(self, *a, **k):
        return _Dummy()


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    return _Dummy
