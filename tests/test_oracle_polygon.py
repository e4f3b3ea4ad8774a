"""The oracle's restatement of matplotlib's point-in-polygon test (the algorithm the
reference's NumPy backend calls for PolygonAperture / FileAperture,
physical_apertures/polygon.py:54-71 -> backend/numpy_backend.py:1125-1137) against the
installed matplotlib itself: random convex / concave / self-intersecting polygons, points
on vertices and edges included."""

import numpy as np
import pytest

from oracle import oracle

mpath = pytest.importorskip("matplotlib.path")


def _cases():
    rng = np.random.default_rng(12)
    yield np.array([[0, 0], [4, 0], [4, 1], [1, 1], [1, 3], [0, 3]], float)   # L shape
    yield np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]], float)              # square
    yield np.array([[0, 0], [2, 2], [0, 2], [2, 0]], float)                  # bow tie
    for n in (3, 5, 8, 13):
        for _ in range(6):
            yield rng.uniform(-3, 3, (n, 2))
    for n in (6, 10):                                                         # star-shaped
        th = np.sort(rng.uniform(0, 2 * np.pi, n))
        r = rng.uniform(0.5, 2.5, n)
        yield np.stack([r * np.cos(th), r * np.sin(th)], 1)


@pytest.mark.parametrize("idx", range(35))
def test_restated_crossings_test_equals_matplotlib(idx):
    cases = list(_cases())
    if idx >= len(cases):
        pytest.skip("no such case")
    v = cases[idx]
    rng = np.random.default_rng(100 + idx)
    pts = [rng.uniform(-3.5, 3.5, (4000, 2)), v.copy(), 0.5 * (v + np.roll(v, -1, 0))]
    # points exactly on horizontal lines through vertices (the yflag >= ties)
    pts.append(np.stack([rng.uniform(-3.5, 3.5, 400), np.repeat(v[:, 1], 400 // len(v) + 1)[:400]], 1))
    pts.append(np.array([[np.nan, 0.0], [0.0, np.inf], [np.nan, np.nan]]))
    p = np.concatenate(pts)
    want = mpath.Path(v).contains_points(p)
    got = oracle.polygon_contains(v, p[:, 0], p[:, 1])
    assert np.array_equal(got, want), (idx, np.flatnonzero(got != want)[:10], p[got != want][:5])
