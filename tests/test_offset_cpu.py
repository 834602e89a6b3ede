"""The offset mode of the clipper gateway, clipper(P, delta, type [, info]) (mexclipper.cpp:146-203): the library's
implementation -- ClipperOffset's path construction on the host (offset.hpp) and the clean-up union by the sweep
(vatti.hpp offset_cleanup), here with the sweep on the host as well -- against the reference's own ClipperOffset
(oracle/_ref), ring for ring and vertex for vertex."""
import numpy as np
import pytest

import clipgen as g

needs_ref = pytest.mark.skipif(g.ref_lib() is None, reason="reference Clipper build not present")


def fx(a):
    return np.round(np.asarray(a, float) * 2 ** 32).astype(np.int64)


def same(a, b):
    return len(a) == len(b) and all(x.shape == y.shape and np.array_equal(x, y) for x, y in zip(a, b))


@needs_ref
def test_square_known_answers():
    sq = np.array([[0, 0], [100, 0], [100, 100], [0, 100]], np.int64)
    out = g.run_host_offset([sq], 10.0, 2)                     # miter, default limit 2: plain corners
    assert same(out, g.run_ref_offset([sq], 10.0, 2)) and len(out) == 1
    assert sorted(map(tuple, out[0].tolist())) == sorted([(-10, -10), (110, -10), (110, 110), (-10, 110)])
    out = g.run_host_offset([sq], -10.0, 2)                    # inset
    assert same(out, g.run_ref_offset([sq], -10.0, 2))
    assert sorted(map(tuple, out[0].tolist())) == sorted([(10, 10), (90, 10), (90, 90), (10, 90)])
    assert g.run_host_offset([sq], -60.0, 2) == [] == g.run_ref_offset([sq], -60.0, 2)      # nothing left
    out = g.run_host_offset([sq[::-1]], 10.0, 0)               # clockwise input is turned round first; square joins
    assert same(out, g.run_ref_offset([sq[::-1]], 10.0, 0)) and len(out[0]) == 8


@needs_ref
@pytest.mark.parametrize("seed", range(6))
def test_floe_outlines_against_the_reference(seed):
    shapes, _ = g.load_shapes()
    rng = np.random.default_rng(seed)
    n = 0
    for trial in range(40):
        s = shapes[rng.integers(len(shapes))]
        ring = fx((s - s.mean(0)) * rng.uniform(0.5, 50))
        rings = [ring]
        if trial % 5 == 0:                                     # a second contour: a hole (opposite sense) well inside
            c = ring.mean(0).astype(np.int64)
            r = int(0.05 * np.abs(ring - c).max())
            rings.append(np.array([[c[0] - r, c[1] - r], [c[0] - r, c[1] + r], [c[0] + r, c[1] + r], [c[0] + r, c[1] - r]], np.int64))
        for join, info in ((0, 0.0), (1, float(rng.uniform(0.0005, 0.05)) * 2 ** 32), (2, 0.0), (2, float(rng.uniform(2, 6)))):
            delta = float(rng.choice([-1, 1]) * rng.uniform(0.002, 0.3) * np.abs(ring).max())
            assert same(g.run_ref_offset(rings, delta, join, info), g.run_host_offset(rings, delta, join, info)), (seed, trial, join, info, delta)
            n += 1
    assert n == 160


@needs_ref
def test_degenerate_inputs():
    tri = np.array([[0, 0], [50, 0], [0, 50]], np.int64)
    for rings in ([np.array([[0, 0], [10, 0]], np.int64)],              # fewer than three distinct points: ignored (AddPath)
                  [np.array([[5, 5]] * 4, np.int64)],
                  [tri, np.array([[0, 0], [0, 0], [7, 7]], np.int64)]):
        for delta in (3.0, -3.0, 0.0):
            assert same(g.run_ref_offset(rings, delta, 2), g.run_host_offset(rings, delta, 2)), (rings, delta)
    dup = np.array([[0, 0], [0, 0], [40, 0], [40, 40], [40, 40], [0, 40], [0, 0]], np.int64)   # repeated and closing points
    for join in (0, 2):
        assert same(g.run_ref_offset([dup], 4.0, join), g.run_host_offset([dup], 4.0, join))


def test_golden_vectors_of_the_reference_build():
    """tests/golden/offset_golden.npz (made by tests/golden/make_offset_golden.py from the reference's ClipperOffset): the
    fixture needs no reference build to check against."""
    cases = g.offset_golden()
    assert len(cases) == 48
    for rings, delta, join, info, want in cases:
        assert same(want, g.run_host_offset(rings, delta, join, info)), (delta, join, info)
