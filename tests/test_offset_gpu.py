"""clipper(P, delta, type [, info]) on the GPU (szOffsetPaths: path construction on the host, clean-up union by the
sweep kernel) against the reference's ClipperOffset, through the C ABI and through the clipper gateway."""
import numpy as np
import pytest

import clipgen as g
import mexharness as mx

pytestmark = pytest.mark.gpu


def fx(a):
    return np.round(np.asarray(a, float) * 2 ** 32).astype(np.int64)


def same(a, b):
    return len(a) == len(b) and all(x.shape == y.shape and np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("seed", range(4))
def test_offset_paths_match_the_reference(ctx, seed):
    shapes, _ = g.load_shapes()
    rng = np.random.default_rng(100 + seed)
    n = 0
    for trial in range(30):
        s = shapes[rng.integers(len(shapes))]
        ring = fx((s - s.mean(0)) * rng.uniform(0.5, 50))
        rings = [ring]
        if trial % 5 == 0:
            c = ring.mean(0).astype(np.int64)
            r = int(0.05 * np.abs(ring - c).max())
            rings.append(np.array([[c[0] - r, c[1] - r], [c[0] - r, c[1] + r], [c[0] + r, c[1] + r], [c[0] + r, c[1] - r]], np.int64))
        for join, info in ((0, 0.0), (1, float(rng.uniform(0.0005, 0.05)) * 2 ** 32), (2, 0.0), (2, float(rng.uniform(2, 6)))):
            delta = float(rng.choice([-1, 1]) * rng.uniform(0.002, 0.3) * np.abs(ring).max())
            assert same(g.run_ref_offset(rings, delta, join, info), ctx.offset_paths(rings, delta, join, info)), (seed, trial, join, info, delta)
            n += 1
    assert n == 120


def test_offset_edge_cases(ctx):
    sq = np.array([[0, 0], [100, 0], [100, 100], [0, 100]], np.int64)
    assert sorted(map(tuple, ctx.offset_paths([sq], 10.0, 2)[0].tolist())) == sorted([(-10, -10), (110, -10), (110, 110), (-10, 110)])
    assert ctx.offset_paths([sq], -60.0, 2) == []
    for rings in ([np.array([[0, 0], [10, 0]], np.int64)], [sq, np.array([[0, 0], [0, 0], [7, 7]], np.int64)]):
        for delta in (3.0, -3.0, 0.0):
            assert same(g.run_ref_offset(rings, delta, 2), ctx.offset_paths(rings, delta, 2)), delta


def test_clipper_gateway_offset_mode():
    gw = mx.Gateway("clipper_mex.cpp", "clipper")
    shapes, _ = g.load_shapes()
    ring = fx((shapes[7] - shapes[7].mean(0)) * 3.0)
    s = mx.struct(1, ["x", "y"])
    mx.set_field(s, 0, "x", mx.numeric(ring[:, 0], "int64")); mx.set_field(s, 0, "y", mx.numeric(ring[:, 1], "int64"))
    for args, (join, info) in (([mx.scalar(2.0e9)], (2, 0.0)), ([mx.scalar(-2.0e9), mx.string("square")], (0, 0.0)),
                               ([mx.scalar(2.0e9), mx.string("r"), mx.scalar(1.0e7)], (1, 1.0e7)), ([mx.scalar(2.0e9), mx.string("Miter"), mx.scalar(4.0)], (2, 4.0))):
        out, = gw.call(1, s, *args)
        delta = mx.to_numpy(args[0]).ravel()[0]
        want = g.run_ref_offset([ring], float(delta), join, info)
        assert mx.numel(out) == len(want) and len(want) >= 1
        for r, w in enumerate(want):
            x = mx.to_numpy(mx.get_field(out, r, "x")).ravel(); y = mx.to_numpy(mx.get_field(out, r, "y")).ravel()
            assert x.dtype == np.float64 and np.array_equal(x, w[:, 0].astype(float)) and np.array_equal(y, w[:, 1].astype(float))


def test_offset_golden_vectors(ctx):
    for rings, delta, join, info, want in g.offset_golden():
        assert same(want, ctx.offset_paths(rings, delta, join, info)), (delta, join, info)
