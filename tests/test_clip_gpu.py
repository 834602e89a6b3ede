"""GPU parity tests of the B1 surface (szClipBatch / szPolyclipBatch), through the C ABI.
Bar: bit-exact rings (count, order, first vertex, direction, coordinates) and status codes."""
import numpy as np
import pytest

import clipgen as g

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", g.GOLDEN_SETS)
def test_gpu_matches_golden(ctx, name):
    inp, res = g.load_golden(name)
    ps = g.ArraysPairSet(inp)
    for m in range(4):
        r = g.run_gpu(ctx, ps, m)
        assert g.compare(res[m], r, len(ps)) == []


@pytest.mark.parametrize("gen,n,kw", [
    (g.convex_pairs, 20000, {}),
    (g.shape_pairs, 3000, {}),
    (g.lattice_pairs, 8000, {}),
    (g.lattice_pairs, 1500, {"grid": 10, "max_verts": 30}),  # exercises the 8x retry tier
    (g.lattice_pairs, 4000, {"grid": 3, "max_verts": 12}),
    (g.rect_pairs, 4000, {}),
    (g.star_pairs, 400, {}),      # long outlines: every slice class of the sweep pass, and pairs whose boxes are apart
])
def test_gpu_matches_cpu_checkers(ctx, gen, n, kw):
    ps = gen(n, seed=777, **kw)
    ref = g.ref_lib()
    for m in range(4):
        r = g.run_gpu(ctx, ps, m)
        h = g.run_host(ps, m)
        assert g.compare(h, r, len(ps)) == []
        if ref is not None:
            assert g.compare(g.run_ref(ps, m), r, len(ps)) == []


def test_gpu_fill_rules(ctx):
    # non-default fill rules of the mex (4th/5th argument): checked against the host build of the engine
    ps = g.lattice_pairs(1500, seed=31, grid=5, max_verts=10)
    for fs, fc in [(1, 1), (2, 2), (3, 3), (1, 0), (0, 2)]:
        for m in range(4):
            assert g.compare(g.run_host(ps, m, fs, fc), g.run_gpu(ctx, ps, m, fs, fc), len(ps)) == []


def test_polyclip_doubles(ctx):
    # polyclip.m:63-80: double -> int64 (x 2^32, MATLAB rounding) -> clip -> double
    ps = g.convex_pairs(500, seed=5)
    ax, ay, aoff, bx, by, boff = ps.arrays()
    fax, fay, fbx, fby = (v.astype(np.float64) / g.SCALE for v in (ax, ay, bx, by))
    ox, oy, roff, proff, st = ctx.polyclip_batch(fax, fay, aoff, fbx, fby, boff, 1)
    r = g.run_gpu(ctx, ps, 1)
    nr = r.pair_ring_off[len(ps)]; nv = r.ring_off[nr]
    assert np.array_equal(proff, r.pair_ring_off) and np.array_equal(roff[:nr + 1], r.ring_off[:nr + 1])
    assert np.array_equal(ox[:nv], r.x[:nv] / g.SCALE) and np.array_equal(oy[:nv], r.y[:nv] / g.SCALE)


@pytest.mark.parametrize("gen,n", [(g.convex_pairs, 2000), (g.neighbour_pairs, 1500), (g.lattice_pairs, 1500), (g.shape_pairs, 250)])
def test_polyclip_doubles_against_the_reference(ctx, gen, n):
    # szPolyclipBatch converts the whole batch on the device (k_rings_to_fixed) and takes the front pass and the sweeps of
    # szClipBatch (round 4): against the reference's own Clipper on the rings as polyclip.m:63-66 would convert them
    ps = gen(n, seed=77)
    ax, ay, aoff, bx, by, boff = ps.arrays()
    fl = [v.astype(np.float64) / g.SCALE for v in (ax, ay, bx, by)]
    fx = [np.rint(v * g.SCALE).astype(np.int64) for v in fl]
    ps2 = g.ArraysPairSet((fx[0], fx[1], aoff, fx[2], fx[3], boff))
    for m in range(4):
        ox, oy, roff, proff, st = ctx.polyclip_batch(fl[0], fl[1], aoff, fl[2], fl[3], boff, m)
        chk = g.run_ref(ps2, m) if g.ref_lib() is not None else g.run_host(ps2, m)
        nr = chk.pair_ring_off[len(ps2)]; nv = chk.ring_off[nr]
        assert np.array_equal(st, chk.status[:len(ps2)])
        assert np.array_equal(proff, chk.pair_ring_off) and np.array_equal(roff[:nr + 1], chk.ring_off[:nr + 1])
        assert np.array_equal(ox[:nv], chk.x[:nv] / g.SCALE) and np.array_equal(oy[:nv], chk.y[:nv] / g.SCALE)


def test_empty_and_degenerate(ctx):
    ps = g.PairSet()
    sq = np.array([[0, 0], [4, 0], [4, 4], [0, 4]], np.int64) << 32
    ps.add(sq, sq[:2], fixed=True)               # clip ring with 2 vertices: rejected, empty result
    ps.add(sq[:1], sq[:1], fixed=True)           # both degenerate: the mex raises "Clipper Error."
    ps.add(sq, np.zeros((0, 2), np.int64), fixed=True)
    ps.add(sq, sq + (1 << 40), fixed=True)       # disjoint
    ps.add(sq, sq, fixed=True)                   # identical
    for m in range(4):
        r, h = g.run_gpu(ctx, ps, m), g.run_host(ps, m)
        assert g.compare(h, r, len(ps)) == []
    r = g.run_gpu(ctx, ps, 1)
    assert r.status.tolist() == [0, 1, 0, 0, 0]
    # Both rings degenerate, their boxes APART and no two vertices of a height (found by tests/campaign_clip_gpu.py, round 4:
    # the front pass answered "apart, empty" where the reference has no local minimum and fails): two distinct points and a
    # closing copy; every vertex on one line; and, for contrast, one degenerate ring beside a proper one (empty, no error).
    seg_a = np.array([[1, 6], [1, 5], [1, 6]], np.int64) << 32
    seg_b = np.array([[2, 0], [0, 1], [2, 0]], np.int64) << 32
    line_a = np.array([[10, 10], [12, 12], [11, 11], [14, 14]], np.int64) << 32
    line_b = np.array([[0, 0], [3, 1], [6, 2]], np.int64) << 32
    tri = np.array([[20, 0], [23, 1], [21, 4]], np.int64) << 32
    d = g.PairSet()
    d.add(seg_a, seg_b, fixed=True); d.add(line_a, line_b, fixed=True); d.add(seg_a, tri, fixed=True); d.add(tri, line_b, fixed=True)
    ref = g.ref_lib()
    for m in range(4):
        r, h = g.run_gpu(ctx, d, m), g.run_host(d, m)
        assert g.compare(h, r, len(d)) == []
        if ref is not None:
            assert g.compare(g.run_ref(d, m), r, len(d)) == []
    assert g.run_gpu(ctx, d, 1).status.tolist() == [1, 1, 0, 0]
    # zero pairs
    e = g.PairSet()
    ox, oy, roff, proff, st = ctx.clip_batch(*e.arrays(), 1)
    assert proff.tolist() == [0] and len(st) == 0


def test_bad_method_is_reported(ctx):
    ps = g.rect_pairs(3, seed=1)
    r = g.run_gpu(ctx, ps, 7)
    assert r.status.tolist() == [2, 2, 2]
