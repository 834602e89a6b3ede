"""CPU suite for the clip primitive (no GPU):
  * the oracle (reference Clipper build, oracle/_ref) reproduces the committed golden vectors and the
    reference's own known answer (private/clipper_test.m:2-12);
  * the device sweep engine, compiled for the host (tests/hostsim), is bit-identical to the golden
    vectors and to the live reference on fresh random input;
  * libstdc++-order sort restatement agrees with std::sort on ties (through the >16 element cases).
"""
import numpy as np
import pytest

import clipgen as g

HAVE_REF = g.ref_lib() is not None


def test_known_answer_clipper_test_m():
    # private/clipper_test.m:2-12 -> intersection is the square (0.5..1)^2, area 0.25
    inp, res = g.load_golden("clipper_test_m")
    r = res[1]
    rings = r.rings(0)
    assert len(rings) == 1
    xy = rings[0] / g.SCALE
    assert np.array_equal(xy, np.array([[1, 1], [0.5, 1], [0.5, 0.5], [1, 0.5]]))
    x, y = xy[:, 0], xy[:, 1]
    area = 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))
    assert area == 0.25


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not built and /root/reference absent")
@pytest.mark.parametrize("name", g.GOLDEN_SETS)
def test_reference_build_reproduces_golden(name):
    inp, res = g.load_golden(name)
    ps = g.ArraysPairSet(inp)
    for m in range(4):
        r = g.run_ref(ps, m)
        assert g.compare(res[m], r, len(ps)) == []


@pytest.mark.parametrize("name", g.GOLDEN_SETS)
def test_device_engine_on_host_matches_golden(name):
    inp, res = g.load_golden(name)
    ps = g.ArraysPairSet(inp)
    for m in range(4):
        h = g.run_host(ps, m)
        assert g.compare(res[m], h, len(ps)) == []


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not built and /root/reference absent")
@pytest.mark.parametrize("gen,n,kw", [
    (g.convex_pairs, 4000, {}),
    (g.shape_pairs, 600, {}),
    (g.lattice_pairs, 4000, {}),
    (g.lattice_pairs, 1500, {"grid": 10, "max_verts": 30}),   # >16 crossings / minima: sort tie order
    (g.lattice_pairs, 3000, {"grid": 3, "max_verts": 12}),    # dense coincidences: joins, horizontals
    (g.rect_pairs, 3000, {}),
])
def test_device_engine_on_host_matches_live_reference(gen, n, kw):
    ps = gen(n, seed=4242, **kw)
    for m in range(4):
        r, h = g.run_ref(ps, m), g.run_host(ps, m)
        assert g.compare(r, h, len(ps)) == []


# The single-minimum fast path (subzero_amd/csrc/convex.hpp): whatever it completes must be the reference's
# answer bit for bit, whatever it declines must go to the general sweep -- on plainly convex input (all taken),
# on input made of exact ties (mostly declined), on rings that share an edge and are then nudged by 1e-12..1e-1 of
# their size, and on one-minimum rings with concave sides whose overlap has several rings.
FAST_SETS = [
    (g.convex_pairs, 6000, {}, 0.99),
    (g.neighbour_pairs, 6000, {}, 0.9),
    (g.monotone_pairs, 6000, {}, 0.5),
    (g.hull_pairs, 6000, {"grid": 6}, 0.0),
    (g.hull_pairs, 6000, {"grid": 40}, 0.2),
    (g.hull_pairs, 6000, {"grid": 1000, "unit": 1 << 20}, 0.8),
    (g.shape_pairs, 1500, {"max_verts": 9}, 0.5),
    (g.lattice_pairs, 4000, {}, 0.0),
    (g.rect_pairs, 2000, {}, 0.0),
]


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not built and /root/reference absent")
@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("gen,n,kw,min_taken", FAST_SETS)
def test_fast_path_matches_live_reference(gen, n, kw, min_taken, mode):
    ps = gen(n, seed=777, **kw)
    r, h = g.run_ref(ps, 1), g.run_host(ps, 1, fast=mode)
    assert g.compare(r, h, len(ps)) == []
    assert h.fast >= min_taken * len(ps), (h.fast, len(ps))


# Rings cut down to their common box before an intersection (trim.hpp): same rings out, vertex for vertex, as the
# reference on the untrimmed input -- on real outlines (most vertices gone), on fracture pieces rubbing along their cut,
# and on lattice-snapped outlines full of ties, where the trim must step aside rather than reorder anything.
TRIM_SETS = [
    (g.shape_pairs, 4000, {"max_verts": 90}, 0.7),
    (g.shape_pairs, 600, {}, 0.7),
    (g.cut_pairs, 4000, {}, 0.7),
    (g.snapped_shape_pairs, 3000, {"grid": 400, "mode": 0}, 1.0),
    (g.snapped_shape_pairs, 3000, {"grid": 400, "mode": 1}, 0.75),
    (g.snapped_shape_pairs, 3000, {"grid": 400, "mode": 2}, 1.0),
    (g.snapped_shape_pairs, 3000, {"grid": 100, "mode": 2}, 1.0),
    (g.convex_pairs, 3000, {}, 1.0),
    (g.lattice_pairs, 2000, {}, 1.0),
]


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not built and /root/reference absent")
@pytest.mark.parametrize("gen,n,kw,max_left", TRIM_SETS)
def test_trimmed_intersection_matches_live_reference(gen, n, kw, max_left):
    ps = gen(n, seed=4242, **kw)
    r, h = g.run_ref(ps, 1), g.run_host(ps, 1, trim=1)
    assert g.compare(r, h, len(ps)) == []
    before, after, declined = h.trim
    assert before > 0 and after <= max_left * before, h.trim


@pytest.mark.parametrize("name", g.GOLDEN_SETS)
def test_trimmed_intersection_matches_golden(name):
    inp, res = g.load_golden(name)
    ps = g.ArraysPairSet(inp)
    h = g.run_host(ps, 1, trim=1)
    assert g.compare(res[1], h, len(ps)) == []


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("name", g.GOLDEN_SETS)
def test_fast_path_matches_golden(name, mode):
    inp, res = g.load_golden(name)
    ps = g.ArraysPairSet(inp)
    h = g.run_host(ps, 1, fast=mode)
    assert g.compare(res[1], h, len(ps)) == []
    if name == "convex":
        assert h.fast > 0.9 * len(ps)


def test_matlab_int64_rounding():
    v = np.array([0.5, -0.5, 1.5, -1.5, 2.4999, -2.5, 1e30, -1e30, 0.49999999999999994])
    out = g.matlab_int64(v)
    assert out.tolist() == [1, -1, 2, -2, 2, -3, 2 ** 63 - 1, -2 ** 63, 0]


def _one_pair(a, b):
    ps = g.PairSet()
    ps.add(np.asarray(a, np.int64) << 32, np.asarray(b, np.int64) << 32, fixed=True)
    return ps


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not built and /root/reference absent")
def test_trim_rules_on_hand_made_rings():
    """What trim.hpp cuts and what it leaves alone, on rings small enough to read (coordinates in metres)."""
    # B: a square; A: a ring whose right part overlaps B and whose left part is a long tail of 9 vertices beyond B
    B = [(10, 0), (20, 3), (21, 11), (9, 12)]
    tail = [(-5, 1), (-9, 2), (-14, 4), (-20, 5), (-22, 6), (-19, 7), (-12, 8), (-8, 9), (-4, 10)]
    A = [(15, -1)] + [(2, 0.5)] + tail + [(3, 10.5), (15, 13)]
    A = [(int(2 * x), int(2 * y)) for x, y in A]; B2 = [(2 * x, 2 * y) for x, y in B]      # integers, no ties in y
    ps = _one_pair(A, B2)
    r, h = g.run_ref(ps, 1), g.run_host(ps, 1, trim=1)
    assert g.compare(r, h, 1) == [] and len(r.rings(0)) == 1
    before, after, declined = h.trim
    assert declined == 0 and 0 < before - after <= len(tail)         # inner vertices of the tail are gone, its ends kept
    # the same pair with one tail vertex moved to the height of a vertex of B: a tie, both rings go in whole
    A_tie = list(A); A_tie[2] = (A_tie[2][0], B2[1][1])            # (vertex 2 is one of the two kept at the start of the run)
    ps = _one_pair(A_tie, B2)
    r, h = g.run_ref(ps, 1), g.run_host(ps, 1, trim=1)
    assert g.compare(r, h, 1) == []
    assert h.trim[2] == 1 and h.trim[0] == h.trim[1]
    # rings whose boxes are apart, and rings too short to have a run: nothing is touched
    far = [(x + 1000, y) for x, y in B2]
    for a_, b_ in ((A, far), (B2, [(x + 5, y + 1) for x, y in B2])):
        ps = _one_pair(a_, b_)
        r, h = g.run_ref(ps, 1), g.run_host(ps, 1, trim=1)
        assert g.compare(r, h, 1) == [] and h.trim[0] == h.trim[1] and h.trim[2] == 0
    # other operations are never trimmed (the union needs every vertex)
    ps = _one_pair(A, B2)
    for m in (0, 2, 3):
        h = g.run_host(ps, m, trim=1)
        assert g.compare(g.run_ref(ps, m), h, 1) == [] and h.trim == (0, 0, 0)


@pytest.mark.parametrize("budget", [10 * 1024, 20 * 1024, 40 * 1024, 160 * 1024])
def test_sweep_inside_a_byte_budget(budget):
    """The slice classes of the clip batch's sweep pass (clip.hip): capacities cut to a byte budget (vatti.hpp caps_budget)
    fit the budget, the sweep stays inside it (canary), what completes there equals the reference, and what outgrows it
    is reported and redone with full capacities."""
    ref = g.ref_lib()
    total = done = 0
    for gen, n, kw in ((g.shape_pairs, 1500, {}), (g.lattice_pairs, 1500, {"grid": 8, "max_verts": 24}), (g.star_pairs, 60, {"apart_every": 0})):
        ps = gen(n, seed=4242, **kw)
        for m in (0, 1, 3):
            h = g.run_host(ps, m, trim=1, budget=budget)
            base = g.run_ref(ps, m) if ref is not None else g.run_host(ps, m)
            assert g.compare(base, h, len(ps)) == []
            total += len(ps); done += h.budget[0]
    assert done > 0.2 * total          # (the small classes hold the short rings; the large ones nearly everything)


@pytest.mark.parametrize("lanes", [16, 32])
def test_team_form_of_the_clip_matches_the_reference(lanes):
    """teamclip.hpp (the clip inside the pair-interaction kernel: one scanline per lane, its lanes as fibers here) against
    the reference's Clipper build: every pair identical or handed on (100; 104 = wants the 32-lane team), and the hand-on
    rates those of the corpora (tests/test_team_clip_gpu.py runs the same on the device)."""
    ref = g.ref_lib()
    for gen, n, least in ((g.convex_pairs, 4000, 0.6), (g.neighbour_pairs, 4000, 0.5), (g.monotone_pairs, 3000, 0.3), (g.hull_pairs, 3000, 0.02)):
        ps = gen(n, seed=99 + lanes)
        t = g.run_team(ps, lanes)
        assert lanes == 16 or not (t.status == 104).any()
        t.status[t.status == 104] = 100
        base = g.run_ref(ps, 1) if ref is not None else g.run_host(ps, 1)
        bad, done = g.compare_where_done(base, t, len(ps))
        assert bad == [], bad[:5]
        assert done >= least * len(ps), (gen.__name__, done)
