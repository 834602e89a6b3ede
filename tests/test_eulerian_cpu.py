"""calc_eulerian_data.m -- the oracle's restatement pinned without MATLAB: the ring-against-rectangle area
(the definition that stands in for polyshape's intersect + area) against the fixed-point 'int' clip of the reference's
own Clipper on real floe outlines, and the conservation laws the cell sums obey."""
import numpy as np
import pytest

import clipgen as g
import oracle
import scenes


def test_ring_box_area_agrees_with_the_reference_clipper():
    if g.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    shapes, _ = g.load_shapes()
    rng = np.random.default_rng(4)
    worst = 0.0
    for k in range(150):
        s = shapes[rng.integers(len(shapes))]
        lo, hi = s.min(0), s.max(0)
        c = lo + rng.uniform(0.1, 0.9, 2) * (hi - lo); half = rng.uniform(0.05, 0.6, 2) * (hi - lo)
        xl, xh, yl, yh = c[0] - half[0], c[0] + half[0], c[1] - half[1], c[1] + half[1]
        a = oracle.ring_box_area(s[:, 0], s[:, 1], xl, xh, yl, yh)
        box = np.array([[xl, yl], [xh, yl], [xh, yh], [xl, yh]])
        rings = g.run_ref_paths([np.round(s * 2.0 ** 32).astype(np.int64)], [np.round(box * 2.0 ** 32).astype(np.int64)], 1)
        b = sum(oracle.ring_area(r[:, 0] / 2.0 ** 32, r[:, 1] / 2.0 ** 32) for r in rings)
        worst = max(worst, abs(a - b) / max(b, 1.0))
    assert worst < 1e-9, worst


def test_cell_sums_conserve_area_and_mass():
    S, P = scenes.pack(300, periodic=False)
    cell = (2e5 / 8) ** 2
    # the fresh pack tiles the box: the cells' ice area adds up to the floes' area, their mass to the floes' mass
    e = oracle.eulerian_data(S, 8, 8, 0, P.c2_boundary, False)
    assert abs(e["area"].sum() - S.area.sum()) <= 1e-9 * S.area.sum()
    assert abs(e["Mtot"].sum() - S.mass.sum()) <= 1e-9 * S.mass.sum()
    assert np.allclose(e["c"], 1.0, rtol=1e-9) and np.allclose(e["h"], 0.25)
    # mass-weighted means: a field that is the same on every floe comes back as that value in every cell
    T = S.copy(); T.Ui[:] = 0.3; T.Stress[:] = [2.0, 0.5, 0.5, -1.0]
    e = oracle.eulerian_data(T, 8, 8, 0, P.c2_boundary, False)
    assert np.allclose(e["u"], 0.3, rtol=1e-12) and np.allclose(e["stressyx"], 0.5, rtol=1e-12)
    lam = 0.5 + np.sqrt(1.5 ** 2 + 0.25)
    assert np.allclose(e["stress"], lam, rtol=1e-12)
    # after some steps: still consistent among themselves
    A = S.copy()
    for it in range(6):
        P.doInt = int(it == 0)
        oracle.step(A, P, threads=4)
    e = oracle.eulerian_data(A, 8, 8, 0, P.c2_boundary, False)
    assert np.allclose(e["c"], e["area"] / cell, rtol=1e-12) and abs(e["area"].sum() - A.area.sum()) < 1e-3 * A.area.sum()
    assert np.abs(e["u"]).max() > 0 and np.isfinite(e["stress"]).all() and np.abs(e["stress"]).max() > 0


def test_topography_is_cut_out_of_the_cells():
    """Nb > 0 (calc_eulerian_data.m:10-24,141-143) on a hand-computed case: one island, one floe beside it, one floe
    leaning over it; then the tiling pack with its first floes declared topography."""
    if g.ref_lib() is None:
        pytest.skip("the topography path clips operands of several contours: reference build only")
    from subzero_amd import synth

    def box(cx, cy, h):
        return np.array([[cx - h, cy - h], [cx - h, cy + h], [cx + h, cy + h], [cx + h, cy - h]], float)
    L = 1e5
    polys = [box(-5e4, 5e4, 1e4),            # the island: 4e8 m2, inside the upper-left cell of a 2 x 2 grid
             box(5e4, 5e4, 2e4),             # a floe in the upper-right cell, clear of the island
             box(-3.5e4, 5e4, 1e4)]          # a floe overlapping the island's right half by 5e3 x 2e4
    S = synth.build_state(polys, height=1.0, seed=1, n_stress=2, n_mc=50)
    P = synth.StepParams(dt=10, periodic=False)
    e = oracle.eulerian_data(S, 2, 2, 1, P.c2_boundary, False)
    cell = L * L
    # (rows count from the top: element [0, 0] is the upper-left cell)
    assert abs(e["c"][0, 1] - 16e8 / cell) < 1e-12                                # untouched cell: plain rectangle
    left_free = cell - 4e8                                                         # the island is cut out of its cell
    assert abs(e["c"][0, 0] - (4e8 - 1e8) / left_free) < 1e-9                      # and out of the floe leaning over it
    assert abs(e["area"][0, 0] - 3e8) < 1e-3 and e["c"][1, 0] == 0 and e["c"][1, 1] == 0
    assert abs(e["Mtot"][0, 0] - S.mass[2] * 3e8 / 4e8) < 1e-3 * S.mass[2]
    # with Nb = 0 the island is one more floe
    e0 = oracle.eulerian_data(S, 2, 2, 0, P.c2_boundary, False)
    assert abs(e0["c"][0, 0] - 8e8 / cell) < 1e-12
    # the tiling pack: whatever the topography covers is missing from ice area and from the cells' free area alike
    S, P = scenes.pack(300, periodic=False)
    nb = 5
    e = oracle.eulerian_data(S, 6, 6, nb, P.c2_boundary, False)
    assert abs(e["area"].sum() - S.area[nb:].sum()) <= 1e-7 * S.area.sum()
    assert np.nanmax(np.abs(e["c"] - 1.0)) < 1e-6
