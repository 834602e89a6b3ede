"""The oracle's restatement of the fracture deformation step (fracture_floe.m:14-52): a hand-computed
case and the invariants of the construction on a colliding field."""
import numpy as np

import oracle
import scenes
from subzero_amd import synth


def _two_squares(gap):
    a = np.array([[0, 0], [0, 10], [10, 10], [10, 0]], float) * 1000
    b = a + np.array([gap, 0.0])
    return synth.build_state([a, b], height=1.0, seed=1, n_stress=2, n_mc=50)


def test_hand_computed_deformation():
    # two 10 km squares overlapping in a 200 m strip; the contact force on floe 1 points in -x
    S = _two_squares(9800.0)
    inter_off = np.array([0, 2, 3], np.int32)
    inter = np.array([[np.inf, 5.0, 0.0, 0, 0, 0, 9e9],          # a boundary row: dropped (:19-21), however large
                      [2, -3.0, 0.0, 9900.0, 5000.0, 0.0, 2.0e6],
                      [1, 3.0, 0.0, 9900.0, 5000.0, 0.0, 2.0e6]])
    ch, X, Y, A, rings = oracle.fracture_deform(S, inter_off, inter, np.array([1, 2], np.int32))
    assert list(ch) == [1, 1]
    # floe 1: region [9800, 10000] x [0, 10000], centroid 100 m from its outline -> partner moved by -50 m -> cut at 9750
    assert abs(A[0] - 9750.0 * 10000.0) < 1e-3 and abs(X[0] - 4875.0) < 1e-6 and abs(Y[0] - 5000.0) < 1e-6
    r = rings[0]
    assert r.shape[0] == 2 and abs(r[0].max() - 4875.0) < 1e-6 and abs(r[0].min() + 4875.0) < 1e-6
    assert abs(np.abs(r[1]).max() - 5000.0) < 1e-6
    # floe 2 mirrors it: partner (floe 1) moved by +50 m -> floe 2 starts at 10050
    assert abs(A[1] - 9750.0 * 10000.0) < 1e-3 and abs(X[1] - (10050.0 + 19800.0) / 2) < 1e-6
    # a partner number beyond the array (a ghost) leaves the floe alone (:26)
    inter2 = inter.copy(); inter2[1, 0] = 3
    ch, *_ = oracle.fracture_deform(S, inter_off, inter2, np.array([1], np.int32))
    assert list(ch) == [0]
    # a deep overlap (more than 10 % of the floe would go) leaves it alone too (:44)
    S = _two_squares(7000.0)
    ch, *_ = oracle.fracture_deform(S, inter_off, inter, np.array([1], np.int32))
    assert list(ch) == [0]


def test_invariants_on_a_colliding_field():
    S, P = scenes.pack(300, periodic=True)
    P.set_reference_ocean()
    S.Xi += 300.0
    for it in range(8):
        P.doInt = int(it % 5 == 0)
        r = oracle.step(S, P, threads=8)
    ids = np.arange(1, S.n + 1, dtype=np.int32)
    ch, X, Y, A, rings = oracle.fracture_deform(S, r.inter_off, r.inter, ids)
    rows = np.diff(r.inter_off)
    assert ch[rows == 0].sum() == 0 and ch.sum() > 50
    for s in np.nonzero(ch)[0]:
        ring = rings[s]
        assert abs(oracle.polyarea(ring[0], ring[1]) - A[s]) <= 1e-9 * A[s]
        cx, cy = oracle.ring_centroid(ring[0], ring[1])
        assert abs(cx) < 1e-6 and abs(cy) < 1e-6                      # c_alpha is centred on the new centroid
        assert 0.9 * S.area[s] < A[s] <= S.area[s] * (1 + 1e-12)
        assert np.hypot(X[s] - S.Xi[s], Y[s] - S.Yi[s]) < 0.1 * np.sqrt(S.area[s])
    for s in np.nonzero(ch == 0)[0]:
        assert rings[s].shape[1] == 0 and A[s] == 0
