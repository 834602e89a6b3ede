"""CPU suite for the step path (no GPU):
  * oracle pins: MATLAB's own cached area / centroid of the reference fixture shapes, known answers for InterX,
    inpolygon, p_poly_dist; self-contained sin/cos/atan2 against libm;
  * the reference's conservation scenarios (test/conservation_test.m:22-54): kinetic energy must not grow;
  * the device pipeline (subzero_amd/csrc/step_core.hpp compiled for the host, tests/hostsim) is
    bit-identical to the oracle, step after step, in every scene family.
"""
import ctypes
import numpy as np
import pytest

import clipgen
import hoststep
import oracle
import scenes
from subzero_amd import synth


def test_area_centroid_against_matlab_polyshape_cache():
    # FloeShapes.mat stores, per polyshape, MATLAB's own [?, area, perimeter, cx, cy, xmin, ymin, xmax, ymax]
    shapes, info = clipgen.load_shapes()
    for s, pi in zip(shapes, info):
        a = oracle.ring_area(s[:, 0], s[:, 1])
        cx, cy = oracle.ring_centroid(s[:, 0], s[:, 1])
        assert abs(a - pi[1]) <= 1e-12 * pi[1]
        scale = np.abs(s).max()
        assert abs(cx - pi[3]) <= 1e-12 * scale and abs(cy - pi[4]) <= 1e-12 * scale


def test_interx_known_answers():
    sq = np.array([[0, 0], [2, 0], [2, 2], [0, 2], [0, 0]], float)
    px, py = oracle.interx(sq[:, 0], sq[:, 1], sq[:, 0] + 1, sq[:, 1] + 1)
    assert np.array_equal(np.stack([px, py], 1), np.array([[1.0, 2.0], [2.0, 1.0]]))   # sorted rows, InterX.m:76
    px, py = oracle.interx(sq[:, 0], sq[:, 1], sq[:, 0] + 5, sq[:, 1])
    assert len(px) == 0


def test_inpolygon_and_poly_dist():
    L = oracle.lib()
    tri = np.array([[0, 0], [4, 0], [0, 4], [0, 0]], float)
    pts = np.array([[1, 1], [3, 3], [2, 2], [0, 0], [-1, 0], [2, 0]], float)   # in, out, on edge, vertex, out, on edge
    xs, ys = np.ascontiguousarray(pts[:, 0]), np.ascontiguousarray(pts[:, 1])
    vx, vy = np.ascontiguousarray(tri[:, 0]), np.ascontiguousarray(tri[:, 1])
    out = np.zeros(len(pts), np.uint8)
    L.or_inpolygon(oracle._p(xs), oracle._p(ys), len(pts), oracle._p(vx), oracle._p(vy), len(tri), oracle._p(out, oracle._bp))
    assert out.tolist() == [1, 0, 1, 1, 0, 1]
    d = np.zeros(len(pts))
    L.or_p_poly_dist(oracle._p(xs), oracle._p(ys), len(pts), oracle._p(vx), oracle._p(vy), len(tri), oracle._p(d))
    assert np.allclose(np.abs(d), [1.0, np.sqrt(2), 0.0, 0.0, 1.0, 0.0], atol=1e-12)


def test_selfcontained_trig_within_2ulp():
    L = oracle.lib()
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-7, 7, 4000), rng.uniform(-1e4, 1e4, 4000), [0.0, np.pi / 2, np.pi, -np.pi / 4]])
    s, c = ctypes.c_double(), ctypes.c_double()
    for x in xs:
        L.or_sincos(float(x), ctypes.byref(s), ctypes.byref(c))
        assert abs(s.value - np.sin(x)) <= 2 * np.spacing(abs(np.sin(x))) + 1e-300
        assert abs(c.value - np.cos(x)) <= 2 * np.spacing(abs(np.cos(x))) + 1e-300
    for y, x in rng.uniform(-1e4, 1e4, (4000, 2)):
        a = L.or_atan2(float(y), float(x))
        assert abs(a - np.arctan2(y, x)) <= 3 * np.spacing(abs(np.arctan2(y, x)))


def lockstep(S, P, steps, do_int_every=10, pot_cap=32):
    A, B = S.copy(), S.copy()
    for it in range(steps):
        P.doInt = int(it % do_int_every == 0)
        ra = oracle.step(A, P, threads=4)
        rb = hoststep.step(B, P, pot_cap=pot_cap)
        assert ra.n_ext == rb.n_ext
        assert np.array_equal(ra.pairs, rb.pairs), "step %d: candidate pairs" % it
        assert np.array_equal(ra.inter_off, rb.inter_off), "step %d: row counts" % it
        assert np.array_equal(ra.inter, rb.inter), "step %d: contact rows" % it
        bad = [b for b in hoststep.compare_states(A, B) if b[0] != "StressH"]
        assert bad == [], "step %d: %s" % (it, bad)
        hoststep.check_stress(A, B, "step %d" % it)
    return A, B, ra


@pytest.mark.parametrize("binned", [True, False])
def test_device_pipeline_matches_oracle_periodic(binned):
    S, P = scenes.pack(300, periodic=True)
    A, B, r = lockstep(S, P, 14, pot_cap=32 if binned else -32)
    assert r.n_ext > S.n and len(r.inter) > 0          # ghosts and contacts were actually exercised


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_every_narrow_phase_variant_matches_oracle(mode):
    """General sweep only / two-stage compact path / single-minimum fast path, a lane per pair (convex.hpp) / the team
    form first: pairs sorted, 16 or 32 lanes per pair (teamcontact.hpp, the GPU default) / 32 lanes for all, unsorted."""
    hoststep.set_narrow_mode(mode)
    try:
        h0, b0 = hoststep.fast_counts()
        S, P = scenes.pack(250, periodic=True, seed=11, vel_seed=12, sigma=0.25)
        lockstep(S, P, 8)
        h1, b1 = hoststep.fast_counts()
        if mode == 2:   # step 0 has every neighbour pair sharing an edge exactly (all bail); later steps do not
            assert h1 - h0 > 1000 and b1 - b0 > 0, (h1 - h0, b1 - b0)
        elif mode < 2:
            assert h1 == h0 and b1 == b0
    finally:
        hoststep.set_narrow_mode(3)


def test_long_convex_rings_are_handed_to_the_general_sweep():
    S, P = scenes.many_sided(64)
    h0, b0 = hoststep.fast_counts()
    A, B, r = lockstep(S, P, 6)
    h1, b1 = hoststep.fast_counts()
    # the fast path ran, and it handed pairs on (rings beyond its 16-vertex blocks)
    assert len(r.inter) > 0 and h1 - h0 > 50 and hoststep.step(B.copy(), P).retried > 10


@pytest.mark.parametrize("name,steps", [("plumbing10", 30), ("convex1000", 5), ("uniaxial200", 12), ("nares_standin", 8)])
def test_baseline_configs_match_oracle(name, steps):
    S, P = scenes.baseline_config(name)
    A, B = S.copy(), S.copy()
    for it in range(steps):
        P.doInt = int(it % 10 == 0)
        if name == "uniaxial200":
            scenes.squeeze_walls(P, it)
        ra = oracle.step(A, P, threads=4)
        rb = hoststep.step(B, P)
        assert np.array_equal(ra.pairs, rb.pairs) and np.array_equal(ra.inter, rb.inter), "step %d" % it
        assert [b for b in hoststep.compare_states(A, B) if b[0] != "StressH"] == [], "step %d" % it
        hoststep.check_stress(A, B, "step %d" % it)
    if name == "uniaxial200":
        assert (ra.inter[:, 0] == np.inf).any()        # the moving walls push on the pack


# (50239 ... 53828: scenes of the round-2 GPU campaign in which the ghost of a boundary floe -- one of the Nb floes,
# crossing the periodic edge -- is the partner of a real floe: entries beyond N0 are candidates whatever their source)
@pytest.mark.parametrize("seed", list(range(2000, 2024)) + [50239, 50249, 50599, 53828])
def test_random_scenes_match_oracle(seed):
    S, P, steps = scenes.random_scene(seed)
    A, B = S.copy(), S.copy()
    for it in range(steps):
        P.doInt = int(it % 3 == 0)
        ra = oracle.step(A, P, threads=4)
        rb = hoststep.step(B, P)
        assert np.array_equal(ra.pairs, rb.pairs) and np.array_equal(ra.inter, rb.inter), "step %d" % it
        assert [b for b in hoststep.compare_states(A, B) if b[0] != "StressH"] == [], "step %d" % it
        hoststep.check_stress(A, B, "step %d" % it)


def test_stress_mean_keeps_no_drift_over_wraps_of_the_history():
    """mean(StressH,3) is carried as an exact (two-double) sum across steps -- device-only state that lives from upload
    to upload.  Several wraps of a short history, sums kept between the calls as on the GPU."""
    S, P = scenes.pack(150, periodic=True, n_stress=6)
    A, B = S.copy(), S.copy()
    hoststep.keep_stress_sums(True)
    try:
        for it in range(40):
            P.doInt = int(it % 10 == 0)
            oracle.step(A, P, threads=4)
            hoststep.step(B, P)
            assert [b for b in hoststep.compare_states(A, B) if b[0] != "StressH"] == [], "step %d" % it
            hoststep.check_stress(A, B, "step %d" % it)
    finally:
        hoststep.keep_stress_sums(False)
    assert (A.StressCount > 6).any() or (A.StressCount < 10).all()


def test_stress_sum_pair_is_the_exact_sum_of_the_window():
    """step_core.hpp stress_sum_init / stress_sum_replace on a stream of values twelve orders of magnitude apart: the
    high word is the exactly rounded sum of the current window after every replacement (math.fsum), however many
    large entries have rotated through; the plain running sum `s - old + new` it replaces is not."""
    import math
    rng = np.random.default_rng(11)
    nslots, nsteps = 16, 5000
    vals = rng.normal(0, 1.0, nslots + nsteps) * 10.0 ** rng.integers(-3, 10, nslots + nsteps)
    hi = hoststep.stress_sum_trace(vals, nslots)
    window = list(vals[:nslots])
    plain = float(np.sum(vals[:nslots]))
    worst_plain = 0.0
    for k in range(nsteps + 1):
        exact = math.fsum(window)
        assert hi[k] == exact, (k, hi[k], exact)
        worst_plain = max(worst_plain, abs(plain - exact))
        if k < nsteps:
            slot = k % nslots
            plain = plain - window[slot] + vals[nslots + k]
            window[slot] = vals[nslots + k]
    assert worst_plain > 1e-6    # the residue of the rotated-out 1e9-sized entries the exact pair does not have


def test_device_pipeline_matches_oracle_walls():
    S, P = scenes.pack(200, periodic=False, seed=7, vel_seed=8, sigma=0.3)
    A, B, r = lockstep(S, P, 14)
    assert (r.inter[:, 0] == np.inf).any()             # floe-vs-boundary rows present


def test_device_pipeline_matches_oracle_concave():
    S, P = scenes.concave(40, periodic=True)
    lockstep(S, P, 8)


def test_device_pipeline_matches_oracle_with_eddy_forcing():
    S, P = scenes.pack(80, periodic=False, seed=21, sigma=0.05)
    P.set_reference_ocean()
    A, B, r = lockstep(S, P, 7, do_int_every=3)
    assert np.abs(A.FxOA).max() > 0


def test_conservation_scenarios_energy_does_not_grow():
    # test/conservation_test.m:22-54 asserts only K(end)/K(1) < 1
    for S, P in (scenes.two_boxes(0.0), scenes.two_boxes(4e3), scenes.boxes_and_triangle()):
        k0 = scenes.kinetic_energy(S)
        A, B, r = lockstep(S, P, 40)
        assert scenes.kinetic_energy(A) / k0 < 1
        assert A.collision_force.any() or len(r.inter) >= 0


@pytest.mark.parametrize("periodic", [False, True])
@pytest.mark.parametrize("nb,collision", [(0, 1), (2, 1), (0, 0)])
def test_edge_cases_match_oracle(periodic, nb, collision):
    # swallowed floe (kill / transfer marks), floe outside the wall, dead floe, NaN floe, Nb > 0, COLLISION off
    S, P = scenes.edge_cases(periodic)
    S.nb = nb; P.COLLISION = collision
    A, B, r = lockstep(S, P, 4)
    if nb == 0 and collision:
        assert A.kill[0] == 2 and A.transfer[1] == 1        # floe 2 merges into floe 1 (floe_interactions_all.m:138-145,175-179)
    if periodic:
        assert A.kill[5] == 6                                # NaN position: removed by calc_trajectory (:282)
    else:
        assert A.alive[3] == 0 and A.alive[5] == 0           # centre outside the box (:152-155)


# ---- the long-ring forms of InterX and of the p_poly_dist test (contact.hpp) against the plain ones ----------------

_dp = ctypes.POINTER(ctypes.c_double)


def _d(v):
    return np.ascontiguousarray(v, np.float64).ctypes.data_as(_dp)


def test_crossings_of_long_rings_match_the_dense_form():
    """crossings_long skips runs of edges that InterX's first sign test is certain to reject: same points, same order."""
    L = hoststep.lib()
    ps = clipgen.shape_pairs(1500, seed=5)
    k0, s0 = ctypes.c_longlong(), ctypes.c_longlong()
    L.host_prune_counts(ctypes.byref(k0), ctypes.byref(s0))
    found = 0
    for p in range(len(ps)):
        a = np.stack([ps.ax[p], ps.ay[p]], 1) / 2.0 ** 32
        b = np.stack([ps.bx[p], ps.by[p]], 1) / 2.0 ** 32
        if p % 5 == 0:                       # exact ties too: ring 2 shares vertices and edges with ring 1
            b = np.vstack([a[: len(a) // 2], b[len(b) // 2:], a[:1]])
        ax, ay, bx, by = (np.ascontiguousarray(v) for v in (a[:, 0], a[:, 1], b[:, 0], b[:, 1]))
        out = []
        for pruned in (0, 1):
            px, py = np.zeros(4096), np.zeros(4096)
            n = L.host_crossings(_d(ax), _d(ay), len(ax), _d(bx), _d(by), len(bx), pruned, _d(px), _d(py), 4096)
            out.append((n, px[:max(n, 0)].copy(), py[:max(n, 0)].copy()))
        assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2]), p
        found += out[0][0]
    k1, s1 = ctypes.c_longlong(), ctypes.c_longlong()
    L.host_prune_counts(ctypes.byref(k1), ctypes.byref(s1))
    assert found > 1000 and (s1.value - s0.value) > 5 * (k1.value - k0.value)      # most runs are skipped


def test_distance_test_from_near_elements_matches_the_full_distance():
    """closer_than_1e8 == (|p_poly_dist| < 1e-8), on and around the outline, at the 1e-8 and the 1e-3 m thresholds."""
    L = hoststep.lib()
    L.host_closer.argtypes = [ctypes.c_double, ctypes.c_double, _dp, _dp, ctypes.c_int, ctypes.c_int]
    shapes, _ = clipgen.load_shapes()
    rng = np.random.default_rng(3)
    n_true = n_all = 0
    for trial in range(1500):
        s = shapes[rng.integers(len(shapes))] + rng.uniform(-9e4, 9e4, 2)
        c = np.vstack([s, s[:1]])
        x, y = np.ascontiguousarray(c[:, 0]), np.ascontiguousarray(c[:, 1])
        for _ in range(12):
            k = rng.integers(len(s))
            t = rng.choice([0.0, 1.0, 0.5, rng.uniform(0, 1), 1e-9, 1 - 1e-9])
            a, b = c[k], c[k + 1]
            nrm = np.array([-(b - a)[1], (b - a)[0]]); nrm /= np.linalg.norm(nrm) + 1e-300
            off = rng.choice([0.0, 0.0, 5e-9, -5e-9, 1.00001e-8, 0.99999e-8, 2e-8, 4e-4, 6e-4, 9.9e-4, 1.01e-3, 1.0, -1.0, 100.0])
            q = a + t * (b - a) + off * nrm
            r = [L.host_closer(float(q[0]), float(q[1]), _d(x), _d(y), len(x), pruned) for pruned in (0, 1)]
            assert r[0] == r[1], (trial, q.tolist(), r)
            n_all += 1; n_true += r[0]
    assert 0.2 * n_all < n_true < 0.8 * n_all


@pytest.mark.parametrize("periodic", [False, True])
def test_one_floe_twenty_times_the_median_grows_the_lists(periodic):
    """One floe 20 times the median radius among 1500 small ones: more than 32 candidates and more than 32 mirrored
    contacts for one floe, a second grid level for it alone.  The lists grow (no SZ_ERR_WORKSPACE), nothing is lost:
    candidate pairs, contact rows and the state are those of the reference's all-pairs loop."""
    S, P, k = scenes.big_among_small(1500, 20.0, periodic=periodic, centre=(5.5e4, 1e4) if periodic else (0.0, 0.0))
    assert S.rmax[k] > 15 * np.median(S.rmax)
    A, B = S.copy(), S.copy()
    for it in range(3):
        P.doInt = int(it == 0)
        ra = oracle.step(A, P, threads=8)
        rb = hoststep.step(B, P)
        grown, cap, levels = hoststep.last_front()
        assert grown >= 1 and cap >= 64 and levels >= 2, (grown, cap, levels)
        assert ra.n_ext == rb.n_ext and np.array_equal(ra.pairs, rb.pairs), "step %d: candidate pairs" % it
        assert np.array_equal(ra.inter_off, rb.inter_off) and np.array_equal(ra.inter, rb.inter), "step %d" % it
        assert [b for b in hoststep.compare_states(A, B) if b[0] != "StressH"] == [], "step %d" % it
    own = (ra.pairs[:, 0] == k + 1).sum(); mirrored = (ra.pairs[:, 1] == k + 1).sum()
    assert own > 32 and mirrored > 32, (own, mirrored)
    rows_big = ra.inter_off[k + 1] - ra.inter_off[k]
    assert rows_big > 32, rows_big


def test_empty_monte_carlo_mask_is_reported_before_anything_changes():
    """calc_trajectory.m:100-110 draws new Monte-Carlo points with rand when a floe's mask is empty and its forcing is
    due.  The step does not guess them and does not fail half-way: it lists the floes and leaves the array untouched;
    after the caller has regenerated them the same step goes through and matches the oracle."""
    S, P = scenes.pack(80, periodic=True, n_stress=4, n_mc=200)
    P.set_reference_ocean()
    S.mcA[[5, 41]] = 0                      # two floes without a single point inside
    S.mcA[17] = 0; S.alive[17] = 0          # a dead one is never integrated: not reported
    B = S.copy()
    P.doInt = 0
    hoststep.step(B, P)                     # no forcing due (doInt off, h = 0.25 m): the masks are not looked at
    A = S.copy(); oracle.step(A, P, threads=2)
    assert hoststep.compare_states(A, B) == []
    before = B.copy()
    P.doInt = 1
    with pytest.raises(hoststep.NeedsMc) as e:
        hoststep.step(B, P)
    assert e.value.ids.tolist() == [6, 42]
    assert hoststep.compare_states(before, B) == [] and np.array_equal(before.Stress, B.Stress)
    # the caller's part (MATLAB: calc_trajectory.m:102-109 with its own rand; here any generator, the same on both sides)
    rng = np.random.default_rng(5)
    for i in (5, 41):
        a, b = S.vert_off[i], S.vert_off[i + 1]
        X = B.rmax[i] * (2 * rng.random(B.n_mc) - 1); Y = B.rmax[i] * (2 * rng.random(B.n_mc) - 1)
        inside = synth._inpolygon(X, Y, B.cax[a:b], B.cay[a:b])
        for T in (A, B):
            T.mcX[i], T.mcY[i], T.mcA[i] = X, Y, inside
    ra = oracle.step(A, P, threads=2)
    rb = hoststep.step(B, P)
    assert np.array_equal(ra.inter, rb.inter) and hoststep.compare_states(A, B) == []
    # a thin floe is due for forcing on every step (h < 0.1 after the thinning, :94)
    B.mcA[7] = 0; B.h[7] = 0.05
    P.doInt = 0
    with pytest.raises(hoststep.NeedsMc) as e:
        hoststep.step(B, P)
    assert e.value.ids.tolist() == [8]
