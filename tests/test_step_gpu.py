"""GPU parity tests of the B2 surface (szFloesUpload / szStep / ...), through the C ABI, against the oracle.

Bar (BASELINE.json north_star): candidate pairs and contact-row partner ids / counts bit-exact; forces,
torques and x / y / theta within 1e-10 relative after N steps.  In fact every field is expected to be
bit-identical (same IEEE operations in the same order on both sides) except mean(StressH,3), which the device
carries as the exact sum of the history (hoststep.check_stress: 1e-10 relative to the value plus the reference's own
summation rounding); the tests assert the 1e-10 bar and report exactness separately."""
import numpy as np
import pytest

import hoststep
import oracle
import scenes
from subzero_amd import capi

pytestmark = pytest.mark.gpu

RTOL = 1e-10


def lockstep(ctx, S, P, steps, do_int_every=10):
    A = S.copy()            # oracle side
    G = S.copy()            # GPU side (host image)
    sim = capi.FloeInteractions(ctx, G)
    exact = True
    for it in range(steps):
        P.doInt = int(it % do_int_every == 0)
        ra = oracle.step(A, P, threads=8)
        sim.step(P)
        pairs, n_ext = sim.pairs()
        off, rows = sim.interactions()
        sim.download(capi.SYNC_ALL)
        assert n_ext == ra.n_ext, "step %d: N_ext %d vs %d" % (it, n_ext, ra.n_ext)
        assert np.array_equal(pairs, ra.pairs), "step %d: candidate pairs differ" % it
        assert np.array_equal(off, ra.inter_off), "step %d: contact-row counts differ" % it
        assert rows.shape == ra.inter.shape
        if len(rows):
            assert np.array_equal(rows[:, 0], ra.inter[:, 0]), "step %d: partner ids differ" % it
            a, b = rows[:, 1:], ra.inter[:, 1:]   # column 0 is the partner id (Inf for the wall)
            scale = np.maximum(np.abs(a), np.abs(b)).max(0)
            err = (np.abs(a - b).max(0) / np.where(scale > 0, scale, 1)).max()
            assert err <= RTOL, "step %d: contact rows off by %.3e" % (it, err)
            exact &= np.array_equal(rows, ra.inter)
        bad = hoststep.compare_states(A, G, exact=False, rtol=RTOL)
        bad = [b for b in bad if b[0] != "StressH"]
        assert bad == [], "step %d: %s" % (it, bad)
        exact &= (hoststep.compare_states(A, G, exact=True) == [])
        # the stress history itself and its mean (the exact sum of the history on the device, stress_sum_*)
        hoststep.check_stress(A, G, "step %d" % it)
    return A, G, exact


def test_periodic_pack(ctx):
    S, P = scenes.pack(400, periodic=True)
    A, G, exact = lockstep(ctx, S, P, 25)
    assert exact, "expected bit-identical state"


def test_walls_pack(ctx):
    S, P = scenes.pack(300, periodic=False, seed=7, vel_seed=8, sigma=0.3)
    A, G, exact = lockstep(ctx, S, P, 25)
    assert exact


def test_concave_floes(ctx):
    S, P = scenes.concave(60, periodic=True)
    A, G, exact = lockstep(ctx, S, P, 15)
    assert exact


def test_conservation_scenarios(ctx):
    # the geometries of the reference's test/conservation_test.m; its assertion is K(end)/K(1) < 1
    for S, P in (scenes.two_boxes(0.0), scenes.two_boxes(4e3), scenes.boxes_and_triangle()):
        k0 = scenes.kinetic_energy(S)
        A, G, exact = lockstep(ctx, S, P, 40)
        assert scenes.kinetic_energy(G) / k0 < 1
        assert exact


def test_reference_ocean_forcing(ctx):
    S, P = scenes.pack(120, periodic=False, seed=21, sigma=0.05)
    P.set_reference_ocean()
    A, G, exact = lockstep(ctx, S, P, 12, do_int_every=3)
    assert exact


def test_thousand_floes_collisions_count(ctx):
    S, P = scenes.pack(1000, periodic=True, n_stress=4)
    A, G, exact = lockstep(ctx, S, P, 6)
    assert exact


def test_two_stage_lds_path_matches_oracle(ctx, monkeypatch):
    # opt-in path: compact sweep records in LDS, first clip / contact stage as two kernels (SUBZERO_FAST_PATH=1)
    monkeypatch.setenv("SUBZERO_FAST_PATH", "1")
    S, P = scenes.pack(400, periodic=True)
    A, G, exact = lockstep(ctx, S, P, 12)
    assert exact
    S, P = scenes.concave(40, periodic=True)      # rings too large for the compact tier: general path
    A, G, exact = lockstep(ctx, S, P, 6)
    assert exact


def test_general_sweep_for_every_pair_matches_oracle(ctx, monkeypatch):
    # SUBZERO_CONVEX_FAST=0: no fast first pass, the general sweep clips every pair
    monkeypatch.setenv("SUBZERO_CONVEX_FAST", "0")
    S, P = scenes.pack(400, periodic=True)
    A, G, exact = lockstep(ctx, S, P, 12)
    assert exact


@pytest.mark.parametrize("seed", [101, 202, 303, 404, 505, 606])
def test_more_fields_through_the_fast_pass(ctx, seed):
    # different Voronoi fields, speeds and boundary types: every one bit-identical to the oracle
    S, P = scenes.pack(260 + seed % 7 * 20, periodic=bool(seed % 2), seed=seed, vel_seed=seed + 1, sigma=0.1 + 0.05 * (seed % 5))
    A, G, exact = lockstep(ctx, S, P, 9)
    assert exact


@pytest.mark.parametrize("name,steps", [("plumbing10", 30), ("convex1000", 8), ("uniaxial200", 15), ("nares_standin", 10)])
def test_baseline_configs(ctx, name, steps):
    # BASELINE.json configs 0-2 and the configs[4] stand-in at test size; the uniaxial case moves its N/S walls between steps
    S, P = scenes.baseline_config(name)
    A, G = S.copy(), S.copy()
    sim = capi.FloeInteractions(ctx, G)
    for it in range(steps):
        P.doInt = int(it % 10 == 0)
        if name == "uniaxial200":
            scenes.squeeze_walls(P, it)
        ra = oracle.step(A, P, threads=16)
        sim.step(P)
        sim.download(capi.SYNC_ALL)
        off, rows = sim.interactions()
        assert np.array_equal(ra.inter_off, off) and np.array_equal(ra.inter, rows), "step %d" % it
        assert hoststep.compare_states(A, G) == [], "step %d" % it
        hoststep.check_stress(A, G, "step %d" % it)


def test_forcing_with_more_points_than_the_kernel_keeps(ctx):
    # k_forcing keeps the samples of 4 points per thread between its two passes (n_mc <= 1024); 1300 points per
    # floe take the path that samples the rest again
    S, P = scenes.pack(60, periodic=True, seed=9, vel_seed=10, sigma=0.2, n_mc=1300)
    P.set_reference_ocean()
    A, G, exact = lockstep(ctx, S, P, 6, do_int_every=2)
    assert exact


@pytest.mark.parametrize("seed", list(range(3000, 3030)) + [50239, 50249, 50599, 53828])   # (+ ghosts of boundary floes as partners)
def test_random_scenes(ctx, seed):
    # random small scenes (Voronoi packs, concave outlines, many-sided floes; walls or periodic; Nb, forcing)
    S, P, steps = scenes.random_scene(seed)
    A, G, exact = lockstep(ctx, S, P, steps, do_int_every=3)
    assert exact


def test_deferred_pairs_with_hbm_workspace(ctx, monkeypatch):
    # SUBZERO_RETRY_LDS=0: the general pass over deferred pairs keeps its sweep workspace in HBM (the default puts
    # it into LDS when the list is short); walls make every step defer the floes along them
    monkeypatch.setenv("SUBZERO_RETRY_LDS", "0")
    S, P = scenes.pack(300, periodic=False, seed=7, vel_seed=8, sigma=0.3)
    A, G, exact = lockstep(ctx, S, P, 10)
    assert exact


def test_evolved_outlines_at_scale(ctx):
    # 600 outlines of the reference's FloeShapes.mat (up to 90 vertices): every pair goes to the general kernel, which
    # trims the rings to their common box, keeps rings, crossings layout and sweep workspace in 64 KB slices of LDS
    S, P = scenes.concave(600, periodic=True)
    A, G, exact = lockstep(ctx, S, P, 5)
    assert exact


@pytest.mark.parametrize("slice_bytes", [8192, 24576, 32768])
def test_evolved_outlines_with_small_lds_slices(ctx, monkeypatch, slice_bytes):
    # 8 KB slices hold the rings only and sweeps of a few vertices; 24 KB also the small arrays of short pairs;
    # 32 KB those of every pair, and sweeps of up to ~36 trimmed vertices -- the rest falls back to the pair's HBM
    # slice array by array, clip by clip
    monkeypatch.setenv("SUBZERO_RETRY_LDS", str(slice_bytes))
    monkeypatch.setenv("SUBZERO_RETRY_LDS_MAX", str(slice_bytes))
    S, P = scenes.concave(150, periodic=False)
    A, G, exact = lockstep(ctx, S, P, 6)
    assert exact


def test_long_convex_rings(ctx):
    # 8..39-vertex convex floes: fast pass (16-vertex blocks) and general sweep for the longer rings
    S, P = scenes.many_sided(64)
    A, G, exact = lockstep(ctx, S, P, 12)
    assert exact


@pytest.mark.parametrize("periodic", [False, True])
@pytest.mark.parametrize("nb,collision", [(0, 1), (2, 1), (0, 0)])
def test_edge_cases(ctx, periodic, nb, collision):
    # swallowed floe (kill / transfer marks), floe outside the wall, dead floe, NaN floe, Nb > 0, COLLISION off
    S, P = scenes.edge_cases(periodic)
    S.nb = nb; P.COLLISION = collision
    A, G, exact = lockstep(ctx, S, P, 4)
    assert exact
    assert np.array_equal(A.kill, G.kill) and np.array_equal(A.transfer, G.transfer)
    if nb == 0 and collision:
        assert G.kill[0] == 2 and G.transfer[1] == 1


def test_back_half_queued_behind_the_narrow_phase_falls_back_when_pairs_are_deferred(ctx):
    """Steady steps queue the whole step without a host round trip; a step whose narrow phase defers pairs (here: two
    floes made to coincide exactly, every vertex a tie) must notice on the device, do nothing after the narrow phase,
    and be finished the ordinary way -- same results either way, and the same again with the queueing switched off."""
    for env in (None, "0"):
        if env is not None:
            import os
            os.environ["SUBZERO_SPECULATE_BACK"] = env
        try:
            S, P = scenes.pack(400, periodic=True)
            P.set_reference_ocean()
            A, G = S.copy(), S.copy()
            sim = capi.FloeInteractions(ctx, G)
            retried = []
            for it in range(12):
                P.doInt = int(it % 5 == 0)
                if it in (6, 9):            # floe 12 is put down with its first vertex exactly on floe 11's: a tie the fast pass hands on
                    for T in (A, G):
                        a, b = T.vert_off[11], T.vert_off[12]
                        T.Xi[12] = (T.cax[a] + T.Xi[11]) - T.cax[b]; T.Yi[12] = (T.cay[a] + T.Yi[11]) - T.cay[b]
                    sim.update(np.array([13], np.int32), capi.F_STATE)
                ra = oracle.step(A, P, threads=8)
                st = sim.step(P)
                retried.append(int(st.retried))
                off, rows = sim.interactions()
                sim.download(capi.SYNC_ALL)
                assert np.array_equal(off, ra.inter_off) and np.array_equal(rows, ra.inter), it
                assert [b for b in hoststep.compare_states(A, G) if b[0] != "StressH"] == [], it
            assert sum(1 for r in retried if r == 0) >= 6 and max(retried) > 0, retried       # steady steps, and steps that deferred pairs
        finally:
            if env is not None:
                os.environ.pop("SUBZERO_SPECULATE_BACK", None)


@pytest.mark.parametrize("early", ["1", "0"])
def test_wall_pairs_beside_the_team_pass(ctx, monkeypatch, early):
    """A field between walls large enough for the team pass to move to the side stream (step.hip step_narrow_first: the
    wall pairs' general sweeps are launched first, on the step's stream): every step against the oracle, with the wall
    list of its own and -- SUBZERO_WALLS_EARLY=0 -- with the wall pairs on the retry list behind the read-back."""
    monkeypatch.setenv("SUBZERO_WALLS_EARLY", early)
    S, P = scenes.pack(2500, periodic=False, seed=17, vel_seed=18, sigma=0.2, n_stress=4, n_mc=120)
    A, G, exact = lockstep(ctx, S, P, 8)
    assert exact


def test_stage_times_lag_one_step_and_szStepTimes_is_exact(ctx):
    """szStepStats: the HIP-event stage times a step returns are those of the step before (read while the host waits
    anyway); szStepTimes waits for the last step's events."""
    S, P = scenes.pack(300, periodic=True)
    sim = capi.FloeInteractions(ctx, S)
    seen = []
    for it in range(4):
        P.doInt = int(it == 0)
        st = sim.step(P)
        seen.append((int(st.times_step), st.step_ms, st.narrow_ms))
    assert [s[0] for s in seen] == [-1, 0, 1, 2] and seen[0][1] == 0
    assert all(s[1] > 0 for s in seen[1:])
    last = sim.step_times()
    assert int(last.times_step) == 3 and last.step_ms > 0 and last.step_ms >= last.narrow_ms >= 0
