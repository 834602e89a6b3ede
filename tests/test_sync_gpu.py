"""GPU tests of the synchronisation protocol of the B2 surface, through
the C ABI: download / re-upload keeps Stress and its history, partial update / fetch of field groups, resync after
topology changes with the histories staying on the device, floes that need new Monte-Carlo points reported before
anything changes, the ghost list for the MATLAB tail, candidate lists that grow instead of failing."""
import numpy as np
import pytest

import hoststep
import oracle
import scenes
from subzero_amd import capi, synth

pytestmark = pytest.mark.gpu


def steps(sim, A, P, k, start=0, every=10):
    for it in range(start, start + k):
        P.doInt = int(it % every == 0)
        ra = oracle.step(A, P, threads=8)
        sim.step(P)
    return ra


def check(sim, A, tag=""):
    G = sim.download(capi.SYNC_ALL)
    assert hoststep.compare_states(A, G) == [], tag
    hoststep.check_stress(A, G, tag)
    return G


def test_step_download_reupload_keeps_stress_and_history(ctx):
    # after every MATLAB-side topology change the array is uploaded again.  The history must survive the
    # round trip: Stress = mean(StressH,3) keeps following the reference afterwards (n_stress = 5: the history wraps twice)
    S, P = scenes.pack(300, periodic=True, n_stress=5)
    P.set_reference_ocean()
    A, G = S.copy(), S.copy()
    sim = capi.FloeInteractions(ctx, G)
    steps(sim, A, P, 7)
    check(sim, A, "before the re-upload")
    assert (A.StressCount > 1).all() and np.abs(A.StressH).max() > 0
    H = G.copy()                                   # what a caller holds after the download
    sim2 = capi.FloeInteractions(ctx, H)           # szFloesUpload again: replaces the resident array
    steps(sim2, A, P, 9, start=7)
    off, rows = sim2.interactions()
    check(sim2, A, "after the re-upload")
    assert len(rows) > 0


def test_update_and_fetch_of_field_groups(ctx):
    S, P = scenes.pack(200, periodic=False, n_stress=6, n_mc=150)
    A, G = S.copy(), S.copy()
    sim = capi.FloeInteractions(ctx, G)
    steps(sim, A, P, 3)
    # the caller only has a stale image; it fetches three floes' state, kicks them, and sends them back
    ids = np.array([7, 150, 42], np.int32)
    G.Ui[:] = -99.0; G.StressH[:] = -1.0           # stale garbage everywhere
    sim.fetch(ids, capi.F_STATE | capi.F_STRESSH | capi.F_RING | capi.F_MC)
    for i in ids - 1:
        assert G.Ui[i] == A.Ui[i] and np.array_equal(G.StressH[i], A.StressH[i]) and G.StressCount[i] == A.StressCount[i]
        a, b = S.vert_off[i], S.vert_off[i + 1]
        assert np.array_equal(G.cax[a:b], A.cax[a:b]) and np.array_equal(G.mcX[i], A.mcX[i])
    untouched = np.setdiff1d(np.arange(S.n), ids - 1)
    assert (G.Ui[untouched] == -99.0).all() and (G.StressH[untouched] == -1.0).all()
    for T in (A, G):
        T.Ui[ids - 1] += 0.25; T.h[ids - 1] *= 0.5; T.StressH[ids - 1] *= 0.5      # e.g. frac_corner.m:157 scales the history
    sim.update(ids, capi.F_STATE | capi.F_STRESSH)
    steps(sim, A, P, 4, start=3)
    check(sim, A, "after update")


def test_resync_keeps_device_histories_and_matches_a_fresh_upload(ctx):
    S, P = scenes.pack(260, periodic=True, n_stress=6, n_mc=150)
    P.set_reference_ocean()
    A, G = S.copy(), S.copy()
    sim = capi.FloeInteractions(ctx, G)
    steps(sim, A, P, 5)
    check(sim, A)
    # a MATLAB-side process: floes 10 and 200 are removed, floe 50 is reshaped (one vertex cut off: a shorter ring, new
    # Monte-Carlo points, history scaled), one brand-new floe is appended
    keep = [i for i in range(S.n) if i not in (9, 199)]
    polys = []
    for i in keep:
        x, y = A.c0x[A.vert_off[i]:A.vert_off[i + 1] - 1], A.c0y[A.vert_off[i]:A.vert_off[i + 1] - 1]
        polys.append(np.stack([x, y], 1) if i != 49 else np.stack([x[1:], y[1:]], 1))
    new_poly = np.array([[0.0, 0.0], [0.0, 900.0], [800.0, 700.0], [900.0, 0.0]]) + [A.Xi[9], A.Yi[9]]
    B = synth.build_state(polys + [new_poly], height=0.25, seed=77, n_stress=6, n_mc=150)
    src = np.zeros(B.n, np.int32)
    for k, i in enumerate(keep):
        if i == 49:
            B.StressH[k] = 0.7 * A.StressH[i]          # changed floe: its bulk rows come from the host
            B.StressCount[k] = A.StressCount[i]
            continue
        src[k] = i + 1
        for f in synth.FloeState.SCALARS + ["alive", "StressCount", "Stress", "strain", "collision_force", "collision_torque", "OverlapArea"]:
            getattr(B, f)[k] = getattr(A, f)[i]
        a, b = A.vert_off[i], A.vert_off[i + 1]; a2, b2 = B.vert_off[k], B.vert_off[k + 1]
        B.c0x[a2:b2], B.c0y[a2:b2], B.cax[a2:b2], B.cay[a2:b2] = A.c0x[a:b], A.c0y[a:b], A.cax[a:b], A.cay[a:b]
    truth = B.copy()                                   # what a full upload would send
    for k, i in enumerate(keep):
        if src[k]:
            truth.StressH[k] = A.StressH[i]; truth.mcX[k] = A.mcX[i]; truth.mcY[k] = A.mcY[i]; truth.mcA[k] = A.mcA[i]
            B.StressH[k] = np.nan; B.mcX[k] = np.nan   # the resync must not read these rows from the host
    sim.resync(B, src)
    got = sim.download(capi.SYNC_ALL)
    assert np.array_equal(got.StressH, truth.StressH)
    sim.fetch(np.arange(1, B.n + 1, dtype=np.int32), capi.F_MC)
    assert np.array_equal(got.mcX, truth.mcX) and np.array_equal(got.mcA, truth.mcA)
    O = truth.copy()
    steps(sim, O, P, 6, start=5)
    check(sim, O, "after resync")


def test_empty_monte_carlo_mask_is_reported_before_anything_changes(ctx):
    S, P = scenes.pack(80, periodic=True, n_stress=4, n_mc=200)
    P.set_reference_ocean()
    S.mcA[[5, 41]] = 0
    S.mcA[17] = 0; S.alive[17] = 0
    A, G = S.copy(), S.copy()
    sim = capi.FloeInteractions(ctx, G)
    P.doInt = 0
    oracle.step(A, P, threads=2); sim.step(P)
    before = check(sim, A).copy()
    P.doInt = 1
    with pytest.raises(capi.NeedsMc) as e:
        sim.step(P)
    assert e.value.ids.tolist() == [6, 42]
    G = sim.download(capi.SYNC_ALL)
    assert hoststep.compare_states(before, G) == [] and np.array_equal(before.Stress, G.Stress)
    rng = np.random.default_rng(5)
    for i in (5, 41):
        a, b = S.vert_off[i], S.vert_off[i + 1]
        X = G.rmax[i] * (2 * rng.random(G.n_mc) - 1); Y = G.rmax[i] * (2 * rng.random(G.n_mc) - 1)
        inside = synth._inpolygon(X, Y, G.cax[a:b], G.cay[a:b])
        for T in (A, G):
            T.mcX[i], T.mcY[i], T.mcA[i] = X, Y, inside
    sim.update(e.value.ids, capi.F_MC)
    ra = oracle.step(A, P, threads=2)
    sim.step(P)
    off, rows = sim.interactions()
    assert np.array_equal(ra.inter, rows)
    check(sim, A, "after regeneration")


@pytest.mark.parametrize("periodic", [False, True])
def test_one_floe_twenty_times_the_median_grows_the_lists(ctx, periodic):
    S, P, k = scenes.big_among_small(3000, 20.0, periodic=periodic, centre=(4.0e4, 1e4) if periodic else (0.0, 0.0))
    A, G = S.copy(), S.copy()
    sim = capi.FloeInteractions(ctx, G)
    for it in range(3):
        P.doInt = int(it == 0)
        ra = oracle.step(A, P, threads=16)
        sim.step(P)
        pairs, n_ext = sim.pairs()
        off, rows = sim.interactions()
        assert n_ext == ra.n_ext and np.array_equal(pairs, ra.pairs), "step %d: candidate pairs" % it
        assert np.array_equal(off, ra.inter_off) and np.array_equal(rows, ra.inter), "step %d" % it
        check(sim, A, "step %d" % it)
    own = (ra.pairs[:, 0] == k + 1).sum(); mirrored = (ra.pairs[:, 1] == k + 1).sum()
    assert own > 32 and mirrored > 32 and ra.inter_off[k + 1] - ra.inter_off[k] > 32, (own, mirrored)


def test_ghosts_for_the_matlab_tail(ctx):
    S, P = scenes.pack(300, periodic=True)
    S.Xi += 300.0; S.Yi += 200.0                    # the pack reaches across the east and north edges: ghosts from the first step on
    A, G = S.copy(), S.copy()
    sim = capi.FloeInteractions(ctx, G)
    for it in range(4):
        P.doInt = int(it == 0)
        want = scenes.ghost_list(A, P)              # from the state the step starts with
        ra = oracle.step(A, P, threads=8)
        sim.step(P)
        src, parent, x, y = sim.ghosts()
        assert len(src) == ra.n_ext - S.n == len(want) and len(want) > 20
        assert [tuple(t) for t in zip(src.tolist(), parent.tolist(), x.tolist(), y.tolist())] == want
    # ids in `interactions` above n refer to these entries (floe_interactions_all.m:305-330 indexes them)
    off, rows = sim.interactions()
    partner = rows[np.isfinite(rows[:, 0]), 0]
    assert partner.max() > S.n and partner.max() <= S.n + len(src)


def test_configs4_standin_at_its_prescribed_size(ctx):
    """BASELINE configs[4] stand-in at the size SURVEY 8(d).5 prescribes: ~1e4 floes in the walled 2e5 x 1e6 m channel,
    12 static concave islands as Nb (30-90 vertex outlines of the reference's FloeShapes.mat, never paired -- as
    shipped, floe_interactions_all.m:76-120), wind V = -10 m/s: every field after each of 6 timesteps identical to the
    oracle, one of them a forcing refresh."""
    import os
    S, P = scenes.nares_full()
    assert S.n > 9000 and S.nb == 12 and not P.PERIODIC
    A, G = S.copy(), S.copy()
    sim = capi.FloeInteractions(ctx, G)
    threads = min(os.cpu_count() or 1, 32)
    for it in range(6):
        P.doInt = int(it == 1)
        ra = oracle.step(A, P, threads=threads)
        st = sim.step(P)
        pairs, n_ext = sim.pairs()
        off, rows = sim.interactions()
        assert np.array_equal(pairs, ra.pairs), "step %d: candidate pairs" % it
        assert np.array_equal(ra.inter_off, off) and np.array_equal(ra.inter, rows), "step %d" % it
        check(sim, A, "step %d" % it)
    assert (rows[:, 0] == np.inf).any() and len(rows) > 15000
    assert (ra.pairs > S.nb).all()                  # nothing is ever paired with the static floes (as shipped)
    assert np.abs(A.FyOA[S.nb:]).min() > 0          # the wind acts on every floe
