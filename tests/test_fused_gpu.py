"""The fused launch chain of the steady step (subzero_amd/csrc/step_fused.hpp) against the scan-based chain it
replaces (SUBZERO_FUSED=0): two device-resident copies of the same field stepped side by side, every downloadable
word compared after every step -- candidate pairs, ghosts, contact rows, the whole floe state, kill / transfer.
(Each chain is also compared with the oracle by the step tests; this file pins the two to each other.)"""
import numpy as np
import pytest

import hoststep
import scenes
from subzero_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture
def ctx2():
    c = capi.Context(0)
    yield c
    c.close()


def run(ctx, ctx2, S, P, steps, monkeypatch, **kw):
    # two contexts: one resident floe array each
    A = S.copy(); B = S.copy()
    monkeypatch.setenv("SUBZERO_FUSED", "0")
    plain = capi.FloeInteractions(ctx, A)
    monkeypatch.setenv("SUBZERO_FUSED", "1")
    fused = capi.FloeInteractions(ctx2, B)
    monkeypatch.delenv("SUBZERO_FUSED")
    do_int_every = kw.get("do_int_every", 10)
    for it in range(steps):
        P.doInt = int(it % do_int_every == 0)
        plain.step(P); fused.step(P)
        pa, na = plain.pairs(); pb, nb = fused.pairs()
        assert na == nb and np.array_equal(pa, pb), "step %d: candidate pairs differ" % it
        for x, y in zip(plain.ghosts(), fused.ghosts()):
            assert np.array_equal(x, y), "step %d: ghosts differ" % it
        oa, ra = plain.interactions(); ob, rb = fused.interactions()
        assert np.array_equal(oa, ob) and np.array_equal(ra, rb), "step %d: contact rows differ" % it
        plain.download(capi.SYNC_ALL); fused.download(capi.SYNC_ALL)
        bad = hoststep.compare_states(A, B, exact=True)
        assert bad == [], "step %d: %s" % (it, bad)
        assert np.array_equal(A.kill, B.kill) and np.array_equal(A.transfer, B.transfer), "step %d: kill / transfer differ" % it
    d = fused.diagnostics()
    assert plain.diagnostics()["steps_fused"] == 0
    if kw.get("expect_fused", True):
        assert d["steps_fused"] >= steps - 3, d
    return d


def test_periodic_pack(ctx, ctx2, monkeypatch):
    S, P = scenes.pack(700, periodic=True)
    run(ctx, ctx2, S, P, 24, monkeypatch)


def test_walls_pack(ctx, ctx2, monkeypatch):
    S, P = scenes.pack(500, periodic=False, seed=7, vel_seed=8, sigma=0.3)
    run(ctx, ctx2, S, P, 24, monkeypatch, expect_fused=False)      # (floes at the walls defer pairs: some steps keep the read-back)


def test_forcing_every_third_step(ctx, ctx2, monkeypatch):
    S, P = scenes.pack(300, periodic=True, seed=21, sigma=0.05)
    P.set_reference_ocean()
    run(ctx, ctx2, S, P, 14, monkeypatch, do_int_every=3)


def test_thin_floes_take_the_force_list(ctx, ctx2, monkeypatch):
    # floes thinner than 0.1 m after the step's thinning get their forcing on every step (calc_trajectory.m:94,121)
    S, P = scenes.pack(300, periodic=True, seed=5)
    S.h[::7] = 0.05
    S.mass[::7] = S.mass[::7] * 0.05 / 1.0
    P.HFo = 0.0
    run(ctx, ctx2, S, P, 12, monkeypatch)


def test_concave_and_many_sided(ctx, ctx2, monkeypatch):
    S, P = scenes.concave(120, periodic=True)
    run(ctx, ctx2, S, P, 10, monkeypatch, expect_fused=False)


def test_topography_floes_and_their_ghosts(ctx, ctx2, monkeypatch):
    # the first floes of the array as Nb: they never start a candidate list, but their GHOSTS are entries beyond N0 and do
    S, P = scenes.pack(400, periodic=True, seed=33)
    S.nb = 40
    run(ctx, ctx2, S, P, 12, monkeypatch)


@pytest.mark.parametrize("seed", range(7000, 7012))
def test_random_scenes(ctx, ctx2, monkeypatch, seed):
    S, P, steps = scenes.random_scene(seed)
    run(ctx, ctx2, S, P, steps + 4, monkeypatch, do_int_every=3, expect_fused=False)


def test_a_full_cell_falls_back_and_grows(ctx, ctx2, monkeypatch):
    monkeypatch.setenv("SUBZERO_CELL_CAP", "2")
    S, P = scenes.pack(600, periodic=True, seed=9)
    d = run(ctx, ctx2, S, P, 12, monkeypatch, expect_fused=False)
    assert d["cell_cap"] > 2 and d["steps_fused"] > 0, d


def test_full_size_field(ctx, ctx2, monkeypatch):
    S = synth.voronoi_field(10000, 1.0, Lx=1e5, Ly=1e5, seed=4, vel_seed=2, vel_sigma=0.1, n_stress=4, n_mc=64)
    P = synth.StepParams(dt=10.0, Lx=1e5, Ly=1e5, periodic=True)
    P.Modulus = synth.reference_modulus(S)
    run(ctx, ctx2, S, P, 8, monkeypatch)


def test_queue_ahead_changes_nothing_but_the_wait(ctx, ctx2):
    """SZ_STEP_QUEUE_AHEAD: the first kernels of the next step queued behind this one's.  Two copies of a field, one stepped
    plainly, one with the promise kept (next doInt announced), broken (announced wrongly; a floe updated between two
    steps) and on steps that refresh the forcing: the same pairs, rows, state and marks after every step; the ghost list
    is refused after a step that queued ahead (also when an update has voided what was queued: the list is overwritten
    all the same) and served after one that did not."""
    S, P = scenes.pack(900, periodic=True)
    P.set_reference_ocean()
    A = S.copy(); B = S.copy()
    plain = capi.FloeInteractions(ctx, A)
    ahead = capi.FloeInteractions(ctx2, B)
    import copy
    Pa = copy.copy(P)
    steps = 30
    refused = 0
    for it in range(steps):
        P.doInt = Pa.doInt = int(it % 5 == 0)
        nxt = int((it + 1) % 5 == 0)
        Pa.queue_ahead = 0 if it in (11, steps - 1) else 1
        Pa.next_doInt = (1 - nxt) if it in (6, 9) else nxt          # (steps 6 and 9 announce the wrong doInt)
        plain.step(P); ahead.step(Pa)
        if it == 14:                                                # the caller changes a floe between two steps: what was queued ahead is void
            for sim, T in ((plain, A), (ahead, B)):
                sim.download(capi.SYNC_ALL)
                T.Ui[3] += 0.05; T.Vi[7] -= 0.02
                sim.update(np.array([4, 8], np.int32), capi.F_STATE)
        pa, na = plain.pairs(); pb, nb = ahead.pairs()
        assert na == nb and np.array_equal(pa, pb), "step %d: candidate pairs differ" % it
        oa, ra = plain.interactions(); ob, rb = ahead.interactions()
        assert np.array_equal(oa, ob) and np.array_equal(ra, rb), "step %d: contact rows differ" % it
        plain.download(capi.SYNC_ALL); ahead.download(capi.SYNC_ALL)
        bad = hoststep.compare_states(A, B, exact=True)
        assert bad == [], "step %d: %s" % (it, bad)
        assert np.array_equal(A.kill, B.kill) and np.array_equal(A.transfer, B.transfer), "step %d: kill / transfer differ" % it
        try:
            got = ahead.ghosts()
        except capi.SubzeroError:                       # the step queued the next one ahead: its ghost list is gone
            assert Pa.queue_ahead, it
            refused += 1
        else:
            for x, y in zip(plain.ghosts(), got):
                assert np.array_equal(x, y)
    assert refused >= steps - 8, refused
    d = ahead.diagnostics()
    assert d["steps_ahead"] >= steps - 8, d           # (not after the first steps, the two wrong announcements, the update, the step that did not ask)
    assert plain.diagnostics()["steps_ahead"] == 0
