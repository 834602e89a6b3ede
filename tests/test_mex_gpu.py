"""The MATLAB gateways driven end to end on the GPU through the test double of the MEX API (tests/mexstub/,
tests/mexharness.py): what a MATLAB session would see from subzero_step / clipper, checked against the oracle and the
reference's own Clipper."""
import numpy as np
import pytest

import clipgen as g
import hoststep
import mexharness as mx
import oracle
import scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stepgw():
    gw = mx.Gateway("subzero_step_mex.cpp", "subzero_step")
    yield gw
    gw.call(0, mx.string("close"))


@pytest.fixture(scope="module")
def clipper():
    return mx.Gateway("clipper_mex.cpp", "clipper")


def gw_step(gw, F, P, nlhs=5):
    out = gw.call(nlhs, mx.string("step"), F, *mx.step_args(P))
    return out


def test_step_gateway_matches_the_oracle_and_keeps_the_history_on_the_device(stepgw):
    S, P = scenes.pack(120, periodic=True, n_stress=5, n_mc=120)
    P.set_reference_ocean()
    S.Xi += 250.0                                     # ghosts from the first step on
    A, G = S.copy(), S.copy()
    F, = stepgw.call(1, mx.string("upload"), mx.floe_struct(G), mx.scalar(0))
    for it in range(8):
        P.doInt = int(it % 4 == 0)
        want_ghosts = scenes.ghost_list(A, P)
        ra = oracle.step(A, P, threads=8)
        F, kill, transfer, ghosts, need = gw_step(stepgw, F, P)
        assert mx.numel(need) == 0
        inter = mx.read_floes(F, G)
        for i in range(S.n):
            want = ra.inter[ra.inter_off[i]:ra.inter_off[i + 1]]
            assert inter[i].shape == ((len(want), 7) if len(want) else (0, 0)) and np.array_equal(inter[i].reshape(-1, 7), want), (it, i)
        bad = [b for b in hoststep.compare_states(A, G) if b[0] not in ("StressH", "kill", "transfer")]
        assert bad == [], (it, bad)
        k = mx.to_numpy(kill).ravel(); t = mx.to_numpy(transfer).ravel()
        assert len(k) == ra.n_ext and np.array_equal(k, A.kill[:ra.n_ext]) and np.array_equal(t, A.transfer[:ra.n_ext])
        gs = [mx.to_numpy(mx.get_field(ghosts, 0, f)).ravel() for f in ("src", "parent", "Xi", "Yi")]
        assert [tuple(v) for v in zip(*[x.tolist() for x in gs])] == want_ghosts and len(want_ghosts) > 5
    # Stress followed the reference although StressH never travelled; the host copy of the history is stale by design
    assert np.abs(A.Stress).max() > 0 and np.abs(G.StressH).max() == 0
    # 'upload' of the stale array is refused; 'fetch' brings the history, then it matches
    with pytest.raises(mx.MexError) as e:
        stepgw.call(1, mx.string("upload"), F, mx.scalar(0))
    assert "newer stress history" in str(e.value)
    F, = stepgw.call(1, mx.string("fetch"), F, mx.row(np.arange(1, S.n + 1)), mx.string("h"))
    mx.read_floes(F, G, bulk=True)
    hoststep.check_stress(A, G, "after fetch")
    # a MATLAB-side change: two floes die and are removed (floe_interactions_all.m:12-13), one floe gets a kick
    keep = [i for i in range(S.n) if i not in (3, 77)]
    A2 = sub_state(A, keep); G2 = sub_state(G, keep)
    F2 = mx.floe_struct(G2)
    for k, i in enumerate(keep):
        mx.set_field(F2, k, "gpuIndex", mx.scalar(i + 1))
        mx.set_field(F2, k, "StressH", mx.numeric(np.zeros((2, 2, 5))))     # stale on purpose: the device's history must win
    F2, = stepgw.call(1, mx.string("resync"), F2, mx.scalar(0))
    for it in range(8, 13):
        P.doInt = int(it % 4 == 0)
        oracle.step(A2, P, threads=8)
        F2, = gw_step(stepgw, F2, P, nlhs=1)
    F2, = stepgw.call(1, mx.string("fetch"), F2, mx.row(np.arange(1, len(keep) + 1)), mx.string("h"))
    mx.read_floes(F2, G2, bulk=True)
    assert [b for b in hoststep.compare_states(A2, G2) if b[0] not in ("kill", "transfer")] == []
    hoststep.check_stress(A2, G2, "after resync")


def sub_state(S, keep):
    from subzero_amd import synth
    vo = np.zeros(len(keep) + 1, np.int32)
    for k, i in enumerate(keep):
        vo[k + 1] = vo[k] + S.vert_off[i + 1] - S.vert_off[i]
    T = synth.FloeState(len(keep), vo, S.nb, S.n_stress, S.n_mc)
    for f in synth.FloeState.ARRAYS:
        if f in ("vert_off", "kill", "transfer"):
            continue
        a = getattr(S, f)
        if f in ("c0x", "c0y", "cax", "cay"):
            for k, i in enumerate(keep):
                getattr(T, f)[vo[k]:vo[k + 1]] = a[S.vert_off[i]:S.vert_off[i + 1]]
        else:
            getattr(T, f)[...] = a[keep]
    return T


def test_step_gateway_reports_floes_that_need_new_monte_carlo_points(stepgw):
    S, P = scenes.pack(40, periodic=False, n_stress=3, n_mc=60)
    S.mcA[11] = 0
    A, G = S.copy(), S.copy()
    stepgw.call(0, mx.string("close"))
    F, = stepgw.call(1, mx.string("upload"), mx.floe_struct(G), mx.scalar(0))
    P.doInt = 1
    F1, kill, transfer, ghosts, need = gw_step(stepgw, F, P)
    assert mx.to_numpy(need).ravel().tolist() == [12.0]
    H = G.copy(); mx.read_floes(F1, H)
    assert hoststep.compare_states(G, H) == []            # nothing changed
    rng = np.random.default_rng(3)
    a, b = S.vert_off[11], S.vert_off[12]
    from subzero_amd import synth
    X = G.rmax[11] * (2 * rng.random(G.n_mc) - 1); Y = G.rmax[11] * (2 * rng.random(G.n_mc) - 1)
    inside = synth._inpolygon(X, Y, G.cax[a:b], G.cay[a:b])
    for T in (A, G):
        T.mcX[11], T.mcY[11], T.mcA[11] = X, Y, inside
    F = mx.floe_struct(G)
    stepgw.call(0, mx.string("update"), F, mx.row([12]), mx.string("m"))
    ra = oracle.step(A, P, threads=4)
    F, kill, transfer, ghosts, need = gw_step(stepgw, F, P)
    assert mx.numel(need) == 0
    mx.read_floes(F, G)
    assert [b for b in hoststep.compare_states(A, G) if b[0] not in ("StressH", "kill", "transfer")] == []


def test_clipper_gateway_boolean_mode_against_the_reference_build(clipper):
    if g.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    ps = g.shape_pairs(40, seed=8)
    for method in range(4):
        ref = g.run_ref(ps, method)
        ax, ay, aoff, bx, by, boff = ps.arrays()
        for p in range(len(ps)):
            a = np.stack([ax[aoff[p]:aoff[p + 1]], ay[aoff[p]:aoff[p + 1]]], 1)
            b = np.stack([bx[boff[p]:boff[p + 1]], by[boff[p]:boff[p + 1]]], 1)

            def ring(v):
                s = mx.struct(1, ["x", "y"])
                mx.set_field(s, 0, "x", mx.numeric(v[:, 0], "int64")); mx.set_field(s, 0, "y", mx.numeric(v[:, 1], "int64"))
                return s
            out, = clipper.call(1, ring(a), ring(b), mx.scalar(method))
            rings = [np.stack([mx.to_numpy(mx.get_field(out, r, "x")).ravel(), mx.to_numpy(mx.get_field(out, r, "y")).ravel()], 1)
                     for r in range(mx.numel(out))]
            want = g.rings_of(ref, p)
            assert len(rings) == len(want), (method, p)
            for r, w in zip(rings, want):
                assert r.dtype == np.float64 and np.array_equal(r, w.astype(np.float64)), (method, p)


def test_clipper_gateway_takes_several_contours_per_operand(clipper):
    if g.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    # subject: two disjoint squares + a hole-like inner ring; clip: two overlapping rings -- every fill rule and operation
    S32 = 1 << 32
    sq = lambda x0, y0, s: np.array([[x0, y0], [x0 + s, y0], [x0 + s, y0 + s], [x0, y0 + s]], np.int64) * S32
    subj = [sq(0, 0, 10), sq(20, 0, 10), sq(2, 2, 4)[::-1]]
    clip = [sq(5, -3, 8), sq(8, 3, 16)]
    for method in range(4):
        for sf in range(4):
            for cf in (0, 1):
                want = g.run_ref_paths(subj, clip, method, sf, cf)
                def rings(vs):
                    s = mx.struct(len(vs), ["x", "y"])
                    for k, v in enumerate(vs):
                        mx.set_field(s, k, "x", mx.numeric(v[:, 0], "int64")); mx.set_field(s, k, "y", mx.numeric(v[:, 1], "int64"))
                    return s
                out, = clipper.call(1, rings(subj), rings(clip), mx.scalar(method), mx.scalar(sf), mx.scalar(cf))
                got = [np.stack([mx.to_numpy(mx.get_field(out, r, "x")).ravel(), mx.to_numpy(mx.get_field(out, r, "y")).ravel()], 1).astype(np.int64)
                       for r in range(mx.numel(out))]
                assert len(got) == len(want), (method, sf, cf)
                for r, w in zip(got, want):
                    assert np.array_equal(r, w), (method, sf, cf)


def test_step_gateway_corners_and_eulerian_commands(stepgw):
    """'corners' (corners.m:13-92 per vertex) and 'eulerian' (calc_eulerian_data.m) as a MATLAB session sees them."""
    S, P = scenes.pack(150, periodic=True, n_stress=5, n_mc=120)
    P.set_reference_ocean()
    S.Xi += 250.0
    A, G = S.copy(), S.copy()
    stepgw.call(0, mx.string("close"))             # (a new model run: whatever an earlier test left resident goes)
    F, = stepgw.call(1, mx.string("upload"), mx.floe_struct(G), mx.scalar(0))
    for it in range(6):
        P.doInt = int(it % 4 == 0)
        ra = oracle.step(A, P, threads=8)
        F, kill, transfer, ghosts, need = gw_step(stepgw, F, P)
    mx.read_floes(F, G)
    ids = np.arange(2, S.n + 1, 2)
    cb = mx.numeric(np.asarray(P.c2_boundary, float))
    C, = stepgw.call(1, mx.string("corners"), mx.row(ids), mx.scalar(3), cb)
    off, cnt, ang, da, N0 = oracle.corner_candidates(A, ra.inter_off, ra.inter, ids.astype(np.int32), 3, P.c2_boundary)
    assert mx.numel(C) == len(ids) and (cnt > 0).sum() > 10 and (cnt[:3] == 0).all()
    for q in range(len(ids)):
        a = mx.to_numpy(mx.get_field(C, q, "angles")).ravel(); d = mx.to_numpy(mx.get_field(C, q, "da")).ravel()
        assert len(a) == cnt[q] == len(d)
        assert np.array_equal(a, ang[off[q]:off[q] + cnt[q]]) and np.array_equal(d, da[off[q]:off[q] + cnt[q]].astype(float))
    # 'cuts': frac_corner.m for the plain case, the pieces as the oracle restates them (two corners of every eligible floe
    # marked: the vertices with the largest `da`-free choice do not matter here, only that both sides see the same flags)
    sel = [q for q in range(len(ids)) if cnt[q] >= 4][:12]
    glist = []
    for q in sel:
        g = np.zeros(cnt[q]); g[1] = 1; g[cnt[q] // 2 + 1 if cnt[q] // 2 + 1 < cnt[q] else 0] = 1
        glist.append(g)
    K, = stepgw.call(1, mx.string("cuts"), mx.row(ids[sel]), mx.numeric(np.concatenate(glist)), mx.row([len(g) for g in glist]))
    foff = np.zeros(len(sel) + 1, np.int32); fg = []
    for k, q in enumerate(sel):
        nst = int(G.vert_off[ids[q]] - G.vert_off[ids[q] - 1])
        gg = np.zeros(nst, np.uint8); gg[:cnt[q]] = glist[k]; fg.append(gg); foff[k + 1] = foff[k] + nst
    onp, opoff, opieces, orings = oracle.frac_corner(G, ids[sel].astype(np.int32), foff, np.concatenate(fg))
    assert mx.numel(K) == len(sel) and (onp > 0).sum() >= 1
    for k in range(len(sel)):
        pm = mx.to_numpy(mx.get_field(K, k, "pieces"))
        assert pm.shape[0] == onp[k]
        if onp[k]:
            assert np.array_equal(pm, opieces[opoff[k]:opoff[k + 1]])
            rings = mx.to_numpy(mx.get_field(K, k, "rings")); ro = mx.to_numpy(mx.get_field(K, k, "ring_off")).ravel().astype(int)
            for t in range(onp[k]):
                assert np.array_equal(rings[:, ro[t]:ro[t + 1]], orings[opoff[k] + t])
    E, = stepgw.call(1, mx.string("eulerian"), mx.scalar(6), mx.scalar(4), mx.scalar(0), cb, mx.scalar(1))
    # (Stress: the device's exact mean of the history against the reference's rounded one, see hoststep.check_stress;
    # the history itself stays on the device in this protocol)
    assert np.abs(A.Stress - G.Stress).max() <= 1e-10 * max(1.0, np.abs(A.Stress).max())
    A.Stress[:] = G.Stress
    want = oracle.eulerian_data(A, 6, 4, 0, P.c2_boundary, True)
    for k in oracle.EULERIAN_FIELDS:
        got = mx.to_numpy(mx.get_field(E, 0, k))
        assert got.shape == (4, 6) and np.array_equal(got, want[k]), k
    with pytest.raises(mx.MexError):
        stepgw.call(1, mx.string("corners"), mx.row([S.n + 1]), mx.scalar(0), cb)


def test_step_gateway_deform_and_potential_commands(stepgw):
    S, P = scenes.pack(150, periodic=False, n_stress=5, n_mc=120)
    P.set_reference_ocean()
    A, G = S.copy(), S.copy()
    stepgw.call(0, mx.string("close"))
    F, = stepgw.call(1, mx.string("upload"), mx.floe_struct(G), mx.scalar(0))
    for it in range(6):
        P.doInt = int(it % 4 == 0)
        ra = oracle.step(A, P, threads=8)
        F, kill, transfer, ghosts, need = gw_step(stepgw, F, P)
    ids = np.arange(1, S.n + 1, 3)
    D, = stepgw.call(1, mx.string("deform"), mx.row(ids))
    ch, X, Y, Ar, rings = oracle.fracture_deform(A, ra.inter_off, ra.inter, ids.astype(np.int32))
    assert ch.sum() > 3
    for q in range(len(ids)):
        assert mx.to_numpy(mx.get_field(D, q, "changed")).ravel()[0] == ch[q]
        if ch[q]:
            assert mx.to_numpy(mx.get_field(D, q, "Xi")).ravel()[0] == X[q] and mx.to_numpy(mx.get_field(D, q, "area")).ravel()[0] == Ar[q]
            assert np.array_equal(mx.to_numpy(mx.get_field(D, q, "c_alpha")), rings[q])
    Pt, = stepgw.call(1, mx.string("potential"), mx.row([1, 50]))
    for q, fid in enumerate([1, 50]):
        i = fid - 1
        d = np.sqrt((A.Xi[i] - A.Xi) ** 2 + (A.Yi[i] - A.Yi) ** 2)
        want = np.nonzero((A.alive != 0) & (d > 1) & (d < A.rmax[i] + A.rmax))[0] + 1
        assert np.array_equal(mx.to_numpy(mx.get_field(Pt, q, "floeNum")).ravel(), want.astype(float))


def test_advance_leaves_the_state_on_the_device_until_it_is_fetched(stepgw):
    """'advance': the timestep with nothing written back.  kill / transfer / dead floes come back every
    step; the stale MATLAB array is refused by 'upload' / 'resync' / 'update' until 'fetch' has brought it up to date;
    after the fetch it equals the oracle's array, interactions included."""
    S, P = scenes.pack(140, periodic=True, n_stress=5, n_mc=120)
    P.set_reference_ocean()
    S.Xi += 250.0
    A, G = S.copy(), S.copy()
    stepgw.call(0, mx.string("close"))
    F, = stepgw.call(1, mx.string("upload"), mx.floe_struct(G), mx.scalar(0))
    for it in range(7):
        P.doInt = int(it % 4 == 0)
        ra = oracle.step(A, P, threads=8)
        kill, transfer, need, dead, ghosts = stepgw.call(5, mx.string("advance"), *mx.step_args(P))
        assert mx.numel(need) == 0
        k = mx.to_numpy(kill).ravel(); t = mx.to_numpy(transfer).ravel()
        assert len(k) == ra.n_ext and np.array_equal(k, A.kill[:ra.n_ext]) and np.array_equal(t, A.transfer[:ra.n_ext])
        assert np.array_equal(mx.to_numpy(dead).ravel(), np.nonzero(A.alive == 0)[0] + 1.0)
        assert mx.numel(mx.get_field(ghosts, 0, "src")) == ra.n_ext - S.n
    # the MATLAB array has not moved
    assert mx.to_numpy(mx.get_field(F, 0, "Xi")).ravel()[0] == S.Xi[0]
    for cmd in ("upload", "resync"):
        with pytest.raises(mx.MexError) as e:
            stepgw.call(1, mx.string(cmd), F, mx.scalar(0))
        assert "'advance'" in str(e.value) and "fetch" in str(e.value)
    with pytest.raises(mx.MexError):
        stepgw.call(0, mx.string("update"), F, mx.row([1]), mx.string("s"))
    F, = stepgw.call(1, mx.string("fetch"), F, mx.row(np.arange(1, S.n + 1)), mx.string("sri"))
    inter = mx.read_floes(F, G)
    bad = [b for b in hoststep.compare_states(A, G) if b[0] not in ("StressH", "kill", "transfer", "Stress")]
    assert bad == [], bad
    assert np.abs(A.Stress - G.Stress).max() <= 1e-10 * max(1.0, np.abs(A.Stress).max())
    for i in range(S.n):
        want = ra.inter[ra.inter_off[i]:ra.inter_off[i + 1]]
        assert np.array_equal(inter[i].reshape(-1, 7), want), i
    F, = stepgw.call(1, mx.string("resync"), F, mx.scalar(0))        # up to date again: accepted
    # and the write-back step continues from there
    P.doInt = 0
    ra = oracle.step(A, P, threads=8)
    F, kill, transfer, ghosts, need = gw_step(stepgw, F, P)
    mx.read_floes(F, G)
    bad = [b for b in hoststep.compare_states(A, G) if b[0] not in ("StressH", "kill", "transfer", "Stress")]
    assert bad == [], bad


def test_update_of_some_floes_does_not_hide_a_change_of_another_from_resync(stepgw):
    """'update' takes over only the rows it sent.  A floe reshaped in MATLAB but not listed in the update is
    still a changed floe for the next 'resync' (its device-resident history is replaced by what the array carries),
    while the floes that stayed keep the history the device has accumulated."""
    S, P = scenes.pack(60, periodic=True, n_stress=4, n_mc=60)
    G = S.copy()
    stepgw.call(0, mx.string("close"))
    F, = stepgw.call(1, mx.string("upload"), mx.floe_struct(G), mx.scalar(0))
    for it in range(3):                                   # the device's history moves on; StressH is not written back
        P.doInt = int(it == 0)
        F = gw_step(stepgw, F, P)[0]
    stale = mx.to_numpy(mx.get_field(F, 8, "StressH")).copy()
    # MATLAB edits floe 5 (a velocity) and reshapes floe 9 (smaller outline), but lists only floe 5 in the update
    mx.set_field(F, 4, "Ui", mx.scalar(0.125))
    for name in ("c0", "c_alpha"):
        mx.set_field(F, 8, name, mx.numeric(0.5 * mx.to_numpy(mx.get_field(F, 8, name))))
    mx.set_field(F, 8, "area", mx.scalar(0.25 * float(mx.to_numpy(mx.get_field(F, 8, "area")).ravel()[0])))
    mx.set_field(F, 8, "rmax", mx.scalar(0.5 * float(mx.to_numpy(mx.get_field(F, 8, "rmax")).ravel()[0])))
    stepgw.call(0, mx.string("update"), F, mx.row([5]), mx.string("s"))
    F, = stepgw.call(1, mx.string("resync"), F, mx.scalar(0))
    H, = stepgw.call(1, mx.string("fetch"), F, mx.row([5, 9, 10]), mx.string("h"))
    assert np.array_equal(mx.to_numpy(mx.get_field(H, 8, "StressH")), stale)          # reshaped: the array's history
    for i in (4, 9):                                                                      # stayed: the device's history
        assert not np.array_equal(mx.to_numpy(mx.get_field(H, i, "StressH")), mx.to_numpy(mx.get_field(F, i, "StressH")))
