"""szCornerCandidates (corners.m:13-92 on the device-resident contact rows) against the oracle: the `da`
flags and the eligible set identical, the interior angles bit-identical with the shared arccosine and within 1e-12
of the host libm's; then the whole decision with one injected random stream (Subzero.m:339-348, corners.m:70-71,91-94)."""
import numpy as np
import pytest

import oracle
import scenes
from subzero_amd import capi, corners

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene,periodic", [("pack", True), ("pack", False), ("concave", True)])
def test_corner_candidates_match_the_oracle(ctx, scene, periodic):
    if scene == "pack":
        S, P = scenes.pack(700, periodic=periodic)
    else:
        S, P = scenes.concave(100, periodic=periodic)
    P.set_reference_ocean()
    if periodic:
        S.Xi += 400.0; S.Yi += 250.0          # outlines across both edges: X images, Y images, images of images
    A, G = S.copy(), S.copy()
    sim = capi.FloeInteractions(ctx, G)
    for it in range(10):                      # (the driver grinds every 10th step, Subzero.m:339)
        P.doInt = int(it % 5 == 0)
        r = oracle.step(A, P, threads=8); sim.step(P)
    off_g, inter_g = sim.interactions()
    assert np.array_equal(off_g, r.inter_off) and np.array_equal(inter_g[:, 0], r.inter[:, 0])
    sim.download(capi.SYNC_ALL)
    # one random stream for both sides
    rng = np.random.default_rng(2)
    Nb = 0
    keep = corners.keep_mask(rng.uniform(size=A.n), A.OverlapArea, A.area, Nb)
    keep_g = corners.keep_mask(np.random.default_rng(2).uniform(size=G.n), G.OverlapArea, G.area, Nb)
    assert np.array_equal(keep, keep_g)
    ids = (np.nonzero(~keep)[0] + 1).astype(np.int32)
    assert len(ids) > 0.5 * A.n
    for nb_arg in (0, 5):                     # (:55 skips the first Nb floes of the SUBSET)
        off, cnt, ang, da, N0 = oracle.corner_candidates(A, r.inter_off, r.inter, ids, nb_arg, P.c2_boundary)
        goff, gcnt, gang, gda, gN0, ms = sim.corner_candidates(ids, nb_arg, P.c2_boundary)
        assert gN0 == N0 and (N0 > A.n or not periodic)
        assert np.array_equal(off, goff) and np.array_equal(cnt, gcnt)
        assert np.array_equal(da, gda)
        assert np.array_equal(ang, gang), float(np.abs(ang - gang).max())
        if nb_arg:
            assert (cnt[:nb_arg] == 0).all()
    with oracle.mode(oracle.MODE_LIBM):
        _, _, ang_libm, _, _ = oracle.corner_candidates(A, r.inter_off, r.inter, ids, 0, P.c2_boundary)
    off, cnt, ang, da, N0 = oracle.corner_candidates(A, r.inter_off, r.inter, ids, 0, P.c2_boundary)
    goff, gcnt, gang, gda, gN0, ms = sim.corner_candidates(ids, 0, P.c2_boundary)
    assert np.abs(ang_libm - gang).max() <= 1e-12 * 360
    u = rng.uniform(size=corners.draws_needed(cnt))
    grind, frac = corners.grind_flags(off, cnt, ang, da, u)
    ggrind, gfrac = corners.grind_flags(goff, gcnt, gang, gda, u)
    assert np.array_equal(grind, ggrind) and np.array_equal(frac, gfrac)
    assert (cnt > 0).sum() > 10 and da.sum() > 20 and grind.sum() > 0
    print("corners %s periodic=%d: %d selected, %d eligible, %d vertices flagged, %d ground, %d floes to reshape, %.3f ms"
          % (scene, periodic, len(ids), (cnt > 0).sum(), da.sum(), grind.sum(), frac.sum(), ms))
    # frac_corner.m for the floes to reshape (corners.m:94-98): the pieces of the plain case, device against oracle
    # (the oracle works on the downloaded state: the same floes the device holds)
    sel = np.nonzero(frac)[0]
    fids = ids[sel]
    foff = np.zeros(len(sel) + 1, np.int32)
    fgrind = []
    for k, q in enumerate(sel):
        n = int(G.vert_off[ids[q]] - G.vert_off[ids[q] - 1])
        g = np.zeros(n, np.uint8); g[:cnt[q]] = grind[off[q]:off[q] + cnt[q]]
        fgrind.append(g); foff[k + 1] = foff[k] + n
    fgrind = np.concatenate(fgrind) if fgrind else np.zeros(0, np.uint8)
    onp, opoff, opieces, orings = oracle.frac_corner(G, fids, foff, fgrind)
    gnp, gpoff, gpieces, grings, cms = sim.corner_cuts(fids, foff, fgrind)
    assert np.array_equal(onp, gnp) and np.array_equal(opoff, gpoff)
    assert np.array_equal(opieces, gpieces), float(np.abs(opieces - gpieces).max())
    assert len(orings) == len(grings) and all(np.array_equal(a, b) for a, b in zip(orings, grings))
    cut = (gnp > 0).sum()
    assert cut > 0 and (gpieces[:, 0] == 1).sum() >= cut
    # what every cut conserves: the pieces of a floe add up to its area and mass (no boolean operation lost or made ice)
    for k in np.nonzero(gnp > 0)[0]:
        i = fids[k] - 1
        a, b = int(G.vert_off[i]), int(G.vert_off[i + 1])
        x, y = G.cax[a:b - 1], G.cay[a:b - 1]
        area0 = 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(np.roll(x, -1), y))
        pc = gpieces[gpoff[k]:gpoff[k + 1]]
        assert abs(pc[:, 3].sum() - area0) <= 1e-9 * area0
        assert abs(pc[:, 4].sum() - G.mass[i]) <= 1e-9 * G.mass[i]
    print("frac_corner: %d floes to reshape, %d cut on the device (%d pieces), %d left to frac_corner.m, %.3f ms" % (len(sel), cut, len(gpieces), (gnp == 0).sum(), cms))


def test_corner_candidates_argument_errors(ctx):
    S, P = scenes.pack(60, periodic=True)
    sim = capi.FloeInteractions(ctx, S.copy())
    with pytest.raises(capi.SubzeroError):                    # no step taken: there are no contact rows
        sim.corner_candidates(np.array([1], np.int32), 0, P.c2_boundary)
    P.set_reference_ocean(); P.doInt = 1
    sim.step(P)
    with pytest.raises(capi.SubzeroError):
        sim.corner_candidates(np.array([61], np.int32), 0, P.c2_boundary)
    off, cnt, ang, da, N0, ms = sim.corner_candidates(np.zeros(0, np.int32), 0, P.c2_boundary)
    assert len(cnt) == 0 and off[0] == 0
