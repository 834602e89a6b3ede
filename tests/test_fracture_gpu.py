"""szFractureDeform (fracture_floe.m:14-52 on the device-resident floes and contact rows) against the
oracle's restatement around the reference's own Clipper: which floes are deformed, their new outlines (every vertex),
centroids and areas bit-identical."""
import numpy as np
import pytest

import oracle
import scenes
from subzero_amd import capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene,periodic", [("pack", True), ("pack", False), ("concave", True)])
def test_fracture_deformation_matches_the_oracle(ctx, scene, periodic):
    if scene == "pack":
        S, P = scenes.pack(500, periodic=periodic)
    else:
        S, P = scenes.concave(120, periodic=periodic)
    P.set_reference_ocean()
    if periodic:
        S.Xi += 300.0                         # ghost partners: rows the deformation must skip (fracture_floe.m:26)
    A, G = S.copy(), S.copy()
    sim = capi.FloeInteractions(ctx, G)
    for it in range(8):
        P.doInt = int(it % 5 == 0)
        r = oracle.step(A, P, threads=8); sim.step(P)
    sim.download(capi.SYNC_ALL)
    ids = np.arange(1, A.n + 1, dtype=np.int32)[::2]
    want = oracle.fracture_deform(A, r.inter_off, r.inter, ids)
    got = sim.fracture_deform(ids)
    assert np.array_equal(want[0], got[0])
    for k in (1, 2, 3):
        assert np.array_equal(want[k], got[k]), k
    for s in range(len(ids)):
        assert want[4][s].shape == got[4][s].shape and np.array_equal(want[4][s], got[4][s]), ids[s]
    touched = sum(1 for i in ids if r.inter_off[i] > r.inter_off[i - 1])
    assert want[0].sum() > 5 and touched > want[0].sum()          # some deformed, some in contact but left alone
    # a deformed floe lost the sliver its partner took: a little smaller, centroid moved a little
    ch = np.nonzero(want[0])[0]
    old = A.area[ids[ch] - 1]
    assert (want[3][ch] <= old * (1 + 1e-12)).all() and (want[3][ch] > 0.9 * old).all()
    print("fracture deform %s periodic=%d: %d listed, %d in contact, %d deformed" % (scene, periodic, len(ids), touched, want[0].sum()))


def test_fracture_deform_argument_errors(ctx):
    S, P = scenes.pack(60, periodic=True)
    sim = capi.FloeInteractions(ctx, S.copy())
    with pytest.raises(capi.SubzeroError):
        sim.fracture_deform(np.array([1], np.int32))
    P.set_reference_ocean(); P.doInt = 1
    sim.step(P)
    with pytest.raises(capi.SubzeroError):
        sim.fracture_deform(np.array([0], np.int32))
    out = sim.fracture_deform(np.zeros(0, np.int32))
    assert len(out[0]) == 0


def test_potential_interactions_match_floesimplify(ctx):
    """FloeSimplify.m:14-34 (numpy transcription of the loop as the checker): alive partners, centres more than 1 m
    apart, bounding circles overlapping, ascending."""
    S, P = scenes.pack(400, periodic=False)
    S.alive[7] = 0
    # one floe as large as a quarter of the domain: far more partners than the first guess of the room per floe
    S.rmax[11] = 6e4
    sim = capi.FloeInteractions(ctx, S.copy())
    ids = np.array([1, 12, 8, 200, 400], np.int32)
    got = sim.potential_interactions(ids)
    for s, fid in enumerate(ids):
        i = fid - 1
        d = np.sqrt((S.Xi[i] - S.Xi) ** 2 + (S.Yi[i] - S.Yi) ** 2)
        want = np.nonzero((S.alive != 0) & (d > 1) & (d < S.rmax[i] + S.rmax))[0] + 1
        assert np.array_equal(got[s], want), fid
    assert len(got[1]) > 64 and 8 not in got[0].tolist()
