"""szEulerianData (calc_eulerian_data.m on the device-resident floes) against the oracle's restatement:
every field of every cell bit-identical, on periodic (ghost images, with the reference's own quirk in the Y images) and
walled fields, coarse and fine grids, real concave outlines."""
import numpy as np
import pytest

import oracle
import scenes
from subzero_amd import capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene,periodic,grid", [("pack", True, (10, 10)), ("pack", False, (40, 25)), ("concave", True, (12, 16))])
def test_eulerian_fields_match_the_oracle(ctx, scene, periodic, grid):
    if scene == "pack":
        S, P = scenes.pack(700, periodic=periodic)
    else:
        S, P = scenes.concave(100, periodic=periodic)
    P.set_reference_ocean()
    S.Xi += 400.0; S.Yi += 250.0 if periodic else 0.0      # floes reach across the edges: images in x (and the quirk in y)
    A, G = S.copy(), S.copy()
    sim = capi.FloeInteractions(ctx, G)
    for it in range(7):
        P.doInt = int(it % 5 == 0)
        oracle.step(A, P, threads=8); sim.step(P)
    A.alive[3] = 0                                          # a dead floe drops out of the list
    sim.download(capi.SYNC_ALL); G.alive[3] = 0; sim.update(np.array([4], np.int32), capi.F_STATE)
    import hoststep
    hoststep.check_stress(A, G)
    A.Stress[:] = G.Stress        # (the device carries the exact mean of the history, the oracle the reference's rounded one: same inputs for this check)
    Nx, Ny = grid
    want = oracle.eulerian_data(A, Nx, Ny, 0, P.c2_boundary, periodic)
    got, ms = sim.eulerian_data(Nx, Ny, 0, P.c2_boundary, periodic)
    for k in oracle.EULERIAN_FIELDS:
        assert got[k].shape == (Ny, Nx)
        assert np.array_equal(want[k], got[k]), (k, float(np.abs(want[k] - got[k]).max()))
    assert want["c"].max() > 0.5 and np.abs(want["stress"]).max() > 0 and np.abs(want["u"]).max() > 0
    print("eulerian %s %dx%d: %.3f ms" % (scene, Nx, Ny, ms))
