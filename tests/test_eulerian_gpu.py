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


@pytest.mark.parametrize("scene,grid", [("nares", (8, 20)), ("nares", (3, 4)), ("pack_nb", (12, 9))])
def test_eulerian_fields_with_topography(ctx, scene, grid):
    """Nb > 0 (calc_eulerian_data.m:10-24,141-143): the first Nb floes are cut out of the cells and do not count; the
    cells that meet them go through the clip kernel, the others through the plain rectangle clip."""
    if scene == "nares":
        S, P = scenes.nares_standin()
        periodic = False
    else:
        S, P = scenes.pack(500, periodic=True)
        S.nb = 4
        S.Xi += 300.0
        periodic = True
    P.set_reference_ocean()
    A, G = S.copy(), S.copy()
    sim = capi.FloeInteractions(ctx, G)
    for it in range(6):
        P.doInt = int(it % 5 == 0)
        oracle.step(A, P, threads=8); sim.step(P)
    sim.download(capi.SYNC_ALL)
    import hoststep
    hoststep.check_stress(A, G)
    A.Stress[:] = G.Stress
    Nx, Ny = grid
    nb = int(S.nb)
    want = oracle.eulerian_data(A, Nx, Ny, nb, P.c2_boundary, periodic)
    got, ms = sim.eulerian_data(Nx, Ny, nb, P.c2_boundary, periodic)
    plain = oracle.eulerian_data(A, Nx, Ny, 0, P.c2_boundary, periodic)
    for k in oracle.EULERIAN_FIELDS:
        assert np.array_equal(want[k], got[k], equal_nan=True), (k, float(np.nanmax(np.abs(want[k] - got[k]))))
    assert (want["c"] != plain["c"]).any() and want["Mtot"].sum() < plain["Mtot"].sum()      # the topography was taken out
    print("eulerian with topography %s %dx%d: %.3f ms in the cell kernel, %d cells changed" % (scene, Nx, Ny, ms, int((want["c"] != plain["c"]).sum())))


def test_a_cell_with_more_floes_than_the_lds_lists_hold(ctx):
    """the reference has no limit on the floes of a cell -- a 1 x 1 grid (a domain average) and a 2 x 1 grid over
    2600 floes put more than 2048 potential floes into a cell; the lists then live in HBM, same sums in the same order."""
    S, P = scenes.pack(2600, periodic=True)
    P.set_reference_ocean()
    S.Xi += 300.0
    A, G = S.copy(), S.copy()
    sim = capi.FloeInteractions(ctx, G)
    for it in range(3):
        P.doInt = int(it == 0)
        oracle.step(A, P, threads=8); sim.step(P)
    sim.download(capi.SYNC_ALL)
    A.Stress[:] = G.Stress
    for Nx, Ny in ((1, 1), (2, 1), (9, 7)):
        want = oracle.eulerian_data(A, Nx, Ny, 0, P.c2_boundary, True)
        got, ms = sim.eulerian_data(Nx, Ny, 0, P.c2_boundary, True)
        for k in oracle.EULERIAN_FIELDS:
            assert np.array_equal(want[k], got[k]), (Nx, Ny, k, float(np.abs(want[k] - got[k]).max()))
    assert want["c"].max() > 0.5
