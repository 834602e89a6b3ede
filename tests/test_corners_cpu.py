"""Corner grinding up to the `grind` decision (corners.m:13-92, polyangles.m): the oracle's restatement
against known answers and against an independent numpy evaluation, and the host-side selection rule."""
import numpy as np

import oracle
import scenes
from subzero_amd import corners


def test_polyangles_known_answers():
    # clockwise square, clockwise regular polygons: every interior angle 180 - 360/n
    sq = np.array([[0, 0], [0, 2], [2, 2], [2, 0]], float)
    assert np.allclose(oracle.polyangles(sq[:, 0], sq[:, 1]), 90.0, atol=1e-12)
    for n in (3, 5, 8, 13):
        t = -2 * np.pi * np.arange(n) / n + 0.3
        a = oracle.polyangles(7 * np.cos(t) + 1.5, 7 * np.sin(t) - 2.5)
        assert np.allclose(a, 180.0 - 360.0 / n, atol=1e-10), (n, a)
    # an L (clockwise from its lower-left corner): one reflex corner of 270 degrees
    Lx = np.array([0, 0, 1, 1, 3, 3], float); Ly = np.array([0, 3, 3, 1, 1, 0], float)
    a = oracle.polyangles(Lx, Ly)
    assert np.allclose(a, [90, 90, 90, 270, 90, 90], atol=1e-10), a
    assert abs(a.sum() - 180.0 * (len(a) - 2)) < 1e-9
    # the routine's own assumption: the first vertex counts as convex whatever it is (polyangles.m:91)
    a = oracle.polyangles(np.roll(Lx, -3), np.roll(Ly, -3))
    assert abs(a[0] - 90.0) < 1e-10


def test_polyangles_libm_mode_agrees():
    rng = np.random.default_rng(5)
    t = -np.sort(rng.uniform(0, 2 * np.pi, 11))
    r = 30 + rng.uniform(-8, 8, 11)                       # star-shaped: simple, with reflex corners
    k = int(np.argmax(r)); t = np.roll(t, -k); r = np.roll(r, -k)   # (start on a convex vertex: polyangles.m:91)
    x, y = r * np.cos(t), r * np.sin(t)
    a = oracle.polyangles(x, y)
    with oracle.mode(oracle.MODE_LIBM):
        b = oracle.polyangles(x, y)
    assert np.abs(a - b).max() < 1e-11 and abs(a.sum() - 180.0 * 9) < 1e-8 and (a > 180).any()


def _ray_inside(px, py, rx, ry):
    """even-odd ray casting, written independently of inpolygon.m (vertices exactly on an edge are not exercised)"""
    inside = False
    n = len(rx)
    for k in range(n):
        xa, ya, xb, yb = rx[k], ry[k], rx[(k + 1) % n], ry[(k + 1) % n]
        if (ya > py) != (yb > py) and px < (xb - xa) * (py - ya) / (yb - ya) + xa:
            inside = not inside
    return inside


def _colliding(periodic):
    S, P = scenes.pack(300, periodic=periodic)
    P.set_reference_ocean()
    S.Xi += 300.0
    r = None
    for it in range(6):
        P.doInt = int(it % 5 == 0)
        r = oracle.step(S, P, threads=8)
    return S, P, r


def test_candidates_against_an_independent_evaluation():
    for periodic in (True, False):
        S, P, r = _colliding(periodic)
        ids = np.arange(1, S.n + 1, 3, dtype=np.int32)
        off, cnt, ang, da, N0 = oracle.corner_candidates(S, r.inter_off, r.inter, ids, 0, P.c2_boundary)
        assert N0 >= S.n and (N0 > S.n or not periodic)   # (corners.m makes images whatever PERIODIC says)
        checked = flagged = 0
        for q, fid in enumerate(ids):
            i = fid - 1
            rows = r.inter[r.inter_off[i]:r.inter_off[i + 1]]
            if len(rows) == 0:
                assert cnt[q] == 0
                continue
            v0, v1 = S.vert_off[i], S.vert_off[i + 1] - 1          # stored rings carry their closing copy
            assert cnt[q] == v1 - v0
            x, y = S.cax[v0:v1] + S.Xi[i], S.cay[v0:v1] + S.Yi[i]
            want = np.zeros(v1 - v0, bool)
            for row in rows:
                if np.isinf(row[0]):
                    c = np.asarray(P.c2_boundary)
                    want |= ~((x >= c[0].min()) & (x <= c[0].max()) & (y >= c[1].min()) & (y <= c[1].max()))
                    continue
                if row[0] > S.n:
                    continue                                      # images: covered by the GPU-vs-oracle test
                j = int(row[0]) - 1
                w0, w1 = S.vert_off[j], S.vert_off[j + 1] - 1
                rx, ry = S.cax[w0:w1] + S.Xi[j], S.cay[w0:w1] + S.Yi[j]
                want |= np.array([_ray_inside(a, b, rx, ry) for a, b in zip(x, y)])
                want[np.argmin((x - row[3]) ** 2 + (y - row[4]) ** 2)] = True
            got = da[off[q]:off[q] + cnt[q]] != 0
            if any(row[0] > S.n and not np.isinf(row[0]) for row in rows):
                assert not (want & ~got).any()                    # the images can only add flags
            else:
                assert np.array_equal(want, got), (fid, want, got)
                checked += 1
            flagged += got.sum()
            a = ang[off[q]:off[q] + cnt[q]]
            assert abs(a.sum() - 180.0 * (cnt[q] - 2)) < 1e-7      # interior angles of a simple polygon
        assert checked > 20 and flagged > 20


def test_selection_rule_and_grind_flags():
    rng = np.random.default_rng(11)
    N, Nb = 50, 3
    u = rng.uniform(size=N)
    area = rng.uniform(1e6, 2e6, N); over = area * rng.uniform(0, 0.3, N)
    keep = corners.keep_mask(u, over, area, Nb)
    assert keep[:Nb].all()
    for k in range(Nb, N):
        assert keep[k] == (u[k] > 0.7 and not over[k] / area[k] > 0.15)
    keep2 = corners.keep_mask(u, over, area, Nb, min_floe_size=1.5e6)
    assert keep2[area < 1.5e6].all()
    # grind: two floes, the second not eligible
    offsets = np.array([0, 5, 9]); count = np.array([4, 0])
    angles = np.array([90.0, 60.0, 120.0, 90.0, 0.0, 0, 0, 0, 0]); da = np.array([1, 1, 0, 1, 0, 0, 0, 0, 0], np.uint8)
    assert corners.draws_needed(count) == 4
    anorm = 180.0 - 360.0 / 4
    draws = np.array([0.99, 0.5, 0.99, 0.999999])
    grind, frac = corners.grind_flags(offsets, count, angles, da, draws)
    assert list(grind[:4]) == [bool(0.99 > 90 / anorm), False, False, bool(0.999999 > 90 / anorm)]
    assert list(frac) == [False, False]
    grind, frac = corners.grind_flags(offsets, count, np.array([30.0, 30, 30, 30, 0, 0, 0, 0, 0]), da, draws)
    assert list(grind[:4]) == [True, True, False, True] and list(frac) == [True, False]


def test_frac_corner_known_answer_on_a_square():
    """frac_corner.m:20-52 on a 10 km square with one corner broken: every angle 90, Anorm = 90, alph = 90, so the cut
    points lie a fifth of the side from the corner -- a 2000 x 2000 m right triangle (2e6 m^2) comes off, the floe keeps
    1e8 - 2e6 m^2, masses and the Stress factors in proportion (:98-108, :166-172)."""
    import scenes
    S, P = scenes.pack(6, periodic=False)
    i = 2
    a, b = int(S.vert_off[i]), int(S.vert_off[i + 1])
    # overwrite floe i's outline by a square of the same vertex count?  -- build a one-floe state instead
    from subzero_amd import synth
    h = 5e3
    ring = np.array([[-h, h], [h, h], [h, -h], [-h, -h], [-h, h]])
    T = S.subset([i]) if hasattr(S, "subset") else None
    if T is None:
        T = S.copy()
        T.n = 1
        T.vert_off = np.array([0, 5], np.int32)
        T.cax = ring[:, 0].copy(); T.cay = ring[:, 1].copy(); T.c0x = ring[:, 0].copy(); T.c0y = ring[:, 1].copy()
        for k in ("Xi", "Yi", "mass", "area", "ksi_ice", "alive", "h"):
            setattr(T, k, np.asarray(getattr(S, k))[:1].copy())
        T.Xi[:] = 500.0; T.Yi[:] = -200.0; T.area[:] = 1e8; T.mass[:] = 920.0 * 1e8 * 0.5; T.ksi_ice[:] = 2e-6
    grind = np.array([0, 1, 0, 0], np.uint8)             # the top-right corner
    npieces, poff, pieces, rings = oracle.frac_corner(T, [1], [0, 4], grind)
    assert npieces[0] == 2 and len(pieces) == 2
    main, tri = pieces
    assert main[0] == 0 and tri[0] == 1
    assert abs(tri[3] - 2e6) < 1e-3 and abs(main[3] - (1e8 - 2e6)) < 1e-3
    assert abs(main[4] - T.mass[0] * (1e8 - 2e6) / 1e8) < 1e-3 and abs(tri[4] - T.mass[0] * 2e6 / 1e8) < 1e-3
    assert abs(main[10] - (1 - 0.02)) < 1e-12 and abs(tri[10] - 0.02) < 1e-12
    assert abs(main[5] - 0.5) < 1e-12 and abs(tri[5] - 0.5) < 1e-12          # h = mass / (rho area): unchanged
    assert main[8] == 1 and tri[8] == 1 and main[11] == 6 and tri[11] == 4   # both alive; closed rings of 5 + 1 and 3 + 1
    # the triangle's centroid: a third of the way from the corner (5000, 5000) to the two cut points
    assert abs(tri[1] - (500 + 5e3 - 2e3 / 3)) < 1e-6 and abs(tri[2] - (-200 + 5e3 - 2e3 / 3)) < 1e-6
    assert abs(tri[7] - np.hypot(2e3 / 3, 4e3 / 3)) < 1e-6                   # rmax: the centroid to a cut point
    # two equal angles with the first and the last vertex both marked: the last one is left alone (:11-14)
    grind = np.array([1, 0, 0, 1], np.uint8)
    npieces, poff, pieces, rings = oracle.frac_corner(T, [1], [0, 4], grind)
    assert npieces[0] == 2 and abs(pieces[1][3] - 2e6) < 1e-3
    # both neighbours of one edge broken: cuts of 2000 m each on a 10 km edge stay apart -> three pieces
    grind = np.array([0, 1, 1, 0], np.uint8)
    npieces, poff, pieces, rings = oracle.frac_corner(T, [1], [0, 4], grind)
    assert npieces[0] == 3 and abs(pieces[0][3] - (1e8 - 4e6)) < 1e-3
    # a sliver whose short edges are under 120 m: the whole edge would go -> not the plain case
    T.cax[:] = np.array([-50, 50, 50, -50, -50.0]); T.c0x[:] = T.cax
    npieces, poff, pieces, rings = oracle.frac_corner(T, [1], [0, 4], np.array([0, 1, 0, 0], np.uint8))
    assert npieces[0] == 0 and len(pieces) == 0


def test_frac_corner_short_edge_goes_whole():
    """frac_corner.m:43-44: an edge shorter than 120 m in x and in y is taken whole -- the cut ends AT the neighbouring
    vertex.  A convex pentagon with one 100 m edge (C -> D), corner C broken: the cut starts 120 m up the long edge from C
    (d = 100 m: the 120 m rule of :45-46) and ends at D; a triangle of 4800 m^2 comes off (dead: under 1e4 m^2, :113-115), the
    floe keeps its five vertices (C replaced by the one new cut point) and the rest of its area; with D broken as well the
    two triangles would overlap -- union / subtract of the reference, not this closed form -- and nothing is returned."""
    import scenes
    S, P = scenes.pack(6, periodic=False)
    ring = np.array([[-5000, 5000], [5000, 5000], [5000, -5000], [4920, -5060], [-5000, -5060], [-5000, 5000.0]])
    x, y = ring[:-1, 0], ring[:-1, 1]
    area = 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))
    T = S.copy()
    T.n = 1
    T.vert_off = np.array([0, 6], np.int32)
    T.cax = ring[:, 0].copy(); T.cay = ring[:, 1].copy(); T.c0x = ring[:, 0].copy(); T.c0y = ring[:, 1].copy()
    for k in ("Xi", "Yi", "mass", "area", "ksi_ice", "alive", "h"):
        setattr(T, k, np.asarray(getattr(S, k))[:1].copy())
    T.Xi[:] = 0.0; T.Yi[:] = 0.0; T.area[:] = area; T.mass[:] = 920.0 * area * 0.5; T.ksi_ice[:] = 2e-6
    npieces, poff, pieces, rings = oracle.frac_corner(T, [1], [0, 5], np.array([0, 0, 1, 0, 0], np.uint8))
    assert npieces[0] == 2
    main, tri = pieces
    assert abs(tri[3] - 4800.0) < 1e-6 and abs(main[3] - (area - 4800.0)) < 1e-3      # areas: nothing lost
    assert abs(main[4] + tri[4] - T.mass[0]) < 1e-3 * T.mass[0] * 1e-9 + 1e-3             # nor mass
    assert tri[8] == 0 and main[8] == 1                                                # the sliver is dead (:113-115)
    assert main[11] == 6 and tri[11] == 4                                              # five vertices + closing copy; triangle
    mx, my = rings[0][0] + (main[1] - T.Xi[0]), rings[0][1] + (main[2] - T.Yi[0])      # the main outline, back in floe coordinates
    got = sorted(zip(np.round(mx[:-1], 6).tolist(), np.round(my[:-1], 6).tolist()))
    want = sorted([(-5000.0, 5000.0), (5000.0, 5000.0), (5000.0, -4880.0), (4920.0, -5060.0), (-5000.0, -5060.0)])
    assert got == want                                                                 # C replaced by the cut point on the long edge; D kept
    # both ends of the short edge broken: not this closed form
    npieces, poff, pieces, rings = oracle.frac_corner(T, [1], [0, 5], np.array([0, 0, 1, 1, 0], np.uint8))
    assert npieces[0] == 0


def test_generated_outlines_have_no_collinear_vertices():
    """polyshape's default KeepCollinearPoints = false would drop a collinear vertex and shift the indices of `angles` and
    `grind` (corners.m:63-71): the restatement takes poly.Vertices to be the stored outline, which is only right when no
    stored vertex is collinear with its neighbours.  Checked for the outlines the scene generators produce."""
    import scenes
    for S, _ in (scenes.pack(400, periodic=True), scenes.concave(60), scenes.many_sided(64)):
        for i in range(S.n):
            a, b = int(S.vert_off[i]), int(S.vert_off[i + 1]) - 1          # without the closing copy
            x, y = S.c0x[a:b], S.c0y[a:b]
            px, py, nx, ny = np.roll(x, 1), np.roll(y, 1), np.roll(x, -1), np.roll(y, -1)
            cross = (x - px) * (ny - y) - (y - py) * (nx - x)
            scale = np.hypot(x - px, y - py) * np.hypot(nx - x, ny - y)
            assert (np.abs(cross) > 1e-9 * scale).all(), i
