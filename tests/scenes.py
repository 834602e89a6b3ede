"""Seeded scenes for the step parity tests (shared by the CPU and GPU suites)."""
import numpy as np
from subzero_amd import synth
import clipgen


def pack(n=400, periodic=True, seed=4, vel_seed=2, sigma=0.1, n_stress=16, **kw):
    S = synth.voronoi_field(n, 1.0, seed=seed, vel_seed=vel_seed, vel_sigma=sigma, n_stress=n_stress, **kw)
    P = synth.StepParams(dt=10, periodic=periodic)
    P.Modulus = synth.reference_modulus(S)
    return S, P


def concave(n=60, periodic=True, seed=11, n_stress=16):
    """Real floe outlines (reference fixture FloeShapes.mat) laid out on a jittered lattice so that
    neighbours overlap a little: exercises the non-convex branches of floe_interactions.m:117-137."""
    rng = np.random.default_rng(seed)
    shapes, _ = clipgen.load_shapes()
    shapes = [s for s in shapes if len(s) <= 90]
    side = int(np.ceil(np.sqrt(n)))
    polys = []
    pitch = 2e5 / side
    for k in range(n):
        s = shapes[rng.integers(len(shapes))]
        c = s.mean(0)
        r = np.sqrt(((s - c) ** 2).sum(1).max())
        sc = 0.56 * pitch / r
        th = rng.uniform(0, 2 * np.pi)
        rot = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        gx, gy = k % side, k // side
        ctr = np.array([-1e5 + (gx + 0.5) * pitch, -1e5 + (gy + 0.5) * pitch]) + rng.uniform(-0.05, 0.05, 2) * pitch
        polys.append((s - c) @ rot.T * sc + ctr)
    S = synth.build_state(polys, height=0.5, seed=seed, n_stress=n_stress)
    vr = np.random.default_rng(seed + 1)
    S.Ui[:] = vr.normal(0, 0.2, S.n); S.Vi[:] = vr.normal(0, 0.2, S.n)
    S.ksi_ice[:] = vr.normal(0, 2e-6, S.n)
    P = synth.StepParams(dt=10, periodic=periodic, height=0.5)
    P.Modulus = synth.reference_modulus(S)
    return S, P


def many_sided(n=64, periodic=True, seed=21, n_stress=16):
    """Convex floes with 8 to 39 vertices (irregular polygons inscribed in circles) on a jittered lattice,
    overlapping a little: rings of up to 16 vertices take the fast pass of the narrow phase (convex.hpp
    FastBlock), the longer ones are handed to the general sweep."""
    rng = np.random.default_rng(seed)
    side = int(np.ceil(np.sqrt(n)))
    pitch = 2e5 / side
    polys = []
    for k in range(n):
        m = int(rng.integers(8, 25)) if k % 7 else int(rng.integers(33, 40))
        ang = np.sort(rng.uniform(0, 2 * np.pi, m))
        ang = ang[np.concatenate([[True], np.diff(ang) > 0.02])]
        r = 0.53 * pitch * rng.uniform(0.9, 1.0)
        gx, gy = k % side, k // side
        ctr = np.array([-1e5 + (gx + 0.5) * pitch, -1e5 + (gy + 0.5) * pitch]) + rng.uniform(-0.03, 0.03, 2) * pitch
        polys.append(np.stack([r * np.cos(-ang), r * np.sin(-ang)], 1) + ctr)
    S = synth.build_state(polys, height=0.5, seed=seed, n_stress=n_stress)
    vr = np.random.default_rng(seed + 1)
    S.Ui[:] = vr.normal(0, 0.2, S.n); S.Vi[:] = vr.normal(0, 0.2, S.n)
    S.ksi_ice[:] = vr.normal(0, 2e-6, S.n)
    P = synth.StepParams(dt=10, periodic=periodic, height=0.5)
    P.Modulus = synth.reference_modulus(S)
    return S, P


def two_boxes(offset=0.0):
    """test/conservation_test.m:22-33: two 20 km boxes approaching head-on (optionally offset in y)."""
    h = 1e4
    b = np.array([[-h, -h], [-h, h], [h, h], [h, -h]], float)
    polys = [b + [-1e4 - 5, 0.0], b + [1e4 + 5, offset]]
    S = synth.build_state(polys, height=1.0, seed=5, n_stress=16)
    S.Ui[:] = [0.1, -0.1]
    P = synth.StepParams(dt=10, periodic=False, height=1.0)
    P.Modulus = 9e7  # test/test_conservation/Modulus.mat
    return S, P


def boxes_and_triangle():
    """test/conservation_test.m:36-41 geometry: two boxes and a triangle between them."""
    h = 1e4
    b = np.array([[-h, -h], [-h, h], [h, h], [h, -h]], float)
    t = np.array([[-8e3, -1e4], [0.0, 1.2e4], [8e3, -1e4]])
    polys = [b + [-1.8005e4, 0.0], t, b + [1.8005e4, 0.0]]
    S = synth.build_state(polys, height=1.0, seed=6, n_stress=16)
    S.Ui[:] = [0.15, 0.0, -0.15]
    P = synth.StepParams(dt=10, periodic=False, height=1.0)
    P.Modulus = 9e7
    return S, P


def baseline_config(name):
    """BASELINE.json configs 0-2 at test size (subzero_amd.synth.config): the out-of-box 10-floe case, 1000 convex
    floes at concentration 0.8 in a periodic box, and the uniaxial-compression case whose N/S walls move inward."""
    if name == "nares_standin":
        return nares_standin()
    S, P = synth.config(name, n_stress=8)
    if name == "convex1000":
        P.set_reference_ocean()
    return S, P


def nares_full():
    """The stand-in for BASELINE configs[4] at the size SURVEY 8(d).5 prescribes: 1e4 Voronoi floes in the
    2e5 x 1e6 m channel (initialize_boundaries.m:14-16), 12 static concave islands (Nb) with outlines of the
    reference's FloeShapes.mat, wind V = -10 m/s."""
    return nares_standin(n=11500, nb=12, seed=5)   # 10204 floes remain around the islands


def ghost_list(S, P):
    """floe_interactions_all.m:16-66 in plain loops: (src, parent, x, y) of the ghost floes in the order the reference
    appends them (1-based floe numbers over the extended list), for checking szGhostsDownload."""
    Lx, Ly = P.c2_boundary[0].max(), P.c2_boundary[1].max()
    ext = [(i, S.Xi[i], S.Yi[i]) for i in range(S.n)]
    out = []
    for dim, L in ((0, Lx), (1, Ly)):
        for e in range(len(ext)):
            i, x, y = ext[e]
            a, b = S.vert_off[i], S.vert_off[i + 1]
            m = np.abs((S.cax[a:b] + x) if dim == 0 else (S.cay[a:b] + y)).max()
            if S.alive[i] and m > L:
                gx = x - 2 * Lx * np.sign(x) if dim == 0 else x
                gy = y - 2 * Ly * np.sign(y) if dim == 1 else y
                ext.append((i, gx, gy))
                out.append((i + 1, e + 1, gx, gy))
    return out


def nares_standin(n=260, nb=3, seed=5):
    """BASELINE configs[4] at test size (the real case's files are absent from the reference checkout):
    a 2e5 x 1e6 m walled channel, `nb` static floes with outlines of the reference's FloeShapes.mat first in the
    array (Nb), Voronoi floes around them, wind V = -10 m/s (README.md:126), ocean at rest.  As shipped the candidate
    search (floe_interactions_all.m:76-120: i from 1+Nb, j > i) never pairs anything with the static floes."""
    Lx, Ly = 1e5, 5e5
    rng = np.random.default_rng(seed)
    shapes, _ = clipgen.load_shapes()
    shapes = [s for s in shapes if 30 <= len(s) <= 90]
    islands, centres, radii = [], [], []
    for k in range(nb):
        s = shapes[rng.integers(len(shapes))]
        c = s.mean(0); r = np.sqrt(((s - c) ** 2).sum(1).max())
        R = rng.uniform(1.5e4, 3e4)
        ctr = np.array([rng.uniform(-0.5, 0.5) * Lx, (-0.6 + 1.2 * (k + 0.5) / nb) * Ly])
        islands.append((s - c) * (R / r) + ctr); centres.append(ctr); radii.append(R)
    seeds = np.stack([rng.uniform(-Lx, Lx, n), rng.uniform(-Ly, Ly, n)], 1)
    cells = [v for v in synth.bounded_voronoi(seeds, Lx, Ly) if v is not None]
    keep = [v for v in cells if all(np.hypot(*(v.mean(0) - c)) > 1.15 * R for c, R in zip(centres, radii))]
    S = synth.build_state(islands + keep, height=0.5, seed=seed + 1000, nb=nb, n_stress=8)
    vr = np.random.default_rng(seed + 1)
    S.Ui[nb:] = vr.normal(0, 0.1, S.n - nb); S.Vi[nb:] = vr.normal(0, 0.1, S.n - nb)
    P = synth.StepParams(dt=10, Lx=Lx, Ly=Ly, periodic=False, height=0.5)
    P.Yo = np.arange(-6e5, 6e5 + 1, 1e4)
    shape = (len(P.Yo), len(P.Xo))
    P.Uocn = np.zeros(shape); P.Vocn = np.zeros(shape); P.Uwind = np.zeros(shape); P.Vwind = np.full(shape, -10.0)
    P.Modulus = synth.reference_modulus(S)
    return S, P


def squeeze_walls(P, step, every=3, by=15.0):
    """README.md:96-107 (validation case 1): the N/S walls move `by` metres inward every `every` steps."""
    if step > 0 and step % every == 0:
        Lx, Ly = P.c2_boundary[0].max(), P.c2_boundary[1].max() - by
        P.c2_boundary = np.array([[-Lx, -Lx, Lx, Lx, -Lx], [-Ly, Ly, Ly, -Ly, -Ly]], np.float64)
        P.fb_area = float(16 * Lx * Ly - 4 * Lx * Ly)


def random_scene(seed):
    """A small scene drawn from the seed: Voronoi pack (concentration 0.6-1), real concave outlines or many-sided
    convex floes; periodic or walled; 20-120 floes; random speeds, time step, topography count Nb and forcing.
    -> (FloeState, StepParams, number of steps)"""
    rng = np.random.default_rng(seed)
    kind = int(rng.integers(4))
    periodic = bool(rng.integers(2))
    n = int(rng.integers(20, 120))
    if kind == 0:
        S, P = pack(n, periodic=periodic, seed=seed, vel_seed=seed + 1, sigma=float(rng.uniform(0.05, 0.6)), n_stress=4, n_mc=200)
    elif kind == 1:
        S, P = concave(max(9, n // 3), periodic=periodic, seed=seed, n_stress=4)
    elif kind == 2:
        S, P = many_sided(max(9, n // 2), periodic=periodic, seed=seed, n_stress=4)
    else:
        S = synth.voronoi_field(n, float(rng.uniform(0.6, 1.0)), seed=seed, vel_seed=seed + 2,
                                vel_sigma=float(rng.uniform(0.05, 0.5)), n_stress=4, n_mc=200)
        P = synth.StepParams(dt=float(rng.choice([5, 10, 20])), periodic=periodic)
        P.Modulus = synth.reference_modulus(S)
    if rng.integers(2):
        P.set_reference_ocean()
    if rng.integers(4) == 0:
        S.nb = int(rng.integers(1, 3))
    return S, P, int(rng.integers(3, 9))


def kinetic_energy(S):
    return 0.5 * (S.mass * (S.Ui ** 2 + S.Vi ** 2)).sum() + 0.5 * (S.inertia_moment * S.ksi_ice ** 2).sum()


def edge_cases(periodic=False):
    """A handful of floes arranged to hit the rarely taken branches of floe_interactions_all.m /
    floe_interactions.m: a small floe swallowed by a big one (overlap > 55 % -> kill / transfer marks, :138-145),
    a floe hanging mostly outside the wall ('dif' area > 75 % -> Inf, floe_interactions.m:35-40), its centre
    outside the box (alive = 0, :152-155), a dead floe, a floe with NaN position, plus ordinary neighbours."""
    def box(cx, cy, hx, hy):
        return np.array([[cx - hx, cy - hy], [cx - hx, cy + hy], [cx + hx, cy + hy], [cx + hx, cy - hy]], float)
    polys = [
        box(-5e4, 0.0, 8e3, 8e3),            # 0 big
        box(-5e4 + 2e3, 1e3, 2.5e3, 2.5e3),  # 1 small, fully inside 0
        box(9.8e4, 3e4, 6e3, 6e3),           # 2 mostly inside the box, touching the east wall
        box(1.04e5, -3e4, 5e3, 5e3),         # 3 centre outside the east wall
        box(0.0, 5e4, 5e3, 5e3),             # 4 dead
        box(2e4, -5e4, 5e3, 5e3),            # 5 NaN position
        box(0.0, 0.0, 6e3, 6e3),             # 6 ordinary pair ...
        box(1.15e4, 1e3, 6e3, 6e3),          # 7 ... overlapping by 500 m
        box(3e4, 3e4, 4e3, 7e3),             # 8
        box(3.75e4, 3.2e4, 4e3, 4e3),        # 9
    ]
    S = synth.build_state(polys, height=0.8, seed=9, n_stress=8)
    S.Ui[:] = [0.0, 0.05, 0.1, 0.0, 0.0, 0.0, 0.1, -0.1, 0.05, -0.05]
    S.alive[4] = 0
    S.Xi[5] = np.nan
    P = synth.StepParams(dt=10, periodic=periodic, height=0.8)
    P.Modulus = 5e7
    return S, P


def big_among_small(n=3000, ratio=20.0, periodic=False, seed=31, centre=(0.0, 0.0), sides=48, n_stress=4, n_mc=200):
    """One floe `ratio` times the median radius in a pack of small ones: a `sides`-gon replaces the
    Voronoi cells under it and overlaps the ring of cells around it a little, so it has far more than 32 candidate
    partners and contacts.  With `periodic` and an off-centre position the big floe reaches across the box edge and
    gets a ghost."""
    S0 = synth.voronoi_field(n, 1.0, seed=seed, vel_seed=seed + 1, vel_sigma=0.1, n_stress=n_stress, n_mc=n_mc)
    R = ratio * float(np.median(S0.rmax))
    c = np.asarray(centre, float)
    polys = []
    for i in range(S0.n):
        x, y = S0.ring(i)
        if np.hypot(S0.Xi[i] - c[0], S0.Yi[i] - c[1]) > R * 0.985:
            polys.append(np.stack([x[:-1], y[:-1]], 1))
    ang = -np.linspace(0, 2 * np.pi, sides, endpoint=False) + 0.1
    big = np.stack([R * np.cos(ang), R * np.sin(ang)], 1) + c
    # the big floe sits in the middle of the array: it is both an `i` with many j > i and a `j` of many i < j
    k = len(polys) // 2
    polys = polys[:k] + [big] + polys[k:]
    S = synth.build_state(polys, height=0.25, seed=seed + 7, n_stress=n_stress, n_mc=n_mc)
    vr = np.random.default_rng(seed + 2)
    S.Ui[:] = vr.normal(0, 0.1, S.n); S.Vi[:] = vr.normal(0, 0.1, S.n)
    P = synth.StepParams(dt=10, periodic=periodic)
    P.Modulus = synth.reference_modulus(S)
    return S, P, k
