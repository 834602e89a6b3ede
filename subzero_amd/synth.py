"""Synthetic floe fields for benchmarks and tests (host-side set-up, numpy only; no GPU arithmetic).

Follows what the reference's initialisation produces (it is the spec of the Floe struct array, not part
of the accelerated path):
  Initialize_Model/initial_concentration.m   bounded Voronoi cells of uniform random seeds in the box
  Initialize_Model/initialize_floe_values.m:12-52   every Floe field
  polygon_operations/PolygonMoments.m:14-32  moment of inertia
  Initialize_Model/initialize_ocean.m        81 x 81 ocean grid, 10 km spacing
  Subzero.m:67-78                            floebound, Modulus
The SURVEY section 8(d) configurations are available by name through `config()`.
"""
import numpy as np
from scipy.spatial import Voronoi

RHO_ICE = 920.0
N_STRESS = 1000
N_MC = 1000


class FloeState:
    """Struct-of-arrays image of the reference's Floe struct array (field names as in
    initialize_floe_values.m).  Arrays are C-contiguous and owned by this object; `as_struct` packs
    pointers to them into the C struct both the HIP library and the oracle take."""

    SCALARS = ["Xi", "Yi", "alpha_i", "Ui", "Vi", "ksi_ice", "h", "area", "mass", "inertia_moment", "rmax",
               "dXi_p", "dYi_p", "dUi_p", "dVi_p", "dalpha_i_p", "dksi_ice_p", "FxOA", "FyOA", "torqueOA"]

    def __init__(self, n, vert_off, nb=0, n_stress=N_STRESS, n_mc=N_MC):
        self.n, self.nb, self.n_stress, self.n_mc = int(n), int(nb), int(n_stress), int(n_mc)
        for k in self.SCALARS:
            setattr(self, k, np.zeros(n, np.float64))
        self.alive = np.ones(n, np.int32)
        self.vert_off = np.ascontiguousarray(vert_off, np.int32)
        nv = int(self.vert_off[-1])
        self.c0x = np.zeros(nv); self.c0y = np.zeros(nv)
        self.cax = np.zeros(nv); self.cay = np.zeros(nv)
        self.StressH = np.zeros((n, n_stress, 4))
        self.StressCount = np.ones(n, np.int32)
        self.Stress = np.zeros((n, 4)); self.strain = np.zeros((n, 4))
        self.mcX = np.zeros((n, n_mc)); self.mcY = np.zeros((n, n_mc))
        self.mcA = np.zeros((n, n_mc), np.uint8)
        self.collision_force = np.zeros((n, 2)); self.collision_torque = np.zeros(n)
        self.OverlapArea = np.zeros(n)
        self.kill = np.zeros(4 * n, np.int32); self.transfer = np.zeros(4 * n, np.int32)

    ARRAYS = SCALARS + ["alive", "vert_off", "c0x", "c0y", "cax", "cay", "StressH", "StressCount", "Stress",
                        "strain", "mcX", "mcY", "mcA", "collision_force", "collision_torque", "OverlapArea",
                        "kill", "transfer"]

    def copy(self):
        o = FloeState(self.n, self.vert_off, self.nb, self.n_stress, self.n_mc)
        for k in self.ARRAYS:
            getattr(o, k)[...] = getattr(self, k)
        return o

    def ring(self, i, rotated=True):
        a, b = self.vert_off[i], self.vert_off[i + 1]
        if rotated:
            return self.cax[a:b] + self.Xi[i], self.cay[a:b] + self.Yi[i]
        return self.c0x[a:b], self.c0y[a:b]


class StepParams:
    """Arguments of floe_interactions_all.m:1 that the accelerated part reads."""

    def __init__(self, dt=10.0, Lx=1e5, Ly=1e5, periodic=True, collision=True, do_int=True, modulus=1.0,
                 hfo=None, mu=0.2, nu=0.3, height=0.25):
        self.dt = float(dt)
        # initialize_ocean.m:38-47
        k, Ta, To, L = 2.14, -20.0, 0.0, 2.93e5
        self.HFo = float(k * (Ta - To) / (RHO_ICE * L)) if hfo is None else float(hfo)
        self.Modulus = float(modulus)
        self.mu, self.nu = float(mu), float(nu)
        self.doInt, self.COLLISION, self.PERIODIC = int(do_int), int(collision), int(periodic)
        # initialize_boundaries.m:4-6 : x=[-1 -1 1 1 -1]*Lx; y=[-1 1 1 -1 -1]*Ly
        self.c2_boundary = np.array([[-Lx, -Lx, Lx, Lx, -Lx], [-Ly, Ly, Ly, -Ly, -Ly]], np.float64)
        # floebound = initialize_floe_values(2x box minus domain) (Subzero.m:67-69)
        self.fb_h = float(height)
        self.fb_area = float(16 * Lx * Ly - 4 * Lx * Ly)
        self.fb_Xi = self.fb_Yi = self.fb_Ui = self.fb_Vi = self.fb_ksi_ice = 0.0
        # initialize_ocean.m:4-30, ocean at rest unless set
        dXo = 1e4
        self.Xo = np.arange(-4e5, 4e5 + 1, dXo); self.Yo = np.arange(-4e5, 4e5 + 1, dXo)
        shape = (len(self.Yo), len(self.Xo))
        self.Uocn = np.zeros(shape); self.Vocn = np.zeros(shape)
        self.Uwind = np.zeros(shape); self.Vwind = np.zeros(shape)
        self.fCoriolis = 1.4e-4
        self.turn_angle = 15 * np.pi / 180

    def set_reference_ocean(self):
        """The eddy field of initialize_ocean.m:17-25."""
        Lx = 4e5; kx = np.pi / Lx; ky = np.pi / Lx; dXo = 1e4
        Xocn, Yocn = np.meshgrid(self.Xo, self.Yo)
        psi = 0.5e4 * (np.sin(4 * kx * Xocn) * np.sin(4 * ky * Yocn))
        self.Uocn[...] = 0; self.Vocn[...] = 0
        self.Uocn[1:, :] = -(psi[1:, :] - psi[:-1, :]) / dXo
        self.Vocn[:, 1:] = (psi[:, 1:] - psi[:, :-1]) / dXo


def _inpolygon(px, py, vx, vy):
    """Even-odd point-in-polygon for the Monte-Carlo mask A (initialize_floe_values.m:33)."""
    n = len(vx)
    inside = np.zeros(len(px), bool)
    j = n - 1
    for i in range(n):
        cond = ((vy[i] > py) != (vy[j] > py))
        with np.errstate(divide="ignore", invalid="ignore"):
            xint = (vx[j] - vx[i]) * (py - vy[i]) / (vy[j] - vy[i]) + vx[i]
        inside ^= cond & (px < xint)
        j = i
    return inside


def _polygon_moments(x, y, h):
    """PolygonMoments.m:14-32 on the closed ring c0' (N+1 rows, the function closes it once more)."""
    x = np.append(x, x[0]); y = np.append(y, y[0])
    wi = x[:-1] * y[1:] - x[1:] * y[:-1]
    Ixx = 1 / 12 * np.sum(wi * ((y[:-1] + y[1:]) ** 2 - y[:-1] * y[1:]))
    Iyy = 1 / 12 * np.sum(wi * ((x[:-1] + x[1:]) ** 2 - x[:-1] * x[1:]))
    return abs(Ixx + Iyy) * h * RHO_ICE


def bounded_voronoi(seeds, Lx, Ly):
    """Voronoi cells of `seeds` clipped to the box [-Lx,Lx]x[-Ly,Ly] by mirroring the seeds in the four
    walls (same cells as polybnd_voronoi.m gives for a rectangular boundary)."""
    pts = [seeds]
    for sx, sy, ox, oy in [(-1, 1, -2 * Lx, 0), (-1, 1, 2 * Lx, 0), (1, -1, 0, -2 * Ly), (1, -1, 0, 2 * Ly)]:
        pts.append(np.stack([sx * seeds[:, 0] + ox, sy * seeds[:, 1] + oy], 1))
    vor = Voronoi(np.concatenate(pts))
    cells = []
    for i in range(len(seeds)):
        reg = vor.regions[vor.point_region[i]]
        if -1 in reg or len(reg) < 3:
            cells.append(None); continue
        v = vor.vertices[reg]
        c = v.mean(0)
        ang = np.arctan2(v[:, 1] - c[1], v[:, 0] - c[0])
        v = v[np.argsort(-ang)]  # clockwise, as polyshape stores outer boundaries
        cells.append(v)
    return cells


def build_state(polys, height=0.25, seed=0, nb=0, n_stress=N_STRESS, n_mc=N_MC):
    """Floe fields from a list of open vertex rings (each (n,2)), as initialize_floe_values.m does."""
    rng = np.random.default_rng(seed)
    polys = [p for p in polys if p is not None and len(p) >= 3]
    vert_off = np.zeros(len(polys) + 1, np.int32)
    for i, p in enumerate(polys):
        vert_off[i + 1] = vert_off[i] + len(p) + 1  # closed ring: first vertex repeated (:17)
    S = FloeState(len(polys), vert_off, nb, n_stress, n_mc)
    for i, p in enumerate(polys):
        x, y = p[:, 0], p[:, 1]
        x2, y2 = np.roll(x, -1), np.roll(y, -1)
        w = x * y2 - x2 * y
        A = 0.5 * w.sum()
        cx, cy = ((x + x2) * w).sum() / (6 * A), ((y + y2) * w).sum() / (6 * A)
        a, b = vert_off[i], vert_off[i + 1]
        S.c0x[a:b - 1] = x - cx; S.c0y[a:b - 1] = y - cy
        S.c0x[b - 1] = x[0] - cx; S.c0y[b - 1] = y[0] - cy
        S.Xi[i], S.Yi[i] = cx, cy
        S.area[i] = abs(A)
        S.h[i] = height
        S.mass[i] = S.area[i] * height * RHO_ICE
        S.inertia_moment[i] = _polygon_moments(S.c0x[a:b], S.c0y[a:b], height)
        S.rmax[i] = np.sqrt(((x - cx) ** 2 + (y - cy) ** 2).max())
        # Monte-Carlo points (:28-36)
        for _ in range(10):
            mx = S.rmax[i] * (2 * rng.random(n_mc) - 1); my = S.rmax[i] * (2 * rng.random(n_mc) - 1)
            inside = _inpolygon(mx, my, S.c0x[a:b], S.c0y[a:b])
            err = abs(inside.sum() / n_mc * 4 * S.rmax[i] ** 2 - S.area[i]) / S.area[i]
            if err <= 0.1:
                break
        S.mcX[i], S.mcY[i], S.mcA[i] = mx, my, inside
    S.cax[:] = S.c0x; S.cay[:] = S.c0y
    return S


def reference_modulus(S):
    """Subzero.m:76-77"""
    r = np.sqrt(S.area[S.alive > 0])
    return 1.5e3 * (r.mean() + r.min())


def voronoi_field(n_floes, concentration=1.0, Lx=1e5, Ly=1e5, seed=1, vel_seed=2, vel_sigma=0.1, height=0.25,
                  n_stress=N_STRESS, n_mc=N_MC):
    """n_floes Voronoi cells of uniform seeds in the box, each shrunk about its centroid by
    sqrt(concentration), random initial velocities N(0, vel_sigma) so that contacts occur."""
    rng = np.random.default_rng(seed)
    seeds = np.stack([rng.uniform(-Lx, Lx, n_floes), rng.uniform(-Ly, Ly, n_floes)], 1)
    cells = bounded_voronoi(seeds, Lx, Ly)
    f = np.sqrt(concentration)
    polys = []
    for v in cells:
        if v is None:
            continue
        c = v.mean(0)
        polys.append(c + f * (v - c))
    S = build_state(polys, height=height, seed=seed + 1000, n_stress=n_stress, n_mc=n_mc)
    vr = np.random.default_rng(vel_seed)
    S.Ui[:] = vr.normal(0, vel_sigma, S.n); S.Vi[:] = vr.normal(0, vel_sigma, S.n)
    return S


def tile_field(S0, copies, Lx):
    """`copies` images of the field S0 (box half-width Lx) side by side along x, numbered copy by copy: floe i of copy c
    is floe c * S0.n + i at x + (2 c - (copies - 1)) Lx.  The weak-scaling workload of the multi-GPU bench: every GPU's
    share is the one-GPU field, and -- like any field whose floes are numbered strip by strip -- the floes a rank holds
    are a run of consecutive numbers."""
    n0, nv0 = S0.n, int(S0.vert_off[-1])
    vo = np.concatenate([S0.vert_off[:-1].astype(np.int64) + c * nv0 for c in range(copies)] + [np.array([copies * nv0])]).astype(np.int32)
    S = FloeState(copies * n0, vo, S0.nb, S0.n_stress, S0.n_mc)
    for k in FloeState.ARRAYS:
        if k in ("vert_off", "kill", "transfer"):
            continue
        src = getattr(S0, k)
        rows = nv0 if k in ("c0x", "c0y", "cax", "cay") else n0
        dst = getattr(S, k)
        for c in range(copies):
            dst[c * rows:(c + 1) * rows] = src
    for c in range(copies):
        S.Xi[c * n0:(c + 1) * n0] += (2 * c - (copies - 1)) * Lx
    return S


def config(name, n_stress=N_STRESS, n_mc=N_MC):
    """The BASELINE.json configurations as (FloeState, StepParams), SURVEY.md section 8(d)."""
    if name == "plumbing10":      # configs[0]: ~10 floes, full pack, dt = 10 s, non-periodic (out-of-box Subzero.m)
        S = voronoi_field(10, 1.0, seed=10, vel_sigma=0.0, n_stress=n_stress, n_mc=n_mc)
        P = StepParams(dt=10, periodic=False)
        P.set_reference_ocean()
    elif name == "convex1000":    # configs[1]: 1000 convex floes, concentration 0.8, periodic, collisions only
        S = voronoi_field(1000, 0.8, seed=1, vel_seed=2, n_stress=n_stress, n_mc=n_mc)
        P = StepParams(dt=10, periodic=True)
    elif name == "uniaxial200":   # configs[2]: 200 floes full pack, h = 1, dt = 5, walls, mu = 0.3
        S = voronoi_field(200, 1.0, seed=3, vel_sigma=0.0, height=1.0, n_stress=n_stress, n_mc=n_mc)
        P = StepParams(dt=5, periodic=False, mu=0.3, height=1.0)
    elif name == "pack10000":     # configs[3]: 1e4 floes, concentration 1.0, periodic
        S = voronoi_field(10000, 1.0, seed=4, vel_seed=2, n_stress=n_stress, n_mc=n_mc)
        P = StepParams(dt=10, periodic=True)
    elif name == "nares_standin":
        # configs[4] stand-in (the real case needs files that are absent from the reference checkout):
        # a 2e5 x 1e6 m channel with walls (initialize_boundaries.m:14-16), Nb static concave floes first in the
        # array (outlines from the reference's FloeShapes.mat when the caller passes them, see tests/scenes.py),
        # Voronoi floes around them, wind V = -10 m/s (README.md:126).  As shipped, the candidate search
        # (floe_interactions_all.m:76-120) starts at 1+Nb and only looks at j > i: nothing is ever paired with the
        # static floes.  Reproduced as shipped.
        raise ValueError("nares_standin needs the fixture outlines: use tests/scenes.py baseline_config")
    else:
        raise ValueError(name)
    P.Modulus = reference_modulus(S)
    if name == "uniaxial200":
        r = np.sqrt(S.area)
        P.Modulus = 2.5e3 * (r.mean() + r.min())  # README.md:90
    return S, P
