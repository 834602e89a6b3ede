"""Seeded generators of ring pairs for the clip parity tests, plus ctypes bindings to the two CPU
clip libraries used as checkers (TEST INFRASTRUCTURE):

  oracle/_ref/libclipper_ref.so      the reference's own Clipper 6.4.2 (built by oracle/Makefile)
  tests/hostsim/libvatti_host.so     the device sweep engine compiled for the host with g++
"""
import ctypes
import os
import subprocess
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCALE = float(2 ** 32)

_i64p = np.ctypeslib.ndpointer(np.int64, flags="C")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C")


def matlab_int64(v):
    """MATLAB int64(double): round half away from zero, saturate (polyclip.m:66)."""
    v = np.asarray(v, np.float64)
    t = np.trunc(v)
    frac = v - t
    r = t + np.where(np.abs(frac) >= 0.5, np.sign(v), 0.0)
    r = np.clip(r, -9.223372036854775808e18, 9.223372036854775807e18)
    out = np.empty(r.shape, np.int64)
    big = r >= 9.223372036854775807e18
    out[~big] = r[~big].astype(np.int64)
    out[big] = np.iinfo(np.int64).max
    return out


def to_fixed(x):
    return matlab_int64(np.asarray(x, np.float64) * SCALE)


class PairSet:
    """CSR of ring pairs on int64 coordinates."""

    def __init__(self):
        self.ax, self.ay, self.aoff = [], [], [0]
        self.bx, self.by, self.boff = [], [], [0]

    def add(self, a, b, fixed=False):
        a = np.asarray(a); b = np.asarray(b)
        if not fixed:
            a = np.stack([to_fixed(a[:, 0]), to_fixed(a[:, 1])], 1)
            b = np.stack([to_fixed(b[:, 0]), to_fixed(b[:, 1])], 1)
        self.ax.append(a[:, 0].astype(np.int64)); self.ay.append(a[:, 1].astype(np.int64))
        self.bx.append(b[:, 0].astype(np.int64)); self.by.append(b[:, 1].astype(np.int64))
        self.aoff.append(self.aoff[-1] + len(a)); self.boff.append(self.boff[-1] + len(b))

    def arrays(self):
        cat = lambda l: np.ascontiguousarray(np.concatenate(l)) if l else np.zeros(0, np.int64)
        return (cat(self.ax), cat(self.ay), np.asarray(self.aoff, np.int32),
                cat(self.bx), cat(self.by), np.asarray(self.boff, np.int32))

    def __len__(self):
        return len(self.aoff) - 1


def load_shapes():
    d = np.load(os.path.join(ROOT, "tests", "golden", "floe_shapes.npz"))
    off = d["off"]
    return [np.stack([d["x"][off[i]:off[i + 1]], d["y"][off[i]:off[i + 1]]], 1) for i in range(len(off) - 1)], d["polygon_info"]


def _rot(p, th):
    c, s = np.cos(th), np.sin(th)
    return p @ np.array([[c, s], [-s, c]])


def shape_pairs(n, seed=0, max_verts=None):
    """Pairs of real floe outlines (reference fixture FloeShapes.mat), second one rotated and moved
    so the two overlap partially, at floe-scale coordinates (1e4..1e5 m)."""
    rng = np.random.default_rng(seed)
    shapes, _ = load_shapes()
    if max_verts:
        shapes = [s for s in shapes if len(s) <= max_verts]
    ps = PairSet()
    for _ in range(n):
        a = shapes[rng.integers(len(shapes))]
        b = shapes[rng.integers(len(shapes))]
        ca, cb = a.mean(0), b.mean(0)
        ra = np.sqrt(((a - ca) ** 2).sum(1).max()); rb = np.sqrt(((b - cb) ** 2).sum(1).max())
        th = rng.uniform(0, 2 * np.pi)
        d = rng.uniform(0.3, 1.1) * (ra + rb)
        ang = rng.uniform(0, 2 * np.pi)
        b2 = _rot(b - cb, th) + ca + d * np.array([np.cos(ang), np.sin(ang)])
        # closed rings, as floe_interactions.m passes them (first vertex repeated)
        ps.add(np.vstack([a, a[:1]]), np.vstack([b2, b2[:1]]))
    return ps


def cut_pairs(n, seed=0):
    """A real floe outline cut in two by a chord (as a fracture leaves it), one piece moved and turned a little:
    long near-coincident edges, T-junctions, crossings at grazing angles."""
    rng = np.random.default_rng(seed)
    shapes, _ = load_shapes()
    shapes = [s for s in shapes if 20 <= len(s) <= 90]
    ps = PairSet()
    while len(ps) < n:
        s = shapes[rng.integers(len(shapes))]
        c = s.mean(0); th = rng.uniform(0, np.pi)
        nrm = np.array([np.cos(th), np.sin(th)])
        d = (s - c) @ nrm - rng.normal(0, 0.1) * np.sqrt(((s - c) ** 2).sum(1).max())
        sgn = d > 0
        flips = np.nonzero(sgn != np.roll(sgn, -1))[0]
        if len(flips) != 2:
            continue
        i, j = flips

        def on_chord(k):
            a, b = s[k], s[(k + 1) % len(s)]
            return a + d[k] / (d[k] - d[(k + 1) % len(s)]) * (b - a)
        pi_, pj = on_chord(i), on_chord(j)
        A = np.vstack([s[i + 1:j + 1], pj, pi_])
        B = np.vstack([s[j + 1:], s[:i + 1], pi_, pj])
        if len(A) < 4 or len(B) < 4:
            continue
        move = rng.normal(0, 1, 2) * 10.0 ** rng.uniform(-4, 1.5)
        turn = rng.normal(0, 1) * 10.0 ** rng.uniform(-7, -3)
        if rng.random() < 0.1:
            move[:] = 0; turn = 0          # still joined: shared vertices, coincident edges
        cb = B.mean(0)
        B2 = _rot(B - cb, turn) + cb + move
        ps.add(np.vstack([A, A[:1]]), np.vstack([B2, B2[:1]]))
    return ps


def snapped_shape_pairs(n, seed=0, grid=400, mode=0):
    """shape_pairs with coordinates snapped to a coarse lattice (metres): mode 0 both axes (horizontal edges, spikes,
    repeated vertices removed), 1 x only (vertical coincidences), 2 both with every other vertex lifted by a quarter
    cell (shared scanlines, T-junctions, no horizontal edges)."""
    src = shape_pairs(n, seed=seed, max_verts=90)
    q = PairSet()
    g = np.int64(grid) << 32

    def snap(x, y):
        x = (x // g) * g
        if mode == 0:
            y = (y // g) * g
        elif mode == 2:
            y = (y // g) * g + (np.arange(len(y)) % 2) * (g // 4)
        k = np.ones(len(x), bool); k[1:] = (x[1:] != x[:-1]) | (y[1:] != y[:-1])
        return np.stack([x[k], y[k]], 1)
    for p in range(len(src)):
        q.add(snap(src.ax[p], src.ay[p]), snap(src.bx[p], src.by[p]), fixed=True)
    return q


def convex_pairs(n, seed=0):
    """Random convex polygons (6-9 vertices, km scale) like fresh Voronoi floes, overlapping a little."""
    rng = np.random.default_rng(seed)
    ps = PairSet()
    for _ in range(n):
        polys = []
        for _k in range(2):
            m = rng.integers(4, 10)
            ang = np.sort(rng.uniform(0, 2 * np.pi, m))
            r = rng.uniform(800, 1500)
            polys.append(np.stack([r * np.cos(ang), r * np.sin(ang)], 1))
        c = rng.uniform(-9e4, 9e4, 2)
        d = rng.uniform(0.6, 2.1) * 1200
        a2 = rng.uniform(0, 2 * np.pi)
        a = polys[0] + c
        b = polys[1] + c + d * np.array([np.cos(a2), np.sin(a2)])
        ps.add(np.vstack([a, a[:1]]), np.vstack([b, b[:1]]))
    return ps


def _hull(pts):
    """Convex hull (Andrew monotone chain) of integer points, strictly convex, counter-clockwise."""
    pts = sorted(set(map(tuple, pts)))
    if len(pts) < 3:
        return None
    def half(seq):
        h = []
        for p in seq:
            while len(h) >= 2 and (h[-1][0] - h[-2][0]) * (p[1] - h[-2][1]) - (h[-1][1] - h[-2][1]) * (p[0] - h[-2][0]) <= 0:
                h.pop()
            h.append(p)
        return h
    lo, up = half(pts), half(pts[::-1])
    h = lo[:-1] + up[:-1]
    return np.array(h, np.int64) if len(h) >= 3 else None


def hull_pairs(n, seed=0, grid=24, unit=1 << 32):
    """Convex hulls of a few random lattice points: convex rings with many exact ties (vertices on one
    scanline, vertices on edges, parallel and shared edges) -- the bail-out conditions of convex.hpp."""
    rng = np.random.default_rng(seed)
    ps = PairSet()
    while len(ps) < n:
        hs = []
        for _k in range(2):
            h = _hull(rng.integers(0, grid + 1, (rng.integers(3, 9), 2)))
            if h is not None and rng.random() < 0.5:
                h = h[::-1]
            hs.append(h)
        if hs[0] is None or hs[1] is None:
            continue
        k0, k1 = rng.integers(len(hs[0])), rng.integers(len(hs[1]))
        a, b = np.roll(hs[0], k0, 0) * unit, np.roll(hs[1], k1, 0) * unit
        if rng.random() < 0.5:
            a = np.vstack([a, a[:1]]); b = np.vstack([b, b[:1]])
        ps.add(a, b, fixed=True)
    return ps


def neighbour_pairs(n, seed=0):
    """Convex floes that share an edge exactly (Voronoi neighbours at concentration 1), the second one
    then nudged by a rigid motion between 1e-12 and 1e-1 of its size: from coincident to plainly crossing."""
    rng = np.random.default_rng(seed)
    ps = PairSet()
    while len(ps) < n:
        c = rng.uniform(-9e4, 9e4, 2)
        m = rng.integers(8, 14)
        pts = rng.uniform(-1500, 1500, (m, 2))
        h = _hull(np.round(pts * 1e3).astype(np.int64))
        if h is None or len(h) < 4 or len(h) > 9:
            continue
        a = h / 1e3
        i = rng.integers(len(a))
        p0, p1 = a[i], a[(i + 1) % len(a)]
        e = (p1 - p0) / np.linalg.norm(p1 - p0)
        nrm = np.array([e[1], -e[0]])                  # outward for a counter-clockwise ring
        # mirror image of a across the shared edge, so the two rings share p0-p1
        rel = a - p0
        b = p0 + rel - 2 * np.outer(rel @ nrm, nrm)
        eps = 10.0 ** rng.uniform(-12, -1) * 1500
        th = rng.normal() * eps / 1500
        b = _rot(b - b.mean(0), th) + b.mean(0) + rng.normal(size=2) * eps
        if rng.random() < 0.5:
            b = b[::-1]
        ps.add(np.vstack([a + c, a[:1] + c]), np.vstack([b + c, b[:1] + c]))
    return ps


def monotone_pairs(n, seed=0):
    """Rings with one local minimum and one maximum in y but concave sides (two zig-zag chains):
    the sweep holds four active edges, yet the overlap may have several rings -- the fast path must bail."""
    rng = np.random.default_rng(seed)
    ps = PairSet()

    def mono():
        nl, nr = rng.integers(1, 4), rng.integers(1, 4)
        ys = np.sort(rng.uniform(-1000, 1000, nl + nr + 2))
        inner = rng.permutation(np.arange(1, nl + nr + 1))
        yl, yr = np.sort(ys[inner[:nl]]), np.sort(ys[inner[nl:]])
        xl, xr = rng.uniform(-1200, -100, nl), rng.uniform(100, 1200, nr)
        bot, top = [rng.uniform(-300, 300), ys[0]], [rng.uniform(-300, 300), ys[-1]]
        ring = [bot] + [[x, y] for x, y in zip(xr, yr)] + [top] + [[x, y] for x, y in zip(xl[::-1], yl[::-1])]
        r = np.array(ring)
        return r[::-1] if rng.random() < 0.5 else r

    for _ in range(n):
        a, b = mono(), mono()
        c = rng.uniform(-9e4, 9e4, 2)
        d = rng.uniform(0, 1800); ang = rng.uniform(0, 2 * np.pi)
        b = _rot(b, rng.uniform(-0.6, 0.6)) + d * np.array([np.cos(ang), np.sin(ang)])
        ps.add(np.vstack([a, a[:1]]) + c, np.vstack([b, b[:1]]) + c)
    return ps


def star_pairs(n, seed=0, sizes=(40, 150, 300, 600, 1400), apart_every=5):
    """Pairs of long star-shaped outlines (radius noise around a circle: simple polygons in general position) that
    overlap in part, every `apart_every`-th pair moved until the boxes are apart: the sizes are chosen to fall into
    every slice class of the clip kernel's sweep pass, the HBM one included (clip.hip)."""
    rng = np.random.default_rng(seed)
    ps = PairSet()

    def star(m):
        th = np.sort(rng.uniform(0, 2 * np.pi, m))
        r = 1.0 + 0.15 * np.sin(3 * th + rng.uniform(0, 6)) + rng.uniform(-0.04, 0.04, m)
        return np.stack([r * np.cos(th), r * np.sin(th)], 1) * rng.uniform(0.5, 2.0)

    for k in range(n):
        a = star(int(sizes[k % len(sizes)] * rng.uniform(0.5, 1.0)))
        b = star(int(sizes[(k // 2) % len(sizes)] * rng.uniform(0.5, 1.0)))
        d = rng.uniform(0.4, 1.6) * _rot(np.array([[1.0, 0.0]]), rng.uniform(0, 2 * np.pi))[0]
        if apart_every and k % apart_every == apart_every - 1:
            d = np.array([a[:, 0].max() - b[:, 0].min() + rng.uniform(0.01, 1.0), rng.uniform(-1, 1)])
            if k % 2:
                d = np.array([rng.uniform(-1, 1), a[:, 1].min() - b[:, 1].max() - rng.uniform(0.01, 1.0)])
        ps.add(a, b + d)
    return ps


def lattice_pairs(n, seed=0, grid=6, max_verts=9):
    """Small polygons on a coarse integer lattice: shared edges, horizontal edges, touching vertices,
    collinear runs, self-intersections -- the degenerate paths of the sweep (joins, horizontals)."""
    rng = np.random.default_rng(seed)
    ps = PairSet()
    unit = 1 << 32
    for _ in range(n):
        polys = []
        for _k in range(2):
            m = rng.integers(3, max_verts + 1)
            p = rng.integers(0, grid + 1, (m, 2)).astype(np.int64) * unit
            polys.append(p)
        ps.add(polys[0], polys[1], fixed=True)
    return ps


def rect_pairs(n, seed=0, grid=8):
    """Axis-aligned rectangles and L-shapes on a lattice: every edge horizontal or vertical."""
    rng = np.random.default_rng(seed)
    ps = PairSet()
    unit = 1 << 32

    def rect():
        x0, x1 = np.sort(rng.choice(grid + 1, 2, replace=False))
        y0, y1 = np.sort(rng.choice(grid + 1, 2, replace=False))
        p = np.array([[x0, y0], [x1, y0], [x1, y1], [x0, y1]], np.int64)
        if rng.random() < 0.5:
            p = p[::-1]
        if rng.random() < 0.4 and x1 - x0 >= 2 and y1 - y0 >= 2:
            xm = rng.integers(x0 + 1, x1); ym = rng.integers(y0 + 1, y1)
            p = np.array([[x0, y0], [x1, y0], [x1, ym], [xm, ym], [xm, y1], [x0, y1]], np.int64)
        k = rng.integers(len(p))
        return np.roll(p, k, 0) * unit

    for _ in range(n):
        ps.add(rect(), rect(), fixed=True)
    return ps


def _build_if_needed(path, cmd):
    if not os.path.exists(path):
        subprocess.check_call(cmd, cwd=ROOT, stdout=sys.stderr)


def ref_lib():
    path = os.path.join(ROOT, "oracle", "_ref", "libclipper_ref.so")
    if not os.path.exists(path):
        if not os.path.isdir("/root/reference"):
            return None
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=sys.stderr)
    L = ctypes.CDLL(path)
    L.ref_clip_batch.argtypes = [ctypes.c_int, _i64p, _i64p, _i32p, _i64p, _i64p, _i32p, ctypes.c_int,
                                 _i64p, _i64p, ctypes.c_int64, _i32p, ctypes.c_int64, _i32p, _i32p]
    return L


def host_lib():
    path = os.path.join(ROOT, "tests", "hostsim", "libvatti_host.so")
    src = os.path.join(ROOT, "tests", "hostsim", "vatti_host.cpp")
    hdrs = [os.path.join(ROOT, "subzero_amd", "csrc", h) for h in ("vatti.hpp", "convex.hpp", "trim.hpp", "offset.hpp")]
    if (not os.path.exists(path)) or os.path.getmtime(path) < max([os.path.getmtime(src)] + [os.path.getmtime(h) for h in hdrs]):
        subprocess.check_call(["g++", "-O2", "-std=c++14", "-fPIC", "-shared", "-ffp-contract=off",
                               src, "-o", path])
    L = ctypes.CDLL(path)
    L.host_clip_batch.argtypes = [ctypes.c_int, _i64p, _i64p, _i32p, _i64p, _i64p, _i32p, ctypes.c_int,
                                  ctypes.c_int, ctypes.c_int,
                                  _i64p, _i64p, ctypes.c_int64, _i32p, ctypes.c_int64, _i32p, _i32p]
    L.host_clip_fast_count.restype = ctypes.c_longlong
    L.host_clip_compact_count.restype = ctypes.c_longlong
    return L


def team_lib():
    """tests/hostsim/libteam_clip_host.so: the team form of the fast intersection (teamclip.hpp), lanes as fibers."""
    path = os.path.join(ROOT, "tests", "hostsim", "libteam_clip_host.so")
    srcs = [os.path.join(ROOT, "tests", "hostsim", f) for f in ("team_clip_host.cpp", "team_host.hpp")]
    hdrs = [os.path.join(ROOT, "subzero_amd", "csrc", h) for h in ("vatti.hpp", "convex.hpp", "team.hpp", "teamclip.hpp")]
    if (not os.path.exists(path)) or os.path.getmtime(path) < max(os.path.getmtime(f) for f in srcs + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++14", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                               srcs[0], "-o", path])
    L = ctypes.CDLL(path)
    L.host_team_clip_batch.argtypes = [ctypes.c_int, ctypes.c_int, _i64p, _i64p, _i32p, _i64p, _i64p, _i32p,
                                       _i64p, _i64p, ctypes.c_int64, _i32p, ctypes.c_int64, _i32p, _i32p]
    return L


class ClipResult:
    def __init__(self, x, y, ring_off, pair_ring_off, status):
        self.x, self.y, self.ring_off, self.pair_ring_off, self.status = x, y, ring_off, pair_ring_off, status

    def rings(self, p):
        out = []
        for r in range(self.pair_ring_off[p], self.pair_ring_off[p + 1]):
            a, b = self.ring_off[r], self.ring_off[r + 1]
            out.append(np.stack([self.x[a:b], self.y[a:b]], 1))
        return out


def _alloc(ps):
    ax, ay, aoff, bx, by, boff = ps.arrays()
    n = len(ps)
    cap_v = int(16 * (len(ax) + len(bx)) + 64 * n + 64)
    cap_r = int(4 * (len(ax) + len(bx)) + 8 * n + 8)
    return (ax, ay, aoff, bx, by, boff, np.zeros(cap_v, np.int64), np.zeros(cap_v, np.int64), cap_v,
            np.zeros(cap_r + 1, np.int32), cap_r, np.zeros(n + 1, np.int32), np.zeros(n, np.int32))


def run_ref(ps, method):
    L = ref_lib()
    ax, ay, aoff, bx, by, boff, ox, oy, cap_v, roff, cap_r, proff, st = _alloc(ps)
    rc = L.ref_clip_batch(len(ps), ax, ay, aoff, bx, by, boff, method, ox, oy, cap_v, roff, cap_r, proff, st)
    assert rc == 0, rc
    return ClipResult(ox, oy, roff, proff, st)


def run_host(ps, method, fill_s=0, fill_c=0, fast=0, trim=0, budget=0):
    """fast = 1 / 2: try the single-minimum fast path (convex.hpp) first, from int64 rings / from rings in metres
    as the step hands them over; what it declines goes to the general sweep.  r.fast = clips it completed.
    trim = 1: the rings of an intersection are first cut down to their common box (trim.hpp), as the step's general
    path does; r.trim = (vertices before, after, pairs handed over untrimmed because of a tie).
    budget = bytes: the sweep's records inside that many bytes first (caps_budget: the slice classes of the clip
    batch), in a buffer of exactly that size; r.budget = (clips completed there, clips that outgrew it)."""
    L = host_lib()
    L.host_clip_set_fast(int(fast))
    L.host_clip_set_trim(int(trim))
    L.host_clip_set_budget(int(budget))
    ax, ay, aoff, bx, by, boff, ox, oy, cap_v, roff, cap_r, proff, st = _alloc(ps)
    rc = L.host_clip_batch(len(ps), ax, ay, aoff, bx, by, boff, method, fill_s, fill_c,
                           ox, oy, cap_v, roff, cap_r, proff, st)
    assert rc == 0, rc
    r = ClipResult(ox, oy, roff, proff, st)
    r.fast = int(L.host_clip_fast_count())
    c = [ctypes.c_longlong() for _ in range(3)]
    L.host_clip_trim_counts(*[ctypes.byref(v) for v in c])
    r.trim = tuple(int(v.value) for v in c)
    c = [ctypes.c_longlong() for _ in range(2)]
    L.host_clip_budget_counts(*[ctypes.byref(v) for v in c])
    r.budget = tuple(int(v.value) for v in c)
    L.host_clip_set_fast(0)
    L.host_clip_set_trim(0)
    L.host_clip_set_budget(0)
    return r


def run_team(ps, G=16):
    """The intersection by a team of G lanes (teamclip.hpp on the host, lanes as fibers): status 100 = bailed out."""
    L = team_lib()
    ax, ay, aoff, bx, by, boff, ox, oy, cap_v, roff, cap_r, proff, st = _alloc(ps)
    rc = L.host_team_clip_batch(G, len(ps), ax, ay, aoff, bx, by, boff, ox, oy, cap_v, roff, cap_r, proff, st)
    assert rc == 0, rc
    r = ClipResult(ox, oy, roff, proff, st)
    sites = (ctypes.c_longlong * 128)()
    L.host_team_bail_sites(sites)
    r.bail_sites = {i: int(sites[i]) for i in range(128) if sites[i]}
    return r


def compare_where_done(ref, team, n):
    """compare() on the pairs the team form completed (status != 100); -> (bad, completed)."""
    bad, done = [], 0
    for p in range(n):
        if team.status[p] == 100:
            continue
        done += 1
        if ref.status[p] != team.status[p]:
            bad.append((p, "status", int(ref.status[p]), int(team.status[p]))); continue
        a, b = ref.rings(p), team.rings(p)
        if len(a) != len(b):
            bad.append((p, "nrings", len(a), len(b))); continue
        for k in range(len(a)):
            if a[k].shape != b[k].shape or not np.array_equal(a[k], b[k]):
                bad.append((p, "ring%d" % k, a[k].shape, b[k].shape)); break
    return bad, done


def compare(r1, r2, n):
    """Bit-exact comparison: status, ring count, ring order, vertex order.  Returns list of bad pairs."""
    bad = []
    for p in range(n):
        if r1.status[p] != r2.status[p]:
            bad.append((p, "status", int(r1.status[p]), int(r2.status[p]))); continue
        a, b = r1.rings(p), r2.rings(p)
        if len(a) != len(b):
            bad.append((p, "nrings", len(a), len(b))); continue
        for k in range(len(a)):
            if a[k].shape != b[k].shape or not np.array_equal(a[k], b[k]):
                bad.append((p, "ring%d" % k, a[k].shape, b[k].shape)); break
    return bad


GOLDEN_SETS = ["shapes", "convex", "lattice", "rect", "clipper_test_m", "clipper_demo"]
GOLDEN_FILES = {"clipper_demo": "clipper_demo_golden.npz"}     # (the input of Clipper's own console demo, as the reference ships it)


def load_golden(name):
    """(PairSet-like arrays, {method: ClipResult}) from tests/golden/clip_golden.npz (clipper_demo_golden.npz)."""
    d = np.load(os.path.join(ROOT, "tests", "golden", GOLDEN_FILES.get(name, "clip_golden.npz")))
    inp = tuple(np.ascontiguousarray(d["%s/%s" % (name, k)]) for k in ("ax", "ay", "aoff", "bx", "by", "boff"))
    res = {}
    for m in range(4):
        res[m] = ClipResult(d["%s/m%d/x" % (name, m)], d["%s/m%d/y" % (name, m)], d["%s/m%d/ring_off" % (name, m)],
                            d["%s/m%d/pair_ring_off" % (name, m)], d["%s/m%d/status" % (name, m)])
    return inp, res


class ArraysPairSet:
    """Adapter so arrays loaded from a fixture can be passed where a PairSet is expected."""

    def __init__(self, arrays):
        self._a = arrays

    def arrays(self):
        return self._a

    def __len__(self):
        return len(self._a[2]) - 1


def run_gpu(ctx, ps, method, fill_s=0, fill_c=0):
    ax, ay, aoff, bx, by, boff = ps.arrays()
    ox, oy, roff, proff, st = ctx.clip_batch(ax, ay, aoff, bx, by, boff, method, fill_s, fill_c)
    return ClipResult(ox, oy, roff, proff, st)


def run_team_gpu(ctx, ps, G=16):
    """run_team on the device: the clip of the pair-interaction kernel alone, G lanes of a wave per pair."""
    ax, ay, aoff, bx, by, boff = ps.arrays()
    return ClipResult(*ctx.team_clip_batch(ax, ay, aoff, bx, by, boff, G))


def rings_of(res, p):
    return res.rings(p)


def run_ref_paths(subj, clip, method, fill_s=0, fill_c=0):
    """One clip with several contours per operand through the reference's own Clipper (oracle/_ref: ref_clip takes
    ring sets, as mexclipper.cpp:291-297 hands them to AddPaths).  -> list of int64 (n, 2) rings in solution order."""
    L = ref_lib()
    L.ref_clip.argtypes = [_i64p, _i64p, _i32p, ctypes.c_int, _i64p, _i64p, _i32p, ctypes.c_int, ctypes.c_int,
                           ctypes.c_int, ctypes.c_int, _i64p, _i64p, ctypes.c_int, _i32p, ctypes.c_int,
                           ctypes.POINTER(ctypes.c_int32)]

    def pack(rs):
        off = np.zeros(len(rs) + 1, np.int32)
        for k, r in enumerate(rs):
            off[k + 1] = off[k] + len(r)
        v = np.concatenate(rs).astype(np.int64)
        return np.ascontiguousarray(v[:, 0]), np.ascontiguousarray(v[:, 1]), off
    sx, sy, soff = pack(subj); cx, cy, coff = pack(clip)
    cap = 16 * (len(sx) + len(cx)) + 256
    ox = np.zeros(cap, np.int64); oy = np.zeros(cap, np.int64); ooff = np.zeros(cap + 1, np.int32)
    nr = ctypes.c_int32()
    rc = L.ref_clip(sx, sy, soff, len(subj), cx, cy, coff, len(clip), method, fill_s, fill_c, ox, oy, cap, ooff, cap, ctypes.byref(nr))
    assert rc == 0, rc
    return [np.stack([ox[ooff[r]:ooff[r + 1]], oy[ooff[r]:ooff[r + 1]]], 1) for r in range(nr.value)]


def run_ref_offset(rings, delta, join, info=0.0):
    """The reference's ClipperOffset through the mex's glue (oracle/ref_clipper_shim.cpp ref_offset) -> list of rings."""
    L = ref_lib()
    L.ref_offset.argtypes = [_i64p, _i64p, _i32p, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_double,
                             _i64p, _i64p, ctypes.c_int, _i32p, ctypes.c_int, ctypes.POINTER(ctypes.c_int32)]
    x = np.ascontiguousarray(np.concatenate([np.asarray(r)[:, 0] for r in rings]), np.int64)
    y = np.ascontiguousarray(np.concatenate([np.asarray(r)[:, 1] for r in rings]), np.int64)
    off = np.zeros(len(rings) + 1, np.int32); off[1:] = np.cumsum([len(r) for r in rings])
    cap = 64 * (len(x) + 16)
    while True:
        ox = np.zeros(cap, np.int64); oy = np.zeros(cap, np.int64); roff = np.zeros(cap + 1, np.int32); nr = ctypes.c_int32()
        rc = L.ref_offset(x, y, off, len(rings), float(delta), int(join), float(info), ox, oy, cap, roff, cap, ctypes.byref(nr))
        if rc == 3:
            cap *= 4
            continue
        assert rc == 0, rc
        return [np.stack([ox[roff[r]:roff[r + 1]], oy[roff[r]:roff[r + 1]]], 1) for r in range(nr.value)]


def run_host_offset(rings, delta, join, info=0.0):
    """The library's offset mode with its clean-up sweep on the host (tests/hostsim) -> list of rings."""
    L = host_lib()
    L.host_offset.argtypes = [_i64p, _i64p, _i32p, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_double,
                              _i64p, _i64p, ctypes.c_int64, _i32p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int32)]
    x = np.ascontiguousarray(np.concatenate([np.asarray(r)[:, 0] for r in rings]), np.int64)
    y = np.ascontiguousarray(np.concatenate([np.asarray(r)[:, 1] for r in rings]), np.int64)
    off = np.zeros(len(rings) + 1, np.int32); off[1:] = np.cumsum([len(r) for r in rings])
    cap = 64 * (len(x) + 16)
    while True:
        ox = np.zeros(cap, np.int64); oy = np.zeros(cap, np.int64); roff = np.zeros(cap + 1, np.int32); nr = ctypes.c_int32()
        rc = L.host_offset(x, y, off, len(rings), float(delta), int(join), float(info), ox, oy, cap, roff, cap, ctypes.byref(nr))
        if rc == 3 or rc == 5:      # ST_OUT_CAPACITY / workspace tiers exhausted at this output size
            cap *= 4
            if cap > 1 << 28:
                raise RuntimeError("host_offset: status %d" % rc)
            continue
        assert rc == 0, rc
        return [np.stack([ox[roff[r]:roff[r + 1]], oy[roff[r]:roff[r + 1]]], 1) for r in range(nr.value)]


def offset_golden():
    """[(rings, delta, join, info, expected rings)] of tests/golden/offset_golden.npz (outputs of the reference build)."""
    d = np.load(os.path.join(ROOT, "tests", "golden", "offset_golden.npz"))
    cases = []
    for c in range(len(d["params"])):
        rings = [np.stack([d["x"][d["ring_off"][r]:d["ring_off"][r + 1]], d["y"][d["ring_off"][r]:d["ring_off"][r + 1]]], 1)
                 for r in range(d["case_ring0"][c], d["case_ring0"][c + 1])]
        want = [np.stack([d["out_x"][d["out_ring_off"][r]:d["out_ring_off"][r + 1]], d["out_y"][d["out_ring_off"][r]:d["out_ring_off"][r + 1]]], 1)
                for r in range(d["case_out0"][c], d["case_out0"][c + 1])]
        delta, join, info = d["params"][c]
        cases.append((rings, float(delta), int(join), float(info), want))
    return cases
