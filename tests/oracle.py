"""ctypes binding of oracle/_build/libsubzero_oracle.so (TEST INFRASTRUCTURE: the CPU checker)."""
import ctypes
import os
import subprocess
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)
_bp = ctypes.POINTER(ctypes.c_uint8)

FLOE_FIELDS = (
    [("n", ctypes.c_int32), ("nb", ctypes.c_int32), ("n_stress", ctypes.c_int32), ("n_mc", ctypes.c_int32)]
    + [(k, _dp) for k in ["Xi", "Yi", "alpha_i", "Ui", "Vi", "ksi_ice", "h", "area", "mass", "inertia_moment",
                          "rmax", "dXi_p", "dYi_p", "dUi_p", "dVi_p", "dalpha_i_p", "dksi_ice_p", "FxOA", "FyOA",
                          "torqueOA"]]
    + [("alive", _ip), ("vert_off", _ip), ("c0x", _dp), ("c0y", _dp), ("cax", _dp), ("cay", _dp),
       ("StressH", _dp), ("StressCount", _ip), ("Stress", _dp), ("strain", _dp), ("mcX", _dp), ("mcY", _dp),
       ("mcA", _bp), ("collision_force", _dp), ("collision_torque", _dp), ("OverlapArea", _dp),
       ("kill", _ip), ("transfer", _ip)])


class CFloes(ctypes.Structure):
    _fields_ = FLOE_FIELDS


class CParams(ctypes.Structure):
    _fields_ = [("dt", ctypes.c_double), ("HFo", ctypes.c_double), ("Modulus", ctypes.c_double),
                ("mu", ctypes.c_double), ("nu", ctypes.c_double),
                ("doInt", ctypes.c_int32), ("COLLISION", ctypes.c_int32), ("PERIODIC", ctypes.c_int32),
                ("reserved", ctypes.c_int32),
                ("c2_boundary", ctypes.c_double * 10),
                ("fb_h", ctypes.c_double), ("fb_area", ctypes.c_double), ("fb_Xi", ctypes.c_double),
                ("fb_Yi", ctypes.c_double), ("fb_Ui", ctypes.c_double), ("fb_Vi", ctypes.c_double),
                ("fb_ksi_ice", ctypes.c_double),
                ("nxo", ctypes.c_int32), ("nyo", ctypes.c_int32),
                ("Xo", _dp), ("Yo", _dp), ("Uocn", _dp), ("Vocn", _dp), ("Uwind", _dp), ("Vwind", _dp),
                ("fCoriolis", ctypes.c_double), ("turn_angle", ctypes.c_double)]


def pack_floes(S, cls=CFloes):
    """Struct of pointers into S's arrays (S must outlive the struct)."""
    c = cls()
    c.n, c.nb, c.n_stress, c.n_mc = S.n, S.nb, S.n_stress, S.n_mc
    for name, typ in cls._fields_[4:]:
        arr = getattr(S, name)
        assert arr.flags["C_CONTIGUOUS"], name
        setattr(c, name, arr.ctypes.data_as(typ))
    return c


def pack_params(P, cls=CParams):
    c = cls()
    for k in ["dt", "HFo", "Modulus", "mu", "nu", "doInt", "COLLISION", "PERIODIC", "fb_h", "fb_area", "fb_Xi",
              "fb_Yi", "fb_Ui", "fb_Vi", "fb_ksi_ice", "fCoriolis", "turn_angle"]:
        setattr(c, k, getattr(P, k))
    c.reserved = 0
    cb = np.asarray(P.c2_boundary, np.float64)
    for k in range(5):
        c.c2_boundary[2 * k] = cb[0, k]; c.c2_boundary[2 * k + 1] = cb[1, k]
    c.nxo, c.nyo = len(P.Xo), len(P.Yo)
    # MATLAB column-major (ny x nx): element (y, x) at y + ny * x  ==  Fortran order of the numpy array
    P._keep = [np.ascontiguousarray(P.Xo, np.float64), np.ascontiguousarray(P.Yo, np.float64)] + \
              [np.ascontiguousarray(np.asarray(a, np.float64).T) for a in (P.Uocn, P.Vocn, P.Uwind, P.Vwind)]
    for name, arr in zip(["Xo", "Yo", "Uocn", "Vocn", "Uwind", "Vwind"], P._keep):
        setattr(c, name, arr.ctypes.data_as(_dp))
    return c


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(ROOT, "oracle", "_build", "libsubzero_oracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("subzero_oracle.cpp", "subzero_oracle.h", "or_math.h")]
    if (not os.path.exists(path)) or os.path.getmtime(path) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], stdout=sys.stderr)   # (bench.py prints ONE line on stdout)
    L = ctypes.CDLL(path)
    L.or_set_clipper.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.or_set_mode.argtypes = [ctypes.c_int]
    L.or_step.argtypes = [ctypes.POINTER(CFloes), ctypes.POINTER(CParams), _ip, _dp, ctypes.c_int64,
                          _ip, _ip, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), _ip, ctypes.c_int32]
    L.or_ring_area.restype = ctypes.c_double
    L.or_ring_area.argtypes = [_dp, _dp, ctypes.c_int]
    L.or_ring_centroid.argtypes = [_dp, _dp, ctypes.c_int, _dp, _dp]
    L.or_polyarea.restype = ctypes.c_double
    L.or_polyarea.argtypes = [_dp, _dp, ctypes.c_int]
    L.or_interx.argtypes = [_dp, _dp, ctypes.c_int, _dp, _dp, ctypes.c_int, _dp, _dp, ctypes.c_int]
    L.or_p_poly_dist.argtypes = [_dp, _dp, ctypes.c_int, _dp, _dp, ctypes.c_int, _dp]
    L.or_inpolygon.argtypes = [_dp, _dp, ctypes.c_int, _dp, _dp, ctypes.c_int, _bp]
    L.or_floe_interactions.argtypes = [_dp, _dp, ctypes.c_int, _dp, _dp, _dp, ctypes.c_int, _dp, ctypes.c_int,
                                       _dp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                       ctypes.c_double, _dp, _dp, _dp, ctypes.c_int]
    L.or_sincos.argtypes = [ctypes.c_double, _dp, _dp]
    L.or_atan2.restype = ctypes.c_double
    L.or_atan2.argtypes = [ctypes.c_double, ctypes.c_double]
    # clip function: the reference's own Clipper when its build is present, else the host build of the
    # device engine (which the CPU suite pins bit-exactly to the reference through the golden vectors)
    ref = os.path.join(ROOT, "oracle", "_ref", "libclipper_ref.so")
    if not os.path.exists(ref) and os.path.isdir("/root/reference/private"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=sys.stderr)
    if os.path.exists(ref):
        rc = L.or_set_clipper(ref.encode(), 0); L.clip_kind = "reference"
    else:
        import clipgen
        clipgen.host_lib()
        rc = L.or_set_clipper(os.path.join(ROOT, "tests", "hostsim", "libvatti_host.so").encode(), 1)
        L.clip_kind = "port"
    assert rc == 0
    _lib = L
    return L


MODE_DEVICE_DEFINED = 0      # or_math.h trig, 256-partial Monte-Carlo fold: what the device reproduces bit for bit
MODE_LIBM, MODE_SEQSUM = 1, 2
MODE_INDEPENDENT = MODE_LIBM | MODE_SEQSUM   # host libm + left-to-right means: nothing shared with the kernels


def wall_order(start, reverse):
    """mode bits for the start vertex / direction of the domain-box ring (floe_interactions.m:31-32)."""
    return ((start & 3) << 8) | ((1 if reverse else 0) << 10)


class mode:
    """with oracle.mode(oracle.MODE_INDEPENDENT): ...  -- the oracle's definitions for the enclosed steps."""

    def __init__(self, m):
        self.m = int(m)

    def __enter__(self):
        self.old = lib().or_set_mode(self.m)
        return self

    def __exit__(self, *a):
        lib().or_set_mode(self.old)
        return False


def _p(a, t=_dp):
    return a.ctypes.data_as(t)


class StepResult:
    pass


def step(S, P, threads=1):
    """One reference timestep on S (in place).  Returns contact rows, candidate pairs, N_ext."""
    L = lib()
    cf, cp = pack_floes(S), pack_params(P)
    cap_rows = 64 * S.n + 1024
    cap_pairs = 64 * S.n + 1024
    inter_off = np.zeros(S.n + 1, np.int32); inter = np.zeros((cap_rows, 7))
    pi = np.zeros(cap_pairs, np.int32); pj = np.zeros(cap_pairs, np.int32)
    npairs = ctypes.c_int64(); next_ = ctypes.c_int32()
    rc = L.or_step(ctypes.byref(cf), ctypes.byref(cp), _p(inter_off, _ip), _p(inter), cap_rows,
                   _p(pi, _ip), _p(pj, _ip), cap_pairs, ctypes.byref(npairs), ctypes.byref(next_), threads)
    if rc != 0:
        raise RuntimeError("or_step failed: %d" % rc)
    r = StepResult()
    r.inter_off = inter_off
    r.inter = inter[:inter_off[-1]].copy()
    r.pairs = np.stack([pi[:npairs.value], pj[:npairs.value]], 1)
    r.n_ext = next_.value
    return r


def ring_area(x, y):
    x = np.ascontiguousarray(x, np.float64); y = np.ascontiguousarray(y, np.float64)
    return lib().or_ring_area(_p(x), _p(y), len(x))


def polyarea(x, y):
    x = np.ascontiguousarray(x, np.float64); y = np.ascontiguousarray(y, np.float64)
    return lib().or_polyarea(_p(x), _p(y), len(x))


def ring_centroid(x, y):
    x = np.ascontiguousarray(x, np.float64); y = np.ascontiguousarray(y, np.float64)
    cx, cy = ctypes.c_double(), ctypes.c_double()
    lib().or_ring_centroid(_p(x), _p(y), len(x), ctypes.byref(cx), ctypes.byref(cy))
    return cx.value, cy.value


def interx(x1, y1, x2, y2):
    a = [np.ascontiguousarray(v, np.float64) for v in (x1, y1, x2, y2)]
    cap = 4 * (len(x1) + len(x2)) + 16
    px, py = np.zeros(cap), np.zeros(cap)
    n = lib().or_interx(_p(a[0]), _p(a[1]), len(x1), _p(a[2]), _p(a[3]), len(x2), _p(px), _p(py), cap)
    return px[:n], py[:n]


def floe_interactions(c1, s1, c2, s2, boundary, c2_boundary, periodic, modulus, dt, mu=0.2, nu=0.3):
    c1x, c1y = np.ascontiguousarray(c1[0], np.float64), np.ascontiguousarray(c1[1], np.float64)
    c2x, c2y = np.ascontiguousarray(c2[0], np.float64), np.ascontiguousarray(c2[1], np.float64)
    s1 = np.ascontiguousarray(s1, np.float64); s2 = np.ascontiguousarray(s2, np.float64)
    cb = np.ascontiguousarray(np.asarray(c2_boundary, np.float64).T.ravel())
    cap = 64
    f, pc, ov = np.zeros((cap, 2)), np.zeros((cap, 2)), np.zeros(cap)
    n = lib().or_floe_interactions(_p(c1x), _p(c1y), len(c1x), _p(s1), _p(c2x), _p(c2y), len(c2x), _p(s2),
                                   int(boundary), _p(cb), int(periodic), modulus, dt, mu, nu, _p(f), _p(pc), _p(ov), cap)
    if n < 0:
        raise RuntimeError("or_floe_interactions failed: %d" % n)
    return f[:n], pc[:n], ov[:n]


EULERIAN_FIELDS = ["u", "v", "du", "dv", "stress", "stressxx", "stressyx", "stressxy", "stressyy", "strainux", "strainvx",
                   "strainuy", "strainvy", "c", "Over", "Mtot", "area", "h"]


def eulerian_data(S, Nx, Ny, Nb, c2_boundary, periodic):
    """calc_eulerian_data.m restated (oracle): {field: (Ny, Nx) array}."""
    L = lib()
    cb = np.ascontiguousarray(np.asarray(c2_boundary, np.float64).T.ravel())
    out = np.zeros((18, Nx, Ny))
    L.or_eulerian.argtypes = [ctypes.POINTER(CFloes), ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp]
    cf = pack_floes(S)
    rc = L.or_eulerian(ctypes.byref(cf), Nx, Ny, Nb, _p(cb), int(periodic), _p(out))
    if rc != 0:
        raise RuntimeError("or_eulerian failed: %d" % rc)
    return {k: out[q].T.copy() for q, k in enumerate(EULERIAN_FIELDS)}


def ring_box_area(x, y, xl, xh, yl, yh):
    L = lib()
    L.or_ring_box_area.restype = ctypes.c_double
    L.or_ring_box_area.argtypes = [_dp, _dp, ctypes.c_int] + [ctypes.c_double] * 4
    x = np.ascontiguousarray(x, np.float64); y = np.ascontiguousarray(y, np.float64)
    return L.or_ring_box_area(_p(x), _p(y), len(x), xl, xh, yl, yh)


def polyangles(x, y):
    """polygon_operations/polyangles.m for a clockwise ring given without its closing copy (degrees)."""
    L = lib()
    L.or_polyangles.argtypes = [_dp, _dp, ctypes.c_int, _dp]
    x = np.ascontiguousarray(x, np.float64); y = np.ascontiguousarray(y, np.float64)
    out = np.zeros(len(x))
    L.or_polyangles(_p(x), _p(y), len(x), _p(out))
    return out


def frac_corner(S, ids, offsets, grind):
    """frac_corner.m:1-188 restated (oracle) for the plain case, for the floes ids (1-based) with the grind flags laid
    out at `offsets` -> (npieces per floe, piece_off, pieces [P x 12], rings: list of 2 x m closed c_alpha)."""
    L = lib()
    ip = ctypes.POINTER(ctypes.c_int32)
    u8 = ctypes.POINTER(ctypes.c_uint8)
    L.or_frac_corner.argtypes = [ctypes.POINTER(CFloes), ip, ctypes.c_int, ip, u8, ip, ip, _dp, ctypes.c_longlong, ip, _dp, _dp, ctypes.c_longlong]
    ids = np.ascontiguousarray(ids, np.int32); offsets = np.ascontiguousarray(offsets, np.int32); grind = np.ascontiguousarray(grind, np.uint8)
    nsel = len(ids)
    cap_p = int(nsel + grind.sum() + 1); cap_r = int(offsets[-1] + nsel + 5 * grind.sum() + 8)
    npieces = np.zeros(max(nsel, 1), np.int32); poff = np.zeros(nsel + 1, np.int32)
    pieces = np.zeros((cap_p, 12)); roff = np.zeros(cap_p + 1, np.int32); rx = np.zeros(cap_r); ry = np.zeros(cap_r)
    cf = pack_floes(S)
    rc = L.or_frac_corner(ctypes.byref(cf), ids.ctypes.data_as(ip), nsel, offsets.ctypes.data_as(ip), grind.ctypes.data_as(u8),
                          npieces.ctypes.data_as(ip), poff.ctypes.data_as(ip), _p(pieces), cap_p, roff.ctypes.data_as(ip), _p(rx), _p(ry), cap_r)
    if rc != 0:
        raise RuntimeError("or_frac_corner failed: %d" % rc)
    P = int(poff[nsel])
    rings = [np.stack([rx[roff[q]:roff[q + 1]], ry[roff[q]:roff[q + 1]]]) for q in range(P)]
    return npieces[:nsel], poff, pieces[:P], rings


def corner_candidates(S, inter_off, inter, ids, Nb, c2_boundary):
    """corners.m:13-92 restated (oracle) for the floes ids (1-based, Floe(~keep) order) ->
    (offsets, count, angles, da, N0)."""
    L = lib()
    ip = ctypes.POINTER(ctypes.c_int32)
    L.or_corner_candidates.argtypes = [ctypes.POINTER(CFloes), ip, _dp, ip, ctypes.c_int, ctypes.c_int, _dp, ip, ip, _dp,
                                       ctypes.POINTER(ctypes.c_uint8), ip]
    cb = np.ascontiguousarray(np.asarray(c2_boundary, np.float64).T.ravel())
    ids = np.ascontiguousarray(ids, np.int32)
    off = np.ascontiguousarray(inter_off, np.int32)
    rows = np.ascontiguousarray(inter, np.float64).reshape(-1, 7) if len(inter) else np.zeros((1, 7))
    nsub = len(ids)
    total = int(sum(S.vert_off[i] - S.vert_off[i - 1] for i in ids))
    offsets = np.zeros(nsub + 1, np.int32); count = np.zeros(max(nsub, 1), np.int32)
    ang = np.zeros(total + 1); da = np.zeros(total + 1, np.uint8)
    n_ext = ctypes.c_int32()
    cf = pack_floes(S)
    rc = L.or_corner_candidates(ctypes.byref(cf), off.ctypes.data_as(ip), _p(rows), ids.ctypes.data_as(ip), nsub, Nb, _p(cb),
                                offsets.ctypes.data_as(ip), count.ctypes.data_as(ip), _p(ang),
                                da.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ctypes.byref(n_ext))
    if rc != 0:
        raise RuntimeError("or_corner_candidates failed: %d" % rc)
    return offsets, count[:nsub], ang[:total], da[:total], n_ext.value


def fracture_deform(S, inter_off, inter, ids):
    """fracture_floe.m:14-52 restated (oracle) -> (changed, Xi, Yi, area, rings)."""
    L = lib()
    ip = ctypes.POINTER(ctypes.c_int32)
    L.or_fracture_deform.argtypes = [ctypes.POINTER(CFloes), ip, _dp, ip, ctypes.c_int, ip, _dp, _dp, _dp, ip, _dp, _dp, ctypes.c_int]
    ids = np.ascontiguousarray(ids, np.int32)
    off = np.ascontiguousarray(inter_off, np.int32)
    rows = np.ascontiguousarray(inter, np.float64).reshape(-1, 7) if len(inter) else np.zeros((1, 7))
    nsel = len(ids)
    changed = np.zeros(max(nsel, 1), np.int32); X = np.zeros(max(nsel, 1)); Y = np.zeros(max(nsel, 1)); A = np.zeros(max(nsel, 1))
    roff = np.zeros(nsel + 1, np.int32)
    cap = 4 * int(S.vert_off[-1]) + 64 * nsel + 64
    cx = np.zeros(cap); cy = np.zeros(cap)
    cf = pack_floes(S)
    rc = L.or_fracture_deform(ctypes.byref(cf), off.ctypes.data_as(ip), _p(rows), ids.ctypes.data_as(ip), nsel, changed.ctypes.data_as(ip),
                              _p(X), _p(Y), _p(A), roff.ctypes.data_as(ip), _p(cx), _p(cy), cap)
    if rc != 0:
        raise RuntimeError("or_fracture_deform failed: %d" % rc)
    rings = [np.stack([cx[roff[s]:roff[s + 1]], cy[roff[s]:roff[s + 1]]]) for s in range(nsel)]
    return changed[:nsel], X[:nsel], Y[:nsel], A[:nsel], rings
