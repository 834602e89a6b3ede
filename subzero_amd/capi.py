"""ctypes binding of include/subzero_hip.h.  No arithmetic here; fails loudly when the HIP library is
missing (there is no CPU fallback by design)."""
import ctypes
import os
import time
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SUBZERO_HIP_LIB") or os.path.join(_HERE, "csrc", "libsubzero_hip.so")   # (the variable: diagnostic builds of the same library)

SZ_OK = 0
SZ_ERR_NO_DEVICE = -1
SZ_ERR_BAD_ARG = -2
SZ_ERR_HIP = -3
SZ_ERR_CAPACITY = -4
SZ_ERR_WORKSPACE = -5
SZ_ERR_STATE = -6

CLIP_DIFFERENCE, CLIP_INTERSECTION, CLIP_XOR, CLIP_UNION = 0, 1, 2, 3

_i64p = np.ctypeslib.ndpointer(np.int64, flags="C")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C")
_vp = ctypes.c_void_p


class SubzeroError(RuntimeError):
    pass


class ClipStats(ctypes.Structure):
    _fields_ = [("kernel_ms", ctypes.c_double), ("pairs", ctypes.c_int64),
                ("input_vertices", ctypes.c_int64), ("rings_out", ctypes.c_int64),
                ("vertices_out", ctypes.c_int64), ("retried", ctypes.c_int64),
                ("failed", ctypes.c_int64)]


_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)
_bp = ctypes.POINTER(ctypes.c_uint8)


class FloeStateC(ctypes.Structure):
    """szFloeState"""
    _fields_ = (
        [("n", ctypes.c_int32), ("nb", ctypes.c_int32), ("n_stress", ctypes.c_int32), ("n_mc", ctypes.c_int32)]
        + [(k, _dp) for k in ["Xi", "Yi", "alpha_i", "Ui", "Vi", "ksi_ice", "h", "area", "mass",
                              "inertia_moment", "rmax", "dXi_p", "dYi_p", "dUi_p", "dVi_p", "dalpha_i_p",
                              "dksi_ice_p", "FxOA", "FyOA", "torqueOA"]]
        + [("alive", _ip), ("vert_off", _ip), ("c0x", _dp), ("c0y", _dp), ("cax", _dp), ("cay", _dp),
           ("StressH", _dp), ("StressCount", _ip), ("Stress", _dp), ("strain", _dp), ("mcX", _dp),
           ("mcY", _dp), ("mcA", _bp), ("collision_force", _dp), ("collision_torque", _dp),
           ("OverlapArea", _dp), ("kill", _ip), ("transfer", _ip)])


class StepParamsC(ctypes.Structure):
    """szStepParams"""
    _fields_ = [("dt", ctypes.c_double), ("HFo", ctypes.c_double), ("Modulus", ctypes.c_double),
                ("mu", ctypes.c_double), ("nu", ctypes.c_double),
                ("doInt", ctypes.c_int32), ("COLLISION", ctypes.c_int32), ("PERIODIC", ctypes.c_int32),
                ("flags", ctypes.c_int32),
                ("c2_boundary", ctypes.c_double * 10),
                ("fb_h", ctypes.c_double), ("fb_area", ctypes.c_double), ("fb_Xi", ctypes.c_double),
                ("fb_Yi", ctypes.c_double), ("fb_Ui", ctypes.c_double), ("fb_Vi", ctypes.c_double),
                ("fb_ksi_ice", ctypes.c_double),
                ("nxo", ctypes.c_int32), ("nyo", ctypes.c_int32),
                ("Xo", _dp), ("Yo", _dp), ("Uocn", _dp), ("Vocn", _dp), ("Uwind", _dp), ("Vwind", _dp),
                ("fCoriolis", ctypes.c_double), ("turn_angle", ctypes.c_double)]


class StepStats(ctypes.Structure):
    _fields_ = [("step_ms", ctypes.c_double), ("broad_ms", ctypes.c_double), ("narrow_ms", ctypes.c_double),
                ("rows_ms", ctypes.c_double), ("integrate_ms", ctypes.c_double),
                ("n_ext", ctypes.c_int64), ("n_pairs", ctypes.c_int64), ("n_rows", ctypes.c_int64),
                ("pair_vertices", ctypes.c_int64), ("retried", ctypes.c_int64),
                ("pair_rows", ctypes.c_int64), ("team_pairs", ctypes.c_int64), ("team_vertices", ctypes.c_int64),
                ("pairs_ms", ctypes.c_double), ("times_step", ctypes.c_int64)]


SYNC_DYNAMIC, SYNC_STRESSH, SYNC_ALL = 1, 2, 3
STEP_QUEUE_AHEAD, STEP_NEXT_DOINT = 1, 2     # szStepParams.flags
F_STATE, F_RING, F_STRESSH, F_MC, F_ALL = 1, 2, 4, 8, 15
SZ_STEP_NEEDS_MC = 2
SZ_STEP_VETOED = 3


class StepVetoed(SubzeroError):
    """szStep returned SZ_STEP_VETOED (spatial partition, queued exchange): nothing was changed; the caller exchanges the
    rest of the messages and steps again (multigpu.SpatialInteractions does)."""


class NeedsMc(SubzeroError):
    """szStep changed nothing: the forcing of these floes (1-based) is due and their Monte-Carlo mask is empty
    (calc_trajectory.m:100-110 draws new points with rand).  Regenerate them, `update(ids, F_MC)`, step again."""

    def __init__(self, ids):
        SubzeroError.__init__(self, "floes %s need new Monte-Carlo points" % (list(ids),))
        self.ids = np.asarray(ids, np.int32)


def pack_floes(S):
    """szFloeState over the arrays of a synth.FloeState-like object (which must outlive the struct)."""
    c = FloeStateC()
    c.n, c.nb, c.n_stress, c.n_mc = S.n, S.nb, S.n_stress, S.n_mc
    for name, typ in FloeStateC._fields_[4:]:
        arr = getattr(S, name)
        if not arr.flags["C_CONTIGUOUS"]:
            raise SubzeroError("field %s must be C-contiguous" % name)
        setattr(c, name, arr.ctypes.data_as(typ))
    return c


def pack_params(P, repack_grids=True):
    c = StepParamsC()
    for k in ["dt", "HFo", "Modulus", "mu", "nu", "doInt", "COLLISION", "PERIODIC", "fb_h", "fb_area", "fb_Xi",
              "fb_Yi", "fb_Ui", "fb_Vi", "fb_ksi_ice", "fCoriolis", "turn_angle"]:
        setattr(c, k, getattr(P, k))
    c.flags = 0
    cb = np.asarray(P.c2_boundary, np.float64)
    for k in range(5):
        c.c2_boundary[2 * k] = cb[0, k]; c.c2_boundary[2 * k + 1] = cb[1, k]
    c.nxo, c.nyo = len(P.Xo), len(P.Yo)
    # (ny, nx) numpy arrays -> MATLAB column-major storage.  The library reads the grids on forcing steps only
    # (and on the first step), as the reference refreshes them (Subzero.m:153-166): no need to repack them otherwise.
    if repack_grids or P.doInt or getattr(P, "_keep_c", None) is None:
        P._keep_c = [np.ascontiguousarray(P.Xo, np.float64), np.ascontiguousarray(P.Yo, np.float64)] + \
                    [np.ascontiguousarray(np.asarray(a, np.float64).T) for a in (P.Uocn, P.Vocn, P.Uwind, P.Vwind)]
    for name, arr in zip(["Xo", "Yo", "Uocn", "Vocn", "Uwind", "Vwind"], P._keep_c):
        setattr(c, name, arr.ctypes.data_as(_dp))
    return c


class ParamPacker:
    """szStepParams for successive steps of one uploaded floe field.  The packed block is reused while nothing but
    doInt changes, and the ocean / wind grids are repacked only when the library reads them (on forcing steps and
    on the first step after an upload): the call should cost microseconds, not the packing."""
    _SCALARS = ("dt", "HFo", "Modulus", "mu", "nu", "COLLISION", "PERIODIC", "fb_h", "fb_area", "fb_Xi", "fb_Yi",
                "fb_Ui", "fb_Vi", "fb_ksi_ice", "fCoriolis", "turn_angle")

    def __init__(self):
        self._cp = self._key = None

    def pack(self, P):
        key = tuple(getattr(P, k) for k in self._SCALARS) + (np.asarray(P.c2_boundary, np.float64).tobytes(),)
        # queue_ahead / next_doInt (optional attributes of P): szStepParams.flags, SZ_STEP_QUEUE_AHEAD | SZ_STEP_NEXT_DOINT
        flags = (STEP_QUEUE_AHEAD if getattr(P, "queue_ahead", 0) else 0) | (STEP_NEXT_DOINT if getattr(P, "next_doInt", 0) else 0)
        if self._cp is not None and not P.doInt and key == self._key:
            self._cp.doInt = 0; self._cp.flags = flags
            return self._cp
        self._cp, self._key = pack_params(P, repack_grids=self._cp is None), key
        self._cp.flags = flags
        return self._cp


# every symbol include/subzero_hip.h declares (tests check the library exports all of them)
EXPORTS = [
    "szAbiVersion", "szCreate", "szDestroy", "szLastError",
    "szClipBatch", "szClipPathsBatch", "szPolyclipBatch", "szClipBatchDevice", "szOrientation",
    "szDebugTeamClipBatch",
    "szFloesUpload", "szStep", "szStepDiagnostics", "szStepTimes", "szFloesDownload", "szInteractionsDownload", "szPairsDownload",
    "szFloesUpdate", "szFloesFetch", "szFloesResync", "szMcEmptyDownload", "szGhostsDownload",
    "szEulerianData", "szCornerCandidates", "szCornerCuts", "szFractureDeform", "szPotentialInteractions", "szOffsetPaths", "szSpatialInit", "szSpatialSetAxis", "szSpatialMessageBytes", "szSpatialPack", "szSpatialUnpack", "szSpatialUnpackKnown", "szSpatialOwned",
    "szSpatialPackQueued", "szSpatialPackFinish", "szSpatialUnpackQueued", "szSpatialHeaders",
    "szShardSegmentBytes", "szStepShardBegin", "szStepShardExport", "szStepShardFinish",
    "szShardForcingBytes", "szStepShardForcing", "szStepShardEnd",
    "szDeviceAlloc", "szDeviceFree", "szCopyToDevice", "szCopyToHost", "szSynchronize", "szSetStream",
]

_lib = None


def load():
    """Load libsubzero_hip.so; raises SubzeroError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SubzeroError(
            "libsubzero_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C subzero_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    L.szAbiVersion.restype = ctypes.c_int
    L.szCreate.argtypes = [ctypes.POINTER(_vp), ctypes.c_int]
    L.szDestroy.argtypes = [_vp]
    L.szLastError.argtypes = [_vp]
    L.szLastError.restype = ctypes.c_char_p
    L.szClipBatch.argtypes = [_vp, ctypes.c_int, _i64p, _i64p, _i32p, _i64p, _i64p, _i32p,
                              ctypes.c_int, ctypes.c_int, ctypes.c_int,
                              _i64p, _i64p, ctypes.c_int64, _i32p, ctypes.c_int64, _i32p, _i32p]
    L.szDebugTeamClipBatch.argtypes = [_vp, ctypes.c_int, ctypes.c_int, _i64p, _i64p, _i32p, _i64p, _i64p, _i32p,
                                       _i64p, _i64p, _i32p, _i32p]
    L.szClipPathsBatch.argtypes = [_vp, ctypes.c_int, _i64p, _i64p, _i32p, _i32p, _i64p, _i64p, _i32p, _i32p,
                                   ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   _i64p, _i64p, ctypes.c_int64, _i32p, ctypes.c_int64, _i32p, _i32p]
    L.szPolyclipBatch.argtypes = [_vp, ctypes.c_int, _f64p, _f64p, _i32p, _f64p, _f64p, _i32p,
                                  ctypes.c_int,
                                  _f64p, _f64p, ctypes.c_int64, _i32p, ctypes.c_int64, _i32p, _i32p]
    L.szClipBatchDevice.argtypes = [_vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ClipStats)]
    L.szOrientation.argtypes = [_i64p, _i64p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.szFloesUpload.argtypes = [_vp, ctypes.POINTER(FloeStateC)]
    L.szStep.argtypes = [_vp, ctypes.POINTER(StepParamsC), ctypes.POINTER(StepStats)]
    L.szStepTimes.argtypes = [_vp, ctypes.POINTER(StepStats)]
    L.szStepDiagnostics.argtypes = [_vp, _i64p, ctypes.c_int32]
    L.szFloesDownload.argtypes = [_vp, ctypes.POINTER(FloeStateC), ctypes.c_int]
    L.szInteractionsDownload.argtypes = [_vp, _ip, _dp, ctypes.c_int64]
    L.szPairsDownload.argtypes = [_vp, _ip, _ip, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), _ip]
    L.szFloesUpdate.argtypes = [_vp, ctypes.POINTER(FloeStateC), _ip, ctypes.c_int32, ctypes.c_int]
    L.szFloesFetch.argtypes = [_vp, ctypes.POINTER(FloeStateC), _ip, ctypes.c_int32, ctypes.c_int]
    L.szFloesResync.argtypes = [_vp, ctypes.POINTER(FloeStateC), _ip]
    L.szMcEmptyDownload.argtypes = [_vp, _ip, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
    L.szGhostsDownload.argtypes = [_vp, _ip, _ip, _dp, _dp, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
    L.szDeviceAlloc.argtypes = [_vp, ctypes.POINTER(_vp), ctypes.c_int64]
    L.szDeviceFree.argtypes = [_vp, _vp]
    L.szCopyToDevice.argtypes = [_vp, _vp, _vp, ctypes.c_int64]
    L.szCopyToHost.argtypes = [_vp, _vp, _vp, ctypes.c_int64]
    L.szSynchronize.argtypes = [_vp]
    L.szSetStream.argtypes = [_vp, _vp]
    _lib = L
    return L


class Context:
    """One szContext (one GPU)."""

    def __init__(self, device=0):
        self.L = load()
        self.h = _vp()
        rc = self.L.szCreate(ctypes.byref(self.h), int(device))
        if rc != SZ_OK:
            self.h = None
            raise SubzeroError("szCreate(device=%d) failed with %d: no usable gfx950 GPU "
                               "(this library has no CPU fallback)" % (device, rc))

    def close(self):
        if self.h:
            self.L.szDestroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc, what):
        if rc != SZ_OK:
            raise SubzeroError("%s failed (%d): %s" % (what, rc, self.L.szLastError(self.h).decode()))

    # ---- raw memory ----
    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        d = _vp()
        self.check(self.L.szDeviceAlloc(self.h, ctypes.byref(d), arr.nbytes), "szDeviceAlloc")
        self.check(self.L.szCopyToDevice(self.h, d, arr.ctypes.data_as(_vp), arr.nbytes), "szCopyToDevice")
        return d

    def free(self, d):
        self.check(self.L.szDeviceFree(self.h, d), "szDeviceFree")

    # ---- B1 ----
    def clip_batch(self, ax, ay, aoff, bx, by, boff, method, subj_fill=0, clip_fill=0):
        n = len(aoff) - 1
        total = int(len(ax) + len(bx))
        cap_v, cap_r = 2 * total + 64 * n + 64, total + 8 * n + 8
        while True:
            ox = np.zeros(cap_v, np.int64); oy = np.zeros(cap_v, np.int64)
            roff = np.zeros(cap_r + 1, np.int32); proff = np.zeros(n + 1, np.int32)
            st = np.zeros(max(n, 1), np.int32)
            rc = self.L.szClipBatch(self.h, n, ax, ay, aoff, bx, by, boff, method, subj_fill, clip_fill,
                                    ox, oy, cap_v, roff, cap_r, proff, st)
            if rc == SZ_ERR_CAPACITY:
                cap_v *= 4; cap_r *= 4
                continue
            self.check(rc, "szClipBatch")
            return ox, oy, roff, proff, st[:n]

    def team_clip_batch(self, ax, ay, aoff, bx, by, boff, lanes=16):
        """The clip inside the pair-interaction kernel alone (szDebugTeamClipBatch; tests): -> the layout of clip_batch;
        status 100 = not the team form's case."""
        n = len(aoff) - 1
        ox = np.zeros(16 * max(n, 1), np.int64); oy = np.zeros(16 * max(n, 1), np.int64)
        cnt = np.zeros(max(n, 1), np.int32); st = np.zeros(max(n, 1), np.int32)
        self.check(self.L.szDebugTeamClipBatch(self.h, lanes, n, ax, ay, aoff, bx, by, boff, ox, oy, cnt, st), "szDebugTeamClipBatch")
        cnt = cnt[:n]; st = st[:n]
        has = (cnt > 0).astype(np.int32)
        proff = np.zeros(n + 1, np.int32); proff[1:] = np.cumsum(has)
        roff = np.zeros(int(proff[-1]) + 1, np.int32); roff[1:] = np.cumsum(cnt[cnt > 0])
        keep = (np.arange(16)[None, :] < cnt[:, None]).ravel()
        return ox[:16 * n][keep], oy[:16 * n][keep], roff, proff, st

    def clip_paths_batch(self, ax, ay, a_ring_off, a_pair_off, bx, by, b_ring_off, b_pair_off, method, subj_fill=0, clip_fill=0):
        """clip_batch with several contours per operand (ring CSR + pair -> ring range CSR)."""
        n = len(a_pair_off) - 1
        total = int(len(ax) + len(bx))
        cap_v, cap_r = 2 * total + 64 * n + 64, total + 8 * n + 8
        while True:
            ox = np.zeros(cap_v, np.int64); oy = np.zeros(cap_v, np.int64)
            roff = np.zeros(cap_r + 1, np.int32); proff = np.zeros(n + 1, np.int32)
            st = np.zeros(max(n, 1), np.int32)
            rc = self.L.szClipPathsBatch(self.h, n, ax, ay, a_ring_off, a_pair_off, bx, by, b_ring_off, b_pair_off,
                                         method, subj_fill, clip_fill, ox, oy, cap_v, roff, cap_r, proff, st)
            if rc == SZ_ERR_CAPACITY:
                cap_v *= 4; cap_r *= 4
                continue
            self.check(rc, "szClipPathsBatch")
            return ox, oy, roff, proff, st[:n]

    def offset_paths(self, rings, delta, join, info=0.0):
        """clipper(P, delta, type [, info]): rings = list of (n, 2) int64 arrays; join 0 square / 1 round / 2 miter.
        -> list of (m, 2) int64 arrays, as the reference's ClipperOffset returns them."""
        x = np.ascontiguousarray(np.concatenate([np.asarray(r)[:, 0] for r in rings]) if rings else np.zeros(0), np.int64)
        y = np.ascontiguousarray(np.concatenate([np.asarray(r)[:, 1] for r in rings]) if rings else np.zeros(0), np.int64)
        off = np.zeros(len(rings) + 1, np.int32); off[1:] = np.cumsum([len(r) for r in rings])
        i64p = ctypes.POINTER(ctypes.c_int64)
        self.L.szOffsetPaths.argtypes = [_vp, i64p, i64p, _ip, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_double, i64p, i64p,
                                         ctypes.c_int64, _ip, ctypes.c_int64, _ip, _ip]
        cap = 64 * (len(x) + 16)
        while True:
            ox = np.zeros(cap, np.int64); oy = np.zeros(cap, np.int64); roff = np.zeros(cap + 1, np.int32)
            nr = ctypes.c_int32(); st = ctypes.c_int32()
            rc = self.L.szOffsetPaths(self.h, x.ctypes.data_as(i64p), y.ctypes.data_as(i64p), off.ctypes.data_as(_ip), len(rings), float(delta),
                                      int(join), float(info), ox.ctypes.data_as(i64p), oy.ctypes.data_as(i64p), cap, roff.ctypes.data_as(_ip),
                                      cap, ctypes.byref(nr), ctypes.byref(st))
            if rc == SZ_ERR_CAPACITY:
                cap *= 4
                continue
            self.check(rc, "szOffsetPaths")
            if st.value != 0:
                raise SubzeroError("szOffsetPaths: clip status %d" % st.value)
            return [np.stack([ox[roff[r]:roff[r + 1]], oy[roff[r]:roff[r + 1]]], 1) for r in range(nr.value)]

    def polyclip_batch(self, ax, ay, aoff, bx, by, boff, method):
        n = len(aoff) - 1
        total = int(len(ax) + len(bx))
        cap_v, cap_r = 2 * total + 64 * n + 64, total + 8 * n + 8
        while True:
            ox = np.zeros(cap_v, np.float64); oy = np.zeros(cap_v, np.float64)
            roff = np.zeros(cap_r + 1, np.int32); proff = np.zeros(n + 1, np.int32)
            st = np.zeros(max(n, 1), np.int32)
            rc = self.L.szPolyclipBatch(self.h, n, ax, ay, aoff, bx, by, boff, method,
                                        ox, oy, cap_v, roff, cap_r, proff, st)
            if rc == SZ_ERR_CAPACITY:
                cap_v *= 4; cap_r *= 4
                continue
            self.check(rc, "szPolyclipBatch")
            return ox, oy, roff, proff, st[:n]

    def clip_batch_device(self, npairs, d_ax, d_ay, d_aoff, d_bx, d_by, d_boff, total_a, total_b, method):
        stats = ClipStats()
        self.check(self.L.szClipBatchDevice(self.h, npairs, d_ax, d_ay, d_aoff, d_bx, d_by, d_boff,
                                            total_a, total_b, method, ctypes.byref(stats)), "szClipBatchDevice")
        return stats


class FloeInteractions:
    """Host mirror of the reference's step call (floe_interactions_all.m:1, lines 16-285): the Floe struct
    array lives on the GPU between steps; `step` is one call of the accelerated part, `download` brings the
    fields MATLAB-side code reads back into the host arrays."""

    def __init__(self, ctx, floes):
        self.ctx, self.floes = ctx, floes
        self._c = pack_floes(floes)
        ctx.check(ctx.L.szFloesUpload(ctx.h, ctypes.byref(self._c)), "szFloesUpload")
        self._packer = ParamPacker()

    def step(self, params):
        cp = self._packer.pack(params)
        st = StepStats()
        rc = self.ctx.L.szStep(self.ctx.h, ctypes.byref(cp), ctypes.byref(st))
        if rc == SZ_STEP_NEEDS_MC:
            raise NeedsMc(self.mc_empty())
        if rc == SZ_STEP_VETOED:
            raise StepVetoed("szStep: vetoed by a queued unpack (a neighbour's message outgrew its plan); nothing was changed")
        self.ctx.check(rc, "szStep")
        return st

    def step_times(self):
        """Stage times of the LAST step (szStepTimes waits for its events; `step` returns those of the step before)."""
        st = StepStats()
        self.ctx.check(self.ctx.L.szStepTimes(self.ctx.h, ctypes.byref(st)), "szStepTimes")
        return st

    DIAG = ("steps", "steps_fused", "spec_misses", "back_misses", "list_hot", "list_cold", "list_wide", "list_handed_on",
            "list_walls", "team_vertices", "cell_cap", "pot_cap", "steps_ahead")

    def diagnostics(self):
        """How the steps of this array have run (szStepDiagnostics): a dict over DIAG."""
        out = np.zeros(len(self.DIAG), np.int64)
        self.ctx.check(self.ctx.L.szStepDiagnostics(self.ctx.h, out, len(out)), "szStepDiagnostics")
        return dict(zip(self.DIAG, (int(v) for v in out)))

    def mc_empty(self):
        """Floes (1-based) the last step asked new Monte-Carlo points for."""
        ids = np.zeros(max(self.floes.n, 1), np.int32)
        cnt = ctypes.c_int64()
        self.ctx.check(self.ctx.L.szMcEmptyDownload(self.ctx.h, ids.ctypes.data_as(_ip), len(ids), ctypes.byref(cnt)),
                       "szMcEmptyDownload")
        return ids[:cnt.value].copy()

    def update(self, ids, what):
        """Field groups `what` (F_*) of the floes `ids` (1-based) from the host arrays to the GPU."""
        ids = np.ascontiguousarray(ids, np.int32)
        self.ctx.check(self.ctx.L.szFloesUpdate(self.ctx.h, ctypes.byref(self._c), ids.ctypes.data_as(_ip), len(ids), what),
                       "szFloesUpdate")

    def fetch(self, ids, what):
        """... and from the GPU into the host arrays."""
        ids = np.ascontiguousarray(ids, np.int32)
        self.ctx.check(self.ctx.L.szFloesFetch(self.ctx.h, ctypes.byref(self._c), ids.ctypes.data_as(_ip), len(ids), what),
                       "szFloesFetch")
        return self.floes

    def resync(self, floes, src):
        """Replace the resident array by `floes` (any size / ring layout); new floe k keeps the device-resident stress
        history and Monte-Carlo points of old floe src[k] (1-based), 0 = read them from `floes`."""
        c = pack_floes(floes)
        src = np.ascontiguousarray(src, np.int32)
        if len(src) != floes.n:
            raise SubzeroError("resync: src must name a predecessor (or 0) for every floe")
        self.ctx.check(self.ctx.L.szFloesResync(self.ctx.h, ctypes.byref(c), src.ctypes.data_as(_ip)), "szFloesResync")
        self.floes, self._c = floes, c
        self._packer = ParamPacker()

    EULERIAN_FIELDS = ["u", "v", "du", "dv", "stress", "stressxx", "stressyx", "stressxy", "stressyy", "strainux", "strainvx",
                       "strainuy", "strainvy", "c", "Over", "Mtot", "area", "h"]

    def eulerian_data(self, Nx, Ny, Nb, c2_boundary, periodic):
        """calc_eulerian_data.m on the device-resident floes -> ({field: (Ny, Nx) array}, kernel ms)."""
        cb = np.ascontiguousarray(np.asarray(c2_boundary, np.float64).T.ravel())
        out = np.zeros((18, Nx, Ny))        # [field][ii][jj]: column-major Ny x Nx per field
        ms = ctypes.c_double()
        self.ctx.L.szEulerianData.argtypes = [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp, _dp]
        t0 = time.perf_counter()
        rc = self.ctx.L.szEulerianData(self.ctx.h, Nx, Ny, Nb, cb.ctypes.data_as(_dp), int(periodic), out.ctypes.data_as(_dp), ctypes.byref(ms))
        self.last_call_ms = (time.perf_counter() - t0) * 1e3
        self.ctx.check(rc, "szEulerianData")
        return {k: out[q].T.copy() for q, k in enumerate(self.EULERIAN_FIELDS)}, ms.value

    def corner_candidates(self, ids, Nb, c2_boundary):
        """corners.m:13-92 for the floes ids (1-based, in Floe(~keep) order) on the contact rows of the last step ->
        (offsets, count, angles, da, N0, kernel ms); see include/subzero_hip.h szCornerCandidates."""
        ids = np.ascontiguousarray(ids, np.int32)
        nsub = len(ids)
        vo = self.floes.vert_off
        ok = ids[(ids >= 1) & (ids <= self.floes.n)].astype(np.int64)
        total = int((vo[ok].astype(np.int64) - vo[ok - 1]).sum())
        offsets = np.zeros(nsub + 1, np.int32); count = np.zeros(max(nsub, 1), np.int32)
        ang = np.zeros(total + 1); da = np.zeros(total + 1, np.uint8)
        n_ext = ctypes.c_int32(); ms = ctypes.c_double()
        cb = np.ascontiguousarray(np.asarray(c2_boundary, np.float64).T.ravel())
        u8 = ctypes.POINTER(ctypes.c_uint8)
        self.ctx.L.szCornerCandidates.argtypes = [_vp, _ip, ctypes.c_int, ctypes.c_int, _dp, _ip, _ip, _dp, u8, ctypes.c_int64,
                                                  _ip, _dp]
        t0 = time.perf_counter()
        rc = self.ctx.L.szCornerCandidates(self.ctx.h, ids.ctypes.data_as(_ip), nsub, int(Nb), cb.ctypes.data_as(_dp),
                                           offsets.ctypes.data_as(_ip), count.ctypes.data_as(_ip), ang.ctypes.data_as(_dp),
                                           da.ctypes.data_as(u8), total, ctypes.byref(n_ext), ctypes.byref(ms))
        self.last_call_ms = (time.perf_counter() - t0) * 1e3       # wall clock of the C-ABI call itself (bench.py)
        self.ctx.check(rc, "szCornerCandidates")
        return offsets, count[:nsub], ang[:total], da[:total], n_ext.value, ms.value

    def corner_cuts(self, ids, offsets, grind):
        """frac_corner.m for the plain case (include/subzero_hip.h szCornerCuts): ids (1-based), offsets / grind as
        corner_candidates laid them out -> (npieces per floe, piece_off, pieces [P x 12], rings: list of 2 x m closed c_alpha,
        kernel ms).  npieces[s] = 0: leave floe s to frac_corner.m."""
        ids = np.ascontiguousarray(ids, np.int32); offsets = np.ascontiguousarray(offsets, np.int32)
        grind = np.ascontiguousarray(grind, np.uint8)
        nsel = len(ids)
        cap_p = int(nsel + grind.sum() + 1); cap_r = int(offsets[-1] + nsel + 5 * grind.sum() + 8)
        npieces = np.zeros(max(nsel, 1), np.int32); poff = np.zeros(nsel + 1, np.int32)
        pieces = np.zeros((cap_p, 12)); roff = np.zeros(cap_p + 1, np.int32); rx = np.zeros(cap_r); ry = np.zeros(cap_r)
        ms = ctypes.c_double()
        u8 = ctypes.POINTER(ctypes.c_uint8)
        L = self.ctx.L
        L.szCornerCuts.argtypes = [_vp, _ip, ctypes.c_int, _ip, u8, _ip, _ip, _dp, ctypes.c_int64, _ip, _dp, _dp, ctypes.c_int64, _dp]
        t0 = time.perf_counter()
        rc = L.szCornerCuts(self.ctx.h, ids.ctypes.data_as(_ip), nsel, offsets.ctypes.data_as(_ip), grind.ctypes.data_as(u8),
                            npieces.ctypes.data_as(_ip), poff.ctypes.data_as(_ip), pieces.ctypes.data_as(_dp), cap_p,
                            roff.ctypes.data_as(_ip), rx.ctypes.data_as(_dp), ry.ctypes.data_as(_dp), cap_r, ctypes.byref(ms))
        self.last_call_ms = (time.perf_counter() - t0) * 1e3
        self.ctx.check(rc, "szCornerCuts")
        P = int(poff[nsel])
        rings = [np.stack([rx[roff[q]:roff[q + 1]], ry[roff[q]:roff[q + 1]]]) for q in range(P)]
        return npieces[:nsel], poff, pieces[:P], rings, ms.value

    def fracture_deform(self, ids):
        """fracture_floe.m:14-52 for the floes ids (1-based) on the contact rows of the last step ->
        (changed, Xi, Yi, area, rings): rings[s] = new c_alpha (2 x m, open) where changed[s]."""
        ids = np.ascontiguousarray(ids, np.int32)
        nsel = len(ids)
        changed = np.zeros(max(nsel, 1), np.int32); X = np.zeros(max(nsel, 1)); Y = np.zeros(max(nsel, 1)); A = np.zeros(max(nsel, 1))
        roff = np.zeros(nsel + 1, np.int32)
        cap = 4 * int(self.floes.vert_off[-1]) + 64 * nsel + 64
        cx = np.zeros(cap); cy = np.zeros(cap)
        self.ctx.L.szFractureDeform.argtypes = [_vp, _ip, ctypes.c_int, _ip, _dp, _dp, _dp, _ip, _dp, _dp, ctypes.c_int64]
        t0 = time.perf_counter()
        rc = self.ctx.L.szFractureDeform(self.ctx.h, ids.ctypes.data_as(_ip), nsel, changed.ctypes.data_as(_ip), X.ctypes.data_as(_dp),
                                         Y.ctypes.data_as(_dp), A.ctypes.data_as(_dp), roff.ctypes.data_as(_ip), cx.ctypes.data_as(_dp),
                                         cy.ctypes.data_as(_dp), cap)
        self.last_call_ms = (time.perf_counter() - t0) * 1e3
        self.ctx.check(rc, "szFractureDeform")
        rings = [np.stack([cx[roff[s]:roff[s + 1]], cy[roff[s]:roff[s + 1]]]) for s in range(nsel)]
        return changed[:nsel], X[:nsel], Y[:nsel], A[:nsel], rings

    def potential_interactions(self, ids):
        """FloeSimplify.m:14-34 for the floes ids (1-based): list of arrays of 1-based partner numbers, ascending."""
        ids = np.ascontiguousarray(ids, np.int32)
        nsel = len(ids)
        off = np.zeros(nsel + 1, np.int32)
        cap = 64 * nsel + 64
        self.ctx.L.szPotentialInteractions.argtypes = [_vp, _ip, ctypes.c_int, _ip, _ip, ctypes.c_int64]
        while True:
            lst = np.zeros(cap, np.int32)
            rc = self.ctx.L.szPotentialInteractions(self.ctx.h, ids.ctypes.data_as(_ip), nsel, off.ctypes.data_as(_ip), lst.ctypes.data_as(_ip), cap)
            if rc == SZ_ERR_CAPACITY:
                cap *= 4
                continue
            self.ctx.check(rc, "szPotentialInteractions")
            return [lst[off[s]:off[s + 1]].copy() for s in range(nsel)]

    def ghosts(self):
        """(src, parent, Xi, Yi) of the ghost floes of the last step; ghost g is Floe(n + g + 1) of the extended list."""
        cap = 3 * self.floes.n + 8
        src = np.zeros(cap, np.int32); par = np.zeros(cap, np.int32); x = np.zeros(cap); y = np.zeros(cap)
        cnt = ctypes.c_int64()
        self.ctx.check(self.ctx.L.szGhostsDownload(self.ctx.h, src.ctypes.data_as(_ip), par.ctypes.data_as(_ip),
                                                   x.ctypes.data_as(_dp), y.ctypes.data_as(_dp), cap, ctypes.byref(cnt)),
                       "szGhostsDownload")
        g = cnt.value
        return src[:g].copy(), par[:g].copy(), x[:g].copy(), y[:g].copy()

    def download(self, what=SYNC_DYNAMIC):
        self.ctx.check(self.ctx.L.szFloesDownload(self.ctx.h, ctypes.byref(self._c), what), "szFloesDownload")
        return self.floes

    def interactions(self):
        """(inter_off[n+1], rows[K,7]) -- Floe(i).interactions of the last step."""
        n = self.floes.n
        off = np.zeros(n + 1, np.int32)
        cap = 64 * n + 1024
        while True:
            rows = np.zeros((cap, 7))
            rc = self.ctx.L.szInteractionsDownload(self.ctx.h, off.ctypes.data_as(_ip), rows.ctypes.data_as(_dp), cap)
            if rc == SZ_ERR_CAPACITY:
                cap *= 4
                continue
            self.ctx.check(rc, "szInteractionsDownload")
            return off, rows[:off[-1]].copy()

    def pairs(self):
        """Candidate pairs (1-based, extended list) of the last step and N (real + ghost floes)."""
        cap = 64 * self.floes.n + 1024
        while True:
            pi = np.zeros(cap, np.int32); pj = np.zeros(cap, np.int32)
            npairs = ctypes.c_int64(); next_ = ctypes.c_int32()
            rc = self.ctx.L.szPairsDownload(self.ctx.h, pi.ctypes.data_as(_ip), pj.ctypes.data_as(_ip), cap,
                                            ctypes.byref(npairs), ctypes.byref(next_))
            if rc == SZ_ERR_CAPACITY:
                cap *= 4
                continue
            self.ctx.check(rc, "szPairsDownload")
            return np.stack([pi[:npairs.value], pj[:npairs.value]], 1), next_.value


def collision_count(rows):
    """calc_collisionNum.m:6 -- finite-partner rows / 2 + boundary rows."""
    if len(rows) == 0:
        return 0.0
    return float((rows[:, 0] < np.inf).sum()) / 2 + float((rows[:, 0] == np.inf).sum())


def orientation(x, y):
    """clipper(P) orientation mode: True when Area(P) >= 0 (host arithmetic, no GPU)."""
    L = load()
    out = ctypes.c_int()
    x = np.ascontiguousarray(x, np.int64); y = np.ascontiguousarray(y, np.int64)
    rc = L.szOrientation(x, y, len(x), ctypes.byref(out))
    if rc != SZ_OK:
        raise SubzeroError("szOrientation failed (%d)" % rc)
    return bool(out.value)
