"""ctypes binding of include/subzero_hip.h.  No arithmetic here; fails loudly when the HIP library is
missing (there is no CPU fallback by design)."""
import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsubzero_hip.so")

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


# every symbol include/subzero_hip.h declares (tests check the library exports all of them)
EXPORTS = [
    "szAbiVersion", "szCreate", "szDestroy", "szLastError",
    "szClipBatch", "szPolyclipBatch", "szClipBatchDevice", "szOrientation",
    "szDeviceAlloc", "szDeviceFree", "szCopyToDevice", "szCopyToHost", "szSynchronize",
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
    L.szPolyclipBatch.argtypes = [_vp, ctypes.c_int, _f64p, _f64p, _i32p, _f64p, _f64p, _i32p,
                                  ctypes.c_int,
                                  _f64p, _f64p, ctypes.c_int64, _i32p, ctypes.c_int64, _i32p, _i32p]
    L.szClipBatchDevice.argtypes = [_vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ClipStats)]
    L.szOrientation.argtypes = [_i64p, _i64p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.szDeviceAlloc.argtypes = [_vp, ctypes.POINTER(_vp), ctypes.c_int64]
    L.szDeviceFree.argtypes = [_vp, _vp]
    L.szCopyToDevice.argtypes = [_vp, _vp, _vp, ctypes.c_int64]
    L.szCopyToHost.argtypes = [_vp, _vp, _vp, ctypes.c_int64]
    L.szSynchronize.argtypes = [_vp]
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


def orientation(x, y):
    """clipper(P) orientation mode: True when Area(P) >= 0 (host arithmetic, no GPU)."""
    L = load()
    out = ctypes.c_int()
    x = np.ascontiguousarray(x, np.int64); y = np.ascontiguousarray(y, np.int64)
    rc = L.szOrientation(x, y, len(x), ctypes.byref(out))
    if rc != SZ_OK:
        raise SubzeroError("szOrientation failed (%d)" % rc)
    return bool(out.value)
