"""ctypes binding of tests/hostsim/libstep_host.so: the device step pipeline (subzero_amd/csrc/step_core.hpp)
compiled for the host.  TEST INFRASTRUCTURE."""
import ctypes
import os
import subprocess
import numpy as np

import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(ROOT, "tests", "hostsim", "libstep_host.so")
    src = os.path.join(ROOT, "tests", "hostsim", "step_host.cpp")
    hdrs = [os.path.join(ROOT, "subzero_amd", "csrc", h) for h in ("step_core.hpp", "contact.hpp", "vatti.hpp", "convex.hpp", "trim.hpp", "detmath.hpp",
                                                                 "team.hpp", "teamclip.hpp", "teamcontact.hpp", "spatial_core.hpp")]
    hdrs.append(os.path.join(ROOT, "tests", "hostsim", "team_host.hpp"))
    if (not os.path.exists(path)) or os.path.getmtime(path) < max(os.path.getmtime(f) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++14", "-fPIC", "-shared", "-ffp-contract=off",
                               "-Wno-unused-function", "-Wno-unknown-pragmas", src, "-o", path])
    L = ctypes.CDLL(path)
    L.host_step.argtypes = [ctypes.POINTER(orc.CFloes), ctypes.POINTER(orc.CParams), orc._ip, orc._dp,
                            ctypes.c_int64, orc._ip, orc._ip, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64),
                            orc._ip, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64)]
    _lib = L
    return L


class NeedsMc(Exception):
    """The step changed nothing: these floes (1-based) are due for forcing with an empty Monte-Carlo mask."""

    def __init__(self, ids):
        Exception.__init__(self, "floes %s need new Monte-Carlo points" % (list(ids),))
        self.ids = np.asarray(ids, np.int32)


def last_front():
    """(times the candidate lists grew, their final capacity, grid levels) of the last `step`."""
    out = (ctypes.c_int * 4)()
    lib().host_last_front(out)
    return out[0], out[1], out[2]


def last_mc_empty():
    ids = np.zeros(1 << 16, np.int32)
    n = lib().host_last_mc_empty(orc._p(ids, orc._ip), len(ids))
    return ids[:n].copy()


def step(S, P, pot_cap=32):
    L = lib()
    cf, cp = orc.pack_floes(S), orc.pack_params(P)
    cap_rows = 64 * S.n + 1024
    cap_pairs = 64 * S.n + 1024
    inter_off = np.zeros(S.n + 1, np.int32); inter = np.zeros((cap_rows, 7))
    pi = np.zeros(cap_pairs, np.int32); pj = np.zeros(cap_pairs, np.int32)
    npairs = ctypes.c_int64(); next_ = ctypes.c_int32(); retried = ctypes.c_int64()
    rc = L.host_step(ctypes.byref(cf), ctypes.byref(cp), orc._p(inter_off, orc._ip), orc._p(inter), cap_rows,
                     orc._p(pi, orc._ip), orc._p(pj, orc._ip), cap_pairs, ctypes.byref(npairs), ctypes.byref(next_),
                     pot_cap, ctypes.byref(retried))
    if rc == 2:
        raise NeedsMc(last_mc_empty())
    if rc != 0:
        raise RuntimeError("host_step failed: %d" % rc)
    r = orc.StepResult()
    r.inter_off = inter_off
    r.inter = inter[:inter_off[-1]].copy()
    r.pairs = np.stack([pi[:npairs.value], pj[:npairs.value]], 1)
    r.n_ext = next_.value
    r.retried = retried.value
    return r


STATE_FIELDS = ["Xi", "Yi", "alpha_i", "Ui", "Vi", "ksi_ice", "h", "area", "mass", "inertia_moment", "rmax",
                "dXi_p", "dYi_p", "dUi_p", "dVi_p", "dalpha_i_p", "dksi_ice_p", "FxOA", "FyOA", "torqueOA",
                "alive", "cax", "cay", "StressCount", "strain", "collision_force", "collision_torque",
                "OverlapArea", "kill", "transfer", "StressH"]


def compare_states(A, B, exact=True, rtol=1e-10):
    """Field-by-field comparison of two FloeStates; returns list of (field, max abs diff, max rel diff)."""
    bad = []
    for k in STATE_FIELDS:
        a, b = np.asarray(getattr(A, k)), np.asarray(getattr(B, k))
        if a.dtype.kind in "iu":
            if not np.array_equal(a, b):
                bad.append((k, int(np.abs(a.astype(np.int64) - b).max()), None))
            continue
        same = (a == b) | (np.isnan(a) & np.isnan(b))
        if same.all():
            continue
        d = np.abs(a - b)
        scale = np.maximum(np.abs(a), np.abs(b))
        rel = np.where(scale > 0, d / np.where(scale > 0, scale, 1), 0)
        if exact or np.nanmax(rel) > rtol:
            bad.append((k, float(np.nanmax(d)), float(np.nanmax(rel))))
    return bad


def set_narrow_mode(mode):
    """0: general sweep only; 1: two-stage compact-record path; 2: single-minimum fast path, a lane per pair;
    3 / 4: the team form first (16 / 32 lanes per pair, the GPU default), its leftovers through mode 2's path."""
    lib().host_set_narrow_mode(int(mode))


def team_counts():
    """(pairs the team pass saw, pairs it handed on, {bail-out site: count}) so far."""
    a, b = ctypes.c_longlong(), ctypes.c_longlong()
    sites = (ctypes.c_longlong * 128)()
    lib().host_team_counts(ctypes.byref(a), ctypes.byref(b), sites)
    return a.value, b.value, {i: int(sites[i]) for i in range(128) if sites[i]}


def fast_counts():
    """(clips the single-minimum fast path completed, clips it handed to the general sweep) so far."""
    h, b = ctypes.c_longlong(), ctypes.c_longlong()
    lib().host_fast_counts(ctypes.byref(h), ctypes.byref(b))
    return h.value, b.value


def set_trim(on):
    """Rings of the general path's intersections cut down to their common box first (trim.hpp); on by default."""
    lib().host_set_trim(int(on))


def trim_counts():
    """(ring vertices handed to the general path's clips, vertices that reached the sweep) so far."""
    b, a = ctypes.c_longlong(), ctypes.c_longlong()
    lib().host_trim_counts(ctypes.byref(b), ctypes.byref(a))
    return b.value, a.value


def set_prune(on):
    """Crossings of long rings skip runs of edges that InterX's first sign test is certain to reject; on by default."""
    lib().host_set_prune(int(on))


def keep_stress_sums(on):
    """The running sums of the stress history are device-only state; on the GPU they live from upload to upload.  With
    this on, successive `step` calls on the same arrays keep them too (off: rebuilt from StressH every call)."""
    lib().host_keep_stress_sums(int(on))


def stress_bound(A, rtol=1e-10):
    """|Stress(device) - Stress(oracle)| allowed per component: rtol relative to the value plus the rounding of the
    reference's own left-to-right mean over the history (<= 1.2e-16 * sum|slots|; the device carries the exact sum)."""
    return rtol * np.abs(A.Stress) + 1.2e-16 * np.abs(A.StressH).sum(1)


def check_stress(A, B, tag=""):
    """Stress of the device state B against the oracle state A (history bit-identical, mean within stress_bound)."""
    assert np.array_equal(A.StressH, B.StressH), tag + " StressH"
    d = np.abs(A.Stress - B.Stress)
    lim = stress_bound(A)
    assert (d <= lim).all(), "%s Stress: worst excess %.3e (value %.3e)" % (tag, float((d - lim).max()), float(np.abs(A.Stress).max()))


def stress_sum_trace(vals, nslots):
    """High word of the stress-sum pair after initialising it with vals[:nslots] and after each replacement of slot
    k % nslots by vals[nslots + k] (the device functions of step_core.hpp, host build)."""
    vals = np.ascontiguousarray(vals, np.float64)
    nsteps = len(vals) - nslots
    out = np.zeros(nsteps + 1)
    L = lib()
    L.host_stress_sum_trace.argtypes = [orc._dp, ctypes.c_int, ctypes.c_int, orc._dp]
    L.host_stress_sum_trace(orc._p(vals), nslots, nsteps, orc._p(out))
    return out


def big_pairs():
    """Pairs handed to the second fast pass (32-vertex blocks) so far."""
    lib().host_big_pairs.restype = ctypes.c_longlong
    return int(lib().host_big_pairs())


def last_ghosts():
    """(src, parent, x, y) of the ghost floes of the last `step` (1-based numbers over the extended list)."""
    cap = 1 << 16
    src = np.zeros(cap, np.int32); par = np.zeros(cap, np.int32); x = np.zeros(cap); y = np.zeros(cap)
    L = lib()
    L.host_last_ghosts.argtypes = [orc._ip, orc._ip, orc._dp, orc._dp, ctypes.c_int]
    g = L.host_last_ghosts(orc._p(src, orc._ip), orc._p(par, orc._ip), orc._p(x), orc._p(y), cap)
    return src[:g].copy(), par[:g].copy(), x[:g].copy(), y[:g].copy()
