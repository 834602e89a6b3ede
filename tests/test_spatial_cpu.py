"""The spatially partitioned step (BASELINE north_star) under a real process group on the CPU (gloo, 2 and 3
ranks): subzero_amd/multigpu.SpatialInteractions, unchanged, over the host build of the kernel bodies -- pack / unpack of
spatial_core.hpp included, the same source the GPU runs.

Checked after every step: the ranks' strips partition the floes; every rank's owned floes carry exactly the state, the
contact rows (partners named by floe number; ghost partners by their root floe and position) and the kill / transfer
marks that the single-process step gives them (which the other suites pin bit for bit to the oracle); floes really
change strip, with their stress history and Monte-Carlo points; a rank never holds current data for more than its
strip plus the halo."""
import ctypes
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class HostSpatialBackend:
    """multigpu.SpatialInteractions backend on tests/hostsim (CPU tensors as messages)."""

    def __init__(self, floes, rank, world, edges, halo, axis=0):
        import torch
        import hoststep
        import oracle
        self.torch, self.orc, self.hs = torch, oracle, hoststep
        self.L = hoststep.lib()
        self.S = floes
        self.cf = oracle.pack_floes(floes)
        L = self.L
        L.host_spatial_init.argtypes = [ctypes.POINTER(oracle.CFloes), ctypes.c_int, ctypes.c_int, oracle._dp, ctypes.c_double]
        L.host_spatial_message_bytes.restype = ctypes.c_int64
        L.host_spatial_message_bytes.argtypes = [ctypes.POINTER(oracle.CFloes), ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
        L.host_spatial_pack.argtypes = [ctypes.POINTER(oracle.CFloes), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
        L.host_spatial_unpack.argtypes = [ctypes.POINTER(oracle.CFloes), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                          ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
        L.host_spatial_owned.argtypes = [ctypes.POINTER(ctypes.c_uint8), ctypes.c_int]
        e = np.ascontiguousarray(edges, np.float64)
        hoststep.keep_stress_sums(True)       # the sums of the stress history are device state that lives across steps
        L.host_spatial_axis(int(axis))
        assert L.host_spatial_init(ctypes.byref(self.cf), rank, world, oracle._p(e), halo) == 0
        self._bufs = {}
        self.migrants_in = 0

    def message_bytes(self, caps):
        return int(self.L.host_spatial_message_bytes(ctypes.byref(self.cf), *caps))

    def buffer(self, name, nbytes):
        b = self._bufs.get(name)
        if b is None or b.numel() < nbytes:
            b = self.torch.zeros(int(nbytes) + 64, dtype=self.torch.uint8)
            self._bufs[name] = b
        return b[:nbytes]

    def stream_scope(self):
        import contextlib
        return contextlib.nullcontext()

    def _info(self, raw):
        from subzero_amd.multigpu import SpatialInfo
        i = SpatialInfo()
        (i.left_records, i.left_doubles, i.left_migrants, i.right_records, i.right_doubles, i.right_migrants, i.need_records,
         i.need_doubles, i.need_migrants, i.overflow, i.has_left, i.has_right, i.n_owned, i.n_present) = [int(v) for v in raw]
        return i

    def pack(self, periodic, left, right, caps):
        raw = (ctypes.c_int64 * 14)()
        assert self.L.host_spatial_pack(ctypes.byref(self.cf), int(periodic), ctypes.c_void_p(left.data_ptr()),
                                        ctypes.c_void_p(right.data_ptr()), caps[0], caps[1], caps[2], raw) == 0
        return self._info(raw)

    def unpack(self, from_left, from_right, caps, counts=None, want_info=True):
        self._q_veto = False          # (the caller has looked at the headers: whatever a queued unpack vetoed is settled)
        raw = (ctypes.c_int64 * 14)()
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        assert self.L.host_spatial_unpack(ctypes.byref(self.cf), p(from_left), p(from_right), caps[0], caps[1], caps[2], raw) == 0
        i = self._info(raw)
        self.migrants_in += i.left_migrants + i.right_migrants
        return i

    def header(self, t):
        h = t.view(self.torch.int64).numpy()
        return int(h[0]), int(h[1]), int(h[2]), int(h[3])

    # -- the queued exchange (include/subzero_hip.h szSpatialPackQueued ...) with the device's part played here: a message
    # within its plan is taken, one beyond it is left alone and vetoes the step.  (The host build packs the rows of the
    # migrants with the message: pack_finish has nothing left to do.)  What this exercises under a real process group is
    # multigpu.SpatialInteractions._step_queued: who sends and receives what, and that both ends of an edge agree.
    def pack_queued(self, periodic, left, right, caps, plan_left=None, plan_right=None):
        self._q_sent = self.pack(periodic, left, right, caps)
        self._q_veto = bool(self._q_sent.overflow)

    def pack_finish(self, left, right, caps):
        return self._q_sent

    def unpack_queued(self, from_left, from_right, caps, plan_left, plan_right):
        self._q_in = {"left": (0, 0, 0, 0), "right": (0, 0, 0, 0)}
        take = {"left": None, "right": None}
        for name, t, plan in (("left", from_left, plan_left), ("right", from_right, plan_right)):
            if t is None:
                continue
            h = self.header(t[:64])
            self._q_in[name] = h
            if h[3] == 0 and h[0] <= plan[0] and h[1] <= plan[1] and h[2] <= plan[2]:
                take[name] = t
            else:
                self._q_veto = True
        veto = self._q_veto
        if take["left"] is not None or take["right"] is not None:
            self.unpack(take["left"], take["right"], caps)
        self._q_veto = veto

    def headers_host(self):
        i = self._q_sent
        st = 1 if i.overflow else 0
        out = {"left": (int(i.left_records), int(i.left_doubles), int(i.left_migrants), st),
               "right": (int(i.right_records), int(i.right_doubles), int(i.right_migrants), st)}
        return out, dict(self._q_in), False

    def bulk_row(self):
        return ((self.S.n_stress * 32 + self.S.n_mc * 17 + 7) // 8) * 8

    def step(self, P):
        if getattr(self, "_q_veto", False):
            from subzero_amd import capi
            self._q_veto = False
            raise capi.StepVetoed("vetoed")
        self.result = self.hs.step(self.S, P)
        self.ghosts = self.hs.last_ghosts()
        return self.result

    def owned(self):
        m = np.zeros(self.S.n, np.uint8)
        self.L.host_spatial_owned(m.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), self.S.n)
        return m.astype(bool)


def named_rows(res, ghosts, n, i):
    """Contact rows of floe i with the partner named independently of the numbering of ghosts: (floe number, 0, 0) for a
    real partner, (root floe, x, y) for a ghost, inf for the wall."""
    rows = res.inter[res.inter_off[i]:res.inter_off[i + 1]]
    src, _, gx, gy = ghosts
    out = []
    for r in rows:
        p = r[0]
        if np.isinf(p):
            name = (np.inf, 0.0, 0.0)
        elif p <= n:
            name = (float(p), 0.0, 0.0)
        else:
            g = int(p) - n - 1
            name = (float(src[g]), float(gx[g]), float(gy[g]))
        out.append(name + tuple(r[1:].tolist()))
    return out


def _worker(rank, world, port, scene, steps, drift, q):
    try:
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        import hoststep
        import scenes
        from subzero_amd import multigpu
        dist.init_process_group("gloo", rank=rank, world_size=world)
        kw = dict(scene[2]); tiny = kw.pop("tiny_migrant_room", False); axis = kw.pop("strip_axis", 0); queued = kw.pop("queued", False)
        made = getattr(scenes, scene[0])(*scene[1], **kw)
        S, P = made[0], made[1]
        P.set_reference_ocean()
        (S.Vi if axis else S.Ui)[:] += drift           # a common drift: floes cross strip edges (and the periodic edge)
        ref = S.copy()
        Lx = float(P.c2_boundary[axis].max())
        edges, halo = multigpu.strip_edges(S, world, Lx, axis=axis)
        be = HostSpatialBackend(S, rank, world, edges, halo, axis=axis)
        sim = multigpu.SpatialInteractions(be, rank, world, S.n, int(S.vert_off[-1]))
        if tiny:                                       # room for ONE floe changing strip: messages must grow, on both ends
            sim.caps = (sim.caps[0], sim.caps[1], 1)
        owned0 = be.owned()
        most_present = 0
        for it in range(steps):
            P.doInt = int(it % 4 == 0)
            hoststep.keep_stress_sums(False); be.L.host_spatial_pause(1)
            r1 = hoststep.step(ref, P)                 # the single-process step on the whole field, unmasked
            g1 = hoststep.last_ghosts()
            hoststep.keep_stress_sums(True); be.L.host_spatial_pause(0)
            sim.step(P)
            own = be.owned()
            # the strips partition the floes
            masks = [torch.zeros(S.n, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(masks, torch.from_numpy(own.astype(np.uint8)))
            assert (sum(m.numpy().astype(int) for m in masks) == 1).all(), "step %d: ownership is not a partition" % it
            ids = np.nonzero(own)[0]
            for f in hoststep.STATE_FIELDS + ["Stress"]:
                a, b = np.asarray(getattr(ref, f)), np.asarray(getattr(S, f))
                if f in ("kill", "transfer"):
                    continue
                if f in ("cax", "cay"):
                    for i in ids:
                        assert np.array_equal(a[S.vert_off[i]:S.vert_off[i + 1]], b[S.vert_off[i]:S.vert_off[i + 1]]), (it, f, i)
                    continue
                same = (a[ids] == b[ids]) | (np.isnan(a[ids].astype(float)) & np.isnan(b[ids].astype(float)))
                if f == "Stress":     # (exact sum on one side, rebuilt sum on the other: both the exact sum of the same history)
                    same = np.abs(a[ids] - b[ids]) <= 1e-10 * np.abs(a[ids]) + 1.2e-16 * np.abs(ref.StressH[ids]).sum(1)
                assert same.all(), "step %d: %s of owned floes differs" % (it, f)
            for i in ids:
                assert named_rows(be.result, be.ghosts, S.n, i) == named_rows(r1, g1, S.n, i), "step %d: contact rows of floe %d" % (it, i + 1)
                assert ref.kill[i] == S.kill[i], (it, i)
            if sim.last is not None:
                most_present = max(most_present, int(sim.last[1].n_present))
            if queued:
                sim.count_present = False            # (from the second step on: the exchange with one host round trip per step)
        moved = int((own != owned0).sum())
        dist.destroy_process_group()
        q.put((rank, "ok", moved, be.migrants_in, most_present, int(own.sum()), sim.bytes_sent // max(sim.exchanges, 1), len(r1.inter), sim.regrown, sim.rounds, sim.exchanges, sim.second_rounds))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "fail: %s\n%s" % (e, traceback.format_exc()), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0))


def run(world, scene, steps, drift):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, scene, steps, drift, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    res.sort()
    for r in res:
        assert r[1] == "ok", "rank %d: %s" % (r[0], r[1])
    return res


@pytest.mark.parametrize("world,periodic", [(2, True), (3, True), (2, False)])
def test_spatial_step_equals_single_process(world, periodic):
    n = 2700 if world == 3 else 1800      # (floes small enough against the strips that the halo leaves most of the field out)
    res = run(world, ("pack", (n,), {"periodic": periodic, "n_stress": 6, "n_mc": 80}), 8, 45.0 if periodic else 12.0)
    total = sum(r[5] for r in res)
    assert 0.9 * n <= total <= n
    assert sum(r[3] for r in res) > 0, "no floe changed strip: the migration path was not exercised"
    assert res[0][7] > 100                              # contacts were exercised
    for r in res:
        assert r[4] < total, "a rank held current data for every floe: nothing was sharded"
        # grouped rounds of send / recv: two on the first step, afterwards one per step plus one whenever floes change
        # strip (their 5 KB rows of history and points here -- 49 KB in a model run -- travel separately) or a message
        # outgrows the size agreed from the step before
        assert r[10] == 8 and r[10] + 1 <= r[9] <= 2 * r[10], (r[9], r[10])


@pytest.mark.parametrize("world", [2, 3])
def test_floes_of_unlike_size_and_a_ghost_across_the_strip_ring(world):
    """The halo is per floe -- floe i travels when it lies within rmax_i + max rmax of a strip edge (round 4; 4 max rmax for
    every floe before) -- so a field of unlike sizes is where a floe too few would show: one floe four times the median
    among small ones, sitting on the periodic x edge (its ghost is a partner of the FIRST strip's floes while the last
    strip owns it), everything drifting across the edges.  Owned floes bit-identical to the single-process step."""
    res = run(world, ("big_among_small", (2500,), {"ratio": 4.0, "periodic": True, "centre": (9.2e4, 1.0e4), "sides": 24, "n_stress": 6, "n_mc": 80}), 6, 30.0)
    assert res[0][7] > 100
    total = sum(r[5] for r in res)
    for r in res:
        assert r[4] < total


def test_strips_along_y_equal_single_process():
    """The same partition cut along y (multigpu.strip_axis picks the longer side of the box: the 2e5 x 1e6 m channel of
    BASELINE configs[4]): floes drift in y across the strip edges and the periodic edge."""
    res = run(2, ("pack", (1800,), {"periodic": True, "n_stress": 6, "n_mc": 80, "strip_axis": 1}), 6, 45.0)
    assert sum(r[3] for r in res) > 0 and res[0][7] > 100
    from subzero_amd import multigpu
    assert multigpu.strip_axis(2e5, 1e6) == 1 and multigpu.strip_axis(1e5, 1e5) == 0


def test_messages_grow_when_more_floes_change_strip_than_fit():
    """Room for one migrant per message: the sender packs again into larger buffers, the receiver lays its buffers out
    for what the header announces -- same results, nobody raises, nobody waits for a peer that has left."""
    res = run(2, ("pack", (1800,), {"periodic": True, "n_stress": 6, "n_mc": 80, "tiny_migrant_room": True}), 6, 90.0)
    assert sum(r[3] for r in res) > 4 and sum(r[8] for r in res) > 0


def test_one_round_per_step_when_no_floe_changes_strip():
    """A pack at rest but for its collisions: after the first step every exchange is a single grouped round."""
    res = run(2, ("pack", (1800,), {"periodic": True, "n_stress": 6, "n_mc": 80}), 6, 0.0)
    for r in res:
        assert r[10] == 6 and r[9] == 7, (r[9], r[10])
    assert sum(r[3] for r in res) == 0


@pytest.mark.parametrize("world", [2, 3])
def test_queued_exchange_equals_single_process(world):
    """The exchange with one host round trip per step (SpatialInteractions._step_queued; the device's part -- take a message
    within its plan, veto the step otherwise -- played by the host backend): floes drift across the strip edges, so some
    steps are vetoed on the receiving rank and settled by a second round; owned floes bit-identical to the single-process
    step after every step, ownership a partition, nobody left waiting."""
    n = 2700 if world == 3 else 1800
    res = run(world, ("pack", (n,), {"periodic": True, "n_stress": 6, "n_mc": 80, "queued": True}), 10, 45.0)
    assert sum(r[3] for r in res) > 0, "no floe changed strip"
    assert sum(r[11] for r in res) > 0, "no step of the queued exchange needed its second round"
    for r in res:
        assert r[10] == 10 and r[9] < 2 * r[10] + 1, (r[9], r[10])


def test_queued_exchange_is_one_round_per_step_at_rest():
    res = run(2, ("pack", (1800,), {"periodic": True, "n_stress": 6, "n_mc": 80, "queued": True}), 6, 0.0)
    for r in res:
        assert r[10] == 6 and r[9] == 7 and r[11] == 0, (r[9], r[10], r[11])
