"""Multi-GPU form of the step: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI)
for the one real exchange of the path.

Design (round 1).  Every rank holds the full Floe struct array on its GPU and runs the cheap stages
(ghost floes, grid, broad-phase, pair list; contact rows, sums, calc_trajectory) redundantly -- they are
deterministic, so all ranks hold identical data.  The dominant stage, the per-pair narrow-phase
(`k_narrow`, ~70 % of a step), is cut into `world` contiguous segments of the candidate-pair list; rank r
computes segment r and the segments are exchanged with ONE all-gather of (rows-per-pair, flags, packed
contact rows) -- ~0.1-0.5 MB per rank at 1e4 floes -- preceded by an 16-byte all-gather of the row counts
that sizes it.  On the steps that refresh the ocean / wind forcing (every 10th) that stage -- the largest of
the replicated ones, 1000 Monte-Carlo points per floe -- is sharded by floe range as well, with one more
all-gather of 24 bytes per floe.  After the exchange every rank continues with the same global pair list and the same row
order as the single-GPU step, so results are bit-identical to it by construction (tests/test_multigpu_cpu.py
checks exactly that under a 2-process gloo group, with the host build of the same kernel bodies).

bench.py reports WEAK scaling (1e4 floes per GPU, box area grows with the GPU count): the narrow-phase is a
latency chain of ~1.6 ms per pair whatever the number of pairs, so cutting a fixed 4.4e4-pair list in 8 would
not shorten it; more floes per step is the axis along which this design scales.  The replicated stages grow
with the total floe count and bound the efficiency.  The planned next step is spatial ownership of floes
(contiguous index ranges after a spatial sort) with the per-floe stages sharded too and a second all-gather of
the dynamic state; see DESIGN.md section 6.

The orchestration below is backend-agnostic: `GpuBackend` drives the C ABI (szStepShardBegin / Export /
Finish) with torch CUDA tensors as exchange buffers; the CPU tests plug in a backend built on tests/hostsim.
"""
import ctypes
import json
import os
import sys
import time

import numpy as np


class ShardInfo(ctypes.Structure):
    """szShardInfo"""
    _fields_ = [("n_pairs", ctypes.c_int64), ("seg_lo", ctypes.c_int64), ("seg_hi", ctypes.c_int64),
                ("seg_max", ctypes.c_int64), ("seg_rows", ctypes.c_int64), ("status", ctypes.c_int32),
                ("replicated_failure", ctypes.c_int32)]


class GpuBackend:
    """One GPU of the job: szContext + device-resident floes + torch CUDA tensors as exchange buffers."""

    def __init__(self, floes, local_rank):
        import torch
        from . import capi
        self.torch, self.capi = torch, capi
        torch.cuda.set_device(local_rank)
        self.device = torch.device("cuda", local_rank)
        self.ctx = capi.Context(local_rank)
        self.sim = capi.FloeInteractions(self.ctx, floes)
        self._packer = capi.ParamPacker()
        L = self.ctx.L
        L.szShardSegmentBytes.restype = ctypes.c_int64
        L.szShardSegmentBytes.argtypes = [ctypes.c_int64, ctypes.c_int64]
        L.szStepShardBegin.argtypes = [ctypes.c_void_p, ctypes.POINTER(capi.StepParamsC), ctypes.c_int, ctypes.c_int,
                                       ctypes.POINTER(ShardInfo)]
        L.szStepShardExport.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32]
        L.szStepShardFinish.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.POINTER(capi.StepStats), ctypes.POINTER(ctypes.c_int32),
                                        ctypes.POINTER(ctypes.c_int64)]
        # kernels and collectives on one stream: the exchange is then ordered by the stream, not by the host
        self.stream = torch.cuda.Stream(self.device)
        self.ctx.check(L.szSetStream(self.ctx.h, ctypes.c_void_p(self.stream.cuda_stream)), "szSetStream")
        L.szShardForcingBytes.restype = ctypes.c_int64
        L.szShardForcingBytes.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.szStepShardForcing.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.szStepShardEnd.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(capi.StepStats)]
        self._bufs = {}

    def begin(self, P, rank, world):
        self._cp = self._packer.pack(P)
        info = ShardInfo()
        rc = self.ctx.L.szStepShardBegin(self.ctx.h, ctypes.byref(self._cp), rank, world, ctypes.byref(info))
        self._err = self.ctx.L.szLastError(self.ctx.h).decode() if rc else ""
        return info

    def last_error(self):
        return self._err

    def segment_bytes(self, seg_max, rows_max):
        return int(self.ctx.L.szShardSegmentBytes(seg_max, rows_max))

    def buffer(self, name, nbytes):
        b = self._bufs.get(name)
        if b is None or b.numel() < nbytes:
            b = self.torch.empty(int(nbytes * 1.25) + 256, dtype=self.torch.uint8, device=self.device)
            self._bufs[name] = b
        return b[:nbytes]

    def stream_scope(self):
        """Context manager under which the caller issues its collectives (torch's current stream = ours)."""
        return self.torch.cuda.stream(self.stream)

    def export(self, send, seg_max, rows_max, status):
        self.ctx.check(self.ctx.L.szStepShardExport(self.ctx.h, ctypes.c_void_p(send.data_ptr()), seg_max, rows_max,
                                                    int(status)), "szStepShardExport")

    def finish(self, recv, seg_max, rows_max):
        """-> (code, rows_needed, stats, forcing_exchange_pending); code 0, or capi.SHARD_RESIZE, or an error"""
        self._st = self.capi.StepStats()
        fx = ctypes.c_int32(0)
        need = ctypes.c_int64(0)
        rc = self.ctx.L.szStepShardFinish(self.ctx.h, ctypes.c_void_p(recv.data_ptr()), seg_max, rows_max,
                                          ctypes.byref(self._st), ctypes.byref(fx), ctypes.byref(need))
        self._err = self.ctx.L.szLastError(self.ctx.h).decode() if rc < 0 else ""
        return rc, int(need.value), self._st, bool(fx.value)

    def forcing_bytes(self, world):
        return int(self.ctx.L.szShardForcingBytes(self.ctx.h, world))

    def forcing_export(self, send):
        self.ctx.check(self.ctx.L.szStepShardForcing(self.ctx.h, ctypes.c_void_p(send.data_ptr())), "szStepShardForcing")

    def end(self, recv):
        self.ctx.check(self.ctx.L.szStepShardEnd(self.ctx.h, ctypes.c_void_p(recv.data_ptr()), ctypes.byref(self._st)),
                       "szStepShardEnd")
        return self._st

    def close(self):
        self.ctx.close()


class ShardedInteractions:
    """floe_interactions_all over `world` ranks; `step` must be called by every rank with the same P."""

    def __init__(self, backend, rank, world, group=None):
        import torch.distributed as dist
        self.dist, self.backend, self.rank, self.world, self.group = dist, backend, rank, world, group

    SHARD_RESIZE = 1   # szStepShardFinish: a segment has more contact rows than the buffers were sized for
    min_rows = 64      # smallest row capacity of a segment buffer

    def step(self, P):
        be, dist = self.backend, self.dist
        with be.stream_scope():
            return self._step(P, be, dist)

    def _step(self, P, be, dist):
        info = be.begin(P, self.rank, self.world)
        if info.status and info.replicated_failure:
            # the replicated stages failed: the same error on every rank, nobody enters the exchange
            raise RuntimeError("sharded step failed in the replicated stages (%d): %s" % (info.status, be.last_error()))
        # Capacities of a segment, the same on every rank (they follow from values every rank has): pairs rounded up
        # so that the layout -- and with it the buffers -- changes rarely; rows from the last step, kept while
        # they fit with some room, and a segment that does not fit is reported by every rank's Finish, after which
        # the exchange is repeated with the size that does.
        seg_max = (int(info.seg_max) + 1023) // 1024 * 1024
        seen = getattr(self, "_rows_seen", seg_max)
        rows_max = getattr(self, "_rows_max", 0)
        if rows_max < 1.1 * seen + 16 or rows_max > 4 * seen + 4096 or rows_max < self.min_rows:
            rows_max = max(self.min_rows, int(1.25 * seen) + 64)
        while True:
            if (seg_max, rows_max) != getattr(self, "_sized_for", None):   # same sizes as last step: same views
                nbytes = be.segment_bytes(seg_max, rows_max)
                self._send = be.buffer("send", nbytes)
                self._recv = be.buffer("recv", nbytes * self.world)
                self._sized_for = (seg_max, rows_max)
            send, recv = self._send, self._recv
            be.export(send, seg_max, rows_max, info.status)
            dist.all_gather_into_tensor(recv, send, group=self.group)
            rc, need, st, forcing_pending = be.finish(recv, seg_max, rows_max)
            if rc == self.SHARD_RESIZE:
                rows_max = int(1.25 * need) + 64
                continue
            if rc != 0:   # every rank sees the same headers and stops together: no rank is left in a collective
                raise RuntimeError("sharded step failed (%d): %s" % (rc, be.last_error()))
            self._rows_seen, self._rows_max = need, rows_max
            break
        if forcing_pending:
            # forcing step: every rank evaluated the ocean / wind forcing of its own floe range
            fb = be.forcing_bytes(self.world)
            fsend = be.buffer("fsend", fb)
            frecv = be.buffer("frecv", fb * self.world)
            be.forcing_export(fsend)
            dist.all_gather_into_tensor(frecv, fsend, group=self.group)
            st = be.end(frecv)
        self.last_info = info
        return st


# =====================================================================================================================
# Spatial partition (BASELINE north_star): floes owned by x-strips, one neighbour exchange per step.
# The design and the proof sketch of its exactness are in subzero_amd/csrc/spatial_core.hpp.
# =====================================================================================================================
class SpatialInfo(ctypes.Structure):
    """szSpatialInfo"""
    _fields_ = [(k, ctypes.c_int64) for k in ("left_records", "left_doubles", "left_migrants", "right_records",
                                              "right_doubles", "right_migrants", "need_records", "need_doubles",
                                              "need_migrants", "n_owned", "n_present")] + \
               [(k, ctypes.c_int32) for k in ("overflow", "has_left", "has_right", "reserved")]


def strip_axis(Lx, Ly):
    """The axis the domain is cut along: its longer one (0 = x, 1 = y), so that the strips are as wide against the
    halo as the box allows (the 2e5 x 1e6 m channel of initialize_boundaries.m:14-16 is cut along y)."""
    return 1 if Ly > Lx else 0


def strip_edges(floes, world, Lx, halo=None, axis=0):
    """Edges of `world` strips over [-Lx, Lx] along `axis` (Lx = the half-extent of the box along it): floe-count
    quantiles of the centroids, moved onto multiples of the broad-phase bin (2 max rmax) and kept at least `halo`
    (default: the reach of a pair, 2 max rmax -- what a floe of the largest rmax needs; a floe of radius r travels within
    r + max rmax of an edge, spatial_core.hpp) wide.  Every rank computes the same edges from the same uploaded array."""
    rmax = float(np.max(floes.rmax))
    bin_ = 2 * rmax
    halo = 2 * rmax if halo is None else halo
    c = floes.Yi if axis else floes.Xi
    x = np.sort(c[np.isfinite(floes.Xi) & np.isfinite(floes.Yi) & (floes.alive > 0)])
    edges = [-float(Lx)]
    for k in range(1, world):
        q = x[min(len(x) - 1, (len(x) * k) // world)] if len(x) else -Lx + 2 * Lx * k / world
        e = -Lx + round((q + Lx) / bin_) * bin_
        edges.append(float(min(max(e, edges[-1] + halo), Lx - (world - k) * halo)))
    edges.append(float(Lx))
    if world > 1 and min(np.diff(edges)) < halo:
        raise ValueError("the domain is too narrow for %d strips with a %.0f m halo (use fewer ranks)" % (world, halo))
    return np.asarray(edges, np.float64), float(halo)


class GpuSpatialBackend:
    """One GPU of the spatially sharded job: szContext, device-resident array, torch CUDA tensors as messages."""

    def __init__(self, floes, local_rank, rank, world, edges, halo, axis=0):
        import torch
        from . import capi
        self.torch, self.capi = torch, capi
        torch.cuda.set_device(local_rank)
        self.device = torch.device("cuda", local_rank)
        self.ctx = capi.Context(local_rank)
        self.sim = capi.FloeInteractions(self.ctx, floes)
        L = self.ctx.L
        L.szSpatialInit.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_double]
        L.szSpatialMessageBytes.restype = ctypes.c_int64
        L.szSpatialMessageBytes.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
        L.szSpatialPack.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                    ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(SpatialInfo)]
        L.szSpatialUnpack.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                      ctypes.c_int64, ctypes.POINTER(SpatialInfo)]
        L.szSpatialOwned.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8)]
        self.stream = torch.cuda.Stream(self.device)
        self.ctx.check(L.szSetStream(self.ctx.h, ctypes.c_void_p(self.stream.cuda_stream)), "szSetStream")
        e = np.ascontiguousarray(edges, np.float64)
        self.ctx.check(L.szSpatialSetAxis(self.ctx.h, int(axis)), "szSpatialSetAxis")
        self.ctx.check(L.szSpatialInit(self.ctx.h, rank, world, e.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), halo), "szSpatialInit")
        self._bufs = {}

    def message_bytes(self, caps):
        return int(self.ctx.L.szSpatialMessageBytes(self.ctx.h, *caps))

    def buffer(self, name, nbytes):
        b = self._bufs.get(name)
        if b is None or b.numel() < nbytes:
            b = self.torch.zeros(int(nbytes * 1.25) + 256, dtype=self.torch.uint8, device=self.device)
            self._bufs[name] = b
        return b[:nbytes]

    def stream_scope(self):
        return self.torch.cuda.stream(self.stream)

    def pack(self, periodic, left, right, caps):
        info = SpatialInfo()
        self.ctx.check(self.ctx.L.szSpatialPack(self.ctx.h, int(periodic), ctypes.c_void_p(left.data_ptr()),
                                                ctypes.c_void_p(right.data_ptr()), caps[0], caps[1], caps[2], ctypes.byref(info)),
                       "szSpatialPack")
        return info

    def unpack(self, from_left, from_right, caps, counts=None, want_info=True):
        """counts: {"left": (records, doubles, migrants), "right": ...} when the caller has read the headers already;
        want_info False: no count of owned / present floes either -- the unpack is then queued without a host round trip."""
        info = SpatialInfo() if want_info else None
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        if counts is None:
            self.ctx.check(self.ctx.L.szSpatialUnpack(self.ctx.h, p(from_left), p(from_right), caps[0], caps[1], caps[2],
                                                      ctypes.byref(info)), "szSpatialUnpack")
            return info
        arr = lambda c: (ctypes.c_int64 * 3)(*c) if c is not None else None
        self.ctx.L.szSpatialUnpackKnown.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                                    ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(SpatialInfo)]
        self.ctx.check(self.ctx.L.szSpatialUnpackKnown(self.ctx.h, p(from_left), p(from_right), caps[0], caps[1], caps[2],
                                                       arr(counts.get("left")), arr(counts.get("right")),
                                                       ctypes.byref(info) if want_info else None), "szSpatialUnpackKnown")
        return info

    def step(self, P):
        return self.sim.step(P)

    # -- the exchange with one host round trip per step (include/subzero_hip.h szSpatialPackQueued ...) --
    def pack_queued(self, periodic, left, right, caps, plan_left=None, plan_right=None):
        L = self.ctx.L
        L.szSpatialPackQueued.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                          ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        arr = lambda c: (ctypes.c_int64 * 3)(*c) if c is not None else None
        self.ctx.check(L.szSpatialPackQueued(self.ctx.h, int(periodic), ctypes.c_void_p(left.data_ptr()), ctypes.c_void_p(right.data_ptr()),
                                             caps[0], caps[1], caps[2], arr(plan_left), arr(plan_right)), "szSpatialPackQueued")

    def pack_finish(self, left, right, caps):
        L = self.ctx.L
        L.szSpatialPackFinish.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(SpatialInfo)]
        info = SpatialInfo()
        self.ctx.check(L.szSpatialPackFinish(self.ctx.h, ctypes.c_void_p(left.data_ptr()), ctypes.c_void_p(right.data_ptr()), caps[0], caps[1], caps[2],
                                             ctypes.byref(info)), "szSpatialPackFinish")
        return info

    def unpack_queued(self, from_left, from_right, caps, plan_left, plan_right):
        L = self.ctx.L
        L.szSpatialUnpackQueued.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                            ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        arr = lambda c: (ctypes.c_int64 * 3)(*c) if c is not None else None
        self.ctx.check(L.szSpatialUnpackQueued(self.ctx.h, p(from_left), p(from_right), caps[0], caps[1], caps[2], arr(plan_left), arr(plan_right)),
                       "szSpatialUnpackQueued")

    def headers_host(self):
        """After the step that followed pack_queued / unpack_queued: ({"left": hdr, "right": hdr} of my messages, the same of
        the received ones, vetoed); hdr = (records, doubles, migrants, status)."""
        L = self.ctx.L
        L.szSpatialHeaders.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32)]
        h = (ctypes.c_int64 * 32)(); v = ctypes.c_int32(0)
        self.ctx.check(L.szSpatialHeaders(self.ctx.h, h, ctypes.byref(v)), "szSpatialHeaders")
        q = lambda k: (int(h[8 * k]), int(h[8 * k + 1]), int(h[8 * k + 2]), int(h[8 * k + 3]))
        return {"left": q(0), "right": q(1)}, {"left": q(2), "right": q(3)}, bool(v.value)

    def header(self, t):
        """(records, doubles, migrants, status) of a received message header (a 64-byte device tensor)."""
        h = t.view(self.torch.int64).cpu().numpy()
        return int(h[0]), int(h[1]), int(h[2]), int(h[3])

    def headers(self, ts):
        """header() of several received messages with ONE read-back (one host wait per step for both neighbours' messages)."""
        h = self.torch.stack([t.view(self.torch.int64) for t in ts]).cpu().numpy()
        return [(int(r[0]), int(r[1]), int(r[2]), int(r[3])) for r in h]

    def bulk_row(self):
        S = self.sim.floes
        return ((S.n_stress * 32 + S.n_mc * 17 + 7) // 8) * 8

    def owned(self):
        m = np.zeros(self.sim.floes.n, np.uint8)
        self.ctx.check(self.ctx.L.szSpatialOwned(self.ctx.h, m.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))), "szSpatialOwned")
        return m.astype(bool)

    def close(self):
        self.ctx.close()


class SpatialInteractions:
    """floe_interactions_all over `world` ranks, each owning the floes of one x-strip; `step` is called by every rank
    with the same P.  Per step: pack -> exchange with the two neighbours (a ring when periodic) -> unpack -> step.
    The exchange is two grouped rounds of send / recv: the 64-byte headers (what each message holds), then exactly
    the used part of the three arrays of each message (record offsets, record doubles, bulk rows of floes that
    changed strip) -- nothing is sized by guesswork, nothing is padded on the wire."""

    def __init__(self, backend, rank, world, n_floes, n_vertices, group=None, count_present=True):
        import torch.distributed as dist
        self.dist, self.be, self.rank, self.world, self.group = dist, backend, rank, world, group
        self.count_present = count_present   # False: no owned / present counts per step (they cost a host round trip)
        # local capacities of the message buffers: every floe could be in a message; a few dozen change strip at most
        self.caps = (int(n_floes), int(47 * n_floes + 2 * n_vertices), int(max(64, n_floes // 16)))
        self.exchanges = 0
        self.bytes_sent = 0
        self.regrown = 0
        self.rounds = 0                      # grouped send / recv rounds so far (one per step in steady state)
        self.second_rounds = 0               # steps of the queued exchange that needed a second round
        # the exchange with one host round trip per step, when the backend has it (SUBZERO_SPATIAL_QUEUED=0: never)
        self.queued = hasattr(backend, "pack_queued") and os.environ.get("SUBZERO_SPATIAL_QUEUED", "1") != "0"
        self.last_out, self.last_in = {}, {}   # per edge: (records, doubles, migrants) of the last message sent / received
        self.last = None

    def _views(self, buf, caps, bulk_row):
        """(header, record offsets, doubles, bulk) sub-tensors of a message buffer (layout of spatial_core.hpp sp_msg_view)."""
        o_rec = 64
        o_dbl = o_rec + ((4 * caps[0] + 7) // 8) * 8
        o_bulk = o_dbl + 8 * caps[1]
        return buf[:64], buf[o_rec:o_dbl], buf[o_dbl:o_bulk], buf[o_bulk:o_bulk + caps[2] * bulk_row]

    def _round(self, ops):
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()

    @staticmethod
    def _plan(c):
        """Transfer size agreed for this step on an edge, from what travelled over it last step (records, doubles,
        migrants): both ends evaluate this on the same numbers."""
        # (no room is agreed for the 49 KB rows of floes that change strip: they are rare and travel in a second round)
        return (c[0] + c[0] // 4 + 16, c[1] + c[1] // 4 + 1024, 0)

    def step(self, P):
        """One step of the partitioned job.  With a backend that has the queued entry points (the GPU one) and an agreed
        size for every edge -- from the second step on -- the host touches the device ONCE per step (`_step_queued`);
        otherwise, and while the caller wants the counts of owned / present floes, `_step_classic`."""
        if self.world > 1 and self.queued and not self.count_present and self._edges_planned(P):
            return self._step_queued(P)
        return self._step_classic(P)

    @staticmethod
    def _plan_queued(c):
        """_plan for the queued exchange: room for the rows of a couple of floes changing strip travels with the first round
        (49 KB each in a model run: microseconds over xGMI), so that the odd floe crossing an edge -- at sea-ice drift speeds
        a step in five or ten sees one -- does not veto the receiver's step."""
        return (c[0] + c[0] // 4 + 16, c[1] + c[1] // 4 + 1024, max(2, c[2] + 1))

    def _my_edges(self, P):
        left, right = (self.rank - 1) % self.world, (self.rank + 1) % self.world
        hl = bool(P.PERIODIC) or self.rank > 0
        hr = bool(P.PERIODIC) or self.rank < self.world - 1
        # (edge, peer, key of my send buffer, tag of what I send, key of my receive buffer, tag of what I receive): see _step_classic
        return [e for e in (("left", left, "send_left", 1, "from_left", 2), ("right", right, "send_right", 2, "from_right", 1))
                if (hl if e[0] == "left" else hr)]

    def _edges_planned(self, P):
        return all(e[0] in self.last_out and e[0] in self.last_in for e in self._my_edges(P))

    def _step_queued(self, P):
        """pack -> exchange -> unpack -> step with nothing read back in between (include/subzero_hip.h szSpatialPackQueued):
        the same first round as `_step_classic`'s steady state -- header + the agreed leading part of the two record arrays,
        each way -- but the DEVICE reads the received headers: a message within its plan is taken, one beyond it (or with
        a floe that changed strip, whose rows travel separately) vetoes the step, which then changes nothing.  After the
        step's own read-back the host reads the four headers from pinned memory; both ends of an edge see the same numbers
        and so agree on whether a second round is due -- the sender packs the rows of its migrants and sends the rest, the
        vetoed receiver unpacks and steps again.  Steady state: one host round trip per step instead of three."""
        from . import capi
        be, dist = self.be, self.dist
        with be.stream_scope():
            bulk_row = be.bulk_row()
            edges = self._my_edges(P)
            plan_out = {e[0]: self._plan_queued(self.last_out[e[0]]) for e in edges}
            plan_in = {e[0]: self._plan_queued(self.last_in[e[0]]) for e in edges}
            for pl in list(plan_out.values()) + list(plan_in.values()):
                self.caps = tuple(max(a, b) for a, b in zip(self.caps, pl))
            caps = self.caps
            nbytes = be.message_bytes(caps)
            bufs = {k: be.buffer(k, nbytes) for k in ("send_left", "send_right", "from_left", "from_right")}
            be.pack_queued(P.PERIODIC, bufs["send_left"], bufs["send_right"], caps, plan_out.get("left"), plan_out.get("right"))
            v = {k: self._views(b, caps, bulk_row) for k, b in bufs.items()}
            recv_order = list(reversed(edges))

            def parts(key, cnt):
                _, rec, dbl, bulk = v[key]
                out = []
                if cnt[0]: out.append(rec[:4 * cnt[0]])
                if cnt[1]: out.append(dbl[:8 * cnt[1]])
                if cnt[2]: out.append(bulk[:cnt[2] * bulk_row])
                return out

            ops = []
            for e in edges:
                ops.append(dist.P2POp(dist.isend, v[e[2]][0], e[1], self.group, tag=e[3])); self.bytes_sent += 64
                for t in parts(e[2], plan_out[e[0]]):
                    ops.append(dist.P2POp(dist.isend, t, e[1], self.group, tag=e[3])); self.bytes_sent += t.numel()
            for e in recv_order:
                ops.append(dist.P2POp(dist.irecv, v[e[4]][0], e[1], self.group, tag=e[5]))
                for t in parts(e[4], plan_in[e[0]]):
                    ops.append(dist.P2POp(dist.irecv, t, e[1], self.group, tag=e[5]))
            self._round(ops)
            self.rounds += 1
            names = [e[0] for e in edges]
            be.unpack_queued(bufs["from_left"] if "left" in names else None, bufs["from_right"] if "right" in names else None, caps,
                             plan_in.get("left"), plan_in.get("right"))
            vetoed, pending, st = False, None, None
            try:
                st = be.step(P)
            except capi.StepVetoed:
                vetoed = True
            except capi.NeedsMc as exc:           # (the step changed nothing and is rank-local from here on: retry_local)
                pending = exc
            out_h, in_h, _ = be.headers_host()
            if any(out_h[k][3] != 0 or in_h[k][3] != 0 for k in names):
                # more floes changed strip across one edge in one step than a message has room for (max(64, n / 16)): the
                # queued exchange has no repair for that (the classic one packs again into larger buffers before it sends)
                be.pack_finish(bufs["send_left"], bufs["send_right"], caps)       # (gives the ownership back)
                raise capi.SubzeroError("spatial exchange: a message overflowed its capacities in the queued exchange "
                                        "(SUBZERO_SPATIAL_QUEUED=0 selects the classic exchange, which grows them)")
            out_cnt = {k: out_h[k][:3] for k in names}
            hdr = {k: in_h[k][:3] for k in names}

            def rest(cnt, plan):
                big = cnt[0] > plan[0] or cnt[1] > plan[1]
                return (cnt[0] if big else 0, cnt[1] if big else 0, cnt[2] if cnt[2] > plan[2] else 0)
            rest_out = {k: rest(out_cnt[k], plan_out[k]) for k in names}
            rest_in = {k: rest(hdr[k], plan_in[k]) for k in names}
            second = any(any(c) for c in rest_out.values()) or any(any(c) for c in rest_in.values())
            rcaps = caps
            if vetoed and not any(any(c) for c in rest_in.values()):
                raise capi.SubzeroError("spatial exchange: the step was vetoed but no received message is beyond its plan")
            if second:
                self.second_rounds += 1
                if any(any(c) for c in rest_out.values()) and any(c[2] for c in out_cnt.values()):
                    be.pack_finish(bufs["send_left"], bufs["send_right"], caps)   # the rows of my floes that changed strip (all of them, again)
                rcaps = tuple(max([caps[q]] + [hdr[k][q] for k in names]) for q in range(3))
                if rcaps != caps:                                 # a neighbour's message is larger than my buffers
                    rbytes = be.message_bytes(rcaps)
                    for e in edges:
                        k = e[4]
                        old = [t.clone() for t in v[k]]
                        bufs[k] = be.buffer(k, rbytes)
                        v[k] = self._views(bufs[k], rcaps, bulk_row)
                        for a, b in zip(v[k], old):
                            m = min(a.numel(), b.numel())
                            a[:m].copy_(b[:m])
                    self.regrown += 1
                ops = []
                for e in edges:
                    for t in parts(e[2], rest_out[e[0]]):
                        ops.append(dist.P2POp(dist.isend, t, e[1], self.group, tag=e[3])); self.bytes_sent += t.numel()
                for e in recv_order:
                    for t in parts(e[4], rest_in[e[0]]):
                        ops.append(dist.P2POp(dist.irecv, t, e[1], self.group, tag=e[5]))
                self._round(ops)
                self.rounds += 1
                if rcaps != caps:
                    # (my send buffers keep their layout for this step; from the next one every buffer has the larger one)
                    self.caps = tuple(max(a, b) for a, b in zip(self.caps, rcaps))
            for k in names:
                self.last_out[k] = out_cnt[k]; self.last_in[k] = hdr[k]
            self.exchanges += 1
            self.last = None
            if vetoed:
                be.unpack(bufs["from_left"] if "left" in names else None, bufs["from_right"] if "right" in names else None, rcaps,
                          counts=hdr, want_info=False)
                st = be.step(P)
            if pending is not None:
                raise pending
            return st

    def _step_classic(self, P):
        """pack -> exchange -> unpack -> step.  In steady state the exchange is ONE grouped round of send / recv per step:
        each message travels as header + the leading part of its three arrays, sized from what the same edge carried
        last step plus a margin (both ends know that number); the header says how much of it is used.  Only the
        first step, and a step whose message outgrew the agreed size, take a second round with the exact sizes."""
        be, dist = self.be, self.dist
        if self.world == 1:
            return be.step(P)
        left, right = (self.rank - 1) % self.world, (self.rank + 1) % self.world
        with be.stream_scope():
            bulk_row = be.bulk_row()
            plan_out = {k: self._plan(c) for k, c in self.last_out.items()}
            plan_in = {k: self._plan(c) for k, c in self.last_in.items()}
            for pl in list(plan_out.values()) + list(plan_in.values()):      # my buffers hold what is agreed to travel
                self.caps = tuple(max(a, b) for a, b in zip(self.caps, pl))
            # my messages: a message that does not fit (more floes changed strip than there is room for) is packed
            # again into larger buffers -- nothing has been sent yet, nothing is lost (include/subzero_hip.h)
            while True:
                caps = self.caps
                nbytes = be.message_bytes(caps)
                bufs = {k: be.buffer(k, nbytes) for k in ("send_left", "send_right", "from_left", "from_right")}
                sent = be.pack(P.PERIODIC, bufs["send_left"], bufs["send_right"], caps)
                if not sent.overflow:
                    break
                self.caps = (max(caps[0], int(sent.need_records)), max(caps[1], int(sent.need_doubles)),
                             max(2 * caps[2], int(sent.need_migrants)))
                self.regrown += 1
            hl, hr = bool(sent.has_left), bool(sent.has_right)
            v = {k: self._views(b, caps, bulk_row) for k, b in bufs.items()}
            out_cnt = {"left": (int(sent.left_records), int(sent.left_doubles), int(sent.left_migrants)),
                       "right": (int(sent.right_records), int(sent.right_doubles), int(sent.right_migrants))}
            # (edge, peer, key of my send buffer, tag of what I send, key of my receive buffer, tag of what I receive);
            # tags name the direction of travel: on a ring of two both neighbours are the same peer, and the order of
            # posting below -- sends leftward then rightward, receives from the right then from the left -- matches on
            # both ends for backends that match by order
            edges = [e for e in (("left", left, "send_left", 1, "from_left", 2), ("right", right, "send_right", 2, "from_right", 1))
                     if (hl if e[0] == "left" else hr)]
            recv_order = list(reversed(edges))

            def parts(key, cnt):
                _, rec, dbl, bulk = v[key]
                out = []
                if cnt[0]: out.append(rec[:4 * cnt[0]])
                if cnt[1]: out.append(dbl[:8 * cnt[1]])
                if cnt[2]: out.append(bulk[:cnt[2] * bulk_row])
                return out

            def relayout(rcaps):
                """receive buffers laid out for larger capacities; what has arrived of each message moves along"""
                rbytes = be.message_bytes(rcaps)
                for e in edges:
                    k = e[4]
                    old = [t.clone() for t in v[k]]
                    bufs[k] = be.buffer(k, rbytes)
                    v[k] = self._views(bufs[k], rcaps, bulk_row)
                    for a, b in zip(v[k], old):
                        m = min(a.numel(), b.numel())
                        a[:m].copy_(b[:m])
                self.regrown += 1

            steady = all(e[0] in plan_out and e[0] in plan_in for e in edges)
            rcaps = caps
            ops = []
            for e in edges:                                   # headers always travel
                ops.append(dist.P2POp(dist.isend, v[e[2]][0], e[1], self.group, tag=e[3])); self.bytes_sent += 64
                if steady:
                    for t in parts(e[2], plan_out[e[0]]):
                        ops.append(dist.P2POp(dist.isend, t, e[1], self.group, tag=e[3])); self.bytes_sent += t.numel()
            for e in recv_order:
                ops.append(dist.P2POp(dist.irecv, v[e[4]][0], e[1], self.group, tag=e[5]))
                if steady:
                    for t in parts(e[4], plan_in[e[0]]):
                        ops.append(dist.P2POp(dist.irecv, t, e[1], self.group, tag=e[5]))
            self._round(ops)
            self.rounds += 1
            if hasattr(be, "headers"):
                hdr = {e[0]: h[:3] for e, h in zip(edges, be.headers([v[e[4]][0] for e in edges]))}
            else:
                hdr = {e[0]: be.header(v[e[4]][0])[:3] for e in edges}
            # second round: everything on the first step; afterwards, per edge, the two arrays of records in full when
            # they outgrew what was agreed, and the rows of the floes that changed strip when there are any
            def rest(cnt, plan):
                if not steady: return cnt
                big = cnt[0] > plan[0] or cnt[1] > plan[1]
                return (cnt[0] if big else 0, cnt[1] if big else 0, cnt[2] if cnt[2] > plan[2] else 0)
            rest_out = {e[0]: rest(out_cnt[e[0]], plan_out.get(e[0])) for e in edges}
            rest_in = {e[0]: rest(hdr[e[0]], plan_in.get(e[0])) for e in edges}
            if any(any(c) for c in rest_out.values()) or any(any(c) for c in rest_in.values()):
                need = tuple(max([rcaps[q]] + [hdr[e[0]][q] for e in edges]) for q in range(3))
                if need != rcaps:                             # a neighbour's message is larger than my buffers: lay them out again
                    rcaps = need
                    relayout(rcaps)
                    self.caps = tuple(max(a, b) for a, b in zip(self.caps, rcaps))
                ops = []
                for e in edges:
                    for t in parts(e[2], rest_out[e[0]]):
                        ops.append(dist.P2POp(dist.isend, t, e[1], self.group, tag=e[3])); self.bytes_sent += t.numel()
                for e in recv_order:
                    for t in parts(e[4], rest_in[e[0]]):
                        ops.append(dist.P2POp(dist.irecv, t, e[1], self.group, tag=e[5]))
                self._round(ops)
                self.rounds += 1
            for e in edges:
                self.last_out[e[0]] = out_cnt[e[0]]; self.last_in[e[0]] = hdr[e[0]]
            self.exchanges += 1
            got = be.unpack(bufs["from_left"] if hl else None, bufs["from_right"] if hr else None, rcaps, counts=hdr, want_info=self.count_present)
            self.last = (sent, got)
            return be.step(P)

    def retry_local(self, P):
        """After `step` raised capi.NeedsMc (the step changed nothing and is rank-local from here on): the caller has
        updated the Monte-Carlo points of the listed floes; run the step again without another exchange."""
        with self.be.stream_scope():
            return self.be.step(P)


def _spatial_run(S, P, rank, local_rank, world, steps, warmup, torch, dist):
    """`steps` timed steps of the spatially sharded job on the field S; -> (seconds: max over ranks, stats)."""
    from . import capi
    Lx, Ly = float(P.c2_boundary[0].max()), float(P.c2_boundary[1].max())
    axis = strip_axis(Lx, Ly)
    edges, halo = strip_edges(S, world, Ly if axis else Lx, axis=axis)
    be = GpuSpatialBackend(S, local_rank, rank, world, edges, halo, axis=axis)
    sim = SpatialInteractions(be, rank, world, S.n, int(S.vert_off[-1]))
    it = 0
    present = S.n if world == 1 else 0
    for _ in range(warmup):
        P.doInt = int(it % 10 == 0); sim.step(P); it += 1
        if sim.last: present = max(present, int(sim.last[1].n_present))
    sim.count_present = False        # (the counts of owned / present floes cost a host round trip per step: warm-up only)
    dist.barrier(); torch.cuda.synchronize()
    rounds0, second0 = sim.rounds, sim.second_rounds
    t0 = time.perf_counter()
    narrow_ms = 0.0; pairs = 0; pair_bytes = 0.0
    for _ in range(steps):
        P.doInt = int(it % 10 == 0)
        st = sim.step(P); it += 1
        narrow_ms += st.narrow_ms; pairs += st.n_pairs          # (stage times: those of the step before, szStepStats -- one step's lag in a steady run)
        # SURVEY.md 8(d) B_pair over the pairs the team kernel processes (its three lists), as bench.py prices it at N = 1
        pair_bytes += 16.0 * st.team_vertices + 112.0 * st.team_pairs + 56.0 * st.pair_rows
    torch.cuda.synchronize(); dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=be.device)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    # collisions of the last step (calc_collisionNum.m:6) over the floes each rank owns, summed over the ranks
    off, rows = be.sim.interactions()
    mine = be.owned()
    keep = np.repeat(mine, np.diff(off))
    cnt = torch.tensor([capi.collision_count(rows[keep]), float(pairs), float(mine.sum()), float(present),
                        float(sim.bytes_sent / max(sim.exchanges, 1)), float(sim.rounds - rounds0), float(sim.second_rounds - second0)],
                       dtype=torch.float64, device=be.device)
    tot = cnt.clone(); dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    mx = cnt.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    stats = {"collisions_last_step": float(tot[0]), "pairs_evaluated": float(tot[1]), "owned_floes_rank0": int(cnt[2]),
             "present_floes_max": int(mx[3]), "bytes_sent_per_rank_step_max": float(mx[4]),
             "narrow_ms": narrow_ms / max(steps, 1), "pair_bytes": pair_bytes / max(steps, 1), "halo_m": halo,
             "strip_axis": "y" if axis else "x", "strip_edges_m": [float(e) for e in edges],
             # the exchange: "queued" = one host round trip per step (szSpatialPackQueued ...); message rounds per step of the
             # busiest rank (1.0: every step in one round), and the rank-steps of the timed region that needed a second round
             "exchange": "queued" if (sim.queued and world > 1) else "classic",
             "message_rounds_per_step_max": float(mx[5]) / max(steps, 1), "second_rounds_all_ranks": int(tot[6])}
    be.close()
    return float(el.item()), stats


def bench_main(args, rank, local_rank, world):
    """bench.py for --gpus > 1 (launched by torch.distributed.run, one rank per GPU): the spatial partition.
    `value` is the WEAK-scaling figure (args.floes floes per GPU: the one-GPU field repeated `world` times along x, one
    strip per GPU); the same line carries a STRONG-scaling run of the one-GPU field cut into `world` strips."""
    import torch
    import torch.distributed as dist
    import bench as bench_mod
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # RCCL prints a version banner on stdout when its first communicator comes up: everything but the one JSON line of
    # rank 0 goes to stderr
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    k = args.steps
    # weak: the one-GPU field repeated `world` times along x (same seeds on every rank -> identical arrays; box 2e5 * world x
    # 2e5 m), floes numbered copy by copy.  SUBZERO_WEAK_RANDOM=1: one Voronoi field of world x as many seeds instead --
    # floe numbers then carry no position, every 64-floe tile of the arrays holds a few floes of every rank, and the
    # per-floe kernels of a rank cost what they cost on the whole field
    if os.environ.get("SUBZERO_WEAK_RANDOM") == "1":
        S, P = bench_mod.build_workload(args.floes * world, Lx=1e5 * world, Ly=1e5)
    else:
        from . import synth
        S0, _ = bench_mod.build_workload(args.floes)
        _, P = bench_mod.build_workload(64, Lx=1e5 * world, Ly=1e5)      # (the parameters of the long box: boundary, ocean grid)
        S = synth.tile_field(S0, world, 1e5)
        P.Modulus = synth.reference_modulus(S)
    el, st = _spatial_run(S, P, rank, local_rank, world, k, args.warmup, torch, dist)
    strong = None
    try:
        S1, P1 = bench_mod.build_workload(args.floes)
        el1, st1 = _spatial_run(S1, P1, rank, local_rank, world, k, args.warmup, torch, dist)
        strong = {"workload": "BASELINE.json configs[3]: the ONE-GPU field (pack%d, +-1e5 m box) cut into %d strips -- fixed total work" % (S1.n, world),
                  "value": S1.n * k / el1, "unit": "floe-timesteps/s", "scaling": "strong", "floes": int(S1.n), "ms_per_step": el1 / k * 1e3,
                  "timesteps_per_s": k / el1, "owned_floes_rank0": st1["owned_floes_rank0"], "present_floes_max": st1["present_floes_max"],
                  "halo_m": st1["halo_m"], "strip_edges_m": st1["strip_edges_m"],
                  "bytes_sent_per_rank_step_max": st1["bytes_sent_per_rank_step_max"],
                  "contact_pairs_per_s": st1["collisions_last_step"] * k / el1}
    except ValueError as e:        # strips narrower than the halo: this field cannot be cut that finely
        strong = {"value": None, "note": str(e)}
    if rank == 0:
        achieved = st["pair_bytes"] / (st["narrow_ms"] * 1e-3) / 1e9 if st["narrow_ms"] > 0 else 0.0
        out = {
            "metric": "floe-timesteps/sec", "value": S.n * k / el, "unit": "floe-timesteps/s",
            "n_gpus": world, "steps": k, "warmup": args.warmup, "ms_per_step": el / k * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64+int64", "data": "synthetic",
            "config": {"workload": "`value` is the WEAK-scaling figure: pack%d, %d Voronoi floes (%d per GPU: the one-GPU field repeated along x, numbered copy by copy), "
                                   "concentration 1.0, periodic +-%.3g x +-1e5 m box, dt=10 s, collisions + ocean drag (forcing refreshed every "
                                   "10th step); the STRONG-scaling figure of BASELINE.json configs[3] (the 1e4-floe field itself cut into %d "
                                   "strips) is the `strong_scaling` object of this line"
                                   % (S.n, S.n, args.floes, 1e5 * world, world),
                       "floes": int(S.n), "floes_per_gpu": int(args.floes),
                       "parallelism": "%d GPUs: floes owned by x-strips, one RCCL send/recv of halo floes with each neighbour per step "
                                      "(ring), border pairs evaluated on both sides" % world},
            "timesteps_per_s": k / el,
            "contact_pairs_per_s": st["collisions_last_step"] * k / el,
            "contact_pairs_basis": "collisions of the last timed step (owned floes, summed over ranks) x steps / time",
            "candidate_pairs_evaluated_per_s": st["pairs_evaluated"] / el,
            "spatial": {k2: st[k2] for k2 in ("owned_floes_rank0", "present_floes_max", "bytes_sent_per_rank_step_max", "halo_m", "exchange",
                                              "message_rounds_per_step_max", "second_rounds_all_ranks")},
            "strong_scaling": strong,
            "roofline": {"bound": "hbm", "kernel": "k_narrow_team (rank 0's launch: the pairs on its hot / cold / wide lists)",
                         "achieved": achieved, "peak": bench_mod.HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / bench_mod.HBM_PEAK_GBS, "traffic": None,
                         "limiter": "FP64 / int64 vector issue of irregular geometry (16 lanes per pair), not HBM",
                         "kernel_ms": st["narrow_ms"], "algorithmic_bytes_per_launch": st["pair_bytes"]},
        }
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    dist.destroy_process_group()
