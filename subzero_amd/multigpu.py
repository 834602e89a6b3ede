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


def bench_main(args, rank, local_rank, world):
    """bench.py for --gpus > 1 (launched by torch.distributed.run, one rank per GPU)."""
    import torch
    import torch.distributed as dist
    from . import capi
    import bench as bench_mod
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    # weak scaling: `floes` per GPU; the box grows with sqrt(world) so that the floe density (and the work per
    # candidate pair) stays that of the one-GPU workload.  Same seeds on every rank: identical arrays.
    S, P = bench_mod.build_workload(args.floes * world, box_scale=float(np.sqrt(world)))
    be = GpuBackend(S, local_rank)
    sim = ShardedInteractions(be, rank, world)
    it = 0
    for _ in range(args.warmup):
        P.doInt = int(it % 10 == 0); sim.step(P); it += 1
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    narrow_ms = 0.0; pairs = 0; rows = 0; seg_pairs = 0; pair_bytes = 0.0
    for _ in range(args.steps):
        P.doInt = int(it % 10 == 0)
        st = sim.step(P); it += 1
        narrow_ms += st.narrow_ms; pairs += st.n_pairs; rows += st.n_rows
        ns = sim.last_info.seg_hi - sim.last_info.seg_lo
        seg_pairs += ns
        share = ns / max(st.n_pairs, 1)
        pair_bytes += share * (16.0 * st.pair_vertices + 112.0 * st.n_pairs) + 56.0 * st.pair_rows
    torch.cuda.synchronize(); dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=be.device)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    el = float(el.item())
    off, inter = be.sim.interactions()
    k = args.steps
    if rank == 0:
        achieved = (pair_bytes / k) / (narrow_ms / k * 1e-3) / 1e9 if narrow_ms > 0 else 0.0
        out = {
            "metric": "floe-timesteps/sec", "value": S.n * k / el, "unit": "floe-timesteps/s",
            "n_gpus": world, "steps": k, "warmup": args.warmup, "ms_per_step": el / k * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64+int64", "data": "synthetic",
            "config": {"workload": "pack%d: %d Voronoi floes (%d per GPU), concentration 1.0, periodic +-%.3g m box, "
                                   "dt=10 s, collisions + ocean drag (forcing refreshed every 10th step)"
                                   % (S.n, S.n, args.floes, 1e5 * np.sqrt(world)),
                       "floes": int(S.n), "floes_per_gpu": int(args.floes),
                       "parallelism": "%d GPUs: narrow-phase sharded over the pair list, 1 RCCL all-gather/step, "
                                      "other stages replicated" % world},
            "timesteps_per_s": k / el,
            "contact_pairs_per_s": capi.collision_count(inter) * k / el,
            "candidate_pairs_per_s": pairs / el,
            "roofline": {"bound": "hbm", "kernel": "k_narrow_convex (rank 0's segment of the pair list)",
                         "achieved": achieved, "peak": bench_mod.HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / bench_mod.HBM_PEAK_GBS, "traffic": None,
                         "kernel_ms": narrow_ms / k, "algorithmic_bytes_per_launch": pair_bytes / k},
        }
        print(json.dumps(out), flush=True)
    be.close()
    dist.destroy_process_group()
