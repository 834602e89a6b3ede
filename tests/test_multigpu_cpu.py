"""The N > 1 path under a real process group on the CPU (gloo, world_size 2).

The orchestration is subzero_amd/multigpu.ShardedInteractions, unchanged; the backend is the host build of the
device kernel bodies (tests/hostsim), whose shard export / import code is the same source the GPU runs.
Checked: after every step both ranks hold exactly the state the single-process step produces (which the
other CPU tests pin bit-for-bit to the oracle), and the exchanged segments really partition the pair list."""
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


class HostBackend:
    """multigpu backend on tests/hostsim (CPU tensors as exchange buffers)."""

    def __init__(self, floes):
        import torch
        import hoststep
        import oracle
        self.torch, self.orc = torch, oracle
        self.L = hoststep.lib()
        self.S = floes
        self.cf = oracle.pack_floes(floes)
        self.L.host_shard_segment_bytes.restype = ctypes.c_int64
        self.L.host_shard_segment_bytes.argtypes = [ctypes.c_int64, ctypes.c_int64]
        self.L.host_shard_begin.argtypes = [ctypes.POINTER(oracle.CFloes), ctypes.POINTER(oracle.CParams), ctypes.c_int,
                                            ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]
        self.L.host_shard_export.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32]
        self.L.host_shard_finish.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, oracle._ip,
                                             ctypes.POINTER(ctypes.c_int64)]
        self.L.host_shard_forcing_bytes.restype = ctypes.c_int64
        self.L.host_shard_forcing_bytes.argtypes = [ctypes.c_int]
        self.L.host_shard_forcing.argtypes = [ctypes.c_void_p]
        self.L.host_shard_end.argtypes = [ctypes.c_void_p]
        self.L.host_shard_outputs.argtypes = [oracle._ip, oracle._dp, ctypes.c_int64, oracle._ip, oracle._ip,
                                              ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), oracle._ip]
        self.exchanges = 0
        self.finish_codes = []
        self._bufs = {}

    def begin(self, P, rank, world):
        from subzero_amd.multigpu import ShardInfo
        self.cp = self.orc.pack_params(P)
        raw = (ctypes.c_int64 * 7)()
        self.L.host_shard_begin(ctypes.byref(self.cf), ctypes.byref(self.cp), rank, world, raw)
        info = ShardInfo()
        (info.n_pairs, info.seg_lo, info.seg_hi, info.seg_max, info.seg_rows, info.status,
         info.replicated_failure) = [int(v) for v in raw]
        return info

    def last_error(self):
        return "host backend"

    def segment_bytes(self, seg_max, rows_max):
        return int(self.L.host_shard_segment_bytes(seg_max, rows_max))

    def buffer(self, name, nbytes):
        b = self._bufs.get(name)
        if b is None or b.numel() < nbytes:
            b = self.torch.zeros(int(nbytes) + 64, dtype=self.torch.uint8)
            self._bufs[name] = b
        return b[:nbytes]

    def stream_scope(self):
        import contextlib
        return contextlib.nullcontext()

    def export(self, send, seg_max, rows_max, status):
        assert self.L.host_shard_export(ctypes.c_void_p(send.data_ptr()), seg_max, rows_max, int(status)) == 0

    def _outputs(self):
        S = self.S
        cap_rows = 64 * S.n + 1024
        self.inter_off = np.zeros(S.n + 1, np.int32); inter = np.zeros((cap_rows, 7))
        pi = np.zeros(cap_rows, np.int32); pj = np.zeros(cap_rows, np.int32)
        npairs = ctypes.c_int64(); next_ = ctypes.c_int32()
        rc = self.L.host_shard_outputs(self.orc._p(self.inter_off, self.orc._ip), self.orc._p(inter), cap_rows,
                                       self.orc._p(pi, self.orc._ip), self.orc._p(pj, self.orc._ip), cap_rows,
                                       ctypes.byref(npairs), ctypes.byref(next_))
        assert rc == 0, rc
        self.inter = inter[:self.inter_off[-1]].copy()
        self.pairs = np.stack([pi[:npairs.value], pj[:npairs.value]], 1)

    def finish(self, recv, seg_max, rows_max):
        fx = ctypes.c_int32(0)
        need = ctypes.c_int64(0)
        rc = self.L.host_shard_finish(ctypes.c_void_p(recv.data_ptr()), seg_max, rows_max, ctypes.byref(fx), ctypes.byref(need))
        self.finish_codes.append(rc)
        if rc == 0 and not fx.value:
            self._outputs()
        return rc, int(need.value), None, bool(fx.value)

    def forcing_bytes(self, world):
        return int(self.L.host_shard_forcing_bytes(world))

    def forcing_export(self, send):
        assert self.L.host_shard_forcing(ctypes.c_void_p(send.data_ptr())) == 0

    def end(self, recv):
        self.exchanges += 1
        assert self.L.host_shard_end(ctypes.c_void_p(recv.data_ptr())) == 0
        self._outputs()
        return None


def _worker(rank, world, port, scene, steps, q):
    try:
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        import torch.distributed as dist
        import hoststep
        import scenes
        from subzero_amd import multigpu
        dist.init_process_group("gloo", rank=rank, world_size=world)
        S, P = getattr(scenes, scene[0])(*scene[1], **scene[2])
        P.set_reference_ocean()                       # non-trivial forcing, so that its exchange matters
        ref = S.copy()
        be = HostBackend(S)
        sim = multigpu.ShardedInteractions(be, rank, world)
        segs = []
        for it in range(steps):
            P.doInt = int(it % 5 == 0)
            r1 = hoststep.step(ref, P)             # single-process device pipeline (pinned to the oracle elsewhere)
            if it == 3:                            # buffers far too small: the exchange must repeat, larger
                sim.min_rows, sim._rows_seen = 1, -1000
            sim.step(P)
            info = sim.last_info
            segs.append((int(info.n_pairs), int(info.seg_lo), int(info.seg_hi)))
            assert np.array_equal(r1.pairs, be.pairs), "pairs differ at step %d" % it
            assert np.array_equal(r1.inter_off, be.inter_off) and np.array_equal(r1.inter, be.inter), "rows differ at step %d" % it
            bad = hoststep.compare_states(ref, S)
            assert bad == [], "state differs at step %d: %s" % (it, bad)
        assert multigpu.ShardedInteractions.SHARD_RESIZE in be.finish_codes     # step 3 went round twice
        # an error on one rank travels in its segment header: every rank stops, none is left in a collective
        real_begin = be.begin
        def failing_begin(P_, rank_, world_):
            info = real_begin(P_, rank_, world_)
            if rank_ == 1:
                info.status = -5
            return info
        be.begin = failing_begin
        with pytest.raises(RuntimeError):
            sim.step(P)
        assert be.finish_codes[-1] == -7
        dist.destroy_process_group()
        assert be.exchanges == 2, be.exchanges        # forcing steps 0 and 5 went through the second all-gather
        assert np.abs(S.FxOA).max() > 0
        q.put((rank, "ok", segs, int(len(be.inter))))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "fail: %s\n%s" % (e, traceback.format_exc()), None, 0))


@pytest.mark.parametrize("scene", [("pack", (160,), {"periodic": True}),
                                   ("pack", (120,), {"periodic": False, "seed": 7, "vel_seed": 8, "sigma": 0.3})])
def test_sharded_step_equals_single_process_gloo_world2(scene):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world, steps = 2, 8
    procs = [ctx.Process(target=_worker, args=(r, world, port, scene, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    res.sort()
    for rank, status, segs, nrows in res:
        assert status == "ok", "rank %d: %s" % (rank, status)
    # the two segments partition the pair list at every step, and contacts were exercised
    for (n0, lo0, hi0), (n1, lo1, hi1) in zip(res[0][2], res[1][2]):
        assert n0 == n1 and lo0 == 0 and hi0 == lo1 and hi1 == n1
    assert res[0][3] > 0


def test_tile_field_numbers_the_copies_one_after_the_other():
    """synth.tile_field (the weak-scaling field of bench.py --gpus N): copy c of the one-GPU field is floes
    [c n0, (c + 1) n0), shifted by (2 c - (copies - 1)) Lx; everything else of a floe is its original's."""
    from subzero_amd import synth
    S0 = synth.voronoi_field(300, 1.0, seed=5, n_stress=4, n_mc=50)
    S = synth.tile_field(S0, 3, 1e5)
    n0, nv0 = S0.n, int(S0.vert_off[-1])
    assert S.n == 3 * n0 and int(S.vert_off[-1]) == 3 * nv0
    for c in range(3):
        sl = slice(c * n0, (c + 1) * n0)
        assert np.array_equal(S.Xi[sl], S0.Xi + (2 * c - 2) * 1e5) and np.array_equal(S.Yi[sl], S0.Yi)
        assert np.array_equal(S.rmax[sl], S0.rmax) and np.array_equal(S.Ui[sl], S0.Ui) and np.array_equal(S.mcA[sl], S0.mcA)
        assert np.array_equal(S.vert_off[c * n0:(c + 1) * n0 + 1] - c * nv0, S0.vert_off)
        assert np.array_equal(S.c0x[c * nv0:(c + 1) * nv0], S0.c0x)
    # strips over the long box: a rank's floes are a run of consecutive numbers but for the few at the copies' seams
    from subzero_amd import multigpu
    edges, halo = multigpu.strip_edges(S, 3, 3e5)
    own = np.digitize(S.Xi, edges[1:-1])
    for r in range(3):
        ids = np.nonzero(own == r)[0]
        assert len(ids) > 0.8 * n0 and (np.diff(ids) == 1).mean() > 0.9
