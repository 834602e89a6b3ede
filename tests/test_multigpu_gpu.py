"""Sharded step on the GPU (C ABI szStepShardBegin / Export / Finish).  A GPU box has one card, so two ranks
are emulated by two contexts on the same GPU whose segments are concatenated by hand (exactly what the
all-gather delivers); a world-of-one RCCL group exercises the real torch.distributed orchestration."""
import ctypes
import os
import socket

import numpy as np
import pytest

import hoststep
import scenes
from subzero_amd import capi, multigpu

pytestmark = pytest.mark.gpu


def test_two_emulated_ranks_equal_single_gpu_step():
    import torch
    S, P = scenes.pack(300, periodic=True)
    P.set_reference_ocean()
    ref = S.copy()
    c0 = capi.Context(0)
    ref_sim = capi.FloeInteractions(c0, ref)
    world = 2
    copies = [S.copy() for _ in range(world)]
    bes = [multigpu.GpuBackend(copies[r], 0) for r in range(world)]
    resized_any = False
    for it in range(10):
        P.doInt = int(it % 5 == 0)
        ref_sim.step(P)
        infos = [bes[r].begin(P, r, world) for r in range(world)]
        assert all(i.status == 0 for i in infos)
        assert infos[0].seg_lo == 0 and infos[0].seg_hi == infos[1].seg_lo and infos[1].seg_hi == infos[1].n_pairs
        seg_max = int(infos[0].seg_max)
        rows_max, resized = 8, False            # far too small on purpose: Finish must ask for a larger exchange
        while True:
            nbytes = bes[0].segment_bytes(seg_max, rows_max)
            sends = []
            for r in range(world):
                s = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
                bes[r].export(s, seg_max, rows_max, infos[r].status)
                sends.append(s)
            torch.cuda.synchronize()            # the backends run on streams of their own
            recv = torch.cat(sends)             # what all_gather_into_tensor delivers, rank order
            torch.cuda.synchronize()
            res = [bes[r].finish(recv, seg_max, rows_max) for r in range(world)]
            codes = [x[0] for x in res]
            if codes == [multigpu.ShardedInteractions.SHARD_RESIZE] * world:
                assert res[0][1] == res[1][1] > rows_max
                rows_max, resized = res[0][1], True
                continue
            assert codes == [0] * world, codes
            break
        resized_any = resized_any or resized
        pend = [x[3] for x in res]
        assert pend == [bool(P.doInt)] * world
        if pend[0]:                          # forcing step: the second exchange, again concatenated by hand
            fb = bes[0].forcing_bytes(world)
            fs = []
            for r in range(world):
                t = torch.empty(fb, dtype=torch.uint8, device="cuda:0")
                bes[r].forcing_export(t)
                fs.append(t)
            torch.cuda.synchronize()
            frecv = torch.cat(fs)
            torch.cuda.synchronize()
            for r in range(world):
                bes[r].end(frecv)
        ref_sim.download(capi.SYNC_ALL)
        off0, rows0 = ref_sim.interactions()
        for r in range(world):
            bes[r].sim.download(capi.SYNC_ALL)
            off, rows = bes[r].sim.interactions()
            assert np.array_equal(off, off0) and np.array_equal(rows, rows0), "rank %d step %d" % (r, it)
            assert hoststep.compare_states(ref, copies[r]) == [], "rank %d step %d" % (r, it)
    assert len(rows0) > 0 and resized_any
    for b in bes:
        b.close()
    c0.close()


def test_world_of_one_rccl_group():
    import torch
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        S, P = scenes.pack(200, periodic=False, seed=7, vel_seed=8, sigma=0.3)
        ref = S.copy()
        c0 = capi.Context(0)
        ref_sim = capi.FloeInteractions(c0, ref)
        be = multigpu.GpuBackend(S, 0)
        sim = multigpu.ShardedInteractions(be, 0, 1)
        for it in range(8):
            P.doInt = int(it % 5 == 0)
            ref_sim.step(P); sim.step(P)
        ref_sim.download(capi.SYNC_ALL); be.sim.download(capi.SYNC_ALL)
        assert hoststep.compare_states(ref, S) == []
        o0, r0 = ref_sim.interactions(); o1, r1 = be.sim.interactions()
        assert np.array_equal(o0, o1) and np.array_equal(r0, r1) and len(r0) > 0
        be.close(); c0.close()
    finally:
        dist.destroy_process_group()
