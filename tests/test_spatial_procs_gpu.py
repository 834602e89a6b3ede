"""The spatially partitioned step END TO END in real processes on the GPU: two and three ranks, a process each, every one
with its own context on the one card of the test box; subzero_amd/multigpu.SpatialInteractions with its real GPU backend
(the C ABI: szSpatialPack ... / szSpatialPackQueued ...), the messages carried by a gloo group through host memory (RCCL
refuses two ranks on one device; `StagedDist` below stands in for torch.distributed's send / recv of device tensors and
changes nothing else).  What only real processes show: that both ends of every edge post matching sends and receives in
every case of the queued exchange (a step taken in one go, a step vetoed on one side and repeated after the second round,
a rank whose own message needs the second round while its step ran) -- nobody waits for a message that is not sent.
Every rank steps the whole field on a context of its own as well and compares its owned floes after every step."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class StagedDist:
    """torch.distributed's P2POp / batch_isend_irecv for device tensors over a gloo group: through host memory."""
    isend, irecv = "isend", "irecv"

    class P2POp:
        def __init__(self, op, tensor, peer, group=None, tag=0):
            self.op, self.tensor, self.peer, self.group, self.tag = op, tensor, peer, group, tag

    class Work:
        def __init__(self, works, back):
            self.works, self.back = works, back

        def wait(self):
            for w in self.works:
                w.wait()
            for t, c in self.back:
                t.copy_(c)

    def __init__(self, dist, torch):
        self.d, self.torch = dist, torch

    def batch_isend_irecv(self, ops):
        real, back = [], []
        for o in ops:
            if o.op == "isend":
                real.append(self.d.P2POp(self.d.isend, o.tensor.cpu(), o.peer, o.group, tag=o.tag))
            else:
                c = self.torch.empty(o.tensor.shape, dtype=o.tensor.dtype)
                real.append(self.d.P2POp(self.d.irecv, c, o.peer, o.group, tag=o.tag))
                back.append((o.tensor, c))
        return [StagedDist.Work(self.d.batch_isend_irecv(real), back)]


def _worker(rank, world, port, queued, drift, steps, q):
    try:
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        os.environ["SUBZERO_SPATIAL_QUEUED"] = "1" if queued else "0"
        import torch
        import torch.distributed as dist
        import hoststep
        import scenes
        from subzero_amd import capi, multigpu
        dist.init_process_group("gloo", rank=rank, world_size=world)
        n = 2700 if world == 3 else 1800
        S, P = scenes.pack(n, periodic=True, n_stress=6, n_mc=80)
        P.set_reference_ocean()
        S.Ui[:] += drift
        ref = S.copy()
        ctx0 = capi.Context(0)
        whole = capi.FloeInteractions(ctx0, ref)               # the one-GPU step on the whole field, on this process's own context
        edges, halo = multigpu.strip_edges(S, world, float(P.c2_boundary[0].max()))
        be = multigpu.GpuSpatialBackend(S, 0, rank, world, edges, halo)
        sim = multigpu.SpatialInteractions(be, rank, world, S.n, int(S.vert_off[-1]))
        sim.dist = StagedDist(dist, torch)
        for it in range(steps):
            P.doInt = int(it % 4 == 0)
            whole.step(P)
            if it >= 1:
                sim.count_present = False
            sim.step(P)
            whole.download(capi.SYNC_ALL)
            G = be.sim.download(capi.SYNC_ALL)
            mine = be.owned()
            masks = [torch.zeros(S.n, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(masks, torch.from_numpy(mine.astype(np.uint8)))
            assert (sum(m.numpy().astype(int) for m in masks) == 1).all(), "step %d: ownership is not a partition" % it
            ids = np.nonzero(mine)[0]
            for f in hoststep.STATE_FIELDS:
                if f in ("kill", "transfer", "cax", "cay"):
                    continue
                a, b = np.asarray(getattr(ref, f))[ids], np.asarray(getattr(G, f))[ids]
                assert ((a == b) | (np.isnan(a.astype(float)) & np.isnan(b.astype(float)))).all(), "step %d rank %d: %s of owned floes differs" % (it, rank, f)
            assert (np.abs(ref.Stress[ids] - G.Stress[ids]) <= hoststep.stress_bound(ref)[ids]).all(), "step %d: Stress" % it
            o0, r0 = whole.interactions(); o1, r1 = be.sim.interactions()
            for i in ids[::5]:
                a = r0[o0[i]:o0[i + 1]]; b = r1[o1[i]:o1[i + 1]]
                assert a.shape == b.shape and np.array_equal(a[:, 1:], b[:, 1:]), (it, rank, i)
        out = (rank, "ok", sim.rounds, sim.exchanges, sim.second_rounds, int(mine.sum()), int(be.sim.diagnostics()["steps"]))
        be.close(); ctx0.close()
        dist.destroy_process_group()
        q.put(out)
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "fail: %s\n%s" % (e, traceback.format_exc()), 0, 0, 0, 0, 0))


def run(world, queued, drift, steps):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_worker, args=(r, world, port, queued, drift, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    res.sort()
    for r in res:
        assert r[1] == "ok", "rank %d: %s" % (r[0], r[1])
    return res


@pytest.mark.parametrize("world,queued,drift", [(2, True, 6.0), (3, True, 6.0), (2, True, 45.0), (2, False, 6.0)])
def test_ranks_in_processes_of_their_own(world, queued, drift):
    steps = 12
    res = run(world, queued, drift, steps)
    for r in res:
        assert r[3] == steps
        if queued:
            # some steps in one round; with floes crossing the edges some with a second one
            assert steps + 1 <= r[2] <= 2 * steps, r
    if queued and drift > 20:
        assert sum(r[4] for r in res) > 0, "no step of the queued exchange needed its second round"
    print("ranks in processes x%d (%s exchange, drift %.0f m/s): rounds %s, second rounds of the queued exchange %s, device steps %s"
          % (world, "queued" if queued else "classic", drift, [r[2] for r in res], [r[4] for r in res], [r[6] for r in res]))
