"""The spatially partitioned step on the GPU through the C ABI (szSpatialInit / Pack / Unpack + szStep): two and three
ranks emulated on the one card of the test box -- one context and one device-resident array per rank, the neighbour
messages copied between their device buffers as RCCL send / recv would deliver them (exactly the used parts, after the
headers) -- against the single-GPU step and the oracle.  The orchestration over torch.distributed is exercised under
gloo in tests/test_spatial_cpu.py and, here, with a one-rank RCCL group."""
import numpy as np
import pytest

import hoststep
import oracle
import scenes
from subzero_amd import capi, multigpu

pytestmark = pytest.mark.gpu


class Rank:
    def __init__(self, S, rank, world, edges, halo):
        self.S = S.copy()
        self.be = multigpu.GpuSpatialBackend(self.S, 0, rank, world, edges, halo)
        self.caps = (S.n, 47 * S.n + 2 * int(S.vert_off[-1]), max(64, S.n // 16))
        nbytes = self.be.message_bytes(self.caps)
        self.buf = {k: self.be.buffer(k, nbytes) for k in ("send_left", "send_right", "from_left", "from_right")}
        self.rank, self.world = rank, world

    def views(self, key):
        caps, row = self.caps, self.be.bulk_row()
        o_rec = 64; o_dbl = o_rec + ((4 * caps[0] + 7) // 8) * 8; o_bulk = o_dbl + 8 * caps[1]
        b = self.buf[key]
        return b[:64], b[o_rec:o_dbl], b[o_dbl:o_bulk], b[o_bulk:o_bulk + caps[2] * row]


def deliver(src, skey, dst, dkey):
    """What the two rounds of send / recv move: the header, then the used part of each array."""
    hs, rs, ds, bs = src.views(skey); hd, rd, dd, bd = dst.views(dkey)
    hd.copy_(hs)
    n_rec, n_dbl, n_migr, status = src.be.header(hs)
    assert status == 0
    row = src.be.bulk_row()
    rd[:4 * n_rec].copy_(rs[:4 * n_rec]); dd[:8 * n_dbl].copy_(ds[:8 * n_dbl]); bd[:n_migr * row].copy_(bs[:n_migr * row])
    return 64 + 4 * n_rec + 8 * n_dbl + n_migr * row, n_migr


@pytest.mark.parametrize("world,periodic", [(2, True), (3, True), (2, False)])
def test_emulated_ranks_match_the_oracle(world, periodic):
    import torch
    n = 2700 if world == 3 else 1800
    S, P = scenes.pack(n, periodic=periodic, n_stress=6, n_mc=80)
    P.set_reference_ocean()
    S.Ui += 45.0 if periodic else 12.0
    A = S.copy()
    edges, halo = multigpu.strip_edges(S, world, float(P.c2_boundary[0].max()))
    ranks = [Rank(S, r, world, edges, halo) for r in range(world)]
    moved = 0; wire = 0; present_max = 0
    for it in range(8):
        P.doInt = int(it % 4 == 0)
        ra = oracle.step(A, P, threads=16)
        infos = []
        for R in ranks:
            with R.be.stream_scope():
                infos.append(R.be.pack(P.PERIODIC, R.buf["send_left"], R.buf["send_right"], R.caps))
        torch.cuda.synchronize()
        for R, sent in zip(ranks, infos):
            assert not sent.overflow
            if sent.has_left:
                b, m = deliver(R, "send_left", ranks[(R.rank - 1) % world], "from_right"); wire += b; moved += m
            if sent.has_right:
                b, m = deliver(R, "send_right", ranks[(R.rank + 1) % world], "from_left"); wire += b; moved += m
        torch.cuda.synchronize()
        own = np.zeros(S.n, int)
        for R, sent in zip(ranks, infos):
            with R.be.stream_scope():
                fl = R.buf["from_left"] if sent.has_left else None; fr = R.buf["from_right"] if sent.has_right else None
                if it % 2:      # the caller has read the headers (as SpatialInteractions does): nothing is read back by the unpack
                    counts = {k: R.be.header(R.views(key)[0])[:3] for k, key, has in (("left", "from_left", sent.has_left), ("right", "from_right", sent.has_right)) if has}
                    got = R.be.unpack(fl, fr, R.caps, counts=counts, want_info=(it % 4 == 1))
                else:
                    got = R.be.unpack(fl, fr, R.caps)
                if got is not None:
                    present_max = max(present_max, int(got.n_present))
                R.be.step(P)
            G = R.be.sim.download(capi.SYNC_ALL)
            off, rows = R.be.sim.interactions()
            mine = R.be.owned()
            own += mine
            ids = np.nonzero(mine)[0]
            for f in hoststep.STATE_FIELDS:
                if f in ("kill", "transfer", "cax", "cay"):
                    continue
                a, b = np.asarray(getattr(A, f))[ids], np.asarray(getattr(G, f))[ids]
                assert ((a == b) | (np.isnan(a.astype(float)) & np.isnan(b.astype(float)))).all(), "step %d rank %d: %s" % (it, R.rank, f)
            d = np.abs(A.Stress[ids] - G.Stress[ids])
            assert (d <= hoststep.stress_bound(A)[ids]).all()
            for i in ids[::7]:        # contact rows of every 7th owned floe (real partners by number; ghost partners are numbered per rank)
                a = ra.inter[ra.inter_off[i]:ra.inter_off[i + 1]]; b = rows[off[i]:off[i + 1]]
                assert a.shape == b.shape and np.array_equal(a[:, 1:], b[:, 1:]), (it, R.rank, i)
                real = a[:, 0] <= S.n
                assert np.array_equal(a[real, 0], b[real, 0])
        assert (own == 1).all(), "step %d: ownership is not a partition" % it
    assert moved > 0 and present_max < S.n
    print("spatial x%d: %.1f KB on the wire per rank and step, most floes present on a rank %d of %d"
          % (world, wire / 8 / world / 1024, present_max, S.n))
    for R in ranks:
        R.be.close()


def plan_of(c):
    return multigpu.SpatialInteractions._plan_queued(c)


def deliver_planned(src, skey, dst, dkey, plan):
    """First round of the queued exchange: the header and the agreed leading part of the two record arrays."""
    hs, rs, ds, bs = src.views(skey); hd, rd, dd, bd = dst.views(dkey)
    row = src.be.bulk_row()
    hd.copy_(hs); rd[:4 * plan[0]].copy_(rs[:4 * plan[0]]); dd[:8 * plan[1]].copy_(ds[:8 * plan[1]]); bd[:plan[2] * row].copy_(bs[:plan[2] * row])


@pytest.mark.parametrize("world,periodic", [(2, True), (3, True), (2, False)])
def test_queued_exchange_on_emulated_ranks(world, periodic):
    """szSpatialPackQueued / UnpackQueued / szStep / szSpatialHeaders / PackFinish through the C ABI, ranks emulated on one
    card: nothing is read back between pack and step; the device takes a message within its plan and vetoes the step
    otherwise (a floe changed strip, a message outgrew its plan); a vetoed step changes nothing and is repeated after the
    second round.  Owned floes against the oracle after every step, ownership a partition."""
    import torch
    n = 2700 if world == 3 else 1800
    S, P = scenes.pack(n, periodic=periodic, n_stress=6, n_mc=80)
    P.set_reference_ocean()
    S.Ui += 45.0 if periodic else 12.0
    A = S.copy()
    edges, halo = multigpu.strip_edges(S, world, float(P.c2_boundary[0].max()))
    ranks = [Rank(S, r, world, edges, halo) for r in range(world)]
    last = {}                                # (sender rank, side) -> (records, doubles, migrants) of the message before
    vetoed_steps = clean_steps = moved = 0
    sides = (("left", "send_left", -1, "from_right"), ("right", "send_right", +1, "from_left"))
    for it in range(10):
        P.doInt = int(it % 4 == 0)
        ra = oracle.step(A, P, threads=16)
        has = lambda R, side: periodic or (R.rank > 0 if side == "left" else R.rank < world - 1)
        if it == 0:                          # the first step: the classic exchange (nothing is agreed yet)
            infos = []
            for R in ranks:
                with R.be.stream_scope():
                    infos.append(R.be.pack(P.PERIODIC, R.buf["send_left"], R.buf["send_right"], R.caps))
            torch.cuda.synchronize()
            for R, sent in zip(ranks, infos):
                for side, skey, d, dkey in sides:
                    if has(R, side):
                        deliver(R, skey, ranks[(R.rank + d) % world], dkey)
                        last[(R.rank, side)] = (int(getattr(sent, side + "_records")), int(getattr(sent, side + "_doubles")), int(getattr(sent, side + "_migrants")))
            torch.cuda.synchronize()
            for R in ranks:
                with R.be.stream_scope():
                    R.be.unpack(R.buf["from_left"] if has(R, "left") else None, R.buf["from_right"] if has(R, "right") else None, R.caps)
                    R.be.step(P)
        else:
            for R in ranks:
                with R.be.stream_scope():
                    R.be.pack_queued(P.PERIODIC, R.buf["send_left"], R.buf["send_right"], R.caps,
                                     plan_of(last[(R.rank, "left")]) if has(R, "left") else None, plan_of(last[(R.rank, "right")]) if has(R, "right") else None)
            torch.cuda.synchronize()         # (stands for the stream order of RCCL's send behind the pack)
            for R in ranks:
                for side, skey, d, dkey in sides:
                    if has(R, side):
                        deliver_planned(R, skey, ranks[(R.rank + d) % world], dkey, plan_of(last[(R.rank, side)]))
            torch.cuda.synchronize()
            veto = {}
            for R in ranks:
                # what arrives from my left is my left neighbour's message to ITS right, and the other way round
                pl = plan_of(last[((R.rank - 1) % world, "right")]) if has(R, "left") else None
                pr = plan_of(last[((R.rank + 1) % world, "left")]) if has(R, "right") else None
                with R.be.stream_scope():
                    R.be.unpack_queued(R.buf["from_left"] if pl else None, R.buf["from_right"] if pr else None, R.caps, pl, pr)
                    before = R.be.sim.download(capi.SYNC_ALL).copy() if it == 3 else None
                    try:
                        R.be.step(P); veto[R.rank] = False
                    except capi.StepVetoed:
                        veto[R.rank] = True
                        if before is not None:       # a vetoed step has changed nothing
                            assert hoststep.compare_states(before, R.be.sim.download(capi.SYNC_ALL), exact=True) == []
                    out_h, in_h, v = R.be.headers_host()
                    assert v == veto[R.rank]
                    R.out_h, R.in_h = out_h, in_h
            # both ends of an edge have seen the same header
            for R in ranks:
                for side, skey, d, dkey in sides:
                    if has(R, side):
                        assert ranks[(R.rank + d) % world].in_h["right" if side == "left" else "left"] == R.out_h[side]
            # second round: the sender whose message is beyond its plan (or carries floes that changed strip) finishes its
            # pack and sends everything; the vetoed receiver unpacks with the counts it has read, and steps again
            for R in ranks:
                beyond = False
                for side, skey, d, dkey in sides:
                    if has(R, side):
                        c, pn = R.out_h[side], plan_of(last[(R.rank, side)])
                        assert c[3] == 0
                        if c[0] > pn[0] or c[1] > pn[1] or c[2] > pn[2]:
                            beyond = True
                if beyond:
                    with R.be.stream_scope():
                        R.be.pack_finish(R.buf["send_left"], R.buf["send_right"], R.caps)
                    torch.cuda.synchronize()
                    for side, skey, d, dkey in sides:
                        if has(R, side):
                            _, m = deliver(R, skey, ranks[(R.rank + d) % world], dkey); moved += m
            torch.cuda.synchronize()
            for R in ranks:
                expect = any(has(R, s) and (R.in_h[s][0] > plan_of(last[((R.rank + (-1 if s == "left" else 1)) % world, "right" if s == "left" else "left")])[0]
                                            or R.in_h[s][1] > plan_of(last[((R.rank + (-1 if s == "left" else 1)) % world, "right" if s == "left" else "left")])[1]
                                            or R.in_h[s][2] > plan_of(last[((R.rank + (-1 if s == "left" else 1)) % world, "right" if s == "left" else "left")])[2]) for s in ("left", "right"))
                assert veto[R.rank] == expect, (it, R.rank)
                if veto[R.rank]:
                    vetoed_steps += 1
                    counts = {s: R.in_h[s][:3] for s in ("left", "right") if has(R, s)}
                    with R.be.stream_scope():
                        R.be.unpack(R.buf["from_left"] if has(R, "left") else None, R.buf["from_right"] if has(R, "right") else None, R.caps, counts=counts, want_info=False)
                        R.be.step(P)
                else:
                    clean_steps += 1
            for R in ranks:
                for side in ("left", "right"):
                    if has(R, side):
                        last[(R.rank, side)] = R.out_h[side][:3]
        own = np.zeros(S.n, int)
        for R in ranks:
            G = R.be.sim.download(capi.SYNC_ALL)
            off, rows = R.be.sim.interactions()
            mine = R.be.owned()
            own += mine
            ids = np.nonzero(mine)[0]
            for f in hoststep.STATE_FIELDS:
                if f in ("kill", "transfer", "cax", "cay"):
                    continue
                a, b = np.asarray(getattr(A, f))[ids], np.asarray(getattr(G, f))[ids]
                assert ((a == b) | (np.isnan(a.astype(float)) & np.isnan(b.astype(float)))).all(), "step %d rank %d: %s" % (it, R.rank, f)
            assert (np.abs(A.Stress[ids] - G.Stress[ids]) <= hoststep.stress_bound(A)[ids]).all()
            assert np.array_equal(A.StressH[ids], G.StressH[ids]), "step %d rank %d: the stress history of a floe that changed strip" % (it, R.rank)
            for i in ids[::7]:
                a = ra.inter[ra.inter_off[i]:ra.inter_off[i + 1]]; b = rows[off[i]:off[i + 1]]
                assert a.shape == b.shape and np.array_equal(a[:, 1:], b[:, 1:]), (it, R.rank, i)
        assert (own == 1).all(), "step %d: ownership is not a partition" % it
    assert vetoed_steps > 0 and clean_steps > 0 and moved > 0, (vetoed_steps, clean_steps, moved)
    print("queued exchange x%d: %d rank-steps taken in one go, %d vetoed and repeated after a second round, %d floes changed strip" % (world, clean_steps, vetoed_steps, moved))
    for R in ranks:
        R.be.close()


def test_orchestration_with_a_one_rank_rccl_group():
    import os
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        S, P = scenes.pack(300, periodic=True, n_stress=4, n_mc=60)
        A = S.copy()
        edges, halo = multigpu.strip_edges(S, 1, float(P.c2_boundary[0].max()))
        be = multigpu.GpuSpatialBackend(S, 0, 0, 1, edges, halo)
        sim = multigpu.SpatialInteractions(be, 0, 1, S.n, int(S.vert_off[-1]))
        for it in range(4):
            P.doInt = int(it == 0)
            oracle.step(A, P, threads=4); sim.step(P)
        G = be.sim.download(capi.SYNC_ALL)
        assert hoststep.compare_states(A, G) == [] and be.owned().all()
        be.close()
    finally:
        dist.destroy_process_group()


def test_a_message_that_overflows_hands_nothing_over():
    """szSpatialPack with no room for a floe that changed strip: overflow is reported with what the edge needs, the
    ownership is as it was, and the same call with room succeeds (include/subzero_hip.h: "Pack can be called again")."""
    S, P = scenes.pack(1800, periodic=True, n_stress=6, n_mc=80)
    P.set_reference_ocean()
    S.Ui += 90.0
    edges, halo = multigpu.strip_edges(S, 2, float(P.c2_boundary[0].max()))
    R = Rank(S, 0, 2, edges, halo)
    sent = None
    for it in range(6):                     # step alone until some of its floes have left the strip
        P.doInt = 0
        R.be.step(P)
    before = R.be.owned()
    small = (R.caps[0], R.caps[1], 0)
    nbytes = R.be.message_bytes(R.caps)
    with R.be.stream_scope():
        sent = R.be.pack(P.PERIODIC, R.buf["send_left"][:nbytes], R.buf["send_right"][:nbytes], small)
    assert sent.overflow and sent.need_migrants >= 1
    assert np.array_equal(R.be.owned(), before)
    with R.be.stream_scope():
        sent = R.be.pack(P.PERIODIC, R.buf["send_left"], R.buf["send_right"], R.caps)
    assert not sent.overflow and sent.left_migrants + sent.right_migrants >= 1
    after = R.be.owned()
    assert before.sum() - after.sum() == sent.left_migrants + sent.right_migrants
    R.be.close()
