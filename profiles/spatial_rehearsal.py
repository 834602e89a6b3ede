"""What one rank of the spatial partition costs per step, measured on ONE GPU: `world` ranks emulated on the card (a context
and a resident array each; the neighbour messages copied between their device buffers as RCCL send / recv would deliver
them -- tests/test_spatial_gpu.py's harness), each rank's pack, unpack and step timed on its own.  No xGMI here: the
figure is the per-rank compute of the STRONG-scaling run of BASELINE configs[3] (1e4 floes cut into `world` strips) and
of the weak-scaling run (1e4 floes per rank), not a scaling measurement.
(Eight contexts in one process are more streams than the process has hardware queues -- context.hpp -- and the wall clock of
the fifth one shows it: 2-3 ms per step with 0.12 ms of kernels.  `device_ms_per_rank` is the sum of a step's HIP-event
stage times and does not suffer from it; a rank of a real job is a process of its own.)
usage: python profiles/spatial_rehearsal.py [world=8] [steps=30]"""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch
import bench
from subzero_amd import multigpu
from test_spatial_gpu import Rank, deliver, deliver_planned, plan_of
from subzero_amd import capi


def run_queued(S, P, world, steps, warm=8):
    """The same with the queued exchange (szSpatialPackQueued / UnpackQueued: one host round trip per step, the step's own):
    per rank, the host time from the queued unpack to the step's return, and -- apart, because the emulation has to wait
    for the pack before it can copy the messages, which a rank of a real job does not -- the pack."""
    Lx = float(P.c2_boundary[0].max())
    edges, halo = multigpu.strip_edges(S, world, Lx)
    ranks = [Rank(S, r, world, edges, halo) for r in range(world)]
    periodic = bool(P.PERIODIC)
    has = lambda R, side: world > 1 and (periodic or (R.rank > 0 if side == "left" else R.rank < world - 1))
    sides = (("left", "send_left", -1, "from_right"), ("right", "send_right", +1, "from_left"))
    last = {}
    t_pack = np.zeros(world); t_rest = np.zeros(world); vetoed = 0; dev = np.zeros(world)
    for it in range(warm + steps):
        P.doInt = int(it % 10 == 0)
        if it == 0:
            infos = []
            for Rk in ranks:
                with Rk.be.stream_scope(): infos.append(Rk.be.pack(P.PERIODIC, Rk.buf["send_left"], Rk.buf["send_right"], Rk.caps))
            torch.cuda.synchronize()
            for Rk, sent in zip(ranks, infos):
                for side, skey, d, dkey in sides:
                    if has(Rk, side):
                        deliver(Rk, skey, ranks[(Rk.rank + d) % world], dkey)
                        last[(Rk.rank, side)] = tuple(int(getattr(sent, side + "_" + f)) for f in ("records", "doubles", "migrants"))
            torch.cuda.synchronize()
            for Rk in ranks:
                with Rk.be.stream_scope():
                    Rk.be.unpack(Rk.buf["from_left"] if has(Rk, "left") else None, Rk.buf["from_right"] if has(Rk, "right") else None, Rk.caps)
                    Rk.be.step(P)
            continue
        for k, Rk in enumerate(ranks):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with Rk.be.stream_scope():
                Rk.be.pack_queued(P.PERIODIC, Rk.buf["send_left"], Rk.buf["send_right"], Rk.caps,
                                  plan_of(last[(Rk.rank, "left")]) if has(Rk, "left") else None, plan_of(last[(Rk.rank, "right")]) if has(Rk, "right") else None)
            torch.cuda.synchronize()
            if it >= warm: t_pack[k] += time.perf_counter() - t0
        for Rk in ranks:
            for side, skey, d, dkey in sides:
                if has(Rk, side): deliver_planned(Rk, skey, ranks[(Rk.rank + d) % world], dkey, plan_of(last[(Rk.rank, side)]))
        torch.cuda.synchronize()
        veto = {}
        for k, Rk in enumerate(ranks):
            pl = plan_of(last[((Rk.rank - 1) % world, "right")]) if has(Rk, "left") else None
            pr = plan_of(last[((Rk.rank + 1) % world, "left")]) if has(Rk, "right") else None
            t0 = time.perf_counter()
            with Rk.be.stream_scope():
                Rk.be.unpack_queued(Rk.buf["from_left"] if pl else None, Rk.buf["from_right"] if pr else None, Rk.caps, pl, pr)
                try:
                    st = Rk.be.step(P); veto[k] = False
                    if it >= warm: dev[k] += st.step_ms
                except capi.StepVetoed:
                    veto[k] = True
                Rk.out_h, Rk.in_h, _ = Rk.be.headers_host()
            if it >= warm: t_rest[k] += time.perf_counter() - t0
        for Rk in ranks:           # second round where a message is beyond its plan
            if any(has(Rk, side) and (Rk.out_h[side][0] > plan_of(last[(Rk.rank, side)])[0] or Rk.out_h[side][1] > plan_of(last[(Rk.rank, side)])[1]
                                      or Rk.out_h[side][2] > plan_of(last[(Rk.rank, side)])[2]) for side in ("left", "right")):
                with Rk.be.stream_scope(): Rk.be.pack_finish(Rk.buf["send_left"], Rk.buf["send_right"], Rk.caps)
                torch.cuda.synchronize()
                for side, skey, d, dkey in sides:
                    if has(Rk, side): deliver(Rk, skey, ranks[(Rk.rank + d) % world], dkey)
        torch.cuda.synchronize()
        for k, Rk in enumerate(ranks):
            if veto[k]:
                vetoed += int(it >= warm)
                counts = {s_: Rk.in_h[s_][:3] for s_ in ("left", "right") if has(Rk, s_)}
                t0 = time.perf_counter()
                with Rk.be.stream_scope():
                    Rk.be.unpack(Rk.buf["from_left"] if has(Rk, "left") else None, Rk.buf["from_right"] if has(Rk, "right") else None, Rk.caps, counts=counts, want_info=False)
                    Rk.be.step(P)
                if it >= warm: t_rest[k] += time.perf_counter() - t0
            for side in ("left", "right"):
                if has(Rk, side): last[(Rk.rank, side)] = Rk.out_h[side][:3]
    for Rk in ranks: Rk.be.close()
    return {"world": world, "floes": int(S.n), "exchange": "queued", "unpack_plus_step_ms_per_rank": [round(float(v), 4) for v in t_rest / steps * 1e3],
            "slowest_rank_unpack_plus_step_ms": float(t_rest.max() / steps * 1e3),
            "pack_queued_ms (with a wait for the device that only the emulation needs)": float(t_pack.max() / steps * 1e3),
            "rank_steps_vetoed_and_repeated": int(vetoed), "rank_steps": int(world * steps)}


def run(S, P, world, steps, warm=8):
    Lx = float(P.c2_boundary[0].max())
    edges, halo = multigpu.strip_edges(S, world, Lx)
    ranks = [Rank(S, r, world, edges, halo) for r in range(world)]
    stage = np.zeros((world, 6)); t_pack = np.zeros(world); t_rest = np.zeros(world); present = np.zeros(world, int); owned = np.zeros(world, int); wire = 0
    for it in range(warm + steps):
        P.doInt = int(it % 10 == 0)
        infos = []
        for k, Rk in enumerate(ranks):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with Rk.be.stream_scope():
                infos.append(Rk.be.pack(P.PERIODIC, Rk.buf["send_left"], Rk.buf["send_right"], Rk.caps))
            torch.cuda.synchronize()
            if it >= warm: t_pack[k] += time.perf_counter() - t0
        for Rk, sent in zip(ranks, infos):
            assert not sent.overflow
            if sent.has_left: wire += deliver(Rk, "send_left", ranks[(Rk.rank - 1) % world], "from_right")[0]
            if sent.has_right: wire += deliver(Rk, "send_right", ranks[(Rk.rank + 1) % world], "from_left")[0]
        torch.cuda.synchronize()
        for k, (Rk, sent) in enumerate(zip(ranks, infos)):
            fl = Rk.buf["from_left"] if sent.has_left else None; fr = Rk.buf["from_right"] if sent.has_right else None
            counts = {n_: Rk.be.header(Rk.views(key)[0])[:3] for n_, key, has in (("left", "from_left", sent.has_left), ("right", "from_right", sent.has_right)) if has}
            t0 = time.perf_counter()
            with Rk.be.stream_scope():
                got = Rk.be.unpack(fl, fr, Rk.caps, counts=counts, want_info=(it == warm - 1))
                st = Rk.be.step(P)
            if it >= warm:
                for q, f in enumerate(("broad_ms", "pairs_ms", "narrow_ms", "rows_ms", "integrate_ms")): stage[k][q] += getattr(st, f)
                stage[k][5] += st.retried
            torch.cuda.synchronize()
            if it >= warm: t_rest[k] += time.perf_counter() - t0
            if got is not None: present[k] = int(got.n_present); owned[k] = int(got.n_owned)
    diag = [Rk.be.sim.diagnostics() for Rk in ranks]
    for Rk in ranks: Rk.be.close()
    per = (t_pack + t_rest) / steps * 1e3
    return {"world": world, "floes": int(S.n), "ms_per_step_per_rank": [round(float(v), 4) for v in per], "slowest_rank_ms": float(per.max()),
            "pack_ms": float(t_pack.max() / steps * 1e3), "unpack_plus_step_ms": float(t_rest.max() / steps * 1e3),
            "owned_floes": owned.tolist(), "present_floes": present.tolist(), "present_over_owned_max": float((present / np.maximum(owned, 1)).max()),
            "wire_KB_per_rank_step": wire / (warm + steps) / world / 1024, "halo_m": halo,
            "stage_ms_slowest_rank [broad, pairs, team kernel, rows, integrate, pairs through the general sweep]": [round(float(v), 4) for v in stage[int(np.argmax(per))] / steps],
            "device_ms_per_rank": [round(float(v), 4) for v in stage[:, :5].sum(1) / steps],
            "steps_fused": [d["steps_fused"] for d in diag], "spec_misses": [d["spec_misses"] for d in diag], "back_misses": [d["back_misses"] for d in diag]}


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    out = {}
    S, P = bench.build_workload(10000)
    out["one_gpu"] = run(S, P, 1, steps)
    for w in sorted({2, 4, world}):
        S, P = bench.build_workload(10000)
        out["strong_x%d" % w] = run(S, P, w, steps)
        print(json.dumps(out["strong_x%d" % w]), flush=True)
    for w in sorted({2, world}):
        S, P = bench.build_workload(10000)
        out["strong_x%d_queued" % w] = run_queued(S, P, w, steps)
        print(json.dumps(out["strong_x%d_queued" % w]), flush=True)
    S, P = bench.build_workload(10000 * 2, Lx=2e5)
    out["weak_x2"] = run(S, P, 2, steps)
    S, P = bench.build_workload(10000 * 2, Lx=2e5)
    out["weak_x2_queued"] = run_queued(S, P, 2, steps)
    print(json.dumps(out["weak_x2_queued"]), flush=True)
    print(json.dumps(out))
