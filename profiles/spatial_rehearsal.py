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
from test_spatial_gpu import Rank, deliver


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
    S, P = bench.build_workload(10000 * 2, Lx=2e5)
    out["weak_x2"] = run(S, P, 2, steps)
    print(json.dumps(out))
