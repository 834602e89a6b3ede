"""One rank of the weak-scaling run alone on the card (no exchange: its halo stays empty), to see what the GLOBAL size of
the arrays costs a rank: world x 1e4 floes uploaded, 1e4 of them owned.  usage: python profiles/one_rank_weak.py [world=8] [rank=3]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch
import bench
from subzero_amd import multigpu
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if os.environ.get("SUBZERO_WEAK_RANDOM") == "1":      # floe numbers without position
    S, P = bench.build_workload(10000 * world, Lx=1e5 * world)
else:                                                  # the bench's weak field: the one-GPU field repeated along x, numbered copy by copy
    from subzero_amd import synth
    S0, _ = bench.build_workload(10000)
    _, P = bench.build_workload(64, Lx=1e5 * world)
    S = synth.tile_field(S0, world, 1e5)
    P.Modulus = synth.reference_modulus(S)
edges, halo = multigpu.strip_edges(S, world, 1e5 * world)
be = multigpu.GpuSpatialBackend(S, 0, rank, world, edges, halo)
caps = (S.n, 47 * S.n + 2 * int(S.vert_off[-1]), max(64, S.n // 16))
nb = be.message_bytes(caps)
bl, br = be.buffer("l", nb), be.buffer("r", nb)
tp = ts = 0.0
for it in range(40):
    P.doInt = int(it % 10 == 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with be.stream_scope():
        be.pack(P.PERIODIC, bl, br, caps)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    with be.stream_scope():
        st = be.step(P)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    if it >= 10: tp += t1 - t0; ts += t2 - t1
print("world %d rank %d: %d floes uploaded, %d owned; pack %.3f ms, step %.3f ms (device %.3f), pairs %d" % (world, rank, S.n, int(be.owned().sum()), tp / 30 * 1e3, ts / 30 * 1e3, st.step_ms, st.n_pairs))
print(be.sim.diagnostics())
be.close()
