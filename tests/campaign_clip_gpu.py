"""GPU campaign of the B1 clip batch against the reference's Clipper build (oracle/_ref) and the host build of the engine:
generator after generator, seed after seed, all four operations, until the time is up (not collected by pytest; run by hand
on an MI355X: python tests/campaign_clip_gpu.py SECONDS SEED0).  The sweep pass runs every pair by a whole wave with the
ring set-up, the trimming, the pending scanlines and the active-edge mirror shared by the lanes (vatti.hpp add_ring_wave /
prebuild_beams / WideLdsWaveMirror, trim.hpp trim_ring_wave): this is their fuzz."""
import sys, time, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import clipgen as g
from subzero_amd import capi

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 70000
ctx = capi.Context(0)
ref = g.ref_lib()
assert ref is not None, "oracle/_ref is not built"
gens = [(g.shape_pairs, 1500, {}), (g.cut_pairs, 1500, {}), (g.snapped_shape_pairs, 1500, {"mode": 0}), (g.snapped_shape_pairs, 1500, {"mode": 1}),
        (g.star_pairs, 200, {}), (g.lattice_pairs, 3000, {}), (g.lattice_pairs, 1500, {"grid": 10, "max_verts": 30}), (g.rect_pairs, 2000, {}),
        (g.convex_pairs, 5000, {}), (g.neighbour_pairs, 3000, {}), (g.monotone_pairs, 3000, {}), (g.hull_pairs, 3000, {})]
t0 = time.time(); bad = 0; clips = 0; rounds = 0
while time.time() - t0 < budget:
    for gen, n, kw in gens:
        seed = seed0 + rounds
        ps = gen(n, seed=seed, **kw)
        for m in range(4):
            r = g.run_gpu(ctx, ps, m)
            d = g.compare(g.run_ref(ps, m), r, len(ps))
            clips += len(ps)
            if d:
                bad += 1
                print("DIFF", gen.__name__, kw, "seed", seed, "method", m, d[:3], flush=True)
        if time.time() - t0 >= budget:
            break
    rounds += 1
    print("round %d bad %d clips %.3g %.0fs" % (rounds, bad, clips, time.time() - t0), flush=True)
print("TOTAL rounds %d BAD %d clips %.3g against the reference Clipper build" % (rounds, bad, clips))
