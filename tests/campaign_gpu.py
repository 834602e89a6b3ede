"""GPU campaign (not collected by pytest; python tests/campaign_gpu.py SECONDS SEED0 on an MI355X): random scenes (small, all kinds) and Voronoi / evolved fields of several
hundred floes; after the steps of each scene the post-step entries (corners, fracture deformation, eulerian, potential
interactions) are compared with the oracle as well."""
import sys, time, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
import scenes, oracle, hoststep
from subzero_amd import capi, synth, corners

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
ctx = capi.Context(0)
t0 = time.time(); bad = 0; done = 0; pairs = 0; general = 0; post = 0
seed = seed0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    if seed % 3 == 0:
        n = int(rng.integers(300, 1500)); periodic = bool(rng.integers(2))
        S, P = scenes.pack(n, periodic=periodic, seed=seed, vel_seed=seed + 1, sigma=float(rng.uniform(0.05, 0.5)), n_stress=4, n_mc=120)
        if rng.integers(2): P.set_reference_ocean()
        if periodic and rng.integers(2): S.Xi += float(rng.uniform(-600, 600)); S.Yi += float(rng.uniform(-600, 600))
        steps = 10
    else:
        S, P, steps = scenes.random_scene(seed)
        periodic = bool(P.PERIODIC)
    A, G = S.copy(), S.copy()
    sim = capi.FloeInteractions(ctx, G)
    ok = True
    try:
        for it in range(steps):
            P.doInt = int(it % 3 == 0)
            ra = oracle.step(A, P, threads=16)
            st = sim.step(P); pairs += st.n_pairs; general += st.retried
            off, rows = sim.interactions(); sim.download(capi.SYNC_ALL)
            if not (np.array_equal(off, ra.inter_off) and np.array_equal(rows, ra.inter)):
                print("MISMATCH rows", seed, it, flush=True); ok = False; break
            d = [b for b in hoststep.compare_states(A, G) if b[0] not in ("StressH", "Stress")]
            if d: print("MISMATCH state", seed, it, d[:2], flush=True); ok = False; break
        if ok and S.n - S.nb > 4:
            ids = np.sort(rng.choice(np.arange(S.nb + 1, S.n + 1), size=max(2, (S.n - S.nb) // 2), replace=False)).astype(np.int32)
            w = oracle.corner_candidates(A, ra.inter_off, ra.inter, ids, int(S.nb), P.c2_boundary)
            g = sim.corner_candidates(ids, int(S.nb), P.c2_boundary)
            if not (all(np.array_equal(a, b) for a, b in zip(w[:4], g[:4])) and w[4] == g[4]):
                print("MISMATCH corners", seed, flush=True); ok = False
            # frac_corner.m (plain case) with random corners marked
            cnt_w = w[1]; cid = [q for q in range(len(ids)) if cnt_w[q] >= 3][:40]
            if cid:
                foff = np.zeros(len(cid) + 1, np.int32); fg = []
                for k, q in enumerate(cid):
                    nst = int(A.vert_off[ids[q]] - A.vert_off[ids[q] - 1])
                    gq = np.zeros(nst, np.uint8); gq[:cnt_w[q]] = rng.uniform(size=cnt_w[q]) < 0.35; fg.append(gq); foff[k + 1] = foff[k] + nst
                fg = np.concatenate(fg)
                wo = oracle.frac_corner(G, ids[cid], foff, fg); go = sim.corner_cuts(ids[cid], foff, fg)
                if not (np.array_equal(wo[0], go[0]) and np.array_equal(wo[2], go[2]) and all(np.array_equal(a, b) for a, b in zip(wo[3], go[3]))):
                    print("MISMATCH cuts", seed, flush=True); ok = False
            w = oracle.fracture_deform(A, ra.inter_off, ra.inter, ids); g = sim.fracture_deform(ids)
            if not (all(np.array_equal(a, b) for a, b in zip(w[:4], g[:4])) and all(np.array_equal(a, b) for a, b in zip(w[4], g[4]))):
                print("MISMATCH deform", seed, flush=True); ok = False
            if True:
                A2 = A.copy(); A2.Stress[:] = G.Stress
                Nx, Ny = int(rng.integers(3, 20)), int(rng.integers(3, 20))
                nbe = int(S.nb) if rng.integers(2) else int(rng.integers(0, 4))
                w = oracle.eulerian_data(A2, Nx, Ny, nbe, P.c2_boundary, periodic); g, _ = sim.eulerian_data(Nx, Ny, nbe, P.c2_boundary, periodic)
                if not all(np.array_equal(w[k], g[k], equal_nan=True) for k in oracle.EULERIAN_FIELDS):
                    print("MISMATCH eulerian", seed, [k for k in oracle.EULERIAN_FIELDS if not np.array_equal(w[k], g[k], equal_nan=True)], flush=True); ok = False
            post += 1
    except Exception as e:
        # (the oracle's -3 = a floe with an empty Monte-Carlo mask needs rand: the device must report the same floes)
        if "or_step failed: -3" in repr(e):
            try:
                sim.step(P); print("MISMATCH needs-mc", seed, flush=True); ok = False
            except capi.NeedsMc:
                pass
        else:
            print("EXCEPTION", seed, repr(e)[:300], flush=True); ok = False
    bad += (not ok); done += 1; seed += 1
    del sim
    if done % 50 == 0: print("done", done, "bad", bad, "pairs %.2e general %.2e post-step checks %d" % (pairs, general, post), "%.0fs" % (time.time() - t0), flush=True)
print("TOTAL scenes", done, "BAD", bad, "pair evaluations %.2e, through the general kernel %.2e, post-step checks %d" % (pairs, general, post))
