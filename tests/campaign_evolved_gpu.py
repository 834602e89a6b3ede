"""GPU campaign on fields of evolved (concave) outlines cut from the reference's FloeShapes.mat: every step of every field
against the oracle (not collected by pytest; run by hand on an MI355X: python tests/campaign_evolved_gpu.py N SEED0)."""
import sys, time, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
import scenes, oracle, hoststep, clipgen
from subzero_amd import capi, synth

def evolved(n, periodic, seed, max_verts, overlap, vel):
    rng = np.random.default_rng(seed)
    shapes, _ = clipgen.load_shapes()
    shapes = [s for s in shapes if len(s) <= max_verts]
    side = int(np.ceil(np.sqrt(n))); pitch = 2e5 / side; polys = []
    for k in range(n):
        s = shapes[rng.integers(len(shapes))]
        c = s.mean(0); r = np.sqrt(((s - c) ** 2).sum(1).max())
        sc = overlap * pitch / r; th = rng.uniform(0, 2 * np.pi)
        rot = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        gx, gy = k % side, k // side
        ctr = np.array([-1e5 + (gx + 0.5) * pitch, -1e5 + (gy + 0.5) * pitch]) + rng.uniform(-0.05, 0.05, 2) * pitch
        polys.append((s - c) @ rot.T * sc + ctr)
    S = synth.build_state(polys, height=0.5, seed=seed, n_stress=4)
    vr = np.random.default_rng(seed + 1)
    S.Ui[:] = vr.normal(0, vel, S.n); S.Vi[:] = vr.normal(0, vel, S.n); S.ksi_ice[:] = vr.normal(0, 2e-6, S.n)
    P = synth.StepParams(dt=10, periodic=periodic, height=0.5); P.Modulus = synth.reference_modulus(S)
    return S, P

ctx = capi.Context(0)
bad = 0; t0 = time.time(); pairs = 0; retried = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
S0 = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
for seed in range(S0, S0 + N):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(80, 700)); periodic = bool(rng.integers(2))
    S, P = evolved(n, periodic, seed, int(rng.choice([60, 90, 200, 600])), float(rng.uniform(0.5, 0.62)), float(rng.uniform(0.1, 1.5)))
    A, G = S.copy(), S.copy()
    sim = capi.FloeInteractions(ctx, G)
    ok = True
    for it in range(6):
        P.doInt = int(it % 3 == 0)
        ra = oracle.step(A, P, threads=16)
        st = sim.step(P); pairs += st.n_pairs; retried += st.retried
        off, rows = sim.interactions(); sim.download(capi.SYNC_ALL)
        if not (np.array_equal(off, ra.inter_off) and np.array_equal(rows, ra.inter)):
            print("MISMATCH rows", seed, it, flush=True); ok = False; break
        d = [b for b in hoststep.compare_states(A, G) if b[0] not in ("StressH", "Stress")]
        if d: print("MISMATCH state", seed, it, d[:2], flush=True); ok = False; break
    bad += (not ok)
    if (seed - S0) % 50 == 49: print("done", seed - S0 + 1, "bad", bad, "pairs %.2e general %.2e" % (pairs, retried), "%.0fs" % (time.time() - t0), flush=True)
print("TOTAL", N, "BAD", bad, "pair evaluations %.2e, through the general kernel %.2e" % (pairs, retried))
