"""Ad-hoc timing probe of szStep (not a test)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from subzero_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
t = time.time()
S = synth.voronoi_field(n, 1.0, seed=4, vel_seed=2, n_stress=1000)
P = synth.StepParams(dt=10, periodic=True); P.Modulus = synth.reference_modulus(S)
print("gen %.1fs n=%d" % (time.time() - t, S.n), flush=True)
ctx = capi.Context(0)
t = time.time(); sim = capi.FloeInteractions(ctx, S); print("upload %.2fs" % (time.time() - t), flush=True)
for it in range(steps):
    P.doInt = int(it % 10 == 0)
    t = time.time(); st = sim.step(P); wall = (time.time() - t) * 1e3
    print("step %3d wall %.2f ms | dev %.2f ms: broad %.2f narrow %.2f rows %.2f integ %.2f | N %d pairs %d rows %d retried %d"
          % (it, wall, st.step_ms, st.broad_ms, st.narrow_ms, st.rows_ms, st.integrate_ms, st.n_ext, st.n_pairs, st.n_rows, st.retried), flush=True)
ctx.close()
