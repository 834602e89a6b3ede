"""Step time on 2500 evolved outlines of FloeShapes.mat (the workload of profiles/r0x_evolved_kernel_stats.csv)."""
import sys, time, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, scenes
from subzero_amd import capi
S,P = scenes.concave(2500, periodic=True)
ctx = capi.Context(0)
sim = capi.FloeInteractions(ctx, S)
for it in range(3):
    P.doInt=0; st=sim.step(P)
t=time.perf_counter(); nar=0; dev=0
for it in range(10):
    P.doInt=0; st=sim.step(P); nar+=st.narrow_ms; dev+=st.step_ms
el=time.perf_counter()-t
print(os.environ.get("SUBZERO_NARROW_LANES"), os.environ.get("SUBZERO_NARROW_LDS"), "ms/step %.3f device %.3f narrow first pass %.3f pairs %d retried %d" % (el/10*1e3, dev/10, nar/10, st.n_pairs, st.retried))
