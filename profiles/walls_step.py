"""Step time of 1e4 floes between walls (bench.py's `other_fields` line of that name), for tuning comparisons of the wall
kernel (SUBZERO_WALLS_LANE0=1: lane 0 alone)."""
import sys, time, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import bench
from subzero_amd import capi
from subzero_amd import synth
S, P0 = bench.build_workload(10000)
P = synth.StepParams(dt=10.0, Lx=1e5, Ly=1e5, periodic=False); P.Modulus = P0.Modulus
ctx = capi.Context(0)
sim = capi.FloeInteractions(ctx, S)
it = 0
for _ in range(12):
    P.doInt = int(it % 10 == 0); sim.step(P); it += 1
t = time.perf_counter(); gen = 0
for _ in range(40):
    P.doInt = int(it % 10 == 0); st = sim.step(P); it += 1; gen += st.retried
print(os.environ.get("SUBZERO_WALLS_LANE0"), "ms/step %.4f  pairs through the general sweep per step %.1f" % ((time.perf_counter() - t) / 40 * 1e3, gen / 40))
