"""Per-part time of the wall kernel (floes against the domain box, general sweep by a wave): make -C subzero_amd/csrc diag_step;
SUBZERO_HIP_LIB=subzero_amd/csrc/libsubzero_hip_stepphases.so python profiles/walls_phases.py"""
import sys, os, ctypes
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, bench
from subzero_amd import capi, synth
S, P0 = bench.build_workload(10000)
P = synth.StepParams(dt=10.0, Lx=1e5, Ly=1e5, periodic=False); P.Modulus = P0.Modulus
ctx = capi.Context(0)
sim = capi.FloeInteractions(ctx, S)
L = ctx.L
for it in range(5):
    P.doInt=0; st=sim.step(P)
out = (ctypes.c_ulonglong*56)()
L.szDebugPhases(out, 1)
names = {1:"clip1 sweep",2:"after clip1",3:"crossings",4:"regions",5:"normals",6:"clip2 sweep",7:"after clip2",8:"setup",9:"?",10:"fill",11:"trim",12:"add_ring",13:"sort minima",14:"insert_minima",15:"horizontals",16:"build_crossings",17:"top_of_beam",18:"after loop",19:"write",20:"do_maxima",21:"top: horizontals",22:"top: promote",23:"crossings: order+process"}
for it in range(3):
    P.doInt=0; st=sim.step(P)
    L.szDebugPhases(out, 1)
    acc = np.array(out[:24], float); worst = np.array(out[24:56], float)
    tot = acc.sum()
    print("step", it, "general", st.retried, "mean pair (ms, one lane) %.4f" % (tot/64/1e5/max(st.retried,1)))
    print("  share:", {names.get(k,k): round(acc[k]/tot,3) for k in range(1,24) if acc[k]})
    print("  worst pair total %.3f ms:" % (worst[0]/1e5), {names.get(k,k): round(worst[k]/1e5,3) for k in range(1,24) if worst[k]})
