import sys, os, ctypes
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SUBZERO_CONVEX_LANES"]=sys.argv[1]
sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
from subzero_amd import capi
capi.LIB_PATH=os.environ.get('SZ_DIAG_LIB', capi.LIB_PATH)   # a library built with make EXTRA=-DSZ_WAVE_CYCLES
import bench, numpy as np
S,P=bench.build_workload(10000)
ctx=capi.Context(0); sim=capi.FloeInteractions(ctx,S)
for it in range(31):
    P.doInt=int(it%10==0); st=sim.step(P)
L=int(sys.argv[1]) or 43
nw=(st.n_pairs+L-1)//L
out=(ctypes.c_ulonglong*nw)()
ctx.L.szDebugWaveCycles(out, nw)
v=np.array(list(out),float)/100.0   # microseconds
pi,pj=sim.pairs() if hasattr(sim,'pairs') else (None,None)
print("L=%s waves %d narrow_ms %.3f  wave us: min %.0f p10 %.0f p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f"%(sys.argv[1], nw, st.narrow_ms, v.min(), *np.percentile(v,[10,50,90,99,99.9]), v.max()))
if L==1:
    off, inter = sim.interactions()
    # which pairs are the slowest
    order=np.argsort(-v)[:10]
    print("slowest pairs (us):", v[order].round(0))
    h,e=np.histogram(v,bins=12); print("hist", list(zip(e.round(0)[:-1],h)))
