import sys, os, subprocess
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    os.environ["SUBZERO_CONVEX_LANES"]=sys.argv[1]
    sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
    from subzero_amd import capi
    if os.environ.get('SZ_LIB'): capi.LIB_PATH=os.path.join(R,'scratch',os.environ['SZ_LIB'],'libsubzero_hip.so')   # variant builds (amdgpu_waves_per_eu)
    import bench, numpy as np, time
    S,P=bench.build_workload(int(sys.argv[2]) if len(sys.argv)>2 else 10000)
    ctx=capi.Context(0); sim=capi.FloeInteractions(ctx,S)
    for it in range(25):
        P.doInt=int(it%10==0); sim.step(P)
    acc=np.zeros(3); n=0
    t0=time.perf_counter()
    for it in range(25,125):
        P.doInt=int(it%10==0); st=sim.step(P)
        acc+= [st.narrow_ms, st.step_ms, st.n_pairs]; n+=1
    el=time.perf_counter()-t0
    print("L=%s floes %d: narrow %.3f ms  step(dev) %.3f ms  wall %.3f ms/step  pairs %d retried %d"%(sys.argv[1], S.n, acc[0]/n, acc[1]/n, el/n*1e3, acc[2]/n, st.retried), flush=True)
else:
    for lib in ["w2","w3"]:
        os.environ['SZ_LIB']=lib
        print("lib", lib, flush=True)
        for v in (["16","20","22","24","32","0"] if lib=="w2" else ["11","14","16","22","0"]):
            subprocess.run([sys.executable, __file__, v])
        subprocess.run([sys.executable, __file__, "0", "40000"])
        subprocess.run([sys.executable, __file__, "32", "40000"])
