"""The same random scenes on the host build of the device pipeline (tests/hostsim) against the oracle, in parallel on the
CPU cores (not collected by pytest: python tests/campaign_host.py SEED0 N)."""
import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
from multiprocessing import Pool
def run(seed):
    import scenes, oracle, hoststep
    try:
        S,P,steps = scenes.random_scene(seed)
        A,G=S.copy(),S.copy()
        for it in range(steps):
            P.doInt=int(it%3==0)
            ra=oracle.step(A,P,threads=1); rg=hoststep.step(G,P)
            if not (np.array_equal(ra.pairs, rg.pairs) and np.array_equal(ra.inter_off, rg.inter_off) and np.array_equal(ra.inter, rg.inter)):
                return (seed, "rows step %d"%it)
            d=[b for b in hoststep.compare_states(A,G) if b[0] not in ("StressH","Stress")]
            if d: return (seed, "state step %d %s"%(it, d[:1]))
        return None
    except Exception as e:
        return (seed, "exc %r"%(e,))
if __name__ == "__main__":
    s0=int(sys.argv[1]); n=int(sys.argv[2])
    t=time.time()
    with Pool(7) as p:
        res=[r for r in p.imap_unordered(run, range(s0, s0+n), chunksize=8) if r]
    print("host campaign seeds %d..%d: %d bad in %.0fs"%(s0, s0+n, len(res), time.time()-t)); print(res[:20])
