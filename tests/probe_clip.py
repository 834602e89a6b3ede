"""Ad-hoc timing probe for the clip kernel (not a test)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clipgen as g
from subzero_amd import capi
ctx = capi.Context(0)
for name, ps in [("convex60k", g.convex_pairs(60000, 1)), ("shapes20k", g.shape_pairs(20000, 2))]:
    ax, ay, aoff, bx, by, boff = ps.arrays()
    d = [ctx.to_device(v) for v in (ax, ay, aoff, bx, by, boff)]
    for it in range(4):
        st = ctx.clip_batch_device(len(ps), *d, len(ax), len(bx), 1)
    print(name, "pairs", st.pairs, "in_verts", st.input_vertices, "kernel_ms %.3f" % st.kernel_ms,
          "rings", st.rings_out, "verts_out", st.vertices_out, "retried", st.retried, "failed", st.failed,
          "clips/s %.3e" % (st.pairs / (st.kernel_ms * 1e-3)))
    t = time.time(); r = g.run_ref(ps, 1) if g.ref_lib() else g.run_host(ps, 1); dt = time.time() - t
    print("   cpu 1 thread: %.3f s -> %.3e clips/s" % (dt, len(ps) / dt))
ctx.close()
