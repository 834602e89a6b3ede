"""Time per part of the sweep for the slowest workgroups of the clip batch (bench.py's clip_kernel workload): make -C
subzero_amd/csrc diag_clip; SUBZERO_HIP_LIB=subzero_amd/csrc/libsubzero_hip_clipclocks.so python profiles/clip_phases.py.
(The records are kept per workgroup number of the sweep launches and add up over the launches of a run: the slowest
workgroup is, to within a few per cent, the longest pair of the batch.)"""
import ctypes, json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
import bench
from subzero_amd import capi
ctx = capi.Context(0)
line = bench.clip_kernel_line(ctx, False)
print("kernel_ms", line["kernel_ms"], "(with the clocks)")
n = 1 << 16
out = (ctypes.c_ulonglong * (16 * n))()
ctx.L.szDebugClipPhases(out, n)
a = np.frombuffer(out, dtype=np.uint64).reshape(n, 16).astype(float)
a[:, 0] = 0      # (slot 0 also holds the time before a workgroup's first mark: add_ring is not resolved here -- profiles/evolved_phases.py has it)
tot = a.sum(1)
names = {12: "add_ring", 13: "sort minima", 14: "insert_minima", 15: "horizontals", 16: "build_crossings", 17: "top_of_beam", 18: "after loop", 19: "write",
         20: "do_maxima", 21: "top: horizontals", 22: "top: promote", 23: "crossings: order+process"}
for b in np.argsort(-tot)[:3]:
    print("workgroup %d: %.3f ms over the run's launches:" % (b, tot[b] / 1e5), {names.get(k + 12, k + 12): round(a[b, k] / tot[b], 3) for k in range(16) if a[b, k]})
print("all workgroups:", {names.get(k + 12, k + 12): round(a[:, k].sum() / tot.sum(), 3) for k in range(16) if a[:, k].sum()})
ctx.close()
