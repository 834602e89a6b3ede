"""The B1 clip batch of bench.py's `clip_kernel` object alone (2e4 pairs of FloeShapes.mat outlines, intersection): the
workload of profiles/r0x_clip_kernel_stats.csv (rocprofv3 --kernel-trace --stats -- python3 profiles/clip_batch.py)."""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import bench
from subzero_amd import capi
ctx = capi.Context(0)
print(json.dumps(bench.clip_kernel_line(ctx, False)))
ctx.close()
