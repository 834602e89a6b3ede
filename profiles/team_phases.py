"""Where the team pass of the narrow phase (narrow_team.hip) spends its time: per wave, the time its lane 0 (first team)
spends in each part of teamcontact.hpp / teamclip.hpp, from a diagnostic build (make -C subzero_amd/csrc diag_team:
libsubzero_hip_teamphases.so, -DSZ_TEAM_PHASES) on the bench field.  usage: python profiles/team_phases.py [floes] [steps]"""
import ctypes
import os
import sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from subzero_amd import capi
capi.LIB_PATH = os.environ.get('SZ_DIAG_LIB', os.path.join(R, 'subzero_amd', 'csrc', 'libsubzero_hip_teamphases.so'))
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
G = int(os.environ.get("SUBZERO_TEAM", "16"))
S, P = bench.build_workload(n)
ctx = capi.Context(0); sim = capi.FloeInteractions(ctx, S)
nm = []
for it in range(steps):
    P.doInt = int(it % 10 == 0); st = sim.step(P)
    nm.append(st.narrow_ms)
nw = min((st.n_pairs + 64 // G - 1) // (64 // G), 1 << 15)
out = (ctypes.c_ulonglong * (nw * 48))()
ctx.L.szDebugTeamPhases(out, nw)
v = np.array(list(out), float).reshape(nw, 48) / 100.0     # microseconds
names = {0: "pair setup (loads, slice)", 1: "staging", 2: "clip 1: to fixed point", 3: "region to metres, area", 4: "merge flags, closing, InterX",
         5: "centroid, nearest vertices, direction", 6: "clip 2: to fixed point", 7: "polyarea, certificates", 8: "force, row", 9: "bookkeeping",
         10: "clip: closing duplicates", 11: "clip: vertex scan", 12: "clip: ranks, events", 13: "clip: prefix counts",
         14: "clip: scanline states", 15: "clip: crossing points", 16: "clip: emissions", 17: "clip: placement",
         18: "clip: final checks, orientation", 19: "after the clip", 47: "end"}
for k in range(30, 40):
    names[k] = "2nd " + names[k - 20]
tot = v.sum(1)
print("G %d pairs %d waves %d  narrow_ms of the last steps %s; lane 0 of each wave, microseconds" % (G, st.n_pairs, nw, np.round(nm[-5:], 3)))
print("wave total: mean %.2f p50 %.2f p90 %.2f max %.2f" % (tot.mean(), *np.percentile(tot, [50, 90]), tot.max()))
full = v[:, 8] > 0        # waves whose first team went all the way to a contact row
print("waves whose first team wrote a row: %d; their mean total %.2f" % (full.sum(), tot[full].mean()))
for k in sorted(names):
    if v[:, k].sum() > 0:
        print("  %2d %-44s mean over all waves %6.2f  over full-path waves %6.2f  share %5.1f%%" % (k, names[k], v[:, k].mean(), v[full, k].mean() if full.any() else 0, 100.0 * v[:, k].sum() / tot.sum()))
