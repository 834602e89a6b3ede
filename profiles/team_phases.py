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
nw = 16384  # (records of workgroups that did not run stay zero and are dropped below)
SL = 50
out = (ctypes.c_ulonglong * (nw * SL))()
ctx.L.szDebugTeamPhases(out, nw)
raw = np.array(list(out), float).reshape(nw, SL)
alive = (raw[:, 49] > 0) & (raw[:, 48] > raw[:, 48].max() - 1e5)    # this launch's workgroups (earlier, larger launches leave records behind)
raw = raw[alive]; nw = int(alive.sum())
t0 = raw[:, 48].min()
start = (raw[:, 48] - t0) / 100.0; end = (raw[:, 49] - t0) / 100.0
v = raw[:, :48] / 100.0     # microseconds
names = {0: "pair setup (loads, slice)", 1: "staging", 2: "clip 1: to fixed point", 3: "region to metres, area", 4: "merge flags, closing, InterX",
         5: "centroid, nearest vertices, direction", 6: "clip 2: to fixed point", 7: "polyarea, certificates", 8: "force, row", 9: "bookkeeping",
         10: "clip: closing duplicates", 11: "clip: vertex scan", 12: "clip: ranks, events", 13: "clip: prefix counts",
         14: "clip: scanline states", 15: "clip: crossing points", 16: "clip: emissions", 17: "clip: placement",
         18: "clip: final checks, orientation", 19: "after the clip", 47: "end"}
for k in range(30, 40):
    names[k] = "2nd " + names[k - 20]
tot = v.sum(1)
print('kernel span %.1f us; waves alive over time:' % end.max())
for t in np.arange(0, end.max(), end.max() / 16):
    print('   t=%6.1f us  running %5d  started %5d' % (t, int(((start <= t) & (end > t)).sum()), int((start <= t).sum())))
dur = end - start
print('wave duration: mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f' % (dur.mean(), *np.percentile(dur, [10, 50, 90]), dur.max()))
idx = np.nonzero(alive)[0]
order = np.argsort(-dur)[:12]
print('slowest workgroups (block index, us, time in part 1 staging / waiting):', [(int(idx[o]), round(float(dur[o]), 1), round(float(v[o, 1]), 1)) for o in order])
half = nw // 2
print('first half of the blocks (hot list first): mean duration %.1f; second half: %.1f' % (dur[:half].mean(), dur[half:].mean()))
print("G %d pairs %d waves %d  narrow_ms of the last steps %s; lane 0 of each wave, microseconds" % (G, st.n_pairs, nw, np.round(nm[-5:], 3)))
print("wave total: mean %.2f p50 %.2f p90 %.2f max %.2f" % (tot.mean(), *np.percentile(tot, [50, 90]), tot.max()))
full = v[:, 8] > 0        # waves whose first team went all the way to a contact row
print("waves whose first team wrote a row: %d; their mean total %.2f" % (full.sum(), tot[full].mean()))
for k in sorted(names):
    if v[:, k].sum() > 0:
        print("  %2d %-44s mean over all waves %6.2f  over full-path waves %6.2f  share %5.1f%%" % (k, names[k], v[:, k].mean(), v[full, k].mean() if full.any() else 0, 100.0 * v[:, k].sum() / tot.sum()))
