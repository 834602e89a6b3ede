#!/usr/bin/env python3
"""Latency of the B1 surface at the sizes polyclip.m calls it with (private/polyclip.m:1-70: ONE pair per mex call) and
of the two post-step entries that go through the clip batch: wall time per call of szPolyclipBatch / szClipBatch with
1, 8 and 64 pairs of FloeShapes.mat outlines, results compared with the reference Clipper build when oracle/_ref is there.
usage (on the GPU box): python profiles/clip_call_latency.py [reps=200]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from subzero_amd import capi


def pairs(n, seed=3):
    z = np.load(os.path.join(R, "tests", "golden", "floe_shapes.npz"))
    off = z["offsets"] if "offsets" in z else z["off"]
    X = z["x"]; Y = z["y"]
    rng = np.random.default_rng(seed)
    ax, ay, bx, by, ao, bo = [], [], [], [], [0], [0]
    for _ in range(n):
        i, j = rng.integers(0, len(off) - 1, 2)
        xa, ya = X[off[i]:off[i + 1]], Y[off[i]:off[i + 1]]
        xb, yb = X[off[j]:off[j + 1]], Y[off[j]:off[j + 1]]
        xa = xa - xa.mean(); ya = ya - ya.mean()
        s = 0.4 * (xa.max() - xa.min())
        xb = xb - xb.mean() + s * rng.uniform(-1, 1); yb = yb - yb.mean() + s * rng.uniform(-1, 1)
        ax += list(xa); ay += list(ya); bx += list(xb); by += list(yb); ao.append(len(ax)); bo.append(len(bx))
    f = lambda v: np.ascontiguousarray(v, np.float64)
    return f(ax), f(ay), np.asarray(ao, np.int32), f(bx), f(by), np.asarray(bo, np.int32)


def convex_pairs(n, nv=8, seed=5):
    """pairs of convex outlines of nv vertices (what a Voronoi field hands polyclip), half overlapped"""
    rng = np.random.default_rng(seed)
    ax, ay, bx, by, ao, bo = [], [], [], [], [0], [0]
    for _ in range(n):
        for (lx, ly, lo, cx, cy) in ((ax, ay, ao, 0.0, 0.0), (bx, by, bo, 900.0 * rng.uniform(0.3, 1), 700.0 * rng.uniform(-1, 1))):
            t = np.sort(rng.uniform(0, 2 * np.pi, nv))[::-1]
            lx += list(cx + 1000.0 * np.cos(t)); ly += list(cy + 800.0 * np.sin(t)); lo.append(len(lx))
    f = lambda v: np.ascontiguousarray(v, np.float64)
    return f(ax), f(ay), np.asarray(ao, np.int32), f(bx), f(by), np.asarray(bo, np.int32)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    ctx = capi.Context(0)
    for n, gen in ((1, convex_pairs), (8, convex_pairs), (64, convex_pairs), (1, pairs), (8, pairs), (64, pairs)):
        a = gen(n)
        for _ in range(10): out = ctx.polyclip_batch(*a, 1)
        t0 = time.perf_counter()
        for _ in range(reps): out = ctx.polyclip_batch(*a, 1)
        dt = (time.perf_counter() - t0) / reps
        fx = [np.round(v * 4294967296.0).astype(np.int64) for v in (a[0], a[1], a[3], a[4])]
        for _ in range(10): ctx.clip_batch(fx[0], fx[1], a[2], fx[2], fx[3], a[5], 1)
        t0 = time.perf_counter()
        for _ in range(reps): ctx.clip_batch(fx[0], fx[1], a[2], fx[2], fx[3], a[5], 1)
        dt2 = (time.perf_counter() - t0) / reps
        print("pairs per call %3d  vertices in %5d  rings out %3d : szPolyclipBatch %.1f us/call, szClipBatch %.1f us/call (Python wrapper included)"
              % (n, len(a[0]) + len(a[3]), int(out[3][-1]), dt * 1e6, dt2 * 1e6), flush=True)


if __name__ == "__main__":
    main()
