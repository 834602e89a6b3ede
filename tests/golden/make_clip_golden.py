"""Generate tests/golden/clip_golden.npz: ring pairs and the rings the REFERENCE's Clipper 6.4.2
returns for them (via oracle/_ref/libclipper_ref.so = the reference's own clipper.cpp compiled in
place).  Runs only in the build container.  Data only: inputs and expected outputs."""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import clipgen as g


def main(out="tests/golden/clip_golden.npz"):
    sets = {
        "shapes": g.shape_pairs(100, seed=101, max_verts=120),
        "convex": g.convex_pairs(150, seed=102),
        "lattice": g.lattice_pairs(150, seed=103, grid=6, max_verts=12),
        "rect": g.rect_pairs(100, seed=104),
    }
    # known answer of the reference's own demo, private/clipper_test.m:2-12: unit square vs itself + 0.5
    ka = g.PairSet()
    sq = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], float)
    ka.add(sq, sq + 0.5)
    sets["clipper_test_m"] = ka
    d = {}
    for name, ps in sets.items():
        ax, ay, aoff, bx, by, boff = ps.arrays()
        d[name + "/ax"], d[name + "/ay"], d[name + "/aoff"] = ax, ay, aoff
        d[name + "/bx"], d[name + "/by"], d[name + "/boff"] = bx, by, boff
        for m in range(4):
            r = g.run_ref(ps, m)
            nr = r.pair_ring_off[len(ps)]
            nv = r.ring_off[nr]
            d["%s/m%d/x" % (name, m)] = r.x[:nv].copy()
            d["%s/m%d/y" % (name, m)] = r.y[:nv].copy()
            d["%s/m%d/ring_off" % (name, m)] = r.ring_off[:nr + 1].copy()
            d["%s/m%d/pair_ring_off" % (name, m)] = r.pair_ring_off.copy()
            d["%s/m%d/status" % (name, m)] = r.status.copy()
    np.savez_compressed(out, **d)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
