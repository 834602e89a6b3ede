"""Generate tests/golden/clipper_demo_golden.npz: the polygon pair that ships with the reference's copy of Clipper 6.4.2
(private/clipper_ver6.4.2/C#/ConsoleDemo/ConsoleDemo/bin/Release/subj.txt, clip.txt: the input of the library's own console
demo -- a subject ring that touches itself at a vertex and a rectangle across it) and the rings the REFERENCE's Clipper
returns for it, for all four operations (oracle/_ref/libclipper_ref.so = the reference's own clipper.cpp compiled in place),
as polyclip.m would pass them (doubles x 2^32) and moved by fractions of a unit so that the crossings fall between lattice
points too.  Runs only in the build container.  Data only: inputs and expected outputs."""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import clipgen as g

DEMO = "/root/reference/private/clipper_ver6.4.2/C#/ConsoleDemo/ConsoleDemo/bin/Release"


def read_polys(path):
    """the demo's file format: number of polygons; per polygon the number of vertices, then `x, y,` lines"""
    tok = open(path).read().replace(",", " ").split()
    k = 0
    n = int(tok[k]); k += 1
    out = []
    for _ in range(n):
        m = int(tok[k]); k += 1
        out.append(np.array([[float(tok[k + 2 * q]), float(tok[k + 2 * q + 1])] for q in range(m)]))
        k += 2 * m
    return out


def main(out="tests/golden/clipper_demo_golden.npz"):
    subj, clip = read_polys(os.path.join(DEMO, "subj.txt")), read_polys(os.path.join(DEMO, "clip.txt"))
    assert len(subj) == 1 and len(clip) == 1
    ps = g.PairSet()
    for dx, dy in [(0, 0), (0.25, 0.125), (-30.5, 7.75), (10.0 / 3, -20.0 / 7), (75, 0), (0, 40)]:
        ps.add(subj[0], clip[0] + np.array([dx, dy]))
        ps.add(clip[0] + np.array([dx, dy]), subj[0])
    d = {}
    name = "clipper_demo"
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
        print("method", m, "rings per pair", np.diff(r.pair_ring_off).tolist(), "status", r.status.tolist())
    np.savez_compressed(out, **d)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
