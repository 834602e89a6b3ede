"""Generates tests/golden/offset_golden.npz: inputs and outputs of the REFERENCE's ClipperOffset (oracle/_ref, built from
/root/reference/private/clipper.cpp by oracle/Makefile) through the glue of mexclipper.cpp:146-203, for the offset
form clipper(P, delta, type [, info]).  Run in the build container (needs /root/reference); the fixture is data only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import clipgen as g  # noqa: E402


def main():
    shapes, _ = g.load_shapes()
    rng = np.random.default_rng(2026)
    xs, ys, roff, case_ring0, params, oxs, oys, oroff, case_out0 = [], [], [0], [0], [], [], [], [0], [0]
    for trial in range(48):
        s = shapes[rng.integers(len(shapes))]
        ring = np.round((s - s.mean(0)) * rng.uniform(0.5, 50) * 2 ** 32).astype(np.int64)
        rings = [ring]
        if trial % 6 == 0:
            c = ring.mean(0).astype(np.int64); r = int(0.05 * np.abs(ring - c).max())
            rings.append(np.array([[c[0] - r, c[1] - r], [c[0] - r, c[1] + r], [c[0] + r, c[1] + r], [c[0] + r, c[1] - r]], np.int64))
        join = int(trial % 3)
        info = 0.0 if join == 0 else (float(rng.uniform(0.0005, 0.05)) * 2 ** 32 if join == 1 else float(rng.choice([0.0, rng.uniform(2, 6)])))
        delta = float(rng.choice([-1, 1]) * rng.uniform(0.002, 0.3) * np.abs(ring).max())
        out = g.run_ref_offset(rings, delta, join, info)
        for r in rings:
            xs.append(r[:, 0]); ys.append(r[:, 1]); roff.append(roff[-1] + len(r))
        case_ring0.append(len(roff) - 1)
        params.append((delta, join, info))
        for r in out:
            oxs.append(r[:, 0]); oys.append(r[:, 1]); oroff.append(oroff[-1] + len(r))
        case_out0.append(len(oroff) - 1)
    np.savez_compressed(os.path.join(HERE, "offset_golden.npz"), x=np.concatenate(xs), y=np.concatenate(ys), ring_off=np.array(roff, np.int64),
                        case_ring0=np.array(case_ring0, np.int64), params=np.array(params, np.float64),
                        out_x=np.concatenate(oxs), out_y=np.concatenate(oys), out_ring_off=np.array(oroff, np.int64),
                        case_out0=np.array(case_out0, np.int64))
    print("cases", len(params), "input rings", len(roff) - 1, "output rings", len(oroff) - 1)


if __name__ == "__main__":
    main()
