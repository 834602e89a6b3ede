#!/usr/bin/env python3
"""Generates tests/golden/step_golden.npz: small scenes stepped by the CPU oracle (oracle/, the restatement of the
reference's MATLAB layers around the reference's own Clipper), with the candidate pairs, the contact rows and the
floe state frozen after 1, 10 and 100 timesteps.  The fixture pins the oracle itself -- a change of
its arithmetic shows up as a diff here -- and lets the GPU suite check the device step without running the oracle.

    python tests/golden/make_step_golden.py

Scenes (seeded, from tests/scenes.py): 10 floes in a walled box (the out-of-box plumbing size), 200 floes in a
walled box, 1000 floes in a periodic box with the reference's ocean eddy as forcing (refreshed every 10th step).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

FIELDS = ["Xi", "Yi", "alpha_i", "Ui", "Vi", "ksi_ice", "h", "mass", "alive", "FxOA", "FyOA", "torqueOA",
          "collision_force", "collision_torque", "OverlapArea", "Stress", "StressCount", "cax", "cay", "kill", "transfer"]
SCENES = {
    "walls10": (dict(n=10, periodic=False, seed=31, vel_seed=32, sigma=0.4, n_stress=8), [1, 10, 100], False),
    "walls200": (dict(n=200, periodic=False, seed=7, vel_seed=8, sigma=0.3, n_stress=8), [1, 10, 100], False),
    "periodic1000": (dict(n=1000, periodic=True, seed=4, vel_seed=2, sigma=0.1, n_stress=4), [1, 10], True),
}


def build(name):
    import scenes
    kw, marks, ocean = SCENES[name]
    S, P = scenes.pack(**kw)
    if ocean:
        P.set_reference_ocean()
    return S, P, marks


def main():
    import oracle
    out = {}
    for name in SCENES:
        S, P, marks = build(name)
        for it in range(max(marks)):
            P.doInt = int(it % 10 == 0)
            r = oracle.step(S, P, threads=os.cpu_count() or 1)
            if it + 1 in marks:
                tag = "%s/%d/" % (name, it + 1)
                out[tag + "pairs"] = r.pairs.astype(np.int32)
                out[tag + "inter_off"] = r.inter_off.astype(np.int32)
                out[tag + "inter"] = r.inter
                for f in FIELDS:
                    out[tag + f] = np.asarray(getattr(S, f)).copy()
        print(name, "steps", max(marks), "contact rows at the end", len(r.inter))
    path = os.path.join(HERE, "step_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
