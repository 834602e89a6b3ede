"""Scene-level golden vectors (tests/golden/step_golden.npz, made by tests/golden/make_step_golden.py from the
CPU oracle): candidate pairs, contact rows and floe state after 1, 10 and 100 timesteps of small scenes.
  * CPU: the oracle still reproduces them bit for bit (a change of its arithmetic must be deliberate: regenerate);
  * CPU: so does the host build of the device pipeline (tests/hostsim);
  * GPU: so does the device step through the C ABI -- without the oracle in the loop."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_step_golden as mk   # noqa: E402  (scene definitions and field list live with the generator)

GOLD = np.load(os.path.join(HERE, "golden", "step_golden.npz"))


def check(name, it, S, pairs, inter_off, inter, exact_stress=True):
    tag = "%s/%d/" % (name, it)
    assert np.array_equal(GOLD[tag + "pairs"], pairs), tag + "pairs"
    assert np.array_equal(GOLD[tag + "inter_off"], inter_off), tag + "inter_off"
    a, b = GOLD[tag + "inter"], inter
    assert a.shape == b.shape and np.array_equal(a[:, 0], b[:, 0]), tag + "partners"
    assert np.array_equal(a[:, 1:], b[:, 1:]), tag + "contact rows"
    for f in mk.FIELDS:
        g, v = GOLD[tag + f], np.asarray(getattr(S, f))
        if f == "Stress" and not exact_stress:   # the device carries the exact sum of the history (step_core.hpp stress_sum_*)
            assert (np.abs(g - v) <= 1e-10 * np.abs(g) + 1.2e-16 * np.abs(S.StressH).sum(1)).all(), tag + f
            continue
        same = (g == v) | (np.isnan(g) & np.isnan(v)) if g.dtype.kind == "f" else (g == v)
        assert same.all(), tag + f


def run(name, stepper, exact_stress):
    S, P, marks = mk.build(name)
    for it in range(max(marks)):
        P.doInt = int(it % 10 == 0)
        pairs, inter_off, inter = stepper(S, P)
        if it + 1 in marks:
            check(name, it + 1, S, pairs, inter_off, inter, exact_stress)


@pytest.mark.parametrize("name", list(mk.SCENES))
def test_oracle_reproduces_step_golden(name):
    import oracle

    def stepper(S, P):
        r = oracle.step(S, P, threads=4)
        return r.pairs, r.inter_off, r.inter
    run(name, stepper, True)


@pytest.mark.parametrize("name", list(mk.SCENES))
def test_host_build_of_device_pipeline_reproduces_step_golden(name):
    import hoststep

    def stepper(S, P):
        r = hoststep.step(S, P)
        return r.pairs, r.inter_off, r.inter
    run(name, stepper, False)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(mk.SCENES))
def test_device_step_reproduces_step_golden(ctx, name):
    from subzero_amd import capi
    S, P, marks = mk.build(name)
    sim = capi.FloeInteractions(ctx, S)
    for it in range(max(marks)):
        P.doInt = int(it % 10 == 0)
        sim.step(P)
        if it + 1 in marks:
            sim.download(capi.SYNC_ALL)
            off, rows = sim.interactions()
            pairs, _n_ext = sim.pairs()
            check(name, it + 1, S, pairs, off, rows, False)
