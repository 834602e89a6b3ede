"""Parity checks that do not lean on the kernels' own definitions (parity items 1 and 3):

  * the oracle in its INDEPENDENT mode -- host libm sin/cos/atan2 and the Monte-Carlo means summed left to right,
    as calc_trajectory.m:139,163-165 reads -- against the device pipeline (host build of the kernel bodies here,
    the GPU in test_independent_gpu.py): positions, angles, velocities, forces, torques and Stress within the
    north-star tolerance of 1e-10 relative after 100+ timesteps;
  * the domain-box ring of the 'dif' clip in all 8 start-vertex / direction orders (floe_interactions.m:31-32 takes
    it from holes(floe2.poly), whose order the reference does not fix): the contact rows must not depend on it;
  * capi.collision_count against calc_collisionNum.m:6 on hand-built interaction arrays.
"""
import numpy as np
import pytest

import hoststep
import oracle
import scenes
from subzero_amd import capi

RTOL = 1e-10   # BASELINE.json north_star: "within 1e-10 relative on forces/torques and floe x/y/theta after N steps"

# fields compared with the tolerance, and the scale each difference is taken relative to
TOL_FIELDS = ["Xi", "Yi", "alpha_i", "Ui", "Vi", "ksi_ice", "collision_force", "collision_torque", "OverlapArea",
              "FxOA", "FyOA", "torqueOA", "Stress", "strain", "cax", "cay", "h", "mass", "inertia_moment",
              "dUi_p", "dVi_p", "dksi_ice_p"]


def rel_diffs(A, B, fields=TOL_FIELDS):
    """max |a - b| / max(|a|, |b|, field scale * 1e-6) per field: relative to the value where it is not tiny against
    the field's own magnitude (a velocity of 1e-18 m/s among velocities of 0.1 m/s has no relative accuracy to keep)."""
    out = {}
    for k in fields:
        a, b = np.asarray(getattr(A, k), np.float64), np.asarray(getattr(B, k), np.float64)
        both_nan = np.isnan(a) & np.isnan(b)
        a = np.where(both_nan, 0.0, a); b = np.where(both_nan, 0.0, b)
        scale = max(np.abs(a).max(), np.abs(b).max())
        if scale == 0:
            out[k] = 0.0
            continue
        den = np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-6 * scale)
        out[k] = float((np.abs(a - b) / den).max())
    return out


NORTH_STAR = ["Xi", "Yi", "alpha_i", "collision_force", "collision_torque", "interactions"]


def compare_step(it, ra, rb, A, B, worst):
    # indices and counts are exact whatever the definitions (north star: bit-exact pair indices / counts)
    assert np.array_equal(ra.pairs, rb.pairs), "step %d: candidate pairs" % it
    assert np.array_equal(ra.inter_off, rb.inter_off), "step %d: contact-row counts" % it
    assert np.array_equal(ra.inter[:, 0], rb.inter[:, 0]), "step %d: partner ids" % it
    for k, v in rel_diffs(A, B).items():
        worst[k] = max(worst.get(k, 0.0), v)
    if len(ra.inter):
        d = np.abs(ra.inter[:, 1:] - rb.inter[:, 1:])
        den = np.maximum(np.maximum(np.abs(ra.inter[:, 1:]), np.abs(rb.inter[:, 1:])), 1e-6 * np.abs(ra.inter[:, 1:]).max(0))
        worst["interactions"] = max(worst.get("interactions", 0.0), float((d / den).max()))


def run_resynced(S, P, steps, step_from, do_int_every=10):
    """Every timestep starts from the SAME state on both sides (the oracle's): the oracle in its independent mode
    advances it, `step_from(state, P)` -- a device pipeline -- advances a copy, and the two results are compared.
    120 steps = 120 different states, each a one-step check of the device's definitions against independent ones,
    free of the growth of differences that the contact dynamics themselves produce (see run_free)."""
    A = S.copy()
    worst = {}
    for it in range(steps):
        P.doInt = int(it % do_int_every == 0)
        B = A.copy()
        with oracle.mode(oracle.MODE_INDEPENDENT):
            ra = oracle.step(A, P, threads=4)
        rb = step_from(B, P)
        compare_step(it, ra, rb, A, B, worst)
    return worst


def run_free(S, P, steps, stepper, do_int_every=10, mode=oracle.MODE_INDEPENDENT, exact_until=None):
    """Oracle (in `mode`) and `stepper` (a device pipeline, or the oracle in another mode) running freely side by
    side from the same start; returns the per-step history of the worst relative difference per field.  Contact
    rows are compared for identical counts / partners while `it < exact_until` (all steps when None)."""
    A = S.copy()
    hist = []
    for it in range(steps):
        P.doInt = int(it % do_int_every == 0)
        with oracle.mode(mode):
            ra = oracle.step(A, P, threads=4)
        rb, B = stepper(P)
        w = {}
        if exact_until is None or it < exact_until:
            compare_step(it, ra, rb, A, B, w)
        else:
            for k, v in rel_diffs(A, B).items():
                w[k] = v
        hist.append(w)
    return A, B, hist


def host_stepper(S):
    B = S.copy()

    def step(P):
        return hoststep.step(B, P), B
    return step


def oracle_stepper(S, mode):
    B = S.copy()

    def step(P):
        with oracle.mode(mode):
            return oracle.step(B, P, threads=4), B
    return step


def colliding_1000():
    """1000 floes, full pack (every floe in contact from the third step on), periodic, eddy ocean: collisions and
    forcing both act on every floe."""
    S, P = scenes.pack(1000, periodic=True, n_stress=8)
    P.set_reference_ocean()
    return S, P


def test_one_step_from_120_states_within_1e10_of_independent_oracle():
    S, P = colliding_1000()
    worst = run_resynced(S, P, 120, lambda B, P: hoststep.step(B, P))
    bad = {k: v for k, v in worst.items() if v > RTOL and k != "strain"}
    assert bad == {}, bad
    # strain is a boundary integral of velocity differences that cancel to ~1e-7 of their terms
    # (calc_trajectory.m:224-234): its own rounding is 1e-9 relative in either implementation
    assert worst["strain"] < 1e-7
    assert max(worst.values()) > 0          # not vacuous: the two sets of definitions do differ in the last bits
    assert worst["collision_force"] > 0 or worst["interactions"] > 0 or worst["Ui"] > 0


def test_configs1_free_run_within_1e10_of_independent_oracle():
    # BASELINE configs[1]: 1000 convex floes, concentration 0.8, periodic (no contacts form in 120 steps at these
    # speeds: this is the forcing / integration path alone), eddy ocean
    S, P = scenes.baseline_config("convex1000")
    A, B, hist = run_free(S, P, 120, host_stepper(S))
    assert A.alive.sum() == B.alive.sum()
    worst = {k: max(h.get(k, 0.0) for h in hist) for k in hist[-1]}
    bad = {k: v for k, v in worst.items() if v > RTOL and k != "strain"}
    assert bad == {}, bad
    assert worst["strain"] < 1e-7


def test_free_run_with_contacts_positions_within_1e10_and_growth_is_the_dynamics_own():
    """Freely running, a packed field amplifies last-bit differences (stiff contacts, the force limiter and the
    acceleration clamp of calc_trajectory.m:42-46,181-204): two runs of the ORACLE ITSELF that differ only in which
    1-ulp-accurate sin/cos they call drift apart by the same orders of magnitude.  So: positions stay within 1e-10
    for the whole run, every north-star field does for the first ten steps, and the device's distance from the
    independent oracle stays within a small factor of the distance between two oracle variants."""
    S, P = colliding_1000()
    steps = 120
    A, B, dev = run_free(S, P, steps, host_stepper(S), exact_until=10)
    for k in ("Xi", "Yi"):
        assert max(h[k] for h in dev) <= RTOL, (k, max(h[k] for h in dev))
    for it in range(10):
        bad = {k: dev[it][k] for k in NORTH_STAR if dev[it].get(k, 0.0) > RTOL}
        assert bad == {}, (it, bad)
    # the oracle with libm trig only vs the oracle with left-to-right sums only: neither is the device's definition
    S2, P2 = colliding_1000()
    _, _, ref = run_free(S2, P2, steps, oracle_stepper(S2, oracle.MODE_SEQSUM), mode=oracle.MODE_LIBM, exact_until=10)
    for k in ("collision_force", "collision_torque", "alpha_i", "Ui", "Vi", "ksi_ice"):
        d = max(h[k] for h in dev[-20:])
        r = max(h[k] for h in ref[-20:])
        assert d <= 1e3 * max(r, 1e-13), (k, d, r)


def test_device_definitions_within_1e10_walled_concave():
    S, P = scenes.concave(36, periodic=False)
    P.set_reference_ocean()
    worst = run_resynced(S, P, 40, lambda B, P: hoststep.step(B, P), do_int_every=5)
    bad = {k: v for k, v in worst.items() if v > RTOL and k != "strain"}
    assert bad == {}, bad


@pytest.mark.parametrize("scene", ["pack_walled", "edge_cases", "uniaxial"])
def test_domain_box_ring_order_is_immaterial(scene):
    if scene == "pack_walled":
        S, P = scenes.pack(150, periodic=False)
        steps = 6
    elif scene == "edge_cases":
        S, P = scenes.edge_cases(periodic=False)
        steps = 4
    else:
        S, P = scenes.baseline_config("uniaxial200")
        steps = 5
    ref = None
    for start in range(4):
        for rev in (False, True):
            A = S.copy()
            rows = []
            with oracle.mode(oracle.wall_order(start, rev)):
                for it in range(steps):
                    P.doInt = int(it == 0)
                    if scene == "uniaxial":
                        scenes.squeeze_walls(P, it)
                    r = oracle.step(A, P, threads=4)
                    rows.append(r.inter.copy())
            if scene == "uniaxial":   # the walls moved: start again from the same box for the next order
                S2, P = scenes.baseline_config("uniaxial200")
            wall_rows = sum(int(np.isinf(r[:, 0]).sum()) for r in rows)
            assert wall_rows > 0, "scene has no wall contact: the test would be vacuous"
            if ref is None:
                ref = (A, rows)
                continue
            for it, (r0, r1) in enumerate(zip(ref[1], rows)):
                assert np.array_equal(r0, r1), "ring order (%d, %s) changes the contact rows at step %d" % (start, rev, it)
            assert hoststep.compare_states(ref[0], A) == []


def test_collision_count_follows_calc_collisionNum():
    # calc_collisionNum.m:6 -- (#rows with a finite partner) / 2 + (#rows with partner Inf); empty -> 0
    assert capi.collision_count(np.zeros((0, 7))) == 0.0
    rows = np.zeros((7, 7))
    rows[:, 0] = [2, 1, 3, np.inf, 5, np.inf, 4]          # five floe-floe rows (two pairs seen twice + one), two wall rows
    assert capi.collision_count(rows) == 5 / 2 + 2
    rows[:, 0] = np.inf
    assert capi.collision_count(rows) == 7.0
    rows[:, 0] = [1, 2, 3, 4, 5, 6, 7]
    assert capi.collision_count(rows) == 3.5             # the reference does not round odd counts
    # and on a real step: every floe-floe contact is listed by both floes, walls by one
    S, P = scenes.pack(120, periodic=False)
    A = S.copy()
    P.doInt = 1
    r = oracle.step(A, P, threads=2)
    finite = np.isfinite(r.inter[:, 0]).sum()
    assert finite % 2 == 0 and capi.collision_count(r.inter) == finite / 2 + np.isinf(r.inter[:, 0]).sum()
