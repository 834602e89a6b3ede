"""The team form of the clip (teamclip.hpp: 16 or 32 lanes of a wave per pair, one scanline per lane) ON THE DEVICE, alone,
against the reference's own Clipper build (oracle/_ref; the host build of the general sweep where that is absent) on the
corpora of tests/clipgen.py -- so that its bit-exactness on the GPU does not rest on contact rows alone.
Every pair is either identical to the reference (ring count, first vertex, direction, coordinates) or handed on
(status 100; 104 from a 16-lane team = "more than 16 scanlines, take 32 lanes": the step's classification sends such a
pair to the wide list); the device hands on exactly the pairs the host build of the same source does."""
import numpy as np
import pytest

import clipgen as g

pytestmark = pytest.mark.gpu
ST_WIDE = 104

CORPORA = [
    (g.convex_pairs, 30000, {}, 0.60),
    (g.neighbour_pairs, 30000, {}, 0.50),
    (g.monotone_pairs, 20000, {}, 0.30),
    (g.hull_pairs, 20000, {}, 0.02),
]


@pytest.mark.parametrize("lanes", [16, 32])
@pytest.mark.parametrize("gen,n,kw,min_done", CORPORA)
def test_team_clip_on_the_device_matches_the_reference(ctx, gen, n, kw, min_done, lanes):
    ps = gen(n, seed=4242 + lanes, **kw)
    dev = g.run_team_gpu(ctx, ps, lanes)
    assert lanes == 16 or not (dev.status == ST_WIDE).any()
    dev.status[dev.status == ST_WIDE] = 100
    chk = g.run_ref(ps, 1) if g.ref_lib() is not None else g.run_host(ps, 1)
    bad, done = g.compare_where_done(chk, dev, len(ps))
    assert bad == [], bad[:5]
    assert done >= min_done * len(ps), (done, len(ps))
    # the same source with its lanes as fibers on the host: same pairs completed, same pairs handed on
    sub = g.ArraysPairSet(_head(ps, 3000))
    host = g.run_team(sub, lanes)
    host.status[host.status == ST_WIDE] = 100
    assert np.array_equal(host.status, dev.status[:len(sub)])
    dsub = g.run_team_gpu(ctx, sub, lanes)
    dsub.status[dsub.status == ST_WIDE] = 100
    assert g.compare(host, dsub, len(sub)) == []


def _head(ps, n):
    ax, ay, aoff, bx, by, boff = ps.arrays()
    n = min(n, len(aoff) - 1)
    return (ax[:aoff[n]], ay[:aoff[n]], aoff[:n + 1].copy(), bx[:boff[n]], by[:boff[n]], boff[:n + 1].copy())


def test_team_clip_hands_on_what_is_not_its_case(ctx):
    # outlines of the reference's FloeShapes.mat (tens of vertices, several minima): every pair is handed on, none answered
    ps = g.shape_pairs(500, seed=9)
    dev = g.run_team_gpu(ctx, ps, 32)
    assert (dev.status == 100).all()
