"""The two MATLAB gateways (subzero_amd/mex/) compile against the test double of the MEX API (tests/mexstub/) and
their argument checking behaves like the reference gateway's (same error texts, mexclipper.cpp:13-62,131-132,204-282)
-- everything that can run without a GPU."""
import numpy as np
import pytest

import mexharness as mx


@pytest.fixture(scope="module")
def clipper():
    return mx.Gateway("clipper_mex.cpp", "clipper")


@pytest.fixture(scope="module")
def stepgw():
    return mx.Gateway("subzero_step_mex.cpp", "subzero_step")


def ring(x, y, cls="int64"):
    s = mx.struct(1, ["x", "y"])
    mx.set_field(s, 0, "x", mx.numeric(np.asarray(x), cls)); mx.set_field(s, 0, "y", mx.numeric(np.asarray(y), cls))
    return s


def test_gateways_compile_and_export_mexFunction(clipper, stepgw):
    assert clipper.fn and stepgw.fn


def test_clipper_orientation_mode_needs_no_gpu(clipper):
    # clipper(P): a double column of 0 / 1, one per contour (mexclipper.cpp:134-145)
    s = mx.struct(2, ["x", "y"])
    mx.set_field(s, 0, "x", mx.numeric(np.array([0, 1, 1, 0]), "int64")); mx.set_field(s, 0, "y", mx.numeric(np.array([0, 0, 1, 1]), "int64"))
    mx.set_field(s, 1, "x", mx.numeric(np.array([0, 0, 1, 1]), "int64")); mx.set_field(s, 1, "y", mx.numeric(np.array([0, 1, 1, 0]), "int64"))
    out, = clipper.call(1, s)
    assert mx.to_numpy(out).ravel().tolist() == [1.0, 0.0] and mx.to_numpy(out).dtype == np.float64


@pytest.mark.parametrize("args,text", [
    (lambda: [mx.scalar(1), ring([0, 1, 1], [0, 0, 1]), mx.scalar(1)], "Input needs to be structure."),
    (lambda: [ring([0, 1, 1], [0, 0, 1], "double"), ring([0, 1, 1], [0, 0, 1]), mx.scalar(1)], "Structure field 'x' must be of type INT64."),
    (lambda: [ring([0, 1, 1], [0, 0, 1]), ring([0, 1, 1], [0, 0, 1]), mx.scalar(7)], "Third input must be 0, 1, 2, or 3."),
    (lambda: [ring([0, 1, 1], [0, 0, 1]), ring([0, 1, 1], [0, 0, 1]), mx.row([1, 2])], "Third input must be scalar."),
    (lambda: [ring([0, 1, 1], [0, 0, 1]), ring([0, 1, 1], [0, 0, 1]), mx.scalar(1), mx.scalar(9)], "Fourth input must be 0, 1, 2, or 3."),
    (lambda: [ring([0, 1, 1], [0, 0, 1]), ring([0, 1, 1], [0, 0]), mx.scalar(1)], "Structure fields x and y must be the same length."),
])
def test_clipper_argument_errors_have_the_reference_texts(clipper, args, text):
    with pytest.raises(mx.MexError) as e:
        clipper.call(1, *args())
    assert str(e.value) == text


def test_clipper_needs_one_output(clipper):
    with pytest.raises(mx.MexError) as e:
        clipper.call(2, ring([0, 1, 1], [0, 0, 1]))
    assert str(e.value) == "One output required."


def test_step_gateway_argument_errors(stepgw):
    with pytest.raises(mx.MexError) as e:
        stepgw.call(0, mx.scalar(1))
    assert "subzero_step(command" in str(e.value)
    with pytest.raises(mx.MexError) as e:
        stepgw.call(0, mx.string("step"))
    assert "no floes on the device" in str(e.value)
    with pytest.raises(mx.MexError) as e:
        stepgw.call(1, mx.string("upload"), mx.struct(0, ["Xi"]), mx.scalar(0))
    assert "non-empty struct array" in str(e.value)


def test_floe_struct_round_trip():
    # the harness's own packing of the Floe struct array (what the gateway tests rely on)
    import scenes
    S, P = scenes.pack(12, periodic=True, n_stress=5, n_mc=40)
    S.StressH[:] = np.random.default_rng(1).normal(size=S.StressH.shape)
    F = mx.floe_struct(S)
    T = S.copy()
    for f in mx.SCALARS:
        getattr(T, f)[:] = -1
    T.StressH[:] = 0; T.mcX[:] = 0
    mx.read_floes(F, T, bulk=True)
    import hoststep
    assert hoststep.compare_states(S, T) == [] and np.array_equal(S.mcX, T.mcX) and np.array_equal(S.mcA, T.mcA)
    sh = mx.to_numpy(mx.get_field(F, 3, "StressH"))
    assert sh.shape == (2, 2, 5) and sh[1, 0, 2] == S.StressH[3, 2, 1] and sh[0, 1, 2] == S.StressH[3, 2, 2]


@pytest.mark.parametrize("args,text", [
    (lambda: [ring([0, 9, 9], [0, 0, 9]), mx.row([1, 2])], "Second input must be either a structure (a second polygon) or a scalar double (outset Delta)."),
    (lambda: [ring([0, 9, 9], [0, 0, 9]), mx.scalar(2), mx.string("x")], "Invalid MiterType string."),
    (lambda: [ring([0, 9, 9], [0, 0, 9]), mx.scalar(2), mx.string("round")], "The MITERINFO (fourth argument) must specify the arc tolerance for round joints."),
    (lambda: [ring([0, 9, 9], [0, 0, 9]), mx.scalar(2), mx.string("m"), mx.scalar(-1)], "The MITERINFO must be a positive double."),
    (lambda: [ring([0, 9, 9], [0, 0, 9]), mx.scalar(2), mx.string("m"), mx.row([2, 3])], "The MITERINFO (fourth argument) must be a scalar double."),
    (lambda: [ring([0, 9, 9], [0, 0, 9]), mx.scalar(3), mx.scalar(1)], "second input must be a polygon structure."),
])
def test_clipper_offset_mode_argument_errors_have_the_reference_texts(clipper, args, text):
    # (mexclipper.cpp:146-203, :304-309; every one is raised before the GPU is touched)
    with pytest.raises(mx.MexError) as e:
        clipper.call(1, *args())
    assert str(e.value) == text
