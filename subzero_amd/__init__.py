"""subzero_amd -- MI355X (gfx950) implementation of SubZero's per-timestep floe-interaction hot path.

The compute lives in ``csrc/libsubzero_hip.so`` (hand-written HIP behind the C ABI declared in
``include/subzero_hip.h``).  This package is the thin host-side mirror of the reference's call surface
(``polyclip``, ``floe_interactions_all``); it holds no arithmetic of its own and has no CPU fallback:
importing :mod:`subzero_amd.capi` without the built library, or creating a context without a gfx950
GPU, raises.
"""
__version__ = "0.1.0"
