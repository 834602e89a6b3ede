"""The C-ABI library loads and exports every symbol include/subzero_hip.h declares (no compute calls)."""
import os
import re

from subzero_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "subzero_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sz[A-Z]\w+)\s*\(", txt)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(capi.EXPORTS)


def test_library_exports_every_declared_symbol():
    L = capi.load()
    for name in declared_symbols():
        assert getattr(L, name) is not None
    assert L.szAbiVersion() == 1


def test_no_cpu_fallback_without_gpu():
    import ctypes
    L = capi.load()
    n_gpu = 0
    try:
        import subprocess
        n_gpu = int(os.path.exists("/dev/kfd"))
    except Exception:
        pass
    if n_gpu:
        return
    h = ctypes.c_void_p()
    assert L.szCreate(ctypes.byref(h), 0) == capi.SZ_ERR_NO_DEVICE


def test_orientation_mode():
    import numpy as np
    # clipper(P) (mexclipper.cpp:134-145): counter-clockwise is positive
    assert capi.orientation(np.array([0, 1, 1, 0]), np.array([0, 0, 1, 1])) is True
    assert capi.orientation(np.array([0, 0, 1, 1]), np.array([0, 1, 1, 0])) is False
