import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# On a GPU box, torch must be the first to load a HIP runtime: it ships its own libamdhip64 / libhsa-runtime64
# (same SONAMEs as /opt/rocm's).  Loaded first, libsubzero_hip.so binds to torch's copies and both work; the
# other order leaves torch without a device.  Only the multi-GPU tests need torch at all.
if os.path.exists("/dev/kfd"):
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover
        pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One szContext for the whole GPU session (one process on the card)."""
    from subzero_amd import capi
    c = capi.Context(0)
    yield c
    c.close()
