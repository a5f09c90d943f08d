import os
import sys

# torch bundles its own HIP runtime; when libfdcm_hip.so (linked against /opt/rocm's) is loaded first, a later
# `import torch` in the same process finds no device.  Tests that hand torch device buffers to the library need both,
# so torch comes first whatever the collection order.
try:
    import torch  # noqa: F401
except Exception:  # not needed by the tests that do not use it
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU-only sweep")


def _gpu_visible():
    """True when libfdcm_hip.so loads and reports at least one device (no torch import needed)."""
    try:
        import ctypes as C
        from openfdcm_amd import _capi
        n = C.c_int()
        return _capi.lib().fdcm_device_count(C.byref(n)) == 0 and n.value >= 1
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a box without a GPU skips the `gpu` tests instead of failing them.  When the
    marker is selected explicitly (-m gpu, as the driver does on the MI355X box) nothing is skipped: a missing
    device or library must fail loudly there."""
    import pytest
    if "gpu" in (config.getoption("-m") or ""):
        return
    if _gpu_visible():
        return
    skip = pytest.mark.skip(reason="no ROCm device visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
