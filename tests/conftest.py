import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present() -> bool:
    if os.path.exists("/dev/kfd"):
        return True
    try:
        import torch
        return bool(torch.cuda.is_available())
    except ImportError:
        return False


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` in a container without a GPU skips the gpu-marked tests instead of failing in
    Context() (the product has no CPU fallback); on a GPU box nothing is skipped."""
    if any(item.get_closest_marker("gpu") for item in items) and not _gpu_present():
        skip = pytest.mark.skip(reason="needs a real MI355X (no GPU in this container)")
        for item in items:
            if item.get_closest_marker("gpu"):
                item.add_marker(skip)


def pytest_collection_finish(session):
    """PyTorch wheels bundle their own ROCm runtime while librrtx_hip.so links the system one; of two
    runtimes in one process the bundled one has to come up first (the other order leaves torch with "No
    HIP GPUs are available").  GPU tests that use torch may run after tests that only use the library, so
    bring torch's runtime up before the first GPU test, whatever the selection or order."""
    if not any(item.get_closest_marker("gpu") for item in session.items):
        return
    try:
        import torch
    except ImportError:
        return
    if torch.cuda.is_available():
        torch.cuda.init()


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure): builds oracle/_build/librrtx_oracle.so with gcc if stale."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def hip_lib():
    """librrtx_hip.so must exist and load (cross-compiled by __graft_entry__.build())."""
    from rrtqx_3d_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):
        from rrtqx_3d_amd import build
        build.build()
    return _capi.load()
