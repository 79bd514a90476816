import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def golden(name: str):
    """One committed fixture (data only; arrays named ref_* come from the reference itself)."""
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu_tracer():
    """One rthip context on GPU 0 for the whole session (contexts are reusable)."""
    from raytracer_amd import tracer as T

    t = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=8, sky=True), device_id=0)
    yield t
    T.rt_tracer_cleanup(t)
