import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device; run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def pkg():
    from __graft_entry__ import load_package
    return load_package()


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(HERE, "golden", "golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def gpu_pkg(pkg):
    """The package with the native library loaded and a gfx950 device present -- fails loudly otherwise."""
    pkg._lib.load()
    n = pkg._lib.device_count()
    assert n >= 1, "no gfx950 device visible: GPU tests must run on an MI355X box (there is no CPU fallback)"
    return pkg
