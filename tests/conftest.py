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
    # a kernel that never returns must not take the whole run with it (pytest-timeout is in the image; a test's own budget is minutes at most):
    # per-test timeout unless the command line gave one
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 900


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


@pytest.fixture(params=["one_wave", "two_waves", "lines"])
def trackp_form(request, monkeypatch):
    """The forms of the one-proposal-per-lane tracked kernel on the same test: zz_local_trackp_kernel (one wavefront per chain), zz_local_trackp2_kernel
    (a helper wavefront per chain: what ensembles of at most 1792 chains run by default) and zz_local_trackl_kernel (the line layout: what ensembles of
    more than 3072 chains run on the plain lattice; a graph the layout does not serve keeps the one-wave form) -- include/pdmp_debug.h:
    pdmp_debug_set_helper_wave / pdmp_debug_set_track_lines, forwarded by engine.Ensemble from PDMP_HELPER_WAVE / PDMP_TRACK_LINES."""
    monkeypatch.setenv("PDMP_HELPER_WAVE", "1" if request.param == "two_waves" else "0")
    monkeypatch.setenv("PDMP_TRACK_LINES", "1" if request.param == "lines" else "0")
    return request.param


@pytest.fixture(scope="session")
def gpu_pkg_parity(gpu_pkg):
    """A SECOND instance of the package bound to lib/libpdmp_mi355.parity.so (build.py --variant parity: the default library's sources plus the
    measured-slower cross-implementations zz_local_exactp_kernel and zz_logistic_rows_kernel, -DPDMP_EXTRA_KERNELS).  Only the tests that hold
    those kernels to the oracle use it; everything else runs on the default library, the one bench.py and the examples load."""
    import importlib.util
    from __graft_entry__ import PKG_DIR
    path = gpu_pkg.build.build(variant="parity")
    name = "zigzagboomerang_jl_amd_parity"
    spec = importlib.util.spec_from_file_location(name, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    old = os.environ.get("PDMP_MI355_LIB")
    os.environ["PDMP_MI355_LIB"] = path
    try:
        spec.loader.exec_module(mod)
        mod._lib.load()
    finally:
        if old is None:
            os.environ.pop("PDMP_MI355_LIB", None)
        else:
            os.environ["PDMP_MI355_LIB"] = old
    assert mod._lib.loaded_path() == path  # (the parity library IS what this instance loaded)
    return mod
