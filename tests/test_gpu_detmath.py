"""gfx950 evaluates the shared numerical contract bit-for-bit like the host (-m gpu)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def test_device_math_matches_host_bitwise(gpu_pkg):
    n = 1 << 20
    for seed in (1, 0x5EED0000):
        dev = gpu_pkg._lib.math_probe(seed, n)
        host = O.math_probe(seed, n)
        for r, name in enumerate(["u01", "log", "div", "sqrt", "poisson_time", "randn", "exp", "sincos"]):
            same = (dev[r] == host[r]) | (np.isnan(dev[r]) & np.isnan(host[r]))
            assert same.all(), (name, int((~same).sum()), dev[r][~same][:3], host[r][~same][:3])
        assert np.isinf(host[4]).any() and np.isfinite(host[4]).any()  # both poisson_time outcomes exercised
