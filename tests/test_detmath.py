"""The shared numerical contract (include/pdmp_detmath.h), evaluated on the host through the oracle library."""
import math

import numpy as np

import oracle_lib as O


def test_philox4x32_10_known_answers():
    # Random123 kat_vectors for philox4x32-10
    assert [hex(v) for v in O.philox([0, 0, 0, 0], [0, 0])] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(v) for v in O.philox([0xffffffff] * 4, [0xffffffff] * 2)] == \
        ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert [hex(v) for v in O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])] == \
        ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_u01_open_interval_and_uniformity():
    L = O.lib()
    u = np.array([L.orc_u01(77, 0, n) for n in range(20000)])
    assert u.min() > 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 4 / math.sqrt(12 * len(u))
    # different streams / seeds decorrelate
    v = np.array([L.orc_u01(77, 1, n) for n in range(20000)])
    assert abs(np.corrcoef(u, v)[0, 1]) < 0.03


def test_log_within_one_ulp_of_libm():
    L = O.lib()
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.random(20000), np.exp(rng.uniform(-700, 700, 20000)), [2.0 ** -53, 1 - 2.0 ** -53, 1.0, 2.0]])
    for x in xs:
        a, b = L.orc_log(float(x)), math.log(float(x))
        assert abs(a - b) <= math.ulp(b) if b != 0 else abs(a) < 1e-300, (x, a, b)


def test_randn_moments():
    L = O.lib()
    z = np.array([L.orc_randn(5, 0, n) for n in range(40000)])
    assert abs(z.mean()) < 4 / math.sqrt(len(z))
    assert abs(z.var() - 1) < 0.05
    assert abs(np.mean(z ** 4) - 3) < 0.3


def test_synthetic_state_shape():
    x, th = O.synthetic_state(0x5EED0000, 4096)
    assert set(np.unique(th)) == {-1.0, 1.0}
    assert abs(x.mean()) < 0.1 and abs(x.std() - 1) < 0.05
    x2, _ = O.synthetic_state(0x5EED0001, 4096)
    assert not np.array_equal(x, x2)
