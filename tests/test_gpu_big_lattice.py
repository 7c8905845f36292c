"""Beyond d = 16384 (-m gpu): the tracked local ZigZag on the 256 x 256 lattice, d = 65536 -- zz_local_trackp_big_kernel, 8192 block bounds in LDS,
four chunks per selection -- bit for bit the oracle's tracked evaluation, index-exact against the moving one; the moving evaluation at that
size runs the one-event kernel and equals its oracle too.  (The reference takes any d: src/sfact.jl:170-179; its own timing notes are
d = 1e4 .. 1.6e5, research/sticky/heart/speed.jl:36-57.)"""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,T", [(256, 0.6), (160, 1.0), (129, 1.5)])
def test_tracked_lattice_beyond_16384(gpu_pkg, n, T):
    pkg = gpu_pkg
    L = pkg._lib
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    c = pkg.problems.column_norms(G)
    nch, seed = 3, 0x5EED0000
    with pkg.Ensemble(nch, d, trace_capacity=int(1.5 * d * T) + 2000) as e:
        e.set_flow(pkg.ZigZag(G, np.zeros(d)))
        e.set_target(pkg.GaussianTarget(G))
        e.set_gradient_tracking(True)
        e.set_state_synthetic(0.0, c, seed)
        e.run(0.4 * T, L.RUN_STOP_BEFORE)  # (two slices: the level-1 bounds are rebuilt from the pairs at every launch)
        e.run(T, L.RUN_STOP_BEFORE)
        assert e.kernel_name() == "zz_local_trackp_big_kernel"
        cn = e.counters()
        assert np.all(cn["status"] == L.CHAIN_OK)
        for k in (0, nch - 1):
            x0, th0 = O.synthetic_state(seed + k, d)
            r = O.spdmp_zigzag(G, None, G, x0, th0, c, T, seed=seed + k, stop_before_T=True, tracked=True)
            ev = e.trace(k, counters=cn)
            assert len(ev) == len(r["events"]) > 0.15 * d * T
            for f in ("i", "t", "x", "theta"):
                assert np.array_equal(ev[f], r["events"][f]), (k, f)
            fs = e.final_state(k, 1)
            assert int(cn["num"][k]) == r["num"] and np.array_equal(fs["acc"][0], r["acc"])
            assert np.array_equal(fs["t"][0], r["t"]) and np.array_equal(fs["x"][0], r["x"]) and np.array_equal(fs["theta"][0], r["theta"])
            if k == 0:  # against the reference's own (moving) evaluation: the same index sequence, floats to 1e-9
                rm = O.spdmp_zigzag(G, None, G, x0, th0, c, T, seed=seed + k, stop_before_T=True)
                assert np.array_equal(ev["i"], rm["events"]["i"]) and np.allclose(ev["t"], rm["events"]["t"], rtol=1e-9, atol=0)


def test_moving_evaluation_at_65536(gpu_pkg):
    pkg = gpu_pkg
    n, T = 256, 0.25
    G = pkg.problems.gmrf_precision(n)
    d = n * n
    c = pkg.problems.column_norms(G)
    with pkg.Ensemble(2, d, trace_capacity=d) as e:
        e.set_flow(pkg.ZigZag(G, np.zeros(d)))
        e.set_target(pkg.GaussianTarget(G))
        e.set_state_synthetic(0.0, c, 99)
        e.run(T, pkg._lib.RUN_STOP_BEFORE)
        cn = e.counters()
        for k in range(2):
            x0, th0 = O.synthetic_state(99 + k, d)
            r = O.spdmp_zigzag(G, None, G, x0, th0, c, T, seed=99 + k, stop_before_T=True)
            ev = e.trace(k, counters=cn)
            assert len(ev) == len(r["events"]) > 1000
            for f in ("i", "t", "x", "theta"):
                assert np.array_equal(ev[f], r["events"][f]), (k, f)
