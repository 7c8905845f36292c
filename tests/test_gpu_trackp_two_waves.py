"""The two-wave form of the one-proposal-per-lane tracked kernel (-m gpu): zz_local_trackp2_kernel gives every chain a helper wavefront (the chain's
uniforms and their logarithms produced ahead into a ring in LDS, the next windows' lines requested early) and is what ensembles of at most 1792
chains run (seven chains per CU) -- a rank's share of the north star's 4096-chain ensemble on 4 or 8 GPUs (the loop each chain runs:
src/sfact.jl:199-208).  Held to the same bar as the one-wave form: bit for bit the oracle's tracked evaluation, and -- at the widths it is
meant for -- every counter of every chain equal to the one-wave form's."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
SEED0 = 0x5EED0000


def _ensemble(pkg, G, c, nch, cap, helper, seed0=SEED0):
    d = G.shape[0]
    e = pkg.Ensemble(nch, d, trace_capacity=cap)
    e.debug_set_helper_wave(helper)
    e.set_flow(pkg.ZigZag(G, np.zeros(d)))
    e.set_target(pkg.GaussianTarget(G))
    e.set_gradient_tracking(True)
    e.set_state_synthetic(0.0, c, seed0)
    return e


def test_default_form_follows_the_ensemble_width(gpu_pkg):
    """<= 1792 chains (seven per CU): two waves per chain; wider: one (pdmp_debug_last_kernel says which ran)."""
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(64)
    c = pkg.problems.column_norms(G)
    for nch, name in ((3, "zz_local_trackp2_kernel"), (1792, "zz_local_trackp2_kernel"), (1793, "zz_local_trackp_kernel")):
        with _ensemble(pkg, G, c, nch, 0, -1) as e:
            e.run(0.05, pkg._lib.RUN_STOP_BEFORE)
            assert e.kernel_name() == name, (nch, e.kernel_name())
            assert np.all(e.counters()["status"] == pkg._lib.CHAIN_OK)


@pytest.mark.parametrize("nch", [512, 1024, 1792])
def test_strong_scaling_shares_equal_the_one_wave_form_and_the_oracle(gpu_pkg, nch):
    """C3's geometry (d = 16384) at the widths of a rank of the 8- and 4-GPU job, T = 2 in two slices with the trace recycled: every chain's
    proposal count, accepted count and draw count equal the one-wave form's; first and last chain bit for bit the oracle's tracked evaluation
    (every event of the second slice, final clocks, positions, velocities, per-coordinate counts)."""
    pkg = gpu_pkg
    G = pkg.problems.gmrf_precision(128)
    d = G.shape[0]
    c = pkg.problems.column_norms(G)
    T1, T = 1.0, 2.0
    res = {}
    for helper in (0, 1):
        e = _ensemble(pkg, G, c, nch, 2 * d + 1024, helper)
        e.run(T1, pkg._lib.RUN_STOP_BEFORE)
        e.trace_reset()
        e.run(T, pkg._lib.RUN_STOP_BEFORE)
        assert e.kernel_name() == ("zz_local_trackp2_kernel" if helper else "zz_local_trackp_kernel")
        res[helper] = (e, e.counters())
    (e1, c1), (e2, c2) = res[0], res[1]
    for f in ("num", "nacc", "nevents", "ndraw_main", "ntrace", "status"):
        assert np.array_equal(c1[f], c2[f]), f
    assert np.array_equal(c1["t_last"], c2["t_last"])
    for k in (0, nch - 1):
        x0, th0 = O.synthetic_state(SEED0 + k, d)
        r1 = O.spdmp_zigzag(G, None, G, x0, th0, c, T1, seed=SEED0 + k, stop_before_T=True, tracked=True)
        r = O.spdmp_zigzag(G, None, G, x0, th0, c, T, seed=SEED0 + k, stop_before_T=True, tracked=True)
        assert r["status"] == 0
        tail = r["events"][len(r1["events"]):]
        for e, cn in ((e1, c1), (e2, c2)):
            ev = e.trace(k, counters=cn)
            assert len(ev) == len(tail)
            for f in ("i", "t", "x", "theta"):
                assert np.array_equal(ev[f], tail[f]), (k, f)
            fs = e.final_state(k, 1)
            assert int(cn["num"][k]) == r["num"] and np.array_equal(fs["acc"][0], r["acc"])
            assert np.array_equal(fs["t"][0], r["t"]) and np.array_equal(fs["x"][0], r["x"]) and np.array_equal(fs["theta"][0], r["theta"])
    e1.close()
    e2.close()


def test_two_wave_form_with_a_small_trace_segment_and_the_reference_tail(gpu_pkg):
    """Trace segments that fill up inside a launch (the chain pauses with TRACE_FULL and resumes: the ring of draws restarts at the chain's
    draw count), then the reference's tail (the last event may lie beyond T, src/sfact.jl:199-202): the concatenated trace equals the oracle's."""
    pkg = gpu_pkg
    L = pkg._lib
    G = pkg.problems.gmrf_precision(48)
    d = G.shape[0]
    c = pkg.problems.column_norms(G)
    T = 4.0
    e = _ensemble(pkg, G, c, 2, 700, 1, seed0=900)
    got = [[], []]
    for _ in range(200):
        e.run(T, L.RUN_REFERENCE_TAIL)
        cn = e.counters()
        for k in range(2):
            got[k].append(e.trace(k, counters=cn))
        e.trace_reset()
        if not L.needs_rerun(cn["status"]):
            break
    assert e.kernel_name() == "zz_local_trackp2_kernel"
    for k in range(2):
        x0, th0 = O.synthetic_state(900 + k, d)
        r = O.spdmp_zigzag(G, None, G, x0, th0, c, T, seed=900 + k, tracked=True)
        ev = np.concatenate(got[k])
        assert len(ev) == len(r["events"]) > 3 * 700
        for f in ("i", "t", "x", "theta"):
            assert np.array_equal(ev[f], r["events"][f]), (k, f)
    e.close()
