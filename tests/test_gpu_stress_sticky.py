"""Randomised stress of the sticky ZigZag kernels (-m gpu; sspdmp, src/ss_fact.jl:78-215): random lattices and banded / random sparse precisions, a
bounding Γ of its own or the target's, flow and target means, per-coordinate thaw rates, adapt, `reversible`, `strong_upperbounds`, speeds that are
not one -- on the speculative kernel and (PDMP_KERNEL=seq) the one-event kernel, bit for bit the oracle.  Round 6 (after the flow-mean finding on
the tracked ZigZag kernel: option combinations no hand-written test had).  Seeds are fixed."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kern", ["auto", "seq"])
@pytest.mark.parametrize("case", range(12))
def test_random_sticky_options(gpu_pkg, monkeypatch, case, kern):
    pkg = gpu_pkg
    if kern == "seq":
        monkeypatch.setenv("PDMP_KERNEL", "seq")
    else:
        monkeypatch.delenv("PDMP_KERNEL", raising=False)
    rng = np.random.default_rng(9100 + case)
    kind = int(rng.integers(0, 3))
    if kind == 0:
        n = int(rng.integers(4, 30))
        G = pkg.problems.gmrf_precision(n, eps=float(rng.uniform(0.05, 1.0)))
    elif kind == 1:
        d = int(rng.integers(10, 400))
        w = int(rng.integers(1, 3))
        diags = [np.full(d, 2.0 * w + 1.0 + rng.random())] + [np.full(d - o, -rng.uniform(0.2, 1.0)) for o in range(1, w + 1)]
        G = sp.diags(diags + diags[1:], [0] + list(range(1, w + 1)) + [-o for o in range(1, w + 1)], format="csc")
    else:
        d = int(rng.integers(8, 150))
        R = sp.random(d, d, density=min(1.5 / d, 0.5), random_state=rng, data_rvs=rng.standard_normal, format="csc")
        A = R + R.T
        G = sp.csc_matrix(A + sp.diags(np.asarray(abs(A).sum(axis=0)).ravel() + 1.0))
    G = sp.csc_matrix(G)
    G.sort_indices()
    d = G.shape[0]
    Gb = sp.csc_matrix(0.9 * G) if rng.integers(0, 2) else G
    mu_b = 0.3 * rng.standard_normal(d) if rng.integers(0, 2) else None
    mu_t = (mu_b if (mu_b is not None and rng.integers(0, 2)) else 0.3 * rng.standard_normal(d)) if rng.integers(0, 2) else None
    nch = 2
    x0 = rng.standard_normal((nch, d))
    th0 = rng.choice([-1.5, -1.0, -0.5, 0.5, 1.0, 1.5], (nch, d))
    adapt = bool(rng.integers(0, 2))
    c = pkg.problems.column_norms(G) * (float(rng.uniform(2.0, 4.0)) if not adapt else float(rng.uniform(0.3, 1.5)))
    kappa = rng.uniform(0.2, 3.0, d)
    rev, strong = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    T = float(rng.uniform(4.0, 30.0)) * min(1.0, 40.0 / d)
    seed = 9300 + 10 * case
    refs = [O.sspdmp_zigzag(Gb, mu_b, G, x0[k], th0[k], c, kappa, T, target_mu=mu_t, seed=seed + k, adapt=adapt, reversible=rev, strong_upperbounds=strong)
            for k in range(nch)]
    if any(r["status"] != 0 for r in refs):
        pytest.skip("bound too small for this draw without adapt")
    Z = pkg.ZigZag(Gb, np.zeros(d) if mu_b is None else mu_b)
    tgt = pkg.GaussianTarget(G) if mu_t is None else pkg.GaussianTarget(G, mu_t)
    tr, (t, x, th), (acc, num), cout = pkg.sspdmp(tgt, 0.0, x0, th0, T, c, Z, kappa, seed=seed, adapt=adapt, reversible=rev, strong_upperbounds=strong)
    what = dict(case=case, kern=kern, kind=kind, d=d, own_bound=Gb is not G, mu_b=mu_b is not None, mu_t=mu_t is not None, adapt=adapt, rev=rev, strong=strong)
    for k in range(nch):
        r = refs[k]
        ev, oe = tr[k].events, r["events"]
        assert len(ev) == len(oe), (what, k, len(ev), len(oe))
        for f in ("i", "t", "x", "theta"):
            assert np.array_equal(ev[f], oe[f]), (what, k, f)
        assert (int(acc[k]), int(num[k])) == (r["nacc"], r["num"]), what
        assert np.array_equal(t[k], r["t"]) and np.array_equal(x[k], r["x"]) and np.array_equal(th[k], r["theta"]), what
        assert np.array_equal(cout[k], r["c"]), what
