"""Bouncy particle sampler on gfx950 vs the CPU oracle, through the C ABI (-m gpu): bit-exact events and state."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle_lib as O

pytestmark = pytest.mark.gpu


def check(pkg, G, mu, x0, th0, c, T, lam, rho=0.0, adapt=False, seed=5, factor=2.0, L="chol", local_bound=False,
          subsample=False):
    """L = "chol": the reference's constructor BouncyParticle(Γ, μ, λ; ρ) with B.L = cholesky(Symmetric(Γ)).L (src/types.jl:43);
    otherwise an explicit factor (the reference's 6-field constructor)."""
    B = pkg.BouncyParticle(G, np.zeros(G.shape[0]) if mu is None else mu, lam, rho, **({} if isinstance(L, str) else {"L": L}))
    cc = pkg.LocalBound(np.array([c])) if local_bound else c
    tr, (t, x, th), (acc, num), cout = pkg.pdmp(None, 0.0, x0, th0, T, cc, B, adapt=adapt, seed=seed, factor=factor,
                                                subsample=subsample)
    for k in range(x0.shape[0]):
        r = O.pdmp_bps(G, mu, x0[k], th0[k], c, T, lambda_ref=lam, rho=rho, adapt=adapt, factor=factor, seed=seed + k,
                       ev_cap=200000, mass_L=B.L, local_bound=local_bound, subsample=subsample)
        assert r["status"] == 0
        assert len(tr[k].t) == r["nevents"], (k, len(tr[k].t), r["nevents"])
        assert np.array_equal(tr[k].t, r["t_ev"]) and np.array_equal(tr[k].x, r["x_ev"]) and np.array_equal(tr[k].θ, r["theta_ev"])
        assert (int(acc[k]), int(num[k])) == (r["nacc"], r["num"])
        assert t[k] == r["t"] and np.array_equal(x[k], r["x"]) and np.array_equal(th[k], r["theta"]) and cout[k] == r["c"]
    return tr


@pytest.mark.parametrize("d", [1, 7, 64, 100, 1024])
def test_isotropic_gaussian_config_c2_shape(gpu_pkg, d):
    """Γ = I, λref = 1, c = 1e-3 (values of scripts/not_fact.jl:23-28); d = 1024 is config C2's dimension."""
    rng = np.random.default_rng(d)
    nch = 3
    tr = check(gpu_pkg, sp.identity(d, format="csc"), None, rng.standard_normal((nch, d)), rng.standard_normal((nch, d)),
               1e-3, 25.0 if d < 1024 else 8.0, 1.0, seed=40 + d)
    assert all(np.all(np.diff(q.t) > 0) for q in tr)


def test_general_sparse_precision_and_mean(gpu_pkg):
    """test/maintest.jl:156-172: Γ = S S', λref = 0.5, c = 1.1 with the mass factor L = cholesky(Γ).L of the reference's
    constructor (reflect! / refresh! solve with L and L', src/dynamics.jl:90-97,112-126), plus a non-zero μ and ρ > 0; then the
    same with an explicit identity factor (the identity-mass process, bit-identical to passing no factor to the oracle)."""
    pkg = gpu_pkg
    G = pkg.problems.maintest_precision(8)
    rng = np.random.default_rng(3)
    x0, th0 = rng.standard_normal((4, 8)), rng.standard_normal((4, 8))
    tr = check(pkg, G, None, x0, th0, 1.1, 60.0, 0.5, seed=8)
    check(pkg, G, rng.standard_normal(8), x0, th0, 1.1, 40.0, 0.7, rho=0.3, seed=9)
    G2 = pkg.problems.gmrf_precision(10)  # d = 100: two slots per lane, pentadiagonal gather and a banded factor through LDS
    check(pkg, G2, None, rng.standard_normal((2, 100)), rng.standard_normal((2, 100)), 0.5, 6.0, 1.0, seed=10)
    I8 = sp.identity(8, format="csc")
    tr_i = check(pkg, G, None, x0, th0, 1.1, 60.0, 0.5, seed=8, L=I8)
    assert not np.array_equal(tr[0].t[:40], tr_i[0].t[:40])  # the factor matters
    for k in range(2):  # identity factor == the oracle without a factor
        r = O.pdmp_bps(G, None, x0[k], th0[k], 1.1, 60.0, lambda_ref=0.5, seed=8 + k, ev_cap=200000)
        assert np.array_equal(tr_i[k].t, r["t_ev"]) and np.array_equal(tr_i[k].θ, r["theta_ev"])


def test_reference_bps_envelope_with_cholesky_mass(gpu_pkg):
    """@testset "Bouncy Particle Sampler" (test/maintest.jl:156-172) on the device as the reference runs it -- B =
    BouncyParticle(Γ0, 0, 0.5) INCLUDING its L = cholesky(Symmetric(Γ0)).L, c = 1.1, T = 300, dt = 0.1 -- inside the reference's
    thresholds 2/sqrt(T) (majority of three seeds), every chain bit-identical to the oracle."""
    pkg = gpu_pkg
    G = pkg.problems.maintest_precision(8)
    d, T = 8, 300.0
    rng = np.random.default_rng(3)
    x0, th0 = rng.standard_normal((3, d)), rng.standard_normal((3, d))
    tr = check(pkg, G, None, x0, th0, 1.1, T, 0.5, seed=8)
    S = np.linalg.inv(G.toarray())
    ok = 0
    for k in range(3):
        ts, xs = pkg.trace.discretize(tr[k], 0.1)
        ok += (np.mean(np.abs(xs.mean(0))) < 2 / np.sqrt(T)) and (np.mean(np.abs(np.cov(xs.T) - S)) < 2 / np.sqrt(T))
    assert ok >= 2


def test_mass_factor_is_required_for_a_general_gamma(gpu_pkg):
    """C ABI: BouncyParticle(Γ ≠ I) without its factor is refused (PDMP_ERR_UNSUPPORTED), never run with a silent L = I; a
    malformed factor is PDMP_ERR_INVALID."""
    pkg = gpu_pkg
    G = pkg.problems.maintest_precision(8)
    B = pkg.BouncyParticle(G, np.zeros(8), 0.5)
    x0 = np.zeros((1, 8))
    with pkg.Ensemble(1, 8, sampler=pkg._lib.SAMPLER_BPS, trace_capacity=8) as ens:
        B_no = pkg.BouncyParticle(G, np.zeros(8), 0.5)
        B_no.L = None
        ens.set_flow_bps(B_no)
        with pytest.raises(pkg._lib.PdmpError) as ei:
            ens.set_state_bps(0.0, x0, x0 + 1.0, 1.1, np.array([1], dtype=np.uint64))
        assert ei.value.code == pkg._lib.PDMP_ERR_UNSUPPORTED and "cholesky" in str(ei.value)
        bad = pkg.BouncyParticle(G, np.zeros(8), 0.5, L=sp.csc_matrix(np.triu(B.L.toarray().T)))  # upper triangular
        with pytest.raises(pkg._lib.PdmpError) as ei:
            ens.set_flow_bps(bad)
        assert ei.value.code == pkg._lib.PDMP_ERR_INVALID
        ens.set_flow_bps(B)  # with the factor: accepted
        ens.set_state_bps(0.0, x0, x0 + 1.0, 1.1, np.array([1], dtype=np.uint64))


def test_boomerang_needs_its_mass_factor_spelled_out(gpu_pkg):
    """C ABI: the library never sees a Boomerang's Γ (set_flow_boomerang takes the TARGET's), so it cannot know whether L = I is right:
    set_state_bps without pdmp_ensemble_set_mass_cholesky is PDMP_ERR_UNSUPPORTED; an identity factor is the explicit opt-in."""
    pkg = gpu_pkg
    L = pkg._lib
    d = 8
    G = pkg.problems.maintest_precision(d)
    cp, rv, nz = (np.ascontiguousarray(a) for a in (G.indptr.astype(np.int64), G.indices.astype(np.int64), G.data.astype(np.float64)))
    mu = np.zeros(d)
    x0 = np.zeros((1, d))
    seeds = np.array([1], dtype=np.uint64)
    with pkg.Ensemble(1, d, sampler=L.SAMPLER_BPS, trace_capacity=8) as ens:
        L.check(ens._L.pdmp_ensemble_set_flow_boomerang(ens._h, cp.ctypes.data, rv.ctypes.data, nz.ctypes.data, None, mu.ctypes.data, 0.5, 0.0))
        with pytest.raises(L.PdmpError) as ei:
            ens.set_state_bps(0.0, x0, x0 + 1.0, 1.1, seeds)
        assert ei.value.code == L.PDMP_ERR_UNSUPPORTED and "cholesky" in str(ei.value)
        I = sp.identity(d, format="csc")
        icp, irv, inz = I.indptr.astype(np.int64), I.indices.astype(np.int64), I.data.astype(np.float64)
        L.check(ens._L.pdmp_ensemble_set_mass_cholesky(ens._h, icp.ctypes.data, irv.ctypes.data, inz.ctypes.data))
        ens.set_state_bps(0.0, x0, x0 + 1.0, 1.1, seeds)


def test_local_bound_and_subsample(gpu_pkg):
    """c::LocalBound of the non-factorised sampler (src/not_fact_samplers.jl:29-31, renew branch :65-71; LocalBound(20) is the
    reference's own value, test/maintest.jl:182) and the `subsample` keyword (:53,90), with and without the mass factor."""
    pkg = gpu_pkg
    G = pkg.problems.maintest_precision(8)
    rng = np.random.default_rng(4)
    x0, th0 = rng.standard_normal((3, 8)), rng.standard_normal((3, 8))
    tr = check(pkg, G, None, x0, th0, 20.0, 30.0, 0.5, seed=3, local_bound=True)
    check(pkg, G, None, x0, th0, 20.0, 30.0, 0.5, seed=3, local_bound=True, L=sp.identity(8, format="csc"))
    check(pkg, G, None, x0, th0, 1.1, 60.0, 0.5, seed=4, local_bound=True, rho=0.2)
    trs = check(pkg, G, None, x0, th0, 1.1, 100.0, 0.5, seed=5, subsample=True)
    assert all(len(q.t) > 20 for q in trs) and all(len(q.t) > 10 for q in tr)
    G2 = pkg.problems.gmrf_precision(10)
    check(pkg, G2, None, rng.standard_normal((2, 100)), rng.standard_normal((2, 100)), 8.0, 4.0, 1.0, seed=6, local_bound=True)


def test_adapt_and_violation(gpu_pkg):
    G = sp.identity(4, format="csc") * 3.0
    rng = np.random.default_rng(1)
    x0, th0 = 4 * rng.standard_normal((2, 4)), rng.standard_normal((2, 4))
    # the bound of a Gaussian target with the same Γ is exact: force violations through the refresh-free tail c < 0
    with pytest.raises(RuntimeError, match="Tuning parameter `c` too small"):
        gpu_pkg.pdmp(None, 0.0, x0, th0, 30.0, -0.5, gpu_pkg.BouncyParticle(G, np.zeros(4), 1.0))
    check(gpu_pkg, G, None, x0, th0, -0.5, 30.0, 1.0, adapt=True, seed=5, factor=-2.0)


def test_golden_bps16(gpu_pkg, golden):
    d = 16
    B = gpu_pkg.BouncyParticle(sp.identity(d, format="csc"), np.zeros(d), 1.0)
    with gpu_pkg.Ensemble(1, d, sampler=gpu_pkg._lib.SAMPLER_BPS, trace_capacity=200) as ens:
        ens.set_flow_bps(B)
        ens.set_state_bps(0.0, golden["bps16_x0"][None], golden["bps16_th0"][None], 1e-3, np.array([99], dtype=np.uint64))
        ens.run(1e9)
        cnt = ens.counters()
        assert cnt["status"][0] == gpu_pkg._lib.CHAIN_TRACE_FULL and cnt["ntrace"][0] == 200
        t, x, th = ens.bps_trace(0, counters=cnt)
    assert np.array_equal(t, golden["bps16_t"])
    assert np.array_equal(x[-1], golden["bps16_x_last"]) and np.array_equal(th[-1], golden["bps16_th_last"])
    assert [int(cnt["num"][0]), int(cnt["nacc"][0]), int(cnt["nrefresh"][0])] == golden["bps16_counts"].tolist()


def test_slicing_is_exact(gpu_pkg):
    pkg = gpu_pkg
    d = 32
    G = sp.identity(d, format="csc")
    rng = np.random.default_rng(6)
    x0, th0 = rng.standard_normal((3, d)), rng.standard_normal((3, d))
    with pkg.Ensemble(3, d, sampler=pkg._lib.SAMPLER_BPS, trace_capacity=7) as ens:
        ens.set_flow_bps(pkg.BouncyParticle(G, np.zeros(d), 1.0))
        ens.set_state_bps(0.0, x0, th0, 1e-3, np.arange(3, dtype=np.uint64) + 70)
        ts = [[] for _ in range(3)]
        for Tk, flag in ((3.3, pkg._lib.RUN_STOP_BEFORE), (11.0, pkg._lib.RUN_STOP_BEFORE), (20.0, pkg._lib.RUN_REFERENCE_TAIL)):
            while True:
                ens.run(Tk, flag)
                cnt = ens.counters()
                for k in range(3):
                    ts[k].append(ens.bps_trace(k, counters=cnt)[0])
                ens.trace_reset()
                if not pkg._lib.needs_rerun(cnt["status"]):
                    break
        fs = ens.bps_final_state()
    for k in range(3):
        r = O.pdmp_bps(G, None, x0[k], th0[k], 1e-3, 20.0, lambda_ref=1.0, seed=70 + k, ev_cap=10000)
        assert np.array_equal(np.concatenate(ts[k]), r["t_ev"]) and np.array_equal(fs["x"][k], r["x"])


def test_boomerang_matches_oracle(gpu_pkg):
    """pdmp(∇ϕ!, t0, x0, θ0, T, c, B::Boomerang) (test/maintest.jl:139-154: Γ = S S' target, λref = 0.5, c = 16) with the
    identity mass the device implements: rotation, grad_correct!, the constant bound; bit-exact events and state.  Also a
    diagonal target with non-zero means (register-only path), d = 100 (two slots per lane) and adapt."""
    pkg = gpu_pkg
    rng = np.random.default_rng(21)

    def chk(Gt, mut, muf, x0, th0, c, T, lam, rho=0.0, adapt=False, seed=5, Gf=None):
        d = Gt.shape[0]
        B = pkg.Boomerang(sp.identity(d, format="csc") if Gf is None else Gf, muf, lam, rho)
        tr, (t, x, th), (acc, num), cout = pkg.pdmp(pkg.GaussianTarget(Gt, mut), 0.0, x0, th0, T, c, B, adapt=adapt, seed=seed)
        for k in range(x0.shape[0]):
            r = O.pdmp_bps(Gt, mut, x0[k], th0[k], c, T, lambda_ref=lam, rho=rho, adapt=adapt, seed=seed + k, ev_cap=200000,
                           boomerang_mu=muf, mass_L=B.L)
            assert r["status"] == 0 and r["nevents"] > 10
            assert len(tr[k].t) == r["nevents"], (k, len(tr[k].t), r["nevents"])
            assert np.array_equal(tr[k].t, r["t_ev"]) and np.array_equal(tr[k].x, r["x_ev"]) and np.array_equal(tr[k].θ, r["theta_ev"])
            assert (int(acc[k]), int(num[k])) == (r["nacc"], r["num"])
            assert t[k] == r["t"] and np.array_equal(x[k], r["x"]) and np.array_equal(th[k], r["theta"]) and cout[k] == r["c"]
        return tr

    G = pkg.problems.maintest_precision(8)
    tr = chk(G, None, np.zeros(8), rng.standard_normal((3, 8)), rng.standard_normal((3, 8)), 16.0, 300.0, 0.5, seed=60)
    # the discretised path of a Boomerang trace rotates between events: sample mean near 0 (loose, short run)
    ts, xs = pkg.trace.discretize(tr[0], 0.1)
    assert len(ts) > 2000 and np.mean(np.abs(xs.mean(0))) < 0.5
    Gd = sp.diags(0.5 + rng.random(20), format="csc")
    chk(Gd, rng.standard_normal(20), 0.3 * rng.standard_normal(20), rng.standard_normal((2, 20)), rng.standard_normal((2, 20)),
        4.0, 100.0, 1.0, rho=0.4, seed=61)
    G2 = pkg.problems.gmrf_precision(10)
    chk(G2, None, np.zeros(100), rng.standard_normal((2, 100)), rng.standard_normal((2, 100)), 2.0, 30.0, 0.8, adapt=True, seed=62)
    # test/maintest.jl:145 as written: B = Boomerang(Γ0, 0, 0.5) with Γ0 = Γ, i.e. L = cholesky(Γ).L in reflect!, refresh! and
    # grad_correct! (src/not_fact_samplers.jl:9-12)
    # (with that factor grad_correct! subtracts Γ⁻¹(x − μ), not Γ(x − μ): the constant bound is no bound any more -- the reference
    # marks its covariance test @test_broken -- so `adapt` is on)
    chk(G, None, np.zeros(8), rng.standard_normal((2, 8)), rng.standard_normal((2, 8)), 16.0, 100.0, 0.5, seed=63, Gf=G, adapt=True)
    chk(G2, None, 0.1 * rng.standard_normal(100), rng.standard_normal((2, 100)), rng.standard_normal((2, 100)), 4.0, 10.0, 0.8,
        rho=0.3, seed=64, Gf=G2, adapt=True)


def test_bouncy_particle_with_a_target_of_its_own(gpu_pkg):
    """pdmp(∇ϕ!, t0, x0, θ0, T, c, B::BouncyParticle): ∇ϕ! is the caller's (src/not_fact_samplers.jl:122) while ab(…GlobalBound…) uses the
    FLOW's B.Γ, B.μ (:26-28).  Target Γt(x − μt) ≠ B.Γ(x − B.μ): gradient, rate and reflection from the target, the bound from the flow
    (B.Γ = 1.4 Γt dominates, so c stays small); with the flow's mass factor; with LocalBound (a = c + θ'∇ϕx, v = θ'Γtθ, :29-31); adapt."""
    pkg = gpu_pkg
    rng = np.random.default_rng(12)
    for d, nch, T in ((8, 3, 40.0), (100, 2, 5.0)):
        Gt = pkg.problems.maintest_precision(8) if d == 8 else pkg.problems.gmrf_precision(10, 0.5)
        mut = 0.3 * rng.standard_normal(d)
        mub = mut + 0.05 * rng.standard_normal(d)
        x0, th0 = rng.standard_normal((nch, d)), rng.standard_normal((nch, d))
        # (scale of B.Γ against Γt, bound, adapt): 1.4 Γt dominates the target; 0.5 Γt with a tiny c violates and adapts
        for scale, c, kw in ((1.4, 2.5, dict()), (1.4, 2.5, dict(local_bound=True)), (0.5, 1e-3, dict(adapt=True))):
            Gb = sp.csc_matrix(scale * Gt)
            B = pkg.BouncyParticle(Gb, mub, 0.6, 0.2)
            cc = pkg.LocalBound(np.array([c])) if kw.get("local_bound") else c
            tr, (t, x, th), (acc, num), cout = pkg.pdmp(pkg.GaussianTarget(Gt, mut), 0.0, x0, th0, T, cc, B, seed=77,
                                                        adapt=kw.get("adapt", False))
            for k in range(nch):
                r = O.pdmp_bps(Gb, mub, x0[k], th0[k], c, T, lambda_ref=0.6, rho=0.2, seed=77 + k, ev_cap=200000, mass_L=B.L,
                               target=(Gt, mut), **kw)
                assert r["status"] == 0 and r["nevents"] > 10
                assert np.array_equal(tr[k].t, r["t_ev"]) and np.array_equal(tr[k].x, r["x_ev"]) and np.array_equal(tr[k].θ, r["theta_ev"])
                assert (int(acc[k]), int(num[k])) == (r["nacc"], r["num"]) and cout[k] == r["c"]
                assert np.array_equal(x[k], r["x"]) and np.array_equal(th[k], r["theta"])
            if kw.get("adapt"):
                assert np.all(cout > 1e-3)
    # the target Γt(x − μt) == B.Γ(x − B.μ) handed over explicitly is the plain BouncyParticle, bit for bit
    G = pkg.problems.maintest_precision(8)
    x0, th0 = rng.standard_normal((2, 8)), rng.standard_normal((2, 8))
    B = pkg.BouncyParticle(G, np.zeros(8), 0.5)
    a = pkg.pdmp(None, 0.0, x0, th0, 30.0, 1.1, B, seed=3)
    b = pkg.pdmp(pkg.GaussianTarget(G, np.zeros(8)), 0.0, x0, th0, 30.0, 1.1, B, seed=3)
    for k in range(2):
        assert np.array_equal(a[0][k].t, b[0][k].t) and np.array_equal(a[0][k].x, b[0][k].x)


@pytest.mark.parametrize("d", [1025, 2048, 3000, 4096])
def test_more_than_1024_coordinates(gpu_pkg, d):
    """The reference has no limit on d (src/not_fact_samplers.jl:117-147); beyond 1024 the vectors leave the plain register file (32 / 64
    slots per lane in AGPRs and scratch: the general instantiation).  Isotropic as config C2, and a tridiagonal Γ with its Cholesky factor
    and a mean (every solve goes through d dependent steps: short horizon)."""
    pkg = gpu_pkg
    rng = np.random.default_rng(d)
    nch = 2
    x0, th0 = rng.standard_normal((nch, d)), rng.standard_normal((nch, d))
    tr = check(pkg, sp.identity(d, format="csc"), None, x0, th0, 1e-3, 3.0, 1.0, seed=70 + d)
    assert all(len(q.t) > 2 for q in tr)
    if d <= 2048:
        G = sp.diags([np.full(d - 1, -0.4), np.full(d, 1.2), np.full(d - 1, -0.4)], [-1, 0, 1], format="csc")
        check(pkg, G, 0.1 * rng.standard_normal(d), x0, th0, 0.5, 0.4, 1.0, rho=0.2, seed=90 + d)


def test_4097_coordinates_are_refused(gpu_pkg):
    pkg = gpu_pkg
    with pytest.raises(pkg._lib.PdmpError) as ei:
        pkg.Ensemble(1, 4097, sampler=pkg._lib.SAMPLER_BPS)
    assert ei.value.code == pkg._lib.PDMP_ERR_UNSUPPORTED
