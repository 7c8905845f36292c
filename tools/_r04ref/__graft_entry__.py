"""Driver entry points: build() compiles every native piece, smoke() runs one tiny hot-path invocation on cuda:0."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "zigzagboomerang.jl_amd")
PKG_NAME = "zigzagboomerang_jl_amd"


def load_package():
    """Import the package directory `zigzagboomerang.jl_amd/` (not a valid identifier) as `zigzagboomerang_jl_amd`."""
    if PKG_NAME in sys.modules:
        return sys.modules[PKG_NAME]
    spec = importlib.util.spec_from_file_location(PKG_NAME, os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[PKG_NAME] = mod
    spec.loader.exec_module(mod)
    return mod


def build():
    """Compile libpdmp_mi355.so for gfx950 (hipcc cross-compiles without a GPU) and the CPU oracle; import the package."""
    pkg = load_package()
    pkg.build.build(force=False, verbose=True)
    pkg.build.build(force=False, verbose=True, variant="parity")  # + the opt-in cross-implementations the parity suite holds to the oracle
    pkg.build.build_examples(verbose=True)  # C++ host programs on include/pdmp_mi355.hpp (g++)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    oracle_lib.build_oracle()
    pkg._lib.load()  # the C-ABI library must load and export every symbol of include/pdmp_mi355.h
    for name in pkg._lib.EXPORTED_SYMBOLS:
        getattr(pkg._lib.load(), name)


def smoke():
    """One small local-ZigZag run on device 0, checked event-by-event against the CPU oracle."""
    import numpy as np
    pkg = load_package()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    if pkg._lib.device_count() < 1:
        raise RuntimeError("smoke(): no gfx950 device visible (the engine has no CPU fallback)")
    G = pkg.problems.gmrf_precision(8)
    d = G.shape[0]
    rng = np.random.default_rng(0)
    x0 = rng.standard_normal((4, d))
    th0 = rng.choice([-1.0, 1.0], (4, d))
    c = pkg.problems.column_norms(G)
    Z = pkg.ZigZag(G, np.zeros(d))
    tr, (t, x, th), (acc, num), _ = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, 5.0, c, Z, seed=11)
    for k in range(4):
        r = O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, 5.0, seed=11 + k)
        ev = tr[k].events
        assert len(ev) == len(r["events"]), (k, len(ev), len(r["events"]))
        assert np.array_equal(ev["i"], r["events"]["i"])
        assert np.array_equal(ev["t"], r["events"]["t"]) and np.array_equal(ev["x"], r["events"]["x"])
        assert int(num[k]) == r["num"] and np.array_equal(acc[k], r["acc"])
        assert np.array_equal(x[k], r["x"]) and np.array_equal(th[k], r["theta"]) and np.array_equal(t[k], r["t"])
    print("smoke ok: 4 chains, d=%d, %d events bit-identical to the oracle" % (d, sum(len(q.events) for q in tr)))
    # the kernel of the headline workload (8 events per iteration) on a 48 x 48 lattice, same check
    G = pkg.problems.gmrf_precision(48)
    d = G.shape[0]
    x0 = rng.standard_normal((2, d))
    th0 = rng.choice([-1.0, 1.0], (2, d))
    c = pkg.problems.column_norms(G)
    tr, (t, x, th), (acc, num), _ = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, 1.0, c, pkg.ZigZag(G, np.zeros(d)), seed=21)
    for k in range(2):
        r = O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, 1.0, seed=21 + k)
        ev = tr[k].events
        assert len(ev) == len(r["events"]) and int(num[k]) == r["num"]
        assert np.array_equal(ev["i"], r["events"]["i"]) and np.array_equal(ev["t"], r["events"]["t"])
        assert np.array_equal(x[k], r["x"]) and np.array_equal(th[k], r["theta"])
    print("smoke ok: 2 chains, d=%d (8-event kernel), %d events bit-identical to the oracle" % (d, sum(len(q.events) for q in tr)))
    # what bench.py times: the tracked-gradient evaluation (one proposal per lane) -- bit for bit the oracle's sequential statement of the
    # tracked arithmetic, and the same index sequence as the reference's (moving) evaluation with floats to 1e-9
    tr, (t, x, th), (acc, num), _ = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, 1.0, c, pkg.ZigZag(G, np.zeros(d)), seed=21, tracked=True)
    for k in range(2):
        ev = tr[k].events
        rt = O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, 1.0, seed=21 + k, tracked=True)
        assert len(ev) == len(rt["events"]) and int(num[k]) == rt["num"] and np.array_equal(acc[k], rt["acc"])
        for f in ("i", "t", "x", "theta"):
            assert np.array_equal(ev[f], rt["events"][f]), f
        assert np.array_equal(x[k], rt["x"]) and np.array_equal(th[k], rt["theta"])
        r = O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, 1.0, seed=21 + k)
        assert len(ev) == len(r["events"]) and int(num[k]) == r["num"] and np.array_equal(acc[k], r["acc"])
        assert np.array_equal(ev["i"], r["events"]["i"]) and np.array_equal(th[k], r["theta"])
        assert np.allclose(ev["t"], r["events"]["t"], rtol=1e-9, atol=0) and np.allclose(x[k], r["x"], rtol=1e-9, atol=1e-9)
    print("smoke ok: 2 chains, d=%d (tracked gradients): bit-identical to the tracked oracle; indices and counters of the moving evaluation, "
          "times within 1e-9" % d)

    # off the benchmark stencil (round 4): a random symmetric pattern, <= 6 entries per column -- the 8-event kernel of the moving evaluation and the
    # one-proposal-per-lane tracked kernel on a graph whose neighbours are not i +- 1, i +- n
    G = pkg.problems.random_sparse_precision(2500, 6, seed=3)
    d = G.shape[0]
    x0 = rng.standard_normal((2, d))
    th0 = rng.choice([-1.0, 1.0], (2, d))
    c = pkg.problems.column_norms(G)
    for tracked in (False, True):
        tr, (t, x, th), (acc, num), _ = pkg.spdmp(pkg.GaussianTarget(G), 0.0, x0, th0, 1.0, c, pkg.ZigZag(G, np.zeros(d)), seed=31, tracked=tracked)
        for k in range(2):
            r = O.spdmp_zigzag(G, None, G, x0[k], th0[k], c, 1.0, seed=31 + k, tracked=tracked)
            ev = tr[k].events
            assert len(ev) == len(r["events"]) and int(num[k]) == r["num"] and np.array_equal(acc[k], r["acc"])
            for f in ("i", "t", "x", "theta"):
                assert np.array_equal(ev[f], r["events"][f]), f
            assert np.array_equal(x[k], r["x"]) and np.array_equal(th[k], r["theta"]) and np.array_equal(t[k], r["t"])
    print("smoke ok: 2 chains on a random sparse graph (d=%d): moving and tracked evaluation bit-identical to their oracles" % d)


if __name__ == "__main__":
    build()
    if "--smoke" in sys.argv:
        smoke()
