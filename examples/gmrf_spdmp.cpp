// gmrf_spdmp.cpp -- the reference's scripts/gaussianrandomfield.jl (local ZigZag on a grid-Laplace GMRF, :15-41) written
// against the C++ host mirror include/pdmp_mi355.hpp:   Γ = 0.01 I + gridlaplacian(n, n) (scripts/gridlaplace.jl:4-21),
// ∇ϕ(x, i, Γ) = idot(Γ, i, x), Z = ZigZag(Γ, 0), c[i] = ‖Γ[:, i]‖₂, spdmp(∇ϕ, t0, x0, θ0, T, c, Z, Γ).
//   usage: gmrf_spdmp [n=16] [T=20] [seed] [tracked]      prints one line: d events num acc fnv1a64(payload) t_last
//   (a 4th argument `tracked` selects the tracked-gradient evaluation, Options::tracked: n x n lattices with n*n >= 2048;
//    `parallel:K` runs pdmp::parallel_spdmp with K chunks, the bound = Γ without the entries that couple two chunks, c doubled;
//    `1d` runs n chains of the 1-d Boomerang sampler of src/zigzagboom1d.jl through pdmp::pdmp(..., Boomerang1d))
// tests/test_gpu_cpp_host.py runs it on the GPU box and checks the line against the CPU oracle.
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "pdmp_mi355.hpp"

static pdmp::SparseCSC gmrf_precision(int n, double eps) {
    // graph Laplacian of the n x n lattice (4-neighbour), plus eps on the diagonal; CSC, rows ascending
    pdmp::SparseCSC G;
    G.n = (int64_t)n * n;
    G.colptr.push_back(0);
    for (int r = 0; r < n; ++r) {
        for (int q = 0; q < n; ++q) {
            const int64_t i = (int64_t)r * n + q;
            int deg = (r > 0) + (r < n - 1) + (q > 0) + (q < n - 1);
            if (r > 0) { G.rowval.push_back(i - n); G.nzval.push_back(-1.0); }
            if (q > 0) { G.rowval.push_back(i - 1); G.nzval.push_back(-1.0); }
            G.rowval.push_back(i);
            G.nzval.push_back((double)deg + eps);
            if (q < n - 1) { G.rowval.push_back(i + 1); G.nzval.push_back(-1.0); }
            if (r < n - 1) { G.rowval.push_back(i + n); G.nzval.push_back(-1.0); }
            G.colptr.push_back((int64_t)G.rowval.size());
        }
    }
    return G;
}

static uint64_t fnv1a(uint64_t h, const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t k = 0; k < n; ++k) {
        h ^= b[k];
        h *= 1099511628211ull;
    }
    return h;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 16;
    const double T = argc > 2 ? std::atof(argv[2]) : 20.0;
    pdmp::Options opt;
    if (argc > 3) opt.seed = std::strtoull(argv[3], nullptr, 0);
    const bool par = argc > 4 && std::strncmp(argv[4], "parallel:", 9) == 0;
    const int K = par ? std::atoi(argv[4] + 9) : 0;
    if (argc > 4 && !par && std::strcmp(argv[4], "1d") != 0) opt.tracked = true;
    if (argc > 4 && std::strcmp(argv[4], "1d") == 0) {
        // test/test1d.jl:51-52 through pdmp::pdmp(..., Boomerang1d): n chains from x0 = 1.41 + 0.01 k, θ0 = 0.5, Boomerang1d(1.1, 1.2, 0.5),
        // the noisy gradient, c = 10; prints: n events(total) fnv1a64(all events, chain by chain) acceptance(chain 0)
        try {
            std::vector<double> x0((size_t)n), th0((size_t)n, 0.5);
            for (int k = 0; k < n; ++k) x0[(size_t)k] = 1.41 + 0.01 * k;
            opt.trace_capacity = 50;  // (every chain is resumed many times)
            const auto R = pdmp::pdmp(pdmp::GaussianTarget1d{3.14159265358979323846 / 3, 1.3, 0.1}, x0, th0, T, 10.0, pdmp::Boomerang1d{1.1, 1.2, 0.5}, opt);
            uint64_t h = 14695981039346656037ull;
            size_t total = 0;
            for (const auto& r : R) {
                total += r.trace.size();
                h = fnv1a(h, r.trace.data(), r.trace.size() * sizeof(pdmp_event1d));
            }
            std::printf("%d %zu %016" PRIx64 " %.17g\n", n, total, h, R[0].acceptance);
            return 0;
        } catch (const std::exception& ex) {
            std::fprintf(stderr, "gmrf_spdmp 1d: %s\n", ex.what());
            return 1;
        }
    }
    try {
        pdmp::ZigZag Z;
        Z.Gamma = gmrf_precision(n, 0.01);
        const int64_t d = Z.Gamma.n;
        Z.mu.assign((size_t)d, 0.0);
        pdmp::GaussianTarget target{Z.Gamma, {}};
        std::vector<double> x0((size_t)d), th0((size_t)d), c((size_t)d);
        for (int64_t i = 0; i < d; ++i) {
            x0[(size_t)i] = (double)((i * 37) % 101) / 50.0 - 1.0;
            th0[(size_t)i] = (i % 3 == 0) ? -1.0 : 1.0;
            double s = 0.0;  // c[i] = norm(Γ[:, i], 2), scripts/gaussianrandomfield.jl:33 (sum in row order)
            for (int64_t p = Z.Gamma.colptr[(size_t)i]; p < Z.Gamma.colptr[(size_t)i + 1]; ++p)
                s += Z.Gamma.nzval[(size_t)p] * Z.Gamma.nzval[(size_t)p];
            c[(size_t)i] = std::sqrt(s);
        }
        pdmp::Result<pdmp::FactTrace> R;
        if (par) {
            // test/testparallel.jl:40-49: the bounding Γ is the target's without the cross-chunk entries (kept as zeros on G's pattern, masked)
            std::vector<uint8_t> mask(Z.Gamma.nzval.size(), 1);
            const int64_t k = d / K;
            for (int64_t i = 0; i < d; ++i)
                for (int64_t p = Z.Gamma.colptr[(size_t)i]; p < Z.Gamma.colptr[(size_t)i + 1]; ++p)
                    if (Z.Gamma.rowval[(size_t)p] / k != i / k) {
                        Z.Gamma.nzval[(size_t)p] = 0.0;
                        mask[(size_t)p] = 0;
                    }
            for (auto& v : c) v *= 2.0;
            R = pdmp::parallel_spdmp(K, target, 0.0, x0, th0, T, c, Z, mask, 0.1, opt);
        } else {
            R = pdmp::spdmp(target, 0.0, x0, th0, T, c, Z, opt);
        }
        uint64_t h = 14695981039346656037ull;
        int64_t acc = 0;
        for (const auto& e : R.trace.events) {
            h = fnv1a(h, &e.t, 8);
            h = fnv1a(h, &e.i, 8);
            h = fnv1a(h, &e.x, 8);
            h = fnv1a(h, &e.theta, 8);
        }
        h = fnv1a(h, R.x.data(), R.x.size() * 8);
        h = fnv1a(h, R.theta.data(), R.theta.size() * 8);
        h = fnv1a(h, R.t.data(), R.t.size() * 8);
        for (int64_t a : R.acc) acc += a;
        std::printf("%" PRId64 " %zu %" PRId64 " %" PRId64 " %016" PRIx64 " %.17g\n", d, R.trace.events.size(), R.num, acc, h,
                    R.trace.events.empty() ? 0.0 : R.trace.events.back().t);
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "gmrf_spdmp: %s\n", ex.what());
        return 1;
    }
    return 0;
}
