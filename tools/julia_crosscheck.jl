# julia_crosscheck.jl -- pins the build's CPU oracle (oracle/pdmp_oracle.c) to ZigZagBoomerang.jl ITSELF.
#
#   julia --project=<an environment with ZigZagBoomerang v0.13.x> tools/julia_crosscheck.jl [path/to/tests/golden]
#
# NOT executable in the build image (no Julia there; SURVEY.md 8c1) -- written for a maintainer of the reference.  It is the one
# route by which "parity unpinned" can be lifted: the reference's samplers take their uniforms from an `rng` argument
# (spdmp_inner!, src/sfact.jl:73; pdmp_inner!, src/not_fact_samplers.jl:52), so a counter-based `PhiloxRNG <: AbstractRNG`
# whose n-th `rand(rng)` is draw n of the engine's stream (include/pdmp_detmath.h: pdmp_u01(seed, PDMP_STREAM_MAIN, n)) makes the
# REFERENCE replay the very chain the oracle and the gfx950 kernels produce.  The fixtures tests/golden/crosscheck_*.txt hold the
# inputs and the oracle's event lists (floats as IEEE-754 bit patterns; written by tests/golden/export_crosscheck.py).
# Families covered (round 3: all the engine builds): local ZigZag (d8, grid8), BouncyParticle with L, sticky ZigZag, FactBoomerang,
# ZigZag with a refresh clock, the factorised LocalBound, Boomerang with L, the subsampled logistic gradient.
#
# What is compared: the event INDEX sequence and (acc, num) exactly; event times, positions and velocities to 1e-9 relative
# (north star: 1e-6).  They cannot be asked to agree to the last bit: the engine's log / sincos (pdmp_log, pdmp_sincos: < 1 ulp,
# bit-reproducible across x86-64 and gfx950) are not Julia's libm, and the non-factorised samplers' dot products and triangular
# solves have a fixed order in the engine and BLAS/LAPACK's in Julia.
#
# The drivers below restate the ~40 set-up lines of spdmp (src/sfact.jl:162-208), pdmp (src/not_fact_samplers.jl:117-147) and
# sspdmp (src/ss_fact.jl:159-215) only to pass `rng` in; every step of the event loop is the reference's own *_inner! function.
using ZigZagBoomerang, SparseArrays, LinearAlgebra, Random
const ZZB = ZigZagBoomerang

# ------------------------------------------------------------------ Philox4x32-10, keyed per chain (include/pdmp_detmath.h:56-101)
const M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
function philox4x32_10(c0::UInt32, c1::UInt32, c2::UInt32, c3::UInt32, k0::UInt32, k1::UInt32)
    for _ in 1:10
        p0 = UInt64(M0) * UInt64(c0)
        p1 = UInt64(M1) * UInt64(c2)
        n0 = (p1 >> 32) % UInt32 ⊻ c1 ⊻ k0
        n1 = p1 % UInt32
        n2 = (p0 >> 32) % UInt32 ⊻ c3 ⊻ k1
        n3 = p0 % UInt32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 += W0
        k1 += W1
    end
    c0, c1, c2, c3
end
bits64(seed::UInt64, stream::UInt32, n::UInt64) = begin
    r = philox4x32_10(n % UInt32, (n >> 32) % UInt32, stream, 0x00000000, seed % UInt32, (seed >> 32) % UInt32)
    (UInt64(r[1]) << 32) | UInt64(r[2])
end
u01(seed, stream, n) = (Float64(bits64(seed, stream, n) >> 12) + 0.5) * 2.0^-52   # pdmp_bits_to_u01: open interval (0, 1)

mutable struct PhiloxRNG <: AbstractRNG
    seed::UInt64
    stream::UInt32
    n::UInt64          # index of the next draw
end
PhiloxRNG(seed::Integer) = PhiloxRNG(UInt64(seed), 0x00000000, UInt64(0))      # PDMP_STREAM_MAIN
next!(r::PhiloxRNG) = (u = u01(r.seed, r.stream, r.n); r.n += 1; u)
Random.rand(r::PhiloxRNG, ::Random.SamplerTrivial{Random.CloseOpen01{Float64}}) = next!(r)
# rand(rng, (-1, 1)) of the refresh branch (src/sfact.jl:101): not used by the fixtures (λref = 0 there)
Random.randexp(r::PhiloxRNG) = -log(next!(r))                                  # pdmp_randexp
# randn(rng, d) of refresh! (src/dynamics.jl:115): element k (0-based) is Box-Muller branch (k >> 6) & 1 of block
# n + 64 (k >> 7) + (k & 63); the vector consumes 64 ⌈d/128⌉ draws (oracle/pdmp_oracle.c: orc_pdmp_bps, refresh branch)
function boxmuller2(seed, stream, n)
    # pdmp_randn2 (include/pdmp_detmath.h:296-306): u1 from words 0,1 and u2 from words 2,3 of ONE Philox block
    r = philox4x32_10(n % UInt32, (n >> 32) % UInt32, stream, 0x00000000, seed % UInt32, (seed >> 32) % UInt32)
    u1 = (Float64(((UInt64(r[1]) << 32) | UInt64(r[2])) >> 12) + 0.5) * 2.0^-52
    u2 = (Float64(((UInt64(r[3]) << 32) | UInt64(r[4])) >> 12) + 0.5) * 2.0^-52
    rad = sqrt(-2.0 * log(u1))
    s, c = sincospi(2 * u2)
    rad * c, rad * s
end
function Random.randn(r::PhiloxRNG, ::Type{Float64}, d::Integer)
    z = Vector{Float64}(undef, d)
    for k in 0:d-1
        z0, z1 = boxmuller2(r.seed, r.stream, r.n + UInt64(64 * (k >> 7) + (k & 63)))
        z[k+1] = ((k >> 6) & 1) == 1 ? z1 : z0
    end
    r.n += UInt64(64 * cld(d, 128))
    z
end
Random.randn(r::PhiloxRNG, d::Integer) = randn(r, Float64, d)

# ------------------------------------------------------------------ fixtures
f64(h) = reinterpret(Float64, parse(UInt64, h; base = 16))
function read_fixture(path)
    L = readlines(path)
    D = Dict{String,Any}()
    k = 1
    while k <= length(L)
        w = split(L[k])
        key = w[1]
        if key in ("Gamma", "L")
            n, nnz = parse(Int, w[2]), parse(Int, w[3])
            I, J, V = Int[], Int[], Float64[]
            for q in 1:nnz
                a = split(L[k+q])
                push!(I, parse(Int, a[1])); push!(J, parse(Int, a[2])); push!(V, f64(a[3]))
            end
            D[key] = sparse(I, J, V, n, n)
            k += nnz
        elseif key == "events"
            n = parse(Int, w[2])
            D["events"] = [split(L[k+q]) for q in 1:n]
            k += n
        elseif key in ("sampler",)
            D[key] = w[2]
        elseif key in ("seed", "num", "nrefresh", "ksub", "ndraw_global", "ndraw")
            D[key] = parse(Int, w[2])
        elseif key == "acc"
            D[key] = parse.(Int, w[2:end])
        else
            v = f64.(w[2:end])
            D[key] = length(v) == 1 && !(key in ("x0", "theta0", "c")) ? v[1] : v
        end
        k += 1
    end
    D
end
relerr(a, b) = abs(a - b) / max(abs(a), abs(b), 1e-300)

# ------------------------------------------------------------------ spdmp with the rng injected (src/sfact.jl:162-208)
function spdmp_with_rng(rng, ∇ϕ, t0, x0, θ0, T, c, F, args...; factor = 1.8, adapt = false)
    n = length(x0)
    t′ = t0
    t = fill(t′, size(θ0)...)
    t_old = copy(t)
    G1 = [i => rowvals(F.Γ)[nzrange(F.Γ, i)] for i in eachindex(θ0)]
    G = G1
    G2 = [i => setdiff(union((G1[j].second for j in G1[i].second)...), G[i].second) for i in eachindex(G1)]
    x, θ = copy(x0), copy(θ0)
    num = 0
    acc = zeros(Int, length(θ))
    Q = ZZB.SPriorityQueue{Int,Float64}()
    b = [ZZB.ab(G1, i, x, θ, c, F) for i in eachindex(θ)]
    for i in eachindex(θ)
        ZZB.enqueue!(Q, i => poisson_time(b[i], rand(rng)))
    end
    events = Tuple{Float64,Int,Float64,Float64}[]
    while t′ < T
        ev, t, x, θ, t′, (acc, num), c, b, t_old = ZZB.spdmp_inner!(rng, G, G1, G2, ∇ϕ, t, x, θ, Q, c, b, t_old, (acc, num), F,
                                                                   args...; factor = factor, adapt = adapt)
        push!(events, ev)
    end
    events, (t, x, θ), (acc, num), c
end

function check_spdmp(path)
    D = read_fixture(path)
    Γ = D["Gamma"]
    Z = ZigZag(D["scale"] * Γ, zeros(size(Γ, 1)))                       # bound Γ = scale·Γ, target Γ (test/maintest.jl:23)
    ∇ϕ(x, i, Γ) = ZZB.idot(Γ, i, x)                                      # test/maintest.jl:9
    ev, _, (acc, num), _ = spdmp_with_rng(PhiloxRNG(D["seed"]), ∇ϕ, 0.0, D["x0"], D["theta0"], D["T"], copy(D["c"]), Z, Γ)
    ref = D["events"]
    @assert length(ev) == length(ref) "event count $(length(ev)) vs $(length(ref))"
    worst = 0.0
    for (e, r) in zip(ev, ref)
        @assert e[2] == parse(Int, r[2]) "event index differs"
        worst = max(worst, relerr(e[1], f64(r[1])), relerr(e[3], f64(r[3])), relerr(e[4], f64(r[4])))
    end
    @assert num == D["num"] && acc == D["acc"]
    @assert worst < 1e-9
    println(basename(path), ": ", length(ev), " events, index sequence and (acc, num) identical, max relative deviation ", worst)
end

# ------------------------------------------------------------------ BouncyParticle with its mass factor (src/not_fact_samplers.jl:117-147)
function check_bps(path)
    D = read_fixture(path)
    Γ, Lf = D["Gamma"], LowerTriangular(Matrix(D["L"]))
    d = size(Γ, 1)
    B = BouncyParticle(Γ, zeros(d), D["lambda_ref"], D["rho"], nothing, Lf)   # the 6-field constructor: L as given
    ∇ϕ!(y, x) = mul!(y, Γ, x)                                                  # test/maintest.jl:163
    rng = PhiloxRNG(D["seed"])
    Flow, T = B, D["T"]
    ∇w = ZZB.Wrapper(∇ϕ!)
    cb = ZZB.GlobalBound(D["c"])
    t, x, θ, ∇ϕx = 0.0, copy(D["x0"]), copy(D["theta0"]), copy(D["theta0"])
    τref = ZZB.waiting_time_ref(rng, Flow)
    ∇ϕx, v = ∇w(∇ϕx, t, x, θ)
    ∇ϕx = ZZB.grad_correct!(∇ϕx, x, Flow)
    num = acc = 0
    abc = ZZB.ab(x, θ, cb, ∇ϕx, v, Flow)
    t′, renew = ZZB.next_time(t, abc, rand(rng))
    ref = D["events"]
    k = 0
    worst = 0.0
    while t < T
        t, x, θ, (acc, num), cb, abc, (t′, renew), τref, v = ZZB.pdmp_inner!(rng, ∇w, ∇ϕx, t, x, θ, cb, abc, (t′, renew), τref, v,
                                                                             (acc, num), Flow)
        k += 1
        r = ref[k]
        worst = max(worst, relerr(t, f64(r[1])))
        for j in 1:d
            worst = max(worst, abs(x[j] - f64(r[1+j])), abs(θ[j] - f64(r[1+d+j])))
        end
    end
    @assert k == length(ref) && num == D["num"] && acc == D["acc"] "event or proposal counts differ"
    @assert worst < 1e-9
    println(basename(path), ": ", k, " events (reflections and refreshments), counts identical, max deviation ", worst)
end

# ------------------------------------------------------------------ sticky ZigZag: sspdmp draws from the GLOBAL rng (`rand()`, src/ss_fact.jl:54-66,96,179)
const GLOBAL_PHILOX = Ref{Union{Nothing,PhiloxRNG}}(nothing)
function check_sspdmp(path)
    D = read_fixture(path)
    σ2, μ, κ = D["sigma2"], D["mu"], D["kappa"]
    ∇ϕ(x, i, μ) = (x[i] - μ) / σ2                                        # test/sticky.jl:13
    Z = ZigZag(sparse([1.0;;]), [0.0])
    GLOBAL_PHILOX[] = PhiloxRNG(D["seed"])
    # the zero-argument rand() is what ss_fact.jl calls: route it (and only it) to the Philox stream for the duration of the run
    @eval Random.rand() = Main.next!(Main.GLOBAL_PHILOX[])
    trace, _, (acc, num), _ = sspdmp(∇ϕ, 0.0, D["x0"], D["theta0"], D["T"], D["c"], Z, [κ], μ)
    ev = trace.events
    ref = D["events"]
    @assert length(ev) == length(ref) && num == D["num"] && acc == D["acc"]
    worst = 0.0
    for (e, r) in zip(ev, ref)
        @assert e[2] == parse(Int, r[2])
        worst = max(worst, relerr(e[1], f64(r[1])), abs(e[3] - f64(r[3])), abs(e[4] - f64(r[4])))
    end
    @assert worst < 1e-9
    println(basename(path), ": ", length(ev), " events (freeze / thaw / reflection), counts identical, max deviation ", worst)
end

# ================================================================== round 3: the other families the engine builds
#
# Draws the reference takes from Julia's GLOBAL rng inside spdmp_inner! -- `rand(1:n)` twice per refresh (src/sfact.jl:80,84) and
# `waiting_time_ref(F)` = randexp()/λref (:108, src/dynamics.jl:99) -- and inside the scripts' subsampler (`rand(sampler)`,
# scripts/logistic.jl:84) are draws of the chain's PDMP_STREAM_GLOBAL stream in the engine (include/pdmp_detmath.h): routed here the way
# check_sspdmp routes `rand()`.
const GLOBAL_STREAM = Ref{Union{Nothing,PhiloxRNG}}(nothing)
global_bits!() = (g = GLOBAL_STREAM[]; b = bits64(g.seed, g.stream, g.n); g.n += 1; b)
global_randint!(n) = Int(((global_bits!() >> 32) * UInt64(n)) >> 32) + 1                 # pdmp_randint, 1-based
global_u01!() = (Float64(global_bits!() >> 12) + 0.5) * 2.0^-52
function route_global_rng!(seed)
    GLOBAL_STREAM[] = PhiloxRNG(UInt64(seed), 0x00000001, UInt64(0))                     # PDMP_STREAM_GLOBAL
    @eval Random.rand(r::UnitRange{Int}) = first(r) - 1 + Main.global_randint!(length(r))
    @eval Random.randexp() = -log(Main.global_u01!())
end
# rand(rng, (-1, 1)) of the ZigZag refresh (src/sfact.jl:101): the engine takes ONE uniform, u < 1/2 -> -1 (oracle/pdmp_oracle.c)
Random.rand(r::PhiloxRNG, ::Random.SamplerTrivial{Tuple{Int,Int}}) = next!(r) < 0.5 ? -1 : 1
Random.rand(r::PhiloxRNG, t::Tuple{Int,Int}) = next!(r) < 0.5 ? t[1] : t[2]
# randn(rng, Float64) of the FactBoomerang refresh (:103): branch 0 of the Box-Muller pair of ONE Philox block (pdmp_randn)
function Random.randn(r::PhiloxRNG, ::Type{Float64})
    z0, _ = boxmuller2(r.seed, r.stream, r.n)
    r.n += 1
    z0
end
Random.randn(r::PhiloxRNG) = randn(r, Float64)

function compare_fact(path, ev, acc, num, D; tol = 1e-9)
    ref = D["events"]
    @assert length(ev) == length(ref) "event count $(length(ev)) vs $(length(ref))"
    worst = 0.0
    for (e, r) in zip(ev, ref)
        @assert e[2] == parse(Int, r[2]) "event index differs"
        worst = max(worst, relerr(e[1], f64(r[1])), abs(e[3] - f64(r[3])), abs(e[4] - f64(r[4])))
    end
    @assert num == D["num"] && collect(acc) == D["acc"]
    @assert worst < tol
    println(basename(path), ": ", length(ev), " events, index sequence and (acc, num) identical, max deviation ", worst)
end

# spdmp with rng injected and the refresh key enqueued (src/sfact.jl:162-208 incl. :188-190)
function spdmp_refresh_with_rng(rng, ∇ϕ, t0, x0, θ0, T, c, F, args...; factor = 1.8, adapt = false)
    n = length(x0)
    t′ = t0
    t = fill(t′, size(θ0)...)
    t_old = copy(t)
    G1 = [i => rowvals(F.Γ)[nzrange(F.Γ, i)] for i in eachindex(θ0)]
    G = G1
    G2 = [i => setdiff(union((G1[j].second for j in G1[i].second)...), G[i].second) for i in eachindex(G1)]
    x, θ = copy(x0), copy(θ0)
    num = 0
    acc = zeros(Int, length(θ))
    Q = ZZB.SPriorityQueue{Int,Float64}()
    b = [ZZB.ab(G1, i, x, θ, c, F) for i in eachindex(θ)]
    for i in eachindex(θ)
        ZZB.enqueue!(Q, i => poisson_time(b[i], rand(rng)))
    end
    if ZZB.hasrefresh(F)
        ZZB.enqueue!(Q, (n + 1) => ZZB.waiting_time_ref(rng, F))                        # :189, the SEEDED stream
    end
    events = Tuple{Float64,Int,Float64,Float64}[]
    while t′ < T
        ev, t, x, θ, t′, (acc, num), c, b, t_old = ZZB.spdmp_inner!(rng, G, G1, G2, ∇ϕ, t, x, θ, Q, c, b, t_old, (acc, num), F,
                                                                   args...; factor = factor, adapt = adapt)
        push!(events, ev)
    end
    events, (t, x, θ), (acc, num), c
end

# FactBoomerang, test/maintest.jl:114-137; and the ZigZag with a refresh clock (src/sfact.jl:78-114)
function check_refreshing(path)
    D = read_fixture(path)
    Γ = D["Gamma"]
    d = size(Γ, 1)
    F = D["sampler"] == "factboomerang" ? FactBoomerang(D["scale"] * Γ, zeros(d), D["lambda_ref"]) :
                                          ZigZag(D["scale"] * Γ, zeros(d), ones(d); λref = D["lambda_ref"])
    ∇ϕ(x, i, Γ) = ZZB.idot(Γ, i, x)
    route_global_rng!(D["seed"])
    ev, _, (acc, num), _ = spdmp_refresh_with_rng(PhiloxRNG(D["seed"]), ∇ϕ, 0.0, D["x0"], D["theta0"], D["T"], copy(D["c"]), F, Γ)
    compare_fact(path, ev, acc, num, D)   # (refresh events are trace events too: the count and the indices cover them)
end

# factorised LocalBound, src/local.jl:95-149, with the Gaussian target's (∇ϕi, vi) callback (performance/smartbound.jl:45-59)
function check_localbound(path)
    D = read_fixture(path)
    Γ = D["Gamma"]
    d = size(Γ, 1)
    F = ZigZag(Γ, zeros(d))
    ∇ϕv(t, x, θ, i, t′, F, Γ) = (ZZB.idot(Γ, i, x), θ[i] * ZZB.idot(Γ, i, θ))
    rng = PhiloxRNG(D["seed"])
    C = ZZB.LocalBound(copy(D["c"]))
    n, t0, T = d, 0.0, D["T"]
    t′ = t0
    t = fill(t′, d)
    t_old = copy(t)
    G = [i => rowvals(F.Γ)[nzrange(F.Γ, i)] for i in 1:d]
    G2 = [i => setdiff(union((G[j].second for j in G[i].second)...), G[i].second) for i in eachindex(G)]
    x, θ = copy(D["x0"]), copy(D["theta0"])
    num = 0
    acc = zeros(Int, d)
    Q = ZZB.SPriorityQueue{Int,Float64}()
    ∇ϕi, vi = ∇ϕv(t, x, θ, 1, t′, F, Γ)
    b = fill(ZZB.ab(G, 1, x, θ, C, ∇ϕi, vi, F), n)
    renew = zeros(Bool, n)
    for i in 1:d                                                                        # src/local.jl:119-124
        ∇ϕi, vi = ∇ϕv(t, x, θ, i, t′, F, Γ)
        b[i] = ZZB.ab(G, i, x, θ, C, ∇ϕi, vi, F)
        τ, renew[i] = ZZB.next_time(t[i], b[i], rand(rng))
        ZZB.enqueue!(Q, i => τ)
    end
    events = Tuple{Float64,Int,Float64,Float64}[]
    while t′ < T
        ev, t, x, θ, t′, (acc, num), C, (b, renew), t_old = ZZB.spdmp_inner!(rng, G, G2, ∇ϕv, t, x, θ, Q, C, (b, renew), t_old, (acc, num), F, Γ)
        push!(events, ev)
    end
    compare_fact(path, events, acc, num, D)
end

# Boomerang, test/maintest.jl:139-154, with its factor as given
function check_boomerang(path)
    D = read_fixture(path)
    Γ, Lf = D["Gamma"], LowerTriangular(Matrix(D["L"]))
    d = size(Γ, 1)
    B = Boomerang(Γ, zeros(d), D["lambda_ref"], D["rho"], Lf)                           # the constructor with L as given (src/types.jl:59-66)
    ∇ϕ!(y, x) = mul!(y, Γ, x)
    rng = PhiloxRNG(D["seed"])
    Flow, T = B, D["T"]
    ∇w = ZZB.Wrapper(∇ϕ!)
    cb = ZZB.GlobalBound(D["c"])
    t, x, θ, ∇ϕx = 0.0, copy(D["x0"]), copy(D["theta0"]), copy(D["theta0"])
    τref = ZZB.waiting_time_ref(rng, Flow)
    ∇ϕx, v = ∇w(∇ϕx, t, x, θ)
    ∇ϕx = ZZB.grad_correct!(∇ϕx, x, Flow)
    num = acc = 0
    abc = ZZB.ab(x, θ, cb, ∇ϕx, v, Flow)
    t′, renew = ZZB.next_time(t, abc, rand(rng))
    ref = D["events"]
    k = 0
    worst = 0.0
    while t < T
        t, x, θ, (acc, num), cb, abc, (t′, renew), τref, v = ZZB.pdmp_inner!(rng, ∇w, ∇ϕx, t, x, θ, cb, abc, (t′, renew), τref, v,
                                                                             (acc, num), Flow)
        k += 1
        r = ref[k]
        worst = max(worst, relerr(t, f64(r[1])))
        for j in 1:d
            worst = max(worst, abs(x[j] - f64(r[1+j])), abs(θ[j] - f64(r[1+d+j])))
        end
    end
    @assert k == length(ref) && num == D["num"] && acc == D["acc"][1] "event or proposal counts differ"
    @assert worst < 1e-8   # (the rotation's sincos and two triangular solves per event: looser than the factorised samplers)
    println(basename(path), ": ", k, " events, counts identical, max deviation ", worst)
end

# the subsampled logistic gradient with its control variate, scripts/logistic.jl:78-95,107,167 -- the script's helper restated here
# (it is not package code) with the package's own idot_moving! (src/common.jl:33-42) and the subsample indices on the global stream
sigmoid_(x) = inv(one(x) + exp(-x))
function fdot_moving_(A, At, j, t, x, θ, t′, F, μ, y, ny, k)
    rows, vals = rowvals(A), nonzeros(A)
    s = zero(eltype(A))
    r = nzrange(A, j)
    l = length(r)
    for _ in 1:k
        i = first(r) - 1 + global_randint!(l)                                           # rand(sampler), :84
        u = ZZB.idot_moving!(At, rows[i], t, x, θ, t′, F)
        s += l/k*vals[i]*y[rows[i]]*sigmoid_(-u)
        s += l/k*vals[i]*ny[rows[i]]*(-sigmoid_(u))
        u0 = ZZB.idot(At, rows[i], μ)
        s -= l/k*vals[i]*y[rows[i]]*sigmoid_(-u0)
        s -= l/k*vals[i]*ny[rows[i]]*(-sigmoid_(u0))
    end
    s
end
function read_logistic(path)   # the fixture carries the design as "A n p nnz" + triplets; everything else as read_fixture
    L = readlines(path)
    k = findfirst(l -> startswith(l, "A "), L)
    w = split(L[k])
    n, p, nnz = parse(Int, w[2]), parse(Int, w[3]), parse(Int, w[4])
    I, J, V = Int[], Int[], Float64[]
    for q in 1:nnz
        a = split(L[k+q])
        push!(I, parse(Int, a[1])); push!(J, parse(Int, a[2])); push!(V, f64(a[3]))
    end
    tmp = tempname()
    write(tmp, join(vcat(L[1:k-1], L[k+nnz+1:end]), "\n") * "\n")
    D = read_fixture(tmp)
    D["A"] = sparse(I, J, V, n, p)
    D
end
function check_logistic(path)
    D = read_logistic(path)
    A = D["A"]
    At = SparseMatrixCSC(A')
    γ0, ksub = D["gamma0"], D["ksub"]
    Z = ZigZag(D["Gamma"], D["mu"], D["sigma"])
    ∇ϕmoving(t, x, θ, i, t′, F, A, At, μ, y, ny, k) = γ0*x[i] - fdot_moving_(A, At, i, t, x, θ, t′, F, μ, y, ny, k)
    route_global_rng!(D["seed"])
    c = copy(D["c"])
    ev, _, (acc, num), cout = spdmp_with_rng(PhiloxRNG(D["seed"]), ∇ϕmoving, 0.0, D["x0"], D["theta0"], D["T"], c, Z,
                                             ZZB.SelfMoving(), A, At, D["mu"], D["y"], D["ny"], ksub; factor = D["factor"], adapt = true)
    compare_fact(path, ev, acc, num, D)
    @assert cout == D["cout"] "adapted bounds differ"
    @assert GLOBAL_STREAM[].n == D["ndraw_global"]
end

# ------------------------------------------------------------------ the 1-d samplers (src/zigzagboom1d.jl:34-67): every draw is a call on the
# GLOBAL generator -- rand() for the event times, the coin and the noise of ∇ϕhat (test/test1d.jl:10), randn() for Boomerang1d's refreshed
# velocity (:44), randexp() inside poisson_time(λref) (:19-20, src/poissontime.jl:80-82) -- and the engine takes them, in program order, from
# the chain's MAIN stream (oracle/pdmp_oracle.c: orc_pdmp_1d).  Run last: it re-routes the zero-argument generators.
function check_1d(path)
    D = read_fixture(path)
    μ, σ2, noise = D["mu"], D["sigma2"], D["noise"]
    GLOBAL_PHILOX[] = PhiloxRNG(D["seed"])
    @eval Random.rand() = Main.next!(Main.GLOBAL_PHILOX[])
    @eval Random.randn() = randn(Main.GLOBAL_PHILOX[], Float64)
    @eval Random.randexp() = -log(Main.next!(Main.GLOBAL_PHILOX[]))
    ∇ϕ(x) = noise == 0 ? (x - μ)/σ2 : (x - μ)/σ2 + noise*(rand() - 0.5)                       # test/test1d.jl:9-10
    Flow = D["sampler"] == "zigzag1d" ? ZigZag1d() : Boomerang1d(D["b_sigma"], D["b_mu"], D["b_lambda"])
    out, ratio = ZZB.pdmp(∇ϕ, D["x0"][1], D["theta0"][1], D["T"], D["c"][1], Flow)
    ref = D["events"]
    @assert length(out) == length(ref) "event counts differ: $(length(out)) vs $(length(ref))"
    worst = 0.0
    for (e, r) in zip(out, ref)
        worst = max(worst, relerr(e[1], f64(r[1])), abs(e[2] - f64(r[2])), abs(e[3] - f64(r[3])))
    end
    @assert worst < 1e-9
    @assert abs(ratio - D["acc"][1]/D["num"]) < 1e-15
    @assert GLOBAL_PHILOX[].n == D["ndraw"]
    println(basename(path), ": ", length(out), " events, acc/num identical, max deviation ", worst)
end

dir = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "tests", "golden")
check_spdmp(joinpath(dir, "crosscheck_spdmp_d8.txt"))
check_spdmp(joinpath(dir, "crosscheck_spdmp_grid8.txt"))
check_bps(joinpath(dir, "crosscheck_bps_d8.txt"))
check_sspdmp(joinpath(dir, "crosscheck_sspdmp_1d.txt"))
check_refreshing(joinpath(dir, "crosscheck_factboomerang_d8.txt"))
check_refreshing(joinpath(dir, "crosscheck_zigzag_refresh_d8.txt"))
check_localbound(joinpath(dir, "crosscheck_localbound_d8.txt"))
check_boomerang(joinpath(dir, "crosscheck_boomerang_d8.txt"))
check_logistic(joinpath(dir, "crosscheck_logistic_p10.txt"))
check_1d(joinpath(dir, "crosscheck_zigzag1d.txt"))
check_1d(joinpath(dir, "crosscheck_boomerang1d.txt"))
println("oracle == ZigZagBoomerang.jl on all fixtures")
