"""Parameter carriers mirroring the reference's dynamics descriptors (src/types.jl).

The reference passes a gradient closure ∇ϕ; a closure cannot cross the C ABI, so targets are an
enumerated set of device-resident families (GaussianTarget so far: the closure of
scripts/gaussianrandomfield.jl:25 and test/maintest.jl:9).
"""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import scipy.sparse as sp


def _csc(A):
    A = sp.csc_matrix(A, dtype=np.float64)
    A.sort_indices()
    A.sum_duplicates()
    return A


@dataclass
class ZigZag:
    """ZigZag(Γ, μ, σ=diag(Γ).^(-0.5); λref=0.0, ρ=0.0)  -- src/types.jl:19-27.

    Γ is the sparse precision used for the affine bounds (and whose column pattern defines the local
    neighbourhoods G1/G2, src/sfact.jl:170-179); μ the approximate target mean.
    """
    Γ: sp.csc_matrix
    μ: np.ndarray
    σ: Optional[np.ndarray] = None
    λref: float = 0.0
    ρ: float = 0.0

    def __post_init__(self):
        self.Γ = _csc(self.Γ)
        self.μ = np.ascontiguousarray(self.μ, dtype=np.float64)
        if self.σ is None:
            self.σ = np.asarray(self.Γ.diagonal(), dtype=np.float64) ** (-0.5)
        self.σ = np.ascontiguousarray(self.σ, dtype=np.float64)

    @property
    def ρ̄(self):
        return float(np.sqrt(1 - self.ρ ** 2))


@dataclass
class FactBoomerang:
    """FactBoomerang(Γ, μ, λ, σ=diag(Γ).^(-0.5); ρ=0.0) -- src/types.jl:71-79: factorised Boomerang dynamics preserving
    N(μ, inv(Diagonal(Γ))), refreshment rate λ > 0."""
    Γ: sp.csc_matrix
    μ: np.ndarray
    λref: float
    σ: Optional[np.ndarray] = None
    ρ: float = 0.0

    def __post_init__(self):
        self.Γ = _csc(self.Γ)
        self.μ = np.ascontiguousarray(self.μ, dtype=np.float64)
        if self.σ is None:
            self.σ = np.asarray(self.Γ.diagonal(), dtype=np.float64) ** (-0.5)
        self.σ = np.ascontiguousarray(self.σ, dtype=np.float64)


def _is_identity(G):
    return G.nnz == G.shape[0] and np.array_equal(G.indices, np.arange(G.shape[0])) and bool(np.all(G.data == 1.0))


def cholesky_lower(Γ):
    """L = cholesky(Symmetric(Γ)).L (src/types.jl:43,66) as a lower-triangular CSC matrix, factorised densely in the NATURAL
    ordering.  For a dense Γ this is the reference's factor up to LAPACK rounding; for a SparseMatrixCSC the reference calls
    CHOLMOD, whose `.L` is the factor of the fill-reducing PERMUTATION of Γ -- a different (equally valid) mass matrix that a
    caller who wants it bit for bit passes explicitly as `L=`."""
    A = Γ.toarray() if sp.issparse(Γ) else np.asarray(Γ, dtype=np.float64)
    A = np.triu(A) + np.triu(A, 1).T  # Symmetric(Γ) reads the upper triangle
    L = sp.csc_matrix(np.tril(np.linalg.cholesky(A)))
    L.sort_indices()
    return L


@dataclass
class BouncyParticle:
    """BouncyParticle(Γ, μ, λ; ρ=0.0) -- src/types.jl:35-45.  L is the mass factor the reference stores in the struct
    (`cholesky(Symmetric(Γ)).L`, :43): computed here when not given (None and Γ = I: identity); the 6-field constructor
    BouncyParticle(Γ, μ, λ, ρ, U, L) of the reference corresponds to passing `L=` explicitly."""
    Γ: sp.csc_matrix
    μ: np.ndarray
    λref: float
    ρ: float = 0.0
    L: Optional[sp.csc_matrix] = None

    def __post_init__(self):
        self.Γ = _csc(self.Γ)
        self.μ = np.ascontiguousarray(self.μ, dtype=np.float64)
        if self.L is None:
            self.L = None if _is_identity(self.Γ) else cholesky_lower(self.Γ)
        else:
            self.L = _csc(self.L)


@dataclass
class LocalBound:
    """LocalBound(c) -- src/types.jl:121-123: pass as `c` to spdmp to select the bounds of src/local.jl."""
    c: np.ndarray


@dataclass
class Boomerang:
    """Boomerang(Γ, μ, λ; ρ=0.0) -- src/types.jl:59-66: Hamiltonian dynamics preserving N(μ, ·) with refreshment rate λ; Γ enters
    through its factor L = cholesky(Symmetric(Γ)).L only (reflect!, refresh!, grad_correct!), see BouncyParticle."""
    Γ: sp.csc_matrix
    μ: np.ndarray
    λref: float
    ρ: float = 0.0
    L: Optional[sp.csc_matrix] = None

    def __post_init__(self):
        self.Γ = _csc(self.Γ)
        self.μ = np.ascontiguousarray(self.μ, dtype=np.float64)
        if self.L is None:
            self.L = None if _is_identity(self.Γ) else cholesky_lower(self.Γ)
        else:
            self.L = _csc(self.L)


@dataclass
class ZigZag1d:
    """ZigZag1d() -- src/types.jl:82-86: the 1-d ZigZag (x(τ), θ(τ)) = (x + θτ, θ), src/dynamics.jl:66-68."""


@dataclass
class Boomerang1d:
    """Boomerang1d(Σ, μ, λ) / Boomerang1d(μ, λ) / Boomerang1d(λ) -- src/types.jl:89-100: rotation around μ (src/dynamics.jl:79-82),
    refreshment θ ~ N(0, Σ) at rate λref."""
    Σ: float = 1.0
    μ: float = 0.0
    λref: float = 1.0

    def __init__(self, *a):
        if len(a) == 1:
            self.Σ, self.μ, self.λref = 1.0, 0.0, float(a[0])
        elif len(a) == 2:
            self.Σ, self.μ, self.λref = 1.0, float(a[0]), float(a[1])
        elif len(a) == 3:
            self.Σ, self.μ, self.λref = float(a[0]), float(a[1]), float(a[2])
        else:
            raise TypeError("Boomerang1d(λ), Boomerang1d(μ, λ) or Boomerang1d(Σ, μ, λ)")


@dataclass
class GaussianTarget1d:
    """∇ϕ(x) = (x − μ)/σ² [+ noise·(rand() − 0.5)]: the closures of test/test1d.jl:9-10 (`∇ϕ`, `∇ϕhat` with noise = 0.1)."""
    μ: float = 0.0
    σ2: float = 1.0
    noise: float = 0.0


@dataclass
class GaussianTarget:
    """∇ϕ(x, i) = Γ[:, i]·x  [− Γ[:, i]·μ]  (idot, src/common.jl:16-24): the device-resident stand-in
    for the reference's `∇ϕ(x, i, Γ) = idot(Γ, i, x)` closure + its `args... = (Γ,)`."""
    Γ: sp.csc_matrix
    μ: Optional[np.ndarray] = None

    def __post_init__(self):
        self.Γ = _csc(self.Γ)
        if self.μ is not None:
            self.μ = np.ascontiguousarray(self.μ, dtype=np.float64)


@dataclass
class LogisticTarget:
    """Subsampled logistic-regression gradient with a control variate at μ, evaluated with SelfMoving():
    ∇ϕmoving(t,x,θ,i,t′,F,A,At,μ,y,ny,k) = γ0*x[i] − fdot_moving(A,At,i,...)  (scripts/logistic.jl:78-95,107,167) --
    the device-resident stand-in for that closure and its `args = (SelfMoving(), A, At, μ, y, ny, k)`."""
    A: sp.csc_matrix      # n x p design
    y: np.ndarray         # [n] successes
    ny: np.ndarray        # [n] failures (m - y)
    μ: np.ndarray         # [p] control-variate point (the mode)
    γ0: float = 0.01
    k: int = 10

    def __post_init__(self):
        self.A = _csc(self.A)
        self.At = _csc(self.A.T)
        self.y = np.ascontiguousarray(self.y, dtype=np.float64)
        self.ny = np.ascontiguousarray(self.ny, dtype=np.float64)
        self.μ = np.ascontiguousarray(self.μ, dtype=np.float64)


@dataclass
class FactTrace:
    """FactTrace(F, t0, x0, θ0, events) -- src/trace.jl:7-13; events are (t, i, x_i, θ_i), i 0-based."""
    F: object
    t0: float
    x0: np.ndarray
    θ0: np.ndarray
    events: np.ndarray = field(default_factory=lambda: np.empty(0))

    def __len__(self):  # Base.length(FT::Trace) = 1 + length(FT.events), src/trace.jl:42
        return 1 + len(self.events)


@dataclass
class PDMPTrace:
    """PDMPTrace(F, t0, x0, θ0, events) -- src/trace.jl:20-27; events are (t, copy(x), copy(θ)) (src/not_fact_samplers.jl:39-41)."""
    F: object
    t0: float
    x0: np.ndarray
    θ0: np.ndarray
    t: np.ndarray = field(default_factory=lambda: np.empty(0))
    x: np.ndarray = field(default_factory=lambda: np.empty((0, 0)))
    θ: np.ndarray = field(default_factory=lambda: np.empty((0, 0)))

    def __len__(self):
        return 1 + len(self.t)
