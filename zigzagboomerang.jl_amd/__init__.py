"""zigzagboomerang.jl_amd -- MI355X-native engine for the PDMP event-loop hot path of ZigZagBoomerang.jl.

Layout: csrc/ (HIP kernels + C ABI, built into lib/libpdmp_mi355.so), _lib.py (ctypes), engine.py (handle
wrapper), samplers.py (spdmp/... mirror of the reference's call shape), trace.py (Trace consumers),
problems.py (inputs of the reference's scripts).  The directory name contains a dot, so it is loaded
through `__graft_entry__.load_package()` under the module name `zigzagboomerang_jl_amd`.
"""
from . import _lib, benchlib, build, ess, parallel, problems, trace  # noqa: F401
from .engine import Ensemble  # noqa: F401
from .samplers import Partition, parallel_spdmp, pdmp, spdmp, sspdmp  # noqa: F401
from .flows import (Boomerang, Boomerang1d, BouncyParticle, LocalBound, FactBoomerang, FactTrace, GaussianTarget, GaussianTarget1d,  # noqa: F401
                    LogisticTarget, PDMPTrace, ZigZag, ZigZag1d)
