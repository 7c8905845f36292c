"""ctypes binding of libpdmp_mi355.so (C ABI: include/pdmp_mi355.h).

Loading never falls back to a CPU implementation: if the shared object is missing or no gfx950 device is
visible, the calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

PDMP_OK = 0
ABI_VERSION = 3  # include/pdmp_mi355.h: PDMP_ABI_VERSION
ERR_NAMES = {0: "PDMP_OK", 1: "PDMP_ERR_INVALID", 2: "PDMP_ERR_NO_DEVICE", 3: "PDMP_ERR_HIP",
             4: "PDMP_ERR_UNSUPPORTED", 5: "PDMP_ERR_NOMEM"}

PDMP_ERR_INVALID, PDMP_ERR_NO_DEVICE, PDMP_ERR_HIP, PDMP_ERR_UNSUPPORTED, PDMP_ERR_NOMEM = 1, 2, 3, 4, 5
SAMPLER_ZIGZAG_LOCAL, SAMPLER_ZIGZAG_ALL, SAMPLER_BPS, SAMPLER_STICKY_ZIGZAG = 0, 1, 2, 3
CHAIN_OK, CHAIN_BOUND_VIOLATED, CHAIN_STALLED, CHAIN_TRACE_FULL, CHAIN_PAUSED = 0, 1, 2, 3, 4


def needs_rerun(status):
    """A launch returned before T for some chain and the next run resumes it: its trace segment is full (drain or reset it first), or its 32-bit
    launch counters are (PDMP_CHAIN_PAUSED: nothing to drain).  Every `run until T` loop of the host side asks this."""
    st = np.asarray(status)
    return bool(np.any((st == CHAIN_TRACE_FULL) | (st == CHAIN_PAUSED)))
RUN_REFERENCE_TAIL, RUN_STOP_BEFORE = 0, 1

EVENT_DTYPE = np.dtype([("t", "<f8"), ("i", "<i8"), ("x", "<f8"), ("theta", "<f8")])
COUNTERS_DTYPE = np.dtype([("t_last", "<f8"), ("num", "<u8"), ("nacc", "<u8"), ("nrefresh", "<u8"),
                           ("ntrace", "<u8"), ("nevents", "<u8"), ("ndraw_main", "<u8"),
                           ("ndraw_global", "<u8"), ("status", "<u4"), ("reserved", "<u4")])
assert EVENT_DTYPE.itemsize == 32 and COUNTERS_DTYPE.itemsize == 72

# every symbol include/pdmp_mi355.h declares (tests check that the library exports all of them)
EXPORTED_SYMBOLS = [
    "pdmp_last_error", "pdmp_abi_version", "pdmp_device_count", "pdmp_ensemble_create",
    "pdmp_ensemble_destroy", "pdmp_ensemble_set_flow_zigzag", "pdmp_ensemble_set_target_gaussian_csc",
    "pdmp_ensemble_set_state", "pdmp_ensemble_set_state_synthetic", "pdmp_ensemble_run", "pdmp_ensemble_run_partitioned", "pdmp_ensemble_sync",
    "pdmp_ensemble_last_run_ms", "pdmp_ensemble_counters", "pdmp_ensemble_totals", "pdmp_ensemble_trace_copy",
    "pdmp_ensemble_trace_reset", "pdmp_ensemble_final_state", "pdmp_ensemble_batch_means",
    "pdmp_ensemble_trace_dev", "pdmp_ensemble_counters_dev", 
    "pdmp_ensemble_set_flow_bps", "pdmp_ensemble_set_state_bps", "pdmp_ensemble_bps_trace_copy",
    "pdmp_ensemble_bps_final_state", "pdmp_ensemble_set_sticky", "pdmp_ensemble_set_adaptscale", "pdmp_ensemble_final_sigma", "pdmp_ensemble_set_flow_boomerang", "pdmp_ensemble_set_local_bound", "pdmp_ensemble_set_target_logistic", "pdmp_ensemble_set_flow_factboomerang",
    "pdmp_ensemble_set_mass_cholesky", "pdmp_ensemble_set_bps_options",
    "pdmp_ensemble_ess_begin", "pdmp_ensemble_ess_batch", "pdmp_ensemble_ess_end", "pdmp_ensemble_set_gradient_tracking",
    "pdmp_ensemble_path_integrals", "pdmp_ensemble_set_path_integrals", "pdmp_ensemble_set_neighbourhood", "pdmp_ensemble_info",
    "pdmp_ensemble_consume_begin", "pdmp_ensemble_consume", "pdmp_ensemble_consume_async", "pdmp_ensemble_last_consume_ms", "pdmp_ensemble_consume_mean", "pdmp_ensemble_consume_inclusion", "pdmp_ensemble_consume_discretized", "pdmp_ensemble_consume_cummean", "pdmp_ensemble_consume_cummean_copy", "pdmp_ensemble_subtrace_copy", "pdmp_1d_run",
    "pdmp_comm_unique_id", "pdmp_comm_init", "pdmp_comm_destroy", "pdmp_comm_info", "pdmp_comm_barrier", "pdmp_comm_allreduce",
    "pdmp_ensemble_gather_traces", "pdmp_ensemble_reduce_moments", "pdmp_comm_gathered_copy",
    "pdmp_ensemble_gather_bps_traces", "pdmp_comm_gathered_bps_copy", "pdmp_ensemble_bps_trace_dev",
]
# include/pdmp_debug.h: diagnostics, not part of the drop-in boundary
DEBUG_SYMBOLS = ["pdmp_debug_set_kernel", "pdmp_debug_set_spec_g2", "pdmp_debug_set_phase_profile", "pdmp_debug_phase_profile",
                 "pdmp_debug_set_proposal_dump", "pdmp_debug_set_track_groups", "pdmp_debug_set_helper_wave", "pdmp_debug_set_track_lines", "pdmp_debug_buffer_addresses", "pdmp_debug_placement", "pdmp_debug_set_placement", "pdmp_debug_move_buffer", "pdmp_debug_set_helper_steering", "pdmp_debug_set_launch_count_limit", "pdmp_debug_host_drain_probe", "pdmp_debug_set_consumer_overlap", "pdmp_debug_last_kernel", "pdmp_debug_set_logistic_rows", "pdmp_debug_math_probe", "pdmp_debug_write_probe", "pdmp_debug_sector_probe"]
DEBUG_KERNELS = {"auto": 0, "seq": 1, "spec4": 2, "spec8": 3, "exactp": 4}


class PdmpConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("sampler", C.c_int32), ("adapt", C.c_int32),
                ("factor", C.c_double), ("nchains", C.c_int64), ("d", C.c_int64), ("trace_capacity", C.c_int64)]


class PdmpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


_lib = None
_loaded_path = None


def loaded_path():
    """Path of the library this module instance actually loaded (None before load())."""
    return _loaded_path


def lib_path():
    """The in-tree engine library; PDMP_MI355_LIB selects another build of it (an experimental variant made by build.py --variant)."""
    return os.environ.get("PDMP_MI355_LIB") or _build.LIB_PATH


def load():
    """Load libpdmp_mi355.so; raise if it has not been built (no fallback)."""
    global _lib, _loaded_path
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} is missing: run `python zigzagboomerang.jl_amd/build.py` (hipcc, gfx950). "
            "There is no CPU fallback for the engine.")
    L = C.CDLL(path)
    vp, i64, f64 = C.c_void_p, C.c_int64, C.c_double
    L.pdmp_last_error.restype = C.c_char_p
    L.pdmp_abi_version.restype = C.c_int
    L.pdmp_device_count.restype = C.c_int
    if L.pdmp_abi_version() != ABI_VERSION:  # (PDMP_MI355_LIB can point at any build)
        raise RuntimeError(f"{path} implements ABI version {L.pdmp_abi_version()}, this binding is written against {ABI_VERSION} "
                           "(include/pdmp_mi355.h: PDMP_ABI_VERSION)")
    L.pdmp_ensemble_create.argtypes = [C.POINTER(PdmpConfig), C.POINTER(vp)]
    L.pdmp_ensemble_destroy.argtypes = [vp]
    L.pdmp_ensemble_destroy.restype = None
    L.pdmp_ensemble_set_flow_zigzag.argtypes = [vp, vp, vp, vp, vp, vp, f64, f64]
    L.pdmp_ensemble_set_flow_factboomerang.argtypes = [vp, vp, vp, vp, vp, vp, f64, f64]
    L.pdmp_ensemble_set_target_gaussian_csc.argtypes = [vp, vp, vp, vp, vp]
    L.pdmp_ensemble_set_state.argtypes = [vp, f64, vp, vp, vp, vp]
    L.pdmp_ensemble_set_state_synthetic.argtypes = [vp, f64, vp, C.c_uint64]
    L.pdmp_ensemble_run.argtypes = [vp, f64, C.c_int, vp]
    L.pdmp_ensemble_run_partitioned.argtypes = [vp, f64, C.c_int, f64, vp, i64, vp]
    L.pdmp_ensemble_sync.argtypes = [vp]
    L.pdmp_ensemble_last_run_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.pdmp_ensemble_counters.argtypes = [vp, vp]
    L.pdmp_ensemble_totals.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.pdmp_ensemble_trace_copy.argtypes = [vp, i64, i64, i64, vp]
    L.pdmp_ensemble_trace_reset.argtypes = [vp]
    L.pdmp_ensemble_final_state.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp]
    L.pdmp_ensemble_batch_means.argtypes = [vp, f64, f64, vp, vp]
    L.pdmp_ensemble_path_integrals.argtypes = [vp, f64, i64, vp, vp]
    L.pdmp_ensemble_set_path_integrals.argtypes = [vp, C.c_int]
    L.pdmp_ensemble_set_neighbourhood.argtypes = [vp, vp, vp]
    L.pdmp_ensemble_info.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(C.c_int)]
    L.pdmp_ensemble_consume_begin.argtypes = [vp, f64, i64]
    L.pdmp_ensemble_consume.argtypes = [vp]
    L.pdmp_ensemble_consume_async.argtypes = [vp, vp]
    L.pdmp_ensemble_last_consume_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.pdmp_debug_set_consumer_overlap.argtypes = [vp, C.c_int]
    L.pdmp_debug_host_drain_probe.argtypes = [vp, i64, C.POINTER(C.c_double)]
    L.pdmp_ensemble_consume_mean.argtypes = [vp, i64, i64, vp, vp]
    L.pdmp_ensemble_consume_inclusion.argtypes = [vp, i64, i64, vp, vp]
    L.pdmp_ensemble_consume_cummean.argtypes = [vp, C.c_int]
    L.pdmp_ensemble_consume_cummean_copy.argtypes = [vp, i64, i64, i64, vp, vp]
    L.pdmp_ensemble_subtrace_copy.argtypes = [vp, i64, vp, i64, vp, i64, vp]
    L.pdmp_ensemble_consume_discretized.argtypes = [vp, i64, i64, i64, vp, C.POINTER(i64), C.POINTER(vp)]
    L.pdmp_1d_run.argtypes = [C.POINTER(Config1d), vp, vp, C.c_double, vp, vp]
    L.pdmp_1d_run.restype = C.c_int
    L.pdmp_comm_unique_id.argtypes = [vp, i64]
    L.pdmp_comm_init.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.pdmp_comm_destroy.argtypes = [vp]
    L.pdmp_comm_destroy.restype = None
    L.pdmp_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.pdmp_comm_barrier.argtypes = [vp]
    L.pdmp_comm_allreduce.argtypes = [vp, vp, i64, C.c_int]
    L.pdmp_ensemble_gather_traces.argtypes = [vp, vp, C.c_int, vp, vp, i64, vp, i64, C.POINTER(vp), C.POINTER(i64)]
    L.pdmp_ensemble_reduce_moments.argtypes = [vp, vp, C.c_int, f64, f64, vp, vp]
    L.pdmp_comm_gathered_copy.argtypes = [vp, vp, i64, i64]
    L.pdmp_ensemble_gather_bps_traces.argtypes = [vp, vp, C.c_int, vp, vp, i64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i64)]
    L.pdmp_comm_gathered_bps_copy.argtypes = [vp, vp, vp, vp, i64, i64]
    L.pdmp_ensemble_bps_trace_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.pdmp_ensemble_trace_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
    L.pdmp_ensemble_counters_dev.argtypes = [vp, C.POINTER(vp)]
    L.pdmp_debug_math_probe.argtypes = [C.c_int, C.c_uint64, i64, vp]
    L.pdmp_debug_write_probe.argtypes = [C.c_int, i64, i64, i64, C.c_int, C.POINTER(C.c_double)]
    L.pdmp_debug_sector_probe.argtypes = [C.c_int, i64, i64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.pdmp_ensemble_set_target_logistic.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, f64, i64]
    L.pdmp_ensemble_set_sticky.argtypes = [vp, vp, C.c_int, C.c_int]
    L.pdmp_ensemble_set_adaptscale.argtypes = [vp, C.c_int]
    L.pdmp_ensemble_set_local_bound.argtypes = [vp, C.c_int]
    L.pdmp_ensemble_set_gradient_tracking.argtypes = [vp, C.c_int]
    L.pdmp_ensemble_final_sigma.argtypes = [vp, i64, i64, vp]
    L.pdmp_ensemble_set_flow_bps.argtypes = [vp, vp, vp, vp, vp, f64, f64]
    L.pdmp_ensemble_set_flow_boomerang.argtypes = [vp, vp, vp, vp, vp, vp, f64, f64]
    L.pdmp_ensemble_set_state_bps.argtypes = [vp, f64, vp, vp, f64, vp]
    L.pdmp_ensemble_ess_begin.argtypes = [vp, f64]
    L.pdmp_ensemble_ess_batch.argtypes = [vp, f64]
    L.pdmp_ensemble_ess_end.argtypes = [vp, vp, vp, vp, vp, C.POINTER(i64), C.POINTER(f64), C.POINTER(f64)]
    L.pdmp_ensemble_set_mass_cholesky.argtypes = [vp, vp, vp, vp]
    L.pdmp_ensemble_set_bps_options.argtypes = [vp, C.c_int, C.c_int]
    L.pdmp_ensemble_bps_trace_copy.argtypes = [vp, i64, i64, i64, vp, vp, vp]
    L.pdmp_ensemble_bps_final_state.argtypes = [vp, i64, i64, vp, vp, vp, vp]
    L.pdmp_debug_set_kernel.argtypes = [vp, C.c_int]
    L.pdmp_debug_set_spec_g2.argtypes = [vp, C.c_int]
    L.pdmp_debug_set_phase_profile.argtypes = [vp, C.c_int]
    L.pdmp_debug_phase_profile.argtypes = [vp, vp, C.POINTER(C.c_int)]
    L.pdmp_debug_set_proposal_dump.argtypes = [vp, i64]
    L.pdmp_debug_set_track_groups.argtypes = [vp, C.c_int]
    L.pdmp_debug_set_helper_wave.argtypes = [vp, C.c_int]
    L.pdmp_debug_set_track_lines.argtypes = [vp, C.c_int]
    L.pdmp_debug_buffer_addresses.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.pdmp_debug_move_buffer.argtypes = [vp, C.c_int]
    L.pdmp_debug_placement.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.pdmp_debug_set_placement.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p]
    L.pdmp_debug_set_launch_count_limit.argtypes = [vp, C.c_uint32]
    L.pdmp_debug_set_helper_steering.argtypes = [vp, C.c_double, C.c_int, C.c_double]
    L.pdmp_debug_last_kernel.argtypes = [vp, C.c_char_p, i64]
    L.pdmp_debug_set_logistic_rows.argtypes = [vp, C.c_int]
    for name in EXPORTED_SYMBOLS + DEBUG_SYMBOLS:
        fn = getattr(L, name)
        if name not in ("pdmp_last_error", "pdmp_abi_version", "pdmp_device_count", "pdmp_ensemble_destroy", "pdmp_comm_destroy"):
            fn.restype = C.c_int
    _lib = L
    _loaded_path = path
    return L


class Config1d(C.Structure):  # pdmp_1d_config
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("flow", C.c_int32), ("adapt", C.c_int32), ("factor", C.c_double),
                ("nchains", C.c_int64), ("trace_capacity", C.c_int64), ("mu", C.c_double), ("sigma2", C.c_double), ("noise", C.c_double),
                ("b_sigma", C.c_double), ("b_mu", C.c_double), ("b_lambda", C.c_double)]


EVENT1D_DTYPE = np.dtype([("t", "<f8"), ("x", "<f8"), ("theta", "<f8")])
STATE1D_DTYPE = np.dtype([("t", "<f8"), ("x", "<f8"), ("theta", "<f8"), ("c", "<f8"), ("a", "<f8"), ("b", "<f8"), ("t_next", "<f8"),
                          ("t_ref", "<f8"), ("ndraw", "<u8"), ("num", "<i8"), ("acc", "<i8"), ("started", "<i4"), ("status", "<i4")])


def check(code):
    if code != PDMP_OK:
        raise PdmpError(code, load().pdmp_last_error().decode("utf-8", "replace"))


def write_probe(nchains, d, nrec, iters=3, device=0):
    """ms per launch of the write-only kernel with the BPS event-record store pattern (HBM write ceiling)."""
    ms = C.c_double()
    check(load().pdmp_debug_write_probe(int(device), int(nchains), int(d), int(nrec), int(iters), C.byref(ms)))
    return ms.value


def sector_probe(nchains, d, rounds, write, iters=3, device=0):
    """ms per launch of the random 32-byte-sector read (write != 0: read + write back) kernel: the scattered-traffic ceiling."""
    ms = C.c_double()
    check(load().pdmp_debug_sector_probe(int(device), int(nchains), int(d), int(rounds), int(write), int(iters), C.byref(ms)))
    return ms.value


def math_probe(seed, n, device=0):
    out = np.empty((8, n))
    check(load().pdmp_debug_math_probe(int(device), int(seed), int(n), out.ctypes.data))
    return out


def device_count():
    return load().pdmp_device_count()
