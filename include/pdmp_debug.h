/*
 * pdmp_debug.h -- diagnostics and measurement hooks of libpdmp_mi355.so.  NOT part of the drop-in boundary (pdmp_mi355.h): nothing a
 * caller of spdmp / pdmp / sspdmp needs lives here.  Used by tests/ (kernel selection for parity runs, the numerical-contract probe)
 * and tools/ (phase profiles, memory-system probes).  All state is per ensemble; the library keeps no globals and reads no
 * environment variables.
 */
#ifndef PDMP_DEBUG_H
#define PDMP_DEBUG_H

#include <stddef.h>

#include "pdmp_mi355.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Which event-loop kernel a local-ZigZag / sticky ensemble uses.  AUTO: the widest speculative kernel the neighbourhood geometry
 * admits (8 events per iteration on lattice-like graphs, else 4, else 1); SEQ: one event per iteration; SPEC4: the 4-event kernel
 * where the 8-event one would run.  All of them produce the same event sequence bit for bit (tests/test_gpu_spec8_parity.py).
 * Must be called before set_flow_*. */
#define PDMP_DEBUG_KERNEL_AUTO 0
#define PDMP_DEBUG_KERNEL_SEQ 1
#define PDMP_DEBUG_KERNEL_SPEC4 2
#define PDMP_DEBUG_KERNEL_SPEC8 3 /* the 8-event kernel (8-lane groups): what AUTO runs for the moving evaluation on a lattice */
#define PDMP_DEBUG_KERNEL_EXACTP 4 /* the one-proposal-per-lane kernel of the moving evaluation (bit-identical; slower) */
pdmp_status pdmp_debug_set_kernel(pdmp_ensemble* ens, int kernel);
/* 4-event kernel: fetch the G2 records of every proposal speculatively instead of on accept only */
pdmp_status pdmp_debug_set_spec_g2(pdmp_ensemble* ens, int on);
/* Record the per-phase cycle counts of chain 0 during the following runs (a profiling instantiation of the same loop);
 * pdmp_debug_phase_profile returns the last run's 16 numbers: kind 1 (speculative kernels) [0..8] cycles per phase, [10] iterations;
 * kind 2 (general kernel) [0..6] select, move G1, gradient, coin + G2, re-bound, re-queue, tail, [10] proposals. */
pdmp_status pdmp_debug_set_phase_profile(pdmp_ensemble* ens, int on);
pdmp_status pdmp_debug_phase_profile(pdmp_ensemble* ens, double* out16, int* kind);
/* gradient tracking: which kernel runs where the one-proposal-per-lane kernel applies (plain lattice graph, no adaptation, 2048 <= d <= 16384).
 * 0 = default (zz_local_trackp_kernel: (key, t_old) pairs in blocks of 8, one dirty line per rejected proposal), 1 = the 8-lane-group kernel
 * (zz_local_track_kernel, which also serves adaptation, a target mean, a looser bounding Γ and lattice-like graphs) -- both compute the same
 * floats and commit the same sequence.  The choice fixes the queue's layout: call it BEFORE set_state.  (Rounds 1-2 carried two more
 * variants, key blocks of 32 and of 16 in the record layout; they were superseded and removed.) */
pdmp_status pdmp_debug_set_track_groups(pdmp_ensemble* ens, int which);
/* zz_local_trackp_kernel / zz_local_trackp2_kernel: the two-wave form gives every chain a helper wavefront (the chain's uniforms and their
 * logarithms produced ahead into a ring in LDS, the next windows' lines requested early); same committed sequence and floats.  -1 = chosen by
 * the ensemble's width (at most seven chains per compute unit, 1792 on an MI355X), 0 = never, 1 = always.  Before the next run. */
pdmp_status pdmp_debug_set_helper_wave(pdmp_ensemble* ens, int mode);
/* zz_local_trackl_kernel (round 6): the same event loop on the LINE layout -- a coordinate pair's (key, t_old) pairs, sums and constants in one
 * 128-byte line, nine-bit wheel images of the pair minima in LDS -- which is what ensembles of more than 12 chains per compute unit run on the
 * plain lattice with an even side (d <= 16384); same committed sequence and floats.  -1 = by the ensemble's width, 0 = never, 1 = wherever the
 * layout serves.  The choice fixes the state's layout: call it BEFORE set_state.  With this kernel pdmp_debug_set_helper_steering's first two
 * arguments are the block minima per quantum of the wheel and the events a window aims at (0: the defaults). */
pdmp_status pdmp_debug_set_track_lines(pdmp_ensemble* ens, int mode);
/* Device addresses of the ensemble's large arrays (tracked records, (key, t_old) pairs, trace slots, chain headers, canonical records, keys,
 * per-coordinate constants, tables): tools/mode_alloc.py relates the full-width slice's timing mode to where they landed. */
pdmp_status pdmp_debug_buffer_addresses(pdmp_ensemble* ens, uint64_t* out8);
/* How the arrays of several GB were laid over the device's three memory classes (csrc/pdmp_place.hip), as text: per array the class of every 1 GB
 * chunk in address order, the chunks created while looking for the classes and the seconds that took -- or "hipMalloc" where the array was not placed. */
pdmp_status pdmp_debug_placement(pdmp_ensemble* ens, char* buf, size_t nbuf);
/* Where the arrays lie.  tune: 1 (the default) = set_state times a short launch of a device-filling ZigZag ensemble, re-allocates the pairs / keys and the
 * records with hipMalloc and keeps the fastest combination (the "timing mode" of a full-width launch belongs to those allocations: DESIGN.md 5); 0 = take
 * what hipMalloc gives; -1 = leave as is.  place: 1 = EXPERIMENTAL chunk-wise placement over the device's three memory classes (csrc/pdmp_place.hip -- its
 * header says why it is off), with optional class patterns ("012", "0", ...) for the records, the pairs and the trace; 0 = off.  Before set_state. */
pdmp_status pdmp_debug_set_placement(pdmp_ensemble* ens, int tune, int place, const char* rec, const char* kp, const char* ev);
/* Copies one array (0 records, 1 pairs, 2 trace, 3 headers, 4 constants, 5 keys) into newly allocated memory and continues on the copy; the old
 * allocation stays reserved until the process ends (so that the copy lands on other pages).  For tools/mode_alloc.py only. */
pdmp_status pdmp_debug_move_buffer(pdmp_ensemble* ens, int which);
/* ... its tuning (none of it changes a result): the selection threshold moves by `gain` of the way towards `target` raw candidates per iteration;
 * the helper requests the lines of the blocks within `ahead` window lengths beyond the current window */
pdmp_status pdmp_debug_set_helper_steering(pdmp_ensemble* ens, double gain, int target, double ahead);
/* PDMP_CHAIN_PAUSED under test: a chain pauses once ONE launch has used n draws of its main stream (the subsampled-logistic kernel: n proposals)
 * instead of 3 * 2^30; 0 restores the default */
pdmp_status pdmp_debug_set_launch_count_limit(pdmp_ensemble* ens, uint32_t n);
/* pdmp_ensemble_consume_async: -1 = the consumer runs beside the next slice where the ensemble leaves SIMDs idle (at most 2048 chains) and between
 * the slices where it fills the device (measured: beside a full-width C3 launch it costs the launch more than its own time), 0 / 1 = always between / beside */
pdmp_status pdmp_debug_set_consumer_overlap(pdmp_ensemble* ens, int mode);
/* what the host could drain instead of consuming on the device: `bytes` of the ensemble's trace buffer copied to pinned host memory, GB/s */
pdmp_status pdmp_debug_host_drain_probe(pdmp_ensemble* ens, int64_t bytes, double* gbps);
/* name of the event-loop kernel the last pdmp_ensemble_run launched (bench.py prints it with every line: no figure without its kernel) */
pdmp_status pdmp_debug_last_kernel(pdmp_ensemble* ens, char* out, int64_t cap);
/* chains per wavefront of the LDS-resident logistic kernel (config C4): -1 the library's default, 0 one chain (pdmp_logistic.hip), 16 or 32 =
 * rows of that many lanes, 4 or 2 chains per wavefront (pdmp_logrows.hip); an ensemble that does not fit the rows asked for is refused at run */
pdmp_status pdmp_debug_set_logistic_rows(pdmp_ensemble* ens, int row_width);
/* one-event kernel: print the first n proposals of chain 0 to stderr during the next run */
pdmp_status pdmp_debug_set_proposal_dump(pdmp_ensemble* ens, int64_t n);

/*
 * Test hook: evaluate the shared numerical contract (include/pdmp_detmath.h) on the device for draws
 * k = 0..n-1 of `seed`; out is [8 x n] row-major: u01, pdmp_log(u), a/b, sqrt, poisson_time, pdmp_randn, pdmp_exp, sin+2cos of pdmp_sincos.
 * A host evaluation of the same expressions must agree bit-for-bit (tests/test_gpu_detmath.py).
 */
pdmp_status pdmp_debug_math_probe(int device, uint64_t seed, int64_t n, double* out);

/*
 * Measurement hook: time (ms per launch, HIP events) of a write-only kernel with the event-record store pattern of the bouncy
 * particle kernel -- one wavefront per chain writing `nrec` records of x[d] and θ[d] -- i.e. the HBM write ceiling that the C2
 * roofline fraction is read against (tools/bench_c2.py).
 */
pdmp_status pdmp_debug_write_probe(int device, int64_t nchains, int64_t d, int64_t nrec, int iters, double* ms_out);

/* Test/measurement hook: time per launch (ms) of a kernel that does nothing but the scattered record traffic of the local ZigZag
 * event loop -- one wavefront per chain, every lane reads the 32-byte first half of 4 pseudo-random 64-byte records of its chain
 * per round, `rounds` times, and with write != 0 stores them back changed.  nchains * d * 64 bytes are allocated for it.  The
 * sector rate it reaches is the practical ceiling quoted beside the event loop's own. */
pdmp_status pdmp_debug_sector_probe(int device, int64_t nchains, int64_t d, int rounds, int write, int iters,
                                             double* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* PDMP_DEBUG_H */
