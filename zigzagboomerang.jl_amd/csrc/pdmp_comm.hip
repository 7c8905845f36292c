// pdmp_comm.hip -- the post-run exchange of a sharded ensemble over RCCL, behind the C ABI (include/pdmp_mi355.h: pdmp_comm_*,
// pdmp_ensemble_gather_traces, pdmp_ensemble_reduce_moments).  SURVEY.md 8(e1).
//
// Chains are independent: rank r of R owns a contiguous block of them and runs it with NO collective.  Afterwards
//   * ncclAllGather of the per-rank event counts (padded to the widest shard);
//   * every rank compacts its trace segments (one [capacity] slab per chain in the engine's buffer) into one contiguous device array, and the
//     peers stream theirs to the root inside ONE ncclGroupStart / ncclGroupEnd -- a gatherv in which each peer uses its own direct xGMI link
//     to the root (xGMI is point-to-point: a ring collective would be per-link bound and world-1 times slower);
//   * ncclReduce(sum) of the batch-mean accumulators (2 d doubles).
// This file uses the engine through its PUBLIC entry points only (device pointers from pdmp_ensemble_trace_dev): it is a second translation
// unit of the same library, linked against librccl.  One communicator per (process, device); calls on it must be serialised by the caller.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pdmp_mi355.h"

extern "C" void pdmp_set_last_error_(const char* msg);  // pdmp_capi.hip: the thread-local string pdmp_last_error() returns

namespace {

pdmp_status cfail(pdmp_status st, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    pdmp_set_last_error_(buf);
    return st;
}

#define C_HIP(expr)                                                                                        \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return cfail(PDMP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));   \
    } while (0)
#define C_NCCL(expr)                                                                                       \
    do {                                                                                                   \
        ncclResult_t r_ = (expr);                                                                          \
        if (r_ != ncclSuccess) return cfail(PDMP_ERR_HIP, "%s failed: %s", #expr, ncclGetErrorString(r_)); \
    } while (0)

struct DBuf {
    void* p = nullptr;
    size_t bytes = 0;
    pdmp_status need(size_t n) {
        if (n <= bytes) return PDMP_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        if (hipMalloc(&p, n) != hipSuccess) {
            p = nullptr;
            return cfail(PDMP_ERR_NOMEM, "hipMalloc(%zu bytes) failed", n);
        }
        bytes = n;
        return PDMP_OK;
    }
    ~DBuf() {
        if (p) (void)hipFree(p);
    }
};

// the segments [chain * cap, chain * cap + count[chain]) of the engine's trace buffer, back to back (offs = exclusive prefix sums)
__global__ __launch_bounds__(256) void compact_traces_kernel(const uint4* __restrict__ ev, int64_t cap, const uint64_t* __restrict__ cnt,
                                                             const uint64_t* __restrict__ offs, uint4* __restrict__ out) {
    const int64_t chain = blockIdx.x;
    const uint64_t n2 = cnt[chain] * 2;  // 16-byte halves of the 32-byte events
    const uint4* src = ev + chain * cap * 2;
    uint4* dst = out + offs[chain] * 2;
    for (uint64_t k = (uint64_t)blockIdx.y * 256 + threadIdx.x; k < n2; k += (uint64_t)gridDim.y * 256) dst[k] = src[k];
}

}  // namespace

struct pdmp_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    DBuf scratch, counts, offs, compact, gathered, red, vote, gx, gth;
    int64_t ngathered = 0;      // events in `gathered` (root of the last pdmp_ensemble_gather_traces)
    int64_t ngathered_bps = 0;  // PDMPTrace events in gathered (t) / gx / gth (root of the last pdmp_ensemble_gather_bps_traces)
    int64_t bps_d = 0;
};

extern "C" {

pdmp_status pdmp_comm_unique_id(void* id, int64_t id_bytes) {
    if (!id || id_bytes < (int64_t)PDMP_COMM_ID_BYTES) return cfail(PDMP_ERR_INVALID, "the id buffer must hold PDMP_COMM_ID_BYTES = %d bytes", PDMP_COMM_ID_BYTES);
    static_assert(sizeof(ncclUniqueId) == PDMP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    C_NCCL(ncclGetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return PDMP_OK;
}

pdmp_status pdmp_comm_init(const void* id, int rank, int world, int device, pdmp_comm** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return cfail(PDMP_ERR_INVALID, "bad argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return cfail(PDMP_ERR_NO_DEVICE, "no HIP device visible: RCCL needs one per rank");
    if (device < 0 || device >= ndev) return cfail(PDMP_ERR_INVALID, "device %d out of range", device);
    C_HIP(hipSetDevice(device));
    pdmp_comm* c = new pdmp_comm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return cfail(PDMP_ERR_HIP, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, device, ncclGetErrorString(r));
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        (void)ncclCommDestroy(c->comm);
        delete c;
        return cfail(PDMP_ERR_HIP, "stream creation failed");
    }
    *out = c;
    return PDMP_OK;
}

void pdmp_comm_destroy(pdmp_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

pdmp_status pdmp_comm_info(const pdmp_comm* c, int* rank, int* world) {
    if (!c) return cfail(PDMP_ERR_INVALID, "null argument");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return PDMP_OK;
}

pdmp_status pdmp_comm_allreduce(pdmp_comm* c, double* inout, int64_t n, int op) {
    if (!c || !inout || n <= 0 || (op != PDMP_COMM_SUM && op != PDMP_COMM_MAX)) return cfail(PDMP_ERR_INVALID, "bad argument");
    C_HIP(hipSetDevice(c->device));
    pdmp_status st = c->scratch.need((size_t)n * sizeof(double));
    if (st != PDMP_OK) return st;
    C_HIP(hipMemcpyAsync(c->scratch.p, inout, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    C_NCCL(ncclAllReduce(c->scratch.p, c->scratch.p, (size_t)n, ncclDouble, op == PDMP_COMM_SUM ? ncclSum : ncclMax, c->comm, c->stream));
    C_HIP(hipMemcpyAsync(inout, c->scratch.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    C_HIP(hipStreamSynchronize(c->stream));
    return PDMP_OK;
}

pdmp_status pdmp_comm_barrier(pdmp_comm* c) {
    double one = 1.0;
    return pdmp_comm_allreduce(c, &one, 1, PDMP_COMM_SUM);
}

}  // extern "C"

namespace {

// Agreement on whether to go on: MAX over the ranks of a local flag.  Argument checks that only ONE rank can make (its counts / events buffers
// are too small, its allocation failed) are voted on before any rank enters the send / recv phase -- a rank that returned on its own would leave
// its peers blocked in the next collective.
pdmp_status agree(pdmp_comm* c, pdmp_status local, const char* what_elsewhere = nullptr) {
    int32_t flag = (local == PDMP_OK) ? 0 : 1;
    pdmp_status st = c->vote.need(2 * sizeof(int32_t));
    if (st != PDMP_OK) return st;  // (64 bytes of device memory: if that fails nothing works)
    int32_t* dv = static_cast<int32_t*>(c->vote.p);
    C_HIP(hipMemcpyAsync(dv, &flag, sizeof flag, hipMemcpyHostToDevice, c->stream));
    C_NCCL(ncclAllReduce(dv, dv + 1, 1, ncclInt32, ncclMax, c->comm, c->stream));
    int32_t any = 0;
    C_HIP(hipMemcpyAsync(&any, dv + 1, sizeof any, hipMemcpyDeviceToHost, c->stream));
    C_HIP(hipStreamSynchronize(c->stream));
    if (local != PDMP_OK) return local;  // (this rank's own message stays in pdmp_last_error)
    if (any) return cfail(PDMP_ERR_INVALID, "%s", what_elsewhere ? what_elsewhere : "the exchange was refused on another rank (an argument check or an allocation failed there); nothing was exchanged");
    return PDMP_OK;
}

// Shard widths and the counts of every chain of the whole ensemble (two ncclAllGather, the counts padded to the widest shard)
pdmp_status exchange_counts(pdmp_comm* c, int64_t nch, std::vector<uint64_t>& mine, std::vector<int64_t>& widths, std::vector<uint64_t>& all,
                            int64_t& wmax) {
    const int W = c->world;
    pdmp_status st;
    widths.assign((size_t)W, 0);
    {
        if ((st = c->scratch.need((size_t)(W + 1) * sizeof(int64_t))) != PDMP_OK) return st;
        int64_t* dw = static_cast<int64_t*>(c->scratch.p);
        C_HIP(hipMemcpyAsync(dw + W, &nch, sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
        C_NCCL(ncclAllGather(dw + W, dw, 1, ncclInt64, c->comm, c->stream));
        C_HIP(hipMemcpyAsync(widths.data(), dw, (size_t)W * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
        C_HIP(hipStreamSynchronize(c->stream));
    }
    wmax = 0;
    for (int r = 0; r < W; ++r) wmax = widths[(size_t)r] > wmax ? widths[(size_t)r] : wmax;
    mine.resize((size_t)wmax, 0);
    all.assign((size_t)wmax * (size_t)W, 0);
    if ((st = c->counts.need((size_t)wmax * (size_t)(W + 1) * sizeof(uint64_t))) != PDMP_OK) return st;
    uint64_t* dc = static_cast<uint64_t*>(c->counts.p);
    C_HIP(hipMemcpyAsync(dc + (size_t)wmax * W, mine.data(), (size_t)wmax * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    C_NCCL(ncclAllGather(dc + (size_t)wmax * W, dc, (size_t)wmax, ncclUint64, c->comm, c->stream));
    C_HIP(hipMemcpyAsync(all.data(), dc, all.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    C_HIP(hipStreamSynchronize(c->stream));
    return PDMP_OK;
}

// gatherv to the root: ONE grouped send / recv (every peer over its own link).  The group is always closed and the stream drained, whatever a call
// inside it returned: an open group would swallow the communicator's next collective.
pdmp_status gatherv_bytes(pdmp_comm* c, int root, const void* src, const std::vector<uint64_t>& bytes_by_rank, char* dst) {
    const int W = c->world;
    ncclResult_t first = ncclSuccess;
    auto keep = [&](ncclResult_t r) {
        if (first == ncclSuccess && r != ncclSuccess) first = r;
    };
    const uint64_t my_bytes = bytes_by_rank[(size_t)c->rank];
    keep(ncclGroupStart());
    if (c->rank == root) {
        uint64_t at = 0;
        for (int r = 0; r < W; ++r) {
            if (r != root && bytes_by_rank[(size_t)r]) keep(ncclRecv(dst + at, (size_t)bytes_by_rank[(size_t)r], ncclChar, r, c->comm, c->stream));
            at += bytes_by_rank[(size_t)r];
        }
    } else if (my_bytes) {
        keep(ncclSend(src, (size_t)my_bytes, ncclChar, root, c->comm, c->stream));
    }
    keep(ncclGroupEnd());
    hipError_t he = hipSuccess;
    if (c->rank == root && my_bytes) {
        uint64_t at = 0;
        for (int r = 0; r < root; ++r) at += bytes_by_rank[(size_t)r];
        he = hipMemcpyAsync(dst + at, src, (size_t)my_bytes, hipMemcpyDeviceToDevice, c->stream);
    }
    const hipError_t hs = hipStreamSynchronize(c->stream);
    if (first != ncclSuccess) return cfail(PDMP_ERR_HIP, "grouped send / recv failed: %s", ncclGetErrorString(first));
    if (he != hipSuccess) return cfail(PDMP_ERR_HIP, "device copy of the root's own segment failed: %s", hipGetErrorString(he));
    if (hs != hipSuccess) return cfail(PDMP_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(hs));
    return PDMP_OK;
}

// [chain * seg_words, chain * seg_words + cnt[chain] * wpe) of a strided array of 8-byte words, back to back (offs = exclusive prefix sums of cnt)
__global__ __launch_bounds__(256) void compact_words_kernel(const uint64_t* __restrict__ src0, int64_t seg_words, const uint64_t* __restrict__ cnt,
                                                            const uint64_t* __restrict__ offs, int64_t wpe, uint64_t* __restrict__ out) {
    const int64_t chain = blockIdx.x;
    const uint64_t n = cnt[chain] * (uint64_t)wpe;
    const uint64_t* src = src0 + chain * seg_words;
    uint64_t* dst = out + offs[chain] * (uint64_t)wpe;
    for (uint64_t k = (uint64_t)blockIdx.y * 256 + threadIdx.x; k < n; k += (uint64_t)gridDim.y * 256) dst[k] = src[k];
}

}  // namespace

extern "C" {

pdmp_status pdmp_ensemble_gather_traces(pdmp_ensemble* ens, pdmp_comm* c, int root, int64_t* nchains_by_rank, uint64_t* counts,
                                        int64_t counts_cap, pdmp_event* events_host, int64_t events_cap, void** events_dev,
                                        int64_t* nevents_total) {
    if (!ens || !c || root < 0 || root >= c->world) return cfail(PDMP_ERR_INVALID, "bad argument");
    int64_t nch = 0, d = 0, cap = 0;
    int dev = 0;
    pdmp_status st = pdmp_ensemble_info(ens, &nch, &d, &cap, &dev);
    if (st != PDMP_OK) return st;
    if (dev != c->device) return cfail(PDMP_ERR_INVALID, "the ensemble lives on device %d, the communicator on %d", dev, c->device);
    if (cap <= 0) return cfail(PDMP_ERR_INVALID, "ensemble was created with trace_capacity = 0");
    C_HIP(hipSetDevice(c->device));
    const int W = c->world;
    c->ngathered = 0;  // (whatever happens below, the previous gather's events are no longer what gathered_copy should serve)
    // From here on every rank takes part in every collective whatever fails on it alone: a local failure is carried in `local` (it contributes no
    // events), voted on by agree() before anything is sent, and failures after the vote are recorded and returned once the grouped exchange has
    // closed -- a rank never leaves its peers waiting in a collective
    pdmp_status local = PDMP_OK;
    std::vector<pdmp_chain_counters> cnt((size_t)nch);
    local = pdmp_ensemble_counters(ens, cnt.data());
    void* evdev = nullptr;
    int64_t capdev = 0;
    if (local == PDMP_OK) local = pdmp_ensemble_trace_dev(ens, &evdev, &capdev);
    std::vector<uint64_t> mine((size_t)nch, 0), all;
    for (int64_t k = 0; local == PDMP_OK && k < nch; ++k) mine[(size_t)k] = cnt[(size_t)k].ntrace;
    std::vector<int64_t> widths;
    int64_t wmax = 0;
    if ((st = exchange_counts(c, nch, mine, widths, all, wmax)) != PDMP_OK) return st;
    int64_t wsum = 0;
    for (int r = 0; r < W; ++r) wsum += widths[(size_t)r];
    if (nchains_by_rank) memcpy(nchains_by_rank, widths.data(), (size_t)W * sizeof(int64_t));
    std::vector<uint64_t> tot((size_t)W, 0);
    uint64_t total = 0;
    for (int r = 0; r < W; ++r) {
        for (int64_t k = 0; k < widths[(size_t)r]; ++k) tot[(size_t)r] += all[(size_t)r * (size_t)wmax + (size_t)k];
        total += tot[(size_t)r];
    }
    if (nevents_total) *nevents_total = (int64_t)total;
    // ---- everything that can fail on ONE rank, checked now (wsum and total are known everywhere) and voted on before anything is sent
    if (local == PDMP_OK && counts && counts_cap < wsum)
        local = cfail(PDMP_ERR_INVALID, "counts holds %lld entries, the ensemble has %lld chains", (long long)counts_cap, (long long)wsum);
    if (local == PDMP_OK && c->rank == root && events_host && (int64_t)total > events_cap)
        local = cfail(PDMP_ERR_INVALID, "events_host holds %lld events, %llu would be gathered", (long long)events_cap, (unsigned long long)total);
    const uint64_t my_total = tot[(size_t)c->rank];
    if (local == PDMP_OK) local = c->offs.need((size_t)(2 * nch) * sizeof(uint64_t));
    if (local == PDMP_OK) local = c->compact.need((size_t)(my_total ? my_total : 1) * sizeof(pdmp_event));
    if (local == PDMP_OK && c->rank == root) local = c->gathered.need((size_t)(total ? total : 1) * sizeof(pdmp_event));
    if ((st = agree(c, local)) != PDMP_OK) return st;
    if (counts) {
        int64_t q = 0;
        for (int r = 0; r < W; ++r)
            for (int64_t k = 0; k < widths[(size_t)r]; ++k) counts[q++] = all[(size_t)r * (size_t)wmax + (size_t)k];
    }
    // ---- compact the local segments (a failure here no longer returns before the exchange: it is reported after it)
    pdmp_status late = PDMP_OK;
    auto note = [&](hipError_t e_, const char* what) {
        if (e_ != hipSuccess && late == PDMP_OK) late = cfail(PDMP_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e_));
    };
    std::vector<uint64_t> offs((size_t)nch, 0);
    for (int64_t k = 1; k < nch; ++k) offs[(size_t)k] = offs[(size_t)k - 1] + mine[(size_t)k - 1];
    uint64_t* doffs = static_cast<uint64_t*>(c->offs.p);
    note(hipMemcpyAsync(doffs, offs.data(), (size_t)nch * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream), "hipMemcpyAsync (offsets)");
    note(hipMemcpyAsync(doffs + nch, mine.data(), (size_t)nch * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream), "hipMemcpyAsync (counts)");
    if (my_total && late == PDMP_OK)
        hipLaunchKernelGGL(compact_traces_kernel, dim3((unsigned)nch, 4), dim3(256), 0, c->stream, static_cast<const uint4*>(evdev), capdev, doffs + nch,
                           doffs, static_cast<uint4*>(c->compact.p));
    note(hipGetLastError(), "compact_traces_kernel launch");
    // ---- gatherv to the root
    pdmp_event* gdst = (c->rank == root) ? static_cast<pdmp_event*>(c->gathered.p) : nullptr;
    std::vector<uint64_t> bytes((size_t)W);
    for (int r = 0; r < W; ++r) bytes[(size_t)r] = tot[(size_t)r] * sizeof(pdmp_event);
    st = gatherv_bytes(c, root, c->compact.p, bytes, reinterpret_cast<char*>(gdst));
    // a second vote AFTER the exchange: a rank whose compaction or transfer failed has sent bytes that are not its events -- every rank, the
    // root first of all, returns an error then and nothing counts as gathered
    if ((st = agree(c, late != PDMP_OK ? late : st, "the exchange failed on another rank after the vote: what the root received is not its events")) != PDMP_OK) {
        c->ngathered = 0;
        return st;
    }
    if (c->rank == root && events_host && total) {
        C_HIP(hipMemcpyAsync(events_host, gdst, (size_t)total * sizeof(pdmp_event), hipMemcpyDeviceToHost, c->stream));
        C_HIP(hipStreamSynchronize(c->stream));
    }
    if (events_dev) *events_dev = (c->rank == root) ? (void*)gdst : nullptr;
    c->ngathered = (c->rank == root) ? (int64_t)total : 0;
    return PDMP_OK;
}

// PDMPTrace events of the non-factorised samplers (src/not_fact_samplers.jl:39-41: (t, copy(x), copy(θ)), 8 (2 d + 1) bytes each): the same
// exchange on the three arrays the engine keeps them in.  On root: t [total], x [total x d], θ [total x d], rank-major, chain-major.
pdmp_status pdmp_ensemble_gather_bps_traces(pdmp_ensemble* ens, pdmp_comm* c, int root, int64_t* nchains_by_rank, uint64_t* counts,
                                            int64_t counts_cap, void** t_dev, void** x_dev, void** theta_dev, int64_t* nevents_total) {
    if (!ens || !c || root < 0 || root >= c->world) return cfail(PDMP_ERR_INVALID, "bad argument");
    int64_t nch = 0, d = 0, cap = 0;
    int dev = 0;
    pdmp_status st = pdmp_ensemble_info(ens, &nch, &d, &cap, &dev);
    if (st != PDMP_OK) return st;
    if (dev != c->device) return cfail(PDMP_ERR_INVALID, "the ensemble lives on device %d, the communicator on %d", dev, c->device);
    if (cap <= 0) return cfail(PDMP_ERR_INVALID, "ensemble was created with trace_capacity = 0");
    void *tdev = nullptr, *xdev = nullptr, *thdev = nullptr;
    if ((st = pdmp_ensemble_bps_trace_dev(ens, &tdev, &xdev, &thdev)) != PDMP_OK) return st;
    C_HIP(hipSetDevice(c->device));
    const int W = c->world;
    c->ngathered_bps = 0;
    pdmp_status local = PDMP_OK;  // (as in pdmp_ensemble_gather_traces: carried to the vote, never returned on one rank alone past this point)
    std::vector<pdmp_chain_counters> cnt((size_t)nch);
    local = pdmp_ensemble_counters(ens, cnt.data());
    std::vector<uint64_t> mine((size_t)nch, 0), all;
    for (int64_t k = 0; local == PDMP_OK && k < nch; ++k) mine[(size_t)k] = cnt[(size_t)k].ntrace;
    std::vector<int64_t> widths;
    int64_t wmax = 0;
    if ((st = exchange_counts(c, nch, mine, widths, all, wmax)) != PDMP_OK) return st;
    int64_t wsum = 0;
    for (int r = 0; r < W; ++r) wsum += widths[(size_t)r];
    if (nchains_by_rank) memcpy(nchains_by_rank, widths.data(), (size_t)W * sizeof(int64_t));
    std::vector<uint64_t> tot((size_t)W, 0);
    uint64_t total = 0;
    for (int r = 0; r < W; ++r) {
        for (int64_t k = 0; k < widths[(size_t)r]; ++k) tot[(size_t)r] += all[(size_t)r * (size_t)wmax + (size_t)k];
        total += tot[(size_t)r];
    }
    if (nevents_total) *nevents_total = (int64_t)total;
    const uint64_t my_total = tot[(size_t)c->rank];
    if (local == PDMP_OK && counts && counts_cap < wsum)
        local = cfail(PDMP_ERR_INVALID, "counts holds %lld entries, the ensemble has %lld chains", (long long)counts_cap, (long long)wsum);
    if (local == PDMP_OK) local = c->offs.need((size_t)(2 * nch) * sizeof(uint64_t));
    if (local == PDMP_OK) local = c->compact.need((size_t)(my_total ? my_total : 1) * (size_t)d * sizeof(double));  // (one array at a time)
    if (local == PDMP_OK && c->rank == root) {
        local = c->gathered.need((size_t)(total ? total : 1) * sizeof(double));
        if (local == PDMP_OK) local = c->gx.need((size_t)(total ? total : 1) * (size_t)d * sizeof(double));
        if (local == PDMP_OK) local = c->gth.need((size_t)(total ? total : 1) * (size_t)d * sizeof(double));
    }
    if ((st = agree(c, local)) != PDMP_OK) return st;
    if (counts) {
        int64_t q = 0;
        for (int r = 0; r < W; ++r)
            for (int64_t k = 0; k < widths[(size_t)r]; ++k) counts[q++] = all[(size_t)r * (size_t)wmax + (size_t)k];
    }
    std::vector<uint64_t> offs((size_t)nch, 0);
    for (int64_t k = 1; k < nch; ++k) offs[(size_t)k] = offs[(size_t)k - 1] + mine[(size_t)k - 1];
    uint64_t* doffs = static_cast<uint64_t*>(c->offs.p);
    pdmp_status late = PDMP_OK;  // (after the vote nothing returns before the three grouped exchanges have closed)
    auto note = [&](hipError_t e_, const char* what) {
        if (e_ != hipSuccess && late == PDMP_OK) late = cfail(PDMP_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e_));
    };
    note(hipMemcpyAsync(doffs, offs.data(), (size_t)nch * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream), "hipMemcpyAsync (offsets)");
    note(hipMemcpyAsync(doffs + nch, mine.data(), (size_t)nch * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream), "hipMemcpyAsync (counts)");
    struct Piece {
        const void* src;
        int64_t wpe;
        DBuf* dst;
    } pieces[3] = {{tdev, 1, &c->gathered}, {xdev, d, &c->gx}, {thdev, d, &c->gth}};
    for (const Piece& pc : pieces) {
        if (my_total && late == PDMP_OK)
            hipLaunchKernelGGL(compact_words_kernel, dim3((unsigned)nch, 4), dim3(256), 0, c->stream, static_cast<const uint64_t*>(pc.src), cap * pc.wpe,
                               doffs + nch, doffs, pc.wpe, static_cast<uint64_t*>(c->compact.p));
        note(hipGetLastError(), "compact_words_kernel launch");
        std::vector<uint64_t> bytes((size_t)W);
        for (int r = 0; r < W; ++r) bytes[(size_t)r] = tot[(size_t)r] * (uint64_t)pc.wpe * sizeof(double);
        const pdmp_status sg = gatherv_bytes(c, root, c->compact.p, bytes, (c->rank == root) ? static_cast<char*>(pc.dst->p) : nullptr);
        if (sg != PDMP_OK && late == PDMP_OK) late = sg;
    }
    if ((st = agree(c, late, "the exchange failed on another rank after the vote: what the root received is not its events")) != PDMP_OK) {
        c->ngathered_bps = 0;
        return st;
    }
    const bool isroot = c->rank == root;
    if (t_dev) *t_dev = isroot ? c->gathered.p : nullptr;
    if (x_dev) *x_dev = isroot ? c->gx.p : nullptr;
    if (theta_dev) *theta_dev = isroot ? c->gth.p : nullptr;
    c->ngathered_bps = isroot ? (int64_t)total : 0;
    c->bps_d = d;
    return PDMP_OK;
}

pdmp_status pdmp_comm_gathered_bps_copy(pdmp_comm* c, double* t, double* x, double* theta, int64_t first, int64_t count) {
    if (!c || first < 0 || count < 0) return cfail(PDMP_ERR_INVALID, "bad argument");
    if (first + count > c->ngathered_bps) return cfail(PDMP_ERR_INVALID, "the last BPS gather left %lld events on this rank", (long long)c->ngathered_bps);
    C_HIP(hipSetDevice(c->device));
    const size_t dd = (size_t)c->bps_d;
    if (count && t) C_HIP(hipMemcpy(t, static_cast<const double*>(c->gathered.p) + first, (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
    if (count && x) C_HIP(hipMemcpy(x, static_cast<const double*>(c->gx.p) + (size_t)first * dd, (size_t)count * dd * sizeof(double), hipMemcpyDeviceToHost));
    if (count && theta) C_HIP(hipMemcpy(theta, static_cast<const double*>(c->gth.p) + (size_t)first * dd, (size_t)count * dd * sizeof(double), hipMemcpyDeviceToHost));
    return PDMP_OK;
}

pdmp_status pdmp_comm_gathered_copy(pdmp_comm* c, pdmp_event* out, int64_t first, int64_t count) {
    if (!c || !out || first < 0 || count < 0) return cfail(PDMP_ERR_INVALID, "bad argument");
    if (first + count > c->ngathered) return cfail(PDMP_ERR_INVALID, "the last gather left %lld events on this rank", (long long)c->ngathered);
    C_HIP(hipSetDevice(c->device));
    if (count) C_HIP(hipMemcpy(out, static_cast<const pdmp_event*>(c->gathered.p) + first, (size_t)count * sizeof(pdmp_event), hipMemcpyDeviceToHost));
    return PDMP_OK;
}

pdmp_status pdmp_ensemble_reduce_moments(pdmp_ensemble* ens, pdmp_comm* c, int root, double T_prev, double T, double* sum_y, double* sum_y2) {
    if (!ens || !c || root < 0 || root >= c->world) return cfail(PDMP_ERR_INVALID, "bad argument");
    int64_t nch = 0, d = 0, cap = 0;
    int dev = 0;
    pdmp_status st = pdmp_ensemble_info(ens, &nch, &d, &cap, &dev);
    if (st != PDMP_OK) return st;
    if (dev != c->device) return cfail(PDMP_ERR_INVALID, "the ensemble lives on device %d, the communicator on %d", dev, c->device);
    std::vector<double> h((size_t)(2 * d));
    C_HIP(hipSetDevice(c->device));
    // (what can fail on ONE rank -- its consumer is not armed, its interval is empty, an allocation -- is voted on before the reduction)
    pdmp_status local = pdmp_ensemble_batch_means(ens, T_prev, T, h.data(), h.data() + d);
    if (local == PDMP_OK) local = c->red.need((size_t)(2 * d) * sizeof(double));
    if ((st = agree(c, local)) != PDMP_OK) return st;
    C_HIP(hipMemcpyAsync(c->red.p, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    C_NCCL(ncclReduce(c->red.p, c->red.p, (size_t)(2 * d), ncclDouble, ncclSum, root, c->comm, c->stream));
    if (c->rank == root) C_HIP(hipMemcpyAsync(h.data(), c->red.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    C_HIP(hipStreamSynchronize(c->stream));
    if (c->rank == root) {
        if (sum_y) memcpy(sum_y, h.data(), (size_t)d * sizeof(double));
        if (sum_y2) memcpy(sum_y2, h.data() + d, (size_t)d * sizeof(double));
    }
    return PDMP_OK;
}

}  // extern "C"
