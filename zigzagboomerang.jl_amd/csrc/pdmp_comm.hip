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
    DBuf scratch, counts, offs, compact, gathered, red;
    int64_t ngathered = 0;  // events in `gathered` (root of the last gather)
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
    // ---- shard widths, then the counts padded to the widest shard (ncclAllGather wants equal pieces)
    std::vector<pdmp_chain_counters> cnt((size_t)nch);
    if ((st = pdmp_ensemble_counters(ens, cnt.data())) != PDMP_OK) return st;
    std::vector<int64_t> widths((size_t)W, 0);
    {
        if ((st = c->scratch.need((size_t)(W + 1) * sizeof(int64_t))) != PDMP_OK) return st;
        int64_t* dw = static_cast<int64_t*>(c->scratch.p);
        C_HIP(hipMemcpyAsync(dw + W, &nch, sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
        C_NCCL(ncclAllGather(dw + W, dw, 1, ncclInt64, c->comm, c->stream));
        C_HIP(hipMemcpyAsync(widths.data(), dw, (size_t)W * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
        C_HIP(hipStreamSynchronize(c->stream));
    }
    int64_t wmax = 0, wsum = 0;
    for (int r = 0; r < W; ++r) {
        wmax = widths[(size_t)r] > wmax ? widths[(size_t)r] : wmax;
        wsum += widths[(size_t)r];
    }
    if (nchains_by_rank) memcpy(nchains_by_rank, widths.data(), (size_t)W * sizeof(int64_t));
    std::vector<uint64_t> mine((size_t)wmax, 0), all((size_t)wmax * (size_t)W, 0);
    for (int64_t k = 0; k < nch; ++k) mine[(size_t)k] = cnt[(size_t)k].ntrace;
    {
        if ((st = c->counts.need((size_t)wmax * (size_t)(W + 1) * sizeof(uint64_t))) != PDMP_OK) return st;
        uint64_t* dc = static_cast<uint64_t*>(c->counts.p);
        C_HIP(hipMemcpyAsync(dc + (size_t)wmax * W, mine.data(), (size_t)wmax * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        C_NCCL(ncclAllGather(dc + (size_t)wmax * W, dc, (size_t)wmax, ncclUint64, c->comm, c->stream));
        C_HIP(hipMemcpyAsync(all.data(), dc, all.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        C_HIP(hipStreamSynchronize(c->stream));
    }
    std::vector<uint64_t> tot((size_t)W, 0);
    uint64_t total = 0;
    {
        int64_t q = 0;
        for (int r = 0; r < W; ++r)
            for (int64_t k = 0; k < widths[(size_t)r]; ++k, ++q) {
                const uint64_t v = all[(size_t)r * (size_t)wmax + (size_t)k];
                tot[(size_t)r] += v;
                if (counts) {
                    if (q >= counts_cap) return cfail(PDMP_ERR_INVALID, "counts holds %lld entries, the ensemble has %lld chains", (long long)counts_cap, (long long)wsum);
                    counts[q] = v;
                }
            }
        for (int r = 0; r < W; ++r) total += tot[(size_t)r];
    }
    if (nevents_total) *nevents_total = (int64_t)total;
    // ---- compact the local segments
    void* evdev = nullptr;
    int64_t capdev = 0;
    if ((st = pdmp_ensemble_trace_dev(ens, &evdev, &capdev)) != PDMP_OK) return st;
    std::vector<uint64_t> offs((size_t)nch, 0);
    for (int64_t k = 1; k < nch; ++k) offs[(size_t)k] = offs[(size_t)k - 1] + mine[(size_t)k - 1];
    const uint64_t my_total = tot[(size_t)c->rank];
    if ((st = c->offs.need((size_t)(2 * nch) * sizeof(uint64_t))) != PDMP_OK) return st;
    if ((st = c->compact.need((size_t)(my_total ? my_total : 1) * sizeof(pdmp_event))) != PDMP_OK) return st;
    uint64_t* doffs = static_cast<uint64_t*>(c->offs.p);
    C_HIP(hipMemcpyAsync(doffs, offs.data(), (size_t)nch * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    C_HIP(hipMemcpyAsync(doffs + nch, mine.data(), (size_t)nch * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    if (my_total)
        hipLaunchKernelGGL(compact_traces_kernel, dim3((unsigned)nch, 4), dim3(256), 0, c->stream, static_cast<const uint4*>(evdev), capdev, doffs + nch,
                           doffs, static_cast<uint4*>(c->compact.p));
    C_HIP(hipGetLastError());
    // ---- gatherv to the root: one grouped send / recv, every peer over its own link
    pdmp_event* gdst = nullptr;
    if (c->rank == root) {
        if ((st = c->gathered.need((size_t)(total ? total : 1) * sizeof(pdmp_event))) != PDMP_OK) return st;
        gdst = static_cast<pdmp_event*>(c->gathered.p);
    }
    C_NCCL(ncclGroupStart());
    if (c->rank == root) {
        uint64_t at = 0;
        for (int r = 0; r < W; ++r) {
            if (r != root && tot[(size_t)r]) C_NCCL(ncclRecv(gdst + at, (size_t)tot[(size_t)r] * sizeof(pdmp_event), ncclChar, r, c->comm, c->stream));
            at += tot[(size_t)r];
        }
    } else if (my_total) {
        C_NCCL(ncclSend(c->compact.p, (size_t)my_total * sizeof(pdmp_event), ncclChar, root, c->comm, c->stream));
    }
    C_NCCL(ncclGroupEnd());
    if (c->rank == root) {
        uint64_t at = 0;
        for (int r = 0; r < root; ++r) at += tot[(size_t)r];
        if (my_total) C_HIP(hipMemcpyAsync(gdst + at, c->compact.p, (size_t)my_total * sizeof(pdmp_event), hipMemcpyDeviceToDevice, c->stream));
        if (events_host) {
            if ((int64_t)total > events_cap) return cfail(PDMP_ERR_INVALID, "events_host holds %lld events, %llu were gathered", (long long)events_cap, (unsigned long long)total);
            if (total) C_HIP(hipMemcpyAsync(events_host, gdst, (size_t)total * sizeof(pdmp_event), hipMemcpyDeviceToHost, c->stream));
        }
    }
    C_HIP(hipStreamSynchronize(c->stream));
    if (events_dev) *events_dev = (c->rank == root) ? (void*)gdst : nullptr;
    c->ngathered = (c->rank == root) ? (int64_t)total : 0;
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
    if ((st = pdmp_ensemble_batch_means(ens, T_prev, T, h.data(), h.data() + d)) != PDMP_OK) return st;
    C_HIP(hipSetDevice(c->device));
    if ((st = c->red.need((size_t)(2 * d) * sizeof(double))) != PDMP_OK) return st;
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
