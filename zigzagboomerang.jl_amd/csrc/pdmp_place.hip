// pdmp_place.hip -- where the large state arrays lie in HBM.
//
// Measured on the MI355X (tools/probes/, DESIGN.md 5 "The timing modes are a property of the allocation"): device memory falls into THREE classes of
// about 96 GB each.  A scatter of 128-byte line reads with small writes between them -- what the event loops do to the per-chain records -- sustains
// 3.7 TB/s of lines while everything it touches is of one class, 4.4 TB/s on two classes, 4.9 on three; a read-only scatter does not care.  One
// hipMalloc of a few GB lands in one class or straddles two by the state of the driver's allocator: that was the "fast" and the "slow" timing mode of
// the full-width launch (38 / 45 ms per slice of C3, the same binary, by the allocation).  The classes are not visible through any API, but they are
// measurable in milliseconds: two chunks of one class run a short two-chunk scatter ~20 % slower than two chunks of different classes.
//
// EXPERIMENTAL, OFF BY DEFAULT.  An array of several GB is built from 1 GB chunks (hipMemCreate), each classified against one
// reference chunk per class found so far, and mapped into one contiguous address range in a chosen order of the classes ("012" cycled, or a pattern per
// array); chunks the pattern has no use for are held until the walk ends (the allocator would hand them out again) and then
// released.  Kernels see an ordinary pointer.  What it showed (tools/probes/place_matrix.sh): records striped over the three classes run the
// full-width C3 slice in 39.5-39.9 ms whatever else happens -- neither the 38.3 nor the 45.5 ms of a plain hipMalloc.  Why it is not the default: on this
// ROCm (7.2) memory that went through hipMemCreate / hipMemMap / hipMemUnmap / hipMemRelease in quick succession was seen to LOSE WRITES (chains of a freshly
// set state dead at t = 0) and, with many ensembles alive, to raise GPU memory access faults.  The shipped answer to the timing modes is therefore the
// probe-and-reallocate loop of pdmp_capi.hip (init_state_tuned), which uses hipMalloc only.
// Switched on per ensemble: pdmp_debug_set_placement (include/pdmp_debug.h; the Python wrapper forwards PDMP_PLACE, PDMP_PLACE_rec / _kp / _ev).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "pdmp_engine.hpp"

namespace pdmp {
namespace {

// waves [0, n/2) scatter inside chunk a, the others inside chunk b; every wave owns a slab of `slab_lines` 128-byte lines (a power of two): a dependent
// walk of (read 16 B of a random line, write them into another line).  The contents are permuted within the chunks, which nobody has written yet.
__global__ __launch_bounds__(64) void place_pair_scatter_kernel(uint4* a, uint4* b, uint32_t slab_lines, uint32_t iters) {
    const uint32_t half = gridDim.x / 2, w = blockIdx.x;
    uint4* slab = (w < half ? a + (size_t)w * slab_lines * 8 : b + (size_t)(w - half) * slab_lines * 8);
    uint32_t s = w * 64u + threadIdx.x + 12345u, acc = 0;
    const uint32_t mask = slab_lines - 1u;
    for (uint32_t k = 0; k < iters; ++k) {
        s = s * 1664525u + 1013904223u;
        const uint32_t i = ((s >> 8) ^ acc) & mask;
        const uint4 v = slab[(size_t)i * 8 + (threadIdx.x & 7)];
        acc += v.x & 0u;  // (only the dependency matters)
        const uint32_t j = ((s >> 9) * 2654435761u >> 7) & mask;
        slab[(size_t)j * 8 + ((threadIdx.x + 3) & 7)] = v;
    }
}

struct Walked {
    hipMemGenericAllocationHandle_t h{};
    void* tmp = nullptr;  // where the chunk is mapped while it is being classified
    int cls = -1;
};

// One reference chunk per class, found by the first placement on a device and kept mapped for the life of the process (3 chunks): the labels 0 / 1 / 2
// then mean the same thing in every array of every ensemble.
struct ClassRefs {
    int dev = -1;
    size_t chunk = 0;
    std::vector<Walked> ref;
    std::vector<float> self_ms;
};
ClassRefs g_refs;

}  // namespace

void placed_free(Placement& p) {
    if (!p.va) return;
    size_t off = 0;
    for (size_t k = 0; k < p.handles.size(); ++k) {
        (void)hipMemUnmap((char*)p.va + off, p.chunk);
        (void)hipMemRelease((hipMemGenericAllocationHandle_t)p.handles[k]);
        off += p.chunk;
    }
    (void)hipMemAddressFree(p.va, p.va_bytes);
    p.handles.clear();
    p.va = nullptr;
    p.va_bytes = 0;
}

bool placed_alloc(size_t bytes, Placement& out, const PlaceConfig* cfg, const std::string* forced_pattern) {
    out = Placement{};
    if (!cfg || !cfg->enabled) return false;  // EXPERIMENTAL, off by default: see the header
    const size_t chunk = cfg->chunk_mb << 20;
    // the classes of the array's chunks in address order, cycled: "012" by default; a pattern of its own per array for experiments
    const bool forced = forced_pattern && !forced_pattern->empty();
    const std::string pattern = forced ? *forced_pattern : std::string("012");
    for (char ch : pattern)
        if (ch < '0' || ch > '2') return false;
    if (chunk < ((size_t)64 << 20) || (chunk & (chunk - 1)) != 0 || (!forced && bytes < (cfg->min_mb << 20))) return false;
    const size_t need = (bytes + chunk - 1) / chunk, max_walk = std::max<size_t>(cfg->max_walk, need);
    if (need < 3 && !forced) return false;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (g_refs.dev != dev || g_refs.chunk != chunk) {
        if (g_refs.dev >= 0) return false;  // (one device and one chunk size per process)
        g_refs.dev = dev;
        g_refs.chunk = chunk;
    }
    const auto t_begin = std::chrono::steady_clock::now();
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        (void)hipGetLastError();
        return false;
    }
    const int waves = 4096;
    const uint32_t slab_lines = (uint32_t)(chunk / 128 / (waves / 2)), iters = 300;
    bool hip_ok = true;
    auto pair_ms = [&](void* a, void* b) -> float {
        place_pair_scatter_kernel<<<waves, 64, 0, nullptr>>>((uint4*)a, (uint4*)b, slab_lines, iters / 4);
        hip_ok &= hipEventRecord(e0, nullptr) == hipSuccess;
        place_pair_scatter_kernel<<<waves, 64, 0, nullptr>>>((uint4*)a, (uint4*)b, slab_lines, iters);
        hip_ok &= hipEventRecord(e1, nullptr) == hipSuccess;
        hip_ok &= hipEventSynchronize(e1) == hipSuccess;
        float ms = 0;
        hip_ok &= hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
        return ms;
    };
    auto drop = [&](Walked& w) {
        if (w.tmp) {
            (void)hipMemUnmap(w.tmp, chunk);
            (void)hipMemAddressFree(w.tmp, chunk);
            w.tmp = nullptr;
        }
        (void)hipMemRelease(w.h);
    };

    // the walk: chunk after chunk until every class has its share (or the limits say stop)
    size_t quota[4] = {0, 0, 0, 0};
    for (size_t k = 0; k < need; ++k) quota[pattern[k % pattern.size()] - '0'] += 1;
    std::vector<Walked> kept, surplus;
    std::vector<Walked>& ref = g_refs.ref;
    std::vector<float>& self_ms = g_refs.self_ms;
    size_t have[4] = {0, 0, 0, 0}, walked = 0;
    bool failed = false;
    while (walked < max_walk && (have[0] < quota[0] || have[1] < quota[1] || have[2] < quota[2])) {
        size_t freeb = 0, totb = 0;
        if (hipMemGetInfo(&freeb, &totb) != hipSuccess || freeb < 4 * chunk) break;
        Walked w;
        if (hipMemCreate(&w.h, chunk, &prop, 0) != hipSuccess) break;
        ++walked;
        if (hipMemAddressReserve(&w.tmp, chunk, 0, nullptr, 0) != hipSuccess) {
            w.tmp = nullptr;
            drop(w);
            failed = true;
            break;
        }
        if (hipMemMap(w.tmp, chunk, 0, w.h, 0) != hipSuccess) {
            (void)hipMemAddressFree(w.tmp, chunk);
            w.tmp = nullptr;
            drop(w);
            failed = true;
            break;
        }
        if (hipMemSetAccess(w.tmp, chunk, &acc, 1) != hipSuccess) {
            drop(w);
            failed = true;
            break;
        }
        // same class as a reference: the pair runs no faster than the reference paired with itself (measured: 1.08 x; another class: 0.89 x)
        for (size_t c = 0; c < ref.size() && w.cls < 0; ++c) {
            float ms = pair_ms(ref[c].tmp, w.tmp);
            if (ms > 0.95f * self_ms[c] && ms < 1.02f * self_ms[c]) ms = std::min(ms, pair_ms(ref[c].tmp, w.tmp));  // (near the line: once more)
            if (ms >= 0.985f * self_ms[c]) w.cls = (int)c;
        }
        if (!hip_ok) {
            drop(w);
            failed = true;
            break;
        }
        if (w.cls < 0 && ref.size() < 3) {  // a class not seen before: this chunk becomes its reference and stays (it is not part of any array)
            w.cls = (int)ref.size();
            self_ms.push_back(pair_ms(w.tmp, w.tmp));
            ref.push_back(w);
            continue;
        }
        if (w.cls < 0) w.cls = 3;  // (like none of the three: a chunk that straddles; only a filler)
        if (w.cls < 3 && have[w.cls] < quota[w.cls]) {
            kept.push_back(w);
            have[w.cls] += 1;
        } else {
            surplus.push_back(w);
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    // classes that stayed short are filled from what the walk set aside
    while (!failed && kept.size() < need && !surplus.empty()) {
        kept.push_back(surplus.back());
        surplus.pop_back();
    }
    if (failed || kept.size() < need) {
        for (auto& w : kept) drop(w);
        for (auto& w : surplus) drop(w);
        (void)hipGetLastError();
        return false;
    }
    // round robin over the classes, into one address range
    std::vector<Walked> order;
    {
        std::vector<std::vector<Walked>> by(4);
        for (auto& w : kept) by[(size_t)w.cls].push_back(w);
        for (size_t k = 0; order.size() < kept.size(); ++k) {
            size_t c = (size_t)(pattern[k % pattern.size()] - '0');
            for (size_t t = 0; t < 4 && by[c].empty(); ++t) c = (c + 1) % 4;  // (a class that stayed short: the next one that has chunks)
            order.push_back(by[c].back());
            by[c].pop_back();
        }
    }
    while (order.size() > need) {  // (a class may have been kept beyond what is needed)
        surplus.push_back(order.back());
        order.pop_back();
    }
    for (auto& w : surplus) drop(w);
    void* va = nullptr;
    if (hipMemAddressReserve(&va, need * chunk, 0, nullptr, 0) != hipSuccess) {
        for (auto& w : order) drop(w);
        (void)hipGetLastError();
        return false;
    }
    bool ok = true;
    size_t mapped = 0;
    for (auto& w : order) {
        (void)hipMemUnmap(w.tmp, chunk);
        (void)hipMemAddressFree(w.tmp, chunk);
        w.tmp = nullptr;
        if (ok && hipMemMap((char*)va + mapped * chunk, chunk, 0, w.h, 0) == hipSuccess) ++mapped;
        else ok = false;
    }
    if (ok) ok = hipMemSetAccess(va, need * chunk, &acc, 1) == hipSuccess;
    if (!ok) {
        for (size_t k = 0; k < mapped; ++k) (void)hipMemUnmap((char*)va + k * chunk, chunk);
        for (auto& w : order) (void)hipMemRelease(w.h);
        (void)hipMemAddressFree(va, need * chunk);
        (void)hipGetLastError();
        return false;
    }
    out.va = va;
    out.va_bytes = need * chunk;
    out.chunk = chunk;
    for (auto& w : order) {
        out.handles.push_back((void*)w.h);
        out.classes.push_back((char)('0' + w.cls));
    }
    out.walked = walked;
    out.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    return true;
}

}  // namespace pdmp
