// Can a CONTIGUOUS virtual range be backed by physical memory that is SPREAD over the device?  Reserve `chains` x 2 MB of address space, back it with
// chunks created one after the other with spacer allocations between them (released afterwards), and run region_probe's scatter on it.
//   ./vmm_probe [chunk_MB=1024] [spacer_GB=15] [chains=4096] [iters=1000]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(64) void chase(uint4* __restrict__ buf, size_t stride_lines, uint32_t lines_per_chain, uint32_t iters, int write, uint32_t* sink) {
    uint4* slab = buf + (size_t)blockIdx.x * stride_lines * 8;
    uint32_t s = blockIdx.x * 64u + threadIdx.x + 12345u, acc = 0;
    const uint32_t mask = lines_per_chain - 1u;
    for (uint32_t k = 0; k < iters; ++k) {
        s = s * 1664525u + 1013904223u;
        uint32_t i = ((s >> 8) ^ acc) & mask;
        uint4 v = slab[(size_t)i * 8 + (threadIdx.x & 7)];
        acc += v.x;
        if (write) {
            uint32_t j = ((s >> 9) * 2654435761u >> 7) & mask;
            slab[(size_t)j * 8 + ((threadIdx.x + 3) & 7)] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

int main(int argc, char** argv) {
    size_t chunk = (size_t)(argc > 1 ? atoi(argv[1]) : 1024) << 20;
    size_t spacer = (size_t)(argc > 2 ? atoi(argv[2]) : 15) << 30;
    int chains = argc > 3 ? atoi(argv[3]) : 4096;
    uint32_t iters = argc > 4 ? (uint32_t)atoi(argv[4]) : 1000;
    int dev = 0;
    CK(hipSetDevice(dev));
    uint32_t* sink;
    CK(hipMalloc(&sink, 4));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    size_t freeb = 0, total = 0;
    CK(hipMemGetInfo(&freeb, &total));
    printf("granularity %zu B; free %.1f of %.1f GB\n", gran, freeb / 1073741824.0, total / 1073741824.0);
    const size_t bytes = (size_t)chains * (2u << 20);
    const size_t nchunk = (bytes + chunk - 1) / chunk;
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, nchunk * chunk, 0, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> keep(nchunk), spacers;
    for (size_t k = 0; k < nchunk; ++k) {
        CK(hipMemCreate(&keep[k], chunk, &prop, 0));
        if (spacer && k + 1 < nchunk) {
            hipMemGenericAllocationHandle_t h;
            CK(hipMemCreate(&h, spacer, &prop, 0));
            spacers.push_back(h);
        }
    }
    for (size_t k = 0; k < nchunk; ++k) CK(hipMemMap((char*)va + k * chunk, chunk, 0, keep[k], 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, nchunk * chunk, &acc, 1));
    for (auto h : spacers) CK(hipMemRelease(h));
    CK(hipMemGetInfo(&freeb, &total));
    printf("%zu chunks of %zu MB, spacers of %zu GB released: free %.1f GB\n", nchunk, chunk >> 20, spacer >> 30, freeb / 1073741824.0);
    CK(hipMemset(va, 0, bytes));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int write = 1; write >= 0; --write)
        for (int rep = 0; rep < 2; ++rep) {
            chase<<<chains, 64>>>((uint4*)va, (2u << 20) / 128, (2u << 20) / 128, iters / 8, write, sink);
            CK(hipEventRecord(a));
            chase<<<chains, 64>>>((uint4*)va, (2u << 20) / 128, (2u << 20) / 128, iters, write, sink);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            printf("%s: %6.2f ms  %.2f TB/s\n", write ? "read a line + write 16 B" : "read only", ms, (double)chains * 64 * iters * (write ? 2 : 1) * 128 / ms / 1e9);
        }
    // the same scatter on a plain hipMalloc of the same size, for the comparison
    void* plain;
    CK(hipMalloc(&plain, bytes));
    CK(hipMemset(plain, 0, bytes));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        chase<<<chains, 64>>>((uint4*)plain, (2u << 20) / 128, (2u << 20) / 128, iters, 1, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("plain hipMalloc, read + write: %6.2f ms  %.2f TB/s\n", ms, (double)chains * 64 * iters * 2 * 128 / ms / 1e9);
    }
    return 0;
}
