// Which chunks of device memory share a "class" (a scatter of line reads + small writes over two chunks runs at 3.7 TB/s when they are of one class,
// 4.4 TB/s when they are not)?  Creates 1 GB chunks one after the other (hipMemCreate + map), classifies each against one reference per class found so far.
//   ./rank_probe [nchunks=240] [iters=300] [chunk_MB=1024]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// waves [0, n/2) scatter inside chunk a, the others inside chunk b: each wave owns a slab of `slab_lines` lines
__global__ __launch_bounds__(64) void pair_scatter(uint4* a, uint4* b, uint32_t slab_lines, uint32_t iters) {
    const uint32_t half = gridDim.x / 2, w = blockIdx.x;
    uint4* slab = (w < half ? a + (size_t)w * slab_lines * 8 : b + (size_t)(w - half) * slab_lines * 8);
    uint32_t s = w * 64u + threadIdx.x + 12345u, acc = 0;
    const uint32_t mask = slab_lines - 1u;
    for (uint32_t k = 0; k < iters; ++k) {
        s = s * 1664525u + 1013904223u;
        uint32_t i = ((s >> 8) ^ acc) & mask;
        uint4 v = slab[(size_t)i * 8 + (threadIdx.x & 7)];
        acc += v.x & 0u;  // (the data may be anything: only the dependency matters)
        uint32_t j = ((s >> 9) * 2654435761u >> 7) & mask;
        slab[(size_t)j * 8 + ((threadIdx.x + 3) & 7)] = v;
    }
}

struct Chunk { hipMemGenericAllocationHandle_t h; void* va; int cls; };
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    int nchunks = argc > 1 ? atoi(argv[1]) : 240;
    uint32_t iters = argc > 2 ? (uint32_t)atoi(argv[2]) : 300;
    size_t chunk = (size_t)(argc > 3 ? atoi(argv[3]) : 1024) << 20;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int waves = 4096;
    const uint32_t slab_lines = (uint32_t)(chunk / 128 / (waves / 2));  // power of two for power-of-two chunks
    auto pair_ms = [&](void* a, void* b) -> float {
        pair_scatter<<<waves, 64>>>((uint4*)a, (uint4*)b, slab_lines, iters / 4);
        hipEventRecord(e0);
        pair_scatter<<<waves, 64>>>((uint4*)a, (uint4*)b, slab_lines, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        return ms;
    };
    std::vector<Chunk> all;
    std::vector<int> ref;  // index of the reference chunk of each class
    double t_create = 0, t_probe = 0;
    float self_ms = 0;
    for (int k = 0; k < nchunks; ++k) {
        size_t freeb, totb;
        CK(hipMemGetInfo(&freeb, &totb));
        if (freeb < 3 * chunk) break;
        Chunk c{};
        double t0 = now();
        CK(hipMemAddressReserve(&c.va, chunk, 0, nullptr, 0));
        CK(hipMemCreate(&c.h, chunk, &prop, 0));
        CK(hipMemMap(c.va, chunk, 0, c.h, 0));
        CK(hipMemSetAccess(c.va, chunk, &acc, 1));
        t_create += now() - t0;
        t0 = now();
        if (k == 0) self_ms = pair_ms(c.va, c.va);  // both halves inside one chunk: the same-class time
        c.cls = -1;
        std::vector<float> tms;
        for (size_t r = 0; r < ref.size() && c.cls < 0; ++r) {
            float ms = pair_ms(all[ref[r]].va, c.va);
            tms.push_back(ms);
            if (ms > 0.92f * self_ms) c.cls = (int)r;  // as slow as one chunk alone: same class
        }
        if (c.cls < 0) {
            c.cls = (int)ref.size();
            ref.push_back(k);
        }
        t_probe += now() - t0;
        all.push_back(c);
        printf("chunk %3d class %d  (self %.2f ms;", k, c.cls, self_ms);
        for (float v : tms) printf(" %.2f", v);
        printf(")\n");
    }
    printf("created %zu chunks in %.2f s (%.1f ms each), probes %.2f s; classes: %zu\n", all.size(), t_create, 1e3 * t_create / all.size(), t_probe, ref.size());
    printf("sequence: ");
    for (auto& c : all) printf("%d", c.cls);
    printf("\n");
    return 0;
}
