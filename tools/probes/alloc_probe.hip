// Does a latency-bound scatter of 128-byte lines see the "timing mode" of an ALLOCATION?  K buffers of `chains` x `bytes_per_chain`, all kept alive;
// on each: `chains` wavefronts, each inside its own chain's slab, `iters` dependent rounds of (64 lanes read one random line each [+ write 16 B of another]).
//   hipcc --offload-arch=gfx950 -O2 -o alloc_probe alloc_probe.hip ;  ./alloc_probe [K=8] [chains=4096] [MB_per_chain=2] [iters=2000] [write=1] [alloc: 0 hipMalloc, 1 one arena]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(64) void chase(uint4* __restrict__ buf, size_t lines_per_chain, uint32_t iters, int write, uint32_t* sink) {
    uint4* slab = buf + (size_t)blockIdx.x * lines_per_chain * 8;  // 8 x 16 B per line
    uint32_t s = blockIdx.x * 64u + threadIdx.x + 12345u;
    uint32_t acc = 0;
    const uint32_t mask = (uint32_t)lines_per_chain - 1u;
    for (uint32_t k = 0; k < iters; ++k) {
        s = s * 1664525u + 1013904223u;
        uint32_t i = ((s >> 8) ^ acc) & mask;
        uint4 v = slab[(size_t)i * 8 + (threadIdx.x & 7)];
        acc += v.x;  // the next index depends on the load
        if (write) {
            uint32_t j = ((s >> 9) * 2654435761u >> 7) & mask;
            slab[(size_t)j * 8 + ((threadIdx.x + 3) & 7)] = make_uint4(0u, 0u, 0u, 0u);  // (keeps the buffer zero: acc stays 0 and the walk is the LCG's)
        }
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

int main(int argc, char** argv) {
    int K = argc > 1 ? atoi(argv[1]) : 8;
    int chains = argc > 2 ? atoi(argv[2]) : 4096;
    size_t mb = argc > 3 ? (size_t)atoi(argv[3]) : 2;
    uint32_t iters = argc > 4 ? (uint32_t)atoi(argv[4]) : 2000;
    int write = argc > 5 ? atoi(argv[5]) : 1;
    int arena = argc > 6 ? atoi(argv[6]) : 0;
    size_t bytes = (size_t)chains * mb << 20, lines_per_chain = (mb << 20) / 128;
    std::vector<uint4*> bufs;
    uint32_t* sink;
    CK(hipMalloc(&sink, 4));
    char* big = nullptr;
    if (arena) CK(hipMalloc(&big, bytes * (size_t)K));
    for (int k = 0; k < K; ++k) {
        uint4* p;
        if (arena) p = (uint4*)(big + bytes * (size_t)k);
        else CK(hipMalloc(&p, bytes));
        CK(hipMemset(p, 0, bytes));
        bufs.push_back(p);
    }
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep)
        for (int k = 0; k < K; ++k) {
            for (int half = 0; half < 2; ++half) {
                int n = half ? chains / 2 : chains;
                chase<<<n, 64>>>(bufs[k], lines_per_chain, iters / 10, write, sink);  // warm
                CK(hipEventRecord(a));
                chase<<<n, 64>>>(bufs[k], lines_per_chain, iters, write, sink);
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float ms;
                CK(hipEventElapsedTime(&ms, a, b));
                double lines = (double)n * 64 * iters * (write ? 2 : 1);
                printf("%s buf %d @%p %d waves: %.2f ms  %.2f TB/s of lines\n", half ? "   " : "rep", k, (void*)bufs[k], n, ms, lines * 128 / ms / 1e9);
            }
        }
    return 0;
}
