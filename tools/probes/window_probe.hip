// Where in one big allocation does a scatter of 128-byte lines (read one line, write 16 B of another; `chains` wavefronts, each inside its own 2 MB slab)
// run fast?  (a) slabs packed (stride 2 MB) into windows at increasing offsets of the arena; (b) the same slabs spread over the arena with larger strides.
//   hipcc --offload-arch=gfx950 -O2 -o region_probe region_probe.hip ;  ./region_probe [arena_GB=128] [chains=4096] [iters=1000] [window step GB=8]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(64) void chase(uint4* __restrict__ buf, size_t second_off_lines, int nfirst, size_t stride_lines, uint32_t lines_per_chain, uint32_t iters, int write, uint32_t* sink) {
    uint4* slab = buf + (size_t)blockIdx.x * stride_lines * 8 + ((int)blockIdx.x >= nfirst ? second_off_lines * 8 : 0);
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

static size_t g_second = 0;
static int g_nfirst = 1 << 30;
static float run(uint4* base, size_t stride_bytes, int chains, uint32_t iters, int write, uint32_t* sink, hipEvent_t a, hipEvent_t b) {
    chase<<<chains, 64>>>(base, g_second / 128, g_nfirst, stride_bytes / 128, (2u << 20) / 128, iters / 8, write, sink);
    hipEventRecord(a);
    chase<<<chains, 64>>>(base, g_second / 128, g_nfirst, stride_bytes / 128, (2u << 20) / 128, iters, write, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main(int argc, char** argv) {
    size_t gb = argc > 1 ? (size_t)atoi(argv[1]) : 128;
    int chains = argc > 2 ? atoi(argv[2]) : 4096;
    uint32_t iters = argc > 3 ? (uint32_t)atoi(argv[3]) : 1000;
    size_t step = argc > 4 ? (size_t)atoi(argv[4]) : 8;
    char* arena;
    uint32_t* sink;
    CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&arena, gb << 30));
    CK(hipMemset(arena, 0, gb << 30));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const size_t packed = (size_t)chains * (2u << 20);
    // two half-footprints: chains [0, chains/2) packed at +0, the others packed X further
    printf("== two windows of %d chains each, read + write; arena %zu GB\n", chains / 2, gb);
    for (size_t x = step; x + packed <= (gb << 30); x += step << 30 >> 0) {
        if (x == step) x = step << 30;
        g_second = x - packed / 2;  // (the second half starts at +x instead of +packed/2)
        g_nfirst = chains / 2;
        float ms = run((uint4*)arena, 2u << 20, chains, iters, 1, sink, a, b);
        printf("second window at +%3zu GB: %6.2f ms  %.2f TB/s\n", x >> 30, ms, (double)chains * 64 * iters * 2 * 128 / ms / 1e9);
    }
    g_second = 0;
    g_nfirst = 1 << 30;
    if (argc > 5) return 0;
    for (int write = 1; write >= 0; --write) {
        printf("== %s, %d waves; arena %zu GB at %p\n", write ? "read a line + write 16 B" : "read only", chains, gb, (void*)arena);
        for (size_t off = 0; off + packed <= (gb << 30); off += step << 30) {
            float ms = run((uint4*)(arena + off), 2u << 20, chains, iters, write, sink, a, b);
            printf("packed at +%3zu GB: %6.2f ms  %.2f TB/s\n", off >> 30, ms, (double)chains * 64 * iters * (write ? 2 : 1) * 128 / ms / 1e9);
        }
        for (size_t stride_mb = 4; (size_t)chains * (stride_mb << 20) <= (gb << 30); stride_mb *= 2) {
            float ms = run((uint4*)arena, stride_mb << 20, chains, iters, write, sink, a, b);
            printf("spread, stride %3zu MB (%3zu GB): %6.2f ms  %.2f TB/s\n", stride_mb, ((size_t)chains * stride_mb) >> 10, ms,
                   (double)chains * 64 * iters * (write ? 2 : 1) * 128 / ms / 1e9);
        }
    }
    return 0;
}
