// What the memory system gives an 8-bit radix pass: a copy of N bytes that reads sequentially and writes granules of G bytes
// the way the pass does -- F write fronts that each advance sequentially, consecutive source granules going to different
// fronts (transpose order) -- or to a pseudo-random permutation of the granules.  Prints GB/s (read + written) per shape.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/scatter_shape.hip -o build/scatter_shape && build/scatter_shape [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// mode 0: sequential; 1: fronts (transpose); 2: random permutation (odd multiplier mod 2^k granules)
template <int MODE>
__global__ __launch_bounds__(512) void k_copy(const ulonglong2 *__restrict__ src, ulonglong2 *__restrict__ dst, u64 ngran, u32 gshift,
                                              u64 fronts, u64 per_front, u64 chunk_gran) {
    // a workgroup takes chunk_gran consecutive source granules, as a chunk of the pass; 16 bytes per thread and step
    const u32 tpg = 1u << (gshift - 4);                       // threads per granule
    const u64 g0 = (u64)blockIdx.x * chunk_gran;
    const u32 gpw = 512u >> (gshift - 4);                     // granules per workgroup step
    for (u64 s = 0; s < chunk_gran; s += gpw) {
        const u64 g = g0 + s + (threadIdx.x >> (gshift - 4));
        if (g >= ngran) break;
        const u32 within = threadIdx.x & (tpg - 1u);
        const ulonglong2 v = src[(g << (gshift - 4)) + within];
        u64 d;
        if (MODE == 0) d = g;
        else if (MODE == 1) d = (g % fronts) * per_front + g / fronts;
        else d = (g * 0x9E3779B97F4A7C15ull) & (ngran - 1);
        dst[(d << (gshift - 4)) + within] = v;
    }
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 32.0;
    u64 bytes = 1; while ((double)(bytes << 1) <= gib * 1073741824.0) bytes <<= 1;   // power of two
    ulonglong2 *a, *b;
    CHK(hipMalloc(&a, bytes)); CHK(hipMalloc(&b, bytes));
    CHK(hipMemset(a, 1, bytes)); CHK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    printf("buffer %.1f GiB each; GB/s = (read + written) / time; peak 8000\n", bytes / 1073741824.0);
    const u64 fronts_list[] = {131072, 32768, 8192};
    for (u32 gshift = 7; gshift <= 12; gshift++) {
        const u64 ngran = bytes >> gshift;
        for (int mode = 0; mode < 5; mode++) {
            if (mode == 0 && gshift != 7) continue;
            const u64 fronts = mode >= 1 && mode <= 3 ? fronts_list[mode - 1] : 1;
            const u64 per_front = ngran / fronts;
            const u64 chunk_gran = (2ull << 20) >> gshift;              // 2 MiB of input per workgroup, as the pass
            const u32 grid = (u32)((ngran + chunk_gran - 1) / chunk_gran);
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                CHK(hipEventRecord(e0));
                if (mode == 0) k_copy<0><<<grid, 512>>>(a, b, ngran, gshift, fronts, per_front, chunk_gran);
                else if (mode <= 3) k_copy<1><<<grid, 512>>>(a, b, ngran, gshift, fronts, per_front, chunk_gran);
                else k_copy<2><<<grid, 512>>>(a, b, ngran, gshift, fronts, per_front, chunk_gran);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const char *names[] = {"sequential", "131072 fronts", "32768 fronts", "8192 fronts", "random granules"};
            printf("granule %5u B  %-16s %8.3f ms  %7.1f GB/s  %.3f of peak\n", 1u << gshift, names[mode], best, 2.0 * bytes / best * 1e-6,
                   2.0 * bytes / best * 1e-6 / 8000.0);
        }
    }
    return 0;
}
