// Does a working set that fits the 256 MB Infinity Cache stream faster than HBM?  Two buffers of S MiB, copied back and forth
// (sequential loads, stores as 128-byte lines to F advancing fronts as a radix pass writes them), REPS times per measurement.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/mall_pingpong.hip -o build/mall_pingpong && build/mall_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(512) void k_copy(const ulonglong2 *__restrict__ src, ulonglong2 *__restrict__ dst, u64 nlines, u64 fronts,
                                              u64 per_front, u64 chunk_lines) {
    const u64 g0 = (u64)blockIdx.x * chunk_lines;
    for (u64 s = 0; s < chunk_lines; s += 64) {
        const u64 g = g0 + s + (threadIdx.x >> 3);
        if (g >= nlines) break;
        const u32 within = threadIdx.x & 7u;
        const ulonglong2 v = src[(g << 3) + within];
        const u64 d = fronts > 1 ? (g % fronts) * per_front + g / fronts : g;
        dst[(d << 3) + within] = v;
    }
}
__global__ __launch_bounds__(256) void k_read(const ulonglong2 *__restrict__ src, u64 n16, u64 *out) {
    u64 acc = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n16; i += (u64)gridDim.x * 256) { const ulonglong2 v = src[i]; acc += v.x ^ v.y; }
    if (acc == 0x1234567ull) out[0] = acc;
}
int main() {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    u64 *out; CHK(hipMalloc(&out, 64));
    for (u64 mib : {16ull, 32ull, 64ull, 96ull, 128ull, 192ull, 256ull, 512ull, 2048ull, 8192ull}) {
        const u64 bytes = mib << 20;
        ulonglong2 *a, *b;
        CHK(hipMalloc(&a, bytes)); CHK(hipMalloc(&b, bytes)); CHK(hipMemset(a, 1, bytes)); CHK(hipMemset(b, 2, bytes));
        const u64 nlines = bytes >> 7;
        const int reps = (int)(16384 / mib) + 2;
        for (u64 fronts : {1ull, 4096ull}) {
            const u64 chunk_lines = 1024;                        // 128 KiB of input per workgroup
            const u32 grid = (u32)((nlines + chunk_lines - 1) / chunk_lines);
            k_copy<<<grid, 512>>>(a, b, nlines, fronts, nlines / fronts, chunk_lines);
            CHK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) {
                if (r & 1) k_copy<<<grid, 512>>>(b, a, nlines, fronts, nlines / fronts, chunk_lines);
                else k_copy<<<grid, 512>>>(a, b, nlines, fronts, nlines / fronts, chunk_lines);
            }
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            printf("2 x %5llu MiB  copy %-12s %8.1f us per pass  %7.1f GB/s (read + written)\n", mib, fronts > 1 ? "4096 fronts" : "sequential",
                   ms / reps * 1e3, 2.0 * bytes * reps / ms * 1e-6);
        }
        k_read<<<4096, 256>>>(a, bytes / 16, out);
        CHK(hipEventRecord(e0));
        for (int r = 0; r < reps; r++) k_read<<<4096, 256>>>(a, bytes / 16, out);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("    %5llu MiB  read only         %8.1f us per pass  %7.1f GB/s\n", mib, ms / reps * 1e3, (double)bytes * reps / ms * 1e-6);
        CHK(hipFree(a)); CHK(hipFree(b));
    }
    return 0;
}
