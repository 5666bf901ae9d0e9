// Calibration of FETCH_SIZE for the GATHER pattern of the blue sort (round 6; VERDICT r05 "what's weak" 5): every lane reads the two
// 64-bit words that hold one SP window (21 symbols x 3 bits at a random symbol index: 8-byte aligned, 7 % of them straddle a 128-byte
// line) out of a table the size of the packed SP code of a 30 Gbp build (1.18 GB), a KNOWN number of gathers (2^28 per launch).
// Run once plain (rate) and once under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (bytes the counter reports per gather):
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/gather16.hip -o build/gather16 && build/gather16
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r06/pmc_gather16 -o c -- build/gather16
// Kernels: k_gather<0> window words (8-byte aligned pair, may straddle), k_gather<1> one 16-byte aligned slot, k_stream (reads the
// whole table once, 16 bytes per lane and step: the wide-read pattern the guide's x2 correction was validated on).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ u64 mix(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
template <int ALIGNED>
__global__ __launch_bounds__(256) void k_gather(const u64 *__restrict__ tab, u64 words, u32 per_lane, u64 *__restrict__ out) {
    const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
    u64 acc = 0;
    for (u32 r = 0; r < per_lane; r += 4) {
        u64 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {                                   // four gathers in flight per lane, as the block kernels hold
            u64 w = mix(gid * 0x9E3779B97F4A7C15ull + r + u) % (words - 2);
            if (ALIGNED) w &= ~1ull;
            a[u] = tab[w]; b[u] = tab[w + 1];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) acc += a[u] ^ b[u];
    }
    if (acc == 0x123456789ull) out[gid & 1023] = acc;
}
__global__ __launch_bounds__(256) void k_stream(const ulonglong2 *__restrict__ tab, u64 n16, u64 *__restrict__ out) {
    u64 acc = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n16; i += (u64)gridDim.x * 256) { const ulonglong2 v = tab[i]; acc += v.x ^ v.y; }
    if (acc == 0x123456789ull) out[threadIdx.x] = acc;
}
int main() {
    const u64 bytes = 1180ull << 20, words = bytes / 8;
    u64 *tab, *out;
    CHK(hipMalloc(&tab, bytes)); CHK(hipMalloc(&out, 1 << 20)); CHK(hipMemset(tab, 1, bytes));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const u32 grid = 16384, per_lane = 64;                              // 2^28 gathers per launch
    const double gathers = (double)grid * 256 * per_lane;
    for (int rep = 0; rep < 3; rep++) {
        float ms;
        CHK(hipEventRecord(e0)); k_gather<0><<<grid, 256>>>(tab, words, per_lane, out); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_gather<0> window words (8-byte aligned pair): %.0f gathers in %.3f ms = %.2f G gathers/s\n", gathers, ms, gathers / ms * 1e-6);
        CHK(hipEventRecord(e0)); k_gather<1><<<grid, 256>>>(tab, words, per_lane, out); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_gather<1> 16-byte aligned slot:               %.0f gathers in %.3f ms = %.2f G gathers/s\n", gathers, ms, gathers / ms * 1e-6);
        CHK(hipEventRecord(e0)); k_stream<<<4096, 256>>>((const ulonglong2 *)tab, bytes / 16, out); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_stream whole table (%.3f GB):                 %.3f ms = %.2f TB/s\n", bytes * 1e-9, ms, bytes / ms * 1e-9);
    }
    printf("known: %.0f gathers per k_gather launch, %llu bytes per k_stream launch\n", gathers, bytes);
    return 0;
}
