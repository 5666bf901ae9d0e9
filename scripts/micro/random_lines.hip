// What the memory system gives a PROBING pass (k_sp_flags: one 16-byte slot out of a random 128-byte line per probe): every lane
// reads 16 bytes at a hashed 128-byte-line address of a table, U independent probes in flight per lane, WPS waves per SIMD asked of
// the compiler, for tables the size of the node table of a 30 Gbp build (4 GiB: HBM), of the Infinity Cache (128 MiB), of one L2
// (2 MiB), and with DEP = 1 the probes of a lane chained (the address of the next depends on the data of the one before: the serial
// candidate loop).  Lines per second and "line bytes" per second = lines x 128 (what FETCH_SIZE counts for such a kernel).
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/random_lines.hip -o build/random_lines && build/random_lines
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
template <int U, int WPS, int DEP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPS, WPS)))
void k_probe(const ulonglong2 *__restrict__ tab, u64 line_mask, u32 rounds, u64 *__restrict__ out) {
    const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
    u64 acc = 0, chain = 0;
    for (u32 r = 0; r < rounds; r++) {
        ulonglong2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u64 h = mix(gid * 0x9E3779B97F4A7C15ull + (u64)r * U + u + chain);
            v[u] = tab[((h & line_mask) << 3) + ((h >> 40) & 7u)];      // one 16-byte slot of a random 128-byte line
            if (DEP) chain = v[u].x & 1ull;                             // the next address waits for this answer
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u].x ^ v[u].y;
    }
    if (acc == 0x123456789ull) out[gid & 1023] = acc;
}
template <int U, int WPS, int DEP> static void run(const ulonglong2 *tab, u64 bytes, u64 *out, const char *what) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const u64 lines = bytes / 128;
    const u32 grid = 256 * 4 * WPS * 8;                                  // eight workgroup waves over the chip
    const u32 rounds = 512 / U;                                          // 512 probes per lane
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        CHK(hipEventRecord(e0));
        k_probe<U, WPS, DEP><<<grid, 256>>>(tab, lines - 1, rounds, out);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double probes = (double)grid * 256.0 * rounds * U;
    printf("%-22s %d probes in flight per lane%s, %d waves/SIMD: %7.2f G lines/s = %6.2f TB/s of lines (%.3f of the 8 TB/s peak)\n", what, U,
           DEP ? " (chained)" : "", WPS, probes / best * 1e-6, probes * 128.0 / best * 1e-9, probes * 128.0 / best * 1e-9 / 8.0);
}
int main() {
    ulonglong2 *tab; u64 *out;
    const u64 big = 4ull << 30;
    CHK(hipMalloc(&tab, big)); CHK(hipMalloc(&out, 1 << 20)); CHK(hipMemset(tab, 1, big));
    struct { u64 bytes; const char *what; } sizes[] = {{4ull << 30, "4 GiB table (HBM)"}, {128ull << 20, "128 MiB table (MALL)"}, {2ull << 20, "2 MiB table (L2)"}};
    for (auto &s : sizes) {
        run<1, 4, 1>(tab, s.bytes, out, s.what);
        run<1, 4, 0>(tab, s.bytes, out, s.what);
        run<2, 4, 0>(tab, s.bytes, out, s.what);
        run<4, 4, 0>(tab, s.bytes, out, s.what);
        run<8, 4, 0>(tab, s.bytes, out, s.what);
        run<1, 8, 0>(tab, s.bytes, out, s.what);
        run<4, 8, 0>(tab, s.bytes, out, s.what);
    }
    return 0;
}
