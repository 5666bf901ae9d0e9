// Read-only streaming rate (what bounds a histogram pass): every workgroup sums a chunk of the array, U 16-byte loads per lane in
// flight.   hipcc -O3 --offload-arch=gfx950 scripts/micro/read_stream.hip -o build/read_stream && build/read_stream [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int U, int NT>
__global__ __launch_bounds__(NT) void k_read(const ulonglong2 *__restrict__ src, u64 n16, u64 chunk16, u64 *__restrict__ out) {
    const u64 beg = (u64)blockIdx.x * chunk16, end = beg + chunk16 < n16 ? beg + chunk16 : n16;
    u64 acc = 0;
    for (u64 i = beg + threadIdx.x; i < end; i += (u64)NT * U) {
        ulonglong2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = i + (u64)u * NT < end ? src[i + (u64)u * NT] : make_ulonglong2(0, 0);
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u].x ^ v[u].y;
    }
    if (acc == 0x123456789ull) out[blockIdx.x] = acc;
}
template <int U, int NT> static void run(const ulonglong2 *a, u64 bytes, u64 *out, u32 chunks, const char *name) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const u64 n16 = bytes / 16, chunk16 = (n16 + chunks - 1) / chunks;
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        CHK(hipEventRecord(e0));
        k_read<U, NT><<<chunks, NT>>>(a, n16, chunk16, out);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-34s chunks %6u  %8.3f ms  %7.1f GB/s  %.3f of peak\n", name, chunks, best, bytes / best * 1e-6, bytes / best * 1e-6 / 8000.0);
}
int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 32.0;
    const u64 bytes = (u64)(gib * 1073741824.0);
    ulonglong2 *a; u64 *out;
    CHK(hipMalloc(&a, bytes)); CHK(hipMalloc(&out, 1 << 20)); CHK(hipMemset(a, 1, bytes));
    for (u32 chunks : {2048u, 4096u, 16384u, 65536u}) {
        run<1, 256>(a, bytes, out, chunks, "256 threads, 1 x 16 B in flight");
        run<2, 256>(a, bytes, out, chunks, "256 threads, 2 x 16 B in flight");
        run<4, 256>(a, bytes, out, chunks, "256 threads, 4 x 16 B in flight");
        run<2, 512>(a, bytes, out, chunks, "512 threads, 2 x 16 B in flight");
        run<4, 1024>(a, bytes, out, chunks, "1024 threads, 4 x 16 B in flight");
    }
    return 0;
}
