// malloc_cost.hip -- what hipMalloc of the build's large buffers costs a cold process on this driver, and whether several host
// threads asking at once get it sooner (the CLI's cold run waits 0.7 - 3.8 s for 26 GB buffers: profiles/r06_cli_3.1G.txt).
// hipcc --offload-arch=gfx950 -O2 -o malloc_cost scripts/micro/malloc_cost.hip -lpthread ; ./malloc_cost [GB per buffer = 24] [buffers = 4] [threads = 1]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const size_t gb = argc > 1 ? (size_t)atoi(argv[1]) : 24;
    const int nbuf = argc > 2 ? atoi(argv[2]) : 4, T = argc > 3 ? atoi(argv[3]) : 1;
    double t0 = now();
    hipSetDevice(0);
    hipFree(nullptr);
    printf("context %.3f s\n", now() - t0);
    std::vector<void *> p((size_t)nbuf, nullptr);
    std::vector<double> took((size_t)nbuf, 0);
    t0 = now();
    auto work = [&](int t) {
        hipSetDevice(0);
        for (int i = t; i < nbuf; i += T) { const double a = now(); if (hipMalloc(&p[(size_t)i], gb << 30) != hipSuccess) p[(size_t)i] = nullptr; took[(size_t)i] = now() - a; }
    };
    { std::vector<std::thread> th; for (int t = 1; t < T; t++) th.emplace_back(work, t); work(0); for (auto &x : th) x.join(); }
    const double all = now() - t0;
    printf("%d x %zu GB by %d thread(s): %.3f s in all = %.1f GB/s;", nbuf, gb, T, all, (double)nbuf * gb / all);
    for (int i = 0; i < nbuf; i++) printf(" %.3f%s", took[(size_t)i], p[(size_t)i] ? "" : "(failed)");
    printf("\n");
    t0 = now();
    for (void *q : p) if (q) hipFree(q);
    printf("freed in %.3f s\n", now() - t0);
    t0 = now();
    for (int i = 0; i < nbuf; i++) { if (hipMalloc(&p[(size_t)i], gb << 30) != hipSuccess) p[(size_t)i] = nullptr; }
    printf("the same again in the same process: %.3f s\n", now() - t0);
    for (void *q : p) if (q) hipFree(q);
    return 0;
}
