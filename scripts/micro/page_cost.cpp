// page_cost.cpp -- what filling and releasing 1 GB of fresh memory costs a process of 16 threads on this host: 4 KB pages against
// transparent huge pages (debwt_amd/csrc/gz_parallel.h big_malloc), regions released one after the other against by all threads.
// g++ -O2 -std=c++17 -o page_cost scripts/micro/page_cost.cpp -lpthread
#include "../../debwt_amd/csrc/gz_parallel.h"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const int T = 16;
    const size_t per = (size_t)64 << 20;                      // 16 x 64 MB = 1 GB
    for (int huge = 0; huge < 2; huge++)
        for (int par_free = 0; par_free < 2; par_free++) {
            std::vector<char *> r(T);
            double t0 = now();
            for (int t = 0; t < T; t++) r[t] = (char *)(huge ? big_malloc(per) : malloc(per));
            const double t_alloc = now() - t0;
            t0 = now();
            { std::vector<std::thread> th; for (int t = 0; t < T; t++) th.emplace_back([&, t] { memset(r[t], 1, per); }); for (auto &x : th) x.join(); }
            const double t_fill = now() - t0;
            t0 = now();
            { std::vector<std::thread> th; for (int t = 0; t < T; t++) th.emplace_back([&, t] { memset(r[t], 2, per); }); for (auto &x : th) x.join(); }
            const double t_again = now() - t0;
            t0 = now();
            if (par_free) { std::vector<std::thread> th; for (int t = 0; t < T; t++) th.emplace_back([&, t] { free(r[t]); }); for (auto &x : th) x.join(); }
            else for (int t = 0; t < T; t++) free(r[t]);
            const double t_free = now() - t0;
            printf("%s pages, released %s: reserve %.4f s, first fill by 16 threads %.4f s, second fill %.4f s, release %.4f s\n", huge ? "huge (madvise)" : "4 KB", par_free ? "by 16 threads   " : "one after another", t_alloc, t_fill, t_again, t_free);
        }
}
