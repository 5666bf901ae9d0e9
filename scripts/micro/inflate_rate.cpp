// inflate_rate.cpp -- the ingest's DEFLATE decoder and CRC-32 (debwt_amd/csrc/fast_inflate.h) against zlib's on FASTA-like text,
// one thread and T threads side by side (host only).  g++ -O3 -std=c++17 -o inflate_rate scripts/micro/inflate_rate.cpp -lz -lpthread
// usage: inflate_rate [Mbp per thread = 100] [threads = 16] [fastq]      (-DFI_LROOT=12: the litlen root table of 12 bits)
#include "../../debwt_amd/csrc/fast_inflate.h"
#include <chrono>
#include <cstdio>
#include <random>
#include <string>
#include <thread>
#include <vector>
using namespace fastinflate;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const size_t len = (size_t)(argc > 1 ? atoi(argv[1]) : 100) * 1000000;
    const int threads = argc > 2 ? atoi(argv[2]) : 16;
    std::mt19937_64 rng(1);
    std::string s(len, 'A');
    for (auto &c : s) c = "ACGT"[rng() & 3];
    for (size_t i = 60; i < len; i += 61) s[i] = '\n';
    const bool fastq = argc > 3 && !strcmp(argv[3], "fastq");       // reads of 100 b with 41 quality values: literals, not matches
    if (fastq) {
        size_t o = 0;
        for (size_t r = 0; o + 220 < len; r++) {
            o += (size_t)snprintf(&s[o], 16, "@r%010zu\n", r);
            for (int i = 0; i < 100; i++) s[o++] = "ACGT"[rng() & 3];
            s[o++] = '\n'; s[o++] = '+'; s[o++] = '\n';
            for (int i = 0; i < 100; i++) s[o++] = (char)(33 + rng() % 41);
            s[o++] = '\n';
        }
        for (; o < len; o++) s[o] = '\n';
    }
    for (int level : {1, 6}) {
        z_stream zs; memset(&zs, 0, sizeof zs);
        deflateInit2(&zs, level, Z_DEFLATED, -15, 8, 0);
        std::vector<unsigned char> z(deflateBound(&zs, (uLong)s.size()) + 64);
        zs.next_in = (Bytef *)s.data(); zs.avail_in = (uInt)s.size(); zs.next_out = z.data(); zs.avail_out = (uInt)z.size();
        deflate(&zs, Z_FINISH); z.resize(z.size() - zs.avail_out + 8); deflateEnd(&zs);
        const uint32_t want_crc = (uint32_t)crc32(0, (const Bytef *)s.data(), (uInt)len);
        for (int T : {1, threads}) {
            std::vector<std::vector<unsigned char>> out(T, std::vector<unsigned char>(len + 64));
            for (int what = 0; what < 4; what++) {                  // 0 fast inflate, 1 zlib inflate, 2 crc32_fast, 3 zlib crc32
                double best = 1e9;
                int ok = 1;
                for (int rep = 0; rep < 3; rep++) {
                    std::vector<int> good(T, 0);
                    auto work = [&](int t) {
                        if (what == 0) {
                            Decoder *d = new Decoder(); d->start(z.data(), z.size() - 8, 0); size_t pos = 0;
                            const int rc = d->run(out[t].data(), 0, &pos, out[t].size(), ~(size_t)0);
                            good[t] = rc == FI_DONE && pos == len && !memcmp(out[t].data(), s.data(), len);
                            delete d;
                        } else if (what == 1) {
                            z_stream is; memset(&is, 0, sizeof is); inflateInit2(&is, -15);
                            is.next_in = z.data(); is.avail_in = (uInt)z.size() - 8; is.next_out = out[t].data(); is.avail_out = (uInt)out[t].size();
                            good[t] = inflate(&is, Z_FINISH) == Z_STREAM_END && is.total_out == len; inflateEnd(&is);
                        } else if (what == 2) good[t] = crc32_fast(0, out[t].data(), len) == want_crc;
                        else good[t] = (uint32_t)crc32(0, out[t].data(), (uInt)len) == want_crc;
                    };
                    const double t0 = now();
                    std::vector<std::thread> th;
                    for (int t = 1; t < T; t++) th.emplace_back(work, t);
                    work(0);
                    for (auto &x : th) x.join();
                    best = std::min(best, now() - t0);
                    for (int g : good) ok &= g;
                }
                static const char *name[4] = {"fast inflate", "zlib inflate", "crc32_fast", "zlib crc32"};
                printf("gzip -%d (%.2f x) %2d thread(s) %-12s %.3f s = %6.2f GB/s of text per thread, %6.2f in all, correct %d\n", level, (double)len / (z.size() - 8), T,
                       name[what], best, len / best / 1e9, T * (len / best / 1e9), ok);
            }
        }
    }
}
