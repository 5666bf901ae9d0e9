// synth_host.cpp -- formula-defined synthetic DNA collections on host threads (include/debwt_synth.h).
//
// The same arithmetic as debwt_amd/synth.py (splitmix64 streams; tests compare the two on small sizes), organised so
// that nothing but the base genome (one byte per base of ONE genome) is ever materialised: the SNP copies of the
// genome, the cut into records and the 2-bit packing happen while the final text words are written, any word range
// on any number of threads.  Measurement tooling -- the reference has no counterpart.
#include "../../include/debwt_hip.h"
#include "../../include/debwt_synth.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

namespace {

constexpr uint64_t GOLD = 0x9E3779B97F4A7C15ull;

inline uint64_t sm64(uint64_t x) {                       // synth.splitmix64
    uint64_t z = x + GOLD;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
template <typename... A> inline uint64_t mixv(A... vals) {  // synth._mix
    uint64_t h = 0x243F6A8885A308D3ull;
    const uint64_t v[] = {(uint64_t)vals...};
    for (uint64_t x : v) h = sm64(h ^ x);
    return h;
}
inline uint8_t uniform_code(uint64_t seed, uint64_t i) { return (uint8_t)((sm64(seed + (i >> 5)) >> (2 * (i & 31))) & 3); }
inline uint64_t rate_threshold(double rate) { return (uint64_t)(rate * 18446744073709551616.0); }   // int(rate * float(1 << 64))
// synth._mutate at array index a
inline uint8_t mutate_code(uint8_t code, uint64_t key, uint64_t a, uint64_t thr) {
    const uint64_t h = sm64(key + a * GOLD);
    return h < thr ? (uint8_t)((code + (h >> 61) % 3 + 1) & 3) : code;
}

// one overwrite of the base genome: positions [p, p + len) <- unit (tandem, period ulen; 0 = no period) mutated at `thr`
struct Overwrite {
    uint64_t p, len, ulen, unit_seed, key, thr;
    int pure;                                             // >= 0: every base is this code (homopolymer)
};

template <class F> void parallel_chunks(uint64_t total, uint64_t chunk, int threads, F &&f) {
    const uint64_t nchunks = (total + chunk - 1) / chunk;
    std::atomic<uint64_t> next{0};
    auto work = [&]() {
        for (;;) {
            const uint64_t c = next.fetch_add(1);
            if (c >= nchunks) return;
            f(c * chunk, std::min(total, (c + 1) * chunk));
        }
    };
    const int nt = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::max(threads, 1), nchunks));
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
}

}  // namespace

struct debwt_synth {
    uint64_t seed = 0, L = 0;
    uint32_t genomes = 1, nchrom = 1;
    std::vector<uint64_t> chrom_len, chrom_off;          // offsets of the records inside a genome
    std::vector<uint64_t> rec_start;                     // text position of every record's first base, + n at the end
    uint64_t n = 0, nrec = 0, snp_thr = 0;
    std::vector<uint64_t> snp_key;                       // per genome
    std::vector<uint8_t> g;                              // base genome
};

static void build_base_genome(debwt_synth *s, const debwt_synth_spec &sp, int threads) {
    const uint64_t L = s->L, seed = s->seed;
    s->g.resize(L);
    uint8_t *g = s->g.data();
    parallel_chunks((L + 31) >> 5, 1u << 16, threads, [&](uint64_t w0, uint64_t w1) {
        for (uint64_t w = w0; w < w1; w++) {
            const uint64_t v = sm64(seed + w);
            const uint64_t lim = std::min<uint64_t>(32, L - (w << 5));
            for (uint64_t j = 0; j < lim; j++) g[(w << 5) + j] = (uint8_t)((v >> (2 * j)) & 3);
        }
    });
    if (L < 2000) return;
    std::vector<Overwrite> ops;
    const uint64_t cap = std::max<uint64_t>(64, L / 8);
    {   // repeat families (synth.base_genome)
        const uint64_t thr = rate_threshold(0.02);
        uint64_t covered = 0, f = 0;
        const uint64_t target = (uint64_t)(sp.repeat_coverage * (double)L);
        while (covered < target) {
            uint64_t clen = 300 + mixv(seed, f, 1) % 5700;
            clen = std::min(clen, cap);
            const uint64_t copies = 5 + mixv(seed, f, 2) % 196;
            const uint64_t cons = mixv(seed, f, 3) & 0x7FFFFFFFFFFFull;
            for (uint64_t c = 0; c < copies && covered < target; c++) {
                ops.push_back(Overwrite{mixv(seed, f, 4, c) % (L - clen), clen, 0, cons, mixv(seed, f, 5, c), thr, -1});
                covered += clen;
            }
            f++;
        }
    }
    if (sp.alu_copies) {
        const uint64_t thr = rate_threshold(sp.alu_divergence);
        const uint64_t cons = mixv(seed, 0xA1, 0) & 0x7FFFFFFFFFFFull;
        for (uint64_t c = 0; c < sp.alu_copies; c++)
            ops.push_back(Overwrite{mixv(seed, 0xA1, 1, c) % (L - 300), 300, 0, cons, mixv(seed, 0xA1, 2, c), thr, -1});
    }
    if (sp.lowcx_fraction > 0) {
        const uint64_t thr = rate_threshold(0.01);
        uint64_t covered = 0, t = 0;
        const uint64_t target = (uint64_t)(sp.lowcx_fraction * (double)L);
        while (covered < target) {
            const uint64_t ulen = 5 + mixv(seed, 0x5A7, t, 1) % 167;
            const uint64_t alen = std::min<uint64_t>(1000 + mixv(seed, 0x5A7, t, 2) % 99000, cap);
            const uint64_t unit = mixv(seed, 0x5A7, t, 3) & 0x7FFFFFFFFFFFull;
            ops.push_back(Overwrite{mixv(seed, 0x5A7, t, 4) % (L - alen), alen, ulen, unit, mixv(seed, 0x5A7, t, 5), thr, -1});
            covered += alen;
            t++;
        }
        const uint64_t ntracts = (uint64_t)((double)L * sp.lowcx_fraction / 150);
        static const int AT[6] = {0, 3, 0, 3, 1, 2};
        for (t = 0; t < ntracts; t++) {
            const uint64_t kind = mixv(seed, 0x7AC, t, 0) % 3;
            Overwrite o{};
            o.pure = -1;
            if (kind < 2) {
                o.ulen = 1; o.len = 20 + mixv(seed, 0x7AC, t, 1) % 181;
                o.pure = AT[mixv(seed, 0x7AC, t, 2) % 6];
            } else {
                o.ulen = 2 + mixv(seed, 0x7AC, t, 1) % 5; o.len = 30 + mixv(seed, 0x7AC, t, 2) % 471;
                o.unit_seed = mixv(seed, 0x7AC, t, 3) & 0x7FFFFFFFFFFFull;
            }
            o.len = std::min(o.len, cap);
            o.p = mixv(seed, 0x7AC, t, 4) % (L - o.len);
            ops.push_back(o);
        }
    }
    // later overwrites win (the numpy definition applies them one after the other): every position chunk applies,
    // in list order, the pieces that fall into it.  Chunks find their overwrites through a start-sorted index.
    if (ops.empty()) return;
    const uint64_t CH = 1u << 20;
    uint64_t maxlen = 0;
    for (const Overwrite &o : ops) maxlen = std::max(maxlen, o.len);
    std::vector<uint32_t> by_start(ops.size());
    for (size_t i = 0; i < ops.size(); i++) by_start[i] = (uint32_t)i;
    std::sort(by_start.begin(), by_start.end(), [&](uint32_t a, uint32_t b) { return ops[a].p < ops[b].p; });
    std::vector<uint64_t> starts(ops.size());
    for (size_t i = 0; i < ops.size(); i++) starts[i] = ops[by_start[i]].p;
    parallel_chunks(L, CH, threads, [&](uint64_t a, uint64_t b) {
        // overwrites that can touch [a, b): start in [a - maxlen, b)
        const uint64_t lo = a > maxlen ? a - maxlen : 0;
        size_t i0 = std::lower_bound(starts.begin(), starts.end(), lo) - starts.begin();
        size_t i1 = std::lower_bound(starts.begin(), starts.end(), b) - starts.begin();
        std::vector<uint32_t> mine;
        for (size_t i = i0; i < i1; i++)
            if (ops[by_start[i]].p + ops[by_start[i]].len > a) mine.push_back(by_start[i]);
        std::sort(mine.begin(), mine.end());              // list order
        for (uint32_t id : mine) {
            const Overwrite &o = ops[id];
            const uint64_t x0 = std::max(a, o.p), x1 = std::min(b, o.p + o.len);
            for (uint64_t x = x0; x < x1; x++) {
                const uint64_t j = x - o.p;
                uint8_t code = o.pure >= 0 ? (uint8_t)o.pure : uniform_code(o.unit_seed, o.ulen ? j % o.ulen : j);
                if (o.thr) code = mutate_code(code, o.key, j, o.thr);
                g[x] = code;
            }
        }
    });
}

extern "C" int debwt_synth_open(const debwt_synth_spec *sp, int threads, debwt_synth **out) {
    if (!sp || !out || !sp->chrom_len || sp->genomes < 1 || sp->nchrom < 1 || sp->genome_len < 33) return DEBWT_EINVAL;
    debwt_synth *s = new (std::nothrow) debwt_synth();
    if (!s) return DEBWT_ENOMEM;
    s->seed = sp->seed; s->L = sp->genome_len; s->genomes = sp->genomes; s->nchrom = sp->nchrom;
    uint64_t sum = 0;
    for (uint32_t c = 0; c < sp->nchrom; c++) {
        if (sp->chrom_len[c] <= 32) { delete s; return DEBWT_EINVAL; }          // src/collect#$.c:41-45
        s->chrom_off.push_back(sum);
        s->chrom_len.push_back(sp->chrom_len[c]);
        sum += sp->chrom_len[c];
    }
    if (sum != sp->genome_len) { delete s; return DEBWT_EINVAL; }
    s->nrec = (uint64_t)sp->genomes * sp->nchrom;
    uint64_t pos = 0;
    for (uint32_t j = 0; j < sp->genomes; j++)
        for (uint32_t c = 0; c < sp->nchrom; c++) { s->rec_start.push_back(pos); pos += s->chrom_len[c] + 1; }
    s->rec_start.push_back(pos);
    s->n = pos;
    s->snp_thr = sp->genomes > 1 ? rate_threshold(sp->snp_rate) : 0;           // synth.pan_genome: one record, no SNPs
    for (uint32_t j = 0; j < sp->genomes; j++) s->snp_key.push_back(mixv(sp->seed, 0xC0FFEE, j));
    try {
        build_base_genome(s, *sp, threads);
    } catch (const std::bad_alloc &) { delete s; return DEBWT_ENOMEM; }
    *out = s;
    return DEBWT_OK;
}

extern "C" void debwt_synth_close(debwt_synth *s) { delete s; }
extern "C" uint64_t debwt_synth_n(const debwt_synth *s) { return s ? s->n : 0; }
extern "C" uint64_t debwt_synth_nrec(const debwt_synth *s) { return s ? s->nrec : 0; }
extern "C" uint64_t debwt_synth_nwords(const debwt_synth *s) { return s ? ((s->n + 63) >> 5) + 2 : 0; }

extern "C" int debwt_synth_sep(const debwt_synth *s, uint64_t *sep) {
    if (!s || !sep) return DEBWT_EINVAL;
    for (uint64_t r = 0; r < s->nrec; r++) sep[r] = s->rec_start[r + 1] - 1;
    return DEBWT_OK;
}

extern "C" int debwt_synth_codes(const debwt_synth *s, uint32_t genome, uint64_t i0, uint64_t i1, uint8_t *dst) {
    if (!s || !dst || genome >= s->genomes || i0 > i1 || i1 > s->L) return DEBWT_EINVAL;
    const uint64_t key = s->snp_key[genome], thr = s->snp_thr;
    const uint8_t *g = s->g.data();
    for (uint64_t i = i0; i < i1; i++) dst[i - i0] = thr ? mutate_code(g[i], key, i, thr) : g[i];
    return DEBWT_OK;
}

extern "C" int debwt_synth_words(const debwt_synth *s, uint64_t w0, uint64_t w1, int threads, uint64_t *dst,
                                 uint64_t census[4]) {
    if (!s || !dst || w0 > w1 || w1 > debwt_synth_nwords(s)) return DEBWT_EINVAL;
    const uint64_t n = s->n;
    const uint8_t *g = s->g.data();
    std::atomic<uint64_t> cen[4];
    for (auto &c : cen) c = 0;
    parallel_chunks(w1 - w0, 1u << 15, threads, [&](uint64_t a, uint64_t b) {
        uint64_t cnt[4] = {0, 0, 0, 0};
        uint64_t P = (w0 + a) << 5;
        const uint64_t Pend = (w0 + b) << 5;
        // record that holds P (or the one whose separator P is); positions >= n are padding
        size_t rec = P < n ? (size_t)(std::upper_bound(s->rec_start.begin(), s->rec_start.end(), P) - s->rec_start.begin()) - 1 : 0;
        uint64_t word = 0;
        auto put = [&](uint64_t code) {
            word |= code << ((31 - (P & 31)) << 1);
            if ((P & 31) == 31) { dst[(P >> 5) - w0] = word; word = 0; }
            P++;
        };
        while (P < Pend) {
            if (P >= n) { put(P < n + 32 ? 3 : 0); continue; }                 // 32 'T' behind the end, then zeros
            const uint64_t rs = s->rec_start[rec], sepos = s->rec_start[rec + 1] - 1;
            const uint32_t j = (uint32_t)(rec / s->nchrom), c = (uint32_t)(rec % s->nchrom);
            const uint64_t key = s->snp_key[j], thr = s->snp_thr;
            uint64_t i = s->chrom_off[c] + (P - rs);
            const uint64_t stop = std::min(sepos, Pend);
            while (P < stop) {
                const uint8_t code = thr ? mutate_code(g[i], key, i, thr) : g[i];
                cnt[code]++;
                i++;
                put(code);
            }
            if (P == sepos && P < Pend) { put(3); rec++; }                      // 'T' at the separator (src/collect#$.c:85)
        }
        for (int q = 0; q < 4; q++) cen[q] += cnt[q];
    });
    if (census) for (int q = 0; q < 4; q++) census[q] += cen[q].load();
    return DEBWT_OK;
}
