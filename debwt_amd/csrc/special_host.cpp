// special_host.cpp -- see special_host.h.  Restates SURVEY 8a items 3-4 on packed windows.
#include "special_host.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <functional>
#include <thread>
#include <chrono>
#include <cstdio>

namespace {

struct Text {
    const uint64_t *w;
    uint64_t n;
    const uint64_t *sep;
    uint64_t nrec;

    // 32 symbols from position i (reference `convert`, src/collect#$.c:243-251)
    uint64_t window(uint64_t i) const {
        uint64_t a = w[i >> 5];
        unsigned sh = (unsigned)(i & 31) << 1;
        return sh ? (a << sh) | (w[(i >> 5) + 1] >> (64 - sh)) : a;
    }
    unsigned base(uint64_t i) const { return (unsigned)(w[i >> 5] >> ((31 - (i & 31)) << 1)) & 3u; }
    uint64_t first_sep_at_or_after(uint64_t i) const { return std::lower_bound(sep, sep + nrec, i) - sep; }

    // true suffix order of two different positions (src/collect#$.c:253-311)
    bool less(uint64_t a, uint64_t b) const { return less(a, first_sep_at_or_after(a), b, first_sep_at_or_after(b)); }
    // the same with the records known: ia, ib = index of the first separator at or after a, b
    bool less(uint64_t a, uint64_t ia, uint64_t b, uint64_t ib) const {
        for (;;) {
            uint64_t da = sep[ia] - a, db = sep[ib] - b;
            uint64_t m = da < db ? da : db;
            while (m) {
                unsigned c = m < 32 ? (unsigned)m : 32u;
                uint64_t wa = window(a) >> (64 - 2 * c), wb = window(b) >> (64 - 2 * c);
                if (wa != wb) return wa < wb;
                a += c; b += c; m -= c;
            }
            if (da != db) return da > db;           // the side at its separator is the larger one
            bool enda = ia == nrec - 1, endb = ib == nrec - 1;
            if (enda != endb) return endb;          // '$' > '#'
            if (enda) return false;                 // same position: not reached for a != b
            a++; b++; ia++; ib++;                   // equal '#': keep comparing
        }
    }

    // equal K-windows, separator of the same kind at the same offset (src/collect#$.c:603-634)
    bool same_window(uint64_t a, uint64_t ia, uint64_t b, uint64_t ib, int K) const {
        uint64_t da = sep[ia] - a, db = sep[ib] - b;
        if (da != db || (ia == nrec - 1) != (ib == nrec - 1)) return false;
        // K symbols from a and from b, the separator's slot (offset da < K) left out of the comparison
        const uint64_t slot = 3ull << (2 * (31 - da));
        const uint64_t keep = (~0ull << (64 - 2 * K)) & ~slot;
        return ((window(a) ^ window(b)) & keep) == 0;
    }
};

// Host threads for the module: the reference sorts the N*K special suffixes with one qsort; collections of many
// records (contigs, reads) make that the long pole, so above a few thousand suffixes the work is cut into chunks.
// DEBWT_SPECIAL_THREADS / DEBWT_SPECIAL_PAR_MIN override the thread count / the threshold (tests).
unsigned special_threads(uint64_t NS) {
    const char *e = getenv("DEBWT_SPECIAL_PAR_MIN");
    const uint64_t par_min = e ? strtoull(e, nullptr, 10) : (1ull << 14);
    if (NS < par_min) return 1;
    const char *t = getenv("DEBWT_SPECIAL_THREADS");
    unsigned nt = t ? (unsigned)atoi(t) : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    return std::max(1u, nt);
}

// fn(chunk) for chunk = 0 .. nchunks-1 on nt threads (dynamic hand-out)
void parallel_chunks(unsigned nt, uint64_t nchunks, const std::function<void(uint64_t)> &fn) {
    if (nt <= 1 || nchunks <= 1) { for (uint64_t c = 0; c < nchunks; c++) fn(c); return; }
    std::atomic<uint64_t> next{0};
    auto worker = [&]() { for (;;) { uint64_t c = next.fetch_add(1); if (c >= nchunks) return; fn(c); } };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(worker);
    worker();
    for (auto &x : th) x.join();
}

}  // namespace

void build_special_tables(const uint64_t *words, uint64_t n, const uint64_t *sep, uint64_t nrec, int K,
                          SpecialTables *out) {
    Text T{words, n, sep, nrec};
    const bool trace = getenv("DEBWT_TRACE_SPECIAL") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "special module: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    const uint64_t NS = nrec * (uint64_t)K;
    const uint64_t node_mask = (1ull << (2 * K)) - 1;
    // The T-padded key is monotone in the true suffix order (a separator ranks above every base and the
    // padding is the largest base), so ordering by key first and by the full suffix comparison only inside
    // runs of equal keys gives exactly the order of the reference's qsort (src/collect#$.c:118-157,253-311)
    // at a fraction of its comparisons.
    // rec: the record whose separator follows pos (index into sep); next: the 32 symbols behind that separator (every
    // record is longer than 32), ~0 behind '$' -- where suffixes with equal keys differ first, as a rule
    struct Item { uint64_t key, pos, rec, next; };
    std::vector<Item> items(NS);
    const unsigned nt = special_threads(NS);
    out->threads_used = nt;
    const uint64_t nchunks = nt > 1 ? (uint64_t)nt * 4 : 1;
    auto cut = [&](uint64_t total, uint64_t c) { return total / nchunks * c + std::min<uint64_t>(c, total % nchunks); };
    parallel_chunks(nt, nchunks, [&](uint64_t c) {
        for (uint64_t r = cut(nrec, c); r < cut(nrec, c + 1); r++) {
            uint64_t m = r * (uint64_t)K;
            const uint64_t next = r + 1 < nrec ? T.window(sep[r] + 1) : ~0ull;
            for (int d = K - 1; d >= 0; d--) {
                uint64_t p = sep[r] - (uint64_t)d;
                // key: the d bases, then 'T' up to K symbols (src/collect#$.c:428-446)
                uint64_t win = d ? (T.window(p) >> (64 - 2 * d)) : 0;
                uint64_t pad = (1ull << (2 * (K - d))) - 1;
                items[m].key = ((win << (2 * (K - d))) | pad) & node_mask;
                items[m].pos = p;
                items[m].rec = r;
                items[m].next = next;
                m++;
            }
        }
    });
    lap("keys");
    // by key (ties by position, so that the result is the same for every thread count).  Many suffixes: one counting
    // pass on the top 12 bits of the key cuts them into 4096 buckets that are then sorted independently on all threads
    // (no merge levels); few: one sort.
    auto by_key = [](const Item &a, const Item &b) { return a.key != b.key ? a.key < b.key : a.pos < b.pos; };
    if (nt > 1) {
        constexpr unsigned NB = 4096;
        const int bshift = 2 * K > 12 ? 2 * K - 12 : 0;
        std::vector<std::vector<uint64_t>> cnt(nchunks, std::vector<uint64_t>(NB, 0));
        parallel_chunks(nt, nchunks, [&](uint64_t c) {
            for (uint64_t i = cut(NS, c); i < cut(NS, c + 1); i++) cnt[c][(items[i].key >> bshift) & (NB - 1)]++;
        });
        std::vector<uint64_t> bstart(NB + 1, 0);
        for (unsigned b = 0; b < NB; b++) {                      // bucket-major, chunk-minor offsets
            uint64_t acc = bstart[b];
            for (uint64_t c = 0; c < nchunks; c++) { const uint64_t v = cnt[c][b]; cnt[c][b] = acc; acc += v; }
            bstart[b + 1] = acc;
        }
        std::vector<Item> tmp(NS);
        parallel_chunks(nt, nchunks, [&](uint64_t c) {
            for (uint64_t i = cut(NS, c); i < cut(NS, c + 1); i++) tmp[cnt[c][(items[i].key >> bshift) & (NB - 1)]++] = items[i];
        });
        items.swap(tmp);
        parallel_chunks(nt, NB, [&](uint64_t b) { std::sort(items.begin() + bstart[b], items.begin() + bstart[b + 1], by_key); });
    } else {
        std::sort(items.begin(), items.end(), by_key);
    }
    lap("sort by key");
    // runs of equal keys: true suffix order inside each (independent of each other)
    std::vector<std::pair<uint64_t, uint64_t>> runs;       // [start, end) of the runs of length > 1
    {
        std::vector<std::vector<std::pair<uint64_t, uint64_t>>> part(nchunks);   // a chunk lists the runs that START inside it
        parallel_chunks(nt, nchunks, [&](uint64_t c) {
            uint64_t i = cut(NS, c);
            const uint64_t end = cut(NS, c + 1);
            while (i < end && i > 0 && items[i].key == items[i - 1].key) i++;
            while (i < end) {
                uint64_t j = i + 1;
                while (j < NS && items[j].key == items[i].key) j++;
                if (j - i > 1) part[c].push_back({i, j});
                i = j;
            }
        });
        for (auto &v : part) runs.insert(runs.end(), v.begin(), v.end());
    }
    // equal keys with the separator at the same offset: '#' ties and the comparison goes on in the next records (32
    // symbols of them are in the items); '$' is larger than '#'
    auto by_suffix = [&](const Item &a, const Item &b) {
        const bool enda = a.rec == nrec - 1, endb = b.rec == nrec - 1;
        const bool same_offset = sep[a.rec] - a.pos == sep[b.rec] - b.pos;     // equal keys may still be "ACG#" and "ACGT#"
        if (same_offset && !enda && !endb && a.next != b.next) return a.next < b.next;
        return a.pos != b.pos && T.less(a.pos, a.rec, b.pos, b.rec);
    };
    // long runs first, each on all threads (pieces sorted, then merged pairwise): the suffixes a base or two before
    // their separator share their key with a large part of the collection.  (The run ends were found before any thread
    // moves an item: looking for them while other runs are being sorted would read items another thread is swapping.)
    constexpr uint64_t LONG_RUN = 1u << 15;
    std::vector<std::pair<uint64_t, uint64_t>> short_runs;
    for (const auto &run : runs) {
        const uint64_t i = run.first, len = run.second - run.first;
        if (nt == 1 || len < LONG_RUN) { short_runs.push_back(run); continue; }
        const uint64_t pieces = std::min<uint64_t>((uint64_t)nt * 2, len / 4096);
        auto pcut = [&](uint64_t c) { return i + len / pieces * c + std::min<uint64_t>(c, len % pieces); };
        parallel_chunks(nt, pieces, [&](uint64_t c) { std::sort(items.begin() + pcut(c), items.begin() + pcut(c + 1), by_suffix); });
        for (uint64_t width = 1; width < pieces; width *= 2) {
            const uint64_t pairs = (pieces + 2 * width - 1) / (2 * width);
            parallel_chunks(nt, pairs, [&](uint64_t q) {
                const uint64_t a = q * 2 * width, b = std::min(pieces, a + width), e = std::min(pieces, a + 2 * width);
                if (b < e) std::inplace_merge(items.begin() + pcut(a), items.begin() + pcut(b), items.begin() + pcut(e), by_suffix);
            });
        }
    }
    const uint64_t rchunks = nt > 1 ? std::min<uint64_t>(short_runs.size(), (uint64_t)nt * 16) : 1;
    parallel_chunks(nt, short_runs.empty() ? 0 : rchunks, [&](uint64_t c) {
        const uint64_t r0 = short_runs.size() / rchunks * c + std::min<uint64_t>(c, short_runs.size() % rchunks);
        const uint64_t r1 = short_runs.size() / rchunks * (c + 1) + std::min<uint64_t>(c + 1, short_runs.size() % rchunks);
        for (uint64_t r = r0; r < r1; r++) std::sort(items.begin() + short_runs[r].first, items.begin() + short_runs[r].second, by_suffix);
    });
    lap("tie runs");
    std::vector<uint64_t> order(NS), orec(NS);
    out->key.resize(NS);
    out->chr.resize(NS);
    parallel_chunks(nt, nchunks, [&](uint64_t c) {
        for (uint64_t s = cut(NS, c); s < cut(NS, c + 1); s++) {
            order[s] = items[s].pos;
            orec[s] = items[s].rec;
            out->key[s] = items[s].key;
            out->chr[s] = (uint8_t)T.base(items[s].pos - 1);      // always a base: records are > K long
        }
    });
    out->pos = order;

    lap("tables");
    // special branches (src/collect#$.c:534-598)
    out->branch.clear();
    {
        // groups of equal windows are consecutive in `order`; a chunk takes the groups that START inside it
        std::vector<std::vector<uint64_t>> part(nchunks);
        parallel_chunks(nt, nchunks, [&](uint64_t c) {
            uint64_t i = cut(NS, c);
            const uint64_t end = cut(NS, c + 1);
            while (i < end && i > 0 && T.same_window(order[i - 1], orec[i - 1], order[i], orec[i], K)) i++;   // inside a group of the chunk before
            while (i < end) {
                uint64_t j = i + 1;
                while (j < NS && T.same_window(order[i], orec[i], order[j], orec[j], K)) j++;
                if (j - i >= 2) {
                    bool differ = false;
                    for (uint64_t q = i + 1; q < j; q++)
                        if (T.base(order[q] + K) != T.base(order[i] + K)) differ = true;
                    if (differ)
                        for (uint64_t q = i; q < j; q++) part[c].push_back(order[q]);
                }
                i = j;
            }
        });
        for (auto &v : part) out->branch.insert(out->branch.end(), v.begin(), v.end());
    }
    std::sort(out->branch.begin(), out->branch.end());

    lap("branches");
    // head# / head$ and tail# nodes (src/collect#$.c:468-533)
    out->head_keys.resize(nrec);
    out->tail_facts.resize(nrec);
    for (uint64_t r = 0; r < nrec; r++) {
        uint64_t start = r ? sep[r - 1] + 1 : 0;
        out->head_keys[r] = ((T.window(start) >> (64 - 2 * K)) << 2) | 3ull;
        out->tail_facts[r] = ((T.window(sep[r] - (uint64_t)K) >> (64 - 2 * K)) << 2) | 1ull;
    }
    std::sort(out->head_keys.begin(), out->head_keys.end());
    lap("heads and tails");
}

void special_order_record_starts(const uint64_t *words, uint64_t n, const uint64_t *sep, uint64_t nrec, uint32_t *ord,
                                 const uint32_t *gid, const uint32_t *act, uint64_t na) {
    Text T{words, n, sep, nrec};
    std::vector<uint64_t> starts;                       // first element of every group in act
    for (uint64_t i = 0; i < na; i++)
        if (i == 0 || gid[act[i]] != gid[act[i - 1]]) starts.push_back(i);
    starts.push_back(na);
    const uint64_t ngroups = starts.size() - 1;
    const unsigned nt = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    parallel_chunks(ngroups > 1 ? nt : 1, ngroups, [&](uint64_t g) {
        std::vector<uint32_t> recs(starts[g + 1] - starts[g]);
        for (size_t x = 0; x < recs.size(); x++) recs[x] = ord[act[starts[g] + x]];
        std::sort(recs.begin(), recs.end(), [&](uint32_t a, uint32_t b) {
            if (a == b) return false;
            const uint64_t pa = a ? sep[a - 1] + 1 : 0, pb = b ? sep[b - 1] + 1 : 0;
            return T.less(pa, a, pb, b);
        });
        for (size_t x = 0; x < recs.size(); x++) ord[act[starts[g] + x]] = recs[x];
    });
}
