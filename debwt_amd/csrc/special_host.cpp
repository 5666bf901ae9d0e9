// special_host.cpp -- see special_host.h.  Restates SURVEY 8a items 3-4 on packed windows.
#include "special_host.h"

#include <algorithm>

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
    bool less(uint64_t a, uint64_t b) const {
        uint64_t ia = first_sep_at_or_after(a), ib = first_sep_at_or_after(b);
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
    bool same_window(uint64_t a, uint64_t b, int K) const {
        uint64_t ia = first_sep_at_or_after(a), ib = first_sep_at_or_after(b);
        uint64_t da = sep[ia] - a, db = sep[ib] - b;
        if (da != db || (ia == nrec - 1) != (ib == nrec - 1)) return false;
        for (int t = 0; t < K; t++) {
            if ((uint64_t)t == da) continue;
            if (base(a + t) != base(b + t)) return false;
        }
        return true;
    }
};

}  // namespace

void build_special_tables(const uint64_t *words, uint64_t n, const uint64_t *sep, uint64_t nrec, int K,
                          SpecialTables *out) {
    Text T{words, n, sep, nrec};
    const uint64_t NS = nrec * (uint64_t)K;
    const uint64_t node_mask = (1ull << (2 * K)) - 1;
    // The T-padded key is monotone in the true suffix order (a separator ranks above every base and the
    // padding is the largest base), so ordering by key first and by the full suffix comparison only inside
    // runs of equal keys gives exactly the order of the reference's qsort (src/collect#$.c:118-157,253-311)
    // at a fraction of its comparisons.
    struct Item { uint64_t key, pos; };
    std::vector<Item> items(NS);
    {
        uint64_t m = 0;
        for (uint64_t r = 0; r < nrec; r++)
            for (int d = K - 1; d >= 0; d--) {
                uint64_t p = sep[r] - (uint64_t)d;
                // key: the d bases, then 'T' up to K symbols (src/collect#$.c:428-446)
                uint64_t win = d ? (T.window(p) >> (64 - 2 * d)) : 0;
                uint64_t pad = (1ull << (2 * (K - d))) - 1;
                items[m].key = ((win << (2 * (K - d))) | pad) & node_mask;
                items[m].pos = p;
                m++;
            }
    }
    std::sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.key < b.key; });
    for (uint64_t i = 0; i < NS;) {
        uint64_t j = i + 1;
        while (j < NS && items[j].key == items[i].key) j++;
        if (j - i > 1)
            std::sort(items.begin() + i, items.begin() + j,
                      [&](const Item &a, const Item &b) { return a.pos != b.pos && T.less(a.pos, b.pos); });
        i = j;
    }
    std::vector<uint64_t> order(NS);
    out->key.resize(NS);
    out->chr.resize(NS);
    for (uint64_t s = 0; s < NS; s++) {
        order[s] = items[s].pos;
        out->key[s] = items[s].key;
        out->chr[s] = (uint8_t)T.base(items[s].pos - 1);          // always a base: records are > K long
    }
    out->pos = order;

    // special branches (src/collect#$.c:534-598)
    out->branch.clear();
    for (uint64_t i = 0; i < NS;) {
        uint64_t j = i + 1;
        while (j < NS && T.same_window(order[i], order[j], K)) j++;
        if (j - i >= 2) {
            bool differ = false;
            for (uint64_t q = i + 1; q < j; q++)
                if (T.base(order[q] + K) != T.base(order[i] + K)) differ = true;
            if (differ)
                for (uint64_t q = i; q < j; q++) out->branch.push_back(order[q]);
        }
        i = j;
    }
    std::sort(out->branch.begin(), out->branch.end());

    // head# / head$ and tail# nodes (src/collect#$.c:468-533)
    out->head_keys.resize(nrec);
    out->tail_facts.resize(nrec);
    for (uint64_t r = 0; r < nrec; r++) {
        uint64_t start = r ? sep[r - 1] + 1 : 0;
        out->head_keys[r] = ((T.window(start) >> (64 - 2 * K)) << 2) | 3ull;
        out->tail_facts[r] = ((T.window(sep[r] - (uint64_t)K) >> (64 - 2 * K)) << 2) | 1ull;
    }
    std::sort(out->head_keys.begin(), out->head_keys.end());
}
