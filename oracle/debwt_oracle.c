/*
 * debwt_oracle.c -- CPU restatement of the deBWT hot path (see debwt_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: checker and reported CPU baseline, never the product path.
 *
 * Every function cites the reference lines (relative to /root/reference) whose arithmetic it
 * restates.  The restatement is in-memory (no temp files, no locks) and keeps the reference's
 * decomposition: k-mer count+sort -> node classification -> special-region module -> SP code ->
 * blue-block sort -> assembly.
 */
#include "debwt_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define SYM_HASH 4
#define SYM_DOLLAR 5

/* ------------------------------------------------------------------------------------------ */
/* helpers                                                                                    */

static void *xmalloc(size_t b) {
    void *p = malloc(b ? b : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory (%zu bytes)\n", b); abort(); }
    return p;
}
static void *xcalloc(size_t n, size_t s) {
    void *p = calloc(n ? n : 1, s ? s : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}

/* LSD radix sort of 64-bit keys with a 64-bit payload, 8-bit digits; passes whose digit is
 * constant over the input are skipped.  Stands in for mySort's bucket+qsort
 * (src/mySort.c:98-176): same result, an ascending order of the keys. */
static void radix_sort_kv(uint64_t *key, uint64_t *val, uint64_t m, int keybits) {
    if (m < 2) return;
    uint64_t *k2 = (uint64_t *)xmalloc(m * 8), *v2 = val ? (uint64_t *)xmalloc(m * 8) : NULL;
    int passes = (keybits + 7) / 8;
    uint64_t (*hist)[256] = (uint64_t(*)[256])xcalloc(passes, sizeof(uint64_t[256]));
    for (uint64_t i = 0; i < m; i++) {
        uint64_t x = key[i];
        for (int p = 0; p < passes; p++) hist[p][(x >> (8 * p)) & 255]++;
    }
    uint64_t *src = key, *dst = k2, *vs = val, *vd = v2;
    for (int p = 0; p < passes; p++) {
        uint64_t *h = hist[p];
        int skip = 0;
        for (int d = 0; d < 256; d++) if (h[d] == m) skip = 1;
        if (skip) continue;
        uint64_t sum = 0;
        for (int d = 0; d < 256; d++) { uint64_t c = h[d]; h[d] = sum; sum += c; }
        int sh = 8 * p;
        if (vs) {
            for (uint64_t i = 0; i < m; i++) {
                uint64_t x = src[i];
                uint64_t o = h[(x >> sh) & 255]++;
                dst[o] = x; vd[o] = vs[i];
            }
        } else {
            for (uint64_t i = 0; i < m; i++) {
                uint64_t x = src[i];
                dst[h[(x >> sh) & 255]++] = x;
            }
        }
        uint64_t *t = src; src = dst; dst = t;
        t = vs; vs = vd; vd = t;
    }
    if (src != key) {
        memcpy(key, src, m * 8);
        if (val) memcpy(val, vs, m * 8);
    }
    free(k2); free(v2); free(hist);
}

static inline int popcount4(unsigned m) { return (m & 1) + ((m >> 1) & 1) + ((m >> 2) & 1) + ((m >> 3) & 1); }

/* ------------------------------------------------------------------------------------------ */
/* text                                                                                       */

uint64_t orc_make_text(const char *seq, const uint64_t *reclen, uint64_t nrec, uint8_t *sym) {
    /* src/main.c:18-23 (trans[]), src/collect#$.c:41-45 (length check), :73-90 (layout) */
    uint64_t o = 0, s = 0;
    for (uint64_t r = 0; r < nrec; r++) {
        if (reclen[r] <= 32) return 0;
        for (uint64_t j = 0; j < reclen[r]; j++, s++) {
            uint8_t c;
            switch (seq[s]) {
                case 'A': case 'a': c = 0; break;
                case 'C': case 'c': c = 1; break;
                case 'G': case 'g': c = 2; break;
                case 'T': case 't': c = 3; break;
                default: return 0;
            }
            sym[o++] = c;
        }
        sym[o++] = (r + 1 == nrec) ? SYM_DOLLAR : SYM_HASH;
    }
    return o;
}

void orc_pack_text(const uint8_t *sym, uint64_t n, uint64_t *words) {
    /* src/collect#$.c:61-90: base j at bit 2*(31-(j&31)) of word j>>5, 'T' at separators,
     * then 32 'T'. */
    uint64_t total = n + 32, nw = (total + 31) >> 5;
    memset(words, 0, nw * 8);
    for (uint64_t j = 0; j < total; j++) {
        uint64_t c = (j < n && sym[j] < 4) ? sym[j] : 3;
        words[j >> 5] |= c << ((31 - (j & 31)) << 1);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a-1 + a-2: k-mer count and sort                                                            */

uint64_t orc_kmer_count(const uint8_t *sym, uint64_t n, int k, uint64_t *kmers, uint64_t *counts) {
    /* Jellyfish `count -m k` without -C then `dump -c -t` (src/kmercounting.sh:8,11): one entry
     * per distinct k-mer lying wholly inside a record.  mySort (src/mySort.c:54-83) packs each
     * left-aligned, 2 bits/base MSB first, and sorts ascending (:98-195). */
    uint64_t m = 0, w = 0, run = 0;
    uint64_t mask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1);
    uint64_t *inst = (uint64_t *)xmalloc(n * 8);
    for (uint64_t i = 0; i < n; i++) {
        if (sym[i] >= 4) { run = 0; w = 0; continue; }
        w = ((w << 2) | sym[i]) & mask;
        if (++run >= (uint64_t)k) inst[m++] = w;
    }
    radix_sort_kv(inst, NULL, m, 2 * k);
    uint64_t d = 0;
    for (uint64_t i = 0; i < m;) {
        uint64_t j = i + 1;
        while (j < m && inst[j] == inst[i]) j++;
        kmers[d] = inst[i] << (64 - 2 * k);
        counts[d] = j - i;
        d++;
        i = j;
    }
    free(inst);
    return d;
}

/* ------------------------------------------------------------------------------------------ */
/* suffix comparison under A<C<G<T<#<$                                                        */

static const uint8_t *g_cmp_sym;
static uint64_t g_cmp_n;

/* True suffix order: src/collect#$.c:253-311 (cmp) -- 32-symbol windows with '#' positions
 * ranked above 'T' (minusDimer :326-346), equal '#' skipped, '$' largest.  Symbol-at-a-time
 * here; the codes 0..5 already carry that order. */
static int suffix_cmp_pos(uint64_t a, uint64_t b) {
    if (a == b) return 0;
    const uint8_t *s = g_cmp_sym;
    uint64_t n = g_cmp_n;
    while (a < n && b < n) {
        uint8_t x = s[a], y = s[b];
        if (x != y) return x < y ? -1 : 1;
        a++; b++;
    }
    return a < n ? 1 : -1; /* unreachable for a != b: '$' is unique */
}
static int suffix_cmp_qsort(const void *pa, const void *pb) {
    return suffix_cmp_pos(*(const uint64_t *)pa, *(const uint64_t *)pb);
}

void orc_naive_bwt(const uint8_t *sym, uint64_t n, uint8_t *out) {
    uint64_t *sa = (uint64_t *)xmalloc(n * 8);
    for (uint64_t i = 0; i < n; i++) sa[i] = i;
    g_cmp_sym = sym; g_cmp_n = n;
    qsort(sa, n, 8, suffix_cmp_qsort);
    /* row of suffix 0 carries '$' (src/generateSP.c:584-605) */
    for (uint64_t r = 0; r < n; r++) out[r] = sa[r] ? sym[sa[r] - 1] : SYM_DOLLAR;
    free(sa);
}

/* ------------------------------------------------------------------------------------------ */
/* blue-block sort                                                                            */

static const uint8_t *g_sp;
static uint64_t g_sp_len;
static uint64_t g_cmp_calls;

/* cmpSP (src/sortBlue.c:109-173): order two blue entries by the SP-code suffix at entry>>4.
 * The reference walks 32-code windows of the 2-bit spCode and consults spSpecialIndex to rank
 * separator codes above 'T'; with one byte per SP symbol (0..5) the same order is a plain
 * symbol compare. */
static int sp_cmp_qsort(const void *pa, const void *pb) {
    uint64_t a = *(const uint64_t *)pa >> 4, b = *(const uint64_t *)pb >> 4;
    g_cmp_calls++;
    if (a == b) return 0;
    while (a < g_sp_len && b < g_sp_len) {
        uint8_t x = g_sp[a], y = g_sp[b];
        if (x != y) return x < y ? -1 : 1;
        a++; b++;
    }
    return a < g_sp_len ? 1 : -1;
}

/* ------------------------------------------------------------------------------------------ */
/* whole path                                                                                 */

typedef struct { uint64_t pos, key; } special_t;

static int special_cmp(const void *pa, const void *pb) {
    return suffix_cmp_pos(((const special_t *)pa)->pos, ((const special_t *)pb)->pos);
}
static int u64_cmp(const void *pa, const void *pb) {
    uint64_t a = *(const uint64_t *)pa, b = *(const uint64_t *)pb;
    return a < b ? -1 : (a > b);
}

int orc_build_bwt(const uint8_t *sym, uint64_t n, int k, int threads,
                  uint64_t *bwt_words, uint64_t *hash_rows, uint64_t *dollar_row, orc_stats *st) {
    return orc_build_bwt_ex(sym, n, k, threads, bwt_words, hash_rows, dollar_row, st, NULL, NULL);
}

int orc_build_bwt_ex(const uint8_t *sym, uint64_t n, int k, int threads,
                     uint64_t *bwt_words, uint64_t *hash_rows, uint64_t *dollar_row, orc_stats *st,
                     uint8_t *sp_out, uint64_t *red_out) {
    (void)threads;
    if (k < 12 || k > 32 || n < 34) return -1;          /* src/main.c:41-47 */
    const int K = k - 1;                                /* node length, KMER_LENGTH */
    orc_stats S; memset(&S, 0, sizeof S);
    S.n = n;

    /* separators: special[] of src/collect#$.c:86 */
    uint64_t nrec = 0;
    for (uint64_t i = 0; i < n; i++) if (sym[i] >= 4) nrec++;
    if (!nrec || sym[n - 1] != SYM_DOLLAR) return -1;
    S.nrec = nrec;
    uint64_t *sep = (uint64_t *)xmalloc(nrec * 8);
    { uint64_t r = 0; for (uint64_t i = 0; i < n; i++) if (sym[i] >= 4) sep[r++] = i; }

    /* ---- node instances: every position whose K-window holds no separator ("main module",
     * src/generateSP.c:534-541 distance>=KMER_LENGTH).  Grouping them by window gives what
     * mergeKmer derives from the sorted edge list + head#/tail# (src/INandOut.c:258-346). */
    uint64_t M = n - nrec * (uint64_t)K;
    uint64_t *nkey = (uint64_t *)xmalloc(M * 8), *npos = (uint64_t *)xmalloc(M * 8);
    {
        uint64_t m = 0, w = 0, run = 0;
        uint64_t mask = (1ull << (2 * K)) - 1;
        for (uint64_t i = 0; i < n; i++) {
            if (sym[i] >= 4) { run = 0; w = 0; continue; }
            w = ((w << 2) | sym[i]) & mask;
            if (++run >= (uint64_t)K) { nkey[m] = w; npos[m] = i + 1 - K; m++; }
        }
        if (m != M) { free(sep); free(nkey); free(npos); return -2; }
    }
    radix_sort_kv(nkey, npos, M, 2 * K);

    /* distinct k-mers = distinct (node, successor base) pairs: only counted for the stats */
    {
        uint64_t *tmp = (uint64_t *)xmalloc(n * 8), *cnt = (uint64_t *)xmalloc(n * 8);
        S.distinct_kmers = orc_kmer_count(sym, n, k, tmp, cnt);
        free(tmp); free(cnt);
    }

    /* per-position flags: bit0 multi-out, bit1 multi-in */
    uint8_t *pflag = (uint8_t *)xcalloc(n, 1);

    /* node table */
    uint64_t Dn = 0;
    for (uint64_t i = 0; i < M; i++) if (i == 0 || nkey[i] != nkey[i - 1]) Dn++;
    uint64_t *nd_key = (uint64_t *)xmalloc(Dn * 8), *nd_start = (uint64_t *)xmalloc((Dn + 1) * 8);
    uint8_t *nd_flag = (uint8_t *)xmalloc(Dn), *nd_pred = (uint8_t *)xmalloc(Dn);
    {
        uint64_t t = 0;
        for (uint64_t i = 0; i < M;) {
            uint64_t j = i;
            unsigned predmask = 0, succmask = 0; int head = 0, tail = 0;
            while (j < M && nkey[j] == nkey[i]) {
                uint64_t p = npos[j];
                /* predecessor: src/generateSP.c:584-605 */
                if (p == 0 || sym[p - 1] >= 4) head = 1; else predmask |= 1u << sym[p - 1];
                /* successor: out-edges (src/INandOut.c:271-281) or tail# hit (:262-269) */
                if (sym[p + K] >= 4) tail = 1; else succmask |= 1u << sym[p + K];
                j++;
            }
            int mo = (popcount4(succmask) >= 2) || tail;   /* src/INandOut.c:260-281 */
            int mi = (popcount4(predmask) >= 2) || head;   /* src/INandOut.c:282-343 */
            nd_key[t] = nkey[i]; nd_start[t] = i; nd_flag[t] = (uint8_t)(mo | (mi << 1));
            unsigned pm = predmask; uint8_t single = 0;
            while (pm > 1) { pm >>= 1; single++; }
            nd_pred[t] = single;                           /* bwtSingle, src/INandOut.c:296-339 */
            if (mo | mi) {
                S.red_capacity++;
                for (uint64_t q = i; q < j; q++) pflag[npos[q]] = nd_flag[t];
            }
            if (mi) { S.blue_bound_num++; S.blue_capacity += j - i; }
            t++; i = j;
        }
        nd_start[Dn] = M;
    }
    S.case3num = 2 * S.blue_bound_num;
    if (red_out) {
        uint64_t r = 0;
        for (uint64_t t = 0; t < Dn; t++) if (nd_flag[t]) red_out[r++] = (nd_key[t] << 2) | nd_flag[t];
    }

    /* ---- special-region module: the K suffixes per record that start <= K-1 before a separator
     * (src/collect#$.c:118-157), sorted by true suffix order (:253-311). */
    uint64_t NS = nrec * (uint64_t)K;
    special_t *sp = (special_t *)xmalloc(NS * sizeof(special_t));
    {
        uint64_t m = 0;
        for (uint64_t r = 0; r < nrec; r++)
            for (int d = K - 1; d >= 0; d--) {
                uint64_t p = sep[r] - (uint64_t)d;
                /* key: bases up to the separator then 'T' padding (src/collect#$.c:428-446) */
                uint64_t key = 0;
                for (int j = 0; j < K; j++) key = (key << 2) | (uint64_t)(j < d ? sym[p + j] : 3);
                sp[m].pos = p; sp[m].key = key; m++;
            }
    }
    g_cmp_sym = sym; g_cmp_n = n;
    qsort(sp, NS, sizeof(special_t), special_cmp);

    /* special branches (src/collect#$.c:534-598, compareI :618-634): specials whose K-window is
     * identical (same symbols, separator of the same kind at the same offset); a group of >= 2
     * whose symbols at offset K differ marks all members multi-out. */
    for (uint64_t i = 0; i < NS;) {
        uint64_t j = i + 1;
        while (j < NS) {
            int same = 1;
            for (int t = 0; t < K && same; t++)
                if (sym[sp[i].pos + t] != sym[sp[j].pos + t]) same = 0;
            if (!same) break;
            j++;
        }
        if (j - i >= 2) {
            int differ = 0;
            /* a '$' window is unique, so pos+K stays inside the text for every group of >= 2 */
            for (uint64_t q = i + 1; q < j; q++)
                if (sym[sp[q].pos + K] != sym[sp[i].pos + K]) differ = 1;
            if (differ)
                for (uint64_t q = i; q < j; q++) { pflag[sp[q].pos] |= 1; S.special_branch_num++; }
        }
        i = j;
    }

    /* ---- SP code (src/generateSP.c:626-660): at every multi-out position the symbol K ahead;
     * the separator itself when the window is immediately followed by it (:630-641). */
    uint32_t *spidx = (uint32_t *)xmalloc(n * 4);
    uint64_t spl = 0;
    for (uint64_t i = 0; i < n; i++) { spidx[i] = (uint32_t)spl; if (pflag[i] & 1) spl++; }
    if (spl >> 32) return -3;
    S.sp_len = spl;
    uint8_t *spc = (uint8_t *)xmalloc(spl + 1);
    { uint64_t s = 0; for (uint64_t i = 0; i < n; i++) if (pflag[i] & 1) spc[s++] = sym[i + K]; }
    if (sp_out) memcpy(sp_out, spc, spl);

    /* ---- rows (src/INandOut.c:344-346,419-439; src/insertCase3.c:56-104) */
    uint8_t *row = (uint8_t *)xmalloc(n);
    uint64_t r = 0, si = 0;
    uint64_t blue_cap = 0;
    for (uint64_t t = 0; t < Dn; t++) {
        uint64_t b = nd_start[t + 1] - nd_start[t];
        if ((nd_flag[t] & 2) && b > blue_cap) blue_cap = b;
    }
    uint64_t *blue = (uint64_t *)xmalloc((blue_cap ? blue_cap : 1) * 8);
    g_sp = spc; g_sp_len = spl; g_cmp_calls = 0;
    for (uint64_t t = 0; t < Dn; t++) {
        while (si < NS && sp[si].key < nd_key[t]) { row[r++] = sym[sp[si].pos - 1]; si++; }
        uint64_t lo = nd_start[t], hi = nd_start[t + 1];
        if (nd_flag[t] & 2) {
            /* blue entries pred | spIndex<<4 (src/generateSP.c:662-680), sorted per block with the
             * single-character early-out of myQsort (src/sortBlue.c:192-219) */
            uint64_t m = 0; unsigned seen = 0;
            for (uint64_t q = lo; q < hi; q++) {
                uint64_t p = npos[q];
                uint64_t c = (p == 0) ? SYM_DOLLAR : sym[p - 1];
                blue[m++] = c | ((uint64_t)spidx[p] << 4);
                seen |= 1u << c;
            }
            if (seen & (seen - 1)) qsort(blue, m, 8, sp_cmp_qsort);
            for (uint64_t q = 0; q < m; q++) row[r++] = (uint8_t)(blue[q] & 15);
        } else {
            for (uint64_t q = lo; q < hi; q++) row[r++] = nd_pred[t];   /* case 2 */
        }
        while (si < NS && sp[si].key == nd_key[t]) { row[r++] = sym[sp[si].pos - 1]; si++; }
    }
    while (si < NS) { row[r++] = sym[sp[si].pos - 1]; si++; }
    S.cmp_calls = g_cmp_calls;
    int rc = (r == n) ? 0 : -4;

    /* ---- on-disk layout (src/insertCase3.c:75-97,115-131) */
    if (rc == 0) {
        uint64_t nw = (n + 31) >> 5, h = 0;
        memset(bwt_words, 0, nw * 8);
        for (uint64_t j = 0; j < n; j++) {
            uint64_t c = row[j];
            if (c == SYM_HASH) { hash_rows[h++] = j; c = 3; }
            else if (c == SYM_DOLLAR) { *dollar_row = j; c = 3; }
            bwt_words[j >> 5] |= c << ((31 - (j & 31)) << 1);
        }
        if (h != nrec - 1) rc = -5;
    }
    if (st) *st = S;
    free(sep); free(nkey); free(npos); free(pflag); free(nd_key); free(nd_start); free(nd_flag);
    free(nd_pred); free(sp); free(spidx); free(spc); free(row); free(blue);
    (void)u64_cmp;
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* verification helpers                                                                       */

void orc_unpack_bwt(const uint64_t *bwt_words, uint64_t n, const uint64_t *hash_rows, uint64_t nrec,
                    uint64_t dollar_row, uint8_t *out) {
    for (uint64_t j = 0; j < n; j++) out[j] = (uint8_t)((bwt_words[j >> 5] >> ((31 - (j & 31)) << 1)) & 3);
    for (uint64_t h = 0; h + 1 < nrec; h++) out[hash_rows[h]] = SYM_HASH;
    out[dollar_row] = SYM_DOLLAR;
}

int orc_inverse_bwt(const uint64_t *bwt_words, uint64_t n, const uint64_t *hash_rows, uint64_t nrec,
                    uint64_t dollar_row, uint8_t *sym_out) {
    /* LF(i) = C[c] + occ(c, i); the j-th '#' row maps to row n-nrec+j (src/LFsearch.c:131),
     * the '$' row to the last row.  The reference samples occ every 32 rows
     * (src/insertCase3.c:141-194); a full table is used here. */
    uint8_t *L = (uint8_t *)xmalloc(n);
    orc_unpack_bwt(bwt_words, n, hash_rows, nrec, dollar_row, L);
    uint64_t C[7] = {0}, cnt[6] = {0};
    for (uint64_t i = 0; i < n; i++) cnt[L[i]]++;
    for (int c = 0; c < 6; c++) C[c + 1] = C[c] + cnt[c];
    uint64_t *lf = (uint64_t *)xmalloc(n * 8);
    uint64_t seen[6] = {0};
    for (uint64_t i = 0; i < n; i++) { uint8_t c = L[i]; lf[i] = C[c] + seen[c]++; }
    uint64_t rowi = n - 1, steps = 0;
    sym_out[n - 1] = SYM_DOLLAR;
    int rc = 0;
    for (uint64_t p = n - 1; p > 0; p--) {
        uint8_t c = L[rowi];
        if (c == SYM_DOLLAR) { rc = -1; break; }
        sym_out[p - 1] = c;
        rowi = lf[rowi];
        steps++;
    }
    if (rc == 0 && L[rowi] != SYM_DOLLAR) rc = -2;
    free(L); free(lf);
    return rc;
}
