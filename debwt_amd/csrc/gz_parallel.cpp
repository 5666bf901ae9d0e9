// gz_parallel.cpp -- a ONE-MEMBER gzip file inflated by all host threads (the reference reads gzip through
// zlib's gzread on one thread, src/collect#$.c:26,34-37 / src/kseq.h).
//
// A deflate stream cannot be entered in the middle for two reasons: block boundaries are not marked (they fall on any
// bit), and a block may copy from the 32 KB of output before it.  Both are dealt with the way pugz / rapidgzip do:
//   1. the compressed bytes are cut into chunks; for every chunk but the first a block start is FOUND by trial: at each
//      bit offset a block header is parsed (stored: LEN / NLEN complement; dynamic: the code-length codes must describe
//      complete, not over-subscribed Huffman codes with an end-of-block symbol) and the block plus the header of the one
//      behind it decoded with the demand that every literal is text -- FASTA and FASTQ are;
//   2. a chunk is decoded from its block start with an UNKNOWN window: a small decoder of our own (RFC 1951, canonical
//      codes decoded the counting way) writes 16-bit symbols, where a copy that reaches into the unknown window leaves a
//      marker (which window byte) instead of a byte.  DNA text compresses by its Huffman codes, not by long copies, so
//      after a few blocks the last 32 KB of output hold no marker: from the next block boundary on the byte decoder of
//      fast_inflate.h continues (it starts at any bit with that 32 KB in front of its output; until round 6 this was
//      zlib's inflate by the recipe of its examples/zran.c, at under half the rate) up to the bit where the next chunk
//      starts, which must be one of the block boundaries it passes;
//   3. the chunks' marker prefixes are resolved in order with the 32 KB before them (a few hundred KB per chunk), the
//      pieces are copied into one buffer by all threads, and the CRC32 and length of the whole (crc32_combine of the
//      chunks') must equal the gzip trailer.
// Anything that does not fit -- several members, no text, a block start that cannot be found, a chunk that does not end
// where the next begins, a CRC that differs -- makes the function return 1 and the caller inflates serially with gzread,
// as before.  Nothing here is trusted without the trailer check.
#include "gz_parallel.h"
#include "fast_inflate.h"

#include <zlib.h>
#include <malloc.h>
#include <pthread.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace {

typedef uint16_t sym_t;                       // < 256: a byte; >= MARK: byte (value - MARK) of the 32 KB before the chunk
constexpr sym_t MARK = 0x8000;
constexpr size_t WIN = 32768;

// the symbols of a piece's head: a plain growing array that is NOT cleared when it grows (std::vector::resize zero-filled
// the slack behind every deflate block: O(n) per block)
struct SymVec {
    sym_t *p = nullptr; size_t n = 0, cap = 0;
    size_t lead = 0;                           // symbols kept in front of p[0] in the same allocation (a piece's window of markers)
    SymVec() = default;
    SymVec(const SymVec &) = delete;
    SymVec &operator=(const SymVec &) = delete;
    SymVec(SymVec &&o) noexcept : p(o.p), n(o.n), cap(o.cap), lead(o.lead) { o.p = nullptr; o.n = o.cap = 0; }
    ~SymVec() { release(); }
    size_t size() const { return n; }
    sym_t &operator[](size_t i) { return p[i]; }
    const sym_t &operator[](size_t i) const { return p[i]; }
    void clear() { n = 0; }
    void release() { if (p) free(p - lead); p = nullptr; n = cap = 0; }
    bool reserve(size_t want) {                // false: out of memory (the old array stays)
        if (want <= cap && p) return true;
        sym_t *q = (sym_t *)big_malloc((want + lead) * sizeof(sym_t));
        if (!q) return false;
        if (p) { memcpy(q, p - lead, (n + lead) * sizeof(sym_t)); free(p - lead); }
        p = q + lead; cap = want;
        return true;
    }
    bool push_back(sym_t v) { if (n == cap && !reserve(cap + cap / 2 + ((size_t)1 << 16))) return false; p[n++] = v; return true; }
};

struct Bits {                                 // LSB-first bit reader over the whole file; pos = absolute bit offset.  The
    const unsigned char *z; size_t nbits, pos; // deflate data is followed by the 8-byte gzip trailer, so the 8-byte loads
    bool over = false;                         // below never leave the file
    inline uint64_t peek() const {            // at least 56 bits from pos on
        uint64_t v;
        memcpy(&v, z + (pos >> 3), 8);
        return v >> (pos & 7);
    }
    inline unsigned get(int n) {              // n <= 16
        if (pos + (size_t)n > nbits) { over = true; return 0; }
        const unsigned v = (unsigned)(peek() & ((1u << n) - 1u));
        pos += (size_t)n;
        return v;
    }
};

constexpr int FB = 10;                        // codes of up to FB bits are decoded by one table look-up
struct Huff {
    uint16_t count[16]; uint16_t symbol[288];
    uint16_t fast[1 << FB];                   // by the next FB input bits: symbol | length << 9; 0: a longer code (bit by bit)
    bool has_fast = false;
};

// canonical code from lengths; returns 0 complete, > 0 incomplete (bits left over), < 0 over-subscribed
int build(Huff &h, const uint8_t *len, int n, bool fast = false) {
    memset(h.count, 0, sizeof h.count);
    h.has_fast = false;
    for (int i = 0; i < n; i++) h.count[len[i]]++;
    if (h.count[0] == n) return 0;            // no codes (complete in the sense that nothing can be decoded)
    int left = 1;
    for (int l = 1; l <= 15; l++) { left <<= 1; left -= h.count[l]; if (left < 0) return left; }
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + h.count[l];
    for (int i = 0; i < n; i++) if (len[i]) h.symbol[offs[len[i]]++] = (uint16_t)i;
    if (fast) {
        memset(h.fast, 0, sizeof h.fast);
        unsigned next[16], code = 0;
        for (int l = 1; l <= 15; l++) { code = (code + h.count[l - 1] * (l > 1 ? 1u : 0u)) << 1; next[l] = code; }
        for (int i = 0; i < n; i++) {
            const int l = len[i];
            if (!l || l > FB) { if (l) next[l]++; continue; }
            unsigned c = next[l]++, r = 0;
            for (int k = 0; k < l; k++) r |= ((c >> k) & 1u) << (l - 1 - k);     // codes are sent most significant bit first
            for (unsigned e = r; e < (1u << FB); e += 1u << l) h.fast[e] = (uint16_t)(i | (l << 9));
        }
        h.has_fast = true;
    }
    return left;
}
inline int decode(Bits &b, const Huff &h) {   // -1: not a code / out of input
    if (h.has_fast) {
        const unsigned e = h.fast[b.peek() & ((1u << FB) - 1u)];
        if (e) {
            b.pos += e >> 9;
            if (b.pos > b.nbits) { b.over = true; return -1; }
            return (int)(e & 511u);
        }
    }
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= 15; l++) {
        code |= (int)b.get(1);
        if (b.over) return -1;
        const int c = h.count[l];
        if (code - c < first) return h.symbol[index + (code - first)];
        index += c; first += c; first <<= 1; code <<= 1;
    }
    return -1;
}

const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
                            8193, 12289, 16385, 24577};
const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline bool is_text(unsigned c) { return c == '\n' || c == '\r' || c == '\t' || (c >= 32 && c < 127); }

struct Fixed {
    Huff lit, dist;
    Fixed() {
        uint8_t l[288];
        for (int i = 0; i < 144; i++) l[i] = 8;
        for (int i = 144; i < 256; i++) l[i] = 9;
        for (int i = 256; i < 280; i++) l[i] = 7;
        for (int i = 280; i < 288; i++) l[i] = 8;
        build(lit, l, 288, true);
        for (int i = 0; i < 30; i++) l[i] = 5;
        build(dist, l, 30, true);
    }
};
const Fixed FIXED;

// One deflate block from b.pos on, appended to `out` as 16-bit symbols (copies that reach before out[0] leave markers).
// strict: literals must be text (the block-start search).  Returns 0 ok (b.pos behind the block, *last = BFINAL), -1 not
// a valid block / out of input, -2 out of memory.  max_out bounds the growth of `out` (search: a wrong start must not
// decode for ever).
int block(Bits &b, SymVec &out, bool strict, int *last, size_t max_out) {
    *last = (int)b.get(1);
    const unsigned type = b.get(2);
    if (b.over || type == 3) return -1;
    if (type == 0) {
        b.pos = (b.pos + 7) & ~(size_t)7;
        const unsigned len = b.get(16), nlen = b.get(16);
        if (b.over || (len ^ nlen) != 0xFFFFu) return -1;
        if (b.pos + 8 * (size_t)len > b.nbits) return -1;
        if (out.size() + len > max_out) return -1;
        const unsigned char *p = b.z + (b.pos >> 3);
        for (unsigned i = 0; i < len; i++) {
            if (strict && !is_text(p[i])) return -1;
            if (!out.push_back(p[i])) return -2;
        }
        b.pos += 8 * (size_t)len;
        return 0;
    }
    Huff lit, dist;
    const Huff *L = &FIXED.lit, *D = &FIXED.dist;
    if (type == 2) {
        const int nlen = (int)b.get(5) + 257, ndist = (int)b.get(5) + 1, ncode = (int)b.get(4) + 4;
        if (b.over || nlen > 286 || ndist > 30) return -1;
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t lengths[320];
        memset(lengths, 0, sizeof lengths);
        for (int i = 0; i < ncode; i++) lengths[order[i]] = (uint8_t)b.get(3);
        if (b.over) return -1;
        Huff cl;
        if (build(cl, lengths, 19) != 0) return -1;          // the code-length code must be complete
        uint8_t ll[320];
        int idx = 0;
        while (idx < nlen + ndist) {
            const int s = decode(b, cl);
            if (s < 0) return -1;
            if (s < 16) ll[idx++] = (uint8_t)s;
            else {
                int prev = 0, rep;
                if (s == 16) { if (idx == 0) return -1; prev = ll[idx - 1]; rep = 3 + (int)b.get(2); }
                else if (s == 17) rep = 3 + (int)b.get(3);
                else rep = 11 + (int)b.get(7);
                if (b.over || idx + rep > nlen + ndist) return -1;
                while (rep--) ll[idx++] = (uint8_t)prev;
            }
        }
        if (ll[256] == 0) return -1;                          // no end-of-block code
        int e = build(lit, ll, nlen, true);
        if (e < 0 || (e > 0 && nlen - lit.count[0] != 1)) return -1;   // incomplete only for a single code
        e = build(dist, ll + nlen, ndist, true);
        if (e < 0 || (e > 0 && ndist - dist.count[0] != 1)) return -1;
        L = &lit; D = &dist;
    }
    size_t n = out.size();
    if (out.cap < n + 65536 + 258 && !out.reserve(n + n / 2 + ((size_t)1 << 20))) return -2;
    sym_t *v = out.p;                                         // written through a pointer; out.n is set on every way out
    size_t cap = out.cap;
    int rc = -1;
    for (;;) {
        if (n + 258 > cap) {
            if (n + 258 > max_out) break;
            if (!out.reserve(n + n / 2 + ((size_t)1 << 20))) { rc = -2; break; }
            v = out.p; cap = out.cap;
        }
        const int s = decode(b, *L);
        if (s < 0) break;
        if (s < 256) {
            if (strict && !is_text((unsigned)s)) break;
            v[n++] = (sym_t)s;
        } else if (s == 256) {
            rc = n <= max_out ? 0 : -1;
            break;
        } else {
            if (s > 285) break;
            const unsigned len = LBASE[s - 257] + b.get(LEXT[s - 257]);
            const int ds = decode(b, *D);
            if (ds < 0 || ds > 29) break;
            const unsigned d = DBASE[ds] + b.get(DEXT[ds]);
            if (b.over) break;
            if (n >= d) {
                const sym_t *src = v + n - d;
                for (unsigned i = 0; i < len; i++) v[n + i] = src[i];
            } else {
                for (unsigned i = 0; i < len; i++) {
                    const long long src = (long long)(n + i) - (long long)d;
                    v[n + i] = src >= 0 ? v[(size_t)src] : (sym_t)(MARK + (sym_t)((long long)WIN + src));
                }
            }
            n += len;
        }
    }
    out.n = n;
    return rc;
}

// first bit offset >= from (< to) at which a non-final DYNAMIC block starts that decodes as text (at least 1024 symbols)
// and is followed by a block that decodes as text too -- or nbits (none).  Only dynamic blocks are taken for a start: their
// header is its own proof (the code-length code and both codes it describes must be complete), while any bit string is
// a valid run of fixed-code symbols and a stored block proves itself with 16 bits; gzip writes dynamic blocks for text of
// any size worth cutting into pieces.
size_t find_block(const unsigned char *z, size_t nbits, size_t from, size_t to) {
    SymVec tmp, t2;
    for (size_t p = from; p < to; p++) {
        const unsigned hdr = (z[p >> 3] >> (p & 7)) | ((p >> 3) + 1 < (nbits + 7) >> 3 ? (unsigned)z[(p >> 3) + 1] << (8 - (p & 7)) : 0u);
        if ((hdr & 7u) != 4u) continue;                       // BFINAL = 0, BTYPE = 2 (bits LSB first: 0, then 0 1)
        Bits b{z, nbits, p};
        tmp.clear();
        int last = 0;
        if (block(b, tmp, true, &last, (size_t)1 << 22) != 0 || last || tmp.size() < 1024) continue;
        Bits c = b;
        t2.clear();
        int last2 = 0;
        if (block(c, t2, true, &last2, (size_t)1 << 22) != 0) continue;
        return p;
    }
    return nbits;
}

struct Piece {
    size_t start_bit = 0, end_bit = 0;       // decoded from / up to (the next piece's start_bit, or the end of the final block)
    SymVec head;                              // symbols decoded with the unknown window (may hold markers)
    char *alloc = nullptr;                    // WIN bytes (the window the body starts from) and the body behind them
    char *body = nullptr; size_t body_len = 0, body_cap = 0;   // what the fast decoder decoded behind the head (final bytes)
    std::vector<char> head_bytes;             // head, resolved
    bool final_block = false;
    int status = 0;                           // 0 ok, 1 does not fit (serial fallback), -1 damaged / out of memory
};

// room for `need` more body bytes.  `first`: the capacity to start with (fast_part sizes it from the piece's compressed bytes, so
// that the body is allocated once -- reserved, not touched -- and never moved)
bool grow(Piece &p, size_t need, size_t first = (size_t)8 << 20) {
    if (p.body_len + need <= p.body_cap) return true;
    size_t cap = p.body_cap ? p.body_cap : first;
    while (cap < p.body_len + need) cap *= 2;
    char *nb = (char *)big_malloc(WIN + cap);
    if (!nb) return false;
    if (p.alloc) { memcpy(nb, p.alloc, WIN + p.body_len); free(p.alloc); }
    p.alloc = nb; p.body = nb + WIN; p.body_cap = cap;
    return true;
}

// the fast decoder (fast_inflate.h) continues at a block boundary `bit` with the 32 KB `dict` before it, until it reaches
// `stop_bit` (a boundary) or the end of the final block
void fast_part(const unsigned char *z, size_t zlen, size_t bit, size_t stop_bit, const unsigned char *dict, size_t dictlen, Piece &p) {
    std::unique_ptr<fastinflate::Decoder> d(new (std::nothrow) fastinflate::Decoder());
    // DNA text deflates to 1/3.2 .. 1/4.5: five times the piece's compressed bytes hold it
    const size_t comp = ((stop_bit == ~(size_t)0 ? 8 * zlen : stop_bit) - bit) / 8;
    if (!d || !grow(p, (size_t)1 << 20, comp * 5 + ((size_t)1 << 20))) { p.status = -1; return; }
    if (dictlen) memcpy(p.body - dictlen, dict, dictlen);      // (dictlen <= WIN: the bytes in front of the body)
    d->start(z, zlen, bit);
    for (;;) {
        if (!grow(p, (size_t)1 << 20)) { p.status = -1; return; }
        size_t pos = p.body_len;
        const int r = d->run((uint8_t *)p.body, dictlen, &pos, p.body_cap, stop_bit);
        p.body_len = pos;
        if (r == fastinflate::FI_NEED_OUTPUT) continue;
        if (r == fastinflate::FI_DONE) { p.final_block = true; p.end_bit = d->bitpos; }
        else if (r == fastinflate::FI_STOPPED) p.end_bit = d->bitpos;
        else p.status = 1;                                      // not deflate from here, the next piece does not start on a boundary of this one, or the input ends inside a block
        return;
    }
}

void decode_piece(const unsigned char *z, size_t zlen, size_t stop_bit, bool first, Piece &p) {
    if (first) { fast_part(z, zlen, p.start_bit, stop_bit, nullptr, 0, p); return; }
    // with an unknown window: 16-bit symbols behind WIN markers (marker i = byte i of the 32 KB before the piece), block by
    // block until the last WIN symbols hold no marker
    std::unique_ptr<fastinflate::Decoder> d(new (std::nothrow) fastinflate::Decoder());
    p.head.lead = WIN;
    // (room for the whole piece -- five symbols per compressed byte -- is reserved, not touched: the head of gzip -6 text ends
    // after 0.5 - 1 M symbols, that of gzip -1 text never does, and neither is moved while it grows)
    const size_t comp = ((stop_bit == ~(size_t)0 ? 8 * zlen : stop_bit) - p.start_bit) / 8;
    if (!d || !p.head.reserve(std::min(comp * 5 + ((size_t)1 << 20), (size_t)1 << 26))) { p.status = -1; return; }
    for (size_t i = 0; i < WIN; i++) p.head.p[(ptrdiff_t)i - (ptrdiff_t)WIN] = (sym_t)(MARK + i);
    d->start(z, zlen, p.start_bit);
    for (;;) {
        const int r = d->run<sym_t>(p.head.p, WIN, &p.head.n, p.head.cap, stop_bit, true);
        if (r == fastinflate::FI_NEED_OUTPUT) {
            // (a text whose copies keep reaching back -- tandem repeats, gzip -1 -- never sheds its markers: a piece is at most
            // 16 MB of the file and stays below 2^26 symbols; beyond that the file is left to the serial path)
            if (p.head.cap >= ((size_t)1 << 26)) { p.status = 1; return; }
            if (!p.head.reserve(std::min(p.head.cap * 2, (size_t)1 << 26))) { p.status = -1; return; }
            continue;
        }
        if (r == fastinflate::FI_DONE) { p.final_block = true; p.end_bit = d->bitpos; return; }
        if (r != fastinflate::FI_STOPPED) { p.status = 1; return; }
        if (d->bitpos == stop_bit) { p.end_bit = d->bitpos; return; }
        size_t clean = 0;                                         // marker-free symbols at the end, as far as it matters
        for (size_t i = p.head.n; i > 0 && clean < WIN && p.head.p[i - 1] < MARK; i--) clean++;
        if (clean >= WIN) break;                                  // a known window: the byte decoder from here
    }
    unsigned char dict[WIN];
    for (size_t i = 0; i < WIN; i++) dict[i] = (unsigned char)p.head[p.head.size() - WIN + i];
    fast_part(z, zlen, d->bitpos, stop_bit, dict, WIN, p);
}

}  // namespace

namespace {
// A large block goes back to the kernel in slices of 64 MB by madvise(MADV_DONTNEED) first -- that takes the address space's
// lock for READING, as a page fault does, and does the work (this host clears what it takes back: 50 ms per GB) -- and the
// munmap inside free() then finds nothing left to do under the lock it takes for WRITING.  One munmap of the 3 GB text stopped
// every page fault and every copy to the device of the process for 0.16 s (cli/deBWT: the build behind a gzip parse took 0.38 s
// instead of 0.17 s).
static void give_back(void *p) {
    const size_t sz = malloc_usable_size(p);
    if (sz >= ((size_t)16 << 20) && !getenv("DEBWT_RELEASE_WHOLE")) {
        const uintptr_t a = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, e = ((uintptr_t)p + sz) & ~(uintptr_t)4095;
        for (uintptr_t x = a; x < e; x += (uintptr_t)64 << 20)
            (void)madvise((void *)x, (size_t)std::min<uintptr_t>((uintptr_t)64 << 20, e - x), MADV_DONTNEED);
    }
    free(p);
}

struct Releaser {                             // one thread, started at the first hand-over, that free()s what is queued
    std::mutex m;
    std::condition_variable *cv = new std::condition_variable();   // (on the heap for the same reason: the child leaves the one
    std::deque<void *> q;                                          //  the parent's thread waits in alone -- destroying it would wait for that thread)
    std::thread *th = nullptr;                // (on the heap: a process forked off this one must not destroy -- join -- a thread it has not got)
    pid_t owner = 0;                          // the process the thread runs in
    bool stop = false;
    int held = 0;                             // > 0: nothing is released for now (the caller's threads are faulting pages in)
    void loop() {
        for (;;) {
            std::unique_lock<std::mutex> g(m);
            cv->wait(g, [&] { return stop || (!q.empty() && held <= 0); });
            if (q.empty()) return;                            // (stop: what was queued has been released first)
            void *p = q.front();
            q.pop_front();
            g.unlock();
            give_back(p);
        }
    }
    bool give(const std::vector<void *> &v) {                 // false: no thread to be had
        std::lock_guard<std::mutex> g(m);
        if (!th) {
            try { th = new std::thread([this] { loop(); }); } catch (...) { th = nullptr; return false; }
            owner = getpid();
        }
        q.insert(q.end(), v.begin(), v.end());
        cv->notify_one();
        return true;
    }
    ~Releaser() {
        { std::lock_guard<std::mutex> g(m); stop = true; }
        cv->notify_one();
        if (th && owner == getpid()) { th->join(); delete th; }
        delete cv;
    }
};
}  // namespace

static Releaser &releaser();
// fork(): the child has the queue and the lock as they were but not the thread.  The lock is taken around the fork so that the
// child does not inherit it held by a thread it has not got; the child forgets the thread (and what was queued: those buffers
// are the parent's to release) and starts one of its own at its first hand-over.
static void releaser_prepare() { releaser().m.lock(); }
static void releaser_parent() { releaser().m.unlock(); }
static void releaser_child() {
    Releaser &r = releaser();
    if (r.th) r.cv = new std::condition_variable();            // (the old one, and the thread object, are left where they are)
    r.th = nullptr; r.q.clear(); r.held = 0;
    r.m.unlock();
}
static Releaser &releaser() {
    static Releaser r;
    static const int once = pthread_atfork(releaser_prepare, releaser_parent, releaser_child);
    (void)once;
    return r;
}

void release_hold(int on) {
    Releaser &r = releaser();
    { std::lock_guard<std::mutex> g(r.m); r.held += on ? 1 : -1; }
    r.cv->notify_one();
}

void release_later(void *const *ptrs, size_t n) {
    Releaser &r = releaser();
    std::vector<void *> v;
    for (size_t i = 0; i < n; i++) if (ptrs[i]) v.push_back(ptrs[i]);
    if (v.empty()) return;
    if (getenv("DEBWT_RELEASE_INLINE") || !r.give(v)) for (void *p : v) free(p);
}

#define GZ_TRACE(...) do { if (trace) fprintf(stderr, "gz_parallel: " __VA_ARGS__); } while (0)

// dst: where the text goes when the caller knows its length (dst_len; a member of a file of several) -- else a new buffer
static int gzip_parallel(const unsigned char *z, size_t zlen, int threads, char *dst, size_t dst_len, char **out_buf, size_t *out_len) {
    const bool trace = getenv("DEBWT_TRACE_GZ") != nullptr;
    if (threads < 2 || zlen < 18 + ((size_t)1 << 16)) return 1;
    // the member header (RFC 1952)
    if (z[0] != 0x1f || z[1] != 0x8b || z[2] != 8 || (z[3] & 0xE0)) return 1;
    const unsigned flg = z[3];
    size_t q = 10;
    if (flg & 4) { if (q + 2 > zlen) return 1; q += 2 + (z[q] | ((size_t)z[q + 1] << 8)); }
    if (flg & 8) { while (q < zlen && z[q]) q++; q++; }
    if (flg & 16) { while (q < zlen && z[q]) q++; q++; }
    if (flg & 2) q += 2;
    if (q + 8 >= zlen) return 1;
    const size_t nbits = 8 * (zlen - 8);                      // the trailer is not deflate data
    // pieces: the compressed bytes cut evenly (DEBWT_GZ_PIECE_BYTES for tests), at least 1 MB each, two per thread (every piece
    // but the first pays for a head of 0.5 - 3 MB at our own decoder's pace, whatever its size)
    // ... and at most 16 MB: more pieces than threads are dealt out as the threads get free, a piece whose markers never clear
    // (gzip -1: copies keep reaching back) stays below the 2^26 symbols a head may grow to, and no thread holds more than one
    // piece's head at our decoder's pace however large the file is
    size_t piece_bytes = (zlen / ((size_t)threads * 2)) + 1;
    if (piece_bytes < ((size_t)1 << 20)) piece_bytes = (size_t)1 << 20;
    if (piece_bytes > ((size_t)16 << 20)) piece_bytes = (size_t)16 << 20;
    if (const char *e = getenv("DEBWT_GZ_PIECE_BYTES")) { const long long v = atoll(e); if (v >= 4096) piece_bytes = (size_t)v; }
    const size_t npieces_max = (zlen - q + piece_bytes - 1) / piece_bytes;
    if (npieces_max < 2) return 1;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
    // 1. block starts, in parallel
    std::vector<size_t> start(npieces_max, nbits);
    start[0] = 8 * q;
    {
        std::atomic<size_t> next{1};
        auto work = [&] {
            for (size_t i; (i = next.fetch_add(1)) < npieces_max;) {
                const size_t from = 8 * (q + i * piece_bytes), to = std::min(nbits, 8 * (q + (i + 1) * piece_bytes));
                start[i] = from < nbits ? find_block(z, nbits, from, to) : nbits;
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < threads; t++) th.emplace_back(work);
        work();
        for (auto &x : th) x.join();
    }
    std::vector<Piece> pc;
    for (size_t i = 0; i < npieces_max; i++)
        if (start[i] < nbits) { pc.emplace_back(); pc.back().start_bit = start[i]; }    // (a piece without a start joins the one before)
    GZ_TRACE("%zu bytes, pieces of %zu bytes: %zu of %zu block starts found after %.3f s\n", zlen, piece_bytes, pc.size(), npieces_max, since());
    if (pc.size() < 2) return 1;
    // 2. the pieces, in parallel
    {
        std::atomic<size_t> next{0};
        auto work = [&] {
            for (size_t i; (i = next.fetch_add(1)) < pc.size();)
                decode_piece(z, zlen - 8, i + 1 < pc.size() ? pc[i + 1].start_bit : ~(size_t)0, i == 0, pc[i]);
        };
        std::vector<std::thread> th;
        for (int t = 1; t < threads; t++) th.emplace_back(work);
        work();
        for (auto &x : th) x.join();
    }
    GZ_TRACE("pieces decoded after %.3f s\n", since());
    auto cleanup = [&] { for (Piece &p : pc) { free(p.alloc); p.alloc = p.body = nullptr; } };
    int st = 0;
    for (size_t i = 0; i < pc.size(); i++) {
        const Piece &p = pc[i];
        GZ_TRACE("piece %zu: bits %zu..%zu, head %zu symbols, body %zu bytes, final %d, status %d\n", i, p.start_bit, p.end_bit, p.head.size(),
                 p.body_len, (int)p.final_block, p.status);
        if (p.status) { st = p.status; break; }
        const bool lastp = i + 1 == pc.size();
        if (p.final_block != lastp) { st = 1; break; }        // the final block in the middle: more than one member, or a wrong start
        if (!lastp && p.end_bit != pc[i + 1].start_bit) { st = 1; break; }
    }
    if (!st && ((pc.back().end_bit + 7) >> 3) != zlen - 8) { GZ_TRACE("bytes behind the final block\n"); st = 1; }   // another member
    if (st) { GZ_TRACE("declined after %.3f s of decoding (status %d)\n", since(), st); cleanup(); return st; }
    // 3. markers resolved piece by piece with the 32 KB before them; total length
    std::vector<size_t> off(pc.size() + 1, 0);
    for (size_t i = 0; i < pc.size(); i++) off[i + 1] = off[i] + pc[i].head.size() + pc[i].body_len;
    const size_t total = off.back();
    const size_t want_len = z[zlen - 4] | ((size_t)z[zlen - 3] << 8) | ((size_t)z[zlen - 2] << 16) | ((size_t)z[zlen - 1] << 24);
    if ((total & 0xFFFFFFFFull) != want_len) { GZ_TRACE("length %zu differs from the trailer's %zu\n", total, want_len); cleanup(); return 1; }
    if (dst && total != dst_len) { cleanup(); return 1; }
    char *buf = dst ? dst : (char *)big_malloc(total + 1);
    if (!buf) { cleanup(); return -1; }
    auto drop_buf = [&] { if (!dst) free(buf); };
    // the window of every piece, in order: only the last WIN bytes of a piece are needed for the next one's
    std::vector<std::vector<unsigned char>> window(pc.size() + 1);   // window[i]: the (up to) WIN bytes before piece i
    auto resolve = [&](const std::vector<unsigned char> &win, sym_t s, unsigned char *byte) {
        if (s < 256) { *byte = (unsigned char)s; return true; }
        const size_t w = (size_t)(s - MARK);                  // byte w of the window of WIN bytes that ends where the piece starts
        if (w + win.size() < WIN) return false;               // reaches before the start of the data
        *byte = win[w - (WIN - win.size())];
        return true;
    };
    for (size_t i = 0; i < pc.size() && !st; i++) {
        const Piece &p = pc[i];
        const size_t hl = p.head.size(), bl = p.body_len;
        std::vector<unsigned char> &nw = window[i + 1];
        if (bl >= WIN) nw.assign((unsigned char *)p.body + bl - WIN, (unsigned char *)p.body + bl);
        else {
            const size_t from_head = std::min(hl, WIN - bl), from_win = std::min(window[i].size(), WIN - bl - from_head);
            nw.insert(nw.end(), window[i].end() - (long)from_win, window[i].end());
            for (size_t j = hl - from_head; j < hl; j++) {
                unsigned char c;
                if (!resolve(window[i], p.head[j], &c)) { st = 1; break; }
                nw.push_back(c);
            }
            nw.insert(nw.end(), (unsigned char *)p.body, (unsigned char *)p.body + bl);
        }
    }
    if (st) { drop_buf(); cleanup(); return st; }
    GZ_TRACE("windows known after %.3f s\n", since());
    // heads resolved, bodies into place and the CRC of every piece, in parallel
    std::vector<uLong> crc(pc.size(), 0);
    std::atomic<int> bad{0};
    {
        std::atomic<size_t> next{0};
        auto work = [&] {
            for (size_t i; (i = next.fetch_add(1)) < pc.size();) {
                Piece &p = pc[i];
                unsigned char *dst = (unsigned char *)buf + off[i];
                const std::vector<unsigned char> &win = window[i];
                const size_t wbase = WIN - win.size();
                const size_t hl = p.head.size();
                if (wbase == 0 && hl >= 4 * WIN) {
                    // a whole window in front (every piece but those at the very start of the text): every marker is good, and a
                    // table by symbol value -- 64 KB: bytes as they are, markers by the window -- turns the head into bytes
                    // without a branch (the head of gzip -1 text is the whole piece)
                    std::vector<unsigned char> lut((size_t)1 << 16, 0);
                    for (size_t v = 0; v < 256; v++) lut[v] = (unsigned char)v;
                    memcpy(lut.data() + MARK, win.data(), WIN);
                    const sym_t *h = p.head.p;
                    for (size_t j = 0; j < hl; j++) dst[j] = lut[h[j]];
                } else {
                    for (size_t j = 0; j < hl; j++) {
                        const sym_t s = p.head[j];
                        if (s < 256) dst[j] = (unsigned char)s;
                        else if ((size_t)(s - MARK) >= wbase) dst[j] = win[(size_t)(s - MARK) - wbase];
                        else { bad = 1; break; }
                    }
                }
                if (p.body_len) memcpy(dst + (off[i + 1] - off[i] - p.body_len), p.body, p.body_len);
                // (the body is released behind the threads' join: an munmap takes the address space's lock for writing and waits
                // for -- and holds up -- every page fault of the threads that are filling `buf`)
                crc[i] = fastinflate::crc32_fast(0, dst, off[i + 1] - off[i]);
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < threads; t++) th.emplace_back(work);
        work();
        for (auto &x : th) x.join();
    }
    GZ_TRACE("resolved, copied and summed after %.3f s\n", since());
    {   // heads and bodies: released behind the caller's back (release_later; by all threads at once it took as long as one after
        // the other, and as long as inflating the text)
        std::vector<void *> gone;
        for (Piece &p : pc) {
            if (p.head.p) gone.push_back(p.head.p - p.head.lead);
            p.head.p = nullptr; p.head.n = p.head.cap = 0;
            gone.push_back(p.alloc);
            p.alloc = p.body = nullptr;
        }
        release_later(gone.data(), gone.size());
    }
    if (bad) { drop_buf(); return 1; }
    uLong all = crc32(0L, Z_NULL, 0);
    for (size_t i = 0; i < pc.size(); i++) all = crc32_combine(all, crc[i], (z_off_t)(off[i + 1] - off[i]));
    const uLong want_crc = z[zlen - 8] | ((uLong)z[zlen - 7] << 8) | ((uLong)z[zlen - 6] << 16) | ((uLong)z[zlen - 5] << 24);
    if (all != want_crc) { GZ_TRACE("CRC differs from the trailer's\n"); drop_buf(); return 1; }             // (the serial path will say whether the file is damaged)
    if (out_buf) *out_buf = buf;
    if (out_len) *out_len = total;
    return 0;
}

int inflate_gzip_parallel(const unsigned char *z, size_t zlen, int threads, char **out_buf, size_t *out_len) {
    return gzip_parallel(z, zlen, threads, nullptr, 0, out_buf, out_len);
}

// ---- several plain members ---------------------------------------------------------------------------------------------
// `cat a.fa.gz b.fa.gz`, one member per chromosome or per genome: members are independent deflate streams, but nothing says
// where the next one starts except the end of the one before.  Member HEADERS are found by their fixed bytes (ID1 ID2 CM = 1f
// 8b 08, reserved flag bits zero, XFL 0 / 2 / 4, a known OS byte -- ~10^-11 false positives per byte), every candidate is
// inflated on its own -- many members: by as many threads; few and large ones: one after the other, each cut
// into pieces by inflate_gzip_parallel -- and checked against the CRC32 and ISIZE that follow its final block; then the
// chain is walked from byte 0: every member must start exactly where the one before ended and the last must end the file.
// A candidate that does not inflate to a checked member (a false positive) is on no chain.  Returns 1 (inflate serially) for
// one member, a broken chain or bytes behind the last member; -1 out of memory.
namespace {

// data offset of the member whose header starts at p, or 0
size_t member_data(const unsigned char *z, size_t zlen, size_t p) {
    if (p + 18 > zlen || z[p] != 0x1f || z[p + 1] != 0x8b || z[p + 2] != 8 || (z[p + 3] & 0xE0)) return 0;
    const unsigned flg = z[p + 3], xfl = z[p + 8], os = z[p + 9];
    if ((xfl != 0 && xfl != 2 && xfl != 4) || (os > 13 && os != 255)) return 0;
    size_t q = p + 10;
    if (flg & 4) { if (q + 2 > zlen) return 0; q += 2 + (z[q] | ((size_t)z[q + 1] << 8)); }
    if (flg & 8) { while (q < zlen && z[q]) q++; q++; }
    if (flg & 16) { while (q < zlen && z[q]) q++; q++; }
    if (flg & 2) q += 2;
    return q + 8 <= zlen ? q : 0;
}

struct Member { size_t start = 0, end = 0; char *out = nullptr; size_t len = 0; int status = 1; };   // status 0: inflated and checked

// one member from its header at m.start; `hint`: compressed bytes up to the next candidate (sizes the first buffer)
void inflate_member(const unsigned char *z, size_t zlen, size_t hint, Member &m) {
    const size_t q = member_data(z, zlen, m.start);
    if (!q) return;
    std::unique_ptr<fastinflate::Decoder> d(new (std::nothrow) fastinflate::Decoder());
    if (!d) { m.status = -1; return; }
    size_t cap = std::max<size_t>(hint * 5, (size_t)1 << 20), len = 0;      // (reserved, touched only as far as the text goes)
    char *out = (char *)big_malloc(cap);
    int st = out ? 1 : -1;
    d->start(z + q, zlen - q, 0);
    while (out) {
        const int r = d->run((uint8_t *)out, 0, &len, cap, ~(size_t)0);
        if (r == fastinflate::FI_NEED_OUTPUT) {
            char *nb = (char *)big_malloc(cap * 2);
            if (!nb) { st = -1; break; }
            memcpy(nb, out, len);
            free(out);
            out = nb; cap *= 2;
            continue;
        }
        if (r == fastinflate::FI_DONE) {
            const size_t in_done = q + ((d->bitpos + 7) >> 3);
            if (in_done + 8 > zlen) break;
            const unsigned char *t = z + in_done;
            const uint32_t want_crc = t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
            const size_t want_len = t[4] | ((size_t)t[5] << 8) | ((size_t)t[6] << 16) | ((size_t)t[7] << 24);
            if (want_len == (len & 0xFFFFFFFFull) && want_crc == fastinflate::crc32_fast(0, (const uint8_t *)out, len)) { st = 0; m.end = in_done + 8; }
        }
        break;                                                  // done, or not a deflate stream: a false candidate (or a damaged file)
    }
    if (st == 0) { m.out = out; m.len = len; } else free(out);
    m.status = st;
}

// the member whose header starts at `start` and whose trailer ends at `end`, straight into dst[0 .. size): 0 when all of it holds
int member_into(const unsigned char *z, size_t zlen, size_t start, size_t end, char *dst, size_t size, fastinflate::Decoder &d) {
    const size_t q = member_data(z, zlen, start);
    if (!q || q + 8 > end) return 1;
    size_t len = 0;
    d.start(z + q, end - 8 - q, 0);
    if (d.run((uint8_t *)dst, 0, &len, size, ~(size_t)0) != fastinflate::FI_DONE || len != size || q + ((d.bitpos + 7) >> 3) != end - 8) return 1;
    const unsigned char *t = z + end - 8;
    const uint32_t want_crc = t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    return want_crc == fastinflate::crc32_fast(0, (const uint8_t *)dst, size) ? 0 : 1;
}

}  // namespace

int inflate_gzip_members(const unsigned char *z, size_t zlen, int threads, char **out_buf, size_t *out_len) {
    const bool trace = getenv("DEBWT_TRACE_GZ") != nullptr;
    if (threads < 1) threads = 1;
    if (!member_data(z, zlen, 0)) return 1;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
    // 1. candidate headers, by slices of the file
    std::vector<std::vector<size_t>> found((size_t)threads);
    {
        auto work = [&](int t) {
            const size_t a = zlen / (size_t)threads * (size_t)t, b = t + 1 == threads ? zlen : zlen / (size_t)threads * (size_t)(t + 1);
            for (size_t p = a; p < b;) {
                const void *hit = memchr(z + p, 0x1f, b - p);
                if (!hit) break;
                p = (size_t)((const unsigned char *)hit - z);
                if (member_data(z, zlen, p)) found[(size_t)t].push_back(p);
                p++;
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < threads; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    }
    std::vector<size_t> cand;
    for (auto &f : found) cand.insert(cand.end(), f.begin(), f.end());
    GZ_TRACE("%zu candidate member headers after %.3f s\n", cand.size(), since());
    if (cand.size() < 2) return 1;                            // one member: inflate_gzip_parallel's case
    // 1b. The straight case first: EVERY candidate is a member.  Then each member ends where the next candidate begins, its
    // ISIZE are the four bytes in front of that, and the place of every member's text in the one buffer is known before anything
    // is inflated: no buffer per member, no copy.  Any member that does not come out at exactly its length, end and CRC-32
    // (a false candidate, a member of 4 GB or more, damage) sends the file to the general way below.
    if (cand[0] == 0 && !getenv("DEBWT_GZ_MEMBERS_GENERAL")) {
        const size_t nm = cand.size();
        std::vector<size_t> off(nm + 1, 0);
        for (size_t i = 0; i < nm; i++) {
            const size_t end = i + 1 < nm ? cand[i + 1] : zlen;
            off[i + 1] = off[i] + (z[end - 4] | ((size_t)z[end - 3] << 8) | ((size_t)z[end - 2] << 16) | ((size_t)z[end - 1] << 24));
        }
        char *buf = off[nm] / 1100 <= zlen ? (char *)big_malloc(off[nm] + 1) : nullptr;      // (deflate cannot expand more than 1032 times)
        std::atomic<int> bad{buf ? 0 : 1};
        if (buf && nm * 2 > (size_t)threads) {
            std::vector<size_t> order(nm);                        // the largest first: the last member to finish is a small one
            for (size_t i = 0; i < nm; i++) order[i] = i;
            std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return off[a + 1] - off[a] > off[b + 1] - off[b]; });
            std::atomic<size_t> next{0};
            auto work = [&] {
                std::unique_ptr<fastinflate::Decoder> d(new (std::nothrow) fastinflate::Decoder());
                if (!d) { bad = 1; return; }
                for (size_t k; !bad && (k = next.fetch_add(1)) < nm;) {
                    const size_t i = order[k];
                    if (member_into(z, zlen, cand[i], i + 1 < nm ? cand[i + 1] : zlen, buf + off[i], off[i + 1] - off[i], *d)) bad = 1;
                }
            };
            std::vector<std::thread> th;
            for (int t = 1; t < threads; t++) th.emplace_back(work);
            work();
            for (auto &x : th) x.join();
        } else if (buf) {
            std::unique_ptr<fastinflate::Decoder> d(new (std::nothrow) fastinflate::Decoder());
            for (size_t i = 0; d && i < nm && !bad; i++) {
                const size_t end = i + 1 < nm ? cand[i + 1] : zlen;
                if (gzip_parallel(z + cand[i], end - cand[i], threads, buf + off[i], off[i + 1] - off[i], nullptr, nullptr) != 0 &&
                    member_into(z, zlen, cand[i], end, buf + off[i], off[i + 1] - off[i], *d)) bad = 1;
            }
            if (!d) bad = 1;
        }
        if (!bad) {
            GZ_TRACE("%zu members, %zu bytes, each inflated into its place, after %.3f s\n", nm, off[nm], since());
            *out_buf = buf; *out_len = off[nm];
            return 0;
        }
        free(buf);
        GZ_TRACE("not every candidate is a member that ends at the next: the general way (after %.3f s)\n", since());
    }
    std::vector<Member> chain;
    auto cleanup = [&](std::vector<Member> &v) { for (Member &m : v) { free(m.out); m.out = nullptr; } };
    if (cand.size() * 2 > (size_t)threads) {
        // many members: every candidate on its own, by all threads (a false candidate fails within its first block)
        std::vector<Member> mem(cand.size());
        for (size_t i = 0; i < cand.size(); i++) mem[i].start = cand[i];
        std::atomic<size_t> next{0};
        auto work = [&] {
            for (size_t i; (i = next.fetch_add(1)) < mem.size();)
                inflate_member(z, zlen, (i + 1 < cand.size() ? cand[i + 1] : zlen) - cand[i], mem[i]);
        };
        std::vector<std::thread> th;
        for (int t = 1; t < threads; t++) th.emplace_back(work);
        work();
        for (auto &x : th) x.join();
        GZ_TRACE("candidates inflated after %.3f s\n", since());
        size_t cur = 0, i = 0;
        bool ok = true, oom = false;
        for (const Member &m : mem) oom = oom || m.status < 0;
        while (ok && !oom && cur < zlen) {
            while (i < mem.size() && mem[i].start < cur) i++;
            if (i == mem.size() || mem[i].start != cur || mem[i].status != 0) { ok = false; break; }
            chain.push_back(mem[i]); mem[i].out = nullptr;     // (the buffer moves to the chain)
            cur = chain.back().end;
        }
        cleanup(mem);
        if (oom) { cleanup(chain); return -1; }
        if (!ok) { GZ_TRACE("the members do not chain from byte %zu on: left to the serial path\n", cur); cleanup(chain); return 1; }
    } else {
        // few members, so large ones: one after the other, each by all threads
        size_t cur = 0, i = 0;
        while (cur < zlen) {
            while (i < cand.size() && cand[i] <= cur) i++;
            const size_t nxt = i < cand.size() ? cand[i] : zlen;
            Member m;
            m.start = cur;
            if (inflate_gzip_parallel(z + cur, nxt - cur, threads, &m.out, &m.len) == 0) { m.end = nxt; m.status = 0; }
            else inflate_member(z, zlen, nxt - cur, m);         // small, not text, or a false candidate cut it short: decoding finds its end
            if (m.status < 0) { cleanup(chain); return -1; }
            if (m.status) { GZ_TRACE("no member at byte %zu: left to the serial path\n", cur); cleanup(chain); return 1; }
            chain.push_back(m);
            cur = m.end;
        }
        GZ_TRACE("%zu members inflated one after the other after %.3f s\n", chain.size(), since());
    }
    // 2. one buffer
    std::vector<size_t> off(chain.size() + 1, 0);
    for (size_t j = 0; j < chain.size(); j++) off[j + 1] = off[j] + chain[j].len;
    char *buf = (char *)big_malloc(off.back() + 1);
    if (!buf) { cleanup(chain); return -1; }
    {
        std::atomic<size_t> next{0};
        auto work = [&] {
            for (size_t j; (j = next.fetch_add(1)) < chain.size();) {
                if (chain[j].len) memcpy(buf + off[j], chain[j].out, chain[j].len);
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < threads && (size_t)t < chain.size(); t++) th.emplace_back(work);
        work();
        for (auto &x : th) x.join();
    }
    {
        std::vector<void *> gone;
        for (Member &m : chain) { gone.push_back(m.out); m.out = nullptr; }
        release_later(gone.data(), gone.size());
    }
    GZ_TRACE("%zu members, %zu bytes after %.3f s\n", chain.size(), off.back(), since());
    *out_buf = buf; *out_len = off.back();
    return 0;
}

