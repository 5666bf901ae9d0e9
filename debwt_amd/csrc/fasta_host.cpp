// fasta_host.cpp -- multi-threaded FASTA (and FASTQ) -> 2-bit packed text.
//
// Replaces the reference's single-threaded ingest (kseq.h + zlib, one base at a time into `reference`,
// src/collect#$.c:34-90): the file is mapped (or inflated, when gzip), cut into one chunk per thread at arbitrary
// byte offsets, and parsed in two parallel passes -- (1) census: bases and header starts per chunk, (2) pack: every
// thread writes its bases, and the separators of the records that end in its chunk, at their final 2-bit positions.
// A chunk start that falls inside a header line is recognised by looking back to the start of its line.
// Layout, alphabet and checks are the reference's: A0 C1 G2 T3 either case (src/main.c:18-23), 'T' at every
// separator and 32 'T' behind the end (src/collect#$.c:78-90), every record longer than 32 bases (:41-45).
#include "fasta_host.h"
#include "fast_inflate.h"
#include "gz_parallel.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <memory>
#include <new>
#include <thread>
#include <vector>

namespace {

enum : uint8_t { C_SKIP = 0x80, C_BAD = 0xC0 };      // bit 7: not a base

struct Lut {
    uint8_t t[256];
    Lut() {
        memset(t, C_BAD, sizeof t);
        t[(int)'A'] = t[(int)'a'] = 0; t[(int)'C'] = t[(int)'c'] = 1;
        t[(int)'G'] = t[(int)'g'] = 2; t[(int)'T'] = t[(int)'t'] = 3;
        t[(int)'\n'] = t[(int)'\r'] = t[(int)' '] = t[(int)'\t'] = C_SKIP;
    }
};
const Lut LUT;

// IUPAC ambiguity letters and the bases they stand for, in the order of the reference's randTable
// (otherTool/transferN.c:8-9,17-27)
struct Iupac {
    uint8_t len[256];
    uint8_t set[256][4];
    Iupac() {
        memset(len, 0, sizeof len); memset(set, 0, sizeof set);
        auto def = [&](char c, const char *bases) {
            for (int u = 0; u < 2; u++) {
                const int ch = u ? c | 0x20 : c;
                len[ch] = (uint8_t)strlen(bases);
                for (int i = 0; bases[i]; i++) set[ch][i] = LUT.t[(unsigned char)bases[i]];
            }
        };
        def('N', "ACGT"); def('V', "ACG"); def('D', "ATG"); def('B', "TCG"); def('H', "ATC"); def('W', "AT");
        def('S', "CG"); def('K', "TG"); def('M', "AC"); def('Y', "CT"); def('R', "AG");
    }
};
const Iupac IUPAC;
inline uint64_t mix64(uint64_t x) {                      // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
inline uint8_t iupac_pick(unsigned char ch, uint64_t pos, uint64_t seed) {
    return IUPAC.set[ch][mix64(seed ^ (pos * 0xD6E8FEB86659FD93ull)) % IUPAC.len[ch]];
}

struct Chunk {
    size_t beg = 0, end = 0;
    bool starts_in_header = false;
    bool starts_a_line = false;           // the chunk is a piece of its own (a FASTQ rewrite's): what lies before beg is not text
    uint64_t head_bases = 0;              // bases before the chunk's first header start (they belong to an earlier record)
    std::vector<uint64_t> rec_bases;      // per header start in the chunk: bases from there to the next start / chunk end
    uint64_t sum_bases = 0;               // of rec_bases
    uint64_t min_complete = ~0ull;        // the shortest of rec_bases but the last (which later chunks may continue)
    // filled by the serial combine
    uint64_t sym0 = 0;                    // text position of the chunk's first symbol
    uint64_t rec0 = 0;                    // records started before the chunk
    long bad_at = -1;                     // offset of the first invalid character
};

// Eight characters at once: their 2-bit codes as 16 bits, first character in the top two bits; false when one of the
// eight is not ACGTacgt.  code = bits (1 ^ 2, 2 ^ 3) of the upper-cased character (A 0x41, C 0x43, G 0x47, T 0x54 ->
// 0, 1, 2, 3); the check rebuilds the character from its code and compares.
inline bool swar8(const char *p, uint64_t *codes) {
    uint64_t x;
    memcpy(&x, p, 8);
    x = __builtin_bswap64(x);                                        // first character in the most significant byte
    const uint64_t y = x & 0xDFDFDFDFDFDFDFDFull;                    // upper case
    const uint64_t c = ((y >> 1) ^ (y >> 2)) & 0x0303030303030303ull;
    const uint64_t lo = c & 0x0101010101010101ull, hi = (c >> 1) & 0x0101010101010101ull;
    const uint64_t back = 0x4141414141414141ull + 2 * lo + 6 * hi + 0x0B * (lo & hi);
    uint64_t v = c;
    v = (v | (v >> 6)) & 0x000F000F000F000Full;
    v = (v | (v >> 12)) & 0x000000FF000000FFull;
    v = (v | (v >> 24)) & 0xFFFFull;
    *codes = v;
    return back == y;
}

// pass 2 state: the word under construction is shifted left two bits per symbol; the first and the last word of a
// chunk may be shared with the neighbouring chunks and are OR-ed in atomically, the words between belong to it alone
struct Packer {
    uint64_t *words;
    uint64_t w_first, pos, acc;
    inline void flush_full() {                // pos is a multiple of 32 here: word pos/32 - 1 is complete
        const uint64_t w = (pos >> 5) - 1;
        if (w == w_first) __atomic_fetch_or(&words[w], acc, __ATOMIC_RELAXED); else words[w] = acc;
        acc = 0;
    }
    inline void put(uint64_t v, unsigned g) { // g <= 32 symbols, first symbol in the top bits of the 2g-bit value
        const unsigned fill = (unsigned)(pos & 31);                  // symbols in the word under construction
        if (fill + g < 32) { acc = (acc << (2 * g)) | v; pos += g; return; }
        const unsigned head = 32 - fill, tail = g - head;            // head symbols complete the word
        acc = (head == 32 ? 0ull : acc << (2 * head)) | (v >> (2 * tail));
        pos += head; flush_full();
        acc = tail ? v & ((1ull << (2 * tail)) - 1ull) : 0ull;
        pos += tail;
    }
    inline void finish() {                    // the last, partial word of the chunk is shared with the next chunk
        if (pos & 31) __atomic_fetch_or(&words[pos >> 5], acc << ((32 - (pos & 31)) << 1), __ATOMIC_RELAXED);
    }
};

// Consumes sequence text from p[0, len): bases (counted in *bases; with PACK appended to pk) and the newlines between
// sequence lines.  Returns the offset of the first character that needs the caller: white space other than such a
// newline, an invalid character, a newline before a header or at the very end -- or len.
template <bool PACK>
size_t run_swar(const char *p, size_t len, Packer *pk, uint64_t *bases) {
    size_t j = 0;
    uint64_t nb = 0;
    for (;;) {
        uint64_t v;
        for (; j + 8 <= len && swar8(p + j, &v); j += 8, nb += 8)
            if (PACK) pk->put(v, 8);
        for (; j < len; j++, nb++) {
            const uint8_t c = LUT.t[(unsigned char)p[j]];
            if (c > 3) break;
            if (PACK) pk->put(c, 1);
        }
        if (j + 1 < len && p[j] == '\n' && p[j + 1] != '>') { j++; continue; }
        break;
    }
    *bases += nb;
    return j;
}

#if defined(__x86_64__)
// 32 characters per step (AVX2, chosen at run time): same code arithmetic per byte, the check by a byte shuffle of
// "ACGT", the 64-bit word by two multiply-adds (4 codes -> a byte) and a byte reversal.
template <bool PACK>
__attribute__((target("avx2"))) size_t run_avx2(const char *p, size_t len, Packer *pk, uint64_t *bases) {
    const __m256i up_mask = _mm256_set1_epi8((char)0xDF), three = _mm256_set1_epi8(3);
    const __m256i acgt = _mm256_setr_epi8('A', 'C', 'G', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                          'A', 'C', 'G', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i newline = _mm256_set1_epi8('\n');
    size_t j = 0;
    uint64_t nb = 0;
    // Blocks sit at fixed offsets while the text is bases with at most one line break per block (any line of 32
    // characters or more), so that the next block's address does not wait for this block's classification.
    while (j + 32 <= len) {
        const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + j));
        const __m256i up = _mm256_and_si256(x, up_mask);
        const __m256i c = _mm256_and_si256(_mm256_xor_si256(_mm256_srli_epi16(up, 1), _mm256_srli_epi16(up, 2)), three);
        const uint32_t ok = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_shuffle_epi8(acgt, c), up));
        uint64_t v = 0;
        if (PACK) {
            const __m256i b4 = _mm256_maddubs_epi16(c, _mm256_set1_epi16(0x0104));        // 2 codes -> 4 bits
            const __m256i b8 = _mm256_madd_epi16(b4, _mm256_set1_epi32(0x00010010));      // 4 codes -> 8 bits per dword
            const __m256i w16 = _mm256_packs_epi32(b8, b8);
            const __m256i w8 = _mm256_packus_epi16(w16, w16);
            const uint64_t le = (uint32_t)_mm256_extract_epi32(w8, 0) | ((uint64_t)(uint32_t)_mm256_extract_epi32(w8, 4) << 32);
            v = __builtin_bswap64(le);                                                    // first character on top
        }
        if (ok == 0xFFFFFFFFu) {
            if (PACK) pk->put(v, 32);
            j += 32; nb += 32;
            continue;
        }
        const uint32_t nl = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(x, newline));
        if ((ok | nl) == 0xFFFFFFFFu && (nl & (nl - 1)) == 0) {                            // one line break, bases around it
            const unsigned f = (unsigned)__builtin_ctz(nl);
            const bool header_next = f == 31 && (j + 32 >= len || p[j + 32] == '>');
            if (!header_next) {
                if (PACK) {
                    if (f) pk->put(v >> (2 * (32 - f)), f);
                    if (f < 31) pk->put(v & ((1ull << (2 * (31 - f))) - 1ull), 31 - f);
                }
                j += 32; nb += 31;
                continue;
            }
        }
        const unsigned g = (unsigned)__builtin_ctz(~ok);                                   // bases before the first other character
        if (PACK && g) pk->put(v >> (2 * (32 - g)), g);
        j += g; nb += g;
        if (j + 1 < len && p[j] == '\n' && p[j + 1] != '>') { j++; continue; }            // next sequence line
        *bases += nb;
        return j;
    }
    *bases += nb;
    return j + run_swar<PACK>(p + j, len - j, pk, bases);
}
#endif

typedef size_t (*RunFn)(const char *, size_t, Packer *, uint64_t *);
template <bool PACK> RunFn pick_run() {
#if defined(__x86_64__)
    if (__builtin_cpu_supports("avx2") && !getenv("DEBWT_INGEST_NO_AVX2")) return run_avx2<PACK>;
#endif
    return run_swar<PACK>;
}

// is the byte at `pos` inside a header line?  (the line's first character is '>')
bool in_header_at(const char *buf, size_t pos) {
    const void *nl = pos ? memrchr(buf, '\n', pos) : nullptr;          // last newline before pos
    const size_t ls = nl ? (size_t)((const char *)nl - buf) + 1 : 0;
    return buf[ls] == '>' && pos > ls;      // pos == ls: the '>' itself is seen by the chunk's own scan
}

// Walks a chunk: on_header() at every header start (bases seen since the last call are in *bases), on_char(ptr) for
// every character of a sequence line that `run` does not take (see there).  Sequence lines are not searched for their
// end first: `run` classifies 32 (or 8) characters at a time.  A header that runs past the chunk end is finished by
// the next chunk, which knows that it starts inside one.
template <class H, class O>
void scan_chunk(const char *buf, const Chunk &c, RunFn run, Packer *pk, uint64_t *bases, H on_header, O on_char) {
    size_t i = c.beg;
    if (c.starts_in_header) {
        const void *nl = memchr(buf + i, '\n', c.end - i);
        if (!nl) return;
        i = (size_t)((const char *)nl - buf) + 1;
    }
    bool bol = c.starts_a_line || i == 0 || buf[i - 1] == '\n';
    while (i < c.end) {
        if (bol && buf[i] == '>') {
            on_header();
            const void *nl = memchr(buf + i, '\n', c.end - i);
            if (!nl) return;
            i = (size_t)((const char *)nl - buf) + 1;
            continue;
        }
        bol = false;
        i += run(buf + i, c.end - i, pk, bases);
        if (i >= c.end) break;
        if (buf[i] == '\n') bol = true; else on_char(buf + i);
        i++;
    }
}

// pass 1: bases per record piece
void census(const char *buf, Chunk &c, IngestOpts opts) {
    uint64_t cur = 0;
    bool any = false;
    const bool iupac = (opts.flags & INGEST_IUPAC_RANDOM) != 0;
    scan_chunk(buf, c, pick_run<false>(), nullptr, &cur,
        [&]() {
            if (any) c.rec_bases.push_back(cur); else c.head_bases = cur;
            any = true; cur = 0;
        },
        [&](const char *p) {
            if (LUT.t[(unsigned char)*p] != C_BAD) return;
            if (iupac && IUPAC.len[(unsigned char)*p]) cur++;
            else if (c.bad_at < 0) c.bad_at = (long)(p - buf);
        });
    if (any) c.rec_bases.push_back(cur); else c.head_bases = cur;
    for (size_t j = 0; j < c.rec_bases.size(); j++) {
        c.sum_bases += c.rec_bases[j];
        if (j + 1 < c.rec_bases.size() && c.rec_bases[j] < c.min_complete) c.min_complete = c.rec_bases[j];
    }
}

// pass 2: symbols of the chunk at their text positions; a header start closes the record before it with a 'T'
void pack(const char *buf, const Chunk &c, uint64_t *words, uint64_t *sep, IngestOpts opts) {
    Packer pk{words, c.sym0 >> 5, c.sym0, 0};
    uint64_t rec = c.rec0, bases = 0;
    scan_chunk(buf, c, pick_run<true>(), &pk, &bases,
        [&]() {
            if (rec > 0) { sep[rec - 1] = pk.pos; pk.put(3, 1); }     // separator of the record that ends here
            rec++;
        },
        [&](const char *p) {                                          // white space, or (pass 1 has rejected anything
            const unsigned char ch = (unsigned char)*p;               // else) an ambiguity letter to be replaced
            if (LUT.t[ch] == C_BAD && IUPAC.len[ch]) pk.put(iupac_pick(ch, pk.pos, opts.seed), 1);
        });
    pk.finish();
}

int fail(char *err, size_t errlen, const std::string &msg) {
    if (err && errlen) snprintf(err, errlen, "%s", msg.c_str());
    return -1;
}

// FASTQ, which the reference's reader accepts as well (klib kseq_read, src/kseq.h:177-201: '@' or '>' header, sequence
// lines up to a line that starts with '+', '@' or '>'; after '+' quality characters until there are as many as bases,
// quality of another length = error; a record that ends at the next header has no qualities): the records are rewritten as
// header-less FASTA (">\n" + the sequence lines) into `out`, which the chunked FASTA parser then takes.
//
// fastq_records walks the records that START in [*pi, stop) (the last one is finished wherever it ends) and leaves *pi behind
// them; out == nullptr: nothing is written, *po only counts.  0, or -1 with a message (err may be null).
inline size_t fq_line_end(const char *buf, size_t len, size_t from) {      // index of the newline that ends the line at `from`, or len
    const void *nl = memchr(buf + from, '\n', len - from);
    return nl ? (size_t)((const char *)nl - buf) : len;
}
inline size_t fq_payload(const char *buf, size_t from, size_t to) {        // characters of [from, to) that are not white space
    // eight at a time: a line of a read set holds no byte below 0x21 (blank, tab and CR all are) -- then all of it counts
    size_t j = from;
    uint64_t low = 0;
    for (; j + 8 <= to; j += 8) {
        uint64_t x;
        memcpy(&x, buf + j, 8);
        low |= (x - 0x2121212121212121ull) & ~x & 0x8080808080808080ull;     // a byte < 0x21 sets its top bit (a borrow from a lower byte only adds false alarms)
    }
    for (; j < to; j++) low |= (unsigned char)buf[j] < 0x21 ? 1u : 0u;
    if (!low) return to - from;
    size_t c = 0;
    for (j = from; j < to; j++) c += !(buf[j] == '\r' || buf[j] == ' ' || buf[j] == '\t');
    return c;
}
int fastq_records(const char *buf, size_t len, size_t *pi, size_t stop, char *out, size_t *po, uint64_t *prec, char *err, size_t errlen) {
    size_t i = *pi, o = *po;
    uint64_t rec = *prec;
    int rc = 0;
    while (i < len && i < stop) {
        if (buf[i] == '\n' || buf[i] == '\r') { i++; continue; }               // blank lines between records
        if (buf[i] != '@' && buf[i] != '>') {
            char m[120];
            snprintf(m, sizeof m, "FASTQ record %llu does not start with '@' or '>' (byte %zu)", (unsigned long long)rec + 1, i);
            fail(err, errlen, m);
            rc = -1;
            break;
        }
        i = std::min(len, fq_line_end(buf, len, i) + 1);                        // header line: the name is not used
        if (out) { out[o] = '>'; out[o + 1] = '\n'; }
        o += 2;
        size_t bases = 0;
        // sequence lines, up to a line that starts with '+' (qualities follow) or with '@' / '>' (the next record: this one
        // has no quality section -- kseq_read ends a sequence at any of the three, src/kseq.h:188) or the end of the input
        while (i < len && buf[i] != '+' && buf[i] != '@' && buf[i] != '>') {
            const size_t e = fq_line_end(buf, len, i);
            bases += fq_payload(buf, i, e);
            if (out) { memcpy(out + o, buf + i, e - i); out[o + (e - i)] = '\n'; }
            o += e - i + 1;
            i = std::min(len, e + 1);
        }
        rec++;
        if (i >= len || buf[i] != '+') continue;                                // a record without qualities
        rec--;
        i = std::min(len, fq_line_end(buf, len, i) + 1);                        // the '+' line
        size_t qual = 0;
        while (i < len && qual < bases) {                                       // quality lines
            const size_t e = fq_line_end(buf, len, i);
            qual += fq_payload(buf, i, e);
            i = std::min(len, e + 1);
        }
        if (qual != bases) {
            char m[160];
            snprintf(m, sizeof m, "FASTQ record %llu: %zu quality characters for %zu bases (truncated file?)",
                     (unsigned long long)rec + 1, qual, bases);
            fail(err, errlen, m);
            rc = -1;
            break;
        }
        rec++;
    }
    *pi = i; *po = o; *prec = rec;
    return rc;
}

// One serial walk over the lines (memchr speed).  Returns the FASTA length, or -1 with a message.
long fastq_to_fasta(const char *buf, size_t len, char *out, char *err, size_t errlen) {
    size_t i = 0, o = 0;
    uint64_t rec = 0;
    return fastq_records(buf, len, &i, len, out, &o, &rec, err, errlen) ? -1 : (long)o;
}

// The same rewrite by all threads.  A quality line may start with '@' or '>', so a record start cannot be RECOGNISED from the
// middle of the file -- but it can be guessed and the guess proved: every thread but the first looks, from its cut of the file
// on, for a line that starts with '@' and begins two records in a row of the full form (header, at least one sequence line of
// letters, a '+' line, as many quality characters as bases, then '@' or the end); thread 0 starts at byte 0, which is a record
// start; every thread walks (as the serial walk does) the records that start before the next thread's guess, and the walk
// must END exactly on that guess.  By induction from byte 0 every guess is then a record start of the one serial walk, and the
// pieces of FASTA, one behind the other, are what it would have written.
// Anything else (a walk that does not end on the next guess, any malformed record) returns -1 and the serial walk runs, with
// its messages.  Chunks without a guess (no full-form record starts there: records without qualities, FASTA records mixed
// in) are walked by the thread before them.
bool fq_probe(const char *buf, size_t len, size_t p, size_t *next) {         // a full-form record at p?  *next: where the one behind it starts
    if (p >= len || buf[p] != '@') return false;
    size_t i = std::min(len, fq_line_end(buf, len, p) + 1), bases = 0;
    while (i < len && buf[i] != '+' && buf[i] != '@' && buf[i] != '>') {
        const size_t e = fq_line_end(buf, len, i);
        for (size_t j = i; j < e; j++) {
            const unsigned char c = (unsigned char)buf[j];
            if (c == '\r' || c == ' ' || c == '\t') continue;
            if (!((c | 0x20) >= 'a' && (c | 0x20) <= 'z') && c != '*' && c != '-' && c != '.') return false;
            bases++;
        }
        i = std::min(len, e + 1);
    }
    if (!bases || i >= len || buf[i] != '+') return false;
    i = std::min(len, fq_line_end(buf, len, i) + 1);
    size_t qual = 0;
    while (i < len && qual < bases) {
        const size_t e = fq_line_end(buf, len, i);
        qual += fq_payload(buf, i, e);
        i = std::min(len, e + 1);
    }
    if (qual != bases) return false;
    while (i < len && (buf[i] == '\n' || buf[i] == '\r')) i++;
    if (i < len && buf[i] != '@') return false;
    *next = i;
    return true;
}
// pieces: (begin, end) in `out` of every thread's FASTA.  A thread's piece is written at the offset at which its records
// begin in the input -- the rewrite never grows (a header line of >= 2 bytes becomes 2, a sequence line stays as long; only a
// last line without its newline gains one: `out` has two bytes of slack), so the pieces cannot meet -- and the parser takes the
// pieces as its chunks: one walk, nothing is moved.
bool fastq_to_fasta_parallel(const char *buf, size_t len, char *out, int threads, std::vector<std::pair<size_t, size_t>> *pieces) {
    size_t per = (size_t)1 << 20;                                              // at least 1 MB per thread (tests: DEBWT_FASTQ_CHUNK_MIN)
    if (const char *e = getenv("DEBWT_FASTQ_CHUNK_MIN")) { const long long v = atoll(e); if (v >= 16) per = (size_t)v; }
    const size_t T = std::min<size_t>((size_t)threads, len / per);
    if (T < 2 || getenv("DEBWT_FASTQ_SERIAL")) return false;
    if (getenv("DEBWT_FASTQ_REQUIRE_PARALLEL")) fprintf(stderr, "fastq: %zu threads\n", T);
    std::vector<size_t> start(T + 1, len), stop(T, len), bytes(T, 0);
    std::vector<int> bad(T, 0);
    start[0] = 0;
    auto all = [&](auto work) {
        std::vector<std::thread> th;
        for (size_t t = 1; t < T; t++) th.emplace_back(work, t);
        work((size_t)0);
        for (auto &x : th) x.join();
    };
    all([&](size_t t) {                                                        // the guesses
        if (t == 0) return;
        const size_t from = len / T * t, to = t + 1 == T ? len : len / T * (t + 1);
        size_t p = from;
        if (p && buf[p - 1] != '\n') p = std::min(len, fq_line_end(buf, len, p) + 1);     // to the next line start
        for (int tries = 0; p < to && tries < 64; tries++) {                   // (a chunk of records of another form: leave it to the thread before)
            size_t n1, n2;
            if (buf[p] == '@' && fq_probe(buf, len, p, &n1) && (n1 >= len || fq_probe(buf, len, n1, &n2))) { start[t] = p; return; }
            p = std::min(len, fq_line_end(buf, len, p) + 1);
        }
    });
    for (size_t t = T; t-- > 0;) stop[t] = t + 1 < T ? (start[t + 1] < len ? start[t + 1] : stop[t + 1]) : len;
    all([&](size_t t) {                                                        // the walk, and the proof
        if (start[t] >= len && t) return;
        size_t i = start[t], o = start[t];
        uint64_t rec = 0;
        if (fastq_records(buf, len, &i, stop[t], out, &o, &rec, nullptr, 0) || i != stop[t]) bad[t] = 1;
        bytes[t] = o - start[t];
    });
    for (size_t t = 0; t < T; t++) if (bad[t]) return false;
    pieces->clear();
    for (size_t t = 0; t < T; t++)
        if (bytes[t]) pieces->emplace_back(start[t], start[t] + bytes[t]);
    return true;
}

}  // namespace

// census, combine and pack over the chunks given (equal cuts of a FASTA text, or the pieces of a FASTQ rewrite)
static int pack_chunks(const char *buf, size_t len, std::vector<Chunk> &ch, PackedText *out, char *err, size_t errlen, IngestOpts opts);

int pack_fasta_buffer(const char *buf, size_t len, int threads, PackedText *out, char *err, size_t errlen, IngestOpts opts) {
    if (threads < 1) threads = 1;
    if (len == 0) return fail(err, errlen, "empty input");
    if (buf[0] == '@') {
        const bool trace = getenv("DEBWT_TRACE_INGEST") != nullptr;
        const auto t_begin = std::chrono::steady_clock::now();
        // header, sequence and '+' lines never grow: 2 bytes for every header of >= 2 (">\n" for "@\n")
        char *fa = (char *)big_malloc(len + 2);
        if (!fa) return fail(err, errlen, "out of memory");
        std::vector<std::pair<size_t, size_t>> pieces;
        int rc;
        if (fastq_to_fasta_parallel(buf, len, fa, threads, &pieces)) {
            if (trace) fprintf(stderr, "ingest: FASTQ rewritten as %zu pieces of FASTA after %.3f s\n", pieces.size(),
                               std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
            std::vector<Chunk> ch(pieces.size());
            for (size_t t = 0; t < pieces.size(); t++) { ch[t].beg = pieces[t].first; ch[t].end = pieces[t].second; ch[t].starts_a_line = true; }
            rc = ch.empty() ? fail(err, errlen, "no FASTQ record") : pack_chunks(fa, len + 2, ch, out, err, errlen, opts);
        } else {
            const long fl = fastq_to_fasta(buf, len, fa, err, errlen);            // one thread, a small file, or the messages
            rc = fl < 0 ? -1 : (fl == 0 ? fail(err, errlen, "no FASTQ record") : pack_fasta_buffer(fa, (size_t)fl, threads, out, err, errlen, opts));
        }
        void *gone = fa;
        release_later(&gone, 1);
        return rc;
    }
    size_t nch = std::min<size_t>((size_t)threads, std::max<size_t>(1, len >> 16));
    std::vector<Chunk> ch(nch);
    for (size_t t = 0; t < nch; t++) {
        ch[t].beg = len / nch * t;
        ch[t].end = t + 1 == nch ? len : len / nch * (t + 1);
        ch[t].starts_in_header = ch[t].beg > 0 && in_header_at(buf, ch[t].beg);
    }
    return pack_chunks(buf, len, ch, out, err, errlen, opts);
}

static int pack_chunks(const char *buf, size_t len, std::vector<Chunk> &ch, PackedText *out, char *err, size_t errlen, IngestOpts opts) {
    const size_t nch = ch.size();
    const bool trace = getenv("DEBWT_TRACE_INGEST") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
    {
        std::vector<std::thread> th;
        for (size_t t = 1; t < nch; t++) th.emplace_back(census, buf, std::ref(ch[t]), opts);
        census(buf, ch[0], opts);
        for (auto &x : th) x.join();
    }
    if (trace) fprintf(stderr, "ingest: census of %zu bytes by %zu threads after %.3f s\n", len, nch, since());
    // serial combine: chunk offsets, and the record that is open across chunk ends (10^7 reads: no walk over the records here)
    uint64_t sym = 0, nrec_all = 0, open_len = 0;
    bool short_rec = false;
    for (size_t t = 0; t < nch; t++) {
        Chunk &c = ch[t];
        if (c.bad_at >= 0) {
            char m[200];
            snprintf(m, sizeof m, "character 0x%02x at byte %ld is not one of ACGTacgt (N and other ambiguity letters: the IUPAC option, or otherTool/transferN)",
                     (unsigned)(unsigned char)buf[c.bad_at], c.bad_at);
            return fail(err, errlen, m);
        }
        c.sym0 = sym; c.rec0 = nrec_all;
        if (c.head_bases) {
            if (!nrec_all) return fail(err, errlen, "sequence before the first header");
            open_len += c.head_bases;
        }
        sym += c.head_bases;
        if (!c.rec_bases.empty()) {
            if (nrec_all && open_len <= 32) short_rec = true;                        // the open record ends at this chunk's first header
            if (c.min_complete <= 32) short_rec = true;
            sym += c.sum_bases + c.rec_bases.size() - (nrec_all ? 0 : 1);            // a separator for every record that ends at a header
            nrec_all += c.rec_bases.size();
            open_len = c.rec_bases.back();
        }
    }
    if (!nrec_all) return fail(err, errlen, "no FASTA record");
    if (short_rec || open_len <= 32) return fail(err, errlen, "Length <= 32!");      // src/collect#$.c:41-45
    if (trace) fprintf(stderr, "ingest: %llu records combined after %.3f s\n", (unsigned long long)nrec_all, since());
    const uint64_t nrec = nrec_all;
    const uint64_t n = sym + 1;                              // + the final '$'
    const uint64_t nwords = ((n + 63) >> 5) + 2;
    // The words are not cleared as a whole (calloc's fresh pages: 4 KB faults from all threads at once; big_malloc: huge pages,
    // which may hold anything): the packers store every word they complete, and OR only into the first word of a chunk -- the
    // last of the chunk before -- and into the words from the final symbol on.  Those are cleared here.
    uint64_t *words = (uint64_t *)big_malloc(nwords * 8);
    uint64_t *sep = (uint64_t *)malloc(nrec * 8);
    if (!words || !sep) { free(words); free(sep); return fail(err, errlen, "out of memory"); }
    if (getenv("DEBWT_INGEST_POISON")) memset(words, 0xA5, nwords * 8);     // tests: nothing may rely on cleared memory
    for (size_t t = 0; t < nch; t++) words[ch[t].sym0 >> 5] = 0;
    for (uint64_t w = sym >> 5; w < nwords; w++) words[w] = 0;
    {
        std::vector<std::thread> th;
        for (size_t t = 1; t < nch; t++) th.emplace_back(pack, buf, std::cref(ch[t]), words, sep, opts);
        pack(buf, ch[0], words, sep, opts);
        for (auto &x : th) x.join();
    }
    sep[nrec - 1] = n - 1;                                   // '$': 'T' there and 32 'T' behind (src/collect#$.c:85-90)
    for (uint64_t j = n - 1; j < n + 32; j++) words[j >> 5] |= 3ull << ((31 - (j & 31)) << 1);
    out->words = words; out->nwords = nwords; out->n = n; out->sep = sep; out->nrec = nrec;
    if (trace) fprintf(stderr, "ingest: packed after %.3f s\n", since());
    return 0;
}

// BGZF (bgzip, the block gzip of htslib -- and any gzip file whose members all carry the 'BC' extra subfield): every member
// says how long it is (BSIZE) and how much it holds (ISIZE, its last four bytes), so the members are found by a walk over
// their headers and inflated independently, each into its place in the output, by `threads` threads (raw inflate + the
// member's CRC32, as gzread checks it).  Returns 1 when the file is not of that shape (the caller inflates it serially:
// a plain gzip stream is one member), 0 on success, -1 on a damaged member.
static int inflate_bgzf(const unsigned char *z, size_t zlen, int threads, char **out_buf, size_t *out_len) {
    std::vector<size_t> off, uoff, dlen;
    size_t p = 0, total = 0;
    while (p < zlen) {
        if (zlen - p < 28 || z[p] != 0x1f || z[p + 1] != 0x8b || z[p + 2] != 8 || z[p + 3] != 4) return 1;   // FEXTRA and nothing else
        const size_t xlen = z[p + 10] | ((size_t)z[p + 11] << 8);
        size_t q = p + 12, bsize = 0;
        const size_t xend = q + xlen;
        if (xend + 8 > zlen) return 1;
        while (q + 4 <= xend) {
            const size_t slen = z[q + 2] | ((size_t)z[q + 3] << 8);
            if (z[q] == 'B' && z[q + 1] == 'C' && slen == 2 && q + 6 <= xend) bsize = (z[q + 4] | ((size_t)z[q + 5] << 8)) + 1;
            q += 4 + slen;
        }
        if (!bsize || p + bsize > zlen || bsize < 12 + xlen + 8) return 1;
        const size_t isize = z[p + bsize - 4] | ((size_t)z[p + bsize - 3] << 8) | ((size_t)z[p + bsize - 2] << 16) | ((size_t)z[p + bsize - 1] << 24);
        off.push_back(p + 12 + xlen); dlen.push_back(bsize - (12 + xlen) - 8); uoff.push_back(total);
        total += isize;
        p += bsize;
    }
    if (off.empty()) return 1;
    char *buf = (char *)big_malloc(total + 1);
    if (!buf) return -1;
    const size_t nb = off.size();
    if (threads < 1) threads = 1;
    const size_t T = std::min<size_t>((size_t)threads, nb);
    std::vector<int> bad(T, 0);
    auto work = [&](size_t t) {
        // every member straight into its place of the one buffer (fast_inflate.h writes nothing behind the ISIZE bytes it is given)
        std::unique_ptr<fastinflate::Decoder> d(new (std::nothrow) fastinflate::Decoder());
        if (!d) { bad[t] = 1; return; }
        for (size_t b = nb * t / T; b < nb * (t + 1) / T && !bad[t]; b++) {
            const size_t want = (b + 1 < nb ? uoff[b + 1] : total) - uoff[b];
            size_t got = 0;
            int r = fastinflate::FI_DONE;
            if (want) {
                d->start(z + off[b], dlen[b], 0);
                r = d->run((uint8_t *)buf + uoff[b], 0, &got, want, ~(size_t)0);
            }
            const unsigned char *tr = z + off[b] + dlen[b];                              // CRC32, ISIZE
            const uint32_t crc = tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
            if (r != fastinflate::FI_DONE || got != want || fastinflate::crc32_fast(0, (const uint8_t *)buf + uoff[b], want) != crc) bad[t] = 1;
        }
    };
    {
        std::vector<std::thread> th;
        for (size_t t = 1; t < T; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    }
    for (int b : bad) if (b) { free(buf); return -1; }
    *out_buf = buf; *out_len = total;
    return 0;
}

int pack_fasta_file(const char *path, int threads, PackedText *out, char *err, size_t errlen, IngestOpts opts) {
    auto t0 = std::chrono::steady_clock::now();
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(err, errlen, "can not open ref file");                   // src/collect#$.c:36
    struct stat st;
    if (fstat(fd, &st) || st.st_size == 0) { close(fd); return fail(err, errlen, "empty or unreadable input"); }
    unsigned char magic[2] = {0, 0};
    if (pread(fd, magic, 2, 0) != 2) { close(fd); return fail(err, errlen, "unreadable input"); }
    int rc;
    if (magic[0] == 0x1f && magic[1] == 0x8b) {
        {   // block gzip first: its members inflate in parallel; then a one-member stream in pieces (gz_parallel.cpp)
            void *zm = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (zm != MAP_FAILED) {
                char *buf = nullptr;
                size_t len = 0;
                struct Hold { Hold() { release_hold(1); } ~Hold() { release_hold(0); } } hold;   // (buffers given up are released behind the parse)
                int br = inflate_bgzf((const unsigned char *)zm, (size_t)st.st_size, threads, &buf, &len);
                if (br == 1 && !getenv("DEBWT_GZ_SERIAL")) {
                    // several plain members (their headers are found first: a file of one member is told apart at once), else
                    // the one member in pieces
                    br = inflate_gzip_members((const unsigned char *)zm, (size_t)st.st_size, threads, &buf, &len);
                    if (br == 1) br = inflate_gzip_parallel((const unsigned char *)zm, (size_t)st.st_size, threads, &buf, &len) == 0 ? 0 : 1;
                    if (br == 1 && getenv("DEBWT_GZ_REQUIRE_PARALLEL")) {            // tests: no silent serial fall-back
                        munmap(zm, (size_t)st.st_size); close(fd);
                        return fail(err, errlen, "the parallel gzip path declined the file");
                    }
                }
                munmap(zm, (size_t)st.st_size);
                if (br < 0) { close(fd); return fail(err, errlen, "gzip stream is damaged"); }
                if (br == 0) {
                    close(fd);
                    out->seconds_read = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    auto t1 = std::chrono::steady_clock::now();
                    rc = pack_fasta_buffer(buf, len, threads, out, err, errlen, opts);
                    out->seconds_pack = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
                    void *gone = buf;
                    release_later(&gone, 1);                            // (the inflated text: 50 ms per GB that nobody has to wait for)
                    return rc;
                }
            }
        }
        // what is left (several plain members, not text, tiny, one thread): serial inflate; the parse behind it is not
        close(fd);
        gzFile f = gzopen(path, "rb");
        if (!f) return fail(err, errlen, "can not open ref file");
        gzbuffer(f, 1 << 22);
        size_t cap = (size_t)st.st_size * 4 + (1 << 20), len = 0;
        char *buf = (char *)malloc(cap);
        if (!buf) { gzclose(f); return fail(err, errlen, "out of memory"); }
        for (;;) {
            if (cap - len < (1u << 22)) {
                cap *= 2;
                char *nb = (char *)realloc(buf, cap);
                if (!nb) { free(buf); gzclose(f); return fail(err, errlen, "out of memory"); }
                buf = nb;
            }
            int r = gzread(f, buf + len, 1u << 22);
            if (r < 0) { free(buf); gzclose(f); return fail(err, errlen, "gzip stream is damaged"); }
            if (r == 0) break;
            len += (size_t)r;
        }
        {   // a stream that ends before its end (gzread hands out what it has and says so only here)
            int zerr = Z_OK;
            (void)gzerror(f, &zerr);
            if (zerr == Z_BUF_ERROR || zerr == Z_DATA_ERROR) { free(buf); gzclose(f); return fail(err, errlen, "gzip stream is cut short or damaged"); }
        }
        gzclose(f);
        out->seconds_read = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        auto t1 = std::chrono::steady_clock::now();
        rc = pack_fasta_buffer(buf, len, threads, out, err, errlen, opts);
        out->seconds_pack = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        free(buf);
        return rc;
    }
    // no MAP_POPULATE: the parser threads fault their own chunks in, in parallel
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return fail(err, errlen, "mmap failed");
    (void)madvise(m, (size_t)st.st_size, MADV_WILLNEED);
    out->seconds_read = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    auto t1 = std::chrono::steady_clock::now();
    rc = pack_fasta_buffer((const char *)m, (size_t)st.st_size, threads, out, err, errlen, opts);
    out->seconds_pack = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    munmap(m, (size_t)st.st_size);
    return rc;
}

uint64_t fasta_text_bound(const char *path) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) return 0;
    struct stat st;
    if (fstat(fd, &st) || st.st_size < 2) { close(fd); return 0; }
    const size_t zlen = (size_t)st.st_size;
    unsigned char h[18] = {0};
    if (pread(fd, h, zlen < 18 ? zlen : 18, 0) < 2) { close(fd); return 0; }
    if (!(h[0] == 0x1f && h[1] == 0x8b)) { close(fd); return (uint64_t)zlen + 1; }          // plain: a symbol takes a byte
    uint64_t bound = 0;
    if (zlen >= 28 && h[2] == 8 && h[3] == 4) {
        // BGZF: the walk of inflate_bgzf over the members' BSIZE, their ISIZE added up
        const unsigned char *z = (const unsigned char *)mmap(nullptr, zlen, PROT_READ, MAP_PRIVATE, fd, 0);
        if (z != MAP_FAILED) {
            size_t p = 0;
            bool ok = true;
            while (ok && p < zlen) {
                if (zlen - p < 28 || z[p] != 0x1f || z[p + 1] != 0x8b || z[p + 2] != 8 || z[p + 3] != 4) { ok = false; break; }
                const size_t xlen = z[p + 10] | ((size_t)z[p + 11] << 8), xend = p + 12 + xlen;
                size_t q = p + 12, bsize = 0;
                if (xend + 8 > zlen) { ok = false; break; }
                while (q + 4 <= xend) {
                    const size_t slen = z[q + 2] | ((size_t)z[q + 3] << 8);
                    if (z[q] == 'B' && z[q + 1] == 'C' && slen == 2 && q + 6 <= xend) bsize = (z[q + 4] | ((size_t)z[q + 5] << 8)) + 1;
                    q += 4 + slen;
                }
                if (!bsize || p + bsize > zlen || bsize < 12 + xlen + 8) { ok = false; break; }
                bound += z[p + bsize - 4] | ((uint64_t)z[p + bsize - 3] << 8) | ((uint64_t)z[p + bsize - 2] << 16) | ((uint64_t)z[p + bsize - 1] << 24);
                p += bsize;
            }
            munmap((void *)z, zlen);
            if (!ok) bound = 0;
        }
    }
    if (!bound && zlen >= 18 && zlen < ((size_t)1 << 30)) {
        // one member (as far as its last four bytes can say: a file of several members gives the last one's length -- the
        // caller's buffers grow when the text turns out longer)
        unsigned char t[4];
        if (pread(fd, t, 4, (off_t)zlen - 4) == 4) bound = t[0] | ((uint64_t)t[1] << 8) | ((uint64_t)t[2] << 16) | ((uint64_t)t[3] << 24);
        if (bound < zlen) bound = 0;                                                         // (not a text that deflated)
    }
    close(fd);
    return bound ? bound + 1 : 0;
}

void free_packed_text(PackedText *p) {
    if (!p) return;
    free(p->words); free(p->sep);
    p->words = nullptr; p->sep = nullptr;
}
