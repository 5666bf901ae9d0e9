// fasta_host.cpp -- multi-threaded FASTA -> 2-bit packed text.
//
// Replaces the reference's single-threaded ingest (kseq.h + zlib, one base at a time into `reference`,
// src/collect#$.c:34-90): the file is mapped (or inflated, when gzip), cut into one chunk per thread at arbitrary
// byte offsets, and parsed in two parallel passes -- (1) census: bases and header starts per chunk, (2) pack: every
// thread writes its bases, and the separators of the records that end in its chunk, at their final 2-bit positions.
// A chunk start that falls inside a header line is recognised by looking back to the start of its line.
// Layout, alphabet and checks are the reference's: A0 C1 G2 T3 either case (src/main.c:18-23), 'T' at every
// separator and 32 'T' behind the end (src/collect#$.c:78-90), every record longer than 32 bases (:41-45).
#include "fasta_host.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
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

struct Chunk {
    size_t beg = 0, end = 0;
    bool starts_in_header = false;
    uint64_t head_bases = 0;              // bases before the chunk's first header start (they belong to an earlier record)
    std::vector<uint64_t> rec_bases;      // per header start in the chunk: bases from there to the next start / chunk end
    // filled by the serial combine
    uint64_t sym0 = 0;                    // text position of the chunk's first symbol
    uint64_t rec0 = 0;                    // records started before the chunk
    long bad_at = -1;                     // offset of the first invalid character
};

// is the byte at `pos` inside a header line?  (the line's first character is '>')
bool in_header_at(const char *buf, size_t pos) {
    const void *nl = pos ? memrchr(buf, '\n', pos) : nullptr;          // last newline before pos
    const size_t ls = nl ? (size_t)((const char *)nl - buf) + 1 : 0;
    return buf[ls] == '>' && pos > ls;      // pos == ls: the '>' itself is seen by the chunk's own scan
}

// Walks the lines of a chunk: on_header() at every header start, on_seq(ptr, len) for every (piece of a) sequence
// line.  Newlines are found with memchr; a header that runs past the chunk end is finished by the next chunk, which
// knows that it starts inside one.
template <class H, class S>
void scan_lines(const char *buf, const Chunk &c, H on_header, S on_seq) {
    size_t i = c.beg;
    if (c.starts_in_header) {
        const void *nl = memchr(buf + i, '\n', c.end - i);
        if (!nl) return;
        i = (size_t)((const char *)nl - buf) + 1;
    }
    while (i < c.end) {
        const void *nl = memchr(buf + i, '\n', c.end - i);
        const size_t e = nl ? (size_t)((const char *)nl - buf) : c.end;
        if (buf[i] == '>' && (i == 0 || buf[i - 1] == '\n')) on_header();
        else if (e > i) on_seq(buf + i, e - i);
        i = e + 1;
    }
}

// pass 1: bases per record piece; a line of nothing but ACGTacgt is counted by its length
void census(const char *buf, Chunk &c) {
    uint64_t cur = 0;
    bool any = false;
    scan_lines(buf, c,
        [&]() {
            if (any) c.rec_bases.push_back(cur); else c.head_bases = cur;
            any = true; cur = 0;
        },
        [&](const char *p, size_t len) {
            uint8_t acc = 0;
            for (size_t j = 0; j < len; j++) acc |= LUT.t[(unsigned char)p[j]];
            if (!(acc & 0x80)) { cur += len; return; }
            for (size_t j = 0; j < len; j++) {                 // a line with white space in it, or an invalid character
                const uint8_t v = LUT.t[(unsigned char)p[j]];
                if (v <= 3) cur++;
                else if (v == C_BAD && c.bad_at < 0) c.bad_at = (long)(p + j - buf);
            }
        });
    if (any) c.rec_bases.push_back(cur); else c.head_bases = cur;
}

// pass 2: symbols of the chunk at their text positions; a header start closes the record before it with a 'T'.
// The word under construction is shifted left two bits per symbol; the first and the last word of a chunk may be
// shared with the neighbouring chunks and are OR-ed in atomically, the words between belong to this chunk alone.
void pack(const char *buf, const Chunk &c, uint64_t *words, uint64_t *sep) {
    uint64_t pos = c.sym0, rec = c.rec0;
    const uint64_t w_first = pos >> 5;
    uint64_t acc = 0;
    auto flush_full = [&]() {                 // pos is a multiple of 32 here: word pos/32 - 1 is complete
        const uint64_t w = (pos >> 5) - 1;
        if (w == w_first) __atomic_fetch_or(&words[w], acc, __ATOMIC_RELAXED); else words[w] = acc;
        acc = 0;
    };
    auto put = [&](uint64_t code) {
        acc = (acc << 2) | code;
        if ((++pos & 31) == 0) flush_full();
    };
    scan_lines(buf, c,
        [&]() {
            if (rec > 0) { sep[rec - 1] = pos; put(3); }        // separator of the record that ends here
            rec++;
        },
        [&](const char *p, size_t len) {
            uint8_t any = 0;
            for (size_t t = 0; t < len; t++) any |= LUT.t[(unsigned char)p[t]];
            if (any & 0x80) {                 // white space inside the line (pass 1 has rejected anything else)
                for (size_t t = 0; t < len; t++) { const uint8_t v = LUT.t[(unsigned char)p[t]]; if (v <= 3) put(v); }
                return;
            }
            // symbols up to the next word boundary, whole words of 32 characters, the rest
            size_t j = 0;
            for (; j < len && (pos & 31); j++) put(LUT.t[(unsigned char)p[j]]);
            for (; j + 32 <= len; j += 32) {
                uint64_t w = 0;
                for (int t = 0; t < 32; t++) w = (w << 2) | LUT.t[(unsigned char)p[j + t]];
                acc = w; pos += 32; flush_full();
            }
            for (; j < len; j++) put(LUT.t[(unsigned char)p[j]]);
        });
    // the last, partial word of the chunk is shared with the next chunk
    if (pos & 31) __atomic_fetch_or(&words[pos >> 5], acc << ((32 - (pos & 31)) << 1), __ATOMIC_RELAXED);
}

int fail(char *err, size_t errlen, const std::string &msg) {
    if (err && errlen) snprintf(err, errlen, "%s", msg.c_str());
    return -1;
}

}  // namespace

int pack_fasta_buffer(const char *buf, size_t len, int threads, PackedText *out, char *err, size_t errlen) {
    if (threads < 1) threads = 1;
    if (len == 0) return fail(err, errlen, "empty input");
    if (buf[0] == '@') return fail(err, errlen, "FASTQ input is not supported (FASTA expected)");
    size_t nch = std::min<size_t>((size_t)threads, std::max<size_t>(1, len >> 16));
    std::vector<Chunk> ch(nch);
    for (size_t t = 0; t < nch; t++) {
        ch[t].beg = len / nch * t;
        ch[t].end = t + 1 == nch ? len : len / nch * (t + 1);
        ch[t].starts_in_header = ch[t].beg > 0 && in_header_at(buf, ch[t].beg);
    }
    {
        std::vector<std::thread> th;
        for (size_t t = 1; t < nch; t++) th.emplace_back(census, buf, std::ref(ch[t]));
        census(buf, ch[0]);
        for (auto &x : th) x.join();
    }
    // serial combine: record lengths, chunk offsets
    std::vector<uint64_t> reclen;
    uint64_t sym = 0;
    for (size_t t = 0; t < nch; t++) {
        Chunk &c = ch[t];
        if (c.bad_at >= 0) {
            char m[160];
            snprintf(m, sizeof m, "character 0x%02x at byte %ld is not one of ACGTacgt (see otherTool/transferN for N)",
                     (unsigned)(unsigned char)buf[c.bad_at], c.bad_at);
            return fail(err, errlen, m);
        }
        c.sym0 = sym; c.rec0 = reclen.size();
        if (c.head_bases) {
            if (reclen.empty()) return fail(err, errlen, "sequence before the first header");
            reclen.back() += c.head_bases;
        }
        sym += c.head_bases;
        for (uint64_t b : c.rec_bases) {
            if (!reclen.empty()) sym++;                     // separator of the record that ends at this header
            reclen.push_back(b);
            sym += b;
        }
    }
    if (reclen.empty()) return fail(err, errlen, "no FASTA record");
    for (uint64_t b : reclen)
        if (b <= 32) return fail(err, errlen, "Length <= 32!");                      // src/collect#$.c:41-45
    const uint64_t nrec = reclen.size();
    const uint64_t n = sym + 1;                              // + the final '$'
    const uint64_t nwords = ((n + 63) >> 5) + 2;
    uint64_t *words = (uint64_t *)calloc(nwords, 8);
    uint64_t *sep = (uint64_t *)malloc(nrec * 8);
    if (!words || !sep) { free(words); free(sep); return fail(err, errlen, "out of memory"); }
    {
        std::vector<std::thread> th;
        for (size_t t = 1; t < nch; t++) th.emplace_back(pack, buf, std::cref(ch[t]), words, sep);
        pack(buf, ch[0], words, sep);
        for (auto &x : th) x.join();
    }
    sep[nrec - 1] = n - 1;                                   // '$': 'T' there and 32 'T' behind (src/collect#$.c:85-90)
    for (uint64_t j = n - 1; j < n + 32; j++) words[j >> 5] |= 3ull << ((31 - (j & 31)) << 1);
    out->words = words; out->nwords = nwords; out->n = n; out->sep = sep; out->nrec = nrec;
    return 0;
}

int pack_fasta_file(const char *path, int threads, PackedText *out, char *err, size_t errlen) {
    auto t0 = std::chrono::steady_clock::now();
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(err, errlen, "can not open ref file");                   // src/collect#$.c:36
    struct stat st;
    if (fstat(fd, &st) || st.st_size == 0) { close(fd); return fail(err, errlen, "empty or unreadable input"); }
    unsigned char magic[2] = {0, 0};
    if (pread(fd, magic, 2, 0) != 2) { close(fd); return fail(err, errlen, "unreadable input"); }
    int rc;
    if (magic[0] == 0x1f && magic[1] == 0x8b) {
        // gzip: inflate is serial; the parse behind it is not
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
        gzclose(f);
        out->seconds_read = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        auto t1 = std::chrono::steady_clock::now();
        rc = pack_fasta_buffer(buf, len, threads, out, err, errlen);
        out->seconds_pack = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        free(buf);
        return rc;
    }
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return fail(err, errlen, "mmap failed");
    out->seconds_read = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    auto t1 = std::chrono::steady_clock::now();
    rc = pack_fasta_buffer((const char *)m, (size_t)st.st_size, threads, out, err, errlen);
    out->seconds_pack = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    munmap(m, (size_t)st.st_size);
    return rc;
}

void free_packed_text(PackedText *p) {
    if (!p) return;
    free(p->words); free(p->sep);
    p->words = nullptr; p->sep = nullptr;
}
