// fast_inflate.h -- a raw DEFLATE (RFC 1951) decoder and a CRC-32 for the gzip ingest (FASTA / gz -> packed text at
// >= 10 GB/s; the reference reads gzip through zlib's gzread on one thread, src/collect#$.c:26,34-37 / src/kseq.h).
//
// zlib's inflate decodes 0.5-0.65 GB/s of FASTA text per host thread of the MI355X box (EPYC 9575F) and its crc32 2 GB/s:
// sixteen threads cannot reach the row's rate through it.  This decoder is built for the ingest's callers (BGZF members, several
// plain members, the pieces of one member) and for what gzip makes of DNA text -- matches of 4..9 bytes, 95 % of the bytes at
// level 6, all of them at level 1; literals where the text is qualities: a 64-bit bit buffer refilled without a branch, an
// 11-bit litlen table whose entries hold up to four literals or a whole match (length and distance in one look-up), the next
// entry loaded before the refill, an 8-bit distance table, longer codes through sub-tables, copies by two unconditional 8-byte
// steps.  1.3-1.5 GB/s per thread on the same host.  Output is bytes, or 16-bit symbols for a start whose window is unknown
// (markers, see run).  It starts at any BIT of the stream with up to 32 KB of history in front of the output, stops at the end
// of the final block, at a given block boundary or at every boundary, and can be resumed when the output buffer is full (the
// symbol that did not fit is not consumed).  It writes nothing at or behind the bound it is given: members decoded side by side
// into one buffer never touch each other's bytes.  Anything malformed is an error, never a wrong byte; every caller still
// checks the member's CRC-32 and ISIZE.
// crc32_fast: the gzip polynomial by carry-less multiplication (PCLMULQDQ folding, the constants of Intel's "Fast CRC Computation
// for Generic Polynomials Using PCLMULQDQ"), zlib's crc32 for the tail and where the instruction is missing.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace fastinflate {

enum { FI_DONE = 0, FI_STOPPED = 1, FI_NEED_OUTPUT = 2, FI_ERROR = -1 };

#ifndef FI_LROOT
#define FI_LROOT 11                    // (12, a 32 KB table: 8 % slower on DNA text, the same on FASTQ -- scripts/micro/inflate_rate.cpp on the box)
#endif
constexpr int LROOT = FI_LROOT, DROOT = 8;
// A table entry (64 bits):  bits 0..7 the input bits it consumes (0: not a code); bits 8..10 how many literals it holds (root
// entries of the litlen table hold up to FOUR -- as many as their codes fit into the root index: one look-up per literal is a
// chain of load -> shift -> load, ~7 cycles); bit 11 end of block; bit 12 pointer to a sub-table (bits 32.. its offset, bits
// 16..20 its index bits); a length / distance: bits 32.. its base, bits 16..20 the number of extra bits, bits 24..28 the length
// of the code alone (bits 0..7 count both, so one shift consumes the symbol and the extra bits are cut out of the buffer as it
// was); in a literal entry bits 16..20 = the code length of the first literal alone (used while the table is built) and bits
// 32..63 the literals, first in the low byte.
// A root entry of the litlen table may also hold a whole MATCH (bit 13): gzip turns DNA text into matches of 4..9 bytes for the
// most part, whose length code, extra bit and distance code together are shorter than the root index -- then the length is
// known (bits 48..56), bits 32..46 are the distance's base, bits 16..20 the number of its extra bits, bits 24..28 where they
// begin, and bits 0..7 (<= 23) consume all of it: one look-up per match instead of two that wait for each other.
// The loop keeps the NEXT entry loaded before it refills the bit buffer: a refill only adds bits above those the look-up reads,
// and its address hangs on the entry before (the bits it consumed), which would otherwise put two loads in a row on the path
// from one symbol to the next.
constexpr uint64_t F_EOB = 1u << 11, F_SUB = 1u << 12, F_PAIR = 1u << 13, M_CNT = 7u << 8;

struct Decoder {
    uint64_t lit[(1 << LROOT) + 4608];
    uint64_t dst[(1 << DROOT) + 3840];
    const uint8_t *in = nullptr;
    size_t in_len = 0;            // readable bytes of `in`
    size_t bitpos = 0;            // next bit of the stream (absolute in `in`)
    bool in_block = false, last = false;
    int btype = 0;
    uint32_t stored_left = 0;

    void start(const uint8_t *z, size_t zlen, size_t bit) { in = z; in_len = zlen; bitpos = bit; in_block = false; last = false; }

    // canonical Huffman table from code lengths; false: over-subscribed, or incomplete (but for one code of one bit, or no code)
    static bool build(uint64_t *tab, int root, const uint8_t *len, int n, bool litlen, const uint64_t *dtab = nullptr) {
        static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        int count[16] = {0};
        for (int i = 0; i < n; i++) count[len[i]]++;
        const int used = n - count[0];
        long left = 1;
        for (int l = 1; l <= 15; l++) { left = left * 2 - count[l]; if (left < 0) return false; }
        if (used && left > 0 && (used > 1 || !count[1])) return false;   // incomplete: only ONE code of ONE bit may be (as zlib: inftrees.c "max != 1"); none at all may
        const size_t rootsz = (size_t)1 << root;
        if (left > 0 || !used) memset(tab, 0, rootsz * sizeof(uint64_t));      // 0 = no code here: an error when it is looked up (a complete code fills the root)
        if (!used) return true;
        uint16_t sorted[288];
        int offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + count[l];
        for (int i = 0; i < n; i++) if (len[i]) sorted[offs[len[i]]++] = (uint16_t)i;
        auto entry = [&](int sym, int codelen) -> uint64_t {
            if (litlen) {
                if (sym < 256) return ((uint64_t)sym << 32) | ((uint64_t)codelen << 16) | (1u << 8) | (uint64_t)codelen;
                if (sym == 256) return F_EOB | (uint64_t)codelen;
                if (sym > 285) return 0u;                         // 286, 287: never valid in a stream
                return ((uint64_t)LBASE[sym - 257] << 32) | ((uint64_t)codelen << 24) | ((uint64_t)LEXT[sym - 257] << 16) | (uint64_t)(codelen + LEXT[sym - 257]);
            }
            if (sym > 29) return 0u;
            return ((uint64_t)DBASE[sym] << 32) | ((uint64_t)codelen << 24) | ((uint64_t)DEXT[sym] << 16) | (uint64_t)(codelen + DEXT[sym]);
        };
        // the code of a symbol, bit-reversed (codes are sent most significant bit first), kept reversed and stepped reversed
        auto rev = [](uint32_t c, int l) { uint32_t r = 0; for (int i = 0; i < l; i++) { r = (r << 1) | (c & 1u); c >>= 1; } return r; };
        int maxl = 15;
        while (!count[maxl]) maxl--;
        uint8_t *maxlen = nullptr;
        if (maxl > root) {                                       // longest code behind every root prefix that has codes longer than the root
            maxlen = (uint8_t *)alloca(rootsz);
            memset(maxlen, 0, rootsz);
            uint32_t code = 0;
            for (int l = 1; l <= 15; l++) {
                if (l > root)
                    for (int c = 0; c < count[l]; c++) { const uint32_t pre = rev(code + (uint32_t)c, l) & (uint32_t)(rootsz - 1); if (maxlen[pre] < l) maxlen[pre] = (uint8_t)l; }
                code = (code + (uint32_t)count[l]) << 1;
            }
        }
        size_t next_sub = rootsz;
        uint32_t code = 0;
        int idx = 0;
        for (int l = 1; l <= maxl; l++) {
            for (int c = 0; c < count[l]; c++, idx++, code++) {
                const int sym = sorted[idx];
                const uint32_t r = rev(code, l);
                if (l <= root) {
                    const uint64_t e = entry(sym, l);
                    for (size_t k = r; k < rootsz; k += (size_t)1 << l) tab[k] = e;
                } else {
                    const uint32_t pre = r & (uint32_t)(rootsz - 1);
                    const int sb = (maxlen[pre] & 0x7F) - root;
                    if (!(maxlen[pre] & 0x80)) {                  // (the root entry itself may be a stale one of the table before)
                        maxlen[pre] |= 0x80;
                        tab[pre] = ((uint64_t)next_sub << 32) | F_SUB | ((uint64_t)sb << 16) | (uint64_t)root;
                        memset(tab + next_sub, 0, ((size_t)1 << sb) * sizeof(uint64_t));
                        next_sub += (size_t)1 << sb;
                    }
                    const size_t base = (size_t)(tab[pre] >> 32);
                    const uint64_t e = entry(sym, l - root);
                    for (size_t k = r >> root; k < ((size_t)1 << sb); k += (size_t)1 << (l - root)) tab[base + k] = e;
                }
            }
            code <<= 1;
        }
        if (litlen && dtab) {
            // a length whose code and extra bits end inside the root index, and behind it a distance code that ends there too
            // (read at the index bits that are left, zeros for the unknown ones: valid when the code is not longer than those)
            for (size_t i = 0; i < rootsz; i++) {
                const uint64_t e = tab[i];
                if ((e & (M_CNT | F_EOB | F_SUB)) || !(e & 0xFF)) continue;
                const unsigned t = (unsigned)(e & 0xFF);
                if (t >= (unsigned)root) continue;
                const uint64_t d = dtab[(i >> t) & (((size_t)1 << DROOT) - 1)];
                const unsigned dl = (unsigned)((d >> 24) & 31);
                if ((d & F_SUB) || !(d & 0xFF) || t + dl > (unsigned)root || t + (unsigned)(d & 0xFF) > 23) continue;
                const uint64_t length = (e >> 32) + ((i >> ((e >> 24) & 31)) & (((uint64_t)1 << ((e >> 16) & 31)) - 1));
                tab[i] = (length << 48) | ((d >> 32) << 32) | ((uint64_t)(t + dl) << 24) | (d & (31u << 16)) | F_PAIR | (uint64_t)(t + (unsigned)(d & 0xFF));
            }
        }
        if (litlen) {
            // several literals per root entry, in place and from the top down: the entry of the bits BEHIND a literal lies at a
            // smaller index (i >> length; i = 0 reads itself before it is written) and still is what the round before left.
            // An entry that consumes t bits depends on its low t index bits only, so the one read at (i >> t), whose upper
            // bits are zeros standing for unknown input, may be appended when it fits into the root's remaining bits.
            for (size_t i = rootsz; i-- > 0;) {                   // singles -> pairs
                const uint64_t e = tab[i];
                if (!(e & M_CNT)) continue;
                const unsigned t = (unsigned)(e & 0xFF);
                const uint64_t f = tab[i >> t];
                if ((f & M_CNT) && t + (unsigned)(f & 0xFF) <= (unsigned)root)
                    tab[i] = (e & 0xFFFF0000ull) | (2u << 8) | (t + (unsigned)(f & 0xFF)) | ((e >> 32) << 32) | ((f >> 32) << 40);
            }
            for (size_t i = rootsz; i-- > 0;) {                   // pairs -> three or four
                const uint64_t e = tab[i];
                if ((e & M_CNT) != (2u << 8)) continue;
                const unsigned t = (unsigned)(e & 0xFF);
                const uint64_t f = tab[i >> t];
                if (!(f & M_CNT)) continue;
                const unsigned tf = (unsigned)(f & 0xFF), l1f = (unsigned)((f >> 16) & 31);
                if (t + tf <= (unsigned)root)
                    tab[i] = (e & 0xFFFFFFFF0000ull) | ((2u + (unsigned)((f >> 8) & 7)) << 8) | (t + tf) | ((f >> 32) << 48);
                else if (t + l1f <= (unsigned)root)
                    tab[i] = (e & 0xFFFFFFFF0000ull) | (3u << 8) | (t + l1f) | (((f >> 32) & 0xFF) << 48);
            }
        }
        return true;
    }

    // Decodes into out[*out_pos ...) up to out_limit (bytes of `out`; nothing is written at or behind it); `hist` bytes in front
    // of out[0] are valid history.  stop_bit: a block boundary to stop at (FI_STOPPED), ~0 = none.  FI_NEED_OUTPUT: the next
    // symbol (a match, or a table entry of up to four literals) does not fit in front of out_limit; call again with more room.
    // T = uint8_t: bytes.  T = uint16_t: the same symbols in 16 bits, for a start in the middle of a stream whose window is not
    // known: the caller puts 32768 MARKERS (values >= 256 naming a window position) in front of the output as its history, and
    // copies carry them along like any other symbol (the pugz / rapidgzip way); out_pos, out_limit, hist count symbols.
    // each_block: also stop (FI_STOPPED) at every block boundary, once at least one block has ended in this call.
    template <typename T>
    int run(T *out, size_t hist, size_t *out_pos, size_t out_limit, size_t stop_bit, bool each_block = false) {
#if defined(__x86_64__)
        static const bool bmi2 = __builtin_cpu_supports("bmi2");     // shifts by a register without the detour through CL
        if (bmi2) return run_bmi2<T>(out, hist, out_pos, out_limit, stop_bit, each_block);
#endif
        return run_body<T>(out, hist, out_pos, out_limit, stop_bit, each_block);
    }
#if defined(__x86_64__)
    template <typename T>
    __attribute__((target("bmi2"))) int run_bmi2(T *out, size_t hist, size_t *out_pos, size_t out_limit, size_t stop_bit, bool each_block) {
        return run_body<T>(out, hist, out_pos, out_limit, stop_bit, each_block);
    }
#endif
    template <typename T>
    __attribute__((always_inline)) inline int run_body(T *out, size_t hist, size_t *out_pos, size_t out_limit, size_t stop_bit, bool each_block) {
        static_assert(sizeof(T) == 1 || sizeof(T) == 2, "bytes or 16-bit symbols");
        constexpr size_t PER = 8 / sizeof(T);                      // symbols per 8-byte step of a copy
        bool ended_one = false;
        const uint8_t *const in_end = in + in_len;
        const uint8_t *in_next = in + (bitpos >> 3);
        uint64_t bitbuf = 0;
        int bitcnt = 0;
        if (in_next > in_end) return FI_ERROR;
        // Bits above bitcnt may be set: they are the stream's own next bits (the 8-byte load takes more than it counts) and the
        // next refill ORs the same bits onto them.
        auto refill = [&]() {
            if (in_next + 8 <= in_end) {
                uint64_t w;
                memcpy(&w, in_next, 8);
                bitbuf |= w << bitcnt;
                in_next += (63 - bitcnt) >> 3;
                bitcnt |= 56;
            } else {
                while (bitcnt <= 56) {
                    const uint64_t b = in_next < in_end ? *in_next : 0u;     // zeros behind the data: noticed by the position check
                    bitbuf |= b << bitcnt;
                    in_next++;
                    bitcnt += 8;
                }
            }
        };
        auto position = [&]() -> size_t { return (size_t)(in_next - in) * 8 - (size_t)bitcnt; };
        refill();
        { const int skip = (int)(bitpos & 7); bitbuf >>= skip; bitcnt -= skip; }
        T *o = out + *out_pos, *const oend = out + out_limit, *const obase = out - hist;
        constexpr uint64_t LMASK = (1u << LROOT) - 1, DMASK = (1u << DROOT) - 1;
        for (;;) {
            if (!in_block) {
                const size_t here = position();
                if (here > in_len * 8) break;                     // ran behind the data
                if (last) { bitpos = here; *out_pos = (size_t)(o - out); return FI_DONE; }
                if (stop_bit != ~(size_t)0) {
                    if (here == stop_bit) { bitpos = here; *out_pos = (size_t)(o - out); return FI_STOPPED; }
                    if (here > stop_bit) break;
                }
                if (each_block && ended_one) { bitpos = here; *out_pos = (size_t)(o - out); return FI_STOPPED; }
                refill();
                last = (bitbuf & 1) != 0;
                btype = (int)((bitbuf >> 1) & 3);
                bitbuf >>= 3; bitcnt -= 3;
                if (btype == 3) break;
                if (btype == 0) {
                    const int pad = bitcnt & 7;                   // to the byte boundary of the stream
                    bitbuf >>= pad; bitcnt -= pad;
                    refill();
                    const uint32_t l = (uint32_t)(bitbuf & 0xFFFF), nl = (uint32_t)((bitbuf >> 16) & 0xFFFF);
                    bitbuf >>= 32; bitcnt -= 32;
                    if ((l ^ nl) != 0xFFFFu) break;
                    stored_left = l;
                } else if (btype == 1) {
                    uint8_t ll[288 + 32];
                    for (int i = 0; i < 144; i++) ll[i] = 8;
                    for (int i = 144; i < 256; i++) ll[i] = 9;
                    for (int i = 256; i < 280; i++) ll[i] = 7;
                    for (int i = 280; i < 288; i++) ll[i] = 8;
                    for (int i = 0; i < 32; i++) ll[288 + i] = 5;
                    if (!build(dst, DROOT, ll + 288, 32, false) || !build(lit, LROOT, ll, 288, true, dst)) break;
                } else {
                    refill();
                    const int nlen = (int)(bitbuf & 31) + 257, ndist = (int)((bitbuf >> 5) & 31) + 1, ncode = (int)((bitbuf >> 10) & 15) + 4;
                    bitbuf >>= 14; bitcnt -= 14;
                    if (nlen > 286 || ndist > 30) break;
                    static const uint8_t ORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                    uint8_t cl[19] = {0};
                    for (int i = 0; i < ncode; i++) {
                        if (bitcnt < 3) refill();
                        cl[ORD[i]] = (uint8_t)(bitbuf & 7); bitbuf >>= 3; bitcnt -= 3;
                    }
                    uint32_t ctab[128];                           // the code-length code: a 7-bit root holds every code
                    {
                        int count[8] = {0};
                        for (int i = 0; i < 19; i++) count[cl[i]]++;
                        long left = 1;
                        bool bad = false;
                        for (int l = 1; l <= 7; l++) { left = left * 2 - count[l]; if (left < 0) { bad = true; break; } }
                        if (bad || left > 0) break;               // the code-length code must be complete (zlib: "invalid code lengths set")
                        memset(ctab, 0, sizeof ctab);
                        uint32_t code = 0;
                        for (int l = 1; l <= 7; l++) {
                            for (int s = 0; s < 19; s++)
                                if (cl[s] == l) {
                                    uint32_t r = 0, c = code;
                                    for (int i = 0; i < l; i++) { r = (r << 1) | (c & 1u); c >>= 1; }
                                    for (uint32_t k = r; k < 128; k += 1u << l) ctab[k] = ((uint32_t)s << 16) | (uint32_t)l;
                                    code++;
                                }
                            code <<= 1;
                        }
                    }
                    uint8_t ll[286 + 30 + 140];
                    int i = 0;
                    bool bad = false;
                    while (i < nlen + ndist) {
                        refill();
                        const uint32_t e = ctab[bitbuf & 127];
                        const int cl_len = (int)(e & 0xFF), s = (int)(e >> 16);
                        if (!cl_len) { bad = true; break; }
                        bitbuf >>= cl_len; bitcnt -= cl_len;
                        if (s < 16) ll[i++] = (uint8_t)s;
                        else {
                            int rep;
                            uint8_t v = 0;
                            if (s == 16) { if (!i) { bad = true; break; } v = ll[i - 1]; rep = 3 + (int)(bitbuf & 3); bitbuf >>= 2; bitcnt -= 2; }
                            else if (s == 17) { rep = 3 + (int)(bitbuf & 7); bitbuf >>= 3; bitcnt -= 3; }
                            else { rep = 11 + (int)(bitbuf & 127); bitbuf >>= 7; bitcnt -= 7; }
                            if (i + rep > nlen + ndist) { bad = true; break; }
                            while (rep--) ll[i++] = v;
                        }
                    }
                    if (bad || ll[256] == 0) break;               // no end-of-block code
                    if (!build(dst, DROOT, ll + nlen, ndist, false) || !build(lit, LROOT, ll, nlen, true, dst)) break;
                }
                in_block = true;
            }
            if (btype == 0) {
                // stored bytes: whatever whole bytes the bit buffer holds first, then straight from the input
                while (stored_left && bitcnt >= 8) {
                    if (o >= oend) goto need_output;
                    *o++ = (T)(uint8_t)bitbuf; bitbuf >>= 8; bitcnt -= 8; stored_left--;
                }
                if (stored_left) {
                    // bitcnt < 8 here and the stream is byte aligned: no bits are pending
                    const uint8_t *src = in_next - (bitcnt >> 3);
                    size_t take = stored_left;
                    if (src > in_end || (size_t)(in_end - src) < take) break;     // the data ends inside the block
                    if ((size_t)(oend - o) < take) take = (size_t)(oend - o);
                    if (sizeof(T) == 1) memcpy(o, src, take); else for (size_t i = 0; i < take; i++) o[i] = (T)src[i];
                    o += take; stored_left -= (uint32_t)take;
                    in_next = src + take; bitbuf = 0; bitcnt = 0;
                    if (stored_left) goto need_output;
                    refill();
                }
                in_block = false; ended_one = true;
                continue;
            }
            // compressed block.  The loop with room on both sides first: one iteration reads at most four refills (<= 32 bytes)
            // and writes at most 3 x 4 literals, one more, a match of 258 and the eight-byte steps' overrun.
            if (in_end - in_next >= 64 && (size_t)(oend - o) >= 320) {
                const uint8_t *const in_fast = in_end - 64;
                T *const o_fast = oend - 320;
#define FI_REFILL() do { uint64_t w_; memcpy(&w_, in_next, 8); bitbuf |= w_ << bitcnt; in_next += (63 - bitcnt) >> 3; bitcnt |= 56; } while (0)
// sixteen bytes whatever the length (most matches of DNA text are shorter), then by eight
#define FI_COPY(len_, off_) do { \
                    const T *src_ = o - (off_); \
                    T *const oe_ = o + (len_); \
                    if ((off_) >= PER) { \
                        uint64_t w_; \
                        memcpy(&w_, src_, 8); memcpy(o, &w_, 8); \
                        memcpy(&w_, src_ + PER, 8); memcpy(o + PER, &w_, 8); \
                        if ((len_) > 2 * PER) { o += 2 * PER; src_ += 2 * PER; do { memcpy(&w_, src_, 8); memcpy(o, &w_, 8); o += PER; src_ += PER; } while (o < oe_); } \
                    } \
                    else if ((off_) == 1) { const T v_ = *src_; while (o < oe_) *o++ = v_; } \
                    else { while (o < oe_) *o++ = *src_++; } \
                    o = oe_; } while (0)
// the literals of an entry: four bytes (or four 16-bit symbols) stored whatever their number
#define FI_LITS(e_) do { \
                    const uint32_t v_ = (uint32_t)((e_) >> 32); \
                    if (sizeof(T) == 1) memcpy(o, &v_, 4); \
                    else { const uint64_t x_ = (uint64_t)(v_ & 0xFFu) | ((uint64_t)(v_ & 0xFF00u) << 8) | ((uint64_t)(v_ & 0xFF0000u) << 16) | ((uint64_t)(v_ & 0xFF000000u) << 24); memcpy(o, &x_, 8); } \
                    o += ((e_) >> 8) & 7; bitbuf >>= ((e_) & 0xFF); bitcnt -= (int)((e_) & 0xFF); } while (0)
                FI_REFILL();
                uint64_t e = lit[bitbuf & LMASK];
                while (in_next <= in_fast && o <= o_fast) {
                    if (e & M_CNT) {
                        FI_LITS(e); e = lit[bitbuf & LMASK];
                        if (e & M_CNT) {
                            FI_LITS(e); e = lit[bitbuf & LMASK];
                            if (e & M_CNT) { FI_LITS(e); e = lit[bitbuf & LMASK]; FI_REFILL(); continue; }      // (>= 23 bits were left)
                        }
                    }
                    if (e & F_PAIR) {                              // a whole match: <= 23 bits, and >= 34 are there
                        const uint32_t len = (uint32_t)(e >> 48);
                        const size_t off = (size_t)((e >> 32) & 0xFFFF) + (size_t)((bitbuf >> ((e >> 24) & 31)) & ((1u << ((e >> 16) & 31)) - 1u));
                        bitbuf >>= (e & 0xFF); bitcnt -= (int)(e & 0xFF);
                        if (off > (size_t)(o - obase)) goto fail;
                        e = lit[bitbuf & LMASK];                   // (>= 11 bits are left: 56 - 2 x 11 - 23)
                        FI_REFILL();
                        FI_COPY(len, off);
                        continue;
                    }
                    // not a literal at the root; >= 34 bits are in the buffer
                    if (e & F_SUB) {
                        bitbuf >>= LROOT; bitcnt -= LROOT;
                        e = lit[(size_t)(e >> 32) + (size_t)(bitbuf & ((1u << ((e >> 16) & 31)) - 1u))];
                        if (e & M_CNT) { FI_LITS(e); e = lit[bitbuf & LMASK]; FI_REFILL(); continue; }
                    }
                    if (e & F_EOB) { bitbuf >>= (e & 0xFF); bitcnt -= (int)(e & 0xFF); in_block = false; ended_one = true; break; }
                    if (!(e & 0xFF)) goto fail;                    // not a code
                    const uint32_t len = (uint32_t)(e >> 32) + (uint32_t)((bitbuf >> ((e >> 24) & 31)) & ((1u << ((e >> 16) & 31)) - 1u));
                    bitbuf >>= (e & 0xFF); bitcnt -= (int)(e & 0xFF);
                    FI_REFILL();
                    uint64_t d = dst[bitbuf & DMASK];
                    if (d & F_SUB) { bitbuf >>= DROOT; bitcnt -= DROOT; d = dst[(size_t)(d >> 32) + (size_t)(bitbuf & ((1u << ((d >> 16) & 31)) - 1u))]; }
                    if (!(d & 0xFF)) goto fail;
                    const size_t off = (size_t)(d >> 32) + (size_t)((bitbuf >> ((d >> 24) & 31)) & ((1u << ((d >> 16) & 31)) - 1u));
                    bitbuf >>= (d & 0xFF); bitcnt -= (int)(d & 0xFF);
                    if (off > (size_t)(o - obase)) goto fail;     // reaches in front of the history
                    e = lit[bitbuf & LMASK];                       // (>= 28 bits are left; under way while the bytes are copied)
                    FI_REFILL();
                    FI_COPY(len, off);
                }
#undef FI_REFILL
#undef FI_LITS
#undef FI_COPY
                if (!in_block) continue;
            }
            // the careful loop: near the end of the input or of the output, symbol by symbol, byte-exact
            for (;;) {
                const uint8_t *const save_in = in_next;
                const uint64_t save_buf = bitbuf;
                const int save_cnt = bitcnt;
                refill();
                uint64_t e = lit[bitbuf & LMASK];
                if (e & F_SUB) { bitbuf >>= LROOT; bitcnt -= LROOT; e = lit[(size_t)(e >> 32) + (size_t)(bitbuf & ((1u << ((e >> 16) & 31)) - 1u))]; }
                const uint64_t at_e = bitbuf;
                bitbuf >>= (e & 0xFF); bitcnt -= (int)(e & 0xFF);
                if (position() > in_len * 8) goto fail;            // the symbol was made of the zeros behind the data: cut short
                if (e & M_CNT) {
                    const unsigned c = (unsigned)((e >> 8) & 7);
                    if ((size_t)(oend - o) < c) { in_next = save_in; bitbuf = save_buf; bitcnt = save_cnt; goto need_output; }
                    uint32_t v = (uint32_t)(e >> 32);
                    for (unsigned i = 0; i < c; i++, v >>= 8) *o++ = (T)(uint8_t)v;
                    if (in_end - in_next >= 64 && (size_t)(oend - o) >= 320) break;      // (room again: a caller gave more output)
                    continue;
                }
                if (e & F_EOB) { in_block = false; ended_one = true; break; }
                if (!(e & 0xFF)) goto fail;                        // not a code
                if (e & F_PAIR) {
                    const uint32_t len = (uint32_t)(e >> 48);
                    const size_t off = (size_t)((e >> 32) & 0xFFFF) + (size_t)((at_e >> ((e >> 24) & 31)) & ((1u << ((e >> 16) & 31)) - 1u));
                    if (off > (size_t)(o - obase)) goto fail;
                    if ((size_t)(oend - o) < len) { in_next = save_in; bitbuf = save_buf; bitcnt = save_cnt; goto need_output; }
                    const T *src = o - off;
                    for (uint32_t i = 0; i < len; i++) o[i] = src[i];
                    o += len;
                } else {
                    const uint32_t len = (uint32_t)(e >> 32) + (uint32_t)((at_e >> ((e >> 24) & 31)) & ((1u << ((e >> 16) & 31)) - 1u));
                    refill();
                    uint64_t d = dst[bitbuf & DMASK];
                    if (d & F_SUB) { bitbuf >>= DROOT; bitcnt -= DROOT; d = dst[(size_t)(d >> 32) + (size_t)(bitbuf & ((1u << ((d >> 16) & 31)) - 1u))]; }
                    if (!(d & 0xFF)) goto fail;
                    const size_t off = (size_t)(d >> 32) + (size_t)((bitbuf >> ((d >> 24) & 31)) & ((1u << ((d >> 16) & 31)) - 1u));
                    bitbuf >>= (d & 0xFF); bitcnt -= (int)(d & 0xFF);
                    if (off > (size_t)(o - obase)) goto fail;     // reaches in front of the history
                    if ((size_t)(oend - o) < len) { in_next = save_in; bitbuf = save_buf; bitcnt = save_cnt; goto need_output; }
                    const T *src = o - off;
                    for (uint32_t i = 0; i < len; i++) o[i] = src[i];
                    o += len;
                }
            }
        }
    fail:
        return FI_ERROR;
    need_output:
        bitpos = position();
        *out_pos = (size_t)(o - out);
        return FI_NEED_OUTPUT;
    }
};

// ---- CRC-32 (gzip) ---------------------------------------------------------------------------------------------------------
#if defined(__x86_64__)
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_pclmul(uint32_t crc, const uint8_t *p, size_t len) {
    // len >= 64, a multiple of 16.  Folding by four 128-bit lanes; constants for the reflected polynomial 0xEDB88320:
    // x^(4*128+32) mod P, x^(4*128-32) mod P (fold by 512 bits), x^(128+32), x^(128-32) (fold by 128), x^64 mod P, then Barrett.
    const __m128i k1k2 = _mm_set_epi64x(0x00000001c6e41596LL, 0x0000000154442bd4LL);
    const __m128i k3k4 = _mm_set_epi64x(0x00000000ccaa009eLL, 0x00000001751997d0LL);
    const __m128i k5 = _mm_set_epi64x(0, 0x0000000163cd6124LL);
    const __m128i poly = _mm_set_epi64x(0x00000001F7011641LL, 0x00000001DB710641LL);
    __m128i x1 = _mm_loadu_si128((const __m128i *)(p + 0)), x2 = _mm_loadu_si128((const __m128i *)(p + 16));
    __m128i x3 = _mm_loadu_si128((const __m128i *)(p + 32)), x4 = _mm_loadu_si128((const __m128i *)(p + 48));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    p += 64; len -= 64;
    while (len >= 64) {
        __m128i t1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), t2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        __m128i t3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), t4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11); x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11); x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, t1), _mm_loadu_si128((const __m128i *)(p + 0)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, t2), _mm_loadu_si128((const __m128i *)(p + 16)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, t3), _mm_loadu_si128((const __m128i *)(p + 32)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, t4), _mm_loadu_si128((const __m128i *)(p + 48)));
        p += 64; len -= 64;
    }
#define FI_FOLD(a, b) _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(a, k3k4, 0x11), _mm_clmulepi64_si128(a, k3k4, 0x00)), b)
    x1 = FI_FOLD(x1, x2); x1 = FI_FOLD(x1, x3); x1 = FI_FOLD(x1, x4);
    while (len >= 16) { const __m128i nx = _mm_loadu_si128((const __m128i *)p); x1 = FI_FOLD(x1, nx); p += 16; len -= 16; }
#undef FI_FOLD
    // 128 -> 64 bits
    __m128i t = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
    const __m128i mask32 = _mm_set_epi32(0, 0, 0, -1);
    t = _mm_clmulepi64_si128(_mm_and_si128(x1, mask32), k5, 0x00);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 4), t);
    // Barrett reduction 64 -> 32
    t = _mm_clmulepi64_si128(_mm_and_si128(x1, mask32), poly, 0x10);
    t = _mm_clmulepi64_si128(_mm_and_si128(t, mask32), poly, 0x00);
    x1 = _mm_xor_si128(x1, t);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

// crc = the CRC so far as zlib counts it (0 for none); returns the CRC of the data appended
inline uint32_t crc32_fast(uint32_t crc, const uint8_t *p, size_t len) {
#if defined(__x86_64__)
    static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (have && len >= 128) {
        const size_t body = len & ~(size_t)15;
        crc = ~crc32_pclmul(~crc, p, body);
        p += body; len -= body;
    }
#endif
    while (len) {
        const size_t m = len < ((size_t)1 << 30) ? len : ((size_t)1 << 30);
        crc = (uint32_t)::crc32(crc, p, (uInt)m);
        p += m; len -= m;
    }
    return crc;
}

}  // namespace fastinflate
