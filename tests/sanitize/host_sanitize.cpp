// host_sanitize.cpp -- the host-side C++ of the product (FASTA ingest, special-region module, synthetic-text generator)
// built WITHOUT HIP under AddressSanitizer + UndefinedBehaviorSanitizer and driven over the inputs the CPU tests use
// ("build runs ASan/UBSan on its CPU restatement"; the reference's own hazards it must not repeat are
// listed at /root/reference/src/getKmer.c:51-52 and src/generateSP.c:351-369).  CPU only -- never run on the GPU box.
// Build + run: make -C tests/sanitize   (log -> profiles/r03_sanitizers.txt)
#include "../../debwt_amd/csrc/fast_inflate.h"
#include "../../debwt_amd/csrc/fasta_host.h"
#include "../../debwt_amd/csrc/special_host.h"
#include "../../include/debwt_synth.h"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } } while (0)

static uint64_t rng_state = 0x5EEDBA5Eull;
static uint64_t rnd() { uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

static std::string fasta_of(const std::vector<std::string> &recs, size_t width, bool crlf, bool lower) {
    std::string s;
    for (size_t r = 0; r < recs.size(); r++) {
        s += ">rec" + std::to_string(r) + " some description\n";
        for (size_t i = 0; i < recs[r].size(); i += width) {
            std::string ln = recs[r].substr(i, width);
            if (lower) for (auto &c : ln) c = (char)tolower(c);
            s += ln; s += crlf ? "\r\n" : "\n";
        }
    }
    return s;
}

static uint64_t digest_tables(const SpecialTables &t) {
    uint64_t h = 7;
    auto mix = [&](uint64_t v) { h = (h ^ v) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; };
    for (auto v : t.pos) mix(v);
    for (auto v : t.key) mix(v);
    for (auto v : t.chr) mix(v);
    for (auto v : t.branch) mix(v);
    for (auto v : t.head_keys) mix(v);
    for (auto v : t.tail_facts) mix(v);
    return h;
}

int main() {
    // ---- FASTA ingest: line widths, CRLF, lower case, 1..8 threads, gzip, IUPAC replacement, the error paths ----------
    std::vector<std::string> recs;
    const char *acgt = "ACGT";
    for (int r = 0; r < 300; r++) {
        size_t len = 33 + rnd() % 3000;
        std::string s(len, 'A');
        for (auto &c : s) c = acgt[rnd() & 3];
        if (r % 7 == 0) s.replace(s.size() - 33, 33, std::string(33, 'T'));       // records ending in T...T
        if (r % 11 == 0 && r) s = recs[r - 1];                                    // duplicates
        recs.push_back(s);
    }
    PackedText ref{};
    char err[256];
    {
        std::string fa = fasta_of(recs, 60, false, false);
        CHECK(pack_fasta_buffer(fa.data(), fa.size(), 1, &ref, err, sizeof err) == 0);
        CHECK(ref.nrec == recs.size());
    }
    for (size_t width : {1ul, 7ul, 60ul, 61ul, 100000ul})
        for (int threads : {1, 2, 3, 8})
            for (int variant = 0; variant < 3; variant++) {
                std::string fa = fasta_of(recs, width, variant == 1, variant == 2);
                PackedText p{};
                CHECK(pack_fasta_buffer(fa.data(), fa.size(), threads, &p, err, sizeof err) == 0);
                CHECK(p.n == ref.n && p.nrec == ref.nrec && p.nwords == ref.nwords);
                CHECK(p.n == ref.n && !memcmp(p.words, ref.words, ref.nwords * 8) && !memcmp(p.sep, ref.sep, ref.nrec * 8));
                free_packed_text(&p);
            }
    {   // gzip through a file
        std::string fa = fasta_of(recs, 70, false, false);
        const char *path = "/tmp/debwt_sanitize.fa.gz";
        gzFile g = gzopen(path, "wb");
        CHECK(g && gzwrite(g, fa.data(), (unsigned)fa.size()) == (int)fa.size());
        gzclose(g);
        PackedText p{};
        CHECK(pack_fasta_file(path, 4, &p, err, sizeof err) == 0);
        CHECK(p.n == ref.n && !memcmp(p.words, ref.words, ref.nwords * 8));
        free_packed_text(&p);
        remove(path);
        CHECK(pack_fasta_file("/nonexistent/x.fa", 2, &p, err, sizeof err) != 0);
    }
    {   // a one-member gzip file in pieces (gz_parallel.cpp): block starts found by trial, marker heads, zlib hand-over,
        // windows, CRC -- compression levels 1 (markers never clear), 6 and 9, pieces of 16 KB and 64 KB; then damaged and cut short
        std::string big;
        {
            std::vector<std::string> rr;
            for (int r = 0; r < 6; r++) { std::string q(300000 + rnd() % 200000, 'A'); for (auto &c : q) c = acgt[rnd() & 3]; rr.push_back(q); }
            big = fasta_of(rr, 80, false, false);
        }
        PackedText want{};
        CHECK(pack_fasta_buffer(big.data(), big.size(), 2, &want, err, sizeof err) == 0);
        const char *path = "/tmp/debwt_sanitize_one_member.fa.gz";
        setenv("DEBWT_GZ_REQUIRE_PARALLEL", "1", 1);
        for (int level : {1, 6, 9})
            for (const char *piece : {"16384", "65536"}) {
                char mode[8];
                snprintf(mode, sizeof mode, "wb%d", level);
                gzFile g = gzopen(path, mode);
                CHECK(g && gzwrite(g, big.data(), (unsigned)big.size()) == (int)big.size());
                gzclose(g);
                setenv("DEBWT_GZ_PIECE_BYTES", piece, 1);
                for (int threads : {2, 8}) {
                    PackedText p{};
                    CHECK(pack_fasta_file(path, threads, &p, err, sizeof err) == 0);
                    CHECK(p.n == want.n && p.nrec == want.nrec && !memcmp(p.words, want.words, want.nwords * 8));
                    free_packed_text(&p);
                }
            }
        unsetenv("DEBWT_GZ_REQUIRE_PARALLEL");
        {   // one byte flipped in the middle, and the last 20 KB missing: errors, whichever path notices
            FILE *f = fopen(path, "rb");
            std::string z;
            char tmp[65536];
            size_t got;
            while ((got = fread(tmp, 1, sizeof tmp, f)) > 0) z.append(tmp, got);
            fclose(f);
            std::string bad = z;
            bad[bad.size() / 2] ^= 0x5A;
            f = fopen(path, "wb"); fwrite(bad.data(), 1, bad.size(), f); fclose(f);
            PackedText p{};
            CHECK(pack_fasta_file(path, 4, &p, err, sizeof err) != 0);
            f = fopen(path, "wb"); fwrite(z.data(), 1, z.size() - 20000, f); fclose(f);
            CHECK(pack_fasta_file(path, 4, &p, err, sizeof err) != 0);
        }
        unsetenv("DEBWT_GZ_PIECE_BYTES");
        free_packed_text(&want);
        remove(path);
    }
    {   // several PLAIN gzip members (gz_parallel.cpp, inflate_gzip_members): many members (every candidate on a thread of its
        // own) and few (one after the other, each in pieces), a damaged member, bytes behind the last member
        std::vector<std::string> rr;
        for (int r = 0; r < 9; r++) { std::string q(120000 + rnd() % 90000, 'A'); for (auto &c : q) c = acgt[rnd() & 3]; rr.push_back(q); }
        const std::string whole = fasta_of(rr, 70, false, false);
        PackedText want{};
        CHECK(pack_fasta_buffer(whole.data(), whole.size(), 2, &want, err, sizeof err) == 0);
        std::vector<size_t> cuts;                              // record starts
        for (size_t i = 0; i < whole.size(); i++) if (whole[i] == '>') cuts.push_back(i);
        cuts.push_back(whole.size());
        const char *path = "/tmp/debwt_sanitize_members.fa.gz";
        setenv("DEBWT_GZ_REQUIRE_PARALLEL", "1", 1);
        setenv("DEBWT_GZ_PIECE_BYTES", "16384", 1);
        for (size_t per : {(size_t)1, (size_t)5}) {               // 9 members, 2 members
            remove(path);
            std::string z;
            for (size_t a = 0; a + 1 < cuts.size(); a += per) {
                const size_t b = std::min(a + per, cuts.size() - 1);
                const char *one = "/tmp/debwt_sanitize_member_one.gz";
                gzFile g = gzopen(one, "wb6");
                CHECK(g && gzwrite(g, whole.data() + cuts[a], (unsigned)(cuts[b] - cuts[a])) == (int)(cuts[b] - cuts[a]));
                gzclose(g);
                FILE *f = fopen(one, "rb");
                char tmp[65536];
                size_t got;
                while ((got = fread(tmp, 1, sizeof tmp, f)) > 0) z.append(tmp, got);
                fclose(f);
                remove(one);
            }
            FILE *f = fopen(path, "wb"); CHECK(f && fwrite(z.data(), 1, z.size(), f) == z.size()); fclose(f);
            for (int threads : {2, 8}) {
                PackedText p{};
                CHECK(pack_fasta_file(path, threads, &p, err, sizeof err) == 0);
                CHECK(p.n == want.n && p.nrec == want.nrec && !memcmp(p.words, want.words, want.nwords * 8));
                free_packed_text(&p);
            }
            std::string bad = z;                                    // damage: no path may hand out a wrong text
            bad[bad.size() / 3] ^= 0x5A;
            f = fopen(path, "wb"); fwrite(bad.data(), 1, bad.size(), f); fclose(f);
            unsetenv("DEBWT_GZ_REQUIRE_PARALLEL");
            PackedText p{};
            CHECK(pack_fasta_file(path, 4, &p, err, sizeof err) != 0);
            z.append(64, '\0');                                     // bytes behind the last member: the serial path's case
            f = fopen(path, "wb"); fwrite(z.data(), 1, z.size(), f); fclose(f);
            CHECK(pack_fasta_file(path, 4, &p, err, sizeof err) == 0);
            CHECK(p.n == want.n && !memcmp(p.words, want.words, want.nwords * 8));
            free_packed_text(&p);
            setenv("DEBWT_GZ_REQUIRE_PARALLEL", "1", 1);
        }
        unsetenv("DEBWT_GZ_REQUIRE_PARALLEL");
        unsetenv("DEBWT_GZ_PIECE_BYTES");
        free_packed_text(&want);
        remove(path);
    }
    {   // block gzip (BGZF: members with the 'BC' subfield, inflated in parallel), whole and with a damaged member
        std::string fa = fasta_of(recs, 70, false, false);
        const char *path = "/tmp/debwt_sanitize_bgzf.fa.gz";
        std::string z;
        const size_t B = 4000;
        for (size_t a = 0; a <= fa.size(); a += B) {
            const size_t len = a < fa.size() ? std::min(B, fa.size() - a) : 0;
            std::vector<unsigned char> body(len + len / 8 + 64);
            z_stream zs; memset(&zs, 0, sizeof zs);
            CHECK(deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) == Z_OK);
            zs.next_in = (Bytef *)fa.data() + a; zs.avail_in = (uInt)len; zs.next_out = body.data(); zs.avail_out = (uInt)body.size();
            CHECK(deflate(&zs, Z_FINISH) == Z_STREAM_END);
            const size_t blen = body.size() - zs.avail_out;
            deflateEnd(&zs);
            const unsigned bsize = (unsigned)(12 + 6 + blen + 8) - 1;
            const unsigned char hd[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (unsigned char)(bsize & 255), (unsigned char)(bsize >> 8)};
            z.append((const char *)hd, 18); z.append((const char *)body.data(), blen);
            const uLong crc = crc32(crc32(0L, Z_NULL, 0), (const Bytef *)fa.data() + a, (uInt)len);
            for (int i = 0; i < 4; i++) z.push_back((char)((crc >> (8 * i)) & 255));
            for (int i = 0; i < 4; i++) z.push_back((char)((len >> (8 * i)) & 255));
            if (!len) break;
        }
        FILE *f = fopen(path, "wb"); CHECK(f && fwrite(z.data(), 1, z.size(), f) == z.size()); fclose(f);
        for (int threads : {1, 5}) {
            PackedText p{};
            CHECK(pack_fasta_file(path, threads, &p, err, sizeof err) == 0);
            CHECK(p.n == ref.n && !memcmp(p.words, ref.words, ref.nwords * 8));
            free_packed_text(&p);
        }
        z[z.size() / 2] ^= 0x5a;
        f = fopen(path, "wb"); CHECK(f && fwrite(z.data(), 1, z.size(), f) == z.size()); fclose(f);
        PackedText p{};
        CHECK(pack_fasta_file(path, 3, &p, err, sizeof err) != 0);
        remove(path);
    }
    {   // the ingest's own DEFLATE decoder and CRC (fast_inflate.h) against zlib: every kind of block (stored, fixed, dynamic,
        // codes longer than the root tables), output handed out in steps (resumed after "need output"), 16-bit symbols behind a
        // window of markers from a block boundary in the middle, stops at block boundaries, damaged and cut-short streams
        using namespace fastinflate;
        auto gen = [&](int kind, size_t len) {
            std::string t(len, 'A');
            switch (kind) {
                case 0: for (auto &c : t) c = "ACGT"[rnd() & 3]; for (size_t i = 60; i < len; i += 61) t[i] = '\n'; break;
                case 1: for (auto &c : t) c = (char)rnd(); break;
                case 2: break;
                case 3: for (size_t i = 0; i < len; i++) t[i] = (char)('a' + (rnd() % 3 ? (i % 7) : (rnd() % 26))); break;
                case 4: {
                    std::string unit(200 + rnd() % 3000, 'A');
                    for (auto &c : unit) c = "ACGT"[rnd() & 3];
                    for (size_t i = 0; i < len; i++) t[i] = rnd() % 50 ? unit[i % unit.size()] : "ACGT"[rnd() & 3];
                    break;
                }
                case 5: for (size_t i = 0; i < len; i++) t[i] = (char)(rnd() % 5 ? 'x' : rnd()); break;
            }
            return t;
        };
        auto deflate_raw = [&](const std::string &t, int level, int strategy, std::vector<size_t> *bounds) {
            z_stream zs; memset(&zs, 0, sizeof zs);
            CHECK(deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy) == Z_OK);
            std::vector<unsigned char> out(deflateBound(&zs, (uLong)t.size()) + 64);
            zs.next_in = (Bytef *)t.data(); zs.avail_in = (uInt)t.size(); zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
            CHECK(deflate(&zs, Z_FINISH) == Z_STREAM_END);
            out.resize(out.size() - zs.avail_out);
            deflateEnd(&zs);
            if (bounds) {                                            // block boundaries (bit, bytes of text before it) as zlib sees them
                z_stream is; memset(&is, 0, sizeof is);
                CHECK(inflateInit2(&is, -15) == Z_OK);
                std::vector<unsigned char> txt(t.size() + 1);
                is.next_in = out.data(); is.avail_in = (uInt)out.size(); is.next_out = txt.data(); is.avail_out = (uInt)txt.size();
                for (;;) {
                    const int r = inflate(&is, Z_BLOCK);
                    if (r != Z_OK) break;
                    if ((is.data_type & 128) && !(is.data_type & 64)) { bounds->push_back(8 * (size_t)is.total_in - (size_t)(is.data_type & 63)); bounds->push_back((size_t)is.total_out); }
                }
                inflateEnd(&is);
            }
            for (int i = 0; i < 8; i++) out.push_back(0xAB);        // (a trailer's worth of bytes behind the data, as in a gzip file)
            return out;
        };
        std::unique_ptr<Decoder> d(new Decoder());
        int cases = 0;
        for (int kind = 0; kind < 6; kind++)
            for (size_t len : {(size_t)0, (size_t)1, (size_t)100, (size_t)4097, (size_t)70000, (size_t)400001})
                for (int level : {0, 1, 6, 9})
                    for (int strategy : {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY}) {
                        const std::string t = gen(kind, len);
                        const auto z = deflate_raw(t, level, strategy, nullptr);
                        for (size_t step : {(size_t)0, (size_t)1, (size_t)333, (size_t)65536}) {
                            if (step == 1 && len > 5000) continue;
                            std::vector<unsigned char> out((step ? step : t.size()) + 1);     // (+ 1: a vector of no bytes has no address)
                            size_t pos = 0, cap = out.size() - 1;
                            int rc;
                            d->start(z.data(), z.size() - 8, 0);
                            while ((rc = d->run(out.data(), 0, &pos, cap, ~(size_t)0)) == FI_NEED_OUTPUT && cap < t.size() + 1000) { cap += step ? step : 1; out.resize(cap + 1); }
                            CHECK(rc == FI_DONE && pos == t.size() && !memcmp(out.data(), t.data(), pos) && (d->bitpos + 7) / 8 == z.size() - 8);
                            cases++;
                        }
                    }
        printf("fast_inflate: %d streams equal to zlib's text\n", cases);
        {   // from a block boundary in the middle: bytes with the true window in front, 16-bit symbols behind markers, a stop bit
            int starts = 0;
            for (int level : {1, 6}) {
                const std::string t = gen(level == 1 ? 4 : 0, 3000000);
                std::vector<size_t> bounds;
                const auto z = deflate_raw(t, level, Z_DEFAULT_STRATEGY, &bounds);
                CHECK(bounds.size() >= 8);
                for (size_t b = 2; b + 3 < bounds.size(); b += 2 * (1 + bounds.size() / 24)) {
                    const size_t bit = bounds[b], at = bounds[b + 1], stop = bounds[bounds.size() - 2], stop_at = bounds[bounds.size() - 1];
                    const size_t hist = std::min<size_t>(at, 32768);
                    std::vector<unsigned char> out(hist + (t.size() - at) + 1);
                    memcpy(out.data(), t.data() + at - hist, hist);
                    size_t pos = 0;
                    d->start(z.data(), z.size() - 8, bit);
                    int rc = d->run(out.data() + hist, hist, &pos, t.size() - at, stop);
                    CHECK(rc == FI_STOPPED && d->bitpos == stop && pos == stop_at - at && !memcmp(out.data() + hist, t.data() + at, pos));
                    rc = d->run(out.data() + hist, hist, &pos, t.size() - at, ~(size_t)0);
                    CHECK(rc == FI_DONE && pos == t.size() - at && !memcmp(out.data() + hist, t.data() + at, pos));
                    // unknown window: markers, block by block, resolved afterwards
                    std::vector<uint16_t> sym(32768 + (t.size() - at) + 1);
                    for (size_t i = 0; i < 32768; i++) sym[i] = (uint16_t)(0x8000 + i);
                    size_t n = 0, blocks = 0;
                    d->start(z.data(), z.size() - 8, bit);
                    while ((rc = d->run<uint16_t>(sym.data() + 32768, 32768, &n, t.size() - at, ~(size_t)0, true)) == FI_STOPPED) blocks++;
                    CHECK(rc == FI_DONE && n == t.size() - at && blocks >= 1);
                    bool same = true;
                    for (size_t i = 0; i < n && same; i++) {
                        const uint16_t v = sym[32768 + i];
                        const size_t w = (size_t)(v - 0x8000);       // byte w of the 32 KB that end at `at`
                        same = v < 256 ? (unsigned char)t[at + i] == v : (v >= 0x8000 && at + w >= 32768 && t[at + w - 32768] == t[at + i]);
                    }
                    CHECK(same);
                    starts++;
                }
            }
            printf("fast_inflate: %d starts in the middle (bytes with a window, 16-bit symbols behind markers, stop bits)\n", starts);
        }
        {   // damaged and cut-short streams: an error or another text, never a fault; the output bound holds
            const std::string t = gen(0, 300000);
            const auto z = deflate_raw(t, 6, Z_DEFAULT_STRATEGY, nullptr);
            int refused = 0;
            for (int k = 0; k < 1500; k++) {
                auto zz = z;
                if (k % 3 == 0) zz.resize(8 + rnd() % (zz.size() - 8)); else zz[rnd() % (zz.size() - 8)] ^= (unsigned char)(1 + rnd() % 255);
                std::vector<unsigned char> out(t.size() + 1000);
                size_t pos = 0;
                d->start(zz.data(), zz.size() - 8, 0);
                const int rc = d->run(out.data(), 0, &pos, out.size(), ~(size_t)0);
                CHECK(pos <= out.size());
                if (!(rc == FI_DONE && pos == t.size() && !memcmp(out.data(), t.data(), pos))) refused++;
            }
            printf("fast_inflate: %d of 1500 damaged streams refused or decoded to another text\n", refused);
            CHECK(refused >= 1400);
        }
        {   // differential: random texts deflated with random level, strategy, memLevel and window, three in four then damaged
            // (a bit flipped, a byte replaced, cut short, a byte inserted) -- zlib's inflate and this decoder must agree on
            // whether the stream is good, and when it is on every byte and on where it ends (150,000 such streams agreed when the
            // decoder was written; zlib is stricter than RFC 1951 about incomplete codes and the decoder follows it)
            long same = 0, refused = 0, differ = 0;
            for (long it = 0; it < 8000; it++) {
                const int kind = (int)(rnd() % 5);
                const size_t len = rnd() % 3 == 0 ? rnd() % 300 : rnd() % 40000;
                std::string t(len, 'A');
                switch (kind) {
                    case 0: for (auto &c : t) c = "ACGT"[rnd() & 3]; break;
                    case 1: for (auto &c : t) c = (char)rnd(); break;
                    case 2: break;
                    case 3: for (size_t i = 0; i < len; i++) t[i] = (char)(33 + rnd() % 41); break;
                    case 4: for (size_t i = 0; i < len; i++) t[i] = (char)('a' + (rnd() % 3 ? (i % 7) : (rnd() % 26))); break;
                }
                const int strategy = (int)(rnd() % 4);
                z_stream zs; memset(&zs, 0, sizeof zs);
                if (deflateInit2(&zs, (int)(rnd() % 10), Z_DEFLATED, -(9 + (int)(rnd() % 7)), 1 + (int)(rnd() % 9), strategy == 3 ? Z_FIXED : strategy) != Z_OK) continue;
                std::vector<unsigned char> z(deflateBound(&zs, (uLong)len) + 64);
                zs.next_in = (Bytef *)t.data(); zs.avail_in = (uInt)len; zs.next_out = z.data(); zs.avail_out = (uInt)z.size();
                deflate(&zs, Z_FINISH); z.resize(z.size() - zs.avail_out); deflateEnd(&zs);
                for (int m = it % 4 == 0 ? 0 : 1 + (int)(rnd() % 3); m > 0 && !z.empty(); m--) {
                    const int how = (int)(rnd() % 4);
                    if (how == 0) z[rnd() % z.size()] ^= (unsigned char)(1u << (rnd() % 8));
                    else if (how == 1) z[rnd() % z.size()] = (unsigned char)rnd();
                    else if (how == 2) z.resize(rnd() % (z.size() + 1));
                    else z.insert(z.begin() + (long)(rnd() % z.size()), (unsigned char)rnd());
                }
                const size_t cap = len + 70000;
                std::vector<unsigned char> o1(cap + 1), o2(cap + 1);
                z_stream is; memset(&is, 0, sizeof is);
                CHECK(inflateInit2(&is, -15) == Z_OK);
                is.next_in = z.data(); is.avail_in = (uInt)z.size(); is.next_out = o1.data(); is.avail_out = (uInt)cap;
                const int zr = inflate(&is, Z_FINISH);
                const size_t zn = is.total_out, zin = is.total_in;
                const bool zfull = zr == Z_BUF_ERROR && is.avail_out == 0;
                inflateEnd(&is);
                const size_t zlen_data = z.size();
                for (int i = 0; i < 8; i++) z.push_back(0);              // (a trailer's worth behind the data, not part of it)
                d->start(z.data(), zlen_data, 0);
                size_t pos = 0, lim = rnd() % 2 ? cap : std::min<size_t>(cap, 1 + rnd() % 500);
                int rc;
                while ((rc = d->run(o2.data(), 0, &pos, lim, ~(size_t)0)) == FI_NEED_OUTPUT && lim < cap) lim = std::min(cap, lim + 1 + rnd() % 5000);
                if (zr == Z_STREAM_END && rc == FI_DONE) { if (zn == pos && !memcmp(o1.data(), o2.data(), pos) && (d->bitpos + 7) / 8 == zin) same++; else differ++; }
                else if (zr != Z_STREAM_END && (rc != FI_DONE || zfull)) refused++;
                else differ++;
            }
            printf("fast_inflate: %ld random streams, three in four damaged: zlib and the decoder both decode %ld to the same bytes, both refuse %ld, differ on %ld\n",
                   same + refused + differ, same, refused, differ);
            CHECK(differ == 0 && same >= 2000);
        }
        {   // CRC-32 by carry-less multiplication against zlib's, any length and start value
            int bad = 0;
            for (int k = 0; k < 600; k++) {
                size_t len = rnd() % 5000;
                if (k % 50 == 0) len = 1000000 + rnd() % 1000;
                std::string t(len, 0);
                for (auto &c : t) c = (char)rnd();
                const uint32_t c0 = (uint32_t)(k % 3 ? rnd() : 0);
                if (crc32_fast(c0, (const uint8_t *)t.data(), len) != (uint32_t)crc32(c0, (const Bytef *)t.data(), (uInt)len)) bad++;
            }
            CHECK(bad == 0);
        }
    }
    {   // IUPAC letters: refused by default, replaced deterministically (independent of threads) with the option
        std::vector<std::string> r2 = recs;
        const char *iupac = "NRYKMSWBDHV";
        for (auto &s : r2) for (size_t i = 5; i < s.size(); i += 97) s[i] = iupac[rnd() % 11];
        std::string fa = fasta_of(r2, 80, false, false);
        PackedText p{}, q{};
        CHECK(pack_fasta_buffer(fa.data(), fa.size(), 2, &p, err, sizeof err) != 0);
        CHECK(pack_fasta_buffer(fa.data(), fa.size(), 1, &p, err, sizeof err, IngestOpts{INGEST_IUPAC_RANDOM, 42}) == 0);
        CHECK(pack_fasta_buffer(fa.data(), fa.size(), 7, &q, err, sizeof err, IngestOpts{INGEST_IUPAC_RANDOM, 42}) == 0);
        CHECK(p.n == q.n && !memcmp(p.words, q.words, p.nwords * 8));
        free_packed_text(&p); free_packed_text(&q);
    }
    {   // FASTQ (src/kseq.h:177-201): multi-line sequence and quality, quality lines that start with '@', '>' or '+'
        std::string fq;
        for (size_t r = 0; r < recs.size(); r++) {
            fq += "@read" + std::to_string(r) + "\n";
            for (size_t i = 0; i < recs[r].size(); i += 61) fq += recs[r].substr(i, 61) + "\n";
            fq += r % 2 ? "+read\n" : "+\n";
            std::string q(recs[r].size(), 'I');
            q[0] = "@>+"[r % 3];
            for (size_t i = 0; i < q.size(); i += 61) fq += q.substr(i, 61) + (r % 5 ? "\n" : "\r\n");
        }
        for (int threads : {1, 3, 8}) {
            PackedText p{};
            CHECK(pack_fasta_buffer(fq.data(), fq.size(), threads, &p, err, sizeof err) == 0);
            CHECK(p.n == ref.n && p.nrec == ref.nrec && !memcmp(p.words, ref.words, ref.nwords * 8));
            free_packed_text(&p);
        }
        // the rewrite by all threads (fastq_to_fasta_parallel): cuts every 64 .. 4096 bytes, guessed record starts proved by
        // the walks; the cut-short file below fails in a walk and gets the serial walk's message
        for (const char *chunk_min : {"64", "300", "4096"}) {
            setenv("DEBWT_FASTQ_CHUNK_MIN", chunk_min, 1);
            for (int threads : {2, 5, 16}) {
                PackedText p{};
                CHECK(pack_fasta_buffer(fq.data(), fq.size(), threads, &p, err, sizeof err) == 0);
                CHECK(p.n == ref.n && p.nrec == ref.nrec && !memcmp(p.words, ref.words, ref.nwords * 8));
                free_packed_text(&p);
            }
        }
        PackedText p{};
        std::string cut = fq.substr(0, fq.size() - 5);                                     // quality string cut short
        CHECK(pack_fasta_buffer(cut.data(), cut.size(), 2, &p, err, sizeof err) != 0);
        cut = fq.substr(0, fq.find('+'));                                  // no '+' line: a record without qualities (kseq)
        CHECK(pack_fasta_buffer(cut.data(), cut.size(), 2, &p, err, sizeof err) == 0 && p.nrec == 1);
        free_packed_text(&p);
        unsetenv("DEBWT_FASTQ_CHUNK_MIN");
    }
    for (const char *bad : {"ACGT\n", ">a\nACGTACGT\n", ">a\nACGTXACGTACGTACGTACGTACGTACGTACGTACGTACGT\n", "@r\nACGT\n+\nIIII\n", "@\n", "@r\nACGT", "", ">only header\n"}) {
        PackedText p{};
        CHECK(pack_fasta_buffer(bad, strlen(bad), 2, &p, err, sizeof err) != 0);
    }

    // ---- special-region module on the packed text: one thread against many, K = 11 .. 31 -------------------------------
    for (int K : {11, 15, 19, 31}) {
        uint64_t d1 = 0, d8 = 0;
        for (int pass = 0; pass < 2; pass++) {
            setenv("DEBWT_SPECIAL_PAR_MIN", pass ? "1" : "4611686018427387904", 1);
            setenv("DEBWT_SPECIAL_THREADS", pass ? "8" : "1", 1);
            SpecialTables t;
            build_special_tables(ref.words, ref.n, ref.sep, ref.nrec, K, &t);
            CHECK(t.pos.size() == ref.nrec * (uint64_t)K && t.head_keys.size() == ref.nrec);
            CHECK(t.threads_used == (pass ? 8u : 1u));
            (pass ? d8 : d1) = digest_tables(t);
        }
        CHECK(d1 == d8);
    }
    free_packed_text(&ref);

    // ---- synthetic-text generator: words and codes of a small pan-genome, chunked against whole ------------------------
    {
        uint64_t lens[3] = {40000, 35000, 25000};
        debwt_synth_spec sp{};
        sp.seed = 0x5EEDBA5Eull; sp.genome_len = 100000; sp.genomes = 3; sp.nchrom = 3; sp.chrom_len = lens;
        sp.snp_rate = 1e-3; sp.repeat_coverage = 0.25; sp.lowcx_fraction = 0.03; sp.alu_copies = 50; sp.alu_divergence = 0.12;
        debwt_synth *s = nullptr;
        CHECK(debwt_synth_open(&sp, 4, &s) == 0 && s);
        const uint64_t nw = debwt_synth_nwords(s);
        std::vector<uint64_t> whole(nw), parts(nw), sep(debwt_synth_nrec(s));
        uint64_t census[4], c2[4];
        CHECK(debwt_synth_words(s, 0, nw, 4, whole.data(), census) == 0);
        for (uint64_t a = 0; a < nw; a += 1000) CHECK(debwt_synth_words(s, a, std::min(nw, a + 1000), 2, parts.data() + a, c2) == 0);
        CHECK(whole == parts);
        CHECK(debwt_synth_sep(s, sep.data()) == 0 && sep.back() == debwt_synth_n(s) - 1);
        std::vector<uint8_t> codes(100000);
        CHECK(debwt_synth_codes(s, 1, 0, 100000, codes.data()) == 0);
        debwt_synth_close(s);
    }
    printf("host_sanitize: %d failures\n", failures);
    return failures ? 1 : 0;
}
