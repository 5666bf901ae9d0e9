// dump_host.cpp -- SURVEY 8f-4: the arrays the stages hand to each other, written in the byte formats of the files (and
// global arrays) the reference's stages leave behind, so that a parity failure can be bisected stage by stage against
// the reference compiled in the build container (`oracle/_ref/ref_driver` writes the same names).  Host code over the
// public C ABI only (debwt_fetch_array, debwt_kmer_count_sorted, debwt_get_stats): no kernel here.
//
//   kmerInfo        D x {u64 k-mer left-aligned, u64 count}, ascending              src/mySort.c:193-195
//   redSeq          R x u64: (low k-1-10 symbols of the node) << 2 | multiin << 1 | multiout   src/INandOut.c:396-404
//   redPoint        R x u64: inclusive end of the last multi-in block at or before the entry   src/INandOut.c:405
//   blueBound       Q x u64: inclusive end of every block                            src/INandOut.c:359-361
//   case3bound      2Q x u64: [first row, last row] of every block                   src/INandOut.c:347-353
//   spCode          ceil(S / 32) x u64: 2 bits per SP symbol, symbol j at bit 2 * (31 - (j & 31)) of word j >> 5,
//                   separators stored as 3                                            src/generateSP.c:626-660
//   spSpecialIndex  N x u64: SP positions of the separators, ascending, the last is '$'        src/generateSP.c:630-641
//   blueTable       B x u64: pred (0..5) | spIndex << 4, block by block, before sortBlue (the order inside a block is
//                   the order the text scan reached the entries: only the set per block is defined)  src/generateSP.c:666-672
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/debwt_hip.h"

namespace {

int write_all(const std::string &dir, const char *name, const void *p, size_t bytes) {
    const std::string path = dir + "/" + name;
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return DEBWT_EIO;
    const size_t w = bytes ? fwrite(p, 1, bytes, f) : 0;
    return (fclose(f) == 0 && w == bytes) ? DEBWT_OK : DEBWT_EIO;
}

int fetch_u64(debwt_ctx *c, debwt_array which, std::vector<uint64_t> &out) {
    uint64_t cnt = 0;
    int rc = debwt_fetch_array(c, which, nullptr, 0, &cnt);
    if (rc) return rc;
    out.assign(cnt ? cnt : 1, 0);
    if (cnt && (rc = debwt_fetch_array(c, which, out.data(), cnt, &cnt))) return rc;
    out.resize(cnt);
    return DEBWT_OK;
}

}  // namespace

extern "C" int debwt_dump_reference_files(debwt_ctx *c, const char *dir, int stage) {
    if (!c || !dir) return DEBWT_EINVAL;
    const std::string d(dir);
    debwt_stats st;
    int rc = debwt_get_stats(c, &st);
    if (rc) return rc;
    if (stage == DEBWT_DUMP_KMERINFO) {
        uint64_t D = 0;
        if ((rc = debwt_kmer_count_sorted(c, nullptr, nullptr, 0, &D))) return rc;
        std::vector<uint64_t> km(D ? D : 1), ct(D ? D : 1), pairs(2 * D);
        if ((rc = debwt_kmer_count_sorted(c, km.data(), ct.data(), D, &D))) return rc;
        for (uint64_t i = 0; i < D; i++) { pairs[2 * i] = km[i]; pairs[2 * i + 1] = ct[i]; }
        return write_all(d, "kmerInfo", pairs.data(), pairs.size() * 8);
    }
    if (stage == DEBWT_DUMP_BLOCKS) {
        std::vector<uint64_t> red, bb, c3;
        if ((rc = fetch_u64(c, DEBWT_ARR_RED, red))) return rc;
        if ((rc = fetch_u64(c, DEBWT_ARR_BLUE_BOUND, bb))) return rc;
        if ((rc = fetch_u64(c, DEBWT_ARR_CASE3_BOUND, c3))) return rc;
        debwt_config cfg;
        if ((rc = debwt_get_config(c, &cfg))) return rc;
        const int k = cfg.k;
        const uint64_t extract = (k - 1 - 10) >= 32 ? ~0ull : ((1ull << (2 * (k - 1 - 10))) - 1);
        std::vector<uint64_t> seq(red.size()), point(red.size());
        uint64_t q = 0;                                        // multi-in entries seen so far
        for (size_t r = 0; r < red.size(); r++) {
            seq[r] = (((red[r] >> 2) & extract) << 2) | (red[r] & 3);
            if (red[r] & 2) q++;
            point[r] = q ? (q - 1 < bb.size() ? bb[q - 1] : ~0ull) : ~0ull;   // blueBound - 1 with blueBound still 0: 2^64 - 1
        }
        if ((rc = write_all(d, "redSeq", seq.data(), seq.size() * 8))) return rc;
        if ((rc = write_all(d, "redPoint", point.data(), point.size() * 8))) return rc;
        if ((rc = write_all(d, "blueBound", bb.data(), bb.size() * 8))) return rc;
        return write_all(d, "case3bound", c3.data(), c3.size() * 8);
    }
    if (stage == DEBWT_DUMP_SP) {
        uint64_t S = 0;
        if ((rc = debwt_fetch_array(c, DEBWT_ARR_SP_SYMBOLS, nullptr, 0, &S))) return rc;
        std::vector<uint8_t> sym(S ? S : 1);
        if (S && (rc = debwt_fetch_array(c, DEBWT_ARR_SP_SYMBOLS, sym.data(), S, &S))) return rc;
        std::vector<uint64_t> code((S + 31) / 32, 0), special;
        for (uint64_t j = 0; j < S; j++) {
            const uint64_t s = sym[j] >= 4 ? 3 : sym[j];
            code[j >> 5] |= s << (2 * (31 - (j & 31)));
            if (sym[j] >= 4) special.push_back(j);
        }
        std::vector<uint64_t> blue;
        if ((rc = fetch_u64(c, DEBWT_ARR_BLUE, blue))) return rc;
        if ((rc = write_all(d, "spCode", code.data(), code.size() * 8))) return rc;
        if ((rc = write_all(d, "spSpecialIndex", special.data(), special.size() * 8))) return rc;
        return write_all(d, "blueTable", blue.data(), blue.size() * 8);
    }
    return DEBWT_EINVAL;
}
