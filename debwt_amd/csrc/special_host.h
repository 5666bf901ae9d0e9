// special_host.h -- host-side special-region module (the reference's `collect` tables,
// /root/reference/src/collect#$.c:118-157,253-311,348-602): the K suffixes per record that start
// at most K-1 symbols before a separator follow their own rules and are few (nrec * K), so they are
// handled on the host while the GPU sorts the n node instances.
#pragma once
#include <stdint.h>
#include <vector>

struct SpecialTables {
    // special suffixes in true suffix order (A<C<G<T<#<$, '#' equal, comparison continues)
    std::vector<uint64_t> pos;      // text position
    std::vector<uint64_t> key;      // K symbols, separator and everything after it replaced by 'T'
    std::vector<uint8_t> chr;       // BWT symbol (the base before the suffix)
    // positions of special suffixes that are multi-out (ascending)
    std::vector<uint64_t> branch;
    // per record: key (node<<2|3) of the node instance at the record start (ascending), and
    // (node<<2|1) of the node instance immediately followed by the separator
    std::vector<uint64_t> head_keys;
    std::vector<uint64_t> tail_facts;
    unsigned threads_used = 1;      // host threads the module ran on
};

// words: reference-format packed text (>= ceil((n+32)/32)+1 words readable); sep: separator
// positions ascending, sep[nrec-1] == n-1.  K = k-1.
void build_special_tables(const uint64_t *words, uint64_t n, const uint64_t *sep, uint64_t nrec, int K,
                          SpecialTables *out);

// Inside every tie group of record starts -- act[0..na): ascending places of `ord`, a group = the run of places with
// equal gid -- sort the records ord[place] by the true suffix order of their first positions.  The device module
// (special_kernels.h) hands over the few groups that are still tied after its last window round (records that are
// identical for thousands of symbols).
void special_order_record_starts(const uint64_t *words, uint64_t n, const uint64_t *sep, uint64_t nrec, uint32_t *ord,
                                 const uint32_t *gid, const uint32_t *act, uint64_t na);
