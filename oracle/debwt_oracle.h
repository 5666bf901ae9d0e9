/*
 * debwt_oracle.h -- CPU restatement of the deBWT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / reported CPU baseline.  The product path
 * (debwt_amd/csrc -> libdebwt_hip.so) never links or calls it.
 *
 * Pinning: the restatement is checked (tests/test_oracle_pinned.py, run in the build
 * container where /root/reference exists) against the reference's own stage functions
 * compiled from /root/reference/src by oracle/Makefile into oracle/_ref/ref_driver,
 * and against the golden vectors under tests/golden/ that were produced by that driver
 * (tests/golden/make_golden.py).  The k-mer counting step (a-1) is Jellyfish 2.x in the
 * reference (src/kmercounting.sh:8,11; un-vendored, absent from this image): its
 * arithmetic -- exact, non-canonical counts of every k-mer inside each record -- is
 * restated in orc_kmer_count() and pinned by definition: tests/refformat.naive_kmer_counts
 * (np.unique over the k-windows of every record) must equal it, the dump ref_driver feeds to
 * the reference's mySort, and the reference's kmerInfo (tests/test_oracle.py, make_golden.py).
 * Intermediates (red table, SP code) are pinned against the files the reference's
 * generateBlocks / generateSP wrote (sha256 in tests/golden/manifest.json).
 *
 * Text model (reference src/main.c:18-23, src/collect#$.c:56-90): T = r0 # r1 # ... $,
 * alphabet A<C<G<T<#<$, all '#' compare equal and comparison continues past them.
 * Symbol codes: A0 C1 G2 T3 #4 $5.
 */
#ifndef DEBWT_ORACLE_H
#define DEBWT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint64_t n;                  /* BWTLEN  (src/collect#$.c:56-57) */
    uint64_t nrec;               /* countRead */
    uint64_t distinct_kmers;     /* D: lines of the Jellyfish dump */
    uint64_t red_capacity;       /* redCapacity   (src/INandOut.c:396-417) */
    uint64_t blue_capacity;      /* blueCapacity  (src/INandOut.c:904) */
    uint64_t blue_bound_num;     /* blueBoundNum  (src/INandOut.c:361-366) */
    uint64_t case3num;           /* case3num = 2 * blocks (src/INandOut.c:349-358) */
    uint64_t sp_len;             /* S: SP-code symbols (src/generateSP.c:626-660) */
    uint64_t special_branch_num; /* specialBranchNum (src/collect#$.c:534-598) */
    uint64_t cmp_calls;          /* SP comparator calls in the blue sort */
} orc_stats;

/* Build the symbol array r0#r1#...$ from concatenated ASCII records (upper or lower case
 * ACGT only; src/main.c:18-23).  sym must hold sum(reclen)+nrec bytes.  Returns n, or 0 on
 * invalid input (a record of <= 32 bases, src/collect#$.c:41-45, or a non-ACGT letter). */
uint64_t orc_make_text(const char *seq, const uint64_t *reclen, uint64_t nrec, uint8_t *sym);

/* 2-bit packed text exactly as the reference keeps it (src/collect#$.c:61-90): 32 symbols per
 * word MSB first, 'T' at every separator, 32 'T' of padding; words = ceil((n+32)/32). */
void orc_pack_text(const uint8_t *sym, uint64_t n, uint64_t *words);

/* a-1 + a-2: exact count of every k-mer inside each record (Jellyfish dump), sorted ascending
 * as mySort writes kmerInfo (src/mySort.c:193-195): kmer left-aligned, 2 bits/base.
 * kmers/counts must hold n entries (upper bound).  Returns D. */
uint64_t orc_kmer_count(const uint8_t *sym, uint64_t n, int k, uint64_t *kmers, uint64_t *counts);

/* Whole path a-2..a-9: BWT in the on-disk layout of src/insertCase3.c:115-131.
 * bwt_words: ceil(n/32) words; hash_rows: nrec-1 ascending rows; dollar_row: 1 row.
 * threads >= 1 parallelises the sorts (result independent of it).  Returns 0 on success. */
int orc_build_bwt(const uint8_t *sym, uint64_t n, int k, int threads,
                  uint64_t *bwt_words, uint64_t *hash_rows, uint64_t *dollar_row, orc_stats *st);

/* Intermediates of the same run, reference-compatible: caller passes NULL to skip.
 * sp_sym: S bytes (0..5); red_nodes: red_capacity entries = node<<2 | multiin<<1 | multiout,
 * node right-aligned 2(k-1) bits, ascending. */
int orc_build_bwt_ex(const uint8_t *sym, uint64_t n, int k, int threads,
                     uint64_t *bwt_words, uint64_t *hash_rows, uint64_t *dollar_row, orc_stats *st,
                     uint8_t *sp_sym, uint64_t *red_nodes);

/* Definition check: naive suffix sort of T under A<C<G<T<#<$ ('#' equal, continue).  out[n]
 * receives the BWT symbols 0..5.  O(n log n * lcp): small inputs only. */
void orc_naive_bwt(const uint8_t *sym, uint64_t n, uint8_t *out);

/* Expand the on-disk layout back to 6-symbol rows. */
void orc_unpack_bwt(const uint64_t *bwt_words, uint64_t n, const uint64_t *hash_rows, uint64_t nrec,
                    uint64_t dollar_row, uint8_t *out);

/* Inverse BWT by LF walk from the '$' row (= last row), the check the reference's dead
 * LFsearch path was written for (src/LFsearch.c:49-166).  Writes n symbols.  Returns 0 if the
 * walk visited every row exactly once. */
int orc_inverse_bwt(const uint64_t *bwt_words, uint64_t n, const uint64_t *hash_rows, uint64_t nrec,
                    uint64_t dollar_row, uint8_t *sym_out);

#ifdef __cplusplus
}
#endif
#endif
