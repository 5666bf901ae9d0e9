// radix_sort.h -- LSD radix sort of 64-bit keys resident in HBM (the mySort replacement,
// /root/reference/src/mySort.c:98-176: 4^12-bin MSD bucketing under 16.7 M rwlocks + per-bucket
// qsort there; lock-free 8-bit LSD passes with LDS-staged histograms and wave-level ranking here).
#pragma once
#include "common.h"

#define RS_BLOCK 256
#ifndef RS_ITEMS
#define RS_ITEMS 16
#endif
#define RS_TILE (RS_BLOCK * RS_ITEMS)   // keys ranked per workgroup iteration
#ifndef RS_MAXCHUNKS
#define RS_MAXCHUNKS 16384              // most workgroups (chunks) of a pass; see rs_plan
#endif
#define RS_MINCHUNKS 2048               // 8 per CU
#ifndef RS_CHUNK_TILES
#define RS_CHUNK_TILES 64               // tiles a chunk should hold when there are more than RS_MINCHUNKS chunks
#endif
#define RS_RADIX 256

struct RadixWorkspace {
    u32 *counts;        // [RS_RADIX][RS_MAXCHUNKS + 1] digit-major chunk histograms / offsets + digit totals
    u32 *over;          // hybrid path: [0] = number of oversize tiles, then over_cap (start,len) u64 pairs at +16 B
    u32 *h_over;        // pinned host word for the count
    u32 over_cap;       // entries the list holds (radix_over_bytes sizes it for one entry per 4096-key tile)
    u32 *skew_list;     // hybrid path: one byte per 4096-key tile, set when a 1024-key wave tile did not fit
};

// Optional key source for the FIRST pass: node keys (node << 2 | pred) computed on the fly from the 2-bit text,
// one per position whose K-window holds no separator -- the key array is then never written out unsorted.
struct TextKeySrc {
    const u64 *text;      // packed text, reference layout
    const u64 *sepbits;   // separator bitmap, bit i of word i>>6
    u64 n;                // positions
    int K;                // node length
    u64 key_lo, key_hi;   // only keys in [key_lo, key_hi) are produced (a k-mer-prefix shard); hi = 0: no upper bound
    u64 pos0;             // item i of a pass is text position pos0 + i
    const u32 *pre_counts;  // optional: the chunk histograms of the first pass, digit-major as rs_hist writes them
                            // (radix_text_hist_ranges computed them for all key ranges in one scan of the text)
    const u8 *bin_tab;      // optional: 4096 bytes, a key whose prefix bin (key >> bin_shift) maps to 0xFF is not
    int bin_shift;          // produced (exchange rounds: only the keys of this round's ranges leave the slice)
};

// One scan of the text for the first-pass chunk histograms of up to RS_MAX_RANGES key ranges at once: range r holds
// the keys whose top 12 bits map to r in range_of_bin (4096 bytes, device), its first pass buckets by the digit at
// shift[r].  counts: RS_MAX_RANGES-or-fewer blocks of radix_text_hist_stride() words, range r's histograms in block r.
#define RS_MAX_RANGES 16
size_t radix_text_hist_stride();
// bit position of the digit the first pass of radix_sort_u64(n keys, key_bits, algo) buckets by
int radix_first_shift(u64 n, int key_bits, int algo);
hipError_t radix_text_hist_ranges(hipStream_t stream, const TextKeySrc &text, const u8 *range_of_bin, int key_bits,
                                  const int *shift, int nranges, u32 *counts);

// bucket function of a pass (see rs_digit)
struct RsDigit {
    int shift; u32 mask;          // mode 0
    int mode;
    const u8 *tab; int tshift;    // mode 1: tab[key >> tshift]
    const u32 *bounds; u32 nb;    // mode 2: owner of block id key >> tshift
    int out_strip;                // rs_scatter_kernel<0,2,0> (the last pass of the blue-entry sort): a routed entry
                                  // (block id << out_strip | SP index << 3 | pred) leaves as a blue entry (pred | SP index << 4)
};

size_t radix_workspace_bytes(u64 max_keys);
size_t radix_over_bytes(u64 max_keys);
// sparse: only a fraction of the text positions yields a key (bin_tab filter): collect the keys of several position
// tiles before ranking (rs_scatter_sparse_kernel)
hipError_t radix_partition_by_shard(hipStream_t stream, const u64 *src, const TextKeySrc *text, u64 count, u64 *dst,
                                    const RsDigit &dg, u32 nshards, const RadixWorkspace &ws, u64 *offs_host,
                                    bool sparse, u64 capacity);

// Sorts `n` keys ascending on their low `key_bits` bits.  a: input; b: scratch of n words.
// Returns the buffer (a or b) that holds the result.  All work is enqueued on `stream`.
// algo: 1 = LSD: histogram + scan + scatter per 8-bit digit, all digits in HBM;
//       3 = hybrid (default): the top digits by the same LSD passes until buckets are a few keys long
//           ("k-mer prefix bucketing"), then one kernel that finishes every bucket inside LDS.
// pass_events (optional): max_pairs pairs of hipEvents; pair i is recorded on `stream` right before and
// after the scatter kernel of pass i (no synchronisation); *npairs receives the number recorded.
// text (optional): when given, the keys are taken from the text in the first pass (`a` is scratch, n = number of
// valid positions); pass events are then recorded for the array-to-array passes only.
// Stable LSD passes over the key bits [lo_bit, hi_bit) only (auxiliary kernels); a: input, b: scratch of n words;
// returns the buffer that holds the result.
// strip_last > 0: when the result comes to lie in `a` (an even number of passes, or an odd number >= 3 with a second scratch
// buffer `third` of n words to rotate through), the last pass writes blue entries instead of routed ones
// (RsDigit::out_strip = strip_last) and *stripped is set; otherwise the caller strips them itself.
u64 *radix_sort_bits(hipStream_t stream, u64 *a, u64 *b, u64 n, int lo_bit, int hi_bit, const RadixWorkspace &ws,
                     hipError_t *err, int strip_last = 0, bool *stripped = nullptr, u64 *third = nullptr);
// The same with the result in `dst`, another buffer than the input `a` (scratch afterwards), stripped by the last pass when
// strip_last > 0; an even number of passes takes its first hop through `third` (n words).  Returns false, with nothing
// launched, when that buffer is missing or the sort is trivial (n < 2, no bits).
bool radix_sort_bits_into(hipStream_t stream, u64 *a, u64 *dst, u64 *third, u64 n, int lo_bit, int hi_bit, const RadixWorkspace &ws,
                          hipError_t *err, int strip_last = 0);
// sink (optional): when the hybrid path runs, the bucket finish also counts the distinct keys of every tile it has
// in registers and writes the row symbols key & 3, and the run-length encoding of the sorted keys -- distinct keys,
// first row of each (row = index in sorted order) -- follows tile by tile without a counting pass over the keys.
// `done` tells the caller whether that happened; if not, it encodes the returned keys itself.
struct RleSink {
    u64 *dk; u32 *dstart; u8 *mchar;   // out: distinct keys, their first rows, one symbol per row
    void *ws;                          // radix_rle_ws_bytes(n) bytes of device scratch
    u32 *h_total;                      // pinned host word: number of distinct keys, valid once the stream drained
    u32 *h_ctr;                        // optional, 4 pinned host words: [0] stretches above a wave tile, [3] those of them
                                       // left to the 4096-key network (valid once the stream drained)
    u32 n_over;                        // out: stretches above 4096 keys, finished by all-HBM passes
    bool no_staging;                   // in: tiles do not stage their distinct keys in the other key buffer (A/B, tests)
    bool drop_sorted;                  // in: the caller reads the encoding only -- a tile that staged its distinct keys need not
                                       // write its sorted keys back (the returned array is then sorted in the other tiles only)
    bool done;
};
size_t radix_rle_ws_bytes(u64 n);
u64 *radix_sort_u64(hipStream_t stream, u64 *a, u64 *b, u64 n, int key_bits, const RadixWorkspace &ws,
                    int algo, hipEvent_t *pass_events, int max_pairs, int *npairs, hipError_t *err,
                    const TextKeySrc *text = nullptr, RleSink *sink = nullptr);
