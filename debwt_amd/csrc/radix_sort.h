// radix_sort.h -- LSD radix sort of 64-bit keys resident in HBM (the mySort replacement,
// /root/reference/src/mySort.c:98-176: 4^12-bin MSD bucketing under 16.7 M rwlocks + per-bucket
// qsort there; lock-free 8-bit LSD passes with LDS-staged histograms and wave-level ranking here).
#pragma once
#include "common.h"

#define RS_BLOCK 256
#define RS_ITEMS 16
#define RS_TILE (RS_BLOCK * RS_ITEMS)   // keys ranked per workgroup iteration
#define RS_MAXCHUNKS 2048               // workgroups per pass (8 per CU)
#define RS_RADIX 256

struct RadixWorkspace {
    u32 *counts;        // [RS_RADIX][RS_MAXCHUNKS] digit-major chunk histograms / offsets
    u32 *lookback;      // single-sweep path: tile status words
    u64 lookback_words; // capacity of `lookback` in u32
    u32 *tile_counter;  // single-sweep path: dynamic tile ids, one per pass
};

size_t radix_workspace_bytes(u64 max_keys);

// Sorts `n` keys ascending on their low `key_bits` bits.  a: input; b: scratch of n words.
// Returns the buffer (a or b) that holds the result.  All work is enqueued on `stream`.
// algo: 1 = histogram + scan + scatter per pass; 2 = single-sweep passes with decoupled look-back.
// pass_events (optional): max_pairs pairs of hipEvents; pair i is recorded on `stream` right before and
// after the scatter kernel of pass i (no synchronisation); *npairs receives the number recorded.
u64 *radix_sort_u64(hipStream_t stream, u64 *a, u64 *b, u64 n, int key_bits, const RadixWorkspace &ws,
                    int algo, hipEvent_t *pass_events, int max_pairs, int *npairs, hipError_t *err);
