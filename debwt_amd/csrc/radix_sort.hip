// radix_sort.hip -- 8-bit LSD radix sort of 64-bit keys for gfx950 (wave64).
//
// Replaces the reference's edge sort (src/mySort.c:98-176, 203-238, 371-401).  Per pass:
//   rs_hist (LDS histogram per chunk of consecutive tiles) -> rs_scan_* (digit-major exclusive scan) -> rs_scatter
// rs_scatter ranks a tile of RS_TILE keys per iteration: coalesced loads (or keys rolled off the staged 2-bit text in
// the first pass), per-wave digit matching through LDS peer masks, per-wave counters, a 256-wide digit scan, staging
// of the tile in LDS in digit order, and stores of whole 128-byte lines out of per-digit carries.
// The hybrid sort (algo 3, default) runs only the top digits that way and finishes every prefix bucket in registers
// (rs_local_kernel); algo 1 runs all digits by LSD passes.
#include "radix_sort.h"

#include <algorithm>
#include <cstddef>
#include <utility>
#include <cstdio>
#include <cstdlib>
#include <vector>

// ---------------------------------------------------------------------------------------------------
// key sources: SRC 0 = key array, SRC 1 = node keys straight from the packed text (first pass only)

// what a pass buckets by: mode 0 = an 8-bit digit of the key; mode 1 = the shard that owns the key's 12-bit
// prefix bin (k-mer bucket exchange); mode 2 = the shard that owns the block id carried in bits 36.. of a blue
// entry (bounds[i] = first block of shard i)
// FIXED0: the pass is known at compile time to bucket by a key digit (the key-sort kernels); 2: and the digit lies
// in the upper key word (shift >= 32: every pass of the hybrid sort for k >= 28) -- one bit-field extract
template <int FIXED0>
__device__ __forceinline__ u32 rs_digit(const RsDigit &g, u64 k) {
    if (FIXED0 == 2) return __builtin_amdgcn_ubfe((u32)(k >> 32), (u32)g.shift - 32u, (u32)__popc(g.mask));
    if (FIXED0 || g.mode == 0) {
        // 32-bit funnel shift instead of a 64-bit one (a digit never needs more than 32 bits of the key);
        // branch-free in the (uniform) shift
        const u32 lo = (u32)k, hi = (u32)(k >> 32);
        const bool up = g.shift >= 32;
        return __builtin_amdgcn_alignbit(up ? 0u : hi, up ? hi : lo, (u32)g.shift & 31u) & g.mask;
    }
    if (g.mode == 1) return g.tab[(k >> g.tshift) & 4095u];   // masked: lanes without a key carry ~0
    u32 q = (u32)(k >> g.tshift), lo = 0, hi = g.nb;
    if (hi <= 16u) {
        // few owners (the key ranges of one GPU, the shards of one node): their bounds come by uniform loads and the owner is
        // a count of compares -- a bisection asks memory three or four times per entry, each time with an address that
        // depends on the answer before (and the digit of an entry is computed three times a pass: histogram, ranking, flush)
        for (u32 i = 1; i < hi; i++) lo += g.bounds[i] <= q ? 1u : 0u;
        return lo;
    }
    while (lo + 1 < hi) { u32 mid = (lo + hi) >> 1; if (g.bounds[mid] <= q) lo = mid; else hi = mid; }
    return lo;
}

template <int SRC>
__device__ __forceinline__ bool rs_load_key(const u64 *__restrict__ in, const TextKeySrc &ts, u64 idx, u64 end,
                                            u64 *key) {
    if (SRC == 0) {
        if (idx >= end) { *key = ~0ull; return false; }
        *key = in[idx];
        return true;
    } else {
        *key = ~0ull;
        if (idx >= end) return false;
        idx += ts.pos0;                                                   // slice of the text (shard exchange)
        u64 sw = sep_window(ts.sepbits, idx);
        if (sw & ((1ull << ts.K) - 1ull)) return false;                   // window holds a separator: no node
        u64 node = text_window(ts.text, idx) >> (64 - 2 * ts.K);
        u32 pred = idx ? text_symbol(ts.text, idx - 1) : 3u;              // 'T' stands at separators
        u64 k = (node << 2) | pred;
        if (k < ts.key_lo || (ts.key_hi && k >= ts.key_hi)) return false;  // not this shard's prefix range
        if (ts.bin_tab && ts.bin_tab[(k >> ts.bin_shift) & 4095u] == 0xFFu) return false;
        *key = k;
        return true;
    }
}

// Text pass: the tile's text and separator words are staged in LDS once (RS_TILE positions = RS_TILE/32 text words
// and RS_TILE/64 bitmap words, plus the word before for the predecessor symbol and the words behind for the
// windows), so a key costs LDS reads instead of five same-address global loads per lane.
#define RH_ITEMS (RS_ITEMS < 16 ? RS_ITEMS : 16)          // positions a lane of the histogram kernels takes per window
#define RS_STEXT (RS_TILE / 32 + 3)
#define RS_SSEP (RS_TILE / 64 + 2)
struct TextStage {
    u64 *stext, *ssep;    // LDS
    u64 tpos, spos;       // text position of stext[0] bit 63.. / of ssep[0] bit 0
};
// stages the words for positions [p0, p0 + RS_TILE); ends with a barrier (and starts with one: the previous
// tile's readers are done)
__device__ __forceinline__ void rs_stage_text(const TextKeySrc &ts, u64 p0, TextStage &st) {
    const u64 wfirst = p0 ? (p0 - 1) >> 5 : 0, sfirst = p0 >> 6;
    const u64 wlim = ((ts.n + 63) >> 5) + 2, slim = (ts.n >> 6) + 3;      // words the two buffers hold
    lds_barrier();
    for (u32 j = threadIdx.x; j < RS_STEXT; j += blockDim.x) st.stext[j] = wfirst + j < wlim ? ts.text[wfirst + j] : 0ull;
    for (u32 j = threadIdx.x; j < RS_SSEP; j += blockDim.x) st.ssep[j] = sfirst + j < slim ? ts.sepbits[sfirst + j] : 0ull;
    st.tpos = wfirst << 5; st.spos = sfirst << 6;
    lds_barrier();
}
// N consecutive tile items from idx0 on: the 64 symbols that start one symbol before the first item and the
// separator bits are fetched once, the window rolls (every shift is a compile-time constant), so a key costs a
// few ALU operations instead of five LDS reads.  Returns the mask of items that are keys.
template <int N>
__device__ __forceinline__ u32 rs_staged_keys(const TextKeySrc &ts, const TextStage &st, u64 idx0, u64 end, u64 (&key)[N]) {
    static_assert(N <= 16, "the rolled window holds 64 symbols");
    u32 vmask = 0;
#pragma unroll
    for (int r = 0; r < N; r++) key[r] = ~0ull;
    if (idx0 >= end) return 0;
    const u64 p0 = ts.pos0 + idx0;
    const u64 rel = p0 ? p0 - 1 - st.tpos : 0;
    const u32 w = (u32)(rel >> 5), sh = (u32)(rel & 31) << 1;
    const u64 a0 = st.stext[w], a1 = st.stext[w + 1], a2 = st.stext[w + 2];
    u64 A = sh ? (a0 << sh) | (a1 >> (64 - sh)) : a0;            // symbols p0-1 .. p0+30
    u64 B = sh ? (a1 << sh) | (a2 >> (64 - sh)) : a1;            // symbols p0+31 .. p0+62
    if (p0 == 0) {                                               // the text's first position: the 'T' that stands
        B = (A << 62) | (B >> 2);                                // at separators goes before it (fake pred 3)
        A = (3ull << 62) | (A >> 2);
    }
    const u64 S = sep_window(st.ssep, p0 - st.spos);             // bit r: separator at p0 + r
    const u64 kmask = (1ull << ts.K) - 1ull;
    const int nsh = 64 - 2 * ts.K;
#pragma unroll
    for (int r = 0; r < N; r++) {
        const int j = r + 1;                                      // item r starts at symbol j of (A, B)
        if (idx0 + r >= end || ((S >> r) & kmask)) continue;      // behind the chunk / window holds a separator
        const u64 win = (A << (2 * j)) | (B >> (64 - 2 * j));
        const u64 k = ((win >> nsh) << 2) | ((A >> (64 - 2 * j)) & 3ull);
        if (k < ts.key_lo || (ts.key_hi && k >= ts.key_hi)) continue;   // not this shard's prefix range
        if (ts.bin_tab && ts.bin_tab[(k >> ts.bin_shift) & 4095u] == 0xFFu) continue;   // not this exchange round's
        key[r] = k;
        vmask |= 1u << r;
    }
    return vmask;
}

// ---------------------------------------------------------------------------------------------------
// algo 1

#ifndef RS_HIST_UNIFORM
#define RS_HIST_UNIFORM 1
#endif
// AUX = 1 instantiates the same kernels under a second name for the small auxiliary sorts (fact lists, '#' rows),
// so that profiler averages of the key-sort passes are not diluted by them
template <int SRC, int AUX>
__global__ __launch_bounds__(RS_BLOCK) void rs_hist_kernel(const u64 *__restrict__ keys, TextKeySrc ts, u64 n,
                                                            u64 chunk, RsDigit dg, u32 *__restrict__ counts,
                                                            u32 nchunks) {
    __shared__ u32 h[RS_RADIX];
    h[threadIdx.x] = 0;
    __syncthreads();
    u64 beg = (u64)blockIdx.x * chunk;
    u64 end = beg + chunk < n ? beg + chunk : n;
    if (SRC == 0) {
        // chunk is a multiple of RS_TILE, so beg is 16-byte aligned: two keys per lane per load
        for (u64 i = beg + 2ull * threadIdx.x; i < end; i += 2ull * RS_BLOCK) {
            if (i + 1 < end) {
                ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(keys + i);
                const u32 d0 = rs_digit<!AUX>(dg, v.x), d1 = rs_digit<!AUX>(dg, v.y);
#if RS_HIST_UNIFORM
                // The wave's 128 keys carry one digit (the passes over the oversize stretches of a real genome, whose keys
                // share their upper digits by the ten thousand): 128 adds to one counter are executed one after the other
                // -- those passes counted at 2.7 TB/s where the key ranges count at 5.8 -- so one lane adds for all.
                const u32 f = (u32)__builtin_amdgcn_readfirstlane((int)d0);
                if (AUX && __ballot(d0 != f || d1 != f) == 0ull) {        // (the key ranges' own passes lose 10 % to the test)
                    const u64 act = __ballot(1);
                    if ((threadIdx.x & 63u) == (u32)__ffsll((long long)act) - 1u) atomicAdd(&h[f], 2u * (u32)__popcll(act));
                    continue;
                }
#endif
                atomicAdd(&h[d0], 1u);
                atomicAdd(&h[d1], 1u);
            } else {
                atomicAdd(&h[rs_digit<!AUX>(dg, keys[i])], 1u);
            }
        }
    } else {
        __shared__ u64 stext[RS_STEXT], ssep[RS_SSEP];
        TextStage st{stext, ssep, 0, 0};
        for (u64 tile = beg; tile < end; tile += RS_TILE) {
            rs_stage_text(ts, ts.pos0 + tile, st);
            for (u32 part = 0; part < RS_ITEMS / RH_ITEMS; part++) {              // (a lane rolls at most 16 positions off one window)
                u64 k[RH_ITEMS];
                u32 vm = rs_staged_keys<RH_ITEMS>(ts, st, tile + (u64)part * (RS_BLOCK * RH_ITEMS) + (u64)threadIdx.x * RH_ITEMS, end, k);
#pragma unroll
                for (u32 r = 0; r < RH_ITEMS; r++)
                    if ((vm >> r) & 1u) atomicAdd(&h[rs_digit<!AUX>(dg, k[r])], 1u);
            }
        }
    }
    __syncthreads();
    counts[(u64)threadIdx.x * nchunks + blockIdx.x] = h[threadIdx.x];
}

// first-pass histograms of several key ranges in one scan of the text (see radix_sort.h)
struct RangeShifts { int s[RS_MAX_RANGES]; };
__global__ __launch_bounds__(RS_BLOCK) void rs_hist_ranges_kernel(TextKeySrc ts, u64 chunk, const u8 *__restrict__ range_of_bin,
                                                                   int bin_shift, RangeShifts sh, int nranges,
                                                                   u32 *__restrict__ counts, u64 stride, u32 nchunks) {
    __shared__ u32 h[RS_MAX_RANGES][RS_RADIX];
    __shared__ u8 rob[4096];
    __shared__ u64 stext[RS_STEXT], ssep[RS_SSEP];
    for (u32 i = threadIdx.x; i < RS_MAX_RANGES * RS_RADIX; i += RS_BLOCK) (&h[0][0])[i] = 0;
    for (u32 i = threadIdx.x; i < 4096; i += RS_BLOCK) rob[i] = range_of_bin[i];
    __syncthreads();
    u64 beg = (u64)blockIdx.x * chunk;
    u64 end = beg + chunk < ts.n ? beg + chunk : ts.n;
    TextStage st{stext, ssep, 0, 0};
    for (u64 tile = beg; tile < end; tile += RS_TILE) {
        rs_stage_text(ts, ts.pos0 + tile, st);
        for (u32 part = 0; part < RS_ITEMS / RH_ITEMS; part++) {
            u64 k[RH_ITEMS];
            u32 vm = rs_staged_keys<RH_ITEMS>(ts, st, tile + (u64)part * (RS_BLOCK * RH_ITEMS) + (u64)threadIdx.x * RH_ITEMS, end, k);
#pragma unroll
            for (u32 r = 0; r < RH_ITEMS; r++)
                if ((vm >> r) & 1u) {
                    const u32 g = rob[(u32)(k[r] >> bin_shift) & 4095u];
                    if (g < (u32)nranges) atomicAdd(&h[g][(u32)(k[r] >> sh.s[g]) & 255u], 1u);
                }
        }
    }
    __syncthreads();
    for (int g = 0; g < nranges; g++)
        counts[(u64)g * stride + (u64)threadIdx.x * nchunks + blockIdx.x] = h[g][threadIdx.x];
}

// The same with a lane per text word, as rs_scatter_sparse_kernel reads the text (32 positions per lane, the 12-bit
// prefix and the digit off the top 32 bits of the position's window): usable when every range's digit lies in those
// 32 bits, i.e. shift >= 34 - (64 - 2 K) -- the first pass of 3 or 4 top passes always does.  chunk is a multiple of
// RS_TILE and ts.pos0 of 32, so a lane's positions are one word.  (51 -> 19 ms at 30 Gbp.)
__global__ __launch_bounds__(RS_BLOCK) void rs_hist_ranges_words_kernel(TextKeySrc ts, u64 chunk, const u8 *__restrict__ range_of_bin,
                                                                         RangeShifts sh, int nranges, u32 *__restrict__ counts,
                                                                         u64 stride, u32 nchunks) {
    __shared__ u32 h[RS_MAX_RANGES][RS_RADIX];
    __shared__ u8 rob[4096];
    __shared__ u32 rsh[RS_MAX_RANGES];
    const int K = ts.K, nsh = 64 - 2 * K;
    for (u32 i = threadIdx.x; i < RS_MAX_RANGES * RS_RADIX; i += RS_BLOCK) (&h[0][0])[i] = 0;
    for (u32 i = threadIdx.x; i < 4096; i += RS_BLOCK) { const u8 g = range_of_bin[i]; rob[i] = g < nranges ? g : 0xFFu; }
    if (threadIdx.x < RS_MAX_RANGES) rsh[threadIdx.x] = (u32)(sh.s[threadIdx.x] + nsh - 2 - 32);   // key >> s = window >> (s + nsh - 2)
    __syncthreads();
    const u64 beg = (u64)blockIdx.x * chunk;
    const u64 end = beg + chunk < ts.n ? beg + chunk : ts.n;
    for (u64 idx0 = beg + ((u64)threadIdx.x << 5); idx0 < end; idx0 += (u64)RS_BLOCK << 5) {
        const u64 p = ts.pos0 + idx0, g = p >> 5;
        const u64 w0 = ts.text[g], w1 = ts.text[g + 1];
        const u64 sa = ts.sepbits[p >> 6], sbw = ts.sepbits[(p >> 6) + 1];
        const u64 sb = (p & 32ull) ? (sa >> 32) | (sbw << 32) : sa;        // bit t: separator at the word's position + t
        const u32 lim = end - idx0 < 32 ? (u32)(end - idx0) : 32u;
        u32 ok = lim < 32 ? (1u << lim) - 1u : 0xFFFFFFFFu;
        if (wave_any(sb != 0ull)) ok &= ~sep_blocked(sb, K);   // a separator within 64 positions: rare, a uniform branch
#pragma unroll
        for (u32 t = 0; t < 32; t++) {
            const u32 top = t ? (u32)(((w0 << (2 * t)) | (w1 >> (64 - 2 * t))) >> 32) : (u32)(w0 >> 32);
            const u32 gi = rob[top >> 20];
            if (gi != 0xFFu && ((ok >> t) & 1u)) atomicAdd(&h[gi][(top >> rsh[gi]) & 255u], 1u);
        }
    }
    __syncthreads();
    for (int g = 0; g < nranges; g++)
        counts[(u64)g * stride + (u64)threadIdx.x * nchunks + blockIdx.x] = h[g][threadIdx.x];
}

// exclusive scan of one digit's chunk counts in place (workgroup d <-> digit d), digit total to tot[d]
__global__ __launch_bounds__(1024) void rs_scan_digit_kernel(u32 *__restrict__ counts, u32 nchunks,
                                                              u32 *__restrict__ tot) {
    __shared__ u32 wsum[16];
    u32 *v = counts + (u64)blockIdx.x * nchunks;
    u32 per = (nchunks + 1023u) / 1024u;
    u32 beg = threadIdx.x * per;
    u32 end = beg + per < nchunks ? beg + per : nchunks;
    u32 s = 0;
    for (u32 i = beg; i < end; i++) s += v[i];
    u32 incl = wave_scan_incl(s);
    u32 w = threadIdx.x >> 6;
    if (lane_id() == 63) wsum[w] = incl;
    __syncthreads();
    u32 base = 0, all = 0;
    for (u32 i = 0; i < 16; i++) { if (i < w) base += wsum[i]; all += wsum[i]; }
    u32 run = base + incl - s;
    for (u32 i = beg; i < end; i++) { u32 c = v[i]; v[i] = run; run += c; }
    if (threadIdx.x == 0) tot[blockIdx.x] = all;
}
// exclusive scan of the 256 digit totals in place
__global__ __launch_bounds__(RS_RADIX) void rs_scan_tot_kernel(u32 *__restrict__ tot) {
    __shared__ u32 tmp[8];
    u32 t, v = tot[threadIdx.x];
    tot[threadIdx.x] = block_scan_excl(v, tmp, &t);
}

// ---- scatter pass -----------------------------------------------------------------------------------
// A workgroup of SC_NT threads ranks a tile of RS_TILE keys per iteration.  Tile layout: ranking unit u (32 lanes) owns
// items [u*32*SC_ITEMS, ...), round r of it the 32 consecutive items at r*32 (a wave's load covers two stretches of 256 bytes).
//
// What bounds the pass is not the ranking but the write pattern: a tile holds ~16 keys per digit, so writing each
// digit's run straight out means 128-byte pieces at arbitrary alignment, every output line is completed by two
// tiles ~20 us apart, and the partial lines of all resident workgroups (256 digits each) overflow L2 -- measured
// 1.27 ms against 0.79 ms for the same kernel storing sequentially.  So the runs are write-combined in LDS: per
// digit a carry of < 16 keys (one 128-byte line) survives between tiles, and a tile emits only whole aligned lines
// [a0, ae): the carried keys + the head of the tile's run complete the pending line (F1, 16 lanes per line), the
// rest of the run up to the last line boundary follows (F2), and the tail behind it becomes the new carry.  Only
// the first and last line of a digit's chunk range are partial.
#ifndef SC_NT
#define SC_NT 512
#endif
#define SC_ITEMS (RS_TILE / SC_NT)
#define SC_WAVES (SC_NT / 64)
#define SC_LINE 16                     // keys per output line (128 bytes)

// Ranking unit: RS_SUB lanes that share their peer-mask words (see rs_rank_tile).  Item layout of a tile: unit u owns
// the items [u * RS_SUB * SC_ITEMS, ...), round r of it the RS_SUB consecutive items at r * RS_SUB -- with 32-lane units
// a wave's load covers two stretches of 256 bytes per round.
#ifndef RS_HALFWAVE
#define RS_HALFWAVE 1
#endif
#ifndef RS_RANK_SWIZZLE
#define RS_RANK_SWIZZLE 1
#endif
#ifndef RS_RANK_UNIFORM
#define RS_RANK_UNIFORM 1
#endif
#define RS_SUB (RS_HALFWAVE ? 32u : 64u)
#define SC_UNITS (SC_NT / RS_SUB)
// offset of item (unit of this thread, round 0) inside a tile
__device__ __forceinline__ u32 rs_item0() {
    return (threadIdx.x / RS_SUB) * (RS_SUB * SC_ITEMS) + (threadIdx.x % RS_SUB);
}
template <int SRC>
__device__ __forceinline__ u32 rs_load_tile(const u64 *__restrict__ in, const TextKeySrc &ts, u64 tile, u64 end,
                                            u64 (&key)[SC_ITEMS]) {
    const u64 wbase = tile + rs_item0();
    u32 vmask = 0;
#pragma unroll
    for (int r = 0; r < SC_ITEMS; r++) {
        u64 idx = wbase + (u64)r * RS_SUB;
        if (rs_load_key<SRC>(in, ts, idx, end, &key[r])) vmask |= 1u << r;
    }
    return vmask;
}
// a tile of keys computed from the staged text words
__device__ __forceinline__ u32 rs_load_tile_text(const TextKeySrc &ts, TextStage &st, u64 tile, u64 end,
                                                 u64 (&key)[SC_ITEMS]) {
    // a lane takes SC_ITEMS consecutive positions (the first pass may rank the keys in any order)
    rs_stage_text(ts, ts.pos0 + tile, st);
    return rs_staged_keys<SC_ITEMS>(ts, st, tile + (u64)threadIdx.x * SC_ITEMS, end, key);
}

// per-digit flush parameters of a tile
// F1: line [a0, a0+nhead): cc carried keys, then the run from LDS slot ls
struct __attribute__((aligned(8))) ScHead { u32 a0; unsigned short ls; u8 cc, nhead; };
// F2: LDS slot j holds position delta + j; j < lo: written by F1; j < fl: goes out; else to carry[j - fl]
struct __attribute__((aligned(8))) ScBody { u32 delta; short lo, fl; };

struct ScShared {
    u64 skeys[RS_TILE];                   // the tile grouped by digit; doubles as rank state: one word per (unit, digit)
    u64 carry[RS_RADIX][SC_LINE];
    unsigned short wavecnt[SC_UNITS][RS_RADIX];   // first slots (< RS_TILE) of the (unit, digit) runs; full-wave units: counts first
    ScHead head[RS_RADIX];
    ScBody body[RS_RADIX];
    u32 run[RS_RADIX];                    // absolute output position of the digit's next key
    u32 cc[RS_RADIX];                     // carried keys of the digit (positions [run - cc, run))
    u32 scan_tmp[SC_WAVES + 1];
};

// rank state (aliases the first SC_UNITS*256 slots of skeys) to zero
__device__ __forceinline__ void rs_clear_rank_state(ScShared &sh) {
    static_assert(SC_UNITS * RS_RADIX <= RS_TILE, "the rank state fits the tile buffer");
    for (u32 i = threadIdx.x; i < SC_UNITS * RS_RADIX; i += SC_NT) {
        if (!RS_HALFWAVE) (&sh.wavecnt[0][0])[i] = 0;
        sh.skeys[i] = 0ull;
    }
}

// Ranks one loaded tile and stages it in sh.skeys grouped by digit (stable); fills head/body for the flush and
// advances run/cc.  Returns the tile's key count.
// Peers (lanes of a wave round that carry the same digit) are found through LDS instead of one ballot per digit
// bit: every lane ORs its lane bit into the 64-bit word of its digit, reads the word back -- that is the peer mask --
// and the first peer clears it for the next round.  LDS executes a wave's instructions in order, so the read sees
// the whole round's ORs and the clear comes after every read.
// FULL: every item of the tile is a key (the common case of the array passes): no per-item predicates.
template <int FIXED0, int FULL>
__device__ __forceinline__ u32 rs_rank_tile(const u64 (&key)[SC_ITEMS], u32 vmask, const RsDigit &dg, ScShared &sh,
                                            u32 oalign, u32 nrounds = SC_ITEMS) {
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    // rank state is clear: rs_clear_rank_state ran during the previous tile's flush
    u32 pk[SC_ITEMS];                         // rank inside the unit's digit run | digit << 16
#if RS_HALFWAVE
    // One 64-bit word per (32-lane unit, digit): the peer mask of the current round in its low half, the unit's count of
    // the digit so far in its high half.  A lane ORs its bit into the low half (32-bit LDS atomic), reads the word back --
    // mask and count in one read -- and the first peer stores count + peers with the mask cleared: three LDS operations
    // and 20 bytes per key where full-wave masks with separate counters took five and 28 (the pass is bound by the LDS
    // pipe: 60 % of its cycles at 4096 keys per tile).
    u64 *wword = sh.skeys + (tid >> 5) * RS_RADIX;
    const u32 lbit = 1u << (tid & 31u);
#pragma unroll
    for (int r = 0; r < SC_ITEMS; r++) {
        if (!FULL && (u32)r >= nrounds) { pk[r] = 0; continue; }       // (workgroup-uniform) no keys in this round
        const bool valid = FULL || ((vmask >> r) & 1u);
        const u32 d = rs_digit<FIXED0>(dg, key[r]);
#if RS_RANK_SWIZZLE
        // the mask half of a digit's word is its low half when bit 4 of the digit is clear, its high half when set: the
        // 32-bit ORs of a round then spread over all 32 banks (bank = 2 (d mod 16) + bit 4) instead of the 16 even ones
        const u32 hs = (d >> 4) & 1u;
#if RS_RANK_UNIFORM
        // Both units of the wave hold one digit each in this round (keys of a low-complexity stretch: the passes over the
        // oversize buckets of a real genome): 32 ORs into one word would be executed one after the other.  The peer mask
        // is then the mask of the unit's valid lanes -- no OR, one read, one store.
        // (whole tiles only: the first pass of a key range, which ranks what it collected, is bound by its instructions and
        // loses 1.3 ms per range to the test)
        const u32 du = FULL ? (u32)(lane < 32u ? __builtin_amdgcn_readlane((int)d, 0) : __builtin_amdgcn_readlane((int)d, 32)) : 0u;
        if (FULL && __ballot(d != du) == 0ull) {
            const u64 cm = __hip_atomic_load(&wword[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const u32 base = hs ? (u32)cm : (u32)(cm >> 32);
            const u32 before = tid & 31u;
            pk[r] = (base + before) | (d << 16);
            __builtin_amdgcn_wave_barrier();
            if (before == 0) {
                const u32 nc = base + 32u;
                __hip_atomic_store(&wword[d], hs ? (u64)nc : (u64)nc << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __builtin_amdgcn_sched_barrier(0);
            continue;
        }
#endif
        if (valid) __hip_atomic_fetch_or(reinterpret_cast<u32 *>(&wword[d]) + hs, lbit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_wave_barrier();
        const u64 cm = valid ? __hip_atomic_load(&wword[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0ull;
        const u32 m = hs ? (u32)(cm >> 32) : (u32)cm, base = hs ? (u32)cm : (u32)(cm >> 32);
        const u32 before = (u32)__popc(m & (lbit - 1u));
        pk[r] = (base + before) | (d << 16);
        __builtin_amdgcn_wave_barrier();
        if (valid && before == 0) {
            const u32 nc = base + (u32)__popc(m);
            __hip_atomic_store(&wword[d], hs ? (u64)nc : (u64)nc << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#else
        if (valid) __hip_atomic_fetch_or(reinterpret_cast<u32 *>(&wword[d]), lbit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_wave_barrier();
        const u64 cm = valid ? __hip_atomic_load(&wword[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0ull;
        const u32 m = (u32)cm, base = (u32)(cm >> 32);
        const u32 before = (u32)__popc(m & (lbit - 1u));
        pk[r] = (base + before) | (d << 16);
        __builtin_amdgcn_wave_barrier();
        if (valid && before == 0)
            __hip_atomic_store(&wword[d], (u64)(base + (u32)__popc(m)) << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
        __builtin_amdgcn_sched_barrier(0);    // keep a round's arithmetic inside the round (register pressure)
    }
#else
    u64 *wmask = sh.skeys + w * RS_RADIX;
    const u64 lbit = 1ull << lane;
#pragma unroll
    for (int r = 0; r < SC_ITEMS; r++) {
        if (!FULL && (u32)r >= nrounds) { pk[r] = 0; continue; }       // (workgroup-uniform) no keys in this round
        const bool valid = FULL || ((vmask >> r) & 1u);
        const u32 d = rs_digit<FIXED0>(dg, key[r]);
        if (valid) __hip_atomic_fetch_or(&wmask[d], lbit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_wave_barrier();
        const u64 m = valid ? __hip_atomic_load(&wmask[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0ull;
        const u32 before = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));   // popc(m & lanes below)
        const u32 base = sh.wavecnt[w][d];
        pk[r] = (base + before) | (d << 16);
        __builtin_amdgcn_wave_barrier();
        if (valid && before == 0) {
            sh.wavecnt[w][d] = (unsigned short)(base + (u32)__popcll(m));
            __hip_atomic_store(&wmask[d], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __builtin_amdgcn_sched_barrier(0);    // keep a round's arithmetic inside the round (register pressure)
    }
#endif
    lds_barrier();
    // digit d = thread d: counts of the waves -> first LDS slot of every (wave, digit) run; flush parameters
    u32 len = 0, c[SC_UNITS];
    if (tid < RS_RADIX) {
#pragma unroll
        for (u32 i = 0; i < SC_UNITS; i++) {
            c[i] = RS_HALFWAVE ? reinterpret_cast<const u32 *>(&sh.skeys[i * RS_RADIX + tid])[RS_RANK_SWIZZLE ? 1u - ((tid >> 4) & 1u) : 1u]
                               : (u32)sh.wavecnt[i][tid];
            len += c[i];
        }
    }
    u32 incl = wave_scan_incl(len);
    if (lane == 63) sh.scan_tmp[w] = incl;
    lds_barrier();
    u32 tile_total = 0;
    if (tid < RS_RADIX) {
        u32 ls = incl - len;
#pragma unroll
        for (u32 i = 0; i < RS_RADIX / 64; i++) { u32 t = sh.scan_tmp[i]; if (i < w) ls += t; tile_total += t; }
        u32 acc = ls;
#pragma unroll
        for (u32 i = 0; i < SC_UNITS; i++) { sh.wavecnt[i][tid] = (unsigned short)acc; acc += c[i]; }
        const u32 run = sh.run[tid], cc = sh.cc[tid];
        const u32 a0 = run - cc, e = run + len;
        const u32 ae = ((e + oalign) & ~(SC_LINE - 1u)) - oalign;           // last line boundary <= e
        const u32 hl = ((a0 + oalign) | (SC_LINE - 1u)) + 1u - oalign;       // first line boundary > a0
        const bool flush = (int)(ae - a0) > 0;                               // a line completes in this tile
        const u32 delta = run - ls;           // mod 2^32
        sh.head[tid] = ScHead{a0, (unsigned short)ls, (u8)cc, (u8)(flush ? hl - a0 : 0u)};
        sh.body[tid] = ScBody{delta, (short)((flush ? hl : a0) - delta), (short)((flush ? ae : a0) - delta)};
        sh.run[tid] = e;
        sh.cc[tid] = flush ? e - ae : cc + len;
    } else {
#pragma unroll
        for (u32 i = 0; i < RS_RADIX / 64; i++) tile_total += sh.scan_tmp[i];
    }
    lds_barrier();
    u32 slot[SC_ITEMS];                       // all offset reads in flight, then the writes
#pragma unroll
    for (int r = 0; r < SC_ITEMS; r++) slot[r] = sh.wavecnt[tid / RS_SUB][pk[r] >> 16] + (pk[r] & 0xFFFFu);
#pragma unroll
    for (int r = 0; r < SC_ITEMS; r++)
        if (FULL || ((vmask >> r) & 1u)) sh.skeys[slot[r]] = key[r];
    lds_barrier();
    return tile_total;
}

// The same for a pass that owes no order to an earlier one (the first pass of a key range): a key's place inside its digit's
// run of the tile is whatever one returning LDS add on the digit's counter hands out -- one LDS operation per key where the
// stable ranking takes three (OR, read, store) and a count per (32-lane unit, digit) that a digit's thread then has to sum over
// sixteen units.  Which key of a digit comes first differs from run to run; the sorted result cannot (equal keys are equal
// words, and every later pass is stable).  Counters: the first 256 words of the tile buffer, cleared by the caller.
template <int FIXED0>
__device__ __forceinline__ u32 rs_rank_tile_any(const u64 (&key)[SC_ITEMS], u32 vmask, const RsDigit &dg, ScShared &sh, u32 oalign, u32 nrounds) {
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    u32 *cnt = reinterpret_cast<u32 *>(sh.skeys);
    u32 pk[SC_ITEMS];
#pragma unroll
    for (int r = 0; r < SC_ITEMS; r++) {
        pk[r] = 0;
        if ((u32)r >= nrounds) continue;                               // (workgroup-uniform)
        if ((vmask >> r) & 1u) {
            const u32 d = rs_digit<FIXED0>(dg, key[r]);
            pk[r] = __hip_atomic_fetch_add(&cnt[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) | (d << 16);
        }
    }
    lds_barrier();
    u32 len = tid < RS_RADIX ? cnt[tid] : 0u;
    u32 incl = wave_scan_incl(len);
    if (lane == 63) sh.scan_tmp[w] = incl;
    lds_barrier();
    u32 tile_total = 0;
    if (tid < RS_RADIX) {
        u32 ls = incl - len;
#pragma unroll
        for (u32 i = 0; i < RS_RADIX / 64; i++) { u32 t = sh.scan_tmp[i]; if (i < w) ls += t; tile_total += t; }
        sh.wavecnt[0][tid] = (unsigned short)ls;                       // first slot of the digit's run
        const u32 run = sh.run[tid], cc = sh.cc[tid];
        const u32 a0 = run - cc, e = run + len;
        const u32 ae = ((e + oalign) & ~(SC_LINE - 1u)) - oalign;           // last line boundary <= e
        const u32 hl = ((a0 + oalign) | (SC_LINE - 1u)) + 1u - oalign;       // first line boundary > a0
        const bool flush = (int)(ae - a0) > 0;                               // a line completes in this tile
        const u32 delta = run - ls;           // mod 2^32
        sh.head[tid] = ScHead{a0, (unsigned short)ls, (u8)cc, (u8)(flush ? hl - a0 : 0u)};
        sh.body[tid] = ScBody{delta, (short)((flush ? hl : a0) - delta), (short)((flush ? ae : a0) - delta)};
        sh.run[tid] = e;
        sh.cc[tid] = flush ? e - ae : cc + len;
    } else {
#pragma unroll
        for (u32 i = 0; i < RS_RADIX / 64; i++) tile_total += sh.scan_tmp[i];
    }
    lds_barrier();                                                     // the counters are read: the keys may take their place
    u32 slot[SC_ITEMS];
#pragma unroll
    for (int r = 0; r < SC_ITEMS; r++) slot[r] = sh.wavecnt[0][pk[r] >> 16] + (pk[r] & 0xFFFFu);
#pragma unroll
    for (int r = 0; r < SC_ITEMS; r++)
        if ((vmask >> r) & 1u) sh.skeys[slot[r]] = key[r];
    lds_barrier();
    return tile_total;
}

// F1: the pending line of every digit, 16 lanes per digit.  Parameter words first, then all key reads, then
// the stores: the LDS round trips of the iterations overlap instead of chaining.
// what a pass stores: the key, or (STRIP: the last pass of the blue-entry sort) the blue entry of a routed entry
template <int STRIP>
__device__ __forceinline__ u64 rs_out(const RsDigit &dg, u64 e) {
    if (!STRIP) return e;
    return (e & 7ull) | (((e & ((1ull << dg.out_strip) - 1ull)) >> 3) << 4);
}
template <int STRIP = 0>
__device__ __forceinline__ void rs_flush_heads(ScShared &sh, u64 *__restrict__ out, const RsDigit &dg) {
    constexpr u32 NI = RS_RADIX * SC_LINE / SC_NT;
    const u32 s = threadIdx.x % SC_LINE, d0 = threadIdx.x / SC_LINE;
    const u64 *hw = reinterpret_cast<const u64 *>(sh.head);
    const u64 *lds = sh.skeys;                               // skeys and carry are contiguous: one index space
    static_assert(offsetof(ScShared, carry) == sizeof(u64) * RS_TILE, "carry follows skeys");
    u64 h[NI], v[NI];
#pragma unroll
    for (u32 i = 0; i < NI; i++) h[i] = hw[d0 + i * (SC_NT / SC_LINE)];
#pragma unroll
    for (u32 i = 0; i < NI; i++) {
        const u32 d = d0 + i * (SC_NT / SC_LINE);
        const u32 ls = (u32)(h[i] >> 32) & 0xFFFFu, cc = (u32)(h[i] >> 48) & 0xFFu;
        v[i] = lds[s < cc ? RS_TILE + d * SC_LINE + s : ls + s - cc];
    }
#pragma unroll
    for (u32 i = 0; i < NI; i++) {
        const u32 a0 = (u32)h[i], nhead = (u32)(h[i] >> 56);
        if (s < nhead) out[a0 + s] = rs_out<STRIP>(dg, v[i]);
    }
}
// F2: the tile's keys (k[r] = LDS slot tid + r*SC_NT): whole lines go out, the tail behind the last line boundary
// becomes the carry
template <int FIXED0, int STRIP = 0>
__device__ __forceinline__ void rs_flush_body(ScShared &sh, const RsDigit &dg, u64 *__restrict__ out,
                                              const u64 (&k)[SC_ITEMS], int tot) {
    const u64 *bw = reinterpret_cast<const u64 *>(sh.body);
    u32 d[SC_ITEMS];
    u64 b[SC_ITEMS];
#pragma unroll
    for (int r = 0; r < SC_ITEMS; r++) { d[r] = rs_digit<FIXED0>(dg, k[r]); b[r] = bw[d[r]]; }
#pragma unroll
    for (int r = 0; r < SC_ITEMS; r++) {
        const int js = (int)(threadIdx.x + r * SC_NT);
        const int lo = (short)(b[r] >> 32), fl = (short)(b[r] >> 48);
        if (js >= tot) continue;
        if (js >= fl) sh.carry[d[r]][js - fl] = k[r];
        else if (js >= lo) out[(u32)b[r] + (u32)js] = rs_out<STRIP>(dg, k[r]);
    }
}
#ifndef SPARSE_DOUBLE_TEST
#define SPARSE_DOUBLE_TEST 0
#endif
#ifndef RS_WAVES_EU
#define RS_WAVES_EU 4                  // 128 VGPRs: two 512-thread workgroups per CU, as the LDS footprint allows
#endif
template <int SRC, int AUX, int HI>
__global__ __launch_bounds__(SC_NT) __attribute__((amdgpu_waves_per_eu(RS_WAVES_EU, RS_WAVES_EU)))
void rs_scatter_kernel(const u64 *__restrict__ in, TextKeySrc ts, u64 *__restrict__ out, u64 n, u64 chunk, RsDigit dg,
                       const u32 *__restrict__ offsets, const u32 *__restrict__ digit_base, u32 nchunks) {
    constexpr int DG = AUX ? 0 : (HI ? 2 : 1);
    __shared__ ScShared sh;
    __shared__ u64 stext[SRC ? RS_STEXT : 1], ssep[SRC ? RS_SSEP : 1];
    const u32 tid = threadIdx.x;
    if (tid < RS_RADIX) {
        sh.run[tid] = offsets[(u64)tid * nchunks + blockIdx.x] + digit_base[tid];
        sh.cc[tid] = 0;
    }
    const u32 oalign = (u32)(reinterpret_cast<uintptr_t>(out) >> 3) & (SC_LINE - 1u);   // lines are 128-byte aligned addresses
    u64 beg = (u64)blockIdx.x * chunk;
    u64 end = beg + chunk < n ? beg + chunk : n;
    TextStage st{stext, ssep, 0, 0};
    rs_clear_rank_state(sh);
    lds_barrier();
    u64 key[SC_ITEMS];
    const u64 *src = in + rs_item0();
    if (SRC == 0 && beg + RS_TILE <= end) {
#pragma unroll
        for (int r = 0; r < SC_ITEMS; r++) key[r] = src[beg + r * RS_SUB];
    }
    for (u64 tile = beg; tile < end; tile += RS_TILE) {
        int tot = RS_TILE;
        if (SRC == 0 && tile + RS_TILE <= end) {
            // whole tile; its keys were loaded while the previous tile was being flushed
            rs_rank_tile<DG, 1>(key, 0xFFFFFFFFu, dg, sh, oalign);
            if (tile + 2 * RS_TILE <= end) {
#pragma unroll
                for (int r = 0; r < SC_ITEMS; r++) key[r] = src[tile + RS_TILE + r * RS_SUB];
            }
        } else {
            u32 vmask = SRC ? rs_load_tile_text(ts, st, tile, end, key) : rs_load_tile<SRC>(in, ts, tile, end, key);
            tot = (int)rs_rank_tile<DG, 0>(key, vmask, dg, sh, oalign);
        }
        rs_flush_heads<AUX == 2>(sh, out, dg);
        u64 k[SC_ITEMS];
#pragma unroll
        for (int r = 0; r < SC_ITEMS; r++) k[r] = sh.skeys[tid + r * SC_NT];
        lds_barrier();                                         // F1 has read the old carry, every wave holds its slots
        rs_flush_body<DG, AUX == 2>(sh, dg, out, k, tot);
        rs_clear_rank_state(sh);                               // the next tile ranks right after the barrier
        lds_barrier();
    }
    // the last, partial line of every digit
#pragma unroll
    for (u32 i = 0; i < RS_RADIX * SC_LINE / SC_NT; i++) {
        const u32 p = i * SC_NT + tid, d = p / SC_LINE, s = p % SC_LINE;
        const u32 cc = sh.cc[d];
        if (s < cc) out[sh.run[d] - cc + s] = rs_out<AUX == 2>(dg, sh.carry[d][s]);
    }
}

// First pass of a key RANGE (a build in key ranges, a shard that scans the whole text, an exchange round): only a
// fraction of the positions yields a key of the range, and what such a pass costs is finding them.  One lane takes one
// text word = 32 consecutive positions straight from global memory (no staging, no barrier); the range test needs only
// the first 12 bits of a window (ranges are cut at 12-bit prefix bins) and a separator test -- a handful of operations
// per position with compile-time shifts -- and leaves a 32-bit mask of the positions that yield a key.  The keys of the
// marked positions are computed and collected in the tile buffer (the whole of skeys: the rank state that aliases its
// first half is cleared again before the ranking); a full tile, RS_TILE keys from however many positions it took, is
// ranked and flushed like a tile of the array passes.  Any density works: what does not fit the tile stays in the masks
// for the next round of the same words.
// key_lo / key_hi must be multiples of a prefix bin (2^(key bits - 12)), as every caller cuts them.
template <int HI, int AUX = 0>
__global__ __launch_bounds__(SC_NT) __attribute__((amdgpu_waves_per_eu(RS_WAVES_EU, RS_WAVES_EU)))
void rs_scatter_sparse_kernel(TextKeySrc ts, u64 *__restrict__ out, u64 n, u64 chunk, RsDigit dg,
                              const u32 *__restrict__ offsets, const u32 *__restrict__ digit_base, u32 nchunks) {
    constexpr int DG = AUX ? 0 : (HI ? 2 : 1);
    __shared__ ScShared sh;
    __shared__ u32 stab[AUX ? 128 : 1];                        // one bit per 12-mer prefix bin: keys of the bin leave the slice in this round
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    if (tid < RS_RADIX) {
        sh.run[tid] = offsets[(u64)tid * nchunks + blockIdx.x] + digit_base[tid];
        sh.cc[tid] = 0;
    }
    if (AUX && tid < 128) {
        u32 bits = 0;
        for (u32 b = 0; b < 32; b++) bits |= ((ts.bin_tab ? ts.bin_tab[tid * 32 + b] : 0) != 0xFFu ? 1u : 0u) << b;
        stab[tid] = bits;
    }
    const u32 oalign = (u32)(reinterpret_cast<uintptr_t>(out) >> 3) & (SC_LINE - 1u);
    const u64 beg = (u64)blockIdx.x * chunk;                  // a multiple of RS_TILE: word-aligned
    const u64 end = beg + chunk < n ? beg + chunk : n;
    const int K = ts.K, nsh = 64 - 2 * K, kb = 2 * K + 2;
    const u32 lo12 = (u32)(ts.key_lo >> (kb - 12)), hi12 = ts.key_hi ? (u32)(ts.key_hi >> (kb - 12)) : 4096u;
    const u32 span12 = hi12 - lo12;
    lds_barrier();
    const u64 nwords = beg < end ? (end - beg + 31) >> 5 : 0;
    u32 ndense = 0;                                           // keys in the tile buffer (uniform)
    // this lane's words of the next round are fetched while the current round is worked on (two workgroups per CU do
    // not hide a global load that is waited for right away)
    u64 nw0 = 0, nw1 = 0, nwp = 0, nsa = 0, nsb = 0;
    auto fetch = [&](u64 wb) {
        const u64 i0 = beg + ((wb + tid) << 5);
        if (wb < nwords && i0 < end) {
            const u64 p = ts.pos0 + i0, g = p >> 5;
            nw0 = ts.text[g]; nw1 = ts.text[g + 1];
            nwp = g ? ts.text[g - 1] : 3ull;                   // the 'T' that stands at separators goes before the text
            nsa = ts.sepbits[p >> 6]; nsb = ts.sepbits[(p >> 6) + 1];
        }
    };
    fetch(0);
    for (u64 wbase = 0;; wbase += SC_NT) {
        const bool more = wbase < nwords;                     // uniform; the round after the last word flushes the rest
        const u64 idx0 = beg + ((wbase + tid) << 5);          // first item of this lane's word
        u32 m = 0;
        const u64 w0 = nw0, w1 = nw1, wp = nwp, sa = nsa, sbw = nsb;
        if (more) fetch(wbase + SC_NT);
        if (more && idx0 < end) {
            const u32 shp = (u32)((ts.pos0 + idx0) & 63ull);   // 0 or 32
            const u64 sb = shp ? (sa >> 32) | (sbw << 32) : sa; // bit t: separator at the word's position + t
            const u32 lim = end - idx0 < 32 ? (u32)(end - idx0) : 32u;
#pragma unroll
            for (u32 t = 0; t < 32; t++) {
                const u32 pre = t <= 26 ? (u32)(w0 >> (52 - 2 * t)) & 0xFFFu
                                        : (u32)(((w0 << (2 * t)) | (w1 >> (64 - 2 * t))) >> 52);
                const bool in = AUX ? ((stab[pre >> 5] >> (pre & 31u)) & 1u) != 0u : (pre - lo12) < span12;
                m |= (in ? 1u : 0u) << t;
            }
#if SPARSE_DOUBLE_TEST
            {   // DIAGNOSTIC (same result): the prefix tests once more on an opaque copy of the words -- what they cost
                u64 c0 = w0, c1 = w1;
                asm volatile("" : "+v"(c0), "+v"(c1));
                u32 m2 = 0;
#pragma unroll
                for (u32 t = 0; t < 32; t++) {
                    const u32 pre = t <= 26 ? (u32)(c0 >> (52 - 2 * t)) & 0xFFFu
                                            : (u32)(((c0 << (2 * t)) | (c1 >> (64 - 2 * t))) >> 52);
                    m2 |= ((pre - lo12) < span12 ? 1u : 0u) << t;
                }
                m &= m2;
            }
#endif
            if (wave_any(sb != 0ull)) m &= ~sep_blocked(sb, K); // a separator within 64 positions: rare, a uniform branch
            if (lim < 32) m &= (1u << lim) - 1u;
        }
        for (;;) {
            const u32 cnt = (u32)__popc(m);
            const u32 incl = wave_scan_incl(cnt);
            if (lane == 63) sh.scan_tmp[w] = incl;
            lds_barrier();
            u32 woff = incl - cnt, tot = 0;
#pragma unroll
            for (u32 i = 0; i < SC_WAVES; i++) { const u32 t = sh.scan_tmp[i]; if (i < w) woff += t; tot += t; }
            const u32 room = RS_TILE - ndense;
            u32 take = woff >= room ? 0u : (cnt < room - woff ? cnt : room - woff);
            u32 o = ndense + woff;
            while (take--) {
                const u32 t = (u32)__ffs(m) - 1u;
                m &= m - 1u;
                const u64 win = t ? ((w0 << (2 * t)) | (w1 >> (64 - 2 * t))) : w0;
                const u64 pred = t ? (w0 >> (2 * (32 - t))) & 3ull : wp & 3ull;
                sh.skeys[o++] = ((win >> nsh) << 2) | pred;
            }
            const bool rest = tot > room;                      // uniform: keys left in the masks
            ndense += rest ? room : tot;
            lds_barrier();                                     // the appends are visible, scan_tmp is free
            if (ndense == RS_TILE || (!more && ndense)) {
                const u32 R = (ndense + SC_NT - 1) / SC_NT;    // rounds: wave w takes keys [(w*R)*64, (w*R + R)*64)
                u64 key[SC_ITEMS];
                u32 vmask = 0;
#pragma unroll
                for (int r = 0; r < SC_ITEMS; r++) {
                    const u32 idx = (w * R + (u32)r) * 64u + lane;
                    const bool have = (u32)r < R && idx < ndense;
                    key[r] = have ? sh.skeys[idx] : ~0ull;
                    vmask |= (have ? 1u : 0u) << r;
                }
                lds_barrier();                                 // every wave holds its keys: skeys becomes rank state
                rs_clear_rank_state(sh);
                lds_barrier();
                const int tile_tot = (int)rs_rank_tile<DG, 0>(key, vmask, dg, sh, oalign, R);
                rs_flush_heads(sh, out, dg);
                u64 k[SC_ITEMS];
#pragma unroll
                for (int r = 0; r < SC_ITEMS; r++) k[r] = sh.skeys[tid + r * SC_NT];
                lds_barrier();
                rs_flush_body<DG>(sh, dg, out, k, tile_tot);
                lds_barrier();
                ndense = 0;
            }
            if (!rest) break;
        }
        if (!more) break;
    }
#pragma unroll
    for (u32 i = 0; i < RS_RADIX * SC_LINE / SC_NT; i++) {
        const u32 p = i * SC_NT + tid, d = p / SC_LINE, s2 = p % SC_LINE;
        const u32 cc = sh.cc[d];
        if (s2 < cc) out[sh.run[d] - cc + s2] = sh.carry[d][s2];
    }
}

// The same pass with the waves of a workgroup scanning on their own (round 6; the default for AUX == 0).  What the kernel
// above pays for beside the ranking is its collection: per round of 512 words a workgroup-wide scan and two barriers, and the
// waves in lockstep.  Here wave w owns an eighth of the workgroup's words and an eighth of the tile buffer (SEG = 512
// slots): it reads, tests and appends batch after batch of 64 words with nothing but a wave scan between them, until its
// segment is full or its words are used up -- what does not fit stays in the lanes' masks for the next tile -- and only the
// ranking and the flush of the tile are workgroup-wide (one barrier per tile where the collection took two per round).
// The first pass may deal a chunk's keys to its digits' runs in any order (there is no earlier pass whose order it would
// have to keep), and the chunk histograms it is handed count keys per chunk, whatever wave finds them.  The prefix tests
// work on 32-bit windows (v_alignbit): a window lies in the range when (x - lo << 20) < (span << 20), and the borrow of
// that comparison is shifted into the mask by an add-with-carry.
#ifndef SPARSE_W32
#define SPARSE_W32 1                   // the append loop's window by two v_alignbit_b32 instead of two 64-bit shifts (A/B: 0)
#endif
template <int HI, int K31>
__global__ __launch_bounds__(SC_NT) __attribute__((amdgpu_waves_per_eu(RS_WAVES_EU, RS_WAVES_EU)))
void rs_scatter_sparse_waves_kernel(TextKeySrc ts, u64 *__restrict__ out, u64 n, u64 chunk, RsDigit dg,
                                    const u32 *__restrict__ offsets, const u32 *__restrict__ digit_base, u32 nchunks) {
    constexpr int DG = HI ? 2 : 1;
    constexpr u32 SEG = RS_TILE / SC_WAVES;
    static_assert(SEG == 64 * SC_ITEMS, "a wave's segment is its eight rounds of 64 keys");
    __shared__ ScShared sh;
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    if (tid < RS_RADIX) {
        sh.run[tid] = offsets[(u64)tid * nchunks + blockIdx.x] + digit_base[tid];
        sh.cc[tid] = 0;
    }
    const u32 oalign = (u32)(reinterpret_cast<uintptr_t>(out) >> 3) & (SC_LINE - 1u);
    const u64 beg = (u64)blockIdx.x * chunk;                  // a multiple of RS_TILE: word-aligned
    const u64 end = beg + chunk < n ? beg + chunk : n;
    const int K = ts.K, nsh = 64 - 2 * K, kb = 2 * K + 2;
    const u32 lo12 = (u32)(ts.key_lo >> (kb - 12)), hi12 = ts.key_hi ? (u32)(ts.key_hi >> (kb - 12)) : 4096u;
    // a proper key range: span < 4096 bins, so both bounds fit the 16-bit windows the tests work on
    const u32 lo16 = lo12 << 4, span16 = (hi12 - lo12) << 4;
    const u32 lo16p = lo16 | (lo16 << 16), span16p = span16 | (span16 << 16), one16p = 0x00010001u;    // packed pairs
    lds_barrier();
    const u64 nwords = beg < end ? (end - beg + 31) >> 5 : 0;
    const u64 per = ((nwords + SC_WAVES - 1) / SC_WAVES + 63) & ~63ull;
    u64 wcur = (u64)w * per < nwords ? (u64)w * per : nwords;         // this wave's words: [wcur, wend)
    const u64 wend = wcur + per < nwords ? wcur + per : nwords;
    u64 nw0 = 0, nw1 = 0, nwp = 0, nsa = 0, nsb = 0;
    // the batch after the one being worked on (wb < wend, wave-uniform).  The loads are UNCONDITIONAL -- lanes behind the wave's
    // last word read that word again and ignore it -- and their results are touched only when the batch is taken up: a load
    // under a per-lane condition is merged with the old value right behind it, and the wait for it with that (the loads of
    // the lockstep kernel were waited for where they were issued).
    auto fetch = [&](u64 wb) {
        const u64 wl = wb + lane < wend ? wb + lane : wend - 1;
        const u64 p = ts.pos0 + beg + (wl << 5), g = p >> 5;
        nw0 = ts.text[g]; nw1 = ts.text[g + 1];
        nwp = ts.text[g ? g - 1 : 0];                          // (word 0 has no word before it: see where the batch is taken up)
        nsa = ts.sepbits[p >> 6]; nsb = ts.sepbits[(p >> 6) + 1];
    };
    if (wcur < wend) fetch(wcur);
    // the current batch of this lane: A = the symbol before its word and the word's first 31 symbols, B1 = (its last symbol
    // and the next word's first 31) >> 1 -- the 64-bit window that starts one symbol before position t is
    // (A << 2t) | (B1 >> (63 - 2t)) for every t in 0..31, no special case
    u64 A = 0, B1 = 0;
    u32 m = 0;
    bool pending = false;                                             // (wave-uniform) keys of the current batch left in the masks
    for (;;) {
        u32 fill = 0;                                                 // (wave-uniform) keys in this wave's segment
        for (;;) {
            if (!pending) {
                if (wcur >= wend) break;
                const u64 idx0 = beg + ((wcur + lane) << 5);
                const bool mine = wcur + lane < wend && idx0 < end;
                const u64 w0 = nw0, w1 = nw1;
                const u64 wp_ = ts.pos0 + idx0 ? nwp : 3ull;  // the 'T' that stands at separators goes before the text
                A = (wp_ << 62) | (w0 >> 2);
                B1 = ((w0 & 3ull) << 61) | (w1 >> 3);
                const u64 sa = nsa, sbw = nsb;
                wcur += 64;
                if (wcur < wend) fetch(wcur);
                m = 0;
                if (mine) {
                    const u32 a0 = (u32)(w0 >> 32), a1 = (u32)w0, a2 = (u32)(w1 >> 32);
                    // the 32-bit window at position t holds the 16-bit windows of t (high half) and t + 8 (low half): one
                    // packed subtract, one packed saturating subtract (span - y: non-zero exactly when y < span) and one
                    // packed minimum with 1 test both; eight windows cover 16 positions
                    u32 acc0 = 0, acc1 = 0;
#pragma unroll
                    for (int t = 7; t >= 0; t--) {
                        const u32 x0 = t == 0 ? a0 : __builtin_amdgcn_alignbit(a0, a1, 32 - 2 * t);
                        const u32 x1 = t == 0 ? a1 : __builtin_amdgcn_alignbit(a1, a2, 32 - 2 * t);
                        // (inline assembly: the compiler turns the generic vector form into two 16-bit compares and two selects)
                        u32 y0, y1, z0, z1;
                        asm("v_pk_sub_u16 %0, %1, %2" : "=v"(y0) : "v"(x0), "v"(lo16p));
                        asm("v_pk_sub_u16 %0, %1, %2" : "=v"(y1) : "v"(x1), "v"(lo16p));
                        asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(z0) : "v"(span16p), "v"(y0));     // span - y, saturating: non-zero iff y < span
                        asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(z1) : "v"(span16p), "v"(y1));
                        asm("v_pk_min_u16 %0, %1, %2" : "=v"(z0) : "v"(z0), "v"(one16p));
                        asm("v_pk_min_u16 %0, %1, %2" : "=v"(z1) : "v"(z1), "v"(one16p));
                        acc0 = (acc0 << 1) | z0;
                        acc1 = (acc1 << 1) | z1;
                    }
                    // acc: bits 16..23 = positions 0..7 of its half-word, bits 0..7 = positions 8..15
                    m = ((acc0 >> 16) & 0xFFu) | ((acc0 & 0xFFu) << 8) | (((acc1 >> 16) & 0xFFu) << 16) | ((acc1 & 0xFFu) << 24);
                    const u32 lim = end - idx0 < 32 ? (u32)(end - idx0) : 32u;
                    if (lim < 32) m &= (1u << lim) - 1u;
                }
                {   // a separator within 64 positions: rare, behind a wave-uniform branch
                    const u32 shp = (u32)((ts.pos0 + idx0) & 63ull);  // 0 or 32
                    const u64 sb = mine ? (shp ? (sa >> 32) | (sbw << 32) : sa) : 0ull;
                    if (wave_any(sb != 0ull)) m &= ~sep_blocked(sb, K);
                }
            }
            const u32 cnt = (u32)__popc(m);
            const u32 incl = wave_scan_incl(cnt);
            const u32 tot = (u32)__builtin_amdgcn_readlane((int)incl, 63);
            const u32 room = SEG - fill;
            const u32 woff = incl - cnt;
            u32 take = woff >= room ? 0u : (cnt < room - woff ? cnt : room - woff);
            u32 o = w * SEG + fill + woff;
#if SPARSE_W32
            // W by 32-bit funnel shifts: Y = (A : B) >> 1 in four words, W = the 64 bits of Y from bit 2t + 1 on -- both halves
            // one v_alignbit_b32 by 31 - (2t mod 32) of the word pair that 2t / 32 selects (no 64-bit shift: A/B in the header)
            const u64 A1 = A >> 1, B2 = (A << 63) | B1;
            const u32 Y0 = (u32)(A1 >> 32), Y1 = (u32)A1, Y2 = (u32)(B2 >> 32), Y3 = (u32)B2;
            while (take--) {
                const u32 s2 = 2u * (u32)__builtin_ctz(m);               // (take > 0: the mask is not empty)
                m &= m - 1u;
                const bool up = s2 > 31u;
                const u32 ya = up ? Y1 : Y0, yb = up ? Y2 : Y1, yc = up ? Y3 : Y2, shf = s2 ^ 31u;
                sh.skeys[o++] = ((u64)__builtin_amdgcn_alignbit(ya, yb, shf) << 32) | __builtin_amdgcn_alignbit(yb, yc, shf);
            }
#else
            while (take--) {
                const u32 s2 = 2u * (u32)__builtin_ctz(m);               // (take > 0: the mask is not empty)
                m &= m - 1u;
                sh.skeys[o++] = (A << s2) | (B1 >> (63u - s2));          // W: pred(t), then the 31 symbols from t on; the
            }                                                            // key is made of it where the wave takes its keys up
#endif
            pending = tot > room;
            fill = pending ? SEG : fill + tot;
            if (fill == SEG) break;
        }
        const bool more = pending || wcur < wend;
        if (lane == 0) sh.scan_tmp[w] = fill | (more ? 0x10000u : 0u);
        // this wave's keys: its own segment (LDS serves a wave in order: its appends are there)
        u64 key[SC_ITEMS];
        u32 vmask = 0;
#pragma unroll
        for (int r = 0; r < SC_ITEMS; r++) {
            const bool have = (u32)r * 64u + lane < fill;
            const u64 W = sh.skeys[w * SEG + (u32)r * 64u + lane];
            u64 kk;
            if (K31) {                                                  // node << 2 | pred: W rotated left by two bits
                const u32 hi = (u32)(W >> 32), lo = (u32)W;
                kk = ((u64)__builtin_amdgcn_alignbit(hi, lo, 30) << 32) | __builtin_amdgcn_alignbit(lo, hi, 30);
            } else kk = (((W << 2) >> nsh) << 2) | (W >> 62);
            key[r] = have ? kk : ~0ull;
            vmask |= (have ? 1u : 0u) << r;
        }
        lds_barrier();                                                // every wave holds its keys: skeys becomes rank state
        u32 total = 0, maxfill = 0, anymore = 0;
#pragma unroll
        for (u32 i = 0; i < SC_WAVES; i++) {
            const u32 v = sh.scan_tmp[i], f = v & 0xFFFFu;
            total += f; maxfill = f > maxfill ? f : maxfill; anymore |= v >> 16;
        }
        if (total) {                                                  // (workgroup-uniform)
            if (tid < RS_RADIX) reinterpret_cast<u32 *>(sh.skeys)[tid] = 0u;      // the digits' counters (rs_rank_tile_any)
            lds_barrier();
            const int tile_tot = (int)rs_rank_tile_any<DG>(key, vmask, dg, sh, oalign, (maxfill + 63u) / 64u);
            rs_flush_heads(sh, out, dg);
            u64 k[SC_ITEMS];
#pragma unroll
            for (int r = 0; r < SC_ITEMS; r++) k[r] = sh.skeys[tid + r * SC_NT];
            lds_barrier();
            rs_flush_body<DG>(sh, dg, out, k, tile_tot);
            lds_barrier();
        }
        if (!anymore) break;
    }
#pragma unroll
    for (u32 i = 0; i < RS_RADIX * SC_LINE / SC_NT; i++) {
        const u32 p = i * SC_NT + tid, d = p / SC_LINE, s2 = p % SC_LINE;
        const u32 cc = sh.cc[d];
        if (s2 < cc) out[sh.run[d] - cc + s2] = sh.carry[d][s2];
    }
}

// ---------------------------------------------------------------------------------------------------
// hybrid: local finish of prefix buckets in LDS

#ifndef RS_BUCKET_TARGET
#define RS_BUCKET_TARGET 64            // HBM passes until a prefix bucket holds at most this many keys on average (measured:
                                       // 0.3-1 G keys are 5 % faster with three passes and buckets of 18-60 than with four)
#endif
#define RL_CAP 4096                    // keys a 256-thread workgroup finishes
#ifndef RL_H
#define RL_H (RL_CAP / 2)              // its tile stride: tile j starts at the first bucket boundary >= j*RL_H; a tile
#endif                                 // fits whenever no bucket in it exceeds RL_CAP - RL_H keys
#define RLW_CAP 1024                   // keys a single wave finishes (16 per lane, no barriers at all)
#define RLW_H (RLW_CAP * 7 / 8)
#define RLU_LONG 16u                   // unfit stretches of this many raster tiles or more: run-length encoded by several workgroups
#ifndef RLU_LONG_NT
#define RLU_LONG_NT 128                //   threads of a workgroup of the long-stretch launch
#endif
#define RLU_ROWS 64                    //   grid rows of that launch
#ifndef RL_MAX_ROUNDS
#define RL_MAX_ROUNDS 16               // merge-split rounds before a wave tile falls back to the full network
#endif

// first index i >= x (0 < x < n) where the bucket prefix changes (or n); one wave, all lanes return it.
// `reach`: how far a boundary may lie for the tile to fit; beyond it the run's end is found by bisection.
__device__ __forceinline__ u64 rl_boundary(const u64 *__restrict__ keys, u64 n, u64 x, int pshift, u32 reach) {
    const u32 lane = threadIdx.x & 63u;
    const u64 p = keys[x - 1] >> pshift;
    for (u64 base = x; base < x + reach; base += 64) {
        u64 i = base + lane;
        bool diff = i >= n || (keys[i] >> pshift) != p;
        u64 mk = __ballot(diff);
        if (mk) return base + (u64)__ffsll((long long)mk) - 1;
    }
    u64 lo = x + reach, hi = n;                     // a run that overflows the tile: bisect for its end
    while (lo < hi) {
        u64 mid = (lo + hi) >> 1;
        if ((keys[mid] >> pshift) <= p) lo = mid + 1; else hi = mid;
    }
    return lo;
}

#define RL_PAD(x) ((x) + ((x) >> 4))          // one spare word per 16: blocked 128-byte reads hit distinct banks

__device__ __forceinline__ void rl_cex(u64 &a, u64 &b, bool up) {      // (a,b) ascending when up
    bool sw = (a > b) == up;
    u64 lo = sw ? b : a, hi = sw ? a : b;
    a = lo; b = hi;
}

// the 16 keys of a lane ascending: a 60-comparator network in 10 layers (the best known for 16 inputs; checked on all
// 2^16 zero-one inputs) instead of the bitonic sorter's 80
__device__ __forceinline__ void rl_sort16(u64 (&k)[16]) {
#define RL_CE(a, b) rl_cex(k[a], k[b], true)
    RL_CE(0, 13); RL_CE(1, 12); RL_CE(2, 15); RL_CE(3, 14); RL_CE(4, 8); RL_CE(5, 6); RL_CE(7, 11); RL_CE(9, 10);
    RL_CE(0, 5); RL_CE(1, 7); RL_CE(2, 9); RL_CE(3, 4); RL_CE(6, 13); RL_CE(8, 14); RL_CE(10, 15); RL_CE(11, 12);
    RL_CE(0, 1); RL_CE(2, 3); RL_CE(4, 5); RL_CE(6, 8); RL_CE(7, 9); RL_CE(10, 11); RL_CE(12, 13); RL_CE(14, 15);
    RL_CE(0, 2); RL_CE(1, 3); RL_CE(4, 10); RL_CE(5, 11); RL_CE(6, 7); RL_CE(8, 9); RL_CE(12, 14); RL_CE(13, 15);
    RL_CE(1, 2); RL_CE(3, 12); RL_CE(4, 6); RL_CE(5, 7); RL_CE(8, 10); RL_CE(9, 11); RL_CE(13, 14);
    RL_CE(1, 4); RL_CE(2, 6); RL_CE(5, 8); RL_CE(7, 10); RL_CE(9, 13); RL_CE(11, 14);
    RL_CE(2, 4); RL_CE(3, 6); RL_CE(9, 12); RL_CE(11, 13);
    RL_CE(3, 5); RL_CE(6, 8); RL_CE(7, 9); RL_CE(10, 12);
    RL_CE(3, 4); RL_CE(5, 6); RL_CE(7, 8); RL_CE(9, 10); RL_CE(11, 12);
    RL_CE(6, 7); RL_CE(8, 9);
#undef RL_CE
}

// bitonic network over 2^LG keys held 16 per thread in registers (blocked layout: steps at distance < 16 never leave
// the thread, distances < 1024 are lane shuffles, larger ones go through LDS).  Threads with !active (whole waves)
// hold padding only and just keep the barriers.
template <int LG>
__device__ __forceinline__ void rl_network(u64 (&k)[16], u64 *A, const u32 tid, const bool active) {
    constexpr int KPT = 16;
#pragma unroll
    for (int lk = 1; lk <= LG; lk++) {
        const u32 kk = 1u << lk;
#pragma unroll
        for (int lj = lk - 1; lj >= 0; lj--) {
            if (lj < 4) {                                         // partner in the same thread
                const int jj = 1 << lj;
                if (active) {
#pragma unroll
                    for (int r = 0; r < KPT; r++)
                        if ((r & jj) == 0) rl_cex(k[r], k[r | jj], ((tid * KPT + r) & kk) == 0);
                }
            } else if (lj < 10) {                                 // partner lane in the same wave
                const int dl = 1 << (lj - 4);
                const bool lower = (tid & dl) == 0;
                if (active) {
#pragma unroll
                    for (int r = 0; r < KPT; r++) {
                        u64 pk = __shfl_xor(k[r], dl, 64);
                        bool up = ((tid * KPT + r) & kk) == 0;
                        bool take_min = lower == up;
                        bool pless = pk < k[r];
                        k[r] = (take_min == pless) ? pk : k[r];
                    }
                }
            } else {                                              // partner in another wave: through LDS
                const u32 dt = 1u << (lj - 4);
                if (active) {
#pragma unroll
                    for (int r = 0; r < KPT; r++) A[RL_PAD(tid * KPT + r)] = k[r];
                }
                __syncthreads();
                const bool lower = (tid & dt) == 0;
                if (active) {
#pragma unroll
                    for (int r = 0; r < KPT; r++) {
                        u64 pk = A[RL_PAD((tid ^ dt) * KPT + r)];
                        bool up = ((tid * KPT + r) & kk) == 0;
                        bool take_min = lower == up;
                        bool pless = pk < k[r];
                        k[r] = (take_min == pless) ? pk : k[r];
                    }
                }
                __syncthreads();
            }
        }
    }
}

// keys are ordered by their top (key_bits - pshift) bits.  Tile j = [B(j*H), B((j+1)*H)) with B(x) the first
// bucket boundary >= x, so tiles are disjoint and bucket-aligned, and hold <= CAP keys unless a bucket overshoots
// a tile start by more than CAP - H.  A tile is finished by NT threads: a bitonic network over CAP = 16*NT keys
// held 16 per thread in registers (blocked layout: steps at distance < 16 never leave the thread, distances
// 16..512 are wave shuffles, only the 256-thread version has 3 steps through LDS) -- its cost does not depend
// on the key distribution, so runs of equal k-mers (repeat families) cost the same as unique ones.
//   NT = 64  (first, over all tiles): a tile that does not fit marks the 4096-key tiles that cover it;
//   NT = 256 (second, marked tiles only; sorting an already sorted stretch again is harmless): a tile that
//             does not fit is queued for the HBM path.
template <int NT>
__global__ __launch_bounds__(NT) void rs_local_kernel(u64 *__restrict__ keys, u64 n, int pshift,
                                                       u32 *__restrict__ over, u32 over_cap, u8 *__restrict__ mark) {
    constexpr int KPT = 16;
    constexpr u32 CAP = NT * KPT;
    constexpr u32 H = NT == 64 ? RLW_H : RL_H;
    constexpr int LOGN = NT == 64 ? 10 : 12;
    static_assert(NT == 64 || NT == 256, "wave or 4-wave workgroup");
    __shared__ u64 A[CAP + CAP / 16];
    __shared__ u64 sb[2];
    const u32 tid = threadIdx.x;
    if (NT == 256 && !mark[blockIdx.x]) return;
    const u64 x0 = (u64)blockIdx.x * H, x1 = x0 + H;
    if (NT == 64) {
        u64 v0 = x0 == 0 ? 0 : rl_boundary(keys, n, x0, pshift, CAP - H);
        u64 v1 = x1 >= n ? n : rl_boundary(keys, n, x1, pshift, CAP - H);
        if (tid == 0) { sb[0] = v0; sb[1] = v1; }
    } else {
        if ((tid >> 6) == 0) { u64 v = x0 == 0 ? 0 : rl_boundary(keys, n, x0, pshift, CAP - H); if (tid == 0) sb[0] = v; }
        if ((tid >> 6) == 1) { u64 v = x1 >= n ? n : rl_boundary(keys, n, x1, pshift, CAP - H); if (tid == 64) sb[1] = v; }
    }
    __syncthreads();
    const u64 s = sb[0], e = sb[1];
    if (s >= e) return;
    const u64 cnt64 = e - s;
    if (cnt64 > CAP) {
        // a stretch that does not fit is often one long run of equal keys (a repeat family's k-mer in a collection of
        // many genomes): nothing to do when it is in order already
        u32 bad = 0;
        for (u64 i = s + tid; i + 1 < e && !bad; i += NT) bad = keys[i] > keys[i + 1] ? 1u : 0u;
        if (NT == 64 ? (__ballot(bad != 0) == 0ull) : !__syncthreads_or((int)bad)) return;
        if (NT == 64) {
            // hand the stretch to the 4096-key tiles that can overlap it
            u64 t0 = s / RL_H, t1 = (e - 1) / RL_H;
            if (t0 > 0) t0--;
            for (u64 t = t0 + tid; t <= t1; t += NT) mark[t] = 1;
        } else if (tid == 0) {
            u32 idx = atomicAdd(&over[0], 1u);
            if (idx < over_cap) {
                u64 *list = reinterpret_cast<u64 *>(over + 4);
                list[2 * idx] = s; list[2 * idx + 1] = cnt64;
            }
        }
        return;
    }
    const u32 cnt = (u32)cnt64;
    for (u32 i = tid; i < CAP; i += NT) A[RL_PAD(i)] = i < cnt ? keys[s + i] : ~0ull;
    __syncthreads();
    u64 k[KPT];
#pragma unroll
    for (int r = 0; r < KPT; r++) k[r] = A[RL_PAD(tid * KPT + r)];
    {   // in order already (the padding behind cnt is ~0): leave the keys where they are
        u32 bad = 0;
#pragma unroll
        for (int r = 0; r + 1 < KPT; r++) bad |= k[r] > k[r + 1] ? 1u : 0u;
        if (tid + 1 < NT) bad |= k[KPT - 1] > A[RL_PAD((tid + 1) * KPT)] ? 1u : 0u;
        if (NT == 64 ? (__ballot(bad != 0) == 0ull) : !__syncthreads_or((int)bad)) return;
    }
    __syncthreads();
    // Wave tiles first try a network sized for what the data looks like here: the keys arrive ordered by bucket and a
    // bucket holds ~16 keys, so after sorting every lane's 16 keys a key is at most a few lanes from its place.
    // Rounds of merge-splits between neighbouring lanes (even pairs, then odd pairs: a block odd-even transposition
    // sort, ~18 operations per key and round instead of the full network's ~270) run until a ballot finds every lane
    // boundary in order -- runs of equal keys (repeat families) are in order from the start.  Anything that is not
    // sorted after RL_MAX_ROUNDS rounds goes through the full bitonic network below.
    bool sorted = false;
    if (NT == 64) {
        rl_sort16(k);
        for (int round = 0; round < RL_MAX_ROUNDS; round++) {
            u64 nxt = __shfl_down(k[0], 1, 64);
            bool ok = tid == 63 || k[KPT - 1] <= nxt;
            if (__ballot(!ok) == 0ull) { sorted = true; break; }
            // partner lane of this round; lanes without one (0 and 63 in odd rounds) keep their keys
            const bool odd = round & 1;
            const bool lower = ((tid ^ (u32)odd) & 1u) == 0;
            const int partner = lower ? (int)tid + 1 : (int)tid - 1;
            const bool active = partner >= 0 && partner < 64;
            const int src = active ? partner : (int)tid;
            u64 t[KPT];
#pragma unroll
            for (int r = 0; r < KPT; r++) t[r] = __shfl(k[KPT - 1 - r], src, 64);
            if (active) {
                // merge-split: the lower lane keeps the 16 smallest of the 32, the upper lane the 16 largest
#pragma unroll
                for (int r = 0; r < KPT; r++) {
                    bool pless = t[r] < k[r];
                    k[r] = (lower == pless) ? t[r] : k[r];
                }
                // each half is bitonic now: four in-lane merge steps sort it ascending
#pragma unroll
                for (int lj = 3; lj >= 0; lj--) {
                    const int jj = 1 << lj;
#pragma unroll
                    for (int r = 0; r < KPT; r++)
                        if ((r & jj) == 0) rl_cex(k[r], k[r | jj], true);
                }
            }
        }
        if (!sorted) {
            u64 nxt = __shfl_down(k[0], 1, 64);
            bool ok = tid == 63 || k[KPT - 1] <= nxt;
            sorted = __ballot(!ok) == 0ull;
        }
    }
    if (!sorted) {
        // a workgroup tile of at most half the capacity: the network over the lower half of the threads only
        if (NT == 256 && cnt <= CAP / 2) rl_network<LOGN - 1>(k, A, tid, tid < NT / 2);
        else rl_network<LOGN>(k, A, tid, true);
    }
#pragma unroll
    for (int r = 0; r < KPT; r++) A[RL_PAD(tid * KPT + r)] = k[r];
    __syncthreads();
    for (u32 i = tid; i < cnt; i += NT) keys[s + i] = A[RL_PAD(i)];
}

// ---------------------------------------------------------------------------------------------------
// bucket finish that also prepares the run-length encoding of the sorted keys (RleSink)
//
//   rs_local_count  one wave per wave tile: stretch start -> bnd; a fit tile is sorted in registers, written back,
//                   its distinct keys counted -> tcnt and its row symbols (key & 3) written; a stretch above RLW_CAP
//                   keys ("unfit") is listed and, unless in order already, handed to the 4096-key tiles
//   rs_local<256> + the HBM path sort the unfit stretches in place (unchanged)
//   rs_unfit_rle<0> counts the distinct keys of every unfit stretch -> tcnt
//   rs_tile_scan    exclusive scan of tcnt -> tex, total
//   rs_tile_emit / rs_unfit_rle<1>  distinct keys and their first rows, tile by tile at tex
// The separate count pass over the sorted keys (and its 8 bytes per key of reads) is gone.

// sorts the 1024 keys a wave holds 16 per lane (blocked layout) ascending; see rs_local_kernel for the method
__device__ __forceinline__ void rlw_sort(u64 (&k)[16], const u32 lane) {
    constexpr int KPT = 16;
    bool sorted;
    {
        u32 bad = 0;
#pragma unroll
        for (int r = 0; r + 1 < KPT; r++) bad |= k[r] > k[r + 1] ? 1u : 0u;
        const u64 nxt = __shfl_down(k[0], 1, 64);
        if (lane < 63) bad |= k[KPT - 1] > nxt ? 1u : 0u;
        sorted = __ballot(bad != 0) == 0ull;
    }
    if (!sorted) {
        rl_sort16(k);
        for (int round = 0; round < RL_MAX_ROUNDS; round++) {
            const u64 nxt = __shfl_down(k[0], 1, 64);
            const bool ok = lane == 63 || k[KPT - 1] <= nxt;
            if (__ballot(!ok) == 0ull) { sorted = true; break; }
            const bool odd = round & 1;
            const bool lower = ((lane ^ (u32)odd) & 1u) == 0;
            const int partner = lower ? (int)lane + 1 : (int)lane - 1;
            const bool active = partner >= 0 && partner < 64;
            const int src = active ? partner : (int)lane;
            u64 t[KPT];
#pragma unroll
            for (int r = 0; r < KPT; r++) t[r] = __shfl(k[KPT - 1 - r], src, 64);
            if (active) {
#pragma unroll
                for (int r = 0; r < KPT; r++) {
                    const bool pless = t[r] < k[r];
                    k[r] = (lower == pless) ? t[r] : k[r];
                }
#pragma unroll
                for (int lj = 3; lj >= 0; lj--) {
                    const int jj = 1 << lj;
#pragma unroll
                    for (int r = 0; r < KPT; r++)
                        if ((r & jj) == 0) rl_cex(k[r], k[r | jj], true);
                }
            }
        }
        if (!sorted) {
            const u64 nxt = __shfl_down(k[0], 1, 64);
            const bool ok = lane == 63 || k[KPT - 1] <= nxt;
            sorted = __ballot(!ok) == 0ull;
        }
    }
    if (!sorted) {
#pragma unroll
        for (int lk = 1; lk <= 10; lk++) {                        // full bitonic network over the wave
            const u32 kk = 1u << lk;
#pragma unroll
            for (int lj = lk - 1; lj >= 0; lj--) {
                if (lj < 4) {
                    const int jj = 1 << lj;
#pragma unroll
                    for (int r = 0; r < KPT; r++)
                        if ((r & jj) == 0) rl_cex(k[r], k[r | jj], ((lane * KPT + r) & kk) == 0);
                } else {
                    const int dl = 1 << (lj - 4);
                    const bool lower = (lane & dl) == 0;
#pragma unroll
                    for (int r = 0; r < KPT; r++) {
                        const u64 pk = __shfl_xor(k[r], dl, 64);
                        const bool up = ((lane * KPT + r) & kk) == 0;
                        const bool take_min = lower == up;
                        const bool pless = pk < k[r];
                        k[r] = (take_min == pless) ? pk : k[r];
                    }
                }
            }
        }
    }
}

// stg (optional: the sort's other key buffer, free by now): a tile whose distinct keys are at most half of its keys
// leaves them -- and their places in the tile, 16 bits each -- at the tile's own offset there, and says so in its count
// (RLT_STAGED); rs_tile_emit_kernel then reads those instead of the whole tile.  In a collection of ten genomes a
// tenth of the keys are distinct: the emit pass reads 1.1 bytes per key instead of 8.
#define RLT_STAGED 0x80000000u
#ifndef RLT_STAGE_RATIO
#define RLT_STAGE_RATIO 2u             // keys per distinct key from which a tile stages: 10 bytes per distinct key written and read
#endif                                 // again against 8 per key (since such a tile keeps its sorted keys to itself -- drop_sorted --
                                       // 2 beats 4: four genomes 669.8 -> 656.5 ms, ten 1498.3 -> 1493.4, one 155.9 -> 154.8)
#ifndef RLW_WINDOW
#define RLW_WINDOW 1                   // the tile and both its boundaries out of ONE window of keys (one global round trip per tile)
#endif
__global__ __launch_bounds__(64) void rs_local_count_kernel(u64 *__restrict__ keys, u64 n, int pshift,
                                                            u8 *__restrict__ mark, u64 *__restrict__ bnd,
                                                            u32 *__restrict__ unfit, u32 *__restrict__ nunfit,
                                                            u32 *__restrict__ tcnt, u8 *__restrict__ mchar,
                                                            u64 *__restrict__ stg, int drop_sorted) {
    constexpr int KPT = 16;
    constexpr u32 CAP = RLW_CAP;
    const u32 lane = threadIdx.x;
    const u64 x0 = (u64)blockIdx.x * RLW_H, x1 = x0 + RLW_H;
#if RLW_WINDOW
    // Slot i of the window holds position x0 - 1 + i: the key before the raster point (its bucket is the one that may
    // reach into the tile), the 128 positions in which the tile may start, and everything up to the last position at
    // which it may end (x1 + 127 = slot 1024).  The start and the end are found in the window -- the version before
    // went to HBM three times per tile, each trip waiting for the one before: the key before the raster point, the
    // keys behind it, then the tile.
    // A tile that starts within the reach (slot 1 + 127 at the latest) and fits (at most 1024 keys) lies inside 1152 slots.
    constexpr u32 REACH = RLW_CAP - RLW_H, NR = 18, WN = NR * 64;
    static_assert(RLW_H + REACH + 1 <= WN && REACH + RLW_CAP <= WN, "the window holds every boundary slot and every fit tile");
    __shared__ u64 A[WN + WN / 16];
    {
        u64 g[NR];
#pragma unroll
        for (u32 r = 0; r < NR; r++) {
            const u64 p1 = x0 + r * 64u + lane;                   // position + 1
            g[r] = (p1 >= 1 && p1 - 1 < n) ? keys[p1 - 1] : ~0ull;
        }
#pragma unroll
        for (u32 r = 0; r < NR; r++) A[RL_PAD(r * 64u + lane)] = g[r];
    }
    // first slot in (from, from + REACH] whose position lies behind the text or holds another bucket than slot `from`;
    // none: the bucket overflows the reach, its end is found by bisection (as rl_boundary does)
    auto boundary = [&](u32 from) -> u64 {
        const u64 p = A[RL_PAD(from)] >> pshift;
        for (u32 b = from + 1; b <= from + REACH; b += 64) {
            const u32 i = b + lane;
            const u64 pos = x0 - 1 + i;
            const bool diff = pos >= n || (A[RL_PAD(i)] >> pshift) != p;
            const u64 mk = __ballot(diff);
            if (mk) return x0 - 1 + b + (u64)__ffsll((long long)mk) - 1;
        }
        u64 lo = x0 + from + REACH, hi = n;
        while (lo < hi) {
            const u64 mid = (lo + hi) >> 1;
            if ((keys[mid] >> pshift) <= p) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const u64 s = x0 == 0 ? 0 : boundary(0);
    const u64 e = x1 >= n ? n : boundary(RLW_H);
#else
    __shared__ u64 A[CAP + CAP / 16];
    const u64 s = x0 == 0 ? 0 : rl_boundary(keys, n, x0, pshift, RLW_CAP - RLW_H);
    const u64 e = x1 >= n ? n : rl_boundary(keys, n, x1, pshift, RLW_CAP - RLW_H);
#endif
    if (lane == 0) bnd[blockIdx.x] = s;
    if (s >= e) { if (lane == 0) tcnt[blockIdx.x] = 0; return; }
    if (e - s > CAP) {
        if (lane == 0) {
            unfit[atomicAdd(nunfit, 1u)] = blockIdx.x;                         // counted by rs_unfit_rle<0>
            // a stretch of RLU_LONG raster tiles or more is listed again from the end of the array (the tiles inside it
            // are empty, so the two lists never meet): several workgroups share its run-length encoding
            if ((e - x0) / RLW_H >= RLU_LONG) unfit[gridDim.x - 1u - atomicAdd(nunfit + 2, 1u)] = blockIdx.x;
        }
        return;                               // rs_local_unfit_kernel takes the stretch from the list (and tests its order there)
    }
    const u32 cnt = (u32)(e - s);
    u64 k[KPT];
    // LDS operations of one wave execute in order: no barrier between the writes and the reads of its own tile
#if RLW_WINDOW
    if (s - x0 + 1 + cnt <= WN) {
        const u32 off = (u32)(s - x0) + 1u;                       // slot of the tile's first key
#pragma unroll
        for (int r = 0; r < KPT; r++) k[r] = lane * KPT + r < cnt ? A[RL_PAD(off + lane * KPT + r)] : ~0ull;
    } else {
        // the tile starts behind the reach (the bucket of the raster point is a long one, its end was found by
        // bisection) and ends outside the window: read it from where it lies
        for (u32 i = lane; i < CAP; i += 64) A[RL_PAD(i)] = i < cnt ? keys[s + i] : ~0ull;
#pragma unroll
        for (int r = 0; r < KPT; r++) k[r] = A[RL_PAD(lane * KPT + r)];
    }
#else
    for (u32 i = lane; i < CAP; i += 64) A[RL_PAD(i)] = i < cnt ? keys[s + i] : ~0ull;
#pragma unroll
    for (int r = 0; r < KPT; r++) k[r] = A[RL_PAD(lane * KPT + r)];
#endif
    rlw_sort(k, lane);
#pragma unroll
    for (int r = 0; r < KPT; r++) A[RL_PAD(lane * KPT + r)] = k[r];
    // a key is a head when it differs from the key before it; the stretch starts at a bucket boundary
    const u64 before = __shfl_up(k[KPT - 1], 1, 64);
    u32 hm = 0;                                                   // heads among this lane's keys
#pragma unroll
    for (int r = 0; r < KPT; r++) {
        const u64 p = r ? k[r - 1] : before;
        hm |= ((lane * KPT + r < cnt && ((lane == 0 && r == 0) || k[r] != p)) ? 1u : 0u) << r;
    }
    const u32 cl = (u32)__popc(hm), incl = wave_scan_incl(cl);
    const u32 c = (u32)__shfl((int)incl, 63, 64);
    const bool staged = stg != nullptr && RLT_STAGE_RATIO * c <= cnt;
    if (lane == 0) tcnt[blockIdx.x] = c | (staged ? RLT_STAGED : 0u);
    if (staged) {
        u64 *kd = stg + s;
        unsigned short *id = reinterpret_cast<unsigned short *>(stg + s + c);
        u32 o = incl - cl;
#pragma unroll
        for (int r = 0; r < KPT; r++)
            if ((hm >> r) & 1u) { kd[o] = k[r]; id[o] = (unsigned short)(lane * KPT + r); o++; }
    }
    if (lane * KPT < cnt) {                                       // row symbols: 16 bytes per lane
        u8 *dst = mchar + s + lane * KPT;
        if (lane * KPT + KPT <= cnt) {
            u32 wv[4] = {0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < KPT; r++) wv[r >> 2] |= ((u32)k[r] & 3u) << (8 * (r & 3));
            __builtin_memcpy(dst, wv, 16);
        } else {
#pragma unroll
            for (int r = 0; r < KPT; r++) if (lane * KPT + r < cnt) dst[r] = (u8)(k[r] & 3ull);
        }
    }
    // (drop_sorted: nobody reads the sorted tile once its distinct keys are staged -- in a collection of ten genomes that
    // is 8 of the 17 bytes per key this kernel moved; the keys stay where they were, unsorted inside their buckets, which is
    // all a re-sort of the whole array, should the oversize path ask for one, needs)
    if (!(staged && drop_sorted))
        for (u32 i = lane; i < cnt; i += 64) keys[s + i] = A[RL_PAD(i)];
}

// The unfit stretches of the counting finish, sorted where they lie: one workgroup per stretch of the list
// rs_local_count_kernel wrote (a stretch = the buckets between two wave-tile starts, so it is bucket-aligned and can be
// sorted on its own).  The 4096-key tiles of rs_local_kernel<256> are cut on a fixed raster and every unfit stretch
// marks the two or three tiles that overlap it: in a collection of many genomes, where a quarter of the wave tiles
// are unfit, 70 % of all tiles ended up in the network, each re-sorting up to 4096 keys around a stretch of ~1500.
// A stretch above 4096 keys goes to the list of the all-HBM path, as before.
//
// How a stretch is sorted (RLU_CLASSIFY, the default): an unfit stretch is a repeat family's 16-mer in every genome of
// the collection -- a thousand keys of which a few dozen are distinct -- between the small buckets of its neighbours, and
// a comparison network (67,000 VALU instructions per wave for 4096 keys, whether or not the keys are equal) is the
// wrong tool for that.  Instead: 256 evenly spaced keys of the stretch are sorted (one key per thread) and their
// distinct values become splitters; every key finds its class by bisection over the splitters in LDS -- "equal to
// splitter i" or "between splitters i-1 and i" -- and its place inside the class from the class counter (LDS atomic);
// a scan of the class counts gives the class starts and the keys move there.  Keys that equal a splitter are in place
// (any frequent key of the stretch is a splitter); the few keys between two splitters are ordered by counting ranks
// inside their class.  ~60 VALU instructions per key instead of ~470.  A class between splitters that holds more than
// RLU_GAP_MAX keys (a stretch without duplicates whose sample happened to be skewed) sends the stretch to the network.
#ifndef RLU_CLASSIFY
#define RLU_CLASSIFY 1
#endif
#define RLU_GAP_MAX 256u
#ifndef RLU_WAVES_EU
#define RLU_WAVES_EU 4                 // 128 VGPRs: four workgroups per CU, as the LDS footprint allows
#endif
#ifndef RLU_WAVES_EU_SMALL
#define RLU_WAVES_EU_SMALL 7           // 72 VGPRs: seven workgroups of the 2048-key instance per CU
#endif

// sorts 256 keys ascending, one per thread of a 256-thread workgroup (bitonic: distances below 64 by lane shuffles, 64
// and 128 through X)
__device__ __forceinline__ u64 rlu_sort256(u64 v, u64 *X, const u32 tid) {
#pragma unroll
    for (int lk = 1; lk <= 8; lk++) {
        const u32 kk = 1u << lk;
#pragma unroll
        for (int lj = lk - 1; lj >= 0; lj--) {
            const u32 d = 1u << lj;
            u64 p;
            if (lj < 6) p = __shfl_xor(v, (int)d, 64);
            else { X[tid] = v; __syncthreads(); p = X[tid ^ d]; __syncthreads(); }
            const bool take_min = ((tid & d) == 0) == ((tid & kk) == 0);
            const bool pless = p < v;
            v = (take_min == pless) ? p : v;
        }
    }
    return v;
}

#define RLU_NET_FLAG 0x80000000u        // list entry: the stretch is left to the network kernel
#define RLU_COUNTED 0x40000000u         // list entry: the classifying kernel has counted the stretch's distinct keys (and written
                                        // its row symbols); its count says whether it staged them too (RLT_STAGED)
#define RLU_TILE(x) ((x) & 0x3FFFFFFFu)

#if RLU_CLASSIFY
// KPT = 8: the stretches of up to 2048 keys (99 % of them in a collection of ten genomes: a wave tile's worth of small
// buckets and one repeat family's bucket) with half the LDS and half the registers -- seven workgroups per CU instead of
// four, and what these kernels run on is workgroups in flight; KPT = 16: the rest.
template <int KPT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KPT == 8 ? RLU_WAVES_EU_SMALL : RLU_WAVES_EU, KPT == 8 ? RLU_WAVES_EU_SMALL : RLU_WAVES_EU)))
void rs_local_unfit_kernel(u64 *__restrict__ keys, u64 n, const u64 *__restrict__ bnd, u32 nwtiles, u32 *__restrict__ unfit,
                           u32 *__restrict__ nunfit, u32 *__restrict__ over, u32 over_cap, u32 gap_max,
                           u32 *__restrict__ tcnt, u8 *__restrict__ mchar, u64 *__restrict__ stg, int drop_sorted) {
    constexpr int NT = 256;
    constexpr u32 CAP = NT * KPT;
    __shared__ u64 A[CAP + CAP / 16];
    __shared__ u64 spl[NT];            // the sample, then its distinct values
    __shared__ u32 ccnt[2 * NT];       // class 2 i: keys between splitters i - 1 and i; class 2 i + 1: keys equal to splitter i
    __shared__ u32 wtmp[DEBWT_WAVES + 1];
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const u32 nu = nunfit[0];
    for (u32 i = blockIdx.x; i < nu; i += gridDim.x) {
        const u32 t = unfit[i];
        if (t & (RLU_NET_FLAG | RLU_COUNTED)) continue;               // done, or left to the network, by the other instance
        const u64 s = bnd[t], e = t + 1 < nwtiles ? bnd[t + 1] : n;
        const u64 cnt64 = e - s;
        if (KPT == 8 ? cnt64 > CAP : cnt64 <= CAP / 2) continue;      // the other instance's
        __syncthreads();                                              // A of the stretch before has been read
        if (cnt64 > CAP) {
            // often one long run of equal keys (a repeat family's k-mer in every genome): nothing to do when in order
            u32 bad = 0;
            for (u64 j = s + tid; j + 1 < e && !bad; j += NT) bad = keys[j] > keys[j + 1] ? 1u : 0u;
            if (__syncthreads_or((int)bad) && tid == 0) {
                const u32 idx = atomicAdd(&over[0], 1u);
                if (idx < over_cap) {
                    u64 *list = reinterpret_cast<u64 *>(over + 4);
                    list[2 * idx] = s; list[2 * idx + 1] = cnt64;
                }
            }
            continue;
        }
        const u32 cnt = (u32)cnt64;
        for (u32 j = tid; j < CAP; j += NT) A[RL_PAD(j)] = j < cnt ? keys[s + j] : ~0ull;
        __syncthreads();
        // striped: key j of the stretch belongs to thread j % 256 (every thread holds cnt / 256 keys, whatever cnt is)
        u64 kk[KPT];
        u32 bad = 0;
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const u32 j = (u32)r * NT + tid;
            kk[r] = A[RL_PAD(j)];                                     // (~0 behind the stretch)
            if (j + 1 < cnt) bad |= kk[r] > A[RL_PAD(j + 1)] ? 1u : 0u;
        }
        if (!__syncthreads_or((int)bad)) continue;                    // in order already
        // the sample: 256 evenly spaced keys (cnt > 1024, so they are different elements), sorted, distinct values kept
        u64 smp = A[RL_PAD((u32)(((u64)tid * cnt) >> 8))];
        smp = rlu_sort256(smp, spl, tid);
        spl[tid] = smp; ccnt[tid] = 0; ccnt[tid + NT] = 0;
        __syncthreads();
        const bool head = tid == 0 || spl[tid - 1] != smp;
        const u64 bm = __ballot(head);
        if (lane == 0) wtmp[w] = (u32)__popcll(bm);
        __syncthreads();                                              // every spl[tid - 1] has been read
        u32 sbase = 0, U = 0;
#pragma unroll
        for (u32 x = 0; x < DEBWT_WAVES; x++) { const u32 c = wtmp[x]; sbase += x < w ? c : 0u; U += c; }
        const u32 sidx = sbase + (u32)__popcll(bm & lanemask_lt());
        if (head && sidx < NT - 1) spl[sidx] = smp;                  // sidx <= tid: the slots of later threads are untouched
        if (U > NT - 1) U = NT - 1;                                   // 2 U + 1 classes fit the 512 counters
        __syncthreads();
        // class of every key and its place inside the class
        u32 cr[KPT];                                                  // class << 16 | place in the class
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            cr[r] = 0xFFFFFFFFu;
            if ((u32)r * NT + tid < cnt) {
                const u64 key = kk[r];
                u32 lo = 0;                                           // splitters below the key
#pragma unroll
                for (u32 step = NT / 2; step; step >>= 1) {
                    const u32 c = lo + step;
                    if (c <= U && spl[c - 1] < key) lo = c;
                }
                const u32 cls = 2u * lo + ((lo < U && spl[lo] == key) ? 1u : 0u);
                cr[r] = (cls << 16) | atomicAdd(&ccnt[cls], 1u);
            }
            if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);       // four searches in flight, not sixteen (registers)
        }
        __syncthreads();
        // class starts: thread t scans the classes 2 t and 2 t + 1
        const u32 v0 = ccnt[2 * tid], v1 = ccnt[2 * tid + 1];
        u32 total;
        const u32 cbase = block_scan_excl(v0 + v1, wtmp, &total);
        if (__syncthreads_or((int)(v0 > gap_max))) {
            // no duplicates to speak of and a skewed sample: the network's (the keys in HBM are untouched)
            if (tid == 0) { unfit[i] = t | RLU_NET_FLAG; atomicAdd(nunfit + 3, 1u); }
            continue;
        }
        ccnt[2 * tid] = cbase; ccnt[2 * tid + 1] = cbase + v0;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < KPT; r++)
            if (cr[r] != 0xFFFFFFFFu) {
                const u32 pos = ccnt[cr[r] >> 16] + (cr[r] & 0xFFFFu);
                A[RL_PAD(pos)] = kk[r];
                cr[r] = (cr[r] & 0xFFFF0000u) | pos;                  // class << 16 | position (< 4096)
            }
        __syncthreads();
        // keys between two splitters: counting rank inside the class (equal keys keep the order they have there)
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const u32 cls = cr[r] >> 16, pos = cr[r] & 0xFFFFu;
            u32 np = 0xFFFFFFFFu;
            if (cr[r] != 0xFFFFFFFFu && !(cls & 1u)) {
                const u32 st = ccnt[cls], en = ccnt[cls + 1];         // class cls + 1 (odd) always has a counter
                if (en - st > 1) {
                    const u64 key = kk[r];
                    u32 less = 0;
                    for (u32 y = st; y < en; y++) {
                        const u64 ky = A[RL_PAD(y)];
                        less += (ky < key || (ky == key && y < pos)) ? 1u : 0u;
                    }
                    np = st + less;
                }
            }
            cr[r] = np;
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < KPT; r++)
            if (cr[r] != 0xFFFFFFFFu) A[RL_PAD(cr[r])] = kk[r];
        __syncthreads();
        if (!tcnt) {
            for (u32 j = tid; j < cnt; j += NT) keys[s + j] = A[RL_PAD(j)];
        } else {
            for (u32 j = tid; j < cnt; j += NT) mchar[s + j] = (u8)(A[RL_PAD(j)] & 3ull);
            // The stretch lies sorted in LDS: its run-length encoding follows here instead of in two more passes over
            // it (rs_unfit_rle_kernel counts, then emits): distinct keys counted into the stretch's raster slot and, few
            // as they are, staged for rs_tile_emit_kernel like those of a wave tile; the list entry is marked RLU_COUNTED.
            u32 hm = 0;
            u64 prev = tid ? A[RL_PAD(tid * KPT - 1)] : 0ull;
#pragma unroll
            for (int r = 0; r < KPT; r++) {
                const u32 j = tid * KPT + (u32)r;
                const u64 key = A[RL_PAD(j)];
                hm |= ((j < cnt && (j == 0 || key != prev)) ? 1u : 0u) << r;
                prev = key;
            }
            const u32 cl = (u32)__popc(hm);
            u32 c;
            u32 o = block_scan_excl(cl, wtmp, &c);
            const bool staged = stg != nullptr && RLT_STAGE_RATIO * c <= cnt;
            if (tid == 0) { tcnt[t] = c | (staged ? RLT_STAGED : 0u); unfit[i] = t | RLU_COUNTED; }
            if (staged) {
                u64 *kd = stg + s;
                unsigned short *id = reinterpret_cast<unsigned short *>(stg + s + c);
#pragma unroll
                for (int r = 0; r < KPT; r++)
                    if ((hm >> r) & 1u) { kd[o] = A[RL_PAD(tid * KPT + r)]; id[o] = (unsigned short)(tid * KPT + r); o++; }
            }
            // (the sorted stretch goes back only when somebody will read it: see rs_local_count_kernel)
            if (!(staged && drop_sorted))
                for (u32 j = tid; j < cnt; j += NT) keys[s + j] = A[RL_PAD(j)];
        }
    }
}
#endif

// The 4096-key network over a stretch: every stretch of the list when RLU_CLASSIFY is off, else the ones the
// classifying kernel flagged (and nothing at all -- one load per workgroup -- when it flagged none).
__global__ __launch_bounds__(256) void rs_local_unfit_net_kernel(u64 *__restrict__ keys, u64 n, const u64 *__restrict__ bnd,
                                                                 u32 nwtiles, u32 *__restrict__ unfit,
                                                                 const u32 *__restrict__ nunfit, u32 *__restrict__ over,
                                                                 u32 over_cap) {
    constexpr int KPT = 16, NT = 256;
    constexpr u32 CAP = NT * KPT;
    __shared__ u64 A[CAP + CAP / 16];
    const u32 tid = threadIdx.x;
    const u32 nu = nunfit[0];
    if (RLU_CLASSIFY && nunfit[3] == 0) return;
    for (u32 i = blockIdx.x; i < nu; i += gridDim.x) {
        u32 t = unfit[i];
        if (RLU_CLASSIFY) {
            if (!(t & RLU_NET_FLAG)) continue;
            t &= ~RLU_NET_FLAG;
            __syncthreads();                                          // (uniform: every thread read the flagged entry)
            if (tid == 0) unfit[i] = t;
        }
        const u64 s = bnd[t], e = t + 1 < nwtiles ? bnd[t + 1] : n;
        const u64 cnt64 = e - s;
        __syncthreads();                                              // A of the stretch before has been read
        if (cnt64 > CAP) {
            if (RLU_CLASSIFY) continue;                               // (listed for the all-HBM path by the classifying kernel)
            // often one long run of equal keys (a repeat family's k-mer in every genome): nothing to do when in order
            u32 bad = 0;
            for (u64 j = s + tid; j + 1 < e && !bad; j += NT) bad = keys[j] > keys[j + 1] ? 1u : 0u;
            if (__syncthreads_or((int)bad) && tid == 0) {
                const u32 idx = atomicAdd(&over[0], 1u);
                if (idx < over_cap) {
                    u64 *list = reinterpret_cast<u64 *>(over + 4);
                    list[2 * idx] = s; list[2 * idx + 1] = cnt64;
                }
            }
            continue;
        }
        const u32 cnt = (u32)cnt64;
        for (u32 j = tid; j < CAP; j += NT) A[RL_PAD(j)] = j < cnt ? keys[s + j] : ~0ull;
        __syncthreads();
        u64 k[KPT];
#pragma unroll
        for (int r = 0; r < KPT; r++) k[r] = A[RL_PAD(tid * KPT + r)];
        u32 bad = 0;
#pragma unroll
        for (int r = 0; r + 1 < KPT; r++) bad |= k[r] > k[r + 1] ? 1u : 0u;
        if (tid + 1 < NT) bad |= k[KPT - 1] > A[RL_PAD((tid + 1) * KPT)] ? 1u : 0u;
        if (!__syncthreads_or((int)bad)) continue;                    // in order already
        if (cnt <= CAP / 2) rl_network<11>(k, A, tid, tid < NT / 2);
        else rl_network<12>(k, A, tid, true);
#pragma unroll
        for (int r = 0; r < KPT; r++) A[RL_PAD(tid * KPT + r)] = k[r];
        __syncthreads();
        for (u32 j = tid; j < cnt; j += NT) keys[s + j] = A[RL_PAD(j)];
    }
}

// exclusive scan of the tile counts: within blocks of 4096 tiles (tex) + the block offsets (boff), added by the readers
#define RLT_BLOCK 4096
__global__ __launch_bounds__(256) void rs_tile_scan1_kernel(const u32 *__restrict__ tcnt, u32 nwtiles,
                                                            u32 *__restrict__ tex, u32 *__restrict__ bsum) {
    __shared__ u32 wsum[4];
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const u32 i0 = blockIdx.x * RLT_BLOCK + tid * 16u;
    u32 v[16], sum = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) { v[r] = i0 + r < nwtiles ? (tcnt[i0 + r] & ~RLT_STAGED) : 0u; sum += v[r]; }
    u32 inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 x = __shfl_up(inc, d, 64); if (lane >= (u32)d) inc += x; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    u32 run = inc - sum;
    for (u32 x = 0; x < w; x++) run += wsum[x];
#pragma unroll
    for (int r = 0; r < 16; r++) { if (i0 + r < nwtiles) tex[i0 + r] = run; run += v[r]; }
    if (tid == 255) bsum[blockIdx.x] = run;
}
__global__ __launch_bounds__(1024) void rs_tile_scan2_kernel(const u32 *__restrict__ bsum, u32 nb, u32 *__restrict__ boff,
                                                             u32 *__restrict__ total) {
    __shared__ u32 part[1024];
    const u32 tid = threadIdx.x;
    const u32 per = (nb + 1023u) / 1024u;
    const u32 lo = tid * per < nb ? tid * per : nb, hi = lo + per < nb ? lo + per : nb;
    u32 sum = 0;
    for (u32 i = lo; i < hi; i++) sum += bsum[i];
    part[tid] = sum;
    __syncthreads();
    for (u32 d = 1; d < 1024; d <<= 1) {
        const u32 v = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    u32 run = part[tid] - sum;
    for (u32 i = lo; i < hi; i++) { const u32 c = bsum[i]; boff[i] = run; run += c; }
    if (tid == 1023) *total = part[1023];
}

// distinct keys and first rows of the fit tiles (sorted in place by rs_local_count): one wave per tile
__global__ __launch_bounds__(256) void rs_tile_emit_kernel(const u64 *__restrict__ keys, u64 n,
                                                           const u64 *__restrict__ bnd, u32 nwtiles,
                                                           const u32 *__restrict__ tex, const u32 *__restrict__ boff,
                                                           u64 *__restrict__ dk, u32 *__restrict__ dstart,
                                                           const u32 *__restrict__ tcnt, const u64 *__restrict__ stg) {
    const u32 lane = threadIdx.x & 63u;
    const u32 t = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (t >= nwtiles) return;
    const u64 s = bnd[t], e = t + 1 < nwtiles ? bnd[t + 1] : n;
    if (s >= e) return;
    const u32 tc = tcnt[t];
    if (e - s > RLW_CAP && !((tc & RLT_STAGED) && stg)) return;   // an unfit stretch: rs_unfit_rle_kernel's, unless it was staged
    u64 off = (u64)tex[t] + boff[t / RLT_BLOCK];
    if ((tc & RLT_STAGED) && stg) {                               // the tile left its distinct keys in the staging buffer
        const u32 c = tc & ~RLT_STAGED;
        const u64 *kd = stg + s;
        const unsigned short *id = reinterpret_cast<const unsigned short *>(stg + s + c);
        for (u32 i = lane; i < c; i += 64) { dk[off + i] = kd[i]; dstart[off + i] = (u32)s + id[i]; }
        return;
    }
    u64 carry = 0;
    for (u64 p = s; p < e; p += 64) {
        const u64 j = p + lane;
        const bool valid = j < e;
        const u64 k = valid ? keys[j] : 0ull;
        u64 before = __shfl_up(k, 1, 64);                 // the key before: lane below, or the last key of the previous step
        if (lane == 0) before = carry;
        carry = __shfl(k, 63, 64);
        const bool head = valid && (j == s || before != k);
        const u64 bm = __ballot(head);
        if (head) {
            const u64 o = off + (u32)__popcll(bm & ((1ull << lane) - 1ull));
            dk[o] = k; dstart[o] = (u32)j;
        }
        off += (u32)__popcll(bm);
    }
}

// Unfit stretches, 8 consecutive keys per thread and step (16-byte loads).  A stretch that is one run of equal keys (a
// homopolymer's k-mer: millions of instances) is recognised by its ends and has exactly one head.  A long stretch is
// cut into pieces along the raster of the wave tiles: the tiles that lie wholly inside it have nothing of their own
// (rs_local_count found their range empty), so piece k counts into tcnt[t + k] and emits at tex[t + k] -- position
// order is tile order.  LONG = 0: every stretch of fewer than RLU_LONG pieces, one workgroup each; LONG = 1: the long
// list (rs_local_count), the workgroups of grid row y take the pieces y, y + gridDim.y, ... of a stretch (a repeat
// family with 10^6 copies gives stretches of millions of keys that one workgroup used to walk alone).
template <int EMIT, int LONG>
__global__ __launch_bounds__(LONG ? RLU_LONG_NT : 256) void rs_unfit_rle_kernel(const u64 *__restrict__ keys, u64 n,
                                                           const u64 *__restrict__ bnd, u32 nwtiles,
                                                           const u32 *__restrict__ unfit, const u32 *__restrict__ nunfit,
                                                           u32 *__restrict__ tcnt, const u32 *__restrict__ tex,
                                                           const u32 *__restrict__ boff, u64 *__restrict__ dk,
                                                           u32 *__restrict__ dstart, u8 *__restrict__ mchar, int staging_ok) {
    // (the pieces of the long stretches are 896 keys: workgroups of RLU_LONG_NT threads, so that twice as many of them are in flight)
    constexpr u32 NT = LONG ? RLU_LONG_NT : 256, V = 8, STEP = NT * V;
    __shared__ u32 wsum[4];
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const u32 nu = LONG ? nunfit[2] : nunfit[0];
    // piece k of the stretch [s, e) listed under raster tile t (K pieces)
    auto piece = [&](const u32 t, const u64 s, const u64 e, const u32 K, const u32 k, const u64 kfirst, const bool one_run) {
            const u64 x0 = (u64)t * RLW_H;
            const u64 lo = x0 + (u64)k * RLW_H, hi = k + 1 == K ? e : lo + RLW_H;
            const u64 ps = lo > s ? lo : s, pe = hi;                         // the piece's part of the stretch
            u32 run = EMIT ? tex[t + k] + boff[(t + k) / RLT_BLOCK] : 0u;   // distinct keys before the piece
            if (ps >= pe) { if (!EMIT && tid == 0) tcnt[t + k] = 0; return; }
            if (one_run) {
                if (!EMIT) { if (tid == 0) tcnt[t + k] = ps == s ? 1u : 0u; return; }
                if (tid == 0 && ps == s) { dk[run] = kfirst; dstart[run] = (u32)s; }
                const u8 sy = (u8)(kfirst & 3);
                for (u64 j = ps + tid; j < pe; j += NT) mchar[j] = sy;
                return;
            }
            for (u64 p = ps; p < pe; p += STEP) {
                const u64 j0 = p + (u64)tid * V;
                u64 kk[V + 1];
                const bool whole = j0 + V <= pe && ((j0 & 1ull) == 0);
                if (whole) {
#pragma unroll
                    for (u32 q = 0; q < V; q += 2) {
                        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(keys + j0 + q);
                        kk[1 + q] = v.x; kk[2 + q] = v.y;
                    }
                } else {
#pragma unroll
                    for (u32 q = 0; q < V; q++) kk[1 + q] = j0 + q < pe ? keys[j0 + q] : 0ull;
                }
                kk[0] = (j0 > s && j0 < pe) ? keys[j0 - 1] : 0ull;
                u32 heads = 0;
#pragma unroll
                for (u32 q = 0; q < V; q++) {
                    const u64 j = j0 + q;
                    if (j < pe && (j == s || kk[q] != kk[1 + q])) heads |= 1u << q;
                }
                const u32 cnt = (u32)__popc(heads);
                const u32 incl = wave_scan_incl(cnt);
                if (lane == 63) wsum[w] = incl;
                __syncthreads();
                u32 before = incl - cnt, tot = 0;
#pragma unroll
                for (u32 x = 0; x < NT / 64; x++) { const u32 v = wsum[x]; before += x < w ? v : 0u; tot += v; }
                if (EMIT && j0 < pe) {
                    u64 off = (u64)run + before;
                    if (j0 + V <= pe && (reinterpret_cast<uintptr_t>(mchar + j0) & 7u) == 0) {   // the thread's eight row symbols as one word
                        u64 sy = 0;
#pragma unroll
                        for (u32 q = 0; q < V; q++) sy |= (kk[1 + q] & 3ull) << (8 * q);
                        *reinterpret_cast<u64 *>(mchar + j0) = sy;
                    } else {
#pragma unroll
                        for (u32 q = 0; q < V; q++) if (j0 + q < pe) mchar[j0 + q] = (u8)(kk[1 + q] & 3);
                    }
#pragma unroll
                    for (u32 q = 0; q < V; q++) {
                        const u64 j = j0 + q;
                        if (j < pe && ((heads >> q) & 1u)) { dk[off] = kk[1 + q]; dstart[off] = (u32)j; off++; }
                    }
                }
                run += tot;
                __syncthreads();
            }
            if (!EMIT && tid == 0) tcnt[t + k] = run;
    };
    if (LONG) {
        // Stretch i belongs to grid column i mod columns, whose workgroup in row y takes piece y: a stretch of fewer pieces
        // than the grid has rows -- nearly all of them -- is done with that.  The pieces from the rows' number on are dealt
        // over ALL workgroups: pieces y, y + rows, ... in the one column left a stretch of 10^5 pieces (a satellite's k-mer:
        // 131 M of the 543 M oversize keys of real10x3G's first key range) to 64 workgroups, 2,290 pieces each, while the
        // other 4,032 had finished -- 7.8 + 6.2 ms for the two passes of that range against 0.8 + 1.2 for the others.
        // Every workgroup finds those stretches itself, NT list entries at a time, one per thread.  (All workgroups walking
        // the whole list entry by entry was tried first: 7,000 entries x 8,192 waves x 80 instructions, 6 ms a pass.)
        const u32 gx = gridDim.x, gy = gridDim.y, G = gx * gy, W = blockIdx.y * gx + blockIdx.x;
        for (u32 i = blockIdx.x; i < nu; i += gx) {
            const u32 t = RLU_TILE(unfit[nwtiles - 1u - i]);
            const u64 s = bnd[t], e = t + 1 < nwtiles ? bnd[t + 1] : n;
            const u64 K64 = (e - (u64)t * RLW_H) / RLW_H;                      // raster tiles wholly inside [x0, e)
            const u32 K = K64 < 1 ? 1u : (u32)K64;
            if (blockIdx.y >= K) continue;                                     // no piece of this stretch for this grid row: on to the
                                                                               // next before its keys are asked for
            const u64 kfirst = keys[s];
            piece(t, s, e, K, blockIdx.y, kfirst, kfirst == keys[e - 1]);      // (sorted: equal ends = every key of the stretch is equal)
        }
        __shared__ u32 l_n, l_t[NT];
        __shared__ u64 l_s[NT], l_e[NT];
        for (u32 base = 0; base < nu; base += NT) {
            __syncthreads();
            if (tid == 0) l_n = 0;
            __syncthreads();
            if (base + tid < nu) {
                const u32 t = RLU_TILE(unfit[nwtiles - 1u - (base + tid)]);
                const u64 s = bnd[t], e = t + 1 < nwtiles ? bnd[t + 1] : n;
                if ((e - (u64)t * RLW_H) / RLW_H > gy) {
                    const u32 q = atomicAdd(&l_n, 1u);
                    l_t[q] = t; l_s[q] = s; l_e[q] = e;
                }
            }
            __syncthreads();
            const u32 m = l_n;
            for (u32 q = 0; q < m; q++) {
                const u32 t = l_t[q];
                const u64 s = l_s[q], e = l_e[q];
                const u32 K = (u32)((e - (u64)t * RLW_H) / RLW_H);
                const u64 k0 = (u64)gy + (W + G - t % G) % G;
                if (k0 >= K) continue;
                const u64 kfirst = keys[s];
                const bool one_run = kfirst == keys[e - 1];
                for (u64 k = k0; k < K; k += G) piece(t, s, e, K, (u32)k, kfirst, one_run);
            }
        }
        return;
    }
    for (u32 i = blockIdx.x; i < nu; i += gridDim.x) {
        const u32 entry = unfit[i];
        const u32 t = RLU_TILE(entry);
        if (entry & RLU_COUNTED) {
            // counted by the classifying kernel; emitted by rs_tile_emit_kernel when staged (unless the staging buffer was lost)
            if (!EMIT || ((tcnt[t] & RLT_STAGED) && staging_ok)) continue;
        }
        const u64 s = bnd[t], e = t + 1 < nwtiles ? bnd[t + 1] : n;
        const u64 K64 = (e - (u64)t * RLW_H) / RLW_H;                          // raster tiles wholly inside [x0, e)
        // (a stretch the classifying kernel counted has its whole count in the slot of its first tile: one piece)
        const u32 K = (K64 < 1 || (entry & RLU_COUNTED)) ? 1u : (u32)K64;
        if (K >= RLU_LONG) continue;                                           // the long list's
        const u64 kfirst = keys[s];
        const bool one_run = kfirst == keys[e - 1];                            // sorted: every key of the stretch is equal
        for (u32 k = 0; k < K; k++) piece(t, s, e, K, k, kfirst, one_run);
    }
}


// workspace of the counting finish, nw tiles in nb scan blocks:
//   [nunfit, total, pad, pad][bnd u64 x nw][tcnt u32 x nw][unfit u32 x nw][tex u32 x nw][bsum u32 x nb][boff u32 x nb]
static size_t rle_nw(u64 n) { return (size_t)(n / RLW_H + 2); }
static size_t rle_nb(u64 n) { return rle_nw(n) / RLT_BLOCK + 2; }
size_t radix_rle_ws_bytes(u64 n) { return 16 + rle_nw(n) * (8 + 12) + rle_nb(n) * 8 + 64; }

// oversize tiles: gather their keys into one contiguous scratch array / copy the sorted result back.  A workgroup takes
// 1024 consecutive elements of the scratch array: the range its first element lies in comes from ONE bisection of the
// range offsets (uniform: scalar loads), every element then walks on from there -- ranges hold thousands of keys, so
// that is a step or none.  (A bisection per element, 17 dependent loads in front of every 8 bytes moved, made this copy
// run at a fifth of the rate of a radix pass.)
// base (optional): the keys travel COMPACTED -- a stretch's keys share their leading bits up to a small difference, so in
// the scratch array a key is (rank of its stretch) << W | (key - base[rank]) with W bits for the largest such difference:
// the all-HBM sort of the gathered keys then takes the passes of W + bits(rank) bits instead of all 64 (distribution R at
// 30 Gbp: 17,500 stretches per key range, W = 32: six passes instead of eight), and the way back adds the base again.
__global__ __launch_bounds__(256) void rs_over_move(u64 *__restrict__ keys, u64 *__restrict__ scratch, const u64 *__restrict__ list,
                                                    const u64 *__restrict__ offs, u32 nranges, u64 total, int back,
                                                    const u64 *__restrict__ base = nullptr, int W = 0) {
    const u64 g0 = (u64)blockIdx.x * 1024u;
    if (g0 >= total) return;
    u32 lo = 0, hi = nranges;
    while (lo + 1 < hi) { u32 mid = (lo + hi) >> 1; if (offs[mid] <= g0) lo = mid; else hi = mid; }
    const u64 wmask = W ? (1ull << W) - 1ull : 0ull;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const u64 g = g0 + (u64)j * 256u + threadIdx.x;
        if (g >= total) break;
        u32 r = lo;
        while (r + 1 < nranges && offs[r + 1] <= g) r++;
        const u64 src = list[2 * r] + (g - offs[r]);
        if (!base) { if (back) keys[src] = scratch[g]; else scratch[g] = keys[src]; }
        else if (back) keys[src] = (scratch[g] & wmask) + base[r];
        else scratch[g] = ((u64)r << W) | (keys[src] - base[r]);
    }
}
// first and last key of every stretch (they are sorted by their leading bits already): what compacting them takes
__global__ void rs_over_ends(const u64 *__restrict__ keys, const u64 *__restrict__ list, u32 nranges, u64 *__restrict__ ends) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nranges) return;
    ends[2 * r] = keys[list[2 * r]];
    ends[2 * r + 1] = keys[list[2 * r] + list[2 * r + 1] - 1];
}

// ---------------------------------------------------------------------------------------------------

size_t radix_workspace_bytes(u64 max_keys) {
    (void)max_keys;
    return (size_t)RS_RADIX * (RS_MAXCHUNKS + 1) * sizeof(u32);
}

// Chunks of a pass: consecutive tiles per workgroup.  2048 chunks up to 2^29 keys; beyond that chunks of 64 tiles, up
// to 16384 of them: the workgroups resident at one time then work on a narrower stretch of the input, and the 256
// write fronts of each stay within fewer pages -- at 4.29 G keys a pass takes 16.9 ms with 16384 chunks, 17.8 ms with
// 2048 and 18.7 ms with 512 (translation misses); more chunks than that buy nothing, and on small inputs more chunks
// only mean more partial lines at the chunk ends.
static void rs_plan(u64 n, u32 *nchunks, u64 *chunk) {
    u64 tiles = (n + RS_TILE - 1) / RS_TILE;
    u64 c = tiles / RS_CHUNK_TILES;
    if (c < RS_MINCHUNKS) c = RS_MINCHUNKS;
    if (c > RS_MAXCHUNKS) c = RS_MAXCHUNKS;
    if (c > tiles) c = tiles;
    if (c == 0) c = 1;
    u64 tiles_per = (tiles + c - 1) / c;
    if (tiles_per == 0) tiles_per = 1;
    *chunk = tiles_per * RS_TILE;
    *nchunks = (u32)((n + *chunk - 1) / *chunk);
    if (*nchunks == 0) *nchunks = 1;
}

size_t radix_over_bytes(u64 max_keys) { return 16 + (size_t)(max_keys / RL_H + 2) * 16 + 16; }

static u64 *rs_lsd(hipStream_t stream, u64 *a, u64 *b, u64 n, int lo_bit, int hi_bit, const RadixWorkspace &ws,
                   hipEvent_t *pass_events, int max_pairs, int *npairs, const TextKeySrc *text = nullptr,
                   bool aux = false, int strip_last = 0, u64 *third = nullptr, u64 *final_dst = nullptr) {
    // stable LSD passes over bits [lo_bit, hi_bit), 8 bits per pass starting at lo_bit.  With `text` the first
    // pass reads node keys from the text (its index space is the ts->n positions) and writes them to `a`.
    // third (auxiliary sorts of an odd number of passes >= 3): a second scratch buffer, so that the LAST pass writes into `a`
    // -- a -> b, b -> third, third -> b, ..., third -> a -- where two buffers would leave the result in b
    // final_dst (auxiliary sorts whose result belongs in ANOTHER buffer than their input): pass i of P writes into final_dst
    // when P - i is even and else into whichever of `a` / `third` it does not read -- a -> final_dst -> a -> final_dst for an
    // odd P, a -> third -> final_dst -> a -> final_dst for an even one (`b` is not used then)
    u64 *src = a, *dst = b;
    int p = 0, ev_idx = 0;
    const int npasses = (hi_bit - lo_bit + 7) / 8;
    if (final_dst) dst = (npasses - 1) % 2 == 0 ? final_dst : third;
    TextKeySrc none{};
    // (auxiliary sorts spread their bits evenly over the passes -- 29 bits of block id: 8 + 7 + 7 + 7, not 8 + 8 + 8 + 5: the
    // histogram of a 5-bit digit adds 128 keys a wave to 32 counters and took 9.2 ms for 3.9 G blue entries where the 8-bit
    // passes before it took 4.1)
    for (int shift = lo_bit, bits = 0; shift < hi_bit; shift += bits, p++) {
        const int left = hi_bit - shift, passes_left = (left + 7) / 8;
        bits = aux ? (left + passes_left - 1) / passes_left : (left < 8 ? left : 8);
        RsDigit dg{};
        dg.shift = shift; dg.mask = (1u << bits) - 1u;
        const bool from_text = text && p == 0;
        u32 nchunks; u64 chunk;
        rs_plan(from_text ? text->n : n, &nchunks, &chunk);
        u32 *digit_tot = ws.counts + (size_t)RS_RADIX * RS_MAXCHUNKS;
        if (from_text) {
            if (text->pre_counts)
                (void)hipMemcpyAsync(ws.counts, text->pre_counts, (size_t)RS_RADIX * nchunks * sizeof(u32), hipMemcpyDeviceToDevice, stream);
            else
                rs_hist_kernel<1, 0><<<nchunks, RS_BLOCK, 0, stream>>>(nullptr, *text, text->n, chunk, dg, ws.counts, nchunks);
            rs_scan_digit_kernel<<<RS_RADIX, 1024, 0, stream>>>(ws.counts, nchunks, digit_tot);
            rs_scan_tot_kernel<<<1, RS_RADIX, 0, stream>>>(digit_tot);
            const bool sparse = text->key_lo || text->key_hi;                  // a key range: most positions yield nothing
            // a key range read off the text: the waves of a workgroup scan on their own (DEBWT_SPARSE_LOCKSTEP=1: the kernel of
            // rounds 3-5, whose waves collect a tile together -- A/B); a range that is the whole key space of a shard's bin
            // table (AUX) and ranges of 4096 bins keep the old kernel
            static const bool lockstep = getenv("DEBWT_SPARSE_LOCKSTEP") != nullptr;
            const bool waves = sparse && !lockstep && (text->pos0 & 31ull) == 0;
            if (waves && shift >= 32 && text->K == 31) rs_scatter_sparse_waves_kernel<1, 1><<<nchunks, SC_NT, 0, stream>>>(*text, a, text->n, chunk, dg, ws.counts, digit_tot, nchunks);
            else if (waves && shift >= 32) rs_scatter_sparse_waves_kernel<1, 0><<<nchunks, SC_NT, 0, stream>>>(*text, a, text->n, chunk, dg, ws.counts, digit_tot, nchunks);
            else if (waves) rs_scatter_sparse_waves_kernel<0, 0><<<nchunks, SC_NT, 0, stream>>>(*text, a, text->n, chunk, dg, ws.counts, digit_tot, nchunks);
            else if (sparse && shift >= 32) rs_scatter_sparse_kernel<1><<<nchunks, SC_NT, 0, stream>>>(*text, a, text->n, chunk, dg, ws.counts, digit_tot, nchunks);
            else if (sparse) rs_scatter_sparse_kernel<0><<<nchunks, SC_NT, 0, stream>>>(*text, a, text->n, chunk, dg, ws.counts, digit_tot, nchunks);
            else if (shift >= 32) rs_scatter_kernel<1, 0, 1><<<nchunks, SC_NT, 0, stream>>>(nullptr, *text, a, text->n, chunk, dg, ws.counts, digit_tot, nchunks);
            else rs_scatter_kernel<1, 0, 0><<<nchunks, SC_NT, 0, stream>>>(nullptr, *text, a, text->n, chunk, dg, ws.counts, digit_tot, nchunks);
            src = a; dst = b;
            continue;
        }
        bool ev = pass_events && ev_idx < max_pairs;
        if (aux) rs_hist_kernel<0, 1><<<nchunks, RS_BLOCK, 0, stream>>>(src, none, n, chunk, dg, ws.counts, nchunks);
        else rs_hist_kernel<0, 0><<<nchunks, RS_BLOCK, 0, stream>>>(src, none, n, chunk, dg, ws.counts, nchunks);
        rs_scan_digit_kernel<<<RS_RADIX, 1024, 0, stream>>>(ws.counts, nchunks, digit_tot);
        rs_scan_tot_kernel<<<1, RS_RADIX, 0, stream>>>(digit_tot);
        if (ev) (void)hipEventRecord(pass_events[2 * ev_idx], stream);
        if (aux && strip_last && shift + bits >= hi_bit) {
            dg.out_strip = strip_last;
            rs_scatter_kernel<0, 2, 0><<<nchunks, SC_NT, 0, stream>>>(src, none, dst, n, chunk, dg, ws.counts, digit_tot, nchunks);
        } else if (aux) rs_scatter_kernel<0, 1, 0><<<nchunks, SC_NT, 0, stream>>>(src, none, dst, n, chunk, dg, ws.counts, digit_tot, nchunks);
        else if (shift >= 32) rs_scatter_kernel<0, 0, 1><<<nchunks, SC_NT, 0, stream>>>(src, none, dst, n, chunk, dg, ws.counts, digit_tot, nchunks);
        else rs_scatter_kernel<0, 0, 0><<<nchunks, SC_NT, 0, stream>>>(src, none, dst, n, chunk, dg, ws.counts, digit_tot, nchunks);
        if (ev) { (void)hipEventRecord(pass_events[2 * ev_idx + 1], stream); ev_idx++; if (npairs) *npairs = ev_idx; }
        if (final_dst) { src = dst; dst = (npasses - (p + 2)) % 2 == 0 ? final_dst : (src == a ? third : a); }
        else if (third) { src = dst; dst = p + 2 == npasses ? a : (src == b ? third : b); }
        else { u64 *t = src; src = dst; dst = t; }
    }
    return src;
}

size_t radix_text_hist_stride() { return (size_t)RS_RADIX * RS_MAXCHUNKS; }

int radix_first_shift(u64 n, int key_bits, int algo) {
    algo &= 15;
    if (key_bits > 64) key_bits = 64;
    int T = 0;
    while ((n >> (8 * T)) > RS_BUCKET_TARGET && T < 4) T++;
    if (algo != 3 || T == 0 || key_bits - 8 * T < 1) return 0;
    return key_bits - 8 * T;
}

hipError_t radix_text_hist_ranges(hipStream_t stream, const TextKeySrc &text, const u8 *range_of_bin, int key_bits,
                                  const int *shift, int nranges, u32 *counts) {
    if (nranges < 1 || nranges > RS_MAX_RANGES) return hipErrorInvalidValue;
    u32 nchunks; u64 chunk;
    rs_plan(text.n, &nchunks, &chunk);
    RangeShifts sh{};
    for (int i = 0; i < nranges; i++) sh.s[i] = shift[i];
    TextKeySrc all = text;
    all.key_lo = 0; all.key_hi = 0; all.pre_counts = nullptr;            // every key: the range comes from the bin table
    bool words = (text.pos0 & 31ull) == 0 && text.K >= 16;
    for (int i = 0; i < nranges; i++) words = words && shift[i] >= 34 - (64 - 2 * text.K) && shift[i] + (64 - 2 * text.K) - 34 <= 24;
    if (words)
        rs_hist_ranges_words_kernel<<<nchunks, RS_BLOCK, 0, stream>>>(all, chunk, range_of_bin, sh, nranges, counts,
                                                                       (u64)radix_text_hist_stride(), nchunks);
    else
        rs_hist_ranges_kernel<<<nchunks, RS_BLOCK, 0, stream>>>(all, chunk, range_of_bin, key_bits - 12, sh, nranges, counts,
                                                                 (u64)radix_text_hist_stride(), nchunks);
    return hipGetLastError();
}

u64 *radix_sort_bits(hipStream_t stream, u64 *a, u64 *b, u64 n, int lo_bit, int hi_bit, const RadixWorkspace &ws,
                     hipError_t *err, int strip_last, bool *stripped, u64 *third) {
    *err = hipSuccess;
    if (stripped) *stripped = false;
    if (n < 2 || hi_bit <= lo_bit) return a;
    const int npasses = (hi_bit - lo_bit + 7) / 8;
    const bool rotate = strip_last > 0 && stripped && third && npasses % 2 == 1 && npasses >= 3;
    const bool fuse = strip_last > 0 && stripped && (npasses % 2 == 0 || rotate);          // the result comes to lie in `a`
    if (fuse) *stripped = true;
    u64 *r = rs_lsd(stream, a, b, n, lo_bit, hi_bit, ws, nullptr, 0, nullptr, nullptr, true, fuse ? strip_last : 0,
                    rotate ? third : nullptr);
    *err = hipGetLastError();
    return r;
}

// The same sort with the result -- stripped by the last pass when strip_last > 0 -- in `dst` (another buffer than the input
// `a`, which is scratch afterwards); an even number of passes needs `third` for its first hop.  False: that buffer is missing
// (nothing was launched).
bool radix_sort_bits_into(hipStream_t stream, u64 *a, u64 *dst, u64 *third, u64 n, int lo_bit, int hi_bit, const RadixWorkspace &ws,
                          hipError_t *err, int strip_last) {
    *err = hipSuccess;
    const int npasses = (hi_bit - lo_bit + 7) / 8;
    if (n < 2 || npasses < 1 || (npasses % 2 == 0 && !third) || a == dst) return false;
    u64 *r = rs_lsd(stream, a, nullptr, n, lo_bit, hi_bit, ws, nullptr, 0, nullptr, nullptr, true, strip_last, third, dst);
    *err = hipGetLastError();
    return r == dst;
}

u64 *radix_sort_u64(hipStream_t stream, u64 *a, u64 *b, u64 n, int key_bits, const RadixWorkspace &ws, int algo,
                    hipEvent_t *pass_events, int max_pairs, int *npairs, hipError_t *err, const TextKeySrc *text,
                    RleSink *sink) {
    *err = hipSuccess;
    if (sink) { sink->done = false; sink->n_over = 0; }
    if (npairs) *npairs = 0;
    const bool aux = (algo & 16) != 0;       // bit 4: auxiliary sort
    const bool net_only = (algo & 32) != 0;  // bit 5: unfit stretches with keys between the splitters go to the network (tests)
    (void)net_only;
    algo &= 15;
    if (key_bits > 64) key_bits = 64;
    if (!text && (n < 2 || key_bits <= 0)) return a;
    // hybrid: T top digits in HBM so that a bucket holds at most RS_BUCKET_TARGET keys on average, the rest in registers
    int T = 0;
    while ((n >> (8 * T)) > RS_BUCKET_TARGET && T < 4) T++;
    if (algo != 3 || T == 0 || key_bits - 8 * T < 1 || !ws.over || !ws.h_over) {
        u64 *r = rs_lsd(stream, a, b, n, 0, key_bits, ws, pass_events, max_pairs, npairs, text, aux);
        *err = hipGetLastError();
        return r;
    }
    const int pshift = key_bits - 8 * T;
    u64 *src = rs_lsd(stream, a, b, n, pshift, key_bits, ws, pass_events, max_pairs, npairs, text, aux);
    u64 *other = src == a ? b : a;
    (void)hipMemsetAsync(ws.over, 0, 16, stream);
    u32 ntiles = (u32)((n + RL_H - 1) / RL_H), nwtiles = (u32)((n + RLW_H - 1) / RLW_H);
    u8 *mark = reinterpret_cast<u8 *>(ws.skew_list);          // one byte per 4096-key tile
    (void)hipMemsetAsync(mark, 0, ntiles + 1, stream);
    u32 *rle_ctr = nullptr, *rle_tcnt = nullptr, *rle_unfit = nullptr, *rle_tex = nullptr, *rle_bsum = nullptr, *rle_boff = nullptr;
    u64 *rle_bnd = nullptr;
    if (sink) {
        const size_t nw = rle_nw(n);
        rle_ctr = static_cast<u32 *>(sink->ws);
        rle_bnd = reinterpret_cast<u64 *>(rle_ctr + 4);
        rle_tcnt = reinterpret_cast<u32 *>(rle_bnd + nw);
        rle_unfit = rle_tcnt + nw;
        rle_tex = rle_unfit + nw;
        rle_bsum = rle_tex + nw;
        rle_boff = rle_bsum + rle_nb(n);
        (void)hipMemsetAsync(rle_ctr, 0, 16, stream);
        // (the other key buffer takes the distinct keys of the tiles that hold few: it is free until the all-HBM path of
        // oversize stretches, which then works in the buffer of the distinct keys instead)
        rs_local_count_kernel<<<nwtiles, 64, 0, stream>>>(src, n, pshift, mark, rle_bnd, rle_unfit, rle_ctr, rle_tcnt, sink->mchar,
                                                          sink->no_staging ? nullptr : other, sink->drop_sorted ? 1 : 0);
        const u32 ugrid = nwtiles < 16384u ? nwtiles : 16384u;
#if RLU_CLASSIFY
        u64 *const stg = sink->no_staging ? nullptr : other;
        rs_local_unfit_kernel<8><<<ugrid, 256, 0, stream>>>(src, n, rle_bnd, nwtiles, rle_unfit, rle_ctr, ws.over, ws.over_cap,
                                                            net_only ? 0u : RLU_GAP_MAX, rle_tcnt, sink->mchar, stg, sink->drop_sorted ? 1 : 0);
        rs_local_unfit_kernel<16><<<ugrid < 4096u ? ugrid : 4096u, 256, 0, stream>>>(src, n, rle_bnd, nwtiles, rle_unfit, rle_ctr, ws.over,
                                                                                      ws.over_cap, net_only ? 0u : RLU_GAP_MAX,
                                                                                      rle_tcnt, sink->mchar, stg, sink->drop_sorted ? 1 : 0);
#endif
        rs_local_unfit_net_kernel<<<ugrid, 256, 0, stream>>>(src, n, rle_bnd, nwtiles, rle_unfit, rle_ctr, ws.over, ws.over_cap);
    } else {
        rs_local_kernel<64><<<nwtiles, 64, 0, stream>>>(src, n, pshift, ws.over, ws.over_cap, mark);
        rs_local_kernel<256><<<ntiles, 256, 0, stream>>>(src, n, pshift, ws.over, ws.over_cap, mark);
    }
    (void)hipMemcpyAsync(ws.h_over, ws.over, 16, hipMemcpyDeviceToHost, stream);
    if ((*err = hipStreamSynchronize(stream)) != hipSuccess) return src;
    u32 nover = ws.h_over[0];
    bool staging_lost = false;
    if (nover) {
        // heavy buckets (low-complexity k-mers): finish them together with the all-HBM passes
        bool whole = nover > ws.over_cap;
        u64 total = 0;
        std::vector<u64> offs, hlist;
        if (!whole) {
            hlist.resize(2 * (size_t)nover);
            (void)hipMemcpyAsync(hlist.data(), ws.over + 4, 16 * (size_t)nover, hipMemcpyDeviceToHost, stream);
            if ((*err = hipStreamSynchronize(stream)) != hipSuccess) return src;
            const u64 *list = hlist.data();
            // ascending by start: the sorted scratch array maps back onto the ranges in address order
            std::vector<std::pair<u64, u64>> rg(nover);
            for (u32 i = 0; i < nover; i++) rg[i] = {list[2 * i], list[2 * i + 1]};
            std::sort(rg.begin(), rg.end());
            offs.resize(3 * (size_t)nover);
            for (u32 i = 0; i < nover; i++) {
                offs[i] = total; total += rg[i].second;
                offs[nover + 2 * i] = rg[i].first; offs[nover + 2 * i + 1] = rg[i].second;
            }
            if (2 * total + 6 * (u64)nover + 64 > n) whole = true;
        }
        if (getenv("DEBWT_TRACE_SORT")) {
            fprintf(stderr, "key sort of %llu keys: %u oversize stretches with %llu keys (%s)\n", (unsigned long long)n, nover,
                    (unsigned long long)total, whole ? "all keys re-sorted" : "gathered and sorted by all-HBM passes");
            if (!whole) {
                u64 hk[40] = {0}; u32 hn[40] = {0};                    // by size class: stretches of [2^b, 2^(b+1)) keys
                for (u32 i = 0; i < nover; i++) {
                    const u64 cnt = offs[nover + 2 * i + 1];
                    int b = 0;
                    while ((cnt >> (b + 1)) && b < 39) b++;
                    hk[b] += cnt; hn[b]++;
                }
                for (int b = 0; b < 40; b++)
                    if (hn[b]) fprintf(stderr, "    2^%d..: %u stretches, %llu keys\n", b, hn[b], (unsigned long long)hk[b]);
            }
        }
        if (whole) {
            src = rs_lsd(stream, src, other, n, 0, key_bits, ws, nullptr, 0, nullptr);
            staging_lost = true;                                   // both key buffers were overwritten
        } else {
            u64 *const free_buf = sink ? sink->dk : other;            // sink: `other` holds staged distinct keys
            u64 *scratch = free_buf, *tmp = free_buf + total, *d_offs = free_buf + 2 * total;
            (void)hipMemcpyAsync(d_offs, offs.data(), 3 * (size_t)nover * 8, hipMemcpyHostToDevice, stream);
            const u64 *d_list = d_offs + nover;
            u32 grid = (u32)((total + 1023) / 1024);
            // Compacted keys (rs_over_move): the first and last key of every stretch tell how many low bits differ inside a
            // stretch; with the stretch's rank in front, the sort of the gathered keys takes fewer passes than their 64 bits
            u64 *d_ends = d_offs + 3 * (size_t)nover, *d_base = d_ends;              // (the ends are read back before the bases go up)
            const u64 *cbase = nullptr;
            int W = 0, sort_bits = key_bits;
            static const bool plain = getenv("DEBWT_OVER_PLAIN") != nullptr;       // A/B: the 64-bit sort of rounds 2-5
            if (!plain && nover > 1) {
                std::vector<u64> ends(2 * (size_t)nover);
                rs_over_ends<<<(nover + 255) / 256, 256, 0, stream>>>(src, d_list, nover, d_ends);
                (void)hipMemcpyAsync(ends.data(), d_ends, 16 * (size_t)nover, hipMemcpyDeviceToHost, stream);
                if ((*err = hipStreamSynchronize(stream)) != hipSuccess) return src;
                // (a stretch is in the order of its leading 64 - pshift bits only: its first key holds the smallest of those,
                // its last the largest, the low pshift bits are anybody's)
                u64 dmax = 0;
                std::vector<u64> base(nover);
                for (u32 i = 0; i < nover; i++) {
                    base[i] = (ends[2 * i] >> pshift) << pshift;
                    dmax = std::max(dmax, (ends[2 * i + 1] >> pshift) - (ends[2 * i] >> pshift));
                }
                int wb = pshift, rb = 1;
                while (wb < 64 && (dmax >> (wb - pshift))) wb++;
                while ((1ull << rb) < nover) rb++;
                if (wb + rb <= key_bits - 8) {                                         // a pass less at least
                    (void)hipMemcpyAsync(d_base, base.data(), 8 * (size_t)nover, hipMemcpyHostToDevice, stream);
                    if ((*err = hipStreamSynchronize(stream)) != hipSuccess) return src;   // base is host memory
                    cbase = d_base; W = wb; sort_bits = wb + rb;
                }
            }
            rs_over_move<<<grid, 256, 0, stream>>>(src, scratch, d_list, d_offs, nover, total, 0, cbase, W);
            // (the auxiliary kernel names: these short passes must not dilute the profile of the key-range passes)
            u64 *r = rs_lsd(stream, scratch, tmp, total, 0, sort_bits, ws, nullptr, 0, nullptr, nullptr, true);
            rs_over_move<<<grid, 256, 0, stream>>>(src, r, d_list, d_offs, nover, total, 1, cbase, W);
            if ((*err = hipStreamSynchronize(stream)) != hipSuccess) return src;   // offs is host memory
        }
    }
    if (sink) {
        const u32 ug = nwtiles < 2048u ? nwtiles : 2048u;
        const u32 nb = (nwtiles + RLT_BLOCK - 1) / RLT_BLOCK;
        rs_unfit_rle_kernel<0, 0><<<ug, 256, 0, stream>>>(src, n, rle_bnd, nwtiles, rle_unfit, rle_ctr, rle_tcnt, rle_tex,
                                                       rle_boff, sink->dk, sink->dstart, sink->mchar, staging_lost ? 0 : 1);
                rs_unfit_rle_kernel<0, 1><<<dim3(64, RLU_ROWS), RLU_LONG_NT, 0, stream>>>(src, n, rle_bnd, nwtiles, rle_unfit, rle_ctr, rle_tcnt, rle_tex,
                                                       rle_boff, sink->dk, sink->dstart, sink->mchar, staging_lost ? 0 : 1);
        rs_tile_scan1_kernel<<<nb, 256, 0, stream>>>(rle_tcnt, nwtiles, rle_tex, rle_bsum);
        rs_tile_scan2_kernel<<<1, 1024, 0, stream>>>(rle_bsum, nb, rle_boff, rle_ctr + 1);
        rs_tile_emit_kernel<<<(nwtiles + 3) / 4, 256, 0, stream>>>(src, n, rle_bnd, nwtiles, rle_tex, rle_boff, sink->dk, sink->dstart,
                                                                   rle_tcnt, staging_lost ? nullptr : other);
        rs_unfit_rle_kernel<1, 0><<<ug, 256, 0, stream>>>(src, n, rle_bnd, nwtiles, rle_unfit, rle_ctr, rle_tcnt, rle_tex,
                                                       rle_boff, sink->dk, sink->dstart, sink->mchar, staging_lost ? 0 : 1);
                rs_unfit_rle_kernel<1, 1><<<dim3(64, RLU_ROWS), RLU_LONG_NT, 0, stream>>>(src, n, rle_bnd, nwtiles, rle_unfit, rle_ctr, rle_tcnt, rle_tex,
                                                       rle_boff, sink->dk, sink->dstart, sink->mchar, staging_lost ? 0 : 1);
        (void)hipMemcpyAsync(sink->h_total, rle_ctr + 1, sizeof(u32), hipMemcpyDeviceToHost, stream);
        if (sink->h_ctr) (void)hipMemcpyAsync(sink->h_ctr, rle_ctr, 4 * sizeof(u32), hipMemcpyDeviceToHost, stream);
        sink->n_over = nover;
        sink->done = true;
    }
    *err = hipGetLastError();
    return src;
}

// One bucketing pass by destination shard: `count` source items (text positions [text->pos0, +count) when `text`
// is given, else the keys in `src`) -> `dst` grouped by shard; offs_host[0..nshards] receives the group offsets.
hipError_t radix_partition_by_shard(hipStream_t stream, const u64 *src, const TextKeySrc *text, u64 count, u64 *dst,
                                    const RsDigit &dg, u32 nshards, const RadixWorkspace &ws, u64 *offs_host, bool sparse,
                                    u64 capacity) {
    TextKeySrc none{};
    u32 nchunks; u64 chunk;
    rs_plan(count, &nchunks, &chunk);
    u32 *digit_tot = ws.counts + (size_t)RS_RADIX * RS_MAXCHUNKS;
    if (text) {
        TextKeySrc ts = *text;
        rs_hist_kernel<1, 1><<<nchunks, RS_BLOCK, 0, stream>>>(nullptr, ts, count, chunk, dg, ws.counts, nchunks);
    } else {
        rs_hist_kernel<0, 1><<<nchunks, RS_BLOCK, 0, stream>>>(src, none, count, chunk, dg, ws.counts, nchunks);
    }
    rs_scan_digit_kernel<<<RS_RADIX, 1024, 0, stream>>>(ws.counts, nchunks, digit_tot);
    rs_scan_tot_kernel<<<1, RS_RADIX, 0, stream>>>(digit_tot);
    // the group totals are known before anything is written: a buffer that cannot hold them is refused here, not overrun
    u32 tot[RS_RADIX];
    hipError_t e = hipMemcpyAsync(tot, digit_tot, sizeof tot, hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    // digit_tot holds exclusive offsets; the last group ends at the number of valid items
    for (u32 i = 0; i < nshards; i++) offs_host[i] = tot[i];
    offs_host[nshards] = nshards < RS_RADIX ? tot[nshards] : 0;   // first empty digit starts where the data ends
    if (offs_host[nshards] > capacity) return hipErrorInvalidValue;
    if (text && sparse) rs_scatter_sparse_kernel<0, 1><<<nchunks, SC_NT, 0, stream>>>(*text, dst, count, chunk, dg, ws.counts, digit_tot, nchunks);
    else if (text) rs_scatter_kernel<1, 1, 0><<<nchunks, SC_NT, 0, stream>>>(nullptr, *text, dst, count, chunk, dg, ws.counts, digit_tot, nchunks);
    else rs_scatter_kernel<0, 1, 0><<<nchunks, SC_NT, 0, stream>>>(src, none, dst, count, chunk, dg, ws.counts, digit_tot, nchunks);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    return hipGetLastError();
}
