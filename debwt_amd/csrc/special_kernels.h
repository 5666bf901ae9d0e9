// special_kernels.h -- the special-region module on the device (single TU: included by debwt_hip.hip).
//
// What `collect` computes for the N*K suffixes that start at most K-1 symbols before a separator
// (/root/reference/src/collect#$.c:118-157 qsort with cmp :253-311, keys :428-455, special branches :534-598, head /
// tail nodes :468-533) restated as sorts and scans, for collections of many records (contigs, reads: N ~ 10^5..10^8),
// where the reference's comparison sort with two binary searches per comparison and the O(N) insert of
// src/INandOut.c:91-108 are the long pole.  special_host.cpp is the host form of the same module (few records).
//
// 1. Record-start ranks.  Two special suffixes with equal T-padded keys and the separator at the same offset tie on
//    everything up to their '#', and the comparison goes on in the records that follow: their order is the order of
//    the suffixes that start at the next records' first bases.  Those N suffixes are ranked once, by refinement
//    rounds on 21-symbol windows of the text in 3-bit codes (A0 C1 G2 T3 #4 $5 -- plain integer order is the
//    reference's A<C<G<T<#<$ with '#' equal to '#' and the comparison running on past it): round w sorts the still
//    tied record starts by their w-th window inside their tie group; a group that is a single record is done.  The
//    sorts are the stable 8-bit radix passes of the key path (radix_sort_bits) over the field bits, with the element's
//    index below them as payload.
// 2. Special suffixes.  Item (record r, offset d) gets the sort key (T-padded node key, K-1-d, rank of record r+1's
//    start) -- a separator ranks above every base and the padding is the largest base, so among equal keys the suffix
//    that reaches its separator later is the smaller one, and '$' ranks above '#' (the last record's follower rank is
//    the largest) -- sorted by three stable LSD sorts, least significant field first ((5 + rank) + 31 + 31 bits); the
//    item id sits below the field as payload and is not sorted on: the passes are stable.
// 3. Special branches: runs of equal K-windows (same offset and kind of separator) whose symbols K ahead differ, by a
//    head-flag scan over the sorted items; head / tail nodes per record.
#pragma once
#include "common.h"

// 21 symbols from text position p as 3-bit codes, symbol t at bits [60 - 3t, 62 - 3t]
__device__ __forceinline__ u64 sx_window3(const u64 *__restrict__ text, const u64 *__restrict__ sepbits, u64 n, u64 p) {
    const u64 x = text_window(text, p);
    const u64 sb = sep_window(sepbits, p);
    u64 out = 0;
#pragma unroll
    for (int t = 0; t < 21; t++) {
        u64 c = (x >> (62 - 2 * t)) & 3ull;
        if ((sb >> t) & 1ull) c = (p + (u64)t == n - 1) ? 5ull : 4ull;
        out |= c << (60 - 3 * t);
    }
    return out;
}

struct SxText {
    const u64 *text; const u64 *sepbits; const u64 *sep; u64 n; u64 nrec; int K;
    __device__ __forceinline__ u64 rec_start(u64 r) const { return r ? sep[r - 1] + 1 : 0; }
    // T-padded key of the special suffix d symbols before the separator of record r (src/collect#$.c:428-446)
    __device__ __forceinline__ u64 item_key(u64 r, int d) const {
        const u64 p = sep[r] - (u64)d;
        const u64 win = d ? (text_window(text, p) >> (64 - 2 * d)) : 0ull;
        const u64 pad = (1ull << (2 * (K - d))) - 1ull;
        return ((win << (2 * (K - d))) | pad) & ((1ull << (2 * K)) - 1ull);
    }
};

// ---- 1. ranks of the record starts ----------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_sx_init(u32 *__restrict__ ord, u32 *__restrict__ gid, u32 *__restrict__ act, u64 N) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    ord[j] = (u32)j; gid[j] = 0; act[j] = (u32)j;
}

// window of round w for the a-th active place; first sort key = low 32 bits of the window | a.  dep != nullptr: the
// window at the place's own depth (jump rounds, below) instead of the common w
__global__ __launch_bounds__(256) void k_sx_round_keys(SxText T, const u32 *__restrict__ ord, const u32 *__restrict__ act, u64 na, u64 w,
                                int bA, u64 *__restrict__ valbuf, u64 *__restrict__ keyA, const u32 *__restrict__ dep) {
    const u64 a = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= na) return;
    const u64 j = ord[act[a]];
    if (dep) w = dep[act[a]];
    const u64 val = sx_window3(T.text, T.sepbits, T.n, T.rec_start(j) + 21ull * w);
    valbuf[a] = val;
    keyA[a] = ((val & 0xFFFFFFFFull) << bA) | a;
}
// second / third sort key of a round, in the order the pass before left (the passes are stable and sort the key bits
// above the payload only, so the payload -- the active element -- rides along): high 31 bits of the window, tie group
__global__ __launch_bounds__(256) void k_sx_rekey(const u64 *__restrict__ src, const u64 *__restrict__ valbuf, const u32 *__restrict__ gid,
                           const u32 *__restrict__ act, u64 na, int bA, int which, u64 *__restrict__ dst) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= na) return;
    const u64 a = src[i] & ((1ull << bA) - 1ull);
    dst[i] = ((which ? (u64)gid[act[a]] : (valbuf[a] >> 32)) << bA) | a;
}
// records, windows and groups in the order (group, window)
__global__ __launch_bounds__(256) void k_sx_gather(const u64 *__restrict__ sorted, int bA, const u64 *__restrict__ valbuf,
                            const u32 *__restrict__ ord, const u32 *__restrict__ gid, const u32 *__restrict__ act, u64 na,
                            u32 *__restrict__ rec_s, u64 *__restrict__ val_s, u32 *__restrict__ gid_s) {
    const u64 f = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= na) return;
    const u64 a = sorted[f] & ((1ull << bA) - 1ull);
    const u32 p = act[a];
    rec_s[f] = ord[p]; val_s[f] = valbuf[a]; gid_s[f] = gid[p];
}
// heads of the refined groups: ordinal of every element's group, place of every group's head
struct SxHeadF {
    const u32 *gid_s; const u64 *val_s; u32 *ordv; u32 *headpos;
    __device__ u32 count(u64 f) const { return (f == 0 || gid_s[f] != gid_s[f - 1] || val_s[f] != val_s[f - 1]) ? 1u : 0u; }
    __device__ u32 recount(u64 f) const { return count(f); }
    __device__ void emit(u64 f, u32 off, u32 c) const {
        const u32 g = off + c - 1u;
        ordv[f] = g;
        if (c) headpos[g] = (u32)f;
    }
};
__global__ void k_sx_sentinel(u32 *__restrict__ headpos, const u32 *__restrict__ total, u32 value) { headpos[*total] = value; }
// new order and groups at the active places; which places stay active (their group still holds several records)
__global__ __launch_bounds__(256) void k_sx_apply(const u32 *__restrict__ rec_s, const u32 *__restrict__ ordv, const u32 *__restrict__ headpos,
                           const u32 *__restrict__ act, u64 na, u32 *__restrict__ ord, u32 *__restrict__ gid,
                           u8 *__restrict__ stay) {
    const u64 f = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= na) return;
    const u32 g = ordv[f], h = headpos[g], sz = headpos[g + 1] - h;
    const u32 p = act[f];
    ord[p] = rec_s[f];
    gid[p] = act[h];
    stay[f] = sz > 1u;
}
// Jump rounds, for record starts that stay tied for long (duplicated contigs: thousands of symbols, a window round sheds
// 21).  All places of a tie group share one depth (windows already known equal).  Every member walks on beside the
// head of its group until a window differs; the fewest windows any member of the group shares with the head, all members
// share with each other: the group's depth advances by that many at once, and the window round that follows tells at
// least one member apart.  A group of g records is done in at most g - 1 such rounds, however long its records tie.
__global__ __launch_bounds__(256) void k_sx_depth_init(const u32 *__restrict__ act, u64 na, u32 w, u32 *__restrict__ dep) {
    const u64 a = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (a < na) dep[act[a]] = w;
}
// (one WAVE per active place: its lanes compare 64 consecutive windows at a time -- a duplicated contig of 80 kb is 3,800
// windows, walked by one thread one dependent gather after the other it cost milliseconds per round)
__global__ __launch_bounds__(256) void k_sx_lcp_min(SxText T, const u32 *__restrict__ ord, const u32 *__restrict__ gid, const u32 *__restrict__ act,
                             u64 na, const u32 *__restrict__ dep, u32 *__restrict__ gmin) {
    const u64 a = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u32 lane = threadIdx.x & 63u;
    if (a >= na) return;
    const u32 p = act[a], h = gid[p];                         // h: the place of the group's head
    if (p == h) return;
    const u64 pj = T.rec_start(ord[p]) + 21ull * dep[p], ph = T.rec_start(ord[h]) + 21ull * dep[p];
    // (two different suffixes differ at the latest where the one nearer the end reaches '$': windows behind that are not
    // read -- a lane whose window would start behind the text reports a difference)
    for (u32 t0 = 0;; t0 += 64) {
        const u64 qj = pj + 21ull * (t0 + lane), qh = ph + 21ull * (t0 + lane);
        const bool diff = qj >= T.n || qh >= T.n ||
                          sx_window3(T.text, T.sepbits, T.n, qj) != sx_window3(T.text, T.sepbits, T.n, qh);
        const u64 m = __ballot(diff);
        if (m) {
            if (lane == 0) atomicMin(&gmin[h], t0 + (u32)__ffsll((long long)m) - 1u);
            return;
        }
    }
}
__global__ __launch_bounds__(256) void k_sx_depth_add(const u32 *__restrict__ gid, const u32 *__restrict__ act, u64 na,
                               const u32 *__restrict__ gmin, u32 *__restrict__ dep, u32 plus) {
    const u64 a = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= na) return;
    const u32 p = act[a];
    const u32 m = gmin ? gmin[gid[p]] : 0u;
    dep[p] += (m == 0xFFFFFFFFu ? 0u : m) + plus;
}
struct SxStayF {
    const u8 *stay; const u32 *act; u32 *act_new;
    __device__ u32 count(u64 f) const { return stay[f]; }
    __device__ u32 recount(u64 f) const { return stay[f]; }
    __device__ void emit(u64 f, u32 off, u32 c) const { if (c) act_new[off] = act[f]; }
};
__global__ __launch_bounds__(256) void k_sx_rank_of(const u32 *__restrict__ ord, u64 N, u32 *__restrict__ rank) {
    const u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < N) rank[ord[p]] = (u32)p;
}

// ---- 2. the special suffixes in suffix order ------------------------------------------------------------------------
// item i = record i / K, offset d = K-1 - i % K (the order the reference enumerates them, src/collect#$.c:118-131)

// first sort key of every item, and its padded key once for all passes: in enumeration order the separator positions and
// the text windows are read in sequence; the passes then fetch one word per item instead of three scattered ones
// (item record: .x = padded key, .y = text position | BWT symbol << 62)
__global__ __launch_bounds__(256) void k_it_pass1(SxText T, const u32 *__restrict__ rank, u64 NS, int bR, int bP, u64 *__restrict__ key,
                           ulonglong2 *__restrict__ item) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NS) return;
    const u64 r = i / (u64)T.K;
    const u64 e = i - r * (u64)T.K;                            // K-1-d: a later separator first
    const u64 follower = r + 1 < T.nrec ? (u64)rank[r + 1] : T.nrec;     // '$' ranks above every '#'
    key[i] = (((e << bR) | follower) << bP) | i;
    const int d = T.K - 1 - (int)e;
    const u64 p = T.sep[r] - (u64)d;
    item[i] = make_ulonglong2(T.item_key(r, d), p | ((u64)text_symbol(T.text, p - 1) << 62));   // always a base before: records are longer than K
}
// the next pass's key bits above the item id, in the order the pass before left: low / high 31 bits of the padded key
__global__ __launch_bounds__(256) void k_it_rekey(const ulonglong2 *__restrict__ item, const u64 *__restrict__ src, u64 NS, int hi, int bP,
                           u64 *__restrict__ dst) {
    const u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= NS) return;
    const u64 i = src[q] & ((1ull << bP) - 1ull);
    const u64 k62 = item[i].x;
    dst[q] = ((hi ? (k62 >> 31) : (k62 & 0x7FFFFFFFull)) << bP) | i;
}
__global__ __launch_bounds__(256) void k_it_out(SxText T, const ulonglong2 *__restrict__ item, const u64 *__restrict__ sorted, int bP, u64 NS,
                         u64 *__restrict__ spkey, u8 *__restrict__ spchr, u64 *__restrict__ sppos, u32 *__restrict__ sprec,
                         u8 *__restrict__ spd) {
    const u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= NS) return;
    const u64 i = sorted[s] & ((1ull << bP) - 1ull), r = i / (u64)T.K;
    const ulonglong2 it = item[i];                             // one 16-byte fetch per item
    spkey[s] = it.x;
    spchr[s] = (u8)(it.y >> 62);
    sppos[s] = it.y & ((1ull << 62) - 1ull);
    sprec[s] = (u32)r;
    spd[s] = (u8)(T.K - 1 - (int)(i - r * (u64)T.K));         // offset of the separator
}

// ---- 3. special branches, head and tail nodes -----------------------------------------------------------------------

// equal K-windows with the separator of the same kind at the same offset (src/collect#$.c:603-634)
struct SxBranchF {
    SxText T; const u64 *sppos; const u32 *sprec; const u64 *spkey; const u8 *spd; u8 *head; u32 *grp;
    __device__ bool same(u64 a, u64 b) const {
        // equal windows have equal padded keys, separator offsets and kinds: all read in sequence, the text only then
        if (spkey[a] != spkey[b]) return false;
        const u64 da = spd[a], db = spd[b];
        const u64 ra = sprec[a], rb = sprec[b];
        if (da != db || (ra == T.nrec - 1) != (rb == T.nrec - 1)) return false;
        const u64 pa = sppos[a], pb = sppos[b];
        const u64 slot = 3ull << (2 * (31 - da));
        const u64 keep = (~0ull << (64 - 2 * T.K)) & ~slot;
        return ((text_window(T.text, pa) ^ text_window(T.text, pb)) & keep) == 0;
    }
    __device__ u32 count(u64 s) const { const u32 h = (s == 0 || !same(s - 1, s)) ? 1u : 0u; head[s] = (u8)h; return h; }
    __device__ u32 recount(u64 s) const { return head[s]; }
    __device__ void emit(u64 s, u32 off, u32 c) const { grp[s] = off + c - 1u; }
};
// a group whose members do not all continue with the same symbol K ahead is a branch (src/collect#$.c:540-593)
__global__ __launch_bounds__(256) void k_br_diff(SxText T, const u64 *__restrict__ sppos, const u32 *__restrict__ grp, u64 NS,
                          u8 *__restrict__ gflag) {
    const u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (s == 0 || s >= NS) return;
    if (grp[s] != grp[s - 1]) return;
    if (text_symbol(T.text, sppos[s] + (u64)T.K) != text_symbol(T.text, sppos[s - 1] + (u64)T.K)) gflag[grp[s]] = 1;
}
struct SxBranchEmitF {
    const u8 *gflag; const u32 *grp; const u64 *sppos; u64 *branch;
    __device__ u32 count(u64 s) const { return gflag[grp[s]]; }
    __device__ u32 recount(u64 s) const { return gflag[grp[s]]; }
    __device__ void emit(u64 s, u32 off, u32 c) const { if (c) branch[off] = sppos[s]; }
};
// node at the start of every record (| 3: the fake predecessor of a record start) and node in front of every
// separator (| 1: multi-out fact), src/collect#$.c:468-533
__global__ __launch_bounds__(256) void k_heads_tails(SxText T, u64 *__restrict__ head_keys, u64 *__restrict__ tail_facts) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= T.nrec) return;
    head_keys[r] = ((text_window(T.text, T.rec_start(r)) >> (64 - 2 * T.K)) << 2) | 3ull;
    tail_facts[r] = ((text_window(T.text, T.sep[r] - (u64)T.K) >> (64 - 2 * T.K)) << 2) | 1ull;
}
// first special suffixes with key >= lo and >= hi (hi = 0: none above): the slice of a key range
__global__ void k_sp_bounds(const u64 *__restrict__ spkey, u64 NS, u64 lo, u64 hi, u64 *__restrict__ out) {
    if (threadIdx.x == 0) out[0] = lower_bound_dev<u64>(spkey, 0, NS, lo);
    if (threadIdx.x == 1) out[1] = hi ? lower_bound_dev<u64>(spkey, 0, NS, hi) : NS;
}
