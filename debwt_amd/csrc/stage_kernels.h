// stage_kernels.h -- device kernels of the deBWT path other than the radix sort (single TU: included
// by debwt_hip.hip only).  Notation follows SURVEY 8: K = k-1 node length, key = node<<2 | pred.
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------------------------------
// chunked two-pass scan/compaction: every functor F has
//     u32  count(u64 i)                 items produced at index i (may write side outputs)
//     u32  recount(u64 i)               the same value again in the emit pass (may read what count stored)
//     void emit(u64 i, u32 off, u32 c)  called for EVERY index with its exclusive prefix
// Chunks are contiguous so output order = input order (deterministic, no atomics).

#ifndef CP_MAXCHUNKS
#define CP_MAXCHUNKS 2048
#endif

#define CP_VEC 4   // consecutive elements per thread per iteration (independent loads in flight)

template <class F>
__global__ __launch_bounds__(DEBWT_BLOCK) void cp_count_kernel(F f, u64 n, u64 chunk, u32 *__restrict__ counts) {
    __shared__ u32 red[DEBWT_WAVES];
    u64 beg = (u64)blockIdx.x * chunk;
    u64 end = beg + chunk < n ? beg + chunk : n;
    u32 local = 0;
    for (u64 i = beg + threadIdx.x; i < end; i += DEBWT_BLOCK * CP_VEC) {
#pragma unroll
        for (int v = 0; v < CP_VEC; v++)
            if (i + (u64)v * DEBWT_BLOCK < end) local += f.count(i + (u64)v * DEBWT_BLOCK);
    }
    u32 incl = wave_scan_incl(local);
    if (lane_id() == 63) red[threadIdx.x >> 6] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 s = 0;
        for (int w = 0; w < DEBWT_WAVES; w++) s += red[w];
        counts[blockIdx.x] = s;
    }
}

// two counters in one sweep: f.count2(i, &a, &b)
template <class F>
__global__ __launch_bounds__(DEBWT_BLOCK) void cp_count2_kernel(F f, u64 n, u64 chunk, u32 *__restrict__ counts_a,
                                                                 u32 *__restrict__ counts_b) {
    __shared__ u32 red[2][DEBWT_WAVES];
    u64 beg = (u64)blockIdx.x * chunk;
    u64 end = beg + chunk < n ? beg + chunk : n;
    u32 la = 0, lb = 0;
    // CP_VEC consecutive elements per thread: the classification stencil re-reads its neighbours
    for (u64 i = beg + (u64)threadIdx.x * CP_VEC; i < end; i += DEBWT_BLOCK * CP_VEC) {
#pragma unroll
        for (int v = 0; v < CP_VEC; v++) {
            if (i + v < end) {
                u32 a, b;
                f.count2(i + v, &a, &b);
                la += a; lb += b;
            }
        }
    }
    u32 ia = wave_scan_incl(la), ib = wave_scan_incl(lb);
    if (lane_id() == 63) { red[0][threadIdx.x >> 6] = ia; red[1][threadIdx.x >> 6] = ib; }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 sa = 0, sb = 0;
        for (int w = 0; w < DEBWT_WAVES; w++) { sa += red[0][w]; sb += red[1][w]; }
        counts_a[blockIdx.x] = sa; counts_b[blockIdx.x] = sb;
    }
}

// exclusive scan of counts[0..m) in place (m <= CP_MAXCHUNKS), total to *total; one workgroup
__global__ __launch_bounds__(1024) void cp_scan_kernel(u32 *__restrict__ counts, u32 m, u32 *__restrict__ total) {
    __shared__ u32 wsum[16];
    u32 per = (m + 1023u) / 1024u;
    u32 beg = threadIdx.x * per, end = beg + per < m ? beg + per : m;
    u32 s = 0;
    for (u32 i = beg; i < end; i++) s += counts[i];
    u32 incl = wave_scan_incl(s);
    u32 w = threadIdx.x >> 6;
    if (lane_id() == 63) wsum[w] = incl;
    __syncthreads();
    u32 base = 0, all = 0;
    for (u32 i = 0; i < 16; i++) { if (i < w) base += wsum[i]; all += wsum[i]; }
    u32 run = base + incl - s;
    for (u32 i = beg; i < end; i++) { u32 c = counts[i]; counts[i] = run; run += c; }
    if (threadIdx.x == 0) *total = all;
}

template <class F>
__global__ __launch_bounds__(DEBWT_BLOCK) void cp_emit_kernel(F f, u64 n, u64 chunk, const u32 *__restrict__ offsets) {
    __shared__ u32 tmp[CP_VEC * DEBWT_WAVES];
    u64 beg = (u64)blockIdx.x * chunk;
    u64 end = beg + chunk < n ? beg + chunk : n;
    u32 base = offsets[blockIdx.x];
    // element (v, tid) of a tile is tile + v*256 + tid: coalesced loads, CP_VEC of them in flight per thread
    for (u64 tile = beg; tile < end; tile += DEBWT_BLOCK * CP_VEC) {
        u32 c[CP_VEC], ex[CP_VEC], tot[CP_VEC];
#pragma unroll
        for (int v = 0; v < CP_VEC; v++) {
            u64 i = tile + (u64)v * DEBWT_BLOCK + threadIdx.x;
            c[v] = i < end ? f.recount(i) : 0;
        }
        block_scan_excl_vec<CP_VEC>(c, ex, tot, tmp);
#pragma unroll
        for (int v = 0; v < CP_VEC; v++) {
            u64 i = tile + (u64)v * DEBWT_BLOCK + threadIdx.x;
            if (i < end) f.emit(i, base + ex[v], c[v]);
            base += tot[v];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// text preparation

__global__ void k_set_sepbits(const u64 *__restrict__ sep, u64 nrec, u64 *__restrict__ bits) {
    u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nrec) atomicOr(&bits[sep[r] >> 6], 1ull << (sep[r] & 63));
}

// One key per position whose L-symbol window holds no separator, compacted by record:
//   mode 0 (L = K):  key = node << 2 | pred     node instances (pipeline)
//   mode 1 (L = k):  key = k-mer                edge instances (stand-alone k-mer count)
// Replaces the Jellyfish enumeration (src/kmercounting.sh:8) -- k-mers never span records.
__global__ __launch_bounds__(DEBWT_BLOCK) void k_extract_keys(const u64 *__restrict__ text,
                                                               const u64 *__restrict__ sepbits,
                                                               const u64 *__restrict__ sep, u64 nrec, u64 n, int L,
                                                               int mode, u64 *__restrict__ keys) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 sw = sep_window(sepbits, i);
    if (sw & ((1ull << L) - 1ull)) return;
    u64 rho = lower_bound_dev<u64>(sep, 0, nrec, i);          // separators before i
    u64 win = text_window(text, i) >> (64 - 2 * L);
    u64 key = win;
    if (mode == 0) {
        u32 pred = i ? text_symbol(text, i - 1) : 3u;          // 'T' stands at separators: fake pred 3
        key = (win << 2) | pred;
    }
    keys[i - rho * (u64)L] = key;
}

// k-mer prefix census for the shard splitters (the reference balances its sort threads on a cumulative
// 4^12-bin histogram the same way, src/mySort.c:98-110): node keys of positions [p0, p1) by their top 12 bits
#define SHARD_BINS 4096
__global__ __launch_bounds__(DEBWT_BLOCK) void k_prefix_hist(const u64 *__restrict__ text,
                                                              const u64 *__restrict__ sepbits, u64 p0, u64 p1, int K,
                                                              u64 *__restrict__ hist) {
    __shared__ u32 h[SHARD_BINS];
    for (u32 b = threadIdx.x; b < SHARD_BINS; b += DEBWT_BLOCK) h[b] = 0;
    __syncthreads();
    const u64 kmask = (1ull << K) - 1ull;
    for (u64 i = p0 + (u64)blockIdx.x * DEBWT_BLOCK + threadIdx.x; i < p1; i += (u64)gridDim.x * DEBWT_BLOCK) {
        if (sep_window(sepbits, i) & kmask) continue;
        u64 node = text_window(text, i) >> (64 - 2 * K);
        atomicAdd(&h[(u32)(node >> (2 * K - 12))], 1u);          // top 12 bits of the key = top 12 bits of the node
    }
    __syncthreads();
    for (u32 b = threadIdx.x; b < SHARD_BINS; b += DEBWT_BLOCK)
        if (h[b]) atomicAdd(&hist[b], (u64)h[b]);
}

// the same with a lane per text word (p0 a multiple of 32): 32 positions off two coalesced 8-byte loads
__global__ __launch_bounds__(DEBWT_BLOCK) void k_prefix_hist_words(const u64 *__restrict__ text,
                                                                    const u64 *__restrict__ sepbits, u64 p0, u64 p1, int K,
                                                                    u64 *__restrict__ hist) {
    __shared__ u32 h[SHARD_BINS];
    for (u32 b = threadIdx.x; b < SHARD_BINS; b += DEBWT_BLOCK) h[b] = 0;
    __syncthreads();
    const u64 g1 = (p1 + 31) >> 5;
    for (u64 g = (p0 >> 5) + (u64)blockIdx.x * DEBWT_BLOCK + threadIdx.x; g < g1; g += (u64)gridDim.x * DEBWT_BLOCK) {
        const u64 w0 = text[g], w1 = text[g + 1], i0 = g << 5;
        const u64 sb = sep_window(sepbits, i0);
        const u32 lim = p1 - i0 < 32 ? (u32)(p1 - i0) : 32u;
        u32 ok = lim < 32 ? (1u << lim) - 1u : 0xFFFFFFFFu;
        if (wave_any(sb != 0ull)) ok &= ~sep_blocked(sb, K);
#pragma unroll
        for (u32 t = 0; t < 32; t++) {
            const u32 pre = t <= 26 ? (u32)(w0 >> (52 - 2 * t)) & 0xFFFu : (u32)(((w0 << (2 * t)) | (w1 >> (64 - 2 * t))) >> 52);
            if ((ok >> t) & 1u) atomicAdd(&h[pre], 1u);
        }
    }
    __syncthreads();
    for (u32 b = threadIdx.x; b < SHARD_BINS; b += DEBWT_BLOCK)
        if (h[b]) atomicAdd(&hist[b], (u64)h[b]);
}

// ---------------------------------------------------------------------------------------------------
// run-length encoding of the sorted keys (kmerInfo analogue, src/mySort.c:193-195) + case-2 symbols

struct RleF {
    const u64 *sk; u64 *dk; u32 *dstart; u8 *mchar;
    __device__ u32 count(u64 j) const {
        u64 k = sk[j];
        if (mchar) mchar[j] = (u8)(k & 3);                     // bwtSingle by row (src/INandOut.c:367-395)
        return (j == 0 || sk[j - 1] != k) ? 1u : 0u;
    }
    __device__ u32 recount(u64 j) const { return (j == 0 || sk[j - 1] != sk[j]) ? 1u : 0u; }
    __device__ void emit(u64 j, u32 off, u32 c) const {
        if (c) { dk[off] = sk[j]; dstart[off] = (u32)j; }
    }
};

// ---------------------------------------------------------------------------------------------------
// node classification over the distinct keys (mergeKmer, src/INandOut.c:258-346)

struct ClassifyCommon {
    const u64 *dk; const u32 *dstart; u64 D; u64 M;
    const u64 *head_keys; u64 nrec;
    // 32-Kbit bitmap (LDS) of the hashed record-start keys, or nullptr: a quarter of all keys carry pred 3, the symbol that
    // also stands before a record start, and every one of them asked the sorted list by bisection -- eight dependent
    // loads at the 240 records of a genome collection; with the bitmap only a key whose bit is set (0.7 % there) does
    const u32 *hbm = nullptr;
    __device__ static u32 head_hash(u64 k) { return (u32)((k * 0x9E3779B97F4A7C15ull) >> 49); }
    __device__ bool maybe_head(u64 k) const {                  // k = node << 2 | 3
        if (!hbm) return true;
        const u32 h = head_hash(k);
        return (hbm[h >> 5] >> (h & 31u)) & 1u;
    }
    // fills the bitmap (1024 words of LDS; all threads of the workgroup; ends with a barrier); false: too many records
    __device__ bool head_bitmap(u32 *bm) {
        if (nrec > 8192) { __syncthreads(); return false; }
        for (u32 i = threadIdx.x; i < 1024; i += blockDim.x) bm[i] = 0;
        __syncthreads();
        for (u32 i = threadIdx.x; i < (u32)nrec; i += blockDim.x) {
            const u32 h = head_hash(head_keys[i]);
            atomicOr(&bm[h >> 5], 1u << (h & 31u));
        }
        __syncthreads();
        hbm = bm;
        return true;
    }
    // does distinct key e have an instance that is a real edge?  (record-start instances carry the fake pred 3:
    // only a key that equals a record-start key needs its instance count)
    __device__ bool real_count(u64 e) const {
        u64 k = dk[e];
        if ((k & 3) != 3 || !maybe_head(k)) return true;
        u64 lo = lower_bound_dev<u64>(head_keys, 0, nrec, k);
        if (lo >= nrec || head_keys[lo] != k) return true;
        u64 hi = upper_bound_dev<u64>(head_keys, lo, nrec, k);
        u32 cnt = (e + 1 < D ? dstart[e + 1] : (u32)M) - dstart[e];
        return cnt > (u32)(hi - lo);
    }
    __device__ bool is_head(u64 node) const {
        u64 k = (node << 2) | 3ull;
        if (!maybe_head(k)) return false;
        u64 lo = lower_bound_dev<u64>(head_keys, 0, nrec, k);
        return lo < nrec && head_keys[lo] == k;
    }
};

// multi-in nodes: >= 2 distinct real predecessors or a record-start occurrence (src/INandOut.c:282-343)
__device__ __forceinline__ bool eval_multi_in(const ClassifyCommon &c, u64 e, u32 *freq) {
    u64 node = c.dk[e] >> 2;
    if (e && (c.dk[e - 1] >> 2) == node) return false;
    u32 preds = 0;
    u64 f = e;
    for (; f < c.D && f < e + 4 && (c.dk[f] >> 2) == node; f++)
        if (c.real_count(f)) preds |= 1u << (c.dk[f] & 3);
    if (freq) *freq = (f < c.D ? c.dstart[f] : (u32)c.M) - c.dstart[e];
    return __popc(preds) >= 2 || c.is_head(node);
}

// multi-out facts: within the group of keys sharing the (K-1)-symbol node prefix W, every pred c with
// >= 2 distinct last symbols d among real keys (W.d, c) makes node c.W multi-out
// (out-degree > 1, src/INandOut.c:271-281, read off the same sorted edge list)
__device__ __forceinline__ u32 eval_multi_out(const ClassifyCommon &c, int K, u64 e, u64 *facts) {
    u64 W = c.dk[e] >> 4;
    if (e && (c.dk[e - 1] >> 4) == W) return 0;
    u32 succ[4] = {0, 0, 0, 0};
    for (u64 f = e; f < c.D && f < e + 16 && (c.dk[f] >> 4) == W; f++) {
        if (!c.real_count(f)) continue;
        u64 k = c.dk[f];
        succ[k & 3] |= 1u << ((k >> 2) & 3);
    }
    u32 m = 0;
    for (u32 p = 0; p < 4; p++)
        if (__popc(succ[p]) >= 2) {
            if (facts) facts[m] = ((((u64)p << (2 * (K - 1))) | W) << 2) | 1ull;
            m++;
        }
    return m;
}

// one sweep over the distinct keys: cf[e] = multi-in node start | (multi-out facts of the W-group) << 1
struct ClassifyFlagsF {
    ClassifyCommon c;
    int K;
    u8 *cf;
};
// The same sweep as a kernel of its own: a thread takes 4 consecutive distinct keys with two 16-byte loads, the
// keys before and behind come from the neighbouring lanes; a key whose (K-1)-prefix group is just itself (the bulk:
// every unique k-mer) is classified in registers, only the others walk their group in memory.  One 4-byte store
// of the four classification bytes.  Chunks as plan_chunks (multiples of 1024 keys).
// Keys inside larger groups (about 1 % of them, but some in nearly every wave) are not walked here -- a few lanes
// chasing dependent loads would stall every wave -- but queued (wave-aggregated append, order irrelevant) for
// k_classify_groups, where every lane has such a key.  The queue of a chunk is wl[beg ..), wl_count[chunk] entries.
__global__ __launch_bounds__(DEBWT_BLOCK) void k_classify_flags(ClassifyFlagsF f, u64 chunk, u32 *__restrict__ counts_a,
                                                                 u32 *__restrict__ counts_b, u32 *__restrict__ wl,
                                                                 u32 *__restrict__ wl_count) {
    __shared__ u32 red[2][DEBWT_WAVES];
    __shared__ u32 qn;
    __shared__ u32 hbm[1024];                                  // record-start keys, hashed (ClassifyCommon::head_bitmap)
    ClassifyCommon c = f.c;
    const u64 D = c.D;
    u64 beg = (u64)blockIdx.x * chunk;
    u64 end = beg + chunk < D ? beg + chunk : D;
    u32 la = 0, lb = 0;
    if (threadIdx.x == 0) qn = 0;
    c.head_bitmap(hbm);
    wl += beg;
    for (u64 tile = beg; tile < end; tile += DEBWT_BLOCK * 4) {
        const u64 e0 = tile + (u64)threadIdx.x * 4;
        u64 k[6];                                              // k[0] = key before e0, k[1..4] = e0..e0+3, k[5] = key behind
        const bool whole = e0 + 4 <= end;                      // end <= D and D's buffer is padded: loads stay inside
        if (whole) {
            ulonglong2 v0 = *reinterpret_cast<const ulonglong2 *>(c.dk + e0);
            ulonglong2 v1 = *reinterpret_cast<const ulonglong2 *>(c.dk + e0 + 2);
            k[1] = v0.x; k[2] = v0.y; k[3] = v1.x; k[4] = v1.y;
        } else {
#pragma unroll
            for (int t = 0; t < 4; t++) k[1 + t] = e0 + t < D ? c.dk[e0 + t] : ~0ull;
        }
        u64 up = __shfl_up(k[4], 1, 64), dn = __shfl_down(k[1], 1, 64);
        const u32 lane = threadIdx.x & 63u;
        // the wave's first/last lane (and lanes next to a partial thread) read their neighbours from memory
        k[0] = (lane == 0 || !whole) ? (e0 && e0 <= D ? c.dk[e0 - 1] : ~0ull) : up;
        k[5] = (lane == 63 || !whole || e0 + 8 > end) ? (e0 + 4 < D ? c.dk[e0 + 4] : ~0ull) : dn;
        u32 word = 0, queued = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const u64 e = e0 + t;
            if (e >= end) break;
            const u64 key = k[1 + t], W = key >> 4;
            const bool alone = (e == 0 || (k[t] >> 4) != W) && (e + 1 >= D || (k[2 + t] >> 4) != W);
            if (alone) {
                u32 mi = ((key & 3) == 3 && c.is_head(key >> 2)) ? 1u : 0u;
                word |= mi << (8 * t);
                la += mi;
            } else {
                queued |= 1u << t;
            }
        }
        {   // append the queued keys of the wave: one atomic per wave and tile
            u32 nq = __popc(queued);
            u32 incl = wave_scan_incl(nq);
            u32 base = 0;
            if (lane == 63 && incl) base = atomicAdd(&qn, incl);
            base = __shfl(base, 63, 64) + incl - nq;
#pragma unroll
            for (int t = 0; t < 4; t++)
                if ((queued >> t) & 1u) wl[base++] = (u32)(e0 + t);
        }
        if (whole) *reinterpret_cast<u32 *>(f.cf + e0) = word;
        else {
#pragma unroll
            for (int t = 0; t < 4; t++) if (e0 + t < end) f.cf[e0 + t] = (u8)(word >> (8 * t));
        }
    }
    u32 ia = wave_scan_incl(la), ib = wave_scan_incl(lb);
    if (lane_id() == 63) { red[0][threadIdx.x >> 6] = ia; red[1][threadIdx.x >> 6] = ib; }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 sa = 0, sb = 0;
        for (int w = 0; w < DEBWT_WAVES; w++) { sa += red[0][w]; sb += red[1][w]; }
        counts_a[blockIdx.x] = sa; counts_b[blockIdx.x] = sb;
        wl_count[blockIdx.x] = qn;
    }
}
// ordered compaction of both fact kinds from the classification bytes: every thread owns 16 consecutive
// bytes (one 16-byte load); one block scan per 4096 distinct keys carries both running offsets
struct FactEmitArgs {
    ClassifyCommon c;
    int K;
    const u8 *cf;
    u64 *mi_fact; u32 *mi_j0; u32 *mi_freq;
    u64 *mo_fact;
    uint4 *work;              // one slot per fact, the item sits in the slot of its first fact, others stay 0
};
__global__ __launch_bounds__(DEBWT_BLOCK) void k_eval_facts(FactEmitArgs a, u64 nslots) {
    __shared__ u32 hbm[1024];
    a.c.head_bitmap(hbm);                                      // (every thread of the workgroup: barriers inside)
    u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslots) return;
    uint4 it = a.work[s];
    if (it.w == 0) return;
    u64 e = it.x;
    if (it.w & 1u) {
        u32 fr;
        eval_multi_in(a.c, e, &fr);
        a.mi_fact[it.y] = ((a.c.dk[e] >> 2) << 2) | 2ull;
        a.mi_j0[it.y] = a.c.dstart[e];
        a.mi_freq[it.y] = fr;
    }
    u32 cnt = it.w >> 1;
    if (cnt) {
        u64 facts[4];
        eval_multi_out(a.c, a.K, e, facts);
#pragma unroll
        for (u32 m = 0; m < 4; m++)
            if (m < cnt) a.mo_fact[it.z + m] = facts[m];
    }
}
__global__ __launch_bounds__(DEBWT_BLOCK) void k_emit_facts(FactEmitArgs a, u64 chunk, const u32 *__restrict__ off_mi,
                                                             const u32 *__restrict__ off_mo) {
    __shared__ u32 tmp[2 * DEBWT_WAVES];
    const u64 D = a.c.D;
    u64 beg = (u64)blockIdx.x * chunk;
    u64 end = beg + chunk < D ? beg + chunk : D;
    u32 base_mi = off_mi[blockIdx.x], base_mo = off_mo[blockIdx.x];
    for (u64 tile = beg; tile < end; tile += DEBWT_BLOCK * 16) {
        u64 e0 = tile + (u64)threadIdx.x * 16;
        u32 w[4] = {0, 0, 0, 0};
        if (e0 + 16 <= end) {
            uint4 v = *reinterpret_cast<const uint4 *>(a.cf + e0);      // chunk and tile are multiples of 16
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
#pragma unroll
            for (u32 t = 0; t < 16; t++)
                if (e0 + t < end) w[t >> 2] |= (u32)a.cf[e0 + t] << ((t & 3) * 8);
        }
        u32 val[2] = {0, 0}, ex[2], tot[2];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            val[0] += __popc(w[q] & 0x01010101u);
            u32 m = (w[q] >> 1) & 0x07070707u;
            val[1] += (m & 0xFF) + ((m >> 8) & 0xFF) + ((m >> 16) & 0xFF) + (m >> 24);
        }
        block_scan_excl_vec<2>(val, ex, tot, tmp);
        if (val[0] | val[1]) {
            // work items only: the look-ahead evaluation runs one item per thread in k_eval_facts
            u32 omi = base_mi + ex[0], omo = base_mo + ex[1];
#pragma unroll
            for (u32 t = 0; t < 16; t++) {                     // unrolled: w[] must stay in registers
                u32 f = (w[t >> 2] >> ((t & 3) * 8)) & 0xFFu;
                if (f) {
                    a.work[omi + omo] = make_uint4((u32)(e0 + t), omi, omo, f);
                    omi += f & 1u;
                    omo += f >> 1;
                }
            }
        }
        base_mi += tot[0]; base_mo += tot[1];
    }
}

// the queued keys of a chunk, one per thread; classification bytes, and the chunk's counters of the sweep above
// grow by what they add
__global__ __launch_bounds__(DEBWT_BLOCK) void k_classify_groups(ClassifyFlagsF f, u64 chunk, u32 *__restrict__ counts_a,
                                                                  u32 *__restrict__ counts_b, const u32 *__restrict__ wl,
                                                                  const u32 *__restrict__ wl_count) {
    __shared__ u32 red[2][DEBWT_WAVES];
    __shared__ u32 hbm[1024];
    const u32 nq = wl_count[blockIdx.x];
    if (nq == 0) return;
    ClassifyCommon c = f.c;
    c.head_bitmap(hbm);                                        // (real_count and is_head ask the record-start keys: see there)
    wl += (u64)blockIdx.x * chunk;
    u32 la = 0, lb = 0;
    for (u32 i = threadIdx.x; i < nq; i += DEBWT_BLOCK) {
        const u64 e = wl[i];
        const u32 mi = eval_multi_in(c, e, nullptr) ? 1u : 0u;
        const u32 mo = eval_multi_out(c, f.K, e, nullptr);
        if (mi | mo) f.cf[e] = (u8)(mi | (mo << 1));
        la += mi; lb += mo;
    }
    u32 ia = wave_scan_incl(la), ib = wave_scan_incl(lb);
    if (lane_id() == 63) { red[0][threadIdx.x >> 6] = ia; red[1][threadIdx.x >> 6] = ib; }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 sa = 0, sb = 0;
        for (int w = 0; w < DEBWT_WAVES; w++) { sa += red[0][w]; sb += red[1][w]; }
        counts_a[blockIdx.x] += sa; counts_b[blockIdx.x] += sb;
    }
}

// red table: sorted fact list -> one entry per node, node<<2 | multiin<<1 | multiout
// (redSeq analogue, src/INandOut.c:396-412).  Facts of one node sort as (X|1)...(X|1)(X|2).
struct RedUniqueF {
    const u64 *facts; u64 nf; u64 *red;
    __device__ u32 count(u64 e) const { return (e + 1 == nf || (facts[e + 1] >> 2) != (facts[e] >> 2)) ? 1u : 0u; }
    __device__ u32 recount(u64 e) const { return count(e); }
    __device__ void emit(u64 e, u32 off, u32 c) const {
        if (!c) return;
        u64 f = facts[e];
        u32 fl = (u32)(f & 3);
        if (fl == 2 && e && (facts[e - 1] >> 2) == (f >> 2)) fl = 3;
        red[off] = ((f >> 2) << 2) | fl;
    }
};
// rank of every multi-in red entry among the multi-in entries = its block id
struct RedBlockF {
    const u64 *red; u32 *red_q;
    __device__ u32 count(u64 r) const { return (u32)(red[r] >> 1) & 1u; }
    __device__ u32 recount(u64 r) const { return count(r); }
    __device__ void emit(u64 r, u32 off, u32 c) const { red_q[r] = c ? off : 0xFFFFFFFFu; }
};
// exclusive scan of block sizes -> first blue slot of each block (blueBound analogue)
struct BlockStartF {
    const u32 *mi_freq; u32 *bstart;
    __device__ u32 count(u64 q) const { return mi_freq[q]; }
    __device__ u32 recount(u64 q) const { return count(q); }
    __device__ void emit(u64 q, u32 off, u32) const { bstart[q] = off; }
};
struct LargeBlockF {
    const u32 *mi_freq; u32 cap; u32 *large_q;
    __device__ u32 count(u64 q) const { return mi_freq[q] > cap ? 1u : 0u; }
    __device__ u32 recount(u64 q) const { return count(q); }
    __device__ void emit(u64 q, u32 off, u32 c) const { if (c) large_q[off] = (u32)q; }
};

// block tables of one key range (local instance / blue indices) -> the context-wide tables with 64-bit offsets
__global__ void k_append_blocks(const u32 *__restrict__ mi_j0, const u32 *__restrict__ mi_freq,
                                const u32 *__restrict__ bstart, u64 Q, u64 mbase, u64 bbase,
                                u64 *__restrict__ blk_j0, u32 *__restrict__ blk_freq, u64 *__restrict__ blk_start) {
    u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    blk_j0[q] = mbase + mi_j0[q];
    blk_freq[q] = mi_freq[q];
    blk_start[q] = bbase + bstart[q];
}
// blocks of a range with lo < rows <= hi (sizes the launch plan of the blue sort)
__global__ void k_count_blocks(const u32 *__restrict__ mi_freq, u64 Q, u32 lo, u32 hi, u32 *__restrict__ counter) {
    // grid-stride: one atomic per wave of a few hundred workgroups, not one per 64 blocks (80,000 on one word took 0.9 ms)
    u32 n = 0;
    for (u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x; q < Q; q += (u64)gridDim.x * blockDim.x) {
        const u32 f = mi_freq[q];
        n += (f > lo && f <= hi) ? 1u : 0u;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor(n, d, 64);
    if (n && (threadIdx.x & 63u) == 0) atomicAdd(counter, n);
}
// large blocks of a range: local block ids -> context-wide ids
__global__ void k_offset_u32(u32 *__restrict__ v, u64 n, u32 add) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] += add;
}

// Node lookup for the text scan (the job of blackTable + BinarySearch_red, src/generateSP.c:53-59,542-566,
// 725-737): an open-addressing table over the red nodes (load <= 1/2, linear probing so a probe sequence stays
// inside one 128-byte line) behind a one-bit-per-bin prefilter that fits L2.  A slot carries everything the blue
// fill needs -- key, the block's fill cursor, its global block id -- so one line serves probe and cursor.
// Both tables are indexed by multiplicative hashes of the node, so neighbouring text positions spread evenly.
struct __attribute__((aligned(16))) HSlot {
    u64 key;    // node<<2 | multiin<<1 | multiout; 0 = empty
    u32 cur;    // fill cursor of the node's block (see k_build_hash), HCURSOR_SKIP when another shard owns it
    u32 q;      // global block id
};
#define HCURSOR_SKIP 0xFFFFFFFFu
__device__ __forceinline__ u32 hmask(int hbits) { return hbits >= 32 ? 0xFFFFFFFFu : (1u << hbits) - 1u; }   // up to 2^32 slots (64 GB)
__device__ __forceinline__ u32 red_hash(u64 node, int bits) { return (u32)((node * 0x9E3779B97F4A7C15ull) >> (64 - bits)); }
__device__ __forceinline__ u32 red_hash2(u64 node, int bits) { return (u32)((node * 0xC2B2AE3D27D4EB4Full) >> (64 - bits)); }

// Node table addressed by minimizer (mzK > 0; cfg.reserved bit 14, NOT the default: measured at 30 Gbp it fetches 24 % fewer
// lines per slice -- 561 M read requests against 735 M -- but the SP flags pass takes 434 ms per build against 424 ms,
// DESIGN.md section 8).  Consecutive text positions
// hold nodes that overlap in all but one symbol and mostly share their minimizer, and in repeat families nearly every
// position is a branching node: with the table hashed by node every one of them is a random 128-byte line (PMC at
// 30 Gbp: 40 bytes fetched per text position).  Addressed by MINIMIZER, the nodes of neighbouring positions are
// neighbours in the table: a minimizer owns a window of 32 slots (512 bytes), and a node sits in the pair of slots
// that belongs to the OFFSET of the minimizer inside the node (0 .. K-16: 16 pairs) -- the next text position has the
// same minimizer one symbol further left, i.e. the pair before.  One 32-byte load per candidate, no search, and a
// stretch of positions walks its window line by line.  A node whose pair is taken (several branching nodes with the same
// minimizer at the same offset, windows of different minimizers that collide) goes to its node hash with ordinary
// linear probing; slots are never freed, so a lookup that finds an empty slot in the pair knows the node is in neither
// place.
#define MZ_W 16
__device__ __forceinline__ u32 mz_hash(u32 x) { x *= 0x9E3779B1u; x ^= x >> 15; return x * 0x85EBCA6Bu; }
__device__ __forceinline__ u32 mz_word(u32 m, int fbits) { return (m * 0xC2B2AE3Du) >> (32 - fbits); }
__device__ __forceinline__ u32 mz_bit(u64 node) { return (u32)((node * 0xC2B2AE3D27D4EB4Full) >> 58); }
// first slot of the pair of (minimizer m at offset o of the node)
__device__ __forceinline__ u32 mz_pair(u32 m, u32 o, int hbits) { return (((m * 0x27D4EB2Fu) >> (32 - (hbits - 5))) << 5) + 2u * o; }
// minimizer of a node and its offset (the first 16-mer with the smallest hash)
// mzw: symbols of a minimizer -- MZ_W for nodes of 24 symbols and more, MZ_W_SHORT (12) for nodes of 16..23 (k = 17..24: nine
// to twelve candidate 12-mers per node; round 6 -- before, such k probed a bitmap once per position: k = 20 on 3.1 Gbp spent
// 82 ms in the SP stage against 28 at k = 32)
#define MZ_W_SHORT 12
__device__ __forceinline__ u32 mz_of_node(u64 node, int K, u32 *off, int mzw = MZ_W) {
    const u64 win = node << (64 - 2 * K);
    const int nw = K - mzw + 1, hs = 32 - 2 * mzw;
    u32 m = 0xFFFFFFFFu, o = 0;
    for (int j = 0; j < nw; j++) {
        const u32 h = mz_hash((u32)(win >> (32 - 2 * j)) >> hs);
        if (h < m) { m = h; o = (u32)j; }
    }
    *off = o;
    return m;
}

__global__ void k_build_hash(const u64 *__restrict__ red, u64 R, const u32 *__restrict__ red_q,
                             const u64 *__restrict__ blk_start, int abs32, u32 qbase, u32 Qlocal, int hbits,
                             HSlot *__restrict__ htab, int pb, u32 *__restrict__ rbits, int mzK) {
    u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const u64 v = red[r], node = v >> 2;
    const u32 mask = hmask(hbits);
    u32 h = 0;
    bool placed = false;
    if (mzK) {
        u32 o;
        const u32 m = mz_of_node(node, mzK, &o);
        h = mz_pair(m, o, hbits);
        placed = atomicCAS(&htab[h].key, 0ull, v) == 0ull;
        if (!placed) placed = atomicCAS(&htab[++h].key, 0ull, v) == 0ull;
    }
    if (!placed) {
        h = red_hash(node, hbits);
        for (;;) {
            u64 old = atomicCAS(&htab[h].key, 0ull, v);
            if (old == 0ull) break;
            h = (h + 1) & mask;
        }
    }
    // fill cursor of a multi-in node (redPoint analogue, src/INandOut.c:413): the next free blue slot of its block
    // when all blue slots of the context fit 32 bits (abs32), else the entries placed so far (the block's first
    // slot is then added from blk_start[q], a 64-bit offset)
    if (v & 2ull) {
        u32 q = red_q[r] - qbase;                                // wraps for blocks before this shard
        htab[h].cur = q < Qlocal ? (abs32 ? (u32)blk_start[q] : 0u) : HCURSOR_SKIP;
        htab[h].q = red_q[r];                                    // global block id (blue-entry exchange)
    }
    if (pb) {                                                    // pb = 0: the minimizer filter is built by k_build_mzfilter
        u32 hb = red_hash2(node, pb);
        atomicOr(&rbits[hb >> 5], 1u << (hb & 31));
    }
}

// Prefilter with locality.  A probe per text position into a bitmap the size of the Infinity Cache is what the SP stage
// of a 30 Gbp build spends its time on: 3 * 10^10 random 64-byte sectors for one bit each.  Consecutive positions hold
// nodes that overlap in all but one symbol, and so mostly share their MINIMIZER (the 16-mer of the node with the smallest
// hash): the filter word of a node is chosen by its minimizer, the bit inside the word by the node itself.  A lane that
// walks 32 consecutive positions then changes words only ~4 times and fetches 4 sectors instead of 32.  Nodes that share
// a minimizer share the word (red nodes cluster around the same loci), the words are sized for ~2 red nodes each; a
// minimizer that very many red nodes share (a homopolymer's) saturates its word and merely sends its positions to the
// node table, as every position went before.  Used for K >= 24 (a node then has >= 9 candidate 16-mers).
__global__ void k_build_mzfilter(const u64 *__restrict__ red, u64 R, int K, int fbits, u64 *__restrict__ fw, int mzw) {
    u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const u64 node = red[r] >> 2;
    u32 o;
    atomicOr(&fw[mz_word(mz_of_node(node, K, &o, mzw), fbits)], 1ull << mz_bit(node));
}

// returns the slot (or 0xFFFFFFFF) and the flags of `node`; mzK > 0: the table is addressed by minimizer (k_build_hash)
__device__ __forceinline__ u32 red_lookup(const HSlot *__restrict__ htab, int hbits, u64 node, u32 *flags, int mzK = 0) {
    const u32 mask = hmask(hbits);
    if (mzK) {
        u32 o;
        const u32 m = mz_of_node(node, mzK, &o);
        const u32 a = mz_pair(m, o, hbits);
        for (u32 h = a; h < a + 2u; h++) {
            const u64 v = htab[h].key;
            if (v == 0ull) { *flags = 0; return 0xFFFFFFFFu; }
            if ((v >> 2) == node) { *flags = (u32)(v & 3); return h; }
        }
    }
    u32 h = red_hash(node, hbits);
    for (;;) {
        u64 v = htab[h].key;
        if (v == 0ull) { *flags = 0; return 0xFFFFFFFFu; }
        if ((v >> 2) == node) { *flags = (u32)(v & 3); return h; }
        h = (h + 1) & mask;
    }
}

// the same with the slot read as one 16-byte word: flags and the block id of `node` (q is meaningful for a multi-in node)
__device__ __forceinline__ u32 red_lookup_q(const HSlot *__restrict__ htab, int hbits, u64 node, u32 *q) {
    const u32 mask = hmask(hbits);
    u32 h = red_hash(node, hbits);
    for (;;) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(&htab[h]);
        if (v.x == 0ull) return 0u;
        if ((v.x >> 2) == node) { *q = (u32)(v.y >> 32); return (u32)(v.x & 3); }
        h = (h + 1) & mask;
    }
}
// minimizer-addressed table: a = first slot of the node's pair (mz_pair)
__device__ __forceinline__ u32 red_lookup_q_mz(const HSlot *__restrict__ htab, int hbits, u64 node, u32 a, u32 *q) {
    const ulonglong2 v0 = *reinterpret_cast<const ulonglong2 *>(&htab[a]);
    const ulonglong2 v1 = *reinterpret_cast<const ulonglong2 *>(&htab[a + 1u]);
    if ((v0.x >> 2) == node && v0.x) { *q = (u32)(v0.y >> 32); return (u32)(v0.x & 3); }
    if ((v1.x >> 2) == node && v1.x) { *q = (u32)(v1.y >> 32); return (u32)(v1.x & 3); }
    if (v0.x == 0ull || v1.x == 0ull) return 0u;
    return red_lookup_q(htab, hbits, node, q);
}

// rows of the special suffixes: rank among the node instances + own rank (src/INandOut.c:419-439)
// (number of instances with key <= X) = first row of the first distinct key above X
__global__ void k_special_rows(const u64 *__restrict__ dk, const u32 *__restrict__ dstart, u64 D, u64 M,
                               const u64 *__restrict__ spkey, u64 NS, u64 row_base, u64 *__restrict__ sprow) {
    u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= NS) return;
    const u64 e = upper_bound_dev<u64>(dk, 0, D, (spkey[s] << 2) | 3ull);
    sprow[s] = row_base + s + (e < D ? (u64)dstart[e] : M);
}

// ---------------------------------------------------------------------------------------------------
// SP code + blue entries (multiGenerateSP, src/generateSP.c:534-683)

// block ids of the multi-in positions, handed from pass 1 to pass 2 (list == nullptr: not kept)
struct SpBlockIds {
    u32 *list;                 // the ids, one run per wave of pass 1
    unsigned long long *count; // bump counter of `list`
    u64 *wave_base;            // first id of the wave that starts at group g0 + 64 * index
    u64 g0;
};

// pass 1: one lane = 32 consecutive positions = one text word (coalesced 8-byte loads); per position the
// node is a shift of the 128-bit (w0,w1) pair; flags go out as two 32-bit masks per group
#ifndef SP_FLAGS_WAVES
#define SP_FLAGS_WAVES 4
#endif
#ifndef SP_EMIT_STAGE
#define SP_EMIT_STAGE 1                // k_sp_emit: SP symbols of a tile through LDS and out as aligned words (0: a byte store per symbol)
#endif
#ifndef SP_PROBE_BATCH
#define SP_PROBE_BATCH 2               // home slots of the node table requested per lane before the first is used (measured at 30 Gbp,
                                       // SP stage: 1 -> 424.4 ms, 2 -> 405.0, 4 -> 412.9, 8 -> 410.8; profiles/r05_experiments.txt)
#endif
template <int MZ>
__global__ __launch_bounds__(DEBWT_BLOCK) __attribute__((amdgpu_waves_per_eu(SP_FLAGS_WAVES))) void k_sp_flags(const u64 *__restrict__ text, const u64 *__restrict__ sepbits,
                                                           u64 n, int K, const HSlot *__restrict__ htab, int hbits,
                                                           const u32 *__restrict__ rbits, int pb,
                                                           const u64 *__restrict__ branch, u64 nbranch,
                                                           u32 *__restrict__ momask, u32 *__restrict__ mimask,
                                                           u64 g0, u64 g1, SpBlockIds ids, const u64 *__restrict__ brbits, int mzw = MZ_W) {
    // block ids of the lane's multi-in positions until the wave knows where they go (a lane reads only its own 32 words)
    __shared__ u32 lq[DEBWT_BLOCK * 32];
    u64 g = g0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = g < g1;
    if (!ids.list && !active) return;
    u32 mo = 0, mi = 0;
    if (active) {
    const u64 w0 = text[g], w1 = text[g + 1];
    const u64 sb = sep_window(sepbits, g << 5);
    const u64 i0 = g << 5;
    u32 lim = (n - i0) < 32 ? (u32)(n - i0) : 32u;
    const u32 inrange = lim == 32 ? 0xFFFFFFFFu : ((1u << lim) - 1u);
    // phase 1: which positions hold a node that may be in the red table
    u32 cand = 0, spec = 0;
    if (MZ) {
        // hashes of the 16-mers that start at symbols 0..46 of the word pair; minimizer of the node at t = the smallest
        // among those of symbols t .. t + K - 16 (pb = filter words as a power of two, rbits = the words)
        const u64 *fw = reinterpret_cast<const u64 *>(rbits);
        const int nw = K - mzw + 1, hs = 32 - 2 * mzw;      // (minimizers of mzw symbols: the leading ones of the 16-symbol window)
        u32 H[47];
#pragma unroll
        for (int j = 0; j < 47; j++) {
            u32 x;
            if (j <= 16) x = (u32)(w0 >> (32 - 2 * j));
            else if (j < 32) x = (u32)(w0 << (2 * j - 32)) | (u32)(w1 >> (96 - 2 * j));
            else x = (u32)(w1 >> (96 - 2 * j));
            H[j] = mz_hash(x >> hs);
        }
        u32 mprev = 0;
        u64 fwv = 0;
#pragma unroll
        for (u32 t = 0; t < 32; t++) {
            u32 m = H[t], o = 0;
#pragma unroll
            for (int d = 1; d < 16; d++)
                if (d < nw) {
                    const bool lt = H[t + d] < m;
                    m = lt ? H[t + d] : m;
                    if (MZ == 2) o = lt ? (u32)d : o;
                }
            if (t == 0 || m != mprev) fwv = fw[mz_word(m, pb)];
            mprev = m;
            const u64 win = t ? ((w0 << (2 * t)) | (w1 >> (64 - 2 * t))) : w0;
            const u32 bit = (u32)((fwv >> mz_bit(win >> (64 - 2 * K))) & 1ull);
            // the node table is addressed by (minimizer, its offset in the node) too: the candidate's pair of slots
            if (MZ == 2 && bit) lq[threadIdx.x * 32 + t] = mz_pair(m, o, hbits);
            cand |= bit << t;
        }
    } else {
#pragma unroll
    for (u32 t = 0; t < 32; t++) {
        u64 win = t ? ((w0 << (2 * t)) | (w1 >> (64 - 2 * t))) : w0;
        u32 hb = red_hash2(win >> (64 - 2 * K), pb);
        u32 bit = (rbits[hb >> 5] >> (hb & 31)) & 1u;
        cand |= bit << t;
    }
    }
    if (wave_any(sb != 0ull)) spec = sep_blocked(sb, K);         // a separator within 64 positions: rare, a uniform branch
    cand &= inrange & ~spec;
    spec &= inrange;
    // phase 2: only the candidates pay for the table search
    if (MZ == 2) {
        // minimizer-addressed table (cfg.reserved bit 14): SP_BATCH candidates are addressed and their pairs of slots
        // requested before the first answer is used (their lines are mostly in L2: what is left is the round trip)
        constexpr int SP_BATCH = 4;
        while (cand) {
            u32 tt[SP_BATCH], ad[SP_BATCH];
            u64 nd[SP_BATCH];
            ulonglong2 v0[SP_BATCH], v1[SP_BATCH];
            bool have[SP_BATCH];
#pragma unroll
            for (int b = 0; b < SP_BATCH; b++) {
                have[b] = cand != 0u;
                const u32 t = have[b] ? (u32)__ffs(cand) - 1u : 0u;
                cand &= cand - 1u;                              // 0 stays 0
                tt[b] = t;
                nd[b] = (t ? ((w0 << (2 * t)) | (w1 >> (64 - 2 * t))) : w0) >> (64 - 2 * K);
                ad[b] = lq[threadIdx.x * 32 + t];
            }
#pragma unroll
            for (int b = 0; b < SP_BATCH; b++) {
                v0[b] = v1[b] = make_ulonglong2(0ull, 0ull);
                if (have[b]) {
                    v0[b] = *reinterpret_cast<const ulonglong2 *>(&htab[ad[b]]);
                    v1[b] = *reinterpret_cast<const ulonglong2 *>(&htab[ad[b] + 1u]);
                }
            }
#pragma unroll
            for (int b = 0; b < SP_BATCH; b++) {
                if (!have[b]) continue;
                u32 fl = 0, q = 0;
                // the pair of (minimizer, offset): a match, an empty slot (the node is in neither place), or on to the node hash
                if (v0[b].x && (v0[b].x >> 2) == nd[b]) { fl = (u32)(v0[b].x & 3ull); q = (u32)(v0[b].y >> 32); }
                else if (v1[b].x && (v1[b].x >> 2) == nd[b]) { fl = (u32)(v1[b].x & 3ull); q = (u32)(v1[b].y >> 32); }
                else if (v0[b].x && v1[b].x) fl = red_lookup_q(htab, hbits, nd[b], &q);
                mo |= (fl & 1u) << tt[b];
                mi |= ((fl >> 1) & 1u) << tt[b];
                if (ids.list && (fl & 2u)) lq[threadIdx.x * 32 + tt[b]] = q;
            }
        }
    }
#if SP_PROBE_BATCH > 1
    // The candidates' home slots are requested SP_PROBE_BATCH at a time before the first answer is looked at: a lane that
    // walks its ~7 candidates one round trip after the other keeps one line in flight, and with four waves per SIMD the
    // chip then holds 2.6e5 requests -- what the memory system needs to deliver its 51 G random lines per second
    // (scripts/micro/random_lines.hip) only if no lane ever computes or idles behind a longer lane of its wave.
    while (cand) {
        u32 tt[SP_PROBE_BATCH], hh[SP_PROBE_BATCH];
        u64 nd[SP_PROBE_BATCH];
        ulonglong2 v[SP_PROBE_BATCH];
        bool have[SP_PROBE_BATCH];
#pragma unroll
        for (int b = 0; b < SP_PROBE_BATCH; b++) {
            have[b] = cand != 0u;
            const u32 t = have[b] ? (u32)__ffs(cand) - 1u : 0u;
            cand &= cand - 1u;                                  // 0 stays 0
            tt[b] = t;
            nd[b] = (t ? ((w0 << (2 * t)) | (w1 >> (64 - 2 * t))) : w0) >> (64 - 2 * K);
            hh[b] = red_hash(nd[b], hbits);
        }
#pragma unroll
        for (int b = 0; b < SP_PROBE_BATCH; b++) {
            v[b] = make_ulonglong2(0ull, 0ull);
            if (have[b]) v[b] = *reinterpret_cast<const ulonglong2 *>(&htab[hh[b]]);
        }
#pragma unroll
        for (int b = 0; b < SP_PROBE_BATCH; b++) {
            if (!have[b] || v[b].x == 0ull) continue;           // no candidate / empty home slot: not a red node
            u32 fl, q = 0;
            if ((v[b].x >> 2) == nd[b]) { fl = (u32)(v[b].x & 3ull); q = (u32)(v[b].y >> 32); }
            else {                                              // the home slot holds another node: on along the line
                const u32 mask = hmask(hbits);
                u32 h = (hh[b] + 1u) & mask;
                fl = 0;
                for (;;) {
                    const ulonglong2 x = *reinterpret_cast<const ulonglong2 *>(&htab[h]);
                    if (x.x == 0ull) break;
                    if ((x.x >> 2) == nd[b]) { fl = (u32)(x.x & 3ull); q = (u32)(x.y >> 32); break; }
                    h = (h + 1u) & mask;
                }
            }
            mo |= (fl & 1u) << tt[b];
            mi |= ((fl >> 1) & 1u) << tt[b];
            if (ids.list && (fl & 2u)) lq[threadIdx.x * 32 + tt[b]] = q;
        }
    }
#else
    while (cand) {
        u32 t = (u32)__ffs(cand) - 1u;
        cand &= cand - 1u;
        u64 win = t ? ((w0 << (2 * t)) | (w1 >> (64 - 2 * t))) : w0;
        u32 q = 0;
        const u32 fl = red_lookup_q(htab, hbits, win >> (64 - 2 * K), &q);      // key, cursor and block id: one 16-byte slot
        mo |= (fl & 1u) << t;
        mi |= ((fl >> 1) & 1u) << t;
        if (ids.list && (fl & 2u)) lq[threadIdx.x * 32 + t] = q;
    }
#endif
    // special module: multi-out iff listed in specialBranch (src/generateSP.c:612-624).  Collections of many records --
    // a third of all positions of a read set are special, and millions of them branches -- carry the list as a bitmap over
    // the text positions (one word beside the text word instead of a binary search per special position)
    if (brbits) {
        if (spec) mo |= (u32)sep_window(brbits, i0) & spec;
    } else if (nbranch) {
        while (spec) {
            u32 t = (u32)__ffs(spec) - 1u;
            spec &= spec - 1u;
            u64 i = i0 + t;
            u64 lo = lower_bound_dev<u64>(branch, 0, nbranch, i);
            if (lo < nbranch && branch[lo] == i) mo |= 1u << t;
        }
    }
    momask[g] = mo;
    mimask[g] = mi;
    }
    if (!ids.list) return;
    // The block id of a multi-in node sat in the table line the search fetched: pass 2 gets it from here instead of
    // searching the table again.  The ids of a wave (64 lanes = 2048 positions) go out as one run, in position order,
    // wherever the bump counter puts it; pass 2 finds the run through wave_base and the lane's rank in the wave.
    const u32 cnt = (u32)__popc(mi);
    u32 incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 v = __shfl_up(incl, d, 64);
        if ((int)lane_id() >= d) incl += v;
    }
    const u32 total = __shfl(incl, 63, 64);
    u64 base = 0;
    if (lane_id() == 0 && total) base = atomicAdd(ids.count, (unsigned long long)total);
    base = ((u64)__shfl((u32)(base >> 32), 0, 64) << 32) | __shfl((u32)base, 0, 64);
    if (lane_id() == 0 && active) ids.wave_base[(g - ids.g0) >> 6] = base;
    u64 o = base + incl - cnt;
    while (mi) {
        const u32 t = (u32)__ffs(mi) - 1u;
        mi &= mi - 1u;
        ids.list[o++] = lq[threadIdx.x * 32 + t];
    }
}

// pass 2 (over groups of 32 positions): spIndex = exclusive scan of the multi-out bits.  SP symbols are
// written in position order; multi-in positions are compacted into a work list of 16-byte items
// (node, spIndex << 4 | pred): the blue fill never touches the text again
struct SpCountF {
    const u32 *momask; const u32 *mimask;
    __device__ void count2(u64 g, u32 *a, u32 *b) const { *a = (u32)__popc(momask[g]); *b = (u32)__popc(mimask[g]); }
};
struct SpEmitArgs {
    const u64 *text; const u64 *sepbits; u64 n; int K;
    const u32 *momask; const u32 *mimask;
    u8 *spsym; ulonglong2 *mi_list;
    u64 g0;                    // first group of the slice (the scan arrays are indexed relative to it)
    u64 sp_base;               // SP symbols emitted by the slices before this one
    // routed != nullptr: a multi-in position becomes block id << qshift | spIndex << 3 | pred at once, the block id
    // from pass 1's list (no work list, no second search of the node table)
    u64 *routed; int qshift; SpBlockIds ids;
};
__global__ __launch_bounds__(DEBWT_BLOCK) void k_sp_emit(SpEmitArgs a, u64 ngroups, u64 chunk,
                                                          const u32 *__restrict__ off_mo,
                                                          const u32 *__restrict__ off_mi) {
    __shared__ u32 tmp[2 * DEBWT_WAVES];
#if SP_EMIT_STAGE
    // the SP symbols of a tile (at most 32 per lane) are gathered in LDS and leave as aligned 4-byte words: a lane's own
    // ~3.4 symbols written byte by byte cost the pass a store instruction per symbol
    __shared__ __attribute__((aligned(16))) u8 ssym[DEBWT_BLOCK * 32 + 16];
    // ... and so do the routed blue entries and the block ids they are made of (a lane's ~3.4 entries lie ~27 bytes from its
    // neighbour's: every store instruction of a wave touched ~14 lines): tiles of up to SP_EMIT_CAP multi-in positions
    constexpr u32 SP_EMIT_CAP = 2048;
    __shared__ u32 sq[SP_EMIT_CAP];
    __shared__ u64 sr[SP_EMIT_CAP];
#endif
    u64 beg = (u64)blockIdx.x * chunk;
    u64 end = beg + chunk < ngroups ? beg + chunk : ngroups;
    u64 base_mo = (u64)off_mo[blockIdx.x] + a.sp_base;
    u32 base_mi = off_mi[blockIdx.x];
    for (u64 tile = beg; tile < end; tile += DEBWT_BLOCK) {
        u64 g = tile + threadIdx.x;
        u32 mo = 0, mi = 0;
        if (g < end) { mo = a.momask[a.g0 + g]; mi = a.mimask[a.g0 + g]; }
        g += a.g0;
        u32 val[2] = {(u32)__popc(mo), (u32)__popc(mi)}, ex[2], tot[2];
        block_scan_excl_vec<2>(val, ex, tot, tmp);
        u64 off = base_mo + ex[0];
        u32 omi = base_mi + ex[1];
        u32 all = mo | mi;
#if SP_EMIT_STAGE
        u32 ls_o = ex[0];
#endif
        u64 w0 = 0, w1 = 0, wp = 0, sbp = 0;
        // block ids of the lane: its run starts at the wave's base + the lane's rank in the wave (the waves of this
        // tile are the waves of pass 1); read 4 ahead of their use
        const u32 *qrun = nullptr;
        u32 qn = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
#if SP_EMIT_STAGE
        const bool staged = a.routed && tot[1] <= SP_EMIT_CAP;               // (uniform)
        u32 lr_o = ex[1];
        if (staged) {
            // every wave brings its own run of block ids in, 64 at a time
            const u32 wex0 = __shfl(ex[1], 0, 64), wcnt = __shfl(ex[1] + (u32)__popc(mi), 63, 64) - wex0;
            if (wcnt) {
                const u32 *run = a.ids.list + a.ids.wave_base[((tile + a.g0 + (threadIdx.x & ~63u)) - a.ids.g0) >> 6];
                for (u32 j = threadIdx.x & 63u; j < wcnt; j += 64) sq[wex0 + j] = run[j];
            }
            __syncthreads();
        } else
#endif
        if (a.routed) {
            const u32 wex = ex[1] - __shfl(ex[1], 0, 64);
            if (mi) {
                qrun = a.ids.list + a.ids.wave_base[(g - a.ids.g0) >> 6] + wex;
                qn = (u32)__popc(mi);
                q0 = qrun[0];
                if (qn > 1) q1 = qrun[1];
                if (qn > 2) q2 = qrun[2];
                if (qn > 3) q3 = qrun[3];
                qrun += 4;
            }
        }
        u64 sbn = 0;                                           // bit b: separator at the group's position + b
        if (all) {                                             // the group's text words, the word and separators before it
            w0 = a.text[g]; w1 = a.text[g + 1];
            if (g) { wp = a.text[g - 1]; sbp = sep_window(a.sepbits, (g << 5) - 1); sbn = sbp >> 1; }
            else sbn = sep_window(a.sepbits, 0);
        }
        while (all) {
            u32 t = (u32)__ffs(all) - 1u;
            all &= all - 1u;
            u64 i = (g << 5) + t;
            if ((mi >> t) & 1u) {
                u64 win = t ? ((w0 << (2 * t)) | (w1 >> (64 - 2 * t))) : w0;
                u64 pred;                                      // src/generateSP.c:584-605
                if (i == 0) pred = 5;
                else if ((sbp >> t) & 1ull) pred = 4;          // bit t of sbp = position i-1
                else pred = t ? ((w0 >> (2 * (32 - t))) & 3ull) : (wp & 3ull);
#if SP_EMIT_STAGE
                if (staged) {
                    sr[lr_o] = ((u64)sq[lr_o] << a.qshift) | (off << 3) | pred;
                    lr_o++;
                } else
#endif
                if (a.routed) {
                    a.routed[omi++] = ((u64)q0 << a.qshift) | (off << 3) | pred;
                    q0 = q1; q1 = q2; q2 = q3;
                    if (qn > 4) { q3 = *qrun++; qn--; }
                } else
                    a.mi_list[omi++] = make_ulonglong2(win >> (64 - 2 * a.K), (off << 4) | pred);
            }
            if ((mo >> t) & 1u) {
                // the symbol K ahead; the separator itself when it follows the window (src/generateSP.c:626-660)
                // (position i + K is symbol t + K <= 62 of the group's word pair: no load)
                const u32 b = t + (u32)a.K;
                const u64 j = i + (u64)a.K;
                u8 sy;
                if ((sbn >> b) & 1ull) sy = (j == a.n - 1) ? 5 : 4;
                else sy = (u8)((b < 32 ? w0 >> (2 * (31 - b)) : w1 >> (2 * (63 - b))) & 3ull);
#if SP_EMIT_STAGE
                ssym[ls_o++] = sy; off++;
#else
                a.spsym[off++] = sy;
#endif
            }
        }
#if SP_EMIT_STAGE
        __syncthreads();
        {
            const u32 ntot = tot[0];
            const u32 head = (u32)((0ull - base_mo) & 3ull) < ntot ? (u32)((0ull - base_mo) & 3ull) : ntot;   // bytes before the first aligned word
            if (threadIdx.x < head) a.spsym[base_mo + threadIdx.x] = ssym[threadIdx.x];
            const u32 nw = (ntot - head) >> 2;
            u32 *dw = reinterpret_cast<u32 *>(a.spsym + base_mo + head);
            for (u32 j = threadIdx.x; j < nw; j += DEBWT_BLOCK) {
                const u32 o = head + 4u * j;
                dw[j] = (u32)ssym[o] | ((u32)ssym[o + 1] << 8) | ((u32)ssym[o + 2] << 16) | ((u32)ssym[o + 3] << 24);
            }
            const u32 done = head + 4u * nw;
            if (threadIdx.x < ntot - done) a.spsym[base_mo + done + threadIdx.x] = ssym[done + threadIdx.x];
            if (staged)
                for (u32 j = threadIdx.x; j < tot[1]; j += DEBWT_BLOCK) a.routed[base_mi + j] = sr[j];
        }
#endif
        base_mo += tot[0]; base_mi += tot[1];
    }
}

// one thread per multi-in position: node -> table slot -> slot in the block through the block's cursor, which
// lives in the same table line (the reference's per-red-entry lock, src/generateSP.c:662-680)
template <int ABS32>
__global__ __launch_bounds__(DEBWT_BLOCK) void k_blue_fill(const ulonglong2 *__restrict__ mi_list, u64 B,
                                                            HSlot *__restrict__ htab, int hbits,
                                                            const u64 *__restrict__ blk_start, u32 qbase,
                                                            u64 *__restrict__ blue, int mzK) {
    u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    ulonglong2 it = mi_list[b];
    u32 fl;
    u32 h = red_lookup(htab, hbits, it.x, &fl, mzK);
    if (!fl) return;                                                        // (a red node carries a flag; with 2^32 slots every h is a slot)
    if (ABS32) {
        if (htab[h].cur == HCURSOR_SKIP) return;                            // block owned by another shard
        u32 slot = atomicAdd(&htab[h].cur, 1u);                             // absolute slot: starts at the block start
        blue[slot] = it.y;                                                  // pred | spIndex << 4 (:666-672)
    } else {
        const uint2 cq = *reinterpret_cast<const uint2 *>(&htab[h].cur);    // cursor and block id: one 8-byte load
        if (cq.x == HCURSOR_SKIP) return;
        u32 slot = atomicAdd(&htab[h].cur, 1u);
        blue[blk_start[cq.y - qbase] + slot] = it.y;
    }
}

// Atomic-free blue fill: k_sp_emit turns a multi-in position into (block id << qshift | spIndex << 3 | pred); sorting these words by
// their block bits puts every entry into its block (the blocks' blue slots are the exclusive scan of their sizes in
// block order), k_blue_strip turns them into blue entries (pred | spIndex << 4).  A sharded build routes the same
// words (global block ids) to the shard that owns the block, where k_blue_place puts them into the block through a
// cursor per owned block.
// (src may be dst: every thread rewrites its own word)
__global__ void k_blue_strip(const u64 *src, u64 *dst, u64 n, int qshift) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 e = src[i];
    dst[i] = (e & 7ull) | (((e & ((1ull << qshift) - 1ull)) >> 3) << 4);
}
__global__ __launch_bounds__(DEBWT_BLOCK) void k_blue_place(const u64 *__restrict__ ent, u64 count, u32 qbase,
                                                             u32 Qlocal, int qshift, u32 *__restrict__ qcursor,
                                                             const u64 *__restrict__ blk_start,
                                                             u64 *__restrict__ blue) {
    u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= count) return;
    u64 e = ent[b];
    u32 q = (u32)(e >> qshift) - qbase;
    if (q >= Qlocal) return;                                                // misrouted entry: never index outside
    u32 slot = atomicAdd(&qcursor[q], 1u);
    blue[blk_start[q] + slot] = (e & 7ull) | (((e & ((1ull << qshift) - 1ull)) >> 3) << 4);   // pred | spIndex << 4
}

// SP symbols -> 3 bits per symbol in one MSB-first bit stream (symbol s at stream bits [3s, 3s+3)); a window
// is the 63 bits = 21 symbols that start at a symbol: integer order of windows = order of the symbol strings
// under A<C<G<T<#<$ (codes 0..5).  One thread packs 64 symbols = 3 words.
#define SP_WIN 21
__global__ void k_pack_sp(const u8 *__restrict__ spsym, u64 S, u64 ntriples, u64 *__restrict__ spn) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntriples) return;
    u64 w[3] = {0, 0, 0};
    u64 base = t << 6;
    // the thread's 64 symbols as four 16-byte loads (64 byte loads per thread, 64 bytes from the neighbour's, ran at 1.2 TB/s)
    u32 q[16];
    if (base + 64 <= S) {
        const uint4 *p4 = reinterpret_cast<const uint4 *>(spsym + base);
#pragma unroll
        for (int i = 0; i < 4; i++) { const uint4 v = p4[i]; q[4 * i] = v.x; q[4 * i + 1] = v.y; q[4 * i + 2] = v.z; q[4 * i + 3] = v.w; }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            u32 x = 0;
            for (int b = 0; b < 4; b++) { const u64 sidx = base + 4 * i + b; x |= (sidx < S ? (u32)spsym[sidx] : 0u) << (8 * b); }
            q[i] = x;
        }
    }
#pragma unroll
    for (u32 j = 0; j < 64; j++) {
        u64 c = (u64)((q[j >> 2] >> (8 * (j & 3))) & 0xFFu);
        const u32 bit = 3 * j;                 // position in the 192-bit group, MSB first
        const u32 wi = bit >> 6, off = bit & 63;
        if (off <= 61) w[wi] |= c << (61 - off);
        else { w[wi] |= c >> (off - 61); w[wi + 1] |= c << (125 - off); }
    }
    spn[3 * t] = w[0]; spn[3 * t + 1] = w[1]; spn[3 * t + 2] = w[2];
}

__device__ __forceinline__ u64 sp_window(const u64 *__restrict__ spn, u64 s) {
    u64 b = 3 * s;
    u64 w = b >> 6;
    u32 sh = (u32)(b & 63);
    u64 a = spn[w];
    u64 v = sh ? ((a << sh) | (spn[w + 1] >> (64 - sh))) : a;
    return v >> 1;                              // 21 symbols
}

// A PAIR of windows (42 symbols from s on) lies in three consecutive words: fetched as three loads that depend on nothing
// but s, so that a thread holding several rows has all their words in flight before it shifts any of them -- the loops that
// asked for one row's windows, shifted them, stored them and only then went on to the next row kept two to four loads in
// flight per lane, and the gather rate of the SP code rises with the requests in flight (52 G window gathers a second at four
// per lane and eight waves per SIMD, scripts/micro/gather16.hip).  Bit-identical to sp_window(s), sp_window(s + SP_WIN).
struct SpTriple { u64 a0, a1, a2; u32 sh; };
__device__ __forceinline__ SpTriple sp_fetch3(const u64 *__restrict__ spn, u64 s) {
    const u64 b = 3 * s, w = b >> 6;
    return SpTriple{spn[w], spn[w + 1], spn[w + 2], (u32)(b & 63)};     // (the packed code ends in two groups of zeros)
}
__device__ __forceinline__ void sp_pair(const SpTriple &t, u64 *w, u64 *x) {
    const u64 v1 = t.sh ? ((t.a0 << t.sh) | (t.a1 >> (64 - t.sh))) : t.a0;
    // the second window starts 63 bits on: bit 63 of a0 when sh == 0, else bit sh - 1 of a1
    const u64 v2 = t.sh == 0 ? ((t.a0 << 63) | (t.a1 >> 1)) : (t.sh == 1 ? t.a1 : ((t.a1 << (t.sh - 1)) | (t.a2 >> (65 - t.sh))));
    *w = v1 >> 1; *x = v2 >> 1;
}
// rows x = x0 + i * stride (i < NB) below m: sw / sx <- their window pair at depth `dd` pairs in, 0 for rows that are not
// `wanted` or start behind the code.  All the words first, then the shifts and the LDS stores.
template <int NB, class Wanted>
__device__ __forceinline__ void sp_gather_pairs(const u64 *__restrict__ spn, u64 S, const u64 *se, u64 *sw, u64 *sx, u32 x0, u32 stride,
                                                u32 m, u64 dd, Wanted wanted) {
    SpTriple t[NB];
    bool live[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) {
        const u32 x = x0 + (u32)i * stride;
        live[i] = false;
        if (x < m) {
            const u64 pos = (se[x] >> 4) + dd * (2 * SP_WIN);
            live[i] = pos < S && wanted(x);
            if (live[i]) t[i] = sp_fetch3(spn, pos);
        }
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
        const u32 x = x0 + (u32)i * stride;
        if (x < m) {
            u64 w = 0ull, v = 0ull;
            if (live[i]) sp_pair(t[i], &w, &v);
            sw[x] = w; sx[x] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// blue-block sort (sortBlue/myQsort/cmpSP, src/sortBlue.c:109-280)

// a < b by SP suffix beyond the first window (entries are distinct SP positions of one node: the
// suffixes differ before the unique '$' that ends the SP code)
__device__ __forceinline__ bool sp_less_deep(const u64 *__restrict__ spn, u64 S, u64 ea, u64 eb) {
    u64 a = (ea >> 4) + SP_WIN, b = (eb >> 4) + SP_WIN;
    while (a < S && b < S) {
        u64 wa = sp_window(spn, a), wb = sp_window(spn, b);
        if (wa != wb) return wa < wb;
        a += SP_WIN; b += SP_WIN;
    }
    return a > b;   // not reached on consistent input
}

#define BLUE_WAVE_CAP 512   // largest block sorted by a single-wave workgroup

// Block sort by level-wise refinement (the data-parallel form of myQsort + cmpSP, src/sortBlue.c:109-280):
// round d orders the still-tied entries by their d-th pair of 21-symbol SP windows (42 symbols); entries whose tie group has
// become a single row, or whose group carries a single BWT symbol (the reference's early-out,
// src/sortBlue.c:192-219), leave the game.  One SP gather per unresolved entry per round -- not per
// comparison.  A workgroup of NT threads holds the block in LDS; each round is a bitonic network on
// (group, window) pairs, a prefix-max for the new group ids and LDS atomics for the group census.
// Hand-off of deep tie groups: in collections of near-identical genomes the rows of a large block (a repeat family's
// node over all genomes) separate into small groups within two or three windows, but a few of those groups -- the
// same locus in several genomes -- stay tied for hundreds of SP symbols, and every further round costs the whole
// workgroup its barriers.  Once all unresolved groups hold <= SPLIT rows (template parameter, 0 = never) a block of
// the two larger size classes writes its rows out in their current order and queues each unresolved group as a block
// of its own (start, rows, first instance, windows already equal) for the wave-per-block kernels.
struct BlueSub {
    u64 *start; u32 *freq; u64 *j0; u32 *depth;      // sub-block table (written by SPLIT kernels, read through depth0)
    u32 *count; u32 cap;                              // entries reserved / capacity
};

#ifndef BLUE_BINS_DIV_1024
#define BLUE_BINS_DIV_1024 128         // rows per range the sample-sort split of the 513..1024-row class aims at: 8 ranges (measured
                                       // at 10 x 300 Mbp, blue stage: 4 ranges 35.8 ms, 8: 32.8, 16: 33.7, 32: 35.3, 64: 41.6)
#endif
#ifndef BLUE_SAMPLE_SPLIT
#define BLUE_SAMPLE_SPLIT 1
#endif
#ifndef BLUE_RANK_MAX
#define BLUE_RANK_MAX 128u   // workgroup kernels: tie groups up to this size are ordered by counting ranks
#endif
#define GC_CNT(v) ((v) & 0xFFFFu)
#define GC_UNRES(v) (((v) & 0xFFFFu) > 1u && (((v) >> 16) & (((v) >> 16) - 1u)) != 0u)   // several rows, several symbols
// Waves per SIMD asked of the compiler per class: these kernels wait on SP gathers, and a few spilled registers cost less
// than the waves they free (measured at 30 Gbp: <=128 rows 7 -> 8 waves 43.6 -> 38.7 ms per launch; 257..512 rows 5 -> 8
// waves 66.9 -> 51.1 ms, their queued ranges 7 -> 8 waves 77.2 -> 68.9 ms; 513..1024 rows stay at 4: 5 waves with 88
// bytes of scratch ran 3 % slower).  129..256 rows are held to 5 by their LDS.
template <int NT, int CAP, int SPLIT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(CAP <= 128 ? 8 : (CAP <= 256 ? 5 : (CAP <= 512 ? 8 : (CAP <= 1024 ? 4 : 2)))))) void k_blue_refine(u64 *__restrict__ blue, const u64 *__restrict__ bstart,
                                                     const u32 *__restrict__ mi_freq, const u64 *__restrict__ mi_j0,
                                                     u32 Q, u32 lo_excl, const u64 *__restrict__ spn, u64 S,
                                                     u8 *__restrict__ mchar, const u32 *__restrict__ depth0,
                                                     const u32 *__restrict__ Qdev, BlueSub sub,
                                                     const u8 *__restrict__ done = nullptr) {
    // done (optional): blocks k_blue_classify has finished (one byte per table entry) are skipped
    __shared__ u64 se[CAP];     // entries
    __shared__ u64 sw[CAP];     // current window, first 21 symbols
    __shared__ u64 sx[CAP];     //                 next 21 symbols
    // 30 bytes of LDS per row (36 with 32-bit group ids and separate census words): the workgroups a CU holds are what
    // these kernels run on
    __shared__ unsigned short sg[CAP];   // tie group = index of its first row
    __shared__ u32 gcm[CAP];    // per group: rows (low 16 bits) | BWT symbols present (high 16 bits)
    __shared__ u32 wtot[NT / 64 + 1];
    __shared__ u32 flag;
    __shared__ u32 smax;
    __shared__ u32 sub_n, sub_i, sub_base;
    // sample-sort split of a large block (workgroup classes): 64 sampled rows, BINS - 1 splitters (measured at 30 Gbp:
    // 8, 16 or 32 ranges per block all give a blue stage of 0.48-0.50 s; 64 ranges overflow the sub-block table)
    constexpr int BINS = (SPLIT && NT == 256) ? (CAP == 1024 ? CAP / BLUE_BINS_DIV_1024 : CAP / 32) : 1, SAMPLES = BINS > 1 ? 4 * BINS : 64;
    static_assert(SAMPLES <= NT || BINS == 1, "one thread per sampled row");
    __shared__ u64 spl_w[BINS], spl_x[BINS];
    // the sampled windows of the split live in the census words, which are not in use before the first round
    static_assert(BINS == 1 || (size_t)CAP * sizeof(u32) >= (size_t)SAMPLES * 2 * sizeof(u64), "samples alias the census words");
    u64 *const smp_w = reinterpret_cast<u64 *>(gcm), *const smp_x = smp_w + SAMPLES;
    __shared__ u32 bin_cnt[BINS], bin_start[BINS], bin_cur[BINS], bin_slot[BINS];
    const u32 tid = threadIdx.x;
    if (Qdev) { u32 qd = *Qdev; Q = qd < Q ? qd : Q; }           // sub-block table: entries written so far (<= its capacity)
    // The kernel of a size class walks the whole block table and skips the blocks of the other classes: 64 table
    // entries per step (one per lane, the same in every wave of the workgroup), a ballot picks the blocks to sort -- one
    // dependent global load per 64 entries instead of one per entry (the sub-block queue of a 30 Gbp build holds
    // several 10^7 entries for each of its three kernels to walk).
    // (a table too short to give every workgroup 64 entries is walked one entry per workgroup and step, as before:
    // otherwise a few workgroups would sort 64 blocks each one after the other while the rest idle)
    const u32 E = (u64)Q >= (u64)gridDim.x * 64 ? 64u : 1u;
    for (u64 q0 = (u64)blockIdx.x * E; q0 < Q; q0 += (u64)gridDim.x * E) {
      const u64 ql = q0 + (tid & 63u);
      const u32 ml = ((tid & 63u) < E && ql < Q) ? mi_freq[ql] : 0u;
      u64 todo = __ballot(ml > lo_excl && ml <= (u32)CAP && !(done && done[ql]));
      for (; todo; todo &= todo - 1) {
        const u32 q = (u32)(q0 + (u32)__builtin_ctzll(todo));
        const u32 m = mi_freq[q];
        const u64 d0 = depth0 ? depth0[q] : 0;
        u32 maxg = m;
        const u64 b0 = bstart[q];
        const u64 j0 = mi_j0[q];
        u32 P = 2;
        while (P < m) P <<= 1;
        u32 mask = 0;
        if (tid == 0) flag = 0;
        for (u32 x = tid; x < P; x += NT) {
            if (x < m) {
                u64 e = blue[b0 + x];
                se[x] = e; sg[x] = 0; mask |= 1u << (e & 15);
            } else { se[x] = ~0ull; sg[x] = 0xFFFFu; sw[x] = ~0ull; sx[x] = ~0ull; }
        }
        __syncthreads();
        if (mask) atomicOr(&flag, mask);
        __syncthreads();
        bool active = (flag & (flag - 1)) != 0;               // >= 2 distinct symbols in the block
        __syncthreads();
        bool handed = false, preloaded = false;
        if (BLUE_SAMPLE_SPLIT && SPLIT && NT == 256 && active && sub.cap) {
            // Sample-sort split: instead of sorting up to 2048 rows with workgroup-wide bitonic rounds, cut the block
            // into BINS ranges of the first 42 SP symbols (splitters from a sorted sample of 64 rows) and queue every
            // range as a block of its own for the wave-per-block kernels -- ranges are ordered among themselves and
            // rows with equal windows share a range, so sorting the ranges sorts the block.
            for (u32 x = tid; x < m; x += 4 * NT)
                sp_gather_pairs<4>(spn, S, se, sw, sx, x, NT, m, d0, [](u32) { return true; });
            if (tid < BINS) { bin_cnt[tid] = 0; bin_cur[tid] = 0; }
            __syncthreads();
            if (tid < SAMPLES) { const u32 i = (u32)(((u64)tid * m) / SAMPLES); smp_w[tid] = sw[i]; smp_x[tid] = sx[i]; }
            __syncthreads();
            if (tid < SAMPLES) {                                // rank of the sample among the samples (counting)
                const u64 w = smp_w[tid], x2 = smp_x[tid];
                u32 rank = 0;
                for (u32 j = 0; j < SAMPLES; j++) {
                    const u64 wj = smp_w[j], xj = smp_x[j];
                    rank += (wj != w ? wj < w : (xj != x2 ? xj < x2 : j < tid)) ? 1u : 0u;
                }
                // splitter b = sorted sample (b+1)*SAMPLES/BINS - 1, for b < BINS-1
                if ((rank + 1) % (SAMPLES / BINS) == 0 && (rank + 1) / (SAMPLES / BINS) <= BINS - 1) {
                    const u32 b = (rank + 1) / (SAMPLES / BINS) - 1;
                    spl_w[b] = w; spl_x[b] = x2;
                }
            }
            __syncthreads();
            for (u32 x = tid; x < m; x += NT) {                // range = number of splitters below the row's windows
                const u64 w = sw[x], x2 = sx[x];
                u32 lo = 0, hi = BINS - 1;
                while (lo < hi) {
                    const u32 mid = (lo + hi) >> 1;
                    const bool below = spl_w[mid] != w ? spl_w[mid] < w : spl_x[mid] < x2;   // splitter < row
                    if (below) lo = mid + 1; else hi = mid;
                }
                sg[x] = lo;
                atomicAdd(&bin_cnt[lo], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                u32 acc = 0, big = 0, nonempty = 0;
                for (int b = 0; b < BINS; b++) {
                    bin_start[b] = acc; acc += bin_cnt[b];
                    big = bin_cnt[b] > big ? bin_cnt[b] : big;
                    bin_slot[b] = nonempty;                     // its entry in the sub-block table, counted from sub_base
                    nonempty += bin_cnt[b] ? 1u : 0u;
                }
                sub_n = 0;                                      // 0: no hand-off
                smax = big;
                if (big > BLUE_RANK_MAX && big <= (u32)SPLIT) {
                    const u32 base = atomicAdd(sub.count, nonempty);
                    if ((u64)base + nonempty <= (u64)sub.cap) { sub_n = nonempty; sub_base = base; }
                }
            }
            __syncthreads();
            if (smax <= BLUE_RANK_MAX) {
                // Every range is small enough to be ordered by counting ranks: keep the block here.  The rows move into
                // range order in LDS, a range becomes a tie group of the first round, whose windows are loaded already --
                // no hand-off of every range and no second gather of its windows by the wave kernels; only groups that
                // stay tied behind these 42 symbols with more than one BWT symbol (about 1 % in a pan-genome) are queued.
                constexpr int EPT = CAP / NT;
                u32 npos[EPT];
                u64 ne[EPT], nw[EPT], nx[EPT];
#pragma unroll
                for (int c = 0; c < EPT; c++) {
                    const u32 x = tid + (u32)c * NT;
                    npos[c] = 0xFFFFFFFFu;
                    if (x < m) {
                        const u32 b = sg[x];
                        npos[c] = bin_start[b] + atomicAdd(&bin_cur[b], 1u);
                        ne[c] = se[x]; nw[c] = sw[x]; nx[c] = sx[x];
                    }
                }
                __syncthreads();
#pragma unroll
                for (int c = 0; c < EPT; c++)
                    if (npos[c] != 0xFFFFFFFFu) { se[npos[c]] = ne[c]; sw[npos[c]] = nw[c]; sx[npos[c]] = nx[c]; }
                __syncthreads();
                if (tid < BINS && bin_cnt[tid]) {
                    const u32 g = bin_start[tid];
                    gcm[g] = bin_cnt[tid] | (0x3u << 16);
                    for (u32 y = g; y < g + bin_cnt[tid]; y++) sg[y] = g;
                }
                maxg = smax;
                preloaded = true;
            } else if (sub_n) {
                for (u32 x = tid; x < m; x += NT) {
                    const u32 b = sg[x];
                    blue[b0 + bin_start[b] + atomicAdd(&bin_cur[b], 1u)] = se[x];
                }
                if (tid < BINS && bin_cnt[tid]) {
                    const u32 e = sub_base + bin_slot[tid];
                    sub.start[e] = b0 + bin_start[tid]; sub.freq[e] = bin_cnt[tid]; sub.j0[e] = j0 + bin_start[tid];
                    sub.depth[e] = (u32)d0;
                }
                handed = true; active = false;
            } else {
                for (u32 x = tid; x < m; x += NT) sg[x] = 0;   // too skewed (or the table is full): the rounds below
            }
            __syncthreads();
        }
        if (!preloaded && !handed) {                           // round 0: one group (its first row is row 0), unresolved
            if (tid == 0) gcm[0] = m | (0x3u << 16);
            __syncthreads();
        }
        for (u64 depth = 0; active; depth++) {
            // 1. next window of every unresolved row (the sample-sort split has loaded those of its round already)
            if (!(preloaded && depth == 0))
            for (u32 x = tid; x < m; x += 4 * NT)
                sp_gather_pairs<4>(spn, S, se, sw, sx, x, NT, m, d0 + depth, [&](u32 y) { return GC_UNRES(gcm[sg[y]]); });
            __syncthreads();
            {   // Rows that tie deeply -- the copies of a repeat in every genome of a collection share hundreds of SP
                // symbols -- have equal windows round after round: when no unresolved row differs from the first row of
                // its group there is nothing to order and no group to split, the block moves on to the next windows.
                u32 differs = 0;
                for (u32 x = tid; x < m; x += NT) {
                    const u32 g = sg[x];
                    if (GC_UNRES(gcm[g])) differs |= (sw[x] != sw[g] || sx[x] != sx[g]) ? 1u : 0u;
                }
                if (!__syncthreads_or((int)differs) && !(preloaded && depth == 0)) {
                    active = (d0 + depth + 1) * (2 * SP_WIN) < S + 2 * SP_WIN;
                    continue;
                }
            }
            // 2. order every unresolved group by window: small groups by counting ranks inside the group
            //    (cost ~ group size), otherwise a bitonic network on (group, window) over the whole block
            if (maxg <= (NT == 256 ? BLUE_RANK_MAX : 32u)) {
                constexpr int EPT = CAP / NT;
                u32 npos[EPT];
                u64 ne[EPT], nw[EPT], nx[EPT];
#pragma unroll
                for (int c = 0; c < EPT; c++) {
                    u32 x = tid + (u32)c * NT;
                    npos[c] = 0xFFFFFFFFu;
                    if (x < m) {
                        u32 g = sg[x];
                        const u32 gc = gcm[g];
                        u32 cnt = GC_CNT(gc);
                        if (GC_UNRES(gc)) {
                            u64 wx = sw[x], xx = sx[x];
                            u32 rank = 0;
                            for (u32 y = g; y < g + cnt; y++) {
                                u64 wy = sw[y], xy = sx[y];
                                bool less = wy != wx ? wy < wx : (xy != xx ? xy < xx : y < x);
                                rank += less ? 1u : 0u;
                            }
                            npos[c] = g + rank; ne[c] = se[x]; nw[c] = wx; nx[c] = xx;
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int c = 0; c < EPT; c++)
                    if (npos[c] != 0xFFFFFFFFu) { se[npos[c]] = ne[c]; sw[npos[c]] = nw[c]; sx[npos[c]] = nx[c]; }
                __syncthreads();
            } else {
                for (u32 kk = 2; kk <= P; kk <<= 1) {
                    for (u32 jj = kk >> 1; jj > 0; jj >>= 1) {
                        for (u32 t = tid; t < (P >> 1); t += NT) {
                            u32 i = ((t & ~(jj - 1)) << 1) | (t & (jj - 1));
                            u32 l = i | jj;
                            u32 gi = sg[i], gl = sg[l];
                            u64 wi = sw[i], wl = sw[l], xi = sx[i], xl = sx[l];
                            bool l_less = gl != gi ? gl < gi : (wl != wi ? wl < wi : xl < xi);
                            bool i_less = gl != gi ? gi < gl : (wl != wi ? wi < wl : xi < xl);
                            bool up = (i & kk) == 0;
                            if (up ? l_less : i_less) {
                                u64 ei = se[i], el = se[l];
                                sg[i] = gl; sg[l] = gi; sw[i] = wl; sw[l] = wi; sx[i] = xl; sx[l] = xi;
                                se[i] = el; se[l] = ei;
                            }
                        }
                        __syncthreads();
                    }
                }
            }
            // 3. new groups: first row of each run of equal (group, window); prefix-max of (index+1)
            const u32 C = P > NT ? P / NT : 1;               // rows per thread, contiguous
            u32 xb = tid * C, run = 0;
            u32 loc[CAP / NT > 0 ? CAP / NT : 1];
            if (xb < P) {
#pragma unroll
                for (u32 c = 0; c < (CAP / NT > 0 ? CAP / NT : 1); c++) {
                    if (c < C) {
                        u32 x = xb + c;
                        bool bnd = x == 0 || sg[x] != sg[x - 1] || sw[x] != sw[x - 1] || sx[x] != sx[x - 1];
                        if (bnd) run = x + 1;
                        loc[c] = run;
                    }
                }
            }
            // exclusive prefix-max of `run` over threads
            u32 incl = run;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                u32 o = __shfl_up(incl, d, 64);
                if ((int)(tid & 63) >= d) incl = incl > o ? incl : o;
            }
            u32 excl = __shfl_up(incl, 1, 64);
            if ((tid & 63) == 0) excl = 0;
            if (NT > 64) {
                if ((tid & 63) == 63) wtot[tid >> 6] = incl;
                __syncthreads();
                for (u32 w = 0; w < (tid >> 6); w++) excl = excl > wtot[w] ? excl : wtot[w];
            }
            __syncthreads();                                   // all reads of sg/sw for boundaries done
            if (xb < P) {
#pragma unroll
                for (u32 c = 0; c < (CAP / NT > 0 ? CAP / NT : 1); c++) {
                    if (c < C) {
                        u32 x = xb + c;
                        u32 v = loc[c] > excl ? loc[c] : excl;
                        if (x < m) sg[x] = v - 1;
                        gcm[x] = 0;
                    }
                }
            }
            if (tid == 0) { flag = 0; smax = 0; }
            __syncthreads();
            // 4. census of the new groups
            for (u32 x = tid; x < m; x += NT) {
                u32 g = sg[x];
                atomicAdd(&gcm[g], 1u);
                atomicOr(&gcm[g], (1u << (se[x] & 15)) << 16);
            }
            __syncthreads();
            u32 any = 0;
            for (u32 x = tid; x < m; x += NT) {
                u32 g = sg[x];
                if (g == x && GC_UNRES(gcm[g])) any = any > GC_CNT(gcm[g]) ? any : GC_CNT(gcm[g]);
            }
            if (any) { flag = 1; atomicMax(&smax, any); }
            __syncthreads();
            active = flag != 0 && (d0 + depth + 1) * (2 * SP_WIN) < S + 2 * SP_WIN;
            maxg = smax;
            __syncthreads();
            if (SPLIT && active && maxg <= (u32)SPLIT) {
                // queue the unresolved groups: one reservation per block, then one entry per group
                if (tid == 0) sub_n = 0;
                __syncthreads();
                u32 mine = 0;
                for (u32 x = tid; x < m; x += NT)
                    if (sg[x] == x && GC_UNRES(gcm[x])) mine++;
                if (mine) atomicAdd(&sub_n, mine);
                __syncthreads();
                if (tid == 0) { sub_base = atomicAdd(sub.count, sub_n); sub_i = 0; }
                __syncthreads();
                const u32 base = sub_base;
                if ((u64)base + sub_n <= (u64)sub.cap) {                 // else: the table is full, keep refining here
                    for (u32 x = tid; x < m; x += NT)
                        if (sg[x] == x && GC_UNRES(gcm[x])) {
                            const u32 e = base + atomicAdd(&sub_i, 1u);
                            sub.start[e] = b0 + x; sub.freq[e] = GC_CNT(gcm[x]); sub.j0[e] = j0 + x;
                            sub.depth[e] = (u32)(d0 + depth + 1);
                        }
                    active = false;
                }
                __syncthreads();
            }
        }
        if (!handed)
            for (u32 x = tid; x < m; x += NT) {
                u64 e = se[x];
                blue[b0 + x] = e;
                mchar[j0 + x] = (u8)(e & 15);
            }
        __syncthreads();
      }
    }
}

// Blocks of at most BLUE_TINY rows, four to a wave: a 16-lane group holds a block, one row per lane.  In a collection of ten
// genomes 70 % of the blocks are a node's one occurrence per genome -- ten rows -- and a wave per block spent ~500
// instructions on each of the 2.6 * 10^7 of them (k_blue_refine<64, 128, 0> is bound by its instructions: 8 waves per SIMD
// that issue a quarter of their cycles each).  The rounds are those of k_blue_refine for groups of up to 32 rows -- windows
// of the unresolved rows, stable counting ranks inside the tie group, new groups, census -- so the rows leave in the same
// order; a group of 16 lanes whose block is finished idles until the other three are.
#define BLUE_TINY 16u
// G = lanes (and rows at most) per block: 16 -> four blocks per wave, 32 -> two (blocks of 17..32 rows)
template <u32 G>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_blue_tiny(   // (64 VGPRs instead of 68: blue stage 173.7 -> 171.9 ms)
    u64 *__restrict__ blue, const u64 *__restrict__ bstart,
                                                  const u32 *__restrict__ mi_freq, const u64 *__restrict__ mi_j0, u32 Q,
                                                  const u64 *__restrict__ spn, u64 S, u8 *__restrict__ mchar,
                                                  const u32 *__restrict__ depth0, const u32 *__restrict__ Qdev) {
    __shared__ u64 se[64], sw[64], sx[64];
    __shared__ u32 gcm[64];                                    // per tie group (slot of its first row): rows | symbols << 16
    constexpr u32 NG = 64u / G;
    constexpr u64 GM = G == 32 ? 0xFFFFFFFFull : 0xFFFFull;
    const u32 lane = threadIdx.x, grp = lane / G, r = lane & (G - 1u), base = grp * G;
    if (Qdev) { const u32 qd = *Qdev; Q = qd < Q ? qd : Q; }
    const u32 E = (u64)Q >= (u64)gridDim.x * 64 ? 64u : 1u;     // the table walk of k_blue_refine
    for (u64 q0 = (u64)blockIdx.x * E; q0 < Q; q0 += (u64)gridDim.x * E) {
        const u64 ql = q0 + lane;
        const u32 ml = (lane < E && ql < Q) ? mi_freq[ql] : 0u;
        u64 todo = __ballot(ml > (G == 32 ? 16u : 0u) && ml <= G);
        while (todo) {
            u32 pick = 64u;                                    // the table entry of this lane's group (64: none left)
#pragma unroll
            for (u32 gi = 0; gi < NG; gi++)
                if (todo) { const u32 b = (u32)__builtin_ctzll(todo); todo &= todo - 1; if (grp == gi) pick = b; }
            const bool has = pick < 64u;
            const u32 q = has ? (u32)(q0 + pick) : 0u;
            const u32 m = has ? mi_freq[q] : 0u;
            const u64 b0 = has ? bstart[q] : 0ull, j0 = has ? mi_j0[q] : 0ull;
            const u64 d0 = (has && depth0) ? depth0[q] : 0ull;
            const bool valid = r < m;
            u64 e = valid ? blue[b0 + r] : ~0ull;
            u32 syms = valid ? 1u << (e & 15) : 0u;
#pragma unroll
            for (int d = (int)G / 2; d >= 1; d >>= 1) syms |= (u32)__shfl_xor((int)syms, d, 64);   // (lanes r ^ d: the group's own)
            u32 gid = valid ? 0u : 0xFFu;                      // first row of the lane's tie group; 0xFF: no row
            u32 gc = m | (syms << 16);                         // census of the lane's tie group
            bool blk_active = m > 1u && (syms & (syms - 1u)) != 0u;
            for (u64 depth = 0; __ballot(blk_active) != 0ull; depth++) {
                const bool unres = blk_active && valid && GC_UNRES(gc);
                const u64 pos = (e >> 4) + (d0 + depth) * (2 * SP_WIN);
                const bool live = unres && pos < S;
                u64 w = live ? sp_window(spn, pos) : 0ull, x = live ? sp_window(spn, pos + SP_WIN) : 0ull;
                se[lane] = e; sw[lane] = w; sx[lane] = x;
                __builtin_amdgcn_wave_barrier();
                const u32 hs = base + (gid & (G - 1u));
                const bool differs = unres && (w != sw[hs] || x != sx[hs]);
                const bool re = blk_active && ((__ballot(differs) >> base) & GM) != 0ull;   // (uniform in the group)
                // stable counting ranks inside the tie groups that are not resolved
                const bool mv = re && unres;
                const u32 cnt = GC_CNT(gc);
                u32 mc = mv ? cnt : 0u;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) { const u32 o = (u32)__shfl_xor((int)mc, d, 64); mc = mc > o ? mc : o; }
                u32 rank = 0;
                for (u32 t = 0; t < mc; t++) {
                    const u32 y = (gid + t) & (G - 1u);
                    const u64 wy = sw[base + y], xy = sx[base + y];
                    const bool less = wy != w ? wy < w : (xy != x ? xy < x : y < r);
                    rank += (mv && t < cnt && less) ? 1u : 0u;
                }
                __builtin_amdgcn_wave_barrier();
                if (mv) { const u32 np = base + gid + rank; se[np] = e; sw[np] = w; sx[np] = x; }
                __builtin_amdgcn_wave_barrier();
                if (re && valid) { e = se[lane]; w = sw[lane]; x = sx[lane]; }
                // new groups: first row of each run of equal (group, windows); prefix maximum of (index + 1)
                const u32 gp = (u32)__shfl_up((int)gid, 1, 64);
                const u64 wp = __shfl_up(w, 1, 64), xp = __shfl_up(x, 1, 64);
                u32 run = (r == 0u || gid != gp || w != wp || x != xp) ? r + 1u : 0u;
#pragma unroll
                for (int d = 1; d < (int)G; d <<= 1) {
                    const u32 o = (u32)__shfl_up((int)run, d, 64);
                    if ((int)r >= d) run = run > o ? run : o;
                }
                if (re && valid) gid = run - 1u;
                gcm[lane] = 0u;
                __builtin_amdgcn_wave_barrier();
                if (re && valid) {
                    atomicAdd(&gcm[base + gid], 1u);
                    atomicOr(&gcm[base + gid], (1u << (e & 15)) << 16);
                }
                __builtin_amdgcn_wave_barrier();
                if (re && valid) gc = gcm[base + gid];
                const bool left = ((__ballot(valid && GC_UNRES(gc)) >> base) & GM) != 0ull;
                if (blk_active) blk_active = (re ? left : true) && (d0 + depth + 1) * (2 * SP_WIN) < S + 2 * SP_WIN;
                __builtin_amdgcn_wave_barrier();
            }
            if (valid) { blue[b0 + r] = e; mchar[j0 + r] = (u8)(e & 15); }
        }
    }
}

// Blocks of a collection of many genomes by CLASSES around sampled splitters (what rs_local_unfit_kernel does for the key
// sort): a block of 513..1024 rows is a repeat family's node in every genome -- a hundred copies times ten genomes -- and
// the rows of one copy share their first 42 SP symbols and, in 99 % of the cases, their BWT symbol.  256 evenly spaced rows
// are sorted by their first pair of windows and their distinct pairs become splitters; every row finds its class by
// bisection -- "equal to splitter i" or "between splitters i-1 and i" -- a class counter gives it a place in its class and
// a class mask collects the BWT symbols of the class.  A class with one symbol is finished wherever its rows land inside
// it; a class with several is queued for the wave kernels as a block of its own -- one pair of windows deeper when it is
// a class of equals.  The rows go straight to their class's slots in HBM: no rounds, no network, no hand-off of every
// range.  A block with a class above `SPLIT` rows that needs sorting (or when the queue is full) is left to
// k_blue_refine, which skips the blocks marked in `done`.
template <int CAP, int SPLIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CAP <= 1024 ? 4 : 2)))
void k_blue_classify(u64 *__restrict__ blue, const u64 *bstart, const u32 *mi_freq, const u64 *mi_j0, u32 Q, u32 lo_excl, const u64 *__restrict__ spn, u64 S,
                     u8 *__restrict__ mchar, BlueSub sub, u8 *__restrict__ done,
                     const u32 *depth0 = nullptr, const u32 *__restrict__ Qsnap = nullptr, u32 *consume = nullptr) {
    // Queue mode (depth0, Qsnap, consume given): the table is the sub-block queue itself -- entries that start depth0 pairs of
    // windows in; only the *Qsnap entries that were complete before the launch are walked (the kernel appends while it
    // runs), and a finished entry is taken out of the queue (consume[q] = 0: the wave kernels behind skip it).
    constexpr int NT = 256, EPT = CAP / NT, NS = 256;
    __shared__ u64 se[CAP], sw[CAP], sx[CAP];
    __shared__ u64 pw[NS], px[NS];            // the sample, then its distinct pairs
    __shared__ u32 ccnt[2 * NS], cmsk[2 * NS];   // class 2 i: between splitters i - 1 and i; 2 i + 1: equal to splitter i
    __shared__ u32 wtmp[DEBWT_WAVES + 1];
    __shared__ u32 flag, sub_base;
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    if (Qsnap) { const u32 qs = *Qsnap; Q = qs < Q ? qs : Q; }
    const u32 E = (u64)Q >= (u64)gridDim.x * 64 ? 64u : 1u;
    for (u64 q0 = (u64)blockIdx.x * E; q0 < Q; q0 += (u64)gridDim.x * E) {
      const u64 ql = q0 + lane;
      const u32 ml = (lane < E && ql < Q) ? mi_freq[ql] : 0u;
      u64 todo = __ballot(ml > lo_excl && ml <= (u32)CAP);
      for (; todo; todo &= todo - 1) {
        const u32 q = (u32)(q0 + (u32)__builtin_ctzll(todo));
        const u32 m = mi_freq[q];
        const u64 b0 = bstart[q], j0 = mi_j0[q];
        const u32 d0 = depth0 ? depth0[q] : 0u;
        __syncthreads();                                       // the block before is out of LDS
        if (tid == 0) flag = 0;
        ccnt[tid] = 0; ccnt[tid + NT] = 0; cmsk[tid] = 0; cmsk[tid + NT] = 0;
        __syncthreads();
        u32 mask = 0;
        {   // the entries, then the words of all their window pairs, then shifts and stores (see sp_gather_pairs)
            u64 ee[EPT];
#pragma unroll
            for (int r = 0; r < EPT; r++) { const u32 x = (u32)r * NT + tid; ee[r] = x < m ? blue[b0 + x] : 0ull; }
            SpTriple t3[EPT];
            bool live[EPT];
#pragma unroll
            for (int r = 0; r < EPT; r++) {
                const u32 x = (u32)r * NT + tid;
                const u64 pos = (ee[r] >> 4) + (u64)d0 * (2 * SP_WIN);
                live[r] = x < m && pos < S;
                if (live[r]) t3[r] = sp_fetch3(spn, pos);
            }
#pragma unroll
            for (int r = 0; r < EPT; r++) {
                const u32 x = (u32)r * NT + tid;
                if (x < m) {
                    u64 w = 0ull, v = 0ull;
                    if (live[r]) sp_pair(t3[r], &w, &v);
                    se[x] = ee[r]; sw[x] = w; sx[x] = v; mask |= 1u << (ee[r] & 15);
                }
            }
        }
        if (mask) atomicOr(&flag, mask);
        __syncthreads();
        if ((flag & (flag - 1)) == 0 || (u64)d0 * (2 * SP_WIN) >= S + 2 * SP_WIN) {   // one BWT symbol (or nothing left to compare): any order is the order
            for (u32 x = tid; x < m; x += NT) mchar[j0 + x] = (u8)(se[x] & 15);
            if (tid == 0) { if (done) done[q] = 1; if (consume) consume[q] = 0; }
            continue;
        }
        // the sample, sorted by (first window, second window), one row per thread
        u64 vw, vx;
        { const u32 i = (u32)(((u64)tid * m) >> 8); vw = sw[i]; vx = sx[i]; }
#pragma unroll
        for (int lk = 1; lk <= 8; lk++) {
            const u32 kk = 1u << lk;
#pragma unroll
            for (int lj = lk - 1; lj >= 0; lj--) {
                const u32 d = 1u << lj;
                u64 qw, qx;
                if (lj < 6) { qw = __shfl_xor(vw, (int)d, 64); qx = __shfl_xor(vx, (int)d, 64); }
                else { pw[tid] = vw; px[tid] = vx; __syncthreads(); qw = pw[tid ^ d]; qx = px[tid ^ d]; __syncthreads(); }
                const bool take_min = ((tid & d) == 0) == ((tid & kk) == 0);
                const bool pless = qw != vw ? qw < vw : qx < vx;
                if (take_min == pless) { vw = qw; vx = qx; }
            }
        }
        pw[tid] = vw; px[tid] = vx;
        __syncthreads();
        const bool head = tid == 0 || pw[tid - 1] != vw || px[tid - 1] != vx;
        const u64 bm = __ballot(head);
        if (lane == 0) wtmp[w] = (u32)__popcll(bm);
        __syncthreads();
        u32 sbase = 0, U = 0;
#pragma unroll
        for (u32 i = 0; i < DEBWT_WAVES; i++) { const u32 c = wtmp[i]; sbase += i < w ? c : 0u; U += c; }
        const u32 sidx = sbase + (u32)__popcll(bm & lanemask_lt());
        if (head && sidx < NS - 1) { pw[sidx] = vw; px[sidx] = vx; }   // sidx <= tid: the slots of later threads are untouched
        if (U > NS - 1) U = NS - 1;
        __syncthreads();
        // class and place in the class of every row
        u32 cr[EPT];
#pragma unroll
        for (int r = 0; r < EPT; r++) {
            const u32 x = (u32)r * NT + tid;
            cr[r] = 0xFFFFFFFFu;
            if (x < m) {
                const u64 rw = sw[x], rx = sx[x];
                u32 lo = 0;                                    // splitters below the row
#pragma unroll
                for (u32 step = NS / 2; step; step >>= 1) {
                    const u32 c = lo + step;
                    if (c <= U) {
                        const u64 cw = pw[c - 1];
                        if (cw != rw ? cw < rw : px[c - 1] < rx) lo = c;
                    }
                }
                const u32 cls = 2u * lo + ((lo < U && pw[lo] == rw && px[lo] == rx) ? 1u : 0u);
                cr[r] = (cls << 16) | atomicAdd(&ccnt[cls], 1u);
                atomicOr(&cmsk[cls], 1u << (se[x] & 15));
            }
        }
        __syncthreads();
        // class starts; classes that still need sorting (several rows, several symbols) become blocks of the queue
        const u32 v0 = ccnt[2 * tid], v1 = ccnt[2 * tid + 1];
        const u32 k0 = cmsk[2 * tid], k1 = cmsk[2 * tid + 1];
        const bool u0 = v0 > 1 && (k0 & (k0 - 1)) != 0, u1 = v1 > 1 && (k1 & (k1 - 1)) != 0;
        u32 total, qtot;
        const u32 cbase = block_scan_excl(v0 + v1, wtmp, &total);
        const u32 qbase_ = block_scan_excl((u0 ? 1u : 0u) + (u1 ? 1u : 0u), wtmp, &qtot);
        const bool too_big = (u0 && v0 > (u32)SPLIT) || (u1 && v1 > (u32)SPLIT);
        if (tid == 0) sub_base = qtot ? atomicAdd(sub.count, qtot) : 0u;
        const bool give_up = __syncthreads_or((int)too_big) != 0;
        if (give_up || (qtot && (u64)sub_base + qtot > (u64)sub.cap)) continue;     // k_blue_refine's (nothing was moved)
        ccnt[2 * tid] = cbase; ccnt[2 * tid + 1] = cbase + v0;
        if (u0) {
            const u32 e = sub_base + qbase_;
            sub.start[e] = b0 + cbase; sub.j0[e] = j0 + cbase; sub.depth[e] = d0; sub.freq[e] = v0;
        }
        if (u1) {
            const u32 e = sub_base + qbase_ + (u0 ? 1u : 0u);
            sub.start[e] = b0 + cbase + v0; sub.j0[e] = j0 + cbase + v0; sub.depth[e] = d0 + 1u; sub.freq[e] = v1;   // 42 more symbols are equal
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < EPT; r++)
            if (cr[r] != 0xFFFFFFFFu) {
                const u32 x = (u32)r * NT + tid;
                const u32 pos = ccnt[cr[r] >> 16] + (cr[r] & 0xFFFFu);
                const u64 e = se[x];
                blue[b0 + pos] = e;
                mchar[j0 + pos] = (u8)(e & 15);
            }
        if (tid == 0) { if (done) done[q] = 1; if (consume) consume[q] = 0; }
      }
    }
}

__global__ void k_copy_u32(const u32 *__restrict__ src, u32 *__restrict__ dst) { *dst = *src; }

#define BLUE_LDS_CAP 2048
#ifndef LS_QUEUE_CAP
#define LS_QUEUE_CAP 512               // largest range the split of a large block queues for the LDS kernels; a larger one is split again
#endif                                 // (measured on distribution R at 3.1 Gbp: blue stage 102 ms with 2048, 87 with 512, 97 with 256, 104 with 128:
                                       //  the 2048-row kernel holds two workgroups per CU, the 512-row one eight waves per SIMD)
static_assert(LS_QUEUE_CAP <= BLUE_WAVE_CAP, "the queue is drained by the kernels of up to BLUE_WAVE_CAP rows");

// Blocks above BLUE_LDS_CAP rows: one level of sample sort in HBM, all pending blocks in the same five launches.
// Per block: the next 42 SP symbols (two windows, `depth` pairs in) of `ns` evenly spaced rows are sorted by one
// workgroup, nb - 1 of them become splitters, every row finds its range by bisection -- below a splitter, or equal
// to it: the rows that tie with a splitter form a range of their own -- and is moved there; ranges of <= BLUE_LDS_CAP
// rows are queued as blocks of their own for the LDS kernels, larger ones go back to the host: split again, one pair
// of windows deeper when it is a range of ties.
#ifndef LS_SAMPLES
#define LS_SAMPLES 4096
#endif
#ifndef LS_BIN_ROWS
#define LS_BIN_ROWS 384                // rows per range the number of ranges of a block aims at
#endif
#ifndef LS_OVERSAMPLE
#define LS_OVERSAMPLE 8u               // samples per range
#endif
#define LS_MAXBINS 1024                // splitters + 1
#define LS_MAXR (2 * LS_MAXBINS)       // ranges: below splitter 0, equal to it, between 0 and 1, equal to 1, ...
#define LS_WG_ROWS 1024u               // rows of ONE block a workgroup of the row kernels takes (four per thread)
struct LsBlock { u64 b0, j0, row0; u32 m, nb, ns, wg0, t0, depth, pivot, sidx; };  // row0: first scratch row; wg0: first workgroup of
                                                                        // LS_WG_ROWS rows; t0: first splitter slot (ranges
                                                                        // from 2 t0); depth: window pairs already equal;
                                                                        // pivot: 1 = keys of a pivot round (ls_row_key);
                                                                        // sidx: which of the batch's blocks of > 256 samples
struct LsOver { u32 blk, st, cnt, ties, adv; };                         // a range above BLUE_LDS_CAP rows: block of the
                                                                        // batch, first row, rows, 1 = rows tie with a
                                                                        // splitter, window pairs they are known to share
// Pivot round, for rows that keep tying (a run of one symbol or a tandem repeat ties for as long as it lasts, and a
// window round sheds only 42 SP symbols of it): every row is compared with ONE row of the block, the pivot, pair of
// windows by pair, until they differ -- after t pairs.  Rows below the pivot order by t ascending (the one that leaves
// the common prefix first is the smaller), rows above it by t descending, so key = t | TCAP | 2 TCAP - t stands in for
// the windows of a window round; rows with the same key share t more pairs (not t + 1: pair t differs from the
// pivot's, not necessarily among them).  All rows of a run that outlast the pivot's leave with the pivot's own t in one
// range: with the pivot in the middle of the block a round halves what is left of a periodic stretch.
#define LS_TCAP 65536u
#define LS_PICK_PAIRS 48u
__device__ __forceinline__ u32 ls_pivot_adv(u64 key) {
    return key < LS_TCAP ? (u32)key : (key == LS_TCAP ? LS_TCAP : (u32)(2 * LS_TCAP - key));
}
struct LargeSplit {
    const LsBlock *blk; u32 nblk;
    u32 *wgblk;                       // per workgroup of the row kernels: its block (written by k_ls_splitters)
    u64 *en;                          // per row: entry (copy)
    u32 *bin;                         // per row: its range
    u64 *spl_w, *spl_x;               // per block: nb splitter slots from t0
    u32 *cnt, *start, *cur;           // per block: 2 nb ranges from 2 t0
    u32 *res;                         // per block: 1 = sub-block table full (nothing queued, nothing moved)
    u32 *piv;                         // per block of a pivot round: its pivot row (chosen by k_ls_splitters)
    u64 *smp_w, *smp_x;               // per block of > 256 samples (LsBlock::sidx): LS_SAMPLES sample keys (k_ls_sample_keys)
    const u32 *bigidx;                // those blocks (sidx -> block of the batch)
    LsOver *over; u32 *nover;         // oversize ranges of the batch (at most rows / LS_QUEUE_CAP of them), their number
};
// The key a round sorts row `i` of block B by: its next pair of SP windows (window round), or how far it follows the
// block's pivot row `piv` (pivot round; second word 0).  e = the row's blue entry.
__device__ __forceinline__ void ls_row_key(const u64 *__restrict__ blue, const u64 *__restrict__ spn, u64 S, const LsBlock &B, u32 piv, u32 i,
                                           u64 e, u64 *kw, u64 *kx, u32 tcap = LS_TCAP) {
    const u64 pos = (e >> 4) + (u64)B.depth * (2 * SP_WIN);
    if (B.pivot) {
        const u64 ppos = (blue[B.b0 + piv] >> 4) + (u64)B.depth * (2 * SP_WIN);
        u64 key = LS_TCAP;                                   // equal to the pivot for as long as we look
        for (u32 t = 0; t < tcap && i != piv; t++) {
            const u64 a = pos + (u64)t * (2 * SP_WIN), b = ppos + (u64)t * (2 * SP_WIN);
            const bool la = a < S, lb = b < S;
            if (!la && !lb) break;                           // both behind the end: zeros from here on
            const u64 aw = la ? sp_window(spn, a) : 0ull, ax = la ? sp_window(spn, a + SP_WIN) : 0ull;
            const u64 bw = lb ? sp_window(spn, b) : 0ull, bx = lb ? sp_window(spn, b + SP_WIN) : 0ull;
            if (aw != bw || ax != bx) {
                key = (aw != bw ? aw < bw : ax < bx) ? (u64)t : (u64)(2 * LS_TCAP - t);
                break;
            }
        }
        *kw = key; *kx = 0ull;
        return;
    }
    const bool live = pos < S;
    *kw = live ? sp_window(spn, pos) : 0ull;
    *kx = live ? sp_window(spn, pos + SP_WIN) : 0ull;
}
// One workgroup per block: the keys of `ns` evenly spaced rows are fetched and sorted, nb - 1 of them become the block's
// splitters; the block's range counters are cleared and its workgroups of the row kernels are given its index.  NS:
// the sample capacity of the instance (LDS); a block whose samples belong to the other instance is skipped -- most large
// blocks are a few thousand rows with 64 samples, and 64 KB of LDS per workgroup for those leaves two workgroups per CU.
// The instance for blocks of more than 256 samples runs in two launches with k_ls_sample_keys between them (PHASE 1: the
// pivot, the counters, the workgroup table; PHASE 2: the sort of the sample keys that kernel fetched): one workgroup
// walking 4096 samples of a block of 10^6 rows of one long run, four per thread one after the other, took milliseconds per
// round while the rest of the chip idled.
template <u32 NS, u32 T, int PHASE = 0>
__global__ __launch_bounds__(T) void k_ls_splitters(const u64 *__restrict__ blue, const u64 *__restrict__ spn, u64 S, LargeSplit ls) {
    __shared__ u64 sw[PHASE == 1 ? 1 : NS], sx[PHASE == 1 ? 1 : NS];
    __shared__ u32 s_cand, s_tmax, s_cnt, s_next;
    const u32 bi = NS < LS_SAMPLES ? blockIdx.x : ls.bigidx[blockIdx.x];   // (the large instance runs over the list of its blocks)
    const LsBlock B = ls.blk[bi];
    const u32 tid = threadIdx.x, ns = B.ns;
    if ((NS < LS_SAMPLES) != (ns <= 256u)) return;
    u32 piv = B.m >> 1;
    if (B.pivot && PHASE != 2) {
        // The pivot of a pivot round: a row that follows the block's periodic stretch for LONG.  The rows that outlast the
        // pivot all leave with the pivot's own t, in one range that needs another round; the rows the pivot outlasts leave
        // with their own t and are done -- so a pivot at the median run length halves the stretch per round (the middle
        // row: 4-5 rounds for the ~7,700 rows of an average tie range of distribution R at 30 Gbp), one near the maximum
        // finishes it in a round or two.  64 evenly spaced rows are measured against the candidate; when two or more of them share
        // the largest t they (in all likelihood) outlast it and the first of them becomes the candidate -- each step halves
        // the rows that still outlast; three steps leave an eighth (every step walks the stretch once more: more steps cost a
        // 7 Mbp collection of nothing but runs and tandem repeats more than they save it).  Any row is a valid pivot: this only picks.
        if (tid == 0) s_cand = piv;
        __syncthreads();
        for (int it = 0; it < 3; it++) {
            if (tid == 0) { s_tmax = 0; s_cnt = 0; s_next = ~0u; }
            __syncthreads();
            const u32 cand = s_cand;
            u32 r = 0, t = 0;
            bool mine = false;
            if (tid < 64u) {
                r = (u32)(((u64)tid * B.m) >> 6);
                if (r != cand) {
                    u64 kw, kx;
                    // (the choice looks LS_PICK_PAIRS pairs far: a stretch of 30,000 equal symbols is 700 pairs, walked one
                    // dependent gather after the other -- rows that follow the candidate that far count as outlasting it)
                    ls_row_key(blue, spn, S, B, cand, r, blue[B.b0 + r], &kw, &kx, LS_PICK_PAIRS);
                    t = ls_pivot_adv(kw);
                    mine = true;
                    atomicMax(&s_tmax, t);
                }
            }
            __syncthreads();
            if (mine && t == s_tmax) { atomicAdd(&s_cnt, 1u); atomicMin(&s_next, r); }
            __syncthreads();
            if (s_cnt < 2u) break;                           // (uniform: read behind a barrier, reset behind the next)
            if (tid == 0) s_cand = s_next;
            __syncthreads();
        }
        piv = s_cand;
        if (tid == 0) ls.piv[bi] = piv;
    }
    if (PHASE != 2) {
        for (u32 i = tid; i < 2 * B.nb; i += T) { ls.cnt[2 * (size_t)B.t0 + i] = 0; ls.cur[2 * (size_t)B.t0 + i] = 0; }
        const u32 nwg = (B.m + LS_WG_ROWS - 1) / LS_WG_ROWS;
        for (u32 i = tid; i < nwg; i += T) ls.wgblk[B.wg0 + i] = bi;
    }
    if (PHASE == 1) return;
    if (PHASE == 2) {
        for (u32 i = tid; i < ns; i += T) { sw[i] = ls.smp_w[(size_t)B.sidx * LS_SAMPLES + i]; sx[i] = ls.smp_x[(size_t)B.sidx * LS_SAMPLES + i]; }
    } else {
        for (u32 i = tid; i < ns; i += T) {
            const u32 r = (u32)(((u64)i * B.m) / ns);
            ls_row_key(blue, spn, S, B, piv, r, blue[B.b0 + r], &sw[i], &sx[i]);
        }
    }
    __syncthreads();
    for (u32 kk = 2; kk <= ns; kk <<= 1)
        for (u32 jj = kk >> 1; jj > 0; jj >>= 1) {
            for (u32 t = tid; t < ns / 2; t += T) {
                const u32 i = ((t & ~(jj - 1)) << 1) | (t & (jj - 1)), l = i | jj;
                const u64 wi = sw[i], wl = sw[l], xi = sx[i], xl = sx[l];
                const bool l_less = wl != wi ? wl < wi : xl < xi;
                const bool i_less = wl != wi ? wi < wl : xi < xl;
                if (((i & kk) == 0) ? l_less : i_less) { sw[i] = wl; sw[l] = wi; sx[i] = xl; sx[l] = xi; }
            }
            __syncthreads();
        }
    // splitter b = sorted sample (b + 1) * ns / nb - 1, b < nb - 1
    for (u32 b = tid; b + 1 < B.nb; b += T) {
        const u32 i = (b + 1) * (ns / B.nb) - 1;
        ls.spl_w[(size_t)B.t0 + b] = sw[i]; ls.spl_x[(size_t)B.t0 + b] = sx[i];
    }
}
// the sample keys of the blocks of more than 256 samples: 256 samples of one block per workgroup (grid: such blocks x LS_SAMPLES / 256)
__global__ __launch_bounds__(256) void k_ls_sample_keys(const u64 *__restrict__ blue, const u64 *__restrict__ spn, u64 S, LargeSplit ls) {
    const u32 bi = ls.bigidx[blockIdx.x];
    const LsBlock B = ls.blk[bi];
    const u32 i = blockIdx.y * 256u + threadIdx.x;
    if (i >= B.ns) return;
    const u32 piv = B.pivot ? ls.piv[bi] : 0u;
    const u32 r = (u32)(((u64)i * B.m) / B.ns);
    u64 kw, kx;
    ls_row_key(blue, spn, S, B, piv, r, blue[B.b0 + r], &kw, &kx);
    ls.smp_w[(size_t)B.sidx * LS_SAMPLES + i] = kw; ls.smp_x[(size_t)B.sidx * LS_SAMPLES + i] = kx;
}
// A workgroup takes LS_WG_ROWS consecutive rows of one block, four per thread: their keys are fetched (the gathers of a
// thread's four rows are in flight together), every row finds its range by bisection over the block's splitters, staged in
// LDS -- below a splitter, or equal to it: the rows that tie with a splitter form a range of their own.  The rows' ranges
// are counted in LDS first and the global range counters receive one atomic per range and workgroup (issued by different
// threads, not one after the other): rows of a low-complexity block crowd into a handful of ranges, and one device-scope
// atomic per row serialises on them.  (Round 5: this kernel was two -- windows to HBM, then the bisection over splitters
// in global memory by workgroups of 256 rows that found their block by a binary search over the batch's blocks: 17 + ~8
// dependent loads in front of every 256 rows' work, 233 ms per 30 Gbp build of distribution R.)
__global__ __launch_bounds__(256) void k_ls_bin(const u64 *__restrict__ blue, const u64 *__restrict__ spn, u64 S, LargeSplit ls) {
    __shared__ u64 spw[LS_MAXBINS], spx[LS_MAXBINS];
    __shared__ u32 h[LS_MAXR];
    const u32 bi = ls.wgblk[blockIdx.x];
    const LsBlock B = ls.blk[bi];
    const u32 nr = 2 * B.nb, tid = threadIdx.x;
    for (u32 r = tid; r < nr; r += 256) h[r] = 0;
    for (u32 b = tid; b + 1 < B.nb; b += 256) { spw[b] = ls.spl_w[(size_t)B.t0 + b]; spx[b] = ls.spl_x[(size_t)B.t0 + b]; }
    const u32 i0 = (blockIdx.x - B.wg0) * LS_WG_ROWS + tid;
    const u32 piv = B.pivot ? ls.piv[bi] : 0u;
    u64 e[4], w[4], x[4];
#pragma unroll
    for (int j = 0; j < 4; j++) e[j] = i0 + j * 256u < B.m ? blue[B.b0 + i0 + j * 256u] : 0ull;
    if (B.pivot) {
        // the four rows of a thread follow the pivot in step: their gathers of a pair are in flight together and the pivot's
        // windows are fetched once per pair (four walks one after the other made a block of 10^6 rows of one long run -- too
        // few workgroups to fill the chip -- four times as slow as one row per thread had been)
        const u64 ppos = (blue[B.b0 + piv] >> 4) + (u64)B.depth * (2 * SP_WIN);
        u64 pos[4];
        bool live[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u32 i = i0 + j * 256u;
            pos[j] = (e[j] >> 4) + (u64)B.depth * (2 * SP_WIN);
            live[j] = i < B.m && i != piv;
            w[j] = LS_TCAP; x[j] = 0ull;                      // equal to the pivot for as long as we look
        }
        for (u32 t = 0; t < LS_TCAP && (live[0] || live[1] || live[2] || live[3]); t++) {
            const u64 b = ppos + (u64)t * (2 * SP_WIN);
            const bool lb = b < S;
            const u64 bw = lb ? sp_window(spn, b) : 0ull, bx = lb ? sp_window(spn, b + SP_WIN) : 0ull;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (!live[j]) continue;
                const u64 a = pos[j] + (u64)t * (2 * SP_WIN);
                const bool la = a < S;
                if (!la && !lb) { live[j] = false; continue; }   // both behind the end: zeros from here on
                const u64 aw = la ? sp_window(spn, a) : 0ull, ax = la ? sp_window(spn, a + SP_WIN) : 0ull;
                if (aw != bw || ax != bx) {
                    w[j] = (aw != bw ? aw < bw : ax < bx) ? (u64)t : (u64)(2 * LS_TCAP - t);
                    live[j] = false;
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            w[j] = x[j] = 0ull;
            if (i0 + j * 256u < B.m) ls_row_key(blue, spn, S, B, piv, i0 + j * 256u, e[j], &w[j], &x[j]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const u32 i = i0 + j * 256u;
        if (i >= B.m) continue;
        u32 lo = 0, hi = B.nb - 1;                           // number of splitters below the row's key
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            const u64 sw = spw[mid], sx = spx[mid];
            if (sw != w[j] ? sw < w[j] : sx < x[j]) lo = mid + 1; else hi = mid;
        }
        const bool tie = lo + 1 < B.nb && spw[lo] == w[j] && spx[lo] == x[j];
        const u32 r = 2 * lo + (tie ? 1u : 0u);
        ls.bin[B.row0 + i] = r;
        ls.en[B.row0 + i] = e[j];
        atomicAdd(&h[r], 1u);
    }
    __syncthreads();
    for (u32 r = tid; r < nr; r += 256)
        if (h[r]) atomicAdd(&ls.cnt[2 * (size_t)B.t0 + r], h[r]);
}
// one workgroup per block: range starts; ranges of <= LS_QUEUE_CAP rows become sub-blocks, larger ones are reported.  NB:
// the splitter capacity of the instance (threads; two ranges per thread); blocks of the other instance are skipped.
template <u32 NB>
__global__ __launch_bounds__(NB) void k_ls_plan(LargeSplit ls, BlueSub sub) {
    __shared__ u32 part[NB];
    __shared__ u32 nsub, base, full;
    const LsBlock B = ls.blk[blockIdx.x];
    if ((NB < LS_MAXBINS) != (B.nb <= 64u)) return;
    const u32 tid = threadIdx.x;
    const u32 nr = 2 * B.nb;
    const size_t r0 = 2 * (size_t)B.t0;
    u32 c[2];
    for (int h = 0; h < 2; h++) c[h] = 2 * tid + h < nr ? ls.cnt[r0 + 2 * tid + h] : 0u;
    part[tid] = c[0] + c[1];
    if (tid == 0) { nsub = 0; full = 0; }
    __syncthreads();
    for (u32 d = 1; d < NB; d <<= 1) {
        const u32 v = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    u32 st[2];
    st[0] = part[tid] - c[0] - c[1]; st[1] = st[0] + c[0];
    u32 my[2] = {0, 0};
    for (int h = 0; h < 2; h++) {
        if (2 * tid + h < nr) ls.start[r0 + 2 * tid + h] = st[h];
        if (c[h] >= 1 && c[h] <= LS_QUEUE_CAP) my[h] = atomicAdd(&nsub, 1u);
    }
    __syncthreads();
    if (tid == 0) {
        base = atomicAdd(sub.count, nsub);
        full = (u64)base + nsub > (u64)sub.cap ? 1u : 0u;
        ls.res[blockIdx.x] = full;
    }
    __syncthreads();
    // pairs of windows the rows of the tie range (h = 1) share beyond B.depth: those of a pivot round the pairs their
    // key counts, those of a window round the pair just compared (the host checks that a deeper pair exists; a
    // sub-block starts conservatively at B.depth there, as before)
    const u32 adv_tie = B.pivot ? (tid + 1 < B.nb ? ls_pivot_adv(ls.spl_w[(size_t)B.t0 + tid]) : 0u) : 1u;
    // ... and the rows of a range BETWEEN two splitters of a pivot round (h = 0): keys k with lower < k < upper.  Below the
    // pivot (upper <= TCAP) k is the row's t, so every row follows the pivot -- and hence every other row of the range --
    // for more than `lower` pairs; above it (lower >= TCAP) k = 2 TCAP - t, so for more than 2 TCAP - upper.  Without this
    // the rows of a homopolymer block that left the pivot after 90..100 pairs were queued at the block's depth and the
    // LDS kernels walked those 90 pairs again, a round per pair.
    u32 adv_btw = 0;
    if (B.pivot) {
        const bool has_lo = tid >= 1, has_up = tid + 1 < B.nb;
        const u64 lower = has_lo ? ls.spl_w[(size_t)B.t0 + tid - 1] : 0ull, upper = has_up ? ls.spl_w[(size_t)B.t0 + tid] : 0ull;
        if (has_up && upper <= LS_TCAP) adv_btw = has_lo ? (u32)lower + 1u : 0u;
        else if (has_lo && lower >= LS_TCAP) adv_btw = has_up ? (u32)(2 * LS_TCAP - upper) + 1u : 0u;
    }
    for (int h = 0; h < 2; h++) {
        const u32 adv = B.pivot ? (h ? adv_tie : adv_btw) : 0u;
        if (c[h] >= 1 && c[h] <= LS_QUEUE_CAP && !full) {
            const u32 e = base + my[h];
            sub.start[e] = B.b0 + st[h]; sub.freq[e] = c[h]; sub.j0[e] = B.j0 + st[h];
            sub.depth[e] = B.depth + adv;
        }
        if (c[h] > LS_QUEUE_CAP && !full)                                    // h = 1: a range of ties
            ls.over[atomicAdd(ls.nover, 1u)] = LsOver{blockIdx.x, st[h], c[h], (u32)h, h ? adv_tie : adv_btw};
    }
}
__global__ __launch_bounds__(256) void k_ls_scatter(u64 *__restrict__ blue, LargeSplit ls) {
    __shared__ u32 h[LS_MAXR];                               // rows of the workgroup per range, then their first slot
    const u32 bi = ls.wgblk[blockIdx.x];
    const LsBlock B = ls.blk[bi];
    if (ls.res[bi]) return;                                  // nothing was queued: the rows stay for the network
    const u32 nr = 2 * B.nb, tid = threadIdx.x;
    const size_t r0 = 2 * (size_t)B.t0;
    for (u32 r = tid; r < nr; r += 256) h[r] = 0;
    __syncthreads();
    const u32 i0 = (blockIdx.x - B.wg0) * LS_WG_ROWS + tid;
    u32 b[4], mine[4];
    u64 e[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bool valid = i0 + j * 256u < B.m;
        b[j] = valid ? ls.bin[B.row0 + i0 + j * 256u] : 0u;
        e[j] = valid ? ls.en[B.row0 + i0 + j * 256u] : 0ull;
        mine[j] = valid ? atomicAdd(&h[b[j]], 1u) : 0u;      // rank among the workgroup's rows of the range
    }
    __syncthreads();
    for (u32 r = tid; r < nr; r += 256) {
        const u32 c = h[r];
        if (c) h[r] = ls.start[r0 + r] + atomicAdd(&ls.cur[r0 + r], c);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (i0 + j * 256u < B.m) blue[B.b0 + h[b[j]] + mine[j]] = e[j];
}

// descriptors of the large blocks (context-wide block ids in large_q) in one gather
__global__ void k_large_gather(const u32 *__restrict__ large_q, u64 nlarge, const u32 *__restrict__ blk_freq,
                               const u64 *__restrict__ blk_start, const u64 *__restrict__ blk_j0, u64 *__restrict__ out) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nlarge) return;
    const u32 q = large_q[t];
    out[3 * t] = blk_start[q]; out[3 * t + 1] = blk_j0[q]; out[3 * t + 2] = blk_freq[q];
}

// large blocks: bitonic network in global memory, one launch per compare-exchange distance
__global__ void k_large_load(const u64 *__restrict__ blue, u64 b0, u32 m, u64 P, const u64 *__restrict__ spn,
                             u64 *__restrict__ k0, u64 *__restrict__ en) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P) return;
    if (t < m) { u64 e = blue[b0 + t]; en[t] = e; k0[t] = sp_window(spn, e >> 4); }
    else { en[t] = ~0ull; k0[t] = ~0ull; }
}
__global__ void k_large_step(u64 *__restrict__ k0, u64 *__restrict__ en, u64 P, u64 kk, u64 jj,
                             const u64 *__restrict__ spn, u64 S) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (P >> 1)) return;
    u64 i = ((t & ~(jj - 1)) << 1) | (t & (jj - 1));
    u64 l = i | jj;
    bool up = (i & kk) == 0;
    u64 ka = k0[i], kb = k0[l], ea = en[i], eb = en[l];
    bool lt;
    if (ka != kb) lt = kb < ka;
    else if (ea == ~0ull || eb == ~0ull) lt = eb < ea;
    else lt = sp_less_deep(spn, S, eb, ea);
    if (lt == up) { k0[i] = kb; k0[l] = ka; en[i] = eb; en[l] = ea; }
}
__global__ void k_large_store(u64 *__restrict__ blue, u64 b0, u32 m, u64 j0, const u64 *__restrict__ en,
                              u8 *__restrict__ mchar) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    u64 e = en[t];
    blue[b0 + t] = e;
    mchar[j0 + t] = (u8)(e & 15);
}

// ---------------------------------------------------------------------------------------------------
// assembly (insertCase3, src/insertCase3.c:56-104): rows = node-instance symbols in key order with the
// special suffixes merged in at their rows; 2 bits per row, '#'/'$' rows stored as 3 and recorded

// A workgroup covers 256 words = 8192 rows; the row symbols it needs (a contiguous stretch of mchar: the rows minus
// the special suffixes before them) are staged in LDS with 16-byte loads; a word whose 32 rows hold no special suffix
// and no separator symbol (nearly all of them) is packed from 8 LDS words with byte-align and shift/or steps, the
// others row by row.
#define ASM_ROWS (DEBWT_BLOCK * 32)
__global__ __launch_bounds__(DEBWT_BLOCK) void k_assemble(const u8 *__restrict__ mchar, u64 M,
                                                           const u64 *__restrict__ sprow,
                                                           const u8 *__restrict__ spchr, u64 NS, u64 n,
                                                           u64 *__restrict__ bwt, u32 *__restrict__ hmask,
                                                           u64 *__restrict__ dollar_row, u8 *__restrict__ rowsym,
                                                           u64 block0, u64 *__restrict__ hlist,
                                                           unsigned long long *__restrict__ hcount) {
    // hlist: the rows that carry '#' are appended there (any order; the host sorts the nrec - 1 of them) and hmask is not
    // written -- a collection of few records then needs no pass over n / 32 mask words to find them.  nullptr: hmask.
    // block0: first 8192-row block of this launch (a build that hands finished row ranges to the host as it goes
    // assembles them range by range)
    __shared__ u32 sm[ASM_ROWS / 4 + 12];
    __shared__ u64 sblk, send;
    const u64 blk = block0 + blockIdx.x;
    const u64 R0 = blk * ASM_ROWS;
    if (threadIdx.x == 0) sblk = lower_bound_dev<u64>(sprow, 0, NS, R0);       // special rows before the block
    if (threadIdx.x == 64) send = lower_bound_dev<u64>(sprow, 0, NS, R0 + ASM_ROWS);   // ... and before the next: a word bisects between
    __syncthreads();                                                           // the two (nearly always nothing lies there)
    const u64 jb = R0 - sblk, jal = jb & ~15ull;                                 // first instance of the block, 16-aligned
    // instances [jal, jal + ASM_ROWS + 16): mchar is padded by 64 bytes behind M
    for (u32 i = threadIdx.x; i < ASM_ROWS / 16 + 2; i += DEBWT_BLOCK) {
        const u64 a = jal + (u64)i * 16;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (a < M + 48) v = *reinterpret_cast<const uint4 *>(mchar + a);
        sm[4 * i] = v.x; sm[4 * i + 1] = v.y; sm[4 * i + 2] = v.z; sm[4 * i + 3] = v.w;
    }
    __syncthreads();
    u64 w = blk * blockDim.x + threadIdx.x;
    u64 nw = (n + 31) >> 5;
    if (w >= nw) return;
    u64 r0 = w << 5;
    u64 s = lower_bound_dev<u64>(sprow, sblk, send, r0);          // special rows before r0
    u64 j = r0 - s;
    u64 next_special = s < NS ? sprow[s] : ~0ull;
    u32 lim = (n - r0) < 32 ? (u32)(n - r0) : 32u;
    if (lim == 32 && next_special >= r0 + 32 && !rowsym) {
        // 32 instances from LDS: 9 words, byte-aligned to the first instance
        const u32 off = (u32)(j - jal);
        const u32 wi = off >> 2, sh = off & 3u;
        u32 q[9];
#pragma unroll
        for (int t = 0; t < 9; t++) q[t] = sm[wi + t];
        u32 bad = 0;
        u64 word = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const u32 x = __builtin_amdgcn_alignbyte(q[t + 1], q[t], sh);        // 4 symbols, first in the low byte
            bad |= x;
            const u32 p = ((x << 6) | (x >> 4) | (x >> 14) | (x >> 24)) & 0xFFu;  // 2 bits each, first symbol on top
            word |= (u64)p << (56 - 8 * t);
        }
        if (!(bad & 0xFCFCFCFCu)) {                                // no '#' / '$' among them
            bwt[w] = word;
            if (!hlist) hmask[w] = 0;
            return;
        }
    }
    u64 word = 0;
    u32 hm = 0;
    for (u32 t = 0; t < lim; t++) {
        u64 r = r0 + t;
        u32 c;
        if (r == next_special) { c = spchr[s]; s++; next_special = s < NS ? sprow[s] : ~0ull; }
        else c = mchar[j++];
        if (rowsym) rowsym[r] = (u8)c;
        if (c == 4) { hm |= 1u << t; c = 3; }
        else if (c == 5) { *dollar_row = r; c = 3; }
        word |= (u64)c << ((31 - t) << 1);
    }
    bwt[w] = word;
    if (!hlist) hmask[w] = hm;
    else if (hm) {
        unsigned long long o = atomicAdd(hcount, (unsigned long long)__popc(hm));
        while (hm) { const u32 t = (u32)__ffs(hm) - 1u; hm &= hm - 1u; hlist[o++] = r0 + t; }
    }
}
struct HashRowsF {
    const u32 *hmask; u64 *hash_rows;
    __device__ u32 count(u64 w) const { return (u32)__popc(hmask[w]); }
    __device__ u32 recount(u64 w) const { return count(w); }
    __device__ void emit(u64 w, u32 off, u32 c) const {
        if (!c) return;
        u32 m = hmask[w];
        while (m) { u32 t = __ffs(m) - 1; m &= m - 1; hash_rows[off++] = (w << 5) + t; }
    }
};

// final concatenation of a sharded build: output word w = rows [32w, 32w + 32), taken from the parts that cover them
struct ConcatPart { u64 word_off, row_base, rows; };
__global__ void k_concat_rows(const u64 *__restrict__ parts, const ConcatPart *__restrict__ pp, u32 nparts, u64 n,
                              u64 *__restrict__ out) {
    const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= ((n + 31) >> 5)) return;
    u64 R = w << 5;
    const u64 Rend = R + 32 < n ? R + 32 : n;
    u32 lo = 0, hi = nparts;                                  // last part that starts at or before row R
    if (nparts <= 16u) {                                      // (the shards of one node: a count of compares on uniform loads)
        for (u32 p = 1; p < nparts; p++) lo += pp[p].row_base <= R ? 1u : 0u;
    } else
        while (lo + 1 < hi) { const u32 mid = (lo + hi) >> 1; if (pp[mid].row_base <= R) lo = mid; else hi = mid; }
    u64 word = 0;
    for (u32 p = lo; p < nparts && R < Rend; p++) {
        const ConcatPart P = pp[p];
        const u64 pe = P.row_base + P.rows;
        if (pe <= R) continue;
        const u64 l = R - P.row_base;                         // first local row wanted
        const u32 cnt = (u32)((pe < Rend ? pe : Rend) - R), sh = (u32)(l & 31) << 1;
        const u64 *pw = parts + P.word_off + (l >> 5);
        u64 v = pw[0] << sh;
        if (sh) v |= pw[1] >> (64 - sh);                      // 32 rows from local row l on, first on top
        if (cnt < 32) v &= ~(~0ull >> (2 * cnt));
        word |= v >> (2 * (u32)(R & 31));
        R += cnt;
    }
    out[w] = word;
}

// symbol census of the packed BWT (rows [0, n): codes 0..3, '#'/'$' rows count as 3): out4[c] += rows with code c
__global__ __launch_bounds__(DEBWT_BLOCK) void k_bwt_census(const u64 *__restrict__ bwt, u64 n, u64 *__restrict__ out4) {
    __shared__ u64 red[4][DEBWT_WAVES];
    const u64 nw = (n + 31) >> 5;
    u64 c1 = 0, c2 = 0, c3 = 0, rows = 0;
    for (u64 w = (u64)blockIdx.x * DEBWT_BLOCK + threadIdx.x; w < nw; w += (u64)gridDim.x * DEBWT_BLOCK) {
        u64 v = bwt[w];
        const u32 lim = (n - (w << 5)) < 32 ? (u32)(n - (w << 5)) : 32u;
        const u64 valid = lim == 32 ? 0x5555555555555555ull : (0x5555555555555555ull << (2 * (32 - lim)));
        const u64 lo = v & valid, hi = (v >> 1) & valid;          // row r of the word sits at bits 2*(31-r)
        c3 += (u64)__popcll(hi & lo); c2 += (u64)__popcll(hi & ~lo); c1 += (u64)__popcll(~hi & lo);
        rows += lim;
    }
    u64 v[4] = {rows - c1 - c2 - c3, c1, c2, c3};
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v[q] += __shfl_xor(v[q], d, 64);
        if (lane_id() == 0) red[q][threadIdx.x >> 6] = v[q];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        u64 t = 0;
        for (int w = 0; w < DEBWT_WAVES; w++) t += red[threadIdx.x][w];
        atomicAdd(&out4[threadIdx.x], t);
    }
}

// counts of equal adjacent keys -> (kmer left-aligned, count) pairs for debwt_kmer_count_sorted
struct KmerInfoF {
    const u64 *sk; u64 M; int k; u64 *kmers; u32 *first;
    __device__ u32 count(u64 j) const { return (j == 0 || sk[j - 1] != sk[j]) ? 1u : 0u; }
    __device__ u32 recount(u64 j) const { return count(j); }
    __device__ void emit(u64 j, u32 off, u32 c) const {
        if (c) { kmers[off] = sk[j] << (64 - 2 * k); first[off] = (u32)j; }
    }
};
