// common.h -- shared device helpers for the gfx950 deBWT kernels (wave64 throughout).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned char u8;

#define DEBWT_WAVE 64
#define DEBWT_BLOCK 256
#define DEBWT_WAVES (DEBWT_BLOCK / DEBWT_WAVE)

// ---- 2-bit packed text (reference layout: base j at bit 2*(31-(j&31)) of word j>>5) ---------------

// 64-bit window = 32 symbols starting at symbol i (reference `convert`, src/collect#$.c:243-251)
__device__ __forceinline__ u64 text_window(const u64 *__restrict__ words, u64 i) {
    u64 w = i >> 5;
    u32 sh = (u32)(i & 31) << 1;
    u64 a = words[w];
    if (sh == 0) return a;
    u64 b = words[w + 1];
    return (a << sh) | (b >> (64 - sh));
}
__device__ __forceinline__ u32 text_symbol(const u64 *__restrict__ words, u64 i) {
    return (u32)(words[i >> 5] >> (((u32)(31 - (i & 31))) << 1)) & 3u;
}

// separator bitmap: bit (i & 63) of word i >> 6 set when position i holds '#' or '$'.
// 64-bit window of bits [i, i+64), bit 0 = position i.
__device__ __forceinline__ u64 sep_window(const u64 *__restrict__ bits, u64 i) {
    u64 w = i >> 6;
    u32 sh = (u32)(i & 63);
    u64 a = bits[w];
    if (sh == 0) return a;
    u64 b = bits[w + 1];
    return (a >> sh) | (b << (64 - sh));
}
__device__ __forceinline__ bool sep_at(const u64 *__restrict__ bits, u64 i) {
    return (bits[i >> 6] >> (i & 63)) & 1ull;
}

// ---- searches ---------------------------------------------------------------------------------------

// first index in [lo, hi) with a[idx] >= key
template <typename T>
__device__ __forceinline__ u64 lower_bound_dev(const T *__restrict__ a, u64 lo, u64 hi, T key) {
    while (lo < hi) {
        u64 mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// first index in [lo, hi) with a[idx] > key
template <typename T>
__device__ __forceinline__ u64 upper_bound_dev(const T *__restrict__ a, u64 lo, u64 hi, T key) {
    while (lo < hi) {
        u64 mid = (lo + hi) >> 1;
        if (a[mid] <= key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global load and
// store of the wave (s_waitcnt vmcnt(0)) -- in a streaming kernel that exposes the full HBM store latency at every
// barrier and defeats prefetching.  Use only where no global data is exchanged between the waves of a workgroup.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Positions of a text word whose K-symbol window holds a separator.  sb: bit j = a separator at the word's first position
// + j (64 positions); returns bit t (t < 32) set when one of the positions t .. t + K - 1 is a separator: every set bit is
// smeared down over the K - 1 positions before it in five or six shift-or steps.  (The 32-iteration form `(sb >> t) & kmask`
// this replaces was if-converted by the compiler in every kernel that holds it -- ~190 64-bit operations per text word,
// executed whether or not a separator was near: profiles/r06_experiments.txt.)  Callers test `sb` with a wave-uniform branch.
__device__ __forceinline__ u32 sep_blocked(u64 sb, int K) {
    u64 x = sb;
    int r = K - 1;
    for (int s = 1; r > 0; s <<= 1) { const int take = s < r ? s : r; x |= x >> take; r -= take; }
    return (u32)x;
}
__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }     // uniform: a scalar branch, never if-converted

// ---- wave / block scans ----------------------------------------------------------------------------

__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ u64 lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// inclusive scan across the 64 lanes of a wave: DPP row shifts inside the rows of 16 lanes, then the row totals broadcast
// (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) -- six adds whose operands arrive by the data-parallel
// primitives of the VALU, where the __shfl_up form went through the LDS crossbar six times (ds_bpermute + wait + select)
__device__ __forceinline__ u32 wave_scan_incl(u32 v) {
#define DEBWT_DPP_ADD(ctrl, rmask) v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, false)
    DEBWT_DPP_ADD(0x111, 0xf);      // row_shr:1
    DEBWT_DPP_ADD(0x112, 0xf);      // row_shr:2
    DEBWT_DPP_ADD(0x114, 0xf);      // row_shr:4
    DEBWT_DPP_ADD(0x118, 0xf);      // row_shr:8
    DEBWT_DPP_ADD(0x142, 0xa);      // row_bcast:15 -> rows 1, 3
    DEBWT_DPP_ADD(0x143, 0xc);      // row_bcast:31 -> rows 2, 3
#undef DEBWT_DPP_ADD
    return v;
}

// exclusive scan over a 256-thread block; *total = block sum.  smem: >= DEBWT_WAVES+1 u32.
// Ends with a barrier, so smem may be reused right after.
__device__ __forceinline__ u32 block_scan_excl(u32 v, u32 *smem, u32 *total) {
    u32 incl = wave_scan_incl(v);
    u32 w = threadIdx.x >> 6;
    if (lane_id() == 63) smem[w] = incl;
    __syncthreads();
    u32 base = 0, sum = 0;
#pragma unroll
    for (u32 i = 0; i < DEBWT_WAVES; i++) {
        u32 t = smem[i];
        if (i < w) base += t;
        sum += t;
    }
    __syncthreads();
    *total = sum;
    return base + incl - v;
}

// exclusive scans of V independent values per thread over a 256-thread block (one pair of barriers for all V);
// tot[v] = block sum of value v.  smem: >= V*DEBWT_WAVES u32.
template <int V>
__device__ __forceinline__ void block_scan_excl_vec(const u32 (&val)[V], u32 (&excl)[V], u32 (&tot)[V], u32 *smem) {
    u32 incl[V];
#pragma unroll
    for (int v = 0; v < V; v++) incl[v] = val[v];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
        for (int v = 0; v < V; v++) {
            u32 t = __shfl_up(incl[v], d, 64);
            if ((int)lane_id() >= d) incl[v] += t;
        }
    }
    const u32 w = threadIdx.x >> 6;
    if (lane_id() == 63) {
#pragma unroll
        for (int v = 0; v < V; v++) smem[v * DEBWT_WAVES + w] = incl[v];
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < V; v++) {
        u32 base = 0, sum = 0;
#pragma unroll
        for (u32 i = 0; i < DEBWT_WAVES; i++) {
            u32 t = smem[v * DEBWT_WAVES + i];
            if (i < w) base += t;
            sum += t;
        }
        excl[v] = base + incl[v] - val[v];
        tot[v] = sum;
    }
    __syncthreads();
}
