// radix_sort.hip -- 8-bit LSD radix sort of 64-bit keys for gfx950 (wave64).
//
// Replaces the reference's edge sort (src/mySort.c:98-176, 203-238, 371-401).  Per pass:
//   algo 1:  rs_hist (LDS histogram per chunk) -> rs_scan (one workgroup) -> rs_scatter
//   algo 2:  rs_onesweep (chunk histograms of all passes up front, then one read + one write per pass,
//            tile prefixes by decoupled look-back over agent-scope status words)
// rs_scatter / rs_onesweep rank a tile of RS_TILE keys per iteration: wave-striped coalesced loads,
// per-wave digit matching with ballots, per-wave LDS counters, a 256-wide digit scan, staging of the
// tile in LDS in digit order, then coalesced stores of each digit's run.
#include "radix_sort.h"

// ---------------------------------------------------------------------------------------------------
// algo 1

__global__ __launch_bounds__(RS_BLOCK) void rs_hist_kernel(const u64 *__restrict__ keys, u64 n, u64 chunk,
                                                            int shift, u32 mask, u32 *__restrict__ counts,
                                                            u32 nchunks) {
    __shared__ u32 h[RS_RADIX];
    h[threadIdx.x] = 0;
    __syncthreads();
    u64 beg = (u64)blockIdx.x * chunk;
    u64 end = beg + chunk < n ? beg + chunk : n;
    // chunk is a multiple of RS_TILE, so beg is 16-byte aligned: two keys per lane per load
    for (u64 i = beg + 2ull * threadIdx.x; i < end; i += 2ull * RS_BLOCK) {
        if (i + 1 < end) {
            ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(keys + i);
            atomicAdd(&h[(u32)(v.x >> shift) & mask], 1u);
            atomicAdd(&h[(u32)(v.y >> shift) & mask], 1u);
        } else {
            atomicAdd(&h[(u32)(keys[i] >> shift) & mask], 1u);
        }
    }
    __syncthreads();
    counts[(u64)threadIdx.x * nchunks + blockIdx.x] = h[threadIdx.x];
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

// Rank one tile.  On return skeys holds the tile's keys grouped by digit (stable), lstart[d] the
// first LDS slot of digit d, and the return value of each thread d is the tile's count of digit d.
// `cnt` is the number of valid keys of the tile (invalid slots only at the very end of the input).
__device__ __forceinline__ u32 rs_rank_tile(const u64 *__restrict__ in, u64 tile, u64 end, int shift, u32 mask,
                                            u64 *skeys, u32 (*wavecnt)[RS_RADIX], u32 *lstart, u32 *scan_tmp,
                                            u32 *tile_total) {
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
#pragma unroll
    for (u32 i = 0; i < DEBWT_WAVES; i++) wavecnt[i][tid] = 0;
    __syncthreads();
    u64 key[RS_ITEMS];
    u32 rnk[RS_ITEMS];
    const u64 wbase = tile + (u64)w * (64u * RS_ITEMS);
    const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        u64 idx = wbase + (u64)r * 64u + lane;
        key[r] = idx < end ? in[idx] : ~0ull;
    }
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        u64 idx = wbase + (u64)r * 64u + lane;
        bool valid = idx < end;
        u32 d = (u32)(key[r] >> shift) & mask;
        u64 m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            bool bit = (d >> b) & 1u;
            u64 bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        u32 before = (u32)__popcll(m & lt);
        u32 base = wavecnt[w][d];
        rnk[r] = base + before;
        if (valid && before == 0) wavecnt[w][d] = base + (u32)__popcll(m);
    }
    __syncthreads();
    u32 c0 = wavecnt[0][tid], c1 = wavecnt[1][tid], c2 = wavecnt[2][tid], c3 = wavecnt[3][tid];
    u32 total = c0 + c1 + c2 + c3;
    wavecnt[0][tid] = 0; wavecnt[1][tid] = c0; wavecnt[2][tid] = c0 + c1; wavecnt[3][tid] = c0 + c1 + c2;
    u32 ls = block_scan_excl(total, scan_tmp, tile_total);
    lstart[tid] = ls;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        u64 idx = wbase + (u64)r * 64u + lane;
        if (idx < end) {
            u32 d = (u32)(key[r] >> shift) & mask;
            skeys[lstart[d] + wavecnt[w][d] + rnk[r]] = key[r];
        }
    }
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(RS_BLOCK) void rs_scatter_kernel(const u64 *__restrict__ in, u64 *__restrict__ out,
                                                               u64 n, u64 chunk, int shift, u32 mask,
                                                               const u32 *__restrict__ offsets,
                                                               const u32 *__restrict__ digit_base, u32 nchunks) {
    __shared__ u64 skeys[RS_TILE];
    __shared__ u32 wavecnt[DEBWT_WAVES][RS_RADIX];
    __shared__ u32 lstart[RS_RADIX];
    __shared__ u32 run[RS_RADIX];
    __shared__ u32 scan_tmp[8];
    const u32 tid = threadIdx.x;
    run[tid] = offsets[(u64)tid * nchunks + blockIdx.x] + digit_base[tid];
    u64 beg = (u64)blockIdx.x * chunk;
    u64 end = beg + chunk < n ? beg + chunk : n;
    for (u64 tile = beg; tile < end; tile += RS_TILE) {
        u32 tot;
        u32 mine = rs_rank_tile(in, tile, end, shift, mask, skeys, wavecnt, lstart, scan_tmp, &tot);
        for (u32 j = tid; j < tot; j += RS_BLOCK) {
            u64 k = skeys[j];
            u32 d = (u32)(k >> shift) & mask;
            out[(u64)run[d] + (j - lstart[d])] = k;
        }
        __syncthreads();
        run[tid] += mine;
    }
}

// ---------------------------------------------------------------------------------------------------

size_t radix_workspace_bytes(u64 max_keys) {
    (void)max_keys;
    return (size_t)RS_RADIX * (RS_MAXCHUNKS + 1) * sizeof(u32);
}

static void rs_plan(u64 n, u32 *nchunks, u64 *chunk) {
    u64 tiles = (n + RS_TILE - 1) / RS_TILE;
    u64 c = tiles < RS_MAXCHUNKS ? tiles : RS_MAXCHUNKS;
    if (c == 0) c = 1;
    u64 tiles_per = (tiles + c - 1) / c;
    if (tiles_per == 0) tiles_per = 1;
    *chunk = tiles_per * RS_TILE;
    *nchunks = (u32)((n + *chunk - 1) / *chunk);
    if (*nchunks == 0) *nchunks = 1;
}

u64 *radix_sort_u64(hipStream_t stream, u64 *a, u64 *b, u64 n, int key_bits, const RadixWorkspace &ws, int algo,
                    hipEvent_t *pass_events, int max_pairs, int *npairs, hipError_t *err) {
    (void)algo;
    *err = hipSuccess;
    if (npairs) *npairs = 0;
    if (n < 2 || key_bits <= 0) return a;
    if (key_bits > 64) key_bits = 64;
    u32 nchunks; u64 chunk;
    rs_plan(n, &nchunks, &chunk);
    int passes = (key_bits + 7) / 8;
    u64 *src = a, *dst = b;
    for (int p = 0; p < passes; p++) {
        int shift = 8 * p;
        int bits = key_bits - shift < 8 ? key_bits - shift : 8;
        u32 mask = (1u << bits) - 1u;
        bool ev = pass_events && p < max_pairs;
        rs_hist_kernel<<<nchunks, RS_BLOCK, 0, stream>>>(src, n, chunk, shift, mask, ws.counts, nchunks);
        u32 *digit_tot = ws.counts + (size_t)RS_RADIX * RS_MAXCHUNKS;
        rs_scan_digit_kernel<<<RS_RADIX, 1024, 0, stream>>>(ws.counts, nchunks, digit_tot);
        rs_scan_tot_kernel<<<1, RS_RADIX, 0, stream>>>(digit_tot);
        if (ev) (void)hipEventRecord(pass_events[2 * p], stream);
        rs_scatter_kernel<<<nchunks, RS_BLOCK, 0, stream>>>(src, dst, n, chunk, shift, mask, ws.counts, digit_tot, nchunks);
        if (ev) { (void)hipEventRecord(pass_events[2 * p + 1], stream); if (npairs) *npairs = p + 1; }
        u64 *t = src; src = dst; dst = t;
    }
    *err = hipGetLastError();
    return src;
}
