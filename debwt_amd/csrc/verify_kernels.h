// verify_kernels.h -- inverse BWT on the device: the LF walk of the reference's LFsearch (src/LFsearch.c:49-166, occ
// tables src/insertCase3.c:141-194 -- dead code there, insertCase3 exits first, src/insertCase3.c:137), rebuilt as
// a verifier that scales to 30 Gbp: the reference walks the text as ONE chain of n dependent steps (one cache miss
// each: hours at 30 Gbp); here the chain is cut into segments whose start rows are found by backward search of the
// text in the BWT itself (FM-index counting), every segment is walked by its own thread and checked symbol by symbol
// against the text, and every segment must end on the row the previous segment started from -- so the segments form
// one chain over all n rows exactly when the BWT is the BWT of the text.
// Single TU: included by debwt_hip.hip only.
#pragma once
#include "common.h"

// rank structure: one 128-byte line per VB_ROWS rows: 4 header words + 12 words of packed rows.
//   header[c] bits 0..39  = rows before the line that hold code c (c = 3: 'T' and the separator rows)
//   header[0] bits 40..63, header[1] bits 40..63 = separator rows before the line (low / high 24 bits)
//   header[2] bits 40..63 = separator rows inside the line
#define VB_WORDS 12
#define VB_ROWS (VB_WORDS * 32)
#define VB_LINE 16
#define VB_CHUNK 256                    // lines per scan chunk
#define VCNT_MASK ((1ull << 40) - 1ull)

struct VCounts { u64 c[4]; u64 sep; };

// codes 1..3 of the 32 rows of a word among its first `lim` rows (row r at bits 2*(31-r)); code 0 = lim - the others
__device__ __forceinline__ void v_count_word(u64 v, u32 lim, u32 *c1, u32 *c2, u32 *c3) {
    const u64 valid = lim >= 32 ? 0x5555555555555555ull : (lim ? (0x5555555555555555ull << (2 * (32 - lim))) : 0ull);
    const u64 lo = v & valid, hi = (v >> 1) & valid;
    *c3 = (u32)__popcll(hi & lo); *c2 = (u32)__popcll(hi & ~lo); *c1 = (u32)__popcll(~hi & lo);
}

// counts of line `b` (rows [b*VB_ROWS, ...) clipped to n) from the packed BWT
__device__ __forceinline__ VCounts v_line_counts(const u64 *__restrict__ bwt, u64 n, u64 b, const u64 *__restrict__ srows,
                                                 u64 nsep) {
    VCounts o{};
    const u64 r0 = b * VB_ROWS;
    if (r0 >= n) return o;
    const u64 rows = n - r0 < VB_ROWS ? n - r0 : VB_ROWS;
    u32 c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
    for (u32 w = 0; w < VB_WORDS; w++) {
        const u64 done = (u64)w * 32;
        if (done >= rows) break;
        u32 a, b2, c;
        v_count_word(bwt[(r0 >> 5) + w], (u32)(rows - done < 32 ? rows - done : 32), &a, &b2, &c);
        c1 += a; c2 += b2; c3 += c;
    }
    o.c[1] = c1; o.c[2] = c2; o.c[3] = c3; o.c[0] = rows - c1 - c2 - c3;
    o.sep = lower_bound_dev<u64>(srows, 0, nsep, r0 + rows) - lower_bound_dev<u64>(srows, 0, nsep, r0);
    return o;
}

// pass 1: per chunk of VB_CHUNK lines, the sum of the line counts (5 words per chunk)
__global__ __launch_bounds__(VB_CHUNK) void k_vidx_count(const u64 *__restrict__ bwt, u64 n, u64 nlines,
                                                          const u64 *__restrict__ srows, u64 nsep, u64 *__restrict__ csum) {
    __shared__ u64 red[5][VB_CHUNK / 64];
    const u64 b = (u64)blockIdx.x * VB_CHUNK + threadIdx.x;
    VCounts v{};
    if (b < nlines) v = v_line_counts(bwt, n, b, srows, nsep);
    u64 x[5] = {v.c[0], v.c[1], v.c[2], v.c[3], v.sep};
#pragma unroll
    for (int q = 0; q < 5; q++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) x[q] += __shfl_xor(x[q], d, 64);
        if ((threadIdx.x & 63u) == 0) red[q][threadIdx.x >> 6] = x[q];
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        u64 t = 0;
        for (int w = 0; w < VB_CHUNK / 64; w++) t += red[threadIdx.x][w];
        csum[(u64)blockIdx.x * 5 + threadIdx.x] = t;
    }
}
// pass 2: exclusive scan of the chunk sums in place (one workgroup); totals[0..4]
__global__ __launch_bounds__(1024) void k_vidx_scan(u64 *__restrict__ csum, u64 nchunks, u64 *__restrict__ totals) {
    __shared__ u64 part[5][1024];
    const u32 tid = threadIdx.x;
    const u64 per = (nchunks + 1023) / 1024;
    const u64 lo = (u64)tid * per < nchunks ? (u64)tid * per : nchunks, hi = lo + per < nchunks ? lo + per : nchunks;
    u64 s[5] = {0, 0, 0, 0, 0};
    for (u64 i = lo; i < hi; i++)
        for (int q = 0; q < 5; q++) s[q] += csum[i * 5 + q];
    for (int q = 0; q < 5; q++) part[q][tid] = s[q];
    __syncthreads();
    for (u32 d = 1; d < 1024; d <<= 1) {
        u64 v[5];
        for (int q = 0; q < 5; q++) v[q] = tid >= d ? part[q][tid - d] : 0ull;
        __syncthreads();
        for (int q = 0; q < 5; q++) part[q][tid] += v[q];
        __syncthreads();
    }
    u64 run[5];
    for (int q = 0; q < 5; q++) run[q] = part[q][tid] - s[q];
    for (u64 i = lo; i < hi; i++)
        for (int q = 0; q < 5; q++) { const u64 c = csum[i * 5 + q]; csum[i * 5 + q] = run[q]; run[q] += c; }
    if (tid == 1023) for (int q = 0; q < 5; q++) totals[q] = part[q][1023];
}
// pass 3: the lines
__global__ __launch_bounds__(VB_CHUNK) void k_vidx_write(const u64 *__restrict__ bwt, u64 n, u64 nlines,
                                                          const u64 *__restrict__ srows, u64 nsep,
                                                          const u64 *__restrict__ csum, u64 *__restrict__ idx) {
    __shared__ u64 wsum[5][VB_CHUNK / 64];
    const u64 b = (u64)blockIdx.x * VB_CHUNK + threadIdx.x;
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    VCounts v{};
    if (b < nlines) v = v_line_counts(bwt, n, b, srows, nsep);
    u64 x[5] = {v.c[0], v.c[1], v.c[2], v.c[3], v.sep}, inc[5];
#pragma unroll
    for (int q = 0; q < 5; q++) {
        inc[q] = x[q];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u64 t = __shfl_up(inc[q], d, 64); if (lane >= (u32)d) inc[q] += t; }
        if (lane == 63) wsum[q][w] = inc[q];
    }
    __syncthreads();
    if (b >= nlines) return;
    u64 before[5];
#pragma unroll
    for (int q = 0; q < 5; q++) {
        u64 base = csum[(u64)blockIdx.x * 5 + q];
        for (u32 i = 0; i < w; i++) base += wsum[q][i];
        before[q] = base + inc[q] - x[q];
    }
    u64 *line = idx + b * VB_LINE;
    line[0] = before[0] | ((before[4] & 0xFFFFFFull) << 40);
    line[1] = before[1] | (((before[4] >> 24) & 0xFFFFFFull) << 40);
    line[2] = before[2] | (v.sep << 40);
    line[3] = before[3];
    const u64 r0 = b * VB_ROWS;
#pragma unroll
    for (u32 t = 0; t < VB_WORDS; t++) line[4 + t] = r0 + (u64)t * 32 < n ? bwt[(r0 >> 5) + t] : 0ull;
}

struct VIndex {
    const u64 *idx;        // nlines lines
    const u64 *hash;       // rows that hold '#', ascending (nhash)
    const u64 *srows;      // '#' rows and the '$' row, ascending (nsep = nhash + 1)
    u64 nhash, nsep, n;
    u64 C[6];              // first row of the suffixes starting with A, C, G, T, '#', '$'
    u64 dollar_row;
};

// rows [0, i) that hold symbol s (0..3 bases, 4 = '#'); i <= n
__device__ __forceinline__ u64 v_occ(const VIndex &V, u32 s, u64 i) {
    if (s == 4) return lower_bound_dev<u64>(V.hash, 0, V.nhash, i);
    const u64 b = i / VB_ROWS;
    const u32 off = (u32)(i - b * VB_ROWS);
    const u64 *line = V.idx + b * VB_LINE;
    const u64 h = line[s];
    u64 cnt = h & VCNT_MASK;
    const u64 pat = 0x5555555555555555ull * s;                // the code in every row position
#pragma unroll
    for (u32 w = 0; w < VB_WORDS; w++) {
        const u32 done = w * 32;
        if (done >= off) break;
        const u32 lim = off - done < 32 ? off - done : 32;
        const u64 x = line[4 + w] ^ pat;
        u64 m = ~(x | (x >> 1)) & 0x5555555555555555ull;      // rows whose code equals s
        if (lim < 32) m &= 0x5555555555555555ull << (2 * (32 - lim));
        cnt += (u64)__popcll(m);
    }
    if (s == 3) {                                              // 'T' rows = code-3 rows minus the separator rows
        const u64 sep_before = ((line[0] >> 40) & 0xFFFFFFull) | (((line[1] >> 40) & 0xFFFFFFull) << 24);
        const u64 in_line = line[2] >> 40;
        cnt -= in_line ? lower_bound_dev<u64>(V.srows, 0, V.nsep, i) : sep_before;
    }
    return cnt;
}

// symbol of row r (0..3, 4 = '#', 5 = '$') and the row of the suffix one position earlier in the text (LF)
__device__ __forceinline__ u64 v_lf(const VIndex &V, u64 r, u32 *sym) {
    const u64 b = r / VB_ROWS;
    const u32 off = (u32)(r - b * VB_ROWS);
    const u64 *line = V.idx + b * VB_LINE;
    u32 s = (u32)(line[4 + (off >> 5)] >> (2 * (31 - (off & 31)))) & 3u;
    if (s == 3 && (line[2] >> 40)) {                           // a code-3 row in a line that holds separator rows
        if (r == V.dollar_row) { *sym = 5; return V.n - 1; }
        const u64 j = lower_bound_dev<u64>(V.hash, 0, V.nhash, r);
        if (j < V.nhash && V.hash[j] == r) { *sym = 4; return V.C[4] + j; }
    }
    *sym = s;
    return V.C[s] + v_occ(V, s, r);
}

// text symbol at position p: 0..3, 4 = '#', 5 = '$'
__device__ __forceinline__ u32 v_text_symbol(const u64 *__restrict__ text, const u64 *__restrict__ sepbits, u64 n, u64 p) {
    if (sep_at(sepbits, p)) return p == n - 1 ? 5u : 4u;
    return text_symbol(text, p);
}

// Backward search: boundary j looks for the row of a suffix near text position q_j = (j + 1) * gap: the pattern
// T[q - m .. q) is extended to the left until exactly one suffix starts with it -- its row is the row of the suffix at
// q - m.  out[2j] = position (or ~0: not unique within maxm symbols / text start reached), out[2j+1] = row.
// fails: the pattern does not occur at all -- the BWT is not the text's.
__global__ __launch_bounds__(256) void k_vsearch(VIndex V, const u64 *__restrict__ text, const u64 *__restrict__ sepbits,
                                                 u64 nbound, u64 gap, u32 maxm, u64 *__restrict__ out,
                                                 u64 *__restrict__ counters) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nbound) return;
    const u64 q = (j + 1) * gap;
    u64 lo = 0, hi = V.n, steps = 0;
    u64 pos = ~0ull, row = 0;
    for (u32 m = 1; m <= maxm && m <= q; m++) {
        const u32 s = v_text_symbol(text, sepbits, V.n, q - m);
        if (s == 5) break;
        lo = V.C[s] + v_occ(V, s, lo);
        hi = V.C[s] + v_occ(V, s, hi);
        steps++;
        if (hi <= lo) { atomicAdd(&counters[2], 1ull); break; }           // a substring of the text that the BWT lacks
        if (hi - lo == 1) { pos = q - m; row = lo; break; }
    }
    out[2 * j] = pos; out[2 * j + 1] = row;
    if (steps) atomicAdd(&counters[3], steps);
}

// Walk: segment j goes from boundary j + 1 (position, row) backwards to boundary j, comparing the symbol of every
// row with the text symbol before the suffix; it must arrive on boundary j's row.  bounds: nseg + 1 (position, row)
// pairs, positions ascending, bounds[0] = (0, row whose symbol is '$'), bounds[nseg] = (n - 1, n - 1).
// counters[0] = symbol mismatches, [1] = segments that did not arrive on their row, [4] = steps walked
__global__ __launch_bounds__(256) void k_vwalk(VIndex V, const u64 *__restrict__ text, const u64 *__restrict__ sepbits,
                                               const u64 *__restrict__ bounds, u64 nseg, u64 *__restrict__ counters) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nseg) return;
    const u64 p0 = bounds[2 * j], r0 = bounds[2 * j + 1];
    u64 p = bounds[2 * j + 2], r = bounds[2 * j + 3];
    u64 steps = 0;
    bool ok = true;
    while (p > p0) {
        u32 sym;
        const u64 nr = v_lf(V, r, &sym);
        const u32 want = v_text_symbol(text, sepbits, V.n, p - 1);
        if (sym != want) { ok = false; atomicAdd(&counters[0], 1ull); break; }
        r = nr; p--; steps++;
    }
    if (ok && r != r0) { ok = false; atomicAdd(&counters[1], 1ull); }
    if (ok && j == 0) {                                        // the suffix at position 0 carries '$'
        u32 sym;
        (void)v_lf(V, r, &sym);
        if (sym != 5) atomicAdd(&counters[0], 1ull);
    }
    if (steps) atomicAdd(&counters[4], steps);
}
