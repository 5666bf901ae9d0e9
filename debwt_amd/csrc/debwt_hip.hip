// debwt_hip.hip -- C ABI (include/debwt_hip.h) and stage orchestration of the gfx950 deBWT path.
//
// One context = one GPU = one HIP stream.  Device buffers grow on demand and are kept between runs,
// so a steady-state run performs no allocation.  The only host work inside a run is the
// special-region module (special_host.cpp), which overlaps the GPU's key sort.
#include "../../include/debwt_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <new>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "radix_sort.h"
#include "special_host.h"
#include "fasta_host.h"
#include "gz_parallel.h"
#include "stage_kernels.h"
#include "special_kernels.h"
#include "verify_kernels.h"

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

enum Stage { ST_EMPTY = 0, ST_LOADED, ST_SORTED, ST_CLASSIFIED, ST_SP, ST_BLUE, ST_ASSEMBLED };

}  // namespace

struct debwt_ctx {
    debwt_config cfg{};
    int K = 31;
    hipStream_t stream = nullptr;
    std::string err;
    Stage stage = ST_EMPTY;
    debwt_stats st{};

    // host side of the loaded text
    const uint64_t *h_text = nullptr;
    std::vector<uint64_t> own_text;
    std::vector<uint64_t> h_sep;
    uint64_t n = 0, nrec = 0, M = 0, NS = 0;
    SpecialTables special;

    // device buffers
    DevBuf rs_rle, text, sepbits, sep, keysA, keysB, rs_counts, cp_counts, dk, dstart, mchar, head_keys, facts, facts_tmp,
        red, red_q, mi_j0, mi_freq, bstart, cursor, blue, spkey, sprow, spchr, branch, pflag, spsym, spn, bwt,
        hmask, hash_rows, dollar, large_q, large_k0, large_en, rowsym, momask, mimask, rbits, rs_over, rs_skew, mi_list, htab, fact_work, facts_all, shard_hist, dest_tab, qbounds, qcursor, qlist, qwave;
    DevBuf brbits;              // the special branches as a bitmap over the text positions (collections of many records)
    DevBuf blue_done;           // one byte per multi-in block: finished by k_blue_classify
    bool branch_bitmap = false;
    DevBuf sx, sppos, sprec, tail_d;   // special-region module on the device: scratch arena, positions / records of the sorted items, tail facts
    bool special_dev = false;   // the tables of this build were made on the device (spkey / spchr / branch / head_keys / tail_d)
    u64 nbranch = 0;            // special branches (specialBranchNum)
    u32 *h_over = nullptr;      // pinned mirror of rs_over
    u64 *sk = nullptr;          // sorted keys (keysA or keysB)
    u32 *h_scalars = nullptr;   // pinned read-back area
    u64 D = 0, Q = 0, Rmo = 0, R = 0, B = 0, S = 0, nlarge = 0, nfacts = 0;
    u64 rQ = 0, rB = 0, rnlarge = 0;   // blocks, blue rows, large blocks of the range just classified
    u64 rn1024 = 0, n1024 = 0;         // blocks of 513..1024 rows (range / context)
    u64 rn512 = 0, n512 = 0;           // blocks of 257..512 rows
    // Key ranges of this context, sorted and classified one after the other over the resident text (one range unless
    // the node instances exceed range_cap: "bucket streaming" for texts whose keys do not fit HBM at once).
    // D, Rmo and the buffers keysA/keysB/dk/dstart/pflag/mi_*/bstart/facts belong to the range being processed;
    // Q, B, nlarge and blk_*/facts_acc/large_q/mchar/sprow cover the whole context with 64-bit offsets.
    struct KeyRange { u64 key_lo, key_hi, M, Mbase, Q, qbase, B, Bbase, s0, s1, l0, nl; };   // l0, nl: its blocks above the LDS capacity in large_q
    std::vector<KeyRange> ranges;
    u64 range_cap = 0;          // 0: from the free HBM at the first build (plan_ranges)
    u64 Mctx = 0;               // node instances of this context (sum over its ranges)
    u64 nfacts_acc = 0;         // facts accumulated over the ranges
    bool local_done = false;    // classify_local already ran per range (multi-range build)
    bool plan_valid = false;    // `ranges` holds the cuts of the loaded text (several ranges)
    DevBuf vidx, vtmp;          // debwt_verify_device: rank structure (when no free key buffer holds it), small arrays
    DevBuf ls_buf, blk_j0, blk_freq, blk_start, facts_acc, large_tmp, blue_tmp, sub_start, sub_j0, sub_freq, sub_depth, range_hist;
    // k-mer-prefix shard of a multi-GPU build (world == 1: the whole key space)
    int shard_rank = 0, shard_world = 1;
    u64 Mfull = 0;              // node instances of the whole text
    u64 key_lo = 0, key_hi = 0; // this shard's key range [lo, hi); hi == 0: unbounded
    u64 Mbase = 0;              // node instances in the shards before this one
    u64 qbase = 0;              // multi-in blocks in the shards before this one
    u64 Btotal = 0;             // multi-in positions of the whole text
    u64 s0 = 0, s1 = 0;         // special suffixes [s0, s1) fall into this shard's node range
    bool facts_ready = false;
    u64 n_hash_local = 0;
    bool route_direct = false;  // SP pass 1 keeps the block ids of the multi-in positions (qlist: one run per wave, found
    u64 gq0 = 0;                //   through qwave, group gq0 first; qwave[0] is the bump counter), pass 2 writes routed entries
    bool exchange = false;      // sharded exchange mode: the keys of every range arrive by alltoallv in a caller buffer
    bool whole_build = false;   // inside debwt_build / debwt_build_to_host: nothing between the stages can be fetched
    bool shard_planned = false; // `ranges` were cut by debwt_shard_plan from the global census
    u64 *sort_a = nullptr, *sort_b = nullptr;   // the two key buffers of the range being sorted
    u64 Dsum = 0;               // distinct keys over the ranges sorted so far
    bool shared_hist = false;   // the first-pass histograms of all ranges came from one scan of the text
    u64 Qtotal = 0;             // multi-in blocks of the whole text (all shards)
    u64 S_rank = 0, B_rank = 0; // SP symbols / multi-in positions of this shard's text slice (all its sub-slices)
    bool reclaim_ok = false;    // an allocation that fails may release the idle buffers of the other stages (reclaim)
    u64 *routed = nullptr;      // ... and its routed blue entries (debwt_shard_sp_emit): a key buffer when one is free, else facts_tmp
    struct SubSlice { u64 g0, g1, S, B; };
    std::vector<SubSlice> sub;  // the slice in pieces of < 2^32 positions
    std::thread special_thread; // host special-region module, runs beside the key sort of the first range
    bool special_running = false;
    u64 g0 = 0, g1 = 0;         // text slice of this shard for the SP stage, in 32-position groups
    u64 S_local = 0, B_slice = 0, sp_off = 0;
    int hbits = 10, pbits = 13;
    bool mzfilter = false;      // the prefilter is indexed by the nodes' minimizers (pbits = log2 of its 64-bit words)
    int mzw = 16;               // ... of this many symbols
    bool mztable = false;       // ... and so is the node table (k_build_hash)
    bool abs32 = true;          // fill cursors hold absolute blue slots

    hipStream_t copy_stream = nullptr;   // debwt_build_to_host: finished row ranges leave on this stream under the blue sort
    hipEvent_t ev_copy = nullptr, ev_copy2 = nullptr;
    std::vector<u64> census;             // 12-mer prefix census of the loaded text, taken piece by piece behind its upload
    bool census_valid = false;
    hipEvent_t ev[8]{};         // stage boundaries
    hipEvent_t ev_pass[16][2]{};
    int n_pass_events = 0;
};

namespace {

#define HIPCHK(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                    \
            return e_ == hipErrorOutOfMemory ? DEBWT_ENOMEM : DEBWT_EDEVICE;                   \
        }                                                                                      \
    } while (0)

// Out of memory inside a build: the buffers of the stages that are over (or have not begun) hold nothing the build still
// needs -- after the classification the workspace of the key ranges (key buffers, distinct keys, their first rows: 30 bytes per
// key), during the key sort what the SP stage and the blue sort of the build BEFORE left behind (node table, work lists,
// split scratch) -- and are released, largest users first, before the allocation is tried once more.  k = 16 on a 3.1 Gbp text
// (1.07 G branching 15-mers: a 64 GB node table, 3 G blue rows) builds this way on one GPU; a text that fits keeps every
// buffer from build to build as before.  Returns the bytes released.
// Only at the two points where no pointer into those buffers is held and every one of them is allocated again before its
// next use (reclaim_ok): the allocations at the start of the key sort and at the start of the SP stage.
size_t reclaim(debwt_ctx *c, const DevBuf *keep) {
    if (!c->reclaim_ok) return 0;
    std::vector<DevBuf *> idle;
    if (c->stage >= ST_CLASSIFIED) {
        idle = {&c->keysB, &c->dk, &c->dstart, &c->rs_rle, &c->pflag};
        if (!c->exchange) idle.push_back(&c->keysA);
        c->sk = nullptr; c->routed = nullptr;
    } else {
        idle = {&c->htab, &c->mi_list, &c->qlist, &c->qwave, &c->blue_tmp, &c->ls_buf, &c->vidx, &c->vtmp, &c->large_k0,
                &c->large_en, &c->rowsym, &c->spn, &c->spsym, &c->rbits, &c->blue, &c->momask, &c->mimask};
    }
    size_t freed = 0;
    (void)hipStreamSynchronize(c->stream);
    for (DevBuf *b : idle)
        if (b != keep && b->p) { freed += b->cap; (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }
    if (getenv("DEBWT_TRACE_ALLOC")) fprintf(stderr, "reclaim: %.2f GB of idle buffers released\n", freed / 1e9);
    return freed;
}

int ensure(debwt_ctx *c, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return DEBWT_OK;
    static const bool trace = getenv("DEBWT_TRACE_ALLOC") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    const size_t old = b.cap;
    if (b.p) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    const auto t1 = std::chrono::steady_clock::now();
    size_t want = bytes + bytes / 16 + 256;
    hipError_t me = hipMalloc(&b.p, want);
    if (me == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        b.p = nullptr;
        if (reclaim(c, &b)) me = hipMalloc(&b.p, want);
        if (me == hipErrorOutOfMemory) { (void)hipGetLastError(); want = bytes + 256; me = hipMalloc(&b.p, want); }   // without the slack
    }
    if (me != hipSuccess) { b.p = nullptr; c->err = std::string("hipMalloc of ") + std::to_string(want) + " bytes: " + hipGetErrorString(me); return me == hipErrorOutOfMemory ? DEBWT_ENOMEM : DEBWT_EDEVICE; }
    b.cap = want;
    if (trace && want > (64u << 20))
        fprintf(stderr, "ensure: %.2f GB (was %.2f): sync+free %.1f ms, malloc %.1f ms\n", want / 1e9, old / 1e9,
                std::chrono::duration<float, std::milli>(t1 - t0).count(),
                std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t1).count());
    return DEBWT_OK;
}
#define ENSURE(c, b, bytes) do { int r_ = ensure((c), (b), (bytes)); if (r_) return r_; } while (0)
// grows a buffer that accumulates over the key ranges: the first `used` bytes survive
int ensure_keep(debwt_ctx *c, DevBuf &b, size_t bytes, size_t used) {
    if (bytes <= b.cap) return DEBWT_OK;
    size_t want = bytes + bytes / 2 + 256;
    void *np = nullptr;
    HIPCHK(c, hipMalloc(&np, want));
    if (b.p) {
        if (used) HIPCHK(c, hipMemcpyAsync(np, b.p, used, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(b.p));
    }
    b.p = np; b.cap = want;
    return DEBWT_OK;
}
#define ENSURE_KEEP(c, b, bytes, used) do { int r_ = ensure_keep((c), (b), (bytes), (used)); if (r_) return r_; } while (0)

void plan_chunks(u64 n, u32 *nchunks, u64 *chunk) {
    const u64 DEBWT_TILE = (u64)DEBWT_BLOCK * CP_VEC;
    u64 tiles = (n + DEBWT_TILE - 1) / DEBWT_TILE;
    u64 c = tiles < CP_MAXCHUNKS ? tiles : CP_MAXCHUNKS;
    if (c == 0) c = 1;
    u64 per = (tiles + c - 1) / c;
    if (per == 0) per = 1;                              // n == 0: one empty chunk
    *chunk = per * DEBWT_TILE;
    *nchunks = (u32)((n + *chunk - 1) / *chunk);
    if (*nchunks == 0) *nchunks = 1;
}

// count + scan of a functor; the total lands in h_scalars[slot] once the stream is synchronised
template <class F> int cp_count(debwt_ctx *c, const F &f, u64 n, u32 *counts, int slot) {
    u32 nchunks; u64 chunk;
    plan_chunks(n, &nchunks, &chunk);
    if (n) cp_count_kernel<F><<<nchunks, DEBWT_BLOCK, 0, c->stream>>>(f, n, chunk, counts);
    else HIPCHK(c, hipMemsetAsync(counts, 0, sizeof(u32), c->stream));
    u32 *total = counts + CP_MAXCHUNKS;
    cp_scan_kernel<<<1, 1024, 0, c->stream>>>(counts, n ? nchunks : 1, total);
    HIPCHK(c, hipMemcpyAsync(&c->h_scalars[slot], total, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    return DEBWT_OK;
}
template <class F> int cp_count2(debwt_ctx *c, const F &f, u64 n, u32 *ca, int slot_a, u32 *cb, int slot_b) {
    u32 nchunks; u64 chunk;
    plan_chunks(n, &nchunks, &chunk);
    if (n) cp_count2_kernel<F><<<nchunks, DEBWT_BLOCK, 0, c->stream>>>(f, n, chunk, ca, cb);
    else { HIPCHK(c, hipMemsetAsync(ca, 0, sizeof(u32), c->stream)); HIPCHK(c, hipMemsetAsync(cb, 0, sizeof(u32), c->stream)); }
    cp_scan_kernel<<<1, 1024, 0, c->stream>>>(ca, n ? nchunks : 1, ca + CP_MAXCHUNKS);
    cp_scan_kernel<<<1, 1024, 0, c->stream>>>(cb, n ? nchunks : 1, cb + CP_MAXCHUNKS);
    HIPCHK(c, hipMemcpyAsync(&c->h_scalars[slot_a], ca + CP_MAXCHUNKS, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&c->h_scalars[slot_b], cb + CP_MAXCHUNKS, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    return DEBWT_OK;
}
template <class F> int cp_emit(debwt_ctx *c, const F &f, u64 n, const u32 *counts) {
    if (!n) return DEBWT_OK;
    u32 nchunks; u64 chunk;
    plan_chunks(n, &nchunks, &chunk);
    cp_emit_kernel<F><<<nchunks, DEBWT_BLOCK, 0, c->stream>>>(f, n, chunk, counts);
    return DEBWT_OK;
}
// each in-flight compaction needs its own counts area: CP_MAXCHUNKS + 1 words per slot
u32 *cp_area(debwt_ctx *c, int slot) { return c->cp_counts.as<u32>() + (size_t)slot * (CP_MAXCHUNKS + 16); }

int sync_check(debwt_ctx *c) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    return DEBWT_OK;
}

// Copies between a CALLER's host buffer and the device are queued on the context's streams: whoever returns early with an
// error must not leave them in flight -- the caller is free to release the buffer as soon as the call is back.
struct DrainOnError {
    debwt_ctx *c;
    bool armed = true;
    ~DrainOnError() {
        if (!armed) return;
        if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
    }
};

inline u32 grid_for(u64 n, u32 block) { return (u32)((n + block - 1) / block); }

RadixWorkspace radix_ws(debwt_ctx *c) {
    RadixWorkspace ws{};
    ws.counts = c->rs_counts.as<u32>();
    ws.over = c->rs_over.as<u32>();
    ws.h_over = c->h_over;
    ws.over_cap = c->rs_over.cap >= 48 ? (u32)std::min<size_t>((c->rs_over.cap - 32) / 16, 0xFFFFFFFFu) : 0;
    ws.skew_list = c->rs_skew.as<u32>();
    return ws;
}

// main: the key sort of a range (kernels named for the profile, keys possibly read off the text); otherwise one of
// the small auxiliary sorts.  Pass events are recorded when `record_passes`.
int sort_keys(debwt_ctx *c, u64 *a, u64 *b, u64 count, int key_bits, u64 **result, bool record_passes,
              const TextKeySrc *text = nullptr, bool main_sort = false, RleSink *sink = nullptr) {
    ENSURE(c, c->rs_over, radix_over_bytes(count));      // one list entry per 4096-key tile can be oversize
    RadixWorkspace ws = radix_ws(c);
    hipError_t e = hipSuccess;
    const bool main = main_sort || record_passes;
    const int net = (c->cfg.reserved & 32768) ? 32 : 0;     // bit 15: the bucket finish prefers the 4096-key network (tests)
    if (record_passes) {
        *result = radix_sort_u64(c->stream, a, b, count, key_bits, ws, c->cfg.sort_algo | net, &c->ev_pass[0][0], 16,
                                 &c->n_pass_events, &e, text, sink);
        c->st.radix_pass_keys = count;
    } else {
        *result = radix_sort_u64(c->stream, a, b, count, key_bits, ws, c->cfg.sort_algo | net | (main ? 0 : 16), nullptr, 0,
                                 nullptr, &e, text, sink);
    }
    if (e != hipSuccess) { c->err = std::string("radix sort: ") + hipGetErrorString(e); return DEBWT_EDEVICE; }
    return DEBWT_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------

// every device buffer of a context (all grow on demand and are reused by the next build)
static std::vector<DevBuf *> all_buffers(debwt_ctx *c) {
    return {&c->text, &c->sepbits, &c->sep, &c->keysA, &c->keysB, &c->rs_counts, &c->cp_counts, &c->dk,
            &c->dstart, &c->mchar, &c->head_keys, &c->facts, &c->facts_tmp, &c->red, &c->red_q,
            &c->mi_j0, &c->mi_freq, &c->bstart, &c->cursor, &c->blue, &c->spkey, &c->sprow, &c->spchr,
            &c->branch, &c->pflag, &c->spsym, &c->spn, &c->bwt, &c->hmask, &c->hash_rows, &c->dollar,
            &c->large_q, &c->large_k0, &c->large_en, &c->rowsym, &c->momask, &c->mimask, &c->rbits, &c->rs_over, &c->rs_skew,
            &c->mi_list, &c->htab, &c->fact_work, &c->facts_all, &c->shard_hist, &c->dest_tab, &c->qbounds, &c->qcursor,
            &c->sx, &c->sppos, &c->sprec, &c->tail_d, &c->brbits,
            &c->blk_j0, &c->blk_freq, &c->blk_start, &c->facts_acc, &c->large_tmp, &c->blue_tmp, &c->sub_start, &c->sub_j0,
            &c->sub_freq, &c->sub_depth, &c->range_hist, &c->rs_rle, &c->ls_buf, &c->vidx, &c->vtmp, &c->qlist, &c->qwave,
            &c->blue_done};
}

extern "C" const char *debwt_strerror(int code) {
    switch (code) {
        case DEBWT_OK: return "ok";
        case DEBWT_EINVAL: return "invalid argument";
        case DEBWT_ENOMEM: return "out of memory";
        case DEBWT_EDEVICE: return "HIP runtime error";
        case DEBWT_ESTATE: return "stage called out of order";
        case DEBWT_ERANGE: return "input exceeds a capacity of this build";
        case DEBWT_EINTERNAL: return "internal consistency check failed";
        case DEBWT_EIO: return "file could not be written";
        default: return "unknown error";
    }
}

extern "C" const char *debwt_last_error(const debwt_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }

extern "C" int debwt_get_config(const debwt_ctx *ctx, debwt_config *out) {
    if (!ctx || !out) return DEBWT_EINVAL;
    *out = ctx->cfg;
    return DEBWT_OK;
}

extern "C" int debwt_create(const debwt_config *cfg, debwt_ctx **out) {
    if (!cfg || !out || cfg->k < 12 || cfg->k > 32) return DEBWT_EINVAL;   // src/main.c:41-47
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || cfg->device < 0 || cfg->device >= ndev) return DEBWT_EDEVICE;
    debwt_ctx *c = new (std::nothrow) debwt_ctx();
    if (!c) return DEBWT_ENOMEM;
    c->cfg = *cfg;
    if (c->cfg.sort_algo == 0) c->cfg.sort_algo = 3;
    c->K = cfg->k - 1;
    int rc = DEBWT_OK;
    auto fail = [&](int code) { debwt_destroy(c); return code; };
    if (hipSetDevice(cfg->device) != hipSuccess) return fail(DEBWT_EDEVICE);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail(DEBWT_EDEVICE);
    if (hipHostMalloc((void **)&c->h_scalars, 64 * sizeof(u32), hipHostMallocDefault) != hipSuccess)
        return fail(DEBWT_ENOMEM);
    for (auto &e : c->ev) if (hipEventCreate(&e) != hipSuccess) return fail(DEBWT_EDEVICE);
    for (auto &p : c->ev_pass) for (auto &e : p) if (hipEventCreate(&e) != hipSuccess) return fail(DEBWT_EDEVICE);
    if ((rc = ensure(c, c->rs_counts, radix_workspace_bytes(0))) != DEBWT_OK) return fail(rc);
    if ((rc = ensure(c, c->rs_over, radix_over_bytes(1 << 20))) != DEBWT_OK) return fail(rc);
    if (hipHostMalloc((void **)&c->h_over, 64, hipHostMallocDefault) != hipSuccess)
        return fail(DEBWT_ENOMEM);
    if ((rc = ensure(c, c->cp_counts, 8 * (CP_MAXCHUNKS + 16) * sizeof(u32))) != DEBWT_OK) return fail(rc);
    if ((rc = ensure(c, c->dollar, 64)) != DEBWT_OK) return fail(rc);
    *out = c;
    return DEBWT_OK;
}

extern "C" void debwt_destroy(debwt_ctx *c) {
    if (!c) return;
    if (c->special_running) { c->special_thread.join(); c->special_running = false; }
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (DevBuf *b : all_buffers(c)) if (b->p) (void)hipFree(b->p);
    if (c->h_scalars) (void)hipHostFree(c->h_scalars);
    if (c->h_over) (void)hipHostFree(c->h_over);
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto &p : c->ev_pass) for (auto &e : p) if (e) (void)hipEventDestroy(e);
    if (c->ev_copy) (void)hipEventDestroy(c->ev_copy);
    if (c->ev_copy2) (void)hipEventDestroy(c->ev_copy2);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static void join_special(debwt_ctx *c);

extern "C" int debwt_load_text(debwt_ctx *c, const uint64_t *packed, uint64_t n, const uint64_t *sep, uint64_t nrec) {
    if (!c || !packed || !sep || nrec == 0 || n < 34) return DEBWT_EINVAL;
    join_special(c);            // a special-region thread left behind by a failed stage still reads the text of the load before
    if (sep[nrec - 1] != n - 1) return DEBWT_EINVAL;
    uint64_t prev = 0;
    for (uint64_t r = 0; r < nrec; r++) {
        uint64_t start = r ? sep[r - 1] + 1 : 0;
        if (sep[r] < start + 33 || sep[r] >= n) return DEBWT_EINVAL;        // records > 32 bases
        prev = sep[r];
    }
    (void)prev;
    const int K = c->K;
    if (n <= nrec * (uint64_t)K) return DEBWT_EINVAL;
    uint64_t M = n - nrec * (uint64_t)K;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    c->stage = ST_EMPTY;                                  // a failed (re-)load leaves an unusable context, not a stale one
    c->h_text = packed;
    c->h_sep.assign(sep, sep + nrec);
    c->n = n; c->nrec = nrec; c->M = M; c->Mfull = M; c->Mctx = M; c->NS = nrec * (uint64_t)K;
    c->shard_rank = 0; c->shard_world = 1; c->key_lo = c->key_hi = 0; c->Mbase = 0; c->qbase = 0;
    c->exchange = false; c->shard_planned = false;
    c->ranges.clear(); c->plan_valid = false;
    size_t tw = (size_t)((n + 63) >> 5) + 2, bw = (size_t)(n >> 6) + 3;
    ENSURE(c, c->text, tw * 8);
    ENSURE(c, c->sepbits, bw * 8);
    ENSURE(c, c->sep, nrec * 8);
    const size_t words = (size_t)((n + 63) >> 5);
    DrainOnError drain{c};                                 // from here on `packed` and `sep` are being read by queued copies
    HIPCHK(c, hipMemcpyAsync(c->sep.p, sep, nrec * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->sepbits.p, 0, bw * 8, c->stream));
    k_set_sepbits<<<grid_for(nrec, 256), 256, 0, c->stream>>>(c->sep.as<u64>(), nrec, c->sepbits.as<u64>());
    HIPCHK(c, hipMemsetAsync(c->text.as<u64>() + words, 0, (tw - words) * 8, c->stream));
    c->census_valid = false;
    if (n >= (1ull << 31)) {
        // A text of this size will be built in key ranges, cut on the census of its 12-mer prefixes (plan_ranges): the text
        // travels in pieces on a second stream and the census of a piece runs while the next one is on its way (8 ms at
        // 30 Gbp that no longer stand between the load and the first key range).
        if (!c->copy_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        if (!c->ev_copy) HIPCHK(c, hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming));
        if (!c->ev_copy2) HIPCHK(c, hipEventCreateWithFlags(&c->ev_copy2, hipEventDisableTiming));
        ENSURE(c, c->shard_hist, SHARD_BINS * 8);
        HIPCHK(c, hipMemsetAsync(c->shard_hist.p, 0, SHARD_BINS * 8, c->stream));
        HIPCHK(c, hipEventRecord(c->ev_copy, c->stream));                       // the separator bitmap and the zeroed census
        HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_copy, 0));            // (the buffers may still be in use by the load before)
        const size_t piece = (size_t)1 << 24;                                   // 128 MB = 2^29 positions
        const size_t npieces = (words + piece - 1) / piece;
        for (size_t k = 0; k <= npieces; k++) {
            if (k < npieces) {
                const size_t a = k * piece, b = std::min(words, a + piece);
                HIPCHK(c, hipMemcpyAsync(c->text.as<u64>() + a, packed + a, (b - a) * 8, hipMemcpyHostToDevice, c->copy_stream));
                HIPCHK(c, hipEventRecord((k & 1) ? c->ev_copy2 : c->ev_copy, c->copy_stream));
                HIPCHK(c, hipStreamWaitEvent(c->stream, (k & 1) ? c->ev_copy2 : c->ev_copy, 0));
            }
            if (k >= 1) {                                                       // census of piece k - 1: its last word needs piece k's first
                const u64 p0 = (u64)(k - 1) * piece * 32, p1 = std::min<u64>(n, (u64)k * piece * 32);
                if (p1 > p0)
                    k_prefix_hist_words<<<1024, DEBWT_BLOCK, 0, c->stream>>>(c->text.as<u64>(), c->sepbits.as<u64>(), p0, p1, c->K,
                                                                              c->shard_hist.as<u64>());
            }
        }
        c->census.resize(SHARD_BINS);
        HIPCHK(c, hipMemcpyAsync(c->census.data(), c->shard_hist.p, SHARD_BINS * 8, hipMemcpyDeviceToHost, c->stream));
        c->census_valid = true;                                                 // (once the stream has drained, below)
    } else {
        HIPCHK(c, hipMemcpyAsync(c->text.p, packed, words * 8, hipMemcpyHostToDevice, c->stream));
    }
    // workspace that depends only on n (the key buffers are sized per key range in debwt_kmer_sort_rle)
    ENSURE(c, c->head_keys, nrec * 8);
    ENSURE(c, c->spkey, c->NS * 8);
    ENSURE(c, c->sprow, c->NS * 8);
    ENSURE(c, c->spchr, c->NS + 64);
    ENSURE(c, c->bwt, (size_t)((n + 31) >> 5) * 8 + 64);
    ENSURE(c, c->hmask, (size_t)((n + 31) >> 5) * 4 + 64);
    ENSURE(c, c->hash_rows, nrec * 8 + 64);
    int rc = sync_check(c);
    if (rc) return rc;
    drain.armed = false;                                   // (both streams have drained: the text copies are ordered before c->stream)
    c->stage = ST_LOADED;
    memset(&c->st, 0, sizeof c->st);
    c->st.n = n; c->st.nrec = nrec; c->st.n_main = M;
    return DEBWT_OK;
}

static_assert(sizeof(debwt_packed_text) == sizeof(PackedText), "C ABI mirror of PackedText");

extern "C" int debwt_pack_fasta_opts(const char *path, int threads, unsigned flags, uint64_t seed, debwt_packed_text *out,
                                     char *errbuf, size_t errlen) {
    if (!path || !out || (flags & ~DEBWT_FASTA_IUPAC_RANDOM)) return DEBWT_EINVAL;
    memset(out, 0, sizeof *out);
    return pack_fasta_file(path, threads, reinterpret_cast<PackedText *>(out), errbuf, errlen, IngestOpts{flags, seed})
               ? DEBWT_EINVAL : DEBWT_OK;
}
extern "C" int debwt_pack_fasta(const char *path, int threads, debwt_packed_text *out, char *errbuf, size_t errlen) {
    return debwt_pack_fasta_opts(path, threads, 0, 0, out, errbuf, errlen);
}
extern "C" void debwt_free_packed(debwt_packed_text *p) { free_packed_text(reinterpret_cast<PackedText *>(p)); }
extern "C" uint64_t debwt_fasta_text_bound(const char *path) { return path ? fasta_text_bound(path) : 0; }
extern "C" void debwt_host_release_hold(int on) { release_hold(on); }

extern "C" int debwt_load_fasta_opts(debwt_ctx *c, const char *path, int threads, unsigned flags, uint64_t seed) {
    if (!c || !path || (flags & ~DEBWT_FASTA_IUPAC_RANDOM)) return DEBWT_EINVAL;
    PackedText pt{};
    char msg[256] = "";
    struct Hold { Hold() { release_hold(1); } ~Hold() { release_hold(0); } } hold;      // (the ingest's buffers are released behind the load)
    if (pack_fasta_file(path, threads, &pt, msg, sizeof msg, IngestOpts{flags, seed})) { c->err = msg; return DEBWT_EINVAL; }
    c->own_text.assign(pt.words, pt.words + pt.nwords);
    std::vector<uint64_t> sep(pt.sep, pt.sep + pt.nrec);
    const uint64_t n = pt.n;
    free_packed_text(&pt);
    return debwt_load_text(c, c->own_text.data(), n, sep.data(), sep.size());
}
extern "C" int debwt_load_fasta(debwt_ctx *c, const char *path, int threads) {
    return debwt_load_fasta_opts(c, path, threads, 0, 0);
}

extern "C" int debwt_set_range_cap(debwt_ctx *c, uint64_t max_instances) {
    if (!c || max_instances < 4096) return DEBWT_EINVAL;
    c->range_cap = std::min<uint64_t>(max_instances, 0xFFFFFFF0ull - 1);
    c->plan_valid = false;
    return DEBWT_OK;
}

extern "C" int debwt_load_ascii(debwt_ctx *c, const char *seq, const uint64_t *reclen, uint64_t nrec) {
    if (!c || !seq || !reclen || !nrec) return DEBWT_EINVAL;
    uint64_t n = nrec;
    for (uint64_t r = 0; r < nrec; r++) {
        if (reclen[r] <= 32) return DEBWT_EINVAL;                              // src/collect#$.c:41-45
        n += reclen[r];
    }
    std::vector<uint64_t> words(((n + 63) >> 5) + 2, 0), sep(nrec);
    uint64_t o = 0, s = 0;
    auto put = [&](uint64_t j, uint64_t code) { words[j >> 5] |= code << ((31 - (j & 31)) << 1); };
    for (uint64_t r = 0; r < nrec; r++) {
        for (uint64_t j = 0; j < reclen[r]; j++, s++, o++) {
            uint64_t code;
            switch (seq[s]) {                                                  // src/main.c:18-23
                case 'A': case 'a': code = 0; break;
                case 'C': case 'c': code = 1; break;
                case 'G': case 'g': code = 2; break;
                case 'T': case 't': code = 3; break;
                default: return DEBWT_EINVAL;
            }
            put(o, code);
        }
        put(o, 3); sep[r] = o; o++;                                            // 'T' at the separator
    }
    for (uint64_t j = 0; j < 32; j++) put(o + j, 3);                           // src/collect#$.c:87-90
    c->own_text.swap(words);
    return debwt_load_text(c, c->own_text.data(), n, sep.data(), nrec);
}

// ---------------------------------------------------------------------------------------------------
// stage 1: keys, sort, RLE                                                            (a-1, a-2, a-3)

// Largest number of node instances one key range may hold: what the free HBM allows at `per_key` bytes of range
// workspace (key buffers, distinct keys, first instances, classification bytes; in exchange mode also the caller's
// send buffer) next to `later` bytes the later stages hold, and always below 2^32 (per-range indices are 32-bit).
static int default_range_cap(debwt_ctx *c, u64 per_key, u64 later, u64 caller_held, u64 *cap) {
    size_t free_b = 0, total_b = 0;
    HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
    u64 held = caller_held;                                  // what this build reuses: the context's own buffers and the
    for (DevBuf *b : all_buffers(c)) held += b->cap;         // caller's exchange buffers -- the plan of a repeated build
    const u64 avail = (free_b + held) / 100 * 94;            // equals the first; 6 %: allocator granularity, RCCL, runtime
    u64 rc = avail > later + (per_key << 28) ? (avail - later) / per_key : (1ull << 28);
    *cap = std::min<u64>(rc, 0xFFFFFFF0ull - (1ull << 20));
    return DEBWT_OK;
}

// Cuts the prefix bins [bin_lo, bin_hi) of the census `hist` into key ranges of at most `cap` node instances each
// (near-equal ranges; the reference balances its sort threads on the same census, src/mySort.c:98-110).
static int cut_ranges(debwt_ctx *c, const u64 *hist, u32 bin_lo, u32 bin_hi, u64 cap, u64 Mshard) {
    c->ranges.clear();
    cap = std::min<u64>(cap, 0xFFFFFFF0ull - 1);
    const u64 P = std::max<u64>(1, (Mshard + cap - 1) / cap);
    const u64 per = (Mshard + P - 1) / P;
    const u64 limit = std::min(cap, per + per / 16);            // near-equal ranges, never above the cap
    const int kb = 2 * c->cfg.k;
    auto push = [&](u32 lo, u32 hi, u64 m, u64 base) {
        debwt_ctx::KeyRange q{};
        q.key_lo = (u64)lo << (kb - 12);
        q.key_hi = hi == SHARD_BINS ? 0ull : ((u64)hi << (kb - 12));
        q.M = m; q.Mbase = base;
        c->ranges.push_back(q);
    };
    u64 acc = 0, base = 0, total = 0;
    u32 lo = bin_lo;
    for (u32 b = bin_lo; b < bin_hi; b++) {
        // a bin above the cap becomes a range of its own (the cap is a target; 2^32 instances is the hard limit)
        if (hist[b] >= 0xFFFFFFF0ull - (1ull << 20)) { c->err = "one 12-mer prefix bin holds 2^32 node instances or more"; return DEBWT_ERANGE; }
        if (acc && acc + hist[b] > limit) { push(lo, b, acc, base); base += acc; acc = 0; lo = b; }
        acc += hist[b]; total += hist[b];
    }
    push(lo, bin_hi, acc, base);
    if (total != Mshard) { c->err = "prefix census differs from the number of node instances"; return DEBWT_EINTERNAL; }
    return DEBWT_OK;
}

// The key ranges of this build.  One GPU: the whole key space, cut by the census of the text when it exceeds the range
// cap.  A shard: the ranges debwt_shard_plan cut, or the one range debwt_shard_set_range gave.
static int plan_ranges(debwt_ctx *c) {
    c->local_done = false;
    if (c->shard_world == 1 && !c->shard_planned) { c->key_lo = c->key_hi = 0; c->Mctx = c->Mfull; c->Mbase = 0; c->exchange = false; }
    if (c->shard_planned || (c->plan_valid && c->ranges.size() > 1 && c->shard_world == 1)) {
        // cut by debwt_shard_plan / by the census an earlier build of this text took: same cuts
        for (auto &r : c->ranges) { r.Q = r.qbase = r.B = r.Bbase = r.s0 = r.s1 = r.l0 = r.nl = 0; }
        return DEBWT_OK;
    }
    c->ranges.clear();
    u64 range_cap = c->range_cap;
    if (!range_cap) {
        // ~30 bytes per key of range workspace next to what the later stages hold for the whole text
        // (~4 bytes per position: row symbols, SP code, flag masks, blue entries, BWT)
        int rc = default_range_cap(c, 30, 4 * c->n + (8ull << 30), 0, &range_cap);
        if (rc) return rc;
        if (range_cap >= c->Mfull && c->Mfull < 0xFFFFFFF0ull) range_cap = c->Mfull;
    }
    if (c->shard_world > 1 || c->Mfull <= range_cap) {
        if (c->Mctx >= 0xFFFFFFF0ull) { c->err = "a key range must hold fewer than 2^32 node instances"; return DEBWT_ERANGE; }
        debwt_ctx::KeyRange r{};
        r.key_lo = c->key_lo; r.key_hi = c->key_hi; r.M = c->Mctx;
        c->ranges.push_back(r);
        return DEBWT_OK;
    }
    std::vector<u64> hist(SHARD_BINS);
    int rc;
    if (c->census_valid) hist = c->census;                      // taken behind the upload of the text (debwt_load_text)
    else {
        ENSURE(c, c->shard_hist, SHARD_BINS * 8);
        HIPCHK(c, hipMemsetAsync(c->shard_hist.p, 0, SHARD_BINS * 8, c->stream));
        k_prefix_hist_words<<<2048, DEBWT_BLOCK, 0, c->stream>>>(c->text.as<u64>(), c->sepbits.as<u64>(), 0, c->n, c->K,
                                                                  c->shard_hist.as<u64>());
        HIPCHK(c, hipMemcpyAsync(hist.data(), c->shard_hist.p, SHARD_BINS * 8, hipMemcpyDeviceToHost, c->stream));
        if ((rc = sync_check(c))) return rc;
    }
    if ((rc = cut_ranges(c, hist.data(), 0, SHARD_BINS, range_cap, c->Mfull))) return rc;
    c->plan_valid = true;
    return DEBWT_OK;
}

static int classify_local(debwt_ctx *c);
static int append_range(debwt_ctx *c, debwt_ctx::KeyRange &r);

static void join_special(debwt_ctx *c) {
    if (c->special_running) { c->special_thread.join(); c->special_running = false; }
}

// ---- special-region module on the device (special_kernels.h) --------------------------------------------

static int bits_for(u64 v) { int b = 1; while (b < 64 && (v >> b)) b++; return b; }     // bits that hold 0..v

// Collections of many records build their special-region tables on the device: from 2^14 special suffixes on
// (DEBWT_SPECIAL_DEVICE_MIN overrides; tests force either path), as long as the payload fields of its sorts hold the
// record ranks and item places (2^27 records, 2^32 special suffixes -- beyond that the host threads take over).
static bool special_wants_device(debwt_ctx *c) {
    const char *e = getenv("DEBWT_SPECIAL_DEVICE_MIN");
    const u64 dev_min = e ? strtoull(e, nullptr, 10) : (1ull << 14);
    if (c->NS < dev_min) return false;
    if (!(c->nrec < (1ull << 27) && c->NS < (1ull << 32) && 5 + bits_for(c->nrec) + bits_for(c->NS - 1) <= 64)) return false;
    // its workspace (59 bytes per special suffix + 12 for their positions, ~60 per record) must fit beside what the build
    // holds by now -- else the host threads build the tables, as for any collection before round 3
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
    if (const char *f = getenv("DEBWT_SPECIAL_FAKE_FREE_BYTES")) free_b = (size_t)strtoull(f, nullptr, 10);   // tests: this branch
    const u64 need = c->NS * 72 + c->nrec * 64 + (64ull << 20);
    const u64 have = free_b + c->sx.cap + c->sppos.cap + c->sprec.cap + c->tail_d.cap;
    return need + (have >> 4) < have;                          // and 1/16 of it stays free
}

// the branch list (c->branch, c->nbranch) as a bitmap for the SP flags pass, where the special suffixes are many
static int special_branch_bitmap(debwt_ctx *c) {
    c->branch_bitmap = c->nbranch > 0 && c->NS >= (1ull << 14);
    if (!c->branch_bitmap) return DEBWT_OK;
    const size_t bw = (size_t)(c->n >> 6) + 3;
    ENSURE(c, c->brbits, bw * 8);
    HIPCHK(c, hipMemsetAsync(c->brbits.p, 0, bw * 8, c->stream));
    k_set_sepbits<<<grid_for(c->nbranch, 256), 256, 0, c->stream>>>(c->branch.as<u64>(), c->nbranch, c->brbits.as<u64>());
    return DEBWT_OK;
}

static int special_device_build(debwt_ctx *c, bool release_arena = true) {
    const u64 N = c->nrec, NS = c->NS, n = c->n;
    const int K = c->K;
    int rc;
    const auto t_begin = std::chrono::steady_clock::now();
    // one scratch arena, carved: u64 sort buffers of NS words, per-record and per-item side arrays
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~size_t(255); return o; };
    const size_t oX = take(NS * 8 + 64), oY = take(NS * 8 + 64);
    const size_t o_ord = take(N * 4), o_gid = take(N * 4), o_actA = take(N * 4), o_actB = take(N * 4), o_val = take(N * 8);
    const size_t o_recs = take(N * 4), o_vals = take(N * 8), o_gids = take(N * 4), o_ordv = take(N * 4);
    const size_t o_head = take((N + 1) * 4), o_stay = take(N), o_rank = take(N * 4), o_dep = take(N * 4), o_gmin = take(N * 4);
    const size_t o_grp = take(NS * 4), o_gflag = take(NS), o_headf = take(NS), o_spd = take(NS), o_item = take(NS * 16), o_bnd = take(64);
    ENSURE(c, c->sx, off);
    ENSURE(c, c->sppos, NS * 8 + 64);
    ENSURE(c, c->sprec, NS * 4 + 64);
    ENSURE(c, c->tail_d, N * 8 + 64);
    ENSURE(c, c->rs_skew, (NS / 2048 + 2) * 4);
    u8 *base = c->sx.as<u8>();
    u64 *X = (u64 *)(base + oX), *Y = (u64 *)(base + oY);
    u32 *ord = (u32 *)(base + o_ord), *gid = (u32 *)(base + o_gid), *act = (u32 *)(base + o_actA), *act2 = (u32 *)(base + o_actB);
    u64 *valbuf = (u64 *)(base + o_val), *val_s = (u64 *)(base + o_vals);
    u32 *rec_s = (u32 *)(base + o_recs), *gid_s = (u32 *)(base + o_gids), *ordv = (u32 *)(base + o_ordv);
    u32 *headpos = (u32 *)(base + o_head), *rank = (u32 *)(base + o_rank), *dep = (u32 *)(base + o_dep), *gmin = (u32 *)(base + o_gmin);
    u8 *stay = base + o_stay;
    u32 *grp = (u32 *)(base + o_grp);
    u8 *gflag = base + o_gflag, *headf = base + o_headf, *spd = base + o_spd;
    ulonglong2 *item = (ulonglong2 *)(base + o_item);
    const SxText T{c->text.as<u64>(), c->sepbits.as<u64>(), c->sep.as<u64>(), n, N, K};
    auto grid = [](u64 m) { return grid_for(m, 256); };
    auto other = [&](u64 *p_) { return p_ == X ? Y : X; };
    // stable 8-bit passes over the bits [lo, hi) of `count` words in a (scratch b); the buffer that holds the result
    auto lsd = [&](u64 *a, u64 *b, u64 count, int lo, int hi) -> u64 * {
        hipError_t e = hipSuccess;
        u64 *res = radix_sort_bits(c->stream, a, b, count, lo, hi, radix_ws(c), &e);
        if (e != hipSuccess) { c->err = std::string("special-region sort: ") + hipGetErrorString(e); return nullptr; }
        return res;
    };

    // 1. ranks of the record starts: refinement rounds on 21-symbol windows inside the tie groups
    k_sx_init<<<grid(N), 256, 0, c->stream>>>(ord, gid, act, N);
    const char *env_rounds = getenv("DEBWT_SPECIAL_MAX_ROUNDS");
    const u64 max_rounds = env_rounds ? strtoull(env_rounds, nullptr, 10) : 64ull;
    u64 na = N > 1 ? N : 0;
    bool one_group = true;
    const bool host_ties = getenv("DEBWT_SPECIAL_HOST_TIES") != nullptr;     // tests, A/B: the long-tied groups by host comparison
    bool jumping = false;
    for (u64 w = 0; na; w++) {
        if (w >= max_rounds && host_ties) {
            // records identical for max_rounds * 21 symbols and more: the few groups left are ordered on the host
            std::vector<u32> h_ord(N), h_gid(N), h_act(na);
            HIPCHK(c, hipMemcpyAsync(h_ord.data(), ord, N * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(h_gid.data(), gid, N * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(h_act.data(), act, na * 4, hipMemcpyDeviceToHost, c->stream));
            if ((rc = sync_check(c))) return rc;
            special_order_record_starts(c->h_text, n, c->h_sep.data(), N, h_ord.data(), h_gid.data(), h_act.data(), na);
            HIPCHK(c, hipMemcpyAsync(ord, h_ord.data(), N * 4, hipMemcpyHostToDevice, c->stream));
            if ((rc = sync_check(c))) return rc;
            break;
        }
        if (w >= max_rounds) {
            // ... on the device: jump rounds (special_kernels.h) -- every group's depth advances by the windows all its members
            // share with its head, then the window round below tells at least one member apart
            if (!jumping) { k_sx_depth_init<<<grid(na), 256, 0, c->stream>>>(act, na, (u32)w, dep); jumping = true; }
            HIPCHK(c, hipMemsetAsync(gmin, 0xFF, N * 4, c->stream));
            k_sx_lcp_min<<<grid(na * 64), 256, 0, c->stream>>>(T, ord, gid, act, na, dep, gmin);
            k_sx_depth_add<<<grid(na), 256, 0, c->stream>>>(gid, act, na, gmin, dep, 0u);
        }
        const int bA = bits_for(na - 1);
        k_sx_round_keys<<<grid(na), 256, 0, c->stream>>>(T, ord, act, na, w, bA, valbuf, X, jumping ? dep : nullptr);
        // three stable sorts, least significant field first: low 32 bits of the window, its high 31 bits, the tie group
        u64 *r = lsd(X, Y, na, bA, bA + 32);
        if (!r) return DEBWT_EDEVICE;
        u64 *o = other(r);
        k_sx_rekey<<<grid(na), 256, 0, c->stream>>>(r, valbuf, gid, act, na, bA, 0, o);
        if (!(r = lsd(o, other(o), na, bA, bA + 31))) return DEBWT_EDEVICE;
        if (!one_group) {
            o = other(r);
            k_sx_rekey<<<grid(na), 256, 0, c->stream>>>(r, valbuf, gid, act, na, bA, 1, o);
            if (!(r = lsd(o, other(o), na, bA, bA + bits_for(N - 1)))) return DEBWT_EDEVICE;
        }
        k_sx_gather<<<grid(na), 256, 0, c->stream>>>(r, bA, valbuf, ord, gid, act, na, rec_s, val_s, gid_s);
        SxHeadF fh{gid_s, val_s, ordv, headpos};
        if ((rc = cp_count(c, fh, na, cp_area(c, 0), 20))) return rc;
        if ((rc = cp_emit(c, fh, na, cp_area(c, 0)))) return rc;
        k_sx_sentinel<<<1, 1, 0, c->stream>>>(headpos, cp_area(c, 0) + CP_MAXCHUNKS, (u32)na);
        k_sx_apply<<<grid(na), 256, 0, c->stream>>>(rec_s, ordv, headpos, act, na, ord, gid, stay);
        SxStayF fs{stay, act, act2};
        if ((rc = cp_count(c, fs, na, cp_area(c, 1), 21))) return rc;
        if ((rc = cp_emit(c, fs, na, cp_area(c, 1)))) return rc;
        if ((rc = sync_check(c))) return rc;
        na = c->h_scalars[21];
        std::swap(act, act2);
        one_group = false;
        if (jumping && na) k_sx_depth_add<<<grid(na), 256, 0, c->stream>>>(gid, act, na, nullptr, dep, 1u);   // the window just used
    }
    k_sx_rank_of<<<grid(N), 256, 0, c->stream>>>(ord, N, rank);

    // 2. the N*K special suffixes by (key, later separator first, follower rank): three stable passes, least significant first
    const int bR = bits_for(N), bP = bits_for(NS - 1);
    u64 *r = nullptr;
    (void)o_bnd;
    k_it_pass1<<<grid(NS), 256, 0, c->stream>>>(T, rank, NS, bR, bP, X, item);
    if (!(r = lsd(X, Y, NS, bP, bP + 5 + bR))) return DEBWT_EDEVICE;
    for (int hi = 0; hi < 2; hi++) {
        u64 *o = other(r);
        k_it_rekey<<<grid(NS), 256, 0, c->stream>>>(item, r, NS, hi, bP, o);
        if (!(r = lsd(o, other(o), NS, bP, bP + 31))) return DEBWT_EDEVICE;
    }
    k_it_out<<<grid(NS), 256, 0, c->stream>>>(T, item, r, bP, NS, c->spkey.as<u64>(), c->spchr.as<u8>(), c->sppos.as<u64>(),
                                              c->sprec.as<u32>(), spd);

    // 3. special branches; head and tail nodes
    SxBranchF fb{T, c->sppos.as<u64>(), c->sprec.as<u32>(), c->spkey.as<u64>(), spd, headf, grp};
    if ((rc = cp_count(c, fb, NS, cp_area(c, 0), 20))) return rc;
    if ((rc = cp_emit(c, fb, NS, cp_area(c, 0)))) return rc;
    HIPCHK(c, hipMemsetAsync(gflag, 0, NS, c->stream));
    k_br_diff<<<grid(NS), 256, 0, c->stream>>>(T, c->sppos.as<u64>(), grp, NS, gflag);
    SxBranchEmitF fe{gflag, grp, c->sppos.as<u64>(), nullptr};
    if ((rc = cp_count(c, fe, NS, cp_area(c, 1), 21))) return rc;
    if ((rc = sync_check(c))) return rc;
    c->nbranch = c->h_scalars[21];
    ENSURE(c, c->branch, c->nbranch * 8 + 64);
    if (c->nbranch) {
        fe.branch = X;
        if ((rc = cp_emit(c, fe, NS, cp_area(c, 1)))) return rc;
        if (!(r = lsd(X, Y, c->nbranch, 0, bits_for(n)))) return DEBWT_EDEVICE;
        HIPCHK(c, hipMemcpyAsync(c->branch.p, r, c->nbranch * 8, hipMemcpyDeviceToDevice, c->stream));
    }
    k_heads_tails<<<grid(N), 256, 0, c->stream>>>(T, X, c->tail_d.as<u64>());
    if (!(r = lsd(X, Y, N, 0, 2 * K + 2))) return DEBWT_EDEVICE;
    HIPCHK(c, hipMemcpyAsync(c->head_keys.p, r, N * 8, hipMemcpyDeviceToDevice, c->stream));
    if ((rc = special_branch_bitmap(c))) return rc;
    if ((rc = sync_check(c))) return rc;
    if (release_arena) {
        // The arena (59 bytes per special suffix) and the positions / records of the sorted items have done their work: a
        // read set whose module workspace is a large share of the HBM gives it back before the key ranges and the later
        // stages allocate theirs (they have no host path to fall back to); small arenas stay for the next build.
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (c->sx.cap + c->sppos.cap + c->sprec.cap) > total_b / 16)
            for (DevBuf *b : {&c->sx, &c->sppos, &c->sprec}) { HIPCHK(c, hipFree(b->p)); b->p = nullptr; b->cap = 0; }
    }
    c->special_dev = true;
    c->st.ms_host_special = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    c->st.special_path = 2;
    c->st.special_threads = 0;
    return DEBWT_OK;
}

// plans the ranges, sizes the range workspace, starts the host special-region module
static int sort_begin(debwt_ctx *c) {
    const u64 n = c->n;
    if (c->stage > ST_LOADED) c->stage = ST_LOADED;       // a new build: what the stages of the last one left is scratch (reclaim)
    int rc = plan_ranges(c);
    if (rc) return rc;
    u64 maxM = 0;
    for (auto &r : c->ranges) maxM = std::max(maxM, r.M);
    struct ReclaimScope { debwt_ctx *c; ~ReclaimScope() { c->reclaim_ok = false; } } rscope{c};
    c->reclaim_ok = true;                                 // (nothing of the last build's SP stage and blue sort is needed again)
    if (!c->exchange) ENSURE(c, c->keysA, maxM * 8 + 64);          // exchange mode: the received keys are buffer A
    ENSURE(c, c->keysB, maxM * 8 + 64);
    ENSURE(c, c->rs_skew, (maxM / 2048 + 2) * 4);
    ENSURE(c, c->rs_rle, radix_rle_ws_bytes(maxM));
    ENSURE(c, c->dk, maxM * 8 + 64);
    ENSURE(c, c->dstart, maxM * 4 + 64);
    ENSURE(c, c->pflag, maxM + 64);                      // classification byte per distinct key of a range
    ENSURE(c, c->mchar, c->Mctx + 64);
    c->reclaim_ok = false;
    c->Q = c->B = c->nlarge = 0; c->nfacts_acc = 0; c->n1024 = 0; c->n512 = 0; c->Dsum = 0;
    c->st.sort_unfit_stretches = c->st.sort_unfit_network = c->st.sort_over_stretches = 0;
    const size_t P = c->ranges.size();
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    // the keys (node << 2 | pred) are read off the text inside the first radix pass: no unsorted key array
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    // special-region module (src/collect#$.c:118-157,348-602): on the device for collections of many records, else on the
    // host beside the GPU's key sort
    join_special(c);
    c->special_dev = false;
    if (special_wants_device(c)) {
        if ((rc = special_device_build(c))) return rc;
    } else {
    c->special_running = true;
    c->special_thread = std::thread([c, n]() {
        auto t0 = std::chrono::steady_clock::now();
        build_special_tables(c->h_text, n, c->h_sep.data(), c->nrec, c->K, &c->special);
        c->st.ms_host_special = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        c->st.special_threads = c->special.threads_used;
        c->st.special_path = c->special.threads_used > 1 ? 1 : 0;
    });
    }
    // several ranges read off the text: the chunk histograms of every range's first pass from ONE scan of the text (a
    // lane per text word, radix_text_hist_ranges); a single range of a long text takes the same kernel -- it is twice
    // as fast as the histogram pass of the sort itself (a shard of 8 reads the whole text for its one range)
    c->shared_hist = P <= RS_MAX_RANGES && !c->exchange &&
                     (P > 1 || (P == 1 && n >= (1ull << 26) && radix_first_shift(c->ranges[0].M, 2 * c->cfg.k, c->cfg.sort_algo) > 0));
    if (c->shared_hist) {
        const int kb = 2 * c->cfg.k;
        std::vector<u8> rob(SHARD_BINS, 0xFF);
        int shifts[RS_MAX_RANGES];
        for (size_t i = 0; i < P; i++) {
            const u32 lo = (u32)(c->ranges[i].key_lo >> (kb - 12));
            const u32 hi = c->ranges[i].key_hi ? (u32)(c->ranges[i].key_hi >> (kb - 12)) : SHARD_BINS;
            for (u32 b = lo; b < hi; b++) rob[b] = (u8)i;
            shifts[i] = radix_first_shift(c->ranges[i].M, kb, c->cfg.sort_algo);
        }
        ENSURE(c, c->dest_tab, SHARD_BINS);
        ENSURE(c, c->range_hist, P * radix_text_hist_stride() * sizeof(u32));
        HIPCHK(c, hipMemcpyAsync(c->dest_tab.p, rob.data(), SHARD_BINS, hipMemcpyHostToDevice, c->stream));
        TextKeySrc all{c->text.as<u64>(), c->sepbits.as<u64>(), n, c->K, 0, 0, 0, nullptr, nullptr, 0};
        hipError_t e = radix_text_hist_ranges(c->stream, all, c->dest_tab.as<u8>(), kb, shifts, (int)P, c->range_hist.as<u32>());
        if (e != hipSuccess) { c->err = std::string("range histograms: ") + hipGetErrorString(e); return DEBWT_EDEVICE; }
        HIPCHK(c, hipStreamSynchronize(c->stream));          // rob is host memory
    }
    return DEBWT_OK;
}

// sorts, run-length encodes and (unless it is the only range of a text-fed build) classifies range i.
// imported: the range's keys in a caller-owned DEVICE buffer (exchange mode: it serves as key buffer A and must stay
// valid until the next range or debwt_shard_sort_end), else the keys are read off the text.
static int sort_range(debwt_ctx *c, size_t i, u64 *imported) {
    const u64 n = c->n;
    const size_t P = c->ranges.size();
    debwt_ctx::KeyRange &r = c->ranges[i];
    int rc;
    c->M = r.M;
    c->sort_a = imported ? imported : c->keysA.as<u64>();
    c->sort_b = c->keysB.as<u64>();
    TextKeySrc ts{c->text.as<u64>(), c->sepbits.as<u64>(), n, c->K, r.key_lo, r.key_hi, 0,
                  c->shared_hist ? c->range_hist.as<u32>() + i * radix_text_hist_stride() : nullptr, nullptr, 0};
    // the bucket finish of the sort counts the distinct keys of its tiles and the encoding follows tile by tile
    // (tune bit 8 = 256: separate count and emit passes over the sorted keys instead)
    RleSink sink{c->dk.as<u64>(), c->dstart.as<u32>(), c->mchar.as<u8>() + r.Mbase, c->rs_rle.p, &c->h_scalars[0],
                 &c->h_scalars[32], 0, (c->cfg.reserved & 131072) != 0,           // bit 17: no staging of distinct keys
                 // the sorted keys can be asked for (debwt_fetch_array) after the sort stage of a one-range build driven
                 // stage by stage on one GPU, and by nobody else: every other build keeps the encoding only (tune bit 23: keeps both -- A/B)
                 (P > 1 || c->exchange || c->whole_build || c->shard_world > 1) && !(c->cfg.reserved & 8388608), false};
    if (imported && r.M < 2) c->sk = c->sort_a;
    else if ((rc = sort_keys(c, c->sort_a, c->sort_b, r.M, 2 * c->cfg.k, &c->sk, i == 0, imported ? nullptr : &ts, true,
                             (c->cfg.reserved & 256) ? nullptr : &sink))) return rc;
    if (P == 1 && !c->exchange) HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    if (!sink.done) {
        RleF f{c->sk, c->dk.as<u64>(), c->dstart.as<u32>(), c->mchar.as<u8>() + r.Mbase};
        if ((rc = cp_count(c, f, r.M, cp_area(c, 0), 0))) return rc;
        if ((rc = cp_emit(c, f, r.M, cp_area(c, 0)))) return rc;
    }
    if (i == 0 && !c->special_dev) {
        join_special(c);
        const SpecialTables &sp = c->special;
        c->nbranch = sp.branch.size();
        ENSURE(c, c->branch, sp.branch.size() * 8 + 64);
        HIPCHK(c, hipMemcpyAsync(c->head_keys.p, sp.head_keys.data(), c->nrec * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->spkey.p, sp.key.data(), c->NS * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->spchr.p, sp.chr.data(), c->NS, hipMemcpyHostToDevice, c->stream));
        if (!sp.branch.empty())
            HIPCHK(c, hipMemcpyAsync(c->branch.p, sp.branch.data(), sp.branch.size() * 8, hipMemcpyHostToDevice, c->stream));
        if ((rc = special_branch_bitmap(c))) return rc;
    }
    // special suffixes whose key lies in this range, and their rows among the context's instances
    u64 *d_bounds = nullptr;
    {
        const std::vector<uint64_t> &key = c->special.key;          // ascending (suffix order implies key order)
        u64 s0 = 0, s1 = c->NS;
        if ((c->shard_world > 1 || P > 1) && c->special_dev) {
            d_bounds = c->dollar.as<u64>() + 5;                       // spare words behind the '$' row and the census
            k_sp_bounds<<<1, 64, 0, c->stream>>>(c->spkey.as<u64>(), c->NS, r.key_lo >> 2, r.key_hi >> 2, d_bounds);
            HIPCHK(c, hipMemcpyAsync(&c->h_scalars[24], d_bounds, 16, hipMemcpyDeviceToHost, c->stream));
        } else if (c->shard_world > 1 || P > 1) {
            s0 = std::lower_bound(key.begin(), key.end(), r.key_lo >> 2) - key.begin();
            s1 = r.key_hi ? (u64)(std::lower_bound(key.begin(), key.end(), r.key_hi >> 2) - key.begin()) : c->NS;
        }
        r.s0 = s0; r.s1 = s1;
    }
    if ((rc = sync_check(c))) return rc;
    if (d_bounds) memcpy(&r.s0, &c->h_scalars[24], 8), memcpy(&r.s1, &c->h_scalars[26], 8);
    c->D = c->h_scalars[0];
    c->Dsum += c->D;
    if (sink.done) {
        c->st.sort_unfit_stretches += c->h_scalars[32]; c->st.sort_unfit_network += c->h_scalars[35];
        c->st.sort_over_stretches += sink.n_over;
    }
    if (r.s1 > r.s0)
        k_special_rows<<<grid_for(r.s1 - r.s0, 256), 256, 0, c->stream>>>(
            c->dk.as<u64>(), c->dstart.as<u32>(), c->D, r.M, c->spkey.as<u64>() + r.s0, r.s1 - r.s0,
            r.Mbase + (r.s0 - c->ranges[0].s0), c->sprow.as<u64>() + r.s0);
    if (P > 1 || c->exchange) {
        // the range's keys are gone after this call: classify them now
        if ((rc = classify_local(c))) return rc;
        if ((rc = append_range(c, r))) return rc;
    }
    return DEBWT_OK;
}

static int sort_end(debwt_ctx *c) {
    c->s0 = c->ranges.front().s0; c->s1 = c->ranges.back().s1;
    if (c->ranges.size() > 1 || c->exchange) {
        c->local_done = true;
        HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
        HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
    }
    c->st.distinct_keys = c->Dsum;
    c->st.special_branch_num = c->nbranch;
    c->stage = ST_SORTED;
    return DEBWT_OK;
}

extern "C" int debwt_kmer_sort_rle(debwt_ctx *c) {
    if (!c) return DEBWT_EINVAL;
    if (c->stage < ST_LOADED) return DEBWT_ESTATE;
    if (c->exchange) { c->err = "exchange-mode shard: use debwt_shard_sort_begin/_range/_end"; return DEBWT_ESTATE; }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc = sort_begin(c);
    for (size_t i = 0; !rc && i < c->ranges.size(); i++) rc = sort_range(c, i, nullptr);
    join_special(c);
    if (rc) return rc;
    return sort_end(c);
}

// ---------------------------------------------------------------------------------------------------
// stage 2: classification                                                                      (a-8)

// local half: classification of this shard's distinct keys -> its fact lists [multi-out | multi-in] and its
// block table (mi_j0, mi_freq, bstart)
static int classify_local(debwt_ctx *c) {
    const u64 M = c->M, D = c->D, nrec = c->nrec;
    int rc;
    if (c->ranges.size() == 1) HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
    ClassifyCommon cc{c->dk.as<u64>(), c->dstart.as<u32>(), D, M, c->head_keys.as<u64>(), nrec};
    // pflag (n bytes) is free until the SP stage: it holds the per-distinct-key classification byte
    u8 *cf = c->pflag.as<u8>();
    ClassifyFlagsF ff{cc, c->K, cf};
    {
        u32 nchunks; u64 chunk;
        plan_chunks(D, &nchunks, &chunk);
        u32 *ca = cp_area(c, 1), *cb = cp_area(c, 2);
        u32 *wl = reinterpret_cast<u32 *>(c->sk == c->sort_a ? c->sort_b : c->sort_a);   // the sort's scratch buffer: >= 8 M bytes
        u32 *wl_count = cp_area(c, 7);
        if (D) {
            k_classify_flags<<<nchunks, DEBWT_BLOCK, 0, c->stream>>>(ff, chunk, ca, cb, wl, wl_count);
            k_classify_groups<<<nchunks, DEBWT_BLOCK, 0, c->stream>>>(ff, chunk, ca, cb, wl, wl_count);
        } else { HIPCHK(c, hipMemsetAsync(ca, 0, sizeof(u32), c->stream)); HIPCHK(c, hipMemsetAsync(cb, 0, sizeof(u32), c->stream)); }
        cp_scan_kernel<<<1, 1024, 0, c->stream>>>(ca, D ? nchunks : 1, ca + CP_MAXCHUNKS);
        cp_scan_kernel<<<1, 1024, 0, c->stream>>>(cb, D ? nchunks : 1, cb + CP_MAXCHUNKS);
        HIPCHK(c, hipMemcpyAsync(&c->h_scalars[1], ca + CP_MAXCHUNKS, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(&c->h_scalars[2], cb + CP_MAXCHUNKS, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    }
    if ((rc = sync_check(c))) return rc;
    c->rQ = c->h_scalars[1];
    c->Rmo = c->h_scalars[2];
    const u64 Q = c->rQ, Rmo = c->Rmo;
    ENSURE(c, c->facts, (Rmo + Q) * 8 + 64);
    ENSURE(c, c->mi_j0, Q * 4 + 64);
    ENSURE(c, c->mi_freq, Q * 4 + 64);
    ENSURE(c, c->bstart, Q * 4 + 64);
    ENSURE(c, c->large_tmp, Q * 4 + 64);
    u64 *facts = c->facts.as<u64>();
    {
        const u64 nslots = Q + Rmo;
        ENSURE(c, c->fact_work, nslots * 16 + 64);
        HIPCHK(c, hipMemsetAsync(c->fact_work.p, 0, nslots * 16 + 16, c->stream));
        FactEmitArgs fa{cc, c->K, cf, facts + Rmo, c->mi_j0.as<u32>(), c->mi_freq.as<u32>(), facts,
                        c->fact_work.as<uint4>()};
        u32 nchunks; u64 chunk;
        plan_chunks(D, &nchunks, &chunk);       // same chunks as the counting sweep (multiples of 1024 keys)
        if (D) k_emit_facts<<<nchunks, DEBWT_BLOCK, 0, c->stream>>>(fa, chunk, cp_area(c, 1), cp_area(c, 2));
        if (nslots) k_eval_facts<<<grid_for(nslots, DEBWT_BLOCK), DEBWT_BLOCK, 0, c->stream>>>(fa, nslots);
    }
    BlockStartF fb{c->mi_freq.as<u32>(), c->bstart.as<u32>()};
    if ((rc = cp_count(c, fb, Q, cp_area(c, 4), 4))) return rc;
    if ((rc = cp_emit(c, fb, Q, cp_area(c, 4)))) return rc;
    LargeBlockF fl{c->mi_freq.as<u32>(), BLUE_LDS_CAP, c->large_tmp.as<u32>()};
    if ((rc = cp_count(c, fl, Q, cp_area(c, 5), 5))) return rc;
    if ((rc = cp_emit(c, fl, Q, cp_area(c, 5)))) return rc;
    {
        u32 *cnt = cp_area(c, 6) + CP_MAXCHUNKS + 8;              // two free words behind the area's scan total
        HIPCHK(c, hipMemsetAsync(cnt, 0, 8, c->stream));
        if (Q) k_count_blocks<<<std::min<u32>(grid_for(Q, 256), 512u), 256, 0, c->stream>>>(c->mi_freq.as<u32>(), Q, 512u, 1024u, cnt);
        if (Q) k_count_blocks<<<std::min<u32>(grid_for(Q, 256), 512u), 256, 0, c->stream>>>(c->mi_freq.as<u32>(), Q, 256u, 512u, cnt + 1);
        HIPCHK(c, hipMemcpyAsync(&c->h_scalars[11], cnt, 8, hipMemcpyDeviceToHost, c->stream));
    }
    if ((rc = sync_check(c))) return rc;
    c->rB = c->h_scalars[4];
    c->rnlarge = c->h_scalars[5];
    c->rn1024 = c->h_scalars[11];
    c->rn512 = c->h_scalars[12];
    return DEBWT_OK;
}

// appends the range just classified (facts, block tables, large-block list) to the context-wide tables
static int append_range(debwt_ctx *c, debwt_ctx::KeyRange &r) {
    const u64 Q = c->rQ, nf = c->Rmo + Q;
    r.Q = Q; r.qbase = c->Q; r.B = c->rB; r.Bbase = c->B; r.l0 = c->nlarge; r.nl = c->rnlarge;
    ENSURE_KEEP(c, c->facts_acc, (c->nfacts_acc + nf) * 8 + 64, c->nfacts_acc * 8);
    ENSURE_KEEP(c, c->blk_j0, (c->Q + Q) * 8 + 64, c->Q * 8);
    ENSURE_KEEP(c, c->blk_freq, (c->Q + Q) * 4 + 64, c->Q * 4);
    ENSURE_KEEP(c, c->blk_start, (c->Q + Q) * 8 + 64, c->Q * 8);
    ENSURE_KEEP(c, c->large_q, (c->nlarge + c->rnlarge) * 4 + 64, c->nlarge * 4);
    if (nf) HIPCHK(c, hipMemcpyAsync(c->facts_acc.as<u64>() + c->nfacts_acc, c->facts.p, nf * 8, hipMemcpyDeviceToDevice, c->stream));
    if (Q)
        k_append_blocks<<<grid_for(Q, 256), 256, 0, c->stream>>>(c->mi_j0.as<u32>(), c->mi_freq.as<u32>(), c->bstart.as<u32>(), Q,
                                                                 r.Mbase, r.Bbase, c->blk_j0.as<u64>() + r.qbase,
                                                                 c->blk_freq.as<u32>() + r.qbase, c->blk_start.as<u64>() + r.qbase);
    if (c->rnlarge) {
        HIPCHK(c, hipMemcpyAsync(c->large_q.as<u32>() + c->nlarge, c->large_tmp.p, c->rnlarge * 4, hipMemcpyDeviceToDevice, c->stream));
        if (r.qbase)
            k_offset_u32<<<grid_for(c->rnlarge, 256), 256, 0, c->stream>>>(c->large_q.as<u32>() + c->nlarge, c->rnlarge, (u32)r.qbase);
    }
    c->nfacts_acc += nf; c->Q += Q; c->B += c->rB; c->nlarge += c->rnlarge; c->n1024 += c->rn1024; c->n512 += c->rn512;
    if (c->Q >= 0xFFFFFFF0ull) { c->err = "more than 2^32 multi-in blocks"; return DEBWT_ERANGE; }
    c->facts_ready = true;
    return sync_check(c);
}

// global half: the red table from the facts of ALL shards (d_facts: device, nfacts words, any order) plus the
// tail# facts; identical on every shard
static int classify_global(debwt_ctx *c, const u64 *d_facts, u64 nfacts, u64 qbase, u64 btotal) {
    const u64 nrec = c->nrec, Q = c->Q;
    int rc;
    const u64 nf = nfacts + nrec;
    c->nfacts = nf;
    c->qbase = qbase;
    c->Btotal = btotal;
    ENSURE(c, c->facts_all, nf * 8 + 64);
    ENSURE(c, c->facts_tmp, nf * 8 + 64);
    ENSURE(c, c->rs_skew, (nf / 2048 + 2) * 4);
    ENSURE(c, c->red, nf * 8 + 64);
    ENSURE(c, c->red_q, nf * 4 + 64);
    u64 *all = c->facts_all.as<u64>();
    if (nfacts) HIPCHK(c, hipMemcpyAsync(all, d_facts, nfacts * 8, hipMemcpyDeviceToDevice, c->stream));
    if (c->special_dev) HIPCHK(c, hipMemcpyAsync(all + nfacts, c->tail_d.p, nrec * 8, hipMemcpyDeviceToDevice, c->stream));
    else HIPCHK(c, hipMemcpyAsync(all + nfacts, c->special.tail_facts.data(), nrec * 8, hipMemcpyHostToDevice, c->stream));
    u64 *sorted_facts = nullptr;
    if ((rc = sort_keys(c, all, c->facts_tmp.as<u64>(), nf, 2 * c->cfg.k, &sorted_facts, false))) return rc;
    RedUniqueF fr{sorted_facts, nf, c->red.as<u64>()};
    if ((rc = cp_count(c, fr, nf, cp_area(c, 3), 3))) return rc;
    if ((rc = cp_emit(c, fr, nf, cp_area(c, 3)))) return rc;
    if ((rc = sync_check(c))) return rc;
    c->R = c->h_scalars[3];
    const u64 R = c->R;
    RedBlockF fq{c->red.as<u64>(), c->red_q.as<u32>()};
    if ((rc = cp_count(c, fq, R, cp_area(c, 6), 6))) return rc;
    if ((rc = cp_emit(c, fq, R, cp_area(c, 6)))) return rc;
    ENSURE(c, c->blue, c->B * 8 + 64);
    if ((rc = sync_check(c))) return rc;
    if (c->shard_world == 1 && c->h_scalars[6] != Q) {
        c->err = "multi-in count mismatch between fact list and red table";
        return DEBWT_EINTERNAL;
    }
    c->st.red_capacity = R; c->st.blue_capacity = c->B; c->st.blue_bound_num = Q; c->st.case3num = 2 * Q;
    c->st.blue_large_blocks = c->nlarge;
    c->Qtotal = c->h_scalars[6];
    c->stage = ST_CLASSIFIED;
    return DEBWT_OK;
}

extern "C" int debwt_classify(debwt_ctx *c) {
    if (!c) return DEBWT_EINVAL;
    if (c->stage < ST_SORTED) return DEBWT_ESTATE;
    if (c->shard_world > 1) { c->err = "sharded context: use debwt_shard_classify_local/_global"; return DEBWT_ESTATE; }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc;
    if (!c->local_done) {
        c->Q = c->B = c->nlarge = 0; c->nfacts_acc = 0; c->n1024 = 0; c->n512 = 0;      // a repeated call starts over
        if ((rc = classify_local(c))) return rc;
        if ((rc = append_range(c, c->ranges[0]))) return rc;
    }
    return classify_global(c, c->facts_acc.as<u64>(), c->nfacts_acc, 0, c->B);
}

// ---------------------------------------------------------------------------------------------------
// stage 3: SP code and blue entries                                                            (a-4)

// SP stage in three steps so that a sharded build can cut it at the exchanges:
//   sp_flags   node table + flags of the text groups [g0, g1) + their multi-out / multi-in counts
//   sp_emit    SP symbols of the slice at their global offset, work list of the slice's multi-in positions
//   sp_finish  4-bit packed SP code of the WHOLE text (after the slices' symbols were all-gathered)
// node table + prefilter from the red table; flag masks sized for the whole text
static int sp_prepare(debwt_ctx *c) {
    HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
    // (the range workspace of the key sort is idle from here on: an allocation of this function that fails may release it)
    struct ReclaimScope { debwt_ctx *c; ~ReclaimScope() { c->reclaim_ok = false; } } rscope{c};
    c->reclaim_ok = true;
    const u64 ngroups = (c->n + 31) >> 5;
    ENSURE(c, c->momask, ngroups * 4 + 64);
    ENSURE(c, c->mimask, ngroups * 4 + 64);
    // node table: 4..8 slots per red node; prefilter: ~8 bits per red node (tuning knob: reserved = delta+8)
    // (4..8 slots per red node: at 2..4 a collection whose red nodes happened to fill the table by half -- eight genomes --
    // spent 14 % more on the SP flags than one that filled it by a quarter -- ten --: 329 -> 281 ms of SP stage for 24 Gbp;
    // ten genomes 364 -> 349, with an Alu-like family 586 -> 540, one genome 32.9 -> 28.6; 8..16 slots: 3 ms more for 30 Gbp)
    int hbits = 10;
    u64 slots_per_node = 4;
    if (const char *e = getenv("DEBWT_NODE_SLOTS")) slots_per_node = std::max<u64>(2, strtoull(e, nullptr, 10));   // A/B
    while ((1ull << hbits) < slots_per_node * c->R) hbits++;
    if (hbits > 32 && (1ull << 32) >= 2 * c->R) hbits = 32;      // (slot numbers are 32-bit: 2..4 per node where 4..8 do not fit --
                                                                 //  k = 16 on a 3.1 Gbp text: 1.07 G branching 15-mers, a 64 GB table)
    // prefilter.  K >= 24: 64-bit words chosen by the node's minimizer, ~2 red nodes per word (stage_kernels.h,
    // k_build_mzfilter) -- a lane that walks 32 consecutive positions fetches ~4 words instead of probing 32 times.
    // Smaller K (or cfg.reserved bit 11: tests): one bit per hashed node, 8 bits per red node; while the node table
    // still fits the Infinity Cache (<= 256 MB) the bitmap is held at L2 size (2 MB, down to 2..4 bits per node).
    // cfg.reserved & 15 = delta + 8 overrides the size (tuning).
    // The plain bitmap is kept for small node tables (below 2^20 slots: a few Mbp); from there on the minimizer filter
    // wins (250 Mbp: SP stage 3.16 ms against 3.41 ms).  cfg.reserved bit 12 forces it for any size (tests).
    // (nodes of 16..23 symbols take minimizers of 12 symbols: k = 17..24 -- round 6)
    c->mzfilter = c->K >= 16 && !(c->cfg.reserved & 2048) && (hbits >= 20 || (c->cfg.reserved & 4096));
    c->mzw = c->K >= 24 ? MZ_W : MZ_W_SHORT;
    c->mztable = c->mzfilter && c->K >= 24 && (c->cfg.reserved & 16384);   // cfg.reserved bit 14: node table addressed by minimizer (A/B, tests)
    int pb;
    if (c->mzfilter) {
        pb = 10;
        while ((2ull << pb) < c->R) pb++;
        if (c->cfg.reserved & 15) pb += (c->cfg.reserved & 15) - 8;
        if (pb < 10) pb = 10;
        if (pb > 30) pb = 30;
    } else {
        pb = (c->cfg.reserved & 15) ? hbits + (c->cfg.reserved & 15) - 8
                                     : (hbits <= 24 ? std::max(hbits, std::min(hbits + 3, 24)) : hbits + 3);
        if (pb < 10) pb = 10;
        if (pb > 31) pb = 31;                                     // (a 256 MB bitmap: nearly every node of so small a K branches anyway)
    }
    if (pb > 31 || hbits > 32) { c->err = "more than 2^31 branching nodes: the node table has 2^32 slots"; return DEBWT_ERANGE; }
    c->hbits = hbits; c->pbits = pb;
    // absolute 32-bit fill cursors unless the context's blue slots need more bits (cfg.reserved bit 4 forces the
    // 64-bit form: tests)
    c->abs32 = c->B < 0xFFFFFFF0ull && !(c->cfg.reserved & 16);
    size_t rb_bytes = (c->mzfilter ? ((size_t)8 << pb) : ((size_t)1 << pb) / 8) + 64, ht_slots = (size_t)1 << hbits;
    ENSURE(c, c->rbits, rb_bytes);
    ENSURE(c, c->htab, ht_slots * sizeof(HSlot));
    HIPCHK(c, hipMemsetAsync(c->rbits.p, 0, rb_bytes, c->stream));
    HIPCHK(c, hipMemsetAsync(c->htab.p, 0, ht_slots * sizeof(HSlot), c->stream));
    if (c->R) {
        k_build_hash<<<grid_for(c->R, 256), 256, 0, c->stream>>>(c->red.as<u64>(), c->R, c->red_q.as<u32>(),
                                                                c->blk_start.as<u64>(), c->abs32 ? 1 : 0, (u32)c->qbase,
                                                                (u32)c->Q, hbits, c->htab.as<HSlot>(), c->mzfilter ? 0 : pb,
                                                                c->rbits.as<u32>(), c->mztable ? c->K : 0);
        if (c->mzfilter)
            k_build_mzfilter<<<grid_for(c->R, 256), 256, 0, c->stream>>>(c->red.as<u64>(), c->R, c->K, pb, c->rbits.as<u64>(), c->mzw);
    }
    return DEBWT_OK;
}

// SP stage of the text groups [g0, g1) (a slice of fewer than 2^32 positions: the compaction counters are 32-bit):
//   sp_flags   flags of the slice + its multi-out / multi-in counts
//   sp_emit    SP symbols of the slice at their global offset, work list of the slice's multi-in positions
//   sp_finish  packed SP code of the WHOLE text (after all slices / after the slices' symbols were all-gathered)
#define SP_SLICE_GROUPS (1ull << 26)
static SpBlockIds sp_block_ids(debwt_ctx *c) {
    if (!c->route_direct) return SpBlockIds{nullptr, nullptr, nullptr, 0};
    return SpBlockIds{c->qlist.as<u32>(), c->qwave.as<unsigned long long>(), c->qwave.as<u64>() + 1, c->gq0};
}
// room for the block ids of the multi-in positions among `groups` text groups that start at group gq0
static int sp_block_ids_begin(debwt_ctx *c, u64 gq0, u64 groups) {
    c->route_direct = true; c->gq0 = gq0;
    ENSURE(c, c->qlist, (std::min<u64>(groups * 32, c->Btotal) + 64) * 4);
    ENSURE(c, c->qwave, ((groups + 63) / 64 + 2) * 8);
    HIPCHK(c, hipMemsetAsync(c->qwave.p, 0, 8, c->stream));
    return DEBWT_OK;
}

static int sp_flags(debwt_ctx *c, u64 g0, u64 g1) {
    int rc;
    if (g1 - g0 > (1ull << 27) - 2) { c->err = "text slice of the SP stage exceeds 2^32 positions"; return DEBWT_ERANGE; }
    c->g0 = g0; c->g1 = g1;
    const u64 ng = g1 - g0;
    if (ng && c->mzfilter && c->mztable)
        k_sp_flags<2><<<grid_for(ng, DEBWT_BLOCK), DEBWT_BLOCK, 0, c->stream>>>(
            c->text.as<u64>(), c->sepbits.as<u64>(), c->n, c->K, c->htab.as<HSlot>(), c->hbits, c->rbits.as<u32>(), c->pbits,
            c->branch.as<u64>(), c->nbranch, c->momask.as<u32>(), c->mimask.as<u32>(), g0, g1,
            sp_block_ids(c), c->branch_bitmap ? c->brbits.as<u64>() : nullptr);
    else if (ng && c->mzfilter)
        k_sp_flags<1><<<grid_for(ng, DEBWT_BLOCK), DEBWT_BLOCK, 0, c->stream>>>(
            c->text.as<u64>(), c->sepbits.as<u64>(), c->n, c->K, c->htab.as<HSlot>(), c->hbits, c->rbits.as<u32>(), c->pbits,
            c->branch.as<u64>(), c->nbranch, c->momask.as<u32>(), c->mimask.as<u32>(), g0, g1,
            sp_block_ids(c), c->branch_bitmap ? c->brbits.as<u64>() : nullptr, c->mzw);
    else if (ng)
        k_sp_flags<0><<<grid_for(ng, DEBWT_BLOCK), DEBWT_BLOCK, 0, c->stream>>>(
            c->text.as<u64>(), c->sepbits.as<u64>(), c->n, c->K, c->htab.as<HSlot>(), c->hbits, c->rbits.as<u32>(), c->pbits,
            c->branch.as<u64>(), c->nbranch, c->momask.as<u32>(), c->mimask.as<u32>(), g0, g1,
            sp_block_ids(c), c->branch_bitmap ? c->brbits.as<u64>() : nullptr);
    SpCountF fc{c->momask.as<u32>() + g0, c->mimask.as<u32>() + g0};
    if ((rc = cp_count2(c, fc, ng, cp_area(c, 0), 8, cp_area(c, 1), 10))) return rc;
    if ((rc = sync_check(c))) return rc;
    c->S_local = c->h_scalars[8];
    c->B_slice = c->h_scalars[10];
    return DEBWT_OK;
}

// routed: nullptr -> work list of the multi-in positions (mi_list); else the slice's B_slice routed blue entries go there
static int sp_emit(debwt_ctx *c, u64 sp_off, u64 *routed = nullptr, int qshift = 0) {
    const u64 ng = c->g1 - c->g0;
    c->sp_off = sp_off;
    // the SP symbol buffer grows with the slices (S is ~0.1 n, the worst case n)
    ENSURE_KEEP(c, c->spsym, sp_off + c->S_local + 64, c->shard_world > 1 ? 0 : sp_off);
    if (!routed) ENSURE(c, c->mi_list, c->B_slice * 16 + 64);
    if (ng) {
        SpEmitArgs ea{c->text.as<u64>(), c->sepbits.as<u64>(), c->n, c->K, c->momask.as<u32>(), c->mimask.as<u32>(),
                      c->spsym.as<u8>(), routed ? nullptr : c->mi_list.as<ulonglong2>(), c->g0, sp_off,
                      routed, qshift, sp_block_ids(c)};
        u32 nchunks; u64 chunk;
        plan_chunks(ng, &nchunks, &chunk);
        k_sp_emit<<<nchunks, DEBWT_BLOCK, 0, c->stream>>>(ea, ng, chunk, cp_area(c, 0), cp_area(c, 1));
    }
    return DEBWT_OK;
}

static int sp_finish(debwt_ctx *c, u64 S) {
    c->S = S;
    u64 ntriples = (S >> 6) + 3;                 // 64 symbols = 3 words; two spare groups of zeros behind the end
    ENSURE(c, c->spn, ntriples * 24);
    k_pack_sp<<<grid_for(ntriples, 256), 256, 0, c->stream>>>(c->spsym.as<u8>(), S, ntriples, c->spn.as<u64>());
    c->st.sp_len = S;
    c->stage = ST_SP;
    return DEBWT_OK;
}

extern "C" int debwt_sp_generate(debwt_ctx *c) {
    if (!c) return DEBWT_EINVAL;
    if (c->stage < ST_CLASSIFIED) return DEBWT_ESTATE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc;
    // the whole text on this context (also every shard of a "replicated scan" sharded build), slice by slice
    if ((rc = sp_prepare(c))) return rc;
    const u64 ngroups = (c->n + 31) >> 5;
    const u64 slice_groups = (c->cfg.reserved & 4194304) ? 4096ull : SP_SLICE_GROUPS;   // bit 22: slices of 2^17 positions (tests)
    u64 S = 0, Bseen = 0;
    // Blue fill without atomics when a routed entry (block id | SP index | pred) fits 64 bits: the entries of all
    // slices are collected in `blue`, sorted by block id, stripped.  Otherwise (or cfg.reserved bit 5: tests) one
    // cursor atomic per entry.
    int qbits = 1;
    while ((1ull << qbits) < c->Q) qbits++;
    const int qshift = 64 - qbits;
    const bool route_sort = c->shard_world == 1 && c->Q > 0 && c->n < (1ull << (qshift - 3)) &&
                            c->B < 0xFFFFFFF0ull - (1ull << 20) &&      // the radix passes index with 32 bits
                            !(c->cfg.reserved & 32) &&
                            !((c->cfg.reserved & 262144) && c->ranges.size() > 1);   // bit 18: by key range, as for >= 2^32 entries (tests)
    // 2^32 blue entries and more (ten genomes with an Alu-like family: 5.6 G) are beyond the 32-bit positions of the radix
    // passes as ONE array, but the blocks -- and so the blue slots -- of a key range are contiguous and fewer than 2^32: the
    // routed entries of a slice are bucketed by key range (the pass that routes entries to their shard on several GPUs) and
    // appended to their range's slots; every range is then sorted by block id on its own, over the bits in which its block
    // ids differ.  (The cursor-atomic fill this replaces took 2.3 s of a 5.0 s build, profiles/r04_bench_real10x3G_first.json.)
    const size_t P = c->ranges.size();
    bool route_ranges = !route_sort && c->shard_world == 1 && P > 1 && P <= RS_RADIX && c->Q > 0 && c->n < (1ull << (qshift - 3)) &&
                        !(c->cfg.reserved & 32);
    for (size_t i = 0; route_ranges && i < P; i++) route_ranges = c->ranges[i].B < 0xFFFFFFF0ull - (1ull << 20);
    std::vector<u64> rfill(P, 0), roffs(P + 1, 0);
    if (route_ranges) {
        std::vector<u32> qb(P + 1);
        for (size_t i = 0; i < P; i++) qb[i] = (u32)c->ranges[i].qbase;
        qb[P] = (u32)c->Q;
        ENSURE(c, c->qbounds, (P + 1) * 4);
        HIPCHK(c, hipMemcpyAsync(c->qbounds.p, qb.data(), (P + 1) * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));              // qb is host memory
    }
    c->route_direct = false;
    for (u64 g0 = 0; g0 < ngroups; g0 += slice_groups) {
        const u64 g1 = std::min(ngroups, g0 + slice_groups);
        // route_sort: pass 1 keeps the block ids of the slice's multi-in positions, pass 2 writes the routed entries
        if ((route_sort || route_ranges) && (rc = sp_block_ids_begin(c, g0, g1 - g0))) return rc;
        if ((rc = sp_flags(c, g0, g1))) return rc;
        if (route_sort) {
            if (Bseen + c->B_slice > c->B) { c->err = "multi-in positions exceed the block total"; return DEBWT_EINTERNAL; }
            if ((rc = sp_emit(c, S, c->blue.as<u64>() + Bseen, qshift))) return rc;
        } else if (route_ranges) {
            // the slice's entries as emitted, and bucketed: in the key buffers (free since the ranges were classified) when
            // those are large enough
            u64 *ta = c->keysA.as<u64>(), *tb = c->keysB.as<u64>();
            size_t tb_cap = c->keysB.cap / 8;
            if (c->keysA.cap < c->B_slice * 8 + 64) { ENSURE(c, c->blue_tmp, c->B_slice * 8 + 64); ta = c->blue_tmp.as<u64>(); }
            if (c->keysB.cap < c->B_slice * 8 + 64) { ENSURE(c, c->mi_list, c->B_slice * 8 + 64); tb = c->mi_list.as<u64>(); tb_cap = c->mi_list.cap / 8; }
            if ((rc = sp_emit(c, S, ta, qshift))) return rc;
            if (c->B_slice) {
                ENSURE(c, c->rs_over, radix_over_bytes(c->B_slice));
                RsDigit dg{};
                dg.mode = 2; dg.bounds = c->qbounds.as<u32>(); dg.nb = (u32)P; dg.tshift = qshift;
                hipError_t e = radix_partition_by_shard(c->stream, ta, nullptr, c->B_slice, tb, dg, (u32)P, radix_ws(c), roffs.data(),
                                                        false, tb_cap);
                if (e != hipSuccess) { c->err = std::string("blue entries by key range: ") + hipGetErrorString(e); return DEBWT_EDEVICE; }
                for (size_t i = 0; i < P; i++) {
                    const u64 cnt = roffs[i + 1] - roffs[i];
                    if (!cnt) continue;
                    if (rfill[i] + cnt > c->ranges[i].B) { c->err = "multi-in positions exceed a key range's block total"; return DEBWT_EINTERNAL; }
                    HIPCHK(c, hipMemcpyAsync(c->blue.as<u64>() + c->ranges[i].Bbase + rfill[i], tb + roffs[i], cnt * 8,
                                             hipMemcpyDeviceToDevice, c->stream));
                    rfill[i] += cnt;
                }
            }
        } else if ((rc = sp_emit(c, S))) return rc;
        if (c->B_slice && !route_sort && !route_ranges) {
            if (c->abs32)
                k_blue_fill<1><<<grid_for(c->B_slice, DEBWT_BLOCK), DEBWT_BLOCK, 0, c->stream>>>(
                    c->mi_list.as<ulonglong2>(), c->B_slice, c->htab.as<HSlot>(), c->hbits, c->blk_start.as<u64>(),
                    (u32)c->qbase, c->blue.as<u64>(), c->mztable ? c->K : 0);
            else
                k_blue_fill<0><<<grid_for(c->B_slice, DEBWT_BLOCK), DEBWT_BLOCK, 0, c->stream>>>(
                    c->mi_list.as<ulonglong2>(), c->B_slice, c->htab.as<HSlot>(), c->hbits, c->blk_start.as<u64>(),
                    (u32)c->qbase, c->blue.as<u64>(), c->mztable ? c->K : 0);
        }
        S += c->S_local; Bseen += c->B_slice;
    }
    if (Bseen != c->Btotal) { c->err = "multi-in positions differ from the block total"; return DEBWT_EINTERNAL; }
    if (route_ranges) {
        for (size_t i = 0; i < P; i++) {
            const debwt_ctx::KeyRange &r = c->ranges[i];
            if (rfill[i] != r.B) { c->err = "multi-in positions differ from a key range's block total"; return DEBWT_EINTERNAL; }
            if (!r.B) continue;
            u64 *reg = c->blue.as<u64>() + r.Bbase, *res = reg;
            const int span = r.Q > 1 ? bits_for(r.qbase ^ (r.qbase + r.Q - 1)) : 0;     // bits in which the range's block ids differ
            bool stripped = false;                                                     // (the last pass strips when it writes to `reg`)
            if (span) {
                ENSURE(c, c->rs_over, radix_over_bytes(r.B));
                u64 *tmp = c->keysA.as<u64>();
                if (c->keysA.cap < r.B * 8 + 64) { ENSURE(c, c->blue_tmp, r.B * 8 + 64); tmp = c->blue_tmp.as<u64>(); }
                hipError_t e = hipSuccess;
                // (an odd number of passes rotates through the other key buffer, so that the last one lands -- stripped -- in `reg`)
                u64 *third = c->keysB.cap >= r.B * 8 + 64 && c->keysB.as<u64>() != tmp ? c->keysB.as<u64>() : nullptr;
                res = radix_sort_bits(c->stream, reg, tmp, r.B, qshift, std::min(64, qshift + span), radix_ws(c), &e, qshift, &stripped, third);
                if (e != hipSuccess) { c->err = std::string("blue entry sort: ") + hipGetErrorString(e); return DEBWT_EDEVICE; }
            }
            if (!stripped) k_blue_strip<<<grid_for(r.B, 256), 256, 0, c->stream>>>(res, reg, r.B, qshift);
        }
    }
    if (route_sort && c->B) {
        // scratch of B words: a key buffer when it is large enough (free since the ranges were classified)
        u64 *tmp;
        if (c->keysA.cap >= c->B * 8) tmp = c->keysA.as<u64>();
        else { ENSURE(c, c->blue_tmp, c->B * 8 + 64); tmp = c->blue_tmp.as<u64>(); }
        ENSURE(c, c->rs_over, radix_over_bytes(c->B));
        hipError_t e = hipSuccess;
        bool stripped = false;                                   // (the last pass strips when it writes to `blue`)
        u64 *third = c->keysB.cap >= c->B * 8 + 64 && c->keysB.as<u64>() != tmp ? c->keysB.as<u64>() : nullptr;
        u64 *r = radix_sort_bits(c->stream, c->blue.as<u64>(), tmp, c->B, qshift, 64, radix_ws(c), &e, qshift, &stripped, third);
        if (e != hipSuccess) { c->err = std::string("blue entry sort: ") + hipGetErrorString(e); return DEBWT_EDEVICE; }
        if (!stripped) k_blue_strip<<<grid_for(c->B, 256), 256, 0, c->stream>>>(r, c->blue.as<u64>(), c->B, qshift);
    }
    return sp_finish(c, S);
}

// ---------------------------------------------------------------------------------------------------
// stage 4: blue-block sort                                                                     (a-5)

// blocks above BLUE_LDS_CAP rows (see LargeSplit)
static int bitonic_large(debwt_ctx *c, u64 b0, u32 m, u64 j0) {
    u64 P = 2;
    while (P < m) P <<= 1;
    ENSURE(c, c->large_k0, P * 8);
    ENSURE(c, c->large_en, P * 8);
    k_large_load<<<grid_for(P, 256), 256, 0, c->stream>>>(c->blue.as<u64>(), b0, m, P, c->spn.as<u64>(),
                                                          c->large_k0.as<u64>(), c->large_en.as<u64>());
    for (u64 kk = 2; kk <= P; kk <<= 1)
        for (u64 jj = kk >> 1; jj > 0; jj >>= 1)
            k_large_step<<<grid_for(P >> 1, 256), 256, 0, c->stream>>>(
                c->large_k0.as<u64>(), c->large_en.as<u64>(), P, kk, jj, c->spn.as<u64>(), c->S);
    k_large_store<<<grid_for(m, 256), 256, 0, c->stream>>>(c->blue.as<u64>(), b0, m, j0, c->large_en.as<u64>(), c->mchar.as<u8>());
    return DEBWT_OK;
}

// the blocks large_q[l0, l0 + nl)
static int sort_large_blocks(debwt_ctx *c, const BlueSub &sub, u64 l0, u64 nl) {
    int rc;
    // descriptors of the large blocks: one gather, one copy
    std::vector<u64> desc3(3 * nl);
    ENSURE(c, c->large_k0, 3 * nl * 8);
    k_large_gather<<<grid_for(nl, 256), 256, 0, c->stream>>>(c->large_q.as<u32>() + l0, nl, c->blk_freq.as<u32>(),
                                                            c->blk_start.as<u64>(), c->blk_j0.as<u64>(), c->large_k0.as<u64>());
    HIPCHK(c, hipMemcpyAsync(desc3.data(), c->large_k0.p, desc3.size() * 8, hipMemcpyDeviceToHost, c->stream));
    if ((rc = sync_check(c))) return rc;
    // rounds: every pending block is split in the same five launches (batches of at most LS_BATCH_ROWS rows of
    // scratch), one synchronisation per batch; the ranges a batch reports as still too large are the next round's
    // blocks: a range of ties one pair of windows deeper, any other range on finer splitters.  A run of one symbol or a
    // tandem repeat ties for as long as it lasts and would shed 42 SP symbols per round: a range of ties that is most of
    // its block gets a PIVOT round next (LsBlock::pivot, stage_kernels.h), which measures how far every row follows one
    // row of the block and so halves what is left of a periodic stretch -- 805 rounds became ~40 on
    // scripts/gpu_lowcomplexity.py.  The bitonic network with its symbol-by-symbol comparator is only the last resort.
    constexpr u64 LS_BATCH_ROWS = 1ull << 29;
    struct Work { u64 b0, j0; u32 m, depth, pivot; };
    std::vector<Work> work, next;
    u32 maxm = 0;
    for (u64 t = 0; t < nl; t++) {
        work.push_back(Work{desc3[3 * t], desc3[3 * t + 1], (u32)desc3[3 * t + 2], 0u, 0u});
        maxm = std::max(maxm, (u32)desc3[3 * t + 2]);
    }
    c->st.blue_max_block = std::max<u64>(c->st.blue_max_block, maxm);
    if (c->cfg.reserved & 1024) {                                  // bit 10: bitonic network only (tests)
        for (const Work &wk : work) if ((rc = bitonic_large(c, wk.b0, wk.m, wk.j0))) return rc;
        return DEBWT_OK;
    }
    std::vector<u32> res;
    std::vector<LsOver> over;
    std::vector<LsBlock> desc;
    std::vector<u32> bigidx;
    u32 round = 0;
    // sweep knobs (scripts/ls_sweep.sh), clamped: a range aims at >= 64 rows, >= 1 sample per range, and never fewer samples
    // than ranges (k_ls_splitters takes sample (b + 1) * (ns / nb) - 1: ns / nb == 0 would index below the array)
    static const u32 bin_rows = getenv("DEBWT_LS_BIN_ROWS") ? (u32)std::min(std::max(atoi(getenv("DEBWT_LS_BIN_ROWS")), 64), 1 << 20) : LS_BIN_ROWS;
    static const u32 oversample = getenv("DEBWT_LS_OVERSAMPLE") ? (u32)std::min(std::max(atoi(getenv("DEBWT_LS_OVERSAMPLE")), 1), 64) : LS_OVERSAMPLE;
    const bool trace = getenv("DEBWT_TRACE_LARGE") != nullptr;
    while (!work.empty()) {
        next.clear();
        u64 tr_rows = 0, tr_tie_n = 0, tr_tie_rows = 0, tr_piv = 0, tr_net = 0;
        for (const Work &wk : work) { tr_rows += wk.m; tr_piv += wk.pivot; }
        for (size_t w0 = 0; w0 < work.size();) {
            desc.clear(); bigidx.clear();
            u64 rows = 0, wgs = 0, slots = 0;
            u32 nbig = 0;                                          // blocks of more than 256 samples in the batch
            bool small_ns = false, big_ns = false, small_nb = false, big_nb = false;
            size_t w1 = w0;
            for (; w1 < work.size() && (desc.empty() || rows + work[w1].m <= LS_BATCH_ROWS) && wgs < 0x7FFF0000ull; w1++) {
                const Work &wk = work[w1];
                u32 nb = 8;
                while (nb < LS_MAXBINS && (u64)nb * bin_rows < wk.m) nb <<= 1;
                u32 ns = 64;
                while (ns < LS_SAMPLES && ns < nb * oversample) ns <<= 1;        // a power of two (bitonic sort), a multiple of nb
                while (nb > ns) nb >>= 1;                                          // (ns is capped at LS_SAMPLES: at least one sample per range)
                desc.push_back(LsBlock{wk.b0, wk.j0, rows, wk.m, nb, ns, (u32)wgs, (u32)slots, wk.depth, wk.pivot, ns > 256u ? nbig : 0u});
                if (ns > 256u) { bigidx.push_back((u32)(w1 - w0)); nbig++; }
                rows += (wk.m + 1u) & ~1u;
                wgs += (wk.m + LS_WG_ROWS - 1u) / LS_WG_ROWS;
                slots += nb;
                (ns <= 256u ? small_ns : big_ns) = true;
                (nb <= 64u ? small_nb : big_nb) = true;
            }
            const size_t nblk = desc.size();
            const size_t over_cap = (size_t)(rows / LS_QUEUE_CAP) + 2;
            // scratch: en (u64 per row), bin (u32 per row); per splitter slot: two splitter words, three words for each of
            // its two ranges; per block: result, descriptor; per workgroup: its block; the batch's oversize ranges
            ENSURE(c, c->ls_buf, rows * 12 + slots * 40 + (size_t)nbig * (LS_SAMPLES * 16 + 4) + nblk * (8 + sizeof(LsBlock)) + wgs * 4 +
                                     over_cap * sizeof(LsOver) + 512);
            u64 *p64 = c->ls_buf.as<u64>();
            LargeSplit ls{};
            ls.nblk = (u32)nblk;
            ls.en = p64;
            ls.spl_w = p64 + rows; ls.spl_x = ls.spl_w + slots;
            ls.smp_w = ls.spl_x + slots; ls.smp_x = ls.smp_w + (size_t)nbig * LS_SAMPLES;
            LsBlock *dblk = reinterpret_cast<LsBlock *>(ls.smp_x + (size_t)nbig * LS_SAMPLES);
            ls.blk = dblk;
            ls.over = reinterpret_cast<LsOver *>(dblk + nblk);
            ls.bin = reinterpret_cast<u32 *>(ls.over + over_cap);
            ls.cnt = ls.bin + rows; ls.start = ls.cnt + 2 * slots; ls.cur = ls.start + 2 * slots;
            ls.wgblk = ls.cur + 2 * slots;
            ls.piv = ls.wgblk + wgs;
            u32 *d_big = ls.piv + nblk;
            ls.bigidx = d_big;
            ls.res = d_big + nbig;
            ls.nover = ls.res + nblk;                               // (read back together with res)
            if (nbig) HIPCHK(c, hipMemcpyAsync(d_big, bigidx.data(), (size_t)nbig * 4, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(dblk, desc.data(), nblk * sizeof(LsBlock), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemsetAsync(ls.nover, 0, 4, c->stream));
            if (small_ns) k_ls_splitters<256, 256><<<(u32)nblk, 256, 0, c->stream>>>(c->blue.as<u64>(), c->spn.as<u64>(), c->S, ls);
            if (big_ns) {
                k_ls_splitters<LS_SAMPLES, 64, 1><<<nbig, 64, 0, c->stream>>>(c->blue.as<u64>(), c->spn.as<u64>(), c->S, ls);
                k_ls_sample_keys<<<dim3(nbig, LS_SAMPLES / 256), 256, 0, c->stream>>>(c->blue.as<u64>(), c->spn.as<u64>(), c->S, ls);
                k_ls_splitters<LS_SAMPLES, 1024, 2><<<nbig, 1024, 0, c->stream>>>(c->blue.as<u64>(), c->spn.as<u64>(), c->S, ls);
            }
            k_ls_bin<<<(u32)wgs, 256, 0, c->stream>>>(c->blue.as<u64>(), c->spn.as<u64>(), c->S, ls);
            if (small_nb) k_ls_plan<64><<<(u32)nblk, 64, 0, c->stream>>>(ls, sub);
            if (big_nb) k_ls_plan<LS_MAXBINS><<<(u32)nblk, LS_MAXBINS, 0, c->stream>>>(ls, sub);
            k_ls_scatter<<<(u32)wgs, 256, 0, c->stream>>>(c->blue.as<u64>(), ls);
            res.resize(nblk + 1);
            HIPCHK(c, hipMemcpyAsync(res.data(), ls.res, (nblk + 1) * 4, hipMemcpyDeviceToHost, c->stream));
            if ((rc = sync_check(c))) return rc;                   // desc and res are host memory
            const u32 nover = res[nblk];
            if (nover > over_cap) { c->err = "large-block split: more oversize ranges than rows allow"; return DEBWT_EINTERNAL; }
            over.resize(nover);
            if (nover) HIPCHK(c, hipMemcpy(over.data(), ls.over, nover * sizeof(LsOver), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < nblk; i++)                       // sub-block table full: nothing of the block moved
                if (res[i]) { const Work &wk = work[w0 + i]; if ((rc = bitonic_large(c, wk.b0, wk.m, wk.j0))) return rc; }
            for (const LsOver &o : over) {
                const Work &wk = work[w0 + o.blk];
                if (o.ties) { tr_tie_n++; tr_tie_rows += o.cnt; }
                // a range of ties shares o.adv more pairs of windows (1 after a window round); one that is most of
                // its block is a stretch that keeps tying: pivot round next.  After a pivot round a range of ties is
                // smaller than its block (the pivot row itself is not in it), so the rounds end.
                const u64 deeper = ((u64)wk.depth + o.adv + 1) * (2 * SP_WIN);
                // (rows that left the pivot at once -- adv 0 after a pivot round -- show no such stretch: window round)
                const u32 pivot = (o.ties && (u64)o.cnt * 4 > wk.m && (!wk.pivot || o.adv >= 1) && !(c->cfg.reserved & 8192)) ? 1u : 0u;
                if (o.ties && (o.adv || wk.pivot) && deeper < c->S + 2 * SP_WIN && o.cnt < wk.m + (wk.pivot ? 0u : 1u))
                    next.push_back(Work{wk.b0 + o.st, wk.j0 + o.st, o.cnt, wk.depth + o.adv, pivot});
                else if (!o.ties && o.cnt < wk.m) next.push_back(Work{wk.b0 + o.st, wk.j0 + o.st, o.cnt, wk.depth + o.adv, wk.pivot});
                else { tr_net++; if ((rc = bitonic_large(c, wk.b0 + o.st, o.cnt, wk.j0 + o.st))) return rc; }
            }
            w0 = w1;
        }
        if (trace) {
            u32 dmax = 0; u64 rows = 0;
            for (const Work &wk : next) { dmax = std::max(dmax, wk.depth); rows += wk.m; }
            fprintf(stderr, "large blocks: round %u of %zu blocks (%llu rows, %llu pivot rounds) -> %zu ranges to split again (%llu rows, deepest %u; "
                            "ranges of ties: %llu with %llu rows; to the network: %llu)\n",
                    round, work.size(), (unsigned long long)tr_rows, (unsigned long long)tr_piv, next.size(), (unsigned long long)rows, dmax,
                    (unsigned long long)tr_tie_n, (unsigned long long)tr_tie_rows, (unsigned long long)tr_net);
        }
        work.swap(next);
        round++;
    }
    return DEBWT_OK;
}

#ifndef BLUE_QUEUE_LEVELS
#define BLUE_QUEUE_LEVELS 1            // (measured at 10 x 300 Mbp: 1, 2, 3, 4, 6 levels all give a blue stage of 19.6-20.0 ms)
#endif
// the sub-block table of a blue sort (BlueSub): the deep tie groups of the larger size classes wait there for the
// wave-per-block kernels
struct BlueQueue { BlueSub sub; u32 *count; u32 cap; };
static int blue_queue(debwt_ctx *c, BlueQueue *bq) {
    const u32 sub_cap = (u32)std::min<u64>(std::max<u64>(c->B / 8, 1u << 16), 1u << 26);
    ENSURE(c, c->sub_start, (size_t)sub_cap * 8);
    ENSURE(c, c->sub_j0, (size_t)sub_cap * 8);
    ENSURE(c, c->sub_freq, (size_t)sub_cap * 4);
    ENSURE(c, c->sub_depth, (size_t)sub_cap * 4 + 16);
    bq->cap = sub_cap;
    bq->count = c->sub_depth.as<u32>() + sub_cap;
    bq->sub = BlueSub{c->sub_start.as<u64>(), c->sub_freq.as<u32>(), c->sub_j0.as<u64>(), c->sub_depth.as<u32>(), bq->count,
                      (c->cfg.reserved & 128) ? 0u : sub_cap};      // bit 7: no hand-off (tests)
    return DEBWT_OK;
}

// Sorts the blocks [q0, q0 + nq) of the context's block table (block order is key order is row order) and, of the blocks
// above the LDS capacity, large_q[l0, l0 + nl).  A part leaves its rows final: debwt_build_to_host runs one part per key
// range and hands the rows over while the next part is sorted.
static int blue_sort_part(debwt_ctx *c, const BlueQueue &bq, u64 q0, u64 nq, u64 l0, u64 nl) {
    int rc;
    if (!nq) return DEBWT_OK;
    const u32 Q = (u32)nq;
    const u64 *bst = c->blk_start.as<u64>() + q0, *bj0 = c->blk_j0.as<u64>() + q0;
    const u32 *bfr = c->blk_freq.as<u32>() + q0;
    const BlueSub &sub = bq.sub;
    const u32 sub_cap = bq.cap;
    u32 *sub_count = bq.count;
    HIPCHK(c, hipMemsetAsync(c->sub_freq.p, 0, (size_t)sub_cap * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(sub_count, 0, 4, c->stream));
    BlueSub none{};
    u32 g1 = (u32)std::min<u64>(Q, 1u << 16);
    // blocks of up to 16 rows -- in a collection of genomes most blocks: a node's one occurrence per genome -- four to a wave
    // (cfg.reserved bit 24: a wave each, like the other blocks of up to 128 rows -- tests, A/B)
    const u32 tiny = (c->cfg.reserved & 16777216) ? 0u : 2 * BLUE_TINY;   // (17..32 rows: two to a wave)
    if (tiny) {
        k_blue_tiny<16><<<g1, 64, 0, c->stream>>>(c->blue.as<u64>(), bst, bfr, bj0, Q, c->spn.as<u64>(), c->S, c->mchar.as<u8>(),
                                                  nullptr, nullptr);
        k_blue_tiny<32><<<g1, 64, 0, c->stream>>>(c->blue.as<u64>(), bst, bfr, bj0, Q, c->spn.as<u64>(), c->S, c->mchar.as<u8>(),
                                                  nullptr, nullptr);
    }
    // small blocks are the bulk: a small LDS footprint keeps 32 single-wave workgroups per CU in flight
    k_blue_refine<64, 128, 0><<<g1, 64, 0, c->stream>>>(c->blue.as<u64>(), bst, bfr, bj0, Q, tiny,
                                                       c->spn.as<u64>(), c->S, c->mchar.as<u8>(), nullptr, nullptr, none);
    // 129..256 rows: one wave per block with the LDS of that capacity (9 KB per workgroup, 17 of them per CU);
    // 257..512 rows: 18 KB per block allow 8 workgroups per CU -- four waves each rather than one: these kernels wait
    // on SP gathers, what they need is waves in flight (measured at 30 Gbp: 129..512 rows by one wave and 18 KB each
    // 86 + 156 ms for blocks + queued ranges; now 6.5 + 67 and 22 + 78 ms)
    k_blue_refine<64, 256, 128><<<g1, 64, 0, c->stream>>>(c->blue.as<u64>(), bst, bfr, bj0, Q, 128u, c->spn.as<u64>(), c->S,
                                                         c->mchar.as<u8>(), nullptr, nullptr, sub);
    // blocks of 257 rows and more in a collection of many genomes: by classes around sampled splitters first
    // (k_blue_classify); what that finishes is marked in `done` and skipped by the kernels behind it, which take the blocks it
    // left (cfg.reserved bit 19: those kernels alone; bit 20: by classes whatever the number of such blocks -- tests)
    const bool split1024 = c->n1024 >= 1024;
    // (... or many blocks of 257..512 rows and few above: four genomes -- the four-wave network over those blocks alone took
    // 64 ms of a 654 ms build, by classes the blue stage is 78 ms instead of 116)
    const bool by_classes = split1024 || c->n512 >= 1024;
    const u8 *done = nullptr;
    if ((by_classes || (c->cfg.reserved & 1048576)) && sub.cap && !(c->cfg.reserved & 524288)) {
        ENSURE(c, c->blue_done, c->Q + 64);
        u8 *dn = c->blue_done.as<u8>() + q0;
        HIPCHK(c, hipMemsetAsync(dn, 0, Q, c->stream));
        const u32 gc = (u32)std::min<u64>(Q, 1u << 14);
        k_blue_classify<BLUE_WAVE_CAP, BLUE_WAVE_CAP><<<gc, 256, 0, c->stream>>>(c->blue.as<u64>(), bst, bfr, bj0, Q, 256u, c->spn.as<u64>(),
                                                                                 c->S, c->mchar.as<u8>(), sub, dn);
        k_blue_classify<1024, BLUE_WAVE_CAP><<<gc, 256, 0, c->stream>>>(c->blue.as<u64>(), bst, bfr, bj0, Q, (u32)BLUE_WAVE_CAP,
                                                                        c->spn.as<u64>(), c->S, c->mchar.as<u8>(), sub, dn);
        k_blue_classify<BLUE_LDS_CAP, BLUE_WAVE_CAP><<<(u32)std::min<u64>(Q, 1u << 12), 256, 0, c->stream>>>(
            c->blue.as<u64>(), bst, bfr, bj0, Q, 1024u, c->spn.as<u64>(), c->S, c->mchar.as<u8>(), sub, dn);
        done = dn;
    }
    k_blue_refine<256, BLUE_WAVE_CAP, 128><<<(u32)std::min<u64>(Q, 1u << 14), 256, 0, c->stream>>>(
        c->blue.as<u64>(), bst, bfr, bj0, Q, 256u, c->spn.as<u64>(), c->S, c->mchar.as<u8>(), nullptr, nullptr, sub, done);
    u32 g2 = (u32)std::min<u64>(Q, 1u << 12);
    // collections of many genomes put most rows into blocks of 513..1024 rows: those get a kernel of their own
    // with half the LDS footprint (four workgroups per CU instead of two); otherwise one kernel for 513..2048
    if (split1024)
        k_blue_refine<256, 1024, BLUE_WAVE_CAP><<<g2, 256, 0, c->stream>>>(c->blue.as<u64>(), bst, bfr, bj0, Q, (u32)BLUE_WAVE_CAP,
                                                                      c->spn.as<u64>(), c->S, c->mchar.as<u8>(), nullptr, nullptr, sub, done);
    k_blue_refine<256, BLUE_LDS_CAP, BLUE_WAVE_CAP><<<g2, 256, 0, c->stream>>>(c->blue.as<u64>(), bst, bfr, bj0, Q,
                                                                  split1024 ? 1024u : (u32)BLUE_WAVE_CAP, c->spn.as<u64>(), c->S,
                                                                  c->mchar.as<u8>(), nullptr, nullptr, sub, done);
    // heavy-tail blocks (satellite / poly-A nodes), before the queue is drained: split in HBM into ranges the LDS
    // kernels take (queued like the tie groups); what a split cannot separate goes through the bitonic network
    if (nl && (rc = sort_large_blocks(c, sub, l0, nl))) return rc;
    // the queued groups: blocks of their own that start `depth` windows in (<= 128 rows from the 512 class,
    // <= 512 rows from the 2048 class and from the split of the large blocks: LS_QUEUE_CAP)
    const u32 gs = std::min<u32>(sub_cap, 1u << 16);
    if (done) {
        // the queued groups of 129..512 rows by classes as well, one pair of windows deeper each time (two levels): what is
        // finished leaves the queue, what still ties is queued again for the wave kernels below
        u32 *snap = sub_count + 1;                                 // (a spare word behind the counter)
        for (int level = 0; level < BLUE_QUEUE_LEVELS; level++) {
            k_copy_u32<<<1, 1, 0, c->stream>>>(sub_count, snap);
            k_blue_classify<BLUE_WAVE_CAP, BLUE_WAVE_CAP><<<std::min<u32>(gs, 1u << 14), 256, 0, c->stream>>>(
                c->blue.as<u64>(), c->sub_start.as<u64>(), c->sub_freq.as<u32>(), c->sub_j0.as<u64>(), sub_cap, 128u, c->spn.as<u64>(),
                c->S, c->mchar.as<u8>(), sub, nullptr, c->sub_depth.as<u32>(), snap, c->sub_freq.as<u32>());
        }
    }
    if (tiny) {
        k_blue_tiny<16><<<gs, 64, 0, c->stream>>>(c->blue.as<u64>(), c->sub_start.as<u64>(), c->sub_freq.as<u32>(), c->sub_j0.as<u64>(),
                                                  sub_cap, c->spn.as<u64>(), c->S, c->mchar.as<u8>(), c->sub_depth.as<u32>(), sub_count);
        k_blue_tiny<32><<<gs, 64, 0, c->stream>>>(c->blue.as<u64>(), c->sub_start.as<u64>(), c->sub_freq.as<u32>(), c->sub_j0.as<u64>(),
                                                  sub_cap, c->spn.as<u64>(), c->S, c->mchar.as<u8>(), c->sub_depth.as<u32>(), sub_count);
    }
    k_blue_refine<64, 128, 0><<<gs, 64, 0, c->stream>>>(
        c->blue.as<u64>(), c->sub_start.as<u64>(), c->sub_freq.as<u32>(), c->sub_j0.as<u64>(), sub_cap, tiny,
        c->spn.as<u64>(), c->S, c->mchar.as<u8>(), c->sub_depth.as<u32>(), sub_count, none);
    k_blue_refine<64, 256, 0><<<gs, 64, 0, c->stream>>>(
        c->blue.as<u64>(), c->sub_start.as<u64>(), c->sub_freq.as<u32>(), c->sub_j0.as<u64>(), sub_cap, 128u,
        c->spn.as<u64>(), c->S, c->mchar.as<u8>(), c->sub_depth.as<u32>(), sub_count, none);
    // (257..512 queued rows are one deep tie group as a rule: four waves per block gather its windows four times as wide)
    k_blue_refine<256, BLUE_WAVE_CAP, 0><<<std::min<u32>(gs, 1u << 14), 256, 0, c->stream>>>(
        c->blue.as<u64>(), c->sub_start.as<u64>(), c->sub_freq.as<u32>(), c->sub_j0.as<u64>(), sub_cap, 256u,
        c->spn.as<u64>(), c->S, c->mchar.as<u8>(), c->sub_depth.as<u32>(), sub_count, none);
    return DEBWT_OK;
}

extern "C" int debwt_blue_sort(debwt_ctx *c) {
    if (!c) return DEBWT_EINVAL;
    if (c->stage < ST_SP) return DEBWT_ESTATE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc;
    HIPCHK(c, hipEventRecord(c->ev[5], c->stream));
    c->st.blue_max_block = 0;
    if (c->Q) {
        BlueQueue bq{};
        if ((rc = blue_queue(c, &bq))) return rc;
        if ((rc = blue_sort_part(c, bq, 0, c->Q, 0, c->nlarge))) return rc;
    }
    c->stage = ST_BLUE;
    return DEBWT_OK;
}

// ---------------------------------------------------------------------------------------------------
// stage 5: assembly                                                                            (a-6)

// rows of this shard: its node instances with its special suffixes merged in (the whole BWT when not sharded)
static u64 shard_rows(const debwt_ctx *c) { return c->Mctx + (c->s1 - c->s0); }

// assembles the 8192-row blocks [b0, b1) of this shard's rows (all of them: b1 = ~0)
// The '#' rows: collections of up to 2^20 records have k_assemble append them to a list (c->hash_rows, counter in the
// spare word 6 behind the '$' row) that assemble_finish sorts; collections of more records (read sets: n is small against
// N there) mark them in hmask and find them by one counting pass over the mask words, as every collection did before.
static bool hash_rows_by_list(const debwt_ctx *c) { return c->nrec <= (1ull << 20) && !(c->cfg.reserved & 2097152); }   // bit 21: masks (tests)
static int run_assemble(debwt_ctx *c, u8 *rowsym, u64 b0 = 0, u64 b1 = ~0ull, bool collect = true) {
    const u64 rows = shard_rows(c);
    const u64 nb = (((rows + 31) >> 5) + DEBWT_BLOCK - 1) / DEBWT_BLOCK;
    b1 = std::min(b1, nb);
    if (b0 >= b1) return DEBWT_OK;
    const bool list = collect && hash_rows_by_list(c);
    k_assemble<<<(u32)(b1 - b0), DEBWT_BLOCK, 0, c->stream>>>(
        c->mchar.as<u8>(), c->Mctx, c->sprow.as<u64>() + c->s0, c->spchr.as<u8>() + c->s0, c->s1 - c->s0, rows,
        c->bwt.as<u64>(), c->hmask.as<u32>(), c->dollar.as<u64>(), rowsym, b0, list ? c->hash_rows.as<u64>() : nullptr,
        reinterpret_cast<unsigned long long *>(c->dollar.as<u64>() + 6));
    return DEBWT_OK;
}

// '#' rows of the assembled words, stage timings, stage = ST_ASSEMBLED
static int assemble_finish(debwt_ctx *c) {
    int rc;
    u64 nw = (shard_rows(c) + 31) >> 5;
    if (hash_rows_by_list(c)) {
        u64 *h_cnt = reinterpret_cast<u64 *>(&c->h_scalars[40]);                   // (8-byte aligned pair of the pinned words)
        HIPCHK(c, hipMemcpyAsync(h_cnt, c->dollar.as<u64>() + 6, 8, hipMemcpyDeviceToHost, c->stream));
        if ((rc = sync_check(c))) return rc;
        const u64 cnt = *h_cnt;
        if (cnt > c->nrec) { c->err = "more '#' rows than records"; return DEBWT_EINTERNAL; }
        c->h_scalars[9] = (u32)cnt;
        if (cnt > 1) {                                                              // ascending (src/insertCase3.c:86-97 walks the rows in order)
            ENSURE(c, c->large_k0, cnt * 8 + 64);
            ENSURE(c, c->rs_over, radix_over_bytes(cnt));
            hipError_t e = hipSuccess;
            u64 *r = radix_sort_bits(c->stream, c->hash_rows.as<u64>(), c->large_k0.as<u64>(), cnt, 0, bits_for(shard_rows(c)), radix_ws(c), &e);
            if (e != hipSuccess) { c->err = std::string("'#' row sort: ") + hipGetErrorString(e); return DEBWT_EDEVICE; }
            if (r != c->hash_rows.as<u64>()) HIPCHK(c, hipMemcpyAsync(c->hash_rows.p, r, cnt * 8, hipMemcpyDeviceToDevice, c->stream));
        }
    } else {
        HashRowsF fh{c->hmask.as<u32>(), c->hash_rows.as<u64>()};
        if ((rc = cp_count(c, fh, nw, cp_area(c, 0), 9))) return rc;
        if ((rc = cp_emit(c, fh, nw, cp_area(c, 0)))) return rc;
    }
    HIPCHK(c, hipEventRecord(c->ev[7], c->stream));
    if ((rc = sync_check(c))) return rc;
    c->n_hash_local = c->h_scalars[9];
    if (c->shard_world == 1 && c->h_scalars[9] != c->nrec - 1) {
        c->err = "number of '#' rows differs from records-1";
        return DEBWT_EINTERNAL;
    }
    // stage timings
    float ms[7] = {0};
    for (int i = 0; i < 7; i++) (void)hipEventElapsedTime(&ms[i], c->ev[i], c->ev[i + 1]);
    c->st.ms_extract = ms[0]; c->st.ms_sort = ms[1]; c->st.ms_classify = ms[2] + ms[3];
    c->st.ms_sp = ms[4]; c->st.ms_blue = ms[5]; c->st.ms_assemble = ms[6];
    (void)hipEventElapsedTime(&c->st.ms_total, c->ev[0], c->ev[7]);
    c->st.radix_pass_launches = (uint32_t)c->n_pass_events;
    c->st.radix_pass_ms = 0.f;
    for (int i = 0; i < c->n_pass_events; i++) {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, c->ev_pass[i][0], c->ev_pass[i][1]);
        c->st.radix_pass_ms += t;
    }
    c->stage = ST_ASSEMBLED;
    return DEBWT_OK;
}

extern "C" int debwt_bwt_assemble(debwt_ctx *c) {
    if (!c) return DEBWT_EINVAL;
    if (c->stage < ST_BLUE) return DEBWT_ESTATE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
    HIPCHK(c, hipMemsetAsync(c->dollar.p, 0xFF, 8, c->stream));
    HIPCHK(c, hipMemsetAsync(c->dollar.as<u64>() + 6, 0, 8, c->stream));
    run_assemble(c, nullptr);
    return assemble_finish(c);
}

// Device memory for a text of up to `n` symbols in `nrec` records, before the text is there.  What a cold build pays for is
// not hipMalloc's work but the driver handing out memory another process released: beyond a pool of cleared pages it
// clears on allocation, 30 ms per GiB on this system (profiles/r04_alloc_probe.txt: 6.2 s before the second 25-GiB block
// of a process started right after one that held 250 GiB) -- seconds for the ~250 GB of a 30 Gbp build, whose kernels take 1.8 s.
// A one-shot host calls this on a helper thread while it still reads and packs its input (the file size bounds n); the
// buffers are those debwt_load_text and the stages would allocate, sized as they would size them, the data-dependent ones
// (blue entries, SP code, sub-block tables) for `branching` of the positions being branching nodes (<= 0: 0.12; a text that
// needs more grows them when it gets there).  Not to be called while another call on the context is running.
extern "C" int debwt_reserve(debwt_ctx *c, uint64_t n, uint64_t nrec, double branching, unsigned flags) {
    if (!c || n < 34 || nrec == 0 || n <= nrec * (uint64_t)c->K || (flags & ~(DEBWT_RESERVE_ONE_SHOT | DEBWT_RESERVE_COMPACT))) return DEBWT_EINVAL;
    // before the text is there, and only then: the buffers below are re-allocated WITHOUT their contents, so a context that
    // holds a loaded text (or a result) would go on with a stale census over uninitialised memory
    if (c->stage >= ST_LOADED) { c->err = "debwt_reserve: the context already holds a text (reserve comes before the load)"; return DEBWT_ESTATE; }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (branching <= 0) branching = 0.12;
    const auto t_begin = std::chrono::steady_clock::now();
    const u64 NS = nrec * (u64)c->K, M = n - NS;
    const size_t tw = (size_t)((n + 63) >> 5) + 2, bw = (size_t)(n >> 6) + 3, ngroups = (size_t)((n + 31) >> 5);
    ENSURE(c, c->text, tw * 8);
    ENSURE(c, c->sepbits, bw * 8);
    ENSURE(c, c->sep, nrec * 8);
    ENSURE(c, c->head_keys, nrec * 8);
    ENSURE(c, c->spkey, NS * 8);
    ENSURE(c, c->sprow, NS * 8);
    ENSURE(c, c->spchr, NS + 64);
    ENSURE(c, c->bwt, ngroups * 8 + 64);
    ENSURE(c, c->hmask, ngroups * 4 + 64);
    ENSURE(c, c->hash_rows, nrec * 8 + 64);
    ENSURE(c, c->mchar, (n - NS) + 64);
    ENSURE(c, c->momask, ngroups * 4 + 64);
    ENSURE(c, c->mimask, ngroups * 4 + 64);
    // the key ranges as plan_ranges will cut them: the cap from the memory that is free now plus what the context holds
    u64 cap = c->range_cap;
    if (!cap) {
        const u64 keep_n = c->n; c->n = n;
        int rc = default_range_cap(c, 30, 4 * n + (8ull << 30), 0, &cap);
        c->n = keep_n;
        if (rc) return rc;
        // A one-shot build that is being handed memory at the driver's clearing rate (the ~2 bytes per position above came
        // slower than 10 ms per GiB) is better off with more, smaller key ranges: 30 bytes per key of range workspace at
        // ~30 ms per GiB against ~34 ms per range and 30 Gbp for another first pass over the text -- the sum is least at
        // P = sqrt(0.9 s x M / 1e9 / ms per range) ranges, but never more than RS_MAX_RANGES: beyond that the first-pass
        // histograms of the ranges no longer come from one scan of the text (28 ranges at 30 Gbp: sort stage 2.1 s instead
        // of 1.0 s, profiles/r04_cli_30G.txt).  16 ranges at 30 Gbp: 54 GB of range workspace instead of 125.  The cap stays
        // with the context (debwt_set_range_cap).
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        const double gib = (double)(tw * 8 + bw * 8 + ngroups * 20 + (n - NS)) / (double)(1ull << 30);
        if ((flags & DEBWT_RESERVE_COMPACT) && (n - NS) <= 20000000000ull) {
            // (Beyond ~20 Gbp the buffers that grow with the text alone -- 3-4 bytes per base -- are half of the device's memory:
            // a run behind another of the same size waits whatever the ranges are, 30 Gbp: 10.3-10.9 s of wall time with 16 ranges,
            // 11.5 with 7, profiles/r06_cli_compact_plan.txt; the rule below stays in charge there.)
            // Key ranges of 2^29 instances whatever the allocations cost so far: the driver clears memory when it is RELEASED
            // and an allocation that is handed a block not yet cleared waits for the whole job (profiles/r06_malloc_cost.txt: 96 GB
            // right behind another process: 72 GB at once, then 2.7 s) -- the first buffers say nothing about the large ones.
            // 3.1 Gbp in 6 ranges: 28 GiB instead of 102, 158 ms per build instead of 146 (scripts/gpu_range_footprint.py).
            const u64 P = std::min<u64>(RS_MAX_RANGES, ((n - NS) + (1ull << 29) - 1) >> 29);
            const u64 want = (u64)((double)(n - NS) / (double)std::max<u64>(1, P) * 1.02) + 1;
            if (P > 1 && want < cap) { cap = std::max<u64>(want, 1ull << 26); c->range_cap = cap; c->plan_valid = false; }
        } else if ((flags & DEBWT_RESERVE_ONE_SHOT) && gib > 4.0 && secs > 0.010 * gib) {
            const double per_range_s = 0.034 * (double)n / 30e9 + 0.004;
            const double P = std::min<double>(RS_MAX_RANGES, std::sqrt(30.0 * (double)(n - NS) / 35e9 / per_range_s));
            const u64 want = (u64)((double)(n - NS) / std::max(1.0, P) * 1.02) + 1;   // (cuts fall on prefix bins: a little slack)
            if (want < cap) { cap = std::max<u64>(want, 1ull << 26); c->range_cap = cap; c->plan_valid = false; }
        }
    }
    cap = std::min<u64>(cap, 0xFFFFFFF0ull - 1);
    u64 maxM = M;
    if (M > cap) { const u64 P = (M + cap - 1) / cap, per = (M + P - 1) / P; maxM = std::min(cap, per + per / 16); }
    ENSURE(c, c->keysA, maxM * 8 + 64);
    ENSURE(c, c->keysB, maxM * 8 + 64);
    ENSURE(c, c->rs_skew, (maxM / 2048 + 2) * 4);
    ENSURE(c, c->rs_rle, radix_rle_ws_bytes(maxM));
    ENSURE(c, c->dk, maxM * 8 + 64);
    ENSURE(c, c->dstart, maxM * 4 + 64);
    ENSURE(c, c->pflag, maxM + 64);
    const u64 Best = (u64)((double)n * branching);
    ENSURE(c, c->blue, Best * 8 + 64);
    ENSURE(c, c->spsym, Best + 64);
    ENSURE(c, c->spn, ((Best >> 6) + 3) * 24);
    const u32 sub_cap = (u32)std::min<u64>(std::max<u64>(Best / 8, 1u << 16), 1u << 26);
    ENSURE(c, c->sub_start, (size_t)sub_cap * 8);
    ENSURE(c, c->sub_j0, (size_t)sub_cap * 8);
    ENSURE(c, c->sub_freq, (size_t)sub_cap * 4);
    ENSURE(c, c->sub_depth, (size_t)sub_cap * 4 + 16);
    ENSURE(c, c->qlist, (std::min<u64>(SP_SLICE_GROUPS * 32, Best) + 64) * 4);
    ENSURE(c, c->qwave, ((std::min<u64>(SP_SLICE_GROUPS, ngroups) + 63) / 64 + 2) * 8);
    return DEBWT_OK;
}

// (whole_build: the stages run back to back, no caller can ask for what lies between them)
struct WholeBuild {
    debwt_ctx *c;
    explicit WholeBuild(debwt_ctx *c_) : c(c_) { c->whole_build = true; }
    ~WholeBuild() { c->whole_build = false; }
};

extern "C" int debwt_build(debwt_ctx *c) {
    int rc;
    if (!c) return DEBWT_EINVAL;
    if (c->stage < ST_LOADED) return DEBWT_ESTATE;
    c->stage = ST_LOADED;
    WholeBuild wb{c};
    if ((rc = debwt_kmer_sort_rle(c))) return rc;
    if ((rc = debwt_classify(c))) return rc;
    if ((rc = debwt_sp_generate(c))) return rc;
    if ((rc = debwt_blue_sort(c))) return rc;
    return debwt_bwt_assemble(c);
}

// debwt_build with the result handed to the host as it becomes final.  A text built in several key ranges has its
// multi-in blocks, like its rows, in key-range order: the blue sort runs range by range, the rows of a range are
// assembled as soon as its blocks are sorted and leave on a second stream while the blocks of the next range are sorted
// (at 30 Gbp: 7.5 GB of rows, 0.13 s of copy under 0.33 s of kernels; only the last range's copy is exposed).  A text
// built in one range: debwt_build, then debwt_fetch_bwt.
extern "C" int debwt_build_to_host(debwt_ctx *c, uint64_t *bwt, uint64_t *hash_rows, uint64_t *dollar_row) {
    int rc;
    if (!c || !bwt || !dollar_row || (c->nrec > 1 && !hash_rows)) return DEBWT_EINVAL;
    if (c->stage < ST_LOADED) return DEBWT_ESTATE;
    if (c->shard_world > 1) { c->err = "sharded context: use the debwt_shard_* calls"; return DEBWT_ESTATE; }
    c->stage = ST_LOADED;
    WholeBuild wb{c};
    if ((rc = debwt_kmer_sort_rle(c))) return rc;
    if ((rc = debwt_classify(c))) return rc;
    if ((rc = debwt_sp_generate(c))) return rc;
    if (c->ranges.size() < 2 || (c->cfg.reserved & 65536)) {        // bit 16: no overlap (tests, A/B)
        if ((rc = debwt_blue_sort(c))) return rc;
        if ((rc = debwt_bwt_assemble(c))) return rc;
        return debwt_fetch_bwt(c, bwt, hash_rows, dollar_row);
    }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (!c->copy_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!c->ev_copy) HIPCHK(c, hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->ev[5], c->stream));
    HIPCHK(c, hipMemsetAsync(c->dollar.p, 0xFF, 8, c->stream));
    HIPCHK(c, hipMemsetAsync(c->dollar.as<u64>() + 6, 0, 8, c->stream));
    c->st.blue_max_block = 0;
    BlueQueue bq{};
    if (c->Q && (rc = blue_queue(c, &bq))) return rc;
    const u64 rows = shard_rows(c), nw = (rows + 31) >> 5;
    u64 bdone = 0;                                                    // 8192-row blocks assembled and on their way
    DrainOnError drain{c};                                            // copies into the caller's `bwt` are queued range by range
    for (size_t i = 0; i < c->ranges.size(); i++) {
        const debwt_ctx::KeyRange &r = c->ranges[i];
        if ((rc = blue_sort_part(c, bq, r.qbase, r.Q, r.l0, r.nl))) return rc;
        // rows below `final` belong to this range or an earlier one: their symbols will not change any more
        const bool last = i + 1 == c->ranges.size();
        const u64 final = last ? rows : r.Mbase + r.M + (r.s1 - c->s0);
        const u64 b1 = last ? (nw + DEBWT_BLOCK - 1) / DEBWT_BLOCK : final / ASM_ROWS;
        if (last) HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
        if (b1 > bdone) {
            run_assemble(c, nullptr, bdone, b1);
            const u64 w0 = bdone * DEBWT_BLOCK, w1 = std::min<u64>(b1 * DEBWT_BLOCK, nw);
            HIPCHK(c, hipEventRecord(c->ev_copy, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_copy, 0));
            HIPCHK(c, hipMemcpyAsync(bwt + w0, c->bwt.as<u64>() + w0, (w1 - w0) * 8, hipMemcpyDeviceToHost, c->copy_stream));
            bdone = b1;
        }
    }
    c->stage = ST_BLUE;
    if ((rc = assemble_finish(c))) return rc;                          // synchronises the context's stream
    if (c->nrec > 1)
        HIPCHK(c, hipMemcpyAsync(hash_rows, c->hash_rows.p, (c->nrec - 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(dollar_row, c->dollar.p, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    if ((rc = sync_check(c))) return rc;
    drain.armed = false;
    return DEBWT_OK;
}

extern "C" int debwt_fetch_bwt(debwt_ctx *c, uint64_t *bwt, uint64_t *hash_rows, uint64_t *dollar_row) {
    if (!c || !bwt || !dollar_row || (c->nrec > 1 && !hash_rows)) return DEBWT_EINVAL;
    if (c->stage < ST_ASSEMBLED) return DEBWT_ESTATE;
    if (c->shard_world > 1) { c->err = "sharded context: use debwt_shard_fetch"; return DEBWT_ESTATE; }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipMemcpyAsync(bwt, c->bwt.p, (size_t)((c->n + 31) >> 5) * 8, hipMemcpyDeviceToHost, c->stream));
    if (c->nrec > 1)
        HIPCHK(c, hipMemcpyAsync(hash_rows, c->hash_rows.p, (c->nrec - 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(dollar_row, c->dollar.p, 8, hipMemcpyDeviceToHost, c->stream));
    return sync_check(c);
}

extern "C" int debwt_fetch_rows(debwt_ctx *c, uint64_t *hash_rows, uint64_t *dollar_row) {
    if (!c || !dollar_row || (c->nrec > 1 && !hash_rows)) return DEBWT_EINVAL;
    if (c->stage < ST_ASSEMBLED) return DEBWT_ESTATE;
    if (c->shard_world > 1) { c->err = "sharded context: use debwt_shard_fetch"; return DEBWT_ESTATE; }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (c->nrec > 1)
        HIPCHK(c, hipMemcpyAsync(hash_rows, c->hash_rows.p, (c->nrec - 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(dollar_row, c->dollar.p, 8, hipMemcpyDeviceToHost, c->stream));
    return sync_check(c);
}

extern "C" int debwt_bwt_census(debwt_ctx *c, uint64_t counts[4]) {
    if (!c || !counts) return DEBWT_EINVAL;
    if (c->stage < ST_ASSEMBLED) return DEBWT_ESTATE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    ENSURE(c, c->dollar, 64);
    u64 *d4 = c->dollar.as<u64>() + 1;                       // four spare words behind the '$' row
    HIPCHK(c, hipMemsetAsync(d4, 0, 32, c->stream));
    k_bwt_census<<<2048, DEBWT_BLOCK, 0, c->stream>>>(c->bwt.as<u64>(), shard_rows(c), d4);
    HIPCHK(c, hipMemcpyAsync(counts, d4, 32, hipMemcpyDeviceToHost, c->stream));
    return sync_check(c);
}

extern "C" int debwt_bwt_device_ptr(debwt_ctx *c, const uint64_t **d_words) {
    if (!c || !d_words) return DEBWT_EINVAL;
    if (c->stage < ST_ASSEMBLED) return DEBWT_ESTATE;
    *d_words = c->bwt.as<uint64_t>();
    return DEBWT_OK;
}

// ---------------------------------------------------------------------------------------------------
// k-mer-prefix sharding of one build over several GPUs.  Every shard holds the whole 2-bit text;
// shard r sorts and classifies the keys of its prefix range, owns their contiguous BWT rows and their blocks.
// Host orchestration and the collectives live in debwt_amd/sharded.py.

static void shard_slice(const debwt_ctx *c, u64 *p0, u64 *p1) {
    // slices are cut at multiples of 32 positions so that the SP stage can work on whole text words
    u64 per = ((c->n / c->shard_world) >> 5) << 5;
    *p0 = per * c->shard_rank;
    *p1 = c->shard_rank + 1 == c->shard_world ? c->n : per * (c->shard_rank + 1);
}

extern "C" int debwt_shard_begin(debwt_ctx *c, int rank, int world) {
    if (!c || world < 1 || world > 255 || rank < 0 || rank >= world) return DEBWT_EINVAL;   // owner tables hold bytes, 0xFF = none
    if (c->stage < ST_LOADED) return DEBWT_ESTATE;
    c->shard_rank = rank; c->shard_world = world; c->exchange = false; c->shard_planned = false;
    c->key_lo = c->key_hi = 0; c->M = c->Mctx = c->Mfull; c->Mbase = 0; c->qbase = 0; c->s0 = 0; c->s1 = c->NS;
    c->ranges.clear(); c->plan_valid = false;
    c->stage = ST_LOADED;
    return DEBWT_OK;
}

extern "C" int debwt_shard_histogram(debwt_ctx *c, uint64_t *hist4096) {
    if (!c || !hist4096) return DEBWT_EINVAL;
    if (c->stage < ST_LOADED) return DEBWT_ESTATE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    ENSURE(c, c->shard_hist, SHARD_BINS * 8);
    HIPCHK(c, hipMemsetAsync(c->shard_hist.p, 0, SHARD_BINS * 8, c->stream));
    // this shard counts the positions of its text slice: the census itself is data-parallel over the text
    u64 p0, p1;
    shard_slice(c, &p0, &p1);
    if (p0 & 31ull)
        k_prefix_hist<<<2048, DEBWT_BLOCK, 0, c->stream>>>(c->text.as<u64>(), c->sepbits.as<u64>(), p0, p1, c->K,
                                                            c->shard_hist.as<u64>());
    else
        k_prefix_hist_words<<<2048, DEBWT_BLOCK, 0, c->stream>>>(c->text.as<u64>(), c->sepbits.as<u64>(), p0, p1, c->K,
                                                                  c->shard_hist.as<u64>());
    HIPCHK(c, hipMemcpyAsync(hist4096, c->shard_hist.p, SHARD_BINS * 8, hipMemcpyDeviceToHost, c->stream));
    return sync_check(c);
}

extern "C" int debwt_shard_set_range(debwt_ctx *c, uint32_t bin_lo, uint32_t bin_hi, uint64_t m_keys, uint64_t m_base) {
    if (!c || bin_lo > bin_hi || bin_hi > SHARD_BINS) return DEBWT_EINVAL;
    if (c->stage < ST_LOADED) return DEBWT_ESTATE;
    if (m_keys > c->Mfull) return DEBWT_EINVAL;
    const int kb = 2 * c->cfg.k;                       // key bits; a bin is the top 12 of them
    c->key_lo = (u64)bin_lo << (kb - 12);
    c->key_hi = bin_hi == SHARD_BINS ? 0ull : ((u64)bin_hi << (kb - 12));   // 0: no upper bound (last shard)
    c->M = c->Mctx = m_keys; c->Mbase = m_base;
    c->shard_planned = false; c->exchange = false; c->ranges.clear();
    c->stage = ST_LOADED;
    return DEBWT_OK;
}

extern "C" int debwt_shard_plan(debwt_ctx *c, const uint64_t *hist4096, uint32_t bin_lo, uint32_t bin_hi, uint64_t m_base,
                                int exchange, uint64_t caller_held_bytes, uint32_t *nranges) {
    if (!c || !hist4096 || !nranges || bin_lo > bin_hi || bin_hi > SHARD_BINS) return DEBWT_EINVAL;
    if (c->stage < ST_LOADED) return DEBWT_ESTATE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const int kb = 2 * c->cfg.k;
    u64 m = 0;
    for (u32 b = bin_lo; b < bin_hi; b++) m += hist4096[b];
    if (m > c->Mfull) return DEBWT_EINVAL;
    c->key_lo = (u64)bin_lo << (kb - 12);
    c->key_hi = bin_hi == SHARD_BINS ? 0ull : ((u64)bin_hi << (kb - 12));
    c->M = c->Mctx = m; c->Mbase = m_base;
    u64 cap = c->range_cap;
    if (!cap) {
        // range workspace per key: 30 bytes, plus the caller's send buffer in exchange mode (the receive buffer is
        // key buffer A); the later stages hold ~3.5 bytes per position of the shard's share (row symbols, blue
        // entries and their routed form, BWT) and ~1.25 bytes per position of the whole text (2-bit text, flag
        // masks, SP code and its gather buffers, node table, the concatenated BWT on the gathering rank)
        // exchange == 2 (keys rescanned, SP code and blue entries sliced): the caller's two blue-entry buffers come
        // after the sort, ~2 bytes per position of the share (8 bytes x 2 x ~0.1 multi-in positions per base); a sliced
        // SP stage (exchange != 0) keeps the block ids of the slice's multi-in positions (4 bytes each) between its passes
        // exchange == 2 since round 6: the two blue-entry buffers ARE the key buffers (debwt_shard_scratch) and the routed
        // entries sit in one of them, so what stays beside a range's workspace is what a one-GPU build holds: per position
        // of the share 1 byte of row symbols, ~1.25 of blue entries and block ids, 0.25 of rows (3 with room for denser
        // collections), per position of the text 0.6 of text, separator bits and flag masks, ~0.35 of SP code and its gather
        // buffers, ~0.3 of node table, 0.25 of concatenated rows (1.5)
        const u64 share = c->n / (u64)c->shard_world;
        int rc = exchange == 2 ? default_range_cap(c, 30, share * 3 + c->n / 2 * 3 + (8ull << 30), caller_held_bytes, &cap)
                               : default_range_cap(c, exchange == 1 ? 40 : 30, share / 2 * 7 + (exchange ? share / 2 : 0) + c->n / 4 * 5 + (8ull << 30),
                                                   caller_held_bytes, &cap);
        if (rc) return rc;
    }
    int rc = cut_ranges(c, reinterpret_cast<const u64 *>(hist4096), bin_lo, bin_hi, cap, m);
    if (rc) return rc;
    c->shard_planned = true; c->exchange = exchange == 1; c->plan_valid = false;
    c->stage = ST_LOADED;
    *nranges = (u32)c->ranges.size();
    return DEBWT_OK;
}

// Key exchange or key rescan (include/debwt_hip.h).  Per-GPU milliseconds of what differs between the two, calibrated
// on 30 Gbp builds in a process group of one (profiles/r02_v24_bench_30G_keys_*.json):
//   rescan    a first radix pass that reads the whole text and keeps one key range takes 32.3 ms per 30 Gbp read, the
//             histogram pass before it about as much                          -> 2.2 ms per Gbp read and key range
//   exchange  sort stage 2195 ms against 1230 ms: the slice is read once per round, its keys are written grouped by
//             owner, copied (208 ms of that: 6.9 ms per Gbp when the copy stays in HBM) and then take an ordinary first
//             pass                                                            -> 40 ms per Gbp of the shard's share
//             plus 8 n / world^2 bytes over every link
extern "C" int debwt_shard_key_mode(uint64_t n, int world, double link_gbytes_per_s, double *exchange_ms, double *rescan_ms) {
    if (world < 1) world = 1;
    const double link = link_gbytes_per_s > 0 ? link_gbytes_per_s : 48.0;
    const double gbp = (double)n * 1e-9, share = gbp / world;
    const double ranges = std::max(1.0, std::ceil(share / 4.29));          // < 2^32 - 2^20 keys per range
    const double t_rescan = ranges * gbp * 2.2;
    const double t_exchange = share * 40.0 + share / world * 6.9 + (world > 1 ? share / world * 8.0 / link * 1e3 : 0.0);
    if (exchange_ms) *exchange_ms = t_exchange;
    if (rescan_ms) *rescan_ms = t_rescan;
    return t_exchange < t_rescan ? DEBWT_KEYS_EXCHANGE : DEBWT_KEYS_RESCAN;
}

extern "C" int debwt_shard_ranges(debwt_ctx *c, uint32_t *bin_bounds, uint64_t *m_keys, uint32_t capacity) {
    if (!c || !bin_bounds || !m_keys) return DEBWT_EINVAL;
    if (!c->shard_planned || c->ranges.size() > capacity) return DEBWT_ESTATE;
    const int kb = 2 * c->cfg.k;
    for (size_t i = 0; i < c->ranges.size(); i++) {
        bin_bounds[i] = (u32)(c->ranges[i].key_lo >> (kb - 12));
        bin_bounds[i + 1] = c->ranges[i].key_hi ? (u32)(c->ranges[i].key_hi >> (kb - 12)) : SHARD_BINS;
        m_keys[i] = c->ranges[i].M;
    }
    return DEBWT_OK;
}

extern "C" int debwt_shard_sort_begin(debwt_ctx *c) {
    if (!c) return DEBWT_EINVAL;
    if (c->stage < ST_LOADED || !c->shard_planned || !c->exchange) return DEBWT_ESTATE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    c->stage = ST_LOADED;
    int rc = sort_begin(c);
    if (rc) join_special(c);
    return rc;
}

extern "C" int debwt_shard_sort_range(debwt_ctx *c, uint32_t range, uint64_t *d_keys, uint64_t count) {
    if (!c || (!d_keys && count)) return DEBWT_EINVAL;
    if (c->stage != ST_LOADED || !c->exchange || range >= c->ranges.size()) return DEBWT_ESTATE;
    if (count != c->ranges[range].M) { c->err = "received keys differ from the census of the range"; return DEBWT_EINTERNAL; }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc = sort_range(c, range, (u64 *)d_keys);
    if (rc) join_special(c);
    return rc;
}

extern "C" int debwt_shard_sort_end(debwt_ctx *c) {
    if (!c) return DEBWT_EINVAL;
    if (c->stage != ST_LOADED || !c->exchange) return DEBWT_ESTATE;
    join_special(c);
    return sort_end(c);
}

extern "C" int debwt_shard_classify_local(debwt_ctx *c, uint64_t *nfacts, uint64_t *nblocks, uint64_t *blue_rows) {
    if (!c) return DEBWT_EINVAL;
    if (c->stage < ST_SORTED) return DEBWT_ESTATE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (!c->local_done) {                                  // one text-fed range: its keys are still there
        if (c->ranges.size() != 1) return DEBWT_ESTATE;
        c->Q = c->B = c->nlarge = 0; c->nfacts_acc = 0; c->n1024 = 0; c->n512 = 0;
        int rc = classify_local(c);
        if (rc) return rc;
        if ((rc = append_range(c, c->ranges[0]))) return rc;
    }
    if (nfacts) *nfacts = c->nfacts_acc;
    if (nblocks) *nblocks = c->Q;
    if (blue_rows) *blue_rows = c->B;
    return DEBWT_OK;
}

extern "C" int debwt_shard_facts_export(debwt_ctx *c, uint64_t *d_dst, uint64_t capacity) {
    if (!c || !d_dst) return DEBWT_EINVAL;
    if (!c->facts_ready || capacity < c->nfacts_acc) return DEBWT_ESTATE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (c->nfacts_acc)
        HIPCHK(c, hipMemcpyAsync(d_dst, c->facts_acc.p, c->nfacts_acc * 8, hipMemcpyDeviceToDevice, c->stream));
    return sync_check(c);
}

extern "C" int debwt_shard_classify_global(debwt_ctx *c, const uint64_t *d_facts, uint64_t nfacts, uint64_t qbase,
                                           uint64_t blue_total) {
    if (!c || (!d_facts && nfacts)) return DEBWT_EINVAL;
    if (c->stage < ST_SORTED || !c->facts_ready) return DEBWT_ESTATE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    return classify_global(c, (const u64 *)d_facts, nfacts, qbase, blue_total);
}

// ---- exchange mode: keys and blue entries travel by alltoallv, every shard scans only its text slice ----------

extern "C" int debwt_shard_partition_keys(debwt_ctx *c, const uint8_t *shard_of_bin, uint64_t *d_out, uint64_t capacity,
                                          uint64_t *offs) {
    // keys of this shard's text slice, grouped by destination shard (bucket exchange); bins whose
    // entry is 0xFF are not part of this exchange round and yield nothing
    if (!c || !shard_of_bin || !d_out || !offs) return DEBWT_EINVAL;
    if (c->stage < ST_LOADED) return DEBWT_ESTATE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    u64 p0, p1;
    shard_slice(c, &p0, &p1);
    bool sparse = false;
    for (u32 b = 0; b < SHARD_BINS; b++) {
        if (shard_of_bin[b] == 0xFF) sparse = true;
        else if (shard_of_bin[b] >= c->shard_world) return DEBWT_EINVAL;
    }
    if (capacity >= 0xFFFFFFF0ull) capacity = 0xFFFFFFF0ull - 1;     // a pass addresses its output with 32 bits
    ENSURE(c, c->dest_tab, SHARD_BINS);
    HIPCHK(c, hipMemcpyAsync(c->dest_tab.p, shard_of_bin, SHARD_BINS, hipMemcpyHostToDevice, c->stream));
    const int tshift = 2 * c->cfg.k - 12;
    TextKeySrc ts{c->text.as<u64>(), c->sepbits.as<u64>(), c->n, c->K, 0, 0, p0, nullptr,
                  sparse ? c->dest_tab.as<u8>() : nullptr, tshift};
    RsDigit dg{};
    dg.mode = 1; dg.tab = c->dest_tab.as<u8>(); dg.tshift = tshift;
    hipError_t e = radix_partition_by_shard(c->stream, nullptr, &ts, p1 - p0, (u64 *)d_out, dg, (u32)c->shard_world,
                                            radix_ws(c), (u64 *)offs, sparse, capacity);
    if (e == hipErrorInvalidValue && offs[c->shard_world] > capacity) {       // refused before the scatter wrote anything
        c->err = "partition output exceeds the caller's buffer";
        return DEBWT_EINTERNAL;
    }
    if (e != hipSuccess) { c->err = hipGetErrorString(e); return DEBWT_EDEVICE; }
    return DEBWT_OK;
}

// flags of the slice, in pieces of < 2^32 positions; totals of the slice
extern "C" int debwt_shard_sp_flags(debwt_ctx *c, uint64_t *sp_symbols, uint64_t *mi_positions) {
    if (!c) return DEBWT_EINVAL;
    if (c->stage < ST_CLASSIFIED) return DEBWT_ESTATE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    u64 p0, p1;
    shard_slice(c, &p0, &p1);
    int rc = sp_prepare(c);
    if (rc) return rc;
    c->sub.clear();
    c->S_rank = c->B_rank = 0;
    const u64 G0 = p0 >> 5, G1 = (p1 + 31) >> 5;
    if ((rc = sp_block_ids_begin(c, G0, G1 - G0 + 1))) return rc;     // block ids of the slice's multi-in positions
    for (u64 g0 = G0; g0 < G1 || c->sub.empty(); g0 += SP_SLICE_GROUPS) {
        const u64 g1 = std::min(G1, g0 + SP_SLICE_GROUPS);
        if ((rc = sp_flags(c, g0, g1))) return rc;
        c->sub.push_back(debwt_ctx::SubSlice{g0, g1, c->S_local, c->B_slice});
        c->S_rank += c->S_local; c->B_rank += c->B_slice;
    }
    if (sp_symbols) *sp_symbols = c->S_rank;
    if (mi_positions) *mi_positions = c->B_rank;
    return DEBWT_OK;
}

// bits of a routed blue entry: block id << qshift | SP index << 3 | pred
static int routed_qshift(const debwt_ctx *c) {
    int qbits = 1;
    while ((1ull << qbits) < c->Qtotal) qbits++;
    return 64 - qbits;
}

extern "C" int debwt_shard_sp_emit(debwt_ctx *c, uint64_t sp_offset, uint8_t *d_dst, uint64_t capacity) {
    // SP symbols of the slice at global offset sp_offset (a copy of them goes to the DEVICE buffer d_dst) and its
    // multi-in positions as routed blue entries (kept in the context for debwt_shard_blue_route)
    if (!c || !d_dst) return DEBWT_EINVAL;
    if (c->stage < ST_CLASSIFIED || c->sub.empty() || capacity < c->S_rank) return DEBWT_ESTATE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const int qshift = routed_qshift(c);
    if (sp_offset + c->S_rank >= (1ull << (qshift - 3))) {
        c->err = "a routed blue entry cannot hold block id and SP index in 61 bits";
        return DEBWT_ERANGE;
    }
    // the routed entries of the slice: in key buffer A when the keys were read off the text (the buffer is free since the
    // ranges were classified; debwt_shard_scratch hands B to the caller as its send buffer and A -- after the route -- as
    // its receive buffer, so that a sliced SP stage holds no exchange buffer of its own beside the key buffers)
    u64 *routed;
    if (!c->exchange && c->keysA.cap >= c->B_rank * 8 + 64) routed = c->keysA.as<u64>();
    else { ENSURE(c, c->facts_tmp, c->B_rank * 8 + 64); routed = c->facts_tmp.as<u64>(); }
    c->routed = routed;
    u64 off = sp_offset, bseen = 0;
    int rc;
    for (const auto &sl : c->sub) {
        // the scan of the piece's flag counts again (the compaction areas hold one piece at a time)
        c->g0 = sl.g0; c->g1 = sl.g1;
        SpCountF fc{c->momask.as<u32>() + sl.g0, c->mimask.as<u32>() + sl.g0};
        if ((rc = cp_count2(c, fc, sl.g1 - sl.g0, cp_area(c, 0), 8, cp_area(c, 1), 10))) return rc;
        c->S_local = sl.S; c->B_slice = sl.B;
        if ((rc = sp_emit(c, off, routed + bseen, qshift))) return rc;
        off += sl.S; bseen += sl.B;
        if ((rc = sync_check(c))) return rc;
    }
    c->sp_off = sp_offset;
    if (c->S_rank)
        HIPCHK(c, hipMemcpyAsync(d_dst, c->spsym.as<u8>() + sp_offset, c->S_rank, hipMemcpyDeviceToDevice, c->stream));
    return sync_check(c);
}

extern "C" int debwt_shard_sp_import(debwt_ctx *c, const uint8_t *d_src, uint64_t sp_total) {
    // the SP symbols of the whole text (slices concatenated in rank order), DEVICE buffer
    if (!c || (!d_src && sp_total)) return DEBWT_EINVAL;
    if (c->stage < ST_CLASSIFIED || sp_total > c->n) return DEBWT_ESTATE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    ENSURE(c, c->spsym, sp_total + 64);
    if (sp_total) HIPCHK(c, hipMemcpyAsync(c->spsym.p, d_src, sp_total, hipMemcpyDeviceToDevice, c->stream));
    int rc = sp_finish(c, sp_total);
    if (rc) return rc;
    return sync_check(c);
}

extern "C" int debwt_shard_blue_route(debwt_ctx *c, const uint32_t *first_block_of_shard, uint64_t *d_out,
                                      uint64_t capacity, uint64_t *offs) {
    // the routed blue entries of this shard's text slice, grouped by the shard that owns their block
    if (!c || !first_block_of_shard || !d_out || !offs) return DEBWT_EINVAL;
    if (c->stage < ST_CLASSIFIED || capacity < c->B_rank) return DEBWT_ESTATE;
    if (c->B_rank >= 0xFFFFFFF0ull) { c->err = "a text slice holds 2^32 multi-in positions or more"; return DEBWT_ERANGE; }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const u32 w = (u32)c->shard_world;
    ENSURE(c, c->qbounds, (w + 1) * 4);
    HIPCHK(c, hipMemcpyAsync(c->qbounds.p, first_block_of_shard, (w + 1) * 4, hipMemcpyHostToDevice, c->stream));
    for (u32 i = 0; i <= w; i++) offs[i] = 0;
    if (!c->B_rank) return sync_check(c);
    RsDigit dg{};
    dg.mode = 2; dg.bounds = c->qbounds.as<u32>(); dg.nb = w; dg.tshift = routed_qshift(c);
    if (!c->routed || (u64 *)d_out == c->routed) { c->err = "blue route: no routed entries, or the output is their own buffer"; return DEBWT_ESTATE; }
    hipError_t e = radix_partition_by_shard(c->stream, c->routed, nullptr, c->B_rank, (u64 *)d_out, dg, w,
                                            radix_ws(c), (u64 *)offs, false, capacity);
    if (e != hipSuccess) { c->err = hipGetErrorString(e); return DEBWT_EDEVICE; }
    return DEBWT_OK;
}

extern "C" int debwt_shard_blue_place(debwt_ctx *c, uint64_t *d_entries, uint64_t count) {
    if (!c || (!d_entries && count)) return DEBWT_EINVAL;
    if (c->stage < ST_SP) return DEBWT_ESTATE;
    if (count != c->B) { c->err = "received blue entries differ from the rows of the owned blocks"; return DEBWT_EINTERNAL; }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const int qshift = routed_qshift(c);
    // As on one GPU (debwt_sp_generate): sorting the routed words by their block bits puts every entry into its block --
    // the owned blocks are a contiguous range of block ids and their blue slots the exclusive scan of their sizes -- and
    // k_blue_strip leaves pred | spIndex << 4.  The passes alternate between the caller's buffer (scratch from here on)
    // and `blue`.  (One cursor atomic and one scattered 8-byte store per entry, k_blue_place, took 177 ms for the 3.1 G
    // entries of 30 Gbp against 91 ms; it stays for entries that the 32-bit passes cannot index and for cfg.reserved
    // bit 5.)
    if (count >= 2 && count < 0xFFFFFFF0ull - (1ull << 20) && !(c->cfg.reserved & 32)) {
        ENSURE(c, c->rs_over, radix_over_bytes(count));
        // the passes of debwt_sp_generate's entry sort: bits spread evenly, the strip folded into the stores of the last one,
        // which writes `blue`; an even number of passes takes its first hop through a free key buffer
        {
            u64 *third = nullptr;
            for (DevBuf *kb : {&c->keysB, &c->keysA})
                if (!third && kb->cap >= count * 8 + 64 && kb->as<u64>() != (u64 *)d_entries && !(c->exchange && kb == &c->keysA)) third = kb->as<u64>();
            hipError_t e = hipSuccess;
            if (radix_sort_bits_into(c->stream, (u64 *)d_entries, c->blue.as<u64>(), third, count, qshift, 64, radix_ws(c), &e, qshift)) {
                if (e != hipSuccess) { c->err = std::string("blue entry sort: ") + hipGetErrorString(e); return DEBWT_EDEVICE; }
                return sync_check(c);
            }
        }
        u64 *src = reinterpret_cast<u64 *>(d_entries), *dst = c->blue.as<u64>();
        for (int shift = qshift; shift < 64; shift += 8) {
            hipError_t e = hipSuccess;
            u64 *r = radix_sort_bits(c->stream, src, dst, count, shift, std::min(shift + 8, 64), radix_ws(c), &e);
            if (e != hipSuccess || r != dst) { c->err = std::string("blue entry sort: ") + hipGetErrorString(e); return DEBWT_EDEVICE; }
            std::swap(src, dst);
        }
        k_blue_strip<<<grid_for(count, 256), 256, 0, c->stream>>>(src, c->blue.as<u64>(), count, qshift);
        return sync_check(c);
    }
    ENSURE(c, c->qcursor, c->Q * 4 + 64);
    if (c->Q) HIPCHK(c, hipMemsetAsync(c->qcursor.p, 0, c->Q * 4, c->stream));
    if (count)
        k_blue_place<<<grid_for(count, DEBWT_BLOCK), DEBWT_BLOCK, 0, c->stream>>>((const u64 *)d_entries, count,
                                                                                 (u32)c->qbase, (u32)c->Q, qshift,
                                                                                 c->qcursor.as<u32>(), c->blk_start.as<u64>(),
                                                                                 c->blue.as<u64>());
    return sync_check(c);
}

extern "C" int debwt_shard_scratch(debwt_ctx *c, int which, void **ptr, uint64_t *bytes) {
    if (!c || !ptr || !bytes || which < 0 || which > 1) return DEBWT_EINVAL;
    *ptr = nullptr; *bytes = 0;
    if (c->stage < ST_CLASSIFIED || c->exchange) return DEBWT_OK;        // exchange mode: key buffer A is the caller's own
    DevBuf &b = which == DEBWT_SCRATCH_SEND ? c->keysB : c->keysA;
    *ptr = b.p; *bytes = b.cap;
    return DEBWT_OK;
}

// Final concatenation: the shards' packed row ranges, each packed from its own first row, are
// shift-merged into the BWT words of the whole text (the reference joins its per-thread SP segments the same way,
// src/generateSP.c:379-405).  d_parts: DEVICE words of all parts; part i starts at word part_word_off[i] (one spare
// readable word behind each part), holds rows [row_base[i], row_base[i] + rows[i]); d_out: ceil(n/32) DEVICE words.
extern "C" int debwt_concat_rows(debwt_ctx *c, const uint64_t *d_parts, uint32_t nparts, const uint64_t *part_word_off,
                                 const uint64_t *row_base, const uint64_t *rows, uint64_t n, uint64_t *d_out) {
    if (!c || !d_parts || !nparts || !part_word_off || !row_base || !rows || !d_out) return DEBWT_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    std::vector<ConcatPart> pp;
    u64 next = 0;
    for (u32 i = 0; i < nparts; i++) {
        if (!rows[i]) continue;
        if (row_base[i] != next) { c->err = "shard row ranges do not tile the BWT"; return DEBWT_EINVAL; }
        pp.push_back(ConcatPart{part_word_off[i], row_base[i], rows[i]});
        next += rows[i];
    }
    if (next != n) { c->err = "shard rows do not add up to n"; return DEBWT_EINVAL; }
    ENSURE(c, c->qbounds, pp.size() * sizeof(ConcatPart) + 64);
    HIPCHK(c, hipMemcpyAsync(c->qbounds.p, pp.data(), pp.size() * sizeof(ConcatPart), hipMemcpyHostToDevice, c->stream));
    const u64 nw = (n + 31) >> 5;
    k_concat_rows<<<grid_for(nw, 256), 256, 0, c->stream>>>((const u64 *)d_parts, c->qbounds.as<ConcatPart>(), (u32)pp.size(), n,
                                                            (u64 *)d_out);
    return sync_check(c);                                   // pp is host memory
}

extern "C" int debwt_shard_info(debwt_ctx *c, uint64_t *row_base, uint64_t *rows, uint64_t *hash_rows) {
    if (!c) return DEBWT_EINVAL;
    if (c->stage < ST_CLASSIFIED) return DEBWT_ESTATE;
    if (row_base) *row_base = c->Mbase + c->s0;
    if (rows) *rows = shard_rows(c);
    if (hash_rows) *hash_rows = c->n_hash_local;
    return DEBWT_OK;
}

extern "C" int debwt_shard_fetch(debwt_ctx *c, uint64_t *words, uint64_t *hash_rows, uint64_t *dollar_row) {
    // words: ceil(rows/32), row j of the shard at bit 2*(31-(j&31)) of word j>>5; hash_rows / dollar_row are
    // GLOBAL rows (dollar_row = ~0 when the '$' row is not in this shard)
    if (!c || !dollar_row) return DEBWT_EINVAL;             // words == NULL: only the row lists
    if (c->stage < ST_ASSEMBLED) return DEBWT_ESTATE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const u64 rows = shard_rows(c), base = c->Mbase + c->s0;
    if (words) HIPCHK(c, hipMemcpyAsync(words, c->bwt.p, (size_t)((rows + 31) >> 5) * 8, hipMemcpyDeviceToHost, c->stream));
    if (c->n_hash_local && hash_rows)
        HIPCHK(c, hipMemcpyAsync(hash_rows, c->hash_rows.p, c->n_hash_local * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(dollar_row, c->dollar.p, 8, hipMemcpyDeviceToHost, c->stream));
    int rc = sync_check(c);
    if (rc) return rc;
    for (u64 i = 0; i < c->n_hash_local && hash_rows; i++) hash_rows[i] += base;
    if (*dollar_row != ~0ull) *dollar_row += base;
    return DEBWT_OK;
}

extern "C" int debwt_pinned_alloc(size_t bytes, void **out) {
    if (!out || !bytes) return DEBWT_EINVAL;
    *out = nullptr;
    return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? DEBWT_OK : DEBWT_ENOMEM;
}
extern "C" void debwt_pinned_free(void *p) { if (p) (void)hipHostFree(p); }

extern "C" int debwt_shard_export(debwt_ctx *c, uint64_t *d_words, uint64_t capacity) {
    // the shard's packed rows to a DEVICE buffer (for the gather of the final concat); the words behind the last
    // row up to `capacity` are zeroed
    if (!c || !d_words) return DEBWT_EINVAL;
    if (c->stage < ST_ASSEMBLED) return DEBWT_ESTATE;
    const u64 nw = (shard_rows(c) + 31) >> 5;
    if (capacity < nw) return DEBWT_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (nw) HIPCHK(c, hipMemcpyAsync(d_words, c->bwt.p, nw * 8, hipMemcpyDeviceToDevice, c->stream));
    if (capacity > nw) HIPCHK(c, hipMemsetAsync(d_words + nw, 0, (capacity - nw) * 8, c->stream));
    return sync_check(c);
}

extern "C" int debwt_census_words(debwt_ctx *c, const uint64_t *d_words, uint64_t n, uint64_t counts[4]) {
    if (!c || !d_words || !counts) return DEBWT_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    u64 *d4 = c->dollar.as<u64>() + 1;
    HIPCHK(c, hipMemsetAsync(d4, 0, 32, c->stream));
    k_bwt_census<<<2048, DEBWT_BLOCK, 0, c->stream>>>((const u64 *)d_words, n, d4);
    HIPCHK(c, hipMemcpyAsync(counts, d4, 32, hipMemcpyDeviceToHost, c->stream));
    return sync_check(c);
}

extern "C" int debwt_get_stats(const debwt_ctx *c, debwt_stats *out) {
    if (!c || !out) return DEBWT_EINVAL;
    *out = c->st;
    return DEBWT_OK;
}

// ---------------------------------------------------------------------------------------------------

extern "C" int debwt_fetch_array(debwt_ctx *c, debwt_array which, void *dst, uint64_t capacity, uint64_t *count) {
    if (!c || !count) return DEBWT_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const void *src = nullptr;
    uint64_t cnt = 0, esz = 8;
    Stage need = ST_SORTED;
    std::vector<uint64_t> host;
    switch (which) {
        case DEBWT_ARR_SORTED_KEYS:
        case DEBWT_ARR_DISTINCT_KEYS:
            if (c->ranges.size() != 1) { c->err = "sorted keys are not kept by a multi-range build"; return DEBWT_ESTATE; }
            // the key buffers are scratch for the later stages (blue-entry sort), and in exchange mode they are the caller's
            if (which == DEBWT_ARR_SORTED_KEYS && (c->stage > ST_CLASSIFIED || c->exchange || c->shard_world > 1)) {
                c->err = "sorted keys are only kept until the SP stage reuses their buffer (and not by a shard)";
                return DEBWT_ESTATE;
            }
            if (which == DEBWT_ARR_SORTED_KEYS) { src = c->sk; cnt = c->M; } else { src = c->dk.p; cnt = c->D; }
            if (!src && cnt) { c->err = "the keys were released when the build ran out of memory"; return DEBWT_ESTATE; }
            break;
        case DEBWT_ARR_RED: src = c->red.p; cnt = c->R; need = ST_CLASSIFIED; break;
        case DEBWT_ARR_SP_SYMBOLS: src = c->spsym.p; cnt = c->S; esz = 1; need = ST_SP; break;
        case DEBWT_ARR_BLUE: src = c->blue.p; cnt = c->B; need = ST_SP; break;
        case DEBWT_ARR_BLUE_BOUND:
        case DEBWT_ARR_CASE3_BOUND: {
            need = ST_CLASSIFIED;
            if (c->stage < need) return DEBWT_ESTATE;
            std::vector<u32> fr(c->Q);
            std::vector<uint64_t> bs(c->Q), j0(c->Q);
            std::vector<uint64_t> sprow(c->NS);
            if (c->Q) {
                HIPCHK(c, hipMemcpy(fr.data(), c->blk_freq.p, c->Q * 4, hipMemcpyDeviceToHost));
                HIPCHK(c, hipMemcpy(bs.data(), c->blk_start.p, c->Q * 8, hipMemcpyDeviceToHost));
                HIPCHK(c, hipMemcpy(j0.data(), c->blk_j0.p, c->Q * 8, hipMemcpyDeviceToHost));
            }
            HIPCHK(c, hipMemcpy(sprow.data(), c->sprow.p, c->NS * 8, hipMemcpyDeviceToHost));
            if (which == DEBWT_ARR_BLUE_BOUND) {
                host.resize(c->Q);
                for (u64 q = 0; q < c->Q; q++) host[q] = bs[q] + fr[q] - 1;   // src/INandOut.c:359-361
            } else {
                // rows: instance j sits below every special suffix whose rank among the instances is <= j
                std::vector<uint64_t> mrank(c->NS);
                for (u64 s = 0; s < c->NS; s++) mrank[s] = sprow[s] - s;
                host.resize(2 * c->Q);
                for (u64 q = 0; q < c->Q; q++) {
                    uint64_t before = std::upper_bound(mrank.begin(), mrank.end(), j0[q]) - mrank.begin();
                    host[2 * q] = j0[q] + before;                                        // src/INandOut.c:349-352
                    host[2 * q + 1] = host[2 * q] + fr[q] - 1;
                }
            }
            cnt = host.size();
            *count = cnt;
            if (dst) memcpy(dst, host.data(), (size_t)std::min(cnt, capacity) * 8);
            return DEBWT_OK;
        }
        case DEBWT_ARR_ROW_SYMBOLS: {
            if (c->stage < ST_ASSEMBLED) return DEBWT_ESTATE;
            ENSURE(c, c->rowsym, c->n + 64);
            run_assemble(c, c->rowsym.as<u8>(), 0, ~0ull, false);      // (symbols only: the '#' rows of the build stay as they are)
            int rc = sync_check(c);
            if (rc) return rc;
            src = c->rowsym.p; cnt = c->n; esz = 1; need = ST_ASSEMBLED;
            break;
        }
        default: return DEBWT_EINVAL;
    }
    if (c->stage < need) return DEBWT_ESTATE;
    *count = cnt;
    uint64_t ncopy = std::min(cnt, capacity);
    if (dst && ncopy) {
        HIPCHK(c, hipMemcpyAsync(dst, src, (size_t)ncopy * esz, hipMemcpyDeviceToHost, c->stream));
        return sync_check(c);
    }
    return DEBWT_OK;
}

extern "C" int debwt_kmer_count_sorted(debwt_ctx *c, uint64_t *kmers, uint64_t *counts, uint64_t capacity,
                                       uint64_t *distinct) {
    if (!c || !distinct) return DEBWT_EINVAL;
    if (c->stage < ST_LOADED) return DEBWT_ESTATE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const int k = c->cfg.k;
    if (c->n <= c->nrec * (u64)k) return DEBWT_EINVAL;
    const u64 Mk = c->n - c->nrec * (u64)k;
    if (c->n >= 0xFFFFFFF0ull) return DEBWT_ERANGE;
    c->stage = ST_LOADED;   // the pipeline buffers are reused
    ENSURE(c, c->keysA, c->n * 8 + 64);
    ENSURE(c, c->keysB, c->n * 8 + 64);
    ENSURE(c, c->rs_skew, (c->n / 2048 + 2) * 4);
    ENSURE(c, c->dk, c->n * 8 + 64);
    ENSURE(c, c->dstart, c->n * 4 + 64);
    int rc;
    k_extract_keys<<<grid_for(c->n, DEBWT_BLOCK), DEBWT_BLOCK, 0, c->stream>>>(
        c->text.as<u64>(), c->sepbits.as<u64>(), c->sep.as<u64>(), c->nrec, c->n, k, 1, c->keysA.as<u64>());
    u64 *sorted = nullptr;
    if ((rc = sort_keys(c, c->keysA.as<u64>(), c->keysB.as<u64>(), Mk, 2 * k, &sorted, false))) return rc;
    KmerInfoF f{sorted, Mk, k, c->dk.as<u64>(), c->dstart.as<u32>()};
    if ((rc = cp_count(c, f, Mk, cp_area(c, 0), 0))) return rc;
    if ((rc = cp_emit(c, f, Mk, cp_area(c, 0)))) return rc;
    if ((rc = sync_check(c))) return rc;
    u64 D = c->h_scalars[0];
    *distinct = D;
    u64 ncopy = std::min<u64>(D, capacity);
    if (kmers && ncopy) HIPCHK(c, hipMemcpy(kmers, c->dk.p, ncopy * 8, hipMemcpyDeviceToHost));
    if (counts && ncopy) {
        std::vector<u32> first(D);
        HIPCHK(c, hipMemcpy(first.data(), c->dstart.p, D * 4, hipMemcpyDeviceToHost));
        for (u64 i = 0; i < ncopy; i++) counts[i] = (i + 1 < D ? first[i + 1] : (u32)Mk) - first[i];
    }
    return DEBWT_OK;
}

extern "C" int debwt_radix_sort_u64(debwt_ctx *c, uint64_t *d_keys, uint64_t *d_tmp, uint64_t count, int key_bits,
                                    float *ms_per_pass) {
    if (!c || !d_keys || !d_tmp || key_bits < 1 || key_bits > 64) return DEBWT_EINVAL;
    if (count >= 0xFFFFFFF0ull) return DEBWT_ERANGE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    ENSURE(c, c->rs_skew, (count / 2048 + 2) * 4);
    ENSURE(c, c->rs_over, radix_over_bytes(count));
    RadixWorkspace ws = radix_ws(c);
    hipError_t e = hipSuccess;
    int np = 0;
    u64 *res = radix_sort_u64(c->stream, (u64 *)d_keys, (u64 *)d_tmp, count, key_bits, ws, c->cfg.sort_algo,
                              ms_per_pass ? &c->ev_pass[0][0] : nullptr, 16, &np, &e);
    if (e != hipSuccess) { c->err = hipGetErrorString(e); return DEBWT_EDEVICE; }
    if (res != (u64 *)d_keys)
        HIPCHK(c, hipMemcpyAsync(d_keys, res, count * 8, hipMemcpyDeviceToDevice, c->stream));
    int rc = sync_check(c);
    if (rc) return rc;
    if (ms_per_pass) {
        float sum = 0.f;
        for (int i = 0; i < np; i++) { float t = 0.f; (void)hipEventElapsedTime(&t, c->ev_pass[i][0], c->ev_pass[i][1]); sum += t; }
        *ms_per_pass = np ? sum / np : 0.f;
    }
    return DEBWT_OK;
}

extern "C" int debwt_special_digest(const uint64_t *packed, uint64_t n, const uint64_t *sep, uint64_t nrec, int k,
                                    uint64_t digest[4]) {
    if (!packed || !sep || !digest || nrec == 0 || k < 12 || k > 32) return DEBWT_EINVAL;
    SpecialTables t;
    build_special_tables(packed, n, sep, nrec, k - 1, &t);
    auto mix = [](uint64_t h, uint64_t v) { h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); return h * 0xBF58476D1CE4E5B9ull; };
    uint64_t d[4] = {1, 2, 3, 4};
    for (uint64_t v : t.pos) d[0] = mix(d[0], v);
    for (size_t i = 0; i < t.key.size(); i++) d[1] = mix(mix(d[1], t.key[i]), t.chr[i]);
    for (uint64_t v : t.branch) d[2] = mix(d[2], v);
    for (uint64_t v : t.head_keys) d[3] = mix(d[3], v);
    for (uint64_t v : t.tail_facts) d[3] = mix(d[3], v);
    for (int i = 0; i < 4; i++) digest[i] = d[i];
    return DEBWT_OK;
}

// The special-region tables of the loaded text built twice -- on the device (special_kernels.h) and by the host module
// (special_host.cpp) -- and compared element by element: mismatch[0..5] = suffix order, keys, BWT symbols, special
// branches, head nodes, tail nodes (a table of different length counts as all of it).
extern "C" int debwt_special_compare(debwt_ctx *c, uint64_t mismatch[6]) {
    if (!c || !mismatch) return DEBWT_EINVAL;
    if (c->stage < ST_LOADED) return DEBWT_ESTATE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    join_special(c);
    if (c->nrec >= (1ull << 27) || c->NS >= (1ull << 32)) return DEBWT_ERANGE;
    // the device tables overwrite those of a build in progress: the context goes back to "loaded" (a build that follows starts
    // over and chooses its own path), and the counters of the last build stay what they were
    const debwt_stats keep = c->st;
    c->stage = ST_LOADED;
    int rc = special_device_build(c, false);
    c->special_dev = false;
    c->st = keep;
    if (rc) return rc;
    SpecialTables t;
    build_special_tables(c->h_text, c->n, c->h_sep.data(), c->nrec, c->K, &t);
    const u64 NS = c->NS, N = c->nrec;
    std::vector<u64> pos(NS), key(NS), br(c->nbranch), hk(N), tf(N);
    std::vector<u8> chr(NS);
    HIPCHK(c, hipMemcpy(pos.data(), c->sppos.p, NS * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(key.data(), c->spkey.p, NS * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(chr.data(), c->spchr.p, NS, hipMemcpyDeviceToHost));
    if (c->nbranch) HIPCHK(c, hipMemcpy(br.data(), c->branch.p, c->nbranch * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(hk.data(), c->head_keys.p, N * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(tf.data(), c->tail_d.p, N * 8, hipMemcpyDeviceToHost));
    auto diff = [](const auto &a, const auto &b) {
        if (a.size() != b.size()) return (uint64_t)std::max(a.size(), b.size());
        uint64_t d = 0;
        for (size_t i = 0; i < a.size(); i++) d += a[i] != b[i];
        return d;
    };
    if (getenv("DEBWT_SPECIAL_DEBUG")) {
        int shown = 0;
        for (u64 s_ = 0; s_ < NS && shown < 12; s_++)
            if (pos[s_] != t.pos[s_]) {
                auto recof = [&](u64 p) { return (u64)(std::lower_bound(c->h_sep.begin(), c->h_sep.end(), p) - c->h_sep.begin()); };
                const u64 rd = recof(pos[s_]), rh = recof(t.pos[s_]);
                fprintf(stderr, "special mismatch at %llu: device pos %llu (rec %llu d %llu) host pos %llu (rec %llu d %llu) key %llx / %llx\n",
                        (unsigned long long)s_, (unsigned long long)pos[s_], (unsigned long long)rd, (unsigned long long)(c->h_sep[rd] - pos[s_]),
                        (unsigned long long)t.pos[s_], (unsigned long long)rh, (unsigned long long)(c->h_sep[rh] - t.pos[s_]),
                        (unsigned long long)key[s_], (unsigned long long)t.key[s_]);
                shown++;
            }
    }
    mismatch[0] = diff(pos, t.pos); mismatch[1] = diff(key, t.key); mismatch[2] = diff(chr, t.chr);
    mismatch[3] = diff(br, t.branch); mismatch[4] = diff(hk, t.head_keys); mismatch[5] = diff(tf, t.tail_facts);
    return DEBWT_OK;
}

// Inverse BWT on the device (verify_kernels.h): rank structure over the packed rows, backward search for the segment
// boundaries, one LF walk per segment against the text in HBM.
extern "C" int debwt_verify_device(debwt_ctx *c, const uint64_t *d_words, const uint64_t *hash_rows, uint64_t dollar_row,
                                   uint64_t segments, debwt_verify_report *rep) {
    if (!c || !rep) return DEBWT_EINVAL;
    memset(rep, 0, sizeof *rep);
    if (c->stage < ST_LOADED) return DEBWT_ESTATE;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const u64 n = c->n, nrec = c->nrec;
    std::vector<u64> hr;
    if (!d_words) {                                           // the context's own result
        if (c->stage < ST_ASSEMBLED || c->shard_world > 1) return DEBWT_ESTATE;
        d_words = c->bwt.as<uint64_t>();
        hr.resize(nrec);
        if (nrec > 1) HIPCHK(c, hipMemcpy(hr.data(), c->hash_rows.p, (nrec - 1) * 8, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(&dollar_row, c->dollar.p, 8, hipMemcpyDeviceToHost));
        hr.resize(nrec - 1);
    } else {
        if (nrec > 1 && !hash_rows) return DEBWT_EINVAL;
        hr.assign(hash_rows, hash_rows + (nrec - 1));
    }
    if (dollar_row >= n) { c->err = "'$' row outside the BWT"; return DEBWT_EINVAL; }
    for (size_t i = 0; i < hr.size(); i++)
        if (hr[i] >= n || (i && hr[i] <= hr[i - 1])) { c->err = "'#' rows are not ascending rows of the BWT"; return DEBWT_EINVAL; }
    std::vector<u64> sr(hr);
    sr.insert(std::upper_bound(sr.begin(), sr.end(), dollar_row), dollar_row);
    // rank structure
    const u64 nlines = n / VB_ROWS + 1, nchunks = (nlines + VB_CHUNK - 1) / VB_CHUNK;
    const size_t idx_bytes = (size_t)nlines * VB_LINE * 8;
    u64 *idx;
    if (c->keysB.cap >= idx_bytes) idx = c->keysB.as<u64>();              // free between builds
    else if (c->keysA.cap >= idx_bytes) idx = c->keysA.as<u64>();
    else { ENSURE(c, c->vidx, idx_bytes); idx = c->vidx.as<u64>(); }
    if (!segments) segments = std::min<u64>(std::max<u64>(n / 16384, 1), 1ull << 20);
    const u64 gap = std::max<u64>(n / segments, 2);
    const u64 nbound = n > 2 ? (n - 2) / gap : 0;                          // (j + 1) * gap < n - 1
    const u32 maxm = (u32)std::min<u64>(std::max<u64>(gap / 2, 1), 1u << 16);
    // small arrays: [csum 5 x nchunks][totals 8][counters 8][hash nrec][srows nrec + 1][bounds 2 x (nbound + 2)]
    const size_t small_words = (size_t)nchunks * 5 + 16 + 2 * nrec + 2 + 2 * (nbound + 2) + 8;
    ENSURE(c, c->vtmp, small_words * 8);
    u64 *csum = c->vtmp.as<u64>(), *totals = csum + nchunks * 5, *counters = totals + 8, *d_hash = counters + 8,
        *d_srows = d_hash + nrec, *d_bounds = d_srows + nrec + 1;
    HIPCHK(c, hipMemsetAsync(totals, 0, 16 * 8, c->stream));
    if (!hr.empty()) HIPCHK(c, hipMemcpyAsync(d_hash, hr.data(), hr.size() * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_srows, sr.data(), sr.size() * 8, hipMemcpyHostToDevice, c->stream));
    hipEvent_t ev[4];
    for (auto &e : ev) HIPCHK(c, hipEventCreate(&e));
    auto drop_events = [&]() { for (auto &e : ev) (void)hipEventDestroy(e); };
    HIPCHK(c, hipEventRecord(ev[0], c->stream));
    const u64 *bw = reinterpret_cast<const u64 *>(d_words);
    k_vidx_count<<<(u32)nchunks, VB_CHUNK, 0, c->stream>>>(bw, n, nlines, d_srows, sr.size(), csum);
    k_vidx_scan<<<1, 1024, 0, c->stream>>>(csum, nchunks, totals);
    k_vidx_write<<<(u32)nchunks, VB_CHUNK, 0, c->stream>>>(bw, n, nlines, d_srows, sr.size(), csum, idx);
    u64 tot[5];
    HIPCHK(c, hipMemcpyAsync(tot, totals, 40, hipMemcpyDeviceToHost, c->stream));
    int rc = sync_check(c);
    if (rc) { drop_events(); return rc; }
    if (tot[4] != nrec || tot[0] + tot[1] + tot[2] + tot[3] != n) { drop_events(); c->err = "rank structure: row census is inconsistent"; return DEBWT_EINTERNAL; }
    VIndex V{};
    V.idx = idx; V.hash = d_hash; V.srows = d_srows; V.nhash = hr.size(); V.nsep = sr.size(); V.n = n;
    V.C[0] = 0; V.C[1] = tot[0]; V.C[2] = tot[0] + tot[1]; V.C[3] = V.C[2] + tot[2];
    V.C[4] = V.C[3] + (tot[3] - nrec); V.C[5] = n - 1;
    V.dollar_row = dollar_row;
    HIPCHK(c, hipEventRecord(ev[1], c->stream));
    // segment boundaries by backward search
    std::vector<u64> found(2 * nbound), bounds;
    if (nbound) {
        k_vsearch<<<grid_for(nbound, 256), 256, 0, c->stream>>>(V, c->text.as<u64>(), c->sepbits.as<u64>(), nbound, gap, maxm,
                                                               d_bounds, counters);
        HIPCHK(c, hipMemcpyAsync(found.data(), d_bounds, nbound * 16, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipEventRecord(ev[2], c->stream));
    if ((rc = sync_check(c))) { drop_events(); return rc; }
    bounds.push_back(0); bounds.push_back(dollar_row);
    u64 last = 0;
    for (u64 j = 0; j < nbound; j++)
        if (found[2 * j] != ~0ull && found[2 * j] > last && found[2 * j] < n - 1) {
            bounds.push_back(found[2 * j]); bounds.push_back(found[2 * j + 1]);
            last = found[2 * j];
        }
    bounds.push_back(n - 1); bounds.push_back(n - 1);                      // the suffix "$" is the last row
    const u64 nseg = bounds.size() / 2 - 1;
    HIPCHK(c, hipMemcpyAsync(d_bounds, bounds.data(), bounds.size() * 8, hipMemcpyHostToDevice, c->stream));
    k_vwalk<<<grid_for(nseg, 256), 256, 0, c->stream>>>(V, c->text.as<u64>(), c->sepbits.as<u64>(), d_bounds, nseg, counters);
    HIPCHK(c, hipEventRecord(ev[3], c->stream));
    u64 ctr[8];
    HIPCHK(c, hipMemcpyAsync(ctr, counters, 64, hipMemcpyDeviceToHost, c->stream));
    if ((rc = sync_check(c))) { drop_events(); return rc; }                 // bounds is host memory
    (void)hipEventElapsedTime(&rep->ms_index, ev[0], ev[1]);
    (void)hipEventElapsedTime(&rep->ms_search, ev[1], ev[2]);
    (void)hipEventElapsedTime(&rep->ms_walk, ev[2], ev[3]);
    drop_events();
    rep->segments = nseg; rep->steps = ctr[4]; rep->mismatches = ctr[0]; rep->broken_links = ctr[1];
    rep->search_failures = ctr[2]; rep->search_steps = ctr[3];
    rep->ok = (ctr[0] == 0 && ctr[1] == 0 && ctr[2] == 0 && ctr[4] == n - 1) ? 1 : 0;
    return DEBWT_OK;
}

extern "C" int debwt_verify_inverse(const uint64_t *bwt, uint64_t n, const uint64_t *hash_rows, uint64_t nrec,
                                    uint64_t dollar_row, uint8_t *sym_out) {
    // LF(i) = C[c] + occ(c, i); '#' rows map in order to rows n-nrec.., '$' row to row n-1
    // (src/LFsearch.c:49-166, src/insertCase3.c:141-194).  The reference walks the whole text as one chain of n
    // dependent steps; here every record is walked by its own thread: the walk that starts at the row of
    // the j-th '#' suffix (row n - nrec + j) spells the record in front of that '#' backwards and stops at the row that
    // carries the previous separator, which names the walk that precedes it in the text.
    if (!bwt || !sym_out || n < 2 || nrec < 1 || dollar_row >= n) return DEBWT_EINVAL;
    std::vector<uint8_t> L(n);
    for (uint64_t j = 0; j < n; j++) L[j] = (uint8_t)((bwt[j >> 5] >> ((31 - (j & 31)) << 1)) & 3);
    for (uint64_t h = 0; h + 1 < nrec; h++) { if (hash_rows[h] >= n) return DEBWT_EINVAL; L[hash_rows[h]] = 4; }
    L[dollar_row] = 5;
    uint64_t C[7] = {0}, cnt[6] = {0}, seen[6] = {0};
    for (uint64_t i = 0; i < n; i++) cnt[L[i]]++;
    for (int s = 0; s < 6; s++) C[s + 1] = C[s] + cnt[s];
    if (cnt[4] != nrec - 1 || cnt[5] != 1) return DEBWT_EINTERNAL;
    std::vector<uint64_t> lf(n);
    for (uint64_t i = 0; i < n; i++) lf[i] = C[L[i]] + seen[L[i]]++;
    // walk w (0 .. nrec-1) starts at row C[4] + w: the '#' suffixes in row order, then the '$' suffix (row n-1)
    struct Walk { std::vector<uint8_t> rev; int64_t prev = -2; bool ok = false; };
    std::vector<Walk> walks(nrec);
    std::atomic<uint64_t> next_walk{0};
    auto worker = [&]() {
        for (;;) {
            uint64_t w = next_walk.fetch_add(1);
            if (w >= nrec) return;
            Walk &wk = walks[w];
            uint64_t row = C[4] + w;
            for (uint64_t steps = 0; steps <= n; steps++) {
                uint8_t sy = L[row];
                if (sy == 5) { wk.prev = -1; wk.ok = true; break; }                    // the text's first record
                if (sy == 4) { wk.prev = (int64_t)(lf[row] - C[4]); wk.ok = true; break; }
                wk.rev.push_back(sy);
                row = lf[row];
            }
        }
    };
    {
        unsigned nt = std::max(1u, std::min<unsigned>((unsigned)std::min<uint64_t>(nrec, 64), std::thread::hardware_concurrency()));
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back(worker);
        worker();
        for (auto &x : th) x.join();
    }
    // chain the records from the last one (the '$' walk) to the first
    uint64_t p = n;
    int64_t w = (int64_t)nrec - 1;
    for (uint64_t r = 0; r < nrec; r++) {
        if (w < 0 || (uint64_t)w >= nrec || !walks[w].ok) return DEBWT_EINTERNAL;
        const Walk &wk = walks[w];
        if (p < wk.rev.size() + 1) return DEBWT_EINTERNAL;
        sym_out[--p] = (r == 0) ? 5 : 4;                                               // the separator behind the record
        for (size_t i = 0; i < wk.rev.size(); i++) sym_out[--p] = wk.rev[i];
        w = wk.prev;
    }
    return (p == 0 && w == -1) ? DEBWT_OK : DEBWT_EINTERNAL;
}
