// multi_host.cpp -- one BWT over the GPUs of one node from ONE host process: debwt_multi_* (include/debwt_hip.h).
//
// The reference is a single process with a thread pool (src/main.c:30, pthreads); its multi-GPU counterpart keeps that
// shape: one host thread per GPU drives that GPU's context through the debwt_shard_* stage calls -- the same sequence
// debwt_amd/sharded.py runs with one process per GPU -- and the exchanges between the stages are direct peer-to-peer
// copies over xGMI (every GPU PULLS its segments out of the senders' buffers: an alltoallv is world x (world - 1)
// independent device-to-device copies, which is what RCCL's alltoallv decomposes into on a point-to-point fabric).
// The threads meet at a barrier around every exchange.  Only the public C ABI and the HIP runtime are used here.
#include "../../include/debwt_hip.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>
// RCCL: types and prototypes only -- the library is loaded on demand (dlopen), never linked; where its header is not
// installed the handful of declarations this file needs is stated here (RCCL's public API, values as in rccl.h)
#if defined(__has_include) && __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1 } ncclDataType_t;
ncclResult_t ncclCommInitAll(ncclComm_t *comm, int ndev, const int *devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclCommAbort(ncclComm_t comm);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
const char *ncclGetErrorString(ncclResult_t result);
}
#endif

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int BINS = 4096;
constexpr int MAXR = 64;

struct DevMem {
    void *p = nullptr;
    size_t cap = 0;
    int dev = 0;
    bool ensure(size_t bytes) {
        if (bytes <= cap) return true;
        (void)hipSetDevice(dev);
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 16 + 256;
        if (hipMalloc(&p, want) != hipSuccess) return false;
        cap = want;
        return true;
    }
    void release() { if (p) { (void)hipSetDevice(dev); (void)hipFree(p); p = nullptr; cap = 0; } }
};

// all threads call sync(rc); it returns false on every thread as soon as any thread brought an error.  The verdict of a
// barrier is latched by the thread that closes it: a thread that runs ahead and fails before a slow waiter of the
// barrier before has woken up must not change what that waiter returns (it would leave alone while the others wait for
// it at the next barrier).
//
// Serial mode (debwt_multi_set_serial; measurement of the shards one by one on a box with fewer GPUs than shards): between
// two barriers the shards take turns in shard order -- enter(r) waits until the shards before r have reached the barrier --
// so that only one of them is on the GPU at a time and the wall time of its steps is its own.
struct Rendezvous {
    std::mutex m;
    std::condition_variable cv;
    int n = 1, waiting = 0, phase = 0;
    int failed = 0;
    bool ok_of_phase[2] = {true, true};
    bool serial = false;
    int turn = 0;
    void enter(int r) {
        if (!serial) return;
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return turn == r; });
    }
    bool sync(int rc) {
        std::unique_lock<std::mutex> lk(m);
        if (rc && !failed) failed = rc;
        const int ph = phase;
        if (serial) { turn++; cv.notify_all(); }
        if (++waiting == n) { waiting = 0; turn = 0; ok_of_phase[ph & 1] = failed == 0; phase++; cv.notify_all(); }
        else cv.wait(lk, [&] { return phase != ph; });
        return ok_of_phase[ph & 1];
    }
};

// RCCL entry points, resolved at run time when a multi-GPU build asks for them (DEBWT_EXCHANGE_RCCL): the library has no
// link-time dependency on RCCL, so it loads on hosts without it, and inside a process that already holds a copy (PyTorch
// brings its own librccl.so.1) the loader hands back that one.
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    bool ready = false;                     // every symbol resolved: only then is any of the pointers used
    bool load(std::string *err) {
        if (ready) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { const char *e = dlerror(); *err = std::string("RCCL is not available: ") + (e ? e : "dlopen failed"); return false; }
        const char *missing = nullptr;
#define RCCL_SYM(f) f = reinterpret_cast<decltype(f)>(dlsym(lib, "nccl" #f)); if (!f && !missing) missing = "nccl" #f;
        RCCL_SYM(CommInitAll) RCCL_SYM(CommDestroy) RCCL_SYM(GroupStart) RCCL_SYM(GroupEnd) RCCL_SYM(Send) RCCL_SYM(Recv)
        RCCL_SYM(GetErrorString) RCCL_SYM(CommAbort)
#undef RCCL_SYM
        if (missing) {                      // a half-loaded library is no library: the next call starts over
            *err = std::string("librccl lacks ") + missing;
            dlclose(lib);
            lib = nullptr;
            return false;
        }
        ready = true;
        return true;
    }
};
// RCCL 2.26 (ROCm 7.0) drops the second half of a point-to-point message above 1 GiB: pieces of 512 MiB
constexpr size_t RCCL_PIECE = (size_t)512 << 20;

}  // namespace

struct debwt_multi {
    int G = 0;
    debwt_config cfg{};
    std::vector<debwt_ctx *> ctx;
    std::vector<int> dev;
    std::vector<hipStream_t> stream;
    std::vector<DevMem> xa, xb, facts, allfacts, sp, allsp, part;
    std::vector<void *> bsend;             // per shard: where its routed blue entries lie for the peers (a key buffer of its context, or xa)
    DevMem parts, out;                     // on the first GPU: the gathered row ranges, the concatenated BWT
    uint64_t n = 0, nrec = 0;
    debwt_packed_text own{};                // text packed by debwt_multi_load_fasta (the contexts read it during a build)
    std::vector<uint64_t> hash_rows;
    uint64_t dollar_row = ~0ull;
    std::string err;
    bool built = false;
    int key_mode = -1;                      // DEBWT_KEYS_EXCHANGE / DEBWT_KEYS_RESCAN, -1: debwt_shard_key_mode decides
    int exchange_backend = DEBWT_EXCHANGE_PEER_COPY;
    Rccl rccl;
    std::vector<ncclComm_t> comm;           // one communicator per shard (ncclCommInitAll), created by the first RCCL build
    bool comm_dead = false;                 // an exchange failed inside its group: the communicators were aborted (abort_comms)
    int posting = 0;                        // shards between "communicators alive" and the end of their group (under rv.m)
    std::condition_variable posted;         // ... signalled when one of them leaves
    debwt_multi_stats st{};
    std::vector<debwt_shard_report> rep;    // per shard: wall ms per step, bytes per exchange, sizes (debwt_multi_get_shard_report)
    // what the threads publish for each other
    std::vector<std::vector<uint64_t>> hist, offs, boffs;
    std::vector<std::vector<uint32_t>> cuts;
    std::vector<uint64_t> nfacts, nblocks, brows, slen, rowbase, rows, nhash;
    std::vector<std::vector<uint64_t>> hrows;
    std::vector<uint64_t> drow;
    Rendezvous rv;
};

namespace {

void set_err(debwt_multi *m, int r, const char *what) {
    std::lock_guard<std::mutex> lk(m->rv.m);
    if (m->err.empty()) m->err = std::string("shard ") + std::to_string(r) + ": " + what + ": " + debwt_last_error(m->ctx[r]);
}

// The exchange between two stages, called by every shard thread: shard d receives from every source s the cnt_of(s, d)
// elements of `esz` bytes that start at element off_of(s, d) of ptr_of(s), in source order, into dst (may be NULL where
// the shard receives nothing).  Returns the number of elements received, or -1.
//   peer copies: the receiver pulls with device-to-device copies -- world x (world - 1) independent copies over xGMI;
//   RCCL:        every shard posts its sends and receives in one group per 512-MiB piece (an alltoallv).
template <class PtrOf, class OffOf, class CntOf>
long long xchg(debwt_multi *m, int r, void *dst, size_t esz, PtrOf ptr_of, OffOf off_of, CntOf cnt_of) {
    (void)hipSetDevice(m->dev[r]);
    const int G = m->G;
    uint64_t o = 0;
    if (m->exchange_backend == DEBWT_EXCHANGE_PEER_COPY) {
        for (int s = 0; s < G; s++) {
            const uint64_t c = cnt_of(s, r);
            if (c && hipMemcpyAsync((char *)dst + o * esz, (const char *)ptr_of(s) + off_of(s, r) * esz, c * esz, hipMemcpyDefault,
                                    m->stream[r]) != hipSuccess) return -1;
            o += c;
        }
        if (hipStreamSynchronize(m->stream[r]) != hipSuccess) return -1;
        return (long long)o;
    }
    // every shard derives the same number of pieces from the published counts
    size_t pieces = 0;
    for (int s = 0; s < G; s++)
        for (int d = 0; d < G; d++) pieces = std::max(pieces, (size_t)((cnt_of(s, d) * esz + RCCL_PIECE - 1) / RCCL_PIECE));
    std::vector<uint64_t> roff(G);                                 // where source s lands in dst (bytes)
    for (int s = 0; s < G; s++) { roff[s] = o * esz; o += cnt_of(s, r); }
    for (size_t k = 0; k < pieces; k++) {
        // check and registration are one step: abort_comms lets every registered shard finish posting (or times out on
        // one that hangs inside RCCL, which is what ncclCommAbort is for) before the handles go
        { std::lock_guard<std::mutex> lk(m->rv.m); if (m->comm_dead) return -1; m->posting++; }
        ncclResult_t e = m->rccl.GroupStart();
        for (int d = 0; d < G && e == ncclSuccess; d++) {
            const size_t bytes = cnt_of(r, d) * esz, a = std::min(bytes, k * RCCL_PIECE), b = std::min(bytes, (k + 1) * RCCL_PIECE);
            if (b > a) e = m->rccl.Send((const char *)ptr_of(r) + off_of(r, d) * esz + a, b - a, ncclUint8, d, m->comm[r], m->stream[r]);
        }
        for (int s = 0; s < G && e == ncclSuccess; s++) {
            const size_t bytes = cnt_of(s, r) * esz, a = std::min(bytes, k * RCCL_PIECE), b = std::min(bytes, (k + 1) * RCCL_PIECE);
            if (b > a) e = m->rccl.Recv((char *)dst + roff[s] + a, b - a, ncclUint8, s, m->comm[r], m->stream[r]);
        }
        const ncclResult_t e2 = m->rccl.GroupEnd();
        { std::lock_guard<std::mutex> lk(m->rv.m); m->posting--; }
        m->posted.notify_all();
        if (e != ncclSuccess || e2 != ncclSuccess) {
            std::lock_guard<std::mutex> lk(m->rv.m);
            if (m->err.empty()) m->err = std::string("RCCL exchange: ") + m->rccl.GetErrorString(e != ncclSuccess ? e : e2);
            return -1;
        }
    }
    if (hipStreamSynchronize(m->stream[r]) != hipSuccess) return -1;
    return (long long)o;
}

const char *const STEP_NAMES[DEBWT_MULTI_STEPS] = {
    "shard_histogram", "shard_plan", "kmer_sort_rle", "shard_partition_keys", "exchange_keys", "shard_sort_range", "shard_sort_end",
    "shard_classify_local", "shard_facts_export", "exchange_facts", "shard_classify_global", "shard_sp_flags", "shard_sp_emit",
    "exchange_sp", "shard_sp_import", "shard_blue_route", "exchange_blue", "shard_blue_place", "blue_sort", "bwt_assemble",
    "shard_export", "exchange_rows", "concat_rows", "waiting"};
enum { S_HIST, S_PLAN, S_RESCAN, S_PARTITION, S_XKEYS, S_SORT_RANGE, S_SORT_END, S_CLASSIFY_LOCAL, S_FACTS_EXPORT, S_XFACTS,
       S_CLASSIFY_GLOBAL, S_SP_FLAGS, S_SP_EMIT, S_XSP, S_SP_IMPORT, S_BLUE_ROUTE, S_XBLUE, S_BLUE_PLACE, S_BLUE_SORT, S_ASSEMBLE,
       S_EXPORT, S_XROWS, S_CONCAT, S_WAITING };
enum { X_KEYS, X_FACTS, X_SP, X_BLUE, X_ROWS };

// wall time of one step of one shard: from here to the end of the scope, with the device drained at both ends in serial
// mode (there the shard is alone on its GPU, so the time is the step's own)
struct StepTimer {
    debwt_multi *m; int r, step;
    std::chrono::steady_clock::time_point t0;
    StepTimer(debwt_multi *m_, int r_, int step_) : m(m_), r(r_), step(step_), t0(std::chrono::steady_clock::now()) {}
    ~StepTimer() {
        if (m->rv.serial) (void)hipDeviceSynchronize();
        m->rep[r].ms[step] += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
};

// an RCCL exchange that failed on one shard: the peers sit in their groups, or will, waiting for this one -- end every
// communicator so that their streams come back (with an error), and make the next RCCL build start from new ones
// (the handles stay in m->comm until the build's threads have joined: a shard that has not reached its group yet finds
// comm_dead set and posts nothing; debwt_multi_build drops the handles afterwards)
void abort_comms(debwt_multi *m) {
    {
        std::unique_lock<std::mutex> lk(m->rv.m);
        if (m->comm_dead) return;
        m->comm_dead = true;                   // nobody registers from here on
        m->posted.wait_for(lk, std::chrono::seconds(5), [&] { return m->posting == 0; });
    }
    for (ncclComm_t cm : m->comm) if (cm) (void)m->rccl.CommAbort(cm);
}

void shard_thread(debwt_multi *m, int r) {
    const int G = m->G;
    debwt_ctx *c = m->ctx[r];
    Rendezvous &rv = m->rv;
    (void)hipSetDevice(m->dev[r]);
    const bool over_rccl = m->exchange_backend == DEBWT_EXCHANGE_RCCL;
    debwt_shard_report &rep = m->rep[r];
    int rc = 0;
#define STEP(id, call, what) do { StepTimer t_(m, r, id); rc = (call); if (rc) set_err(m, r, what); } while (0)
    // the barrier between two stages; in serial mode the shards then take turns again.  (The time a shard spends here is
    // its wait for the slowest shard: reported, since it is what imbalance costs.)
    auto barrier = [&](int rc_) {
        const auto t0 = std::chrono::steady_clock::now();
        const bool ok = rv.sync(rc_);
        if (ok) rv.enter(r);
        if (!rv.serial) rep.ms[S_WAITING] += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return ok;
    };
    // An RCCL exchange is a collective: either every shard posts its sends and receives or none may -- a shard that
    // skipped its group after a local error (out of memory, a failed step) would leave the others waiting in theirs for
    // ever.  So over RCCL the shards agree on rc before every exchange; peer copies need no such gate (a pull that is
    // skipped is missed by nobody).  On an error INSIDE the group all communicators are ended (abort_comms).
#define EXCHANGE(id, xid, dst, esz, ptr_of, off_of, cnt_of) do { \
        if (over_rccl && !barrier(rc)) return; \
        if (!rc) { \
            StepTimer t_(m, r, id); \
            uint64_t in_ = 0, out_ = 0; \
            for (int s_ = 0; s_ < G; s_++) if (s_ != r) { in_ += (cnt_of)(s_, r); out_ += (cnt_of)(r, s_); } \
            rep.bytes_in[xid] += in_ * (esz); rep.bytes_out[xid] += out_ * (esz); \
            if (xchg(m, r, dst, esz, ptr_of, off_of, cnt_of) < 0) { rc = DEBWT_EDEVICE; if (over_rccl) abort_comms(m); } \
        } \
    } while (0)

    rv.enter(r);
    // 1. census of the slices -> splitters over the shards (instance counts balanced on the summed census, the
    //    reference's segCount idea, src/mySort.c:104-110), key ranges inside every shard
    rc = debwt_shard_begin(c, r, G);
    if (rc) set_err(m, r, "shard_begin");
    if (!rc) STEP(S_HIST, debwt_shard_histogram(c, m->hist[r].data()), "shard_histogram");
    if (!barrier(rc)) return;
    std::vector<uint64_t> total(BINS, 0), cum(BINS + 1, 0);
    for (int s = 0; s < G; s++) for (int b = 0; b < BINS; b++) total[b] += m->hist[s][b];
    for (int b = 0; b < BINS; b++) cum[b + 1] = cum[b] + total[b];
    std::vector<uint32_t> bins(G + 1, BINS);
    bins[0] = 0;
    for (int s = 1; s < G; s++) {
        const uint64_t target = cum[BINS] * (uint64_t)s / (uint64_t)G;
        uint32_t b = (uint32_t)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
        bins[s] = std::min<uint32_t>(std::max(b, bins[s - 1]), BINS);
    }
    rep.bin_lo = bins[r]; rep.bin_hi = bins[r + 1];
    rep.keys = cum[bins[r + 1]] - cum[bins[r]];
    uint32_t nr = 0;
    const uint64_t held = m->xa[r].cap + m->xb[r].cap + m->facts[r].cap + m->allfacts[r].cap + m->sp[r].cap + m->allsp[r].cap +
                          m->part[r].cap + (r == 0 ? m->parts.cap + m->out.cap : 0);
    const int keys = m->key_mode >= 0 ? m->key_mode : debwt_shard_key_mode(m->n, G, 0.0, nullptr, nullptr);
    const bool exchange = keys == DEBWT_KEYS_EXCHANGE;
    STEP(S_PLAN, debwt_shard_plan(c, total.data(), bins[r], bins[r + 1], cum[bins[r]], exchange ? 1 : 2, held, &nr), "shard_plan");
    std::vector<uint64_t> mk(MAXR, 0);
    m->cuts[r].assign(MAXR + 1, 0);
    if (!rc) { rc = debwt_shard_ranges(c, m->cuts[r].data(), mk.data(), MAXR); if (rc) set_err(m, r, "shard_ranges"); }
    m->cuts[r].resize(rc ? 1 : nr + 1);
    rep.key_ranges = rc ? 0 : nr;
    if (!barrier(rc)) return;
    size_t rounds = 0;
    for (int s = 0; s < G; s++) rounds = std::max(rounds, m->cuts[s].size() - 1);
    if (r == 0) { m->st.rounds = (uint32_t)rounds; m->st.key_mode = (uint32_t)keys; m->st.exchange_backend = (uint32_t)m->exchange_backend; }

    // 2. the keys of the shard's ranges: read from the shard's own copy of the text (key rescan) ...
    if (!exchange) STEP(S_RESCAN, debwt_kmer_sort_rle(c), "kmer_sort_rle");
    // ... or the k-mer bucket exchange, one round per key range
    if (exchange) { rc = debwt_shard_sort_begin(c); if (rc) set_err(m, r, "shard_sort_begin"); }
    if (!barrier(rc)) return;
    for (size_t t = 0; exchange && t < rounds; t++) {
        std::vector<uint8_t> tab(BINS, 0xFF);
        for (int s = 0; s < G; s++)
            if (t + 1 < m->cuts[s].size())
                for (uint32_t b = m->cuts[s][t]; b < m->cuts[s][t + 1]; b++) tab[b] = (uint8_t)s;
        uint64_t ns = 0;
        for (int b = 0; b < BINS; b++) if (tab[b] != 0xFF) ns += m->hist[r][b];
        rc = m->xa[r].ensure((ns + 64) * 8) ? 0 : DEBWT_ENOMEM;
        if (!rc) STEP(S_PARTITION, debwt_shard_partition_keys(c, tab.data(), (uint64_t *)m->xa[r].p, m->xa[r].cap / 8, m->offs[r].data()),
                      "shard_partition_keys");
        if (!barrier(rc)) return;
        uint64_t nrecv = 0;
        for (int s = 0; s < G; s++) nrecv += m->offs[s][r + 1] - m->offs[s][r];
        rc = m->xb[r].ensure((nrecv + 64) * 8) ? 0 : DEBWT_ENOMEM;
        EXCHANGE(S_XKEYS, X_KEYS, m->xb[r].p, 8, [&](int s) { return m->xa[s].p; }, [&](int s, int d) { return m->offs[s][d]; },
                 [&](int s, int d) { return m->offs[s][d + 1] - m->offs[s][d]; });
        if (r == 0) { uint64_t moved = 0; for (int s = 1; s < G; s++) moved += m->offs[s][1] - m->offs[s][0]; m->st.key_bytes_in += moved * 8; }
        if (!barrier(rc)) return;                                   // every pull is done: the send buffers are free again
        if (t + 1 < m->cuts[r].size())
            STEP(S_SORT_RANGE, debwt_shard_sort_range(c, (uint32_t)t, (uint64_t *)m->xb[r].p, nrecv), "shard_sort_range");
        if (!barrier(rc)) return;
    }
    if (exchange) STEP(S_SORT_END, debwt_shard_sort_end(c), "shard_sort_end");

    // 3. local classification totals, red table from everybody's facts
    if (!rc) STEP(S_CLASSIFY_LOCAL, debwt_shard_classify_local(c, &m->nfacts[r], &m->nblocks[r], &m->brows[r]), "shard_classify_local");
    if (!rc) rc = m->facts[r].ensure((m->nfacts[r] + 1) * 8) ? 0 : DEBWT_ENOMEM;
    if (!rc) STEP(S_FACTS_EXPORT, debwt_shard_facts_export(c, (uint64_t *)m->facts[r].p, m->facts[r].cap / 8), "shard_facts_export");
    rep.blocks = m->nblocks[r]; rep.blue_rows = m->brows[r];
    if (!barrier(rc)) return;
    uint64_t allf = 0, qbase = 0, btotal = 0;
    std::vector<uint32_t> first_block(G + 1, 0);
    for (int s = 0; s < G; s++) {
        allf += m->nfacts[s]; btotal += m->brows[s];
        if (s < r) qbase += m->nblocks[s];
        first_block[s + 1] = first_block[s] + (uint32_t)m->nblocks[s];
    }
    rc = m->allfacts[r].ensure((allf + 1) * 8) ? 0 : DEBWT_ENOMEM;
    EXCHANGE(S_XFACTS, X_FACTS, m->allfacts[r].p, 8, [&](int s) { return m->facts[s].p; }, [&](int, int) { return (uint64_t)0; },
             [&](int s, int) { return m->nfacts[s]; });
    if (!rc) STEP(S_CLASSIFY_GLOBAL, debwt_shard_classify_global(c, (const uint64_t *)m->allfacts[r].p, allf, qbase, btotal), "shard_classify_global");

    // 4. SP code of the slices, symbols gathered everywhere
    uint64_t bloc = 0;
    if (!rc) STEP(S_SP_FLAGS, debwt_shard_sp_flags(c, &m->slen[r], &bloc), "shard_sp_flags");
    if (!barrier(rc)) return;
    uint64_t sp_off = 0, sp_total = 0;
    for (int s = 0; s < G; s++) { if (s < r) sp_off += m->slen[s]; sp_total += m->slen[s]; }
    rc = m->sp[r].ensure(m->slen[r] + 64) ? 0 : DEBWT_ENOMEM;
    if (!rc) STEP(S_SP_EMIT, debwt_shard_sp_emit(c, sp_off, (uint8_t *)m->sp[r].p, m->sp[r].cap), "shard_sp_emit");
    if (!barrier(rc)) return;
    rc = m->allsp[r].ensure(sp_total + 64) ? 0 : DEBWT_ENOMEM;
    EXCHANGE(S_XSP, X_SP, m->allsp[r].p, 1, [&](int s) { return m->sp[s].p; }, [&](int, int) { return (uint64_t)0; },
             [&](int s, int) { return m->slen[s]; });
    if (!rc) STEP(S_SP_IMPORT, debwt_shard_sp_import(c, (const uint8_t *)m->allsp[r].p, sp_total), "shard_sp_import");

    // 5. blue entries of the slice -> the owners of their blocks.  The send and the receive buffer are the shard's own key
    //    buffers where the keys were read off the text (free since the sort; debwt_shard_scratch), else allocations
    void *bsend = nullptr, *brcv = nullptr;
    uint64_t bsend_bytes = 0, brcv_bytes = 0;
    if (!rc) { rc = debwt_shard_scratch(c, DEBWT_SCRATCH_SEND, &bsend, &bsend_bytes); if (rc) set_err(m, r, "shard_scratch"); }
    if (!rc && bsend_bytes < (bloc + 64) * 8) {
        rc = m->xa[r].ensure((bloc + 64) * 8) ? 0 : DEBWT_ENOMEM;
        bsend = m->xa[r].p; bsend_bytes = m->xa[r].cap;
    }
    m->bsend[r] = bsend;
    if (!rc) STEP(S_BLUE_ROUTE, debwt_shard_blue_route(c, first_block.data(), (uint64_t *)bsend, bsend_bytes / 8, m->boffs[r].data()),
                  "shard_blue_route");
    if (!barrier(rc)) return;
    uint64_t brecv = 0;
    for (int s = 0; s < G; s++) brecv += m->boffs[s][r + 1] - m->boffs[s][r];
    rc = debwt_shard_scratch(c, DEBWT_SCRATCH_RECV, &brcv, &brcv_bytes);       // (the routed entries it held are in the send buffer now)
    if (!rc && brcv_bytes < (brecv + 64) * 8) {
        rc = m->xb[r].ensure((brecv + 64) * 8) ? 0 : DEBWT_ENOMEM;
        brcv = m->xb[r].p;
    }
    EXCHANGE(S_XBLUE, X_BLUE, brcv, 8, [&](int s) { return m->bsend[s]; }, [&](int s, int d) { return m->boffs[s][d]; },
             [&](int s, int d) { return m->boffs[s][d + 1] - m->boffs[s][d]; });
    if (r == 0) { uint64_t moved = 0; for (int s = 1; s < G; s++) moved += m->boffs[s][1] - m->boffs[s][0]; m->st.blue_bytes_in = moved * 8; }
    if (!barrier(rc)) return;
    STEP(S_BLUE_PLACE, debwt_shard_blue_place(c, (uint64_t *)brcv, brecv), "shard_blue_place");

    // 6. owned blocks and rows
    if (!rc) STEP(S_BLUE_SORT, debwt_blue_sort(c), "blue_sort");
    if (!rc) STEP(S_ASSEMBLE, debwt_bwt_assemble(c), "bwt_assemble");
    if (!rc) { rc = debwt_shard_info(c, &m->rowbase[r], &m->rows[r], &m->nhash[r]); if (rc) set_err(m, r, "shard_info"); }
    rep.rows = m->rows[r];
    if (!barrier(rc)) return;

    // 7. final concat on the first GPU: the packed row ranges are pulled there and shift-merged by row offset
    uint64_t maxw = 0;
    for (int s = 0; s < G; s++) maxw = std::max<uint64_t>(maxw, (m->rows[s] + 31) / 32 + 1);
    rc = m->part[r].ensure(maxw * 8) ? 0 : DEBWT_ENOMEM;
    if (!rc) STEP(S_EXPORT, debwt_shard_export(c, (uint64_t *)m->part[r].p, maxw), "shard_export");
    m->hrows[r].assign(std::max<uint64_t>(m->nhash[r], 1), 0);
    if (!rc) { rc = debwt_shard_fetch(c, nullptr, m->hrows[r].data(), &m->drow[r]); if (rc) set_err(m, r, "shard_fetch"); }
    m->hrows[r].resize(m->nhash[r]);
    if (!barrier(rc)) return;
    if (r == 0) rc = (m->parts.ensure(maxw * G * 8) && m->out.ensure(((m->n + 31) / 32 + 1) * 8)) ? 0 : DEBWT_ENOMEM;
    if (r == 0 || over_rccl)                                       // (a gather: everybody sends, the first GPU receives)
        EXCHANGE(S_XROWS, X_ROWS, r == 0 ? m->parts.p : nullptr, 8, [&](int s) { return m->part[s].p; }, [&](int, int) { return (uint64_t)0; },
                 [&](int, int d) { return d == 0 ? maxw : (uint64_t)0; });
    if (r == 0) {
        std::vector<uint64_t> poff(G), pbase(G), prows(G);
        for (int s = 0; s < G; s++) { poff[s] = maxw * s; pbase[s] = m->rowbase[s]; prows[s] = m->rows[s]; }
        if (!rc) STEP(S_CONCAT, debwt_concat_rows(c, (const uint64_t *)m->parts.p, (uint32_t)G, poff.data(), pbase.data(), prows.data(), m->n,
                                        (uint64_t *)m->out.p), "concat_rows");
        m->hash_rows.clear();
        m->dollar_row = ~0ull;
        for (int s = 0; s < G; s++) {
            m->hash_rows.insert(m->hash_rows.end(), m->hrows[s].begin(), m->hrows[s].end());
            if (m->drow[s] != ~0ull) m->dollar_row = m->drow[s];
        }
        std::sort(m->hash_rows.begin(), m->hash_rows.end());
        if (!rc && (m->hash_rows.size() != m->nrec - 1 || m->dollar_row == ~0ull)) {
            rc = DEBWT_EINTERNAL;
            std::lock_guard<std::mutex> lk(rv.m);
            if (m->err.empty()) m->err = "the shards' '#' / '$' rows do not add up";
        }
    }
    (void)rv.sync(rc);
#undef STEP
#undef EXCHANGE
}

}  // namespace

extern "C" int debwt_multi_create(const debwt_config *cfg, const int *devices, int ngpus, debwt_multi **out) {
    if (!cfg || !out || ngpus < 1 || ngpus > 255) return DEBWT_EINVAL;
    debwt_multi *m = new (std::nothrow) debwt_multi();
    if (!m) return DEBWT_ENOMEM;
    m->G = ngpus; m->cfg = *cfg;
    m->ctx.assign(ngpus, nullptr); m->dev.resize(ngpus); m->stream.assign(ngpus, nullptr);
    for (auto *v : {&m->xa, &m->xb, &m->facts, &m->allfacts, &m->sp, &m->allsp, &m->part}) v->resize(ngpus);
    m->hist.assign(ngpus, std::vector<uint64_t>(BINS));
    m->offs.assign(ngpus, std::vector<uint64_t>(ngpus + 1));
    m->boffs.assign(ngpus, std::vector<uint64_t>(ngpus + 1));
    m->cuts.assign(ngpus, {});
    for (auto *v : {&m->nfacts, &m->nblocks, &m->brows, &m->slen, &m->rowbase, &m->rows, &m->nhash, &m->drow}) v->assign(ngpus, 0);
    m->hrows.assign(ngpus, {});
    m->bsend.assign(ngpus, nullptr);
    m->rep.assign(ngpus, debwt_shard_report{});
    m->rv.n = ngpus;
    int rc = DEBWT_OK;
    for (int r = 0; r < ngpus && !rc; r++) {
        m->dev[r] = devices ? devices[r] : r;
        debwt_config c1 = *cfg;
        c1.device = m->dev[r];
        rc = debwt_create(&c1, &m->ctx[r]);
        if (rc) break;
        for (auto *v : {&m->xa, &m->xb, &m->facts, &m->allfacts, &m->sp, &m->allsp, &m->part}) (*v)[r].dev = m->dev[r];
        if (hipSetDevice(m->dev[r]) != hipSuccess || hipStreamCreateWithFlags(&m->stream[r], hipStreamNonBlocking) != hipSuccess)
            rc = DEBWT_EDEVICE;
    }
    m->parts.dev = m->out.dev = m->dev[0];
    if (!rc)                                                      // direct peer access where the GPUs differ (xGMI)
        for (int a = 0; a < ngpus; a++)
            for (int b = 0; b < ngpus; b++)
                if (m->dev[a] != m->dev[b]) {
                    int can = 0;
                    (void)hipSetDevice(m->dev[a]);
                    if (hipDeviceCanAccessPeer(&can, m->dev[a], m->dev[b]) == hipSuccess && can)
                        (void)hipDeviceEnablePeerAccess(m->dev[b], 0);   // "already enabled" is fine
                }
    (void)hipGetLastError();
    if (rc) { debwt_multi_destroy(m); return rc; }
    *out = m;
    return DEBWT_OK;
}

extern "C" void debwt_multi_destroy(debwt_multi *m) {
    if (!m) return;
    for (int r = 0; r < m->G; r++) {
        for (auto *v : {&m->xa, &m->xb, &m->facts, &m->allfacts, &m->sp, &m->allsp, &m->part}) (*v)[r].release();
        if (m->stream[r]) { (void)hipSetDevice(m->dev[r]); (void)hipStreamDestroy(m->stream[r]); }
        if (m->ctx[r]) debwt_destroy(m->ctx[r]);
    }
    m->parts.release(); m->out.release();
    if (!m->comm_dead) for (ncclComm_t cm : m->comm) if (cm && m->rccl.ready) (void)m->rccl.CommDestroy(cm);
    if (m->own.words) debwt_free_packed(&m->own);
    delete m;
}

extern "C" const char *debwt_multi_last_error(const debwt_multi *m) { return m ? m->err.c_str() : ""; }
extern "C" debwt_ctx *debwt_multi_shard(debwt_multi *m, int shard) { return m && shard >= 0 && shard < m->G ? m->ctx[shard] : nullptr; }

extern "C" int debwt_multi_load_text(debwt_multi *m, const uint64_t *packed, uint64_t n, const uint64_t *sep, uint64_t nrec) {
    if (!m) return DEBWT_EINVAL;
    m->built = false; m->err.clear();
    for (int r = 0; r < m->G; r++) {                              // the same text into the HBM of every GPU
        int rc = debwt_load_text(m->ctx[r], packed, n, sep, nrec);
        if (rc) { m->err = std::string("load on GPU ") + std::to_string(m->dev[r]) + ": " + debwt_last_error(m->ctx[r]); return rc; }
    }
    m->n = n; m->nrec = nrec;
    return DEBWT_OK;
}

extern "C" int debwt_multi_load_fasta(debwt_multi *m, const char *path, int threads, unsigned flags, uint64_t seed) {
    if (!m || !path) return DEBWT_EINVAL;
    debwt_packed_text pt;
    char msg[256] = "";
    int rc = debwt_pack_fasta_opts(path, threads, flags, seed, &pt, msg, sizeof msg);
    if (rc) { m->err = msg; return rc; }
    rc = debwt_multi_load_text(m, pt.words, pt.n, pt.sep, pt.nrec);
    if (m->own.words) debwt_free_packed(&m->own);
    m->own = pt;                                                  // the contexts read the host text during a build
    return rc;
}

extern "C" int debwt_multi_set_key_mode(debwt_multi *m, int key_mode) {
    if (!m || key_mode < -1 || key_mode > DEBWT_KEYS_RESCAN) return DEBWT_EINVAL;
    m->key_mode = key_mode;
    return DEBWT_OK;
}

extern "C" int debwt_multi_set_exchange(debwt_multi *m, int backend) {
    if (!m || (backend != DEBWT_EXCHANGE_PEER_COPY && backend != DEBWT_EXCHANGE_RCCL)) return DEBWT_EINVAL;
    if (backend == DEBWT_EXCHANGE_RCCL) {
        for (int a = 0; a < m->G; a++)
            for (int b = a + 1; b < m->G; b++)
                if (m->dev[a] == m->dev[b]) { m->err = "RCCL needs one GPU per shard (a device ordinal repeats)"; return DEBWT_EINVAL; }
        if (!m->rccl.load(&m->err)) return DEBWT_EDEVICE;
        if (m->comm.empty()) {
            m->comm.assign(m->G, nullptr);
            const ncclResult_t e = m->rccl.CommInitAll(m->comm.data(), m->G, m->dev.data());
            if (e != ncclSuccess) { m->comm.clear(); m->err = std::string("ncclCommInitAll: ") + m->rccl.GetErrorString(e); return DEBWT_EDEVICE; }
        }
    }
    m->exchange_backend = backend;
    return DEBWT_OK;
}

extern "C" int debwt_multi_build(debwt_multi *m) {
    if (!m || !m->n) return DEBWT_ESTATE;
    m->err.clear(); m->built = false;
    if (m->rv.serial && m->exchange_backend == DEBWT_EXCHANGE_RCCL) {
        m->err = "serial mode takes the shards one at a time: an RCCL exchange needs all of them at once (use peer copies)";
        return DEBWT_EINVAL;
    }
    if (m->exchange_backend == DEBWT_EXCHANGE_RCCL && m->comm.empty()) {
        // the communicators of the last build were aborted after a failed exchange: new ones, or a clear error -- never a
        // silent change of the backend the caller configured
        m->comm.assign(m->G, nullptr);
        const ncclResult_t e = m->rccl.CommInitAll(m->comm.data(), m->G, m->dev.data());
        if (e != ncclSuccess) {
            m->comm.clear();
            m->err = std::string("ncclCommInitAll after an aborted exchange: ") + m->rccl.GetErrorString(e) +
                     " (debwt_multi_set_exchange(m, DEBWT_EXCHANGE_PEER_COPY) builds without RCCL)";
            return DEBWT_EDEVICE;
        }
    }
    m->rv.failed = 0; m->rv.waiting = 0; m->rv.turn = 0;
    m->st = debwt_multi_stats{};
    m->rep.assign(m->G, debwt_shard_report{});
    m->st.ngpus = (uint32_t)m->G;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int r = 1; r < m->G; r++) th.emplace_back(shard_thread, m, r);
    shard_thread(m, 0);
    for (auto &t : th) t.join();
    m->st.ms_build = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (m->comm_dead) {      // aborted communicators are gone; the configured backend stays: the next build makes new ones (above)
        m->comm.clear(); m->comm_dead = false;
    }
    if (m->rv.failed) return m->rv.failed;
    m->built = true;
    return DEBWT_OK;
}

extern "C" int debwt_multi_fetch_bwt(debwt_multi *m, uint64_t *bwt, uint64_t *hash_rows, uint64_t *dollar_row) {
    if (!m || !bwt || !dollar_row || (m->nrec > 1 && !hash_rows)) return DEBWT_EINVAL;
    if (!m->built) return DEBWT_ESTATE;
    (void)hipSetDevice(m->dev[0]);
    if (hipMemcpy(bwt, m->out.p, (size_t)((m->n + 31) >> 5) * 8, hipMemcpyDeviceToHost) != hipSuccess) return DEBWT_EDEVICE;
    if (m->nrec > 1) memcpy(hash_rows, m->hash_rows.data(), (m->nrec - 1) * 8);
    *dollar_row = m->dollar_row;
    return DEBWT_OK;
}

extern "C" int debwt_multi_get_stats(const debwt_multi *m, debwt_multi_stats *out, debwt_stats *shard0) {
    if (!m || !out) return DEBWT_EINVAL;
    *out = m->st;
    out->n = m->n; out->nrec = m->nrec;
    if (shard0) return debwt_get_stats(m->ctx[0], shard0);
    return DEBWT_OK;
}

extern "C" int debwt_multi_set_serial(debwt_multi *m, int serial) {
    if (!m) return DEBWT_EINVAL;
    m->rv.serial = serial != 0;
    return DEBWT_OK;
}

extern "C" int debwt_multi_get_shard_report(const debwt_multi *m, int shard, debwt_shard_report *out) {
    if (!m || !out || shard < 0 || shard >= m->G) return DEBWT_EINVAL;
    *out = m->rep[shard];
    return DEBWT_OK;
}

extern "C" const char *debwt_multi_step_name(int step) { return step >= 0 && step < DEBWT_MULTI_STEPS ? STEP_NAMES[step] : ""; }

extern "C" int debwt_multi_verify(debwt_multi *m, debwt_verify_report *rep) {
    // inverse BWT of the concatenated result on the first GPU (it holds the text like every GPU)
    if (!m || !rep) return DEBWT_EINVAL;
    if (!m->built) return DEBWT_ESTATE;
    return debwt_verify_device(m->ctx[0], (const uint64_t *)m->out.p, m->hash_rows.data(), m->dollar_row, 0, rep);
}
