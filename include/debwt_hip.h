/*
 * debwt_hip.h -- C ABI of libdebwt_hip.so, the MI355X (gfx950) implementation of deBWT's
 * de Bruijn branch-encode suffix sort and BWT assembly path.
 *
 * The reference has no plugin/FFI interface: its stages are C functions that pass state through
 * ~30 globals and temp files and exit(1) on error (/root/reference/src/main.h:1-8,
 * src/main.c:79-160).  Each entry point below replaces one of those stage calls; the comment on
 * each names the call it replaces.  Conventions that differ from the reference on purpose:
 *   - an opaque context instead of globals; explicit caller-owned buffers with sizes;
 *   - every function returns int: 0 = ok, negative = DEBWT_E* (never exits, never prints);
 *   - no temp files, no Jellyfish: k-mers are enumerated from the packed text on the GPU;
 *   - functions are not re-entrant on one context; one host thread drives one context/GPU.
 *
 * Data formats are the reference's: text 2 bits/base A0 C1 G2 T3, 32 bases per
 * uint64_t, base j at bit 2*(31-(j&31)) of word j>>5, 'T' stored at every separator and 32 'T'
 * of padding after the end (src/collect#$.c:61-90); BWT output in the same packing with '#'/'$'
 * rows stored as 3 and listed separately (src/insertCase3.c:75-97,115-131).
 */
#ifndef DEBWT_HIP_H
#define DEBWT_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DEBWT_OK 0
#define DEBWT_EINVAL (-1)   /* bad argument (k outside 12..32, n too small, NULL pointer, ...) */
#define DEBWT_ENOMEM (-2)   /* device or host allocation failed */
#define DEBWT_EDEVICE (-3)  /* HIP runtime error (see debwt_last_error) */
#define DEBWT_ESTATE (-4)   /* stage called out of order */
#define DEBWT_ERANGE (-5)   /* input exceeds a capacity of this build (see debwt_last_error) */
#define DEBWT_EINTERNAL (-6)/* consistency check failed */
#define DEBWT_EIO (-7)      /* a file could not be created or written (debwt_dump_reference_files) */

typedef struct debwt_ctx debwt_ctx;

typedef struct {
    int k;          /* edge length, KMER_LENGTH_PlusOne, 12..32 (src/main.c:41-47); node = k-1 */
    int device;     /* HIP device ordinal */
    int sort_algo;  /* 0 = default (3); 1 = all digits by LSD passes in HBM; 3 = top digits in HBM, rest in LDS */
    int reserved;
} debwt_config;

/* counters of one run; the first block mirrors the reference's globals so that runs can be
 * compared stage by stage (src/generateSP.c:28-31, src/collect#$.c:59) */
typedef struct {
    uint64_t n;                 /* BWTLEN */
    uint64_t nrec;              /* countRead */
    uint64_t red_capacity;      /* redCapacity */
    uint64_t blue_capacity;     /* blueCapacity */
    uint64_t blue_bound_num;    /* blueBoundNum */
    uint64_t case3num;          /* case3num */
    uint64_t sp_len;            /* SP symbols (spCodeLen - 32) */
    uint64_t special_branch_num;/* specialBranchNum */
    uint64_t n_main;            /* node instances sorted (n - nrec*(k-1)) */
    uint64_t distinct_keys;     /* distinct (node,pred) keys incl. record-start instances */
    uint64_t blue_large_blocks; /* blocks sorted by the global-memory path */
    uint64_t blue_max_block;
    /* device time of the last run, milliseconds (hipEvents on the context's stream) */
    float ms_extract, ms_sort, ms_classify, ms_sp, ms_blue, ms_assemble, ms_total, ms_host_special;
    /* dominant kernel (one radix scatter pass): launches and total ms in the last run */
    uint32_t radix_pass_launches;
    float radix_pass_ms;
    uint64_t radix_pass_keys;   /* keys moved per launch */
    /* how the special-region tables of the last run were built (collect, src/collect#$.c:118-157,348-602):
     * special_path 0 = one host thread, 1 = host threads, 2 = device; special_threads = host threads used */
    uint32_t special_path, special_threads;
    /* bucket finish of the key sort (all key ranges of the last run): stretches above a wave tile (1024 keys), those of
     * them the classifying kernel left to the 4096-key network, stretches above 4096 keys (all-HBM passes) */
    uint64_t sort_unfit_stretches, sort_unfit_network, sort_over_stretches;
} debwt_stats;

int debwt_create(const debwt_config *cfg, debwt_ctx **out);
void debwt_destroy(debwt_ctx *ctx);
const char *debwt_strerror(int code);
const char *debwt_last_error(const debwt_ctx *ctx);   /* text of the last HIP/internal failure */
int debwt_get_config(const debwt_ctx *ctx, debwt_config *out);   /* the configuration the context was created with */

/* Replaces `collect`'s text hand-over (src/collect#$.c:61-90,100-113: files `reference`,
 * `specialSA`).  packed: ceil((n+32)/32) words in the format above (host memory, must stay valid
 * until the context is destroyed or the next load); sep: the nrec separator positions ascending,
 * sep[nrec-1] == n-1 ('$').  Copies the text to HBM and sizes the workspace. */
int debwt_load_text(debwt_ctx *ctx, const uint64_t *packed, uint64_t n, const uint64_t *sep, uint64_t nrec);

/* Convenience for hosts that hold ASCII: records concatenated without separators, upper or
 * lower case ACGT only (src/main.c:18-23), every record > 32 bases (src/collect#$.c:41-45).
 * Packs on the host, then behaves like debwt_load_text (the packed copy is owned by ctx). */
int debwt_load_ascii(debwt_ctx *ctx, const char *seq, const uint64_t *reclen, uint64_t nrec);

/* FASTA ingest (replaces the reference's single-threaded kseq.h + zlib reader, src/collect#$.c:34-90): the file is
 * mapped (gzip: inflated), parsed and packed by `threads` host threads (the reference's -t) into the text format
 * above.  FASTQ (first byte '@') is read as the reference's reader reads it (src/kseq.h:177-201: sequence lines up to a
 * line that starts with '+', '@' or '>'; after '+' as many quality characters as bases, which are dropped; records
 * without a quality section and '>' records may be mixed in).  Characters other than ACGTacgt and white
 * space, sequence before the first header, a quality string of another length than its sequence and records of 32 bases
 * or fewer (src/collect#$.c:41-45) are errors.  debwt_pack_fasta is host-only (no GPU needed);
 * debwt_load_fasta = debwt_pack_fasta + debwt_load_text with the packed copy owned by ctx. */
typedef struct {
    uint64_t *words;      /* ((n + 63) >> 5) + 2 words */
    uint64_t nwords;
    uint64_t n;           /* BWTLEN */
    uint64_t *sep;        /* nrec separator positions, sep[nrec-1] == n-1 */
    uint64_t nrec;
    double seconds_read, seconds_pack;
} debwt_packed_text;
int debwt_pack_fasta(const char *path, int threads, debwt_packed_text *out, char *errbuf, size_t errlen);
void debwt_free_packed(debwt_packed_text *p);
int debwt_load_fasta(debwt_ctx *ctx, const char *path, int threads);
/* The same with options.  DEBWT_FASTA_IUPAC_RANDOM: N and the other IUPAC ambiguity letters are replaced by one of the
 * bases they stand for -- what the reference leaves to otherTool/transferN.c (:8-32 tables, :57-60 draw with rand()),
 * made deterministic: the draw is a hash of `seed` and the base's text position, independent of the thread count. */
#define DEBWT_FASTA_IUPAC_RANDOM 1u
int debwt_pack_fasta_opts(const char *path, int threads, unsigned flags, uint64_t seed, debwt_packed_text *out,
                          char *errbuf, size_t errlen);
int debwt_load_fasta_opts(debwt_ctx *ctx, const char *path, int threads, unsigned flags, uint64_t seed);
/* An upper bound of the text length (symbols incl. separators) the file can hold, from its size and framing alone -- a plain
 * file: its bytes; block gzip: the members' ISIZE added up; one gzip member below 1 GB: its ISIZE -- or 0 when that cannot be
 * told (several plain members, a large member).  For the n of a debwt_reserve that runs beside the parse; a text that turns out
 * longer only makes the buffers grow at the load.  (The reference knows its text's length only after reading it,
 * src/collect#$.c:52-59.) */
uint64_t debwt_fasta_text_bound(const char *path);
/* The ingest of a gzip file gives up buffers as large as the text (inflated text, pieces); the library releases them on a thread
 * of its own once the text is packed -- returning memory costs this host's kernel 50 ms per GB, and it holds the address space's
 * lock meanwhile.  Between debwt_host_release_hold(1) and (0) nothing is released: a host that copies the packed text to the
 * device right behind the parse (debwt_load_text from pageable memory) brackets both.  Calls nest. */
void debwt_host_release_hold(int on);

/* Device memory for a text of up to n symbols in nrec records, allocated before the text is there: a one-shot host (the
 * reference is one: src/main.c:16-173) calls this on a helper thread while it still reads and packs its input, so that the
 * seconds the driver takes to hand out a few hundred GB (it clears what another process released) pass behind the
 * ingest instead of inside the first build.  branching: expected fraction of positions that are branching nodes, sizes the
 * data-dependent buffers (<= 0: 0.12); everything grows later if the text needs more.  Optional: without it the same
 * buffers are allocated by debwt_load_text and the stages.  Not re-entrant with other calls on the context; DEBWT_ESTATE
 * on a context that already holds a text (its buffers would be re-allocated under it).
 * DEBWT_RESERVE_ONE_SHOT: the context will build once.  When the first buffers arrive at the driver's clearing rate (the
 * memory was another process's a moment ago) the text is cut into more, smaller key ranges than a context that is reused
 * would take -- less workspace to wait for, a few more passes over the text.  That choice STAYS with the context for every
 * later build, exactly like a debwt_set_range_cap (which also undoes it). */
#define DEBWT_RESERVE_ONE_SHOT 1u
/* DEBWT_RESERVE_COMPACT: key ranges of 2^29 instances at most (up to 16 ranges), whatever the first buffers cost: a process
 * that runs right behind another one is handed memory the driver is still clearing, and waits for ALL of the clearing as soon as
 * one of its buffers lies in it -- a third of the workspace mostly comes from memory that is clean.  3.1 Gbp: 28 GiB of device
 * memory instead of 102, 8 % more time per build: cli/deBWT right behind another run 0.9 s instead of 4.3 s.  Texts above 20 Gbp
 * are left to ONE_SHOT's rule (their text-sized buffers alone fill half the device).  cli/deBWT asks for it; like ONE_SHOT the
 * choice stays with the context. */
#define DEBWT_RESERVE_COMPACT 2u
int debwt_reserve(debwt_ctx *ctx, uint64_t n, uint64_t nrec, double branching, unsigned flags);

/* Texts whose node instances (one 8-byte key per base) do not fit HBM at once, or number 2^32 or more, are built
 * in key ranges: prefix ranges of the k-mer space holding at most `max_instances` keys each, sorted and classified
 * one after the other over the resident 2-bit text -- the single-GPU form of SURVEY 8e's bucket sharding (the
 * reference's analogue is its per-thread bucket segments, src/mySort.c:98-110).  Default 2^31; tests lower it to
 * drive the multi-range path on small inputs.  The result does not depend on it. */
int debwt_set_range_cap(debwt_ctx *ctx, uint64_t max_instances);

/* ---- stage entry points, to be called in this order after a load ------------------------------ */
/* Capacities of this build (the reference has none: uint64_t throughout, src/collect#$.h) -- each is a 32-bit index somewhere,
 * each answers DEBWT_ERANGE with the reason in debwt_last_error, none has a slower fall-back:
 *   debwt_kmer_sort_rle / debwt_shard_plan  one 12-mer prefix bin (the unit key ranges are cut at) holds 2^32 - 2^20 node
 *                        instances or more: a key range indexes its keys with 32 bits.  30 Gbp of ten genomes: the fullest
 *                        bin holds 0.1 G; a text of > 4 G copies of one 12-mer (a 4 Gbp homopolymer) would hit it.
 *   debwt_sp_generate    more than 2^31 branching nodes (slot numbers of the node table are 32-bit: 2^32 slots = 64 GB;
 *                        k = 16 on 3.1 Gbp has 0.7 G); a sharded build: a text slice with 2^32 multi-in positions or more
 *                        (debwt_shard_blue_route; ten genomes with an Alu-like family at N = 1: use more shards or the
 *                        one-GPU build, which buckets its 5.6 G entries by key range), or a routed entry whose block id
 *                        and SP index do not fit 61 bits (debwt_shard_sp_emit).
 *   debwt_shard_begin    world > 255 (the owner table holds bytes).
 * DEBWT_ENOMEM is not a capacity: the stages release what the other stages left (debwt_hip.hip, reclaim) before they give up. */

/* Replaces kmercounting.sh + mySort (src/main.c:70,83; src/mySort.c:26-201) and getKmer
 * (src/getKmer.c:12-49): enumerates every node instance of the text as a key
 * (node << 2 | predecessor base) -- the edge list grouped by its (k-1)-suffix, i.e. getKmer's
 * multiIn{A,C,G,T} view -- radix-sorts the keys and run-length encodes them. */
int debwt_kmer_sort_rle(debwt_ctx *ctx);
/* Replaces generateBlocks/mergeKmer (src/INandOut.c:13-89,159-943) plus the special-region
 * tables of collect (src/collect#$.c:118-157,348-602): node flags, red table, block bounds,
 * case-2 characters, rows of the special suffixes. */
int debwt_classify(debwt_ctx *ctx);
/* Replaces generateSP (src/generateSP.c:19-272): SP code and blue entries. */
int debwt_sp_generate(debwt_ctx *ctx);
/* Replaces sortBlue (src/sortBlue.c:10-57). */
int debwt_blue_sort(debwt_ctx *ctx);
/* Replaces insertCase3 up to the file writes (src/insertCase3.c:13-104). */
int debwt_bwt_assemble(debwt_ctx *ctx);

/* All five stages back to back (src/main.c:83-149). */
int debwt_build(debwt_ctx *ctx);

/* Copies the result to host memory: bwt ceil(n/32) words, hash_rows nrec-1 rows ascending,
 * dollar_row 1 row -- the contents of OUT, OUT.#, OUT.$ (src/insertCase3.c:115-131). */
int debwt_fetch_bwt(debwt_ctx *ctx, uint64_t *bwt, uint64_t *hash_rows, uint64_t *dollar_row);
/* debwt_build + debwt_fetch_bwt in one call, with the copy hidden behind the build: the reference writes OUT only after
 * its last stage (src/insertCase3.c:115-131); here a text that is built in several key ranges has the rows of a range
 * assembled as soon as the range's blocks are sorted, and they travel to `bwt` (page-locked memory for the overlap,
 * debwt_pinned_alloc) while the blocks of the next range are sorted.  Same buffers and contents as debwt_fetch_bwt. */
int debwt_build_to_host(debwt_ctx *ctx, uint64_t *bwt, uint64_t *hash_rows, uint64_t *dollar_row);
/* Only the row lists (OUT.#, OUT.$) -- for callers that keep the BWT words in HBM. */
int debwt_fetch_rows(debwt_ctx *ctx, uint64_t *hash_rows, uint64_t *dollar_row);
/* Device address of the packed BWT words of the last run (for callers that keep it in HBM). */
int debwt_bwt_device_ptr(debwt_ctx *ctx, const uint64_t **d_words);

/* Symbol census of the result in HBM: counts[c] = rows of the packed BWT that hold code c ('#' and '$' rows are
 * stored as 3, src/insertCase3.c:86-97) -- a BWT is a permutation of its text, so the census must equal the text's. */
int debwt_bwt_census(debwt_ctx *ctx, uint64_t counts[4]);

int debwt_get_stats(const debwt_ctx *ctx, debwt_stats *out);

/* Page-locked host memory for the text handed to debwt_load_text and the buffers debwt_fetch_bwt fills: the copies
 * then run at link rate and asynchronously (the reference keeps both in ordinary heap memory, src/collect#$.c:61,
 * src/insertCase3.c:115-119; pageable buffers work here too, only slower). */
int debwt_pinned_alloc(size_t bytes, void **out);
void debwt_pinned_free(void *p);

/* ---- one build over several GPUs: k-mer-prefix shards ----------------------------------
 * The reference has no distributed path; these entry points extend the stage sequence above for the case
 * that one text is built by `world` contexts (one per GPU, each holding the whole 2-bit text).  Shard r sorts
 * and classifies the keys of one prefix range -- in several key ranges one after the other when they do not
 * fit HBM at once -- and owns the BWT rows and the multi-in blocks of those nodes.
 * The collectives (bracketed) are the caller's (debwt_amd/sharded.py uses torch.distributed: RCCL or gloo).
 *
 * The keys reach their shard in one of two ways (debwt_shard_key_mode picks; everything after the sort is the same):
 * Key exchange (every shard scans only its 1/world slice of the text and ships 8-byte keys):
 *   load -> shard_begin -> shard_histogram -> [all-gather of the slice censuses] -> shard_plan(exchange = 1) ->
 *   shard_ranges -> [all-gather of the range cuts] -> shard_sort_begin ->
 *   per exchange round t:  shard_partition_keys (keys of the slice that fall into round t's ranges, grouped by owner)
 *                          -> [alltoallv of 8-byte keys: the k-mer bucket exchange] -> shard_sort_range(t) ->
 *   shard_sort_end -> ...
 * Key rescan (every shard holds the text anyway: it reads all of it once per key range and keeps its own keys --
 * n/4 bytes from its HBM instead of 8 n / world bytes from the fabric):
 *   ... -> shard_plan(exchange = 2) -> kmer_sort_rle -> ...
 * Then, sliced again (per-GPU work O(n / world)):
 *   shard_classify_local -> shard_facts_export -> [all-gather of the fact lists] ->
 *   shard_classify_global -> shard_sp_flags -> [all-gather of the slice SP lengths] -> shard_sp_emit ->
 *   [all-gather of the slice SP symbols] -> shard_sp_import -> shard_blue_route -> [alltoallv of 8-byte blue entries]
 *   -> shard_blue_place -> blue_sort -> bwt_assemble -> shard_info -> [gather of the packed row ranges] -> concat_rows.
 * Scan mode (no bulk exchange at all: tests and 2-GPU hosts without peer access):
 *   ... shard_plan(exchange = 0) or shard_set_range -> kmer_sort_rle -> shard_classify_local -> ... ->
 *   shard_classify_global -> sp_generate -> blue_sort -> bwt_assemble. */
#define DEBWT_KEYS_EXCHANGE 0
#define DEBWT_KEYS_RESCAN 1
/* Cost model of the two key paths for n text positions on `world` GPUs that each hold the text; link_gbytes_per_s:
 * sustained rate of one GPU-to-GPU link in one direction (<= 0: 48, i.e. 5/8 of an xGMI link's 76.8 GB/s).
 * Returns DEBWT_KEYS_EXCHANGE or DEBWT_KEYS_RESCAN, the estimated per-GPU milliseconds in *exchange_ms, *rescan_ms
 * (either may be NULL).  The constants are measured ones, see DESIGN.md section 7. */
int debwt_shard_key_mode(uint64_t n, int world, double link_gbytes_per_s, double *exchange_ms, double *rescan_ms);
int debwt_shard_begin(debwt_ctx *ctx, int rank, int world);           /* world <= 255 */
/* counts of this shard's slice of text positions by the top 12 bits of their key: 4096 words (host) */
int debwt_shard_histogram(debwt_ctx *ctx, uint64_t *hist4096);
/* this shard takes the keys whose top 12 bits lie in [bin_lo, bin_hi) as ONE key range: m_keys of them (from the
 * summed histogram), m_base keys in the shards before it */
int debwt_shard_set_range(debwt_ctx *ctx, uint32_t bin_lo, uint32_t bin_hi, uint64_t m_keys, uint64_t m_base);
/* the same from the summed census hist4096 (host), cut into as many key ranges as the free HBM (or debwt_set_range_cap)
 * asks for: the reference's segCount balancing (src/mySort.c:104-110) applied twice, over GPUs and over rounds.
 * exchange: 1 = the keys of every range will arrive through debwt_shard_sort_range; 0 = the shard reads them from
 * its text (debwt_kmer_sort_rle); 2 = the same with a sliced SP stage whose blue-entry exchange runs in the key buffers
 * (debwt_shard_scratch).  caller_held_bytes: device memory the caller already holds for
 * the exchanges of this build (counted as available: it is reused). */
int debwt_shard_plan(debwt_ctx *ctx, const uint64_t *hist4096, uint32_t bin_lo, uint32_t bin_hi, uint64_t m_base,
                     int exchange, uint64_t caller_held_bytes, uint32_t *nranges);
/* the cuts of debwt_shard_plan: range i = bins [bin_bounds[i], bin_bounds[i+1]) with m_keys[i] keys */
int debwt_shard_ranges(debwt_ctx *ctx, uint32_t *bin_bounds, uint64_t *m_keys, uint32_t capacity);
/* exchange mode: shard_of_bin: 4096 bytes (host), owner of each 12-bit prefix bin in this round, 0xFF = the bin is
 * not exchanged in this round; d_out: DEVICE buffer of `capacity` words; offs: world+1 host words,
 * group i = d_out[offs[i] .. offs[i+1]). */
int debwt_shard_partition_keys(debwt_ctx *ctx, const uint8_t *shard_of_bin, uint64_t *d_out, uint64_t capacity,
                               uint64_t *offs);
int debwt_shard_sort_begin(debwt_ctx *ctx);
/* d_keys: the `count` keys of range `range` in a DEVICE buffer the caller owns; it is used as sort workspace and must
 * stay untouched until the next shard_sort_range / shard_sort_end */
int debwt_shard_sort_range(debwt_ctx *ctx, uint32_t range, uint64_t *d_keys, uint64_t count);
int debwt_shard_sort_end(debwt_ctx *ctx);
int debwt_shard_classify_local(debwt_ctx *ctx, uint64_t *nfacts, uint64_t *nblocks, uint64_t *blue_rows);
/* copies the shard's nfacts fact words (node<<2 | 1 multi-out, | 2 multi-in) to a DEVICE buffer */
int debwt_shard_facts_export(debwt_ctx *ctx, uint64_t *d_dst, uint64_t capacity);
/* d_facts: DEVICE buffer with the facts of all shards (any order); qbase: blocks owned by the shards before
 * this one; blue_total: sum of blue_rows over all shards */
int debwt_shard_classify_global(debwt_ctx *ctx, const uint64_t *d_facts, uint64_t nfacts, uint64_t qbase,
                                uint64_t blue_total);
int debwt_shard_sp_flags(debwt_ctx *ctx, uint64_t *sp_symbols, uint64_t *mi_positions);
int debwt_shard_sp_emit(debwt_ctx *ctx, uint64_t sp_offset, uint8_t *d_dst, uint64_t capacity);
int debwt_shard_sp_import(debwt_ctx *ctx, const uint8_t *d_src, uint64_t sp_total);
/* first_block_of_shard: world+1 host words (exclusive scan of the shards' block counts) */
int debwt_shard_blue_route(debwt_ctx *ctx, const uint32_t *first_block_of_shard, uint64_t *d_out, uint64_t capacity,
                           uint64_t *offs);
/* Exchange buffers without an allocation: after the sort stage of a shard whose keys were read off the text (plan mode 0
 * or 2) its two key buffers are free until the next build.  DEBWT_SCRATCH_SEND: the buffer to route into
 * (debwt_shard_blue_route's d_out; the peers read it during the exchange); DEBWT_SCRATCH_RECV: the buffer to receive into
 * (debwt_shard_blue_place's d_entries) -- it holds the routed entries of the slice until debwt_shard_blue_route has run,
 * so it may be written only after that call.  *bytes = 0: none (keys exchanged: the receive buffer is the caller's own).
 * A host uses a scratch buffer when it is large enough and its own allocation otherwise; debwt_shard_plan mode 2 leaves
 * no room for exchange buffers beside the key buffers. */
#define DEBWT_SCRATCH_SEND 0
#define DEBWT_SCRATCH_RECV 1
int debwt_shard_scratch(debwt_ctx *ctx, int which, void **d_ptr, uint64_t *bytes);
/* d_entries: the `count` routed entries this shard received (DEVICE); the buffer is used as scratch and holds nothing
 * meaningful afterwards */
int debwt_shard_blue_place(debwt_ctx *ctx, uint64_t *d_entries, uint64_t count);
/* Final concat on one GPU: d_parts holds the shards' packed row ranges (part i from word part_word_off[i], one spare
 * word behind each, rows [row_base[i], row_base[i] + rows[i])); the ranges are not 32-row aligned and are
 * shift-merged into d_out (ceil(n/32) DEVICE words), as src/generateSP.c:379-405 joins SP segments. */
int debwt_concat_rows(debwt_ctx *ctx, const uint64_t *d_parts, uint32_t nparts, const uint64_t *part_word_off,
                      const uint64_t *row_base, const uint64_t *rows, uint64_t n, uint64_t *d_out);

/* first global row of the shard, its row count, and (after assemble) its number of '#' rows */
int debwt_shard_info(debwt_ctx *ctx, uint64_t *row_base, uint64_t *rows, uint64_t *hash_rows);
/* shard result to host: ceil(rows/32) words packed from the shard's first row (words may be NULL: row lists only);
 * GLOBAL '#' rows; the GLOBAL '$' row or ~0 when it is not in this shard */
int debwt_shard_fetch(debwt_ctx *ctx, uint64_t *words, uint64_t *hash_rows, uint64_t *dollar_row);

/* the shard's packed rows into a DEVICE buffer of `capacity` words (zero-filled behind the last row) */
int debwt_shard_export(debwt_ctx *ctx, uint64_t *d_words, uint64_t capacity);
/* debwt_bwt_census over any packed rows in HBM (n rows at d_words) */
int debwt_census_words(debwt_ctx *ctx, const uint64_t *d_words, uint64_t n, uint64_t counts[4]);

/* ---- the same build from ONE host process: one host thread per GPU, exchanges as peer-to-peer copies over xGMI --------
 * What the reference's single process with its thread pool (src/main.c:30) becomes on a node of GPUs: debwt_multi_build
 * runs the sharded stage sequence above (keys exchanged or rescanned) on `ngpus` contexts (devices[i] = HIP ordinal of shard i; NULL = 0, 1, ...;
 * ordinals may repeat -- several shards on one GPU -- which is how the path is tested on a one-GPU box), every shard
 * pulling its keys / facts / SP symbols / blue entries out of the other shards' buffers with device-to-device copies,
 * and leaves the concatenated BWT in the HBM of the first GPU.  cli/deBWT --gpus G is the C host on top of it. */
typedef struct debwt_multi debwt_multi;
typedef struct {
    uint64_t n, nrec;
    uint32_t ngpus, rounds;          /* exchange rounds = key ranges of the busiest shard */
    uint64_t key_bytes_in;           /* bytes of k-mers shard 0 pulled from the other shards (all rounds) */
    uint64_t blue_bytes_in;          /* bytes of blue entries shard 0 pulled from the other shards */
    float ms_build;                  /* wall time of debwt_multi_build */
    uint32_t key_mode;               /* DEBWT_KEYS_EXCHANGE or DEBWT_KEYS_RESCAN: how the keys reached their shards */
    uint32_t exchange_backend;       /* DEBWT_EXCHANGE_PEER_COPY or DEBWT_EXCHANGE_RCCL: what moved the data between the shards */
} debwt_multi_stats;
int debwt_multi_create(const debwt_config *cfg, const int *devices, int ngpus, debwt_multi **out);   /* cfg->device unused */
void debwt_multi_destroy(debwt_multi *m);
const char *debwt_multi_last_error(const debwt_multi *m);
debwt_ctx *debwt_multi_shard(debwt_multi *m, int shard);      /* the context of one shard (debwt_set_range_cap, stats) */
int debwt_multi_load_text(debwt_multi *m, const uint64_t *packed, uint64_t n, const uint64_t *sep, uint64_t nrec);
int debwt_multi_load_fasta(debwt_multi *m, const char *path, int threads, unsigned flags, uint64_t seed);
/* DEBWT_KEYS_EXCHANGE / DEBWT_KEYS_RESCAN, or -1 (the default): debwt_shard_key_mode decides at every build */
int debwt_multi_set_key_mode(debwt_multi *m, int key_mode);
/* What carries the exchanges between the shards (k-mer buckets, facts, SP symbols, blue entries, final row ranges):
 * DEBWT_EXCHANGE_PEER_COPY (default): every shard pulls its segments with device-to-device copies;
 * DEBWT_EXCHANGE_RCCL: one grouped ncclSend / ncclRecv alltoallv per exchange, one communicator per GPU of this process
 * (ncclCommInitAll; RCCL is loaded on demand -- DEBWT_EDEVICE where it is missing -- and needs one distinct GPU per shard).
 * The configured backend stays until this call changes it: when an RCCL exchange fails inside a build, every communicator
 * is aborted (the build returns the error), and the NEXT debwt_multi_build makes new communicators first -- or returns
 * DEBWT_EDEVICE with the reason in debwt_multi_last_error; it never falls back to peer copies by itself. */
#define DEBWT_EXCHANGE_PEER_COPY 0
#define DEBWT_EXCHANGE_RCCL 1
int debwt_multi_set_exchange(debwt_multi *m, int backend);
int debwt_multi_build(debwt_multi *m);
int debwt_multi_fetch_bwt(debwt_multi *m, uint64_t *bwt, uint64_t *hash_rows, uint64_t *dollar_row);
int debwt_multi_get_stats(const debwt_multi *m, debwt_multi_stats *out, debwt_stats *shard0);
/* What one shard of the last debwt_multi_build did, step by step -- the evidence for load balance over the shards (the
 * reference balances its per-thread segments by instance counts, src/mySort.c:104-110; whether that balances the later
 * stages too is a measurement).  ms[i]: wall milliseconds of step i (debwt_multi_step_name(i): the debwt_shard_* calls of
 * the sequence above, the exchanges, and "waiting" = time at the barriers between the stages, i.e. what the slowest
 * shard costs this one); bytes_in / bytes_out: what the shard received from / sent to OTHER shards in each exchange
 * (0 keys, 1 facts, 2 SP symbols, 3 blue entries, 4 final rows).
 * debwt_multi_set_serial(m, 1): the shards of a build take turns between the barriers, one on its GPU at a time, device
 * drained around every step -- the steps' times are then each shard's own even when several shards share one GPU (the
 * way N = 2, 4, 8 are measured on a one-GPU box; peer-copy exchanges only).  The result of the build is the same. */
#define DEBWT_MULTI_STEPS 24
typedef struct {
    float ms[DEBWT_MULTI_STEPS];
    uint64_t bytes_in[5], bytes_out[5];
    uint64_t keys, key_ranges, blocks, blue_rows, rows;   /* node instances, key ranges, multi-in blocks, their rows, BWT rows */
    uint32_t bin_lo, bin_hi;                              /* the shard's 12-bit prefix bins [bin_lo, bin_hi) */
} debwt_shard_report;
int debwt_multi_set_serial(debwt_multi *m, int serial);
int debwt_multi_get_shard_report(const debwt_multi *m, int shard, debwt_shard_report *out);
const char *debwt_multi_step_name(int step);

/* ---- intermediates, for stage-by-stage parity ---------------------------------- */
typedef enum {
    DEBWT_ARR_SORTED_KEYS = 1, /* u64 x n_main: (node<<2|pred) ascending; after debwt_kmer_sort_rle of a one-range
                                  build driven stage by stage on one GPU only (debwt_build and shards keep the
                                  run-length encoding alone: DEBWT_ESTATE, as in a build of several key ranges)    */
    DEBWT_ARR_DISTINCT_KEYS,   /* u64 x distinct_keys                                                 */
    DEBWT_ARR_RED,             /* u64 x red_capacity: node<<2 | multiin<<1 | multiout, ascending      */
    DEBWT_ARR_SP_SYMBOLS,      /* u8  x sp_len: SP symbols 0..5                                       */
    DEBWT_ARR_BLUE,            /* u64 x blue_capacity: pred | spIndex<<4 (src/generateSP.c:666-672)   */
    DEBWT_ARR_BLUE_BOUND,      /* u64 x blue_bound_num: inclusive end of each block (blueBound)       */
    DEBWT_ARR_CASE3_BOUND,     /* u64 x case3num: [first row,last row] per block (case3bound)         */
    DEBWT_ARR_ROW_SYMBOLS      /* u8  x n: BWT symbols 0..5 by row (after assemble)                   */
} debwt_array;
/* Copies up to `capacity` elements; *count receives the element count of the array. */
int debwt_fetch_array(debwt_ctx *ctx, debwt_array which, void *dst, uint64_t capacity, uint64_t *count);

/* The same intermediates written as files in the byte formats of the reference's own temp files and global arrays, so
 * that a build can be bisected stage by stage against the reference (cli/deBWT --dump DIR drives this; the names are the
 * reference's, `oracle/_ref/ref_driver` in the build container writes the same set as OUT.<name>):
 *   DEBWT_DUMP_KMERINFO  any time after the load (runs debwt_kmer_count_sorted: the pipeline is back at "text loaded"):
 *                        kmerInfo = D x {u64 k-mer left-aligned, u64 count} (src/mySort.c:193-195);
 *   DEBWT_DUMP_BLOCKS    after debwt_classify: redSeq, redPoint (src/INandOut.c:396-405), blueBound (:359-361),
 *                        case3bound (:347-353), raw u64 arrays;
 *   DEBWT_DUMP_SP        after debwt_sp_generate, before debwt_blue_sort: spCode (ceil(S / 32) words, 2 bits per symbol,
 *                        separators as 3; src/generateSP.c:626-660), spSpecialIndex (N x u64, :630-641), blueTable
 *                        (B x u64 pred | spIndex << 4, :666-672; the order INSIDE a block is the arrival order of the
 *                        scan -- thread-dependent in the reference too -- only the set per block is defined).
 * One-range builds driven stage by stage on one GPU (as debwt_fetch_array).  DEBWT_EIO: dir missing or not writable. */
#define DEBWT_DUMP_KMERINFO 1
#define DEBWT_DUMP_BLOCKS 2
#define DEBWT_DUMP_SP 3
int debwt_dump_reference_files(debwt_ctx *ctx, const char *dir, int stage);

/* ---- primitives exposed for measurement and parity ------------------------------------------- */

/* Stand-alone a-1+a-2 in the reference's own output format: every k-mer inside a record,
 * sorted ascending, left-aligned, with its count -- the contents of `kmerInfo`
 * (src/mySort.c:193-195).  Needs a loaded text.  kmers/counts: host arrays of `capacity`
 * entries; *distinct receives D. */
int debwt_kmer_count_sorted(debwt_ctx *ctx, uint64_t *kmers, uint64_t *counts, uint64_t capacity,
                            uint64_t *distinct);

/* LSD radix sort of `count` 64-bit keys resident in HBM (d_keys, d_tmp: device pointers, both
 * `count` words; result in d_keys).  key_bits: significant low bits (1..64).
 * ms_per_pass (optional) receives the mean device time of one scatter pass. */
int debwt_radix_sort_u64(debwt_ctx *ctx, uint64_t *d_keys, uint64_t *d_tmp, uint64_t count, int key_bits,
                         float *ms_per_pass);

/* Host-only: checksums of the special-region tables (`collect`'s specialSA / specialBwt / specialBranch / head / tail
 * tables, src/collect#$.c:118-157,348-602) that debwt_kmer_sort_rle builds on host threads for the loaded text layout;
 * digest[0..3] = order of the special suffixes, their keys + BWT symbols, the special branches, the head/tail nodes.
 * Lets tests compare the threaded module with its single-threaded run (DEBWT_SPECIAL_THREADS / DEBWT_SPECIAL_PAR_MIN). */
int debwt_special_digest(const uint64_t *packed, uint64_t n, const uint64_t *sep, uint64_t nrec, int k, uint64_t digest[4]);

/* Needs a GPU and a loaded text: builds the same tables on the device (the path collections of many records take,
 * SURVEY 8f-1) and by the host module, and counts the elements that differ: mismatch[0..5] = suffix order of the
 * special suffixes, their keys, their BWT symbols, the special branches, the head nodes, the tail nodes.  A build in
 * progress is discarded (the context is back at "text loaded"); the counters of the last build are left alone. */
int debwt_special_compare(debwt_ctx *ctx, uint64_t mismatch[6]);

/* Verification tool standing in for the dead LFsearch path (src/LFsearch.c:14-48): inverse BWT
 * by LF walk on the host from a fetched result; writes the n symbols (0..5).  Returns 0 when the
 * walk closes. */
int debwt_verify_inverse(const uint64_t *bwt, uint64_t n, const uint64_t *hash_rows, uint64_t nrec,
                         uint64_t dollar_row, uint8_t *sym_out);

/* The same check on the device, for results that never leave HBM and for sizes where one chain of n dependent LF steps
 * (the reference's walk, src/LFsearch.c:49-166, ~0.2 us per step on a host core) takes hours: a sampled rank structure
 * over the packed rows (the reference's occ tables, src/insertCase3.c:141-194, as one 128-byte line per 384 rows), the
 * rows of ~`segments` text positions found by backward search of the text in the BWT, and one LF walk per segment, each
 * compared symbol by symbol with the loaded text and required to end on the row the previous segment starts from --
 * together one chain over all n rows.  d_words: DEVICE packed rows (NULL: the context's own result, with its row
 * lists); hash_rows / dollar_row: OUT.# / OUT.$ (host).  segments 0 = default.  ok = 1: the inverse BWT of the rows is
 * the loaded text. */
typedef struct {
    uint64_t segments, steps, mismatches, broken_links, search_failures, search_steps;
    float ms_index, ms_search, ms_walk;
    int ok;
} debwt_verify_report;
int debwt_verify_device(debwt_ctx *ctx, const uint64_t *d_words, const uint64_t *hash_rows, uint64_t dollar_row,
                        uint64_t segments, debwt_verify_report *report);
/* the same on the concatenated result of debwt_multi_build (first GPU; every GPU holds the text) */
int debwt_multi_verify(debwt_multi *m, debwt_verify_report *report);

#ifdef __cplusplus
}
#endif
#endif
