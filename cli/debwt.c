/*
 * debwt.c -- host program with the reference's command line over libdebwt_hip.so.
 *
 *   deBWT -o OUT [-t T] [-k K] [-j DIR] [--device D | --gpus G [--devices a,b,...] [--keys auto|exchange|rescan] [--exchange peer|rccl]] [--iupac SEED]
 *         [--verify] [--dump DIR] INPUT.fa[.gz]
 *
 * Same contract as /root/reference/src/main.c:25-53,175-186: options are `flag value` pairs, INPUT last;
 * -k 12..32 (default 32); -t (default 8) = host threads of the FASTA ingest; -j accepted and ignored (no Jellyfish); OUT is probed by create+remove before any work
 * (src/main.c:55-58); exit status 0 on success, 1 with a message on stderr otherwise.  Output files OUT,
 * OUT.#, OUT.$ are those of src/insertCase3.c:115-131.
 * --verify (the only flag without a value) runs the job of the reference's unreachable developer mode (src/LFsearch.c:14-48 with
 * the occ tables of src/insertCase3.c:139-208) on the device after the build: the inverse BWT of the rows must be the input
 * text; exit status 1 when it is not.  --dump DIR writes the intermediates of every stage into DIR under the reference's file
 * names and in its byte formats (kmerInfo src/mySort.c:193-195; redSeq, redPoint, blueBound, case3bound
 * src/INandOut.c:347-366,396-417; spCode, spSpecialIndex, blueTable src/generateSP.c:626-672) and drives the stages one by one
 * as src/main.c:83-149 does.
 */
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "../include/debwt_hip.h"

static void usage(void) {
    fprintf(stderr, "usage:\n"
                    "deBWT [options] reference\n"
                    "Please make sure your sequence don't contain any uncertain characters like 'N'\n"
                    "options:\n"
                    "-o: output bwt file(binary)\n"
                    "-t (optional): maximum thread number(default 8; host threads of the FASTA ingest)\n"
                    "-k (optional): k-mer length (from 12 to 32, default 32)\n"
                    "-j (optional): jellyfish directory (accepted, ignored)\n"
                    "--device (optional): GPU ordinal (default 0)\n"
                    "--gpus (optional): build with G GPUs (k-mer-prefix shards, one host thread per GPU, exchanges over xGMI)\n"
                    "--devices (optional): comma-separated GPU ordinals of the G shards (default 0,1,...; may repeat)\n"
                    "--keys (optional): with --gpus, how the k-mers reach their shard: exchange (alltoallv of the keys), rescan (every\n"
                    "       GPU reads its own copy of the text), auto (default: the cheaper one by the library's cost model)\n"
                    "--exchange (optional): with --gpus, what carries the exchanges between the GPUs: peer (device-to-device copies,\n"
                    "       default) or rccl (grouped ncclSend / ncclRecv over xGMI; one distinct GPU per shard)\n"
                    "--iupac (optional): seed; N and other ambiguity letters become pseudo-random bases of their sets\n"
                    "                    (what otherTool/transferN does, reproducibly)\n"
                    "--verify (optional, takes no value): after the build, the inverse BWT is walked on the GPU and compared with the\n"
                    "       input (the reference's LFsearch developer mode); exit status 1 on a mismatch\n"
                    "--dump (optional): directory; kmerInfo, redSeq, redPoint, blueBound, case3bound, spCode, spSpecialIndex and\n"
                    "       blueTable are written there in the reference's formats (one GPU, texts sorted in one key range)\n"
                    "reference: sequence in fasta format (plain or gzip)\n");
}

double now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

/* a large file (the 7.5 GB of rows of a 30 Gbp collection) is written by several threads, each its slice at its offset */
struct write_job { int fd; const char *p; size_t off, bytes; int rc; };
static void *write_main(void *arg) {
    struct write_job *j = arg;
    while (j->bytes) {
        ssize_t w = pwrite(j->fd, j->p + j->off, j->bytes > ((size_t)1 << 30) ? (size_t)1 << 30 : j->bytes, (off_t)j->off);
        if (w <= 0) { j->rc = -1; return NULL; }
        j->off += (size_t)w; j->bytes -= (size_t)w;
    }
    return NULL;
}
static int write_file(const char *path, const void *p, size_t bytes) {
    enum { WT = 8 };
    if (bytes < ((size_t)256 << 20)) {
        FILE *f = fopen(path, "wb");
        if (!f) { fprintf(stderr, "cannot create %s!\n", path); return -1; }
        size_t w = fwrite(p, 1, bytes, f);
        return (fclose(f) == 0 && w == bytes) ? 0 : -1;
    }
    int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { fprintf(stderr, "cannot create %s!\n", path); return -1; }
    struct write_job job[WT];
    pthread_t th[WT];
    int started[WT], rc = 0;
    const size_t per = ((bytes + WT - 1) / WT + 4095) & ~(size_t)4095;
    for (int i = 0; i < WT; i++) {
        const size_t a = per * (size_t)i < bytes ? per * (size_t)i : bytes, b = a + per < bytes ? a + per : bytes;
        job[i] = (struct write_job){fd, p, a, b - a, 0};
        started[i] = b > a && pthread_create(&th[i], NULL, write_main, &job[i]) == 0;
        if (b > a && !started[i]) write_main(&job[i]);                     /* no thread: this one writes the slice */
    }
    for (int i = 0; i < WT; i++) { if (started[i]) pthread_join(th[i], NULL); if (job[i].rc) rc = -1; }
    if (close(fd)) rc = -1;
    return rc;
}

/* the outputs of src/insertCase3.c:115-131 */
static int write_outputs(const char *obj, const uint64_t *bwt, uint64_t n, const uint64_t *hash_rows, uint64_t nrec, uint64_t dollar) {
    size_t ol = strlen(obj);
    char *p = malloc(ol + 3);
    if (!p) return -1;
    int rc = write_file(obj, bwt, (size_t)((n + 31) >> 5) * 8);
    memcpy(p, obj, ol); memcpy(p + ol, ".#", 3);
    if (!rc) rc = write_file(p, hash_rows, (nrec - 1) * 8);
    memcpy(p + ol, ".$", 3);
    if (!rc) rc = write_file(p, &dollar, 8);
    free(p);
    return rc;
}

/* A one-shot program pays for its device memory (the driver clears what another process released, ~33 GiB/s): the helper
 * thread allocates the workspace (debwt_reserve) and the page-locked output buffers while the main thread still reads and
 * packs the FASTA file -- the file size bounds the text length. */
struct reserve_job { debwt_ctx *ctx; uint64_t n_bound; uint64_t *bwt; int rc; double seconds; unsigned flags; };
double now(void);
static void *reserve_main(void *arg) {
    struct reserve_job *j = arg;
    double t0 = now();
    j->rc = debwt_reserve(j->ctx, j->n_bound, 1, 0.0, j->flags);
    void *p = NULL;
    if (debwt_pinned_alloc((size_t)((j->n_bound + 31) >> 5) * 8 + 64, &p) == DEBWT_OK) j->bwt = p;
    j->seconds = now() - t0;
    return NULL;
}

/* --verify: one summary line; 0 when the inverse BWT of the rows is the text */
static int report_verify(int rc, const debwt_verify_report *r, const char *errtext) {
    if (rc) { fprintf(stderr, "verify: %s %s\n", debwt_strerror(rc), errtext); return 1; }
    printf("verify: inverse BWT %s: %lu LF steps in %lu segments, %lu mismatches, %lu broken links, %lu search failures "
           "(index %.1f ms, search %.1f ms, walk %.1f ms)\n", r->ok ? "ok" : "FAILED", (unsigned long)r->steps,
           (unsigned long)r->segments, (unsigned long)r->mismatches, (unsigned long)r->broken_links,
           (unsigned long)r->search_failures, r->ms_index, r->ms_search, r->ms_walk);
    if (!r->ok) fprintf(stderr, "verify: the inverse BWT of the result is NOT the input text\n");
    return r->ok ? 0 : 1;
}

/* --gpus G: the same program over G GPUs (debwt_multi_*: one host thread per GPU inside the library) */
static int multi_main(const char *source, const char *obj, int k, int threads, int iupac, unsigned long long seed, int gpus,
                      const int *devs, int key_mode, int exchange, int verify) {
    double t0 = now();
    debwt_config cfg = {k, 0, 0, 0};
    debwt_multi *m = NULL;
    int rc = debwt_multi_create(&cfg, devs, gpus, &m);
    if (rc) { fprintf(stderr, "debwt_multi_create (%d GPUs): %s\n", gpus, debwt_strerror(rc)); return 1; }
    debwt_multi_set_key_mode(m, key_mode);
    if (exchange != DEBWT_EXCHANGE_PEER_COPY && (rc = debwt_multi_set_exchange(m, exchange))) {
        fprintf(stderr, "--exchange rccl: %s %s\n", debwt_strerror(rc), debwt_multi_last_error(m));
        debwt_multi_destroy(m);
        return 1;
    }
    double t1 = now();
    rc = debwt_multi_load_fasta(m, source, threads, iupac ? DEBWT_FASTA_IUPAC_RANDOM : 0u, seed);
    if (rc) {
        fprintf(stderr, "%s (sequence must be ACGT only unless --iupac is given, records > 32 bases)\n", debwt_multi_last_error(m));
        debwt_multi_destroy(m);
        return 1;
    }
    double t2 = now();
    rc = debwt_multi_build(m);
    if (rc) { fprintf(stderr, "build: %s %s\n", debwt_strerror(rc), debwt_multi_last_error(m)); debwt_multi_destroy(m); return 1; }
    double t3 = now();
    debwt_multi_stats st;
    debwt_stats s0;
    debwt_multi_get_stats(m, &st, &s0);
    uint64_t *bwt = malloc((size_t)((st.n + 31) >> 5) * 8), *hash_rows = malloc((st.nrec ? st.nrec : 1) * 8), dollar = 0;
    rc = (bwt && hash_rows) ? debwt_multi_fetch_bwt(m, bwt, hash_rows, &dollar) : DEBWT_ENOMEM;
    if (!rc && write_outputs(obj, bwt, st.n, hash_rows, st.nrec, dollar)) rc = DEBWT_EINVAL;
    double t4 = now();
    if (!rc) {
        printf("BWTLEN=%lu\n", (unsigned long)st.n);
        printf("%u GPUs, exchanges by %s, keys %s, %u key round(s): init %.3f s, read+pack+load (%d threads) %.3f s, build %.3f s, fetch+write %.3f s; "
               "shard 0 received %.3f GB of k-mers and %.3f GB of blue entries\n", st.ngpus,
               st.exchange_backend == DEBWT_EXCHANGE_RCCL ? "RCCL send/recv groups" : "peer-to-peer copies", st.key_mode == DEBWT_KEYS_EXCHANGE ? "exchanged" : "rescanned", st.rounds, t1 - t0, threads, t2 - t1,
               t3 - t2, t4 - t3, st.key_bytes_in / 1e9, st.blue_bytes_in / 1e9);
        fprintf(stderr, "success output bwt!\n");
    } else fprintf(stderr, "fetch/write: %s\n", debwt_strerror(rc));
    if (!rc && verify) {
        debwt_verify_report rep;
        memset(&rep, 0, sizeof rep);
        int vrc = debwt_multi_verify(m, &rep);
        rc = report_verify(vrc, &rep, debwt_multi_last_error(m));
    }
    free(bwt); free(hash_rows);
    debwt_multi_destroy(m);
    return rc ? 1 : 0;
}

int main(int argc, char **argv) {
    int verify = 0;
    {   /* --verify is a flag without a value: taken out before the reference's pair-wise parse (src/main.c:25-48) */
        int w = 1;
        for (int i = 1; i < argc; i++) {
            if (i < argc - 1 && !strcmp(argv[i], "--verify")) { verify = 1; continue; }
            argv[w++] = argv[i];
        }
        argc = w;
    }
    if (argc < 4 || (argc & 1) == 1) { usage(); return 1; }            /* src/main.c:25 */
    const char *source = argv[argc - 1], *obj = NULL, *dump = NULL;
    int k = 32, device = 0, iupac = 0, gpus = 0, devs[256], ndevs = 0, key_mode = -1, exchange = DEBWT_EXCHANGE_PEER_COPY;
    unsigned long long iupac_seed = 0;
    long threads = 8;
    for (int i = 1; i < argc - 1; i += 2) {
        if (!strcmp(argv[i], "-o")) obj = argv[i + 1];
        else if (!strcmp(argv[i], "-t")) {
            threads = atol(argv[i + 1]);
            if (threads == 0) { fprintf(stderr, "thread number must be a number!\n"); return 1; }
        } else if (!strcmp(argv[i], "-j")) { /* ignored */ }
        else if (!strcmp(argv[i], "-k")) {
            k = atoi(argv[i + 1]);
            if (k < 12 || k > 32) { fprintf(stderr, "-k: k-mer length (from 12 to 32, default 32)\n"); return 1; }
        } else if (!strcmp(argv[i], "--device")) device = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--gpus")) {
            gpus = atoi(argv[i + 1]);
            if (gpus < 1 || gpus > 255) { fprintf(stderr, "--gpus: 1 to 255\n"); return 1; }
        } else if (!strcmp(argv[i], "--devices")) {
            ndevs = 0;
            for (const char *p = argv[i + 1]; *p;) {
                char *end;
                long v = strtol(p, &end, 10);
                if (end == p || v < 0 || v > 1023 || ndevs >= 255 || (*end && *end != ',') || (*end == ',' && !end[1])) {
                    fprintf(stderr, "--devices: a comma-separated list of GPU ordinals, e.g. 0,1,2,3\n");
                    return 1;
                }
                devs[ndevs++] = (int)v;
                p = *end ? end + 1 : end;
            }
            if (!ndevs) { fprintf(stderr, "--devices: a comma-separated list of GPU ordinals, e.g. 0,1,2,3\n"); return 1; }
        }
        else if (!strcmp(argv[i], "--keys")) {
            if (!strcmp(argv[i + 1], "exchange")) key_mode = DEBWT_KEYS_EXCHANGE;
            else if (!strcmp(argv[i + 1], "rescan")) key_mode = DEBWT_KEYS_RESCAN;
            else if (!strcmp(argv[i + 1], "auto")) key_mode = -1;
            else { usage(); return 1; }
        }
        else if (!strcmp(argv[i], "--exchange")) {
            if (!strcmp(argv[i + 1], "rccl")) exchange = DEBWT_EXCHANGE_RCCL;
            else if (!strcmp(argv[i + 1], "peer")) exchange = DEBWT_EXCHANGE_PEER_COPY;
            else { usage(); return 1; }
        }
        else if (!strcmp(argv[i], "--dump")) dump = argv[i + 1];
        else if (!strcmp(argv[i], "--iupac")) { iupac = 1; iupac_seed = strtoull(argv[i + 1], NULL, 10); }
        else { usage(); return 1; }
    }
    if (!obj) { usage(); return 1; }
    FILE *probe = fopen(obj, "wb");                                       /* src/main.c:55-58 */
    if (!probe) { fprintf(stderr, "cannot create %s!\n", obj); return 1; }
    fclose(probe);
    remove(obj);

    fprintf(stderr, "run deBWT (MI355X path): sequence file %s, output %s, k-mer length %d\n", source, obj, k);
    if (ndevs && !gpus) gpus = ndevs;                                       /* --devices alone names the GPUs */
    if (ndevs && ndevs != gpus) {
        fprintf(stderr, "--devices names %d GPUs but --gpus asks for %d\n", ndevs, gpus);
        return 1;
    }
    if (dump) {
        struct stat sb;
        if (gpus) { fprintf(stderr, "--dump: one GPU only (the intermediates of a sharded build are spread over its GPUs)\n"); return 1; }
        if (stat(dump, &sb) || !S_ISDIR(sb.st_mode)) { fprintf(stderr, "--dump: %s is not a directory\n", dump); return 1; }
    }
    if (gpus) return multi_main(source, obj, k, (int)(threads > 256 ? 256 : threads), iupac, iupac_seed, gpus,
                                ndevs ? devs : NULL, key_mode, exchange, verify);
    double t0 = now();
    debwt_config cfg = {k, device, 0, 0};
    debwt_ctx *ctx = NULL;
    int rc = debwt_create(&cfg, &ctx);
    if (rc) { fprintf(stderr, "debwt_create: %s\n", debwt_strerror(rc)); return 1; }
    double t1 = now();
    /* workspace and output buffer on a helper thread while the file is read (plain FASTA: a base takes a byte of the file) */
    /* (--dump wants the text sorted in one key range: no compact plan then; DEBWT_CLI_NO_COMPACT: A/B) */
    struct reserve_job job = {ctx, 0, NULL, 0, 0.0, DEBWT_RESERVE_ONE_SHOT | (dump || getenv("DEBWT_CLI_NO_COMPACT") ? 0u : DEBWT_RESERVE_COMPACT)};
    pthread_t helper;
    int have_helper = 0;
    {
        /* a plain file: a base takes a byte.  (gzip: debwt_fasta_text_bound knows how much a block-gzip or a moderate one-member
         * file holds, but the driver's calls beside sixteen inflating threads cost them more -- the address space's lock --
         * than they save: read+pack 0.50 s instead of 0.28 s for 3.1 Gbp of BGZF; the workspace is reserved behind the parse) */
        size_t sl = strlen(source);
        int gz = sl > 3 && !strcmp(source + sl - 3, ".gz");
        job.n_bound = gz ? 0 : debwt_fasta_text_bound(source);
        if (job.n_bound > 64) have_helper = pthread_create(&helper, NULL, reserve_main, &job) == 0;
    }
    /* the reference's collect (src/collect#$.c:34-90): here `threads` host threads parse and pack the file */
    debwt_packed_text pt;
    char msg[256] = "";
    debwt_host_release_hold(1);          /* what the ingest gives up is released behind the load, not beside it */
    rc = debwt_pack_fasta_opts(source, (int)(threads > 256 ? 256 : threads), iupac ? DEBWT_FASTA_IUPAC_RANDOM : 0u, iupac_seed, &pt,
                               msg, sizeof msg);
    double t1b = now();
    if (have_helper) pthread_join(helper, NULL);
    if (!rc && job.bwt && pt.n > job.n_bound) { debwt_pinned_free(job.bwt); job.bwt = NULL; }   /* (a bound that did not hold: several gzip members) */
    double t1c = now();
    /* input whose length was not known beside the parse: the same compact plan now (a few key ranges more, a third
     * of the device memory: a process right behind another waits for the driver to clear what that one released) */
    if (!rc && !have_helper && !dump && !getenv("DEBWT_CLI_NO_COMPACT")) (void)debwt_reserve(ctx, pt.n, pt.nrec, 0.0, DEBWT_RESERVE_ONE_SHOT | DEBWT_RESERVE_COMPACT);
    /* (no page-locked OUT buffer on this way: the rows of 3.1 Gbp reach pageable memory 0.08 s later, page-locking 0.8 GB takes 0.15 s) */
    if (!rc) { rc = debwt_load_text(ctx, pt.words, pt.n, pt.sep, pt.nrec); if (rc) snprintf(msg, sizeof msg, "%s", debwt_last_error(ctx)); }
    if (rc) debwt_host_release_hold(0);
    if (rc) {
        fprintf(stderr, "%s (sequence must be ACGT only unless --iupac is given, records > 32 bases)\n", msg);
        if (job.bwt) debwt_pinned_free(job.bwt);
        debwt_free_packed(&pt);
        debwt_destroy(ctx);
        return 1;
    }
    double t2 = now();
    const uint64_t n = pt.n, nrec = pt.nrec;
    size_t nw = (size_t)((n + 31) >> 5);
    /* rows of finished key ranges travel to the host while the next range is sorted (debwt_build_to_host) */
    uint64_t *bwt = job.bwt, *hash_rows = malloc((nrec ? nrec : 1) * 8), dollar = 0;
    int bwt_pinned = bwt != NULL;
    if (!bwt) bwt = malloc(nw * 8);
    if (!bwt || !hash_rows) rc = DEBWT_ENOMEM;
    else if (!dump) rc = debwt_build_to_host(ctx, bwt, hash_rows, &dollar);
    else {
        /* the reference's stage sequence (src/main.c:83-149), the files of every stage written as it ends */
        const char *what = "kmerInfo";
        rc = debwt_dump_reference_files(ctx, dump, DEBWT_DUMP_KMERINFO);
        if (!rc) { what = "kmer sort"; rc = debwt_kmer_sort_rle(ctx); }
        if (!rc) { what = "classify"; rc = debwt_classify(ctx); }
        if (!rc) { what = "redSeq / redPoint / blueBound / case3bound"; rc = debwt_dump_reference_files(ctx, dump, DEBWT_DUMP_BLOCKS); }
        if (!rc) { what = "SP code"; rc = debwt_sp_generate(ctx); }
        if (!rc) { what = "spCode / spSpecialIndex / blueTable"; rc = debwt_dump_reference_files(ctx, dump, DEBWT_DUMP_SP); }
        if (!rc) { what = "blue sort"; rc = debwt_blue_sort(ctx); }
        if (!rc) { what = "assembly"; rc = debwt_bwt_assemble(ctx); }
        if (!rc) { what = "fetch"; rc = debwt_fetch_bwt(ctx, bwt, hash_rows, &dollar); }
        if (rc) fprintf(stderr, "--dump %s: stage '%s' failed\n", dump, what);
    }
    if (rc) fprintf(stderr, "build: %s %s\n", debwt_strerror(rc), debwt_last_error(ctx));
    debwt_host_release_hold(0);          /* (the build allocates too: the ingest's buffers go back beside the writing of the files) */
    double t3 = now();
    debwt_stats st;
    debwt_get_stats(ctx, &st);
    if (!rc && write_outputs(obj, bwt, n, hash_rows, nrec, dollar)) rc = DEBWT_EINVAL;   /* src/insertCase3.c:115-131 */
    if (rc) {
        if (bwt_pinned) debwt_pinned_free(bwt); else free(bwt);
        free(hash_rows);
        debwt_free_packed(&pt);
        debwt_destroy(ctx);
        return 1;
    }
    double t4 = now();
    printf("BWTLEN=%lu\n", (unsigned long)st.n);                           /* src/collect#$.c:59 */
    printf("the case3num is %lu\nthe blueBoundNum is %lu\nthe redCapacity is %lu\nthe blueCapacity is %lu\n",
           (unsigned long)st.case3num, (unsigned long)st.blue_bound_num, (unsigned long)st.red_capacity,
           (unsigned long)st.blue_capacity);                               /* src/generateSP.c:28-31 */
    printf("device init %.3f s, read+pack (%ld threads) %.3f s [workspace reserved beside it in %.3f s, %.3f s of that after the "
           "parse], load %.3f s, build+fetch %.3f s (device %.3f ms: extract %.2f sort %.2f classify %.2f SP %.2f blue %.2f "
           "assemble %.2f), write %.3f s\n",
           t1 - t0, threads, t1b - t1, job.seconds, t1c - t1b, t2 - t1c, t3 - t2, st.ms_total, st.ms_extract, st.ms_sort,
           st.ms_classify, st.ms_sp, st.ms_blue, st.ms_assemble, t4 - t3);
    fprintf(stderr, "success output bwt!\n");
    if (dump) printf("intermediates in %s: kmerInfo redSeq redPoint blueBound case3bound spCode spSpecialIndex blueTable\n", dump);
    int vfail = 0;
    if (verify) {
        debwt_verify_report rep;
        memset(&rep, 0, sizeof rep);
        int vrc = debwt_verify_device(ctx, NULL, NULL, 0, 0, &rep);
        vfail = report_verify(vrc, &rep, debwt_last_error(ctx));
    }
    if (n > ((uint64_t)1 << 30) && !getenv("DEBWT_CLI_TEARDOWN")) {
        /* the files are written and closed: a one-shot program leaves the hundreds of GB of device memory and page-locked
         * buffers to the operating system instead of unmapping them one by one first (1.5 s at 30 Gbp) */
        fflush(stdout); fflush(stderr);
        _exit(vfail);
    }
    if (bwt_pinned) debwt_pinned_free(bwt); else free(bwt);
    free(hash_rows);
    debwt_free_packed(&pt);
    debwt_destroy(ctx);
    return vfail;
}
