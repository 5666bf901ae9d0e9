/*
 * debwt.c -- host program with the reference's command line over libdebwt_hip.so.
 *
 *   deBWT -o OUT [-t T] [-k K] [-j DIR] [--device D | --gpus G [--devices a,b,...] [--keys auto|exchange|rescan]] [--iupac SEED] INPUT.fa[.gz]
 *
 * Same contract as /root/reference/src/main.c:25-53,175-186: options are `flag value` pairs, INPUT last;
 * -k 12..32 (default 32); -t (default 8) = host threads of the FASTA ingest; -j accepted and ignored (no Jellyfish); OUT is probed by create+remove before any work
 * (src/main.c:55-58); exit status 0 on success, 1 with a message on stderr otherwise.  Output files OUT,
 * OUT.#, OUT.$ are those of src/insertCase3.c:115-131.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

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
                    "--iupac (optional): seed; N and other ambiguity letters become pseudo-random bases of their sets\n"
                    "                    (what otherTool/transferN does, reproducibly)\n"
                    "reference: sequence in fasta format (plain or gzip)\n");
}

static double now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

static int write_file(const char *path, const void *p, size_t bytes) {
    FILE *f = fopen(path, "wb");
    if (!f) { fprintf(stderr, "cannot create %s!\n", path); return -1; }
    size_t w = fwrite(p, 1, bytes, f);
    fclose(f);
    return w == bytes ? 0 : -1;
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

/* --gpus G: the same program over G GPUs (debwt_multi_*: one host thread per GPU inside the library) */
static int multi_main(const char *source, const char *obj, int k, int threads, int iupac, unsigned long long seed, int gpus,
                      const int *devs, int key_mode) {
    double t0 = now();
    debwt_config cfg = {k, 0, 0, 0};
    debwt_multi *m = NULL;
    int rc = debwt_multi_create(&cfg, devs, gpus, &m);
    if (rc) { fprintf(stderr, "debwt_multi_create (%d GPUs): %s\n", gpus, debwt_strerror(rc)); return 1; }
    debwt_multi_set_key_mode(m, key_mode);
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
        printf("%u GPUs, keys %s, %u key round(s): init %.3f s, read+pack+load (%d threads) %.3f s, build %.3f s, fetch+write %.3f s; "
               "shard 0 received %.3f GB of k-mers and %.3f GB of blue entries\n", st.ngpus,
               st.key_mode == DEBWT_KEYS_EXCHANGE ? "exchanged" : "rescanned", st.rounds, t1 - t0, threads, t2 - t1,
               t3 - t2, t4 - t3, st.key_bytes_in / 1e9, st.blue_bytes_in / 1e9);
        fprintf(stderr, "success output bwt!\n");
    } else fprintf(stderr, "fetch/write: %s\n", debwt_strerror(rc));
    free(bwt); free(hash_rows);
    debwt_multi_destroy(m);
    return rc ? 1 : 0;
}

int main(int argc, char **argv) {
    if (argc < 4 || (argc & 1) == 1) { usage(); return 1; }            /* src/main.c:25 */
    const char *source = argv[argc - 1], *obj = NULL;
    int k = 32, device = 0, iupac = 0, gpus = 0, devs[256], ndevs = 0, key_mode = -1;
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
    if (gpus) return multi_main(source, obj, k, (int)(threads > 256 ? 256 : threads), iupac, iupac_seed, gpus,
                                ndevs ? devs : NULL, key_mode);
    double t0 = now();
    debwt_config cfg = {k, device, 0, 0};
    debwt_ctx *ctx = NULL;
    int rc = debwt_create(&cfg, &ctx);
    if (rc) { fprintf(stderr, "debwt_create: %s\n", debwt_strerror(rc)); return 1; }
    double t1 = now();
    /* the reference's collect (src/collect#$.c:34-90): here `threads` host threads parse and pack the file */
    rc = debwt_load_fasta_opts(ctx, source, (int)(threads > 256 ? 256 : threads), iupac ? DEBWT_FASTA_IUPAC_RANDOM : 0u, iupac_seed);
    if (rc) {
        fprintf(stderr, "%s (sequence must be ACGT only unless --iupac is given, records > 32 bases)\n", debwt_last_error(ctx));
        debwt_destroy(ctx);
        return 1;
    }
    double t2 = now();
    rc = debwt_build(ctx);
    if (rc) { fprintf(stderr, "build: %s %s\n", debwt_strerror(rc), debwt_last_error(ctx)); debwt_destroy(ctx); return 1; }
    double t3 = now();
    debwt_stats st;
    debwt_get_stats(ctx, &st);
    const uint64_t n = st.n, nrec = st.nrec;
    size_t nw = (size_t)((n + 31) >> 5);
    uint64_t *bwt = malloc(nw * 8), *hash_rows = malloc((nrec ? nrec : 1) * 8), dollar = 0;
    rc = (bwt && hash_rows) ? debwt_fetch_bwt(ctx, bwt, hash_rows, &dollar) : DEBWT_ENOMEM;
    if (rc) fprintf(stderr, "fetch: %s\n", debwt_strerror(rc));
    else if (write_outputs(obj, bwt, n, hash_rows, nrec, dollar)) rc = DEBWT_EINVAL;   /* src/insertCase3.c:115-131 */
    if (rc) { free(bwt); free(hash_rows); debwt_destroy(ctx); return 1; }
    double t4 = now();
    printf("BWTLEN=%lu\n", (unsigned long)st.n);                           /* src/collect#$.c:59 */
    printf("the case3num is %lu\nthe blueBoundNum is %lu\nthe redCapacity is %lu\nthe blueCapacity is %lu\n",
           (unsigned long)st.case3num, (unsigned long)st.blue_bound_num, (unsigned long)st.red_capacity,
           (unsigned long)st.blue_capacity);                               /* src/generateSP.c:28-31 */
    printf("device init %.3f s, read+pack+load (%ld threads) %.3f s, build %.3f s (device %.3f ms: extract %.2f sort %.2f classify %.2f "
           "SP %.2f blue %.2f assemble %.2f), write %.3f s\n",
           t1 - t0, threads, t2 - t1, t3 - t2, st.ms_total, st.ms_extract, st.ms_sort, st.ms_classify, st.ms_sp, st.ms_blue,
           st.ms_assemble, t4 - t3);
    fprintf(stderr, "success output bwt!\n");
    debwt_destroy(ctx);
    free(bwt); free(hash_rows);
    return 0;
}
