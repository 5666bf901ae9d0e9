/*
 * ref_driver.c -- runs the reference's OWN stage functions (compiled by oracle/Makefile from
 * /root/reference/src/*.c where they lie; nothing of them is copied here) on one FASTA input,
 * so that the restatement in debwt_oracle.c can be pinned against real reference behaviour.
 *
 * TEST INFRASTRUCTURE ONLY (see debwt_oracle.h).
 *
 * The reference's main() (src/main.c:70) first shells out to Jellyfish, which is un-vendored and
 * absent from this image.  This driver therefore does what main() does from src/main.c:79 on --
 * the same stage calls in the same order with the same argument block -- and supplies the text
 * file `BIN/out` ("KMER<ws>COUNT" per distinct k-mer, the format of `jellyfish dump -c -t`,
 * src/kmercounting.sh:11) from orc_kmer_count().  Everything from mySort on is reference code.
 *
 * usage: ref_driver WORKDIR INPUT.fa OUT K THREADS
 *   writes OUT, OUT.#, OUT.$ (src/insertCase3.c:115-131), OUT.kmerInfo (copy of the sorted edge
 *   file of src/mySort.c:193-195), OUT.counters (the globals printed at src/generateSP.c:28-31),
 *   and the intermediates the stages hand to each other, copied out of WORKDIR between the stage
 *   calls before the next stage consumes and removes them: OUT.kmerdump (the k-mer dump mySort
 *   parses), OUT.redSeq / OUT.redPoint / OUT.blueBound / OUT.case3bound (generateBlocks,
 *   src/INandOut.c:347-366,396-417), OUT.spCode / OUT.blueTable / OUT.spSpecialIndex (generateSP,
 *   src/generateSP.c:626-672) and OUT.timing.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>
#include "debwt_oracle.h"

/* wall-clock per stage -> OUT.timing ("stage seconds" lines); insertCase3 exits the process itself
 * (src/insertCase3.c:137), so the last stage is closed by an atexit handler */
static char g_timing_path[2048];
static double g_t_last;
static const char *g_stage_open;
static double wall(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
static void stage_done(const char *name) {
    double t = wall();
    FILE *f = fopen(g_timing_path, "a");
    if (f) { fprintf(f, "%s %.6f\n", name, t - g_t_last); fclose(f); }
    g_t_last = t;
}
static void close_last_stage(void) { if (g_stage_open) stage_done(g_stage_open); }

/* reference globals (tentative definitions in its headers; linked with -fcommon) */
extern uint64_t trans[256];
extern int KMER_LENGTH_PlusOne, KMER_LENGTH;
extern uint64_t case3num, blueBoundNum, redCapacity, blueCapacity, spCodeLen, specialBranchNum, BWTLEN,
    countRead;
extern uint64_t *spCode, *blueTable;     /* src/sortBlue.h:1,5: filled by generateSP */
/* reference stage entry points, src/main.h:1-8 */
void *collect(void *arg_collect);
int generateBlocks(char *bin);
int generateSP(void **arg);
int sortBlue(void **arg);
int insertCase3(char *obj, char *bin);
void *getKmer(void *str);
int mySort(void **arg);

static char *read_fasta(const char *path, uint64_t **reclen_out, uint64_t *nrec_out) {
    FILE *f = fopen(path, "r");
    if (!f) { perror(path); exit(2); }
    size_t cap = 1 << 20, len = 0, rcap = 16, nrec = 0;
    char *seq = malloc(cap);
    uint64_t *reclen = malloc(rcap * 8);
    int c, in_header = 0, have = 0;
    while ((c = fgetc(f)) != EOF) {
        if (in_header) { if (c == '\n') in_header = 0; continue; }
        if (c == '>') {
            in_header = 1;
            if (nrec == rcap) { rcap *= 2; reclen = realloc(reclen, rcap * 8); }
            reclen[nrec++] = 0; have = 1;
            continue;
        }
        if (c == '\n' || c == '\r' || c == ' ' || c == '\t') continue;
        if (!have) { fprintf(stderr, "sequence before header\n"); exit(2); }
        if (len == cap) { cap *= 2; seq = realloc(seq, cap); }
        seq[len++] = (char)c; reclen[nrec - 1]++;
    }
    fclose(f);
    *reclen_out = reclen; *nrec_out = nrec;
    return seq;
}

static void copy_file(const char *a, const char *b) {
    FILE *fa = fopen(a, "rb"), *fb = fopen(b, "wb");
    if (!fa || !fb) { fprintf(stderr, "copy %s -> %s failed\n", a, b); exit(2); }
    char buf[1 << 16]; size_t r;
    while ((r = fread(buf, 1, sizeof buf, fa)) > 0) fwrite(buf, 1, r, fb);
    fclose(fa); fclose(fb);
}

int main(int argc, char **argv) {
    if (argc != 6) { fprintf(stderr, "usage: %s WORKDIR INPUT.fa OUT K THREADS\n", argv[0]); return 2; }
    char *bin = argv[1], *source = argv[2], *obj = argv[3];
    int k = atoi(argv[4]);
    uint64_t threads = (uint64_t)atoi(argv[5]);
    if (k < 12 || k > 32 || threads < 1) { fprintf(stderr, "bad k/threads\n"); return 2; }

    /* src/main.c:18-23,41-47 */
    trans['A'] = trans['a'] = 0; trans['C'] = trans['c'] = 1;
    trans['G'] = trans['g'] = 2; trans['T'] = trans['t'] = 3;
    trans['#'] = 4; trans['$'] = 5;
    KMER_LENGTH_PlusOne = k; KMER_LENGTH = k - 1;

    snprintf(g_timing_path, sizeof g_timing_path, "%s.timing", obj);
    remove(g_timing_path);
    g_t_last = wall();
    /* the Jellyfish dump, from the restated counter */
    uint64_t *reclen, nrec;
    char *seq = read_fasta(source, &reclen, &nrec);
    uint64_t total = 0;
    for (uint64_t r = 0; r < nrec; r++) total += reclen[r];
    uint8_t *sym = malloc(total + nrec);
    uint64_t n = orc_make_text(seq, reclen, nrec, sym);
    if (!n) { fprintf(stderr, "invalid input\n"); return 2; }
    uint64_t *kmers = malloc(n * 8), *counts = malloc(n * 8);
    uint64_t D = orc_kmer_count(sym, n, k, kmers, counts);
    char path[2048], dst[2048];
    snprintf(path, sizeof path, "%s/out", bin);
    FILE *fo = fopen(path, "w");
    if (!fo) { perror(path); return 2; }
    /* written in a scrambled order: a hash dump has no order and mySort must not rely on one */
    for (uint64_t t = 0; t < D; t++) {
        uint64_t i = (t * 7919 + 13) % D;
        if (D % 7919 == 0) i = t;
        char s[40];
        for (int j = 0; j < k; j++) s[j] = "ACGT"[(kmers[i] >> (62 - 2 * j)) & 3];
        s[k] = 0;
        fprintf(fo, "%s\t%lu\n", s, (unsigned long)counts[i]);
    }
    fclose(fo);
    snprintf(dst, sizeof dst, "%s.kmerdump", obj);
    copy_file(path, dst);
    free(kmers); free(counts); free(sym); free(seq); free(reclen);
    stage_done("kmer_dump_standin");

    /* src/main.c:79-83 */
    void *arg[3];
    arg[0] = (void *)bin; arg[1] = (void *)threads; arg[2] = (void *)source;
    if (mySort((void **)arg) != 0) { fprintf(stderr, "mySort failed\n"); return 1; }
    stage_done("mySort");
    snprintf(path, sizeof path, "%s/kmerInfo", bin);
    snprintf(dst, sizeof dst, "%s.kmerInfo", obj);
    copy_file(path, dst);
    /* src/main.c:88-103 (the two run concurrently there; they share nothing) */
    collect((void *)arg);
    getKmer((void *)bin);
    stage_done("collect_getKmer");
    if (generateBlocks(bin) != 1) return 1;          /* src/main.c:109 */
    stage_done("generateBlocks");
    {
        static const char *names[] = {"redSeq", "redPoint", "blueBound", "case3bound"};
        for (int i = 0; i < 4; i++) {
            snprintf(path, sizeof path, "%s/%s", bin, names[i]);
            snprintf(dst, sizeof dst, "%s.%s", obj, names[i]);
            copy_file(path, dst);
        }
        g_t_last = wall();
    }
    if (generateSP((void **)arg) != 1) return 1;     /* src/main.c:122 */
    stage_done("generateSP");
    {
        /* spCode: 2 bits per SP symbol, 32 per word; the last 32 of spCodeLen are padding (src/generateSP.c:215-216) */
        snprintf(dst, sizeof dst, "%s.spCode", obj);
        FILE *f = fopen(dst, "wb");
        fwrite(spCode, 8, (size_t)((spCodeLen - 32 + 31) / 32), f);
        fclose(f);
        snprintf(dst, sizeof dst, "%s.blueTable", obj);
        f = fopen(dst, "wb");
        fwrite(blueTable, 8, (size_t)blueCapacity, f);
        fclose(f);
        snprintf(path, sizeof path, "%s/spSpecialIndex", bin);
        snprintf(dst, sizeof dst, "%s.spSpecialIndex", obj);
        copy_file(path, dst);
    }
    snprintf(dst, sizeof dst, "%s.counters", obj);
    FILE *fc = fopen(dst, "w");
    fprintf(fc, "BWTLEN %lu\ncountRead %lu\ncase3num %lu\nblueBoundNum %lu\nredCapacity %lu\n"
                "blueCapacity %lu\nspCodeLen %lu\nspecialBranchNum %lu\ndistinctKmers %lu\n",
            (unsigned long)BWTLEN, (unsigned long)countRead, (unsigned long)case3num,
            (unsigned long)blueBoundNum, (unsigned long)redCapacity, (unsigned long)blueCapacity,
            (unsigned long)spCodeLen, (unsigned long)specialBranchNum, (unsigned long)D);
    fclose(fc);
    g_t_last = wall();                               /* the counters file above is this driver's, not a stage */
    if (sortBlue((void **)arg) != 1) return 1;       /* src/main.c:136 */
    stage_done("sortBlue");
    g_stage_open = "insertCase3";
    atexit(close_last_stage);
    insertCase3(obj, bin);                           /* src/main.c:149; exit(0) inside */
    return 1;
}
