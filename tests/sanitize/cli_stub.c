/* cli_stub.c -- stand-ins for the C ABI so that cli/debwt.c (argument handling, file writes, error paths) links and runs
 * under ASan/UBSan on a box without a GPU.  The "build" is a fixed tiny result: the point is the host program's own
 * memory handling, not the BWT.  Test infrastructure only. */
#include <stdlib.h>
#include <string.h>

#include "../../include/debwt_hip.h"

struct debwt_ctx { int loaded; };
struct debwt_multi { int g; int loaded; };
#define N_ROWS 100u
#define N_REC 3u

int debwt_create(const debwt_config *cfg, debwt_ctx **out) { (void)cfg; *out = calloc(1, sizeof **out); return *out ? 0 : DEBWT_ENOMEM; }
void debwt_destroy(debwt_ctx *c) { free(c); }
const char *debwt_strerror(int code) { return code ? "stub error" : "ok"; }
const char *debwt_last_error(const debwt_ctx *c) { (void)c; return "stub: no such file"; }
int debwt_load_fasta_opts(debwt_ctx *c, const char *path, int threads, unsigned flags, uint64_t seed) {
    (void)threads; (void)flags; (void)seed;
    if (strstr(path, "missing")) return DEBWT_EINVAL;
    c->loaded = 1; return 0;
}
int debwt_reserve(debwt_ctx *c, uint64_t n, uint64_t nrec, double branching, unsigned flags) { (void)c; (void)nrec; (void)branching; (void)flags; return n ? 0 : DEBWT_EINVAL; }
int debwt_pinned_alloc(size_t bytes, void **out) { *out = malloc(bytes); return *out ? 0 : DEBWT_ENOMEM; }
void debwt_pinned_free(void *p) { free(p); }
uint64_t debwt_fasta_text_bound(const char *path) { (void)path; return 0; }
void debwt_host_release_hold(int on) { (void)on; }
int debwt_pack_fasta_opts(const char *path, int threads, unsigned flags, uint64_t seed, debwt_packed_text *out, char *errbuf,
                          size_t errlen) {
    (void)threads; (void)flags; (void)seed;
    memset(out, 0, sizeof *out);
    if (strstr(path, "missing")) { if (errlen) { strncpy(errbuf, "stub: no such file", errlen - 1); errbuf[errlen - 1] = 0; } return DEBWT_EINVAL; }
    out->nwords = (N_ROWS + 63) / 32 + 2; out->words = calloc(out->nwords, 8); out->n = N_ROWS; out->nrec = N_REC;
    out->sep = calloc(N_REC, 8);
    return out->words && out->sep ? 0 : DEBWT_ENOMEM;
}
void debwt_free_packed(debwt_packed_text *p) { free(p->words); free(p->sep); p->words = NULL; p->sep = NULL; }
int debwt_load_text(debwt_ctx *c, const uint64_t *packed, uint64_t n, const uint64_t *sep, uint64_t nrec) {
    (void)sep; if (!packed || n != N_ROWS || nrec != N_REC) return DEBWT_EINVAL;
    c->loaded = 1; return 0;
}
int debwt_build_to_host(debwt_ctx *c, uint64_t *bwt, uint64_t *hash_rows, uint64_t *dollar) {
    if (!c->loaded) return DEBWT_ESTATE;
    memset(bwt, 0x1B, ((N_ROWS + 31) / 32) * 8); hash_rows[0] = 5; hash_rows[1] = 9; *dollar = 77; return 0;
}
int debwt_multi_set_exchange(debwt_multi *m, int backend) { (void)m; return backend == 0 || backend == 1 ? 0 : DEBWT_EINVAL; }
int debwt_build(debwt_ctx *c) { return c->loaded ? 0 : DEBWT_ESTATE; }
int debwt_get_stats(const debwt_ctx *c, debwt_stats *st) { (void)c; memset(st, 0, sizeof *st); st->n = N_ROWS; st->nrec = N_REC; return 0; }
int debwt_fetch_bwt(debwt_ctx *c, uint64_t *bwt, uint64_t *hash_rows, uint64_t *dollar) {
    (void)c; memset(bwt, 0x1B, ((N_ROWS + 31) / 32) * 8); hash_rows[0] = 5; hash_rows[1] = 9; *dollar = 77; return 0;
}
int debwt_multi_create(const debwt_config *cfg, const int *devices, int ngpus, debwt_multi **out) {
    (void)cfg;
    *out = calloc(1, sizeof **out);
    if (!*out) return DEBWT_ENOMEM;
    (*out)->g = ngpus;
    if (devices) for (int i = 0; i < ngpus; i++) if (devices[i] < 0) return DEBWT_EINVAL;       /* reads every entry */
    return 0;
}
void debwt_multi_destroy(debwt_multi *m) { free(m); }
const char *debwt_multi_last_error(const debwt_multi *m) { (void)m; return "stub: no such file"; }
int debwt_multi_set_key_mode(debwt_multi *m, int mode) { (void)m; (void)mode; return 0; }
int debwt_multi_load_fasta(debwt_multi *m, const char *path, int threads, unsigned flags, uint64_t seed) {
    (void)threads; (void)flags; (void)seed;
    if (strstr(path, "missing")) return DEBWT_EINVAL;
    m->loaded = 1; return 0;
}
int debwt_multi_build(debwt_multi *m) { return m->loaded ? 0 : DEBWT_ESTATE; }
int debwt_multi_get_stats(const debwt_multi *m, debwt_multi_stats *st, debwt_stats *s0) {
    memset(st, 0, sizeof *st); memset(s0, 0, sizeof *s0);
    st->n = N_ROWS; st->nrec = N_REC; st->ngpus = (uint32_t)m->g; st->rounds = 1; return 0;
}
int debwt_multi_fetch_bwt(debwt_multi *m, uint64_t *bwt, uint64_t *hash_rows, uint64_t *dollar) {
    (void)m; memset(bwt, 0x1B, ((N_ROWS + 31) / 32) * 8); hash_rows[0] = 5; hash_rows[1] = 9; *dollar = 77; return 0;
}
/* --verify / --dump (round 6).  DEBWT_STUB_VERIFY_FAIL in the environment makes the stub's verifier report a mismatch. */
#include <stdio.h>
static int stub_verify(debwt_verify_report *r) {
    memset(r, 0, sizeof *r);
    r->segments = 4; r->steps = N_ROWS; r->ok = getenv("DEBWT_STUB_VERIFY_FAIL") ? 0 : 1; r->mismatches = r->ok ? 0 : 3;
    return 0;
}
int debwt_verify_device(debwt_ctx *c, const uint64_t *d_words, const uint64_t *hash_rows, uint64_t dollar_row, uint64_t segments,
                        debwt_verify_report *r) {
    (void)c; (void)d_words; (void)hash_rows; (void)dollar_row; (void)segments; return stub_verify(r);
}
int debwt_multi_verify(debwt_multi *m, debwt_verify_report *r) { (void)m; return stub_verify(r); }
int debwt_kmer_sort_rle(debwt_ctx *c) { return c->loaded ? 0 : DEBWT_ESTATE; }
int debwt_classify(debwt_ctx *c) { return c->loaded ? 0 : DEBWT_ESTATE; }
int debwt_sp_generate(debwt_ctx *c) { return c->loaded ? 0 : DEBWT_ESTATE; }
int debwt_blue_sort(debwt_ctx *c) { return c->loaded ? 0 : DEBWT_ESTATE; }
int debwt_bwt_assemble(debwt_ctx *c) { return c->loaded ? 0 : DEBWT_ESTATE; }
int debwt_dump_reference_files(debwt_ctx *c, const char *dir, int stage) {
    (void)c;
    char path[1024];
    snprintf(path, sizeof path, "%s/stage%d", dir, stage);
    FILE *f = fopen(path, "wb");
    if (!f) return DEBWT_EIO;
    fclose(f);
    return 0;
}
