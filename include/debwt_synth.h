/*
 * debwt_synth.h -- formula-defined synthetic DNA collections for the measurements, host side.
 *
 * MEASUREMENT TOOLING, not a reference interface: the reference ships no generator and no data.  bench.py and the
 * large-size tests need texts of up to 30 Gbp that both boxes regenerate from a handful of numbers; the definition
 * of the distributions is debwt_amd/synth.py (numpy, used by the small tests and checked against this generator),
 * this is the same arithmetic on host threads, written straight into the reference's 2-bit text layout
 * (/root/reference/src/collect#$.c:61-90) so that a 30 Gbp text costs 7.5 GB of host memory and seconds, not 30 GB
 * and minutes.
 *
 * Distribution (synth.pan_chromosomes): one base genome of genome_len bases (uniform + repeat families at
 * repeat_coverage, optionally low-complexity content, see flags) -> `genomes` copies with independent SNPs at
 * snp_rate each (none when genomes == 1) -> every copy cut into the same nchrom records of chrom_len[] bases.
 * Text = records in order (genome-major), one separator behind each.
 */
#ifndef DEBWT_SYNTH_H
#define DEBWT_SYNTH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct debwt_synth debwt_synth;

typedef struct {
    uint64_t seed;
    uint64_t genome_len;        /* bases of one genome */
    uint32_t genomes;           /* >= 1 */
    uint32_t nchrom;            /* records per genome, >= 1 */
    const uint64_t *chrom_len;  /* nchrom lengths, each > 32, summing to genome_len */
    double snp_rate;            /* per base and genome (synth.py: 1e-3) */
    double repeat_coverage;     /* fraction of the base genome written over by repeat families (0.25) */
    double lowcx_fraction;      /* distribution R: fraction of the base genome that is low-complexity content
                                   (satellite arrays, homopolymer and microsatellite tracts); 0 = distribution P */
    uint32_t alu_copies;        /* distribution R: copies of one 300-base family at alu_divergence */
    double alu_divergence;
} debwt_synth_spec;

/* Builds the base genome (genome_len bytes of host memory) on `threads` host threads. 0 or DEBWT_E*. */
int debwt_synth_open(const debwt_synth_spec *spec, int threads, debwt_synth **out);
void debwt_synth_close(debwt_synth *s);
/* BWTLEN (bases + one separator per record), records, words of the packed text (((n + 63) >> 5) + 2) */
uint64_t debwt_synth_n(const debwt_synth *s);
uint64_t debwt_synth_nrec(const debwt_synth *s);
uint64_t debwt_synth_nwords(const debwt_synth *s);
/* the nrec separator positions, ascending, sep[nrec-1] == n-1 */
int debwt_synth_sep(const debwt_synth *s, uint64_t *sep);
/* packed words [w0, w1) of the text ('T' at separators, 32 'T' behind the end, zeros after) into dst[0 .. w1-w0);
 * census (optional): += number of A, C, G, T bases among the positions of those words */
int debwt_synth_words(const debwt_synth *s, uint64_t w0, uint64_t w1, int threads, uint64_t *dst, uint64_t census[4]);
/* codes (A0 C1 G2 T3) of bases [i0, i1) of genome `genome` (after its SNPs) */
int debwt_synth_codes(const debwt_synth *s, uint32_t genome, uint64_t i0, uint64_t i1, uint8_t *dst);

#ifdef __cplusplus
}
#endif
#endif
