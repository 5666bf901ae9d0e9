// fasta_host.h -- multi-threaded FASTA -> 2-bit packed text (host side; see fasta_host.cpp)
#pragma once
#include <stddef.h>
#include <stdint.h>

struct PackedText {
    uint64_t *words;      // ((n + 63) >> 5) + 2 words, reference layout, 'T' at separators and 32 'T' behind the end
    uint64_t nwords;
    uint64_t n;           // BWTLEN: bases + one separator per record
    uint64_t *sep;        // nrec separator positions ascending, sep[nrec-1] == n-1
    uint64_t nrec;
    double seconds_read;  // open + map (or inflate)
    double seconds_pack;  // parse + pack
};

// flags: INGEST_IUPAC_RANDOM -- N and the other IUPAC ambiguity letters become one of the bases they stand for (the job
// of the reference's otherTool/transferN.c:8-32,57-60, which draws with rand(); here the draw is a hash of `seed` and
// the base's text position, so the packed text does not depend on the run or on the number of threads)
enum : unsigned { INGEST_IUPAC_RANDOM = 1u };
struct IngestOpts { unsigned flags; uint64_t seed; };

// 0 on success; on failure -1 and a message in err
int pack_fasta_buffer(const char *buf, size_t len, int threads, PackedText *out, char *err, size_t errlen,
                      IngestOpts opts = IngestOpts{0, 0});
int pack_fasta_file(const char *path, int threads, PackedText *out, char *err, size_t errlen,
                    IngestOpts opts = IngestOpts{0, 0});
void free_packed_text(PackedText *p);

// An upper bound of the number of text symbols the file can hold, from the file's size and framing alone (no inflate): a plain
// file: its bytes; block gzip (BGZF): the ISIZE of its members added up; one gzip member of less than 1 GB: its ISIZE (a text of
// 4 GB and more wraps there, and no DNA text deflates to less than a quarter).  0: not known (several plain members, a large
// member, not readable).  For a host that reserves device memory beside the parse (debwt_reserve).
uint64_t fasta_text_bound(const char *path);
