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

// 0 on success; on failure -1 and a message in err
int pack_fasta_buffer(const char *buf, size_t len, int threads, PackedText *out, char *err, size_t errlen);
int pack_fasta_file(const char *path, int threads, PackedText *out, char *err, size_t errlen);
void free_packed_text(PackedText *p);
