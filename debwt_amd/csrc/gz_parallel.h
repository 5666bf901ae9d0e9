// gz_parallel.h -- a one-member gzip file inflated by all host threads (see gz_parallel.cpp)
#pragma once
#include <stddef.h>

// z[0 .. zlen): the whole .gz file.  0: *out_buf (malloc'ed, caller frees) holds the *out_len inflated bytes, CRC32 and
// length checked against the gzip trailer; 1: not done (several members, not text, too small, a block start that could not
// be found or verified ...) -- inflate it serially; -1: out of memory.
int inflate_gzip_parallel(const unsigned char *z, size_t zlen, int threads, char **out_buf, size_t *out_len);

// z[0 .. zlen): a gzip file of SEVERAL plain members (cat a.gz b.gz ...): the member headers are found by their fixed bytes,
// every member is inflated on its own (many members: one thread each; few large ones: one after the other, each in pieces by
// inflate_gzip_parallel), checked against its CRC32 and ISIZE, and the members must chain from byte 0 to the end of the file.
// Same return values; 1 also for a file of one member.
int inflate_gzip_members(const unsigned char *z, size_t zlen, int threads, char **out_buf, size_t *out_len);
