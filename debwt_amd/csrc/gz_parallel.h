// gz_parallel.h -- a one-member gzip file inflated by all host threads (see gz_parallel.cpp)
#pragma once
#include <stddef.h>
#include <stdlib.h>
#include <sys/mman.h>

// A large buffer that is written once by all threads (inflated text): 2 MB aligned and marked for transparent huge pages, so
// that filling 1 GB is 512 page faults instead of 262,144 -- measured on the MI355X box's host, 16 threads faulting 4 KB pages
// of one address space took as long as inflating the text (profiles/r06_ingest_gz_rate.txt).  free() releases it; memory that
// is reserved but never touched costs nothing.
static inline void *big_malloc(size_t n) {
    if (n < ((size_t)8 << 20)) return malloc(n);
    void *p = NULL;
    if (posix_memalign(&p, (size_t)2 << 20, n)) return NULL;
    (void)madvise(p, n, MADV_HUGEPAGE);        // (refused where the kernel has them switched off: ordinary pages then)
    return p;
}

// Buffers handed over to be free()d by a thread of their own while the caller goes on (the parse of the text, the upload):
// returning 1 GB of touched memory to this host's kernel takes 50 ms whether huge pages or not, one thread or sixteen
// (scripts/micro/page_cost.cpp) -- as long as inflating it.  One thread, started at the first call, works the queue off; it is
// joined, the queue empty, when the library is unloaded or the process ends.  DEBWT_RELEASE_INLINE=1: free() here and now.
void release_later(void *const *ptrs, size_t n);
// release_hold(1) ... release_hold(0): nothing is released in between (an munmap takes the address space's lock for writing: the
// page faults of threads that are filling fresh buffers wait for it -- measured: the parse ran at half its rate beside it)
void release_hold(int on);

// z[0 .. zlen): the whole .gz file.  0: *out_buf (malloc'ed, caller frees) holds the *out_len inflated bytes, CRC32 and
// length checked against the gzip trailer; 1: not done (several members, not text, too small, a block start that could not
// be found or verified ...) -- inflate it serially; -1: out of memory.
int inflate_gzip_parallel(const unsigned char *z, size_t zlen, int threads, char **out_buf, size_t *out_len);

// z[0 .. zlen): a gzip file of SEVERAL plain members (cat a.gz b.gz ...): the member headers are found by their fixed bytes,
// every member is inflated on its own (many members: one thread each; few large ones: one after the other, each in pieces by
// inflate_gzip_parallel), checked against its CRC32 and ISIZE, and the members must chain from byte 0 to the end of the file.
// Same return values; 1 also for a file of one member.
int inflate_gzip_members(const unsigned char *z, size_t zlen, int threads, char **out_buf, size_t *out_len);
