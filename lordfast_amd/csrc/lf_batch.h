/* lf_batch.h -- the read batch object (lf_reader.c makes them out of files, lf_batch_create out of a caller's strings) and its mapper-ready
 * form: lengths and the bit planes the batch crosses the host link as, made ONCE (at parse / create time) instead of inside every mapping call.
 * Reference: the chunk readChunk hands to mapSeqMT, made outside the mapping timer (src/Reads.cpp:84-104, src/baseFAST.cpp:59-75). */
#ifndef LF_BATCH_H
#define LF_BATCH_H
#include <stdint.h>
#include <stddef.h>

typedef struct lf_prepack {
    int min_read_len;                   /* reads shorter than this are not mapped (-l) and not packed */
    uint64_t bases, QW;                 /* packed bases; words per plane (planes[x * QW + w], x = lo / hi / valid) */
    uint64_t *planes; int pinned;       /* hipHostMalloc'd when a device runtime is there (uploads at link speed), else malloc'd */
    uint64_t *boff;                     /* [n + 1]: bit offset of read i in the planes (a read that is not packed: the offset of the next one) */
    uint64_t *exc_pos; uint8_t *exc_byte; uint64_t n_exc;      /* bytes outside upper-case ACGT, ascending positions */
} lf_prepack_t;

struct lf_read_batch {
    int n; uint64_t bases;
    const char **names, **seqs, **quals; uint32_t *lens; int rcap;
    char **blobs; size_t *blob_caps; int nblobs, capblobs;              /* window parser: one blob per piece */
    char *blob; size_t blob_n, blob_cap;
    size_t *off; int cap;                          /* 3 offsets per record into blob */
    lf_prepack_t *pre;                             /* NULL: not prepacked */
};

int  lf_read_batch_prepack(struct lf_read_batch *b, int min_read_len, int threads);      /* idempotent for one min_read_len */
void lf_prepack_free(lf_prepack_t *p);
#endif
