/* lfo_internal.h -- TEST INFRASTRUCTURE ONLY (see lf_oracle.h). */
#ifndef LFO_INTERNAL_H
#define LFO_INTERNAL_H
#include <stddef.h>
#include "lf_oracle.h"

typedef int (*lfo_less_fn)(const void *a, const void *b, void *ctx);

void lfo_std_sort(void *base, size_t n, size_t es, lfo_less_fn less, void *ctx);
void lfo_push_heap(void *base, size_t n, size_t es, lfo_less_fn less, void *ctx);
void lfo_pop_heap(void *base, size_t n, size_t es, lfo_less_fn less, void *ctx);
void lfo_sort_heap(void *base, size_t n, size_t es, lfo_less_fn less, void *ctx);
int  lfo_pos2rid(const lfo_index_t *ix, int64_t pos);

#endif
