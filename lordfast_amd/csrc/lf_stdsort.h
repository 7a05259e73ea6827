/*
 * lf_stdsort.h -- order-exact restatements of the libstdc++ (GCC 11) algorithms whose tie order reaches
 * lordFAST's output (hard part 1): std::sort (introsort: threshold 16, median-of-3 of
 * first+1/mid/last-1 moved to first, depth 2*floor(lg n), heap-sort fallback, guarded + unguarded final
 * insertion; bits/stl_algo.h:1855-1960) and push_heap / pop_heap / sort_heap (bits/stl_heap.h).
 *
 * Macro-generated, typed: LF_DEFINE_STDSORT(name, T, LESS) where LESS(a,b) is an expression on `const T *`.
 */
#ifndef LF_STDSORT_H
#define LF_STDSORT_H
#include <stddef.h>

#ifndef LF_STDSORT_FN
#define LF_STDSORT_FN static            /* HIP translation units set this to `static __device__` */
#endif

#define LF_DEFINE_STDSORT(NAME, T, LESS)                                                                  \
LF_STDSORT_FN void NAME##_sift_up(T *a, long hole, long top, T v) {                                              \
    long parent = (hole - 1) / 2;                                                                         \
    while (hole > top && LESS(&a[parent], &v)) { a[hole] = a[parent]; hole = parent; parent = (hole - 1) / 2; } \
    a[hole] = v;                                                                                          \
}                                                                                                         \
LF_STDSORT_FN void NAME##_sift_down(T *a, long hole, long len, T v) {                                            \
    const long top = hole; long child = hole;                                                             \
    while (child < (len - 1) / 2) {                                                                       \
        child = 2 * (child + 1);                                                                          \
        if (LESS(&a[child], &a[child - 1])) child--;                                                      \
        a[hole] = a[child]; hole = child;                                                                 \
    }                                                                                                     \
    if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); a[hole] = a[child - 1]; hole = child - 1; } \
    NAME##_sift_up(a, hole, top, v);                                                                      \
}                                                                                                         \
LF_STDSORT_FN void NAME##_push_heap(T *a, long n) { T v = a[n - 1]; NAME##_sift_up(a, n - 1, 0, v); }            \
LF_STDSORT_FN void NAME##_pop_heap(T *a, long n) { if (n > 1) { T v = a[n - 1]; a[n - 1] = a[0]; NAME##_sift_down(a, 0, n - 1, v); } } \
LF_STDSORT_FN void NAME##_sort_heap(T *a, long n) { while (n > 1) { NAME##_pop_heap(a, n); n--; } }              \
LF_STDSORT_FN void NAME##_make_heap(T *a, long n) {                                                              \
    if (n < 2) return;                                                                                    \
    for (long parent = (n - 2) / 2;; parent--) { T v = a[parent]; NAME##_sift_down(a, parent, n, v); if (parent == 0) return; } \
}                                                                                                         \
LF_STDSORT_FN void NAME##_linear_insert(T *a, long last) {                                                       \
    T v = a[last]; long next = last - 1;                                                                  \
    while (LESS(&v, &a[next])) { a[last] = a[next]; last = next; next--; }                                \
    a[last] = v;                                                                                          \
}                                                                                                         \
LF_STDSORT_FN void NAME##_insertion(T *a, long first, long last) {                                               \
    if (first == last) return;                                                                            \
    for (long i = first + 1; i != last; i++) {                                                            \
        if (LESS(&a[i], &a[first])) { T v = a[i]; for (long k = i; k > first; k--) a[k] = a[k - 1]; a[first] = v; } \
        else NAME##_linear_insert(a, i);                                                                  \
    }                                                                                                     \
}                                                                                                         \
LF_STDSORT_FN void NAME##_intro(T *a, long first, long last, long depth) {                                       \
    while (last - first > 16) {                                                                           \
        if (depth == 0) { NAME##_make_heap(a + first, last - first); NAME##_sort_heap(a + first, last - first); return; } \
        depth--;                                                                                          \
        long x = first + 1, y = first + (last - first) / 2, z = last - 1, med;                            \
        if (LESS(&a[x], &a[y])) med = LESS(&a[y], &a[z]) ? y : (LESS(&a[x], &a[z]) ? z : x);             \
        else med = LESS(&a[x], &a[z]) ? x : (LESS(&a[y], &a[z]) ? z : y);                                 \
        { T t_ = a[first]; a[first] = a[med]; a[med] = t_; }                                              \
        long lo = first + 1, hi = last;                                                                   \
        for (;;) {                                                                                        \
            while (LESS(&a[lo], &a[first])) lo++;                                                         \
            hi--;                                                                                         \
            while (LESS(&a[first], &a[hi])) hi--;                                                         \
            if (!(lo < hi)) break;                                                                        \
            { T t_ = a[lo]; a[lo] = a[hi]; a[hi] = t_; }                                                  \
            lo++;                                                                                         \
        }                                                                                                 \
        NAME##_intro(a, lo, last, depth);                                                                 \
        last = lo;                                                                                        \
    }                                                                                                     \
}                                                                                                         \
LF_STDSORT_FN void NAME##_sort(T *a, long n) {                                                                   \
    if (n <= 0) return;                                                                                   \
    long lg = 0; for (long t_ = n; t_ > 1; t_ >>= 1) lg++;                                                \
    NAME##_intro(a, 0, n, 2 * lg);                                                                        \
    if (n > 16) { NAME##_insertion(a, 0, 16); for (long i = 16; i < n; i++) NAME##_linear_insert(a, i); } \
    else NAME##_insertion(a, 0, n);                                                                       \
}

#endif
