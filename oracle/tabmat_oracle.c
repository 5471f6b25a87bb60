/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the tabmat sandwich / matvec hot
 * path.  Nothing under oracle/ is a product path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline ("port").
 *
 * It is a plain-C (C11 + OpenMP) restatement of the reference's native loops
 * (Quantco/tabmat, src/tabmat/ext/ directory).  Each function cites the reference
 * file:line it follows; see oracle_kernels.inc.h.
 *
 * Parity status: PINNED against the reference's own known-answer tests and its
 * only data fixture (tests/test_oracle_*.py restate tests/test_matrices.py,
 * tests/test_fast_sandwich.py, tests/test_split_matrix.py and
 * tests/test_real_matrix.py case by case).  The reference's native kernels
 * themselves cannot be built in this image (they need mako, xsimd and jemalloc,
 * none of which is present, and stand-ins are not allowed), so there is no
 * oracle/_ref build.
 */
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- dense / categorical pass ---- */
#define F float
#define FS f32
#include "oracle_kernels.inc.h"
#undef F
#undef FS
#define F double
#define FS f64
#include "oracle_kernels.inc.h"
#undef F
#undef FS

/* ---- sparse pass: F x I ---- */
#define ORACLE_SPARSE_PASS 1
#define F float
#define FS f32
#define I int32_t
#define IS i32
#include "oracle_kernels.inc.h"
#undef I
#undef IS
#define I int64_t
#define IS i64
#include "oracle_kernels.inc.h"
#undef I
#undef IS
#undef F
#undef FS
#define F double
#define FS f64
#define I int32_t
#define IS i32
#include "oracle_kernels.inc.h"
#undef I
#undef IS
#define I int64_t
#define IS i64
#include "oracle_kernels.inc.h"
#undef I
#undef IS
#undef F
#undef FS

/*
 * split_col_subsets: map a sorted global column list onto per-block local
 * columns and output positions.
 * Reference: ext/split.pyx:157-209.  `indices` = concatenated per-block int64
 * index arrays, `offs[b]..offs[b+1]` delimits block b.  Outputs are written
 * block-major into sub_idx / sub_cols (capacity n_cols each, shared), with
 * per-block counts in counts[b]; entries of block b start at starts[b].
 */
void orc_split_col_subsets(const int64_t *indices, const int64_t *offs, int64_t n_blocks,
                           const int32_t *cols, int64_t n_cols,
                           int32_t *sub_idx, int32_t *sub_cols,
                           int64_t *counts, int64_t *starts)
{
    /* two passes so the block-major output layout is known */
    for (int pass = 0; pass < 2; pass++) {
        int64_t *next = (int64_t *)calloc((size_t)n_blocks, sizeof(int64_t));
        int64_t *fill = (int64_t *)calloc((size_t)n_blocks, sizeof(int64_t));
        if (pass == 1) {
            int64_t s = 0;
            for (int64_t b = 0; b < n_blocks; b++) { starts[b] = s; s += counts[b]; }
        } else {
            for (int64_t b = 0; b < n_blocks; b++) counts[b] = 0;
        }
        for (int64_t i = 0; i < n_cols; i++) {
            for (int64_t b = 0; b < n_blocks; b++) {
                const int64_t *ind = indices + offs[b];
                int64_t len = offs[b + 1] - offs[b];
                while (next[b] < len && ind[next[b]] < cols[i]) next[b]++;
                if (next[b] < len && ind[next[b]] == cols[i]) {
                    if (pass == 0) counts[b]++;
                    else {
                        sub_idx[starts[b] + fill[b]] = (int32_t)i;
                        sub_cols[starts[b] + fill[b]] = (int32_t)next[b];
                        fill[b]++;
                    }
                    next[b]++;
                    break;
                }
            }
        }
        free(next); free(fill);
    }
}

/* Reference: ext/split.pyx:211-217 (is_sorted). */
int orc_is_sorted_i64(const int64_t *a, int64_t n)
{
    for (int64_t i = 0; i + 1 < n; i++) if (a[i + 1] < a[i]) return 0;
    return 1;
}

int orc_num_threads(void) { return omp_get_max_threads(); }
void orc_set_num_threads(int n) { omp_set_num_threads(n); }
