/*
 * tabmat_hip.h -- C ABI of libtabmat_hip.so: the MI355X (gfx950) implementation of
 * tabmat's sandwich / matvec / transpose-matvec hot path.
 *
 * This is the drop-in boundary.  Each entry point replaces one native loop that
 * Quantco/tabmat's Cython layer (the .pyx files under src/tabmat/ext/) binds; the reference
 * interface it stands in for is cited per function (paths relative to the
 * reference repository).  Signatures are plain C: raw pointers and sizes, no
 * torch / numpy types.
 *
 * Conventions
 *   - Every data pointer is a DEVICE (HBM) pointer unless its name starts with h_.
 *     Block storage (X, CSR/CSC arrays, category codes) is uploaded once by the
 *     host side and stays resident; per-call vectors (d, v, rows, cols) and the
 *     outputs are device buffers too.
 *   - F is float (_f32) or double (_f64).  Sparse blocks use int32 column/row
 *     indices and int64 indptr on the device (the host side narrows/widens the
 *     int32-or-int64 arrays the reference accepts, ext/sparse.pyx:13-15).
 *   - rows == NULL means "all n rows" (n_rows is then ignored);
 *     cols == NULL means "all columns".  Index lists hold unique entries, as in
 *     the reference (dense_matrix.py:208).  d / v are always indexed by ORIGINAL
 *     row / column id (dense_helpers-tmpl.cpp:224).
 *   - order_f: 0 = C-contiguous (row-major), 1 = F-contiguous (column-major).
 *   - *_sandwich functions OVERWRITE out (the reference's wrappers hand the
 *     kernels a zeroed out and accumulate; here the zeroing is part of the call).
 *     *_matvec / *_rmatvec / *_transpose_matvec functions ACCUMULATE (out += ...),
 *     as the reference's in-place kernels do (ext/categorical.pyx:23-180,
 *     ext/sparse.pyx:79-103).
 *   - stream is a hipStream_t passed as void* (NULL = the null stream).  Calls are
 *     asynchronous with respect to the host; results are ordered on the stream.
 *   - Return value: 0 on success, a negative TM_E* code or a positive hipError_t
 *     otherwise; tm_last_error() returns a message for the calling thread.
 *     (The reference's kernels return void and cannot fail; all argument
 *     validation stays in the host language, util.py:27-67.)
 *   - Scratch: kernels use a workspace owned by the library, one per (device,
 *     stream), grown on demand with hipMalloc -- calls on different streams never
 *     share partial-sum buffers.  Growth waits for that stream only and is refused
 *     (TM_ENOMEM) while the stream is being captured into a HIP graph: run the op
 *     once before capturing.  Caller memory given with tm_set_workspace() serves
 *     every stream of its device; calls that share it must be stream-ordered.
 */
#ifndef TABMAT_HIP_H
#define TABMAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TM_OK 0
#define TM_EINVAL (-1)   /* bad argument */
#define TM_ENOMEM (-2)   /* workspace / allocation failure */
#define TM_EUNSUPPORTED (-3)

/* ---- runtime / memory helpers (for hosts that do not bring their own allocator) ---- */
int tm_version(void);
const char *tm_last_error(void);
int tm_device_count(int *count);
int tm_set_device(int device);
int tm_device_info(char *name, int name_len, int *compute_units, int64_t *hbm_bytes);
int tm_malloc(void **ptr, size_t bytes);
int tm_free(void *ptr);
int tm_memcpy_h2d(void *dst, const void *h_src, size_t bytes, void *stream);
int tm_memcpy_d2h(void *h_dst, const void *src, size_t bytes, void *stream);
int tm_memset(void *dst, int value, size_t bytes, void *stream);
int tm_stream_synchronize(void *stream);
/* Give the library caller-owned scratch for the current device (ptr == NULL: go back
 * to the internal allocation). */
int tm_set_workspace(void *ptr, size_t bytes);
/* Counter that changes whenever the library's workspace pointer of any device changes (growth or
 * tm_set_workspace).  A caller that captured kernel launches into a HIP graph must re-capture
 * when the value differs from the one read at capture time: the graph holds the old pointer. */
int tm_workspace_generation(int64_t *generation);
/* ---- event helpers so a ctypes host can time kernels on the launch stream ---- */
int tm_event_create(void **event);
int tm_event_destroy(void *event);
int tm_event_record(void *event, void *stream);
int tm_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises on stop */

/* ---- tuning knobs: integer settings that pick between launch geometries of one kernel (the
 * defaults are the measured optima; a profile run compares geometries inside one process).
 * Keys in use: "k2_waves" (8 / 12 / 16 waves per workgroup of the chunked sparse sandwich),
 * "catdense_waves" / "catsparse_waves" (waves of the fused categorical cross terms), "co_grid"
 * (workgroups of the co-resident syrk), "wg_log" (device pointer to a placement log: uint64
 * {count, capacity, 0, 0} followed by records {XCC_ID << 32 | HW_ID, start, end, kernel tag} that
 * the instrumented kernels append per workgroup, s_memrealtime ticks; 0 = off); experiment knobs that used to be
 * environment variables: "k2_slots" (8 / 4 / 2 slots per row and chunk of the chunked sparse sandwich, 0 = by
 * density), "lg_lockstep" (soft lockstep of the lane-group K3's column-half workgroups, default 1), "k2b_waves",
 * "ent_rounds" / "lg_rounds" (workgroup rounds of the K3 kernels), "ent_prof" (device pointer for the -DEN_PROF
 * build's section counters).  A knob changes launch geometry, never results. ---- */
/* (value INT64_MIN removes the setting: the built-in default applies again) */
int tm_tune_set(const char *h_key, int64_t value);
int tm_tune_get(const char *h_key, int64_t dflt, int64_t *value);

/* ---- in-library timing of an op's MAIN kernel (events on the launch stream) ---- */
int tm_profile_enable(int on);
int tm_profile_last_ms(float *ms); /* duration of the last recorded main kernel; synchronises */

/* =====================================================================================
 * Dense block  (reference: ext/dense.pyx + ext/dense_helpers-tmpl.cpp)
 * ===================================================================================== */

/* out[Ci,Cj] = sum_{k in rows} X[k,cols[Ci]] * d[k] * X[k,cols[Cj]];  out is
 * [n_cols x n_cols] row-major, full symmetric matrix.
 * Replaces _dense{C,F}_sandwich<int,F> (ext/dense_helpers-tmpl.cpp:266-311) as bound by
 * dense_sandwich (ext/dense.pyx:19-44).  MFMA row-weighted syrk. */
int tm_dense_sandwich_f32(const float *X, int64_t n, int64_t m, int order_f, const float *d,
                          const int32_t *rows, int64_t n_rows, const int32_t *cols,
                          int64_t n_cols, float *out, void *stream);
int tm_dense_sandwich_f64(const double *X, int64_t n, int64_t m, int order_f, const double *d,
                          const int32_t *rows, int64_t n_rows, const int32_t *cols,
                          int64_t n_cols, double *out, void *stream);

/* The same product for an unrestricted, 16-byte aligned, C-ordered f64 block of m <= 128 columns
 * (any parity since round 5), as a kernel sized to SHARE its compute units with an LDS- or HBM-bound
 * partner that runs on another stream (27.8 KB of LDS, <= 168 registers, work items handed out
 * through an atomic counter): the dense term of SplitMatrix.sandwich (split_matrix.py:337-354,
 * ext/dense_helpers-tmpl.cpp:266-311) overlapped with the sparse / categorical terms.
 * colsum (length m, may be NULL) receives X' d from the same pass (standardized_mat.py:149-150). */
int tm_dense_sandwich_co_f64(const double *X, int64_t n, int64_t m, const double *d, double *out,
                             double *colsum, void *stream);

/* X' diag(d) X of an unrestricted, 16-byte aligned, C-ordered f64 block of m <= 128 columns (any parity
 * since round 5) on the INT8 matrix cores (Ozaki-style slicing): diag(sqrt d) X in 40-bit fixed point per
 * column (scale from colmax[i] = max_r |X[r][i]|, length m, computed once per block by the caller,
 * and from max d), five balanced base-256 digits per entry, the 22 digit-pair products of weight
 * >= 2^16 accumulated exactly in int32 with v_mfma_i32_16x16x64_i8 and folded into f64 every 2048
 * rows (error ~2e-14 of max |out| on well-scaled data; model in csrc/syrk_i8.hip).  The part of
 * the envelope that depends on d is checked on the device -- a negative or non-finite weight before
 * the product, colmax_i^2 max(d) <= min(64, 2^27 / n) out[i][i] after it -- and a call outside it runs
 * the f64 kernel of tm_dense_sandwich_co_f64 instead (no host synchronisation).  The caller vouches
 * for finite X.  Replaces _denseC_sandwich<int, double> (ext/dense_helpers-tmpl.cpp:266-311).
 * out (m, m) is overwritten.
 * _xtd_: colsum (length m) = X' d from the same pass (replaces the transpose_matvec call of
 * standardized_mat.py:149-150). */
int tm_dense_sandwich_i8_f64(const double *X, int64_t n, int64_t m, const double *d, const double *colmax,
                             double *out, void *stream);
int tm_dense_sandwich_i8_xtd_f64(const double *X, int64_t n, int64_t m, const double *d, const double *colmax,
                                 double *out, double *colsum, void *stream);
/* Blocks of 130 .. 512 (even) columns: 128-column panels -- the diagonal panels on the int8 matrix cores in
 * place (each with its own envelope check and f64 hand-over), the off-diagonal panel pairs on the float64
 * MFMA (the j-panels of ext/dense_helpers-tmpl.cpp:289).  colmax: length m.  out (m, m) is overwritten. */
int tm_dense_sandwich_i8_wide_f64(const double *X, int64_t n, int64_t m, const double *d, const double *colmax,
                                  double *out, void *stream);
/* CENTRED forms (round 5): the product of X - 1 center' -- (X - 1 c')' diag(d) (X - 1 c') and, in colsum,
 * (X - 1 c')' d -- with the centre subtracted on the way in (K1e: before the fixed-point conversion, so the 40
 * bits are spent on x - c; colmax[i] is then max_r |X[r][i] - center[i]|).  What StandardizedMatrix.sandwich
 * (standardized_mat.py:123-172) needs: the reference subtracts mean-sized rank-one terms from the raw product,
 * which amplifies any error of the product by (mean / std)^2; here those terms never form.  center: length m,
 * indexed by the column of X (also under cols).  colsum / history may be NULL.  The generic form takes any
 * order / rows / cols like tm_dense_sandwich_*. */
int tm_dense_sandwich_centered_f32(const float *X, int64_t n, int64_t m, int order_f, const float *d,
                                   const int32_t *rows, int64_t n_rows, const int32_t *cols, int64_t n_cols,
                                   const float *center, float *out, void *stream);
int tm_dense_sandwich_centered_f64(const double *X, int64_t n, int64_t m, int order_f, const double *d,
                                   const int32_t *rows, int64_t n_rows, const int32_t *cols, int64_t n_cols,
                                   const double *center, double *out, void *stream);
int tm_dense_sandwich_co_centered_f64(const double *X, int64_t n, int64_t m, const double *d, const double *center,
                                      double *out, double *colsum, void *stream);
int tm_dense_sandwich_i8_centered_f64(const double *X, int64_t n, int64_t m, const double *d, const double *colmax,
                                      const double *center, double *out, double *colsum, int32_t *history,
                                      void *stream);
int tm_dense_sandwich_i8_wide_centered_f64(const double *X, int64_t n, int64_t m, const double *d,
                                           const double *colmax, const double *center, double *out, void *stream);
/* The same with a per-matrix HISTORY (int32[tm_dense_sandwich_i8_history_words()] in device memory, zeroed by
 * the caller once; colsum may be NULL): [0] counts consecutive calls whose weights left the envelope after the
 * product, [1] the calls, [4 ..] hold the diagonal of the previous call's result (128 doubles).  A call whose
 * weights fail the envelope test against THAT diagonal skips the int8 kernel on the device before the product
 * (round 4: a miss costs the f64 kernel alone -- the weights of an IRLS solver move slowly); after three
 * misses in a row the int8 kernel is skipped anyway and tried again every 32nd call. */
int tm_dense_sandwich_i8_history_words(void);
int tm_dense_sandwich_i8_hist_f64(const double *X, int64_t n, int64_t m, const double *d, const double *colmax,
                                  double *out, double *colsum, int32_t *history, void *stream);

/* X' diag(d) X of an unrestricted, 16-byte aligned, C-ordered FLOAT32 block of m = 4 k <= 256 columns
 * on the bf16 matrix cores: every element of diag(sqrt|d|) X is split into three bf16 pieces (24
 * mantissa bits) and the six leading piece products are accumulated in f32 with
 * v_mfma_f32_16x16x32_bf16 -- the accuracy of an f32 product accumulated in f32, at 16x the rate of
 * the f32-input MFMA (which runs at the f32 vector rate on gfx950).  Negative weights flip the bf16
 * sign bits of the A operand.  Replaces _denseC_sandwich<int, float> (ext/dense_helpers-tmpl.cpp:
 * 266-311) for BASELINE configs[1] (10M x 256).  out (m, m) is overwritten. */
int tm_dense_sandwich_bf16x3_f32(const float *X, int64_t n, int64_t m, const float *d, float *out,
                                 void *stream);

/* out[Ci] += sum_{Cj} X[rows[Ci], cols[Cj]] * v[cols[Cj]]   (v has length m).
 * Replaces _dense{C,F}_matvec (ext/dense_helpers-tmpl.cpp:385-417; ext/dense.pyx:76-101) and,
 * with rows == cols == NULL, the BLAS gemv of dense_matrix.py:212-217. */
int tm_dense_matvec_f32(const float *X, int64_t n, int64_t m, int order_f, const float *v,
                        const int32_t *rows, int64_t n_rows, const int32_t *cols,
                        int64_t n_cols, float *out, void *stream);
int tm_dense_matvec_f64(const double *X, int64_t n, int64_t m, int order_f, const double *v,
                        const int32_t *rows, int64_t n_rows, const int32_t *cols,
                        int64_t n_cols, double *out, void *stream);

/* out[Cj] += sum_{Ci} X[rows[Ci], cols[Cj]] * v[rows[Ci]]   (v has length n; out length n_cols).
 * Replaces _dense{C,F}_rmatvec (ext/dense_helpers-tmpl.cpp:314-383; ext/dense.pyx:48-73) and the
 * unrestricted X.T.dot(v). */
int tm_dense_rmatvec_f32(const float *X, int64_t n, int64_t m, int order_f, const float *v,
                         const int32_t *rows, int64_t n_rows, const int32_t *cols,
                         int64_t n_cols, float *out, void *stream);
int tm_dense_rmatvec_f64(const double *X, int64_t n, int64_t m, int order_f, const double *v,
                         const int32_t *rows, int64_t n_rows, const int32_t *cols,
                         int64_t n_cols, double *out, void *stream);

/* out[j] += sum_i w[i] * (X[i,j] - shift[j])^2, all rows / columns (standardize()'s column
 * variances).  Replaces transpose_square_dot_weights (ext/dense.pyx:103-122). */
int tm_dense_col_sq_dev_f32(const float *X, int64_t n, int64_t m, int order_f, const float *w,
                            const float *shift, float *out, void *stream);
int tm_dense_col_sq_dev_f64(const double *X, int64_t n, int64_t m, int order_f, const double *w,
                            const double *shift, double *out, void *stream);

/* =====================================================================================
 * Sparse block  (reference: ext/sparse.pyx + ext/sparse_helpers-tmpl.cpp)
 * Only the CSR twin (sparse_matrix.py:133-143) lives on the device:
 * (csr_data[nnz], csr_indices[nnz] int32, csr_indptr[n+1] int64), indices sorted per row.
 * ===================================================================================== */

/* out = A[rows,cols]^T diag(d) A[rows,cols], [n_cols x n_cols] row-major, full symmetric.
 * Replaces sparse_sandwich (ext/sparse.pyx:17-77).  Row-streaming over the CSR twin. */
int tm_sparse_sandwich_f32(const float *csr_data, const int32_t *csr_indices,
                           const int64_t *csr_indptr, int64_t n, int64_t m, const float *d,
                           const int32_t *rows, int64_t n_rows, const int32_t *cols,
                           int64_t n_cols, float *out, void *stream);
int tm_sparse_sandwich_f64(const double *csr_data, const int32_t *csr_indices,
                           const int64_t *csr_indptr, int64_t n, int64_t m, const double *d,
                           const int32_t *rows, int64_t n_rows, const int32_t *cols,
                           int64_t n_cols, double *out, void *stream);

/* Unrestricted fast path of the same product on the CHUNK-MAJOR twin of the block: the nnz
 * entries regrouped by column chunk c = column / tm_sparse_chunk_cols(), inside a chunk by row,
 * column order kept (cm_data / cm_indices, global column numbers);
 * cptr[c * (n + 1) + k] = index (into cm_data / cm_indices, < 2^31) of the first entry of row k in
 * chunk c, cptr[c * (n + 1) + n] = end of chunk c; c = 0 .. NCH - 1, NCH = ceil(m / chunk_cols).
 * Rows must be sorted by column and duplicate-free.
 * Restrictions are applied by the host side (masked d, sub-selection of the result). */
int tm_sparse_chunk_cols(void);
int tm_sparse_sandwich_chunked_f32(const float *cm_data, const int32_t *cm_indices,
                                   const int32_t *cptr, int64_t n, int64_t m, int64_t nnz,
                                   const float *d, float *out, void *stream);
int tm_sparse_sandwich_chunked_f64(const double *cm_data, const int32_t *cm_indices,
                                   const int32_t *cptr, int64_t n, int64_t m, int64_t nnz,
                                   const double *d, double *out, void *stream);

/* K2e (round 5): the unrestricted sparse self sandwich of WIDE blocks at a cost proportional to the entries and
 * PAIRS of nonzeros of a row, as the reference's loop is (ext/sparse.pyx:55-74), instead of rows x tiles: the
 * output is accumulated tile by tile in LDS (128 x 128, lower triangle of tiles) and the work of tile (I, J) is the
 * stream of the entries of column chunk I, lane <-> entry; every lane walks its row's list in chunk J (diagonal
 * tiles: its own row's list up to itself), one LDS atomic per pair.  Operands: cptr int32 [ceil(m / 128)][n + 1] of
 * the chunk-major twin (tm_sparse_sandwich_chunked_*) and cm_rec int32 [nnz][4], one 16-byte record per
 * chunk-major entry: f64 {value low word, value high word, column, row}, f32 {value bits, column, row, 0}.
 * m <= 16384, nnz < 2^31.  out (m, m) is overwritten. */
int tm_sparse_sandwich_pairs_f32(const int32_t *cm_rec, const int32_t *cptr, int64_t n, int64_t m, int64_t nnz,
                                 const float *d, float *out, void *stream);
int tm_sparse_sandwich_pairs_f64(const int32_t *cm_rec, const int32_t *cptr, int64_t n, int64_t m, int64_t nnz,
                                 const double *d, double *out, void *stream);

/* The same on PACKED records: {value, row << 7 | column inside the 128-column chunk} -- 12 bytes per entry for f64
 * (three words), 8 for f32 -- for blocks of fewer than 2^25 rows; the array must be readable 4 bytes beyond its end. */
int tm_sparse_sandwich_pairs_pk_f32(const int32_t *cm_rec_pk, const int32_t *cptr, int64_t n, int64_t m, int64_t nnz,
                                    const float *d, float *out, void *stream);
int tm_sparse_sandwich_pairs_pk_f64(const int32_t *cm_rec_pk, const int32_t *cptr, int64_t n, int64_t m, int64_t nnz,
                                    const double *d, double *out, void *stream);

/* The same product on a static BLOCK LIST (csrc/sparse_blocks.hip): every (row, tile) of the
 * chunk-major twin is cut, once per matrix, into blocks of at most 8 x 8 entries -- block (a, b)
 * pairs entries 8a .. 8a+7 of the row's list in chunk I with entries 8b .. 8b+7 of its list in chunk
 * J (b <= a on diagonal tiles) -- so that rows with more than 8 entries in a chunk need no overhang
 * phases in the kernel.  blocks: int32 [n_blocks][4] = {first A entry, first B entry, row,
 * nA | nB << 8} (entry indices into cm_data / cm_indices), tile after tile in the order
 * part = I (I + 1) / 2 + J.  Flag bits of the 4th word: 1 << 16 / 1 << 17 = the A / B side holds
 * <= 4 entries next to a longer other side (lane t then takes entry t & 3 of it).  wg_tab: int32
 * [n_wg][8], one record per workgroup = {part, slot, first block, end block, end of the FULL blocks,
 * waves on the FULL list, first row, last row}: the host deals every tile's blocks to its workgroups
 * in row order (slot = index of the workgroup inside its tile, < max_nb); inside a workgroup the
 * FULL blocks (both sides > 4 entries: 8 DPP steps) come first, then the HALF ones (4 steps), and the
 * workgroup's 16 waves are split between the two lists in proportion to their cost.
 * n < 2^29, nnz < 2^31. */
int tm_sparse_sandwich_blocks_f32(const float *cm_data, const int32_t *cm_indices, const int32_t *cptr,
                                  int64_t n, int64_t m, int64_t nnz, const int32_t *blocks,
                                  const int32_t *wg_tab, int n_wg, int max_nb, const float *d, float *out,
                                  void *stream);
int tm_sparse_sandwich_blocks_f64(const double *cm_data, const int32_t *cm_indices, const int32_t *cptr,
                                  int64_t n, int64_t m, int64_t nnz, const int32_t *blocks,
                                  const int32_t *wg_tab, int n_wg, int max_nb, const double *d, double *out,
                                  void *stream);

/* The same with the columns as ONE BYTE per entry: cm_col8[p] = column of chunk-major entry p inside its
 * 128-column chunk (cm_indices[p] % tm_sparse_chunk_cols()).  The entry gathers of the kernel move a quarter of the
 * index bytes. */
int tm_sparse_sandwich_blocks_u8_f32(const float *cm_data, const uint8_t *cm_col8, const int32_t *cptr,
                                     int64_t n, int64_t m, int64_t nnz, const int32_t *blocks,
                                     const int32_t *wg_tab, int n_wg, int max_nb, const float *d, float *out,
                                     void *stream);
int tm_sparse_sandwich_blocks_u8_f64(const double *cm_data, const uint8_t *cm_col8, const int32_t *cptr,
                                     int64_t n, int64_t m, int64_t nnz, const int32_t *blocks,
                                     const int32_t *wg_tab, int n_wg, int max_nb, const double *d, double *out,
                                     void *stream);
/* Byte columns AND 12-byte block descriptors (round 6): blocks12 [B][3] = {first A entry, first B entry,
 * row | (nA - 1) << 24 | (nB - 1) << 27 | short-A flag << 30 | short-B flag << 31}; blocks of fewer than 2^24 rows.
 * Same kernel, a quarter less descriptor storage (the reference loop: ext/sparse.pyx:55-74). */
int tm_sparse_sandwich_blocks_p12_f32(const float *cm_data, const uint8_t *cm_col8, const int32_t *cptr,
                                      int64_t n, int64_t m, int64_t nnz, const int32_t *blocks12,
                                      const int32_t *wg_tab, int n_wg, int max_nb, const float *d, float *out,
                                      void *stream);
int tm_sparse_sandwich_blocks_p12_f64(const double *cm_data, const uint8_t *cm_col8, const int32_t *cptr,
                                      int64_t n, int64_t m, int64_t nnz, const int32_t *blocks12,
                                      const int32_t *wg_tab, int n_wg, int max_nb, const double *d, double *out,
                                      void *stream);

/* out[nA x nB] = A[rows,A_cols]^T diag(d) B[rows,B_cols]; A sparse (CSR, n x m), B dense (n x r).
 * Replaces _csr_dense{C,F}_sandwich (ext/sparse_helpers-tmpl.cpp:23-146) as bound by
 * csr_dense_sandwich (ext/sparse.pyx:211-260). */
int tm_csr_dense_sandwich_f32(const float *csr_data, const int32_t *csr_indices,
                              const int64_t *csr_indptr, int64_t n, int64_t m, const float *B,
                              int64_t r, int order_f, const float *d, const int32_t *rows,
                              int64_t n_rows, const int32_t *A_cols, int64_t nA,
                              const int32_t *B_cols, int64_t nB, float *out, void *stream);
int tm_csr_dense_sandwich_f64(const double *csr_data, const int32_t *csr_indices,
                              const int64_t *csr_indptr, int64_t n, int64_t m, const double *B,
                              int64_t r, int order_f, const double *d, const int32_t *rows,
                              int64_t n_rows, const int32_t *A_cols, int64_t nA,
                              const int32_t *B_cols, int64_t nB, double *out, void *stream);

/* Same product as tm_csr_dense_sandwich_* with rows = A_cols = B_cols = NULL (B C- or F-ordered),
 * on the slab-blocked column-major twin of the sparse block (the fast path; restrictions are
 * applied by the host side through a masked d and sub-selection of the small result).
 * Layout: rows cut into slabs of tm_slab_rows() rows; within a slab nonzeros ordered by
 * (column, row); vals[e] = A[k,i]; koff[e] = (k - slab*R) * 64 * sizeof(F);
 * cnt[slab][mpad] run length per (slab, column), mpad = C*ceil(m/C), C = tm_slab_group_cols();
 * gptr[slab*G + g] start of column-group g (C columns) of that slab, G = ceil(m/C),
 * with one trailing total.  out is [m x r] row-major. */
int tm_slab_rows(void);
int tm_slab_group_cols(void);
int tm_csr_dense_sandwich_slab_f32(const float *vals, const uint32_t *koff, const uint16_t *cnt,
                                   const int64_t *gptr, int64_t n, int64_t m, const float *B,
                                   int64_t r, int order_f, const float *d, float *out, void *stream);
int tm_csr_dense_sandwich_slab_f64(const double *vals, const uint32_t *koff, const uint16_t *cnt,
                                   const int64_t *gptr, int64_t n, int64_t m, const double *B,
                                   int64_t r, int order_f, const double *d, double *out, void *stream);

/* Same unrestricted product for a C-ordered B with 16-byte aligned rows on the interleaved-ELL
 * twin of the sparse block.  Rows cut into slabs of R = tm_slab_rows() rows, columns into groups
 * of C = tm_slab_group_cols() = 32, G = ceil(m / C).  The nonzeros of one (slab, group) are I
 * iterations of 64 slots, I = ceil(longest column run of the group in the slab / 2);
 * slot it*64 + 2c + u = the (2 it + u)-th nonzero (rows ascending) of column c of the group,
 * or padding {vals 0, koff 0xFFFFFFFF}; vals[e] = A[k,i], koff[e] = (k - slab*R) * 64 * sizeof(F).
 * gptr[slab*G + g] = first slot of the block (a multiple of 64), one trailing total.
 * m = G*C kernel columns (the host may permute / pad the columns); out is [m x r] row-major. */
int tm_csr_dense_sandwich_ell_f32(const float *vals, const uint32_t *koff, const int64_t *gptr,
                                  int64_t n, int64_t m, const float *B, int64_t r, const float *d,
                                  float *out, void *stream);
int tm_csr_dense_sandwich_ell_f64(const double *vals, const uint32_t *koff, const int64_t *gptr,
                                  int64_t n, int64_t m, const double *B, int64_t r, const double *d,
                                  double *out, void *stream);

/* Wide form of the interleaved-ELL product for dense operands with more than 64 columns: the
 * kernel keeps 128 dense columns per lane pair, so the sparse stream is read once per 128 dense
 * columns.  Same layout as tm_csr_dense_sandwich_ell_* with R = tm_ellw_rows() = 64 rows per slab,
 * C = tm_ellw_group_cols() = 16 columns per group, 4 slots per column and iteration
 * (slot it*64 + 4c + u = the (4 it + u)-th nonzero of column c, I = ceil(longest run / 4)) and
 * koff[e] = (k - slab*R) * 128 * sizeof(F).  m must be a multiple of C. */
int tm_ellw_rows(void);
int tm_ellw_group_cols(void);
int tm_csr_dense_sandwich_ellw_f32(const float *vals, const uint32_t *koff, const int64_t *gptr,
                                   int64_t n, int64_t m, const float *B, int64_t r, const float *d,
                                   float *out, void *stream);
int tm_csr_dense_sandwich_ellw_f64(const double *vals, const uint32_t *koff, const int64_t *gptr,
                                   int64_t n, int64_t m, const double *B, int64_t r, const double *d,
                                   double *out, void *stream);

/* Lane-group form of the same unrestricted product (ext/sparse.pyx:211-260 csr_dense_sandwich ->
 * ext/sparse_helpers-tmpl.cpp:23-146) for a C-ordered B with 16-byte aligned rows and more than 64
 * columns: rows in slabs of R = tm_lg_rows() = 64, columns in groups of C = tm_lg_group_cols() = 16
 * (m a multiple of C); column w of a group belongs to wave half h = w / 8 and is the half's column
 * j = w % 8.  A round of a (slab, group) block = 4 chunks x 32 slots, chunk c = columns j = 2c,
 * 2c + 1 of both halves, 8 positions each: slot h*16 + (j&1)*8 + it = the (8*round + it)-th
 * nonzero of column 8h + j of the slab; koff = (1 + row in slab) * tm_lg_row_bytes(sizeof(F))
 * (the row stride of the kernel's LDS slab: 1152 for f64, 512 for f32), 0 = padding (value 0).  vals / koff hold round 0 of block (slab * (m / C) + group) at slot offset block * 128.
 * Entries beyond a column's 8th of a slab are overflow ENTRIES in xkoff (16 bytes each, 16-byte
 * aligned: value (8 bytes; float32 in the first 4), koff, column w = 8h + j of the group), block
 * by block; the block header sits in round 0, chunk 0: koff bits 20..31 of slot 0 = number of
 * entries of the block, bits 20..31 of slots 1..3 = 3 x 12 bits of the index of its first entry.
 * xptr and xvals are not read by the kernel (kept in the signature).  unconditional = 2 or 4
 * positions of every column executed without a test.  out: (m, r), overwritten. */
int tm_lg_rows(void);
int tm_lg_group_cols(void);
int tm_lg_row_bytes(int elem_size);
int tm_csr_dense_sandwich_lg_f32(const float *vals, const uint32_t *koff, const int64_t *xptr,
                                 const float *xvals, const uint32_t *xkoff, int64_t n, int64_t m,
                                 const float *B, int64_t r, const float *d, int unconditional,
                                 float *out, void *stream);
int tm_csr_dense_sandwich_lg_f64(const double *vals, const uint32_t *koff, const int64_t *xptr,
                                 const double *xvals, const uint32_t *xkoff, int64_t n, int64_t m,
                                 const double *B, int64_t r, const double *d, int unconditional,
                                 double *out, void *stream);

/* The same pass, additionally colsum (length m, kernel column order -- the caller applies the same
 * permutation as to the rows of out) = A^T d: `value * d` is formed for every stream slot anyway, so
 * StandardizedMatrix.sandwich needs no second pass over the sparse block (the reference calls
 * transpose_matvec, standardized_mat.py:149-150). */
int tm_csr_dense_sandwich_lg_xtd_f32(const float *vals, const uint32_t *koff, const int64_t *xptr,
                                     const float *xvals, const uint32_t *xkoff, int64_t n, int64_t m,
                                     const float *B, int64_t r, const float *d, int unconditional,
                                     float *out, float *colsum, void *stream);
int tm_csr_dense_sandwich_lg_xtd_f64(const double *vals, const uint32_t *koff, const int64_t *xptr,
                                     const double *xvals, const uint32_t *xkoff, int64_t n, int64_t m,
                                     const double *B, int64_t r, const double *d, int unconditional,
                                     double *out, double *colsum, void *stream);

/* The same product on the COMPACT lane-group stream (round 3: 2.7 instead of 7.7 GB at BASELINE configs[3]):
 * cvals F[real slots + 1]: the values of the real (non-padding) slots of round 0 only, block after block,
 *   inside a block chunk after chunk in slot order;
 * cmap uint8 [S * G][32 slots][4 chunks]: 1 + row in slab of the slot's nonzero, 0 = padding (the kernel
 *   reads one dword per lane: the four chunks of its slot);
 * crec int64 [S * G][2]: {index of the block's first value in cvals, number of overflow entries of the
 *   block | index of its first overflow entry << 32};
 * xkoff: the overflow entries of tm_csr_dense_sandwich_lg_* (unchanged).  colsum (length m, kernel column
 * order, A' d from the same pass) may be NULL. */
int tm_csr_dense_sandwich_lgc_f32(const float *cvals, const uint32_t *cmap, const int64_t *crec,
                                  const uint32_t *xkoff, int64_t n, int64_t m, const float *B, int64_t r,
                                  const float *d, int unconditional, float *out, float *colsum, void *stream);
int tm_csr_dense_sandwich_lgc_f64(const double *cvals, const uint32_t *cmap, const int64_t *crec,
                                  const uint32_t *xkoff, int64_t n, int64_t m, const double *B, int64_t r,
                                  const double *d, int unconditional, double *out, double *colsum,
                                  void *stream);

/* Entry-list form of the same unrestricted product (round 4; ext/sparse.pyx:211-260 csr_dense_sandwich ->
 * ext/sparse_helpers-tmpl.cpp:23-146) for a C-ordered B with 16-byte aligned rows: the accumulator of a
 * nonzero is picked at run time (VGPR index mode), so the stream is a plain list of entries -- one LDS read
 * and two FMAs per nonzero, no padding to a partner column (csrc/sparse_ent.hip).  Rows in slabs of
 * R = tm_ent_rows() = 64, columns in groups of C = tm_ent_group_cols() = 16 (m a multiple of C; kernel
 * column = group * C + column in group), G = m / C groups, S = ceil(n / R) slabs, n < 2^28.  The entries of
 * block (group, slab) are padded to whole batches of tm_ent_batch_slots() = 16 slots (padding: value 0 and
 * the row of a real entry of the block); the blocks of a group follow one another slab after slab, group
 * after group:
 *   vals F[16 * batches + 192]        value (192 slots of slack at the end are read, not used: zeros)
 *   meta uint16[16 * batches + 192]   (slab & 63) << 10 | row in slab << 4 | column in group   (round 6; a uint32
 *                                     row << 4 | column before: 12 -> 10 bytes per slot).  The kernels rebuild a
 *                                     slot's slab from the 6-bit tag and a running slab, so two consecutive batches of
 *                                     a group must lie fewer than 64 slabs apart: the builder gives an EMPTY block one
 *                                     padding batch (value 0, row = the slab's first row) at every 32nd slab.
 *   bstart uint32[G][S + 1]           first batch of block (group, slab); entry S = the end of the group
 * colsum (length m, kernel column order, = A' d from the same pass; reference standardized_mat.py:149-150)
 * may be NULL.  out: (m, r), kernel column order, overwritten. */
int tm_ent_rows(void);
int tm_ent_group_cols(void);
int tm_ent_batch_slots(void);
int tm_csr_dense_sandwich_ent_f32(const float *vals, const uint16_t *meta, const uint32_t *bstart, int64_t n,
                                  int64_t m, const float *B, int64_t r, const float *d, float *out,
                                  float *colsum, void *stream);
int tm_csr_dense_sandwich_ent_f64(const double *vals, const uint16_t *meta, const uint32_t *bstart, int64_t n,
                                  int64_t m, const double *B, int64_t r, const double *d, double *out,
                                  double *colsum, void *stream);

/* ---- Index widths of the sparse entry points.  csr_indices are int32, csr_indptr int64 everywhere below: the
 * reference's fused integral type (ext/sparse.pyx:13-15, `win_integral` = int32 | int64) is NOT mirrored with
 * separate _i64 symbols.  A binder narrows int64 column indices (each is < m, and m fits int32 for every matrix
 * this path can hold in HBM) and widens int32 row pointers -- on upload (CsrDev.from_scipy) or, for arrays that
 * already sit in HBM, with the two device-side conversions below; tests/test_gpu_reference_cases.py loads the golden fixture with both widths. ---- */

/* The two conversions, on the device, for a binder that holds the other width (CsrDev.from_device_arrays):
 *   tm_index_narrow_i64: dst[i] = (int32) src[i]; bad[0] |= 1 when some src[i] lies outside [0, limit)
 *                        (limit <= 2^31; bad is a device int32 the caller zeroed);
 *   tm_index_widen_i32:  dst[i] = (int64) src[i]. */
int tm_index_narrow_i64(const int64_t *src, int64_t count, int64_t limit, int32_t *dst, int32_t *bad, void *stream);
int tm_index_widen_i32(const int32_t *src, int64_t count, int64_t *dst, void *stream);

/* out[Ci] += sum_{j in cols} X[rows[Ci], j] * v[j]      (CSR twin; v length m).
 * Replaces csr_matvec_unrestricted / csr_matvec (ext/sparse.pyx:79-140). */
int tm_csr_matvec_f32(const float *csr_data, const int32_t *csr_indices,
                      const int64_t *csr_indptr, int64_t n, int64_t m, const float *v,
                      const int32_t *rows, int64_t n_rows, const int32_t *cols, int64_t n_cols,
                      float *out, void *stream);
int tm_csr_matvec_f64(const double *csr_data, const int32_t *csr_indices,
                      const int64_t *csr_indptr, int64_t n, int64_t m, const double *v,
                      const int32_t *rows, int64_t n_rows, const int32_t *cols, int64_t n_cols,
                      double *out, void *stream);

/* The unrestricted form (all rows, all columns) on a 16-BIT twin of the column indices (blocks of at most 65536
 * columns; csr_indices16[k] = (uint16_t)csr_indices[k]): 10 instead of 12 bytes per entry leave HBM.  The
 * coefficient vector is staged in LDS: needs sizeof(F) * (m + 4096) <= 64 KB. */
int tm_csr_matvec_u16_f32(const float *csr_data, const uint16_t *csr_indices16, const int64_t *csr_indptr,
                          int64_t n, int64_t m, const float *v, float *out, void *stream);
int tm_csr_matvec_u16_f64(const double *csr_data, const uint16_t *csr_indices16, const int64_t *csr_indptr,
                          int64_t n, int64_t m, const double *v, double *out, void *stream);

/* out[j] += sum_i v[i] * X[i, j] (all rows, all columns) on the same 16-bit twin; values and 16-bit columns must
 * start at entries of the same parity (both freshly allocated arrays do); accumulators in LDS:
 * 8 * (m + 1) + sizeof(F) * 4096 <= 128 KB. */
int tm_csr_rmatvec_u16_f32(const float *csr_data, const uint16_t *csr_indices16, const int64_t *csr_indptr,
                           int64_t n, int64_t m, const float *v, float *out, void *stream);
int tm_csr_rmatvec_u16_f64(const double *csr_data, const uint16_t *csr_indices16, const int64_t *csr_indptr,
                           int64_t n, int64_t m, const double *v, double *out, void *stream);

/* out[Cj] += sum_{i in rows} X[i, cols[Cj]] * v[i]      (v length n; out length n_cols).
 * Replaces csc_rmatvec_unrestricted / csc_rmatvec (ext/sparse.pyx:142-199).  The reference
 * walks CSC columns; on the GPU the CSR twin is streamed row by row (the order rows are
 * sharded in) and the <= few-thousand output columns are LDS-privatised bins. */
int tm_csr_rmatvec_f32(const float *csr_data, const int32_t *csr_indices,
                       const int64_t *csr_indptr, int64_t n, int64_t m, const float *v,
                       const int32_t *rows, int64_t n_rows, const int32_t *cols, int64_t n_cols,
                       float *out, void *stream);
int tm_csr_rmatvec_f64(const double *csr_data, const int32_t *csr_indices,
                       const int64_t *csr_indptr, int64_t n, int64_t m, const double *v,
                       const int32_t *rows, int64_t n_rows, const int32_t *cols, int64_t n_cols,
                       double *out, void *stream);

/* out[j] += sum_i w[i] * X[i,j]^2 on the CSR twin.  Replaces transpose_square_dot_weights
 * (ext/sparse.pyx:262-282). */
int tm_csr_col_sq_f32(const float *csr_data, const int32_t *csr_indices, const int64_t *csr_indptr,
                      int64_t n, int64_t m, const float *w, float *out, void *stream);
int tm_csr_col_sq_f64(const double *csr_data, const int32_t *csr_indices,
                      const int64_t *csr_indptr, int64_t n, int64_t m, const double *w,
                      double *out, void *stream);

/* int64 COLUMN-INDEX forms of the sparse entry points (ext/sparse.pyx:13-15 `win_integral`: the reference's Cython
 * functions take int32 or int64 index arrays).  Every kernel reads int32 column indices; a binding that holds int64
 * ones should narrow them once per block (tm_index_narrow_i64) and call the int32 symbols.  These forms do it per
 * call, on the device, into a scratch buffer the library owns per (device, stream) -- one extra pass over the index
 * array -- and then run the int32 kernel; nnz = number of stored entries (= csr_indptr[n]).  A column index outside
 * [0, m) is clamped to 0 and remembered: tm_index_check_i64 reports it (synchronises the stream, clears the flag). */
int tm_index_check_i64(void *stream, int32_t *h_bad);
int tm_sparse_sandwich_i64_f32(const float *csr_data, const int64_t *csr_indices, const int64_t *csr_indptr,
                               int64_t n, int64_t m, int64_t nnz, const float *d, const int32_t *rows,
                               int64_t n_rows, const int32_t *cols, int64_t n_cols, float *out, void *stream);
int tm_sparse_sandwich_i64_f64(const double *csr_data, const int64_t *csr_indices, const int64_t *csr_indptr,
                               int64_t n, int64_t m, int64_t nnz, const double *d, const int32_t *rows,
                               int64_t n_rows, const int32_t *cols, int64_t n_cols, double *out, void *stream);
int tm_csr_dense_sandwich_i64_f32(const float *csr_data, const int64_t *csr_indices, const int64_t *csr_indptr,
                                  int64_t n, int64_t m, int64_t nnz, const float *B, int64_t r, int order_f,
                                  const float *d, const int32_t *rows, int64_t n_rows, const int32_t *A_cols,
                                  int64_t nA, const int32_t *B_cols, int64_t nB, float *out, void *stream);
int tm_csr_dense_sandwich_i64_f64(const double *csr_data, const int64_t *csr_indices, const int64_t *csr_indptr,
                                  int64_t n, int64_t m, int64_t nnz, const double *B, int64_t r, int order_f,
                                  const double *d, const int32_t *rows, int64_t n_rows, const int32_t *A_cols,
                                  int64_t nA, const int32_t *B_cols, int64_t nB, double *out, void *stream);
int tm_csr_matvec_i64_f32(const float *csr_data, const int64_t *csr_indices, const int64_t *csr_indptr, int64_t n,
                          int64_t m, int64_t nnz, const float *v, const int32_t *rows, int64_t n_rows,
                          const int32_t *cols, int64_t n_cols, float *out, void *stream);
int tm_csr_matvec_i64_f64(const double *csr_data, const int64_t *csr_indices, const int64_t *csr_indptr, int64_t n,
                          int64_t m, int64_t nnz, const double *v, const int32_t *rows, int64_t n_rows,
                          const int32_t *cols, int64_t n_cols, double *out, void *stream);
int tm_csr_rmatvec_i64_f32(const float *csr_data, const int64_t *csr_indices, const int64_t *csr_indptr, int64_t n,
                           int64_t m, int64_t nnz, const float *v, const int32_t *rows, int64_t n_rows,
                           const int32_t *cols, int64_t n_cols, float *out, void *stream);
int tm_csr_rmatvec_i64_f64(const double *csr_data, const int64_t *csr_indices, const int64_t *csr_indptr, int64_t n,
                           int64_t m, int64_t nnz, const double *v, const int32_t *rows, int64_t n_rows,
                           const int32_t *cols, int64_t n_cols, double *out, void *stream);
int tm_csr_col_sq_i64_f32(const float *csr_data, const int64_t *csr_indices, const int64_t *csr_indptr, int64_t n,
                          int64_t m, int64_t nnz, const float *w, float *out, void *stream);
int tm_csr_col_sq_i64_f64(const double *csr_data, const int64_t *csr_indices, const int64_t *csr_indptr, int64_t n,
                          int64_t m, int64_t nnz, const double *w, double *out, void *stream);

/* =====================================================================================
 * Categorical block  (reference: ext/categorical.pyx, ext/split.pyx,
 * ext/cat_split_helpers-tmpl.cpp).  codes[n] int32: -1 = missing (contributes nothing);
 * with drop_first the column of code c is c-1 and code 0 contributes nothing
 * (cat_split_helpers-tmpl.cpp:24-28,68-81,127-129).  n_cols = #categories - drop_first.
 * ===================================================================================== */

/* out[c] += sum_{i in rows, col(i)==c, c in cols} v[i]; out has length n_cols (full block
 * width); with cols != NULL only the listed entries are touched.
 * Replaces _transpose_matvec_all_rows_{fast,complex} (cat_split_helpers-tmpl.cpp:4-41) and the
 * restricted loops of transpose_matvec_{fast,complex} (ext/categorical.pyx:23-117); also
 * sandwich_categorical_{fast,complex} (ext/categorical.pyx:183-218) with v = d. */
int tm_cat_transpose_matvec_f32(const int32_t *codes, int64_t n, int64_t n_cols, int drop_first,
                                const float *v, const int32_t *rows, int64_t n_rows,
                                const int32_t *cols, int64_t n_cols_sel, float *out,
                                void *stream);
int tm_cat_transpose_matvec_f64(const int32_t *codes, int64_t n, int64_t n_cols, int drop_first,
                                const double *v, const int32_t *rows, int64_t n_rows,
                                const int32_t *cols, int64_t n_cols_sel, double *out,
                                void *stream);

/* out[i] += v[col(i)] for every row i whose column is in cols (v has length n_cols).
 * Replaces matvec_{fast,complex} (ext/categorical.pyx:128-180). */
int tm_cat_matvec_f32(const int32_t *codes, int64_t n, int64_t n_cols, int drop_first,
                      const float *v, const int32_t *cols, int64_t n_cols_sel, float *out,
                      void *stream);
int tm_cat_matvec_f64(const int32_t *codes, int64_t n, int64_t n_cols, int drop_first,
                      const double *v, const int32_t *cols, int64_t n_cols_sel, double *out,
                      void *stream);

/* The same into FRESH storage: out[i] = v[col(i)] for the rows whose column is selected, 0 for every other row
 * (out need not be initialised; no read of out).  What CategoricalMatrix.matvec computes when the caller passes
 * no `out` (categorical_matrix.py:495-541 allocates zeros and adds). */
int tm_cat_matvec_assign_f32(const int32_t *codes, int64_t n, int64_t n_cols, int drop_first,
                             const float *v, const int32_t *cols, int64_t n_cols_sel, float *out,
                             void *stream);
int tm_cat_matvec_assign_f64(const int32_t *codes, int64_t n, int64_t n_cols, int drop_first,
                             const double *v, const int32_t *cols, int64_t n_cols_sel, double *out,
                             void *stream);

/* out[i_ncol x j_ncol] (row-major): out[col_i(k), col_j(k)] = sum_{k in rows} d[k].
 * Replaces _sandwich_cat_cat_{fast,complex} (cat_split_helpers-tmpl.cpp:44-94) as bound by
 * sandwich_cat_cat (ext/split.pyx:83-111). */
int tm_cat_cat_sandwich_f32(const int32_t *i_codes, const int32_t *j_codes, int64_t n,
                            const float *d, const int32_t *rows, int64_t n_rows, int64_t i_ncol,
                            int64_t j_ncol, int i_drop_first, int j_drop_first, float *out,
                            void *stream);
int tm_cat_cat_sandwich_f64(const int32_t *i_codes, const int32_t *j_codes, int64_t n,
                            const double *d, const int32_t *rows, int64_t n_rows, int64_t i_ncol,
                            int64_t j_ncol, int i_drop_first, int j_drop_first, double *out,
                            void *stream);

/* The same table with GLOBAL atomics as soon as the LDS-tiled form would need more than 12 passes over
 * the codes (a table of more than ~12 x 128 KB: 1000 x 1000 levels in f64 takes 63 passes, 0.16 ms for
 * 1M rows against 0.05 ms here).  The caller vouches that no cell collects more than a few thousand
 * rows (atomics on one address serialise); the host layer checks the level counts
 * (CategoricalMatrix._hot_count). */
int tm_cat_cat_sandwich_atomic_f32(const int32_t *i_codes, const int32_t *j_codes, int64_t n,
                                   const float *d, const int32_t *rows, int64_t n_rows, int64_t i_ncol,
                                   int64_t j_ncol, int i_drop_first, int j_drop_first, float *out,
                                   void *stream);
int tm_cat_cat_sandwich_atomic_f64(const int32_t *i_codes, const int32_t *j_codes, int64_t n,
                                   const double *d, const int32_t *rows, int64_t n_rows, int64_t i_ncol,
                                   int64_t j_ncol, int i_drop_first, int j_drop_first, double *out,
                                   void *stream);

/* The same table from rows GROUPED BY the level of categorical i (static per pair, built by the host layer
 * once): perm[p] = row of position p (rows without a column of i left out), ci_sorted[p] / cj_sorted[p] = the
 * column indices of that row (drop_first applied; cj_sorted < 0: the row has no column of j), lptr[c] = first
 * position of level c (i_ncol + 1 entries).  out[i_ncol][j_ncol] is overwritten with
 * sum over p of d[perm[p]] at (ci_sorted[p], cj_sorted[p]).  One pass over the rows whatever the size of the
 * table (each LDS tile of levels reads only its own positions); needs j_ncol * 8 <= 128 KB.  A row restriction
 * is a d masked with zeros.  Reference: ext/split.pyx:83-111. */
int tm_cat_cat_sandwich_sorted_f32(const int32_t *ci_sorted, const int32_t *cj_sorted, const int32_t *perm,
                                   const int64_t *lptr, int64_t n_sorted, const float *d, int64_t i_ncol,
                                   int64_t j_ncol, float *out, void *stream);
int tm_cat_cat_sandwich_sorted_f64(const int32_t *ci_sorted, const int32_t *cj_sorted, const int32_t *perm,
                                   const int64_t *lptr, int64_t n_sorted, const double *d, int64_t i_ncol,
                                   int64_t j_ncol, double *out, void *stream);

/* out[i_ncol x n_j] (row-major): out[col(k), jc] = sum_{k in rows} d[k] * M[k, j_cols[jc]].
 * Replaces _sandwich_cat_dense{C,F}_{fast,complex} (cat_split_helpers-tmpl.cpp:97-151) as bound
 * by sandwich_cat_dense (ext/split.pyx:32-80). */
int tm_cat_dense_sandwich_f32(const int32_t *codes, int64_t n, int64_t i_ncol, int drop_first,
                              const float *d, const int32_t *rows, int64_t n_rows, const float *M,
                              int64_t M_ncol, int order_f, const int32_t *j_cols, int64_t n_j,
                              float *out, void *stream);
int tm_cat_dense_sandwich_f64(const int32_t *codes, int64_t n, int64_t i_ncol, int drop_first,
                              const double *d, const int32_t *rows, int64_t n_rows,
                              const double *M, int64_t M_ncol, int order_f, const int32_t *j_cols,
                              int64_t n_j, double *out, void *stream);

/* out[i_ncol x n_cols] (row-major): out[col(k), Cj] = sum_{k in rows} d[k] * S[k, cols[Cj]],
 * S sparse given by its CSR twin.  The reference has no kernel here: CategoricalMatrix.
 * _cross_sparse (categorical_matrix.py:825-838) builds a scipy CSR of diag(d)*onehot and calls
 * scipy.sparse matmul; this entry point is that product. */
int tm_cat_sparse_sandwich_f32(const int32_t *codes, int64_t n, int64_t i_ncol, int drop_first,
                               const float *csr_data, const int32_t *csr_indices,
                               const int64_t *csr_indptr, int64_t s_ncol, const float *d,
                               const int32_t *rows, int64_t n_rows, const int32_t *cols,
                               int64_t n_cols, float *out, void *stream);
int tm_cat_sparse_sandwich_f64(const int32_t *codes, int64_t n, int64_t i_ncol, int drop_first,
                               const double *csr_data, const int32_t *csr_indices,
                               const int64_t *csr_indptr, int64_t s_ncol, const double *d,
                               const int32_t *rows, int64_t n_rows, const int32_t *cols,
                               int64_t n_cols, double *out, void *stream);

/* Fused cross terms of ALL categorical blocks of a SplitMatrix with its dense block (or with
 * its sparse block in slab form): one pass over the wide operand serves every categorical, so it
 * is read from HBM once instead of once per categorical.  Replaces n_cats calls of
 * sandwich_cat_dense (ext/split.pyx:32-80) resp. of the scipy product of
 * categorical_matrix.py:825-838 made by the loop of split_matrix.py:346-354.
 * h_codes / h_ncols / h_drop_first are HOST arrays of length n_cats (<= 16) holding the device
 * pointer of each code vector, its number of columns and its drop_first flag.  All rows, all
 * columns (restrictions: masked d + sub-selection on the host side).
 * out is [sum(h_ncols) x M_ncol] (resp. x m) row-major: the blocks of the categoricals stacked. */
int tm_multi_cat_dense_sandwich_f32(const void *const *h_codes, const int64_t *h_ncols,
                                    const int32_t *h_drop_first, int n_cats, int64_t n,
                                    const float *d, const float *M, int64_t M_ncol, int order_f,
                                    float *out, void *stream);
int tm_multi_cat_dense_sandwich_f64(const void *const *h_codes, const int64_t *h_ncols,
                                    const int32_t *h_drop_first, int n_cats, int64_t n,
                                    const double *d, const double *M, int64_t M_ncol, int order_f,
                                    double *out, void *stream);
/* The same with a row list (C-ordered M with 16-byte aligned rows, <= 4 categoricals, levels
 * within one LDS tile -- else TM_EUNSUPPORTED): only rows[0 .. n_rows) of M, d and the codes are
 * read, cost proportional to n_rows (the `for k in rows` of ext/split.pyx:32-80). */
int tm_multi_cat_dense_sandwich_rows_f32(const void *const *h_codes, const int64_t *h_ncols,
                                         const int32_t *h_drop_first, int n_cats, int64_t n,
                                         const float *d, const float *M, int64_t M_ncol,
                                         const int32_t *rows, int64_t n_rows, float *out,
                                         void *stream);
int tm_multi_cat_dense_sandwich_rows_f64(const void *const *h_codes, const int64_t *h_ncols,
                                         const int32_t *h_drop_first, int n_cats, int64_t n,
                                         const double *d, const double *M, int64_t M_ncol,
                                         const int32_t *rows, int64_t n_rows, double *out,
                                         void *stream);
/* Row-list form of the fused categorical x sparse cross terms: cost proportional to n_sel (the
 * reference works on self[rows], categorical_matrix.py:825-838).  cm_data / cm_indices: the
 * chunk-major twin of tm_sparse_sandwich_chunked_*; row_ranges [n_chunks][n_sel][2] = {start, end} of
 * every selected row in every 128-column chunk; rows [n_sel]: the selected ORIGINAL row ids (index
 * the codes); d_sel [n_sel]: their weights.  out: stacked [sum(n_cols), m], overwritten. */
int tm_multi_cat_sparse_sandwich_rows_f32(const void *const *h_codes, const int64_t *h_ncols,
                                          const int32_t *h_drop_first, int n_cats,
                                          const float *cm_data, const int32_t *cm_indices,
                                          const int32_t *row_ranges, const int32_t *rows,
                                          int64_t n_sel, int64_t m, const float *d_sel, float *out,
                                          void *stream);
int tm_multi_cat_sparse_sandwich_rows_f64(const void *const *h_codes, const int64_t *h_ncols,
                                          const int32_t *h_drop_first, int n_cats,
                                          const double *cm_data, const int32_t *cm_indices,
                                          const int32_t *row_ranges, const int32_t *rows,
                                          int64_t n_sel, int64_t m, const double *d_sel, double *out,
                                          void *stream);
/* The same with the columns of the chunk-major entries as ONE BYTE each (cm_col8: the column inside its 128-column
 * chunk, the array tm_sparse_sandwich_blocks_u8_* streams): since round 6 the host side keeps no int32 copy of the
 * chunk-major columns (-1.0 GB at BASELINE configs[3]). */
int tm_multi_cat_sparse_sandwich_rows_u8_f32(const void *const *h_codes, const int64_t *h_ncols,
                                             const int32_t *h_drop_first, int n_cats,
                                             const float *cm_data, const uint8_t *cm_col8,
                                             const int32_t *row_ranges, const int32_t *rows,
                                             int64_t n_sel, int64_t m, const float *d_sel, float *out,
                                             void *stream);
int tm_multi_cat_sparse_sandwich_rows_u8_f64(const void *const *h_codes, const int64_t *h_ncols,
                                             const int32_t *h_drop_first, int n_cats,
                                             const double *cm_data, const uint8_t *cm_col8,
                                             const int32_t *row_ranges, const int32_t *rows,
                                             int64_t n_sel, int64_t m, const double *d_sel, double *out,
                                             void *stream);
/* The same fused categorical x sparse cross terms on the ENTRY twin of the sparse block (round 4; layout at
 * tm_csr_dense_sandwich_ent_*): no slab-form twin is needed for a block whose sparse x dense term runs on the entry
 * kernel.  mk = 16 * groups kernel columns; out [sum(n_cols)][mk] in kernel column order (the caller applies the
 * twin's column permutation), overwritten.  Replaces CategoricalMatrix._cross_sparse
 * (categorical_matrix.py:825-838, a scipy.sparse product) for all categoricals of a SplitMatrix at once.
 * n_slots (round 6): slots of the whole stream (16 x batches; 0 = unknown).  Two kernels sit behind the entry point:
 * where a (group, slab) block holds at least 44 slots on average (n_slots / (groups x slabs)) and the LDS holds tile +
 * staging, the rows' operands are fetched per slab and parked in LDS (multi_cat_sparse_ent_staged_kernel); else they
 * are gathered per slot. */
int tm_multi_cat_sparse_sandwich_ent_f32(const void *const *h_codes, const int64_t *h_ncols,
                                         const int32_t *h_drop_first, int n_cats, int64_t n, const float *d,
                                         const float *vals, const uint16_t *meta, const uint32_t *bstart,
                                         int64_t n_slots, int64_t mk, float *out, void *stream);
int tm_multi_cat_sparse_sandwich_ent_f64(const void *const *h_codes, const int64_t *h_ncols,
                                         const int32_t *h_drop_first, int n_cats, int64_t n, const double *d,
                                         const double *vals, const uint16_t *meta, const uint32_t *bstart,
                                         int64_t n_slots, int64_t mk, double *out, void *stream);
/* The same with the categoricals' codes PACKED (round 5; at most 3 categoricals, 1022 stacked levels):
 * packed[row] = sum_c field_c << (10 c), field_c = the stacked output row of the row's level in categorical c
 * (offset of c + code - drop_first) or 1023 when the row has none there.  tm_multi_cat_pack_codes builds the
 * vector once per set of categoricals (it depends on the codes only); the kernel then gathers ONE word per entry
 * slot instead of one code per categorical. */
int tm_multi_cat_pack_codes(const void *const *h_codes, const int64_t *h_ncols, const int32_t *h_drop_first, int n_cats,
                            int64_t n, uint32_t *packed, void *stream);
int tm_multi_cat_sparse_sandwich_entp_f32(const void *const *h_codes, const int64_t *h_ncols,
                                          const int32_t *h_drop_first, int n_cats, int64_t n, const float *d,
                                          const float *vals, const uint16_t *meta, const uint32_t *bstart,
                                          int64_t n_slots, int64_t mk, const uint32_t *packed, float *out, void *stream);
int tm_multi_cat_sparse_sandwich_entp_f64(const void *const *h_codes, const int64_t *h_ncols,
                                          const int32_t *h_drop_first, int n_cats, int64_t n, const double *d,
                                          const double *vals, const uint16_t *meta, const uint32_t *bstart,
                                          int64_t n_slots, int64_t mk, const uint32_t *packed, double *out, void *stream);

/* ecol[e] = column of entry e within its column group (0 .. tm_slab_group_cols()-1). */
int tm_multi_cat_sparse_sandwich_slab_f32(const void *const *h_codes, const int64_t *h_ncols,
                                          const int32_t *h_drop_first, int n_cats, int64_t n,
                                          const float *d, const float *vals, const uint32_t *koff,
                                          const uint8_t *ecol, const int64_t *gptr, int64_t m,
                                          float *out, void *stream);
int tm_multi_cat_sparse_sandwich_slab_f64(const void *const *h_codes, const int64_t *h_ncols,
                                          const int32_t *h_drop_first, int n_cats, int64_t n,
                                          const double *d, const double *vals, const uint32_t *koff,
                                          const uint8_t *ecol, const int64_t *gptr, int64_t m,
                                          double *out, void *stream);

/* =====================================================================================
 * ALL categorical x categorical blocks (and the categorical diagonals) of one sandwich in one pass
 * over the codes (ext/split.pyx:83-111 sandwich_cat_cat, ext/categorical.pyx:183-218): the pair
 * tables are packed into bundles of at most tm_multi_cat_pairs_max_bins() doubles (one LDS tile).
 * A bundle is a rectangle of tables: lists A and B of at most tm_multi_cat_pairs_max_slots()
 * categoricals, tile = [sum of A's levels] x [sum of B's levels] row-major (width = tile width),
 * table (a, b) the sub-rectangle at (first tile row of a, first tile column of b).
 * cat_tab: device int64 [n categoricals][4] = {device pointer of the int32 codes, code of the first
 * kept level (1 with drop_first, else 0; codes below it -- also -1 = missing -- contribute nothing),
 * first entry of the block's column positions in `pos`, 0}.
 * bundles: device int32, tm_multi_cat_pairs_row_words() words per bundle:
 *   {|A|, |B|, mode, tile width, first workgroup, workgroups, bins of the tile, 0,
 *    A: {index into cat_tab, first tile row x tile width} x max_slots,
 *    B: {index into cat_tab, first tile column} x max_slots}
 *   mode 0: all tables A x B;  1: B = A, tables a <= b ("triangle"; the diagonal of table (a, a) is
 *   the categorical's own diagonal);  2: diagonals only, tile width 1, one bin per level.
 * wg_map: device int32 [n_wg]: the bundle every workgroup works for (a bundle's workgroups are
 *   consecutive from its "first workgroup"; the row range is cut evenly among them).
 * slots: largest |A| or |B| (selects the kernel).  bins: largest tile (<= max_bins).
 * tables: device double [n_bundles][bins], overwritten with the bundle tiles.  desc: device int64
 * [n_pairs][8] = {offset of the table's first bin in `tables`, L_i, L_j, tile width (diagonal
 * entry: distance between consecutive diagonal bins), first entry of block i's / block j's
 * positions in pos, diagonal flag, 0}; pos: device int64 positions of the blocks' columns in the
 * p x p float64 `out` (out[pos_i[a], pos_j[b]] = table[a, b] and its mirror; a negative position =
 * level not selected, nothing written; out may be NULL: tables only; p = 0: `out` is a VECTOR and
 * only the diagonal entries are written, out[pos_i[a]] = diagonal a).  rows: int32 device row list or NULL.
 * ===================================================================================== */
int tm_multi_cat_pairs_max_bins(void);
int tm_multi_cat_pairs_max_slots(void);
int tm_multi_cat_pairs_row_words(void);
int tm_multi_cat_pairs_f32(const int64_t *cat_tab, int64_t n, const float *d, const int32_t *rows,
                           int64_t n_rows, const int32_t *bundles, int n_bundles,
                           const int32_t *wg_map, int n_wg, int slots, int64_t bins,
                           const int64_t *desc, int64_t n_pairs, const int64_t *pos, double *tables,
                           double *out, int64_t p, void *stream);
int tm_multi_cat_pairs_f64(const int64_t *cat_tab, int64_t n, const double *d, const int32_t *rows,
                           int64_t n_rows, const int32_t *bundles, int n_bundles,
                           const int32_t *wg_map, int n_wg, int slots, int64_t bins,
                           const int64_t *desc, int64_t n_pairs, const int64_t *pos, double *tables,
                           double *out, int64_t p, void *stream);

/* =====================================================================================
 * SplitMatrix.matvec over ALL categorical blocks in one pass (split_matrix.py:373-420 calling
 * ext/categorical.pyx:110-136 block by block):  out[k] += sum_c v[pos[first_c + code_c[k] - kept_c]].
 * cat_tab / pos as for tm_multi_cat_pairs_*; v: the full coefficient vector (device, length = number
 * of columns of the split matrix); out: device, length n, accumulated into.
 * ===================================================================================== */
int tm_multi_cat_matvec_f32(const int64_t *cat_tab, int n_cats, const int64_t *pos, const float *v,
                            int64_t n, float *out, void *stream);
int tm_multi_cat_matvec_f64(const int64_t *cat_tab, int n_cats, const int64_t *pos, const double *v,
                            int64_t n, double *out, void *stream);

/* =====================================================================================
 * A narrow column selection (`cols=`, a solver's active set) as one row-major dense block T, on
 * which the unrestricted kernels then run (the reference's restricted loops, ext/dense.pyx:24-54,
 * ext/sparse.pyx:17-77 / 211-260 with `cols`, cost in proportion to the selection):
 *   tm_csr_densify_cols_*: T[r, colmap[c]] += value for every stored entry (r, c) of the CSR block
 *     with colmap[c] >= 0 (colmap: device int32 [m], target column of T or -1); T zeroed by the caller.
 *   tm_csc_densify_cols_*: the same from the CSC form (rows / vals: the entries sorted by column):
 *     seg: device int64 [n_sel][2] = {first entry, end} of every selected column, tcol: device int32
 *     [n_sel] its column of T; max_len: the longest selected column (sizes the launch).
 *   tm_dense_gather_cols_*: T[r, t0 + q] = X[r, cols[q]], X (n, m) C- or F-ordered (order_f).
 * T: device, row stride ld elements.
 * ===================================================================================== */
int tm_csr_densify_cols_f32(const float *data, const int32_t *indices, const int64_t *indptr, int64_t n,
                            const int32_t *colmap, float *T, int64_t ld, void *stream);
int tm_csr_densify_cols_f64(const double *data, const int32_t *indices, const int64_t *indptr, int64_t n,
                            const int32_t *colmap, double *T, int64_t ld, void *stream);
int tm_csc_densify_cols_f32(const int32_t *rows, const float *vals, const int64_t *seg, const int32_t *tcol,
                            int64_t n_sel, int64_t max_len, float *T, int64_t ld, void *stream);
int tm_csc_densify_cols_f64(const int32_t *rows, const double *vals, const int64_t *seg, const int32_t *tcol,
                            int64_t n_sel, int64_t max_len, double *T, int64_t ld, void *stream);
int tm_dense_gather_cols_f32(const float *X, int64_t n, int64_t m, int order_f, const int32_t *cols,
                             int64_t n_sel, float *T, int64_t ld, int64_t t0, void *stream);
int tm_dense_gather_cols_f64(const double *X, int64_t n, int64_t m, int order_f, const int32_t *cols,
                             int64_t n_sel, double *T, int64_t ld, int64_t t0, void *stream);

/* =====================================================================================
 * Assembly helper for SplitMatrix.sandwich (split_matrix.py:336-354): scatter a block
 * result into the float64 p x p output at the block's global column positions,
 *   out[ri[a], ci[b]] = src[a, b]  (and the transpose when mirror != 0);
 * diag != 0: src is a length-nr vector added to out[ri[a], ri[a]].
 * ===================================================================================== */
int tm_scatter_block_f32(const float *src, int64_t nr, int64_t nc, const int64_t *ri,
                         const int64_t *ci, double *out, int64_t p, int mirror, int diag,
                         void *stream);
int tm_scatter_block_f64(const double *src, int64_t nr, int64_t nc, const int64_t *ri,
                         const int64_t *ci, double *out, int64_t p, int mirror, int diag,
                         void *stream);

/* =====================================================================================
 * Row-restricted fast paths at a cost proportional to the number of selected rows (the reference's
 * `for k in rows` loops: ext/sparse.pyx:46-48, ext/sparse_helpers-tmpl.cpp:67-131).  Both work on
 * the chunk-major twin of tm_sparse_sandwich_chunked_* plus a table row_ranges[n_chunks][n_sel][2]
 * = {start, end} of every SELECTED row (ascending row order) inside every 128-column chunk, and
 * d_sel[n_sel] = d of those rows; rows[n_sel] = their row numbers (for the dense operand).
 *   tm_sparse_sandwich_chunked_rows_*: out (m, m)  = A[rows]' diag(d[rows]) A[rows]
 *   tm_csr_dense_sandwich_rows_*:      out (m, r)  = A[rows]' diag(d[rows]) B[rows]   (B C- or F-ordered)
 * Results overwrite out.
 * ===================================================================================== */
int tm_sparse_sandwich_chunked_rows_f32(const float *cm_data, const int32_t *cm_indices,
                                        const int32_t *row_ranges, int64_t n_sel, int64_t m,
                                        int64_t nnz, const float *d_sel, float *out, void *stream);
int tm_sparse_sandwich_chunked_rows_f64(const double *cm_data, const int32_t *cm_indices,
                                        const int32_t *row_ranges, int64_t n_sel, int64_t m,
                                        int64_t nnz, const double *d_sel, double *out, void *stream);

/* Both forms with the columns as ONE BYTE per chunk-major entry (cm_col8[p] = cm_indices[p] % 128, the column inside
 * the entry's chunk): the entry gathers of the kernel move a quarter of the index bytes. */
int tm_sparse_sandwich_chunked_u8_f32(const float *cm_data, const uint8_t *cm_col8, const int32_t *cptr, int64_t n,
                                      int64_t m, int64_t nnz, const float *d, float *out, void *stream);
int tm_sparse_sandwich_chunked_u8_f64(const double *cm_data, const uint8_t *cm_col8, const int32_t *cptr, int64_t n,
                                      int64_t m, int64_t nnz, const double *d, double *out, void *stream);
int tm_sparse_sandwich_chunked_rows_u8_f32(const float *cm_data, const uint8_t *cm_col8, const int32_t *row_ranges,
                                           int64_t n_sel, int64_t m, int64_t nnz, const float *d_sel, float *out,
                                           void *stream);
int tm_sparse_sandwich_chunked_rows_u8_f64(const double *cm_data, const uint8_t *cm_col8, const int32_t *row_ranges,
                                           int64_t n_sel, int64_t m, int64_t nnz, const double *d_sel, double *out,
                                           void *stream);
/* The same product for WIDE, VERY SPARSE blocks (fewer than one nonzero per row and 128-column
 * chunk): plain CSR in, one L2 atomic per pair into a double accumulator, cost proportional to the
 * number of pairs instead of rows x tiles (csrc/sparse_direct.hip).  All rows (a row restriction is
 * a masked d: rows with d == 0 are not read).  out: (m, m), both triangles, overwritten. */
int tm_sparse_sandwich_direct_f32(const float *csr_data, const int32_t *csr_indices,
                                  const int64_t *csr_indptr, int64_t n, int64_t m, const float *d,
                                  float *out, void *stream);
int tm_sparse_sandwich_direct_f64(const double *csr_data, const int32_t *csr_indices,
                                  const int64_t *csr_indptr, int64_t n, int64_t m, const double *d,
                                  double *out, void *stream);
int tm_csr_dense_sandwich_rows_f32(const float *cm_data, const int32_t *cm_indices,
                                   const int32_t *row_ranges, int64_t n_sel, const int32_t *rows,
                                   const float *d_sel, int64_t n, int64_t m, const float *B, int64_t r,
                                   int order_f, float *out, void *stream);
int tm_csr_dense_sandwich_rows_f64(const double *cm_data, const int32_t *cm_indices,
                                   const int32_t *row_ranges, int64_t n_sel, const int32_t *rows,
                                   const double *d_sel, int64_t n, int64_t m, const double *B, int64_t r,
                                   int order_f, double *out, void *stream);
/* (byte columns inside the 128-column chunk instead of int32 block columns: see tm_multi_cat_sparse_sandwich_rows_u8_*;
 * reference loop: ext/sparse_helpers-tmpl.cpp:67-131) */
int tm_csr_dense_sandwich_rows_u8_f32(const float *cm_data, const uint8_t *cm_col8,
                                      const int32_t *row_ranges, int64_t n_sel, const int32_t *rows,
                                      const float *d_sel, int64_t n, int64_t m, const float *B, int64_t r,
                                      int order_f, float *out, void *stream);
int tm_csr_dense_sandwich_rows_u8_f64(const double *cm_data, const uint8_t *cm_col8,
                                      const int32_t *row_ranges, int64_t n_sel, const int32_t *rows,
                                      const double *d_sel, int64_t n, int64_t m, const double *B, int64_t r,
                                      int order_f, double *out, void *stream);

/* =====================================================================================
 * Deterministic categorical transpose_matvec / sandwich diagonal (ext/categorical.pyx:23-42 with
 * the thread-ordered reduction of ext/cat_split_helpers-tmpl.cpp:33-38: the reference's K4a is
 * bitwise reproducible).  perm: the rows that fall into a column, grouped by column (stable by
 * row); every column's run is cut into blocks of tm_cat_det_block_rows() rows: bstart[n_blocks + 1]
 * = block limits inside perm, cat_bptr[n_cols + 1] = first block of every column.
 * out[c] (+)= sum over the column's rows of v[row], each block summed in a fixed order in double,
 * the block sums added in order.  A row restriction is expressed through v (0 outside).
 * ===================================================================================== */
int tm_cat_det_block_rows(void);
int tm_cat_transpose_matvec_det_f32(const int32_t *perm, const int64_t *bstart, int64_t n_blocks,
                                    const int64_t *cat_bptr, int64_t n_cols, const float *v,
                                    float *out, int accumulate, void *stream);
int tm_cat_transpose_matvec_det_f64(const int32_t *perm, const int64_t *bstart, int64_t n_blocks,
                                    const int64_t *cat_bptr, int64_t n_cols, const double *v,
                                    double *out, int accumulate, void *stream);

/* Categorical cross terms on the SAME row grouping, for categoricals with many levels
 * (ext/split.pyx:32-80 sandwich_cat_dense; the scipy product of categorical_matrix.py:825-838):
 * out (n_cols, m) = C^T diag(d) Y with C the one-hot block described by perm / bstart / cat_bptr as
 * above.  A workgroup sums d[k] * Y[k, :] over the rows of one block (one level), the blocks of a
 * level are added in order: cost independent of the number of levels, results reproducible from run
 * to run.  Dense Y: C-ordered (m columns, row stride m), 16-byte aligned, m a multiple of
 * 16 / sizeof(F).  Sparse Y: CSR.  Rows with d == 0 are not read.  out: overwritten. */
int tm_cat_dense_sandwich_sorted_f32(const int32_t *perm, const int64_t *bstart, int64_t n_blocks,
                                     const int64_t *cat_bptr, int64_t n_cols, const float *d,
                                     const float *Y, int64_t m, float *out, void *stream);
int tm_cat_dense_sandwich_sorted_f64(const int32_t *perm, const int64_t *bstart, int64_t n_blocks,
                                     const int64_t *cat_bptr, int64_t n_cols, const double *d,
                                     const double *Y, int64_t m, double *out, void *stream);
/* The dense-operand kernel above with a VALUE per entry: the sparse x dense term
 * A^T diag(d) Y (ext/sparse.pyx:211-260 csr_dense_sandwich) on the CSC form of A, column by column
 * -- for blocks with only a few nonzeros per row (wide and very sparse), where the slab / gather
 * kernels pay per (slab, column group).  rows / vals: CSC row indices and values; the entries of a
 * column cut into blocks of at most tm_cat_det_block_rows(): bstart (n_blocks + 1), blocks of column
 * j = col_bptr[j] .. col_bptr[j + 1].  out: (n_cols, m), overwritten. */
int tm_csc_dense_sandwich_sorted_f32(const int32_t *rows, const float *vals, const int64_t *bstart,
                                     int64_t n_blocks, const int64_t *col_bptr, int64_t n_cols,
                                     const float *d, const float *Y, int64_t m, float *out,
                                     void *stream);
int tm_csc_dense_sandwich_sorted_f64(const int32_t *rows, const double *vals, const int64_t *bstart,
                                     int64_t n_blocks, const int64_t *col_bptr, int64_t n_cols,
                                     const double *d, const double *Y, int64_t m, double *out,
                                     void *stream);
int tm_cat_sparse_sandwich_sorted_f32(const int32_t *perm, const int64_t *bstart, int64_t n_blocks,
                                      const int64_t *cat_bptr, int64_t n_cols, const float *d,
                                      const float *csr_data, const int32_t *csr_indices,
                                      const int64_t *csr_indptr, int64_t m, float *out, void *stream);
int tm_cat_sparse_sandwich_sorted_f64(const int32_t *perm, const int64_t *bstart, int64_t n_blocks,
                                      const int64_t *cat_bptr, int64_t n_cols, const double *d,
                                      const double *csr_data, const int32_t *csr_indices,
                                      const int64_t *csr_indptr, int64_t m, double *out, void *stream);

/* =====================================================================================
 * Multi-right-hand-side matvec / transpose-matvec (2-D `vec`).  The reference hands these to
 * scipy.sparse (sparse_matrix.py:252-254, 266-268) and NumPy BLAS (dense_matrix.py:212-217 with
 * a 2-D operand).  V: (m, K) for matvec, (n, K) for rmatvec, row-major; out: (n_rows, K) resp.
 * (n_cols, K) row-major, ACCUMULATED into.  rows / cols as in the 1-D entry points
 * (NULL = all); for the matvec forms `cols` masks the columns that take part (V keeps m rows).
 * ===================================================================================== */
int tm_csr_matvec_multi_f32(const float *csr_data, const int32_t *csr_indices,
                            const int64_t *csr_indptr, int64_t n, int64_t m, const float *V,
                            int64_t K, const int32_t *rows, int64_t n_rows, const int32_t *cols,
                            int64_t n_cols, float *out, void *stream);
int tm_csr_rmatvec_multi_f32(const float *csr_data, const int32_t *csr_indices,
                             const int64_t *csr_indptr, int64_t n, int64_t m, const float *V,
                             int64_t K, const int32_t *rows, int64_t n_rows, const int32_t *cols,
                             int64_t n_cols, float *out, void *stream);
int tm_dense_matvec_multi_f32(const float *X, int64_t n, int64_t m, int order_f, const float *V,
                              int64_t K, const int32_t *rows, int64_t n_rows, const int32_t *cols,
                              int64_t n_cols, float *out, void *stream);
int tm_dense_rmatvec_multi_f32(const float *X, int64_t n, int64_t m, int order_f, const float *V,
                               int64_t K, const int32_t *rows, int64_t n_rows, const int32_t *cols,
                               int64_t n_cols, float *out, void *stream);
int tm_csr_matvec_multi_f64(const double *csr_data, const int32_t *csr_indices,
                            const int64_t *csr_indptr, int64_t n, int64_t m, const double *V,
                            int64_t K, const int32_t *rows, int64_t n_rows, const int32_t *cols,
                            int64_t n_cols, double *out, void *stream);
int tm_csr_rmatvec_multi_f64(const double *csr_data, const int32_t *csr_indices,
                             const int64_t *csr_indptr, int64_t n, int64_t m, const double *V,
                             int64_t K, const int32_t *rows, int64_t n_rows, const int32_t *cols,
                             int64_t n_cols, double *out, void *stream);
int tm_dense_matvec_multi_f64(const double *X, int64_t n, int64_t m, int order_f, const double *V,
                              int64_t K, const int32_t *rows, int64_t n_rows, const int32_t *cols,
                              int64_t n_cols, double *out, void *stream);
int tm_dense_rmatvec_multi_f64(const double *X, int64_t n, int64_t m, int order_f, const double *V,
                               int64_t K, const int32_t *rows, int64_t n_rows, const int32_t *cols,
                               int64_t n_cols, double *out, void *stream);

/* =====================================================================================
 * StandardizedMatrix on device blocks (standardized_mat.py:123-230).
 * tm_vec_sum_*: out[0] = sum_i v[rows[i]] (rows == NULL: v[0..n)), accumulated in double in a
 *   fixed order (the np.sum(d[rows]) / other.sum(0) of standardized_mat.py:159,215).
 * tm_standardize_sandwich_f64: the rank-one corrections of standardized_mat.py:148-171 in place:
 *   inout[i, j] = inout[i, j] mult[i] mult[j] + m[i] shift[j] + shift[i] m[j] + shift[i] shift[j] S,
 *   m = mult * xtd (xtd = mat' d), S = sum_d[0]; mult == NULL: ones; inner_diag != NULL: the inner
 *   sandwich is diagonal (a categorical block) and given as that length-k vector, inout is
 *   overwritten.  All pointers are device pointers.
 * ===================================================================================== */
int tm_vec_sum_f32(const float *v, const int32_t *rows, int64_t n, double *out, void *stream);
int tm_vec_sum_f64(const double *v, const int32_t *rows, int64_t n, double *out, void *stream);
int tm_standardize_sandwich_f64(double *inout, const double *inner_diag, const double *xtd,
                                const double *shift, const double *mult, const double *sum_d,
                                int64_t k, void *stream);
/* The same result from a PARTLY CENTRED inner product (the _centered_ dense sandwiches above): column i of
 * the inner matrix was taken as x_i - center[i] (0: as it is), xtd[i] = sum_r d_r (x_ri - center[i]);
 * group[i] >= 0 names the block whose self term was computed centred: inout[i][j] holds the centred product
 * where group[i] == group[j] >= 0 and the raw product elsewhere (those entries are centred here with
 * rank-one terms).  inout[i][j] <- S'_ij mult_i mult_j + mult_i xtd_i delta_j + delta_i mult_j xtd_j
 * + delta_i delta_j S,  delta = shift + center * mult (rounding-sized for center = -shift / mult). */
int tm_standardize_sandwich_centered_f64(double *inout, const double *xtd, const double *center,
                                         const int32_t *group, const double *shift, const double *mult,
                                         const double *sum_d, int64_t k, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TABMAT_HIP_H */
