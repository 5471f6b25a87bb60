/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle, not a product path.
 *
 * Type-generic body of the oracle kernels.  This file is included several
 * times by tabmat_oracle.c with
 *     F    floating type (float | double)
 *     FS   suffix for F  (f32 | f64)
 * and, for the sparse kernels only (ORACLE_SPARSE_PASS defined),
 *     I    index type    (int32_t | int64_t)
 *     IS   suffix for I  (i32 | i64)
 *
 * Every function restates, in plain C, the algorithm of one native loop of
 * Quantco/tabmat.  File:line citations are relative to /root/reference/.
 * Outputs follow the reference convention: the caller hands in a ZEROED `out`
 * and the kernel accumulates with += (ext/dense.pyx:25, ext/sparse.pyx:39,235,
 * ext/split.pyx:48,99).
 */

#define CAT2(a, b) a##_##b
#define CAT(a, b) CAT2(a, b)
#ifndef ORACLE_SPARSE_PASS
/* ------------------------------------------------------------------------- */
/* dense / categorical pass: symbols are  orc_<name>_<f32|f64>               */
/* ------------------------------------------------------------------------- */
#define FN(name) CAT(orc_##name, FS)

/*
 * K1  dense sandwich  out[Ci,Cj] = sum_k X[rows[k],cols[Ci]] d[rows[k]] X[rows[k],cols[Cj]]
 * Reference: src/tabmat/ext/dense_helpers-tmpl.cpp:266-311 (_dense{C,F}_sandwich),
 * k_loop 198-263, dense_base 161-196, middle_j 41-143.
 * Restated with the reference's structure: k-blocks of kratio*thresh1d = 512
 * rows (line 201), pack R = d o X (224/229) and L = X (251/256) for the block,
 * accumulate only the lower triangle j <= i (148-151), add the block partial
 * into out (137-139), mirror at the end (302-306).  Parallel over k-blocks
 * with a per-thread partial (the reference's kparallel branch, line 278).
 */
void FN(dense_sandwich)(const F *X, int64_t n, int64_t m, int order_f,
                        const F *d, const int32_t *rows, int64_t in_n,
                        const int32_t *cols, int64_t out_m, F *out)
{
    if (in_n == 0 || out_m == 0) return;
    const int64_t KB = 512;
    const int64_t nblk = (in_n + KB - 1) / KB;
#pragma omp parallel
    {
        F *R = (F *)malloc(sizeof(F) * KB * out_m);   /* [col][k]  d*X */
        F *L = (F *)malloc(sizeof(F) * KB * out_m);   /* [col][k]    X */
        F *part = (F *)calloc((size_t)out_m * out_m, sizeof(F));
#pragma omp for schedule(dynamic, 1)
        for (int64_t b = 0; b < nblk; b++) {
            int64_t k0 = b * KB, k1 = k0 + KB;
            if (k1 > in_n) k1 = in_n;
            int64_t kl = k1 - k0;
            for (int64_t c = 0; c < out_m; c++) {
                int64_t jj = cols[c];
                for (int64_t k = 0; k < kl; k++) {
                    int64_t kk = rows[k0 + k];
                    F x = order_f ? X[jj * n + kk] : X[kk * m + jj];
                    L[c * KB + k] = x;
                    R[c * KB + k] = d[kk] * x;
                }
            }
            for (int64_t i = 0; i < out_m; i++) {
                const F *Li = L + i * KB;
                for (int64_t j = 0; j <= i; j++) {
                    const F *Rj = R + j * KB;
                    F acc = 0;
#pragma omp simd reduction(+ : acc)
                    for (int64_t k = 0; k < kl; k++) acc += Li[k] * Rj[k];
                    part[i * out_m + j] += acc;
                }
            }
        }
#pragma omp critical
        for (int64_t e = 0; e < out_m * out_m; e++) out[e] += part[e];
        free(R); free(L); free(part);
    }
    for (int64_t i = 0; i < out_m; i++)
        for (int64_t j = 0; j <= i; j++) out[j * out_m + i] = out[i * out_m + j];
}

/*
 * K5a restricted dense  X[rows,cols]^T v[rows]
 * Reference: dense_helpers-tmpl.cpp:314-383 (_dense{C,F}_rmatvec), 256-row
 * blocks (328), per-thread outlocal + merge (338,375-378).
 */
void FN(dense_rmatvec)(const F *X, int64_t n, int64_t m, int order_f, const F *v,
                       const int32_t *rows, int64_t n_rows,
                       const int32_t *cols, int64_t n_cols, F *out)
{
    if (n_rows == 0 || n_cols == 0) return;
    const int64_t RB = 256;
    const int64_t nblk = (n_rows + RB - 1) / RB;
#pragma omp parallel
    {
        F *loc = (F *)calloc((size_t)n_cols, sizeof(F));
#pragma omp for
        for (int64_t b = 0; b < nblk; b++) {
            int64_t r1 = (b + 1) * RB < n_rows ? (b + 1) * RB : n_rows;
            for (int64_t c = 0; c < n_cols; c++) {
                int64_t j = cols[c];
                F acc = 0;
                for (int64_t r = b * RB; r < r1; r++) {
                    int64_t i = rows[r];
                    acc += (order_f ? X[j * n + i] : X[i * m + j]) * v[i];
                }
                loc[c] += acc;
            }
        }
#pragma omp critical
        for (int64_t c = 0; c < n_cols; c++) out[c] += loc[c];
        free(loc);
    }
}

/*
 * K5b restricted dense  X[rows,cols] v[cols]   (v has full length m)
 * Reference: dense_helpers-tmpl.cpp:385-417 (_dense{C,F}_matvec).
 */
void FN(dense_matvec)(const F *X, int64_t n, int64_t m, int order_f, const F *v,
                      const int32_t *rows, int64_t n_rows,
                      const int32_t *cols, int64_t n_cols, F *out)
{
#pragma omp parallel for
    for (int64_t r = 0; r < n_rows; r++) {
        int64_t i = rows[r];
        F acc = 0;
        for (int64_t c = 0; c < n_cols; c++) {
            int64_t j = cols[c];
            acc += (order_f ? X[j * n + i] : X[i * m + j]) * v[j];
        }
        out[r] += acc;
    }
}

/*
 * K4a  res[idx[i]-drop_first] += other[i]     (all rows, all cols)
 * Reference: cat_split_helpers-tmpl.cpp:4-41 (_transpose_matvec_all_rows_*):
 * per-thread private bins (17), then a deterministic column-wise reduction in
 * thread order (33-38).  Restricted variants: ext/categorical.pyx:44-67, 92-117
 * (serial loops with an optional rows list and an int32 col-included mask).
 * idx == -1 (missing) and idx == 0 under drop_first contribute nothing
 * (cat_split_helpers-tmpl.cpp:24-28).
 */
void FN(cat_transpose_matvec)(const int32_t *idx, int64_t n, const F *other,
                              int64_t n_cols, int drop_first,
                              const int32_t *rows, int64_t n_rows,  /* NULL = all */
                              const int32_t *col_included,         /* NULL = all */
                              F *out)
{
    if (rows == NULL && col_included == NULL) {
        int nt = omp_get_max_threads();
        F *all = (F *)calloc((size_t)nt * (size_t)(n_cols > 0 ? n_cols : 1), sizeof(F));
#pragma omp parallel
        {
            F *sl = all + (size_t)omp_get_thread_num() * n_cols;
#pragma omp for
            for (int64_t i = 0; i < n; i++) {
                int64_t c = (int64_t)idx[i] - drop_first;
                if (c >= 0) sl[c] += other[i];
            }
#pragma omp for
            for (int64_t c = 0; c < n_cols; c++)
                for (int t = 0; t < nt; t++) out[c] += all[(size_t)t * n_cols + c];
        }
        free(all);
        return;
    }
    int64_t cnt = rows ? n_rows : n;
    for (int64_t r = 0; r < cnt; r++) {
        int64_t i = rows ? rows[r] : r;
        int64_t c = (int64_t)idx[i] - drop_first;
        if (c >= 0 && (col_included == NULL || col_included[c])) out[c] += other[i];
    }
}

/*
 * K4e  out[i] += other[idx[i]-drop_first]   (gather)
 * Reference: ext/categorical.pyx:128-180 (matvec_fast / matvec_complex).
 */
void FN(cat_matvec)(const int32_t *idx, int64_t n, const F *other, int drop_first,
                    const int32_t *col_included /* NULL = all */, F *out)
{
#pragma omp parallel for
    for (int64_t i = 0; i < n; i++) {
        int64_t c = (int64_t)idx[i] - drop_first;
        if (c >= 0 && (col_included == NULL || col_included[c] == 1)) out[i] += other[c];
    }
}

/*
 * K4b  res[idx[k]-drop_first] += d[k]  for k in rows  (diagonal of cat sandwich)
 * Reference: ext/categorical.pyx:183-218 (sandwich_categorical_fast/_complex),
 * a serial loop.
 */
void FN(cat_sandwich_diag)(const int32_t *idx, const F *d, const int32_t *rows,
                           int64_t n_rows, int drop_first, F *res)
{
    for (int64_t r = 0; r < n_rows; r++) {
        int64_t k = rows[r];
        int64_t c = (int64_t)idx[k] - drop_first;
        if (c >= 0) res[c] += d[k];
    }
}

/*
 * K4c  res[i_idx[k]-di, j_idx[k]-dj] += d[k]
 * Reference: cat_split_helpers-tmpl.cpp:44-94 (_sandwich_cat_cat_*):
 * per-thread dense restemp (63) + atomic merge (88-91).
 */
void FN(cat_cat_sandwich)(const int32_t *i_idx, const int32_t *j_idx,
                          const F *d, const int32_t *rows, int64_t n_rows,
                          int64_t i_ncol, int64_t j_ncol,
                          int i_drop_first, int j_drop_first, F *res)
{
    (void)i_ncol;
    for (int64_t r = 0; r < n_rows; r++) {
        int64_t k = rows[r];
        int64_t i = (int64_t)i_idx[k] - i_drop_first;
        if (i < 0) continue;
        int64_t j = (int64_t)j_idx[k] - j_drop_first;
        if (j < 0) continue;
        res[i * j_ncol + j] += d[k];
    }
}

/*
 * K4d  res[idx[k]-drop, jc] += d[k] * M[k, j_cols[jc]]
 * Reference: cat_split_helpers-tmpl.cpp:97-151 (_sandwich_cat_dense{C,F}_*):
 * per-thread restemp (118) + atomic merge (145-148).
 */
void FN(cat_dense_sandwich)(const int32_t *idx, const F *d,
                            const int32_t *rows, int64_t n_rows,
                            const int32_t *j_cols, int64_t n_j,
                            const F *M, int64_t M_nrow, int64_t M_ncol, int order_f,
                            int drop_first, int64_t i_ncol, F *res)
{
    if (n_rows == 0 || n_j == 0 || i_ncol == 0) return;
#pragma omp parallel
    {
        F *tmp = (F *)calloc((size_t)i_ncol * n_j, sizeof(F));
#pragma omp for
        for (int64_t r = 0; r < n_rows; r++) {
            int64_t k = rows[r];
            int64_t i = (int64_t)idx[k] - drop_first;
            if (i < 0) continue;
            for (int64_t jc = 0; jc < n_j; jc++) {
                int64_t j = j_cols[jc];
                F x = order_f ? M[j * M_nrow + k] : M[k * M_ncol + j];
                tmp[i * n_j + jc] += d[k] * x;
            }
        }
#pragma omp critical
        for (int64_t e = 0; e < i_ncol * n_j; e++) res[e] += tmp[e];
        free(tmp);
    }
}

/*
 * K7  weighted second moment of dense columns  out[j] = sum_i w[i] (X[i,j]-shift[j])^2
 * Reference: ext/dense.pyx:103-122 (transpose_square_dot_weights).
 */
void FN(dense_col_sq_dev)(const F *X, int64_t n, int64_t m, int order_f,
                          const F *w, const F *shift, F *out)
{
#pragma omp parallel for
    for (int64_t j = 0; j < m; j++) {
        F acc = 0;
        for (int64_t i = 0; i < n; i++) {
            F x = (order_f ? X[j * n + i] : X[i * m + j]) - shift[j];
            acc += w[i] * x * x;
        }
        out[j] += acc;
    }
}

#else /* ORACLE_SPARSE_PASS */
/* ------------------------------------------------------------------------- */
/* sparse pass: symbols are  orc_<name>_<f32|f64>_<i32|i64>                  */
/* ------------------------------------------------------------------------- */
#define FN(name) CAT(CAT(orc_##name, FS), IS)

/*
 * K2  sparse self-sandwich  out = A[rows,cols]^T diag(d) A[rows,cols]
 * Reference: ext/sparse.pyx:17-77 (sparse_sandwich).  A in CSC, AT = CSR twin
 * of the same matrix; for each output column Cj (prange, 55) walk CSC column j,
 * and for each of its entries k walk CSR row k up to column j (break at i > j,
 * 64-67; needs sorted indices), scattering into row Cj of out; finally
 * out += tril(out,-1).T (76).  uint8 row mask 46-48, int32 col_map 50-52.
 */
void FN(sparse_sandwich)(const F *Adata, const I *Aindices, const I *Aindptr,
                         const F *ATdata, const I *ATindices, const I *ATindptr,
                         int64_t n, int64_t ncol, const F *d,
                         const I *rows, int64_t n_rows,
                         const I *cols, int64_t m, F *out)
{
    if (m == 0) return;
    uint8_t *row_inc = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
    for (int64_t r = 0; r < n_rows; r++) row_inc[rows[r]] = 1;
    int32_t *col_map = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ncol > 0 ? ncol : 1));
    for (int64_t c = 0; c < ncol; c++) col_map[c] = -1;
    for (int64_t c = 0; c < m; c++) col_map[cols[c]] = (int32_t)c;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t Cj = 0; Cj < m; Cj++) {
        int64_t j = cols[Cj];
        for (int64_t a = Aindptr[j]; a < Aindptr[j + 1]; a++) {
            int64_t k = Aindices[a];
            if (!row_inc[k]) continue;
            F Aval = Adata[a] * d[k];
            for (int64_t t = ATindptr[k]; t < ATindptr[k + 1]; t++) {
                int64_t i = ATindices[t];
                if (i > j) break;
                int32_t Ci = col_map[i];
                if (Ci == -1) continue;
                out[Cj * m + Ci] += ATdata[t] * Aval;
            }
        }
    }
    /* out += tril(out, -1).T  (ext/sparse.pyx:76) */
    for (int64_t i = 0; i < m; i++)
        for (int64_t j = 0; j < i; j++) out[j * m + i] += out[i * m + j];
    free(row_inc); free(col_map);
}

/*
 * K3  out[nA x nB] = A[rows,A_cols]^T diag(d) B[rows,B_cols],  A in CSR
 * Reference: sparse_helpers-tmpl.cpp:23-146 (_csr_dense{C,F}_sandwich):
 * Acol_map (50-54), per-thread outtemp (57-65), 128-row blocks (67-68),
 * R = d o B packed per block (80-91), AXPY outtemp[Ci,:] += A[k,i] R[k,:] per
 * nonzero (93-131), merge (135-140).
 */
void FN(csr_dense_sandwich)(const F *Adata, const I *Aindices, const I *Aindptr,
                            const F *B, const F *d, F *out,
                            int64_t m, int64_t n, int64_t r, int order_f,
                            const I *rows, const I *A_cols, const I *B_cols,
                            int64_t nrows, int64_t nA, int64_t nB)
{
    if (nrows == 0 || nA == 0 || nB == 0) return;
    int64_t *Acol_map = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m > 0 ? m : 1));
    for (int64_t c = 0; c < m; c++) Acol_map[c] = -1;
    for (int64_t c = 0; c < nA; c++) Acol_map[A_cols[c]] = c;
    const int64_t KBK = 128;
    const int64_t nblk = (nrows + KBK - 1) / KBK;
#pragma omp parallel
    {
        F *tmp = (F *)calloc((size_t)nA * nB, sizeof(F));
        F *R = (F *)malloc(sizeof(F) * KBK * nB);
#pragma omp for schedule(dynamic, 4)
        for (int64_t b = 0; b < nblk; b++) {
            int64_t k0 = b * KBK, k1 = k0 + KBK;
            if (k1 > nrows) k1 = nrows;
            for (int64_t ck = k0; ck < k1; ck++) {
                int64_t k = rows[ck];
                for (int64_t cj = 0; cj < nB; cj++) {
                    int64_t j = B_cols[cj];
                    F bv = order_f ? B[j * n + k] : B[k * r + j];
                    R[(ck - k0) * nB + cj] = d[k] * bv;
                }
            }
            for (int64_t ck = k0; ck < k1; ck++) {
                int64_t k = rows[ck];
                const F *Rk = R + (ck - k0) * nB;
                for (int64_t a = Aindptr[k]; a < Aindptr[k + 1]; a++) {
                    int64_t Ci = Acol_map[Aindices[a]];
                    if (Ci == -1) continue;
                    F Q = Adata[a];
                    F *o = tmp + Ci * nB;
                    for (int64_t cj = 0; cj < nB; cj++) o[cj] += Q * Rk[cj];
                }
            }
        }
#pragma omp critical
        for (int64_t e = 0; e < nA * nB; e++) out[e] += tmp[e];
        free(tmp); free(R);
    }
    free(Acol_map);
}

/*
 * K6a  CSR matvec, all rows / all cols:  out[i] += sum X[i,j] v[j]
 * Reference: ext/sparse.pyx:79-103 (csr_matvec_unrestricted).
 */
void FN(csr_matvec_unrestricted)(const F *Xd, const I *Xi, const I *Xp, int64_t n,
                                 const F *v, F *out)
{
#pragma omp parallel for
    for (int64_t i = 0; i < n; i++) {
        F acc = out[i];
        for (int64_t t = Xp[i]; t < Xp[i + 1]; t++) acc += Xd[t] * v[Xi[t]];
        out[i] = acc;
    }
}

/*
 * K6b  CSR matvec restricted: out[Ci] += sum_{j in cols} X[rows[Ci], j] v[j]
 * Reference: ext/sparse.pyx:105-140 (csr_matvec): uint8 col_included (127-129).
 */
void FN(csr_matvec)(const F *Xd, const I *Xi, const I *Xp, int64_t ncol,
                    const F *v, const I *rows, int64_t n_rows,
                    const I *cols, int64_t n_cols, F *out)
{
    uint8_t *inc = (uint8_t *)calloc((size_t)(ncol > 0 ? ncol : 1), 1);
    for (int64_t c = 0; c < n_cols; c++) inc[cols[c]] = 1;
#pragma omp parallel for
    for (int64_t Ci = 0; Ci < n_rows; Ci++) {
        int64_t i = rows[Ci];
        F acc = out[Ci];
        for (int64_t t = Xp[i]; t < Xp[i + 1]; t++) {
            int64_t j = Xi[t];
            if (inc[j]) acc += Xd[t] * v[j];
        }
        out[Ci] = acc;
    }
    free(inc);
}

/*
 * K6c  CSC transpose-matvec, all rows / cols: out[j] += sum_i X[i,j] v[i]
 * Reference: ext/sparse.pyx:142-166 (csc_rmatvec_unrestricted).
 */
void FN(csc_rmatvec_unrestricted)(const F *Xd, const I *Xi, const I *Xp, int64_t m,
                                  const F *v, F *out)
{
#pragma omp parallel for
    for (int64_t j = 0; j < m; j++) {
        F acc = out[j];
        for (int64_t t = Xp[j]; t < Xp[j + 1]; t++) acc += Xd[t] * v[Xi[t]];
        out[j] = acc;
    }
}

/*
 * K6d  CSC transpose-matvec restricted: out[Cj] += sum_{i in rows} X[i,cols[Cj]] v[i]
 * Reference: ext/sparse.pyx:168-199 (csc_rmatvec): uint8 row_included (186-188).
 */
void FN(csc_rmatvec)(const F *Xd, const I *Xi, const I *Xp, int64_t nrow,
                     const F *v, const I *rows, int64_t n_rows,
                     const I *cols, int64_t n_cols, F *out)
{
    uint8_t *inc = (uint8_t *)calloc((size_t)(nrow > 0 ? nrow : 1), 1);
    for (int64_t r = 0; r < n_rows; r++) inc[rows[r]] = 1;
#pragma omp parallel for
    for (int64_t Cj = 0; Cj < n_cols; Cj++) {
        int64_t j = cols[Cj];
        F acc = out[Cj];
        for (int64_t t = Xp[j]; t < Xp[j + 1]; t++) {
            int64_t i = Xi[t];
            if (inc[i]) acc += Xd[t] * v[i];
        }
        out[Cj] = acc;
    }
    free(inc);
}

/*
 * cat x sparse cross term  res[cat[k]-drop, Cj] += d[k] * S[k, cols[Cj]],  k in rows
 * Reference: categorical_matrix.py:825-838 (_cross_sparse) builds
 * term_1 = CSR(n x ncat, data=d) (multiply, 840-876; multiply_complex,
 * ext/categorical.pyx:221-271, drops idx < drop_first) and evaluates
 * term_1[rows, L_cols].T.dot(S[rows, R_cols]).toarray() with scipy.sparse.
 * THIRD-PARTY arithmetic: scipy.sparse (unpinned in setup.py:157; >=1.7.3 in
 * pixi.toml:104; 1.15.3 in this image).  scipy evaluates the product with the
 * SMMP row-by-row algorithm (csr_matmat in sparsetools/csr.h): for every row
 * of the left CSR operand, every entry (i,k,v) scatters v * right[k,:] into an
 * accumulator row.  With the left operand = (diag(d) onehot)^T that is exactly
 * the loop below: one entry per matrix row k, scattering d[k] * S[k,:] into
 * result row cat[k].  S is walked through its CSR twin; cols mapped by col_map.
 */
void FN(cat_sparse_sandwich)(const int32_t *idx, int drop_first, int64_t i_ncol,
                             const F *Sd, const I *Si, const I *Sp /* CSR of S */,
                             int64_t s_ncol, const F *d,
                             const I *rows, int64_t n_rows,
                             const I *cols, int64_t n_cols, F *res)
{
    (void)i_ncol;
    int64_t *col_map = (int64_t *)malloc(sizeof(int64_t) * (size_t)(s_ncol > 0 ? s_ncol : 1));
    for (int64_t c = 0; c < s_ncol; c++) col_map[c] = -1;
    for (int64_t c = 0; c < n_cols; c++) col_map[cols[c]] = c;
    for (int64_t r = 0; r < n_rows; r++) {
        int64_t k = rows[r];
        int64_t i = (int64_t)idx[k] - drop_first;
        if (i < 0) continue;
        for (int64_t t = Sp[k]; t < Sp[k + 1]; t++) {
            int64_t Cj = col_map[Si[t]];
            if (Cj >= 0) res[i * n_cols + Cj] += d[k] * Sd[t];
        }
    }
    free(col_map);
}

/*
 * K7  sparse weighted column second moment  out[j] = sum_k w[i_k] v_k^2
 * Reference: ext/sparse.pyx:262-282 (transpose_square_dot_weights).
 */
void FN(csc_col_sq)(const F *data, const I *indices, const I *indptr, int64_t ncol,
                    const F *w, F *out)
{
#pragma omp parallel for
    for (int64_t j = 0; j < ncol; j++) {
        F acc = 0;
        for (int64_t t = indptr[j]; t < indptr[j + 1]; t++)
            acc += w[indices[t]] * data[t] * data[t];
        out[j] += acc;
    }
}
#endif /* ORACLE_SPARSE_PASS */

#undef FN
#undef CAT
#undef CAT2
