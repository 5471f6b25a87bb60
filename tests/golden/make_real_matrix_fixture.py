"""Generate tests/golden/real_matrix_blocks.npz from the reference's only data
fixture, /root/reference/tests/real_matrix.pkl (a 100 x 9 pandas DataFrame used
by /root/reference/tests/test_real_matrix.py).

Run in the build container only (needs /root/reference and pandas):
    python tests/golden/make_real_matrix_fixture.py

The npz holds DATA: the blocks that tabmat.from_df(df, np.float64) produces
(restating constructor.py:29-212 with its default thresholds: categoricals with
>= 4 levels become categorical blocks in place ("expand"), smaller ones are
one-hot expanded and split dense/sparse at density 0.1, numeric columns go to
one dense or one sparse block), plus expected outputs computed by dense
extended-precision algebra on the one-hot design matrix (the same check
test_real_matrix.py performs against DenseMatrix).
"""
import os

import numpy as np
import pandas as pd

SRC = "/root/reference/tests/real_matrix.pkl"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "real_matrix_blocks.npz")


def main():
    build(DST, cat_threshold=4, sparse_threshold=0.1)
    # Variant that FORCES a sparse block (the default thresholds yield [cat, dense, cat, cat, cat]):
    # cat_threshold above every level count one-hot expands all categoricals and
    # sparse_threshold = 0.3 sends their low-density columns to the sparse block
    # (constructor.py:128-153 allows both), so K2 / K3 / categorical-free cross terms see
    # reference-held data too.  The tests also load the sparse block with int64 CSC indices.
    build(DST.replace(".npz", "_sparse.npz"), cat_threshold=10**6, sparse_threshold=0.3)
    # ... and one with all three kinds side by side: [sparse, dense, cat, cat]
    build(DST.replace(".npz", "_mixed.npz"), cat_threshold=8, sparse_threshold=0.3)


def build(dst, cat_threshold, sparse_threshold):
    df = pd.read_pickle(SRC)
    n = df.shape[0]
    out = {}
    kinds = []          # per block: 'cat' | 'dense' | 'sparse'
    col = 0
    dense_cols, dense_idx, sparse_cols, sparse_idx = [], [], [], []
    design = []         # dense one-hot design matrix columns, in SplitMatrix column order
    blocks = []         # (kind, payload, indices)
    for name in df.columns:
        s = df[name]
        if isinstance(s.dtype, pd.CategoricalDtype):
            codes = s.cat.codes.to_numpy().astype(np.int32)
            ncat = len(s.cat.categories)
            onehot = np.zeros((n, ncat))
            onehot[np.arange(n), codes] = 1.0
            if ncat < cat_threshold:   # constructor.py:128-153 -> _split_sparse_and_dense_parts
                dens = (onehot != 0).mean(0)
                d_loc = np.where(dens > sparse_threshold)[0]
                s_loc = np.where(dens <= sparse_threshold)[0]
                if len(d_loc):
                    blocks.append(("dense", np.asfortranarray(onehot[:, d_loc]), col + d_loc))
                if len(s_loc):
                    blocks.append(("sparse", onehot[:, s_loc], col + s_loc))
            else:
                blocks.append(("cat", (codes, ncat), col + np.arange(ncat)))
            design.append(onehot)
            col += ncat
        else:
            x = s.to_numpy().astype(np.float64)
            if (x != 0).mean() <= sparse_threshold:
                sparse_cols.append(x); sparse_idx.append(col)
            else:
                dense_cols.append(x); dense_idx.append(col)
            design.append(x[:, None])
            col += 1
    if dense_cols:
        blocks.append(("dense", np.column_stack(dense_cols), np.asarray(dense_idx)))
    if sparse_cols:
        blocks.append(("sparse", np.column_stack(sparse_cols), np.asarray(sparse_idx)))

    # SplitMatrix.__init__ merges all dense blocks into one and all sparse
    # blocks into one, columns sorted by global index (split_matrix.py:85-141)
    for kind in ("dense", "sparse"):
        which = [i for i, b in enumerate(blocks) if b[0] == kind]
        if len(which) > 1:
            idx = np.concatenate([blocks[i][2] for i in which])
            arr = np.hstack([blocks[i][1] for i in which])
            srt = np.argsort(idx)
            blocks[which[0]] = (kind, np.asfortranarray(arr[:, srt]), idx[srt])
            blocks = [b for i, b in enumerate(blocks) if i not in which[1:]]

    X = np.hstack(design)
    assert X.shape[1] == col
    for b, (kind, payload, idx) in enumerate(blocks):
        kinds.append(kind)
        out[f"b{b}_indices"] = np.asarray(idx, dtype=np.int64)
        if kind == "cat":
            out[f"b{b}_codes"] = payload[0]
            out[f"b{b}_ncat"] = np.int64(payload[1])
        else:
            out[f"b{b}_array"] = payload
    out["kinds"] = np.array(kinds)
    out["design"] = X

    rng = np.random.default_rng(20240917)
    d = rng.random(n)
    v = rng.standard_normal(col)
    w = rng.standard_normal(n)
    XL = X.astype(np.longdouble)
    out["d"] = d
    out["v"] = v
    out["w"] = w
    out["sandwich"] = ((XL.T * d.astype(np.longdouble)) @ XL).astype(np.float64)
    out["matvec"] = (XL @ v.astype(np.longdouble)).astype(np.float64)
    out["transpose_matvec"] = (XL.T @ w.astype(np.longdouble)).astype(np.float64)
    rows = np.sort(rng.choice(n, size=37, replace=False)).astype(np.int32)
    cols = np.sort(rng.choice(col, size=21, replace=False)).astype(np.int32)
    Xs = XL[np.ix_(rows, cols)]
    out["rows"] = rows
    out["cols"] = cols
    out["sandwich_rows_cols"] = ((Xs.T * d[rows].astype(np.longdouble)) @ Xs).astype(np.float64)
    np.savez_compressed(dst, **out)
    print("wrote", dst, "blocks:", kinds, "p =", col)


if __name__ == "__main__":
    main()
