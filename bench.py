#!/usr/bin/env python
"""bench.py -- SplitMatrix.sandwich throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one SplitMatrix.sandwich(d) over one batch of synthetic rows resident in HBM:
the default workload is BASELINE.json configs[3] -- dense 128 cols + CSC-sparse 512 cols @5% +
3 categoricals (256, 96, 32 levels) => p = 1024, 10M rows per GPU, float64 -- the configuration
the metric ("10M x 1k mixed") is quoted on.  With N > 1 every rank owns its own 10M-row shard
(weak scaling, configs[4]) and the p x p partials are summed with one RCCL all-reduce per step.

Rank 0 prints ONE JSON line.  value = algorithmic bytes of the whole job (every operand read
once, result written once) / wall time of a step, in GB/s; gflops is reported
next to it.  `roofline` describes the dominant kernel, timed live with HIP events on the
launch stream (tm_profile_* in the C ABI); `cpu_baseline` is the CPU oracle ("port") timed on
this box's host cores on a bounded row sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_F64_PEAK_TFLOPS = 78.6    # MI355X FP64 matrix (AMD datasheet; not in the guide's table)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md
MFMA_BF16_PEAK_TFLOPS = 2500.0 # dense bf16 (MI355X_MICROARCH.md; 2:1-sparsity figures are never used)
INT8_PEAK_TOPS = 5000.0        # dense int8 MFMA = the fp8 rate (MI355X_MICROARCH.md: ~5 P dense; >= 3944 TOPS measured)
# LDS: "Aggregate with every CU streaming (~2.4 GHz): ~150 TB/s for ds_read_b64/b128" (MI355X_MICROARCH.md, LDS
# section; 64 banks x 4 B x 256 CUs x 2.4 GHz = 157 TB/s is the array's width)
LDS_PEAK_GBS = 150000.0


# cfg4 (10M rows): what the CURRENT formulation -- one kernel per block product, each streaming its operands in the
# order it works best on -- could reach with every kernel at the floor its own ablations show (profiles/r4_k3_ent.txt,
# profiles/r4_k2b.txt, profiles/r3_syrk_i8.txt).  The north star's 2.8 ms (0.60 of HBM peak) is
# below the f64 arithmetic of the dense term alone at any rate this chip has: the step is printed
# against both.
DESIGN_FLOOR_CFG4_MS = {
    "sparse x dense (K3)": (2.56, "memory side alone: entry stream + slab copies, no batches (r4_k3_ent.txt)"),
    "sparse self (K2b)": (3.70, "every gather served from the L2: the LDS-atomic pipe alone (r4_k2b.txt)"),
    "dense self (K1e)": (1.91, "copy-only rate of its LDS-DMA pattern (r3_syrk_i8.txt)"),
    "categorical x dense": (1.30, "10.4 GB at the 8 TB/s HBM peak"),
    "categorical x sparse": (0.46, "3.7 GB at the 8 TB/s HBM peak"),
    "tables, reductions, scatter": (0.40, "~25 small launches (r4_bench_cfg4_kernel_stats.txt)"),
}


def _baseline_metric():
    """BASELINE.json's metric string, verbatim."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json"), encoding="utf-8") as f:
            return json.load(f)["metric"]
    except Exception:
        return "SplitMatrix.sandwich GFLOP/s + effective HBM GB/s, 10M\u00d71k mixed"


METRIC = _baseline_metric()


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=10_000_000, help="rows per GPU")
    ap.add_argument("--workload", default="cfg4",
                    choices=["cfg4", "cfg2", "cfg3", "cfg1", "dense", "sparse", "sparse_narrow",
                             "sparse_wide", "one_cat", "two_cat", "dense_cat", "dense_smallcat"],
                    help="cfg1-4: BASELINE.json configs[0..3]; the others: the reference's own "
                         "benchmark designs (benchmark/generate_matrices.py:90-100)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a captured HIP graph (SplitMatrix.sandwich_graph) "
                         "instead of launching every kernel eagerly; single-GPU only")
    ap.add_argument("--cpu-rows", type=int, default=0,
                    help="rows of the bounded CPU-baseline sample (0 = per-workload default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="per-op kernel times to stderr")
    ap.add_argument("--out", default=None, help="also write the JSON (+breakdown) to this file")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the rocprofv3 FETCH_SIZE / WRITE_SIZE passes over the dominant kernel")
    ap.add_argument("--traffic-child", default=None, help=argparse.SUPPRESS)
    return ap.parse_args()


def build_workload(name, n, seed):
    from tabmat_amd import synth

    if name == "cfg4":
        return synth.mixed_split(n, 128, 512, (256, 96, 32), 0.05, torch.float64, seed), torch.float64
    if name == "cfg2":
        return synth.dense_block(n, 256, torch.float32, seed), torch.float32
    if name == "cfg3":
        return synth.cat_block(n if n != 10_000_000 else 50_000_000, 10_000, seed), torch.float64
    if name == "cfg1":
        return synth.dense_block(n if n != 10_000_000 else 100_000, 64, torch.float64, seed), torch.float64
    if name in synth.REFERENCE_DESIGNS:
        return synth.reference_design(name, None if n == 10_000_000 else n, seed), torch.float64
    raise ValueError(name)


def _rebuild_twins_ms(mat):
    """Steady-state time of to_device(): fresh wrappers over the SAME device arrays (nothing cached on them), twins
    built, wrappers dropped.  None for blocks without derived forms."""
    import tabmat_amd as tm
    from tabmat_amd.ext._types import CsrDev, release_index_scratch

    mats = mat.matrices if isinstance(mat, tm.SplitMatrix) else [mat]
    if not any(isinstance(m, tm.SparseMatrix) for m in mats):
        return None
    fresh = []
    for m in mats:
        if isinstance(m, tm.SparseMatrix):
            A = m._dev()
            fresh.append(tm.SparseMatrix.from_device(CsrDev(A.data, A.indices.clone(), A.indptr, A.n, A.m)))
        else:
            fresh.append(m)
    release_index_scratch()
    clone = tm.SplitMatrix(fresh, mat.indices) if isinstance(mat, tm.SplitMatrix) else fresh[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    clone.to_device()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    del clone, fresh
    torch.cuda.empty_cache()
    return round(ms, 1)


def kernel_ops(mat, d):
    """[(name, thunk)] of every block / block-pair op that one sandwich launches."""
    import tabmat_amd as tm

    mats = mat.matrices if isinstance(mat, tm.SplitMatrix) else [mat]
    ops = []
    for i, mi in enumerate(mats):
        ki = type(mi).__name__.replace("Matrix", "").lower()
        if isinstance(mi, tm.CategoricalMatrix):
            ops.append((f"{ki}{i}.self", lambda mi=mi: mi._sandwich_diag_dev(d, None, None)))
        else:
            ops.append((f"{ki}{i}.self", lambda mi=mi: mi._sandwich_dev(d, None, None)))
        for j in range(i + 1, len(mats)):
            kj = type(mats[j]).__name__.replace("Matrix", "").lower()
            ops.append((f"{ki}{i}x{kj}{j}", lambda mi=mi, mj=mats[j]: mi._cross_sandwich_dev(
                mj, d, None, None, None)))
    if isinstance(mat, tm.SplitMatrix):
        from tabmat_amd.ext import split as xsplit

        cat_ms = [m for m in mats if isinstance(m, tm.CategoricalMatrix)]
        if len(cat_ms) >= 2:
            cats = [(m._dev(), m.shape[1], m.drop_first) for m in cat_ms]
            fused = []
            for i, mw in enumerate(mats):
                if isinstance(mw, tm.DenseMatrix):
                    from tabmat_amd.ext import sparse as xs

                    # same choice as the product (SplitMatrix._fused_cats)
                    if xsplit.multi_cat_dense_wide_ok(cats, mw._dev()) or xsplit.multi_cat_dense_tile_ok(cats, mw._dev()):
                        fused.append((f"allcats_x_dense{i}", lambda mw=mw: xsplit.multi_cat_dense_sandwich(
                            cats, d, mw._dev())))
                    elif xsplit.cat_dense_sorted_ok(mw._dev()):
                        pass                                               # pair by pair: the per-pair ops stay
                    else:
                        cat_ids = [k for k, m in enumerate(mats) if isinstance(m, tm.CategoricalMatrix)]
                        oh, _ = mat._onehot_slab(cat_ids)
                        fused.append((f"allcats_x_dense{i}", lambda mw=mw, oh=oh: xs.csr_dense_sandwich_slab(
                            oh, mw._dev(), d)))
                elif isinstance(mw, tm.SparseMatrix):
                    # same choice as the product (`_entblk` is False when the twin was refused)
                    ent = mw._ent() if getattr(mw, "_entblk", None) else None
                    if ent is not None:
                        pk = xsplit.pack_codes(cats)                          # as SplitMatrix._fused_cats does
                        fused.append((f"allcats_x_sparse{i}", lambda ent=ent, pk=pk: xsplit.multi_cat_sparse_sandwich_ent(
                            cats, d, ent, pk)))
                    else:
                        fused.append((f"allcats_x_sparse{i}", lambda mw=mw: xsplit.multi_cat_sparse_sandwich(
                            cats, d, mw._slab())))
            # the fused kernels replace the per-pair categorical cross terms
            done = {name.split("_x_")[1] for name, _ in fused}
            ops = [o for o in ops if not ("categorical" in o[0] and "x" in o[0]
                                          and any(o[0].endswith("x" + w) or o[0].startswith(w + "x") for w in done))] + fused
    return ops


def kernel_breakdown(mat, d, reps=3):
    """Main-kernel time of every block / block-pair op of one sandwich, via tm_profile_*
    (HIP events on the launch stream around the op's main kernel)."""
    import ctypes as C

    from tabmat_amd import _lib

    ops = kernel_ops(mat, d)
    out = {}
    _lib.call("tm_profile_enable", 1)
    try:
        for name, fn in ops:
            ts = []
            for _ in range(reps):
                fn()
                ms = C.c_float(0)
                _lib.call("tm_profile_last_ms", C.byref(ms))
                ts.append(ms.value)
            out[name] = float(np.mean(ts[1:] if len(ts) > 1 else ts))
    finally:
        _lib.call("tm_profile_enable", 0)
    return out


def op_algorithmic_bytes(mat, name):
    """Algorithmic HBM bytes of ONE launch of the named op's main kernel: each operand of that
    kernel read once + its output written once ('algorithmic bytes')."""
    import tabmat_amd as tm

    mats = mat.matrices if isinstance(mat, tm.SplitMatrix) else [mat]
    isz = np.dtype(mat.dtype).itemsize
    n = mat.shape[0]

    def blk_bytes(m):
        if isinstance(m, tm.DenseMatrix):
            return n * m.shape[1] * isz
        if isinstance(m, tm.SparseMatrix):
            c = m._dev()
            return c.data.numel() * (isz + 4) + (n + 1) * 8
        return n * 4

    if name.endswith(".self"):
        i = int("".join(ch for ch in name.split(".")[0] if ch.isdigit()))
        k = mats[i].shape[1]
        outb = k * isz if isinstance(mats[i], tm.CategoricalMatrix) else k * k * isz
        return blk_bytes(mats[i]) + n * isz + outb
    if name.startswith("allcats_x_"):
        w = int("".join(ch for ch in name if ch.isdigit()))
        cat_ms = [m for m in mats if isinstance(m, tm.CategoricalMatrix)]
        return (blk_bytes(mats[w]) + sum(blk_bytes(m) for m in cat_ms) + n * isz
                + sum(m.shape[1] for m in cat_ms) * mats[w].shape[1] * isz)
    a, b = name.split("x")
    i = int("".join(ch for ch in a if ch.isdigit()))
    j = int("".join(ch for ch in b if ch.isdigit()))
    return blk_bytes(mats[i]) + blk_bytes(mats[j]) + n * isz + mats[i].shape[1] * mats[j].shape[1] * isz


def op_lds_bytes(mat, name):
    """Algorithmic LDS bytes of ONE launch of the named op's main kernel, for the kernels whose inner loop is
    an LDS gather / scatter ("the bounding roofline"): sparse x dense = one row of the dense
    operand (its columns x itemsize) read from the LDS slab per nonzero; sparse self = one 8-byte LDS atomic per
    pair of nonzeros of a row (lower triangle).  None for the streaming kernels."""
    import tabmat_amd as tm

    mats = mat.matrices if isinstance(mat, tm.SplitMatrix) else [mat]
    isz = np.dtype(mat.dtype).itemsize
    if name.endswith(".self"):
        i = int("".join(ch for ch in name.split(".")[0] if ch.isdigit()))
        if isinstance(mats[i], tm.SparseMatrix):
            c = (mats[i]._dev().indptr[1:] - mats[i]._dev().indptr[:-1]).to(torch.float64)
            return float((c * (c + 1) / 2).sum().item()) * 8.0
        return None
    if name.startswith("allcats_x_"):
        return None
    a, b = name.split("x")
    i = int("".join(ch for ch in a if ch.isdigit()))
    j = int("".join(ch for ch in b if ch.isdigit()))
    pair = {type(mats[i]), type(mats[j])}
    if pair == {tm.SparseMatrix, tm.DenseMatrix}:
        sp = mats[i] if isinstance(mats[i], tm.SparseMatrix) else mats[j]
        dn = mats[j] if sp is mats[i] else mats[i]
        return float(sp._dev().data.numel()) * dn.shape[1] * isz
    return None


def op_flops(mat, name):
    import tabmat_amd as tm

    mats = mat.matrices if isinstance(mat, tm.SplitMatrix) else [mat]
    n = mat.shape[0]
    if name.endswith(".self"):
        i = int("".join(ch for ch in name.split(".")[0] if ch.isdigit()))
        m = mats[i]
        if isinstance(m, tm.DenseMatrix):
            return float(n) * m.shape[1] * (m.shape[1] + 1)
        return None
    return None


# Reference CPU timings measured in the survey container (8 host cores, the
# reference's ext/ built with its own scalar non-xsimd fallback) -- context printed next to
# cpu_baseline; other hardware, other sizes, never a denominator.
REFERENCE_CPU_SURVEY = {
    "cfg1": "DenseMatrix.sandwich f64 100k x 64, full size: 13.2 ms (62 GFLOP/s)",
    "cfg2": "DenseMatrix.sandwich f32 1M x 256 (1/10 of the rows): 1.38 s (95 GFLOP/s)",
    "cfg3": "CategoricalMatrix 5M x 10k (1/10 of the rows): sandwich 17 ms, transpose_matvec 4 ms, matvec 22 ms",
    "cfg4": "SplitMatrix dense128 + CSC512@5% + cats(1000,300,50), 1M rows, p=1990: sandwich 4.40 s, "
            "matvec 0.137 s, transpose_matvec 0.200 s",
    "dense_cat": "the reference's illustrative CLI output (benchmark/main.py:274-277, hardware unstated): "
                 "sandwich 0.159682 s on 3M x (5 + 1000 + 1000)",
}


def _host_blocks(mat, rows):
    """Oracle blocks of the first `rows` rows of a device matrix (pulled back from HBM), with the
    column indices of every block."""
    from scipy import sparse as sps

    import tabmat_amd as tm

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _cases as cs

    mats = mat.matrices if isinstance(mat, tm.SplitMatrix) else [mat]
    idx = [np.asarray(i) for i in mat.indices] if isinstance(mat, tm.SplitMatrix) else \
        [np.arange(mat.shape[1])]
    specs = []
    for m in mats:
        if isinstance(m, tm.DenseMatrix):
            specs.append(("dense", m._dev().as_2d()[:rows].cpu().numpy()))
        elif isinstance(m, tm.SparseMatrix):
            c = m._dev()
            ptr = c.indptr[:rows + 1].cpu().numpy()
            e = int(ptr[-1])
            S = sps.csr_matrix((c.data[:e].cpu().numpy(), c.indices[:e].cpu().numpy(), ptr),
                               shape=(rows, c.m))
            specs.append(("sparse", S.tocsc()))
        else:
            specs.append(("cat", m._dev()[:rows].cpu().numpy(), m.shape[1] + int(m.drop_first),
                          m.drop_first))
    return [cs.to_oracle_block(sp) for sp in specs], idx


def cpu_baseline(workload, rows, mat, d):
    """Time the CPU oracle ("port") on a bounded sample: the first `rows` rows of the SAME data
    (pulled back from HBM), all host cores (OpenMP).  Returns the JSON object or None."""
    try:
        from oracle import oracle as orc

        blocks, idx = _host_blocks(mat, rows)
    except Exception as e:  # oracle not built
        return {"error": f"oracle unavailable: {e}"}
    import tabmat_amd as tm

    threads = orc.num_threads()
    dh = d[:rows].cpu().numpy()
    isz = dh.dtype.itemsize
    alg_bytes = rows * isz
    for b in blocks:
        if b.kind == "sparse":
            alg_bytes += b.csc.nnz * (isz + 4) + (rows + 1) * 8
        elif b.kind == "dense":
            alg_bytes += b.X.size * isz
        else:
            alg_bytes += rows * 4
    p = mat.shape[1]
    if isinstance(mat, tm.SplitMatrix):
        alg_bytes += p * p * 8
        fn = lambda: orc.split_sandwich(blocks, idx, dh)
    else:
        alg_bytes += (p if blocks[0].kind == "cat" else p * p) * isz
        fn = lambda: orc.block_sandwich(blocks[0], dh, None, None)
    fn()  # warm (threads, page faults)
    best = float("inf")
    t_all = time.time()
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
        if time.time() - t_all > 25:
            break
    return {
        "value": round(alg_bytes / best / 1e9, 4),
        "unit": "GB/s",
        "cores": int(threads),
        "kind": "port",
        "sample": (f"{workload}: first {rows} rows of the same data, min of <=3 runs, {best * 1e3:.1f} ms "
                   "(a bounded sample, not the full rows: the oracle needs host copies of the blocks -- 13.5 GB "
                   "pulled back over PCIe and ~10 s per pass at 10M rows -- and the default run has to finish in "
                   "minutes; the oracle's cost is linear in the rows, --cpu-rows N times any other sample)"),
        "seconds": round(best, 4),
        "reference_cpu_survey": REFERENCE_CPU_SURVEY.get(workload),
    }


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script through
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) and pass rank 0's
    JSON line through."""
    import socket
    import subprocess

    if torch.cuda.device_count() < args.gpus and os.environ.get("TABMAT_BENCH_BACKEND", "nccl") == "nccl":
        raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) visible "
                         "(TABMAT_BENCH_BACKEND=gloo lets ranks share devices for a dry run)")
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def measure_traffic(args, dom):
    """HBM bytes per launch of the dominant kernel from the L2's memory-side counters: two
    rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters only, no trace domains) over a child
    run of this script that launches that op a few times.  FETCH_SIZE is doubled (gfx950 counts
    wide coalesced reads at half, MI355X_MICROARCH.md "HBM"); WRITE_SIZE is taken as reported (KiB).
    Returns bytes or None."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    total = 0.0
    try:
        for counter, scale in (("FETCH_SIZE", 2.0 * 1024.0), ("WRITE_SIZE", 1024.0)):
            tmp = tempfile.mkdtemp(prefix="tm_traffic_", dir="/tmp")
            env = dict(os.environ, TMPDIR="/tmp")
            cmd = [exe, "--pmc", counter, "-d", tmp, "-o", "t", "--output-format", "csv", "--",
                   sys.executable, os.path.abspath(__file__), "--traffic-child", dom,
                   "--workload", args.workload, "--rows", str(args.rows)]
            subprocess.run(cmd, cwd="/tmp", env=env, timeout=240, check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
            best = {}
            for r in csv.DictReader(open(files[0])):
                if r["Counter_Name"] == counter and "tmh::" in r["Kernel_Name"]:
                    best.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
            # the op's main kernel is the one with the largest counter value per dispatch
            val = max(sum(v) / len(v) for v in best.values())
            total += val * scale
            shutil.rmtree(tmp, ignore_errors=True)
        return int(total)
    except Exception:
        return None


def traffic_child(args):
    """Child of measure_traffic: build the workload, launch the named op three times."""
    mat, tdt = build_workload(args.workload, args.rows, 3)
    d = torch.rand(mat.shape[0], dtype=tdt, device="cuda")
    import tabmat_amd as tm

    ops = dict(kernel_ops(mat, d))
    for _ in range(3):
        ops[args.traffic_child]()
    torch.cuda.synchronize()


def main():
    args = parse_args()
    # RCCL / device-memory sharing across the ranks of one node needs dmabuf IPC on this driver
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.traffic_child:
        return traffic_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # one process per GPU; TABMAT_BENCH_BACKEND=gloo lets the N > 1 control flow be exercised on a
    # box with fewer GPUs than ranks (ranks then share devices; not a performance configuration)
    backend = os.environ.get("TABMAT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from tabmat_amd import synth
    from tabmat_amd.distributed import RowShardedMatrix

    seed = 3 + rank
    # ---- ingest (outside the timed region, reported beside it): the derived forms of the blocks ("twins") are built
    # on the device by to_device().  ingest_ms is the FIRST build of this process (what a fit pays once; on a fresh box
    # it includes the one-off loads of the torch sort / scan code objects the builders use), ingest_rebuild_ms the same
    # twins built a second time from the same arrays on fresh wrappers (steady state), which are then dropped.
    t_syn = time.perf_counter()
    mat, tdt = build_workload(args.workload, args.rows, seed)
    torch.cuda.synchronize()
    synth_ms = (time.perf_counter() - t_syn) * 1e3
    data_bytes = torch.cuda.memory_allocated()
    torch.cuda.reset_peak_memory_stats()
    t_ing = time.perf_counter()
    if hasattr(mat, "to_device"):
        mat.to_device()
    torch.cuda.synchronize()
    ingest = {"ingest_ms": round((time.perf_counter() - t_ing) * 1e3, 1),
              "ingest_peak_bytes": int(torch.cuda.max_memory_allocated()),
              "data_bytes": int(data_bytes), "synth_ms": round(synth_ms, 1)}
    ingest["ingest_rebuild_ms"] = _rebuild_twins_ms(mat)
    n_local, p = mat.shape
    g = torch.Generator(device="cuda")
    g.manual_seed(100 + rank)
    d = torch.rand(n_local, dtype=tdt, device="cuda", generator=g)
    # every rank owns its own 10M-row shard (weak scaling); the wrapper adds the one collective
    # of the path: an all-reduce of the p x p result over RCCL
    sharded = RowShardedMatrix(mat, bounds=(rank * n_local, (rank + 1) * n_local),
                               n_global=world * n_local)

    import tabmat_amd as tm

    def product(dd):
        # the PUBLIC entry point, device vector in -> device result out (no host traffic):
        # SplitMatrix.sandwich / DenseMatrix.sandwich; a CategoricalMatrix returns a scipy
        # dia_matrix from its public method, so its device diagonal is timed instead
        if isinstance(mat, tm.CategoricalMatrix):
            return mat._sandwich_diag_dev(dd, None, None)
        return mat.sandwich(dd)

    # One step = one pass of the hot path over the shard, launched eagerly.  --graph replays the
    # same launch sequence (same kernels, order, arguments) from a HIP graph, the form a GLM solver
    # would use for its per-iteration sandwich (SplitMatrix.sandwich_graph); at 10M rows the two
    # are within 0.3 % (the step is ~45 launches of 0.02-11 ms each), it pays for small matrices.
    use_graph = args.graph and world == 1
    if not use_graph:
        run = product
    else:
        from tabmat_amd.graph import CapturedProduct

        run = CapturedProduct(lambda dd: mat._sandwich_dev(dd, None, None), d)
        d = run._static_in

    def step():
        out = run(d)
        if world > 1:
            out = sharded._all_reduce(out)  # RCCL over xGMI: p x p float64 (8 MB at p = 1024)
        return out

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    rank_ms = None
    if world > 1:
        # every rank's own clock over the same K steps: the max is the job's time (the contract), min / max side
        # by side make the first real multi-GPU run self-diagnosing (a straggler GPU, a slow link)
        mine = torch.zeros(world, dtype=torch.float64, device="cuda")
        mine[rank] = elapsed
        dist.all_reduce(mine)
        rank_ms = [round(float(x) / args.steps * 1e3, 4) for x in mine.tolist()]
        elapsed = float(mine.max().item())
    ms_per_step = elapsed / args.steps * 1e3
    # min over individually timed steps next to the mean (the reference harness reports the
    # minimum, benchmark/main.py:108-128); outside the timed region above
    per_step = []
    for _ in range(min(args.steps, 10)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step()
        e1.record()
        e1.synchronize()
        per_step.append(e0.elapsed_time(e1))
    ms_min = min(per_step)

    # what arithmetic the dense self term ran in, and the same step with the int8-sliced syrk switched off
    dense_term, handovers, ms_f64_only = None, None, None
    if tdt == torch.float64 and not use_graph:
        dms = [m for m in (mat.matrices if isinstance(mat, tm.SplitMatrix) else [mat])
               if isinstance(m, tm.DenseMatrix)]
        took_i8 = [m for m in dms if getattr(m, "_i8_hist", None)]
        if dms:
            dense_term = ("int8x5 (40-bit fixed point per column on the int8 matrix cores, K1e; hand-over to "
                          "the f64 MFMA kernel outside its envelope)") if took_i8 else "f64 MFMA"
        if took_i8:
            hist = took_i8[0]._i8_history(None)[:2].cpu().tolist()   # {consecutive envelope misses, calls}
            handovers = {"consecutive_envelope_misses": int(hist[0]), "calls": int(hist[1])}
            was = tm.set_strict_f64(True)
            try:
                for _ in range(max(1, args.warmup)):
                    step()
                if world > 1:
                    dist.barrier()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                ms_f64_only = (time.perf_counter() - t1) / args.steps * 1e3
            finally:
                tm.set_strict_f64(was)

    # the same launch sequence replayed from a HIP graph (the form a solver's per-iteration product takes,
    # SplitMatrix.sandwich_graph), reported beside the eager step -- never `value`
    ms_graph = None
    if world == 1 and not use_graph and hasattr(mat, "_sandwich_dev") and not isinstance(mat, tm.CategoricalMatrix):
        from tabmat_amd.graph import CapturedProduct

        try:
            cap = CapturedProduct(lambda dd: mat._sandwich_dev(dd, None, None), d)
            for _ in range(max(1, args.warmup)):
                cap(cap._static_in)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                cap(cap._static_in)
            torch.cuda.synchronize()
            ms_graph = (time.perf_counter() - t1) / args.steps * 1e3
            del cap
        except RuntimeError as e:      # a side measurement: the eager line stands without it
            print(f"[bench] hipgraph replay not measured: {e}", file=sys.stderr)

    # matvec / transpose_matvec of the SAME resident matrix (the north star names them; the reference's harness
    # times all three products, benchmark/main.py:58-62,108-128): per-GPU sub-records, never `value`.
    # Algorithmic bytes = the blocks once + the input and output vectors.
    mv = {}
    if True:
        blk_bytes = synth.algorithmic_bytes(mat) - n_local * np.dtype(mat.dtype).itemsize - (
            p * p * 8 if isinstance(mat, tm.SplitMatrix) else
            p * np.dtype(mat.dtype).itemsize if isinstance(mat, tm.CategoricalMatrix) else
            p * p * np.dtype(mat.dtype).itemsize)
        isz = np.dtype(mat.dtype).itemsize
        vg = torch.Generator(device="cuda")
        vg.manual_seed(7)
        vcoef = torch.randn(p, dtype=tdt, device="cuda", generator=vg)
        for name, fn in (("matvec", lambda: mat.matvec(vcoef)), ("transpose_matvec", lambda: mat.transpose_matvec(d))):
            try:
                for _ in range(max(1, args.warmup)):
                    fn()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    fn()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t1) / args.steps * 1e3
                # sparse blocks that stream their 16-bit column twin (tm_csr_{matvec,rmatvec}_u16_*) read 2 bytes per
                # column index, not 4: the bytes claimed follow what the kernel has to read
                blocks_ = mat.matrices if isinstance(mat, tm.SplitMatrix) else [mat]
                saved = sum(2 * int(b._dev().data.numel()) for b in blocks_
                            if isinstance(b, tm.SparseMatrix) and getattr(b._dev(), "_ind16", None) is not None)
                by = blk_bytes + (n_local + p) * isz - saved
                mv[name] = {"ms": round(ms, 4), "gbs": round(by / (ms * 1e-3) / 1e9, 1),
                            "frac_hbm": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "algorithmic_bytes": int(by), "sparse_index_bytes": 2 if saved else 4}
            except Exception as e:        # a side measurement: the sandwich line stands without it
                mv[name] = {"error": str(e)[:200]}

    alg_bytes = synth.algorithmic_bytes(mat)
    flops = synth.algorithmic_flops(mat) if isinstance(mat, tm.SplitMatrix) else (
        float(n_local) * p * (p + 1) if isinstance(mat, tm.DenseMatrix) else float(n_local))
    if world > 1:
        tot = torch.tensor([alg_bytes, flops], dtype=torch.float64, device="cuda")
        dist.all_reduce(tot)
        alg_bytes, flops = float(tot[0].item()), float(tot[1].item())

    result = None
    if rank == 0:
        bd = kernel_breakdown(mat, d)
        dom = max(bd, key=bd.get)
        dom_ms = bd[dom]
        dom_bytes = op_algorithmic_bytes(mat, dom)
        dom_flops = op_flops(mat, dom)
        hbm_time = dom_bytes / (HBM_PEAK_GBS * 1e9)
        peak_tf = MFMA_F64_PEAK_TFLOPS if tdt == torch.float64 else MFMA_F32_PEAK_TFLOPS
        mfma_note = None
        if (dom_flops and tdt == torch.float32 and isinstance(mat, tm.DenseMatrix) and 128 < p <= 256
                and p % 4 == 0):
            # the f32 syrk of 129..256 columns runs as SIX bf16 piece products on the bf16 matrix cores
            # (csrc/syrk_bf16.hip): priced against the dense bf16 peak with the flops it really issues
            dom_flops *= 6.0
            peak_tf = MFMA_BF16_PEAK_TFLOPS
            mfma_note = ("f32 syrk as 6 bf16 piece products (3-piece split, f32 accumulation): achieved = "
                         "6 x n k (k + 1) bf16 flop / kernel time, peak = dense bf16 MFMA")
        mfma_time = (dom_flops / (peak_tf * 1e12)) if dom_flops else 0.0
        dom_lds = op_lds_bytes(mat, dom)
        lds_time = (dom_lds / (LDS_PEAK_GBS * 1e9)) if dom_lds else 0.0
        # the bounding roofline of the dominant kernel ("hbm" | "mfma"): MFMA when its matrix time
        # exceeds its HBM time, else HBM -- `frac` is against THAT bound.  A kernel whose inner loop is an LDS
        # gather additionally carries `frac_lds` (its algorithmic LDS bytes against the guide's aggregate LDS
        # rate): a second opinion on what limits it, never `frac`.
        if mfma_time > hbm_time:
            roof = {"bound": "mfma", "achieved": round(dom_flops / (dom_ms * 1e-3) / 1e12, 3),
                    "peak": peak_tf, "unit": "TFLOP/s"}
        else:
            roof = {"bound": "hbm", "achieved": round(dom_bytes / (dom_ms * 1e-3) / 1e9, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s"}
        roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
        # (the same kernel against the HBM roofline, whatever binds it)
        roof["frac_hbm"] = round(dom_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if dom_lds:
            roof["frac_lds"] = round(dom_lds / (dom_ms * 1e-3) / 1e9 / LDS_PEAK_GBS, 4)
            roof["algorithmic_lds_bytes_per_launch"] = int(dom_lds)
            roof["lds_peak_gbs"] = LDS_PEAK_GBS
        # HBM bytes per launch of that kernel, measured NOW: rocprofv3 --pmc passes over a child
        # run that launches the same op (null when rocprofv3 is unavailable or N > 1)
        roof["traffic"] = None
        if world == 1 and not args.no_traffic:
            roof["traffic"] = measure_traffic(args, dom)
        # the dense self-sandwich: against the matrix spec of its dtype when it ran on that dtype's MFMA; when it
        # ran in 40-bit fixed point on the INT8 matrix cores (K1e) an f64-equivalent flop rate against the f64
        # spec would not be a utilisation: it is priced against HBM (it streams the block once) and by the int8
        # operations it issues (22 digit-pair products of n k (k + 1) / 2 MACs each, padded to 128 columns)
        mfma_frac, dense_self = None, None
        i8_ran = bool(dense_term and dense_term.startswith("int8"))
        for name, ms in bd.items():
            fl = op_flops(mat, name)
            if not fl:
                continue
            if i8_ran:
                by = op_algorithmic_bytes(mat, name)
                kk = [m for m in (mat.matrices if isinstance(mat, tm.SplitMatrix) else [mat])
                      if isinstance(m, tm.DenseMatrix)][0].shape[1]
                kp = -(-kk // 128) * 128 if kk <= 128 else kk
                dense_self = {"kernel": name, "ms": round(ms, 4), "arithmetic": "int8x5 digits, 22 digit pairs, int32 acc",
                              "frac_hbm": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "int8_tops": round(22.0 * n_local * kp * (kp + 16) / (ms * 1e-3) / 1e12, 1),
                              "int8_peak_tops_dense": INT8_PEAK_TOPS}
            else:
                mfma_frac = round(fl * (6.0 if mfma_note else 1.0) / (ms * 1e-3) / (peak_tf * 1e12), 4)
        if mfma_note:
            roof["note"] = mfma_note
        roof["kernel"] = dom
        roof["kernel_ms"] = round(dom_ms, 4)
        roof["algorithmic_bytes_per_launch"] = int(dom_bytes)
        wl_names = {
            "cfg4": f"SplitMatrix.sandwich: dense128 + csr512@5% + cats(256,96,32), {n_local} rows/GPU, p={p}, float64 (BASELINE configs[3])",
            "cfg2": f"DenseMatrix.sandwich float32 {n_local}x256 (BASELINE configs[1])",
            "cfg3": f"CategoricalMatrix.sandwich {n_local} rows x 10k categories float64 (BASELINE configs[2])",
            "cfg1": f"DenseMatrix.sandwich float64 {n_local}x64 (BASELINE configs[0])",
        }
        if args.workload not in wl_names:
            wl_names[args.workload] = (f"reference benchmark design '{args.workload}' "
                                       f"(benchmark/generate_matrices.py:90-100): {type(mat).__name__} "
                                       f"{n_local} x {p}, float64")
        result = {
            "metric": METRIC,
            "value": round(alg_bytes / (ms_per_step * 1e-3) / 1e9, 2),
            "unit": "GB/s",
            "gflops": round(flops / (ms_per_step * 1e-3) / 1e9, 1),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if tdt == torch.float64 else "f32",
            "data": "synthetic",
            "config": {"workload": wl_names[args.workload], "rows_per_gpu": n_local, "p": p,
                       "launch": "hipgraph replay" if use_graph else "eager",
                       "sharding": "rows" if world > 1 else "none",
                       "collective": "all_reduce(p*p f64)" if world > 1 else "none"},
            "roofline": roof,
            "ms_per_step_min": round(ms_min, 4),
            "hbm_frac_whole_job": round(alg_bytes / world / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            # whole step against max(HBM, MFMA): bytes / 8 TB/s vs flops / dense matrix peak of the dtype
            "mixed_roofline_ms": round(max(alg_bytes / world / (HBM_PEAK_GBS * 1e9),
                                           flops * (6.0 if mfma_note else 1.0) / world / (peak_tf * 1e12)) * 1e3, 4),
            "mfma_frac_of_spec": mfma_frac,
            "dense_self": dense_self,
            "sum_kernel_ms": round(sum(bd.values()), 4),
            "matvec": mv.get("matvec"),
            "transpose_matvec": mv.get("transpose_matvec"),
            "dense_term": dense_term,
            "dense_term_handover": handovers,
            "ms_per_step_f64_only": None if ms_f64_only is None else round(ms_f64_only, 4),
            "ms_per_step_hipgraph": None if ms_graph is None else round(ms_graph, 4),
        }
        if args.workload == "cfg4":
            scale = n_local / 10_000_000
            floor = sum(v[0] for v in DESIGN_FLOOR_CFG4_MS.values()) * scale
            result["design_floor_ms"] = round(floor, 2)
            result["design_floor"] = {
                "what": ("per-GPU step time of the CURRENT formulation (one kernel per block product) with every "
                         "kernel at the floor its own ablation shows; the north star's 0.60 of HBM peak would be "
                         f"{alg_bytes / world / (0.6 * HBM_PEAK_GBS * 1e9) * 1e3:.2f} ms, below the f64 arithmetic of the "
                         "dense term at any rate this chip has (SURVEY.md 8d)"),
                "terms_ms_at_10M_rows": {k: {"ms": v[0], "from": v[1]} for k, v in DESIGN_FLOOR_CFG4_MS.items()},
                "step_over_floor": round(ms_per_step / floor, 3),
            }
        # ingest as a measured quantity: to_device() = every twin the three products stream, built
        # on the device; resident_bytes = live HBM allocations once sandwich, matvec and transpose_matvec have all run
        # (data + twins + index uploads; workspaces included, the allocator's cached free blocks not)
        result.update(ingest)
        result["resident_bytes"] = int(torch.cuda.memory_allocated())
        result["ingest_note"] = ("ingest_ms: first to_device() of this process (on a fresh box it includes one-off code "
                                 "object loads); ingest_rebuild_ms: the same twins built again on fresh wrappers of the "
                                 "same arrays (steady state); ingest_peak_bytes: peak HBM during the first build, the "
                                 "data included; resident_bytes: live allocations after all three products have run")
        if world > 1:
            result["ranks"] = {"backend": backend, "world_size": world,
                               "rccl_ranks": world if backend == "nccl" else 0,
                               "ms_per_step_by_rank": rank_ms, "ms_per_step_min_rank": min(rank_ms),
                               "ms_per_step_max_rank": max(rank_ms)}
        if not args.no_cpu_baseline and world == 1:   # the CPU leg runs on rank 0 at N = 1 only
            cpu_rows = args.cpu_rows or {"cfg4": 1_000_000, "cfg2": 2_000_000,
                                         "cfg3": 50_000_000}.get(args.workload, 1_000_000)
            result["cpu_baseline"] = cpu_baseline(args.workload, min(cpu_rows, n_local), mat, d)
        if args.breakdown:
            for k, v in sorted(bd.items(), key=lambda kv: -kv[1]):
                print(f"  {k:24s} {v:9.4f} ms", file=sys.stderr)
        result_full = dict(result, breakdown_ms={k: round(v, 4) for k, v in bd.items()})
        if args.out:
            os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
            with open(args.out, "w") as f:
                json.dump(result_full, f, indent=1)
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
