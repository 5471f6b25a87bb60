"""N > 1 host logic on CPU: world_size-2 gloo process group, row sharding + all-reduce of the
p x p sandwich / length-p transpose_matvec.  The local products are the CPU oracle here (the HIP
kernels need a GPU); on the GPU box the same wrapper runs the HIP path over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _cases as cs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q, rows_mode="spread"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        from tabmat_amd.distributed import RowShardedMatrix, bucket_rows, shard_bounds

        n = 1001
        specs, idx = cs.mixed_specs(n, 8, 20, (6, 4), seed=7)
        blocks = [cs.to_oracle_block(s) for s in specs]
        rng = np.random.default_rng(0)
        d = rng.random(n)
        w = rng.standard_normal(n)
        rows_g = np.sort(rng.choice(n, 600, replace=False))
        if rows_mode == "low":            # every selected row in the first three shards of eight: the others get none
            rows_g = np.sort(rng.choice(3 * (n // 8), 200, replace=False))
        full = orc.split_sandwich(blocks, idx, d)
        full_rows = orc.split_sandwich(blocks, idx, d, rows_g)
        full_tmv = orc.split_transpose_matvec(blocks, idx, w)

        lo, hi = shard_bounds(n, world, rank)
        local_specs = []
        for s in specs:
            if s[0] == "dense":
                local_specs.append(("dense", np.ascontiguousarray(s[1][lo:hi])))
            elif s[0] == "sparse":
                local_specs.append(("sparse", s[1].tocsr()[lo:hi].tocsc()))
            else:
                local_specs.append(("cat", s[1][lo:hi], s[2], s[3]))
        lblocks = [cs.to_oracle_block(s) for s in local_specs]

        class Local:
            shape = (hi - lo, sum(len(i) for i in idx))
            dtype = np.dtype(np.float64)

        sh = RowShardedMatrix(
            Local(),
            local_sandwich=lambda dd, rows, cols: orc.split_sandwich(lblocks, idx, dd, rows, cols),
            local_transpose_matvec=lambda vv, rows, cols: orc.split_transpose_matvec(
                lblocks, idx, vv, rows, cols))
        got = sh.sandwich(d[lo:hi])
        got_rows = sh.sandwich(d[lo:hi], bucket_rows(rows_g, lo, hi))
        got_tmv = sh.transpose_matvec(w[lo:hi])
        # the collective started, another product issued meanwhile, then waited for
        pending = sh.sandwich_async(d[lo:hi])
        tmv2 = sh.transpose_matvec(w[lo:hi])
        got_async = pending.wait()
        ok = (np.allclose(got, full, rtol=1e-12, atol=1e-12)
              and np.allclose(got_async, full, rtol=1e-12, atol=1e-12)
              and np.allclose(tmv2, full_tmv, rtol=1e-12, atol=1e-12)
              and np.allclose(got_rows, full_rows, rtol=1e-12, atol=1e-12)
              and np.allclose(got_tmv, full_tmv, rtol=1e-12, atol=1e-12))
        q.put((rank, bool(ok), (lo, hi)))
    finally:
        dist.destroy_process_group()


def test_shard_bounds_cover():
    from tabmat_amd.distributed import shard_bounds

    for n in (0, 1, 7, 1000, 1001):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_row_sharded_sandwich_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res


def test_row_sharded_sandwich_gloo_world8_uneven_rows_and_empty_row_lists():
    """The shape of the 8-GPU job (BASELINE configs[4]) on CPU: eight ranks, n = 1001 rows (not divisible by 8:
    one rank gets an extra row), and a `rows=` list that leaves five of the eight shards without a single selected
    row -- their partial is all zeros and still takes part in the all-reduce (a rank that skipped the collective
    would hang the job)."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, "low")) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    bounds = sorted(b for _, _, b in res)
    assert bounds[0][0] == 0 and bounds[-1][1] == 1001
    assert sorted(h - l for l, h in bounds) == [125] * 7 + [126]
