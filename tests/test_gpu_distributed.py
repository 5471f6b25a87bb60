"""Row-sharded products with the REAL HIP blocks: world_size 2 on one GPU (gloo process group,
both ranks on cuda:0), shards cut by tabmat_amd.distributed.shard(); the all-reduced results
must equal the single-rank HIP result and the oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q, device_resident, backend="gloo", ndev=1):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank % ndev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        import _cases as cs
        from _gpu_util import to_tm_split
        from oracle import oracle as orc
        from tabmat_amd.distributed import shard

        n = 30_011
        specs, idx = cs.mixed_specs(n, 24, 70, (11, 5, 3), seed=11)
        full = to_tm_split(specs, idx)
        if device_resident:
            full.to_device()          # shards are then cut in HBM
        blocks = [cs.to_oracle_block(s) for s in specs]
        rng = np.random.default_rng(5)
        d = rng.random(n)
        w = rng.standard_normal(n)
        v = rng.standard_normal(full.shape[1])
        rows_g = np.sort(rng.choice(n, n // 3, replace=False))
        sh = shard(full, always_reduce=True)     # world 1: the collective is issued all the same
        lo, hi = sh.bounds
        assert sh.local.shape[0] == hi - lo
        got = sh.sandwich_global(d)
        got_rows = sh.sandwich_global(d, rows_g)
        got_dev = sh.sandwich(torch.from_numpy(d[lo:hi]).cuda())        # device in -> device out
        got_tmv = sh.transpose_matvec_global(w)
        got_mv = sh.matvec(v)
        want = orc.split_sandwich(blocks, idx, d)
        want_rows = orc.split_sandwich(blocks, idx, d, rows_g)
        want_tmv = orc.split_transpose_matvec(blocks, idx, w)
        want_mv = orc.split_matvec(blocks, idx, v)[lo:hi]
        single = full.sandwich(d)

        def rel(a, b):
            return float(np.abs(np.asarray(a) - b).max() / max(np.abs(b).max(), 1e-300))

        errs = dict(sand=rel(got, want), rows=rel(got_rows, want_rows),
                    dev=rel(got_dev.cpu().numpy(), want), tmv=rel(got_tmv, want_tmv),
                    mv=rel(got_mv, want_mv), single=rel(got, single))
        q.put((rank, errs, (lo, hi)))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("device_resident", [False, True])
def test_world2_hip_shards_match_single_rank_and_oracle(device_resident):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, device_resident)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    bounds = sorted(r[2] for r in res)
    assert bounds[0][0] == 0 and bounds[0][1] == bounds[1][0] and bounds[1][1] == 30_011
    for rank, errs, _ in res:
        for k, e in errs.items():
            assert e < 1e-10, (rank, k, e)


def _run(world, device_resident, backend, ndev):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, device_resident, backend, ndev))
             for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, errs, _ in res:
        for k, e in errs.items():
            assert e < 1e-10, (rank, k, e)
    return res


@pytest.mark.gpu
def test_rccl_all_reduce_world1():
    """backend "nccl" (= RCCL) on one GPU: the communicator is created and the all-reduce of the
    p x p device result and of the numpy-convention result run through RCCL (world size 1 -- the
    code path that the 8-GPU job takes, minus the xGMI transfers)."""
    res = _run(1, True, "nccl", 1)
    assert res[0][2] == (0, 30_011)


@pytest.mark.gpu
def test_rccl_all_reduce_world2_two_devices():
    """Two ranks on two GPUs over RCCL, when the box has two."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs (RCCL refuses two ranks on one device)")
    _run(2, True, "nccl", 2)


@pytest.mark.gpu
def test_bench_two_ranks_gloo():
    """`python bench.py --gpus 2` end to end on ONE GPU: the script spawns its own two ranks through
    torch.distributed.run, the ranks share the device over the gloo backend (TABMAT_BENCH_BACKEND=gloo; RCCL
    refuses two ranks on one device), every rank times its own shard between barriers and rank 0 prints the
    one JSON line with the whole-job value (the 8-GPU job runs the same code over RCCL)."""
    import json
    import subprocess

    env = dict(os.environ, TABMAT_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "200000",
                          "--steps", "3", "--warmup", "1", "--no-traffic", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["scaling"] == "weak"
    assert r["config"]["sharding"] == "rows" and r["config"]["collective"].startswith("all_reduce")
    assert r["config"]["rows_per_gpu"] == 200000
    assert np.isfinite(r["value"]) and r["value"] > 0 and r["ms_per_step"] > 0
    assert r["roofline"]["kernel_ms"] > 0
