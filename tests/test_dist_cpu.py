"""N > 1 plumbing on CPU: two gloo ranks shard independent systems and reduce timings the way bench.py does."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    from kvxopt_amd import dist as kd
    from kvxopt_amd import workloads
    from kvxopt_amd.chol import Factor
    r, w, d = kd.init("gloo")
    lo, hi = kd.shard(5, r, w)                     # 5 independent systems over 2 ranks -> 3 + 2
    flops = 0.0
    for i in range(lo, hi):
        n, cp, ri, vx = workloads.laplacian_2d(10 + i)
        flops += Factor(n, cp, ri).info()["flops"]          # host-side analysis only (no GPU here)
    tot = kd.sum_over_ranks(flops, d)
    tmax = kd.max_over_ranks(1.0 + r, d)
    d.barrier()
    q.put((r, lo, hi, tot, tmax))
    d.destroy_process_group()


def test_two_rank_gloo_sharding():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 3), (3, 5)]
    assert res[0][3] == res[1][3] > 0 and res[0][4] == res[1][4] == 2.0
    from kvxopt_amd import workloads
    from kvxopt_amd.chol import Factor
    ref = sum(Factor(*workloads.laplacian_2d(10 + i)[:3]).info()["flops"] for i in range(5))
    assert abs(res[0][3] - ref) < 1e-6 * ref


def test_shard_edges():
    from kvxopt_amd.dist import shard
    assert [shard(8, r, 8) for r in range(8)] == [(r, r + 1) for r in range(8)]
    assert [shard(3, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 3), (3, 3)]
    assert shard(0, 0, 2) == (0, 0)
