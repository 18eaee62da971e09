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


@pytest.mark.parametrize("g,nranks", [(60, 2), (60, 3), (90, 8), (7, 4)])
def test_subtree_partition_properties(g, nranks):
    """kvx_chol_dist_owner (host only): every front below the cut belongs to exactly one rank, ownership is
    closed under descendants (a rank owns whole subtrees), the top is unowned, and the work is balanced."""
    from kvxopt_amd import dist as kd
    from kvxopt_amd import workloads
    from kvxopt_amd.chol import Factor
    F = Factor(*workloads.laplacian_2d(g)[:3])
    sup, nrows, parent, level = F.supernodes()
    owner, cut = kd.partition(F, nranks)
    assert 1 <= cut <= level.max() if level.max() > 0 else cut == 0
    assert np.all(owner[level < cut] == -1)
    below = level >= cut
    assert np.all((owner[below] >= 0) & (owner[below] < nranks))
    deeper = level > cut
    assert np.all(owner[deeper] == owner[parent[deeper]])              # whole subtrees
    k = np.diff(sup).astype(float); m = nrows.astype(float)
    w = k * m * m + 1.0
    work = np.array([np.sum(w[below & (owner == r)]) for r in range(nranks)])
    # the cut minimises replicated-top work + the heaviest rank (longest-first assignment), over all depths
    wsub = w.copy()
    for s in range(len(w)):
        if parent[s] >= 0:
            wsub[parent[s]] += wsub[s]
    def cost(d):
        load = np.zeros(nranks)
        for x in sorted(wsub[level == d], reverse=True):
            load[np.argmin(load)] += x
        return w[level < d].sum() + load.max()
    if level.max() > 0:
        costs = {d: cost(d) for d in range(1, level.max() + 1)}
        assert abs(costs[cut] - min(costs.values())) <= 1e-9 * costs[cut]
        assert abs((w[level < cut].sum() + work.max()) - costs[cut]) <= 1e-9 * costs[cut]
    o2, c2 = kd.partition(F, nranks)
    assert c2 == cut and np.array_equal(o2, owner)                     # deterministic
    o1, c1 = kd.partition(F, 1)
    assert c1 == 0 and np.all(o1 == 0)
