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


def _system(kind, g):
    from kvxopt_amd import workloads
    return workloads.laplacian_2d(g) if kind == "2d" else workloads.laplacian_3d(g)


@pytest.mark.parametrize("kind,g,nranks,ob,min_m", [("2d", 60, 2, 64, 128), ("2d", 60, 3, 64, 128), ("2d", 90, 8, 64, 192),
                                                   ("2d", 7, 4, 64, 128), ("3d", 18, 4, 64, 192), ("3d", 24, 8, 128, 256),
                                                   ("3d", 24, 6, 512, 6144)])
def test_proportional_mapping_properties(kind, g, nranks, ob, min_m):
    """kvx_chol_dist_map (host only): rank ranges are nested along the elimination tree, a range of one rank owns a whole
    subtree, the roots share all ranks, block-cyclic fronts are shared big fronts of order >= min_m, the per-rank flop
    counts add up to the factorisation's flops (+ the replicated fronts once more per extra rank), and the map is
    deterministic."""
    from kvxopt_amd import dist as kd
    from kvxopt_amd.chol import Factor
    F = Factor(*_system(kind, g)[:3])
    sup, nrows, parent, level = F.supernodes()
    M = kd.partition(F, nranks, ob, min_m)
    glo, ghi, mode = M["glo"].astype(int), M["ghi"].astype(int), M["mode"]
    assert np.all((0 <= glo) & (glo < ghi) & (ghi <= nranks))
    has_p = parent >= 0
    assert np.all(glo[has_p] >= glo[parent[has_p]]) and np.all(ghi[has_p] <= ghi[parent[has_p]])     # nested
    single_parent = has_p & ((ghi - glo)[np.where(has_p, parent, 0)] == 1)
    assert np.all(glo[single_parent] == glo[parent[single_parent]])                                   # whole subtrees
    roots = ~has_p
    if roots.sum() == 1:
        assert glo[roots][0] == 0 and ghi[roots][0] == nranks
    k = np.diff(sup).astype(float); m = nrows.astype(float)
    shared = (ghi - glo) > 1
    assert np.all(~mode.astype(bool) | (shared & (nrows >= min_m) & ((nrows > 128) | (k > 64)) & (k >= min(ob, 256))))
    S2 = lambda x: x * (x + 1) * (2 * x + 1) / 6.0
    f = S2(m) - S2(m - k)
    # (the fronts' flops: >= the sum_j c_j^2 of the unamalgamated factor that info()["flops"] reports)
    assert f.sum() >= F.info()["flops"] * (1 - 1e-12) and abs(M["flops"] - f.sum()) <= 1e-9 * f.sum()
    repl = shared & ~mode.astype(bool)
    assert abs(M["replicated"] - f[repl].sum()) <= 1e-9 * max(f.sum(), 1)
    expect = f.sum() + (f[repl] * ((ghi - glo)[repl] - 1)).sum()
    assert abs(M["rank_flops"].sum() - expect) <= 1e-9 * expect
    assert np.all(M["panel_flops"] <= M["rank_flops"] + 1e-6)
    # every rank gets work when there are at least as many leaves as ranks; nobody exceeds the total
    assert M["rank_flops"].max() <= f.sum() * (1 + 1e-12)
    M2 = kd.partition(F, nranks, ob, min_m)
    assert all(np.array_equal(M[key], M2[key]) for key in ("glo", "ghi", "mode", "rank_flops"))
    M1 = kd.partition(F, 1, ob, min_m)
    assert np.all(M1["glo"] == 0) and np.all(M1["ghi"] == 1) and not M1["mode"].any()
    assert abs(M1["rank_flops"][0] - f.sum()) <= 1e-9 * f.sum()


def test_sharding_of_a_3d_grid_balances_the_flops():
    """Config 5's structure at a size the CPU suite affords (7-point Laplacian 32^3): with block-cyclic top fronts the most
    loaded of 4 ranks executes < 40 % of the flops (the replicated design of round 1 could not go below the top's share)
    and the fronts replicated on several ranks hold < 10 % of them."""
    from kvxopt_amd import dist as kd
    from kvxopt_amd import workloads
    from kvxopt_amd.chol import Factor
    F = Factor(*workloads.laplacian_3d(32)[:3])
    M = kd.partition(F, 4, 128, 384)
    share = M["rank_flops"] / M["flops"]
    assert share.max() < 0.40 and M["replicated"] / M["flops"] < 0.10, (share, M["replicated"] / M["flops"])
    assert M["mode"].sum() >= 1
