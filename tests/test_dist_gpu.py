"""Subtree-sharded factor + solve (kvxopt_amd.dist.DistFactor, kvx_chol_dist_*): 2 and 3 ranks over gloo, all on
the one GPU of the box (the collectives are backend-agnostic; a real node uses backend "nccl" = RCCL, one GPU per
rank).  Every rank must end with the solution of the single-process path."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _systems():
    import scipy.sparse as sp
    from kvxopt_amd import workloads
    out = [workloads.laplacian_2d(120), workloads.laplacian_2d(33, 71)]
    M = sp.random(1500, 1500, 0.004, random_state=3, format="csc")
    S = sp.tril((M @ M.T + sp.eye(1500) * 4.0).tocsc()).tocsc(); S.sort_indices()
    out.append((1500, S.indptr.astype(np.int64), S.indices.astype(np.int64), S.data))
    return out


def _worker(rank, world, port, q):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": "0",
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    import torch
    import torch.distributed as dist
    from kvxopt_amd import workloads
    from kvxopt_amd.chol import Factor
    from kvxopt_amd.dist import DistFactor
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    res = []
    for (n, cp, ri, vx) in _systems():
        nrhs = 2
        B = np.asfortranarray(np.random.default_rng(7).standard_normal((n, nrhs)))
        DF = DistFactor(n, cp, ri)
        v_d = torch.from_numpy(vx).to(dev)
        for _ in range(2):                                   # a second numeric factorisation on the same handle
            DF.factorize(v_d)
        b_d = torch.from_numpy(B.reshape(-1, order="F").copy()).to(dev)
        DF.solve(b_d, nrhs)
        X = b_d.cpu().numpy().reshape(n, nrhs, order="F")
        F1 = Factor(n, cp, ri)                               # single-process path, same device
        F1.factorize(vx)
        X1 = B.copy(order="F"); F1.solve(X1)
        R = workloads.sym_matvec(n, cp, ri, vx, X) - B
        res.append((DF.cut, float(np.abs(X - X1).max() / np.abs(X1).max()), float(np.linalg.norm(R) / np.linalg.norm(B))))
        # a non-positive pivot inside ONE rank's subtree is reported by every rank, with the single-process column
        bad = vx.copy(); bad[cp[int(F1.perm()[3])]] = -1.0
        try:
            F1.factorize(bad); ref = None
        except ArithmeticError as e:
            ref = e.args[0]
        try:
            DF.factorize(torch.from_numpy(bad).to(dev)); got = None
        except ArithmeticError as e:
            got = e.args[0]
        res.append(("minor", ref, got))
        DF.factorize(v_d)
    dist.barrier()
    q.put((rank, res))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_factor_solve_matches_single_process(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, res in out:
        for item in res:
            if item[0] == "minor":
                assert item[1] is not None and item[1] == item[2], (rank, item)
            else:
                cut, dx, rr = item
                assert cut >= 1, (rank, item)
                assert dx < 1e-11 and rr < 1e-11, (rank, item)        # same L up to summation order of the root updates
