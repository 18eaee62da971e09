"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" = RCCL on ROCm).

Two ways the factor/solve path shards (multi-GPU):
  * independent systems -- `shard()`: every rank factors its own systems, no data-path collective
    (bench.py --gpus N, weak scaling);
  * ONE system over the ranks -- `DistFactor`: the elimination tree is cut below its top separators, every
    rank factors and solves the subtrees it owns, the top of the tree is replicated; the real exchange steps
    are all-reduce sums of (a) the update matrices of the subtree roots, once per factorisation, (b) their
    update vectors and (c) the owned pieces of x, per solve.  The library does the packing
    (`kvx_chol_dist_*`, include/kvxhip.h), torch.distributed the collectives (RCCL on a node; gloo in
    tests/test_dist_gpu.py, where the ranks share the one GPU of the box).
"""
import os


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment; returns (rank, world, dist or None)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world <= 1:
        return 0, 1, None
    import torch
    import torch.distributed as dist
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend, **kw)
    return rank, world, dist


def shard(n_items, rank, world):
    """Contiguous block of `n_items` independent work units owned by `rank` (first ranks get the remainder)."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(value, dist, device="cpu"):
    """Max of a python float over all ranks (the timing rule of bench.py)."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist, device="cpu"):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def partition(factor, nranks):
    """Host-only: (owner per front, cut depth) of the subtree sharding of `factor` (kvxopt_amd.chol.Factor)."""
    import ctypes

    import numpy as np

    from ._lib import lib, raise_for
    ns = factor.info()["nsuper"]
    owner = np.zeros(max(int(ns), 1), dtype=np.int32)
    cut = ctypes.c_int(0)
    raise_for(lib().kvx_chol_dist_owner(factor._h, int(nranks), owner.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                        ctypes.byref(cut)), "partition failed")
    return owner[:ns], cut.value


class DistFactor:
    """One SPD system factored and solved by all ranks of `group` (one GPU each); see the module docstring.
    Every rank passes the same matrix; values / right-hand sides are torch tensors on the rank's device."""

    def __init__(self, n, colptr, rowind, uplo="L", perm=None, opts=None, group=None, device=None):
        import ctypes

        import numpy as np
        import torch
        import torch.distributed as dist

        from ._lib import lib, raise_for
        from .chol import Factor
        self._dist = dist if dist.is_available() and dist.is_initialized() else None
        self.group = group
        self.rank = self._dist.get_rank(group) if self._dist else 0
        self.world = self._dist.get_world_size(group) if self._dist else 1
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.F = Factor(n, colptr, rowind, uplo, perm, opts)
        self.n = int(n)
        info = np.zeros(4, dtype=np.int64)
        with torch.cuda.device(self.device):
            raise_for(lib().kvx_chol_dist_setup(self.F._h, self.rank, self.world, info.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))),
                      "sharded setup failed")
        self.cut, self.ulen, self.wlen = int(info[0]), int(info[1]), int(info[2])
        self._xchg = torch.zeros(max(self.ulen, self.n, 1), dtype=torch.float64, device=self.device)

    def _allreduce(self, count, op=None):
        if self._dist is not None and self.world > 1 and count > 0:
            self._dist.all_reduce(self._xchg[:count], op=op or self._dist.ReduceOp.SUM, group=self.group)

    def _buf(self, count):
        import torch
        if self._xchg.numel() < count:
            self._xchg = torch.zeros(count, dtype=torch.float64, device=self.device)
        return self._xchg

    def factorize(self, values_dev):
        """values_dev: float64 tensor (nnz of the analysed triangle) on this rank's device, identical on all ranks."""
        import ctypes

        import torch

        from ._lib import KVX_ENOTPOSDEF, lib, raise_for
        L = lib()
        minor = ctypes.c_int64(self.n)
        with torch.cuda.device(self.device):
            torch.cuda.current_stream().synchronize()
            raise_for(L.kvx_chol_dist_factor_phase(self.F._h, 0, values_dev.data_ptr(), self._xchg.data_ptr(), None), "factorization failed")
            self._allreduce(self.ulen)
            torch.cuda.current_stream().synchronize()
            rc = L.kvx_chol_dist_factor_phase(self.F._h, 1, values_dev.data_ptr(), self._xchg.data_ptr(), ctypes.byref(minor))
        m = int(minor.value)
        if self._dist is not None and self.world > 1:          # a failing column may sit in another rank's subtree
            t = torch.tensor([m], dtype=torch.int64, device=self.device)
            self._dist.all_reduce(t, op=self._dist.ReduceOp.MIN, group=self.group)
            m = int(t.item())
        if m < self.n:
            raise ArithmeticError(m)
        if rc != KVX_ENOTPOSDEF:
            raise_for(rc, "factorization failed")

    def solve(self, B_dev, nrhs=1, ldB=None):
        """Solve A X = B in place; B_dev: float64 tensor (column-major n x nrhs, ld = ldB) on this rank's device,
        identical on all ranks; every rank ends with the full solution."""
        import torch

        from ._lib import lib, raise_for
        L = lib()
        ldB = self.n if ldB is None else int(ldB)
        need = max(self.wlen, self.n) * int(nrhs)
        buf = self._buf(need)
        with torch.cuda.device(self.device):
            torch.cuda.current_stream().synchronize()
            raise_for(L.kvx_chol_dist_solve_phase(self.F._h, 0, B_dev.data_ptr(), int(nrhs), ldB, buf.data_ptr()), "solve failed")
            self._allreduce(self.wlen * int(nrhs))
            torch.cuda.current_stream().synchronize()
            raise_for(L.kvx_chol_dist_solve_phase(self.F._h, 1, B_dev.data_ptr(), int(nrhs), ldB, buf.data_ptr()), "solve failed")
            self._allreduce(self.n * int(nrhs))
            torch.cuda.current_stream().synchronize()
            raise_for(L.kvx_chol_dist_solve_phase(self.F._h, 2, B_dev.data_ptr(), int(nrhs), ldB, buf.data_ptr()), "solve failed")
