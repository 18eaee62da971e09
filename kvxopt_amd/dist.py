"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" = RCCL on ROCm).

Two ways the factor/solve path shards (multi-GPU):
  * independent systems -- `shard()`: every rank factors its own systems, no data-path collective
    (bench.py --gpus N, weak scaling);
  * ONE system over the ranks -- `DistFactor`: proportional mapping of the elimination tree gives every front a
    contiguous range of ranks.  Subtrees mapped to one rank are factored and solved by it alone; the big fronts of
    the top separators are block-cyclic over their range (the owner of a pivot block factors the panel and
    broadcasts it, every rank updates the column blocks it owns); small shared fronts are replicated on their range.
    Exchange steps: broadcasts of child update matrices / panels / update vectors inside a front's range, one
    all-reduce of x per solve, one MIN per factorisation.  The library decides what travels and packs it
    (`kvx_chol_dist_*`, include/kvxhip.h); the collectives themselves run here through torch.distributed
    (backend "nccl" = RCCL over xGMI on a node; gloo in tests/test_dist_gpu.py, where the ranks share one GPU).
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


def partition(factor, nranks, ob=0, min_m=0):
    """Host-only: the map of `factor` (kvxopt_amd.chol.Factor) over `nranks` ranks -- dict with, per front, the rank range
    [glo, ghi) and mode (1 = block-cyclic), per rank the factorisation flops it executes (`rank_flops`) and the panel part
    of them (`panel_flops`), and the totals (`flops` = sum_j c_j^2, `replicated` = flops of the small replicated fronts)."""
    import ctypes

    import numpy as np

    from ._lib import f64p, lib, raise_for
    ns = max(int(factor.info()["nsuper"]), 1)
    glo = np.zeros(ns, dtype=np.int32); ghi = np.zeros(ns, dtype=np.int32); mode = np.zeros(ns, dtype=np.uint8)
    rf = np.zeros(nranks); pf = np.zeros(nranks); tot = np.zeros(2)
    i32p = ctypes.POINTER(ctypes.c_int32)
    raise_for(lib().kvx_chol_dist_map(factor._h, int(nranks), int(ob), int(min_m), glo.ctypes.data_as(i32p), ghi.ctypes.data_as(i32p),
                                      mode.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), rf.ctypes.data_as(f64p),
                                      pf.ctypes.data_as(f64p), tot.ctypes.data_as(f64p)), "partition failed")
    ns = int(factor.info()["nsuper"])
    return {"glo": glo[:ns], "ghi": ghi[:ns], "mode": mode[:ns], "rank_flops": rf, "panel_flops": pf,
            "flops": float(tot[0]), "replicated": float(tot[1])}


def _ptr(x):
    """Device address of a torch tensor, a _lib.DeviceBuffer or a plain integer."""
    if hasattr(x, "data_ptr"):
        return int(x.data_ptr())
    if hasattr(x, "ptr"):
        return int(x.ptr)
    return int(x)


class DistFactor:
    """One SPD system factored and solved by all ranks (one GPU each); see the module docstring.
    Every rank passes the same matrix.  Two transports for the collectives:
      * comm = kvxopt_amd.rccl.World -- RCCL bound directly: the callback is the C function kvx_rccl_comm, the process never imports
        torch and stays on the system HIP runtime; values / right-hand sides are device addresses (_lib.DeviceBuffer or int);
      * comm = None -- torch.distributed (`group`; "nccl" = RCCL through torch, "gloo" in the one-GPU rehearsals of
        tests/test_dist_gpu.py); values / right-hand sides are torch tensors on the rank's device.
    ob / min_m: column-block width and smallest order of the block-cyclic fronts (0 = library defaults 512 / 6144)."""

    def __init__(self, n, colptr, rowind, uplo="L", perm=None, opts=None, group=None, device=None, ob=0, min_m=0, xchg_doubles=0, comm=None):
        import ctypes

        import numpy as np

        from . import _lib
        from ._lib import lib, raise_for
        from .chol import Factor
        self._world_obj = comm
        self.group = group
        if comm is not None:
            self._dist, self._torch = None, None
            self.rank, self.world = int(comm.rank), int(comm.world)
            self.device = comm.device
        else:
            import torch
            import torch.distributed as dist
            self._torch = torch
            self._dist = dist if dist.is_available() and dist.is_initialized() else None
            self.rank = self._dist.get_rank(group) if self._dist else 0
            self.world = self._dist.get_world_size(group) if self._dist else 1
            self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.F = Factor(n, colptr, rowind, uplo, perm, opts)
        self.n = int(n)
        info = np.zeros(8, dtype=np.int64)
        with self._on_device():
            raise_for(lib().kvx_chol_dist_setup(self.F._h, self.rank, self.world, int(ob), int(min_m),
                                                info.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))), "sharded setup failed")
        self.nshared, self.ncyclic, self.ob, self.min_m = int(info[3]), int(info[4]), int(info[6]), int(info[7])
        # one communicator per distinct rank range of the map (the same list, in the same order, on every rank)
        ng = int(info[5])
        lohi = np.zeros(max(2 * ng, 2), dtype=np.int32)
        if ng:
            raise_for(lib().kvx_chol_dist_groups(self.F._h, lohi.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))
        cnt = max(int(info[1]), int(xchg_doubles) if xchg_doubles else int(info[2]))
        self.collectives = 0
        self.bytes_moved = 0
        self._err = None
        if comm is not None:
            for i in range(ng):
                comm.split(int(lohi[2 * i]), int(lohi[2 * i + 1]))
            self._xchg = _lib.DeviceBuffer(8 * cnt)
            raise_for(lib().kvx_chol_dist_set_xchg(self.F._h, self._xchg.ptr, cnt))
            self.backend = comm.backend
            self._cb, self._ctx = comm.comm_fn, comm.comm_ctx
            self._stats0 = comm.stats()
        else:
            torch = self._torch
            self._ranks = list(range(self.world)) if (self._dist is None or group is None) else self._dist.get_process_group_ranks(group)
            self._groups = {}
            for i in range(ng):
                lo, hi = int(lohi[2 * i]), int(lohi[2 * i + 1])
                if (lo, hi) == (0, self.world):
                    self._groups[(lo, hi)] = group
                elif self._dist is not None:
                    self._groups[(lo, hi)] = self._dist.new_group([self._ranks[r] for r in range(lo, hi)])
            self._groups.setdefault((0, self.world), group)
            self._xchg = torch.zeros(cnt, dtype=torch.float64, device=self.device)
            raise_for(lib().kvx_chol_dist_set_xchg(self.F._h, self._xchg.data_ptr(), cnt))
            self._base = self._xchg.data_ptr()
            # RCCL collectives are enqueued in stream order (torch puts them behind the current = null stream, which do_comm has put
            # behind the factor's stream): no host synchronisation.  gloo stages through the host: settle the producers first.
            # KVX_DIST_STREAM_ORDERED=1 rehearses the stream-ordered branch over gloo (torch's gloo backend orders its staging
            # copies against the current stream itself), so the event ordering around the callback runs on a one-GPU box too.
            self.backend = self._dist.get_backend(group) if self._dist is not None else "none"
            self._host_staged = self._dist is not None and self.backend != "nccl" and os.environ.get("KVX_DIST_STREAM_ORDERED") != "1"
            self._cb, self._ctx = _lib.DIST_COMM_FN(self._comm), None      # keep the callback object alive as long as the factor
        inf = self.F.info()
        self.dev_bytes = int(inf["dev_bytes"])                  # this rank's large device buffers (panels, inverted blocks, update matrices, values)
        self.lsize_local, self.lsize_total = int(inf["lsize_local"]), int(inf["lsize"])

    def _on_device(self):
        import contextlib
        return self._torch.cuda.device(self.device) if self._torch is not None else contextlib.nullcontext()

    def _sync_stats(self):
        if self._world_obj is not None:
            c, b = self._world_obj.stats()
            self.collectives, self.bytes_moved = c - self._stats0[0], b - self._stats0[1]

    def _comm(self, ctx, op_p):
        """One collective on a slice of the exchange buffer, in the current (null) stream's order (torch.distributed transport)."""
        try:
            torch = self._torch
            op = op_p.contents
            if self._dist is None or self.world == 1:
                return 0
            off = (int(op.buf_dev) - self._base) // 8
            t = self._xchg[off:off + int(op.count)]
            g = self._groups[(int(op.lo), int(op.hi))]
            if self._host_staged:                                   # gloo stages through the host: settle the producers first
                torch.cuda.current_stream().synchronize()
            if op.kind == 1:
                self._dist.broadcast(t, src=self._ranks[int(op.root)], group=g)
            elif op.kind == 2:
                self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=g)
            elif op.kind == 3:
                self._dist.all_reduce(t, op=self._dist.ReduceOp.MIN, group=g)
            else:
                raise ValueError("unknown collective kind %d" % op.kind)
            if self._host_staged:
                torch.cuda.current_stream().synchronize()
            self.collectives += 1
            self.bytes_moved += 8 * int(op.count)
            return 0
        except BaseException as e:                                  # never let an exception cross the C frames
            self._err = e
            return 1

    def _check(self, rc, what):
        from ._lib import raise_for
        if self._err is not None:
            e, self._err = self._err, None
            raise e
        raise_for(rc, what)

    def factorize(self, values_dev):
        """values_dev: the nnz doubles of the analysed triangle on this rank's device (tensor / DeviceBuffer / address), identical
        on all ranks.  Raises ArithmeticError(failing column) on every rank when the matrix is not positive definite."""
        import ctypes

        from ._lib import KVX_ENOTPOSDEF, lib
        minor = ctypes.c_int64(self.n)
        with self._on_device():
            rc = lib().kvx_chol_dist_factorize(self.F._h, _ptr(values_dev), self._cb, self._ctx, ctypes.byref(minor))
        self._sync_stats()
        if rc == KVX_ENOTPOSDEF and self._err is None:
            raise ArithmeticError(int(minor.value))
        self._check(rc, "factorization failed")

    def solve(self, B_dev, nrhs=1, ldB=None):
        """Solve A X = B in place; B_dev: column-major n x nrhs doubles (ld = ldB) on this rank's device, identical on all ranks;
        every rank ends with the full solution."""
        from ._lib import lib
        ldB = self.n if ldB is None else int(ldB)
        with self._on_device():
            rc = lib().kvx_chol_dist_solve(self.F._h, _ptr(B_dev), int(nrhs), ldB, self._cb, self._ctx)
        self._sync_stats()
        self._check(rc, "solve failed")
