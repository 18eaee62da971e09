"""RCCL bound directly (csrc/rccl_comm.cpp, include/kvxhip.h `kvx_rccl_*`): the ranks of the sharded factor without torch.

A process runs on the HIP runtime it loads first; `import torch` makes that the runtime inside the PyTorch wheel, under which
the library's one-enqueue launch graph is unusable and every collective is a C -> Python callback.  A
rank built on this module never imports torch: it picks its GPU (`kvx_set_device`), receives rank 0's 128-byte RCCL id over a
TCP socket on MASTER_ADDR:MASTER_PORT (the rendezvous the launcher already names), and hands `kvx_rccl_comm` -- a C function
-- to `kvx_chol_dist_factorize / solve` as the collective callback.  One rank per device (RCCL refuses two).

Launch: one process per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (torchrun's environment;
`bench.py --gpus N --comm rccl` starts the ranks itself)."""
import ctypes
import os
import socket
import struct
import time

import numpy as np

from . import _lib
from ._lib import f64p, lib, raise_for


def _exchange_id_file(rank, world, port, make_id, timeout=120.0):
    """Ranks of one node: rank 0 writes the id to a file named after the launcher they share (parent process id + rendezvous port:
    under torchrun MASTER_PORT itself is held by the agent's store, so a socket there is not ours to bind), the others poll it."""
    path = os.environ.get("KVX_RCCL_ID_FILE") or os.path.join(os.environ.get("TMPDIR", "/tmp"), "kvx_rccl_id.%d.%d" % (os.getppid(), port))
    if rank == 0:
        uid = make_id()
        tmp = path + ".tmp.%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)                          # (atomic: a reader sees nothing or all 128 bytes)
        return uid, path
    t_end = time.time() + timeout
    while True:
        try:
            with open(path, "rb") as f:
                buf = f.read()
            if len(buf) == 128:
                return buf, None
        except FileNotFoundError:
            pass
        if time.time() > t_end:
            raise TimeoutError("no RCCL id from rank 0 at %s" % path)
        time.sleep(0.02)


def _exchange_id(rank, world, addr, port, make_id, timeout=120.0):
    """Rank 0 serves the id to world - 1 clients on (addr, port); the others fetch it (retrying until rank 0 listens)."""
    if world == 1:
        return make_id()
    if rank == 0:
        uid = make_id()
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr if addr not in ("localhost",) else "127.0.0.1", port))
        srv.listen(world)
        srv.settimeout(timeout)
        seen = set()
        try:
            while len(seen) < world - 1:
                c, _ = srv.accept()
                with c:
                    r = struct.unpack("<i", c.recv(4, socket.MSG_WAITALL))[0]
                    c.sendall(uid)
                    seen.add(r)
        finally:
            srv.close()
        return uid
    t_end = time.time() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as c:
                c.sendall(struct.pack("<i", rank))
                buf = b""
                while len(buf) < 128:
                    part = c.recv(128 - len(buf))
                    if not part:
                        raise ConnectionError("rank 0 closed the id socket early")
                    buf += part
                return buf
        except (ConnectionRefusedError, ConnectionResetError, socket.timeout, OSError):
            if time.time() > t_end:
                raise
            time.sleep(0.05)


class World:
    """The ranks of one job, one GPU each, talking through librccl.so.  No torch in the process."""

    def __init__(self, rank=None, world=None, local_rank=None, addr=None, port=None):
        env = os.environ
        self.rank = int(env.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(env.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.local_rank = int(env.get("LOCAL_RANK", str(self.rank))) if local_rank is None else int(local_rank)
        addr = env.get("MASTER_ADDR", "127.0.0.1") if addr is None else addr
        port = int(env.get("MASTER_PORT", "29500")) if port is None else int(port)
        _lib.require_device()
        ndev = max(int(lib().kvx_device_count()), 1)
        raise_for(lib().kvx_set_device(self.local_rank % ndev), "cannot select the rank's GPU")
        self.device = self.local_rank % ndev

        def make_id():
            buf = ctypes.create_string_buffer(128)
            raise_for(lib().kvx_rccl_unique_id(buf), "ncclGetUniqueId failed")
            return buf.raw
        # KVX_RCCL_RENDEZVOUS=tcp: the id over a socket on MASTER_ADDR:MASTER_PORT + 1 (ranks on several nodes); default: a file
        # in TMPDIR named after the common launcher (ranks of one node, the contract of bench.py)
        id_file = None
        if self.world == 1:
            uid = make_id()
        elif env.get("KVX_RCCL_RENDEZVOUS", "file") == "tcp":
            uid = _exchange_id(self.rank, self.world, addr, port + 1, make_id)
        else:
            uid, id_file = _exchange_id_file(self.rank, self.world, port, make_id)
        h = ctypes.c_void_p()
        raise_for(lib().kvx_rccl_init(self.rank, self.world, uid, ctypes.byref(h)), "ncclCommInitRank failed")
        self._h = h
        if id_file:                                    # every rank has joined (ncclCommInitRank is collective): the file has served
            try:
                os.unlink(id_file)
            except OSError:
                pass
        v = ctypes.c_int(0)
        raise_for(lib().kvx_rccl_version(ctypes.byref(v)))
        self.version = int(v.value)
        self.backend = "rccl-direct"
        # the callback the sharded factor calls for every collective: the C function itself, ctx = the communicator
        self.comm_fn = ctypes.cast(lib().kvx_rccl_comm, _lib.DIST_COMM_FN)
        self.comm_ctx = self._h

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib().kvx_rccl_free(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def split(self, lo, hi):
        """Collective over all ranks: the communicator of the rank range [lo, hi)."""
        raise_for(lib().kvx_rccl_split(self._h, int(lo), int(hi)), "ncclCommSplit failed")

    def barrier(self):
        raise_for(lib().kvx_rccl_barrier(self._h), "barrier failed")

    def _allreduce(self, values, op):
        a = np.ascontiguousarray(np.atleast_1d(np.asarray(values, dtype=np.float64))).copy()
        raise_for(lib().kvx_rccl_allreduce_host(self._h, a.ctypes.data_as(f64p), int(a.size), op), "all-reduce failed")
        return a

    def max(self, value):
        return float(self._allreduce([value], 1)[0])

    def min(self, value):
        return float(self._allreduce([value], 2)[0])

    def sum(self, value):
        return float(self._allreduce([value], 0)[0])

    def all_gather(self, values):
        a = np.ascontiguousarray(np.atleast_1d(np.asarray(values, dtype=np.float64)))
        out = np.zeros(a.size * self.world)
        raise_for(lib().kvx_rccl_allgather_host(self._h, a.ctypes.data_as(f64p), int(a.size), out.ctypes.data_as(f64p)), "all-gather failed")
        return out.reshape(self.world, a.size)

    def stats(self):
        s = np.zeros(2, dtype=np.int64)
        raise_for(lib().kvx_rccl_stats(self._h, s.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))))
        return int(s[0]), int(s[1])
