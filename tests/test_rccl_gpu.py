"""RCCL bound directly (csrc/rccl_comm.cpp, kvxopt_amd/rccl.py): the transport a multi-GPU node uses.  A box of this build has
one GPU and RCCL refuses two ranks on one device, so the GPU rehearsal is a world of one rank through the real library; the
rendezvous that hands rank 0's id to the others is tested with real processes on the CPU."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_rank_process_on_rccl_direct_has_no_torch_and_matches_the_single_gpu_factor():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29591")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(HERE, "rccl_child.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert out["torch_loaded"] is False                       # the rank never imported torch: it runs on the system HIP runtime
    assert out["backend"] == "rccl-direct" and out["world"] == 1 and out["version"] >= 21800
    assert out["max"] == 3.5 and out["gather"] == [[1.0, 2.0]]
    assert out["buf"] == list(np.arange(8.0))                 # broadcast / SUM / MIN over one rank leave the buffer as it was
    assert out["solve_equal"] and out["residual"] < 1e-12
    assert out["minor_dist"] == out["minor_single"]           # ArithmeticError(failing column) through the MIN all-reduce
    assert out["direct_calls"] == 3                           # counted inside rccl_comm.cpp: the callback is the C function
    assert out["collectives"] == 0                            # (a world of one rank exchanges nothing inside the factor)


@pytest.mark.gpu
def test_bench_quick_line_with_one_rank_on_rccl_direct():
    """bench.py as a rank process of a one-rank job on the RCCL-direct transport (what every rank of `--gpus 8` runs)."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29592")
    r = subprocess.run([sys.executable, os.path.join(HERE, "..", "bench.py"), "--quick", "--grid", "200", "--steps", "3", "--warmup", "2"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["rel_residual"] < 1e-10


def _make_id():
    return bytes(range(128))


@pytest.mark.parametrize("how", ["file", "tcp"])
def test_id_rendezvous_between_processes_without_a_gpu(how, tmp_path):
    """Rank 0's 128-byte id reaches three other ranks, whichever of them starts first (no GPU, no RCCL: the transport only)."""
    from kvxopt_amd import rccl
    got = {}

    def rank(r):
        if how == "file":
            os.environ["KVX_RCCL_ID_FILE"] = str(tmp_path / "id")
            got[r] = rccl._exchange_id_file(r, 4, 29650, _make_id, timeout=30.0)[0]
        else:
            got[r] = rccl._exchange_id(r, 4, "127.0.0.1", 29651, _make_id, timeout=30.0)
    try:
        ts = [threading.Thread(target=rank, args=(r,)) for r in (2, 1, 3, 0)]      # rank 0 last: the others have to wait for it
        for t in ts:
            t.start()
        for t in ts:
            t.join(60)
    finally:
        os.environ.pop("KVX_RCCL_ID_FILE", None)
    assert all(got.get(r) == _make_id() for r in range(4)), got.keys()
