"""The judged entry point itself: `python bench.py --gpus N` starts its own ranks (no torchrun environment needed), N > 1 shards
ONE system over them by default (strong scaling), and the line carries the objects the contract names."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_gpus_flag_starts_the_ranks_and_reports_a_failing_rank():
    """CPU box: both children stop at 'needs a HIP device' -- the parent (which never imports torch) must relay that as a
    non-zero exit instead of printing a one-rank line."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--quick"], env=_clean_env(HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES=""),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stderr.count("needs a HIP device") == 2 and "a rank exited with code" in r.stderr
    assert r.stdout.strip() == ""


def test_parent_of_the_ranks_does_not_import_torch():
    """The process that starts the ranks must not have initialised the GPU: it must not even import torch."""
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2', '--quick']\n"
            "import subprocess\n"
            "class P:\n"
            "    def __init__(self, *a, **k): self.env = k['env']\n"
            "    def poll(self): return 0\n"
            "    def terminate(self): pass\n"
            "started = []\n"
            "def popen(*a, **k):\n"
            "    assert 'torch' not in sys.modules\n"
            "    p = P(*a, **k); started.append(p); return p\n"
            "subprocess.Popen = popen\n"
            "try:\n"
            "    runpy.run_path(%r, run_name='__main__')\n"
            "except SystemExit as e:\n"
            "    assert e.code == 0, e.code\n"
            "assert 'torch' not in sys.modules\n"
            "assert [p.env['RANK'] for p in started] == ['0', '1'] and all(p.env['WORLD_SIZE'] == '2' for p in started)\n"
            "assert all(p.env['MASTER_ADDR'] == '127.0.0.1' for p in started) and len({p.env['MASTER_PORT'] for p in started}) == 1\n"
            "print('ok')\n" % BENCH)
    r = subprocess.run([sys.executable, "-c", code], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.stdout, r.stderr)


@pytest.mark.gpu
def test_two_rank_bench_line_shards_one_system():
    """`bench.py --gpus 2` on the one GPU of the box (gloo instead of RCCL -- RCCL refuses two ranks on one device): the line says
    n_gpus 2, strong scaling, names the backend and the per-rank device bytes, and the sharded result meets the residual bar."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--grid", "300", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extra"],
                       env=_clean_env(KVX_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["rel_residual"] < 1e-10
    assert d["ranks"]["backend"] == "gloo" and d["ranks"]["world_size"] == 2 and len(d["ranks"]["factor_bytes_by_rank"]) == 2
    assert d["sharding"]["collectives_per_step_rank0"] >= 2
    assert 0 < d["sharding"]["panel_doubles_rank0"] < d["sharding"]["panel_doubles_total"]       # per-rank layout, not the whole factor
    assert d["roofline"]["frac"] > 0 and d["value"] > 0


@pytest.mark.gpu
def test_one_gpu_line_carries_extra_one_shot_and_baselines():
    """The default N = 1 line at a reduced grid: roofline, cpu_baseline with the cores used / available, ranks; the `extra`
    and `one_shot` legs are tied to the full-size headline and are exercised by the driver's own bench run."""
    r = subprocess.run([sys.executable, BENCH, "--grid", "200", "--steps", "3", "--warmup", "1", "--no-ipm", "--no-splu"],
                       env=_clean_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["ranks"]["world_size"] == 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] <= cb["cores_available"] and cb["by_threads"]
    assert d["roofline"]["bound"] in ("hbm", "mfma") and "timing_mode" in d["roofline"]
