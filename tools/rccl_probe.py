"""Can two RCCL ranks share the one GPU of a dev box?  (decides whether the stream-ordered branch of dist.py can be rehearsed)"""
import os, sys, subprocess, socket
if "RANK" not in os.environ:
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ps = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        ps.append(subprocess.Popen([sys.executable, __file__], env=env))
    rc = 0
    for p in ps:
        try:
            rc |= p.wait(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill(); rc |= 99
    print("probe rc", rc)
    sys.exit(0)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    t = torch.ones(1024, device="cuda", dtype=torch.float64) * (dist.get_rank() + 1)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("rank", dist.get_rank(), "allreduce ->", float(t[0]))
    dist.broadcast(t, src=1)
    torch.cuda.synchronize()
    print("rank", dist.get_rank(), "bcast ->", float(t[0]))
    dist.destroy_process_group()
except Exception as e:
    print("rank", os.environ["RANK"], "FAILED", repr(e)[:400])
    sys.exit(3)
