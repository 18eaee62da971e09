import os, sys, faulthandler, socket, subprocess, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if "RANK" not in os.environ:
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ps = [subprocess.Popen([sys.executable, "-u", __file__], env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))) for r in range(2)]
    for p in ps:
        try: p.wait(timeout=100)
        except subprocess.TimeoutExpired: p.kill()
    sys.exit(0)
faulthandler.dump_traceback_later(40, exit=True)
import numpy as np, torch, torch.distributed as dist
from kvxopt_amd import workloads
from kvxopt_amd.dist import DistFactor
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=2)
dev = torch.device("cuda", 0)
for name, (n, cp, ri, vx) in [("lap2d120", workloads.laplacian_2d(120)), ("lap2d33x71", workloads.laplacian_2d(33, 71)), ("lap3d24", workloads.laplacian_3d(24))]:
    print(rank, name, "setup", flush=True)
    DF = DistFactor(n, cp, ri, ob=64, min_m=96)
    v_d = torch.from_numpy(vx).to(dev)
    print(rank, name, "factor", flush=True)
    DF.factorize(v_d)
    print(rank, name, "factor done", flush=True)
    B = np.random.default_rng(7).standard_normal((n, 2))
    b_d = torch.from_numpy(B.reshape(-1, order="F").copy()).to(dev)
    DF.solve(b_d, 2)
    X = b_d.cpu().numpy().reshape(n, 2, order="F")
    R = workloads.sym_matvec(n, cp, ri, vx, X) - B
    print(rank, name, "res", np.linalg.norm(R) / np.linalg.norm(B), flush=True)
dist.barrier()
dist.destroy_process_group()
