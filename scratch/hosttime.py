import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from kvxopt_amd import workloads
from kvxopt_amd.chol import Factor
n, cp, ri, vx = workloads.laplacian_2d(1000)
F = Factor(n, cp, ri)
dev = torch.device('cuda', 0)
v = torch.from_numpy(vx).to(dev); b = torch.ones(n, dtype=torch.float64, device=dev)
F.factorize_dev(v.data_ptr()); F.solve_dev(b.data_ptr(), 0, 1, n)
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); F.factorize_dev(v.data_ptr(), sync=False); t1 = time.perf_counter(); F.status(); t2 = time.perf_counter()
    F.solve_dev(b.data_ptr(), 0, 1, n); t3 = time.perf_counter()
    print('factor enqueue %.2f ms, wait %.2f ms; solve (enqueue+wait) %.2f ms; gpu timing' % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3), F.timing())
