import torch, time
for n,k in [(8192,256),(8192,8192),(10240,256)]:
    a=torch.randn(n,k,dtype=torch.float64,device='cuda'); c=torch.randn(n,n,dtype=torch.float64,device='cuda')
    for _ in range(3): c.addmm_(a,a.t(),alpha=-1.0)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): c.addmm_(a,a.t(),alpha=-1.0)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
    print("rocBLAS dgemm C(%d x %d) -= A(%d x %d) A^T: %.3f ms, %.1f TF/s"%(n,n,n,k,dt*1e3,2.0*n*n*k/dt/1e12))
