// micro-benchmarks: instruction latencies on MI355X for the latency-bound front kernels
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

__device__ inline double rl(double v, int lane){ long long b=__double_as_longlong(v); int lo=(int)(b&0xffffffffll), hi=(int)(b>>32); lo=__builtin_amdgcn_readlane(lo,lane); hi=__builtin_amdgcn_readlane(hi,lane); return __longlong_as_double(((long long)hi<<32)|(unsigned)lo);}

__global__ void k_fma_chain(double* out, int n, unsigned long long* cyc){
  double a = out[threadIdx.x], b = 1.0000001, c = 1e-9;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i=0;i<n;i++){ a = __builtin_fma(a,b,c); }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x]=a; if(threadIdx.x==0) cyc[0]=t1-t0;
}
__global__ void k_fma_indep(double* out, int n, unsigned long long* cyc){
  double a[8]; for(int q=0;q<8;q++) a[q]=out[threadIdx.x]+q; double b=1.0000001,c=1e-9;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i=0;i<n;i++){
#pragma unroll
    for(int q=0;q<8;q++) a[q]=__builtin_fma(a[q],b,c); }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s=0; for(int q=0;q<8;q++) s+=a[q]; out[threadIdx.x]=s; if(threadIdx.x==0) cyc[0]=t1-t0;
}
__global__ void k_readlane_chain(double* out, int n, unsigned long long* cyc){
  double a = out[threadIdx.x];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i=0;i<n;i++){ double s = rl(a, i&63); a = __builtin_fma(a, 0.999, s*1e-9); }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x]=a; if(threadIdx.x==0) cyc[0]=t1-t0;
}
__global__ void k_readlane_indep(double* out, int n, unsigned long long* cyc){
  double a[8]; for(int q=0;q<8;q++) a[q]=out[threadIdx.x]+q; double v = out[threadIdx.x]*0.5;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i=0;i<n;i++){
#pragma unroll
    for(int q=0;q<8;q++){ double s = rl(v, (i+q)&63); a[q]=__builtin_fma(-v,s,a[q]); } }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s=0; for(int q=0;q<8;q++) s+=a[q]; out[threadIdx.x]=s; if(threadIdx.x==0) cyc[0]=t1-t0;
}
__global__ void k_lds_bcast(double* out, int n, unsigned long long* cyc){
  __shared__ double L[4096]; for(int i=threadIdx.x;i<4096;i+=blockDim.x) L[i]=i*1e-6; __syncthreads();
  double a[8]; for(int q=0;q<8;q++) a[q]=out[threadIdx.x]+q; double v = out[threadIdx.x]*0.5;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i=0;i<n;i++){
#pragma unroll
    for(int q=0;q<8;q++){ a[q]=__builtin_fma(-v,L[(i*8+q)&4095],a[q]); } }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s=0; for(int q=0;q<8;q++) s+=a[q]; out[threadIdx.x]=s; if(threadIdx.x==0) cyc[0]=t1-t0;
}
__global__ void k_rsq_chain(double* out, int n, unsigned long long* cyc){
  double a = out[threadIdx.x]+2.0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i=0;i<n;i++){ double r=__builtin_amdgcn_rsq(a); const double hd=0.5*a; r=r*__builtin_fma(-hd*r,r,1.5); r=r*__builtin_fma(-hd*r,r,1.5); a = a*r + 1.5; }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x]=a; if(threadIdx.x==0) cyc[0]=t1-t0;
}
__global__ void k_barrier(double* out, int n, unsigned long long* cyc){
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i=0;i<n;i++){ __syncthreads(); }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if(threadIdx.x==0) cyc[0]=t1-t0;
}
__global__ void k_gload_chain(const int* idx, int n, unsigned long long* cyc, int* out){
  int p = threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i=0;i<n;i++){ p = idx[p]; }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x]=p; if(threadIdx.x==0) cyc[0]=t1-t0;
}
__global__ void k_mfma_chain(double* out, int n, unsigned long long* cyc){
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 acc={0,0,0,0}; double a=out[threadIdx.x], b=a*0.5;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i=0;i<n;i++){ acc=__builtin_amdgcn_mfma_f64_16x16x4f64(a,b,acc,0,0,0); }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x]=acc[0]+acc[1]; if(threadIdx.x==0) cyc[0]=t1-t0;
}
__global__ void k_empty(){}

int main(){
  double* d; unsigned long long* c; int* idx; int* io;
  CK(hipMalloc(&d, 1024*8)); CK(hipMalloc(&c, 64)); CK(hipMemset(d,0,1024*8));
  const int NI=1<<22; std::vector<int> h(NI); for(int i=0;i<NI;i++) h[i]=(int)(((long long)i*1664525LL+1013904223LL)%NI);
  CK(hipMalloc(&idx, NI*4)); CK(hipMalloc(&io, 4096)); CK(hipMemcpy(idx,h.data(),NI*4,hipMemcpyHostToDevice));
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run=[&](const char* name, auto launch, int n, int per){
    for(int rep=0;rep<3;rep++){
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms,e0,e1); unsigned long long cy; hipMemcpy(&cy,c,8,hipMemcpyDeviceToHost);
      if(rep==2) printf("%-18s n=%d  %.1f us  memtime ticks/op %.2f  ns/op %.2f\n", name, n, ms*1e3, (double)cy/(n*per), ms*1e6/(n*(double)per));
    }};
  int n=20000;
  run("empty", [&]{ hipLaunchKernelGGL(k_empty,1,64,0,0); }, 1, 1);
  run("fma_chain", [&]{ hipLaunchKernelGGL(k_fma_chain,1,64,0,0,d,n,c); }, n,1);
  run("fma_indep8", [&]{ hipLaunchKernelGGL(k_fma_indep,1,64,0,0,d,n,c); }, n,8);
  run("readlane_chain", [&]{ hipLaunchKernelGGL(k_readlane_chain,1,64,0,0,d,n,c); }, n,1);
  run("readlane_indep8", [&]{ hipLaunchKernelGGL(k_readlane_indep,1,64,0,0,d,n,c); }, n,8);
  run("lds_bcast_fma8", [&]{ hipLaunchKernelGGL(k_lds_bcast,1,64,0,0,d,n,c); }, n,8);
  run("lds_bcast_fma8x4w", [&]{ hipLaunchKernelGGL(k_lds_bcast,1,256,0,0,d,n,c); }, n,8);
  run("rsq_newton_chain", [&]{ hipLaunchKernelGGL(k_rsq_chain,1,64,0,0,d,n,c); }, n,1);
  run("barrier_2w", [&]{ hipLaunchKernelGGL(k_barrier,1,128,0,0,d,n,c); }, n,1);
  run("barrier_4w", [&]{ hipLaunchKernelGGL(k_barrier,1,256,0,0,d,n,c); }, n,1);
  run("barrier_8w", [&]{ hipLaunchKernelGGL(k_barrier,1,512,0,0,d,n,c); }, n,1);
  run("gload_chain", [&]{ hipLaunchKernelGGL(k_gload_chain,1,64,0,0,idx,2000,c,io); }, 2000,1);
  run("mfma_f64_chain", [&]{ hipLaunchKernelGGL(k_mfma_chain,1,64,0,0,d,n,c); }, n,1);
  // many-wave variants (fill the chip) to see clock effects
  run("fma_chain_full", [&]{ hipLaunchKernelGGL(k_fma_chain,1024,256,0,0,d,n,c); }, n,1);
  return 0;
}
