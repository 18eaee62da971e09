// lab: where do the cycles of the 64x64 potrf+inverse sweep go?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
constexpr int NB = 64;
__device__ __forceinline__ void sqrt_rsqrt(double d, double &root, double &inv){ double r=__builtin_amdgcn_rsq(d); const double hd=0.5*d; r=r*__builtin_fma(-hd*r,r,1.5); r=r*__builtin_fma(-hd*r,r,1.5); double x=d*r; x=__builtin_fma(0.5*r,__builtin_fma(-x,x,d),x); root=x; inv=r; }

template <int JS, int FLAGS>   // FLAGS bit0: barrier, bit1: chain, bit2: factor updates, bit3: inverse role work, bit4: publish
__device__ __forceinline__ void substep(double (&a)[16], double *cb2, double *yr2, int g, int i, int q, bool inv_role)
{
    const int j = 4 * g + JS;
    double *cb = cb2 + (j & 1) * NB;
    double *yr = yr2 + (j & 1) * NB;
    if (FLAGS & 16) {
    if (!inv_role) { if (q == JS) cb[i] = a[0]; }
    else if ((FLAGS & 8) && i == j) {
#pragma unroll
        for (int t = 0; t < 16; t++) if (t <= g) yr[q + 4 * (g - t)] = a[t];
    } }
    if (FLAGS & 1) __syncthreads();
    double d = cb[j];
    if (!(d > 0.0)) d = 1.0;
    double ljj = 1.0, inv = 1.0;
    if (FLAGS & 2) sqrt_rsqrt(d, ljj, inv);
    const double ci = cb[i];
    const double w = (i > j) ? ci * (inv * inv) : 0.0;
    if (!inv_role) {
        if (FLAGS & 4) {
        const double w0 = (q > JS) ? w : 0.0;
        double lc[16];
#pragma unroll
        for (int t = 0; t < 16; t++) lc[t] = cb[min(q + 4 * (g + t), NB - 1)];
        a[0] = __builtin_fma(-w0, lc[0], a[0]);
#pragma unroll
        for (int t = 1; t < 16; t++) a[t] = __builtin_fma(-w, lc[t], a[t]);
        }
        if (q == JS) a[0] = (i == j) ? ljj : (i > j ? ci * inv : 0.0);
    } else if (FLAGS & 8) {
        const bool piv = (i == j);
        double yv[16];
#pragma unroll
        for (int t = 0; t < 16; t++) yv[t] = yr[max(q + 4 * (g - t), 0)];
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const bool act = (t <= g) && (t > 0 || q <= JS);
            const double wm = act ? w : 0.0;
            const double upd = __builtin_fma(-wm, yv[t], a[t]);
            const double scl = yv[t] * inv;
            a[t] = (piv && act) ? scl : upd;
        }
    }
}

template <int FLAGS, int NT>
__global__ __launch_bounds__(NT) void k_potrf(double *P, double *Yg, int m, unsigned long long *cyc)
{
    __shared__ double cb2[2 * NB];
    __shared__ double yr2[2 * NB];
    const int tid = threadIdx.x, i = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
    const bool inv_role = __builtin_amdgcn_readfirstlane(tid >> 8) != 0;
    double a[16];
#pragma unroll
    for (int t = 0; t < 16; t++) { const int c = q + 4 * t; const double v = P[i + (c <= i ? c : 0) * m]; a[t] = inv_role ? 0.0 : (c <= i ? v : 0.0); }
    if (tid < 2*NB) { cb2[tid] = 1.0; yr2[tid] = 0.0; }
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int g = 0; g < 16; g++) {
        if (inv_role) {
#pragma unroll
            for (int t = 15; t > 0; t--) a[t] = a[t - 1];
            a[0] = (q + 4 * g == i) ? 1.0 : 0.0;
        }
        substep<0, FLAGS>(a, cb2, yr2, g, i, q, inv_role);
        substep<1, FLAGS>(a, cb2, yr2, g, i, q, inv_role);
        substep<2, FLAGS>(a, cb2, yr2, g, i, q, inv_role);
        substep<3, FLAGS>(a, cb2, yr2, g, i, q, inv_role);
        if (!inv_role) {
            const int c = q + 4 * g;
            if (c <= i) P[i + c * m] = a[0];
#pragma unroll
            for (int t = 0; t < 15; t++) a[t] = a[t + 1];
            a[15] = 0.0;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (inv_role) {
#pragma unroll
        for (int t = 0; t < 16; t++) { const int c = q + 4 * (15 - t); if (c >= 0 && c <= i) Yg[i + c * NB] = a[t]; }
    }
    if (tid == 0) cyc[0] = t1 - t0;
}

int main(){
  const int m = 64; std::vector<double> A(m*m);
  for (int j=0;j<m;j++) for (int i=0;i<m;i++) A[i+j*m] = (i==j) ? 70.0 : 1.0/(1+abs(i-j));
  double *dP, *dY; unsigned long long *dc; hipMalloc(&dP, m*m*8); hipMalloc(&dY, 64*64*8); hipMalloc(&dc, 8);
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run=[&](const char* name, auto kern, int nt){
     float best=1e9; unsigned long long cy=0;
     for (int rep=0; rep<5; rep++){ hipMemcpy(dP, A.data(), m*m*8, hipMemcpyHostToDevice); hipEventRecord(e0); hipLaunchKernelGGL(kern, 1, nt, 0, 0, dP, dY, m, dc); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); if (ms<best) best=ms; hipMemcpy(&cy, dc, 8, hipMemcpyDeviceToHost);} 
     printf("%-44s %7.1f us (event)  sweep %8llu cycles = %.0f per column\n", name, best*1e3, cy, cy/64.0);
  };
  run("full 512 (barrier+chain+fupd+inv+publish)", k_potrf<31,512>, 512);
  run("no inverse work 512", k_potrf<23,512>, 512);
  run("factor only 256 threads", k_potrf<23,256>, 256);
  run("factor only 256, no chain", k_potrf<21,256>, 256);
  run("factor only 256, no updates", k_potrf<19,256>, 256);
  run("factor only 256, no barrier", k_potrf<22,256>, 256);
  run("factor only 256, barrier only", k_potrf<1,256>, 256);
  run("factor only 256, barrier+publish", k_potrf<17,256>, 256);
  run("factor only 256, barrier+publish+chain", k_potrf<19,256>, 256);
  run("full 512 no chain", k_potrf<29,512>, 512);
  run("full 512 no barrier", k_potrf<30,512>, 512);
  return 0;
}
