s = open('/root/repo/scratch/potrf_lab2.hip').read()
a = s.index("// cbp: [2][128] (pivot column, rows 64..127 zero)")
b = s.index("int main(){")
new = r'''// LDS: cbp[3][128]: unscaled pivot column k at [0..63], 1/l_kk at [64], zeros above (triple-buffered: the
// inverse role reads column k-1 while the factor role already publishes column k+1);
// yrp[2][128]: row of Y at [64..127], zeros below (windows that reach "column < 0" read zeros).
// Barrier B_k: the factor role has published column k; the inverse role has published row k-1 of Y and
// then runs step k-1 (one step behind, so it needs no rsqrt chain: 1/l is read from LDS).
template <int JS>
__device__ __forceinline__ void fstep(double (&a)[16], double *cbp, int &kb, int g, int i, int q, int *status)
{
    const int j = 4 * g + JS;
    double *cb = cbp + kb * 128;
    kb = (kb == 2) ? 0 : kb + 1;
    if (q == JS) cb[i] = a[0];
    __syncthreads();
    double d = cb[j];
    if (!(d > 0.0)) { if (q == JS && i == 0) atomicMin(status, j); d = 1.0; }
    double ljj, inv;
    sqrt_rsqrt(d, ljj, inv);
    if (q == JS && i == 0) cb[64] = inv;
    const double ci = cb[i];
    const double w = (i > j) ? ci * (inv * inv) : 0.0;
    const double *src = cb + q + 4 * g;
    double lc[16];
#pragma unroll
    for (int t = 0; t < 16; t++) lc[t] = src[4 * t];
    const double w0 = (q > JS) ? w : 0.0;
    a[0] = __builtin_fma(-w0, lc[0], a[0]);
#pragma unroll
    for (int t = 1; t < 16; t++) a[t] = __builtin_fma(-w, lc[t], a[t]);
    if (q == JS) a[0] = (i == j) ? ljj : (i > j ? ci * inv : 0.0);
}

// inverse role, column j = 4g + JS: publish row j of Y, barrier B_{j+1}, update with column j
template <int JS>
__device__ __forceinline__ void istep(double (&a)[16], double &myinv, double *cbp, double *yrp, int &kb, int g, int i, int q)
{
    const int j = 4 * g + JS;
    double *yr = yrp + (j & 1) * 128 + 64 + q + 4 * g;
    if (i == j) {
#pragma unroll
        for (int t = 0; t < 16; t++) yr[-4 * t] = a[t];
    }
    __syncthreads();
    const double *cb = cbp + kb * 128;
    kb = (kb == 2) ? 0 : kb + 1;
    const double inv = cb[64];
    const double ci = cb[i];
    const double w = (i > j) ? ci * (inv * inv) : 0.0;
    myinv = (i == j) ? inv : myinv;
    double yv[16];
#pragma unroll
    for (int t = 0; t < 16; t++) yv[t] = yr[-4 * t];
    const double w0 = (q <= JS) ? w : 0.0;
    a[0] = __builtin_fma(-w0, yv[0], a[0]);
#pragma unroll
    for (int t = 1; t < 16; t++) a[t] = __builtin_fma(-w, yv[t], a[t]);
}

__global__ __launch_bounds__(512) void k_potrf(double *P, double *Yg, int m, unsigned long long *cyc, int *status)
{
    __shared__ double cbp[3 * 128];
    __shared__ double yrp[2 * 128];
    const int tid = threadIdx.x, i = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
    const bool inv_role = __builtin_amdgcn_readfirstlane(tid >> 8) != 0;
    double a[16];
#pragma unroll
    for (int t = 0; t < 16; t++) { const int c = q + 4 * t; const double v = P[i + (c <= i ? c : 0) * m]; a[t] = inv_role ? 0.0 : (c <= i ? v : 0.0); }
    if (tid < 384) cbp[tid] = 0.0;
    if (tid < 256) yrp[tid] = 0.0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const int ngrp = 16;
    int kb = 0;
    if (!inv_role) {
        for (int g = 0; g < ngrp; g++) {
            fstep<0>(a, cbp, kb, g, i, q, status);
            fstep<1>(a, cbp, kb, g, i, q, status);
            fstep<2>(a, cbp, kb, g, i, q, status);
            fstep<3>(a, cbp, kb, g, i, q, status);
            const int c = q + 4 * g;
            if (c <= i) P[i + c * m] = a[0];
#pragma unroll
            for (int t = 0; t < 15; t++) a[t] = a[t + 1];
            a[15] = 0.0;
        }
        __syncthreads();                       // B_64: lets the inverse role finish column 63
    } else {
        double myinv = 1.0;
        __syncthreads();                       // B_0
        for (int g = 0; g < ngrp; g++) {
#pragma unroll
            for (int t = 15; t > 0; t--) a[t] = a[t - 1];
            a[0] = (q + 4 * g == i) ? 1.0 : 0.0;
            istep<0>(a, myinv, cbp, yrp, kb, g, i, q);
            istep<1>(a, myinv, cbp, yrp, kb, g, i, q);
            istep<2>(a, myinv, cbp, yrp, kb, g, i, q);
            istep<3>(a, myinv, cbp, yrp, kb, g, i, q);
        }
#pragma unroll
        for (int t = 0; t < 16; t++) { const int c = q + 4 * (15 - t); if (c >= 0 && c <= i) Yg[i + c * NB] = a[t] * myinv; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[0] = t1 - t0;
}

'''
s = s[:a] + new + s[b:]
open('/root/repo/scratch/potrf_lab4.hip', 'w').write(s)
