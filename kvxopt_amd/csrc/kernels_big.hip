// Big fronts (order m > 128 or k > 64): blocked right-looking factorisation in HBM/L2 with
// 64-column panel steps, and multi-workgroup triangular solves.
//
// Per panel step jb (three dependent launches, every big front of the level batched):
//   k_potrf_blk   : Cholesky of the 64x64 diagonal block AND its inverse.  512 threads: waves 0-3
//                   factor, waves 4-7 build the inverse by the same column sweep; thread (row i,
//                   column phase q) keeps 16 columns in registers; the pivot column and the pivot row
//                   of the inverse are broadcast through LDS (measured: LDS broadcast + fma ~10-13
//                   cycles, v_readlane pair + fma ~40); one barrier per column;
//   k_trsm_blk    : X := A * Linv' for the rows below, FP64 MFMA (v_mfma_f64_16x16x4_f64);
//   k_syrk_trailing: C -= X X' on 64x64 tiles, FP64 MFMA.
// The inverses of the diagonal blocks stay resident: the solves use them as 64x64 mat-vecs, so a
// big front's triangular solve has no 64-long dependent chain and is spread over workgroups.
//
// Reference role: cholmod_l_factorize / cholmod_l_solve (src/C/cholmod.c:362, 483).
#include "device.hpp"

#include <algorithm>
#include <utility>

namespace kvx {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NB = KVX_NB;

// ------------------------------------------------------------------------------------------
// 64x64 diagonal block: Cholesky factor AND its inverse in one 512-thread workgroup.
// Thread (row i = tid & 63, column phase q) keeps 16 columns in registers.  Factor role (waves
// 0-3): a[t] = D[i][q + 4(g + t)], the window slides down as column groups finish.  Inverse role
// (waves 4-7): a[t] = Y[i][q + 4(g - t)], Y = D^{-1} by forward substitution on the identity with
// the same column sweep, run ONE STEP BEHIND the factor role so that it needs no rsqrt chain.
// LDS: cbp[3][128] = unscaled pivot column k at [0..63], 1/l_kk at [64], zeros above (triple
// buffered: the inverse role reads column k-1 while column k+1 is being published);
// yrp[2][128] = row of Y at [64..127], zeros below (windows reaching "column < 0" read zeros).
// Everything inside the 16-register loops is branch-free with immediate LDS offsets: these kernels
// are bound by instruction issue of single waves (lab measurements in DESIGN.md).
template <int JS>
__device__ __forceinline__ void potrf_fstep(double (&a)[16], double *cbp, int &kb, int g, int i, int q,
                                            int *status, int col0)
{
    const int j = 4 * g + JS;
    double *cb = cbp + kb * 128;
    kb = (kb == 2) ? 0 : kb + 1;
    if (q == JS) cb[i] = a[0];
    __syncthreads();
    double d = cb[j];
    if (!(d > 0.0)) {
        if (q == JS && i == 0) atomicMin(status, col0 + j);
        d = 1.0;
    }
    double ljj, inv;
    kvx_sqrt_rsqrt(d, ljj, inv);
    if (q == JS && i == 0) cb[64] = inv;
    const double ci = cb[i];
    const double w = (i > j) ? ci * (inv * inv) : 0.0;
    const double *src = cb + q + 4 * g;
    double lc[16];
#pragma unroll
    for (int t = 0; t < 16; t++) lc[t] = src[4 * t];
    const double w0 = (q > JS) ? w : 0.0;              // window 0: only columns right of the pivot
    a[0] = __builtin_fma(-w0, lc[0], a[0]);
#pragma unroll
    for (int t = 1; t < 16; t++) a[t] = __builtin_fma(-w, lc[t], a[t]);
    if (q == JS) a[0] = (i == j) ? ljj : (i > j ? ci * inv : 0.0);
}

// inverse role, column j = 4g + JS: publish row j of Y, barrier, update with column j of L
template <int JS>
__device__ __forceinline__ void potrf_istep(double (&a)[16], double &myinv, double *cbp, double *yrp, int &kb,
                                            int g, int i, int q)
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
    myinv = (i == j) ? inv : myinv;                    // row i of Y is scaled by 1/l_ii once, at the end
    double yv[16];
#pragma unroll
    for (int t = 0; t < 16; t++) yv[t] = yr[-4 * t];
    const double w0 = (q <= JS) ? w : 0.0;
    a[0] = __builtin_fma(-w0, yv[0], a[0]);
#pragma unroll
    for (int t = 1; t < 16; t++) a[t] = __builtin_fma(-w, yv[t], a[t]);
}

__global__ __launch_bounds__(512) void k_potrf_blk(DevSym ds, const int32_t *__restrict__ list, int jb,
                                                   double *__restrict__ Lx, double *__restrict__ Linv, int *status)
{
    __shared__ double cbp[3 * 128];
    __shared__ double yrp[2 * 128];
    const FrontDesc fd = ds.fd[list[blockIdx.x]];
    const int k = fd.k, m = fd.m;
    if (jb >= k) return;
    const int nbk = min(NB, k - jb);
    const int tid = threadIdx.x, i = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
    const bool inv_role = __builtin_amdgcn_readfirstlane(tid >> 8) != 0;
    double *P = Lx + fd.px;
    double *Yg = Linv + fd.linv + (int64_t)(jb / NB) * NB * NB;
    const int col0 = fd.first + jb;
    double a[16];
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const int c = q + 4 * t;
        const double v = kvx_ld0(P, (jb + i) + (int64_t)(jb + c) * m, !inv_role && i < nbk && c <= i);
        a[t] = inv_role ? 0.0 : ((i < nbk && c <= i) ? v : (c == i ? 1.0 : 0.0));   // identity padding
    }
    if (tid < 384) cbp[tid] = 0.0;
    if (tid < 256) yrp[tid] = 0.0;
    __syncthreads();
    const int ngrp = (nbk + 3) >> 2;                       // the padding beyond nbk needs no sweep
    int kb = 0;
    if (!inv_role) {
        for (int g = 0; g < ngrp; g++) {
            potrf_fstep<0>(a, cbp, kb, g, i, q, status, col0);
            potrf_fstep<1>(a, cbp, kb, g, i, q, status, col0);
            potrf_fstep<2>(a, cbp, kb, g, i, q, status, col0);
            potrf_fstep<3>(a, cbp, kb, g, i, q, status, col0);
            const int c = q + 4 * g;                       // this thread's finished column
            if (i < nbk && c <= i) P[(jb + i) + (int64_t)(jb + c) * m] = a[0];
#pragma unroll
            for (int t = 0; t < 15; t++) a[t] = a[t + 1];
            a[15] = 0.0;
        }
        __syncthreads();                                   // lets the inverse role finish the last column
    } else {
        double myinv = 1.0;
        __syncthreads();                                   // pairs with the factor role's first barrier
        for (int g = 0; g < ngrp; g++) {
#pragma unroll
            for (int t = 15; t > 0; t--) a[t] = a[t - 1];
            a[0] = (q + 4 * g == i) ? 1.0 : 0.0;           // open the window on column group g: Y starts as I
            potrf_istep<0>(a, myinv, cbp, yrp, kb, g, i, q);
            potrf_istep<1>(a, myinv, cbp, yrp, kb, g, i, q);
            potrf_istep<2>(a, myinv, cbp, yrp, kb, g, i, q);
            potrf_istep<3>(a, myinv, cbp, yrp, kb, g, i, q);
        }
        if (i < nbk) {
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const int c = q + 4 * (ngrp - 1 - t);
                if (c >= 0 && c <= i) Yg[i + c * NB] = a[t] * myinv;
            }
        }
    }
}

void launch_potrf_blk(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int jb,
                      double *Lx, double *Linv, int *status)
{
    if (count <= 0) return;
    hipLaunchKernelGGL(k_potrf_blk, dim3((unsigned)count), dim3(512), 0, st, ds, list, jb, Lx, Linv, status);
}

// ------------------------------------------------------------------------------------------
// X := A * Linv' for a 64-row block below the diagonal block.  Output roles swapped as in the
// trailing update (D[i][j]: i <-> panel column, j <-> row) so that stores run along rows.  All
// operand loads of the 16 k-steps are branch-free; Linv is lower triangular, so k-groups above a
// column tile are skipped.
__global__ __launch_bounds__(256) void k_trsm_blk(DevSym ds, const int32_t *__restrict__ list, int jb,
                                                  double *__restrict__ Lx, const double *__restrict__ Linv)
{
    const FrontDesc fd = ds.fd[list[blockIdx.y]];
    const int k = fd.k, m = fd.m;
    if (jb >= k) return;
    const int nbk = min(NB, k - jb);
    const int r0 = jb + nbk + blockIdx.x * 64;
    if (r0 >= m) return;
    double *P = Lx + fd.px;
    const double *Y = Linv + fd.linv + (int64_t)(jb / NB) * NB * NB;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, lr = l & 15, lk = l >> 4;
    const int rr = r0 + 16 * w + lr;
    const bool rin = rr < m;
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kg = 0; kg < NB; kg += 16) {
        if (kg < nbk) {                             // wave-uniform
            double bq[4], aq[4][4];
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                const int kc = kg + 4 * qq + lk;
                const bool kin = kc < nbk;
                bq[qq] = kvx_ld0(P, rr + (int64_t)(jb + kc) * m, kin && rin);
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int cc = 16 * t + lr;
                    aq[qq][t] = (16 * t + 15 >= kg) ? kvx_ld0(Y, cc + kc * NB, kin && cc < nbk && kc <= cc) : 0.0;
                }
            }
#pragma unroll
            for (int qq = 0; qq < 4; qq++)
#pragma unroll
                for (int t = 0; t < 4; t++)
                    if (16 * t + 15 >= kg) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[qq][t], bq[qq], acc[t], 0, 0, 0);
        }
    }
    if (rin) {
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                const int c = 16 * t + lk + 4 * qq;
                if (c < nbk) P[rr + (int64_t)(jb + c) * m] = acc[t][qq];
            }
    }
}

void launch_trsm_blk(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int jb,
                     double *Lx, const double *Linv)
{
    if (count <= 0) return;
    int rows = max_m - jb - 1;
    if (rows <= 0) return;
    dim3 grid((unsigned)((rows + 63) / 64), (unsigned)count);
    hipLaunchKernelGGL(k_trsm_blk, grid, dim3(256), 0, st, ds, list, jb, Lx, Linv);
}

// ------------------------------------------------------------------------------------------
// Trailing update C -= X X' on 64x64 tiles (FP64 MFMA).  The trailing matrix spans the rest of
// the panel (columns < k, ld = m, in Lx) and the update matrix (columns >= k, ld = u).
__global__ __launch_bounds__(256) void k_syrk_trailing(DevSym ds, const int32_t *__restrict__ list, int jb,
                                                       double *__restrict__ Lx, double *__restrict__ Uo)
{
    const FrontDesc fd = ds.fd[list[blockIdx.z]];
    const int k = fd.k, m = fd.m, u = m - k;
    if (jb >= k) return;
    const int ti = blockIdx.x, tj = blockIdx.y;
    if (tj > ti) return;
    const int nbk = min(NB, k - jb);
    const int t0 = jb + nbk;
    const int r0 = t0 + KVX_TILE * ti, c0 = t0 + KVX_TILE * tj;
    if (r0 >= m) return;
    double *P = Lx + fd.px;
    double *U = Uo + fd.ux;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, lr = l & 15, lk = l >> 4;
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    const int rr = r0 + 16 * w + lr;
    const bool rin = rr < m;
    bool cin[4];
#pragma unroll
    for (int t = 0; t < 4; t++) cin[t] = (c0 + 16 * t + lr) < m;
    // Operand loads of 4 k-steps (20 loads per lane, branch-free) are issued before their 16 MFMAs.
#pragma unroll
    for (int kg = 0; kg < NB; kg += 16) {
        if (kg < nbk) {                             // wave-uniform
            double bq[4], aq[4][4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int kc = kg + 4 * q + lk;
                const bool kin = kc < nbk;
                const int64_t coff = (int64_t)(jb + kc) * m;
                bq[q] = kvx_ld0(P, rr + coff, kin && rin);
#pragma unroll
                for (int t = 0; t < 4; t++) aq[q][t] = kvx_ld0(P, (c0 + 16 * t + lr) + coff, kin && cin[t]);
            }
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[q][t], bq[q], acc[t], 0, 0, 0);
        }
    }
    // lane holds D[i = (l>>4) + 4q][j = l&15] with i <-> tile column, j <-> tile row.
    // Branch-free read-modify-write: all 16 loads go out (clamped addresses), then 16 predicated stores.
    const int rs = min(rr, m - 1);
    double *ptr[4][4];
    double old[4][4];
    bool ok[4][4];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c = c0 + 16 * t + lk + 4 * q;
            ok[t][q] = rin && c <= rr;
            const int cs = min(c, rs);
            ptr[t][q] = (cs < k) ? P + rs + (int64_t)cs * m : U + (rs - k) + (int64_t)(cs - k) * u;
            old[t][q] = *ptr[t][q];
        }
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (ok[t][q]) *ptr[t][q] = old[t][q] - acc[t][q];
}

void launch_syrk_trailing(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int jb,
                          double *Lx, double *Uout)
{
    if (count <= 0) return;
    int rows = max_m - jb - 1;
    if (rows <= 0) return;
    unsigned T = (unsigned)((rows + KVX_TILE - 1) / KVX_TILE);
    dim3 grid(T, T, (unsigned)count);
    hipLaunchKernelGGL(k_syrk_trailing, grid, dim3(256), 0, st, ds, list, jb, Lx, Uout);
}

// ------------------------------------------------------------------------------------------
// Big-front forward solve.  wk (n doubles per rhs) carries the running right-hand side of the
// pivot rows; the update rows accumulate directly in the level's update-vector buffer.
__global__ __launch_bounds__(256) void k_fwd_big_init(DevSym ds, const int32_t *__restrict__ list,
                                                      const double *__restrict__ X, int64_t ldx,
                                                      double *__restrict__ WK, int64_t ldw,
                                                      const double *__restrict__ Wc, double *__restrict__ Wo, int64_t wstride)
{
    const int s = list[blockIdx.x];
    const int k = ds.k[s], m = ds.m[s], u = m - k, f = ds.first[s], tid = threadIdx.x;
    const double *x = X + (int64_t)blockIdx.y * ldx;
    double *wk = WK + (int64_t)blockIdx.y * ldw + f;
    const double *wc = Wc + (int64_t)blockIdx.y * wstride;
    double *wo = Wo + (int64_t)blockIdx.y * wstride + ds.wx[s];
    for (int i = tid; i < k; i += 256) wk[i] = x[f + i];
    for (int i = tid; i < u; i += 256) wo[i] = 0.0;
    __syncthreads();
    for (int64_t c = ds.childptr[s]; c < ds.childptr[s + 1]; c++) {
        const int ch = ds.children[c];
        const int kc = ds.k[ch], uc = ds.m[ch] - kc;
        if (uc == 0) continue;
        const int32_t *rl = ds.rel + ds.rowptr[ch] + kc;
        const double *src = wc + ds.wx[ch];
        for (int i = tid; i < uc; i += 256) {
            const int t = rl[i];
            if (t < k) wk[t] += src[i];
            else wo[t - k] += src[i];
        }
        __syncthreads();
    }
}

// step jb: y_b = Linv_b * w_b (every workgroup recomputes it), x_b := y_b, rows below -= L(:,b) y_b
__global__ __launch_bounds__(256) void k_fwd_big_step(DevSym ds, const int32_t *__restrict__ list, int jb,
                                                      const double *__restrict__ Lx, const double *__restrict__ Linv,
                                                      double *__restrict__ X, int64_t ldx,
                                                      double *__restrict__ WK, int64_t ldw,
                                                      double *__restrict__ Wo, int64_t wstride)
{
    __shared__ double part[4][NB];
    __shared__ double ysh[NB];
    const int s = list[blockIdx.y];
    const int k = ds.k[s], m = ds.m[s], f = ds.first[s], tid = threadIdx.x;
    if (jb >= k) return;
    const int nbk = min(NB, k - jb);
    const int rbase = jb + nbk + blockIdx.x * 256;
    if (blockIdx.x > 0 && rbase >= m) return;
    const double *P = Lx + ds.px[s];
    const double *Y = Linv + ds.linv[s] + (int64_t)(jb / NB) * NB * NB;
    double *x = X + (int64_t)blockIdx.z * ldx + f;
    double *wk = WK + (int64_t)blockIdx.z * ldw + f;
    double *wo = Wo + (int64_t)blockIdx.z * wstride + ds.wx[s];
    {
        const int i = tid & 63, q = tid >> 6;
        double acc = 0.0;
        if (i < nbk)
            for (int p = q; p <= i; p += 4) acc += Y[i + p * NB] * wk[jb + p];
        part[q][i] = acc;
    }
    __syncthreads();
    if (tid < NB) {
        const double y = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
        ysh[tid] = y;
        if (blockIdx.x == 0 && tid < nbk) x[jb + tid] = y;
    }
    __syncthreads();
    const int r = rbase + tid;
    if (r < m) {
        const double *Pr = P + r + (int64_t)jb * m;
        double acc = 0.0;
        for (int j = 0; j < nbk; j++) acc += Pr[(int64_t)j * m] * ysh[j];
        if (r < k) wk[r] -= acc;
        else wo[r - k] -= acc;
    }
}

// Big-front backward solve: t = y - L21' x_below (one wave per pivot column), then block steps
// from the last block to the first.
__global__ __launch_bounds__(256) void k_bwd_big_init(DevSym ds, const int32_t *__restrict__ list,
                                                      const double *__restrict__ Lx, const double *__restrict__ X,
                                                      int64_t ldx, double *__restrict__ WK, int64_t ldw)
{
    const int s = list[blockIdx.y];
    const int k = ds.k[s], m = ds.m[s], f = ds.first[s];
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), ln = threadIdx.x & 63;
    if (c >= k) return;
    const double *x = X + (int64_t)blockIdx.z * ldx;
    double *wk = WK + (int64_t)blockIdx.z * ldw + f;
    const double *Pc = Lx + ds.px[s] + (int64_t)c * m;
    const int32_t *rows = ds.rowidx + ds.rowptr[s];
    double acc = 0.0;
    for (int i = k + ln; i < m; i += 64) acc += Pc[i] * x[rows[i]];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (ln == 0) wk[c] = x[f + c] - acc;
}

__global__ __launch_bounds__(256) void k_bwd_big_step(DevSym ds, const int32_t *__restrict__ list, int bidx,
                                                      const double *__restrict__ Lx, const double *__restrict__ Linv,
                                                      double *__restrict__ X, int64_t ldx,
                                                      double *__restrict__ WK, int64_t ldw)
{
    __shared__ double part[4][NB];
    __shared__ double xsh[NB];
    const int s = list[blockIdx.y];
    const int k = ds.k[s], m = ds.m[s], f = ds.first[s], tid = threadIdx.x;
    const int jb = bidx * NB;
    if (jb >= k) return;
    const int nbk = min(NB, k - jb);
    if (blockIdx.x > 0 && (int)blockIdx.x * 256 >= jb) return;
    const double *P = Lx + ds.px[s];
    const double *Y = Linv + ds.linv[s] + (int64_t)bidx * NB * NB;
    double *x = X + (int64_t)blockIdx.z * ldx + f;
    double *wk = WK + (int64_t)blockIdx.z * ldw + f;
    {
        // x_b = Linv_b' t_b :  x[i] = sum_{p >= i} Y[p][i] t[p]
        const int i = tid & 63, q = tid >> 6;
        double acc = 0.0;
        if (i < nbk)
            for (int p = i + q; p < nbk; p += 4) acc += Y[p + i * NB] * wk[jb + p];
        part[q][i] = acc;
    }
    __syncthreads();
    if (tid < NB) {
        const double v = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
        xsh[tid] = v;
        if (blockIdx.x == 0 && tid < nbk) x[jb + tid] = v;
    }
    __syncthreads();
    const int c = blockIdx.x * 256 + tid;        // earlier pivot column
    if (c < jb) {
        const double *Pc = P + jb + (int64_t)c * m;
        double acc = 0.0;
        for (int i = 0; i < nbk; i++) acc += Pc[i] * xsh[i];
        wk[c] -= acc;
    }
}

void launch_fwd_big(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int max_k,
                    const double *Lx, const double *Linv, double *X, int64_t ldx, int nrhs,
                    double *WK, int64_t ldw, const double *Wchild, double *Wout, int64_t wstride)
{
    if (count <= 0 || nrhs <= 0) return;
    hipLaunchKernelGGL(k_fwd_big_init, dim3((unsigned)count, (unsigned)nrhs), dim3(256), 0, st, ds, list, X, ldx, WK, ldw,
                       Wchild, Wout, wstride);
    for (int jb = 0; jb < max_k; jb += NB) {
        int rows = max_m - jb - 1;
        unsigned gx = (unsigned)std::max(1, (rows + 255) / 256);
        hipLaunchKernelGGL(k_fwd_big_step, dim3(gx, (unsigned)count, (unsigned)nrhs), dim3(256), 0, st, ds, list, jb, Lx, Linv,
                           X, ldx, WK, ldw, Wout, wstride);
    }
}

void launch_bwd_big(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int max_k,
                    const double *Lx, const double *Linv, double *X, int64_t ldx, int nrhs, double *WK, int64_t ldw)
{
    if (count <= 0 || nrhs <= 0) return;
    (void)max_m;
    hipLaunchKernelGGL(k_bwd_big_init, dim3((unsigned)((max_k + 3) / 4), (unsigned)count, (unsigned)nrhs), dim3(256), 0, st,
                       ds, list, Lx, X, ldx, WK, ldw);
    for (int b = (max_k + NB - 1) / NB - 1; b >= 0; b--) {
        unsigned gx = (unsigned)std::max(1, (b * NB + 255) / 256);
        hipLaunchKernelGGL(k_bwd_big_step, dim3(gx, (unsigned)count, (unsigned)nrhs), dim3(256), 0, st, ds, list, b, Lx, Linv,
                           X, ldx, WK, ldw);
    }
}

}  // namespace kvx
