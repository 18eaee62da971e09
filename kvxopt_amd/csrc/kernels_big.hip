// Big fronts (order m > KVX_SMALL_MAX): blocked right-looking factorisation in HBM/L2 with
// 64-column panel steps, and multi-workgroup triangular solves.
//
// Per panel step jb (three dependent launches, every big front of the level batched):
//   k_potrf_blk   : Cholesky of the 64x64 diagonal block AND its inverse, register-resident
//                   (512 threads: waves 0-3 factor, waves 4-7 carry the inverse), one barrier per
//                   column;
//   k_trsm_blk    : X := A * Linv' for the rows below, FP64 MFMA (v_mfma_f64_16x16x4_f64);
//   k_syrk_trailing: C -= X X' on 64x64 tiles, FP64 MFMA.
// The inverses of the diagonal blocks stay resident: the solves use them as 64x64 mat-vecs, so a
// big front's triangular solve has no 64-long dependent chain and can be spread over workgroups.
//
// Reference role: cholmod_l_factorize / cholmod_l_solve (src/C/cholmod.c:362, 483).
#include "device.hpp"

#include <algorithm>
#include <utility>

namespace kvx {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NB = KVX_NB;

__device__ inline double readlane_d(double v, int lane)
{
    const long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Register-resident 64x64 Cholesky + inverse, one barrier per column.
// Thread (i = tid & 63, q = (tid >> 6) & 3): the factor role (waves 0-3) holds D[i][q + 4(g + t)] in
// a[t] while column group g (columns 4g .. 4g+3) is eliminated -- after each group the finished
// column is stored and the register window shifts down, so the loop body is compiled once (a fully
// unrolled 64-step sweep is 125 KB of code and runs at instruction-fetch speed).  The inverse role
// (waves 4-7) builds Y = D^{-1} by forward substitution on the identity with the same column
// sweep; it holds Y[i][q + 4(g - t)] in a[t] (window shifts up; columns "< 0" are zero).
// Blocks shorter than 64 are padded with the identity.
template <int JS>
__device__ __forceinline__ void potrf_substep(double (&a)[16], double (*colbuf)[NB + 1], int g, int i, int q,
                                              bool inv_role, int *status, int col0)
{
    const int j = 4 * g + JS;
    double *cb = colbuf[JS & 1];
    if (!inv_role && q == JS) {
        double v = a[0];
        double d = readlane_d(v, j);
        if (!(d > 0.0)) {
            if (i == 0) atomicMin(status, col0 + j);
            d = 1.0;
        }
        double ljj, inv;
        kvx_sqrt_rsqrt(d, ljj, inv);
        v = (i == j) ? ljj : (i > j ? v * inv : 0.0);
        a[0] = v;
        cb[i] = v;
        if (i == 0) cb[NB] = inv;
    }
    __syncthreads();
    const double li = cb[i];      // column j of L: 0 above the diagonal, l_jj on it
    // Updates are branch-free per register and skip dead windows with SCALAR branches only (g, q, JS
    // are wave-uniform).  Entries above the diagonal (c > i) and windows past column 63 carry
    // don't-care values that are never stored, so they need no per-lane predicate.
    if (!inv_role) {
        const int nlive = 16 - g;                          // windows t < nlive map to columns < 64
#pragma unroll
        for (int t = 0; t < 16; t++) {
            if (t < nlive) {
                const double lc = cb[q + 4 * (g + t)];
                if (t > 0 || q > JS) a[t] = __builtin_fma(-li, lc, a[t]);
            }
        }
    } else {
        const double invd = cb[NB];
        const bool piv = (i == j);
        const double lo = piv ? 0.0 : li;                  // rows above j have li == 0 already
#pragma unroll
        for (int t = 0; t < 16; t++) {
            if (t <= g && (t > 0 || q <= JS)) {            // columns c = q + 4(g - t) in [0, j]
                const double yj = readlane_d(a[t], j) * invd;
                const double v = __builtin_fma(-lo, yj, a[t]);
                a[t] = piv ? yj : v;
            }
        }
    }
}

__global__ __launch_bounds__(512) void k_potrf_blk(DevSym ds, const int32_t *__restrict__ list, int jb,
                                                   double *__restrict__ Lx, double *__restrict__ Linv, int *status)
{
    __shared__ double colbuf[2][NB + 1];       // column j of L, plus 1/l_jj in slot NB
    const int s = list[blockIdx.x];
    const int k = ds.k[s], m = ds.m[s];
    if (jb >= k) return;
    const int nbk = min(NB, k - jb);
    const int tid = threadIdx.x, i = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);       // wave-uniform: keep it scalar
    const bool inv_role = __builtin_amdgcn_readfirstlane(tid >> 8) != 0;
    double *P = Lx + ds.px[s];
    double *Yg = Linv + ds.linv[s] + (int64_t)(jb / NB) * NB * NB;
    const int col0 = ds.first[s] + jb;
    double a[16];
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const int c = q + 4 * t;
        if (inv_role) a[t] = 0.0;
        else if (i < nbk && c <= i) a[t] = P[(jb + i) + (int64_t)(jb + c) * m];
        else a[t] = (c == i) ? 1.0 : 0.0;
    }
    const int ngrp = (nbk + 3) >> 2;                       // identity padding beyond nbk needs no sweep
    for (int g = 0; g < ngrp; g++) {
        if (inv_role) {
            // open the window on column group g: Y starts as the identity
#pragma unroll
            for (int t = 15; t > 0; t--) a[t] = a[t - 1];
            a[0] = (q + 4 * g == i) ? 1.0 : 0.0;
        }
        potrf_substep<0>(a, colbuf, g, i, q, inv_role, status, col0);
        potrf_substep<1>(a, colbuf, g, i, q, inv_role, status, col0);
        potrf_substep<2>(a, colbuf, g, i, q, inv_role, status, col0);
        potrf_substep<3>(a, colbuf, g, i, q, inv_role, status, col0);
        if (!inv_role) {
            const int c = q + 4 * g;                       // this thread's finished column
            if (i < nbk && c <= i) P[(jb + i) + (int64_t)(jb + c) * m] = a[0];
#pragma unroll
            for (int t = 0; t < 15; t++) a[t] = a[t + 1];
            a[15] = 0.0;
        }
    }
    if (inv_role && i < nbk) {
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int c = q + 4 * (ngrp - 1 - t);
            if (c >= 0 && c <= i) Yg[i + c * NB] = a[t];
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
// trailing update (D[i][j]: i <-> panel column, j <-> row) so that stores run along rows.
__global__ __launch_bounds__(256) void k_trsm_blk(DevSym ds, const int32_t *__restrict__ list, int jb,
                                                  double *__restrict__ Lx, const double *__restrict__ Linv)
{
    const int s = list[blockIdx.y];
    const int k = ds.k[s], m = ds.m[s];
    if (jb >= k) return;
    const int nbk = min(NB, k - jb);
    const int r0 = jb + nbk + blockIdx.x * 64;
    if (r0 >= m) return;
    double *P = Lx + ds.px[s];
    const double *Y = Linv + ds.linv[s] + (int64_t)(jb / NB) * NB * NB;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, lr = l & 15, lk = l >> 4;
    const int rr = r0 + 16 * w + lr;
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int ks = 0; ks < nbk; ks += 4) {
        const int kc = ks + lk;
        const bool kin = kc < nbk;
        const double b = (kin && rr < m) ? P[rr + (int64_t)(jb + kc) * m] : 0.0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (16 * t + 15 < ks) continue;            // Linv is lower triangular: Y[c][p] = 0 for p > c
            const int cc = 16 * t + lr;
            const double a = (kin && cc < nbk && kc <= cc) ? Y[cc + kc * NB] : 0.0;
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
    }
    if (rr < m) {
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
    const int s = list[blockIdx.z];
    const int k = ds.k[s], m = ds.m[s], u = m - k;
    if (jb >= k) return;
    const int ti = blockIdx.x, tj = blockIdx.y;
    if (tj > ti) return;
    const int nbk = min(NB, k - jb);
    const int t0 = jb + nbk;
    const int r0 = t0 + KVX_TILE * ti, c0 = t0 + KVX_TILE * tj;
    if (r0 >= m) return;
    double *P = Lx + ds.px[s];
    double *U = Uo + ds.ux[s];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, lr = l & 15, lk = l >> 4;
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    const int rr = r0 + 16 * w + lr;
    for (int ks = 0; ks < nbk; ks += 4) {
        const int kc = ks + lk;
        const bool kin = kc < nbk;
        const int64_t coff = (int64_t)(jb + kc) * m;
        const double b = (kin && rr < m) ? P[rr + coff] : 0.0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int cc = c0 + 16 * t + lr;
            const double a = (kin && cc < m) ? P[cc + coff] : 0.0;
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
    }
    // lane holds D[i = (l>>4) + 4q][j = l&15] with i <-> tile column, j <-> tile row
    if (rr < m) {
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                const int c = c0 + 16 * t + lk + 4 * qq;
                if (c <= rr) {
                    if (c < k) P[rr + (int64_t)c * m] -= acc[t][qq];
                    else U[(rr - k) + (int64_t)(c - k) * u] -= acc[t][qq];
                }
            }
    }
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
