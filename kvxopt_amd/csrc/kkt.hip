// HIP kernels for the parts of the KKT step outside the Cholesky factorisation:
//  * Nesterov-Todd scaling on the orthant ('l') cone -- reference src/python/misc.py:284-287,
//    444-464, 951-952 and src/C/misc_solvers.c:132-141, 287-298, 662-669, 793-800, 1018, 1065-1071;
//  * S = G' diag(w) G (+P) on a fixed pattern -- misc.py:1418-1426,1451-1455 -> src/C/sparse.c:
//    1260-1283, 2176-2256, restated as a precomputed product list (one gather-multiply-reduce per
//    stored entry of S instead of the reference's sparse-accumulator scatter per column);
//  * CCS sparse mat-vec -- src/C/sparse.c:1073-1104.
// All of it is HBM/latency-bound streaming work: coalesced grid-stride loops, wavefront (64-lane)
// shuffles for the reductions, no MFMA.
#include "kkt.hpp"
#include <cstdlib>
#include <algorithm>

namespace kvx {

static inline unsigned grid_for(int64_t n, int bs = 256)
{
    int64_t b = (n + bs - 1) / bs;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// a product that stays a product (never contracted into a following addition)
__device__ __forceinline__ double kvx_mul_rn(double a, double b)
{
#pragma clang fp contract(off)
    return a * b;
}

#define GS_LOOP(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

__global__ void k_compute_scaling(int64_t n, const double *__restrict__ s, const double *__restrict__ z,
                                  double *__restrict__ d, double *__restrict__ di, double *__restrict__ lm)
{
    GS_LOOP(i, n) {
        const double si = s[i], zi = z[i];
        const double dd = sqrt(si / zi);
        d[i] = dd;
        di[i] = 1.0 / dd;
        lm[i] = sqrt(si * zi);
    }
}
__global__ void k_update_scaling(int64_t n, double *__restrict__ s, double *__restrict__ z, double *__restrict__ d,
                                 double *__restrict__ di, double *__restrict__ lm)
{
    GS_LOOP(i, n) {
        const double ss = sqrt(s[i]), zz = sqrt(z[i]);
        s[i] = ss;
        z[i] = zz;
        const double dd = (d[i] * ss) / zz;
        d[i] = dd;
        di[i] = 1.0 / dd;
        lm[i] = ss * zz;
    }
}
// ---- fused 'l'-cone steps of the interior-point iteration -----------------------------------------------------------
// The loop is host-bound (one launch + one ctypes call per BLAS-1 operation of the reference); these three kernels
// each replace a fixed run of 6-10 of them.  Same operations in the same order per element.
// (1) right-hand side of a Newton system and the first half of f6_no_ir (coneprog.py:1250-1298, 1146-1157):
//       ds := -(lmbdasq (+ ws3 - shift)) ./ lmbda ;   dz := -(scale * rz + d .* ds)
__global__ void k_lp_newton_rhs(int64_t n, const double *__restrict__ lsq, const double *__restrict__ ws3, double shift,
                                double scale, const double *__restrict__ rz, const double *__restrict__ lm,
                                const double *__restrict__ d, double *__restrict__ ds, double *__restrict__ dz)
{
    GS_LOOP(i, n) {
        double v;
        if (lsq) v = lsq[i];
        else { const double l0 = lm[i]; v = kvx_mul_rn(l0, l0); }             // lmbda o lmbda, rounded as misc.ssqr's product
        if (ws3) v = (v + ws3[i]) - shift;
        v = -(v / lm[i]);
        ds[i] = v;
        dz[i] = -(scale * rz[i] + v * d[i]);
    }
}
// (2) second half of f6_no_ir and the step-length scaling (coneprog.py:1186-1191, 1303-1316):
//       dz += dtau * z1 ;  ds -= dz ;  [ws3 := ds .* dz] ;  ds ./= lmbda ;  dz ./= lmbda
__global__ void k_lp_step_post(int64_t n, double dtau, const double *__restrict__ z1, const double *__restrict__ lm,
                               double *__restrict__ ds, double *__restrict__ dz, double *__restrict__ ws3)
{
    GS_LOOP(i, n) {
        const double zz = dz[i] + dtau * z1[i];
        const double ss = ds[i] - zz;
        if (ws3) ws3[i] = ss * zz;
        const double l = lm[i];
        ds[i] = ss / l;
        dz[i] = zz / l;
    }
}
// (3) end of the iteration (coneprog.py:1343-1432, 'l' block): ds := (step ds + 1) .* lmbda, same for dz, then
//     update_scaling (misc.py:444-464) and the unscaled iterates s = W' lmbda, z = W^-1 lmbda
__device__ __forceinline__ void lp_update_elem(int64_t i, double step, double *__restrict__ ds, double *__restrict__ dz,
                                               double *__restrict__ d, double *__restrict__ di, double *__restrict__ lm,
                                               double *__restrict__ s, double *__restrict__ z)
{
    const double l = lm[i];
    const double ss = sqrt((step * ds[i] + 1.0) * l), zz = sqrt((step * dz[i] + 1.0) * l);
    ds[i] = ss;
    dz[i] = zz;
    const double dd = (d[i] * ss) / zz;
    d[i] = dd;
    const double dinv = 1.0 / dd;
    di[i] = dinv;
    const double ln = ss * zz;
    lm[i] = ln;
    s[i] = ln * dd;
    z[i] = ln * dinv;
}
__global__ void k_lp_update(int64_t n, double step, double *__restrict__ ds, double *__restrict__ dz, double *__restrict__ d,
                            double *__restrict__ di, double *__restrict__ lm, double *__restrict__ s, double *__restrict__ z)
{
    GS_LOOP(i, n) lp_update_elem(i, step, ds, dz, d, di, lm, s, z);
}

// ---- second half of f6_no_ir with dtau kept on the device (coneprog.py:1162-1195, 1303-1316): the host used to fetch the
// three inner products, form dtau, launch the updates and fetch the two step bounds -- two round trips; here dtau is formed
// by a one-thread kernel from the reduction results and the dependent kernels read it from memory: one round trip.
//   r[0..2] = c'dx, b'dy, th'dz ; r[3] = z1'z1 (or the host value when r3_host >= 0)
//   out[0] = dtau = dgi (dtau0 + r0 + r1 + r2) / (1 + z1z1) ; out[1] = z1z1
__global__ void k_lp_dtau(const double *__restrict__ r, double dgi, double dtau0, double z1z1_host, int use_host, double *__restrict__ out)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const double zz = use_host ? z1z1_host : r[3];
        out[0] = dgi * (dtau0 + r[0] + r[1] + r[2]) / (1.0 + zz);
        out[1] = zz;
    }
}
__global__ void k_axpy_devalpha(int64_t n, const double *__restrict__ alpha, const double *__restrict__ x, double *__restrict__ y)
{
    const double a = alpha[0];
    GS_LOOP(i, n) y[i] += a * x[i];
}
__global__ void k_lp_step_post_dev(int64_t n, const double *__restrict__ dtau_p, const double *__restrict__ z1, const double *__restrict__ lm,
                                   double *__restrict__ ds, double *__restrict__ dz, double *__restrict__ ws3)
{
    const double dtau = dtau_p[0];
    GS_LOOP(i, n) {
        const double zz = dz[i] + dtau * z1[i];
        const double ss = ds[i] - zz;
        if (ws3) ws3[i] = ss * zz;
        const double l = lm[i];
        ds[i] = ss / l;
        dz[i] = zz / l;
    }
}
void launch_lp_dtau(hipStream_t st, const double *r, double dgi, double dtau0, double z1z1_host, int use_host, double *out)
{ hipLaunchKernelGGL(k_lp_dtau, dim3(1), dim3(64), 0, st, r, dgi, dtau0, z1z1_host, use_host, out); }
void launch_axpy_devalpha(hipStream_t st, int64_t n, const double *alpha, const double *x, double *y)
{ if (n > 0) hipLaunchKernelGGL(k_axpy_devalpha, dim3(grid_for(n)), dim3(256), 0, st, n, alpha, x, y); }
void launch_lp_step_post_devalpha(hipStream_t st, int64_t n, const double *dtau, const double *z1, const double *lm, double *ds, double *dz,
                                  double *ws3)
{ if (n > 0) hipLaunchKernelGGL(k_lp_step_post_dev, dim3(grid_for(n)), dim3(256), 0, st, n, dtau, z1, lm, ds, dz, ws3); }

void launch_lp_newton_rhs(hipStream_t st, int64_t n, const double *lsq, const double *ws3, double shift, double scale,
                          const double *rz, const double *lm, const double *d, double *ds, double *dz)
{ if (n > 0) hipLaunchKernelGGL(k_lp_newton_rhs, dim3(grid_for(n)), dim3(256), 0, st, n, lsq, ws3, shift, scale, rz, lm, d, ds, dz); }
void launch_lp_step_post(hipStream_t st, int64_t n, double dtau, const double *z1, const double *lm, double *ds, double *dz, double *ws3)
{ if (n > 0) hipLaunchKernelGGL(k_lp_step_post, dim3(grid_for(n)), dim3(256), 0, st, n, dtau, z1, lm, ds, dz, ws3); }
void launch_lp_update(hipStream_t st, int64_t n, double step, double *ds, double *dz, double *d, double *di, double *lm, double *s, double *z)
{ if (n > 0) hipLaunchKernelGGL(k_lp_update, dim3(grid_for(n)), dim3(256), 0, st, n, step, ds, dz, d, di, lm, s, z); }

__global__ void k_scale(int64_t n, int64_t ldx, double *__restrict__ x, const double *__restrict__ w)
{
    double *xc = x + (int64_t)blockIdx.y * ldx;
    GS_LOOP(i, n) xc[i] *= w[i];
}
__global__ void k_div(int64_t n, double *__restrict__ x, const double *__restrict__ y) { GS_LOOP(i, n) x[i] /= y[i]; }
__global__ void k_mul(int64_t n, double *__restrict__ x, const double *__restrict__ y) { GS_LOOP(i, n) x[i] *= y[i]; }
__global__ void k_sqr(int64_t n, double *__restrict__ x, const double *__restrict__ y) { GS_LOOP(i, n) { const double v = y[i]; x[i] = v * v; } }

__global__ void k_axpy(int64_t n, double alpha, const double *__restrict__ x, double *__restrict__ y) { GS_LOOP(i, n) y[i] += alpha * x[i]; }
__global__ void k_scal(int64_t n, double alpha, double *__restrict__ x) { GS_LOOP(i, n) x[i] *= alpha; }
__global__ void k_addc(int64_t n, double c, double *__restrict__ x) { GS_LOOP(i, n) x[i] += c; }
__global__ void k_fill(int64_t n, double c, double *__restrict__ x) { GS_LOOP(i, n) x[i] = c; }
// z := a * (x .* y) + b * z
__global__ void k_xmy(int64_t n, double a, const double *__restrict__ x, const double *__restrict__ y, double b, double *__restrict__ z)
{ GS_LOOP(i, n) z[i] = a * (x[i] * y[i]) + (b == 0.0 ? 0.0 : b * z[i]); }
// y := alpha * A x + beta * y for a dense column-major m x n matrix (ld = lda), one thread per row: the reads of a column are
// coalesced across the threads, x is read by every thread (cached).  nrhs columns of x / y (ldx, ldy) in grid.y.
__global__ __launch_bounds__(256) void k_dense_gemv(int64_t m, int64_t n, double alpha, const double *__restrict__ A, int64_t lda,
                                                    const double *__restrict__ x, int64_t ldx, double beta, double *__restrict__ y,
                                                    int64_t ldy)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const double *xc = x + (int64_t)blockIdx.y * ldx;
    double *yc = y + (int64_t)blockIdx.y * ldy;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    int64_t j = 0;
    for (; j + 4 <= n; j += 4) {
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = __builtin_fma(A[i + (j + q) * lda], xc[j + q], acc[q]);
    }
    for (; j < n; j++) acc[0] = __builtin_fma(A[i + j * lda], xc[j], acc[0]);
    const double r = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    yc[i] = (beta == 0.0) ? alpha * r : __builtin_fma(alpha, r, beta * yc[i]);
}
void launch_dense_gemv(hipStream_t st, int64_t m, int64_t n, int64_t nrhs, double alpha, const double *A, int64_t lda, const double *x,
                       int64_t ldx, double beta, double *y, int64_t ldy)
{
    if (m <= 0 || nrhs <= 0) return;
    hipLaunchKernelGGL(k_dense_gemv, dim3((unsigned)((m + 255) / 256), (unsigned)nrhs), dim3(256), 0, st, m, n, alpha, A, lda, x, ldx, beta, y, ldy);
}

// z := a * x + b * y (b == 0: z := a * x, y is not read); z may be x or y
__global__ void k_lincomb(int64_t n, double a, const double *x, double b, const double *y, double *z)
{ GS_LOOP(i, n) z[i] = (b == 0.0) ? a * x[i] : __builtin_fma(b, y[i], a * x[i]); }
void launch_lincomb(hipStream_t st, int64_t n, double a, const double *x, double b, const double *y, double *z)
{ if (n > 0) hipLaunchKernelGGL(k_lincomb, dim3(grid_for(n)), dim3(256), 0, st, n, a, x, b, y, z); }
void launch_axpy(hipStream_t st, int64_t n, double alpha, const double *x, double *y)
{ if (n > 0) hipLaunchKernelGGL(k_axpy, dim3(grid_for(n)), dim3(256), 0, st, n, alpha, x, y); }
void launch_vscal(hipStream_t st, int64_t n, double alpha, double *x)
{ if (n > 0) hipLaunchKernelGGL(k_scal, dim3(grid_for(n)), dim3(256), 0, st, n, alpha, x); }
void launch_addc(hipStream_t st, int64_t n, double c, double *x)
{ if (n > 0) hipLaunchKernelGGL(k_addc, dim3(grid_for(n)), dim3(256), 0, st, n, c, x); }
void launch_fill(hipStream_t st, int64_t n, double c, double *x)
{ if (n > 0) hipLaunchKernelGGL(k_fill, dim3(grid_for(n)), dim3(256), 0, st, n, c, x); }
void launch_xmy(hipStream_t st, int64_t n, double a, const double *x, const double *y, double b, double *z)
{ if (n > 0) hipLaunchKernelGGL(k_xmy, dim3(grid_for(n)), dim3(256), 0, st, n, a, x, y, b, z); }

void launch_compute_scaling(hipStream_t st, int64_t n, const double *s, const double *z, double *d, double *di, double *lm)
{ if (n > 0) hipLaunchKernelGGL(k_compute_scaling, dim3(grid_for(n)), dim3(256), 0, st, n, s, z, d, di, lm); }
void launch_update_scaling(hipStream_t st, int64_t n, double *s, double *z, double *d, double *di, double *lm)
{ if (n > 0) hipLaunchKernelGGL(k_update_scaling, dim3(grid_for(n)), dim3(256), 0, st, n, s, z, d, di, lm); }
void launch_scale(hipStream_t st, int64_t n, int64_t ncols, int64_t ldx, double *x, const double *w)
{ if (n > 0 && ncols > 0) hipLaunchKernelGGL(k_scale, dim3(grid_for(n), (unsigned)ncols), dim3(256), 0, st, n, ldx, x, w); }
void launch_div(hipStream_t st, int64_t n, double *x, const double *y)
{ if (n > 0) hipLaunchKernelGGL(k_div, dim3(grid_for(n)), dim3(256), 0, st, n, x, y); }
void launch_mul(hipStream_t st, int64_t n, double *x, const double *y)
{ if (n > 0) hipLaunchKernelGGL(k_mul, dim3(grid_for(n)), dim3(256), 0, st, n, x, y); }
void launch_sqr(hipStream_t st, int64_t n, double *x, const double *y)
{ if (n > 0) hipLaunchKernelGGL(k_sqr, dim3(grid_for(n)), dim3(256), 0, st, n, x, y); }

// ---- reductions: fixed partial-sum tree (bitwise reproducible) -------------------------------
constexpr int RED_BLOCKS = 256;

template <bool MAXNEG>
__device__ inline double wave_red(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double t = __shfl_xor(v, o);
        v = MAXNEG ? fmax(v, t) : v + t;
    }
    return v;
}

template <bool MAXNEG>
__global__ __launch_bounds__(256) void k_reduce1(int64_t n, const double *__restrict__ x, const double *__restrict__ y,
                                                 double *__restrict__ part)
{
    __shared__ double sh[4];
    double acc = MAXNEG ? -1.7976931348623157e308 : 0.0;
    // contiguous chunk per block keeps the summation order independent of the grid-stride
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t b0 = per * blockIdx.x, b1 = min(n, b0 + per);
    for (int64_t i = b0 + threadIdx.x; i < b1; i += 256) {
        if (MAXNEG) acc = fmax(acc, -x[i]);
        else acc += x[i] * y[i];
    }
    acc = wave_red<MAXNEG>(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sh[0];
        for (int w = 1; w < 4; w++) r = MAXNEG ? fmax(r, sh[w]) : r + sh[w];
        part[blockIdx.x] = r;
    }
}
template <bool MAXNEG>
__global__ __launch_bounds__(64) void k_reduce2(int nb, const double *__restrict__ part, double *__restrict__ out)
{
    double acc = MAXNEG ? -1.7976931348623157e308 : 0.0;
    for (int i = threadIdx.x; i < nb; i += 64) acc = MAXNEG ? fmax(acc, part[i]) : acc + part[i];
    acc = wave_red<MAXNEG>(acc);
    if (threadIdx.x == 0) *out = acc;
}

void launch_dot(hipStream_t st, int64_t n, const double *x, const double *y, double *part, double *out)
{
    hipLaunchKernelGGL(k_reduce1<false>, dim3(RED_BLOCKS), dim3(256), 0, st, n, x, y, part);
    hipLaunchKernelGGL(k_reduce2<false>, dim3(1), dim3(64), 0, st, RED_BLOCKS, part, out);
}
void launch_maxneg(hipStream_t st, int64_t n, const double *x, double *part, double *out)
{
    hipLaunchKernelGGL(k_reduce1<true>, dim3(RED_BLOCKS), dim3(256), 0, st, n, x, x, part);
    hipLaunchKernelGGL(k_reduce2<true>, dim3(1), dim3(64), 0, st, RED_BLOCKS, part, out);
}
// Several reductions in two launches (kvx_nt_reduce_multi_dev: the residual norms and objectives of an interior-point iteration):
// blockIdx.y = reduction; per reduction the same chunking and summation order as k_reduce1 / k_reduce2, so the values are the
// single launches' bit for bit.
__global__ __launch_bounds__(256) void k_reduce1_multi(MultiRed mr, double *__restrict__ part)
{
    __shared__ double sh[4];
    const int r = blockIdx.y;
    const bool mx = mr.kind[r] != 0;
    const int64_t n = mr.n[r];
    const double *x = mr.x[r], *y = mr.y[r];
    double acc = mx ? -1.7976931348623157e308 : 0.0;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t b0 = per * blockIdx.x, b1 = min(n, b0 + per);
    if (mx) {
        for (int64_t i = b0 + threadIdx.x; i < b1; i += 256) acc = fmax(acc, -x[i]);
        acc = wave_red<true>(acc);
    } else {
        for (int64_t i = b0 + threadIdx.x; i < b1; i += 256) acc += x[i] * y[i];
        acc = wave_red<false>(acc);
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = sh[0];
        for (int w = 1; w < 4; w++) t = mx ? fmax(t, sh[w]) : t + sh[w];
        part[(int64_t)r * RED_BLOCKS + blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(64) void k_reduce2_multi(MultiRed mr, const double *__restrict__ part, double *__restrict__ out)
{
    const int r = blockIdx.x;
    const bool mx = mr.kind[r] != 0;
    const double *pr = part + (int64_t)r * RED_BLOCKS;
    double acc = mx ? -1.7976931348623157e308 : 0.0;
    if (mx) {
        for (int i = threadIdx.x; i < RED_BLOCKS; i += 64) acc = fmax(acc, pr[i]);
        acc = wave_red<true>(acc);
    } else {
        for (int i = threadIdx.x; i < RED_BLOCKS; i += 64) acc = acc + pr[i];
        acc = wave_red<false>(acc);
    }
    if (threadIdx.x == 0) out[r] = acc;
}
void launch_reduce_multi(hipStream_t st, const MultiRed &mr, double *part, double *out)
{
    if (mr.count <= 0) return;
    hipLaunchKernelGGL(k_reduce1_multi, dim3(RED_BLOCKS, (unsigned)mr.count), dim3(256), 0, st, mr, part);
    hipLaunchKernelGGL(k_reduce2_multi, dim3((unsigned)mr.count), dim3(64), 0, st, mr, part, out);
}
int reduce_scratch_doubles() { return 32 * RED_BLOCKS + 1; }

// ---- S = G' diag(w) G (+ P) on a fixed pattern ----------------------------------------------------
// Two launches.  (1) k_atda_scale streams G once: wg[a] = w[row(a)] * G[a] (coalesced reads of the CCS arrays, the weights
// gathered along a column's ascending rows).  (2) k_atda: FOUR lanes per stored S entry e walk its product list
// [pp[e], pp[e + 1]) with stride four -- acc += wg[pa[q]] * G[pb[q]] -- and meet by two shuffles in a fixed order (bitwise
// reproducible).  A product costs ONE scattered 8-byte gather (wg[pa]: the partner entry of the row sits in another column);
// G[pb] runs along the entry's own column.  The form this replaces -- one lane per entry, three dependent gathers
// w[gi[pa]] * G[pa] * G[pb] per product, a grid capped at 2048 workgroups -- took 80 us on the random-pattern calibration of
// BASELINE.md (ml = 2e5, 4 entries per row, nnz(S) = 1.25e6: entries with 16 products serialised 16 dependent round trips).
template <bool SQ>                                    // SQ: w holds di, the weight is its square (misc.py:1420: W^-1 applied twice)
__global__ void k_atda_scale(int64_t gnz, const int32_t *__restrict__ gi, const double *__restrict__ gx, const double *__restrict__ w,
                             double *__restrict__ wg)
{
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a < gnz) {
        const double v = w[gi[a]];
        const double wv = SQ ? v * v : v;
        wg[a] = wv * gx[a];
    }
}
template <int LPE>                                   // lanes per entry: 1, 2 or 4
__global__ __launch_bounds__(256) void k_atda(int64_t snz, const int64_t *__restrict__ pp, const int32_t *__restrict__ pa,
                                              const int32_t *__restrict__ pb, const double *__restrict__ wg, const double *__restrict__ gx,
                                              double *__restrict__ sx)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t e = t / LPE;
    const int sub = (int)(t % LPE);
    double acc = 0.0;
    if (e < snz) {
        const int64_t q1 = pp[e + 1];
        for (int64_t q = pp[e] + sub; q < q1; q += LPE) acc += wg[pa[q]] * gx[pb[q]];
    }
    if (LPE >= 2) acc += __shfl_xor(acc, 1);
    if (LPE >= 4) acc += __shfl_xor(acc, 2);
    if (sub == 0 && e < snz) sx[e] = acc;
}
__global__ void k_add_at(int64_t pnz, const int64_t *__restrict__ slot, const double *__restrict__ px, double *__restrict__ sx)
{
    GS_LOOP(q, pnz) sx[slot[q]] += px[q];
}
void launch_atda(hipStream_t st, int64_t snz, int64_t gnz, const int64_t *pp, const int32_t *pa, const int32_t *pb,
                 const int32_t *gi, const double *gx, const double *w, double *wg, double *sx, bool w_is_di)
{
    if (snz <= 0) return;
    if (gnz > 0 && w_is_di) hipLaunchKernelGGL(k_atda_scale<true>, dim3((unsigned)((gnz + 255) / 256)), dim3(256), 0, st, gnz, gi, gx, w, wg);
    else if (gnz > 0) hipLaunchKernelGGL(k_atda_scale<false>, dim3((unsigned)((gnz + 255) / 256)), dim3(256), 0, st, gnz, gi, gx, w, wg);
    static const int lpe = [] { const char *e = getenv("KVX_ATDA_LPE"); return e ? atoi(e) : 4; }();
    if (lpe >= 4) hipLaunchKernelGGL(k_atda<4>, dim3((unsigned)((4 * snz + 255) / 256)), dim3(256), 0, st, snz, pp, pa, pb, wg, gx, sx);
    else if (lpe == 2) hipLaunchKernelGGL(k_atda<2>, dim3((unsigned)((2 * snz + 255) / 256)), dim3(256), 0, st, snz, pp, pa, pb, wg, gx, sx);
    else hipLaunchKernelGGL(k_atda<1>, dim3((unsigned)((snz + 255) / 256)), dim3(256), 0, st, snz, pp, pa, pb, wg, gx, sx);
}
void launch_add_at(hipStream_t st, int64_t pnz, const int64_t *slot, const double *px, double *sx)
{ if (pnz > 0) hipLaunchKernelGGL(k_add_at, dim3(grid_for(pnz)), dim3(256), 0, st, pnz, slot, px, sx); }

// ---- CCS mat-vec -------------------------------------------------------------------------------------
// 'T': y_j = beta*y_j + alpha * <A(:,j), x>  -- one 16-lane group per column (columns are short).
__global__ __launch_bounds__(256) void k_spmv_t(int64_t n, const int64_t *__restrict__ Ap, const int64_t *__restrict__ Ai,
                                                const double *__restrict__ Ax, double alpha, const double *__restrict__ x,
                                                double beta, double *__restrict__ y)
{
    const int sub = threadIdx.x & 15;
    int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 4;
    for (; j < n; j += stride) {
        double acc = 0.0;
        for (int64_t p = Ap[j] + sub; p < Ap[j + 1]; p += 16) acc += Ax[p] * x[Ai[p]];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (sub == 0) y[j] = (beta == 0.0 ? 0.0 : beta * y[j]) + alpha * acc;
    }
}
__global__ void k_beta(int64_t n, double beta, double *__restrict__ y)
{
    GS_LOOP(i, n) y[i] = (beta == 0.0) ? 0.0 : beta * y[i];
}
// 'N': y += alpha * A x by column scatter with hardware FP64 atomics (order-dependent rounding;
// the kvx_spmat handle offers the reproducible row-gather form).
__global__ void k_spmv_n(int64_t n, const int64_t *__restrict__ Ap, const int64_t *__restrict__ Ai,
                         const double *__restrict__ Ax, double alpha, const double *__restrict__ x, double *__restrict__ y)
{
    GS_LOOP(j, n) {
        const double xj = alpha * x[j];
        for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) unsafeAtomicAdd(&y[Ai[p]], Ax[p] * xj);
    }
}
void launch_spmv(hipStream_t st, int trans, int64_t m, int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax,
                 double alpha, const double *x, double beta, double *y)
{
    if (trans == 'T') {
        if (n > 0) hipLaunchKernelGGL(k_spmv_t, dim3(grid_for(n * 16)), dim3(256), 0, st, n, Ap, Ai, Ax, alpha, x, beta, y);
    } else {
        if (m > 0 && beta != 1.0) hipLaunchKernelGGL(k_beta, dim3(grid_for(m)), dim3(256), 0, st, m, beta, y);
        if (n > 0) hipLaunchKernelGGL(k_spmv_n, dim3(grid_for(n)), dim3(256), 0, st, n, Ap, Ai, Ax, alpha, x, y);
    }
}

// ---- dense helpers of the equality-constrained KKT path with a general S (misc.py:1476-1487): K = A S^-1 A' is formed
// as a dense p x p matrix from X = S^-1 A' (n x p, multi-right-hand-side solve) when p is moderate ------------------
// Y(j, c) = sum_i A(i, j) X(i, c): the gather ('T') form of the mat-vec on every column of X
__global__ __launch_bounds__(256) void k_spmm_t(int64_t n, const int64_t *__restrict__ Ap, const int64_t *__restrict__ Ai,
                                                const double *__restrict__ Ax, const double *__restrict__ X, int64_t ldx,
                                                double *__restrict__ Y, int64_t ldy)
{
    const int sub = threadIdx.x & 15;
    int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 4;
    const double *__restrict__ x = X + (int64_t)blockIdx.y * ldx;
    double *__restrict__ y = Y + (int64_t)blockIdx.y * ldy;
    for (; j < n; j += stride) {
        double acc = 0.0;
        for (int64_t p = Ap[j] + sub; p < Ap[j + 1]; p += 16) acc += Ax[p] * x[Ai[p]];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (sub == 0) y[j] = acc;
    }
}
// D (m x n, leading dimension ld, zeroed by the caller) := the CCS matrix
__global__ void k_dense_from_ccs(int64_t n, const int64_t *__restrict__ Ap, const int64_t *__restrict__ Ai,
                                 const double *__restrict__ Ax, double *__restrict__ D, int64_t ld)
{
    GS_LOOP(j, n) {
        for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) D[Ai[p] + j * ld] = Ax[p];
    }
}
// out := lower triangle of the dense p x p matrix K by columns (the value array of a dense lower CCS pattern)
__global__ void k_pack_lower(int64_t p, const double *__restrict__ K, int64_t ld, double *__restrict__ out)
{
    const int64_t j = blockIdx.y;
    const int64_t off = j * p - j * (j - 1) / 2;
    for (int64_t i = j + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p; i += (int64_t)gridDim.x * blockDim.x)
        out[off + (i - j)] = K[i + j * ld];
}
void launch_spmm_t(hipStream_t st, int64_t n, int64_t ncols, const int64_t *Ap, const int64_t *Ai, const double *Ax, const double *X,
                   int64_t ldx, double *Y, int64_t ldy)
{
    if (n > 0 && ncols > 0)
        hipLaunchKernelGGL(k_spmm_t, dim3(std::min<unsigned>(grid_for(n * 16), 4096u), (unsigned)ncols), dim3(256), 0, st, n, Ap, Ai, Ax, X, ldx, Y, ldy);
}
void launch_dense_from_ccs(hipStream_t st, int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax, double *D, int64_t ld)
{
    if (n > 0) hipLaunchKernelGGL(k_dense_from_ccs, dim3(grid_for(n)), dim3(256), 0, st, n, Ap, Ai, Ax, D, ld);
}
void launch_pack_lower(hipStream_t st, int64_t p, const double *K, int64_t ld, double *out)
{
    if (p > 0) hipLaunchKernelGGL(k_pack_lower, dim3((unsigned)std::min<int64_t>((p + 255) / 256, 64), (unsigned)p), dim3(256), 0, st, p, K, ld, out);
}

// ---- round 3: the interior-point iteration in a fifth of the launches --------------------------------------------------------
// A kernel of a few microseconds costs its launch and the drain before the next dependent one (4-5 us each on one stream): the
// ~85 BLAS-1-sized launches of an iteration were a fifth of its time.  The kernels below do the same arithmetic, element for
// element in the same order and with the same roundings as the sequences they replace (products that the separate kernels
// rounded before an addition are rounded here too: `fp contract(off)` and explicit fma where the old code had one), so an
// interior-point run is bit for bit the one of the unfused calls (tests/test_kkt_gpu.py::test_fused_iteration_is_bitwise).
// Reductions are ONE launch: the second stage of the fixed tree (256 partial results per reduction) is run by the consumer -- the
// host after the copy it makes anyway, or every workgroup of the next kernel for itself -- with the same association as k_reduce2.
// (A "last workgroup finishes" form with a ticket was measured first: its device-scope release/acquire fences write back and
// invalidate the L2 of every XCD, 25-48 us per launch against 5 for a second launch.)

// second stage of the fixed tree for one reduction, by one wavefront (the arithmetic of k_reduce2)
template <bool MAXNEG>
__device__ __forceinline__ double reduce2_wave(const double *__restrict__ part, int lane)
{
    double acc = MAXNEG ? -1.7976931348623157e308 : 0.0;
    for (int i = lane; i < RED_BLOCKS; i += 64) acc = MAXNEG ? fmax(acc, part[i]) : acc + part[i];
    return wave_red<MAXNEG>(acc);
}
// first stage only (k_reduce1 / k_reduce1_multi): the caller fetches the RED_BLOCKS partial results per reduction and runs the
// second stage on the host with the same association (kkt_api.cpp, host_reduce2) -- one launch per reduction call instead of two
void launch_reduce_multi_stage1(hipStream_t st, const MultiRed &mr, double *part)
{
    if (mr.count <= 0) return;
    hipLaunchKernelGGL(k_reduce1_multi, dim3(RED_BLOCKS, (unsigned)mr.count), dim3(256), 0, st, mr, part);
}
int reduce_blocks() { return RED_BLOCKS; }

// ---- KKT solve with misc.kkt_chol2's factor (misc.py:1489-1563, p = 0), the parts around the triangular solves:
//   pre :  x2_k := xs_k xin_k + G' (di .* (zin_k .* di))                                     [z := W^-1 z ; x += Gs' z]
//   post:  xout_k := xos_k x2_k ;  zout_k := zos_k (di .* (G x2_k) - zin_k .* di)            [z := Gs x - z]
// for one or two right-hand sides (grid.y), one launch each: mul + xmy + spmv + copy, copy + spmv + xmy (+ two scal) before.
template <int LPC>                                   // lanes per column of G: 16, 8 or 4 by the longest column (same bits, see k_kkt_post)
__global__ __launch_bounds__(256) void k_kkt_pre(int64_t n, const int64_t *__restrict__ Ap, const int64_t *__restrict__ Ai,
                                                 const double *__restrict__ Ax, const double *__restrict__ di, KktSides r,
                                                 double *__restrict__ x2, int64_t ld)
{
#pragma clang fp contract(off)
    const KktSide sd = r.s[blockIdx.y];
    const double *__restrict__ z = sd.zin;
    double *__restrict__ out = x2 + (int64_t)blockIdx.y * ld;
    const int sub = threadIdx.x & (LPC - 1);
    int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPC;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) / LPC;
    for (; j < n; j += stride) {
        double acc = 0.0;
        for (int64_t p = Ap[j] + sub; p < Ap[j + 1]; p += LPC) {
            const int64_t i = Ai[p];
            const double dd = di[i];
            const double zs = z[i] * dd;                      // z := W^-1 z
            const double t = dd * zs;                         // t := di .* z
            acc = __builtin_fma(Ax[p], t, acc);
        }
#pragma unroll
        for (int o = LPC / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (sub == 0) {
            const double x0 = sd.xs * sd.xin[j];
            out[j] = x0 + acc;
        }
    }
}
// LPR lanes per row of G: 16, or 4 where no row holds more than four entries (every lane then holds at most one product and the
// butterfly adds them in the order the 16-lane form does: the same bits)
template <int LPR>
__global__ __launch_bounds__(256) void k_kkt_post(int64_t ml, int64_t n, unsigned nb_rows, const int64_t *__restrict__ Ap,
                                                  const int64_t *__restrict__ Ai, const double *__restrict__ Ax,
                                                  const double *__restrict__ di, KktSides r, const double *__restrict__ x2, int64_t ld)
{
#pragma clang fp contract(off)
    const KktSide sd = r.s[blockIdx.y];
    const double *__restrict__ xk = x2 + (int64_t)blockIdx.y * ld;
    if (blockIdx.x >= nb_rows) {                                                 // the x part
        const int64_t stride = (int64_t)(gridDim.x - nb_rows) * 256;
        for (int64_t j = (int64_t)(blockIdx.x - nb_rows) * 256 + threadIdx.x; j < n; j += stride) sd.xout[j] = xk[j] * sd.xos;
        return;
    }
    const int sub = threadIdx.x & (LPR - 1);
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPR;
    const int64_t stride = ((int64_t)nb_rows * 256) / LPR;
    for (; i < ml; i += stride) {
        double acc = 0.0;
        for (int64_t p = Ap[i] + sub; p < Ap[i + 1]; p += LPR) acc = __builtin_fma(Ax[p], xk[Ai[p]], acc);
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (sub == 0) {
            const double t = 0.0 + acc;                       // t := G x  (beta = 0, alpha = 1 of the mat-vec)
            const double dd = di[i];
            const double zs = sd.zin[i] * dd;
            const double pr = dd * t;
            const double v = pr - zs;
            sd.zout[i] = v * sd.zos;
        }
    }
}
static inline unsigned groups16(int64_t rows, int lanes = 16)
{ return (unsigned)std::min<int64_t>(std::max<int64_t>((rows * lanes + 255) / 256, 1), 16384); }
// lanes that sum a row / column whose longest has `longest` entries (0 = not known): with at most L entries every lane of an L-lane
// group holds at most one product and the butterfly L/2 .. 1 adds them as the 16-lane one does
static inline int lanes_for(int64_t longest) { return longest > 0 && longest <= 4 ? 4 : (longest > 0 && longest <= 8 ? 8 : 16); }
void launch_kkt_pre(hipStream_t st, int64_t n, const int64_t *Gp, const int64_t *Gi, const double *Gx, const double *di, int nrhs,
                    const KktSides &r, double *x2, int64_t ld, int64_t max_col)
{
    if (n <= 0 || nrhs <= 0) return;
    const int lanes = lanes_for(max_col);
    const dim3 grid(groups16(n, lanes), (unsigned)nrhs);
    if (lanes == 4) hipLaunchKernelGGL(k_kkt_pre<4>, grid, dim3(256), 0, st, n, Gp, Gi, Gx, di, r, x2, ld);
    else if (lanes == 8) hipLaunchKernelGGL(k_kkt_pre<8>, grid, dim3(256), 0, st, n, Gp, Gi, Gx, di, r, x2, ld);
    else hipLaunchKernelGGL(k_kkt_pre<16>, grid, dim3(256), 0, st, n, Gp, Gi, Gx, di, r, x2, ld);
}
void launch_kkt_post(hipStream_t st, int64_t ml, int64_t n, const int64_t *tGp, const int64_t *tGi, const double *tGx, const double *di,
                     int nrhs, const KktSides &r, const double *x2, int64_t ld, int64_t max_row)
{
    if (nrhs <= 0 || (ml <= 0 && n <= 0)) return;
    const int lanes = lanes_for(max_row);
    const unsigned nbr = ml > 0 ? groups16(ml, lanes) : 0, nbx = n > 0 ? (unsigned)std::min<int64_t>((n + 255) / 256, 4096) : 0;
    const dim3 grid(nbr + nbx, (unsigned)nrhs);
    if (lanes == 4) hipLaunchKernelGGL(k_kkt_post<4>, grid, dim3(256), 0, st, ml, n, nbr, tGp, tGi, tGx, di, r, x2, ld);
    else if (lanes == 8) hipLaunchKernelGGL(k_kkt_post<8>, grid, dim3(256), 0, st, ml, n, nbr, tGp, tGi, tGx, di, r, x2, ld);
    else hipLaunchKernelGGL(k_kkt_post<16>, grid, dim3(256), 0, st, ml, n, nbr, tGp, tGi, tGx, di, r, x2, ld);
}

// ---- residuals of an iteration (coneprog.py:861-896, p = 0): hrx := -G'z ; rx := hrx - tau c ; hrz := G x + s ; rz := hrz - tau h
// in one launch (fill + two mat-vecs + axpy + two lincomb before): workgroups [0, nb_c) take the columns of G, the rest its rows.
template <int LPC, int LPR>
__global__ __launch_bounds__(256) void k_lp_residuals(int64_t ml, int64_t n, unsigned nb_c, const int64_t *__restrict__ Gp,
                                                      const int64_t *__restrict__ Gi, const double *__restrict__ Gx,
                                                      const int64_t *__restrict__ Tp, const int64_t *__restrict__ Ti,
                                                      const double *__restrict__ Tx, const double *__restrict__ x,
                                                      const double *__restrict__ z, const double *__restrict__ s,
                                                      const double *__restrict__ c, const double *__restrict__ h, double tau,
                                                      double *__restrict__ hrx, double *__restrict__ rx, double *__restrict__ hrz,
                                                      double *__restrict__ rz)
{
#pragma clang fp contract(off)
    if (blockIdx.x < nb_c) {
        const int sub = threadIdx.x & (LPC - 1);
        int64_t j = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPC;
        const int64_t stride = ((int64_t)nb_c * 256) / LPC;
        for (; j < n; j += stride) {
            double acc = 0.0;
            for (int64_t p = Gp[j] + sub; p < Gp[j + 1]; p += LPC) acc = __builtin_fma(Gx[p], z[Gi[p]], acc);
#pragma unroll
            for (int o = LPC / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if (sub == 0) {
                const double v = __builtin_fma(-1.0, acc, 0.0);                 // hrx := 0 ; hrx := -G'z + hrx
                hrx[j] = v;
                rx[j] = __builtin_fma(-tau, c[j], v);                           // rx := hrx - tau c
            }
        }
    } else {
        const int sr = threadIdx.x & (LPR - 1);
        int64_t i = ((int64_t)(blockIdx.x - nb_c) * 256 + threadIdx.x) / LPR;
        const int64_t stride = ((int64_t)(gridDim.x - nb_c) * 256) / LPR;
        for (; i < ml; i += stride) {
            double acc = 0.0;
            for (int64_t p = Tp[i] + sr; p < Tp[i + 1]; p += LPR) acc = __builtin_fma(Tx[p], x[Ti[p]], acc);
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if (sr == 0) {
                const double t = __builtin_fma(1.0, acc, 0.0);                  // hrz := G x
                const double v = __builtin_fma(1.0, s[i], t);                   // hrz += s
                hrz[i] = v;
                rz[i] = __builtin_fma(-tau, h[i], v);                           // rz := hrz - tau h
            }
        }
    }
}
void launch_lp_residuals(hipStream_t st, int64_t ml, int64_t n, const int64_t *Gp, const int64_t *Gi, const double *Gx, const int64_t *Tp,
                         const int64_t *Ti, const double *Tx, const double *x, const double *z, const double *s, const double *c,
                         const double *h, double tau, double *hrx, double *rx, double *hrz, double *rz, int64_t max_col, int64_t max_row)
{
    const int lc = lanes_for(max_col) == 16 ? 16 : 8, lr = lanes_for(max_row) == 4 ? 4 : 16;     // (four of the nine combinations)
    const unsigned nbc = n > 0 ? groups16(n, lc) : 0, nbr = ml > 0 ? groups16(ml, lr) : 0;
    if (nbc + nbr == 0) return;
#define KVX_RES(LC, LR) hipLaunchKernelGGL((k_lp_residuals<LC, LR>), dim3(nbc + nbr), dim3(256), 0, st, ml, n, nbc, Gp, Gi, Gx, Tp, Ti, Tx, x, z, s, c, h, tau, hrx, rx, hrz, rz)
    if (lc == 8 && lr == 4) KVX_RES(8, 4);
    else if (lc == 8) KVX_RES(8, 16);
    else if (lr == 4) KVX_RES(16, 4);
    else KVX_RES(16, 16);
#undef KVX_RES
}

// ---- second half of f6_no_ir in two launches (sixteen before; kvx_lp_second_half_dev) ------------------------------------
//  A: first stage of the three or four inner products (k_reduce1_multi);
//  B: every workgroup finishes them and forms dtau (k_lp_dtau's expression), then dx += dtau x1 [dy += dtau y1], the step_post
//     update of ds / dz / ws3 and the first stage of the step bounds max(-ds), max(-dz) of the new values; the host fetches the
//     partial maxima with dtau and finishes them.
__device__ __forceinline__ void lp_dtau_eval(const double *r, double dgi, double dtau0, double z1z1_host, int use_host, double *out)
{
    const double zz = use_host ? z1z1_host : r[3];
    out[0] = dgi * (dtau0 + r[0] + r[1] + r[2]) / (1.0 + zz);
    out[1] = zz;
}
__device__ __forceinline__ void lp_step_post_elem(int64_t i, double dtau, const double *__restrict__ z1, const double *__restrict__ lm,
                                                  double *__restrict__ ds, double *__restrict__ dz, double *__restrict__ ws3,
                                                  double &ss_out, double &zz_out)
{
    const double zz = dz[i] + dtau * z1[i];
    const double ss = ds[i] - zz;
    if (ws3) ws3[i] = ss * zz;
    const double l = lm[i];
    ss_out = ss / l;
    zz_out = zz / l;
    ds[i] = ss_out;
    dz[i] = zz_out;
}
__global__ __launch_bounds__(256) void k_lp_half_b(LpHalf a, double dgi, double dtau0, double z1z1_host, int use_host,
                                                   const double *__restrict__ part, double *__restrict__ part2, double *__restrict__ sc)
{
    __shared__ double sh[8];
    __shared__ double s_dtau;
    {   // second stage of the inner products of launch A and dtau, by every workgroup for itself (wave w: reduction w)
        const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const bool lv = !(wv == 1 && a.p <= 0) && !(wv == 3 && use_host);
        const double v = lv ? reduce2_wave<false>(part + (int64_t)wv * RED_BLOCKS, lane) : 0.0;
        if (lane == 0) sh[wv] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            double o[2];
            lp_dtau_eval(sh, dgi, dtau0, z1z1_host, use_host, o);
            s_dtau = o[0];
            if (blockIdx.x == 0) { sc[0] = o[0]; sc[1] = o[1]; }
        }
        __syncthreads();
    }
    const double dtau = s_dtau;
    if (blockIdx.x >= RED_BLOCKS) {                                             // dx += dtau x1, dy += dtau y1
        const int64_t t = (int64_t)(blockIdx.x - RED_BLOCKS) * 256 + threadIdx.x;
        const int64_t stride = (int64_t)(gridDim.x - RED_BLOCKS) * 256;
        for (int64_t i = t; i < a.n; i += stride) a.dxw[i] += dtau * a.x1[i];
        for (int64_t i = t; i < a.p; i += stride) a.dyw[i] += dtau * a.y1[i];
        return;
    }
    double as = -1.7976931348623157e308, az = -1.7976931348623157e308;
    const int64_t per = (a.ml + RED_BLOCKS - 1) / RED_BLOCKS;
    const int64_t b0 = per * blockIdx.x, b1 = min(a.ml, b0 + per);
    for (int64_t i = b0 + threadIdx.x; i < b1; i += 256) {
        double ss, zz;
        lp_step_post_elem(i, dtau, a.z1, a.lm, a.dsw, a.dzw, a.ws3, ss, zz);
        as = fmax(as, -ss);
        az = fmax(az, -zz);
    }
    as = wave_red<true>(as);
    az = wave_red<true>(az);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = as; sh[4 + (threadIdx.x >> 6)] = az; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r0 = sh[0], r1 = sh[4];
        for (int w = 1; w < 4; w++) { r0 = fmax(r0, sh[w]); r1 = fmax(r1, sh[4 + w]); }
        part2[blockIdx.x] = r0;                                                 // first stage of max(-ds), max(-dz): the host finishes
        part2[RED_BLOCKS + blockIdx.x] = r1;
    }
}
// part: 4 * RED_BLOCKS partial sums (launch A) ; part2: 2 * RED_BLOCKS partial maxima + sc[2] = (dtau, z1'z1) behind them
void launch_lp_second_half(hipStream_t st, const LpHalf &a, double dgi, double dtau0, double z1z1, double *part, double *part2)
{
    const int use_host = z1z1 < 0.0 ? 0 : 1;
    MultiRed mr;
    mr.count = use_host ? 3 : 4;
    for (int i = 0; i < 32; i++) { mr.kind[i] = 0; mr.n[i] = 0; mr.x[i] = nullptr; mr.y[i] = nullptr; }
    mr.n[0] = a.n; mr.x[0] = a.c; mr.y[0] = a.dx;
    mr.n[1] = a.p > 0 ? a.p : 0; mr.x[1] = a.b; mr.y[1] = a.dy;                  // p = 0: an empty sum (its partials are not read)
    mr.n[2] = a.ml; mr.x[2] = a.th; mr.y[2] = a.dz;
    mr.n[3] = a.ml; mr.x[3] = a.z1; mr.y[3] = a.z1;
    hipLaunchKernelGGL(k_reduce1_multi, dim3(RED_BLOCKS, (unsigned)mr.count), dim3(256), 0, st, mr, part);
    const int64_t nx = std::max(a.n, a.p);
    const unsigned nbx = (unsigned)std::min<int64_t>(std::max<int64_t>((nx + 255) / 256, 1), 2048);
    hipLaunchKernelGGL(k_lp_half_b, dim3(RED_BLOCKS + nbx), dim3(256), 0, st, a, dgi, dtau0, z1z1, use_host, part, part2,
                       part2 + 2 * RED_BLOCKS);
}

// ---- end of the iteration: the fused 'l'-cone update (k_lp_update) and x += step dx in one launch
__global__ void k_lp_update_x(int64_t ml, int64_t n, unsigned nb_ml, double step, double *__restrict__ ds, double *__restrict__ dz,
                              double *__restrict__ d, double *__restrict__ di, double *__restrict__ lm, double *__restrict__ s,
                              double *__restrict__ z, const double *__restrict__ dx, double *__restrict__ x)
{
    if (blockIdx.x >= nb_ml) {
        const int64_t stride = (int64_t)(gridDim.x - nb_ml) * blockDim.x;
        for (int64_t i = (int64_t)(blockIdx.x - nb_ml) * blockDim.x + threadIdx.x; i < n; i += stride) x[i] += step * dx[i];
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ml; i += (int64_t)nb_ml * blockDim.x)
        lp_update_elem(i, step, ds, dz, d, di, lm, s, z);
}
void launch_lp_update_x(hipStream_t st, int64_t ml, int64_t n, double step, double *ds, double *dz, double *d, double *di, double *lm,
                        double *s, double *z, const double *dx, double *x)
{
    const unsigned nbm = ml > 0 ? grid_for(ml) : 0, nbx = n > 0 ? grid_for(n) : 0;
    if (nbm + nbx == 0) return;
    hipLaunchKernelGGL(k_lp_update_x, dim3(nbm + nbx), dim3(256), 0, st, ml, n, nbm, step, ds, dz, d, di, lm, s, z, dx, x);
}

}  // namespace kvx
