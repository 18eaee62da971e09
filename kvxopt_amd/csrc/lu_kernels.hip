// HIP kernels of the sparse LU path (gfx950): multifrontal LU on a static structure with threshold partial
// pivoting inside each front's pivot block.  API role: klu_factor / klu_solve / klu_tsolve as called from
// src/C/klu.c:161, :187-198, :651-665.  Design notes in lu_symbolic.hpp.
//
// One workgroup per front and one launch per (level, size class).  A front is a dense m x m matrix: k pivot rows /
// columns first, then u = m - k update rows / columns.  Small fronts live in LDS for the whole assemble-factor-store
// sequence; larger ones in their own m x m region of the arena in HBM (their update matrix is read from there by
// the parent).  These are byte/latency-bound kernels -- no MFMA here (fronts of circuit / power-flow matrices are
// tens of rows; a blocked MFMA path for large LU fronts is future work).
#include "lu_device.hpp"

namespace kvx {

namespace {

constexpr int LU_NT_LDS = 256;
constexpr int LU_NT_BIG = 1024;
constexpr int LU_NT_SOLVE = 256;

// Right-looking elimination of the k pivot columns of the m x m front Fm (leading dimension ld).
// KLU's pivot rule (klu.h: Common.tol = 0.001, "partial pivoting with diagonal preference"): keep the diagonal
// entry when |d| >= tol * max|candidates|, else take the largest candidate -- candidates are the rows of the pivot
// block only.  A step fails when the best candidate is zero / not finite or smaller than stol * (largest entry of
// the whole front column): the host then merges this front into its parent and factors again.
template <int NT>
__device__ void lu_factor_front(double *Fm, const int ld, const int m, const int k, int32_t *ipiv, int32_t *fail_slot,
                                const double tol, const double stol, const int reuse, int *sh_i, double *sh_d)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int tx = tid & 63, ty = tid >> 6;
    constexpr int NW = NT / 64;
    bool failed = false;
    for (int j = 0; j < k; j++) {
        if (tid < 64) {
            double bmax = -1.0, amax = 0.0;
            int bidx = j;
            for (int i = j + lane; i < m; i += 64) {
                const double a = fabs(Fm[i + (int64_t)j * ld]);
                if (i < k && a > bmax) { bmax = a; bidx = i; }
                amax = fmax(amax, a);
            }
            for (int off = 32; off; off >>= 1) {
                const double ob = __shfl_xor(bmax, off), oa = __shfl_xor(amax, off);
                const int oi = __shfl_xor(bidx, off);
                if (ob > bmax || (ob == bmax && oi < bidx)) { bmax = ob; bidx = oi; }
                amax = fmax(amax, oa);
            }
            if (lane == 0) {
                const double diag = fabs(Fm[j + (int64_t)j * ld]);
                int r = (diag > 0.0 && diag >= tol * bmax) ? j : bidx;
                if (reuse) r = j + ipiv[j];
                double pv = Fm[r + (int64_t)j * ld];
                const double ap = fabs(pv);
                const bool bad = !(ap > 0.0) || !(ap <= 1.7e308) || ap < stol * amax;
                if (bad && !failed) { failed = true; *fail_slot = j + 1; }
                if (!(ap > 0.0) || !(ap <= 1.7e308)) pv = 1.0;          // keep going with finite numbers; the result is discarded
                if (!reuse) ipiv[j] = r - j;
                sh_i[0] = r;
                sh_d[0] = pv;
            }
        }
        __syncthreads();
        const int r = sh_i[0];
        const double pv = sh_d[0];
        if (r != j)
            for (int c = tid; c < m; c += NT) {
                const double a = Fm[j + (int64_t)c * ld], b = Fm[r + (int64_t)c * ld];
                Fm[j + (int64_t)c * ld] = b;
                Fm[r + (int64_t)c * ld] = a;
            }
        __syncthreads();
        for (int i = j + 1 + tid; i < m; i += NT) Fm[i + (int64_t)j * ld] /= pv;
        __syncthreads();
        for (int c = j + 1 + ty; c < m; c += NW) {
            const double ujc = Fm[j + (int64_t)c * ld];
            if (ujc != 0.0)
                for (int i = j + 1 + tx; i < m; i += 64) Fm[i + (int64_t)c * ld] -= Fm[i + (int64_t)j * ld] * ujc;
        }
        __syncthreads();
    }
}

template <bool LDS, int NT>
__global__ __launch_bounds__(NT) void k_lu_front(const LuDev d, const int32_t *__restrict__ list, const double *__restrict__ Ax,
                                                  const double tol, const double stol, const int reuse, const int lds_m)
{
    extern __shared__ double smem[];
    __shared__ int sh_i[2];
    __shared__ double sh_d[2];
    const int tid = threadIdx.x;
    const int f = list[blockIdx.x];
    const LuFrontD F = d.fr[f];
    const int m = F.m, k = F.k, u = m - k;
    double *Fm = LDS ? smem : d.arena + (F.upd_off - k - (int64_t)k * m);
    int32_t *sh_lp = (int32_t *)(LDS ? smem + (int64_t)lds_m * lds_m : smem);
    const int ld = m;
    if (tid == 0) d.fail[f] = 0;
    for (int64_t idx = tid; idx < (int64_t)m * m; idx += NT) Fm[idx] = 0.0;
    __syncthreads();
    // entries of A (scaled rows); every entry has its own slot
    for (int64_t e = tid; e < F.acnt; e += NT) {
        const int64_t src = d.a_src[F.aptr + e];
        Fm[d.a_dst[F.aptr + e]] += Ax[src] * d.rinv[d.ai32[src]];
    }
    __syncthreads();
    // extend-add the children's update matrices (parent pulls: no atomics, reproducible)
    for (int c = 0; c < F.nchild; c++) {
        const LuFrontD C = d.fr[d.children[F.childptr + c]];
        const int uc = C.m - C.k, ldc = C.upd_ld;
        const int32_t *__restrict__ relc = d.rel + C.rowptr + C.k;
        const double *__restrict__ Uc = d.arena + C.upd_off;
        const int tx = tid & 63, ty = tid >> 6;
        for (int jc = ty; jc < uc; jc += NT / 64) {
            const int64_t cj = (int64_t)relc[jc] * ld;
            for (int ic = tx; ic < uc; ic += 64) Fm[relc[ic] + cj] += Uc[ic + (int64_t)jc * ldc];
        }
        __syncthreads();
    }
    lu_factor_front<NT>(Fm, ld, m, k, d.ipiv + F.p0, d.fail + f, tol, stol, reuse, sh_i, sh_d);
    // which front row ended in each pivot slot
    for (int t = tid; t < k; t += NT) sh_lp[t] = t;
    __syncthreads();
    if (tid == 0)
        for (int j = 0; j < k; j++) {
            const int r = j + d.ipiv[F.p0 + j];
            if (r != j) { const int a = sh_lp[j]; sh_lp[j] = sh_lp[r]; sh_lp[r] = a; }
        }
    __syncthreads();
    for (int t = tid; t < k; t += NT) d.lperm[F.p0 + t] = sh_lp[t];
    // panels: L(:, 0:k) as is, U(0:k, :) transposed, both m x k column-major
    double *__restrict__ Lp = d.Lx + F.px, *__restrict__ Up = d.Ux + F.px;
    for (int64_t idx = tid; idx < (int64_t)m * k; idx += NT) Lp[idx] = Fm[idx];
    {
        const int tx = tid & 63, ty = tid >> 6;
        for (int t = ty; t < k; t += NT / 64)
            for (int c = tx; c < m; c += 64) Up[c + (int64_t)t * m] = Fm[t + (int64_t)c * ld];
    }
    if (LDS) {
        double *__restrict__ Uo = d.arena + F.upd_off;
        const int tx = tid & 63, ty = tid >> 6;
        for (int j = ty; j < u; j += NT / 64)
            for (int i = tx; i < u; i += 64) Uo[i + (int64_t)j * u] = Fm[(k + i) + (int64_t)(k + j) * ld];
    }
}

// Forward sweep of one front: f = [x(pivots); 0] + children's update vectors; (row permutation); solve with the
// k x k lower block; f(update) -= panel21 * y.  UNIT: L panel (unit diagonal, permutation); else U' (divide).
template <bool UNIT>
__global__ __launch_bounds__(LU_NT_SOLVE) void k_lu_fwd(const LuDev d, const int32_t *__restrict__ list, double *__restrict__ X,
                                                         const int64_t ldx, double *__restrict__ W, const int64_t wsize, const int max_m)
{
    extern __shared__ double smem[];
    constexpr int NT = LU_NT_SOLVE;
    const int tid = threadIdx.x;
    const int f = list[blockIdx.x];
    const LuFrontD F = d.fr[f];
    const int m = F.m, k = F.k;
    double *fv = smem, *g = smem + max_m;
    double *x = X + (int64_t)blockIdx.y * ldx;
    double *w = W + (int64_t)blockIdx.y * wsize;
    const double *__restrict__ panel = (UNIT ? d.Lx : d.Ux) + F.px;
    for (int t = tid; t < m; t += NT) fv[t] = t < k ? x[F.p0 + t] : 0.0;
    __syncthreads();
    for (int c = 0; c < F.nchild; c++) {
        const LuFrontD C = d.fr[d.children[F.childptr + c]];
        const int uc = C.m - C.k;
        const int32_t *__restrict__ relc = d.rel + C.rowptr + C.k;
        const double *__restrict__ wc = w + C.wx;
        for (int i = tid; i < uc; i += NT) fv[relc[i]] += wc[i];
        __syncthreads();
    }
    if (UNIT) {
        for (int t = tid; t < k; t += NT) g[t] = fv[d.lperm[F.p0 + t]];
        __syncthreads();
        for (int t = tid; t < k; t += NT) fv[t] = g[t];
        __syncthreads();
    }
    for (int t = 0; t < k; t++) {
        double y = fv[t];
        if (!UNIT) y /= panel[t + (int64_t)t * m];
        for (int i = t + 1 + tid; i < m; i += NT) fv[i] -= panel[i + (int64_t)t * m] * y;
        __syncthreads();
    }
    for (int t = tid; t < k; t += NT) x[F.p0 + t] = UNIT ? fv[t] : fv[t] / panel[t + (int64_t)t * m];
    double *__restrict__ ws = w + F.wx;
    for (int i = k + tid; i < m; i += NT) ws[i - k] = fv[i];
}

// Backward sweep of one front: y = x(pivots) - panel21' * x(update rows); back substitution with the transposed
// k x k lower block; UNIT (L'): the in-front permutation is applied to the result.
template <bool UNIT>
__global__ __launch_bounds__(LU_NT_SOLVE) void k_lu_bwd(const LuDev d, const int32_t *__restrict__ list, double *__restrict__ X,
                                                         const int64_t ldx, const int max_m)
{
    extern __shared__ double smem[];
    constexpr int NT = LU_NT_SOLVE, NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f = list[blockIdx.x];
    const LuFrontD F = d.fr[f];
    const int m = F.m, k = F.k;
    double *fv = smem, *g = smem + max_m;
    double *x = X + (int64_t)blockIdx.y * ldx;
    const double *__restrict__ panel = (UNIT ? d.Lx : d.Ux) + F.px;
    const int32_t *__restrict__ rows = d.rowidx + F.rowptr;
    for (int t = tid; t < m; t += NT) fv[t] = x[rows[t]];
    __syncthreads();
    for (int t = wave; t < k; t += NW) {
        double s = 0.0;
        for (int i = k + lane; i < m; i += 64) s += panel[i + (int64_t)t * m] * fv[i];
        for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) fv[t] -= s;
    }
    __syncthreads();
    for (int t = k - 1; t >= 0; t--) {
        double y = fv[t];
        if (!UNIT) y /= panel[t + (int64_t)t * m];
        if (tid == 0) g[t] = y;
        for (int s = tid; s < t; s += NT) fv[s] -= panel[t + (int64_t)s * m] * y;
        __syncthreads();
    }
    for (int t = tid; t < k; t += NT) x[F.p0 + (UNIT ? d.lperm[F.p0 + t] : t)] = g[t];
}

__global__ void k_lu_rowmax(const int64_t nnz, const int32_t *__restrict__ ai32, const double *__restrict__ Ax,
                            unsigned long long *__restrict__ rmax)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    const double a = fabs(Ax[p]);
    if (a > 0.0) atomicMax(rmax + ai32[p], (unsigned long long)__double_as_longlong(a));   // non-negative doubles order like integers
}
__global__ void k_lu_rinv(const int64_t n, const double *__restrict__ rmax, double *__restrict__ rinv)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double r = rmax[i];
    rinv[i] = (r > 0.0 && r <= 1.7e308) ? 1.0 / r : 1.0;
}
__global__ void k_lu_gather(const int64_t n, const int64_t *__restrict__ idx, const double *__restrict__ scale,
                            const double *__restrict__ B, const int64_t ldb, double *__restrict__ X, const int64_t ldx)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int64_t i = idx[p];
    X[p + blockIdx.y * ldx] = B[i + blockIdx.y * ldb] * (scale ? scale[i] : 1.0);
}
__global__ void k_lu_scatter(const int64_t n, const int64_t *__restrict__ idx, const double *__restrict__ scale,
                             const double *__restrict__ X, const int64_t ldx, double *__restrict__ B, const int64_t ldb)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int64_t i = idx[p];
    B[i + blockIdx.y * ldb] = X[p + blockIdx.y * ldx] * (scale ? scale[i] : 1.0);
}
__global__ void k_lu_udiag(const LuDev d, const int nfront, double *__restrict__ out)
{
    const int f = blockIdx.x;
    if (f >= nfront) return;
    const LuFrontD F = d.fr[f];
    for (int t = threadIdx.x; t < F.k; t += blockDim.x) out[F.p0 + t] = d.Ux[F.px + t + (int64_t)t * F.m];
}

}  // namespace

void launch_lu_fronts(const LuDev &d, const int32_t *list, int cnt, int lds_m, int max_k, const double *Ax, double tol,
                      double stol, int reuse, hipStream_t st)
{
    if (cnt <= 0) return;
    if (lds_m > 0) {
        const size_t sm = (size_t)lds_m * lds_m * sizeof(double) + (size_t)lds_m * sizeof(int32_t);
        hipLaunchKernelGGL((k_lu_front<true, LU_NT_LDS>), dim3(cnt), dim3(LU_NT_LDS), sm, st, d, list, Ax, tol, stol, reuse, lds_m);
    } else {
        const size_t sm = (size_t)max_k * sizeof(int32_t) + 16;
        hipLaunchKernelGGL((k_lu_front<false, LU_NT_BIG>), dim3(cnt), dim3(LU_NT_BIG), sm, st, d, list, Ax, tol, stol, reuse, 0);
    }
}

void launch_lu_fwd(const LuDev &d, const int32_t *list, int cnt, int max_m, int max_k, int unit, double *X, int64_t ldx, int nrhs,
                   double *W, int64_t wsize, hipStream_t st)
{
    if (cnt <= 0 || nrhs <= 0) return;
    const size_t sm = (size_t)(max_m + max_k) * sizeof(double);
    if (unit) hipLaunchKernelGGL(k_lu_fwd<true>, dim3(cnt, nrhs), dim3(LU_NT_SOLVE), sm, st, d, list, X, ldx, W, wsize, max_m);
    else hipLaunchKernelGGL(k_lu_fwd<false>, dim3(cnt, nrhs), dim3(LU_NT_SOLVE), sm, st, d, list, X, ldx, W, wsize, max_m);
}

void launch_lu_bwd(const LuDev &d, const int32_t *list, int cnt, int max_m, int max_k, int unit, double *X, int64_t ldx, int nrhs,
                   hipStream_t st)
{
    if (cnt <= 0 || nrhs <= 0) return;
    const size_t sm = (size_t)(max_m + max_k) * sizeof(double);
    if (unit) hipLaunchKernelGGL(k_lu_bwd<true>, dim3(cnt, nrhs), dim3(LU_NT_SOLVE), sm, st, d, list, X, ldx, max_m);
    else hipLaunchKernelGGL(k_lu_bwd<false>, dim3(cnt, nrhs), dim3(LU_NT_SOLVE), sm, st, d, list, X, ldx, max_m);
}

void launch_lu_rowmax(int64_t nnz, const int32_t *ai32, const double *Ax, double *rmax, hipStream_t st)
{
    if (nnz <= 0) return;
    hipLaunchKernelGGL(k_lu_rowmax, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, nnz, ai32, Ax, (unsigned long long *)rmax);
}
void launch_lu_rinv(int64_t n, const double *rmax, double *rinv, hipStream_t st)
{
    hipLaunchKernelGGL(k_lu_rinv, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, rmax, rinv);
}
void launch_lu_gather(int64_t n, int nrhs, const int64_t *idx, const double *scale, const double *B, int64_t ldb, double *X,
                      int64_t ldx, hipStream_t st)
{
    hipLaunchKernelGGL(k_lu_gather, dim3((unsigned)((n + 255) / 256), nrhs), dim3(256), 0, st, n, idx, scale, B, ldb, X, ldx);
}
void launch_lu_scatter(int64_t n, int nrhs, const int64_t *idx, const double *scale, const double *X, int64_t ldx, double *B,
                       int64_t ldb, hipStream_t st)
{
    hipLaunchKernelGGL(k_lu_scatter, dim3((unsigned)((n + 255) / 256), nrhs), dim3(256), 0, st, n, idx, scale, X, ldx, B, ldb);
}
void launch_lu_udiag(const LuDev &d, int nfront, double *out, hipStream_t st)
{
    if (nfront <= 0) return;
    hipLaunchKernelGGL(k_lu_udiag, dim3(nfront), dim3(64), 0, st, d, nfront, out);
}

}  // namespace kvx
