// Nesterov-Todd scaling for the semidefinite ('s') blocks of a cone program: the 's' parts of misc.compute_scaling /
// update_scaling (src/python/misc.py:354-419, 582-634), misc_solvers.scale / scale2 / sprod / sinv / sdot / max_step
// (src/C/misc_solvers.c:188-240, 343-397, 700-770, 845-882, 1029-1046, 1086-1160) and the storage helpers pack / pack2 /
// unpack / symm / trisc / triusc (misc_solvers.c:412-632, 887-988) -- SURVEY 8(f) item 4.
//
// Own design.  The reference walks the blocks one by one through LAPACK (potrf, gesvd, syevd/syevr) and BLAS-3 calls;
// here ONE 256-thread workgroup owns one block and one launch serves all blocks of a vector (two offset tables: squares
// and orders).  The dense kernels a block needs are written for that shape:
//   * Cholesky: right-looking, three barriers per column;
//   * SVD: one-sided Jacobi (Hestenes) on the columns, round-robin ordering -- in a round the m/2 column pairs are disjoint,
//     each wavefront takes pairs and its 64 lanes stride the rows; the three inner products of a pair are shuffle
//     reductions, no atomics, so a block's result does not depend on scheduling.  Singular values are delivered in
//     descending order as gesvd does (the order of lambda and of the columns of r, rti follows from it);
//   * symmetric eigenvalues (max_step): the same Jacobi sweep on x + c I, c = 1.5 |x|_F (positive definite, so the left
//     singular vectors are the eigenvectors and sigma - c the eigenvalues; absolute error ~ eps |x|_F like syevd's);
//   * the m x m products: one thread per output entry.
// Matrices live in HBM / L2 (orders of a few hundred at most; a block's working set is a few hundred KB), the kernels are
// latency-bound by their barriers -- this is API coverage, not a hot path.  The signs of singular / eigen-vectors are
// whatever the rotations give (gesvd's are as arbitrary): r r' and rti rti' -- the scaling itself -- do not depend on them.
#include "../../include/kvxhip.h"
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>

namespace kvx { void set_last_error(const std::string &s); }

namespace {

constexpr int NT = 256;

struct Blk { int64_t o2, o1; int m; };
__device__ __forceinline__ Blk get_blk(const int64_t *__restrict__ off2, const int64_t *__restrict__ off1, int k)
{
    return Blk{off2[k], off1[k], (int)(off1[k + 1] - off1[k])};
}

__device__ __forceinline__ double wave_sum_all(double x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// workgroup sum, every thread gets the total (fixed tree)
__device__ __forceinline__ double block_sum1(double x, double *sh)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    x = wave_sum_all(x);
    __syncthreads();
    if (lane == 0) sh[w] = x;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// In-place lower Cholesky of the m x m matrix A (column-major, ld = m); the strict upper triangle is set to zero.
// Returns (to every thread) -1 on success or the failing column.
__device__ int chol_lower(double *A, int m, int *sh_fail)
{
    const int tid = threadIdx.x;
    if (tid == 0) *sh_fail = -1;
    __syncthreads();
    for (int j = 0; j < m; j++) {
        const double d = A[j + (int64_t)j * m];
        if (!(d > 0.0)) {                                    // uniform: every thread read the same value
            if (tid == 0) *sh_fail = j;
            break;
        }
        const double l = sqrt(d);
        __syncthreads();
        for (int i = j + tid; i < m; i += NT) A[i + (int64_t)j * m] = (i == j) ? l : A[i + (int64_t)j * m] / l;
        __syncthreads();
        const int t = m - j - 1;
        for (int e = tid; e < t * t; e += NT) {
            const int i = j + 1 + e % t, c = j + 1 + e / t;
            if (c <= i) A[i + (int64_t)c * m] -= A[i + (int64_t)j * m] * A[c + (int64_t)j * m];
        }
        __syncthreads();
    }
    __syncthreads();
    for (int e = tid; e < m * m; e += NT) {
        const int i = e % m, c = e / m;
        if (i < c) A[e] = 0.0;
    }
    __syncthreads();
    return *sh_fail;
}

// C := op(A) op(B), all m x m column-major; ta / tb: 0 plain, 1 transposed.  C must not alias A or B.
__device__ void mm(double *C, const double *A, int ta, const double *B, int tb, int m)
{
    for (int e = threadIdx.x; e < m * m; e += NT) {
        const int i = e % m, j = e / m;
        double acc = 0.0;
        for (int p = 0; p < m; p++) {
            const double a = ta ? A[p + (int64_t)i * m] : A[i + (int64_t)p * m];
            const double b = tb ? B[j + (int64_t)p * m] : B[p + (int64_t)j * m];
            acc = __builtin_fma(a, b, acc);
        }
        C[e] = acc;
    }
    __syncthreads();
}

// X := L^{-T} X for a lower-triangular L (back substitution, thread c owns column c of X)
__device__ void trsm_lt(const double *L, double *X, int m)
{
    for (int c = threadIdx.x; c < m; c += NT) {
        double *x = X + (int64_t)c * m;
        for (int i = m - 1; i >= 0; i--) {
            double v = x[i];
            for (int p = i + 1; p < m; p++) v = __builtin_fma(-L[p + (int64_t)i * m], x[p], v);
            x[i] = v / L[i + (int64_t)i * m];
        }
    }
    __syncthreads();
}

// One-sided Jacobi: on return the columns of A are mutually orthogonal (A_out = A_in V); V (optional, m x m) accumulates
// the rotations starting from the identity.  sh: LDS int[2].
__device__ void jacobi_sweeps(double *A, double *V, int m, int *sh)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (V != nullptr) {
        for (int e = tid; e < m * m; e += NT) V[e] = (e % m == e / m) ? 1.0 : 0.0;
    }
    __syncthreads();
    if (m < 2) return;
    const int n = m + (m & 1);                               // players of the round-robin (a dummy when m is odd)
    const double tol = 2.220446049250313e-16 * sqrt((double)m);     // dgesvj's default: sqrt(m) eps
    for (int sweep = 0; sweep < 60; sweep++) {
        if (tid == 0) sh[0] = 0;
        __syncthreads();
        for (int r = 0; r < n - 1; r++) {
            for (int pk = wv; pk < n / 2; pk += NT / 64) {
                int p, q;
                if (pk == 0) { p = n - 1; q = r; }
                else { p = (r + pk) % (n - 1); q = (r - pk + (n - 1)) % (n - 1); }
                if (p > q) { const int t = p; p = q; q = t; }
                if (q >= m) continue;
                double *ap = A + (int64_t)p * m, *aq = A + (int64_t)q * m;
                double al = 0.0, be = 0.0, ga = 0.0;
                for (int i = lane; i < m; i += 64) {
                    const double x = ap[i], y = aq[i];
                    al = __builtin_fma(x, x, al);
                    be = __builtin_fma(y, y, be);
                    ga = __builtin_fma(x, y, ga);
                }
                al = wave_sum_all(al); be = wave_sum_all(be); ga = wave_sum_all(ga);
                if (fabs(ga) <= tol * sqrt(al * be) || ga == 0.0) continue;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = lane; i < m; i += 64) {
                    const double x = ap[i], y = aq[i];
                    ap[i] = c * x - s * y;
                    aq[i] = s * x + c * y;
                }
                if (V != nullptr) {
                    double *vp = V + (int64_t)p * m, *vq = V + (int64_t)q * m;
                    for (int i = lane; i < m; i += 64) {
                        const double x = vp[i], y = vq[i];
                        vp[i] = c * x - s * y;
                        vq[i] = s * x + c * y;
                    }
                }
                if (lane == 0) atomicAdd(&sh[0], 1);         // only a count: the order does not matter
            }
            __syncthreads();
        }
        const int rot = sh[0];
        __syncthreads();
        if (rot == 0) break;
    }
}

// After jacobi_sweeps: sig_out[rank] = |a_j|, U_out(:, rank) = a_j / |a_j|, V_out(:, rank) = v_j (optional; vt = 1 stores
// V' instead), ranks by descending (desc = 1) or ascending norm, ties by column index.  nrm: m doubles of scratch in HBM.
__device__ void jacobi_collect(const double *A, const double *V, int m, int desc, double *nrm, double *sig_out, double *U_out,
                               double *V_out, int vt)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int j = wv; j < m; j += NT / 64) {
        double a = 0.0;
        for (int i = lane; i < m; i += 64) { const double x = A[i + (int64_t)j * m]; a = __builtin_fma(x, x, a); }
        a = wave_sum_all(a);
        if (lane == 0) nrm[j] = sqrt(a);
    }
    __syncthreads();
    for (int j = wv; j < m; j += NT / 64) {
        const double sj = nrm[j];
        int rk = 0;
        for (int i = lane; i < m; i += 64) {
            const double si = nrm[i];
            const bool before = desc ? (si > sj || (si == sj && i < j)) : (si < sj || (si == sj && i < j));
            rk += before ? 1 : 0;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) rk += __shfl_xor(rk, o);
        if (lane == 0) sig_out[rk] = sj;
        const double inv = sj > 0.0 ? 1.0 / sj : 0.0;
        for (int i = lane; i < m; i += 64) {
            U_out[i + (int64_t)rk * m] = A[i + (int64_t)j * m] * inv;
            if (V_out != nullptr) {
                const double v = V[i + (int64_t)j * m];
                if (vt) V_out[rk + (int64_t)i * m] = v; else V_out[i + (int64_t)rk * m] = v;
            }
        }
    }
    __syncthreads();
}

// ---- compute_scaling (misc.py:354-419) ------------------------------------------------------------------------------
// work: 4 m^2 doubles per block at 4 * o2 (Ls, Lz, W = Lz' Ls, U).
__global__ __launch_bounds__(NT) void k_s_compute_scaling(const int64_t *__restrict__ off2, const int64_t *__restrict__ off1,
                                                          const double *__restrict__ s, const double *__restrict__ z,
                                                          double *__restrict__ r, double *__restrict__ rti,
                                                          double *__restrict__ lm, double *__restrict__ work, int *status)
{
    __shared__ int sh[4];
    const Blk b = get_blk(off2, off1, blockIdx.x);
    const int m = b.m;
    if (m == 0) return;
    const int64_t mm2 = (int64_t)m * m;
    double *Ls = work + 4 * b.o2, *Lz = Ls + mm2, *W = Lz + mm2, *U = W + mm2;
    for (int e = threadIdx.x; e < mm2; e += NT) { Ls[e] = s[b.o2 + e]; Lz[e] = z[b.o2 + e]; }
    __syncthreads();
    int f = chol_lower(Ls, m, &sh[2]);
    if (f < 0) f = chol_lower(Lz, m, &sh[2]);
    if (f >= 0) {                                            // not positive definite: lapack.potrf's ArithmeticError
        if (threadIdx.x == 0) atomicMin(status, f);
        return;
    }
    mm(W, Lz, 1, Ls, 0, m);                                  // W = Lz' Ls
    jacobi_sweeps(W, nullptr, m, sh);
    // U (left singular vectors) and lambda, descending; the norms go through r (free until the end)
    jacobi_collect(W, nullptr, m, 1, r + b.o2, lm + b.o1, U, nullptr, 0);
    // r = Lz^{-T} U diag(sqrt(lambda)),  rti = Lz U diag(1 / sqrt(lambda))
    mm(rti + b.o2, Lz, 0, U, 0, m);
    trsm_lt(Lz, U, m);
    for (int e = threadIdx.x; e < mm2; e += NT) {
        const double a = sqrt(lm[b.o1 + e / m]);
        r[b.o2 + e] = U[e] * a;
        rti[b.o2 + e] = rti[b.o2 + e] / a;
    }
}

// ---- update_scaling (misc.py:582-634): s_k, z_k hold Ls, Lz on entry, U and V' on return ----------------------------
__global__ __launch_bounds__(NT) void k_s_update_scaling(const int64_t *__restrict__ off2, const int64_t *__restrict__ off1,
                                                         double *__restrict__ s, double *__restrict__ z,
                                                         double *__restrict__ r, double *__restrict__ rti,
                                                         double *__restrict__ lm, double *__restrict__ work)
{
    __shared__ int sh[4];
    const Blk b = get_blk(off2, off1, blockIdx.x);
    const int m = b.m;
    if (m == 0) return;
    const int64_t mm2 = (int64_t)m * m;
    double *T = work + 4 * b.o2, *W = T + mm2, *V = W + mm2, *T2 = V + mm2;
    double *sk = s + b.o2, *zk = z + b.o2, *rk = r + b.o2, *tk = rti + b.o2;
    mm(T, rk, 0, sk, 0, m);                                  // r := r Ls
    mm(T2, tk, 0, zk, 0, m);                                 // rti := rti Lz
    mm(W, zk, 1, sk, 0, m);                                  // W = Lz' Ls
    for (int e = threadIdx.x; e < mm2; e += NT) { rk[e] = T[e]; tk[e] = T2[e]; }
    __syncthreads();
    jacobi_sweeps(W, V, m, sh);
    jacobi_collect(W, V, m, 1, T, lm + b.o1, sk, zk, 1);     // s_k = U, z_k = V'
    mm(T, rk, 0, zk, 1, m);                                  // r V
    mm(T2, tk, 0, sk, 0, m);                                 // rti U
    for (int e = threadIdx.x; e < mm2; e += NT) {
        const double a = 1.0 / sqrt(lm[b.o1 + e / m]);
        rk[e] = T[e] * a;
        tk[e] = T2[e] * a;
    }
}

// ---- scale (misc_solvers.c:188-240): x_k := R' X R (form 0) or R X R' (form 1), X = the symmetric matrix whose lower
// triangle is x_k; only the lower triangle of x_k is written (dsyr2k 'L').  work: m^2 per (block, column).
__global__ __launch_bounds__(NT) void k_s_scale(const int64_t *__restrict__ off2, const int64_t *__restrict__ off1,
                                                const double *__restrict__ R, double *__restrict__ x, int64_t ldx, int form,
                                                double *__restrict__ work, int64_t wstride)
{
    const Blk b = get_blk(off2, off1, blockIdx.x);
    const int m = b.m;
    if (m == 0) return;
    double *xk = x + b.o2 + (int64_t)blockIdx.y * ldx;
    double *T = work + b.o2 + (int64_t)blockIdx.y * wstride;
    const double *Rk = R + b.o2;
    // T = X R (form 0) or X R' (form 1)
    for (int e = threadIdx.x; e < m * m; e += NT) {
        const int i = e % m, j = e / m;
        double acc = 0.0;
        for (int p = 0; p < m; p++) {
            const double xv = (i >= p) ? xk[i + (int64_t)p * m] : xk[p + (int64_t)i * m];
            const double rv = form ? Rk[j + (int64_t)p * m] : Rk[p + (int64_t)j * m];
            acc = __builtin_fma(xv, rv, acc);
        }
        T[e] = acc;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < m * m; e += NT) {
        const int i = e % m, j = e / m;
        if (i < j) continue;
        double acc = 0.0;
        for (int p = 0; p < m; p++) {
            const double rv = form ? Rk[i + (int64_t)p * m] : Rk[p + (int64_t)i * m];
            acc = __builtin_fma(rv, T[p + (int64_t)j * m], acc);
        }
        xk[e] = acc;
    }
}

// ---- scale2 (misc_solvers.c:343-397): every entry (i, j) of x_k divided (inverse 0) or multiplied by sqrt(l_i) sqrt(l_j)
__global__ __launch_bounds__(NT) void k_s_scale2(const int64_t *__restrict__ off2, const int64_t *__restrict__ off1,
                                                 const double *__restrict__ lm, double *__restrict__ x, int inverse)
{
    const Blk b = get_blk(off2, off1, blockIdx.x);
    const int m = b.m;
    for (int e = threadIdx.x; e < m * m; e += NT) {
        const double c = sqrt(lm[b.o1 + e % m]) * sqrt(lm[b.o1 + e / m]);
        x[b.o2 + e] = inverse ? x[b.o2 + e] * c : x[b.o2 + e] / c;
    }
}

// ---- sprod with full 's' blocks (misc_solvers.c:700-742): x_k := (Y X + X Y) / 2 on the lower triangle, X and Y the
// symmetric matrices of the lower triangles; like the reference the upper triangle of y_k is overwritten by the mirror
// image of its lower triangle.  work: m^2 per block.
__global__ __launch_bounds__(NT) void k_s_sprod_full(const int64_t *__restrict__ off2, const int64_t *__restrict__ off1,
                                                     double *__restrict__ x, double *__restrict__ y, double *__restrict__ work)
{
    const Blk b = get_blk(off2, off1, blockIdx.x);
    const int m = b.m;
    double *xk = x + b.o2, *yk = y + b.o2, *A = work + b.o2;
    for (int e = threadIdx.x; e < m * m; e += NT) {
        const int i = e % m, j = e / m;
        A[e] = (i >= j) ? xk[e] : xk[j + (int64_t)i * m];
        if (i < j) yk[e] = yk[j + (int64_t)i * m];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < m * m; e += NT) {
        const int i = e % m, j = e / m;
        if (i < j) continue;
        double acc = 0.0;
        for (int p = 0; p < m; p++)
            acc += A[i + (int64_t)p * m] * yk[j + (int64_t)p * m] + yk[i + (int64_t)p * m] * A[j + (int64_t)p * m];
        xk[e] = 0.5 * acc;
    }
}

// ---- sprod / sinv with diagonal 's' blocks of y (misc_solvers.c:744-770, 845-882): the lower triangle of x_k is
// multiplied (op 0) or divided (op 1) entrywise by (y_i + y_j) / 2; y holds only the diagonals (orders table)
__global__ __launch_bounds__(NT) void k_s_prod_diag(const int64_t *__restrict__ off2, const int64_t *__restrict__ off1,
                                                    double *__restrict__ x, const double *__restrict__ yd, int op)
{
    const Blk b = get_blk(off2, off1, blockIdx.x);
    const int m = b.m;
    for (int e = threadIdx.x; e < m * m; e += NT) {
        const int i = e % m, j = e / m;
        if (i < j) continue;
        const double c = 0.5 * (yd[b.o1 + i] + yd[b.o1 + j]);
        x[b.o2 + e] = op ? x[b.o2 + e] / c : x[b.o2 + e] * c;
    }
}

// ---- sdot (misc_solvers.c:1029-1046): out[k] = sum_i x_ii y_ii + 2 sum_{i > j} x_ij y_ij
__global__ __launch_bounds__(NT) void k_s_dot(const int64_t *__restrict__ off2, const int64_t *__restrict__ off1,
                                              const double *__restrict__ x, const double *__restrict__ y, double *__restrict__ out)
{
    __shared__ double sh[4];
    const Blk b = get_blk(off2, off1, blockIdx.x);
    const int m = b.m;
    double acc = 0.0;
    for (int e = threadIdx.x; e < m * m; e += NT) {
        const int i = e % m, j = e / m;
        if (i < j) continue;
        const double v = x[b.o2 + e] * y[b.o2 + e];
        acc += (i == j) ? v : 2.0 * v;
    }
    acc = block_sum1(acc, sh);
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

// ---- max_step (misc_solvers.c:1086-1160): out[k] = -lambda_min(x_k); with vectors != 0 the eigenvalues (ascending) go to
// sigma and the eigenvectors replace x_k (dsyevd 'V'), otherwise x is left alone (dsyevr on a copy).  work: 3 m^2 + 2 m per block.
__global__ __launch_bounds__(NT) void k_s_max_step(const int64_t *__restrict__ off2, const int64_t *__restrict__ off1,
                                                   double *__restrict__ x, double *__restrict__ sigma, int vectors,
                                                   double *__restrict__ out, double *__restrict__ work)
{
    __shared__ int sh[4];
    __shared__ double shd[4];
    const Blk b = get_blk(off2, off1, blockIdx.x);
    const int m = b.m;
    if (m == 0) { if (threadIdx.x == 0) out[blockIdx.x] = -1.79769313486231570815e308; return; }
    const int64_t mm2 = (int64_t)m * m;
    double *B = work + 3 * b.o2 + 2 * b.o1, *Q = B + mm2, *sg = Q + mm2;      // sg: m values + m of scratch (the third m^2 is free for them when m >= 2)
    double *xk = x + b.o2;
    double fro = 0.0;
    for (int e = threadIdx.x; e < mm2; e += NT) {
        const int i = e % m, j = e / m;
        const double v = (i >= j) ? xk[e] : xk[j + (int64_t)i * m];
        B[e] = v;
        fro = __builtin_fma(v, v, fro);
    }
    fro = sqrt(block_sum1(fro, shd));
    const double c = fro > 0.0 ? 1.5 * fro : 1.0;                     // (the zero matrix: any positive shift)
    for (int i = threadIdx.x; i < m; i += NT) B[i + (int64_t)i * m] += c;
    __syncthreads();
    jacobi_sweeps(B, nullptr, m, sh);
    jacobi_collect(B, nullptr, m, 0, sg + m, sg, Q, nullptr, 0);
    if (threadIdx.x == 0) out[blockIdx.x] = -(sg[0] - c);
    if (vectors) {
        for (int i = threadIdx.x; i < m; i += NT) sigma[b.o1 + i] = sg[i] - c;
        for (int e = threadIdx.x; e < mm2; e += NT) xk[e] = Q[e];
    }
}

// ---- storage helpers --------------------------------------------------------------------------------------------------
// mode 0 symm  : upper triangle := mirror of the lower (misc_solvers.c:610-632)
// mode 1 trisc : upper := 0, strict lower *= 2 (:887-938)       mode 2 triusc: strict lower *= 0.5 (:940-988)
__global__ __launch_bounds__(NT) void k_s_tri(const int64_t *__restrict__ off2, const int64_t *__restrict__ off1,
                                              double *__restrict__ x, int mode)
{
    const Blk b = get_blk(off2, off1, blockIdx.x);
    const int m = b.m;
    for (int e = threadIdx.x; e < m * m; e += NT) {
        const int i = e % m, j = e / m;
        if (mode == 0) { if (i < j) x[b.o2 + e] = x[b.o2 + j + (int64_t)i * m]; }
        else if (mode == 1) { if (i < j) x[b.o2 + e] = 0.0; else if (i > j) x[b.o2 + e] *= 2.0; }
        else { if (i > j) x[b.o2 + e] *= 0.5; }
    }
}

// pack (dir 0, misc_solvers.c:412-468): yp[offp_k + packed index] = lower triangle of x_k by columns, off-diagonal entries
// times sqrt(2); unpack (dir 1, :552-608): the reverse into the lower triangle, off-diagonal entries divided by sqrt(2), the
// strict upper triangle of the unpacked block is left as it is.  offp: packed offsets (sum of m (m + 1) / 2).
__global__ __launch_bounds__(NT) void k_s_pack(const int64_t *__restrict__ off2, const int64_t *__restrict__ off1,
                                               const int64_t *__restrict__ offp, double *__restrict__ full,
                                               double *__restrict__ packed, int dir)
{
    const Blk b = get_blk(off2, off1, blockIdx.x);
    const int m = b.m;
    const int64_t op = offp[blockIdx.x];
    const double rt2 = 1.41421356237309504880;
    for (int e = threadIdx.x; e < m * m; e += NT) {
        const int i = e % m, j = e / m;
        if (i < j) continue;
        const int64_t pi = op + (int64_t)j * m - ((int64_t)j * (j - 1)) / 2 + (i - j);
        // the reference's own roundings: pack divides the diagonal by sqrt(2) before the common scaling, unpack multiplies
        // the off-diagonal entries by the rounded reciprocal
        if (dir == 0) packed[pi] = (i == j) ? (full[b.o2 + e] / rt2) * rt2 : full[b.o2 + e] * rt2;
        else if (dir == 2) packed[pi] = (i == j) ? full[b.o2 + e] : full[b.o2 + e] * rt2;      // pack2: the diagonal is copied
        else full[b.o2 + e] = (i == j) ? packed[pi] : packed[pi] * (1.0 / rt2);
    }
}

#define LAUNCH_OK(what)                                                                                                   \
    do {                                                                                                                  \
        hipError_t e_ = hipGetLastError();                                                                                \
        if (e_ != hipSuccess) {                                                                                           \
            kvx::set_last_error(std::string(what) + ": " + hipGetErrorString(e_));                                        \
            return KVX_EDEVICE;                                                                                           \
        }                                                                                                                 \
    } while (0)

}  // namespace

extern "C" {

int kvx_nts_compute_scaling_dev(int64_t ns, const int64_t *off2_dev, const int64_t *off1_dev, const double *s_dev,
                                const double *z_dev, double *r_dev, double *rti_dev, double *lmbda_dev, double *work_dev,
                                int *status_dev)
{
    if (ns < 0) return KVX_EINVAL;
    if (ns == 0) return KVX_OK;
    hipLaunchKernelGGL(k_s_compute_scaling, dim3((unsigned)ns), dim3(NT), 0, nullptr, off2_dev, off1_dev, s_dev, z_dev, r_dev,
                       rti_dev, lmbda_dev, work_dev, status_dev);
    LAUNCH_OK("k_s_compute_scaling");
    return KVX_OK;
}

int kvx_nts_update_scaling_dev(int64_t ns, const int64_t *off2_dev, const int64_t *off1_dev, double *s_dev, double *z_dev,
                               double *r_dev, double *rti_dev, double *lmbda_dev, double *work_dev)
{
    if (ns < 0) return KVX_EINVAL;
    if (ns == 0) return KVX_OK;
    hipLaunchKernelGGL(k_s_update_scaling, dim3((unsigned)ns), dim3(NT), 0, nullptr, off2_dev, off1_dev, s_dev, z_dev, r_dev,
                       rti_dev, lmbda_dev, work_dev);
    LAUNCH_OK("k_s_update_scaling");
    return KVX_OK;
}

int kvx_nts_scale_dev(int64_t ns, const int64_t *off2_dev, const int64_t *off1_dev, const double *R_dev, double *x_dev,
                      int64_t ldx, int64_t ncols, int form, double *work_dev, int64_t wstride)
{
    if (ns < 0 || ncols < 0 || ncols > 65535 || form < 0 || form > 1) return KVX_EINVAL;
    if (ns == 0 || ncols == 0) return KVX_OK;
    hipLaunchKernelGGL(k_s_scale, dim3((unsigned)ns, (unsigned)ncols), dim3(NT), 0, nullptr, off2_dev, off1_dev, R_dev, x_dev, ldx,
                       form, work_dev, wstride);
    LAUNCH_OK("k_s_scale");
    return KVX_OK;
}

int kvx_nts_scale2_dev(int64_t ns, const int64_t *off2_dev, const int64_t *off1_dev, const double *lmbda_dev, double *x_dev,
                       int inverse)
{
    if (ns < 0) return KVX_EINVAL;
    if (ns == 0) return KVX_OK;
    hipLaunchKernelGGL(k_s_scale2, dim3((unsigned)ns), dim3(NT), 0, nullptr, off2_dev, off1_dev, lmbda_dev, x_dev, inverse ? 1 : 0);
    LAUNCH_OK("k_s_scale2");
    return KVX_OK;
}

int kvx_nts_prod_dev(int64_t ns, const int64_t *off2_dev, const int64_t *off1_dev, double *x_dev, double *y_dev, int op,
                     double *work_dev)
{
    if (ns < 0 || op < 0 || op > 2) return KVX_EINVAL;
    if (ns == 0) return KVX_OK;
    if (op == 0) hipLaunchKernelGGL(k_s_sprod_full, dim3((unsigned)ns), dim3(NT), 0, nullptr, off2_dev, off1_dev, x_dev, y_dev, work_dev);
    else hipLaunchKernelGGL(k_s_prod_diag, dim3((unsigned)ns), dim3(NT), 0, nullptr, off2_dev, off1_dev, x_dev, y_dev, op - 1);
    LAUNCH_OK("k_s_prod");
    return KVX_OK;
}

int kvx_nts_dot_dev(int64_t ns, const int64_t *off2_dev, const int64_t *off1_dev, const double *x_dev, const double *y_dev,
                    double *out_dev)
{
    if (ns < 0) return KVX_EINVAL;
    if (ns == 0) return KVX_OK;
    hipLaunchKernelGGL(k_s_dot, dim3((unsigned)ns), dim3(NT), 0, nullptr, off2_dev, off1_dev, x_dev, y_dev, out_dev);
    LAUNCH_OK("k_s_dot");
    return KVX_OK;
}

int kvx_nts_max_step_dev(int64_t ns, const int64_t *off2_dev, const int64_t *off1_dev, double *x_dev, double *sigma_dev,
                         double *out_dev, double *work_dev)
{
    if (ns < 0) return KVX_EINVAL;
    if (ns == 0) return KVX_OK;
    hipLaunchKernelGGL(k_s_max_step, dim3((unsigned)ns), dim3(NT), 0, nullptr, off2_dev, off1_dev, x_dev, sigma_dev,
                       sigma_dev != nullptr ? 1 : 0, out_dev, work_dev);
    LAUNCH_OK("k_s_max_step");
    return KVX_OK;
}

int kvx_nts_tri_dev(int64_t ns, const int64_t *off2_dev, const int64_t *off1_dev, double *x_dev, int mode)
{
    if (ns < 0 || mode < 0 || mode > 2) return KVX_EINVAL;
    if (ns == 0) return KVX_OK;
    hipLaunchKernelGGL(k_s_tri, dim3((unsigned)ns), dim3(NT), 0, nullptr, off2_dev, off1_dev, x_dev, mode);
    LAUNCH_OK("k_s_tri");
    return KVX_OK;
}

int kvx_nts_pack_dev(int64_t ns, const int64_t *off2_dev, const int64_t *off1_dev, const int64_t *offp_dev, double *full_dev,
                     double *packed_dev, int dir)
{
    if (ns < 0 || dir < 0 || dir > 2) return KVX_EINVAL;
    if (ns == 0) return KVX_OK;
    hipLaunchKernelGGL(k_s_pack, dim3((unsigned)ns), dim3(NT), 0, nullptr, off2_dev, off1_dev, offp_dev, full_dev, packed_dev, dir);
    LAUNCH_OK("k_s_pack");
    return KVX_OK;
}

}  // extern "C"
