// Triangular solves with MANY right-hand sides (cholmod.solve with nrhs >= 64, cholmod.c:483-499; the column blocks of
// X = S^-1 A' in misc.kkt_chol2, misc.py:1483-1487; cholmod.spsolve).
//
// The kernels of kernels.hip / kernels_wave.hip / kernels_big.hip put one right-hand side on a workgroup: 64 right-hand sides
// read the factor 8 - 64 times and spend their time in per-column substitution chains (n = 1e6, 64 rhs: 18.5 ms for 15 Gflop
// and 2 GB of unavoidable traffic).  Here the right-hand sides are the fast dimension of the data:
//   * X is kept RHS-MAJOR in chunks of 64 right-hand sides: XT[chunk][row][64] -- a row of the block is one 512-byte line,
//     so the gathers of the extend-add and of the backward sweep are coalesced whatever the row indices are;
//   * every product with a panel is a small GEMM  out(16 rows x 64 rhs) -= L(16 x 4) * Y(4 x 64)  on the FP64 matrix pipe
//     (v_mfma_f64_16x16x4_f64): the A operand is loaded straight from the factor (one 8-byte load per lane feeds four MFMAs),
//     the B operand is the block of solved unknowns in LDS;
//   * the only sequential part is the 16 x 16 diagonal block of a small front: one row per register, one right-hand side per
//     lane, the 120 multipliers read as LDS broadcasts; big fronts use the inverted 64 x 64 diagonal blocks the factorisation
//     leaves behind (a GEMM as well);
//   * the children's update vectors are pulled per parent row through an inverse map (built once per analysis, api.cpp),
//     children in list order: no accumulator indexed by a run-time row number is needed, and the sum order is fixed.
// One wavefront per (small front, chunk); one 1024-thread workgroup per (big front, chunk).  The columns agree with the
// single-rhs kernels to rounding (the summation order of a GEMM differs from the substitution chains), not bit for bit.
#include "device.hpp"

#include <algorithm>

namespace kvx {
typedef double d4 __attribute__((ext_vector_type(4)));

namespace {

// [rows][64] LDS tile read as MFMA operands: lanes (lk, lr) = (l >> 4, l & 15) touch rows r + lk, columns c + lr.  Rows are 512
// bytes apart (the same banks); flipping bit 4 of the column in odd rows puts two consecutive rows on disjoint halves of the banks.
__device__ __forceinline__ int sw(int row, int col) { return row * 64 + (col ^ ((row & 1) << 4)); }

__device__ __forceinline__ d4 mfma(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

}  // namespace

// ---- layout changes ---------------------------------------------------------------------------------------------------------
// XT[chunk][iperm[j]][b] = B[j + (64 chunk + b) ldB]: driven by the CALLER's row index j, so the reads of B run down its columns
// (coalesced) and every write is one whole 512-byte row of XT wherever the permutation sends it.  iperm == nullptr: identity.
// Right-hand sides past nrhs are zero.
// diag / mode: optional row scaling on the way (LDL' views: the systems with D are the LL' sweeps with diag(Lc) applied to the
// right-hand side or the result, indexed by the row's position in the factor): mode 0 multiplies by diag, 1 divides.
__global__ __launch_bounds__(256) void k_wide_gather(const int32_t *__restrict__ iperm, int64_t n, int nrhs,
                                                     const double *__restrict__ B, int64_t ldB, double *__restrict__ XT,
                                                     const double *__restrict__ diag, int mode)
{
    __shared__ double tile[64][65];
    __shared__ int dst[64];
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    const int c = blockIdx.y;
    const int nv = min(64, nrhs - 64 * c);
    const int i = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t row = row0 + i;
    const bool rin = row < n;
    if (g == 0) dst[i] = rin ? (iperm ? iperm[row] : (int)row) : 0;
    for (int b = g; b < 64; b += 4) {
        const bool ok = rin && b < nv;
        const double v = B[ok ? row + (int64_t)(64 * c + b) * ldB : 0];
        tile[i][b] = ok ? v : 0.0;
    }
    __syncthreads();
    double *out = XT + (int64_t)c * n * 64;
    for (int r = g; r < 64; r += 4)
        if (row0 + r < n) {
            double v = tile[r][i];
            if (diag) { const double dg = diag[dst[r]]; v = mode ? v / dg : v * dg; }
            out[(int64_t)dst[r] * 64 + i] = v;
        }
}

// B[j + (64 chunk + b) ldB] = XT[chunk][iperm[j]][b]
__global__ __launch_bounds__(256) void k_wide_scatter(const int32_t *__restrict__ iperm, int64_t n, int nrhs,
                                                      const double *__restrict__ XT, double *__restrict__ B, int64_t ldB,
                                                      const double *__restrict__ diag, int mode)
{
    __shared__ double tile[64][65];
    __shared__ int src[64];
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    const int c = blockIdx.y;
    const int nv = min(64, nrhs - 64 * c);
    const int i = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t row = row0 + i;
    if (g == 0) src[i] = row < n ? (iperm ? iperm[row] : (int)row) : 0;
    __syncthreads();
    const double *in = XT + (int64_t)c * n * 64;
    for (int r = g; r < 64; r += 4) {
        double v = (row0 + r < n) ? in[(int64_t)src[r] * 64 + i] : 0.0;
        if (diag && row0 + r < n) { const double dg = diag[src[r]]; v = mode ? v / dg : v * dg; }
        tile[r][i] = v;
    }
    __syncthreads();
    if (row >= n) return;
    for (int b = g; b < nv; b += 4) B[row + (int64_t)(64 * c + b) * ldB] = tile[i][b];
}

// ---- small fronts (k <= KMAX <= 64, m <= 128): one wavefront per (front, chunk) ----------------------------------------------
// Everything stays in the accumulator layout of the MFMA: a block of 16 rows x 64 right-hand sides is four d4 registers per lane,
// D[q1][q] = (row 16 b + lk + 4 q, rhs 16 q1 + lr).  That layout IS the B operand of the four k-steps that multiply with those 16
// rows (B[kk = lk][n = lr] of step s is D[q1][s]), so the solved pivot blocks of a front never leave the registers: no LDS image
// of y, no layout change.  The 16 x 16 substitution runs in the same layout: the pivot row's values travel from their lane group
// to the other three through the LDS crossbar (ds_bpermute), the multipliers are four LDS reads per step from the 2 KB image
// of the diagonal block.  The children's update rows are pulled per lane group through the inverse map.

// rows r0 + lk + 4 q of the front's right-hand side: x for the pivot rows + the children's update rows (forward sweep)
__device__ __forceinline__ void wide_assemble(d4 (&D)[4], int r0, int k, int m, int first, int lk, int lr, const double *__restrict__ xt,
                                              const double *__restrict__ wc, const int32_t *__restrict__ ip,
                                              const int32_t *__restrict__ inv_src)
{
    int e0[4], e1[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int r = r0 + lk + 4 * q;
        const int rc = min(r, m - 1);
        e0[q] = ip[rc];
        e1[q] = r < m ? ip[rc + 1] : e0[q];
#pragma unroll
        for (int q1 = 0; q1 < 4; q1++) D[q1][q] = kvx_ld0(xt, (int64_t)(first + r) * 64 + 16 * q1 + lr, r < k);
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
        for (int e = e0[q]; e < e1[q]; e++) {
            const double *src = wc + (int64_t)inv_src[e] * 64 + lr;
#pragma unroll
            for (int q1 = 0; q1 < 4; q1++) D[q1][q] += src[16 * q1];
        }
}

// image of the 16 x 16 diagonal block at (r0, r0) (column-major, zero above the diagonal and past the pivots) + reciprocal diagonal
__device__ __forceinline__ void wide_diag_image(const double *__restrict__ P, int r0, int k, int m, int l, double *Ld, double *Ldi)
{
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int idx = l + 64 * t, ii = idx & 15, jj = idx >> 4;
        Ld[idx] = kvx_ld0(P, (r0 + ii) + (int64_t)(r0 + jj) * m, r0 + ii < m && r0 + jj < k && jj <= ii);
    }
    if (l < 16) {
        const double dg = kvx_ld0(P, (r0 + l) + (int64_t)(r0 + l) * m, r0 + l < k);
        Ldi[l] = 1.0 / (r0 + l < k ? dg : 1.0);
    }
}

template <int KMAX>
__global__ __launch_bounds__(64, KMAX <= 32 ? 3 : 2) void k_wide_fwd_small(DevSym ds, const int32_t *__restrict__ list, const double *__restrict__ Lx,
                                                       double *__restrict__ XT, int64_t n, const double *__restrict__ Wch,
                                                       double *__restrict__ Wout, int64_t wstride,
                                                       const int32_t *__restrict__ inv_ptr, const int32_t *__restrict__ inv_src)
{
    constexpr int KB = KMAX / 16;
    __shared__ double Ld[16 * 16];
    __shared__ double Ldi[16];
    unsigned fi, ch;
    kvx_front_rhs(fi, ch);
    const FrontDesc fd = ds.fd[list[fi]];
    const int k = fd.k, m = fd.m, first = fd.first;
    const int l = threadIdx.x, lr = l & 15, lk = l >> 4;
    const double *P = Lx + fd.px;
    double *xt = XT + (int64_t)ch * n * 64;
    const double *wc = Wch + (int64_t)ch * wstride * 64;
    double *wo = Wout + ((int64_t)ch * wstride + fd.wx) * 64;
    const int32_t *ip = inv_ptr + fd.rowptr;
    d4 Y[KB][4];
#pragma unroll
    for (int b = 0; b < KB; b++)
#pragma unroll
        for (int q1 = 0; q1 < 4; q1++) Y[b][q1] = (d4){0.0, 0.0, 0.0, 0.0};
    // ---- the pivot blocks
#pragma unroll
    for (int b = 0; b < KB; b++) {
        const int r0 = 16 * b;
        if (r0 < k) {                              // wave-uniform
            wide_diag_image(P, r0, k, m, l, Ld, Ldi);
            d4 D[4];
            wide_assemble(D, r0, k, m, first, lk, lr, xt, wc, ip, inv_src);
            const bool rin = r0 + lr < m;
#pragma unroll
            for (int pb = 0; pb < b; pb++) {
                double av[4];
#pragma unroll
                for (int s = 0; s < 4; s++) av[s] = -kvx_ld0(P, (r0 + lr) + (int64_t)(16 * pb + 4 * s + lk) * m, rin);
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int q1 = 0; q1 < 4; q1++) D[q1] = mfma(av[s], Y[pb][q1][s], D[q1]);
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (r0 + i < k) {                  // wave-uniform
                    const int gi = i & 3, qi = i >> 2;
                    const double ri = Ldi[i];
                    double mult[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) mult[q] = (lk + 4 * q > i) ? Ld[i * 16 + lk + 4 * q] : 0.0;
#pragma unroll
                    for (int q1 = 0; q1 < 4; q1++) {
                        const double y = __shfl(D[q1][qi] * ri, gi * 16 + lr);
                        D[q1][qi] = (lk == gi) ? y : D[q1][qi];
#pragma unroll
                        for (int q = 0; q < 4; q++) D[q1][q] = __builtin_fma(-mult[q], y, D[q1][q]);
                    }
                }
            }
            __syncthreads();                       // (the image is rewritten by the next block)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int r = r0 + lk + 4 * q;
#pragma unroll
                for (int q1 = 0; q1 < 4; q1++) {
                    if (r < k) xt[(int64_t)(first + r) * 64 + 16 * q1 + lr] = D[q1][q];
                    else if (r < m) wo[(int64_t)(r - k) * 64 + 16 * q1 + lr] = D[q1][q];
                    Y[b][q1][q] = r < k ? D[q1][q] : 0.0;
                }
            }
        }
    }
    // ---- the rows past the last pivot block: update rows only
    for (int r0 = 16 * ((k + 15) / 16); r0 < m; r0 += 16) {
        d4 D[4];
        wide_assemble(D, r0, k, m, first, lk, lr, xt, wc, ip, inv_src);
        const bool rin = r0 + lr < m;
#pragma unroll
        for (int pb = 0; pb < KB; pb++) {
            if (16 * pb < k) {                     // wave-uniform
                double av[4];
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const int j = 16 * pb + 4 * s + lk;
                    av[s] = -kvx_ld0(P, (r0 + lr) + (int64_t)j * m, rin && j < k);
                }
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int q1 = 0; q1 < 4; q1++) D[q1] = mfma(av[s], Y[pb][q1][s], D[q1]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r = r0 + lk + 4 * q;
            if (r < m) {
#pragma unroll
                for (int q1 = 0; q1 < 4; q1++) wo[(int64_t)(r - k) * 64 + 16 * q1 + lr] = D[q1][q];
            }
        }
    }
}

// Backward: pivot blocks of 16 from the last.  t = y - L(rows below the block, block)' x: the front's own later pivot blocks from
// registers, the rows past the pivots gathered from XT; then the 16 x 16 back substitution.
template <int KMAX>
__global__ __launch_bounds__(64) void k_wide_bwd_small(DevSym ds, const int32_t *__restrict__ list, const double *__restrict__ Lx,
                                                       double *__restrict__ XT, int64_t n)
{
    constexpr int KB = KMAX / 16;
    __shared__ double Ld[16 * 16];
    __shared__ double Ldi[16];
    unsigned fi, ch;
    kvx_front_rhs(fi, ch);
    const FrontDesc fd = ds.fd[list[fi]];
    const int k = fd.k, m = fd.m, first = fd.first;
    const int l = threadIdx.x, lr = l & 15, lk = l >> 4;
    const double *P = Lx + fd.px;
    double *xt = XT + (int64_t)ch * n * 64;
    const int32_t *rows = ds.rowidx + fd.rowptr;
    d4 X[KB][4];
#pragma unroll
    for (int b = 0; b < KB; b++)
#pragma unroll
        for (int q1 = 0; q1 < 4; q1++) X[b][q1] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int b = KB - 1; b >= 0; b--) {
        const int p0 = 16 * b;
        if (p0 < k) {                              // wave-uniform
            wide_diag_image(P, p0, k, m, l, Ld, Ldi);
            d4 D[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int p = p0 + lk + 4 * q;
#pragma unroll
                for (int q1 = 0; q1 < 4; q1++) D[q1][q] = kvx_ld0(xt, (int64_t)(first + p) * 64 + 16 * q1 + lr, p < k);
            }
            const bool pin = p0 + lr < k;
            const double *Pc = P + (int64_t)(pin ? p0 + lr : 0) * m;
            // later pivot blocks of this front
#pragma unroll
            for (int lb = b + 1; lb < KB; lb++) {
                if (16 * lb < k) {                 // wave-uniform
                    double av[4];
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        const int r = 16 * lb + 4 * s + lk;
                        av[s] = -kvx_ld0(Pc, r, pin && r < k);
                    }
#pragma unroll
                    for (int s = 0; s < 4; s++)
#pragma unroll
                        for (int q1 = 0; q1 < 4; q1++) D[q1] = mfma(av[s], X[lb][q1][s], D[q1]);
                }
            }
            // rows past the pivots (the block that holds row k may start inside the pivots: masked)
            double avn[4];
            int grn[4];
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int r = 16 * (k / 16) + 4 * s + lk;
                avn[s] = -kvx_ld0(Pc, r, pin && r >= k && r < m);
                grn[s] = rows[min(r, m - 1)];
            }
            for (int rr0 = 16 * (k / 16); rr0 < m; rr0 += 16) {
                double av[4], xg[4][4];
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    av[s] = avn[s];
#pragma unroll
                    for (int q1 = 0; q1 < 4; q1++) xg[s][q1] = xt[(int64_t)grn[s] * 64 + 16 * q1 + lr];
                }
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const int r = rr0 + 16 + 4 * s + lk;
                    avn[s] = -kvx_ld0(Pc, r, pin && r >= k && r < m);
                    grn[s] = rows[min(r, m - 1)];
                }
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const int r = rr0 + 4 * s + lk;
                    const bool rin = r >= k && r < m;
#pragma unroll
                    for (int q1 = 0; q1 < 4; q1++) D[q1] = mfma(av[s], rin ? xg[s][q1] : 0.0, D[q1]);
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 15; i >= 0; i--) {
                if (p0 + i < k) {                  // wave-uniform
                    const int gi = i & 3, qi = i >> 2;
                    const double ri = Ldi[i];
                    double mult[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) mult[q] = (lk + 4 * q < i) ? Ld[(lk + 4 * q) * 16 + i] : 0.0;
#pragma unroll
                    for (int q1 = 0; q1 < 4; q1++) {
                        const double x = __shfl(D[q1][qi] * ri, gi * 16 + lr);
                        D[q1][qi] = (lk == gi) ? x : D[q1][qi];
#pragma unroll
                        for (int q = 0; q < 4; q++) D[q1][q] = __builtin_fma(-mult[q], x, D[q1][q]);
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int p = p0 + lk + 4 * q;
#pragma unroll
                for (int q1 = 0; q1 < 4; q1++) {
                    if (p < k) xt[(int64_t)(first + p) * 64 + 16 * q1 + lr] = D[q1][q];
                    X[b][q1][q] = p < k ? D[q1][q] : 0.0;
                }
            }
        }
    }
}

// ---- big fronts: several 256-thread workgroups per (front, chunk), one launch per 64-column block ---------------------------
// A big front is worked on by many workgroups; the 64-column blocks of its pivot rows are the sequential dimension.  The diagonal
// solve of a block (a GEMM with the inverted 64 x 64 block the factorisation left behind) is done by the workgroup that has just
// finished the rows of that block, at the END of the launch that made them final -- so the next launch finds y (x) of the block
// in XT, every workgroup stages it into LDS and applies it to its own 64 rows.  No workgroup reads rows that another one writes
// in the same launch.

// y = Linv_blk * w (TRANS = false) or x = Linv_blk' * t (TRANS = true) for the 64-row block held in Bf (rows past nb zero); wave w
// computes rows 16 w .. 16 w + 15 for all 64 right-hand sides.  The result replaces Bf and goes to XT rows `dst`.
// the 16 A operands of a wave (loaded early by the caller: their latency hides behind the caller's own work)
template <bool TRANS>
__device__ __forceinline__ void wide_diag_load(const double *__restrict__ Yi, int nb, int w, int lr, int lk, double (&av)[16])
{
    const int ri = 16 * w + lr;
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const int c = 4 * s + lk;
        av[s] = TRANS ? kvx_ld0(Yi, c + ri * 64, c < nb && ri < nb && c >= ri) : kvx_ld0(Yi, ri + c * 64, ri < nb && c <= ri);
    }
}
template <bool TRANS>
__device__ __forceinline__ void wide_diag_solve(const double (&av)[16], int nb, double *Bf, double *dst, int w, int lr, int lk)
{
    d4 D[4];
#pragma unroll
    for (int q1 = 0; q1 < 4; q1++) D[q1] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 16; s++) {
        if (TRANS ? (4 * s + 3 >= 16 * w) : (4 * s < 16 * (w + 1))) {       // wave-uniform: the triangle of the block
            const int c = 4 * s + lk;
#pragma unroll
            for (int q1 = 0; q1 < 4; q1++) D[q1] = mfma(av[s], Bf[sw(c, 16 * q1 + lr)], D[q1]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int i = 16 * w + lk + 4 * q;
#pragma unroll
        for (int q1 = 0; q1 < 4; q1++) {
            Bf[sw(i, 16 * q1 + lr)] = D[q1][q];
            if (i < nb) dst[(int64_t)i * 64 + 16 * q1 + lr] = D[q1][q];
        }
    }
}

// Forward, first launch of a level: extend-add of all rows (workgroup x = 64 rows, a wave = 16 of them), then y of the first block.
__global__ __launch_bounds__(256) void k_wide_fwd_big_asm(DevSym ds, const int32_t *__restrict__ list, const double *__restrict__ Linv,
                                                          double *__restrict__ XT, int64_t n, const double *__restrict__ Wch,
                                                          double *__restrict__ Wout, int64_t wstride,
                                                          const int32_t *__restrict__ inv_ptr, const int32_t *__restrict__ inv_src)
{
    __shared__ double Bf[64 * 64];
    unsigned bx, by, ch;
    kvx_part_front_rhs(bx, by, ch);
    const FrontDesc fd = ds.fd[list[by]];
    const int k = fd.k, m = fd.m, first = fd.first;
    if ((int)bx * 64 >= m) return;
    const int tid = threadIdx.x, l = tid & 63, lr = l & 15, lk = l >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *xt = XT + (int64_t)ch * n * 64;
    const double *wc = Wch + (int64_t)ch * wstride * 64;
    double *wo = Wout + ((int64_t)ch * wstride + fd.wx) * 64;
    const int32_t *ip = inv_ptr + fd.rowptr;
#pragma unroll 4
    for (int i = 0; i < 16; i++) {
        const int r = (int)bx * 64 + 16 * w + i;
        if (r < m) {                               // wave-uniform
            const int e0 = ip[r], e1 = ip[r + 1];
            if (r >= k || e1 > e0) {
                double val = r < k ? xt[(int64_t)(first + r) * 64 + l] : 0.0;
                for (int e = e0; e < e1; e++) val += wc[(int64_t)inv_src[e] * 64 + l];
                if (r < k) xt[(int64_t)(first + r) * 64 + l] = val;
                else wo[(int64_t)(r - k) * 64 + l] = val;
            }
        }
    }
    if (bx != 0) return;
    const int nb = min(64, k);
    double dv[16];
    wide_diag_load<false>(Linv + fd.linv, nb, w, lr, lk, dv);
    __syncthreads();
    for (int i = w; i < 64; i += 4) Bf[sw(i, l)] = kvx_ld0(xt, (int64_t)(first + i) * 64 + l, i < nb);
    __syncthreads();
    wide_diag_solve<false>(dv, nb, Bf, xt + (int64_t)first * 64, w, lr, lk);
}

// Forward, block jb: every workgroup stages y_blk and takes it off its 64 rows below the block; workgroup 0's rows are the next
// block, which it then solves.
__global__ __launch_bounds__(256) void k_wide_fwd_big_step(DevSym ds, const int32_t *__restrict__ list, int jb,
                                                           const double *__restrict__ Lx, const double *__restrict__ Linv,
                                                           double *__restrict__ XT, int64_t n, double *__restrict__ Wout, int64_t wstride)
{
    __shared__ double Bf[64 * 64];
    unsigned bx, by, ch;
    kvx_part_front_rhs(bx, by, ch);
    const FrontDesc fd = ds.fd[list[by]];
    const int k = fd.k, m = fd.m, first = fd.first;
    if (jb >= k) return;
    const int nb = min(64, k - jb);
    const int rbeg = jb + nb + (int)bx * 64;
    if (rbeg >= m) return;
    const int tid = threadIdx.x, l = tid & 63, lr = l & 15, lk = l >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double *P = Lx + fd.px;
    double *xt = XT + (int64_t)ch * n * 64;
    double *wo = Wout + ((int64_t)ch * wstride + fd.wx) * 64;
    const bool next = bx == 0 && jb + nb < k;       // this workgroup's 64 rows are the next pivot block: it solves it at the end
    const int jn = jb + nb, nb2 = min(64, k - jn);
    double dv[16];
    if (next) wide_diag_load<false>(Linv + fd.linv + (int64_t)(jn / 64) * 4096, nb2, w, lr, lk, dv);
    for (int i = w; i < 64; i += 4) Bf[sw(i, l)] = kvx_ld0(xt, (int64_t)(first + jb + i) * 64 + l, i < nb);
    const int r0 = rbeg + 16 * w;
    d4 D[4];
    double *ptr[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int rs = min(r0 + lk + 4 * q, m - 1);
        ptr[q] = (rs < k ? xt + (int64_t)(first + rs) * 64 : wo + (int64_t)(rs - k) * 64) + lr;
#pragma unroll
        for (int q1 = 0; q1 < 4; q1++) D[q1][q] = ptr[q][16 * q1];
    }
    const bool ain = r0 + lr < m;
    double av[16];
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const int j = 4 * s + lk;
        av[s] = -kvx_ld0(P, (r0 + lr) + (int64_t)(jb + j) * m, ain && j < nb);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const int j = 4 * s + lk;
#pragma unroll
        for (int q1 = 0; q1 < 4; q1++) D[q1] = mfma(av[s], Bf[sw(j, 16 * q1 + lr)], D[q1]);
    }
    if (!next) {
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (r0 + lk + 4 * q < m) {
#pragma unroll
                for (int q1 = 0; q1 < 4; q1++) ptr[q][16 * q1] = D[q1][q];
            }
        return;
    }
    __syncthreads();                                // every wave is done with y_blk
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int i = 16 * w + lk + 4 * q;
        const int r = jn + i;
#pragma unroll
        for (int q1 = 0; q1 < 4; q1++) {
            Bf[sw(i, 16 * q1 + lr)] = i < nb2 ? D[q1][q] : 0.0;
            if (i >= nb2 && r < m) ptr[q][16 * q1] = D[q1][q];       // rows of this workgroup past the pivots: final as they are
        }
    }
    __syncthreads();
    wide_diag_solve<false>(dv, nb2, Bf, xt + (int64_t)(first + jn) * 64, w, lr, lk);
}

// Backward, first launch of a level: t = y - L21' x_below for 64 pivots per workgroup (a wave = 16 of them; the rows of x are
// gathered from XT), then x of the LAST block by the workgroup that holds it.
__global__ __launch_bounds__(256) void k_wide_bwd_big_head(DevSym ds, const int32_t *__restrict__ list, const double *__restrict__ Lx,
                                                           const double *__restrict__ Linv, double *__restrict__ XT, int64_t n)
{
    __shared__ double Bf[64 * 64];
    unsigned bx, by, ch;
    kvx_part_front_rhs(bx, by, ch);
    const FrontDesc fd = ds.fd[list[by]];
    const int k = fd.k, m = fd.m, first = fd.first;
    if ((int)bx * 64 >= k) return;
    const int tid = threadIdx.x, l = tid & 63, lr = l & 15, lk = l >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double *P = Lx + fd.px;
    double *xt = XT + (int64_t)ch * n * 64;
    const int32_t *rows = ds.rowidx + fd.rowptr;
    const int p0 = (int)bx * 64 + 16 * w;
    const bool last = ((int)bx + 1) * 64 >= k;
    const int jb = (int)bx * 64, nb = k - jb;
    double dv[16];
    if (last) wide_diag_load<true>(Linv + fd.linv + (int64_t)(jb / 64) * 4096, nb, w, lr, lk, dv);
    d4 D[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int p = p0 + lk + 4 * q;
#pragma unroll
        for (int q1 = 0; q1 < 4; q1++) D[q1][q] = kvx_ld0(xt, (int64_t)(first + p) * 64 + 16 * q1 + lr, p < k);
    }
    if (p0 < k) {                                  // wave-uniform
        const bool pin = p0 + lr < k;
        const double *Pc = P + (int64_t)(pin ? p0 + lr : 0) * m;
        // (the row indices and panel entries of the NEXT 16 rows are fetched while the gathered rows of x of this step arrive)
        double avn[4];
        int grn[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int r = k + 4 * s + lk;
            avn[s] = -kvx_ld0(Pc, r, pin && r < m);
            grn[s] = rows[min(r, m - 1)];
        }
        // two steps deep: the gathered rows of step i + 1 and the indices of step i + 2 are in flight during the MFMAs of step i
        double xgn[4][4], av1[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            av1[s] = avn[s];
#pragma unroll
            for (int q1 = 0; q1 < 4; q1++) xgn[s][q1] = xt[(int64_t)grn[s] * 64 + 16 * q1 + lr];
            const int r = k + 16 + 4 * s + lk;
            avn[s] = -kvx_ld0(Pc, r, pin && r < m);
            grn[s] = rows[min(r, m - 1)];
        }
        for (int r0 = k; r0 < m; r0 += 16) {
            double av[4], xg[4][4];
#pragma unroll
            for (int s = 0; s < 4; s++) {
                av[s] = av1[s];
                av1[s] = avn[s];
#pragma unroll
                for (int q1 = 0; q1 < 4; q1++) {
                    xg[s][q1] = xgn[s][q1];
                    xgn[s][q1] = xt[(int64_t)grn[s] * 64 + 16 * q1 + lr];
                }
            }
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int r = r0 + 32 + 4 * s + lk;
                avn[s] = -kvx_ld0(Pc, r, pin && r < m);
                grn[s] = rows[min(r, m - 1)];
            }
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const bool rin = r0 + 4 * s + lk < m;
#pragma unroll
                for (int q1 = 0; q1 < 4; q1++) D[q1] = mfma(av[s], rin ? xg[s][q1] : 0.0, D[q1]);
            }
        }
    }
    if (!last) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int p = p0 + lk + 4 * q;
            if (p < k) {
#pragma unroll
                for (int q1 = 0; q1 < 4; q1++) xt[(int64_t)(first + p) * 64 + 16 * q1 + lr] = D[q1][q];
            }
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int i = 16 * w + lk + 4 * q;
#pragma unroll
        for (int q1 = 0; q1 < 4; q1++) Bf[sw(i, 16 * q1 + lr)] = i < nb ? D[q1][q] : 0.0;
    }
    __syncthreads();
    wide_diag_solve<true>(dv, nb, Bf, xt + (int64_t)(first + jb) * 64, w, lr, lk);
}

// Backward, step s: block b = (blocks of the front) - 1 - s holds its x; workgroup c < b takes L(b, c)' x_b off its 64 pivots, and
// the one just before the block (c == b - 1) goes on to solve its own.
__global__ __launch_bounds__(256) void k_wide_bwd_big_step(DevSym ds, const int32_t *__restrict__ list, int step,
                                                           const double *__restrict__ Lx, const double *__restrict__ Linv,
                                                           double *__restrict__ XT, int64_t n)
{
    __shared__ double Bf[64 * 64];
    unsigned bx, by, ch;
    kvx_part_front_rhs(bx, by, ch);
    const FrontDesc fd = ds.fd[list[by]];
    const int k = fd.k, m = fd.m, first = fd.first;
    const int b = (k + 63) / 64 - 1 - step;
    if ((int)bx >= b) return;                       // (b <= 0: nothing left for this front)
    const int jb = 64 * b, nb = min(64, k - jb);
    const int tid = threadIdx.x, l = tid & 63, lr = l & 15, lk = l >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double *P = Lx + fd.px;
    double *xt = XT + (int64_t)ch * n * 64;
    double dv[16];
    if ((int)bx == b - 1) wide_diag_load<true>(Linv + fd.linv + (int64_t)(b - 1) * 4096, 64, w, lr, lk, dv);
    for (int i = w; i < 64; i += 4) Bf[sw(i, l)] = kvx_ld0(xt, (int64_t)(first + jb + i) * 64 + l, i < nb);
    const int p0 = (int)bx * 64 + 16 * w;           // < jb: whole blocks of pivots
    d4 D[4];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int q1 = 0; q1 < 4; q1++) D[q1][q] = xt[(int64_t)(first + p0 + lk + 4 * q) * 64 + 16 * q1 + lr];
    const double *Pc = P + (int64_t)(p0 + lr) * m + jb;
    double av[16];
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const int j = 4 * s + lk;
        av[s] = -kvx_ld0(Pc, j, j < nb);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const int j = 4 * s + lk;
#pragma unroll
        for (int q1 = 0; q1 < 4; q1++) D[q1] = mfma(av[s], Bf[sw(j, 16 * q1 + lr)], D[q1]);
    }
    if ((int)bx != b - 1) {
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int q1 = 0; q1 < 4; q1++) xt[(int64_t)(first + p0 + lk + 4 * q) * 64 + 16 * q1 + lr] = D[q1][q];
        return;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int q1 = 0; q1 < 4; q1++) Bf[sw(16 * w + lk + 4 * q, 16 * q1 + lr)] = D[q1][q];
    __syncthreads();
    wide_diag_solve<true>(dv, 64, Bf, xt + (int64_t)(first + 64 * (b - 1)) * 64, w, lr, lk);
}

// ---- launchers ----------------------------------------------------------------------------------------------------------------
void launch_wide_gather(hipStream_t st, const int32_t *iperm, int64_t n, int nrhs, const double *B, int64_t ldB, double *XT,
                        const double *diag, int mode)
{
    if (n <= 0 || nrhs <= 0) return;
    hipLaunchKernelGGL(k_wide_gather, dim3((unsigned)((n + 63) / 64), (unsigned)((nrhs + 63) / 64)), dim3(256), 0, st, iperm, n, nrhs, B, ldB, XT,
                       diag, mode);
}
void launch_wide_scatter(hipStream_t st, const int32_t *iperm, int64_t n, int nrhs, const double *XT, double *B, int64_t ldB,
                         const double *diag, int mode)
{
    if (n <= 0 || nrhs <= 0) return;
    hipLaunchKernelGGL(k_wide_scatter, dim3((unsigned)((n + 63) / 64), (unsigned)((nrhs + 63) / 64)), dim3(256), 0, st, iperm, n, nrhs, XT, B, ldB,
                       diag, mode);
}
void launch_wide_fwd_small(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int kmax, int nchunk, const double *Lx,
                           double *XT, int64_t n, const double *Wch, double *Wout, int64_t wstride, const int32_t *inv_ptr,
                           const int32_t *inv_src)
{
    if (count <= 0 || nchunk <= 0) return;
    const dim3 g((unsigned)count, (unsigned)nchunk);
    if (kmax <= 32) hipLaunchKernelGGL(k_wide_fwd_small<32>, g, dim3(64), 0, st, ds, list, Lx, XT, n, Wch, Wout, wstride, inv_ptr, inv_src);
    else hipLaunchKernelGGL(k_wide_fwd_small<64>, g, dim3(64), 0, st, ds, list, Lx, XT, n, Wch, Wout, wstride, inv_ptr, inv_src);
}
void launch_wide_bwd_small(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int kmax, int nchunk, const double *Lx,
                           double *XT, int64_t n)
{
    if (count <= 0 || nchunk <= 0) return;
    const dim3 g((unsigned)count, (unsigned)nchunk);
    if (kmax <= 32) hipLaunchKernelGGL(k_wide_bwd_small<32>, g, dim3(64), 0, st, ds, list, Lx, XT, n);
    else hipLaunchKernelGGL(k_wide_bwd_small<64>, g, dim3(64), 0, st, ds, list, Lx, XT, n);
}
void launch_wide_fwd_big(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int max_k, int nchunk,
                         const double *Lx, const double *Linv, double *XT, int64_t n, const double *Wch, double *Wout,
                         int64_t wstride, const int32_t *inv_ptr, const int32_t *inv_src)
{
    if (count <= 0 || nchunk <= 0) return;
    hipLaunchKernelGGL(k_wide_fwd_big_asm, dim3((unsigned)((max_m + 63) / 64), (unsigned)count, (unsigned)nchunk), dim3(256), 0, st, ds, list,
                       Linv, XT, n, Wch, Wout, wstride, inv_ptr, inv_src);
    for (int jb = 0; jb < max_k; jb += 64) {
        const int rows = max_m - jb - 1;            // (a front whose block is shorter than 64 has more rows below: one workgroup more)
        if (rows <= 0) break;
        hipLaunchKernelGGL(k_wide_fwd_big_step, dim3((unsigned)((rows + 63) / 64), (unsigned)count, (unsigned)nchunk), dim3(256), 0, st, ds, list,
                           jb, Lx, Linv, XT, n, Wout, wstride);
    }
}
void launch_wide_bwd_big(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_k, int nchunk, const double *Lx,
                         const double *Linv, double *XT, int64_t n)
{
    if (count <= 0 || nchunk <= 0) return;
    const int nblk = (max_k + 63) / 64;
    hipLaunchKernelGGL(k_wide_bwd_big_head, dim3((unsigned)nblk, (unsigned)count, (unsigned)nchunk), dim3(256), 0, st, ds, list, Lx, Linv, XT, n);
    for (int s = 0; s + 1 < nblk; s++)
        hipLaunchKernelGGL(k_wide_bwd_big_step, dim3((unsigned)(nblk - 1 - s), (unsigned)count, (unsigned)nchunk), dim3(256), 0, st, ds, list, s,
                           Lx, Linv, XT, n);
}

}  // namespace kvx
