// HIP kernels of the sparse LU path (gfx950): multifrontal LU on a static structure with threshold partial
// pivoting inside each front's pivot block.  API role: klu_factor / klu_solve / klu_tsolve as called from
// src/C/klu.c:161, :187-198, :651-665.  Design notes in lu_symbolic.hpp.
//
// One workgroup per front and one launch per (level, size class).  A front is a dense m x m matrix: k pivot rows /
// columns first, then u = m - k update rows / columns.  Small fronts live in LDS for the whole assemble-factor-store
// sequence; larger ones in their own m x m region of the arena in HBM (their update matrix is read from there by
// the parent) and are factored by column blocks of 32 pivots with two or three launches per block (panel; row interchanges
// where the block has any; rank-32 update on 64 x 64 FP64-MFMA tiles, which also solves its own columns of U12).
// Latency-bound work: fronts of circuit / power-flow matrices are tens to hundreds of rows.
#include "lu_device.hpp"
#include <algorithm>
#include <cstdlib>

namespace kvx {

namespace {

constexpr int LU_NT_LDS = 256;
constexpr int LU_NT_BIG = 1024;
constexpr int LU_NT_SOLVE = 256;
constexpr int LU_SB = 32;                        // pivots per block in the triangular sweeps

__device__ __forceinline__ double readlane_d(const double v, const int lane)      // lane must be wave-uniform
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// Right-looking elimination of the k pivot columns of the m x m front Fm (leading dimension ld).
// KLU's pivot rule (klu.h: Common.tol = 0.001, "partial pivoting with diagonal preference"): keep the diagonal
// entry when |d| >= tol * max|candidates|, else take the largest candidate -- candidates are the rows of the pivot
// block only.  A step fails when the best candidate is zero / not finite or smaller than stol * (largest entry of
// the whole front column): the host then merges this front into its parent and factors again.
template <int NT>
__device__ void lu_factor_front(double *Fm, const int ld, const int m, const int k, int32_t *ipiv, int32_t *fail_slot,
                                const double tol, const double stol, const int reuse, int *sh_i, double *sh_d, int32_t *sh_piv)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int tx = tid & 63, ty = tid >> 6;
    constexpr int NW = NT / 64;
    bool failed = false;
    double lmax = 0.0;                                           // largest multiplier seen by this thread (acceptance test below)
    if (reuse) {                                                 // the recorded pivot sequence: one coalesced load, not one per step
        for (int t = tid; t < k; t += NT) sh_piv[t] = ipiv[t];
        __syncthreads();
    }
    for (int j = 0; j < k; j++) {
        if (tid < 64) {                                          // search among the rows of the pivot block only
            double bmax = -1.0;
            int bidx = j;
            for (int i = j + lane; i < k; i += 64) {
                const double a = fabs(Fm[i + (int64_t)j * ld]);
                if (a > bmax) { bmax = a; bidx = i; }
            }
            for (int off = 32; off; off >>= 1) {
                const double ob = __shfl_xor(bmax, off);
                const int oi = __shfl_xor(bidx, off);
                if (ob > bmax || (ob == bmax && oi < bidx)) { bmax = ob; bidx = oi; }
            }
            if (lane == 0) {
                const double diag = fabs(Fm[j + (int64_t)j * ld]);
                int r = (diag > 0.0 && diag >= tol * bmax) ? j : bidx;
                if (reuse) r = j + sh_piv[j];
                double pv = Fm[r + (int64_t)j * ld];
                const double ap = fabs(pv);
                const bool bad = !(ap > 0.0) || !(ap <= 1.7e308);
                if (bad && !failed) { failed = true; *fail_slot = j + 1; }
                if (bad) pv = 1.0;                                // keep going with finite numbers; the result is discarded
                sh_piv[j] = r - j;
                sh_i[0] = r;
                sh_d[0] = pv;
            }
        }
        __syncthreads();
        const int r = sh_i[0];
        const double pv = sh_d[0];
        if (r != j) {                                            // uniform
            for (int c = tid; c < m; c += NT) {
                const double a = Fm[j + (int64_t)c * ld], b = Fm[r + (int64_t)c * ld];
                Fm[j + (int64_t)c * ld] = b;
                Fm[r + (int64_t)c * ld] = a;
            }
            __syncthreads();
        }
        // scale and rank-1 update in one pass: every wave forms the multipliers of its rows itself
        for (int i = j + 1 + tx; i < m; i += 64) {
            const double l = Fm[i + (int64_t)j * ld] / pv;
            lmax = fmax(lmax, fabs(l));
            for (int c = j + 1 + ty; c < m; c += NW) Fm[i + (int64_t)c * ld] -= l * Fm[j + (int64_t)c * ld];
        }
        __syncthreads();
        if (ty == 0)
            for (int i = j + 1 + tx; i < m; i += 64) Fm[i + (int64_t)j * ld] /= pv;
    }
    // |pivot| >= stol * max|column|  <=>  max|multiplier| <= 1 / stol : one test per front instead of one per pivot
    if (!(lmax * stol <= 1.0)) atomicMax(fail_slot, 1);
    __syncthreads();
    if (!reuse)
        for (int t = tid; t < k; t += NT) ipiv[t] = sh_piv[t];
}

// Assemble a front: zero, scatter the (row-scaled) entries of A, extend-add the children's update matrices
// (the parent pulls: no atomics, reproducible).
template <int NT>
__device__ void lu_assemble_front(const LuDev &d, const LuFrontD &F, double *Fm, const int ld, const double *__restrict__ Ax)
{
    const int tid = threadIdx.x;
    const int m = F.m;
    for (int64_t idx = tid; idx < (int64_t)m * m; idx += NT) Fm[idx] = 0.0;
    __syncthreads();
    for (int64_t e = tid; e < F.acnt; e += NT) {                 // every entry has its own slot
        const int64_t src = d.a_src[F.aptr + e];
        Fm[d.a_dst[F.aptr + e]] += Ax[src] * d.rinv[d.ai32[src]];
    }
    __syncthreads();
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
}

// Panels and in-front permutation of a factored front: L(:, 0:k) as is, U(0:k, :) transposed, both m x k column-major.
template <int NT>
__device__ void lu_store_front(const LuDev &d, const LuFrontD &F, const double *Fm, const int ld, int32_t *sh_lp, int32_t *sh_piv)
{
    const int tid = threadIdx.x;
    const int m = F.m, k = F.k;
    for (int t = tid; t < k; t += NT) { sh_lp[t] = t; sh_piv[t] = d.ipiv[F.p0 + t]; }
    __syncthreads();
    if (tid == 0)
        for (int j = 0; j < k; j++) {
            const int r = j + sh_piv[j];
            if (r != j) { const int a = sh_lp[j]; sh_lp[j] = sh_lp[r]; sh_lp[r] = a; }
        }
    __syncthreads();
    for (int t = tid; t < k; t += NT) d.lperm[F.p0 + t] = sh_lp[t];
    double *__restrict__ Lp = d.Lx + F.px, *__restrict__ Up = d.Ux + F.px;
    for (int64_t idx = tid; idx < (int64_t)m * k; idx += NT) Lp[idx] = Fm[idx];
    const int tx = tid & 63, ty = tid >> 6;
    for (int t = ty; t < k; t += NT / 64)
        for (int c = tx; c < m; c += 64) Up[c + (int64_t)t * m] = Fm[t + (int64_t)c * ld];
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
    int32_t *sh_piv = sh_lp + (LDS ? lds_m : k);
    const int ld = m;
    if (tid == 0) d.fail[f] = 0;
    lu_assemble_front<NT>(d, F, Fm, ld, Ax);
    lu_factor_front<NT>(Fm, ld, m, k, d.ipiv + F.p0, d.fail + f, tol, stol, reuse, sh_i, sh_d, sh_piv);
    __syncthreads();
    lu_store_front<NT>(d, F, Fm, ld, sh_lp, sh_piv);
    if (LDS) {
        double *__restrict__ Uo = d.arena + F.upd_off;
        const int tx = tid & 63, ty = tid >> 6;
        for (int j = ty; j < u; j += NT / 64)
            for (int i = tx; i < u; i += 64) Uo[i + (int64_t)j * u] = Fm[(k + i) + (int64_t)(k + j) * ld];
    }
}


// Register-tiled elimination of an LDS-resident front (m <= 16 T).  The LDS version above moves 24 bytes through LDS per
// multiply-add of the rank-1 update and is LDS-bandwidth bound (a front of order 136 with 100 pivots: ~90 us).  Here the
// 256 threads form a 16 x 16 grid, thread (tx, ty) keeps the entries (tx + 16 a, ty + 16 b), a, b < T, in registers; per
// pivot the pivot row and the multiplier column travel through LDS once (2 T reads per thread instead of 2 T^2) and the
// update is T^2 register FMAs.  Rows are never moved: each thread carries the logical position (`slot`) of its rows under
// the LAPACK-style interchange sequence, the tile is written back to the logical positions at the end.  Pivot steps are
// unrolled over the column-block index so that every register index is a compile-time constant.
template <int T>
struct TileCtx {
    double *lcol, *urow, *sh_pv;
    int *sh_r, *sh_piv;
    int32_t *fail_slot;
    double tol, lmax;
    int m, k, tx, ty, reuse;
    int slot[T];
    bool failed;
};

// Reductions over the 16 lanes of a DPP row by row rotations (ror 8, 4, 2, 1: every lane of the row ends with the result): the pivot
// search of a tiled front ran four rounds of six __shfl_xor (nine ds_bpermute, an LDS crossbar round trip each) per pivot.
template <int N>
__device__ __forceinline__ int row_ror_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x120 + N, 0xf, 0xf, false); }
template <int N>
__device__ __forceinline__ double row_ror_f64(double v)
{ return __hiloint2double(row_ror_i32<N>(__double2hiint(v)), row_ror_i32<N>(__double2loint(v))); }
__device__ __forceinline__ double row16_max_f64(double v)
{ v = fmax(v, row_ror_f64<8>(v)); v = fmax(v, row_ror_f64<4>(v)); v = fmax(v, row_ror_f64<2>(v)); return fmax(v, row_ror_f64<1>(v)); }
__device__ __forceinline__ int row16_min_i32(int v)
{ v = min(v, row_ror_i32<8>(v)); v = min(v, row_ror_i32<4>(v)); v = min(v, row_ror_i32<2>(v)); return min(v, row_ror_i32<1>(v)); }
__device__ __forceinline__ int row16_max_i32(int v)
{ v = max(v, row_ror_i32<8>(v)); v = max(v, row_ror_i32<4>(v)); v = max(v, row_ror_i32<2>(v)); return max(v, row_ror_i32<1>(v)); }

template <int T, int BJ>
__device__ __forceinline__ void tile_block(double (&a)[T][T], TileCtx<T> &x)
{
    if constexpr (BJ < T) {
        for (int jj = 0; jj < 16; jj++) {
            const int j = 16 * BJ + jj;
            if (j >= x.k) break;                                  // uniform
            // (1) the 16 threads holding column j (ty == jj: 16 consecutive lanes of one wavefront) choose the pivot
            if (x.ty == jj) {
                double bmax = -1.0, bval = 0.0, dval = 0.0;
                int brow = -1, bslot = 0x7fffffff, qrow = -1;
#pragma unroll
                for (int ar = 0; ar < T; ar++) {
                    const int i = x.tx + 16 * ar;
                    const int sl = x.slot[ar];
                    const double v = a[ar][BJ], av = fabs(v);
                    const bool cand = i < x.k && sl >= j;         // rows of the pivot block not yet used
                    if (x.reuse) {
                        if (cand && sl == j + x.sh_piv[j]) { bmax = av; bval = v; brow = i; bslot = sl; }
                    } else if (cand && (av > bmax || (av == bmax && sl < bslot))) { bmax = av; bval = v; brow = i; bslot = sl; }
                    if (i < x.k && sl == j) { qrow = i; dval = v; }
                }
                // the largest candidate of the 16 lanes, the smallest slot among those that hold it, the diagonal's row: three reductions
                // by row rotations; the values of the chosen row never travel -- the lane that holds it publishes
                const double gmax = row16_max_f64(bmax);
                const int gslot = row16_min_i32(bmax == gmax ? bslot : 0x7fffffff);     // (0x7fffffff: no candidate row at all)
                const int gq = row16_max_i32(qrow);
                // KLU's rule: keep the diagonal (the row at logical position j) when |d| >= tol * max -- the lane that holds it decides
                const int keep = row16_max_i32((!x.reuse && qrow >= 0 && fabs(dval) > 0.0 && fabs(dval) >= x.tol * gmax) ? 1 : 0);
                bool me;
                if (keep) { me = qrow >= 0; brow = qrow; bval = dval; bslot = j; }
                else if (gslot != 0x7fffffff) me = bmax == gmax && bslot == gslot;
                else { me = x.tx == 0; brow = -1; bval = 0.0; }
                if (me) {
                    const double ap = fabs(bval);
                    const bool bad = brow < 0 || !(ap > 0.0) || !(ap <= 1.7e308);
                    if (bad && *x.fail_slot == 0) *x.fail_slot = j + 1;
                    if (bad) { bval = 1.0; if (brow < 0) { brow = gq; bslot = j; } }
                    x.sh_r[0] = brow; x.sh_r[1] = gq; x.sh_r[2] = bslot;
                    x.sh_pv[0] = bval;
                    x.sh_piv[j] = bslot - j;
                }
            }
            __syncthreads();
            const int r = x.sh_r[0], q = x.sh_r[1], sr = x.sh_r[2];
            const double pv = x.sh_pv[0];
            // (2) interchange = relabel; the owners of the pivot row publish it, the owners of column j the multipliers
#pragma unroll
            for (int ar = 0; ar < T; ar++) {
                const int i = x.tx + 16 * ar;
                if (i == r) x.slot[ar] = j;
                else if (i == q) x.slot[ar] = sr;
            }
#pragma unroll
            for (int ar = 0; ar < T; ar++)
                if (x.tx + 16 * ar == r) {                        // (tested on the row number, not on `ar`: a comparison of the
#pragma unroll                                                    //  index itself lets LLVM index the tile dynamically -> scratch)
                    for (int bc = 0; bc < T; bc++) x.urow[x.ty + 16 * bc] = a[ar][bc];
                }
            if (x.ty == jj) {
#pragma unroll
                for (int ar = 0; ar < T; ar++) {
                    const int i = x.tx + 16 * ar;
                    double l = 0.0;
                    if (i < x.m && x.slot[ar] > j) {
                        l = a[ar][BJ] / pv;
                        a[ar][BJ] = l;
                        x.lmax = fmax(x.lmax, fabs(l));
                    }
                    x.lcol[i] = l;
                }
            }
            __syncthreads();
            // (3) rank-1 update of the columns right of j, in registers
            double lr[T];
#pragma unroll
            for (int ar = 0; ar < T; ar++) lr[ar] = x.lcol[x.tx + 16 * ar];
            if (x.ty > jj) {
                const double u0 = x.urow[x.ty + 16 * BJ];
#pragma unroll
                for (int ar = 0; ar < T; ar++) a[ar][BJ] -= lr[ar] * u0;
            }
#pragma unroll
            for (int bc = BJ + 1; bc < T; bc++) {
                const double uc = x.urow[x.ty + 16 * bc];
#pragma unroll
                for (int ar = 0; ar < T; ar++) a[ar][bc] -= lr[ar] * uc;
            }
        }
        tile_block<T, BJ + 1>(a, x);
    }
}

template <int T>
__device__ void lu_factor_front_tiled(double *Fm, const int ld, const int m, const int k, int32_t *ipiv, int32_t *fail_slot,
                                      const double tol, const double stol, const int reuse, double *lcol, double *urow, int *sh_r,
                                      double *sh_pv, int32_t *sh_piv)
{
    const int tid = threadIdx.x;
    TileCtx<T> x;
    x.lcol = lcol; x.urow = urow; x.sh_pv = sh_pv; x.sh_r = sh_r; x.sh_piv = sh_piv; x.fail_slot = fail_slot;
    x.tol = tol; x.lmax = 0.0; x.m = m; x.k = k; x.tx = tid & 15; x.ty = tid >> 4; x.reuse = reuse; x.failed = false;
    if (reuse) {
        for (int t = tid; t < k; t += 256) sh_piv[t] = ipiv[t];
    }
    double a[T][T];
#pragma unroll
    for (int ar = 0; ar < T; ar++) {
        const int i = x.tx + 16 * ar;
        x.slot[ar] = i;
#pragma unroll
        for (int bc = 0; bc < T; bc++) {
            const int c = x.ty + 16 * bc;
            a[ar][bc] = (i < m && c < m) ? Fm[i + (int64_t)c * ld] : 0.0;
        }
    }
    __syncthreads();
    tile_block<T, 0>(a, x);
    __syncthreads();
    // back to LDS, every row at its logical position
#pragma unroll
    for (int ar = 0; ar < T; ar++) {
        const int i = x.tx + 16 * ar;
        if (i < m) {
#pragma unroll
            for (int bc = 0; bc < T; bc++) {
                const int c = x.ty + 16 * bc;
                if (c < m) Fm[x.slot[ar] + (int64_t)c * ld] = a[ar][bc];
            }
        }
    }
    if (!(x.lmax * stol <= 1.0)) atomicMax(fail_slot, 1);        // |pivot| >= stol * max|column|  <=>  max|multiplier| <= 1 / stol
    __syncthreads();
    if (!reuse)
        for (int t = tid; t < k; t += 256) ipiv[t] = sh_piv[t];
}

template <int T>
__global__ __launch_bounds__(256, 1) void k_lu_front_tiled(const LuDev d, const int32_t *__restrict__ list, const double *__restrict__ Ax,
                                                         const double tol, const double stol, const int reuse, const int lds_m)
{
    extern __shared__ double smem[];
    __shared__ int sh_r[4];
    __shared__ double sh_pv[2];
    __shared__ double lcol[16 * T], urow[16 * T];
    const int tid = threadIdx.x;
    const int f = list[blockIdx.x];
    const LuFrontD F = d.fr[f];
    const int m = F.m, k = F.k, u = m - k;
    double *Fm = smem;
    int32_t *sh_lp = (int32_t *)(smem + (int64_t)lds_m * lds_m);
    int32_t *sh_piv = sh_lp + lds_m;
    const int ld = m;
    if (tid == 0) d.fail[f] = 0;
    lu_assemble_front<256>(d, F, Fm, ld, Ax);
    lu_factor_front_tiled<T>(Fm, ld, m, k, d.ipiv + F.p0, d.fail + f, tol, stol, reuse, lcol, urow, sh_r, sh_pv, sh_piv);
    __syncthreads();
    lu_store_front<256>(d, F, Fm, ld, sh_lp, sh_piv);
    double *__restrict__ Uo = d.arena + F.upd_off;
    const int tx = tid & 63, ty = tid >> 6;
    for (int j = ty; j < u; j += 4)
        for (int i = tx; i < u; i += 64) Uo[i + (int64_t)j * u] = Fm[(k + i) + (int64_t)(k + j) * ld];
}


// ------------------------------------------------------------------------------------------
// Round 4: blocked elimination of an LDS-resident front, the pivot panels factored by ONE wavefront.
// k_lu_front_tiled pays two workgroup barriers and three LDS round trips per pivot (rocprofv3, ACTIVSg2000: 1.3 us per pivot on a
// leaf level of k <= 28, 2.5 us on the 106-pivot root -- 260 of the refactorisation's 720 us).  Here the pivot columns go in
// panels of 32: wave 0 holds the panel in registers (lane = physical row, two rows per lane from order 65 on) and runs its
// pivot steps without a barrier; then every thread solves one column of U12 against the panel's unit triangle (staged in LDS,
// broadcast reads), and the 16 x 16 thread grid applies the rank-32 update to the columns right of the panel from register
// tiles: three barriers per 32 pivots.
// Rows never move: `slot` (logical position under the LAPACK-style interchange sequence, the same record k_lu_front_tiled and
// the blocked HBM path keep in ipiv) and `prow` (physical row of every pivot) live in LDS; the store reads through them.  A
// refactorisation (reuse: klu.c:296-308, the recorded pivot sequence) knows both before it starts -- lperm of the previous
// factorisation IS prow -- so its pivot steps carry no search, no ballot and no relabelling at all.
// The front has one row more than its order, kept zero (the lanes past the last row read it), and an odd leading dimension
// (the column-per-lane accesses of the U12 solve hit distinct banks).
// What the steps cost (MI355X, phase counters of a development build): the steps are bound by instruction issue of the one
// wave, and v_readlane_b32 is ~12 cycles apiece -- broadcasting the pivot row's 31 remaining entries by 62 of them made a
// step 1270 cycles.  The owner lane now writes its row to LDS (b128) and every lane reads it back as a broadcast.
constexpr int LU_PB = 32;
#ifdef KVX_LU_PHASE
__device__ unsigned long long g_lu_phase[16];
#define LU_STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_lu_phase[i], t_ - lu_t_); lu_t_ = t_; } } while (0)
#define LU_TPARAM , unsigned long long &lu_t_
#define LU_TARG , lu_t_
#else
#define LU_STAMP(i) do { } while (0)
#define LU_TPARAM
#define LU_TARG
#endif

// lanes of one wavefront exchanging data through LDS: the hardware runs a wave's LDS operations in order, the compiler has to be
// told that another lane's store is visible to this lane's load (it forwarded the lane's own stored value otherwise)
__device__ __forceinline__ void wp_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double wp_max_f64(double v)
{
    v = row16_max_f64(v);
    return fmax(fmax(readlane_d(v, 0), readlane_d(v, 16)), fmax(readlane_d(v, 32), readlane_d(v, 48)));
}
__device__ __forceinline__ int wp_min_i32(int v)
{
    v = row16_min_i32(v);
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

struct WpCtx {
    int j0, nb, k, lane;
    int pivreg;                    // lane jj: reuse -- physical row of pivot j0 + jj
    int32_t *sh_piv, *sh_prow;
    int32_t *fail_slot;
    double *bc;                    // 32 doubles of LDS: the pivot row on its way to the other lanes
    double tol, lmax;
};

// Pivot step jj of the panel (pivot j = j0 + jj).  The panel lives in a WINDOW of W register columns per row: a[r][Q] is the
// step's pivot column, a[r][c], c > Q, the columns right of it (lane's physical rows lane + 64 r).
template <int R, int W, int Q, bool REUSE>
__device__ __forceinline__ void wp_step(double (&a)[R][W], int (&slot)[R], WpCtx &x, const int jj)
{
    if (jj >= x.nb) return;                                       // uniform
    const int j = x.j0 + jj;
    int rl, rh;                                                   // lane and register row of the pivot row (uniform)
    if (REUSE) {
        const int pr = __builtin_amdgcn_readlane(x.pivreg, jj);
        rl = pr & 63; rh = pr >> 6;
    } else {
        // KLU's rule (see lu_factor_front): the diagonal if |d| >= tol * max|candidates|, else the largest candidate (the smallest
        // logical position among equals); candidates are the rows of the pivot block not used yet
        double bmax = -1.0, dabs = -1.0;
        int bslot = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const double av = fabs(a[r][Q]);
            const bool cand = x.lane + 64 * r < x.k && slot[r] >= j;
            if (cand && (av > bmax || (av == bmax && slot[r] < bslot))) { bmax = av; bslot = slot[r]; }
            if (slot[r] == j) dabs = av;
        }
        const double gmax = wp_max_f64(bmax);
        const int gslot = wp_min_i32(bmax == gmax ? bslot : 0x7fffffff);
        const double gd = wp_max_f64(dabs);                       // (one lane holds the row at position j; NaN compares false below)
        const int want = (gd > 0.0 && gd >= x.tol * gmax) ? j : (gslot != 0x7fffffff ? gslot : j);
        // the lane and register row of `want`, and of the row at position j; interchange = relabel
        const unsigned long long w0 = __ballot(slot[0] == want), w1 = R == 2 ? __ballot(slot[R - 1] == want) : 0ull;
        const unsigned long long q0 = __ballot(slot[0] == j), q1 = R == 2 ? __ballot(slot[R - 1] == j) : 0ull;
        rh = w0 ? 0 : 1;
        const int qh = q0 ? 0 : 1;
        rl = __builtin_amdgcn_readfirstlane(__ffsll((long long)(w0 ? w0 : w1)) - 1);
        const int ql = __builtin_amdgcn_readfirstlane(__ffsll((long long)(q0 ? q0 : q1)) - 1);
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (x.lane == rl && r == rh) slot[r] = j;
            else if (x.lane == ql && r == qh) slot[r] = want;
        }
        if (x.lane == 0) { x.sh_piv[j] = want - j; x.sh_prow[j] = rl + 64 * rh; }
    }
#ifndef WP_BCAST_READLANE
    wp_wave_sync();                                               // (the previous step's reads are done)
    if (x.lane == rl) {                                           // the pivot row, columns right of the pivot, on its way to everybody
#pragma unroll
        for (int c = Q + 1; c < W; c++) x.bc[c] = R == 2 && rh ? a[R - 1][c] : a[0][c];
    }
    wp_wave_sync();
#endif
    double pv = readlane_d(R == 2 && rh ? a[R - 1][Q] : a[0][Q], rl);
    const double ap = fabs(pv);
    const bool bad = !(ap > 0.0) || !(ap <= 1.7e308);            // uniform
    if (bad) {
        if (x.lane == 0 && *x.fail_slot == 0) *x.fail_slot = j + 1;
        pv = 1.0;                                                 // keep going with finite numbers; the result is discarded
    }
    double l[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const bool on = slot[r] > j;                              // rows not used yet (pivot block and update rows)
        const double q = a[r][Q] / pv;
        l[r] = on ? q : 0.0;
        a[r][Q] = on ? q : a[r][Q];
        x.lmax = fmax(x.lmax, fabs(l[r]));
    }
#pragma unroll
    for (int c = Q + 1; c < W; c++) {
#ifdef WP_BCAST_READLANE
        const double uc = readlane_d(R == 2 && rh ? a[R - 1][c] : a[0][c], rl);
#else
        const double uc = x.bc[c];
#endif
#pragma unroll
        for (int r = 0; r < R; r++) a[r][c] -= l[r] * uc;
    }
}

// Four pivot steps per round on the window, then the four finished columns go out to the front and the window moves on.
template <int R, int W, bool REUSE>
__device__ __forceinline__ void wp_window(double (&a)[R][W], int (&slot)[R], const bool (&mine)[R], WpCtx &x, int &jj, const int stop,
                                          double *Fm, const int ld)
{
    while (x.nb - jj > stop) {
        wp_step<R, W, 0, REUSE>(a, slot, x, jj);
        wp_step<R, W, 1, REUSE>(a, slot, x, jj + 1);
        wp_step<R, W, 2, REUSE>(a, slot, x, jj + 2);
        wp_step<R, W, 3, REUSE>(a, slot, x, jj + 3);
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (mine[r]) {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (jj + q < x.nb) Fm[x.lane + 64 * r + (x.j0 + jj + q) * ld] = a[r][q];
            }
#pragma unroll
            for (int c = 0; c < W - 4; c++) a[r][c] = a[r][c + 4];
#pragma unroll
            for (int c = W - 4; c < W; c++) a[r][c] = 0.0;
        }
        jj += 4;
    }
}
// (columns past the panel: whatever the front holds there -- they are updated along and never written back)
template <int R, int W>
__device__ __forceinline__ void wp_window_load(double (&a)[R][W], const double *Fm, const int ld, const int m, const WpCtx &x, const int jj)
{
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = min(x.lane + 64 * r, m);                    // row m: zeros
#pragma unroll
        for (int c = 0; c < W; c++) a[r][c] = Fm[i + min(x.j0 + jj + c, m - 1) * ld];
    }
}

// the panel of the nb <= 32 pivot columns from j0 on: wave 0; windows of 32, 16 and 8 columns as the panel shrinks
template <int R, bool REUSE>
__device__ __forceinline__ void wp_panel(double *Fm, const int ld, const int m, WpCtx &x, int32_t *sh_slot, double *L11)
{
    const int lane = x.lane, nb = x.nb, j0 = x.j0;
    int slot[R];
    bool mine[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = lane + 64 * r;
        slot[r] = i < m ? sh_slot[i] : 0x3fffffff;
        mine[r] = i < m && slot[r] >= j0;                         // rows used by earlier panels hold finished entries of U
    }
    x.pivreg = (REUSE && lane < nb) ? x.sh_prow[j0 + lane] : 0;
    int jj = 0;
    double a16[R][16], a8[R][8];
    if (nb > 16) {
        double a32[R][32];
        wp_window_load<R, 32>(a32, Fm, ld, m, x, 0);
        wp_window<R, 32, REUSE>(a32, slot, mine, x, jj, 16, Fm, ld);
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int c = 0; c < 16; c++) a16[r][c] = a32[r][c];
    } else if (nb > 8) wp_window_load<R, 16>(a16, Fm, ld, m, x, 0);
    if (nb > 8) {
        wp_window<R, 16, REUSE>(a16, slot, mine, x, jj, 8, Fm, ld);
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int c = 0; c < 8; c++) a8[r][c] = a16[r][c];
    } else wp_window_load<R, 8>(a8, Fm, ld, m, x, 0);
    wp_window<R, 8, REUSE>(a8, slot, mine, x, jj, 0, Fm, ld);
    // the new logical positions, and the rows of the pivot block in logical order for the solve that follows
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = lane + 64 * r;
        if (mine[r]) {
            if (!REUSE) sh_slot[i] = slot[r];
            if (slot[r] < j0 + nb) {
                double *Lr = L11 + (slot[r] - j0);
                const double *Fr = Fm + i;
                for (int c0 = 0; c0 < nb; c0 += 8) {
                    double v[8];
#pragma unroll
                    for (int t = 0; t < 8; t++) v[t] = Fr[min(j0 + c0 + t, m - 1) * ld];     // (past the panel: columns of the front, unused entries of L11)
#pragma unroll
                    for (int t = 0; t < 8; t++) Lr[(c0 + t) * LU_PB] = v[t];
                }
            }
        }
    }
}

// U12 = L11^-1 A12 for column c of the front (one thread), rows = the panel's pivot rows in order, by the same moving window
template <int W>
__device__ __forceinline__ void wp_solve_window(double (&xv)[W], int &jb, const int stop, double *Fm, const int ld, const int c, const int j0,
                                                const int nb, const int32_t *sh_prow, const double *L11)
{
    while (nb - jb > stop) {
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (jb + q < nb) {                                    // uniform
                const double xq = xv[q];
                const double *Lc = L11 + (jb + q) * LU_PB + jb;
#pragma unroll
                for (int j2 = q + 1; j2 < W; j2++) xv[j2] -= Lc[j2] * xq;
                Fm[sh_prow[j0 + jb + q] + c * ld] = xq;
            }
#pragma unroll
        for (int t = 0; t < W - 4; t++) xv[t] = xv[t + 4];
#pragma unroll
        for (int t = W - 4; t < W; t++) xv[t] = 0.0;
        jb += 4;
    }
}
template <int W>
__device__ __forceinline__ void wp_solve_load(double (&xv)[W], const double *Fm, const int ld, const int c, const int j0, const int nb,
                                              const int32_t *sh_prow)
{
#pragma unroll
    for (int t = 0; t < W; t++) {
        const double v = Fm[sh_prow[j0 + min(t, nb - 1)] + c * ld];
        xv[t] = t < nb ? v : 0.0;
    }
}
__device__ __forceinline__ void wp_solve(double *Fm, const int ld, const int c, const int j0, const int nb, const int32_t *sh_prow, const double *L11)
{
    int jb = 0;
    double x16[16], x8[8];
    if (nb > 16) {
        double x32[32];
        wp_solve_load<32>(x32, Fm, ld, c, j0, nb, sh_prow);
        wp_solve_window<32>(x32, jb, 16, Fm, ld, c, j0, nb, sh_prow, L11);
#pragma unroll
        for (int t = 0; t < 16; t++) x16[t] = x32[t];
    } else if (nb > 8) wp_solve_load<16>(x16, Fm, ld, c, j0, nb, sh_prow);
    if (nb > 8) {
        wp_solve_window<16>(x16, jb, 8, Fm, ld, c, j0, nb, sh_prow, L11);
#pragma unroll
        for (int t = 0; t < 8; t++) x8[t] = x16[t];
    } else wp_solve_load<8>(x8, Fm, ld, c, j0, nb, sh_prow);
    wp_solve_window<8>(x8, jb, 0, Fm, ld, c, j0, nb, sh_prow, L11);
}

// Assemble an LDS front of leading dimension ld (m + 1 rows): zero, A's entries (a_dst encodes r + c m), then the children's
// update matrices.  A child costs three dependent trips to memory (its record, its relative indices, its update matrix:
// ~1.6 us measured) and a front of a power-grid matrix has up to 28 children, so: the records of up to 32 children are fetched
// together, then their index lists (into LDS), and the columns of the update matrices go in batches of 16 loads in flight per
// wavefront.  Wave w owns the target columns c with c % 4 == w and walks the children in their order: a target entry
// receives its contributions in the order of lu_assemble_front whatever the waves' pace (reproducible sums), and no barrier
// stands between two children.
constexpr int WP_CH = 32, WP_REL = 1024, WP_ITEMS = WP_REL + 4 * WP_CH;
// A work item = one column of a child's update matrix that this wave owns: word 0 = offset of the column in the arena, word 1 =
// target column | rows << 7 | offset of the child's index list in `rel` << 14 (rows <= 127, target < 128, list offset < 1024).
// A wave's items are in child order, every child's run padded to a multiple of four (0 / 0 = nothing): four consecutive,
// aligned items belong to ONE child, i.e. to four different target columns.
struct WpAsmLds {
    int64_t ch_off[WP_CH], ch_relptr[WP_CH];
    int32_t ch_uc[WP_CH], ch_ldc[WP_CH], ch_rel[WP_CH];
    int32_t rel[WP_REL];
    uint2 items[4][WP_ITEMS];
    int32_t cnt;
};

// Sixteen items of wave w from position `it` on: the four 16-lane groups of the wave take four items at a time (group G the
// item it + 4 s + G, its lanes the rows r16 + 16 rr + row0 of the column), all 16 loads of a lane are in flight before the
// first of them is added into the front.
__device__ __forceinline__ void wp_run_items(const LuDev &d, WpAsmLds &S, double *Fm, const int ld, const int w, const int lane, const int it,
                                             const int cnt, const int row0)
{
    const int G = lane >> 4, r16 = lane & 15;
    double v[4][4];
    int dst[4][4];
#pragma unroll
    for (int sl = 0; sl < 4; sl++) {
        const int e = it + 4 * sl + G;
        const uint2 item = e < cnt ? S.items[w][e] : make_uint2(0u, 0u);
        const int tj = (int)(item.y & 127u), uc = (int)((item.y >> 7) & 127u), ro = (int)(item.y >> 14);
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int i = row0 + r16 + 16 * rr;
            const bool ok = i < uc;                               // (a padding item has no rows)
            v[sl][rr] = ok ? d.arena[(int64_t)item.x + i] : 0.0;
            dst[sl][rr] = ok ? S.rel[ro + i] + tj * ld : -1;
        }
    }
#pragma unroll
    for (int sl = 0; sl < 4; sl++)
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
            if (dst[sl][rr] >= 0) Fm[dst[sl][rr]] += v[sl][rr];
}

__device__ __forceinline__ void wp_assemble(const LuDev &d, const LuFrontD &F, double *Fm, const int ld, const double *__restrict__ Ax,
                                            WpAsmLds &S LU_TPARAM)
{
    const int tid = threadIdx.x, m = F.m;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    for (int idx = tid; idx < m * ld; idx += 256) Fm[idx] = 0.0;
    __syncthreads();
    for (int e = tid; e < F.acnt; e += 256) {                    // every entry has its own slot
        const int64_t src = d.a_src[F.aptr + e];
        const int dst = d.a_dst[F.aptr + e];
        const int c = dst / m;
        Fm[(dst - c * m) + c * ld] += Ax[src] * d.rinv[d.ai32[src]];
    }
    LU_STAMP(5);
    for (int c0 = 0; c0 < F.nchild;) {
        const int nc = min(WP_CH, F.nchild - c0);
        __syncthreads();                                          // (the tables of the previous chunk are no longer read; A's entries are in)
        if (tid < nc) {
            const LuFrontD C = d.fr[d.children[F.childptr + c0 + tid]];
            S.ch_uc[tid] = C.m - C.k; S.ch_ldc[tid] = C.upd_ld; S.ch_off[tid] = C.upd_off; S.ch_relptr[tid] = C.rowptr + C.k;
        }
        __syncthreads();
        if (tid == 0) {                                           // as many children as the index table holds (a child has <= 112 rows)
            int off = 0, n = 0;
            while (n < nc && off + S.ch_uc[n] <= WP_REL) { S.ch_rel[n] = off; off += S.ch_uc[n]; n++; }      // (and so <= WP_ITEMS items a wave)
            S.cnt = n;
        }
        __syncthreads();
        const int n = S.cnt;
        for (int c = w; c < n; c += 4) {
            const int uc = S.ch_uc[c];
            const int32_t *__restrict__ relc = d.rel + S.ch_relptr[c];
            for (int i = lane; i < uc; i += 64) S.rel[S.ch_rel[c] + i] = relc[i];
        }
        __syncthreads();
        // this wave's items: the columns of every child whose target column it owns, compacted by ballot / prefix count
        int cnt = 0;
        bool tall = false;                                        // a child with more than 64 update rows (second pass over its rows)
        for (int c = 0; c < n; c++) {
            const int uc = S.ch_uc[c], ldc = S.ch_ldc[c], ro = S.ch_rel[c];
            const int64_t off = S.ch_off[c];
            tall = tall || uc > 64;
            for (int base = 0; base < uc; base += 64) {
                const int l = base + lane;
                const int tj = l < uc ? S.rel[ro + l] : -1;
                const bool own = tj >= 0 && (tj & 3) == w;
                const unsigned long long mask = __ballot(own);
                if (own) S.items[w][cnt + __popcll(mask & ((1ull << lane) - 1ull))] =
                             make_uint2((unsigned)(off + (int64_t)l * ldc), (unsigned)tj | ((unsigned)uc << 7) | ((unsigned)ro << 14));
                cnt += __popcll(mask);
            }
            const int pad = (4 - (cnt & 3)) & 3;
            if (lane < pad) S.items[w][cnt + lane] = make_uint2(0u, 0u);
            cnt += pad;
        }
        wp_wave_sync();
        for (int it = 0; it < cnt; it += 16) wp_run_items(d, S, Fm, ld, w, lane, it, cnt, 0);
        if (tall)
            for (int it = 0; it < cnt; it += 16) wp_run_items(d, S, Fm, ld, w, lane, it, cnt, 64);
        wp_wave_sync();
        c0 += n;
    }
    __syncthreads();
}

// LDS of a workgroup: the front, 16 T rows x (16 T + 2) (every tile access of the update stays inside it whatever m), one dump slot
constexpr size_t wp_lds_bytes(int T) { return ((size_t)(16 * T) * (16 * T + 2) + 2) * sizeof(double); }

template <int T>
__global__ __launch_bounds__(256, (T <= 4 ? 2 : 1)) void k_lu_front_wp(const LuDev d, const int32_t *__restrict__ list, const double *__restrict__ Ax,
                                                      const double tol, const double stol, const int reuse)
{
    constexpr int R = T > 4 ? 2 : 1;
    constexpr int DUMP = 16 * T * (16 * T + 2);
    extern __shared__ double smem[];
    __shared__ __attribute__((aligned(16))) double L11[LU_PB * LU_PB + 8 * LU_PB];   // the panel's rows of the pivot block in logical order: L11[j * 32 + j2]
    __shared__ __attribute__((aligned(16))) double bcast[LU_PB];
    __shared__ int32_t sh_slot[16 * T], sh_prow[16 * T], sh_piv[16 * T];
    __shared__ WpAsmLds asm_lds;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int f = list[blockIdx.x];
    const LuFrontD F = d.fr[f];
    const int m = F.m, k = F.k, u = m - k;
    const int ld = (m + 1) | 1;
    double *Fm = smem;
#ifdef KVX_LU_PHASE
    unsigned long long lu_t_ = wall_clock64();
#endif
    if (tid == 0) d.fail[f] = 0;
    for (int t = tid; t < 16 * T; t += 256) {
        if (reuse && t < k) {                                     // the recorded sequence: pivot t is physical row lperm[t]
            const int pr = d.lperm[F.p0 + t];
            sh_prow[t] = pr; sh_slot[pr] = t; sh_piv[t] = d.ipiv[F.p0 + t];
        } else {
            if (!reuse || t >= k) sh_slot[t] = t;
            if (t >= k || !reuse) { sh_prow[t] = t; sh_piv[t] = 0; }
        }
    }
    wp_assemble(d, F, Fm, ld, Ax, asm_lds LU_TARG);
    LU_STAMP(0);
    double lmax = 0.0;
    const int tx = tid & 15, ty = tid >> 4;
    for (int j0 = 0; j0 < k; j0 += LU_PB) {
        const int nb = min(LU_PB, k - j0);
        // ---- (a) the panel: wave 0, registers
        if (wv == 0) {
            WpCtx x;
            x.j0 = j0; x.nb = nb; x.k = k; x.lane = lane; x.sh_piv = sh_piv; x.sh_prow = sh_prow; x.bc = bcast;
            x.fail_slot = d.fail + f; x.tol = tol; x.lmax = lmax;
            if (reuse) wp_panel<R, true>(Fm, ld, m, x, sh_slot, L11);
            else wp_panel<R, false>(Fm, ld, m, x, sh_slot, L11);
            lmax = x.lmax;
        }
        __syncthreads();
        LU_STAMP(1);
        if (j0 + nb >= m) break;                                  // nothing right of the panel
        // ---- (b) U12 = L11^-1 A12: one column per thread
        {
            const int c = j0 + nb + tid;
            if (c < m) wp_solve(Fm, ld, c, j0, nb, sh_prow, L11);
        }
        __syncthreads();
        LU_STAMP(2);
        // ---- (c) rank-nb update of the rows not used yet, columns right of the panel: thread (tx, ty) owns rows tx + 16 a, columns
        // ty + 16 b.  Entries outside the region are computed along on whatever the LDS holds there and not stored: no predicate
        // in the loop (a tile element depends on its own row and column operands only).
        {
            const int cfirst = j0 + nb;
            double acc[T][T];
            bool rowon[T], colon[T];
#pragma unroll
            for (int bc = 0; bc < T; bc++) { const int c = ty + 16 * bc; colon[bc] = c >= cfirst && c < m; }
#pragma unroll
            for (int ar = 0; ar < T; ar++) {
                const int i = tx + 16 * ar;
                rowon[ar] = i < m && sh_slot[min(i, 16 * T - 1)] >= cfirst;
#pragma unroll
#ifdef WP_UPDATE_MASKED
                for (int bc = 0; bc < T; bc++) acc[ar][bc] = (rowon[ar] && colon[bc]) ? Fm[i + (ty + 16 * bc) * ld] : 0.0;
#else
                for (int bc = 0; bc < T; bc++) acc[ar][bc] = Fm[i + (ty + 16 * bc) * ld];
#endif
            }
            int prr = lane < nb ? sh_prow[j0 + lane] : 0;         // the panel's pivot rows, one per lane
            for (int jq = 0; jq < nb; jq += 4) {                  // four pivots per round: their operand reads are in flight together
                double lr[4][T], ub[4][T];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int pr = __builtin_amdgcn_readlane(prr, min(jq + q, nb - 1));
                    const double z = jq + q < nb ? 1.0 : 0.0;     // uniform
#pragma unroll
#ifdef WP_UPDATE_MASKED
                    for (int ar = 0; ar < T; ar++) lr[q][ar] = (z != 0.0 && rowon[ar]) ? Fm[tx + 16 * ar + (j0 + min(jq + q, nb - 1)) * ld] : 0.0;
#pragma unroll
                    for (int bc = 0; bc < T; bc++) ub[q][bc] = colon[bc] ? Fm[pr + (ty + 16 * bc) * ld] : 0.0;
#else
                    for (int ar = 0; ar < T; ar++) lr[q][ar] = z * Fm[tx + 16 * ar + (j0 + min(jq + q, nb - 1)) * ld];
#pragma unroll
                    for (int bc = 0; bc < T; bc++) ub[q][bc] = Fm[pr + (ty + 16 * bc) * ld];
#endif
                }
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int ar = 0; ar < T; ar++)
#pragma unroll
                        for (int bc = 0; bc < T; bc++) acc[ar][bc] -= lr[q][ar] * ub[q][bc];
            }
#pragma unroll
            for (int ar = 0; ar < T; ar++)
#pragma unroll
                for (int bc = 0; bc < T; bc++) Fm[(rowon[ar] && colon[bc]) ? tx + 16 * ar + (ty + 16 * bc) * ld : DUMP] = acc[ar][bc];
        }
        __syncthreads();
        LU_STAMP(3);
    }
    // |pivot| >= stol * max|column|  <=>  max|multiplier| <= 1 / stol : one test per front instead of one per pivot
    if (wv == 0 && !(lmax * stol <= 1.0)) atomicMax(d.fail + f, 1);
    // ---- store: panels (L(:, 0:k) as is, U(0:k, :) transposed, both m x k column-major), in-front permutation, update matrix
    for (int t = tid; t < k; t += 256) {
        d.lperm[F.p0 + t] = sh_prow[t];
        if (!reuse) d.ipiv[F.p0 + t] = sh_piv[t];
    }
    double *__restrict__ Lp = d.Lx + F.px, *__restrict__ Up = d.Ux + F.px;
    for (int j = wv; j < k; j += 4)
        for (int i = lane; i < m; i += 64) Lp[i + (int64_t)j * m] = Fm[(i < k ? sh_prow[i] : i) + j * ld];
    for (int t = wv; t < k; t += 4) {
        const int pr = sh_prow[t];
        for (int c = lane; c < m; c += 64) Up[c + (int64_t)t * m] = Fm[pr + c * ld];
    }
    double *__restrict__ Uo = d.arena + F.upd_off;
    for (int j = wv; j < u; j += 4)
        for (int i = lane; i < u; i += 64) Uo[i + (int64_t)j * u] = Fm[(k + i) + (k + j) * ld];
    LU_STAMP(4);
}

#ifdef KVX_LU_PHASE
extern "C" int kvx_dbg_lu_phase_read(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lu_phase), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_lu_phase), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif

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
    // blocks of 32 pivots: the first wavefront solves the 32 x 32 triangle in registers (pivot values travel by
    // v_readlane, no barrier inside), then every thread applies the block to its rows below: two barriers per block
    __shared__ double ysh[LU_SB];
    for (int t0 = 0; t0 < k; t0 += LU_SB) {
        const int nbk = min(LU_SB, k - t0);
        if (tid < 64) {
            const bool on = tid < nbk;
            double fi = on ? fv[t0 + tid] : 0.0, rd = 1.0;
            double lb[LU_SB];
#pragma unroll
            for (int t = 0; t < LU_SB; t++) lb[t] = (on && t < tid) ? panel[(t0 + tid) + (int64_t)(t0 + t) * m] : 0.0;
            if (!UNIT && on) rd = 1.0 / panel[(t0 + tid) + (int64_t)(t0 + tid) * m];
#pragma unroll
            for (int t = 0; t < LU_SB; t++) {
                double yt = readlane_d(fi, t);
                if (!UNIT) yt *= readlane_d(rd, t);
                fi -= lb[t] * yt;
            }
            if (on) { const double y = UNIT ? fi : fi * rd; ysh[tid] = y; fv[t0 + tid] = y; }
        }
        __syncthreads();
        for (int i = t0 + nbk + tid; i < m; i += NT) {
            double acc = 0.0;
            for (int t = 0; t < nbk; t++) acc += panel[i + (int64_t)(t0 + t) * m] * ysh[t];
            fv[i] -= acc;
        }
        __syncthreads();
    }
    for (int t = tid; t < k; t += NT) x[F.p0 + t] = fv[t];
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
    __shared__ double ysh[LU_SB];
    for (int t0 = ((k - 1) / LU_SB) * LU_SB; t0 >= 0; t0 -= LU_SB) {      // blocks of 32 pivots, last block first (see k_lu_fwd)
        const int nbk = min(LU_SB, k - t0);
        if (tid < 64) {
            const bool on = tid < nbk;
            double ws = on ? fv[t0 + tid] : 0.0, rd = 1.0;
            double cb[LU_SB];
#pragma unroll
            for (int t = 0; t < LU_SB; t++) cb[t] = (on && t > tid && t < nbk) ? panel[(t0 + t) + (int64_t)(t0 + tid) * m] : 0.0;
            if (!UNIT && on) rd = 1.0 / panel[(t0 + tid) + (int64_t)(t0 + tid) * m];
#pragma unroll
            for (int t = LU_SB - 1; t >= 0; t--) {
                double yt = readlane_d(ws, t);
                if (!UNIT) yt *= readlane_d(rd, t);
                ws -= cb[t] * yt;
            }
            if (on) { const double y = UNIT ? ws : ws * rd; ysh[tid] = y; g[t0 + tid] = y; }
        }
        __syncthreads();
        for (int s2 = tid; s2 < t0; s2 += NT) {
            const double *__restrict__ colp = panel + t0 + (int64_t)s2 * m;
            double acc = 0.0;
            for (int t = 0; t < nbk; t++) acc += colp[t] * ysh[t];
            fv[s2] -= acc;
        }
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


// ---------------------------------------------------------------------------------------------------------------
// Blocked path for fronts that do not fit in LDS: the front stays in its m x m region of the arena and is factored
// by column blocks of LU_NB pivots, three launches per block over ALL big fronts of the level:
//   k_lub_panel  one workgroup per front: the (m - jb) x nb panel (staged in LDS when it fits) with pivoting,
//   k_lub_trsm   one thread per remaining column: the block's row interchanges, then U12 = L11^-1 A12,
//   k_lub_gemm   64 x 64 tiles: A22 -= L21 U12.
constexpr int LU_NB = 32;
constexpr int LU_PANEL_LDS_DOUBLES = 19456;       // 152 KB of the 160 KB a gfx950 workgroup may use: the panel is staged in LDS when (m - jb) * nb fits

// Assembly of a big front, one workgroup per 16 columns: zero them, scatter the entries of A that land there, pull
// the children's update-matrix columns that map there (children one after the other: their targets overlap).
// first index of the ascending list a[0..n) whose value is >= v (n if none), by the 64 lanes of a wavefront together: 64 probes per
// round (two or three dependent loads for a list of thousands, where a bisection by one lane is eleven).  Uniform arguments, every
// lane of the wave calls it.
__device__ __forceinline__ int lu_lower_bound(const int32_t *__restrict__ a, int n, int v)
{
    const int lane = threadIdx.x & 63;
    int lo = 0, hi = n;                                       // the answer is in [lo, hi]
    while (hi - lo > 0) {
        const int len = hi - lo, step = (len + 63) >> 6;      // probes at lo + step * lane (< hi)
        const int idx = lo + step * lane;
        const bool lt = idx < hi && a[idx] < v;
        const int cnt = __popcll(__ballot(lt));               // ascending: the probes with a < v are the first cnt
        if (cnt == 0) { hi = lo; break; }                     // a[lo] >= v
        const int last = lo + step * (cnt - 1);               // a[last] < v, and the next probe (if any) is >= v
        lo = last + 1;
        hi = min(hi, last + step);
    }
    return lo;
}
// Round 3: the rows of the 16 columns are split over grid.z in chunks of LU_ASM_ROWS -- a workgroup with 16 whole columns of a front of
// a few thousand rows moved ~1 MB through one CU (~25 GB/s): 63 us per level for work the machine does in 10.
constexpr int LU_ASM_COLS = 16, LU_ASM_ROWS = 256;
__global__ __launch_bounds__(256) void k_lub_assemble(const LuDev d, const int32_t *__restrict__ list, const double *__restrict__ Ax)
{
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int f = list[blockIdx.y];
    const LuFrontD F = d.fr[f];
    const int m = F.m;
    const int c0 = blockIdx.x * LU_ASM_COLS, r0 = blockIdx.z * LU_ASM_ROWS;
    if (c0 >= m || r0 >= m) return;
    const int c1 = min(m, c0 + LU_ASM_COLS), r1 = min(m, r0 + LU_ASM_ROWS);
    double *Fm = d.arena + (F.upd_off - F.k - (int64_t)F.k * m);
    if (blockIdx.x == 0 && blockIdx.z == 0 && tid == 0) d.fail[f] = 0;
    for (int c = c0 + ty; c < c1; c += 4)
        for (int i = r0 + tx; i < r1; i += 64) Fm[i + (int64_t)c * m] = 0.0;
    __syncthreads();
    {   // the entries of A in these columns: the front's list ascends by destination (lu_symbolic.cpp) -- bisected, not scanned
        const int32_t *__restrict__ ad = d.a_dst + F.aptr;
        const int e0 = lu_lower_bound(ad, (int)F.acnt, c0 * m), e1 = lu_lower_bound(ad, (int)F.acnt, c1 * m);
        for (int e = e0 + tid; e < e1; e += 256) {
            const int dst = ad[e];
            const int row = dst % m;
            if (row >= r0 && row < r1) {
                const int64_t src = d.a_src[F.aptr + e];
                Fm[dst] += Ax[src] * d.rinv[d.ai32[src]];
            }
        }
    }
    __syncthreads();
    for (int c = 0; c < F.nchild; c++) {
        const LuFrontD C = d.fr[d.children[F.childptr + c]];
        const int uc = C.m - C.k, ldc = C.upd_ld;
        const int32_t *__restrict__ relc = d.rel + C.rowptr + C.k;
        const double *__restrict__ Uc = d.arena + C.upd_off;
        // the child's rows / columns that land in this workgroup's columns [c0, c1) and rows [r0, r1): relc ascends, so both are
        // ranges -- found by bisection (a scan of all uc columns with a dependent index load each was 20 us of latency per workgroup)
        const int j0 = lu_lower_bound(relc, uc, c0), j1 = lu_lower_bound(relc, uc, c1);
        const int i0 = lu_lower_bound(relc, uc, r0), i1 = lu_lower_bound(relc, uc, r1);
        for (int jc = j0 + ty; jc < j1; jc += 4) {
            const int pc = relc[jc];
            for (int ic = i0 + tx; ic < i1; ic += 64) Fm[relc[ic] + (int64_t)pc * m] += Uc[ic + (int64_t)jc * ldc];
        }
        __syncthreads();
    }
}

// Panels of a factored big front, one workgroup per 64 front columns; the first workgroup also turns the interchange
// sequence into the in-front permutation.
// first column right of the pivot block that pivot t belongs to (the caller caps it at the front's k: a front's last block is
// shorter), for a level whose tallest front has max_m rows: the blocks
// are those of launch_lu_big_level (8 pivots while more than 2048 rows remain, 16 down to 1024, then 32)
__device__ __forceinline__ int lub_block_end(const int t, const int max_m)
{
    const int jb1 = max_m > 2048 ? (max_m - 2048 + 7) / 8 * 8 : 0;
    const int jb2 = max_m - jb1 > 1024 ? jb1 + (max_m - jb1 - 1024 + 15) / 16 * 16 : jb1;
    if (t < jb1) return t / 8 * 8 + 8;
    if (t < jb2) return jb1 + (t - jb1) / 16 * 16 + 16;
    return jb2 + (t - jb2) / 32 * 32 + 32;
}

__global__ __launch_bounds__(256) void k_lub_store(const LuDev d, const int32_t *__restrict__ list, const int max_m)
{
    extern __shared__ double smem[];
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const LuFrontD F = d.fr[list[blockIdx.y]];
    const int m = F.m, k = F.k;
    // Round 3: grid.z = chunks of 256 rows (of the L columns) / 256 pivot rows (of the U rows), and the U rows go through an LDS
    // tile -- they are a transposition (Up[c + t m] = Fm[t + c m]), read with a stride of m before: 70 us per level, per-CU bandwidth.
    __shared__ double tile[64][65];
    const int c0 = blockIdx.x * 64, z0 = blockIdx.z * 256;
    if (c0 >= m || z0 >= m) return;
    const double *Fm = d.arena + (F.upd_off - k - (int64_t)k * m);
    double *__restrict__ Lp = d.Lx + F.px, *__restrict__ Up = d.Ux + F.px;
    for (int t0 = z0; t0 < min(k, z0 + 256); t0 += 64) {      // (workgroup-uniform bounds)
        for (int cc = ty; cc < 64; cc += 4)                   // tile[cc][tt] = Fm[t0 + tt + (c0 + cc) m]: lanes along the rows
            tile[cc][tx] = (t0 + tx < k && c0 + cc < m) ? Fm[(t0 + tx) + (int64_t)(c0 + cc) * m] : 0.0;
        __syncthreads();
        for (int tt = ty; tt < 64; tt += 4)                   // Up[c + t m]: lanes along the columns
            if (t0 + tt < k && c0 + tx < min(k, lub_block_end(t0 + tt, max_m))) Up[(c0 + tx) + (int64_t)(t0 + tt) * m] = tile[tx][tt];   // (right of the block: k_lub_gemm)
        __syncthreads();
    }
    for (int cc = c0 + ty; cc < min(k, c0 + 64); cc += 4)
        for (int i = z0 + tx; i < min(m, z0 + 256); i += 64) Lp[i + (int64_t)cc * m] = Fm[i + (int64_t)cc * m];
    if (blockIdx.x == 0 && blockIdx.z == 0) {
        int32_t *sh_lp = (int32_t *)smem, *sh_piv = sh_lp + k;
        for (int t = tid; t < k; t += 256) { sh_lp[t] = t; sh_piv[t] = d.ipiv[F.p0 + t]; }
        __syncthreads();
        // the interchanges in order -- only the pivots that moved a row (few): wave 0 finds them 64 at a time by a ballot, its first
        // lane applies them (a loop of one thread over all k pivots was 30 us of this kernel on a front of 757)
        if (tid < 64) {
            for (int base = 0; base < k; base += 64) {
                unsigned long long mv = __ballot(base + tid < k && sh_piv[base + tid] != 0);
                if (tid == 0)
                    while (mv) {
                        const int j = base + __builtin_ctzll(mv);
                        const int r = j + sh_piv[j];
                        const int a = sh_lp[j]; sh_lp[j] = sh_lp[r]; sh_lp[r] = a;
                        mv &= mv - 1;
                    }
            }
        }
        __syncthreads();
        for (int t = tid; t < k; t += 256) d.lperm[F.p0 + t] = sh_lp[t];
    }
}

__global__ __launch_bounds__(LU_NT_LDS) void k_lub_panel(const LuDev d, const int32_t *__restrict__ list, const int jb, const int nbs,
                                                          const double tol, const double stol, const int reuse)
{
    extern __shared__ double smem[];
    __shared__ int sh_i[2];
    __shared__ double sh_d[2];
    constexpr int NT = LU_NT_LDS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int f = list[blockIdx.x];
    const LuFrontD F = d.fr[f];
    const int m = F.m, k = F.k;
    if (jb >= k) return;
    const int nbk = min(nbs, k - jb), rows = m - jb, cand = k - jb;
    if (rows <= 4096) return;                                // handled by k_lub_panel_reg
    double *Fm = d.arena + (F.upd_off - k - (int64_t)k * m);
    double *G = Fm + jb + (int64_t)jb * m;                   // panel in the front, ld = m
    const bool staged = (int64_t)rows * nbk <= LU_PANEL_LDS_DOUBLES;
    double *Pm = staged ? smem : G;
    const int pld = staged ? rows : m;
    if (staged) {
        for (int idx = tid; idx < rows * nbk; idx += NT) { const int i = idx % rows, c = idx / rows; Pm[i + c * pld] = G[i + (int64_t)c * m]; }
        __syncthreads();
    }
    int32_t *ipiv = d.ipiv + F.p0 + jb;
    __shared__ int32_t sh_piv[LU_NB];
    if (reuse && tid < nbk) sh_piv[tid] = ipiv[tid];
    if (reuse) __syncthreads();
    bool failed = false;
    for (int j = 0; j < nbk; j++) {
        if (tid < 64) {
            double bmax = -1.0, amax = 0.0;
            int bidx = j;
            for (int i = j + lane; i < rows; i += 64) {
                const double a = fabs(Pm[i + (int64_t)j * pld]);
                if (i < cand && a > bmax) { bmax = a; bidx = i; }
                amax = fmax(amax, a);
            }
            for (int off = 32; off; off >>= 1) {
                const double ob = __shfl_xor(bmax, off), oa = __shfl_xor(amax, off);
                const int oi = __shfl_xor(bidx, off);
                if (ob > bmax || (ob == bmax && oi < bidx)) { bmax = ob; bidx = oi; }
                amax = fmax(amax, oa);
            }
            if (lane == 0) {
                const double diag = fabs(Pm[j + (int64_t)j * pld]);
                int r = (diag > 0.0 && diag >= tol * bmax) ? j : bidx;
                if (reuse) r = j + sh_piv[j];
                double pv = Pm[r + (int64_t)j * pld];
                const double ap = fabs(pv);
                const bool bad = !(ap > 0.0) || !(ap <= 1.7e308) || ap < stol * amax;
                if (bad && !failed && d.fail[f] == 0) { failed = true; d.fail[f] = jb + j + 1; }
                if (!(ap > 0.0) || !(ap <= 1.7e308)) pv = 1.0;
                sh_piv[j] = r - j;
                sh_i[0] = r;
                sh_d[0] = pv;
            }
        }
        __syncthreads();
        const int r = sh_i[0];
        const double pv = sh_d[0];
        if (r != j && tid < nbk) {
            const double a = Pm[j + (int64_t)tid * pld], b = Pm[r + (int64_t)tid * pld];
            Pm[j + (int64_t)tid * pld] = b;
            Pm[r + (int64_t)tid * pld] = a;
        }
        __syncthreads();
        for (int i = j + 1 + tid; i < rows; i += NT) Pm[i + (int64_t)j * pld] /= pv;
        __syncthreads();
        const int tx = tid & 63, ty = tid >> 6;
        for (int c = j + 1 + ty; c < nbk; c += NT / 64) {
            const double ujc = Pm[j + (int64_t)c * pld];
            if (ujc != 0.0)
                for (int i = j + 1 + tx; i < rows; i += 64) Pm[i + (int64_t)c * pld] -= Pm[i + (int64_t)j * pld] * ujc;
        }
        __syncthreads();
    }
    if (!reuse && tid < nbk) ipiv[tid] = sh_piv[tid];
    if (staged)
        for (int idx = tid; idx < rows * nbk; idx += NT) { const int i = idx % rows, c = idx / rows; G[i + (int64_t)c * m] = Pm[i + c * pld]; }
}


// Register panel: ONE THREAD PER PANEL ROW (rows <= 1024), the row's nb entries in registers.  No data moves when a
// pivot is chosen: every thread carries the logical position (`slot`) its row currently has under the LAPACK-style
// interchange sequence; the pivot row publishes itself in LDS, everyone below eliminates with it.  Two barriers per
// pivot; the row lands at its logical position when the panel is written back (identical to applying the
// interchanges, which k_lub_trsm does for the other columns).  The pivot steps are expanded by template recursion so
// that every index into the row is a compile-time constant (a `#pragma unroll` loop left the row in scratch memory).
// Wave-wide max of a double / min of an int by DPP row operations (no LDS round trips: a __shfl_xor butterfly over 64 lanes is six
// dependent ds_bpermute rounds of three permutes each -- 1400-1900 cycles of the 4200 a pivot step of the register panel took).
// row_shr 1, 2, 4, 8 leave the row's result in its last lane, row_bcast 15 / 31 carry it across the rows; lane 63 holds the wave's.
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ double dpp_max_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWS, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWS, 0xf, false);
    return fmax(v, __hiloint2double(hi, lo));
}
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ int dpp_min_i32(int v) { return min(v, __builtin_amdgcn_update_dpp(v, v, CTRL, ROWS, 0xf, false)); }
__device__ __forceinline__ double wave_max_f64(double v)
{
    v = dpp_max_f64<0x111>(v); v = dpp_max_f64<0x112>(v); v = dpp_max_f64<0x114>(v); v = dpp_max_f64<0x118>(v);
    v = dpp_max_f64<0x142, 0xa>(v); v = dpp_max_f64<0x143, 0xc>(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ int wave_min_i32(int v)
{
    v = dpp_min_i32<0x111>(v); v = dpp_min_i32<0x112>(v); v = dpp_min_i32<0x114>(v); v = dpp_min_i32<0x118>(v);
    v = dpp_min_i32<0x142, 0xa>(v); v = dpp_min_i32<0x143, 0xc>(v);
    return __builtin_amdgcn_readlane(v, 63);
}
// the largest value of the wave and, among the lanes that hold it, the smallest slot (the rule of the butterfly it replaces)
__device__ __forceinline__ void wave_argmax(double &bmax, int &bslot)
{
    const double vm = wave_max_f64(bmax);
    bslot = wave_min_i32(bmax == vm ? bslot : 0x7fffffff);
    bmax = vm;
}

template <int RPT>
struct PanelCtx {
    double *sh_b, *urow, *sh_diag, *sh_rinv;
    int *sh_s, *sh_piv;
    int32_t *fail_slot;
    double tol, lmax;
    int lane, wave, ncw, nbk, cand, reuse, jb;
    int slot[RPT];
    bool has[RPT], failed;
};

template <int J, int NB, int RPT>
__device__ __forceinline__ void panel_step(double (&a)[RPT][NB], PanelCtx<RPT> &x)
{
    if constexpr (J < NB) {
        if (J < x.nbk) {                                          // uniform
            double *sh_b = x.sh_b + (J & 1) * 16, *sh_diag = x.sh_diag + (J & 1);
            int *sh_s = x.sh_s + (J & 1) * 16;                    // (the search results are double-buffered for the same reason)
            double *uspec = x.urow + (J & 1) * NB;                // the row at position J publishes itself SPECULATIVELY (two
            if (x.wave < x.ncw) {                                 // buffers, by parity of J): when the diagonal is kept -- the
                double bmax = -1.0;                               // common case -- the step needs one barrier, not two
                int bslot = 0x7fffffff;
#pragma unroll
                for (int q = 0; q < RPT; q++) {
                    const bool c = x.has[q] && x.slot[q] >= J && x.slot[q] < x.cand;
                    const double av = fabs(a[q][J]);
                    if (c && (av > bmax || (av == bmax && x.slot[q] < bslot))) { bmax = av; bslot = x.slot[q]; }
                    if (x.has[q] && x.slot[q] == J) {
                        sh_diag[0] = av;
                        x.sh_rinv[J & 1] = 1.0 / a[q][J];         // the reciprocal travels with the row: one division per pivot, not one per row
#pragma unroll
                        for (int c2 = J; c2 < NB; c2++) uspec[c2] = a[q][c2];      // (columns left of the pivot are not read again)
                    }
                }
                wave_argmax(bmax, bslot);
                if (x.lane == 0) { sh_b[x.wave] = bmax; sh_s[x.wave] = bslot; }
            }
            __syncthreads();
            // the waves' results: lane w reads wave w's, one more wave-wide reduction (every wave for itself)
            double bmax = x.lane < x.ncw ? sh_b[x.lane & 15] : -1.0;
            int bslot = x.lane < x.ncw ? sh_s[x.lane & 15] : 0x7fffffff;
            wave_argmax(bmax, bslot);
            const double dg = sh_diag[0];
            int r = (dg > 0.0 && dg >= x.tol * bmax) ? J : bslot;
            if (x.reuse) r = J + x.sh_piv[J];
            if (r == 0x7fffffff) r = J;                           // nothing but NaNs: keep the diagonal, the step is flagged below
            if (r != J) {                                         // uniform: an interchange -- the chosen row replaces the speculation
                __syncthreads();                                  // (everybody has read sh_b / sh_diag; nobody reads uspec yet)
#pragma unroll
                for (int q = 0; q < RPT; q++) {
                    if (x.has[q] && x.slot[q] == r) {
                        x.sh_rinv[J & 1] = 1.0 / a[q][J];
#pragma unroll
                        for (int c = J; c < NB; c++) uspec[c] = a[q][c];
                        x.slot[q] = J;
                    } else if (x.has[q] && x.slot[q] == J) x.slot[q] = r;
                }
                __syncthreads();
            }
            double pv = uspec[J];
            const double ap = fabs(pv);
            const bool bad = !(ap > 0.0) || !(ap <= 1.7e308);
            if (bad) pv = 1.0;
            const double rpv = bad ? 1.0 : x.sh_rinv[J & 1];
            if (threadIdx.x == 0) {
                if (bad && !x.failed && *x.fail_slot == 0) { x.failed = true; *x.fail_slot = x.jb + J + 1; }
                x.sh_piv[J] = r - J;
            }
#pragma unroll
            for (int q = 0; q < RPT; q++) {
                if (x.has[q] && x.slot[q] > J) {
                    const double l = a[q][J] * rpv;
                    a[q][J] = l;
                    x.lmax = fmax(x.lmax, fabs(l));
#pragma unroll
                    for (int c = J + 1; c < NB; c++) a[q][c] -= l * uspec[c];
                }
            }
        }
        panel_step<J + 1, NB, RPT>(a, x);
    }
}

// A refactorisation (reuse: the recorded interchanges) knows where every row ends before the panel starts: slot = final position,
// the row with slot J is pivot row J, rows with a larger slot are the ones still to be eliminated.  No search, no reduction
// across the waves, one barrier per pivot (the publishing buffers alternate by parity of J).
template <int J, int NB, int RPT>
__device__ __forceinline__ void panel_step_reuse(double (&a)[RPT][NB], PanelCtx<RPT> &x)
{
    if constexpr (J < NB) {
        if (J < x.nbk) {                                          // uniform
            double *uspec = x.urow + (J & 1) * NB;
#pragma unroll
            for (int q = 0; q < RPT; q++)
                if (x.slot[q] == J) {
                    x.sh_rinv[J & 1] = 1.0 / a[q][J];
#pragma unroll
                    for (int c2 = J; c2 < NB; c2++) uspec[c2] = a[q][c2];
                }
            __syncthreads();
            const double pv = uspec[J];
            const double ap = fabs(pv);
            const bool bad = !(ap > 0.0) || !(ap <= 1.7e308);
            const double rpv = bad ? 1.0 : x.sh_rinv[J & 1];
            if (bad && threadIdx.x == 0 && !x.failed && *x.fail_slot == 0) { x.failed = true; *x.fail_slot = x.jb + J + 1; }
#pragma unroll
            for (int q = 0; q < RPT; q++) {
                if (x.has[q] && x.slot[q] > J) {
                    const double l = a[q][J] * rpv;
                    a[q][J] = l;
                    x.lmax = fmax(x.lmax, fabs(l));
#pragma unroll
                    for (int c = J + 1; c < NB; c++) a[q][c] -= l * uspec[c];
                }
            }
        }
        panel_step_reuse<J + 1, NB, RPT>(a, x);
    }
}

// The acceptance test |pivot| >= stol * max|column| of lu_factor_front is applied afterwards in its equivalent form
// max|multiplier| <= 1 / stol (one reduction per panel instead of one per pivot).  NB pivots per panel, RPT rows per
// thread: <32, 1> up to 1024 rows, <16, 2> up to 2048, <8, 4> up to 4096 (the register budget of a 1024-thread workgroup).
template <int NB, int RPT, bool REUSE>
__global__ __launch_bounds__(1024) void k_lub_panel_reg(const LuDev d, const int32_t *__restrict__ list, const int jb, const double tol,
                                                         const double stol)
{
    constexpr int reuse = REUSE ? 1 : 0;
    __shared__ double sh_b[32], sh_diag[2], urow[2 * NB], sh_rinv[2];
    __shared__ int sh_s[32], sh_piv[NB];
    const int tid = threadIdx.x, nth = blockDim.x;
    const int f = list[blockIdx.x];
    const LuFrontD F = d.fr[f];
    const int m = F.m, k = F.k;
    if (jb >= k) return;
    const int nbk = min(NB, k - jb), rows = m - jb;
    if (rows > RPT * nth) return;                             // handled by k_lub_panel
    double *G = d.arena + (F.upd_off - k - (int64_t)k * m) + jb + (int64_t)jb * m;
    int32_t *ipiv = d.ipiv + F.p0 + jb;
    if (reuse && tid < nbk) sh_piv[tid] = ipiv[tid];
    PanelCtx<RPT> x;
    x.sh_b = sh_b; x.sh_diag = sh_diag; x.urow = urow; x.sh_s = sh_s; x.sh_piv = sh_piv; x.sh_rinv = sh_rinv;
    x.fail_slot = d.fail + f; x.tol = tol; x.lmax = 0.0;
    x.lane = tid & 63; x.wave = tid >> 6; x.nbk = nbk; x.cand = k - jb; x.reuse = reuse; x.jb = jb;
    x.ncw = RPT > 1 ? (nth >> 6) : (min(x.cand, rows) + 63) >> 6;     // with several rows per thread every wave may hold candidates
    x.failed = false;
    double a[RPT][NB];
#pragma unroll
    for (int q = 0; q < RPT; q++) {
        const int row = tid + q * nth;
        x.has[q] = row < rows;
        x.slot[q] = x.has[q] ? row : 0x7fffffff;
#pragma unroll
        for (int c = 0; c < NB; c++) a[q][c] = (x.has[q] && c < nbk) ? G[row + (int64_t)c * m] : 0.0;
    }
    __syncthreads();
    if constexpr (REUSE) {
        // where the recorded interchanges take every row: follow them (32 steps of integer work, the sequence read as a broadcast)
#pragma unroll
        for (int q = 0; q < RPT; q++) {
            int pos = x.slot[q];
            for (int j = 0; j < nbk; j++) {
                const int r = j + sh_piv[j];
                pos = pos == j ? r : (pos == r ? j : pos);
            }
            x.slot[q] = pos;
        }
        panel_step_reuse<0, NB, RPT>(a, x);
    } else panel_step<0, NB, RPT>(a, x);
    double lm = x.lmax;
#pragma unroll
    for (int off = 32; off; off >>= 1) lm = fmax(lm, __shfl_xor(lm, off));
    __syncthreads();
    if (x.lane == 0) sh_b[x.wave] = lm;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < (nth >> 6); w++) lm = fmax(lm, sh_b[w]);
        if (!(lm * stol <= 1.0) && d.fail[f] == 0) d.fail[f] = jb + 1;
    }
    if (!reuse && tid < nbk) ipiv[tid] = sh_piv[tid];
#pragma unroll
    for (int q = 0; q < RPT; q++)
        if (x.has[q]) {
#pragma unroll
            for (int c = 0; c < NB; c++)
                if (c < nbk) G[x.slot[q] + (int64_t)c * m] = a[q][c];
        }
}

// The block's row interchanges, every column outside the panel (LAPACK's laswp; the panel itself was written back in order).
// Most blocks of a refactorisation of a diagonally heavy matrix have none: the launch ends there.
__global__ __launch_bounds__(64) void k_lub_trsm(const LuDev d, const int32_t *__restrict__ list, const int jb, const int nbs)
{
    const int tid = threadIdx.x;
    const LuFrontD F = d.fr[list[blockIdx.y]];
    const int m = F.m, k = F.k;
    const int c0 = blockIdx.x * 64;
    if (jb >= k || c0 >= m) return;
    const int nbk = min(nbs, k - jb);
    const int piv = tid < nbk ? d.ipiv[F.p0 + jb + tid] : 0;
    if (__ballot(piv != 0) == 0ull) return;
    double *Fm = d.arena + (F.upd_off - k - (int64_t)k * m);
    const int c = c0 + tid;
    if (c < m && !(c >= jb && c < jb + nbk)) {
        double *col = Fm + jb + (int64_t)c * m;
        for (int j = 0; j < nbk; j++) {
            const int r = j + __builtin_amdgcn_readlane(piv, j);
            if (r != j) { const double a = col[j], b = col[r]; col[j] = b; col[r] = a; }
        }
    }
}

// A22 -= L21 U12 on 64 x 64 tiles, rank LU_NB, on the FP64 matrix pipe (v_mfma_f64_16x16x4_f64): both operand tiles are staged in
// LDS (rows padded to 80 doubles: the four k-lanes of an operand read land on disjoint banks), wave w owns rows 16 w .. 16 w + 15 of
// the tile and all four column tiles -- 8 k-steps x 4 MFMAs.  Operand roles as in the Cholesky kernels (A operand <- the tile's
// columns, B operand <- its rows): the 16 lanes that share an accumulator register hold consecutive ROWS of one column of the
// column-major front, so the read-modify-write of A22 is coalesced.
constexpr int LU_GLD = 80;
typedef double lu_d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_lub_gemm(const LuDev d, const int32_t *__restrict__ list, const int jb, const int nbs)
{
    __shared__ double Ls[LU_NB * LU_GLD], Us[LU_NB * LU_GLD], L11[LU_NB * LU_NB];
    const int tid = threadIdx.x;
    const LuFrontD F = d.fr[list[blockIdx.z]];
    const int m = F.m, k = F.k;
    if (jb >= k) return;
    const int nbk = min(nbs, k - jb), t0 = jb + nbk;
    const int r0 = t0 + blockIdx.x * 64, c0 = t0 + blockIdx.y * 64;
    if (r0 >= m || c0 >= m) return;
    double *Fm = d.arena + (F.upd_off - k - (int64_t)k * m);
    for (int idx = tid; idx < LU_NB * LU_NB; idx += 256) {    // the panel's unit triangle (strictly lower part, zero padding)
        const int i = idx % LU_NB, t = idx / LU_NB;
        L11[idx] = (i < nbk && t < nbk && i > t) ? Fm[(jb + i) + (int64_t)(jb + t) * m] : 0.0;
    }
    for (int idx = tid; idx < LU_NB * 64; idx += 256) {
        const int i = idx & 63, t = idx >> 6;                 // L21 tile: rows r0.., block column t
        Ls[t * LU_GLD + i] = (t < nbk && r0 + i < m) ? Fm[(r0 + i) + (int64_t)(jb + t) * m] : 0.0;
    }
    for (int idx = tid; idx < LU_NB * 64; idx += 256) {
        const int t = idx % LU_NB, c = idx / LU_NB;           // U12 tile: block row t, columns c0..
        Us[t * LU_GLD + c] = (t < nbk && c0 + c < m) ? Fm[(jb + t) + (int64_t)(c0 + c) * m] : 0.0;
    }
    __syncthreads();
    // U12 = L11^-1 A12 for the tile's 64 columns, here rather than in a launch of its own (k_lub_trsm: 20 us a step on the chain
    // panel -> solve -> update; every tile row repeats the 32 x 64 solve -- 500 multiply-adds a thread of one wavefront -- and the
    // first one writes it back)
    if (tid < 64) {
        double v[LU_NB];
#pragma unroll
        for (int t = 0; t < LU_NB; t++) v[t] = Us[t * LU_GLD + tid];
#pragma unroll
        for (int t = 0; t < LU_NB; t++) {
#pragma unroll
            for (int s2 = t + 1; s2 < LU_NB; s2++) v[s2] -= L11[s2 + t * LU_NB] * v[t];
        }
#pragma unroll
        for (int t = 0; t < LU_NB; t++) Us[t * LU_GLD + tid] = v[t];
        // (not into the front: the other tile rows of this launch read the unsolved block from there.  Nothing reads these rows of
        //  the front again but k_lub_store, which leaves the entries right of a pivot's block alone: they go to their final place)
        if (blockIdx.x == 0 && c0 + tid < m) {
            double *__restrict__ Up = d.Ux + F.px;
#pragma unroll
            for (int t = 0; t < LU_NB; t++)
                if (t < nbk) Up[(c0 + tid) + (int64_t)(jb + t) * m] = v[t];
        }
    }
    __syncthreads();
    const int w = tid >> 6, l = tid & 63, lr = l & 15, lk = l >> 4;
    lu_d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = (lu_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < LU_NB; ks += 4) {
        const double bv = Ls[(ks + lk) * LU_GLD + 16 * w + lr];
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Us[(ks + lk) * LU_GLD + 16 * t + lr], bv, acc[t], 0, 0, 0);
    }
    // lane holds (L21 U12)[row 16 w + lr][column 16 t + lk + 4 q]
    const int i = r0 + 16 * w + lr;
    if (i < m) {
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int c = c0 + 16 * t + lk + 4 * q;
                if (c < m) Fm[i + (int64_t)c * m] -= acc[t][q];
            }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Triangular sweeps of fronts too large for one workgroup (m > KVX_LU_SOLVE_BIG_M): a single workgroup streams the
// panel at ~10 GB/s.  The front's work vector (m doubles, at W + wx - k) lives in HBM; one launch per block of 32
// pivots, every workgroup solves the 32 x 32 triangle redundantly in its first wavefront and applies the block to
// its own 1024 rows (forward) / columns (backward).
constexpr int LU_BIG_CHUNK = 256;         // rows (pivot columns in the backward steps) per workgroup: 1024 was ~256 KB through one CU per step, 15 us a launch

template <bool UNIT>
__device__ __forceinline__ void wave_tri_fwd(const double *__restrict__ panel, const int m, const int t0, const int nbk, const double *fvec,
                                             double *ysh)
{
    const int tid = threadIdx.x;
    const bool on = tid < nbk;
    double fi = on ? fvec[t0 + tid] : 0.0, rd = 1.0;
    double lb[LU_SB];
#pragma unroll
    for (int t = 0; t < LU_SB; t++) lb[t] = (on && t < tid) ? panel[(t0 + tid) + (int64_t)(t0 + t) * m] : 0.0;
    if (!UNIT && on) rd = 1.0 / panel[(t0 + tid) + (int64_t)(t0 + tid) * m];
#pragma unroll
    for (int t = 0; t < LU_SB; t++) {
        double yt = readlane_d(fi, t);
        if (!UNIT) yt *= readlane_d(rd, t);
        fi -= lb[t] * yt;
    }
    if (on) ysh[tid] = UNIT ? fi : fi * rd;
}

template <bool UNIT>
__device__ __forceinline__ void wave_tri_bwd(const double *__restrict__ panel, const int m, const int t0, const int nbk, const double *fvec,
                                             double *ysh)
{
    const int tid = threadIdx.x;
    const bool on = tid < nbk;
    double ws = on ? fvec[t0 + tid] : 0.0, rd = 1.0;
    double cb[LU_SB];
#pragma unroll
    for (int t = 0; t < LU_SB; t++) cb[t] = (on && t > tid && t < nbk) ? panel[(t0 + t) + (int64_t)(t0 + tid) * m] : 0.0;
    if (!UNIT && on) rd = 1.0 / panel[(t0 + tid) + (int64_t)(t0 + tid) * m];
#pragma unroll
    for (int t = LU_SB - 1; t >= 0; t--) {
        double yt = readlane_d(ws, t);
        if (!UNIT) yt *= readlane_d(rd, t);
        ws -= cb[t] * yt;
    }
    if (on) ysh[tid] = UNIT ? ws : ws * rd;
}

template <bool UNIT>
__global__ __launch_bounds__(LU_NT_SOLVE) void k_lu_fwd_big_init(const LuDev d, const int32_t *__restrict__ list, const double *__restrict__ X,
                                                                  const int64_t ldx, double *__restrict__ W, const int64_t wsize)
{
    extern __shared__ double smem[];
    constexpr int NT = LU_NT_SOLVE;
    const int tid = threadIdx.x;
    const LuFrontD F = d.fr[list[blockIdx.x]];
    const int m = F.m, k = F.k;
    const double *x = X + (int64_t)blockIdx.y * ldx;
    double *w = W + (int64_t)blockIdx.y * wsize;
    double *fvec = w + F.wx - k;
    for (int t = tid; t < m; t += NT) fvec[t] = t < k ? x[F.p0 + t] : 0.0;
    __syncthreads();
    for (int c = 0; c < F.nchild; c++) {
        const LuFrontD C = d.fr[d.children[F.childptr + c]];
        const int uc = C.m - C.k;
        const int32_t *__restrict__ relc = d.rel + C.rowptr + C.k;
        const double *__restrict__ wc = w + C.wx;
        for (int i = tid; i < uc; i += NT) fvec[relc[i]] += wc[i];
        __syncthreads();
    }
    if (UNIT) {
        for (int t = tid; t < k; t += NT) smem[t] = fvec[d.lperm[F.p0 + t]];
        __syncthreads();
        for (int t = tid; t < k; t += NT) fvec[t] = smem[t];
    }
}

template <bool UNIT>
__global__ __launch_bounds__(LU_NT_SOLVE) void k_lu_fwd_big_step(const LuDev d, const int32_t *__restrict__ list, double *__restrict__ X,
                                                                  const int64_t ldx, double *__restrict__ W, const int64_t wsize, const int t0)
{
    __shared__ double ysh[LU_SB];
    constexpr int NT = LU_NT_SOLVE;
    const int tid = threadIdx.x;
    const LuFrontD F = d.fr[list[blockIdx.y]];
    const int m = F.m, k = F.k;
    if (t0 >= k) return;
    const int nbk = min(LU_SB, k - t0), t1 = t0 + nbk;
    const int r0 = max(t1, (int)blockIdx.x * LU_BIG_CHUNK), r1 = min(m, ((int)blockIdx.x + 1) * LU_BIG_CHUNK);
    if (r0 >= r1 && blockIdx.x != 0) return;
    double *fvec = W + (int64_t)blockIdx.z * wsize + F.wx - k;
    const double *__restrict__ panel = (UNIT ? d.Lx : d.Ux) + F.px;
    if (tid < 64) wave_tri_fwd<UNIT>(panel, m, t0, nbk, fvec, ysh);
    __syncthreads();
    for (int i = r0 + tid; i < r1; i += NT) {
        double acc = 0.0;
        for (int t = 0; t < nbk; t++) acc += panel[i + (int64_t)(t0 + t) * m] * ysh[t];
        fvec[i] -= acc;
    }
    if (blockIdx.x == 0 && tid < nbk) X[(int64_t)blockIdx.z * ldx + F.p0 + t0 + tid] = ysh[tid];
}

// w(t) = x(p0 + t) - panel21(:, t)' * x(update rows): one wavefront per pivot column
template <bool UNIT>
__global__ __launch_bounds__(LU_NT_SOLVE) void k_lu_bwd_big_init(const LuDev d, const int32_t *__restrict__ list, const double *__restrict__ X,
                                                                  const int64_t ldx, double *__restrict__ W, const int64_t wsize)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const LuFrontD F = d.fr[list[blockIdx.y]];
    const int m = F.m, k = F.k;
    const int t = blockIdx.x * (LU_NT_SOLVE / 64) + wave;
    if (t >= k) return;
    const double *x = X + (int64_t)blockIdx.z * ldx;
    double *fvec = W + (int64_t)blockIdx.z * wsize + F.wx - k;
    const double *__restrict__ panel = (UNIT ? d.Lx : d.Ux) + F.px;
    const int32_t *__restrict__ rows = d.rowidx + F.rowptr;
    double s = 0.0;
    for (int i = k + lane; i < m; i += 64) s += panel[i + (int64_t)t * m] * x[rows[i]];
    for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) fvec[t] = x[F.p0 + t] - s;
}

template <bool UNIT>
__global__ __launch_bounds__(LU_NT_SOLVE) void k_lu_bwd_big_step(const LuDev d, const int32_t *__restrict__ list, double *__restrict__ X,
                                                                  const int64_t ldx, double *__restrict__ W, const int64_t wsize, const int t0)
{
    __shared__ double ysh[LU_SB];
    constexpr int NT = LU_NT_SOLVE;
    const int tid = threadIdx.x;
    const LuFrontD F = d.fr[list[blockIdx.y]];
    const int m = F.m, k = F.k;
    if (t0 >= k) return;
    const int nbk = min(LU_SB, k - t0);
    const int s0 = blockIdx.x * LU_BIG_CHUNK, s1 = min(t0, ((int)blockIdx.x + 1) * LU_BIG_CHUNK);
    if (s0 >= s1 && blockIdx.x != 0) return;
    double *fvec = W + (int64_t)blockIdx.z * wsize + F.wx - k;
    const double *__restrict__ panel = (UNIT ? d.Lx : d.Ux) + F.px;
    if (tid < 64) wave_tri_bwd<UNIT>(panel, m, t0, nbk, fvec, ysh);
    __syncthreads();
    for (int s2 = s0 + tid; s2 < s1; s2 += NT) {
        const double *__restrict__ colp = panel + t0 + (int64_t)s2 * m;
        double acc = 0.0;
        for (int t = 0; t < nbk; t++) acc += colp[t] * ysh[t];
        fvec[s2] -= acc;
    }
    if (blockIdx.x == 0 && tid < nbk)
        X[(int64_t)blockIdx.z * ldx + F.p0 + (UNIT ? d.lperm[F.p0 + t0 + tid] : t0 + tid)] = ysh[tid];
}


// ---- block triangular form: the off-diagonal blocks F of R P A Q are never eliminated; a block level subtracts their
// products with the parts of the solution already known before its own sweeps (KLU's block back substitution) ----------
__global__ void k_lu_fvals(const int64_t nf, const int64_t *__restrict__ src, const double *__restrict__ Ax, const double *__restrict__ rinv,
                           const int32_t *__restrict__ ai32, double *__restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nf) return;
    const int64_t p = src[e];
    out[e] = Ax[p] * rinv[ai32[p]];
}
// X[pos] -= sum_e val[e] * X[idx[e]] over the entries ptr[pos] .. ptr[pos + 1) for every listed position
__global__ void k_lu_fterm(const int64_t cnt, const int32_t *__restrict__ poslist, const int64_t *__restrict__ ptr,
                           const int32_t *__restrict__ idx, const double *__restrict__ val, double *__restrict__ X, const int64_t ldx)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= cnt) return;
    const int32_t pos = poslist[q];
    double *x = X + (int64_t)blockIdx.y * ldx;
    double acc = 0.0;
    for (int64_t e = ptr[pos]; e < ptr[pos + 1]; e++) acc += val[e] * x[idx[e]];
    if (ptr[pos + 1] > ptr[pos]) x[pos] -= acc;
}

}  // namespace

// gfx950: a workgroup may use all 160 KB of a CU's LDS; beyond 64 KB the kernel must be told so once.
static void allow_large_lds()
{
    static bool done = false;
    if (done) return;
    done = true;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lu_front<true, LU_NT_LDS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lu_front_tiled<7>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lu_front_wp<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lu_front_wp<7>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lub_panel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
}

void launch_lu_fronts(const LuDev &d, const int32_t *list, int cnt, int lds_m, int max_k, const double *Ax, double tol,
                      double stol, int reuse, hipStream_t st)
{
    if (cnt <= 0) return;
    allow_large_lds();
    if (lds_m > 0) {
        const size_t sm = (size_t)lds_m * lds_m * sizeof(double) + 2 * (size_t)lds_m * sizeof(int32_t);
        static const bool legacy = getenv("KVX_LU_LDS_LEGACY") != nullptr;       // the LDS-resident elimination (debugging aid)
        if (legacy) {
            hipLaunchKernelGGL((k_lu_front<true, LU_NT_LDS>), dim3(cnt), dim3(LU_NT_LDS), sm, st, d, list, Ax, tol, stol, reuse, lds_m);
            return;
        }
        static const bool wp = [] { const char *e = getenv("KVX_LU_WP"); return !e || e[0] != '0'; }();   // 0: the round-3 kernel (two barriers per pivot)
        // ... where a launch is a level's few dozen fronts (its time is that of its slowest front); thousands of fronts per launch are
        // bound by how many workgroups a CU holds, and the round-3 kernel is the smaller one (KVX_LU_WP_MAXCNT, default 512)
        static const int wp_maxcnt = [] { const char *e = getenv("KVX_LU_WP_MAXCNT"); return e ? atoi(e) : 512; }();
        if (wp && cnt <= wp_maxcnt && d.arena_size < (int64_t)1 << 32) {      // (its work items hold 32-bit offsets into the arena)
            switch ((lds_m + 15) / 16) {
            case 1: hipLaunchKernelGGL((k_lu_front_wp<1>), dim3(cnt), dim3(256), wp_lds_bytes(1), st, d, list, Ax, tol, stol, reuse); break;
            case 2: hipLaunchKernelGGL((k_lu_front_wp<2>), dim3(cnt), dim3(256), wp_lds_bytes(2), st, d, list, Ax, tol, stol, reuse); break;
            case 3: hipLaunchKernelGGL((k_lu_front_wp<3>), dim3(cnt), dim3(256), wp_lds_bytes(3), st, d, list, Ax, tol, stol, reuse); break;
            case 4: hipLaunchKernelGGL((k_lu_front_wp<4>), dim3(cnt), dim3(256), wp_lds_bytes(4), st, d, list, Ax, tol, stol, reuse); break;
            case 5: case 6: hipLaunchKernelGGL((k_lu_front_wp<6>), dim3(cnt), dim3(256), wp_lds_bytes(6), st, d, list, Ax, tol, stol, reuse); break;
            default: hipLaunchKernelGGL((k_lu_front_wp<7>), dim3(cnt), dim3(256), wp_lds_bytes(7), st, d, list, Ax, tol, stol, reuse); break;
            }
            return;
        }
        switch ((lds_m + 15) / 16) {
        case 1: hipLaunchKernelGGL((k_lu_front_tiled<1>), dim3(cnt), dim3(256), sm, st, d, list, Ax, tol, stol, reuse, lds_m); break;
        case 2: hipLaunchKernelGGL((k_lu_front_tiled<2>), dim3(cnt), dim3(256), sm, st, d, list, Ax, tol, stol, reuse, lds_m); break;
        case 3: hipLaunchKernelGGL((k_lu_front_tiled<3>), dim3(cnt), dim3(256), sm, st, d, list, Ax, tol, stol, reuse, lds_m); break;
        case 4: hipLaunchKernelGGL((k_lu_front_tiled<4>), dim3(cnt), dim3(256), sm, st, d, list, Ax, tol, stol, reuse, lds_m); break;
        case 5: case 6: hipLaunchKernelGGL((k_lu_front_tiled<6>), dim3(cnt), dim3(256), sm, st, d, list, Ax, tol, stol, reuse, lds_m); break;
        default: hipLaunchKernelGGL((k_lu_front_tiled<7>), dim3(cnt), dim3(256), sm, st, d, list, Ax, tol, stol, reuse, lds_m); break;
        }
    } else {
        const size_t sm = 2 * (size_t)max_k * sizeof(int32_t) + 16;
        hipLaunchKernelGGL((k_lu_front<false, LU_NT_BIG>), dim3(cnt), dim3(LU_NT_BIG), sm, st, d, list, Ax, tol, stol, reuse, 0);
    }
}


void launch_lu_big_level(const LuDev &d, const int32_t *list, int cnt, int max_m, int max_k, const double *Ax, double tol,
                         double stol, int reuse, hipStream_t st, const uint8_t *swap_steps)
{
    if (cnt <= 0) return;
    allow_large_lds();
    hipLaunchKernelGGL(k_lub_assemble, dim3((max_m + LU_ASM_COLS - 1) / LU_ASM_COLS, cnt, (max_m + LU_ASM_ROWS - 1) / LU_ASM_ROWS), dim3(256), 0, st,
                       d, list, Ax);
    const int tiles = (max_m + 63) / 64;
    for (int jb = 0, step = 0; jb < max_k; step++) {
        const int rows = max_m - jb;                          // tallest panel of this step
        const int nth = std::min(1024, (rows + 63) / 64 * 64);
        int nbs = LU_NB;
        if (rows <= 1024) {
            if (reuse) hipLaunchKernelGGL((k_lub_panel_reg<32, 1, true>), dim3(cnt), dim3(nth), 0, st, d, list, jb, tol, stol);
            else hipLaunchKernelGGL((k_lub_panel_reg<32, 1, false>), dim3(cnt), dim3(nth), 0, st, d, list, jb, tol, stol);
        } else if (rows <= 2048) {
            nbs = 16;
            if (reuse) hipLaunchKernelGGL((k_lub_panel_reg<16, 2, true>), dim3(cnt), dim3(1024), 0, st, d, list, jb, tol, stol);
            else hipLaunchKernelGGL((k_lub_panel_reg<16, 2, false>), dim3(cnt), dim3(1024), 0, st, d, list, jb, tol, stol);
        } else {
            nbs = 8;
            if (reuse) hipLaunchKernelGGL((k_lub_panel_reg<8, 4, true>), dim3(cnt), dim3(1024), 0, st, d, list, jb, tol, stol);
            else hipLaunchKernelGGL((k_lub_panel_reg<8, 4, false>), dim3(cnt), dim3(1024), 0, st, d, list, jb, tol, stol);
            if (rows > 4096)
                hipLaunchKernelGGL(k_lub_panel, dim3(cnt), dim3(LU_NT_LDS), (size_t)LU_PANEL_LDS_DOUBLES * sizeof(double), st, d, list, jb, nbs,
                                   tol, stol, reuse);
        }
        // (a refactorisation knows from the recorded sequence which blocks interchange rows at all: lu_api.cpp, refresh_swap_steps)
        if (!swap_steps || swap_steps[step]) hipLaunchKernelGGL(k_lub_trsm, dim3(tiles, cnt), dim3(64), 0, st, d, list, jb, nbs);
        hipLaunchKernelGGL(k_lub_gemm, dim3(tiles, tiles, cnt), dim3(256), 0, st, d, list, jb, nbs);
        jb += nbs;
    }
    hipLaunchKernelGGL(k_lub_store, dim3(tiles, cnt, (max_m + 255) / 256), dim3(256), 2 * (size_t)max_k * sizeof(int32_t) + 16, st, d, list, max_m);
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


void launch_lu_fwd_big(const LuDev &d, const int32_t *list, int cnt, int max_m, int max_k, int unit, double *X, int64_t ldx, int nrhs,
                       double *W, int64_t wsize, hipStream_t st)
{
    if (cnt <= 0 || nrhs <= 0) return;
    const int chunks = (max_m + LU_BIG_CHUNK - 1) / LU_BIG_CHUNK;
    if (unit) hipLaunchKernelGGL(k_lu_fwd_big_init<true>, dim3(cnt, nrhs), dim3(LU_NT_SOLVE), (size_t)max_k * sizeof(double), st, d, list, X, ldx, W, wsize);
    else hipLaunchKernelGGL(k_lu_fwd_big_init<false>, dim3(cnt, nrhs), dim3(LU_NT_SOLVE), 0, st, d, list, X, ldx, W, wsize);
    for (int t0 = 0; t0 < max_k; t0 += LU_SB) {
        if (unit) hipLaunchKernelGGL(k_lu_fwd_big_step<true>, dim3(chunks, cnt, nrhs), dim3(LU_NT_SOLVE), 0, st, d, list, X, ldx, W, wsize, t0);
        else hipLaunchKernelGGL(k_lu_fwd_big_step<false>, dim3(chunks, cnt, nrhs), dim3(LU_NT_SOLVE), 0, st, d, list, X, ldx, W, wsize, t0);
    }
}

void launch_lu_bwd_big(const LuDev &d, const int32_t *list, int cnt, int max_m, int max_k, int unit, double *X, int64_t ldx, int nrhs,
                       double *W, int64_t wsize, hipStream_t st)
{
    if (cnt <= 0 || nrhs <= 0) return;
    const int chunks = (max_k + LU_BIG_CHUNK - 1) / LU_BIG_CHUNK, colwg = (max_k + LU_NT_SOLVE / 64 - 1) / (LU_NT_SOLVE / 64);
    if (unit) hipLaunchKernelGGL(k_lu_bwd_big_init<true>, dim3(colwg, cnt, nrhs), dim3(LU_NT_SOLVE), 0, st, d, list, X, ldx, W, wsize);
    else hipLaunchKernelGGL(k_lu_bwd_big_init<false>, dim3(colwg, cnt, nrhs), dim3(LU_NT_SOLVE), 0, st, d, list, X, ldx, W, wsize);
    for (int t0 = (max_k - 1) / LU_SB * LU_SB; t0 >= 0; t0 -= LU_SB) {
        if (unit) hipLaunchKernelGGL(k_lu_bwd_big_step<true>, dim3(chunks, cnt, nrhs), dim3(LU_NT_SOLVE), 0, st, d, list, X, ldx, W, wsize, t0);
        else hipLaunchKernelGGL(k_lu_bwd_big_step<false>, dim3(chunks, cnt, nrhs), dim3(LU_NT_SOLVE), 0, st, d, list, X, ldx, W, wsize, t0);
    }
}

void launch_lu_fvals(int64_t nf, const int64_t *src, const double *Ax, const double *rinv, const int32_t *ai32, double *out, hipStream_t st)
{
    if (nf > 0) hipLaunchKernelGGL(k_lu_fvals, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, nf, src, Ax, rinv, ai32, out);
}
void launch_lu_fterm(int64_t cnt, int nrhs, const int32_t *poslist, const int64_t *ptr, const int32_t *idx, const double *val, double *X,
                     int64_t ldx, hipStream_t st)
{
    if (cnt > 0 && nrhs > 0)
        hipLaunchKernelGGL(k_lu_fterm, dim3((unsigned)((cnt + 255) / 256), nrhs), dim3(256), 0, st, cnt, poslist, ptr, idx, val, X, ldx);
}

void launch_lu_rowmax(int64_t nnz, const int32_t *ai32, const double *Ax, double *rmax, hipStream_t st)
{
    if (nnz <= 0) return;
    hipLaunchKernelGGL(k_lu_rowmax, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, nnz, ai32, Ax, (unsigned long long *)rmax);
}
__global__ void k_lu_zero(const int64_t n, double *__restrict__ x)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = 0.0;
}
void launch_lu_zero(int64_t n, double *x, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(k_lu_zero, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, x);
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
