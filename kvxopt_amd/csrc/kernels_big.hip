// Big fronts (order m > 128 or k > 64): blocked right-looking factorisation in HBM/L2 with
// 64-column panel steps, and multi-workgroup triangular solves.
//
// Per panel step jb (two dependent launches, every big front of the level batched):
//   k_trsm_blk    : X := A * Linv' for the rows below, FP64 MFMA (v_mfma_f64_16x16x4_f64);
//   k_syrk_trailing: C -= X X' on 64x64 tiles, FP64 MFMA; its (0, 0) workgroup goes on to factor and
//                   invert the NEXT diagonal block (potrf_lds), so the 64 sequential pivots of panel
//                   jb + 64 overlap the update of panel jb;
//   k_potrf_blk   : the same factor + inverse of a 64x64 diagonal block as a kernel of its own, for the
//                   first panel of a front only.  In LDS: 16-column blocks, the 16x16 diagonal blocks by a
//                   register column sweep of one wave (factor and inverse in the same instruction
//                   stream), everything else FP64 MFMA.
// Fronts of order >= 6144: two-level blocking (launch_syrk_inner / launch_syrk_outer, k_syrk_trailing128).
// The inverses of the diagonal blocks stay resident: the solves use them as 64x64 mat-vecs, so a
// big front's triangular solve has no 64-long dependent chain and is spread over workgroups.
//
// Reference role: cholmod_l_factorize / cholmod_l_solve (src/C/cholmod.c:362, 483).
#include "device.hpp"

#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace kvx {

typedef double d4 __attribute__((ext_vector_type(4)));

// -DKVX_PHASE_TIMING (scratch/phase_timing.sh builds such a library next to the product one): cycle stamps along the path of the
// workgroup that updates tile (0, 0) and factors the next diagonal block, summed in g_phase[] and read back by kvx_dbg_phase_read:
// [0] launches counted, [1] operands + MFMA update, [2] read-modify-write of the tile, [3] tile into LDS, [4] potrf_lds as a whole,
// [5] store of factor + inverse, [6] / [7] / [8] phases A (pivot sweep) / B / C inside potrf_lds.
#ifdef KVX_PHASE_TIMING
__device__ unsigned long long g_phase[16];
#define KVX_STAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#define KVX_PHASE_ADD(slot, a, b) do { if (threadIdx.x == 0) atomicAdd(&g_phase[slot], (unsigned long long)((b) - (a))); } while (0)
#else
#define KVX_STAMP(var) do { } while (0)
#define KVX_PHASE_ADD(slot, a, b) do { } while (0)
#endif
constexpr int NB = KVX_NB;

// ------------------------------------------------------------------------------------------
// Big fronts: extend-add in HBM.  Workgroup (x, front) owns target columns [16x, 16x+16) of the
// parent front: it first zeroes their part of the update matrix (no separate memset on the level's
// critical path), then pulls the matching columns of every child, children in sequence (parent-pull:
// no atomics, bitwise reproducible).  The child columns that land in the tile come from a host-built
// table (ChildDesc::tile).  The kernel is a chain of indirect accesses, i.e. latency-bound, so each
// wave works on its (up to) four child columns at once -- 16 independent row updates per lane in
// flight -- and the next child's descriptor is fetched while the current child is added.
// (256 threads, contains barriers: every thread of the workgroup must call it)
// skip > 0: the entries (row < skip, column < skip) -- the first diagonal block -- are left to the workgroup of
// k_assemble_big_potrf that assembles that block in LDS and factors it
__device__ __forceinline__ void assemble_cols(const DevSym &ds, const FrontDesc &fd, const int ct, double *__restrict__ Lx,
                                              const double *__restrict__ Uc, double *__restrict__ Uo, const int skip = 0)
{
    const int k = fd.k, m = fd.m, u = m - k;
    const int c0 = ct * KVX_ASM_TC;
    double *P = Lx + fd.px;
    double *U = Uo + fd.ux;
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    ChildDesc cd{};
    if (fd.nchild > 0) cd = ds.cd[fd.childptr];
    for (int c = max(c0, k); c < min(c0 + KVX_ASM_TC, m); c++) {
        double *col = U + (int64_t)(c - k) * u - k;
        for (int i = c + (int)threadIdx.x; i < m; i += 256) col[i] = 0.0;
    }
    __syncthreads();
    for (int c = 0; c < fd.nchild; c++) {
        ChildDesc nx = cd;
        if (c + 1 < fd.nchild) nx = ds.cd[fd.childptr + c + 1];
        const int uc = cd.uc;
        if (uc > 0) {
            const int32_t *rl = ds.rel + cd.rel;
            const int jlo = ds.tiles[cd.tile + ct], jhi = ds.tiles[cd.tile + ct + 1];
            const double *Uch = Uc + cd.ux;
            if (jlo < jhi) {                           // workgroup-uniform
                int jc[4];
                bool okc[4];
                const double *src[4];
                double *dst[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    jc[t] = jlo + wv + 4 * t;
                    okc[t] = jc[t] < jhi;
                    if (!okc[t]) jc[t] = jlo;
                }
                int tcs[4];
#pragma unroll
                for (int t = 0; t < 4; t++) tcs[t] = rl[jc[t]];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    src[t] = Uch + (int64_t)jc[t] * uc;
                    dst[t] = (tcs[t] < k) ? P + (int64_t)tcs[t] * m : U + (int64_t)(tcs[t] - k) * u - k;
                }
                for (int base = ln; base < uc - jlo; base += 256) {
                    int r[4][4];
                    double v[4][4], old[4][4];
                    bool ok[4][4];
#pragma unroll
                    for (int t = 0; t < 4; t++)
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int i = jc[t] + base + 64 * q;
                            ok[t][q] = okc[t] && i < uc;
                            const int ii = ok[t][q] ? i : jc[t];       // row j of column j is always in range
                            r[t][q] = rl[ii];
                            v[t][q] = src[t][ii];
                        }
#pragma unroll
                    for (int t = 0; t < 4; t++)
#pragma unroll
                        for (int q = 0; q < 4; q++) old[t][q] = dst[t][r[t][q]];
#pragma unroll
                    for (int t = 0; t < 4; t++)
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if (ok[t][q] && !(tcs[t] < skip && r[t][q] < skip)) dst[t][r[t][q]] = old[t][q] + v[t][q];
                }
            }
        }
        __syncthreads();
        cd = nx;
    }
}

__global__ __launch_bounds__(256) void k_assemble_big(DevSym ds, const int32_t *__restrict__ list,
                                                      double *__restrict__ Lx, const double *__restrict__ Uc,
                                                      double *__restrict__ Uo)
{
    const FrontDesc fd = ds.fd[list[blockIdx.y]];
    if ((int)blockIdx.x * KVX_ASM_TC >= fd.m) return;
    assemble_cols(ds, fd, (int)blockIdx.x, Lx, Uc, Uo);
}

void launch_assemble_big(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m,
                         double *Lx, const double *Uchild, double *Uout)
{
    if (count <= 0) return;
    dim3 grid((unsigned)((max_m + KVX_ASM_TC - 1) / KVX_ASM_TC), (unsigned)count);
    hipLaunchKernelGGL(k_assemble_big, grid, dim3(256), 0, st, ds, list, Lx, Uchild, Uout);
}

// ------------------------------------------------------------------------------------------
// 64x64 diagonal block: Cholesky factor AND its inverse, one 256-thread workgroup, entirely in LDS.
// Right-looking over four 16-column blocks; per block step s:
//   A  wave 0 factors the 16x16 diagonal block and inverts it in the SAME instruction stream: lanes
//      0-15 hold the rows of the block, lanes 16-31 the columns of Y = D^{-1} (forward substitution
//      on the identity).  Both are "acc[t] -= mult * column_j[t], t > j" with a per-lane multiplier
//      (row r: a_rj / d_j; column c of Y: y_jc / d_j), so the inverse costs no extra instruction.
//      The unscaled pivot column travels by LDS broadcast, the pivot by v_readlane; no barrier.
//   B  FP64 MFMA, one 16x16 task per wave: the tiles below, X = A Dinv' (TRSM through the inverse),
//      and block row s of the 64x64 inverse, Y(s,j) = -Dinv_s sum_p L(s,p) Y(p,j).
//   C  FP64 MFMA trailing update of the remaining tiles, C -= X X'.
// Three barriers per block step.  The sequential part is the 64 pivot steps of phase A (~300 cycles
// each: rsqrt chain + one LDS round trip); everything else is a handful of MFMAs.
// LDS layout: only the ten 16x16 blocks of the lower block triangle are kept, block (bi, bj) at
// slot bi (bi + 1) / 2 + bj.  Blocks of the factor have ld 16 (MFMA operand reads are then 512
// contiguous bytes), blocks of the inverse ld 17 (their transposed operand reads hit distinct banks).
// ~49 KB in all, so three workgroups fit a CU -- this matters because the trailing-update kernel
// carries this structure too (see k_syrk_trailing).
constexpr int SBS = 16 * 16, YBS = 16 * 17;
__device__ __forceinline__ constexpr int blk_slot(int bi, int bj) { return bi * (bi + 1) / 2 + bj; }
__device__ __forceinline__ int s_idx(int i, int c) { return blk_slot(i >> 4, c >> 4) * SBS + (i & 15) + (c & 15) * 16; }
__device__ __forceinline__ int y_idx(int i, int c) { return blk_slot(i >> 4, c >> 4) * YBS + (i & 15) + (c & 15) * 17; }

// One pivot step, branch-free so that the 16 unrolled steps form one basic block the scheduler can
// interleave: every lane stores (the non-factor lanes into a dump slot), a failed pivot is only
// recorded in `bad`, and the diagonal is d * rsqrt(d) like every other entry of the column.
template <int J>
__device__ __forceinline__ void diag16_step(double (&acc)[16], double *colbuf, int wslot, int rr, int &bad, const PivRule pr)
{
    double *cb = colbuf + (J & 1) * 80;
    const double aj = acc[J];
    cb[wslot] = aj;
    double d = kvx_readlane(aj, J);
    const bool neg = !(d > pr.floor);                           // same instruction count as the plain rule (floor = 0, sub = 1):
    bad = (neg && bad > J) ? J : bad;                           // whether a recorded step counts is decided after the sweep
    d = neg ? pr.sub : d;
    double inv = __builtin_amdgcn_rsq(d);          // 1/sqrt(d): hardware seed + two Newton steps
    const double hd = 0.5 * d;
    inv = inv * __builtin_fma(-hd * inv, inv, 1.5);
    inv = inv * __builtin_fma(-hd * inv, inv, 1.5);
    const double lj = aj * inv;                    // l_rj; on the diagonal d / sqrt(d)
    const double w = (rr <= J) ? 0.0 : lj * inv;   // rr: row of a factor lane, a large number on the other lanes
#pragma unroll
    for (int t = J + 1; t < 16; t++) acc[t] = __builtin_fma(-w, cb[t], acc[t]);
    acc[J] = (rr < J) ? 0.0 : lj;
}
template <int... Js>
__device__ __forceinline__ void diag16_steps(double (&acc)[16], double *colbuf, int wslot, int rr, int &bad, const PivRule pr,
                                             std::integer_sequence<int, Js...>)
{
    (diag16_step<Js>(acc, colbuf, wslot, rr, bad, pr), ...);
}

struct PotrfLds {
    double S[10 * SBS];        // the block, then its factor (lower block triangle)
    double Yl[10 * YBS];       // the inverse of the factor
    double colbuf[2 * 80];     // pivot column at [0, 16), dump slots of the other lanes behind it
    double scr[3 * 16 * 17];   // per-wave staging of the 16x16 products of phase B
};

// Factor + invert the block held in lds.S (lower triangle, identity padding beyond nbk; the caller has
// filled it and passed a barrier).  256 threads.  Ends with a barrier: S = L, Yl = L^{-1}.
__device__ __forceinline__ void potrf_lds(PotrfLds &lds, int nbk, int tid, int *status, int col0, const PivRule pr)
{
    const int nblk = (nbk + 15) >> 4;
    const int i = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = tid & 15, lk = (tid >> 4) & 3;
#pragma unroll 1
    for (int s = 0; s < nblk; s++) {
        double *Sd = lds.S + blk_slot(s, s) * SBS;
        double *Yd = lds.Yl + blk_slot(s, s) * YBS;
        KVX_STAMP(pa);
        // ---- A: diagonal block and its inverse (wave 0)
        if (wv == 0) {
            const bool fac = i < 16;
            double acc[16];
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const double lv = Sd[lr + 16 * c];
                acc[c] = fac ? lv : ((i < 32 && c == lr) ? 1.0 : 0.0);
            }
            int bad = 16;
            diag16_steps(acc, lds.colbuf, fac ? lr : i, fac ? lr : 1000, bad, pr, std::make_integer_sequence<int, 16>());
            if (bad < 16 && i == 0 && pr.flag_all) atomicMin(status, col0 + 16 * s + bad);
            if (fac) {
#pragma unroll
                for (int c = 0; c < 16; c++) Sd[lr + 16 * c] = acc[c];                 // zeros above the diagonal
            } else if (i < 32) {
#pragma unroll
                for (int r = 0; r < 16; r++) Yd[r + 17 * lr] = acc[r];                 // column lr of Dinv
            }
        }
        __syncthreads();
        KVX_STAMP(pb);
        KVX_PHASE_ADD(6, pa, pb);
        // ---- B: tiles below (X = A Dinv') and block row s of the inverse; task t -> wave t
        {
            const int ntr = nblk - 1 - s;
            if (wv < ntr) {
                double *St = lds.S + blk_slot(s + 1 + wv, s) * SBS;
                d4 x = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k0 = 0; k0 < 16; k0 += 4) {
                    const double av = Yd[lr + 17 * (k0 + lk)];
                    const double bv = St[lr + 16 * (k0 + lk)];
                    x = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, x, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) St[lr + 16 * (lk + 4 * q)] = x[q];
            } else if (wv < ntr + s) {
                const int j = wv - ntr;                      // block (s, j), j < s
                double *sc = lds.scr + wv * (16 * 17);
                d4 t = (d4){0.0, 0.0, 0.0, 0.0};
                for (int pb = j; pb < s; pb++) {
                    const double *Yp = lds.Yl + blk_slot(pb, j) * YBS;
                    const double *Sp = lds.S + blk_slot(s, pb) * SBS;
#pragma unroll
                    for (int k0 = 0; k0 < 16; k0 += 4) {
                        const double av = Yp[(k0 + lk) + 17 * lr];
                        const double bv = Sp[lr + 16 * (k0 + lk)];
                        t = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, t, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; q++) sc[lr + 17 * (lk + 4 * q)] = t[q];         // T[r][c] at r + 17 c
                d4 y = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k0 = 0; k0 < 16; k0 += 4) {
                    const double av = -sc[(k0 + lk) + 17 * lr];
                    const double bv = Yd[lr + 17 * (k0 + lk)];
                    y = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, y, 0, 0, 0);
                }
                double *Yo = lds.Yl + blk_slot(s, j) * YBS;
#pragma unroll
                for (int q = 0; q < 4; q++) Yo[lr + 17 * (lk + 4 * q)] = y[q];
            }
        }
        __syncthreads();
        KVX_STAMP(pc);
        KVX_PHASE_ADD(7, pb, pc);
        // ---- C: trailing update of the tiles (ti, tj), s < tj <= ti < nblk; task t -> wave t & 3
        {
            int t = 0;
            for (int ti = s + 1; ti < nblk; ti++)
                for (int tj = s + 1; tj <= ti; tj++, t++) {
                    if ((t & 3) != wv) continue;
                    double *Sc = lds.S + blk_slot(ti, tj) * SBS;
                    const double *Sa = lds.S + blk_slot(tj, s) * SBS;
                    const double *Sb = lds.S + blk_slot(ti, s) * SBS;
                    d4 c;
#pragma unroll
                    for (int q = 0; q < 4; q++) c[q] = Sc[lr + 16 * (lk + 4 * q)];
#pragma unroll
                    for (int k0 = 0; k0 < 16; k0 += 4) {
                        const double av = -Sa[lr + 16 * (k0 + lk)];
                        const double bv = Sb[lr + 16 * (k0 + lk)];
                        c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c, 0, 0, 0);
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) Sc[lr + 16 * (lk + 4 * q)] = c[q];
                }
        }
        __syncthreads();
        KVX_STAMP(pd);
        KVX_PHASE_ADD(8, pc, pd);
    }
}

// write the factor into the panel and the inverse into the resident Linv slot (lower triangles)
__device__ __forceinline__ void potrf_store(const PotrfLds &lds, int nbk, int tid, double *P, int m, int jb, double *Yg)
{
    const int i = tid & 63, q = tid >> 6;
    if (i < nbk) {
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int c = q + 4 * t;
            if (c <= i) {
                P[(jb + i) + (int64_t)(jb + c) * m] = lds.S[s_idx(i, c)];
                Yg[i + c * NB] = lds.Yl[y_idx(i, c)];
            }
        }
    }
}

// load the diagonal block jb of the panel into lds.S (identity padding), factor + invert it, store both (256 threads, barriers)
__device__ __forceinline__ void potrf_block(const DevSym &ds, const FrontDesc &fd, int jb, double *__restrict__ Lx,
                                            double *__restrict__ Linv, int *status, PotrfLds &lds)
{
    const int k = fd.k, m = fd.m;
    const int nbk = min(NB, k - jb);
    const int tid = threadIdx.x, i = tid & 63, q = tid >> 6;
    double *P = Lx + fd.px;
    if (tid < 256) {                                // (workgroups of more than four waves: the others only keep the barriers company)
        double v[16];
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int c = q + 4 * t;
            v[t] = kvx_ld0(P, (jb + i) + (int64_t)(jb + c) * m, i < nbk && c <= i);
        }
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int c = q + 4 * t;
            if ((c >> 4) <= (i >> 4)) lds.S[s_idx(i, c)] = (i < nbk && c <= i) ? v[t] : (c == i ? 1.0 : 0.0);   // identity padding
        }
    }
    __syncthreads();
    potrf_lds(lds, nbk, tid, status, fd.first + jb, make_piv_rule(ds));
    if (tid < 256) potrf_store(lds, nbk, tid, P, m, jb, Linv + fd.linv + (int64_t)(jb / NB) * NB * NB);
}

__global__ __launch_bounds__(256) void k_potrf_blk(DevSym ds, const int32_t *__restrict__ list, int jb,
                                                   double *__restrict__ Lx, double *__restrict__ Linv, int *status)
{
    __shared__ PotrfLds lds;
    const FrontDesc fd = ds.fd[list[blockIdx.x]];
    if (jb >= fd.k) return;
    potrf_block(ds, fd, jb, Lx, Linv, status, lds);
}

void launch_potrf_blk(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int jb,
                      double *Lx, double *Linv, int *status)
{
    if (count <= 0) return;
    hipLaunchKernelGGL(k_potrf_blk, dim3((unsigned)count), dim3(256), 0, st, ds, list, jb, Lx, Linv, status);
}

// Extend-add AND the first diagonal block in one launch, for the levels at the top of the tree (a handful of fronts: the
// chain  extend-add -> diagonal block -> panel solve -> ...  is a sequence of nearly empty launches there, and the diagonal block
// of a front needs only 64 x 64 of what the extend-add produces).  One more workgroup per front assembles that block in LDS --
// the entries of A from the panel, then the children in the order the extend-add takes them, so the sums are bit for bit the
// ones it would have stored -- and goes straight on to factor and invert it (potrf_lds); the other workgroups leave the block
// alone.  Saves the k_potrf_blk launch of the level (~20 us of its critical path).  Every workgroup of this kernel reserves the
// 49 KB of the block structure: used where the launch is small (launch_assemble_big_potrf's caller decides).
__global__ __launch_bounds__(256) void k_assemble_big_potrf(DevSym ds, const int32_t *__restrict__ list, double *__restrict__ Lx,
                                                            const double *__restrict__ Uc, double *__restrict__ Uo,
                                                            double *__restrict__ Linv, int *status, unsigned nct)
{
    __shared__ PotrfLds lds;
    const FrontDesc fd = ds.fd[list[blockIdx.y]];
    const int k = fd.k, m = fd.m;
    const int nbk = min(NB, k);
    if (blockIdx.x < nct) {
        if ((int)blockIdx.x * KVX_ASM_TC >= m) return;
        assemble_cols(ds, fd, (int)blockIdx.x, Lx, Uc, Uo, nbk);
        return;
    }
    const int tid = threadIdx.x, i = tid & 63, q = tid >> 6;
    double *P = Lx + fd.px;
    {
        double v[16];
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int c = q + 4 * t;
            v[t] = kvx_ld0(P, i + (int64_t)c * m, i < nbk && c <= i);
        }
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int c = q + 4 * t;
            if ((c >> 4) <= (i >> 4)) lds.S[s_idx(i, c)] = (i < nbk && c <= i) ? v[t] : (c == i ? 1.0 : 0.0);   // identity padding
        }
    }
    __syncthreads();
    if (fd.nchild > 0) {
        ChildDesc cd = ds.cd[fd.childptr];
        for (int c = 0; c < fd.nchild; c++) {
            ChildDesc nx = cd;
            if (c + 1 < fd.nchild) nx = ds.cd[fd.childptr + c + 1];
            const int uc = cd.uc;
            if (uc > 0) {
                const int32_t *rl = ds.rel + cd.rel;
                const double *src = Uc + cd.ux;
                const int J = ds.tiles[cd.tile + NB / KVX_ASM_TC];     // child columns that land left of parent column 64 (m > 128: the table is longer)
                // wave q takes the child columns q, q + 4, ...: all loads of its (up to) 16 columns first, one round trip per child
                int tc[16], rr[16];
                double v[16];
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    const int jc = q + 4 * t;
                    const bool okc = jc < J;
                    const int ii = jc + i;
                    const bool okr = okc && ii < uc;
                    tc[t] = okc ? rl[jc] : nbk;
                    rr[t] = okr ? rl[ii] : nbk;
                    v[t] = kvx_ld0(src, (int64_t)(okc ? jc : 0) * uc + (okr ? ii : 0), okr);
                }
#pragma unroll
                for (int t = 0; t < 16; t++)
                    if (tc[t] < nbk && rr[t] < nbk) lds.S[s_idx(rr[t], tc[t])] += v[t];
            }
            __syncthreads();
            cd = nx;
        }
    }
    potrf_lds(lds, nbk, tid, status, fd.first, make_piv_rule(ds));
    potrf_store(lds, nbk, tid, P, m, 0, Linv + fd.linv);
}

void launch_assemble_big_potrf(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m,
                               double *Lx, const double *Uchild, double *Uout, double *Linv, int *status)
{
    if (count <= 0) return;
    const unsigned nct = (unsigned)((max_m + KVX_ASM_TC - 1) / KVX_ASM_TC);
    hipLaunchKernelGGL(k_assemble_big_potrf, dim3(nct + 1, (unsigned)count), dim3(256), 0, st, ds, list, Lx, Uchild, Uout, Linv, status, nct);
}

// ------------------------------------------------------------------------------------------
// X := A * Linv' for a 64-row block below the diagonal block.  Output roles swapped as in the
// trailing update (D[i][j]: i <-> panel column, j <-> row) so that stores run along rows.  All
// operand loads of the 16 k-steps are branch-free; Linv is lower triangular, so k-groups above a
// column tile are skipped.
__device__ __forceinline__ void trsm_rows(const FrontDesc &fd, int jb, int rb, double *__restrict__ Lx, const double *__restrict__ Linv,
                                          const int lt)                  // lt: thread 0..255 of the four waves doing this row block
{
    const int k = fd.k, m = fd.m;
    const int nbk = min(NB, k - jb);
    const int r0 = jb + nbk + rb * 64;
    if (r0 >= m) return;
    double *P = Lx + fd.px;
    const double *Y = Linv + fd.linv + (int64_t)(jb / NB) * NB * NB;
    const int w = lt >> 6, l = lt & 63, lr = l & 15, lk = l >> 4;
    const int rr = r0 + 16 * w + lr;
    const bool rin = rr < m;
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    // every operand of the 16 k-steps is requested before the first MFMA (56 loads per lane, predicated, no branch in between): the
    // kernel is one memory round trip + 40 MFMAs instead of four rounds of load -> wait -> MFMA (6 -> 4 us per launch on the pivot chain)
    double bq[4][4], aq[4][4][4];
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
            const int kc = 16 * g + 4 * qq + lk;
            const bool kin = kc < nbk;
            bq[g][qq] = kvx_ld0(P, rr + (int64_t)(jb + kc) * m, kin && rin);
#pragma unroll
            for (int t = g; t < 4; t++) {            // (Linv is lower triangular: column tiles left of the k-group hold nothing)
                const int cc = 16 * t + lr;
                aq[g][qq][t] = kvx_ld0(Y, cc + kc * NB, kin && cc < nbk && kc <= cc);
            }
        }
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
        for (int qq = 0; qq < 4; qq++)
#pragma unroll
            for (int t = g; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[g][qq][t], bq[g][qq], acc[t], 0, 0, 0);
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

__global__ __launch_bounds__(256) void k_trsm_blk(DevSym ds, const int32_t *__restrict__ list, int jb,
                                                  double *__restrict__ Lx, const double *__restrict__ Linv)
{
    const FrontDesc fd = ds.fd[list[blockIdx.y]];
    if (jb >= fd.k) return;
    trsm_rows(fd, jb, (int)blockIdx.x, Lx, Linv, (int)threadIdx.x);
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

// Numbering of the workgroups of a trailing-update launch over size classes of fronts (TileClasses, device.hpp; the host side
// is make_tile_classes below): workgroup id -> front index in the launch's list and tile (ti, tj).  false: nothing to do.
__device__ __forceinline__ void tri_inv(unsigned L, int &ti, int &tj)
{
    unsigned si = (unsigned)((__builtin_sqrtf(8.0f * (float)L + 1.0f) - 1.0f) * 0.5f);
    while ((si + 1) * (si + 2) / 2 <= L) si++;
    while (si * (si + 1) / 2 > L) si--;
    ti = (int)si;
    tj = (int)(L - si * (si + 1) / 2);
}
__device__ __forceinline__ bool cls_decode(const TileClasses &tc, const unsigned wgid, int &fi, int &ti, int &tj)
{
    int c = 0;
    while (c + 1 < tc.ncls && wgid >= tc.wg[c + 1]) c++;                   // (uniform)
    const unsigned local = wgid - tc.wg[c];
    const int T = tc.T[c], TC = tc.TC[c];
    if (TC > 0) {                                      // a few tile columns of T tile rows (column-limited launches)
        const unsigned tpf = (unsigned)T * (unsigned)TC;
        const unsigned t = local % tpf;
        fi = tc.first[c] + (int)(local / tpf);
        ti = (int)(t % (unsigned)T);
        tj = (int)(t / (unsigned)T);
    } else {
        const unsigned tri = (unsigned)T * (unsigned)(T + 1) / 2;
        if (T >= 16) {
            const unsigned chunk = (tri + 7) / 8, tpf = 8 * chunk;
            const unsigned t = local % tpf;
            fi = tc.first[c] + (int)(local / tpf);
            const unsigned L = (t & 7u) * chunk + (t >> 3);
            if (L >= tri) return false;
            tri_inv(L, ti, tj);
        } else {
            fi = tc.first[c] + (int)(local / tri);
            tri_inv(local % tri, ti, tj);
        }
    }
    return tj <= ti && fi < tc.first[tc.ncls];         // (ids in the padding in front of an XCD-numbered class)
}

// ------------------------------------------------------------------------------------------
// Trailing update C -= X X' on 64x64 tiles (FP64 MFMA).  The trailing matrix spans the rest of
// the panel (columns < k, ld = m, in Lx) and the update matrix (columns >= k, ld = u).
// The workgroup of tile (0, 0) owns the NEXT diagonal block: it keeps the updated tile in LDS and
// goes straight on to factor and invert it (potrf_lds) while the other tiles of the launch are still
// being updated -- the 64 sequential pivot steps of the next panel overlap this panel's update
// instead of waiting for a launch of their own.  (Running the two parts as separate launches on two
// streams was measured and lost: a cross-queue dependency costs ~10 us inside a graph replay.)
// The factor code costs the kernel nothing in occupancy: 126 VGPRs and 50 KB of LDS keep three
// workgroups per CU, what the plain update (116 VGPRs) had.
// col_lim: only the columns < min(col_lim, k) are updated (the pair schedule's narrow launch, the inner launches of the two-level
// blocking); INT_MAX = the whole trailing matrix.
// KW = 64: the K range is ONE 64-column panel [jb, jb + 64).  KW = 128: TWO panels [jb, jb + 128) in one pass over C (the pair
// schedule of launch_panel_chain): a rank-64 update moves 16 bytes of C per 128 flops and the big levels of the ~20-nnz/row
// systems are bound by exactly that traffic (rocprofv3: 30 GB per factorisation of the 21-point system, 1000 x 1000 grid) --
// with two panels per pass it is halved.
template <int KW>
__global__ __launch_bounds__(256) void k_syrk_trailing(DevSym ds, const int32_t *__restrict__ list, int jb,
                                                       double *__restrict__ Lx, double *__restrict__ Uo,
                                                       double *__restrict__ Linv, int *status, int col_lim, const TileClasses tc)
{
    int ti, tj, fi;
    if (tc.ncls > 0) {                                  // numbered over size classes of the fronts (round 4)
        if (!cls_decode(tc, blockIdx.x, fi, ti, tj)) return;
    } else {                                            // (tiles of the largest front) x (fronts): sharded mode, KVX_SYRK_DIRECT=1
        ti = blockIdx.x; tj = blockIdx.y; fi = blockIdx.z;
        if (tj > ti) return;
    }
    KVX_STAMP(q0);
    const FrontDesc fd = ds.fd[list[fi]];
    const int k = fd.k, m = fd.m, u = m - k;
    if (jb >= k) return;
    const int nbk = min(KW, k - jb);
    const int t0 = jb + nbk;
    const int r0 = t0 + KVX_TILE * ti, c0 = t0 + KVX_TILE * tj;
    if (r0 >= m) return;
    const int cend = col_lim == INT_MAX ? m : min(col_lim, k);
    if (c0 >= cend) return;
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
#ifdef KVX_PHASE_TIMING
    unsigned long long qr[5];
    qr[0] = qr[1] = qr[2] = qr[3] = qr[4] = __builtin_readcyclecounter();
#endif
#pragma unroll
    for (int kg = 0; kg < KW; kg += 16) {
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
#ifdef KVX_PHASE_TIMING
        if (kg < 64) {
            asm volatile("s_nop 0" ::: "memory");
            qr[kg / 16 + 1] = __builtin_readcyclecounter();
        }
#endif
    }
    // lane holds D[i = (l>>4) + 4q][j = l&15] with i <-> tile column, j <-> tile row.
    // Branch-free read-modify-write: all 16 loads go out (clamped addresses), then 16 predicated stores.
    KVX_STAMP(q1);
    const int rs = min(rr, m - 1);
    double *ptr[4][4];
    double old[4][4];
    bool ok[4][4];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c = c0 + 16 * t + lk + 4 * q;
            ok[t][q] = rin && c <= rr && c < cend;
            const int cs = min(c, rs);
            ptr[t][q] = (cs < k) ? P + rs + (int64_t)cs * m : U + (rs - k) + (int64_t)(cs - k) * u;
            old[t][q] = *ptr[t][q];
        }
    // the workgroup that goes on to factor the next diagonal block does not write that block back first: nobody reads it before
    // potrf_store puts the factor there, and the barrier below would wait for those stores
    const bool fused = ti == 0 && tj == 0 && t0 < cend && t0 < k;   // workgroup-uniform
    const int dend = fused ? t0 + min(NB, k - t0) : 0;              // rows below dend (a partial last block) are written as always
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (ok[t][q] && rr >= dend) *ptr[t][q] = old[t][q] - acc[t][q];
    if (fused) {
        KVX_STAMP(q2);
        __shared__ PotrfLds lds;
        const int nb2 = min(NB, k - t0);
        const int i = 16 * w + lr;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int cc = 16 * t + lk + 4 * q;
                if (t <= w) lds.S[s_idx(i, cc)] = (i < nb2 && cc <= i) ? old[t][q] - acc[t][q] : (cc == i ? 1.0 : 0.0);
            }
        __syncthreads();
        KVX_STAMP(q3);
        potrf_lds(lds, nb2, threadIdx.x, status, fd.first + t0, make_piv_rule(ds));
        KVX_STAMP(q4);
        potrf_store(lds, nb2, threadIdx.x, P, m, t0, Linv + fd.linv + (int64_t)(t0 / NB) * NB * NB);
#ifdef KVX_PHASE_TIMING
        __threadfence();
#endif
        KVX_STAMP(q5);
#ifdef KVX_PHASE_TIMING
        if (threadIdx.x == 0) atomicAdd(&g_phase[0], 1ull);
#endif
#ifdef KVX_PHASE_TIMING
        KVX_PHASE_ADD(9, q0, qr[0]);
        for (int g = 0; g < 4; g++) KVX_PHASE_ADD(10 + g, qr[g], qr[g + 1]);
#endif
        KVX_PHASE_ADD(1, q0, q1); KVX_PHASE_ADD(2, q1, q2); KVX_PHASE_ADD(3, q2, q3); KVX_PHASE_ADD(4, q3, q4); KVX_PHASE_ADD(5, q4, q5);
    }
}

#ifdef KVX_PHASE_TIMING
extern "C" int kvx_dbg_phase_read(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif

// ------------------------------------------------------------------------------------------
// Round 4: the trailing update with its operands staged through LDS.  k_syrk_trailing has every one of its four waves fetch
// the whole 64-row column strip of the tile for itself, 8 bytes per lane: 320 K doubles of L2 traffic per 64 x 64 x K tile.
// Here the workgroup stages the two strips ONCE (128 K doubles), 16 bytes per lane, in chunks of 16 panel columns, double
// buffered (the global loads of chunk c + 1 are in flight under the MFMAs of chunk c, one barrier per chunk); wave (wr, wc)
// owns a 32 x 32 quarter (2 x 2 MFMA tiles, 4 LDS operand reads per 4 MFMAs).  Measured on MI355X (scratch/syrk_lab.hip, dense
// u = 5000, K = 512): 39.4 TF/s against 29.2 for the direct form and 37.4 for 128 x 128 tiles -- v_mfma_f64_16x16x4_f64 issues
// once per 99 cycles per SIMD with two or more waves resident (130 with one), i.e. the pipe's ceiling is 50.7 TF/s, not the
// nominal 78.6 (tools/fp64_peak.hip, profiles/r04_fp64_peak.txt), and small tiles keep three workgroups per CU.
// The update region is rows and columns >= t0, lower triangle, columns < cend; the K range is the panel columns
// [kb, kb + min(klen, k - kb)).
//   UONLY = false (chain step): t0 = kb + K range, cend = min(col_lim, k) or the whole front (col_lim = INT_MAX); the
//                 workgroup of tile (0, 0) factors and inverts the next diagonal block as in k_syrk_trailing.
//   UONLY = true  (deferred, "far" update): t0 = min(col_lim, k), cend = m -- everything from column col_lim on, update matrix
//                 included (col_lim >= k: the update matrix alone); any K range; never a diagonal block that is factored next.
// Reads one double past row m - 1 of a panel column (16-byte loads at clamped rows): Lx carries two doubles of slack.
constexpr int SL_KC = 16, SL_LD = KVX_TILE + 16;        // LD = 64 + 16: the lanes of one LDS pass (32 lanes = 2 k) hit distinct banks
typedef double d2v __attribute__((ext_vector_type(2)));
struct SyrkLdsStage {
    double xa[2][SL_KC * SL_LD];       // X[columns of the tile][k chunk], k-major
    double xb[2][SL_KC * SL_LD];       // X[rows of the tile][k chunk]
};
union SyrkLdsU {
    SyrkLdsStage st;
    PotrfLds po;
};
struct SyrkLdsOnly {
    SyrkLdsStage st;
    struct { double S[1]; } po;      // (never used)
};

// Numbering of the workgroups of one launch (TileClasses, device.hpp).  A launch updates every big front of a level that is still
// in the chain; their trailing matrices differ by an order of magnitude, and a (tiles of the largest front) x (fronts) grid is
// mostly workgroups that find nothing to do -- the dispatcher starts one per ~2.6 ns, and the bottom levels of the 21-point
// system launched 870 000 of them per step for 30 000 tiles of work (2.3 ms, measured).  The host therefore hands the kernel
// the fronts sorted by size and cut into classes of similar tile counts; a class is a (tiles of ITS largest front) x (its
// fronts) block of consecutive workgroup ids.
// tile t of a front -> (ti, tj).  Triangular classes of 16 or more tile rows: the 8 XCDs (workgroup ids go round-robin over them,
// each has its own L2) take contiguous eighths of the row-major tile order, so that the workgroups resident on one XCD work on
// neighbouring tiles of a few tile rows and share their operand strips.
static inline unsigned cls_tiles_per_front(int T, int TC)
{
    if (TC > 0) return (unsigned)T * (unsigned)TC;
    const unsigned tri = (unsigned)T * (unsigned)(T + 1) / 2;
    return T >= 16 ? ((tri + 7) / 8) * 8 : tri;
}

template <bool UONLY, class Lds>
__device__ __forceinline__ void syrk_lds_tile(Lds &lds, const unsigned wgid, const DevSym &ds, const int32_t *__restrict__ list, int kb, int klen,
                                              double *__restrict__ Lx, double *__restrict__ Uo,
                                              double *__restrict__ Linv, int *status, int col_lim, const TileClasses &tc)
{
    int ti, tj, fi;
    if (!cls_decode(tc, wgid, fi, ti, tj)) return;
    const FrontDesc fd = ds.fd[list[fi]];
    const int k = fd.k, m = fd.m, u = m - k;
    if (kb >= k) return;
    const int nbk = min(klen, k - kb);
    const int t0 = UONLY ? min(col_lim, k) : kb + nbk;
    const int r0 = t0 + KVX_TILE * ti, c0 = t0 + KVX_TILE * tj;
    if (r0 >= m) return;
    const int cend = (UONLY || col_lim == INT_MAX) ? m : min(col_lim, k);
    if (c0 >= cend) return;
    double *P = Lx + fd.px;
    double *U = Uo + fd.ux;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wr = w >> 1, wc = w & 1;
    const int l = tid & 63, lr = l & 15, lk = l >> 4;
    // staging map: a strip chunk is 32 row pairs x 16 k = 512 16-byte units, two per thread
    const int spr = tid & 31, sk = tid >> 5;           // unit j: k = sk + 8 j
    const double *Pa = P + min(c0 + 2 * spr, m - 1) + (int64_t)kb * m;
    const double *Pb = P + min(r0 + 2 * spr, m - 1) + (int64_t)kb * m;
    d2v ga[2], gb[2];
    auto ldg = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int kk = k0 + sk + 8 * j;
            const int64_t off = (int64_t)min(kk, nbk - 1) * m;
            ga[j] = *(const d2v *)(Pa + off);
            gb[j] = *(const d2v *)(Pb + off);
            if (kk >= nbk) { ga[j] = (d2v){0.0, 0.0}; gb[j] = (d2v){0.0, 0.0}; }
        }
    };
    d4 acc[2][2];
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int t = 0; t < 2; t++) acc[s][t] = (d4){0.0, 0.0, 0.0, 0.0};
    const int nchunk = (nbk + SL_KC - 1) / SL_KC;
    ldg(0);
    for (int ch = 0; ch < nchunk; ch++) {
        double *xa = lds.st.xa[ch & 1], *xb = lds.st.xb[ch & 1];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            *(d2v *)(xa + (sk + 8 * j) * SL_LD + 2 * spr) = ga[j];
            *(d2v *)(xb + (sk + 8 * j) * SL_LD + 2 * spr) = gb[j];
        }
        if (ch + 1 < nchunk) ldg((ch + 1) * SL_KC);
        __syncthreads();
        const double *oa = xa + 32 * wc + lr;
        const double *ob = xb + 32 * wr + lr;
#pragma unroll
        for (int ks = 0; ks < SL_KC; ks += 4) {
            const double a0 = oa[(ks + lk) * SL_LD], a1 = oa[(ks + lk) * SL_LD + 16];
            const double b0 = ob[(ks + lk) * SL_LD], b1 = ob[(ks + lk) * SL_LD + 16];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        // (the other buffer is free again once every wave has passed this chunk's barrier: one barrier per chunk)
    }
    // epilogue: lane holds C[row = r0 + 32 wr + 16 s + lr][col = c0 + 32 wc + 16 t + lk + 4 q]; branch-free read-modify-write
    const bool fused = !UONLY && ti == 0 && tj == 0 && t0 < cend && t0 < k;      // workgroup-uniform
    const int nb2 = min(NB, k - t0);
    const int dend = fused ? t0 + nb2 : 0;             // rows below dend (a partial last block) are written as always
    if (fused) __syncthreads();                        // the staging buffers become the image of the diagonal block
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int rr = r0 + 32 * wr + 16 * s + lr;
        const bool rin = rr < m;
        const int rs = min(rr, m - 1);
        double *ptr[2][4];
        double old[2][4];
        bool ok[2][4];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int c = c0 + 32 * wc + 16 * t + lk + 4 * q;
                ok[t][q] = rin && c <= rr && c < cend;
                const int cs = min(c, rs);
                ptr[t][q] = (cs < k) ? P + rs + (int64_t)cs * m : U + (rs - k) + (int64_t)(cs - k) * u;
                old[t][q] = *ptr[t][q];
            }
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const double v = old[t][q] - acc[s][t][q];
                if (ok[t][q] && rr >= dend) *ptr[t][q] = v;
                if constexpr (!UONLY) if (fused) {
                    const int i = 32 * wr + 16 * s + lr, cc = 32 * wc + 16 * t + lk + 4 * q;
                    if ((cc >> 4) <= (i >> 4)) lds.po.S[s_idx(i, cc)] = (i < nb2 && cc <= i) ? v : (cc == i ? 1.0 : 0.0);
                }
            }
    }
    if constexpr (!UONLY) if (fused) {
        __syncthreads();
        potrf_lds(lds.po, nb2, tid, status, fd.first + t0, make_piv_rule(ds));
        potrf_store(lds.po, nb2, tid, P, m, t0, Linv + fd.linv + (int64_t)(t0 / NB) * NB * NB);
    }
}

// One workgroup per tile, or (opt-in, KVX_FAR_WGS) a fixed number of resident workgroups that walk the tiles with stride
// gridDim.x.  Beside a far launch the chain's launches wait (an inner step of 27 us took 200, a panel solve of 11 us 38:
// rocprofv3 timeline of the 21-point system); neither two resident far workgroups per CU (nothing of the far launch pending in
// the dispatcher, room for a 50 KB chain workgroup on every CU) nor LDS padding to the same effect changed that (21-point
// 23.0 - 23.6 ms against 23.3 - 23.5, 100^3 122 against 109 ms): the chain's tiles share SIMDs whose matrix pipe the far
// tiles keep busy.
template <bool UONLY>
__global__ __launch_bounds__(256, 2) void k_syrk_lds(DevSym ds, const int32_t *__restrict__ list, int kb, int klen,
                                                     double *__restrict__ Lx, double *__restrict__ Uo,
                                                     double *__restrict__ Linv, int *status, int col_lim, const TileClasses tc)
{
    // (the update of the update matrices never factors a diagonal block: 40 KB instead of 50)
    __shared__ __attribute__((aligned(16))) typename std::conditional<UONLY, SyrkLdsOnly, SyrkLdsU>::type lds;
    const unsigned total = tc.wg[tc.ncls];
    for (unsigned wgid = blockIdx.x; wgid < total; wgid += gridDim.x) {
        syrk_lds_tile<UONLY>(lds, wgid, ds, list, kb, klen, Lx, Uo, Linv, status, col_lim, tc);
        if (gridDim.x < total) __syncthreads();        // (the next tile's first chunk goes into the buffer the last one may still be read from)
    }
}

// Size classes of a launch.  hm / hk: order and pivot columns of the fronts in list order (host copies; the list is sorted by the
// order of the update region, largest first, so classes are runs of the list).  A class ends where the tile count of the next
// front falls below ~0.7 of the class's largest, or rises above it.
static TileClasses make_tile_classes(bool uonly, const int32_t *hm, const int32_t *hk, int count, int kb, int klen, int col_lim)
{
    TileClasses tc;
    int c = -1;
    for (int i = 0; i < count; i++) {
        const int m = hm[i], k = hk[i];
        int R = 0, C = 0;
        if (kb < k) {
            const int t0 = uonly ? std::min(col_lim, k) : kb + std::min(klen, k - kb);
            const int cend = (uonly || col_lim == INT_MAX) ? m : std::min(col_lim, k);
            R = std::max(m - t0, 0);
            C = std::max(cend - t0, 0);
        }
        const int T = (R + KVX_TILE - 1) / KVX_TILE, TCf = (C + KVX_TILE - 1) / KVX_TILE;
        if (c == KVX_MAXCLS - 1) {                     // out of classes: the last one takes the rest, whatever its sizes
            tc.T[c] = std::max(tc.T[c], T);
            tc.TC[c] = std::max(tc.TC[c], TCf);
            continue;
        }
        if (c >= 0 && T <= tc.T[c] && T * 10 >= tc.T[c] * 7) {
            tc.TC[c] = std::max(tc.TC[c], TCf);
            continue;
        }
        c++;
        tc.first[c] = i;
        tc.T[c] = T;
        tc.TC[c] = TCf;
    }
    tc.ncls = c + 1;
    tc.wg[0] = 0;
    tc.first[tc.ncls] = count;
    for (int q = 0; q < tc.ncls; q++) {
        const int T = tc.T[q];
        if (tc.TC[q] * 2 >= T) tc.TC[q] = 0;           // rectangular numbering only where it saves at least half of the workgroups
        unsigned w0 = tc.wg[q];
        if (tc.TC[q] == 0 && T >= 16) w0 = (w0 + 7u) & ~7u;                   // XCD numbering: the class starts on XCD 0
        tc.wg[q] = w0;
        tc.wg[q + 1] = w0 + (T > 0 ? cls_tiles_per_front(T, tc.TC[q]) : 0u) * (unsigned)(tc.first[q + 1] - tc.first[q]);
    }
    return tc;
}

static void launch_syrk_lds_cls(hipStream_t st, bool uonly, const DevSym &ds, const int32_t *list, const TileClasses &tc, int kb, int klen,
                                double *Lx, double *Uout, double *Linv, int *status, int col_lim)
{
    if (tc.ncls <= 0) return;
    const unsigned gx = tc.wg[tc.ncls];
    if (gx == 0) return;
    // KVX_FAR_WGS > 0: a far launch as that many resident workgroups walking the tiles (measured: no gain, see k_syrk_lds)
    static const unsigned far_wgs = [] { const char *e = getenv("KVX_FAR_WGS"); return e ? (unsigned)atoi(e) : 0u; }();
    if (uonly)
        hipLaunchKernelGGL(k_syrk_lds<true>, dim3(far_wgs ? std::min(gx, far_wgs) : gx), dim3(256), 0, st, ds, list, kb, klen, Lx, Uout, Linv, status, col_lim, tc);
    else
        hipLaunchKernelGGL(k_syrk_lds<false>, dim3(gx), dim3(256), 0, st, ds, list, kb, klen, Lx, Uout, Linv, status, col_lim, tc);
}

// without size information: one class, every front gets the tiles of an update region of order `rows`
// (an over-estimate is fine: empty tiles exit at once)
static void launch_syrk_lds(hipStream_t st, bool uonly, const DevSym &ds, const int32_t *list, int count, int rows, int kb, int klen,
                            double *Lx, double *Uout, double *Linv, int *status, int col_lim)
{
    if (count <= 0 || rows <= 0) return;
    TileClasses tc;
    tc.ncls = 1;
    tc.first[0] = 0; tc.first[1] = count;
    tc.T[0] = (rows + KVX_TILE - 1) / KVX_TILE;
    const int TC = col_lim >= KVX_COLS_PIVOT ? tc.T[0] : std::min(tc.T[0], (std::max(col_lim - kb - klen, 1) + KVX_TILE - 1) / KVX_TILE);
    tc.TC[0] = TC * 2 < tc.T[0] ? TC : 0;
    tc.wg[0] = 0;
    tc.wg[1] = cls_tiles_per_front(tc.T[0], tc.TC[0]) * (unsigned)count;
    launch_syrk_lds_cls(st, uonly, ds, list, tc, kb, klen, Lx, Uout, Linv, status, col_lim);
}

// the launches of the single-GPU chain: `list` = the big fronts still in the chain at panel step kb, largest update region
// first (api.cpp build_chain_lists), hm / hk their orders and pivot counts.  Launches of few tiles with one or two panels go
// to the direct kernel (every wave requests all operands of a tile up front: one memory round trip on the chain's critical
// path where the LDS-staged kernel has one per 16 panel columns); KVX_SYRK_LDS_TILES = tile count from which on the staged one.
void launch_syrk_step(hipStream_t st, const DevSym &ds, const int32_t *list, const int32_t *hm, const int32_t *hk, int count, int kb, int klen,
                      double *Lx, double *Uout, double *Linv, int *status, int col_lim)
{
    if (count <= 0) return;
    const TileClasses tc = make_tile_classes(false, hm, hk, count, kb, klen, col_lim);
    static const unsigned lds_tiles = [] { const char *e = getenv("KVX_SYRK_LDS_TILES"); return e ? (unsigned)atoll(e) : 3000u; }();
    const unsigned gx = tc.ncls > 0 ? tc.wg[tc.ncls] : 0u;
    if (gx == 0) return;
    if (gx < lds_tiles && klen <= 2 * NB) {
        if (klen <= NB) hipLaunchKernelGGL(k_syrk_trailing<64>, dim3(gx), dim3(256), 0, st, ds, list, kb, Lx, Uout, Linv, status, col_lim, tc);
        else hipLaunchKernelGGL(k_syrk_trailing<128>, dim3(gx), dim3(256), 0, st, ds, list, kb, Lx, Uout, Linv, status, col_lim, tc);
        return;
    }
    launch_syrk_lds_cls(st, false, ds, list, tc, kb, klen, Lx, Uout, Linv, status, col_lim);
}
// deferred ("far") update with the panel block [kb, kb + klen): everything from column t0 on -- later pivot columns and the
// update matrix; t0 >= k: the update matrix alone -- for the fronts of the list
void launch_syrk_far(hipStream_t st, const DevSym &ds, const int32_t *list, const int32_t *hm, const int32_t *hk, int count, int kb, int klen,
                     int t0, double *Lx, double *Uout)
{
    if (count <= 0) return;
    const TileClasses tc = make_tile_classes(true, hm, hk, count, kb, klen, t0);
    launch_syrk_lds_cls(st, true, ds, list, tc, kb, klen, Lx, Uout, nullptr, nullptr, t0);
}

// ------------------------------------------------------------------------------------------
// The same update on 128 x 128 tiles (opt-in, see launch_syrk_trailing for the measurement): the
// two operand strips X[rows, :] and X[cols, :] staged through LDS in chunks of 16 panel columns
// (k-major, double-buffered, global loads of chunk c + 1 in flight under the MFMAs of chunk c).
// Wave (wr, wc) owns a 64 x 64 quarter: 4 x 4 MFMA tiles, 8 LDS operand reads per 16 MFMAs.  The staging code is
// branch-free (a diagonal tile simply stages its strip twice): branches around the loads cost more than the loads.
// Per tile 192 KB of operand reads drop to 32 KB (the 64-tile kernel streams every operand from L2
// once per wave), and the read-modify-write of C is amortised over four times the flops per launch
// overhead.  Tile (0, 0) still factors the next diagonal block (its top-left quarter, wave (0, 0)).
constexpr int SY_T = 128, SY_KC = 16, SY_LD = 144;      // LD = 128 + 16: the 4 k-lanes of an operand read hit distinct banks

struct SyrkStage {
    double xa[2][SY_KC * SY_LD];       // X[cols of the tile][k chunk], k-major
    double xb[2][SY_KC * SY_LD];       // X[rows of the tile][k chunk]
};
union SyrkLds {
    SyrkStage st;
    PotrfLds po;
};

// Columns [kb, min(kb + klen, k)) of the panel are the update's K range (klen = 64: one panel; 256: the outer
// update of the two-level blocking, which is what makes this kernel pay: four times the flops per pass over C).
// (launch bound: two waves per SIMD -- the 128 accumulator registers live in AGPRs, the rest must fit 128 VGPRs;
// at one wave per SIMD nothing hides the staging loads and the kernel ran at 21 TF/s)
// DIST (sharded mode, dist_api.cpp): the columns of the front are dealt out in blocks of own.ob columns, round-robin over
// own.g ranks -- the pivot columns from column 0, the update columns from column k on (so that the blocks of the update
// matrix do not depend on k mod ob); this rank (own.r) updates only the columns it owns and the next diagonal block is
// factored by a launch of its own (its owner is whoever owns the block, not this kernel's (0, 0) tile).
struct ColOwner { int ob, g, r, c_from, c_to, fuse; };   // + the columns [c_from, c_to) this launch may touch (look-ahead: the next pivot block first);
                                                          // fuse: the (0, 0) tile still factors the next diagonal block (single-GPU look-ahead, ob = g = 1)
__device__ __forceinline__ bool col_owned(const ColOwner &o, int c, int k, int nkb)
{
    const int blk = c < k ? c / o.ob : nkb + (c - k) / o.ob;
    return blk % o.g == o.r && c >= o.c_from && c < o.c_to;
}
template <bool DIST>
__global__ __launch_bounds__(512, 4) void k_syrk_trailing128(DevSym ds, const int32_t *__restrict__ list, int kb, int klen,
                                                          double *__restrict__ Lx, double *__restrict__ Uo,
                                                          double *__restrict__ Linv, int *status, ColOwner own)
{
    __shared__ SyrkLds lds;
    // XCD-aware tile order: workgroup ids go round-robin over the 8 XCDs, each with its own 4 MB L2.  The lower
    // triangle of tiles is cut into 8 x 8 super-blocks and XCD x takes the super-blocks x, x + 8, ...: the 64
    // workgroups resident on an XCD then share 16 operand strips (4 MB) instead of touching the whole panel.
    int ti, tj;
    {
        const unsigned id = blockIdx.x, xcd = id & 7u, slot = id >> 3;
        const unsigned L = ((slot >> 6) * 8u + xcd) * 64u + (slot & 63u);
        const unsigned SB = L >> 6, within = L & 63u;
        unsigned si = (unsigned)((__builtin_sqrtf(8.0f * (float)SB + 1.0f) - 1.0f) * 0.5f);
        while ((si + 1) * (si + 2) / 2 <= SB) si++;
        while (si * (si + 1) / 2 > SB) si--;
        const unsigned sj = SB - si * (si + 1) / 2;
        ti = (int)(8 * si + (within >> 3));
        tj = (int)(8 * sj + (within & 7u));
    }
    if (tj > ti) return;
    const FrontDesc fd = ds.fd[list[blockIdx.z]];
    const int k = fd.k, m = fd.m, u = m - k;
    if (kb >= k) return;
    const int jb = kb;
    const int nbk = min(klen, k - kb);                             // K of this update
    const int t0 = kb + nbk;
    const int r0 = t0 + SY_T * ti, c0 = t0 + SY_T * tj;
    if (r0 >= m) return;
    const int nkb = DIST ? (k + own.ob - 1) / own.ob : 0;
    if (DIST) {                                                    // workgroup-uniform: a tile without a column of this rank
        bool any = false;
        for (int c = max(c0, own.c_from); c < min(min(c0 + SY_T, m), own.c_to) && !any; ) {
            any = col_owned(own, c, k, nkb);
            const int step = c < k ? min(own.ob - c % own.ob, k - c) : own.ob - (c - k) % own.ob;   // first column of the next block
            c += step;
        }
        if (!any) return;
    }
    double *P = Lx + fd.px;
    double *U = Uo + fd.ux;
    const int tid = threadIdx.x;
    // eight waves: wave (wr, wc) owns rows 32 wr .. +31 and columns 64 wc .. +63 of the tile (2 x 4 MFMA tiles, 64
    // accumulator registers), so four waves per SIMD are resident to cover each other's LDS and barrier waits
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wr = w >> 1, wc = w & 1;
    const int l = tid & 63, lr = l & 15, lk = l >> 4;
    const bool diag = ti == tj;
    const bool active = !(diag && wr < 2 && wc == 1);              // the upper-right quarter of a diagonal tile is not stored
    // staging map: thread -> (row of the strip, 4 of the 16 k of a chunk)
    const int srow = tid & 127, sk0 = (tid >> 7) * 4;
    const bool ra_ok = c0 + srow < m, rb_ok = r0 + srow < m;
    const double *Pa = P + min(c0 + srow, m - 1) + (int64_t)jb * m;
    const double *Pb = P + min(r0 + srow, m - 1) + (int64_t)jb * m;
    double ga[4], gb[4];
    const int nchunk = (nbk + SY_KC - 1) / SY_KC;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int kk = sk0 + j;
        ga[j] = kvx_ld0(Pa, (int64_t)kk * m, ra_ok && kk < nbk);
        gb[j] = kvx_ld0(Pb, (int64_t)kk * m, rb_ok && kk < nbk);
    }
    d4 acc[2][4];
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int t = 0; t < 4; t++) acc[s2][t] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int ch = 0; ch < nchunk; ch++) {
        double *xa = lds.st.xa[ch & 1], *xb = lds.st.xb[ch & 1];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            xa[(sk0 + j) * SY_LD + srow] = ga[j];
            xb[(sk0 + j) * SY_LD + srow] = gb[j];
        }
        if (ch + 1 < nchunk) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int kk = (ch + 1) * SY_KC + sk0 + j;
                ga[j] = kvx_ld0(Pa, (int64_t)kk * m, ra_ok && kk < nbk);
                gb[j] = kvx_ld0(Pb, (int64_t)kk * m, rb_ok && kk < nbk);
            }
        }
        __syncthreads();
        if (active) {
            const double *oa = xa + 64 * wc + lr;
            const double *ob = xb + 32 * wr + lr;
#pragma unroll
            for (int ks = 0; ks < SY_KC; ks += 4) {
                double av[4], bv[2];
#pragma unroll
                for (int t = 0; t < 4; t++) av[t] = oa[(ks + lk) * SY_LD + 16 * t];
#pragma unroll
                for (int t = 0; t < 2; t++) bv[t] = ob[(ks + lk) * SY_LD + 16 * t];
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                    for (int t = 0; t < 4; t++)
                        acc[s2][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], bv[s2], acc[s2][t], 0, 0, 0);
            }
        }
        // the other buffer is free again once every wave has passed this chunk's barrier; the next
        // iteration writes it before its own barrier, so one barrier per chunk suffices
    }
    // epilogue: lane holds C[row = .. + 16 s2 + lr][col = .. + 16 t + lk + 4 q]; branch-free RMW in batches of 16
    const bool fuse = (!DIST || own.fuse) && ti == 0 && tj == 0 && t0 < k;      // workgroup-uniform
    const int nb2 = min(NB, k - t0);
    if (fuse) __syncthreads();                                     // staging buffers are about to become the potrf image
    if (active) {
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
            const int rr = r0 + 32 * wr + 16 * s2 + lr;
            const bool rin = rr < m;
            const int rs = min(rr, m - 1);
            double *ptr[4][4];
            double old[4][4];
            bool ok[4][4];
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int c = c0 + 64 * wc + 16 * t + lk + 4 * q;
                    ok[t][q] = rin && c <= rr && (!DIST || col_owned(own, c, k, nkb));
                    const int cs = min(c, rs);
                    ptr[t][q] = (cs < k) ? P + rs + (int64_t)cs * m : U + (rs - k) + (int64_t)(cs - k) * u;
                    old[t][q] = *ptr[t][q];
                }
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const double v = old[t][q] - acc[s2][t][q];
                    if (ok[t][q]) *ptr[t][q] = v;
                    if (fuse && wr < 2 && wc == 0) {               // top-left quarter = the next diagonal block
                        const int i = 32 * wr + 16 * s2 + lr, cc = 16 * t + lk + 4 * q;
                        if (t <= 2 * wr + s2) lds.po.S[s_idx(i, cc)] = (i < nb2 && cc <= i) ? v : (cc == i ? 1.0 : 0.0);
                    }
                }
        }
    }
    if (fuse) {
        __syncthreads();
        potrf_lds(lds.po, nb2, tid, status, fd.first + t0, make_piv_rule(ds));   // (waves 4-7 only keep its barriers company)
        if (tid < 256) potrf_store(lds.po, nb2, tid, P, m, t0, Linv + fd.linv + (int64_t)(t0 / NB) * NB * NB);
    }
}

// grid of k_syrk_trailing128: 64 workgroups per 8 x 8 super-block of the lower tile triangle, padded to a multiple of 8 super-blocks
static dim3 syrk128_grid(int rows, int count)
{
    const unsigned T2 = (unsigned)((rows + SY_T - 1) / SY_T), S8 = (T2 + 7) / 8;
    const unsigned nsb = ((S8 * (S8 + 1) / 2 + 7) / 8) * 8;
    return dim3(nsb * 64u, 1, (unsigned)count);
}

static TileClasses no_classes() { TileClasses tc; tc.ncls = 0; return tc; }

// KVX_SYRK_DIRECT=1: the round-3 kernels (every wave loads its operands from global memory) instead of the LDS-staged tiles
static bool syrk_direct()
{
    static const bool v = [] { const char *e = getenv("KVX_SYRK_DIRECT"); return e && e[0] == '1'; }();
    return v;
}

void launch_syrk_trailing(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int jb,
                          double *Lx, double *Uout, double *Linv, int *status, int col_lim)
{
    if (count <= 0) return;
    int rows = max_m - jb - 1;
    if (rows <= 0) return;
    if (!syrk_direct()) { launch_syrk_lds(st, false, ds, list, count, rows, jb, NB, Lx, Uout, Linv, status, col_lim); return; }
    const int64_t T = (rows + KVX_TILE - 1) / KVX_TILE;
    // Measured (MI355X): with 64-column panels the 128-tile kernel LOSES (21-point stencil, n = 1e6: factor 24 -> 35 ms):
    // a rank-64 update is bound by the read-modify-write of C (16 B per 128 flops), not by operand traffic, and the
    // 64-tile kernel overlaps that traffic better at three workgroups per CU.  It pays with more panel columns per pass
    // over C (launch_syrk_inner / launch_syrk_outer); for single panels it is opt-in: KVX_SYRK128_TILES = tile count
    // from which on it is used.
    const char *e = getenv("KVX_SYRK128_TILES");
    const int64_t big_limit = e ? atoll(e) : INT64_MAX;
    if (T * (T + 1) / 2 * count >= big_limit) {
        hipLaunchKernelGGL(k_syrk_trailing128<false>, syrk128_grid(rows, count), dim3(512), 0, st, ds, list, jb, NB, Lx, Uout, Linv, status, ColOwner{1, 1, 0, 0, INT_MAX, 0});
    } else {                                          // latency regime: more, smaller workgroups
        hipLaunchKernelGGL(k_syrk_trailing<64>, dim3((unsigned)T, (unsigned)T, (unsigned)count), dim3(256), 0, st, ds, list, jb, Lx, Uout, Linv, status, col_lim, no_classes());
    }
}

// Pair schedule, second launch: the panels [jb, jb + 64) and [jb + 64, jb + 128) applied together to everything right of them
// (the first launch -- launch_syrk_inner(jb, jb + 128) -- has brought the second panel's own columns up to date with the first).
void launch_syrk_pair(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int jb,
                      double *Lx, double *Uout, double *Linv, int *status, int col_lim)
{
    if (count <= 0) return;
    int rows = max_m - jb - 1;                        // (an over-estimate of the trailing order: empty tiles exit at once)
    if (rows <= 0) return;
    if (!syrk_direct()) { launch_syrk_lds(st, false, ds, list, count, rows, jb, 2 * NB, Lx, Uout, Linv, status, col_lim); return; }
    const unsigned T = (unsigned)((rows + KVX_TILE - 1) / KVX_TILE);
    hipLaunchKernelGGL(k_syrk_trailing<128>, dim3(T, T, (unsigned)count), dim3(256), 0, st, ds, list, jb, Lx, Uout, Linv, status, col_lim, no_classes());
}

// Two-level blocking for the fronts that are flop-bound (level with a front of order >= KVX_TWO_LEVEL_M): the pivot
// columns are taken in outer blocks of 256; inside one, each 64-column panel updates only the rest of the outer block
// (launch_syrk_inner, at most three column tiles wide), and everything to the right of it -- later pivot columns and
// the update matrix -- gets ONE rank-256 update per outer block (launch_syrk_outer): a quarter of the passes over C.
// Both still factor the next diagonal block in their (0, 0) workgroup.
void launch_syrk_inner(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int jb, int ob_end,
                       double *Lx, double *Uout, double *Linv, int *status)
{
    if (count <= 0 || jb + NB >= ob_end) return;
    int rows = max_m - jb - 1;
    if (rows <= 0) return;
    if (!syrk_direct()) { launch_syrk_lds(st, false, ds, list, count, rows, jb, NB, Lx, Uout, Linv, status, ob_end); return; }
    const unsigned T = (unsigned)((rows + KVX_TILE - 1) / KVX_TILE);
    const unsigned TC = (unsigned)std::min<int>((int)T, (ob_end - jb - NB + KVX_TILE - 1) / KVX_TILE);
    hipLaunchKernelGGL(k_syrk_trailing<64>, dim3(T, TC, (unsigned)count), dim3(256), 0, st, ds, list, jb, Lx, Uout, Linv, status, ob_end, no_classes());
}

void launch_syrk_outer(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int ob, int ob_len,
                       double *Lx, double *Uout, double *Linv, int *status)
{
    if (count <= 0) return;
    int rows = max_m - ob - 1;                        // (an over-estimate of the trailing order: empty tiles exit at once)
    if (rows <= 0) return;
    hipLaunchKernelGGL(k_syrk_trailing128<false>, syrk128_grid(rows, count), dim3(512), 0, st, ds, list, ob, ob_len, Lx, Uout, Linv, status, ColOwner{1, 1, 0, 0, INT_MAX, 0});
}

// sharded mode: the rank-ob_len update of the columns this rank owns (block-cyclic, see ColOwner); no fused factorisation
void launch_syrk_outer_dist(hipStream_t st, const DevSym &ds, const int32_t *list, int max_m, int ob, int ob_len,
                            int own_ob, int own_g, int own_r, int c_from, int c_to, double *Lx, double *Uout)
{
    int rows = max_m - ob - 1;
    if (rows <= 0 || c_from >= c_to) return;
    hipLaunchKernelGGL(k_syrk_trailing128<true>, syrk128_grid(rows, 1), dim3(512), 0, st, ds, list, ob, ob_len, Lx, Uout,
                       (double *)nullptr, (int *)nullptr, ColOwner{own_ob, own_g, own_r, c_from, c_to, 0});
}

// ------------------------------------------------------------------------------------------
// Big-front forward solve.  wk (n doubles per rhs) carries the running right-hand side of the
// pivot rows; the update rows accumulate directly in the level's update-vector buffer.
// Super-step over up to SB = 256 pivot columns [jb0, jb0 + nb).  The solves of the top fronts are a
// chain of dependent launches, so the step is built for latency: every workgroup (1024 threads) solves
// the nb x nb diagonal part redundantly and then updates its own 256 rows below the super-block.
// All global operands of the diagonal solve -- the four 64 x 64 inverses and the six sub-diagonal
// 64 x 64 blocks of L, 40 doubles per thread -- are loaded into registers up front (they do not depend
// on the running vector); the dependent chain  y_s = Linv_s w_s,  w_t -= L(t, s) y_s (t > s)  then
// runs out of registers and LDS only.  Lane = row, wave = 4 columns; partial sums meet in LDS in a
// fixed order (bitwise reproducible).
constexpr int SB = 256;
constexpr int SOLVE_NT = 1024;

// sub-diagonal block (ib, s), ib > s, of the 4 x 4 block lower triangle -> 0..5
__device__ __forceinline__ constexpr int sblk(int ib, int s) { return ib * (ib - 1) / 2 + s; }

// FIRST = true is the step of the first 256 columns and also assembles the front's right-hand side:
// every workgroup gathers, for the diagonal rows (redundantly) and for its own 256 rows below, the
// entries of x (pivot rows) and the children's update vectors (parent-pull, children in sequence) in
// LDS -- no separate initialisation launch in front of the dependent chain.
// ROWS = rows below the super-block per workgroup, 256 or 64.  One CU streams a panel at ~25 GB/s: the 256 x 256 doubles of a
// 256-row workgroup are 512 KB = 20 us of one CU's time, four dependent rounds of 16 loads per lane.  Where a launch holds few fronts
// (the top of the tree, where the solves are a chain of these steps) 64 rows per workgroup spread the same bytes over four times
// as many CUs and every lane needs ONE round of at most 16 loads; where a level holds hundreds of fronts the wide form repeats the
// diagonal chain less often (launch_fwd_big chooses).  The partial sums of a row meet in LDS in a fixed order in both forms.
template <bool FIRST, int ROWS>
__global__ __launch_bounds__(SOLVE_NT) void k_fwd_big_step(DevSym ds, const int32_t *__restrict__ list, int jb0,
                                                           const double *__restrict__ Lx, const double *__restrict__ Linv,
                                                           double *__restrict__ X, const double *__restrict__ X0, int64_t ldx,
                                                           double *__restrict__ WK, int64_t ldw,
                                                           const double *__restrict__ Wc,
                                                           double *__restrict__ Wo, int64_t wstride)
{
    unsigned bx, by, rh;
    kvx_part_front_rhs(bx, by, rh);
    __shared__ double red[3 * 16 * NB];
    __shared__ double own[FIRST ? ROWS : 1];
    __shared__ double wsh[SB];
    __shared__ double ysh[SB];
    const FrontDesc fd = ds.fd[list[by]];
    const int k = fd.k, m = fd.m, f = fd.first, tid = threadIdx.x;
    if (jb0 >= k) return;
    const int nb = min(SB, k - jb0);
    const int rbase = jb0 + nb + bx * ROWS;
    if (bx > 0 && rbase >= m) return;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double *P = Lx + fd.px;
    const double *Y = Linv + fd.linv + (int64_t)(jb0 / NB) * NB * NB;
    double *x = X + (int64_t)rh * ldx + f;
    double *wk = WK + (int64_t)rh * ldw + f;
    double *wo = Wo + (int64_t)rh * wstride + fd.wx;
    const int nsub = (nb + NB - 1) / NB;

    double yI[4][4], lB[6][4];
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int p = 4 * w + c;
            yI[s][c] = kvx_ld0(Y, s * NB * NB + lane + p * NB, s * NB + lane < nb && p <= lane);
        }
#pragma unroll
    for (int ib = 1; ib < 4; ib++)
#pragma unroll
        for (int s = 0; s < ib; s++)
#pragma unroll
            for (int c = 0; c < 4; c++)
                lB[sblk(ib, s)][c] = kvx_ld0(P, (int64_t)(jb0 + ib * NB + lane) + (int64_t)(jb0 + s * NB + 4 * w + c) * m,
                                             ib * NB + lane < nb);
    if (tid < SB) {
        const double *x0 = X0 + (int64_t)rh * ldx + f;     // rhs as it was before the sweep (x gets y meanwhile)
        wsh[tid] = kvx_ld0(FIRST ? x0 : wk, jb0 + tid, tid < nb);
        ysh[tid] = 0.0;
        if (FIRST && tid < ROWS) own[tid] = kvx_ld0(x0, rbase + tid, rbase + tid < k);
    }
    __syncthreads();
    if (FIRST && fd.nchild > 0) {
        const double *wc = Wc + (int64_t)rh * wstride;
        ChildDesc cd = ds.cd[fd.childptr];
        for (int c = 0; c < fd.nchild; c++) {
            ChildDesc nx = cd;
            if (c + 1 < fd.nchild) nx = ds.cd[fd.childptr + c + 1];
            const int32_t *rl = ds.rel + cd.rel;
            const double *src = wc + cd.wx;
            for (int i = tid; i < cd.uc; i += SOLVE_NT) {
                const int t = rl[i];
                const double v = src[i];
                if (t < nb) wsh[t] += v;
                else if (t >= rbase && t < rbase + ROWS) own[t - rbase] += v;
            }
            __syncthreads();
            cd = nx;
        }
    }
#pragma unroll
    for (int s = 0; s < 4; s++) {
        if (s < nsub) {
            double part = 0.0;
#pragma unroll
            for (int c = 0; c < 4; c++) part = __builtin_fma(yI[s][c], wsh[s * NB + 4 * w + c], part);
            red[w * NB + lane] = part;
            __syncthreads();
            if (tid < NB) {
                double t = 0.0;
#pragma unroll
                for (int q = 0; q < 16; q++) t += red[q * NB + tid];
                ysh[s * NB + tid] = t;
            }
            __syncthreads();
            if (s + 1 < nsub) {
#pragma unroll
                for (int ib = s + 1; ib < 4; ib++) {
                    double pp = 0.0;
#pragma unroll
                    for (int c = 0; c < 4; c++) pp = __builtin_fma(lB[sblk(ib, s)][c], ysh[s * NB + 4 * w + c], pp);
                    red[((ib - s - 1) * 16 + w) * NB + lane] = pp;
                }
                __syncthreads();
                if (tid < (3 - s) * NB) {
                    const int ib0 = tid >> 6;               // 0 .. 2-s
                    double t = 0.0;
#pragma unroll
                    for (int q = 0; q < 16; q++) t += red[(ib0 * 16 + q) * NB + lane];
                    wsh[(s + 1) * NB + tid] -= t;
                }
                __syncthreads();
            }
        }
    }
    if (bx == 0 && tid < nb) x[jb0 + tid] = ysh[tid];
    // rows below the super-block: thread = (row, quarter of the nb columns), partial sums meet in LDS
    constexpr int NQ = SOLVE_NT / ROWS;            // column groups: 4 quarters (ROWS = 256) or 16 groups (ROWS = 64)
    const int rr = tid % ROWS;
    const int q = __builtin_amdgcn_readfirstlane(tid / ROWS);
    const int r = rbase + rr;
    double acc = 0.0;
    const int cq = ROWS == 256 ? (((nb + 63) >> 6) << 4) : ((nb + 15) >> 4);   // columns per group: a multiple of 16 (16 .. 64) / 1 .. 16
    if (q * cq < nb && rbase < m) {
        const bool okr = r < m;
        const double *Pr = P + (okr ? r : 0) + (int64_t)(jb0 + q * cq) * m;
        const double *yq = ysh + q * cq;
        const int nc = min(cq, nb - q * cq);
#pragma unroll 1
        for (int j0 = 0; j0 < nc; j0 += 16) {
            double v[16];
#pragma unroll
            for (int j = 0; j < 16; j++) v[j] = kvx_ld0(Pr, (int64_t)(j0 + j) * m, okr && j0 + j < nc);
#pragma unroll
            for (int j = 0; j < 16; j++) acc = __builtin_fma(v[j], yq[j0 + j], acc);
        }
    }
    red[q * ROWS + rr] = acc;
    __syncthreads();
    if (tid < ROWS && r < m) {
        double t;
        if (NQ == 4) {
            t = (red[rr] + red[ROWS + rr]) + (red[2 * ROWS + rr] + red[3 * ROWS + rr]);
        } else {
            t = 0.0;
#pragma unroll
            for (int g = 0; g < NQ; g++) t += red[g * ROWS + rr];
        }
        if (FIRST) {
            if (r < k) wk[r] = own[rr] - t;
            else wo[r - k] = own[rr] - t;
        } else {
            if (r < k) wk[r] -= t;
            else wo[r - k] -= t;
        }
    }
}

// Big-front backward solve: t = y - L21' x_below (one wave per pivot column), then block steps
// from the last block to the first.
__global__ __launch_bounds__(256) void k_bwd_big_init(DevSym ds, const int32_t *__restrict__ list,
                                                      const double *__restrict__ Lx, const double *__restrict__ X,
                                                      int64_t ldx, double *__restrict__ WK, int64_t ldw)
{
    unsigned bx, by, rh;
    kvx_part_front_rhs(bx, by, rh);
    const int s = list[by];
    const int k = ds.k[s], m = ds.m[s], f = ds.first[s];
    const int c = bx * 4 + (threadIdx.x >> 6), ln = threadIdx.x & 63;
    if (c >= k) return;
    const double *x = X + (int64_t)rh * ldx;
    double *wk = WK + (int64_t)rh * ldw + f;
    const double *Pc = Lx + ds.px[s] + (int64_t)c * m;
    const int32_t *rows = ds.rowidx + ds.rowptr[s];
    double acc = 0.0;
    for (int i = k + ln; i < m; i += 64) acc += Pc[i] * x[rows[i]];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (ln == 0) wk[c] = x[f + c] - acc;
}

// sum over the 64 lanes of four values at once: after the call the lanes 0 .. 3 hold the total of v[o],
// o = 2 * (lane & 1) + (lane >> 1) (the other lanes of a class (lane & 3) hold it up to the order of the additions).
// Halving exchange first (4 -> 2 -> 1 values per lane), then the sum over the sixteen quads; fixed order.  Round 3: the
// exchanges inside a row of 16 lanes are DPP moves (quad permutes, row rotations by 4 and 8) and the two across the rows
// gfx950's permlane swaps -- a __shfl_xor is two ds_bpermute, an LDS crossbar round trip each, and a backward super-step
// chains up to ten of these calls.
// the sum over the four 16-lane rows, position by position, in every lane: r[i] + r[i ^ 16] first, then + the same of i ^ 32 (the
// order of the two shuffles this replaces).  v_permlane16_swap / v_permlane32_swap are gfx950 VALU instructions (no LDS crossbar).
__device__ __forceinline__ double rows_sum_f64(double r)
{
    const int lo = __double2loint(r), hi = __double2hiint(r);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double s = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    const int lo2 = __double2loint(s), hi2 = __double2hiint(s);
    const auto c = __builtin_amdgcn_permlane32_swap(lo2, lo2, false, false);
    const auto d = __builtin_amdgcn_permlane32_swap(hi2, hi2, false, false);
    return __hiloint2double(d[0], c[0]) + __hiloint2double(d[1], c[1]);
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum4(double v0, double v1, double v2, double v3, int lane)
{
    const bool b0 = lane & 1, b1 = lane & 2;
    const double s0 = b0 ? v0 : v2, s1 = b0 ? v1 : v3;
    const double k0 = (b0 ? v2 : v0) + dpp_mov_f64<0xB1>(s0);                   // quad_perm [1, 0, 3, 2]: lane ^ 1
    const double k1 = (b0 ? v3 : v1) + dpp_mov_f64<0xB1>(s1);
    double r = (b1 ? k1 : k0) + dpp_mov_f64<0x4E>(b1 ? k0 : k1);                // quad_perm [2, 3, 0, 1]: lane ^ 2
    r += dpp_mov_f64<0x124>(r);                                                 // row_ror 4
    r += dpp_mov_f64<0x128>(r);                                                 // row_ror 8: the row's four quads
    return rows_sum_f64(r);
}

// Backward super-step over the pivot columns [jb0, jb0 + nb), nb <= 256, the mirror image of the
// forward one:  x_s = Linv_s' t_s  from the last 64-column sub-block to the first,  t_c -= L(s, c)' x_s
// for the earlier sub-blocks, operands prefetched into registers.  The products are transposed ones,
// so lane = row (the summation index, contiguous in memory) and wave = 4 outputs, summed across the
// wave with wave_sum4.  Workgroup x then owns 64 earlier pivot columns of the front:
// t_c -= L(b, c)' x_b over the nb rows of the super-block.
__global__ __launch_bounds__(SOLVE_NT) void k_bwd_big_step(DevSym ds, const int32_t *__restrict__ list, int sidx,
                                                           const double *__restrict__ Lx, const double *__restrict__ Linv,
                                                           double *__restrict__ X, int64_t ldx,
                                                           double *__restrict__ WK, int64_t ldw)
{
    unsigned bx, by, rh;
    kvx_part_front_rhs(bx, by, rh);
    __shared__ double tsh[SB];
    __shared__ double xsh[SB];
    const FrontDesc fd = ds.fd[list[by]];
    const int k = fd.k, m = fd.m, f = fd.first, tid = threadIdx.x;
    const int jb0 = sidx * SB;
    if (jb0 >= k) return;
    const int nb = min(SB, k - jb0);
    if (bx > 0 && (int)bx * NB >= jb0) return;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int o = 2 * (lane & 1) + ((lane >> 1) & 1);        // the output wave_sum4 leaves in this lane
    const double *P = Lx + fd.px;
    const double *Y = Linv + fd.linv + (int64_t)(jb0 / NB) * NB * NB;
    double *x = X + (int64_t)rh * ldx + f;
    double *wk = WK + (int64_t)rh * ldw + f;
    const int nsub = (nb + NB - 1) / NB;

    double yI[4][4], lB[6][4];
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int i = 4 * w + c;                          // output (column of Linv_s), lane = row p >= i
            yI[s][c] = kvx_ld0(Y, s * NB * NB + lane + i * NB, s * NB + lane < nb && i <= lane);
        }
#pragma unroll
    for (int ib = 1; ib < 4; ib++)
#pragma unroll
        for (int s = 0; s < ib; s++)
#pragma unroll
            for (int c = 0; c < 4; c++)
                lB[sblk(ib, s)][c] = kvx_ld0(P, (int64_t)(jb0 + ib * NB + lane) + (int64_t)(jb0 + s * NB + 4 * w + c) * m,
                                             ib * NB + lane < nb);
    if (tid < SB) {
        tsh[tid] = kvx_ld0(wk, jb0 + tid, tid < nb);
        xsh[tid] = 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int s = 3; s >= 0; s--) {
        if (s < nsub) {
            const double tv = tsh[s * NB + lane];
            const double r = wave_sum4(yI[s][0] * tv, yI[s][1] * tv, yI[s][2] * tv, yI[s][3] * tv, lane);
            if (lane < 4) xsh[s * NB + 4 * w + o] = r;
            __syncthreads();
            if (s > 0) {
                const double xv = xsh[s * NB + lane];
#pragma unroll
                for (int c = 0; c < s; c++) {
                    const double u = wave_sum4(lB[sblk(s, c)][0] * xv, lB[sblk(s, c)][1] * xv, lB[sblk(s, c)][2] * xv,
                                               lB[sblk(s, c)][3] * xv, lane);
                    if (lane < 4) tsh[c * NB + 4 * w + o] -= u;
                }
                __syncthreads();
            }
        }
    }
    if (bx == 0 && tid < nb) x[jb0 + tid] = xsh[tid];
    // earlier pivot columns c0 + 4w .. +3 of the front, rows jb0 + lane + 64 j
    const int c0 = bx * NB + 4 * w;
    if (c0 < jb0) {
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        double v[4][4];
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                v[c][j] = kvx_ld0(P, (int64_t)(jb0 + lane + j * NB) + (int64_t)(c0 + c) * m, lane + j * NB < nb && c0 + c < jb0);
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int j = 0; j < 4; j++) a[c] = __builtin_fma(v[c][j], xsh[lane + j * NB], a[c]);
        const double u = wave_sum4(a[0], a[1], a[2], a[3], lane);
        if (lane < 4 && c0 + o < jb0) wk[c0 + o] -= u;
    }
}

// ------------------------------------------------------------------------------------------
// The same three kernels for a BLOCK of RB right-hand sides per workgroup (used from KVX_BIG_MR_FROM right-hand sides on).  With the
// right-hand sides spread over the grid every workgroup fetches the diagonal inverses, the sub-diagonal blocks, the panel rows
// and the children's relative indices again; here they are fetched once per block and the right-hand sides of the block follow
// each other through the dependent part.  Same operations in the same order per right-hand side as the 256-row form of the single-rhs step: every column is bit-identical
// to its single-rhs solve.  Right-hand sides past nrhs (ragged last block) are skipped.
// MODE 1: the diagonal solve only (grid x = 1; writes y into X).  MODE 2: the rows below only (reads y back from X) -- with the GPU
// saturated by right-hand sides, every row block repeating the diagonal solve (as the single-rhs kernel does to save a launch)
// would be the bulk of the work.
template <bool FIRST, int RB, int MODE>
__global__ __launch_bounds__(SOLVE_NT) void k_fwd_big_step_mr(DevSym ds, const int32_t *__restrict__ list, int jb0,
                                                              const double *__restrict__ Lx, const double *__restrict__ Linv,
                                                              double *__restrict__ X, const double *__restrict__ X0, int64_t ldx, int nrhs,
                                                              double *__restrict__ WK, int64_t ldw,
                                                              const double *__restrict__ Wc,
                                                              double *__restrict__ Wo, int64_t wstride)
{
    unsigned bx, by, rg;
    kvx_part_front_rhs(bx, by, rg);
    __shared__ double red[3 * 16 * NB];
    __shared__ double own[FIRST ? RB : 1][FIRST ? 256 : 1];
    __shared__ double wsh[SB];
    __shared__ double ysh[RB][SB];
    const FrontDesc fd = ds.fd[list[by]];
    const int k = fd.k, m = fd.m, f = fd.first, tid = threadIdx.x;
    if (jb0 >= k) return;
    const int nb = min(SB, k - jb0);
    const int rbase = jb0 + nb + bx * 256;
    if (MODE == 2 && rbase >= m) return;
    const int r0 = (int)rg * RB, nv = min(RB, nrhs - r0);
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double *P = Lx + fd.px;
    const double *Y = Linv + fd.linv + (int64_t)(jb0 / NB) * NB * NB;
    const int nsub = (nb + NB - 1) / NB;

    double yI[4][4], lB[6][4];
    if (MODE == 1) {
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int p = 4 * w + c;
                yI[s][c] = kvx_ld0(Y, s * NB * NB + lane + p * NB, s * NB + lane < nb && p <= lane);
            }
#pragma unroll
        for (int ib = 1; ib < 4; ib++)
#pragma unroll
            for (int s = 0; s < ib; s++)
#pragma unroll
                for (int c = 0; c < 4; c++)
                    lB[sblk(ib, s)][c] = kvx_ld0(P, (int64_t)(jb0 + ib * NB + lane) + (int64_t)(jb0 + s * NB + 4 * w + c) * m,
                                                 ib * NB + lane < nb);
    }
    for (int b = 0; b < nv; b++) {                 // workgroup-uniform
        const int rh = r0 + b;
        double *x = X + (int64_t)rh * ldx + f;
        double *wk = WK + (int64_t)rh * ldw + f;
        __syncthreads();                           // wsh / red of the previous right-hand side are free
        if (tid < SB) {
            const double *x0 = X0 + (int64_t)rh * ldx + f;
            if (MODE == 1) {
                wsh[tid] = kvx_ld0(FIRST ? x0 : wk, jb0 + tid, tid < nb);
                ysh[b][tid] = 0.0;
            } else {
                ysh[b][tid] = kvx_ld0(x, jb0 + tid, tid < nb);      // y of this super-step, written by the MODE 1 launch
                if (FIRST) own[b][tid] = kvx_ld0(x0, rbase + tid, rbase + tid < k);
            }
        }
        __syncthreads();
        if (FIRST && fd.nchild > 0) {
            const double *wc = Wc + (int64_t)rh * wstride;
            ChildDesc cd = ds.cd[fd.childptr];
            for (int c = 0; c < fd.nchild; c++) {
                ChildDesc nx = cd;
                if (c + 1 < fd.nchild) nx = ds.cd[fd.childptr + c + 1];
                const int32_t *rl = ds.rel + cd.rel;
                const double *src = wc + cd.wx;
                for (int i = tid; i < cd.uc; i += SOLVE_NT) {
                    const int t = rl[i];
                    const double v = src[i];
                    if (MODE == 1) { if (t < nb) wsh[t] += v; }
                    else if (t >= rbase && t < rbase + 256) own[b][t - rbase] += v;
                }
                __syncthreads();
                cd = nx;
            }
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (MODE == 1 && s < nsub) {
                double part = 0.0;
#pragma unroll
                for (int c = 0; c < 4; c++) part = __builtin_fma(yI[s][c], wsh[s * NB + 4 * w + c], part);
                red[w * NB + lane] = part;
                __syncthreads();
                if (tid < NB) {
                    double t = 0.0;
#pragma unroll
                    for (int q = 0; q < 16; q++) t += red[q * NB + tid];
                    ysh[b][s * NB + tid] = t;
                }
                __syncthreads();
                if (s + 1 < nsub) {
#pragma unroll
                    for (int ib = s + 1; ib < 4; ib++) {
                        double pp = 0.0;
#pragma unroll
                        for (int c = 0; c < 4; c++) pp = __builtin_fma(lB[sblk(ib, s)][c], ysh[b][s * NB + 4 * w + c], pp);
                        red[((ib - s - 1) * 16 + w) * NB + lane] = pp;
                    }
                    __syncthreads();
                    if (tid < (3 - s) * NB) {
                        const int ib0 = tid >> 6;
                        double t = 0.0;
#pragma unroll
                        for (int q = 0; q < 16; q++) t += red[(ib0 * 16 + q) * NB + lane];
                        wsh[(s + 1) * NB + tid] -= t;
                    }
                    __syncthreads();
                }
            }
        }
        if (MODE == 1 && tid < nb) x[jb0 + tid] = ysh[b][tid];
    }
    if (MODE == 1) return;
    __syncthreads();
    // rows below the super-block: the panel entries are loaded once and used for every right-hand side of the block
    const int rr = tid & 255;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int r = rbase + rr;
    double acc[RB];
#pragma unroll
    for (int b = 0; b < RB; b++) acc[b] = 0.0;
    const int cq = ((nb + 63) >> 6) << 4;
    if (q * cq < nb && rbase < m) {
        const bool okr = r < m;
        const double *Pr = P + (okr ? r : 0) + (int64_t)(jb0 + q * cq) * m;
        const int nc = min(cq, nb - q * cq);
#pragma unroll 1
        for (int j0 = 0; j0 < nc; j0 += 16) {
            double v[16];
#pragma unroll
            for (int j = 0; j < 16; j++) v[j] = kvx_ld0(Pr, (int64_t)(j0 + j) * m, okr && j0 + j < nc);
#pragma unroll
            for (int b = 0; b < RB; b++) {
                const double *yq = ysh[b] + q * cq;
#pragma unroll
                for (int j = 0; j < 16; j++) acc[b] = __builtin_fma(v[j], yq[j0 + j], acc[b]);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < RB; b++) {
        if (b < nv) {                              // workgroup-uniform
            __syncthreads();
            red[q * 256 + rr] = acc[b];
            __syncthreads();
            if (tid < 256 && r < m) {
                const int rh = r0 + b;
                double *wk = WK + (int64_t)rh * ldw + f;
                double *wo = Wo + (int64_t)rh * wstride + fd.wx;
                const double t = (red[rr] + red[256 + rr]) + (red[512 + rr] + red[768 + rr]);
                if (FIRST) {
                    if (r < k) wk[r] = own[b][rr] - t;
                    else wo[r - k] = own[b][rr] - t;
                } else {
                    if (r < k) wk[r] -= t;
                    else wo[r - k] -= t;
                }
            }
        }
    }
}

template <int RB>
__global__ __launch_bounds__(256) void k_bwd_big_init_mr(DevSym ds, const int32_t *__restrict__ list,
                                                         const double *__restrict__ Lx, const double *__restrict__ X,
                                                         int64_t ldx, int nrhs, double *__restrict__ WK, int64_t ldw)
{
    unsigned bx, by, rg;
    kvx_part_front_rhs(bx, by, rg);
    const int s = list[by];
    const int k = ds.k[s], m = ds.m[s], f = ds.first[s];
    const int c = bx * 4 + (threadIdx.x >> 6), ln = threadIdx.x & 63;
    if (c >= k) return;
    const int r0 = (int)rg * RB, nv = min(RB, nrhs - r0);
    const double *Pc = Lx + ds.px[s] + (int64_t)c * m;
    const int32_t *rows = ds.rowidx + ds.rowptr[s];
    const double *xb[RB];
#pragma unroll
    for (int b = 0; b < RB; b++) xb[b] = X + (int64_t)(r0 + min(b, nv - 1)) * ldx;
    double acc[RB];
#pragma unroll
    for (int b = 0; b < RB; b++) acc[b] = 0.0;
    for (int i = k + ln; i < m; i += 64) {
        const double pv = Pc[i];
        const int ri = rows[i];
#pragma unroll
        for (int b = 0; b < RB; b++) acc[b] += pv * xb[b][ri];
    }
#pragma unroll
    for (int b = 0; b < RB; b++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[b] += __shfl_xor(acc[b], o);
        if (ln == 0 && b < nv) WK[(int64_t)(r0 + b) * ldw + f + c] = xb[b][f + c] - acc[b];
    }
}

template <int RB, int MODE>      // MODE 1: diagonal solve (grid x = 1); MODE 2: the earlier pivot columns (reads x back from X)
__global__ __launch_bounds__(SOLVE_NT) void k_bwd_big_step_mr(DevSym ds, const int32_t *__restrict__ list, int sidx,
                                                              const double *__restrict__ Lx, const double *__restrict__ Linv,
                                                              double *__restrict__ X, int64_t ldx, int nrhs,
                                                              double *__restrict__ WK, int64_t ldw)
{
    unsigned bx, by, rg;
    kvx_part_front_rhs(bx, by, rg);
    __shared__ double tsh[SB];
    __shared__ double xsh[RB][SB];
    const FrontDesc fd = ds.fd[list[by]];
    const int k = fd.k, m = fd.m, f = fd.first, tid = threadIdx.x;
    const int jb0 = sidx * SB;
    if (jb0 >= k) return;
    const int nb = min(SB, k - jb0);
    if (MODE == 2 && (int)bx * NB >= jb0) return;
    const int r0 = (int)rg * RB, nv = min(RB, nrhs - r0);
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int o = 2 * (lane & 1) + ((lane >> 1) & 1);
    const double *P = Lx + fd.px;
    const double *Y = Linv + fd.linv + (int64_t)(jb0 / NB) * NB * NB;
    const int nsub = (nb + NB - 1) / NB;

    double yI[4][4], lB[6][4];
    if (MODE == 1) {
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int i = 4 * w + c;
                yI[s][c] = kvx_ld0(Y, s * NB * NB + lane + i * NB, s * NB + lane < nb && i <= lane);
            }
#pragma unroll
        for (int ib = 1; ib < 4; ib++)
#pragma unroll
            for (int s = 0; s < ib; s++)
#pragma unroll
                for (int c = 0; c < 4; c++)
                    lB[sblk(ib, s)][c] = kvx_ld0(P, (int64_t)(jb0 + ib * NB + lane) + (int64_t)(jb0 + s * NB + 4 * w + c) * m,
                                                 ib * NB + lane < nb);
    }
    for (int b = 0; b < nv; b++) {                 // workgroup-uniform
        const int rh = r0 + b;
        double *x = X + (int64_t)rh * ldx + f;
        const double *wk = WK + (int64_t)rh * ldw + f;
        __syncthreads();
        if (tid < SB) {
            if (MODE == 1) {
                tsh[tid] = kvx_ld0(wk, jb0 + tid, tid < nb);
                xsh[b][tid] = 0.0;
            } else {
                xsh[b][tid] = kvx_ld0(x, jb0 + tid, tid < nb);      // x of this super-step, written by the MODE 1 launch
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 3; s >= 0; s--) {
            if (MODE == 1 && s < nsub) {
                const double tv = tsh[s * NB + lane];
                const double r = wave_sum4(yI[s][0] * tv, yI[s][1] * tv, yI[s][2] * tv, yI[s][3] * tv, lane);
                if (lane < 4) xsh[b][s * NB + 4 * w + o] = r;
                __syncthreads();
                if (s > 0) {
                    const double xv = xsh[b][s * NB + lane];
#pragma unroll
                    for (int c = 0; c < s; c++) {
                        const double u = wave_sum4(lB[sblk(s, c)][0] * xv, lB[sblk(s, c)][1] * xv, lB[sblk(s, c)][2] * xv,
                                                   lB[sblk(s, c)][3] * xv, lane);
                        if (lane < 4) tsh[c * NB + 4 * w + o] -= u;
                    }
                    __syncthreads();
                }
            }
        }
        if (MODE == 1 && tid < nb) x[jb0 + tid] = xsh[b][tid];
    }
    if (MODE == 1) return;
    __syncthreads();
    const int c0 = bx * NB + 4 * w;
    if (c0 < jb0) {
        double v[4][4];
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                v[c][j] = kvx_ld0(P, (int64_t)(jb0 + lane + j * NB) + (int64_t)(c0 + c) * m, lane + j * NB < nb && c0 + c < jb0);
#pragma unroll
        for (int b = 0; b < RB; b++) {
            if (b < nv) {                          // workgroup-uniform
                double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int j = 0; j < 4; j++) a[c] = __builtin_fma(v[c][j], xsh[b][lane + j * NB], a[c]);
                const double u = wave_sum4(a[0], a[1], a[2], a[3], lane);
                if (lane < 4 && c0 + o < jb0) WK[(int64_t)(r0 + b) * ldw + f + c0 + o] -= u;
            }
        }
    }
}

constexpr int KVX_BIG_RB = 8;          // right-hand sides per workgroup in the blocked big-front steps
// Measured (MI355X, scratch/multirhs.py): the block makes one workgroup's dependent chain RB times longer, so it pays only once the
// grid saturates the GPU anyway -- n = 1e6, 64 rhs: 21.7 -> 19.1 ms; n = 50 000, 200 rhs: 3.04 -> 2.90 ms; but 32 rhs at
// n = 50 000: 0.90 -> 1.24 ms.  Hence the threshold.
constexpr int KVX_BIG_MR_FROM = 64;    // used from this many right-hand sides on

void launch_fwd_big(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int max_k,
                    const double *Lx, const double *Linv, double *X, const double *X0, int64_t ldx, int nrhs,
                    double *WK, int64_t ldw, const double *Wchild, double *Wout, int64_t wstride, int level_count)
{
    if (count <= 0 || nrhs <= 0) return;
    // 64-row workgroups while the launch stays within ~two workgroups per CU (KVX_FWD_NARROW_WGS, 0 = never), 256-row ones beyond
    static const int narrow_wgs = [] { const char *e = getenv("KVX_FWD_NARROW_WGS"); return e ? atoi(e) : 512; }();
    if (nrhs >= KVX_BIG_MR_FROM) {
        for (int jb = 0; jb < max_k; jb += SB) {
            int rows = max_m - jb - 1;
            // The rhs-blocked kernels add a row's partial sums in the association of the 256-row form.  Where the single-rhs sweep
            // of this level takes 64-row workgroups (sixteen partials added in sequence) they would give a column other bits than
            // a solve with fewer right-hand sides does: those steps -- the top of the tree, a handful of fronts -- go to the
            // single-rhs kernel, one grid layer per right-hand side (round-3 advisor finding; test_solution_bits_do_not_depend_on_nrhs).
            if ((int64_t)std::max(1, (rows + 63) / 64) * std::max(level_count, count) <= narrow_wgs) {
                dim3 grid((unsigned)std::max(1, (rows + 63) / 64), (unsigned)count, (unsigned)nrhs);
                if (jb == 0) hipLaunchKernelGGL((k_fwd_big_step<true, 64>), grid, dim3(SOLVE_NT), 0, st, ds, list, jb, Lx, Linv, X, X0, ldx, WK, ldw, Wchild, Wout, wstride);
                else hipLaunchKernelGGL((k_fwd_big_step<false, 64>), grid, dim3(SOLVE_NT), 0, st, ds, list, jb, Lx, Linv, X, X0, ldx, WK, ldw, Wchild, Wout, wstride);
                continue;
            }
            const unsigned gz = (unsigned)((nrhs + KVX_BIG_RB - 1) / KVX_BIG_RB);
            // (blocks of 1 for the diagonal launch -- more, shorter workgroups -- were measured and lost: 18.5 -> 20.4 ms at 64 rhs,
            // n = 1e6: a 1024-thread workgroup per front and right-hand side is the cost, not the length of its chain)
            dim3 g1(1, (unsigned)count, gz), g2((unsigned)std::max(1, (rows + 255) / 256), (unsigned)count, gz);
            if (jb == 0) {
                hipLaunchKernelGGL((k_fwd_big_step_mr<true, KVX_BIG_RB, 1>), g1, dim3(SOLVE_NT), 0, st, ds, list, jb, Lx, Linv, X, X0, ldx, nrhs, WK, ldw, Wchild, Wout, wstride);
                hipLaunchKernelGGL((k_fwd_big_step_mr<true, KVX_BIG_RB, 2>), g2, dim3(SOLVE_NT), 0, st, ds, list, jb, Lx, Linv, X, X0, ldx, nrhs, WK, ldw, Wchild, Wout, wstride);
            } else {
                hipLaunchKernelGGL((k_fwd_big_step_mr<false, KVX_BIG_RB, 1>), g1, dim3(SOLVE_NT), 0, st, ds, list, jb, Lx, Linv, X, X0, ldx, nrhs, WK, ldw, Wchild, Wout, wstride);
                hipLaunchKernelGGL((k_fwd_big_step_mr<false, KVX_BIG_RB, 2>), g2, dim3(SOLVE_NT), 0, st, ds, list, jb, Lx, Linv, X, X0, ldx, nrhs, WK, ldw, Wchild, Wout, wstride);
            }
        }
        return;
    }
    for (int jb = 0; jb < max_k; jb += SB) {
        int rows = max_m - jb - 1;
        // (a function of the LEVEL -- its big fronts, all of them -- not of this launch: a sweep over a part of the level (spsolve's
        // reach) and a sweep with more right-hand sides must sum every row in the same order as the full single-rhs sweep)
        const int64_t wg64 = (int64_t)std::max(1, (rows + 63) / 64) * std::max(level_count, count);
        const bool narrow = wg64 <= narrow_wgs;
        dim3 grid((unsigned)std::max(1, narrow ? (rows + 63) / 64 : (rows + 255) / 256), (unsigned)count, (unsigned)nrhs);
        if (jb == 0) {
            if (narrow) hipLaunchKernelGGL((k_fwd_big_step<true, 64>), grid, dim3(SOLVE_NT), 0, st, ds, list, jb, Lx, Linv, X, X0, ldx, WK, ldw, Wchild, Wout, wstride);
            else hipLaunchKernelGGL((k_fwd_big_step<true, 256>), grid, dim3(SOLVE_NT), 0, st, ds, list, jb, Lx, Linv, X, X0, ldx, WK, ldw, Wchild, Wout, wstride);
        } else {
            if (narrow) hipLaunchKernelGGL((k_fwd_big_step<false, 64>), grid, dim3(SOLVE_NT), 0, st, ds, list, jb, Lx, Linv, X, X0, ldx, WK, ldw, Wchild, Wout, wstride);
            else hipLaunchKernelGGL((k_fwd_big_step<false, 256>), grid, dim3(SOLVE_NT), 0, st, ds, list, jb, Lx, Linv, X, X0, ldx, WK, ldw, Wchild, Wout, wstride);
        }
    }
}

void launch_bwd_big(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int max_k,
                    const double *Lx, const double *Linv, double *X, int64_t ldx, int nrhs, double *WK, int64_t ldw)
{
    if (count <= 0 || nrhs <= 0) return;
    (void)max_m;
    if (nrhs >= KVX_BIG_MR_FROM) {
        const unsigned gz = (unsigned)((nrhs + KVX_BIG_RB - 1) / KVX_BIG_RB);
        hipLaunchKernelGGL((k_bwd_big_init_mr<KVX_BIG_RB>), dim3((unsigned)((max_k + 3) / 4), (unsigned)count, gz), dim3(256), 0, st,
                           ds, list, Lx, X, ldx, nrhs, WK, ldw);
        for (int b = (max_k + SB - 1) / SB - 1; b >= 0; b--) {
            hipLaunchKernelGGL((k_bwd_big_step_mr<KVX_BIG_RB, 1>), dim3(1, (unsigned)count, gz), dim3(SOLVE_NT), 0, st, ds, list, b, Lx, Linv,
                               X, ldx, nrhs, WK, ldw);
            if (b > 0)
                hipLaunchKernelGGL((k_bwd_big_step_mr<KVX_BIG_RB, 2>), dim3((unsigned)(b * SB / NB), (unsigned)count, gz), dim3(SOLVE_NT), 0, st, ds, list, b, Lx, Linv,
                                   X, ldx, nrhs, WK, ldw);
        }
        return;
    }
    hipLaunchKernelGGL(k_bwd_big_init, dim3((unsigned)((max_k + 3) / 4), (unsigned)count, (unsigned)nrhs), dim3(256), 0, st,
                       ds, list, Lx, X, ldx, WK, ldw);
    for (int b = (max_k + SB - 1) / SB - 1; b >= 0; b--) {
        unsigned gx = (unsigned)std::max(1, b * SB / NB);
        hipLaunchKernelGGL(k_bwd_big_step, dim3(gx, (unsigned)count, (unsigned)nrhs), dim3(SOLVE_NT), 0, st, ds, list, b, Lx, Linv,
                           X, ldx, WK, ldw);
    }
}

}  // namespace kvx
