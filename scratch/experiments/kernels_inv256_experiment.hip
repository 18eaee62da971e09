// EXPERIMENT, NOT BUILT (round 2; kept for the record, see DESIGN.md section 4 "measured and not adopted").
// Measured on MI355X, config 2 (n = 1e6, 1 rhs), rocprofv3 timeline: the chain-free super-steps below take 13 us at the top levels
// (k_fwd_big_step 23.5 us) but 61 us where a level holds hundreds of fronts (38 us: every 64-row workgroup takes a whole CU), and
// k_inv256 as written costs 50-100 us per level (bound by the latency of its operand loads): solve 1.475 -> 1.42 ms, factor
// 3.79 -> 4.25 ms.  Restricted to the top four levels the net gain would be ~0.1 ms; it also ends the bit-identity of the
// 64-rhs blocked kernels with the single-rhs ones.  Dropped.
//
// Big-front triangular solves with one right-hand side (or a few): 256-column super-steps WITHOUT a dependent chain inside.
//
// kernels_big.hip solves the 256 x 256 diagonal part of a super-step from the four resident 64 x 64 inverses and the six
// sub-diagonal blocks: seven dependent mat-vec stages with two barriers each, and the rows below the super-block are read
// only after that chain (16 loads in flight, four rounds).  On config 2 the top levels are a chain of ~40 such steps at
// 23.5 us each -- 10 us for the diagonal part, 13 us for the row part.  Here:
//   * the factorisation leaves, per 256-column block b of a big front, the full inverse of its diagonal block L_bb and the
//     transpose of that inverse (k_inv256: block column s of the inverse from the 64 x 64 inverses Y and the panel,
//     Inv(t, s) = -Y_t sum_{s <= p < t} L(t, p) Inv(p, s), FP64 MFMA, one workgroup per (front, block, s); it runs on a side
//     stream beside the next level's pivot chain);
//   * a forward super-step is then ONE triangular mat-vec y = Inv w (rows r and 255 - r folded into one list of 257 entries
//     of 64-row column steps dealt out to the eight waves: 80 coalesced loads per thread, all in flight from the start) and the rows below it in
//     workgroups of 64 rows whose 32 panel loads per thread are issued BEFORE the mat-vec -- nothing of the step waits for
//     memory twice;  the backward super-step is the same with the stored transpose.
// Sums meet in LDS in a fixed order: bitwise reproducible.  Storage: 2 x 512 KB per 256 pivot columns of a big front.
//
// Reference role: cholmod_l_solve (src/C/cholmod.c:483) for the big supernodes.
#include "device.hpp"

#include <algorithm>
#include <cstdint>
#include <type_traits>
#include <utility>

namespace kvx {

typedef double d4 __attribute__((ext_vector_type(4)));
namespace {
constexpr int NB = KVX_NB;
constexpr int IB = 256;                 // columns per super-step
constexpr int NT2 = 512;                // 8 waves with up to 256 registers each: a workgroup fills a CU like the 1024 x 128 steps of kernels_big.hip,
                                        // but every operand of the step fits in flight at once
constexpr int RB2 = 64;                 // rows below the super-block per workgroup (forward)
constexpr int CB2 = 32;                 // earlier pivot columns per workgroup (backward): four per wave
constexpr int TLD = 65;
}

// ------------------------------------------------------------------------------------------
// grid (s, b, front): block column s (64 columns) of the inverse of the diagonal block b (256 columns) of the front
__global__ __launch_bounds__(256) void k_inv256(DevSym ds, const int32_t *__restrict__ list, const int64_t *__restrict__ ioff,
                                                const double *__restrict__ Lx, const double *__restrict__ Linv,
                                                double *__restrict__ Inv)
{
    __shared__ double Tl[64 * TLD];
    const int fid = list[blockIdx.z];
    const FrontDesc fd = ds.fd[fid];
    const int k = fd.k, m = fd.m;
    const int jb0 = blockIdx.y * IB, s = blockIdx.x;
    if (jb0 >= k) return;
    const int nb = min(IB, k - jb0), nsub = (nb + NB - 1) / NB;
    if (s >= nsub) return;
    const double *P = Lx + fd.px;
    const double *Y = Linv + fd.linv + (int64_t)(jb0 / NB) * NB * NB;
    double *In = Inv + ioff[fid] + (int64_t)blockIdx.y * (2 * IB * IB), *InT = In + IB * IB;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, lk = l >> 4;
    // diagonal block: Inv(s, s) = Y_s (zeros above the diagonal, the later stages read the whole block)
    for (int e = tid; e < NB * NB; e += 256) {
        const int i = e & 63, j = e >> 6;
        const double v = kvx_ld0(Y, (int64_t)s * NB * NB + i + j * NB, i >= j && NB * s + i < nb);
        In[(NB * s + i) + (NB * s + j) * IB] = v;
        InT[(NB * s + j) + (NB * s + i) * IB] = v;
    }
    __syncthreads();
    for (int t = s + 1; t < nsub; t++) {
        // T = sum_p L(t, p) Inv(p, s): wave w owns rows 16 w .. 16 w + 15 of the 64 x 64 result
        d4 acc[4];
#pragma unroll
        for (int tj = 0; tj < 4; tj++) acc[tj] = (d4){0.0, 0.0, 0.0, 0.0};
        const int rrow = jb0 + NB * t + 16 * w + lr;                   // front row of this lane's A operand
        const bool rin = NB * t + 16 * w + lr < nb;
        for (int p = s; p < t; p++) {
#pragma unroll 4
            for (int k0 = 0; k0 < NB; k0 += 4) {
                const double av = kvx_ld0(P, rrow + (int64_t)(jb0 + NB * p + k0 + lk) * m, rin);
#pragma unroll
                for (int tj = 0; tj < 4; tj++) {
                    const double bv = In[(NB * p + k0 + lk) + (NB * s + 16 * tj + lr) * IB];
                    acc[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[tj], 0, 0, 0);
                }
            }
        }
        // acc[tj][q] = T[16 w + lk + 4 q][16 tj + lr]
#pragma unroll
        for (int tj = 0; tj < 4; tj++)
#pragma unroll
            for (int q = 0; q < 4; q++) Tl[(16 * w + lk + 4 * q) + TLD * (16 * tj + lr)] = acc[tj][q];
        __syncthreads();
        // Inv(t, s) = -Y_t T
        d4 r[4];
#pragma unroll
        for (int tj = 0; tj < 4; tj++) r[tj] = (d4){0.0, 0.0, 0.0, 0.0};
        const int yi = 16 * w + lr;
#pragma unroll 4
        for (int k0 = 0; k0 < NB; k0 += 4) {
            const int kk = k0 + lk;
            const double av = -kvx_ld0(Y, (int64_t)t * NB * NB + yi + kk * NB, kk <= yi && NB * t + yi < nb);
#pragma unroll
            for (int tj = 0; tj < 4; tj++) r[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Tl[kk + TLD * (16 * tj + lr)], r[tj], 0, 0, 0);
        }
#pragma unroll
        for (int tj = 0; tj < 4; tj++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = NB * t + 16 * w + lk + 4 * q, j = NB * s + 16 * tj + lr;
                In[i + j * IB] = r[tj][q];
                InT[j + i * IB] = r[tj][q];
            }
        __syncthreads();                            // Tl is reused; Inv(t, s) is read back by the next stage
    }
}

void launch_inv256(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_k, const int64_t *ioff,
                   const double *Lx, const double *Linv, double *Inv)
{
    if (count <= 0 || max_k <= 0) return;
    hipLaunchKernelGGL(k_inv256, dim3(4, (unsigned)((max_k + IB - 1) / IB), (unsigned)count), dim3(256), 0, st, ds, list, ioff, Lx, Linv, Inv);
}

// ------------------------------------------------------------------------------------------
// y = M w for the lower-triangular M (UPPER = false: y_r = sum_{c <= r} M[r + 256 c] w_c) or the upper-triangular
// M (UPPER = true: y_r = sum_{c >= r} M[r + 256 c] w_c) of order nb <= 256, both stored 256 x 256 column-major.  512 threads =
// 8 waves; lane = row inside a block of 64 rows, so every load is 512 contiguous bytes.  Row block R needs 64 (R + 1) columns
// (lower) / 64 (4 - R) columns (upper); blocks R and 3 - R are paired -- 320 column steps per pair -- and the four waves of a
// pair take 80 consecutive steps each.  Which (row block, column) a step is depends on the wave alone: the eight cases are
// expanded at compile time (TriSeg), so inside a wave every index is a constant, the operand of w is an LDS broadcast and there
// is no select in the accumulation.  tri_load issues the 80 loads (no dependence on w); tri_apply multiplies, and the partial
// sums of a row block meet in LDS in a fixed order (part: 4 row blocks x 4 waves x 64 doubles).  Ends with a barrier,
// ysh[0 .. 255] = y.
constexpr int EV = 80;
template <bool UPPER, int W>
struct TriSeg {
    static constexpr int P = W >> 2, J = W & 3;
    static constexpr int RA = P, RB = 3 - P;                                   // the pair of row blocks
    static constexpr int NA = UPPER ? 64 * (4 - RA) : 64 * (RA + 1);           // column steps of block RA (then those of RB)
    static constexpr int S0 = EV * J;
    static constexpr int N1 = S0 >= NA ? 0 : (NA - S0 < EV ? NA - S0 : EV);    // steps of this wave that belong to RA
    // step e -> row block and column
    static constexpr int rblk(int e) { return e < N1 ? RA : RB; }
    static constexpr int col(int e)
    {
        const int idx = e < N1 ? S0 + e : (S0 + e - NA);                       // position inside the block's column list
        return UPPER ? 64 * rblk(e) + idx : idx;
    }
};

template <bool UPPER, int W, int... Es>
__device__ __forceinline__ void tri_load_w(const double *__restrict__ M, int nb, int lane, double (&v)[EV], std::integer_sequence<int, Es...>)
{
    using T = TriSeg<UPPER, W>;
    auto one = [&](auto ec) {
        constexpr int e = decltype(ec)::value;
        constexpr int R = T::rblk(e), c = T::col(e);
        const int row = 64 * R + lane;
        // the triangle cuts only the diagonal 64 x 64 blocks; rows / columns past nb do not exist
        const bool ok = (UPPER ? (c < nb && ((c >> 6) != R || c >= row)) : (row < nb && ((c >> 6) != R || c <= row)));
        v[e] = kvx_ld0(M, row + c * IB, ok);
    };
    (one(std::integral_constant<int, Es>{}), ...);
}
template <bool UPPER, int W, int... Es>
__device__ __forceinline__ void tri_apply_w(const double (&v)[EV], const double *wsh, double &sa, double &sb, std::integer_sequence<int, Es...>)
{
    using T = TriSeg<UPPER, W>;
    auto one = [&](auto ec) {
        constexpr int e = decltype(ec)::value;
        constexpr int c = T::col(e);
        if (e < T::N1) sa = __builtin_fma(v[e], wsh[c], sa);
        else sb = __builtin_fma(v[e], wsh[c], sb);
    };
    (one(std::integral_constant<int, Es>{}), ...);
}
template <bool UPPER>
__device__ __forceinline__ void tri_load(const double *__restrict__ M, int nb, int tid, double (&v)[EV])
{
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto seq = std::make_integer_sequence<int, EV>();
    switch (w) {
    case 0: tri_load_w<UPPER, 0>(M, nb, lane, v, seq); break;
    case 1: tri_load_w<UPPER, 1>(M, nb, lane, v, seq); break;
    case 2: tri_load_w<UPPER, 2>(M, nb, lane, v, seq); break;
    case 3: tri_load_w<UPPER, 3>(M, nb, lane, v, seq); break;
    case 4: tri_load_w<UPPER, 4>(M, nb, lane, v, seq); break;
    case 5: tri_load_w<UPPER, 5>(M, nb, lane, v, seq); break;
    case 6: tri_load_w<UPPER, 6>(M, nb, lane, v, seq); break;
    default: tri_load_w<UPPER, 7>(M, nb, lane, v, seq); break;
    }
}
template <bool UPPER>
__device__ __forceinline__ void tri_apply(const double (&v)[EV], const double *wsh, double *part, double *ysh, int tid)
{
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    double sa = 0.0, sb = 0.0;
    auto seq = std::make_integer_sequence<int, EV>();
    switch (w) {
    case 0: tri_apply_w<UPPER, 0>(v, wsh, sa, sb, seq); break;
    case 1: tri_apply_w<UPPER, 1>(v, wsh, sa, sb, seq); break;
    case 2: tri_apply_w<UPPER, 2>(v, wsh, sa, sb, seq); break;
    case 3: tri_apply_w<UPPER, 3>(v, wsh, sa, sb, seq); break;
    case 4: tri_apply_w<UPPER, 4>(v, wsh, sa, sb, seq); break;
    case 5: tri_apply_w<UPPER, 5>(v, wsh, sa, sb, seq); break;
    case 6: tri_apply_w<UPPER, 6>(v, wsh, sa, sb, seq); break;
    default: tri_apply_w<UPPER, 7>(v, wsh, sa, sb, seq); break;
    }
    // part[(row block) * 256 + (wave of the pair) * 64 + lane]: block RA = P of the pair from sa, block RB = 3 - P from sb
    const int P = w >> 2, j = w & 3;
    part[P * 256 + j * 64 + lane] = sa;
    part[(3 - P) * 256 + j * 64 + lane] = sb;
    __syncthreads();
    if (tid < IB) {
        const double *pp = part + (tid >> 6) * 256 + (tid & 63);
        ysh[tid] = (pp[0] + pp[64]) + (pp[128] + pp[192]);
    }
    __syncthreads();
}

// Forward super-step over the pivot columns [jb0, jb0 + nb).  Workgroup x: the rows jb0 + nb + 64 x .. + 63 below the super-block
// (x = 0 also stores y).  FIRST assembles the right-hand side of the front first (pivot rows from X0, children's update vectors
// pulled in sequence), as k_fwd_big_step<true> does.
template <bool FIRST>
__global__ __launch_bounds__(NT2) void k_fwd_big_step2(DevSym ds, const int32_t *__restrict__ list, const int64_t *__restrict__ ioff,
                                                       int jb0, const double *__restrict__ Lx, const double *__restrict__ Inv,
                                                       double *__restrict__ X, const double *__restrict__ X0, int64_t ldx,
                                                       double *__restrict__ WK, int64_t ldw, const double *__restrict__ Wc,
                                                       double *__restrict__ Wo, int64_t wstride)
{
    unsigned bx, by, rh;
    kvx_part_front_rhs(bx, by, rh);
    __shared__ double part[1024];
    __shared__ double wsh[IB];
    __shared__ double ysh[IB];
    __shared__ double own[FIRST ? RB2 : 1];
    const int fid = list[by];
    const FrontDesc fd = ds.fd[fid];
    const int k = fd.k, m = fd.m, f = fd.first, tid = threadIdx.x;
    if (jb0 >= k) return;
    const int nb = min(IB, k - jb0);
    const int rbase = jb0 + nb + bx * RB2;
    if (bx > 0 && rbase >= m) return;
    const double *P = Lx + fd.px;
    const double *M = Inv + ioff[fid] + (int64_t)(jb0 / IB) * (2 * IB * IB);
    double *x = X + (int64_t)rh * ldx + f;
    double *wk = WK + (int64_t)rh * ldw + f;
    double *wo = Wo + (int64_t)rh * wstride + fd.wx;
    // everything that does not depend on the running vector goes out first: the inverse and this workgroup's panel rows
    double v[EV], u[32];
    tri_load<false>(M, nb, tid, v);
    const int rr = tid & 63, g = tid >> 6;                   // row of the block, group of 32 columns
    const int r = rbase + rr;
#pragma unroll
    for (int j = 0; j < 32; j++) u[j] = kvx_ld0(P, r + (int64_t)(jb0 + 32 * g + j) * m, r < m && 32 * g + j < nb);
    if (tid < IB) {
        const double *x0 = X0 + (int64_t)rh * ldx + f;      // rhs as it was before the sweep (x gets y meanwhile)
        wsh[tid] = kvx_ld0(FIRST ? x0 : wk, jb0 + tid, tid < nb);
        if (FIRST && tid < RB2) own[tid] = kvx_ld0(x0, rbase + tid, rbase + tid < k);
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
            for (int i = tid; i < cd.uc; i += NT2) {
                const int t = rl[i];
                const double val = src[i];
                if (t < nb) wsh[t] += val;
                else if (t >= rbase && t < rbase + RB2) own[t - rbase] += val;
            }
            __syncthreads();
            cd = nx;
        }
    }
    tri_apply<false>(v, wsh, part, ysh, tid);
    if (bx == 0 && tid < nb) x[jb0 + tid] = ysh[tid];
    if (rbase >= m) return;                                  // (x = 0 of a front without rows below the super-block)
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 32; j++) acc = __builtin_fma(u[j], ysh[32 * g + j], acc);
    part[g * 64 + rr] = acc;
    __syncthreads();
    if (tid < RB2 && r < m) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 8; q++) t += part[q * 64 + tid];
        if (FIRST) {
            if (r < k) wk[r] = own[tid] - t;
            else wo[r - k] = own[tid] - t;
        } else {
            if (r < k) wk[r] -= t;
            else wo[r - k] -= t;
        }
    }
}

// Backward super-step over the pivot columns [jb0, jb0 + nb), jb0 = 256 sidx:  x_b = Inv_bb' t_b (the stored transpose, an upper
// triangular mat-vec), then workgroup x owns the 32 earlier pivot columns 32 x .. of the front:  t_c -= L(b, c)' x_b, lane = row of
// the super-block (the summation index, contiguous in memory), wave = 4 columns, their 16 panel loads issued before the mat-vec.
__global__ __launch_bounds__(NT2) void k_bwd_big_step2(DevSym ds, const int32_t *__restrict__ list, const int64_t *__restrict__ ioff,
                                                       int sidx, const double *__restrict__ Lx, const double *__restrict__ Inv,
                                                       double *__restrict__ X, int64_t ldx, double *__restrict__ WK, int64_t ldw)
{
    unsigned bx, by, rh;
    kvx_part_front_rhs(bx, by, rh);
    __shared__ double part[1024];
    __shared__ double tsh[IB];
    __shared__ double xsh[IB];
    const int fid = list[by];
    const FrontDesc fd = ds.fd[fid];
    const int k = fd.k, m = fd.m, f = fd.first, tid = threadIdx.x;
    const int jb0 = sidx * IB;
    if (jb0 >= k) return;
    const int nb = min(IB, k - jb0);
    if (bx > 0 && (int)bx * CB2 >= jb0) return;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double *P = Lx + fd.px;
    const double *MT = Inv + ioff[fid] + (int64_t)sidx * (2 * IB * IB) + IB * IB;
    double *x = X + (int64_t)rh * ldx + f;
    double *wk = WK + (int64_t)rh * ldw + f;
    double v[EV], pv[4][4];
    tri_load<true>(MT, nb, tid, v);
    const int c0 = bx * CB2 + 4 * w;                         // this wave's four earlier columns
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            pv[c][j] = kvx_ld0(P, (int64_t)(jb0 + lane + j * NB) + (int64_t)(c0 + c) * m, lane + j * NB < nb && c0 + c < jb0);
    if (tid < IB) tsh[tid] = kvx_ld0(wk, jb0 + tid, tid < nb);
    __syncthreads();
    tri_apply<true>(v, tsh, part, xsh, tid);
    if (bx == 0 && tid < nb) x[jb0 + tid] = xsh[tid];
    if (c0 < jb0) {
        // four column sums over the 64 lanes (halving exchange, then a butterfly: fixed order)
        const int o = 2 * (lane & 1) + ((lane >> 1) & 1);    // the column whose total this lane holds
        const bool b0 = lane & 1, b1 = lane & 2;
        double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int j = 0; j < 4; j++) a[c] = __builtin_fma(pv[c][j], xsh[lane + j * NB], a[c]);
        const double s0 = b0 ? a[0] : a[2], s1 = b0 ? a[1] : a[3];
        const double k0 = (b0 ? a[2] : a[0]) + __shfl_xor(s0, 1);
        const double k1 = (b0 ? a[3] : a[1]) + __shfl_xor(s1, 1);
        double rsum = (b1 ? k1 : k0) + __shfl_xor(b1 ? k0 : k1, 2);
        rsum += __shfl_xor(rsum, 4);
        rsum += __shfl_xor(rsum, 8);
        rsum += __shfl_xor(rsum, 16);
        rsum += __shfl_xor(rsum, 32);
        if (lane < 4 && c0 + o < jb0) wk[c0 + o] -= rsum;
    }
}

void launch_fwd_big2(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int max_k, const int64_t *ioff,
                     const double *Lx, const double *Inv, double *X, const double *X0, int64_t ldx, int nrhs,
                     double *WK, int64_t ldw, const double *Wchild, double *Wout, int64_t wstride)
{
    if (count <= 0 || nrhs <= 0) return;
    for (int jb = 0; jb < max_k; jb += IB) {
        const int rows = max_m - jb - 1;
        dim3 grid((unsigned)std::max(1, (rows + RB2 - 1) / RB2), (unsigned)count, (unsigned)nrhs);
        if (jb == 0)
            hipLaunchKernelGGL(k_fwd_big_step2<true>, grid, dim3(NT2), 0, st, ds, list, ioff, jb, Lx, Inv, X, X0, ldx, WK, ldw, Wchild, Wout, wstride);
        else
            hipLaunchKernelGGL(k_fwd_big_step2<false>, grid, dim3(NT2), 0, st, ds, list, ioff, jb, Lx, Inv, X, X0, ldx, WK, ldw, Wchild, Wout, wstride);
    }
}

void launch_bwd_big_steps2(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_k, const int64_t *ioff,
                           const double *Lx, const double *Inv, double *X, int64_t ldx, int nrhs, double *WK, int64_t ldw)
{
    if (count <= 0 || nrhs <= 0) return;
    for (int b = (max_k + IB - 1) / IB - 1; b >= 0; b--) {
        const unsigned gx = (unsigned)std::max(1, b * IB / CB2);
        hipLaunchKernelGGL(k_bwd_big_step2, dim3(gx, (unsigned)count, (unsigned)nrhs), dim3(NT2), 0, st, ds, list, ioff, b, Lx, Inv,
                           X, ldx, WK, ldw);
    }
}

}  // namespace kvx
