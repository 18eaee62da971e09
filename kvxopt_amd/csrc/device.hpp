// Device-side view of the symbolic analysis + launch helpers shared by the HIP sources.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <utility>

namespace kvx {

// Structure-of-arrays view of the fronts (all device pointers).  See symbolic.hpp.
struct DevSym {
    const int32_t *k;         // pivot columns per front
    const int32_t *m;         // front order
    const int32_t *first;     // first (permuted) column of the front
    const int64_t *px;        // panel offset in Lx
    const int64_t *rowptr;    // offset into rowidx / rel
    const int32_t *rowidx;    // global row index of each front row
    const int32_t *rel;       // position of each update row in the parent front
    const int64_t *ux;        // update-matrix offset in its parity buffer
    const int64_t *wx;        // solve update-vector offset in its parity buffer
    const int64_t *childptr;
    const int32_t *children;
    const int64_t *linv;      // big fronts: offset of the reciprocal pivots in Dinv (-1 otherwise)
    // Array-of-structures twins of the above: one 64-byte record per front and one 32-byte record
    // per (parent, child) edge, so that a workgroup reaches its operands after two dependent
    // loads instead of five (the per-front kernels are latency-bound on small levels).
    const struct FrontDesc *fd;
    const struct ChildDesc *cd;
    const int32_t *tiles;     // per child of a big front: first child column landing in each 16-column parent tile
    // Pivot rule (cholmod.options['dbound'], cholmod.c:116-117): a pivot d <= piv_floor is replaced by piv_repl
    // (piv_repl > 0) instead of ending the factorisation.  0 / 0 = the plain rule: d <= 0 (or NaN) fails.
    double piv_floor, piv_repl;
};

struct FrontDesc {            // 64 bytes
    int32_t k, m, first, nchild;
    int64_t px, rowptr, ux, wx, childptr, linv;
};
struct ChildDesc {            // 40 bytes, indexed like DevSym::children
    int32_t uc, kc;           // update rows / pivot columns of the child
    int64_t rel;              // index into DevSym::rel of the child's first update row
    int64_t ux, wx;           // child's update matrix / update vector offsets (previous level's parity buffer)
    int64_t tile;             // offset into DevSym::tiles (children of big fronts only, else -1)
};

// A leaf subtree handled by ONE wavefront in the triangular solves (kernels_wave.hip, k_fwd_subtree / k_bwd_subtree):
// fronts lo..hi (a contiguous range of the postordered numbering, hi = root), whose pivot columns are the contiguous
// slice [col0, col0 + ncols) of the permuted vector.
struct SubDesc {
    int32_t lo, hi, col0, ncols;
};
constexpr int KVX_SUB_MAXF = 48;      // hard limit of fronts per subtree (default limit 12: KVX_SUB_MAXF)
constexpr int KVX_SUB_MAXCOLS = 256;  // pivot columns per subtree (LDS slice of x)
constexpr int KVX_SUB_STACK = 512;    // doubles of update vectors alive at once (LDS stack)

#ifdef __HIPCC__
// Predicated load WITHOUT a branch: hipcc turns `c ? p[i] : 0.0` into an exec-masked branch with
// an exposed s_waitcnt per load (64 such loads serialise into 64 L2 round trips, ~20 us); an
// always-valid address plus a select keeps all loads of an unrolled batch in flight.
__device__ __forceinline__ double kvx_ld0(const double *__restrict__ p, int64_t idx, bool ok)
{
    const double v = p[ok ? idx : 0];
    return ok ? v : 0.0;
}
__device__ __forceinline__ double kvx_readlane(double v, int lane)
{
    const long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// sqrt(d) and 1/sqrt(d) without the IEEE division / square-root expansions (each a long chain
// of dependent FP64 ops): hardware v_rsq_f64 seed (~2^-26) + two Newton steps, then one
// correction of the root.  Results are within 1-2 ulp; d must be > 0 and normal.
__device__ __forceinline__ void kvx_sqrt_rsqrt(double d, double &root, double &inv)
{
    double r = __builtin_amdgcn_rsq(d);
    const double hd = 0.5 * d;
    r = r * __builtin_fma(-hd * r, r, 1.5);
    r = r * __builtin_fma(-hd * r, r, 1.5);
    double x = d * r;
    x = __builtin_fma(0.5 * r, __builtin_fma(-x, x, d), x);
    root = x;
    inv = r;
}

// Column step J of a register-resident right-looking Cholesky sweep: lane r holds row r, the
// columns live in a[0..KMAX).  The pivot travels by one v_readlane; the multipliers of column J
// are broadcast through a 64-double LDS buffer (measured on MI355X: a readlane pair + fma costs
// ~40 cycles per use, an LDS broadcast read + fma ~10), and the loads of the buffer overlap the
// rsqrt chain because the UNSCALED column is published and scaled by 1/d on the consumer side.
// cb2: LDS double[2][64] (double-buffered by J parity); dinv: optional output of 1/l_jj.
// floor: pivots d <= floor (or NaN) are replaced by sub; flag_all: 1 = such a pivot also ends the factorisation (the plain
// rule; sub = 1 only keeps the arithmetic finite), 0 = it does not (dbound mode: no failure is reported, NaN included).
struct PivRule { double floor, sub; int flag_all; };
__device__ __forceinline__ PivRule make_piv_rule(const DevSym &ds)
{
    return PivRule{ds.piv_floor, ds.piv_repl > 0.0 ? ds.piv_repl : 1.0, ds.piv_repl > 0.0 ? 0 : 1};
}
template <int KMAX, int J>
__device__ __forceinline__ void kvx_col_step(double (&a)[KMAX], int k, int r, int *status, int col0,
                                             double *dinv, double *cb2, const PivRule pr)
{
    if (J < k) {                                   // wave-uniform
        double *cb = cb2 + (J & 1) * 64;
        const double aj = a[J];
        cb[r] = aj;
        double d = kvx_readlane(aj, J);
        if (!(d > pr.floor)) {
            if (pr.flag_all) {
                if (r == 0) atomicMin(status, col0 + J);
            }
            d = pr.sub;
        }
        double ljj, inv;
        kvx_sqrt_rsqrt(d, ljj, inv);
        if (dinv != nullptr && r == 0) dinv[J] = inv;
        const double w = (r > J) ? aj * (inv * inv) : 0.0;
#pragma unroll
        for (int c = J + 1; c < KMAX; c++) a[c] = __builtin_fma(-w, cb[c], a[c]);
        a[J] = (r == J) ? ljj : (r > J ? aj * inv : 0.0);
    }
}
template <int KMAX, int... Js>
__device__ __forceinline__ void kvx_col_steps(double (&a)[KMAX], int k, int r, int *status, int col0, double *dinv,
                                              double *cb2, const PivRule pr, std::integer_sequence<int, Js...>)
{
    (kvx_col_step<KMAX, Js>(a, k, r, status, col0, dinv, cb2, pr), ...);
}
#endif

// Workgroup numbering of a launch over fronts of very different sizes: up to KVX_MAXCLS size classes, class c = the fronts
// list[first[c] .. first[c + 1]), each given the tiles of a T[c]-tile-row update region (TC[c] > 0: T[c] x TC[c] tiles, a
// column-limited region; 0: the lower triangle), workgroup ids wg[c] .. wg[c + 1).  Passed by value as a kernel argument.
constexpr int KVX_MAXCLS = 16;
struct TileClasses {
    int32_t ncls;
    int32_t first[KVX_MAXCLS + 1];
    int32_t T[KVX_MAXCLS];
    int32_t TC[KVX_MAXCLS];
    uint32_t wg[KVX_MAXCLS + 1];
};

constexpr int KVX_NB = 64;           // panel width of the blocked big-front factorisation
constexpr int KVX_SMALL_MAX = 128;   // fronts up to this order are factored inside LDS
constexpr int KVX_TILE = 64;         // trailing-update tile
constexpr int KVX_ASM_TC = 16;       // parent columns owned by one extend-add workgroup (big fronts)

// ---- launchers (kernels.hip) ---------------------------------------------------------------

// Right-hand sides are a grid dimension of the solve kernels.  Workgroups are dispatched in linear order (x fastest), so with the
// natural (front, rhs) = (blockIdx.x, blockIdx.y) numbering the nrhs workgroups that read one front's panel run far apart and
// every one of them fetches it from HBM again (n = 1e6, 64 right-hand sides: 1.1 TB/s of panel re-reads).  These helpers renumber
// the same grid rhs-fastest: the workgroups of one front are dispatched back to back (round-robin over the XCDs), the panel is
// fetched once per XCD and then hits in its L2.  With one right-hand side the numbering is the natural one.
__device__ __forceinline__ void kvx_front_rhs(unsigned &fi, unsigned &rh)
{
    if (gridDim.y == 1) { fi = blockIdx.x; rh = 0; return; }
    const unsigned long long L = blockIdx.x + (unsigned long long)gridDim.x * blockIdx.y;
    fi = __builtin_amdgcn_readfirstlane((unsigned)(L / gridDim.y));
    rh = __builtin_amdgcn_readfirstlane((unsigned)(L % gridDim.y));
}
// 3-D grids (x = part of a front, y = front, z = rhs): rhs fastest, then x, then y
__device__ __forceinline__ void kvx_part_front_rhs(unsigned &bx, unsigned &by, unsigned &rh)
{
    if (gridDim.z == 1) { bx = blockIdx.x; by = blockIdx.y; rh = 0; return; }
    const unsigned long long L = blockIdx.x + (unsigned long long)gridDim.x * (blockIdx.y + (unsigned long long)gridDim.y * blockIdx.z);
    const unsigned long long rest = L / gridDim.z;
    rh = __builtin_amdgcn_readfirstlane((unsigned)(L % gridDim.z));
    bx = __builtin_amdgcn_readfirstlane((unsigned)(rest % gridDim.x));
    by = __builtin_amdgcn_readfirstlane((unsigned)(rest / gridDim.x));
}

void launch_scatter_a(hipStream_t st, const double *Ax, const int64_t *amap, int64_t nnz, double *Lx);
// zero L + scatter A in one pass: the scatter map grouped by chunk of 2^init_factor_shift() doubles of L (cptr: nchunk + 1 offsets)
int init_factor_shift();
void launch_scatter_group_count(hipStream_t st, const int64_t *amap, int64_t nnz, int sh, int64_t *cnt);
void launch_scatter_group_place(hipStream_t st, const int64_t *amap, int64_t nnz, int sh, int64_t *cursor, int64_t *sdst, int32_t *ssrc);
void launch_init_factor(hipStream_t st, const double *Ax, const int32_t *src, const int64_t *dst, const int64_t *cptr, int64_t lsize,
                        double *Lx, int *status);
void launch_publish_status(hipStream_t st, const int *d_status, int *host_status_dev);   // *host := *d_status (pinned, device-visible)
void launch_copy_d(hipStream_t st, double *dst, const double *src, int64_t n);          // dst := src, 16-byte aligned blocks
void launch_clear_factor(hipStream_t st, double *Lx, int64_t n, int *status);   // Lx := 0, *status := 0x7f7f7f7f ("no failing column")
// LDS-front kernel (kernels_wave.hip): m <= mcap (96 or 128), k <= kmax (32 or 64)
void launch_front_small(hipStream_t st, int mcap, int kmax, const DevSym &ds, const int32_t *list, int count,
                        double *Lx, const double *Uchild, double *Uout, int *status);
// wave-per-front kernel (kernels_wave.hip): m <= mcap (32/48/64), k <= kmax (16/32)
void launch_front_wave(hipStream_t st, int mcap, int kmax, const DevSym &ds, const int32_t *list, int count,
                       double *Lx, const double *Uchild, double *Uout, int *status);
// leaf subtrees of the factorisation (kernels_wave.hip): one wavefront factors a whole subtree; U0 / U1: the parity buffers
void launch_factor_subtree(hipStream_t st, int mcap, const DevSym &ds, const SubDesc *subs, int nsub, const int32_t *depth,
                           double *Lx, double *U0, double *U1, int *status);
void launch_assemble_big(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m,
                         double *Lx, const double *Uchild, double *Uout);
// extend-add + the first diagonal block (factor and inverse) in one launch: for launches of few workgroups (every one reserves 49 KB of LDS)
void launch_assemble_big_potrf(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m,
                               double *Lx, const double *Uchild, double *Uout, double *Linv, int *status);
void launch_potrf_blk(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int jb,
                      double *Lx, double *Linv, int *status);
void launch_trsm_blk(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int jb,
                     double *Lx, const double *Linv);
// C -= X X' on the trailing tiles; the workgroup of tile (0, 0) also factors + inverts the NEXT diagonal
// block (jb + 64), so launch_potrf_blk is needed for the first panel of a front only
// col_lim: INT_MAX = the whole trailing matrix; KVX_COLS_PIVOT = the pivot columns only (the update matrix is brought up to
// date by launch_syrk_u afterwards, with all of the front's panels in few passes)
constexpr int KVX_COLS_PIVOT = 0x7ffffffe;
void launch_syrk_trailing(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int jb,
                          double *Lx, double *Uout, double *Linv, int *status, int col_lim = 0x7fffffff);
// LDS-staged trailing update over size classes of the fronts (kernels_big.hip): list = the fronts still in the chain at panel
// step kb, sorted by the order of their update region (largest first); hm / hk = their orders and pivot counts (host arrays)
void launch_syrk_step(hipStream_t st, const DevSym &ds, const int32_t *list, const int32_t *hm, const int32_t *hk, int count, int kb, int klen,
                      double *Lx, double *Uout, double *Linv, int *status, int col_lim);
// deferred ("far") update with the panel block [kb, kb + klen): C -= X X' on everything from column t0 on (later pivot columns
// and the update matrix; t0 >= k: the update matrix alone)
void launch_syrk_far(hipStream_t st, const DevSym &ds, const int32_t *list, const int32_t *hm, const int32_t *hk, int count, int kb, int klen,
                     int t0, double *Lx, double *Uout);

// two-level blocking (fronts that are flop-bound): per panel only the rest of the 256-column outer block [.., ob_end),
// then one rank-(<= ob_len) update of everything right of the outer block [ob, ob + ob_len)
void launch_syrk_inner(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int jb, int ob_end,
                       double *Lx, double *Uout, double *Linv, int *status);
void launch_syrk_outer(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int ob, int ob_len,
                       double *Lx, double *Uout, double *Linv, int *status);
// pair schedule: the panels jb and jb + 64 in ONE pass over everything right of them (64-tile kernel, K = 128; tile (0, 0) factors
// the diagonal block at jb + 128); launch_syrk_inner(jb, jb + 128) goes in front of the second panel's solve
void launch_syrk_pair(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int jb,
                      double *Lx, double *Uout, double *Linv, int *status, int col_lim = 0x7fffffff);

// sharded mode: rank-ob_len update with the panel [ob, ob + ob_len) of the columns in [c_from, c_to) that rank own_r owns
void launch_syrk_outer_dist(hipStream_t st, const DevSym &ds, const int32_t *list, int max_m, int ob, int ob_len,
                            int own_ob, int own_g, int own_r, int c_from, int c_to, double *Lx, double *Uout);

// solves: X is n x nrhs (ld = ldx) in PERMUTED order; W* are parity workspaces, each rhs
// column uses a slice of wstride doubles.
void launch_fwd_level(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m,
                      const double *Lx, double *X, int64_t ldx, int nrhs,
                      const double *Wchild, double *Wout, int64_t wstride);
void launch_bwd_level(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m,
                      const double *Lx, double *X, int64_t ldx, int nrhs);
// wave classes (m <= 64, k <= 32): one wavefront per front, no LDS image (kernels_wave.hip)
void launch_fwd_wave(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int kmax,
                     const double *Lx, double *X, int64_t ldx, int nrhs, const double *Wchild, double *Wout, int64_t wstride);
void launch_bwd_wave(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int mcap, int kmax,
                     const double *Lx, double *X, int64_t ldx, int nrhs);
// leaf subtrees: one wavefront walks a whole subtree (all of its fronts are wave-class), update vectors on an LDS stack
void launch_fwd_subtree(hipStream_t st, const DevSym &ds, const SubDesc *subs, int nsub, const int32_t *edges,
                        const double *Lx, double *X, int64_t ldx, int nrhs, double *W0, double *W1, int64_t wstride,
                        const int32_t *depth);
void launch_bwd_subtree(hipStream_t st, const DevSym &ds, const SubDesc *subs, int nsub, int nsub32, const double *Lx, double *X,
                        int64_t ldx, int nrhs);
void launch_bwd_subtree_group(hipStream_t st, int mcap, const DevSym &ds, const SubDesc *subs, int count, const double *Lx, double *X,
                              int64_t ldx);   // single rhs: one size group (32 / 48 / 64 rows) of the subtree table
// LDS classes (m <= 128, k <= 64): two wavefronts per front (kernels_wave.hip)
void launch_fwd_lds(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int kmax,
                    const double *Lx, double *X, int64_t ldx, int nrhs, const double *Wchild, double *Wout, int64_t wstride);
void launch_bwd_lds(hipStream_t st, const DevSym &ds, const int32_t *list, int count,
                    const double *Lx, double *X, int64_t ldx, int nrhs);
// big fronts (m > KVX_SMALL_MAX): multi-workgroup solves using the inverted diagonal blocks
void launch_fwd_big(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int max_k,
                    const double *Lx, const double *Linv, double *X, const double *X0, int64_t ldx, int nrhs,
                    double *WK, int64_t ldw, const double *Wchild, double *Wout, int64_t wstride, int level_count = 0);
void launch_bwd_big(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int max_k,
                    const double *Lx, const double *Linv, double *X, int64_t ldx, int nrhs, double *WK, int64_t ldw);
// out[k + r*ldo] = in[perm[k] + r*ldi]  (gather)   /   out[perm[k] + r*ldo] = in[k + r*ldi]  (scatter)
void launch_perm_gather(hipStream_t st, const int32_t *perm, int64_t n, int nrhs, const double *in, int64_t ldi,
                        double *out, int64_t ldo);
void launch_perm_scatter(hipStream_t st, const int32_t *perm, int64_t n, int nrhs, const double *in, int64_t ldi,
                         double *out, int64_t ldo);
// many right-hand sides, rhs-major blocks of 64 (kernels_wide.hip): XT[chunk][row][64]; W*: [chunk][wstride rows][64]
void launch_wide_gather(hipStream_t st, const int32_t *iperm, int64_t n, int nrhs, const double *B, int64_t ldB, double *XT,
                        const double *diag, int mode);   // diag != nullptr: rows scaled on the way (mode 0: times diag, 1: divided by it)
void launch_wide_scatter(hipStream_t st, const int32_t *iperm, int64_t n, int nrhs, const double *XT, double *B, int64_t ldB,
                         const double *diag, int mode);
void launch_wide_fwd_small(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int kmax, int nchunk, const double *Lx,
                           double *XT, int64_t n, const double *Wch, double *Wout, int64_t wstride, const int32_t *inv_ptr,
                           const int32_t *inv_src);
void launch_wide_bwd_small(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int kmax, int nchunk, const double *Lx,
                           double *XT, int64_t n);
void launch_wide_fwd_big(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int max_k, int nchunk,
                         const double *Lx, const double *Linv, double *XT, int64_t n, const double *Wch, double *Wout,
                         int64_t wstride, const int32_t *inv_ptr, const int32_t *inv_src);
void launch_wide_bwd_big(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_k, int nchunk, const double *Lx,
                         const double *Linv, double *XT, int64_t n);
void launch_zero_slots(hipStream_t st, const int64_t *slots, int count, int nrhs, double *W, int64_t wstride);
void launch_scatter_entries(hipStream_t st, const int64_t *pos, const double *val, int64_t count, double *X);
void launch_mask_rows(hipStream_t st, const uint8_t *keep, int64_t n, int nrhs, double *X, int64_t ldx);
void launch_extract_diag(hipStream_t st, const DevSym &ds, int64_t nsuper, const double *Lx, double *d);
void launch_diag_scale(hipStream_t st, int64_t n, int nrhs, const double *d, double *X, int64_t ldx, int mode);

}  // namespace kvx
