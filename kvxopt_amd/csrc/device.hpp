// Device-side view of the symbolic analysis + launch helpers shared by the HIP sources.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

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
    const int64_t *linv;      // big fronts: offset of the inverted 64x64 diagonal blocks (-1 otherwise)
    // Array-of-structures twins of the above: one 64-byte record per front and one 32-byte record
    // per (parent, child) edge, so that a workgroup reaches its operands after two dependent
    // loads instead of five (the per-front kernels are latency-bound on small levels).
    const struct FrontDesc *fd;
    const struct ChildDesc *cd;
};

struct FrontDesc {            // 64 bytes
    int32_t k, m, first, nchild;
    int64_t px, rowptr, ux, wx, childptr, linv;
};
struct ChildDesc {            // 32 bytes, indexed like DevSym::children
    int32_t uc, kc;           // update rows / pivot columns of the child
    int64_t rel;              // index into DevSym::rel of the child's first update row
    int64_t ux, wx;           // child's update matrix / update vector offsets (previous level's parity buffer)
};

#ifdef __HIPCC__
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
#endif

constexpr int KVX_NB = 64;           // panel width of the blocked big-front factorisation
constexpr int KVX_SMALL_MAX = 128;   // fronts up to this order are factored inside LDS
constexpr int KVX_TILE = 64;         // trailing-update tile

// ---- launchers (kernels.hip) ---------------------------------------------------------------
void launch_scatter_a(hipStream_t st, const double *Ax, const int64_t *amap, int64_t nnz, double *Lx);
// LDS-front kernel (kernels_wave.hip): m <= mcap (96 or 128), k <= kmax (32 or 64)
void launch_front_small(hipStream_t st, int mcap, int kmax, const DevSym &ds, const int32_t *list, int count,
                        double *Lx, const double *Uchild, double *Uout, int *status);
// wave-per-front kernel (kernels_wave.hip): m <= mcap (32/48/64), k <= kmax (16/32)
void launch_front_wave(hipStream_t st, int mcap, int kmax, const DevSym &ds, const int32_t *list, int count,
                       double *Lx, const double *Uchild, double *Uout, int *status);
void launch_assemble_big(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m,
                         double *Lx, const double *Uchild, double *Uout);
void launch_potrf_blk(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int jb,
                      double *Lx, double *Linv, int *status);
void launch_trsm_blk(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int jb,
                     double *Lx, const double *Linv);
void launch_syrk_trailing(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int jb,
                          double *Lx, double *Uout);

// solves: X is n x nrhs (ld = ldx) in PERMUTED order; W* are parity workspaces, each rhs
// column uses a slice of wstride doubles.
void launch_fwd_level(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m,
                      const double *Lx, double *X, int64_t ldx, int nrhs,
                      const double *Wchild, double *Wout, int64_t wstride);
void launch_bwd_level(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m,
                      const double *Lx, double *X, int64_t ldx, int nrhs);
// big fronts (m > KVX_SMALL_MAX): multi-workgroup solves using the inverted diagonal blocks
void launch_fwd_big(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int max_k,
                    const double *Lx, const double *Linv, double *X, int64_t ldx, int nrhs,
                    double *WK, int64_t ldw, const double *Wchild, double *Wout, int64_t wstride);
void launch_bwd_big(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m, int max_k,
                    const double *Lx, const double *Linv, double *X, int64_t ldx, int nrhs, double *WK, int64_t ldw);
// out[k + r*ldo] = in[perm[k] + r*ldi]  (gather)   /   out[perm[k] + r*ldo] = in[k + r*ldi]  (scatter)
void launch_perm_gather(hipStream_t st, const int32_t *perm, int64_t n, int nrhs, const double *in, int64_t ldi,
                        double *out, int64_t ldo);
void launch_perm_scatter(hipStream_t st, const int32_t *perm, int64_t n, int nrhs, const double *in, int64_t ldi,
                         double *out, int64_t ldo);
void launch_extract_diag(hipStream_t st, const DevSym &ds, int64_t nsuper, const double *Lx, double *d);

}  // namespace kvx
