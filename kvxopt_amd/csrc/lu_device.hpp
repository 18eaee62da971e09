// Device-side records and launchers of the sparse LU path (lu_kernels.hip); host plan in lu_symbolic.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>

namespace kvx {

void set_last_error(const std::string &s);     // api.cpp

struct LuFrontD {
    int32_t k, m, p0, nchild;
    int64_t px;          // offset of the m x k panels in Lx and in Ux (U stored transposed: U'(:, 0:k))
    int64_t rowptr;      // into rowidx / rel
    int64_t childptr;    // into children
    int64_t aptr;        // into a_src / a_dst
    int64_t upd_off;     // update matrix in the arena (for a big front: inside its own m x m region)
    int64_t wx;          // update vector of the solves
    int32_t upd_ld, acnt;
};

struct LuDev {           // all device pointers
    const LuFrontD *fr;
    const int32_t *rowidx, *rel, *children;
    const int64_t *a_src;
    const int32_t *a_dst;
    const int32_t *ai32;     // row index of every caller entry
    const double *rinv;      // 1 / row scale, by ORIGINAL row
    double *Lx, *Ux, *arena;
    int32_t *ipiv;           // [n] swap partner (local index) chosen at each pivot step
    int32_t *lperm;          // [n] local index of the front row that ended in each pivot slot
    int32_t *fail;           // [nfront] 0 = ok, else 1 + first pivot step without an acceptable pivot
    int64_t arena_size;      // doubles in `arena` (k_lu_front_wp keeps 32-bit offsets into it)
};

// Factor the fronts list[0..cnt) (one workgroup each).  lds_m > 0: fronts of order <= lds_m held in LDS; 0: in HBM.
void launch_lu_fronts(const LuDev &d, const int32_t *list, int cnt, int lds_m, int max_k, const double *Ax,
                      double tol, double stol, int reuse, hipStream_t st);
// All fronts of a level that do not fit in LDS (blocked: lu_kernels.hip k_lub_*).
void launch_lu_big_level(const LuDev &d, const int32_t *list, int cnt, int max_m, int max_k, const double *Ax, double tol,
                         double stol, int reuse, hipStream_t st, const uint8_t *swap_steps = nullptr);
// pivots per block while `rows` rows of the level's tallest front remain (the register budget of the panel's workgroup)
inline int lu_big_block_width(int rows) { return rows <= 1024 ? 32 : (rows <= 2048 ? 16 : 8); }
// Triangular sweeps over one level.  unit = 1: the L panels (unit diagonal, in-front row permutation);
// unit = 0: the U' panels.  W: update vectors, wsize doubles per right-hand side.
void launch_lu_fwd(const LuDev &d, const int32_t *list, int cnt, int max_m, int max_k, int unit, double *X, int64_t ldx,
                   int nrhs, double *W, int64_t wsize, hipStream_t st);
void launch_lu_bwd(const LuDev &d, const int32_t *list, int cnt, int max_m, int max_k, int unit, double *X, int64_t ldx,
                   int nrhs, hipStream_t st);
// The same sweeps for fronts of order > KVX_LU_SOLVE_BIG_M: many workgroups, one launch per 32 pivots.
void launch_lu_fwd_big(const LuDev &d, const int32_t *list, int cnt, int max_m, int max_k, int unit, double *X, int64_t ldx, int nrhs,
                       double *W, int64_t wsize, hipStream_t st);
void launch_lu_bwd_big(const LuDev &d, const int32_t *list, int cnt, int max_m, int max_k, int unit, double *X, int64_t ldx, int nrhs,
                       double *W, int64_t wsize, hipStream_t st);
// Off-diagonal blocks of the block triangular form: values (scaled entries of A) and the update X[pos] -= F(pos, :) X.
void launch_lu_fvals(int64_t nf, const int64_t *src, const double *Ax, const double *rinv, const int32_t *ai32, double *out, hipStream_t st);
void launch_lu_fterm(int64_t cnt, int nrhs, const int32_t *poslist, const int64_t *ptr, const int32_t *idx, const double *val, double *X,
                     int64_t ldx, hipStream_t st);
void launch_lu_rowmax(int64_t nnz, const int32_t *ai32, const double *Ax, double *rmax, hipStream_t st);   // rmax zeroed by the caller
void launch_lu_rinv(int64_t n, const double *rmax, double *rinv, hipStream_t st);
void launch_lu_zero(int64_t n, double *x, hipStream_t st);          // x := 0 (a kernel: the numeric pass is captured into a launch graph)
// X[p] = B[idx[p]] * (scale ? scale[idx[p]] : 1)   /   B[idx[p]] = X[p] * (scale ? scale[idx[p]] : 1)
void launch_lu_gather(int64_t n, int nrhs, const int64_t *idx, const double *scale, const double *B, int64_t ldb, double *X,
                      int64_t ldx, hipStream_t st);
void launch_lu_scatter(int64_t n, int nrhs, const int64_t *idx, const double *scale, const double *X, int64_t ldx, double *B,
                       int64_t ldb, hipStream_t st);
// d[p0 + t] = U(p0+t, p0+t) for every front
void launch_lu_udiag(const LuDev &d, int nfront, double *out, hipStream_t st);

}  // namespace kvx
