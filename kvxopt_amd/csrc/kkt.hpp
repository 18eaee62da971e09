// Launchers of kkt.hip (NT scaling, normal-equations assembly, sparse mat-vec).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace kvx {
// up to 32 reductions in two launches: kind 0 = dot(x, y), 1 = max(-x)
struct MultiRed {
    int count;
    int kind[32];
    int64_t n[32];
    const double *x[32];
    const double *y[32];
};
// one right-hand side of a KKT solve around the triangular solves (k_kkt_pre / k_kkt_post)
struct KktSide {
    const double *xin; double xs; const double *zin;
    double *xout; double xos; double *zout; double zos;
};
struct KktSides { KktSide s[2]; };
// operands of the second half of f6_no_ir (k_lp_half_a / k_lp_half_b)
struct LpHalf {
    int64_t ml, n, p;
    const double *c, *dx, *b, *dy, *th, *dz, *z1, *x1, *y1, *lm;
    double *dxw, *dyw, *dsw, *dzw, *ws3;
};
void launch_reduce_multi_stage1(hipStream_t st, const MultiRed &mr, double *part);   // RED_BLOCKS partial results per reduction
int reduce_blocks();
void launch_kkt_pre(hipStream_t st, int64_t n, const int64_t *Gp, const int64_t *Gi, const double *Gx, const double *di, int nrhs,
                    const KktSides &r, double *x2, int64_t ld, int64_t max_col = 0);
void launch_kkt_post(hipStream_t st, int64_t ml, int64_t n, const int64_t *tGp, const int64_t *tGi, const double *tGx, const double *di,
                     int nrhs, const KktSides &r, const double *x2, int64_t ld, int64_t max_row = 0);
void launch_lp_residuals(hipStream_t st, int64_t ml, int64_t n, const int64_t *Gp, const int64_t *Gi, const double *Gx, const int64_t *Tp,
                         const int64_t *Ti, const double *Tx, const double *x, const double *z, const double *s, const double *c,
                         const double *h, double tau, double *hrx, double *rx, double *hrz, double *rz, int64_t max_col = 0, int64_t max_row = 0);
void launch_lp_second_half(hipStream_t st, const LpHalf &a, double dgi, double dtau0, double z1z1, double *part, double *part2);
void launch_lp_update_x(hipStream_t st, int64_t ml, int64_t n, double step, double *ds, double *dz, double *d, double *di, double *lm,
                        double *s, double *z, const double *dx, double *x);
void launch_dense_gemv(hipStream_t st, int64_t m, int64_t n, int64_t nrhs, double alpha, const double *A, int64_t lda, const double *x,
                       int64_t ldx, double beta, double *y, int64_t ldy);
void launch_lincomb(hipStream_t st, int64_t n, double a, const double *x, double b, const double *y, double *z);
void launch_reduce_multi(hipStream_t st, const MultiRed &mr, double *part, double *out);
void launch_compute_scaling(hipStream_t st, int64_t n, const double *s, const double *z, double *d, double *di, double *lm);
void launch_update_scaling(hipStream_t st, int64_t n, double *s, double *z, double *d, double *di, double *lm);
void launch_lp_newton_rhs(hipStream_t st, int64_t n, const double *lsq, const double *ws3, double shift, double scale,
                          const double *rz, const double *lm, const double *d, double *ds, double *dz);
void launch_lp_dtau(hipStream_t st, const double *r, double dgi, double dtau0, double z1z1_host, int use_host, double *out);
void launch_axpy_devalpha(hipStream_t st, int64_t n, const double *alpha, const double *x, double *y);
void launch_lp_step_post_devalpha(hipStream_t st, int64_t n, const double *dtau, const double *z1, const double *lm, double *ds, double *dz,
                                  double *ws3);
void launch_lp_step_post(hipStream_t st, int64_t n, double dtau, const double *z1, const double *lm, double *ds, double *dz, double *ws3);
void launch_lp_update(hipStream_t st, int64_t n, double step, double *ds, double *dz, double *d, double *di, double *lm, double *s, double *z);
void launch_scale(hipStream_t st, int64_t n, int64_t ncols, int64_t ldx, double *x, const double *w);
void launch_div(hipStream_t st, int64_t n, double *x, const double *y);
void launch_mul(hipStream_t st, int64_t n, double *x, const double *y);
void launch_sqr(hipStream_t st, int64_t n, double *x, const double *y);
void launch_axpy(hipStream_t st, int64_t n, double alpha, const double *x, double *y);
void launch_vscal(hipStream_t st, int64_t n, double alpha, double *x);
void launch_addc(hipStream_t st, int64_t n, double c, double *x);
void launch_fill(hipStream_t st, int64_t n, double c, double *x);
void launch_xmy(hipStream_t st, int64_t n, double a, const double *x, const double *y, double b, double *z);
void launch_dot(hipStream_t st, int64_t n, const double *x, const double *y, double *part, double *out);
void launch_maxneg(hipStream_t st, int64_t n, const double *x, double *part, double *out);
int reduce_scratch_doubles();
void launch_atda(hipStream_t st, int64_t snz, int64_t gnz, const int64_t *pp, const int32_t *pa, const int32_t *pb,
                 const int32_t *gi, const double *gx, const double *w, double *wg, double *sx, bool w_is_di = false);
void launch_add_at(hipStream_t st, int64_t pnz, const int64_t *slot, const double *px, double *sx);
void launch_spmm_t(hipStream_t st, int64_t n, int64_t ncols, const int64_t *Ap, const int64_t *Ai, const double *Ax, const double *X,
                   int64_t ldx, double *Y, int64_t ldy);
void launch_dense_from_ccs(hipStream_t st, int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax, double *D, int64_t ld);
void launch_pack_lower(hipStream_t st, int64_t p, const double *K, int64_t ld, double *out);
void launch_spmv(hipStream_t st, int trans, int64_t m, int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax,
                 double alpha, const double *x, double beta, double *y);
}  // namespace kvx
