/*
 * kvxhip.h -- C ABI of libkvxhip.so: MI355X (gfx950) implementation of the KKT
 * factor/solve hot path of sanurielf/kvxopt.
 *
 * Every entry point is what the reference's Python/C extension layer would bind for
 * this path; the reference interface each one replaces is cited as file:line relative
 * to the reference tree.  Plain pointers and sizes only; indices are int64 (the
 * reference's int_t = Py_ssize_t, src/C/kvxopt.h:46), values are IEEE double.
 *
 * Pointer kinds: *_host = host memory, *_dev = device (HBM) memory of the current
 * HIP device.  Unless a name ends in _dev, pointers are host pointers.
 *
 * Return value: 0 = KVX_OK, otherwise one of the KVX_E* codes below.  The thin host
 * layer maps codes to the exceptions the reference raises (see each function).
 */
#ifndef KVXHIP_H
#define KVXHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KVX_OK            0
#define KVX_EINVAL        1   /* bad argument            -> ValueError / TypeError          */
#define KVX_ENOMEM        2   /* host or device OOM     -> MemoryError                      */
#define KVX_ENOTPOSDEF    3   /* factor does not exist  -> ArithmeticError(minor)           */
#define KVX_ESYMBOLIC     4   /* numeric op on a symbolic-only factor -> ValueError         */
#define KVX_ESINGULAR     5   /* solve on a failed factor -> ArithmeticError("singular")    */
#define KVX_EDEVICE       6   /* HIP runtime error / no GPU -> RuntimeError (never a CPU fallback) */
#define KVX_EPERM         7   /* p is not a valid permutation -> ValueError                 */

typedef struct kvx_chol kvx_chol;          /* opaque factor: replaces the cholmod_factor capsule (cholmod.c:287-290) */
typedef struct kvx_atda kvx_atda;          /* opaque plan for S = G' D G on a fixed pattern (misc.py:1418-1462) */

/* Options: replaces cholmod.options / set_options() (cholmod.c:87-129). */
typedef struct {
    int32_t supernodal;     /* cholmod.options['supernodal'] (spsolvers.rst:731-736).  2 (default): P A P' = L L'.  0: P A P' = L D L'.
                             * 1: LL' if flops / nnz(L) >= 40 (CHOLMOD's rule), else LDL'.  One set of kernels computes Lc with
                             * P A P' = Lc Lc'; the LDL' factor is that result seen as L = Lc diag(Lc)^-1, D = diag(Lc)^2: solve
                             * sys = 2..6, getfactor (D on the diagonal, unit diagonal of L implicit) and diag (refused) follow it */
    int32_t ordering;       /* the library's own ordering (used when p == NULL, or compared with p when reserved[4] = 1):
                             * 0 = best of nested dissection (from order reserved[6] on) and, up to order reserved[5] (default 200 000), approximate minimum degree --
                             * the one with the least fill, CHOLMOD's strategy of trying several methods (cholmod.c:65-76);
                             * 1 = natural; 2 = nested dissection only; 3 = approximate minimum degree only (the role of AMD) */
    int32_t postorder;      /* 1 (default) elimination-tree postorder on top of the ordering (cholmod.c:113-115) */
    int32_t relax_small;    /* relaxed-amalgamation: always merge if merged width <= this (default 4)    */
    double  relax_z1;       /* zero-fraction bounds for widths <=16, <=48, any (defaults .8, .1, and for any width    */
    double  relax_z2;       /* relax_z3 < 0 = by the order of the matrix: .2 up to 150 000 columns, .075 beyond;      */
    double  relax_z3;       /* 0 = no relaxation of wide supernodes)                                                   */
    double  dbound;         /* cholmod.options['dbound'] (cholmod.c:116-117); 0 = off: a pivot d <= 0 fails.  > 0: CHOLMOD's rule,
                             * L_kk < dbound is replaced by dbound; with reserved[3] = 1 by 1e64 (the row drops out of the solves) */
    int32_t reserved[8];    /* [0] nd_leaf, [1] leaf_cols, [2] leaf_rows, [3] dbound mode, [4] 1 = a given p competes with the library's own
                             * ordering(s) and the least fill wins (cholmod.options['nmethods'] = 0 or 2); 0 = p is used as given
                             * (nmethods = 1), [5] largest order for the minimum-degree candidate of ordering 0, [6] smallest order for which
                             * ordering 0 computes the dissection beside the minimum degree (default 20 000; -1 = always)                */
} kvx_chol_opts;

void kvx_chol_default_opts(kvx_chol_opts *o);

/* Library / device probes (host layer uses these to fail loudly when there is no GPU). */
const char *kvx_version(void);
int  kvx_device_count(void);            /* number of visible HIP devices (0 on a CPU-only box) */
int  kvx_current_device(void);          /* the calling thread's current HIP device, -1 without one: device objects (factors, plans,
                                         * KKT objects) live on the device that was current when they were built; the host layer's
                                         * caches key on it                                    */
int  kvx_set_device(int dev);           /* make `dev` the calling thread's current HIP device (a rank process of the sharded mode
                                         * picks its GPU with this before anything else touches HIP) */
const char *kvx_last_error(void);       /* text of the last error on this thread               */
int  kvx_graph_instantiate_failures(void); /* launch graphs whose instantiation (on a thread of the library) failed since the process
                                         * started: those handles run their launches eagerly -- correct, slower */

/* ---- sparse Cholesky: replaces kvxopt.cholmod ------------------------------------------ */

/* symbolic(A, p, uplo)  -- cholmod.c:244-291 (pack :132-181, analyze_p :274).
 * A is n x n CCS (colptr[n+1], rowind[nnz], rows sorted within a column); only the `uplo`
 * ('L' or 'U') triangle is read.  perm: NULL, or n int64 with P*A*P' = L*L',
 * (PAP')(i,j) = A(perm[i],perm[j]).  Host-only work; no GPU needed. */
int kvx_chol_analyze(int64_t n, const int64_t *colptr, const int64_t *rowind, int uplo,
                     const int64_t *perm, const kvx_chol_opts *opts, kvx_chol **out);

/* numeric(A, F) -- cholmod.c:322-398.  values[nnz] are the entries of the matrix whose
 * pattern was analysed (same order).  On KVX_ENOTPOSDEF, *minor = failing column (index in
 * the permuted matrix, as CHOLMOD's L->minor, cholmod.c:376-379). */
int kvx_chol_factorize(kvx_chol *F, const double *values, int64_t *minor);
int kvx_chol_factorize_dev(kvx_chol *F, const double *values_dev, int64_t *minor);
/* Asynchronous variant: enqueue only; status is read later by kvx_chol_status(). */
int kvx_chol_factorize_async_dev(kvx_chol *F, const double *values_dev);
int kvx_chol_status(kvx_chol *F, int64_t *minor);      /* synchronises the factor's stream */

/* solve(F, B, sys, nrhs, ldB, offsetB) -- cholmod.c:429-499; sys 0..8 = A, LDL', LD, DL', L,
 * L', D, P, P' (:437-439) with D = I for an LL' factor (spsolvers.rst:640-668).  B (n x nrhs, leading dimension ldB >= max(1,n)) is
 * overwritten.  The caller applies offsetB to the pointer.  After kvx_chol_factorize_async_dev the solve
 * is queued behind the factorisation without a host round trip; a failed factorisation is then reported by
 * the solve (KVX_ESINGULAR, B undefined), as the reference's solve does on a failed factor (cholmod.c:456).
 * With many right-hand sides (48, or 8 when nrhs * n >= 6e6; KVX_WIDE_FROM) an LL' solve works on rhs-major blocks of 64 with GEMM-shaped panel products (FP64 MFMA):
 * its columns agree with single-rhs solves to rounding, not bit for bit -- CHOLMOD's BLAS-3 solves do not either. */
int kvx_chol_solve(kvx_chol *F, int sys, double *B, int64_t nrhs, int64_t ldB);
int kvx_chol_solve_dev(kvx_chol *F, int sys, double *B_dev, int64_t nrhs, int64_t ldB);
/* Enqueue only: no host synchronisation; work the caller submits to the null stream afterwards is ordered behind the
 * solve.  Errors of a factorisation still in flight are reported by the next synchronising call (kvx_chol_status). */
int kvx_chol_solve_async_dev(kvx_chol *F, int sys, double *B_dev, int64_t nrhs, int64_t ldB);
/* numeric(A, F) followed by solve(F, B) with sys = 0 (cholmod.c:322-398 then :429-499; what linsolve does after its analysis,
 * cholmod.c:618-753) as ONE enqueue on device buffers: the forward sweep follows the factorisation level by level on streams of
 * its own, so only the backward sweep and the root's forward step are left when the last front is factored.  Same kernels, same
 * order per front: the factor and X are bitwise those of kvx_chol_factorize_dev + kvx_chol_solve_dev.  Synchronises; returns
 * KVX_ENOTPOSDEF with *minor = failing column (B_dev then holds garbage).  The factor stays usable for further solves. */
int kvx_chol_factorize_solve_dev(kvx_chol *F, const double *values_dev, double *B_dev, int64_t nrhs, int64_t ldB, int64_t *minor);
/* ... with host buffers: the numeric + solve part of linsolve(A, B) (cholmod.c:663-753) in one call. */
int kvx_chol_factorize_solve(kvx_chol *F, const double *values, double *B, int64_t nrhs, int64_t ldB, int64_t *minor);
/* ... enqueue only: the caller's later null-stream work is ordered behind it; kvx_chol_status reports the factorisation afterwards
 * (the form misc.kkt_chol2's factor + first solves take inside the interior-point loop). */
int kvx_chol_factorize_solve_async_dev(kvx_chol *F, const double *values_dev, double *B_dev, int64_t nrhs, int64_t ldB);

/* spsolve(F, B, sys) -- cholmod.c:524-587.  B is n x ncol CCS; the result is returned as a
 * newly malloc'ed CCS triple the caller frees with kvx_free() (entries that are exactly zero are dropped, as
 * cholmod_spsolve does on a supernodal factor).  The forward systems (sys 2: L D x = b, 4: L x = b) sweep only the reach
 * of each block of 64 columns in the supernodal elimination tree -- the fronts that hold a nonzero row of the block and
 * their ancestors -- and copy back only the rows of those fronts; the other codes solve dense column blocks. */
int kvx_chol_spsolve(kvx_chol *F, int sys, int64_t ncol, const int64_t *Bp, const int64_t *Bi,
                     const double *Bx, int64_t **Xp, int64_t **Xi, double **Xx);

/* diag(F) -- cholmod.c:900-945: diagonal of L in permuted order, n doubles.  An LDL' factor is refused (KVX_ESYMBOLIC,
 * "F must be a nonsingular supernodal Cholesky factor", cholmod.c:919-922): its D comes from solve(sys = 6). */
int kvx_chol_diag(kvx_chol *F, double *d);

/* getfactor(F) -- cholmod.c:948-985: L as CCS (lower, sorted). Call with Lp=Li=Lx=NULL to
 * query *lnz first. */
int kvx_chol_get_factor(kvx_chol *F, int64_t *lnz, int64_t *Lp, int64_t *Li, double *Lx);

/* Introspection used by bench/tests: the measure of SURVEY 8(d). */
typedef struct {
    int64_t n, nnz_a;        /* order, entries of the analysed triangle                      */
    int64_t lnz;             /* sum_j c_j: nnz(L) for the permutation used (simplicial count)  */
    double  flops;           /* sum_j c_j^2                                                   */
    int64_t nsuper;          /* supernodes (fronts)                                           */
    int64_t lsize;           /* doubles stored for L (dense panels, >= lnz)                   */
    int64_t nlevels;         /* elimination-tree levels = dependent launch stages             */
    int64_t max_front;       /* largest front order m                                         */
    int64_t upd_size;        /* doubles in the update-matrix workspace                        */
    int64_t is_numeric;      /* 1 after a successful factorize                                */
    int64_t minor;
    int64_t solve_rowidx;    /* sum_s m_s: index entries read per triangular sweep           */
    int64_t is_ll;           /* 1: LL' factor, 0: LDL' (options['supernodal'] = 0, or 1 on a sparse factor) */
    int64_t dev_bytes;       /* bytes of the large device buffers of this handle (panels, inverted diagonal blocks, update
                              * matrices, values); 0 before the first device use.  Sharded mode: THIS rank's share          */
    int64_t lsize_local;     /* panel doubles resident on this rank (= lsize except in sharded mode: own + shared fronts)   */
    int64_t reserved[2];
} kvx_chol_info;
int kvx_chol_get_info(kvx_chol *F, kvx_chol_info *info);
int kvx_chol_get_perm(kvx_chol *F, int64_t *perm);     /* final permutation (ordering o postorder) */
/* Supernode layout for tests: super[nsuper+1] column starts, nrows[nsuper] front orders,
 * parent[nsuper], level[nsuper]. Any pointer may be NULL. */
int kvx_chol_get_supernodes(kvx_chol *F, int64_t *super, int64_t *nrows, int64_t *parent, int64_t *level);
/* Row structure of the fronts (CHOLMOD's L->s / L->pi of a supernodal factor, cholmod.c:927-943): rowptr (nsuper + 1) and the
 * sorted permuted row indices of every front, pivot rows first.  rowidx may be NULL (sizes only: rowptr[nsuper] entries). */
int kvx_chol_get_front_rows(kvx_chol *F, int64_t *rowptr, int64_t *rowidx);
/* Dominant-kernel timing of the last factorize/solve, measured with HIP events on the
 * factor's own stream (bench.py roofline leg). ms_factor / ms_solve may be NULL. */
int kvx_chol_last_timing(kvx_chol *F, double *ms_factor, double *ms_solve);
/* Which form the last kvx_chol_factorize_solve* call took: 1 = the one-enqueue form (forward sweep pipelined behind the
 * factorisation, one launch graph), 2 = factorisation and solve as two enqueues (sharded / LDL' factors, more than 16
 * right-hand sides, family timing, KVX_NO_GRAPH, and every process whose HIP runtime is older than 7.2 -- the runtime inside
 * the PyTorch wheel, 7.0.51831, crashes in hipGraphLaunch of the one-enqueue graph), 0 = no such call yet. */
int kvx_chol_last_fused_path(kvx_chol *F);
/* Per-kernel-family timing for the roofline leg of bench.py: HIP events are recorded around
 * every launch of ONE family on the factor's own stream while it is selected.
 * family: -1 off, 0 scatter_a, 1 front_small (LDS fronts), 2 assemble_big, 3 potrf_diag,
 * 4 trsm_panel, 5 syrk_trailing (FP64 MFMA), 6 fwd_level, 7 bwd_level.
 * prof_read returns the summed kernel time and launch count since the last select. */
int kvx_chol_prof_select(kvx_chol *F, int family);
int kvx_chol_prof_read(kvx_chol *F, double *total_ms, int64_t *launches);
/* ---- sharded mode: ONE system factored and solved by nranks processes, one GPU each ----------------------------
 * (the reference is single-process: the calls this stands in for are cholmod_l_factorize / cholmod_l_solve,
 * src/C/cholmod.c:362-364, 483.)  Every rank analyses the same matrix (the analysis is deterministic) and computes the
 * same map (kvx_chol_dist_map): proportional mapping of the elimination tree gives every front a contiguous range of ranks;
 *   - a range of one rank owns the front and its whole subtree: factored and solved by that rank alone;
 *   - small fronts shared by several ranks are replicated on them (identical arithmetic, nothing exchanged inside the front);
 *   - shared fronts of order >= min_m are block-cyclic: columns dealt out in blocks of `ob` columns round-robin over the
 *     range; right-looking, the owner of a pivot block factors the panel and broadcasts it (the panel ends up on every rank
 *     of the range), every rank applies the rank-ob update to the blocks it owns -- pivot columns and update matrix alike.
 * Exchange steps, all broadcasts inside the range of the receiving front: a child's update matrix (its column blocks, from
 * their owners) before the parent assembles; each factored panel; a child's update vector per solve.  One all-reduce over
 * all ranks ends a solve (every rank contributes the entries of x it reports) and one MIN ends a factorisation (failing
 * column).  No zero padding: only lower trapezoids of the blocks travel.
 * The library is collective-agnostic: it packs a message into the caller's device buffer (kvx_chol_dist_set_xchg) and calls
 * `comm` for every collective; kvxopt_amd/dist.py runs them with torch.distributed (backend "nccl" = RCCL over xGMI on a
 * node, "gloo" in the tests), a C caller would call ncclBroadcast / ncclAllReduce on its stream.  Stream contract of the
 * callback: the message is complete in null-stream order when `comm` is entered, and the collective's result must be
 * visible in null-stream order when it returns (no host synchronisation needed).  Return nonzero to abort (KVX_ECOMM). */
#define KVX_ECOMM         8   /* a collective callback failed -> RuntimeError */
#define KVX_DIST_BCAST        1   /* buf[0..count) from global rank `root` to the ranks [lo, hi)            */
#define KVX_DIST_ALLREDUCE    2   /* element-wise SUM over the ranks [lo, hi), result on every one of them   */
#define KVX_DIST_ALLREDUCE_MIN 3  /* element-wise MIN                                                        */
typedef struct { int32_t kind, root, lo, hi; int64_t count; double *buf_dev; } kvx_dist_op;
typedef int (*kvx_dist_comm_fn)(void *ctx, const kvx_dist_op *op);
/* host-only: the map for nranks ranks.  glo/ghi/mode: nsuper entries each (mode 0 = owned or replicated, 1 = block-cyclic);
 * rank_flops / panel_flops: nranks doubles -- factorisation flops each rank executes, and the part of them that is panel
 * factorisation of block-cyclic fronts; totals[0] = flops of all fronts (sum over fronts of sum_{j<k} (m-j)^2: the work executed, amalgamation zeros
 * included), totals[1] = flops of the replicated shared fronts.
 * Any output may be NULL.  ob <= 0 / min_m <= 0: defaults (512, 6144). */
int kvx_chol_dist_map(kvx_chol *F, int nranks, int ob, int min_m, int32_t *glo, int32_t *ghi, uint8_t *mode,
                      double *rank_flops, double *panel_flops, double totals[2]);
/* info[0] = n, [1] = smallest exchange buffer (doubles), [2] = recommended, [3] = shared fronts this rank takes part in,
 * [4] = of which block-cyclic, [5] = number of distinct rank ranges (kvx_chol_dist_groups), [6] = ob, [7] = min_m */
int kvx_chol_dist_setup(kvx_chol *F, int rank, int nranks, int ob, int min_m, int64_t info[8]);
int kvx_chol_dist_groups(kvx_chol *F, int32_t *lohi /* 2 * info[5] entries: lo, hi pairs; ranges of >= 2 ranks */);
int kvx_chol_dist_set_xchg(kvx_chol *F, double *xchg_dev, int64_t count);
/* numeric factorisation of the sharded factor; values_dev identical on all ranks.  *minor as kvx_chol_factorize_dev,
 * already the minimum over the ranks (every rank returns the same status). */
int kvx_chol_dist_factorize(kvx_chol *F, const double *values_dev, kvx_dist_comm_fn comm, void *ctx, int64_t *minor);
/* A X = B in place: B_dev (n x nrhs, ld = ldB) identical on all ranks on entry, the full solution on every rank on return. */
int kvx_chol_dist_solve(kvx_chol *F, double *B_dev, int64_t nrhs, int64_t ldB, kvx_dist_comm_fn comm, void *ctx);

/* ---- RCCL bound directly: the collectives of the sharded mode without torch.distributed -------------------------------
 * One process per GPU.  librccl.so is opened with dlopen at the first of these calls (KVX_RCCL_LIB overrides the path).
 * Rank 0 makes the id (kvx_rccl_unique_id) and hands its 128 bytes to the other ranks by any channel (kvxopt_amd/rccl.py: a
 * TCP socket on MASTER_ADDR:MASTER_PORT); every rank then calls kvx_rccl_init on the device it has made current (one rank per
 * device: RCCL refuses two).  kvx_rccl_split is collective over ALL ranks and creates the communicator of one rank range of
 * the factor's map (the ranges of kvx_chol_dist_groups, same order on every rank).  kvx_rccl_comm is a kvx_dist_comm_fn whose
 * ctx is the kvx_rccl*: pass both to kvx_chol_dist_factorize / kvx_chol_dist_solve.  The *_host calls move a few doubles
 * (at most 1024) through a device scratch buffer: timing rules and barriers of a benchmark.  Nothing here has a counterpart in
 * the reference (single process, src/C/cholmod.c:85). */
typedef struct kvx_rccl kvx_rccl;
int kvx_rccl_version(int *version);                       /* ncclGetVersion of the library that was opened */
int kvx_rccl_unique_id(char id[128]);
int kvx_rccl_init(int rank, int nranks, const char id[128], kvx_rccl **out);
int kvx_rccl_split(kvx_rccl *W, int lo, int hi);
int kvx_rccl_comm(void *ctx, const kvx_dist_op *op);
int kvx_rccl_allreduce_host(kvx_rccl *W, double *vals, int n, int op /* 0 SUM, 1 MAX, 2 MIN */);
int kvx_rccl_allgather_host(kvx_rccl *W, const double *mine, int n, double *all /* n * nranks */);
int kvx_rccl_barrier(kvx_rccl *W);                        /* every rank is here and its earlier device work is complete */
int kvx_rccl_stats(kvx_rccl *W, int64_t out[2]);          /* collectives through kvx_rccl_comm, bytes they moved */
void kvx_rccl_free(kvx_rccl *W);

void kvx_chol_free(kvx_chol *F);
void kvx_free(void *p);

/* ---- normal-equations assembly: replaces base.gemm(partial)+base.syrk(partial) ---------- */

/* Plan S = G' diag(w) G (+ P) on a fixed lower-triangular CCS pattern (misc.py:1418-1426,
 * 1451-1455 -> sparse.c:1260-1283, 2176-2256).  G is ml x n CCS.  The plan computes the
 * pattern of tril(G'G) (union with the lower pattern of P when Pp != NULL) and returns it
 * through kvx_atda_pattern(). */
int kvx_atda_plan(int64_t ml, int64_t n, const int64_t *Gp, const int64_t *Gi,
                  const int64_t *Pp, const int64_t *Pi, kvx_atda **out);
int kvx_atda_pattern(kvx_atda *T, int64_t *snz, int64_t *Sp, int64_t *Si);
/* Sx[snz] = sum_k w[k] G[k,i] G[k,j] (+ P[i,j]); w = di.^2. Host and device variants. */
int kvx_atda_assemble(kvx_atda *T, const double *Gx, const double *w, const double *Px, double *Sx);
int kvx_atda_assemble_dev(kvx_atda *T, const double *Gx_dev, const double *w_dev,
                          const double *Px_dev, double *Sx_dev);
/* the same with the square roots of the weights: S = G' diag(di)^2 G (+ P), what misc.kkt_chol2 forms (misc.py:1418-1426: W^-1
 * applied to G, then syrk) -- the square is taken while G is scaled, same roundings as ssqr followed by assemble */
int kvx_atda_assemble_sq_dev(kvx_atda *T, const double *Gx_dev, const double *di_dev, const double *Px_dev, double *Sx_dev);
void kvx_atda_free(kvx_atda *T);

/* ---- Nesterov-Todd scaling, orthant ('l') cone: replaces kvxopt.misc / misc_solvers ------ */
/* All vectors are device pointers of length ml (x: ml x ncols with leading dimension ldx). */

/* compute_scaling 'l' block (misc.py:284-287): d = sqrt(s./z), di = 1./d, lmbda = sqrt(s.*z) */
int kvx_nt_compute_scaling_dev(int64_t ml, const double *s, const double *z,
                               double *d, double *di, double *lmbda);
/* update_scaling 'l' block (misc.py:444-464), in place: s:=sqrt(s), z:=sqrt(z),
 * d:=d.*s./z, di:=1./d, lmbda:=s.*z */
int kvx_nt_update_scaling_dev(int64_t ml, double *s, double *z, double *d, double *di, double *lmbda);
/* scale 'l' block (misc_solvers.c:132-141): x := w .* x for each of ncols columns */
int kvx_nt_scale_dev(int64_t ml, int64_t ncols, int64_t ldx, double *x, const double *w);
/* scale2 (misc_solvers.c:287-298): inverse=0: x := x./lmbda ; inverse=1: x := x.*lmbda */
int kvx_nt_scale2_dev(int64_t ml, const double *lmbda, double *x, int inverse);
/* sprod (misc_solvers.c:662-669) x := x.*y ; sinv (:793-800) x := x./y ; ssqr (misc.py:951-952) x := y.*y */
int kvx_nt_sprod_dev(int64_t ml, double *x, const double *y);
int kvx_nt_sinv_dev(int64_t ml, double *x, const double *y);
int kvx_nt_ssqr_dev(int64_t ml, double *x, const double *y);
/* sdot (misc_solvers.c:1018) and max_step (misc_solvers.c:1065-1071: max_i -x_i) */
int kvx_nt_sdot_dev(int64_t ml, const double *x, const double *y, double *result_host);
int kvx_nt_max_step_dev(int64_t ml, const double *x, double *result_host);
/* Fused 'l'-cone steps of one interior-point iteration (each replaces a fixed run of the reference's BLAS-1 /
 * misc calls, same operations per element; the loop is bound by launches, not by bytes):
 *   newton_rhs: ds := -(lmbdasq (+ ws3 - shift)) ./ lmbda,  dz := -(scale*rz + d.*ds)      coneprog.py:1250-1298, 1146-1157
 *               (ws3 may be NULL: predictor);
 *   step_post : dz += dtau*z1, ds -= dz, [ws3 := ds.*dz], ds ./= lmbda, dz ./= lmbda        coneprog.py:1186-1191, 1303-1316
 *   update    : ds := (step*ds + 1).*lmbda (same for dz), misc.update_scaling (misc.py:444-464), s := W'lmbda,
 *               z := W^-1 lmbda                                                             coneprog.py:1343-1432  */
int kvx_lp_newton_rhs_dev(int64_t ml, const double *lmbdasq, const double *ws3, double shift, double scale, const double *rz,
                          const double *lmbda, const double *d, double *ds, double *dz);
int kvx_lp_step_post_dev(int64_t ml, double dtau, const double *z1, const double *lmbda, double *ds, double *dz, double *ws3);
int kvx_lp_update_dev(int64_t ml, double step, double *ds, double *dz, double *d, double *di, double *lmbda, double *s, double *z);
/* (newton_rhs: lmbdasq may be NULL -- lmbda o lmbda is then formed in the kernel, rounded as misc.ssqr rounds it.)
 * The remaining short launches of an iteration, fused (round 3; a launch of a few microseconds costs 4-5 us on its stream, ~85 of them
 * were a fifth of an iteration).  Same arithmetic per element, same roundings: an interior-point run is bit for bit the unfused one.
 *   update_x  : kvx_lp_update_dev and x += step*dx (coneprog.py:1339) in one launch;
 *   residuals : hrx := -G'z, rx := hrx - tau*c, hrz := G*x + s, rz := hrz - tau*h (coneprog.py:861-896, p = 0) in one launch; G by its
 *               CCS arrays and by those of its transpose (device pointers, int64 indices, as for kvx_spmv_dev);
 *   kkt_solve_pre / _post: misc.kkt_chol2's solve (misc.py:1489-1563, p = 0) around the triangular solves, for one or two right-hand
 *               sides:  x2(:,k) := xscale*xin + G'(di.*(zin.*di))   [z := W^-1 z, x += Gs'z]   and, after S^-1 on x2,
 *               xout := xoscale*x2(:,k),  zout := zoscale*(di.*(G*x2(:,k)) - zin.*di)           [z := Gs*x - z].
 *   max_col_nnz / max_row_nnz: the largest number of entries in a column / row of G (0 = not known): with at most four (eight), a
 *               column or row is summed by 4 (8) lanes instead of 16 -- the same association, the same bits. */
typedef struct kvx_kkt_side {
    const double *xin;  double xscale;  const double *zin;     /* pre */
    double *xout;       double xoscale; double *zout; double zoscale;   /* post (zin is read again) */
} kvx_kkt_side;
int kvx_lp_update_x_dev(int64_t ml, int64_t n, double step, double *ds, double *dz, double *d, double *di, double *lmbda, double *s,
                        double *z, const double *dx, double *x);
int kvx_lp_residuals_dev(int64_t ml, int64_t n, const int64_t *Gp, const int64_t *Gi, const double *Gx, int64_t max_col_nnz,
                         const int64_t *GTp, const int64_t *GTi, const double *GTx, int64_t max_row_nnz, const double *x, const double *z, const double *s,
                         const double *c, const double *h, double tau, double *hrx, double *rx, double *hrz, double *rz);
int kvx_kkt_solve_pre_dev(int64_t ml, int64_t n, const int64_t *Gp, const int64_t *Gi, const double *Gx, int64_t max_col_nnz,
                          const double *di, int nrhs, const kvx_kkt_side *sides, double *x2, int64_t ldx2);
int kvx_kkt_solve_post_dev(int64_t ml, int64_t n, const int64_t *GTp, const int64_t *GTi, const double *GTx, int64_t max_row_nnz,
                           const double *di, int nrhs, const kvx_kkt_side *sides, const double *x2, int64_t ldx2);
/* count <= 32 reductions with one host synchronisation: kind[i] = 0 sdot(x_i, y_i), 1 max_step(x_i) -- bitwise the
 * values of the single calls (the per-iteration residual norms / objectives of coneprog.py:861-896 in one go). */
/* second half of f6_no_ir + step bounds (coneprog.py:1162-1195, 1303-1321) in one host round trip: dtau is formed on the
 * device from c'dx + b'dy + th'dz (and z1'z1 when z1z1 < 0) and consumed there by dx += dtau x1, dy += dtau y1,
 * dz += dtau z1, ds -= dz, [ws3 = ds.*dz], ds ./= lmbda, dz ./= lmbda.  out_host = {dtau, z1'z1, max(-ds), max(-dz)}. */
int kvx_lp_second_half_dev(int64_t ml, int64_t n, int64_t p, const double *c, const double *b, const double *th, const double *x1,
                           const double *y1, const double *z1, const double *lmbda, double *dx, double *dy, double *dz, double *ds,
                           double *ws3, double dgi, double dtau0, double z1z1, double out_host[4]);
int kvx_nt_reduce_multi_dev(int count, const int32_t *kind, const int64_t *n, const double *const *x,
                            const double *const *y, double *out_host);
/* One interior-point iteration of conelp on the orthant without equality constraints (coneprog.py:859-1436, p = 0) in FOUR calls
 * instead of fifteen (round 4): the same kernels in the same order as the calls above -- bit for bit the same iterates -- but
 * the launches between two host synchronisations are issued from C back to back (a Python -> ctypes hop costs 10-15 us, and
 * the GPU waited for them after every synchronisation: ~250 us of a 1.9 ms iteration, rocprofv3 timeline).  All pointers are
 * device pointers; plan / F / Sx / x2 are the KKT object's (S = G' W^-2 G on its fixed pattern, its factor, two columns of work).
 *   residuals : kvx_lp_residuals_dev + the reductions of coneprog.py:861-896 -> out[10] = hrx'hrx, rx'rx, 0, 0, hrz'hrz, rz'rz,
 *               c'x, 0, h'z, lmbda'lmbda (the order lp.py reads them in);
 *   predictor : kvx_lp_newton_rhs_dev (predictor), assembly of S, kvx_kkt_solve_pre_dev, factorisation + two-column solve as one
 *               enqueue, kvx_kkt_solve_post_dev [x1 := dgi S^-1(-c ..), z1; dx, dz], th := h .* di, kvx_lp_second_half_dev
 *               -> out[4] = dtau, z1'z1, max(-ds), max(-dz); the factorisation's status stays deferred (kvx_chol_status);
 *   corrector : kvx_lp_newton_rhs_dev (shift = sigma mu, scale = 1 - sigma, ws3), KKT solve with the factor, second half;
 *   update    : kvx_lp_update_x_dev(step), then `residuals` of the next iteration with tau_next. */
typedef struct kvx_lp_ctx {
    int64_t ml, n;
    const int64_t *Gp, *Gi; const double *Gx; int64_t max_col;
    const int64_t *GTp, *GTi; const double *GTx; int64_t max_row;
    kvx_atda *plan; kvx_chol *F; double *Sx, *x2;
    double *x, *s, *z, *c, *h, *hrx, *rx, *hrz, *rz, *lmbda, *d, *di, *ds, *dz, *dx, *x1, *z1, *th, *ws3;
} kvx_lp_ctx;
int kvx_lp_iter_residuals(const kvx_lp_ctx *L, double tau, double out[10]);
int kvx_lp_iter_predictor(const kvx_lp_ctx *L, double dgi, double dtau0, double out[4]);
int kvx_lp_iter_corrector(const kvx_lp_ctx *L, double shift, double scale, double dgi, double dtau0, double z1z1, double out[4]);
int kvx_lp_iter_update(const kvx_lp_ctx *L, double step, double tau_next, double out[10]);

/* ---- BLAS-1 glue on device vectors: replaces the blas.axpy / scal / copy calls and elementwise
 * products the interior-point loop makes between KKT solves (coneprog.py:1126-1433). ------------- */
int kvx_vec_axpy_dev(int64_t n, double alpha, const double *x, double *y);
/* z := a x + b y in one pass (blas.copy + blas.axpy / blas.scal pairs of coneprog.py:861-896, 1295-1298); b == 0: z := a x */
int kvx_vec_lincomb_dev(int64_t n, double a, const double *x_dev, double b, const double *y_dev, double *z_dev);     /* y += alpha x        */
int kvx_vec_scal_dev(int64_t n, double alpha, double *x);                       /* x *= alpha          */
int kvx_vec_addc_dev(int64_t n, double c, double *x);                           /* x += c              */
int kvx_vec_fill_dev(int64_t n, double c, double *x);                           /* x := c              */
int kvx_vec_copy_dev(int64_t n, const double *x, double *y);                    /* y := x              */
int kvx_vec_copy_strided_dev(int64_t n, const double *x, int64_t incx, double *y);   /* y[i] := x[i * incx] (e.g. a dense diagonal) */
int kvx_vec_xmy_dev(int64_t n, double a, const double *x, const double *y, double b, double *z); /* z := a x.*y + b z */

/* ---- sparse mat-vec: replaces base.gemv on spmatrix (sparse.c:1073-1104) ----------------- */
/* y := alpha*op(A)*x + beta*y, A m x n CCS with int64 indices on the device. trans 'N'/'T'. */
int kvx_spmv_dev(int trans, int64_t m, int64_t n, const int64_t *Ap_dev, const int64_t *Ai_dev,
                 const double *Ax_dev, double alpha, const double *x_dev, double beta, double *y_dev);

/* ---- second-order-cone ('q') blocks of the Nesterov-Todd scaling: the 'q' parts of
 * misc.compute_scaling / update_scaling (misc.py:290-352, 467-580), misc_solvers.scale / scale2 / sprod / sinv
 * (misc_solvers.c:144-186, 301-341, 671-700, 803-835), misc.ssqr (misc.py:951-959), max_step (misc_solvers.c:1073-1085).
 * off_dev: nq + 1 cone boundaries [0, q0, q0 + q1, ...]; every vector pointer addresses the START of the 'q' section of its
 * vector (i.e. the caller adds mnl + dims['l']); one workgroup per cone, all cones of a call in one launch (null stream). */
int kvx_ntq_compute_scaling_dev(int64_t nq, const int64_t *off_dev, const double *s_dev, const double *z_dev, double *v_dev,
                                double *beta_dev /* nq */, double *lmbda_dev);
/* in place: s, z leave as st / a, zt / b (misc.py:517-523); v, beta, lmbda are updated */
int kvx_ntq_update_scaling_dev(int64_t nq, const int64_t *off_dev, double *s_dev, double *z_dev, double *v_dev, double *beta_dev,
                               double *lmbda_dev);
/* x_k := beta_k (2 v_k v_k' - J) x_k for every column of x (leading dimension ldx), or the inverse scaling */
int kvx_ntq_scale_dev(int64_t nq, const int64_t *off_dev, const double *v_dev, const double *beta_dev, double *x_dev, int64_t ldx,
                      int64_t ncols, int inverse);
int kvx_ntq_scale2_dev(int64_t nq, const int64_t *off_dev, const double *lmbda_dev, double *x_dev, int inverse);
/* op 0: x := y o x (sprod);  1: x := y o\ x (sinv);  2: x := y o y (ssqr) */
int kvx_ntq_prod_dev(int64_t nq, const int64_t *off_dev, double *x_dev, const double *y_dev, int op);
/* out_dev[k] = |x_k1| - x_k0 (the caller takes the maximum) */
int kvx_ntq_max_step_dev(int64_t nq, const int64_t *off_dev, const double *x_dev, double *out_dev);

/* ---- semidefinite ('s') blocks of the Nesterov-Todd scaling: the 's' parts of misc.compute_scaling /
 * update_scaling (misc.py:354-419, 582-634), misc_solvers.scale / scale2 / sprod / sinv / sdot / max_step
 * (misc_solvers.c:188-240, 343-397, 700-770, 845-882, 1029-1046, 1086-1160) and pack / unpack / symm / trisc / triusc
 * (misc_solvers.c:412-632, 887-988).  Block k has order m_k; off2_dev: ns + 1 offsets of the blocks in the 's' section of a
 * vector [0, m0^2, m0^2 + m1^2, ...]; off1_dev: ns + 1 offsets of their diagonals / eigenvalues [0, m0, m0 + m1, ...].  Vector
 * pointers address the START of the 's' section (r, rti: the blocks W['r'][k], W['rti'][k] back to back; lmbda: its 's' part).
 * One workgroup per block, all blocks of a call in one launch (null stream).  work_dev: scratch, sizes given per entry. */
/* r_k' z_k r_k = r_k^-1 s_k r_k^-T = diag(lmbda_k), rti_k = r_k^-T; lmbda_k descending as lapack.gesvd returns it.
 * work: 4 sum m_k^2 doubles.  status_dev: int, preset to INT_MAX; a block that is not positive definite stores its failing
 * column with atomicMin (the reference's lapack.potrf raises ArithmeticError). */
int kvx_nts_compute_scaling_dev(int64_t ns, const int64_t *off2_dev, const int64_t *off1_dev, const double *s_dev,
                                const double *z_dev, double *r_dev, double *rti_dev, double *lmbda_dev, double *work_dev,
                                int *status_dev);
/* in place (misc.py:582-634): on entry s_k, z_k hold the factors Ls, Lz of the new iterates in the current scaling, on return
 * the singular vectors U and V' of Lz' Ls; r, rti, lmbda are updated.  work: 4 sum m_k^2 doubles */
int kvx_nts_update_scaling_dev(int64_t ns, const int64_t *off2_dev, const int64_t *off1_dev, double *s_dev, double *z_dev,
                               double *r_dev, double *rti_dev, double *lmbda_dev, double *work_dev);
/* x_k := R_k' X_k R_k (form 0) or R_k X_k R_k' (form 1) for every column of x (leading dimension ldx); X_k is the symmetric
 * matrix stored in the lower triangle of x_k and only that triangle is written.  R = r or rti as the reference chooses by
 * (trans, inverse).  work: ncols * wstride doubles, wstride >= sum m_k^2 */
int kvx_nts_scale_dev(int64_t ns, const int64_t *off2_dev, const int64_t *off1_dev, const double *R_dev, double *x_dev,
                      int64_t ldx, int64_t ncols, int form, double *work_dev, int64_t wstride);
/* x_k(i, j) := x_k(i, j) / (sqrt(l_i) sqrt(l_j)) (inverse 0) or times it (inverse 1), all entries of the block */
int kvx_nts_scale2_dev(int64_t ns, const int64_t *off2_dev, const int64_t *off1_dev, const double *lmbda_dev, double *x_dev,
                       int inverse);
/* op 0: x := y o x with full 's' blocks in y (the upper triangles of y are filled by mirroring, as the reference does;
 * work: sum m_k^2);  op 1 / 2: sprod / sinv with DIAGONAL 's' blocks -- y_dev then addresses the diagonals (off1 layout) */
int kvx_nts_prod_dev(int64_t ns, const int64_t *off2_dev, const int64_t *off1_dev, double *x_dev, double *y_dev, int op,
                     double *work_dev);
/* out_dev[k] = trace inner product of the symmetric matrices stored in the lower triangles of x_k, y_k (the caller adds) */
int kvx_nts_dot_dev(int64_t ns, const int64_t *off2_dev, const int64_t *off1_dev, const double *x_dev, const double *y_dev,
                    double *out_dev);
/* out_dev[k] = -(smallest eigenvalue of x_k); with sigma_dev != NULL the eigenvalues (ascending) are stored there and the
 * eigenvectors replace x_k (misc_solvers.c:1128-1133), otherwise x is not modified.  work: 3 sum m_k^2 + 2 sum m_k */
int kvx_nts_max_step_dev(int64_t ns, const int64_t *off2_dev, const int64_t *off1_dev, double *x_dev, double *sigma_dev,
                         double *out_dev, double *work_dev);
/* mode 0 symm (upper := mirror of lower), 1 trisc (upper := 0, strict lower *= 2), 2 triusc (strict lower *= 0.5) */
int kvx_nts_tri_dev(int64_t ns, const int64_t *off2_dev, const int64_t *off1_dev, double *x_dev, int mode);
/* dir 0 pack: packed (offp_dev: ns offsets [0, m0 (m0 + 1) / 2, ...]) := lower triangles of the blocks by columns, off-diagonal
 * entries times sqrt(2) (dir 2: pack2's form, which copies the diagonal instead of dividing and multiplying it by sqrt(2));
 * dir 1 unpack: the reverse into the lower triangles (the strict upper triangles are not touched) */
int kvx_nts_pack_dev(int64_t ns, const int64_t *off2_dev, const int64_t *off1_dev, const int64_t *offp_dev, double *full_dev,
                     double *packed_dev, int dir);

/* ---- dense helpers of the equality-constrained KKT solve with a general S (misc.py:1476-1487, 1545): K = A S^-1 A' formed
 * as a dense p x p matrix from X = S^-1 A' (kvx_chol_solve_dev with nrhs = p) when p is moderate ------------------------- */
/* Y(j, c) = sum_i A(i, j) X(i, c) for the CCS matrix A with n columns and every column c < ncols of the dense X */
int kvx_spmm_t_dev(int64_t n, int64_t ncols, const int64_t *Ap_dev, const int64_t *Ai_dev, const double *Ax_dev,
                   const double *X_dev, int64_t ldx, double *Y_dev, int64_t ldy);
/* D (m x n dense, leading dimension ld) := the CCS matrix */
int kvx_dense_from_ccs_dev(int64_t m, int64_t n, const int64_t *Ap_dev, const int64_t *Ai_dev, const double *Ax_dev,
                           double *D_dev, int64_t ld);
/* out := lower triangle of the dense p x p matrix K, column by column (the value array of a dense lower CCS pattern) */
int kvx_pack_lower_dev(int64_t p, const double *K_dev, int64_t ld, double *out_dev);
/* y := alpha A x + beta y, A dense column-major m x n (ld = lda), nrhs columns of x (ldx) and y (ldy): with X = S^-1 A' at hand,
 * S^-1 (b - A' uy) = S^-1 b - X uy is a product instead of a second solve with S (the role of misc.py:1545-1553) */
int kvx_dense_gemv_dev(int64_t m, int64_t n, int64_t nrhs, double alpha, const double *A_dev, int64_t lda, const double *x_dev, int64_t ldx,
                       double beta, double *y_dev, int64_t ldy);

/* ---- device memory plumbing for hosts without their own allocator ------------------------ */
int kvx_dev_malloc(void **p, int64_t bytes);
int kvx_dev_free(void *p);
int kvx_dev_upload(void *dst_dev, const void *src_host, int64_t bytes);
int kvx_dev_download(void *dst_host, const void *src_dev, int64_t bytes);
int kvx_dev_sync(void);
/* Device buffers, streams and events released by the library are kept in a caching pool (up to KVX_POOL_MAX_MB, default 8192)
 * and handed out again; kvx_dev_trim() gives everything cached back to the driver (e.g. before another framework needs the
 * memory).  No reference counterpart. */
int kvx_dev_trim(void);
int kvx_dev_mem_info(int64_t *free_bytes, int64_t *total_bytes);   /* hipMemGetInfo of the current device */

/* ---------------------------------------------------------------------------------------------------------
 * Sparse LU (the kvxopt.klu API, src/C/klu.c; SURVEY 8(f)1, BASELINE configs[2]).  Real 'd' matrices, square,
 * CCS with int64 indices.  Static-structure multifrontal LU with threshold partial pivoting inside the pivot
 * block of each front; fronts without an acceptable pivot are merged into their parents and the factorisation
 * repeats (kvxopt_amd/csrc/lu_symbolic.hpp).  Factorisation:  R P A Q = L U + F  as in KLU: P, Q bring A to block upper
 * triangular form (strongly connected components after the matching), L U are the factors of the diagonal blocks, F the
 * off-diagonal blocks (never eliminated), L unit lower, R = diag(1 / Rs).  (Chains of more than 64 dependent block levels
 * are factored as one block instead: they would serialise the solves.)
 * --------------------------------------------------------------------------------------------------------- */
typedef struct kvx_lu_sym kvx_lu_sym;      /* replaces the "KLU SYM D FACTOR" capsule (klu.c:36,276-279)   */
typedef struct kvx_lu_num kvx_lu_num;      /* replaces the "KLU NUM D FACTOR" capsule (klu.c:38,336-339)   */

/* symbolic(A) -- klu.c:242-291 (klu_analyze :264).  values may be NULL (pattern only: plain maximum
 * transversal); with values the matching maximises the product of the scaled diagonal.  A structurally
 * singular pattern is accepted here and reported by the numeric phase (KVX_ESINGULAR), as KLU does.
 * KVX_EINVAL: n < 1, malformed colptr/rowind. */
int kvx_lu_analyze(int64_t n, const int64_t *colptr, const int64_t *rowind, const double *values, kvx_lu_sym **out);
void kvx_lu_free_symbolic(kvx_lu_sym *S);
/* info: n, nnz, base supernodes, merges learned so far, structurally singular (0/1), nnz(L) bound of the
 * symmetrised pattern, levels, largest base front */
int kvx_lu_sym_info(kvx_lu_sym *S, int64_t info[8]);
int kvx_lu_sym_matching(kvx_lu_sym *S, int64_t *rowfor /* n: row on the diagonal of column j */);
/* block triangular form (KLU's BTF): number of diagonal blocks, number of block levels of the back substitution, and (blk
 * may be NULL) the block of every column j -- row rowfor[j] is in the same block; entries satisfy blk[row] <= blk[col] */
int kvx_lu_sym_btf(kvx_lu_sym *S, int64_t *nblocks, int64_t *nlevels, int64_t *blk);

/* numeric(A, Fs) -- klu.c:310-379 (klu_factor :336).  nnz must equal the analysed pattern's.  The symbolic
 * object is updated when fronts are merged (it must outlive the numeric object).
 * KVX_ESINGULAR -> ArithmeticError("singular matrix") (klu.c:370-371); KVX_EDEVICE: no GPU (never a CPU fallback). */
/* Stream contract of every *_dev entry point of this header (Cholesky, LU, KKT): the library works on its own non-blocking
 * streams; on entry it orders them behind whatever the caller has already submitted to the legacy null stream (torch's
 * default stream, the kvx_nt_* / kvx_atda_* / kvx_spmv_* kernels), so a buffer written by a kernel immediately before the
 * call is read after that kernel.  Synchronous entry points return with their results complete; the *_async_* ones order
 * the null stream behind their own work instead. */
int kvx_lu_factor(kvx_lu_sym *S, int64_t nnz, const double *values, kvx_lu_num **out);
int kvx_lu_factor_dev(kvx_lu_sym *S, int64_t nnz, const double *values_dev, kvx_lu_num **out);
/* numeric with a previous factorisation (doc/source/spsolvers.rst:377-388: "a refactorization is performed"):
 * same pivot sequence, no search; falls back to a full factorisation when a reused pivot is unacceptable. */
int kvx_lu_refactor(kvx_lu_num *N, int64_t nnz, const double *values);
int kvx_lu_refactor_dev(kvx_lu_num *N, int64_t nnz, const double *values_dev);
void kvx_lu_free_numeric(kvx_lu_num *N);
/* info: fronts, levels, largest front order, largest pivot block, panel doubles, arena doubles, numeric passes, factored */
int kvx_lu_num_info(kvx_lu_num *N, int64_t info[8]);
/* Work of one numeric factorisation of the plan (bench.py roofline of the LU path): work[0] = flops (2 per multiply-add: pivot
 * block, the two panels and the rank-k update of every front), [1] = sum over fronts of the L and U panel entries 2 m k - k^2,
 * [2] = sum over fronts of u^2 (update matrices written once and read once by the parent), [3] = fronts with m > the
 * one-workgroup limit (blocked path), [4] = their share of the flops.  Algorithmic bytes of a factorisation:
 * 8 (work[1] + 2 work[2]) + 12 nnz(A). */
int kvx_lu_num_work(kvx_lu_num *N, double work[5]);
/* Launch graphs of the steady state (klu.c:296-308 refactorisation on the recorded pivot sequence, klu.c:651-665 solves on the same
 * buffers): how many times a captured sequence of launches has been replayed by this factor (0: every pass went out launch by
 * launch -- first factorisations always do, and so does a call whose buffers differ from the previous call's: a sequence is
 * captured when the same buffers come twice in a row; KVX_LU_GRAPH=0 turns the graphs off). */
int kvx_lu_num_graph_replays(kvx_lu_num *N, int64_t *replays);

/* solve(A, Fs, Fn, B, trans) -- klu.c:593-690 (klu_solve / klu_tsolve :651-665).  trans: 0 = 'N', 1 = 'T'.
 * B is n x nrhs column-major with leading dimension ldB >= max(1, n), overwritten by the solution. */
int kvx_lu_solve(kvx_lu_num *N, int trans, double *B, int64_t nrhs, int64_t ldB);
int kvx_lu_solve_dev(kvx_lu_num *N, int trans, double *B_dev, int64_t nrhs, int64_t ldB);

/* get_numeric(A, Fs, Fn) -- klu.c:392-566 (klu_extract :444).  L, U, F come back as malloc'ed CCS triples
 * (free with kvx_free), sorted rows, no explicit zeros; P[k] = row of A that is pivot row k, Q[k] = column of A
 * that is pivot column k, Rs[k] = scale factor of pivot row k (the reference inverts it, klu.c:503-509),
 * r = block boundaries (nblocks + 1 entries). */
int kvx_lu_extract(kvx_lu_num *N, int64_t *lnz, int64_t **Lp, int64_t **Li, double **Lx, int64_t *unz, int64_t **Up,
                   int64_t **Ui, double **Ux, int64_t *fnz, int64_t **Fp, int64_t **Fi, double **Fx, int64_t *P,
                   int64_t *Q, double *Rs, int64_t *nblocks, int64_t **r);

/* get_det(A, Fs, Fn) -- klu.c:707-828: prod(Udiag[k] * Rs[k]) * sign(P) * sign(Q). */
int kvx_lu_det(kvx_lu_num *N, double *det);

#ifdef __cplusplus
}
#endif
#endif
