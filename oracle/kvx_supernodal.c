/*
 * TEST / BENCH INFRASTRUCTURE ONLY (oracle/): host supernodal multifrontal Cholesky on all cores -- the CPU baseline
 * SURVEY 8(d) asks for beside the GPU numbers ("the build's own host restatement, OpenMP over etree subtrees + BLAS-3 in
 * fronts").  Nothing in kvxopt_amd/ links or calls this; bench.py's cpu_baseline leg and tests/ do.
 *
 * What it restates: the numeric phase and the solves of a supernodal CHOLMOD factorisation as the reference reaches it
 * (cholmod_l_factorize / cholmod_l_solve, src/C/cholmod.c:362-364, 483, Common.supernodal = 2, cholmod.c:96) -- dense
 * panels per supernode, BLAS-3 inside them (dpotrf / dtrsm / dsyrk), assembled multifrontally.  It works on the SAME
 * supernodes and permutation as the GPU library (passed in), so both sides execute the same flops.  BLAS / LAPACK: the
 * OpenBLAS inside scipy (LP64, symbols prefixed scipy_).  Parallelism: the independent subtrees below the top of the
 * elimination tree run one per OpenMP thread (largest first); the few big fronts of the top run front after front as a tiled
 * right-looking factorisation whose tile operations are spread over the OpenMP threads.  OpenBLAS always runs single-threaded.
 *
 * Parity: checked against oracle/kvx_oracle.c (simplicial up-looking, itself pinned on the reference's documented
 * answers) in tests/test_oracle.py.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;

extern void scipy_dpotrf_(const char *uplo, const int *n, double *a, const int *lda, int *info);
extern void scipy_dtrsm_(const char *side, const char *uplo, const char *trans, const char *diag, const int *m, const int *n,
                         const double *alpha, const double *a, const int *lda, double *b, const int *ldb);
extern void scipy_dsyrk_(const char *uplo, const char *trans, const int *n, const int *k, const double *alpha, const double *a,
                         const int *lda, const double *beta, double *c, const int *ldc);
extern void scipy_dgemm_(const char *ta, const char *tb, const int *m, const int *n, const int *k, const double *alpha,
                         const double *a, const int *lda, const double *b, const int *ldb, const double *beta, double *c, const int *ldc);
extern void scipy_dtrsv_(const char *uplo, const char *trans, const char *diag, const int *n, const double *a, const int *lda,
                         double *x, const int *incx);
extern void scipy_dgemv_(const char *trans, const int *m, const int *n, const double *alpha, const double *a, const int *lda,
                         const double *x, const int *incx, const double *beta, double *y, const int *incy);
extern void scipy_openblas_set_num_threads(int n);

typedef struct {
    i64 n, ns, nnz;
    i64 *super, *rowptr, *rowidx, *parent, *px, *amap, *perm;
    i64 *childptr, *children, *rel;      /* rel: per front row >= k, its position in the parent front */
    i64 *subptr, *sublist;               /* independent subtrees below the top: fronts of subtree t = sublist[subptr[t] .. subptr[t+1]), children first */
    i64 nsub;
    i64 *toplist;                        /* the fronts above them (few, big), children first */
    i64 ntop;
    double *Lx;
    double **U;                          /* update matrix of every front (u x u, lower), alive until its parent assembled */
    i64 minor;
    int nthreads;
} kvxs_factor;

void kvxs_free(kvxs_factor *F)
{
    if (!F) return;
    free(F->super); free(F->rowptr); free(F->rowidx); free(F->parent); free(F->px); free(F->amap); free(F->perm);
    free(F->childptr); free(F->children); free(F->rel); free(F->subptr); free(F->sublist); free(F->toplist); free(F->Lx);
    if (F->U) { for (i64 s = 0; s < F->ns; s++) free(F->U[s]); free(F->U); }
    free(F);
}

static void *zalloc(size_t bytes) { return calloc(bytes ? bytes : 1, 1); }

/* Split the tree for the host threads: a front whose subtree holds more than 1/(4 threads) of the work belongs to the top
 * (closed under parents); the maximal subtrees below are independent of each other -- one thread each, largest first. */
static void kvxs_partition(kvxs_factor *F)
{
    const i64 ns = F->ns;
    double *w = zalloc(sizeof(double) * ns), tot = 0.0;
    for (i64 s = 0; s < ns; s++) {
        const double k = (double)(F->super[s + 1] - F->super[s]), m = (double)(F->rowptr[s + 1] - F->rowptr[s]);
        w[s] += k * m * m + 1000.0;                                  /* (+ a per-front cost: assembly, calls) */
        tot += k * m * m + 1000.0;
        if (F->parent[s] >= 0) w[F->parent[s]] += w[s];
    }
    const double thresh = tot / (4.0 * (F->nthreads > 1 ? F->nthreads : 1));
    free(F->subptr); free(F->sublist); free(F->toplist);
    F->subptr = zalloc(sizeof(i64) * (ns + 1)); F->sublist = zalloc(sizeof(i64) * ns); F->toplist = zalloc(sizeof(i64) * ns);
    F->nsub = 0; F->ntop = 0;
    char *top = zalloc(ns);
    for (i64 s = 0; s < ns; s++) top[s] = F->nthreads > 1 ? (w[s] > thresh) : 0;
    /* roots of the independent subtrees, by decreasing work */
    i64 *roots = zalloc(sizeof(i64) * ns), nr = 0;
    for (i64 s = 0; s < ns; s++)
        if (!top[s] && (F->parent[s] < 0 || top[F->parent[s]])) roots[nr++] = s;
    for (i64 i = 1; i < nr; i++) {                                   /* insertion sort is fine: nr is a few hundred at most ... */
        if (nr > 4096) break;                                        /* ... unless the forest is flat (then order does not matter) */
        i64 r = roots[i], j = i;
        while (j > 0 && w[roots[j - 1]] < w[r]) { roots[j] = roots[j - 1]; j--; }
        roots[j] = r;
    }
    /* fronts of each subtree, children first: reverse of a depth-first walk from the root */
    i64 *stack = zalloc(sizeof(i64) * ns), pos = 0;
    for (i64 t = 0; t < nr; t++) {
        i64 sp = 0, start = pos;
        stack[sp++] = roots[t];
        while (sp > 0) {
            const i64 s = stack[--sp];
            F->sublist[pos++] = s;
            for (i64 q = F->childptr[s]; q < F->childptr[s + 1]; q++) stack[sp++] = F->children[q];
        }
        for (i64 a = start, b = pos - 1; a < b; a++, b--) { i64 x = F->sublist[a]; F->sublist[a] = F->sublist[b]; F->sublist[b] = x; }
        F->subptr[++F->nsub] = pos;
    }
    /* the top, children first: ascending index works when parents have larger numbers (postorder); otherwise sort by depth */
    for (i64 s = 0; s < ns; s++) if (top[s]) F->toplist[F->ntop++] = s;
    int postordered = 1;
    for (i64 s = 0; s < ns; s++) if (F->parent[s] >= 0 && F->parent[s] < s) postordered = 0;
    if (!postordered) {
        i64 *depth = zalloc(sizeof(i64) * ns);
        for (i64 pass = 0; pass < ns; pass++) {                      /* depth by relaxation (rare path, small inputs) */
            int changed = 0;
            for (i64 s = 0; s < ns; s++) if (F->parent[s] >= 0 && depth[s] != depth[F->parent[s]] + 1) { depth[s] = depth[F->parent[s]] + 1; changed = 1; }
            if (!changed) break;
        }
        for (i64 i = 1; i < F->ntop; i++) {
            i64 r = F->toplist[i], j = i;
            while (j > 0 && depth[F->toplist[j - 1]] < depth[r]) { F->toplist[j] = F->toplist[j - 1]; j--; }
            F->toplist[j] = r;
        }
        free(depth);
    }
    free(w); free(top); free(roots); free(stack);
}

/* A: lower CCS (Ap, Ai) of the analysed triangle; perm: new -> old; supernodes: super (ns + 1 first columns), rowptr /
 * rowidx (sorted permuted rows of every front, pivots first), parent (-1 = root), numbered in postorder. */
kvxs_factor *kvxs_analyze(i64 n, const i64 *Ap, const i64 *Ai, const i64 *perm, i64 ns, const i64 *super, const i64 *rowptr,
                          const i64 *rowidx, const i64 *parent)
{
    kvxs_factor *F = zalloc(sizeof(*F));
    if (!F) return NULL;
    F->n = n; F->ns = ns; F->nnz = Ap[n]; F->minor = n;
    F->nthreads = omp_get_max_threads() > 32 ? 32 : omp_get_max_threads();
    F->super = zalloc(sizeof(i64) * (ns + 1)); F->rowptr = zalloc(sizeof(i64) * (ns + 1));
    F->rowidx = zalloc(sizeof(i64) * rowptr[ns]); F->parent = zalloc(sizeof(i64) * ns);
    F->px = zalloc(sizeof(i64) * (ns + 1)); F->amap = zalloc(sizeof(i64) * F->nnz); F->perm = zalloc(sizeof(i64) * n);
    F->childptr = zalloc(sizeof(i64) * (ns + 2)); F->children = zalloc(sizeof(i64) * ns); F->rel = zalloc(sizeof(i64) * rowptr[ns]);
    F->U = zalloc(sizeof(double *) * ns);
    memcpy(F->super, super, sizeof(i64) * (ns + 1)); memcpy(F->rowptr, rowptr, sizeof(i64) * (ns + 1));
    memcpy(F->rowidx, rowidx, sizeof(i64) * rowptr[ns]); memcpy(F->parent, parent, sizeof(i64) * ns);
    memcpy(F->perm, perm, sizeof(i64) * n);
    for (i64 s = 0; s < ns; s++) F->px[s + 1] = F->px[s] + (rowptr[s + 1] - rowptr[s]) * (super[s + 1] - super[s]);
    F->Lx = zalloc(sizeof(double) * F->px[ns]);
    /* children lists, depths, level lists (level = distance from the root) */
    for (i64 s = 0; s < ns; s++) if (parent[s] >= 0) F->childptr[parent[s] + 2]++;
    for (i64 s = 0; s < ns; s++) F->childptr[s + 2] += F->childptr[s + 1];
    for (i64 s = 0; s < ns; s++) if (parent[s] >= 0) F->children[F->childptr[parent[s] + 1]++] = s;
    kvxs_partition(F);
    /* relative indices */
    for (i64 s = 0; s < ns; s++) {
        const i64 p = parent[s];
        if (p < 0) continue;
        i64 b = rowptr[p];
        for (i64 a = rowptr[s] + (super[s + 1] - super[s]); a < rowptr[s + 1]; a++) {
            while (rowidx[b] < rowidx[a]) b++;
            F->rel[a] = b - rowptr[p];
        }
    }
    /* scatter map of A's entries into the panels */
    i64 *iperm = zalloc(sizeof(i64) * n), *col2s = zalloc(sizeof(i64) * n);
    for (i64 i = 0; i < n; i++) iperm[perm[i]] = i;
    for (i64 s = 0; s < ns; s++) for (i64 j = super[s]; j < super[s + 1]; j++) col2s[j] = s;
    for (i64 j = 0; j < n; j++)
        for (i64 p = Ap[j]; p < Ap[j + 1]; p++) {
            i64 a = iperm[Ai[p]], b = iperm[j];
            i64 r = a > b ? a : b, c = a > b ? b : a;
            i64 s = col2s[c], m = rowptr[s + 1] - rowptr[s];
            i64 lo = rowptr[s], hi = rowptr[s + 1];
            while (lo < hi) { i64 mid = (lo + hi) / 2; if (rowidx[mid] < r) lo = mid + 1; else hi = mid; }
            F->amap[p] = F->px[s] + (lo - rowptr[s]) + (c - super[s]) * m;
        }
    free(iperm); free(col2s);
    return F;
}

/* at most 32: the OpenBLAS inside scipy (MAX_THREADS = 64) keeps a fixed table of work buffers that overflows when more threads call it at once */
void kvxs_set_threads(kvxs_factor *F, int nthreads) { F->nthreads = nthreads > 32 ? 32 : (nthreads > 0 ? nthreads : 1); kvxs_partition(F); }
i64 kvxs_minor(const kvxs_factor *F) { return F->minor; }

/* one front: assemble (own panel is already scattered), partial Cholesky, keep the update matrix for the parent */
static int front_factor(kvxs_factor *F, i64 s)
{
    const i64 k = F->super[s + 1] - F->super[s], m = F->rowptr[s + 1] - F->rowptr[s], u = m - k;
    double *P = F->Lx + F->px[s];
    double *U = NULL;
    if (u > 0) { U = calloc((size_t)(u * u), sizeof(double)); if (!U) return -2; }
    for (i64 q = F->childptr[s]; q < F->childptr[s + 1]; q++) {
        const i64 c = F->children[q];
        const i64 kc = F->super[c + 1] - F->super[c], uc = F->rowptr[c + 1] - F->rowptr[c] - kc;
        const i64 *rl = F->rel + F->rowptr[c] + kc;
        const double *Uc = F->U[c];
        for (i64 j = 0; j < uc; j++) {
            const i64 tc = rl[j];
            double *dst = tc < k ? P + tc * m : U + (tc - k) * u - k;
            const double *src = Uc + j * uc;
            for (i64 i = j; i < uc; i++) dst[rl[i]] += src[i];
        }
        free(F->U[c]); F->U[c] = NULL;
    }
    const int ik = (int)k, im = (int)m, iu = (int)u;
    int info = 0;
    scipy_dpotrf_("L", &ik, P, &im, &info);
    if (info != 0) { free(U); return info > 0 ? (int)(F->super[s] + info - 1) + 1 : -1; }   /* failing column + 1 */
    if (u > 0) {
        const double one = 1.0, mone = -1.0;
        scipy_dtrsm_("R", "L", "T", "N", &iu, &ik, &one, P, &im, P + k, &im);
        scipy_dsyrk_("L", "N", &iu, &ik, &mone, P + k, &im, &one, U, &iu);
    }
    F->U[s] = U;
    return 0;
}

/* A big front of the top of the tree on all threads: tiled right-looking Cholesky, every tile operation one single-threaded
 * BLAS call, the tiles of a phase spread over the OpenMP threads (OpenBLAS itself always runs single-threaded here: switching
 * its own thread pool on and off around OpenMP regions crashed inside the library). */
static int front_factor_parallel(kvxs_factor *F, i64 s)
{
    const i64 k = F->super[s + 1] - F->super[s], m = F->rowptr[s + 1] - F->rowptr[s], u = m - k;
    double *P = F->Lx + F->px[s];
    double *U = NULL;
    const int nt = F->nthreads;
    if (u > 0) { U = calloc((size_t)(u * u), sizeof(double)); if (!U) return -2; }
    for (i64 q = F->childptr[s]; q < F->childptr[s + 1]; q++) {
        const i64 c = F->children[q];
        const i64 kc = F->super[c + 1] - F->super[c], uc = F->rowptr[c + 1] - F->rowptr[c] - kc;
        const i64 *rl = F->rel + F->rowptr[c] + kc;
        const double *Uc = F->U[c];
#pragma omp parallel for schedule(dynamic, 16) num_threads(nt)
        for (i64 j = 0; j < uc; j++) {               /* distinct child columns land in distinct parent columns */
            const i64 tc = rl[j];
            double *dst = tc < k ? P + tc * m : U + (tc - k) * u - k;
            const double *src = Uc + j * uc;
            for (i64 i = j; i < uc; i++) dst[rl[i]] += src[i];
        }
        free(F->U[c]); F->U[c] = NULL;
    }
    const i64 NB = 256, RB = 512;
    const double one = 1.0, mone = -1.0;
    int fail = 0;
    typedef struct { i64 r0, r1, c0, c1; } tile_t;
    tile_t *tiles = malloc(sizeof(tile_t) * (size_t)(((m + NB - 1) / NB + 2) * ((m + RB - 1) / RB + 2)));
    if (!tiles) { free(U); return -2; }
    for (i64 jb = 0; jb < k && !fail; jb += NB) {
        const i64 nb = k - jb < NB ? k - jb : NB, t0 = jb + nb;
        const int inb = (int)nb, im = (int)m, iu = (int)u;
        int info = 0;
        scipy_dpotrf_("L", &inb, P + jb + jb * m, &im, &info);
        if (info != 0) { fail = info > 0 ? (int)(F->super[s] + jb + info - 1) + 1 : -1; break; }
        const i64 nrb = (m - t0 + RB - 1) / RB;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
        for (i64 b = 0; b < nrb; b++) {
            const i64 r0 = t0 + b * RB;
            const int rows = (int)(m - r0 < RB ? m - r0 : RB);
            scipy_dtrsm_("R", "L", "T", "N", &rows, &inb, &one, P + jb + jb * m, &im, P + r0 + jb * m, &im);
        }
        /* trailing update, by column blocks that do not straddle the panel / update-matrix boundary, cut into row chunks */
        i64 ntile = 0;
        for (i64 c0 = t0; c0 < m; ) {
            const i64 lim = c0 < k ? k : m;
            const i64 c1 = c0 + NB < lim ? c0 + NB : lim;
            for (i64 r0 = c0; r0 < m; r0 += RB) { tiles[ntile].r0 = r0; tiles[ntile].r1 = r0 + RB < m ? r0 + RB : m; tiles[ntile].c0 = c0; tiles[ntile].c1 = c1; ntile++; }
            c0 = c1;
        }
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
        for (i64 t = 0; t < ntile; t++) {
            const tile_t T = tiles[t];
            const int nc = (int)(T.c1 - T.c0);
            double *C; int ldc;
            if (T.c0 < k) { C = P + T.c0 * m; ldc = im; } else { C = U + (T.c0 - k) * u - k; ldc = iu; }   /* C[row] = column T.c0, global row index */
            i64 r0 = T.r0;
            if (r0 == T.c0) {                        /* the diagonal tile of the column block: symmetric update of its lower part */
                scipy_dsyrk_("L", "N", &nc, &inb, &mone, P + T.c0 + jb * m, &im, &one, C + T.c0, &ldc);
                r0 = T.c1;
            }
            const int rows = (int)(T.r1 - r0);
            if (rows > 0)
                scipy_dgemm_("N", "T", &rows, &nc, &inb, &mone, P + r0 + jb * m, &im, P + T.c0 + jb * m, &im, &one, C + r0, &ldc);
        }
    }
    free(tiles);
    if (fail) { free(U); return fail; }
    F->U[s] = U;
    return 0;
}

/* numeric factorisation; returns 0, or 1 when not positive definite (kvxs_minor = failing column), or -1 out of memory */
int kvxs_factorize(kvxs_factor *F, const double *Ax)
{
    const i64 ns = F->ns;
    const int tim = getenv("KVXS_TIMING") != NULL;
    double t0 = omp_get_wtime();
    {
        const i64 tot = F->px[ns];
#pragma omp parallel for schedule(static) num_threads(F->nthreads)
        for (i64 c = 0; c < (tot + 65535) / 65536; c++)
            memset(F->Lx + c * 65536, 0, sizeof(double) * (size_t)((c + 1) * 65536 <= tot ? 65536 : tot - c * 65536));
#pragma omp parallel for schedule(static) num_threads(F->nthreads)
        for (i64 p = 0; p < F->nnz; p++) F->Lx[F->amap[p]] = Ax[p];
    }
    double t1 = omp_get_wtime();
    for (i64 s = 0; s < ns; s++) { free(F->U[s]); F->U[s] = NULL; }
    F->minor = F->n;
    int bad = 0;
    i64 minor = F->n;
    /* the independent subtrees: one thread each, BLAS single-threaded */
    scipy_openblas_set_num_threads(1);
#pragma omp parallel for schedule(dynamic, 1) num_threads(F->nthreads)
    for (i64 t = 0; t < F->nsub; t++) {
        for (i64 q = F->subptr[t]; q < F->subptr[t + 1]; q++) {
            int rc = front_factor(F, F->sublist[q]);
            if (rc != 0) {
#pragma omp critical
                { if (rc < 0) bad = -1; else { if (bad >= 0) bad = 1; if (rc - 1 < minor) minor = rc - 1; } }
                break;                       /* (a failed front leaves garbage for its ancestors: stop like CHOLMOD does) */
            }
        }
    }
    double t2 = omp_get_wtime();
    /* the top of the tree: front after front, each on all threads */
    for (i64 q = 0; q < F->ntop && bad == 0; q++) {
        int rc = front_factor_parallel(F, F->toplist[q]);
        if (rc < 0) bad = -1;
        else if (rc > 0) { bad = 1; if (rc - 1 < minor) minor = rc - 1; }
    }
    if (tim) fprintf(stderr, "kvxs: scatter %.3f s, %lld subtrees %.3f s, %lld top fronts %.3f s\n", t1 - t0, (long long)F->nsub, t2 - t1,
                     (long long)F->ntop, omp_get_wtime() - t2);
    for (i64 s = 0; s < ns; s++) { free(F->U[s]); F->U[s] = NULL; }
    if (bad < 0) return -1;
    if (bad > 0) { F->minor = minor; return 1; }
    return 0;
}

/* forward / backward step of one front on the permuted vector x; w: m doubles of scratch.  Fronts of up to 64 pivot columns
 * are swept column by column in place (one contiguous pass over the panel: a BLAS call costs more than the arithmetic of a
 * front of a dozen columns, and there are tens of thousands of them); larger ones go to dtrsv / dgemv.
 * lim: rows below lim belong to the calling thread's own subtree and are updated plainly, the others atomically. */
static void front_fwd(const kvxs_factor *F, i64 s, double *x, double *w, i64 lim)
{
    const i64 k = F->super[s + 1] - F->super[s], m = F->rowptr[s + 1] - F->rowptr[s], u = m - k, f = F->super[s];
    const double *P = F->Lx + F->px[s];
    const i64 *rows = F->rowidx + F->rowptr[s] + k;
    if (k <= 64) {
        for (i64 i = 0; i < u; i++) w[i] = 0.0;
        for (i64 j = 0; j < k; j++) {
            const double *c = P + j * m;
            const double xj = x[f + j] / c[j];
            x[f + j] = xj;
            for (i64 i = j + 1; i < k; i++) x[f + i] -= c[i] * xj;
            for (i64 i = 0; i < u; i++) w[i] += c[k + i] * xj;
        }
    } else {
        const int one = 1, ik = (int)k, im = (int)m, iu = (int)u;
        const double d1 = 1.0, d0 = 0.0;
        scipy_dtrsv_("L", "N", "N", &ik, P, &im, x + f, &one);
        if (u > 0) scipy_dgemv_("N", &iu, &ik, &d1, P + k, &im, x + f, &one, &d0, w, &one);
    }
    for (i64 i = 0; i < u; i++) {
        const i64 r = rows[i];
        if (r < lim) x[r] -= w[i];
        else {
#pragma omp atomic
            x[r] -= w[i];
        }
    }
}

static void front_bwd(const kvxs_factor *F, i64 s, double *x, double *w)
{
    const i64 k = F->super[s + 1] - F->super[s], m = F->rowptr[s + 1] - F->rowptr[s], u = m - k, f = F->super[s];
    const double *P = F->Lx + F->px[s];
    const i64 *rows = F->rowidx + F->rowptr[s] + k;
    for (i64 i = 0; i < u; i++) w[i] = x[rows[i]];
    if (k <= 64) {
        for (i64 j = k - 1; j >= 0; j--) {
            const double *c = P + j * m;
            double t = x[f + j];
            for (i64 i = j + 1; i < k; i++) t -= c[i] * x[f + i];
            for (i64 i = 0; i < u; i++) t -= c[k + i] * w[i];
            x[f + j] = t / c[j];
        }
    } else {
        const int one = 1, ik = (int)k, im = (int)m, iu = (int)u;
        const double d1 = 1.0, dm1 = -1.0;
        if (u > 0) scipy_dgemv_("T", &iu, &ik, &dm1, P + k, &im, w, &one, &d1, x + f, &one);
        scipy_dtrsv_("L", "T", "N", &ik, P, &im, x + f, &one);
    }
}

/* A x = b for nrhs columns (ld = ldB), in place; x in the caller's (unpermuted) order.  Subtrees in parallel (their fronts
 * update ancestor rows above the subtree atomically), then the top of the tree in sequence; the backward sweep mirrors it. */
int kvxs_solve(const kvxs_factor *F, double *B, i64 nrhs, i64 ldB)
{
    if (F->minor < F->n) return 1;
    const i64 n = F->n, ns = F->ns;
    double *x = malloc(sizeof(double) * (n ? n : 1));
    i64 maxm = 1;
    for (i64 s = 0; s < ns; s++) if (F->rowptr[s + 1] - F->rowptr[s] > maxm) maxm = F->rowptr[s + 1] - F->rowptr[s];
    const int nt = F->nthreads;
    double *wall = malloc(sizeof(double) * (size_t)maxm * (size_t)(nt + 1));
    if (!x || !wall) { free(x); free(wall); return -1; }
    scipy_openblas_set_num_threads(1);
    for (i64 r = 0; r < nrhs; r++) {
        double *b = B + r * ldB;
#pragma omp parallel for schedule(static) num_threads(nt)
        for (i64 i = 0; i < n; i++) x[i] = b[F->perm[i]];
#pragma omp parallel num_threads(nt)
        {
            double *wl = wall + (size_t)maxm * (size_t)omp_get_thread_num();
#pragma omp for schedule(dynamic, 1)
            for (i64 t = 0; t < F->nsub; t++) {
                /* (the fronts of a subtree are a contiguous postorder range: its columns end where its root's do) */
                const i64 lim = F->super[F->sublist[F->subptr[t + 1] - 1] + 1];
                for (i64 q = F->subptr[t]; q < F->subptr[t + 1]; q++) front_fwd(F, F->sublist[q], x, wl, lim);
            }
        }
        double *w = wall + (size_t)maxm * (size_t)nt;
        for (i64 q = 0; q < F->ntop; q++) front_fwd(F, F->toplist[q], x, w, n);
        for (i64 q = F->ntop - 1; q >= 0; q--) front_bwd(F, F->toplist[q], x, w);
#pragma omp parallel num_threads(nt)
        {
            double *wl = wall + (size_t)maxm * (size_t)omp_get_thread_num();
#pragma omp for schedule(dynamic, 1)
            for (i64 t = 0; t < F->nsub; t++)
                for (i64 q = F->subptr[t + 1] - 1; q >= F->subptr[t]; q--) front_bwd(F, F->sublist[q], x, wl);
        }
#pragma omp parallel for schedule(static) num_threads(nt)
        for (i64 i = 0; i < n; i++) b[F->perm[i]] = x[i];
    }
    free(x); free(wall);
    return 0;
}
