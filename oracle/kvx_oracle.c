/*
 * kvx_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU checker; never on the product path).
 *
 * Plain-C restatement of the arithmetic on the KKT factor/solve hot path of
 * sanurielf/kvxopt, used by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg to check the HIP path.  Nothing in kvxopt_amd/ links or calls it.
 *
 * What it restates, and from where:
 *  - sparse Cholesky P*A*P' = L*L' + triangular solves: the reference delegates this
 *    to SuiteSparse CHOLMOD 7.8.2 (un-vendored third-party; call sites
 *    /root/reference/src/C/cholmod.c:274 analyze_p, :362 factorize, :483 solve).
 *    CHOLMOD is absent from this image, so the published algorithm is restated:
 *    elimination tree (Liu 1990), row-subtree reach + up-looking Cholesky
 *    (Davis, "Direct Methods for Sparse Linear Systems", SIAM 2006, ch. 4).
 *    The factor L of P*A*P' is unique for a given permutation, so any correct
 *    Cholesky code is a valid oracle for values; PARITY PINNING: the doc known
 *    answers /root/reference/doc/source/spsolvers.rst:555-563, 580-585, 700-708,
 *    759-772 (tests/test_oracle.py) + dense numpy.linalg.cholesky cross-checks.
 *    No CHOLMOD-produced vectors exist in the reference's tests ("parity unpinned"
 *    for third-party arithmetic beyond those doc answers).
 *  - sys codes 0..8 and the P / P' convention: cholmod.c:437-439, SURVEY 8(a) a13.
 *  - triangle packing (only `uplo` triangle read): cholmod.c:132-181.
 *  - S = G' diag(di^2) G on a fixed pattern: misc.py:1418-1419,1451 ->
 *    sparse.c:1260-1283 (row scale) and sparse.c:2176-2198 (partial syrk).
 *  - y := alpha*op(A)*x + beta*y on CCS: sparse.c:1073-1104.
 *
 * All indices int64 (int_t = Py_ssize_t, kvxopt.h:46), all values double.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef int64_t i64;

typedef struct {
    i64 n;
    i64 *perm;    /* perm[k] = original index of permuted row/col k        */
    i64 *iperm;
    i64 *Cp, *Ci; /* upper triangle of C = P A P' (pattern), CSC, sorted   */
    i64 *Cmap;    /* for each C entry: index into the caller's A values    */
    i64 *parent;  /* elimination tree                                       */
    i64 *Lp, *Li; /* column pointers / row indices of L (sorted)            */
    double *Lx;
    i64 minor;    /* n if factor OK, else failing column (permuted index)   */
    int numeric;  /* 0 symbolic only, 1 numeric present                      */
} kvxo_factor;

static void *xmalloc(size_t s) { void *p = malloc(s ? s : 1); return p; }

void kvxo_chol_free(kvxo_factor *F)
{
    if (!F) return;
    free(F->perm); free(F->iperm); free(F->Cp); free(F->Ci); free(F->Cmap);
    free(F->parent); free(F->Lp); free(F->Li); free(F->Lx); free(F);
}

/* row-subtree reach: pattern of row k of L = nodes reached in the etree from the
 * off-diagonal entries of column k of upper(C), stopping at marked nodes.
 * Output s[top..n-1] in topological order. */
static i64 ereach(const i64 *Cp, const i64 *Ci, i64 k, const i64 *parent,
                  i64 *s, i64 *w, i64 n)
{
    i64 top = n, p, i, len;
    w[k] = k;
    for (p = Cp[k]; p < Cp[k + 1]; p++) {
        i = Ci[p];
        if (i > k) continue;
        for (len = 0; w[i] != k; i = parent[i]) { s[len++] = i; w[i] = k; }
        while (len > 0) s[--top] = s[--len];
    }
    return top;
}

/* analyze: uplo = 'L' or 'U' selects the triangle of A that is read
 * (cholmod.c:132-181).  perm may be NULL (identity). Returns NULL on bad input. */
kvxo_factor *kvxo_chol_analyze(i64 n, const i64 *Ap, const i64 *Ai, int uplo,
                               const i64 *perm)
{
    kvxo_factor *F = (kvxo_factor *)calloc(1, sizeof(kvxo_factor));
    i64 j, p, k;
    if (!F) return NULL;
    F->n = n; F->minor = n;
    F->perm = (i64 *)xmalloc(sizeof(i64) * n);
    F->iperm = (i64 *)xmalloc(sizeof(i64) * n);
    for (k = 0; k < n; k++) F->iperm[k] = -1;
    for (k = 0; k < n; k++) {
        i64 q = perm ? perm[k] : k;
        if (q < 0 || q >= n || F->iperm[q] != -1) { kvxo_chol_free(F); return NULL; }
        F->perm[k] = q; F->iperm[q] = k;
    }
    /* count entries of upper(C) per column */
    F->Cp = (i64 *)calloc(n + 1, sizeof(i64));
    i64 *cnt = (i64 *)calloc(n + 1, sizeof(i64));
    for (j = 0; j < n; j++)
        for (p = Ap[j]; p < Ap[j + 1]; p++) {
            i64 i = Ai[p];
            if ((uplo == 'L' && i < j) || (uplo != 'L' && i > j)) continue;
            i64 i2 = F->iperm[i], j2 = F->iperm[j];
            cnt[i2 > j2 ? i2 : j2]++;
        }
    for (j = 0; j < n; j++) F->Cp[j + 1] = F->Cp[j] + cnt[j];
    i64 cnz = F->Cp[n];
    F->Ci = (i64 *)xmalloc(sizeof(i64) * cnz);
    F->Cmap = (i64 *)xmalloc(sizeof(i64) * cnz);
    for (j = 0; j < n; j++) cnt[j] = F->Cp[j];
    for (j = 0; j < n; j++)
        for (p = Ap[j]; p < Ap[j + 1]; p++) {
            i64 i = Ai[p];
            if ((uplo == 'L' && i < j) || (uplo != 'L' && i > j)) continue;
            i64 i2 = F->iperm[i], j2 = F->iperm[j];
            i64 c = i2 > j2 ? i2 : j2, r = i2 > j2 ? j2 : i2;
            i64 q = cnt[c]++;
            F->Ci[q] = r; F->Cmap[q] = p;
        }
    /* sort rows within each column (insertion sort: columns are short) */
    for (j = 0; j < n; j++)
        for (p = F->Cp[j] + 1; p < F->Cp[j + 1]; p++) {
            i64 r = F->Ci[p], m = F->Cmap[p], q = p;
            while (q > F->Cp[j] && F->Ci[q - 1] > r) {
                F->Ci[q] = F->Ci[q - 1]; F->Cmap[q] = F->Cmap[q - 1]; q--;
            }
            F->Ci[q] = r; F->Cmap[q] = m;
        }
    free(cnt);
    /* elimination tree (Liu), with path compression through `anc` */
    F->parent = (i64 *)xmalloc(sizeof(i64) * n);
    i64 *anc = (i64 *)xmalloc(sizeof(i64) * n);
    for (k = 0; k < n; k++) {
        F->parent[k] = -1; anc[k] = -1;
        for (p = F->Cp[k]; p < F->Cp[k + 1]; p++) {
            i64 i = F->Ci[p], inext;
            for (; i != -1 && i < k; i = inext) {
                inext = anc[i]; anc[i] = k;
                if (inext == -1) F->parent[i] = k;
            }
        }
    }
    free(anc);
    /* column counts by row-subtree traversal (O(|L|), simple and obviously right) */
    i64 *s = (i64 *)xmalloc(sizeof(i64) * n), *w = (i64 *)xmalloc(sizeof(i64) * n);
    i64 *cc = (i64 *)xmalloc(sizeof(i64) * n);
    for (k = 0; k < n; k++) { w[k] = -1; cc[k] = 1; }
    for (k = 0; k < n; k++) {
        i64 top = ereach(F->Cp, F->Ci, k, F->parent, s, w, n);
        for (; top < n; top++) cc[s[top]]++;
    }
    F->Lp = (i64 *)xmalloc(sizeof(i64) * (n + 1));
    F->Lp[0] = 0;
    for (k = 0; k < n; k++) F->Lp[k + 1] = F->Lp[k] + cc[k];
    free(s); free(w); free(cc);
    F->Li = (i64 *)xmalloc(sizeof(i64) * F->Lp[n]);
    F->Lx = (double *)xmalloc(sizeof(double) * F->Lp[n]);
    return F;
}

i64 kvxo_chol_n(const kvxo_factor *F) { return F->n; }
i64 kvxo_chol_lnz(const kvxo_factor *F) { return F->Lp[F->n]; }
i64 kvxo_chol_minor(const kvxo_factor *F) { return F->minor; }
void kvxo_chol_get_perm(const kvxo_factor *F, i64 *perm) { memcpy(perm, F->perm, sizeof(i64) * F->n); }
void kvxo_chol_get_parent(const kvxo_factor *F, i64 *parent) { memcpy(parent, F->parent, sizeof(i64) * F->n); }
/* sum_j c_j^2 with c_j = nnz(L(:,j)) incl. diagonal: SURVEY 8(d) flop measure */
double kvxo_chol_flops(const kvxo_factor *F)
{
    double f = 0; i64 j;
    for (j = 0; j < F->n; j++) { double c = (double)(F->Lp[j + 1] - F->Lp[j]); f += c * c; }
    return f;
}
void kvxo_chol_get_L(const kvxo_factor *F, i64 *Lp, i64 *Li, double *Lx)
{
    memcpy(Lp, F->Lp, sizeof(i64) * (F->n + 1));
    memcpy(Li, F->Li, sizeof(i64) * F->Lp[F->n]);
    if (Lx) memcpy(Lx, F->Lx, sizeof(double) * F->Lp[F->n]);
}

/* numeric up-looking Cholesky.  Ax are the caller's values for the pattern given
 * to analyze.  Returns 0 OK, 1 not positive definite (minor set; CHOLMOD_NOT_POSDEF
 * semantics documented at cholmod.c:308-310, 376-379). */
int kvxo_chol_factorize(kvxo_factor *F, const double *Ax)
{
    i64 n = F->n, k, p, top;
    i64 *c = (i64 *)xmalloc(sizeof(i64) * n), *s = (i64 *)xmalloc(sizeof(i64) * n);
    i64 *w = (i64 *)xmalloc(sizeof(i64) * n);
    double *x = (double *)xmalloc(sizeof(double) * n);
    for (k = 0; k < n; k++) { c[k] = F->Lp[k]; w[k] = -1; x[k] = 0.0; }
    F->minor = n; F->numeric = 1;
    for (k = 0; k < n; k++) {
        top = ereach(F->Cp, F->Ci, k, F->parent, s, w, n);
        double d = 0.0;
        for (p = F->Cp[k]; p < F->Cp[k + 1]; p++) {
            i64 i = F->Ci[p];
            if (i < k) x[i] += Ax[F->Cmap[p]];
            else if (i == k) d += Ax[F->Cmap[p]];
        }
        for (; top < n; top++) {
            i64 i = s[top];
            double lki = x[i] / F->Lx[F->Lp[i]];
            x[i] = 0.0;
            for (p = F->Lp[i] + 1; p < c[i]; p++) x[F->Li[p]] -= F->Lx[p] * lki;
            d -= lki * lki;
            p = c[i]++;
            F->Li[p] = k; F->Lx[p] = lki;
        }
        if (!(d > 0.0)) {   /* also catches NaN */
            F->minor = k;
            free(c); free(s); free(w); free(x);
            return 1;
        }
        p = c[k]++;
        F->Li[p] = k; F->Lx[p] = sqrt(d);
    }
    free(c); free(s); free(w); free(x);
    return 0;
}

static void lsolve(const kvxo_factor *F, double *x)
{
    i64 j, p;
    for (j = 0; j < F->n; j++) {
        x[j] /= F->Lx[F->Lp[j]];
        for (p = F->Lp[j] + 1; p < F->Lp[j + 1]; p++) x[F->Li[p]] -= F->Lx[p] * x[j];
    }
}
static void ltsolve(const kvxo_factor *F, double *x)
{
    i64 j, p;
    for (j = F->n - 1; j >= 0; j--) {
        for (p = F->Lp[j] + 1; p < F->Lp[j + 1]; p++) x[j] -= F->Lx[p] * x[F->Li[p]];
        x[j] /= F->Lx[F->Lp[j]];
    }
}

/* sys: 0 A, 1 LDL', 2 LD, 3 DL', 4 L, 5 L', 6 D, 7 P, 8 P'  (cholmod.c:437-439)
 * with D = I for an LL' factor.  sys 7: x[k] = b[perm[k]];  sys 8: x[perm[k]] = b[k]
 * (SURVEY 8(a) a13).  B is n x nrhs, column-major, leading dimension ldB, in place.
 * Returns 0 OK, 1 singular factor, 2 symbolic-only factor, 3 bad sys. */
int kvxo_chol_solve(const kvxo_factor *F, int sys, double *B, i64 nrhs, i64 ldB)
{
    i64 n = F->n, k, r;
    if (!F->numeric) return 2;
    if (F->minor < n) return 1;
    if (sys < 0 || sys > 8) return 3;
    if (n == 0) return 0;
    double *t = (double *)xmalloc(sizeof(double) * n);
    for (r = 0; r < nrhs; r++) {
        double *b = B + r * ldB;
        switch (sys) {
        case 0:
            for (k = 0; k < n; k++) t[k] = b[F->perm[k]];
            lsolve(F, t); ltsolve(F, t);
            for (k = 0; k < n; k++) b[F->perm[k]] = t[k];
            break;
        case 1: lsolve(F, b); ltsolve(F, b); break;
        case 2: case 4: lsolve(F, b); break;
        case 3: case 5: ltsolve(F, b); break;
        case 6: break;
        case 7:
            for (k = 0; k < n; k++) t[k] = b[F->perm[k]];
            memcpy(b, t, sizeof(double) * n); break;
        case 8:
            for (k = 0; k < n; k++) t[F->perm[k]] = b[k];
            memcpy(b, t, sizeof(double) * n); break;
        }
    }
    free(t);
    return 0;
}

/* diag(L) in permuted order (cholmod.c:900-945 reads it panel by panel) */
void kvxo_chol_diag(const kvxo_factor *F, double *d)
{
    i64 j;
    for (j = 0; j < F->n; j++) d[j] = F->Lx[F->Lp[j]];
}

/* ---------------------------------------------------------------------------
 * y := alpha * op(A) * x + beta * y for CCS A (m x n).   sparse.c:1073-1104
 * trans = 'N' or 'T'.
 */
void kvxo_spmv(int trans, i64 m, i64 n, const i64 *Ap, const i64 *Ai, const double *Ax,
               double alpha, const double *x, double beta, double *y)
{
    i64 j, p, ly = (trans == 'N') ? m : n;
    for (j = 0; j < ly; j++) y[j] = (beta == 0.0) ? 0.0 : beta * y[j];
    if (trans == 'N') {
        for (j = 0; j < n; j++)
            for (p = Ap[j]; p < Ap[j + 1]; p++) y[Ai[p]] += alpha * Ax[p] * x[j];
    } else {
        for (j = 0; j < n; j++) {
            double s = 0.0;
            for (p = Ap[j]; p < Ap[j + 1]; p++) s += Ax[p] * x[Ai[p]];
            y[j] += alpha * s;
        }
    }
}

/* Gs = diag(di) * G on G's pattern     (misc.py:1418-1419 -> sparse.c:1260-1283) */
void kvxo_rowscale(i64 n, const i64 *Gp, const i64 *Gi, const double *Gx,
                   const double *di, double *Gsx)
{
    i64 j, p;
    for (j = 0; j < n; j++)
        for (p = Gp[j]; p < Gp[j + 1]; p++) Gsx[p] = di[Gi[p]] * Gx[p];
}

/* S(i,j) = <Gs(:,i), Gs(:,j)> for every stored S(i,j), i >= j, on S's fixed
 * pattern (misc.py:1451 -> sparse.c:2176-2198: scatter column i, gather over j).
 * Gs is ml x n CCS with sorted rows; S is n x n lower CCS. `work` has ml doubles. */
void kvxo_syrk_partial(i64 ml, i64 n, const i64 *Gp, const i64 *Gi, const double *Gsx,
                       const i64 *Sp, const i64 *Si, double *Sx)
{
    double *work = (double *)calloc(ml > 0 ? ml : 1, sizeof(double));
    i64 j, p, q;
    for (j = 0; j < n; j++) {
        for (p = Gp[j]; p < Gp[j + 1]; p++) work[Gi[p]] = Gsx[p];
        for (q = Sp[j]; q < Sp[j + 1]; q++) {
            i64 i = Si[q];
            double s = 0.0;
            for (p = Gp[i]; p < Gp[i + 1]; p++) s += Gsx[p] * work[Gi[p]];
            Sx[q] = s;
        }
        for (p = Gp[j]; p < Gp[j + 1]; p++) work[Gi[p]] = 0.0;
    }
    free(work);
}
