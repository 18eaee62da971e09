/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the sparse LU that the reference's kvxopt.klu module obtains from
 * SuiteSparse KLU (third-party, absent from /root/reference and from this image; pinned 7.8.2 at
 * .ci/config/versions.env:7-8).  Reference call sites: src/C/klu.c:141 (klu_analyze), :161 (klu_factor),
 * :187-198 (klu_solve / klu_tsolve), :444-449 (klu_extract), :760-822 (determinant from Udiag, Rs, Pnum, Q).
 *
 * What is restated is KLU's PUBLISHED algorithm (T. A. Davis, E. Palamadai Natarajan, "Algorithm 907: KLU, a direct
 * sparse solver for circuit simulation problems", ACM TOMS 37(3), 2010): row scaling by the largest magnitude of
 * each row (Common.scale = 2, the default), then a left-looking column LU (Gilbert-Peierls) with threshold partial
 * pivoting and diagonal preference (Common.tol = 0.001): column k of L and U comes from a sparse triangular solve
 * with the columns already computed; the diagonal entry is kept as pivot when |d| >= tol * max|candidates|.
 * Not restated: the block-triangular permutation and the AMD ordering inside blocks (they change the order of
 * elimination and fill, not the solution); the column order is the caller's Q (natural when NULL).
 *
 * PARITY UNPINNED against the KLU binary (no SuiteSparse here).  Pinned instead on the reference's own known answers
 * (doc/source/spsolvers.rst:333-345, 420-439) and on the identities its tests assert
 * (tests/test_sparse_solvers.py:214-323: R P A Q = L U + F, A x = b, A' x = b, determinant) -- tests/test_oracle.py.
 * The triangular solve visits the earlier pivots in order (O(n) per column) instead of a depth-first reach: same
 * arithmetic in the same order, simpler code; fine for the sizes the tests use (n <= 4000). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;

typedef struct {
    i64 n;
    i64 *Lp, *Li, *Up, *Ui;      /* L by column: (original row, multiplier), unit diagonal implied; U by column: (pivot index < k, value) */
    double *Lx, *Ux, *Udiag, *Rs;
    i64 *P, *pinv, *Q;           /* P[k] = original row of pivot k; Q[k] = original column eliminated at step k */
    i64 lcap, ucap;
    int singular;
} kvxo_klu;

void kvxo_klu_free(kvxo_klu *F)
{
    if (!F) return;
    free(F->Lp); free(F->Li); free(F->Up); free(F->Ui); free(F->Lx); free(F->Ux); free(F->Udiag); free(F->Rs);
    free(F->P); free(F->pinv); free(F->Q);
    free(F);
}

static int grow(i64 **idx, double **val, i64 *cap, i64 need)
{
    if (need <= *cap) return 1;
    i64 nc = *cap * 2 > need ? *cap * 2 : need;
    i64 *ni = (i64 *)realloc(*idx, (size_t)nc * sizeof(i64));
    if (!ni) return 0;
    *idx = ni;
    double *nv = (double *)realloc(*val, (size_t)nc * sizeof(double));
    if (!nv) return 0;
    *val = nv;
    *cap = nc;
    return 1;
}

kvxo_klu *kvxo_klu_factor(i64 n, const i64 *Ap, const i64 *Ai, const double *Ax, const i64 *Q, double tol)
{
    kvxo_klu *F = (kvxo_klu *)calloc(1, sizeof(kvxo_klu));
    if (!F) return NULL;
    F->n = n;
    F->lcap = F->ucap = Ap[n] + n + 16;
    F->Lp = (i64 *)calloc((size_t)n + 1, sizeof(i64));
    F->Up = (i64 *)calloc((size_t)n + 1, sizeof(i64));
    F->Li = (i64 *)malloc((size_t)F->lcap * sizeof(i64));
    F->Lx = (double *)malloc((size_t)F->lcap * sizeof(double));
    F->Ui = (i64 *)malloc((size_t)F->ucap * sizeof(i64));
    F->Ux = (double *)malloc((size_t)F->ucap * sizeof(double));
    F->Udiag = (double *)calloc((size_t)n + 1, sizeof(double));
    F->Rs = (double *)calloc((size_t)n + 1, sizeof(double));
    F->P = (i64 *)malloc(((size_t)n + 1) * sizeof(i64));
    F->pinv = (i64 *)malloc(((size_t)n + 1) * sizeof(i64));
    F->Q = (i64 *)malloc(((size_t)n + 1) * sizeof(i64));
    double *x = (double *)calloc((size_t)n + 1, sizeof(double));
    if (!F->Lp || !F->Up || !F->Li || !F->Lx || !F->Ui || !F->Ux || !F->Udiag || !F->Rs || !F->P || !F->pinv || !F->Q || !x) {
        free(x); kvxo_klu_free(F); return NULL;
    }
    for (i64 i = 0; i < n; i++) { F->pinv[i] = -1; F->Q[i] = Q ? Q[i] : i; }
    /* klu_scale, scale = 2: Rs[i] = max_j |a_ij|; a zero row is singular */
    for (i64 p = 0; p < Ap[n]; p++) { double a = fabs(Ax[p]); if (a > F->Rs[Ai[p]]) F->Rs[Ai[p]] = a; }
    for (i64 i = 0; i < n; i++) if (!(F->Rs[i] > 0.0)) { F->singular = 1; F->Rs[i] = 1.0; }
    i64 lnz = 0, unz = 0;
    for (i64 k = 0; k < n; k++) {
        const i64 j = F->Q[k];
        for (i64 p = Ap[j]; p < Ap[j + 1]; p++) x[Ai[p]] += Ax[p] / F->Rs[Ai[p]];
        /* x := L(:, 0:k) \ x, earlier pivots in order */
        F->Up[k] = unz;
        for (i64 t = 0; t < k; t++) {
            const i64 r = F->P[t];
            const double u = x[r];
            if (u == 0.0) continue;
            if (!grow(&F->Ui, &F->Ux, &F->ucap, unz + 1)) { free(x); kvxo_klu_free(F); return NULL; }
            F->Ui[unz] = t; F->Ux[unz] = u; unz++;
            x[r] = 0.0;
            for (i64 p = F->Lp[t]; p < F->Lp[t + 1]; p++) x[F->Li[p]] -= F->Lx[p] * u;
        }
        /* threshold partial pivoting with diagonal preference */
        double amax = 0.0; i64 imax = -1;
        for (i64 i = 0; i < n; i++)
            if (F->pinv[i] < 0) { double a = fabs(x[i]); if (a > amax) { amax = a; imax = i; } }
        i64 piv = imax;
        if (imax >= 0 && F->pinv[j] < 0 && fabs(x[j]) >= tol * amax && x[j] != 0.0) piv = j;
        if (piv < 0 || !(amax > 0.0)) {
            F->singular = 1;
            for (i64 i = 0; i < n; i++) if (F->pinv[i] < 0) { piv = i; break; }
            x[piv] = 1.0;
        }
        const double d = x[piv];
        F->Udiag[k] = d; F->P[k] = piv; F->pinv[piv] = k;
        x[piv] = 0.0;
        F->Lp[k] = lnz;
        for (i64 i = 0; i < n; i++)
            if (F->pinv[i] < 0 && x[i] != 0.0) {
                if (!grow(&F->Li, &F->Lx, &F->lcap, lnz + 1)) { free(x); kvxo_klu_free(F); return NULL; }
                F->Li[lnz] = i; F->Lx[lnz] = x[i] / d; lnz++;
                x[i] = 0.0;
            }
        F->Lp[k + 1] = lnz;
        F->Up[k + 1] = unz;
    }
    free(x);
    return F;
}

int kvxo_klu_singular(const kvxo_klu *F) { return F->singular; }
i64 kvxo_klu_lnz(const kvxo_klu *F) { return F->Lp[F->n]; }
i64 kvxo_klu_unz(const kvxo_klu *F) { return F->Up[F->n]; }

/* klu_solve (trans = 0) / klu_tsolve (trans = 1), B is n x nrhs with leading dimension ldB, overwritten */
int kvxo_klu_solve(const kvxo_klu *F, int trans, double *B, i64 nrhs, i64 ldB)
{
    const i64 n = F->n;
    if (F->singular) return 1;
    double *y = (double *)malloc(((size_t)n + 1) * sizeof(double));
    if (!y) return 2;
    for (i64 c = 0; c < nrhs; c++) {
        double *b = B + c * ldB;
        if (!trans) {
            for (i64 k = 0; k < n; k++) y[k] = b[F->P[k]] / F->Rs[F->P[k]];
            for (i64 k = 0; k < n; k++)
                for (i64 p = F->Lp[k]; p < F->Lp[k + 1]; p++) y[F->pinv[F->Li[p]]] -= F->Lx[p] * y[k];
            for (i64 k = n - 1; k >= 0; k--) {
                y[k] /= F->Udiag[k];
                for (i64 p = F->Up[k]; p < F->Up[k + 1]; p++) y[F->Ui[p]] -= F->Ux[p] * y[k];
            }
            for (i64 k = 0; k < n; k++) b[F->Q[k]] = y[k];
        } else {
            for (i64 k = 0; k < n; k++) y[k] = b[F->Q[k]];
            for (i64 k = 0; k < n; k++) {
                for (i64 p = F->Up[k]; p < F->Up[k + 1]; p++) y[k] -= F->Ux[p] * y[F->Ui[p]];
                y[k] /= F->Udiag[k];
            }
            for (i64 k = n - 1; k >= 0; k--)
                for (i64 p = F->Lp[k]; p < F->Lp[k + 1]; p++) y[k] -= F->Lx[p] * y[F->pinv[F->Li[p]]];
            for (i64 k = 0; k < n; k++) b[F->P[k]] = y[k] / F->Rs[F->P[k]];
        }
    }
    free(y);
    return 0;
}

/* klu_extract: L (unit diagonal stored), U as CCS in pivotal coordinates; P, Q, Rs (scale of pivot row k) */
void kvxo_klu_extract(const kvxo_klu *F, i64 *Lp, i64 *Li, double *Lx, i64 *Up, i64 *Ui, double *Ux, i64 *P, i64 *Q, double *Rs)
{
    const i64 n = F->n;
    i64 l = 0, u = 0;
    for (i64 k = 0; k < n; k++) {
        Lp[k] = l;
        Li[l] = k; Lx[l] = 1.0; l++;
        for (i64 p = F->Lp[k]; p < F->Lp[k + 1]; p++) { Li[l] = F->pinv[F->Li[p]]; Lx[l] = F->Lx[p]; l++; }
        Up[k] = u;
        for (i64 p = F->Up[k]; p < F->Up[k + 1]; p++) { Ui[u] = F->Ui[p]; Ux[u] = F->Ux[p]; u++; }
        Ui[u] = k; Ux[u] = F->Udiag[k]; u++;
        P[k] = F->P[k]; Q[k] = F->Q[k]; Rs[k] = F->Rs[F->P[k]];
    }
    Lp[n] = l; Up[n] = u;
}

/* klu.c:760-822 */
double kvxo_klu_det(const kvxo_klu *F)
{
    const i64 n = F->n;
    double det = 1.0;
    for (i64 k = 0; k < n; k++) det *= F->Udiag[k] * F->Rs[F->P[k]];
    i64 *w = (i64 *)malloc(((size_t)n + 1) * sizeof(i64));
    i64 npiv = 0;
    for (int pass = 0; pass < 2; pass++) {
        for (i64 i = 0; i < n; i++) w[i] = pass ? F->Q[i] : F->P[i];
        for (i64 i = 0; i < n; i++)
            while (w[i] != i) { i64 t = w[w[i]]; w[w[i]] = w[i]; w[i] = t; npiv++; }
    }
    free(w);
    return (npiv & 1) ? -det : det;
}
