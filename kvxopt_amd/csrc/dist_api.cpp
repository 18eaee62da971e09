// Sharded mode of the sparse Cholesky (include/kvxhip.h, "sharded mode"): ONE system factored and solved by several
// ranks, one GPU each.  Host orchestration only -- the kernels are the single-GPU ones plus the column-ownership variant of
// the rank-ob update (kernels_big.hip, k_syrk_trailing128<true>).  The collectives themselves are the caller's
// (kvx_dist_comm_fn): this file decides what travels, packs it, and orders the streams around the callback.
//
// Stands in for cholmod_l_factorize / cholmod_l_solve (reference call sites src/C/cholmod.c:362-364, 483); the reference
// has no multi-process counterpart.
#include "chol_internal.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

using namespace kvx;

namespace kvx {

struct DistState {
    DistMap M;
    int rank = 0, nranks = 1;
    double *xchg = nullptr;
    int64_t xchg_cap = 0;
    std::vector<std::vector<int32_t>> shared_at;   // per level: fronts with a range of >= 2 ranks that includes this rank, ascending
    std::vector<int32_t> dist_pos;                 // front -> index into d_dist_ids (block-cyclic fronts of this rank), -1 otherwise
    int32_t *d_dist_ids = nullptr;
    std::vector<int32_t> groups;                   // distinct (lo, hi) ranges of >= 2 ranks
    struct Report { int32_t root; int64_t col0, len; };
    std::vector<Report> report;                    // column intervals of x and the rank that hands them to the others after a solve
    int64_t nshared = 0, ncyclic = 0;
};

void dist_release(kvx_chol *F)
{
    if (!F->dist) return;
    if (F->dist->d_dist_ids) (void)pool_free(F->dist->d_dist_ids);
    delete F->dist;
    F->dist = nullptr;
}

}  // namespace kvx

namespace {

struct Comm {
    kvx_chol *F;
    DistState &D;
    kvx_dist_comm_fn fn;
    void *ctx;
};

// One collective on the exchange buffer.  The caller's collective runs in null-stream order: the null stream is put behind
// the factor's stream (the message was packed there), and the factor's stream behind the null stream afterwards.
int do_comm(Comm &C, int kind, int root, int lo, int hi, int64_t count)
{
    kvx_chol *F = C.F;
    if (!C.fn) { set_err("sharded factor used without a collective callback"); return KVX_EINVAL; }
    HIPCHK(hipEventRecord(F->ev_out, F->stream));
    HIPCHK(hipStreamWaitEvent(nullptr, F->ev_out, 0));
    kvx_dist_op op;
    op.kind = kind; op.root = root; op.lo = lo; op.hi = hi; op.count = count; op.buf_dev = C.D.xchg;
    if (C.fn(C.ctx, &op) != 0) { set_err("collective callback failed"); return KVX_ECOMM; }
    return wait_for_caller(F);
}

struct Region { double *base; int64_t pitch, rows, cols; };   // `cols` columns of `rows` doubles each, `pitch` doubles apart

// Broadcast a list of 2-D regions from `root` to the ranks [lo, hi): packed into messages of whole columns that fit the
// exchange buffer; the root packs, everybody calls the collective, the others unpack into the same addresses.
int bcast_regions(Comm &C, int root, int lo, int hi, const std::vector<Region> &regs)
{
    kvx_chol *F = C.F;
    DistState &D = C.D;
    hipStream_t st = F->stream;
    const bool mine = D.rank == root;
    struct Piece { double *base; int64_t pitch, rows, cols, off; };
    std::vector<Piece> msg;
    size_t i = 0;
    int64_t coloff = 0;
    while (i < regs.size()) {
        msg.clear();
        int64_t used = 0;
        while (i < regs.size()) {
            const Region &R = regs[i];
            if (R.rows <= 0 || R.cols <= 0) { i++; coloff = 0; continue; }
            const int64_t room = (D.xchg_cap - used) / R.rows;
            if (room <= 0) break;
            const int64_t nc = std::min(room, R.cols - coloff);
            msg.push_back(Piece{R.base + coloff * R.pitch, R.pitch, R.rows, nc, used});
            used += nc * R.rows;
            coloff += nc;
            if (coloff >= R.cols) { i++; coloff = 0; } else break;
        }
        if (msg.empty()) {
            if (i >= regs.size()) break;
            set_err("exchange buffer smaller than one front column (kvx_chol_dist_set_xchg)");
            return KVX_EINVAL;
        }
        if (mine)
            for (const Piece &p : msg)
                HIPCHK(hipMemcpy2DAsync(D.xchg + p.off, p.rows * sizeof(double), p.base, p.pitch * sizeof(double),
                                        p.rows * sizeof(double), p.cols, hipMemcpyDeviceToDevice, st));
        int rc = do_comm(C, KVX_DIST_BCAST, root, lo, hi, used);
        if (rc) return rc;
        if (!mine)
            for (const Piece &p : msg)
                HIPCHK(hipMemcpy2DAsync(p.base, p.pitch * sizeof(double), D.xchg + p.off, p.rows * sizeof(double),
                                        p.rows * sizeof(double), p.cols, hipMemcpyDeviceToDevice, st));
    }
    return KVX_OK;
}

// Before a shared front s (level l) assembles: every rank of its range needs the update matrices of all children.
int gather_children_updates(Comm &C, int32_t s, int l)
{
    kvx_chol *F = C.F;
    const Symbolic &S = F->S;
    const DistMap &M = C.D.M;
    const int lo = M.glo[s], hi = M.ghi[s], OB = M.ob;
    double *Uch = F->d_U[(l + 1) & 1];
    std::vector<Region> regs;
    for (int64_t q = S.childptr[s]; q < S.childptr[s + 1]; q++) {
        const int32_t c = S.children[q];
        const int64_t kc = S.sn_k[c], uc = S.sn_m[c] - kc;
        if (uc == 0) continue;
        double *U = Uch + S.ux[c];
        const int gc = M.ghi[c] - M.glo[c];
        if (M.mode[c]) {
            // block-cyclic child: block nkb + j / OB of its update matrix sits on rank glo + (that index mod gc)
            const int64_t nkb = (kc + OB - 1) / OB;
            for (int r = 0; r < gc; r++) {
                regs.clear();
                for (int64_t j0 = 0; j0 < uc; j0 += OB)
                    if ((int)((nkb + j0 / OB) % gc) == r)
                        regs.push_back(Region{U + j0 + j0 * uc, uc, uc - j0, std::min<int64_t>(OB, uc - j0)});
                if (regs.empty()) continue;
                int rc = bcast_regions(C, M.glo[c] + r, lo, hi, regs);
                if (rc) return rc;
            }
        } else if (M.glo[c] == lo && M.ghi[c] == hi) {
            continue;                                   // replicated on exactly these ranks: everybody computed it
        } else {
            // owned by one rank, or replicated on a sub-range: its first rank sends the lower trapezoid, block by block
            regs.clear();
            for (int64_t j0 = 0; j0 < uc; j0 += OB)
                regs.push_back(Region{U + j0 + j0 * uc, uc, uc - j0, std::min<int64_t>(OB, uc - j0)});
            int rc = bcast_regions(C, M.glo[c], lo, hi, regs);
            if (rc) return rc;
        }
    }
    return KVX_OK;
}

// Block-cyclic front: right-looking over the pivot blocks.  Every rank of the range assembles the whole front (HBM-bound,
// cheap beside the updates) and ends with the complete panel of L (the broadcasts) and with ITS blocks of the update matrix.
int factor_cyclic_front(Comm &C, int32_t s, int l)
{
    kvx_chol *F = C.F;
    const Symbolic &S = F->S;
    DistState &D = C.D;
    const DistMap &M = D.M;
    hipStream_t st = F->stream;
    const int lo = M.glo[s], hi = M.ghi[s], g = hi - lo, r = D.rank - lo, OB = M.ob;
    const int m = S.sn_m[s], k = S.sn_k[s];
    const int32_t *list = D.d_dist_ids + D.dist_pos[s];
    double *Uout = F->d_U[l & 1];
    const double *Uch = F->d_U[(l + 1) & 1];
    double *P = F->d_Lx + S.px[s];
    double *Y = F->d_Linv + F->linv_off_host[s];
    { ProfScope ps(F, FAM_ASSEMBLE); launch_assemble_big(st, F->ds, list, 1, m, F->d_Lx, Uch, Uout); }
    std::vector<Region> regs;
    // the panel of a pivot block: the single-GPU chain restricted to the block (the update of panel jb also factors the
    // diagonal block of panel jb + 64 while that is inside the block)
    auto factor_panel = [&](int o, int nb) {
        { ProfScope ps(F, FAM_POTRF); launch_potrf_blk(st, F->ds, list, 1, o, F->d_Lx, F->d_Linv, F->d_status); }
        for (int jb = o; jb < o + nb; jb += KVX_NB) {
            { ProfScope ps(F, FAM_TRSM); launch_trsm_blk(st, F->ds, list, 1, m, jb, F->d_Lx, F->d_Linv); }
            { ProfScope ps(F, FAM_SYRK); launch_syrk_inner(st, F->ds, list, 1, m, jb, o + nb, F->d_Lx, Uout, F->d_Linv, F->d_status); }
        }
    };
    auto update = [&](int o, int nb, int c_from, int c_to) {
        ProfScope ps(F, FAM_SYRK);
        launch_syrk_outer_dist(st, F->ds, list, m, o, nb, OB, g, r, c_from, c_to, F->d_Lx, Uout);
    };
    if (D.rank == lo) factor_panel(0, std::min(OB, k));
    int b = 0;
    for (int o = 0; o < k; o += OB, b++) {
        const int nb = std::min(OB, k - o), owner = lo + b % g;
        regs.clear();
        regs.push_back(Region{P + o + (int64_t)o * m, m, m - o, nb});
        regs.push_back(Region{Y + (int64_t)(o / KVX_NB) * KVX_NB * KVX_NB, KVX_NB * KVX_NB, KVX_NB * KVX_NB, (nb + KVX_NB - 1) / KVX_NB});
        int rc = bcast_regions(C, owner, lo, hi, regs);
        if (rc) return rc;
        if (o + nb >= m) break;
        // Look-ahead: the owner of the NEXT pivot block brings that block up to date first, factors its panel, and only
        // then joins the rest of this update -- with a stream-ordered collective (RCCL) its panel chain runs while the other
        // ranks are still updating, and the next broadcast finds the panel ready.  Every entry still receives the same
        // updates in the same order as without look-ahead: the factor is bitwise the same.
        const int o2 = o + nb;
        if (o2 < k && D.rank == lo + (b + 1) % g) {
            const int nb2 = std::min(OB, k - o2);
            update(o, nb, o2, o2 + nb2);
            factor_panel(o2, nb2);
            update(o, nb, o2 + nb2, m);
        } else {
            update(o, nb, o2, m);
        }
    }
    HIPCHK(hipGetLastError());
    return KVX_OK;
}

int gather_children_vectors(Comm &C, int32_t s, int l, int nr, int64_t wstride)
{
    kvx_chol *F = C.F;
    const Symbolic &S = F->S;
    const DistMap &M = C.D.M;
    const int lo = M.glo[s], hi = M.ghi[s];
    double *Wch = F->d_W[(l + 1) & 1];
    std::vector<Region> regs(1);
    for (int64_t q = S.childptr[s]; q < S.childptr[s + 1]; q++) {
        const int32_t c = S.children[q];
        const int64_t uc = S.sn_m[c] - S.sn_k[c];
        if (uc == 0 || (M.glo[c] == lo && M.ghi[c] == hi)) continue;       // (the ranks of a shared child all hold the same vector)
        regs[0] = Region{Wch + S.wx[c], wstride, uc, nr};
        int rc = bcast_regions(C, M.glo[c], lo, hi, regs);
        if (rc) return rc;
    }
    return KVX_OK;
}

int dist_factorize_impl(kvx_chol *F, const double *values_dev, kvx_dist_comm_fn fn, void *ctx, int64_t *minor)
{
    if (!F || !F->dev_ready || !F->dist) { set_err("kvx_chol_dist_setup has not been called"); return KVX_EINVAL; }
    Symbolic &S = F->S;
    DistState &D = *F->dist;
    Comm C{F, D, fn, ctx};
    hipStream_t st = F->stream;
    int rc;
    if ((rc = wait_for_caller(F))) return rc;
    if (S.nnzA > 0) HIPCHK(hipMemcpyAsync(F->d_Ax, values_dev, S.nnzA * sizeof(double), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipEventRecord(F->ev[0], st));
    F->diag_valid = false;
    F->numeric = false;
    HIPCHK(hipMemsetAsync(F->d_Lx, 0, std::max<int64_t>(S.lsize, 1) * sizeof(double), st));
    HIPCHK(hipMemsetAsync(F->d_status, 0x7f, sizeof(int), st));
    launch_scatter_a(st, F->d_Ax, F->d_amap, S.nnzA, F->d_Lx);
    int run_from = S.nlevels - 1;
    for (int l = S.nlevels - 1; l >= 0; l--) {
        if (D.shared_at[l].empty()) continue;
        if (run_from > l && (rc = enqueue_factor_body(F, run_from, l + 1, false, false))) return rc;   // levels without an exchange: one batch
        for (int32_t s : D.shared_at[l])
            if ((rc = gather_children_updates(C, s, l))) return rc;
        if ((rc = enqueue_factor_body(F, l, l, false, false))) return rc;        // this rank's own and replicated fronts of the level
        for (int32_t s : D.shared_at[l])
            if (D.M.mode[s] && (rc = factor_cyclic_front(C, s, l))) return rc;
        run_from = l - 1;
    }
    if (run_from >= 0 && (rc = enqueue_factor_body(F, run_from, 0, false, false))) return rc;
    HIPCHK(hipMemcpyAsync(F->h_status, F->d_status, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(F->ev[1], st));
    HIPCHK(hipStreamSynchronize(st));
    const int stw = *F->h_status;
    double mn = (stw >= 0x7f7f7f7f) ? (double)S.n : (double)stw;
    if (D.nranks > 1) {                              // a failing column may sit in another rank's subtree: MIN over the ranks
        HIPCHK(hipMemcpyAsync(D.xchg, &mn, sizeof(double), hipMemcpyHostToDevice, st));
        if ((rc = do_comm(C, KVX_DIST_ALLREDUCE_MIN, 0, 0, D.nranks, 1))) return rc;
        HIPCHK(hipMemcpyAsync(&mn, D.xchg, sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    float ms = 0;
    if (hipEventElapsedTime(&ms, F->ev[0], F->ev[1]) == hipSuccess) { F->ms_factor = ms; F->have_ftime = true; }
    prof_collect(F);
    F->pending = false;
    F->numeric = true;
    F->minor = (int64_t)mn;
    if (minor) *minor = F->minor;
    return F->minor < S.n ? KVX_ENOTPOSDEF : KVX_OK;
}

int dist_solve_impl(kvx_chol *F, double *B, int64_t nrhs, int64_t ldB, kvx_dist_comm_fn fn, void *ctx)
{
    if (!F || !F->dev_ready || !F->dist) { set_err("kvx_chol_dist_setup has not been called"); return KVX_EINVAL; }
    if (nrhs < 0 || nrhs > 65535) { set_err("nrhs out of range"); return KVX_EINVAL; }
    if (!F->numeric) { set_err("called with symbolic factor"); return KVX_ESYMBOLIC; }
    if (F->minor < F->S.n) { set_err("singular matrix"); return KVX_ESINGULAR; }
    Symbolic &S = F->S;
    DistState &D = *F->dist;
    Comm C{F, D, fn, ctx};
    hipStream_t st = F->stream;
    const int64_t n = S.n;
    if (n == 0 || nrhs == 0) return KVX_OK;
    if (ldB < n) { set_err("ldB must be >= max(1,n)"); return KVX_EINVAL; }
    const int nr = (int)nrhs;
    int rc;
    if ((rc = ensure_solve_ws(F, nr))) return rc;
    const int64_t wstride = std::max(S.wrk_size[0], S.wrk_size[1]);
    if ((rc = wait_for_caller(F))) return rc;
    HIPCHK(hipEventRecord(F->ev[2], st));
    launch_perm_gather(st, F->d_perm, n, nr, B, ldB, F->d_X, n);
    HIPCHK(hipMemcpyAsync(F->d_X0, F->d_X, (size_t)n * nr * sizeof(double), hipMemcpyDeviceToDevice, st));
    int run_from = S.nlevels - 1;
    for (int l = S.nlevels - 1; l >= 0; l--) {
        if (D.shared_at[l].empty()) continue;
        if (run_from > l) enqueue_fwd(F, F->d_X, n, nr, run_from, l + 1);
        for (int32_t s : D.shared_at[l])
            if ((rc = gather_children_vectors(C, s, l, nr, wstride))) return rc;
        enqueue_fwd(F, F->d_X, n, nr, l, l);
        run_from = l - 1;
    }
    if (run_from >= 0) enqueue_fwd(F, F->d_X, n, nr, run_from, 0);
    // backward: every rank holds x of all ancestors of its fronts (it is in their ranges): no exchange
    enqueue_bwd(F, F->d_X, n, nr, 0, S.nlevels - 1);
    static const bool allreduce_x = [] { const char *e = getenv("KVX_DIST_ALLREDUCE_X"); return e && e[0] == '1'; }();
    if (D.nranks > 1 && !allreduce_x) {
        // Round 4: after the backward sweep a rank holds x of its own subtrees and of every shared front above them; what it lacks
        // is the other ranks' parts.  The first rank of a front's range broadcasts that front's entries (runs of fronts merged
        // into intervals, packed into messages): n nrhs doubles received per rank, where the all-reduce of the whole masked
        // vector that stood here moved twice that and added zeros (KVX_DIST_ALLREDUCE_X=1 restores it).
        std::vector<Region> regs;
        for (int r = 0; r < D.nranks; r++) {
            regs.clear();
            for (const DistState::Report &q : D.report)
                if (q.root == r)
                    for (int64_t o = 0; o < q.len; o += D.xchg_cap)      // (a message holds whole columns of a region: no longer than the buffer)
                        regs.push_back(Region{F->d_X + q.col0 + o, n, std::min(D.xchg_cap, q.len - o), (int64_t)nr});
            if (regs.empty()) continue;
            if ((rc = bcast_regions(C, r, 0, D.nranks, regs))) return rc;
        }
    } else if (D.nranks > 1) {
        // every entry of x is reported by ONE rank (the first of its front's range); the sum over the ranks is x
        launch_mask_rows(st, F->d_keep, n, nr, F->d_X, n);
        const int64_t total = n * nr;
        for (int64_t o = 0; o < total; o += D.xchg_cap) {
            const int64_t cnt = std::min(D.xchg_cap, total - o);
            HIPCHK(hipMemcpyAsync(D.xchg, F->d_X + o, cnt * sizeof(double), hipMemcpyDeviceToDevice, st));
            if ((rc = do_comm(C, KVX_DIST_ALLREDUCE, 0, 0, D.nranks, cnt))) return rc;
            HIPCHK(hipMemcpyAsync(F->d_X + o, D.xchg, cnt * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
    }
    launch_perm_scatter(st, F->d_perm, n, nr, F->d_X, n, B, ldB);
    HIPCHK(hipEventRecord(F->ev[3], st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    prof_collect(F);
    float ms = 0;
    if (hipEventElapsedTime(&ms, F->ev[2], F->ev[3]) == hipSuccess) { F->ms_solve = ms; F->have_stime = true; }
    return KVX_OK;
}

// Per-rank layout (before anything is allocated): a rank holds the panels (and inverted diagonal blocks) of the fronts whose
// rank range includes it -- its own subtrees plus the shared fronts above them, whose complete panels reach it through the
// broadcasts -- and an update-matrix slot for those fronts and for the children of its shared fronts (received from their
// owners before the parent assembles).  Everything else of the factor never exists on this rank: S.px / S.ux / S.amap are
// rewritten to the compact layout, S.lsize / S.upd_size shrink, F->lsize_total keeps the size of the whole factor.
void trim_to_rank(kvx_chol *F, const DistMap &M, int rank)
{
    Symbolic &S = F->S;
    const int64_t ns = S.nsuper;
    F->part.assign((size_t)ns, 0);
    for (int64_t s = 0; s < ns; s++) F->part[s] = (rank >= M.glo[s] && rank < M.ghi[s]) ? 1 : 0;
    std::vector<int64_t> npx((size_t)ns + 1, 0);
    for (int64_t s = 0; s < ns; s++) npx[s + 1] = npx[s] + (F->part[s] ? (int64_t)S.sn_m[s] * S.sn_k[s] : 0);
    for (size_t e = 0; e < S.amap.size(); e++) {
        const int64_t a = S.amap[e];
        if (a < 0) continue;
        const int64_t s = (int64_t)(std::upper_bound(S.px.begin(), S.px.end(), a) - S.px.begin()) - 1;
        S.amap[e] = F->part[s] ? npx[s] + (a - S.px[s]) : -1;
    }
    F->lsize_total = S.lsize;
    S.px = npx;
    S.lsize = npx[ns];
    S.upd_size[0] = S.upd_size[1] = 0;
    for (int32_t l = 0; l < S.nlevels; l++) {
        int64_t off = 0;
        for (int64_t q = S.levelptr[l]; q < S.levelptr[l + 1]; q++) {
            const int32_t s = S.levellist[q];
            const int32_t p = S.sparent[s];
            if (!(F->part[s] || (p >= 0 && F->part[p]))) { S.ux[s] = 0; continue; }
            const int64_t u = S.sn_m[s] - S.sn_k[s];
            S.ux[s] = off;
            off += u * u;
        }
        S.upd_size[l & 1] = std::max(S.upd_size[l & 1], off);
    }
}

int dist_setup_impl(kvx_chol *F, int rank, int nranks, int ob, int min_m, int64_t info[8])
{
    if (!F || nranks < 1 || rank < 0 || rank >= nranks || !info) return KVX_EINVAL;
    Symbolic &S = F->S;
    if (const char *e = getenv("KVX_DIST_OB")) ob = atoi(e);
    if (const char *e = getenv("KVX_DIST_MIN_M")) min_m = atoi(e);
    DistMap M0;
    dist_map(S, nranks, ob > 0 ? ob : 512, min_m > 0 ? min_m : 6144, M0);
    // the device objects are created HERE when the handle has none yet: panels, inverted diagonal blocks and update matrices
    // are then laid out for the fronts this rank takes part in only (KVX_DIST_NO_TRIM=1: the whole factor on every rank)
    if (!F->dev_ready && nranks > 1 && !(getenv("KVX_DIST_NO_TRIM") && getenv("KVX_DIST_NO_TRIM")[0] == '1')) trim_to_rank(F, M0, rank);
    int rc = ensure_device(F);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(F->stream));
    dist_release(F);
    F->dist = new DistState();
    DistState &D = *F->dist;
    D.rank = rank; D.nranks = nranks;
    D.M = std::move(M0);
    const DistMap &M = D.M;
    // level lists: the fronts this rank takes part in (solves); without the block-cyclic ones (factorisation)
    std::vector<int32_t> flists;
    std::vector<int64_t> flptr((size_t)S.nlevels + 1, 0);
    F->lists_host.clear();
    F->lptr_host.assign((size_t)S.nlevels + 1, 0);
    D.shared_at.assign((size_t)S.nlevels, std::vector<int32_t>());
    D.dist_pos.assign((size_t)S.nsuper, -1);
    std::vector<int32_t> dist_ids;
    for (int l = 0; l < S.nlevels; l++) {
        for (int64_t q = S.levelptr[l]; q < S.levelptr[l + 1]; q++) {
            const int32_t s = S.levellist[q];
            if (rank < M.glo[s] || rank >= M.ghi[s]) continue;
            F->lists_host.push_back(s);
            if (!M.mode[s]) flists.push_back(s);
            if (M.ghi[s] - M.glo[s] > 1) {
                D.shared_at[l].push_back(s);
                D.nshared++;
                if (M.mode[s]) { D.dist_pos[s] = (int32_t)dist_ids.size(); dist_ids.push_back(s); D.ncyclic++; }
            }
        }
        std::sort(D.shared_at[l].begin(), D.shared_at[l].end());          // the same order on every rank
        F->lptr_host[l + 1] = (int64_t)F->lists_host.size();
        flptr[l + 1] = (int64_t)flists.size();
    }
    if (F->d_lists) { HIPCHK(pool_free(F->d_lists)); F->d_lists = nullptr; }
    if ((rc = upload(&F->d_lists, F->lists_host))) return rc;
    if (F->d_flists) { HIPCHK(pool_free(F->d_flists)); F->d_flists = nullptr; }
    if ((rc = upload(&F->d_flists, flists))) return rc;
    F->flists_host = flists;
    if ((rc = upload(&D.d_dist_ids, dist_ids))) return rc;
    build_plan(F);
    build_plan_from(S, flists, flptr, F->fplan);
    F->fplan_on = true;
    F->dist_rank = rank; F->dist_nranks = nranks;        // (with more than one rank the leaf subtrees of the solves stay in the level lists)
    if ((rc = build_subtrees(F))) return rc;
    destroy_graphs(F);
    // which entries of x this rank reports
    std::vector<uint8_t> keep((size_t)std::max<int64_t>(S.n, 1), 0);
    for (int64_t s = 0; s < S.nsuper; s++)
        if (M.glo[s] == rank)
            for (int64_t j = S.super[s]; j < S.super[s + 1]; j++) keep[j] = 1;
    if (F->d_keep) { HIPCHK(pool_free(F->d_keep)); F->d_keep = nullptr; }
    if ((rc = upload(&F->d_keep, keep))) return rc;
    // who hands which part of x to the others at the end of a solve: the first rank of a front's range, runs of fronts merged
    D.report.clear();
    for (int64_t s = 0; s < S.nsuper; s++) {
        const int64_t c0 = S.super[s], len = S.super[s + 1] - S.super[s];
        if (len <= 0) continue;
        if (!D.report.empty() && D.report.back().root == M.glo[s] && D.report.back().col0 + D.report.back().len == c0) D.report.back().len += len;
        else D.report.push_back(DistState::Report{M.glo[s], c0, len});
    }
    // distinct ranges of >= 2 ranks (the caller creates one communicator per range)
    for (int64_t s = 0; s < S.nsuper; s++) {
        if (M.ghi[s] - M.glo[s] < 2) continue;
        bool seen = false;
        for (size_t i = 0; i < D.groups.size(); i += 2) seen = seen || (D.groups[i] == M.glo[s] && D.groups[i + 1] == M.ghi[s]);
        if (!seen) { D.groups.push_back(M.glo[s]); D.groups.push_back(M.ghi[s]); }
    }
    int64_t need = std::max<int64_t>(std::max<int64_t>(S.max_m, KVX_NB * KVX_NB), 1);
    info[0] = S.n; info[1] = need;
    info[2] = std::max(need, std::min<int64_t>((int64_t)1 << 25, std::max<int64_t>((int64_t)S.max_m * M.ob, S.n)));
    info[3] = D.nshared; info[4] = D.ncyclic; info[5] = (int64_t)D.groups.size() / 2; info[6] = M.ob; info[7] = M.min_m;
    return KVX_OK;
}

}  // namespace

extern "C" {

int kvx_chol_dist_map(kvx_chol *F, int nranks, int ob, int min_m, int32_t *glo, int32_t *ghi, uint8_t *mode,
                      double *rank_flops, double *panel_flops, double totals[2])
{
    return guarded([&] {
        if (!F || nranks < 1) return (int)KVX_EINVAL;
        DistMap M;
        dist_map(F->S, nranks, ob > 0 ? ob : 512, min_m > 0 ? min_m : 6144, M);
        if (glo) std::copy(M.glo.begin(), M.glo.end(), glo);
        if (ghi) std::copy(M.ghi.begin(), M.ghi.end(), ghi);
        if (mode) std::copy(M.mode.begin(), M.mode.end(), mode);
        if (rank_flops) std::copy(M.rank_flops.begin(), M.rank_flops.end(), rank_flops);
        if (panel_flops) std::copy(M.rank_panel_flops.begin(), M.rank_panel_flops.end(), panel_flops);
        if (totals) { totals[0] = M.total_flops; totals[1] = M.replicated_flops; }
        return (int)KVX_OK;
    });
}

int kvx_chol_dist_setup(kvx_chol *F, int rank, int nranks, int ob, int min_m, int64_t info[8])
{
    return guarded([&] { return dist_setup_impl(F, rank, nranks, ob, min_m, info); });
}

int kvx_chol_dist_groups(kvx_chol *F, int32_t *lohi)
{
    if (!F || !F->dist || !lohi) return KVX_EINVAL;
    std::copy(F->dist->groups.begin(), F->dist->groups.end(), lohi);
    return KVX_OK;
}

int kvx_chol_dist_set_xchg(kvx_chol *F, double *xchg_dev, int64_t count)
{
    if (!F || !F->dist || !xchg_dev || count < 1) return KVX_EINVAL;
    F->dist->xchg = xchg_dev;
    F->dist->xchg_cap = count;
    return KVX_OK;
}

int kvx_chol_dist_factorize(kvx_chol *F, const double *values_dev, kvx_dist_comm_fn comm, void *ctx, int64_t *minor)
{
    return guarded([&] {
        if (F && F->dist && !F->dist->xchg && F->dist->nranks > 1) { set_err("no exchange buffer (kvx_chol_dist_set_xchg)"); return (int)KVX_EINVAL; }
        return dist_factorize_impl(F, values_dev, comm, ctx, minor);
    });
}

int kvx_chol_dist_solve(kvx_chol *F, double *B_dev, int64_t nrhs, int64_t ldB, kvx_dist_comm_fn comm, void *ctx)
{
    return guarded([&] {
        if (F && F->dist && !F->dist->xchg && F->dist->nranks > 1) { set_err("no exchange buffer (kvx_chol_dist_set_xchg)"); return (int)KVX_EINVAL; }
        return dist_solve_impl(F, B_dev, nrhs, ldB, comm, ctx);
    });
}

}  // extern "C"
