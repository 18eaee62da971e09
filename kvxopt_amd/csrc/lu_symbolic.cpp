// Host analysis of the sparse LU path: see lu_symbolic.hpp.  Reference role: klu_analyze (src/C/klu.c:141,264).
#include "lu_symbolic.hpp"
#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <queue>
#include <stdexcept>

namespace kvx {

// ------------------------------------------------------------------------------------------------------------
// Maximum-product matching: successive shortest augmenting paths on the bipartite graph columns -> rows with
// costs c_ij = log(max_i |a_ij|) - log|a_ij| >= 0 (rows pre-scaled), dual variables u (rows), v (columns) keep
// the reduced costs c_ij - u_i - v_j non-negative and zero on matched edges (the classical sparse assignment
// algorithm; what HSL MC64 job 5 is known for).  Entries that are exactly zero are not candidates.
int64_t lu_matching(int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax, const double *rinv,
                    std::vector<int64_t> &rowfor)
{
    const double INF = std::numeric_limits<double>::infinity();
    const int64_t nnz = n ? Ap[n] : 0;
    std::vector<double> cost((size_t)nnz, 0.0);
    if (Ax) {
        for (int64_t j = 0; j < n; j++) {
            double cmax = 0;
            for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) cmax = std::max(cmax, std::fabs(Ax[p]) * rinv[Ai[p]]);
            const double lmax = cmax > 0 ? std::log(cmax) : 0.0;
            for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
                const double a = std::fabs(Ax[p]) * rinv[Ai[p]];
                cost[p] = (a > 0 && std::isfinite(a)) ? std::max(0.0, lmax - std::log(a)) : INF;
            }
        }
    }
    std::vector<int64_t> rowmatch((size_t)n, -1), colmatch((size_t)n, -1);
    std::vector<double> u((size_t)n, 0.0), v((size_t)n, 0.0);
    for (int64_t j = 0; j < n; j++)                       // cheap start: the column maximum when its row is free
        for (int64_t p = Ap[j]; p < Ap[j + 1]; p++)
            if (cost[p] == 0.0 && rowmatch[Ai[p]] < 0) { rowmatch[Ai[p]] = j; colmatch[j] = Ai[p]; break; }

    std::vector<double> d((size_t)n, INF);
    std::vector<int64_t> pr((size_t)n, -1);
    std::vector<char> done((size_t)n, 0);
    std::vector<int64_t> touched, finalized;
    std::vector<std::pair<int64_t, double>> treecols;
    typedef std::pair<double, int64_t> HeapItem;
    int64_t matched = 0;
    for (int64_t j = 0; j < n; j++) matched += colmatch[j] >= 0;
    for (int64_t j0 = 0; j0 < n; j0++) {
        if (colmatch[j0] >= 0) continue;
        std::priority_queue<HeapItem, std::vector<HeapItem>, std::greater<HeapItem>> heap;
        touched.clear(); finalized.clear(); treecols.clear();
        double lsp = 0;
        int64_t j = j0, sink = -1;
        treecols.emplace_back(j0, 0.0);
        for (;;) {
            for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
                const int64_t i = Ai[p];
                if (done[i] || cost[p] == INF) continue;
                const double dn = lsp + std::max(0.0, cost[p] - u[i] - v[j]);
                if (dn < d[i]) {
                    if (d[i] == INF) touched.push_back(i);
                    d[i] = dn; pr[i] = j;
                    heap.emplace(dn, i);
                }
            }
            int64_t i = -1;
            while (!heap.empty()) {
                const HeapItem t = heap.top(); heap.pop();
                if (!done[t.second] && t.first == d[t.second]) { i = t.second; break; }
            }
            if (i < 0) break;                               // no augmenting path: structurally deficient column
            done[i] = 1; finalized.push_back(i); lsp = d[i];
            if (rowmatch[i] < 0) { sink = i; break; }
            j = rowmatch[i];
            treecols.emplace_back(j, lsp);
        }
        if (sink >= 0) {
            for (auto &t : treecols) v[t.first] += lsp - t.second;
            for (int64_t i : finalized) u[i] -= lsp - d[i];
            int64_t i = sink;
            for (;;) {
                const int64_t jj = pr[i], inext = colmatch[jj];
                colmatch[jj] = i; rowmatch[i] = jj;
                if (jj == j0) break;
                i = inext;
            }
            matched++;
        }
        for (int64_t i : touched) { d[i] = INF; pr[i] = -1; done[i] = 0; }
    }
    // complete a deficient matching arbitrarily so that rowfor is a permutation
    if (matched < n) {
        int64_t fr = 0;
        for (int64_t j = 0; j < n; j++) {
            if (colmatch[j] >= 0) continue;
            while (rowmatch[fr] >= 0) fr++;
            colmatch[j] = fr; rowmatch[fr] = j;
        }
    }
    rowfor.assign(colmatch.begin(), colmatch.end());
    return matched;
}

void lu_analyze(int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax, LuSymbolic &Y)
{
    if (n < 1) throw std::runtime_error("A must have at least one row and column");
    if (n >= (int64_t)1 << 31) throw std::runtime_error("order exceeds 2^31-1");
    if (Ap[0] != 0) throw std::runtime_error("colptr[0] must be 0");
    for (int64_t j = 0; j < n; j++) {
        if (Ap[j + 1] < Ap[j]) throw std::runtime_error("colptr not monotone");
        for (int64_t p = Ap[j]; p < Ap[j + 1]; p++)
            if (Ai[p] < 0 || Ai[p] >= n) throw std::runtime_error("row index out of range");
    }
    Y = LuSymbolic();
    Y.n = n;
    Y.nnz = Ap[n];
    Y.Ap.assign(Ap, Ap + n + 1);
    Y.Ai.assign(Ai, Ai + Y.nnz);
    std::vector<double> rinv((size_t)n, 1.0);
    if (Ax) {
        std::vector<double> rmax((size_t)n, 0.0);
        for (int64_t p = 0; p < Y.nnz; p++) rmax[Ai[p]] = std::max(rmax[Ai[p]], std::fabs(Ax[p]));
        for (int64_t i = 0; i < n; i++) rinv[i] = (rmax[i] > 0 && std::isfinite(rmax[i])) ? 1.0 / rmax[i] : 1.0;
    }
    const int64_t matched = lu_matching(n, Ap, Ai, Ax, rinv.data(), Y.rowfor);
    Y.structurally_singular = matched < n;
    // pattern of tril(M + M') with the diagonal, M(colof[i], j) = A(i, j)
    std::vector<int64_t> colof((size_t)n);
    for (int64_t j = 0; j < n; j++) colof[Y.rowfor[j]] = j;
    std::vector<int64_t> cnt((size_t)n + 1, 0);
    for (int64_t j = 0; j < n; j++) {
        cnt[j + 1]++;                                                  // diagonal
        for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
            const int64_t r = colof[Ai[p]];
            if (r != j) cnt[std::min(r, j) + 1]++;
        }
    }
    for (int64_t j = 0; j < n; j++) cnt[j + 1] += cnt[j];
    std::vector<int64_t> idx((size_t)cnt[n]), cur(cnt.begin(), cnt.end() - 1);
    for (int64_t j = 0; j < n; j++) {
        idx[(size_t)cur[j]++] = j;
        for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
            const int64_t r = colof[Ai[p]];
            if (r != j) idx[(size_t)cur[std::min(r, j)]++] = std::max(r, j);
        }
    }
    std::vector<int64_t> ptr((size_t)n + 1, 0), uniq;
    uniq.reserve(idx.size());
    for (int64_t j = 0; j < n; j++) {
        std::sort(idx.begin() + cnt[j], idx.begin() + cnt[j + 1]);
        auto e = std::unique(idx.begin() + cnt[j], idx.begin() + cnt[j + 1]);
        uniq.insert(uniq.end(), idx.begin() + cnt[j], e);
        ptr[j + 1] = (int64_t)uniq.size();
    }
    SymOpts so;
    analyze(n, ptr.data(), uniq.data(), 'L', nullptr, so, Y.S);
    Y.uf.resize((size_t)Y.S.nsuper);
    std::iota(Y.uf.begin(), Y.uf.end(), 0);
}

static int32_t uf_find(const std::vector<int32_t> &uf, int32_t a)
{
    while (uf[a] != a) a = uf[a];
    return a;
}

void lu_build_plan(const LuSymbolic &Y, LuPlan &P)
{
    const Symbolic &S = Y.S;
    const int64_t n = Y.n, ns = S.nsuper;
    P = LuPlan();
    P.n = n;
    // groups of base supernodes: a merge always links a group's top member to a member of the parent group, so
    // the top member (largest index: parents come later in the postorder) identifies the group
    std::vector<int32_t> rep((size_t)ns), top((size_t)ns, -1);
    for (int64_t s = 0; s < ns; s++) rep[s] = uf_find(Y.uf, (int32_t)s);
    for (int64_t s = 0; s < ns; s++) top[rep[s]] = std::max(top[rep[s]], (int32_t)s);
    // contracted tree over group representatives
    std::vector<int32_t> gparent((size_t)ns, -1);
    std::vector<std::vector<int32_t>> members((size_t)ns), kids((size_t)ns);
    std::vector<int32_t> roots;
    for (int64_t s = 0; s < ns; s++) members[rep[s]].push_back((int32_t)s);
    for (int64_t g = 0; g < ns; g++) {
        if (members[g].empty()) continue;
        const int32_t ps = S.sparent[top[g]];
        gparent[g] = ps >= 0 ? rep[ps] : -1;
        if (gparent[g] == (int32_t)g) throw std::runtime_error("internal: group is its own parent");
    }
    // order groups by their top member so that children lists come out ascending
    std::vector<int32_t> glist;
    for (int64_t g = 0; g < ns; g++) if (!members[g].empty()) glist.push_back((int32_t)g);
    std::sort(glist.begin(), glist.end(), [&](int32_t a, int32_t b) { return top[a] < top[b]; });
    for (int32_t g : glist) {
        if (gparent[g] >= 0) kids[gparent[g]].push_back(g);
        else roots.push_back(g);
    }
    // postorder of the contracted forest
    std::vector<int32_t> order;
    order.reserve(glist.size());
    {
        std::vector<std::pair<int32_t, size_t>> st;
        for (int32_t r : roots) {
            st.emplace_back(r, 0);
            while (!st.empty()) {
                auto &t = st.back();
                if (t.second < kids[t.first].size()) { const int32_t c = kids[t.first][t.second++]; st.emplace_back(c, 0); }
                else { order.push_back(t.first); st.pop_back(); }
            }
        }
    }
    const int64_t nf = (int64_t)order.size();
    P.nfront = nf;
    std::vector<int32_t> g2f((size_t)ns, -1);
    for (int64_t f = 0; f < nf; f++) g2f[order[f]] = (int32_t)f;
    P.group_of.resize((size_t)ns);
    for (int64_t s = 0; s < ns; s++) P.group_of[s] = g2f[rep[s]];
    // positions
    std::vector<int32_t> newpos((size_t)n, -1), col2front((size_t)n);
    P.fr.resize((size_t)nf);
    P.qcol.resize((size_t)n);
    P.prow.resize((size_t)n);
    int64_t run = 0;
    for (int64_t f = 0; f < nf; f++) {
        const int32_t g = order[f];
        P.fr[f].p0 = (int32_t)run;
        for (int32_t s : members[g])
            for (int64_t c = S.super[s]; c < S.super[s + 1]; c++) {
                newpos[c] = (int32_t)run;
                col2front[run] = (int32_t)f;
                const int64_t mlabel = S.perm[c];
                P.qcol[run] = mlabel;
                P.prow[run] = Y.rowfor[mlabel];
                run++;
            }
        P.fr[f].k = (int32_t)(run - P.fr[f].p0);
        P.fr[f].parent = gparent[g] >= 0 ? g2f[gparent[g]] : -1;
        P.fr[f].nchild = (int32_t)kids[g].size();
    }
    if (run != n) throw std::runtime_error("internal: positions do not cover the matrix");
    // front row lists
    P.rowptr.assign((size_t)nf + 1, 0);
    P.rowidx.clear();
    {
        std::vector<int32_t> tail;
        for (int64_t f = 0; f < nf; f++) {
            const int32_t g = order[f];
            const int32_t p0 = P.fr[f].p0, k = P.fr[f].k;
            tail.clear();
            for (int32_t s : members[g])
                for (int64_t q = S.rowptr[s] + S.sn_k[s]; q < S.rowptr[s + 1]; q++) {
                    const int32_t r = newpos[S.rowidx[q]];
                    if (r < p0 || r >= p0 + k) tail.push_back(r);
                }
            std::sort(tail.begin(), tail.end());
            tail.erase(std::unique(tail.begin(), tail.end()), tail.end());
            if (!tail.empty() && tail.front() < p0 + k) throw std::runtime_error("internal: update row before the pivot block");
            for (int32_t t = 0; t < k; t++) P.rowidx.push_back(p0 + t);
            P.rowidx.insert(P.rowidx.end(), tail.begin(), tail.end());
            P.rowptr[f + 1] = (int64_t)P.rowidx.size();
            P.fr[f].m = k + (int32_t)tail.size();
        }
    }
    auto local_index = [&](int64_t f, int32_t pos) -> int32_t {
        const int32_t p0 = P.fr[f].p0, k = P.fr[f].k;
        if (pos >= p0 && pos < p0 + k) return pos - p0;
        const int32_t *b = P.rowidx.data() + P.rowptr[f] + k, *e = P.rowidx.data() + P.rowptr[f + 1];
        const int32_t *it = std::lower_bound(b, e, pos);
        if (it == e || *it != pos) return -1;
        return k + (int32_t)(it - b);
    };
    P.rel.assign(P.rowidx.size(), -1);
    for (int64_t f = 0; f < nf; f++) {
        const int32_t pf = P.fr[f].parent;
        const int32_t k = P.fr[f].k, m = P.fr[f].m;
        if (pf < 0) {
            if (m != k) throw std::runtime_error("internal: root front has update rows");
            continue;
        }
        for (int32_t i = k; i < m; i++) {
            const int32_t li = local_index(pf, P.rowidx[P.rowptr[f] + i]);
            if (li < 0) throw std::runtime_error("internal: child row missing in parent front");
            P.rel[P.rowptr[f] + i] = li;
        }
    }
    // children, storage, levels
    P.childptr.assign((size_t)nf + 1, 0);
    for (int64_t f = 0; f < nf; f++) if (P.fr[f].parent >= 0) P.childptr[P.fr[f].parent + 1]++;
    for (int64_t f = 0; f < nf; f++) P.childptr[f + 1] += P.childptr[f];
    P.children.resize((size_t)P.childptr[nf]);
    {
        std::vector<int64_t> cur(P.childptr.begin(), P.childptr.end() - 1);
        for (int64_t f = 0; f < nf; f++) if (P.fr[f].parent >= 0) P.children[(size_t)cur[P.fr[f].parent]++] = (int32_t)f;
    }
    P.px.assign((size_t)nf + 1, 0);
    P.upd_off.resize((size_t)nf);
    P.upd_ld.resize((size_t)nf);
    P.wx.resize((size_t)nf);
    for (int64_t f = 0; f < nf; f++) {
        const int64_t k = P.fr[f].k, m = P.fr[f].m, u = m - k;
        P.px[f + 1] = P.px[f] + m * k;
        if (m <= KVX_LU_LDS_M) { P.upd_off[f] = P.arena; P.upd_ld[f] = (int32_t)u; P.arena += u * u; }
        else { P.upd_off[f] = P.arena + k + k * m; P.upd_ld[f] = (int32_t)m; P.arena += m * m; }
        if (m > KVX_LU_SOLVE_BIG_M) { P.wx[f] = P.wsize + k; P.wsize += m; }     // whole work vector; update part at offset k
        else { P.wx[f] = P.wsize; P.wsize += u; }
        P.max_m = std::max(P.max_m, (int32_t)m);
        P.max_k = std::max(P.max_k, (int32_t)k);
        P.lnz_bound += k * m - k * (k - 1) / 2;
        P.unz_bound += k * m - k * (k - 1) / 2;
    }
    P.lsize = P.px[nf];
    int32_t maxd = -1;
    for (int64_t f = nf - 1; f >= 0; f--) {
        P.fr[f].depth = P.fr[f].parent >= 0 ? P.fr[P.fr[f].parent].depth + 1 : 0;
        maxd = std::max(maxd, P.fr[f].depth);
    }
    P.nlevels = maxd + 1;
    P.levelptr.assign((size_t)P.nlevels + 1, 0);
    for (int64_t f = 0; f < nf; f++) P.levelptr[P.fr[f].depth + 1]++;
    for (int32_t l = 0; l < P.nlevels; l++) P.levelptr[l + 1] += P.levelptr[l];
    P.levellist.resize((size_t)nf);
    {
        std::vector<int64_t> cur(P.levelptr.begin(), P.levelptr.end() - 1);
        for (int64_t f = 0; f < nf; f++) P.levellist[(size_t)cur[P.fr[f].depth]++] = (int32_t)f;
    }
    P.nlds.assign((size_t)P.nlevels, 0);
    P.nsbig.assign((size_t)P.nlevels, 0);
    for (int32_t l = 0; l < P.nlevels; l++) {
        std::stable_sort(P.levellist.begin() + P.levelptr[l], P.levellist.begin() + P.levelptr[l + 1],
                         [&](int32_t a, int32_t b) {
                             const bool la = P.fr[a].m <= KVX_LU_LDS_M, lb = P.fr[b].m <= KVX_LU_LDS_M;
                             if (la != lb) return la;
                             return la ? P.fr[a].m > P.fr[b].m : P.fr[a].m < P.fr[b].m;
                         });
        for (int64_t q = P.levelptr[l]; q < P.levelptr[l + 1]; q++) {
            P.nlds[l] += P.fr[P.levellist[q]].m <= KVX_LU_LDS_M;
            P.nsbig[l] += P.fr[P.levellist[q]].m > KVX_LU_SOLVE_BIG_M;
        }
    }
    // scatter map of the caller's entries
    std::vector<int32_t> posrow((size_t)n), poscol((size_t)n);
    for (int64_t p = 0; p < n; p++) { posrow[P.prow[p]] = (int32_t)p; poscol[P.qcol[p]] = (int32_t)p; }
    P.aptr.assign((size_t)nf + 1, 0);
    std::vector<int32_t> ef((size_t)Y.nnz);
    for (int64_t j = 0; j < n; j++)
        for (int64_t p = Y.Ap[j]; p < Y.Ap[j + 1]; p++) {
            const int32_t f = col2front[std::min(posrow[Y.Ai[p]], poscol[j])];
            ef[p] = f;
            P.aptr[f + 1]++;
        }
    for (int64_t f = 0; f < nf; f++) P.aptr[f + 1] += P.aptr[f];
    P.a_src.resize((size_t)Y.nnz);
    P.a_dst.resize((size_t)Y.nnz);
    {
        std::vector<int64_t> cur(P.aptr.begin(), P.aptr.end() - 1);
        for (int64_t j = 0; j < n; j++)
            for (int64_t p = Y.Ap[j]; p < Y.Ap[j + 1]; p++) {
                const int32_t f = ef[p];
                const int32_t lr = local_index(f, posrow[Y.Ai[p]]), lc = local_index(f, poscol[j]);
                if (lr < 0 || lc < 0) throw std::runtime_error("internal: entry outside the front structure");
                const int64_t q = cur[f]++;
                P.a_src[q] = p;
                P.a_dst[q] = lr + lc * P.fr[f].m;
            }
    }
}

bool lu_merge_fronts(LuSymbolic &Y, const LuPlan &P, const std::vector<int32_t> &fronts)
{
    // a representative base supernode of every front
    std::vector<int32_t> any((size_t)P.nfront, -1);
    for (int64_t s = 0; s < Y.S.nsuper; s++) any[P.group_of[s]] = (int32_t)s;
    for (int32_t f : fronts) {
        const int32_t pf = P.fr[f].parent;
        if (pf < 0) return false;
        const int32_t a = uf_find(Y.uf, any[f]), b = uf_find(Y.uf, any[pf]);
        if (a != b) { Y.uf[a] = b; Y.nmerges++; }
    }
    return true;
}

}  // namespace kvx
