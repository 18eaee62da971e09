// Host analysis of the sparse LU path: see lu_symbolic.hpp.  Reference role: klu_analyze (src/C/klu.c:141,264).
#include "lu_symbolic.hpp"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
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
    const bool tim = std::getenv("KVX_ANALYZE_TIMING") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *w) { if (tim) { auto t1 = std::chrono::steady_clock::now(); fprintf(stderr, "  lu_analyze %-20s %8.3f ms\n", w, std::chrono::duration<double, std::milli>(t1 - t0).count()); t0 = t1; } };
    const int64_t matched = lu_matching(n, Ap, Ai, Ax, rinv.data(), Y.rowfor);
    lap("matching");
    Y.structurally_singular = matched < n;
    // M(colof[i], j) = A(i, j)
    std::vector<int64_t> colof((size_t)n);
    for (int64_t j = 0; j < n; j++) colof[Y.rowfor[j]] = j;
    // ---- block triangular form: strongly connected components of the graph r -> c for every M(r, c) != 0 (Tarjan,
    // iterative).  Components come out sinks first; numbering them backwards makes every entry satisfy blk[r] <= blk[c].
    Y.blk.assign((size_t)n, 0);
    Y.nblocks = 1; Y.nblev = 1;
    Y.blev.assign(1, 0);
    {
        // CSR of M (rows = M row labels)
        std::vector<int64_t> rp((size_t)n + 1, 0);
        for (int64_t p = 0; p < Y.nnz; p++) rp[colof[Ai[p]] + 1]++;
        for (int64_t i = 0; i < n; i++) rp[i + 1] += rp[i];
        std::vector<int32_t> rc((size_t)Y.nnz);
        {
            std::vector<int64_t> cur(rp.begin(), rp.end() - 1);
            for (int64_t j = 0; j < n; j++)
                for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) rc[(size_t)cur[colof[Ai[p]]]++] = (int32_t)j;
        }
        std::vector<int32_t> index((size_t)n, -1), low((size_t)n, 0), comp((size_t)n, -1), stack, callv;
        std::vector<int64_t> callp;
        std::vector<char> onstack((size_t)n, 0);
        int32_t counter = 0, ncomp = 0;
        for (int64_t root = 0; root < n; root++) {
            if (index[root] >= 0) continue;
            callv.push_back((int32_t)root); callp.push_back(rp[root]);
            index[root] = low[root] = counter++; stack.push_back((int32_t)root); onstack[root] = 1;
            while (!callv.empty()) {
                const int32_t v = callv.back();
                int64_t &pp = callp.back();
                if (pp < rp[v + 1]) {
                    const int32_t w = rc[(size_t)pp++];
                    if (index[w] < 0) {
                        index[w] = low[w] = counter++; stack.push_back(w); onstack[w] = 1;
                        callv.push_back(w); callp.push_back(rp[w]);
                    } else if (onstack[w]) low[v] = std::min(low[v], index[w]);
                } else {
                    if (low[v] == index[v]) {
                        for (;;) { const int32_t w = stack.back(); stack.pop_back(); onstack[w] = 0; comp[w] = ncomp; if (w == v) break; }
                        ncomp++;
                    }
                    callv.pop_back(); callp.pop_back();
                    if (!callv.empty()) low[callv.back()] = std::min(low[callv.back()], low[v]);
                }
            }
        }
        const char *env = std::getenv("KVX_LU_NO_BTF");
        if (ncomp > 1 && !(env && env[0] == '1') && !Y.structurally_singular) {
            std::vector<int32_t> blk((size_t)n);
            for (int64_t v = 0; v < n; v++) blk[v] = ncomp - 1 - comp[v];
            // block levels for the back substitution: block k waits for the later blocks its rows touch
            std::vector<int32_t> blev((size_t)ncomp, 0);
            std::vector<std::vector<int32_t>> rows((size_t)ncomp);
            for (int64_t v = 0; v < n; v++) rows[blk[v]].push_back((int32_t)v);
            int32_t maxlev = 0;
            for (int32_t k = ncomp - 1; k >= 0; k--) {
                int32_t lv = 0;
                for (int32_t r : rows[k])
                    for (int64_t q = rp[r]; q < rp[r + 1]; q++) {
                        const int32_t j = blk[rc[(size_t)q]];
                        if (j != k) lv = std::max(lv, blev[j] + 1);
                    }
                blev[k] = lv;
                maxlev = std::max(maxlev, lv);
            }
            if (maxlev < 64) {                  // deep chains of tiny blocks are a sequential sparse triangular solve: one block then
                Y.blk = blk; Y.blev = blev; Y.nblocks = ncomp; Y.nblev = maxlev + 1;
            }
        }
    }
    lap("btf");
    // pattern of tril(D + D') with the diagonal, D = the diagonal blocks of M
    std::vector<int64_t> cnt((size_t)n + 1, 0);
    for (int64_t j = 0; j < n; j++) {
        cnt[j + 1]++;                                                  // diagonal
        for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
            const int64_t r = colof[Ai[p]];
            if (r != j && Y.blk[r] == Y.blk[j]) cnt[std::min(r, j) + 1]++;
        }
    }
    for (int64_t j = 0; j < n; j++) cnt[j + 1] += cnt[j];
    std::vector<int64_t> idx((size_t)cnt[n]), cur(cnt.begin(), cnt.end() - 1);
    for (int64_t j = 0; j < n; j++) {
        idx[(size_t)cur[j]++] = j;
        for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
            const int64_t r = colof[Ai[p]];
            if (r != j && Y.blk[r] == Y.blk[j]) idx[(size_t)cur[std::min(r, j)]++] = std::max(r, j);
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
    lap("pattern");
    SymOpts so;
    analyze(n, ptr.data(), uniq.data(), 'L', nullptr, so, Y.S);
    lap("cholesky analysis");
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
    // block triangular form: the trees (one per diagonal block) follow each other in block order
    auto block_of_group = [&](int32_t g) { return Y.blk[(size_t)S.perm[S.super[top[g]]]]; };
    std::stable_sort(roots.begin(), roots.end(), [&](int32_t a, int32_t b) { return block_of_group(a) < block_of_group(b); });
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
    // ---- position blocks, solve stages ----------------------------------------------------------------------
    std::vector<int32_t> pblk((size_t)n);
    for (int64_t p = 0; p < n; p++) pblk[p] = Y.blk[(size_t)P.qcol[p]];
    P.rblocks.clear();
    for (int64_t p = 0; p < n; p++) {
        if (p == 0 || pblk[p] != pblk[p - 1]) {
            if (p > 0 && pblk[p] < pblk[p - 1]) throw std::runtime_error("internal: blocks out of order");
            P.rblocks.push_back(p);
        }
    }
    P.rblocks.push_back(n);
    if ((int64_t)P.rblocks.size() != Y.nblocks + 1) throw std::runtime_error("internal: a diagonal block is not contiguous");
    P.nblev = (int32_t)Y.nblev;
    {
        std::vector<int32_t> flev((size_t)nf), maxdep((size_t)P.nblev, -1);
        for (int64_t f = 0; f < nf; f++) {
            flev[f] = Y.blev[(size_t)pblk[P.fr[f].p0]];
            maxdep[flev[f]] = std::max(maxdep[flev[f]], P.fr[f].depth);
        }
        P.levstage.assign((size_t)P.nblev + 1, 0);
        for (int32_t l = 0; l < P.nblev; l++) P.levstage[l + 1] = P.levstage[l] + maxdep[l] + 1;
        P.nstage = P.levstage[P.nblev];
        P.stageptr.assign((size_t)P.nstage + 1, 0);
        for (int64_t f = 0; f < nf; f++) P.stageptr[P.levstage[flev[f]] + P.fr[f].depth + 1]++;
        for (int32_t t = 0; t < P.nstage; t++) P.stageptr[t + 1] += P.stageptr[t];
        P.stagelist.resize((size_t)nf);
        std::vector<int64_t> cur(P.stageptr.begin(), P.stageptr.end() - 1);
        for (int64_t f = 0; f < nf; f++) P.stagelist[(size_t)cur[P.levstage[flev[f]] + P.fr[f].depth]++] = (int32_t)f;
        P.stage_nbig.assign((size_t)P.nstage, 0);
        P.stage_smallm.assign((size_t)P.nstage, 0); P.stage_smallk.assign((size_t)P.nstage, 0);
        P.stage_bigm.assign((size_t)P.nstage, 0); P.stage_bigk.assign((size_t)P.nstage, 0);
        for (int32_t t = 0; t < P.nstage; t++) {
            std::stable_sort(P.stagelist.begin() + P.stageptr[t], P.stagelist.begin() + P.stageptr[t + 1],
                             [&](int32_t x, int32_t y) {
                                 const bool bx = P.fr[x].m > KVX_LU_SOLVE_BIG_M, by = P.fr[y].m > KVX_LU_SOLVE_BIG_M;
                                 return bx != by ? !bx : P.fr[x].m < P.fr[y].m;
                             });
            for (int64_t q = P.stageptr[t]; q < P.stageptr[t + 1]; q++) {
                const LuFrontH &fh = P.fr[P.stagelist[q]];
                if (fh.m > KVX_LU_SOLVE_BIG_M) {
                    P.stage_nbig[t]++;
                    P.stage_bigm[t] = std::max(P.stage_bigm[t], fh.m); P.stage_bigk[t] = std::max(P.stage_bigk[t], fh.k);
                } else {
                    P.stage_smallm[t] = std::max(P.stage_smallm[t], fh.m); P.stage_smallk[t] = std::max(P.stage_smallk[t], fh.k);
                }
            }
        }
        P.flevptr.assign((size_t)P.nblev + 1, 0);
        for (int64_t p = 0; p < n; p++) P.flevptr[Y.blev[(size_t)pblk[p]] + 1]++;
        for (int32_t l = 0; l < P.nblev; l++) P.flevptr[l + 1] += P.flevptr[l];
        P.flevpos.resize((size_t)n);
        std::vector<int64_t> cur2(P.flevptr.begin(), P.flevptr.end() - 1);
        for (int64_t p = 0; p < n; p++) P.flevpos[(size_t)cur2[Y.blev[(size_t)pblk[p]]]++] = (int32_t)p;
    }
    // ---- scatter map of the caller's entries (diagonal blocks) and the off-diagonal part F ------------------------
    std::vector<int32_t> posrow((size_t)n), poscol((size_t)n);
    for (int64_t p = 0; p < n; p++) { posrow[P.prow[p]] = (int32_t)p; poscol[P.qcol[p]] = (int32_t)p; }
    P.aptr.assign((size_t)nf + 1, 0);
    P.fptr_r.assign((size_t)n + 1, 0);
    P.fptr_c.assign((size_t)n + 1, 0);
    std::vector<int32_t> ef((size_t)Y.nnz);
    int64_t nF = 0;
    for (int64_t j = 0; j < n; j++)
        for (int64_t p = Y.Ap[j]; p < Y.Ap[j + 1]; p++) {
            const int32_t pr = posrow[Y.Ai[p]], pc = poscol[j];
            if (pblk[pr] != pblk[pc]) {
                if (pblk[pr] > pblk[pc]) throw std::runtime_error("internal: entry below the block diagonal");
                ef[p] = -1; nF++;
                P.fptr_r[pr + 1]++; P.fptr_c[pc + 1]++;
                continue;
            }
            const int32_t f = col2front[std::min(pr, pc)];
            ef[p] = f;
            P.aptr[f + 1]++;
        }
    for (int64_t f = 0; f < nf; f++) P.aptr[f + 1] += P.aptr[f];
    for (int64_t p = 0; p < n; p++) { P.fptr_r[p + 1] += P.fptr_r[p]; P.fptr_c[p + 1] += P.fptr_c[p]; }
    P.a_src.resize((size_t)(Y.nnz - nF));
    P.a_dst.resize((size_t)(Y.nnz - nF));
    P.fcol.resize((size_t)nF); P.frow.resize((size_t)nF); P.fsrc_r.resize((size_t)nF); P.fsrc_c.resize((size_t)nF);
    {
        std::vector<int64_t> cur(P.aptr.begin(), P.aptr.end() - 1), cr(P.fptr_r.begin(), P.fptr_r.end() - 1), cc(P.fptr_c.begin(), P.fptr_c.end() - 1);
        for (int64_t j = 0; j < n; j++)
            for (int64_t p = Y.Ap[j]; p < Y.Ap[j + 1]; p++) {
                const int32_t f = ef[p];
                const int32_t pr = posrow[Y.Ai[p]], pc = poscol[j];
                if (f < 0) {
                    const int64_t a1 = cr[pr]++, a2 = cc[pc]++;
                    P.fcol[(size_t)a1] = pc; P.fsrc_r[(size_t)a1] = p;
                    P.frow[(size_t)a2] = pr; P.fsrc_c[(size_t)a2] = p;
                    continue;
                }
                const int32_t lr = local_index(f, pr), lc = local_index(f, pc);
                if (lr < 0 || lc < 0) throw std::runtime_error("internal: entry outside the front structure");
                const int64_t q = cur[f]++;
                P.a_src[q] = p;
                P.a_dst[q] = lr + lc * P.fr[f].m;
            }
        // per front by ascending destination: the blocked assembly (k_lub_assemble) bisects the list for its columns instead of
        // scanning it in every workgroup (destinations are distinct: the order is free)
        std::vector<std::pair<int32_t, int64_t>> tmp;
        for (int64_t f = 0; f < nf; f++) {
            const int64_t a = P.aptr[f], b = P.aptr[f + 1];
            if (b - a < 2) continue;
            tmp.resize((size_t)(b - a));
            for (int64_t q = a; q < b; q++) tmp[(size_t)(q - a)] = {P.a_dst[(size_t)q], P.a_src[(size_t)q]};
            std::sort(tmp.begin(), tmp.end());
            for (int64_t q = a; q < b; q++) { P.a_dst[(size_t)q] = tmp[(size_t)(q - a)].first; P.a_src[(size_t)q] = tmp[(size_t)(q - a)].second; }
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
