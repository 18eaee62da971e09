// Symbolic analysis (host): see symbolic.hpp.  Reference role: cholmod_l_analyze_p as
// called from src/C/cholmod.c:274 after pack() (:132-181).
#include "symbolic.hpp"
#include "par.hpp"

#include <algorithm>
#include <string>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <stdexcept>

namespace kvx {
namespace {

// Pattern of the strictly-lower part of C = P A P' by column (rows sorted), built from
// the caller's `uplo` triangle.  Also returns, for every caller entry, (row, col) of the
// permuted lower-triangular position (or col = -1 when the entry is not read).
struct LowerPattern {
    std::vector<int64_t> ptr;
    std::vector<int32_t> idx;
};

void build_lower(int64_t n, const int64_t *Ap, const int64_t *Ai, int uplo,
                 const std::vector<int64_t> &iperm, LowerPattern &Lo)
{
    // threads scan disjoint ranges of the caller's columns and claim slots of the destination columns with atomic counters;
    // the order inside a column depends on the race, the sort at the end makes the result canonical
    const int T = analyze_threads();
    Lo.ptr.assign((size_t)n + 1, 0);
    int64_t *cnt = Lo.ptr.data() + 1;
    parallel_for(n, T, 1 << 15, [&](int64_t lo, int64_t hi) {
        for (int64_t j = lo; j < hi; j++)
            for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
                int64_t i = Ai[p];
                if (i == j) continue;
                if ((uplo == 'L' && i < j) || (uplo != 'L' && i > j)) continue;
                int64_t a = iperm[i], b = iperm[j];
                __atomic_fetch_add(&cnt[std::min(a, b)], (int64_t)1, __ATOMIC_RELAXED);
            }
    });
    for (int64_t j = 0; j < n; j++) Lo.ptr[j + 1] += Lo.ptr[j];
    Lo.idx.resize((size_t)Lo.ptr[n]);
    std::vector<int64_t> cur(Lo.ptr.begin(), Lo.ptr.end() - 1);
    parallel_for(n, T, 1 << 15, [&](int64_t lo, int64_t hi) {
        for (int64_t j = lo; j < hi; j++)
            for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
                int64_t i = Ai[p];
                if (i == j) continue;
                if ((uplo == 'L' && i < j) || (uplo != 'L' && i > j)) continue;
                int64_t a = iperm[i], b = iperm[j];
                const int64_t slot = __atomic_fetch_add(&cur[(size_t)std::min(a, b)], (int64_t)1, __ATOMIC_RELAXED);
                Lo.idx[(size_t)slot] = (int32_t)std::max(a, b);
            }
    });
    parallel_for(n, T, 1 << 15, [&](int64_t lo, int64_t hi) {
        for (int64_t j = lo; j < hi; j++) std::sort(Lo.idx.begin() + Lo.ptr[j], Lo.idx.begin() + Lo.ptr[j + 1]);
    });
}

// Elimination tree from the strictly-lower pattern by column.  Liu's algorithm needs,
// for each k, the entries (k, i) with i < k, i.e. rows of the lower pattern: process
// column-by-column through a transposed sweep.
void etree_from_lower(int64_t n, const LowerPattern &Lo, std::vector<int32_t> &parent)
{
    // build row lists: for row k, the columns i < k with C(k,i) != 0
    std::vector<int64_t> rptr((size_t)n + 1, 0);
    for (int64_t p = 0; p < (int64_t)Lo.idx.size(); p++) rptr[(size_t)Lo.idx[p] + 1]++;
    for (int64_t k = 0; k < n; k++) rptr[k + 1] += rptr[k];
    std::vector<int32_t> ridx(Lo.idx.size());
    std::vector<int64_t> cur(rptr.begin(), rptr.end() - 1);
    for (int64_t j = 0; j < n; j++)
        for (int64_t p = Lo.ptr[j]; p < Lo.ptr[j + 1]; p++) ridx[(size_t)cur[Lo.idx[p]]++] = (int32_t)j;
    parent.assign((size_t)n, -1);
    std::vector<int32_t> anc((size_t)n, -1);
    for (int64_t k = 0; k < n; k++)
        for (int64_t p = rptr[k]; p < rptr[k + 1]; p++) {
            int32_t i = ridx[p];
            while (i != -1 && i < k) {
                int32_t nxt = anc[i];
                anc[i] = (int32_t)k;
                if (nxt == -1) parent[i] = (int32_t)k;
                i = nxt;
            }
        }
}

// The same tree from the full symmetric adjacency (original labels) and the permutation: no permuted pattern is formed.
// closed (optional): disjoint position ranges whose vertices have no earlier neighbour outside their range (the subdomains a
// nested dissection finished as a whole).  Liu's algorithm on the rows of such a range touches the range only, so the ranges
// run on different threads; the rows outside them (the separators) follow on the calling thread in ascending order -- each of
// them finds every smaller row done, which is all the algorithm asks for.
void etree_from_adjacency(int64_t n, const std::vector<int64_t> &aptr, const std::vector<int32_t> &adj,
                          const std::vector<int64_t> &perm, const std::vector<int64_t> &iperm, std::vector<int32_t> &parent,
                          const std::vector<std::pair<int64_t, int64_t>> *closed = nullptr)
{
    parent.assign((size_t)n, -1);
    std::vector<int32_t> anc((size_t)n, -1), ip32((size_t)n);
    for (int64_t v = 0; v < n; v++) ip32[(size_t)v] = (int32_t)iperm[(size_t)v];
    auto rows = [&](int64_t k0, int64_t k1) {
        for (int64_t k = k0; k < k1; k++) {
            const int64_t v = perm[(size_t)k];
            for (int64_t p = aptr[(size_t)v]; p < aptr[(size_t)v + 1]; p++) {
                int32_t i = ip32[(size_t)adj[(size_t)p]];
                while (i != -1 && i < k) {
                    const int32_t nxt = anc[i];
                    anc[i] = (int32_t)k;
                    if (nxt == -1) parent[i] = (int32_t)k;
                    i = nxt;
                }
            }
        }
    };
    const int T = analyze_threads();
    if (!closed || closed->size() < 2 || T <= 1) { rows(0, n); return; }
    const std::vector<std::pair<int64_t, int64_t>> &R = *closed;
    for (size_t r = 0; r < R.size(); r++)          // (sorted, disjoint, inside [0, n): anything else and the plain pass runs)
        if (R[r].first < 0 || R[r].second > n || R[r].first >= R[r].second || (r > 0 && R[r].first < R[r - 1].second)) { rows(0, n); return; }
    parallel_for((int64_t)R.size(), T, 1, [&](int64_t a, int64_t b) {
        for (int64_t r = a; r < b; r++) rows(R[(size_t)r].first, R[(size_t)r].second);
    });
    int64_t k = 0;
    for (size_t r = 0; r <= R.size(); r++) {
        const int64_t stop = r < R.size() ? R[r].first : n;
        rows(k, stop);
        if (r < R.size()) k = R[r].second;
    }
}

// Postorder of a forest given by parent[] (parent[j] > j or -1).  Children are visited
// in increasing subtree size so that the heaviest child ends adjacent to its parent
// (it is the amalgamation candidate).
void postorder(int64_t n, const std::vector<int32_t> &parent, std::vector<int32_t> &post)
{
    std::vector<int64_t> size((size_t)n, 1);
    for (int64_t j = 0; j < n; j++)
        if (parent[j] >= 0) size[parent[j]] += size[j];
    // child lists by increasing size (ties: larger index first): the children of every node are collected in index order (two counting
    // passes) and each list is sorted on its own -- lists are short, a global sort of all n nodes by size is not needed
    std::vector<int64_t> cptr((size_t)n + 2, 0);
    for (int64_t j = 0; j < n; j++) cptr[(size_t)(parent[j] >= 0 ? parent[j] : n) + 1]++;
    for (int64_t j = 0; j <= n; j++) cptr[j + 1] += cptr[j];
    std::vector<int32_t> clist((size_t)n);
    {
        std::vector<int64_t> cur(cptr.begin(), cptr.end() - 1);
        for (int64_t j = 0; j < n; j++) clist[(size_t)cur[parent[j] >= 0 ? parent[j] : n]++] = (int32_t)j;
    }
    parallel_for(n + 1, analyze_threads(), 1 << 15, [&](int64_t lo, int64_t hi) {
        for (int64_t v = lo; v < hi; v++)
            if (cptr[v + 1] - cptr[v] > 1)
                std::sort(clist.begin() + cptr[v], clist.begin() + cptr[v + 1],
                          [&](int32_t a, int32_t b) { return size[a] != size[b] ? size[a] < size[b] : a > b; });
    });
    std::vector<int32_t> head((size_t)n, -1), next((size_t)n, -1);
    int32_t root_list = -1;
    for (int64_t v = 0; v <= n; v++)
        for (int64_t q = cptr[v + 1] - 1; q >= cptr[v]; q--) {       // push back to front: list heads are the smallest
            const int32_t j = clist[q];
            if (v < n) { next[j] = head[v]; head[v] = j; }
            else { next[j] = root_list; root_list = j; }
        }
    post.resize((size_t)n);
    int64_t k = 0;
    std::vector<int32_t> stack;
    for (int32_t r = root_list; r != -1; r = next[r]) {
        stack.push_back(r);
        while (!stack.empty()) {
            int32_t v = stack.back();
            int32_t c = head[v];
            if (c == -1) { post[(size_t)k++] = v; stack.pop_back(); }
            else { head[v] = next[c]; stack.push_back(c); }
        }
    }
}

// Column counts of L (Gilbert, Ng, Peyton 1994: skeleton-matrix / least-common-ancestor
// algorithm) for a matrix whose elimination tree is already postordered (parent[j] > j,
// subtrees contiguous).
void column_counts(int64_t n, const LowerPattern &Lo, const std::vector<int32_t> &parent,
                   std::vector<int32_t> &cc)
{
    std::vector<int64_t> delta((size_t)n, 0);
    std::vector<int32_t> first((size_t)n), maxfirst((size_t)n, -1), prevleaf((size_t)n, -1), anc((size_t)n);
    std::vector<int64_t> size((size_t)n, 1);
    std::vector<char> haschild((size_t)n, 0);
    for (int64_t j = 0; j < n; j++)
        if (parent[j] >= 0) { size[parent[j]] += size[j]; haschild[parent[j]] = 1; }
    for (int64_t j = 0; j < n; j++) {
        first[j] = (int32_t)(j - size[j] + 1);
        delta[j] = haschild[j] ? 0 : 1;
        anc[j] = (int32_t)j;
    }
    for (int64_t j = 0; j < n; j++) {
        if (parent[j] >= 0) delta[parent[j]]--;
        for (int64_t p = Lo.ptr[j]; p < Lo.ptr[j + 1]; p++) {
            int32_t i = Lo.idx[p];
            if (first[j] <= maxfirst[i]) continue;   // j is not a leaf of row subtree i
            maxfirst[i] = first[j];
            int32_t jprev = prevleaf[i];
            prevleaf[i] = (int32_t)j;
            delta[j]++;
            if (jprev != -1) {
                int32_t q = jprev;
                while (q != anc[q]) q = anc[q];
                for (int32_t s = jprev; s != q;) { int32_t sp = anc[s]; anc[s] = q; s = sp; }
                delta[q]--;
            }
        }
        if (parent[j] >= 0) anc[j] = parent[j];
    }
    cc.resize((size_t)n);
    for (int64_t j = 0; j < n; j++)
        if (parent[j] >= 0) delta[parent[j]] += delta[j];
    for (int64_t j = 0; j < n; j++) cc[j] = (int32_t)delta[j];
}

}  // namespace

namespace {
struct PhaseTimer {                                  // KVX_ANALYZE_TIMING=1: wall time of every phase of analyze() on stderr
    bool on;
    std::chrono::steady_clock::time_point t;
    PhaseTimer() : on(getenv("KVX_ANALYZE_TIMING") != nullptr), t(std::chrono::steady_clock::now()) {}
    void lap(const char *what)
    {
        if (!on) return;
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "  analyze %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t).count());
        t = t1;
    }
};
}  // namespace

void analyze(int64_t n, const int64_t *Ap, const int64_t *Ai, int uplo, const int64_t *user_perm,
             const SymOpts &opts, Symbolic &S)
{
    if (n < 0) throw std::runtime_error("negative dimension");
    if (n >= (int64_t)1 << 31) throw std::runtime_error("order exceeds 2^31-1");
    if (uplo != 'L' && uplo != 'U') throw std::runtime_error("uplo must be 'L' or 'U'");
    PhaseTimer pt;
    S = Symbolic();
    S.n = n;
    S.nnzA = n ? Ap[n] : 0;
    for (int64_t j = 0; j < n; j++) {
        if (Ap[j + 1] < Ap[j]) throw std::runtime_error("colptr not monotone");
        for (int64_t p = Ap[j]; p < Ap[j + 1]; p++)
            if (Ai[p] < 0 || Ai[p] >= n) throw std::runtime_error("row index out of range");
    }

    // ---- 1. initial permutation: candidates (user's, nested dissection, minimum degree), the least fill wins ----------
    std::vector<std::vector<int64_t>> cand;
    std::vector<const char *> cand_name;
    // fill of an ordering: nnz(L) from the column counts (elimination tree + postorder + Gilbert-Ng-Peyton), no factor formed
    auto fill_of = [&](const std::vector<int64_t> &pp) -> double {
        std::vector<int64_t> ip((size_t)n);
        for (int64_t k = 0; k < n; k++) ip[pp[k]] = k;
        LowerPattern Lc;
        std::vector<int32_t> par, post, cc;
        build_lower(n, Ap, Ai, uplo, ip, Lc);
        etree_from_lower(n, Lc, par);
        postorder(n, par, post);
        std::vector<int32_t> ipost((size_t)n), np2((size_t)n);
        for (int64_t k = 0; k < n; k++) ipost[post[k]] = (int32_t)k;
        for (int64_t k = 0; k < n; k++) np2[k] = par[post[k]] >= 0 ? ipost[par[post[k]]] : -1;
        for (int64_t k = 0; k < n; k++) ip[pp[post[k]]] = k;
        build_lower(n, Ap, Ai, uplo, ip, Lc);
        column_counts(n, Lc, np2, cc);
        double lnz = 0.0;
        for (int64_t j = 0; j < n; j++) lnz += cc[j];
        return lnz;
    };
    std::vector<double> known_fill;        // fill of cand[i] when it was needed to decide what else to compute (else < 0)
    std::vector<std::pair<int64_t, int64_t>> nd_closed;   // closed subdomains of the dissection candidate (order_nd)
    std::vector<int64_t> aptr;             // full symmetric adjacency (built when an ordering is computed here; phase 2 reuses it)
    std::vector<int32_t> adj;
    if (user_perm) {
        std::vector<int64_t> pu((size_t)n);
        std::vector<char> seen((size_t)n, 0);
        for (int64_t k = 0; k < n; k++) {
            int64_t q = user_perm[k];
            if (q < 0 || q >= n || seen[q]) throw std::invalid_argument("p is not a valid permutation");
            seen[q] = 1;
            pu[k] = q;
        }
        cand.push_back(std::move(pu)); cand_name.push_back("given"); known_fill.push_back(-1.0);
    }
    if (!user_perm || opts.compare_given) {
        if (opts.ordering == 1) {
            if (!user_perm) { cand.emplace_back((size_t)n); std::iota(cand.back().begin(), cand.back().end(), 0); cand_name.push_back("natural"); known_fill.push_back(-1.0); }
        } else if (n > 0) {
            // full symmetric adjacency of the analysed triangle
            aptr.assign((size_t)n + 1, 0);
            for (int64_t j = 0; j < n; j++)
                for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
                    int64_t i = Ai[p];
                    if (i == j || (uplo == 'L' && i < j) || (uplo != 'L' && i > j)) continue;
                    aptr[i + 1]++;
                    aptr[j + 1]++;
                }
            for (int64_t j = 0; j < n; j++) aptr[j + 1] += aptr[j];
            adj.resize((size_t)aptr[n]);
            std::vector<int64_t> cur(aptr.begin(), aptr.end() - 1);
            for (int64_t j = 0; j < n; j++)
                for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
                    int64_t i = Ai[p];
                    if (i == j || (uplo == 'L' && i < j) || (uplo != 'L' && i > j)) continue;
                    adj[(size_t)cur[i]++] = (int32_t)j;
                    adj[(size_t)cur[j]++] = (int32_t)i;
                }
            pt.lap("1a adjacency");
            // the two orderings are independent: the minimum-degree candidate runs on a thread of its own beside the dissection
            // (an exception of either is carried to the caller; the result does not depend on the overlap)
            const bool want_md = opts.ordering == 3 || (opts.ordering == 0 && n <= opts.amd_auto_max);
            bool want_nd = opts.ordering != 3;
            // Small systems (n < nd_min_n), ordering 0: the minimum-degree ordering first; when it leaves next to no fill
            // (nnz(L) <= 3 nnz(tril A): power grids, circuits -- ACTIVSg2000: 43 456 against the dissection's 165 132) the dissection
            // is not computed at all (it costs as much as the rest of the analysis: 3.0 -> 1.5 ms there).  Meshes fill more than that
            // even when tiny (40 x 40 grid: 4.4 nnz) and get both candidates as before.
            const bool md_first = opts.ordering == 0 && want_md && n < opts.nd_min_n;
            std::vector<int64_t> perm_nd, perm_md;
            std::exception_ptr md_err;
            auto run_md = [&] {
                try {
                    // (duplicate entries of the caller's pattern would show up as repeated neighbours: harmless for the
                    // dissection, but the quotient graph wants clean lists)
                    std::vector<int64_t> aptr2((size_t)n + 1, 0);
                    std::vector<int32_t> adj2;
                    adj2.reserve(adj.size());
                    for (int64_t i = 0; i < n; i++) {
                        const size_t b0 = adj2.size();
                        adj2.insert(adj2.end(), adj.begin() + aptr[i], adj.begin() + aptr[i + 1]);
                        std::sort(adj2.begin() + b0, adj2.end());
                        adj2.erase(std::unique(adj2.begin() + b0, adj2.end()), adj2.end());
                        aptr2[i + 1] = (int64_t)adj2.size();
                    }
                    order_amd(n, aptr2, adj2, perm_md);
                } catch (...) { md_err = std::current_exception(); }
            };
            std::thread md_thread;
            bool md_async = false;
            double md_fill = -1.0;
            if (md_first) {
                run_md();
                if (md_err) std::rethrow_exception(md_err);
                md_fill = fill_of(perm_md);
                int64_t tri = n;                     // entries of the analysed triangle, diagonal counted once per column
                for (int64_t v = 0; v < n; v++) tri += (aptr[(size_t)v + 1] - aptr[(size_t)v]);
                tri = (tri - n) / 2 + n;
                if (md_fill <= 3.0 * (double)tri) want_nd = false;
            }
            if (!md_first && want_md && want_nd && analyze_threads() > 1) {
                try { md_thread = std::thread(run_md); md_async = true; } catch (...) { md_async = false; }
            }
            try {
                if (want_nd) order_nd(n, aptr, adj, opts.nd_leaf, perm_nd, &nd_closed);
            } catch (...) {
                if (md_async) md_thread.join();
                throw;
            }
            if (md_async) md_thread.join();
            else if (want_md && !md_first) run_md();
            if (md_err) std::rethrow_exception(md_err);
            if (want_nd) { cand.push_back(std::move(perm_nd)); cand_name.push_back("nested dissection"); known_fill.push_back(-1.0); }
            if (want_md) { cand.push_back(std::move(perm_md)); cand_name.push_back("minimum degree"); known_fill.push_back(md_fill); }
        } else if (!user_perm) {
            cand.emplace_back(); cand_name.push_back("empty"); known_fill.push_back(-1.0);
        }
    }
    size_t pick = 0;
    if (cand.size() > 1) {
        // the candidates are evaluated side by side (independent work on private vectors)
        std::vector<double> fill(cand.size(), 0.0);
        std::vector<std::exception_ptr> ferr(cand.size());
        known_fill.resize(cand.size(), -1.0);
        auto eval = [&](size_t c) {
            try {
                fill[c] = known_fill[c] >= 0.0 ? known_fill[c] : fill_of(cand[c]);
            } catch (...) { ferr[c] = std::current_exception(); }
        };
        {
            std::vector<std::thread> th;
            if (analyze_threads() > 1) {
                try {
                    for (size_t c = 1; c < cand.size(); c++) th.emplace_back(eval, c);
                } catch (...) { }
            }
            const size_t started = th.size();
            eval(0);
            for (size_t c = 1 + started; c < cand.size(); c++) eval(c);
            for (auto &t : th) t.join();
        }
        for (auto &e : ferr)
            if (e) std::rethrow_exception(e);
        double best = 0.0;
        for (size_t c = 0; c < cand.size(); c++) {
            if (pt.on) fprintf(stderr, "  analyze candidate %-18s nnz(L) = %.0f\n", cand_name[c], fill[c]);
            if (c == 0 || fill[c] < best) { best = fill[c]; pick = c; }
        }
    }
    std::vector<int64_t> perm0((size_t)n), iperm0((size_t)n, -1);
    if (!cand.empty() && (int64_t)cand[pick].size() == n) perm0 = cand[pick];
    else std::iota(perm0.begin(), perm0.end(), 0);
    for (int64_t k = 0; k < n; k++) iperm0[perm0[k]] = k;

    pt.lap("1 ordering");
    // ---- 2. etree + postorder, fold the postorder into the permutation ---------------------
    LowerPattern Lo;
    std::vector<int32_t> parent;
    bool lower_built = false;
    if (!adj.empty() || (n > 0 && !aptr.empty() && aptr[n] == 0)) {
        // the adjacency of the ordering phase is at hand: Liu's algorithm straight from it (row k of the permuted matrix = the
        // neighbours of perm0[k] that come earlier), no permuted pattern and no row lists for this pass
        const bool nd_picked = !cand.empty() && pick < cand_name.size() && std::string(cand_name[pick]) == "nested dissection";
        etree_from_adjacency(n, aptr, adj, perm0, iperm0, parent, nd_picked ? &nd_closed : nullptr);
    } else {
        build_lower(n, Ap, Ai, uplo, iperm0, Lo);
        etree_from_lower(n, Lo, parent);
        lower_built = true;
    }
    pt.lap("2a etree");
    S.perm = perm0;
    S.iperm = iperm0;
    // Supernodes need contiguous subtrees, so the postorder is always applied (CHOLMOD's
    // supernodal analysis does the same whatever options['postorder'] says).
    {
        std::vector<int32_t> post;
        postorder(n, parent, post);
        pt.lap("2b postorder");
        bool identity = true;
        for (int64_t k = 0; k < n && identity; k++) identity = (post[k] == k);
        if (!identity) {
            for (int64_t k = 0; k < n; k++) S.perm[k] = perm0[post[k]];
            for (int64_t k = 0; k < n; k++) S.iperm[S.perm[k]] = k;
            pt.lap("2c compose");
            build_lower(n, Ap, Ai, uplo, S.iperm, Lo);
            pt.lap("2d permuted pattern");
            // the elimination tree of the relabelled matrix is the relabelled tree: no second pass of Liu's algorithm
            std::vector<int32_t> ipost((size_t)n), np((size_t)n);
            for (int64_t k = 0; k < n; k++) ipost[post[k]] = (int32_t)k;
            for (int64_t k = 0; k < n; k++) np[k] = parent[post[k]] >= 0 ? ipost[parent[post[k]]] : -1;
            parent.swap(np);
            lower_built = true;
        }
    }
    if (!lower_built) build_lower(n, Ap, Ai, uplo, S.iperm, Lo);
    { std::vector<int64_t>().swap(aptr); std::vector<int32_t>().swap(adj); }

    pt.lap("2 etree+postorder");
    // ---- 3. column counts ---------------------------------------------------------------
    column_counts(n, Lo, parent, S.colcount);
    S.lnz = 0;
    S.flops = 0;
    for (int64_t j = 0; j < n; j++) {
        S.lnz += S.colcount[j];
        S.flops += (double)S.colcount[j] * (double)S.colcount[j];
    }

    pt.lap("3 colcounts");
    // ---- 4. supernodes: maximal zero-fill chains, then relaxed amalgamation ----------------
    std::vector<int64_t> sstart;           // first column of each supernode
    for (int64_t j = 0; j < n; j++) {
        bool join = j > 0 && parent[j - 1] == j && S.colcount[j - 1] == S.colcount[j] + 1;
        if (!join) sstart.push_back(j);
    }
    int64_t ns0 = (int64_t)sstart.size();
    sstart.push_back(n);
    std::vector<int32_t> col2s((size_t)n);
    for (int64_t s = 0; s < ns0; s++)
        for (int64_t j = sstart[s]; j < sstart[s + 1]; j++) col2s[j] = (int32_t)s;
    std::vector<int32_t> sp0((size_t)ns0, -1);
    for (int64_t s = 0; s < ns0; s++) {
        int64_t last = sstart[s + 1] - 1;
        if (parent[last] >= 0) sp0[s] = col2s[parent[last]];
    }
    // relaxed amalgamation: merge s into its parent when s is the parent's last child
    // (adjacent columns) and the explicit zeros introduced stay below the bounds.
    std::vector<int64_t> ncol((size_t)ns0), nrow((size_t)ns0);
    std::vector<double> zeros((size_t)ns0, 0.0);
    std::vector<char> dead((size_t)ns0, 0);
    for (int64_t s = 0; s < ns0; s++) {
        ncol[s] = sstart[s + 1] - sstart[s];
        nrow[s] = S.colcount[sstart[s]];
    }
    std::vector<int64_t> mstart(sstart.begin(), sstart.end() - 1);  // merged first column
    // (a) leaf-subtree amalgamation: a supernode whose children are all leaves of the (merged)
    // tree is fused with ALL of them into one dense leaf front when the fused front stays small.
    // A GPU pays a fixed price per front (a workgroup, dependent index loads, a launch slot);
    // the minimum-degree leaves of a 2-D/3-D mesh otherwise produce hundreds of thousands of
    // fronts of order < 16.
    if (opts.leaf_cols > 0) {
        std::vector<char> haskids((size_t)ns0, 0), allleaf((size_t)ns0, 1);
        std::vector<int64_t> kidcols((size_t)ns0, 0), firstkid((size_t)ns0, -1);
        for (int64_t s = 0; s < ns0; s++) {      // postorder: the children of s were decided before s
            bool leaf = !haskids[s];
            if (haskids[s] && allleaf[s]) {
                const int64_t ktot = ncol[s] + kidcols[s];
                const int64_t u = nrow[s] - ncol[s];
                if (ktot <= opts.leaf_cols && ktot + u <= opts.leaf_rows) {
                    const int64_t fk = firstkid[s];
                    for (int64_t c = fk; c < s; c++) dead[c] = 1;
                    const double dense = (double)ktot * (double)(ktot + u) - (double)ktot * (double)(ktot - 1) / 2;
                    double have = 0;
                    for (int64_t j = mstart[fk]; j < sstart[s + 1]; j++) have += S.colcount[j];
                    zeros[s] = dense - have;
                    ncol[s] = ktot;
                    nrow[s] = ktot + u;
                    mstart[s] = mstart[fk];
                    leaf = true;                 // fused: a leaf of the merged tree
                }
            }
            const int64_t p = sp0[s];
            if (p >= 0) {
                if (!haskids[p]) { haskids[p] = 1; firstkid[p] = s; }
                if (!leaf) allleaf[p] = 0;
                kidcols[p] += ncol[s];
            }
        }
    }
    // (b) chain amalgamation along last-child edges
    for (int64_t s = 0; s + 1 < ns0; s++) {
        int64_t p = sp0[s];
        if (dead[s] || p < 0 || sstart[s + 1] != mstart[p]) continue;
        double nsc = (double)ncol[s], npc = (double)ncol[p];
        double mp = (double)nrow[p], ms = (double)nrow[s];
        double newz = nsc * (nsc + mp - ms);
        double z = zeros[s] + zeros[p] + newz;
        double tot = nsc + npc, mm = nsc + mp;
        double lsz = tot * mm - tot * (tot - 1) / 2;
        bool merge;
        if (tot <= opts.relax_small) merge = true;
        else {
            double frac = z / lsz;
            merge = (tot <= 16 && frac < opts.relax_z1) || (tot <= 48 && frac < opts.relax_z2) || frac < opts.relax_z3;
            if (newz == 0) merge = true;
        }
        if (!merge) continue;
        dead[s] = 1;
        ncol[p] += ncol[s];
        nrow[p] = (int64_t)mm;
        zeros[p] = z;
        mstart[p] = mstart[s];
    }
    S.super.clear();
    std::vector<int32_t> old2new((size_t)ns0, -1);
    for (int64_t s = 0; s < ns0; s++)
        if (!dead[s]) { old2new[s] = (int32_t)S.super.size(); S.super.push_back(mstart[s]); }
    S.nsuper = (int64_t)S.super.size();
    S.super.push_back(n);
    for (int64_t s = ns0 - 1; s >= 0; s--)      // dead supernodes map to the survivor above them
        if (dead[s]) old2new[s] = old2new[s + 1];
    for (int64_t j = 0; j < n; j++) col2s[j] = old2new[col2s[j]];
    const int64_t ns = S.nsuper;
    S.sn_k.resize((size_t)ns);
    S.sparent.assign((size_t)ns, -1);
    for (int64_t s = 0; s < ns; s++) {
        S.sn_k[s] = (int32_t)(S.super[s + 1] - S.super[s]);
        int64_t last = S.super[s + 1] - 1;
        if (parent[last] >= 0) S.sparent[s] = col2s[parent[last]];
    }
    // children lists
    S.childptr.assign((size_t)ns + 1, 0);
    for (int64_t s = 0; s < ns; s++)
        if (S.sparent[s] >= 0) S.childptr[S.sparent[s] + 1]++;
    for (int64_t s = 0; s < ns; s++) S.childptr[s + 1] += S.childptr[s];
    S.children.resize((size_t)S.childptr[ns]);
    {
        std::vector<int64_t> cur(S.childptr.begin(), S.childptr.end() - 1);
        for (int64_t s = 0; s < ns; s++)
            if (S.sparent[s] >= 0) S.children[(size_t)cur[S.sparent[s]]++] = (int32_t)s;
    }

    pt.lap("4 supernodes");
    // ---- 5. front row structures (supernodal symbolic factorisation) -------------------------
    // The order of every front is known from the amalgamation (a merged front = its pivots + the rows of the supernode it was
    // merged into), so the row lists have their final places before they are computed: independent subtrees of the supernodal tree
    // are filled in by different threads (each front needs its own columns and its children's lists only), the top of the tree
    // after them on the calling thread.  A front whose list does not come out at the predicted length (it cannot) sends the
    // whole phase back to the plain serial pass.
    S.rowptr.assign((size_t)ns + 1, 0);
    S.sn_m.resize((size_t)ns);
    {
        for (int64_t s0 = 0; s0 < ns0; s0++)
            if (!dead[s0]) S.sn_m[(size_t)old2new[s0]] = (int32_t)nrow[s0];
        for (int64_t s = 0; s < ns; s++) S.rowptr[s + 1] = S.rowptr[s] + S.sn_m[s];
        S.rowidx.assign((size_t)S.rowptr[ns], 0);
        std::atomic<bool> mismatch{false};
        auto fill = [&](int64_t s, std::vector<int32_t> &mark, std::vector<int32_t> &tail) {
            const int64_t f = S.super[s], e = S.super[s + 1];
            tail.clear();
            for (int64_t j = f; j < e; j++)
                for (int64_t p = Lo.ptr[j]; p < Lo.ptr[j + 1]; p++) {
                    const int32_t i = Lo.idx[p];
                    if (i >= e && mark[i] != s) { mark[i] = (int32_t)s; tail.push_back(i); }
                }
            for (int64_t c = S.childptr[s]; c < S.childptr[s + 1]; c++) {
                const int32_t ch = S.children[c];
                for (int64_t p = S.rowptr[ch] + S.sn_k[ch]; p < S.rowptr[ch + 1]; p++) {
                    const int32_t i = S.rowidx[p];
                    if (i >= e && mark[i] != s) { mark[i] = (int32_t)s; tail.push_back(i); }
                }
            }
            if ((int64_t)tail.size() + (e - f) != S.sn_m[s]) { mismatch.store(true); return; }
            std::sort(tail.begin(), tail.end());
            int32_t *dst = S.rowidx.data() + S.rowptr[s];
            for (int64_t j = f; j < e; j++) *dst++ = (int32_t)j;
            std::copy(tail.begin(), tail.end(), dst);
        };
        // tasks: maximal subtrees of at most `thr` columns (a subtree is a contiguous range of the postordered numbering)
        const int T = analyze_threads();
        std::vector<int64_t> scols((size_t)ns), scnt((size_t)ns, 1);
        for (int64_t s = 0; s < ns; s++) scols[s] = S.sn_k[s];
        for (int64_t s = 0; s < ns; s++)
            if (S.sparent[s] >= 0) { scols[S.sparent[s]] += scols[s]; scnt[S.sparent[s]] += scnt[s]; }
        const int64_t thr = std::max<int64_t>(n / (8 * (int64_t)std::max(T, 1)), 2048);
        std::vector<int64_t> task_lo, task_hi;
        std::vector<char> in_task((size_t)ns, 0);
        if (T > 1 && n >= 8 * thr)
            for (int64_t s = 0; s < ns; s++)
                if (scols[s] <= thr && (S.sparent[s] < 0 || scols[S.sparent[s]] > thr)) {
                    task_lo.push_back(s - scnt[s] + 1);
                    task_hi.push_back(s);
                }
        for (size_t t = 0; t < task_lo.size(); t++)
            for (int64_t s = task_lo[t]; s <= task_hi[t]; s++) in_task[s] = 1;
        parallel_for((int64_t)task_lo.size(), T, 1, [&](int64_t a, int64_t b) {
            std::vector<int32_t> mark((size_t)n, -1), tail;
            for (int64_t t = a; t < b && !mismatch.load(); t++)
                for (int64_t s = task_lo[t]; s <= task_hi[t]; s++) fill(s, mark, tail);
        });
        {
            std::vector<int32_t> mark((size_t)n, -1), tail;
            for (int64_t s = 0; s < ns && !mismatch.load(); s++)
                if (!in_task[s]) fill(s, mark, tail);
        }
        if (mismatch.load()) {                     // (never observed: the plain pass, lists appended as they are found)
            if (pt.on) fprintf(stderr, "  analyze 5: predicted front orders did not hold -- serial pass\n");
            S.rowidx.clear();
            S.rowidx.reserve((size_t)(4 * n));
            std::vector<int32_t> mark((size_t)n, -1), tail;
            for (int64_t s = 0; s < ns; s++) {
                const int64_t f = S.super[s], e = S.super[s + 1];
                tail.clear();
                for (int64_t j = f; j < e; j++)
                    for (int64_t p = Lo.ptr[j]; p < Lo.ptr[j + 1]; p++) {
                        const int32_t i = Lo.idx[p];
                        if (i >= e && mark[i] != s) { mark[i] = (int32_t)s; tail.push_back(i); }
                    }
                for (int64_t c = S.childptr[s]; c < S.childptr[s + 1]; c++) {
                    const int32_t ch = S.children[c];
                    for (int64_t p = S.rowptr[ch] + S.sn_k[ch]; p < S.rowptr[ch + 1]; p++) {
                        const int32_t i = S.rowidx[p];
                        if (i >= e && mark[i] != s) { mark[i] = (int32_t)s; tail.push_back(i); }
                    }
                }
                std::sort(tail.begin(), tail.end());
                for (int64_t j = f; j < e; j++) S.rowidx.push_back((int32_t)j);
                S.rowidx.insert(S.rowidx.end(), tail.begin(), tail.end());
                S.rowptr[s + 1] = (int64_t)S.rowidx.size();
                S.sn_m[s] = (int32_t)(e - f + (int64_t)tail.size());
            }
        }
    }
    pt.lap("5 front rows");
    // ---- 6. relative indices child -> parent --------------------------------------------------
    S.rel.assign(S.rowidx.size(), -1);
    parallel_for(ns, analyze_threads(), 1 << 12, [&](int64_t lo, int64_t hi) {     // (disjoint slices of rel; exceptions travel to the caller)
        for (int64_t s = lo; s < hi; s++) {
            int32_t p = S.sparent[s];
            if (p < 0) {
                if (S.sn_m[s] != S.sn_k[s]) throw std::runtime_error("internal: root front has update rows");
                continue;
            }
            int64_t a = S.rowptr[s] + S.sn_k[s], ae = S.rowptr[s + 1];
            int64_t b = S.rowptr[p], be = S.rowptr[p + 1];
            for (; a < ae; a++) {
                while (b < be && S.rowidx[b] < S.rowidx[a]) b++;
                if (b >= be || S.rowidx[b] != S.rowidx[a]) throw std::runtime_error("internal: child row missing in parent front");
                S.rel[a] = (int32_t)(b - S.rowptr[p]);
            }
        }
    });
    pt.lap("6 relative indices");
    // ---- 7. storage offsets, levels ---------------------------------------------------------------
    S.px.assign((size_t)ns + 1, 0);
    S.max_m = 0; S.max_k = 0; S.sum_m = 0;
    for (int64_t s = 0; s < ns; s++) {
        S.px[s + 1] = S.px[s] + (int64_t)S.sn_m[s] * S.sn_k[s];
        S.max_m = std::max(S.max_m, S.sn_m[s]);
        S.max_k = std::max(S.max_k, S.sn_k[s]);
        S.sum_m += S.sn_m[s];
    }
    S.lsize = S.px[ns];
    S.depth.assign((size_t)ns, 0);
    int32_t maxd = -1;
    for (int64_t s = ns - 1; s >= 0; s--) {
        S.depth[s] = S.sparent[s] >= 0 ? S.depth[S.sparent[s]] + 1 : 0;
        maxd = std::max(maxd, S.depth[s]);
    }
    S.nlevels = maxd + 1;
    S.levelptr.assign((size_t)S.nlevels + 1, 0);
    for (int64_t s = 0; s < ns; s++) S.levelptr[S.depth[s] + 1]++;
    for (int32_t l = 0; l < S.nlevels; l++) S.levelptr[l + 1] += S.levelptr[l];
    S.levellist.resize((size_t)ns);
    {
        std::vector<int64_t> cur(S.levelptr.begin(), S.levelptr.end() - 1);
        for (int64_t s = 0; s < ns; s++) S.levellist[(size_t)cur[S.depth[s]]++] = (int32_t)s;
    }
    // within a level: grouped by kernel class (big fronts first: they get a contiguous head of
    // the update buffer), largest fronts first inside a class (long workgroups dispatched first)
    for (int32_t l = 0; l < S.nlevels; l++)
        std::stable_sort(S.levellist.begin() + S.levelptr[l], S.levellist.begin() + S.levelptr[l + 1],
                         [&](int32_t a, int32_t b) {
                             const int ca = front_class(S.sn_m[a], S.sn_k[a]), cb = front_class(S.sn_m[b], S.sn_k[b]);
                             return ca != cb ? ca < cb : S.sn_m[a] > S.sn_m[b];
                         });
    S.ux.assign((size_t)ns, 0);
    S.wx.assign((size_t)ns, 0);
    S.upd_size[0] = S.upd_size[1] = 0;
    S.wrk_size[0] = S.wrk_size[1] = 0;
    for (int32_t l = 0; l < S.nlevels; l++) {
        int64_t off = 0, woff = 0;
        for (int64_t q = S.levelptr[l]; q < S.levelptr[l + 1]; q++) {
            int32_t s = S.levellist[q];
            int64_t u = S.sn_m[s] - S.sn_k[s];
            S.ux[s] = off;
            S.wx[s] = woff;
            off += u * u;
            woff += u;
        }
        S.upd_size[l & 1] = std::max(S.upd_size[l & 1], off);
        S.wrk_size[l & 1] = std::max(S.wrk_size[l & 1], woff);
    }
    pt.lap("7 offsets+levels");
    // ---- 8. scatter map caller entries -> panels ------------------------------------------------
    S.amap.assign((size_t)S.nnzA, -1);
    {
        std::vector<int64_t> tri_part(64, 0);
        int part = 0;
        std::mutex mu;
        parallel_for(n, analyze_threads(), 1 << 14, [&](int64_t lo, int64_t hi) {
            int64_t tri = 0;
            for (int64_t j = lo; j < hi; j++)
                for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
                    int64_t i = Ai[p];
                    if ((uplo == 'L' && i < j) || (uplo != 'L' && i > j)) continue;
                    tri++;
                    int64_t a = S.iperm[i], b = S.iperm[j];
                    int64_t r = std::max(a, b), c = std::min(a, b);
                    int32_t s = col2s[c];
                    const int32_t *rb = S.rowidx.data() + S.rowptr[s];
                    const int32_t *re = S.rowidx.data() + S.rowptr[s + 1];
                    const int32_t *it = std::lower_bound(rb, re, (int32_t)r);
                    if (it == re || *it != r) throw std::runtime_error("internal: entry outside the front structure");
                    S.amap[p] = S.px[s] + (it - rb) + (c - S.super[s]) * (int64_t)S.sn_m[s];
                }
            std::lock_guard<std::mutex> lk(mu);
            tri_part[(size_t)part++ % 64] += tri;
        });
        S.nnzTri = 0;
        for (int64_t v : tri_part) S.nnzTri += v;
    }
    pt.lap("8 scatter map");
}

// flops of eliminating the k pivots of a front of order m: sum_{j < k} (m - j)^2
static double front_flops(double m, double k)
{
    auto S2 = [](double x) { return x * (x + 1.0) * (2.0 * x + 1.0) / 6.0; };
    return S2(m) - S2(m - k);
}

void dist_map(const Symbolic &S, int nranks, int ob, int min_m, DistMap &M)
{
    const int64_t ns = S.nsuper;
    M.nranks = std::max(nranks, 1);
    M.ob = std::max(64, ob / 64 * 64);
    M.min_m = min_m;
    M.glo.assign((size_t)ns, 0);
    M.ghi.assign((size_t)ns, 1);
    M.mode.assign((size_t)ns, 0);
    M.rank_flops.assign((size_t)M.nranks, 0.0);
    M.rank_panel_flops.assign((size_t)M.nranks, 0.0);
    M.total_flops = 0.0;
    M.replicated_flops = 0.0;
    const int P = M.nranks;
    // subtree work (children precede parents in the postordered numbering)
    std::vector<double> wsub((size_t)ns, 0.0);
    for (int64_t s = 0; s < ns; s++) {
        wsub[s] += front_flops(S.sn_m[s], S.sn_k[s]) + 1.0;
        if (S.sparent[s] >= 0) wsub[S.sparent[s]] += wsub[s];
    }
    // Proportional mapping, top down: a front worked on by the ranks [lo, hi) hands each child a contiguous window of them
    // whose width follows the child's share of the work below the front; a child whose share is below 1.5 ranks goes to ONE
    // rank (the least loaded), and from there down its whole subtree belongs to that rank.
    auto assign_children = [&](std::vector<int32_t> &ch, int lo, int hi) {
        const int g = hi - lo;
        if (g <= 1) {
            for (int32_t c : ch) { M.glo[c] = lo; M.ghi[c] = lo + 1; }
            return;
        }
        std::stable_sort(ch.begin(), ch.end(), [&](int32_t a, int32_t b) { return wsub[a] != wsub[b] ? wsub[a] > wsub[b] : a < b; });
        double W = 0.0;
        for (int32_t c : ch) W += wsub[c];
        std::vector<double> load((size_t)g, 0.0);
        for (int32_t c : ch) {
            const double want = W > 0.0 ? wsub[c] / W * g : 0.0;
            int gi = (int)(want + 0.5);
            if (gi >= 2) {
                gi = std::min(gi, g);
                int best = 0;
                double bsum = 0.0, bmax = 0.0;
                for (int a = 0; a + gi <= g; a++) {
                    double sm = 0.0, mx = 0.0;
                    for (int r = a; r < a + gi; r++) { sm += load[r]; mx = std::max(mx, load[r]); }
                    if (a == 0 || sm < bsum || (sm == bsum && mx < bmax)) { best = a; bsum = sm; bmax = mx; }
                }
                for (int r = best; r < best + gi; r++) load[r] += wsub[c] / gi;
                M.glo[c] = lo + best; M.ghi[c] = lo + best + gi;
            } else {
                int best = 0;
                for (int r = 1; r < g; r++)
                    if (load[r] < load[best]) best = r;
                load[best] += wsub[c];
                M.glo[c] = lo + best; M.ghi[c] = lo + best + 1;
            }
        }
    };
    std::vector<int32_t> ch;
    for (int64_t s = 0; s < ns; s++)
        if (S.sparent[s] < 0) ch.push_back((int32_t)s);
    assign_children(ch, 0, P);                                  // the roots of the forest share all ranks
    for (int64_t s = ns - 1; s >= 0; s--) {
        ch.assign(S.children.begin() + S.childptr[s], S.children.begin() + S.childptr[s + 1]);
        if (!ch.empty()) assign_children(ch, M.glo[s], M.ghi[s]);
    }
    // block-cyclic fronts + the flops every rank executes
    for (int64_t s = 0; s < ns; s++) {
        const int g = M.ghi[s] - M.glo[s];
        const double m = S.sn_m[s], k = S.sn_k[s];
        const double f = front_flops(m, k);
        M.total_flops += f;
        const bool big = front_class(S.sn_m[s], S.sn_k[s]) == KVX_CLS_BIG;
        if (g > 1 && big && S.sn_m[s] >= min_m && S.sn_k[s] >= std::min(M.ob, 256)) M.mode[s] = 1;
        if (g == 1) { M.rank_flops[M.glo[s]] += f; continue; }
        if (!M.mode[s]) {
            for (int r = M.glo[s]; r < M.ghi[s]; r++) M.rank_flops[r] += f;
            M.replicated_flops += f;
            continue;
        }
        const int64_t mm = S.sn_m[s], kk = S.sn_k[s], OB = M.ob;
        const int64_t nkb = (kk + OB - 1) / OB;
        for (int64_t b = 0; b < nkb; b++) {
            const int64_t o = b * OB, nb = std::min<int64_t>(OB, kk - o), t = mm - o - nb;
            const double tot = front_flops((double)(mm - o), (double)nb);
            const double upd = (double)nb * t * (t + 1);
            const int owner = M.glo[s] + (int)(b % g);
            M.rank_flops[owner] += tot - upd;
            M.rank_panel_flops[owner] += tot - upd;
            // the rank-nb update of the columns right of the block, by column block
            for (int64_t c0 = o + nb; c0 < mm; ) {
                const int64_t blk = c0 < kk ? c0 / OB : nkb + (c0 - kk) / OB;
                const int64_t c1 = std::min<int64_t>(mm, c0 < kk ? std::min<int64_t>((blk + 1) * OB, kk) : kk + (blk - nkb + 1) * OB);
                const double ent = 0.5 * (double)(c1 - c0) * (double)((mm - c0) + (mm - c1 + 1));   // lower-trapezoid entries
                M.rank_flops[M.glo[s] + (int)(blk % g)] += 2.0 * nb * ent;
                c0 = c1;
            }
        }
    }
}

}  // namespace kvx
