// Approximate minimum degree ordering on the quotient graph (the algorithm family of Amestoy, Davis & Duff 1996: elements
// instead of explicit fill, approximate external degrees, element absorption, mass elimination of indistinguishable
// variables).  Own implementation -- not SuiteSparse AMD -- written for clarity: per-node adjacency vectors cleaned lazily.
//
// Role in this library: the ordering CHOLMOD would obtain from AMD (reference call path src/C/cholmod.c:274 ->
// cholmod_l_analyze_p, and the `amd.order` module, src/C/amd.c:131-223).  symbolic.cpp compares it with the nested-
// dissection ordering (and a user permutation) by the fill it produces and keeps the best, CHOLMOD's `nmethods` strategy
// (cholmod.c:65-76).  Unstructured patterns (bcsstk13, random SPD) favour it; grids favour the dissection.
#include "symbolic.hpp"

#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

namespace kvx {

void order_amd(int64_t n64, const std::vector<int64_t> &adjptr, const std::vector<int32_t> &adj, std::vector<int64_t> &perm)
{
    const int32_t n = (int32_t)n64;
    perm.resize((size_t)n);
    if (n == 0) return;
    // state of node i: a VARIABLE (nv > 0: principal supervariable of that many original variables; nv == 0 and not yet
    // eliminated: merged into another supervariable), or -- once eliminated -- an ELEMENT (its variable list in elv)
    std::vector<std::vector<int32_t>> adjv((size_t)n), adje((size_t)n), elv((size_t)n);
    std::vector<int32_t> nv((size_t)n, 1), deg((size_t)n), elem_deg((size_t)n, 0);
    std::vector<uint8_t> is_elem((size_t)n, 0), dead((size_t)n, 0);   // dead element = absorbed into a newer one
    std::vector<int32_t> merged_into((size_t)n, -1);
    for (int32_t i = 0; i < n; i++) {
        adjv[i].assign(adj.begin() + adjptr[i], adj.begin() + adjptr[i + 1]);
        deg[i] = (int32_t)adjv[i].size();
    }
    // degree buckets (doubly linked lists)
    std::vector<int32_t> head((size_t)n + 1, -1), next((size_t)n, -1), prev((size_t)n, -1);
    auto bucket_add = [&](int32_t i) {
        const int32_t d = std::min(std::max(deg[i], 0), n);
        next[i] = head[d]; prev[i] = -1;
        if (head[d] >= 0) prev[head[d]] = i;
        head[d] = i;
    };
    auto bucket_del = [&](int32_t i) {
        const int32_t d = std::min(std::max(deg[i], 0), n);
        if (prev[i] >= 0) next[prev[i]] = next[i]; else head[d] = next[i];
        if (next[i] >= 0) prev[next[i]] = prev[i];
    };
    for (int32_t i = 0; i < n; i++) bucket_add(i);
    std::vector<int64_t> w((size_t)n, 0);         // w[e] - wflg = weight of elv[e] outside the current pivot element
    int64_t wflg = 1;
    std::vector<int32_t> mark((size_t)n, -1);     // mark[i] == pivot id: i belongs to the pivot element
    std::vector<int32_t> Lp, order;               // order: pivots in elimination order
    std::vector<uint32_t> hash((size_t)n, 0);
    order.reserve((size_t)n);
    int32_t nel = 0, mindeg = 0;
    while (nel < n) {
        while (mindeg <= n && head[mindeg] < 0) mindeg++;
        const int32_t p = head[mindeg];
        bucket_del(p);
        // ---- the new element: variables adjacent to p, directly or through its elements (which p absorbs)
        Lp.clear();
        mark[p] = p;
        int64_t degme = 0;
        for (int32_t j : adjv[p])
            if (nv[j] > 0 && !is_elem[j] && mark[j] != p) { mark[j] = p; Lp.push_back(j); degme += nv[j]; }
        for (int32_t e : adje[p]) {
            if (dead[e]) continue;
            for (int32_t j : elv[e])
                if (nv[j] > 0 && !is_elem[j] && mark[j] != p) { mark[j] = p; Lp.push_back(j); degme += nv[j]; }
            dead[e] = 1;
            std::vector<int32_t>().swap(elv[e]);
        }
        is_elem[p] = 1;
        nel += nv[p];
        order.push_back(p);
        std::vector<int32_t>().swap(adjv[p]);
        std::vector<int32_t>().swap(adje[p]);
        // ---- first pass over the element: |Le \ Lp| for every element adjacent to a variable of Lp
        for (int32_t i : Lp) {
            bucket_del(i);
            for (int32_t e : adje[i]) {
                if (dead[e]) continue;
                if (w[e] < wflg) w[e] = elem_deg[e] + wflg;
                w[e] -= nv[i];
            }
        }
        // ---- second pass: clean the lists, approximate degrees, hashes
        for (int32_t i : Lp) {
            int64_t d = 0;
            uint32_t h = 0;
            size_t k = 0;
            for (int32_t e : adje[i]) {
                if (dead[e]) continue;
                const int64_t out = w[e] - wflg;
                if (out > 0) { d += out; h += (uint32_t)e; adje[i][k++] = e; }
                else { dead[e] = 1; std::vector<int32_t>().swap(elv[e]); }      // aggressive absorption: Le is inside Lp
            }
            adje[i].resize(k);
            adje[i].push_back(p);
            h += (uint32_t)p;
            k = 0;
            for (int32_t j : adjv[i])
                if (nv[j] > 0 && !is_elem[j] && mark[j] != p) { d += nv[j]; h += (uint32_t)j; adjv[i][k++] = j; }
            adjv[i].resize(k);
            const int64_t ext = degme - nv[i];
            int64_t dn = std::min<int64_t>((int64_t)deg[i] + ext, d + ext);
            dn = std::min<int64_t>(dn, (int64_t)n - nel - nv[i]);
            deg[i] = (int32_t)std::max<int64_t>(dn, 0);
            hash[i] = h;
        }
        wflg += (int64_t)n + 1;                    // invalidates every w[e] of this step
        // ---- mass elimination: variables of Lp with identical adjacency are indistinguishable -> one supervariable
        if (Lp.size() > 1) {
            std::vector<int32_t> byhash(Lp);
            std::sort(byhash.begin(), byhash.end(), [&](int32_t a, int32_t b) { return hash[a] != hash[b] ? hash[a] < hash[b] : a < b; });
            for (size_t a = 0; a < byhash.size(); a++) {
                const int32_t i = byhash[a];
                if (nv[i] == 0) continue;
                bool sorted_i = false;
                for (size_t b = a + 1; b < byhash.size() && hash[byhash[b]] == hash[i]; b++) {
                    const int32_t j = byhash[b];
                    if (nv[j] == 0 || adjv[j].size() != adjv[i].size() || adje[j].size() != adje[i].size()) continue;
                    if (!sorted_i) { std::sort(adjv[i].begin(), adjv[i].end()); std::sort(adje[i].begin(), adje[i].end()); sorted_i = true; }
                    std::sort(adjv[j].begin(), adjv[j].end());
                    std::sort(adje[j].begin(), adje[j].end());
                    if (adjv[j] != adjv[i] || adje[j] != adje[i]) continue;
                    nv[i] += nv[j];                // j joins i
                    deg[i] -= nv[j];
                    nv[j] = 0;
                    merged_into[j] = i;
                    std::vector<int32_t>().swap(adjv[j]);
                    std::vector<int32_t>().swap(adje[j]);
                }
            }
        }
        // ---- the element itself, and the survivors back into the degree lists
        elv[p].clear();
        int64_t dp = 0;
        for (int32_t i : Lp)
            if (nv[i] > 0) { elv[p].push_back(i); dp += nv[i]; if (deg[i] < 0) deg[i] = 0; bucket_add(i); mindeg = std::min(mindeg, std::min(deg[i], n)); }
        elem_deg[p] = (int32_t)dp;
        if (elv[p].empty()) dead[p] = 1;
    }
    // ---- permutation: every pivot followed by the variables merged into it (transitively)
    std::vector<std::vector<int32_t>> members((size_t)n);
    for (int32_t j = 0; j < n; j++)
        if (merged_into[j] >= 0) members[merged_into[j]].push_back(j);
    int64_t pos = 0;
    std::vector<int32_t> stack;
    for (int32_t p : order) {
        stack.assign(1, p);
        while (!stack.empty()) {
            const int32_t v = stack.back(); stack.pop_back();
            perm[(size_t)pos++] = v;
            for (int32_t c : members[v]) stack.push_back(c);
        }
    }
}

}  // namespace kvx
