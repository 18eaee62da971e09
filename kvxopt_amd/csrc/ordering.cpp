// Fill-reducing ordering: automatic nested dissection on level structures with a
// halo-aware minimum-degree ordering of the leaf subdomains.
//
// Role in the reference: the ordering CHOLMOD picks inside cholmod_l_analyze_p
// (src/C/cholmod.c:274; AMD by default, doc/source/spsolvers.rst:738-752).  This is an
// independent design chosen for the GPU schedule: nested dissection yields a wide,
// balanced elimination tree (many independent fronts per level) and large dense
// separator fronts at the top (MFMA-friendly), which minimum-degree orderings do not.
#include "symbolic.hpp"

#include <algorithm>
#include <cstring>
#include <numeric>

namespace kvx {
namespace {

struct Graph {
    int64_t n;
    const int64_t *ptr;
    const int32_t *adj;
};

struct NDState {
    Graph g;
    std::vector<int32_t> verts;    // task ranges are contiguous slices; slice position = final position
    std::vector<int32_t> region;   // region id of every vertex (-1 = already numbered)
    std::vector<int32_t> level;    // BFS scratch
    std::vector<int32_t> queue;    // BFS scratch
    int32_t next_region = 1;
};

// BFS restricted to region `rid` from `root`; fills st.queue[0..cnt) in BFS order and
// st.level[v]; returns cnt, sets nlev. Levels are reset by the caller through `touched`.
int64_t bfs(NDState &st, int32_t rid, int32_t root, int32_t &nlev)
{
    int64_t head = 0, tail = 0;
    st.queue[tail++] = root;
    st.level[root] = 0;
    nlev = 1;
    while (head < tail) {
        int32_t v = st.queue[head++];
        int32_t lv = st.level[v];
        for (int64_t p = st.g.ptr[v]; p < st.g.ptr[v + 1]; p++) {
            int32_t u = st.g.adj[p];
            if (st.region[u] != rid || st.level[u] >= 0) continue;
            st.level[u] = lv + 1;
            if (lv + 2 > nlev) nlev = lv + 2;
            st.queue[tail++] = u;
        }
    }
    return tail;
}

void reset_levels(NDState &st, int64_t cnt)
{
    for (int64_t i = 0; i < cnt; i++) st.level[st.queue[i]] = -1;
}

// Halo-aware exact minimum degree on a small vertex set (bitset elimination graph).
// verts[lo,hi) are reordered in place into elimination order.
void leaf_min_degree(NDState &st, int64_t lo, int64_t hi, std::vector<int32_t> &local_id)
{
    const int64_t s = hi - lo;
    if (s <= 2) return;
    // collect halo: outside neighbours (already-numbered separators or other regions)
    std::vector<int32_t> nodes(st.verts.begin() + lo, st.verts.begin() + hi);
    for (int64_t i = 0; i < s; i++) local_id[nodes[i]] = (int32_t)i;
    for (int64_t i = 0; i < s; i++) {
        int32_t v = nodes[i];
        for (int64_t p = st.g.ptr[v]; p < st.g.ptr[v + 1]; p++) {
            int32_t u = st.g.adj[p];
            if (local_id[u] < 0) { local_id[u] = (int32_t)nodes.size(); nodes.push_back(u); }
        }
    }
    const int64_t t = (int64_t)nodes.size();
    const int64_t W = (t + 63) / 64;
    std::vector<uint64_t> bits((size_t)(t * W), 0);
    auto row = [&](int64_t i) { return bits.data() + i * W; };
    for (int64_t i = 0; i < s; i++) {
        int32_t v = nodes[i];
        for (int64_t p = st.g.ptr[v]; p < st.g.ptr[v + 1]; p++) {
            int32_t u = local_id[st.g.adj[p]];
            row(i)[u >> 6] |= 1ull << (u & 63);
            row(u)[i >> 6] |= 1ull << (i & 63);
        }
    }
    std::vector<char> done((size_t)s, 0);
    std::vector<int32_t> order;
    order.reserve((size_t)s);
    for (int64_t step = 0; step < s; step++) {
        int64_t best = -1, bestdeg = INT64_MAX;
        for (int64_t i = 0; i < s; i++) {
            if (done[i]) continue;
            int64_t d = 0;
            const uint64_t *r = row(i);
            for (int64_t w = 0; w < W; w++) d += __builtin_popcountll(r[w]);
            if (d < bestdeg) { bestdeg = d; best = i; }
        }
        done[best] = 1;
        order.push_back(nodes[best]);
        const uint64_t *rb = row(best);
        // neighbours of `best` become a clique; remove `best` from their rows
        for (int64_t w = 0; w < W; w++) {
            uint64_t m = rb[w];
            while (m) {
                int64_t u = w * 64 + __builtin_ctzll(m);
                m &= m - 1;
                uint64_t *ru = row(u);
                for (int64_t x = 0; x < W; x++) ru[x] |= rb[x];
                ru[u >> 6] &= ~(1ull << (u & 63));
                ru[best >> 6] &= ~(1ull << (best & 63));
            }
        }
    }
    for (int64_t i = 0; i < t; i++) local_id[nodes[i]] = -1;
    std::copy(order.begin(), order.end(), st.verts.begin() + lo);
}

}  // namespace

void order_nd(int64_t n, const std::vector<int64_t> &adjptr, const std::vector<int32_t> &adj,
              int leaf, std::vector<int64_t> &perm)
{
    perm.resize((size_t)n);
    if (n == 0) return;
    NDState st;
    st.g = Graph{n, adjptr.data(), adj.data()};
    st.verts.resize((size_t)n);
    std::iota(st.verts.begin(), st.verts.end(), 0);
    st.region.assign((size_t)n, 0);
    st.level.assign((size_t)n, -1);
    st.queue.resize((size_t)n);
    std::vector<int32_t> local_id((size_t)n, -1);
    std::vector<int64_t> levcnt;
    if (leaf < 4) leaf = 4;

    struct Task { int64_t lo, hi; int32_t rid; };
    std::vector<Task> stack;
    stack.push_back(Task{0, n, 0});

    while (!stack.empty()) {
        Task t = stack.back();
        stack.pop_back();
        const int64_t sz = t.hi - t.lo;
        if (sz <= 0) continue;
        if (sz <= leaf) {
            leaf_min_degree(st, t.lo, t.hi, local_id);
            for (int64_t i = t.lo; i < t.hi; i++) st.region[st.verts[i]] = -1;
            continue;
        }
        // pseudo-peripheral root: repeat BFS from a min-degree vertex of the last level
        int32_t root = st.verts[t.lo];
        int32_t nlev = 0;
        int64_t cnt = 0;
        for (int iter = 0; iter < 4; iter++) {
            cnt = bfs(st, t.rid, root, nlev);
            int32_t cand = -1;
            int64_t cdeg = INT64_MAX;
            for (int64_t i = cnt - 1; i >= 0 && st.level[st.queue[i]] == nlev - 1; i--) {
                int32_t v = st.queue[i];
                int64_t d = st.g.ptr[v + 1] - st.g.ptr[v];
                if (d < cdeg) { cdeg = d; cand = v; }
            }
            if (iter == 3 || cand == root || cand < 0) break;
            // does the new root give a deeper structure? try it
            int32_t old_nlev = nlev, old_root = root;
            reset_levels(st, cnt);
            root = cand;
            cnt = bfs(st, t.rid, root, nlev);
            if (nlev <= old_nlev) {
                if (nlev < old_nlev) { reset_levels(st, cnt); root = old_root; cnt = bfs(st, t.rid, root, nlev); }
                break;
            }
            reset_levels(st, cnt);
        }
        // queue[0..cnt) holds one connected component in BFS order with levels set
        if (cnt < sz) {
            // disconnected: peel this component off as its own task, no separator
            int32_t ra = st.next_region++, rb = st.next_region++;
            for (int64_t i = 0; i < cnt; i++) st.region[st.queue[i]] = ra;
            int64_t a = t.lo, b = t.lo + cnt;
            std::vector<int32_t> tmp(st.verts.begin() + t.lo, st.verts.begin() + t.hi);
            for (int32_t v : tmp) {
                if (st.region[v] == ra) st.verts[a++] = v;
                else { st.region[v] = rb; st.verts[b++] = v; }
            }
            reset_levels(st, cnt);
            stack.push_back(Task{t.lo + cnt, t.hi, rb});
            stack.push_back(Task{t.lo, t.lo + cnt, ra});
            continue;
        }
        if (nlev < 3) {
            // diameter too small for a level-set separator (near-clique): number by degree
            reset_levels(st, cnt);
            std::sort(st.verts.begin() + t.lo, st.verts.begin() + t.hi, [&](int32_t a, int32_t b) {
                int64_t da = st.g.ptr[a + 1] - st.g.ptr[a], db = st.g.ptr[b + 1] - st.g.ptr[b];
                return da != db ? da < db : a < b;
            });
            for (int64_t i = t.lo; i < t.hi; i++) st.region[st.verts[i]] = -1;
            continue;
        }
        levcnt.assign((size_t)nlev, 0);
        for (int64_t i = 0; i < cnt; i++) levcnt[st.level[st.queue[i]]]++;
        // pick the separator level: smallest level among those leaving >= 30% on each side,
        // else the level where the cumulative count crosses one half
        int32_t best = -1;
        int64_t bestcnt = INT64_MAX, cum = 0, half_lev = 1;
        for (int32_t l = 0; l < nlev; l++) {
            int64_t below = cum, above = sz - cum - levcnt[l];
            if (l >= 1 && l <= nlev - 2 && below * 10 >= sz * 3 && above * 10 >= sz * 3 && levcnt[l] < bestcnt) {
                bestcnt = levcnt[l];
                best = l;
            }
            if (cum * 2 < sz) half_lev = l;
            cum += levcnt[l];
        }
        if (best < 0) best = std::min<int32_t>(std::max<int32_t>((int32_t)half_lev, 1), nlev - 2);
        // separator = vertices of level `best` with a neighbour in level best+1;
        // the rest of level `best` joins part A
        int32_t ra = st.next_region++, rb = st.next_region++;
        int64_t nA = 0, nB = 0, nS = 0;
        for (int64_t i = 0; i < cnt; i++) {
            int32_t v = st.queue[i];
            int32_t lv = st.level[v];
            if (lv < best) { st.region[v] = ra; nA++; }
            else if (lv > best) { st.region[v] = rb; nB++; }
            else {
                bool touches = false;
                for (int64_t p = st.g.ptr[v]; p < st.g.ptr[v + 1] && !touches; p++) {
                    int32_t u = st.g.adj[p];
                    touches = (st.level[u] == best + 1) && (st.region[u] == t.rid || st.region[u] == rb);
                }
                if (touches) { st.region[v] = -1; nS++; }
                else { st.region[v] = ra; nA++; }
            }
        }
        // lay out verts[lo,hi) as [A | B | S], each in BFS order
        int64_t a = t.lo, b = t.lo + nA, s = t.lo + nA + nB;
        for (int64_t i = 0; i < cnt; i++) {
            int32_t v = st.queue[i];
            if (st.region[v] == ra) st.verts[a++] = v;
            else if (st.region[v] == rb) st.verts[b++] = v;
            else st.verts[s++] = v;
        }
        reset_levels(st, cnt);
        stack.push_back(Task{t.lo + nA, t.lo + nA + nB, rb});
        stack.push_back(Task{t.lo, t.lo + nA, ra});
    }
    for (int64_t i = 0; i < n; i++) perm[(size_t)i] = st.verts[(size_t)i];
}

}  // namespace kvx
