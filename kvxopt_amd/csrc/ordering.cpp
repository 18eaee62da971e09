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
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <exception>
#include <thread>
#include <utility>

namespace kvx {
namespace {

struct Graph {
    int64_t n;
    const int64_t *ptr;
    const int32_t *adj;
};

// Subdomains are dissected by several host threads (order_nd).  A task owns its slice of `verts` and `queue` and writes
// region / level of its own vertices only, but it READS region / level of neighbours that may belong to another task; those
// reads can never change a decision (a foreign region id equals neither the task's own id nor the ids it hands out), so the
// ordering is the same for any thread count.  Relaxed atomics make the concurrent accesses well defined; they are plain moves.
struct RelaxedI32 {
    std::atomic<int32_t> v{0};
    RelaxedI32() = default;
    RelaxedI32(const RelaxedI32 &o) : v(o.v.load(std::memory_order_relaxed)) {}
    operator int32_t() const { return v.load(std::memory_order_relaxed); }
    RelaxedI32 &operator=(int32_t x) { v.store(x, std::memory_order_relaxed); return *this; }
};

struct NDState {
    Graph g;
    std::vector<int32_t> verts;      // task ranges are contiguous slices; slice position = final position
    struct RL { RelaxedI32 region, level; };   // region id (-1 = already numbered) and BFS level (-1 = not visited) of a vertex, side by
    std::vector<RL> rl;                        // side: a breadth-first pass looks at both for every neighbour (one cache line, not two)
    std::vector<int32_t> queue;      // BFS scratch: a task uses the slice of its own vertex range
    std::atomic<int32_t> next_region{1};
};

// BFS restricted to region `rid` from `root`; fills st.queue[0..cnt) in BFS order and
// st.level[v]; returns cnt, sets nlev. Levels are reset by the caller through `touched`.
int64_t bfs(NDState &st, int32_t *queue, int32_t rid, int32_t root, int32_t &nlev)
{
    int64_t head = 0, tail = 0;
    queue[tail++] = root;
    st.rl[root].level = 0;
    nlev = 1;
    while (head < tail) {
        int32_t v = queue[head++];
        int32_t lv = st.rl[v].level;
        for (int64_t p = st.g.ptr[v]; p < st.g.ptr[v + 1]; p++) {
            int32_t u = st.g.adj[p];
            if (st.rl[u].region != rid || st.rl[u].level >= 0) continue;
            st.rl[u].level = lv + 1;
            if (lv + 2 > nlev) nlev = lv + 2;
            queue[tail++] = u;
        }
    }
    return tail;
}

void reset_levels(NDState &st, const int32_t *queue, int64_t cnt)
{
    for (int64_t i = 0; i < cnt; i++) st.rl[queue[i]].level = -1;
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
    // degrees are kept up to date instead of recounted: an elimination changes the rows of the eliminated vertex's neighbours
    // only, and only the rows of the leaf's own vertices are ever read (a halo vertex is never eliminated and its row never
    // consulted), so those are the only ones touched.  Same choices as recounting everything at every step.
    std::vector<char> done((size_t)s, 0);
    std::vector<int32_t> order, deg((size_t)s);
    order.reserve((size_t)s);
    for (int64_t i = 0; i < s; i++) {
        int32_t d = 0;
        const uint64_t *r = row(i);
        for (int64_t w = 0; w < W; w++) d += (int32_t)__builtin_popcountll(r[w]);
        deg[(size_t)i] = d;
    }
    for (int64_t step = 0; step < s; step++) {
        int64_t best = -1;
        int32_t bestdeg = INT32_MAX;
        for (int64_t i = 0; i < s; i++)
            if (!done[i] && deg[(size_t)i] < bestdeg) { bestdeg = deg[(size_t)i]; best = i; }
        done[best] = 1;
        order.push_back(nodes[best]);
        const uint64_t *rb = row(best);
        // neighbours of `best` become a clique; remove `best` from their rows
        for (int64_t w = 0; w < W; w++) {
            uint64_t m = rb[w];
            while (m) {
                const int64_t u = w * 64 + __builtin_ctzll(m);
                m &= m - 1;
                if (u >= s) continue;
                uint64_t *ru = row(u);
                int32_t d = 0;
                for (int64_t x = 0; x < W; x++) ru[x] |= rb[x];
                ru[u >> 6] &= ~(1ull << (u & 63));
                ru[best >> 6] &= ~(1ull << (best & 63));
                for (int64_t x = 0; x < W; x++) d += (int32_t)__builtin_popcountll(ru[x]);
                deg[(size_t)u] = d;
            }
        }
    }
    for (int64_t i = 0; i < t; i++) local_id[nodes[i]] = -1;
    std::copy(order.begin(), order.end(), st.verts.begin() + lo);
}

struct NDTask { int64_t lo, hi; int32_t rid; };

// KVX_ANALYZE_TIMING=1: where the dissection spends its time (summed over the host threads, and along the chain of the largest region)
struct NDProf {
    bool on = getenv("KVX_ANALYZE_TIMING") != nullptr;
    std::atomic<int64_t> bfs_ns{0}, pass_ns{0}, leaf_ns{0}, chain_ns{0};
};
static NDProf *g_ndprof = nullptr;
static inline int64_t nd_now()
{
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct NDWorker {
    std::vector<int32_t> local_id;   // leaf_min_degree scratch, indexed by vertex (halo vertices are shared between leaves)
    std::vector<int64_t> levcnt;
};

// One dissection step on the task's slice; pushes 0..2 child tasks (first pushed = processed last on a LIFO stack).
void nd_step(NDState &st, NDWorker &wk, const NDTask &t, int leaf, std::vector<NDTask> &out)
{
    const int64_t sz = t.hi - t.lo;
    if (sz <= 0) return;
    if (sz <= leaf) {
        if (wk.local_id.empty()) wk.local_id.assign((size_t)st.g.n, -1);
        const int64_t t0 = g_ndprof && g_ndprof->on ? nd_now() : 0;
        leaf_min_degree(st, t.lo, t.hi, wk.local_id);
        if (g_ndprof && g_ndprof->on) g_ndprof->leaf_ns += nd_now() - t0;
        for (int64_t i = t.lo; i < t.hi; i++) st.rl[st.verts[i]].region = -1;
        return;
    }
    int32_t *queue = st.queue.data() + t.lo;       // a connected component of the slice has at most sz vertices
    // pseudo-peripheral root: repeat BFS from a min-degree vertex of the last level
    const bool prof = g_ndprof && g_ndprof->on;
    const int64_t tb0 = prof ? nd_now() : 0;
    int32_t root = st.verts[t.lo];
    int32_t nlev = 0;
    int64_t cnt = bfs(st, queue, t.rid, root, nlev);
    for (int iter = 0; iter < 4; iter++) {
        // (the level structure of `root` is in place here: a deeper candidate of the previous round is not swept again)
        int32_t cand = -1;
        int64_t cdeg = INT64_MAX;
        for (int64_t i = cnt - 1; i >= 0 && st.rl[queue[i]].level == nlev - 1; i--) {
            int32_t v = queue[i];
            int64_t d = st.g.ptr[v + 1] - st.g.ptr[v];
            if (d < cdeg) { cdeg = d; cand = v; }
        }
        if (iter == 3 || cand == root || cand < 0) break;
        // does the new root give a deeper structure? try it
        int32_t old_nlev = nlev, old_root = root;
        reset_levels(st, queue, cnt);
        root = cand;
        cnt = bfs(st, queue, t.rid, root, nlev);
        if (nlev <= old_nlev) {
            if (nlev < old_nlev) { reset_levels(st, queue, cnt); root = old_root; cnt = bfs(st, queue, t.rid, root, nlev); }
            break;
        }
    }
    const int64_t tb1 = prof ? nd_now() : 0;
    if (prof) { g_ndprof->bfs_ns += tb1 - tb0; if (sz * 2 >= st.g.n || t.lo == 0) g_ndprof->chain_ns += tb1 - tb0; }
    // queue[0..cnt) holds one connected component in BFS order with levels set
    if (cnt < sz) {
        // disconnected: every connected component of the slice becomes a task of its own, no separator.  One sweep labels them
        // all (a slice that falls apart into hundreds of pieces -- power grids, LP bases -- used to be peeled one component per
        // step, each step rewriting the rest of the slice); the layout is the one the peeling produced: components in the order
        // of their first vertex in the slice, every component in slice order.
        reset_levels(st, queue, cnt);
        std::vector<int32_t> tmp(st.verts.begin() + t.lo, st.verts.begin() + t.hi);
        std::vector<int32_t> rid_of;                 // region id of component c
        std::vector<int64_t> csize;
        for (int32_t v0 : tmp) {
            if (st.rl[v0].region != t.rid) continue;    // already in a component
            const int32_t rc = st.next_region.fetch_add(1, std::memory_order_relaxed);
            const int32_t ci = (int32_t)rid_of.size();
            rid_of.push_back(rc);
            int64_t head = 0, tail = 0;
            queue[tail++] = v0;
            st.rl[v0].region = rc;
            st.rl[v0].level = ci;                       // (component index, cleared below)
            while (head < tail) {
                const int32_t v = queue[head++];
                for (int64_t p = st.g.ptr[v]; p < st.g.ptr[v + 1]; p++) {
                    const int32_t u = st.g.adj[p];
                    if (st.rl[u].region != t.rid) continue;
                    st.rl[u].region = rc;
                    st.rl[u].level = ci;
                    queue[tail++] = u;
                }
            }
            csize.push_back(tail);
        }
        std::vector<int64_t> pos(csize.size() + 1, t.lo);
        for (size_t c = 0; c < csize.size(); c++) pos[c + 1] = pos[c] + csize[c];
        {
            std::vector<int64_t> cur(pos.begin(), pos.end() - 1);
            for (int32_t v : tmp) {
                st.verts[(size_t)cur[(size_t)(int32_t)st.rl[v].level]++] = v;
                st.rl[v].level = -1;
            }
        }
        for (size_t c = csize.size(); c-- > 0;) out.push_back(NDTask{pos[c], pos[c + 1], rid_of[c]});   // (LIFO: the first component is taken first)
        return;
    }
    if (nlev < 3) {
        // diameter too small for a level-set separator (near-clique): number by degree
        reset_levels(st, queue, cnt);
        std::sort(st.verts.begin() + t.lo, st.verts.begin() + t.hi, [&](int32_t a, int32_t b) {
            int64_t da = st.g.ptr[a + 1] - st.g.ptr[a], db = st.g.ptr[b + 1] - st.g.ptr[b];
            return da != db ? da < db : a < b;
        });
        for (int64_t i = t.lo; i < t.hi; i++) st.rl[st.verts[i]].region = -1;
        return;
    }
    std::vector<int64_t> &levcnt = wk.levcnt;
    levcnt.assign((size_t)nlev, 0);
    for (int64_t i = 0; i < cnt; i++) levcnt[st.rl[queue[i]].level]++;
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
    int32_t ra = st.next_region.fetch_add(2, std::memory_order_relaxed), rb = ra + 1;
    int64_t nA = 0, nB = 0, nS = 0;
    for (int64_t i = 0; i < cnt; i++) {
        int32_t v = queue[i];
        int32_t lv = st.rl[v].level;
        if (lv < best) { st.rl[v].region = ra; nA++; }
        else if (lv > best) { st.rl[v].region = rb; nB++; }
        else {
            bool touches = false;
            for (int64_t p = st.g.ptr[v]; p < st.g.ptr[v + 1] && !touches; p++) {
                int32_t u = st.g.adj[p];
                const int32_t ru = st.rl[u].region;
                touches = (ru == t.rid || ru == rb) && (st.rl[u].level == best + 1);
            }
            if (touches) { st.rl[v].region = -1; nS++; }
            else { st.rl[v].region = ra; nA++; }
        }
    }
    // lay out verts[lo,hi) as [A | B | S], each in BFS order
    int64_t a = t.lo, b = t.lo + nA, s = t.lo + nA + nB;
    for (int64_t i = 0; i < cnt; i++) {
        int32_t v = queue[i];
        if (st.rl[v].region == ra) st.verts[a++] = v;
        else if (st.rl[v].region == rb) st.verts[b++] = v;
        else st.verts[s++] = v;
    }
    reset_levels(st, queue, cnt);
    if (prof) g_ndprof->pass_ns += nd_now() - tb1;
    out.push_back(NDTask{t.lo + nA, t.lo + nA + nB, rb});
    out.push_back(NDTask{t.lo, t.lo + nA, ra});
}

// a whole subtree of tasks, depth first, on the calling thread
void nd_run_local(NDState &st, NDWorker &wk, NDTask t0, int leaf)
{
    std::vector<NDTask> stack{t0};
    while (!stack.empty()) {
        NDTask t = stack.back();
        stack.pop_back();
        nd_step(st, wk, t, leaf, stack);
    }
}

}  // namespace

void order_nd(int64_t n, const std::vector<int64_t> &adjptr, const std::vector<int32_t> &adj,
              int leaf, std::vector<int64_t> &perm, std::vector<std::pair<int64_t, int64_t>> *closed)
{
    if (closed) closed->clear();
    perm.resize((size_t)n);
    if (n == 0) return;
    NDProf prof_obj;
    g_ndprof = prof_obj.on ? &prof_obj : nullptr;
    NDState st;
    st.g = Graph{n, adjptr.data(), adj.data()};
    st.verts.resize((size_t)n);
    std::iota(st.verts.begin(), st.verts.end(), 0);
    st.rl = std::vector<NDState::RL>((size_t)n);                // region 0
    for (int64_t i = 0; i < n; i++) st.rl[(size_t)i].level = -1;
    st.queue.resize((size_t)n);
    if (leaf < 4) leaf = 4;

    // Host threads: subdomains are independent once their separator is numbered.  Tasks of at least `cutoff` vertices go
    // through a shared pool (their two halves become new pool tasks); smaller ones are finished depth-first by whoever takes
    // them.  The result does not depend on the thread count (see NDState).  KVX_ND_THREADS = 1 runs the whole tree in place.
    int nthreads = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char *e = getenv("KVX_ND_THREADS")) nthreads = std::max(1, std::min(64, atoi(e)));
    const int64_t cutoff = 8192;
    if (n < 4 * cutoff) nthreads = 1;
    if (nthreads == 1) {
        NDWorker wk;
        nd_run_local(st, wk, NDTask{0, n, 0}, leaf);
    } else {
        std::mutex mu;
        std::condition_variable cv;
        std::vector<NDTask> pool{NDTask{0, n, 0}};
        int busy = 0;                               // tasks taken and not finished (guarded by mu)
        std::exception_ptr failure;                 // first exception of any worker (guarded by mu); rethrown on the caller
        bool stop = false;
        auto worker = [&]() {
            try {
                NDWorker wk;
                std::vector<NDTask> out;
                for (;;) {
                    NDTask t;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return stop || !pool.empty() || busy == 0; });
                        if (stop || pool.empty()) return;   // failed elsewhere / nothing queued and nobody who could queue more
                        t = pool.back();
                        pool.pop_back();
                        busy++;
                    }
                    try {
                        out.clear();
                        if (t.hi - t.lo < cutoff) {
                            nd_run_local(st, wk, t, leaf);
                            if (closed) { std::lock_guard<std::mutex> lk(mu); closed->emplace_back(t.lo, t.hi); }
                        }
                        else nd_step(st, wk, t, leaf, out);
                        std::lock_guard<std::mutex> lk(mu);
                        for (const NDTask &c : out) pool.push_back(c);
                        busy--;
                    } catch (...) {
                        std::lock_guard<std::mutex> lk(mu);
                        if (!failure) failure = std::current_exception();
                        stop = true;
                        busy--;
                    }
                    cv.notify_all();
                }
            } catch (...) {                         // allocation of the worker's own state
                std::lock_guard<std::mutex> lk(mu);
                if (!failure) failure = std::current_exception();
                stop = true;
                cv.notify_all();
            }
        };
        std::vector<std::thread> th;
        try {
            for (int i = 1; i < nthreads; i++) th.emplace_back(worker);
        } catch (...) {                             // could not start every thread: the ones running finish the work
        }
        worker();
        for (auto &x : th) x.join();
        if (failure) std::rethrow_exception(failure);   // -> KVX_ENOMEM / KVX_EINVAL at the C ABI (abi_guard.hpp, kvx_chol_analyze)
    }
    for (int64_t i = 0; i < n; i++) perm[(size_t)i] = st.verts[(size_t)i];
    if (closed) std::sort(closed->begin(), closed->end());     // (which subdomains were handed over whole does not depend on the threads; their order of completion does)
    if (prof_obj.on && n >= 100000)
        fprintf(stderr, "  dissection: breadth-first passes %.1f ms, separator passes %.1f ms, leaf orderings %.1f ms (summed over threads); first-region chain %.1f ms\n",
                prof_obj.bfs_ns / 1e6, prof_obj.pass_ns / 1e6, prof_obj.leaf_ns / 1e6, prof_obj.chain_ns / 1e6);
    g_ndprof = nullptr;
}

}  // namespace kvx
