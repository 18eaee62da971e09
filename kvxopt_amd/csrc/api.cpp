// C-ABI of libkvxhip.so (include/kvxhip.h): host orchestration of the HIP path.
// There is NO CPU fallback: every numeric entry point needs a HIP device and returns
// KVX_EDEVICE otherwise.
#include "chol_internal.hpp"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <mutex>
#include <vector>

using namespace kvx;

static thread_local std::string g_err;

// ---- launch-graph instantiations in flight (chol_internal.hpp, LazyExec) ------------------------------------------------
namespace {
struct LazyRegistry {
    std::mutex mu;
    std::vector<std::shared_future<hipGraphExec_t>> futs;
    ~LazyRegistry()                                  // static destruction = process exit: nothing of HIP may run behind this point
    {
        std::lock_guard<std::mutex> lk(mu);
        for (auto &f : futs)
            if (f.valid()) f.wait();
    }
};
LazyRegistry &lazy_registry() { static LazyRegistry R; return R; }
}  // namespace
std::atomic<int> &lazy_exec_failures() { static std::atomic<int> n{0}; return n; }
void lazy_exec_track(const std::shared_future<hipGraphExec_t> &f)
{
    LazyRegistry &R = lazy_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    // finished ones go; the list stays as short as the number of instantiations in flight
    R.futs.erase(std::remove_if(R.futs.begin(), R.futs.end(), [](const std::shared_future<hipGraphExec_t> &g) {
                     return !g.valid() || g.wait_for(std::chrono::seconds(0)) == std::future_status::ready; }), R.futs.end());
    R.futs.push_back(f);
}
extern "C" int kvx_graph_instantiate_failures(void) { return lazy_exec_failures().load(); }
namespace kvx { void set_last_error(const std::string &s) { g_err = s; } }   // (also used by lu_api.cpp, dist_api.cpp)


namespace kvx {

// per-level launch plan of the level lists (lists / lptr: fronts grouped by level, each level sorted by kernel class)
void build_plan_from(const Symbolic &S, const std::vector<int32_t> &lists, const std::vector<int64_t> &lptr, std::vector<LevelPlan> &plan)
{
    plan.assign((size_t)S.nlevels, LevelPlan());
    for (int l = 0; l < S.nlevels; l++) {
        LevelPlan &P = plan[l];
        for (int c = 0; c < KVX_NCLS; c++) { P.off[c] = 0; P.cnt[c] = 0; P.maxm[c] = 0; P.maxk[c] = 0; }
        for (int g = 0; g < 3; g++) { P.soff[g] = 0; P.scnt[g] = 0; P.smaxm[g] = 0; }
        for (int64_t q = lptr[l]; q < lptr[l + 1]; q++) {
            int s = lists[q];
            int m = S.sn_m[s], k = S.sn_k[s];
            int c = front_class(m, k);
            if (P.cnt[c] == 0) P.off[c] = q;
            P.cnt[c]++;
            P.maxm[c] = std::max(P.maxm[c], m);
            P.maxk[c] = std::max(P.maxk[c], k);
            if (c == KVX_CLS_BIG) {
                P.big_maxk = std::max(P.big_maxk, k);
                P.chain_maxk = std::max(P.chain_maxk, k);
                P.big_maxu = std::max(P.big_maxu, m - k);
                P.big_flops += (double)k * k * k / 3.0 + (double)(m - k) * k * (double)m;   // potrf + panel solve + trailing update
                P.big_u_len = S.ux[s] + (int64_t)(m - k) * (m - k);
            }
            int g = c == KVX_CLS_BIG ? 0 : (c < KVX_CLS_WAVE0 ? 1 : 2);
            if (P.scnt[g] == 0) P.soff[g] = q;
            P.scnt[g]++;
            P.smaxm[g] = std::max(P.smaxm[g], m);
        }
    }
}

// from F->lists_host / F->lptr_host (the level lists already uploaded to d_lists)
void build_plan(kvx_chol *F) { build_plan_from(F->S, F->lists_host, F->lptr_host, F->plan); }

void destroy_graphs(kvx_chol *F)
{
    F->g_factor.drop();
    for (auto &g : F->g_solve) g.exec.drop();
    F->g_solve.clear();
    for (auto &g : F->g_fused) g.exec.drop();
    F->g_fused.clear();
}

// Leaf subtrees for the solves: maximal subtrees made of wave-class fronts only, small enough for one wavefront
// (front count, pivot columns, LDS stack of update vectors).  Host analysis, once: the update vector of a subtree
// root is written long before its parent's level runs, so it gets a slot of its own behind the recycled part of
// its parity buffer (S.wx / S.wrk_size are adjusted before anything is uploaded).
void analyze_subtrees(kvx_chol *F)
{
    Symbolic &S = F->S;
    const int64_t ns = S.nsuper;
    std::vector<int32_t> cnt((size_t)ns, 1), minidx((size_t)ns);
    std::vector<uint8_t> ok((size_t)ns, 0);
    // fronts per subtree: longer walks serialise more fronts in one wavefront, shorter ones leave more to the level loop (flat
    // optimum 8..16 on the 1e6-unknown systems).  Round 4: a small system has a few hundred subtrees on an idle machine and its
    // sweeps are chains of dependent launches -- a walk of 12 fronts is then the longest link (41 / 56 us of config 4b's 440 us
    // solve); with 4 fronts per walk the loop of config 4b runs at 587-599 it/s against 548-575 (2: 560-598, 3: 513-601, 5: 557-588).
    int maxf = S.n <= 150000 ? 4 : 12;
    { const char *e = getenv("KVX_SUB_MAXF"); if (e) maxf = std::max(1, std::min(atoi(e), KVX_SUB_MAXF)); }
    F->in_sub.assign((size_t)ns, 0);
    F->subs_host.clear();
    F->cd_woff_host.assign(S.children.size(), 0);
    for (int64_t s = 0; s < ns; s++) {
        minidx[s] = (int32_t)s;
        bool good = front_class(S.sn_m[s], S.sn_k[s]) >= KVX_CLS_WAVE0;
        for (int64_t c = S.childptr[s]; c < S.childptr[s + 1]; c++) {
            const int32_t ch = S.children[c];
            good = good && ok[ch];
            cnt[s] += cnt[ch];
            minidx[s] = std::min(minidx[s], minidx[ch]);
        }
        const int64_t lo = s - cnt[s] + 1;
        good = good && cnt[s] <= maxf && minidx[s] == lo && lo >= 0 &&
               (S.super[s + 1] - S.super[lo]) <= KVX_SUB_MAXCOLS;
        ok[s] = good;
    }
    int64_t extra[2] = {0, 0};
    const int64_t base[2] = {S.wrk_size[0], S.wrk_size[1]};
    // factorisation: every front of a subtree gets a slot of its parity buffer that no other front reuses (the level schedule
    // recycles the buffers level by level; subtrees are factored before the level loop, at all depths at once)
    int64_t uextra[2] = {0, 0};
    const int64_t ubase[2] = {S.upd_size[0], S.upd_size[1]};
    // (opt-in, KVX_FACTOR_SUBTREES=1 -- measured slower than the level schedule, see build_subtrees: the slots cost
    // sum u^2 doubles over the subtree fronts, 270 MB on config 2)
    const bool uniq = [] { const char *e = getenv("KVX_FACTOR_SUBTREES"); return e && e[0] == '1'; }();
    for (int64_t s = ns - 1; s >= 0; s--) {
        if (!ok[s] || F->in_sub[s]) continue;
        if (S.sparent[s] >= 0 && ok[S.sparent[s]]) continue;      // not maximal
        const int64_t lo = s - cnt[s] + 1;
        // LDS stack of update vectors in postorder: a front pops its children, then pushes its own
        int64_t sp = 0, top = 0;
        std::vector<int64_t> woff((size_t)cnt[s], 0);
        bool fits = true;
        for (int64_t q = lo; q <= s; q++) {
            for (int64_t c = S.childptr[q]; c < S.childptr[q + 1]; c++) sp -= S.sn_m[S.children[c]] - S.sn_k[S.children[c]];
            woff[q - lo] = sp;
            if (q != s) sp += S.sn_m[q] - S.sn_k[q];
            top = std::max(top, sp);
            if (sp < 0) fits = false;
        }
        if (!fits || top > KVX_SUB_STACK) continue;               // stays in the level lists
        for (int64_t q = lo; q <= s; q++) {
            F->in_sub[q] = 1;
            for (int64_t c = S.childptr[q]; c < S.childptr[q + 1]; c++) F->cd_woff_host[c] = (int32_t)woff[S.children[c] - lo];
            if (uniq) {
                const int pq = S.depth[q] & 1;
                const int64_t uq = S.sn_m[q] - S.sn_k[q];
                S.ux[q] = ubase[pq] + uextra[pq];
                uextra[pq] += uq * uq;
            }
        }
        F->subs_host.push_back(SubDesc{(int32_t)lo, (int32_t)s, (int32_t)S.super[lo], (int32_t)(S.super[s + 1] - S.super[lo])});
        const int p = S.depth[s] & 1;
        S.wx[s] = base[p] + extra[p];
        extra[p] += S.sn_m[s] - S.sn_k[s];
    }
    S.wrk_size[0] = base[0] + extra[0];
    S.wrk_size[1] = base[1] + extra[1];
    S.upd_size[0] = ubase[0] + uextra[0];
    S.upd_size[1] = ubase[1] + uextra[1];
}

// per-level solve lists without the subtree fronts, and the subtree tables, on the device
// Lists for the LDS-staged trailing update (chol_internal.hpp: chain_steps / u_steps), from the plan the factorisation uses.
int build_chain_lists(kvx_chol *F)
{
    Symbolic &S = F->S;
    { const char *e = getenv("KVX_U_BLOCK"); if (e) F->u_block = std::max(64, atoi(e) / 64 * 64); }
    const std::vector<LevelPlan> &plan = F->fplan_on ? F->fplan : F->plan;
    const std::vector<int32_t> &lists = F->fplan_on ? F->flists_host : F->lists_host;
    F->chain_steps.assign((size_t)S.nlevels, {});
    F->u_steps.assign((size_t)S.nlevels, {});
    F->chain_host.clear(); F->chain_m.clear(); F->chain_k.clear();
    std::vector<int32_t> fr;
    for (int l = 0; l < S.nlevels && l < (int)plan.size(); l++) {
        const LevelPlan &P = plan[l];
        const int nbig = P.cnt[KVX_CLS_BIG];
        if (nbig == 0) continue;
        fr.assign(lists.begin() + P.off[KVX_CLS_BIG], lists.begin() + P.off[KVX_CLS_BIG] + nbig);
        auto emit = [&](std::vector<kvx_chol::ChainList> &out, int kb, bool far) {
            // far: the fronts with anything right of column kb + 2 u_block (update matrix included), by the order of that region
            auto region = [&](int32_t f) { return far ? S.sn_m[f] - std::min(kb + 2 * F->u_block, S.sn_k[f]) : S.sn_m[f]; };
            std::vector<int32_t> act;
            for (int32_t f : fr)
                if (S.sn_k[f] > kb && region(f) > 0) act.push_back(f);
            std::stable_sort(act.begin(), act.end(), [&](int32_t a, int32_t b) { return region(a) > region(b); });
            out.push_back(kvx_chol::ChainList{(int64_t)F->chain_host.size(), (int)act.size()});
            for (int32_t f : act) { F->chain_host.push_back(f); F->chain_m.push_back(S.sn_m[f]); F->chain_k.push_back(S.sn_k[f]); }
        };
        for (int jb = 0; jb < P.chain_maxk; jb += KVX_NB) emit(F->chain_steps[l], jb, false);
        for (int kb = 0; kb < P.chain_maxk; kb += F->u_block) emit(F->u_steps[l], kb, true);
    }
    if (F->d_chain) { (void)pool_free(F->d_chain); F->d_chain = nullptr; }
    if (F->chain_host.empty()) return KVX_OK;
    return upload(&F->d_chain, F->chain_host);
}

int build_subtrees(kvx_chol *F)
{
    Symbolic &S = F->S;
    const bool enabled = F->use_subtrees && F->dist_nranks == 1 && (int64_t)S.rel.size() < INT32_MAX;
    std::vector<int32_t> lsw;
    F->sw_off.assign((size_t)S.nlevels, 0);
    F->sw_cnt.assign((size_t)S.nlevels, 0);
    F->sw_kmax.assign((size_t)S.nlevels, 0);
    for (int l = 0; l < S.nlevels; l++) {
        F->sw_off[l] = (int64_t)lsw.size();
        // the LDS-class fronts and the wave-class fronts left outside the subtrees share one launch per level and sweep
        int kmax = 0;
        for (int64_t q = F->lptr_host[l]; q < F->lptr_host[l + 1]; q++) {
            const int32_t f = F->lists_host[q];
            const int c = front_class(S.sn_m[f], S.sn_k[f]);
            if (c == KVX_CLS_BIG || (c >= KVX_CLS_WAVE0 && enabled && F->in_sub[f])) continue;
            if (!enabled && c < KVX_CLS_WAVE0) continue;           // without subtrees: wave fronts only, their own launch
            lsw.push_back(f);
            kmax = std::max(kmax, (int)S.sn_k[f]);
        }
        F->sw_cnt[l] = (int)((int64_t)lsw.size() - F->sw_off[l]);
        F->sw_kmax[l] = kmax;
    }
    if (lsw.empty()) lsw.push_back(0);
    std::vector<SubDesc> subs;
    F->nsub32 = 0;
    F->nsub48 = 0;
    for (int pass = 0; pass < 3; pass++)           // three size groups: largest front of the subtree <= 32, <= 48, larger
        for (const SubDesc &d : F->subs_host) {
            int mm = 0;
            for (int q = d.lo; q <= d.hi; q++) mm = std::max(mm, S.sn_m[q]);
            const int grp = mm <= 32 ? 0 : (mm <= 48 ? 1 : 2);
            if (grp == pass) subs.push_back(d);
            if (grp == pass && pass == 0) F->nsub32++;
            if (grp == pass && pass <= 1) F->nsub48++;
        }
    F->nsub = enabled ? (int)subs.size() : 0;
    F->solve_merged = enabled;
    if (!enabled) { F->nsub32 = 0; F->nsub48 = 0; }
    if (subs.empty()) subs.push_back(SubDesc{0, -1, 0, 0});
    // edge records of the subtree walk: (update rows, offset of the relative indices, LDS stack offset) per tree edge
    std::vector<int32_t> cd_woff(3 * std::max<size_t>(S.children.size(), 1), 0);
    for (int64_t q = 0; q < S.nsuper; q++)
        for (int64_t c = S.childptr[q]; c < S.childptr[q + 1]; c++) {
            const int32_t ch = S.children[c];
            cd_woff[3 * c] = S.sn_m[ch] - S.sn_k[ch];
            cd_woff[3 * c + 1] = (int32_t)(S.rowptr[ch] + S.sn_k[ch]);
            cd_woff[3 * c + 2] = F->cd_woff_host[c];
        }
    int rc;
    for (void *p : {(void *)F->d_subs, (void *)F->d_cd_woff, (void *)F->d_lists_sw, (void *)F->d_depth})
        if (p) (void)pool_free(p);
    F->d_subs = nullptr; F->d_cd_woff = nullptr; F->d_lists_sw = nullptr; F->d_depth = nullptr;
    if ((rc = upload(&F->d_subs, subs))) return rc;
    if ((rc = upload(&F->d_cd_woff, cd_woff))) return rc;
    F->lsw_host = lsw;
    if ((rc = upload(&F->d_lists_sw, lsw))) return rc;
    // factorisation: the subtrees grouped by the LDS image their largest front needs (32 / 48 / 64 rows), and level lists without
    // their fronts (the fplan / d_flists pair the sharded mode uses for its own filtered lists; it keeps subtrees off)
    // Opt-in (KVX_FACTOR_SUBTREES=1).  Measured on MI355X, config 2: the two large groups of subtrees take 0.49 / 0.55 ms side by
    // side and the level loop reaches its first big front at 0.86 ms instead of 0.81; factor 3.69 -> 4.00 ms.  A front costs a
    // wavefront ~30 us under load either way (pivot sweeps are issue-bound FP64, the rest memory latency); the level schedule
    // keeps every front of a level in flight, a walk only one front per subtree.
    { const char *e = getenv("KVX_FACTOR_SUBTREES"); F->factor_subtrees = enabled && F->dist == nullptr && e && e[0] == '1'; }
    F->nsubf[0] = F->nsubf[1] = F->nsubf[2] = 0;
    if (F->factor_subtrees) {
        std::vector<SubDesc> fs;
        for (int g = 0; g < 3; g++)
            for (const SubDesc &d : F->subs_host) {
                int mm = 0;
                for (int q = d.lo; q <= d.hi; q++) mm = std::max(mm, S.sn_m[q]);
                if ((mm <= 32 ? 0 : (mm <= 48 ? 1 : 2)) == g) { fs.push_back(d); F->nsubf[g]++; }
            }
        if (fs.empty()) fs.push_back(SubDesc{0, -1, 0, 0});
        if (F->d_subs_f) { (void)pool_free(F->d_subs_f); F->d_subs_f = nullptr; }
        if ((rc = upload(&F->d_subs_f, fs))) return rc;
        std::vector<int32_t> fl;
        std::vector<int64_t> flp((size_t)S.nlevels + 1, 0);
        for (int l = 0; l < S.nlevels; l++) {
            for (int64_t q = F->lptr_host[l]; q < F->lptr_host[l + 1]; q++)
                if (!F->in_sub[F->lists_host[q]]) fl.push_back(F->lists_host[q]);
            flp[l + 1] = (int64_t)fl.size();
        }
        if (F->d_flists) { (void)pool_free(F->d_flists); F->d_flists = nullptr; }
        if ((rc = upload(&F->d_flists, fl))) return rc;
        build_plan_from(S, fl, flp, F->fplan);
        F->fplan_on = true;
        F->flists_host = fl;
    }
    std::vector<int32_t> dep(S.depth.begin(), S.depth.end());
    if (dep.empty()) dep.push_back(0);
    if ((rc = upload(&F->d_depth, dep))) return rc;
    return build_chain_lists(F);
}

int ensure_device(kvx_chol *F)
{
    if (F->dev_ready) return KVX_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_err("no HIP device visible: the kvxhip numeric path has no CPU fallback");
        return KVX_EDEVICE;
    }
    Symbolic &S = F->S;
    analyze_subtrees(F);
    HIPCHK(pool_stream_get(&F->stream));
    for (int i = 0; i < 4; i++) HIPCHK(pool_event_get(&F->ev[i], true));
    for (int i = 0; i < 4; i++) {
        HIPCHK(pool_stream_get(&F->side[i]));
        HIPCHK(pool_event_get(&F->ev_join[i], false));
    }
    HIPCHK(pool_event_get(&F->ev_fork, false));
    HIPCHK(pool_event_get(&F->ev_fork2, false));
    HIPCHK(pool_event_get(&F->ev_in, false));
    HIPCHK(pool_event_get(&F->ev_out, false));
    int rc;
    std::vector<int32_t> first((size_t)S.nsuper), perm32((size_t)S.n);
    for (int64_t s = 0; s < S.nsuper; s++) first[s] = (int32_t)S.super[s];
    for (int64_t i = 0; i < S.n; i++) perm32[i] = (int32_t)S.perm[i];
    if ((rc = upload(&F->d_k, S.sn_k))) return rc;
    if ((rc = upload(&F->d_m, S.sn_m))) return rc;
    if ((rc = upload(&F->d_first, first))) return rc;
    if ((rc = upload(&F->d_rowidx, S.rowidx))) return rc;
    if ((rc = upload(&F->d_rel, S.rel))) return rc;
    if ((rc = upload(&F->d_children, S.children))) return rc;
    if ((rc = upload(&F->d_perm, perm32))) return rc;
    if ((rc = upload(&F->d_lists, S.levellist))) return rc;
    std::vector<int64_t> px(S.px.begin(), S.px.end());
    if ((rc = upload(&F->d_px, px))) return rc;
    if ((rc = upload(&F->d_rowptr, S.rowptr))) return rc;
    if ((rc = upload(&F->d_ux, S.ux))) return rc;
    if ((rc = upload(&F->d_wx, S.wx))) return rc;
    if ((rc = upload(&F->d_childptr, S.childptr))) return rc;
    if ((rc = upload(&F->d_amap, S.amap))) return rc;
    if (S.nnzA < INT32_MAX && F->part.empty() && !getenv("KVX_INIT_TWO_PASSES")) {
        // the scatter map once more, grouped by the chunk of the factor an entry goes to: k_init_factor zeroes L and scatters A in one
        // pass.  A counting sort ON THE DEVICE over the map just uploaded (histogram, the scan of the ~10^5 chunk counters on the host,
        // placement) -- on the host it was 12-18 ms of every first call on a new pattern with 3 M entries, more than a thousand of the
        // steps it speeds up by 0.05 ms would give back.  (Sharded factors keep the two launches: their layout is trimmed afterwards.)
        const int sh = init_factor_shift();
        const int64_t nchunk = std::max<int64_t>((S.lsize + ((int64_t)1 << sh) - 1) >> sh, 1);
        int64_t *d_cnt = nullptr;
        HIPCHK(pool_malloc((void **)&F->d_scptr, (size_t)(nchunk + 1) * sizeof(int64_t)));
        HIPCHK(pool_malloc((void **)&d_cnt, (size_t)(nchunk + 1) * sizeof(int64_t)));
        HIPCHK(pool_malloc((void **)&F->d_sdst, (size_t)std::max<int64_t>(S.nnzA, 1) * sizeof(int64_t)));
        HIPCHK(pool_malloc((void **)&F->d_ssrc, (size_t)std::max<int64_t>(S.nnzA, 1) * sizeof(int32_t)));
        HIPCHK(hipMemsetAsync(d_cnt, 0, (size_t)(nchunk + 1) * sizeof(int64_t), nullptr));
        launch_scatter_group_count(nullptr, F->d_amap, S.nnzA, sh, d_cnt);
        std::vector<int64_t> cptr((size_t)nchunk + 1);
        HIPCHK(hipMemcpy(cptr.data(), d_cnt, (size_t)(nchunk + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
        // slot q + 1 counted chunk q: the running sum turns the slots into "entries in the chunks before q", the start of chunk q
        std::vector<int64_t> start((size_t)nchunk + 1);
        int64_t run = 0;
        for (int64_t q = 0; q < nchunk; q++) { start[(size_t)q] = run; run += cptr[(size_t)q + 1]; }
        start[(size_t)nchunk] = run;
        HIPCHK(hipMemcpy(F->d_scptr, start.data(), (size_t)(nchunk + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_cnt, start.data(), (size_t)(nchunk + 1) * sizeof(int64_t), hipMemcpyHostToDevice));          // the cursors
        launch_scatter_group_place(nullptr, F->d_amap, S.nnzA, sh, d_cnt, F->d_sdst, F->d_ssrc);
        HIPCHK(hipDeviceSynchronize());
        (void)pool_free(d_cnt);
        F->scnt = run;
    }
    HIPCHK(pool_malloc((void **)&F->d_Lx, (std::max<int64_t>(S.lsize, 1) + 2) * sizeof(double)));   // + 2: k_syrk_lds reads row pairs (16-byte loads at clamped rows)
    for (int p = 0; p < 2; p++)
        HIPCHK(pool_malloc((void **)&F->d_U[p], std::max<int64_t>(S.upd_size[p], 1) * sizeof(double)));
    HIPCHK(pool_malloc((void **)&F->d_Ax, std::max<int64_t>(S.nnzA, 1) * sizeof(double)));
    HIPCHK(pool_malloc((void **)&F->d_status, sizeof(int)));
    HIPCHK(hipHostMalloc((void **)&F->h_status, sizeof(int), hipHostMallocMapped));
    *F->h_status = 0x7f7f7f7f;
    if (hipHostGetDevicePointer((void **)&F->h_status_dev, F->h_status, 0) != hipSuccess) { (void)hipGetLastError(); F->h_status_dev = nullptr; }
    std::vector<int64_t> &loff_host = F->linv_off_host;
    {
        // inverted diagonal blocks of the big fronts: ceil(k/NB) blocks of NB x NB each
        std::vector<int64_t> &loff = loff_host;
        loff.assign((size_t)S.nsuper, -1);
        int64_t tot = 0;
        for (int64_t s = 0; s < S.nsuper; s++)
            if (front_class(S.sn_m[s], S.sn_k[s]) == KVX_CLS_BIG && (F->part.empty() || F->part[s])) { loff[s] = tot; tot += (int64_t)((S.sn_k[s] + KVX_NB - 1) / KVX_NB) * KVX_NB * KVX_NB; }
        if ((rc = upload(&F->d_linv_off, loff))) return rc;
        HIPCHK(pool_malloc((void **)&F->d_Linv, std::max<int64_t>(tot, 1) * sizeof(double)));
        F->dev_bytes = (S.lsize + S.upd_size[0] + S.upd_size[1] + S.nnzA + tot) * (int64_t)sizeof(double);
    }
    {
        std::vector<FrontDesc> fd((size_t)S.nsuper);
        std::vector<ChildDesc> cd(S.children.size());
        std::vector<int32_t> tiles;
        for (int64_t s = 0; s < S.nsuper; s++) {
            FrontDesc &d = fd[s];
            d.k = S.sn_k[s]; d.m = S.sn_m[s]; d.first = (int32_t)S.super[s];
            d.nchild = (int32_t)(S.childptr[s + 1] - S.childptr[s]);
            d.px = S.px[s]; d.rowptr = S.rowptr[s]; d.ux = S.ux[s]; d.wx = S.wx[s]; d.childptr = S.childptr[s];
            d.linv = loff_host[s];
            for (int64_t c = S.childptr[s]; c < S.childptr[s + 1]; c++) {
                int32_t ch = S.children[c];
                ChildDesc &e = cd[c];
                e.kc = S.sn_k[ch]; e.uc = S.sn_m[ch] - S.sn_k[ch];
                e.rel = S.rowptr[ch] + S.sn_k[ch]; e.ux = S.ux[ch]; e.wx = S.wx[ch];
                e.tile = -1;
                if (front_class(d.m, d.k) == KVX_CLS_BIG) {
                    // tiles[x] = first update column j of the child with rel[j] >= x * KVX_ASM_TC
                    e.tile = (int64_t)tiles.size();
                    const int ntile = (d.m + KVX_ASM_TC - 1) / KVX_ASM_TC;
                    const int32_t *rl = S.rel.data() + e.rel;
                    int j = 0;
                    for (int x = 0; x <= ntile; x++) {
                        while (j < e.uc && rl[j] < x * KVX_ASM_TC) j++;
                        tiles.push_back(j);
                    }
                }
            }
        }
        if ((rc = upload(&F->d_fd, fd))) return rc;
        if ((rc = upload(&F->d_cd, cd))) return rc;
        if ((rc = upload(&F->d_tiles, tiles))) return rc;
    }
    F->ds = DevSym{F->d_k, F->d_m, F->d_first, F->d_px, F->d_rowptr, F->d_rowidx, F->d_rel,
                   F->d_ux, F->d_wx, F->d_childptr, F->d_children, F->d_linv_off, F->d_fd, F->d_cd, F->d_tiles, 0.0, 0.0};
    if (F->opts.dbound > 0.0) {
        // cholmod.options['dbound'] (cholmod.c:116-117; CHOLMOD: "entries of L_kk smaller than dbound are replaced by dbound").
        // reserved[3] = 1: replace by 1e64 instead -- the row drops out of the solves (the customary cure for normal equations
        // A D A' that lose rank numerically near the end of an interior-point run; used by lp.KKTDiagEqDev).
        F->ds.piv_floor = F->opts.dbound * F->opts.dbound;
        F->ds.piv_repl = F->opts.reserved[3] == 1 ? 1e128 : F->ds.piv_floor;
    }
    F->lists_host = S.levellist;
    F->lptr_host = S.levelptr;
    build_plan(F);
    { const char *e = getenv("KVX_NO_SUBTREES"); F->use_subtrees = !(e && e[0] == '1'); }
    if ((rc = build_subtrees(F))) return rc;
    { const char *e = getenv("KVX_NO_GRAPH"); F->use_graph = !(e && e[0] == '1'); }
    { const char *e = getenv("KVX_WIDE_FROM"); if (e) F->wide_from = std::max(0, atoi(e)); }
    { const char *e = getenv("KVX_SIDE_SPREAD"); if (e) F->side_spread = atoi(e); }
    { const char *e = getenv("KVX_TWO_LEVEL_M"); if (e) F->two_level_m = atoi(e); }
    { const char *e = getenv("KVX_OUTER_BLOCK"); if (e && atoi(e) >= 64) F->outer_block = atoi(e) / 64 * 64; }
    F->dev_ready = true;
    return KVX_OK;
}

int ensure_solve_ws(kvx_chol *F, int64_t nrhs)
{
    if (nrhs <= F->x_cap) return KVX_OK;
    Symbolic &S = F->S;
    // room for two right-hand sides from the start: an interior-point loop solves with one at its starting point and with two
    // inside the iteration, and growing the workspace drops the captured sweeps -- which WAITS for an instantiation in flight
    // (10-20 ms inside the first iteration of a first call, measured)
    nrhs = std::max<int64_t>(nrhs, 2);
    for (auto &g : F->g_solve) g.exec.drop();
    F->g_solve.clear();                          // the captured sweeps point into the old workspace
    for (auto &g : F->g_fused) g.exec.drop();
    F->g_fused.clear();
    if (F->d_X) { (void)pool_free(F->d_X); F->d_X = nullptr; }
    if (F->d_X0) { (void)pool_free(F->d_X0); F->d_X0 = nullptr; }
    if (F->d_WK) { (void)pool_free(F->d_WK); F->d_WK = nullptr; }
    for (int p = 0; p < 2; p++)
        if (F->d_W[p]) { (void)pool_free(F->d_W[p]); F->d_W[p] = nullptr; }
    F->x_cap = 0;
    HIPCHK(pool_malloc((void **)&F->d_X, std::max<int64_t>(S.n * nrhs, 1) * sizeof(double)));
    HIPCHK(pool_malloc((void **)&F->d_X0, std::max<int64_t>(S.n * nrhs, 1) * sizeof(double)));
    HIPCHK(pool_malloc((void **)&F->d_WK, std::max<int64_t>(S.n * nrhs, 1) * sizeof(double)));
    const int64_t wmax = std::max(S.wrk_size[0], S.wrk_size[1]);   // common per-rhs stride of both parity buffers
    for (int p = 0; p < 2; p++)
        HIPCHK(pool_malloc((void **)&F->d_W[p], std::max<int64_t>(wmax * nrhs, 1) * sizeof(double)));
    F->x_cap = nrhs;
    return KVX_OK;
}

// Device-pointer entry points: the caller's producers run on the legacy null stream (the kvx_nt_* /
// kvx_atda_* / kvx_spmv_* calls, torch's default stream); the factor's stream is non-blocking, so
// order it explicitly behind them.
int wait_for_caller(kvx_chol *F)
{
    HIPCHK(hipEventRecord(F->ev_in, nullptr));
    HIPCHK(hipStreamWaitEvent(F->stream, F->ev_in, 0));
    return KVX_OK;
}

// enqueue the numeric factorisation; d_Ax already holds the values
// levels lfrom, lfrom - 1, ..., lto; prologue = zero L, reset the status word, scatter A; epilogue = fetch the status
int enqueue_factor_body(kvx_chol *F, int lfrom, int lto, bool prologue, bool epilogue)
{
    Symbolic &S = F->S;
    hipStream_t st = F->stream;
    if (prologue) {
        // Plain kernels, not memset nodes: replayed from a captured graph under the HIP runtime that ships inside the PyTorch wheel
        // (7.0.51831, the one a process gets once torch is imported), the memset nodes of a SMALL factor were not ordered before
        // the kernels behind them -- a dense 200 x 200 K of misc.kkt_chol2 failed at column 0 in 28 of 30 replays
        // (scratch/graph_stress.py; ROCm 7.2's own runtime replays them correctly).  KVX_DBG_MEMSET_NODES=1 restores the nodes.
        if (F->d_scptr && !getenv("KVX_DBG_MEMSET_NODES")) {
            ProfScope ps(F, FAM_SCATTER);          // zero L, reset the status word and scatter A: one pass over L
            launch_init_factor(st, F->d_Ax, F->d_ssrc, F->d_sdst, F->d_scptr, S.lsize, F->d_Lx, F->d_status);
        } else {
            if (getenv("KVX_DBG_MEMSET_NODES")) {
                HIPCHK(hipMemsetAsync(F->d_Lx, 0, std::max<int64_t>(S.lsize, 1) * sizeof(double), st));
                HIPCHK(hipMemsetAsync(F->d_status, 0x7f, sizeof(int), st));   // 0x7f7f7f7f = "no failing column"
            } else {
                launch_clear_factor(st, F->d_Lx, S.lsize, F->d_status);
            }
            ProfScope ps(F, FAM_SCATTER);
            launch_scatter_a(st, F->d_Ax, F->d_amap, S.nnzA, F->d_Lx);
        }
    }
    if (lfrom < 0) lfrom = S.nlevels - 1;
    if (F->factor_subtrees && prologue && lfrom == S.nlevels - 1) {
        // the leaf subtrees: one wavefront each, all depths at once, the three LDS sizes on three streams
        const int caps[3] = {32, 48, 64};
        hipStream_t ss[3] = {st, F->side[0], F->side[1]};
        const bool fork[3] = {false, F->nsubf[1] > 0, F->nsubf[2] > 0};
        if (fork[1] || fork[2]) {
            HIPCHK(hipEventRecord(F->ev_fork, st));
            for (int g = 1; g < 3; g++)
                if (fork[g]) HIPCHK(hipStreamWaitEvent(ss[g], F->ev_fork, 0));
        }
        int64_t off = 0;
        // (largest images first: they hold the fewest subtrees per CU)
        int64_t offs[3] = {0, F->nsubf[0], F->nsubf[0] + F->nsubf[1]};
        (void)off;
        for (int g = 2; g >= 0; g--) {
            if (F->nsubf[g] == 0) continue;
            ProfScope ps(F, FAM_SMALL, ss[g]);
            launch_factor_subtree(ss[g], caps[g], F->ds, F->d_subs_f + offs[g], F->nsubf[g], F->d_depth, F->d_Lx, F->d_U[0], F->d_U[1], F->d_status);
        }
        for (int g = 1; g < 3; g++)
            if (fork[g]) { HIPCHK(hipEventRecord(F->ev_join[g - 1], ss[g])); HIPCHK(hipStreamWaitEvent(st, F->ev_join[g - 1], 0)); }
    }
    const int32_t *lbase = F->fplan_on ? F->d_flists : F->d_lists;
    for (int l = lfrom; l >= lto; l--) {
        const LevelPlan &P = F->fplan_on ? F->fplan[l] : F->plan[l];
        double *Uout = F->d_U[l & 1];
        const double *Uch = F->d_U[(l + 1) & 1];
        // The fronts of one level are independent.  The big-front chain keeps the main stream; the small-
        // front launches (two LDS classes, three wave row capacities) are each latency-bound by their
        // slowest front, so they are spread over the streams by estimated duration (longest first onto
        // the least loaded stream) instead of queueing up: two side streams beside a big chain, main +
        // two side streams on the levels without big fronts.  Joined at level end.
        const bool have_big = P.cnt[KVX_CLS_BIG] > 0;
        struct Item { int c; int cnt; int64_t off; double est; int stream; int mcap; int kmax; };
        Item items[5];
        int nitems = 0;
        for (int c = KVX_CLS_LDS128; c < KVX_CLS_WAVE0; c++)
            if (P.cnt[c] > 0) {
                const double slots = c == KVX_CLS_LDS128 ? 512.0 : 1024.0;         // packed LDS image: two / four fronts per CU
                items[nitems++] = Item{c, P.cnt[c], P.off[c], (P.maxk[c] > 32 ? 80.0 : 45.0) * std::max(1.0, P.cnt[c] / slots), 0, 0, 0};
            }
        for (int c = KVX_CLS_WAVE0; c < KVX_NCLS; c += 2) {
            const int cnt = P.cnt[c] + P.cnt[c + 1];
            if (cnt == 0) continue;
            const int mcap = wave_class_mcap(c);
            const double base = mcap == 64 ? 35.0 : (mcap == 48 ? 28.0 : 18.0), slots = mcap == 64 ? 2048.0 : (mcap == 48 ? 4096.0 : 8192.0);
            items[nitems++] = Item{c, cnt, P.cnt[c] > 0 ? P.off[c] : P.off[c + 1], base * std::max(1.0, cnt / slots), 0, mcap, P.cnt[c] > 0 ? 32 : 16};
        }
        std::sort(items, items + nitems, [](const Item &x, const Item &y) { return x.est > y.est; });
        double load[3] = {have_big ? 1e30 : 0.0, 0.0, 0.0};                          // main, side[0], side[1]
        if (F->side_spread == 0) load[0] = have_big ? 1e30 : -1e30;                 // KVX_SIDE_SPREAD=0: everything small on one stream
        bool side_used[3] = {false, false, false};
        for (int i = 0; i < nitems; i++) {
            int best = 0;
            for (int t = 1; t < 3; t++)
                if (load[t] < load[best]) best = t;
            if (F->side_spread == 0) best = have_big ? 1 : 0;
            load[best] += items[i].est;
            items[i].stream = best;
            if (best > 0) side_used[best - 1] = true;
        }
        if (side_used[0] || side_used[1]) {
            HIPCHK(hipEventRecord(F->ev_fork, st));
            for (int i = 0; i < 2; i++)
                if (side_used[i]) HIPCHK(hipStreamWaitEvent(F->side[i], F->ev_fork, 0));
        }
        for (int i = 0; i < nitems; i++) {
            const Item &it = items[i];
            hipStream_t sl = it.stream == 0 ? st : F->side[it.stream - 1];
            ProfScope ps(F, FAM_SMALL, sl);
            if (it.c < KVX_CLS_WAVE0)
                launch_front_small(sl, it.c == KVX_CLS_LDS128 ? 128 : 96, P.maxk[it.c] <= 32 ? 32 : 64, F->ds, lbase + it.off, it.cnt, F->d_Lx, Uch, Uout, F->d_status);
            else    // the k <= 32 and k <= 16 lists of one row capacity are adjacent -> one launch
                launch_front_wave(sl, it.mcap, it.kmax, F->ds, lbase + it.off, it.cnt, F->d_Lx, Uch, Uout, F->d_status);
        }
        const int nchain = P.cnt[KVX_CLS_BIG];
        // Few workgroups in the extend-add (the top of the tree, small systems): the first diagonal block is assembled and factored
        // by a workgroup of the same launch (k_assemble_big_potrf) -- one launch less on the level's critical path.
        // KVX_ASM_POTRF_WGS = largest extend-add launch (workgroups) that takes this form; 0 = never.
        static const int64_t asm_potrf_wgs = [] { const char *e = getenv("KVX_ASM_POTRF_WGS"); return e ? atoll(e) : (int64_t)1024; }();
        const bool asm_potrf = have_big && nchain > 0 &&
                               (int64_t)P.cnt[KVX_CLS_BIG] * ((P.maxm[KVX_CLS_BIG] + KVX_ASM_TC - 1) / KVX_ASM_TC) <= asm_potrf_wgs;
        if (have_big) {                                     // extend-add of every big front of the level, one launch
            ProfScope ps(F, FAM_ASSEMBLE);
            if (asm_potrf)
                launch_assemble_big_potrf(st, F->ds, lbase + P.off[KVX_CLS_BIG], P.cnt[KVX_CLS_BIG], P.maxm[KVX_CLS_BIG], F->d_Lx, Uch, Uout,
                                          F->d_Linv, F->d_status);
            else
                launch_assemble_big(st, F->ds, lbase + P.off[KVX_CLS_BIG], P.cnt[KVX_CLS_BIG], P.maxm[KVX_CLS_BIG], F->d_Lx, Uch, Uout);
        }
        if (nchain > 0) {
            const int nbig = nchain, bigm = P.maxm[KVX_CLS_BIG];
            const int32_t *list = lbase + P.off[KVX_CLS_BIG];
            if (!asm_potrf) { ProfScope ps(F, FAM_POTRF); launch_potrf_blk(st, F->ds, list, nbig, 0, F->d_Lx, F->d_Linv, F->d_status); }
            if (bigm >= F->two_level_m && (getenv("KVX_DEFER_U") ? atoi(getenv("KVX_DEFER_U")) == 0 : false)) {
                // outer blocks of `outer_block` (1024) columns, one rank-1024 update of the trailing matrix per block (128-tile kernel: 34 TF/s
                // on a dense trailing matrix; rocBLAS dgemm at K = 256 reaches 48-59).  Measured on MI355X against the
                // single-level path: dense n = 10240 14.8 vs 16.7 ms, 3-D 80^3 49.6 vs 51.0 ms, but 21-point 1000^2
                // (fronts <= 5007, many per level) 29.1 vs 24.3 ms -- the outer update is an extra serial launch per
                // block, so it is used for very large fronts only (look-ahead -- the outer update of block b beside the panel chain
                // of block b + 1 on a second stream -- was measured slower, docs/lab.md)
                const int OB = F->outer_block;
                for (int ob = 0; ob < P.chain_maxk; ob += OB) {
                    for (int jb = ob; jb < std::min(ob + OB, P.chain_maxk); jb += KVX_NB) {
                        { ProfScope ps(F, FAM_TRSM); launch_trsm_blk(st, F->ds, list, nbig, bigm, jb, F->d_Lx, F->d_Linv); }
                        { ProfScope ps(F, FAM_SYRK); launch_syrk_inner(st, F->ds, list, nbig, bigm, jb, ob + OB, F->d_Lx, Uout, F->d_Linv, F->d_status); }
                    }
                    { ProfScope ps(F, FAM_SYRK); launch_syrk_outer(st, F->ds, list, nbig, bigm, ob, OB, F->d_Lx, Uout, F->d_Linv, F->d_status); }
                }
            } else {
                // Pair schedule where a launch holds many tiles (the levels bound by the read-modify-write of the trailing matrices): panel
                // jb updates only the columns of panel jb + 64 (one tile column), panel jb + 64 is solved, and ONE pass over everything
                // right of both applies the two panels together -- half the passes over the trailing matrices, the same number of
                // launches.  Where a launch is a handful of tiles (the pivot chain at the top of the tree) the pass over C is not what the
                // step waits for and the eight operand rounds of a K = 128 tile would lengthen the chain: one panel per launch there.
                // KVX_PAIR_TILES = tile count (upper estimate: largest front x fronts in the launch) from which on pairs are used.
                static const int64_t pair_tiles = [] { const char *e = getenv("KVX_PAIR_TILES"); return e ? atoll(e) : (int64_t)3000; }();
                // Round 4: the update matrices are left out of the chain (launches limited to the pivot columns) wherever a level's
                // fronts have more than one panel, and brought up to date afterwards by rank-(<= u_block) updates with LDS-staged
                // tiles (launch_syrk_u): K = 64 per pass over C moved 16 bytes per 128 flops and bound the ~20-nnz/row systems by
                // exactly that traffic.  KVX_DEFER_U=0: the round-3 schedule; KVX_U_BLOCK: panel columns per pass (default 256).
                static const int defer_u = [] { const char *e = getenv("KVX_DEFER_U"); return e ? atoi(e) : 1; }();
                static const bool direct = [] { const char *e = getenv("KVX_SYRK_DIRECT"); return e && e[0] == '1'; }();
                const bool have_lists = (size_t)l < F->chain_steps.size() && (int)F->chain_steps[(size_t)l].size() * KVX_NB >= P.chain_maxk &&
                                        (int)F->u_steps[(size_t)l].size() * F->u_block >= P.chain_maxk;
                // ... where a level's chain is bound by throughput: its flops per panel step would keep the machine busy for longer
                // than the ~30 us of latency a step has anyway (KVX_BLOCKED_GF: Gflop per step from which on, default 0.5.  21-point
                // system, one box, thresholds 0 / 0.3 / 0.7 / 1.5 / never: 23.6 / 23.8 / 23.9 / 24.4 / 26.3 ms; config 2, where no level
                // reaches 0.3: blocked everywhere 4.97 - 5.45 ms against 4.76 - 4.97)
                static const double blocked_gf = [] { const char *e = getenv("KVX_BLOCKED_GF"); return e ? atof(e) : 0.5; }();
                const int nsteps = (P.chain_maxk + KVX_NB - 1) / KVX_NB;
                const bool blocked = defer_u && !direct && have_lists && P.big_flops * 1e-9 >= blocked_gf * nsteps;
                const bool cls = !direct && have_lists;
                // one trailing update of the chain: the fronts still in it at step jb, numbered over size classes (LDS-staged tiles),
                // or the round-3 launches over (tiles of the largest front) x (all big fronts of the level)
                auto syrk = [&](int jb, int klen, int col_lim) {
                    ProfScope ps(F, FAM_SYRK);
                    if (cls) {
                        const kvx_chol::ChainList &cl = F->chain_steps[(size_t)l][(size_t)(jb / KVX_NB)];
                        launch_syrk_step(st, F->ds, F->d_chain + cl.off, F->chain_m.data() + cl.off, F->chain_k.data() + cl.off, cl.cnt, jb, klen,
                                         F->d_Lx, Uout, F->d_Linv, F->d_status, col_lim);
                    } else if (klen == 2 * KVX_NB) {
                        launch_syrk_pair(st, F->ds, list, nbig, bigm, jb, F->d_Lx, Uout, F->d_Linv, F->d_status, col_lim);
                    } else if (col_lim < KVX_COLS_PIVOT) {
                        launch_syrk_inner(st, F->ds, list, nbig, bigm, jb, col_lim, F->d_Lx, Uout, F->d_Linv, F->d_status);
                    } else {
                        launch_syrk_trailing(st, F->ds, list, nbig, bigm, jb, F->d_Lx, Uout, F->d_Linv, F->d_status, col_lim);
                    }
                };
                if (blocked) {
                    // Round 4, the blocked schedule.  The pivot columns go in blocks of OB = u_block (256).  On the chain's stream a
                    // panel of block b updates only what is left of the block ("inner", at most three tile columns, K = 64); when the
                    // block is solved, ONE rank-OB update ("near") brings the next block's columns up to date and factors its first
                    // diagonal block; everything further right -- later pivot columns and the update matrix -- gets its rank-OB update
                    // ("far") on a stream of its own, beside the next block's chain: K = 64 per pass over C moved 16 bytes per 128
                    // flops and bound the ~20-nnz/row systems by exactly that traffic.  near(b + 1) and far(b) meet in the columns of
                    // block b + 2: near(b + 1) waits for far(b); far(b + 1) follows far(b) on its stream.
                    // KVX_DEFER_U=0: the round-3 schedule; KVX_U_STREAM=0: the far updates on the chain's stream.
                    static const int u_stream = [] { const char *e = getenv("KVX_U_STREAM"); return e ? atoi(e) : 1; }();
                    const int OB = F->u_block;
                    // (family timing sums the durations of single launches: the far updates stay behind the chain then, so that a launch's
                    //  duration is its own and not that of two kernels sharing the machine)
                    hipStream_t su = (u_stream && F->prof_family < 0) ? F->side[2] : st;
                    bool forked = false;
                    for (int ob = 0, b = 0; ob < P.chain_maxk; ob += OB, b++) {
                        const int bend = std::min(ob + OB, P.chain_maxk);
                        for (int jb = ob; jb < bend; jb += KVX_NB) {
                            { ProfScope ps(F, FAM_TRSM); launch_trsm_blk(st, F->ds, list, nbig, bigm, jb, F->d_Lx, F->d_Linv); }
                            if (jb + KVX_NB < bend) syrk(jb, KVX_NB, ob + OB);
                        }
                        const kvx_chol::ChainList &fl = F->u_steps[(size_t)l][(size_t)b];
                        if (fl.cnt > 0) {
                            if (su != st) {
                                while ((int)F->ev_u.size() <= 2 * b + 1) { hipEvent_t e = nullptr; HIPCHK(pool_event_get(&e, false)); F->ev_u.push_back(e); }
                                HIPCHK(hipEventRecord(F->ev_u[(size_t)(2 * b)], st));
                                HIPCHK(hipStreamWaitEvent(su, F->ev_u[(size_t)(2 * b)], 0));
                                forked = true;
                            }
                            {
                                ProfScope ps(F, FAM_SYRK, su);
                                launch_syrk_far(su, F->ds, F->d_chain + fl.off, F->chain_m.data() + fl.off, F->chain_k.data() + fl.off, fl.cnt, ob, OB,
                                                ob + 2 * OB, F->d_Lx, Uout);
                            }
                            if (su != st) HIPCHK(hipEventRecord(F->ev_u[(size_t)(2 * b + 1)], su));
                        }
                        if (bend < P.chain_maxk) {
                            // near(b) touches the columns of block b + 1, which far(b - 1) has updated with block b - 1
                            if (su != st && b >= 1 && F->u_steps[(size_t)l][(size_t)(b - 1)].cnt > 0)
                                HIPCHK(hipStreamWaitEvent(st, F->ev_u[(size_t)(2 * (b - 1) + 1)], 0));
                            syrk(ob, OB, ob + 2 * OB);
                        }
                    }
                    if (forked) {
                        if (!F->ev_ujoin) HIPCHK(pool_event_get(&F->ev_ujoin, false));
                        HIPCHK(hipEventRecord(F->ev_ujoin, su));
                        HIPCHK(hipStreamWaitEvent(st, F->ev_ujoin, 0));
                    }
                } else {
                    for (int jb = 0; jb < P.chain_maxk;) {
                        const int64_t T = (bigm - jb - 1 + KVX_TILE - 1) / KVX_TILE;
                        const bool pair = jb + KVX_NB < P.chain_maxk && T * (T + 1) / 2 * nbig >= pair_tiles;
                        // the trailing update of panel jb also factors and inverts the diagonal block of panel jb + 64
                        { ProfScope ps(F, FAM_TRSM); launch_trsm_blk(st, F->ds, list, nbig, bigm, jb, F->d_Lx, F->d_Linv); }
                        if (!pair) {
                            syrk(jb, KVX_NB, INT_MAX);
                            jb += KVX_NB;
                            continue;
                        }
                        syrk(jb, KVX_NB, jb + 2 * KVX_NB);
                        { ProfScope ps(F, FAM_TRSM); launch_trsm_blk(st, F->ds, list, nbig, bigm, jb + KVX_NB, F->d_Lx, F->d_Linv); }
                        syrk(jb, 2 * KVX_NB, INT_MAX);
                        jb += 2 * KVX_NB;
                    }
                }
            }
        }
        for (int i = 0; i < 3; i++)
            if (side_used[i]) { HIPCHK(hipEventRecord(F->ev_join[i], F->side[i])); HIPCHK(hipStreamWaitEvent(st, F->ev_join[i], 0)); }
        if (F->pipe_on) {
            // kvx_chol_factorize_solve_dev: level l is complete -- its forward sweep goes onto the sweep's own streams right here, so
            // that its launches sit between the factorisation's in submission order too (enqueued after the whole factorisation they
            // were submitted -- eagerly and from a replayed graph alike -- only when the last front had been)
            HIPCHK(hipEventRecord(F->ev_lvl[(size_t)l], st));
            // The sweep goes onto side[0] -- the stream of the factorisation's small-front launches, which has nothing left to do at the
            // top of the tree -- not onto a stream of its own: a replayed graph runs its parallel branches on streams the executable
            // creates for itself, as many as the capture is wide, and a process gets four hardware queues; with a fourth / fifth
            // branch two of them share a queue and the step was 4.65 or 4.9 ms from one process to the next, depending on whether the
            // pivot chain's queue was the shared one.  KVX_PIPE_OWN_STREAM=1: side[2] (the old form, for comparison).
            static const bool own = [] { const char *e = getenv("KVX_PIPE_OWN_STREAM"); return e && e[0] == '1'; }();
            hipStream_t sw = own ? F->side[2] : F->side[0];
            const SweepStreams ss2{sw, sw, sw, F->ev_pipe[1], F->ev_pipe[2], F->ev_pipe[3]};
            // Below pipe_from the levels hold thousands of small fronts that fill the CUs: a sweep beside them only takes their
            // wavefront slots (measured: the factorisation lost what the sweep gained).  From pipe_from up the factorisation is a chain
            // of small launches on an idle machine: the sweep of everything below starts there in one go, then follows level by level.
            if (l == F->pipe_from) enqueue_fwd(F, F->d_X, S.n, F->pipe_nr, S.nlevels - 1, l, &ss2, true, true);
            else if (l < F->pipe_from) enqueue_fwd(F, F->d_X, S.n, F->pipe_nr, l, l, &ss2, true);
        }
    }
    if (epilogue) {
        if (F->h_status_dev && !getenv("KVX_DBG_MEMSET_NODES")) launch_publish_status(st, F->d_status, F->h_status_dev);
        else HIPCHK(hipMemcpyAsync(F->h_status, F->d_status, sizeof(int), hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipGetLastError());
    return KVX_OK;
}

// Capture `body` (which enqueues on F->stream and, through events, on the side streams) into an
// executable graph.  Returns nullptr (and leaves the stream usable) if capture is not possible.
template <class Body>
void capture_graph(kvx_chol *F, Body body, LazyExec &out)
{
    out.tried = true;
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(F->stream, hipStreamCaptureModeRelaxed) != hipSuccess) { (void)hipGetLastError(); return; }
    int rc = body();
    hipError_t e = hipStreamEndCapture(F->stream, &graph);
    if (rc != KVX_OK || e != hipSuccess || !graph) { (void)hipGetLastError(); if (graph) (void)hipGraphDestroy(graph); return; }
    if (const char *dot = getenv("KVX_DBG_GRAPH_DOT")) {           // debugging: the captured graph (nodes and edges) as a .dot file
        static int serial = 0;
        const std::string path = std::string(dot) + "." + std::to_string(serial++) + ".dot";
        if (hipGraphDebugDotPrint(graph, path.c_str(), 0) != hipSuccess) (void)hipGetLastError();
    }
    out.start(graph);                              // instantiation on a thread of its own; the graph is destroyed there
}

int enqueue_factor(kvx_chol *F)
{
    if (F->dist_nranks > 1) { set_err("sharded factor: use kvx_chol_dist_factorize"); return KVX_EINVAL; }
    hipStream_t st = F->stream;
    HIPCHK(hipEventRecord(F->ev[0], st));
    const char *dbg_ng = getenv("KVX_DBG_NO_FACTOR_GRAPH");      // debugging: "1" = no factor graphs at all, "<n>" = none for factors of order n
    const bool graph_ok = F->use_graph && F->prof_family < 0 && !(dbg_ng && (atoll(dbg_ng) == 1 || atoll(dbg_ng) == F->S.n));
    F->factor_calls++;
    F->diag_valid = false;
    // (a capture records the launches without running them: the call that takes it still runs its own launches below)
    if (graph_ok && !F->g_factor.tried && F->factor_calls >= 2)
        capture_graph(F, [&] { return enqueue_factor_body(F); }, F->g_factor);   // (sharded mode drives the body itself)
    hipGraphExec_t fexec = graph_ok ? F->g_factor.ready() : nullptr;
    if (fexec) {
        if (const char *e = getenv("KVX_DBG_GRAPH_SYNC")) { if (atoi(e) & 1) HIPCHK(hipStreamSynchronize(st)); }
        HIPCHK(hipGraphLaunch(fexec, st));
        if (const char *e = getenv("KVX_DBG_GRAPH_SYNC")) { if (atoi(e) & 2) HIPCHK(hipStreamSynchronize(st)); }
    } else {
        int rc = enqueue_factor_body(F);
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(F->ev[1], st));
    HIPCHK(hipGetLastError());
    F->pending = true;
    F->have_ftime = false;
    return KVX_OK;
}

int finish_factor(kvx_chol *F, int64_t *minor)
{
    if (F->pending) {
        HIPCHK(hipStreamSynchronize(F->stream));
        F->pending = false;
        prof_collect(F);
        float ms = 0;
        if (hipEventElapsedTime(&ms, F->ev[0], F->ev[1]) == hipSuccess) { F->ms_factor = ms; F->have_ftime = true; }
        int st = *F->h_status;
        F->numeric = true;
        F->minor = (st >= 0x7f7f7f7f) ? F->S.n : (int64_t)st;
    }
    if (minor) *minor = F->minor;
    if (!F->numeric) return KVX_ESYMBOLIC;
    return F->minor < F->S.n ? KVX_ENOTPOSDEF : KVX_OK;
}

// Fork the independent kernel groups of one level onto the side streams; join at level end.
struct LevelStreams {
    SweepStreams S;
    hipStream_t lds, wave;
    bool fork_lds, fork_wave;
    static SweepStreams own(kvx_chol *F)
    {
        static const bool nofork = [] { const char *e = getenv("KVX_SOLVE_NOFORK"); return e && atoi(e) != 0; }();
        if (nofork) return SweepStreams{F->stream, F->stream, F->stream, F->ev_fork, F->ev_join[0], F->ev_join[1]};
        return SweepStreams{F->stream, F->side[0], F->side[1], F->ev_fork, F->ev_join[0], F->ev_join[1]};
    }
    LevelStreams(kvx_chol *F, bool have_big, bool have_lds, bool have_wave, const SweepStreams *ss = nullptr) : S(ss ? *ss : own(F))
    {
        fork_lds = have_lds && (have_big || have_wave) && S.lds != S.main;
        fork_wave = have_wave && have_big && S.wave != S.main;
        lds = fork_lds ? S.lds : S.main;
        wave = fork_wave ? S.wave : S.main;
        if (fork_lds || fork_wave) {
            (void)hipEventRecord(S.fork, S.main);
            if (fork_lds) (void)hipStreamWaitEvent(S.lds, S.fork, 0);
            if (fork_wave) (void)hipStreamWaitEvent(S.wave, S.fork, 0);
        }
    }
    void join()
    {
        if (fork_lds) { (void)hipEventRecord(S.join0, S.lds); (void)hipStreamWaitEvent(S.main, S.join0, 0); }
        if (fork_wave) { (void)hipEventRecord(S.join1, S.wave); (void)hipStreamWaitEvent(S.main, S.join1, 0); }
    }
};

void enqueue_fwd(kvx_chol *F, double *X, int64_t ldx, int nrhs, int lfrom, int lto, const SweepStreams *ss, bool wait_levels, bool sub_tail)
{
    Symbolic &S = F->S;
    const int64_t wstride = std::max(S.wrk_size[0], S.wrk_size[1]);
    hipStream_t sm = ss ? ss->main : F->stream;
    if (lfrom < 0) lfrom = S.nlevels - 1;
    if (lfrom == S.nlevels - 1 && F->nsub > 0 && !wait_levels) {       // the leaf subtrees: one wavefront each, before any level
        ProfScope ps(F, FAM_FWD, sm);
        launch_fwd_subtree(sm, F->ds, F->d_subs, F->nsub, F->d_cd_woff, F->d_Lx, X, ldx, nrhs, F->d_W[0], F->d_W[1], wstride, F->d_depth);
    }
    if (wait_levels && sub_tail && F->nsub > 0) {
        // the part of the tree at level lto and below is factored: its subtrees (the groups are stored by ascending level) in one launch
        (void)hipStreamWaitEvent(sm, F->ev_lvl[(size_t)lto], 0);
        const int off = F->sub_lvl_off[(size_t)lto];
        launch_fwd_subtree(sm, F->ds, F->d_subs_lvl + off, F->nsub - off, F->d_cd_woff, F->d_Lx, X, ldx, nrhs, F->d_W[0], F->d_W[1], wstride, F->d_depth);
    }
    for (int l = lfrom; l >= lto; l--) {
        const LevelPlan &P = F->plan[l];
        const double *Wch = F->d_W[(l + 1) & 1];
        double *Wout = F->d_W[l & 1];
        const int64_t woff = F->sw_off[l];
        const int wcnt = F->sw_cnt[l];
        const int nlds = F->solve_merged ? 0 : P.scnt[1];
        const int nsubl = (wait_levels && !sub_tail && F->nsub > 0) ? F->sub_lvl_cnt[(size_t)l] : 0;
        if (wcnt == 0 && P.scnt[0] == 0 && nlds == 0 && nsubl == 0) continue;
        if (wait_levels) (void)hipStreamWaitEvent(sm, F->ev_lvl[(size_t)l], 0);         // level l (and every deeper one) is factored
        if (nsubl > 0)      // behind a factorisation in flight: the subtrees ROOTED at this level (their fronts are at this depth or deeper)
            launch_fwd_subtree(sm, F->ds, F->d_subs_lvl + F->sub_lvl_off[(size_t)l], nsubl, F->d_cd_woff, F->d_Lx, X, ldx, nrhs, F->d_W[0], F->d_W[1],
                               wstride, F->d_depth);
        if (wcnt == 0 && P.scnt[0] == 0 && nlds == 0) continue;
        // with subtrees: every small front of the level that is outside them goes into ONE launch of the LDS kernel;
        // without (sharded mode): the wave classes keep their own kernel and stream
        LevelStreams ls(F, P.scnt[0] > 0, F->solve_merged ? wcnt > 0 : nlds > 0, F->solve_merged ? false : wcnt > 0, ss);
        if (wcnt > 0) {
            if (F->solve_merged) {
                ProfScope ps(F, FAM_FWD, ls.lds);
                launch_fwd_lds(ls.lds, F->ds, F->d_lists_sw + woff, wcnt, F->sw_kmax[l], F->d_Lx, X, ldx, nrhs, Wch, Wout, wstride);
            } else {
                ProfScope ps(F, FAM_FWD, ls.wave);
                launch_fwd_wave(ls.wave, F->ds, F->d_lists_sw + woff, wcnt, 32, F->d_Lx, X, ldx, nrhs, Wch, Wout, wstride);
            }
        }
        if (nlds > 0) {
            ProfScope ps(F, FAM_FWD, ls.lds);
            launch_fwd_lds(ls.lds, F->ds, F->d_lists + P.soff[1], nlds, std::max(P.maxk[KVX_CLS_LDS128], P.maxk[KVX_CLS_LDS96]),
                           F->d_Lx, X, ldx, nrhs, Wch, Wout, wstride);
        }
        if (P.scnt[0] > 0) {
            ProfScope ps(F, FAM_FWD, sm);
            launch_fwd_big(sm, F->ds, F->d_lists + P.soff[0], P.scnt[0], P.smaxm[0], P.big_maxk, F->d_Lx, F->d_Linv,
                           X, F->d_X0, ldx, nrhs, F->d_WK, S.n, Wch, Wout, wstride, P.scnt[0]);
        }
        ls.join();
    }
}

void enqueue_bwd(kvx_chol *F, double *X, int64_t ldx, int nrhs, int lfrom, int lto)
{
    Symbolic &S = F->S;
    if (lto < 0) lto = S.nlevels - 1;
    for (int l = lfrom; l <= lto; l++) {
        const LevelPlan &P = F->plan[l];
        const int64_t woff = F->sw_off[l];
        const int wcnt = F->sw_cnt[l];
        const int nlds = F->solve_merged ? 0 : P.scnt[1];
        if (wcnt == 0 && P.scnt[0] == 0 && nlds == 0) continue;
        LevelStreams ls(F, P.scnt[0] > 0, F->solve_merged ? wcnt > 0 : nlds > 0, F->solve_merged ? false : wcnt > 0);
        if (wcnt > 0) {
            if (F->solve_merged) {
                ProfScope ps(F, FAM_BWD, ls.lds);
                launch_bwd_lds(ls.lds, F->ds, F->d_lists_sw + woff, wcnt, F->d_Lx, X, ldx, nrhs);
            } else {
                ProfScope ps(F, FAM_BWD, ls.wave);
                launch_bwd_wave(ls.wave, F->ds, F->d_lists_sw + woff, wcnt, 64, 32, F->d_Lx, X, ldx, nrhs);
            }
        }
        if (nlds > 0) {
            ProfScope ps(F, FAM_BWD, ls.lds);
            launch_bwd_lds(ls.lds, F->ds, F->d_lists + P.soff[1], nlds, F->d_Lx, X, ldx, nrhs);
        }
        if (P.scnt[0] > 0) {
            ProfScope ps(F, FAM_BWD);
            launch_bwd_big(F->stream, F->ds, F->d_lists + P.soff[0], P.scnt[0], P.smaxm[0], P.big_maxk, F->d_Lx, F->d_Linv,
                           X, ldx, nrhs, F->d_WK, S.n);
        }
        ls.join();
    }
    if (lto == S.nlevels - 1 && F->nsub > 0) {        // the leaf subtrees last: every ancestor is solved
        if (nrhs == 1 && F->prof_family < 0) {
            // one right-hand side: the three size groups (registers for 32 / 48 / 64 rows of a column) side by side on three streams --
            // the walks of the largest group alone fit the GPU in one round instead of two to three for all of them in its kernel
            const int cnt[3] = {F->nsub32, F->nsub48 - F->nsub32, F->nsub - F->nsub48};
            const int off[3] = {0, F->nsub32, F->nsub48};
            const int cap[3] = {32, 48, 64};
            hipStream_t ss[3] = {F->stream, F->side[0], F->side[1]};
            const bool fork = (cnt[0] > 0) + (cnt[1] > 0) + (cnt[2] > 0) > 1;
            if (fork) {
                (void)hipEventRecord(F->ev_fork, F->stream);
                for (int g = 1; g < 3; g++)
                    if (cnt[g] > 0) (void)hipStreamWaitEvent(ss[g], F->ev_fork, 0);
            }
            for (int g = 2; g >= 0; g--)               // (the longest walks first)
                launch_bwd_subtree_group(fork ? ss[g] : F->stream, cap[g], F->ds, F->d_subs + off[g], cnt[g], F->d_Lx, X, ldx);
            if (fork)
                for (int g = 1; g < 3; g++)
                    if (cnt[g] > 0) { (void)hipEventRecord(F->ev_join[g - 1], ss[g]); (void)hipStreamWaitEvent(F->stream, F->ev_join[g - 1], 0); }
        } else {
            ProfScope ps(F, FAM_BWD);
            launch_bwd_subtree(F->stream, F->ds, F->d_subs, F->nsub, F->nsub32, F->d_Lx, X, ldx, nrhs);
        }
    }
}

// ---- many right-hand sides: rhs-major blocks of 64 (kernels_wide.hip) ---------------------------------------------------------
// The inverse of the relative indices: for every row of every front, the rows of its children's update vectors that are added to
// it, children in list order (the order the single-rhs kernels add them in).  Host pass over the tree, once per analysis.
int ensure_wide(kvx_chol *F)
{
    if (F->wide_state != 0) return KVX_OK;
    Symbolic &S = F->S;
    F->wide_state = -1;
    if (F->dist_nranks != 1 || S.sum_m >= INT32_MAX - 1 || (int64_t)S.rel.size() >= INT32_MAX) return KVX_OK;
    if (std::max(S.wrk_size[0], S.wrk_size[1]) >= INT32_MAX) return KVX_OK;
    const int64_t nrow = S.rowptr[S.nsuper];
    std::vector<int32_t> ptr((size_t)nrow + 1, 0);
    for (int64_t s = 0; s < S.nsuper; s++)
        for (int64_t ci = S.childptr[s]; ci < S.childptr[s + 1]; ci++) {
            const int32_t c = S.children[ci];
            const int64_t kc = S.sn_k[c], uc = S.sn_m[c] - kc;
            const int32_t *rel = S.rel.data() + S.rowptr[c] + kc;
            for (int64_t i = 0; i < uc; i++) ptr[(size_t)(S.rowptr[s] + rel[i]) + 1]++;
        }
    for (int64_t r = 0; r < nrow; r++) ptr[(size_t)r + 1] += ptr[(size_t)r];
    std::vector<int32_t> src((size_t)std::max<int64_t>(ptr[(size_t)nrow], 1), 0);
    {
        std::vector<int32_t> cur(ptr.begin(), ptr.end() - 1);
        for (int64_t s = 0; s < S.nsuper; s++)
            for (int64_t ci = S.childptr[s]; ci < S.childptr[s + 1]; ci++) {
                const int32_t c = S.children[ci];
                const int64_t kc = S.sn_k[c], uc = S.sn_m[c] - kc;
                const int32_t *rel = S.rel.data() + S.rowptr[c] + kc;
                for (int64_t i = 0; i < uc; i++) src[(size_t)cur[(size_t)(S.rowptr[s] + rel[i])]++] = (int32_t)(S.wx[c] + i);
            }
    }
    int rc;
    std::vector<int32_t> ip32((size_t)S.n);
    for (int64_t j = 0; j < S.n; j++) ip32[(size_t)j] = (int32_t)S.iperm[(size_t)j];
    if ((rc = upload(&F->d_iperm, ip32))) return rc;
    if ((rc = upload(&F->d_inv_ptr, ptr))) return rc;
    if ((rc = upload(&F->d_inv_src, src))) return rc;
    F->wide_state = 1;
    return KVX_OK;
}

// the small fronts of a level: the LDS classes (k <= 64) and the wave classes (k <= 32), each with the largest pivot count it holds
static void small_lists(const LevelPlan &P, int64_t off[2], int cnt[2], int kmax[2])
{
    off[0] = P.soff[1]; cnt[0] = P.scnt[1]; kmax[0] = std::max(P.maxk[KVX_CLS_LDS128], P.maxk[KVX_CLS_LDS96]);
    off[1] = P.soff[2]; cnt[1] = P.scnt[2]; kmax[1] = 0;
    for (int c = KVX_CLS_WAVE0; c < KVX_NCLS; c++) kmax[1] = std::max(kmax[1], P.maxk[c]);
}

void enqueue_fwd_wide(kvx_chol *F, double *XT, int nchunk)
{
    Symbolic &S = F->S;
    const int64_t wstride = std::max(S.wrk_size[0], S.wrk_size[1]);
    for (int l = S.nlevels - 1; l >= 0; l--) {
        const LevelPlan &P = F->plan[l];
        const double *Wch = F->d_W[(l + 1) & 1];
        double *Wout = F->d_W[l & 1];
        int64_t off[2]; int cnt[2], kmax[2];
        small_lists(P, off, cnt, kmax);
        if (cnt[0] == 0 && cnt[1] == 0 && P.scnt[0] == 0) continue;
        LevelStreams ls(F, P.scnt[0] > 0, cnt[0] > 0, cnt[1] > 0);
        for (int g = 0; g < 2; g++)
            if (cnt[g] > 0) {
                hipStream_t sg = g == 0 ? ls.lds : ls.wave;
                ProfScope ps(F, FAM_FWD, sg);
                launch_wide_fwd_small(sg, F->ds, F->d_lists + off[g], cnt[g], kmax[g], nchunk, F->d_Lx, XT, S.n, Wch, Wout, wstride,
                                      F->d_inv_ptr, F->d_inv_src);
            }
        if (P.scnt[0] > 0) {
            ProfScope ps(F, FAM_FWD);
            launch_wide_fwd_big(F->stream, F->ds, F->d_lists + P.soff[0], P.scnt[0], P.smaxm[0], P.big_maxk, nchunk, F->d_Lx, F->d_Linv,
                                XT, S.n, Wch, Wout, wstride, F->d_inv_ptr, F->d_inv_src);
        }
        ls.join();
    }
}

void enqueue_bwd_wide(kvx_chol *F, double *XT, int nchunk)
{
    Symbolic &S = F->S;
    for (int l = 0; l < S.nlevels; l++) {
        const LevelPlan &P = F->plan[l];
        int64_t off[2]; int cnt[2], kmax[2];
        small_lists(P, off, cnt, kmax);
        if (cnt[0] == 0 && cnt[1] == 0 && P.scnt[0] == 0) continue;
        LevelStreams ls(F, P.scnt[0] > 0, cnt[0] > 0, cnt[1] > 0);
        for (int g = 0; g < 2; g++)
            if (cnt[g] > 0) {
                hipStream_t sg = g == 0 ? ls.lds : ls.wave;
                ProfScope ps(F, FAM_BWD, sg);
                launch_wide_bwd_small(sg, F->ds, F->d_lists + off[g], cnt[g], kmax[g], nchunk, F->d_Lx, XT, S.n);
            }
        if (P.scnt[0] > 0) {
            ProfScope ps(F, FAM_BWD);
            launch_wide_bwd_big(F->stream, F->ds, F->d_lists + P.soff[0], P.scnt[0], P.big_maxk, nchunk, F->d_Lx, F->d_Linv, XT, S.n);
        }
        ls.join();
    }
}

// B_dev: n x nrhs, leading dimension ldB, device memory.
int solve_dev(kvx_chol *F, int sys, double *B, int64_t nrhs, int64_t ldB, bool async = false)
{
    Symbolic &S = F->S;
    const int64_t n = S.n;
    if (sys < 0 || sys > 8) { set_err("invalid value for sys"); return KVX_EINVAL; }
    // A factorisation still in flight on the factor's stream (kvx_chol_factorize_async_dev): the solve is queued
    // behind it at once -- no host round trip between the two -- and its status is examined when both are done
    // (on failure B holds garbage and the call reports the singular factor, as it would have before starting).
    const bool deferred = F->pending;
    int rc = KVX_OK;
    if (!deferred) {
        rc = finish_factor(F, nullptr);
        if (rc == KVX_ESYMBOLIC) { set_err("called with symbolic factor"); return rc; }
        if (rc == KVX_ENOTPOSDEF) { set_err("singular matrix"); return KVX_ESINGULAR; }
        if (rc) return rc;
    }
    if (n == 0 || nrhs == 0) return deferred ? ((rc = finish_factor(F, nullptr)) == KVX_ENOTPOSDEF ? KVX_ESINGULAR : rc) : KVX_OK;
    if (ldB < std::max<int64_t>(1, n)) { set_err("ldB must be >= max(1,n)"); return KVX_EINVAL; }
    if (sys == 6 && F->is_ll) return KVX_OK;   // D = I for an LL' factor
    hipStream_t st = F->stream;
    const bool ldl = !F->is_ll && sys >= 2 && sys <= 6;
    if (ldl && !F->diag_valid) {
        if (!F->d_diag) HIPCHK(pool_malloc((void **)&F->d_diag, (size_t)n * sizeof(double)));
        launch_extract_diag(st, F->ds, S.nsuper, F->d_Lx, F->d_diag);
        F->diag_valid = true;
    }
    // wstride: both parity buffers are allocated with wrk_size[p]*x_cap; use a common stride
    // many right-hand sides of a plain LL' system: rhs-major blocks of 64 (kernels_wide.hip)
    const int kind0 = (sys == 0 || sys == 1) ? 0 : ((sys == 2 || sys == 4) ? 1 : ((sys == 3 || sys == 5) ? 2 : -1));
    // A block of 64 costs the same whatever it holds, the older kernels grow with every right-hand side: measured break-even
    // (scratch/wide_thresh.py, 2-D grids) at 48 right-hand sides for n = 5e4, ~22 for n = 2.5e5, ~9 for n = 1e6.
    const bool wide_by_size = nrhs >= 48 || (nrhs >= 8 && (double)nrhs * (double)n >= 6e6);
    bool wide = kind0 >= 0 && F->prof_family < 0 && (F->wide_from < 0 ? wide_by_size : (F->wide_from > 0 && nrhs >= F->wide_from));
    if (wide) {
        if ((rc = ensure_wide(F))) return rc;
        wide = F->wide_state == 1;
    }
    // rhs-major passes: at most 1024 right-hand sides, fewer on very large systems (the workspace is a few blocks of n x pass doubles)
    const int wide_pass = (int)std::max<int64_t>(64, std::min<int64_t>(1024, ((int64_t)(1e9 / (double)std::max<int64_t>(n, 1)) / 64) * 64));
    const int chunk_max = wide ? wide_pass : 65535;
    for (int64_t r0 = 0; r0 < nrhs; r0 += chunk_max) {
        int nr = (int)std::min<int64_t>(chunk_max, nrhs - r0);
        double *Bc = B + r0 * ldB;
        if ((rc = ensure_solve_ws(F, wide ? (int64_t)((nr + 63) / 64) * 64 : nr))) return rc;
        HIPCHK(hipEventRecord(F->ev[2], st));
        if (wide) {
            const int nchunk = (nr + 63) / 64;
            // LDL' view: D L' x = b  ->  Lc' x = diag^-1 b;  L' x = b  ->  Lc' x = diag b (on the way in);
            //            L D x = b   ->  x = diag^-1 Lc^-1 b;  L x = b   ->  x = diag Lc^-1 b (on the way out)
            const bool sc_in = ldl && (sys == 3 || sys == 5), sc_out = ldl && (sys == 2 || sys == 4);
            launch_wide_gather(st, sys == 0 ? F->d_iperm : nullptr, n, nr, Bc, ldB, F->d_X, sc_in ? F->d_diag : nullptr, sys == 3 ? 1 : 0);
            auto body = [&]() -> int {
                if (kind0 == 0 || kind0 == 1) enqueue_fwd_wide(F, F->d_X, nchunk);
                if (kind0 == 0 || kind0 == 2) enqueue_bwd_wide(F, F->d_X, nchunk);
                return hipGetLastError() == hipSuccess ? KVX_OK : KVX_EDEVICE;
            };
            hipGraphExec_t exec = nullptr;
            if (F->use_graph && !getenv("KVX_DBG_NO_SOLVE_GRAPH")) {
                kvx_chol::SolveGraph *slot = nullptr;
                for (auto &g : F->g_solve)
                    if (g.kind == kind0 + 8 && g.nrhs == nchunk) slot = &g;
                if (!slot) { F->g_solve.push_back({kind0 + 8, nchunk, 0, LazyExec{}}); slot = &F->g_solve.back(); }
                slot->calls++;
                if (!slot->exec.tried && slot->calls >= 2 && F->g_solve.size() <= 16) capture_graph(F, body, slot->exec);
                exec = slot->exec.ready();
            }
            if (exec) HIPCHK(hipGraphLaunch(exec, st));
            else if ((rc = body())) return rc;
            launch_wide_scatter(st, sys == 0 ? F->d_iperm : nullptr, n, nr, F->d_X, Bc, ldB, sc_out ? F->d_diag : nullptr, sys == 2 ? 1 : 0);
            HIPCHK(hipEventRecord(F->ev[3], st));
            HIPCHK(hipGetLastError());
            continue;
        }
        // every system is solved on the staging block d_X (n x nr, ld = n): fixed pointers, so the
        // triangular sweeps can be replayed from a captured graph
        const int kind = (sys == 0 || sys == 1) ? 0 : ((sys == 2 || sys == 4) ? 1 : ((sys == 3 || sys == 5) ? 2 : -1));
        if (sys == 0 || sys == 7) launch_perm_gather(st, F->d_perm, n, nr, Bc, ldB, F->d_X, n);
        else if (sys == 8) launch_perm_scatter(st, F->d_perm, n, nr, Bc, ldB, F->d_X, n);
        else HIPCHK(hipMemcpy2DAsync(F->d_X, n * sizeof(double), Bc, ldB * sizeof(double), n * sizeof(double), nr, hipMemcpyDeviceToDevice, st));
        // LDL' view: D L' x = b  ->  Lc' x = diag^-1 b;  L' x = b  ->  Lc' x = diag b;  D x = b  ->  x = diag^-2 b
        if (ldl && (sys == 3 || sys == 5 || sys == 6)) launch_diag_scale(st, n, nr, F->d_diag, F->d_X, n, sys == 3 ? 1 : (sys == 5 ? 0 : 2));
        if (kind >= 0) {
            auto body = [&]() -> int {
                if (kind == 0 || kind == 1) {
                    // the first forward step of a big front is spread over workgroups that all read the front's
                    // pivot entries of the rhs while one of them overwrites them with y: they read this copy
                    launch_copy_d(F->stream, F->d_X0, F->d_X, n * (int64_t)nr);         // (a kernel, not a memcpy node: see enqueue_factor_body)
                    enqueue_fwd(F, F->d_X, n, nr);
                }
                if (kind == 0 || kind == 2) enqueue_bwd(F, F->d_X, n, nr);
                return hipGetLastError() == hipSuccess ? KVX_OK : KVX_EDEVICE;
            };
            hipGraphExec_t exec = nullptr;
            if (F->use_graph && F->prof_family < 0 && !getenv("KVX_DBG_NO_SOLVE_GRAPH")) {
                kvx_chol::SolveGraph *slot = nullptr;
                for (auto &g : F->g_solve)
                    if (g.kind == kind && g.nrhs == nr) slot = &g;
                if (!slot) { F->g_solve.push_back({kind, nr, 0, LazyExec{}}); slot = &F->g_solve.back(); }
                slot->calls++;
                if (!slot->exec.tried && slot->calls >= 2 && F->g_solve.size() <= 16) capture_graph(F, body, slot->exec);
                exec = slot->exec.ready();
            }
            if (exec) HIPCHK(hipGraphLaunch(exec, st));
            else if ((rc = body())) return rc;
        }
        // L D x = b  ->  x = diag^-1 Lc^-1 b;  L x = b  ->  x = diag Lc^-1 b
        if (ldl && (sys == 2 || sys == 4)) launch_diag_scale(st, n, nr, F->d_diag, F->d_X, n, sys == 2 ? 1 : 0);
        if (sys == 0) launch_perm_scatter(st, F->d_perm, n, nr, F->d_X, n, Bc, ldB);
        else HIPCHK(hipMemcpy2DAsync(Bc, ldB * sizeof(double), F->d_X, n * sizeof(double), n * sizeof(double), nr, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipEventRecord(F->ev[3], st));
        HIPCHK(hipGetLastError());
    }
    if (async) {
        // no host synchronisation: the caller's (null-stream) work is ordered behind the solve by an event; a factorisation
        // that was still in flight stays pending and its status is examined at the next synchronising call
        HIPCHK(hipEventRecord(F->ev_out, st));
        HIPCHK(hipStreamWaitEvent(nullptr, F->ev_out, 0));
        return KVX_OK;
    }
    HIPCHK(hipStreamSynchronize(st));
    if (deferred) {
        rc = finish_factor(F, nullptr);
        if (rc == KVX_ENOTPOSDEF) { set_err("singular matrix"); return KVX_ESINGULAR; }
        if (rc) return rc;
    }
    prof_collect(F);
    float ms = 0;
    if (hipEventElapsedTime(&ms, F->ev[2], F->ev[3]) == hipSuccess) { F->ms_solve = ms; F->have_stime = true; }
    return KVX_OK;
}

// Numeric factorisation AND the solve of A X = B (sys 0) as ONE enqueue: the right-hand sides are known before the factorisation
// starts, so the forward sweep does not have to wait for all of it -- level l of the sweep needs the fronts of level l and below
// only.  The sweep runs on the stream of the factorisation's small-front launches (side[0], idle at the top of the tree) behind one event per level of the factorisation: by the
// time the root front is factored the sweep has reached the top of the tree, and what is left of it is the root's own step
// (config 2: 0.65 ms of forward sweep hidden under the pivot chain of the top levels).  Same kernels on the same data in the
// same order per front as kvx_chol_factorize_dev + kvx_chol_solve_dev: bitwise the same factor and solution.  The whole
// sequence replays from a captured graph from the second call with the same (nrhs, B, ldB) on.
int factor_solve_dev(kvx_chol *F, const double *values_dev, double *B, int64_t nrhs, int64_t ldB, bool async = false)
{
    int rc = ensure_device(F);
    if (rc) return rc;
    Symbolic &S = F->S;
    const int64_t n = S.n;
    if (nrhs < 0) { set_err("nrhs out of range"); return KVX_EINVAL; }
    if (n > 0 && nrhs > 0 && ldB < n) { set_err("ldB must be >= max(1,n)"); return KVX_EINVAL; }
    // outside the pipelined form: sharded factors, LDL' views, the rhs-major path of many right-hand sides, family timing
    // ... and HIP runtimes before 7.2: under 7.0.51831 (the one inside the PyTorch wheel, which a process gets when it imports torch
    // before this library) hipGraphLaunch of the captured five-stream sequence crashes inside the runtime (the three-stream graphs
    // of the separate calls replay correctly there); without a graph the pipelined form is slower than the two replayed graphs
    static const bool old_runtime = [] {
        int v = 0;
        if (hipRuntimeGetVersion(&v) != hipSuccess) { (void)hipGetLastError(); return true; }
        const char *e = getenv("KVX_DBG_FUSED_ANY_RUNTIME");          // debugging only: reproduces the crash of DESIGN.md section 5
        return v < 70200000 && !(e && e[0] == '1');
    }();
    // (KVX_FACTOR_SUBTREES=1 keeps the pipelined form since round 4 -- the subtree launches precede the level loop on the factor's
    //  stream, so a level's completion event covers them: config 4b's first direction 1.41 -> 1.29 ms, still behind the level
    //  schedule's 1.20-1.26; KVX_SUBTREES_PLAIN=1: two enqueues as before)
    const bool plain = F->dist_nranks != 1 || !F->is_ll || nrhs == 0 || nrhs > 16 || n == 0 || F->prof_family >= 0 || (F->factor_subtrees && getenv("KVX_SUBTREES_PLAIN")) ||
                       old_runtime || !F->use_graph;
    if ((rc = wait_for_caller(F))) return rc;
    if (S.nnzA > 0) HIPCHK(hipMemcpyAsync(F->d_Ax, values_dev, S.nnzA * sizeof(double), hipMemcpyDeviceToDevice, F->stream));
    F->last_fused_path = plain ? 2 : 1;
    if (plain) {
        if ((rc = enqueue_factor(F))) return rc;
        rc = solve_dev(F, 0, B, nrhs, ldB, async);
        // the factorisation was enqueued by THIS call: its failure is the call's result (KVX_ENOTPOSDEF and the failing column, as
        // the one-enqueue form and kvx_chol_factorize report it), not solve_dev's "singular matrix" for a factor found unusable.
        // (Enqueue-only form: the status stays deferred to kvx_chol_status, which reports KVX_ENOTPOSDEF too.)
        if (rc == KVX_ESINGULAR && !async && F->numeric && F->minor < S.n) { set_err("matrix is not positive definite"); return KVX_ENOTPOSDEF; }
        return rc;
    }
    const int nr = (int)nrhs;
    static const bool dbg_t = getenv("KVX_DBG_T") != nullptr;
    const auto t_in = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (dbg_t) fprintf(stderr, "factor_solve_dev %s: %.0f us\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_in).count());
    };
    if ((rc = ensure_solve_ws(F, nr))) return rc;
    lap("solve workspace");
    if (F->ev_lvl.empty()) {
        F->ev_lvl.assign((size_t)S.nlevels, nullptr);
        for (auto &e : F->ev_lvl) HIPCHK(pool_event_get(&e, false));
        for (int i = 0; i < 4; i++) HIPCHK(pool_event_get(&F->ev_pipe[i], false));
        if (F->nsub > 0) {
            std::vector<std::vector<SubDesc>> by((size_t)S.nlevels);
            for (const SubDesc &d : F->subs_host) by[(size_t)S.depth[(size_t)d.hi]].push_back(d);      // (fronts of a subtree are numbered in postorder: hi is the root)
            std::vector<SubDesc> flat;
            F->sub_lvl_off.assign((size_t)S.nlevels, 0);
            F->sub_lvl_cnt.assign((size_t)S.nlevels, 0);
            for (int l = 0; l < S.nlevels; l++) {
                F->sub_lvl_off[(size_t)l] = (int)flat.size();
                F->sub_lvl_cnt[(size_t)l] = (int)by[(size_t)l].size();
                flat.insert(flat.end(), by[(size_t)l].begin(), by[(size_t)l].end());
            }
            if (F->d_subs_lvl) { (void)pool_free(F->d_subs_lvl); F->d_subs_lvl = nullptr; }
            if ((rc = upload(&F->d_subs_lvl, flat))) return rc;
        }
        // the level the sweep starts at: the deepest one from which up no level holds more than KVX_PIPE_FRONTS fronts.  Default 4 --
        // on the 2-D systems the sweep then starts when the children of the root are factored and runs beside the root's own pivot
        // chain, a handful of workgroups per launch (config 2: step 5.15 -> 4.66 ms; started two levels earlier, beside launches of
        // thousands of tiles, 4.83; five levels earlier 5.2: the sweep's workgroups then delay the factorisation by what they gain)
        int lim = 4;
        if (const char *e = getenv("KVX_PIPE_FRONTS")) lim = atoi(e);
        F->pipe_from = 0;
        for (int l = 0; l < S.nlevels; l++) {
            if (S.levelptr[(size_t)l + 1] - S.levelptr[(size_t)l] > lim) break;
            F->pipe_from = l;
        }
    }
    lap("pipeline set-up");
    static const bool own_stream = [] { const char *e = getenv("KVX_PIPE_OWN_STREAM"); return e && e[0] == '1'; }();
    hipStream_t st = F->stream, s2 = own_stream ? F->side[2] : F->side[0];
    auto body = [&]() -> int {
        // the sweep's stream joins behind the values (and, in a capture, the capture): right-hand sides into the work vector first
        HIPCHK(hipEventRecord(F->ev_pipe[0], st));
        HIPCHK(hipStreamWaitEvent(s2, F->ev_pipe[0], 0));
        launch_perm_gather(s2, F->d_perm, n, nr, B, ldB, F->d_X, n);
        launch_copy_d(s2, F->d_X0, F->d_X, n * (int64_t)nr);
        F->pipe_on = true;
        F->pipe_nr = nr;
        const int rb = enqueue_factor_body(F);                     // (with the forward sweep of every level right behind that level)
        F->pipe_on = false;
        if (rb) return rb;
        HIPCHK(hipEventRecord(F->ev_pipe[0], s2));
        HIPCHK(hipStreamWaitEvent(st, F->ev_pipe[0], 0));
        enqueue_bwd(F, F->d_X, n, nr);
        launch_perm_scatter(st, F->d_perm, n, nr, F->d_X, n, B, ldB);
        return hipGetLastError() == hipSuccess ? KVX_OK : KVX_EDEVICE;
    };
    HIPCHK(hipEventRecord(F->ev[0], st));
    F->factor_calls++;
    F->diag_valid = false;
    hipGraphExec_t exec = nullptr;
    if (F->use_graph && !getenv("KVX_DBG_NO_FACTOR_GRAPH") && !getenv("KVX_FUSED_EAGER")) {
        kvx_chol::FusedGraph *slot = nullptr;
        for (auto &g : F->g_fused)
            if (g.nrhs == nr && g.B == B && g.ldB == ldB) slot = &g;
        if (!slot) {
            if (F->g_fused.size() >= 4) {                           // (right-hand sides at changing addresses: no pile of graphs)
                for (auto &g : F->g_fused) g.exec.drop();
                F->g_fused.clear();
            }
            F->g_fused.push_back({nr, B, ldB, 0, LazyExec{}});
            slot = &F->g_fused.back();
        }
        slot->calls++;
        if (!slot->exec.tried && slot->calls >= 2) capture_graph(F, body, slot->exec);
        lap("capture");
        exec = slot->exec.ready();
    }
    if (exec) HIPCHK(hipGraphLaunch(exec, st));
    else if ((rc = body())) return rc;
    lap(exec ? "graph launch" : "eager enqueue");
    HIPCHK(hipEventRecord(F->ev[1], st));                           // (the two parts are not separable here: last_timing reports the whole
    HIPCHK(hipGetLastError());                                      //  call as the factorisation and 0 for the solve)
    F->pending = true;
    F->have_ftime = false;
    if (async) {
        // no host synchronisation: the caller's (null-stream) work is ordered behind the call by an event; the status of the
        // factorisation is examined at the next synchronising call (kvx_chol_status)
        HIPCHK(hipEventRecord(F->ev_out, st));
        HIPCHK(hipStreamWaitEvent(nullptr, F->ev_out, 0));
        return KVX_OK;
    }
    HIPCHK(hipStreamSynchronize(st));
    rc = finish_factor(F, nullptr);
    F->ms_solve = 0.0; F->have_stime = true;
    if (rc == KVX_ENOTPOSDEF) { set_err("singular matrix"); return KVX_ENOTPOSDEF; }
    return rc;
}

}  // namespace kvx

extern "C" {

const char *kvx_version(void) { return "kvxhip 0.1 (gfx950)"; }
const char *kvx_last_error(void) { return g_err.c_str(); }
int kvx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int kvx_current_device(void)
{
    int n = 0, d = -1;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return -1; }
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return d;
}

int kvx_set_device(int dev)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); set_err("no HIP device visible"); return KVX_EDEVICE; }
    if (dev < 0 || dev >= n) { set_err("kvx_set_device: no such device"); return KVX_EINVAL; }
    HIPCHK(hipSetDevice(dev));
    return KVX_OK;
}

void kvx_chol_default_opts(kvx_chol_opts *o)
{
    memset(o, 0, sizeof(*o));
    o->supernodal = 2;
    o->ordering = 0;
    o->postorder = 1;
    o->relax_small = 4;
    o->relax_z1 = 0.8;
    o->relax_z2 = 0.1;
    o->relax_z3 = -1.0;     // < 0: by the order of the matrix -- 0.075 (CHOLMOD's own default is 0.05: measured on MI355X, section 3 of
                            // DESIGN.md), 0.2 up to 150 000 columns (see kvx_chol_analyze)
    // (experiments: KVX_RELAX_Z1 / _Z2 / _Z3 override the defaults of every analysis in the process)
    if (const char *e = getenv("KVX_RELAX_Z1")) o->relax_z1 = atof(e);
    if (const char *e = getenv("KVX_RELAX_Z2")) o->relax_z2 = atof(e);
    if (const char *e = getenv("KVX_RELAX_Z3")) o->relax_z3 = atof(e);
    o->dbound = 0.0;
}

int kvx_chol_analyze(int64_t n, const int64_t *colptr, const int64_t *rowind, int uplo, const int64_t *perm,
                     const kvx_chol_opts *opts, kvx_chol **out)
{
    if (!out || n < 0 || (n > 0 && (!colptr || (!rowind && colptr[n] > 0)))) { set_err("bad arguments"); return KVX_EINVAL; }
    kvx_chol_opts o;
    if (opts) o = *opts; else kvx_chol_default_opts(&o);
    if (o.supernodal < 0 || o.supernodal > 2) { set_err("options['supernodal'] must be 0, 1 or 2"); return KVX_EINVAL; }
    if (o.ordering < 0 || o.ordering > 3) { set_err("ordering must be 0 (best of the library's own), 1 (natural), 2 (nested dissection) or 3 (minimum degree)"); return KVX_EINVAL; }
    kvx_chol *F = nullptr;
    try {
        F = new kvx_chol();
        F->opts = o;
        SymOpts so;
        so.ordering = o.ordering;
        so.postorder = o.postorder;
        so.relax_small = o.relax_small;
        so.relax_z1 = o.relax_z1; so.relax_z2 = o.relax_z2; so.relax_z3 = o.relax_z3;
        // Zero fraction a wide chain supernode may take on when it joins its parent.  A small factorisation is a chain of levels on an
        // idle machine: explicit zeros cost nothing there and every level saved is 50-100 us (5-pt Laplacian 200 x 200: 12 -> 9
        // levels, step 1.25 -> 1.08 ms; the KKT system of the interior-point leg, n = 50 000: 520 -> 565 iterations/s); from a few
        // 10^5 columns on the merged fronts lengthen the pivot chains at the top of the tree by more than the levels save
        // (1000 x 1000: 4.62 -> 4.81 ms with 0.2; 64^3: 17.2 -> 18.7 ms).  Crossover measured between 1.2e5 and 2.5e5 columns.
        if (so.relax_z3 < 0) so.relax_z3 = n <= 150000 ? 0.2 : 0.075;
        if (o.reserved[0] > 0) so.nd_leaf = o.reserved[0];
        if (o.reserved[1] != 0) so.leaf_cols = o.reserved[1] < 0 ? 0 : o.reserved[1];
        if (o.reserved[2] > 0) so.leaf_rows = o.reserved[2];
        so.compare_given = o.reserved[4] == 1 ? 1 : 0;
        if (o.reserved[5] > 0) so.amd_auto_max = o.reserved[5];
        if (o.reserved[6] != 0) so.nd_min_n = std::max<int64_t>(0, o.reserved[6]);      // (-1: always compute the dissection too)
        static const int64_t zero = 0;
        analyze(n, n ? colptr : &zero, rowind, uplo, perm, so, F->S);
        // options['supernodal'] (spsolvers.rst:731-736): 2 -> LL'; 0 -> LDL'; 1 -> whichever CHOLMOD would find cheaper,
        // by its own rule flops / nnz(L) >= 40 -> supernodal LL'.  The arithmetic is the supernodal LL' kernels either
        // way; an LDL' factor is the same numbers seen as L = Lc diag(Lc)^-1, D = diag(Lc)^2 (solve sys = 2..6,
        // getfactor and diag follow that form).
        F->is_ll = o.supernodal == 2 || (o.supernodal == 1 && F->S.lnz > 0 && F->S.flops / (double)F->S.lnz >= 40.0);
        F->minor = n;
    } catch (const std::invalid_argument &e) {
        delete F; set_err(e.what()); return KVX_EPERM;
    } catch (const std::bad_alloc &) {
        delete F; set_err("out of host memory"); return KVX_ENOMEM;
    } catch (const std::exception &e) {
        delete F; set_err(e.what()); return KVX_EINVAL;
    }
    *out = F;
    return KVX_OK;
}

static int kvx_chol_factorize_async_dev_impl(kvx_chol *F, const double *values_dev)
{
    if (!F) return KVX_EINVAL;
    int rc = ensure_device(F);
    if (rc) return rc;
    if ((rc = wait_for_caller(F))) return rc;
    if (F->S.nnzA > 0)
        HIPCHK(hipMemcpyAsync(F->d_Ax, values_dev, F->S.nnzA * sizeof(double), hipMemcpyDeviceToDevice, F->stream));
    return enqueue_factor(F);
}

int kvx_chol_factorize_async_dev(kvx_chol *F, const double *values_dev)
{
    return guarded([&] { return kvx_chol_factorize_async_dev_impl(F, values_dev); });
}

int kvx_chol_status(kvx_chol *F, int64_t *minor)
{
    if (!F) return KVX_EINVAL;
    return finish_factor(F, minor);
}

int kvx_chol_factorize_dev(kvx_chol *F, const double *values_dev, int64_t *minor)
{
    int rc = kvx_chol_factorize_async_dev(F, values_dev);
    if (rc) return rc;
    return finish_factor(F, minor);
}

static int kvx_chol_factorize_impl(kvx_chol *F, const double *values, int64_t *minor)
{
    if (!F) return KVX_EINVAL;
    int rc = ensure_device(F);
    if (rc) return rc;
    if (F->S.nnzA > 0)
        HIPCHK(hipMemcpyAsync(F->d_Ax, values, F->S.nnzA * sizeof(double), hipMemcpyHostToDevice, F->stream));
    rc = enqueue_factor(F);
    if (rc) return rc;
    return finish_factor(F, minor);
}

int kvx_chol_factorize(kvx_chol *F, const double *values, int64_t *minor)
{
    return guarded([&] { return kvx_chol_factorize_impl(F, values, minor); });
}

static int kvx_chol_solve_dev_impl(kvx_chol *F, int sys, double *B_dev, int64_t nrhs, int64_t ldB)
{
    if (!F) return KVX_EINVAL;
    if (!F->dev_ready) { set_err("called with symbolic factor"); return KVX_ESYMBOLIC; }
    int rc = wait_for_caller(F);
    if (rc) return rc;
    return solve_dev(F, sys, B_dev, nrhs, ldB);
}

int kvx_chol_solve_dev(kvx_chol *F, int sys, double *B_dev, int64_t nrhs, int64_t ldB)
{
    return guarded([&] { return kvx_chol_solve_dev_impl(F, sys, B_dev, nrhs, ldB); });
}

static int kvx_chol_solve_async_dev_impl(kvx_chol *F, int sys, double *B_dev, int64_t nrhs, int64_t ldB)
{
    if (!F) return KVX_EINVAL;
    if (!F->dev_ready) { set_err("called with symbolic factor"); return KVX_ESYMBOLIC; }
    int rc = wait_for_caller(F);
    if (rc) return rc;
    return solve_dev(F, sys, B_dev, nrhs, ldB, true);
}

int kvx_chol_solve_async_dev(kvx_chol *F, int sys, double *B_dev, int64_t nrhs, int64_t ldB)
{
    return guarded([&] { return kvx_chol_solve_async_dev_impl(F, sys, B_dev, nrhs, ldB); });
}

int kvx_chol_factorize_solve_async_dev(kvx_chol *F, const double *values_dev, double *B_dev, int64_t nrhs, int64_t ldB)
{
    return guarded([&] {
        if (!F) return (int)KVX_EINVAL;
        return factor_solve_dev(F, values_dev, B_dev, nrhs, ldB, true);
    });
}

int kvx_chol_factorize_solve_dev(kvx_chol *F, const double *values_dev, double *B_dev, int64_t nrhs, int64_t ldB, int64_t *minor)
{
    return guarded([&] {
        if (!F) return (int)KVX_EINVAL;
        int rc = factor_solve_dev(F, values_dev, B_dev, nrhs, ldB);
        if (minor) *minor = F->minor;
        return rc;
    });
}

static int kvx_chol_solve_impl(kvx_chol *F, int sys, double *B, int64_t nrhs, int64_t ldB)
{
    if (!F) return KVX_EINVAL;
    if (!F->dev_ready) { set_err("called with symbolic factor"); return KVX_ESYMBOLIC; }
    const int64_t n = F->S.n;
    if (sys < 0 || sys > 8) { set_err("invalid value for sys"); return KVX_EINVAL; }
    int rc = finish_factor(F, nullptr);
    if (rc == KVX_ESYMBOLIC) { set_err("called with symbolic factor"); return rc; }
    if (rc == KVX_ENOTPOSDEF) { set_err("singular matrix"); return KVX_ESINGULAR; }
    if (n == 0 || nrhs == 0) return KVX_OK;
    if (ldB < std::max<int64_t>(1, n)) { set_err("ldB must be >= max(1,n)"); return KVX_EINVAL; }
    double *d_B = nullptr;
    HIPCHK(pool_malloc((void **)&d_B, n * nrhs * sizeof(double)));
    hipError_t e = hipMemcpy2D(d_B, n * sizeof(double), B, ldB * sizeof(double), n * sizeof(double), nrhs, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = solve_dev(F, sys, d_B, nrhs, n);
        if (rc == KVX_OK)
            e = hipMemcpy2D(B, ldB * sizeof(double), d_B, n * sizeof(double), n * sizeof(double), nrhs, hipMemcpyDeviceToHost);
    }
    (void)pool_free(d_B);
    if (e != hipSuccess) { set_err(hipGetErrorString(e)); return KVX_EDEVICE; }
    return rc;
}

int kvx_chol_solve(kvx_chol *F, int sys, double *B, int64_t nrhs, int64_t ldB)
{
    return guarded([&] { return kvx_chol_solve_impl(F, sys, B, nrhs, ldB); });
}

// numeric + solve (sys 0) with HOST buffers: what cholmod.linsolve does after its analysis (cholmod.c:663-753), as the one-enqueue
// form.  The staging block of B keeps its address from call to call (the captured graph reads and writes it).
static int kvx_chol_factorize_solve_impl(kvx_chol *F, const double *values, double *B, int64_t nrhs, int64_t ldB, int64_t *minor)
{
    if (!F) return KVX_EINVAL;
    int rc = ensure_device(F);
    if (rc) return rc;
    const int64_t n = F->S.n;
    if (nrhs < 0) { set_err("nrhs out of range"); return KVX_EINVAL; }
    if (n > 0 && nrhs > 0 && ldB < n) { set_err("ldB must be >= max(1,n)"); return KVX_EINVAL; }
    double *d_vals = nullptr, *d_B = nullptr;
    HIPCHK(pool_malloc((void **)&d_vals, std::max<int64_t>(F->S.nnzA, 1) * sizeof(double)));
    if (pool_malloc((void **)&d_B, std::max<int64_t>(n * nrhs, 1) * sizeof(double)) != hipSuccess) { (void)pool_free(d_vals); set_err("out of device memory"); return KVX_ENOMEM; }
    hipError_t e = hipSuccess;
    if (F->S.nnzA > 0) e = hipMemcpy(d_vals, values, F->S.nnzA * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess && n > 0 && nrhs > 0)
        e = hipMemcpy2D(d_B, n * sizeof(double), B, ldB * sizeof(double), n * sizeof(double), nrhs, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = factor_solve_dev(F, d_vals, d_B, nrhs, n);
        if (minor) *minor = F->minor;
        if (rc == KVX_OK && n > 0 && nrhs > 0)
            e = hipMemcpy2D(B, ldB * sizeof(double), d_B, n * sizeof(double), n * sizeof(double), nrhs, hipMemcpyDeviceToHost);
    }
    (void)pool_free(d_vals);
    (void)pool_free(d_B);
    if (e != hipSuccess) { set_err(hipGetErrorString(e)); return KVX_EDEVICE; }
    return rc;
}

int kvx_chol_factorize_solve(kvx_chol *F, const double *values, double *B, int64_t nrhs, int64_t ldB, int64_t *minor)
{
    return guarded([&] { return kvx_chol_factorize_solve_impl(F, values, B, nrhs, ldB, minor); });
}

// Sparse right-hand sides, forward systems (L x = b, L D x = b): only the REACH of a block of columns is swept -- the fronts
// that hold a nonzero row of the block and their ancestors in the supernodal elimination tree (the supernodal form of
// CHOLMOD's sparse-rhs solve, cholmod.c:524-587; misc.kkt_chol2 forms L^-1 P A' this way, misc.py:1483-1487).  Everything
// outside the reach is zero and is neither computed nor copied back.  Per block of up to 64 columns: host marks the reach
// (leaf subtrees are taken whole: they are one launch anyway), uploads the filtered level lists, the device sweeps them with
// the ordinary forward kernels (the update vectors of children outside the reach are cleared first -- the parents pull them),
// and only the rows of the swept fronts come back.
static int spsolve_forward_reach(kvx_chol *F, int sys, int64_t ncol, const int64_t *Bp, const int64_t *Bi, const double *Bx,
                                 std::vector<int64_t> &xp, std::vector<int64_t> &xi, std::vector<double> &xx)
{
    Symbolic &S = F->S;
    const int64_t n = S.n, ns = S.nsuper;
    hipStream_t st = F->stream;
    if (F->col2sn.empty()) {
        F->col2sn.resize((size_t)n);
        for (int64_t s = 0; s < ns; s++)
            for (int64_t c = S.super[s]; c < S.super[s + 1]; c++) F->col2sn[(size_t)c] = (int32_t)s;
        F->sub_of.assign((size_t)ns, -1);
        for (size_t i = 0; i < F->subs_host.size(); i++)
            for (int q = F->subs_host[i].lo; q <= F->subs_host[i].hi; q++) F->sub_of[(size_t)q] = (int32_t)i;
    }
    const bool subs_on = F->nsub > 0;
    const int64_t wstride = std::max(S.wrk_size[0], S.wrk_size[1]);
    const int64_t chunk = 64;
    std::vector<uint8_t> mark((size_t)ns, 0), submark(F->subs_host.size(), 0);
    std::vector<int32_t> touched;                            // fronts marked in this block (for the reset)
    std::vector<int64_t> pos;
    std::vector<double> val, back;
    std::vector<int32_t> lists, rows;
    std::vector<int64_t> slots;
    std::vector<SubDesc> subs;
    int rc;
    for (int64_t c0 = 0; c0 < ncol; c0 += chunk) {
        const int nc = (int)std::min<int64_t>(chunk, ncol - c0);
        touched.clear(); pos.clear(); val.clear();
        std::vector<size_t> touched_subs;
        for (int j = 0; j < nc; j++) {
            const size_t first = pos.size();
            for (int64_t p = Bp[c0 + j]; p < Bp[c0 + j + 1]; p++) {
                const int64_t r = Bi[p];
                if (r < 0 || r >= n) { set_err("row index out of range in B"); return KVX_EINVAL; }
                bool dup = false;
                for (size_t q = first; q < pos.size() && !dup; q++)       // (columns are short; duplicates are summed as the dense path does)
                    if (pos[q] == r + (int64_t)j * n) { val[q] += Bx[p]; dup = true; }
                if (!dup) { pos.push_back(r + (int64_t)j * n); val.push_back(Bx[p]); }
                for (int32_t f = F->col2sn[(size_t)r]; f >= 0 && !mark[(size_t)f]; f = S.sparent[(size_t)f]) {
                    mark[(size_t)f] = 1;
                    touched.push_back(f);
                }
            }
        }
        if (subs_on)
            for (size_t t = 0, e = touched.size(); t < e; t++) {                  // a touched subtree is swept whole
                const int32_t sb = F->sub_of[(size_t)touched[t]];
                if (sb < 0 || submark[(size_t)sb]) continue;
                submark[(size_t)sb] = 1;
                touched_subs.push_back((size_t)sb);
                for (int q = F->subs_host[(size_t)sb].lo; q <= F->subs_host[(size_t)sb].hi; q++)
                    if (!mark[(size_t)q]) { mark[(size_t)q] = 1; touched.push_back(q); }
            }
        // filtered lists: per level [big | lds (unmerged mode only) | small], the slots to clear, the subtrees, the rows to fetch
        struct Lv { int64_t big, lds, sw, zs; int nbig, nlds, nsw, nz; };
        std::vector<Lv> lv((size_t)S.nlevels);
        lists.clear(); slots.clear(); subs.clear(); rows.clear();
        for (size_t sb : touched_subs) subs.push_back(F->subs_host[sb]);
        for (int l = 0; l < S.nlevels; l++) {
            const LevelPlan &P = F->plan[l];
            Lv &v = lv[(size_t)l];
            auto take = [&](const int32_t *src, int cnt, int64_t &off, int &out) {
                off = (int64_t)lists.size();
                for (int i = 0; i < cnt; i++)
                    if (mark[(size_t)src[i]]) lists.push_back(src[i]);
                out = (int)((int64_t)lists.size() - off);
            };
            take(S.levellist.data() + P.soff[0], P.scnt[0], v.big, v.nbig);
            if (F->solve_merged) { v.lds = 0; v.nlds = 0; }
            else take(S.levellist.data() + P.soff[1], P.scnt[1], v.lds, v.nlds);
            take(F->lsw_host.data() + F->sw_off[l], F->sw_cnt[l], v.sw, v.nsw);
            v.zs = (int64_t)slots.size() / 2;
            for (int64_t q = v.big; q < (int64_t)lists.size(); q++) {
                const int32_t f = lists[(size_t)q];
                for (int64_t c = S.childptr[f]; c < S.childptr[f + 1]; c++) {
                    const int32_t ch = S.children[(size_t)c];
                    if (!mark[(size_t)ch] && S.sn_m[ch] > S.sn_k[ch]) { slots.push_back(S.wx[ch]); slots.push_back(S.sn_m[ch] - S.sn_k[ch]); }
                }
            }
            v.nz = (int)((int64_t)slots.size() / 2 - v.zs);
        }
        std::sort(touched.begin(), touched.end());
        for (int32_t f : touched)
            for (int64_t c = S.super[f]; c < S.super[f + 1]; c++) rows.push_back((int32_t)c);
        const int64_t nrow = (int64_t)rows.size();
        // device side
        if ((rc = ensure_solve_ws(F, nc))) return rc;
        int32_t *d_l = nullptr, *d_rows = nullptr;
        int64_t *d_slots = nullptr, *d_pos = nullptr;
        double *d_val = nullptr, *d_back = nullptr;
        SubDesc *d_sb = nullptr;
        auto release = [&] {
            for (void *q : {(void *)d_l, (void *)d_rows, (void *)d_slots, (void *)d_pos, (void *)d_val, (void *)d_back, (void *)d_sb})
                if (q) (void)pool_free(q);
        };
        auto up = [&](void **dst, const void *src, size_t bytes) -> int {
            HIPCHK(pool_malloc(dst, std::max<size_t>(bytes, 8)));
            if (bytes) HIPCHK(hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, st));
            return KVX_OK;
        };
        rc = up((void **)&d_l, lists.data(), lists.size() * sizeof(int32_t));
        if (!rc) rc = up((void **)&d_rows, rows.data(), rows.size() * sizeof(int32_t));
        if (!rc) rc = up((void **)&d_slots, slots.data(), slots.size() * sizeof(int64_t));
        if (!rc) rc = up((void **)&d_pos, pos.data(), pos.size() * sizeof(int64_t));
        if (!rc) rc = up((void **)&d_val, val.data(), val.size() * sizeof(double));
        if (!rc) rc = up((void **)&d_sb, subs.data(), subs.size() * sizeof(SubDesc));
        if (!rc && hipSuccess != pool_malloc((void **)&d_back, std::max<size_t>((size_t)(nrow * nc), 1) * sizeof(double))) rc = KVX_EDEVICE;
        if (rc) { release(); return rc; }
        auto body = [&]() -> int {
            HIPCHK(hipMemsetAsync(F->d_X, 0, (size_t)n * nc * sizeof(double), st));
            launch_scatter_entries(st, d_pos, d_val, (int64_t)pos.size(), F->d_X);
            HIPCHK(hipMemcpyAsync(F->d_X0, F->d_X, (size_t)n * nc * sizeof(double), hipMemcpyDeviceToDevice, st));
            if (!subs.empty())
                launch_fwd_subtree(st, F->ds, d_sb, (int)subs.size(), F->d_cd_woff, F->d_Lx, F->d_X, n, nc, F->d_W[0], F->d_W[1], wstride, F->d_depth);
            for (int l = S.nlevels - 1; l >= 0; l--) {
                const LevelPlan &P = F->plan[l];
                const Lv &v = lv[(size_t)l];
                if (v.nbig + v.nlds + v.nsw == 0) continue;
                double *Wch = F->d_W[(l + 1) & 1], *Wout = F->d_W[l & 1];
                launch_zero_slots(st, d_slots + 2 * v.zs, v.nz, nc, Wch, wstride);
                if (v.nsw > 0) {
                    if (F->solve_merged) launch_fwd_lds(st, F->ds, d_l + v.sw, v.nsw, F->sw_kmax[l], F->d_Lx, F->d_X, n, nc, Wch, Wout, wstride);
                    else launch_fwd_wave(st, F->ds, d_l + v.sw, v.nsw, 32, F->d_Lx, F->d_X, n, nc, Wch, Wout, wstride);
                }
                if (v.nlds > 0)
                    launch_fwd_lds(st, F->ds, d_l + v.lds, v.nlds, std::max(P.maxk[KVX_CLS_LDS128], P.maxk[KVX_CLS_LDS96]), F->d_Lx, F->d_X,
                                   n, nc, Wch, Wout, wstride);
                if (v.nbig > 0)
                    launch_fwd_big(st, F->ds, d_l + v.big, v.nbig, P.smaxm[0], P.big_maxk, F->d_Lx, F->d_Linv, F->d_X, F->d_X0, n, nc,
                                   F->d_WK, S.n, Wch, Wout, wstride, P.scnt[0]);
            }
            if (!F->is_ll) {                                      // LDL' view: L D x = b -> diag^-1 Lc^-1 b;  L x = b -> diag Lc^-1 b
                if (!F->diag_valid) {
                    if (!F->d_diag) HIPCHK(pool_malloc((void **)&F->d_diag, (size_t)n * sizeof(double)));
                    launch_extract_diag(st, F->ds, S.nsuper, F->d_Lx, F->d_diag);
                    F->diag_valid = true;
                }
                launch_diag_scale(st, n, nc, F->d_diag, F->d_X, n, sys == 2 ? 1 : 0);
            }
            launch_perm_gather(st, d_rows, nrow, nc, F->d_X, n, d_back, nrow);
            HIPCHK(hipGetLastError());
            back.resize((size_t)(nrow * nc));
            if (nrow * nc) HIPCHK(hipMemcpyAsync(back.data(), d_back, back.size() * sizeof(double), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            return KVX_OK;
        };
        rc = body();
        release();
        if (rc) return rc;
        for (int j = 0; j < nc; j++) {
            for (int64_t i = 0; i < nrow; i++) {
                const double v = back[(size_t)(i + (int64_t)j * nrow)];
                if (v != 0.0) { xi.push_back(rows[(size_t)i]); xx.push_back(v); }
            }
            xp[(size_t)(c0 + j + 1)] = (int64_t)xi.size();
        }
        for (int32_t f : touched) mark[(size_t)f] = 0;
        for (size_t sb : touched_subs) submark[sb] = 0;
    }
    return KVX_OK;
}

static int kvx_chol_spsolve_impl(kvx_chol *F, int sys, int64_t ncol, const int64_t *Bp, const int64_t *Bi, const double *Bx,
                     int64_t **Xp, int64_t **Xi, double **Xx)
{
    if (!F || !Xp || !Xi || !Xx || ncol < 0) return KVX_EINVAL;
    const int64_t n = F->S.n;
    if (sys < 0 || sys > 8) { set_err("invalid value for sys"); return KVX_EINVAL; }
    if (!F->dev_ready) { set_err("called with symbolic factor"); return KVX_ESYMBOLIC; }
    int rc = finish_factor(F, nullptr);
    if (rc == KVX_ESYMBOLIC) { set_err("called with symbolic factor"); return rc; }
    if (rc == KVX_ENOTPOSDEF) { set_err("singular matrix"); return KVX_ESINGULAR; }
    std::vector<int64_t> xp((size_t)ncol + 1, 0), xi;
    std::vector<double> xx;
    auto deliver = [&]() -> int {
        for (int64_t j = 0; j < ncol; j++) xp[j + 1] = std::max(xp[j + 1], xp[j]);
        *Xp = (int64_t *)malloc(sizeof(int64_t) * (ncol + 1));
        *Xi = (int64_t *)malloc(sizeof(int64_t) * std::max<size_t>(xi.size(), 1));
        *Xx = (double *)malloc(sizeof(double) * std::max<size_t>(xx.size(), 1));
        if (!*Xp || !*Xi || !*Xx) { free(*Xp); free(*Xi); free(*Xx); return KVX_ENOMEM; }
        memcpy(*Xp, xp.data(), sizeof(int64_t) * (ncol + 1));
        if (!xi.empty()) { memcpy(*Xi, xi.data(), sizeof(int64_t) * xi.size()); memcpy(*Xx, xx.data(), sizeof(double) * xx.size()); }
        return KVX_OK;
    };
    // forward systems: only the reach of the columns is swept (KVX_SPSOLVE_DENSE=1: the dense column blocks below, for comparison)
    if ((sys == 2 || sys == 4) && n > 0 && ncol > 0 && F->dist_nranks == 1 && !getenv("KVX_SPSOLVE_DENSE")) {
        if ((rc = wait_for_caller(F))) return rc;
        if ((rc = spsolve_forward_reach(F, sys, ncol, Bp, Bi, Bx, xp, xi, xx))) return rc;
        return deliver();
    }
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(ncol, (int64_t)(1 << 26) / std::max<int64_t>(n, 1)));
    std::vector<double> dense;
    for (int64_t c0 = 0; c0 < ncol && n > 0; c0 += chunk) {
        int64_t nc = std::min(chunk, ncol - c0);
        dense.assign((size_t)(n * nc), 0.0);
        for (int64_t j = 0; j < nc; j++)
            for (int64_t p = Bp[c0 + j]; p < Bp[c0 + j + 1]; p++) {
                if (Bi[p] < 0 || Bi[p] >= n) { set_err("row index out of range in B"); return KVX_EINVAL; }
                dense[(size_t)(Bi[p] + j * n)] += Bx[p];
            }
        rc = kvx_chol_solve(F, sys, dense.data(), nc, n);
        if (rc) return rc;
        for (int64_t j = 0; j < nc; j++) {
            for (int64_t i = 0; i < n; i++) {
                double v = dense[(size_t)(i + j * n)];
                if (v != 0.0) { xi.push_back(i); xx.push_back(v); }
            }
            xp[(size_t)(c0 + j + 1)] = (int64_t)xi.size();
        }
    }
    return deliver();
}

int kvx_chol_spsolve(kvx_chol *F, int sys, int64_t ncol, const int64_t *Bp, const int64_t *Bi, const double *Bx,
                     int64_t **Xp, int64_t **Xi, double **Xx)
{
    return guarded([&] { return kvx_chol_spsolve_impl(F, sys, ncol, Bp, Bi, Bx, Xp, Xi, Xx); });
}

int kvx_chol_diag(kvx_chol *F, double *d)
{
    if (!F || !d) return KVX_EINVAL;
    if (!F->part.empty()) { set_err("diag: this rank holds only its own fronts of a sharded factor"); return KVX_EINVAL; }
    if (!F->dev_ready || !F->is_ll) { set_err("F must be a nonsingular supernodal Cholesky factor"); return KVX_ESYMBOLIC; }
    int rc = finish_factor(F, nullptr);
    if (rc == KVX_ENOTPOSDEF) { set_err("F must be a nonsingular supernodal Cholesky factor"); return KVX_ESINGULAR; }
    if (rc) return rc;
    if (F->S.n == 0) return KVX_OK;
    double *dd = nullptr;
    HIPCHK(pool_malloc((void **)&dd, F->S.n * sizeof(double)));
    launch_extract_diag(F->stream, F->ds, F->S.nsuper, F->d_Lx, dd);
    hipError_t e = hipStreamSynchronize(F->stream);
    if (e == hipSuccess) e = hipMemcpy(d, dd, F->S.n * sizeof(double), hipMemcpyDeviceToHost);
    (void)pool_free(dd);
    if (e != hipSuccess) { set_err(hipGetErrorString(e)); return KVX_EDEVICE; }
    return KVX_OK;
}

static int kvx_chol_get_factor_impl(kvx_chol *F, int64_t *lnz, int64_t *Lp, int64_t *Li, double *Lx)
{
    if (!F) return KVX_EINVAL;
    if (!F->part.empty()) { set_err("getfactor: this rank holds only its own fronts of a sharded factor"); return KVX_EINVAL; }
    Symbolic &S = F->S;
    // structural entries of the supernodal factor: lower trapezoid of every panel
    int64_t cnt = 0;
    for (int64_t s = 0; s < S.nsuper; s++) {
        int64_t k = S.sn_k[s], m = S.sn_m[s];
        cnt += k * m - k * (k - 1) / 2;
    }
    if (lnz) *lnz = cnt;
    if (!Lp && !Li && !Lx) return KVX_OK;
    if (!F->dev_ready) { set_err("F must be a numeric Cholesky factor"); return KVX_ESYMBOLIC; }
    int rc = finish_factor(F, nullptr);
    if (rc == KVX_ESYMBOLIC) { set_err("F must be a numeric Cholesky factor"); return rc; }
    std::vector<double> host((size_t)std::max<int64_t>(S.lsize, 1));
    if (S.lsize > 0) HIPCHK(hipMemcpy(host.data(), F->d_Lx, S.lsize * sizeof(double), hipMemcpyDeviceToHost));
    int64_t q = 0;
    for (int64_t s = 0; s < S.nsuper; s++) {
        int64_t k = S.sn_k[s], m = S.sn_m[s], f = S.super[s];
        const int32_t *rows = S.rowidx.data() + S.rowptr[s];
        for (int64_t j = 0; j < k; j++) {
            if (Lp) Lp[f + j] = q;
            // LDL' form (as cholmod_factor_to_sparse returns it): D on the diagonal, the unit diagonal of L implicit
            const double dj = host[(size_t)(S.px[s] + j + j * m)];
            for (int64_t i = j; i < m; i++) {
                if (Li) Li[q] = rows[i];
                if (Lx) {
                    const double v = host[(size_t)(S.px[s] + i + j * m)];
                    Lx[q] = F->is_ll ? v : (i == j ? v * v : v / dj);
                }
                q++;
            }
        }
    }
    if (Lp) Lp[S.n] = q;
    return KVX_OK;
}

int kvx_chol_get_factor(kvx_chol *F, int64_t *lnz, int64_t *Lp, int64_t *Li, double *Lx)
{
    return guarded([&] { return kvx_chol_get_factor_impl(F, lnz, Lp, Li, Lx); });
}

int kvx_chol_get_info(kvx_chol *F, kvx_chol_info *info)
{
    if (!F || !info) return KVX_EINVAL;
    memset(info, 0, sizeof(*info));
    Symbolic &S = F->S;
    info->n = S.n;
    info->nnz_a = S.nnzTri;
    info->lnz = S.lnz;
    info->flops = S.flops;
    info->nsuper = S.nsuper;
    info->lsize = F->lsize_total >= 0 ? F->lsize_total : S.lsize;
    info->dev_bytes = F->dev_bytes;                   // bytes of the large device buffers this handle holds (0 before the first device use)
    info->lsize_local = S.lsize;                      // panel doubles resident on THIS rank (= lsize unless the sharded layout was trimmed)
    info->nlevels = S.nlevels;
    info->max_front = S.max_m;
    info->upd_size = S.upd_size[0] + S.upd_size[1];
    info->is_numeric = (F->numeric && !F->pending) ? 1 : 0;
    info->minor = F->minor;
    info->solve_rowidx = S.sum_m;
    info->is_ll = F->is_ll ? 1 : 0;
    return KVX_OK;
}

int kvx_chol_get_perm(kvx_chol *F, int64_t *perm)
{
    if (!F || (!perm && F->S.n > 0)) return KVX_EINVAL;
    if (F->S.n > 0) memcpy(perm, F->S.perm.data(), sizeof(int64_t) * F->S.n);
    return KVX_OK;
}

static int kvx_chol_get_supernodes_impl(kvx_chol *F, int64_t *super, int64_t *nrows, int64_t *parent, int64_t *level)
{
    if (!F) return KVX_EINVAL;
    Symbolic &S = F->S;
    if (super) memcpy(super, S.super.data(), sizeof(int64_t) * (S.nsuper + 1));
    for (int64_t s = 0; s < S.nsuper; s++) {
        if (nrows) nrows[s] = S.sn_m[s];
        if (parent) parent[s] = S.sparent[s];
        if (level) level[s] = S.depth[s];
    }
    return KVX_OK;
}

int kvx_chol_get_supernodes(kvx_chol *F, int64_t *super, int64_t *nrows, int64_t *parent, int64_t *level)
{
    return guarded([&] { return kvx_chol_get_supernodes_impl(F, super, nrows, parent, level); });
}

int kvx_chol_get_front_rows(kvx_chol *F, int64_t *rowptr, int64_t *rowidx)
{
    if (!F || !rowptr) return KVX_EINVAL;
    const Symbolic &S = F->S;
    std::copy(S.rowptr.begin(), S.rowptr.end(), rowptr);
    if (rowidx) std::copy(S.rowidx.begin(), S.rowidx.end(), rowidx);
    return KVX_OK;
}

int kvx_chol_last_timing(kvx_chol *F, double *ms_factor, double *ms_solve)
{
    if (!F) return KVX_EINVAL;
    if (F->pending) finish_factor(F, nullptr);
    if (ms_factor) *ms_factor = F->have_ftime ? F->ms_factor : -1.0;
    if (ms_solve) *ms_solve = F->have_stime ? F->ms_solve : -1.0;
    return KVX_OK;
}

int kvx_chol_last_fused_path(kvx_chol *F) { return F ? F->last_fused_path : 0; }

int kvx_chol_prof_select(kvx_chol *F, int family)
{
    if (!F || family < -1 || family > 8) return KVX_EINVAL;
    if (F->pending) finish_factor(F, nullptr);
    F->prof_family = family;
    F->prof_ms = 0;
    F->prof_launches = 0;
    F->prof_used = 0;
    return KVX_OK;
}

int kvx_chol_prof_read(kvx_chol *F, double *total_ms, int64_t *launches)
{
    if (!F) return KVX_EINVAL;
    if (F->pending) finish_factor(F, nullptr);
    if (total_ms) *total_ms = F->prof_ms;
    if (launches) *launches = F->prof_launches;
    return KVX_OK;
}

void kvx_chol_free(kvx_chol *F)
{
    if (!F) return;
    const bool tim = getenv("KVX_FREE_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t0 = now();
    auto lap = [&](const char *what) { if (tim) { auto t1 = now(); fprintf(stderr, "  free %-10s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count()); t0 = t1; } };
    {                                              // (also after a device set-up that failed half way: every member starts out null)
        if (F->stream) (void)hipStreamSynchronize(F->stream);
        lap("sync");
        void *ptrs[] = {F->d_k, F->d_m, F->d_first, F->d_rowidx, F->d_rel, F->d_children, F->d_perm, F->d_lists,
                        F->d_px, F->d_rowptr, F->d_ux, F->d_wx, F->d_childptr, F->d_amap, F->d_sdst, F->d_ssrc, F->d_scptr, F->d_Lx, F->d_U[0], F->d_U[1],
                        F->d_Ax, F->d_X, F->d_X0, F->d_diag, F->d_W[0], F->d_W[1], F->d_status, F->d_WK, F->d_Linv, F->d_linv_off, F->d_fd, F->d_cd, F->d_tiles};
        for (void *p : ptrs)
            if (p) (void)pool_free(p);
        lap("buffers");
        if (F->h_status) (void)hipHostFree(F->h_status);
        lap("hostfree");
        for (int i = 0; i < 4; i++)
            if (F->ev[i]) pool_event_put(F->ev[i], true);
        F->g_factor.drop();
        for (auto &g : F->g_solve) g.exec.drop();
        for (auto &g : F->g_fused) g.exec.drop();
        lap("graphs");
        for (hipEvent_t e : F->prof_ev)
            if (e) pool_event_put(e, true);
        for (int i = 0; i < 4; i++) {
            if (F->side[i]) pool_stream_put(F->side[i]);
            if (F->ev_join[i]) pool_event_put(F->ev_join[i], false);
        }
        if (F->d_iperm) (void)pool_free(F->d_iperm);
        if (F->d_inv_ptr) (void)pool_free(F->d_inv_ptr);
        if (F->d_inv_src) (void)pool_free(F->d_inv_src);
        if (F->d_keep) (void)pool_free(F->d_keep);
        if (F->d_flists) (void)pool_free(F->d_flists);
        if (F->d_chain) (void)pool_free(F->d_chain);
        dist_release(F);
        for (void *p : {(void *)F->d_subs, (void *)F->d_subs_f, (void *)F->d_subs_lvl, (void *)F->d_cd_woff, (void *)F->d_lists_sw, (void *)F->d_depth})
            if (p) (void)pool_free(p);
        if (F->ev_fork) pool_event_put(F->ev_fork, false);
        if (F->ev_fork2) pool_event_put(F->ev_fork2, false);
        for (hipEvent_t e : F->ev_u)
            if (e) pool_event_put(e, false);
        if (F->ev_ujoin) pool_event_put(F->ev_ujoin, false);
        for (hipEvent_t e : F->ev_lvl)
            if (e) pool_event_put(e, false);
        for (int i = 0; i < 4; i++)
            if (F->ev_pipe[i]) pool_event_put(F->ev_pipe[i], false);
        if (F->ev_in) pool_event_put(F->ev_in, false);
        if (F->ev_out) pool_event_put(F->ev_out, false);
        if (F->stream) pool_stream_put(F->stream);
        lap("streams");
    }
    delete F;
    lap("delete");
}

void kvx_free(void *p) { free(p); }

int kvx_dev_malloc(void **p, int64_t bytes) { HIPCHK(pool_malloc(p, (size_t)std::max<int64_t>(bytes, 1))); return KVX_OK; }
int kvx_dev_free(void *p) { HIPCHK(pool_free(p)); return KVX_OK; }
int kvx_dev_upload(void *dst, const void *src, int64_t bytes) { if (bytes > 0) HIPCHK(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice)); return KVX_OK; }
int kvx_dev_download(void *dst, const void *src, int64_t bytes) { if (bytes > 0) HIPCHK(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost)); return KVX_OK; }
int kvx_dev_sync(void) { HIPCHK(hipDeviceSynchronize()); return KVX_OK; }
int kvx_dev_trim(void) { pool_release_all(); return KVX_OK; }
int kvx_dev_mem_info(int64_t *free_bytes, int64_t *total_bytes)
{
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return KVX_OK;
}

}  // extern "C"
