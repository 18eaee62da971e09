// C-ABI entry points of the sparse LU path (include/kvxhip.h, kvx_lu_*): what the reference's src/C/klu.c binds
// from SuiteSparse KLU (klu_analyze :141, klu_factor :161, klu_solve/klu_tsolve :187-198, klu_extract :444-449,
// Udiag/Rs/Pnum/Q for the determinant :760-822).  Device-only numeric phase: no CPU fallback.
#include "../../include/kvxhip.h"
#include "abi_guard.hpp"
#include "devpool.hpp"
#include "lu_device.hpp"
#include "lu_symbolic.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

using namespace kvx;

#define HIPCHK(call)                                                             \
    do {                                                                         \
        hipError_t e_ = (call);                                                  \
        if (e_ != hipSuccess) {                                                  \
            set_last_error(std::string(#call) + ": " + hipGetErrorString(e_));  \
            return KVX_EDEVICE;                                                  \
        }                                                                        \
    } while (0)

struct kvx_lu_sym {
    LuSymbolic Y;
};

struct kvx_lu_num {
    kvx_lu_sym *sym = nullptr;
    LuPlan P;
    int64_t n = 0, nnz = 0;
    bool dev = false, factored = false;
    hipStream_t st = nullptr, st2 = nullptr, st3 = nullptr;   // st2: the blocked big-front chain of a level runs beside its LDS fronts;
                                                              // st3: every other size class of a wide level
    std::vector<hipEvent_t> evA, evB, evC, evD;
    hipEvent_t ev0 = nullptr;
    LuFrontD *d_fr = nullptr;
    int32_t *d_rowidx = nullptr, *d_rel = nullptr, *d_children = nullptr, *d_adst = nullptr, *d_ai32 = nullptr;
    int32_t *d_ipiv = nullptr, *d_lperm = nullptr, *d_fail = nullptr, *d_lists = nullptr, *d_slists = nullptr;
    int32_t *d_fcol = nullptr, *d_frow = nullptr, *d_flevpos = nullptr;          // block triangular form: F by rows / by columns
    int64_t *d_fptr_r = nullptr, *d_fptr_c = nullptr, *d_fsrc_r = nullptr, *d_fsrc_c = nullptr;
    double *d_fval_r = nullptr, *d_fval_c = nullptr;
    int64_t *d_asrc = nullptr, *d_prow = nullptr, *d_qcol = nullptr;
    double *d_rinv = nullptr, *d_rmax = nullptr, *d_Lx = nullptr, *d_Ux = nullptr, *d_arena = nullptr, *d_Ax = nullptr;
    double *d_W = nullptr, *d_X = nullptr, *d_B = nullptr;
    void *d_struct = nullptr, *d_base = nullptr;               // one device block each: the plan's arrays + factor storage / the per-matrix arrays
    int64_t cap_rhs = 0;
    std::vector<int32_t> lvl_maxm, lvl_maxk, lvl_smallm, lvl_smallk;   // per level: all fronts / those swept by one workgroup
    double tol = 1e-3, stol = 1e-3;
    int64_t attempts = 0;
    // per level: does any big front of the level interchange rows in pivot block `step`?  Read from the recorded pivot sequence after
    // a factorisation; a refactorisation (same sequence) leaves out the interchange launch of every block that has none
    std::vector<std::vector<uint8_t>> swap_steps;
    // Launch graphs of the steady state (klu.c:296-308: refactorisation on the recorded pivot sequence, then solves): the launches of
    // a pass / of a solve captured once per (buffer addresses, right-hand sides) and replayed.  `version` changes with everything
    // a captured sequence depends on besides its key: the plan (front merges), the interchange flags, the work buffers.
    struct Graph {
        hipGraphExec_t exec = nullptr;
        const void *ptr = nullptr;
        int64_t a = 0, b = 0;
        uint64_t version = 0;
        const void *seen_ptr = nullptr;                           // the key of the previous call (a sequence is captured when a key comes twice in a row)
        int64_t seen_a = 0, seen_b = 0;
        uint64_t seen_version = 0;
        void drop() { if (exec) (void)hipGraphExecDestroy(exec); exec = nullptr; }
    };
    Graph g_pass, g_solve[2];
    uint64_t version = 1, swap_version = 0;                   // (swap_version: the interchange flags -- the passes depend on them, the solves do not)
    bool graphs_on = [] { const char *e = std::getenv("KVX_LU_GRAPH"); return !e || e[0] != '0'; }();
    int64_t graph_replays = 0;
    bool unblocked = std::getenv("KVX_LU_UNBLOCKED") != nullptr;   // debugging aid: big fronts by one workgroup each
};

namespace {

template <class T>
int dalloc(T **dst, int64_t count)
{
    HIPCHK(pool_malloc((void **)dst, (size_t)std::max<int64_t>(count, 1) * sizeof(T)));
    return KVX_OK;
}

struct LuLap {                                      // KVX_LU_TIMING=1: wall time of the phases of a numeric factorisation on stderr
    bool on = std::getenv("KVX_LU_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what)
    {
        if (!on) return;
        auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "  lu %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

// hipMalloc costs 0.1-0.7 ms on MI355X and a numeric object owns ~30 arrays: on a new pattern (nothing of the right size cached in
// the pool) that was most of a first klu.linsolve call.  The arrays of one (re)build share ONE device block and ONE host-to-device
// copy: uploads first (packed into a staging buffer in the same layout), then the uninitialised ones, every slot 256-byte aligned.
struct Arena {
    std::vector<void **> dst;
    std::vector<size_t> off, bytes;
    std::vector<const void *> src;
    size_t total = 0, upload_end = 0;
    void add(void **p, size_t b, const void *s)
    {
        dst.push_back(p); off.push_back(total); bytes.push_back(b); src.push_back(s);
        total += (std::max<size_t>(b, 1) + 255) & ~(size_t)255;
        if (s) upload_end = total;
    }
    template <class T> void up(T **p, const std::vector<T> &v) { add((void **)p, v.size() * sizeof(T), v.empty() ? (const void *)&total : (const void *)v.data()); }
    template <class T> void alloc(T **p, int64_t count) { add((void **)p, (size_t)std::max<int64_t>(count, 1) * sizeof(T), nullptr); }
    int commit(void **base)
    {
        HIPCHK(pool_malloc(base, std::max<size_t>(total, 256)));
        std::vector<char> stage(upload_end, 0);
        for (size_t i = 0; i < dst.size(); i++) {
            *dst[i] = (char *)*base + off[i];
            if (src[i] && bytes[i]) memcpy(stage.data() + off[i], src[i], bytes[i]);
        }
        if (upload_end) HIPCHK(hipMemcpy(*base, stage.data(), upload_end, hipMemcpyHostToDevice));
        return KVX_OK;
    }
};

void free_structure(kvx_lu_num *N)
{
    for (void *p : {(void *)N->d_struct, (void *)N->d_W, (void *)N->d_X, (void *)N->d_B})
        if (p) (void)pool_free(p);
    N->d_struct = nullptr;
    N->d_slists = N->d_fcol = N->d_frow = N->d_flevpos = nullptr;
    N->d_fptr_r = N->d_fptr_c = N->d_fsrc_r = N->d_fsrc_c = nullptr;
    N->d_fval_r = N->d_fval_c = nullptr;
    N->d_fr = nullptr; N->d_rowidx = N->d_rel = N->d_children = N->d_adst = N->d_ipiv = N->d_lperm = N->d_fail = N->d_lists = nullptr;
    N->d_asrc = N->d_prow = N->d_qcol = nullptr;
    N->d_Lx = N->d_Ux = N->d_arena = N->d_W = N->d_X = N->d_B = nullptr;
    N->cap_rhs = 0;
}

// (Re)build the plan from the symbolic object's merge state and upload it.
int upload_structure(kvx_lu_num *N)
{
    LuLap tl;
    N->version++;                                                 // (captured launch sequences belong to the old plan)
    free_structure(N);
    tl.lap("free structure");
    try {
        lu_build_plan(N->sym->Y, N->P);
    } catch (const std::bad_alloc &) {
        return KVX_ENOMEM;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return KVX_EINVAL;
    }
    const LuPlan &P = N->P;
    if ((int64_t)P.max_k > 8000) {                              // k_lu_fwd_big_init keeps the permuted pivot part in LDS
        set_last_error("LU pivot block of " + std::to_string(P.max_k) + " columns exceeds what the solve kernels hold in LDS");
        return KVX_EINVAL;
    }
    std::vector<LuFrontD> fd((size_t)P.nfront);
    for (int64_t f = 0; f < P.nfront; f++) {
        LuFrontD &F = fd[f];
        F.k = P.fr[f].k; F.m = P.fr[f].m; F.p0 = P.fr[f].p0; F.nchild = P.fr[f].nchild;
        F.px = P.px[f]; F.rowptr = P.rowptr[f]; F.childptr = P.childptr[f]; F.aptr = P.aptr[f];
        F.upd_off = P.upd_off[f]; F.wx = P.wx[f]; F.upd_ld = P.upd_ld[f]; F.acnt = (int32_t)(P.aptr[f + 1] - P.aptr[f]);
    }
    Arena A;
    A.up(&N->d_fr, fd);
    A.up(&N->d_rowidx, P.rowidx);
    A.up(&N->d_rel, P.rel);
    A.up(&N->d_children, P.children);
    A.up(&N->d_adst, P.a_dst);
    A.up(&N->d_asrc, P.a_src);
    A.up(&N->d_prow, P.prow);
    A.up(&N->d_qcol, P.qcol);
    A.up(&N->d_lists, P.levellist);
    A.up(&N->d_slists, P.stagelist);
    A.up(&N->d_fcol, P.fcol);
    A.up(&N->d_frow, P.frow);
    A.up(&N->d_flevpos, P.flevpos);
    A.up(&N->d_fptr_r, P.fptr_r);
    A.up(&N->d_fptr_c, P.fptr_c);
    A.up(&N->d_fsrc_r, P.fsrc_r);
    A.up(&N->d_fsrc_c, P.fsrc_c);
    A.alloc(&N->d_fval_r, (int64_t)P.fcol.size());
    A.alloc(&N->d_fval_c, (int64_t)P.fcol.size());
    A.alloc(&N->d_ipiv, N->n);
    A.alloc(&N->d_lperm, N->n);
    A.alloc(&N->d_fail, P.nfront);
    A.alloc(&N->d_Lx, P.lsize);
    A.alloc(&N->d_Ux, P.lsize);
    A.alloc(&N->d_arena, P.arena);
    tl.lap("build plan");
    if (int rc = A.commit(&N->d_struct)) return rc;
    tl.lap("upload plan");
    N->lvl_maxm.assign((size_t)P.nlevels, 0);
    N->lvl_maxk.assign((size_t)P.nlevels, 0);
    N->lvl_smallm.assign((size_t)P.nlevels, 0);
    N->lvl_smallk.assign((size_t)P.nlevels, 0);
    for (int32_t l = 0; l < P.nlevels; l++)
        for (int64_t q = P.levelptr[l]; q < P.levelptr[l + 1]; q++) {
            const LuFrontH &fh = P.fr[P.levellist[q]];
            N->lvl_maxm[l] = std::max(N->lvl_maxm[l], fh.m);
            N->lvl_maxk[l] = std::max(N->lvl_maxk[l], fh.k);
            if (fh.m <= KVX_LU_SOLVE_BIG_M) {
                N->lvl_smallm[l] = std::max(N->lvl_smallm[l], fh.m);
                N->lvl_smallk[l] = std::max(N->lvl_smallk[l], fh.k);
            }
        }
    return KVX_OK;
}

LuDev dev_view(const kvx_lu_num *N)
{
    LuDev d;
    d.fr = N->d_fr; d.rowidx = N->d_rowidx; d.rel = N->d_rel; d.children = N->d_children;
    d.a_src = N->d_asrc; d.a_dst = N->d_adst; d.ai32 = N->d_ai32; d.rinv = N->d_rinv;
    d.Lx = N->d_Lx; d.Ux = N->d_Ux; d.arena = N->d_arena; d.ipiv = N->d_ipiv; d.lperm = N->d_lperm; d.fail = N->d_fail;
    d.arena_size = N->P.arena;
    return d;
}

int lds_class(int m)
{
    static const int cls[] = {16, 32, 48, 64, 88, KVX_LU_LDS_M};
    for (int c : cls) if (m <= c) return c;
    return KVX_LU_LDS_M;
}

// Device-pointer entry points: the caller's producers (torch's default stream, the kvx_* kernels of the KKT layer) run on
// the legacy null stream, N->st is non-blocking: order it behind them explicitly (st2 / st3 fork from st).  Same contract
// as the Cholesky path (api.cpp wait_for_caller); documented in include/kvxhip.h.
int lu_wait_for_caller(kvx_lu_num *N)
{
    HIPCHK(hipEventRecord(N->ev0, nullptr));
    HIPCHK(hipStreamWaitEvent(N->st, N->ev0, 0));
    return KVX_OK;
}

// Run `body` (enqueues on N->st, forks to the side streams by events and joins them again) from a launch graph: replayed when
// `g` was captured under the same key, captured when the key of the previous call comes again, launch by launch otherwise.  Whatever goes wrong with capture or instantiation turns the
// graphs of this factor off; the launches then go out one by one as before.
template <class Body>
int run_graphed(kvx_lu_num *N, kvx_lu_num::Graph &g, const void *ptr, int64_t a, int64_t b, Body body)
{
    if (!N->graphs_on) return body();
    if (g.exec && g.ptr == ptr && g.a == a && g.b == b && g.version == N->version) {
        HIPCHK(hipGraphLaunch(g.exec, N->st));
        N->graph_replays++;
        return KVX_OK;
    }
    // A capture and an instantiation cost milliseconds: a caller that hands over another buffer at every call must not pay them
    // every time.  The launches go out one by one until the same key comes twice in a row.
    const bool again = g.seen_ptr == ptr && g.seen_a == a && g.seen_b == b && g.seen_version == N->version && ptr != nullptr;
    g.seen_ptr = ptr; g.seen_a = a; g.seen_b = b; g.seen_version = N->version;
    if (!again) return body();
    g.drop();
    if (hipStreamBeginCapture(N->st, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); N->graphs_on = false; return body(); }
    const int rc = body();
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(N->st, &graph);
    if (rc || e != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        N->graphs_on = false;
        return rc ? rc : body();
    }
    const hipError_t ei = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) { (void)hipGetLastError(); g.exec = nullptr; N->graphs_on = false; return body(); }
    g.ptr = ptr; g.a = a; g.b = b; g.version = N->version;
    HIPCHK(hipGraphLaunch(g.exec, N->st));
    return KVX_OK;
}

// One numeric pass over the current plan.  fail_host receives the per-front flags.
int enqueue_pass(kvx_lu_num *N, const double *Ax_dev, int reuse)
{
    const LuPlan &P = N->P;
    const LuDev d = dev_view(N);
    launch_lu_zero(N->n, N->d_rmax, N->st);                       // (a kernel, not a memset node of the launch graph)
    launch_lu_rowmax(N->nnz, N->d_ai32, Ax_dev, N->d_rmax, N->st);
    launch_lu_rinv(N->n, N->d_rmax, N->d_rinv, N->st);
    launch_lu_fvals((int64_t)P.fcol.size(), N->d_fsrc_r, Ax_dev, N->d_rinv, N->d_ai32, N->d_fval_r, N->st);
    launch_lu_fvals((int64_t)P.fcol.size(), N->d_fsrc_c, Ax_dev, N->d_rinv, N->d_ai32, N->d_fval_c, N->st);
    HIPCHK(hipEventRecord(N->ev0, N->st));
    HIPCHK(hipStreamWaitEvent(N->st2, N->ev0, 0));
    if (std::getenv("KVX_LU_DUMP_PLAN")) {                        // per level: fronts, LDS-resident, largest m / k / child count, children of the level
        for (int32_t l = P.nlevels - 1; l >= 0; l--) {
            int mm = 0, mk = 0, mc = 0; int64_t nc = 0, mku = 0;
            for (int64_t q = P.levelptr[l]; q < P.levelptr[l + 1]; q++) {
                const LuFrontH &f = P.fr[P.levellist[q]];
                mm = std::max(mm, f.m); mk = std::max(mk, f.k); mc = std::max(mc, f.nchild); nc += f.nchild;
                mku = std::max<int64_t>(mku, (int64_t)f.k * f.m);
            }
            fprintf(stderr, "  lu level %2d: %6lld fronts (%lld in LDS)  max m %4d  max k %4d  max children %3d  children %lld  max k*m %lld\n", (int)l,
                    (long long)(P.levelptr[l + 1] - P.levelptr[l]), (long long)P.nlds[l], mm, mk, mc, (long long)nc, (long long)mku);
        }
    }
    int32_t lastA = -1, lastB = -1;                            // deepest-so-far levels with work recorded on st / st2
    for (int32_t l = P.nlevels - 1; l >= 0; l--) {
        const int64_t b = P.levelptr[l], e = P.levelptr[l + 1], nl = P.nlds[l];
        const bool hasA = nl > 0, hasB = e > b + nl;
        // everything of the levels below must be complete: each stream waits for the other's latest record
        if (hasA && lastB >= 0) HIPCHK(hipStreamWaitEvent(N->st, N->evB[lastB], 0));
        if (hasB && lastA >= 0) HIPCHK(hipStreamWaitEvent(N->st2, N->evA[lastA], 0));
        int64_t q = b;
        int run = 0;
        bool forked = false;
        while (q < b + nl) {                                  // runs of one LDS size class (list sorted by m descending);
            const int c = lds_class(P.fr[P.levellist[q]].m);  // a thin level goes out as ONE launch sized for its largest front
            int64_t q2 = q;
            while (q2 < b + nl && (nl <= 256 || lds_class(P.fr[P.levellist[q2]].m) == c)) q2++;
            hipStream_t s = N->st;
            if (run & 1) {                                    // independent launches, each as long as its slowest front: alternate streams
                if (!forked) {
                    HIPCHK(hipEventRecord(N->evC[l], N->st));
                    HIPCHK(hipStreamWaitEvent(N->st3, N->evC[l], 0));
                    forked = true;
                }
                s = N->st3;
            }
            launch_lu_fronts(d, N->d_lists + q, (int)(q2 - q), c, 0, Ax_dev, N->tol, N->stol, reuse, s);
            q = q2;
            run++;
        }
        if (forked) {
            HIPCHK(hipEventRecord(N->evD[l], N->st3));
            HIPCHK(hipStreamWaitEvent(N->st, N->evD[l], 0));
        }
        if (hasB) {
            int bm = 0, bk = 0;                                // big fronts of the level: blocked multi-launch path
            for (int64_t qq = b + nl; qq < e; qq++) { bm = std::max(bm, P.fr[P.levellist[qq]].m); bk = std::max(bk, P.fr[P.levellist[qq]].k); }
            if (N->unblocked) launch_lu_fronts(d, N->d_lists + b + nl, (int)(e - b - nl), 0, bk, Ax_dev, N->tol, N->stol, reuse, N->st2);
            else {
                const uint8_t *sw = (reuse && (size_t)l < N->swap_steps.size() && !N->swap_steps[(size_t)l].empty()) ? N->swap_steps[(size_t)l].data() : nullptr;
                launch_lu_big_level(d, N->d_lists + b + nl, (int)(e - b - nl), bm, bk, Ax_dev, N->tol, N->stol, reuse, N->st2, sw);
            }
        }
        if (hasA) { HIPCHK(hipEventRecord(N->evA[l], N->st)); lastA = l; }
        if (hasB) { HIPCHK(hipEventRecord(N->evB[l], N->st2)); lastB = l; }
    }
    if (lastB >= 0) HIPCHK(hipStreamWaitEvent(N->st, N->evB[lastB], 0));
    else if (!N->evB.empty()) {                                   // st2 forked from st above: joined again whether or not it got work
        HIPCHK(hipEventRecord(N->evB[0], N->st2));
        HIPCHK(hipStreamWaitEvent(N->st, N->evB[0], 0));
    }
    HIPCHK(hipGetLastError());
    return KVX_OK;
}

int numeric_pass(kvx_lu_num *N, const double *Ax_dev, int reuse, std::vector<int32_t> &fail_host)
{
    const LuPlan &P = N->P;
    // the events of the level schedule exist before anything is captured
    while ((int32_t)N->evA.size() < P.nlevels) {
        hipEvent_t a, b2, c2, d2;
        HIPCHK(pool_event_get(&a, false));
        HIPCHK(pool_event_get(&b2, false));
        HIPCHK(pool_event_get(&c2, false));
        HIPCHK(pool_event_get(&d2, false));
        N->evA.push_back(a); N->evB.push_back(b2); N->evC.push_back(c2); N->evD.push_back(d2);
    }
    int rc;
    if (reuse) rc = run_graphed(N, N->g_pass, Ax_dev, (int64_t)N->swap_version, 0, [&] { return enqueue_pass(N, Ax_dev, reuse); });   // the steady state: replayed
    else rc = enqueue_pass(N, Ax_dev, reuse);
    if (rc) return rc;
    if (std::getenv("KVX_LU_TIMING")) fprintf(stderr, "  lu   (pass enqueued, %d levels)\n", (int)P.nlevels);
    fail_host.resize((size_t)P.nfront);
    HIPCHK(hipMemcpyAsync(fail_host.data(), N->d_fail, (size_t)P.nfront * sizeof(int32_t), hipMemcpyDeviceToHost, N->st));
    HIPCHK(hipStreamSynchronize(N->st));
    N->attempts++;
    return KVX_OK;
}

// After a factorisation that chose its pivots: which pivot blocks of the blocked fronts interchange rows at all (the blocks are
// those of launch_lu_big_level: lu_big_block_width).
int refresh_swap_steps(kvx_lu_num *N)
{
    const LuPlan &P = N->P;
    N->swap_version++;                                            // (the launches of a refactorisation depend on these flags)
    N->swap_steps.assign((size_t)P.nlevels, {});
    if (N->unblocked) return KVX_OK;
    bool any_big = false;
    for (int32_t l = 0; l < P.nlevels; l++) any_big = any_big || P.levelptr[l + 1] > P.levelptr[l] + P.nlds[l];
    if (!any_big) return KVX_OK;
    std::vector<int32_t> ipiv((size_t)N->n);
    HIPCHK(hipMemcpy(ipiv.data(), N->d_ipiv, (size_t)N->n * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int32_t l = 0; l < P.nlevels; l++) {
        const int64_t b = P.levelptr[l] + P.nlds[l], e = P.levelptr[l + 1];
        if (e <= b) continue;
        int bm = 0, bk = 0;
        for (int64_t q = b; q < e; q++) { bm = std::max(bm, P.fr[P.levellist[q]].m); bk = std::max(bk, P.fr[P.levellist[q]].k); }
        std::vector<uint8_t> &fl = N->swap_steps[(size_t)l];
        for (int jb = 0; jb < bk;) {
            const int nbs = lu_big_block_width(bm - jb);
            uint8_t any = 0;
            for (int64_t q = b; q < e && !any; q++) {
                const LuFrontH &f = P.fr[P.levellist[q]];
                for (int t = jb; t < std::min(jb + nbs, (int)f.k) && !any; t++) any = ipiv[(size_t)f.p0 + t] != 0;
            }
            fl.push_back(any);
            jb += nbs;
        }
    }
    return KVX_OK;
}

// Factor with the merge-and-retry loop of lu_symbolic.hpp (4).
int factor_loop(kvx_lu_num *N, const double *Ax_dev, int reuse)
{
    N->factored = false;
    if (N->sym->Y.structurally_singular) {
        set_last_error("singular matrix (structurally rank deficient)");
        return KVX_ESINGULAR;
    }
    std::vector<int32_t> fail;
    LuLap tl;
    for (int iter = 0; iter < 100000; iter++) {
        int rc = numeric_pass(N, Ax_dev, reuse, fail);
        if (rc) return rc;
        tl.lap("numeric pass");
        const LuPlan &P = N->P;
        std::vector<int32_t> minimal;
        std::vector<char> below((size_t)P.nfront, 0);
        for (int64_t f = 0; f < P.nfront; f++) {
            const bool flagged = fail[f] != 0;
            if (flagged && !below[f]) minimal.push_back((int32_t)f);
            if ((flagged || below[f]) && P.fr[f].parent >= 0) below[P.fr[f].parent] = 1;
        }
        if (minimal.empty()) {
            N->factored = true;
            if (!reuse && (rc = refresh_swap_steps(N))) return rc;
            return KVX_OK;
        }
        if (reuse) { reuse = 0; continue; }                   // klu.c:296-303: a refactorisation that runs into numerical trouble becomes a full one
        if (!lu_merge_fronts(N->sym->Y, P, minimal)) {
            set_last_error("singular matrix");
            return KVX_ESINGULAR;
        }
        tl.lap("merge fronts");
        if ((rc = upload_structure(N))) return rc;
        tl.t = std::chrono::steady_clock::now();
    }
    set_last_error("singular matrix");
    return KVX_ESINGULAR;
}

int ensure_device(kvx_lu_num *N)
{
    if (N->dev) return KVX_OK;
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) {
        set_last_error("no HIP device: the LU numeric phase has no CPU fallback");
        return KVX_EDEVICE;
    }
    LuLap tl;
    HIPCHK(pool_stream_get(&N->st));
    HIPCHK(pool_stream_get(&N->st2));
    HIPCHK(pool_stream_get(&N->st3));
    HIPCHK(pool_event_get(&N->ev0, false));
    tl.lap("streams + event");
    std::vector<int32_t> ai32((size_t)N->nnz);
    for (int64_t p = 0; p < N->nnz; p++) ai32[p] = (int32_t)N->sym->Y.Ai[p];
    Arena A;
    A.up(&N->d_ai32, ai32);
    A.alloc(&N->d_rinv, N->n);
    A.alloc(&N->d_rmax, N->n);
    A.alloc(&N->d_Ax, N->nnz);
    if (int rc = A.commit(&N->d_base)) return rc;
    tl.lap("per-matrix arrays");
    N->dev = true;
    return upload_structure(N);
}

int ensure_rhs(kvx_lu_num *N, int64_t nrhs)
{
    if (nrhs <= N->cap_rhs) return KVX_OK;
    if (N->d_W) (void)pool_free(N->d_W);
    if (N->d_X) (void)pool_free(N->d_X);
    if (N->d_B) (void)pool_free(N->d_B);
    N->d_W = N->d_X = N->d_B = nullptr;
    N->cap_rhs = 0;
    int rc;
    if ((rc = dalloc(&N->d_W, N->P.wsize * nrhs))) return rc;
    if ((rc = dalloc(&N->d_X, N->n * nrhs))) return rc;
    if ((rc = dalloc(&N->d_B, N->n * nrhs))) return rc;
    N->cap_rhs = nrhs;
    N->version++;                                                 // (new work buffers)
    return KVX_OK;
}

int enqueue_solve(kvx_lu_num *N, int trans, double *B_dev, int64_t nrhs, int64_t ldB);
int solve_on_device(kvx_lu_num *N, int trans, double *B_dev, int64_t nrhs, int64_t ldB)
{
    return run_graphed(N, N->g_solve[trans ? 1 : 0], B_dev, nrhs, ldB, [&] { return enqueue_solve(N, trans, B_dev, nrhs, ldB); });
}
int enqueue_solve(kvx_lu_num *N, int trans, double *B_dev, int64_t nrhs, int64_t ldB)
{
    const LuPlan &P = N->P;
    const LuDev d = dev_view(N);
    const int64_t n = N->n;
    // A x = b:  L U (Q' x) = R P b         A' x = b:  U' L' (R^-1 P x) = Q' b
    if (!trans) launch_lu_gather(n, (int)nrhs, N->d_prow, N->d_rinv, B_dev, ldB, N->d_X, n, N->st);
    else launch_lu_gather(n, (int)nrhs, N->d_qcol, nullptr, B_dev, ldB, N->d_X, n, N->st);
    // block levels (one without BTF); inside a level the stages are the tree depths of its blocks: forward sweep leaves ->
    // roots, backward sweep roots -> leaves.  A x = b walks the block levels upwards (a block after the later blocks its rows
    // touch), A' x = b downwards; the products with the off-diagonal blocks F come first.
    auto sweep = [&](int32_t t, bool fwd, int unit) {
        const int64_t sb = P.stageptr[t], se = P.stageptr[t + 1], nb = P.stage_nbig[t];
        if (fwd) {
            launch_lu_fwd(d, N->d_slists + sb, (int)(se - nb - sb), P.stage_smallm[t], P.stage_smallk[t], unit, N->d_X, n, (int)nrhs, N->d_W,
                          P.wsize, N->st);
            launch_lu_fwd_big(d, N->d_slists + se - nb, (int)nb, P.stage_bigm[t], P.stage_bigk[t], unit, N->d_X, n, (int)nrhs, N->d_W, P.wsize,
                              N->st);
        } else {
            launch_lu_bwd(d, N->d_slists + sb, (int)(se - nb - sb), P.stage_smallm[t], P.stage_smallk[t], unit, N->d_X, n, (int)nrhs, N->st);
            launch_lu_bwd_big(d, N->d_slists + se - nb, (int)nb, P.stage_bigm[t], P.stage_bigk[t], unit, N->d_X, n, (int)nrhs, N->d_W, P.wsize,
                              N->st);
        }
    };
    for (int32_t li = 0; li < P.nblev; li++) {
        const int32_t l = trans ? P.nblev - 1 - li : li;
        if (P.nblev > 1 && !P.fcol.empty()) {
            const int64_t cnt = P.flevptr[l + 1] - P.flevptr[l];
            if (!trans) launch_lu_fterm(cnt, (int)nrhs, N->d_flevpos + P.flevptr[l], N->d_fptr_r, N->d_fcol, N->d_fval_r, N->d_X, n, N->st);
            else launch_lu_fterm(cnt, (int)nrhs, N->d_flevpos + P.flevptr[l], N->d_fptr_c, N->d_frow, N->d_fval_c, N->d_X, n, N->st);
        }
        for (int32_t t = P.levstage[l + 1] - 1; t >= P.levstage[l]; t--) sweep(t, true, trans ? 0 : 1);
        for (int32_t t = P.levstage[l]; t < P.levstage[l + 1]; t++) sweep(t, false, trans ? 1 : 0);
    }
    if (!trans) launch_lu_scatter(n, (int)nrhs, N->d_qcol, nullptr, N->d_X, n, B_dev, ldB, N->st);
    else launch_lu_scatter(n, (int)nrhs, N->d_prow, N->d_rinv, N->d_X, n, B_dev, ldB, N->st);
    HIPCHK(hipGetLastError());
    return KVX_OK;
}

int64_t *mdup(const std::vector<int64_t> &v)
{
    int64_t *p = (int64_t *)std::malloc(std::max<size_t>(v.size(), 1) * sizeof(int64_t));
    if (p && !v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(int64_t));
    return p;
}
double *mdup(const std::vector<double> &v)
{
    double *p = (double *)std::malloc(std::max<size_t>(v.size(), 1) * sizeof(double));
    if (p && !v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(double));
    return p;
}

struct Trip { int64_t r, c; double v; };
void to_ccs(int64_t n, std::vector<Trip> &t, std::vector<int64_t> &ptr, std::vector<int64_t> &idx, std::vector<double> &val)
{
    std::sort(t.begin(), t.end(), [](const Trip &a, const Trip &b) { return a.c != b.c ? a.c < b.c : a.r < b.r; });
    ptr.assign((size_t)n + 1, 0);
    idx.resize(t.size());
    val.resize(t.size());
    for (size_t q = 0; q < t.size(); q++) { ptr[t[q].c + 1]++; idx[q] = t[q].r; val[q] = t[q].v; }
    for (int64_t j = 0; j < n; j++) ptr[j + 1] += ptr[j];
}

}  // namespace

extern "C" {

int kvx_lu_analyze(int64_t n, const int64_t *colptr, const int64_t *rowind, const double *values, kvx_lu_sym **out)
{
    if (!out || !colptr || (!rowind && n > 0 && colptr[n] > 0)) return KVX_EINVAL;
    *out = nullptr;
    kvx_lu_sym *S = new (std::nothrow) kvx_lu_sym();
    if (!S) return KVX_ENOMEM;
    try {
        lu_analyze(n, colptr, rowind, values, S->Y);
    } catch (const std::bad_alloc &) {
        delete S;
        return KVX_ENOMEM;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        delete S;
        return KVX_EINVAL;
    }
    *out = S;
    return KVX_OK;
}

void kvx_lu_free_symbolic(kvx_lu_sym *S) { delete S; }

void kvx_lu_free_numeric(kvx_lu_num *N)
{
    if (!N) return;
    if (N->st) (void)hipDeviceSynchronize();       // streams and events go back to the pool idle
    N->g_pass.drop();
    N->g_solve[0].drop();
    N->g_solve[1].drop();
    free_structure(N);
    if (N->d_base) (void)pool_free(N->d_base);    // d_ai32, d_rinv, d_rmax, d_Ax
    for (hipEvent_t e : N->evA) pool_event_put(e, false);
    for (hipEvent_t e : N->evB) pool_event_put(e, false);
    for (hipEvent_t e : N->evC) pool_event_put(e, false);
    for (hipEvent_t e : N->evD) pool_event_put(e, false);
    if (N->ev0) pool_event_put(N->ev0, false);
    if (N->st3) pool_stream_put(N->st3);
    if (N->st2) pool_stream_put(N->st2);
    if (N->st) pool_stream_put(N->st);
    delete N;
}

int kvx_lu_sym_info(kvx_lu_sym *S, int64_t info[8])
{
    if (!S || !info) return KVX_EINVAL;
    info[0] = S->Y.n; info[1] = S->Y.nnz; info[2] = S->Y.S.nsuper; info[3] = S->Y.nmerges;
    info[4] = S->Y.structurally_singular ? 1 : 0; info[5] = S->Y.S.lnz; info[6] = S->Y.S.nlevels; info[7] = S->Y.S.max_m;
    return KVX_OK;
}

static int kvx_lu_sym_btf_impl(kvx_lu_sym *S, int64_t *nblocks, int64_t *nlevels, int64_t *blk)
{
    if (!S || !nblocks || !nlevels) return KVX_EINVAL;
    *nblocks = S->Y.nblocks;
    *nlevels = S->Y.nblev;
    if (blk) for (int64_t j = 0; j < S->Y.n; j++) blk[j] = S->Y.blk[(size_t)j];
    return KVX_OK;
}

int kvx_lu_sym_btf(kvx_lu_sym *S, int64_t *nblocks, int64_t *nlevels, int64_t *blk)
{
    return guarded([&] { return kvx_lu_sym_btf_impl(S, nblocks, nlevels, blk); });
}

int kvx_lu_sym_matching(kvx_lu_sym *S, int64_t *rowfor)
{
    if (!S || !rowfor) return KVX_EINVAL;
    std::copy(S->Y.rowfor.begin(), S->Y.rowfor.end(), rowfor);
    return KVX_OK;
}

static int new_numeric(kvx_lu_sym *S, int64_t nnz, kvx_lu_num **out)
{
    if (!S || !out) return KVX_EINVAL;
    *out = nullptr;
    if (nnz != S->Y.nnz) { set_last_error("A does not have the analysed sparsity pattern"); return KVX_EINVAL; }
    kvx_lu_num *N = new (std::nothrow) kvx_lu_num();
    if (!N) return KVX_ENOMEM;
    N->sym = S; N->n = S->Y.n; N->nnz = S->Y.nnz;
    *out = N;
    return KVX_OK;
}

static int kvx_lu_factor_dev_impl(kvx_lu_sym *S, int64_t nnz, const double *values_dev, kvx_lu_num **out)
{
    int rc = new_numeric(S, nnz, out);
    if (rc) return rc;
    kvx_lu_num *N = *out;
    if ((rc = ensure_device(N)) || (rc = lu_wait_for_caller(N)) || (rc = factor_loop(N, values_dev, 0))) { kvx_lu_free_numeric(N); *out = nullptr; return rc; }
    return KVX_OK;
}

int kvx_lu_factor_dev(kvx_lu_sym *S, int64_t nnz, const double *values_dev, kvx_lu_num **out)
{
    return guarded([&] { return kvx_lu_factor_dev_impl(S, nnz, values_dev, out); });
}

static int kvx_lu_factor_impl(kvx_lu_sym *S, int64_t nnz, const double *values, kvx_lu_num **out)
{
    int rc = new_numeric(S, nnz, out);
    if (rc) return rc;
    kvx_lu_num *N = *out;
    if ((rc = ensure_device(N))) { kvx_lu_free_numeric(N); *out = nullptr; return rc; }
    if (hipMemcpy(N->d_Ax, values, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = KVX_EDEVICE;
    if (!rc) rc = factor_loop(N, N->d_Ax, 0);
    if (rc) { kvx_lu_free_numeric(N); *out = nullptr; return rc; }
    return KVX_OK;
}

int kvx_lu_factor(kvx_lu_sym *S, int64_t nnz, const double *values, kvx_lu_num **out)
{
    return guarded([&] { return kvx_lu_factor_impl(S, nnz, values, out); });
}

static int kvx_lu_refactor_dev_impl(kvx_lu_num *N, int64_t nnz, const double *values_dev)
{
    if (!N || nnz != N->nnz) return KVX_EINVAL;
    if (int rc = lu_wait_for_caller(N)) return rc;
    if (!N->factored) return factor_loop(N, values_dev, 0);
    return factor_loop(N, values_dev, 1);
}

int kvx_lu_refactor_dev(kvx_lu_num *N, int64_t nnz, const double *values_dev)
{
    return guarded([&] { return kvx_lu_refactor_dev_impl(N, nnz, values_dev); });
}

static int kvx_lu_refactor_impl(kvx_lu_num *N, int64_t nnz, const double *values)
{
    if (!N || nnz != N->nnz || !values) return KVX_EINVAL;
    HIPCHK(hipMemcpy(N->d_Ax, values, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice));
    return kvx_lu_refactor_dev(N, nnz, N->d_Ax);
}

int kvx_lu_refactor(kvx_lu_num *N, int64_t nnz, const double *values)
{
    return guarded([&] { return kvx_lu_refactor_impl(N, nnz, values); });
}

static int kvx_lu_solve_dev_impl(kvx_lu_num *N, int trans, double *B_dev, int64_t nrhs, int64_t ldB)
{
    if (!N || (trans != 0 && trans != 1) || nrhs < 0 || ldB < std::max<int64_t>(1, N ? N->n : 1)) return KVX_EINVAL;
    if (!N->factored) { set_last_error("singular matrix"); return KVX_ESINGULAR; }
    if (nrhs == 0) return KVX_OK;
    int rc = ensure_rhs(N, nrhs);
    if (rc) return rc;
    if ((rc = lu_wait_for_caller(N))) return rc;
    if ((rc = solve_on_device(N, trans, B_dev, nrhs, ldB))) return rc;
    HIPCHK(hipStreamSynchronize(N->st));
    return KVX_OK;
}

int kvx_lu_solve_dev(kvx_lu_num *N, int trans, double *B_dev, int64_t nrhs, int64_t ldB)
{
    return guarded([&] { return kvx_lu_solve_dev_impl(N, trans, B_dev, nrhs, ldB); });
}

static int kvx_lu_solve_impl(kvx_lu_num *N, int trans, double *B, int64_t nrhs, int64_t ldB)
{
    if (!N || !B || (trans != 0 && trans != 1) || nrhs < 0 || ldB < std::max<int64_t>(1, N->n)) return KVX_EINVAL;
    if (!N->factored) { set_last_error("singular matrix"); return KVX_ESINGULAR; }
    if (nrhs == 0) return KVX_OK;
    int rc = ensure_rhs(N, nrhs);
    if (rc) return rc;
    const int64_t n = N->n;
    HIPCHK(hipMemcpy2DAsync(N->d_B, (size_t)n * sizeof(double), B, (size_t)ldB * sizeof(double), (size_t)n * sizeof(double),
                            (size_t)nrhs, hipMemcpyHostToDevice, N->st));
    if ((rc = solve_on_device(N, trans, N->d_B, nrhs, n))) return rc;
    HIPCHK(hipMemcpy2DAsync(B, (size_t)ldB * sizeof(double), N->d_B, (size_t)n * sizeof(double), (size_t)n * sizeof(double),
                            (size_t)nrhs, hipMemcpyDeviceToHost, N->st));
    HIPCHK(hipStreamSynchronize(N->st));
    return KVX_OK;
}

int kvx_lu_solve(kvx_lu_num *N, int trans, double *B, int64_t nrhs, int64_t ldB)
{
    return guarded([&] { return kvx_lu_solve_impl(N, trans, B, nrhs, ldB); });
}

int kvx_lu_num_info(kvx_lu_num *N, int64_t info[8])
{
    if (!N || !info) return KVX_EINVAL;
    info[0] = N->P.nfront; info[1] = N->P.nlevels; info[2] = N->P.max_m; info[3] = N->P.max_k;
    info[4] = N->P.lsize; info[5] = N->P.arena; info[6] = N->attempts; info[7] = N->factored ? 1 : 0;
    return KVX_OK;
}

int kvx_lu_num_graph_replays(kvx_lu_num *N, int64_t *replays)
{
    if (!N || !replays) return KVX_EINVAL;
    *replays = N->graph_replays;
    return KVX_OK;
}

int kvx_lu_num_work(kvx_lu_num *N, double work[5])
{
    if (!N || !work) return KVX_EINVAL;
    for (int i = 0; i < 5; i++) work[i] = 0.0;
    for (const LuFrontH &f : N->P.fr) {
        const double k = f.k, m = f.m, u = m - k;
        // columns j = 0 .. k-1 of a front of order m: (m - j - 1) divisions + 2 (m - j - 1)^2 flops of the rank-1 update
        const double fl = 2.0 * (k * u * u + u * k * (k - 1.0) + (k - 1.0) * k * (2.0 * k - 1.0) / 6.0) + k * u + k * (k - 1.0) / 2.0;
        work[0] += fl;
        work[1] += 2.0 * m * k - k * k;
        work[2] += u * u;
        if (f.m > KVX_LU_LDS_M) { work[3] += 1.0; work[4] += fl; }
    }
    return KVX_OK;
}

static int kvx_lu_extract_impl(kvx_lu_num *N, int64_t *lnz, int64_t **Lp, int64_t **Li, double **Lx, int64_t *unz, int64_t **Up,
                   int64_t **Ui, double **Ux, int64_t *fnz, int64_t **Fp, int64_t **Fi, double **Fx, int64_t *P_out,
                   int64_t *Q_out, double *Rs, int64_t *nblocks, int64_t **r_out)
{
    if (!N || !lnz || !Lp || !Li || !Lx || !unz || !Up || !Ui || !Ux || !fnz || !Fp || !Fi || !Fx || !P_out || !Q_out || !Rs ||
        !nblocks || !r_out)
        return KVX_EINVAL;
    if (!N->factored) { set_last_error("singular matrix"); return KVX_ESINGULAR; }
    const LuPlan &P = N->P;
    const int64_t n = N->n;
    std::vector<double> hL((size_t)P.lsize), hU((size_t)P.lsize), rinv((size_t)n);
    std::vector<int32_t> lperm((size_t)n);
    HIPCHK(hipMemcpy(hL.data(), N->d_Lx, (size_t)P.lsize * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hU.data(), N->d_Ux, (size_t)P.lsize * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(rinv.data(), N->d_rinv, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(lperm.data(), N->d_lperm, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    std::vector<int64_t> finalpos((size_t)n);
    for (int64_t f = 0; f < P.nfront; f++) {
        const int32_t p0 = P.fr[f].p0, k = P.fr[f].k;
        for (int32_t t = 0; t < k; t++) {
            finalpos[p0 + lperm[p0 + t]] = p0 + t;
            P_out[p0 + t] = P.prow[p0 + lperm[p0 + t]];
        }
    }
    for (int64_t j = 0; j < n; j++) { Q_out[j] = P.qcol[j]; Rs[j] = 1.0 / rinv[P_out[j]]; }
    std::vector<Trip> tl, tu;
    tl.reserve((size_t)P.lnz_bound);
    tu.reserve((size_t)P.unz_bound);
    for (int64_t f = 0; f < P.nfront; f++) {
        const int32_t p0 = P.fr[f].p0, k = P.fr[f].k, m = P.fr[f].m;
        const int32_t *rows = P.rowidx.data() + P.rowptr[f];
        const double *lp = hL.data() + P.px[f], *upn = hU.data() + P.px[f];
        for (int32_t t = 0; t < k; t++) {
            tl.push_back({p0 + t, p0 + t, 1.0});
            for (int32_t i = t + 1; i < m; i++) {
                const double v = lp[i + (int64_t)t * m];
                if (v != 0.0) tl.push_back({i < k ? (int64_t)(p0 + i) : finalpos[rows[i]], p0 + t, v});
            }
            for (int32_t i = t; i < m; i++) {
                const double v = upn[i + (int64_t)t * m];
                if (v != 0.0 || i == t) tu.push_back({p0 + t, i < k ? (int64_t)(p0 + i) : (int64_t)rows[i], v});
            }
        }
    }
    std::vector<int64_t> lp_, li_, up_, ui_;
    std::vector<double> lx_, ux_;
    to_ccs(n, tl, lp_, li_, lx_);
    to_ccs(n, tu, up_, ui_, ux_);
    *lnz = (int64_t)li_.size();
    *unz = (int64_t)ui_.size();
    // F: the off-diagonal blocks, rows in final pivotal order (the in-front interchanges permute them with their rows)
    std::vector<Trip> tf;
    {
        std::vector<double> hF(P.fcol.size());
        if (!hF.empty()) HIPCHK(hipMemcpy(hF.data(), N->d_fval_r, hF.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int64_t pr = 0; pr < n; pr++)
            for (int64_t e = P.fptr_r[pr]; e < P.fptr_r[pr + 1]; e++)
                if (hF[(size_t)e] != 0.0) tf.push_back({finalpos[pr], (int64_t)P.fcol[(size_t)e], hF[(size_t)e]});
    }
    std::vector<int64_t> fp_, fi_;
    std::vector<double> fx_;
    to_ccs(n, tf, fp_, fi_, fx_);
    *fnz = (int64_t)fi_.size();
    *nblocks = (int64_t)P.rblocks.size() - 1;
    *Lp = mdup(lp_); *Li = mdup(li_); *Lx = mdup(lx_);
    *Up = mdup(up_); *Ui = mdup(ui_); *Ux = mdup(ux_);
    *Fp = mdup(fp_); *Fi = mdup(fi_); *Fx = mdup(fx_);
    *r_out = mdup(P.rblocks);
    if (!*Lp || !*Li || !*Lx || !*Up || !*Ui || !*Ux || !*Fp || !*Fi || !*Fx || !*r_out) return KVX_ENOMEM;
    return KVX_OK;
}

int kvx_lu_extract(kvx_lu_num *N, int64_t *lnz, int64_t **Lp, int64_t **Li, double **Lx, int64_t *unz, int64_t **Up,
                   int64_t **Ui, double **Ux, int64_t *fnz, int64_t **Fp, int64_t **Fi, double **Fx, int64_t *P_out,
                   int64_t *Q_out, double *Rs, int64_t *nblocks, int64_t **r_out)
{
    return guarded([&] { return kvx_lu_extract_impl(N, lnz, Lp, Li, Lx, unz, Up, Ui, Ux, fnz, Fp, Fi, Fx, P_out, Q_out, Rs, nblocks, r_out); });
}

// Determinant as the reference computes it (klu.c:760-822): prod(Udiag[k] * Rs[k]) times the signs of P and Q.
static int kvx_lu_det_impl(kvx_lu_num *N, double *det)
{
    if (!N || !det) return KVX_EINVAL;
    if (!N->factored) { set_last_error("singular matrix"); return KVX_ESINGULAR; }
    const LuPlan &P = N->P;
    const int64_t n = N->n;
    int rc = ensure_rhs(N, 1);
    if (rc) return rc;
    launch_lu_udiag(dev_view(N), (int)P.nfront, N->d_X, N->st);
    std::vector<double> ud((size_t)n), rinv((size_t)n);
    std::vector<int32_t> lperm((size_t)n);
    HIPCHK(hipMemcpyAsync(ud.data(), N->d_X, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, N->st));
    HIPCHK(hipMemcpyAsync(rinv.data(), N->d_rinv, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, N->st));
    HIPCHK(hipMemcpyAsync(lperm.data(), N->d_lperm, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, N->st));
    HIPCHK(hipStreamSynchronize(N->st));
    std::vector<int64_t> pf((size_t)n);
    for (int64_t f = 0; f < P.nfront; f++)
        for (int32_t t = 0; t < P.fr[f].k; t++) pf[P.fr[f].p0 + t] = P.prow[P.fr[f].p0 + lperm[P.fr[f].p0 + t]];
    double dd = 1.0;
    for (int64_t k = 0; k < n; k++) dd *= ud[k] / rinv[pf[k]];
    int64_t npiv = 0;
    std::vector<int64_t> w;
    for (int pass = 0; pass < 2; pass++) {
        w = pass ? P.qcol : pf;
        for (int64_t i = 0; i < n; i++)
            while (w[i] != i) { const int64_t t = w[w[i]]; w[w[i]] = w[i]; w[i] = t; npiv++; }
    }
    *det = (npiv & 1) ? -dd : dd;
    return KVX_OK;
}

int kvx_lu_det(kvx_lu_num *N, double *det)
{
    return guarded([&] { return kvx_lu_det_impl(N, det); });
}

}  // extern "C"
