// Internal (library-private) view of the Cholesky factor object shared by api.cpp and dist_api.cpp.
#pragma once
#include "../../include/kvxhip.h"
#include "abi_guard.hpp"
#include "devpool.hpp"
#include "device.hpp"
#include "symbolic.hpp"

#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <atomic>
#include <future>
#include <string>
#include <vector>

namespace kvx { void set_last_error(const std::string &s); struct DistState; }
#define set_err(s) ::kvx::set_last_error(s)
using namespace kvx;

#define HIPCHK(call)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            set_err(std::string(#call) + ": " + hipGetErrorString(e_));                 \
            return KVX_EDEVICE;                                                          \
        }                                                                                \
    } while (0)

// An executable graph whose INSTANTIATION runs on a host thread of its own: hipGraphInstantiate of a ~300-node capture takes 10-12 ms
// under ROCm 7.2's runtime (1-2 ms under 7.0.51831) -- as much as a dozen replays save -- while the capture itself is ~1 ms.  The
// calls that arrive before the executable is ready run their launches eagerly, as the first call does; results do not depend on
// which way a call ran.  KVX_GRAPH_SYNC_INSTANTIATE=1: instantiate in the calling thread (as before).
// instantiations in flight are joined before the HIP runtime is torn down at exit (a cached handle -- cholmod._SYMBOLIC_CACHE,
// lp._KKT_CACHE -- may still be alive then); failures are counted (round-3 advisor finding)
std::atomic<int> &lazy_exec_failures();
void lazy_exec_track(const std::shared_future<hipGraphExec_t> &f);

struct LazyExec {
    hipGraphExec_t exec = nullptr;
    std::shared_future<hipGraphExec_t> fut;
    bool pending = false;
    bool tried = false;                          // a capture was taken (or failed): no second attempt
    hipGraphExec_t ready()
    {
        if (pending && fut.wait_for(std::chrono::seconds(0)) == std::future_status::ready) { exec = fut.get(); pending = false; }
        return pending ? nullptr : exec;
    }
    void start(hipGraph_t graph)
    {
        tried = true;
        int dev = 0;
        (void)hipGetDevice(&dev);
        auto work = [graph, dev]() -> hipGraphExec_t {
            (void)hipSetDevice(dev);
            hipGraphExec_t e = nullptr;
            if (hipGraphInstantiate(&e, graph, nullptr, nullptr, 0) != hipSuccess) {
                (void)hipGetLastError();
                e = nullptr;
                lazy_exec_failures()++;              // (the handle then runs eagerly for good: kvx_graph_instantiate_failures says so)
            }
            (void)hipGraphDestroy(graph);
            return e;
        };
        // (KVX_GRAPH_SYNC_INSTANTIATE=1: on the calling thread.  Measured in round 3 against the suspicion that the thread of its own
        // changes which hardware queues the executable's branches share: 8 alternating processes each way on one box, step
        // 4.57-4.73 ms on the calling thread, 4.52-4.74 on its own -- the process-to-process spread is there either way.)
        const char *sy = getenv("KVX_GRAPH_SYNC_INSTANTIATE");
        if (sy && sy[0] == '1') { exec = work(); return; }
        fut = std::async(std::launch::async, work).share();
        pending = true;
        lazy_exec_track(fut);                        // joined at process exit if the handle is still alive then (api.cpp)
    }
    void drop()
    {
        if (pending) { exec = fut.get(); pending = false; }
        if (exec) (void)hipGraphExecDestroy(exec);
        exec = nullptr;
        tried = false;
    }
};

struct LevelPlan {
    // fronts of the level grouped by kernel class (symbolic.hpp front_class), big first
    int64_t off[KVX_NCLS];   // offset into d_lists of class c
    int cnt[KVX_NCLS];
    int maxm[KVX_NCLS];
    int maxk[KVX_NCLS];
    int big_maxk = 0;          // largest pivot count among the big fronts (solves)
    int chain_maxk = 0;        // ... among those factored by the batched multi-workgroup chain (all of them)
    int big_maxu = 0;          // largest update matrix (m - k) among the big fronts
    double big_flops = 0.0;    // flops of the big fronts of the level
    int64_t big_u_len = 0;   // doubles of the parity buffer used by the big fronts (head)
    // solve groups: [big], [LDS classes: 256 threads], [wave classes: 64 threads]
    int64_t soff[3];
    int scnt[3];
    int smaxm[3];
};

struct kvx_chol {
    Symbolic S;
    kvx_chol_opts opts;
    bool dev_ready = false;
    bool numeric = false;
    bool pending = false;     // a factorisation was enqueued and its status not yet read
    bool is_ll = true;        // false: the factor is presented as P A P' = L D L' (options['supernodal'] = 0, or 1 on a sparse factor)
    double *d_diag = nullptr; // diag(Lc) for the LDL' view, extracted after every factorisation
    bool diag_valid = false;
    int64_t minor = 0;
    hipStream_t stream = nullptr;
    hipStream_t side[4] = {nullptr, nullptr, nullptr, nullptr};   // [3]: trailing updates beside the pivot chain   // independent kernel classes of one level run concurrently
    hipEvent_t ev_fork = nullptr, ev_fork2 = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_out = nullptr;                // orders the caller's (null-stream) work after an asynchronous solve
    hipEvent_t ev_in = nullptr;                 // orders the factor's stream after the caller's (null-stream) work
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // factor + solve in one call (kvx_chol_factorize_solve_dev): the forward sweep follows the factorisation level by level on
    // the stream of the small-front launches (side[0]) -- a level's fronts are swept while the levels above it are still being factored
    std::vector<hipEvent_t> ev_lvl;             // [nlevels]: recorded on the factor's stream when a level is complete (while pipe_on)
    bool pipe_on = false;
    int pipe_nr = 0;                            // right-hand sides of the sweep that follows the factorisation in flight
    int pipe_from = 0;                          // the sweep starts when this level is factored (everything below it in one go), then follows level by level
    hipEvent_t ev_pipe[4] = {nullptr, nullptr, nullptr, nullptr};   // fork / join of the forward sweep's streams
    SubDesc *d_subs_lvl = nullptr;              // the leaf subtrees once more, grouped by the level of their ROOT front: the pipelined sweep
    std::vector<int> sub_lvl_off, sub_lvl_cnt;  // walks the subtrees rooted at level l when level l is factored ([nlevels] offsets / counts)
    struct FusedGraph { int nrhs; double *B; int64_t ldB; int calls; LazyExec exec; };
    std::vector<FusedGraph> g_fused;
    bool have_ftime = false, have_stime = false;
    double ms_factor = 0, ms_solve = 0;

    int32_t *d_k = nullptr, *d_m = nullptr, *d_first = nullptr, *d_rowidx = nullptr, *d_rel = nullptr,
            *d_children = nullptr, *d_perm = nullptr, *d_lists = nullptr;
    int64_t *d_px = nullptr, *d_rowptr = nullptr, *d_ux = nullptr, *d_wx = nullptr, *d_childptr = nullptr,
            *d_amap = nullptr, *d_sdst = nullptr, *d_scptr = nullptr;   // d_sdst / d_ssrc / d_scptr: the scatter map grouped by chunk of L (k_init_factor)
    double *d_Lx = nullptr, *d_U[2] = {nullptr, nullptr}, *d_Ax = nullptr;
    double *d_X = nullptr, *d_X0 = nullptr, *d_W[2] = {nullptr, nullptr}, *d_WK = nullptr;   // d_X0: untouched copy of the rhs for the forward sweep
    double *d_Linv = nullptr;
    int32_t *d_ssrc = nullptr;
    int64_t scnt = 0;
    int64_t *d_linv_off = nullptr;
    FrontDesc *d_fd = nullptr;
    ChildDesc *d_cd = nullptr;
    int32_t *d_tiles = nullptr;
    int64_t x_cap = 0;        // right-hand sides the solve workspace holds
    int *d_status = nullptr;
    int *h_status = nullptr;  // pinned
    int *h_status_dev = nullptr;   // the device's address of h_status (hipHostGetDevicePointer); null: copied by a memcpy instead
    DevSym ds{};
    // sharded mode (kvx_chol_dist_*, dist_api.cpp): the level lists hold only the fronts this rank takes part in; the
    // factorisation additionally leaves out the block-cyclic fronts (factored by dist_api.cpp's panel loop): fplan / d_flists
    int dist_rank = 0, dist_nranks = 1;
    std::vector<uint8_t> part;                 // sharded mode with a per-rank layout (dist_api.cpp trim_to_rank): 1 = this rank holds the front; empty = all fronts
    int64_t lsize_total = -1;                  // panel doubles of the WHOLE factor when S.lsize was trimmed to this rank's fronts (-1: S.lsize is the whole)
    int64_t dev_bytes = 0;                     // bytes of the factor's large device buffers (panels, inverted blocks, update matrices, values)
    kvx::DistState *dist = nullptr;
    bool fplan_on = false;
    std::vector<LevelPlan> fplan;
    std::vector<int32_t> flists_host;              // host copy of d_flists
    // Per (level, panel step) the big fronts still in the chain, largest trailing matrix first, and per (level, block of
    // u_block pivot columns) those whose update matrix the block updates, largest update matrix first: the LDS-staged
    // trailing update numbers its workgroups over size classes of these lists (api.cpp build_chain_lists, device.hpp TileClasses)
    struct ChainList { int64_t off; int cnt; };
    std::vector<std::vector<ChainList>> chain_steps, u_steps;     // [level][jb / 64], [level][kb / u_block]
    std::vector<int32_t> chain_host, chain_m, chain_k;            // the lists concatenated; order and pivot count of every entry
    int32_t *d_chain = nullptr;
    std::vector<hipEvent_t> ev_u;                  // per block of a level's chain: its panels are solved (the deferred updates' stream waits)
    hipEvent_t ev_ujoin = nullptr;
    int last_fused_path = 0;                       // kvx_chol_last_fused_path
    int u_block = 384;                             // pivot columns per pass of the deferred update (KVX_U_BLOCK)
    int32_t *d_flists = nullptr;
    std::vector<int64_t> linv_off_host;        // per front: offset of its inverted diagonal blocks in d_Linv (-1: not a big front)
    uint8_t *d_keep = nullptr;                 // per permuted column: 1 = this rank reports the entry of x
    std::vector<int32_t> lists_host;           // level lists in use (filtered in sharded mode)
    std::vector<int64_t> lptr_host;
    int outer_block = 1024;    // columns per outer block of the two-level update (KVX_OUTER_BLOCK; a multiple of 64): config 5 runs at 33.3 / 37.5 / 38.6 / 37.4 TF/s with 256 / 512 / 1024 / 2048
    int two_level_m = 6144;    // levels whose largest front is at least this order use the two-level blocked update (KVX_TWO_LEVEL_M)
    // leaf subtrees walked by one wavefront each in the solves (build_subtrees)
    SubDesc *d_subs = nullptr;
    SubDesc *d_subs_f = nullptr;               // the same subtrees grouped by LDS image size (32 / 48 / 64 rows) for the factorisation
    int nsubf[3] = {0, 0, 0};
    bool factor_subtrees = false;              // opt-in KVX_FACTOR_SUBTREES=1: the subtrees are factored by one wavefront each before the level loop (measured slower)
    int32_t *d_cd_woff = nullptr, *d_depth = nullptr, *d_lists_sw = nullptr;
    int nsub = 0, nsub32 = 0, nsub48 = 0;      // subtrees; the first nsub32 hold only fronts of order <= 32, the next nsub48 - nsub32 of order <= 48
    bool use_subtrees = true;
    std::vector<SubDesc> subs_host;
    std::vector<int32_t> cd_woff_host;
    std::vector<uint8_t> in_sub;
    std::vector<int32_t> lsw_host;             // host copy of d_lists_sw
    std::vector<int32_t> col2sn, sub_of;       // sparse right-hand sides (spsolve): front of a permuted column, subtree of a front (-1: none)
    std::vector<int64_t> sw_off;               // per level: the wave-class fronts NOT in a subtree (offset, count into d_lists_sw)
    std::vector<int> sw_cnt, sw_kmax;
    // many right-hand sides (kernels_wide.hip): per front row, the children's update rows that land on it (built at the first such solve)
    int32_t *d_inv_ptr = nullptr, *d_inv_src = nullptr, *d_iperm = nullptr;   // d_iperm: position of every caller row in the permuted order
    int wide_state = 0;                        // 0 = not built yet, 1 = ready, -1 = not available for this factor (sharded mode, index range)
    int wide_from = -1;                        // right-hand sides from which the rhs-major path is used (KVX_WIDE_FROM; 0 = never; -1 = by size, see solve_dev)
    bool solve_merged = false;                 // sw lists hold every small front outside the subtrees (one launch per level)
    int side_spread = 1;      // spread the small-front launches of a level over the streams (KVX_SIDE_SPREAD=0: one stream)
    std::vector<LevelPlan> plan;
    // hipGraph replay of the (static) launch sequences: captured on the second call, replayed after.
    // Disabled while a kernel family is being event-timed and by KVX_NO_GRAPH=1.
    bool use_graph = true;
    int factor_calls = 0;
    LazyExec g_factor;
    struct SolveGraph { int kind; int nrhs; int calls; LazyExec exec; };
    std::vector<SolveGraph> g_solve;
    // optional per-kernel-family timing (bench.py roofline leg): HIP events around every launch
    // of ONE selected family on the factor's stream
    int prof_family = -1;
    std::vector<hipEvent_t> prof_ev;
    size_t prof_used = 0;
    double prof_ms = 0;
    int64_t prof_launches = 0;
};

namespace kvx {

template <class T>
int upload(T **dst, const std::vector<T> &src)
{
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    HIPCHK(pool_malloc((void **)dst, bytes));
    if (!src.empty()) HIPCHK(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return KVX_OK;
}

// optional per-kernel-family timing: HIP events around every launch of the selected family, on the stream it is launched on
struct ProfScope {
    kvx_chol *F;
    bool on;
    hipStream_t st;
    ProfScope(kvx_chol *F_, int fam, hipStream_t st_ = nullptr) : F(F_), on(F_->prof_family == fam), st(st_ ? st_ : F_->stream)
    {
        if (!on) return;
        if (F->prof_used + 2 > F->prof_ev.size()) {
            size_t old = F->prof_ev.size();
            F->prof_ev.resize(old + 256, nullptr);
            for (size_t i = old; i < F->prof_ev.size(); i++) (void)pool_event_get(&F->prof_ev[i], true);
        }
        (void)hipEventRecord(F->prof_ev[F->prof_used++], st);
    }
    ~ProfScope()
    {
        if (on) (void)hipEventRecord(F->prof_ev[F->prof_used++], st);
    }
};
enum { FAM_SCATTER = 0, FAM_SMALL = 1, FAM_ASSEMBLE = 2, FAM_POTRF = 3, FAM_TRSM = 4, FAM_SYRK = 5, FAM_FWD = 6, FAM_BWD = 7 };

inline void prof_collect(kvx_chol *F)
{
    for (size_t i = 0; i + 1 < F->prof_used; i += 2) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, F->prof_ev[i], F->prof_ev[i + 1]) == hipSuccess) { F->prof_ms += ms; F->prof_launches++; }
    }
    F->prof_used = 0;
}


// api.cpp
void build_plan_from(const Symbolic &S, const std::vector<int32_t> &lists, const std::vector<int64_t> &lptr, std::vector<LevelPlan> &plan);
void build_plan(kvx_chol *F);
int build_subtrees(kvx_chol *F);
int build_chain_lists(kvx_chol *F);
int ensure_device(kvx_chol *F);
int ensure_solve_ws(kvx_chol *F, int64_t nrhs);
int wait_for_caller(kvx_chol *F);
int enqueue_factor_body(kvx_chol *F, int lfrom = -1, int lto = 0, bool prologue = true, bool epilogue = true);
int finish_factor(kvx_chol *F, int64_t *minor);
// the streams and events a triangular sweep forks its kernel classes over (default: the factor's own main / side[0] / side[1])
struct SweepStreams { hipStream_t main, lds, wave; hipEvent_t fork, join0, join1; };
// wait_levels: the sweep follows a factorisation in flight -- each level waits for that level's completion event (ev_lvl)
// sub_tail (with wait_levels): the subtrees rooted at level lto or deeper are walked by ONE launch in front of the levels (instead of
// one launch per level for the subtrees rooted there)
void enqueue_fwd(kvx_chol *F, double *X, int64_t ldx, int nrhs, int lfrom = -1, int lto = 0, const SweepStreams *ss = nullptr, bool wait_levels = false,
                 bool sub_tail = false);
void enqueue_bwd(kvx_chol *F, double *X, int64_t ldx, int nrhs, int lfrom = 0, int lto = -1);
void destroy_graphs(kvx_chol *F);
// dist_api.cpp
void dist_release(kvx_chol *F);

}  // namespace kvx
