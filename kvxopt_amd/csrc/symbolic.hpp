// Host-side symbolic analysis for the supernodal multifrontal Cholesky factorisation.
//
// Replaces what the reference obtains from cholmod_l_analyze_p (reference call site
// src/C/cholmod.c:274): fill-reducing ordering, elimination tree, postorder, column
// counts, supernodes with relaxed amalgamation -- plus what the GPU schedule needs on
// top: front row structures, child->parent relative indices, A->panel scatter map and
// the elimination-tree level schedule.  Own design; no SuiteSparse code.
#pragma once
#include <cstdint>
#include <utility>
#include <vector>
#include <string>

namespace kvx {

// Kernel class of a front (drives the order inside a level and the launch plan).
//   0            : big     (m > 128 or k > 64) blocked HBM/L2 path, MFMA panel steps
//   1, 2         : LDS     (m <= 128 / 96, k <= 64) one 256-thread workgroup per front
//   3 .. 8       : wave    (m <= 64/48/32, k <= 32/16) one wavefront per front
enum { KVX_CLS_BIG = 0, KVX_CLS_LDS128 = 1, KVX_CLS_LDS96 = 2, KVX_CLS_WAVE0 = 3, KVX_NCLS = 9 };
inline int front_class(int m, int k)
{
    if (m > 128 || k > 64) return KVX_CLS_BIG;
    if (m > 64 || k > 32) return m > 96 ? KVX_CLS_LDS128 : KVX_CLS_LDS96;
    const int mc = m <= 32 ? 2 : (m <= 48 ? 1 : 0);
    return KVX_CLS_WAVE0 + 2 * mc + (k <= 16 ? 1 : 0);
}
inline int wave_class_mcap(int cls) { static const int c[3] = {64, 48, 32}; return c[(cls - KVX_CLS_WAVE0) / 2]; }
inline int wave_class_kmax(int cls) { return ((cls - KVX_CLS_WAVE0) & 1) ? 16 : 32; }

struct SymOpts {
    int ordering = 0;        // 0 = best of nested dissection (n >= nd_min_n) and approximate minimum degree (n <= amd_auto_max), by fill;
                             // 1 = natural; 2 = nested dissection only; 3 = approximate minimum degree only
    int compare_given = 0;   // with a user permutation: 1 = it is one candidate among the library's own (cholmod.options['nmethods']
                             // 0 / 2, cholmod.c:65-76), the ordering with the least fill wins; 0 = it is used as given (nmethods = 1)
    int64_t amd_auto_max = 200000;   // ordering 0: largest order for which the minimum-degree candidate is computed as well
    int64_t nd_min_n = 20000;        // ordering 0: below this order the minimum-degree ordering is taken without computing the dissection as well (on small
                             // systems the dissection rarely wins on fill and costs as much as the rest of the analysis: ACTIVSg2000, n = 4000, 3.0 -> 1.5 ms)
    int postorder = 1;
    int relax_small = 4;
    double relax_z1 = 0.8, relax_z2 = 0.1, relax_z3 = 0.075;   // zero fractions tolerated when a chain supernode joins its parent
                             // (merged width <= 16, <= 48, any); CHOLMOD uses 0.8 / 0.1 / 0.05 -- 0.075 for wide supernodes removes five of the
                             // sixteen levels that hold big fronts on config 2 (each costs an extend-add + a first diagonal block + a join)
    int nd_leaf = 96;        // nested-dissection leaf size
    int leaf_cols = 32;      // leaf-subtree amalgamation: fuse while pivots <= leaf_cols ...
    int leaf_rows = 64;      // ... and front order <= leaf_rows (0 cols = off)
};

struct Symbolic {
    int64_t n = 0;
    int64_t nnzA = 0;                 // entries in the caller's CCS arrays (both triangles)
    int64_t nnzTri = 0;               // entries in the analysed triangle
    std::vector<int64_t> perm, iperm; // final permutation: C = P A P', C(i,j) = A(perm[i],perm[j])
    std::vector<int32_t> colcount;    // nnz(L(:,j)) incl. diagonal, simplicial count
    int64_t lnz = 0;
    double flops = 0;                 // sum_j colcount_j^2

    // supernodes (= fronts), numbered in postorder
    int64_t nsuper = 0;
    std::vector<int64_t> super;       // [nsuper+1] first column of each supernode
    std::vector<int32_t> sn_k;        // pivot columns
    std::vector<int32_t> sn_m;        // front order (rows of the panel)
    std::vector<int64_t> rowptr;      // [nsuper+1] into rowidx
    std::vector<int32_t> rowidx;      // sorted global (permuted) row indices of each front
    std::vector<int32_t> rel;         // parallel to rowidx: position in the PARENT front (rows >= k only)
    std::vector<int64_t> px;          // [nsuper+1] panel offsets in Lx (m*k doubles, column-major, ld = m)
    std::vector<int32_t> sparent;     // supernodal elimination tree (-1 = root)
    std::vector<int32_t> depth;       // distance from the root of its tree
    std::vector<int64_t> childptr;    // [nsuper+1]
    std::vector<int32_t> children;
    std::vector<int64_t> amap;        // [nnzA] destination in Lx of every caller entry, -1 = not read

    // level schedule: level L holds the fronts at depth L; processed nlevels-1 ... 0
    int32_t nlevels = 0;
    std::vector<int64_t> levelptr;    // [nlevels+1]
    std::vector<int32_t> levellist;   // fronts grouped by level
    std::vector<int64_t> ux;          // [nsuper] offset of the u x u update matrix in its parity buffer
    std::vector<int64_t> wx;          // [nsuper] offset of the solve update vector (u doubles)
    int64_t upd_size[2] = {0, 0};     // doubles per parity buffer (even / odd depth)
    int64_t wrk_size[2] = {0, 0};     // solve workspace per parity buffer
    int64_t lsize = 0;                // = px[nsuper]
    int32_t max_m = 0, max_k = 0;
    int64_t sum_m = 0;
};

// Throws std::runtime_error with a message on invalid input. perm may be nullptr.
// uplo: 'L' or 'U' (cholmod.c:132-181: only that triangle is read).
void analyze(int64_t n, const int64_t *colptr, const int64_t *rowind, int uplo,
             const int64_t *perm, const SymOpts &opts, Symbolic &S);

// Fill-reducing ordering of a symmetric graph given as full adjacency without
// diagonal (adjptr[n+1], adj[]).  Returns perm (new -> old).
// closed (optional): the subdomains the dissection finished as a whole, as position ranges [lo, hi) of the ordering -- a vertex of
// such a range has no neighbour at an earlier position outside the range (its other neighbours are separators, numbered later)
void order_nd(int64_t n, const std::vector<int64_t> &adjptr, const std::vector<int32_t> &adj,
              int leaf, std::vector<int64_t> &perm, std::vector<std::pair<int64_t, int64_t>> *closed = nullptr);

// Approximate minimum degree on the quotient graph (amd_order.cpp); same graph format as order_nd.
void order_amd(int64_t n, const std::vector<int64_t> &adjptr, const std::vector<int32_t> &adj, std::vector<int64_t> &perm);

// Sharding of ONE factorisation over nranks processes, host-only and deterministic: every rank computes the same map.
// Proportional mapping of the elimination tree: front s is worked on by the contiguous rank range [glo[s], ghi[s]).  A range of
// one rank owns the front (and then its whole subtree).  A front shared by several ranks is either replicated on them
// (mode 0: small fronts of the top of the tree -- every rank of the range factors it, no communication inside the front) or
// block-cyclic (mode 1: order >= min_m): its columns are dealt out in blocks of `ob` columns round-robin over the range, the
// owner of a pivot block factors the panel and broadcasts it, every rank applies the rank-ob update to its own blocks.
struct DistMap {
    int nranks = 1, ob = 512, min_m = 6144;
    std::vector<int32_t> glo, ghi;
    std::vector<uint8_t> mode;
    std::vector<double> rank_flops;        // factorisation flops executed by each rank (replicated fronts count on every rank)
    std::vector<double> rank_panel_flops;  // of which: panel factorisations of block-cyclic fronts (serial within the front's range)
    double total_flops = 0, replicated_flops = 0;   // sum_j c_j^2 ; flops of the replicated (mode 0, range > 1) fronts, counted once
};
void dist_map(const Symbolic &S, int nranks, int ob, int min_m, DistMap &M);

}  // namespace kvx
