// Host-side analysis for the sparse LU path (kvxopt.klu API; reference call sites
// src/C/klu.c:141 `klu_analyze`, :161 `klu_factor`).
//
// Own design, not KLU's: KLU is a left-looking column algorithm whose row structure is discovered while it
// pivots -- inherently sequential.  Here the structure is STATIC so that a GPU can factor whole levels of an
// elimination tree at once:
//   1. row scaling by max |a_ij| and a maximum-product bipartite matching put large entries on the diagonal
//      (the role of KLU's maximum transversal + scaling, klu_analyze / klu_scale);
//   2. the fill-reducing ordering, elimination tree and supernodes of the SYMMETRISED pattern M + M' come from
//      the Cholesky analysis (symbolic.hpp), every supernode becomes a square front with an L panel and a U panel;
//   3. threshold partial pivoting (KLU's rule: keep the diagonal when |d| >= tol * max, klu.h Common.tol = 0.001)
//      is restricted to the pivot block of a front;
//   4. when no acceptable pivot exists inside a pivot block the numeric phase reports the front and the host
//      MERGES it into its parent front (the parent's pivot rows become candidates: a delayed pivot expressed as a
//      static re-partition) and factors again.  Merges are remembered in the symbolic object.  A failure in a
//      root front means the matrix is numerically singular (klu.c:172-174 ArithmeticError).
#pragma once
#include "symbolic.hpp"
#include <cstdint>
#include <vector>

namespace kvx {

struct LuSymbolic {
    int64_t n = 0, nnz = 0;
    std::vector<int64_t> Ap, Ai;          // the caller's pattern (numeric() must be called with the same one)
    std::vector<int64_t> rowfor;          // matching: row rowfor[j] sits on the diagonal of column j
    bool structurally_singular = false;
    Symbolic S;                           // analysis of pattern(M + M'), M = A(rowfor, :)
    // block triangular form (KLU's BTF, klu_analyze): strongly connected components of M, numbered so that every entry
    // M(r, c) has blk[r] <= blk[c] (block UPPER triangular).  nblocks == 1: not used (one block, F empty).
    int64_t nblocks = 1, nblev = 1;
    std::vector<int32_t> blk;             // [n] block of every M label (row and column share it: the diagonal is matched)
    std::vector<int32_t> blev;            // [nblocks] solve level: 1 + max level of the later blocks this one's rows touch (0: none)
    std::vector<int32_t> uf;              // union-find over S's supernodes: learned merges (uf[s] = representative link)
    int64_t nmerges = 0;
};

// One front of the LU plan (positions = indices in the final pivotal order).
struct LuFrontH {
    int32_t k, m, p0, nchild;
    int32_t parent;
    int32_t depth;
};

struct LuPlan {
    int64_t n = 0, nfront = 0;
    std::vector<int64_t> qcol;            // [n] column of A at position j
    std::vector<int64_t> prow;            // [n] row of A at position i before in-front pivoting
    std::vector<LuFrontH> fr;
    std::vector<int32_t> group_of;        // [S.nsuper] front index of every base supernode
    std::vector<int64_t> rowptr;          // [nfront+1] into rowidx
    std::vector<int32_t> rowidx;          // front rows as positions: k pivot positions then sorted update rows
    std::vector<int32_t> rel;             // parallel to rowidx: index in the PARENT front (update rows only)
    std::vector<int64_t> px;              // [nfront+1] panel offsets (m*k doubles) -- same for the L and U' panels
    std::vector<int64_t> upd_off;         // [nfront] offset of the update matrix / of the whole front (big) in the arena
    std::vector<int32_t> upd_ld;          // [nfront] leading dimension of the update matrix there
    std::vector<int64_t> wx;              // [nfront] offset of the solve update vector (u doubles)
    std::vector<int64_t> childptr;        // [nfront+1]
    std::vector<int32_t> children;
    std::vector<int64_t> aptr;            // [nfront+1] into a_src / a_dst
    std::vector<int64_t> a_src;           // index of the entry in the caller's value array
    std::vector<int32_t> a_dst;           // r + c*m inside the front
    int32_t nlevels = 0;
    std::vector<int64_t> levelptr;        // level L = fronts at depth L (roots: 0)
    std::vector<int32_t> levellist;       // inside a level: LDS-resident fronts first (largest first), then the big ones (smallest first)
    std::vector<int32_t> nlds;            // [nlevels] how many fronts of the level are LDS resident (m <= lds_m)
    std::vector<int32_t> nsbig;           // [nlevels] how many fronts (the LAST of the level's list) take the multi-workgroup solve path
    // triangular solves: stages = (block level, tree depth) pairs; without BTF one block level, stages = tree levels
    int32_t nblev = 1, nstage = 0;
    std::vector<int32_t> levstage;        // [nblev+1] stage range of every block level (stage = levstage[l] + depth)
    std::vector<int64_t> stageptr;        // [nstage+1] into stagelist
    std::vector<int32_t> stagelist;       // fronts of a stage: one-workgroup fronts first, then the multi-workgroup ones (m ascending)
    std::vector<int32_t> stage_nbig, stage_smallm, stage_smallk, stage_bigm, stage_bigk;   // [nstage]
    // F = off-diagonal blocks of R P A Q (never eliminated, klu_extract's F): by row position (CSR, for A x = b) and by
    // column position (CSC, for A' x = b); fsrc = index of the entry in the caller's value array
    std::vector<int64_t> fptr_r, fptr_c;  // [n+1]
    std::vector<int32_t> fcol, frow;      // column position of a CSR entry / row position of a CSC entry
    std::vector<int64_t> fsrc_r, fsrc_c;
    std::vector<int64_t> flevptr;         // [nblev+1] into flevpos
    std::vector<int32_t> flevpos;         // positions of the block level (all of them: rows and columns share positions)
    std::vector<int64_t> rblocks;         // [nblocks+1] block boundaries in pivotal positions (klu_extract's r)
    int64_t arena = 0, wsize = 0, lsize = 0;
    int32_t max_m = 0, max_k = 0;
    int64_t lnz_bound = 0, unz_bound = 0;
};

constexpr int KVX_LU_SOLVE_BIG_M = 384;   // fronts of order > this are swept by many workgroups (one launch per 32 pivots); their solve
                                          // work vector holds all m entries (the update part at offset k)
constexpr int KVX_LU_LDS_M = 112;        // fronts of order <= this are assembled in LDS and eliminated in registers (7 x 7 tile per thread)

// values may be nullptr (pattern-only: plain maximum transversal).  Throws std::runtime_error on invalid input.
void lu_analyze(int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax, LuSymbolic &Y);

// Builds the plan for the current merge state of Y.
void lu_build_plan(const LuSymbolic &Y, LuPlan &P);

// Merge the base-supernode groups whose fronts are listed (front indices of `P`) into their parents.
// Returns false when one of them is a root (nothing to merge into: singular).
bool lu_merge_fronts(LuSymbolic &Y, const LuPlan &P, const std::vector<int32_t> &fronts);

// Maximum-product matching (rows scaled by rinv).  Returns the number of matched columns.
int64_t lu_matching(int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax, const double *rinv,
                    std::vector<int64_t> &rowfor);

}  // namespace kvx
