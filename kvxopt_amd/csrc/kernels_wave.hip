// Small fronts (order m <= 128): the bulk of nnz(L) on mesh problems lives here.
//
//   k_front_wave<KMAX> : m <= 64, k <= 32.  One 64-lane wavefront owns one front, no workgroup
//                        barrier anywhere.
//   k_front_lds<KMAX>  : m <= 128, k <= 64.  One 256-thread workgroup per front: wave (h, p) holds
//                        rows 64 h.. and the columns of parity p; one barrier per pivot column.
//
// Shared design:
//   * lane r holds ROW r of the panel in registers a[0..KMAX) -- column j of L is then one
//     register across the wave; its pivot travels by v_readlane and its multipliers by an LDS
//     broadcast (kvx_col_step, device.hpp), no barrier per column; an LDS front splits rows and
//     column parities over its four waves (quad_col_step) and pays one barrier per column;
//   * the children's update matrices are extend-added into an LDS image of the front with all
//     HBM loads of a batch in flight at once (these kernels are latency-bound on small levels);
//   * the Schur complement U = F22 - L21 L21' is FP64 MFMA (v_mfma_f64_16x16x4_f64) from the LDS
//     image, streamed to HBM; the panel is read and written exactly once, coalesced.
// HBM-bound by design: algorithmic bytes = 16 m k + 8 u(u+1)/2 + the children's triangles.
//
// Reference role: cholmod_l_factorize (src/C/cholmod.c:362) for these supernodes.
#include "device.hpp"

#include <cstdlib>
#include <utility>

namespace kvx {

typedef double d4 __attribute__((ext_vector_type(4)));

// Column step J of an LDS front (m <= 128, k <= 64) spread over the four waves of the workgroup:
// wave (h, p) holds rows 64 h .. 64 h + 63 and the columns of parity p (column c = 2 t + p in a[t]).
// The owner parity publishes the UNSCALED column J (all 128 rows) into the J-parity half of cb2, one
// workgroup barrier, then every wave reads the pivot and its multipliers by LDS broadcast and updates
// its own columns; double buffering makes the one barrier per column sufficient.
template <int KMAX, int J>
__device__ __forceinline__ void quad_col_step(double (&a)[KMAX / 2], int k, int row, int p, int *status, int col0,
                                              double *cb2, const PivRule pr)
{
    if (J < k) {                                   // workgroup-uniform
        double *cb = cb2 + (J & 1) * 128;
        const bool owner = p == (J & 1);           // wave-uniform
        if (owner) cb[row] = a[J >> 1];
        __syncthreads();
        const double aj = cb[row];
        double d = cb[J];
        if (!(d > pr.floor)) {
            if (pr.flag_all) {
                if (row == 0 && p == 0) atomicMin(status, col0 + J);
            }
            d = pr.sub;
        }
        double ljj, inv;
        kvx_sqrt_rsqrt(d, ljj, inv);
        const double w = (row > J) ? aj * (inv * inv) : 0.0;
        const double *cbp = cb + p;
        if ((J & 1) == 0) {                        // column J + 1 = 2 (J/2) + 1 belongs to parity 1 only
            const double w1 = p ? w : 0.0;
            a[J >> 1] = owner ? a[J >> 1] : __builtin_fma(-w1, cb[J + 1], a[J >> 1]);
        }
#pragma unroll
        for (int t = (J >> 1) + 1; t < KMAX / 2; t++) a[t] = __builtin_fma(-w, cbp[2 * t], a[t]);
        if (owner) a[J >> 1] = (row == J) ? ljj : (row > J ? aj * inv : 0.0);
    }
}
template <int KMAX, int... Js>
__device__ __forceinline__ void quad_col_steps(double (&a)[KMAX / 2], int k, int row, int p, int *status, int col0,
                                               double *cb2, const PivRule pr, std::integer_sequence<int, Js...>)
{
    (quad_col_step<KMAX, Js>(a, k, row, p, status, col0, cb2, pr), ...);
}

// The LDS image of a front holds its LOWER TRIANGLE only, packed by columns: entry (row, col), row >= col, of a front of
// order m sits at pk(row, col, m).  Half the footprint of a square image: two fronts of order 128 (66 KB each) share a CU
// instead of one, nine wave fronts of order 64 instead of four -- these kernels are bound by occupancy x per-front latency.
__device__ __forceinline__ int pk(int row, int col, int m) { return row + col * (m - 1) - ((col * (col - 1)) >> 1); }

// Extend-add of one child's update matrix (lower triangle, uc x uc, ld = uc) into the LDS image F.
// Thread (i = tid % RP, ph = tid / RP) owns child row i and the columns j = ph (mod NP); the HBM
// loads of a batch of B columns are issued before any LDS update so that B loads are in flight.
template <int NT, int RP>
__device__ __forceinline__ void extend_add_child(double *F, int m, const int *relsh, const double *U, int uc, int tid)
{
    constexpr int NP = NT / RP, B = 16;
    const int i = tid % RP, ph = tid / RP;
    const bool row_ok = i < uc;
    const int myrow = row_ok ? relsh[i] : 0;
    for (int jb = 0; jb < uc; jb += NP * B) {
        double v[B];
#pragma unroll
        for (int q = 0; q < B; q++) {
            const int j = jb + ph + q * NP;
            v[q] = kvx_ld0(U, i + (int64_t)j * uc, row_ok && j <= i);
        }
#pragma unroll
        for (int q = 0; q < B; q++) {
            const int j = jb + ph + q * NP;
            if (row_ok && j <= i) F[pk(myrow, relsh[j], m)] += v[q];
        }
    }
}

// All children of a front: the descriptor and the relative indices of child c + 1 are fetched while
// child c is added, so each child costs one HBM round trip (its update matrix) instead of three.
template <int NT, int RP>
__device__ __forceinline__ void extend_add_children(const DevSym &ds, const FrontDesc &fd, double *F, int m, int *relsh,
                                                    const double *Uc, int tid)
{
    ChildDesc cd = ds.cd[fd.childptr];
    int myrel = tid < cd.uc ? ds.rel[cd.rel + tid] : 0;
    for (int c = 0; c < fd.nchild; c++) {
        const bool more = c + 1 < fd.nchild;
        ChildDesc nx = cd;
        if (more) nx = ds.cd[fd.childptr + c + 1];
        __syncthreads();                           // (single-wave workgroups: orders the LDS traffic)
        if (tid < RP) relsh[tid] = myrel;
        __syncthreads();
        const int nrel = (more && tid < nx.uc) ? ds.rel[nx.rel + tid] : 0;
        extend_add_child<NT, RP>(F, m, relsh, Uc + cd.ux, cd.uc, tid);
        cd = nx;
        myrel = nrel;
    }
    __syncthreads();
}

// Schur complement tile (ti, tj) of U = F22 - X X', X = rows k.. of the panel in the LDS image
__device__ __forceinline__ void schur_tile(const double *F, int m, int k, int u, int ti, int tj, bool kids,
                                           double *Uout, int lane)
{
    const int lr = lane & 15, lk = lane >> 4;
    const int rr = 16 * ti + lr, cc = 16 * tj + lr;
    d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
    for (int ks = 0; ks < k; ks += 4) {
        const int kc = ks + lk;
        const bool kin = kc < k;
        const double av = kvx_ld0(F, pk(k + cc, kc, m), kin && cc < u);
        const double bv = kvx_ld0(F, pk(k + rr, kc, m), kin && rr < u);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
    // lane holds D[i = lk + 4q][j = lr]: i <-> tile column, j <-> tile row
    if (rr < u) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c = 16 * tj + lk + 4 * q;
            if (c <= rr) {
                const double base = kvx_ld0(F, pk(k + rr, k + c, m), kids);
                Uout[rr + (int64_t)c * u] = base - acc[q];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// One wave-class front (m <= 64, k <= 32) by one wavefront: the body shared by the one-front-per-wavefront kernel of the level
// schedule and by the leaf-subtree walk.  F: packed LDS image of the front, cb2: 2 x 64 column buffer, relsh: 64 ints.
template <int KMAX>
__device__ __forceinline__ void front_wave_body(const DevSym &ds, const FrontDesc &fd, double *__restrict__ Lx,
                                                const double *__restrict__ Uc, double *__restrict__ Uo, int *status,
                                                double *F, double *cb2, int *relsh, const int r)
{
    const int k = fd.k, m = fd.m, u = m - k;
    double *P = Lx + fd.px;
    const bool kids = fd.nchild > 0;
    // the panel goes straight from HBM to registers (all KMAX loads in flight while the children are
    // assembled); the LDS image starts at zero, collects the children and is added on top
    double a[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; j++) a[j] = kvx_ld0(P, r + (int64_t)j * m, j < k && r < m);
    if (kids) {
        const int mm = m * (m + 1) / 2;
        for (int i = r; i < mm; i += 64) F[i] = 0.0;
        extend_add_children<64, 64>(ds, fd, F, m, relsh, Uc, r);
#pragma unroll
        for (int j = 0; j < KMAX; j++) a[j] += kvx_ld0(F, pk(r, j, m), j < k && r < m && r >= j);
    }
    kvx_col_steps<KMAX>(a, k, r, status, fd.first, nullptr, cb2, make_piv_rule(ds), std::make_integer_sequence<int, KMAX>());
    if (r < m) {
#pragma unroll
        for (int j = 0; j < KMAX; j++)
            if (j < k) {
                P[r + (int64_t)j * m] = a[j];
                if (r >= j) F[pk(r, j, m)] = a[j];
            }
    }
    if (u == 0) return;
    __syncthreads();
    double *Uout = Uo + fd.ux;
    const int T = (u + 15) >> 4;
    for (int ti = 0; ti < T; ti++)
        for (int tj = 0; tj <= ti; tj++) schur_tile(F, m, k, u, ti, tj, kids, Uout, r);
}

// OCC: wavefronts per SIMD the register allocation must leave room for.  The levels that hold thousands of these fronts are bound
// by (fronts in flight) x (latency of one front); left alone the compiler takes 129 + 8 registers for KMAX = 32 -- three waves per
// SIMD -- where 128 (four waves, no spill) serve as well, and 96 + 8 for KMAX = 16 (four waves; 95: five).
template <int KMAX, int OCC>
__global__ __launch_bounds__(64, OCC) void k_front_wave(DevSym ds, const int32_t *__restrict__ list,
                                                        double *__restrict__ Lx, const double *__restrict__ Uc,
                                                        double *__restrict__ Uo, int *status, int mcap)
{
    extern __shared__ double F[];                  // packed lower triangle of the front (pk), 2x64 column buffer, 64 ints
    double *cb2 = F + mcap * (mcap + 1) / 2;
    int *relsh = (int *)(cb2 + 128);
    const FrontDesc fd = ds.fd[list[blockIdx.x]];
    front_wave_body<KMAX>(ds, fd, Lx, Uc, Uo, status, F, cb2, relsh, (int)threadIdx.x);
}

// Leaf subtrees in the FACTORISATION: the bottom of the elimination tree is thousands of small independent subtrees whose fronts
// are all wave-class.  Level by level they cost one launch + one stream join per level for a few microseconds of work per front
// (config 2: eight levels, 0.7 ms, before the first big front starts).  Here ONE wavefront factors a whole subtree, its fronts in
// postorder, before the level loop starts: the next front's descriptor is fetched while the current one is factored, and a parent
// finds its children's update matrices where the level schedule would have put them -- every front of a subtree owns a slot of
// the parity buffers that no other front reuses (analyze_subtrees), so subtrees at different depths cannot collide.  Same
// arithmetic per front as k_front_wave (the same body): bitwise the same factor.
template <int KMAX, int OCC>
__global__ __launch_bounds__(64, OCC) void k_factor_subtree(DevSym ds, const SubDesc *__restrict__ subs, const int32_t *__restrict__ depth,
                                                            double *__restrict__ Lx, double *__restrict__ U0, double *__restrict__ U1,
                                                            int *status, int mcap)
{
    extern __shared__ double F[];
    double *cb2 = F + mcap * (mcap + 1) / 2;
    int *relsh = (int *)(cb2 + 128);
    const SubDesc sd = subs[blockIdx.x];
    FrontDesc nxt = ds.fd[sd.lo];
    int ndep = depth[sd.lo];
    for (int s = sd.lo; s <= sd.hi; s++) {
        const FrontDesc fd = nxt;
        const int dp = ndep;
        if (s < sd.hi) { nxt = ds.fd[s + 1]; ndep = depth[s + 1]; }
        double *Uo = (dp & 1) ? U1 : U0;
        const double *Uc = (dp & 1) ? U0 : U1;
        front_wave_body<KMAX>(ds, fd, Lx, Uc, Uo, status, F, cb2, relsh, (int)threadIdx.x);
        __syncthreads();                           // the LDS image is reused; the update matrix just stored is a child's of a later front
    }
}

// ------------------------------------------------------------------------------------------
template <int KMAX>
__global__ __launch_bounds__(256) void k_front_lds(DevSym ds, const int32_t *__restrict__ list,
                                                   double *__restrict__ Lx, const double *__restrict__ Uc,
                                                   double *__restrict__ Uo, int *status, int mcap)
{
    extern __shared__ double F[];                  // packed lower triangle (pk), 2x128 column buffer, 128 ints
    double *cb2 = F + mcap * (mcap + 1) / 2;
    int *relsh = (int *)(cb2 + 256);
    const FrontDesc fd = ds.fd[list[blockIdx.x]];
    const int k = fd.k, m = fd.m, u = m - k, tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
    double *P = Lx + fd.px;
    const bool kids = fd.nchild > 0;
    const int p = wv >> 1, row = (wv & 1) * 64 + ln;    // column parity, row of this thread
    double a[KMAX / 2];
#pragma unroll
    for (int t = 0; t < KMAX / 2; t++) a[t] = kvx_ld0(P, row + (int64_t)(2 * t + p) * m, 2 * t + p < k && row < m);
    if (kids) {
        const int mm = m * (m + 1) / 2;
        for (int i = tid; i < mm; i += 256) F[i] = 0.0;
        extend_add_children<256, 128>(ds, fd, F, m, relsh, Uc, tid);
#pragma unroll
        for (int t = 0; t < KMAX / 2; t++) a[t] += kvx_ld0(F, pk(row, 2 * t + p, m), 2 * t + p < k && row < m && row >= 2 * t + p);
    }
    quad_col_steps<KMAX>(a, k, row, p, status, fd.first, cb2, make_piv_rule(ds), std::make_integer_sequence<int, KMAX>());
    if (row < m) {
#pragma unroll
        for (int t = 0; t < KMAX / 2; t++)
            if (2 * t + p < k) { if (row >= 2 * t + p) F[pk(row, 2 * t + p, m)] = a[t]; P[row + (int64_t)(2 * t + p) * m] = a[t]; }
    }
    if (u == 0) return;
    __syncthreads();
    double *Uout = Uo + fd.ux;
    const int T = (u + 15) >> 4;
    int t = 0;
    for (int ti = 0; ti < T; ti++)
        for (int tj = 0; tj <= ti; tj++, t++)
            if ((t & 3) == wv) schur_tile(F, m, k, u, ti, tj, kids, Uout, ln);
}

// ------------------------------------------------------------------------------------------
// Triangular solves for the wave classes (m <= 64, k <= 32): one wavefront per front and rhs.
// Forward: lane r = row r with its panel row in registers (the panel is read once, coalesced);
// the running right-hand side lives one value per lane and the pivot value travels by readlane.
template <int KMAX>
__global__ __launch_bounds__(64) void k_fwd_wave(DevSym ds, const int32_t *__restrict__ list,
                                                 const double *__restrict__ Lx, double *__restrict__ X, int64_t ldx,
                                                 const double *__restrict__ Wc, double *__restrict__ Wo, int64_t wstride)
{
    unsigned fi, rh;
    kvx_front_rhs(fi, rh);
    __shared__ double wsh[64];
    const FrontDesc fd = ds.fd[list[fi]];
    const int k = fd.k, m = fd.m, r = threadIdx.x;
    const double *P = Lx + fd.px;
    double *x = X + (int64_t)rh * ldx + fd.first;
    const double *wc = Wc + (int64_t)rh * wstride;
    double *wo = Wo + (int64_t)rh * wstride + fd.wx;
    double a[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; j++) a[j] = kvx_ld0(P, r + (int64_t)j * m, j < k && r < m);
    double w = kvx_ld0(x, r, r < k);
    if (fd.nchild > 0) {
        wsh[r] = w;
        for (int c = 0; c < fd.nchild; c++) {
            const ChildDesc cd = ds.cd[fd.childptr + c];
            if (cd.uc == 0) continue;
            const bool ok = r < cd.uc;
            const int t = ok ? ds.rel[cd.rel + r] : 0;
            const double v = kvx_ld0(wc + cd.wx, r, ok);
            __syncthreads();
            if (ok) wsh[t] += v;
        }
        __syncthreads();
        w = wsh[r];
    }
#pragma unroll
    for (int j = 0; j < KMAX; j++) {
        if (j < k) {                               // wave-uniform
            const double yj = kvx_readlane(w, j) / kvx_readlane(a[j], j);
            w = (r == j) ? yj : (r > j ? __builtin_fma(-a[j], yj, w) : w);
        }
    }
    if (r < k) x[r] = w;
    else if (r < m) wo[r - k] = w;
}

// Backward: lane j = pivot column j with its panel COLUMN in registers; rows are swept from the
// bottom, the solved value of row r is broadcast by readlane and every column to its left adds
// its L[r][j] * x_r.
template <int KMAX, int MMAX>
__global__ __launch_bounds__(64) void k_bwd_wave(DevSym ds, const int32_t *__restrict__ list,
                                                 const double *__restrict__ Lx, double *__restrict__ X, int64_t ldx)
{
    unsigned fi, rh;
    kvx_front_rhs(fi, rh);
    const FrontDesc fd = ds.fd[list[fi]];
    const int k = fd.k, m = fd.m, ln = threadIdx.x;
    const double *P = Lx + fd.px;
    double *xg = X + (int64_t)rh * ldx;
    const int32_t *rows = ds.rowidx + fd.rowptr;
    // lane ln < k holds column ln: a[r] = L[r][ln]; lane ln also carries x of row ln (rows < m)
    const int col = ln < k ? ln : 0;
    double a[MMAX];
#pragma unroll
    for (int rr = 0; rr < MMAX; rr++) a[rr] = kvx_ld0(P, rr + (int64_t)col * m, ln < k && rr < m && rr >= ln);
    const int grow = (ln < m) ? (ln < k ? fd.first + ln : rows[ln]) : 0;
    double xv = kvx_ld0(xg, grow, ln < m);         // y (pivot rows) or already-solved ancestors (update rows)
    double acc = 0.0;
#pragma unroll
    for (int rr = MMAX - 1; rr >= 0; rr--) {
        if (rr < m) {                              // wave-uniform
            double xr;
            if (rr < k) {
                // finalise row rr: lane rr owns acc and the diagonal
                const double t = (xv - acc) / ((ln == rr) ? a[rr] : 1.0);
                xr = kvx_readlane(t, rr);
                if (ln == rr) xv = xr;
            } else {
                xr = kvx_readlane(xv, rr);
            }
            acc = (ln < rr) ? __builtin_fma(a[rr], xr, acc) : acc;
        }
    }
    if (ln < k) xg[fd.first + ln] = xv;
}

// ------------------------------------------------------------------------------------------
// Triangular solves for the LDS classes (m <= 128, k <= 64): two wavefronts per front and rhs, same
// register layout as the wave kernels.  Forward: thread r = row r with its panel row in registers;
// wave 0 (rows 0..63, which hold every pivot) runs the k dependent steps -- the pivot value travels by
// readlane, the division is a multiplication by a reciprocal computed once per lane -- and wave 1
// (rows 64..) then needs only a dot product with the finished y.
template <int KMAX>
__global__ __launch_bounds__(128) void k_fwd_lds(DevSym ds, const int32_t *__restrict__ list,
                                                 const double *__restrict__ Lx, double *__restrict__ X, int64_t ldx,
                                                 const double *__restrict__ Wc, double *__restrict__ Wo, int64_t wstride)
{
    unsigned fi, rh;
    kvx_front_rhs(fi, rh);
    __shared__ double wsh[128];
    __shared__ double ysh[64];
    const FrontDesc fd = ds.fd[list[fi]];
    const int k = fd.k, m = fd.m, r = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(r >> 6);
    const double *P = Lx + fd.px;
    double *x = X + (int64_t)rh * ldx + fd.first;
    const double *wc = Wc + (int64_t)rh * wstride;
    double *wo = Wo + (int64_t)rh * wstride + fd.wx;
    // strictly below the diagonal only: with a[j] = 0 for r <= j a step is one unconditional fma per lane (a lane above the pivot
    // adds zero), and the finished y_r = w_r / d_r is formed once, after the sweep -- no select in the dependent chain.
    // FINITE DATA ASSUMED: 0 * y_j is 0 only for a finite y_j -- an Inf / NaN in one component of a front's part of y reaches all
    // of its rows (the select form left the rows above j alone).  A valid factor and a finite right-hand side give finite y; the
    // enqueue-only entry points (kvx_chol_*_async_dev) may run these sweeps through a FAILED factor before its status is read:
    // what they leave in X is then garbage as a whole, never partly valid (kvx_chol_status / the next synchronising call
    // reports KVX_ENOTPOSDEF, and the host layer discards X: lp.py checks the status before it uses a direction).
    double a[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; j++) a[j] = kvx_ld0(P, r + (int64_t)j * m, j < k && r < m && r > j);
    const double dg = kvx_ld0(P, r + (int64_t)r * m, r < k);
    double w = kvx_ld0(x, r, r < k);
    if (fd.nchild > 0) {
        wsh[r] = w;
        ChildDesc cd = ds.cd[fd.childptr];
        bool ok = r < cd.uc;
        int t = ok ? ds.rel[cd.rel + r] : 0;
        double v = kvx_ld0(wc + cd.wx, r, ok);
        for (int c = 0; c < fd.nchild; c++) {
            ChildDesc nx = cd;
            if (c + 1 < fd.nchild) nx = ds.cd[fd.childptr + c + 1];
            __syncthreads();
            if (ok) wsh[t] += v;
            if (c + 1 < fd.nchild) {
                ok = r < nx.uc;
                t = ok ? ds.rel[nx.rel + r] : 0;
                v = kvx_ld0(wc + nx.wx, r, ok);
            }
            cd = nx;
        }
        __syncthreads();
        w = wsh[r];
    }
    if (wv == 0) {
        const double rinv = 1.0 / (r < k ? dg : 1.0);
#pragma unroll
        for (int j = 0; j < KMAX; j++) {
            if (j < k) {                           // workgroup-uniform
                const double yj = kvx_readlane(w * rinv, j);
                w = __builtin_fma(-a[j], yj, w);
            }
        }
        if (r < k) { w *= rinv; ysh[r] = w; }
    }
    __syncthreads();
    if (wv == 1) {
#pragma unroll
        for (int j = 0; j < KMAX; j++) w = __builtin_fma(-a[j], ysh[j < k ? j : 0], w);   // a[j] = 0 for j >= k
    }
    if (r < k) x[r] = w;
    else if (r < m) wo[r - k] = w;
}

// Backward: thread (h, j) holds rows 64 h .. 64 h + 63 of pivot column j.  The part of the update rows
// (already-solved ancestors) is a dot product shared by the two waves; the pivot rows are then swept
// from the bottom by wave 0, the solved value travelling by readlane.
__global__ __launch_bounds__(128) void k_bwd_lds(DevSym ds, const int32_t *__restrict__ list,
                                                 const double *__restrict__ Lx, double *__restrict__ X, int64_t ldx)
{
    unsigned fi, rh;
    kvx_front_rhs(fi, rh);
    __shared__ double xf[128];     // y (pivot rows) / solved ancestors (update rows)
    __shared__ double xu[128];     // the same with the pivot rows zeroed
    __shared__ double accsh[64];
    const FrontDesc fd = ds.fd[list[fi]];
    const int k = fd.k, m = fd.m, tid = threadIdx.x, ln = tid & 63;
    const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double *P = Lx + fd.px;
    double *xg = X + (int64_t)rh * ldx;
    const int32_t *rows = ds.rowidx + fd.rowptr;
    const int col = ln < k ? ln : 0;
    double a[64];
#pragma unroll
    for (int rr = 0; rr < 64; rr++) {
        const int row = 64 * h + rr;
        a[rr] = kvx_ld0(P, row + (int64_t)col * m, ln < k && row < m && row > ln);      // strictly lower: no select in the sweep
    }
    const double dg = kvx_ld0(P, ln + (int64_t)ln * m, ln < k);
    const int grow = (tid < m) ? (tid < k ? fd.first + tid : rows[tid]) : 0;
    const double xin = kvx_ld0(xg, grow, tid < m);
    xf[tid] = xin;
    xu[tid] = (tid >= k) ? xin : 0.0;
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int rr = 0; rr < 64; rr++) acc = __builtin_fma(a[rr], xu[64 * h + rr], acc);
    if (h == 1) accsh[ln] = acc;
    __syncthreads();
    if (h == 0) {
        acc += accsh[ln];
        const double rinv = 1.0 / (ln < k ? dg : 1.0);
        double xv = xf[ln];
#pragma unroll
        for (int rr = 63; rr >= 0; rr--) {
            if (rr < k) {                          // workgroup-uniform
                const double xr = kvx_readlane((xv - acc) * rinv, rr);
                xv = (ln == rr) ? xr : xv;
                acc = __builtin_fma(a[rr], xr, acc);
            }
        }
        if (ln < k) xg[fd.first + ln] = xv;
    }
}

void launch_fwd_lds(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int kmax,
                    const double *Lx, double *X, int64_t ldx, int nrhs, const double *Wchild, double *Wout, int64_t wstride)
{
    if (count <= 0 || nrhs <= 0) return;
    dim3 grid((unsigned)count, (unsigned)nrhs);
    if (kmax <= 32)
        hipLaunchKernelGGL(k_fwd_lds<32>, grid, dim3(128), 0, st, ds, list, Lx, X, ldx, Wchild, Wout, wstride);
    else
        hipLaunchKernelGGL(k_fwd_lds<64>, grid, dim3(128), 0, st, ds, list, Lx, X, ldx, Wchild, Wout, wstride);
}

void launch_bwd_lds(hipStream_t st, const DevSym &ds, const int32_t *list, int count,
                    const double *Lx, double *X, int64_t ldx, int nrhs)
{
    if (count <= 0 || nrhs <= 0) return;
    dim3 grid((unsigned)count, (unsigned)nrhs);
    hipLaunchKernelGGL(k_bwd_lds, grid, dim3(128), 0, st, ds, list, Lx, X, ldx);
}

// ------------------------------------------------------------------------------------------
// Leaf subtrees in the solves.  The bottom of the elimination tree is thousands of small independent subtrees
// (a nested-dissection leaf ordered by minimum degree: 10-40 wave-class fronts, 8 levels deep).  Level by level
// they cost one launch + one stream join per level and direction for a few microseconds of work per front.  Here
// ONE wavefront walks a whole subtree: fronts in postorder (forward) / reverse postorder (backward), its slice of
// x in LDS, the update vectors of finished children on an LDS stack (offsets precomputed on the host) -- nothing
// but the panels is read from HBM and no level boundary is crossed.  Only the root's update vector goes out.
// Per front the walk costs two dependent HBM round trips, not five: the next front's descriptor is fetched while
// the current one is solved, the panel / diagonal / child records of a front go out together, and the children's
// relative indices (second trip) for all children at once.  edges[3 c .. 3 c + 2] = (update rows, offset of the
// relative indices, LDS stack offset) of child c, indexed like DevSym::children.
template <int KMAX>
__global__ __launch_bounds__(64) void k_fwd_subtree(DevSym ds, const SubDesc *__restrict__ subs,
                                                    const int32_t *__restrict__ edges, const double *__restrict__ Lx,
                                                    double *__restrict__ X, int64_t ldx,
                                                    double *__restrict__ W0, double *__restrict__ W1, int64_t wstride,
                                                    const int32_t *__restrict__ depth)
{
    unsigned fi, rh;
    kvx_front_rhs(fi, rh);
    constexpr int NC = 4;                          // children handled per batch
    __shared__ double xs[KVX_SUB_MAXCOLS];
    __shared__ double stk[KVX_SUB_STACK];
    __shared__ double wsh[64];
    const SubDesc sd = subs[fi];
    const int r = threadIdx.x;
    double *x = X + (int64_t)rh * ldx + sd.col0;
    FrontDesc nxt = ds.fd[sd.lo];
    for (int i = r; i < sd.ncols; i += 64) xs[i] = x[i];
    __syncthreads();
    int sp = 0;                                    // stack pointer: mirrors the host's offsets (children are on top)
    for (int s = sd.lo; s <= sd.hi; s++) {
        const FrontDesc fd = nxt;
        if (s < sd.hi) nxt = ds.fd[s + 1];
        const int k = fd.k, m = fd.m, xo = fd.first - sd.col0;
        const double *P = Lx + fd.px;
        double a[KMAX];
#pragma unroll
        for (int j0 = 0; j0 < KMAX; j0 += 8) {
            if (j0 < k) {                          // wave-uniform: leaf fronts have few pivots, skip the unused column groups
#pragma unroll
                for (int j = j0; j < j0 + 8; j++) a[j] = kvx_ld0(P, r + (int64_t)j * m, j < k && r < m && r > j);   // strictly lower (k_fwd_lds)
            } else {
#pragma unroll
                for (int j = j0; j < j0 + 8; j++) a[j] = 0.0;
            }
        }
        const double dg = kvx_ld0(P, r + (int64_t)r * m, r < k);
        double w = r < k ? xs[xo + r] : 0.0;
        if (fd.nchild > 0) {
            wsh[r] = w;
            for (int c0 = 0; c0 < fd.nchild; c0 += NC) {
                int uc[NC], rp[NC], wo[NC], t[NC];
                double v[NC];
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const bool has = c0 + c < fd.nchild;
                    const int32_t *e = edges + 3 * (fd.childptr + (has ? c0 + c : 0));
                    uc[c] = has ? e[0] : 0;
                    rp[c] = e[1];
                    wo[c] = e[2];
                }
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const bool ok = r < uc[c];
                    t[c] = ds.rel[rp[c] + (ok ? r : 0)];
                    v[c] = ok ? stk[wo[c] + r] : 0.0;
                }
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    __syncthreads();               // (one wave: orders the LDS read-modify-writes of successive children)
                    if (r < uc[c]) wsh[t[c]] += v[c];
                    sp -= uc[c];
                }
            }
            __syncthreads();
            w = wsh[r];
        }
        const double rinv = 1.0 / (r < k ? dg : 1.0);
#pragma unroll
        for (int j = 0; j < KMAX; j++) {
            if (j < k) {                           // wave-uniform
                const double yj = kvx_readlane(w * rinv, j);
                w = __builtin_fma(-a[j], yj, w);
            }
        }
        if (r < k) xs[xo + r] = w * rinv;
        else if (r < m) {
            if (s == sd.hi) {
                double *wo = ((depth[s] & 1) ? W1 : W0) + (int64_t)rh * wstride + fd.wx;
                wo[r - k] = w;
            } else {
                stk[sp + r - k] = w;
            }
        }
        sp += m - k;
        __syncthreads();
    }
    for (int i = r; i < sd.ncols; i += 64) x[i] = xs[i];
}

template <int MMAX>
__global__ __launch_bounds__(64) void k_bwd_subtree(DevSym ds, const SubDesc *__restrict__ subs,
                                                    const double *__restrict__ Lx, double *__restrict__ X, int64_t ldx)
{
    unsigned fi, rh;
    kvx_front_rhs(fi, rh);
    __shared__ double xs[KVX_SUB_MAXCOLS];
    const SubDesc sd = subs[fi];
    const int ln = threadIdx.x;
    double *xg = X + (int64_t)rh * ldx;
    FrontDesc nxt = ds.fd[sd.hi];
    int nrow = (ln < nxt.m && ln >= nxt.k) ? ds.rowidx[nxt.rowptr + ln] : sd.col0;   // global index of update row ln
    for (int i = ln; i < sd.ncols; i += 64) xs[i] = xg[sd.col0 + i];
    __syncthreads();
    for (int s = sd.hi; s >= sd.lo; s--) {
        const FrontDesc fd = nxt;
        const int myrow = nrow;
        if (s > sd.lo) nxt = ds.fd[s - 1];
        const int k = fd.k, m = fd.m;
        const double *P = Lx + fd.px;
        const int col = ln < k ? ln : 0;
        // lane = pivot column with its entries in registers (a column per lane is an uncoalesced read; staging the panel
        // through LDS instead was measured and lost: 17 KB per wave costs more occupancy than the coalescing wins)
        double a[MMAX];
#pragma unroll
        for (int r0 = 0; r0 < MMAX; r0 += 16) {
            if (r0 < m) {                          // wave-uniform
#pragma unroll
                for (int rr = r0; rr < r0 + 16; rr++) a[rr] = kvx_ld0(P, rr + (int64_t)col * m, ln < k && rr < m && rr > ln);   // strictly lower
            } else {
#pragma unroll
                for (int rr = r0; rr < r0 + 16; rr++) a[rr] = 0.0;
            }
        }
        const double dg = kvx_ld0(P, ln + (int64_t)ln * m, ln < k);
        const int grow = (ln < m) ? (ln < k ? fd.first + ln : myrow) : sd.col0;
        const int loc = grow - sd.col0;
        const bool inl = loc >= 0 && loc < sd.ncols;         // ancestors inside the subtree live in the LDS slice
        const double xglob = kvx_ld0(xg, grow, ln < m && !inl);
        if (s > sd.lo) nrow = (ln < nxt.m && ln >= nxt.k) ? ds.rowidx[nxt.rowptr + ln] : sd.col0;   // next front's rows: in flight during the sweep
        double xv = inl ? xs[loc] : xglob;
        if (ln >= m) xv = 0.0;
        const double rinv = 1.0 / (ln < k ? dg : 1.0);
        double acc = 0.0;
#pragma unroll
        for (int rr = MMAX - 1; rr >= 0; rr--) {
            if (rr < m) {                          // wave-uniform
                double xr;
                if (rr < k) {
                    xr = kvx_readlane((xv - acc) * rinv, rr);
                    if (ln == rr) xv = xr;
                } else {
                    xr = kvx_readlane(xv, rr);
                }
                acc = __builtin_fma(a[rr], xr, acc);
            }
        }
        if (ln < k) xs[fd.first - sd.col0 + ln] = xv;
        __syncthreads();
    }
    for (int i = ln; i < sd.ncols; i += 64) xg[sd.col0 + i] = xs[i];
}

// ------------------------------------------------------------------------------------------
// Subtree walks for a BLOCK of RB right-hand sides per wavefront (used from 4 right-hand sides on).  The walk of
// k_fwd_subtree / k_bwd_subtree pays, per front, a descriptor, the panel (a column per lane in the backward sweep: uncoalesced)
// and the relative indices, then one dependent chain of readlane steps; with one right-hand side per wave all of that is paid
// nrhs times and the chain has no instruction-level parallelism.  Here the panel registers are loaded once per front and the RB
// substitution chains are independent instruction streams that the compiler interleaves.  Right-hand sides past nrhs (a ragged
// last block) are computed on a duplicate of the last valid one and not stored.
template <int KMAX, int RB>
__global__ __launch_bounds__(64) void k_fwd_subtree_mr(DevSym ds, const SubDesc *__restrict__ subs,
                                                       const int32_t *__restrict__ edges, const double *__restrict__ Lx,
                                                       double *__restrict__ X, int64_t ldx, int nrhs,
                                                       double *__restrict__ W0, double *__restrict__ W1, int64_t wstride,
                                                       const int32_t *__restrict__ depth)
{
    __shared__ double xs[RB][KVX_SUB_MAXCOLS];
    __shared__ double stk[RB][KVX_SUB_STACK];
    __shared__ double wsh[RB][64];
    unsigned fi, rg;
    kvx_front_rhs(fi, rg);
    const SubDesc sd = subs[fi];
    const int r = threadIdx.x;
    const int r0 = (int)rg * RB, nv = min(RB, nrhs - r0);
    int64_t xoff[RB];
#pragma unroll
    for (int b = 0; b < RB; b++) xoff[b] = (int64_t)(r0 + min(b, nv - 1)) * ldx + sd.col0;
    FrontDesc nxt = ds.fd[sd.lo];
#pragma unroll
    for (int b = 0; b < RB; b++)
        for (int i = r; i < sd.ncols; i += 64) xs[b][i] = X[xoff[b] + i];
    __syncthreads();
    int sp = 0;
    for (int s = sd.lo; s <= sd.hi; s++) {
        const FrontDesc fd = nxt;
        if (s < sd.hi) nxt = ds.fd[s + 1];
        const int k = fd.k, m = fd.m, xo = fd.first - sd.col0;
        const double *P = Lx + fd.px;
        double a[KMAX];
#pragma unroll
        for (int j0 = 0; j0 < KMAX; j0 += 8) {
            if (j0 < k) {
#pragma unroll
                for (int j = j0; j < j0 + 8; j++) a[j] = kvx_ld0(P, r + (int64_t)j * m, j < k && r < m && r > j);   // strictly lower (k_fwd_lds)
            } else {
#pragma unroll
                for (int j = j0; j < j0 + 8; j++) a[j] = 0.0;
            }
        }
        const double dg = kvx_ld0(P, r + (int64_t)r * m, r < k);
        double w[RB];
#pragma unroll
        for (int b = 0; b < RB; b++) w[b] = r < k ? xs[b][xo + r] : 0.0;
        if (fd.nchild > 0) {
#pragma unroll
            for (int b = 0; b < RB; b++) wsh[b][r] = w[b];
            for (int c = 0; c < fd.nchild; c++) {
                const int32_t *e = edges + 3 * (fd.childptr + c);
                const int uc = e[0], rp = e[1], wo = e[2];
                const bool ok = r < uc;
                const int t = ds.rel[rp + (ok ? r : 0)];
                __syncthreads();                   // (one wave: orders the LDS read-modify-writes of successive children)
#pragma unroll
                for (int b = 0; b < RB; b++)
                    if (ok) wsh[b][t] += stk[b][wo + r];
                sp -= uc;
            }
            __syncthreads();
#pragma unroll
            for (int b = 0; b < RB; b++) w[b] = wsh[b][r];
        }
        const double rinv = 1.0 / (r < k ? dg : 1.0);
#pragma unroll
        for (int j = 0; j < KMAX; j++) {
            if (j < k) {                           // wave-uniform
#pragma unroll
                for (int b = 0; b < RB; b++) {
                    const double yj = kvx_readlane(w[b] * rinv, j);
                    w[b] = __builtin_fma(-a[j], yj, w[b]);
                }
            }
        }
        if (r < k) {
#pragma unroll
            for (int b = 0; b < RB; b++) xs[b][xo + r] = w[b] * rinv;
        } else if (r < m) {
            if (s == sd.hi) {
                double *wo = ((depth[s] & 1) ? W1 : W0) + fd.wx + (r - k);
#pragma unroll
                for (int b = 0; b < RB; b++)
                    if (b < nv) wo[(int64_t)(r0 + b) * wstride] = w[b];
            } else {
#pragma unroll
                for (int b = 0; b < RB; b++) stk[b][sp + r - k] = w[b];
            }
        }
        sp += m - k;
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < RB; b++)
        if (b < nv)
            for (int i = r; i < sd.ncols; i += 64) X[xoff[b] + i] = xs[b][i];
}

template <int MMAX, int RB>
__global__ __launch_bounds__(64) void k_bwd_subtree_mr(DevSym ds, const SubDesc *__restrict__ subs,
                                                       const double *__restrict__ Lx, double *__restrict__ X, int64_t ldx, int nrhs)
{
    __shared__ double xs[RB][KVX_SUB_MAXCOLS];
    unsigned fi, rg;
    kvx_front_rhs(fi, rg);
    const SubDesc sd = subs[fi];
    const int ln = threadIdx.x;
    const int r0 = (int)rg * RB, nv = min(RB, nrhs - r0);
    int64_t xoff[RB];
#pragma unroll
    for (int b = 0; b < RB; b++) xoff[b] = (int64_t)(r0 + min(b, nv - 1)) * ldx;
    FrontDesc nxt = ds.fd[sd.hi];
    int nrow = (ln < nxt.m && ln >= nxt.k) ? ds.rowidx[nxt.rowptr + ln] : sd.col0;
#pragma unroll
    for (int b = 0; b < RB; b++)
        for (int i = ln; i < sd.ncols; i += 64) xs[b][i] = X[xoff[b] + sd.col0 + i];
    __syncthreads();
    for (int s = sd.hi; s >= sd.lo; s--) {
        const FrontDesc fd = nxt;
        const int myrow = nrow;
        if (s > sd.lo) nxt = ds.fd[s - 1];
        const int k = fd.k, m = fd.m;
        const double *P = Lx + fd.px;
        const int col = ln < k ? ln : 0;
        double a[MMAX];
#pragma unroll
        for (int q0 = 0; q0 < MMAX; q0 += 16) {
            if (q0 < m) {                          // wave-uniform
#pragma unroll
                for (int rr = q0; rr < q0 + 16; rr++) a[rr] = kvx_ld0(P, rr + (int64_t)col * m, ln < k && rr < m && rr > ln);
            } else {
#pragma unroll
                for (int rr = q0; rr < q0 + 16; rr++) a[rr] = 0.0;
            }
        }
        const double dg = kvx_ld0(P, ln + (int64_t)ln * m, ln < k);
        const int grow = (ln < m) ? (ln < k ? fd.first + ln : myrow) : sd.col0;
        const int loc = grow - sd.col0;
        const bool inl = loc >= 0 && loc < sd.ncols;
        double xv[RB], acc[RB];
#pragma unroll
        for (int b = 0; b < RB; b++) xv[b] = kvx_ld0(X + xoff[b], grow, ln < m && !inl);
        if (s > sd.lo) nrow = (ln < nxt.m && ln >= nxt.k) ? ds.rowidx[nxt.rowptr + ln] : sd.col0;
#pragma unroll
        for (int b = 0; b < RB; b++) {
            xv[b] = inl ? xs[b][loc] : xv[b];
            if (ln >= m) xv[b] = 0.0;
            acc[b] = 0.0;
        }
        const double rinv = 1.0 / (ln < k ? dg : 1.0);
#pragma unroll
        for (int rr = MMAX - 1; rr >= 0; rr--) {
            if (rr < m) {                          // wave-uniform
#pragma unroll
                for (int b = 0; b < RB; b++) {
                    double xr;
                    if (rr < k) {
                        xr = kvx_readlane((xv[b] - acc[b]) * rinv, rr);
                        if (ln == rr) xv[b] = xr;
                    } else {
                        xr = kvx_readlane(xv[b], rr);
                    }
                    acc[b] = __builtin_fma(a[rr], xr, acc[b]);
                }
            }
        }
        if (ln < k) {
#pragma unroll
            for (int b = 0; b < RB; b++) xs[b][fd.first - sd.col0 + ln] = xv[b];
        }
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < RB; b++)
        if (b < nv)
            for (int i = ln; i < sd.ncols; i += 64) X[xoff[b] + sd.col0 + i] = xs[b][i];
}

constexpr int KVX_SUB_RB = 4;      // right-hand sides per wavefront in the blocked subtree walks
// Measured (MI355X, scratch/multirhs.py): the blocked walks win from ~16 right-hand sides on (n = 1e6: 64 rhs 25.4 -> 21.7 ms,
// 16 rhs 7.8 -> 6.9 ms; n = 50 000: 200 rhs 3.55 -> 3.04 ms) and lose below that on small systems, where the solve is bound by
// the length of one wave's dependent chain (n = 50 000, 4 rhs: 0.56 -> 0.64 ms).
constexpr int KVX_SUB_MR_FROM = 16;

void launch_fwd_subtree(hipStream_t st, const DevSym &ds, const SubDesc *subs, int nsub, const int32_t *edges,
                        const double *Lx, double *X, int64_t ldx, int nrhs, double *W0, double *W1, int64_t wstride,
                        const int32_t *depth)
{
    if (nsub <= 0 || nrhs <= 0) return;
    if (nrhs >= KVX_SUB_MR_FROM) {
        hipLaunchKernelGGL((k_fwd_subtree_mr<32, KVX_SUB_RB>), dim3((unsigned)nsub, (unsigned)((nrhs + KVX_SUB_RB - 1) / KVX_SUB_RB)), dim3(64), 0, st,
                           ds, subs, edges, Lx, X, ldx, nrhs, W0, W1, wstride, depth);
        return;
    }
    hipLaunchKernelGGL(k_fwd_subtree<32>, dim3((unsigned)nsub, (unsigned)nrhs), dim3(64), 0, st, ds, subs, edges, Lx, X, ldx,
                       W0, W1, wstride, depth);
}

// one size group of the subtree table (every front of its subtrees has order <= mcap, 32 / 48 / 64: registers for a column of that
// height only, so more walks are resident at once) on a stream of its own: enqueue_bwd runs the three groups side by side
void launch_bwd_subtree_group(hipStream_t st, int mcap, const DevSym &ds, const SubDesc *subs, int count, const double *Lx, double *X,
                              int64_t ldx)
{
    if (count <= 0) return;
    if (mcap <= 32) hipLaunchKernelGGL(k_bwd_subtree<32>, dim3((unsigned)count, 1u), dim3(64), 0, st, ds, subs, Lx, X, ldx);
    else if (mcap <= 48) hipLaunchKernelGGL(k_bwd_subtree<48>, dim3((unsigned)count, 1u), dim3(64), 0, st, ds, subs, Lx, X, ldx);
    else hipLaunchKernelGGL(k_bwd_subtree<64>, dim3((unsigned)count, 1u), dim3(64), 0, st, ds, subs, Lx, X, ldx);
}

// the subtree table holds the subtrees whose fronts all have order <= 32 first (nsub32 of them): half the registers
void launch_bwd_subtree(hipStream_t st, const DevSym &ds, const SubDesc *subs, int nsub, int nsub32, const double *Lx, double *X,
                        int64_t ldx, int nrhs)
{
    if (nsub <= 0 || nrhs <= 0) return;
    if (nrhs >= KVX_SUB_MR_FROM) {
        const unsigned gy = (unsigned)((nrhs + KVX_SUB_RB - 1) / KVX_SUB_RB);
        if (nsub32 > 0)
            hipLaunchKernelGGL((k_bwd_subtree_mr<32, KVX_SUB_RB>), dim3((unsigned)nsub32, gy), dim3(64), 0, st, ds, subs, Lx, X, ldx, nrhs);
        if (nsub > nsub32)
            hipLaunchKernelGGL((k_bwd_subtree_mr<64, KVX_SUB_RB>), dim3((unsigned)(nsub - nsub32), gy), dim3(64), 0, st, ds, subs + nsub32, Lx, X, ldx, nrhs);
        return;
    }
    if (nsub32 > 0)
        hipLaunchKernelGGL(k_bwd_subtree<32>, dim3((unsigned)nsub32, (unsigned)nrhs), dim3(64), 0, st, ds, subs, Lx, X, ldx);
    if (nsub > nsub32)
        hipLaunchKernelGGL(k_bwd_subtree<64>, dim3((unsigned)(nsub - nsub32), (unsigned)nrhs), dim3(64), 0, st, ds, subs + nsub32, Lx, X, ldx);
}

void launch_fwd_wave(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int kmax,
                     const double *Lx, double *X, int64_t ldx, int nrhs, const double *Wchild, double *Wout, int64_t wstride)
{
    if (count <= 0 || nrhs <= 0) return;
    dim3 grid((unsigned)count, (unsigned)nrhs);
    if (kmax <= 16)
        hipLaunchKernelGGL(k_fwd_wave<16>, grid, dim3(64), 0, st, ds, list, Lx, X, ldx, Wchild, Wout, wstride);
    else
        hipLaunchKernelGGL(k_fwd_wave<32>, grid, dim3(64), 0, st, ds, list, Lx, X, ldx, Wchild, Wout, wstride);
}

void launch_bwd_wave(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int mcap, int kmax,
                     const double *Lx, double *X, int64_t ldx, int nrhs)
{
    if (count <= 0 || nrhs <= 0) return;
    (void)kmax;
    dim3 grid((unsigned)count, (unsigned)nrhs);
    if (mcap <= 32)
        hipLaunchKernelGGL((k_bwd_wave<32, 32>), grid, dim3(64), 0, st, ds, list, Lx, X, ldx);
    else if (mcap <= 48)
        hipLaunchKernelGGL((k_bwd_wave<32, 48>), grid, dim3(64), 0, st, ds, list, Lx, X, ldx);
    else
        hipLaunchKernelGGL((k_bwd_wave<32, 64>), grid, dim3(64), 0, st, ds, list, Lx, X, ldx);
}

// wave kernel: mcap = LDS image capacity (32 / 48 / 64); kmax: 16 or 32
void launch_front_wave(hipStream_t st, int mcap, int kmax, const DevSym &ds, const int32_t *list, int count,
                       double *Lx, const double *Uchild, double *Uout, int *status)
{
    if (count <= 0) return;
    const size_t lds = (size_t)mcap * (mcap + 1) / 2 * sizeof(double) + 128 * sizeof(double) + 64 * sizeof(int);
    static const bool tight = [] { const char *e = getenv("KVX_WAVE_OCC"); return !(e && e[0] == '0'); }();   // KVX_WAVE_OCC=0: the compiler's own allocation
    if (kmax <= 16) {
        if (tight) hipLaunchKernelGGL((k_front_wave<16, 5>), dim3((unsigned)count), dim3(64), lds, st, ds, list, Lx, Uchild, Uout, status, mcap);
        else hipLaunchKernelGGL((k_front_wave<16, 1>), dim3((unsigned)count), dim3(64), lds, st, ds, list, Lx, Uchild, Uout, status, mcap);
    } else {
        if (tight) hipLaunchKernelGGL((k_front_wave<32, 4>), dim3((unsigned)count), dim3(64), lds, st, ds, list, Lx, Uchild, Uout, status, mcap);
        else hipLaunchKernelGGL((k_front_wave<32, 1>), dim3((unsigned)count), dim3(64), lds, st, ds, list, Lx, Uchild, Uout, status, mcap);
    }
}

// leaf subtrees of the factorisation: one wavefront per subtree; mcap = LDS image capacity of the largest front of these subtrees
void launch_factor_subtree(hipStream_t st, int mcap, const DevSym &ds, const SubDesc *subs, int nsub, const int32_t *depth,
                           double *Lx, double *U0, double *U1, int *status)
{
    if (nsub <= 0) return;
    const size_t lds = (size_t)mcap * (mcap + 1) / 2 * sizeof(double) + 128 * sizeof(double) + 64 * sizeof(int);
    hipLaunchKernelGGL((k_factor_subtree<32, 3>), dim3((unsigned)nsub), dim3(64), lds, st, ds, subs, depth, Lx, U0, U1, status, mcap);
}

// LDS kernel: mcap = 96 or 128; kmax = 32 or 64 (k of every front in the list must be <= kmax)
void launch_front_small(hipStream_t st, int mcap, int kmax, const DevSym &ds, const int32_t *list, int count,
                        double *Lx, const double *Uchild, double *Uout, int *status)
{
    if (count <= 0) return;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)k_front_lds<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
        (void)hipFuncSetAttribute((const void *)k_front_lds<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
        attr_set = true;
    }
    const size_t lds = (size_t)mcap * (mcap + 1) / 2 * sizeof(double) + 256 * sizeof(double) + 128 * sizeof(int);
    if (kmax <= 32)
        hipLaunchKernelGGL(k_front_lds<32>, dim3((unsigned)count), dim3(256), lds, st, ds, list, Lx, Uchild, Uout, status, mcap);
    else
        hipLaunchKernelGGL(k_front_lds<64>, dim3((unsigned)count), dim3(256), lds, st, ds, list, Lx, Uchild, Uout, status, mcap);
}

}  // namespace kvx
