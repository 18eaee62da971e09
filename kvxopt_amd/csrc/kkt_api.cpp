// C-ABI entry points for the NT scaling, normal-equations assembly and sparse mat-vec
// (include/kvxhip.h).  Device-only: no CPU fallback.
#include "../../include/kvxhip.h"
#include "abi_guard.hpp"
#include "devpool.hpp"
#include "kkt.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

using namespace kvx;

static thread_local std::string g_err2;
extern "C" const char *kvx_last_error(void);

#define HIPCHK(call)                                                             \
    do {                                                                         \
        hipError_t e_ = (call);                                                  \
        if (e_ != hipSuccess) return KVX_EDEVICE;                                \
    } while (0)

struct kvx_atda {
    int64_t ml = 0, n = 0, gnz = 0, snz = 0, pnz = 0;
    std::vector<int64_t> Sp, Si;
    std::vector<int64_t> pp;        // [snz+1]
    std::vector<int32_t> pa, pb;    // product list: indices into Gx
    std::vector<int64_t> pslot;     // per entry of the caller's P arrays: slot in S (or -1)
    bool dev = false;
    int64_t *d_pp = nullptr, *d_pslot = nullptr;
    int32_t *d_pa = nullptr, *d_pb = nullptr, *d_gi = nullptr;
    std::vector<int32_t> gi32;
    double *d_gx = nullptr, *d_w = nullptr, *d_px = nullptr, *d_sx = nullptr;   // staging for the host variant
    double *d_wg = nullptr;         // w[row] * G, refreshed by every assembly (gnz doubles)
};

namespace {

struct Scratch {
    double *part = nullptr;
    double *host = nullptr;     // pinned, as many doubles as `part`
    double *multi = nullptr;    // device, 32 results of kvx_nt_reduce_multi_dev
};
// KVX_LP_UNFUSED=1: the separate launches of rounds 1-2 (kept for the bitwise A/B of the fused kernels)
bool lp_unfused()
{
    static const bool v = [] { const char *e = getenv("KVX_LP_UNFUSED"); return e && atoi(e) != 0; }();
    return v;
}
Scratch &scratch()
{
    static thread_local Scratch s;
    return s;
}
int ensure_scratch()
{
    Scratch &s = scratch();
    if (s.part) return KVX_OK;
    HIPCHK(pool_malloc((void **)&s.part, reduce_scratch_doubles() * sizeof(double)));
    HIPCHK(hipHostMalloc((void **)&s.host, reduce_scratch_doubles() * sizeof(double), hipHostMallocDefault));
    HIPCHK(pool_malloc((void **)&s.multi, 32 * sizeof(double)));
    return KVX_OK;
}

template <class T>
int up(T **dst, const std::vector<T> &src)
{
    HIPCHK(pool_malloc((void **)dst, std::max<size_t>(src.size(), 1) * sizeof(T)));
    if (!src.empty()) HIPCHK(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return KVX_OK;
}

int atda_device(kvx_atda *T)
{
    if (T->dev) return KVX_OK;
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) return KVX_EDEVICE;
    int rc;
    if ((rc = up(&T->d_pp, T->pp))) return rc;
    if ((rc = up(&T->d_pa, T->pa))) return rc;
    if ((rc = up(&T->d_pb, T->pb))) return rc;
    if ((rc = up(&T->d_gi, T->gi32))) return rc;
    if ((rc = up(&T->d_pslot, T->pslot))) return rc;
    HIPCHK(pool_malloc((void **)&T->d_wg, std::max<int64_t>(T->gnz, 1) * sizeof(double)));
    T->dev = true;
    return KVX_OK;
}

}  // namespace

extern "C" {

static int kvx_atda_plan_impl(int64_t ml, int64_t n, const int64_t *Gp, const int64_t *Gi, const int64_t *Pp, const int64_t *Pi,
                  kvx_atda **out)
{
    if (!out || ml < 0 || n < 0 || (n > 0 && !Gp)) return KVX_EINVAL;
    std::unique_ptr<kvx_atda> hold(new kvx_atda());
    kvx_atda *T = hold.get();
    T->ml = ml; T->n = n;
    const int64_t gnz = n ? Gp[n] : 0;
    if (gnz >= ((int64_t)1 << 31)) { return KVX_EINVAL; }
    T->gnz = gnz;
    T->gi32.resize((size_t)gnz);
    for (int64_t p = 0; p < gnz; p++) {
        if (Gi[p] < 0 || Gi[p] >= ml) { return KVX_EINVAL; }
        T->gi32[p] = (int32_t)Gi[p];
    }
    // CSR view of G: for every row the (column, CCS position) pairs, columns ascending
    std::vector<int64_t> rptr((size_t)ml + 1, 0);
    for (int64_t p = 0; p < gnz; p++) rptr[Gi[p] + 1]++;
    for (int64_t r = 0; r < ml; r++) rptr[r + 1] += rptr[r];
    std::vector<int32_t> rcol((size_t)gnz), rpos((size_t)gnz);
    {
        std::vector<int64_t> cur(rptr.begin(), rptr.end() - 1);
        for (int64_t j = 0; j < n; j++)
            for (int64_t p = Gp[j]; p < Gp[j + 1]; p++) {
                int64_t q = cur[Gi[p]]++;
                rcol[q] = (int32_t)j;
                rpos[q] = (int32_t)p;
            }
    }
    // pattern of tril(G'G) U tril(P), column by column
    std::vector<int64_t> mark((size_t)n, -1);
    std::vector<int64_t> col;
    T->Sp.assign((size_t)n + 1, 0);
    for (int64_t j = 0; j < n; j++) {
        col.clear();
        for (int64_t p = Gp[j]; p < Gp[j + 1]; p++) {
            int64_t r = Gi[p];
            for (int64_t q = rptr[r]; q < rptr[r + 1]; q++) {
                int64_t i = rcol[q];
                if (i >= j && mark[i] != j) { mark[i] = j; col.push_back(i); }
            }
        }
        if (Pp)
            for (int64_t p = Pp[j]; p < Pp[j + 1]; p++) {
                int64_t i = Pi[p];
                if (i < 0 || i >= n) { return KVX_EINVAL; }
                if (i >= j && mark[i] != j) { mark[i] = j; col.push_back(i); }
            }
        std::sort(col.begin(), col.end());
        T->Si.insert(T->Si.end(), col.begin(), col.end());
        T->Sp[j + 1] = (int64_t)T->Si.size();
    }
    T->snz = (int64_t)T->Si.size();
    // product lists
    T->pp.assign((size_t)T->snz + 1, 0);
    std::vector<int64_t> slot((size_t)n, -1);
    for (int pass = 0; pass < 2; pass++) {
        std::vector<int64_t> cur;
        if (pass == 1) {
            for (int64_t e = 0; e < T->snz; e++) T->pp[e + 1] += T->pp[e];
            if (T->pp[T->snz] >= ((int64_t)1 << 40)) { return KVX_ENOMEM; }
            T->pa.resize((size_t)T->pp[T->snz]);
            T->pb.resize((size_t)T->pp[T->snz]);
            cur.assign(T->pp.begin(), T->pp.end() - 1);
        }
        for (int64_t j = 0; j < n; j++) {
            for (int64_t e = T->Sp[j]; e < T->Sp[j + 1]; e++) slot[T->Si[e]] = e;
            for (int64_t p = Gp[j]; p < Gp[j + 1]; p++) {
                int64_t r = Gi[p];
                for (int64_t q = rptr[r]; q < rptr[r + 1]; q++) {
                    int64_t i = rcol[q];
                    if (i < j) continue;
                    int64_t e = slot[i];
                    if (pass == 0) T->pp[e + 1]++;
                    else { int64_t t = cur[e]++; T->pa[t] = rpos[q]; T->pb[t] = (int32_t)p; }
                }
            }
        }
    }
    if (Pp) {
        T->pnz = Pp[n];
        T->pslot.assign((size_t)T->pnz, -1);
        for (int64_t j = 0; j < n; j++) {
            for (int64_t e = T->Sp[j]; e < T->Sp[j + 1]; e++) slot[T->Si[e]] = e;
            for (int64_t p = Pp[j]; p < Pp[j + 1]; p++)
                if (Pi[p] >= j) T->pslot[p] = slot[Pi[p]];
        }
        // entries of P above the diagonal are ignored (lower triangle is read, as for 'L' storage)
        for (int64_t p = 0; p < T->pnz; p++)
            if (T->pslot[p] < 0) T->pslot[p] = 0;   // neutralised below by a zero value
    }
    *out = hold.release();
    return KVX_OK;
}

int kvx_atda_plan(int64_t ml, int64_t n, const int64_t *Gp, const int64_t *Gi, const int64_t *Pp, const int64_t *Pi,
                  kvx_atda **out)
{
    return guarded([&] { return kvx_atda_plan_impl(ml, n, Gp, Gi, Pp, Pi, out); });
}

static int kvx_atda_pattern_impl(kvx_atda *T, int64_t *snz, int64_t *Sp, int64_t *Si)
{
    if (!T) return KVX_EINVAL;
    if (snz) *snz = T->snz;
    if (Sp) memcpy(Sp, T->Sp.data(), sizeof(int64_t) * (T->n + 1));
    if (Si && T->snz) memcpy(Si, T->Si.data(), sizeof(int64_t) * T->snz);
    return KVX_OK;
}

int kvx_atda_pattern(kvx_atda *T, int64_t *snz, int64_t *Sp, int64_t *Si)
{
    return guarded([&] { return kvx_atda_pattern_impl(T, snz, Sp, Si); });
}

int kvx_atda_assemble_dev(kvx_atda *T, const double *Gx, const double *w, const double *Px, double *Sx)
{
    if (!T) return KVX_EINVAL;
    int rc = atda_device(T);
    if (rc) return rc;
    launch_atda(nullptr, T->snz, T->gnz, T->d_pp, T->d_pa, T->d_pb, T->d_gi, Gx, w, T->d_wg, Sx);
    if (Px && T->pnz > 0) launch_add_at(nullptr, T->pnz, T->d_pslot, Px, Sx);
    HIPCHK(hipGetLastError());
    return KVX_OK;
}
// the same with the weights given by their square roots: S = G' diag(di)^2 G (+ P) -- what misc.kkt_chol2 forms (misc.py:1418-1426);
// the square is taken while G is scaled (one launch less than ssqr + assemble, the same roundings)
int kvx_atda_assemble_sq_dev(kvx_atda *T, const double *Gx, const double *di, const double *Px, double *Sx)
{
    if (!T) return KVX_EINVAL;
    int rc = atda_device(T);
    if (rc) return rc;
    launch_atda(nullptr, T->snz, T->gnz, T->d_pp, T->d_pa, T->d_pb, T->d_gi, Gx, di, T->d_wg, Sx, true);
    if (Px && T->pnz > 0) launch_add_at(nullptr, T->pnz, T->d_pslot, Px, Sx);
    HIPCHK(hipGetLastError());
    return KVX_OK;
}

static int kvx_atda_assemble_impl(kvx_atda *T, const double *Gx, const double *w, const double *Px, double *Sx)
{
    if (!T) return KVX_EINVAL;
    int rc = atda_device(T);
    if (rc) return rc;
    if (!T->d_gx) {
        HIPCHK(pool_malloc((void **)&T->d_gx, std::max<int64_t>(T->gnz, 1) * sizeof(double)));
        HIPCHK(pool_malloc((void **)&T->d_w, std::max<int64_t>(T->ml, 1) * sizeof(double)));
        HIPCHK(pool_malloc((void **)&T->d_px, std::max<int64_t>(T->pnz, 1) * sizeof(double)));
        HIPCHK(pool_malloc((void **)&T->d_sx, std::max<int64_t>(T->snz, 1) * sizeof(double)));
    }
    if (T->gnz) HIPCHK(hipMemcpy(T->d_gx, Gx, T->gnz * sizeof(double), hipMemcpyHostToDevice));
    if (T->ml) HIPCHK(hipMemcpy(T->d_w, w, T->ml * sizeof(double), hipMemcpyHostToDevice));
    std::vector<double> ptmp;
    if (Px && T->pnz) {
        HIPCHK(hipMemcpy(T->d_px, Px, T->pnz * sizeof(double), hipMemcpyHostToDevice));
    }
    rc = kvx_atda_assemble_dev(T, T->d_gx, T->d_w, (Px && T->pnz) ? T->d_px : nullptr, T->d_sx);
    if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());
    if (T->snz) HIPCHK(hipMemcpy(Sx, T->d_sx, T->snz * sizeof(double), hipMemcpyDeviceToHost));
    return KVX_OK;
}

int kvx_atda_assemble(kvx_atda *T, const double *Gx, const double *w, const double *Px, double *Sx)
{
    return guarded([&] { return kvx_atda_assemble_impl(T, Gx, w, Px, Sx); });
}

void kvx_atda_free(kvx_atda *T)
{
    if (!T) return;
    void *ptrs[] = {T->d_pp, T->d_pslot, T->d_pa, T->d_pb, T->d_gi, T->d_gx, T->d_w, T->d_px, T->d_sx, T->d_wg};
    for (void *p : ptrs)
        if (p) (void)pool_free(p);
    delete T;
}

// ---- NT scaling ------------------------------------------------------------------------------------
int kvx_nt_compute_scaling_dev(int64_t ml, const double *s, const double *z, double *d, double *di, double *lmbda)
{ launch_compute_scaling(nullptr, ml, s, z, d, di, lmbda); HIPCHK(hipGetLastError()); return KVX_OK; }
int kvx_nt_update_scaling_dev(int64_t ml, double *s, double *z, double *d, double *di, double *lmbda)
{ launch_update_scaling(nullptr, ml, s, z, d, di, lmbda); HIPCHK(hipGetLastError()); return KVX_OK; }
int kvx_lp_newton_rhs_dev(int64_t ml, const double *lmbdasq, const double *ws3, double shift, double scale, const double *rz,
                          const double *lmbda, const double *d, double *ds, double *dz)
{ launch_lp_newton_rhs(nullptr, ml, lmbdasq, ws3, shift, scale, rz, lmbda, d, ds, dz); HIPCHK(hipGetLastError()); return KVX_OK; }
int kvx_lp_step_post_dev(int64_t ml, double dtau, const double *z1, const double *lmbda, double *ds, double *dz, double *ws3)
{ launch_lp_step_post(nullptr, ml, dtau, z1, lmbda, ds, dz, ws3); HIPCHK(hipGetLastError()); return KVX_OK; }
int kvx_lp_update_dev(int64_t ml, double step, double *ds, double *dz, double *d, double *di, double *lmbda, double *s, double *z)
{ launch_lp_update(nullptr, ml, step, ds, dz, d, di, lmbda, s, z); HIPCHK(hipGetLastError()); return KVX_OK; }
int kvx_nt_scale_dev(int64_t ml, int64_t ncols, int64_t ldx, double *x, const double *w)
{ launch_scale(nullptr, ml, ncols, ldx, x, w); HIPCHK(hipGetLastError()); return KVX_OK; }
int kvx_nt_scale2_dev(int64_t ml, const double *lmbda, double *x, int inverse)
{ if (inverse) launch_mul(nullptr, ml, x, lmbda); else launch_div(nullptr, ml, x, lmbda); HIPCHK(hipGetLastError()); return KVX_OK; }
int kvx_nt_sprod_dev(int64_t ml, double *x, const double *y)
{ launch_mul(nullptr, ml, x, y); HIPCHK(hipGetLastError()); return KVX_OK; }
int kvx_nt_sinv_dev(int64_t ml, double *x, const double *y)
{ launch_div(nullptr, ml, x, y); HIPCHK(hipGetLastError()); return KVX_OK; }
int kvx_nt_ssqr_dev(int64_t ml, double *x, const double *y)
{ launch_sqr(nullptr, ml, x, y); HIPCHK(hipGetLastError()); return KVX_OK; }

// Second stage of the fixed reduction tree on the host: the arithmetic of k_reduce2 (64 lanes, each over its strided partial
// results in ascending order, then the xor butterfly 32, 16, ..., 1; lane 0's value) -- the same bits as the device second stage.
static double host_reduce2(const double *part, bool mx)
{
    const int nb = reduce_blocks();
    double v[64], t[64];
    for (int l = 0; l < 64; l++) {
        double acc = mx ? -1.7976931348623157e308 : 0.0;
        for (int i = l; i < nb; i += 64) acc = mx ? std::fmax(acc, part[i]) : acc + part[i];
        v[l] = acc;
    }
    for (int o = 32; o > 0; o >>= 1) {
        for (int l = 0; l < 64; l++) t[l] = mx ? std::fmax(v[l], v[l ^ o]) : v[l] + v[l ^ o];
        std::memcpy(v, t, sizeof(v));
    }
    return v[0];
}
static MultiRed one_reduction(int kind, int64_t n, const double *x, const double *y)
{
    MultiRed mr;
    mr.count = 1;
    for (int i = 0; i < 32; i++) { mr.kind[i] = 0; mr.n[i] = 0; mr.x[i] = nullptr; mr.y[i] = nullptr; }
    mr.kind[0] = kind; mr.n[0] = n; mr.x[0] = x; mr.y[0] = y;
    return mr;
}
int kvx_nt_sdot_dev(int64_t ml, const double *x, const double *y, double *result_host)
{
    if (!result_host) return KVX_EINVAL;
    int rc = ensure_scratch();
    if (rc) return rc;
    Scratch &s = scratch();
    if (!lp_unfused()) {
        launch_reduce_multi_stage1(nullptr, one_reduction(0, ml, x, y), s.part);
        HIPCHK(hipMemcpy(s.host, s.part, reduce_blocks() * sizeof(double), hipMemcpyDeviceToHost));
        *result_host = host_reduce2(s.host, false);
        return KVX_OK;
    }
    launch_dot(nullptr, ml, x, y, s.part, s.part + reduce_scratch_doubles() - 1);
    HIPCHK(hipMemcpy(s.host, s.part + reduce_scratch_doubles() - 1, sizeof(double), hipMemcpyDeviceToHost));
    *result_host = *s.host;
    return KVX_OK;
}
int kvx_nt_max_step_dev(int64_t ml, const double *x, double *result_host)
{
    if (!result_host) return KVX_EINVAL;
    int rc = ensure_scratch();
    if (rc) return rc;
    Scratch &s = scratch();
    if (!lp_unfused()) {
        launch_reduce_multi_stage1(nullptr, one_reduction(1, ml, x, x), s.part);
        HIPCHK(hipMemcpy(s.host, s.part, reduce_blocks() * sizeof(double), hipMemcpyDeviceToHost));
        *result_host = host_reduce2(s.host, true);
        return KVX_OK;
    }
    launch_maxneg(nullptr, ml, x, s.part, s.part + reduce_scratch_doubles() - 1);
    HIPCHK(hipMemcpy(s.host, s.part + reduce_scratch_doubles() - 1, sizeof(double), hipMemcpyDeviceToHost));
    *result_host = *s.host;
    return KVX_OK;
}

// Several reductions, ONE host synchronisation: the interior-point loop needs 6-9 norms and inner products at the
// same point of every iteration (coneprog.py:861-896), each of which would otherwise stall the GPU for a round trip.
// kind[i] = 0: sum_j x_i[j] * y_i[j] (sdot);  1: max_j(-x_i[j]) (max_step, 'l' block).  Same kernels, same fixed
// reduction tree as the single calls, so the values are bitwise those of kvx_nt_sdot_dev / kvx_nt_max_step_dev.
int kvx_nt_reduce_multi_dev(int count, const int32_t *kind, const int64_t *n, const double *const *x,
                            const double *const *y, double *out_host)
{
    if (count < 0 || count > 32 || (count > 0 && (!kind || !n || !x || !out_host))) return KVX_EINVAL;
    if (count == 0) return KVX_OK;
    int rc = ensure_scratch();
    if (rc) return rc;
    Scratch &s = scratch();
    MultiRed mr;
    mr.count = count;
    for (int i = 0; i < 32; i++) { mr.kind[i] = 0; mr.n[i] = 0; mr.x[i] = nullptr; mr.y[i] = nullptr; }
    for (int i = 0; i < count; i++) {
        if (kind[i] != 0 && kind[i] != 1) return KVX_EINVAL;
        if (kind[i] == 0 && (!y || !y[i])) return KVX_EINVAL;
        mr.kind[i] = kind[i];
        mr.n[i] = n[i];
        mr.x[i] = x[i];
        mr.y[i] = kind[i] == 0 ? y[i] : x[i];
    }
    if (!lp_unfused()) {                                            // one launch; the second stage on the host
        launch_reduce_multi_stage1(nullptr, mr, s.part);
        HIPCHK(hipMemcpy(s.host, s.part, (size_t)count * reduce_blocks() * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < count; i++) out_host[i] = host_reduce2(s.host + (size_t)i * reduce_blocks(), kind[i] != 0);
        return KVX_OK;
    }
    launch_reduce_multi(nullptr, mr, s.part, s.multi);              // all of them in two launches
    HIPCHK(hipMemcpy(s.host, s.multi, (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < count; i++) out_host[i] = s.host[i];
    return KVX_OK;
}

// Second half of f6_no_ir for the orthant cone in ONE host round trip (kkt.hip, k_lp_dtau): returns dtau, z1'z1 and the
// step bounds max(-ds ./ lmbda), max(-dz ./ lmbda) of coneprog.py:1316-1321.  y-blocks may be empty (p = 0: b, dy, y1 NULL).
// z1z1 < 0: compute z1'z1 in the same reduction (first call of an iteration), else the value of that call.
int kvx_lp_second_half_dev(int64_t ml, int64_t n, int64_t p, const double *c, const double *b, const double *th, const double *x1,
                           const double *y1, const double *z1, const double *lmbda, double *dx, double *dy, double *dz, double *ds,
                           double *ws3, double dgi, double dtau0, double z1z1, double out_host[4])
{
    if (ml < 0 || n < 0 || p < 0 || !out_host) return KVX_EINVAL;
    int rc = ensure_scratch();
    if (rc) return rc;
    Scratch &s = scratch();
    double *r = s.multi, *sc = s.multi + 8, *mx = s.multi + 16;
    if (!lp_unfused()) {
        LpHalf a;
        a.ml = ml; a.n = n; a.p = p;
        a.c = c; a.dx = dx; a.b = b; a.dy = dy; a.th = th; a.dz = dz; a.z1 = z1; a.x1 = x1; a.y1 = y1; a.lm = lmbda;
        a.dxw = dx; a.dyw = dy; a.dsw = ds; a.dzw = dz; a.ws3 = ws3;
        const int nb = reduce_blocks();
        double *part2 = s.part + 4 * nb;                            // 2 nb partial maxima, then (dtau, z1'z1)
        launch_lp_second_half(nullptr, a, dgi, dtau0, z1z1, s.part, part2);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(s.host, part2, (2 * nb + 2) * sizeof(double), hipMemcpyDeviceToHost));
        out_host[0] = s.host[2 * nb]; out_host[1] = s.host[2 * nb + 1];
        out_host[2] = host_reduce2(s.host, true); out_host[3] = host_reduce2(s.host + nb, true);
        return KVX_OK;
    }
    launch_dot(nullptr, n, c, dx, s.part, r + 0);
    if (p > 0) launch_dot(nullptr, p, b, dy, s.part, r + 1);
    else HIPCHK(hipMemsetAsync(r + 1, 0, sizeof(double), nullptr));
    launch_dot(nullptr, ml, th, dz, s.part, r + 2);
    if (z1z1 < 0.0) launch_dot(nullptr, ml, z1, z1, s.part, r + 3);
    launch_lp_dtau(nullptr, r, dgi, dtau0, z1z1, z1z1 < 0.0 ? 0 : 1, sc);
    launch_axpy_devalpha(nullptr, n, sc, x1, dx);
    if (p > 0) launch_axpy_devalpha(nullptr, p, sc, y1, dy);
    launch_lp_step_post_devalpha(nullptr, ml, sc, z1, lmbda, ds, dz, ws3);
    launch_maxneg(nullptr, ml, ds, s.part, mx + 0);
    launch_maxneg(nullptr, ml, dz, s.part, mx + 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(s.host, s.multi, 18 * sizeof(double), hipMemcpyDeviceToHost));
    out_host[0] = s.host[8]; out_host[1] = s.host[9]; out_host[2] = s.host[16]; out_host[3] = s.host[17];
    return KVX_OK;
}

// ---- round 3: the rest of an interior-point iteration in few launches (kkt.hip, "round 3") ----
// KKT solve with the factor of S, the parts around the triangular solves (misc.py:1489-1563 with p = 0) for nrhs = 1 or 2 sides:
//   pre :  x2(:,k) := xscale_k * xin_k + G' (di .* (zin_k .* di))
//   post:  xout_k := xoscale_k * x2(:,k) ;  zout_k := zoscale_k * (di .* (G x2(:,k)) - zin_k .* di)
// G by its CCS arrays (pre) and by the CCS arrays of its transpose (post), int64 indices on the device as for kvx_spmv_dev.
int kvx_kkt_solve_pre_dev(int64_t ml, int64_t n, const int64_t *Gp, const int64_t *Gi, const double *Gx, int64_t max_col_nnz,
                          const double *di, int nrhs, const kvx_kkt_side *sides, double *x2, int64_t ldx2)
{
    if (ml < 0 || n < 0 || nrhs < 1 || nrhs > 2 || !sides || !x2 || ldx2 < std::max<int64_t>(n, 1)) return KVX_EINVAL;
    KktSides r;
    for (int k = 0; k < 2; k++) {
        const kvx_kkt_side &q = sides[k < nrhs ? k : 0];
        r.s[k] = KktSide{q.xin, q.xscale, q.zin, q.xout, q.xoscale, q.zout, q.zoscale};
    }
    launch_kkt_pre(nullptr, n, Gp, Gi, Gx, di, nrhs, r, x2, ldx2, max_col_nnz);
    HIPCHK(hipGetLastError());
    return KVX_OK;
}
int kvx_kkt_solve_post_dev(int64_t ml, int64_t n, const int64_t *GTp, const int64_t *GTi, const double *GTx, int64_t max_row_nnz,
                           const double *di, int nrhs, const kvx_kkt_side *sides, const double *x2, int64_t ldx2)
{
    if (ml < 0 || n < 0 || nrhs < 1 || nrhs > 2 || !sides || !x2 || ldx2 < std::max<int64_t>(n, 1)) return KVX_EINVAL;
    KktSides r;
    for (int k = 0; k < 2; k++) {
        const kvx_kkt_side &q = sides[k < nrhs ? k : 0];
        r.s[k] = KktSide{q.xin, q.xscale, q.zin, q.xout, q.xoscale, q.zout, q.zoscale};
    }
    launch_kkt_post(nullptr, ml, n, GTp, GTi, GTx, di, nrhs, r, x2, ldx2, max_row_nnz);
    HIPCHK(hipGetLastError());
    return KVX_OK;
}
// residuals of an iteration without equality constraints (coneprog.py:861-896): hrx := -G'z, rx := hrx - tau c, hrz := G x + s,
// rz := hrz - tau h in one launch
int kvx_lp_residuals_dev(int64_t ml, int64_t n, const int64_t *Gp, const int64_t *Gi, const double *Gx, int64_t max_col_nnz,
                         const int64_t *GTp, const int64_t *GTi, const double *GTx, int64_t max_row_nnz, const double *x, const double *z, const double *s,
                         const double *c, const double *h, double tau, double *hrx, double *rx, double *hrz, double *rz)
{
    if (ml < 0 || n < 0) return KVX_EINVAL;
    launch_lp_residuals(nullptr, ml, n, Gp, Gi, Gx, GTp, GTi, GTx, x, z, s, c, h, tau, hrx, rx, hrz, rz, max_col_nnz, max_row_nnz);
    HIPCHK(hipGetLastError());
    return KVX_OK;
}
// kvx_lp_update_dev and x += step dx in one launch
int kvx_lp_update_x_dev(int64_t ml, int64_t n, double step, double *ds, double *dz, double *d, double *di, double *lmbda, double *s,
                        double *z, const double *dx, double *x)
{
    if (ml < 0 || n < 0) return KVX_EINVAL;
    launch_lp_update_x(nullptr, ml, n, step, ds, dz, d, di, lmbda, s, z, dx, x);
    HIPCHK(hipGetLastError());
    return KVX_OK;
}

// ---- BLAS-1 glue of the interior-point loop on device vectors (blas.axpy/scal/copy/dot calls of coneprog.py) ----
int kvx_vec_axpy_dev(int64_t n, double alpha, const double *x, double *y)
{ launch_axpy(nullptr, n, alpha, x, y); HIPCHK(hipGetLastError()); return KVX_OK; }
int kvx_dense_gemv_dev(int64_t m, int64_t n, int64_t nrhs, double alpha, const double *A, int64_t lda, const double *x, int64_t ldx,
                       double beta, double *y, int64_t ldy)
{
    if (m < 0 || n < 0 || nrhs < 0 || lda < std::max<int64_t>(m, 1)) { kvx::set_last_error("kvx_dense_gemv_dev: bad dimensions"); return KVX_EINVAL; }
    launch_dense_gemv(nullptr, m, n, nrhs, alpha, A, lda, x, ldx, beta, y, ldy);
    HIPCHK(hipGetLastError());
    return KVX_OK;
}
int kvx_vec_lincomb_dev(int64_t n, double a, const double *x, double b, const double *y, double *z)
{ launch_lincomb(nullptr, n, a, x, b, y, z); HIPCHK(hipGetLastError()); return KVX_OK; }
int kvx_vec_scal_dev(int64_t n, double alpha, double *x)
{ launch_vscal(nullptr, n, alpha, x); HIPCHK(hipGetLastError()); return KVX_OK; }
int kvx_vec_addc_dev(int64_t n, double c, double *x)
{ launch_addc(nullptr, n, c, x); HIPCHK(hipGetLastError()); return KVX_OK; }
int kvx_vec_fill_dev(int64_t n, double c, double *x)
{ launch_fill(nullptr, n, c, x); HIPCHK(hipGetLastError()); return KVX_OK; }
int kvx_vec_copy_dev(int64_t n, const double *x, double *y)
{ if (n > 0) HIPCHK(hipMemcpyAsync(y, x, n * sizeof(double), hipMemcpyDeviceToDevice, nullptr)); return KVX_OK; }
int kvx_vec_copy_strided_dev(int64_t n, const double *x, int64_t incx, double *y)     /* y[i] := x[i * incx] */
{
    if (n > 0 && incx > 0)
        HIPCHK(hipMemcpy2DAsync(y, sizeof(double), x, (size_t)incx * sizeof(double), sizeof(double), (size_t)n, hipMemcpyDeviceToDevice, nullptr));
    return KVX_OK;
}
int kvx_vec_xmy_dev(int64_t n, double a, const double *x, const double *y, double b, double *z)
{ launch_xmy(nullptr, n, a, x, y, b, z); HIPCHK(hipGetLastError()); return KVX_OK; }

int kvx_spmv_dev(int trans, int64_t m, int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax, double alpha,
                 const double *x, double beta, double *y)
{
    if (trans != 'N' && trans != 'T') return KVX_EINVAL;
    launch_spmv(nullptr, trans, m, n, Ap, Ai, Ax, alpha, x, beta, y);
    HIPCHK(hipGetLastError());
    return KVX_OK;
}

int kvx_spmm_t_dev(int64_t n, int64_t ncols, const int64_t *Ap, const int64_t *Ai, const double *Ax, const double *X, int64_t ldx,
                   double *Y, int64_t ldy)
{
    if (n < 0 || ncols < 0 || ncols > 65535) return KVX_EINVAL;
    launch_spmm_t(nullptr, n, ncols, Ap, Ai, Ax, X, ldx, Y, ldy);
    HIPCHK(hipGetLastError());
    return KVX_OK;
}

int kvx_dense_from_ccs_dev(int64_t m, int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax, double *D, int64_t ld)
{
    if (m < 0 || n < 0 || ld < std::max<int64_t>(1, m)) return KVX_EINVAL;
    if (m > 0 && n > 0) HIPCHK(hipMemset2DAsync(D, (size_t)ld * sizeof(double), 0, (size_t)m * sizeof(double), (size_t)n, nullptr));
    launch_dense_from_ccs(nullptr, n, Ap, Ai, Ax, D, ld);
    HIPCHK(hipGetLastError());
    return KVX_OK;
}

int kvx_pack_lower_dev(int64_t p, const double *K, int64_t ld, double *out)
{
    if (p < 0 || p > 65535 || ld < std::max<int64_t>(1, p)) return KVX_EINVAL;
    launch_pack_lower(nullptr, p, K, ld, out);
    HIPCHK(hipGetLastError());
    return KVX_OK;
}

// ---- round 4: an interior-point iteration in four calls (include/kvxhip.h, kvx_lp_ctx) ---------------------------------
// The calls above, in the order lp.py made them, from C: no arithmetic of their own.
static int lp_stats(const kvx_lp_ctx *L, double out[10])
{
    const int32_t kind[7] = {0, 0, 0, 0, 0, 0, 0};
    const int64_t nn[7] = {L->n, L->n, L->ml, L->ml, L->n, L->ml, L->ml};
    const double *xs[7] = {L->hrx, L->rx, L->hrz, L->rz, L->c, L->h, L->lmbda};
    const double *ys[7] = {L->hrx, L->rx, L->hrz, L->rz, L->x, L->z, L->lmbda};
    double r[7];
    int rc = kvx_nt_reduce_multi_dev(7, kind, nn, xs, ys, r);
    if (rc) return rc;
    out[0] = r[0]; out[1] = r[1]; out[2] = 0.0; out[3] = 0.0; out[4] = r[2]; out[5] = r[3]; out[6] = r[4]; out[7] = 0.0; out[8] = r[5]; out[9] = r[6];
    return KVX_OK;
}

int kvx_lp_iter_residuals(const kvx_lp_ctx *L, double tau, double out[10])
{
    if (!L || !out) return KVX_EINVAL;
    int rc = kvx_lp_residuals_dev(L->ml, L->n, L->Gp, L->Gi, L->Gx, L->max_col, L->GTp, L->GTi, L->GTx, L->max_row, L->x, L->z, L->s, L->c, L->h,
                                  tau, L->hrx, L->rx, L->hrz, L->rz);
    if (rc) return rc;
    return lp_stats(L, out);
}

static int lp_kkt(const kvx_lp_ctx *L, int nrhs, const kvx_kkt_side *sides, bool with_factor)
{
    const int64_t ld = std::max<int64_t>(1, L->n);
    int rc;
    if (with_factor && (rc = kvx_atda_assemble_sq_dev(L->plan, L->Gx, L->di, nullptr, L->Sx))) return rc;
    if ((rc = kvx_kkt_solve_pre_dev(L->ml, L->n, L->Gp, L->Gi, L->Gx, L->max_col, L->di, nrhs, sides, L->x2, ld))) return rc;
    if (with_factor) rc = kvx_chol_factorize_solve_async_dev(L->F, L->Sx, L->x2, nrhs, ld);
    else rc = kvx_chol_solve_async_dev(L->F, 0, L->x2, nrhs, ld);
    if (rc) return rc;
    return kvx_kkt_solve_post_dev(L->ml, L->n, L->GTp, L->GTi, L->GTx, L->max_row, L->di, nrhs, sides, L->x2, ld);
}

int kvx_lp_iter_predictor(const kvx_lp_ctx *L, double dgi, double dtau0, double out[4])
{
    if (!L || !out) return KVX_EINVAL;
    int rc = kvx_lp_newton_rhs_dev(L->ml, nullptr, nullptr, 0.0, 1.0, L->rz, L->lmbda, L->d, L->ds, L->dz);
    if (rc) return rc;
    // (x1, z1) := dgi * K^-1 (-c, h) and the predictor's (dx, dz) := K^-1 (rx, dz): two right-hand sides, one factorisation
    const kvx_kkt_side sides[2] = {{L->c, -1.0, L->h, L->x1, dgi, L->z1, dgi}, {L->rx, 1.0, L->dz, L->dx, 1.0, L->dz, 1.0}};
    if ((rc = lp_kkt(L, 2, sides, true))) return rc;
    if ((rc = kvx_vec_xmy_dev(L->ml, 1.0, L->h, L->di, 0.0, L->th))) return rc;
    return kvx_lp_second_half_dev(L->ml, L->n, 0, L->c, nullptr, L->th, L->x1, nullptr, L->z1, L->lmbda, L->dx, nullptr, L->dz, L->ds, L->ws3, dgi,
                                  dtau0, -1.0, out);
}

int kvx_lp_iter_corrector(const kvx_lp_ctx *L, double shift, double scale, double dgi, double dtau0, double z1z1, double out[4])
{
    if (!L || !out) return KVX_EINVAL;
    int rc = kvx_lp_newton_rhs_dev(L->ml, nullptr, L->ws3, shift, scale, L->rz, L->lmbda, L->d, L->ds, L->dz);
    if (rc) return rc;
    const kvx_kkt_side sides[2] = {{L->rx, scale, L->dz, L->dx, 1.0, L->dz, 1.0}, {L->rx, scale, L->dz, L->dx, 1.0, L->dz, 1.0}};
    if ((rc = lp_kkt(L, 1, sides, false))) return rc;
    return kvx_lp_second_half_dev(L->ml, L->n, 0, L->c, nullptr, L->th, L->x1, nullptr, L->z1, L->lmbda, L->dx, nullptr, L->dz, L->ds, nullptr, dgi,
                                  dtau0, z1z1, out);
}

int kvx_lp_iter_update(const kvx_lp_ctx *L, double step, double tau_next, double out[10])
{
    if (!L || !out) return KVX_EINVAL;
    int rc = kvx_lp_update_x_dev(L->ml, L->n, step, L->ds, L->dz, L->d, L->di, L->lmbda, L->s, L->z, L->dx, L->x);
    if (rc) return rc;
    return kvx_lp_iter_residuals(L, tau_next, out);
}

}  // extern "C"
