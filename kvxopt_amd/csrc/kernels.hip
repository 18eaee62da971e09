// HIP kernels (gfx950 / CDNA4) for the supernodal multifrontal Cholesky factorisation and the
// level-scheduled triangular solves.
//
// Replaces the arithmetic the reference obtains from cholmod_l_factorize / cholmod_l_solve
// (reference call sites src/C/cholmod.c:362, :483, :677, :735).  Design (own, MI355X-first):
//   * fronts (supernodes) of one elimination-tree level are independent -> one launch per
//     level and size class, one workgroup per front;
//   * a front of order m <= 128 is factored on chip (kernels_wave.hip): registers + an LDS image,
//     every HBM byte of the front touched once;
//   * larger fronts run a blocked right-looking factorisation in HBM/L2 (kernels_big.hip)
//     whose panel solve and trailing update are FP64 MFMA (v_mfma_f64_16x16x4_f64);
//   * extend-add is parent-pull (each workgroup owns target columns), so there are no
//     atomics and the result is bitwise reproducible.
#include "device.hpp"

#include <algorithm>

namespace kvx {

typedef double d4 __attribute__((ext_vector_type(4)));

#define KVX_LAUNCH_CHECK() (void)0

// ------------------------------------------------------------------------------------------
// scatter the caller's values into the (zeroed) panels
__global__ void k_scatter_a(const double *__restrict__ Ax, const int64_t *__restrict__ amap, int64_t nnz,
                            double *__restrict__ Lx)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < nnz; i += stride) {
        int64_t d = amap[i];
        if (d >= 0) Lx[d] = Ax[i];
    }
}

// zero the panels and reset the status word: plain kernels instead of memset nodes (see enqueue_factor_body)
__global__ __launch_bounds__(256) void k_clear_factor(double2 *__restrict__ Lx2, int64_t n2, double *__restrict__ Lx, int64_t n, int *status)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (i == 0) *status = 0x7f7f7f7f;
    if (i == 0 && (n & 1)) Lx[n - 1] = 0.0;
    for (; i < n2; i += stride) Lx2[i] = make_double2(0.0, 0.0);
}
void launch_clear_factor(hipStream_t st, double *Lx, int64_t n, int *status)
{
    const int64_t n2 = n / 2;                       // (pool blocks are 256-byte aligned)
    int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((n2 + 255) / 256, 8192));
    hipLaunchKernelGGL(k_clear_factor, dim3((unsigned)blocks), dim3(256), 0, st, (double2 *)Lx, n2, Lx, n, status);
}

// the status word of a factorisation into pinned host memory (device-visible mapping): the last node of the captured factorisation
// is a kernel too, not a device-to-host memcpy node
__global__ void k_publish_status(const int *__restrict__ d_status, int *host_status)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        *host_status = *d_status;
        __threadfence_system();
    }
}
void launch_publish_status(hipStream_t st, const int *d_status, int *host_status_dev)
{
    hipLaunchKernelGGL(k_publish_status, dim3(1), dim3(64), 0, st, d_status, host_status_dev);
}

// dst := src (n doubles, both 16-byte aligned): a kernel instead of a memcpy node inside the captured solve sweeps
__global__ __launch_bounds__(256) void k_copy_d(double2 *__restrict__ dst2, const double2 *__restrict__ src2, int64_t n2,
                                                double *__restrict__ dst, const double *__restrict__ src, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (i == 0 && (n & 1)) dst[n - 1] = src[n - 1];
    for (; i < n2; i += stride) dst2[i] = src2[i];
}
void launch_copy_d(hipStream_t st, double *dst, const double *src, int64_t n)
{
    if (n <= 0) return;
    const int64_t n2 = n / 2;
    int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((n2 + 255) / 256, 8192));
    hipLaunchKernelGGL(k_copy_d, dim3((unsigned)blocks), dim3(256), 0, st, (double2 *)dst, (const double2 *)src, n2, dst, src, n);
}

// Zero the factor AND scatter the caller's values in ONE pass over L.  A's entries land in nearly every cache line of the small
// fronts' panels (three entries per column of 20-60 rows on the 5-point system), so the scatter behind the zeroing was a second
// full read-modify-write pass over L -- 64 us behind the 63 us of the zeroing on config 2.  Here a workgroup owns a chunk of
// KVX_INIT_CHUNK doubles of L (16 KB; 4 / 8 / 16 / 32 / 64 KB measured: 101 / 105 / 87-95 / 91 / 84 us): zeros in LDS, the chunk's entries on top (the scatter map grouped by chunk on the host, api.cpp),
// one coalesced store of the chunk.  The gather from the caller's value array (24 MB, cached) is the cheap side.
#ifndef KVX_INIT_SHIFT_V
#define KVX_INIT_SHIFT_V 11
#endif
constexpr int KVX_INIT_SHIFT = KVX_INIT_SHIFT_V, KVX_INIT_CHUNK = 1 << KVX_INIT_SHIFT, KVX_INIT_NT = KVX_INIT_CHUNK / 8;
__global__ __launch_bounds__(KVX_INIT_NT) void k_init_factor(const double *__restrict__ Ax, const int32_t *__restrict__ src,
                                                     const int64_t *__restrict__ dst, const int64_t *__restrict__ cptr, int64_t lsize,
                                                     double *__restrict__ Lx, int *status)
{
    __shared__ double t[KVX_INIT_CHUNK];
    const int64_t base = (int64_t)blockIdx.x << KVX_INIT_SHIFT;
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0) *status = 0x7f7f7f7f;
    const int64_t e0 = cptr[blockIdx.x], e1 = cptr[blockIdx.x + 1];
#pragma unroll
    for (int i = tid; i < KVX_INIT_CHUNK; i += KVX_INIT_NT) t[i] = 0.0;
    __syncthreads();
    for (int64_t e = e0 + tid; e < e1; e += KVX_INIT_NT) t[dst[e] & (KVX_INIT_CHUNK - 1)] = Ax[src[e]];
    __syncthreads();
    if (base + KVX_INIT_CHUNK <= lsize) {                                        // (pool blocks are 256-byte aligned)
        double2 *o = (double2 *)(Lx + base);
#pragma unroll
        for (int i = tid; i < KVX_INIT_CHUNK / 2; i += KVX_INIT_NT) o[i] = make_double2(t[2 * i], t[2 * i + 1]);
    } else {
        for (int i = tid; base + i < lsize; i += KVX_INIT_NT) Lx[base + i] = t[i];
    }
}
int init_factor_shift() { return KVX_INIT_SHIFT; }
// grouping of the scatter map by chunk on the device: cnt[chunk + 1] += 1 per live entry; after the host's scan the cursors hand out
// the places (the order inside a chunk does not matter: destinations are distinct)
__global__ void k_scatter_group_count(const int64_t *__restrict__ amap, int64_t nnz, int sh, unsigned long long *__restrict__ cnt)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t d = amap[e];
        if (d >= 0) atomicAdd(&cnt[(d >> sh) + 1], 1ull);
    }
}
__global__ void k_scatter_group_place(const int64_t *__restrict__ amap, int64_t nnz, int sh, unsigned long long *__restrict__ cursor,
                                      int64_t *__restrict__ sdst, int32_t *__restrict__ ssrc)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t d = amap[e];
        if (d < 0) continue;
        const unsigned long long q = atomicAdd(&cursor[d >> sh], 1ull);
        sdst[q] = d;
        ssrc[q] = (int32_t)e;
    }
}
void launch_scatter_group_count(hipStream_t st, const int64_t *amap, int64_t nnz, int sh, int64_t *cnt)
{
    if (nnz <= 0) return;
    hipLaunchKernelGGL(k_scatter_group_count, dim3((unsigned)std::min<int64_t>((nnz + 255) / 256, 8192)), dim3(256), 0, st, amap, nnz, sh,
                       (unsigned long long *)cnt);
}
void launch_scatter_group_place(hipStream_t st, const int64_t *amap, int64_t nnz, int sh, int64_t *cursor, int64_t *sdst, int32_t *ssrc)
{
    if (nnz <= 0) return;
    hipLaunchKernelGGL(k_scatter_group_place, dim3((unsigned)std::min<int64_t>((nnz + 255) / 256, 8192)), dim3(256), 0, st, amap, nnz, sh,
                       (unsigned long long *)cursor, sdst, ssrc);
}
void launch_init_factor(hipStream_t st, const double *Ax, const int32_t *src, const int64_t *dst, const int64_t *cptr, int64_t lsize,
                        double *Lx, int *status)
{
    const int64_t nchunk = std::max<int64_t>((lsize + KVX_INIT_CHUNK - 1) >> KVX_INIT_SHIFT, 1);
    hipLaunchKernelGGL(k_init_factor, dim3((unsigned)nchunk), dim3(KVX_INIT_NT), 0, st, Ax, src, dst, cptr, lsize, Lx, status);
}

void launch_scatter_a(hipStream_t st, const double *Ax, const int64_t *amap, int64_t nnz, double *Lx)
{
    if (nnz <= 0) return;
    int64_t blocks = (nnz + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_scatter_a, dim3((unsigned)blocks), dim3(256), 0, st, Ax, amap, nnz, Lx);
}

// ------------------------------------------------------------------------------------------
// Triangular solves, one workgroup per front and right-hand side.
constexpr int SOLVE_B = 64, SOLVE_LD = 65;

template <int NT>
__global__ __launch_bounds__(NT) void k_fwd_level(DevSym ds, const int32_t *__restrict__ list,
                                                  const double *__restrict__ Lx, double *__restrict__ X,
                                                  int64_t ldx, const double *__restrict__ Wc,
                                                  double *__restrict__ Wo, int64_t wstride, int wcap)
{
    unsigned fi, rh;
    kvx_front_rhs(fi, rh);
    extern __shared__ double sm[];
    double *w = sm, *D = sm + wcap;
    const int s = list[fi];
    const int k = ds.k[s], m = ds.m[s], u = m - k, f = ds.first[s], tid = threadIdx.x;
    double *x = X + (int64_t)rh * ldx;
    const double *wc = Wc + (int64_t)rh * wstride;
    double *wo = Wo + (int64_t)rh * wstride;
    const double *P = Lx + ds.px[s];
    for (int i = tid; i < m; i += NT) w[i] = (i < k) ? x[f + i] : 0.0;
    __syncthreads();
    for (int64_t c = ds.childptr[s]; c < ds.childptr[s + 1]; c++) {
        const int ch = ds.children[c];
        const int kc = ds.k[ch], uc = ds.m[ch] - kc;
        if (uc == 0) continue;
        const int32_t *rl = ds.rel + ds.rowptr[ch] + kc;
        const double *src = wc + ds.wx[ch];
        for (int i = tid; i < uc; i += NT) w[rl[i]] += src[i];
        __syncthreads();
    }
    for (int jb = 0; jb < k; jb += SOLVE_B) {
        const int nbk = min(SOLVE_B, k - jb);
        for (int idx = tid; idx < nbk * nbk; idx += NT) {
            const int j = idx / nbk, i = idx - j * nbk;
            D[i * SOLVE_LD + j] = P[(jb + i) + (int64_t)(jb + j) * m];
        }
        __syncthreads();
        if (tid < 64) {
            const int l = tid;
            double wl = (l < nbk) ? w[jb + l] : 0.0;
            for (int j = 0; j < nbk; j++) {
                const double yj = __shfl(wl, j) / D[j * SOLVE_LD + j];
                if (l == j) wl = yj;
                else if (l > j && l < nbk) wl -= D[l * SOLVE_LD + j] * yj;
            }
            if (l < nbk) w[jb + l] = wl;
        }
        __syncthreads();
        for (int i = jb + nbk + tid; i < m; i += NT) {
            const double *Pi = P + i + (int64_t)jb * m;
            double acc = 0.0;
            for (int j = 0; j < nbk; j++) acc += Pi[(int64_t)j * m] * w[jb + j];
            w[i] -= acc;
        }
        __syncthreads();
    }
    for (int i = tid; i < k; i += NT) x[f + i] = w[i];
    if (u > 0) {
        double *dst = wo + ds.wx[s];
        for (int i = tid; i < u; i += NT) dst[i] = w[k + i];
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void k_bwd_level(DevSym ds, const int32_t *__restrict__ list,
                                                  const double *__restrict__ Lx, double *__restrict__ X,
                                                  int64_t ldx, int wcap)
{
    unsigned fi, rh;
    kvx_front_rhs(fi, rh);
    extern __shared__ double sm[];
    double *xf = sm, *D = sm + wcap;
    const int s = list[fi];
    const int k = ds.k[s], m = ds.m[s], f = ds.first[s], tid = threadIdx.x;
    double *x = X + (int64_t)rh * ldx;
    const double *P = Lx + ds.px[s];
    const int32_t *rows = ds.rowidx + ds.rowptr[s];
    for (int i = tid; i < m; i += NT) xf[i] = (i < k) ? x[f + i] : x[rows[i]];
    __syncthreads();
    const int nblk = (k + SOLVE_B - 1) / SOLVE_B;
    const int wv = tid >> 6, ln = tid & 63;
    for (int b = nblk - 1; b >= 0; b--) {
        const int jb = b * SOLVE_B;
        const int nbk = min(SOLVE_B, k - jb);
        for (int idx = tid; idx < nbk * nbk; idx += NT) {
            const int j = idx / nbk, i = idx - j * nbk;
            D[i * SOLVE_LD + j] = P[(jb + i) + (int64_t)(jb + j) * m];
        }
        for (int j = wv; j < nbk; j += NT / 64) {
            const double *Pc = P + (int64_t)(jb + j) * m;
            double acc = 0.0;
            for (int i = jb + nbk + ln; i < m; i += 64) acc += Pc[i] * xf[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if (ln == 0) xf[jb + j] -= acc;
        }
        __syncthreads();
        if (tid < 64) {
            const int l = tid;
            double tl = (l < nbk) ? xf[jb + l] : 0.0;
            for (int j = nbk - 1; j >= 0; j--) {
                const double xj = __shfl(tl, j) / D[j * SOLVE_LD + j];
                if (l == j) tl = xj;
                else if (l < j) tl -= D[j * SOLVE_LD + l] * xj;
            }
            if (l < nbk) xf[jb + l] = tl;
        }
        __syncthreads();
    }
    for (int i = tid; i < k; i += NT) x[f + i] = xf[i];
}

static size_t solve_lds(int wcap) { return ((size_t)wcap + SOLVE_B * SOLVE_LD) * sizeof(double); }

void launch_fwd_level(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m,
                      const double *Lx, double *X, int64_t ldx, int nrhs,
                      const double *Wchild, double *Wout, int64_t wstride)
{
    if (count <= 0 || nrhs <= 0) return;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)k_fwd_level<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)k_bwd_level<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    dim3 grid((unsigned)count, (unsigned)nrhs);
    if (max_m <= 64)
        hipLaunchKernelGGL(k_fwd_level<64>, grid, dim3(64), solve_lds(64), st, ds, list, Lx, X, ldx, Wchild, Wout, wstride, 64);
    else
        hipLaunchKernelGGL(k_fwd_level<256>, grid, dim3(256), solve_lds(max_m), st, ds, list, Lx, X, ldx, Wchild, Wout, wstride, max_m);
}

void launch_bwd_level(hipStream_t st, const DevSym &ds, const int32_t *list, int count, int max_m,
                      const double *Lx, double *X, int64_t ldx, int nrhs)
{
    if (count <= 0 || nrhs <= 0) return;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)k_fwd_level<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)k_bwd_level<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    dim3 grid((unsigned)count, (unsigned)nrhs);
    if (max_m <= 64)
        hipLaunchKernelGGL(k_bwd_level<64>, grid, dim3(64), solve_lds(64), st, ds, list, Lx, X, ldx, 64);
    else
        hipLaunchKernelGGL(k_bwd_level<256>, grid, dim3(256), solve_lds(max_m), st, ds, list, Lx, X, ldx, max_m);
}

// ------------------------------------------------------------------------------------------
__global__ void k_perm_gather(const int32_t *__restrict__ perm, int64_t n, const double *__restrict__ in,
                              int64_t ldi, double *__restrict__ out, int64_t ldo)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i + blockIdx.y * ldo] = in[perm[i] + blockIdx.y * ldi];
}
__global__ void k_perm_scatter(const int32_t *__restrict__ perm, int64_t n, const double *__restrict__ in,
                               int64_t ldi, double *__restrict__ out, int64_t ldo)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[perm[i] + blockIdx.y * ldo] = in[i + blockIdx.y * ldi];
}
void launch_perm_gather(hipStream_t st, const int32_t *perm, int64_t n, int nrhs, const double *in, int64_t ldi,
                        double *out, int64_t ldo)
{
    if (n <= 0 || nrhs <= 0) return;
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)nrhs);
    hipLaunchKernelGGL(k_perm_gather, grid, dim3(256), 0, st, perm, n, in, ldi, out, ldo);
}
void launch_perm_scatter(hipStream_t st, const int32_t *perm, int64_t n, int nrhs, const double *in, int64_t ldi,
                         double *out, int64_t ldo)
{
    if (n <= 0 || nrhs <= 0) return;
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)nrhs);
    hipLaunchKernelGGL(k_perm_scatter, grid, dim3(256), 0, st, perm, n, in, ldi, out, ldo);
}

// Sparse right-hand sides (kvx_chol_spsolve, reach-restricted forward sweep).
// zero the update vectors of the listed fronts: slots[2 i] = offset in the parity buffer, slots[2 i + 1] = length
__global__ void k_zero_slots(const int64_t *__restrict__ slots, double *__restrict__ W, int64_t wstride)
{
    const int64_t off = slots[2 * blockIdx.x], len = slots[2 * blockIdx.x + 1];
    double *w = W + (int64_t)blockIdx.y * wstride + off;
    for (int64_t i = threadIdx.x; i < len; i += blockDim.x) w[i] = 0.0;
}
void launch_zero_slots(hipStream_t st, const int64_t *slots, int count, int nrhs, double *W, int64_t wstride)
{
    if (count <= 0 || nrhs <= 0) return;
    hipLaunchKernelGGL(k_zero_slots, dim3((unsigned)count, (unsigned)nrhs), dim3(64), 0, st, slots, W, wstride);
}
// X[pos[i]] = val[i] (the entries of a block of sparse columns; duplicates were summed on the host)
__global__ void k_scatter_entries(const int64_t *__restrict__ pos, const double *__restrict__ val, int64_t count, double *__restrict__ X)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) X[pos[i]] = val[i];
}
void launch_scatter_entries(hipStream_t st, const int64_t *pos, const double *val, int64_t count, double *X)
{
    if (count <= 0) return;
    hipLaunchKernelGGL(k_scatter_entries, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, pos, val, count, X);
}

// x[i] := keep[i] ? x[i] : 0 (sharded solves: every entry of x is reported by exactly one rank, the sum over ranks is x)
__global__ void k_mask_rows(const uint8_t *__restrict__ keep, int64_t n, double *__restrict__ X, int64_t ldx)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !keep[i]) X[i + (int64_t)blockIdx.y * ldx] = 0.0;
}
void launch_mask_rows(hipStream_t st, const uint8_t *keep, int64_t n, int nrhs, double *X, int64_t ldx)
{
    if (n <= 0 || nrhs <= 0) return;
    hipLaunchKernelGGL(k_mask_rows, dim3((unsigned)((n + 255) / 256), (unsigned)nrhs), dim3(256), 0, st, keep, n, X, ldx);
}

__global__ void k_extract_diag(DevSym ds, int64_t nsuper, const double *__restrict__ Lx, double *__restrict__ d)
{
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsuper) return;
    const int k = ds.k[s], m = ds.m[s], f = ds.first[s];
    const double *P = Lx + ds.px[s];
    for (int j = 0; j < k; j++) d[f + j] = P[j + (int64_t)j * m];
}
void launch_extract_diag(hipStream_t st, const DevSym &ds, int64_t nsuper, const double *Lx, double *d)
{
    if (nsuper <= 0) return;
    hipLaunchKernelGGL(k_extract_diag, dim3((unsigned)((nsuper + 255) / 256)), dim3(256), 0, st, ds, nsuper, Lx, d);
}

// LDL' view of the LL' factor (options['supernodal'] = 0: PAP' = L D L', L = Lc diag(Lc)^-1, D = diag(Lc)^2):
// X := diag(d) X (mode 0), diag(d)^-1 X (mode 1), diag(d)^-2 X (mode 2) on an n x nrhs block.
__global__ void k_diag_scale(int64_t n, const double *__restrict__ d, double *__restrict__ X, int64_t ldx, int mode)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double *x = X + (int64_t)blockIdx.y * ldx + i;
    const double v = d[i];
    *x = mode == 0 ? *x * v : (mode == 1 ? *x / v : *x / (v * v));
}
void launch_diag_scale(hipStream_t st, int64_t n, int nrhs, const double *d, double *X, int64_t ldx, int mode)
{
    if (n <= 0 || nrhs <= 0) return;
    hipLaunchKernelGGL(k_diag_scale, dim3((unsigned)((n + 255) / 256), (unsigned)nrhs), dim3(256), 0, st, n, d, X, ldx, mode);
}

}  // namespace kvx
