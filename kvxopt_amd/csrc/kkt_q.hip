// Nesterov-Todd scaling for the second-order-cone ('q') blocks of a cone program: the 'q' parts of
// misc.compute_scaling / update_scaling (src/python/misc.py:290-352, 467-580), misc_solvers.scale / scale2 / sprod / sinv
// (src/C/misc_solvers.c:144-186, 301-341, 671-700, 803-835), misc.ssqr (misc.py:951-959) and max_step
// (misc_solvers.c:1073-1085) -- SURVEY 8(f) item 4.  Own design: ONE workgroup per cone and launch for all cones of a vector
// (offsets table), the few inner products a cone needs reduced in a fixed order (shuffles + LDS: bitwise reproducible), the
// elementwise part in the same kernel.  Byte-bound work: every entry read and written once.
#include "../../include/kvxhip.h"
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>

namespace kvx { void set_last_error(const std::string &s); }

namespace {

constexpr int NT = 256;

// sum of up to N per-thread partials over the workgroup, every thread gets the totals (fixed tree)
template <int N>
__device__ __forceinline__ void block_sum(double (&v)[N], double *sh)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < N; k++) {
        double x = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
        if (lane == 0) sh[k * 4 + w] = x;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = (sh[k * 4] + sh[k * 4 + 1]) + (sh[k * 4 + 2] + sh[k * 4 + 3]);
    __syncthreads();
}

// hyperbolic norm as the reference forms it (misc.py jnrm2): sqrt(x0 - |x1|) * sqrt(x0 + |x1|)
__device__ __forceinline__ double jnrm(double x0, double tail_sq)
{
    const double a = sqrt(tail_sq);
    return sqrt(x0 - a) * sqrt(x0 + a);
}

__global__ __launch_bounds__(NT) void k_q_compute_scaling(const int64_t *__restrict__ off, const double *__restrict__ s,
                                                          const double *__restrict__ z, double *__restrict__ v,
                                                          double *__restrict__ beta, double *__restrict__ lm)
{
    __shared__ double sh[16];
    const int64_t o = off[blockIdx.x], m = off[blockIdx.x + 1] - o;
    const double *sk = s + o, *zk = z + o;
    double r[3] = {0.0, 0.0, 0.0};                       // |s1|^2, |z1|^2, s'z
    for (int64_t i = threadIdx.x; i < m; i += NT) {
        const double a = sk[i], b = zk[i];
        if (i > 0) { r[0] += a * a; r[1] += b * b; }
        r[2] += a * b;
    }
    block_sum(r, sh);
    const double s0 = sk[0], z0 = zk[0];
    const double aa = jnrm(s0, r[0]), bb = jnrm(z0, r[1]);
    const double cc = sqrt((r[2] / aa / bb + 1.0) / 2.0);
    // v = (s/a + J z/b) / (2c), then v := (v + e) / sqrt(2 (v0 + 1))
    const double v0 = ((-(z0 * (-1.0 / bb))) + (1.0 / aa) * s0) * (1.0 / 2.0 / cc) + 1.0;
    const double vs = 1.0 / sqrt(2.0 * v0);
    const double dd = 2.0 * cc + s0 / aa + z0 / bb;
    const double cs = (cc + z0 / bb) / dd / aa, cz = (cc + s0 / aa) / dd / bb, sab = sqrt(aa * bb);
    for (int64_t i = threadIdx.x; i < m; i += NT) {
        if (i == 0) { v[o] = v0 * vs; lm[o] = cc * sab; }
        else {
            v[o + i] = ((zk[i] * (-1.0 / bb) + (1.0 / aa) * sk[i]) * (1.0 / 2.0 / cc)) * vs;
            lm[o + i] = (sk[i] * cs + zk[i] * cz) * sab;
        }
    }
    if (threadIdx.x == 0) beta[blockIdx.x] = sqrt(aa / bb);
}

__global__ __launch_bounds__(NT) void k_q_update_scaling(const int64_t *__restrict__ off, double *__restrict__ s, double *__restrict__ z,
                                                         double *__restrict__ v, double *__restrict__ beta, double *__restrict__ lm)
{
    __shared__ double sh[24];
    const int64_t o = off[blockIdx.x], m = off[blockIdx.x + 1] - o;
    double *sk = s + o, *zk = z + o, *vk = v + o;
    double r[5] = {0.0, 0.0, 0.0, 0.0, 0.0};            // |s1|^2, |z1|^2, s'z, v's, v'Jz
    for (int64_t i = threadIdx.x; i < m; i += NT) {
        const double a = sk[i], b = zk[i], c = vk[i];
        if (i > 0) { r[0] += a * a; r[1] += b * b; r[4] -= c * b; } else r[4] += c * b;
        r[2] += a * b; r[3] += c * a;
    }
    block_sum(r, sh);
    const double aa = jnrm(sk[0], r[0]), bb = jnrm(zk[0], r[1]);
    const double s0 = sk[0] / aa, z0 = zk[0] / bb, vk0 = vk[0];
    const double cc = sqrt((1.0 + r[2] / aa / bb) / 2.0);
    const double vsd = r[3] / aa, vzd = r[4] / bb;
    const double vq = (vsd + vzd) / 2.0 / cc, vu = vsd - vzd;
    const double wk0 = 2.0 * vk0 * vq - (s0 + z0) / 2.0 / cc;
    const double dd = (vk0 * vu - s0 / 2.0 + z0 / 2.0) / (wk0 + 1.0);
    const double sab = sqrt(aa * bb);
    // new v0 before normalisation: 2 vq v0 - s0/(2c) - z0/(2c) + 1
    const double nv0 = 2.0 * vq * vk0 - s0 / 2.0 / cc - 0.5 / cc * z0 + 1.0;
    const double nvs = 1.0 / sqrt(2.0 * nv0);
    __syncthreads();                                     // every thread has read element 0 before anybody overwrites it
    for (int64_t i = threadIdx.x; i < m; i += NT) {
        const double si = sk[i] / aa, zi = zk[i] / bb, vi = vk[i];
        sk[i] = si; zk[i] = zi;                          // the reference leaves st / a and zt / b behind (misc.py:517-523)
        if (i == 0) { lm[o] = cc * sab; vk[0] = nv0 * nvs; }
        else {
            lm[o + i] = (vi * (2.0 * (-dd * vq + 0.5 * vu)) + si * (0.5 * (1.0 - dd / cc)) + zi * (0.5 * (1.0 + dd / cc))) * sab;
            vk[i] = (2.0 * vq * vi + 0.5 / cc * si - 0.5 / cc * zi) * nvs;
        }
    }
    if (threadIdx.x == 0) beta[blockIdx.x] *= sqrt(aa / bb);
}

// x := beta (2 v v' - J) x, or its inverse (misc_solvers.c:144-186); blockIdx.y = column of x
__global__ __launch_bounds__(NT) void k_q_scale(const int64_t *__restrict__ off, const double *__restrict__ v, const double *__restrict__ beta,
                                                double *__restrict__ x, int64_t ldx, int inverse)
{
    __shared__ double sh[4];
    const int64_t o = off[blockIdx.x], m = off[blockIdx.x + 1] - o;
    const double *vk = v + o;
    double *xk = x + o + (int64_t)blockIdx.y * ldx;
    double r[1] = {0.0};
    for (int64_t i = threadIdx.x; i < m; i += NT) r[0] += (inverse && i == 0 ? -vk[i] : vk[i]) * xk[i];
    block_sum(r, sh);
    const double w = r[0], b = inverse ? 1.0 / beta[blockIdx.x] : beta[blockIdx.x];
    for (int64_t i = threadIdx.x; i < m; i += NT) {
        double t;
        if (!inverse) t = (i == 0 ? -xk[i] : xk[i]) + 2.0 * vk[i] * w;
        else { t = xk[i] + 2.0 * vk[i] * w; if (i == 0) t = -t; }
        xk[i] = t * b;
    }
}

__global__ __launch_bounds__(NT) void k_q_scale2(const int64_t *__restrict__ off, const double *__restrict__ lm, double *__restrict__ x, int inverse)
{
    __shared__ double sh[8];
    const int64_t o = off[blockIdx.x], m = off[blockIdx.x + 1] - o;
    const double *lk = lm + o;
    double *xk = x + o;
    double r[2] = {0.0, 0.0};                            // |l1|^2, l1'x1
    for (int64_t i = threadIdx.x + 1; i < m; i += NT) { r[0] += lk[i] * lk[i]; r[1] += lk[i] * xk[i]; }
    block_sum(r, sh);
    const double l0 = lk[0], x0 = xk[0];
    const double nl = sqrt(r[0]);
    double a = sqrt(l0 + nl) * sqrt(l0 - nl);
    const double lx = (inverse ? (l0 * x0 + r[1]) : (l0 * x0 - r[1])) / a;
    double b = (x0 + lx) / (l0 / a + 1.0) / a;
    if (!inverse) { b = -b; a = 1.0 / a; }
    __syncthreads();
    for (int64_t i = threadIdx.x; i < m; i += NT) xk[i] = (i == 0 ? lx : xk[i] + b * lk[i]) * a;
}

// op 0: x := y o x; 1: x := y o\ x; 2: x := y o y
__global__ __launch_bounds__(NT) void k_q_prod(const int64_t *__restrict__ off, double *__restrict__ x, const double *__restrict__ y, int op)
{
    __shared__ double sh[8];
    const int64_t o = off[blockIdx.x], m = off[blockIdx.x + 1] - o;
    const double *yk = y + o;
    double *xk = x + o;
    double r[2] = {0.0, 0.0};
    for (int64_t i = threadIdx.x; i < m; i += NT) {
        if (op == 0) r[0] += yk[i] * xk[i];
        else if (op == 1) { if (i > 0) { r[0] += yk[i] * yk[i]; r[1] += xk[i] * yk[i]; } }
        else r[0] += yk[i] * yk[i];
    }
    block_sum(r, sh);
    const double y0 = yk[0], x0 = op == 2 ? 0.0 : xk[0];
    __syncthreads();
    if (op == 0) {
        for (int64_t i = threadIdx.x; i < m; i += NT) xk[i] = i == 0 ? r[0] : y0 * xk[i] + x0 * yk[i];
    } else if (op == 1) {
        const double nl = sqrt(r[0]);
        const double a = (y0 + nl) * (y0 - nl), d = r[1];
        const double al1 = a / y0, al2 = d / y0 - x0, inv = 1.0 / a;
        for (int64_t i = threadIdx.x; i < m; i += NT) xk[i] = (i == 0 ? x0 * y0 - d : al1 * xk[i] + al2 * yk[i]) * inv;
    } else {
        const double nrm = sqrt(r[0]);
        for (int64_t i = threadIdx.x; i < m; i += NT) xk[i] = i == 0 ? nrm * nrm : 2.0 * y0 * yk[i];
    }
}

__global__ __launch_bounds__(NT) void k_q_max_step(const int64_t *__restrict__ off, const double *__restrict__ x, double *__restrict__ out)
{
    __shared__ double sh[4];
    const int64_t o = off[blockIdx.x], m = off[blockIdx.x + 1] - o;
    double r[1] = {0.0};
    for (int64_t i = threadIdx.x + 1; i < m; i += NT) r[0] += x[o + i] * x[o + i];
    block_sum(r, sh);
    if (threadIdx.x == 0) out[blockIdx.x] = sqrt(r[0]) - x[o];
}

int fail(hipError_t e, const char *what)
{
    kvx::set_last_error(std::string(what) + ": " + hipGetErrorString(e));
    return KVX_EDEVICE;
}
#define LAUNCH_OK(what) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(e_, what); } while (0)

}  // namespace

extern "C" {

int kvx_ntq_compute_scaling_dev(int64_t nq, const int64_t *off_dev, const double *s_dev, const double *z_dev, double *v_dev,
                                double *beta_dev, double *lmbda_dev)
{
    if (nq < 0) return KVX_EINVAL;
    if (nq == 0) return KVX_OK;
    hipLaunchKernelGGL(k_q_compute_scaling, dim3((unsigned)nq), dim3(NT), 0, nullptr, off_dev, s_dev, z_dev, v_dev, beta_dev, lmbda_dev);
    LAUNCH_OK("k_q_compute_scaling");
    return KVX_OK;
}

int kvx_ntq_update_scaling_dev(int64_t nq, const int64_t *off_dev, double *s_dev, double *z_dev, double *v_dev, double *beta_dev,
                               double *lmbda_dev)
{
    if (nq < 0) return KVX_EINVAL;
    if (nq == 0) return KVX_OK;
    hipLaunchKernelGGL(k_q_update_scaling, dim3((unsigned)nq), dim3(NT), 0, nullptr, off_dev, s_dev, z_dev, v_dev, beta_dev, lmbda_dev);
    LAUNCH_OK("k_q_update_scaling");
    return KVX_OK;
}

int kvx_ntq_scale_dev(int64_t nq, const int64_t *off_dev, const double *v_dev, const double *beta_dev, double *x_dev, int64_t ldx,
                      int64_t ncols, int inverse)
{
    if (nq < 0 || ncols < 0 || ncols > 65535) return KVX_EINVAL;
    if (nq == 0 || ncols == 0) return KVX_OK;
    hipLaunchKernelGGL(k_q_scale, dim3((unsigned)nq, (unsigned)ncols), dim3(NT), 0, nullptr, off_dev, v_dev, beta_dev, x_dev, ldx, inverse ? 1 : 0);
    LAUNCH_OK("k_q_scale");
    return KVX_OK;
}

int kvx_ntq_scale2_dev(int64_t nq, const int64_t *off_dev, const double *lmbda_dev, double *x_dev, int inverse)
{
    if (nq < 0) return KVX_EINVAL;
    if (nq == 0) return KVX_OK;
    hipLaunchKernelGGL(k_q_scale2, dim3((unsigned)nq), dim3(NT), 0, nullptr, off_dev, lmbda_dev, x_dev, inverse ? 1 : 0);
    LAUNCH_OK("k_q_scale2");
    return KVX_OK;
}

int kvx_ntq_prod_dev(int64_t nq, const int64_t *off_dev, double *x_dev, const double *y_dev, int op)
{
    if (nq < 0 || op < 0 || op > 2) return KVX_EINVAL;
    if (nq == 0) return KVX_OK;
    hipLaunchKernelGGL(k_q_prod, dim3((unsigned)nq), dim3(NT), 0, nullptr, off_dev, x_dev, y_dev, op);
    LAUNCH_OK("k_q_prod");
    return KVX_OK;
}

int kvx_ntq_max_step_dev(int64_t nq, const int64_t *off_dev, const double *x_dev, double *out_dev)
{
    if (nq < 0) return KVX_EINVAL;
    if (nq == 0) return KVX_OK;
    hipLaunchKernelGGL(k_q_max_step, dim3((unsigned)nq), dim3(NT), 0, nullptr, off_dev, x_dev, out_dev);
    LAUNCH_OK("k_q_max_step");
    return KVX_OK;
}

}  // extern "C"
