// lab 5: the 16-pivot register sweep of potrf_lds (phase A: one wavefront factors a 16 x 16 diagonal block and inverts it
// in the same instruction stream) in isolation.  Cycles per sweep for variants of the pivot step; results checked against
// a host Cholesky.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/potrf_lab5.hip -o /tmp/lab5
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <utility>
#include <vector>

struct PivRule { double floor, sub; int flag_all; };

__device__ __forceinline__ double rl(double v, int lane)
{
    const long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// ---- V0: the step as it is in kernels_big.hip ------------------------------------------------------------------------
template <int J>
__device__ __forceinline__ void step_v0(double (&acc)[16], double *colbuf, int wslot, int rr, int &bad, const PivRule pr)
{
    double *cb = colbuf + (J & 1) * 80;
    const double aj = acc[J];
    cb[wslot] = aj;
    double d = rl(aj, J);
    const bool neg = !(d > pr.floor);
    bad = (neg && bad > J) ? J : bad;
    d = neg ? pr.sub : d;
    double inv = __builtin_amdgcn_rsq(d);
    const double hd = 0.5 * d;
    inv = inv * __builtin_fma(-hd * inv, inv, 1.5);
    inv = inv * __builtin_fma(-hd * inv, inv, 1.5);
    const double lj = aj * inv;
    const double w = (rr <= J) ? 0.0 : lj * inv;
#pragma unroll
    for (int t = J + 1; t < 16; t++) acc[t] = __builtin_fma(-w, cb[t], acc[t]);
    acc[J] = (rr < J) ? 0.0 : lj;
}

// ---- V1: multiplier through a reciprocal (quotient refinement), the root off the dependent path -------------------
// w = a / d: r0 = rcp(d) (~2^-26), q0 = a r0, q1 = q0 + r0 (a - d q0)  (2^-52); the stored column l = a rsqrt(d) is
// computed beside it and feeds nothing in the sweep.
template <int J>
__device__ __forceinline__ void step_v1(double (&acc)[16], double *colbuf, int wslot, int rr, int &bad, const PivRule pr)
{
    double *cb = colbuf + (J & 1) * 80;
    const double aj = acc[J];
    cb[wslot] = aj;
    double d = rl(aj, J);
    const bool neg = !(d > pr.floor);
    bad = (neg && bad > J) ? J : bad;
    d = neg ? pr.sub : d;
    const double r0 = __builtin_amdgcn_rcp(d);
    const double q0 = aj * r0;
    const double q1 = __builtin_fma(__builtin_fma(-d, q0, aj), r0, q0);
    const double w = (rr <= J) ? 0.0 : q1;
#pragma unroll
    for (int t = J + 1; t < 16; t++) acc[t] = __builtin_fma(-w, cb[t], acc[t]);
    double inv = __builtin_amdgcn_rsq(d);
    const double hd = 0.5 * d;
    inv = inv * __builtin_fma(-hd * inv, inv, 1.5);
    inv = inv * __builtin_fma(-hd * inv, inv, 1.5);
    acc[J] = (rr < J) ? 0.0 : aj * inv;
}

// ---- V2: V1 + the diagonal carried in a register of its own: the next pivot does not wait for the LDS round trip -------
// lane r keeps dg = a_rr; step J: dg -= w a_rJ (its own entries).  The pivot of step J + 1 is readlane(dg, J + 1) as soon as
// w is known; acc[J + 1] of the other lanes (the column) still goes through the published column.
template <int J>
__device__ __forceinline__ void step_v2(double (&acc)[16], double &dg, double *colbuf, int wslot, int rr, int &bad, const PivRule pr)
{
    double *cb = colbuf + (J & 1) * 80;
    const double aj = acc[J];
    cb[wslot] = aj;
    double d = rl(dg, J);
    const bool neg = !(d > pr.floor);
    bad = (neg && bad > J) ? J : bad;
    d = neg ? pr.sub : d;
    const double r0 = __builtin_amdgcn_rcp(d);
    const double ajj = (rr == J) ? d : aj;          // the pivot lane's own entry of column J is the pivot itself
    const double q0 = ajj * r0;
    const double q1 = __builtin_fma(__builtin_fma(-d, q0, ajj), r0, q0);
    const double w = (rr <= J) ? 0.0 : q1;
    dg = __builtin_fma(-w, ajj, dg);
#pragma unroll
    for (int t = J + 1; t < 16; t++) acc[t] = __builtin_fma(-w, cb[t], acc[t]);
    double inv = __builtin_amdgcn_rsq(d);
    const double hd = 0.5 * d;
    inv = inv * __builtin_fma(-hd * inv, inv, 1.5);
    inv = inv * __builtin_fma(-hd * inv, inv, 1.5);
    acc[J] = (rr < J) ? 0.0 : ajj * inv;
}

// ---- V3: V0's arithmetic, software-pipelined by hand: the first update of step J (column J + 1, the next pivot column) is done
// first, the next step's publish / pivot / rsqrt chain is started at once, and the rest of step J's updates (columns J + 2 ..)
// are interleaved one by one between the dependent operations of that chain.
struct Carry { double w; const double *cb; };
template <int J>
__device__ __forceinline__ void step_v3(double (&acc)[16], double *colbuf, int wslot, int rr, int &bad, const PivRule pr, Carry &c)
{
    // on entry: c.w / c.cb belong to step J - 1 (J > 0); the columns J .. 15 still lack step J - 1's update
    double *cbn = colbuf + (J & 1) * 80;
    const double wp = c.w;
    const double *cbp = c.cb;
#define UPD(T) do { if (J > 0 && (T) < 16) acc[(T)] = __builtin_fma(-wp, cbp[(T)], acc[(T)]); } while (0)
    UPD(J);                                         // the pivot column first: everything below waits for it
    const double aj = acc[J];
    cbn[wslot] = aj;
    double d = rl(aj, J);
    const bool neg = !(d > pr.floor);
    UPD(J + 1);
    bad = (neg && bad > J) ? J : bad;
    d = neg ? pr.sub : d;
    UPD(J + 2);
    double inv = __builtin_amdgcn_rsq(d);
    UPD(J + 3);
    const double hd = 0.5 * d;
    UPD(J + 4);
    double t = hd * inv;
    UPD(J + 5);
    double u = __builtin_fma(-t, inv, 1.5);
    UPD(J + 6);
    inv = inv * u;
    UPD(J + 7);
    t = hd * inv;
    UPD(J + 8);
    u = __builtin_fma(-t, inv, 1.5);
    UPD(J + 9);
    inv = inv * u;
    UPD(J + 10);
    const double lj = aj * inv;
    UPD(J + 11);
    const double w = (rr <= J) ? 0.0 : lj * inv;
    UPD(J + 12);
    UPD(J + 13);
    UPD(J + 14);
    UPD(J + 15);
#undef UPD
    acc[J] = (rr < J) ? 0.0 : lj;
    c.w = w;
    c.cb = cbn;
}

// ---- V4: V3 with the published column of the previous step held in REGISTERS: its LDS reads are issued right after the publish of
// that step (their latency disappears behind that step's rsqrt chain), so the delayed updates interleaved with the next chain
// wait for nothing.
struct Carry4 { double w; double cb[16]; };
template <int J>
__device__ __forceinline__ void step_v4(double (&acc)[16], double *colbuf, int wslot, int rr, int &bad, const PivRule pr, Carry4 &c)
{
    double *cbn = colbuf + (J & 1) * 80;
    const double wp = c.w;
#define UPD(T) do { if (J > 0 && (T) < 16) acc[(T)] = __builtin_fma(-wp, c.cb[(T)], acc[(T)]); } while (0)
    UPD(J);
    const double aj = acc[J];
    cbn[wslot] = aj;
    double d = rl(aj, J);
    double nb[16];
#pragma unroll
    for (int t = 0; t < 16; t++) nb[t] = (t > J) ? cbn[t] : 0.0;      // (in order behind the publish: the reads see it)
    const bool neg = !(d > pr.floor);
    UPD(J + 1);
    bad = (neg && bad > J) ? J : bad;
    d = neg ? pr.sub : d;
    UPD(J + 2);
    double inv = __builtin_amdgcn_rsq(d);
    UPD(J + 3);
    const double hd = 0.5 * d;
    UPD(J + 4);
    double t = hd * inv;
    UPD(J + 5);
    double u = __builtin_fma(-t, inv, 1.5);
    UPD(J + 6);
    inv = inv * u;
    UPD(J + 7);
    t = hd * inv;
    UPD(J + 8);
    u = __builtin_fma(-t, inv, 1.5);
    UPD(J + 9);
    inv = inv * u;
    UPD(J + 10);
    const double lj = aj * inv;
    UPD(J + 11);
    const double w = (rr <= J) ? 0.0 : lj * inv;
    UPD(J + 12);
    UPD(J + 13);
    UPD(J + 14);
    UPD(J + 15);
#undef UPD
    acc[J] = (rr < J) ? 0.0 : lj;
    c.w = w;
#pragma unroll
    for (int t = 0; t < 16; t++) c.cb[t] = nb[t];
}

// ---- V5: V4 with a shorter dependent chain per pivot: (a) the pivot rule is applied AFTER the hardware rsqrt (a select against the
// constant seed of the substitute pivot; the comparison runs beside the rsqrt), (b) the multiplier uses 1 / d refined on its own
// from the seed, r0 = s^2, e = 1 - d r0, r = r0 (1 + e + e^2) (third order: the 2^-26 seed gives 2^-75), three dependent operations
// shorter than two Newton steps of the root followed by two multiplications; (c) 1 / sqrt(d) for the stored column comes from the
// same e beside the chain: s (1 + e / 2 + 3 e^2 / 8).
template <int J>
__device__ __forceinline__ void step_v5(double (&acc)[16], double *colbuf, int wslot, int rr, int &bad, const PivRule pr, Carry4 &c, double sub_seed)
{
    double *cbn = colbuf + (J & 1) * 80;
    const double wp = c.w;
#define UPD(T) do { if (J > 0 && (T) < 16) acc[(T)] = __builtin_fma(-wp, c.cb[(T)], acc[(T)]); } while (0)
    UPD(J);
    const double aj = acc[J];
    cbn[wslot] = aj;
    const double draw = rl(aj, J);
    double nb[16];
#pragma unroll
    for (int t = 0; t < 16; t++) nb[t] = (t > J) ? cbn[t] : 0.0;
    const bool neg = !(draw > pr.floor);
    double s0 = __builtin_amdgcn_rsq(draw);
    UPD(J + 1);
    bad = (neg && bad > J) ? J : bad;
    const double d = neg ? pr.sub : draw;
    s0 = neg ? sub_seed : s0;
    UPD(J + 2);
    const double r0 = s0 * s0;
    UPD(J + 3);
    const double e = __builtin_fma(-d, r0, 1.0);
    UPD(J + 4);
    const double g = __builtin_fma(e, e, e);
    UPD(J + 5);
    const double r = __builtin_fma(r0, g, r0);
    UPD(J + 6);
    const double w = (rr <= J) ? 0.0 : aj * r;
    UPD(J + 7);
    // beside the chain: the root's reciprocal for the stored column
    const double q = s0 * e;
    UPD(J + 8);
    const double pz = __builtin_fma(0.375, e, 0.5);
    UPD(J + 9);
    const double inv = __builtin_fma(q, pz, s0);
    UPD(J + 10);
    const double lj = aj * inv;
    UPD(J + 11);
    UPD(J + 12);
    UPD(J + 13);
    UPD(J + 14);
    UPD(J + 15);
#undef UPD
    acc[J] = (rr < J) ? 0.0 : lj;
    c.w = w;
#pragma unroll
    for (int t = 0; t < 16; t++) c.cb[t] = nb[t];
}

template <int V, int... Js>
__device__ __forceinline__ void sweep(double (&acc)[16], double *colbuf, int wslot, int rr, int lr, int &bad, const PivRule pr,
                                      std::integer_sequence<int, Js...>)
{
    if (V == 0) (step_v0<Js>(acc, colbuf, wslot, rr, bad, pr), ...);
    else if (V == 1) (step_v1<Js>(acc, colbuf, wslot, rr, bad, pr), ...);
    else if (V == 5) {
        Carry4 c;
        c.w = 0.0;
#pragma unroll
        for (int t = 0; t < 16; t++) c.cb[t] = 0.0;
        const double sub_seed = __builtin_amdgcn_rsq(pr.sub);
        (step_v5<Js>(acc, colbuf, wslot, rr, bad, pr, c, sub_seed), ...);
    } else if (V == 4) {
        Carry4 c;
        c.w = 0.0;
#pragma unroll
        for (int t = 0; t < 16; t++) c.cb[t] = 0.0;
        (step_v4<Js>(acc, colbuf, wslot, rr, bad, pr, c), ...);
    } else if (V == 3) {
        Carry c{0.0, colbuf};
        (step_v3<Js>(acc, colbuf, wslot, rr, bad, pr, c), ...);
    } else {
        // dg: the diagonal entry of a factor lane (lane r: acc[r]); the inverse lanes carry nothing useful there
        double dg = 0.0;
#pragma unroll
        for (int c = 0; c < 16; c++) dg = (c == lr) ? acc[c] : dg;
        (step_v2<Js>(acc, dg, colbuf, wslot, rr, bad, pr), ...);
    }
}

// one wavefront: REPS sweeps on the same block (reloaded from LDS each time, as potrf_lds does)
template <int V>
__global__ __launch_bounds__(64) void k_lab(const double *A, double *L, double *Y, long long *cycles, int reps)
{
    __shared__ double Sd[16 * 16];
    __shared__ double Yd[16 * 17];
    __shared__ double colbuf[2 * 80];
    const int i = threadIdx.x, lr = i & 15;
    for (int e = i; e < 256; e += 64) Sd[e] = A[e];
    __syncthreads();
    const PivRule pr{0.0, 1.0, 1};
    const bool fac = i < 16;
    double acc[16];
    long long t0 = 0, t1 = 0;
    for (int rep = 0; rep < reps + 1; rep++) {
        if (rep == 1) t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const double lv = Sd[lr + 16 * c];
            acc[c] = fac ? ((c <= lr) ? lv : 0.0) : ((i < 32 && c == lr) ? 1.0 : 0.0);
        }
        int bad = 16;
        sweep<V>(acc, colbuf, fac ? lr : i, fac ? lr : 1000, lr, bad, pr, std::make_integer_sequence<int, 16>());
        if (fac) {
#pragma unroll
            for (int c = 0; c < 16; c++) Yd[lr + 17 * c] = acc[c];          // (a store the compiler cannot drop)
        } else if (i < 32) {
#pragma unroll
            for (int r = 0; r < 16; r++) Yd[r + 17 * lr] += acc[r];
        }
        __syncthreads();
    }
    t1 = __builtin_readcyclecounter();
    if (i == 0) cycles[0] = (t1 - t0);
    // one more sweep for the results
#pragma unroll
    for (int c = 0; c < 16; c++) {
        const double lv = Sd[lr + 16 * c];
        acc[c] = fac ? ((c <= lr) ? lv : 0.0) : ((i < 32 && c == lr) ? 1.0 : 0.0);
    }
    int bad = 16;
    sweep<V>(acc, colbuf, fac ? lr : i, fac ? lr : 1000, lr, bad, pr, std::make_integer_sequence<int, 16>());
    if (fac) { for (int c = 0; c < 16; c++) L[lr + 16 * c] = acc[c]; }
    else if (i < 32) { for (int r = 0; r < 16; r++) Y[r + 16 * lr] = acc[r]; }
}

int main()
{
    std::vector<double> A(256), Lh(256, 0.0);
    unsigned s = 12345;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / (1 << 24) - 0.5; };
    std::vector<double> B(256);
    for (auto &b : B) b = rnd();
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            double v = (i == j) ? 4.0 : 0.0;
            for (int k = 0; k < 16; k++) v += B[i + 16 * k] * B[j + 16 * k];
            A[i + 16 * j] = v;
        }
    // host Cholesky and inverse of the factor
    std::vector<double> M = A;
    for (int j = 0; j < 16; j++) {
        double d = M[j + 16 * j];
        for (int k = 0; k < j; k++) d -= Lh[j + 16 * k] * Lh[j + 16 * k];
        Lh[j + 16 * j] = std::sqrt(d);
        for (int i = j + 1; i < 16; i++) {
            double v = M[i + 16 * j];
            for (int k = 0; k < j; k++) v -= Lh[i + 16 * k] * Lh[j + 16 * k];
            Lh[i + 16 * j] = v / Lh[j + 16 * j];
        }
    }
    std::vector<double> Yh(256, 0.0);
    for (int c = 0; c < 16; c++)
        for (int r = c; r < 16; r++) {
            double v = (r == c) ? 1.0 : 0.0;
            for (int k = c; k < r; k++) v -= Lh[r + 16 * k] * Yh[k + 16 * c];
            Yh[r + 16 * c] = v / Lh[r + 16 * r];
        }
    double *dA, *dL, *dY;
    long long *dc;
    hipMalloc(&dA, 2048); hipMalloc(&dL, 2048); hipMalloc(&dY, 2048); hipMalloc(&dc, 8);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
    const int reps = 2000;
    auto run = [&](int v) {
        hipMemset(dL, 0, 2048); hipMemset(dY, 0, 2048);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int it = 0; it < 3; it++) {
            hipEventRecord(e0, 0);
            if (v == 0) hipLaunchKernelGGL(k_lab<0>, dim3(1), dim3(64), 0, 0, dA, dL, dY, dc, reps);
            if (v == 1) hipLaunchKernelGGL(k_lab<1>, dim3(1), dim3(64), 0, 0, dA, dL, dY, dc, reps);
            if (v == 2) hipLaunchKernelGGL(k_lab<2>, dim3(1), dim3(64), 0, 0, dA, dL, dY, dc, reps);
            if (v == 5) hipLaunchKernelGGL(k_lab<5>, dim3(1), dim3(64), 0, 0, dA, dL, dY, dc, reps);
            if (v == 4) hipLaunchKernelGGL(k_lab<4>, dim3(1), dim3(64), 0, 0, dA, dL, dY, dc, reps);
            if (v == 3) hipLaunchKernelGGL(k_lab<3>, dim3(1), dim3(64), 0, 0, dA, dL, dY, dc, reps);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1);
        }
        std::vector<double> L(256), Y(256);
        long long cyc;
        hipMemcpy(L.data(), dL, 2048, hipMemcpyDeviceToHost);
        hipMemcpy(Y.data(), dY, 2048, hipMemcpyDeviceToHost);
        hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
        double el = 0, ey = 0;
        for (int c = 0; c < 16; c++)
            for (int r = c; r < 16; r++) {
                el = std::fmax(el, std::fabs(L[r + 16 * c] - Lh[r + 16 * c]));
                ey = std::fmax(ey, std::fabs(Y[r + 16 * c] - Yh[r + 16 * c]));
            }
        printf("V%d: %.1f counter ticks per 16-pivot sweep incl. load/store (%.1f per pivot); %.3f us per sweep by events; max err L %.2e Y %.2e\n", v,
               (double)cyc / reps, (double)cyc / reps / 16, 1e3 * ms / (reps + 2), el, ey);
    };
    run(0); run(1); run(2); run(3); run(4); run(5);
    return 0;
}
