// Lab: C -= X X' (lower triangle), X = u x K column-major (ld = ldx), C = u x u (ld = ldc); LDS-staged FP64 MFMA tiles.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/syrk_lab.hip -o scratch/syrk_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// SM x SN 16-blocks per wave, waves 2 x 2: tile = 32 SM rows x 32 SN columns.  KC = k per chunk.
template <int SM, int SN, int KC, int MINW>
__global__ __launch_bounds__(256, MINW) void k_syrk_lds(const double *__restrict__ X, int ldx, int K, double *__restrict__ C, int ldc, int u, int T)
{
    constexpr int TR = 32 * SM, TC = 32 * SN;
    constexpr int LDR = TR + 16, LDC = TC + 16;
    __shared__ double xb[2][KC * LDR];      // rows strip
    __shared__ double xa[2][KC * LDC];      // cols strip
    // triangular tile index (row-tile ti >= col-tile tj when TR == TC; for TR != TC use rectangular + skip)
    int ti, tj;
    if (TR == TC) {
        const unsigned L = blockIdx.x;
        unsigned si = (unsigned)((__builtin_sqrtf(8.0f * (float)L + 1.0f) - 1.0f) * 0.5f);
        while ((si + 1) * (si + 2) / 2 <= L) si++;
        while (si * (si + 1) / 2 > L) si--;
        ti = si; tj = L - si * (si + 1) / 2;
    } else {
        ti = blockIdx.x; tj = blockIdx.y;
    }
    const int r0 = TR * ti, c0 = TC * tj;
    if (r0 >= u || c0 >= u) return;
    if (c0 > r0 + TR - 1) return;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wr = w >> 1, wc = w & 1;
    const int l = tid & 63, lr = l & 15, lk = l >> 4;
    // staging: 16-byte units; strip rows: TR/2 pairs per k
    constexpr int UR = TR / 2 * KC / 256, UC = TC / 2 * KC / 256;      // units per thread
    static_assert(UR >= 1 && UC >= 1, "chunk too small");
    d2 gr[UR], gc[UC];
    auto ldg = [&](int k0) {
#pragma unroll
        for (int j = 0; j < UR; j++) {
            const int id = tid + 256 * j, kk = id / (TR / 2), pr = id % (TR / 2);
            const int row = min(r0 + 2 * pr, u - 1);      // (one double of slack behind X)
            const int kc = min(k0 + kk, K - 1);
            gr[j] = *(const d2 *)(X + row + (int64_t)kc * ldx);
            if (k0 + kk >= K) gr[j] = (d2){0.0, 0.0};
        }
#pragma unroll
        for (int j = 0; j < UC; j++) {
            const int id = tid + 256 * j, kk = id / (TC / 2), pr = id % (TC / 2);
            const int row = min(c0 + 2 * pr, u - 1);
            const int kc = min(k0 + kk, K - 1);
            gc[j] = *(const d2 *)(X + row + (int64_t)kc * ldx);
            if (k0 + kk >= K) gc[j] = (d2){0.0, 0.0};
        }
    };
    auto sts = [&](int b) {
#pragma unroll
        for (int j = 0; j < UR; j++) {
            const int id = tid + 256 * j, kk = id / (TR / 2), pr = id % (TR / 2);
            *(d2 *)(&xb[b][kk * LDR + 2 * pr]) = gr[j];
        }
#pragma unroll
        for (int j = 0; j < UC; j++) {
            const int id = tid + 256 * j, kk = id / (TC / 2), pr = id % (TC / 2);
            *(d2 *)(&xa[b][kk * LDC + 2 * pr]) = gc[j];
        }
    };
    d4 acc[SM][SN];
#pragma unroll
    for (int s = 0; s < SM; s++)
#pragma unroll
        for (int t = 0; t < SN; t++) acc[s][t] = (d4){0.0, 0.0, 0.0, 0.0};
    const int nchunk = (K + KC - 1) / KC;
    ldg(0);
    for (int ch = 0; ch < nchunk; ch++) {
        const int b = ch & 1;
        sts(b);
        if (ch + 1 < nchunk) ldg((ch + 1) * KC);
        __syncthreads();
        const double *oa = xa[b] + 16 * SN * wc + lr;
        const double *ob = xb[b] + 16 * SM * wr + lr;
#pragma unroll
        for (int ks = 0; ks < KC; ks += 4) {
            double av[SN], bv[SM];
#pragma unroll
            for (int t = 0; t < SN; t++) av[t] = oa[(ks + lk) * LDC + 16 * t];
#pragma unroll
            for (int s = 0; s < SM; s++) bv[s] = ob[(ks + lk) * LDR + 16 * s];
#pragma unroll
            for (int s = 0; s < SM; s++)
#pragma unroll
                for (int t = 0; t < SN; t++) acc[s][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], bv[s], acc[s][t], 0, 0, 0);
        }
    }
    // epilogue
#pragma unroll
    for (int s = 0; s < SM; s++) {
        const int rr = r0 + 16 * SM * wr + 16 * s + lr;
        const bool rin = rr < u;
        const int rs = min(rr, u - 1);
        double old[SN][4];
#pragma unroll
        for (int t = 0; t < SN; t++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int c = c0 + 16 * SN * wc + 16 * t + lk + 4 * q;
                const int cs = min(c, rs);
                old[t][q] = C[rs + (int64_t)cs * ldc];
            }
#pragma unroll
        for (int t = 0; t < SN; t++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int c = c0 + 16 * SN * wc + 16 * t + lk + 4 * q;
                if (rin && c <= rr) C[rr + (int64_t)c * ldc] = old[t][q] - acc[s][t][q];
            }
    }
}

// the existing form: 64 x 64 tile, every wave loads its operands from global (8 B per lane)
template <int KW>
__global__ __launch_bounds__(256) void k_syrk_direct(const double *__restrict__ X, int ldx, int K, double *__restrict__ C, int ldc, int u)
{
    const int ti = blockIdx.x, tj = blockIdx.y;
    if (tj > ti) return;
    const int r0 = 64 * ti, c0 = 64 * tj;
    if (r0 >= u) return;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, lr = l & 15, lk = l >> 4;
    d4 acc[4];
    for (int t = 0; t < 4; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    const int rr = r0 + 16 * w + lr;
    const bool rin = rr < u;
    for (int kg = 0; kg < K; kg += 16) {
        double bq[4], aq[4][4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int kc = kg + 4 * q + lk;
            const bool kin = kc < K;
            const int64_t coff = (int64_t)(kin ? kc : 0) * ldx;
            bq[q] = kin && rin ? X[rr + coff] : 0.0;
#pragma unroll
            for (int t = 0; t < 4; t++) { const int cc = c0 + 16 * t + lr; aq[q][t] = (kin && cc < u) ? X[min(cc, u - 1) + coff] : 0.0; }
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[q][t], bq[q], acc[t], 0, 0, 0);
    }
    const int rs = min(rr, u - 1);
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c = c0 + 16 * t + lk + 4 * q;
            if (rin && c <= rr) C[rs + (int64_t)c * ldc] -= acc[t][q];
        }
}

__global__ void k_ref(const double *X, int ldx, int K, double *C, int ldc, int u)
{
    const int r = blockIdx.x * 16 + threadIdx.x, c = blockIdx.y * 16 + threadIdx.y;
    if (r >= u || c > r) return;
    double s = 0;
    for (int k = 0; k < K; k++) s += X[r + (int64_t)k * ldx] * X[c + (int64_t)k * ldx];
    C[r + (int64_t)c * ldc] -= s;
}

template <class F>
float time_it(F f, int reps)
{
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    f();
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f();
    CHK(hipEventRecord(b));
    CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const int u = argc > 1 ? atoi(argv[1]) : 3000, K = argc > 2 ? atoi(argv[2]) : 512;
    const int check = argc > 3 ? atoi(argv[3]) : 1;
    const int ldx = u + 7, ldc = u;          // odd-ish ld on purpose (8-byte aligned only)
    std::vector<double> hX((size_t)ldx * K), hC((size_t)ldc * u);
    srand(1);
    for (auto &v : hX) v = (rand() % 2001 - 1000) / 1000.0;
    for (auto &v : hC) v = (rand() % 2001 - 1000) / 1000.0;
    double *X, *C, *Cr;
    CHK(hipMalloc(&X, hX.size() * 8 + 64)); CHK(hipMalloc(&C, hC.size() * 8)); CHK(hipMalloc(&Cr, hC.size() * 8));
    CHK(hipMemcpy(X, hX.data(), hX.size() * 8, hipMemcpyHostToDevice));
    const double flops = (double)u * (u + 1) * K;     // u(u+1)/2 entries x 2K
    auto verify = [&](const char *name) {
        if (!check) return;
        std::vector<double> a(hC.size()), b(hC.size());
        CHK(hipMemcpy(a.data(), C, a.size() * 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(b.data(), Cr, b.size() * 8, hipMemcpyDeviceToHost));
        double md = 0;
        for (int c = 0; c < u; c++) for (int r = c; r < u; r++) md = fmax(md, fabs(a[r + (size_t)c * ldc] - b[r + (size_t)c * ldc]));
        // upper triangle untouched?
        double mu = 0;
        for (int c = 0; c < u; c++) for (int r = 0; r < c; r++) mu = fmax(mu, fabs(a[r + (size_t)c * ldc] - hC[r + (size_t)c * ldc]));
        printf("   %s: max diff %.3e, upper-triangle change %.3e\n", name, md, mu);
    };
    if (check) {
        CHK(hipMemcpy(Cr, hC.data(), hC.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_ref, dim3((u + 15) / 16, (u + 15) / 16), dim3(16, 16), 0, 0, X, ldx, K, Cr, ldc, u);
        CHK(hipDeviceSynchronize());
    }
#define RUN_LDS(SM, SN, KC, MINW) do { \
        constexpr int TR = 32 * SM, TC = 32 * SN; \
        const int T = (u + TR - 1) / TR; \
        dim3 grid = TR == TC ? dim3((unsigned)(T * (T + 1) / 2)) : dim3((unsigned)T, (unsigned)((u + TC - 1) / TC)); \
        CHK(hipMemcpy(C, hC.data(), hC.size() * 8, hipMemcpyHostToDevice)); \
        hipLaunchKernelGGL((k_syrk_lds<SM, SN, KC, MINW>), grid, dim3(256), 0, 0, X, ldx, K, C, ldc, u, T); \
        CHK(hipDeviceSynchronize()); \
        verify("lds " #SM "x" #SN " KC" #KC); \
        float ms = time_it([&] { hipLaunchKernelGGL((k_syrk_lds<SM, SN, KC, MINW>), grid, dim3(256), 0, 0, X, ldx, K, C, ldc, u, T); }, 10); \
        printf("lds tile %dx%d KC=%d minw=%d: %.3f ms  %.1f TF/s (%.1f%% of 78.6)\n", TR, TC, KC, MINW, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 78.6 * 100); \
    } while (0)
    printf("u = %d, K = %d, %.2f GF\n", u, K, flops * 1e-9);
    RUN_LDS(4, 4, 16, 2);
    RUN_LDS(4, 4, 8, 2);
    RUN_LDS(4, 4, 16, 1);
    RUN_LDS(2, 2, 16, 2);
    RUN_LDS(2, 2, 32, 2);
    RUN_LDS(2, 2, 16, 4);
    RUN_LDS(4, 2, 16, 2);
    {
        const int T = (u + 63) / 64;
        CHK(hipMemcpy(C, hC.data(), hC.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_syrk_direct<64>, dim3(T, T), dim3(256), 0, 0, X, ldx, K, C, ldc, u);
        CHK(hipDeviceSynchronize());
        verify("direct");
        float ms = time_it([&] { hipLaunchKernelGGL(k_syrk_direct<64>, dim3(T, T), dim3(256), 0, 0, X, ldx, K, C, ldc, u); }, 10);
        printf("direct 64x64 (one pass, K=%d): %.3f ms  %.1f TF/s (%.1f%%)\n", K, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 78.6 * 100);
    }
    return 0;
}
