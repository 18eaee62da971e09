// FP64 ceilings of MI355X measured in registers: v_mfma_f64_16x16x4_f64 and v_fma_f64 alone and interleaved, 1 / 2 / 4 waves per SIMD,
// cycles by s_memtime and the clock held (s_memtime / s_memrealtime).  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fp64_peak.hip -o /tmp/fp64_peak
// Results of round 4: profiles/r04_fp64_peak.txt (MFMA 99 cycles per instruction per SIMD with two or more waves, 130 with one: 50.7 TF/s;
// vector FMA 54 TF/s at the 1.95 GHz the chip holds under it; interleaved 56 TF/s) -- the nominal 78.6 TF/s is not reachable.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
// NM MFMAs + NV vector FMAs per iteration, all independent chains
template <int NM, int NV>
__global__ __launch_bounds__(256) void k_mix(double *out, unsigned long long *st, int iters, const double *in)
{
    d4 acc[NM > 0 ? NM : 1];
    double v[NV > 0 ? NV : 1];
    for (int i = 0; i < (NM > 0 ? NM : 1); i++) acc[i] = (d4){0, 0, 0, 0};
    double a = in[threadIdx.x], b = in[256 + threadIdx.x];
    for (int i = 0; i < (NV > 0 ? NV : 1); i++) v[i] = a * i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NM; i++) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV / (NM > 0 ? NM : 1); j++) { const int q = i * (NV / (NM > 0 ? NM : 1)) + j; v[q] = __builtin_fma(a, b, v[q]); }
        }
        if (NM == 0) {
#pragma unroll
            for (int j = 0; j < NV; j++) v[j] = __builtin_fma(a, b, v[j]);
        }
    }
    double s = 0;
    for (int i = 0; i < NM; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < NV; i++) s += v[i];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = t1 - t0; st[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int NM, int NV>
void run(int wg_per_cu, double *out, unsigned long long *st, const double *in, int iters)
{
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_mix<NM, NV>), dim3(grid), dim3(256), 0, 0, out, st, iters, in);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mix<NM, NV>), dim3(grid), dim3(256), 0, 0, out, st, iters, in);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * grid);
    CHK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc, clk;
    for (int i = 0; i < grid; i++) { cyc.push_back((double)h[2 * i]); clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double fl = (double)grid * 4 * iters * (NM * 2048.0 + NV * 128.0);
    printf("NM=%2d NV=%2d waves/SIMD=%d: %.3f ms %.1f TF/s; cycles per iteration per wave %.1f, clock %.0f MHz\n", NM, NV, wg_per_cu, ms, fl / ms * 1e-9,
           cyc[grid / 2] / (double)iters, clk[grid / 2]);
}
int main()
{
    double *out, *in; unsigned long long *st;
    CHK(hipMalloc(&out, 256 * 8 * 256 * 8)); CHK(hipMalloc(&st, 2 * 8 * 2048 * 8)); CHK(hipMalloc(&in, 512 * 8));
    std::vector<double> h(512);
    for (auto &v : h) v = (rand() % 2001 - 1000) / 1000.0;
    CHK(hipMemcpy(in, h.data(), 512 * 8, hipMemcpyHostToDevice));
    const int it = 20000;
    printf("-- v_fma_f64 alone: NV independent chains per wave (128 flops per instruction)\n");
    run<0, 32>(1, out, st, in, it); run<0, 32>(2, out, st, in, it); run<0, 32>(4, out, st, in, it);
    printf("-- v_mfma_f64_16x16x4_f64 alone: NM independent accumulators per wave\n");
    run<4, 0>(1, out, st, in, it); run<4, 0>(2, out, st, in, it); run<4, 0>(4, out, st, in, it); run<16, 0>(1, out, st, in, it); run<16, 0>(2, out, st, in, it);
    printf("-- interleaved\n");
    run<4, 32>(1, out, st, in, it); run<4, 32>(2, out, st, in, it);
    run<4, 64>(1, out, st, in, it); run<4, 64>(2, out, st, in, it);
    run<4, 96>(1, out, st, in, it); run<4, 96>(2, out, st, in, it);
    return 0;
}
