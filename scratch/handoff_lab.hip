// lab: what does it cost to hand work from one workgroup to the next through a flag in HBM, against a kernel boundary?
//   chain<<<W>>>: workgroup i waits (bounded spin) for flag[i - 1], touches a cache line, releases flag[i]
//   launches:    W dependent launches of a one-workgroup kernel, directly and replayed from a hipGraph
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/handoff_lab.hip -o /tmp/handoff_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k_chain(int *flag, double *data, int *timeout)
{
    const int b = blockIdx.x;
    if (b > 0 && threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(&flag[b - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) { atomicAdd(timeout, 1); break; }       // bounded: never hangs
        }
    }
    __syncthreads();
    data[b * 32 + (threadIdx.x & 31)] = (b > 0 ? data[(b - 1) * 32 + (threadIdx.x & 31)] : 0.0) + 1.0;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&flag[b], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void k_one(double *data, int b)
{
    data[b * 32 + (threadIdx.x & 31)] = (b > 0 ? data[(b - 1) * 32 + (threadIdx.x & 31)] : 0.0) + 1.0;
}

int main()
{
    const int W = 200;
    int *flag, *to;
    double *data;
    hipMalloc(&flag, W * sizeof(int)); hipMalloc(&to, sizeof(int)); hipMalloc(&data, W * 32 * sizeof(double));
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipMemsetAsync(flag, 0, W * sizeof(int), st); hipMemsetAsync(to, 0, sizeof(int), st); hipMemsetAsync(data, 0, W * 32 * sizeof(double), st);
        hipEventRecord(e0, st);
        hipLaunchKernelGGL(k_chain, dim3(W), dim3(256), 0, st, flag, data, to);
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        hipEventElapsedTime(&ms, e0, e1);
    }
    int h_to = 0; double last = 0;
    hipMemcpy(&h_to, to, sizeof(int), hipMemcpyDeviceToHost);
    hipMemcpy(&last, data + (W - 1) * 32, sizeof(double), hipMemcpyDeviceToHost);
    printf("flag chain through %d workgroups: %.1f us total, %.2f us per hand-off (timeouts %d, last value %.0f)\n", W, ms * 1e3, ms * 1e3 / W, h_to, last);
    // dependent launches, direct
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, st);
        for (int b = 0; b < W; b++) hipLaunchKernelGGL(k_one, dim3(1), dim3(256), 0, st, data, b);
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%d dependent launches (stream): %.1f us total, %.2f us per launch\n", W, ms * 1e3, ms * 1e3 / W);
    // the same from a graph
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int b = 0; b < W; b++) hipLaunchKernelGGL(k_one, dim3(1), dim3(256), 0, st, data, b);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, st);
        hipGraphLaunch(ge, st);
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%d dependent launches (graph replay): %.1f us total, %.2f us per launch\n", W, ms * 1e3, ms * 1e3 / W);
    return 0;
}
