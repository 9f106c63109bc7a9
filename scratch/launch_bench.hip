#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void mid(float* p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    float* d; CK(hipMalloc(&d, 1 << 24));
    CK(hipMemset(d, 0, 1 << 24));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 2000;
    for (int grid : {1, 64, 256}) {
        // eager
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(mid, dim3(grid), dim3(256), 0, s, d, grid * 256);
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::high_resolution_clock::now();
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(mid, dim3(grid), dim3(256), 0, s, d, grid * 256);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        auto t1 = std::chrono::high_resolution_clock::now();
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("eager  grid %4d: %.2f us/kernel (events), %.2f us/kernel (wall)\n", grid, ms * 1e3 / N,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
        // graph of 100 nodes
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(mid, dim3(grid), dim3(256), 0, s, d, grid * 256);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("graph  grid %4d: %.2f us/kernel (100-node graph x50)\n", grid, ms * 1e3 / 5000);
        // graph with 4 parallel branches of 25
        hipStream_t br[3]; hipEvent_t fork, join[3];
        CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        for (int b = 0; b < 3; ++b) { CK(hipStreamCreate(&br[b])); CK(hipEventCreateWithFlags(&join[b], hipEventDisableTiming)); }
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        CK(hipEventRecord(fork, s));
        for (int b = 0; b < 3; ++b) CK(hipStreamWaitEvent(br[b], fork, 0));
        for (int i = 0; i < 25; ++i) {
            hipLaunchKernelGGL(mid, dim3(grid), dim3(256), 0, s, d, grid * 256);
            for (int b = 0; b < 3; ++b) hipLaunchKernelGGL(mid, dim3(grid), dim3(256), 0, br[b], d + (b + 1) * (1 << 20), grid * 256);
        }
        for (int b = 0; b < 3; ++b) { CK(hipEventRecord(join[b], br[b])); CK(hipStreamWaitEvent(s, join[b], 0)); }
        hipGraph_t g2; hipGraphExec_t ge2;
        CK(hipStreamEndCapture(s, &g2));
        CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge2, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge2, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("graph4 grid %4d: %.2f us/kernel (4 branches x 25 nodes, x50)\n", grid, ms * 1e3 / 5000);
    }
    return 0;
}
