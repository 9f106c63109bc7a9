// Developer probe: replay time of a linear hipGraph of N small dependent kernels, N = 24..72 -- is there a per-batch bubble?
//   hipcc --offload-arch=gfx950 -O3 tools/kbench/graph_nodes.cpp -o tools/kbench/bin/graph_nodes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k(float *p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
struct Big { float *p; int n; int pad[250]; };       // ~1 KB of kernel arguments, like a grouped-GEMM launch
__global__ void kbig(Big a) { int i = blockIdx.x * 256 + threadIdx.x; if (i < a.n) a.p[i] = a.p[i] * 1.0001f + 1.f + (float)a.pad[blockIdx.x & 127]; }
int main(int argc, char **argv) {
    const bool big = argc > 1;
    float *p; CK(hipMalloc(&p, 1 << 20)); CK(hipMemset(p, 0, 1 << 20));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    double prev = 0;
    for (int N = 24; N <= 72; N += (big ? 2 : 1)) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < N; ++i) { if (big) { Big a{}; a.p = p; a.n = 16384; hipLaunchKernelGGL(kbig, dim3(64), dim3(256), 0, st, a); } else hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, st, p, 16384); }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        if (getenv("GRAPH_UPLOAD")) CK(hipGraphUpload(ge, st));
        for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, st));
        std::vector<double> r;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(a, st));
            for (int i = 0; i < 500; ++i) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); r.push_back(ms * 1e3 / 500);
        }
        std::sort(r.begin(), r.end());
        printf("N=%2d  %7.2f us/replay  (+%5.2f)  %5.2f us/node\n", N, r[2], r[2] - prev, r[2] / N);
        prev = r[2];
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
