// Developer probe: the fixed cost of a hipGraph REPLAY (≈ 8 us on top of the per-node cost, tools/kbench/graph_nodes.cpp) -- does it go
// away when consecutive replays alternate between two / four instantiations of the same graph, when the graph holds K copies of the
// chain, or with hipGraphUpload?  34 dependent small kernels per "step", like BASELINE configs[1].
//   hipcc --offload-arch=gfx950 -O3 tools/kbench/graph_replay_gap.cpp -o tools/kbench/bin/graph_replay_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k(float *p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
int main() {
    float *p; CK(hipMalloc(&p, 1 << 20)); CK(hipMemset(p, 0, 1 << 20));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int N = 34;
    for (int copies : {1, 2, 4}) {
        for (int nexec : {1, 2, 4}) {
            hipGraph_t g;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int i = 0; i < N * copies; ++i) hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, st, p, 16384);
            CK(hipStreamEndCapture(st, &g));
            std::vector<hipGraphExec_t> ge(nexec);
            for (auto &e : ge) CK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
            for (int i = 0; i < 64; ++i) CK(hipGraphLaunch(ge[i % nexec], st));
            std::vector<double> r;
            const int reps = 512 / copies;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(a, st));
                for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge[i % nexec], st));
                CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b)); r.push_back(ms * 1e3 / (reps * copies));
            }
            std::sort(r.begin(), r.end());
            printf("%d chain(s) of %d nodes per graph, %d instantiation(s) in turn: %7.2f us per chain  (%5.2f us/node)\n", copies, N, nexec, r[2], r[2] / N);
            for (auto &e : ge) CK(hipGraphExecDestroy(e));
            CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
