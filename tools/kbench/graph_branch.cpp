// Developer probe: does a hipGraph with a forked side branch overlap it with the main chain on MI355X, and what does a
// cross-branch edge cost?  Main chain of NA kernels (each ~D us of spin on 64 workgroups), side branch of NB kernels forked
// after node F and joined before the last node; variants: one fork / one join, or one cross edge per side node.
//   hipcc --offload-arch=gfx950 -O3 tools/kbench/graph_branch.cpp -o tools/kbench/bin/graph_branch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void spin(float *p, int ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0) p[blockIdx.x] += 1.f;
}
static double replay_us(hipGraphExec_t ge, hipStream_t st, hipEvent_t a, hipEvent_t b) {
    for (int i = 0; i < 30; ++i) hipGraphLaunch(ge, st);
    std::vector<double> r;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a, st);
        for (int i = 0; i < 300; ++i) hipGraphLaunch(ge, st);
        hipEventRecord(b, st); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); r.push_back(ms * 1e3 / 300);
    }
    std::sort(r.begin(), r.end());
    return r[2];
}
int main(int argc, char **argv) {
    const int ticks = argc > 1 ? atoi(argv[1]) : 300;        // 100 MHz counter: 300 = 3 us
    const int wgs = argc > 2 ? atoi(argv[2]) : 64;
    float *p, *q; CK(hipMalloc(&p, 1 << 20)); CK(hipMemset(p, 0, 1 << 20)); CK(hipMalloc(&q, 1 << 20)); CK(hipMemset(q, 0, 1 << 20));
    hipStream_t st, side; CK(hipStreamCreate(&st)); CK(hipStreamCreate(&side));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<hipEvent_t> ev(64);
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const int NA = 30;
    for (int NB : {0, 5, 10, 20}) {
        for (int mode = 0; mode < (NB ? 3 : 1); ++mode) {     // 0: fork once / join once; 1: every side node waits for a main node; 2: linear (side nodes appended to the chain)
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            int sb = 0;
            for (int i = 0; i < NA; ++i) {
                hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, st, p, ticks);
                if (mode == 0 && NB && i == 4) {
                    CK(hipEventRecord(ev[0], st)); CK(hipStreamWaitEvent(side, ev[0], 0));
                    for (int j = 0; j < NB; ++j) hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, side, q, ticks);
                    CK(hipEventRecord(ev[1], side));
                }
                if (mode == 1 && NB && i >= 4 && sb < NB) {
                    CK(hipEventRecord(ev[2 + sb], st)); CK(hipStreamWaitEvent(side, ev[2 + sb], 0));
                    hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, side, q, ticks);
                    ++sb;
                    if (sb == NB) CK(hipEventRecord(ev[1], side));
                }
                if (mode != 2 && NB && i == NA - 2) CK(hipStreamWaitEvent(st, ev[1], 0));
            }
            if (mode == 2) for (int j = 0; j < NB; ++j) hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, st, q, ticks);
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            const double us = replay_us(ge, st, a, b);
            const char *names[] = {"fork/join once", "edge per side node", "all linear"};
            printf("main %d + side %2d nodes, %-18s: %8.2f us/replay  (%5.2f us per main node)\n", NA, NB, names[mode], us, us / NA);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
