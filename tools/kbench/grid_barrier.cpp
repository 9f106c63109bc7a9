// Developer probe: what does a device-wide barrier INSIDE one kernel cost on MI355X next to a kernel boundary in a hipGraph?
// A chain of P dependent phases -- every workgroup reads 16 KB that OTHER workgroups (other XCDs) wrote in the previous phase,
// reduces it and writes 16 KB of its own -- run (a) as P graph nodes and (b) as ONE kernel with P-1 barriers (monotonic 64-bit
// arrival counter, agent-scope release / acquire, bounded spin: a barrier that is not met within ~2 ms raises a flag and the
// kernel returns instead of hanging).  Results are checked against the closed form, so a coherence hole shows as "MISMATCH".
//   hipcc --offload-arch=gfx950 -O3 tools/kbench/grid_barrier.cpp -o tools/kbench/bin/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int THREADS = 256;
constexpr int PER_WG = THREADS * 4 * 4;           // floats per workgroup and phase (16 KB)

__device__ __forceinline__ void phase_work(const float *__restrict__ src, float *__restrict__ dst, int phase, int grid) {
    // read the block a workgroup three XCDs away wrote (round-robin XCD placement: neighbour indices sit on other XCDs)
    const int from = (blockIdx.x + 3 + phase) % grid;
    const float4 *s = reinterpret_cast<const float4 *>(src + (size_t)from * PER_WG);
    float4 *d = reinterpret_cast<float4 *>(dst + (size_t)blockIdx.x * PER_WG);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float4 v = s[i * THREADS + threadIdx.x];
        v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
        d[i * THREADS + threadIdx.x] = v;
    }
}

__global__ void phase_kernel(const float *src, float *dst, int phase, int grid) { phase_work(src, dst, phase, grid); }

__device__ __forceinline__ bool grid_barrier(unsigned long long *ctr, unsigned long long target, int *flag) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > 200000ull) { *flag = 1; break; }     // 100 MHz: 2 ms
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}

__global__ void fused_kernel(float *a, float *b, int phases, unsigned long long *ctr, int *flag) {
    const int grid = gridDim.x;
    __shared__ unsigned long long base;
    if (threadIdx.x == 0) {
        const unsigned long long per_launch = (unsigned long long)(phases - 1) * grid;
        const unsigned long long c = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        base = per_launch ? c / per_launch * per_launch : 0;
    }
    __syncthreads();
    unsigned long long target = base;
    float *src = a, *dst = b;
    for (int p = 0; p < phases; ++p) {
        phase_work(src, dst, p, grid);
        float *t = src; src = dst; dst = t;
        if (p + 1 < phases) { target += grid; grid_barrier(ctr, target, flag); }
    }
}

static double time_us(hipGraphExec_t ge, hipStream_t st, hipEvent_t a, hipEvent_t b) {
    for (int i = 0; i < 30; ++i) hipGraphLaunch(ge, st);
    std::vector<double> r;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a, st);
        for (int i = 0; i < 200; ++i) hipGraphLaunch(ge, st);
        hipEventRecord(b, st); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); r.push_back(ms * 1e3 / 200);
    }
    std::sort(r.begin(), r.end());
    return r[2];
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t ea, eb; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    float *a, *b; unsigned long long *ctr; int *flag;
    const int MAXG = 1024;
    CK(hipMalloc(&a, (size_t)MAXG * PER_WG * 4)); CK(hipMalloc(&b, (size_t)MAXG * PER_WG * 4));
    CK(hipMalloc(&ctr, 256)); CK(hipMemset(ctr, 0, 256)); CK(hipMalloc(&flag, 4)); CK(hipMemset(flag, 0, 4));
    std::vector<float> host((size_t)MAXG * PER_WG);
    printf("grid phases  chain_us  fused_us  per_boundary_chain  per_barrier_fused  check\n");
    for (int grid : {64, 192, 256, 512, 1024}) {
        for (int phases : {1, 2, 4, 8}) {
            // (a) chain of kernels
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            { float *s = a, *d = b; for (int p = 0; p < phases; ++p) { hipLaunchKernelGGL(phase_kernel, dim3(grid), dim3(THREADS), 0, st, s, d, p, grid); std::swap(s, d); } }
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            const double chain = time_us(ge, st, ea, eb);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            // (b) one kernel
            CK(hipMemset(ctr, 0, 256));
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            hipLaunchKernelGGL(fused_kernel, dim3(grid), dim3(THREADS), 0, st, a, b, phases, ctr + (phases % 8) * 2, flag);
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            const double fused = time_us(ge, st, ea, eb);
            // correctness: zero the input, one launch, every element of the final buffer must equal `phases`
            CK(hipMemsetAsync(a, 0, (size_t)MAXG * PER_WG * 4, st)); CK(hipMemsetAsync(b, 0, (size_t)MAXG * PER_WG * 4, st));
            int bad = 0;
            for (int rep = 0; rep < 20; ++rep) {
                CK(hipMemsetAsync(a, 0, (size_t)grid * PER_WG * 4, st));
                CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
                CK(hipMemcpy(host.data(), (phases & 1) ? b : a, (size_t)grid * PER_WG * 4, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < (size_t)grid * PER_WG; ++i) bad += host[i] != (float)phases;
            }
            int f; CK(hipMemcpy(&f, flag, 4, hipMemcpyDeviceToHost));
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            printf("%4d %6d  %8.2f  %8.2f  %18.2f  %17.2f  %s%s\n", grid, phases, chain, fused,
                   phases > 1 ? (chain - 0) / phases : chain, phases > 1 ? (fused - 0) / phases : fused,
                   bad ? "MISMATCH" : "ok", f ? " TIMEOUT-FLAG" : "");
            if (f) { CK(hipMemset(flag, 0, 4)); }
        }
    }
    return 0;
}
