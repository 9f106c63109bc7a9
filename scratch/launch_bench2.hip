#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct Big { float* p; int n; int pad[250]; };
template <int ID> __global__ void kid(float* p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + ID; }
__global__ void kbig(Big b) { int i = blockIdx.x * 256 + threadIdx.x; if (i < b.n) b.p[i] = b.p[i] * 1.0001f + b.pad[7]; }
__global__ void klds(float* p, int n) { __shared__ float s[8192]; int i = blockIdx.x * 256 + threadIdx.x; s[threadIdx.x] = p[i % n]; __syncthreads(); if (i < n) p[i] = s[(threadIdx.x + 1) & 255] + 1.f; }
// producer/consumer on distinct buffers: each node reads what previous wrote (64 blocks -> spread over XCDs)
__global__ void kdep(const float* __restrict__ in, float* __restrict__ out, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) out[i] = in[(i * 7 + 13) % n] + 1.f; }
typedef void (*launcher)(hipStream_t, float*, float*, int, int);
int main() {
    float *d = nullptr; float *d2 = nullptr; CK(hipMalloc((void**)&d, 1 << 24)); CK(hipMalloc((void**)&d2, 1 << 24));
    CK(hipMemset(d, 0, 1 << 24)); CK(hipMemset(d2, 0, 1 << 24));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 64, n = grid * 256;
    for (int mode = 0; mode < 5; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < 96; ++i) {
            if (mode == 0) hipLaunchKernelGGL(kid<0>, dim3(grid), dim3(256), 0, s, d, n);
            if (mode == 1) { switch (i % 8) {
                case 0: hipLaunchKernelGGL(kid<0>, dim3(grid), dim3(256), 0, s, d, n); break;
                case 1: hipLaunchKernelGGL(kid<1>, dim3(grid), dim3(256), 0, s, d, n); break;
                case 2: hipLaunchKernelGGL(kid<2>, dim3(grid), dim3(256), 0, s, d, n); break;
                case 3: hipLaunchKernelGGL(kid<3>, dim3(grid), dim3(256), 0, s, d, n); break;
                case 4: hipLaunchKernelGGL(kid<4>, dim3(grid), dim3(256), 0, s, d, n); break;
                case 5: hipLaunchKernelGGL(kid<5>, dim3(grid), dim3(256), 0, s, d, n); break;
                case 6: hipLaunchKernelGGL(kid<6>, dim3(grid), dim3(256), 0, s, d, n); break;
                default: hipLaunchKernelGGL(kid<7>, dim3(grid), dim3(256), 0, s, d, n); } }
            if (mode == 2) { Big b; b.p = d; b.n = n; b.pad[7] = 1; hipLaunchKernelGGL(kbig, dim3(grid), dim3(256), 0, s, b); }
            if (mode == 3) hipLaunchKernelGGL(klds, dim3(grid), dim3(256), 0, s, d, n);
            if (mode == 4) { if (i & 1) hipLaunchKernelGGL(kdep, dim3(grid), dim3(256), 0, s, d2, d, n); else hipLaunchKernelGGL(kdep, dim3(grid), dim3(256), 0, s, d, d2, n); }
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const char* names[] = {"same tiny kernel", "8 distinct kernels", "1KB kernarg", "32KB LDS kernel", "producer->consumer gather"};
        printf("mode %d %-28s: %.2f us/node\n", mode, names[mode], ms * 1e3 / (96 * 50));
    }
    return 0;
}
