// shader clock under light load: dependent FMA chain, s_memtime (shader cycles) vs s_memrealtime (100 MHz)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return 1; } } while (0)
__global__ void probe(unsigned long long *out, int iters, float seed) {
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = seed + threadIdx.x;
    for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, 1.0000001f, 0.5f);
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 3] = c1 - c0; out[blockIdx.x * 3 + 1] = r1 - r0; out[blockIdx.x * 3 + 2] = (unsigned long long)x; }
}
int main() {
    unsigned long long *d; CK(hipMalloc(&d, 4096 * 3 * 8));
    for (int iters : {200, 2000, 20000, 200000}) for (int grid : {1, 64, 256, 1024}) {
        for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f); }
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(grid * 3); CK(hipMemcpy(h.data(), d, grid * 3 * 8, hipMemcpyDeviceToHost));
        double cyc = 0, rt = 0; for (int b = 0; b < grid; ++b) { cyc += h[b * 3]; rt += h[b * 3 + 1]; }
        printf("iters %6d grid %4d: %.0f shader cycles in %.2f us -> %.2f GHz; %.2f cycles per dependent fma\n", iters, grid, cyc / grid, rt / grid * 0.01, cyc / (rt * 10.0), cyc / grid / iters);
    }
    return 0;
}
