// Developer probe (not part of the product): is the stored-canvas backward of the inverse write, in the throughput regime, bound by
// latency (time ~ 1 / resident workgroups) or by a saturated pipe (time flat in occupancy)?  The shipped kernel is launched with extra,
// unused dynamic LDS so that 5 (as shipped: 96 VGPRs), 4, 3, 2, 1 workgroups of 256 threads fit a CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kbench/canvas_occ.cpp -o tools/kbench/bin/canvas_occ
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include "../../attend_infer_repeat_amd/csrc/canvas_kernels.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static float frand() { return (float)rand() / (float)RAND_MAX; }
template <typename T> static T *dev(const std::vector<T> &v) {
    T *p; CK(hipMalloc(&p, v.size() * sizeof(T))); CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return p;
}
static float *devz(size_t n) { float *p; CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 0, n * 4)); return p; }
template <typename F> static double time_us(F fn, int reps, hipStream_t st) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) fn();
    std::vector<double> r;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(a, st));
        for (int i = 0; i < reps; ++i) fn();
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); r.push_back(ms * 1e3 / reps);
    }
    std::sort(r.begin(), r.end());
    return r[r.size() / 2];
}
int main(int argc, char **argv) {
    const int T = 3, H = 50, W = 50, h = 20, w = 20;
    const int B = argc > 1 ? atoi(argv[1]) : 8192;
    const float lo = argc > 2 ? atof(argv[2]) : 0.45f, hi = argc > 3 ? atof(argv[3]) : 0.65f;
    const int n = T * B, HW = H * W, hw = h * w;
    srand(1);
    std::vector<float> glm((size_t)n * hw), where((size_t)n * 4), pres(n), obs((size_t)B * HW), fin((size_t)B * HW);
    for (auto &v : glm) v = frand() - 0.5f;
    for (int k = 0; k < n; ++k) {
        where[4 * k] = lo + (hi - lo) * frand(); where[4 * k + 2] = lo + (hi - lo) * frand();
        where[4 * k + 1] = 0.6f * frand() - 0.3f; where[4 * k + 3] = 0.6f * frand() - 0.3f;
        pres[k] = frand() < 0.7f ? 1.f : 0.f;
    }
    for (auto &v : obs) v = frand();
    for (auto &v : fin) v = frand();
    float *d_glm = dev(glm), *d_where = dev(where), *d_pres = dev(pres), *d_obs = dev(obs), *d_fin = dev(fin);
    float *d_dgl = devz((size_t)n * hw), *d_dwh = devz((size_t)n * 4);
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t lds0 = carve_bwd_bytes(H, W, h, w, 1);
    const WriteBwdArgs a = {d_glm, d_where, d_pres, nullptr, d_fin, d_obs, d_dgl, d_dwh, nullptr, T, B, H, W, h, w,
                            lin_step(W), lin_step(H), 1.0f, 0.3f, 1.0f / B, 1, 1, 1};
    const NvilArgs nv = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1, nullptr};
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(st_write_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    printf("stored-canvas backward, %d images x %d glimpses, scales %.2f-%.2f, kernel's own LDS %zu B; 256-thread workgroups, grid %d\n", B, T, lo, hi, lds0,
           cv_grid((long)T * B, 256 * 8));
    const int wgs[] = {5, 4, 3, 2, 1};
    for (int k = 0; k < 5; ++k) {
        // largest LDS request that still lets `wgs[k]` workgroups share 160 KiB (k = 0: the shipped request)
        size_t lds = k == 0 ? lds0 : (size_t)(160 * 1024 / wgs[k]) - 256;
        if (lds < lds0) lds = lds0;
        if (lds > 150 * 1024) lds = 150 * 1024;
        for (int grid_mode = 0; grid_mode < 2; ++grid_mode) {
            const int grid = grid_mode == 0 ? cv_grid((long)T * B, 256 * 8) : cv_grid((long)T * B, 256 * wgs[k]);
            auto fn = [&]() { hipLaunchKernelGGL(st_write_bwd_kernel<false>, dim3(grid), dim3(256), lds, st, a, nv); };
            const double us = time_us(fn, B >= 16384 ? 5 : 20, st);
            printf("  <= %d workgroups / CU (LDS %6zu B, grid %5d): %9.2f us\n", wgs[k], lds, grid, us);
        }
    }
    // wider grid: one unit per workgroup (no grid-stride loop)
    {
        auto fn = [&]() { hipLaunchKernelGGL(st_write_bwd_kernel<false>, dim3(n), dim3(256), lds0, st, a, nv); };
        printf("  one unit per workgroup (grid %d): %9.2f us\n", n, time_us(fn, B >= 16384 ? 5 : 20, st));
    }
    return 0;
}
