// Developer micro-benchmark for the dense products of the throughput regime (batch 1024): the shipped dispatch (air_gemm_grouped
// with one problem) against explicit kernel variants, same data, results compared.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kbench/gemm_bench.cpp attend_infer_repeat_amd/csrc/loss_kernels.hip -o tools/kbench/bin/gemm_bench
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include "../../attend_infer_repeat_amd/csrc/gemm_kernels.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static float frand() { return (float)rand() / (float)RAND_MAX * 2.f - 1.f; }
static float *devr(size_t n) { std::vector<float> v(n); for (auto &x : v) x = frand(); float *p; CK(hipMalloc(&p, n * 4)); CK(hipMemcpy(p, v.data(), n * 4, hipMemcpyHostToDevice)); return p; }
static float *devz(size_t n) { float *p; CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 0, n * 4)); return p; }
template <typename F> static double time_us(F fn, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 10; ++i) fn();
    std::vector<double> r;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < reps; ++i) fn();
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); r.push_back(ms * 1e3 / reps);
    }
    std::sort(r.begin(), r.end());
    return r[r.size() / 2];
}
struct Shape { int ta, tb, M, N, K; const char *what; int lda_over = 0; };

template <int MT, int KW, bool BF, bool TA, bool TB> static void wide(const GemmArgs &g) {
    const int tiles = air_cdiv(g.M, 16 * MT) * air_cdiv(g.N, 64);
    hipLaunchKernelGGL((gemm_wide_kernel<MT, KW, BF, TA, TB>), dim3(tiles), dim3(64 * KW), 0, 0, g);
}

int main(int argc, char **argv) {
    const int bf = argc > 1 ? atoi(argv[1]) : 0;
    const Shape shapes[] = {
        {0, 0, 3072, 256, 256, "MLP fwd"}, {0, 0, 3072, 256, 400, "glimpse enc l0 fwd"}, {0, 0, 3072, 400, 256, "decoder out fwd"},
        {0, 0, 1024, 256, 2500, "input enc l0 fwd"}, {0, 0, 1024, 1024, 256, "LSTM recurrent fwd"}, {0, 0, 3072, 100, 256, "what head fwd"},
        {0, 1, 3072, 256, 256, "MLP dX"}, {0, 1, 3072, 256, 400, "decoder out dX"}, {0, 1, 1024, 256, 1024, "LSTM dh"}, {0, 1, 3072, 400, 256, "glimpse enc l0 dX"}, {0, 1, 3072, 128, 64, "steps l1 dX"}, {0, 1, 3072, 256, 100, "what dX"},
        {1, 0, 256, 256, 3072, "MLP dW"}, {1, 0, 256, 1024, 3072, "LSTM dWx"}, {1, 0, 2500, 256, 1024, "input enc l0 dW"}, {1, 0, 400, 256, 3072, "glimpse enc l0 dW"}, {1, 0, 1024, 1024, 3072, "square dW (256 tiles: one per CU)"}, {1, 0, 2048, 1024, 3072, "512 tiles"}, {1, 0, 1024, 1024, 1024, "256 tiles, K = 1024"},
        {1, 0, 48, 256, 3072, "decoder l0 dW (48 of 50 rows, lda 50: unaligned 16-byte loads)", 50}, {1, 0, 676, 256, 1024, "baseline latent dW (676 of 677, lda 677)", 677},
    };
    for (const Shape &s : shapes) {
        const int lda = s.lda_over ? s.lda_over : (s.ta ? s.M : s.K), ldb = s.tb ? s.K : s.N;
        float *A = devr((size_t)(s.lda_over ? s.lda_over : s.M) * s.K + 64), *B = devr((size_t)s.N * s.K), *bias = devr(s.N), *C0 = devz((size_t)s.M * s.N), *C1 = devz((size_t)s.M * s.N);
        float *col0 = devz(s.N), *col1 = devz(s.N);
        AirGemmDesc d{};
        d.ta = s.ta; d.tb = s.tb; d.M = s.M; d.N = s.N; d.K = s.K; d.A = A; d.lda = lda; d.B = B; d.ldb = ldb; d.C = C0; d.ldc = s.N;
        d.bias = s.ta ? nullptr : bias; d.epilogue = s.ta ? AIR_EPI_NONE : AIR_EPI_BIAS_ELU; d.colsum = s.ta ? col0 : nullptr;
        d.precision = bf ? AIR_PREC_BF16 : AIR_PREC_F32;
        int st = air_gemm_grouped(&d, 1, nullptr);
        if (st) { printf("shipped dispatch failed: %d\n", st); return 1; }
        const double t0 = time_us([&] { air_gemm_grouped(&d, 1, nullptr); }, 200);
        GemmArgs g; AirGemmDesc d1 = d; d1.C = C1; d1.colsum = s.ta ? col1 : nullptr;
        fill_gemm_args(g, d1);
        struct Var { const char *name; void (*fn)(const GemmArgs &); };
        std::vector<Var> vars;
#define V(MT, KW, TA, TB) vars.push_back({#MT "x" #KW, bf ? &wide<MT, KW, true, TA, TB> : &wide<MT, KW, false, TA, TB>})
        if (!s.ta && !s.tb) { V(1, 4, false, false); V(2, 4, false, false); V(1, 8, false, false); V(2, 8, false, false); V(4, 4, false, false); }
        else if (!s.ta && s.tb) { V(1, 4, false, true); V(2, 4, false, true); V(1, 8, false, true); V(2, 8, false, true); V(4, 4, false, true); V(4, 8, false, true); }
        else { V(4, 4, true, false); V(4, 8, true, false); }
#undef V
        const double gf = 2.0 * s.M * s.N * s.K * 1e-9;
        printf("%-20s %s%s %5dx%5dx%5d  floor %5.2f  shipped %6.2f us |", s.what, s.ta ? "T" : "N", s.tb ? "T" : "N", s.M, s.N, s.K, gf / (bf ? 2500.0 : 157.0) * 1e3 * 1e-3, t0);
        std::vector<float> h0((size_t)s.M * s.N), h1(h0.size());
        CK(hipMemcpy(h0.data(), C0, h0.size() * 4, hipMemcpyDeviceToHost));
        for (const Var &v : vars) {
            CK(hipMemset(C1, 0, h1.size() * 4));
            v.fn(g); CK(hipDeviceSynchronize());
            CK(hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost));
            double mx = 0;
            for (size_t i = 0; i < h0.size(); ++i) mx = std::max(mx, (double)fabsf(h0[i] - h1[i]));
            const double t1 = time_us([&] { v.fn(g); }, 200);
            printf(" %s %6.2f (%.0e)", v.name, t1, mx);
        }
        printf("\n");
        hipFree(A); hipFree(B); hipFree(bias); hipFree(C0); hipFree(C1); hipFree(col0); hipFree(col1);
    }
    return 0;
}
