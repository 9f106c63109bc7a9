// Developer micro-benchmark: the glimpse read at out-of-cache sizes against a pure streaming kernel with the SAME traffic
// (read HW floats per image once, write T*hw floats per image) -- the practical ceiling for this access pattern on the box.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../../attend_infer_repeat_amd/csrc/st_kernels.hip"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// same traffic, no arithmetic: every workgroup streams its images through registers and writes T*hw floats per image
__global__ __launch_bounds__(256) void stream_like_read(const float4 *__restrict__ img, float4 *__restrict__ out, int n_img, int nq_in, int nq_out) {
    for (int b = blockIdx.x; b < n_img; b += gridDim.x) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = threadIdx.x; q < nq_in; q += 256) { const float4 v = img[(size_t)b * nq_in + q]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        for (int q = threadIdx.x; q < nq_out; q += 256) out[(size_t)b * nq_out + q] = acc;
    }
}
template <typename F> static double time_us(F fn, int reps, hipStream_t st) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) fn();
    std::vector<double> r;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(a, st)); for (int i = 0; i < reps; ++i) fn(); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); r.push_back(ms * 1e3 / reps);
    }
    std::sort(r.begin(), r.end()); return r[r.size() / 2];
}
int main(int argc, char **argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 50, W = H, h = argc > 2 ? atoi(argv[2]) : 20, w = h, T = argc > 3 ? atoi(argv[3]) : 3;
    const int HW = H * W, hw = h * w;
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int n_img : {8192, 65536, 196608}) {
        const size_t n = (size_t)n_img * T;
        if ((size_t)n_img * HW * 4 + n * hw * 4 > (size_t)12 << 30) continue;
        float *img, *where, *out; CK(hipMalloc(&img, (size_t)n_img * HW * 4)); CK(hipMalloc(&where, n * 16)); CK(hipMalloc(&out, n * hw * 4));
        std::vector<float> hwh(n * 4); srand(2);
        for (size_t k = 0; k < n; ++k) { hwh[4 * k] = 0.45f + 0.2f * rand() / RAND_MAX; hwh[4 * k + 1] = 0.6f * rand() / RAND_MAX - 0.3f; hwh[4 * k + 2] = 0.45f + 0.2f * rand() / RAND_MAX; hwh[4 * k + 3] = 0.6f * rand() / RAND_MAX - 0.3f; }
        CK(hipMemcpy(where, hwh.data(), n * 16, hipMemcpyHostToDevice)); CK(hipMemset(img, 0x3c, (size_t)n_img * HW * 4));
        const double bytes = 4.0 * ((double)n_img * HW + (double)n * (hw + 4));
        const double us = time_us([&] { air_st_read_fwd(img, where, out, (int)n, n_img, H, W, h, w, st); }, 10, st);
        const double us_c = time_us([&] { hipLaunchKernelGGL(stream_like_read, dim3(2048), dim3(256), 0, st, (const float4 *)img, (float4 *)out, n_img, HW / 4, T * hw / 4); }, 10, st);
        printf("%dx%d/%dx%d T=%d images %7d: read kernel %8.2f us = %6.3f TB/s (minimal bytes %.3f GB) | same-traffic stream %8.2f us = %6.3f TB/s | ratio %.3f\n",
               H, W, h, w, T, n_img, us, bytes / us * 1e-6, bytes * 1e-9, us_c, bytes / us_c * 1e-6, us_c / us);
        CK(hipFree(img)); CK(hipFree(where)); CK(hipFree(out));
    }
    return 0;
}
