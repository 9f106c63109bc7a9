// What does this box sustain for "read R bytes + write Wr bytes" streams out of cache?  Reference points for the ST read roofline.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NT_LD, int NT_ST>
__global__ __launch_bounds__(256) void copy_flat(const f4 *__restrict__ in, f4 *__restrict__ out, size_t nq_in, size_t nq_out) {
    // every nq_in/nq_out-th element is written: reads nq_in float4, writes nq_out float4, both fully coalesced across the grid
    const size_t stride = (size_t)gridDim.x * 256;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    size_t qo = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nq_in; q += stride) {
        const f4 v = NT_LD ? __builtin_nontemporal_load(in + q) : in[q];
        acc += v;
        if (qo < nq_out && (q * nq_out) / nq_in >= qo) { if (NT_ST) __builtin_nontemporal_store(acc, out + qo); else out[qo] = acc; qo += stride; }
    }
    if (qo < nq_out) out[qo] = acc;
}
template <int NT_LD>
__global__ __launch_bounds__(256) void read_only(const f4 *__restrict__ in, f4 *__restrict__ out, size_t nq_in) {
    const size_t stride = (size_t)gridDim.x * 256;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nq_in; q += stride) acc += NT_LD ? __builtin_nontemporal_load(in + q) : in[q];
    if (acc.x == 12345.678f) out[0] = acc;
}
template <int NT_LD, int NT_ST>
__global__ __launch_bounds__(256) void per_image(const f4 *__restrict__ img, f4 *__restrict__ out, int n_img, int nq_in, int nq_out) {
    for (int b = blockIdx.x; b < n_img; b += gridDim.x) {
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int q = threadIdx.x; q < nq_in; q += 256) acc += NT_LD ? __builtin_nontemporal_load(img + (size_t)b * nq_in + q) : img[(size_t)b * nq_in + q];
        for (int q = threadIdx.x; q < nq_out; q += 256) { if (NT_ST) __builtin_nontemporal_store(acc, out + (size_t)b * nq_out + q); else out[(size_t)b * nq_out + q] = acc; }
    }
}
template <typename F> static double time_us(F fn, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) fn();
    std::vector<double> r;
    for (int rep = 0; rep < 5; ++rep) { CK(hipEventRecord(a, 0)); for (int i = 0; i < reps; ++i) fn(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); r.push_back(ms * 1e3 / reps); }
    std::sort(r.begin(), r.end()); return r[r.size() / 2];
}
int main() {
    const int n_img = 65536, nq_in = 625, nq_out = 300;
    const size_t NI = (size_t)n_img * nq_in, NO = (size_t)n_img * nq_out;
    f4 *in, *out; CK(hipMalloc(&in, NI * 16)); CK(hipMalloc(&out, NO * 16)); CK(hipMemset(in, 0x3c, NI * 16));
    const double bytes = (NI + NO) * 16.0;
    auto rep = [&](const char *n, double us, double by) { printf("%-44s %8.2f us  %6.3f TB/s\n", n, us, by / us * 1e-6); };
    for (int grid : {1024, 2048, 4096, 16384}) {
        printf("grid %d\n", grid);
        rep("  per-image  (ld, st)", time_us([&] { hipLaunchKernelGGL((per_image<0, 0>), dim3(grid), dim3(256), 0, 0, in, out, n_img, nq_in, nq_out); }, 10), bytes);
        rep("  per-image  (nt ld, st)", time_us([&] { hipLaunchKernelGGL((per_image<1, 0>), dim3(grid), dim3(256), 0, 0, in, out, n_img, nq_in, nq_out); }, 10), bytes);
        rep("  per-image  (nt ld, nt st)", time_us([&] { hipLaunchKernelGGL((per_image<1, 1>), dim3(grid), dim3(256), 0, 0, in, out, n_img, nq_in, nq_out); }, 10), bytes);
        rep("  per-image  (ld, nt st)", time_us([&] { hipLaunchKernelGGL((per_image<0, 1>), dim3(grid), dim3(256), 0, 0, in, out, n_img, nq_in, nq_out); }, 10), bytes);
        rep("  flat 625:300 (ld, st)", time_us([&] { hipLaunchKernelGGL((copy_flat<0, 0>), dim3(grid), dim3(256), 0, 0, in, out, NI, NO); }, 10), bytes);
        rep("  flat 625:300 (nt ld, nt st)", time_us([&] { hipLaunchKernelGGL((copy_flat<1, 1>), dim3(grid), dim3(256), 0, 0, in, out, NI, NO); }, 10), bytes);
        rep("  read only (ld)", time_us([&] { hipLaunchKernelGGL((read_only<0>), dim3(grid), dim3(256), 0, 0, in, out, NI); }, 10), NI * 16.0);
        rep("  read only (nt ld)", time_us([&] { hipLaunchKernelGGL((read_only<1>), dim3(grid), dim3(256), 0, 0, in, out, NI); }, 10), NI * 16.0);
    }
    // plain 1:1 copy of the same total size for reference (the guide's 6.29 TB/s figure)
    rep("1:1 copy (ld, st), grid 4096", time_us([&] { hipLaunchKernelGGL((copy_flat<0, 0>), dim3(4096), dim3(256), 0, 0, in, out, NO, NO); }, 10), NO * 32.0);
    return 0;
}
