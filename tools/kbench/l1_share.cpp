// Developer probe: when two waves of one workgroup read the SAME lines a few hundred cycles apart (a 128x128 GEMM tile computed by
// 2x2 waves that each fetch their own 64-row / 64-column operand fragments register-direct), does the CU's vector L1 serve the
// second request, i.e. does the workgroup pull the line through the L2 -> L1 path only once?  256 workgroups of 512 threads stream
// 8-byte-per-lane loads (the access shape of the interleaved bf16 operand loaders) from a 24 MB region (beyond the 4 MB L2 of an
// XCD, inside the Infinity Cache); "dup = d": groups of d waves issue identical addresses.  Reported: issued bytes per second per CU
// and unique bytes per second per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/kbench/l1_share.cpp -o tools/kbench/bin/l1_share
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int DUP>
__global__ __launch_bounds__(512) void stream_kernel(const uint2 *__restrict__ src, size_t n_lines, int rounds, unsigned *sink, int share) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int group = wave / DUP;                       // waves of one group read the same addresses
    const int n_groups = 8 / DUP;
    // a "row" = 64 lanes x 8 B = 512 B; a chunk = 16 rows (one operand's share of a 32-deep chunk); rows 2 KB apart like a k-strided operand
    unsigned acc = 0;
    // share = s > 1: the s workgroups with the same (blockIdx / 8) / ... that sit on one XCD (ids congruent mod 8) read the SAME stream:
    // their lines cross the Infinity Cache -> L2 path once per XCD and are L2 hits for the other s - 1 workgroups
    const int bid = share > 1 ? (int)(blockIdx.x & 7) + 8 * (int)((blockIdx.x >> 3) / share) : (int)blockIdx.x;
    size_t base = ((size_t)bid * 977 + group * 131) % (n_lines - 4096);
    for (int r = 0; r < rounds; ++r) {
        uint2 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = src[(base + (size_t)j * 4) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += v[j].x ^ v[j].y;
        base = (base + 64 * n_groups + 17) % (n_lines - 4096);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const size_t bytes = 24u << 20, n_lines = bytes / 512;
    uint2 *src; unsigned *sink;
    CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 1, bytes)); CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int rounds = 48;
    printf("%5s %6s %12s %14s %14s\n", "dup", "share", "us", "issued GB/s/CU", "unique GB/s/CU (per CU, L1 level)");
    for (int share : {1, 2, 4, 8, 32})
    for (int dup : {1, 2}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            if (dup == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(256), dim3(512), 0, 0, src, n_lines, rounds, sink, share);
            if (dup == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(256), dim3(512), 0, 0, src, n_lines, rounds, sink, share);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double issued = 8.0 * rounds * 16 * 512;          // bytes per workgroup
        printf("%5d %6d %12.2f %14.1f %14.1f\n", dup, share, best * 1e3, issued / (best * 1e-3) / 1e9, issued / dup / (best * 1e-3) / 1e9);
    }
    return 0;
}
