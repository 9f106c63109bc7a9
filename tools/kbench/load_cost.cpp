// Developer probe: what does ONE CU pay per wave-level global load instruction, as a function of its shape?  The GEMM kernels of the
// step are bound by "per-CU operand ingest" (DESIGN 3); this separates the candidates -- bytes, cache lines, row segments or plain
// instruction count -- by timing a fixed number of independent loads per wave for several lane -> address patterns, operands L2 /
// Infinity-Cache resident (a 64 KB window per workgroup), 8 or 16 waves per CU, one workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/kbench/load_cost.cpp -o tools/kbench/bin/load_cost
// pattern = (bytes per lane, lanes per row segment): a wave instruction touches 64 / lanes_per_row rows, each a contiguous segment of
// lanes_per_row * bytes bytes; rows are `pitch` bytes apart (2 KB: different cache lines, as the rows of an activation matrix).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int BYTES> struct Vec;
template <> struct Vec<4> { typedef float T; };
template <> struct Vec<8> { typedef float2 T; };
template <> struct Vec<16> { typedef float4 T; };
__device__ __forceinline__ float sum(float v) { return v; }
__device__ __forceinline__ float sum(float2 v) { return v.x + v.y; }
__device__ __forceinline__ float sum(float4 v) { return v.x + v.y + v.z + v.w; }

// every wave issues `iters` batches of U independent loads; batch b of wave w reads rows [ (b*waves + w) * rows_per_instr * U ... )
#ifndef AIR_WINDOW_KB
#define AIR_WINDOW_KB 64
#endif
constexpr int PITCH = 2048, WINDOW = AIR_WINDOW_KB << 10, ROWS_W = WINDOW / PITCH;     // powers of two: the address arithmetic is shifts and masks
template <int BYTES, int LPR, int U>
__global__ void load_kernel(const char *__restrict__ base, float *out, int iters, int, size_t) {
    typedef typename Vec<BYTES>::T T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    constexpr int RPI = 64 / LPR;                              // rows per instruction
    const char *p = base + (size_t)blockIdx.x * WINDOW + (lane / LPR) * PITCH + (lane % LPR) * BYTES;
    float acc = 0.f;
    unsigned n = (unsigned)wave * U * RPI;
    const unsigned step = (unsigned)waves * U * RPI;
    for (int b = 0; b < iters; ++b, n += step) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // rows of the window in turn; every wrap moves one segment to the right, so a line is not touched again before the
            // whole window (twice the 32 KB L1) has gone by: L1 misses, L2 hits
            const unsigned m = n + u * RPI, row0 = m & (ROWS_W - 1), wrap = m / ROWS_W;
            v[u] = *reinterpret_cast<const T *>(p + row0 * PITCH + ((wrap * (LPR * BYTES)) & (PITCH / 2 - 1)));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += sum(v[u]);
    }
    if (acc == 12345.678f) out[threadIdx.x] = acc;             // never true: keeps the loads
}

template <int BYTES, int LPR, int U>
static int run(const char *name, const char *buf, float *out, int waves, int iters, int pitch, size_t wg_window) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto launch = [&] { hipLaunchKernelGGL((load_kernel<BYTES, LPR, U>), dim3(256), dim3(64 * waves), 0, 0, buf, out, iters, pitch, wg_window); };
    for (int i = 0; i < 5; ++i) launch();
    std::vector<double> r;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); r.push_back(ms * 1e3 / 20);
    }
    std::sort(r.begin(), r.end());
    const double us = r[2] - 3.0;                              // minus an empty launch (back-to-back, same stream)
    const double instrs = (double)waves * iters * U;           // per CU
    const double bytes = instrs * 64 * BYTES;
    const int rows = 64 / LPR, seg = LPR * BYTES;
    printf("%-34s %2d waves  %4d B/lane  %2d rows x %4d B  U=%d  %8.2f us  %6.2f ns/instr/CU  %6.1f GB/s/CU  %5.2f ns/row-segment\n", name, waves,
           BYTES, rows, seg, U, r[2], us * 1e3 / instrs, bytes / (us * 1e3), us * 1e3 / (instrs * rows));
    return 0;
}

int main() {
    const size_t wg_window = (size_t)WINDOW, total = 256 * wg_window;    // per workgroup; -DAIR_WINDOW_KB=64 (twice the L1; an XCD's 32 windows fit its 4 MB L2), 128 (4x the L1), 1024 (beyond L2: Infinity Cache)
    char *buf; float *out;
    CK(hipMalloc(&buf, total + (1 << 20))); CK(hipMemset(buf, 0, total + (1 << 20))); CK(hipMalloc(&out, 4096));
    const int pitch = 2048;
    for (int waves : {8, 16}) {
        const int iters = 2048 / waves;                            // the same number of instructions per CU for both
        run<16, 64, 8>("dwordx4, 1 row x 1 KB (stream)", buf, out, waves, iters / 8, pitch, wg_window);
        run<16, 16, 8>("dwordx4, 4 rows x 256 B (fp32 TN)", buf, out, waves, iters / 8, pitch, wg_window);
        run<8, 16, 8>("dwordx2, 4 rows x 128 B (bf16 TN)", buf, out, waves, iters / 8, pitch, wg_window);
        run<16, 4, 8>("dwordx4, 16 rows x 64 B (k-contig)", buf, out, waves, iters / 8, pitch, wg_window);
        run<4, 16, 8>("dword, 4 rows x 64 B (k-strided)", buf, out, waves, iters / 8, pitch, wg_window);
        run<4, 64, 8>("dword, 1 row x 256 B", buf, out, waves, iters / 8, pitch, wg_window);
        run<8, 8, 8>("dwordx2, 8 rows x 64 B", buf, out, waves, iters / 8, pitch, wg_window);
        run<16, 8, 8>("dwordx4, 8 rows x 128 B", buf, out, waves, iters / 8, pitch, wg_window);
    }
    return 0;
}
