// Shared device/host helpers for the gfx950 kernels of libair_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include "../../include/air_hip.h"

#define AIR_WAVE 64

#define AIR_REQUIRE(cond, code) do { if (!(cond)) return (code); } while (0)
#define AIR_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)

static inline hipStream_t air_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
static inline bool air_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
__device__ __forceinline__ bool air_aligned16_dev(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int air_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
// rider workgroups of a launch that carries an optimiser slice of nq float4 (optimizer_device.h): AIR_RIDER_QPT float4 per rider
// thread, default 2.  Measured (profiles/r05_bptt_entry_fold_ab.txt): more, smaller rider workgroups are SLOWER -- 1 / 0.5 / 0.25
// float4 per thread cost +1.0 ... +1.4 us per step at configs[1] against 2 -- the riders then compete with the tile workgroups they
// ride beside.
static inline size_t air_rider_blocks(size_t nq, int nth, size_t cap) {
    static const double qpt = getenv("AIR_RIDER_QPT") ? atof(getenv("AIR_RIDER_QPT")) : 2.0;
    const double per_wg = (qpt > 0.05 ? qpt : 2.0) * nth;
    size_t extra = (size_t)(((double)nq + per_wg - 1.0) / per_wg);
    return extra > cap ? cap : extra;
}
// Grid of a grid-stride ("persistent") launch whose work items outnumber the workgroups the chip holds at once: exactly the resident
// workgroups (CUs x what the kernel's registers and LDS allow per CU).  A cap that is not a multiple of that number runs in two
// unequal phases -- the workgroups past the resident set only start when the first ones have finished ALL their items: canvas forward
// (6 per CU: 1536 resident) 132 us with 1536 or 3072, 167 us with 2048, 176 with 1792; stored-canvas backward (5 per CU) 171 us with 1280,
// 200 with 1536, 177 with 2048 at 8192 images (profiles/r04_canvas_grid_sweep.txt).
template <typename K>
static inline int air_resident_grid(K kernel, int threads, size_t lds, int fallback) {
    // memoised per (kernel, threads, lds, device): the occupancy query is three host-side runtime calls, paid in front of EVERY
    // eager launch otherwise.  A handful of distinct keys per process.
    struct Entry { const void *k; int threads; size_t lds; int dev; int grid; };
    static Entry cache[32];
    static int n_cached = 0;
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fallback;
    const void *key = reinterpret_cast<const void *>(kernel);
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < n_cached; ++i)
        if (cache[i].k == key && cache[i].threads == threads && cache[i].lds == lds && cache[i].dev == dev) return cache[i].grid;
    int per_cu = 0, cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds) != hipSuccess || per_cu < 1 || cus < 1)
        return fallback;
    if (n_cached < 32) cache[n_cached++] = Entry{key, threads, lds, dev, per_cu * cus};
    return per_cu * cus;
}

// A zero the compiler cannot see through.  Adding it to a wave-uniform index moves that load from the scalar path (s_load,
// lgkmcnt) to the vector path (global_load, vmcnt).  Scalar loads return out of order, so the first use of ANY scalar-loaded
// value -- the kernel-argument pointers every address needs -- waits lgkmcnt(0), i.e. also for a slow uniform load from
// global memory issued next to them: seen in the ISA of the canvas backward as one full memory round trip in front of every
// vector load of the prologue.  On the vector path the same load is just one more request in flight.
__device__ __forceinline__ int opaque_zero() {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}

// ---- wave / block reductions (64-wide wavefronts) -----------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;  // valid in lane 0
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;  // valid in every lane
}

// Eight per-lane partial sums -> the eight totals over the 64 lanes of a wave in 10 cross-lane moves (a plain butterfly per
// value takes 48): every exchange step halves the number of live values.  Afterwards lane L holds the total of output
// o(L) = 4*bit5(L) + 2*bit4(L) + bit3(L) (the eight lanes that share bits 3..5 hold the same number).  Fixed order.
template <typename T>
__device__ __forceinline__ T wave_reduce8(const T (&v)[8]) {
    const int lane = threadIdx.x & 63;
    const bool h5 = (lane & 32) != 0, h4 = (lane & 16) != 0, h3 = (lane & 8) != 0;
    T a[4], b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = (h5 ? v[i + 4] : v[i]) + __shfl_xor(h5 ? v[i] : v[i + 4], 32, 64);
#pragma unroll
    for (int i = 0; i < 2; ++i) b[i] = (h4 ? a[i + 2] : a[i]) + __shfl_xor(h4 ? a[i] : a[i + 2], 16, 64);
    T c = (h3 ? b[1] : b[0]) + __shfl_xor(h3 ? b[0] : b[1], 8, 64);
    c += __shfl_xor(c, 4, 64);
    c += __shfl_xor(c, 2, 64);
    c += __shfl_xor(c, 1, 64);
    return c;
}
// which of the eight outputs of wave_reduce8 this lane holds
__device__ __forceinline__ int wave_reduce8_slot() {
    const int lane = threadIdx.x & 63;
    return ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
}

// Sum NV values per thread across a block of up to 1024 threads; result valid in thread 0.
// `scratch` must hold NV * (blockDim.x/64) floats; caller syncs before reusing scratch.
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float *scratch) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) scratch[wid * NV + k] = v[k];
    }
    __syncthreads();
    if (wid == 0) {
        // second stage by the first wave (fixed tree order => deterministic); a serial loop in thread 0 costs
        // nw dependent LDS reads per value, which is microseconds with 16 waves
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            float x = lane < nw ? scratch[lane * NV + k] : 0.f;
            v[k] = wave_sum(x);
        }
    }
}

// ---- scalar math with the exact op order of the oracle (no fp contraction) -----------------------------------
__device__ __forceinline__ float lin_m11(int k, int n, double step) {
#pragma clang fp contract(off)
    // np.linspace(-1, 1, n)[k] evaluated in fp64 (arange*step + start, last element = stop) then rounded to fp32
    if (n <= 1) return -1.0f;
    if (k == n - 1) return 1.0f;
    const double prod = (double)k * step;
    return (float)(prod + -1.0);
}
// coord = ((s*X + t) + 1) * half_extent, each op rounded separately (matches numpy / torch-CPU eager)
__device__ __forceinline__ float grid_coord(float s, float X, float t, float half_extent) {
#pragma clang fp contract(off)
    const float a = s * X;
    const float b = a + t;
    const float c = b + 1.0f;
    return c * half_extent;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float softplus_acc(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float elu_acc(float x) { return x > 0.0f ? x : expm1f(x); }
