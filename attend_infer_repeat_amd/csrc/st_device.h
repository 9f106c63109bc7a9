// Device helpers shared by the spatial-transformer kernels (st_kernels.hip: glimpse read, attend fusion) and the canvas kernels
// (canvas_kernels.hip: inverse write, its backward): bilinear taps in the oracle's operation order, per-axis table entries,
// the zero-bordered LDS glimpse copy, exact source ranges of the transposed bilinear map.
// Every function whose result must be bit-identical to the oracle rounds each operation separately: it carries its own
// `#pragma clang fp contract(off)`, so it keeps that property when it is inlined into a translation unit that contracts.
#pragma once
#include <limits.h>
#include <math.h>
#include "air_common.h"

#define ST_THREADS 256
#define ST_INVALID INT_MIN

// Developer tracing (tools/kbench/st_trace.cpp builds this file with -DAIR_TRACE): thread 0 of every workgroup stamps the
// chip-wide 100 MHz counter at the marked phase boundaries.  Compiles to nothing in the product build.
#ifdef AIR_TRACE
#ifndef AIR_TRACE_BLOCKS
#define AIR_TRACE_BLOCKS 512
#endif
#define AIR_TRACE_PHASES 12
__device__ unsigned long long air_trace[AIR_TRACE_BLOCKS * AIR_TRACE_PHASES];
// stamps go to LDS (a global store in front of a barrier would add its own latency to the phase being measured) and are
// flushed once when the workgroup is done
#define AIR_TR_INIT() __shared__ unsigned long long air_tr_lds[AIR_TRACE_PHASES]; do { if (threadIdx.x < AIR_TRACE_PHASES) air_tr_lds[threadIdx.x] = 0; __syncthreads(); } while (0)
#define AIR_TRT(t_, i) do { if (threadIdx.x == (t_)) air_tr_lds[(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define AIR_TR(i) AIR_TRT(0, i)
#define AIR_TR_FLUSH() do { __syncthreads(); if (threadIdx.x < AIR_TRACE_PHASES && blockIdx.x < AIR_TRACE_BLOCKS) air_trace[blockIdx.x * AIR_TRACE_PHASES + threadIdx.x] = air_tr_lds[threadIdx.x]; } while (0)
#else
#define AIR_TR_INIT() do { } while (0)
#define AIR_TRT(t_, i) do { } while (0)
#define AIR_TR(i) do { } while (0)
#define AIR_TR_FLUSH() do { } while (0)
#endif

struct Taps { float ff, fc, cf, cc; };

// bilinear taps around (fy, fx) of an LDS-resident Hs x Ws source; out-of-range taps are zero
__device__ __forceinline__ Taps load_taps(const float *s, int Hs, int Ws, int fy, int fx) {
    const bool x0 = fx >= 0, x1 = fx + 1 <= Ws - 1, y0 = fy >= 0, y1 = fy + 1 <= Hs - 1;
    Taps t;
    const int base = fy * Ws + fx;
    t.ff = (x0 && y0) ? s[base] : 0.f;
    t.fc = (x1 && y0) ? s[base + 1] : 0.f;
    t.cf = (x0 && y1) ? s[base + Ws] : 0.f;
    t.cc = (x1 && y1) ? s[base + Ws + 1] : 0.f;
    return t;
}
// dx*dy*ff + (1-dx)*(1-dy)*cc + dx*(1-dy)*cf + (1-dx)*dy*fc, left-to-right, no contraction (== oracle)
__device__ __forceinline__ float bilerp(const Taps &t, float dx, float dy) {
#pragma clang fp contract(off)
    // plain operators: the file-level `fp contract(off)` keeps every op separately rounded (HIP's __fmul_rn etc. are
    // header functions whose instructions carry the default contract flag and WOULD be fused after inlining)
    const float mx = 1.f - dx, my = 1.f - dy;
    float r = (dx * dy) * t.ff;
    r = r + (mx * my) * t.cc;
    r = r + (dx * my) * t.cf;
    r = r + (mx * dy) * t.fc;
    return r;
}
// one axis entry: coordinate -> (floor index or ST_INVALID, d = (floor+1) - coord)
__device__ __forceinline__ void axis_entry(float coord, int extent, int *f_out, float *d_out) {
#pragma clang fp contract(off)
    const bool valid = (coord > -1.0f) && (coord < (float)extent);   // NaN -> invalid
    const float fl = floorf(coord);
    *f_out = valid ? (int)fl : ST_INVALID;
    *d_out = (fl + 1.0f) - coord;
}

// one packed axis entry
__device__ __forceinline__ float2 axis_entry2(float coord, int extent) {
    int f; float d;
    axis_entry(coord, extent, &f, &d);
    return make_float2(__int_as_float(f), d);
}
// taps with unconditional (clamped) LDS reads and a select: no divergent branches around the four loads
__device__ __forceinline__ Taps load_taps_sel(const float *s, int Hs, int Ws, int fy, int fx) {
    const bool x0 = fx >= 0, x1 = fx + 1 <= Ws - 1, y0 = fy >= 0, y1 = fy + 1 <= Hs - 1;
    const int cx0 = x0 ? fx : 0, cx1 = x1 ? fx + 1 : Ws - 1, cy0 = y0 ? fy : 0, cy1 = y1 ? fy + 1 : Hs - 1;
    Taps t;
    const float a = s[cy0 * Ws + cx0], b = s[cy0 * Ws + cx1], c = s[cy1 * Ws + cx0], d = s[cy1 * Ws + cx1];
    t.ff = (x0 && y0) ? a : 0.f;
    t.fc = (x1 && y0) ? b : 0.f;
    t.cf = (x0 && y1) ? c : 0.f;
    t.cc = (x1 && y1) ? d : 0.f;
    return t;
}

// The canvas kernels keep a glimpse in LDS with a ONE-ELEMENT ZERO BORDER, (h + 2) x (w + 2): the four taps of any valid floor pair
// (fy in [-1, h-1], fx in [-1, w-1]) are then four in-bounds reads -- an out-of-range tap lands on the border and reads the +0.0f
// load_taps_sel selects -- with one address computation instead of four clamps, four index products and four selects
// (~50 -> ~20 instructions per bilinear read; measured -3 ... -8 % on the canvas launches, profiles/r03_probe_canvas_scaling.txt).
__device__ __forceinline__ int pad_count(int h, int w) { return ((h + 2) * (w + 2) + 3) & ~3; }
static inline size_t pad_count_host(int h, int w) { return (size_t)(((h + 2) * (w + 2) + 3) & ~3); }
__device__ __forceinline__ Taps load_taps_pad(const float *s, int pitch, int fy, int fx) {
    const float *q = s + (fy + 1) * pitch + (fx + 1);
    Taps t;
    t.ff = q[0]; t.fc = q[1]; t.cf = q[pitch]; t.cc = q[pitch + 1];
    return t;
}
// n / d for n >= 0, d > 0 with inv_d = 1.0f / d, exact while the quotient stays below 2^21 (row indices here): the three roundings
// (n -> float, 1/d, the product) move the float quotient by less than one, so the truncated value is off by at most one and the
// remainder test corrects it -- a third of the instructions of the compiler's expansion of a 32-bit division
__device__ __forceinline__ int div_small(int n, int d, float inv_d) {
    int q = (int)((float)n * inv_d);
    const int r = n - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}
// element e (row-major h x w) of a glimpse -> its place in the bordered LDS copy
__device__ __forceinline__ int pad_index(int e, int w, float inv_w) {
    const int row = div_small(e, w, inv_w);
    return (row + 1) * (w + 2) + (e - row * w) + 1;
}
// the 2 (w + 2) + 2 h border elements of one bordered glimpse, k = 0 .. pad_border(h, w) - 1 -> index
__device__ __forceinline__ int pad_border(int h, int w) { return 2 * (w + 2) + 2 * h; }
__device__ __forceinline__ int pad_border_index(int k, int h, int w) {
    const int pitch = w + 2;
    if (k < pitch) return k;
    if (k < 2 * pitch) return (h + 1) * pitch + (k - pitch);
    const int k2 = k - 2 * pitch;
    return (1 + (k2 >> 1)) * pitch + ((k2 & 1) ? w + 1 : 0);
}

// LDS-only workgroup barrier: waits for this wave's LDS traffic, NOT for its outstanding global loads (a plain
// __syncthreads() drains vmcnt too, which would serialise the register prefetch of the next image behind the barrier).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Canvas indices J (of n) whose source coordinate x(J) = cs*((a*X_J + b) + 1), X_J = -1 + 2J/(n-1), can land in [x0, x1].
// The map is affine in J, so the set is an interval; it is obtained from the inverse map with a margin of one index on each
// side (callers re-check every candidate exactly, so a superset is all that is needed) and degenerates to the full range for
// non-finite or zero scales.
__device__ __forceinline__ void src_range(float inv_a, float b, float inv_cs, float x0, float x1, int n, int *lo, int *hi) {
    const float half = 0.5f * (float)(n - 1);
    const float J0 = (((x0 * inv_cs - 1.0f) - b) * inv_a + 1.0f) * half;
    const float J1 = (((x1 * inv_cs - 1.0f) - b) * inv_a + 1.0f) * half;
    if (!(fabsf(J0) < 1e8f && fabsf(J1) < 1e8f)) { *lo = 0; *hi = n - 1; return; }     // NaN / inf / degenerate
    const float l = fminf(J0, J1), u = fmaxf(J0, J1);
    const int a0 = (int)floorf(l) - 1, a1 = (int)ceilf(u) + 1;
    *lo = a0 < 0 ? 0 : (a0 > n - 1 ? n - 1 : a0);         // both ends inside the table (a superset of the true set is fine)
    *hi = a1 > n - 1 ? n - 1 : (a1 < 0 ? 0 : a1);
}
// exact [lo, hi] of canvas indices whose taps touch source index j (floor == j-1 or j); empty => lo > hi.  `tab(J)` returns the axis
// entry of canvas index J: from the LDS table (TabAcc) or evaluated on the spot from the transform (FlyAcc: the same calls as the table
// build, so the same bits -- used where the ranges are wanted before the table is visible).
// The candidate interval of src_range is at most a handful of indices: up to eight are tested with independent reads (one round
// trip); longer intervals (degenerate scales) fall back to a scan from both ends.
struct TabAcc {
    const float2 *tab;
    __device__ __forceinline__ float2 operator()(int J) const { return tab[J]; }
};
struct FlyAcc {
    float a, b, cs;          // coord(J) = ((a * X_J + b) + 1) * cs, X_J = linspace(-1, 1, n)[J]
    int ext, n;
    double step;
    __device__ __forceinline__ float2 operator()(int J) const { return axis_entry2(grid_coord(a, lin_m11(J, n, step), b, cs), ext); }
};
template <typename Acc>
__device__ __forceinline__ int2 touch_range_t(const Acc &tab, float b, float inv_a, float inv_cs, int j, int n) {
    int lo, hi;
    src_range(inv_a, b, inv_cs, (float)(j - 1), (float)(j + 1), n, &lo, &hi);
    if (hi - lo < 8) {
        unsigned mask = 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int J = lo + u;
            const int f = __float_as_int(tab(J <= hi ? J : lo).x);
            if (J <= hi && f != ST_INVALID && (f == j || f + 1 == j)) mask |= 1u << u;
        }
        if (!mask) return make_int2(1, 0);
        return make_int2(lo + __ffs((int)mask) - 1, lo + 31 - __clz((int)mask));
    }
    while (lo <= hi) { const int f = __float_as_int(tab(lo).x); if (f != ST_INVALID && (f == j || f + 1 == j)) break; ++lo; }
    while (hi >= lo) { const int f = __float_as_int(tab(hi).x); if (f != ST_INVALID && (f == j || f + 1 == j)) break; --hi; }
    return make_int2(lo, hi);
}
__device__ __forceinline__ int2 touch_range(const float2 *tab, float b, float inv_a, float inv_cs, int j, int n) {
    return touch_range_t(TabAcc{tab}, b, inv_a, inv_cs, j, n);
}
// [first, last] index of an axis table with a valid entry (the valid set of a monotone map is an interval); every lane of the
// wave gets the result; empty => first > last
__device__ __forceinline__ int2 valid_span(const float2 *tab, int n) {
    const int lane = threadIdx.x & 63;
    int first = n, last = -1;
    for (int base = 0; base < n; base += 64) {
        const int k = base + lane;
        const bool v = k < n && __float_as_int(tab[k < n ? k : n - 1].x) != ST_INVALID;
        const unsigned long long m = __ballot(v);
        if (m) {
            const int lo = base + (int)__ffsll((long long)m) - 1, hi = base + 63 - (int)__clzll((long long)m);
            first = lo < first ? lo : first;
            last = hi > last ? hi : last;
        }
    }
    return make_int2(first, last);
}
