// Spatial-transformer kernels for gfx950: fused affine-grid + bilinear glimpse read and the "attend" fusions around it
// (the inverse canvas write lives in canvas_kernels.hip).
//
// Replaces snt.AffineGridWarper + snt.resampler (+ gradient) behind attend_infer_repeat/modules.py:94-109
// (called at cell.py:135 and cell.py:159-165).  Semantics: SURVEY.md Appendix A.4 / A.7; oracle: oracle/air_oracle.py
// st_read / st_write and oracle/st_loops.c.
//
// Design (HBM-bound op, ~6 flop/byte): one 256-thread workgroup per source image.  The source (50x50 image for the
// read, 20x20 glimpse for the write) is staged ONCE into LDS with coalesced 16-byte loads; the affine grid is never
// materialised -- because the warp has no shear, x depends only on the output column and y only on the output row,
// so per-axis tables (floor index, fractional weight) are built in LDS by h+w threads and every output pixel costs
// four LDS gathers.  When T glimpses are read from one image (batched unroll) the image is staged once for all T.
// Forward arithmetic uses explicitly-rounded ops in the oracle's order, so forward results are bit-identical to it.
// Forward arithmetic must round every op separately to be bit-identical to the oracle: HIP's __fmul_rn/__fadd_rn are
// plain operators, so contraction into FMA is disabled for this whole translation unit.
#pragma clang fp contract(off)
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include "air_common.h"
#include "nvil_device.h"

#include "st_device.h"

__device__ __forceinline__ void stage_to_lds(float *dst, const float *src, int count, bool vec4) {
    const int nt = blockDim.x;
    if (vec4) {
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
        float4 *d4 = reinterpret_cast<float4 *>(dst);
        for (int q = threadIdx.x; q < (count >> 2); q += nt) d4[q] = s4[q];
    } else {
        for (int p = threadIdx.x; p < count; p += nt) dst[p] = src[p];
    }
}

// LDS carve: [src: cnt_src padded to 4][aux: cnt_aux padded to 4][fx:W_][dx:W_][X:W_][fy:H_][dy:H_][Y:H_][scratch 32]
struct Carve {
    float *src, *aux, *dx, *X, *dy, *Y, *scratch;
    int *fx, *fy;
};
__device__ __forceinline__ Carve carve_lds(float *smem, int cnt_src, int cnt_aux, int nx, int ny) {
    Carve c;
    float *p = smem;
    c.src = p; p += (cnt_src + 3) & ~3;
    c.aux = p; p += (cnt_aux + 3) & ~3;
    c.fx = reinterpret_cast<int *>(p); p += nx;
    c.dx = p; p += nx;
    c.X = p; p += nx;
    c.fy = reinterpret_cast<int *>(p); p += ny;
    c.dy = p; p += ny;
    c.Y = p; p += ny;
    c.scratch = p;
    return c;
}
static inline size_t carve_bytes(int cnt_src, int cnt_aux, int nx, int ny) {
    return sizeof(float) * (size_t)(((cnt_src + 3) & ~3) + ((cnt_aux + 3) & ~3) + 3 * nx + 3 * ny + 160);
}

// ============================================================================================================
// read: glimpse[k] = bilinear(img[k % n_img]; x = (W-1)/2*(sx*X_j+tx+1), y = (H-1)/2*(sy*Y_i+ty+1))
// ============================================================================================================
__global__ __launch_bounds__(ST_THREADS) void st_read_fwd_kernel(
    const float *__restrict__ img, const float *__restrict__ where, float *__restrict__ out,
    int n, int n_img, int H, int W, int h, int w, double stepx, double stepy, int vec4) {
    extern __shared__ __align__(16) float smem[];
    const int HW = H * W, hw = h * w, tid = threadIdx.x;
    Carve c = carve_lds(smem, HW, 0, w, h);
    const float cxs = (float)((W - 1) / 2.0), cys = (float)((H - 1) / 2.0);
    for (int b = blockIdx.x; b < n_img; b += gridDim.x) {
        __syncthreads();
        stage_to_lds(c.src, img + (size_t)b * HW, HW, vec4 != 0);
        for (int k = b; k < n; k += n_img) {
            __syncthreads();
            const float sx = where[4 * (size_t)k + 0], tx = where[4 * (size_t)k + 1];
            const float sy = where[4 * (size_t)k + 2], ty = where[4 * (size_t)k + 3];
            for (int a = tid; a < w + h; a += ST_THREADS) {
                if (a < w) axis_entry(grid_coord(sx, lin_m11(a, w, stepx), tx, cxs), W, &c.fx[a], &c.dx[a]);
                else axis_entry(grid_coord(sy, lin_m11(a - w, h, stepy), ty, cys), H, &c.fy[a - w], &c.dy[a - w]);
            }
            __syncthreads();
            float *o = out + (size_t)k * hw;
            for (int p = tid; p < hw; p += ST_THREADS) {
                const int i = p / w, j = p - i * w;
                const int fx = c.fx[j], fy = c.fy[i];
                float v = 0.f;
                if (fx != ST_INVALID && fy != ST_INVALID) v = bilerp(load_taps(c.src, H, W, fy, fx), c.dx[j], c.dy[i]);
                o[p] = v;
            }
        }
    }
}

// Software-pipelined variant used when the image is 16-byte addressable: while a workgroup computes the glimpses of
// image b out of LDS, the 16-byte loads of its NEXT image (and the `where` row of its next glimpse) are already in
// flight in registers, so the HBM stream never stops between images (the un-pipelined kernel above alternates
// load / compute and tops out near half of the achievable bandwidth when every glimpse has its own image).
// (three NAMED float4 registers per thread: an indexed register array is demoted to scratch by hipcc here, which
//  serialises every load behind a scratch store; blockDim.x = 256 covers images up to 768 float4, 1024 up to 3072)
// Streaming policy of the out-of-cache regime.  tools/kbench/stream_ceiling.cpp on this box (970 MB, "10 KB in, 4.8 KB out
// per image"): plain loads + stores 5.0 TB/s, non-temporal loads AND stores 5.4-6.0 TB/s, read-only nt 7.0 TB/s; this kernel:
// 4.4-4.6 TB/s plain -> 5.0-5.6 TB/s non-temporal with a 16 k-workgroup grid.  In cache-resident launches (the train step:
// the glimpses are consumed by the next GEMM) the default policy is kept.
template <bool NT>
__device__ __forceinline__ float4 ld_stream4(const float4 *p) {
    if (NT) {
        typedef float nt_f4 __attribute__((ext_vector_type(4)));
        const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *p;
}
template <bool NT>
__device__ __forceinline__ void st_stream(float *p, float v) {
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}
// Software-pipelined read used when the image is 16-byte addressable: while a workgroup computes the glimpses of image b out of
// LDS, the 16-byte loads of its NEXT image are already in flight in registers, so the HBM stream never stops between images
// (three NAMED float4 registers per thread: an indexed array is demoted to scratch here), behind LDS-only barriers.
// Round 4 put it on an instruction diet.  At 65536 images the round-2/3 form was bound by VALU ISSUE, not by memory: ~90
// instructions per output pixel (a runtime integer division for (row, column), four bounds-checked taps with their selects, the
// un-fused bilinear form) x 1200 outputs per image = ~2000 wave-instructions per image, 0.2 ms for the chip at one wave-instruction per
// four cycles per SIMD -- exactly the measured launch.  Here the image sits in LDS with a one-element ZERO BORDER ((H+2) x (W+2): an
// out-of-range tap of a valid floor pair reads the +0.0f the bounds check would have selected -- same bits), so the four taps are two
// ds_read2 off one address; an axis entry is one 8-byte LDS read {floor, d}; (row, column) come from a float reciprocal (div_small).
// The oracle's arithmetic (grid_coord, axis_entry, bilerp) is untouched: results stay bit-identical.
template <bool NT>
__device__ __forceinline__ void st_stream4(float *p, float a, float b, float c, float d) {
    typedef float nt_f4 __attribute__((ext_vector_type(4)));
    const nt_f4 v = {a, b, c, d};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<nt_f4 *>(p)); else *reinterpret_cast<nt_f4 *>(p) = v;
}
// VEC (w % 4 == 0, 16-byte aligned output): the axis tables of ALL R glimpses of the staged image are built behind ONE barrier pair
// (2 barriers per image instead of 1 + 2R) and a thread forms FOUR consecutive outputs of a glimpse row -- one row entry, four
// column entries, one 16-byte (non-temporal) store.
template <bool NT, bool VEC>
__global__ __launch_bounds__(1024) void st_read_fwd_lean_kernel(
    const float *__restrict__ img, const float *__restrict__ where, float *__restrict__ out,
    int n, int n_img, int H, int W, int h, int w, double stepx, double stepy) {
    extern __shared__ __align__(16) float smem[];
    const int HW = H * W, hw = h * w, tid = threadIdx.x, nt = blockDim.x, nq = HW >> 2;
    const int R = n / n_img;                                          // glimpses per staged image
    const int pitch = W + 2, padn = ((H + 2) * pitch + 3) & ~3;
    float *s_img = smem;                                              // bordered image
    float2 *xe = reinterpret_cast<float2 *>(smem + padn), *ye = xe + (VEC ? R : 1) * w;      // [R][w], [R][h] when VEC
    float *sX = reinterpret_cast<float *>(ye + (VEC ? R : 1) * h), *sY = sX + w;
    const float cxs = (float)((W - 1) / 2.0), cys = (float)((H - 1) / 2.0);
    const float inv_W = 1.0f / (float)W, inv_w = 1.0f / (float)w;
    const float4 *where4 = reinterpret_cast<const float4 *>(where);
    const int q0 = tid < nq ? tid : nq - 1, q1 = tid + nt < nq ? tid + nt : nq - 1;
    const int q2 = tid + 2 * nt < nq ? tid + 2 * nt : nq - 1;
    // where element 4q of the row-major image lands in the bordered copy: e + 2 row(e) + pitch + 1; an element past the end of
    // its row sits two further (the two border elements between rows)
    const int r0 = div_small(4 * q0, W, inv_W), r1 = div_small(4 * q1, W, inv_W), r2 = div_small(4 * q2, W, inv_W);
    const int c0 = 4 * q0 - r0 * W, c1 = 4 * q1 - r1 * W, c2 = 4 * q2 - r2 * W;
    const int b0 = 4 * q0 + 2 * r0 + pitch + 1, b1 = 4 * q1 + 2 * r1 + pitch + 1, b2 = 4 * q2 + 2 * r2 + pitch + 1;
    float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0, p2 = p0;
    int b = blockIdx.x;
    if (b < n_img) {
        const float4 *s4 = reinterpret_cast<const float4 *>(img + (size_t)b * HW);
        p0 = ld_stream4<NT>(s4 + q0); p1 = ld_stream4<NT>(s4 + q1); p2 = ld_stream4<NT>(s4 + q2);
    }
    float4 wnext = (b < n_img && !VEC) ? where4[b] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = tid; a < w + h; a += nt) {                   // the linspace tables do not depend on the image
        if (a < w) sX[a] = lin_m11(a, w, stepx); else sY[a - w] = lin_m11(a - w, h, stepy);
    }
    for (int e = tid; e < 2 * pitch + 2 * H; e += nt) {       // the border, once: rows 0 and H+1, columns 0 and W+1
        int idx;
        if (e < pitch) idx = e;
        else if (e < 2 * pitch) idx = (H + 1) * pitch + (e - pitch);
        else { const int k2 = e - 2 * pitch; idx = (1 + (k2 >> 1)) * pitch + ((k2 & 1) ? W + 1 : 0); }
        s_img[idx] = 0.f;
    }
    const int wq = w >> 2, hwq = hw >> 2;
    for (; b < n_img; b += gridDim.x) {
        lds_barrier();                                        // readers of the previous image (and of its tables) are done
        if (tid < nq) { s_img[b0] = p0.x; s_img[b0 + 1 + (c0 + 1 >= W ? 2 : 0)] = p0.y; s_img[b0 + 2 + (c0 + 2 >= W ? 2 : 0)] = p0.z; s_img[b0 + 3 + (c0 + 3 >= W ? 2 : 0)] = p0.w; }
        if (tid + nt < nq) { s_img[b1] = p1.x; s_img[b1 + 1 + (c1 + 1 >= W ? 2 : 0)] = p1.y; s_img[b1 + 2 + (c1 + 2 >= W ? 2 : 0)] = p1.z; s_img[b1 + 3 + (c1 + 3 >= W ? 2 : 0)] = p1.w; }
        if (tid + 2 * nt < nq) { s_img[b2] = p2.x; s_img[b2 + 1 + (c2 + 1 >= W ? 2 : 0)] = p2.y; s_img[b2 + 2 + (c2 + 2 >= W ? 2 : 0)] = p2.z; s_img[b2 + 3 + (c2 + 3 >= W ? 2 : 0)] = p2.w; }
        const int nb = b + gridDim.x;
        if (nb < n_img) {                                     // next image: in flight during this image's compute
            const float4 *s4 = reinterpret_cast<const float4 *>(img + (size_t)nb * HW);
            p0 = ld_stream4<NT>(s4 + q0); p1 = ld_stream4<NT>(s4 + q1); p2 = ld_stream4<NT>(s4 + q2);
        }
        if (VEC) {
            for (int a = tid; a < R * (w + h); a += nt) {     // every glimpse's tables (its `where` row: an L2 / L1 hit, 16 B)
                const int r = a / (w + h), e = a - r * (w + h);
                const float4 wk = where4[b + (size_t)r * n_img];
                if (e < w) xe[r * w + e] = axis_entry2(grid_coord(wk.x, sX[e], wk.y, cxs), W);
                else ye[r * h + (e - w)] = axis_entry2(grid_coord(wk.z, sY[e - w], wk.w, cys), H);
            }
            lds_barrier();
            for (int g = tid; g < R * hwq; g += nt) {         // four consecutive outputs of one glimpse row per thread
                const int r = g / hwq, q = g - r * hwq;
                const int i = div_small(q, wq, 1.0f / (float)wq), j = 4 * (q - i * wq);
                const float2 ey = ye[r * h + i];
                const int fy = __float_as_int(ey.x);
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (fy != ST_INVALID) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float2 ex = xe[r * w + j + u];
                        const int fx = __float_as_int(ex.x);
                        if (fx != ST_INVALID) v[u] = bilerp(load_taps_pad(s_img, pitch, fy, fx), ex.y, ey.y);
                    }
                }
                st_stream4<NT>(out + ((size_t)b + (size_t)r * n_img) * hw + 4 * q, v[0], v[1], v[2], v[3]);
            }
            continue;
        }
        for (int k = b; k < n; k += n_img) {
            const float4 wk = wnext;
            const bool more = k + n_img < n;
            if (more || nb < n_img) wnext = where4[more ? k + n_img : nb];
            lds_barrier();
            for (int a = tid; a < w + h; a += nt) {
                if (a < w) xe[a] = axis_entry2(grid_coord(wk.x, sX[a], wk.y, cxs), W);
                else ye[a - w] = axis_entry2(grid_coord(wk.z, sY[a - w], wk.w, cys), H);
            }
            lds_barrier();
            float *o = out + (size_t)k * hw;
            for (int p = tid; p < hw; p += nt) {
                const int i = div_small(p, w, inv_w), j = p - i * w;
                const float2 ex = xe[j], ey = ye[i];
                const int fx = __float_as_int(ex.x), fy = __float_as_int(ey.x);
                float v = 0.f;
                if (fx != ST_INVALID && fy != ST_INVALID) v = bilerp(load_taps_pad(s_img, pitch, fy, fx), ex.y, ey.y);
                st_stream<NT>(o + p, v);
            }
        }
    }
}
static inline size_t read_lean_bytes(int H, int W, int h, int w, int R) {
    return sizeof(float) * (size_t)((((H + 2) * (W + 2) + 3) & ~3) + 2 * R * (w + h) + (w + h) + 32);
}

// dwhere[k,4] = sum_ij dglimpse * d out / d(x,y) * d(x,y)/d where ; optional dimg (n_img == n)
__global__ __launch_bounds__(1024) void st_read_bwd_kernel(
    const float *__restrict__ img, const float *__restrict__ where, const float *__restrict__ dout,
    float *__restrict__ dwhere, float *__restrict__ dimg,
    int n, int n_img, int H, int W, int h, int w, double stepx, double stepy, int vec4, int per_glimpse) {
    extern __shared__ __align__(16) float smem[];
    const int HW = H * W, hw = h * w, tid = threadIdx.x, nt = blockDim.x;
    Carve c = carve_lds(smem, HW, dimg ? HW : 0, w, h);
    const float cxs = (float)((W - 1) / 2.0), cys = (float)((H - 1) / 2.0);
    // per_glimpse: one workgroup per glimpse (few glimpses, idle chip: parallelism beats re-using the staged image)
    const int n_units = per_glimpse ? n : n_img;
    for (int b = blockIdx.x; b < n_units; b += gridDim.x) {
        __syncthreads();
        stage_to_lds(c.src, img + (size_t)(b % n_img) * HW, HW, vec4 != 0);
        if (dimg) for (int p = tid; p < HW; p += nt) c.aux[p] = 0.f;
        for (int k = b; k < (per_glimpse ? b + 1 : n); k += n_img) {
            __syncthreads();
            const float sx = where[4 * (size_t)k + 0], tx = where[4 * (size_t)k + 1];
            const float sy = where[4 * (size_t)k + 2], ty = where[4 * (size_t)k + 3];
            for (int a = tid; a < w + h; a += nt) {
                if (a < w) {
                    const float X = lin_m11(a, w, stepx);
                    c.X[a] = X;
                    axis_entry(grid_coord(sx, X, tx, cxs), W, &c.fx[a], &c.dx[a]);
                } else {
                    const float Y = lin_m11(a - w, h, stepy);
                    c.Y[a - w] = Y;
                    axis_entry(grid_coord(sy, Y, ty, cys), H, &c.fy[a - w], &c.dy[a - w]);
                }
            }
            __syncthreads();
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            const float *g = dout + (size_t)k * hw;
            for (int p = tid; p < hw; p += nt) {
                const int i = p / w, j = p - i * w;
                const int fx = c.fx[j], fy = c.fy[i];
                if (fx == ST_INVALID || fy == ST_INVALID) continue;
                const float dx = c.dx[j], dy = c.dy[i], go = g[p];
                const Taps t = load_taps(c.src, H, W, fy, fx);
                const float gx = dy * (t.fc - t.ff) + (1.f - dy) * (t.cc - t.cf);
                const float gy = dx * (t.cf - t.ff) + (1.f - dx) * (t.cc - t.fc);
                const float ax = go * gx * cxs, ay = go * gy * cys;
                acc[0] += ax * c.X[j]; acc[1] += ax;
                acc[2] += ay * c.Y[i]; acc[3] += ay;
                if (dimg) {
                    const bool x0 = fx >= 0, x1 = fx + 1 <= W - 1, y0 = fy >= 0, y1 = fy + 1 <= H - 1;
                    const int base = fy * W + fx;
                    if (x0 && y0) atomicAdd(&c.aux[base], dx * dy * go);
                    if (x1 && y0) atomicAdd(&c.aux[base + 1], (1.f - dx) * dy * go);
                    if (x0 && y1) atomicAdd(&c.aux[base + W], dx * (1.f - dy) * go);
                    if (x1 && y1) atomicAdd(&c.aux[base + W + 1], (1.f - dx) * (1.f - dy) * go);
                }
            }
            block_sum<4>(acc, c.scratch);
            if (tid == 0) {
                float *d = dwhere + 4 * (size_t)k;
                d[0] = acc[0]; d[1] = acc[1]; d[2] = acc[2]; d[3] = acc[3];
            }
        }
        if (dimg) {
            __syncthreads();
            for (int p = tid; p < HW; p += nt) dimg[(size_t)b * HW + p] = c.aux[p];
        }
    }
}

// ============================================================================================================
// host side
// ============================================================================================================
static inline double lin_step(int n) { return n > 1 ? 2.0 / (double)(n - 1) : 0.0; }
static inline int st_grid(int items, int cap = 256 * 8) {   // 256 CUs x up to 8 resident 256-thread workgroups; grid-stride beyond that
    return items < cap ? items : cap;
}
static inline int st_check_dims(int n, int H, int W, int h, int w) {
    if (n <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0) return AIR_E_SHAPE;
    return AIR_OK;
}
#define ST_MAX_LDS (160 * 1024)
// dynamic LDS above 64 KiB must be opted into per kernel
template <typename K>
static inline int st_allow_lds(K kernel, size_t lds) {
    if (lds <= 64 * 1024) return AIR_OK;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return e == hipSuccess ? AIR_OK : (int)e;
}

extern "C" int air_st_read_fwd(const float *img, const float *where, float *glimpse, int n, int n_img, int H, int W,
                               int h, int w, void *stream) {
    AIR_REQUIRE(img && where && glimpse, AIR_E_NULL);
    int st = st_check_dims(n, H, W, h, w);
    if (st) return st;
    AIR_REQUIRE(n_img > 0 && n % n_img == 0, AIR_E_SHAPE);
    const size_t lds = carve_bytes(H * W, 0, w, h);
    AIR_REQUIRE(lds <= ST_MAX_LDS, AIR_E_UNSUPPORTED);
    const int vec4 = ((H * W) % 4 == 0) && air_aligned16(img);
    const int nq = (H * W) / 4;
    // (W < 4: a 16-byte group of the row-major image can cross more than one row boundary, which the lean kernel's bordered staging
    //  -- "+2 once past the end of the row" -- does not handle: narrow images take the generic kernel)
    if (vec4 && W >= 4 && air_aligned16(where) && nq <= 3 * 1024) {
        const int threads = nq <= 3 * 256 ? 256 : 1024;
        // out of cache (more than ~1/4 of the 256 MiB Infinity Cache touched once): streaming loads / stores, many short workgroups
        const size_t touched = sizeof(float) * ((size_t)n_img * H * W + (size_t)n * h * w);
        const int R = n / n_img;
        static const int lean = getenv("AIR_ST_READ_LEAN") ? atoi(getenv("AIR_ST_READ_LEAN")) : 2;      // 1: one glimpse at a time (probes)
        // (one glimpse per image: nothing to share behind the barrier pair, and four outputs per thread would leave 100 threads busy)
        const bool vec = lean >= 2 && R >= 2 && (w % 4 == 0) && air_aligned16(glimpse) && read_lean_bytes(H, W, h, w, R) <= ST_MAX_LDS;
        const size_t lds_l = read_lean_bytes(H, W, h, w, vec ? R : 1);
        if (lean && lds_l <= ST_MAX_LDS) {
            const bool big = touched > ((size_t)64 << 20);
            int nthr = threads;
            {   // (measured, profiles/r04_st_read_lean.txt: 256 threads beat 320 / 384 / 512 out of cache although 320 would form the
                //  300 output groups of a 50x50 / 20x20 / T=3 image in one round: 186 against 198 / 212 / 210 us at 65536 images)
                static const int forced = getenv("AIR_ST_READ_THREADS") ? atoi(getenv("AIR_ST_READ_THREADS")) : 0;
                if (forced >= 64 && forced <= 1024 && forced % 64 == 0 && 3 * forced >= nq) nthr = forced;
            }
            // out of cache: at least four images per workgroup (the per-workgroup preamble -- linspace, border, index set-up -- and the
            // register prefetch of the next image only pay then: 25.1 against 26.8-27.2 us at 8192 images, 49.5-51 against 55-57 us at
            // 16384, profiles/r04_st_read_grid.txt), at most 16384 workgroups (177.6 against 186 us with 2048 at 65536 images)
            static const int grid_forced = getenv("AIR_ST_READ_GRID") ? atoi(getenv("AIR_ST_READ_GRID")) : 0;
            // The grid is a whole multiple of the workgroups resident at once (air_resident_grid: other caps run in unequal phases).
            const dim3 th(nthr);
#define AIR_READ_CASE(NT_, VEC_) do { { int st_ = st_allow_lds(st_read_fwd_lean_kernel<NT_, VEC_>, lds_l); if (st_) return st_; } \
            int cap_ = 256 * 8;                                                                                               \
            if (n_img > cap_) {                                                                                               \
                const int res_ = air_resident_grid(st_read_fwd_lean_kernel<NT_, VEC_>, nthr, lds_l, 256 * 8);                 \
                int want_ = big ? (VEC_ ? n_img / 4 : 16384) : res_;    /* (one glimpse per image: 46 against 50 us at 24576 images with 16384) */ \
                want_ = want_ > 16384 ? 16384 : want_;                                                                        \
                /* images above 3 x 256 groups (100x100: 1024-thread workgroups, two per CU): exactly the resident workgroups, each    \
                   walking its images with the next one in flight -- 0.74 / 0.75 / 0.70 of 8 TB/s at 8192 / 32768 / 65536 images of    \
                   100x100 / 28x28 / T=5 against 0.71 / 0.73 / 0.66 with the rule above; beyond 48 k images two images per workgroup   \
                   (0.70 against 0.70: a tie, kept for the short workgroups' tail) -- profiles/r06_read_c4_grid.txt */                 \
                if (big && nthr == 1024) want_ = n_img >= 49152 ? n_img / 2 : res_;                                              \
                if (grid_forced > 0) want_ = grid_forced;                                                                     \
                cap_ = want_ <= res_ ? res_ : (want_ / res_) * res_;                                                          \
            }                                                                                                                 \
            hipLaunchKernelGGL((st_read_fwd_lean_kernel<NT_, VEC_>), dim3(st_grid(n_img, cap_)), th, lds_l, air_stream(stream), img, where, glimpse, \
                               n, n_img, H, W, h, w, lin_step(w), lin_step(h)); } while (0)
            if (big) { if (vec) AIR_READ_CASE(true, true); else AIR_READ_CASE(true, false); }
            else { if (vec) AIR_READ_CASE(false, true); else AIR_READ_CASE(false, false); }
#undef AIR_READ_CASE
            AIR_LAUNCH_CHECK();
            return AIR_OK;
        }
    }
    { int st_ = st_allow_lds(st_read_fwd_kernel, lds); if (st_) return st_; }
    const int cap_f = n_img > 256 * 8 ? air_resident_grid(st_read_fwd_kernel, ST_THREADS, lds, 256 * 8) : 256 * 8;
    hipLaunchKernelGGL(st_read_fwd_kernel, dim3(st_grid(n_img, cap_f)), dim3(ST_THREADS), lds, air_stream(stream), img, where,
                       glimpse, n, n_img, H, W, h, w, lin_step(w), lin_step(h), vec4);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

extern "C" int air_st_read_bwd(const float *img, const float *where, const float *dglimpse, float *dwhere,
                               float *dimg, int n, int n_img, int H, int W, int h, int w, void *stream) {
    AIR_REQUIRE(img && where && dglimpse && dwhere, AIR_E_NULL);
    int st = st_check_dims(n, H, W, h, w);
    if (st) return st;
    AIR_REQUIRE(n_img > 0 && n % n_img == 0, AIR_E_SHAPE);
    AIR_REQUIRE(!dimg || n_img == n, AIR_E_UNSUPPORTED);
    const size_t lds = carve_bytes(H * W, dimg ? H * W : 0, w, h);
    AIR_REQUIRE(lds <= ST_MAX_LDS, AIR_E_UNSUPPORTED);
    const int vec4 = ((H * W) % 4 == 0) && air_aligned16(img);
    { int st_ = st_allow_lds(st_read_bwd_kernel, lds); if (st_) return st_; }
    const int per_glimpse = (!dimg && n <= 2048 && n_img < n) ? 1 : 0;
    const int units_b = per_glimpse ? n : n_img;
    const int cap_b = units_b > 256 * 8 ? air_resident_grid(st_read_bwd_kernel, ST_THREADS, lds, 256 * 8) : 256 * 8;
    hipLaunchKernelGGL(st_read_bwd_kernel, dim3(st_grid(units_b, cap_b)), dim3(ST_THREADS), lds,
                       air_stream(stream), img, where, dglimpse, dwhere, dimg, n, n_img, H, W, h, w, lin_step(w),
                       lin_step(h), vec4, per_glimpse);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// ============================================================================================================
// "attend": the last (tiny) layers of the transform / steps MLPs, where-sampling, presence + num-steps posterior and the
// glimpse read of all T steps in ONE launch (cell.py:129-151 + modules.py:104-109).  At batch 64 these were three
// dependent launches (a 192x8x256 | 192x1x64 GEMM pair, the heads kernel, the ST read) of ~4-5 us each with almost no work.
//   role A, one workgroup per image b: its image is prefetched into registers first (nothing depends on it), then
//           pre[t*B+b, 0:8] = h2 . W + bias for every t (one wave per row), where ~ N(loc, softplus(raw)) + KL row, and the
//           T glimpses are resampled out of the LDS-staged image with `where` taken straight from LDS;
//   role B, one workgroup per 64 batch columns: logit = s2 . w + b (64-deep dots), then the sequential presence chain and
//           the float64 num-steps posterior / KL / step weights / log q(n) (presence_numsteps_fwd_body).
// The two roles share nothing but the launch.
// ============================================================================================================
#include "engine_device.h"

struct AttendFwdArgs {
    const float *tr_h, *tr_w, *tr_b; int tr_k;       // [M, tr_k] . [tr_k, 8] + [8]   -> pre[M, 8]
    const float *st_h, *st_w, *st_b; int st_k;       // [M, st_k] . [st_k, 1] + [1]   -> logit[M]
    float *pre, *logit;
    const float *eps; float raw_offset, pl0, ps0, pl1, ps1, guard_eps;
    float *loc, *scale, *where, *kl_row;
    const float *u; float step_bias, explore_eps; const double *prior;
    float *prob, *pres, *q, *kl_ps, *logp, *step_w;
    const float *img; float *glimpse;
    int T, B, H, W, h, w, bf16;
    int img_major;      // role A: one workgroup per IMAGE runs its T glimpses (the image is staged once) instead of one per glimpse
    int lean;           // ... and (image-major only) resamples them as st_read_fwd_lean_kernel does: bordered image, axis tables, 4 outputs per thread
    double stepx, stepy;
};
// operand of a dense product: as is, or rounded to bf16 (EngineConfig.mfma_dtype = "bf16": same arithmetic as the MFMA path)
__device__ __forceinline__ float opnd(float v, int bf16) { return bf16 ? (float)(__bf16)v : v; }

// Role A runs one workgroup per GLIMPSE (t, b) -- T*B workgroups, so a batch of 64 already covers 192 of the 256 CUs (one
// workgroup per image left three quarters of the chip idle and serialised the T reads behind 2T barriers).  Each workgroup
// prefetches its image into registers, wave 0 forms the 8 outputs of the transform layer for row t*B+b (one memory round trip,
// wave_reduce8) and samples `where`, which reaches the other waves through LDS behind the ONE barrier of the kernel; the
// glimpse pixels then compute their own axis entries (no table phase) and gather four taps from the LDS-staged image.
// (5 waves per SIMD: with 4 -- 104 VGPRs -- a batch of 1024 fills every slot of the chip with role-A workgroups and the last
//  few dozen only start when the first retire: two rounds instead of one)
template <int MT, int NT, bool EXACT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT <= 256 ? 5 : 4))) void attend_fwd_kernel(AttendFwdArgs g) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int T = EXACT ? MT : g.T, B = g.B;                // EXACT: T is a compile-time constant, every t-loop unrolls flat
    const int n = g.img_major ? B : T * B;                  // role-A workgroups
    // role B owns the FIRST workgroups of the grid: it is the longer chain, and once role A alone fills the chip (batch 1024)
    // workgroups at the end of the grid only start when the first ones retire (13.6 -> 8 us for the launch)
    const int nb_ = (int)gridDim.x - n, bid = (int)blockIdx.x - nb_;
    AIR_TR_INIT();
    if (bid < 0) {
        // ---- role B: steps-predictor output layer + presence / num-steps for 64 batch columns ----------------------
        // Four adjacent lanes per batch column: each sums a quarter of the K range of row (t, c) for every t with 16-byte
        // loads (the four lanes of a column cover one contiguous row segment), two cross-lane adds finish the dot product in
        // every lane, and the T logits stay in registers through the presence chain and the float64 posterior.  (The r01
        // form -- one lane per column, scalar loads -- walked 64 different cache lines per load instruction and read the
        // logits back from memory: 5.9 us for this role against 3.5 us for the glimpse role, traced.)
        const int vb = (int)blockIdx.x, vgrid = nb_;
        AIR_TR(4);
        const int z0 = opaque_zero();                          // vector-path loads of wave-uniform operands (see opaque_zero)
        const float bias = g.st_b[z0];
        const int cc = tid >> 2, sl = tid & 3;
        const bool vec = (g.st_k & 15) == 0 && air_aligned16_dev(g.st_h) && air_aligned16_dev(g.st_w);
        const int kq = vec ? g.st_k >> 2 : (g.st_k + 3) >> 2, k0 = sl * kq, k1 = (k0 + kq < g.st_k) ? k0 + kq : g.st_k;
        for (int base = vb * 64; base < B; base += vgrid * 64) {
            if (tid >= 256) break;
            const int b = base + cc;
            const bool live = b < B;
            float uu[MT], lg[MT];
            double pri[MT + 1];
#pragma unroll
            for (int t = 0; t < MT; ++t) uu[t] = (t < T && live && g.u) ? g.u[(size_t)t * B + b] : 0.f;
#pragma unroll
            for (int q = 0; q <= MT; ++q) pri[q] = q <= T ? g.prior[q + z0] : 1.0;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                float acc = 0.f;
                if (t < T && live) {
                    const float *x = g.st_h + ((size_t)t * B + b) * g.st_k;
                    if (vec) {
                        const float4 *x4 = reinterpret_cast<const float4 *>(x + k0);
                        const float4 *w4 = reinterpret_cast<const float4 *>(g.st_w + k0);
#pragma unroll 4
                        for (int q = 0; q < (kq >> 2); ++q) {
                            const float4 xv = x4[q], wv = w4[q];
                            acc += opnd(xv.x, g.bf16) * opnd(wv.x, g.bf16);
                            acc += opnd(xv.y, g.bf16) * opnd(wv.y, g.bf16);
                            acc += opnd(xv.z, g.bf16) * opnd(wv.z, g.bf16);
                            acc += opnd(xv.w, g.bf16) * opnd(wv.w, g.bf16);
                        }
                    } else {
                        for (int k = k0; k < k1; ++k) acc += opnd(x[k], g.bf16) * opnd(g.st_w[k], g.bf16);
                    }
                }
                acc += __shfl_xor(acc, 1, 64);
                acc += __shfl_xor(acc, 2, 64);
                lg[t] = acc + bias;
            }
            AIR_TR(5);
            if (live && sl == 0) {
#pragma unroll
                for (int t = 0; t < MT; ++t) if (t < T) g.logit[(size_t)t * B + b] = lg[t];
                presence_numsteps_col<MT>(b, lg, uu, pri, g.step_bias, g.explore_eps, g.prob, g.pres, g.q, g.kl_ps, g.logp,
                                          g.step_w, T, B, g.u != nullptr);
            }
        }
        AIR_TR(6);
        AIR_TR_FLUSH();
        return;
    }
    AIR_TR(0);
    // ---- role A: glimpse (t, b), or -- image-major, the throughput regime -- all T glimpses of image b ------------------
    const int b = bid % B;
    const int t0 = g.img_major ? 0 : bid / B, t1 = g.img_major ? T : t0 + 1;
    const int H = g.H, W = g.W, h = g.h, w = g.w, HW = H * W, hw = h * w, nq = HW >> 2;
    Carve c = carve_lds(smem, HW, 0, w, h);
    // lean layout (g.lean, see below): bordered image | xe[T][w] | ye[T][h] | sX[w] | sY[h] | where rows
    const int pitch = W + 2, padn = ((H + 2) * pitch + 3) & ~3;
    float2 *lxe = reinterpret_cast<float2 *>(smem + padn), *lye = lxe + T * w;
    float *lsX = reinterpret_cast<float *>(lye + T * h), *lsY = lsX + w;
    float *s_where = g.lean ? lsY + h : c.scratch;             // [t1 - t0][4]
    const float cxs = (float)((W - 1) / 2.0), cys = (float)((H - 1) / 2.0);
    // (512 threads: five 16-byte groups per thread -- a 100x100 image with two workgroups per CU, see air_attend_fwd)
    constexpr int NP = NT == 512 ? 5 : 3;
    const int nwv = nt >> 6;
    const bool wr = wave < t1 - t0;
    const float4 *s4 = reinterpret_cast<const float4 *>(g.img + (size_t)b * HW);
    float4 pf[NP];                                             // in flight while the first waves form `where`
    // (requesting the row's operands BEFORE the image -- loads return in issue order -- changed nothing, traced: the `where` phase waits
    //  for its own cold operands and, at batch 1024, behind the chip-wide queue of every workgroup's image, not behind this wave's share)
#pragma unroll
    for (int u = 0; u < NP; ++u) pf[u] = s4[tid + u * NT < nq ? tid + u * NT : nq - 1];
    // one wave per `where` row: wave 0 for a single glimpse; image-major, the T rows of the image go to different waves
    // (serially on wave 0 they were the longest phase of the workgroup: 4.9 of 8.3 us at T = 3)
    if (wr) {
        const int o = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1), d = o & 3;
        const float bias_o = g.tr_b[o];
        for (int t = t0 + wave; t < t1; t += nwv) {
            const size_t m = (size_t)t * B + b;                // row t*B + b
            const float eps_d = g.eps[m * 4 + d];
            const float *x = g.tr_h + m * g.tr_k;
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int k = lane; k < g.tr_k; k += 64) {
                const float xv = opnd(x[k], g.bf16);
                float4 wa = *reinterpret_cast<const float4 *>(g.tr_w + (size_t)k * 8);
                float4 wb = *reinterpret_cast<const float4 *>(g.tr_w + (size_t)k * 8 + 4);
                if (g.bf16) {
                    wa.x = opnd(wa.x, 1); wa.y = opnd(wa.y, 1); wa.z = opnd(wa.z, 1); wa.w = opnd(wa.w, 1);
                    wb.x = opnd(wb.x, 1); wb.y = opnd(wb.y, 1); wb.z = opnd(wb.z, 1); wb.w = opnd(wb.w, 1);
                }
                acc[0] += xv * wa.x; acc[1] += xv * wa.y; acc[2] += xv * wa.z; acc[3] += xv * wa.w;
                acc[4] += xv * wb.x; acc[5] += xv * wb.y; acc[6] += xv * wb.z; acc[7] += xv * wb.w;
            }
            const float e_o = wave_reduce8(acc) + bias_o;      // lanes < 32: e_loc[d]; lanes >= 32: e_raw[d]
            const float e_partner = __shfl_xor(e_o, 32, 64);
            if (lane < 32) {
                const float e_loc = e_o, e_raw = e_partner;
                const float mu = (d & 1) ? tanhf(e_loc) : sigmoid_acc(e_loc);                    // modules.py:41-46
                const float sc = guard_scale(softplus_acc(e_raw + g.raw_offset), g.guard_eps);
                const float v = guard_where(mu + sc * eps_d, d, 1, g.guard_eps);                // cell.py:130-133
                float kl = (d & 1) ? normal_kl(mu, sc, g.pl1, g.ps1) : normal_kl(mu, sc, g.pl0, g.ps0);
                kl += __shfl_xor(kl, 8, 64);
                kl += __shfl_xor(kl, 16, 64);
                if ((lane & 7) == 0) {
                    g.pre[m * 8 + d] = e_loc;
                    g.pre[m * 8 + 4 + d] = e_raw;
                    const size_t oo = m * 4 + d;
                    g.loc[oo] = mu; g.scale[oo] = sc; g.where[oo] = v;
                    s_where[4 * (t - t0) + d] = v;
                    if (lane == 0) g.kl_row[m] = kl;
                }
            }
        }
        AIR_TR(1);
    }
    if (g.lean) {
        // Image-major in the throughput regime (round 6): the chip holds four or five of these workgroups per CU and the launch is bound by
        // VALU issue, as the stand-alone read was before its instruction diet -- ~90 instructions per output pixel in the loop below
        // (per-pixel axis entries, an integer division, four bounds-checked taps).  Same diet here: the image goes to LDS with a zero
        // border (an out-of-range tap reads the +0.0f the bounds check selects), the axis entries of the T glimpses are formed once
        // (T (w + h) instead of 2 T h w), a thread forms four consecutive outputs of a glimpse row and stores 16 bytes.  Arithmetic
        // untouched (grid_coord / axis_entry2 / bilerp): bit-identical outputs.  The waves that form no `where` row zero the border and
        // fill the linspace tables meanwhile; LDS-only barriers (nobody waits for the stores of the `where` phase).
        float *s_img = smem;
        const int n_aux = 2 * pitch + 2 * H + w + h;
        const int wf0 = (t1 - t0) < nwv ? (t1 - t0) : 0;      // first wave without a `where` row (all of them when there is none)
        for (int e = tid - wf0 * 64; e >= 0 && e < n_aux; e += NT - wf0 * 64) {
            if (e < pitch) s_img[e] = 0.f;
            else if (e < 2 * pitch) s_img[(H + 1) * pitch + (e - pitch)] = 0.f;
            else if (e < 2 * pitch + 2 * H) { const int k2 = e - 2 * pitch; s_img[(1 + (k2 >> 1)) * pitch + ((k2 & 1) ? W + 1 : 0)] = 0.f; }
            else { const int a = e - 2 * pitch - 2 * H; if (a < w) lsX[a] = lin_m11(a, w, g.stepx); else lsY[a - w] = lin_m11(a - w, h, g.stepy); }
        }
        const float inv_W = 1.0f / (float)W;
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int q = tid + u * NT;
            if (q < nq) {
                const int r = div_small(4 * q, W, inv_W), cc = 4 * q - r * W, bb = 4 * q + 2 * r + pitch + 1;
                s_img[bb] = pf[u].x; s_img[bb + 1 + (cc + 1 >= W ? 2 : 0)] = pf[u].y;
                s_img[bb + 2 + (cc + 2 >= W ? 2 : 0)] = pf[u].z; s_img[bb + 3 + (cc + 3 >= W ? 2 : 0)] = pf[u].w;
            }
        }
        lds_barrier();                                         // image, border, linspace tables, `where` rows
        AIR_TR(2);
        for (int a = tid; a < T * (w + h); a += NT) {
            const int r = a / (w + h), e = a - r * (w + h);
            const float *sw = s_where + 4 * r;
            if (e < w) lxe[r * w + e] = axis_entry2(grid_coord(sw[0], lsX[e], sw[1], cxs), W);
            else lye[r * h + (e - w)] = axis_entry2(grid_coord(sw[2], lsY[e - w], sw[3], cys), H);
        }
        lds_barrier();
        const int wq = w >> 2, hwq = hw >> 2;
        const float inv_wq = 1.0f / (float)wq;
        for (int gi = tid; gi < T * hwq; gi += NT) {
            const int r = gi / hwq, q = gi - r * hwq;
            const int i = div_small(q, wq, inv_wq), j = 4 * (q - i * wq);
            const float2 ey = lye[r * h + i];
            const int fy = __float_as_int(ey.x);
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (fy != ST_INVALID) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float2 ex = lxe[r * w + j + u];
                    const int fx = __float_as_int(ex.x);
                    if (fx != ST_INVALID) v[u] = bilerp(load_taps_pad(s_img, pitch, fy, fx), ex.y, ey.y);
                }
            }
            st_stream4<false>(g.glimpse + ((size_t)r * B + b) * hw + 4 * q, v[0], v[1], v[2], v[3]);
        }
        AIR_TR(3);
        AIR_TR_FLUSH();
        return;
    }
    float4 *d4 = reinterpret_cast<float4 *>(c.src);
#pragma unroll
    for (int u = 0; u < NP; ++u) if (tid + u * NT < nq) d4[tid + u * NT] = pf[u];
    __syncthreads();                                           // image + every `where` row of this workgroup visible
    AIR_TR(2);
    for (int t = t0; t < t1; ++t) {
        const float *sw = s_where + 4 * (t - t0);
        const float sx = sw[0], tx = sw[1], sy = sw[2], ty = sw[3];
        float *o = g.glimpse + ((size_t)t * B + b) * hw;
        for (int p = tid; p < hw; p += nt) {
            const int i = p / w, j = p - i * w;
            int fx, fy; float dx, dy;
            axis_entry(grid_coord(sx, lin_m11(j, w, g.stepx), tx, cxs), W, &fx, &dx);
            axis_entry(grid_coord(sy, lin_m11(i, h, g.stepy), ty, cys), H, &fy, &dy);
            float v = 0.f;
            if (fx != ST_INVALID && fy != ST_INVALID) v = bilerp(load_taps_sel(c.src, H, W, fy, fx), dx, dy);
            o[p] = v;
        }
    }
    AIR_TR(3);
    AIR_TR_FLUSH();
}

extern "C" int air_attend_fwd(const float *tr_h, const float *tr_w, const float *tr_b, int tr_k, const float *st_h,
                              const float *st_w, const float *st_b, int st_k, float *pre, float *logit,
                              const float *eps, float raw_offset, float p_loc_even, float p_scale_even, float p_loc_odd,
                              float p_scale_odd, float *loc, float *scale, float *where, float *kl_row, const float *u,
                              float step_bias, float explore_eps, const double *prior_f64, float *presence_prob,
                              float *presence, float *q, float *kl_per_sample, float *logp, float *step_weight,
                              const float *img, float *glimpse, int T, int B, int H, int W, int h, int w, int precision,
                              float guard_eps, void *stream) {
    AIR_REQUIRE(precision == AIR_PREC_F32 || precision == AIR_PREC_BF16, AIR_E_UNSUPPORTED);
    AIR_REQUIRE(tr_h && tr_w && tr_b && st_h && st_w && st_b && pre && logit && eps && loc && scale && where && kl_row &&
                    prior_f64 && presence_prob && presence && q && kl_per_sample && logp && step_weight && img &&
                    glimpse, AIR_E_NULL);                      // u == NULL: discrete_steps=False (presence = presence_prob, cell.py:150-151)
    AIR_REQUIRE(T > 0 && T <= 32 && tr_k > 0 && st_k > 0, AIR_E_SHAPE);
    int st = st_check_dims(B, H, W, h, w);
    if (st) return st;
    const int nq = (H * W) / 4;
    // the register-prefetch staging needs a 16-byte addressable image of at most 3 float4 per thread
    AIR_REQUIRE((H * W) % 4 == 0 && air_aligned16(img) && nq <= 3 * 1024 && air_aligned16(tr_w), AIR_E_UNSUPPORTED);
    size_t lds = carve_bytes(H * W, 0, w, h);
    AIR_REQUIRE(lds <= ST_MAX_LDS, AIR_E_UNSUPPORTED);
    AttendFwdArgs g;
    g.tr_h = tr_h; g.tr_w = tr_w; g.tr_b = tr_b; g.tr_k = tr_k; g.st_h = st_h; g.st_w = st_w; g.st_b = st_b; g.st_k = st_k;
    g.pre = pre; g.logit = logit; g.eps = eps; g.raw_offset = raw_offset; g.guard_eps = guard_eps;
    g.pl0 = p_loc_even; g.ps0 = p_scale_even; g.pl1 = p_loc_odd; g.ps1 = p_scale_odd;
    g.loc = loc; g.scale = scale; g.where = where; g.kl_row = kl_row; g.u = u; g.step_bias = step_bias;
    g.explore_eps = explore_eps; g.prior = prior_f64; g.prob = presence_prob; g.pres = presence; g.q = q;
    g.kl_ps = kl_per_sample; g.logp = logp; g.step_w = step_weight; g.img = img; g.glimpse = glimpse;
    g.T = T; g.B = B; g.H = H; g.W = W; g.h = h; g.w = w; g.stepx = lin_step(w); g.stepy = lin_step(h);
    g.bf16 = precision == AIR_PREC_BF16 ? 1 : 0;
    // one workgroup per glimpse while that is what fills the chip; beyond ~4 workgroups per CU one per image, which stages its
    // image once for the T reads (measured at batch 1024: 20 -> see DESIGN.md)
    g.img_major = ((long)T * B > 2048 && T > 1) ? 1 : 0;
    // image-major: the read role in the lean kernel's form when its outputs come in 16-byte groups (AIR_ATTEND_LEAN=0: the per-pixel form, A/B)
    const size_t lds_lean = read_lean_bytes(H, W, h, w, T) + 16 * (size_t)T;
    const char *env_lean = getenv("AIR_ATTEND_LEAN");
    g.lean = (g.img_major && w % 4 == 0 && air_aligned16(glimpse) && lds_lean <= ST_MAX_LDS && !(env_lean && atoi(env_lean) == 0)) ? 1 : 0;
    if (g.lean && lds_lean > lds) lds = lds_lean;
    const int grid = (g.img_major ? B : T * B) + air_cdiv(B, 64);
#define AIR_ATTEND_FWD_LAUNCH(MT_, NT_, EX_)                                                                         \
    do {                                                                                                                \
        int st_ = st_allow_lds(attend_fwd_kernel<MT_, NT_, EX_>, lds);                                                 \
        if (st_) return st_;                                                                                            \
        hipLaunchKernelGGL((attend_fwd_kernel<MT_, NT_, EX_>), dim3(grid), dim3(NT_), lds, air_stream(stream), g);     \
    } while (0)
#define AIR_ATTEND_FWD_BY_T(NT_)                                                                                      \
    do {                                                                                                                \
        if (T == 3) AIR_ATTEND_FWD_LAUNCH(3, NT_, true);                                                                \
        else if (T == 5) AIR_ATTEND_FWD_LAUNCH(5, NT_, true);                                                           \
        else if (T <= 8) AIR_ATTEND_FWD_LAUNCH(8, NT_, false);                                                          \
        else AIR_ATTEND_FWD_LAUNCH(32, NT_, false);                                                                     \
    } while (0)
    // 1024-thread workgroups sit one per CU (16 waves): a grid of more of them than CUs runs in two rounds (configs[3]: 320 glimpses,
    // the last 64 started when the first retired -- 3.6 us into an 8.3 us launch, traced); 512 threads with five groups each: two per CU
    if (nq <= 3 * 256) AIR_ATTEND_FWD_BY_T(256);
    else if (nq <= 5 * 512 && !g.img_major && (long)T * B > 256 && getenv("AIR_ATTEND_FWD_1024") == nullptr) AIR_ATTEND_FWD_BY_T(512);
    else AIR_ATTEND_FWD_BY_T(1024);
#undef AIR_ATTEND_FWD_BY_T
#undef AIR_ATTEND_FWD_LAUNCH
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// Backward counterpart: role A, one workgroup per glimpse k = t*B + b: d where (through the read) by the same staging /
// tables / fixed-order reduction as st_read_bwd_kernel, then -- in the same workgroup -- the where-sampling backward of row k
// (needs dwhere from the canvas write AND from the read, plus the KL term): d pre[k, 0:8].  Role B: backward of the
// num-steps KL / step weights / REINFORCE term wrt the steps logit (numsteps_presence_bwd_body).
struct AttendBwdArgs;
__device__ __forceinline__ float attend_dwhere_w(const AttendBwdArgs &g, size_t e);
struct AttendBwdArgs {
    const float *img, *where, *dglimpse; float *dwhere_r;
    const float *pre, *eps; float raw_offset, pl0, ps0, pl1, ps1, guard_eps;
    const float *loc, *scale, *dwhere_w, *dkl_row; float dkl_scale; float *dpre;
    int dwhere_w_slabs;   // dwhere_w[slabs][T*B][4]: the canvas backward may write its dwhere as several partial slabs (their sum, in order)
    const float *prob, *presence; const double *prior; float kl_scale; const float *kl_a, *kl_b; float w_scale;
    const float *dlogp, *dpres, *logit; float step_bias, explore_eps; float *dlogit;
    int T, B, H, W, h, w, vec4;
    double stepx, stepy;
    // optional: dX of the output layers of the transform / steps MLPs in the same launch (the two 8- and 1-deep products that
    // otherwise need a launch of their own on the backward chain):
    //   tr_dx[k, n] = (sum_o dpre[k, o] * tr_w[n, o]) * elu'(tr_y[k, n]),  n < tr_k   (tr_y = that layer's input activation;
    //   st_dx[k, n] = dlogit[k] * st_w[n]              * elu'(st_y[k, n]),  n < st_k    NULL: the input is not an ELU output)
    const float *tr_w, *tr_y; float *tr_dx; int tr_k, tr_ld;
    const float *st_w, *st_y; float *st_dx; int st_k, st_ld;
    int bf16;
    int img_major;      // role A: one workgroup per IMAGE runs the backward of its T glimpses (image staged once)
};
__device__ __forceinline__ float attend_dwhere_w(const AttendBwdArgs &g, size_t e) {
    float v = g.dwhere_w[e];
    const size_t slab = (size_t)g.T * g.B * 4;
    for (int q = 1; q < g.dwhere_w_slabs; ++q) v += g.dwhere_w[(size_t)q * slab + e];
    return v;
}

template <int MT, int NT, bool EXACT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT <= 256 ? 5 : 4))) void attend_bwd_kernel(AttendBwdArgs g) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wid = tid >> 6, nw = nt >> 6;
    const int T = EXACT ? MT : g.T, B = g.B;
    const int n = g.img_major ? B : T * B;                     // role-A workgroups
    const int nb_ = (int)gridDim.x - n, bid = (int)blockIdx.x - nb_;       // role B first (see attend_fwd_kernel)
    AIR_TR_INIT();
    if (bid < 0) {
        AIR_TR(4);
        if (g.st_dx == nullptr) {
            numsteps_presence_bwd_body<MT>((int)blockIdx.x, nb_, g.prob, g.presence, g.prior, g.kl_scale,
                                           g.kl_a, g.kl_b, g.w_scale, g.dlogp, g.logit, g.step_bias, g.explore_eps, g.dlogit,
                                           T, B, g.dpres);
        } else {
            // 16 batch columns per workgroup: threads 0..15 run the float64 chain of their column, then all threads form
            // st_dx for the 16 x T rows (their input activations were requested before the chain started)
            __shared__ float s_dl[MT][16];
            const int vb = (int)blockIdx.x, vgrid = nb_;
            for (int base = vb * 16; base < B; base += vgrid * 16) {
                const int nout = 16 * T * g.st_k;
                float yv[4], wv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {                  // output e = (c, t, nn): column c, step t, unit nn
                    const int e = tid + u * nt, ec = e < nout ? e : 0;
                    const int nn = ec % g.st_k, ct = ec / g.st_k, t = ct % T, c = ct / T;
                    const int b = base + c < B ? base + c : B - 1;
                    yv[u] = g.st_y ? g.st_y[((size_t)t * B + b) * g.st_ld + nn] : 1.f;
                    wv[u] = g.st_w[nn];
                }
                if (tid < 16 && base + tid < B) {
                    float dl[MT];
                    numsteps_presence_bwd_col<MT>(base + tid, g.prob, g.presence, g.prior, g.kl_scale, g.kl_a, g.kl_b, g.w_scale,
                                                  g.dlogp, g.logit, g.step_bias, g.explore_eps, dl, T, B, g.dpres);
#pragma unroll
                    for (int t = 0; t < MT; ++t) if (t < T) { g.dlogit[(size_t)t * B + base + tid] = dl[t]; s_dl[t][tid] = dl[t]; }
                }
                __syncthreads();
                for (int e0 = tid; e0 < nout; e0 += 4 * nt) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int e = e0 + u * nt;
                        if (e >= nout) break;
                        const int nn = e % g.st_k, ct = e / g.st_k, t = ct % T, c = ct / T;
                        if (base + c >= B) continue;
                        const size_t row = (size_t)t * B + base + c;
                        const float y = e0 == tid ? yv[u] : (g.st_y ? g.st_y[row * g.st_ld + nn] : 1.f);
                        const float w_ = e0 == tid ? wv[u] : g.st_w[nn];
                        float v = opnd(s_dl[t][c], g.bf16) * opnd(w_, g.bf16);
                        if (g.st_y) v *= (y > 0.f ? 1.f : y + 1.f);
                        g.st_dx[row * g.st_ld + nn] = v;
                    }
                }
                __syncthreads();
            }
        }
        AIR_TR(6);
        AIR_TR_FLUSH();
        return;
    }
    AIR_TR(0);
    const int b = bid % B;
    const int t0 = g.img_major ? 0 : bid / B, t1 = g.img_major ? T : t0 + 1;
    const int H = g.H, W = g.W, h = g.h, w = g.w, HW = H * W, hw = h * w;
    Carve c = carve_lds(smem, HW, 0, w, h);
    const float cxs = (float)((W - 1) / 2.0), cys = (float)((H - 1) / 2.0);
    // image-major with a few glimpses: three barriers for the whole image instead of three per glimpse -- the pixel passes of all T
    // glimpses run back to back (per-glimpse partial sums side by side in the unused axis-table area of the carve), then wave
    // t mod nw finishes glimpse t (final reduction + where-sampling backward: the T tails in parallel), then every thread takes
    // its share of the T x tr_k outputs of the fused transform-layer dX
    if (g.img_major && T * (nw * 8 + 8) <= 3 * w + 3 * h + 160) {
        float *part = reinterpret_cast<float *>(c.fx);         // [T][nw * 8]
        float *dps = part + T * nw * 8;                        // [T][8]
        stage_to_lds(c.src, g.img + (size_t)b * HW, HW, g.vec4 != 0);
        __syncthreads();
        AIR_TR(1);
        const int z0 = opaque_zero();
        for (int t = 0; t < T; ++t) {
            const size_t k = (size_t)t * B + b;
            const float sx = g.where[4 * k + z0], tx = g.where[4 * k + 1 + z0];
            const float sy = g.where[4 * k + 2 + z0], ty = g.where[4 * k + 3 + z0];
            const float *go_p = g.dglimpse + k * hw;
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int p = tid; p < hw; p += nt) {
                const float go = go_p[p];
                const int i = p / w, j = p - i * w;
                const float X = lin_m11(j, w, g.stepx), Y = lin_m11(i, h, g.stepy);
                int fx, fy; float dx, dy;
                axis_entry(grid_coord(sx, X, tx, cxs), W, &fx, &dx);
                axis_entry(grid_coord(sy, Y, ty, cys), H, &fy, &dy);
                if (fx == ST_INVALID || fy == ST_INVALID) continue;
                const Taps tp = load_taps_sel(c.src, H, W, fy, fx);
                const float gx = dy * (tp.fc - tp.ff) + (1.f - dy) * (tp.cc - tp.cf);
                const float gy = dx * (tp.cf - tp.ff) + (1.f - dx) * (tp.cc - tp.fc);
                const float ax = go * gx * cxs, ay = go * gy * cys;
                acc[0] += ax * X; acc[1] += ax;
                acc[2] += ay * Y; acc[3] += ay;
            }
            const float r = wave_reduce8(acc);
            if ((lane & 7) == 0) part[t * nw * 8 + wid * 8 + wave_reduce8_slot()] = r;
        }
        __syncthreads();
        AIR_TR(2);
        for (int t = wid; t < T; t += nw) {
            const size_t k = (size_t)t * B + b;
            float pr[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) pr[q] = (lane < nw && q < 4) ? part[t * nw * 8 + lane * 8 + q] : 0.f;
            const float tot = wave_reduce8(pr);
            const float accd = __shfl(tot, 8 * (lane & 3), 64);
            if (lane < 4) {
                const int d_ = lane;
                const size_t e = k * 4 + d_;
                const float mu = g.loc[e], sc = g.scale[e], s_dw = attend_dwhere_w(g, e), s_eps = g.eps[e];
                const float s_raw = g.pre[k * 8 + 4 + d_] + g.raw_offset;
                const float s_dk = g.dkl_row ? g.dkl_row[k] * g.dkl_scale : 0.f;
                g.dwhere_r[4 * k + d_] = accd;
                const float pm = (d_ & 1) ? g.pl1 : g.pl0, ps = (d_ & 1) ? g.ps1 : g.ps0;
                const float ds = s_dw + accd;
                float dmu = ds + s_dk * kl_mean_diff(mu, pm) / (ps * ps);
                const float dsc = ds * s_eps + normal_kl_dscale(s_dk, sc, ps);
                dmu *= (d_ & 1) ? (1.f - mu * mu) : mu * (1.f - mu);
                float dsp = s_raw > 20.f ? 1.f : sigmoid_acc(s_raw);
                if (g.guard_eps > 0.f && sc <= g.guard_eps) dsp = 0.f;
                g.dpre[k * 8 + d_] = dmu;
                g.dpre[k * 8 + 4 + d_] = dsc * dsp;
                dps[t * 8 + d_] = dmu; dps[t * 8 + 4 + d_] = dsc * dsp;
            }
        }
        if (g.tr_dx) {
            __syncthreads();
            for (int e = tid; e < T * g.tr_k; e += nt) {
                const int t = e / g.tr_k, nn = e - t * g.tr_k;
                const size_t k = (size_t)t * B + b;
                const float4 wa = *reinterpret_cast<const float4 *>(g.tr_w + (size_t)nn * 8);
                const float4 wb = *reinterpret_cast<const float4 *>(g.tr_w + (size_t)nn * 8 + 4);
                const float *dpt = dps + t * 8;
                float v = opnd(dpt[0], g.bf16) * opnd(wa.x, g.bf16);
                v += opnd(dpt[1], g.bf16) * opnd(wa.y, g.bf16); v += opnd(dpt[2], g.bf16) * opnd(wa.z, g.bf16);
                v += opnd(dpt[3], g.bf16) * opnd(wa.w, g.bf16); v += opnd(dpt[4], g.bf16) * opnd(wb.x, g.bf16);
                v += opnd(dpt[5], g.bf16) * opnd(wb.y, g.bf16); v += opnd(dpt[6], g.bf16) * opnd(wb.z, g.bf16);
                v += opnd(dpt[7], g.bf16) * opnd(wb.w, g.bf16);
                if (g.tr_y) { const float y = g.tr_y[k * g.tr_ld + nn]; v *= (y > 0.f ? 1.f : y + 1.f); }
                g.tr_dx[k * g.tr_ld + nn] = v;
            }
        }
        AIR_TR(3);
        AIR_TR_FLUSH();
        return;
    }
  for (int t = t0; t < t1; ++t) {                              // one glimpse, or (image-major) the T glimpses of image b
    const int k = t * B + b;
    // every operand of this unit is requested before anything waits: the image, the `where` row, the incoming glimpse
    // gradient of this thread's (first four) pixels, and -- lanes 0..3 of wave 0 -- the operands of the where-sampling backward
    const int z0 = opaque_zero();                              // vector-path loads of the wave-uniform `where` row (see opaque_zero)
    const float sx = g.where[4 * (size_t)k + z0], tx = g.where[4 * (size_t)k + 1 + z0];
    const float sy = g.where[4 * (size_t)k + 2 + z0], ty = g.where[4 * (size_t)k + 3 + z0];
    const float *go_p = g.dglimpse + (size_t)k * hw;
    float gov[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int p = tid + u * nt; gov[u] = go_p[p < hw ? p : hw - 1]; }     // clamped, branch-free
    // fused transform-layer dX: this thread's weight row(s) and input activation(s), requested now
    float4 twa[2], twb[2]; float tyv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        twa[u] = make_float4(0.f, 0.f, 0.f, 0.f); twb[u] = twa[u]; tyv[u] = 1.f;
        if (g.tr_dx) {
            const int nn = tid + u * nt, nc = nn < g.tr_k ? nn : g.tr_k - 1;
            twa[u] = *reinterpret_cast<const float4 *>(g.tr_w + (size_t)nc * 8);
            twb[u] = *reinterpret_cast<const float4 *>(g.tr_w + (size_t)nc * 8 + 4);
            if (g.tr_y) tyv[u] = g.tr_y[(size_t)k * g.tr_ld + nc];
        }
    }
    float s_mu = 0.f, s_sc = 1.f, s_dw = 0.f, s_eps = 0.f, s_raw = 0.f, s_dk = 0.f;
    if (tid < 4) {
        const size_t e = (size_t)k * 4 + tid;
        s_mu = g.loc[e]; s_sc = g.scale[e]; s_dw = attend_dwhere_w(g, e); s_eps = g.eps[e];
        s_raw = g.pre[(size_t)k * 8 + 4 + tid] + g.raw_offset;
        s_dk = g.dkl_row ? g.dkl_row[k] * g.dkl_scale : 0.f;
    }
    if (t == t0) stage_to_lds(c.src, g.img + (size_t)b * HW, HW, g.vec4 != 0);
    __syncthreads();                                           // image staged / the previous glimpse's scratch consumers are done
    AIR_TR(1);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int p0 = tid; p0 < hw; p0 += 4 * nt) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * nt;
            if (p >= hw) break;
            const float go = p0 == tid ? gov[u] : go_p[p];
            const int i = p / w, j = p - i * w;
            const float X = lin_m11(j, w, g.stepx), Y = lin_m11(i, h, g.stepy);
            int fx, fy; float dx, dy;
            axis_entry(grid_coord(sx, X, tx, cxs), W, &fx, &dx);
            axis_entry(grid_coord(sy, Y, ty, cys), H, &fy, &dy);
            if (fx == ST_INVALID || fy == ST_INVALID) continue;
            const Taps t = load_taps_sel(c.src, H, W, fy, fx);
            const float gx = dy * (t.fc - t.ff) + (1.f - dy) * (t.cc - t.cf);
            const float gy = dx * (t.cf - t.ff) + (1.f - dx) * (t.cc - t.fc);
            const float ax = go * gx * cxs, ay = go * gy * cys;
            acc[0] += ax * X; acc[1] += ax;
            acc[2] += ay * Y; acc[3] += ay;
        }
    }
    {
        const float r = wave_reduce8(acc);
        if ((lane & 7) == 0) c.scratch[wid * 8 + wave_reduce8_slot()] = r;
    }
    __syncthreads();
    AIR_TR(2);
    if (wid == 0) {
        float part[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) part[q] = (lane < nw && q < 4) ? c.scratch[lane * 8 + q] : 0.f;
        const float tot = wave_reduce8(part);                  // lane 8*d holds the total of dimension d
        const float accd = __shfl(tot, 8 * (lane & 3), 64);
        if (lane < 4) {
            // d where through the read, then the where-sampling backward of row k (gauss_bwd_body with D = 4, loc_mode = 1):
            // one dimension per lane
            const int d_ = lane;
            g.dwhere_r[4 * (size_t)k + d_] = accd;
            const float mu = s_mu, sc = s_sc;
            const float pm = (d_ & 1) ? g.pl1 : g.pl0, ps = (d_ & 1) ? g.ps1 : g.ps0;
            const float ds = s_dw + accd;
            float dmu = ds + s_dk * kl_mean_diff(mu, pm) / (ps * ps);
            const float dsc = ds * s_eps + normal_kl_dscale(s_dk, sc, ps);
            dmu *= (d_ & 1) ? (1.f - mu * mu) : mu * (1.f - mu);
            float dsp = s_raw > 20.f ? 1.f : sigmoid_acc(s_raw);
            if (g.guard_eps > 0.f && sc <= g.guard_eps) dsp = 0.f;
            g.dpre[(size_t)k * 8 + d_] = dmu;
            g.dpre[(size_t)k * 8 + 4 + d_] = dsc * dsp;
            if (g.tr_dx) { c.scratch[128 + d_] = dmu; c.scratch[128 + 4 + d_] = dsc * dsp; }     // (scratch[0:128] holds the partial sums)
        }
    }
    if (g.tr_dx) {
        __syncthreads();
        float dp[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) dp[o] = opnd(c.scratch[128 + o], g.bf16);
        for (int n0 = tid; n0 < g.tr_k; n0 += 2 * nt) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int nn = n0 + u * nt;
                if (nn >= g.tr_k) break;
                float4 wa = twa[u], wb = twb[u]; float y = tyv[u];
                if (n0 != tid) {
                    wa = *reinterpret_cast<const float4 *>(g.tr_w + (size_t)nn * 8);
                    wb = *reinterpret_cast<const float4 *>(g.tr_w + (size_t)nn * 8 + 4);
                    if (g.tr_y) y = g.tr_y[(size_t)k * g.tr_ld + nn];
                }
                float v = dp[0] * opnd(wa.x, g.bf16);
                v += dp[1] * opnd(wa.y, g.bf16); v += dp[2] * opnd(wa.z, g.bf16); v += dp[3] * opnd(wa.w, g.bf16);
                v += dp[4] * opnd(wb.x, g.bf16); v += dp[5] * opnd(wb.y, g.bf16); v += dp[6] * opnd(wb.z, g.bf16);
                v += dp[7] * opnd(wb.w, g.bf16);
                if (g.tr_y) v *= (y > 0.f ? 1.f : y + 1.f);
                g.tr_dx[(size_t)k * g.tr_ld + nn] = v;
            }
        }
    }
  }
    AIR_TR(3);
    AIR_TR_FLUSH();
}

static int attend_bwd_launch(AttendBwdArgs &g, int T, int B, int H, int W, int h, int w, size_t lds, void *stream) {
    const int role_b = g.st_dx ? air_cdiv(B, 16) : air_cdiv(B, 64);
    g.img_major = ((long)T * B > 2048 && T > 1) ? 1 : 0;       // as air_attend_fwd
    const int grid = (g.img_major ? B : T * B) + role_b;
    const int hw_ = h * w;
#define AIR_ATTEND_BWD_LAUNCH(MT_, NT_, EX_)                                                                         \
    do {                                                                                                                \
        int st_ = st_allow_lds(attend_bwd_kernel<MT_, NT_, EX_>, lds);                                                 \
        if (st_) return st_;                                                                                            \
        hipLaunchKernelGGL((attend_bwd_kernel<MT_, NT_, EX_>), dim3(grid), dim3(NT_), lds, air_stream(stream), g);     \
    } while (0)
#define AIR_ATTEND_BWD_BY_T(NT_)                                                                                      \
    do {                                                                                                                \
        if (T == 3) AIR_ATTEND_BWD_LAUNCH(3, NT_, true);                                                                \
        else if (T == 5) AIR_ATTEND_BWD_LAUNCH(5, NT_, true);                                                           \
        else if (T <= 8) AIR_ATTEND_BWD_LAUNCH(8, NT_, false);                                                          \
        else AIR_ATTEND_BWD_LAUNCH(32, NT_, false);                                                                     \
    } while (0)
    // about one glimpse pixel per thread
    // (image-major: 256 threads, so that one workgroup per image -- B of them -- is resident at once: 5 per CU)
    // (512 threads at 28x28 with 320 units -- two workgroups per CU instead of two rounds, as air_attend_fwd does -- measured slower
    //  in the configs[3] step: 13.9 against 12.0 us, two pixels per thread on the long per-pixel path)
    if (hw_ <= 256 || (g.img_major && hw_ <= 1024)) AIR_ATTEND_BWD_BY_T(256);
    else if (hw_ <= 512) AIR_ATTEND_BWD_BY_T(512); else AIR_ATTEND_BWD_BY_T(1024);
#undef AIR_ATTEND_BWD_BY_T
#undef AIR_ATTEND_BWD_LAUNCH
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

extern "C" int air_attend_bwd(const float *img, const float *where, const float *dglimpse, float *dwhere_r,
                              const float *pre, const float *eps, float raw_offset, float p_loc_even, float p_scale_even,
                              float p_loc_odd, float p_scale_odd, const float *loc, const float *scale,
                              const float *dwhere_w, int dwhere_w_slabs, const float *dkl_row, float dkl_scale, float *dpre,
                              const float *presence_prob, const float *presence, const double *prior_f64, float kl_scale,
                              const float *kl_row_a, const float *kl_row_b, float w_scale, const float *dlogp,
                              const float *dpresence, const float *logit, float step_bias, float explore_eps, float *dlogit, int T, int B, int H,
                              int W, int h, int w, float guard_eps, void *stream) {
    AIR_REQUIRE(img && where && dglimpse && dwhere_r && pre && eps && loc && scale && dwhere_w && dpre && presence_prob &&
                    prior_f64 && logit && dlogit, AIR_E_NULL);
    AIR_REQUIRE(!dlogp || presence, AIR_E_NULL);
    AIR_REQUIRE(T > 0 && T <= 32 && dwhere_w_slabs >= 1 && dwhere_w_slabs <= 4, AIR_E_SHAPE);
    int st = st_check_dims(B, H, W, h, w);
    if (st) return st;
    const size_t lds = carve_bytes(H * W, 0, w, h);
    AIR_REQUIRE(lds <= ST_MAX_LDS, AIR_E_UNSUPPORTED);
    AttendBwdArgs g;
    g.img = img; g.where = where; g.dglimpse = dglimpse; g.dwhere_r = dwhere_r; g.pre = pre; g.eps = eps;
    g.raw_offset = raw_offset; g.guard_eps = guard_eps; g.pl0 = p_loc_even; g.ps0 = p_scale_even; g.pl1 = p_loc_odd; g.ps1 = p_scale_odd;
    g.loc = loc; g.scale = scale; g.dwhere_w = dwhere_w; g.dwhere_w_slabs = dwhere_w_slabs; g.dkl_row = dkl_row; g.dkl_scale = dkl_scale; g.dpre = dpre;
    g.prob = presence_prob; g.presence = presence; g.prior = prior_f64; g.kl_scale = kl_scale; g.kl_a = kl_row_a;
    g.kl_b = kl_row_b; g.w_scale = w_scale; g.dlogp = dlogp; g.dpres = dpresence; g.logit = logit; g.step_bias = step_bias;
    g.explore_eps = explore_eps; g.dlogit = dlogit; g.T = T; g.B = B; g.H = H; g.W = W; g.h = h; g.w = w;
    g.vec4 = (((H * W) % 4 == 0) && air_aligned16(img)) ? 1 : 0;
    g.stepx = lin_step(w); g.stepy = lin_step(h);
    g.tr_w = nullptr; g.tr_y = nullptr; g.tr_dx = nullptr; g.tr_k = 0; g.tr_ld = 0;
    g.st_w = nullptr; g.st_y = nullptr; g.st_dx = nullptr; g.st_k = 0; g.st_ld = 0; g.bf16 = 0;
    return attend_bwd_launch(g, T, B, H, W, h, w, lds, stream);
}

// air_attend_bwd + the dX of the two MLP output layers (see AttendBwdArgs): tr_dx[T*B, tr_k], st_dx[T*B, st_k]
extern "C" int air_attend_bwd_dx(const float *img, const float *where, const float *dglimpse, float *dwhere_r,
                              const float *pre, const float *eps, float raw_offset, float p_loc_even, float p_scale_even,
                              float p_loc_odd, float p_scale_odd, const float *loc, const float *scale,
                              const float *dwhere_w, int dwhere_w_slabs, const float *dkl_row, float dkl_scale, float *dpre,
                              const float *presence_prob, const float *presence, const double *prior_f64, float kl_scale,
                              const float *kl_row_a, const float *kl_row_b, float w_scale, const float *dlogp,
                              const float *dpresence, const float *logit, float step_bias, float explore_eps, float *dlogit, int T, int B, int H,
                              int W, int h, int w, const float *tr_w, const float *tr_y, float *tr_dx, int tr_k, int tr_ld,
                                 const float *st_w, const float *st_y, float *st_dx, int st_k, int st_ld, int precision,
                                 float guard_eps, void *stream) {
    AIR_REQUIRE(img && where && dglimpse && dwhere_r && pre && eps && loc && scale && dwhere_w && dpre && presence_prob &&
                    prior_f64 && logit && dlogit, AIR_E_NULL);
    AIR_REQUIRE(!dlogp || presence, AIR_E_NULL);
    AIR_REQUIRE(T > 0 && T <= 32 && dwhere_w_slabs >= 1 && dwhere_w_slabs <= 4, AIR_E_SHAPE);
    int st = st_check_dims(B, H, W, h, w);
    if (st) return st;
    const size_t lds = carve_bytes(H * W, 0, w, h);
    AIR_REQUIRE(lds <= ST_MAX_LDS, AIR_E_UNSUPPORTED);
    AttendBwdArgs g;
    g.img = img; g.where = where; g.dglimpse = dglimpse; g.dwhere_r = dwhere_r; g.pre = pre; g.eps = eps;
    g.raw_offset = raw_offset; g.guard_eps = guard_eps; g.pl0 = p_loc_even; g.ps0 = p_scale_even; g.pl1 = p_loc_odd; g.ps1 = p_scale_odd;
    g.loc = loc; g.scale = scale; g.dwhere_w = dwhere_w; g.dwhere_w_slabs = dwhere_w_slabs; g.dkl_row = dkl_row; g.dkl_scale = dkl_scale; g.dpre = dpre;
    g.prob = presence_prob; g.presence = presence; g.prior = prior_f64; g.kl_scale = kl_scale; g.kl_a = kl_row_a;
    g.kl_b = kl_row_b; g.w_scale = w_scale; g.dlogp = dlogp; g.dpres = dpresence; g.logit = logit; g.step_bias = step_bias;
    g.explore_eps = explore_eps; g.dlogit = dlogit; g.T = T; g.B = B; g.H = H; g.W = W; g.h = h; g.w = w;
    g.vec4 = (((H * W) % 4 == 0) && air_aligned16(img)) ? 1 : 0;
    g.stepx = lin_step(w); g.stepy = lin_step(h);
    AIR_REQUIRE(tr_w && tr_dx && st_w && st_dx && tr_k > 0 && st_k > 0 && tr_ld >= tr_k && st_ld >= st_k, AIR_E_NULL);
    AIR_REQUIRE(air_aligned16(tr_w), AIR_E_ALIGN);
    AIR_REQUIRE(precision == AIR_PREC_F32 || precision == AIR_PREC_BF16, AIR_E_UNSUPPORTED);
    g.tr_w = tr_w; g.tr_y = tr_y; g.tr_dx = tr_dx; g.tr_k = tr_k; g.tr_ld = tr_ld;
    g.st_w = st_w; g.st_y = st_y; g.st_dx = st_dx; g.st_k = st_k; g.st_ld = st_ld; g.bf16 = precision == AIR_PREC_BF16 ? 1 : 0;
    return attend_bwd_launch(g, T, B, H, W, h, w, lds, stream);
}
