// Spatial-transformer kernels for gfx950: fused affine-grid + bilinear glimpse read, and the inverse canvas write.
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
#include "air_common.h"
#include "nvil_device.h"

#define ST_THREADS 256
#define ST_INVALID INT_MIN

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
    const bool valid = (coord > -1.0f) && (coord < (float)extent);   // NaN -> invalid
    const float fl = floorf(coord);
    *f_out = valid ? (int)fl : ST_INVALID;
    *d_out = (fl + 1.0f) - coord;
}

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
    return sizeof(float) * (size_t)(((cnt_src + 3) & ~3) + ((cnt_aux + 3) & ~3) + 3 * nx + 3 * ny + 128);
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
// LDS-only workgroup barrier: waits for this wave's LDS traffic, NOT for its outstanding global loads (a plain
// __syncthreads() drains vmcnt too, which would serialise the register prefetch of the next image behind the barrier).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// (three NAMED float4 registers per thread: an indexed register array is demoted to scratch by hipcc here, which
//  serialises every load behind a scratch store; blockDim.x = 256 covers images up to 768 float4, 1024 up to 3072)
__global__ __launch_bounds__(1024) void st_read_fwd_pipe_kernel(
    const float *__restrict__ img, const float *__restrict__ where, float *__restrict__ out,
    int n, int n_img, int H, int W, int h, int w, double stepx, double stepy) {
    extern __shared__ __align__(16) float smem[];
    const int HW = H * W, hw = h * w, tid = threadIdx.x, nt = blockDim.x, nq = HW >> 2;
    Carve c = carve_lds(smem, HW, 0, w, h);
    const float cxs = (float)((W - 1) / 2.0), cys = (float)((H - 1) / 2.0);
    const float4 *where4 = reinterpret_cast<const float4 *>(where);
    const int q0 = tid < nq ? tid : nq - 1, q1 = tid + nt < nq ? tid + nt : nq - 1;
    const int q2 = tid + 2 * nt < nq ? tid + 2 * nt : nq - 1;
    float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0, p2 = p0;
    int b = blockIdx.x;
    if (b < n_img) {
        const float4 *s4 = reinterpret_cast<const float4 *>(img + (size_t)b * HW);
        p0 = s4[q0]; p1 = s4[q1]; p2 = s4[q2];
    }
    float4 wnext = (b < n_img) ? where4[b] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = tid; a < w + h; a += nt) {                   // the linspace tables do not depend on the image
        if (a < w) c.X[a] = lin_m11(a, w, stepx); else c.Y[a - w] = lin_m11(a - w, h, stepy);
    }
    float4 *d4 = reinterpret_cast<float4 *>(c.src);
    for (; b < n_img; b += gridDim.x) {
        lds_barrier();                                        // readers of the previous image are done
        if (tid < nq) d4[tid] = p0;
        if (tid + nt < nq) d4[tid + nt] = p1;
        if (tid + 2 * nt < nq) d4[tid + 2 * nt] = p2;
        const int nb = b + gridDim.x;
        if (nb < n_img) {                                     // next image: in flight during this image's compute
            const float4 *s4 = reinterpret_cast<const float4 *>(img + (size_t)nb * HW);
            p0 = s4[q0]; p1 = s4[q1]; p2 = s4[q2];
        }
        for (int k = b; k < n; k += n_img) {
            const float4 wk = wnext;
            const bool more = k + n_img < n;
            if (more || nb < n_img) wnext = where4[more ? k + n_img : nb];
            lds_barrier();
            for (int a = tid; a < w + h; a += nt) {
                if (a < w) axis_entry(grid_coord(wk.x, c.X[a], wk.y, cxs), W, &c.fx[a], &c.dx[a]);
                else axis_entry(grid_coord(wk.z, c.Y[a - w], wk.w, cys), H, &c.fy[a - w], &c.dy[a - w]);
            }
            lds_barrier();
            float *o = out + (size_t)k * hw;
            for (int p = tid; p < hw; p += nt) {
                const int i = p / w, j = p - i * w;
                const int fx = c.fx[j], fy = c.fy[i];
                float v = 0.f;
                if (fx != ST_INVALID && fy != ST_INVALID) v = bilerp(load_taps(c.src, H, W, fy, fx), c.dx[j], c.dy[i]);
                o[p] = v;
            }
        }
    }
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
// write: canvas += presence * bilinear(glimpse; x_g = (w-1)/2*(X_J/sx - tx/sx + 1), y_g likewise)
// One workgroup per image accumulates all T steps in LDS, optionally emitting every intermediate canvas and the
// per-sample reconstruction term of the final canvas.
// ============================================================================================================
// Single-phase form: ALL T glimpses of an image and their T axis tables are staged in LDS behind ONE barrier, then
// each thread walks its canvas pixels with the running canvas in a register (t inner, in order, so the accumulation is
// the oracle's ((0 + p0*v0) + p1*v1) + ...).  One memory round trip per image instead of one per step.
struct CarveWr {
    float *glm, *dx, *dy, *X, *Y, *pres, *scratch;
    int *fx, *fy;
    int hwp;
};
__device__ __forceinline__ CarveWr carve_wr(float *smem, int T, int H, int W, int h, int w) {
    CarveWr c;
    c.hwp = (h * w + 3) & ~3;
    float *p = smem;
    c.glm = p; p += (size_t)T * c.hwp;
    c.fx = reinterpret_cast<int *>(p); p += T * W;
    c.dx = p; p += T * W;
    c.fy = reinterpret_cast<int *>(p); p += T * H;
    c.dy = p; p += T * H;
    c.X = p; p += W;
    c.Y = p; p += H;
    c.pres = p; p += (T + 3) & ~3;
    c.scratch = p;
    return c;
}
static inline size_t carve_wr_bytes(int T, int H, int W, int h, int w) {
    return sizeof(float) * ((size_t)T * ((h * w + 3) & ~3) + 2 * (size_t)T * (W + H) + W + H + ((T + 3) & ~3) + 128);
}

__global__ __launch_bounds__(1024) void st_write_fwd_kernel(
    const float *__restrict__ glimpse, const float *__restrict__ where, const float *__restrict__ presence,
    const float *__restrict__ canvas_in, const float *__restrict__ obs,
    float *__restrict__ canvas_steps, float *__restrict__ final_canvas, float *__restrict__ rec,
    int T, int B, int H, int W, int h, int w, double stepX, double stepY, float mult, float std, int vec4_canvas,
    int vec4_glimpse) {
    extern __shared__ __align__(16) float smem[];
    const int HW = H * W, hw = h * w, tid = threadIdx.x, nt = blockDim.x;
    CarveWr c = carve_wr(smem, T, H, W, h, w);
    const float cxs = (float)((w - 1) / 2.0), cys = (float)((h - 1) / 2.0);
    const float cst = 0.5f * logf(6.283185307179586f) + logf(std);
    for (int a = tid; a < W + H; a += nt) {           // linspace tables: image independent
        if (a < W) c.X[a] = lin_m11(a, W, stepX); else c.Y[a - W] = lin_m11(a - W, H, stepY);
    }
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();                                      // previous image done (and X/Y visible)
        for (int t = 0; t < T; ++t)
            stage_to_lds(c.glm + (size_t)t * c.hwp, glimpse + ((size_t)t * B + b) * hw, hw, vec4_glimpse != 0);
        for (int a = tid; a < T * (W + H); a += nt) {
            const int t = a / (W + H), r = a - t * (W + H);
            const float *wk = where + 4 * ((size_t)t * B + b);
            if (r < W) {
                const float sx = wk[0], tx = wk[1];
                axis_entry(grid_coord(1.0f / sx, c.X[r], -tx / sx, cxs), w, &c.fx[t * W + r], &c.dx[t * W + r]);
            } else {
                const float sy = wk[2], ty = wk[3];
                const int i = r - W;
                axis_entry(grid_coord(1.0f / sy, c.Y[i], -ty / sy, cys), h, &c.fy[t * H + i], &c.dy[t * H + i]);
            }
        }
        if (tid < T) c.pres[tid] = presence ? presence[(size_t)tid * B + b] : 1.0f;
        __syncthreads();
        float s[1] = {0.f};
        for (int p = tid; p < HW; p += nt) {
            const int I = p / W, J = p - I * W;
            const float xo = rec ? obs[(size_t)b * HW + p] : 0.f;
            float acc = canvas_in ? canvas_in[(size_t)b * HW + p] : 0.f;
            for (int t = 0; t < T; ++t) {
                const int fx = c.fx[t * W + J], fy = c.fy[t * H + I];
                float v = 0.f;
                if (fx != ST_INVALID && fy != ST_INVALID)
                    v = bilerp(load_taps(c.glm + (size_t)t * c.hwp, h, w, fy, fx), c.dx[t * W + J], c.dy[t * H + I]);
                acc = acc + c.pres[t] * v;
                if (canvas_steps) canvas_steps[((size_t)t * B + b) * HW + p] = acc;
            }
            if (final_canvas) final_canvas[(size_t)b * HW + p] = acc;
            if (rec) {
                const float z = (xo - mult * acc) / std;
                s[0] += 0.5f * z * z + cst;
            }
        }
        if (rec) {
            block_sum<1>(s, c.scratch);
            if (tid == 0) rec[b] = s[0];
        }
    }
}

// Backward of the write for every (t, b): dglimpse, dwhere, optional dpresence.
// dcanvas either given per step ([T*B,H,W]) or formed on the fly from the reconstruction term:
//   dcanvas[b,p] = loss_scale * mult * (mult*final[b,p] - obs[b,p]) / std^2   (shared by all t)
// dglimpse is the transpose of a separable bilinear map, dG = Wy^T . g . Wx with two non-zeros per row of Wy / Wx,
// evaluated as two small LDS passes in a fixed order (no float atomics => bitwise reproducible):
//   T1[I,j] = sum_J g[I,J] * wx[J,j]   over the contiguous J-range that touches glimpse column j
//   dG[i,j] = sum_I wy[I,i] * T1[I,j]  over the contiguous I-range that touches glimpse row i
struct CarveBwd {
    float *src, *g, *t1, *dx, *X, *dy, *Y, *scratch;
    int *fx, *fy, *jlo, *jhi, *ilo, *ihi;
};
__device__ __forceinline__ CarveBwd carve_bwd(float *smem, int H, int W, int h, int w) {
    CarveBwd c;
    float *p = smem;
    c.src = p; p += (h * w + 3) & ~3;
    c.g = p; p += (H * W + 3) & ~3;
    c.t1 = p; p += (H * w + 3) & ~3;
    c.fx = reinterpret_cast<int *>(p); p += W;
    c.dx = p; p += W;
    c.X = p; p += W;
    c.fy = reinterpret_cast<int *>(p); p += H;
    c.dy = p; p += H;
    c.Y = p; p += H;
    c.jlo = reinterpret_cast<int *>(p); p += w;
    c.jhi = reinterpret_cast<int *>(p); p += w;
    c.ilo = reinterpret_cast<int *>(p); p += h;
    c.ihi = reinterpret_cast<int *>(p); p += h;
    c.scratch = p;
    return c;
}
static inline size_t carve_bwd_bytes(int H, int W, int h, int w) {
    return sizeof(float) * (size_t)(((h * w + 3) & ~3) + ((H * W + 3) & ~3) + ((H * w + 3) & ~3) + 3 * W + 3 * H +
                                    2 * w + 2 * h + 128);
}

__global__ __launch_bounds__(1024) void st_write_bwd_kernel(
    const float *__restrict__ glimpse, const float *__restrict__ where, const float *__restrict__ presence,
    const float *__restrict__ dcanvas, const float *__restrict__ final_canvas, const float *__restrict__ obs,
    float *__restrict__ dglimpse, float *__restrict__ dwhere, float *__restrict__ dpresence,
    int T, int B, int H, int W, int h, int w, double stepX, double stepY, float mult, float std, float loss_scale,
    int vec4_glimpse, NvilArgs nv) {
    extern __shared__ __align__(16) float smem[];
    // optional second role: the LAST workgroup evaluates the NVIL objective (independent of the canvas gradient; it only
    // has to precede the baseline / logit backward that follow this launch)
    const int grid_st = nv.imp ? (int)gridDim.x - 1 : (int)gridDim.x;
    if ((int)blockIdx.x >= grid_st) {
        nvil_body(nv);
        return;
    }
    const int HW = H * W, hw = h * w, tid = threadIdx.x, nt = blockDim.x;
    CarveBwd c = carve_bwd(smem, H, W, h, w);
    const float cxs = (float)((w - 1) / 2.0), cys = (float)((h - 1) / 2.0);
    const float coef = loss_scale * mult / (std * std);
    const int n = T * B;
    for (int k = blockIdx.x; k < n; k += grid_st) {
        const int b = k % B;
        __syncthreads();
        stage_to_lds(c.src, glimpse + (size_t)k * hw, hw, vec4_glimpse != 0);
        {   // incoming canvas gradient -> LDS up front: these global loads do not depend on the tables, so their
            // latency overlaps the table construction instead of sitting inside the per-pixel dependency chain
            const float *dcp = dcanvas ? dcanvas + (size_t)k * HW : nullptr;
            const float *fcp = final_canvas ? final_canvas + (size_t)b * HW : nullptr;
            const float *obp = obs ? obs + (size_t)b * HW : nullptr;
            for (int p = tid; p < HW; p += nt) c.g[p] = dcp ? dcp[p] : coef * (mult * fcp[p] - obp[p]);
        }
        for (int a = tid; a < w + h; a += nt) {               // empty index ranges (lo > hi)
            if (a < w) { c.jlo[a] = W; c.jhi[a] = -1; } else { c.ilo[a - w] = H; c.ihi[a - w] = -1; }
        }
        const float sx = where[4 * (size_t)k + 0], tx = where[4 * (size_t)k + 1];
        const float sy = where[4 * (size_t)k + 2], ty = where[4 * (size_t)k + 3];
        const float ax = 1.0f / sx, bx = -tx / sx;
        const float ay = 1.0f / sy, by = -ty / sy;
        __syncthreads();
        // axis tables; each canvas column / row also registers itself in the contiguous source range of the (at most
        // two) glimpse columns / rows it touches -- integer LDS min / max atomics: order independent, deterministic
        for (int a = tid; a < W + H; a += nt) {
            if (a < W) {
                const float X = lin_m11(a, W, stepX);
                c.X[a] = X;
                int f; float d;
                axis_entry(grid_coord(ax, X, bx, cxs), w, &f, &d);
                c.fx[a] = f; c.dx[a] = d;
                if (f != ST_INVALID) {
                    if (f >= 0) { atomicMin(&c.jlo[f], a); atomicMax(&c.jhi[f], a); }
                    if (f + 1 <= w - 1) { atomicMin(&c.jlo[f + 1], a); atomicMax(&c.jhi[f + 1], a); }
                }
            } else {
                const int i = a - W;
                const float Y = lin_m11(i, H, stepY);
                c.Y[i] = Y;
                int f; float d;
                axis_entry(grid_coord(ay, Y, by, cys), h, &f, &d);
                c.fy[i] = f; c.dy[i] = d;
                if (f != ST_INVALID) {
                    if (f >= 0) { atomicMin(&c.ilo[f], i); atomicMax(&c.ihi[f], i); }
                    if (f + 1 <= h - 1) { atomicMin(&c.ilo[f + 1], i); atomicMax(&c.ihi[f + 1], i); }
                }
            }
        }
        __syncthreads();
        const float pres = presence ? presence[k] : 1.0f;
        float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};      // d/d(ax), d/d(bx), d/d(ay), d/d(by), dpresence
        for (int p = tid; p < HW; p += nt) {
            const int I = p / W, J = p - I * W;
            const int fx = c.fx[J], fy = c.fy[I];
            float go = 0.f;
            if (fx != ST_INVALID && fy != ST_INVALID) {
                const float dc = c.g[p];
                const float dx = c.dx[J], dy = c.dy[I];
                const Taps t = load_taps(c.src, h, w, fy, fx);
                const float v = bilerp(t, dx, dy);
                const float gx = dy * (t.fc - t.ff) + (1.f - dy) * (t.cc - t.cf);
                const float gy = dx * (t.cf - t.ff) + (1.f - dx) * (t.cc - t.fc);
                go = pres * dc;
                const float gax = go * gx * cxs, gay = go * gy * cys;
                acc[0] += gax * c.X[J]; acc[1] += gax;
                acc[2] += gay * c.Y[I]; acc[3] += gay;
                acc[4] += dc * v;
            }
            c.g[p] = go;
        }
        __syncthreads();
        for (int e = tid; e < H * w; e += nt) {            // pass 1: contract canvas columns
            const int I = e / w, j = e - I * w;
            float s = 0.f;
            if (c.fy[I] != ST_INVALID) {
                const float *grow = c.g + I * W;
                for (int J = c.jlo[j]; J <= c.jhi[j]; ++J) {
                    const int fx = c.fx[J];
                    if (fx == ST_INVALID) continue;
                    const float dx = c.dx[J];
                    const float wgt = (fx == j ? dx : 0.f) + (fx + 1 == j ? 1.f - dx : 0.f);
                    s += grow[J] * wgt;
                }
            }
            c.t1[e] = s;
        }
        __syncthreads();
        float *dg = dglimpse + (size_t)k * hw;
        for (int e = tid; e < hw; e += nt) {               // pass 2: contract canvas rows
            const int i = e / w, j = e - i * w;
            float s = 0.f;
            for (int I = c.ilo[i]; I <= c.ihi[i]; ++I) {
                const int fy = c.fy[I];
                if (fy == ST_INVALID) continue;
                const float dy = c.dy[I];
                const float wgt = (fy == i ? dy : 0.f) + (fy + 1 == i ? 1.f - dy : 0.f);
                s += c.t1[I * w + j] * wgt;
            }
            dg[e] = s;
        }
        block_sum<5>(acc, c.scratch);
        if (tid == 0) {
            // chain through a = 1/s, b = -t/s
            float *d = dwhere + 4 * (size_t)k;
            d[0] = acc[0] * (-1.0f / (sx * sx)) + acc[1] * (tx / (sx * sx));
            d[1] = acc[1] * (-1.0f / sx);
            d[2] = acc[2] * (-1.0f / (sy * sy)) + acc[3] * (ty / (sy * sy));
            d[3] = acc[3] * (-1.0f / sy);
            if (dpresence) dpresence[k] = acc[4];
        }
    }
}

// ============================================================================================================
// host side
// ============================================================================================================
static inline double lin_step(int n) { return n > 1 ? 2.0 / (double)(n - 1) : 0.0; }
static inline int st_grid(int items) {
    const int cap = 256 * 8;   // 256 CUs x up to 8 resident 256-thread workgroups; grid-stride beyond that
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
    if (vec4 && air_aligned16(where) && nq <= 3 * 1024) {
        const int threads = nq <= 3 * 256 ? 256 : 1024;
        { int st_ = st_allow_lds(st_read_fwd_pipe_kernel, lds); if (st_) return st_; }
        hipLaunchKernelGGL(st_read_fwd_pipe_kernel, dim3(st_grid(n_img)), dim3(threads), lds, air_stream(stream), img,
                           where, glimpse, n, n_img, H, W, h, w, lin_step(w), lin_step(h));
        AIR_LAUNCH_CHECK();
        return AIR_OK;
    }
    { int st_ = st_allow_lds(st_read_fwd_kernel, lds); if (st_) return st_; }
    hipLaunchKernelGGL(st_read_fwd_kernel, dim3(st_grid(n_img)), dim3(ST_THREADS), lds, air_stream(stream), img, where,
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
    hipLaunchKernelGGL(st_read_bwd_kernel, dim3(st_grid(per_glimpse ? n : n_img)), dim3(ST_THREADS), lds,
                       air_stream(stream), img, where, dglimpse, dwhere, dimg, n, n_img, H, W, h, w, lin_step(w),
                       lin_step(h), vec4, per_glimpse);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

static int launch_write_fwd(const float *glimpse, const float *where, const float *presence, const float *canvas_in,
                            const float *obs, float *canvas_steps, float *final_canvas, float *rec, int T, int B,
                            int H, int W, int h, int w, float mult, float std, void *stream) {
    const size_t lds = carve_wr_bytes(T, H, W, h, w);
    AIR_REQUIRE(lds <= ST_MAX_LDS, AIR_E_UNSUPPORTED);
    const int vec4c = ((H * W) % 4 == 0) && (!canvas_in || air_aligned16(canvas_in)) &&
                      (!final_canvas || air_aligned16(final_canvas));
    const int vec4g = ((h * w) % 4 == 0) && air_aligned16(glimpse);
    { int st_ = st_allow_lds(st_write_fwd_kernel, lds); if (st_) return st_; }
    // few images: the chip is mostly idle, so give each image 16 waves (4 per SIMD) to hide LDS / global latency
    const int wr_threads = (long)B * T <= 4096 ? 1024 : ST_THREADS;
    hipLaunchKernelGGL(st_write_fwd_kernel, dim3(st_grid(B)), dim3(wr_threads), lds, air_stream(stream), glimpse, where,
                       presence, canvas_in, obs, canvas_steps, final_canvas, rec, T, B, H, W, h, w, lin_step(W),
                       lin_step(H), mult, std, vec4c, vec4g);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

extern "C" int air_st_write_fwd(const float *glimpse, const float *where, const float *presence,
                                const float *canvas_in, float *canvas_out, int n, int H, int W, int h, int w,
                                void *stream) {
    AIR_REQUIRE(glimpse && where && canvas_out, AIR_E_NULL);
    int st = st_check_dims(n, H, W, h, w);
    if (st) return st;
    return launch_write_fwd(glimpse, where, presence, canvas_in, nullptr, nullptr, canvas_out, nullptr, 1, n, H, W, h,
                            w, 1.0f, 1.0f, stream);
}

extern "C" int air_canvas_unroll_fwd(const float *glimpse, const float *where, const float *presence,
                                     const float *obs, float *canvas_steps, float *final_canvas,
                                     float *rec_per_sample, int T, int B, int H, int W, int h, int w, float mult,
                                     float std, void *stream) {
    AIR_REQUIRE(glimpse && where && (final_canvas || canvas_steps), AIR_E_NULL);
    AIR_REQUIRE(!rec_per_sample || obs, AIR_E_NULL);
    AIR_REQUIRE(T > 0, AIR_E_SHAPE);
    int st = st_check_dims(B, H, W, h, w);
    if (st) return st;
    return launch_write_fwd(glimpse, where, presence, nullptr, obs, canvas_steps, final_canvas, rec_per_sample, T, B,
                            H, W, h, w, mult, std, stream);
}

static int launch_write_bwd(const float *glimpse, const float *where, const float *presence, const float *dcanvas,
                            const float *final_canvas, const float *obs, float *dglimpse, float *dwhere,
                            float *dpresence, int T, int B, int H, int W, int h, int w, float mult, float std,
                            float loss_scale, void *stream, const NvilArgs *nvil = nullptr) {
    const size_t lds = carve_bwd_bytes(H, W, h, w);
    NvilArgs nv = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    if (nvil) nv = *nvil;
    AIR_REQUIRE(lds <= ST_MAX_LDS, AIR_E_UNSUPPORTED);
    const int vec4g = ((h * w) % 4 == 0) && air_aligned16(glimpse);
    { int st_ = st_allow_lds(st_write_bwd_kernel, lds); if (st_) return st_; }
    const int wr_threads = (long)B * T <= 4096 ? 1024 : ST_THREADS;
    hipLaunchKernelGGL(st_write_bwd_kernel, dim3(st_grid(T * B) + (nvil ? 1 : 0)), dim3(wr_threads), lds,
                       air_stream(stream), glimpse, where, presence, dcanvas, final_canvas, obs, dglimpse, dwhere,
                       dpresence, T, B, H, W, h, w, lin_step(W), lin_step(H), mult, std, loss_scale, vec4g, nv);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

extern "C" int air_st_write_bwd(const float *glimpse, const float *where, const float *presence,
                                const float *dcanvas, float *dglimpse, float *dwhere, float *dpresence, int n, int H,
                                int W, int h, int w, void *stream) {
    AIR_REQUIRE(glimpse && where && dcanvas && dglimpse && dwhere, AIR_E_NULL);
    int st = st_check_dims(n, H, W, h, w);
    if (st) return st;
    return launch_write_bwd(glimpse, where, presence, dcanvas, nullptr, nullptr, dglimpse, dwhere, dpresence, 1, n, H,
                            W, h, w, 1.0f, 1.0f, 1.0f, stream);
}

extern "C" int air_canvas_unroll_bwd(const float *glimpse, const float *where, const float *presence,
                                     const float *obs, const float *final_canvas, float *dglimpse, float *dwhere,
                                     int T, int B, int H, int W, int h, int w, float mult, float std,
                                     float loss_scale, void *stream) {
    AIR_REQUIRE(glimpse && where && obs && final_canvas && dglimpse && dwhere, AIR_E_NULL);
    AIR_REQUIRE(T > 0, AIR_E_SHAPE);
    int st = st_check_dims(B, H, W, h, w);
    if (st) return st;
    return launch_write_bwd(glimpse, where, presence, nullptr, final_canvas, obs, dglimpse, dwhere, nullptr, T, B, H, W,
                            h, w, mult, std, loss_scale, stream);
}

extern "C" int air_canvas_unroll_bwd_nvil(const float *glimpse, const float *where, const float *presence,
                                          const float *obs, const float *final_canvas, float *dglimpse, float *dwhere,
                                          int T, int B, int H, int W, int h, int w, float mult, float std,
                                          float loss_scale, const float *imp, const float *baseline, const float *logp,
                                          float *nvil_out, float *dlogp, float *dbaseline, void *stream) {
    AIR_REQUIRE(glimpse && where && obs && final_canvas && dglimpse && dwhere, AIR_E_NULL);
    AIR_REQUIRE(imp && baseline && logp && nvil_out, AIR_E_NULL);
    AIR_REQUIRE(T > 0, AIR_E_SHAPE);
    int st = st_check_dims(B, H, W, h, w);
    if (st) return st;
    const NvilArgs nv = {imp, baseline, logp, nvil_out, dlogp, dbaseline, B};
    return launch_write_bwd(glimpse, where, presence, nullptr, final_canvas, obs, dglimpse, dwhere, nullptr, T, B, H, W,
                            h, w, mult, std, loss_scale, stream, &nv);
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
    const float *eps; float raw_offset, pl0, ps0, pl1, ps1;
    float *loc, *scale, *where, *kl_row;
    const float *u; float step_bias, explore_eps; const double *prior;
    float *prob, *pres, *q, *kl_ps, *logp, *step_w;
    const float *img; float *glimpse;
    int T, B, H, W, h, w, bf16;
    double stepx, stepy;
};
// operand of a dense product: as is, or rounded to bf16 (EngineConfig.mfma_dtype = "bf16": same arithmetic as the MFMA path)
__device__ __forceinline__ float opnd(float v, int bf16) { return bf16 ? (float)(__bf16)v : v; }

template <int MT>
__global__ __launch_bounds__(1024) void attend_fwd_kernel(AttendFwdArgs g) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const int T = g.T, B = g.B;
    if ((int)blockIdx.x >= B) {
        // ---- role B: steps-predictor output layer + presence / num-steps for 64 batch columns ----------------------
        const int vb = (int)blockIdx.x - B, vgrid = (int)gridDim.x - B;
        // steps-predictor output layer for this block's 64 columns x T rows.  Thread (c = tid & 63, part = tid >> 6) sums a
        // quarter of the K range of row (t, c) for every t -- all its loads are independent, so the whole layer is ONE memory
        // round trip (a row-at-a-time loop measured 8 us: a dozen dependent round trips) -- then the 4 parts meet in LDS.
        __shared__ float s_part[4][MT][64];
        const float bias = g.st_b[0];
        const int cc = tid & 63, part = tid >> 6;
        const int chunk = (g.st_k + 3) >> 2, k0 = part * chunk, k1 = (k0 + chunk < g.st_k) ? k0 + chunk : g.st_k;
        for (int base = vb * 64; base < B; base += vgrid * 64) {
            const int b = base + cc;
            if (part < 4) {
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    float acc = 0.f;
                    if (t < T && b < B) {
                        const float *x = g.st_h + ((size_t)t * B + b) * g.st_k;
#pragma unroll 16
                        for (int k = k0; k < k1; ++k) acc += opnd(x[k], g.bf16) * opnd(g.st_w[k], g.bf16);
                    }
                    s_part[part][t][cc] = acc;
                }
            }
            __syncthreads();
            if (tid < 64 && b < B) {
                for (int t = 0; t < T; ++t)
                    g.logit[(size_t)t * B + b] = ((s_part[0][t][cc] + s_part[1][t][cc]) + (s_part[2][t][cc] + s_part[3][t][cc])) + bias;
            }
            __syncthreads();
        }
        __syncthreads();                                       // (same thread re-reads what it wrote; compiler fence)
        presence_numsteps_fwd_body<MT>(vb, vgrid, g.logit, g.u, g.step_bias, g.explore_eps, g.prior, g.prob, g.pres, g.q,
                                       g.kl_ps, g.logp, g.step_w, T, B);
        return;
    }
    // ---- role A: image b ------------------------------------------------------------------------------------------
    const int b = blockIdx.x;
    const int H = g.H, W = g.W, h = g.h, w = g.w, HW = H * W, hw = h * w, nq = HW >> 2;
    Carve c = carve_lds(smem, HW, 0, w, h);
    float *s_where = c.scratch;                                // [T][4] (the carve reserves 128 floats; T <= 28)
    const float cxs = (float)((W - 1) / 2.0), cys = (float)((H - 1) / 2.0);
    const int q0 = tid < nq ? tid : nq - 1, q1 = tid + nt < nq ? tid + nt : nq - 1;
    const int q2 = tid + 2 * nt < nq ? tid + 2 * nt : nq - 1;
    const float4 *s4 = reinterpret_cast<const float4 *>(g.img + (size_t)b * HW);
    const float4 p0 = s4[q0], p1 = s4[q1], p2 = s4[q2];       // in flight while the output layer below runs
    for (int a = tid; a < w + h; a += nt) {
        if (a < w) c.X[a] = lin_m11(a, w, g.stepx); else c.Y[a - w] = lin_m11(a - w, h, g.stepy);
    }
    // transform output layer: one wave per time step (row t*B + b)
    for (int t = wave; t < T; t += nw) {
        const size_t m = (size_t)t * B + b;
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
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o] = wave_sum_all(acc[o]);
        if (lane < 4) {
            const int d = lane;
            const float e_loc = (d == 0 ? acc[0] : d == 1 ? acc[1] : d == 2 ? acc[2] : acc[3]) + g.tr_b[d];
            const float e_raw = (d == 0 ? acc[4] : d == 1 ? acc[5] : d == 2 ? acc[6] : acc[7]) + g.tr_b[4 + d];
            g.pre[m * 8 + d] = e_loc;
            g.pre[m * 8 + 4 + d] = e_raw;
            const float mu = (d & 1) ? tanhf(e_loc) : sigmoid_acc(e_loc);                    // modules.py:41-46
            const float s = softplus_acc(e_raw + g.raw_offset);
            const size_t o = m * 4 + d;
            const float v = mu + s * g.eps[o];                                              // cell.py:130-133
            g.loc[o] = mu; g.scale[o] = s; g.where[o] = v;
            s_where[4 * t + d] = v;
            float kl = (d & 1) ? normal_kl(mu, s, g.pl1, g.ps1) : normal_kl(mu, s, g.pl0, g.ps0);
            kl += __shfl_xor(kl, 1, 64);
            kl += __shfl_xor(kl, 2, 64);
            if (d == 0) g.kl_row[m] = kl;
        }
    }
    float4 *d4 = reinterpret_cast<float4 *>(c.src);
    if (tid < nq) d4[tid] = p0;
    if (tid + nt < nq) d4[tid + nt] = p1;
    if (tid + 2 * nt < nq) d4[tid + 2 * nt] = p2;
    for (int t = 0; t < T; ++t) {
        __syncthreads();                                       // image + s_where visible / previous tables consumed
        const float sx = s_where[4 * t + 0], tx = s_where[4 * t + 1], sy = s_where[4 * t + 2], ty = s_where[4 * t + 3];
        for (int a = tid; a < w + h; a += nt) {
            if (a < w) axis_entry(grid_coord(sx, c.X[a], tx, cxs), W, &c.fx[a], &c.dx[a]);
            else axis_entry(grid_coord(sy, c.Y[a - w], ty, cys), H, &c.fy[a - w], &c.dy[a - w]);
        }
        __syncthreads();
        float *o = g.glimpse + ((size_t)t * B + b) * hw;
        for (int p = tid; p < hw; p += nt) {
            const int i = p / w, j = p - i * w;
            const int fx = c.fx[j], fy = c.fy[i];
            float v = 0.f;
            if (fx != ST_INVALID && fy != ST_INVALID) v = bilerp(load_taps(c.src, H, W, fy, fx), c.dx[j], c.dy[i]);
            o[p] = v;
        }
    }
}

extern "C" int air_attend_fwd(const float *tr_h, const float *tr_w, const float *tr_b, int tr_k, const float *st_h,
                              const float *st_w, const float *st_b, int st_k, float *pre, float *logit,
                              const float *eps, float raw_offset, float p_loc_even, float p_scale_even, float p_loc_odd,
                              float p_scale_odd, float *loc, float *scale, float *where, float *kl_row, const float *u,
                              float step_bias, float explore_eps, const double *prior_f64, float *presence_prob,
                              float *presence, float *q, float *kl_per_sample, float *logp, float *step_weight,
                              const float *img, float *glimpse, int T, int B, int H, int W, int h, int w, int precision,
                              void *stream) {
    AIR_REQUIRE(precision == AIR_PREC_F32 || precision == AIR_PREC_BF16, AIR_E_UNSUPPORTED);
    AIR_REQUIRE(tr_h && tr_w && tr_b && st_h && st_w && st_b && pre && logit && eps && loc && scale && where && kl_row &&
                    u && prior_f64 && presence_prob && presence && q && kl_per_sample && logp && step_weight && img &&
                    glimpse, AIR_E_NULL);
    AIR_REQUIRE(T > 0 && T <= 28 && tr_k > 0 && st_k > 0, AIR_E_SHAPE);   // s_where lives in the 128-float carve pad
    int st = st_check_dims(B, H, W, h, w);
    if (st) return st;
    const int nq = (H * W) / 4;
    // the register-prefetch staging needs a 16-byte addressable image of at most 3 float4 per thread
    AIR_REQUIRE((H * W) % 4 == 0 && air_aligned16(img) && nq <= 3 * 1024 && air_aligned16(tr_w), AIR_E_UNSUPPORTED);
    const size_t lds = carve_bytes(H * W, 0, w, h);
    AIR_REQUIRE(lds <= ST_MAX_LDS, AIR_E_UNSUPPORTED);
    AttendFwdArgs g;
    g.tr_h = tr_h; g.tr_w = tr_w; g.tr_b = tr_b; g.tr_k = tr_k; g.st_h = st_h; g.st_w = st_w; g.st_b = st_b; g.st_k = st_k;
    g.pre = pre; g.logit = logit; g.eps = eps; g.raw_offset = raw_offset;
    g.pl0 = p_loc_even; g.ps0 = p_scale_even; g.pl1 = p_loc_odd; g.ps1 = p_scale_odd;
    g.loc = loc; g.scale = scale; g.where = where; g.kl_row = kl_row; g.u = u; g.step_bias = step_bias;
    g.explore_eps = explore_eps; g.prior = prior_f64; g.prob = presence_prob; g.pres = presence; g.q = q;
    g.kl_ps = kl_per_sample; g.logp = logp; g.step_w = step_weight; g.img = img; g.glimpse = glimpse;
    g.T = T; g.B = B; g.H = H; g.W = W; g.h = h; g.w = w; g.stepx = lin_step(w); g.stepy = lin_step(h);
    g.bf16 = precision == AIR_PREC_BF16 ? 1 : 0;
    const int threads = nq <= 3 * 256 ? 256 : 1024;
    const int grid = B + air_cdiv(B, 64);
    if (T <= 8) {
        { int st_ = st_allow_lds(attend_fwd_kernel<8>, lds); if (st_) return st_; }
        hipLaunchKernelGGL(attend_fwd_kernel<8>, dim3(grid), dim3(threads), lds, air_stream(stream), g);
    } else {
        { int st_ = st_allow_lds(attend_fwd_kernel<32>, lds); if (st_) return st_; }
        hipLaunchKernelGGL(attend_fwd_kernel<32>, dim3(grid), dim3(threads), lds, air_stream(stream), g);
    }
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// Backward counterpart: role A, one workgroup per glimpse k = t*B + b: d where (through the read) by the same staging /
// tables / fixed-order reduction as st_read_bwd_kernel, then -- in the same workgroup -- the where-sampling backward of row k
// (needs dwhere from the canvas write AND from the read, plus the KL term): d pre[k, 0:8].  Role B: backward of the
// num-steps KL / step weights / REINFORCE term wrt the steps logit (numsteps_presence_bwd_body).
struct AttendBwdArgs {
    const float *img, *where, *dglimpse; float *dwhere_r;
    const float *pre, *eps; float raw_offset, pl0, ps0, pl1, ps1;
    const float *loc, *scale, *dwhere_w, *dkl_row; float dkl_scale; float *dpre;
    const float *prob, *presence; const double *prior; float kl_scale; const float *kl_a, *kl_b; float w_scale;
    const float *dlogp, *logit; float step_bias, explore_eps; float *dlogit;
    int T, B, H, W, h, w, vec4;
    double stepx, stepy;
};

template <int MT>
__global__ __launch_bounds__(1024) void attend_bwd_kernel(AttendBwdArgs g) {
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int T = g.T, B = g.B, n = T * B;
    if ((int)blockIdx.x >= n) {
        numsteps_presence_bwd_body<MT>((int)blockIdx.x - n, (int)gridDim.x - n, g.prob, g.presence, g.prior, g.kl_scale,
                                       g.kl_a, g.kl_b, g.w_scale, g.dlogp, g.logit, g.step_bias, g.explore_eps, g.dlogit,
                                       T, B);
        return;
    }
    const int k = blockIdx.x, b = k % B;
    const int H = g.H, W = g.W, h = g.h, w = g.w, HW = H * W, hw = h * w;
    Carve c = carve_lds(smem, HW, 0, w, h);
    const float cxs = (float)((W - 1) / 2.0), cys = (float)((H - 1) / 2.0);
    stage_to_lds(c.src, g.img + (size_t)b * HW, HW, g.vec4 != 0);
    const float sx = g.where[4 * (size_t)k + 0], tx = g.where[4 * (size_t)k + 1];
    const float sy = g.where[4 * (size_t)k + 2], ty = g.where[4 * (size_t)k + 3];
    for (int a = tid; a < w + h; a += nt) {
        if (a < w) {
            const float X = lin_m11(a, w, g.stepx);
            c.X[a] = X;
            axis_entry(grid_coord(sx, X, tx, cxs), W, &c.fx[a], &c.dx[a]);
        } else {
            const float Y = lin_m11(a - w, h, g.stepy);
            c.Y[a - w] = Y;
            axis_entry(grid_coord(sy, Y, ty, cys), H, &c.fy[a - w], &c.dy[a - w]);
        }
    }
    __syncthreads();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float *go_p = g.dglimpse + (size_t)k * hw;
    for (int p = tid; p < hw; p += nt) {
        const int i = p / w, j = p - i * w;
        const int fx = c.fx[j], fy = c.fy[i];
        if (fx == ST_INVALID || fy == ST_INVALID) continue;
        const float dx = c.dx[j], dy = c.dy[i], go = go_p[p];
        const Taps t = load_taps(c.src, H, W, fy, fx);
        const float gx = dy * (t.fc - t.ff) + (1.f - dy) * (t.cc - t.cf);
        const float gy = dx * (t.cf - t.ff) + (1.f - dx) * (t.cc - t.fc);
        const float ax = go * gx * cxs, ay = go * gy * cys;
        acc[0] += ax * c.X[j]; acc[1] += ax;
        acc[2] += ay * c.Y[i]; acc[3] += ay;
    }
    block_sum<4>(acc, c.scratch);                              // valid in wave 0 (every lane after its wave_sum? no: lane 0)
    if (tid == 0) {
        float *d = g.dwhere_r + 4 * (size_t)k;
        d[0] = acc[0]; d[1] = acc[1]; d[2] = acc[2]; d[3] = acc[3];
        // where-sampling backward of row k (gauss_bwd_body with D = 4, loc_mode = 1), four dims by this one thread
#pragma unroll
        for (int d_ = 0; d_ < 4; ++d_) {
            const size_t e = (size_t)k * 4 + d_;
            const float mu = g.loc[e], s = g.scale[e];
            const float pm = (d_ & 1) ? g.pl1 : g.pl0, ps = (d_ & 1) ? g.ps1 : g.ps0;
            const float ds = g.dwhere_w[e] + acc[d_];
            const float dk = g.dkl_row ? g.dkl_row[k] * g.dkl_scale : 0.f;
            float dmu = ds + dk * (mu - pm) / (ps * ps);
            const float dsc = ds * g.eps[e] + dk * (s / (ps * ps) - 1.f / s);
            dmu *= (d_ & 1) ? (1.f - mu * mu) : mu * (1.f - mu);
            const float raw = g.pre[(size_t)k * 8 + 4 + d_] + g.raw_offset;
            const float dsp = raw > 20.f ? 1.f : sigmoid_acc(raw);
            g.dpre[(size_t)k * 8 + d_] = dmu;
            g.dpre[(size_t)k * 8 + 4 + d_] = dsc * dsp;
        }
    }
}

extern "C" int air_attend_bwd(const float *img, const float *where, const float *dglimpse, float *dwhere_r,
                              const float *pre, const float *eps, float raw_offset, float p_loc_even, float p_scale_even,
                              float p_loc_odd, float p_scale_odd, const float *loc, const float *scale,
                              const float *dwhere_w, const float *dkl_row, float dkl_scale, float *dpre,
                              const float *presence_prob, const float *presence, const double *prior_f64, float kl_scale,
                              const float *kl_row_a, const float *kl_row_b, float w_scale, const float *dlogp,
                              const float *logit, float step_bias, float explore_eps, float *dlogit, int T, int B, int H,
                              int W, int h, int w, void *stream) {
    AIR_REQUIRE(img && where && dglimpse && dwhere_r && pre && eps && loc && scale && dwhere_w && dpre && presence_prob &&
                    prior_f64 && logit && dlogit, AIR_E_NULL);
    AIR_REQUIRE(!dlogp || presence, AIR_E_NULL);
    AIR_REQUIRE(T > 0 && T <= 32, AIR_E_SHAPE);
    int st = st_check_dims(B, H, W, h, w);
    if (st) return st;
    const size_t lds = carve_bytes(H * W, 0, w, h);
    AIR_REQUIRE(lds <= ST_MAX_LDS, AIR_E_UNSUPPORTED);
    AttendBwdArgs g;
    g.img = img; g.where = where; g.dglimpse = dglimpse; g.dwhere_r = dwhere_r; g.pre = pre; g.eps = eps;
    g.raw_offset = raw_offset; g.pl0 = p_loc_even; g.ps0 = p_scale_even; g.pl1 = p_loc_odd; g.ps1 = p_scale_odd;
    g.loc = loc; g.scale = scale; g.dwhere_w = dwhere_w; g.dkl_row = dkl_row; g.dkl_scale = dkl_scale; g.dpre = dpre;
    g.prob = presence_prob; g.presence = presence; g.prior = prior_f64; g.kl_scale = kl_scale; g.kl_a = kl_row_a;
    g.kl_b = kl_row_b; g.w_scale = w_scale; g.dlogp = dlogp; g.logit = logit; g.step_bias = step_bias;
    g.explore_eps = explore_eps; g.dlogit = dlogit; g.T = T; g.B = B; g.H = H; g.W = W; g.h = h; g.w = w;
    g.vec4 = (((H * W) % 4 == 0) && air_aligned16(img)) ? 1 : 0;
    g.stepx = lin_step(w); g.stepy = lin_step(h);
    const int grid = T * B + air_cdiv(B, 64);
    if (T <= 8) {
        { int st_ = st_allow_lds(attend_bwd_kernel<8>, lds); if (st_) return st_; }
        hipLaunchKernelGGL(attend_bwd_kernel<8>, dim3(grid), dim3(ST_THREADS), lds, air_stream(stream), g);
    } else {
        { int st_ = st_allow_lds(attend_bwd_kernel<32>, lds); if (st_) return st_; }
        hipLaunchKernelGGL(attend_bwd_kernel<32>, dim3(grid), dim3(ST_THREADS), lds, air_stream(stream), g);
    }
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
