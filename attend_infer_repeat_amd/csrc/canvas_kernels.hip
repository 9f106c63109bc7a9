// Inverse spatial transformer ("canvas write", cell.py:159-165 / modules.py:101-109) for gfx950.
//
//   forward : canvas_t = canvas_{t-1} + presence_t * bilinear(glimpse_t; x_g = (w-1)/2*(X_J/sx - tx/sx + 1), y_g likewise), all T
//             steps of an image in one pass: per-step canvases, final canvas and the band's share of the Gaussian
//             log-likelihood (model.py:319-324);
//   backward: dglimpse_t = Wy^T . (presence_t * dcanvas) . Wx, dwhere_t, optional dpresence_t, for every (t, b).
//
// The forward's arithmetic is the oracle's, operation by operation (st_device.h: bilerp, axis_entry; air_common.h: grid_coord,
// lin_m11 -- each carries its own `fp contract(off)`; the forward body sets it for its own accumulation), so its results are
// bit-identical to oracle/air_oracle.py st_write; the recompute form of the backward re-forms the canvas with the same calls.
// Everything else in this file contracts to FMA.
// Measured and rejected in round 4 (profiles/r04_canvas_rowstream_ab.txt, r04_canvas_lean_bwd.txt): a row-streaming form of these
// kernels (lane = canvas column: parity-green, 1.5-3x slower) and a stripped stored-canvas backward for the throughput regime
// (no full-image dcanvas staging, two pixels in flight per thread: -7 % at mid scales without spills, slower at 6-8 waves per SIMD
// with them).  At 65536 images the backward costs 0.9 ms + 0.6 us per thousand footprint pixels: two thirds of it is the
// per-unit fixed part (tables, staging, the two contractions, three barriers), not the pixel pass.
#include <stdlib.h>
#include "st_device.h"
#include "nvil_device.h"

// [first, last] index of an axis table whose floor index lies in [f_lo, f_hi] (the map is monotone: the set is an interval); every
// lane gets the result; empty => first > last
__device__ __forceinline__ int2 floor_span(const float2 *tab, int n, int f_lo, int f_hi) {
    const int lane = threadIdx.x & 63;
    int first = n, last = -1;
    for (int base = 0; base < n; base += 64) {
        const int k = base + lane;
        const int f = __float_as_int(tab[k < n ? k : n - 1].x);
        const bool v = k < n && f != ST_INVALID && f >= f_lo && f <= f_hi;
        const unsigned long long m = __ballot(v);
        if (m) {
            const int lo = base + (int)__ffsll((long long)m) - 1, hi = base + 63 - (int)__clzll((long long)m);
            first = lo < first ? lo : first;
            last = hi > last ? hi : last;
        }
    }
    return make_int2(first, last);
}
// The backward's per-pixel arithmetic with its FMAs spelled out (and implicit contraction off): the recompute form, the stored-canvas
// form and the split / unsplit instantiations are separate compilations of the same expressions, and "the same bits in every form" (the
// tests compare them with torch.equal) must not depend on which products the compiler chooses to fuse in each.
__device__ __forceinline__ float dcanvas_of(float coef, float mult, float canvas, float obs) {
#pragma clang fp contract(off)
    return coef * __builtin_fmaf(mult, canvas, -obs);
}
// canvas accumulation step, rounded as the oracle's `canvas + presence * inversed` (cell.py:164)
__device__ __forceinline__ float acc_step(float acc, float p, float v) {
#pragma clang fp contract(off)
    const float pv = p * v;
    return acc + pv;
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
    float *glm, *pres, *scratch;
    float2 *xe, *ye;                 // per (t, column) / (t, band row): {floor index as int bits | ST_INVALID, d}
    int hwp;
};
__device__ __forceinline__ CarveWr carve_wr(float *smem, int T, int RB, int W, int h, int w) {
    CarveWr c;
    c.hwp = pad_count(h, w);
    float *p = smem;
    c.glm = p; p += (size_t)T * c.hwp;
    c.xe = reinterpret_cast<float2 *>(p); p += 2 * T * W;
    c.ye = reinterpret_cast<float2 *>(p); p += 2 * T * RB;
    c.pres = p; p += (T + 3) & ~3;
    c.scratch = p;
    return c;
}
static inline size_t carve_wr_bytes(int T, int RB, int W, int h, int w) {
    return sizeof(float) * ((size_t)T * pad_count_host(h, w) + 2 * (size_t)T * (W + RB) + ((T + 3) & ~3) + 128);
}
// One workgroup per (image, row band): band `q` of `NB` covers canvas rows [q*RB, min(H, (q+1)*RB)).  A batch of 64 images in
// 4 bands fills the 256 CUs (one workgroup per image left three quarters of the chip idle while each busy CU was bound by
// VALU issue: 2500 pixels x T steps x ~45 instructions on 4 SIMDs).  Every global operand (all T glimpses, the `where` rows,
// presence, this thread's observation pixels) is requested up front -- one memory round trip -- then ONE barrier, then each
// thread walks its pixels with the running canvas in a register (t inner, in order, so the accumulation is the oracle's
// ((0 + p0*v0) + p1*v1) + ...).  rec_parts[q*B + b] receives the band's share of the reconstruction term; with NB = 1 that
// IS rec[b], with NB > 1 the consumer (air_nvil_parts / air_canvas_unroll_bwd_nvil / air_sum_leading) adds the NB shares
// in band order (no float atomics: bitwise reproducible).
struct WriteFwdArgs {
    const float *glimpse, *where, *presence, *canvas_in, *obs;
    float *canvas_steps, *final_canvas, *rec_parts;
    int T, B, NB, RB, H, W, h, w;
    double stepX, stepY;
    float mult, std;
    int vec4_glimpse;
};
// (vblock of vgrid: the workgroup's index among the workgroups that run this role -- the whole grid of st_write_fwd_kernel, the
//  second part of the grid of canvas_fused_gs_kernel)
__device__ __forceinline__ void st_write_fwd_body(const WriteFwdArgs &a, float *smem, const int vblock, const int vgrid) {
#pragma clang fp contract(off)
    const float *__restrict__ glimpse = a.glimpse, *__restrict__ where = a.where, *__restrict__ presence = a.presence;
    const float *__restrict__ canvas_in = a.canvas_in, *__restrict__ obs = a.obs;
    float *__restrict__ canvas_steps = a.canvas_steps, *__restrict__ final_canvas = a.final_canvas, *__restrict__ rec_parts = a.rec_parts;
    const int T = a.T, B = a.B, NB = a.NB, RB = a.RB, H = a.H, W = a.W, h = a.h, w = a.w, vec4_glimpse = a.vec4_glimpse;
    const double stepX = a.stepX, stepY = a.stepY;
    const float mult = a.mult, std = a.std;
    AIR_TR_INIT();
    const int HW = H * W, hw = h * w, tid = threadIdx.x, nt = blockDim.x;
    CarveWr c = carve_wr(smem, T, RB, W, h, w);
    const float cxs = (float)((w - 1) / 2.0), cys = (float)((h - 1) / 2.0);
    const float cst = 0.5f * logf(6.283185307179586f) + logf(std);
    const int n_units = B * NB;
    const int pitch = w + 2;
    const float inv_w = 1.0f / (float)w, inv_W = 1.0f / (float)W;
    // the zero borders of the T bordered glimpses: written once, never overwritten (visible after the first barrier below)
    for (int e = tid; e < T * pad_border(h, w); e += nt) {
        const int t = e / pad_border(h, w);
        c.glm[(size_t)t * c.hwp + pad_border_index(e - t * pad_border(h, w), h, w)] = 0.f;
    }
    for (int unit = vblock; unit < n_units; unit += vgrid) {
        const int b = unit % B, band = unit / B;
        const int r0 = band * RB, r1 = (r0 + RB < H) ? r0 + RB : H, npx = (r1 - r0) * W, pbase = r0 * W;
        AIR_TR(0);
        // ---- every global load of this unit --------------------------------------------------------------------------
        const float *ob = rec_parts ? obs + (size_t)b * HW + pbase : where;     // (any valid address when rec is not wanted)
        const int ob_last = rec_parts ? npx - 1 : 0;
        float xo[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                          // unconditional loads from clamped addresses (no branches)
            const int p = tid + u * nt;
            xo[u] = ob[p < ob_last ? p : ob_last];
        }
        if (unit != vblock) __syncthreads();                   // grid-stride reuse of the carve
        if (vec4_glimpse) {                                    // (w % 4 == 0: a 16-byte group never straddles a glimpse row)
            const int nq = hw >> 2;
            for (int e = tid; e < T * nq; e += nt) {
                const int t = e / nq, q = e - t * nq;
                const float4 v = reinterpret_cast<const float4 *>(glimpse + ((size_t)t * B + b) * hw)[q];
                float *d = c.glm + (size_t)t * c.hwp + pad_index(4 * q, w, inv_w);
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        } else {
            for (int e = tid; e < T * hw; e += nt) {
                const int t = e / hw, q = e - t * hw;
                c.glm[(size_t)t * c.hwp + pad_index(q, w, inv_w)] = glimpse[((size_t)t * B + b) * hw + q];
            }
        }
        const int nrow = r1 - r0;
        for (int a = tid; a < T * (W + nrow); a += nt) {
            const int t = a / (W + nrow), r = a - t * (W + nrow);
            const float *wk = where + 4 * ((size_t)t * B + b);
            if (r < W) {
                const float sx = wk[0], tx = wk[1];
                c.xe[t * W + r] = axis_entry2(grid_coord(1.0f / sx, lin_m11(r, W, stepX), -tx / sx, cxs), w);
            } else {
                const float sy = wk[2], ty = wk[3];
                const int i = r - W;
                c.ye[t * RB + i] = axis_entry2(grid_coord(1.0f / sy, lin_m11(r0 + i, H, stepY), -ty / sy, cys), h);
            }
        }
        if (tid < T) c.pres[tid] = presence ? presence[(size_t)tid * B + b] : 1.0f;
        AIR_TR(1);
        __syncthreads();
        AIR_TR(2);
        float s[1] = {0.f};
        for (int p0 = tid; p0 < npx; p0 += 4 * nt) {
            float xn[4] = {0.f, 0.f, 0.f, 0.f};
            if (p0 + 4 * nt < npx) {                           // next chunk's observations (bands above 4 pixels per thread)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = p0 + (4 + u) * nt;
                    xn[u] = ob[p < ob_last ? p : ob_last];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = p0 + u * nt;
                if (p >= npx) break;
                const int Ib = div_small(p, W, inv_W), J = p - Ib * W;
                const size_t gp = (size_t)b * HW + pbase + p;
                float acc = canvas_in ? canvas_in[gp] : 0.f;
                for (int t = 0; t < T; ++t) {
                    const float2 ex = c.xe[t * W + J], ey = c.ye[t * RB + Ib];
                    const int fx = __float_as_int(ex.x), fy = __float_as_int(ey.x);
                    float v = 0.f;
                    if (fx != ST_INVALID && fy != ST_INVALID)
                        v = bilerp(load_taps_pad(c.glm + (size_t)t * c.hwp, pitch, fy, fx), ex.y, ey.y);
                    acc = acc + c.pres[t] * v;
                    if (canvas_steps) canvas_steps[((size_t)t * B + b) * HW + pbase + p] = acc;
                }
                if (final_canvas) final_canvas[gp] = acc;
                if (rec_parts) {
                    const float z = (xo[u] - mult * acc) / std;
                    s[0] += 0.5f * z * z + cst;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) xo[u] = xn[u];
        }
        AIR_TR(3);
        if (rec_parts) {
            block_sum<1>(s, c.scratch);
            if (tid == 0) rec_parts[(size_t)band * B + b] = s[0];
        }
        AIR_TR(4);
    }
    AIR_TR_FLUSH();
}
__global__ __launch_bounds__(1024) void st_write_fwd_kernel(WriteFwdArgs a) {
    extern __shared__ __align__(16) float smem[];
    st_write_fwd_body(a, smem, (int)blockIdx.x, (int)gridDim.x);
}

// Backward of the write for every (t, b): dglimpse, dwhere, optional dpresence.
// dcanvas either given per step ([T*B,H,W]) or formed on the fly from the reconstruction term:
//   dcanvas[b,p] = loss_scale * mult * (mult*final[b,p] - obs[b,p]) / std^2   (shared by all t)
// dglimpse is the transpose of a separable bilinear map, dG = Wy^T . g . Wx with two non-zeros per row of Wy / Wx,
// evaluated as two small LDS passes in a fixed order (no float atomics => bitwise reproducible):
//   T1[I,j] = sum_J g[I,J] * wx[J,j]   over the contiguous J-range that touches glimpse column j
//   dG[i,j] = sum_I wy[I,i] * T1[I,j]  over the contiguous I-range that touches glimpse row i
// Rounds 2-5 ran this with a PIXEL PASS for dwhere (st_write_bwd_kernel: one workgroup per unit; round 5's st_write_bwd_img_kernel:
// one per image; both in the history of round 6's first commits, same-box A/Bs in profiles/r06_canvas_gs_ab.txt); round 6's
// glimpse-space form below replaced them in every regime.
struct WriteBwdArgs {
    const float *glimpse, *where, *presence, *dcanvas, *final_canvas, *obs;
    float *dglimpse, *dwhere, *dpresence;
    int T, B, H, W, h, w;
    double stepX, stepY;
    float mult, std, loss_scale;
    int vec4_glimpse, vec4_canvas;
    int NS;                           // workgroups per unit: dglimpse rows are disjoint, dwhere is written as NS slabs [NS][T*B][4]
};
// the (up to) four contraction weights of source index j from canvas index lo on: weight of canvas index J for source index j is
// d_J if floor_J == j, 1 - d_J if floor_J + 1 == j (the transpose of the bilinear taps), 0 past the range
template <typename Acc>
__device__ __forceinline__ float4 touch_weights_t(const Acc &tab, int2 r, int j) {
    float wv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int J = r.x + u;
        const bool in = J <= r.y;
        const float2 e = tab(in ? J : (r.x <= r.y ? r.x : 0));
        const int f = __float_as_int(e.x);
        const float wgt = (f == j ? e.y : 0.f) + (f + 1 == j ? 1.f - e.y : 0.f);
        wv[u] = in ? wgt : 0.f;
    }
    return make_float4(wv[0], wv[1], wv[2], wv[3]);
}
__device__ __forceinline__ float4 touch_weights(const float2 *tab, int2 r, int j) { return touch_weights_t(TabAcc{tab}, r, j); }
// ---- glimpse-space backward (round 6) ---------------------------------------------------------------------------------------
// Both forms above pay a PIXEL PASS for dwhere: every canvas pixel of a unit's footprint evaluates its four taps and the two
// coordinate derivatives (~40 vector instructions per pixel; 10 000 pixels per unit once a glimpse covers a 100x100 canvas: the
// 18-39 us, state-dependent position of the configs[3] step and 0.086 of the HBM roofline out of cache, profiles/r05_*).  The
// bilinear map is separable, so the same five sums can be taken where the column contraction already is:
//   p  [I,j] = sum_J dc[I,J] wx [J,j]            (T1 of the forms above; wx = the transposed x taps)
//   px [I,j] = sum_J dc[I,J] dwx[J,j]            dwx[J,j] = +1 if floor_x(J)+1 == j, -1 if floor_x(J) == j   (d wx / d x_g)
//   pxX[I,j] = sum_J dc[I,J] X_J dwx[J,j]
//   R  [I,j] = dy_I G[fy_I, j] + (1-dy_I) G[fy_I+1, j]        (the glimpse resampled along y only; zero outside)
//   D  [I,j] = G[fy_I+1, j] - G[fy_I, j]                       (its derivative along y)
//   d/d(ax) = cxs sum_{I,j} pxX R,  d/d(bx) = cxs sum px R,  d/d(ay) = cys sum Y_I p D,  d/d(by) = cys sum p D,  d/dpresence = sum p R
// (sums over the valid canvas rows I and the w glimpse columns j; x_g = cxs (ax X_J + bx + 1), y_g likewise) -- fh*w elements of
// ~35 instructions instead of fh*fw pixels of ~40, no taps, no `go` image: dcanvas stays as it is and the unit's presence multiplies the
// finished dglimpse element and the four sums (both are linear in it).  dglimpse = presence Wy^T p as before.
// One body, three uses: unit-major (one workgroup per (t, b), latency regime; optional NS workgroups per unit), image-major (one per
// image: dcanvas staged once, the T units' column contractions in one phase, their row contractions in the next: 4 barriers per
// image) and the recompute form of the fused forward+backward launch (the canvas re-formed on the unit's footprint with the
// forward's own calls, bit-identical, then the same two phases).  Ranges, weights, tables and the dwhere chain are the ones of the
// forms above, every reduction has a fixed order; unit-major and recompute form are the same arithmetic on the same bits.
struct CarveGs {
    float *g, *t1, *src, *X, *Y, *pres, *wh, *scratch;
    float2 *xe, *ye;
    int2 *jr, *ir, *rows, *cols;
    float4 *wx4, *dx4, *xx4, *wy4;
    int hwp, t1s;
};
// TU = units a workgroup runs side by side (1 | T), TS = steps whose glimpse copy and axis tables it holds (1 | T)
__device__ __forceinline__ CarveGs carve_gs(float *smem, int H, int W, int h, int w, int TU, int TS, int nw) {
    CarveGs c;
    float *p = smem;
    c.hwp = pad_count(h, w);
    c.t1s = (H * w + 3) & ~3;
    c.g = p; p += (H * W + 3) & ~3;
    c.wx4 = reinterpret_cast<float4 *>(p); p += 4 * TU * w;
    c.dx4 = reinterpret_cast<float4 *>(p); p += 4 * TU * w;
    c.xx4 = reinterpret_cast<float4 *>(p); p += 4 * TU * w;
    c.wy4 = reinterpret_cast<float4 *>(p); p += 4 * TU * h;
    c.t1 = p; p += (size_t)TU * c.t1s;
    c.src = p; p += (size_t)TS * c.hwp;
    c.xe = reinterpret_cast<float2 *>(p); p += 2 * TS * W;
    c.ye = reinterpret_cast<float2 *>(p); p += 2 * TS * H;
    c.jr = reinterpret_cast<int2 *>(p); p += 2 * TU * w;
    c.ir = reinterpret_cast<int2 *>(p); p += 2 * TU * h;
    c.rows = reinterpret_cast<int2 *>(p); p += 2 * ((TU + 1) & ~1);
    c.cols = reinterpret_cast<int2 *>(p); p += 2 * ((TU + 1) & ~1);
    c.X = p; p += W;
    c.Y = p; p += H;
    c.pres = p; p += 2 * ((TS + 3) & ~3);                // (two buffers: image-major keeps the NEXT image's presences / `where` rows
    c.wh = p; p += 8 * TS;                               //  beside the current one's, see the image loop)
    c.scratch = p;                                       // [nw][TU][8]
    (void)nw;
    return c;
}
static inline size_t carve_gs_bytes(int H, int W, int h, int w, int TU, int TS, int nw) {
    return sizeof(float) * (size_t)(((H * W + 3) & ~3) + 12 * TU * w + 4 * TU * h + (size_t)TU * ((H * w + 3) & ~3) +
                                    (size_t)TS * pad_count_host(h, w) + 2 * TS * (W + H) + 2 * TU * (w + h) + 4 * ((TU + 1) & ~1) +
                                    W + H + 2 * ((TS + 3) & ~3) + 8 * TS + (size_t)nw * TU * 8 + 16);
}
// touch_weights for the three column sums: the (up to) four canvas indices from r.x on -> wx, dwx, X dwx
template <typename Acc>
__device__ __forceinline__ void touch_weights3(const Acc &tab, const float *Xt, int n, double step, int2 r, int j, float4 *w4, float4 *d4, float4 *x4) {
    float wv[4], dv[4], xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int J = r.x + u;
        const bool in = J <= r.y;
        const int Jc = in ? J : (r.x <= r.y ? r.x : 0);
        const float2 e = tab(Jc);
        const int f = __float_as_int(e.x);
        const float wgt = (f == j ? e.y : 0.f) + (f + 1 == j ? 1.f - e.y : 0.f);
        const float dw = (f + 1 == j ? 1.f : 0.f) - (f == j ? 1.f : 0.f);
        wv[u] = in ? wgt : 0.f;
        dv[u] = in ? dw : 0.f;
        xv[u] = in ? (Xt ? Xt[Jc] : lin_m11(Jc, n, step)) * dw : 0.f;     // (Xt: the linspace table when it is already visible)
    }
    *w4 = make_float4(wv[0], wv[1], wv[2], wv[3]);
    *d4 = make_float4(dv[0], dv[1], dv[2], dv[3]);
    *x4 = make_float4(xv[0], xv[1], xv[2], xv[3]);
}
// Barriers per item: [operands, axis tables, ranges + weights] (1) [RC only: canvas on the footprint (2)] column contraction + sums (3)
// row contraction.  The ranges and weights are formed from the transform itself (FlyAcc: the table build's own calls) by the LAST
// threads of the workgroup while the first ones build the tables, so they cost no phase of their own.
// Spans of one unit from its axis tables: {first valid canvas row, number of valid rows (-1: absent step), first touched glimpse
// column, number of touched columns}.  An ABSENT step (presence exactly 0) has dglimpse = 0 and zero dwhere sums: both phases are
// skipped (the dwhere chain still runs on the zero sums, so a degenerate scale gives the same NaN); only a caller that wants
// dpresence needs them.  The touched glimpse columns are the floors of the first and last valid canvas column and their right
// neighbours (the map is monotone; a superset is fine, an untouched column in between has an empty range): only they are walked --
// at scales beyond 1 a third of the columns carries all the work and a lane per column would leave the wave two thirds idle.
template <bool SPLIT>
__device__ __forceinline__ int4 gs_unit_spans(const float2 *xe, const float2 *ye, int W, int H, int w, int i0, int i1, bool absent) {
    const int2 vy = !SPLIT ? valid_span(ye, H) : floor_span(ye, H, i0 - 1, i1 - 1);
    const int2 vx = valid_span(xe, W);
    const int fh = absent ? -1 : (vy.y - vy.x + 1 > 0 ? vy.y - vy.x + 1 : 0);
    int ja = 0, nj = 0;
    if (vx.x <= vx.y) {
        const int fa = __float_as_int(xe[vx.x].x), fb = __float_as_int(xe[vx.y].x);
        const int lo = fa < fb ? fa : fb, hi = (fa < fb ? fb : fa) + 1;
        ja = lo < 0 ? 0 : lo;
        nj = (hi > w - 1 ? w - 1 : hi) - ja + 1;
        if (nj < 0) nj = 0;
    }
    return make_int4(vy.x, fh, ja, nj);
}
template <bool RC, bool IM, bool SPLIT>
__device__ __forceinline__ void st_write_bwd_gs_body(const WriteBwdArgs &a, const NvilArgs &nv, float *smem, const int vblock, const int vgrid) {
    const float *__restrict__ glimpse = a.glimpse, *__restrict__ where = a.where, *__restrict__ presence = a.presence;
    const float *__restrict__ dcanvas = a.dcanvas, *__restrict__ final_canvas = a.final_canvas, *__restrict__ obs = a.obs;
    float *__restrict__ dglimpse = a.dglimpse, *__restrict__ dwhere = a.dwhere, *__restrict__ dpresence = a.dpresence;
    const int T = a.T, B = a.B, H = a.H, W = a.W, h = a.h, w = a.w;
    const int grid_st = nv.imp ? vgrid - 1 : vgrid;
    const int bid0 = nv.imp ? vblock - 1 : vblock;
    AIR_TR_INIT();
    if (bid0 < 0) { nvil_body(nv); AIR_TR_FLUSH(); return; }             // (first workgroup of the role: the long float64 chain starts at once)
    const int HW = H * W, hw = h * w, tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wid = tid >> 6, nw = nt >> 6;
    const int TU = IM ? T : 1, TS = (RC || IM) ? T : 1;
    CarveGs c = carve_gs(smem, H, W, h, w, TU, TS, nw);
    const float cxs = (float)((w - 1) / 2.0), cys = (float)((h - 1) / 2.0);
    const float inv_cxs = 1.0f / cxs, inv_cys = 1.0f / cys;
    const float coef = a.loss_scale * a.mult / (a.std * a.std), mult = a.mult;
    const int pitch = w + 2, nQ = HW >> 2, nq = hw >> 2;
    const float inv_w = 1.0f / (float)w;
    const bool v4c = a.vec4_canvas != 0, v4g = a.vec4_glimpse != 0;
    // once per workgroup: the linspace tables (shapes only) and the zero borders of the glimpse copies (visible after barrier (1))
    for (int e = tid; e < W + H; e += nt) {
        if (e < W) c.X[e] = lin_m11(e, W, a.stepX); else c.Y[e - W] = lin_m11(e - W, H, a.stepY);
    }
    for (int e = tid; e < TS * pad_border(h, w); e += nt) {
        const int tt = e / pad_border(h, w);
        c.src[(size_t)tt * c.hwp + pad_border_index(e - tt * pad_border(h, w), h, w)] = 0.f;
    }
    const int NS = SPLIT ? a.NS : 1;
    const int n_items = IM ? B : T * B * NS;
    // (measured and rejected, profiles/r06_canvas_gs_ab.txt: image-major with the NEXT image's canvas operands requested during the row
    //  contraction and held in registers over the barrier -- 28 more live registers at 1024 threads spill, 4819 against 4614 us at
    //  65536 images of 100x100)
    // Image-major: the `where` rows and presences of an image (5 T floats) are requested one image AHEAD and parked in LDS (two buffers), so
    // that the table build at the top of an image does not start with a global round trip (3.0 us from image start to the tables, traced)
    const int prs = (TS + 3) & ~3;
    if (IM && bid0 < n_items) {
        if (tid < TS * 4) c.wh[tid] = where[4 * ((size_t)(tid >> 2) * B + bid0) + (tid & 3)];
        else if (tid < TS * 5) c.pres[tid - TS * 4] = presence ? presence[(size_t)(tid - TS * 4) * B + bid0] : 1.0f;
    }
    int par = 0;
    for (int it = bid0; it < n_items; it += grid_st, par ^= (IM ? 1 : 0)) {
        // unit-major: item = (unit k = t_own*B + b, split sp); image-major: item = image b, local unit lu = step
        const int k_um = IM ? 0 : it / NS, sp = IM ? 0 : it - k_um * NS;
        const int b = IM ? it : k_um % B, t_own = IM ? 0 : k_um / B;
        const int i0 = (int)(((long)h * sp) / NS), i1 = (int)(((long)h * (sp + 1)) / NS);     // this workgroup's dglimpse rows
        const int ts_own = (RC && !IM) ? t_own : 0;          // table / copy index of local unit 0 (image-major: lu itself)
        AIR_TR(0);
        if (it != bid0 || IM) __syncthreads();               // (0) the previous item's readers are done with the carve
        float *presb = c.pres + par * prs;
        const float *whb = c.wh + par * 4 * TS;
        float wnx = 0.f;
        if (IM && it + grid_st < n_items) {
            const int nb = it + grid_st;
            if (tid < TS * 4) wnx = where[4 * ((size_t)(tid >> 2) * B + nb) + (tid & 3)];
            else if (tid < TS * 5) wnx = presence ? presence[(size_t)(tid - TS * 4) * B + nb] : 1.0f;
        }
        // ---- operands -------------------------------------------------------------------------------------------------------
        // first 16-byte group of the canvas operands requested before anything else waits (latency regime: the tables below only
        // need `where`); RC: the observation (c.g holds it until the canvas pass replaces the footprint)
        const float *pa = RC ? obs + (size_t)b * HW : (dcanvas ? dcanvas + (size_t)k_um * HW : final_canvas + (size_t)b * HW);
        const float *pb = (RC || dcanvas) ? pa : obs + (size_t)b * HW;
        const bool form = !RC && !dcanvas;                    // g = coef * (mult * final - obs)
        float4 qa = make_float4(0.f, 0.f, 0.f, 0.f), qb = qa, qa1 = qa, qb1 = qa, gq0 = qa;
        if (v4c) {
            const int q = tid < nQ ? tid : nQ - 1, q1 = tid + nt < nQ ? tid + nt : nQ - 1;
            qa = reinterpret_cast<const float4 *>(pa)[q];
            if (form) qb = reinterpret_cast<const float4 *>(pb)[q];
            if (!IM) {
                qa1 = reinterpret_cast<const float4 *>(pa)[q1];
                if (form) qb1 = reinterpret_cast<const float4 *>(pb)[q1];
            }
        }
        if (v4g && !IM) {
            const int q = tid < TS * nq ? tid : TS * nq - 1, tt = q / nq;
            gq0 = reinterpret_cast<const float4 *>(glimpse + ((size_t)(RC ? tt : t_own) * B + b) * hw)[q - tt * nq];
        }
        // exact contraction ranges + weights of every glimpse column / row of the TU units: EIGHT lanes per column / row, one candidate
        // canvas index each (src_range leaves at most eight at ordinary scales), entry evaluated from the transform itself (FlyAcc: the
        // table build's own calls, the same bits), hits gathered with one ballot, the four weights fetched from the lanes that hold
        // their entries -- ~50 instructions per thread beside the table build instead of a serial phase of its own
        if (!IM) {
            const int n_it = TU * (w + h);
            for (int v0 = 0; v0 < n_it * 8; v0 += nt) {       // (uniform trip count: the ballot / shuffles need every lane)
                const int v = v0 + tid, e = v >> 3, u = v & 7;
                const bool live = e < n_it;
                const int ec = live ? e : 0;
                const int lu = ec / (w + h), r = ec - lu * (w + h);
                const bool isx = r < w;
                const int jj = isx ? r : r - w;
                const float *wk = where + 4 * ((size_t)(IM ? lu : t_own) * B + b) + (isx ? 0 : 2);
                const float s_ = wk[0], t_ = wk[1];
                const int n_c = isx ? W : H;
                const FlyAcc fa = {1.0f / s_, -t_ / s_, isx ? cxs : cys, isx ? w : h, n_c, isx ? a.stepX : a.stepY};
                int lo, hi;
                src_range(s_, fa.b, isx ? inv_cxs : inv_cys, (float)(jj - 1), (float)(jj + 1), n_c, &lo, &hi);
                const bool narrow = hi - lo < 8;
                const int J = lo + u;
                const float2 en = fa(J <= hi ? J : lo);
                const int f = __float_as_int(en.x);
                const bool hit = narrow && J <= hi && f != ST_INVALID && (f == jj || f + 1 == jj);
                const unsigned mask = (unsigned)(__ballot(hit) >> (lane & 56)) & 0xffu;
                int2 rg = make_int2(1, 0);
                if (mask) rg = make_int2(lo + __ffs((int)mask) - 1, lo + 31 - __clz((int)mask));
                // weight slot u (< 4) of this column / row: canvas index rg.x + u, whose entry lane (rg.x - lo + u) of the group holds
                const int srcl = (lane & 56) + ((rg.x - lo + u) & 7);
                const float ef = __shfl(en.x, srcl, 64), ed = __shfl(en.y, srcl, 64);
                if (live && narrow && u < 4) {
                    const int Jw = rg.x + u;
                    const bool in = Jw <= rg.y;
                    const int fw_ = __float_as_int(ef);
                    const float wgt = in ? (fw_ == jj ? ed : 0.f) + (fw_ + 1 == jj ? 1.f - ed : 0.f) : 0.f;
                    if (isx) {
                        const float dw = in ? (fw_ + 1 == jj ? 1.f : 0.f) - (fw_ == jj ? 1.f : 0.f) : 0.f;
                        const float xw = in ? lin_m11(Jw, W, a.stepX) * dw : 0.f;
                        reinterpret_cast<float *>(&c.wx4[lu * w + jj])[u] = wgt;
                        reinterpret_cast<float *>(&c.dx4[lu * w + jj])[u] = dw;
                        reinterpret_cast<float *>(&c.xx4[lu * w + jj])[u] = xw;
                        if (u == 0) c.jr[lu * w + jj] = rg;
                    } else {
                        reinterpret_cast<float *>(&c.wy4[lu * h + jj])[u] = wgt;
                        if (u == 0) c.ir[lu * h + jj] = rg;
                    }
                }
                if (live && !narrow && u == 0) {               // degenerate scales: candidate interval of the whole axis, scanned by one lane
                    const int2 rs = touch_range_t(fa, fa.b, s_, isx ? inv_cxs : inv_cys, jj, n_c);
                    if (isx) {
                        c.jr[lu * w + jj] = rs;
                        touch_weights3(fa, nullptr, W, a.stepX, rs, jj, &c.wx4[lu * w + jj], &c.dx4[lu * w + jj], &c.xx4[lu * w + jj]);
                    } else {
                        c.ir[lu * h + jj] = rs;
                        c.wy4[lu * h + jj] = touch_weights_t(fa, rs, jj);
                    }
                }
            }
        }
        AIR_TR(1);
        // axis tables of the TS steps, by the first threads (column entries, then row entries from the next wave on: no wave holds both)
        const int n_xe = TS * W, n_xe_pad = (n_xe + 63) & ~63;
        for (int a0 = tid; a0 < n_xe_pad + TS * H; a0 += nt) {
            if (a0 < n_xe) {
                const int tt = a0 / W, r = a0 - tt * W;
                const float *wk = IM ? whb + 4 * tt : where + 4 * ((size_t)(RC ? tt : t_own) * B + b);
                const float s_ = wk[0], t_ = wk[1];
                c.xe[tt * W + r] = axis_entry2(grid_coord(1.0f / s_, lin_m11(r, W, a.stepX), -t_ / s_, cxs), w);
            } else if (a0 >= n_xe_pad) {
                const int a1 = a0 - n_xe_pad, tt = a1 / H, r = a1 - tt * H;
                const float *wk = IM ? whb + 4 * tt : where + 4 * ((size_t)(RC ? tt : t_own) * B + b);
                const float s_ = wk[2], t_ = wk[3];
                c.ye[tt * H + r] = axis_entry2(grid_coord(1.0f / s_, lin_m11(r, H, a.stepY), -t_ / s_, cys), h);
            }
        }
        if (!IM && tid < TS) c.pres[tid] = presence ? presence[(size_t)(RC ? tid : t_own) * B + b] : 1.0f;
        // glimpse copies (bordered)
        if (v4g) {
            for (int q = tid; q < TS * nq; q += nt) {
                const int tt = q / nq;
                const float4 v = (!IM && q == tid) ? gq0
                                                   : reinterpret_cast<const float4 *>(glimpse + ((size_t)((RC || IM) ? tt : t_own) * B + b) * hw)[q - tt * nq];
                float *d = c.src + (size_t)tt * c.hwp + pad_index(4 * (q - tt * nq), w, inv_w);
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        } else {
            for (int q = tid; q < TS * hw; q += nt) {
                const int tt = q / hw;
                c.src[(size_t)tt * c.hwp + pad_index(q - tt * hw, w, inv_w)] = glimpse[((size_t)((RC || IM) ? tt : t_own) * B + b) * hw + (q - tt * hw)];
            }
        }
        // canvas operand -> c.g
        if (v4c) {
            if (tid < nQ) {
                float4 gv = qa;
                if (form) { gv.x = dcanvas_of(coef, mult, qa.x, qb.x); gv.y = dcanvas_of(coef, mult, qa.y, qb.y);
                            gv.z = dcanvas_of(coef, mult, qa.z, qb.z); gv.w = dcanvas_of(coef, mult, qa.w, qb.w); }
                reinterpret_cast<float4 *>(c.g)[tid] = gv;
            }
            if (!IM && tid + nt < nQ) {
                float4 gv = qa1;
                if (form) { gv.x = dcanvas_of(coef, mult, qa1.x, qb1.x); gv.y = dcanvas_of(coef, mult, qa1.y, qb1.y);
                            gv.z = dcanvas_of(coef, mult, qa1.z, qb1.z); gv.w = dcanvas_of(coef, mult, qa1.w, qb1.w); }
                reinterpret_cast<float4 *>(c.g)[tid + nt] = gv;
            }
#pragma unroll 4
            for (int q = tid + (IM ? 1 : 2) * nt; q < nQ; q += nt) {
                const float4 a4 = reinterpret_cast<const float4 *>(pa)[q];
                float4 gv = a4;
                if (form) {
                    const float4 b4 = reinterpret_cast<const float4 *>(pb)[q];
                    gv.x = dcanvas_of(coef, mult, a4.x, b4.x); gv.y = dcanvas_of(coef, mult, a4.y, b4.y);
                    gv.z = dcanvas_of(coef, mult, a4.z, b4.z); gv.w = dcanvas_of(coef, mult, a4.w, b4.w);
                }
                reinterpret_cast<float4 *>(c.g)[q] = gv;
            }
        } else {
            for (int p = tid; p < HW; p += nt) c.g[p] = form ? dcanvas_of(coef, mult, pa[p], pb[p]) : pa[p];
        }
        AIR_TR(2);
        __syncthreads();                                       // (1)
        AIR_TR(3);
        if (IM) {
            // image-major (throughput regime): ranges + weights of the T units from the LDS tables, one thread per column / row -- a
            // fifth of the instructions of the eight-lane form above, for one more barrier per image
            // (column items first, row items from the next wave on: a wave that held both kinds ran the two paths one after the other)
            const int n_x = TU * w, n_x_pad = (n_x + 63) & ~63;
            for (int e = tid; e < n_x_pad + TU * h; e += nt) {
                if (e < n_x) {
                    const int lu = e / w, r = e - lu * w;
                    const float *wk = whb + 4 * lu;
                    const float s_ = wk[0], t_ = wk[1];
                    const TabAcc ta = {c.xe + lu * W};
                    const int2 rg = touch_range_t(ta, -t_ / s_, s_, inv_cxs, r, W);
                    c.jr[lu * w + r] = rg;
                    touch_weights3(ta, c.X, W, a.stepX, rg, r, &c.wx4[lu * w + r], &c.dx4[lu * w + r], &c.xx4[lu * w + r]);
                } else if (e >= n_x_pad) {
                    const int e2 = e - n_x_pad, lu = e2 / h, i = e2 - lu * h;
                    const float *wk = whb + 4 * lu;
                    const float s_ = wk[2], t_ = wk[3];
                    const TabAcc ta = {c.ye + lu * H};
                    const int2 rg = touch_range_t(ta, -t_ / s_, s_, inv_cys, i, H);
                    c.ir[lu * h + i] = rg;
                    c.wy4[lu * h + i] = touch_weights_t(ta, rg, i);
                }
            }
            // valid canvas rows and touched glimpse columns of each unit, one wave per unit (see gs_unit_spans)
            for (int lu = nw - 1 - wid; lu < TU; lu += nw) {   // (by the LAST waves: the first ones hold the range items)
                const int4 sp4 = gs_unit_spans<SPLIT>(c.xe + lu * W, c.ye + lu * H, W, H, w, i0, i1, presb[lu] == 0.f && !dpresence);
                if (lane == 0) { c.rows[lu] = make_int2(sp4.x, sp4.y); c.cols[lu] = make_int2(sp4.z, sp4.w); }
            }
            AIR_TR(10);
            __syncthreads();                                   // (1b)
            AIR_TR(11);
        }
        if (RC) {
            // the canvas on this unit's footprint, accumulated as the forward does -- ((0 + p0*v0) + p1*v1) + ... over ALL steps with
            // the forward's table entries and taps: bit-identical values -- then dcanvas from it and the observation c.g holds
            const float2 *xo = c.xe + ts_own * W, *yo = c.ye + ts_own * H;
            const int2 vx = valid_span(xo, W), vy = !SPLIT ? valid_span(yo, H) : floor_span(yo, H, i0 - 1, i1 - 1);
            const int J0 = vx.x, I0 = vy.x, fw = vx.y - vx.x + 1, fh = vy.y - vy.x + 1;
            const int npx = (fw > 0 && fh > 0) ? fw * fh : 0;
            const float inv_fw = 1.0f / (float)(fw > 0 ? fw : 1);
            for (int idx = tid; idx < npx; idx += nt) {
                const int Ir = div_small(idx, fw, inv_fw), I = I0 + Ir, J = J0 + (idx - Ir * fw), p = I * W + J;
                float cv = 0.f;
                for (int tt = 0; tt < T; ++tt) {
                    const float2 ext = c.xe[tt * W + J], eyt = c.ye[tt * H + I];
                    const int fxt = __float_as_int(ext.x), fyt = __float_as_int(eyt.x);
                    float vt = 0.f;
                    if (fxt != ST_INVALID && fyt != ST_INVALID)
                        vt = bilerp(load_taps_pad(c.src + (size_t)tt * c.hwp, pitch, fyt, fxt), ext.y, eyt.y);
                    cv = acc_step(cv, presb[tt], vt);
                }
                c.g[p] = dcanvas_of(coef, mult, cv, c.g[p]);
            }
            AIR_TR(4);
            __syncthreads();                                   // (2)
        }
        // ---- phase 1: p, px, pxX on the unit's valid rows; the five sums; T1 = p
        for (int lu = 0; lu < TU; ++lu) {
            const int tt = IM ? lu : ts_own;
            const float2 *xe = c.xe + tt * W, *ye = c.ye + tt * H;
            // valid canvas rows / touched glimpse columns of the unit: unit-major every wave finds them itself (a few ballots over the
            // tables), image-major they were left in LDS by one wave per unit (sixteen waves x T units of ballots cost half a phase)
            int4 sp4;
            if (IM) { const int2 rw_ = c.rows[lu], cw_ = c.cols[lu]; sp4 = make_int4(rw_.x, rw_.y, cw_.x, cw_.y); }
            else {
                sp4 = gs_unit_spans<SPLIT>(xe, ye, W, H, w, i0, i1, presb[tt] == 0.f && !dpresence);
                if (tid == 0) { c.rows[lu] = make_int2(sp4.x, sp4.y); c.cols[lu] = make_int2(sp4.z, sp4.w); }
            }
            const int I0 = sp4.x, fh = sp4.y, ja = sp4.z, nj = sp4.w;
            const float inv_nj = 1.0f / (float)(nj > 0 ? nj : 1);
            const float *src = c.src + (size_t)tt * c.hwp;
            float *t1 = c.t1 + (size_t)lu * c.t1s;
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // (measured and rejected, profiles/r06_canvas_gs_ab.txt: a thread keeping ONE glimpse column per unit -- range and weights read
            //  once, rows walked with a stride of nt / nj -- issues fewer instructions per element but pays its set-up in every thread and
            //  unit: 4802 against 4563 us at 65536 images of 100x100)
            for (int e = tid; e < fh * nj; e += nt) {
                const int Ir = div_small(e, nj, inv_nj), I = I0 + Ir, j = ja + (e - Ir * nj);
                const int2 r = c.jr[lu * w + j];
                const float2 ey = ye[I];
                const int fy = __float_as_int(ey.x);         // valid by construction of the row span
                const float *gq = src + (fy + 1) * pitch + (j + 1);
                const float G0 = gq[0], G1 = gq[pitch];
                float p = 0.f, px = 0.f, pxX = 0.f;
                if (r.x <= r.y) {
                    const float4 wv = c.wx4[lu * w + j], dv = c.dx4[lu * w + j], xv = c.xx4[lu * w + j];
                    const float *grow = c.g + I * W;
                    const int Jb = r.x;
                    const int j1 = Jb + 1 <= r.y ? Jb + 1 : Jb, j2 = Jb + 2 <= r.y ? Jb + 2 : Jb, j3 = Jb + 3 <= r.y ? Jb + 3 : Jb;
                    const float g0 = grow[Jb], g1 = grow[j1], g2 = grow[j2], g3 = grow[j3];
                    p = g0 * wv.x; px = g0 * dv.x; pxX = g0 * xv.x;
                    p = __builtin_fmaf(g1, wv.y, p); px = __builtin_fmaf(g1, dv.y, px); pxX = __builtin_fmaf(g1, xv.y, pxX);
                    p = __builtin_fmaf(g2, wv.z, p); px = __builtin_fmaf(g2, dv.z, px); pxX = __builtin_fmaf(g2, xv.z, pxX);
                    p = __builtin_fmaf(g3, wv.w, p); px = __builtin_fmaf(g3, dv.w, px); pxX = __builtin_fmaf(g3, xv.w, pxX);
                    // wide ranges (scales towards 1 and beyond): the general form, four entries per round trip while four remain
                    int J = r.x + 4;
                    for (; J + 3 <= r.y; J += 4) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float2 ex = xe[J + u];
                            const float gv = grow[J + u], Xv = c.X[J + u];
                            const int fx = __float_as_int(ex.x);
                            const float wgt = (fx == j ? ex.y : 0.f) + (fx + 1 == j ? 1.f - ex.y : 0.f);
                            const float dw = (fx + 1 == j ? 1.f : 0.f) - (fx == j ? 1.f : 0.f);
                            p = __builtin_fmaf(gv, wgt, p); px = __builtin_fmaf(gv, dw, px); pxX = __builtin_fmaf(gv, Xv * dw, pxX);
                        }
                    }
                    for (; J <= r.y; ++J) {
                        const float2 ex = xe[J];
                        const float gv = grow[J], Xv = c.X[J];
                        const int fx = __float_as_int(ex.x);
                        const float wgt = (fx == j ? ex.y : 0.f) + (fx + 1 == j ? 1.f - ex.y : 0.f);
                        const float dw = (fx + 1 == j ? 1.f : 0.f) - (fx == j ? 1.f : 0.f);
                        p = __builtin_fmaf(gv, wgt, p); px = __builtin_fmaf(gv, dw, px); pxX = __builtin_fmaf(gv, Xv * dw, pxX);
                    }
                }
                t1[Ir * w + j] = p;
                const float R = __builtin_fmaf(ey.y, G0, (1.f - ey.y) * G1), D = G1 - G0;
                const int fyc = fy < 0 ? 0 : (fy > h - 1 ? h - 1 : fy);
                if (!SPLIT || (fyc >= i0 && fyc < i1)) {       // (SPLIT: the row's owner among the unit's NS workgroups)
                    const float pD = p * D;
                    acc[0] = __builtin_fmaf(pxX, R, acc[0]); acc[1] = __builtin_fmaf(px, R, acc[1]);
                    acc[2] = __builtin_fmaf(c.Y[I], pD, acc[2]); acc[3] += pD;
                    acc[4] = __builtin_fmaf(p, R, acc[4]);
                }
            }
            const float rsum = wave_reduce8(acc);
            if ((lane & 7) == 0) c.scratch[(wid * TU + lu) * 8 + wave_reduce8_slot()] = rsum;
        }
        AIR_TR(5); AIR_TRT(nt - 64, 9);
        __syncthreads();                                       // (3)
        AIR_TR(6);
        // ---- phase 2: dG[lu][i, j] = presence * sum_I wy[I, i] * T1[lu][I, j], this workgroup's rows
        const int nrow = i1 - i0;
        const float inv_rw = 1.0f / (float)(nrow * w > 0 ? nrow * w : 1);
        for (int e0 = tid; e0 < TU * nrow * w; e0 += nt) {
            const int lu = IM ? div_small(e0, nrow * w, inv_rw) : 0;
            const int e = i0 * w + (e0 - lu * nrow * w);
            const int i = div_small(e, w, inv_w), j = e - i * w;
            const int tt = IM ? lu : ts_own;
            float *dg = dglimpse + ((size_t)(IM ? lu : t_own) * B + b) * hw;
            const int2 rw = c.rows[lu], cw = c.cols[lu];
            float s = 0.f;
            if (rw.y > 0 && j >= cw.x && j < cw.x + cw.y) {
                const int2 r = c.ir[lu * h + i];
                if (r.x <= r.y) {                              // (rows outside the span were never written: weight 0 is not enough)
                    const float4 wv = c.wy4[lu * h + i];
                    const float *t1 = c.t1 + (size_t)lu * c.t1s + j - rw.x * w;
                    const int r1 = r.x + 1 <= r.y ? r.x + 1 : r.x, r2 = r.x + 2 <= r.y ? r.x + 2 : r.x, r3 = r.x + 3 <= r.y ? r.x + 3 : r.x;
                    s = t1[r.x * w] * wv.x;
                    s = __builtin_fmaf(t1[r1 * w], wv.y, s);
                    s = __builtin_fmaf(t1[r2 * w], wv.z, s);
                    s = __builtin_fmaf(t1[r3 * w], wv.w, s);
                    const float2 *ye = c.ye + tt * H;
                    int I = r.x + 4;
                    for (; I + 3 <= r.y; I += 4) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float2 ey = ye[I + u];
                            const int fy = __float_as_int(ey.x);
                            s = __builtin_fmaf(t1[(I + u) * w], (fy == i ? ey.y : 0.f) + (fy + 1 == i ? 1.f - ey.y : 0.f), s);
                        }
                    }
                    for (; I <= r.y; ++I) {
                        const float2 ey = ye[I];
                        const int fy = __float_as_int(ey.x);
                        s = __builtin_fmaf(t1[I * w], (fy == i ? ey.y : 0.f) + (fy + 1 == i ? 1.f - ey.y : 0.f), s);
                    }
                }
                s *= presb[tt];
            }
            dg[e] = s;
        }
        AIR_TR(7);
        // ---- dwhere / dpresence: the per-wave partials (visible since barrier 3), fixed order, one wave per unit (from the last
        //      wave down: it has the least contraction work)
        for (int lu = nw - 1 - wid; lu < TU; lu += nw) {
            float part[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) part[q] = (lane < nw && q < 5) ? c.scratch[(lane * TU + lu) * 8 + q] : 0.f;
            const float tot = wave_reduce8(part);
            const float s0 = __shfl(tot, 0, 64), s1 = __shfl(tot, 8, 64), s2 = __shfl(tot, 16, 64), s3 = __shfl(tot, 24, 64),
                        r4 = __shfl(tot, 32, 64);
            if (lane == 0) {
                const size_t k = (size_t)(IM ? lu : t_own) * B + b;
                const float pres = presb[IM ? lu : ts_own];
                const float r0 = (pres * s0) * cxs, r1 = (pres * s1) * cxs, r2 = (pres * s2) * cys, r3 = (pres * s3) * cys;
                const float *wr = IM ? whb + 4 * lu : where + 4 * k;
                const float sx = wr[0], tx = wr[1], sy = wr[2], ty = wr[3];
                const float ax = 1.0f / sx, bx = -tx / sx, ay = 1.0f / sy, by = -ty / sy;
                // chain through a = 1/s, b = (-t)/s as automatic differentiation evaluates the two divisions:
                // degenerate scales give NaN / inf / 0 exactly where the reference's gradient does
                float *d = dwhere + 4 * ((size_t)sp * T * B + k);
                d[0] = -(r0 * (ax / sx)) - r1 * (bx / sx);
                d[1] = -(r1 / sx);
                d[2] = -(r2 * (ay / sy)) - r3 * (by / sy);
                d[3] = -(r3 / sy);
                if (dpresence) dpresence[(size_t)sp * T * B + k] = r4;
            }
        }
        AIR_TR(8);
        if (IM) {                                            // the next image's rows into the OTHER buffer (nobody reads it during this image)
            if (tid < TS * 4) c.wh[(par ^ 1) * 4 * TS + tid] = wnx;
            else if (tid < TS * 5) c.pres[(par ^ 1) * prs + tid - TS * 4] = wnx;
        }
    }
    AIR_TR_FLUSH();
}
template <bool RC, bool IM>
__global__ __launch_bounds__(1024) void st_write_bwd_gs_kernel(WriteBwdArgs a, NvilArgs nv) {
    extern __shared__ __align__(16) float smem[];
    st_write_bwd_gs_body<RC, IM, false>(a, nv, smem, (int)blockIdx.x, (int)gridDim.x);
}

// Canvas forward and backward of a train step in ONE launch (latency regime).  The recompute form of the backward reads nothing
// the forward writes, so the two are independent roles of one grid: workgroups [0, n_fwd) run st_write_fwd_body (image x row
// band: per-step canvases, final canvas, reconstruction shares), the rest st_write_bwd_gs_body<true, ...> (one per glimpse).  One
// dependent launch less on the step's chain; NVIL -- which needs the forward's reconstruction shares -- rides on a later launch
// (air_gauss_sample_bwd_nvil).
// ============================================================================================================
// host side
// ============================================================================================================
static inline double lin_step(int n) { return n > 1 ? 2.0 / (double)(n - 1) : 0.0; }
static inline int cv_grid(long items, int cap) { return (int)(items < cap ? items : cap); }
// (developer override of the resident-workgroup cap of the throughput-regime launches: air_resident_grid in air_common.h)
template <typename K>
static inline int cv_resident_cap(K kernel, int threads, size_t lds, int fallback) {
    static const int forced = getenv("AIR_CANVAS_GRID") ? atoi(getenv("AIR_CANVAS_GRID")) : 0;
    return forced > 0 ? forced : air_resident_grid(kernel, threads, lds, fallback);
}
static inline int cv_check_dims(int n, int H, int W, int h, int w) {
    if (n <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0) return AIR_E_SHAPE;
    return AIR_OK;
}
#define CV_MAX_LDS (160 * 1024)
template <typename K>
static inline int cv_allow_lds(K kernel, size_t lds) {      // dynamic LDS above 64 KiB must be opted into per kernel
    if (lds <= 64 * 1024) return AIR_OK;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return e == hipSuccess ? AIR_OK : (int)e;
}
// rows per band / number of bands actually used for a request of `want` bands
static inline void wr_bands(int H, int want, int *NB, int *RB) {
    int nb = want < 1 ? 1 : (want > H ? H : want);
    const int rb = (H + nb - 1) / nb;
    nb = (H + rb - 1) / rb;                                  // drop empty trailing bands
    *NB = nb; *RB = rb;
}
static inline int bwd_threads(long units) { return units <= 512 ? 512 : ST_THREADS; }
static int launch_write_fwd(const float *glimpse, const float *where, const float *presence, const float *canvas_in,
                            const float *obs, float *canvas_steps, float *final_canvas, float *rec_parts, int n_bands,
                            int T, int B, int H, int W, int h, int w, float mult, float std, void *stream) {
    int NB, RB;
    wr_bands(H, n_bands, &NB, &RB);
    AIR_REQUIRE(NB == n_bands || !rec_parts, AIR_E_SHAPE);   // the caller sized rec_parts for exactly n_bands shares
    const size_t lds = carve_wr_bytes(T, RB, W, h, w);
    AIR_REQUIRE(lds <= CV_MAX_LDS, AIR_E_UNSUPPORTED);
    const int vec4g = (w % 4 == 0) && air_aligned16(glimpse);   // 16-byte groups that never straddle a glimpse row
    { int st_ = cv_allow_lds(st_write_fwd_kernel, lds); if (st_) return st_; }
    // one pixel per thread while the launch is far from filling the chip (latency regime), 256-thread workgroups beyond
    const long units = (long)B * NB;
    int wr_threads = units <= 512 ? 512 : ST_THREADS;          // (measured at 50x50: 512 units 14 us with 512 threads, 16 / 17.5 with 1024 / 256)
    if (units <= 256) {
        const int px = RB * W;
        wr_threads = px >= 1024 ? 1024 : ((px + 63) / 64) * 64;
        if (wr_threads < 64) wr_threads = 64;
    }
    const WriteFwdArgs a = {glimpse, where, presence, canvas_in, obs, canvas_steps, final_canvas, rec_parts, T, B, NB, RB, H, W, h, w,
                            lin_step(W), lin_step(H), mult, std, vec4g};
    const int cap = units > 256 * 8 ? cv_resident_cap(st_write_fwd_kernel, wr_threads, lds, 256 * 8) : 256 * 8;
    hipLaunchKernelGGL(st_write_fwd_kernel, dim3(cv_grid(units, cap)), dim3(wr_threads), lds, air_stream(stream), a);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

extern "C" int air_st_write_fwd(const float *glimpse, const float *where, const float *presence,
                                const float *canvas_in, float *canvas_out, int n, int H, int W, int h, int w,
                                void *stream) {
    AIR_REQUIRE(glimpse && where && canvas_out, AIR_E_NULL);
    int st = cv_check_dims(n, H, W, h, w);
    if (st) return st;
    return launch_write_fwd(glimpse, where, presence, canvas_in, nullptr, nullptr, canvas_out, nullptr, 1, 1, n, H, W, h,
                            w, 1.0f, 1.0f, stream);
}

extern "C" int air_canvas_unroll_fwd(const float *glimpse, const float *where, const float *presence,
                                     const float *obs, float *canvas_steps, float *final_canvas,
                                     float *rec_per_sample, int T, int B, int H, int W, int h, int w, float mult,
                                     float std, void *stream) {
    AIR_REQUIRE(glimpse && where && (final_canvas || canvas_steps), AIR_E_NULL);
    AIR_REQUIRE(!rec_per_sample || obs, AIR_E_NULL);
    AIR_REQUIRE(T > 0, AIR_E_SHAPE);
    int st = cv_check_dims(B, H, W, h, w);
    if (st) return st;
    // the complete per-sample reconstruction term needs the whole image in one workgroup; without it the bands are free
    int nb = 1;
    if (!rec_per_sample && (long)B * T <= 1024) nb = 256 / B < 1 ? 1 : 256 / B;
    int NB, RB;
    wr_bands(H, nb, &NB, &RB);
    return launch_write_fwd(glimpse, where, presence, nullptr, obs, canvas_steps, final_canvas, rec_per_sample, NB, T, B,
                            H, W, h, w, mult, std, stream);
}

extern "C" int air_canvas_unroll_bands(int B, int H) {
    int nb = 256 / (B < 1 ? 1 : B);
    if (nb > 8) nb = 8;
    int NB, RB;
    wr_bands(H, nb, &NB, &RB);
    return NB;
}

extern "C" int air_canvas_unroll_fwd_banded(const float *glimpse, const float *where, const float *presence,
                                            const float *obs, float *canvas_steps, float *final_canvas,
                                            float *rec_parts, int n_bands, int T, int B, int H, int W, int h, int w,
                                            float mult, float std, void *stream) {
    AIR_REQUIRE(glimpse && where && (final_canvas || canvas_steps), AIR_E_NULL);
    AIR_REQUIRE(!rec_parts || obs, AIR_E_NULL);
    AIR_REQUIRE(T > 0 && n_bands > 0, AIR_E_SHAPE);
    int st = cv_check_dims(B, H, W, h, w);
    if (st) return st;
    return launch_write_fwd(glimpse, where, presence, nullptr, obs, canvas_steps, final_canvas, rec_parts, n_bands, T, B,
                            H, W, h, w, mult, std, stream);
}

static int launch_write_bwd(const float *glimpse, const float *where, const float *presence, const float *dcanvas,
                            const float *final_canvas, const float *obs, float *dglimpse, float *dwhere,
                            float *dpresence, int T, int B, int H, int W, int h, int w, float mult, float std,
                            float loss_scale, void *stream, const NvilArgs *nvil = nullptr) {
    const bool rc = !dcanvas && !final_canvas;                // recompute form: the canvas is re-formed on the unit's footprint
    NvilArgs nv = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1, nullptr, nullptr};
    if (nvil) nv = *nvil;
    const int vec4g = (w % 4 == 0) && air_aligned16(glimpse);   // 16-byte groups that never straddle a glimpse row
    const int vec4c = ((H * W) % 4 == 0) && (dcanvas ? air_aligned16(dcanvas) : ((rc || air_aligned16(final_canvas)) && air_aligned16(obs)));
    // 512 threads while the launch does not fill the chip; beyond that 256-thread workgroups hide each other's barriers
    const int wr_threads = bwd_threads((long)B * T);
    const WriteBwdArgs a = {glimpse, where, presence, dcanvas, final_canvas, obs, dglimpse, dwhere, dpresence, T, B, H, W, h, w,
                            lin_step(W), lin_step(H), mult, std, loss_scale, vec4g, vec4c, 1};
    // developer switches, read at every call (tests and A/B runs switch forms inside one process; a captured graph keeps what it
    // captured): AIR_CANVAS_BWD_IMG=0 keeps the unit-major form beyond 2048 units, AIR_CANVAS_GS_THREADS forces the workgroup size,
    // AIR_CANVAS_IMG_MIN_UNITS moves the switch to the image-major form
    const char *env_img = getenv("AIR_CANVAS_BWD_IMG");
    const int img_major = env_img ? atoi(env_img) : 1;
    const char *env_thr = getenv("AIR_CANVAS_GS_THREADS");
    int thr_f = env_thr ? atoi(env_thr) : 0;
    if (thr_f != 128 && thr_f != 256 && thr_f != 512 && thr_f != 1024) thr_f = 0;
    const char *env_min = getenv("AIR_CANVAS_IMG_MIN_UNITS");
    const long img_min_units = env_min ? atol(env_min) : 256 * 8;
    if (rc) {                                                   // recompute form (the stand-alone launch; the fused one: air_canvas_unroll_fwd_bwd)
        const int thr_r = thr_f && thr_f <= 512 ? thr_f : wr_threads;
        const size_t lds_r = carve_gs_bytes(H, W, h, w, 1, T, thr_r / 64);
        AIR_REQUIRE(lds_r <= CV_MAX_LDS, AIR_E_UNSUPPORTED);
        { int st_ = cv_allow_lds(st_write_bwd_gs_kernel<true, false>, lds_r); if (st_) return st_; }
        const int grid_r = cv_grid((long)T * B, 256 * 8) + (nvil ? 1 : 0);
        hipLaunchKernelGGL((st_write_bwd_gs_kernel<true, false>), dim3(grid_r), dim3(thr_r), lds_r, air_stream(stream), a, nv);
        AIR_LAUNCH_CHECK();
        return AIR_OK;
    }
    if (img_major && !dcanvas && final_canvas && (long)T * B > img_min_units && T <= 8) {
        // throughput regime: one workgroup per image; threads from the carve: as many workgroups per CU as the LDS allows, 16 waves per CU
        int thr_i = thr_f;
        if (!thr_i) {
            const size_t l256 = carve_gs_bytes(H, W, h, w, T, T, 4);
            thr_i = l256 <= 40 * 1024 ? 256 : (l256 <= 78 * 1024 ? 512 : 1024);
        }
        const size_t lds_i = carve_gs_bytes(H, W, h, w, T, T, thr_i / 64);
        if (lds_i <= CV_MAX_LDS) {
            { int st_ = cv_allow_lds(st_write_bwd_gs_kernel<false, true>, lds_i); if (st_) return st_; }
            const int cap_i = cv_resident_cap(st_write_bwd_gs_kernel<false, true>, thr_i, lds_i, 256 * 2);
            const int grid_i = cv_grid(B, cap_i) + (nvil ? 1 : 0);
            hipLaunchKernelGGL((st_write_bwd_gs_kernel<false, true>), dim3(grid_i), dim3(thr_i), lds_i, air_stream(stream), a, nv);
            AIR_LAUNCH_CHECK();
            return AIR_OK;
        }
    }
    int thr_u = thr_f && thr_f <= 512 ? thr_f : wr_threads;
    if (!thr_f && carve_gs_bytes(H, W, h, w, 1, 1, 4) > 40 * 1024) thr_u = 512;
    const size_t lds_u = carve_gs_bytes(H, W, h, w, 1, 1, thr_u / 64);
    AIR_REQUIRE(lds_u <= CV_MAX_LDS, AIR_E_UNSUPPORTED);
    { int st_ = cv_allow_lds(st_write_bwd_gs_kernel<false, false>, lds_u); if (st_) return st_; }
    int cap_u = 256 * 8;
    if ((long)T * B > cap_u) cap_u = cv_resident_cap(st_write_bwd_gs_kernel<false, false>, thr_u, lds_u, cap_u);
    const int grid_u = cv_grid((long)T * B, cap_u) + (nvil ? 1 : 0);
    hipLaunchKernelGGL((st_write_bwd_gs_kernel<false, false>), dim3(grid_u), dim3(thr_u), lds_u, air_stream(stream), a, nv);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

extern "C" int air_st_write_bwd(const float *glimpse, const float *where, const float *presence,
                                const float *dcanvas, float *dglimpse, float *dwhere, float *dpresence, int n, int H,
                                int W, int h, int w, void *stream) {
    AIR_REQUIRE(glimpse && where && dcanvas && dglimpse && dwhere, AIR_E_NULL);
    int st = cv_check_dims(n, H, W, h, w);
    if (st) return st;
    return launch_write_bwd(glimpse, where, presence, dcanvas, nullptr, nullptr, dglimpse, dwhere, dpresence, 1, n, H,
                            W, h, w, 1.0f, 1.0f, 1.0f, stream);
}

extern "C" int air_canvas_unroll_bwd(const float *glimpse, const float *where, const float *presence,
                                     const float *obs, const float *final_canvas, float *dglimpse, float *dwhere,
                                     int T, int B, int H, int W, int h, int w, float mult, float std,
                                     float loss_scale, void *stream) {
    AIR_REQUIRE(glimpse && where && obs && dglimpse && dwhere, AIR_E_NULL);      // final_canvas == NULL: the recompute form
    AIR_REQUIRE(T > 0, AIR_E_SHAPE);
    int st = cv_check_dims(B, H, W, h, w);
    if (st) return st;
    return launch_write_bwd(glimpse, where, presence, nullptr, final_canvas, obs, dglimpse, dwhere, nullptr, T, B, H, W,
                            h, w, mult, std, loss_scale, stream);
}

// air_canvas_unroll_bwd + dpresence[T,B] = sum_pix dL/dcanvas * (the step's write): what a CONTINUOUS presence (discrete_steps=False,
// cell.py:150-151, 163) receives from the canvas write
extern "C" int air_canvas_unroll_bwd_dpresence(const float *glimpse, const float *where, const float *presence,
                                               const float *obs, const float *final_canvas, float *dglimpse, float *dwhere,
                                               float *dpresence, int T, int B, int H, int W, int h, int w, float mult, float std,
                                               float loss_scale, void *stream) {
    AIR_REQUIRE(glimpse && where && obs && dglimpse && dwhere && dpresence, AIR_E_NULL);
    AIR_REQUIRE(T > 0, AIR_E_SHAPE);
    int st = cv_check_dims(B, H, W, h, w);
    if (st) return st;
    return launch_write_bwd(glimpse, where, presence, nullptr, final_canvas, obs, dglimpse, dwhere, dpresence, T, B, H, W,
                            h, w, mult, std, loss_scale, stream);
}

extern "C" int air_canvas_unroll_bwd_nvil(const float *glimpse, const float *where, const float *presence,
                                          const float *obs, const float *final_canvas, float *dglimpse, float *dwhere,
                                          int T, int B, int H, int W, int h, int w, float mult, float std,
                                          float loss_scale, const float *imp_parts, int n_parts, float *imp_sum,
                                          const float *baseline, const float *logp, float *nvil_out, float *dlogp,
                                          float *dbaseline, float *ema_dev, void *stream) {
    AIR_REQUIRE(glimpse && where && obs && dglimpse && dwhere, AIR_E_NULL);      // final_canvas == NULL: the recompute form
    AIR_REQUIRE(imp_parts && baseline && logp && nvil_out, AIR_E_NULL);
    AIR_REQUIRE(T > 0 && n_parts > 0, AIR_E_SHAPE);
    int st = cv_check_dims(B, H, W, h, w);
    if (st) return st;
    const NvilArgs nv = {imp_parts, baseline, logp, nvil_out, dlogp, dbaseline, B, n_parts, imp_sum, ema_dev};
    return launch_write_bwd(glimpse, where, presence, nullptr, final_canvas, obs, dglimpse, dwhere, nullptr, T, B, H, W,
                            h, w, mult, std, loss_scale, stream, &nv);
}

template <bool SPLIT>
__global__ __launch_bounds__(1024) void canvas_fused_gs_kernel(WriteFwdArgs f, WriteBwdArgs b, int n_fwd) {
    extern __shared__ __align__(16) float smem[];
    // the backward role FIRST in the grid: it is the longer chain (operands, tables, canvas pass, two contractions against the
    // forward's one pass), and workgroups are dispatched in index order
    const int n_bwd = (int)gridDim.x - n_fwd;
    if ((int)blockIdx.x >= n_bwd) st_write_fwd_body(f, smem, (int)blockIdx.x - n_bwd, n_fwd);
    else {
        const NvilArgs none = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1, nullptr, nullptr};
        st_write_bwd_gs_body<true, false, SPLIT>(b, none, smem, (int)blockIdx.x, n_bwd);
    }
}
// The fused launch's shape for a problem: threads per workgroup, and whether it fits (LDS of both roles, both grids small
// enough to run side by side).  n_split = workgroups per backward unit (1 or 2).
static int fused_shape(int n_bands, int n_split, int T, int B, int H, int W, int h, int w, int *threads, size_t *lds) {
    int NB, RB;
    wr_bands(H, n_bands, &NB, &RB);
    if (NB != n_bands) return AIR_E_SHAPE;
    if (n_split < 1 || n_split > 4) return AIR_E_SHAPE;
    if ((long)B * NB > 4096 || (long)B * T * n_split > 4096) return AIR_E_UNSUPPORTED;
    const int nt = bwd_threads((long)B * T * n_split);
    const size_t lds_f = carve_wr_bytes(T, RB, W, h, w), lds_b = carve_gs_bytes(H, W, h, w, 1, T, nt / 64);
    *lds = lds_f > lds_b ? lds_f : lds_b;
    *threads = nt;
    return *lds <= CV_MAX_LDS ? AIR_OK : AIR_E_UNSUPPORTED;
}
// 1 when air_canvas_unroll_fwd_bwd takes this problem, 0 otherwise (the caller then plans the two launches)
extern "C" int air_canvas_unroll_fwd_bwd_fits(int n_bands, int n_split, int T, int B, int H, int W, int h, int w) {
    int threads; size_t lds;
    if (T <= 0 || cv_check_dims(B, H, W, h, w)) return 0;
    return fused_shape(n_bands, n_split, T, B, H, W, h, w, &threads, &lds) == AIR_OK ? 1 : 0;
}
// forward (banded, as air_canvas_unroll_fwd_banded) + backward (recompute form of air_canvas_unroll_bwd) as ONE launch.
// n_split workgroups per backward unit: dwhere then holds n_split slabs [n_split][T*B][4] whose SUM is the gradient.
extern "C" int air_canvas_unroll_fwd_bwd(const float *glimpse, const float *where, const float *presence, const float *obs,
                                         float *canvas_steps, float *final_canvas, float *rec_parts, int n_bands,
                                         float *dglimpse, float *dwhere, int n_split, int T, int B, int H, int W, int h, int w,
                                         float mult, float std, float loss_scale, void *stream) {
    AIR_REQUIRE(glimpse && where && obs && (final_canvas || canvas_steps) && rec_parts && dglimpse && dwhere, AIR_E_NULL);
    AIR_REQUIRE(T > 0 && n_bands > 0, AIR_E_SHAPE);
    int st = cv_check_dims(B, H, W, h, w);
    if (st) return st;
    int threads; size_t lds;
    st = fused_shape(n_bands, n_split, T, B, H, W, h, w, &threads, &lds);
    if (st) return st;
    int NB, RB;
    wr_bands(H, n_bands, &NB, &RB);
    const int vec4g = (w % 4 == 0) && air_aligned16(glimpse);
    const int vec4c = ((H * W) % 4 == 0) && air_aligned16(obs);
    { int st_ = n_split > 1 ? cv_allow_lds(canvas_fused_gs_kernel<true>, lds) : cv_allow_lds(canvas_fused_gs_kernel<false>, lds); if (st_) return st_; }
    const WriteFwdArgs f = {glimpse, where, presence, nullptr, obs, canvas_steps, final_canvas, rec_parts, T, B, NB, RB, H, W, h, w,
                            lin_step(W), lin_step(H), mult, std, vec4g};
    const WriteBwdArgs b = {glimpse, where, presence, nullptr, nullptr, obs, dglimpse, dwhere, nullptr, T, B, H, W, h, w,
                            lin_step(W), lin_step(H), mult, std, loss_scale, vec4g, vec4c, n_split};
    const int n_fwd = B * NB;
    if (n_split > 1) hipLaunchKernelGGL(canvas_fused_gs_kernel<true>, dim3(n_fwd + T * B * n_split), dim3(threads), lds, air_stream(stream), f, b, n_fwd);
    else hipLaunchKernelGGL(canvas_fused_gs_kernel<false>, dim3(n_fwd + T * B), dim3(threads), lds, air_stream(stream), f, b, n_fwd);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
