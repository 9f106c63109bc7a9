// Inverse spatial transformer ("canvas write", cell.py:159-165 / modules.py:101-109) for gfx950, row-streaming form (round 4).
//
//   forward : canvas_t = canvas_{t-1} + presence_t * bilinear(glimpse_t; x_g = (w-1)/2*(X_J/sx - tx/sx + 1), y_g likewise), all T
//             steps of an image in one pass, per-step canvases, final canvas and the band's share of the Gaussian
//             log-likelihood (model.py:319-324);
//   backward: dglimpse_t = Wy^T . (presence_t * dcanvas) . Wx, dwhere_t, optional dpresence_t, for every (t, b).
//
// Both are VALU-issue bound on this chip (a wave64 instruction occupies its SIMD for four cycles; rounds 1-3 measured the flat
// "one thread per pixel" kernels at 45 / 130-250 instructions per pixel).  What this form does about the instruction count:
//   * lane = canvas COLUMN, a wave walks canvas ROWS.  Everything that depends on the row only -- the row's table entry
//     (floor index, weights), its validity, the LDS row base of the taps -- is wave-uniform: it lives in SGPRs
//     (v_readfirstlane) and "this row is outside step t's footprint" is a SCALAR branch that skips the whole bilinear read.
//     No per-pixel index division, no per-pixel validity selects for y.
//   * table entries carry (floor, d, 1 - d), so the complementary weights are not recomputed per pixel.
//   * backward: the row contraction of dG = Wy^T . g . Wx is fused into the pixel pass -- a pixel adds g*dy and g*(1-dy) to the
//     two glimpse rows it touches, S[i, J], with LDS adds on lane-private addresses in row order (deterministic) -- and the
//     column contraction runs once per GLIMPSE row afterwards (lane = glimpse column, weights in registers).  The full-image
//     dcanvas staging of the old kernel (2500 pixels for a 750-pixel footprint) is gone: dcanvas is formed per row from the
//     final canvas and the observation, one row ahead.
//   * waves of a workgroup split a unit by OUTPUT rows (canvas rows in the forward, glimpse rows in the backward), so they only
//     meet at two barriers (operands staged | the dwhere partials); a unit can also be split over `NS` workgroups -- disjoint
//     dglimpse rows, dwhere as NS slabs the consumer adds (air_attend_bwd) -- which is what shortens the per-unit chain in the
//     latency regime (batch 64: 192 units on 256 CUs).
// The forward's arithmetic is the oracle's, operation by operation (st_device.h: bilerp_pre == bilerp, axis_entry, grid_coord),
// so its results stay bit-identical to oracle/air_oracle.py st_write; the recompute form of the backward re-forms the canvas
// with the same calls.  Everything else in this file contracts to FMA.
#include "st_device.h"
#include "nvil_device.h"

// ---- table entries: {floor index as int bits | ST_INVALID, d = (floor + 1) - coord, 1 - d, unused} --------------------------
__device__ __forceinline__ float4 axis_entry4(float coord, int extent) {
#pragma clang fp contract(off)
    int f; float d;
    axis_entry(coord, extent, &f, &d);
    return make_float4(__int_as_float(f), d, 1.0f - d, 0.f);
}
// bilerp of st_device.h with the complementary weights handed in (mx = 1 - dx and my = 1 - dy are the same roundings there)
__device__ __forceinline__ float bilerp_pre(const Taps &t, float dx, float mx, float dy, float my) {
#pragma clang fp contract(off)
    float r = (dx * dy) * t.ff;
    r = r + (mx * my) * t.cc;
    r = r + (dx * my) * t.cf;
    r = r + (mx * dy) * t.fc;
    return r;
}
// canvas accumulation step, rounded as the oracle's `canvas + presence * inversed` (cell.py:164)
__device__ __forceinline__ float acc_step(float acc, float p, float v) {
#pragma clang fp contract(off)
    const float pv = p * v;
    return acc + pv;
}
__device__ __forceinline__ int rfl_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float rfl_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// bordered LDS copies of `n_g` glimpses of image b (step-major in global memory: glimpse[(t*B + b)*hw]); the borders were zeroed
// once by the caller
__device__ __forceinline__ void stage_glimpses(float *dst, int hwp, const float *__restrict__ glimpse, int t0, int n_g, int B, int b,
                                               int h, int w, bool vec4) {
    const int hw = h * w, tid = threadIdx.x, nt = blockDim.x;
    const float inv_w = 1.0f / (float)w;
    if (vec4) {                                                // (w % 4 == 0: a 16-byte group never straddles a glimpse row)
        const int nq = hw >> 2;
        for (int e = tid; e < n_g * nq; e += nt) {
            const int t = e / nq, q = e - t * nq;
            const float4 v = reinterpret_cast<const float4 *>(glimpse + ((size_t)(t0 + t) * B + b) * hw)[q];
            float *d = dst + (size_t)t * hwp + pad_index(4 * q, w, inv_w);
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    } else {
        for (int e = tid; e < n_g * hw; e += nt) {
            const int t = e / hw, q = e - t * hw;
            dst[(size_t)t * hwp + pad_index(q, w, inv_w)] = glimpse[((size_t)(t0 + t) * B + b) * hw + q];
        }
    }
}
__device__ __forceinline__ void zero_borders(float *dst, int hwp, int n_g, int h, int w) {
    const int nb = pad_border(h, w);
    for (int e = threadIdx.x; e < n_g * nb; e += blockDim.x) {
        const int t = e / nb;
        dst[(size_t)t * hwp + pad_border_index(e - t * nb, h, w)] = 0.f;
    }
}

// ============================================================================================================
// forward
// ============================================================================================================
struct WriteFwdArgs {
    const float *glimpse, *where, *presence, *canvas_in, *obs;
    float *canvas_steps, *final_canvas, *rec_parts;
    int T, B, NB, RB, H, W, h, w;
    double stepX, stepY;
    float mult, std;
    int vec4_glimpse;
};
struct FwdCarve {
    float *glm, *pres, *scratch;
    float4 *xe, *ye;
    int hwp;
};
__device__ __forceinline__ FwdCarve carve_fwd(float *smem, int T, int RB, int W, int h, int w) {
    FwdCarve c;
    c.hwp = pad_count(h, w);
    float *p = smem;
    c.glm = p; p += (size_t)T * c.hwp;
    c.xe = reinterpret_cast<float4 *>(p); p += 4 * T * W;
    c.ye = reinterpret_cast<float4 *>(p); p += 4 * T * RB;
    c.pres = p; p += (T + 3) & ~3;
    c.scratch = p;
    return c;
}
static inline size_t carve_fwd_bytes(int T, int RB, int W, int h, int w) {
    return sizeof(float) * ((size_t)T * pad_count_host(h, w) + 4 * (size_t)T * (W + RB) + ((T + 3) & ~3) + 64);
}

// One workgroup per (image, row band); its waves take contiguous runs of the band's rows.  rec_parts[band*B + b] receives the
// band's share of the reconstruction term (NB = 1: the complete per-sample term); the consumer adds the NB shares in band order.
__device__ __forceinline__ void canvas_fwd_body(const WriteFwdArgs &a, float *smem, const int vblock, const int vgrid) {
    const float *__restrict__ where = a.where, *__restrict__ presence = a.presence;
    const float *__restrict__ canvas_in = a.canvas_in, *__restrict__ obs = a.obs;
    float *__restrict__ canvas_steps = a.canvas_steps, *__restrict__ final_canvas = a.final_canvas, *__restrict__ rec_parts = a.rec_parts;
    const int T = a.T, B = a.B, NB = a.NB, RB = a.RB, H = a.H, W = a.W, h = a.h, w = a.w;
    const float mult = a.mult, std = a.std;
    AIR_TR_INIT();
    const int HW = H * W, tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
    FwdCarve c = carve_fwd(smem, T, RB, W, h, w);
    const float cxs = (float)((w - 1) / 2.0), cys = (float)((h - 1) / 2.0);
    const float cst = 0.5f * logf(6.283185307179586f) + logf(std);
    const float inv_std = 1.0f / std;
    const int n_units = B * NB, pitch = w + 2;
    zero_borders(c.glm, c.hwp, T, h, w);
    for (int unit = vblock; unit < n_units; unit += vgrid) {
        const int b = unit % B, band = unit / B;
        const int r0 = band * RB, r1 = (r0 + RB < H) ? r0 + RB : H, nrow = r1 - r0;
        AIR_TR(0);
        if (unit != vblock) __syncthreads();                   // grid-stride reuse of the carve
        stage_glimpses(c.glm, c.hwp, a.glimpse, 0, T, B, b, h, w, a.vec4_glimpse != 0);
        for (int e = tid; e < T * (W + nrow); e += nt) {
            const int t = e / (W + nrow), r = e - t * (W + nrow);
            const float *wk = where + 4 * ((size_t)t * B + b);
            if (r < W) {
                const float sx = wk[0], tx = wk[1];
                c.xe[t * W + r] = axis_entry4(grid_coord(1.0f / sx, lin_m11(r, W, a.stepX), -tx / sx, cxs), w);
            } else {
                const float sy = wk[2], ty = wk[3];
                const int i = r - W;
                c.ye[t * RB + i] = axis_entry4(grid_coord(1.0f / sy, lin_m11(r0 + i, H, a.stepY), -ty / sy, cys), h);
            }
        }
        if (tid < T) c.pres[tid] = presence ? presence[(size_t)tid * B + b] : 1.0f;
        AIR_TR(1);
        __syncthreads();
        AIR_TR(2);
        const int rpw = (nrow + nw - 1) / nw;
        const int ra = r0 + wv * rpw, rb = (ra + rpw < r1) ? ra + rpw : r1;
        float s = 0.f;
        for (int I = ra; I < rb; ++I) {
            const int Ib = I - r0;
            for (int jc = 0; jc < W; jc += 64) {
                const int J = jc + lane;
                const bool on = J < W;
                const int Jc = on ? J : W - 1;
                const size_t gp = (size_t)b * HW + (size_t)I * W + Jc;
                const float xo = rec_parts ? obs[gp] : 0.f;
                float acc = canvas_in ? canvas_in[gp] : 0.f;
                for (int t = 0; t < T; ++t) {
                    const float4 ey = c.ye[t * RB + Ib];
                    const int fy = rfl_i(__float_as_int(ey.x));
                    float v = 0.f;
                    if (fy != ST_INVALID) {                    // scalar branch: rows outside step t's footprint cost nothing
                        const float dy = rfl_f(ey.y), my = rfl_f(ey.z);
                        const float4 ex = c.xe[t * W + Jc];
                        const int fx = __float_as_int(ex.x);
                        const Taps tp = load_taps_pad(c.glm + (size_t)t * c.hwp, pitch, fy, fx != ST_INVALID ? fx : -1);
                        const float r = bilerp_pre(tp, ex.y, ex.z, dy, my);
                        v = fx != ST_INVALID ? r : 0.f;
                    }
                    acc = acc_step(acc, c.pres[t], v);
                    if (canvas_steps && on) canvas_steps[((size_t)t * B + b) * HW + (size_t)I * W + J] = acc;
                }
                if (final_canvas && on) final_canvas[gp] = acc;
                if (rec_parts && on) {
                    const float z = (xo - mult * acc) * inv_std;
                    s += 0.5f * z * z + cst;
                }
            }
        }
        AIR_TR(3);
        if (rec_parts) {                                       // lanes -> wave -> the workgroup's waves in order
            s = wave_sum(s);
            if (nw == 1) {
                if (lane == 0) rec_parts[(size_t)band * B + b] = s;
            } else {
                if (lane == 0) c.scratch[wv] = s;
                __syncthreads();
                if (tid == 0) {
                    float tot = c.scratch[0];
                    for (int q = 1; q < nw; ++q) tot += c.scratch[q];
                    rec_parts[(size_t)band * B + b] = tot;
                }
            }
        }
        AIR_TR(4);
    }
    AIR_TR_FLUSH();
}
__global__ __launch_bounds__(1024) void canvas_fwd_kernel(WriteFwdArgs a) {
    extern __shared__ __align__(16) float smem[];
    canvas_fwd_body(a, smem, (int)blockIdx.x, (int)gridDim.x);
}

// ============================================================================================================
// backward
// ============================================================================================================
// dcanvas is either given per step ([T*B,H,W]), or formed on the fly from the reconstruction term
//   dcanvas[b,p] = loss_scale * mult * (mult*final[b,p] - obs[b,p]) / std^2   (shared by all t)
// with final either read (stored-canvas form) or re-formed on the unit's footprint from the T glimpses of its image, with the
// forward's own calls (RC, "recompute" form: the backward then reads nothing the forward writes, and the two can be roles of one
// launch).
struct WriteBwdArgs {
    const float *glimpse, *where, *presence, *dcanvas, *final_canvas, *obs;
    float *dglimpse, *dwhere, *dpresence;
    int T, B, H, W, h, w;
    double stepX, stepY;
    float mult, std, loss_scale;
    int vec4_glimpse, vec4_canvas;
    int NS;                           // workgroups per unit: dglimpse rows are disjoint, dwhere is written as NS slabs [NS][T*B][4]
};
struct BwdCarve {
    float *src, *X, *Y, *pres, *img, *S, *scratch;
    float4 *xe, *ye;
    int hwp, s_rows;
};
// n_src = 1: the unit's own glimpse / tables; n_src = T (recompute form): those of all T steps of the unit's image, step-major
__device__ __forceinline__ int bwd_s_rows(int h, int G) { return (h + G - 1) / G; }
__device__ __forceinline__ BwdCarve carve_bwd(float *smem, int H, int W, int h, int w, int n_src, bool rc, int nw, int G) {
    BwdCarve c;
    float *p = smem;
    c.hwp = pad_count(h, w);
    c.src = p; p += (size_t)n_src * c.hwp;
    c.xe = reinterpret_cast<float4 *>(p); p += 4 * W * n_src;
    c.ye = reinterpret_cast<float4 *>(p); p += 4 * H * n_src;
    c.X = p; p += (W + 3) & ~3;
    c.Y = p; p += (H + 3) & ~3;
    c.pres = p; p += (n_src + 3) & ~3;
    c.img = p; p += rc ? ((H * W + 3) & ~3) : 0;
    c.s_rows = bwd_s_rows(h, G);
    c.S = p; p += (size_t)nw * c.s_rows * W;
    c.scratch = p;
    return c;
}
static inline size_t carve_bwd_bytes(int H, int W, int h, int w, int n_src, bool rc, int nw, int G) {
    const size_t s_rows = (size_t)((h + G - 1) / G);
    return sizeof(float) * ((size_t)n_src * pad_count_host(h, w) + 4 * (size_t)n_src * (W + H) + ((W + 3) & ~3) + ((H + 3) & ~3) +
                            ((n_src + 3) & ~3) + (rc ? ((H * W + 3) & ~3) : 0) + (size_t)nw * s_rows * W + 16 * 8 + 16);
}
// [first, last] row of an axis table whose floor index lies in [f_lo, f_hi] (a monotone map: the set is an interval); every lane
// gets the result; empty => first > last
__device__ __forceinline__ int2 floor_span(const float4 *tab, int n, int f_lo, int f_hi) {
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
// exact [lo, hi] of canvas indices whose taps touch source index j, from a float4 axis table (see touch_range of st_device.h)
__device__ __forceinline__ int2 touch_range4(const float4 *tab, float b, float inv_a, float inv_cs, int j, int n) {
    int lo, hi;
    src_range(inv_a, b, inv_cs, (float)(j - 1), (float)(j + 1), n, &lo, &hi);
    if (hi - lo < 8) {
        unsigned mask = 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int J = lo + u;
            const int f = __float_as_int(tab[J <= hi ? J : lo].x);
            if (J <= hi && f != ST_INVALID && (f == j || f + 1 == j)) mask |= 1u << u;
        }
        if (!mask) return make_int2(1, 0);
        return make_int2(lo + __ffs((int)mask) - 1, lo + 31 - __clz((int)mask));
    }
    while (lo <= hi) { const int f = __float_as_int(tab[lo].x); if (f != ST_INVALID && (f == j || f + 1 == j)) break; ++lo; }
    while (hi >= lo) { const int f = __float_as_int(tab[hi].x); if (f != ST_INVALID && (f == j || f + 1 == j)) break; --hi; }
    return make_int2(lo, hi);
}

template <bool RC>
__device__ __forceinline__ void canvas_bwd_body(const WriteBwdArgs &a, const NvilArgs &nv, float *smem, const int vblock, const int vgrid) {
    const float *__restrict__ where = a.where, *__restrict__ presence = a.presence;
    const float *__restrict__ dcanvas = a.dcanvas, *__restrict__ final_canvas = a.final_canvas, *__restrict__ obs = a.obs;
    float *__restrict__ dglimpse = a.dglimpse, *__restrict__ dwhere = a.dwhere, *__restrict__ dpresence = a.dpresence;
    const int T = a.T, B = a.B, H = a.H, W = a.W, h = a.h, w = a.w, NS = a.NS;
    const float mult = a.mult;
    AIR_TR_INIT();
    // optional second role: the FIRST workgroup evaluates the NVIL objective (a long float64 chain, independent of the canvas
    // gradient; it only has to precede the baseline / logit backward that follow this launch)
    const int grid_st = nv.imp ? vgrid - 1 : vgrid;
    const int bid0 = nv.imp ? vblock - 1 : vblock;
    if (bid0 < 0) {
        AIR_TR(5);
        nvil_body(nv);
        AIR_TR(6);
        AIR_TR_FLUSH();
        return;
    }
    AIR_TR(0);
    const int HW = H * W, hw = h * w, tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
    const int G = NS * nw;                                     // row owners of a unit: NS workgroups x nw waves
    const int n_src = RC ? T : 1;
    BwdCarve c = carve_bwd(smem, H, W, h, w, n_src, RC, nw, G);
    const float cxs = (float)((w - 1) / 2.0), cys = (float)((h - 1) / 2.0);
    const float inv_cxs = 1.0f / cxs;
    const float coef = a.loss_scale * mult / (a.std * a.std);
    const int n = T * B, pitch = w + 2;
    zero_borders(c.src, c.hwp, n_src, h, w);
    float *const Sw = c.S + (size_t)wv * c.s_rows * W;        // this wave's S[il, J]: rows i0 .. i1-1 of dglimpse . Wx^-1 (see below)
    for (int u = bid0; u < n * NS; u += grid_st) {
        const int k = u / NS, sp = u - k * NS;
        const int b = k % B, t_own = k / B;
        const int g = sp * nw + wv;                            // this wave's owner index
        const int i0 = (int)(((long)h * g) / G), i1 = (int)(((long)h * (g + 1)) / G);   // its dglimpse rows [i0, i1)
        if (u != bid0) __syncthreads();                        // grid-stride reuse of the carve
        // ---- operands: `where` (vector path, see opaque_zero), the glimpse(s), RC: the observation; tables while they fly ----
        const int z0 = opaque_zero();
        const float sx = where[4 * (size_t)k + z0], tx = where[4 * (size_t)k + 1 + z0];
        const float sy = where[4 * (size_t)k + 2 + z0], ty = where[4 * (size_t)k + 3 + z0];
        const float pres = presence ? presence[k + z0] : 1.0f;
        const float *obp = obs ? obs + (size_t)b * HW : nullptr;
        float4 q_img = make_float4(0.f, 0.f, 0.f, 0.f);
        const int nQ = HW >> 2;
        if (RC && a.vec4_canvas) q_img = reinterpret_cast<const float4 *>(obp)[tid < nQ ? tid : nQ - 1];
        stage_glimpses(c.src, c.hwp, a.glimpse, RC ? 0 : t_own, n_src, B, b, h, w, a.vec4_glimpse != 0);
        for (int e = tid; e < n_src * (W + H); e += nt) {
            const int tt = e / (W + H), r = e - tt * (W + H);
            const float *wk = where + 4 * ((size_t)(RC ? tt : t_own) * B + b);
            if (r < W) {
                const float s_ = wk[0], t_ = wk[1];
                const float X = lin_m11(r, W, a.stepX);
                if (tt == 0) c.X[r] = X;
                c.xe[tt * W + r] = axis_entry4(grid_coord(1.0f / s_, X, -t_ / s_, cxs), w);
            } else {
                const float s_ = wk[2], t_ = wk[3];
                const int i = r - W;
                const float Y = lin_m11(i, H, a.stepY);
                if (tt == 0) c.Y[i] = Y;
                c.ye[tt * H + i] = axis_entry4(grid_coord(1.0f / s_, Y, -t_ / s_, cys), h);
            }
        }
        if (RC) {
            if (tid < T) c.pres[tid] = presence ? presence[(size_t)tid * B + b] : 1.0f;
            if (a.vec4_canvas) {
                if (tid < nQ) reinterpret_cast<float4 *>(c.img)[tid] = q_img;
                for (int q = tid + nt; q < nQ; q += nt) reinterpret_cast<float4 *>(c.img)[q] = reinterpret_cast<const float4 *>(obp)[q];
            } else {
                for (int p = tid; p < HW; p += nt) c.img[p] = obp[p];
            }
        }
        for (int e = lane; e < (i1 - i0) * W; e += 64) Sw[e] = 0.f;
        AIR_TR(7);
        __syncthreads();                                       // (1)
        AIR_TR(1);
        const float4 *xe = c.xe + (RC ? t_own * W : 0), *ye = c.ye + (RC ? t_own * H : 0);
        const float *src = c.src + (RC ? (size_t)t_own * c.hwp : 0);
        // footprint columns (valid x entries) and the canvas rows whose taps touch this wave's dglimpse rows: floor in [i0-1, i1-1]
        const int2 vx = floor_span(xe, W, -1, w - 1);
        const int2 vy = (i1 > i0) ? floor_span(ye, H, i0 - 1, i1 - 1) : make_int2(1, 0);
        const int J0 = vx.x, J1 = vx.y, Ia = vy.x, Ib = vy.y;
        AIR_TR(8);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // d/d(ax), d/d(bx), d/d(ay), d/d(by), dpresence, -, -, -
        const float *dcp = dcanvas ? dcanvas + (size_t)k * HW : nullptr;
        const float *fcp = final_canvas ? final_canvas + (size_t)b * HW : nullptr;
        const bool want_v = RC || dpresence != nullptr;
        if (J1 >= J0 && Ib >= Ia) {
            for (int jc = J0; jc <= J1; jc += 64) {
                const int J = jc + lane;
                const bool on = J <= J1;
                const int Jc = on ? J : J1;
                const float4 ex = xe[Jc];
                const int fx = __float_as_int(ex.x);           // valid: inside the footprint
                const float dx = ex.y, mx = ex.z, XJ = c.X[Jc];
                // the row's dcanvas operands, requested one row ahead (stored-canvas / given-dcanvas forms)
                float f_nxt = 0.f, o_nxt = 0.f;
                if (!RC) {
                    const size_t p = (size_t)Ia * W + Jc;
                    if (dcp) f_nxt = dcp[p]; else { f_nxt = fcp[p]; o_nxt = obp[p]; }
                }
                for (int I = Ia; I <= Ib; ++I) {
                    const float f_cur = f_nxt, o_cur = o_nxt;
                    if (!RC) {
                        const size_t p = (size_t)(I < Ib ? I + 1 : I) * W + Jc;
                        if (dcp) f_nxt = dcp[p]; else { f_nxt = fcp[p]; o_nxt = obp[p]; }
                    }
                    const float4 ey = ye[I];
                    const int fy = rfl_i(__float_as_int(ey.x));   // in [i0-1, i1-1] by construction of [Ia, Ib]
                    const float dy = rfl_f(ey.y), my = rfl_f(ey.z);
                    const Taps tp = load_taps_pad(src, pitch, fy, fx);
                    float v = 0.f;
                    if (want_v) v = bilerp_pre(tp, dx, mx, dy, my);
                    float dc;
                    if (RC) {
                        // the canvas at this pixel, accumulated as the forward does: ((0 + p0*v0) + p1*v1) + ... over ALL steps
                        float cv = 0.f;
                        for (int tt = 0; tt < T; ++tt) {
                            float vt = v;
                            if (tt != t_own) {
                                vt = 0.f;
                                const float4 eyt = c.ye[tt * H + I];
                                const int fyt = rfl_i(__float_as_int(eyt.x));
                                if (fyt != ST_INVALID) {
                                    const float4 ext = c.xe[tt * W + Jc];
                                    const int fxt = __float_as_int(ext.x);
                                    const Taps tq = load_taps_pad(c.src + (size_t)tt * c.hwp, pitch, fyt, fxt != ST_INVALID ? fxt : -1);
                                    const float r = bilerp_pre(tq, ext.y, ext.z, rfl_f(eyt.y), rfl_f(eyt.z));
                                    vt = fxt != ST_INVALID ? r : 0.f;
                                }
                            }
                            cv = acc_step(cv, c.pres[tt], vt);
                        }
                        dc = coef * (mult * cv - c.img[I * W + Jc]);
                    } else {
                        dc = dcp ? f_cur : coef * (mult * f_cur - o_cur);
                    }
                    if (!on) dc = 0.f;
                    const float go = pres * dc;
                    // dwhere / dpresence: every footprint row has ONE owner among the G row owners -- the one that holds
                    // dglimpse row clamp(fy, 0, h-1)
                    const int fyc = fy < 0 ? 0 : (fy > h - 1 ? h - 1 : fy);
                    if (fyc >= i0 && fyc < i1) {               // (scalar)
                        const float gx = dy * (tp.fc - tp.ff) + my * (tp.cc - tp.cf);
                        const float gy = dx * (tp.cf - tp.ff) + mx * (tp.cc - tp.fc);
                        const float gax = go * gx, gay = go * gy;
                        acc[0] += gax * XJ; acc[1] += gax;
                        acc[2] += gay * c.Y[I]; acc[3] += gay;
                        acc[4] += dc * v;
                    }
                    // row contraction, fused: S[i, J] += wy[I, i] * go for the (at most two) owned rows this canvas row touches.
                    // Lane-private addresses, rows in order: a plain LDS add, deterministic.
                    if (fy >= i0 && fy < i1) atomicAdd(&Sw[(fy - i0) * W + Jc], go * dy);
                    if (fy + 1 >= i0 && fy + 1 < i1) atomicAdd(&Sw[(fy + 1 - i0) * W + Jc], go * my);
                }
            }
        }
        AIR_TR(9);
        {
            const float r = wave_reduce8(acc);
            if ((lane & 7) == 0) c.scratch[wv * 8 + wave_reduce8_slot()] = r;
        }
        // column contraction: dG[i, j] = sum_J S[i, J] * wx[J, j] over the exact canvas-column range of glimpse column j
        // (lane = glimpse column: range and weights are loop invariants)
        __builtin_amdgcn_wave_barrier();
        if (i1 > i0) {
            float *dg = dglimpse + (size_t)k * hw;
            for (int j0 = 0; j0 < w; j0 += 64) {
                const int j = j0 + lane;
                if (j < w) {
                    const int2 r = touch_range4(xe, -tx / sx, sx, inv_cxs, j, W);     // 1/ax = sx
                    float wgt[4]; int Jq[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int J = r.x + q;
                        const bool in = J <= r.y;
                        Jq[q] = in ? J : (r.x <= r.y ? r.x : 0);
                        const float4 ex = xe[Jq[q]];
                        const int fx = __float_as_int(ex.x);
                        wgt[q] = in ? ((fx == j ? ex.y : 0.f) + (fx + 1 == j ? ex.z : 0.f)) : 0.f;
                    }
                    for (int il = 0; il < i1 - i0; ++il) {
                        const float *Srow = Sw + il * W;
                        float s = Srow[Jq[0]] * wgt[0] + Srow[Jq[1]] * wgt[1] + Srow[Jq[2]] * wgt[2] + Srow[Jq[3]] * wgt[3];
                        for (int J = r.x + 4; J <= r.y; ++J) {
                            const float4 ex = xe[J];
                            const int fx = __float_as_int(ex.x);
                            s += Srow[J] * ((fx == j ? ex.y : 0.f) + (fx + 1 == j ? ex.z : 0.f));
                        }
                        dg[(i0 + il) * w + j] = s;
                    }
                }
            }
        }
        AIR_TR(2);
        __syncthreads();                                       // (2) the waves' dwhere partials
        if (wv == 0) {
            float part[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) part[q] = (lane < nw && q < 5) ? c.scratch[lane * 8 + q] : 0.f;
            const float tot = wave_reduce8(part);
            const float r0 = __shfl(tot, 0, 64) * cxs, r1 = __shfl(tot, 8, 64) * cxs, r2 = __shfl(tot, 16, 64) * cys,
                        r3 = __shfl(tot, 24, 64) * cys, r4 = __shfl(tot, 32, 64);
            if (lane == 0) {
                // chain through a = 1/s, b = (-t)/s (linear in the partial sums: each of the NS slabs carries its share), written as
                // automatic differentiation evaluates the two divisions -- d(x/y) = g/y for x, -g * ((x/y)/y) for y -- so that a
                // degenerate scale (0, denormal: 1/s = inf; 1e-20: 1/s^2 = inf) gives NaN / inf / 0 exactly where the reference's
                // gradient does (tests/test_extreme_scales.py): e.g. a zero partial sum over s = 1e-40 is 0/s = 0, not 0 * (1/s) = NaN
                float *d = dwhere + 4 * ((size_t)sp * n + k);
                const float ax = 1.0f / sx, bx = (-tx) / sx, ay = 1.0f / sy, by = (-ty) / sy;
                d[0] = -(r0 * (ax / sx)) - r1 * (bx / sx);
                d[1] = -(r1 / sx);
                d[2] = -(r2 * (ay / sy)) - r3 * (by / sy);
                d[3] = -(r3 / sy);
                if (dpresence) dpresence[(size_t)sp * n + k] = r4;
            }
        }
        AIR_TR(4);
    }
    AIR_TR_FLUSH();
}
template <bool RC>
__global__ __launch_bounds__(1024) void canvas_bwd_kernel(WriteBwdArgs a, NvilArgs nv) {
    extern __shared__ __align__(16) float smem[];
    canvas_bwd_body<RC>(a, nv, smem, (int)blockIdx.x, (int)gridDim.x);
}
// throughput regime: small workgroups (one or two waves per unit), many resident per CU
template <bool RC>
__global__ __launch_bounds__(128, 4) void canvas_bwd_small_kernel(WriteBwdArgs a, NvilArgs nv) {
    extern __shared__ __align__(16) float smem[];
    canvas_bwd_body<RC>(a, nv, smem, (int)blockIdx.x, (int)gridDim.x);
}
__global__ __launch_bounds__(256, 2) void canvas_fwd_small_kernel(WriteFwdArgs a) {
    extern __shared__ __align__(16) float smem[];
    canvas_fwd_body(a, smem, (int)blockIdx.x, (int)gridDim.x);
}
// Canvas forward and backward of a train step in ONE launch (latency regime).  The recompute form of the backward reads nothing
// the forward writes, so the two are independent roles of one grid: workgroups [0, n_fwd) run the forward (image x row band:
// per-step canvases, final canvas, reconstruction shares), the rest the recompute-form backward (NS per glimpse).  NVIL -- which
// needs the forward's reconstruction shares -- rides on a later launch (air_gauss_sample_bwd_nvil).
__global__ __launch_bounds__(1024) void canvas_fused_kernel(WriteFwdArgs f, WriteBwdArgs b, int n_fwd) {
    extern __shared__ __align__(16) float smem[];
    if ((int)blockIdx.x < n_fwd) canvas_fwd_body(f, smem, (int)blockIdx.x, n_fwd);
    else {
        const NvilArgs none = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1, nullptr};
        canvas_bwd_body<true>(b, none, smem, (int)blockIdx.x - n_fwd, (int)gridDim.x - n_fwd);
    }
}

// ============================================================================================================
// host side
// ============================================================================================================
static inline double lin_step(int n) { return n > 1 ? 2.0 / (double)(n - 1) : 0.0; }
static inline int cv_grid(long items, int cap) { return (int)(items < cap ? items : cap); }
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
// Workgroup shapes.  Latency regime (the launch does not fill the chip): many waves per unit, each with a row or two, so the
// unit's chain is short; throughput regime: few waves per unit, many units resident per CU (the kernels are VALU-issue bound:
// what matters there is that every SIMD always has a wave to issue from).
static inline int fwd_threads(long units, int RB) {
    if (units > 512) return 256;
    int nw = RB < 16 ? RB : 16;
    if (units > 256 && nw > 8) nw = 8;
    return 64 * (nw < 1 ? 1 : nw);
}
static inline int bwd_threads(long units_x_ns, int h) {
    if (units_x_ns > 1024) return 64;
    if (units_x_ns > 512) return 128;
    int nw = h < 8 ? h : 8;
    return 64 * (nw < 1 ? 1 : nw);
}
static int launch_write_fwd(const float *glimpse, const float *where, const float *presence, const float *canvas_in,
                            const float *obs, float *canvas_steps, float *final_canvas, float *rec_parts, int n_bands,
                            int T, int B, int H, int W, int h, int w, float mult, float std, void *stream) {
    int NB, RB;
    wr_bands(H, n_bands, &NB, &RB);
    AIR_REQUIRE(NB == n_bands || !rec_parts, AIR_E_SHAPE);   // the caller sized rec_parts for exactly n_bands shares
    const size_t lds = carve_fwd_bytes(T, RB, W, h, w);
    AIR_REQUIRE(lds <= CV_MAX_LDS, AIR_E_UNSUPPORTED);
    const int vec4g = (w % 4 == 0) && air_aligned16(glimpse);   // 16-byte groups that never straddle a glimpse row
    const long units = (long)B * NB;
    const int threads = fwd_threads(units, RB);
    const WriteFwdArgs a = {glimpse, where, presence, canvas_in, obs, canvas_steps, final_canvas, rec_parts, T, B, NB, RB, H, W, h, w,
                            lin_step(W), lin_step(H), mult, std, vec4g};
    if (threads <= 256) {
        { int st_ = cv_allow_lds(canvas_fwd_small_kernel, lds); if (st_) return st_; }
        hipLaunchKernelGGL(canvas_fwd_small_kernel, dim3(cv_grid(units, 256 * 8)), dim3(threads), lds, air_stream(stream), a);
    } else {
        { int st_ = cv_allow_lds(canvas_fwd_kernel, lds); if (st_) return st_; }
        hipLaunchKernelGGL(canvas_fwd_kernel, dim3(cv_grid(units, 256 * 8)), dim3(threads), lds, air_stream(stream), a);
    }
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
    NvilArgs nv = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1, nullptr};
    if (nvil) nv = *nvil;
    const long units = (long)T * B;
    const int threads = bwd_threads(units, h), nw = threads / 64;
    const size_t lds = carve_bwd_bytes(H, W, h, w, rc ? T : 1, rc, nw, nw);
    AIR_REQUIRE(lds <= CV_MAX_LDS, AIR_E_UNSUPPORTED);
    const int vec4g = (w % 4 == 0) && air_aligned16(glimpse);
    const int vec4c = ((H * W) % 4 == 0) && air_aligned16(obs);
    const WriteBwdArgs a = {glimpse, where, presence, dcanvas, final_canvas, obs, dglimpse, dwhere, dpresence, T, B, H, W, h, w,
                            lin_step(W), lin_step(H), mult, std, loss_scale, vec4g, vec4c, 1};
    const int grid = cv_grid(units, 256 * 16) + (nvil ? 1 : 0);
    // (the NVIL rider is one wave-0 chain with a workgroup broadcast: any workgroup size serves it)
    if (threads <= 128) {
        if (rc) {
            { int st_ = cv_allow_lds(canvas_bwd_small_kernel<true>, lds); if (st_) return st_; }
            hipLaunchKernelGGL(canvas_bwd_small_kernel<true>, dim3(grid), dim3(threads), lds, air_stream(stream), a, nv);
        } else {
            { int st_ = cv_allow_lds(canvas_bwd_small_kernel<false>, lds); if (st_) return st_; }
            hipLaunchKernelGGL(canvas_bwd_small_kernel<false>, dim3(grid), dim3(threads), lds, air_stream(stream), a, nv);
        }
    } else if (rc) {
        { int st_ = cv_allow_lds(canvas_bwd_kernel<true>, lds); if (st_) return st_; }
        hipLaunchKernelGGL(canvas_bwd_kernel<true>, dim3(grid), dim3(threads), lds, air_stream(stream), a, nv);
    } else {
        { int st_ = cv_allow_lds(canvas_bwd_kernel<false>, lds); if (st_) return st_; }
        hipLaunchKernelGGL(canvas_bwd_kernel<false>, dim3(grid), dim3(threads), lds, air_stream(stream), a, nv);
    }
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

extern "C" int air_canvas_unroll_bwd_nvil(const float *glimpse, const float *where, const float *presence,
                                          const float *obs, const float *final_canvas, float *dglimpse, float *dwhere,
                                          int T, int B, int H, int W, int h, int w, float mult, float std,
                                          float loss_scale, const float *imp_parts, int n_parts, float *imp_sum,
                                          const float *baseline, const float *logp, float *nvil_out, float *dlogp,
                                          float *dbaseline, void *stream) {
    AIR_REQUIRE(glimpse && where && obs && dglimpse && dwhere, AIR_E_NULL);      // final_canvas == NULL: the recompute form
    AIR_REQUIRE(imp_parts && baseline && logp && nvil_out, AIR_E_NULL);
    AIR_REQUIRE(T > 0 && n_parts > 0, AIR_E_SHAPE);
    int st = cv_check_dims(B, H, W, h, w);
    if (st) return st;
    const NvilArgs nv = {imp_parts, baseline, logp, nvil_out, dlogp, dbaseline, B, n_parts, imp_sum};
    return launch_write_bwd(glimpse, where, presence, nullptr, final_canvas, obs, dglimpse, dwhere, nullptr, T, B, H, W,
                            h, w, mult, std, loss_scale, stream, &nv);
}

// The fused launch's shape for a problem: threads per workgroup, and whether it fits (LDS of both roles, both grids small
// enough to run side by side).  n_split = workgroups per backward unit (1 or 2).
static int fused_shape(int n_bands, int n_split, int T, int B, int H, int W, int h, int w, int *threads, size_t *lds) {
    int NB, RB;
    wr_bands(H, n_bands, &NB, &RB);
    if (NB != n_bands) return AIR_E_SHAPE;
    if (n_split < 1 || n_split > 4) return AIR_E_SHAPE;
    if ((long)B * NB > 4096 || (long)B * T * n_split > 4096) return AIR_E_UNSUPPORTED;
    const int nt = bwd_threads((long)B * T * n_split, h) < 512 ? bwd_threads((long)B * T * n_split, h) : 512;
    const int nw = nt / 64;
    const size_t lds_f = carve_fwd_bytes(T, RB, W, h, w), lds_b = carve_bwd_bytes(H, W, h, w, T, true, nw, n_split * nw);
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
    { int st_ = cv_allow_lds(canvas_fused_kernel, lds); if (st_) return st_; }
    const WriteFwdArgs f = {glimpse, where, presence, nullptr, obs, canvas_steps, final_canvas, rec_parts, T, B, NB, RB, H, W, h, w,
                            lin_step(W), lin_step(H), mult, std, vec4g};
    const WriteBwdArgs b = {glimpse, where, presence, nullptr, nullptr, obs, dglimpse, dwhere, nullptr, T, B, H, W, h, w,
                            lin_step(W), lin_step(H), mult, std, loss_scale, vec4g, vec4c, n_split};
    const int n_fwd = B * NB;
    hipLaunchKernelGGL(canvas_fused_kernel, dim3(n_fwd + T * B * n_split), dim3(threads), lds, air_stream(stream), f, b, n_fwd);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
