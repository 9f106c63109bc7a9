// Device-side bodies shared by the stand-alone kernels (pointwise_kernels.hip / loss_kernels.hip) and by the fused
// "heads" launches of the engine (engine_kernels.hip): reparameterised Gaussian (+KL) rows and the register-resident
// number-of-steps posterior.  Every body takes a VIRTUAL block index / grid size so that two independent bodies can
// share one launch (the train step is launch bound at batch 64).
#pragma once
#include <math.h>
#include "air_common.h"

#ifndef PW_THREADS
#define PW_THREADS 256
#endif

// float64 log as ONE out-of-line copy: inlined, each call is ~1.5 KB of straight-line code, and in the train step every kernel
// starts with a cold instruction cache (36 different kernels per step share 64 KB per CU pair), so code size is latency
__device__ __noinline__ double air_log_f64(double x) { return log(x); }

// pm = NaN: the prior is centred on the posterior's own mean (a where-shift prior given without `loc`, model.py:203-207): the
// (mu - pm)^2 term and its gradient vanish
__device__ __forceinline__ float kl_mean_diff(float mu, float pm) { return pm == pm ? mu - pm : 0.f; }
__device__ __forceinline__ float normal_kl(float mu, float s, float pm, float ps) {
    const float ratio = (s * s) / (ps * ps);
    const float d = kl_mean_diff(mu, pm);
    return d * d / (2.f * ps * ps) + 0.5f * (ratio - 1.f - logf(ratio));
}
// d KL / d scale, evaluated in the order the reference's automatic differentiation walks the expression above (TF 1.1
// _kl_normal_normal, model.py:188-214: ratio = s^2 / ps^2; LogGrad = g * 1/ratio; RealDivGrad; SquareGrad = g * 2s):
//     g_ratio = .5 dk - (.5 dk) / ratio,   ds = (g_ratio / ps^2) * 2 s.
// Equal to dk * (s / ps^2 - 1 / s) wherever `ratio` is a normal number -- and non-finite exactly where the reference's gradient
// is: -inf once s^2 underflows (s < 3.7e-23: the KL row itself is +inf there), NaN for s == 0.  The closed form would stay
// finite down to s = 1e-38 and hide the blow-up the reference's arithmetic has (match, don't clamp).
__device__ __forceinline__ float normal_kl_dscale(float dk, float s, float ps) {
#pragma clang fp contract(off)
    const float ps2 = ps * ps;
    const float ratio = (s * s) / ps2;
    const float gr = 0.5f * dk - (0.5f * dk) / ratio;
    return (gr / ps2) * (2.f * s);
}
// The raw-scale offset of a Gaussian head together with the optional numerical guard (default off: the reference's arithmetic,
// inf / NaN placement included -- tests/test_extreme_scales.py).  guard > 0 (AIRModel(guard_degenerate=...)):
//   * scale = max(softplus(raw + offset), guard): the KL rows of model.py:188-214 stay finite when a scale head wanders towards
//     sigma^2 underflow (the gradient through a floored scale is 0);
//   * transform heads (loc_mode 1): the SAMPLED scale components of `where` keep |s| >= guard (sign kept, +guard for 0;
//     straight-through), so the inverse warp's 1/s (modules.py:101-102) never divides by an exact zero.
struct RawOffset {
    float v, guard;
    __host__ __device__ RawOffset(float v_ = 0.f, float g_ = 0.f) : v(v_), guard(g_) {}
};
__device__ __forceinline__ float guard_scale(float s, float guard) { return (guard > 0.f && s < guard) ? guard : s; }
__device__ __forceinline__ float guard_where(float v, int d, int loc_mode, float guard) {
    return (guard > 0.f && loc_mode == 1 && !(d & 1) && fabsf(v) < guard) ? copysignf(guard, v) : v;
}
// one 64-lane wave per row: D elements strided over lanes, KL reduced with a wave reduction
__device__ __forceinline__ void gauss_fwd_body(int vblock, int vgrid, const float *__restrict__ pre, int ld_pre,
                                                               const float *__restrict__ eps, RawOffset raw_offset,
                                                               int loc_mode, float pl0, float ps0, float pl1, float ps1,
                                                               float *__restrict__ loc, float *__restrict__ scale,
                                                               float *__restrict__ sample, float *__restrict__ kl_row,
                                                               int M, int D, float *__restrict__ sample_bm = nullptr,
                                                               int B_bm = 1, int ld_bm = 0) {
    // sample_bm (optional): a second, batch-major copy of the sample -- row m = t*B_bm + b lands at
    // sample_bm[b*ld_bm + t*D + d] (the `what` columns of the baseline input, modules.py:131-139)
    const int lane = threadIdx.x & 63;
    const int wave_global = (int)((vblock * (size_t)PW_THREADS + threadIdx.x) >> 6);
    const int nwaves = (vgrid * PW_THREADS) >> 6;
    for (int m = wave_global; m < M; m += nwaves) {
        const float *pr = pre + (size_t)m * ld_pre;
        float kl = 0.f;
        for (int d = lane; d < D; d += 64) {
            float mu = pr[d];
            if (loc_mode == 1) mu = (d & 1) ? tanhf(mu) : sigmoid_acc(mu);
            const float s = guard_scale(softplus_acc(pr[D + d] + raw_offset.v), raw_offset.guard);
            const size_t o = (size_t)m * D + d;
            loc[o] = mu; scale[o] = s;
            if (sample) {
                const float v = guard_where(mu + s * eps[o], d, loc_mode, raw_offset.guard);
                sample[o] = v;
                if (sample_bm) { const int t = m / B_bm, b = m - t * B_bm; sample_bm[(size_t)b * ld_bm + t * D + d] = v; }
            }
            kl += (d & 1) ? normal_kl(mu, s, pl1, ps1) : normal_kl(mu, s, pl0, ps0);
        }
        kl = wave_sum(kl);
        if (kl_row && lane == 0) kl_row[m] = kl;
    }
}
// One (row, dim) element of a Gaussian head's backward: d pre_loc, d pre_raw from the sample gradient `ds` (has_ds), the row's KL
// weight `dk`, the stored loc / scale and the raw scale (+ offset).  ONE function with contraction off, so that the stand-alone
// launch (gauss_bwd_body) and the GEMM epilogue that folds it (gemm_kernels.hip, air_gemm_grouped_gauss_bwd) give the same bits.
__device__ __forceinline__ void gauss_bwd_elem(float ds, bool has_ds, float eps, float mu, float s, float pm, float ps, float dk,
                                               float raw, int loc_mode, int d, float guard, float &dloc, float &draw) {
#pragma clang fp contract(off)
    float dmu = ds + dk * kl_mean_diff(mu, pm) / (ps * ps);
    const float dsc = (has_ds ? ds * eps : 0.f) + normal_kl_dscale(dk, s, ps);
    if (loc_mode == 1) dmu *= (d & 1) ? (1.f - mu * mu) : mu * (1.f - mu);
    float dsp = raw > 20.f ? 1.f : sigmoid_acc(raw);                // d softplus
    if (guard > 0.f && s <= guard) dsp = 0.f;                       // a floored scale passes no gradient
    dloc = dmu;
    draw = dsc * dsp;
}
// The KL rows of a head whose forward left them as per-tile shares (air_what_head_fwd: kl_parts[n_parts][M]): a later launch of the
// same head adds the shares in tile order into kl_row_out[M] with ONE extra workgroup (the row sums are only consumed further down
// the backward chain and by the read-outs).
struct KlParts { const float *parts; float *out; int n_parts; };
__device__ __forceinline__ void kl_parts_sum(const KlParts &kp, int M) {
    for (int m = threadIdx.x; m < M; m += blockDim.x) {
        float s = kp.parts[m];
        for (int p = 1; p < kp.n_parts; ++p) s += kp.parts[(size_t)p * M + m];
        kp.out[m] = s;
    }
}
__device__ __forceinline__ void gauss_bwd_body(int vblock, int vgrid, const float *__restrict__ pre, int ld_pre,
                                                               const float *__restrict__ eps, RawOffset raw_offset,
                                                               int loc_mode, float pl0, float ps0, float pl1, float ps1,
                                                               const float *__restrict__ loc,
                                                               const float *__restrict__ scale,
                                                               const float *__restrict__ dsample,
                                                               const float *__restrict__ dsample2,
                                                               const float *__restrict__ dkl_row, float dkl_scale,
                                                               float *__restrict__ dpre, int ld_dpre, int M, int D) {
    const size_t n = (size_t)M * D;
    for (size_t e = (size_t)vblock * PW_THREADS + threadIdx.x; e < n; e += (size_t)vgrid * PW_THREADS) {
        const size_t m = e / D;
        const int d = (int)(e - m * D);
        const float mu = loc[e], s = scale[e];
        const float pm = (d & 1) ? pl1 : pl0, ps = (d & 1) ? ps1 : ps0;
        const float ds = (dsample ? dsample[e] : 0.f) + (dsample2 ? dsample2[e] : 0.f);
        const float dk = dkl_row ? dkl_row[m] * dkl_scale : 0.f;
        float dloc, draw;
        gauss_bwd_elem(ds, dsample || dsample2, (dsample || dsample2) ? eps[e] : 0.f, mu, s, pm, ps, dk,
                       pre[m * ld_pre + D + d] + raw_offset.v, loc_mode, d, raw_offset.guard, dloc, draw);
        dpre[m * ld_dpre + d] = dloc;
        dpre[m * ld_dpre + D + d] = draw;
    }
}

// Baseline input assembly (modules.py:131-139): out[B, P + T*A + T*4 + T + S0 + S1] = [img | what | where | presence | h | c]
// from time-major what / where / presence; only columns >= c_begin are written (c_begin = P + T*A when the `what`
// columns are filled by the sampling kernel itself).
__device__ __forceinline__ void baseline_pack_body(int vblock, int vgrid, const float *__restrict__ img,
                                                   const float *__restrict__ what, const float *__restrict__ where,
                                                   const float *__restrict__ presence, const float *__restrict__ s0,
                                                   const float *__restrict__ s1, float *__restrict__ out, int T, int B,
                                                   int P, int A, int S0, int S1, int c_begin) {
    const int width = P + T * A + T * 4 + T + S0 + S1, span = width - c_begin;
    const size_t n = (size_t)B * span;
    for (size_t e = (size_t)vblock * PW_THREADS + threadIdx.x; e < n; e += (size_t)vgrid * PW_THREADS) {
        const size_t b = e / span;
        const int col = c_begin + (int)(e - b * span);
        int c = col;
        float v;
        if (c < P) v = img[b * P + c];
        else if ((c -= P) < T * A) { const int t = c / A, a = c - t * A; v = what[((size_t)t * B + b) * A + a]; }
        else if ((c -= T * A) < T * 4) { const int t = c / 4, a = c - t * 4; v = where[((size_t)t * B + b) * 4 + a]; }
        else if ((c -= T * 4) < T) v = presence[(size_t)c * B + b];
        else if ((c -= T) < S0) v = s0[b * S0 + c];
        else v = s1[b * S1 + (c - S0)];
        out[b * width + col] = v;
    }
}

template <int MT>
struct NumStepsR {
    double p[MT], P[MT + 1], q[MT + 1], S;   // p_t, prefix products P[n] = prod_{j<n} p_j, posterior q(n), normaliser
    float q32[MT + 1];
};
// posterior from the T presence probabilities held in registers
template <int MT>
__device__ __forceinline__ void posterior_p(const float (&p32)[MT], int T, NumStepsR<MT> &s) {
    s.P[0] = 1.0;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        s.p[t] = t < T ? (double)p32[t] : 1.0;
        s.P[t + 1] = s.P[t] * s.p[t];
    }
    double u[MT + 1];
    s.S = 0.0;
#pragma unroll
    for (int n = 0; n <= MT; ++n) {
        u[n] = n < T ? (1.0 - s.p[n < MT ? n : 0]) * s.P[n] : (n == T ? s.P[n] : 0.0);
        if (n <= T) s.S += u[n];
    }
#pragma unroll
    for (int n = 0; n <= MT; ++n) { s.q[n] = n <= T ? u[n] / s.S : 0.0; s.q32[n] = (float)s.q[n]; }
}
template <int MT>
__device__ __forceinline__ void posterior_r(const float *__restrict__ prob, int T, int B, int b, NumStepsR<MT> &s) {
    float p32[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) p32[t] = t < T ? prob[(size_t)t * B + b] : 1.0f;
    posterior_p<MT>(p32, T, s);
}

// presence (cell.py:137-151) + q(n) / KL / step weights / log q(n_sampled) in one launch
// One batch column b with its T logits and uniform variates already in registers: nothing is read back from memory, so the
// whole chain (sigmoid, Bernoulli chain, float64 posterior, KL, step weights, log q(n*)) is pure ALU after the operand loads.
template <int MT>
__device__ __forceinline__ void presence_numsteps_col(int b, const float (&lg)[MT], const float (&uu)[MT],
    const double (&pri)[MT + 1], float step_bias, float eps, float *__restrict__ prob, float *__restrict__ pres,
    float *__restrict__ q, float *__restrict__ kl_ps, float *__restrict__ logp, float *__restrict__ step_w, int T, int B,
    bool discrete = true) {
    float run = 1.0f, nsteps = 0.f, p32[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        p32[t] = 1.0f;
        if (t < T) {
            const size_t k = (size_t)t * B + b;
            float p = 1.0f / (1.0f + expf(-(lg[t] + step_bias)));
            if (eps >= 0.f) p = eps / 2 + (1 - eps) * p;
            p32[t] = p;
            prob[k] = p;
            // cell.py:147-151: the Bernoulli chain, or (discrete_steps=False; the callers pass no uniform variates) the probability itself
            run = discrete ? run * ((uu[t] < p) ? 1.0f : 0.0f) : p;
            pres[k] = run;
            nsteps += run;
        }
    }
    NumStepsR<MT> s;
    posterior_p<MT>(p32, T, s);
    float kl = 0.f, w = 0.f, qstar = 0.f;
    const int nstar = (int)nsteps;
#pragma unroll
    for (int n = 0; n <= MT; ++n) {
        if (n <= T) {
            q[(size_t)b * (T + 1) + n] = s.q32[n];
            const double pn = (double)s.q32[n];
            kl += (pn > 0.0) ? (float)(pn * air_log_f64(pn / pri[n])) : 0.f;
            if (n == nstar) qstar = s.q32[n];
        }
    }
    kl_ps[b] = kl;
#pragma unroll
    for (int t = MT - 1; t >= 0; --t) {
        if (t < T) { w += s.q32[t + 1]; step_w[(size_t)t * B + b] = w; }
    }
    logp[b] = logf(fmaxf(qstar, 1e-32f));
}

// presence (cell.py:137-151) + q(n) / KL / step weights / log q(n_sampled) in one launch
template <int MT>
__device__ __forceinline__ void presence_numsteps_fwd_body(int vblock, int vgrid,
    const float *__restrict__ logit, const float *__restrict__ u, float step_bias, float eps,
    const double *__restrict__ prior, float *__restrict__ prob, float *__restrict__ pres, float *__restrict__ q,
    float *__restrict__ kl_ps, float *__restrict__ logp, float *__restrict__ step_w, int T, int B) {
    for (int b = vblock * 64 + (int)threadIdx.x; b < B; b += vgrid * 64) {
        if (threadIdx.x >= 64) break;
        float lg[MT], uu[MT];
        double pri[MT + 1];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            lg[t] = t < T ? logit[(size_t)t * B + b] : 0.f;
            uu[t] = (t < T && u) ? u[(size_t)t * B + b] : 0.f;
        }
#pragma unroll
        for (int n = 0; n <= MT; ++n) pri[n] = n <= T ? prior[n] : 1.0;
        presence_numsteps_col<MT>(b, lg, uu, pri, step_bias, eps, prob, pres, q, kl_ps, logp, step_w, T, B, u != nullptr);
    }
}

// backward of the above wrt the steps-predictor logit: the step-weight gradient is formed in place from the two
// per-row KL buffers (dstep_w[t,b] = w_scale * (kl_a[t,b] + kl_b[t,b])), then d/dq -> d/du -> d/dp (products only, no
// divisions: safe at p = 0 like the reference's scan-based cumprod), then sigmoid'.
template <int MT>
__device__ __forceinline__ void numsteps_presence_bwd_col(int b,
    const float *__restrict__ prob, const float *__restrict__ presence, const double *__restrict__ prior,
    float kl_scale, const float *__restrict__ kl_a, const float *__restrict__ kl_b, float w_scale,
    const float *__restrict__ dlogp, const float *__restrict__ logit, float step_bias, float eps,
    float (&dl)[MT], int T, int B, const float *__restrict__ dpres = nullptr) {
    NumStepsR<MT> s;
    posterior_r<MT>(prob, T, B, b, s);
    int nstar = -1;
    if (dlogp) {
        float ns = 0.f;
#pragma unroll
        for (int t = 0; t < MT; ++t) if (t < T) ns += presence[(size_t)t * B + b];
        nstar = (int)ns;
    }
    double gq[MT + 1], wsum = 0.0, dot = 0.0;
#pragma unroll
    for (int n = 0; n <= MT; ++n) {
        double g = 0.0;
        if (n <= T) {
            const double pn = (double)s.q32[n];
            g = (pn > 0.0) ? (double)kl_scale * (air_log_f64(pn / prior[n]) + 1.0) : 0.0;
            if (n >= 1) {
                const size_t k = (size_t)(n - 1) * B + b;
                wsum += (double)(w_scale * ((kl_a ? kl_a[k] : 0.f) + (kl_b ? kl_b[k] : 0.f)));
            }
            g += wsum;
            if (n == nstar) g += (double)dlogp[b] / (double)fmaxf(s.q32[n], 1e-32f);
            dot += g * s.q[n];
        }
        gq[n] = g;
    }
    double gu[MT + 1];
#pragma unroll
    for (int n = 0; n <= MT; ++n) gu[n] = n <= T ? (gq[n] - dot) / s.S : 0.0;
#pragma unroll
    for (int k = 0; k < MT; ++k) {
        dl[k] = 0.f;
        if (k < T) {
            // u_k = (1-p_k) P[k];  u_n (n>k, n<T) = (1-p_n) P[k] p_k R_n with R_n = prod_{k<j<n} p_j;  u_T = P[k] p_k R_T
            double g = -gu[k] * s.P[k];
            double R = 1.0;
#pragma unroll
            for (int n = k + 1; n <= MT; ++n) {
                if (n < T) g += gu[n] * (1.0 - s.p[n < MT ? n : 0]) * s.P[k] * R;
                else if (n == T) g += gu[n] * s.P[k] * R;
                if (n < MT) R *= s.p[n];
            }
            const size_t idx = (size_t)k * B + b;
            const float sg = 1.0f / (1.0f + expf(-(logit[idx] + step_bias)));
            float gg = (float)g;
            // discrete_steps=False (cell.py:150-151): presence IS the probability, so what the canvas write's backward holds for it joins here
            if (dpres) gg += dpres[idx];
            if (eps >= 0.f) gg *= (1 - eps);
            dl[k] = gg * sg * (1.f - sg);
        }
    }
}
template <int MT>
__device__ __forceinline__ void numsteps_presence_bwd_body(int vblock, int vgrid,
    const float *__restrict__ prob, const float *__restrict__ presence, const double *__restrict__ prior,
    float kl_scale, const float *__restrict__ kl_a, const float *__restrict__ kl_b, float w_scale,
    const float *__restrict__ dlogp, const float *__restrict__ logit, float step_bias, float eps,
    float *__restrict__ dlogit, int T, int B, const float *__restrict__ dpres = nullptr) {
    for (int b = vblock * 64 + (int)threadIdx.x; b < B; b += vgrid * 64) {
        if (threadIdx.x >= 64) break;
        float dl[MT];
        numsteps_presence_bwd_col<MT>(b, prob, presence, prior, kl_scale, kl_a, kl_b, w_scale, dlogp, logit, step_bias, eps, dl,
                                      T, B, dpres);
#pragma unroll
        for (int k = 0; k < MT; ++k) if (k < T) dlogit[(size_t)k * B + b] = dl[k];
    }
}
