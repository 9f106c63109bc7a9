// Pointwise / row-wise kernels of the AIR step for gfx950: LSTM gates, reparameterised Gaussians (+KL), presence,
// reconstruction log-likelihood, baseline-input packing, centred RMSProp, Philox noise.  All are HBM/latency bound
// elementwise passes: one coalesced read of every input, one coalesced write of every output, fp32 math.
#include <math.h>
#include "air_common.h"

#include "optimizer_device.h"
#define PW_THREADS 256
static inline int pw_blocks(size_t n) {
    size_t b = (n + PW_THREADS - 1) / PW_THREADS;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}
#define PW_LOOP(i, n) for (size_t i = (size_t)blockIdx.x * PW_THREADS + threadIdx.x; i < (n); i += (size_t)gridDim.x * PW_THREADS)

// ---- LSTM pointwise (Sonnet v1 LSTM, gate order i,j,f,o; cell.py:126-127) ---------------------------------------
__global__ __launch_bounds__(PW_THREADS) void lstm_pw_fwd_kernel(const float *__restrict__ gates,
                                                                 const float *__restrict__ c_prev,
                                                                 float *__restrict__ h, float *__restrict__ c,
                                                                 float *__restrict__ gate_act, int M, int Hd, float fb) {
    const size_t n = (size_t)M * Hd;
    PW_LOOP(e, n) {
        const size_t m = e / Hd, u = e - m * Hd;
        const float *gr = gates + m * 4 * (size_t)Hd;
        const float gi = sigmoid_acc(gr[u]);
        const float gj = tanhf(gr[Hd + u]);
        const float gf = sigmoid_acc(gr[2 * (size_t)Hd + u] + fb);
        const float go = sigmoid_acc(gr[3 * (size_t)Hd + u]);
        const float cn = gf * c_prev[e] + gi * gj;
        c[e] = cn;
        h[e] = tanhf(cn) * go;
        if (gate_act) {
            float *ar = gate_act + m * 4 * (size_t)Hd;
            ar[u] = gi; ar[Hd + u] = gj; ar[2 * (size_t)Hd + u] = gf; ar[3 * (size_t)Hd + u] = go;
        }
    }
}
__device__ __forceinline__ void lstm_pw_bwd_body(const float *__restrict__ gate_act, const float *__restrict__ c_prev,
                                                 const float *__restrict__ c, const float *__restrict__ dh,
                                                 const float *__restrict__ dh2, const float *__restrict__ dc_in,
                                                 float *__restrict__ dgates, float *__restrict__ dc_prev, int M, int Hd,
                                                 int vblock, int vgrid, unsigned short *__restrict__ dgates16 = nullptr) {
    const size_t n = (size_t)M * Hd;
    for (size_t e = (size_t)vblock * PW_THREADS + threadIdx.x; e < n; e += (size_t)vgrid * PW_THREADS) {
        const size_t m = e / Hd, u = e - m * Hd;
        const float *ar = gate_act + m * 4 * (size_t)Hd;
        const float gi = ar[u], gj = ar[Hd + u], gf = ar[2 * (size_t)Hd + u], go = ar[3 * (size_t)Hd + u];
        const float tc = tanhf(c[e]);
        const float dhe = (dh ? dh[e] : 0.f) + (dh2 ? dh2[e] : 0.f);
        const float dct = (dc_in ? dc_in[e] : 0.f) + dhe * go * (1.f - tc * tc);
        float *dr = dgates + m * 4 * (size_t)Hd;
        dr[u] = dct * gj * gi * (1.f - gi);
        dr[Hd + u] = dct * gi * (1.f - gj * gj);
        dr[2 * (size_t)Hd + u] = dct * c_prev[e] * gf * (1.f - gf);
        dr[3 * (size_t)Hd + u] = dhe * tc * go * (1.f - go);
        dc_prev[e] = dct * gf;
        if (dgates16) {                                       // bf16 mirror of dgates (bf16 data path: the weight gradient and the
            unsigned short *d16 = dgates16 + m * 4 * (size_t)Hd;                        // next BPTT link read it)
            d16[u] = __builtin_bit_cast(unsigned short, (__bf16)dr[u]);
            d16[Hd + u] = __builtin_bit_cast(unsigned short, (__bf16)dr[Hd + u]);
            d16[2 * (size_t)Hd + u] = __builtin_bit_cast(unsigned short, (__bf16)dr[2 * (size_t)Hd + u]);
            d16[3 * (size_t)Hd + u] = __builtin_bit_cast(unsigned short, (__bf16)dr[3 * (size_t)Hd + u]);
        }
    }
}
__global__ __launch_bounds__(PW_THREADS) void lstm_pw_bwd_mirror_kernel(const float *__restrict__ gate_act,
                                                                        const float *__restrict__ c_prev,
                                                                        const float *__restrict__ c,
                                                                        const float *__restrict__ dh,
                                                                        const float *__restrict__ dh2,
                                                                        const float *__restrict__ dc_in,
                                                                        float *__restrict__ dgates,
                                                                        float *__restrict__ dc_prev, int M, int Hd,
                                                                        unsigned short *__restrict__ dgates16) {
    lstm_pw_bwd_body(gate_act, c_prev, c, dh, dh2, dc_in, dgates, dc_prev, M, Hd, blockIdx.x, gridDim.x, dgates16);
}
__global__ __launch_bounds__(PW_THREADS) void lstm_pw_bwd_kernel(const float *__restrict__ gate_act,
                                                                 const float *__restrict__ c_prev,
                                                                 const float *__restrict__ c,
                                                                 const float *__restrict__ dh,
                                                                 const float *__restrict__ dh2,
                                                                 const float *__restrict__ dc_in,
                                                                 float *__restrict__ dgates,
                                                                 float *__restrict__ dc_prev, int M, int Hd) {
    lstm_pw_bwd_body(gate_act, c_prev, c, dh, dh2, dc_in, dgates, dc_prev, M, Hd, blockIdx.x, gridDim.x);
}
// the same with an optimiser slice on workgroups [main_blocks, gridDim.x)
__global__ __launch_bounds__(PW_THREADS) void lstm_pw_bwd_opt_kernel(const float *__restrict__ gate_act,
                                                                     const float *__restrict__ c_prev,
                                                                     const float *__restrict__ c,
                                                                     const float *__restrict__ dh,
                                                                     const float *__restrict__ dh2,
                                                                     const float *__restrict__ dc_in,
                                                                     float *__restrict__ dgates,
                                                                     float *__restrict__ dc_prev, int M, int Hd,
                                                                     int main_blocks, RmspropSlice opt) {
    if ((int)blockIdx.x >= main_blocks) {
        rmsprop_slice_body(opt, (int)blockIdx.x - main_blocks, (int)gridDim.x - main_blocks);
        return;
    }
    lstm_pw_bwd_body(gate_act, c_prev, c, dh, dh2, dc_in, dgates, dc_prev, M, Hd, blockIdx.x, main_blocks);
}
extern "C" int air_lstm_pointwise_fwd(const float *gates, const float *c_prev, float *h, float *c, float *gate_act,
                                      int M, int Hd, float forget_bias, void *stream) {
    AIR_REQUIRE(gates && c_prev && h && c, AIR_E_NULL);
    AIR_REQUIRE(M > 0 && Hd > 0, AIR_E_SHAPE);
    hipLaunchKernelGGL(lstm_pw_fwd_kernel, dim3(pw_blocks((size_t)M * Hd)), dim3(PW_THREADS), 0, air_stream(stream),
                       gates, c_prev, h, c, gate_act, M, Hd, forget_bias);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
extern "C" int air_lstm_pointwise_bwd(const float *gate_act, const float *c_prev, const float *c, const float *dh,
                                      const float *dh2, const float *dc, float *dgates, float *dc_prev, int M, int Hd,
                                      void *stream) {
    AIR_REQUIRE(gate_act && c_prev && c && dgates && dc_prev, AIR_E_NULL);
    AIR_REQUIRE(dh || dc, AIR_E_NULL);
    AIR_REQUIRE(M > 0 && Hd > 0, AIR_E_SHAPE);
    hipLaunchKernelGGL(lstm_pw_bwd_kernel, dim3(pw_blocks((size_t)M * Hd)), dim3(PW_THREADS), 0, air_stream(stream),
                       gate_act, c_prev, c, dh, dh2, dc, dgates, dc_prev, M, Hd);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
extern "C" int air_lstm_pointwise_bwd_bf16(const float *gate_act, const float *c_prev, const float *c, const float *dh,
                                           const float *dh2, const float *dc, float *dgates, void *dgates_bf16, float *dc_prev,
                                           int M, int Hd, void *stream) {
    AIR_REQUIRE(gate_act && c_prev && c && dgates && dc_prev && dgates_bf16, AIR_E_NULL);
    AIR_REQUIRE(dh || dc, AIR_E_NULL);
    AIR_REQUIRE(M > 0 && Hd > 0, AIR_E_SHAPE);
    hipLaunchKernelGGL(lstm_pw_bwd_mirror_kernel, dim3(pw_blocks((size_t)M * Hd)), dim3(PW_THREADS), 0, air_stream(stream),
                       gate_act, c_prev, c, dh, dh2, dc, dgates, dc_prev, M, Hd, (unsigned short *)dgates_bf16);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
extern "C" int air_lstm_pointwise_bwd_opt(const float *gate_act, const float *c_prev, const float *c, const float *dh,
                                          const float *dh2, const float *dc, float *dgates, float *dc_prev, int M, int Hd,
                                          const AirRmspropSlice *opt, void *stream) {
    RmspropSlice s; size_t nq;
    int st = rmsprop_slice_from_abi(opt, s, &nq);
    if (st) return st;
    if (nq == 0) return air_lstm_pointwise_bwd(gate_act, c_prev, c, dh, dh2, dc, dgates, dc_prev, M, Hd, stream);
    AIR_REQUIRE(gate_act && c_prev && c && dgates && dc_prev, AIR_E_NULL);
    AIR_REQUIRE(dh || dc, AIR_E_NULL);
    AIR_REQUIRE(M > 0 && Hd > 0, AIR_E_SHAPE);
    const int main_blocks = pw_blocks((size_t)M * Hd);
    size_t extra = air_rider_blocks(nq, PW_THREADS, 768);                  // about two float4 per thread
    hipLaunchKernelGGL(lstm_pw_bwd_opt_kernel, dim3(main_blocks + (int)extra), dim3(PW_THREADS), 0, air_stream(stream), gate_act,
                       c_prev, c, dh, dh2, dc, dgates, dc_prev, M, Hd, main_blocks, s);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// ---- reparameterised Gaussian + KL (cell.py:130-133,154-156; modules.py:17-24,41-46,58-63; model.py:174-209) ----
#include "engine_device.h"
#include "nvil_device.h"
__global__ __launch_bounds__(PW_THREADS) void gauss_fwd_kernel(const float *__restrict__ pre, int ld_pre,
                                                               const float *__restrict__ eps, RawOffset raw_offset,
                                                               int loc_mode, float pl0, float ps0, float pl1, float ps1,
                                                               float *__restrict__ loc, float *__restrict__ scale,
                                                               float *__restrict__ sample, float *__restrict__ kl_row,
                                                               int M, int D) {
    gauss_fwd_body(blockIdx.x, gridDim.x, pre, ld_pre, eps, raw_offset, loc_mode, pl0, ps0, pl1, ps1, loc, scale, sample,
                   kl_row, M, D);
}
__global__ __launch_bounds__(PW_THREADS) void gauss_bwd_kernel(const float *__restrict__ pre, int ld_pre,
                                                               const float *__restrict__ eps, RawOffset raw_offset,
                                                               int loc_mode, float pl0, float ps0, float pl1, float ps1,
                                                               const float *__restrict__ loc,
                                                               const float *__restrict__ scale,
                                                               const float *__restrict__ dsample,
                                                               const float *__restrict__ dsample2,
                                                               const float *__restrict__ dkl_row, float dkl_scale,
                                                               float *__restrict__ dpre, int ld_dpre, int M, int D, KlParts kp) {
    if (kp.n_parts > 0 && blockIdx.x == gridDim.x - 1) { kl_parts_sum(kp, M); return; }
    const int vg = kp.n_parts > 0 ? gridDim.x - 1 : gridDim.x;
    gauss_bwd_body(blockIdx.x, vg, pre, ld_pre, eps, raw_offset, loc_mode, pl0, ps0, pl1, ps1, loc, scale, dsample,
                   dsample2, dkl_row, dkl_scale, dpre, ld_dpre, M, D);
}
// the same launch with ONE extra workgroup (the first of the grid) that evaluates the NVIL objective (nvil_body): the two are
// independent, and in the step whose canvas forward + backward are one launch (air_canvas_unroll_fwd_bwd) this is the first
// launch behind the reconstruction shares NVIL needs that has room for a rider
__global__ __launch_bounds__(PW_THREADS) void gauss_bwd_nvil_kernel(const float *__restrict__ pre, int ld_pre,
                                                                    const float *__restrict__ eps, RawOffset raw_offset,
                                                                    int loc_mode, float pl0, float ps0, float pl1, float ps1,
                                                                    const float *__restrict__ loc,
                                                                    const float *__restrict__ scale,
                                                                    const float *__restrict__ dsample,
                                                                    const float *__restrict__ dsample2,
                                                                    const float *__restrict__ dkl_row, float dkl_scale,
                                                                    float *__restrict__ dpre, int ld_dpre, int M, int D, NvilArgs nv,
                                                                    KlParts kp) {
    if (blockIdx.x == 0) { nvil_body(nv); return; }
    if (kp.n_parts > 0 && blockIdx.x == gridDim.x - 1) { kl_parts_sum(kp, M); return; }
    const int vg = (kp.n_parts > 0 ? gridDim.x - 1 : gridDim.x) - 1;
    gauss_bwd_body(blockIdx.x - 1, vg, pre, ld_pre, eps, raw_offset, loc_mode, pl0, ps0, pl1, ps1, loc, scale, dsample,
                   dsample2, dkl_row, dkl_scale, dpre, ld_dpre, M, D);
}
extern "C" int air_gauss_sample_bwd_nvil(const float *pre, int ld_pre, const float *eps, float raw_offset, int loc_mode,
                                         float p_loc_even, float p_scale_even, float p_loc_odd, float p_scale_odd,
                                         const float *loc, const float *scale, const float *dsample,
                                         const float *dsample2, const float *dkl_row, float dkl_scale, float *dpre,
                                         int ld_dpre, int M, int D, const float *imp_parts, int n_parts, float *imp_sum,
                                         const float *baseline, const float *logp, float *nvil_out, float *dlogp,
                                         float *dbaseline, int B, float guard_eps, float *ema_dev, const float *kl_parts, int n_kl_parts,
                                         float *kl_row_out, void *stream) {
    AIR_REQUIRE(pre && loc && scale && dpre, AIR_E_NULL);
    AIR_REQUIRE(n_kl_parts >= 0 && (n_kl_parts == 0 || (kl_parts && kl_row_out)), AIR_E_NULL);
    AIR_REQUIRE(!(dsample || dsample2) || eps, AIR_E_NULL);
    AIR_REQUIRE(M > 0 && D > 0 && ld_pre >= 2 * D && ld_dpre >= 2 * D, AIR_E_SHAPE);
    AIR_REQUIRE(imp_parts && baseline && logp && nvil_out, AIR_E_NULL);
    AIR_REQUIRE(B > 0 && n_parts > 0, AIR_E_SHAPE);
    const NvilArgs nv = {imp_parts, baseline, logp, nvil_out, dlogp, dbaseline, B, n_parts, imp_sum, ema_dev};
    const KlParts kp = {kl_parts, kl_row_out, n_kl_parts};
    hipLaunchKernelGGL(gauss_bwd_nvil_kernel, dim3(pw_blocks((size_t)M * D) + 1 + (n_kl_parts > 0 ? 1 : 0)), dim3(PW_THREADS), 0,
                       air_stream(stream), pre, ld_pre, eps, RawOffset(raw_offset, guard_eps), loc_mode, p_loc_even, p_scale_even,
                       p_loc_odd, p_scale_odd, loc, scale, dsample, dsample2, dkl_row, dkl_scale, dpre, ld_dpre, M, D, nv, kp);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
extern "C" int air_gauss_sample_fwd(const float *pre, int ld_pre, const float *eps, float raw_offset, int loc_mode,
                                    float p_loc_even, float p_scale_even, float p_loc_odd, float p_scale_odd,
                                    float *loc, float *scale, float *sample, float *kl_row, int M, int D,
                                    float guard_eps, void *stream) {
    AIR_REQUIRE(pre && loc && scale, AIR_E_NULL);
    AIR_REQUIRE(!sample || eps, AIR_E_NULL);
    AIR_REQUIRE(M > 0 && D > 0 && ld_pre >= 2 * D, AIR_E_SHAPE);
    AIR_REQUIRE(loc_mode == 0 || loc_mode == 1, AIR_E_UNSUPPORTED);
    hipLaunchKernelGGL(gauss_fwd_kernel, dim3(pw_blocks((size_t)M * 64)), dim3(PW_THREADS), 0, air_stream(stream), pre,
                       ld_pre, eps, RawOffset(raw_offset, guard_eps), loc_mode, p_loc_even, p_scale_even, p_loc_odd, p_scale_odd, loc, scale,
                       sample, kl_row, M, D);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
extern "C" int air_gauss_sample_bwd(const float *pre, int ld_pre, const float *eps, float raw_offset, int loc_mode,
                                    float p_loc_even, float p_scale_even, float p_loc_odd, float p_scale_odd,
                                    const float *loc, const float *scale, const float *dsample,
                                    const float *dsample2, const float *dkl_row, float dkl_scale, float *dpre,
                                    int ld_dpre, int M, int D, float guard_eps, const float *kl_parts, int n_kl_parts,
                                    float *kl_row_out, void *stream) {
    AIR_REQUIRE(pre && loc && scale && dpre, AIR_E_NULL);
    AIR_REQUIRE(n_kl_parts >= 0 && (n_kl_parts == 0 || (kl_parts && kl_row_out)), AIR_E_NULL);
    AIR_REQUIRE(!(dsample || dsample2) || eps, AIR_E_NULL);
    AIR_REQUIRE(M > 0 && D > 0 && ld_pre >= 2 * D && ld_dpre >= 2 * D, AIR_E_SHAPE);
    const KlParts kp = {kl_parts, kl_row_out, n_kl_parts};
    hipLaunchKernelGGL(gauss_bwd_kernel, dim3(pw_blocks((size_t)M * D) + (n_kl_parts > 0 ? 1 : 0)), dim3(PW_THREADS), 0,
                       air_stream(stream), pre, ld_pre, eps, RawOffset(raw_offset, guard_eps), loc_mode, p_loc_even, p_scale_even,
                       p_loc_odd, p_scale_odd, loc, scale, dsample, dsample2, dkl_row, dkl_scale, dpre, ld_dpre, M, D, kp);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

__global__ __launch_bounds__(PW_THREADS) void normal_kl_fwd_kernel(const float *__restrict__ loc,
                                                                   const float *__restrict__ scale, float pl0,
                                                                   float ps0, float pl1, float ps1,
                                                                   float *__restrict__ kl_row, int M, int D) {
    const int lane = threadIdx.x & 63;
    const int wave_global = (int)((blockIdx.x * (size_t)PW_THREADS + threadIdx.x) >> 6);
    const int nwaves = (gridDim.x * PW_THREADS) >> 6;
    for (int m = wave_global; m < M; m += nwaves) {
        float kl = 0.f;
        for (int d = lane; d < D; d += 64) {
            const size_t o = (size_t)m * D + d;
            kl += (d & 1) ? normal_kl(loc[o], scale[o], pl1, ps1) : normal_kl(loc[o], scale[o], pl0, ps0);
        }
        kl = wave_sum(kl);
        if (lane == 0) kl_row[m] = kl;
    }
}
__global__ __launch_bounds__(PW_THREADS) void normal_kl_bwd_kernel(const float *__restrict__ loc,
                                                                   const float *__restrict__ scale, float pl0,
                                                                   float ps0, float pl1, float ps1,
                                                                   const float *__restrict__ dkl,
                                                                   float *__restrict__ dloc,
                                                                   float *__restrict__ dscale, int M, int D) {
    PW_LOOP(e, (size_t)M * D) {
        const size_t m = e / D;
        const int d = (int)(e - m * D);
        const float pm = (d & 1) ? pl1 : pl0, ps = (d & 1) ? ps1 : ps0;
        const float g = dkl[m], s = scale[e];
        dloc[e] = g * kl_mean_diff(loc[e], pm) / (ps * ps);
        dscale[e] = normal_kl_dscale(g, s, ps);
    }
}
extern "C" int air_normal_kl_fwd(const float *loc, const float *scale, float p_loc_even, float p_scale_even,
                                 float p_loc_odd, float p_scale_odd, float *kl_row, int M, int D, void *stream) {
    AIR_REQUIRE(loc && scale && kl_row, AIR_E_NULL);
    AIR_REQUIRE(M > 0 && D > 0, AIR_E_SHAPE);
    hipLaunchKernelGGL(normal_kl_fwd_kernel, dim3(pw_blocks((size_t)M * 64)), dim3(PW_THREADS), 0, air_stream(stream),
                       loc, scale, p_loc_even, p_scale_even, p_loc_odd, p_scale_odd, kl_row, M, D);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
extern "C" int air_normal_kl_bwd(const float *loc, const float *scale, float p_loc_even, float p_scale_even,
                                 float p_loc_odd, float p_scale_odd, const float *dkl_row, float *dloc, float *dscale,
                                 int M, int D, void *stream) {
    AIR_REQUIRE(loc && scale && dkl_row && dloc && dscale, AIR_E_NULL);
    AIR_REQUIRE(M > 0 && D > 0, AIR_E_SHAPE);
    hipLaunchKernelGGL(normal_kl_bwd_kernel, dim3(pw_blocks((size_t)M * D)), dim3(PW_THREADS), 0, air_stream(stream),
                       loc, scale, p_loc_even, p_scale_even, p_loc_odd, p_scale_odd, dkl_row, dloc, dscale, M, D);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// ---- presence (cell.py:137-151) ---------------------------------------------------------------------------------
__global__ __launch_bounds__(PW_THREADS) void presence_fwd_kernel(const float *__restrict__ logit,
                                                                  const float *__restrict__ u,
                                                                  const float *__restrict__ presence_in,
                                                                  float step_bias, float eps, int discrete,
                                                                  float *__restrict__ prob, float *__restrict__ pres,
                                                                  int T, int B) {
    PW_LOOP(b, (size_t)B) {
        float run = presence_in ? presence_in[b] : 1.0f;
        for (int t = 0; t < T; ++t) {
            const size_t k = (size_t)t * B + b;
            float p = sigmoid_acc(logit[k] + step_bias);
            if (eps >= 0.f) p = eps / 2 + (1 - eps) * p;
            prob[k] = p;
            if (discrete) {
                run *= (u[k] < p) ? 1.0f : 0.0f;
                pres[k] = run;
            } else {
                pres[k] = p;
            }
        }
    }
}
__global__ __launch_bounds__(PW_THREADS) void presence_bwd_kernel(const float *__restrict__ logit, float step_bias,
                                                                  float eps, int discrete,
                                                                  const float *__restrict__ dprob,
                                                                  const float *__restrict__ dpres,
                                                                  float *__restrict__ dlogit, size_t n) {
    PW_LOOP(k, n) {
        const float s = sigmoid_acc(logit[k] + step_bias);
        float g = dprob ? dprob[k] : 0.f;
        if (!discrete && dpres) g += dpres[k];
        if (eps >= 0.f) g *= (1 - eps);
        dlogit[k] = g * s * (1.f - s);
    }
}
extern "C" int air_presence_fwd(const float *logit, const float *u, const float *presence_in, float step_bias,
                                float explore_eps, int discrete, float *presence_prob, float *presence, int T, int B,
                                void *stream) {
    AIR_REQUIRE(logit && presence_prob && presence, AIR_E_NULL);
    AIR_REQUIRE(!discrete || u, AIR_E_NULL);
    AIR_REQUIRE(T > 0 && B > 0, AIR_E_SHAPE);
    hipLaunchKernelGGL(presence_fwd_kernel, dim3(pw_blocks(B)), dim3(PW_THREADS), 0, air_stream(stream), logit, u,
                       presence_in, step_bias, explore_eps, discrete, presence_prob, presence, T, B);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
extern "C" int air_presence_bwd(const float *logit, float step_bias, float explore_eps, int discrete,
                                const float *dpresence_prob, const float *dpresence, float *dlogit, int T, int B,
                                void *stream) {
    AIR_REQUIRE(logit && dlogit, AIR_E_NULL);
    AIR_REQUIRE(T > 0 && B > 0, AIR_E_SHAPE);
    hipLaunchKernelGGL(presence_bwd_kernel, dim3(pw_blocks((size_t)T * B)), dim3(PW_THREADS), 0, air_stream(stream),
                       logit, step_bias, explore_eps, discrete, dpresence_prob, dpresence, dlogit, (size_t)T * B);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// ---- reconstruction term (model.py:319-324) ---------------------------------------------------------------------
__global__ __launch_bounds__(PW_THREADS) void rec_fwd_kernel(const float *__restrict__ obs,
                                                             const float *__restrict__ canvas, float mult, float std,
                                                             float *__restrict__ per_sample, int B, int P) {
    __shared__ float scratch[8];
    const float cst = 0.5f * logf(6.283185307179586f) + logf(std);
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const float *x = obs + (size_t)b * P, *c = canvas + (size_t)b * P;
        float s[1] = {0.f};
        for (int p = threadIdx.x; p < P; p += PW_THREADS) {
            const float z = (x[p] - mult * c[p]) / std;
            s[0] += 0.5f * z * z + cst;
        }
        __syncthreads();
        block_sum<1>(s, scratch);
        if (threadIdx.x == 0) per_sample[b] = s[0];
    }
}
__global__ __launch_bounds__(PW_THREADS) void rec_bwd_kernel(const float *__restrict__ obs,
                                                             const float *__restrict__ canvas, float mult, float std,
                                                             const float *__restrict__ dps, float scale,
                                                             float *__restrict__ dcanvas, int B, int P) {
    const size_t n = (size_t)B * P;
    const float k = mult / (std * std);
    PW_LOOP(e, n) {
        const float g = dps ? dps[e / P] : scale;
        dcanvas[e] = g * k * (mult * canvas[e] - obs[e]);
    }
}
extern "C" int air_rec_loglik_fwd(const float *obs, const float *canvas, float mult, float std, float *per_sample,
                                  int B, int P, void *stream) {
    AIR_REQUIRE(obs && canvas && per_sample, AIR_E_NULL);
    AIR_REQUIRE(B > 0 && P > 0, AIR_E_SHAPE);
    hipLaunchKernelGGL(rec_fwd_kernel, dim3(B < 2048 ? B : 2048), dim3(PW_THREADS), 0, air_stream(stream), obs, canvas,
                       mult, std, per_sample, B, P);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
extern "C" int air_rec_loglik_bwd(const float *obs, const float *canvas, float mult, float std,
                                  const float *dper_sample, float scale, float *dcanvas, int B, int P, void *stream) {
    AIR_REQUIRE(obs && canvas && dcanvas, AIR_E_NULL);
    AIR_REQUIRE(B > 0 && P > 0, AIR_E_SHAPE);
    hipLaunchKernelGGL(rec_bwd_kernel, dim3(pw_blocks((size_t)B * P)), dim3(PW_THREADS), 0, air_stream(stream), obs,
                       canvas, mult, std, dper_sample, scale, dcanvas, B, P);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// ---- baseline input packing (modules.py:131-139) ----------------------------------------------------------------
__global__ __launch_bounds__(PW_THREADS) void baseline_pack_kernel(const float *__restrict__ img,
                                                                   const float *__restrict__ what,
                                                                   const float *__restrict__ where,
                                                                   const float *__restrict__ presence,
                                                                   const float *__restrict__ s0,
                                                                   const float *__restrict__ s1,
                                                                   float *__restrict__ out, int T, int B, int P, int A,
                                                                   int S0, int S1) {
    baseline_pack_body(blockIdx.x, gridDim.x, img, what, where, presence, s0, s1, out, T, B, P, A, S0, S1, 0);
}
extern "C" int air_baseline_pack(const float *img, const float *what, const float *where, const float *presence,
                                 const float *state0, const float *state1, float *out, int T, int B, int P, int A,
                                 int S0, int S1, void *stream) {
    AIR_REQUIRE(what && where && presence && out, AIR_E_NULL);
    AIR_REQUIRE((P == 0 || img) && (S0 == 0 || state0) && (S1 == 0 || state1), AIR_E_NULL);
    AIR_REQUIRE(T > 0 && B > 0 && P >= 0 && A > 0 && S0 >= 0 && S1 >= 0, AIR_E_SHAPE);
    const size_t n = (size_t)B * (P + T * A + T * 4 + T + S0 + S1);
    hipLaunchKernelGGL(baseline_pack_kernel, dim3(pw_blocks(n)), dim3(PW_THREADS), 0, air_stream(stream), img, what,
                       where, presence, state0, state1, out, T, B, P, A, S0, S1);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// ---- RMSProp with momentum, centred or not (TF semantics; model.py:265,355-367) ---------------------------------
// The centred instantiation is the literal expression of rmsprop_elem (optimizer_device.h) -- the riders and the closing
// update must agree with it bit for bit; the plain form keeps the mg slot up to date but never reads it
template <bool CENTRED>
__global__ __launch_bounds__(PW_THREADS) void rmsprop_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                             float *__restrict__ ms, float *__restrict__ mg,
                                                             float *__restrict__ mom, size_t n,
                                                             const float *__restrict__ lr_dev, float lr_mult,
                                                             float decay, float momentum, float eps, float gscale) {
    const float lr = lr_dev[0] * lr_mult;
    PW_LOOP(i, n) {
        if (CENTRED) {
            float pv = p[i], a = ms[i], b = mg[i], c = mom[i];
            rmsprop_elem(pv, g[i], a, b, c, lr, decay, momentum, eps, gscale);
            ms[i] = a; mg[i] = b; mom[i] = c; p[i] = pv;
        } else {
            const float gi = g[i] * gscale;
            const float msi = decay * ms[i] + (1.f - decay) * gi * gi;
            const float mgi = decay * mg[i] + (1.f - decay) * gi;
            const float mo = momentum * mom[i] + lr * gi / sqrtf(msi + eps);
            ms[i] = msi; mg[i] = mgi; mom[i] = mo;
            p[i] -= mo;
        }
    }
}
extern "C" int air_rmsprop(float *p, const float *g, float *ms, float *mg, float *mom, size_t n, const float *lr_dev,
                           float lr_mult, float decay, float momentum, float eps, int centered, float grad_scale,
                           void *stream) {
    AIR_REQUIRE(p && g && ms && mg && mom && lr_dev, AIR_E_NULL);
    AIR_REQUIRE(n > 0, AIR_E_SHAPE);
    if (centered)
        hipLaunchKernelGGL(rmsprop_kernel<true>, dim3(pw_blocks(n)), dim3(PW_THREADS), 0, air_stream(stream), p, g, ms, mg, mom,
                           n, lr_dev, lr_mult, decay, momentum, eps, grad_scale);
    else
        hipLaunchKernelGGL(rmsprop_kernel<false>, dim3(pw_blocks(n)), dim3(PW_THREADS), 0, air_stream(stream), p, g, ms, mg, mom,
                           n, lr_dev, lr_mult, decay, momentum, eps, grad_scale);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
extern "C" int air_rmsprop_centered(float *p, const float *g, float *ms, float *mg, float *mom, size_t n,
                                    const float *lr_dev, float lr_mult, float decay, float momentum, float eps,
                                    float grad_scale, void *stream) {
    return air_rmsprop(p, g, ms, mg, mom, n, lr_dev, lr_mult, decay, momentum, eps, 1, grad_scale, stream);
}

// ---- Philox4x32-10 noise ----------------------------------------------------------------------------------------
#include "prologue_device.h"

__global__ __launch_bounds__(PW_THREADS) void rng_fill_kernel(float *__restrict__ normal, size_t n_normal,
                                                              float *__restrict__ uniform, size_t n_uniform,
                                                              const uint64_t *__restrict__ state) {
    const uint64_t seed = state[0], offset = state[1];
    const size_t q_normal = (n_normal + 3) / 4, q_uniform = (n_uniform + 3) / 4;
    PW_LOOP(q, q_normal + q_uniform) {
        uint32_t r[4];
        philox4x32(offset + q, 0, seed, r);
        if (q < q_normal) {
            // Box-Muller: two pairs -> four normals
            float z[4];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float rad = sqrtf(-2.0f * logf(u01_open(r[2 * k])));
                float sn, cs;
                sincosf(6.283185307179586f * u01(r[2 * k + 1]), &sn, &cs);
                z[2 * k] = rad * cs; z[2 * k + 1] = rad * sn;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) if (4 * q + k < n_normal) normal[4 * q + k] = z[k];
        } else {
            const size_t qq = q - q_normal;
#pragma unroll
            for (int k = 0; k < 4; ++k) if (4 * qq + k < n_uniform) uniform[4 * qq + k] = u01(r[k]);
        }
    }
}
__global__ void rng_advance_kernel(uint64_t *state, uint64_t inc) {
    if (threadIdx.x == 0 && blockIdx.x == 0) state[1] += inc;
}
extern "C" int air_rng_fill(float *normal, size_t n_normal, float *uniform, size_t n_uniform,
                            const uint64_t *state_dev, void *stream) {
    AIR_REQUIRE(state_dev, AIR_E_NULL);
    AIR_REQUIRE((n_normal == 0 || normal) && (n_uniform == 0 || uniform), AIR_E_NULL);
    const size_t q = (n_normal + 3) / 4 + (n_uniform + 3) / 4;
    if (q == 0) return AIR_OK;
    hipLaunchKernelGGL(rng_fill_kernel, dim3(pw_blocks(q)), dim3(PW_THREADS), 0, air_stream(stream), normal, n_normal,
                       uniform, n_uniform, state_dev);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
extern "C" int air_rng_advance(uint64_t *state_dev, uint64_t increment, void *stream) {
    AIR_REQUIRE(state_dev, AIR_E_NULL);
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, air_stream(stream), state_dev, increment);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// ---- fused step prologue / epilogue (the train step is launch bound: 6 tiny launches become 2) -------------------
// prologue: Philox noise for the whole step + annealed geometric prior (float64) + tiling of the trainable LSTM
//           initial state over the batch.  Roles are split by block index.
__global__ __launch_bounds__(PW_THREADS) void step_prologue_kernel(PrologueArgs a) {
    step_prologue_body(a, blockIdx.x, gridDim.x);
}
extern "C" int air_step_prologue(float *normal, size_t n_normal, float *uniform, size_t n_uniform,
                                 const uint64_t *rng_state_dev, const int64_t *global_step_dev, int anneal_type,
                                 double init, double final_value, double anneal_steps, double hold_for,
                                 double steps_div, double *prior_out_f64, int T, const float *h0, const float *c0,
                                 float *h_tiled, float *c_tiled, int B, int Hd, void *stream) {
    AIR_REQUIRE(rng_state_dev && global_step_dev && prior_out_f64 && h0 && c0 && h_tiled && c_tiled, AIR_E_NULL);
    AIR_REQUIRE((n_normal == 0 || normal) && (n_uniform == 0 || uniform), AIR_E_NULL);
    AIR_REQUIRE(T > 0 && B > 0 && Hd > 0 && anneal_type >= 0 && anneal_type <= 2, AIR_E_SHAPE);
    const PrologueArgs a = make_prologue_args(normal, n_normal, uniform, n_uniform, rng_state_dev, global_step_dev,
                                              anneal_type, init, final_value, anneal_steps, hold_for, steps_div,
                                              prior_out_f64, T, h0, c0, h_tiled, c_tiled, B, Hd);
    hipLaunchKernelGGL(step_prologue_kernel, dim3(prologue_blocks(a)), dim3(PW_THREADS), 0, air_stream(stream), a);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
// the prologue with the bf16 conversion of the observation batch riding as extra workgroups (bf16 data path, round 5: the two
// independent tiny launches that opened the batch-1024 step are one)
__global__ __launch_bounds__(PW_THREADS) void step_prologue_cvt_kernel(PrologueArgs a, int pro_blocks, const float4 *__restrict__ x,
                                                                       uint2 *__restrict__ out, size_t nq) {
    if ((int)blockIdx.x < pro_blocks) { step_prologue_body(a, blockIdx.x, pro_blocks); return; }
    const size_t vb = blockIdx.x - pro_blocks, vg = gridDim.x - pro_blocks;
    for (size_t q = vb * PW_THREADS + threadIdx.x; q < nq; q += vg * PW_THREADS) {
        const float4 v = x[q];
        const unsigned lo = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v.x) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v.y) << 16);
        const unsigned hi = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v.z) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v.w) << 16);
        out[q] = make_uint2(lo, hi);
    }
}
extern "C" int air_step_prologue_cvt(float *normal, size_t n_normal, float *uniform, size_t n_uniform,
                                     const uint64_t *rng_state_dev, const int64_t *global_step_dev, int anneal_type,
                                     double init, double final_value, double anneal_steps, double hold_for,
                                     double steps_div, double *prior_out_f64, int T, const float *h0, const float *c0,
                                     float *h_tiled, float *c_tiled, int B, int Hd, const float *x, void *x_bf16, size_t n_x,
                                     void *stream) {
    AIR_REQUIRE(rng_state_dev && global_step_dev && prior_out_f64 && h0 && c0 && h_tiled && c_tiled && x && x_bf16, AIR_E_NULL);
    AIR_REQUIRE((n_normal == 0 || normal) && (n_uniform == 0 || uniform), AIR_E_NULL);
    AIR_REQUIRE(T > 0 && B > 0 && Hd > 0 && anneal_type >= 0 && anneal_type <= 2 && n_x > 0 && n_x % 4 == 0, AIR_E_SHAPE);
    AIR_REQUIRE(air_aligned16(x) && ((uintptr_t)x_bf16 % 8 == 0), AIR_E_ALIGN);
    const PrologueArgs a = make_prologue_args(normal, n_normal, uniform, n_uniform, rng_state_dev, global_step_dev,
                                              anneal_type, init, final_value, anneal_steps, hold_for, steps_div,
                                              prior_out_f64, T, h0, c0, h_tiled, c_tiled, B, Hd);
    const int pb = prologue_blocks(a);
    hipLaunchKernelGGL(step_prologue_cvt_kernel, dim3(pb + pw_blocks(n_x >> 2)), dim3(PW_THREADS), 0, air_stream(stream), a, pb,
                       reinterpret_cast<const float4 *>(x), reinterpret_cast<uint2 *>(x_bf16), n_x >> 2);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// ... and with the HBM feeder attached (round 6): the rows the conversion reads come straight from the resident dataset -- row b = item
// idx_b, idx_b drawn exactly as air_batch_gather draws it -- and the workgroup writes BOTH the fp32 batch every later launch reads and its
// bf16 mirror: the gather launch that opened the throughput-regime step (7 us at batch 1024) is gone.  One workgroup per row at a time.
struct GatherRows {
    const float *data; long long n_items; int item_floats, shuffle, B;
    const uint64_t *seed; const int64_t *step; float *obs; int64_t *idx_out;
};
__global__ __launch_bounds__(PW_THREADS) void step_prologue_gather_cvt_kernel(PrologueArgs a, int pro_blocks, GatherRows gr, uint2 *__restrict__ out16) {
    if ((int)blockIdx.x < pro_blocks) { step_prologue_body(a, blockIdx.x, pro_blocks); return; }
    const int vb = (int)blockIdx.x - pro_blocks, vg = (int)gridDim.x - pro_blocks;
    const long long step = gr.step[0];
    const int nq = gr.item_floats >> 2;
    for (int b = vb; b < gr.B; b += vg) {
        const unsigned long long ctr = (unsigned long long)step * (unsigned long long)gr.B + (unsigned long long)b;
        long long idx;
        if (gr.shuffle) {
            uint32_t r[4];
            philox4x32(ctr, 1, gr.seed[0], r);
            const unsigned long long wide = ((unsigned long long)r[0] << 32) | r[1];
            idx = (long long)(((unsigned __int128)wide * (unsigned __int128)gr.n_items) >> 64);     // uniform in [0, n): as batch_gather_kernel
        } else {
            idx = (long long)(ctr % (unsigned long long)gr.n_items);
        }
        if (threadIdx.x == 0 && gr.idx_out) gr.idx_out[b] = idx;
        const float4 *src = reinterpret_cast<const float4 *>(gr.data + (size_t)idx * gr.item_floats);
        float4 *dst = reinterpret_cast<float4 *>(gr.obs + (size_t)b * gr.item_floats);
        uint2 *d16 = out16 + (size_t)b * nq;
        for (int q = threadIdx.x; q < nq; q += PW_THREADS) {
            const float4 v = src[q];
            dst[q] = v;
            const unsigned lo = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v.x) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v.y) << 16);
            const unsigned hi = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v.z) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v.w) << 16);
            d16[q] = make_uint2(lo, hi);
        }
    }
}
extern "C" int air_step_prologue_gather_cvt(float *normal, size_t n_normal, float *uniform, size_t n_uniform,
                                            const uint64_t *rng_state_dev, const int64_t *global_step_dev, int anneal_type,
                                            double init, double final_value, double anneal_steps, double hold_for,
                                            double steps_div, double *prior_out_f64, int T, const float *h0, const float *c0,
                                            float *h_tiled, float *c_tiled, int B, int Hd, const AirBatchGather *bg, void *x_bf16,
                                            void *stream) {
    AIR_REQUIRE(rng_state_dev && global_step_dev && prior_out_f64 && h0 && c0 && h_tiled && c_tiled && bg && x_bf16, AIR_E_NULL);
    AIR_REQUIRE(bg->dataset && bg->seed_dev && bg->step_dev && bg->obs, AIR_E_NULL);
    AIR_REQUIRE((n_normal == 0 || normal) && (n_uniform == 0 || uniform), AIR_E_NULL);
    AIR_REQUIRE(T > 0 && B > 0 && Hd > 0 && anneal_type >= 0 && anneal_type <= 2, AIR_E_SHAPE);
    AIR_REQUIRE(bg->n_items > 0 && bg->item_floats > 0 && bg->item_floats % 4 == 0 && bg->B > 0, AIR_E_SHAPE);
    AIR_REQUIRE(air_aligned16(bg->dataset) && air_aligned16(bg->obs) && ((uintptr_t)x_bf16 % 8 == 0), AIR_E_ALIGN);
    const PrologueArgs a = make_prologue_args(normal, n_normal, uniform, n_uniform, rng_state_dev, global_step_dev,
                                              anneal_type, init, final_value, anneal_steps, hold_for, steps_div,
                                              prior_out_f64, T, h0, c0, h_tiled, c_tiled, B, Hd);
    const int pb = prologue_blocks(a);
    const GatherRows gr = {bg->dataset, bg->n_items, bg->item_floats, bg->shuffle ? 1 : 0, bg->B, bg->seed_dev, bg->step_dev, bg->obs, bg->idx_out};
    hipLaunchKernelGGL(step_prologue_gather_cvt_kernel, dim3(pb + (bg->B < 4096 ? bg->B : 4096)), dim3(PW_THREADS), 0, air_stream(stream), a, pb, gr,
                       reinterpret_cast<uint2 *>(x_bf16));
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// epilogue: both centred-RMSProp updates (model segment [0, n_model) at lr, baseline segment at lr * lr_mult_tail) in
//           one pass over the flat buffers, then the device counters (global step, Philox offset) advance.
// 16-byte accesses: the pass moves 9 x 4 B per parameter (94 MB at the 50x50 configuration) and is bound by memory-pipe
// instructions, not arithmetic; the segment boundary n_model and every tensor start are multiples of 4 floats by construction
// of the flat layout, so one learning rate applies to a whole float4.  VEC = false covers unaligned / odd-sized buffers.
template <bool VEC>
__global__ __launch_bounds__(PW_THREADS) void step_epilogue_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                                   float *__restrict__ ms, float *__restrict__ mg,
                                                                   float *__restrict__ mom, size_t n_model, size_t n_total,
                                                                   const float *__restrict__ lr_dev, float lr_mult_tail,
                                                                   float decay, float momentum, float eps, float gscale,
                                                                   int64_t *__restrict__ gstep, uint64_t *__restrict__ rng_state,
                                                                   uint64_t rng_inc, unsigned short *__restrict__ p16) {
    const float lr0 = lr_dev[0];
    if (VEC) {
        float4 *p4 = reinterpret_cast<float4 *>(p), *ms4 = reinterpret_cast<float4 *>(ms), *mg4 = reinterpret_cast<float4 *>(mg),
               *mom4 = reinterpret_cast<float4 *>(mom);
        const float4 *g4 = reinterpret_cast<const float4 *>(g);
        PW_LOOP(q, n_total >> 2) {
            const float lr = (q << 2) < n_model ? lr0 : lr0 * lr_mult_tail;
            float4 pv = p4[q], gv = g4[q], a = ms4[q], b = mg4[q], c = mom4[q];
            rmsprop_elem(pv.x, gv.x, a.x, b.x, c.x, lr, decay, momentum, eps, gscale);
            rmsprop_elem(pv.y, gv.y, a.y, b.y, c.y, lr, decay, momentum, eps, gscale);
            rmsprop_elem(pv.z, gv.z, a.z, b.z, c.z, lr, decay, momentum, eps, gscale);
            rmsprop_elem(pv.w, gv.w, a.w, b.w, c.w, lr, decay, momentum, eps, gscale);
            ms4[q] = a; mg4[q] = b; mom4[q] = c; p4[q] = pv;
            if (p16) {       // bf16 shadow of the weights (bf16 data path): the next step's products read it, the fp32 copy stays the master
                const unsigned lo = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)pv.x) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)pv.y) << 16);
                const unsigned hi = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)pv.z) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)pv.w) << 16);
                reinterpret_cast<uint2 *>(p16)[q] = make_uint2(lo, hi);
            }
        }
    } else {
        PW_LOOP(i, n_total) {
            const float lr = i < n_model ? lr0 : lr0 * lr_mult_tail;
            float pv = p[i], a = ms[i], b = mg[i], c = mom[i];
            rmsprop_elem(pv, g[i], a, b, c, lr, decay, momentum, eps, gscale);
            ms[i] = a; mg[i] = b; mom[i] = c; p[i] = pv;
            if (p16) p16[i] = __builtin_bit_cast(unsigned short, (__bf16)pv);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (gstep) gstep[0] += 1;
        if (rng_state) rng_state[1] += rng_inc;
    }
}
extern "C" int air_step_epilogue_shadow(float *p, const float *g, float *ms, float *mg, float *mom, size_t n_model,
                                        size_t n_total, const float *lr_dev, float lr_mult_tail, float decay, float momentum,
                                        float eps, float grad_scale, int64_t *global_step_dev, uint64_t *rng_state_dev,
                                        uint64_t rng_increment, void *p_bf16, void *stream);
extern "C" int air_step_epilogue(float *p, const float *g, float *ms, float *mg, float *mom, size_t n_model,
                                 size_t n_total, const float *lr_dev, float lr_mult_tail, float decay, float momentum,
                                 float eps, float grad_scale, int64_t *global_step_dev, uint64_t *rng_state_dev,
                                 uint64_t rng_increment, void *stream) {
    return air_step_epilogue_shadow(p, g, ms, mg, mom, n_model, n_total, lr_dev, lr_mult_tail, decay, momentum, eps, grad_scale,
                                    global_step_dev, rng_state_dev, rng_increment, nullptr, stream);
}
extern "C" int air_step_epilogue_shadow(float *p, const float *g, float *ms, float *mg, float *mom, size_t n_model,
                                        size_t n_total, const float *lr_dev, float lr_mult_tail, float decay, float momentum,
                                        float eps, float grad_scale, int64_t *global_step_dev, uint64_t *rng_state_dev,
                                        uint64_t rng_increment, void *p_bf16, void *stream) {
    AIR_REQUIRE(p && g && ms && mg && mom && lr_dev, AIR_E_NULL);
    AIR_REQUIRE(n_total > 0 && n_model <= n_total, AIR_E_SHAPE);
    unsigned short *p16 = (unsigned short *)p_bf16;
    AIR_REQUIRE(!p16 || ((uintptr_t)p16 % 8 == 0), AIR_E_ALIGN);
    const bool vec = (n_total % 4 == 0) && (n_model % 4 == 0) && air_aligned16(p) && air_aligned16(g) && air_aligned16(ms) &&
                     air_aligned16(mg) && air_aligned16(mom);
    if (vec)
        hipLaunchKernelGGL(step_epilogue_kernel<true>, dim3(pw_blocks(n_total >> 2)), dim3(PW_THREADS), 0, air_stream(stream), p,
                           g, ms, mg, mom, n_model, n_total, lr_dev, lr_mult_tail, decay, momentum, eps, grad_scale,
                           global_step_dev, rng_state_dev, rng_increment, p16);
    else
        hipLaunchKernelGGL(step_epilogue_kernel<false>, dim3(pw_blocks(n_total)), dim3(PW_THREADS), 0, air_stream(stream), p,
                           g, ms, mg, mom, n_model, n_total, lr_dev, lr_mult_tail, decay, momentum, eps, grad_scale,
                           global_step_dev, rng_state_dev, rng_increment, p16);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// out16[i] = bf16(x[i]) (round to nearest even): the bf16 mirror of a buffer no kernel of this library produces -- the observation
// batch at the start of a step, the parameters after a load
__global__ __launch_bounds__(PW_THREADS) void f32_to_bf16_kernel(const float *__restrict__ x, unsigned short *__restrict__ out, size_t n) {
    PW_LOOP(i, n) out[i] = __builtin_bit_cast(unsigned short, (__bf16)x[i]);
}
__global__ __launch_bounds__(PW_THREADS) void f32_to_bf16_vec_kernel(const float4 *__restrict__ x, uint2 *__restrict__ out, size_t nq) {
    PW_LOOP(q, nq) {
        const float4 v = x[q];
        const unsigned lo = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v.x) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v.y) << 16);
        const unsigned hi = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v.z) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v.w) << 16);
        out[q] = make_uint2(lo, hi);
    }
}
extern "C" int air_f32_to_bf16(const float *x, void *out_bf16, size_t n, void *stream) {
    AIR_REQUIRE(x && out_bf16, AIR_E_NULL);
    AIR_REQUIRE(n > 0, AIR_E_SHAPE);
    if (n % 4 == 0 && air_aligned16(x) && ((uintptr_t)out_bf16 % 8 == 0))
        hipLaunchKernelGGL(f32_to_bf16_vec_kernel, dim3(pw_blocks(n >> 2)), dim3(PW_THREADS), 0, air_stream(stream),
                           reinterpret_cast<const float4 *>(x), reinterpret_cast<uint2 *>(out_bf16), n >> 2);
    else
        hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(pw_blocks(n)), dim3(PW_THREADS), 0, air_stream(stream), x,
                           (unsigned short *)out_bf16, n);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// ---- L2 weight decay of the objective (model.py:346-353): g += l2 * w over the 2-D model weights ------------------------------
// The reference adds l2_weight * sum(w^2) / 2 over every model variable with a 2-D shape to the loss; its gradient is one axpy per
// such tensor.  One launch over up to AIR_L2_MAX_RANGES [lo, hi) slices of the flat buffers (biases and baseline variables lie
// between the slices and are left alone).
struct L2Ranges { size_t lo[AIR_L2_MAX_RANGES], hi[AIR_L2_MAX_RANGES]; int n; };
__global__ __launch_bounds__(PW_THREADS) void l2_grad_kernel(float *__restrict__ g, const float *__restrict__ p, L2Ranges r, float l2) {
    for (int k = 0; k < r.n; ++k) {
        const size_t lo = r.lo[k], hi = r.hi[k];
        for (size_t i = lo + (size_t)blockIdx.x * PW_THREADS + threadIdx.x; i < hi; i += (size_t)gridDim.x * PW_THREADS)
            g[i] = __builtin_fmaf(l2, p[i], g[i]);
    }
}
extern "C" int air_l2_grad_add(float *g, const float *p, const size_t *range_lo, const size_t *range_hi, int n_ranges, float l2_weight,
                               void *stream) {
    AIR_REQUIRE(g && p && range_lo && range_hi, AIR_E_NULL);
    AIR_REQUIRE(n_ranges > 0 && n_ranges <= AIR_L2_MAX_RANGES, AIR_E_SHAPE);
    L2Ranges r;
    size_t longest = 0;
    for (int k = 0; k < AIR_L2_MAX_RANGES; ++k) {
        r.lo[k] = k < n_ranges ? range_lo[k] : 0; r.hi[k] = k < n_ranges ? range_hi[k] : 0;
        AIR_REQUIRE(r.lo[k] <= r.hi[k], AIR_E_SHAPE);
        if (r.hi[k] - r.lo[k] > longest) longest = r.hi[k] - r.lo[k];
    }
    r.n = n_ranges;
    hipLaunchKernelGGL(l2_grad_kernel, dim3(pw_blocks(longest ? longest : 1)), dim3(PW_THREADS), 0, air_stream(stream), g, p, r, l2_weight);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// ---- utilities ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PW_THREADS) void fill_kernel(float *p, size_t n, float v) { PW_LOOP(i, n) p[i] = v; }
__global__ __launch_bounds__(PW_THREADS) void axpby_kernel(const float *__restrict__ a, float alpha,
                                                           const float *__restrict__ b, float beta,
                                                           float *__restrict__ out, size_t n) {
    PW_LOOP(i, n) out[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
}
__global__ __launch_bounds__(PW_THREADS) void tile_rows_kernel(const float *__restrict__ src, float *__restrict__ out,
                                                               int rows, int cols) {
    PW_LOOP(i, (size_t)rows * cols) out[i] = src[i % cols];
}
__global__ __launch_bounds__(PW_THREADS) void colsum_kernel(const float *__restrict__ x, int ld, float *__restrict__ out,
                                                            int M, int N) {
    // one thread per column, rows in order (deterministic); coalesced across columns
    PW_LOOP(n, (size_t)N) {
        float s = 0.f;
        for (int m = 0; m < M; ++m) s += x[(size_t)m * ld + n];
        out[n] = s;
    }
}
extern "C" int air_fill(float *p, size_t n, float v, void *stream) {
    AIR_REQUIRE(p, AIR_E_NULL);
    if (n == 0) return AIR_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(pw_blocks(n)), dim3(PW_THREADS), 0, air_stream(stream), p, n, v);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
extern "C" int air_axpby(const float *a, float alpha, const float *b, float beta, float *out, size_t n, void *stream) {
    AIR_REQUIRE(a && out, AIR_E_NULL);
    if (n == 0) return AIR_OK;
    hipLaunchKernelGGL(axpby_kernel, dim3(pw_blocks(n)), dim3(PW_THREADS), 0, air_stream(stream), a, alpha, b, beta, out, n);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
extern "C" int air_tile_rows(const float *src, float *out, int rows, int cols, void *stream) {
    AIR_REQUIRE(src && out, AIR_E_NULL);
    AIR_REQUIRE(rows > 0 && cols > 0, AIR_E_SHAPE);
    hipLaunchKernelGGL(tile_rows_kernel, dim3(pw_blocks((size_t)rows * cols)), dim3(PW_THREADS), 0, air_stream(stream),
                       src, out, rows, cols);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
extern "C" int air_colsum(const float *x, int ld, float *out, int M, int N, void *stream) {
    AIR_REQUIRE(x && out, AIR_E_NULL);
    AIR_REQUIRE(M > 0 && N > 0 && ld >= N, AIR_E_SHAPE);
    hipLaunchKernelGGL(colsum_kernel, dim3(pw_blocks(N)), dim3(PW_THREADS), 0, air_stream(stream), x, ld, out, M, N);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

__global__ __launch_bounds__(PW_THREADS) void sum_leading_kernel(const float *__restrict__ x, float *__restrict__ out,
                                                                 int T, size_t n) {
    PW_LOOP(i, n) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += x[(size_t)t * n + i];
        out[i] = s;
    }
}
extern "C" int air_sum_leading(const float *x, float *out, int T, size_t n, void *stream) {
    AIR_REQUIRE(x && out, AIR_E_NULL);
    AIR_REQUIRE(T > 0 && n > 0, AIR_E_SHAPE);
    hipLaunchKernelGGL(sum_leading_kernel, dim3(pw_blocks(n)), dim3(PW_THREADS), 0, air_stream(stream), x, out, T, n);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
