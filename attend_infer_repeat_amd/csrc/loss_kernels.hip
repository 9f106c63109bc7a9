// Objective-side kernels of AIR for gfx950: number-of-steps posterior / KL (float64 like the reference's prior.py),
// the NVIL / REINFORCE scalars with the reference's [B]-[B,1] broadcast, plus hipGraph / event plumbing.
#include <math.h>
#include "air_common.h"

#define NS_MAXT 32

// ---- q(n), KL(q || geometric prior), step weights, log q(n_sampled)   (prior.py:62-151, model.py:139-163) --------
struct NumSteps {
    double p[NS_MAXT], u[NS_MAXT + 1], q[NS_MAXT + 1], S;
    float q32[NS_MAXT + 1];
};
__device__ __forceinline__ void numsteps_posterior(const float *__restrict__ prob, int T, int B, int b, NumSteps &s) {
    for (int t = 0; t < T; ++t) s.p[t] = (double)prob[(size_t)t * B + b];
    double cum = 1.0;
    s.u[0] = 1.0 - s.p[0];
    for (int n = 1; n < T; ++n) { cum *= s.p[n - 1]; s.u[n] = (1.0 - s.p[n]) * cum; }
    cum *= s.p[T - 1];
    s.u[T] = cum;
    s.S = 0.0;
    for (int n = 0; n <= T; ++n) s.S += s.u[n];
    for (int n = 0; n <= T; ++n) { s.q[n] = s.u[n] / s.S; s.q32[n] = (float)s.q[n]; }
}
__device__ __forceinline__ int sampled_steps(const float *__restrict__ presence, int T, int B, int b) {
    float n = 0.f;
    for (int t = 0; t < T; ++t) n += presence[(size_t)t * B + b];
    int k = (int)n;
    return k < 0 ? 0 : (k > T ? T : k);
}

__global__ __launch_bounds__(256) void numsteps_fwd_kernel(const float *__restrict__ prob,
                                                           const float *__restrict__ presence,
                                                           const double *__restrict__ prior, float *__restrict__ q,
                                                           float *__restrict__ kl_ps, float *__restrict__ logp,
                                                           float *__restrict__ step_w, int T, int B) {
    for (int b = blockIdx.x * 256 + threadIdx.x; b < B; b += gridDim.x * 256) {
        NumSteps s;
        numsteps_posterior(prob, T, B, b, s);
        float kl = 0.f;
        for (int n = 0; n <= T; ++n) {
            if (q) q[(size_t)b * (T + 1) + n] = s.q32[n];
            const double pn = (double)s.q32[n];                       // tabular_kl re-casts the f32 posterior to f64
            kl += (pn > 0.0) ? (float)(pn * log(pn / prior[n])) : 0.f;
        }
        if (kl_ps) kl_ps[b] = kl;
        if (step_w) {
            float w = 0.f;
            for (int t = T - 1; t >= 0; --t) { w += s.q32[t + 1]; step_w[(size_t)t * B + b] = w; }
        }
        if (logp) {
            const float pr = s.q32[sampled_steps(presence, T, B, b)];
            logp[b] = logf(fmaxf(pr, 1e-32f));
        }
    }
}

__global__ __launch_bounds__(256) void numsteps_bwd_kernel(const float *__restrict__ prob,
                                                           const float *__restrict__ presence,
                                                           const double *__restrict__ prior, float kl_scale,
                                                           const float *__restrict__ dstep_w,
                                                           const float *__restrict__ dlogp,
                                                           float *__restrict__ dprob, int T, int B) {
    for (int b = blockIdx.x * 256 + threadIdx.x; b < B; b += gridDim.x * 256) {
        NumSteps s;
        numsteps_posterior(prob, T, B, b, s);
        double gq[NS_MAXT + 1];
        double wsum = 0.0;
        const int nstar = (dlogp && presence) ? sampled_steps(presence, T, B, b) : -1;
        for (int n = 0; n <= T; ++n) {
            const double pn = (double)s.q32[n];
            double g = (pn > 0.0) ? (double)kl_scale * (log(pn / prior[n]) + 1.0) : 0.0;
            if (n >= 1 && dstep_w) wsum += (double)dstep_w[(size_t)(n - 1) * B + b];   // w_t includes q(n) for all n > t
            g += wsum;
            if (n == nstar) g += (double)dlogp[b] / (double)fmaxf(s.q32[n], 1e-32f);
            gq[n] = g;
        }
        // q = u / S
        double dot = 0.0;
        for (int n = 0; n <= T; ++n) dot += gq[n] * s.q[n];
        double gu[NS_MAXT + 1];
        for (int n = 0; n <= T; ++n) gu[n] = (gq[n] - dot) / s.S;
        // u -> p, products taken without division (safe at p = 0, like the reference's scan-based cumprod)
        for (int k = 0; k < T; ++k) {
            double g = 0.0;
            for (int n = 0; n <= T; ++n) {
                double d;
                if (n < T) {
                    if (k > n) continue;
                    if (k == n) {                                       // d/dp_n of (1-p_n) * prod_{j<n} p_j
                        d = -1.0;
                        for (int j = 0; j < n; ++j) d *= s.p[j];
                    } else {                                            // k < n
                        d = 1.0 - s.p[n];
                        for (int j = 0; j < n; ++j) if (j != k) d *= s.p[j];
                    }
                } else {                                                // u_T = prod_j p_j
                    d = 1.0;
                    for (int j = 0; j < T; ++j) if (j != k) d *= s.p[j];
                }
                g += gu[n] * d;
            }
            dprob[(size_t)k * B + b] = (float)g;
        }
    }
}

extern "C" int air_numsteps_fwd(const float *presence_prob, const float *presence, const double *prior_f64, float *q,
                                float *kl_per_sample, float *logp, float *step_weight, int T, int B, void *stream) {
    AIR_REQUIRE(presence_prob && prior_f64, AIR_E_NULL);
    AIR_REQUIRE(!logp || presence, AIR_E_NULL);
    AIR_REQUIRE(T > 0 && T <= NS_MAXT && B > 0, AIR_E_SHAPE);
    hipLaunchKernelGGL(numsteps_fwd_kernel, dim3(air_cdiv(B, 256)), dim3(256), 0, air_stream(stream), presence_prob,
                       presence, prior_f64, q, kl_per_sample, logp, step_weight, T, B);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
extern "C" int air_numsteps_bwd(const float *presence_prob, const float *presence, const double *prior_f64,
                                float kl_scale, const float *dstep_weight, const float *dlogp, float *dpresence_prob,
                                int T, int B, void *stream) {
    AIR_REQUIRE(presence_prob && prior_f64 && dpresence_prob, AIR_E_NULL);
    AIR_REQUIRE(!dlogp || presence, AIR_E_NULL);
    AIR_REQUIRE(T > 0 && T <= NS_MAXT && B > 0, AIR_E_SHAPE);
    hipLaunchKernelGGL(numsteps_bwd_kernel, dim3(air_cdiv(B, 256)), dim3(256), 0, air_stream(stream), presence_prob,
                       presence, prior_f64, kl_scale, dstep_weight, dlogp, dpresence_prob, T, B);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// ---- fused forms used by the engine (one thread per image; the step is launch bound) -----------------------------
// Templated on a compile-time bound MT >= T so that every per-image array lives in registers (the generic kernels
// above index float64 arrays dynamically and run out of scratch memory: ~10 us for 64 images).
#include "engine_device.h"
#include "nvil_device.h"
template <int MT>
__global__ __launch_bounds__(64) void presence_numsteps_fwd_kernel(
    const float *__restrict__ logit, const float *__restrict__ u, float step_bias, float eps,
    const double *__restrict__ prior, float *__restrict__ prob, float *__restrict__ pres, float *__restrict__ q,
    float *__restrict__ kl_ps, float *__restrict__ logp, float *__restrict__ step_w, int T, int B) {
    presence_numsteps_fwd_body<MT>(blockIdx.x, gridDim.x, logit, u, step_bias, eps, prior, prob, pres, q, kl_ps, logp,
                                   step_w, T, B);
}
extern "C" int air_presence_numsteps_fwd(const float *logit, const float *u, float step_bias, float explore_eps,
                                         const double *prior_f64, float *presence_prob, float *presence, float *q,
                                         float *kl_per_sample, float *logp, float *step_weight, int T, int B,
                                         void *stream) {
    AIR_REQUIRE(logit && prior_f64 && presence_prob && presence && q && kl_per_sample && logp && step_weight,
                AIR_E_NULL);                                   // u == NULL: continuous steps (presence = presence_prob, cell.py:150-151)
    AIR_REQUIRE(T > 0 && T <= NS_MAXT && B > 0, AIR_E_SHAPE);
    if (T <= 8)
        hipLaunchKernelGGL(presence_numsteps_fwd_kernel<8>, dim3(air_cdiv(B, 64)), dim3(64), 0, air_stream(stream), logit,
                           u, step_bias, explore_eps, prior_f64, presence_prob, presence, q, kl_per_sample, logp,
                           step_weight, T, B);
    else
        hipLaunchKernelGGL(presence_numsteps_fwd_kernel<NS_MAXT>, dim3(air_cdiv(B, 64)), dim3(64), 0, air_stream(stream),
                           logit, u, step_bias, explore_eps, prior_f64, presence_prob, presence, q, kl_per_sample, logp,
                           step_weight, T, B);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

template <int MT>
__global__ __launch_bounds__(64) void numsteps_presence_bwd_kernel(
    const float *__restrict__ prob, const float *__restrict__ presence, const double *__restrict__ prior,
    float kl_scale, const float *__restrict__ kl_a, const float *__restrict__ kl_b, float w_scale,
    const float *__restrict__ dlogp, const float *__restrict__ dpres, const float *__restrict__ logit, float step_bias, float eps,
    float *__restrict__ dlogit, int T, int B) {
    numsteps_presence_bwd_body<MT>(blockIdx.x, gridDim.x, prob, presence, prior, kl_scale, kl_a, kl_b, w_scale, dlogp,
                                   logit, step_bias, eps, dlogit, T, B, dpres);
}
extern "C" int air_numsteps_presence_bwd(const float *presence_prob, const float *presence, const double *prior_f64,
                                         float kl_scale, const float *kl_row_a, const float *kl_row_b, float w_scale,
                                         const float *dlogp, const float *dpresence, const float *logit, float step_bias,
                                         float explore_eps, float *dlogit, int T, int B, void *stream) {
    AIR_REQUIRE(presence_prob && prior_f64 && logit && dlogit, AIR_E_NULL);
    AIR_REQUIRE(!dlogp || presence, AIR_E_NULL);
    AIR_REQUIRE(T > 0 && T <= NS_MAXT && B > 0, AIR_E_SHAPE);
    if (T <= 8)
        hipLaunchKernelGGL(numsteps_presence_bwd_kernel<8>, dim3(air_cdiv(B, 64)), dim3(64), 0, air_stream(stream),
                           presence_prob, presence, prior_f64, kl_scale, kl_row_a, kl_row_b, w_scale, dlogp, dpresence, logit,
                           step_bias, explore_eps, dlogit, T, B);
    else
        hipLaunchKernelGGL(numsteps_presence_bwd_kernel<NS_MAXT>, dim3(air_cdiv(B, 64)), dim3(64), 0, air_stream(stream),
                           presence_prob, presence, prior_f64, kl_scale, kl_row_a, kl_row_b, w_scale, dlogp, dpresence, logit,
                           step_bias, explore_eps, dlogit, T, B);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// ---- annealed geometric prior on device (model.py:106-124, prior.py:26-32) -------------------------------------------
__global__ void steps_prior_kernel(const int64_t *__restrict__ gstep, int anneal_type, double init, double fin,
                                   double anneal_steps, double hold_for, double steps_div,
                                   double *__restrict__ prior, int T) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double s = init;
    if (anneal_type != 0) {
        double step = (double)gstep[0] - hold_for;
        if (step < 0.0) step = 0.0;
        double val = init;
        if (anneal_type == 1) {
            const double decay_rate = pow(fin / init, steps_div / anneal_steps);
            val = init * pow(decay_rate, step / steps_div);
        } else {
            val = fin + (init - fin) * (1.0 - step / anneal_steps);
        }
        s = val > fin ? val : fin;
    }
    s = s < 1e-7 ? 1e-7 : (s > 1.0 - 1e-15 ? 1.0 - 1e-15 : s);
    const double probs = 1.0 - s;
    for (int n = 0; n <= T; ++n) prior[n] = exp((double)n * log1p(-probs) + log(probs));
}
__global__ void counter_add_kernel(int64_t *c, int64_t inc) {
    if (blockIdx.x == 0 && threadIdx.x == 0) c[0] += inc;
}
extern "C" int air_steps_prior(const int64_t *global_step_dev, int anneal_type, double init, double final_value,
                               double anneal_steps, double hold_for, double steps_div, double *prior_out_f64, int T,
                               void *stream) {
    AIR_REQUIRE(global_step_dev && prior_out_f64, AIR_E_NULL);
    AIR_REQUIRE(T > 0 && anneal_type >= 0 && anneal_type <= 2, AIR_E_SHAPE);
    hipLaunchKernelGGL(steps_prior_kernel, dim3(1), dim3(64), 0, air_stream(stream), global_step_dev, anneal_type,
                       init, final_value, anneal_steps, hold_for, steps_div, prior_out_f64, T);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
extern "C" int air_counter_add(int64_t *counter_dev, int64_t increment, void *stream) {
    AIR_REQUIRE(counter_dev, AIR_E_NULL);
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, air_stream(stream), counter_dev, increment);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// ---- NVIL / REINFORCE (model.py:218-259) --------------------------------------------------------------------------
// importance_weight[i,j] = imp[j] - baseline[i]  ([B]-[B,1] broadcast), so
//   reinforce_loss = mean_j (imp_j - mean_i b_i) * logp_j ;  baseline_loss = 0.5 * mean_ij (imp_j - b_i)^2.
__global__ __launch_bounds__(256) void nvil_kernel(NvilArgs a) { nvil_body(a); }
// REINFORCE importance weight of a NON-analytic num-steps prior (model.py:339-340: reinforce_imp_weight += prior_loss.per_sample):
//   rec[b] = sum of the reconstruction shares (in share order); imp[b] = rec[b] + nsp_weight * kl_n[b] + sum_t w[t,b] (kl_a[t,b] + kl_b[t,b])
__global__ __launch_bounds__(256) void imp_weight_kernel(const float *__restrict__ parts, int n_parts, float *__restrict__ rec,
                                                         const float *__restrict__ kl_n, float nsp_w, const float *__restrict__ kl_a,
                                                         const float *__restrict__ kl_b, const float *__restrict__ w, int T, int B,
                                                         float *__restrict__ imp, float *__restrict__ dpres, float dkl_scale) {
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
        float r = parts[b];
        for (int p = 1; p < n_parts; ++p) r += parts[(size_t)p * B + b];
        if (rec) rec[b] = r;
        float pr = kl_n ? nsp_w * kl_n[b] : 0.f;
        for (int t = 0; t < T; ++t) {
            const size_t k = (size_t)t * B + b;
            const float kl = (kl_a ? kl_a[k] : 0.f) + (kl_b ? kl_b[k] : 0.f);
            pr += w[k] * kl;
            // continuous steps (cell.py:150-151): the step weight IS the presence probability, so the weighted KL rows reach it directly
            if (dpres) dpres[k] += dkl_scale * kl;
        }
        if (imp) imp[b] = r + pr;
    }
}
extern "C" int air_imp_weight(const float *rec_parts, int n_parts, float *rec_out, const float *kl_n, float nsp_weight,
                              const float *kl_row_a, const float *kl_row_b, const float *step_weight, int T, int B, float *imp_out,
                              float *dpresence_inout, float dkl_scale, void *stream) {
    AIR_REQUIRE(rec_parts && step_weight && (imp_out || dpresence_inout), AIR_E_NULL);
    AIR_REQUIRE(n_parts > 0 && T > 0 && B > 0, AIR_E_SHAPE);
    hipLaunchKernelGGL(imp_weight_kernel, dim3(air_cdiv(B, 256)), dim3(256), 0, air_stream(stream), rec_parts, n_parts, rec_out, kl_n,
                       nsp_weight, kl_row_a, kl_row_b, step_weight, T, B, imp_out, dpresence_inout, dkl_scale);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
extern "C" int air_nvil(const float *imp, const float *baseline, const float *logp, float *out, float *dlogp,
                        float *dbaseline, int B, float *ema_dev, void *stream) {
    AIR_REQUIRE(imp && baseline && logp && out, AIR_E_NULL);
    AIR_REQUIRE(B > 0, AIR_E_SHAPE);
    NvilArgs a = {imp, baseline, logp, out, dlogp, dbaseline, B, 1, nullptr, ema_dev};
    hipLaunchKernelGGL(nvil_kernel, dim3(1), dim3(256), 0, air_stream(stream), a);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
extern "C" int air_nvil_parts(const float *imp_parts, int n_parts, float *imp_sum, const float *baseline,
                              const float *logp, float *out, float *dlogp, float *dbaseline, int B, float *ema_dev,
                              void *stream) {
    AIR_REQUIRE(imp_parts && baseline && logp && out, AIR_E_NULL);
    AIR_REQUIRE(B > 0 && n_parts > 0, AIR_E_SHAPE);
    NvilArgs a = {imp_parts, baseline, logp, out, dlogp, dbaseline, B, n_parts, imp_sum, ema_dev};
    hipLaunchKernelGGL(nvil_kernel, dim3(1), dim3(256), 0, air_stream(stream), a);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// ---- hipGraph + event plumbing -----------------------------------------------------------------------------------
extern "C" int air_graph_begin_capture(void *stream) {
    hipError_t e = hipStreamBeginCapture(air_stream(stream), hipStreamCaptureModeRelaxed);
    return (int)e;
}
extern "C" int air_graph_end_capture(void *stream, void **graph_exec_out) {
    AIR_REQUIRE(graph_exec_out, AIR_E_NULL);
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(air_stream(stream), &graph);
    if (e != hipSuccess) return (int)e;
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return (int)e;
    *graph_exec_out = (void *)exec;
    return AIR_OK;
}
extern "C" int air_graph_launch(void *graph_exec, void *stream) {
    AIR_REQUIRE(graph_exec, AIR_E_NULL);
    return (int)hipGraphLaunch((hipGraphExec_t)graph_exec, air_stream(stream));
}
extern "C" int air_graph_destroy(void *graph_exec) {
    if (!graph_exec) return AIR_OK;
    return (int)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
}
extern "C" int air_event_create(void **event_out) {
    AIR_REQUIRE(event_out, AIR_E_NULL);
    hipEvent_t ev;
    hipError_t e = hipEventCreate(&ev);
    if (e != hipSuccess) return (int)e;
    *event_out = (void *)ev;
    return AIR_OK;
}
extern "C" int air_event_record(void *event, void *stream) {
    AIR_REQUIRE(event, AIR_E_NULL);
    return (int)hipEventRecord((hipEvent_t)event, air_stream(stream));
}
extern "C" int air_event_elapsed_ms(void *start, void *stop, float *ms_host_out) {
    AIR_REQUIRE(start && stop && ms_host_out, AIR_E_NULL);
    hipError_t e = hipEventSynchronize((hipEvent_t)stop);
    if (e != hipSuccess) return (int)e;
    return (int)hipEventElapsedTime(ms_host_out, (hipEvent_t)start, (hipEvent_t)stop);
}
extern "C" int air_event_destroy(void *event) {
    if (!event) return AIR_OK;
    return (int)hipEventDestroy((hipEvent_t)event);
}

extern "C" int air_abi_version(void) { return AIR_ABI_VERSION; }
extern "C" int air_engine_abi_version(void) { return AIR_ENGINE_ABI_VERSION; }
#ifndef AIR_BUILD_DIGEST
#define AIR_BUILD_DIGEST "unstamped"
#endif
extern "C" const char *air_build_digest(void) { return AIR_BUILD_DIGEST; }
extern "C" const char *air_status_string(int status) {
    switch (status) {
        case AIR_OK: return "ok";
        case AIR_E_NULL: return "AIR_E_NULL: required pointer is NULL";
        case AIR_E_SHAPE: return "AIR_E_SHAPE: bad dimension";
        case AIR_E_ALIGN: return "AIR_E_ALIGN: alignment requirement violated";
        case AIR_E_WORKSPACE: return "AIR_E_WORKSPACE: workspace too small";
        case AIR_E_UNSUPPORTED: return "AIR_E_UNSUPPORTED: unsupported argument combination";
        default: return status > 0 ? hipGetErrorString((hipError_t)status) : "unknown AIR status";
    }
}
