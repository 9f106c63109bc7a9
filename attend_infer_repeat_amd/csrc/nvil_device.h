// NVIL objective as a device body, so that it can run as its own launch (air_nvil) or as one extra workgroup of the
// canvas backward launch (air_canvas_unroll_bwd_nvil): the two are independent and the step is launch bound.
#pragma once
#include "air_common.h"

struct NvilArgs {
    const float *imp, *base, *logp;
    float *out, *dlogp, *dbase;
    int B;
    // imp may arrive as `imp_parts` shares per sample (imp[p*B + i], the row bands of air_canvas_unroll_fwd_banded): they are
    // added in share order in fp32 -- exactly what a single-band launch would have stored -- and the sum is written to
    // imp_sum (the complete rec_loss_per_sample) when that pointer is given
    int imp_parts;
    float *imp_sum;
    // optional EMA normalisation of the importance weight (decay_rate of model.py:232-239, ops.py:46-64), a DEVICE block of four
    // floats {moving_mean, moving_var, decay_rate, update}: the [B,B] weight is shifted by the moving mean and divided by
    // max(sqrt(moving_var), 1) -- the values the variables hold BEFORE this step's update, as the reference's graph reads them --
    // and, when update != 0 (train steps; evaluation passes only read), the two averages then move towards this batch's mean /
    // variance of the [B,B] weight.  NULL = the script's decay_rate=None.
    float *ema;
};
__device__ __forceinline__ float nvil_imp(const NvilArgs &g, int i) {
    float r = g.imp[i];
    for (int p = 1; p < g.imp_parts; ++p) r += g.imp[(size_t)p * g.B + i];
    return r;
}

// importance_weight[i,j] = imp[j] - baseline[i]  ([B]-[B,1] broadcast, model.py:218-259 / SURVEY Appendix B-1), so
//   reinforce_loss = mean_j (imp_j - mean_i b_i) * logp_j ;  baseline_loss = 0.5 * mean_ij (imp_j - b_i)^2.
// One workgroup of any size up to 1024 threads.  The six fp64 sums are formed by wave 0 alone (lane-strided over the batch,
// then ONE multi-value wave reduction), so there is a single barrier -- the broadcast of the two means -- and no serial
// cross-wave stage; fixed order => bitwise reproducible.
__device__ __forceinline__ void nvil_body(const NvilArgs &g) {
    __shared__ double tot[4];
    const int nt = blockDim.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (wid == 0) {
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // sum imp, imp^2, b, b^2, imp*logp, logp, (unused x2)
        for (int i = lane; i < g.B; i += 64) {
            const float xf = nvil_imp(g, i);
            if (g.imp_sum) g.imp_sum[i] = xf;
            const double x = xf, b = g.base[i], l = g.logp[i];
            a[0] += x; a[1] += x * x; a[2] += b; a[3] += b * b; a[4] += x * l; a[5] += l;
        }
        const double r = wave_reduce8(a);            // lane 8*o holds total o
        double t[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) t[k] = __shfl(r, 8 * k, 64);
        if (lane == 0) {
            const double n = (double)g.B;
            const double mi = t[0] / n, mb = t[2] / n;
            const double vi = t[1] / n - mi * mi, vb = t[3] / n - mb * mb;
            double shift = 0.0, inv_f = 1.0;
            if (g.ema) {
                const float mm = g.ema[0], mv = g.ema[1], d = g.ema[2];
                shift = (double)mm;
                const float f = sqrtf(mv);
                inv_f = 1.0 / (double)(f > 1.f ? f : 1.f);
                if (g.ema[3] != 0.f) {                                                 // assign_moving_average, zero_debias=False
                    g.ema[0] = d * mm + (1.f - d) * (float)(mi - mb);
                    g.ema[1] = d * mv + (1.f - d) * (float)(vi + vb);
                }
            }
            g.out[0] = (float)(((t[4] - mb * t[5]) - shift * t[5]) * inv_f / n);      // reinforce_loss
            g.out[1] = (float)(0.5 * (vi + vb + (mi - mb) * (mi - mb)));              // baseline_loss (not normalised, model.py:253-256)
            g.out[2] = (float)((mi - mb - shift) * inv_f);                             // imp_weight_mean over [B,B]
            g.out[3] = (float)((vi + vb) * inv_f * inv_f);                             // imp_weight_var  over [B,B]
            tot[0] = mi; tot[1] = mb; tot[2] = shift; tot[3] = inv_f;
        }
    }
    __syncthreads();
    const double mi = tot[0], mb = tot[1], shift = tot[2], inv_f = tot[3];
    for (int i = threadIdx.x; i < g.B; i += nt) {
        if (g.dlogp) g.dlogp[i] = (float)((((double)nvil_imp(g, i) - mb) - shift) * inv_f / (double)g.B);
        if (g.dbase) g.dbase[i] = (float)(-(mi - (double)g.base[i]) / (double)g.B);
    }
}
