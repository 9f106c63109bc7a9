// NVIL objective as a device body, so that it can run as its own launch (air_nvil) or as one extra workgroup of the
// canvas backward launch (air_canvas_unroll_bwd_nvil): the two are independent and the step is launch bound.
#pragma once
#include "air_common.h"

struct NvilArgs {
    const float *imp, *base, *logp;
    float *out, *dlogp, *dbase;
    int B;
};

// importance_weight[i,j] = imp[j] - baseline[i]  ([B]-[B,1] broadcast, model.py:218-259 / SURVEY Appendix B-1), so
//   reinforce_loss = mean_j (imp_j - mean_i b_i) * logp_j ;  baseline_loss = 0.5 * mean_ij (imp_j - b_i)^2.
// One workgroup of any size up to 1024 threads; fp64 accumulation in a fixed order.
__device__ __forceinline__ void nvil_body(const NvilArgs &g) {
    __shared__ double red[16][6];
    __shared__ double tot[2];
    const int nt = blockDim.x, nw = (nt + 63) >> 6;
    double a[6] = {0, 0, 0, 0, 0, 0};               // sum imp, imp^2, b, b^2, imp*logp, logp
    for (int i = threadIdx.x; i < g.B; i += nt) {
        const double x = g.imp[i], b = g.base[i], l = g.logp[i];
        a[0] += x; a[1] += x * x; a[2] += b; a[3] += b * b; a[4] += x * l; a[5] += l;
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) a[k] = wave_sum(a[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) red[wid][k] = a[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[6];
        for (int k = 0; k < 6; ++k) {
            double v = 0.0;
            for (int q = 0; q < nw; ++q) v += red[q][k];
            t[k] = v;
        }
        const double n = (double)g.B;
        const double mi = t[0] / n, mb = t[2] / n;
        const double vi = t[1] / n - mi * mi, vb = t[3] / n - mb * mb;
        g.out[0] = (float)((t[4] - mb * t[5]) / n);                                // reinforce_loss
        g.out[1] = (float)(0.5 * (vi + vb + (mi - mb) * (mi - mb)));              // baseline_loss
        g.out[2] = (float)(mi - mb);                                               // imp_weight_mean over [B,B]
        g.out[3] = (float)(vi + vb);                                               // imp_weight_var  over [B,B]
        tot[0] = mi; tot[1] = mb;
    }
    __syncthreads();
    const double mi = tot[0], mb = tot[1];
    for (int i = threadIdx.x; i < g.B; i += nt) {
        if (g.dlogp) g.dlogp[i] = (float)(((double)g.imp[i] - mb) / (double)g.B);
        if (g.dbase) g.dbase[i] = (float)(-(mi - (double)g.base[i]) / (double)g.B);
    }
}
