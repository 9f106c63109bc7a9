// Fused "heads" launches of the engine: two INDEPENDENT small kernels share one dispatch (blocks are split by role),
// because at batch 64 the train step is bound by the number of dependent launches, not by work.
//   forward : where ~ N(loc, softplus(raw))  (cell.py:129-133)   ||  presence + q(n) / KL / step weights (cell.py:137-151,
//             prior.py:62-151)                 -- both only read the transform / steps MLP outputs
//   backward: d where-parameters (needs dwhere from both ST kernels)  ||  d steps-logit (num-steps KL, step weights, REINFORCE)
#include "engine_device.h"
#include "prologue_device.h"

template <int MT>
__global__ __launch_bounds__(PW_THREADS) void heads_fwd_kernel(
    int gauss_blocks,
    const float *__restrict__ pre, int ld_pre, const float *__restrict__ eps, RawOffset raw_offset, int loc_mode, float pl0,
    float ps0, float pl1, float ps1, float *__restrict__ loc, float *__restrict__ scale, float *__restrict__ sample,
    float *__restrict__ kl_row, int M, int D,
    const float *__restrict__ logit, const float *__restrict__ u, float step_bias, float explore_eps,
    const double *__restrict__ prior, float *__restrict__ prob, float *__restrict__ pres, float *__restrict__ q,
    float *__restrict__ kl_ps, float *__restrict__ logp, float *__restrict__ step_w, int T, int B) {
    if ((int)blockIdx.x < gauss_blocks)
        gauss_fwd_body(blockIdx.x, gauss_blocks, pre, ld_pre, eps, raw_offset, loc_mode, pl0, ps0, pl1, ps1, loc, scale,
                       sample, kl_row, M, D);
    else
        presence_numsteps_fwd_body<MT>(blockIdx.x - gauss_blocks, gridDim.x - gauss_blocks, logit, u, step_bias,
                                       explore_eps, prior, prob, pres, q, kl_ps, logp, step_w, T, B);
}

template <int MT>
__global__ __launch_bounds__(PW_THREADS) void heads_bwd_kernel(
    int gauss_blocks,
    const float *__restrict__ pre, int ld_pre, const float *__restrict__ eps, RawOffset raw_offset, int loc_mode, float pl0,
    float ps0, float pl1, float ps1, const float *__restrict__ loc, const float *__restrict__ scale,
    const float *__restrict__ dsample, const float *__restrict__ dsample2, const float *__restrict__ dkl_row,
    float dkl_scale, float *__restrict__ dpre, int ld_dpre, int M, int D,
    const float *__restrict__ prob, const float *__restrict__ presence, const double *__restrict__ prior, float kl_scale,
    const float *__restrict__ kl_a, const float *__restrict__ kl_b, float w_scale, const float *__restrict__ dlogp,
    const float *__restrict__ dpres, const float *__restrict__ logit, float step_bias, float explore_eps, float *__restrict__ dlogit,
    int T, int B) {
    if ((int)blockIdx.x < gauss_blocks)
        gauss_bwd_body(blockIdx.x, gauss_blocks, pre, ld_pre, eps, raw_offset, loc_mode, pl0, ps0, pl1, ps1, loc, scale,
                       dsample, dsample2, dkl_row, dkl_scale, dpre, ld_dpre, M, D);
    else
        numsteps_presence_bwd_body<MT>(blockIdx.x - gauss_blocks, gridDim.x - gauss_blocks, prob, presence, prior,
                                       kl_scale, kl_a, kl_b, w_scale, dlogp, logit, step_bias, explore_eps, dlogit, T, B, dpres);
}

static inline int blocks_for(size_t n) {
    size_t b = (n + PW_THREADS - 1) / PW_THREADS;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

extern "C" int air_heads_fwd(const float *pre, int ld_pre, const float *eps, float raw_offset, int loc_mode,
                             float p_loc_even, float p_scale_even, float p_loc_odd, float p_scale_odd, float *loc,
                             float *scale, float *sample, float *kl_row, int M, int D, const float *logit,
                             const float *u, float step_bias, float explore_eps, const double *prior_f64,
                             float *presence_prob, float *presence, float *q, float *kl_per_sample, float *logp,
                             float *step_weight, int T, int B, float guard_eps, void *stream) {
    AIR_REQUIRE(pre && eps && loc && scale && sample && kl_row && logit && prior_f64 && presence_prob && presence &&   /* u NULL: continuous steps */
                    q && kl_per_sample && logp && step_weight, AIR_E_NULL);
    AIR_REQUIRE(M > 0 && D > 0 && ld_pre >= 2 * D && T > 0 && T <= 32 && B > 0, AIR_E_SHAPE);
    const int gb = blocks_for((size_t)M * 64), nb = air_cdiv(B, 64);
    if (T <= 8)
        hipLaunchKernelGGL(heads_fwd_kernel<8>, dim3(gb + nb), dim3(PW_THREADS), 0, air_stream(stream), gb, pre, ld_pre,
                           eps, RawOffset(raw_offset, guard_eps), loc_mode, p_loc_even, p_scale_even, p_loc_odd, p_scale_odd, loc, scale, sample,
                           kl_row, M, D, logit, u, step_bias, explore_eps, prior_f64, presence_prob, presence, q,
                           kl_per_sample, logp, step_weight, T, B);
    else
        hipLaunchKernelGGL(heads_fwd_kernel<32>, dim3(gb + nb), dim3(PW_THREADS), 0, air_stream(stream), gb, pre, ld_pre,
                           eps, RawOffset(raw_offset, guard_eps), loc_mode, p_loc_even, p_scale_even, p_loc_odd, p_scale_odd, loc, scale, sample,
                           kl_row, M, D, logit, u, step_bias, explore_eps, prior_f64, presence_prob, presence, q,
                           kl_per_sample, logp, step_weight, T, B);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

extern "C" int air_heads_bwd(const float *pre, int ld_pre, const float *eps, float raw_offset, int loc_mode,
                             float p_loc_even, float p_scale_even, float p_loc_odd, float p_scale_odd, const float *loc,
                             const float *scale, const float *dsample, const float *dsample2, const float *dkl_row,
                             float dkl_scale, float *dpre, int ld_dpre, int M, int D, const float *presence_prob,
                             const float *presence, const double *prior_f64, float kl_scale, const float *kl_row_a,
                             const float *kl_row_b, float w_scale, const float *dlogp, const float *dpresence, const float *logit,
                             float step_bias, float explore_eps, float *dlogit, int T, int B, float guard_eps, void *stream) {
    AIR_REQUIRE(pre && eps && loc && scale && dpre && presence_prob && prior_f64 && logit && dlogit, AIR_E_NULL);
    AIR_REQUIRE(!dlogp || presence, AIR_E_NULL);
    AIR_REQUIRE(M > 0 && D > 0 && ld_pre >= 2 * D && ld_dpre >= 2 * D && T > 0 && T <= 32 && B > 0, AIR_E_SHAPE);
    const int gb = blocks_for((size_t)M * D), nb = air_cdiv(B, 64);
    if (T <= 8)
        hipLaunchKernelGGL(heads_bwd_kernel<8>, dim3(gb + nb), dim3(PW_THREADS), 0, air_stream(stream), gb, pre, ld_pre,
                           eps, RawOffset(raw_offset, guard_eps), loc_mode, p_loc_even, p_scale_even, p_loc_odd, p_scale_odd, loc, scale, dsample,
                           dsample2, dkl_row, dkl_scale, dpre, ld_dpre, M, D, presence_prob, presence, prior_f64, kl_scale,
                           kl_row_a, kl_row_b, w_scale, dlogp, dpresence, logit, step_bias, explore_eps, dlogit, T, B);
    else
        hipLaunchKernelGGL(heads_bwd_kernel<32>, dim3(gb + nb), dim3(PW_THREADS), 0, air_stream(stream), gb, pre, ld_pre,
                           eps, RawOffset(raw_offset, guard_eps), loc_mode, p_loc_even, p_scale_even, p_loc_odd, p_scale_odd, loc, scale, dsample,
                           dsample2, dkl_row, dkl_scale, dpre, ld_dpre, M, D, presence_prob, presence, prior_f64, kl_scale,
                           kl_row_a, kl_row_b, w_scale, dlogp, dpresence, logit, step_bias, explore_eps, dlogit, T, B);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// ---- what ~ N(loc, softplus(raw + offset)) (cell.py:154-156) + KL rows, writing the sample twice: time-major for the
// decoder and batch-major straight into the baseline input, whose remaining latent columns (where, presence, h, c -- all
// final by now) are copied by a second group of workgroups.  Replaces air_gauss_sample_fwd + air_baseline_pack.
__global__ __launch_bounds__(PW_THREADS) void what_sample_pack_kernel(
    int gauss_blocks, const float *__restrict__ pre, int ld_pre, const float *__restrict__ eps, RawOffset raw_offset,
    float pl, float ps, float *__restrict__ loc, float *__restrict__ scale, float *__restrict__ sample,
    float *__restrict__ kl_row, int M, int D, const float *__restrict__ where, const float *__restrict__ presence,
    const float *__restrict__ s0, const float *__restrict__ s1, float *__restrict__ pack, int T, int B, int S0, int S1) {
    const int width = T * D + T * 4 + T + S0 + S1;
    if ((int)blockIdx.x < gauss_blocks)
        gauss_fwd_body(blockIdx.x, gauss_blocks, pre, ld_pre, eps, raw_offset, 0, pl, ps, pl, ps, loc, scale, sample, kl_row,
                       M, D, pack, B, width);
    else
        baseline_pack_body(blockIdx.x - gauss_blocks, gridDim.x - gauss_blocks, nullptr, sample, where, presence, s0, s1,
                           pack, T, B, 0, D, S0, S1, T * D);
}

extern "C" int air_what_sample_pack(const float *pre, int ld_pre, const float *eps, float raw_offset, float p_loc,
                                    float p_scale, float *loc, float *scale, float *sample, float *kl_row, int D,
                                    const float *where, const float *presence, const float *state0, const float *state1,
                                    float *pack_out, int T, int B, int S0, int S1, float guard_eps, void *stream) {
    AIR_REQUIRE(pre && eps && loc && scale && sample && kl_row && where && presence && pack_out, AIR_E_NULL);
    AIR_REQUIRE((S0 == 0 || state0) && (S1 == 0 || state1), AIR_E_NULL);
    AIR_REQUIRE(T > 0 && B > 0 && D > 0 && ld_pre >= 2 * D && S0 >= 0 && S1 >= 0, AIR_E_SHAPE);
    const int M = T * B;
    const int gb = blocks_for((size_t)M * 64), pb = blocks_for((size_t)B * (T * 5 + S0 + S1));
    hipLaunchKernelGGL(what_sample_pack_kernel, dim3(gb + pb), dim3(PW_THREADS), 0, air_stream(stream), gb, pre, ld_pre, eps,
                       RawOffset(raw_offset, guard_eps), p_loc, p_scale, loc, scale, sample, kl_row, M, D, where, presence, state0, state1,
                       pack_out, T, B, S0, S1);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}


// ---- HBM-resident batch feeder (reference: tensors_from_data -> np.random.choice(n, batch_size) behind tf.py_func, data.py:121-158:
// a host round trip per step).  Here the dataset lives in HBM and the gather is a launch of the step itself: row b of the batch is
// item idx_b of the dataset, idx_b drawn with replacement from Philox(seed, stream 1, counter step*B + b) -- or, sequential,
// (step*B + b) mod n -- with `step` read from the DEVICE step counter, so a captured step (or several per graph replay) draws a
// fresh batch every time it runs and a resumed run continues the sequence.  idx_out (optional) receives the indices (labels).
__global__ __launch_bounds__(PW_THREADS) void batch_gather_kernel(const float *__restrict__ data, long long n_items, int item_floats,
                                                                 const uint64_t *__restrict__ seed_dev,
                                                                 const int64_t *__restrict__ step_dev, int shuffle,
                                                                 float *__restrict__ out, int B, int64_t *__restrict__ idx_out,
                                                                 int vec4) {
    const long long step = step_dev[0];
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const unsigned long long ctr = (unsigned long long)step * (unsigned long long)B + (unsigned long long)b;
        long long idx;
        if (shuffle) {
            uint32_t r[4];
            philox4x32(ctr, 1, seed_dev[0], r);
            const unsigned long long wide = ((unsigned long long)r[0] << 32) | r[1];
            idx = (long long)(((unsigned __int128)wide * (unsigned __int128)n_items) >> 64);        // uniform in [0, n)
        } else {
            idx = (long long)(ctr % (unsigned long long)n_items);
        }
        if (threadIdx.x == 0 && idx_out) idx_out[b] = idx;
        const float *src = data + (size_t)idx * item_floats;
        float *dst = out + (size_t)b * item_floats;
        if (vec4) {
            for (int q = threadIdx.x; q < (item_floats >> 2); q += PW_THREADS)
                reinterpret_cast<float4 *>(dst)[q] = reinterpret_cast<const float4 *>(src)[q];
        } else {
            for (int q = threadIdx.x; q < item_floats; q += PW_THREADS) dst[q] = src[q];
        }
    }
}
extern "C" int air_batch_gather(const float *dataset, long long n_items, int item_floats, const uint64_t *seed_dev,
                                const int64_t *step_dev, int shuffle, float *out, int B, int64_t *idx_out, void *stream) {
    AIR_REQUIRE(dataset && seed_dev && step_dev && out, AIR_E_NULL);
    AIR_REQUIRE(n_items > 0 && item_floats > 0 && B > 0, AIR_E_SHAPE);
    const int vec4 = (item_floats % 4 == 0) && air_aligned16(dataset) && air_aligned16(out);
    hipLaunchKernelGGL(batch_gather_kernel, dim3(B < 4096 ? B : 4096), dim3(PW_THREADS), 0, air_stream(stream), dataset, n_items,
                       item_floats, seed_dev, step_dev, shuffle ? 1 : 0, out, B, idx_out, vec4);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
