// Step prologue as a device body: Philox noise for the whole step + annealed geometric prior (float64) + tiling of the
// trainable LSTM initial state over the batch.  Runs as its own launch (air_step_prologue) or as extra workgroups of the first
// LSTM step's launch (air_lstm_step_fwd_prologue) -- nothing before that launch's successors needs its outputs.
#pragma once
#include <math.h>
#include "air_common.h"

#ifndef PW_THREADS
#define PW_THREADS 256
#endif

// ---- Philox4x32-10 ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4x32(uint64_t ctr, uint64_t stream_id, uint64_t seed, uint32_t (&out)[4]) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }          // [0,1)
__device__ __forceinline__ float u01_open(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)


struct PrologueArgs {
    float *normal; size_t n_normal; float *uniform; size_t n_uniform;
    const uint64_t *rng_state; int rng_blocks;
    const int64_t *gstep; int anneal_type; double init, fin, anneal_steps, hold_for, steps_div;
    double *prior; int T;
    const float *h0, *c0; float *h_out, *c_out; int B, Hd, tile_blocks;
};
static inline int prologue_pw_blocks(size_t n) {
    size_t b = (n + PW_THREADS - 1) / PW_THREADS;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}
static inline PrologueArgs make_prologue_args(float *normal, size_t n_normal, float *uniform, size_t n_uniform,
                                              const uint64_t *rng_state, const int64_t *gstep, int anneal_type,
                                              double init, double fin, double anneal_steps, double hold_for,
                                              double steps_div, double *prior, int T, const float *h0, const float *c0,
                                              float *h_out, float *c_out, int B, int Hd) {
    PrologueArgs a;
    a.normal = normal; a.n_normal = n_normal; a.uniform = uniform; a.n_uniform = n_uniform; a.rng_state = rng_state;
    const size_t q = (n_normal + 3) / 4 + (n_uniform + 3) / 4;
    a.rng_blocks = q ? prologue_pw_blocks(q) : 1;
    a.gstep = gstep; a.anneal_type = anneal_type; a.init = init; a.fin = fin; a.anneal_steps = anneal_steps;
    a.hold_for = hold_for; a.steps_div = steps_div; a.prior = prior; a.T = T; a.h0 = h0; a.c0 = c0; a.h_out = h_out;
    a.c_out = c_out; a.B = B; a.Hd = Hd;
    a.tile_blocks = prologue_pw_blocks((size_t)B * Hd);
    return a;
}
static inline int prologue_blocks(const PrologueArgs &a) { return a.rng_blocks + 1 + a.tile_blocks; }

// roles by (virtual) block index: [0, rng_blocks) noise, rng_blocks: prior, the rest: tiling
__device__ __forceinline__ void step_prologue_body(const PrologueArgs &a, int bid, int nblocks) {
    if (bid < a.rng_blocks) {
        const uint64_t seed = a.rng_state[0], offset = a.rng_state[1];
        const size_t q_normal = (a.n_normal + 3) / 4, q_uniform = (a.n_uniform + 3) / 4;
        for (size_t q = (size_t)bid * PW_THREADS + threadIdx.x; q < q_normal + q_uniform; q += (size_t)a.rng_blocks * PW_THREADS) {
            uint32_t r[4];
            philox4x32(offset + q, 0, seed, r);
            if (q < q_normal) {
                float z[4];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float rad = sqrtf(-2.0f * logf(u01_open(r[2 * k])));
                    float sn, cs;
                    sincosf(6.283185307179586f * u01(r[2 * k + 1]), &sn, &cs);
                    z[2 * k] = rad * cs; z[2 * k + 1] = rad * sn;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) if (4 * q + k < a.n_normal) a.normal[4 * q + k] = z[k];
            } else {
                const size_t qq = q - q_normal;
#pragma unroll
                for (int k = 0; k < 4; ++k) if (4 * qq + k < a.n_uniform) a.uniform[4 * qq + k] = u01(r[k]);
            }
        }
    } else if (bid == a.rng_blocks) {
        if (threadIdx.x == 0) {
            double s = a.init;
            if (a.anneal_type != 0) {
                double step = (double)a.gstep[0] - a.hold_for;
                if (step < 0.0) step = 0.0;
                double val = (a.anneal_type == 1)
                                 ? a.init * pow(pow(a.fin / a.init, a.steps_div / a.anneal_steps), step / a.steps_div)
                                 : a.fin + (a.init - a.fin) * (1.0 - step / a.anneal_steps);
                s = val > a.fin ? val : a.fin;
            }
            s = s < 1e-7 ? 1e-7 : (s > 1.0 - 1e-15 ? 1.0 - 1e-15 : s);
            const double probs = 1.0 - s;
            for (int n = 0; n <= a.T; ++n) a.prior[n] = exp((double)n * log1p(-probs) + log(probs));
        }
    } else {
        const int tb = bid - a.rng_blocks - 1, ntb = nblocks - a.rng_blocks - 1;
        const size_t n = (size_t)a.B * a.Hd;
        for (size_t i = (size_t)tb * PW_THREADS + threadIdx.x; i < n; i += (size_t)ntb * PW_THREADS) {
            a.h_out[i] = a.h0[i % a.Hd];
            a.c_out[i] = a.c0[i % a.Hd];
        }
    }
}
