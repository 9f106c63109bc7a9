/* Scalar-loop C restatement of the dense / recurrent / probabilistic pieces of the AIR step -- TEST INFRASTRUCTURE ONLY.
 *
 * An independent second coding of what oracle/air_oracle.py states with torch ops (as oracle/st_loops.c is for the spatial
 * transformer): plain loops, no BLAS, no autograd, so a slip in one coding shows up as a disagreement with the other
 * (tests/test_oracle_net_loops.py).  Each function cites the reference lines it follows (attend_infer_repeat/...).
 * PARITY STATUS: like the torch oracle, unpinned w.r.t. the real TF 1.1 / Sonnet 1.1 stack (not runnable here) except for the
 * num-steps math, whose known answers (test/prior_test.py) are checked in tests/test_oracle_net_loops.py.
 * Build: oracle/Makefile -> oracle/_build/net_loops_{f32,f64}.so   (REAL = float | double)
 */
#include <math.h>
#include <stddef.h>

#ifndef REAL
#define REAL double
#endif
typedef REAL real;

static real elu1(real v) { return v > 0 ? v : (real)expm1((double)v); }
static real sigm(real v) { return (real)(1.0 / (1.0 + exp(-(double)v))); }
static real softplus1(real v) { return v > 20 ? v : (real)log1p(exp((double)v)); }

/* Affine = transfer(x.W + b), neural.py:56-60; W is [K,N] (Sonnet layout), ELU(alpha=1) when elu != 0 */
void affine(const real *x, const real *w, const real *b, real *y, int M, int K, int N, int elu) {
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            real acc = 0;
            for (int k = 0; k < K; ++k) acc += x[(size_t)m * K + k] * w[(size_t)k * N + n];
            acc += b[n];
            y[(size_t)m * N + n] = elu ? elu1(acc) : acc;
        }
}

/* snt.LSTM step (mnist_model.py:35, cell.py:126-127): gates = [x,h].W + b, order i,j,f,o;
 * c' = sigmoid(f + forget_bias) c + sigmoid(i) tanh(j);  h' = tanh(c') sigmoid(o).  W is [I+H, 4H]. */
void lstm_step(const real *x, const real *h, const real *c, const real *w, const real *b, real *h2, real *c2,
               int M, int I, int H, real forget_bias) {
    for (int m = 0; m < M; ++m)
        for (int u = 0; u < H; ++u) {
            real g[4];
            for (int q = 0; q < 4; ++q) {
                const int col = q * H + u;
                real acc = 0;
                for (int k = 0; k < I; ++k) acc += x[(size_t)m * I + k] * w[(size_t)k * 4 * H + col];
                for (int k = 0; k < H; ++k) acc += h[(size_t)m * H + k] * w[(size_t)(I + k) * 4 * H + col];
                g[q] = acc + b[col];
            }
            const real cn = sigm(g[2] + forget_bias) * c[(size_t)m * H + u] + sigm(g[0]) * (real)tanh((double)g[1]);
            c2[(size_t)m * H + u] = cn;
            h2[(size_t)m * H + u] = (real)tanh((double)cn) * sigm(g[3]);
        }
}

/* loc, scale = softplus(raw + offset), sample = loc + scale * eps from pre = [loc | raw]  (modules.py:17-24, cell.py:154-156),
 * and KL( N(loc, scale) || N(pm, ps) ) summed over the D dims of a row (model.py:174-187):
 *   (mu - pm)^2 / (2 ps^2) + 0.5 (s^2/ps^2 - 1 - log(s^2/ps^2)) */
void gauss_sample_kl(const real *pre, const real *eps, real offset, real pm, real ps, real *loc, real *scale, real *sample,
                     real *kl_row, int M, int D) {
    for (int m = 0; m < M; ++m) {
        real kl = 0;
        for (int d = 0; d < D; ++d) {
            const real mu = pre[(size_t)m * 2 * D + d];
            const real s = softplus1(pre[(size_t)m * 2 * D + D + d] + offset);
            loc[(size_t)m * D + d] = mu; scale[(size_t)m * D + d] = s;
            sample[(size_t)m * D + d] = mu + s * eps[(size_t)m * D + d];
            const real ratio = (s * s) / (ps * ps);
            kl += (mu - pm) * (mu - pm) / (2 * ps * ps) + (real)0.5 * (ratio - 1 - (real)log((double)ratio));
        }
        kl_row[m] = kl;
    }
}

/* q(n) from per-step presence probabilities p[B,T] (prior.py:35-68): q = [1-p1, p1(1-p2), ..., prod p], renormalised;
 * always double (the reference computes this in float64, prior.py:63) */
void numsteps_posterior(const double *p, double *q, int B, int T) {
    for (int b = 0; b < B; ++b) {
        double run = 1.0, sum = 0.0;
        for (int n = 0; n <= T; ++n) {
            const double u = n < T ? run * (1.0 - p[(size_t)b * T + n]) : run;
            q[(size_t)b * (T + 1) + n] = u;
            sum += u;
            if (n < T) run *= p[(size_t)b * T + n];
        }
        for (int n = 0; n <= T; ++n) q[(size_t)b * (T + 1) + n] /= sum;
    }
}

/* geometric prior pi(n) = (1-s) s^n, n = 0..T, NOT renormalised, s clipped to [1e-7, 1-1e-15] (prior.py:26-32) */
void geometric_prior(double success_prob, double *pi, int T) {
    double s = success_prob < 1e-7 ? 1e-7 : (success_prob > 1.0 - 1e-15 ? 1.0 - 1e-15 : success_prob);
    const double probs = 1.0 - s;
    for (int n = 0; n <= T; ++n) pi[n] = exp((double)n * log1p(-probs) + log(probs));
}

/* tabular KL per sample: sum_n q log(q / pi) over q > 0 only (prior.py:71-90) */
void tabular_kl(const double *q, const double *pi, double *kl, int B, int T) {
    for (int b = 0; b < B; ++b) {
        double acc = 0.0;
        for (int n = 0; n <= T; ++n) {
            const double v = q[(size_t)b * (T + 1) + n];
            if (v > 0.0) acc += v * log(v / pi[n]);
        }
        kl[b] = acc;
    }
}

/* centred RMSProp with momentum, TF semantics (model.py:265,355-367): ms <- d ms + (1-d) g^2; mg <- d mg + (1-d) g;
 * mom <- m mom + lr g / sqrt(ms - mg^2 + eps); p <- p - mom */
void rmsprop_centered(real *p, const real *g, real *ms, real *mg, real *mom, long n, real lr, real decay, real momentum,
                      real eps) {
    for (long i = 0; i < n; ++i) {
        ms[i] = decay * ms[i] + (1 - decay) * g[i] * g[i];
        mg[i] = decay * mg[i] + (1 - decay) * g[i];
        mom[i] = momentum * mom[i] + lr * g[i] / (real)sqrt((double)(ms[i] - mg[i] * mg[i] + eps));
        p[i] -= mom[i];
    }
}
