// Developer micro-benchmark for the latency-regime ST kernels (not part of the product; built and run by hand on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAIR_TRACE tools/kbench/st_trace.cpp -o gpurun_out/st_trace && gpurun_out/st_trace
// Includes the kernel translation unit directly, so the kernels measured are the shipped ones; with -DAIR_TRACE thread 0 of
// every workgroup stamps s_memrealtime (100 MHz, chip-wide) at the phase boundaries marked AIR_TR(i) in the kernels.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <string>
#include <functional>
#include "../../attend_infer_repeat_amd/csrc/st_kernels.hip"
#define lin_step lin_step_cv          // (both translation units define this host helper)
#include "../../attend_infer_repeat_amd/csrc/canvas_kernels.hip"
#undef lin_step

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static float frand() { return (float)rand() / (float)RAND_MAX; }
template <typename T> static T *dev(const std::vector<T> &v) {
    T *p; CK(hipMalloc(&p, v.size() * sizeof(T))); CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return p;
}
static float *devz(size_t n) { float *p; CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 0, n * 4)); return p; }

template <typename F> static double time_us(F fn, int reps, hipStream_t st) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 20; ++i) fn();
    std::vector<double> r;
    for (int rep = 0; rep < 7; ++rep) {
        CK(hipEventRecord(a, st));
        for (int i = 0; i < reps; ++i) fn();
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); r.push_back(ms * 1e3 / reps);
    }
    std::sort(r.begin(), r.end());
    return r[r.size() / 2];
}

#ifdef AIR_TRACE
static void dump_trace(const char *name, int nblocks, int nphase) {
    std::vector<unsigned long long> h(AIR_TRACE_BLOCKS * AIR_TRACE_PHASES);
    CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(air_trace), h.size() * 8));
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < nblocks; ++b) if (h[b * AIR_TRACE_PHASES]) t0 = std::min(t0, h[b * AIR_TRACE_PHASES]);
    printf("trace %-22s (10 ns ticks since first workgroup start; mean / max over %d workgroups)\n", name, nblocks);
    for (int ph = 0; ph < nphase; ++ph) {
        double sum = 0; unsigned long long mx = 0; int cnt = 0;
        for (int b = 0; b < nblocks; ++b) { unsigned long long v = h[b * AIR_TRACE_PHASES + ph]; if (!v) continue; v -= t0; sum += v; mx = std::max(mx, v); ++cnt; }
        if (cnt) {
            // per quarter of the grid: when do later workgroups reach this phase?
            double qs[4] = {0, 0, 0, 0}; int qc[4] = {0, 0, 0, 0};
            for (int b = 0; b < nblocks; ++b) { unsigned long long v = h[b * AIR_TRACE_PHASES + ph]; if (!v) continue; const int q = (int)((long)b * 4 / nblocks); qs[q] += v - t0; ++qc[q]; }
            printf("   phase %d: mean %7.2f us  max %7.2f us  (%d wgs)   by grid quarter:", ph, sum / cnt * 0.01, mx * 0.01, cnt);
            for (int q = 0; q < 4; ++q) printf(" %6.2f", qc[q] ? qs[q] / qc[q] * 0.01 : 0.0);
            printf("\n");
        }
    }
    std::vector<unsigned long long> z(h.size(), 0);
    CK(hipMemcpyToSymbol(HIP_SYMBOL(air_trace), z.data(), z.size() * 8));
}
#endif

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, T = argc > 2 ? atoi(argv[2]) : 3;
    const int H = argc > 3 ? atoi(argv[3]) : 50, W = H, h = argc > 4 ? atoi(argv[4]) : 20, w = h;
    const int HW = H * W, hw = h * w, M = T * B, A = 50, Kt = 256, Ks = 64;
    srand(1);
    std::vector<float> obs(B * HW), where(M * 4), glm(M * hw), pres(M), dgl(M * hw), trh(M * Kt), trw(Kt * 8), trb(8), sth(M * Ks),
        stw(Ks), stb(1), eps(M * 4), u(M), imp(B), base(B), logp(B);
    for (auto &x : obs) x = frand() < 0.15f ? frand() : 0.f;
    const float s_lo = argc > 7 ? atof(argv[7]) : 0.4f, s_hi = argc > 8 ? atof(argv[8]) : 0.7f;
    for (int k = 0; k < M; ++k) { where[4 * k] = s_lo + (s_hi - s_lo) * frand(); where[4 * k + 1] = 0.6f * frand() - 0.3f; where[4 * k + 2] = s_lo + (s_hi - s_lo) * frand(); where[4 * k + 3] = 0.6f * frand() - 0.3f; }
    for (auto &x : glm) x = frand() - 0.5f;
    for (auto &x : dgl) x = frand() - 0.5f;
    for (auto &x : pres) x = frand() < 0.7f ? 1.f : 0.f;
    for (auto &x : trh) x = frand() - 0.5f;
    for (auto &x : trw) x = 0.1f * (frand() - 0.5f);
    for (auto &x : sth) x = frand() - 0.5f;
    for (auto &x : stw) x = 0.1f * (frand() - 0.5f);
    for (auto &x : eps) x = frand() - 0.5f;
    for (auto &x : u) x = frand();
    for (auto &x : imp) x = -500.f + 100.f * frand();
    for (auto &x : base) x = frand();
    for (auto &x : logp) x = -frand();
    std::vector<double> prior(T + 1); for (int n = 0; n <= T; ++n) prior[n] = 0.5 * pow(0.5, n);
    float *d_obs = dev(obs), *d_where = dev(where), *d_glm = dev(glm), *d_pres = dev(pres), *d_dgl = dev(dgl), *d_trh = dev(trh),
          *d_trw = dev(trw), *d_trb = dev(trb), *d_sth = dev(sth), *d_stw = dev(stw), *d_stb = dev(stb), *d_eps = dev(eps), *d_u = dev(u),
          *d_imp = dev(imp), *d_base = dev(base), *d_logp = dev(logp);
    double *d_prior = dev(prior);
    float *d_steps = devz((size_t)M * HW), *d_final = devz((size_t)B * HW), *d_rec = devz(B), *d_dglm = devz((size_t)M * hw),
          *d_dwhere = devz(M * 4 * 4), *d_nvil = devz(4), *d_dlogp = devz(B), *d_dbase = devz(B), *d_pre = devz(M * 8), *d_logit = devz(M),
          *d_loc = devz(M * 4), *d_scale = devz(M * 4), *d_wh2 = devz(M * 4), *d_klrow = devz(M), *d_prob = devz(M), *d_pr2 = devz(M),
          *d_q = devz(B * (T + 1)), *d_klps = devz(B), *d_lp2 = devz(B), *d_stepw = devz(M), *d_glimpse = devz((size_t)M * hw),
          *d_dwr = devz(M * 4), *d_dpre = devz(M * 8), *d_dlogit = devz(M), *d_kla = devz(M), *d_klb = devz(M);
    hipStream_t st; CK(hipStreamCreate(&st));
    const int keep = argc > 5 ? atoi(argv[5]) : 1;
    const int NB = argc > 6 ? atoi(argv[6]) : air_canvas_unroll_bands(B, H);
    const int NS = getenv("NS") ? atoi(getenv("NS")) : 1;      // workgroups per backward unit of the fused canvas launch
    float *d_recp = devz((size_t)NB * B);
    { std::vector<float> ip((size_t)NB * B); for (auto &x : ip) x = (-500.f + 100.f * frand()) / NB; CK(hipMemcpy(d_recp, ip.data(), ip.size() * 4, hipMemcpyHostToDevice)); }
    auto f_cfwd = [&] { air_canvas_unroll_fwd_banded(d_glm, d_where, d_pres, d_obs, keep ? d_steps : nullptr, d_final, d_recp, NB, T, B, H, W, h, w, 0.5f, 0.3f, st); };
    auto f_cfwd1 = [&] { air_canvas_unroll_fwd(d_glm, d_where, d_pres, d_obs, keep ? d_steps : nullptr, d_final, d_rec, T, B, H, W, h, w, 0.5f, 0.3f, st); };
    auto f_cbwd0 = [&] { air_canvas_unroll_bwd(d_glm, d_where, d_pres, d_obs, d_final, d_dglm, d_dwhere, T, B, H, W, h, w, 0.5f, 0.3f, 1.0f / B, st); };
    auto f_cbwd_rc = [&] { air_canvas_unroll_bwd(d_glm, d_where, d_pres, d_obs, nullptr, d_dglm, d_dwhere, T, B, H, W, h, w, 0.5f, 0.3f, 1.0f / B, st); };
    auto f_cfused = [&] { air_canvas_unroll_fwd_bwd(d_glm, d_where, d_pres, d_obs, keep ? d_steps : nullptr, d_final, d_recp, NB, d_dglm, d_dwhere, NS, T, B, H, W, h, w, 0.5f, 0.3f, 1.0f / B, st); };
    auto f_rfwd = [&] { air_st_read_fwd(d_obs, d_where, d_glimpse, M, B, H, W, h, w, st); };
    const int prec = getenv("PREC") ? atoi(getenv("PREC")) : 0;   // 1: operands rounded to bf16 (the configs[4] plan)
    auto f_attend = [&] { air_attend_fwd(d_trh, d_trw, d_trb, Kt, d_sth, d_stw, d_stb, Ks, d_pre, d_logit, d_eps, 0.5f, 0.f, 1.f, 0.f, 1.f, d_loc, d_scale, d_wh2, d_klrow,
                                         d_u, 0.f, 0.f, d_prior, d_prob, d_pr2, d_q, d_klps, d_lp2, d_stepw, d_obs, d_glimpse, T, B, H, W, h, w, prec, 1e-3f, st); };
    const int n_attend = ((long)T * B > 2048 && T > 1 ? B : M) + (B + 63) / 64;
    struct { const char *n; std::function<void()> f; int nb; } K[] = {
        {"canvas_unroll_fwd_banded", f_cfwd, B * NB}, {"canvas_unroll_fwd(1 band)", f_cfwd1, B}, {"canvas_unroll_bwd", f_cbwd0, M},
        {"canvas_unroll_bwd(recompute)", f_cbwd_rc, M}, {"canvas_fused(fwd+bwd)", f_cfused, B * NB + M},
        {"st_read_fwd", f_rfwd, B}, {"attend_fwd", f_attend, n_attend}};
    f_cfwd(); CK(hipStreamSynchronize(st));
    printf("B=%d T=%d %dx%d glimpse %dx%d keep_steps=%d bands=%d\n", B, T, H, W, h, w, keep, NB);
    const char *only = getenv("ONLY");
    for (auto &k : K) {
        if (only && !strstr(k.n, only)) continue;
        double us = time_us(k.f, 200, st);
        printf("%-26s %7.2f us/launch (median of 7 x 200 back-to-back)\n", k.n, us);
#ifdef AIR_TRACE
        CK(hipStreamSynchronize(st));
        { std::vector<unsigned long long> z(AIR_TRACE_BLOCKS * AIR_TRACE_PHASES, 0); CK(hipMemcpyToSymbol(HIP_SYMBOL(air_trace), z.data(), z.size() * 8)); }
        k.f(); CK(hipStreamSynchronize(st));
        dump_trace(k.n, std::min(k.nb, (int)AIR_TRACE_BLOCKS), AIR_TRACE_PHASES);
#endif
    }
    return 0;
}
