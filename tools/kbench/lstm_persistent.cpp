// Developer probe (VERDICT r03 "next" #4): the T recurrent LSTM steps of the train step's forward as ONE persistent launch with
// in-launch hand-offs of h_t, against the T launches of the product's air_lstm_step_fwd -- both replayed from a hipGraph.
//
// The product: one launch per step, tile = 16 batch rows x 4 hidden units (its 16 accumulator columns are the i,j,f,o gates of
// those units), 256 workgroups at batch 64 / hidden 256; every launch re-reads its 16 KB slice of W_h and the 16 KB of h rows.
// The persistent form: the same tiles, one workgroup each, resident for all T steps; the W_h slice stays in registers, the cell
// state in a register, and h_t travels as DATA-TAGGED 8-BYTE GRANULES {tag, value} (cdna_hip_programming.md, Guideline 16, recipe
// R2: ONE sc1 store per granule, the consumer re-reads its granules relaxed until every tag is current -- no flag, no fence):
// the 64 workgroups of a row group each publish 16 x 4 values, each of them sweeps the row group's 16 x 256 granules (32 KB) --
// MI355X_MICROARCH.md's price-list row "allgather" (32 KB: 2.9 us parked, 4.2 us streaming).
// Tags never repeat: tag = (launches of this row group so far) * T + t + 1, the count kept on the device (a kernel argument is
// frozen under graph replay).  Every spin is bounded (2 ms): a missed hand-off raises a flag instead of hanging the GPU.
// Results are compared bit for bit with the T launches.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kbench/lstm_persistent.cpp -o tools/kbench/bin/lstm_persistent
//   tools/kbench/bin/lstm_persistent [batch=64] [T=3]
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include "../../attend_infer_repeat_amd/csrc/gemm_kernels.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef __attribute__((address_space(1))) unsigned long long gu64;
struct PersArgs {
    const float *h0, *c0, *w_h, *gx;
    float *h_seq, *c_seq, *gate_act;         // h_seq / c_seq: [T+1, M, Hd] (slot 0 = the initial state), gate_act [T, M, 4 Hd]
    unsigned long long *hg;                  // granules [T, M, Hd]
    unsigned *launches;                      // per row group: launches so far
    int *flag;
    int M, Hd, T, ldw, ldgx;
    float fb;
};
constexpr int HD = 256, LDH = HD + 4;
__global__ __launch_bounds__(256) void lstm_fwd_persistent_kernel(PersArgs g) {
    constexpr int KW = 4, LDT = 20;
    __shared__ float s_tile[KW][16 * LDT];
    __shared__ __attribute__((aligned(16))) float s_h[16 * LDH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int tiles_u = HD >> 2;
    const int tm = blockIdx.x / tiles_u, tu = blockIdx.x - tm * tiles_u;
    const int m0 = tm * 16, u0 = tu * 4;
    const int colB = (li >> 2) * HD + u0 + (li & 3);
    // the workgroup's slice of W_h, stationary: wave `wave` owns the 16-deep chunks wave, wave + 4, wave + 8, wave + 12
    f32x4 fbw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) fbw[u] = ld_kstrided_full((gcf)g.w_h, g.ldw, colB, ((wave + u * KW) << 4) + 4 * lg);
    const unsigned epoch0 = __hip_atomic_load(g.launches + tm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * (unsigned)g.T;
    const int er = tid >> 2, eu = u0 + (tid & 3), em = m0 + er;
    const bool e_ok = tid < 64 && em < g.M;
    float e_gx[4] = {0.f, 0.f, 0.f, 0.f}, c_state = 0.f;
    if (e_ok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) e_gx[q] = g.gx[(size_t)em * g.ldgx + (size_t)q * HD + eu];
        c_state = g.c0[(size_t)em * HD + eu];
    }
    const size_t MH = (size_t)g.M * HD;
    for (int t = 0; t < g.T; ++t) {
        // ---- h_{t} rows m0 .. m0+15 into LDS: step 0 from the initial state, later steps from the granules of step t-1 ----
        if (t == 0) {
            for (int e = tid; e < 16 * (HD / 4); e += 256) {
                const int r = e / (HD / 4), q = e - r * (HD / 4);
                const int m = m0 + r < g.M ? m0 + r : g.M - 1;
                *reinterpret_cast<float4 *>(&s_h[r * LDH + 4 * q]) = *reinterpret_cast<const float4 *>(g.h0 + (size_t)m * HD + 4 * q);
            }
        } else {
            const unsigned want = epoch0 + (unsigned)t;                    // tag of step t-1's output
            gu64 *src = (gu64 *)(g.hg + (size_t)(t - 1) * MH + (size_t)m0 * HD);
            unsigned v[16];
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < 16; ++k) {                               // 16 x 256 granules, thread-strided
                    const unsigned long long x = __hip_atomic_load(src + tid + 256 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v[k] = (unsigned)x;
                    ok &= (unsigned)(x >> 32) == want;
                }
                if (__all(ok)) break;
                if (__builtin_amdgcn_s_memrealtime() - t_start > 200000ull) { if (lane == 0) *g.flag = 1; break; }   // 100 MHz: 2 ms
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int e = tid + 256 * k, r = e >> 8, col = e & 255;
                s_h[r * LDH + col] = __uint_as_float(v[k]);
            }
        }
        __syncthreads();
        f32x4 acc[1][1];
        acc[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = ((wave + u * KW) << 4) + 4 * lg;
            f32x4 fa[1], fb[1];
            fa[0] = *reinterpret_cast<const f32x4 *>(&s_h[li * LDH + k]);
            fb[0] = fbw[u];
            mfma_chunk<1, 1, false>(acc, fa, fb);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) s_tile[wave][(4 * lg + r) * LDT + li] = acc[0][0][r];
        __syncthreads();
        if (e_ok) {
            float pre[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int off = er * LDT + 4 * q + (tid & 3);
                pre[q] = ((s_tile[0][off] + s_tile[1][off]) + (s_tile[2][off] + s_tile[3][off])) + e_gx[q];
            }
            const float gi = sigmoid_acc(pre[0]);
            const float gj = tanhf(pre[1]);
            const float gf = sigmoid_acc(pre[2] + g.fb);
            const float go = sigmoid_acc(pre[3]);
            const float cn = gf * c_state + gi * gj;
            const float hn = tanhf(cn) * go;
            c_state = cn;
            const size_t e = (size_t)em * HD + eu;
            if (t + 1 < g.T)        // the hand-off first: ONE 8-byte sc1 store {tag, value}
                __hip_atomic_store((gu64 *)(g.hg + (size_t)t * MH + e), ((unsigned long long)(epoch0 + (unsigned)t + 1u) << 32) | __float_as_uint(hn),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g.c_seq[(size_t)(t + 1) * MH + e] = cn;
            g.h_seq[(size_t)(t + 1) * MH + e] = hn;
            float *ar = g.gate_act + (size_t)t * 4 * MH + (size_t)em * 4 * HD + eu;
            ar[0] = gi; ar[HD] = gj; ar[2 * (size_t)HD] = gf; ar[3 * (size_t)HD] = go;
        }
        // (s_h / s_tile are rewritten only after the next step's sweep + barrier: every reader has passed the barrier above)
    }
    // the row group's launch count: by its first workgroup, after it consumed the whole group's last hand-off (so every member has
    // read the count already)
    if (tu == 0 && tid == 0) __hip_atomic_fetch_add(g.launches + tm, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static float frand() { return (float)rand() / (float)RAND_MAX * 2.f - 1.f; }

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, T = argc > 2 ? atoi(argv[2]) : 3;
    const int M = B, Hd = HD;
    if (M % 16) { printf("batch must be a multiple of 16\n"); return 1; }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    auto dev = [&](size_t n, bool rnd, float scale) {
        std::vector<float> h(n, 0.f);
        if (rnd) for (auto &x : h) x = frand() * scale;
        float *d = nullptr;
        if (hipMalloc(&d, n * 4) != hipSuccess) return (float *)nullptr;
        hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        return d;
    };
    float *w_h = dev((size_t)Hd * 4 * Hd, true, 0.06f), *gx = dev((size_t)M * 4 * Hd, true, 1.0f);
    const size_t MH = (size_t)M * Hd;
    float *h_a = dev((T + 1) * MH, true, 0.5f), *c_a = dev((T + 1) * MH, true, 0.5f), *act_a = dev((size_t)T * 4 * MH, false, 0);
    float *h_b = dev((T + 1) * MH, false, 0), *c_b = dev((T + 1) * MH, false, 0), *act_b = dev((size_t)T * 4 * MH, false, 0);
    CK(hipMemcpy(h_b, h_a, MH * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(c_b, c_a, MH * 4, hipMemcpyDeviceToDevice));
    unsigned long long *hg; CK(hipMalloc(&hg, (size_t)T * MH * 8)); CK(hipMemset(hg, 0, (size_t)T * MH * 8));
    unsigned *launches; CK(hipMalloc(&launches, 64 * 4)); CK(hipMemset(launches, 0, 64 * 4));
    int *flag; CK(hipMalloc(&flag, 4)); CK(hipMemset(flag, 0, 4));
    const int tiles = (M / 16) * (Hd / 4);

    auto base_steps = [&](int nsteps) {
        for (int t = 0; t < nsteps; ++t)
            if (air_lstm_step_fwd(h_a + t * MH, c_a + t * MH, w_h, 4 * Hd, gx, 4 * Hd, h_a + (t + 1) * MH, c_a + (t + 1) * MH, act_a + (size_t)t * 4 * MH,
                                  M, Hd, 1.0f, 0, st) != 0) { printf("air_lstm_step_fwd failed\n"); exit(1); }
    };
    auto pers = [&](int nsteps) {
        PersArgs a = {h_b, c_b, w_h, gx, h_b, c_b, act_b, hg, launches, flag, M, Hd, nsteps, 4 * Hd, 4 * Hd, 1.0f};
        hipLaunchKernelGGL(lstm_fwd_persistent_kernel, dim3(tiles), dim3(256), 0, st, a);
    };
    // ---- results: bit for bit ----
    base_steps(T); pers(T);
    CK(hipStreamSynchronize(st));
    std::vector<float> ra((T + 1) * MH), rb((T + 1) * MH), ca((T + 1) * MH), cb((T + 1) * MH), ga((size_t)T * 4 * MH), gb((size_t)T * 4 * MH);
    CK(hipMemcpy(ra.data(), h_a, ra.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(rb.data(), h_b, rb.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ca.data(), c_a, ca.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(cb.data(), c_b, cb.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ga.data(), act_a, ga.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gb.data(), act_b, gb.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < ra.size(); ++i) bad += (ra[i] != rb[i]) + (ca[i] != cb[i]);
    for (size_t i = 0; i < ga.size(); ++i) bad += ga[i] != gb[i];
    int hflag = 0; CK(hipMemcpy(&hflag, flag, 4, hipMemcpyDeviceToHost));
    printf("batch %d, T %d, %d workgroups: persistent vs %d launches: %zu differing values, timeout flag %d\n", B, T, tiles, T, bad, hflag);

    // ---- timing: hipGraph replays of [T launches] and of [1 persistent launch], steps 1..T ----
    auto time_graph = [&](auto fn, int reps, float *us) -> int {
        hipGraph_t gr; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        fn();
        CK(hipStreamEndCapture(st, &gr));
        CK(hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ex, st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ex, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        *us = ms * 1e3f / reps;
        CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(gr));
        return 0;
    };
    printf("%6s %16s %16s   (us per graph replay; a replay itself costs ~8 us + 1.7 us per node)\n", "steps", "launches", "persistent");
    for (int n = 1; n <= T; ++n) {
        float ub = 0, up = 0;
        if (time_graph([&] { base_steps(n); }, 2000, &ub)) return 1;
        if (time_graph([&] { pers(n); }, 2000, &up)) return 1;
        printf("%6d %16.2f %16.2f\n", n, ub, up);
    }
    // the same T steps embedded in a longer dependent chain (10 repetitions per replay): the per-replay overhead amortised
    {
        float ub = 0, up = 0;
        if (time_graph([&] { for (int r = 0; r < 10; ++r) base_steps(T); }, 500, &ub)) return 1;
        if (time_graph([&] { for (int r = 0; r < 10; ++r) pers(T); }, 500, &up)) return 1;
        printf("10 x %d steps per replay: %.2f us per %d steps as launches, %.2f as one persistent launch\n", T, ub / 10, T, up / 10);
    }
    CK(hipMemcpy(&hflag, flag, 4, hipMemcpyDeviceToHost));
    printf("timeout flag after timing: %d\n", hflag);
    return 0;
}
