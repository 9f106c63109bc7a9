// Developer probe for the bf16 weight-gradient launches of the throughput regime (batch 1024: dW[Kin, Nout] = X^T . dY over the
// M = T*B = 3072 rows; VERDICT r03 "next" #3).  The shipped form (air_gemm_grouped -> gemm_big_group_wide16_tn_kernel: 64x64 output
// tiles, 8 waves split K inside the workgroup) against a 128x128-tile form:
//   * a workgroup = 4 waves = 2x2 quadrants of 64x64; every wave fetches its own 64-row / 64-column operand fragments
//     register-direct exactly as the shipped kernel does (8-byte loads along the contiguous dimension, v_perm transposes,
//     v_mfma_f32_16x16x32_bf16) -- the two waves that need the same fragment hit the vector L1 (tools/kbench/l1_share.cpp), so the
//     workgroup moves half the unique bytes per flop, with no LDS staging and no in-workgroup reduction;
//   * K is split ACROSS workgroups (slices of `slice` rows); a slice writes its fp32 partial tile to slab s of the problem, and a
//     second, tiny launch adds the slabs in slab order (fixed order => bitwise reproducible) -- in the engine that sum would ride in
//     the optimiser launch, which reads every gradient once anyway.
// Both on the problems of the first weight-gradient launch of BASELINE configs[4] (nine problems, K = 3072), operands as bf16 mirrors.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kbench/dw_bench.cpp attend_infer_repeat_amd/csrc/loss_kernels.hip -o tools/kbench/bin/dw_bench
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../../attend_infer_repeat_amd/csrc/gemm_kernels.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static float frand() { return (float)rand() / (float)RAND_MAX * 2.f - 1.f; }

struct DwItem { int p, m0, n0, k0, k1, slab; };
struct DwProblem { const unsigned short *A16, *B16; float *slabs; int M, N, K, lda, ldb; };   // A16 [K, M] (lda), B16 [K, N] (ldb): k-strided both
struct DwArgs { const DwProblem *prob; const DwItem *items; };

template <int U>
__global__ __launch_bounds__(256) void dw128_kernel(DwArgs a) {
    const DwItem it = a.items[blockIdx.x];
    if (it.p < 0) return;
    const DwProblem pr = a.prob[it.p];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int m0 = it.m0 + 64 * (wave >> 1), n0 = it.n0 + 64 * (wave & 1);
    const gch hA = (gch)pr.A16, hB = (gch)pr.B16;
    int offA = m0 + 4 * li; if (offA > pr.M - 4) offA = pr.M - 4;
    int offB = n0 + 4 * li; if (offB > pr.N - 4) offB = pr.N - 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int c_end = it.k1 >> 5;
#pragma nounroll
    for (int c = it.k0 >> 5; c < c_end; c += U) {
        u32x4 fa[U][4], fb[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int cu = c + u; if (cu > c_end - 1) cu = c_end - 1;
            const int k = (cu << 5) + 8 * lg;
            u32x2 wa[8], wb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) wa[j] = *(gcu2)(hA + (size_t)(k + j) * pr.lda + offA);
#pragma unroll
            for (int j = 0; j < 8; ++j) wb[j] = *(gcu2)(hB + (size_t)(k + j) * pr.ldb + offB);
            fa[u][0] = tr16<0>(wa); fa[u][1] = tr16<1>(wa); fa[u][2] = tr16<2>(wa); fa[u][3] = tr16<3>(wa);
            fb[u][0] = tr16<0>(wb); fb[u][1] = tr16<1>(wb); fb[u][2] = tr16<2>(wb); fb[u][3] = tr16<3>(wb);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c + u >= c_end) break;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[u][i]), __builtin_bit_cast(bf16x8, fb[u][j]),
                                                                       acc[i][j], 0, 0, 0);
        }
    }
    // accumulator (a, b)[r] of lane (li, lg) = C[m0 + 4*(4*lg + r) + a][n0 + 4*li + b]  (rows and columns interleaved by four)
    float *C = pr.slabs + (size_t)it.slab * pr.M * pr.N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 4 * (4 * lg + r) + i, col = n0 + 4 * li;
            if (row < pr.M && col + 3 < pr.N)          // (M, N multiples of 4: a clamped fragment lane is entirely out of range)
                *(f32x4 *)(C + (size_t)row * pr.N + col) = (f32x4){acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
        }
}
// LDS-staged form: the 128x128 tile's operand chunks ([32 k][128 m] and [32 k][128 n], bf16, k-major as they sit in memory) are
// fetched ONCE per workgroup with 16-byte loads (a quarter of the load instructions per flop of the register-direct forms, half
// the unique bytes of the 64x64 tile), double-buffered in LDS, and the MFMA fragments -- 8 consecutive k per lane -- come out of
// the gfx950 transposing LDS read: ds_read_b64_tr_b16 hands lane (li, lg) column li of a [4 k][16 m] block whose rows the 16
// lanes of the group address (tools/kbench/tr16_probe.cpp), two reads per fragment, sixteen per sixteen MFMAs.
typedef short v4s16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s16 *lds_v4;
constexpr int LP = 128 + 16;                       // LDS row pitch in elements (288 B: rows 4 apart land 8 banks apart)
template <bool A8, bool B8>                        // A8 / B8: the operand's row length is a multiple of 8 elements (16-byte loads)
__global__ __launch_bounds__(256) void dw_lds_kernel(DwArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short sA[2][32 * LP], sB[2][32 * LP];
    const DwItem it = a.items[blockIdx.x];
    if (it.p < 0) return;
    const DwProblem pr = a.prob[it.p];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const gch hA = (gch)pr.A16, hB = (gch)pr.B16;
    // global -> register staging: thread (row = tid >> 4, group = tid & 15) fetches 8 elements of rows row and row + 16
    const int lrow = tid >> 4, lcol = 8 * (tid & 15);
    auto fetch = [&](gch h, int ld, int lim, int c0, int k, bool w8, u32x4 &v) {
        const int col = c0 + lcol;
        const size_t base = (size_t)k * ld;
        if (w8) {
            const int cc = col + 8 <= lim ? col : (lim >= 8 ? lim - 8 : 0);
            v = *(gcu4)(h + base + cc);
            if (col + 8 > lim) v = (u32x4){0u, 0u, 0u, 0u};
        } else {
            const int c_lo = col + 4 <= lim ? col : lim - 4, c_hi = col + 8 <= lim ? col + 4 : lim - 4;
            u32x2 lo = *(gcu2)(h + base + c_lo), hi = *(gcu2)(h + base + c_hi);
            if (col + 4 > lim) lo = (u32x2){0u, 0u};
            if (col + 8 > lim) hi = (u32x2){0u, 0u};
            v = (u32x4){lo.x, lo.y, hi.x, hi.y};
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int c_begin = it.k0 >> 5, c_end = it.k1 >> 5;
    u32x4 ra0, ra1, rb0, rb1;
    auto fetch_chunk = [&](int c) {
        const int k = c << 5;
        fetch(hA, pr.lda, pr.M, it.m0, k + lrow, A8, ra0); fetch(hA, pr.lda, pr.M, it.m0, k + lrow + 16, A8, ra1);
        fetch(hB, pr.ldb, pr.N, it.n0, k + lrow, B8, rb0); fetch(hB, pr.ldb, pr.N, it.n0, k + lrow + 16, B8, rb1);
    };
    auto stash = [&](int buf) {
        *(u32x4 *)&sA[buf][lrow * LP + lcol] = ra0; *(u32x4 *)&sA[buf][(lrow + 16) * LP + lcol] = ra1;
        *(u32x4 *)&sB[buf][lrow * LP + lcol] = rb0; *(u32x4 *)&sB[buf][(lrow + 16) * LP + lcol] = rb1;
    };
    fetch_chunk(c_begin);
    stash(0);
    __syncthreads();
    const int rowsel = 8 * lg + (li >> 2), colsel = 4 * (li & 3);
    const int wm = 64 * (wave >> 1), wn = 64 * (wave & 1);
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = (c - c_begin) & 1;
        if (c + 1 < c_end) fetch_chunk(c + 1);                 // in flight during this chunk's MFMAs
        bf16x8 fa[4], fb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const v4s16 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)&sA[buf][rowsel * LP + wm + 16 * t + colsel]);
            const v4s16 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)&sA[buf][(rowsel + 4) * LP + wm + 16 * t + colsel]);
            const v4s16 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)&sB[buf][rowsel * LP + wn + 16 * t + colsel]);
            const v4s16 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)&sB[buf][(rowsel + 4) * LP + wn + 16 * t + colsel]);
            typedef short v8s16 __attribute__((ext_vector_type(8)));
            const v8s16 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            const v8s16 bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            fa[t] = __builtin_bit_cast(bf16x8, av);
            fb[t] = __builtin_bit_cast(bf16x8, bv);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        if (c + 1 < c_end) stash(buf ^ 1);
        __syncthreads();
    }
    // accumulator (i, j)[r] of lane (li, lg) = C[m0 + wm + 16 i + 4 lg + r][n0 + wn + 16 j + li]
    float *C = pr.slabs + (size_t)it.slab * pr.M * pr.N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = it.m0 + wm + 16 * i + 4 * lg + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = it.n0 + wn + 16 * j + li;
                if (row < pr.M && col < pr.N) C[(size_t)row * pr.N + col] = acc[i][j][r];
            }
        }
}
struct RedProblem { float *slabs, *C; int n4, S; };
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const RedProblem *rp, const int *start, int count) {
    // grid-stride over all float4 of all problems; problem of a float4 by a short scan of the prefix table
    const int total = start[count];
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        int p = 0;
        for (int i = 1; i < count; ++i) if (e >= start[i]) p = i;
        const RedProblem r = rp[p];
        const int q = e - start[p];
        float4 v = reinterpret_cast<const float4 *>(r.slabs)[q];
        for (int s = 1; s < r.S; ++s) {
            const float4 w = reinterpret_cast<const float4 *>(r.slabs + (size_t)s * r.n4 * 4)[q];
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        reinterpret_cast<float4 *>(r.C)[q] = v;
    }
}

template <typename F> static double time_us(F fn, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 10; ++i) fn();
    std::vector<double> r;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < reps; ++i) fn();
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); r.push_back(ms * 1e3 / reps);
    }
    std::sort(r.begin(), r.end());
    return r[r.size() / 2];
}

int main(int argc, char **argv) {
    const int set = argc > 1 ? atoi(argv[1]) : 0;
    struct Sh { int M, N, K; };
    // set 0: the first weight-gradient launch at batch 1024 (K = 3072); set 1: the K = 1024 problems of the second one
    std::vector<Sh> shapes = set == 0 ? std::vector<Sh>{{256, 1024, 3072}, {256, 400, 3072}, {400, 256, 3072}, {256, 256, 3072}, {256, 256, 3072},
                                                       {256, 256, 3072}, {256, 256, 3072}, {256, 128, 3072}}
                                      : std::vector<Sh>{{2500, 256, 1024}, {2500, 256, 1024}, {256, 1024, 1024}, {676, 256, 1024}, {256, 256, 1024}, {256, 128, 1024}};
    const int P = (int)shapes.size();
    std::vector<AirGemmDesc> descs(P);
    std::vector<DwProblem> probs(P);
    std::vector<float *> Cref(P), Cnew(P);
    std::vector<std::vector<float>> hostA(P), hostB(P);
    double flops = 0;
    for (int p = 0; p < P; ++p) {
        const Sh s = shapes[p];
        std::vector<float> A((size_t)s.K * s.M), B((size_t)s.K * s.N);
        for (auto &x : A) x = frand();
        for (auto &x : B) x = frand();
        std::vector<unsigned short> A16(A.size()), B16(B.size());
        auto bf = [](float v) { unsigned u; memcpy(&u, &v, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); };
        for (size_t i = 0; i < A.size(); ++i) A16[i] = bf(A[i]);
        for (size_t i = 0; i < B.size(); ++i) B16[i] = bf(B[i]);
        float *dA, *dB; unsigned short *dA16, *dB16;
        CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dA16, A.size() * 2)); CK(hipMalloc(&dB16, B.size() * 2));
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dA16, A16.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB16, B16.data(), B.size() * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&Cref[p], (size_t)s.M * s.N * 4)); CK(hipMalloc(&Cnew[p], (size_t)s.M * s.N * 4));
        AirGemmDesc d{};
        d.ta = 1; d.tb = 0; d.M = s.M; d.N = s.N; d.K = s.K; d.A = dA; d.lda = s.M; d.B = dB; d.ldb = s.N; d.C = Cref[p]; d.ldc = s.N;
        d.epilogue = AIR_EPI_NONE; d.precision = AIR_PREC_BF16; d.A16 = dA16; d.B16 = dB16;
        descs[p] = d;
        probs[p] = {dA16, dB16, nullptr, s.M, s.N, s.K, s.M, s.N};
        flops += 2.0 * s.M * s.N * s.K;
    }
    // shipped: one big-group launch (needs more than AIR_GEMM_GROUP_MAX problems to take that path: pad with a repeat of the last one)
    std::vector<AirGemmDesc> big = descs;
    float *dummyC; CK(hipMalloc(&dummyC, (size_t)shapes[P - 1].M * shapes[P - 1].N * 4));
    while ((int)big.size() <= AIR_GEMM_GROUP_MAX) { AirGemmDesc d = descs[P - 1]; d.C = dummyC; big.push_back(d); flops += 0; }
    int st = air_gemm_grouped(big.data(), (int)big.size(), nullptr);
    if (st) { printf("shipped launch failed: %d\n", st); return 1; }
    const double t_ship = time_us([&] { air_gemm_grouped(big.data(), (int)big.size(), nullptr); }, 100);
    const double pad_flops = 2.0 * shapes[P - 1].M * shapes[P - 1].N * shapes[P - 1].K * ((int)big.size() - P);
    printf("set %d: %d problems, %.2f GFLOP (+%.2f GFLOP of padding problems in the shipped launch)\n", set, P, flops / 1e9, pad_flops / 1e9);
    printf("shipped 64x64 / 8-wave K split, one launch of %zu problems: %8.2f us  (%.0f TF)\n", big.size(), t_ship, (flops + pad_flops) / t_ship / 1e6);

    for (int slice : {256, 384, 512, 768, 1024, 1536, 3072}) {
        std::vector<DwItem> items;
        std::vector<RedProblem> red(P);
        std::vector<int> start(P + 1, 0);
        std::vector<float *> slabs(P);
        bool ok = true;
        for (int p = 0; p < P; ++p) {
            const Sh s = shapes[p];
            if (slice > s.K && slice != 3072) { }
            const int sl = std::min(slice, s.K);
            if (s.K % sl || sl % 32) { ok = false; break; }
            const int S = s.K / sl;
            CK(hipMalloc(&slabs[p], (size_t)S * s.M * s.N * 4));
            probs[p].slabs = slabs[p];
            for (int tm = 0; tm < (s.M + 127) / 128; ++tm)
                for (int tn = 0; tn < (s.N + 127) / 128; ++tn)
                    for (int k = 0; k < S; ++k) items.push_back({p, tm * 128, tn * 128, k * sl, (k + 1) * sl, k});
            red[p] = {slabs[p], Cnew[p], s.M * s.N / 4, S};
            start[p + 1] = start[p] + s.M * s.N / 4;
        }
        if (!ok) continue;
        if (getenv("DW_XCD") && atoi(getenv("DW_XCD"))) {
            // XCD-aware order: workgroup b runs on XCD b % 8 (observed placement).  All tiles of one (problem, K slice) read the same
            // operand rows; they go to ONE XCD, so that every operand block crosses the Infinity Cache -> L2 path once
            std::vector<std::vector<DwItem>> per(8);
            std::vector<size_t> load(8, 0);
            size_t i0 = 0;
            std::vector<DwItem> sorted = items;
            std::stable_sort(sorted.begin(), sorted.end(), [](const DwItem &x, const DwItem &y) { return x.p != y.p ? x.p < y.p : x.slab < y.slab; });
            while (i0 < sorted.size()) {
                size_t i1 = i0;
                while (i1 < sorted.size() && sorted[i1].p == sorted[i0].p && sorted[i1].slab == sorted[i0].slab) ++i1;
                int x = (int)(std::min_element(load.begin(), load.end()) - load.begin());
                for (size_t i = i0; i < i1; ++i) per[x].push_back(sorted[i]);
                load[x] += i1 - i0;
                i0 = i1;
            }
            size_t mx = 0; for (auto &v : per) mx = std::max(mx, v.size());
            items.clear();
            for (size_t j = 0; j < mx; ++j)
                for (int x = 0; x < 8; ++x) items.push_back(j < per[x].size() ? per[x][j] : DwItem{-1, 0, 0, 0, 0, 0});
        }
        // long slices first (all equal here), problems interleaved so that neighbours in the grid share operand slabs
        DwItem *d_items; DwProblem *d_prob; RedProblem *d_red; int *d_start;
        CK(hipMalloc(&d_items, items.size() * sizeof(DwItem))); CK(hipMemcpy(d_items, items.data(), items.size() * sizeof(DwItem), hipMemcpyHostToDevice));
        CK(hipMalloc(&d_prob, P * sizeof(DwProblem))); CK(hipMemcpy(d_prob, probs.data(), P * sizeof(DwProblem), hipMemcpyHostToDevice));
        CK(hipMalloc(&d_red, P * sizeof(RedProblem))); CK(hipMemcpy(d_red, red.data(), P * sizeof(RedProblem), hipMemcpyHostToDevice));
        CK(hipMalloc(&d_start, (P + 1) * 4)); CK(hipMemcpy(d_start, start.data(), (P + 1) * 4, hipMemcpyHostToDevice));
        DwArgs a = {d_prob, d_items};
        const int n_items = (int)items.size(), red_grid = std::min(1024, (start[P] + 255) / 256);
        auto run_mm = [&](int U) {
            if (U == 1) hipLaunchKernelGGL(dw128_kernel<1>, dim3(n_items), dim3(256), 0, 0, a);
            else if (U == 2) hipLaunchKernelGGL(dw128_kernel<2>, dim3(n_items), dim3(256), 0, 0, a);
            else hipLaunchKernelGGL(dw128_kernel<3>, dim3(n_items), dim3(256), 0, 0, a);
        };
        auto run_red = [&] { hipLaunchKernelGGL(reduce_slabs_kernel, dim3(red_grid), dim3(256), 0, 0, d_red, d_start, P); };
        for (int U : {2}) {
            run_mm(U); run_red();
            CK(hipDeviceSynchronize());
            // check against the shipped result (different summation order: tolerance)
            double worst = 0;
            for (int p = 0; p < P; ++p) {
                const size_t n = (size_t)shapes[p].M * shapes[p].N;
                std::vector<float> r(n), g(n);
                CK(hipMemcpy(r.data(), Cref[p], n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(g.data(), Cnew[p], n * 4, hipMemcpyDeviceToHost));
                double mx = 0, err = 0;
                for (size_t i = 0; i < n; ++i) { mx = std::max(mx, (double)fabsf(r[i])); err = std::max(err, (double)fabsf(r[i] - g[i])); }
                worst = std::max(worst, err / (mx + 1e-30));
            }
            const double t_mm = time_us([&] { run_mm(U); }, 100), t_red = time_us(run_red, 100), t_both = time_us([&] { run_mm(U); run_red(); }, 100);
            printf("128x128, slice %4d, U=%d: %5d items  product %7.2f us (%4.0f TF)  slab sum %6.2f us  both %7.2f us   max rel diff vs shipped %.1e\n",
                   slice, U, n_items, t_mm, flops / t_mm / 1e6, t_red, t_both, worst);
        }
        {   // the LDS-staged form on the same items
            bool all8 = true;
            for (int p = 0; p < P; ++p) all8 = all8 && shapes[p].M % 8 == 0 && shapes[p].N % 8 == 0;
            auto run_lds = [&] {
                if (all8) hipLaunchKernelGGL((dw_lds_kernel<true, true>), dim3(n_items), dim3(256), 0, 0, a);
                else hipLaunchKernelGGL((dw_lds_kernel<false, false>), dim3(n_items), dim3(256), 0, 0, a);
            };
            for (int p = 0; p < P; ++p) CK(hipMemset(Cnew[p], 0, (size_t)shapes[p].M * shapes[p].N * 4));
            run_lds(); run_red();
            CK(hipDeviceSynchronize());
            double worst = 0;
            for (int p = 0; p < P; ++p) {
                const size_t n = (size_t)shapes[p].M * shapes[p].N;
                std::vector<float> r(n), g(n);
                CK(hipMemcpy(r.data(), Cref[p], n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(g.data(), Cnew[p], n * 4, hipMemcpyDeviceToHost));
                double mx = 0, err = 0;
                for (size_t i = 0; i < n; ++i) { mx = std::max(mx, (double)fabsf(r[i])); err = std::max(err, (double)fabsf(r[i] - g[i])); }
                worst = std::max(worst, err / (mx + 1e-30));
            }
            const double t_mm = time_us(run_lds, 100), t_both = time_us([&] { run_lds(); run_red(); }, 100);
            printf("128x128 LDS + tr16, slice %4d: %5d items  product %7.2f us (%4.0f TF)  both %7.2f us   max rel diff vs shipped %.1e%s\n",
                   slice, n_items, t_mm, flops / t_mm / 1e6, t_both, worst, all8 ? "" : "  (8-byte loads: row lengths not multiples of 8)");
        }
        for (int p = 0; p < P; ++p) CK(hipFree(slabs[p]));
        CK(hipFree(d_items)); CK(hipFree(d_prob)); CK(hipFree(d_red)); CK(hipFree(d_start));
    }
    return 0;
}
