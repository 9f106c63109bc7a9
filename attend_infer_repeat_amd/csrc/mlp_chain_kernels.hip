// Row-slab dX chains of the MLPs on the bf16 data path (BASELINE configs[4]; round 6).
//
// Through an MLP (neural.py:93-102: Linear + ELU per layer) row r of a layer's input gradient needs only row r of its output
// gradient:  dA_{l-1} = (dA_l . W_l^T) * elu'(out_{l-1}).  The throughput plan ran that as one launch per layer -- 5-11 us each for
// < 1 us of MFMA work at 192 tiles, the latency chain of a dependent launch (profiles/r05_t_positions_c5_b1024.txt, positions
// 20-30) -- although nothing crosses rows.  Here a workgroup owns a SLAB of 16 rows and walks the whole chain: the slab of the
// incoming gradient sits in LDS as bf16 (the A operand of v_mfma_f32_16x16x32_bf16: lane (i, lg) reads 8 consecutive k with one
// 16-byte LDS load), W_l comes from the bf16 shadow of the parameters -- row n_out of W[n_out, n_in] is contiguous along the
// contraction, so the B operand is two 8-byte global loads per lane per 32-deep chunk, served by the L2 (128 KB per 256x256 layer,
// shared by the 24-32 workgroups of an XCD) -- and the epilogue multiplies by elu' of the saved activation, stores the fp32
// gradient (+ its bf16 mirror) that the deferred weight-gradient launch reads, and leaves bf16(dA_{l-1}) in the other LDS slab
// for the next layer.  Several chains (different MLPs, different row counts) share one launch.  fp32 accumulate; the values a
// layer consumes are bf16(fp32 result), exactly what the per-layer launches read from the mirrors.
// Layers narrower than 32 inputs (the baseline's 1-wide output layer) take a scalar path.
#include <stdlib.h>
#include "air_common.h"

typedef float f32x4c __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8c __attribute__((ext_vector_type(8)));
typedef unsigned u32x4c __attribute__((ext_vector_type(4)));
typedef unsigned u32x2c __attribute__((ext_vector_type(2)));
typedef const u32x2c __attribute__((address_space(1))) *gcu2c;

#define DXC_THREADS 512
#define DXC_MAX_WIDTH 1024

struct DxcLayer { const unsigned short *W16; const float *aux; float *out; unsigned short *out16; int n_in, n_out, ldaux, ldout; };
struct DxcChain { const float *g_in; int ld_in, rows, n_layers, slab0; DxcLayer layer[AIR_DXC_MAX_LAYERS]; };
struct DxcArgs { int n_chains, n_slabs, kp; DxcChain chain[AIR_DXC_MAX_CHAINS]; };

__device__ __forceinline__ unsigned short dxc_bf16(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
__device__ __forceinline__ float dxc_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

// One slab = 16 rows.  MEASURED AND NOT ADOPTED (profiles/r06_dx_chain_rejected.txt; AIR_DX_CHAIN=1 switches the plan pass on): the
// kernel is correct (tests/test_hip_kernels.py::test_mlp_dx_chain_matches_per_layer_products, the bf16 engine tests) but a layer inside
// it costs what a per-layer launch costs -- 24.0 us for the 3-layer decoder chain against 23.8 us for three grouped launches, stand-alone
// and warm -- and in the replayed configs[4] step the five chain launches that replace ten per-layer ones are 50-80 us SLOWER
// (0.47-0.50 against 0.415-0.425 ms).  Neither larger slabs (32 / 64 rows: a half / a quarter of the weight traffic, 28.9 / 41.4 us) nor
// taking both memory round trips off the chain (every layer's saved activations requested at kernel start, the next tile pair's /
// next layer's weight fragments requested under the current products: 27.9 us) helped: the per-layer cost at 3072 rows is not the
// dependent launch VERDICT r05 item 3 suspected, and not L2 bandwidth either.  Kept as the measured form of that experiment.
#define DXC_ROWS 16
__global__ __launch_bounds__(DXC_THREADS) void mlp_dx_chain16_kernel(DxcArgs a) {
    extern __shared__ __align__(16) unsigned short dxc_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = DXC_THREADS / 64, bid = blockIdx.x;
    int ci = 0;
#pragma unroll
    for (int q = 1; q < AIR_DXC_MAX_CHAINS; ++q) if (q < a.n_chains && bid >= a.chain[q].slab0) ci = q;
    const DxcChain &ch = a.chain[ci];
    const int r0 = (bid - ch.slab0) * DXC_ROWS;
    const int rows_valid = ch.rows - r0 < DXC_ROWS ? ch.rows - r0 : DXC_ROWS;
    const int KP = a.kp;                                      // slab pitch in bf16 elements (multiple of 8, +8 against bank conflicts)
    unsigned short *bufA = dxc_smem, *bufB = dxc_smem + DXC_ROWS * KP;
    // ---- the incoming gradient's slab -> bf16 in LDS (columns past n_in up to the next multiple of 32: zero); four 16-byte requests
    //      per thread in flight at a time
    {
        const int n_in = ch.layer[0].n_in, n_pad = (n_in + 31) & ~31, gq = n_pad >> 2;
        const bool v4 = (ch.ld_in & 3) == 0 && (n_in & 3) == 0 && air_aligned16_dev(ch.g_in);
        for (int e0 = 0; e0 < DXC_ROWS * gq; e0 += 4 * DXC_THREADS) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * DXC_THREADS + tid, row = e / gq, col = 4 * (e - row * gq);
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < DXC_ROWS * gq && row < rows_valid && col < n_in) {
                    const float *src = ch.g_in + (size_t)(r0 + row) * ch.ld_in + col;
                    if (v4) v[u] = *reinterpret_cast<const float4 *>(src);
                    else { v[u].x = src[0]; if (col + 1 < n_in) v[u].y = src[1]; if (col + 2 < n_in) v[u].z = src[2]; if (col + 3 < n_in) v[u].w = src[3]; }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * DXC_THREADS + tid, row = e / gq, col = 4 * (e - row * gq);
                if (e < DXC_ROWS * gq) {
                    unsigned short *d = bufA + row * KP + col;
                    d[0] = dxc_bf16(v[u].x); d[1] = dxc_bf16(v[u].y); d[2] = dxc_bf16(v[u].z); d[3] = dxc_bf16(v[u].w);
                }
            }
        }
    }
    __syncthreads();
    for (int l = 0; l < ch.n_layers; ++l) {
        const DxcLayer &ly = ch.layer[l];
        const int n_in = ly.n_in, n_out = ly.n_out, n_out_pad = (n_out + 31) & ~31;
        if (n_in >= 32 && (n_in & 3) == 0) {
            const int tiles = (n_out_pad + 15) >> 4;          // (the padded columns come out as exact zeros: next layer's k padding)
            const int i = lane & 15, lg = lane >> 4;
            // A wave forms TWO 16-column tiles side by side.  EIGHT chunks of both tiles' weight fragments are requested before the first
            // product (32 outstanding 8-byte loads per lane: a 256-deep contraction is one round trip to the L2), next to the saved
            // activations the epilogue differentiates through.
            for (int t0 = wid; t0 < tiles; t0 += 2 * nw) {
                f32x4c acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                const unsigned short *wrow[2];
                bool live[2];
                float yv[2][4];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int col = 16 * (t0 + q * nw) + i;
                    live[q] = (t0 + q * nw) < tiles && col < n_out;
                    wrow[q] = ly.W16 + (size_t)(live[q] ? col : 0) * n_in;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        yv[q][r] = 1.f;                        // (elu' = 1 where nothing is differentiated through)
                        if (ly.aux && live[q] && 4 * lg + r < rows_valid) yv[q][r] = ly.aux[(size_t)(r0 + 4 * lg + r) * ly.ldaux + col];
                    }
                }
                const int chunks = (n_in + 31) >> 5;
                for (int c0 = 0; c0 < chunks; c0 += 8) {
                    u32x4c fa[8], fb[2][8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int k = 32 * (c0 + u) + 8 * lg;
                        // (n_in % 4 == 0: each half of a lane's eight k is inside or outside as a whole; the shadow of a parameter tensor
                        //  is 8-byte aligned -- its fp32 master is 16-byte aligned -- so a fragment is two 8-byte loads)
                        const bool kin0 = k < n_in, kin1 = k + 4 < n_in;
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            u32x2c lo = {0u, 0u}, hi = {0u, 0u};
                            if (live[q] && kin0) lo = *(gcu2c)(wrow[q] + k);    // (explicit global address space: no flat load)
                            if (live[q] && kin1) hi = *(gcu2c)(wrow[q] + k + 4);
                            fb[q][u] = (u32x4c){lo.x, lo.y, hi.x, hi.y};
                        }
                        fa[u] = (u32x4c){0u, 0u, 0u, 0u};
                        if (c0 + u < chunks) fa[u] = *reinterpret_cast<const u32x4c *>(bufA + i * KP + k);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8c, fa[u]), __builtin_bit_cast(bf16x8c, fb[q][u]), acc[q], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int t = t0 + q * nw;
                    if (t >= tiles) continue;
                    const int col = 16 * t + i;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 4 * lg + r;
                        const bool in = row < rows_valid && col < n_out;
                        const float y = yv[q][r];               // elu' from the saved OUTPUT y: y > 0 ? 1 : y + 1
                        const float v = in ? acc[q][r] * (y > 0.f ? 1.f : y + 1.f) : 0.f;
                        const unsigned short hv = dxc_bf16(v);
                        if (in) {
                            ly.out[(size_t)(r0 + row) * ly.ldout + col] = v;
                            if (ly.out16) ly.out16[(size_t)(r0 + row) * ly.ldout + col] = hv;
                        }
                        bufB[row * KP + col] = hv;
                    }
                }
            }
        } else {
            // narrow contraction (the baseline's output layer has ONE column): plain dot products
            for (int e = tid; e < DXC_ROWS * n_out_pad; e += DXC_THREADS) {
                const int row = e / n_out_pad, col = e - row * n_out_pad;
                const bool in = row < rows_valid && col < n_out;
                float v = 0.f;
                if (in) {
                    const unsigned short *wr = ly.W16 + (size_t)col * n_in;
                    for (int k = 0; k < n_in; ++k) v = __builtin_fmaf(dxc_f32(bufA[row * KP + k]), dxc_f32(wr[k]), v);
                    if (ly.aux) { const float y = ly.aux[(size_t)(r0 + row) * ly.ldaux + col]; v *= y > 0.f ? 1.f : y + 1.f; }
                    ly.out[(size_t)(r0 + row) * ly.ldout + col] = v;
                    if (ly.out16) ly.out16[(size_t)(r0 + row) * ly.ldout + col] = dxc_bf16(v);
                }
                bufB[row * KP + col] = dxc_bf16(v);
            }
        }
        __syncthreads();
        unsigned short *tmp = bufA; bufA = bufB; bufB = tmp;
    }
}

// 1 when air_mlp_dx_chain_bf16 takes a layer of this shape (the caller plans the per-layer launch otherwise)
extern "C" int air_mlp_dx_chain_fits(int n_in, int n_out) {
    return (n_in >= 1 && n_out >= 1 && n_in <= DXC_MAX_WIDTH && n_out <= DXC_MAX_WIDTH && (n_in < 32 || (n_in & 3) == 0)) ? 1 : 0;
}

extern "C" int air_mlp_dx_chain_bf16(const AirDxChain *chains, int n_chains, void *stream) {
    AIR_REQUIRE(chains, AIR_E_NULL);
    AIR_REQUIRE(n_chains >= 1 && n_chains <= AIR_DXC_MAX_CHAINS, AIR_E_SHAPE);
    DxcArgs a;
    a.n_chains = n_chains;
    int widest = 32;
    const int slab_rows = DXC_ROWS;
    int slabs = 0;
    for (int c = 0; c < n_chains; ++c) {
        const AirDxChain &s = chains[c];
        AIR_REQUIRE(s.g_in, AIR_E_NULL);
        AIR_REQUIRE(s.rows > 0 && s.n_layers >= 1 && s.n_layers <= AIR_DXC_MAX_LAYERS && s.ld_in >= s.layer[0].n_in, AIR_E_SHAPE);
        DxcChain &d = a.chain[c];
        d.g_in = s.g_in; d.ld_in = s.ld_in; d.rows = s.rows; d.n_layers = s.n_layers; d.slab0 = slabs;
        for (int l = 0; l < s.n_layers; ++l) {
            const AirDxLayer &y = s.layer[l];
            AIR_REQUIRE(y.w_bf16 && y.out, AIR_E_NULL);
            AIR_REQUIRE(air_mlp_dx_chain_fits(y.n_in, y.n_out) == 1 && y.ldout >= y.n_out && (!y.aux || y.ldaux >= y.n_out), AIR_E_SHAPE);
            AIR_REQUIRE(l == 0 || y.n_in == s.layer[l - 1].n_out, AIR_E_SHAPE);
            AIR_REQUIRE(y.n_in < 32 || (reinterpret_cast<uintptr_t>(y.w_bf16) & 7u) == 0, AIR_E_ALIGN);    // (8-byte loads along a row of n_in % 4 == 0 bf16)
            d.layer[l] = DxcLayer{(const unsigned short *)y.w_bf16, y.aux, y.out, (unsigned short *)y.out_bf16, y.n_in, y.n_out, y.ldaux, y.ldout};
            const int w_in = (y.n_in + 31) & ~31, w_out = (y.n_out + 31) & ~31;
            widest = w_in > widest ? w_in : widest;
            widest = w_out > widest ? w_out : widest;
        }
        slabs += (s.rows + slab_rows - 1) / slab_rows;
    }
    a.n_slabs = slabs;
    a.kp = widest + 8;
    const size_t lds = (size_t)2 * slab_rows * a.kp * sizeof(unsigned short);
    AIR_REQUIRE(lds <= 160 * 1024, AIR_E_UNSUPPORTED);
    if (lds > 64 * 1024) {
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_dx_chain16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e_ != hipSuccess) return (int)e_;
    }
    hipLaunchKernelGGL(mlp_dx_chain16_kernel, dim3(slabs), dim3(DXC_THREADS), lds, air_stream(stream), a);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
