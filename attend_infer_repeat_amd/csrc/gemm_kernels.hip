// fp32 MFMA GEMM for the small, weight-resident layers of AIR (snt.Linear / snt.LSTM matmuls: neural.py:42-60,
// mnist_model.py:35; backward of the same).  C[M,N] = epi(op(A)[M,K].op(B)[K,N] + beta*C).
//
// Regime: M = B or T*B rows (64..3072), K,N in {1..3177}; all weights (10.5 MB) live in L2 / Infinity Cache, so at
// the headline batch the problem is latency/occupancy bound, not HBM bound.  Design for that:
//   * v_mfma_f32_16x16x4_f32 (exact fp32 fma chain; 157 TF peak = the fp32 vector rate, no TF32 on gfx950).
//   * operands go global -> VGPR directly in MFMA fragment order, no LDS staging: lane (i = l&15, g = l>>4)
//     supplies k = kc + 4g + j in MFMA step j for BOTH A and B, so a k-contiguous operand is one 16-byte load per
//     lane per 16-deep chunk and a k-strided operand is four dword loads covering 64-byte row segments.
//   * occupancy from split-K: the 4 waves of a workgroup interleave 16-deep K chunks of one output tile and reduce
//     through LDS (deterministic); a second, cross-workgroup split writes fp32 slabs that a small epilogue kernel
//     sums in a fixed order (no atomics => bitwise reproducible run to run).
//   * epilogues fuse bias, ELU, ELU' (backward), residual add; the dW form fuses the bias gradient (column sums).
// Arbitrary M/N/K and leading dimensions are handled with masked edge loads (test/cell_test.py uses 3x3 images and
// hidden sizes 5/7/11/13/17).
#include "air_common.h"
#include "prologue_device.h"
#include "optimizer_device.h"
#include "engine_device.h"
#include "nvil_device.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
// explicit global address space: descriptors that travel through memory (grouped launch) would otherwise make every
// operand access a FLAT load with a 64-bit VGPR address (+100 VGPRs, half the occupancy)
typedef const float __attribute__((address_space(1))) *gcf;
typedef float __attribute__((address_space(1))) *gf;
typedef gf gf_t;
typedef const f32x4 __attribute__((address_space(1))) *gcf4;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(1))) *gcu4;
typedef const u32x2 __attribute__((address_space(1))) *gcu2;
typedef const unsigned short __attribute__((address_space(1))) *gch;
typedef unsigned short __attribute__((address_space(1))) *gh_t;
struct Gemm16Ptrs { const void *A16, *B16; void *C16; };

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
    const bf16x2 v = __builtin_convertvector((f32x2){lo, hi}, bf16x2);
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned short bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }

// bf16 operand mode (BASELINE config 5, "bf16 MFMA MLP path"): storage stays fp32; the four k-values a lane holds for a
// 16-deep chunk are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) in registers and ONE v_mfma_f32_16x16x16_bf16 replaces the
// four v_mfma_f32_16x16x4_f32 -- same lane->k mapping (k = 4g..4g+3), fp32 accumulate.  1/8 of the MFMA issue cycles.
__device__ __forceinline__ s16x4 to_bf16x4(f32x4 v) {
    const bf16x2 lo = __builtin_convertvector((f32x2){v.x, v.y}, bf16x2);
    const bf16x2 hi = __builtin_convertvector((f32x2){v.z, v.w}, bf16x2);
    const u32x2 p = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
    return __builtin_bit_cast(s16x4, p);
}
template <int MT, int NT, bool BF>
__device__ __forceinline__ void mfma_chunk(f32x4 (&acc)[MT][NT], const f32x4 (&fa)[MT], const f32x4 (&fb)[NT]) {
    if (BF) {
        s16x4 ha[MT], hb[NT];
#pragma unroll
        for (int a = 0; a < MT; ++a) ha[a] = to_bf16x4(fa[a]);
#pragma unroll
        for (int b = 0; b < NT; ++b) hb[b] = to_bf16x4(fb[b]);
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ha[a], hb[b], acc[a][b], 0, 0, 0);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a][j], fb[b][j], acc[a][b], 0, 0, 0);
    }
}

// consumer-side reduction of a K-split producer (AirGemmDesc.A2 ...): a separate kernel argument of the one kernel that uses it --
// every field added to GemmArgs is loaded by EVERY GEMM launch (x8 in a grouped launch): five more cost 7 us per step, measured
struct AproArgs {
    const float *A2, *a_bias;
    float *a_out;
    int a_elu;
};
// the centred-RMSProp update folded into the epilogue of the weight-gradient tiles that FORM the gradients (single GPU: a tile's
// gradient is final when formed), plus rider workgroups for slices whose gradients earlier launches left final, plus the step
// counters: the closing launch of the train step disappears (air_gemm_grouped_opt).  A separate kernel argument of the kernels
// that use it, like AproArgs.
#define AIR_OPT_MAX_RANGES 4
struct OptFold {
    float *p; const float *g0; float *ms, *mg, *mom;       // bases of the flat buffers; g0: the flat gradient buffer C / colsum point into
    size_t n_model;
    const float *lr_dev;
    float lr_mult_tail, decay, momentum, eps, gscale;
    unsigned fold_mask;                                     // bit i: problem i's C and colsum elements are updated where they are formed
    int n_ranges, tiles;                                    // rider slices [lo, hi) (multiples of 4); workgroups >= tiles are riders
    size_t lo[AIR_OPT_MAX_RANGES], hi[AIR_OPT_MAX_RANGES];
    int64_t *gstep; uint64_t *rng_state; uint64_t rng_inc;
};
// the backward of a Gaussian head (loc_mode 0: the `what` head) folded into the epilogue of the product that forms its sample
// gradient (the decoder's first-layer dX, d_what[T*B, D]): the tile's element (m, d) IS dsample[m, d], and the thread that finished it
// writes dpre[m, d] and dpre[m, D + d] (engine_device.h gauss_bwd_elem) instead of storing it for a pointwise launch to re-read.
struct GaussEpi {
    const float *pre, *eps, *loc, *scale, *dkl_row;
    float *dpre;
    int ld_pre, ld_dpre, D;
    float raw_offset, pl, ps, dkl_scale, guard;
    unsigned mask;                                          // bit i: problem i carries the epilogue
};
// The HBM feeder's gather folded into the step's FIRST product (round 6): the problems of that launch contract over the pixels of the
// observation batch, so row m of their A operand is read straight from item idx_m of the resident dataset -- idx_m drawn exactly as
// air_batch_gather draws it (Philox(seed, stream 1, counter step*B + m), or the sequential walk) -- and the first column of tiles of the
// problems in copy_mask writes the fragments it loaded into the observation buffer, which every later launch of the step reads as before.
// One dependent launch less at the head of the step.  A separate kernel argument of the one kernel that uses it, like AproArgs.
struct GatherArgs {
    const float *data, *obs;
    float *obs_out;
    long long n_items;
    int item_floats, shuffle, B;
    const uint64_t *seed;
    const int64_t *step;
    int64_t *idx_out;
    unsigned copy_mask;
};
__device__ __forceinline__ long long gather_item(const GatherArgs &gt, int m) {
    const unsigned long long ctr = (unsigned long long)gt.step[0] * (unsigned long long)gt.B + (unsigned long long)m;
    if (gt.shuffle) {
        uint32_t r[4];
        philox4x32(ctr, 1, gt.seed[0], r);
        const unsigned long long wide = ((unsigned long long)r[0] << 32) | r[1];
        return (long long)(((unsigned __int128)wide * (unsigned __int128)gt.n_items) >> 64);        // uniform in [0, n)
    }
    return (long long)(ctr % (unsigned long long)gt.n_items);
}
struct GemmArgs {
    const float *A, *B, *bias, *aux;
    float *C, *colsum, *ws;
    int M, N, K, lda, ldb, ldc, ldaux;
    int ta, tb, epi, S, chunks_per_split, vecA, vecB;
    float beta;
};

// element k..k+3 of a k-contiguous operand row (row-major [rows, K]); zero outside
__device__ __forceinline__ f32x4 ld_kcontig(gcf p, int ld, int row, bool row_ok, int k, int K, bool vec) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row_ok && k < K) {
        gcf q = p + (size_t)row * ld + k;
        if (vec && k + 3 < K) {
            v = *(gcf4)q;
        } else {
            v.x = q[0];
            if (k + 1 < K) v.y = q[1];
            if (k + 2 < K) v.z = q[2];
            if (k + 3 < K) v.w = q[3];
        }
    }
    return v;
}
// rows k..k+3, fixed column, of a k-strided operand (row-major [K, cols]); zero outside
__device__ __forceinline__ f32x4 ld_kstrided(gcf p, int ld, int col, bool col_ok, int k, int K) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (col_ok && k < K) {
        gcf q = p + (size_t)k * ld + col;
        v.x = q[0];
        if (k + 1 < K) v.y = q[ld];
        if (k + 2 < K) v.z = q[2 * (size_t)ld];
        if (k + 3 < K) v.w = q[3 * (size_t)ld];
    }
    return v;
}

// Unmasked variants for chunks that lie fully inside K.  Rows / columns beyond the matrix are CLAMPED to a valid
// address instead of masked: they only feed accumulator rows / columns that are never stored.
__device__ __forceinline__ f32x4 ld_kcontig_full(gcf p, int ld, int row, int k, bool vec) {
    gcf q = p + (size_t)row * ld + k;
    if (vec) return *(gcf4)q;
    f32x4 v;
    v.x = q[0]; v.y = q[1]; v.z = q[2]; v.w = q[3];
    return v;
}
__device__ __forceinline__ f32x4 ld_kstrided_full(gcf p, int ld, int col, int k) {
    gcf q = p + (size_t)k * ld + col;
    f32x4 v;
    v.x = q[0]; v.y = q[ld]; v.z = q[2 * (size_t)ld]; v.w = q[3 * (size_t)ld];
    return v;
}

__device__ __forceinline__ float apply_epilogue(float v, int m, int n, const GemmArgs &g) {
    if (g.beta != 0.f) v += g.beta * g.C[(size_t)m * g.ldc + n];
    switch (g.epi) {
        case AIR_EPI_BIAS: v += g.bias[n]; break;
        case AIR_EPI_BIAS_ELU: v = elu_acc(v + g.bias[n]); break;
        case AIR_EPI_MUL_DELU: {
            const float y = g.aux[(size_t)m * g.ldaux + n];
            v *= (y > 0.f ? 1.f : y + 1.f);
        } break;
        case AIR_EPI_ADD_AUX:
            v += g.aux[(size_t)m * g.ldaux + n];
            if (g.bias) v += g.bias[n];
            break;
        case AIR_EPI_ADD_AUX_ELU:
            v += g.aux[(size_t)m * g.ldaux + n];
            if (g.bias) v += g.bias[n];
            v = elu_acc(v);
            break;
        default: break;
    }
    return v;
}

// MT x NT 16x16 MFMA tiles per wave.  KW = 4 / 16: the workgroup's KW waves split K for ONE tile (LDS reduce; 16
// waves = 1024 threads for long-K problems with few tiles, so no second split-K launch is needed);
// KW = 1: the 4 waves own 4 neighbouring N-tiles.
template <int MT, int NT, int KW, bool BF, bool APRO = false, bool OPT = false, bool GBW = false, bool GATH = false>
__device__ __forceinline__ void gemm_body(const GemmArgs &g, const int block_tile, const int split, const AproArgs pro = AproArgs(),
                                          void *c16 = nullptr, const OptFold *opt = nullptr, const bool fold = false,
                                          const GaussEpi *gb = nullptr, const GatherArgs *gt = nullptr, const bool gcopy = false) {
    const gh_t hC = (gh_t)c16;             // bf16 mirror of C (bf16 data path: the next product reads it instead of the fp32 value)
    constexpr int TM = 16 * MT, TN = 16 * NT;
    constexpr int NWN = (KW == 1) ? 4 : 1;               // waves across N
    constexpr int NWV = (KW == 1) ? 4 : KW;               // waves per workgroup
    constexpr int NTH = 64 * NWV;                         // threads per workgroup
    constexpr int LDT = TN + 4;                           // padded LDS tile row
    __shared__ float s_tile[NWV][TM * LDT];
    __shared__ float s_col[NWV][TN];

    const gcf gA = (gcf)g.A, gB = (gcf)g.B, gBias = (gcf)g.bias, gAux = (gcf)g.aux;
    const gf gC = (gf)g.C, gWs = (gf)g.ws, gCol = (gf)g.colsum;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int tiles_n = (g.N + TN * NWN - 1) / (TN * NWN);
    const int tm = block_tile / tiles_n, tn = block_tile - tm * tiles_n;
    const int m0 = tm * TM;
    const int n0 = (tn * NWN + (KW == 1 ? wave : 0)) * TN;

    const int total_chunks = (g.K + 15) >> 4;
    const int c_begin = split * g.chunks_per_split;
    int c_end = c_begin + g.chunks_per_split;
    if (c_end > total_chunks) c_end = total_chunks;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float csum[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) csum[b] = 0.f;
    const bool want_colsum = (g.colsum != nullptr) && (tm == 0);

    int rowA[MT], colB[NT];
    bool okA[MT], okB[NT];
#pragma unroll
    for (int a = 0; a < MT; ++a) { rowA[a] = m0 + 16 * a + li; okA[a] = rowA[a] < g.M; }
#pragma unroll
    for (int b = 0; b < NT; ++b) { colB[b] = n0 + 16 * b + li; okB[b] = colB[b] < g.N; }

    // Epilogue operands (bias / aux / beta*C) are fetched NOW, before the K loop, so their memory round trip overlaps
    // the operand loads instead of following the LDS reduction (the kernel is a chain of round trips, not of flops).
    constexpr int EPT = (KW > 1) ? (TM * TN + NTH - 1) / NTH : 1;
    float e_bias[EPT], e_aux[EPT], e_c[EPT];
    if (KW > 1 && g.S == 1) {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = threadIdx.x + NTH * i;
            const int r = e / TN, cidx = e - r * TN;
            const int m = m0 + r, n = n0 + cidx;
            const bool ok = (e < TM * TN) && m < g.M && n < g.N;
            e_bias[i] = (ok && g.bias != nullptr) ? gBias[n] : 0.f;
            e_aux[i] = (ok && g.epi >= AIR_EPI_MUL_DELU) ? gAux[(size_t)m * g.ldaux + n] : 0.f;
            e_c[i] = (ok && g.beta != 0.f) ? gC[(size_t)m * g.ldc + n] : 0.f;
        }
    }
    // Gaussian-head backward in the epilogue: its operands are requested NOW as well
    float b_loc[GBW ? EPT : 1], b_scale[GBW ? EPT : 1], b_eps[GBW ? EPT : 1], b_raw[GBW ? EPT : 1], b_dk[GBW ? EPT : 1];
    if (GBW && KW > 1 && fold) {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = threadIdx.x + NTH * i;
            const int r = e / TN, cidx = e - r * TN;
            const int m = m0 + r, n = n0 + cidx;
            const bool ok = (e < TM * TN) && m < g.M && n < g.N;
            const size_t o = ok ? (size_t)m * gb->D + n : 0;
            b_loc[i] = ok ? ((gcf)gb->loc)[o] : 0.f;
            b_scale[i] = ok ? ((gcf)gb->scale)[o] : 1.f;
            b_eps[i] = ok ? ((gcf)gb->eps)[o] : 0.f;
            b_raw[i] = ok ? ((gcf)gb->pre)[(size_t)m * gb->ld_pre + gb->D + n] : 0.f;
            b_dk[i] = (ok && gb->dkl_row) ? ((gcf)gb->dkl_row)[m] : 0.f;
        }
    }
    // folded update: the element's parameter and RMSProp slots are requested NOW, with the operands (same flat offset as its gradient)
    float o_p[OPT ? EPT : 1], o_ms[OPT ? EPT : 1], o_mg[OPT ? EPT : 1], o_mom[OPT ? EPT : 1];
    float o_lr0 = 0.f;
    if (OPT && KW > 1 && fold) {
        o_lr0 = ((gcf)opt->lr_dev)[0];
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = threadIdx.x + NTH * i;
            const int r = e / TN, cidx = e - r * TN;
            const int m = m0 + r, n = n0 + cidx;
            const bool ok = (e < TM * TN) && m < g.M && n < g.N;
            const size_t idx = ok ? (size_t)((gC + (size_t)m * g.ldc + n) - (gf)opt->g0) : 0;
            o_p[i] = ok ? ((gf)opt->p)[idx] : 0.f;
            o_ms[i] = ok ? ((gf)opt->ms)[idx] : 0.f;
            o_mg[i] = ok ? ((gf)opt->mg)[idx] : 0.f;
            o_mom[i] = ok ? ((gf)opt->mom)[idx] : 0.f;
        }
    }

    const int c_step = (KW > 1) ? KW : 1;
    int c = c_begin + ((KW > 1) ? wave : 0);
    // ---- main loop: U chunks in flight per wave.  The problem is latency bound (operands sit in L2 / Infinity
    // Cache, each wave owns only a handful of 16-deep chunks), so all loads of U chunks are issued before the first
    // MFMA: one memory round trip per U chunks instead of one per chunk.
    constexpr int U = (MT * NT == 1) ? (KW == 16 ? 10 : 8) : 4;   // chunks in flight: small tiles can afford more registers
    int full_end = g.K >> 4;                               // chunks [0, full_end) need no k masking
    if (full_end > c_end) full_end = c_end;
    int rowAc[MT], colBc[NT];
#pragma unroll
    for (int a = 0; a < MT; ++a) rowAc[a] = okA[a] ? rowA[a] : g.M - 1;
#pragma unroll
    for (int b = 0; b < NT; ++b) colBc[b] = okB[b] ? colB[b] : g.N - 1;
    // GATH: row m of A = item idx_m of the dataset, at the problem's column offset inside an observation row (A points into row 0 of obs)
    const int gcol = GATH ? (int)(gA - (gcf)gt->obs) : 0;
    const gcf gAs = GATH ? (gcf)gt->data + gcol : gA;
    const int ldAs = GATH ? gt->item_floats : g.lda;
    int rowAs[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a) rowAs[a] = GATH ? (int)gather_item(*gt, rowAc[a]) : rowAc[a];
    if (GATH && gcopy && tn == 0 && split == 0 && wave == 0 && lg == 0 && gt->idx_out && gcol == 0) {
#pragma unroll
        for (int a = 0; a < MT; ++a) if (okA[a]) gt->idx_out[rowA[a]] = rowAs[a];
    }
#pragma nounroll
    for (; c < full_end; c += U * c_step) {   // (unrolling this loop doubles the live operand registers: 94 -> 194 VGPRs)
        f32x4 fa[U][MT], fb[U][NT];
        f32x4 fa2[APRO ? U : 1][APRO ? MT : 1], fab[APRO ? U : 1];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cu = c + u * c_step;
            if (cu < full_end) {                           // wave-uniform: a short tail group issues fewer loads
                const int k = (cu << 4) + 4 * lg;
#pragma unroll
                for (int a = 0; a < MT; ++a) {
                    fa[u][a] = g.ta ? ld_kstrided_full(gA, g.lda, rowAc[a], k)
                                    : ld_kcontig_full(gAs, ldAs, rowAs[a], k, g.vecA != 0);
                    if (APRO) {       // second K-split slab + bias of the producing layer: requested with the operand itself
                        fa2[u][a] = ld_kcontig_full((gcf)pro.A2, g.lda, rowAc[a], k, g.vecA != 0);
                        fab[u] = ld_kcontig_full((gcf)pro.a_bias, 0, 0, k, g.vecA != 0);
                    }
                }
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    fb[u][b] = g.tb ? ld_kcontig_full(gB, g.ldb, colBc[b], k, g.vecB != 0)
                                    : ld_kstrided_full(gB, g.ldb, colBc[b], k);
            } else {
#pragma unroll
                for (int a = 0; a < MT; ++a) fa[u][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int b = 0; b < NT; ++b) fb[u][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c + u * c_step >= full_end) break;
            if (APRO) {
#pragma unroll
                for (int a = 0; a < MT; ++a) {
                    f32x4 v = (fa[u][a] + fa2[u][a]) + fab[u];
                    if (pro.a_elu) { v.x = elu_acc(v.x); v.y = elu_acc(v.y); v.z = elu_acc(v.z); v.w = elu_acc(v.w); }
                    fa[u][a] = v;
                    if (pro.a_out != nullptr && tn == 0 && okA[a])        // the reduced activation, for the backward pass
                        *(f32x4 *)(pro.a_out + (size_t)rowA[a] * g.lda + (((c + u * c_step) << 4) + 4 * lg)) = v;
                }
            }
            if (GATH && gcopy && tn == 0) {                // the gathered rows, for every later reader of the observation batch
#pragma unroll
                for (int a = 0; a < MT; ++a)
                    if (okA[a]) *(f32x4 *)(gt->obs_out + (size_t)rowA[a] * g.lda + gcol + (((c + u * c_step) << 4) + 4 * lg)) = fa[u][a];
            }
            mfma_chunk<MT, NT, BF>(acc, fa[u], fb[u]);
            if (want_colsum) {
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    csum[b] += okB[b] ? (fb[u][b].x + fb[u][b].y) + (fb[u][b].z + fb[u][b].w) : 0.f;
            }
        }
    }
    // ---- the partial last chunk of K (if any), fully masked, done by the wave that owns it
    const int pc = g.K >> 4;
    if ((g.K & 15) && pc >= c_begin && pc < c_end && (KW == 1 || ((pc - c_begin) % KW) == wave)) {
        const int k = (pc << 4) + 4 * lg;
        f32x4 fa[MT], fb[NT];
#pragma unroll
        for (int a = 0; a < MT; ++a)
            fa[a] = g.ta ? ld_kstrided(gA, g.lda, rowA[a], okA[a], k, g.K)
                         : ld_kcontig(gAs, ldAs, GATH ? rowAs[a] : rowA[a], okA[a], k, g.K, g.vecA != 0);
        if (GATH && gcopy && tn == 0) {
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                if (!okA[a]) continue;
                float *d = gt->obs_out + (size_t)rowA[a] * g.lda + gcol + k;
                if (k < g.K) d[0] = fa[a].x;
                if (k + 1 < g.K) d[1] = fa[a].y;
                if (k + 2 < g.K) d[2] = fa[a].z;
                if (k + 3 < g.K) d[3] = fa[a].w;
            }
        }
#pragma unroll
        for (int b = 0; b < NT; ++b)
            fb[b] = g.tb ? ld_kcontig(gB, g.ldb, colB[b], okB[b], k, g.K, g.vecB != 0)
                         : ld_kstrided(gB, g.ldb, colB[b], okB[b], k, g.K);
        mfma_chunk<MT, NT, BF>(acc, fa, fb);
        if (want_colsum) {
#pragma unroll
            for (int b = 0; b < NT; ++b) csum[b] += (fb[b].x + fb[b].y) + (fb[b].z + fb[b].w);
        }
    }

    // accumulators -> LDS tile (C/D map: col = lane&15, row = (lane>>4)*4 + r)
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_tile[wave][(16 * a + 4 * lg + r) * LDT + 16 * b + li] = acc[a][b][r];
    if (want_colsum) {
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            float v = csum[b];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lg == 0) s_col[wave][16 * b + li] = v;
        }
    }
    __syncthreads();

    if (KW > 1) {
        // all threads reduce the KW per-wave partial tiles (fixed order) and finish one TM x TN tile
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = threadIdx.x + NTH * i;
            if (e >= TM * TN) continue;
            const int r = e / TN, cidx = e - r * TN;
            const int m = m0 + r, n = n0 + cidx;
            if (m >= g.M || n >= g.N) continue;
            const int off = r * LDT + cidx;
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < NWV; q += 4)
                v += (s_tile[q][off] + s_tile[q + 1][off]) + (s_tile[q + 2][off] + s_tile[q + 3][off]);
            if (g.S > 1) {
                gWs[((size_t)split * g.M + m) * g.N + n] = v;
            } else {
                if (g.beta != 0.f) v += g.beta * e_c[i];
                switch (g.epi) {
                    case AIR_EPI_BIAS: v += e_bias[i]; break;
                    case AIR_EPI_BIAS_ELU: v = elu_acc(v + e_bias[i]); break;
                    case AIR_EPI_MUL_DELU: v *= (e_aux[i] > 0.f ? 1.f : e_aux[i] + 1.f); break;
                    case AIR_EPI_ADD_AUX: v += e_aux[i] + e_bias[i]; break;
                    case AIR_EPI_ADD_AUX_ELU: v = elu_acc(v + e_aux[i] + e_bias[i]); break;
                    default: break;
                }
                gC[(size_t)m * g.ldc + n] = v;
                if (BF && hC) hC[(size_t)m * g.ldc + n] = bf16_bits(v);
                if (GBW && fold) {                          // v = dsample[m, n] of the head: its pre-activation gradients, here
                    float dloc, draw;
                    gauss_bwd_elem(v, true, b_eps[i], b_loc[i], b_scale[i], gb->pl, gb->ps, b_dk[i] * gb->dkl_scale,
                                   b_raw[i] + gb->raw_offset, 0, n, gb->guard, dloc, draw);
                    const gf dp_ = (gf)gb->dpre + (size_t)m * gb->ld_dpre;
                    dp_[n] = dloc; dp_[gb->D + n] = draw;
                }
                if (OPT && fold) {
                    const size_t idx = (size_t)((gC + (size_t)m * g.ldc + n) - (gf)opt->g0);
                    const float lr = idx < opt->n_model ? o_lr0 : o_lr0 * opt->lr_mult_tail;
                    float pv = o_p[i], a = o_ms[i], b = o_mg[i], c = o_mom[i];
                    rmsprop_elem(pv, v, a, b, c, lr, opt->decay, opt->momentum, opt->eps, opt->gscale);
                    ((gf)opt->ms)[idx] = a; ((gf)opt->mg)[idx] = b; ((gf)opt->mom)[idx] = c; ((gf)opt->p)[idx] = pv;
                }
            }
        }
        if (want_colsum && threadIdx.x < TN) {
            const int n = n0 + threadIdx.x;
            if (n < g.N) {
                float v = 0.f;
#pragma unroll
                for (int q = 0; q < NWV; q += 4)
                    v += (s_col[q][threadIdx.x] + s_col[q + 1][threadIdx.x]) + (s_col[q + 2][threadIdx.x] + s_col[q + 3][threadIdx.x]);
                gCol[n] = v;
                if (OPT && fold) {                          // the bias gradient this tile column finishes: same treatment
                    const size_t idx = (size_t)((gCol + n) - (gf)opt->g0);
                    const float lr = idx < opt->n_model ? o_lr0 : o_lr0 * opt->lr_mult_tail;
                    float pv = ((gf)opt->p)[idx], a = ((gf)opt->ms)[idx], b = ((gf)opt->mg)[idx], c = ((gf)opt->mom)[idx];
                    rmsprop_elem(pv, v, a, b, c, lr, opt->decay, opt->momentum, opt->eps, opt->gscale);
                    ((gf)opt->ms)[idx] = a; ((gf)opt->mg)[idx] = b; ((gf)opt->mom)[idx] = c; ((gf)opt->p)[idx] = pv;
                }
            }
        }
    } else {
        // each wave finishes its own tile
        for (int e = lane; e < TM * TN; e += 64) {
            const int r = e / TN, cidx = e - r * TN;
            const int m = m0 + r, n = n0 + cidx;
            if (m >= g.M || n >= g.N) continue;
            const float v = s_tile[wave][r * LDT + cidx];
            if (g.S > 1) gWs[((size_t)split * g.M + m) * g.N + n] = v;
            else {
                const float o = apply_epilogue(v, m, n, g);
                gC[(size_t)m * g.ldc + n] = o;
                if (BF && hC) hC[(size_t)m * g.ldc + n] = bf16_bits(o);
            }
        }
        if (want_colsum && lane < TN) {
            const int n = n0 + lane;
            if (n < g.N) gCol[n] = s_col[wave][lane];
        }
    }
}

template <int MT, int NT, int KW, bool BF>
__global__ __launch_bounds__(KW == 1 ? 256 : 64 * KW) void gemm_f32_mfma_kernel(GemmArgs g) {
    gemm_body<MT, NT, KW, BF>(g, blockIdx.x, blockIdx.y);
}
// bf16 data path, small tiles: fp32 operand fetch (rounded in registers) + the bf16 mirror of C
template <int MT, int NT, int KW>
__global__ __launch_bounds__(KW == 1 ? 256 : 64 * KW) void gemm_bf16_c16_kernel(GemmArgs g, void *c16) {
    gemm_body<MT, NT, KW, true>(g, blockIdx.x, blockIdx.y, AproArgs(), c16);
}
// the same body with the A-operand prologue (consumer-side reduction of a K-split producer) compiled in
template <int MT, int NT, int KW, bool BF>
__global__ __launch_bounds__(KW == 1 ? 256 : 64 * KW) void gemm_f32_mfma_apro_kernel(GemmArgs g, AproArgs pro) {
    gemm_body<MT, NT, KW, BF, true>(g, blockIdx.x, blockIdx.y, pro);
}

// ---- throughput regime (thousands of rows): "wide" wave tiles with every operand load 16 bytes per lane ---------------------
// gemm_body feeds a k-strided operand (forward: W[K,N]; weight gradient: both operands) with dword loads -- 64-byte segments, 16
// address cycles per wave instruction for 256 bytes: at batch 1024 the load path, not the MFMA pipe, bounds those products (the
// texture-address unit is busier than the matrix cores: 640 against 512 cycles per 16-deep chunk of a 32x32 tile, 4 waves per CU).
// Here a k-strided operand is read with ONE 16-byte load per lane ALONG ITS CONTIGUOUS DIMENSION, and the four values a lane gets
// are handed to FOUR DIFFERENT 16-wide MFMA tiles: tile t of a 64-wide block owns the INTERLEAVED columns n0 + 4 i + t (i = 0..15)
// instead of the contiguous columns n0 + 16 t + i.  MFMA does not care which column a lane's value belongs to, only that A and B
// agree on k; the accumulator of lane (i, g) then holds C[row][n0 + 4 i + 0..3] across the four tiles: a 16-byte store.  A
// k-contiguous operand keeps the block layout (one 16-byte load per lane per chunk along k).  Wave tile (16 MT) x 64: MT + 4
// 16-byte loads per 16-deep chunk feed 4 MT (bf16) or 16 MT (fp32) MFMAs; KW waves split K and reduce through LDS in fixed order.
template <int MT, int KW, bool BF, bool TA, bool TB>
__device__ __forceinline__ void gemm_wide_body(const GemmArgs &g, const int block_tile) {
    constexpr int NT = 4, TM = 16 * MT, TN = 64;
    constexpr int NTH = 64 * KW;
    constexpr int LDT = TN + 4;
    static_assert(!TA || MT == 4, "an m-contiguous A operand is read 4 rows per lane: four interleaved row tiles");
    __shared__ float s_tile[KW][TM * LDT];
    __shared__ float s_col[KW][TN];

    const gcf gA = (gcf)g.A, gB = (gcf)g.B, gBias = (gcf)g.bias, gAux = (gcf)g.aux;
    const gf gC = (gf)g.C, gCol = (gf)g.colsum;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int tiles_n = (g.N + TN - 1) / TN;
    const int tm = block_tile / tiles_n, tn = block_tile - tm * tiles_n;
    const int m0 = tm * TM, n0 = tn * TN;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float csum[NT] = {0.f, 0.f, 0.f, 0.f};
    const bool want_colsum = TA && (g.colsum != nullptr) && (tm == 0);

    // operand addressing.  Block layout: tile t row/col 16 t + li, one address per tile.  Interleaved: rows/cols 4 li + t, ONE
    // address for all four tiles (clamped inside the matrix: lanes past the edge feed accumulators that are never stored; the
    // dispatcher guarantees M % 4 == 0 / N % 4 == 0 for an interleaved operand, so a lane is entirely inside or entirely outside)
    int offA[TA ? 1 : MT], offB[TB ? NT : 1];
    if (TA) { int r = m0 + 4 * li; if (r > g.M - 4) r = g.M - 4; offA[0] = r; }
    else {
#pragma unroll
        for (int a = 0; a < MT; ++a) { int r = m0 + 16 * a + li; if (r > g.M - 1) r = g.M - 1; offA[a] = r * g.lda; }
    }
    if (!TB) { int c = n0 + 4 * li; if (c > g.N - 4) c = g.N - 4; offB[0] = c; }
    else {
#pragma unroll
        for (int b = 0; b < NT; ++b) { int c = n0 + 16 * b + li; if (c > g.N - 1) c = g.N - 1; offB[b] = c * g.ldb; }
    }

    // epilogue operands requested before the K loop (their round trip hides under the operand loads)
    constexpr int EPT = (TM * TN) / NTH;
    static_assert((TM * TN) % NTH == 0, "tile / threads");
    float e_bias[EPT], e_aux[EPT], e_c[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = threadIdx.x + NTH * i;
        const int r = e / TN, cidx = e - r * TN;
        int m = m0 + r, n = n0 + cidx;
        if (m > g.M - 1) m = g.M - 1;
        if (n > g.N - 1) n = g.N - 1;
        e_bias[i] = g.bias != nullptr ? gBias[n] : 0.f;
        e_aux[i] = g.epi >= AIR_EPI_MUL_DELU ? gAux[(size_t)m * g.ldaux + n] : 0.f;
        e_c[i] = g.beta != 0.f ? gC[(size_t)m * g.ldc + n] : 0.f;
    }

    constexpr int U = (MT == 4) ? 3 : 4;                   // chunks in flight per wave
    const int full_end = g.K >> 4;
    auto load_chunk = [&](int k, f32x4 (&fa)[MT], f32x4 (&fb)[NT]) {       // k = first of the lane group's four k values
        if (TA) {
            f32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *(gcf4)(gA + (size_t)(k + j) * g.lda + offA[0]);
#pragma unroll
            for (int a = 0; a < MT; ++a) fa[a] = (f32x4){v[0][a], v[1][a], v[2][a], v[3][a]};
        } else {
#pragma unroll
            for (int a = 0; a < MT; ++a) fa[a] = *(gcf4)(gA + offA[a] + k);
        }
        if (!TB) {
            f32x4 w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = *(gcf4)(gB + (size_t)(k + j) * g.ldb + offB[0]);
#pragma unroll
            for (int b = 0; b < NT; ++b) fb[b] = (f32x4){w[0][b], w[1][b], w[2][b], w[3][b]};
        } else {
#pragma unroll
            for (int b = 0; b < NT; ++b) fb[b] = *(gcf4)(gB + offB[b] + k);
        }
    };
    int c = wave;
#pragma nounroll
    for (; c < full_end; c += U * KW) {
        f32x4 fa[U][MT], fb[U][NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int cu = c + u * KW;
            if (cu > full_end - 1) cu = full_end - 1;      // a short tail group re-reads the last chunk (never multiplied)
            load_chunk((cu << 4) + 4 * lg, fa[u], fb[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c + u * KW >= full_end) break;
            mfma_chunk<MT, NT, BF>(acc, fa[u], fb[u]);
            if (want_colsum) {
#pragma unroll
                for (int b = 0; b < NT; ++b) csum[b] += (fb[u][b].x + fb[u][b].y) + (fb[u][b].z + fb[u][b].w);
            }
        }
    }
    // the partial last chunk of K (K % 16 != 0).  A 4-deep lane group is entirely inside K, entirely outside (it reads a valid
    // address and contributes zeros) or -- K % 4 != 0, k-contiguous A only -- straddles the end: that one group loads element by
    // element (nothing is read past the end of a row, so the last row of a matrix never reads past its allocation)
    if ((g.K & 15) && (full_end % KW) == wave) {
        const int k = (full_end << 4) + 4 * lg;
        f32x4 fa[MT], fb[NT];
        if (TA || k + 3 < g.K || k >= g.K) {
            load_chunk(k < g.K ? k : 0, fa, fb);
            if (k >= g.K) {
#pragma unroll
                for (int a = 0; a < MT; ++a) fa[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int b = 0; b < NT; ++b) fb[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        } else {
#pragma unroll
            for (int a = 0; a < MT; ++a) fa[a] = ld_kcontig(gA + offA[TA ? 0 : a], 0, 0, true, k, g.K, false);
            if (!TB) {
                f32x4 w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    w[j] = (k + j < g.K) ? *(gcf4)(gB + (size_t)(k + j) * g.ldb + offB[0]) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int b = 0; b < NT; ++b) fb[b] = (f32x4){w[0][b], w[1][b], w[2][b], w[3][b]};
            } else {
#pragma unroll
                for (int b = 0; b < NT; ++b) fb[b] = ld_kcontig(gB + offB[TB ? b : 0], 0, 0, true, k, g.K, false);
            }
        }
        mfma_chunk<MT, NT, BF>(acc, fa, fb);
        if (want_colsum) {
#pragma unroll
            for (int b = 0; b < NT; ++b) csum[b] += (fb[b].x + fb[b].y) + (fb[b].z + fb[b].w);
        }
    }

    // accumulators -> LDS tile in OUTPUT coordinates (C/D map of a 16x16 tile: column index li, row index 4 lg + r)
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = TA ? 4 * (4 * lg + r) + a : 16 * a + 4 * lg + r;
            if (!TB) {
                *(f32x4 *)&s_tile[wave][row * LDT + 4 * li] = (f32x4){acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
            } else {
#pragma unroll
                for (int b = 0; b < NT; ++b) s_tile[wave][row * LDT + 16 * b + li] = acc[a][b][r];
            }
        }
    if (want_colsum) {
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            float v = csum[b];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lg == 0) s_col[wave][TB ? 16 * b + li : 4 * li + b] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = threadIdx.x + NTH * i;
        const int r = e / TN, cidx = e - r * TN;
        const int m = m0 + r, n = n0 + cidx;
        const int off = r * LDT + cidx;
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < KW; q += 4) v += (s_tile[q][off] + s_tile[q + 1][off]) + (s_tile[q + 2][off] + s_tile[q + 3][off]);
        if (g.beta != 0.f) v += g.beta * e_c[i];
        switch (g.epi) {
            case AIR_EPI_BIAS: v += e_bias[i]; break;
            case AIR_EPI_BIAS_ELU: v = elu_acc(v + e_bias[i]); break;
            case AIR_EPI_MUL_DELU: v *= (e_aux[i] > 0.f ? 1.f : e_aux[i] + 1.f); break;
            case AIR_EPI_ADD_AUX: v += e_aux[i] + e_bias[i]; break;
            case AIR_EPI_ADD_AUX_ELU: v = elu_acc(v + e_aux[i] + e_bias[i]); break;
            default: break;
        }
        if (m < g.M && n < g.N) gC[(size_t)m * g.ldc + n] = v;
    }
    if (want_colsum && threadIdx.x < TN) {
        const int n = n0 + threadIdx.x;
        if (n < g.N) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < KW; q += 4)
                v += (s_col[q][threadIdx.x] + s_col[q + 1][threadIdx.x]) + (s_col[q + 2][threadIdx.x] + s_col[q + 3][threadIdx.x]);
            gCol[n] = v;
        }
    }
}
template <int MT, int KW, bool BF, bool TA, bool TB>
__global__ __launch_bounds__(64 * KW) void gemm_wide_kernel(GemmArgs g) {
    gemm_wide_body<MT, KW, BF, TA, TB>(g, blockIdx.x);
}

// ---- the bf16 DATA path (BASELINE configs[4]) --------------------------------------------------------------------------------------
// gemm_wide_body<BF = true> only ROUNDS to bf16: every operand is still fetched as fp32 and fed to the CDNA3-form 16-deep MFMA.
// Here the operands may live in memory as bf16 -- a bf16 shadow of the flat parameter buffer (rewritten by the optimiser launch)
// and bf16 mirrors of the activations / gradients that GEMM epilogues produce (written by the producing epilogue next to the fp32
// value the non-GEMM consumers keep using) -- so an operand costs half the bytes, and the product runs on the gfx950 instruction
// v_mfma_f32_16x16x32_bf16: 32 k-slots per instruction, lane (i, lg) supplies slots 8 lg .. 8 lg + 7.  MFMA only needs A and B to
// agree on which k sits in a slot, so every loader below -- k-contiguous or interleaved along the contiguous dimension, fp32 or
// bf16 in memory -- produces "k = 32 c + 8 lg + j in slot 8 lg + j" and any combination of operand kinds multiplies correctly:
//   k-contiguous bf16 : ONE 16-byte load per lane per 32-deep chunk (8 consecutive k);
//   k-contiguous fp32 : two 16-byte loads, v_cvt_pk_bf16_f32 x4 (an operand without a mirror: LSTM state, sampled latents ...);
//   interleaved bf16  : eight 8-byte loads (row k + j, the lane's 4 columns) and a 16-bit transpose with v_perm_b32: the four
//                       values of a load go to four different MFMA tiles, exactly as in gemm_wide_body;
//   interleaved fp32  : eight 16-byte loads, converted pairwise.
// Same wave tile (16 MT x 64), same fixed-order K split over KW waves through LDS, same fused epilogues; the epilogue also writes
// the bf16 mirror of C when the descriptor names one.  fp32 accumulate, fp32 master copy of everything.
__device__ __forceinline__ float bf16_to_f32(unsigned bits16) { return __uint_as_float(bits16 << 16); }
__device__ __forceinline__ u32x4 pk8(f32x4 lo, f32x4 hi) {
    return (u32x4){pk_bf16(lo.x, lo.y), pk_bf16(lo.z, lo.w), pk_bf16(hi.x, hi.y), pk_bf16(hi.z, hi.w)};
}
// tile t (0..3) of eight 4-wide bf16 rows w[0..7] (row j = k + j; w[j].x = columns 0,1, w[j].y = columns 2,3)
template <int T_>
__device__ __forceinline__ u32x4 tr16(const u32x2 (&w)[8]) {
    constexpr unsigned sel = (T_ & 1) ? 0x07060302u : 0x05040100u;     // v_perm_b32: {hi half | lo half} of (S0, S1)
    u32x4 r;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const unsigned lo = (T_ < 2) ? w[2 * d].x : w[2 * d].y, hi = (T_ < 2) ? w[2 * d + 1].x : w[2 * d + 1].y;
        r[d] = __builtin_amdgcn_perm(hi, lo, sel);
    }
    return r;
}
template <int T_>
__device__ __forceinline__ u32x4 tr32(const f32x4 (&w)[8]) {
    return (u32x4){pk_bf16(w[0][T_], w[1][T_]), pk_bf16(w[2][T_], w[3][T_]), pk_bf16(w[4][T_], w[5][T_]), pk_bf16(w[6][T_], w[7][T_])};
}

template <int MT, int KW>
struct Wide16Lds {                       // declared once per KERNEL and handed to the body: a kernel that holds several
    float tile[KW][16 * MT * (64 + 4)];  // instantiations of the body (per-problem operand kinds) must not hold several copies
    float col[KW][64];
};
template <int MT, int KW, bool TA, bool TB, bool A16, bool B16>
__device__ __forceinline__ void gemm_wide16_body(const GemmArgs &g, const Gemm16Ptrs &h, const int block_tile, Wide16Lds<MT, KW> &lds) {
    constexpr int NT = 4, TM = 16 * MT, TN = 64;
    constexpr int NTH = 64 * KW;
    constexpr int LDT = TN + 4;
    static_assert(!TA || MT == 4, "an m-contiguous A operand is read 4 rows per lane: four interleaved row tiles");
    float (&s_tile)[KW][TM * LDT] = lds.tile;
    float (&s_col)[KW][TN] = lds.col;

    const gcf gA = (gcf)g.A, gB = (gcf)g.B, gBias = (gcf)g.bias, gAux = (gcf)g.aux;
    const gch hA = (gch)h.A16, hB = (gch)h.B16;
    const gf gC = (gf)g.C, gCol = (gf)g.colsum;
    const gh_t hC = (gh_t)h.C16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int tiles_n = (g.N + TN - 1) / TN;
    const int tm = block_tile / tiles_n, tn = block_tile - tm * tiles_n;
    const int m0 = tm * TM, n0 = tn * TN;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float csum[NT] = {0.f, 0.f, 0.f, 0.f};
    const bool want_colsum = TA && (g.colsum != nullptr) && (tm == 0);

    int offA[TA ? 1 : MT], offB[TB ? NT : 1];
    if (TA) { int r = m0 + 4 * li; if (r > g.M - 4) r = g.M - 4; offA[0] = r; }
    else {
#pragma unroll
        for (int a = 0; a < MT; ++a) { int r = m0 + 16 * a + li; if (r > g.M - 1) r = g.M - 1; offA[a] = r * g.lda; }
    }
    if (!TB) { int c = n0 + 4 * li; if (c > g.N - 4) c = g.N - 4; offB[0] = c; }
    else {
#pragma unroll
        for (int b = 0; b < NT; ++b) { int c = n0 + 16 * b + li; if (c > g.N - 1) c = g.N - 1; offB[b] = c * g.ldb; }
    }

    constexpr int EPT = (TM * TN) / NTH;
    static_assert((TM * TN) % NTH == 0, "tile / threads");
    float e_bias[EPT], e_aux[EPT], e_c[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = threadIdx.x + NTH * i;
        const int r = e / TN, cidx = e - r * TN;
        int m = m0 + r, n = n0 + cidx;
        if (m > g.M - 1) m = g.M - 1;
        if (n > g.N - 1) n = g.N - 1;
        e_bias[i] = g.bias != nullptr ? gBias[n] : 0.f;
        e_aux[i] = g.epi >= AIR_EPI_MUL_DELU ? gAux[(size_t)m * g.ldaux + n] : 0.f;
        e_c[i] = g.beta != 0.f ? gC[(size_t)m * g.ldc + n] : 0.f;
    }

    // one 32-deep chunk: k = first of the lane group's eight k values, kn = number of them inside K (8 in the main loop)
    auto load_chunk = [&](int k, int kn, u32x4 (&fa)[MT], u32x4 (&fb)[NT], bool full) {
        if (TA) {
            if (A16) {
                u32x2 w[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int kj = (full || j < kn) ? k + j : k;
                    w[j] = *(gcu2)(hA + (size_t)kj * g.lda + offA[0]);
                    if (!full && j >= kn) w[j] = (u32x2){0u, 0u};
                }
                fa[0] = tr16<0>(w); fa[1] = tr16<1>(w); fa[2] = tr16<2>(w); fa[3] = tr16<3>(w);
            } else {
                f32x4 w[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int kj = (full || j < kn) ? k + j : k;
                    w[j] = *(gcf4)(gA + (size_t)kj * g.lda + offA[0]);
                    if (!full && j >= kn) w[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                fa[0] = tr32<0>(w); fa[1] = tr32<1>(w); fa[2] = tr32<2>(w); fa[3] = tr32<3>(w);
            }
        } else {
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                if (full) {
                    if (A16) fa[a] = *(gcu4)(hA + offA[a] + k);
                    else fa[a] = pk8(*(gcf4)(gA + offA[a] + k), *(gcf4)(gA + offA[a] + k + 4));
                } else {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        v[j] = j < kn ? (A16 ? bf16_to_f32(hA[offA[a] + k + j]) : gA[offA[a] + k + j]) : 0.f;
                    fa[a] = (u32x4){pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
                }
            }
        }
        if (!TB) {
            if (B16) {
                u32x2 w[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int kj = (full || j < kn) ? k + j : k;
                    w[j] = *(gcu2)(hB + (size_t)kj * g.ldb + offB[0]);
                    if (!full && j >= kn) w[j] = (u32x2){0u, 0u};
                }
                fb[0] = tr16<0>(w); fb[1] = tr16<1>(w); fb[2] = tr16<2>(w); fb[3] = tr16<3>(w);
                if (want_colsum) {       // the bias gradient sums the UNROUNDED gradient: the first row of tiles also reads the fp32 rows
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (full || j < kn) {
                            const f32x4 r = *(gcf4)(gB + (size_t)(k + j) * g.ldb + offB[0]);
                            csum[0] += r.x; csum[1] += r.y; csum[2] += r.z; csum[3] += r.w;
                        }
                    }
                }
            } else {
                f32x4 w[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int kj = (full || j < kn) ? k + j : k;
                    w[j] = *(gcf4)(gB + (size_t)kj * g.ldb + offB[0]);
                    if (!full && j >= kn) w[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                fb[0] = tr32<0>(w); fb[1] = tr32<1>(w); fb[2] = tr32<2>(w); fb[3] = tr32<3>(w);
                if (want_colsum) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { csum[0] += w[j].x; csum[1] += w[j].y; csum[2] += w[j].z; csum[3] += w[j].w; }
                }
            }
        } else {
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                if (full) {
                    if (B16) fb[b] = *(gcu4)(hB + offB[b] + k);
                    else fb[b] = pk8(*(gcf4)(gB + offB[b] + k), *(gcf4)(gB + offB[b] + k + 4));
                } else {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        v[j] = j < kn ? (B16 ? bf16_to_f32(hB[offB[b] + k + j]) : gB[offB[b] + k + j]) : 0.f;
                    fb[b] = (u32x4){pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
                }
            }
        }
    };
    auto mma = [&](const u32x4 (&fa)[MT], const u32x4 (&fb)[NT]) {
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[a]), __builtin_bit_cast(bf16x8, fb[b]),
                                                                   acc[a][b], 0, 0, 0);
    };

    constexpr int U = (MT == 4) ? ((A16 && B16) ? 2 : 1) : 2;          // 32-deep chunks in flight per wave
    const int full_end = g.K >> 5;
    int c = wave;
#pragma nounroll
    for (; c < full_end; c += U * KW) {
        u32x4 fa[U][MT], fb[U][NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int cu = c + u * KW;
            if (cu > full_end - 1) cu = full_end - 1;      // a short tail group re-reads the last chunk (never multiplied)
            load_chunk((cu << 5) + 8 * lg, 8, fa[u], fb[u], true);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c + u * KW >= full_end) break;
            mma(fa[u], fb[u]);
        }
    }
    // the partial last chunk (K % 32 != 0), element-masked, by the wave whose turn it is
    if ((g.K & 31) && (full_end % KW) == wave) {
        const int k = (full_end << 5) + 8 * lg;
        int kn = g.K - k;
        kn = kn < 0 ? 0 : (kn > 8 ? 8 : kn);
        u32x4 fa[MT], fb[NT];
        load_chunk(kn > 0 ? k : 0, kn, fa, fb, false);
        mma(fa, fb);
    }

#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = TA ? 4 * (4 * lg + r) + a : 16 * a + 4 * lg + r;
            if (!TB) {
                *(f32x4 *)&s_tile[wave][row * LDT + 4 * li] = (f32x4){acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
            } else {
#pragma unroll
                for (int b = 0; b < NT; ++b) s_tile[wave][row * LDT + 16 * b + li] = acc[a][b][r];
            }
        }
    if (want_colsum) {
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            float v = csum[b];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lg == 0) s_col[wave][TB ? 16 * b + li : 4 * li + b] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = threadIdx.x + NTH * i;
        const int r = e / TN, cidx = e - r * TN;
        const int m = m0 + r, n = n0 + cidx;
        const int off = r * LDT + cidx;
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < KW; q += 4) v += (s_tile[q][off] + s_tile[q + 1][off]) + (s_tile[q + 2][off] + s_tile[q + 3][off]);
        if (g.beta != 0.f) v += g.beta * e_c[i];
        switch (g.epi) {
            case AIR_EPI_BIAS: v += e_bias[i]; break;
            case AIR_EPI_BIAS_ELU: v = elu_acc(v + e_bias[i]); break;
            case AIR_EPI_MUL_DELU: v *= (e_aux[i] > 0.f ? 1.f : e_aux[i] + 1.f); break;
            case AIR_EPI_ADD_AUX: v += e_aux[i] + e_bias[i]; break;
            case AIR_EPI_ADD_AUX_ELU: v = elu_acc(v + e_aux[i] + e_bias[i]); break;
            default: break;
        }
        if (m < g.M && n < g.N) {
            gC[(size_t)m * g.ldc + n] = v;
            if (hC) hC[(size_t)m * g.ldc + n] = bf16_bits(v);
        }
    }
    if (want_colsum && threadIdx.x < TN) {
        const int n = n0 + threadIdx.x;
        if (n < g.N) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < KW; q += 4)
                v += (s_col[q][threadIdx.x] + s_col[q + 1][threadIdx.x]) + (s_col[q + 2][threadIdx.x] + s_col[q + 3][threadIdx.x]);
            gCol[n] = v;
        }
    }
}
template <int MT, int KW, bool TA, bool TB, bool A16, bool B16>
__global__ __launch_bounds__(64 * KW) void gemm_wide16_kernel(GemmArgs g, Gemm16Ptrs h) {
    __shared__ Wide16Lds<MT, KW> lds;
    gemm_wide16_body<MT, KW, TA, TB, A16, B16>(g, h, blockIdx.x, lds);
}

// Several independent GEMMs in ONE launch (the step is launch/latency bound: a dW / dX pair, or the two heads that
// read the same hidden state, cost one dispatch instead of two).  Problem p owns blocks [tile_start[p], tile_start[p+1]).
#define AIR_GEMM_GROUP_MAX 8
struct GroupArgs {
    GemmArgs g[AIR_GEMM_GROUP_MAX];
    int tile_start[AIR_GEMM_GROUP_MAX + 1];
    int count;
    int xcd_map;          // wide-tile launches: blockIdx -> tile through xcd_contiguous_tile
    unsigned sk_mask;     // bit p: problem p is a short-K weight gradient on the streaming body (shortk_dw_body); its range = its workgroups
};
template <int MT, int NT, int KW, bool BF>
__global__ __launch_bounds__(KW == 1 ? 256 : 64 * KW) void gemm_grouped_kernel(GroupArgs ga) {
    int p = 0;
#pragma unroll
    for (int i = 1; i < AIR_GEMM_GROUP_MAX; ++i)
        if (i < ga.count && (int)blockIdx.x >= ga.tile_start[i]) p = i;
    p = __builtin_amdgcn_readfirstlane(p);                 // provably wave-uniform
    // One specialised copy of the body per descriptor slot: with a CONSTANT index the descriptor is read straight from
    // the kernarg SGPRs like a single-GEMM launch (94 VGPRs).  A dynamic index (or a descriptor copied through
    // memory) costs +100 VGPRs and halves the occupancy of exactly the launches that have the most workgroups.
    // (blockIdx.y is always 0 here, but passing it instead of a literal 0 keeps hipcc from restructuring the K loop
    //  around a known start, which doubles the live registers: 94 -> 194 VGPRs, measured)
    switch (p) {
        case 0: gemm_body<MT, NT, KW, BF>(ga.g[0], (int)blockIdx.x - ga.tile_start[0], blockIdx.y); break;
        case 1: gemm_body<MT, NT, KW, BF>(ga.g[1], (int)blockIdx.x - ga.tile_start[1], blockIdx.y); break;
        case 2: gemm_body<MT, NT, KW, BF>(ga.g[2], (int)blockIdx.x - ga.tile_start[2], blockIdx.y); break;
        case 3: gemm_body<MT, NT, KW, BF>(ga.g[3], (int)blockIdx.x - ga.tile_start[3], blockIdx.y); break;
        case 4: gemm_body<MT, NT, KW, BF>(ga.g[4], (int)blockIdx.x - ga.tile_start[4], blockIdx.y); break;
        case 5: gemm_body<MT, NT, KW, BF>(ga.g[5], (int)blockIdx.x - ga.tile_start[5], blockIdx.y); break;
        case 6: gemm_body<MT, NT, KW, BF>(ga.g[6], (int)blockIdx.x - ga.tile_start[6], blockIdx.y); break;
        default: gemm_body<MT, NT, KW, BF>(ga.g[7], (int)blockIdx.x - ga.tile_start[7], blockIdx.y); break;
    }
}

// gemm_grouped_kernel<1, 1, 16> whose problems read their A rows through the feeder's index (GatherArgs): the first launch of a
// latency-regime train step with an HBM-resident dataset attached (air_gemm_grouped_gather)
__global__ __launch_bounds__(1024) void gemm_grouped_gather_kernel(GroupArgs ga, GatherArgs gt) {
    int p = 0;
#pragma unroll
    for (int i = 1; i < AIR_GEMM_GROUP_MAX; ++i)
        if (i < ga.count && (int)blockIdx.x >= ga.tile_start[i]) p = i;
    p = __builtin_amdgcn_readfirstlane(p);
    const bool cp = (gt.copy_mask >> p) & 1u;
#define AIR_GG_CASE(I_) gemm_body<1, 1, 16, false, false, false, false, true>(ga.g[I_], (int)blockIdx.x - ga.tile_start[I_], blockIdx.y, AproArgs(), nullptr, nullptr, false, nullptr, &gt, cp)
    switch (p) {       // constant descriptor index per copy, as in gemm_grouped_kernel
        case 0: AIR_GG_CASE(0); break;
        case 1: AIR_GG_CASE(1); break;
        case 2: AIR_GG_CASE(2); break;
        case 3: AIR_GG_CASE(3); break;
        case 4: AIR_GG_CASE(4); break;
        case 5: AIR_GG_CASE(5); break;
        case 6: AIR_GG_CASE(6); break;
        default: AIR_GG_CASE(7); break;
    }
#undef AIR_GG_CASE
}

// gemm_grouped_kernel whose weight-gradient problems (fold_mask) apply the centred-RMSProp update to the elements they finish, with
// rider workgroups (blockIdx >= tiles) updating the slices earlier launches left final and advancing the step counters
template <int MT, int NT, int KW, bool BF>
__global__ __launch_bounds__(KW == 1 ? 256 : 64 * KW) void gemm_grouped_opt_kernel(GroupArgs ga, OptFold opt) {
    if ((int)blockIdx.x >= opt.tiles) {
        const int vb = (int)blockIdx.x - opt.tiles, vg = (int)gridDim.x - opt.tiles;
        for (int r = 0; r < opt.n_ranges; ++r) {
            RmspropSlice sl;
            sl.p = opt.p; sl.g = opt.g0; sl.ms = opt.ms; sl.mg = opt.mg; sl.mom = opt.mom;
            sl.lo = opt.lo[r]; sl.hi = opt.hi[r]; sl.n_model = opt.n_model; sl.lr_dev = opt.lr_dev;
            sl.lr_mult_tail = opt.lr_mult_tail; sl.decay = opt.decay; sl.momentum = opt.momentum; sl.eps = opt.eps; sl.gscale = opt.gscale;
            rmsprop_slice_body(sl, vb, vg);
        }
        if (vb == 0 && threadIdx.x == 0) {
            if (opt.gstep) opt.gstep[0] += 1;
            if (opt.rng_state) opt.rng_state[1] += opt.rng_inc;
        }
        return;
    }
    int p = 0;
#pragma unroll
    for (int i = 1; i < AIR_GEMM_GROUP_MAX; ++i)
        if (i < ga.count && (int)blockIdx.x >= ga.tile_start[i]) p = i;
    p = __builtin_amdgcn_readfirstlane(p);
    // (ONE instantiation of the body per slot, the fold a wave-uniform run-time flag: two instantiations would each bring their own
    //  static LDS tile and halve the workgroups a CU holds)
#define AIR_OPT_CASE(I_)                                                                                                          \
    gemm_body<MT, NT, KW, BF, false, true>(ga.g[I_], (int)blockIdx.x - ga.tile_start[I_], blockIdx.y, AproArgs(), nullptr, &opt,  \
                                           ((opt.fold_mask >> I_) & 1u) != 0)
    switch (p) {
        case 0: AIR_OPT_CASE(0); break;
        case 1: AIR_OPT_CASE(1); break;
        case 2: AIR_OPT_CASE(2); break;
        case 3: AIR_OPT_CASE(3); break;
        case 4: AIR_OPT_CASE(4); break;
        case 5: AIR_OPT_CASE(5); break;
        case 6: AIR_OPT_CASE(6); break;
        default: AIR_OPT_CASE(7); break;
    }
#undef AIR_OPT_CASE
}

// ---- short-K weight gradients of the wide first layers (latency regime: dW[M, N] = X^T . dY with K = batch rows <= 64, M = 2500 /
// 10000 pixels, N <= 256): thousands of output tiles each fed by ONE 16-deep chunk per wave.  On the tile kernels every workgroup is
// a chain of one cold operand round trip, an LDS reduction of four single-chunk partials, a barrier and the epilogue -- 2500 such
// workgroups in three resident rounds: 10-17 us beside a 5 us critical path, 30 us where the update rides (configs[3]).  Here the
// product STREAMS: a wave keeps the whole dY[K, 64 columns] operand of its column quarter in registers (K <= 64: sixteen float4) and
// walks 16-row slabs of the output grid-stride -- four dword loads per chunk of X^T, sixteen MFMAs per chunk, the accumulators
// stored (and, folded, updated) straight from the MFMA layout; the next slab's operand is requested before the current one is
// multiplied.  No LDS, no barrier, no K split (one accumulation chain per element: a fixed order).
template <bool OPT>
__device__ __forceinline__ void shortk_dw_body(const GemmArgs &g, const int vb, const int vg, const OptFold *opt, const bool fold) {
    const gcf gA = (gcf)g.A, gB = (gcf)g.B;
    const gf gC = (gf)g.C, gCol = (gf)g.colsum;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int n_base = 64 * wave;
    if (n_base >= g.N) return;                                   // (N % 64 == 0: whole quarters; no barrier anywhere below)
    const int nch = g.K >> 4;                                    // K % 16 == 0, K <= 64
    f32x4 fb[4][4];                                              // fb[c][t][j] = dY[16c + 4lg + j, n_base + 16t + li]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            fb[c][t] = c < nch ? ld_kstrided_full(gB, g.ldb, n_base + 16 * t + li, (c << 4) + 4 * lg) : (f32x4){0.f, 0.f, 0.f, 0.f};
    float o_lr0 = 0.f;
    if (OPT && fold) o_lr0 = ((gcf)opt->lr_dev)[0];
    if (gCol != nullptr && vb == 0) {                            // the bias gradient: column sums of dY, by the first workgroup
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float v = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) v += (fb[c][t].x + fb[c][t].y) + (fb[c][t].z + fb[c][t].w);
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lg == 0) {
                const int n = n_base + 16 * t + li;
                gCol[n] = v;
                if (OPT && fold) {
                    const size_t idx = (size_t)((gCol + n) - (gf)opt->g0);
                    const float lr = idx < opt->n_model ? o_lr0 : o_lr0 * opt->lr_mult_tail;
                    float pv = ((gf)opt->p)[idx], a = ((gf)opt->ms)[idx], b = ((gf)opt->mg)[idx], cmo = ((gf)opt->mom)[idx];
                    rmsprop_elem(pv, v, a, b, cmo, lr, opt->decay, opt->momentum, opt->eps, opt->gscale);
                    ((gf)opt->ms)[idx] = a; ((gf)opt->mg)[idx] = b; ((gf)opt->mom)[idx] = cmo; ((gf)opt->p)[idx] = pv;
                }
            }
        }
    }
    const int tiles_m = (g.M + 15) >> 4;
    int mt = vb;
    if (mt >= tiles_m) return;
    f32x4 fa[4];
    {
        int m = (mt << 4) + li; if (m > g.M - 1) m = g.M - 1;
#pragma unroll
        for (int c = 0; c < 4; ++c) fa[c] = c < nch ? ld_kstrided_full(gA, g.lda, m, (c << 4) + 4 * lg) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma nounroll
    for (; mt < tiles_m; mt += vg) {
        f32x4 fn[4];
        {   // the next slab's operand, requested before this one is multiplied (clamped: a slab past the end is never used)
            int mn = ((mt + vg) << 4) + li; if (mn > g.M - 1) mn = g.M - 1;
#pragma unroll
            for (int c = 0; c < 4; ++c) fn[c] = c < nch ? ld_kstrided_full(gA, g.lda, mn, (c << 4) + 4 * lg) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const int mrow0 = (mt << 4) + 4 * lg;
        // folded update: parameter and slots of this lane's 16 elements, requested before the product
        f32x4 o_p[4], o_ms[4], o_mg[4], o_mom[4];               // [t][r]
#pragma unroll
        for (int t = 0; t < 4; ++t) { o_p[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; o_ms[t] = o_p[t]; o_mg[t] = o_p[t]; o_mom[t] = o_p[t]; }
        if (OPT && fold) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int mr = mrow0 + r; if (mr > g.M - 1) mr = g.M - 1;
                const size_t idx0 = (size_t)((gC + (size_t)mr * g.ldc + n_base + li) - (gf)opt->g0);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    o_p[t][r] = ((gf)opt->p)[idx0 + 16 * t]; o_ms[t][r] = ((gf)opt->ms)[idx0 + 16 * t];
                    o_mg[t][r] = ((gf)opt->mg)[idx0 + 16 * t]; o_mom[t][r] = ((gf)opt->mom)[idx0 + 16 * t];
                }
            }
        }
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c][j], fb[c][t][j], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mr = mrow0 + r;
                if (mr < g.M) {
                    const size_t off = (size_t)mr * g.ldc + n_base + 16 * t + li;
                    const float v = acc[t][r];
                    gC[off] = v;
                    if (OPT && fold) {
                        const size_t idx = (size_t)((gC + off) - (gf)opt->g0);
                        const float lr = idx < opt->n_model ? o_lr0 : o_lr0 * opt->lr_mult_tail;
                        float pv = o_p[t][r], a = o_ms[t][r], b = o_mg[t][r], cmo = o_mom[t][r];
                        rmsprop_elem(pv, v, a, b, cmo, lr, opt->decay, opt->momentum, opt->eps, opt->gscale);
                        ((gf)opt->ms)[idx] = a; ((gf)opt->mg)[idx] = b; ((gf)opt->mom)[idx] = cmo; ((gf)opt->p)[idx] = pv;
                    }
                }
            }
#pragma unroll
        for (int c = 0; c < 4; ++c) fa[c] = fn[c];
    }
}
// what the streaming body takes (host side): an fp32 TN weight gradient with no epilogue, K in {16, 32, 48, 64}, N a multiple of 64 up
// to 256, and at least AIR_GEMM_SHORTK_MIN_M (4096) output rows.  Measured (profiles/r05_shortk_dw_ab.txt): at 10000 rows (configs[3])
// the launch beside the decoder's dX drops from 16.6 to 10.3 us and the closing launch with the folded update from 30.4 to 25.5 us;
// at 2500 rows (configs[1]: 157 slabs, ONE per workgroup, so the register-resident dY operand is loaded for a single slab) the
// closing launch is SLOWER (12.0 against 9.7 us) -- hence the row threshold.  The same rule in every entry point (a problem's
// product must not depend on which launch carries it: the folded and the unfolded plan are bit-identical).  AIR_GEMM_SHORTK=0: never.
static inline bool shortk_eligible(const AirGemmDesc &d) {
    static const int on = getenv("AIR_GEMM_SHORTK") ? atoi(getenv("AIR_GEMM_SHORTK")) : 1;
    static const int min_m = getenv("AIR_GEMM_SHORTK_MIN_M") ? atoi(getenv("AIR_GEMM_SHORTK_MIN_M")) : 4096;
    return on && d.ta && !d.tb && d.precision == AIR_PREC_F32 && d.K >= 16 && d.K <= 64 && d.K % 16 == 0 && d.N >= 64 && d.N <= 256 &&
           d.N % 64 == 0 && d.M >= min_m && d.epilogue == AIR_EPI_NONE && d.beta == 0.f && !d.A2 && !d.C16 && !d.bias;
}
static inline int shortk_workgroups(int M) {
    static const int cap = getenv("AIR_GEMM_SHORTK_WGS") ? atoi(getenv("AIR_GEMM_SHORTK_WGS")) : 384;
    const int t = air_cdiv(M, 16);
    return t < cap ? t : (cap < 1 ? 1 : cap);
}

// gemm_grouped_kernel (4-wave tiles) with short-K weight gradients of ga.sk_mask on the streaming body
template <int MT, int NT>
__global__ __launch_bounds__(256) void gemm_grouped_sk_kernel(GroupArgs ga) {
    int p = 0;
#pragma unroll
    for (int i = 1; i < AIR_GEMM_GROUP_MAX; ++i)
        if (i < ga.count && (int)blockIdx.x >= ga.tile_start[i]) p = i;
    p = __builtin_amdgcn_readfirstlane(p);
#define AIR_SK_CASE(I_)                                                                                                        \
    do {                                                                                                                       \
        if ((ga.sk_mask >> I_) & 1u)                                                                                           \
            shortk_dw_body<false>(ga.g[I_], (int)blockIdx.x - ga.tile_start[I_], ga.tile_start[I_ + 1] - ga.tile_start[I_], nullptr, false); \
        else gemm_body<MT, NT, 4, false>(ga.g[I_], (int)blockIdx.x - ga.tile_start[I_], blockIdx.y);                           \
    } while (0)
    switch (p) {
        case 0: AIR_SK_CASE(0); break;
        case 1: AIR_SK_CASE(1); break;
        case 2: AIR_SK_CASE(2); break;
        case 3: AIR_SK_CASE(3); break;
        case 4: AIR_SK_CASE(4); break;
        case 5: AIR_SK_CASE(5); break;
        case 6: AIR_SK_CASE(6); break;
        default: AIR_SK_CASE(7); break;
    }
#undef AIR_SK_CASE
}
// gemm_grouped_opt_kernel's counterpart for launches whose problems are ALL short-K weight gradients (the closing launch of the
// latency-regime step: the first layer over the pixels of obs): a folded problem updates its elements from the MFMA layout
__global__ __launch_bounds__(256) void gemm_grouped_opt_sk_kernel(GroupArgs ga, OptFold opt) {
    if ((int)blockIdx.x >= opt.tiles) {
        const int vb = (int)blockIdx.x - opt.tiles, vg = (int)gridDim.x - opt.tiles;
        for (int r = 0; r < opt.n_ranges; ++r) {
            RmspropSlice sl;
            sl.p = opt.p; sl.g = opt.g0; sl.ms = opt.ms; sl.mg = opt.mg; sl.mom = opt.mom;
            sl.lo = opt.lo[r]; sl.hi = opt.hi[r]; sl.n_model = opt.n_model; sl.lr_dev = opt.lr_dev;
            sl.lr_mult_tail = opt.lr_mult_tail; sl.decay = opt.decay; sl.momentum = opt.momentum; sl.eps = opt.eps; sl.gscale = opt.gscale;
            rmsprop_slice_body(sl, vb, vg);
        }
        if (vb == 0 && threadIdx.x == 0) {
            if (opt.gstep) opt.gstep[0] += 1;
            if (opt.rng_state) opt.rng_state[1] += opt.rng_inc;
        }
        return;
    }
    int p = 0;
#pragma unroll
    for (int i = 1; i < AIR_GEMM_GROUP_MAX; ++i)
        if (i < ga.count && (int)blockIdx.x >= ga.tile_start[i]) p = i;
    p = __builtin_amdgcn_readfirstlane(p);
#define AIR_SK_CASE(I_)                                                                                                        \
    shortk_dw_body<true>(ga.g[I_], (int)blockIdx.x - ga.tile_start[I_], ga.tile_start[I_ + 1] - ga.tile_start[I_], &opt,        \
                         ((opt.fold_mask >> I_) & 1u) != 0)
    switch (p) {
        case 0: AIR_SK_CASE(0); break;
        case 1: AIR_SK_CASE(1); break;
        case 2: AIR_SK_CASE(2); break;
        case 3: AIR_SK_CASE(3); break;
        case 4: AIR_SK_CASE(4); break;
        case 5: AIR_SK_CASE(5); break;
        case 6: AIR_SK_CASE(6); break;
        default: AIR_SK_CASE(7); break;
    }
#undef AIR_SK_CASE
}

// gemm_grouped_kernel whose problem(s) of `gb.mask` finish a Gaussian head's backward in their epilogue, with up to two rider
// workgroups behind the tiles: the NVIL objective (nvil_device.h; nv.imp == NULL: none) and the sum of the head's KL shares
// (engine_device.h kl_parts_sum; kp.n_parts == 0: none) -- what air_gauss_sample_bwd_nvil did as a launch of its own
template <int MT, int NT, int KW, bool BF>
__global__ __launch_bounds__(KW == 1 ? 256 : 64 * KW) void gemm_grouped_gb_kernel(GroupArgs ga, GaussEpi gb, NvilArgs nv, KlParts kp, int tiles,
                                                                                  int kl_rows) {
    if ((int)blockIdx.x >= tiles) {
        const int r = (int)blockIdx.x - tiles;
        if (r == 0 && nv.imp) nvil_body(nv);
        else kl_parts_sum(kp, kl_rows);
        return;
    }
    int p = 0;
#pragma unroll
    for (int i = 1; i < AIR_GEMM_GROUP_MAX; ++i)
        if (i < ga.count && (int)blockIdx.x >= ga.tile_start[i]) p = i;
    p = __builtin_amdgcn_readfirstlane(p);
#define AIR_GB_CASE(I_)                                                                                                                  \
    gemm_body<MT, NT, KW, BF, false, false, true>(ga.g[I_], (int)blockIdx.x - ga.tile_start[I_], blockIdx.y, AproArgs(), nullptr, nullptr,  \
                                                  ((gb.mask >> I_) & 1u) != 0, &gb)
    switch (p) {
        case 0: AIR_GB_CASE(0); break;
        case 1: AIR_GB_CASE(1); break;
        case 2: AIR_GB_CASE(2); break;
        case 3: AIR_GB_CASE(3); break;
        case 4: AIR_GB_CASE(4); break;
        case 5: AIR_GB_CASE(5); break;
        case 6: AIR_GB_CASE(6); break;
        default: AIR_GB_CASE(7); break;
    }
#undef AIR_GB_CASE
}

struct C16Ptrs { void *p[AIR_GEMM_GROUP_MAX]; };
template <int MT, int NT, int KW>
__global__ __launch_bounds__(KW == 1 ? 256 : 64 * KW) void gemm_grouped_c16_kernel(GroupArgs ga, C16Ptrs c16) {
    int p = 0;
#pragma unroll
    for (int i = 1; i < AIR_GEMM_GROUP_MAX; ++i)
        if (i < ga.count && (int)blockIdx.x >= ga.tile_start[i]) p = i;
    p = __builtin_amdgcn_readfirstlane(p);
    switch (p) {
        case 0: gemm_body<MT, NT, KW, true>(ga.g[0], (int)blockIdx.x - ga.tile_start[0], blockIdx.y, AproArgs(), c16.p[0]); break;
        case 1: gemm_body<MT, NT, KW, true>(ga.g[1], (int)blockIdx.x - ga.tile_start[1], blockIdx.y, AproArgs(), c16.p[1]); break;
        case 2: gemm_body<MT, NT, KW, true>(ga.g[2], (int)blockIdx.x - ga.tile_start[2], blockIdx.y, AproArgs(), c16.p[2]); break;
        case 3: gemm_body<MT, NT, KW, true>(ga.g[3], (int)blockIdx.x - ga.tile_start[3], blockIdx.y, AproArgs(), c16.p[3]); break;
        case 4: gemm_body<MT, NT, KW, true>(ga.g[4], (int)blockIdx.x - ga.tile_start[4], blockIdx.y, AproArgs(), c16.p[4]); break;
        case 5: gemm_body<MT, NT, KW, true>(ga.g[5], (int)blockIdx.x - ga.tile_start[5], blockIdx.y, AproArgs(), c16.p[5]); break;
        case 6: gemm_body<MT, NT, KW, true>(ga.g[6], (int)blockIdx.x - ga.tile_start[6], blockIdx.y, AproArgs(), c16.p[6]); break;
        default: gemm_body<MT, NT, KW, true>(ga.g[7], (int)blockIdx.x - ga.tile_start[7], blockIdx.y, AproArgs(), c16.p[7]); break;
    }
}

// consecutive workgroup ids go to different XCDs (each with its own L2): give every XCD a CONTIGUOUS range of tiles -- same
// problem, same row slab, neighbouring column slabs -- so that an operand slab is pulled into one L2, not into all eight
__device__ __forceinline__ int xcd_contiguous_tile(int b, int G) {
    const int full = G >> 3, rem = G & 7, x = b & 7, i = b >> 3;
    return x * full + (x < rem ? x : rem) + i;
}
template <int MT, int KW, bool BF, bool TA, bool TB>
__global__ __launch_bounds__(64 * KW) void gemm_grouped_wide_kernel(GroupArgs ga) {
    const int vb = ga.xcd_map ? xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    int p = 0;
#pragma unroll
    for (int i = 1; i < AIR_GEMM_GROUP_MAX; ++i)
        if (i < ga.count && vb >= ga.tile_start[i]) p = i;
    p = __builtin_amdgcn_readfirstlane(p);
    switch (p) {       // constant descriptor index per copy, as in gemm_grouped_kernel
        case 0: gemm_wide_body<MT, KW, BF, TA, TB>(ga.g[0], vb - ga.tile_start[0]); break;
        case 1: gemm_wide_body<MT, KW, BF, TA, TB>(ga.g[1], vb - ga.tile_start[1]); break;
        case 2: gemm_wide_body<MT, KW, BF, TA, TB>(ga.g[2], vb - ga.tile_start[2]); break;
        case 3: gemm_wide_body<MT, KW, BF, TA, TB>(ga.g[3], vb - ga.tile_start[3]); break;
        case 4: gemm_wide_body<MT, KW, BF, TA, TB>(ga.g[4], vb - ga.tile_start[4]); break;
        case 5: gemm_wide_body<MT, KW, BF, TA, TB>(ga.g[5], vb - ga.tile_start[5]); break;
        case 6: gemm_wide_body<MT, KW, BF, TA, TB>(ga.g[6], vb - ga.tile_start[6]); break;
        default: gemm_wide_body<MT, KW, BF, TA, TB>(ga.g[7], vb - ga.tile_start[7]); break;
    }
}

struct GroupArgs16 {
    GemmArgs g[AIR_GEMM_GROUP_MAX];
    Gemm16Ptrs h[AIR_GEMM_GROUP_MAX];
    int tile_start[AIR_GEMM_GROUP_MAX + 1];
    int count;
    int xcd_map;
};
template <int MT, int KW, bool TA, bool TB, bool A16, bool B16>
__global__ __launch_bounds__(64 * KW) void gemm_grouped_wide16_kernel(GroupArgs16 ga) {
    __shared__ Wide16Lds<MT, KW> lds;
    const int vb = ga.xcd_map ? xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    int p = 0;
#pragma unroll
    for (int i = 1; i < AIR_GEMM_GROUP_MAX; ++i)
        if (i < ga.count && vb >= ga.tile_start[i]) p = i;
    p = __builtin_amdgcn_readfirstlane(p);
    switch (p) {       // constant descriptor index per copy, as in gemm_grouped_kernel
        case 0: gemm_wide16_body<MT, KW, TA, TB, A16, B16>(ga.g[0], ga.h[0], vb - ga.tile_start[0], lds); break;
        case 1: gemm_wide16_body<MT, KW, TA, TB, A16, B16>(ga.g[1], ga.h[1], vb - ga.tile_start[1], lds); break;
        case 2: gemm_wide16_body<MT, KW, TA, TB, A16, B16>(ga.g[2], ga.h[2], vb - ga.tile_start[2], lds); break;
        case 3: gemm_wide16_body<MT, KW, TA, TB, A16, B16>(ga.g[3], ga.h[3], vb - ga.tile_start[3], lds); break;
        case 4: gemm_wide16_body<MT, KW, TA, TB, A16, B16>(ga.g[4], ga.h[4], vb - ga.tile_start[4], lds); break;
        case 5: gemm_wide16_body<MT, KW, TA, TB, A16, B16>(ga.g[5], ga.h[5], vb - ga.tile_start[5], lds); break;
        case 6: gemm_wide16_body<MT, KW, TA, TB, A16, B16>(ga.g[6], ga.h[6], vb - ga.tile_start[6], lds); break;
        default: gemm_wide16_body<MT, KW, TA, TB, A16, B16>(ga.g[7], ga.h[7], vb - ga.tile_start[7], lds); break;
    }
}

// All weight gradients of a step in ONE launch: up to 24 all-TN problems (a dynamic, wave-uniform descriptor index: scalar
// loads from the kernel-argument segment).  With every long-K and short-K tile in the same grid -- long ones first -- the CUs
// that finish early keep pulling short tiles instead of idling until the launch ends (two launches of 8: 72 + 69 us in fp32 at
// batch 1024 with 184 and 408 tiles on 256 CUs).
#define AIR_GEMM_BIG_GROUP_MAX 24
struct BigGroupArgs {
    GemmArgs g[AIR_GEMM_BIG_GROUP_MAX];
    int tile_start[AIR_GEMM_BIG_GROUP_MAX + 1];
    int count;
    int xcd_map;          // 0: none, 1: XCD-contiguous tiles within each problem, 2: over the whole grid
    unsigned small_mask;  // bit p: problem p is not wide-tile eligible (one to three rows, one column ...): 16x16 tiles, gemm_body
};
// the same within ONE problem's range of workgroup ids [s, e): the problem keeps its place in the dispatch order (long-K problems
// first), and the workgroups of it that land on one XCD (ids congruent mod 8) get a contiguous run of its tiles
__device__ __forceinline__ int xcd_contiguous_tile_in_range(int b, int s, int e) {
    const int x = b & 7;
    int start = 0, first_x = 0;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        const int f = s + ((y - s) & 7);                    // first id >= s on XCD y
        const int cnt = f < e ? ((e - 1 - f) >> 3) + 1 : 0;
        if (y < x) start += cnt;
        if (y == x) first_x = f;
    }
    return s + start + ((b - first_x) >> 3);
}
// (the odd-shaped rest of the weight gradients -- eight long-K reductions with one to three rows or a single column -- rides in
//  the same grid on the 16x16-tile body with the same 8 K-splitting waves: a launch of their own cost 14.5 us at batch 1024)
template <int MT, int KW, bool BF>
__global__ __launch_bounds__(64 * KW) void gemm_big_group_wide_tn_kernel(BigGroupArgs ga) {
    const int b = ga.xcd_map == 2 ? xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    int p = 0;
    for (int i = 1; i < ga.count; ++i)
        if (b >= ga.tile_start[i]) p = i;
    p = __builtin_amdgcn_readfirstlane(p);
    const int s0 = ga.tile_start[p];
    if ((ga.small_mask >> p) & 1u) {
        gemm_body<1, 1, KW, BF>(ga.g[p], b - s0, 0);
        return;
    }
    const int vb = ga.xcd_map == 1 ? xcd_contiguous_tile_in_range(b, s0, ga.tile_start[p + 1]) : b;
    gemm_wide_body<MT, KW, BF, true, false>(ga.g[p], vb - s0);
}

// bf16 data path: the same launch with the operands' bf16 mirrors (weight gradients: A = an activation, B = a gradient, both
// interleaved reads); the odd-shaped members keep the fp32-fetch small-tile body
struct BigGroupArgs16 {
    GemmArgs g[AIR_GEMM_BIG_GROUP_MAX];
    Gemm16Ptrs h[AIR_GEMM_BIG_GROUP_MAX];
    int tile_start[AIR_GEMM_BIG_GROUP_MAX + 1];
    int count;
    int xcd_map;
    unsigned small_mask;
};
// (the operand kinds are per PROBLEM here -- a weight-gradient launch mixes activations with and without a mirror -- chosen by a
//  wave-uniform branch between the four instantiations of the body)
template <int MT, int KW>
__global__ __launch_bounds__(64 * KW) void gemm_big_group_wide16_tn_kernel(BigGroupArgs16 ga) {
    __shared__ Wide16Lds<MT, KW> lds;
    const int b = ga.xcd_map == 2 ? xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    int p = 0;
    for (int i = 1; i < ga.count; ++i)
        if (b >= ga.tile_start[i]) p = i;
    p = __builtin_amdgcn_readfirstlane(p);
    const int s0 = ga.tile_start[p];
    if ((ga.small_mask >> p) & 1u) {
        gemm_body<1, 1, KW, true>(ga.g[p], b - s0, 0);
        return;
    }
    const int vb = ga.xcd_map == 1 ? xcd_contiguous_tile_in_range(b, s0, ga.tile_start[p + 1]) : b;
    const bool a16 = ga.h[p].A16 != nullptr, b16 = ga.h[p].B16 != nullptr;
    if (a16 && b16) gemm_wide16_body<MT, KW, true, false, true, true>(ga.g[p], ga.h[p], vb - s0, lds);
    else if (a16) gemm_wide16_body<MT, KW, true, false, true, false>(ga.g[p], ga.h[p], vb - s0, lds);
    else if (b16) gemm_wide16_body<MT, KW, true, false, false, true>(ga.g[p], ga.h[p], vb - s0, lds);
    else gemm_wide16_body<MT, KW, true, false, false, false>(ga.g[p], ga.h[p], vb - s0, lds);
}

// sums the S split-K slabs in fixed order and applies the epilogue
__global__ __launch_bounds__(256) void gemm_splitk_epilogue_kernel(GemmArgs g) {
    const size_t total = (size_t)g.M * g.N;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int m = (int)(e / g.N), n = (int)(e - (size_t)m * g.N);
        float v = 0.f;
        for (int s = 0; s < g.S; ++s) v += g.ws[(size_t)s * total + e];
        g.C[(size_t)m * g.ldc + n] = apply_epilogue(v, m, n, g);
    }
}

template <int MT, int NT, int KW, bool BF>
static int launch_gemm(const GemmArgs &g, hipStream_t st) {
    constexpr int TM = 16 * MT, TN = 16 * NT * ((KW == 1) ? 4 : 1);
    const int tiles = air_cdiv(g.M, TM) * air_cdiv(g.N, TN);
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<MT, NT, KW, BF>), dim3(tiles, g.S), dim3(KW == 1 ? 256 : 64 * KW), 0, st, g);
    AIR_LAUNCH_CHECK();
    if (g.S > 1) {
        const size_t total = (size_t)g.M * g.N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(gemm_splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, st, g);
        AIR_LAUNCH_CHECK();
    }
    return AIR_OK;
}

extern "C" size_t air_gemm_workspace_bytes(int M, int N, int K) {
    (void)K;
    if (M <= 0 || N <= 0) return 0;
    return (size_t)16 * (size_t)M * (size_t)N * sizeof(float);     // up to 16 split-K slabs
}

template <bool BF>
static int gemm_dispatch(int ta, int tb, int M, int N, int K, const float *A, int lda, const float *B, int ldb,
                         float *C, int ldc, const float *bias, int epilogue, const float *aux, int ldaux, float beta,
                         float *colsum, void *ws, size_t ws_bytes, void *stream) {
    AIR_REQUIRE(A && B && C, AIR_E_NULL);
    AIR_REQUIRE(M > 0 && N > 0 && K > 0, AIR_E_SHAPE);
    AIR_REQUIRE(lda >= (ta ? M : K) && ldb >= (tb ? K : N) && ldc >= N, AIR_E_SHAPE);
    AIR_REQUIRE(epilogue >= AIR_EPI_NONE && epilogue <= AIR_EPI_ADD_AUX_ELU, AIR_E_UNSUPPORTED);
    if (epilogue == AIR_EPI_BIAS || epilogue == AIR_EPI_BIAS_ELU) AIR_REQUIRE(bias, AIR_E_NULL);
    if (epilogue >= AIR_EPI_MUL_DELU) AIR_REQUIRE(aux && ldaux >= N, AIR_E_NULL);
    AIR_REQUIRE(!colsum || ta, AIR_E_UNSUPPORTED);

    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.aux = aux; g.colsum = colsum; g.ws = (float *)ws;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = ldaux;
    g.ta = ta ? 1 : 0; g.tb = tb ? 1 : 0; g.epi = epilogue; g.beta = beta;
    g.vecA = (!ta && (lda % 4 == 0) && air_aligned16(A)) ? 1 : 0;
    g.vecB = (tb && (ldb % 4 == 0) && air_aligned16(B)) ? 1 : 0;

    const int chunks = (K + 15) / 16;
    // tile shape: few rows -> narrow tiles + intra-workgroup split-K; many tiles -> one tile per wave
    const long tiles22 = (long)air_cdiv(M, 32) * air_cdiv(N, 32);
    int S = 1;
    hipStream_t st = air_stream(stream);
    if (tiles22 >= 2048) {
        g.S = 1; g.chunks_per_split = chunks;
        return launch_gemm<2, 2, 1, BF>(g, st);
    }
    // latency regime (the whole problem fits a fraction of the chip): 16x16 tiles, one per workgroup, 4 waves split K,
    // 8 chunks in flight -- every wave does one or two memory round trips whatever the shape
    const bool narrow = true;
    const long tiles = (long)air_cdiv(M, 16) * air_cdiv(N, 16);
    // cross-workgroup split-K costs a second (epilogue) launch, ~4.5 us on this part: only worth it for long K
    if (!colsum && ws && chunks >= 64) {
        long want = 1024 / (tiles > 0 ? tiles : 1);                      // aim for ~4 workgroups per CU
        long max_by_k = chunks / 16;                                     // >= 4 chunks per wave per split
        long max_by_ws = (long)(ws_bytes / ((size_t)M * N * sizeof(float)));
        long s = want;
        if (s > max_by_k) s = max_by_k;
        if (s > max_by_ws) s = max_by_ws;
        if (s > 16) s = 16;
        if (s >= 2) S = (int)s;
    }
    g.S = S;
    g.chunks_per_split = (chunks + S - 1) / S;
    g.S = (chunks + g.chunks_per_split - 1) / g.chunks_per_split;       // drop empty tail splits
    (void)narrow;
    return launch_gemm<1, 1, 4, BF>(g, st);
}

extern "C" int air_gemm(int ta, int tb, int M, int N, int K, const float *A, int lda, const float *B, int ldb,
                        float *C, int ldc, const float *bias, int epilogue, const float *aux, int ldaux, float beta,
                        float *colsum, void *ws, size_t ws_bytes, void *stream) {
    return gemm_dispatch<false>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, epilogue, aux, ldaux, beta, colsum, ws,
                                ws_bytes, stream);
}
extern "C" int air_gemm_bf16(int ta, int tb, int M, int N, int K, const float *A, int lda, const float *B, int ldb,
                             float *C, int ldc, const float *bias, int epilogue, const float *aux, int ldaux, float beta,
                             float *colsum, void *ws, size_t ws_bytes, void *stream) {
    return gemm_dispatch<true>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, epilogue, aux, ldaux, beta, colsum, ws,
                               ws_bytes, stream);
}

static int fill_gemm_args(GemmArgs &g, const AirGemmDesc &d) {
    AIR_REQUIRE(d.A && d.B && d.C, AIR_E_NULL);
    AIR_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, AIR_E_SHAPE);
    AIR_REQUIRE(d.lda >= (d.ta ? d.M : d.K) && d.ldb >= (d.tb ? d.K : d.N) && d.ldc >= d.N, AIR_E_SHAPE);
    AIR_REQUIRE(d.epilogue >= AIR_EPI_NONE && d.epilogue <= AIR_EPI_ADD_AUX_ELU, AIR_E_UNSUPPORTED);
    if (d.epilogue == AIR_EPI_BIAS || d.epilogue == AIR_EPI_BIAS_ELU) AIR_REQUIRE(d.bias, AIR_E_NULL);
    if (d.epilogue >= AIR_EPI_MUL_DELU) AIR_REQUIRE(d.aux && d.ldaux >= d.N, AIR_E_NULL);
    AIR_REQUIRE(!d.colsum || d.ta, AIR_E_UNSUPPORTED);
    if (d.A2) {       // the A prologue: k-contiguous A, both slabs and the bias readable with the same 16-byte loads, K % 16 == 0
        AIR_REQUIRE(!d.ta && d.a_bias && (d.K % 16 == 0) && (d.lda % 4 == 0), AIR_E_UNSUPPORTED);
        AIR_REQUIRE(air_aligned16(d.A) && air_aligned16(d.A2) && air_aligned16(d.a_bias) && (!d.a_out || air_aligned16(d.a_out)),
                    AIR_E_ALIGN);
    }
    g.A = d.A; g.B = d.B; g.C = d.C; g.bias = d.bias; g.aux = d.aux; g.colsum = d.colsum; g.ws = nullptr;
    g.M = d.M; g.N = d.N; g.K = d.K; g.lda = d.lda; g.ldb = d.ldb; g.ldc = d.ldc; g.ldaux = d.ldaux;
    g.ta = d.ta ? 1 : 0; g.tb = d.tb ? 1 : 0; g.epi = d.epilogue; g.beta = d.beta;
    g.vecA = (!d.ta && (d.lda % 4 == 0) && air_aligned16(d.A)) ? 1 : 0;
    g.vecB = (d.tb && (d.ldb % 4 == 0) && air_aligned16(d.B)) ? 1 : 0;
    g.S = 1; g.chunks_per_split = (d.K + 15) / 16;
    return AIR_OK;
}

// more than AIR_GEMM_GROUP_MAX problems: only the all-TN wide-tile form (the deferred weight gradients of a step)
static int launch_big_tn_group(const AirGemmDesc *descs, int count, void *stream) {
    AIR_REQUIRE(count <= AIR_GEMM_BIG_GROUP_MAX, AIR_E_SHAPE);
    BigGroupArgs ga;
    const bool bf = descs[0].precision == AIR_PREC_BF16;
    int wt = 0, min_k = 1 << 30, n_wide = 0;
    ga.small_mask = 0u;
    for (int i = 0; i < count; ++i) {
        const AirGemmDesc &d = descs[i];
        AIR_REQUIRE(d.precision == AIR_PREC_F32 || d.precision == AIR_PREC_BF16, AIR_E_UNSUPPORTED);
        AIR_REQUIRE((d.precision == AIR_PREC_BF16) == bf, AIR_E_UNSUPPORTED);
        AIR_REQUIRE(!d.A2, AIR_E_UNSUPPORTED);
        const bool wide = d.ta && !d.tb && air_aligned16(d.B) && d.ldb % 4 == 0 && d.K % 4 == 0 && d.M >= 4 && d.N >= 4 &&
                          d.M % 4 == 0 && d.N % 4 == 0;
        int st = fill_gemm_args(ga.g[i], d);
        if (st) return st;
        ga.tile_start[i] = wt;
        if (wide) {
            wt += air_cdiv(d.M, 64) * air_cdiv(d.N, 64);
            if (d.K < min_k) min_k = d.K;
            ++n_wide;
        } else {
            ga.small_mask |= 1u << i;
            wt += air_cdiv(d.M, 16) * air_cdiv(d.N, 16);
        }
    }
    AIR_REQUIRE(n_wide > 0, AIR_E_UNSUPPORTED);            // this form exists for the weight gradients of a step
    for (int i = count; i <= AIR_GEMM_BIG_GROUP_MAX; ++i) ga.tile_start[i] = wt;
    for (int i = count; i < AIR_GEMM_BIG_GROUP_MAX; ++i) ga.g[i] = ga.g[0];
    ga.count = count;
    // XCD-contiguous tiles: within each problem in fp32 (a map over the whole grid would hand every long-K problem -- they come
    // first -- to the first XCDs); over the whole grid with bf16 operands (launches of equal-K problems, L2-traffic bound)
    static const int big_xcd = getenv("AIR_GEMM_BIG_XCD") ? atoi(getenv("AIR_GEMM_BIG_XCD")) : 1;
    ga.xcd_map = (big_xcd && min_k >= 1024) ? (bf ? 2 : 1) : 0;
    hipStream_t st = air_stream(stream);
    if (bf) {
        static const int use16 = getenv("AIR_GEMM_BF16_STORAGE") ? atoi(getenv("AIR_GEMM_BF16_STORAGE")) : 1;
        BigGroupArgs16 g16;
        for (int i = 0; i < AIR_GEMM_BIG_GROUP_MAX; ++i) {
            const int j = i < count ? i : 0;
            const AirGemmDesc &d = descs[j];
            g16.g[i] = ga.g[j];
            g16.h[i].A16 = (use16 && d.A16 && d.lda % 2 == 0 && ((uintptr_t)d.A16 % 4 == 0)) ? d.A16 : nullptr;
            g16.h[i].B16 = (use16 && d.B16 && d.ldb % 2 == 0 && ((uintptr_t)d.B16 % 4 == 0)) ? d.B16 : nullptr;
            g16.h[i].C16 = nullptr;
        }
        for (int i = 0; i <= AIR_GEMM_BIG_GROUP_MAX; ++i) g16.tile_start[i] = ga.tile_start[i];
        g16.count = count; g16.xcd_map = ga.xcd_map; g16.small_mask = ga.small_mask;
        hipLaunchKernelGGL((gemm_big_group_wide16_tn_kernel<4, 8>), dim3(wt), dim3(512), 0, st, g16);
    } else hipLaunchKernelGGL((gemm_big_group_wide_tn_kernel<4, 8, false>), dim3(wt), dim3(512), 0, st, ga);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// a latency-regime group with short-K weight gradients among its problems: those on the streaming body, the rest on the 4-wave
// tile body (16x16 tiles, 32x32 once the REST holds more than 1536 of them), one launch
static unsigned shortk_mask(const AirGemmDesc *descs, int count) {
    unsigned m = 0;
    for (int i = 0; i < count; ++i) {
        if (descs[i].A2 || descs[i].C16 || descs[i].precision != AIR_PREC_F32) return 0;     // (the plain fp32 group only)
        if (shortk_eligible(descs[i])) m |= 1u << i;
    }
    return m;
}
static int launch_grouped_sk(const AirGemmDesc *descs, int count, unsigned sk_mask, void *stream) {
    GroupArgs ga;
    long tiles16 = 0;
    for (int i = 0; i < count; ++i)
        if (!((sk_mask >> i) & 1u)) tiles16 += (long)air_cdiv(descs[i].M, 16) * air_cdiv(descs[i].N, 16);
    const int T_ = tiles16 > 1536 ? 32 : 16;
    int tiles = 0;
    for (int i = 0; i < count; ++i) {
        int st = fill_gemm_args(ga.g[i], descs[i]);
        if (st) return st;
        ga.tile_start[i] = tiles;
        tiles += ((sk_mask >> i) & 1u) ? shortk_workgroups(descs[i].M) : air_cdiv(descs[i].M, T_) * air_cdiv(descs[i].N, T_);
    }
    for (int i = count; i <= AIR_GEMM_GROUP_MAX; ++i) ga.tile_start[i] = tiles;
    for (int i = count; i < AIR_GEMM_GROUP_MAX; ++i) ga.g[i] = ga.g[0];
    ga.count = count; ga.xcd_map = 0; ga.sk_mask = sk_mask;
    hipStream_t st = air_stream(stream);
    if (T_ == 16) hipLaunchKernelGGL((gemm_grouped_sk_kernel<1, 1>), dim3(tiles), dim3(256), 0, st, ga);
    else hipLaunchKernelGGL((gemm_grouped_sk_kernel<2, 2>), dim3(tiles), dim3(256), 0, st, ga);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// air_gemm_grouped for the first product(s) of a train step whose input batch is drawn from an HBM-resident dataset: every problem's A
// operand lies inside the observation buffer g->obs [B, item_floats] (row 0 + a column offset, lda = item_floats) and is read from the
// dataset through the feeder's index instead; the problems of g->copy_mask write what they read into g->obs (together they must
// cover every column once); g->idx_out receives the indices.  16x16 tiles, 16 waves splitting K (the latency-regime form of these
// long-K products).  Returns AIR_E_UNSUPPORTED for anything else: the caller then plans air_batch_gather + air_gemm_grouped.
extern "C" int air_gemm_grouped_gather_fits(const AirGemmDesc *descs, int count, const AirBatchGather *g) {
    if (!descs || !g || count < 1 || count > AIR_GEMM_GROUP_MAX) return 0;
    if (!g->dataset || !g->obs || !g->seed_dev || !g->step_dev || g->n_items <= 0 || g->item_floats <= 0 || g->B <= 0) return 0;
    if (g->item_floats % 4 || !air_aligned16(g->dataset) || !air_aligned16(g->obs)) return 0;
    long tiles16 = 0;
    for (int i = 0; i < count; ++i) {
        const AirGemmDesc &d = descs[i];
        if (d.ta || d.A2 || d.C16 || d.precision != AIR_PREC_F32 || d.M != g->B || d.lda != g->item_floats) return 0;
        const long off = (const float *)d.A - (const float *)g->obs;
        if (off < 0 || off + d.K > g->item_floats || off % 4) return 0;
        if (d.K < 512) return 0;
        tiles16 += (long)air_cdiv(d.M, 16) * air_cdiv(d.N, 16);
    }
    return tiles16 <= 1024 ? 1 : 0;
}
extern "C" int air_gemm_grouped_gather(const AirGemmDesc *descs, int count, const AirBatchGather *g, void *stream) {
    AIR_REQUIRE(descs && g, AIR_E_NULL);
    AIR_REQUIRE(air_gemm_grouped_gather_fits(descs, count, g) == 1, AIR_E_UNSUPPORTED);
    GroupArgs ga;
    int tiles = 0;
    for (int i = 0; i < count; ++i) {
        int st = fill_gemm_args(ga.g[i], descs[i]);
        if (st) return st;
        ga.g[i].vecA = 1;                                  // (dataset rows and column offsets are 16-byte aligned: checked above)
        ga.tile_start[i] = tiles;
        tiles += air_cdiv(descs[i].M, 16) * air_cdiv(descs[i].N, 16);
    }
    for (int i = count; i <= AIR_GEMM_GROUP_MAX; ++i) ga.tile_start[i] = tiles;
    for (int i = count; i < AIR_GEMM_GROUP_MAX; ++i) ga.g[i] = ga.g[0];
    ga.count = count; ga.xcd_map = 0; ga.sk_mask = 0;
    const GatherArgs gt = {g->dataset, g->obs, g->obs, g->n_items, g->item_floats, g->shuffle ? 1 : 0, g->B, g->seed_dev, g->step_dev,
                           g->idx_out, g->copy_mask};
    hipLaunchKernelGGL(gemm_grouped_gather_kernel, dim3(tiles), dim3(1024), 0, air_stream(stream), ga, gt);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

extern "C" int air_gemm_grouped(const AirGemmDesc *descs, int count, void *stream) {
    AIR_REQUIRE(descs, AIR_E_NULL);
    if (count > AIR_GEMM_GROUP_MAX) return launch_big_tn_group(descs, count, stream);
    AIR_REQUIRE(count > 0 && count <= AIR_GEMM_GROUP_MAX, AIR_E_SHAPE);
    if (const unsigned skm = shortk_mask(descs, count)) return launch_grouped_sk(descs, count, skm, stream);
    GroupArgs ga;
    // tile shape for the whole group: 16x16 tiles (more, shorter-lived workgroups) while the group is far from filling
    // the chip, 32x32 tiles once it holds thousands of them (less operand re-read, fewer workgroup rounds)
    long tiles16 = 0;
    for (int i = 0; i < count; ++i) tiles16 += (long)air_cdiv(descs[i].M, 16) * air_cdiv(descs[i].N, 16);
    const int T_ = tiles16 > 1536 ? 32 : 16;
    int tiles = 0;
    for (int i = 0; i < count; ++i) {
        int st = fill_gemm_args(ga.g[i], descs[i]);
        if (st) return st;
        ga.tile_start[i] = tiles;
        tiles += air_cdiv(descs[i].M, T_) * air_cdiv(descs[i].N, T_);
    }
    for (int i = count; i <= AIR_GEMM_GROUP_MAX; ++i) ga.tile_start[i] = tiles;
    for (int i = count; i < AIR_GEMM_GROUP_MAX; ++i) ga.g[i] = ga.g[0];
    ga.count = count;
    ga.xcd_map = 0;
    ga.sk_mask = 0;
    // long K on a handful of tiles (the BPTT products, 64x256x1024): 16 waves split K inside the workgroup, so every wave
    // still needs only one or two memory round trips and no second (split-K epilogue) launch is paid
    bool long_k = tiles16 <= 1024;
    for (int i = 0; i < count; ++i) long_k = long_k && descs[i].K >= 512 && descs[i].K >= 8 * (descs[i].M < descs[i].N ? descs[i].M : descs[i].N);
    const bool bf = descs[0].precision == AIR_PREC_BF16;
    for (int i = 0; i < count; ++i) {
        AIR_REQUIRE(descs[i].precision == AIR_PREC_F32 || descs[i].precision == AIR_PREC_BF16, AIR_E_UNSUPPORTED);
        AIR_REQUIRE((descs[i].precision == AIR_PREC_BF16) == bf, AIR_E_UNSUPPORTED);   // one precision per launch
    }
    hipStream_t st = air_stream(stream);
    // throughput regime: the wide-tile kernels (every operand load 16 bytes per lane) when the whole group has one operand
    // layout and every problem meets the alignment the interleaved loads need
    {
        static const long wide_min = getenv("AIR_GEMM_WIDE_MIN_TILES") ? atol(getenv("AIR_GEMM_WIDE_MIN_TILES")) : 1000;
        static const long wide_tn_bf = getenv("AIR_GEMM_WIDE_TN_BF16") ? atol(getenv("AIR_GEMM_WIDE_TN_BF16")) : 48;
        static const long wide_tn_f32 = getenv("AIR_GEMM_WIDE_TN_F32") ? atol(getenv("AIR_GEMM_WIDE_TN_F32")) : 48;
        static const long wide_nt_k = getenv("AIR_GEMM_WIDE_NT_K") ? atol(getenv("AIR_GEMM_WIDE_NT_K")) : 512;
        bool ok = tiles16 > wide_min;
        const int ta = descs[0].ta ? 1 : 0, tb = descs[0].tb ? 1 : 0;
        long tiles64 = 0;
        int min_k = 1 << 30;
        for (int i = 0; i < count && ok; ++i) {
            const AirGemmDesc &d = descs[i];
            ok = ok && (d.ta ? 1 : 0) == ta && (d.tb ? 1 : 0) == tb && !(ta && tb) && !d.A2;
            // 16-byte operand loads need not be 16-byte aligned (any leading dimension of A; K of any length when A is
            // k-contiguous); an interleaved operand must hold whole groups of 4 rows / columns; B as the weights are laid out
            ok = ok && air_aligned16(d.B) && d.ldb % 4 == 0 && (!ta || d.K % 4 == 0) && (!tb || d.K % 4 == 0);
            ok = ok && d.M >= 4 && d.N >= 4 && d.K >= 4 && (!ta || d.M % 4 == 0) && (tb || d.N % 4 == 0);
            // (the unaligned / odd-K forms only where the launch is far into the throughput regime: around a thousand tiles
            //  the 16x16-tile kernel is the better one for them -- K = 50 leaves half of the 8 K-splitting waves idle)
            if (tiles16 <= 2048) ok = ok && air_aligned16(d.A) && d.lda % 4 == 0 && d.K % 4 == 0;
            tiles64 += (long)air_cdiv(d.M, 64) * air_cdiv(d.N, 64);
            if (d.K < min_k) min_k = d.K;
        }
        if (ok && ta) ok = tiles64 >= (bf ? wide_tn_bf : wide_tn_f32) && min_k >= 256;     // (K = rows: short at small batch)
        // dX products (both operands k-contiguous): the 16x64 tile pays from K = 512; below that a 32x64 tile is 10-20 % faster than
        // the 32x32-tile kernel with bf16 operands (L1 operand traffic) and no faster in fp32 (profiles/r02_j_kbench_gemm_*.txt)
        const bool nt_short = !ta && tb && min_k < wide_nt_k;
        if (ok && nt_short) ok = bf && tiles16 > 2048;
        if (ok && bf && ta) return launch_big_tn_group(descs, count, stream);   // per-problem operand kinds (mirrors) in one launch
        if (ok) {
            const int TMw = ta ? 64 : (nt_short ? 32 : 16);
            int wt = 0;
            for (int i = 0; i < count; ++i) {
                ga.tile_start[i] = wt;
                wt += air_cdiv(descs[i].M, TMw) * air_cdiv(descs[i].N, 64);
            }
            for (int i = count; i <= AIR_GEMM_GROUP_MAX; ++i) ga.tile_start[i] = wt;
            // long-K weight-gradient groups (1.5 MB of operands per tile): measured +1.3 % on the batch-1024 step with bf16
            // operands, nothing in fp32 (MFMA issue bound), -1.5 % at batch 256 (K = 768)
            ga.xcd_map = (ta && min_k >= 1024) ? 1 : 0;
            if (bf) {
                // the bf16 data path: operands from their bf16 mirrors where EVERY problem of the launch has one (4-byte
                // aligned rows), from the fp32 buffers otherwise; mirrors of C written by the epilogue
                static const int use16 = getenv("AIR_GEMM_BF16_STORAGE") ? atoi(getenv("AIR_GEMM_BF16_STORAGE")) : 1;
                bool a16 = use16 != 0, b16 = use16 != 0;
                for (int i = 0; i < count; ++i) {
                    a16 = a16 && descs[i].A16 && descs[i].lda % 2 == 0 && ((uintptr_t)descs[i].A16 % 4 == 0);
                    b16 = b16 && descs[i].B16 && descs[i].ldb % 2 == 0 && ((uintptr_t)descs[i].B16 % 4 == 0);
                }
                GroupArgs16 g16;
                for (int i = 0; i < AIR_GEMM_GROUP_MAX; ++i) {
                    const int j = i < count ? i : 0;
                    g16.g[i] = ga.g[j];
                    g16.h[i].A16 = a16 ? descs[j].A16 : nullptr;
                    g16.h[i].B16 = b16 ? descs[j].B16 : nullptr;
                    g16.h[i].C16 = descs[j].C16;
                }
                for (int i = 0; i <= AIR_GEMM_GROUP_MAX; ++i) g16.tile_start[i] = ga.tile_start[i];
                g16.count = count; g16.xcd_map = ga.xcd_map;
#define AIR_WIDE16_LAUNCH2(MT_, TA_, TB_, A16_, B16_)                                                                              \
                do {                                                                                                       \
                    if (count == 1) hipLaunchKernelGGL((gemm_wide16_kernel<MT_, 8, TA_, TB_, A16_, B16_>), dim3(wt), dim3(512), 0, st, g16.g[0], g16.h[0]); \
                    else hipLaunchKernelGGL((gemm_grouped_wide16_kernel<MT_, 8, TA_, TB_, A16_, B16_>), dim3(wt), dim3(512), 0, st, g16);                   \
                } while (0)
#define AIR_WIDE16_LAUNCH(MT_, TA_, TB_)                                                                                   \
                do {                                                                                                       \
                    if (a16 && b16) AIR_WIDE16_LAUNCH2(MT_, TA_, TB_, true, true);                                         \
                    else if (a16) AIR_WIDE16_LAUNCH2(MT_, TA_, TB_, true, false);                                          \
                    else if (b16) AIR_WIDE16_LAUNCH2(MT_, TA_, TB_, false, true);                                          \
                    else AIR_WIDE16_LAUNCH2(MT_, TA_, TB_, false, false);                                                  \
                } while (0)
                if (ta) AIR_WIDE16_LAUNCH(4, true, false);
                else if (nt_short) AIR_WIDE16_LAUNCH(2, false, true);
                else if (tb) AIR_WIDE16_LAUNCH(1, false, true);
                else AIR_WIDE16_LAUNCH(1, false, false);
#undef AIR_WIDE16_LAUNCH
#undef AIR_WIDE16_LAUNCH2
                AIR_LAUNCH_CHECK();
                return AIR_OK;
            }
#define AIR_WIDE_LAUNCH(MT_, TA_, TB_)                                                                                     \
            do {                                                                                                           \
                if (count == 1) hipLaunchKernelGGL((gemm_wide_kernel<MT_, 8, false, TA_, TB_>), dim3(wt), dim3(512), 0, st, ga.g[0]);     \
                else hipLaunchKernelGGL((gemm_grouped_wide_kernel<MT_, 8, false, TA_, TB_>), dim3(wt), dim3(512), 0, st, ga);             \
            } while (0)
            if (ta) AIR_WIDE_LAUNCH(4, true, false);
            else if (nt_short) AIR_WIDE_LAUNCH(2, false, true);
            else if (tb) AIR_WIDE_LAUNCH(1, false, true);
            else AIR_WIDE_LAUNCH(1, false, false);
#undef AIR_WIDE_LAUNCH
            AIR_LAUNCH_CHECK();
            return AIR_OK;
        }
    }
#define AIR_GROUP_LAUNCH(MT_, NT_, KW_, NTH_)                                                                        \
    do {                                                                                                             \
        if (bf) hipLaunchKernelGGL((gemm_grouped_kernel<MT_, NT_, KW_, true>), dim3(tiles), dim3(NTH_), 0, st, ga);   \
        else hipLaunchKernelGGL((gemm_grouped_kernel<MT_, NT_, KW_, false>), dim3(tiles), dim3(NTH_), 0, st, ga);     \
    } while (0)
    // a lone problem goes through the single-GEMM kernel: its descriptor sits directly in the kernarg SGPRs, so the
    // tile_start lookup (one more dependent scalar-memory round trip before the first operand load) is skipped
#define AIR_SINGLE_LAUNCH(MT_, NT_, KW_, NTH_)                                                                       \
    do {                                                                                                             \
        if (bf) hipLaunchKernelGGL((gemm_f32_mfma_kernel<MT_, NT_, KW_, true>), dim3(tiles, 1), dim3(NTH_), 0, st, ga.g[0]);  \
        else hipLaunchKernelGGL((gemm_f32_mfma_kernel<MT_, NT_, KW_, false>), dim3(tiles, 1), dim3(NTH_), 0, st, ga.g[0]);    \
    } while (0)
    for (int i = 0; i < count; ++i) AIR_REQUIRE(!descs[i].A2 || count == 1, AIR_E_UNSUPPORTED);
    bool any_c16 = false;
    for (int i = 0; i < count; ++i) any_c16 = any_c16 || (descs[i].C16 != nullptr);
    if (any_c16) {
        // bf16 data path on small tiles: fp32 operand fetch, rounded in registers; the epilogue also writes the bf16 mirror of C
        AIR_REQUIRE(bf && !descs[0].A2, AIR_E_UNSUPPORTED);
        C16Ptrs c16;
        for (int i = 0; i < AIR_GEMM_GROUP_MAX; ++i) c16.p[i] = i < count ? descs[i].C16 : nullptr;
#define AIR_C16_LAUNCH(MT_, NT_, KW_, NTH_)                                                                          \
        do {                                                                                                         \
            if (count == 1) hipLaunchKernelGGL((gemm_bf16_c16_kernel<MT_, NT_, KW_>), dim3(tiles, 1), dim3(NTH_), 0, st, ga.g[0], c16.p[0]);  \
            else hipLaunchKernelGGL((gemm_grouped_c16_kernel<MT_, NT_, KW_>), dim3(tiles), dim3(NTH_), 0, st, ga, c16);                     \
        } while (0)
        if (long_k) AIR_C16_LAUNCH(1, 1, 16, 1024);
        else if (T_ == 16) AIR_C16_LAUNCH(1, 1, 4, 256);
        else AIR_C16_LAUNCH(2, 2, 4, 256);
#undef AIR_C16_LAUNCH
        AIR_LAUNCH_CHECK();
        return AIR_OK;
    }
    if (count == 1 && descs[0].A2) {
        // latency-regime consumer of a K-split producer; a long K (the first hidden layer >= 512 wide at a small batch) stays
        // on the 4-wave prologue kernel: correct for any K, the 16-wave K split has no prologue form
        AIR_REQUIRE(T_ == 16, AIR_E_UNSUPPORTED);
        const AproArgs pro = {descs[0].A2, descs[0].a_bias, descs[0].a_out, descs[0].a_elu};
        if (bf) hipLaunchKernelGGL((gemm_f32_mfma_apro_kernel<1, 1, 4, true>), dim3(tiles, 1), dim3(256), 0, st, ga.g[0], pro);
        else hipLaunchKernelGGL((gemm_f32_mfma_apro_kernel<1, 1, 4, false>), dim3(tiles, 1), dim3(256), 0, st, ga.g[0], pro);
    } else if (count == 1) {
        if (long_k) AIR_SINGLE_LAUNCH(1, 1, 16, 1024);
        else if (T_ == 16) AIR_SINGLE_LAUNCH(1, 1, 4, 256);
        else AIR_SINGLE_LAUNCH(2, 2, 4, 256);
    } else if (long_k) AIR_GROUP_LAUNCH(1, 1, 16, 1024);
    else if (T_ == 16) AIR_GROUP_LAUNCH(1, 1, 4, 256);
    else {
        // (whole-tile waves instead of a 4-way K split for short contractions -- K = batch in the weight gradients of the wide
        //  first layers -- were measured in round 3 and removed in round 4: 0.2131-0.2223 against 0.2077 ms at configs[1])
        AIR_GROUP_LAUNCH(2, 2, 4, 256);
    }
#undef AIR_SINGLE_LAUNCH
#undef AIR_GROUP_LAUNCH
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// air_gemm_grouped with the backward of a Gaussian head (loc_mode 0) in the epilogue of problem `gb->problem` + NVIL / KL-share riders
// (include/air_hip.h).  Latency-regime tile kernels only.
extern "C" int air_gemm_grouped_gauss_bwd(const AirGemmDesc *descs, int count, const AirGaussBwdEpi *e, const float *imp_parts,
                                          int n_parts, float *imp_sum, const float *baseline, const float *logp, float *nvil_out,
                                          float *dlogp, float *dbaseline, int B, float *ema_dev, const float *kl_parts,
                                          int n_kl_parts, float *kl_row_out, int kl_rows, void *stream) {
    AIR_REQUIRE(descs && e, AIR_E_NULL);
    AIR_REQUIRE(count > 0 && count <= AIR_GEMM_GROUP_MAX && e->problem >= 0 && e->problem < count, AIR_E_SHAPE);
    AIR_REQUIRE(e->pre && e->eps && e->loc && e->scale && e->dpre && e->D > 0 && e->ld_pre >= 2 * e->D && e->ld_dpre >= 2 * e->D, AIR_E_NULL);
    AIR_REQUIRE(!imp_parts || (baseline && logp && nvil_out && B > 0 && n_parts > 0), AIR_E_NULL);
    AIR_REQUIRE(n_kl_parts >= 0 && (n_kl_parts == 0 || (kl_parts && kl_row_out && kl_rows > 0)), AIR_E_NULL);
    {
        const AirGemmDesc &d = descs[e->problem];          // the product that forms dsample[M, D]
        AIR_REQUIRE(d.N == e->D && d.beta == 0.f && d.epilogue == AIR_EPI_NONE && !d.colsum, AIR_E_UNSUPPORTED);
    }
    GroupArgs ga;
    long tiles16 = 0;
    for (int i = 0; i < count; ++i) tiles16 += (long)air_cdiv(descs[i].M, 16) * air_cdiv(descs[i].N, 16);
    AIR_REQUIRE(tiles16 <= 1000, AIR_E_UNSUPPORTED);         // (beyond: the 32x32 / wide-tile dispatch of air_gemm_grouped -- no fold there)
    int tiles = 0;
    const bool bf = descs[0].precision == AIR_PREC_BF16;
    bool long_k = true;
    for (int i = 0; i < count; ++i) {
        const AirGemmDesc &d = descs[i];
        AIR_REQUIRE(d.precision == AIR_PREC_F32 || d.precision == AIR_PREC_BF16, AIR_E_UNSUPPORTED);
        AIR_REQUIRE((d.precision == AIR_PREC_BF16) == bf && !d.A2 && !d.C16, AIR_E_UNSUPPORTED);
        int st = fill_gemm_args(ga.g[i], d);
        if (st) return st;
        ga.tile_start[i] = tiles;
        tiles += air_cdiv(d.M, 16) * air_cdiv(d.N, 16);
        long_k = long_k && d.K >= 512 && d.K >= 8 * (d.M < d.N ? d.M : d.N);
    }
    for (int i = count; i <= AIR_GEMM_GROUP_MAX; ++i) ga.tile_start[i] = tiles;
    for (int i = count; i < AIR_GEMM_GROUP_MAX; ++i) ga.g[i] = ga.g[0];
    ga.count = count;
    ga.xcd_map = 0;
    ga.sk_mask = 0;
    GaussEpi gb;
    gb.pre = e->pre; gb.eps = e->eps; gb.loc = e->loc; gb.scale = e->scale; gb.dkl_row = e->dkl_row; gb.dpre = e->dpre;
    gb.ld_pre = e->ld_pre; gb.ld_dpre = e->ld_dpre; gb.D = e->D; gb.raw_offset = e->raw_offset; gb.pl = e->p_loc; gb.ps = e->p_scale;
    gb.dkl_scale = e->dkl_scale; gb.guard = e->guard_eps; gb.mask = 1u << e->problem;
    const NvilArgs nv = {imp_parts, baseline, logp, nvil_out, dlogp, dbaseline, B, n_parts > 0 ? n_parts : 1, imp_sum, ema_dev};
    const KlParts kp = {kl_parts, kl_row_out, n_kl_parts};
    const int riders = (imp_parts ? 1 : 0) + (n_kl_parts > 0 ? 1 : 0);
    // (rider 0 is NVIL when there is one; the KL-share sum is the other -- or the only -- rider)
    hipStream_t st = air_stream(stream);
#define AIR_GB_LAUNCH(KW_, NTH_)                                                                                                   \
    do {                                                                                                                           \
        if (bf) hipLaunchKernelGGL((gemm_grouped_gb_kernel<1, 1, KW_, true>), dim3(tiles + riders), dim3(NTH_), 0, st, ga, gb, nv, kp, tiles, kl_rows);  \
        else hipLaunchKernelGGL((gemm_grouped_gb_kernel<1, 1, KW_, false>), dim3(tiles + riders), dim3(NTH_), 0, st, ga, gb, nv, kp, tiles, kl_rows);   \
    } while (0)
    if (long_k) AIR_GB_LAUNCH(16, 1024); else AIR_GB_LAUNCH(4, 256);
#undef AIR_GB_LAUNCH
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// air_gemm_grouped with the optimiser folded in (include/air_hip.h): latency-regime tile kernels only (what the closing launch of
// the batch-64 train step runs on); a group the wide-tile dispatch would take is declined (AIR_E_UNSUPPORTED) -- the caller keeps
// the plain launch + air_step_epilogue there.
extern "C" int air_gemm_grouped_opt(const AirGemmDesc *descs, int count, const AirOptFold *o, void *stream) {
    AIR_REQUIRE(descs && o, AIR_E_NULL);
    AIR_REQUIRE(count > 0 && count <= AIR_GEMM_GROUP_MAX, AIR_E_SHAPE);
    AIR_REQUIRE(o->p && o->g && o->ms && o->mg && o->mom && o->lr_dev, AIR_E_NULL);
    AIR_REQUIRE(o->n_ranges >= 0 && o->n_ranges <= AIR_OPT_MAX_RANGES && o->n_model % 4 == 0, AIR_E_SHAPE);
    AIR_REQUIRE(air_aligned16(o->p) && air_aligned16(o->g) && air_aligned16(o->ms) && air_aligned16(o->mg) && air_aligned16(o->mom), AIR_E_ALIGN);
    GroupArgs ga;
    // short-K weight gradients on the streaming body (their range = their workgroups).  A folded launch takes them when EVERY problem
    // is one (the engine's closing launch); a folded launch that mixes one with tile problems is declined -- the caller keeps the
    // unfolded launch + the closing update for it, so the product is the streaming body's in either plan
    const unsigned sk_mask = shortk_mask(descs, count);
    AIR_REQUIRE(sk_mask == 0 || sk_mask == (1u << count) - 1u, AIR_E_UNSUPPORTED);
    long tiles16 = 0;
    for (int i = 0; i < count; ++i)
        if (!((sk_mask >> i) & 1u)) tiles16 += (long)air_cdiv(descs[i].M, 16) * air_cdiv(descs[i].N, 16);
    const int T_ = tiles16 > 1536 ? 32 : 16;
    int tiles = 0;
    const bool bf = descs[0].precision == AIR_PREC_BF16;
    bool long_k = tiles16 <= 1024 && sk_mask == 0;
    for (int i = 0; i < count; ++i) {
        const AirGemmDesc &d = descs[i];
        AIR_REQUIRE(d.precision == AIR_PREC_F32 || d.precision == AIR_PREC_BF16, AIR_E_UNSUPPORTED);
        AIR_REQUIRE((d.precision == AIR_PREC_BF16) == bf && !d.A2 && !d.C16, AIR_E_UNSUPPORTED);
        int st = fill_gemm_args(ga.g[i], d);
        if (st) return st;
        ga.tile_start[i] = tiles;
        tiles += ((sk_mask >> i) & 1u) ? shortk_workgroups(d.M) : air_cdiv(d.M, T_) * air_cdiv(d.N, T_);
        long_k = long_k && d.K >= 512 && d.K >= 8 * (d.M < d.N ? d.M : d.N);
        if ((o->fold_mask >> i) & 1u) {
            // a folded problem: a plain weight gradient written into the flat gradient buffer (its parameter sits at the same offset)
            AIR_REQUIRE(d.beta == 0.f && d.epilogue == AIR_EPI_NONE, AIR_E_UNSUPPORTED);
            AIR_REQUIRE(d.C >= o->g && (!d.colsum || d.colsum >= o->g), AIR_E_SHAPE);
        }
    }
    {   // the wide-tile regime has its own kernels (and its own fold, air_gemm_grouped's deferred-gradient launch): not here
        static const long wide_min = getenv("AIR_GEMM_WIDE_MIN_TILES") ? atol(getenv("AIR_GEMM_WIDE_MIN_TILES")) : 1000;
        int min_k = 1 << 30;
        bool all_tn = true;
        for (int i = 0; i < count; ++i) { all_tn = all_tn && descs[i].ta && !descs[i].tb; if (descs[i].K < min_k) min_k = descs[i].K; }
        AIR_REQUIRE(!(tiles16 > wide_min && all_tn && min_k >= 256), AIR_E_UNSUPPORTED);
    }
    for (int i = count; i <= AIR_GEMM_GROUP_MAX; ++i) ga.tile_start[i] = tiles;
    for (int i = count; i < AIR_GEMM_GROUP_MAX; ++i) ga.g[i] = ga.g[0];
    ga.count = count;
    ga.xcd_map = 0;
    ga.sk_mask = 0;
    OptFold f;
    f.p = o->p; f.g0 = o->g; f.ms = o->ms; f.mg = o->mg; f.mom = o->mom; f.n_model = o->n_model; f.lr_dev = o->lr_dev;
    f.lr_mult_tail = o->lr_mult_tail; f.decay = o->decay; f.momentum = o->momentum; f.eps = o->eps; f.gscale = o->grad_scale;
    f.fold_mask = o->fold_mask & ((1u << count) - 1u);
    f.n_ranges = o->n_ranges; f.tiles = tiles;
    size_t nq = 0;
    for (int r = 0; r < AIR_OPT_MAX_RANGES; ++r) {
        f.lo[r] = r < o->n_ranges ? o->range_lo[r] : 0; f.hi[r] = r < o->n_ranges ? o->range_hi[r] : 0;
        AIR_REQUIRE(f.lo[r] <= f.hi[r] && f.lo[r] % 4 == 0 && f.hi[r] % 4 == 0, AIR_E_SHAPE);
        nq += (f.hi[r] - f.lo[r]) >> 2;
    }
    f.gstep = o->global_step_dev; f.rng_state = o->rng_state_dev; f.rng_inc = o->rng_increment;
    const int nth = long_k ? 1024 : 256;
    size_t extra = air_rider_blocks(nq, nth, 512);                          // about two float4 per rider thread
    if (extra < 1) extra = 1;                                               // (the counters live in rider workgroup 0)
    hipStream_t st = air_stream(stream);
#define AIR_GROUP_OPT_LAUNCH(MT_, NT_, KW_)                                                                                                   \
    do {                                                                                                                                      \
        if (bf) hipLaunchKernelGGL((gemm_grouped_opt_kernel<MT_, NT_, KW_, true>), dim3(tiles + (int)extra), dim3(nth), 0, st, ga, f);         \
        else hipLaunchKernelGGL((gemm_grouped_opt_kernel<MT_, NT_, KW_, false>), dim3(tiles + (int)extra), dim3(nth), 0, st, ga, f);           \
    } while (0)
    ga.sk_mask = sk_mask;
    if (sk_mask) hipLaunchKernelGGL(gemm_grouped_opt_sk_kernel, dim3(tiles + (int)extra), dim3(256), 0, st, ga, f);
    else if (long_k) AIR_GROUP_OPT_LAUNCH(1, 1, 16);
    else if (T_ == 16) AIR_GROUP_OPT_LAUNCH(1, 1, 4);
    else AIR_GROUP_OPT_LAUNCH(2, 2, 4);
#undef AIR_GROUP_OPT_LAUNCH
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// ---- LSTM recurrence with the gate math fused into the GEMM (snt.LSTM, mnist_model.py:35 / cell.py:126-127) -----------
// The recurrence is the one truly sequential part of the step (T dependent products that cannot be batched over time),
// and at batch 64 each link of that chain costs a launch (~4.5 us) far more than its flops.  Fusing the gate
// non-linearities into the product halves the chain: forward T launches instead of 2T, backward T instead of 2T+1.
//
// One 16x16 accumulator tile per workgroup, KW waves interleave the 16-deep K chunks (as gemm_body<1,1,KW>); A is always
// k-contiguous, B is k-strided (forward: W_h[K=Hd, 4Hd]) or k-contiguous (backward: W_h read as [N=Hd, K=4Hd]).
template <int KW, bool BF, bool B_KCONTIG>
__device__ __forceinline__ void tile16_kloop(f32x4 (&acc)[1][1], gcf gA, int lda, int rowA, bool okA, bool vecA, gcf gB,
                                             int ldb, int colB, bool okB, bool vecB, int K, int limA, int limB) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane >> 4;
    constexpr int U = 4;
    const int full_end = K >> 4;
    const int rowAc = okA ? rowA : limA - 1, colBc = okB ? colB : limB;
#pragma nounroll
    for (int c = wave; c < full_end; c += U * KW) {
        f32x4 fa[U][1], fb[U][1];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cu = c + u * KW;
            if (cu < full_end) {
                const int k = (cu << 4) + 4 * lg;
                fa[u][0] = ld_kcontig_full(gA, lda, rowAc, k, vecA);
                fb[u][0] = B_KCONTIG ? ld_kcontig_full(gB, ldb, colBc, k, vecB) : ld_kstrided_full(gB, ldb, colBc, k);
            } else {
                fa[u][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                fb[u][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c + u * KW >= full_end) break;
            mfma_chunk<1, 1, BF>(acc, fa[u], fb[u]);
        }
    }
    const int pc = K >> 4;
    if ((K & 15) && (pc % KW) == wave) {
        const int k = (pc << 4) + 4 * lg;
        f32x4 fa[1], fb[1];
        fa[0] = ld_kcontig(gA, lda, rowA, okA, k, K, vecA);
        fb[0] = B_KCONTIG ? ld_kcontig(gB, ldb, colB, okB, k, K, vecB) : ld_kstrided(gB, ldb, colB, okB, k, K);
        mfma_chunk<1, 1, BF>(acc, fa, fb);
    }
}

// Two independent contractions of one tile position with the loads of BOTH issued before the first MFMA (the first LSTM step with
// the folded input product: x . W_x and h0 . W_h): one memory round trip instead of two per group of U chunks.  Chunk c of segment s
// goes to the wave tile16_kloop gives it to, and each segment accumulates its chunks in tile16_kloop's order: same bits.
template <int KW, bool BF>
__device__ __forceinline__ void tile16_kloop2(f32x4 (&acc0)[1][1], f32x4 (&acc1)[1][1], gcf gA0, int lda0, bool vecA0, gcf gB0, int K0,
                                              gcf gA1, int lda1, bool vecA1, gcf gB1, int K1, int ldb, int rowA, bool okA, int colB,
                                              bool okB, int limA, int limB) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane >> 4;
    constexpr int U = 4;
    const int full0 = K0 >> 4, full1 = K1 >> 4;
    const int rowAc = okA ? rowA : limA - 1, colBc = okB ? colB : limB;
    const int fmax = full0 > full1 ? full0 : full1;
#pragma nounroll
    for (int c = wave; c < fmax; c += U * KW) {
        f32x4 fa0[U][1], fb0[U][1], fa1[U][1], fb1[U][1];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cu = c + u * KW;
            const int k = (cu << 4) + 4 * lg;
            if (cu < full0) { fa0[u][0] = ld_kcontig_full(gA0, lda0, rowAc, k, vecA0); fb0[u][0] = ld_kstrided_full(gB0, ldb, colBc, k); }
            else { fa0[u][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; fb0[u][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            if (cu < full1) { fa1[u][0] = ld_kcontig_full(gA1, lda1, rowAc, k, vecA1); fb1[u][0] = ld_kstrided_full(gB1, ldb, colBc, k); }
            else { fa1[u][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; fb1[u][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c + u * KW < full0) mfma_chunk<1, 1, BF>(acc0, fa0[u], fb0[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c + u * KW < full1) mfma_chunk<1, 1, BF>(acc1, fa1[u], fb1[u]);
        }
    }
    const int pc0 = K0 >> 4, pc1 = K1 >> 4;
    if ((K0 & 15) && (pc0 % KW) == wave) {
        const int k = (pc0 << 4) + 4 * lg;
        f32x4 fa[1], fb[1];
        fa[0] = ld_kcontig(gA0, lda0, rowA, okA, k, K0, vecA0);
        fb[0] = ld_kstrided(gB0, ldb, colB, okB, k, K0);
        mfma_chunk<1, 1, BF>(acc0, fa, fb);
    }
    if ((K1 & 15) && (pc1 % KW) == wave) {
        const int k = (pc1 << 4) + 4 * lg;
        f32x4 fa[1], fb[1];
        fa[0] = ld_kcontig(gA1, lda1, rowA, okA, k, K1, vecA1);
        fb[0] = ld_kstrided(gB1, ldb, colB, okB, k, K1);
        mfma_chunk<1, 1, BF>(acc1, fa, fb);
    }
}

struct LstmFwdArgs {
    const float *h_prev, *w_h, *gx, *c_prev;
    float *h, *c, *gate_act;
    int M, Hd, ldw, ldgx, vecA;
    int ldh, ldc;            // row strides of h_prev / c_prev: Hd, or 0 = one row broadcast over the batch (trainable h0, c0)
    int tiles;               // workgroups [tiles, gridDim.x) run the step prologue instead (first step of a train step)
    float fb;
};
// tile = 16 batch rows x 4 hidden units: its 16 accumulator columns are the i,j,f,o gates of those 4 units (column
// 4*gate + unit  <->  W_h column gate*Hd + unit), so the gate math of a unit never leaves the workgroup
template <bool BF>
__global__ __launch_bounds__(256) void lstm_fwd_fused_kernel(LstmFwdArgs g, PrologueArgs pro) {
    constexpr int KW = 4, LDT = 20;
    __shared__ float s_tile[KW][16 * LDT];
    if ((int)blockIdx.x >= g.tiles) {       // independent role: noise / prior / tiled initial state for the rest of the step
        step_prologue_body(pro, (int)blockIdx.x - g.tiles, (int)gridDim.x - g.tiles);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int tiles_u = (g.Hd + 3) >> 2;
    const int tm = blockIdx.x / tiles_u, tu = blockIdx.x - tm * tiles_u;
    const int m0 = tm * 16, u0 = tu * 4;
    const int ub = u0 + (li & 3);
    const bool okB = ub < g.Hd;
    const int colB = (li >> 2) * g.Hd + ub;
    const int rowA = m0 + li;
    const bool okA = rowA < g.M;
    // epilogue operands of thread (r, uu): fetched before the K loop so their round trip overlaps the operand loads
    const int er = threadIdx.x >> 2, eu = u0 + (threadIdx.x & 3), em = m0 + er;
    const bool e_ok = threadIdx.x < 64 && em < g.M && eu < g.Hd;
    float e_gx[4] = {0.f, 0.f, 0.f, 0.f}, e_c = 0.f;
    if (e_ok) {
        const gcf gx = (gcf)g.gx + (size_t)em * g.ldgx + eu;
#pragma unroll
        for (int q = 0; q < 4; ++q) e_gx[q] = gx[(size_t)q * g.Hd];
        e_c = ((gcf)g.c_prev)[(size_t)em * g.ldc + eu];
    }
    f32x4 acc[1][1];
    acc[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    tile16_kloop<KW, BF, false>(acc, (gcf)g.h_prev, g.ldh, rowA, okA, g.vecA != 0, (gcf)g.w_h, g.ldw, colB, okB, false, g.Hd,
                                g.M, (li >> 2) * g.Hd + g.Hd - 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) s_tile[wave][(4 * lg + r) * LDT + li] = acc[0][0][r];
    __syncthreads();
    if (e_ok) {
        float pre[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int off = er * LDT + 4 * q + (threadIdx.x & 3);
            pre[q] = ((s_tile[0][off] + s_tile[1][off]) + (s_tile[2][off] + s_tile[3][off])) + e_gx[q];
        }
        const float gi = sigmoid_acc(pre[0]);
        const float gj = tanhf(pre[1]);
        const float gf = sigmoid_acc(pre[2] + g.fb);
        const float go = sigmoid_acc(pre[3]);
        const float cn = gf * e_c + gi * gj;
        const size_t e = (size_t)em * g.Hd + eu;
        ((gf_t)g.c)[e] = cn;
        ((gf_t)g.h)[e] = tanhf(cn) * go;
        const gf_t ar = (gf_t)g.gate_act + (size_t)em * 4 * g.Hd + eu;
        ar[0] = gi; ar[g.Hd] = gj; ar[2 * (size_t)g.Hd] = gf; ar[3 * (size_t)g.Hd] = go;
    }
}

// The FIRST step of the unroll with the hoisted input product folded in (round 5: one dependent launch fewer on the forward chain).
// gx = x . W_x + b does not depend on t (the image never changes, cell.py:121-125), so it used to be a launch of its own in front of
// the recurrence; but step 0's recurrent operand is the trainable initial state -- ONE row for the whole batch -- so its product
// h0 . W_h costs nothing to add here: the tile accumulates BOTH contractions (x[M,E] . W_x[E,4Hd] and h0[1,Hd] . W_h[Hd,4Hd]) into
// two accumulators, writes gx = (x . W_x) + b for the later steps and finishes step 0's gate math on gx + h0 . W_h -- the same
// sums in the same order as the two launches it replaces (bit-identical h_1, c_1, gate_act_0, gx).
struct LstmFirstArgs { const float *x, *w_x, *b; float *gx_out; int E, ldx, vecX; };
template <bool BF>
__global__ __launch_bounds__(256) void lstm_fwd_first_kernel(LstmFwdArgs g, LstmFirstArgs f, PrologueArgs pro) {
    constexpr int KW = 4, LDT = 20;
    __shared__ float s_x[KW][16 * LDT], s_h[KW][16 * LDT];
    if ((int)blockIdx.x >= g.tiles) {
        step_prologue_body(pro, (int)blockIdx.x - g.tiles, (int)gridDim.x - g.tiles);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int tiles_u = (g.Hd + 3) >> 2;
    const int tm = blockIdx.x / tiles_u, tu = blockIdx.x - tm * tiles_u;
    const int m0 = tm * 16, u0 = tu * 4;
    const int ub = u0 + (li & 3);
    const bool okB = ub < g.Hd;
    const int colB = (li >> 2) * g.Hd + ub;
    const int rowA = m0 + li;
    const bool okA = rowA < g.M;
    const int er = threadIdx.x >> 2, eu = u0 + (threadIdx.x & 3), em = m0 + er;
    const bool e_ok = threadIdx.x < 64 && em < g.M && eu < g.Hd;
    float e_b[4] = {0.f, 0.f, 0.f, 0.f}, e_c = 0.f;
    if (e_ok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) e_b[q] = ((gcf)f.b)[(size_t)q * g.Hd + eu];
        e_c = ((gcf)g.c_prev)[(size_t)em * g.ldc + eu];
    }
    f32x4 ax[1][1], ah[1][1];
    ax[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ah[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int limB = (li >> 2) * g.Hd + g.Hd - 1;
    tile16_kloop2<KW, BF>(ax, ah, (gcf)f.x, f.ldx, f.vecX != 0, (gcf)f.w_x, f.E, (gcf)g.h_prev, g.ldh, g.vecA != 0, (gcf)g.w_h, g.Hd,
                          g.ldw, rowA, okA, colB, okB, g.M, limB);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        s_x[wave][(4 * lg + r) * LDT + li] = ax[0][0][r];
        s_h[wave][(4 * lg + r) * LDT + li] = ah[0][0][r];
    }
    __syncthreads();
    if (e_ok) {
        float pre[4];
        const gf_t gxo = (gf_t)f.gx_out + (size_t)em * g.ldgx + eu;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int off = er * LDT + 4 * q + (threadIdx.x & 3);
            const float gxv = ((s_x[0][off] + s_x[1][off]) + (s_x[2][off] + s_x[3][off])) + e_b[q];      // the BIAS epilogue of the gx product
            gxo[(size_t)q * g.Hd] = gxv;
            pre[q] = ((s_h[0][off] + s_h[1][off]) + (s_h[2][off] + s_h[3][off])) + gxv;
        }
        const float gi = sigmoid_acc(pre[0]);
        const float gj = tanhf(pre[1]);
        const float gf = sigmoid_acc(pre[2] + g.fb);
        const float go = sigmoid_acc(pre[3]);
        const float cn = gf * e_c + gi * gj;
        const size_t e = (size_t)em * g.Hd + eu;
        ((gf_t)g.c)[e] = cn;
        ((gf_t)g.h)[e] = tanhf(cn) * go;
        const gf_t ar = (gf_t)g.gate_act + (size_t)em * 4 * g.Hd + eu;
        ar[0] = gi; ar[g.Hd] = gj; ar[2 * (size_t)g.Hd] = gf; ar[3 * (size_t)g.Hd] = go;
    }
}

// ---- the `what` head in ONE launch (round 5): q = x . W + b (modules.py:20-21), what ~ N(loc, softplus(raw + offset)) with its KL
// terms (cell.py:154-156, model.py:174-186) and the latent columns of the baseline input (modules.py:131-139) -- three things the
// step used to spend two dependent launches on (the product, then air_what_sample_pack).  A tile is 16 rows x 8 LATENT DIMS: its 16
// accumulator columns are the loc pre-activations of those dims AND their raw scales (gathered W columns a and A + a, the trick of the
// fused LSTM step), so after the in-workgroup K reduction a thread holds both halves of its (row, dim) and samples right there.  The
// KL row sum spans the ceil(A / 8) tiles of a row: each tile writes its 8-dim share to kl_parts[tile][M] (summed in the tile by three
// lane exchanges) and a later launch of the step adds the shares in tile order (air_gauss_sample_bwd*: kl_parts / kl_row_out).
struct WhatHeadArgs {
    const float *x, *w, *b, *eps;
    float *q, *loc, *scale, *what, *kl_parts, *pack;
    const float *where, *presence, *s0, *s1;
    int M, K, A, ldx, vecX, T, B, S0, S1, tiles;
    float raw_offset, pl, ps, guard;
};
template <bool BF>
__global__ __launch_bounds__(256) void what_head_kernel(WhatHeadArgs g) {
    constexpr int KW = 4, LDT = 20;
    __shared__ float s_tile[KW][16 * LDT];
    const int width = g.T * g.A + g.T * 4 + g.T + g.S0 + g.S1;
    if ((int)blockIdx.x >= g.tiles) {       // independent role: the where / presence / state columns of the baseline input
        baseline_pack_body((int)blockIdx.x - g.tiles, (int)gridDim.x - g.tiles, nullptr, g.what, g.where, g.presence, g.s0, g.s1,
                           g.pack, g.T, g.B, 0, g.A, g.S0, g.S1, g.T * g.A);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int tiles_n = (g.A + 7) >> 3;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * 16, a0 = tn * 8;
    const int ab = a0 + (li & 7);
    const bool okB = ab < g.A;
    const int colB = (li >> 3) * g.A + ab;
    const int rowA = m0 + li;
    const bool okA = rowA < g.M;
    const int er = threadIdx.x >> 3, ed = threadIdx.x & 7, em = m0 + er, ea = a0 + ed;
    const bool e_ok = threadIdx.x < 128 && em < g.M && ea < g.A;
    float e_bl = 0.f, e_br = 0.f, e_eps = 0.f;
    if (e_ok) {
        e_bl = ((gcf)g.b)[ea]; e_br = ((gcf)g.b)[g.A + ea];
        e_eps = ((gcf)g.eps)[(size_t)em * g.A + ea];
    }
    f32x4 acc[1][1];
    acc[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    tile16_kloop<KW, BF, false>(acc, (gcf)g.x, g.ldx, rowA, okA, g.vecX != 0, (gcf)g.w, 2 * g.A, colB, okB, false, g.K, g.M, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) s_tile[wave][(4 * lg + r) * LDT + li] = acc[0][0][r];
    __syncthreads();
    if (threadIdx.x < 128) {
        float kl = 0.f;
        if (e_ok) {
            const int o0 = er * LDT + ed, o1 = o0 + 8;
            const float locp = ((s_tile[0][o0] + s_tile[1][o0]) + (s_tile[2][o0] + s_tile[3][o0])) + e_bl;   // the BIAS epilogue
            const float raw = ((s_tile[0][o1] + s_tile[1][o1]) + (s_tile[2][o1] + s_tile[3][o1])) + e_br;
            const gf_t qr = (gf_t)g.q + (size_t)em * 2 * g.A;
            qr[ea] = locp; qr[g.A + ea] = raw;
            const float s = guard_scale(softplus_acc(raw + g.raw_offset), g.guard);
            const float v = locp + s * e_eps;
            const size_t o = (size_t)em * g.A + ea;
            ((gf_t)g.loc)[o] = locp; ((gf_t)g.scale)[o] = s; ((gf_t)g.what)[o] = v;
            const int t = em / g.B, bb = em - t * g.B;
            ((gf_t)g.pack)[(size_t)bb * width + t * g.A + ea] = v;
            kl = normal_kl(locp, s, g.pl, g.ps);
        }
        kl += __shfl_xor(kl, 1, 64);
        kl += __shfl_xor(kl, 2, 64);
        kl += __shfl_xor(kl, 4, 64);
        if (ed == 0 && em < g.M) ((gf_t)g.kl_parts)[(size_t)tn * g.M + em] = kl;
    }
}
extern "C" int air_what_head_parts(int A) { return (A + 7) / 8; }
extern "C" int air_what_head_fwd(const float *x, int ldx, int K, const float *w, const float *b, const float *eps, float raw_offset,
                                 float p_loc, float p_scale, float *q, float *loc, float *scale, float *sample, float *kl_parts,
                                 int A, const float *where, const float *presence, const float *state0, const float *state1,
                                 float *pack_out, int T, int B, int S0, int S1, float guard_eps, int precision, void *stream) {
    AIR_REQUIRE(x && w && b && eps && q && loc && scale && sample && kl_parts && where && presence && pack_out, AIR_E_NULL);
    AIR_REQUIRE((S0 == 0 || state0) && (S1 == 0 || state1), AIR_E_NULL);
    AIR_REQUIRE(T > 0 && B > 0 && A > 0 && K > 0 && ldx >= K && S0 >= 0 && S1 >= 0, AIR_E_SHAPE);
    AIR_REQUIRE(precision == AIR_PREC_F32 || precision == AIR_PREC_BF16, AIR_E_UNSUPPORTED);
    WhatHeadArgs g;
    g.x = x; g.w = w; g.b = b; g.eps = eps; g.q = q; g.loc = loc; g.scale = scale; g.what = sample; g.kl_parts = kl_parts; g.pack = pack_out;
    g.where = where; g.presence = presence; g.s0 = state0; g.s1 = state1;
    g.M = T * B; g.K = K; g.A = A; g.ldx = ldx; g.vecX = ((ldx % 4) == 0 && air_aligned16(x)) ? 1 : 0;
    g.T = T; g.B = B; g.S0 = S0; g.S1 = S1;
    g.tiles = air_cdiv(g.M, 16) * air_cdiv(A, 8);
    g.raw_offset = raw_offset; g.pl = p_loc; g.ps = p_scale; g.guard = guard_eps;
    const size_t n_pack = (size_t)B * (T * 5 + S0 + S1);
    int pb = (int)((n_pack + PW_THREADS - 1) / PW_THREADS);
    if (pb > 256) pb = 256;
    if (pb < 1) pb = 1;
    if (precision == AIR_PREC_BF16) hipLaunchKernelGGL((what_head_kernel<true>), dim3(g.tiles + pb), dim3(256), 0, air_stream(stream), g);
    else hipLaunchKernelGGL((what_head_kernel<false>), dim3(g.tiles + pb), dim3(256), 0, air_stream(stream), g);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// ---- throughput regime (more than 512 tiles of 16 x 16): the same fusion on the wide-tile scheme ---------------------------
// Workgroup tile = 16 batch rows x 64 hidden units x the 4 gates: SIXTEEN 16-wide MFMA tiles per wave, tile (q, c) = gate q,
// units u0 + 4 i + c (i = 0..15).  W_h[k, q*Hd + u] is k-strided; per k step a lane issues FOUR 16-byte loads (one per gate,
// along the contiguous unit dimension) and every loaded value feeds a different tile -- no dword loads, no permuted copy of
// w_gates.  8 waves split K (Hd = 256: two 16-deep chunks each, all loads of a wave in ONE round trip) and reduce through LDS in
// a fixed order; the epilogue thread of (row, unit) then holds all four gates: `gates` never exists, one launch per step instead
// of a product and a pointwise pass.
template <bool BF>
__global__ __launch_bounds__(512) void lstm_fwd_wide_kernel(LstmFwdArgs g, PrologueArgs pro) {
    constexpr int KW = 8, LDT = 256 + 4;
    __shared__ float s_tile[KW][16 * LDT];                  // local column = gate * 64 + unit
    if ((int)blockIdx.x >= g.tiles) {
        if (threadIdx.x < PW_THREADS) step_prologue_body(pro, (int)blockIdx.x - g.tiles, (int)gridDim.x - g.tiles);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int tiles_u = g.Hd >> 6;
    const int tm = blockIdx.x / tiles_u, tu = blockIdx.x - tm * tiles_u;
    const int m0 = tm * 16, u0 = tu * 64;
    const gcf gA = (gcf)g.h_prev, gW = (gcf)g.w_h;
    int rowA = m0 + li; if (rowA > g.M - 1) rowA = g.M - 1;
    const size_t offA = (size_t)rowA * g.ldh;
    const int colb = u0 + 4 * li;
    // epilogue operands of this thread's two (row, unit) pairs, requested before the K loop
    float e_gx[2][4], e_c[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = threadIdx.x + 512 * i, r = e >> 6, u = e & 63;
        int m = m0 + r; if (m > g.M - 1) m = g.M - 1;
        const gcf gx = (gcf)g.gx + (size_t)m * g.ldgx + u0 + u;
#pragma unroll
        for (int q = 0; q < 4; ++q) e_gx[i][q] = gx[(size_t)q * g.Hd];
        e_c[i] = ((gcf)g.c_prev)[(size_t)m * g.ldc + u0 + u];
    }
    f32x4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int U = 2;
    const int nchunks = g.Hd >> 4;
#pragma nounroll
    for (int c = wave; c < nchunks; c += U * KW) {
        f32x4 fa[U], fw[U][4][4];                           // fw[u][j][q] = W_h[k + j, q*Hd + colb .. colb+3]
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int cu = c + u * KW; if (cu > nchunks - 1) cu = nchunks - 1;
            const int k = (cu << 4) + 4 * lg;
            fa[u] = *(gcf4)(gA + offA + k);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) fw[u][j][q] = *(gcf4)(gW + (size_t)(k + j) * g.ldw + (size_t)q * g.Hd + colb);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c + u * KW >= nchunks) break;
            if (BF) {
                const s16x4 ha = to_bf16x4(fa[u]);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const s16x4 hb = to_bf16x4((f32x4){fw[u][0][q][cc], fw[u][1][q][cc], fw[u][2][q][cc], fw[u][3][q][cc]});
                        acc[q * 4 + cc] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ha, hb, acc[q * 4 + cc], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc)
                            acc[q * 4 + cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][j], fw[u][j][q][cc], acc[q * 4 + cc], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *(f32x4 *)&s_tile[wave][(4 * lg + r) * LDT + q * 64 + 4 * li] =
                (f32x4){acc[q * 4 + 0][r], acc[q * 4 + 1][r], acc[q * 4 + 2][r], acc[q * 4 + 3][r]};
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = threadIdx.x + 512 * i, r = e >> 6, u = e & 63;
        const int m = m0 + r;
        float pre[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int off = r * LDT + q * 64 + u;
            float v = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < KW; w4 += 4)
                v += (s_tile[w4][off] + s_tile[w4 + 1][off]) + (s_tile[w4 + 2][off] + s_tile[w4 + 3][off]);
            pre[q] = v + e_gx[i][q];
        }
        if (m < g.M) {
            const float gi = sigmoid_acc(pre[0]);
            const float gj = tanhf(pre[1]);
            const float gf = sigmoid_acc(pre[2] + g.fb);
            const float go = sigmoid_acc(pre[3]);
            const float cn = gf * e_c[i] + gi * gj;
            const size_t eo = (size_t)m * g.Hd + u0 + u;
            ((gf_t)g.c)[eo] = cn;
            ((gf_t)g.h)[eo] = tanhf(cn) * go;
            const gf_t ar = (gf_t)g.gate_act + (size_t)m * 4 * g.Hd + u0 + u;
            ar[0] = gi; ar[g.Hd] = gj; ar[2 * (size_t)g.Hd] = gf; ar[3 * (size_t)g.Hd] = go;
        }
    }
}

struct LstmBwdArgs {
    const float *dgates_next, *w_h, *dh_a, *dh_b, *dc_in, *gate_act, *c_prev, *c, *dgx_in;
    float *dgates, *dc_prev, *dgx_out;
    int M, Hd, vecA, vecB;
};
// dh[m,u] = sum_k dgates_{t+1}[m,k] W_h[u,k]  (+ the direct dh terms of step t), then the pointwise backward of step t for
// that (m,u): dgates_t (4 values), dc_{t-1}, and the running sum over time of dgates (what x.W_x receives) -- all
// element-wise in (m,u), so the 16x16 output tile finishes everything it owns
// (opt: an optimiser slice on the workgroups past the tiles -- a separate kernel argument, untouched by the tile workgroups)
template <int KW, bool BF>
__global__ __launch_bounds__(64 * KW) void lstm_bwd_fused_kernel(LstmBwdArgs g, RmspropSlice opt) {
    constexpr int LDT = 20;
    __shared__ float s_tile[KW][16 * LDT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int tiles_n = (g.Hd + 15) >> 4;
    {
        const int tiles = ((g.M + 15) >> 4) * tiles_n;
        if ((int)blockIdx.x >= tiles) {
            rmsprop_slice_body(opt, (int)blockIdx.x - tiles, (int)gridDim.x - tiles);
            return;
        }
    }
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * 16, n0 = tn * 16;
    const int rowA = m0 + li, colB = n0 + li;
    const bool okA = rowA < g.M, okB = colB < g.Hd;
    const int K = 4 * g.Hd;
    const int er = threadIdx.x >> 4, ec = threadIdx.x & 15, em = m0 + er, eu = n0 + ec;
    const bool e_ok = threadIdx.x < 256 && em < g.M && eu < g.Hd;
    float gi = 0.f, gj = 0.f, gff = 0.f, go = 0.f, cp = 0.f, cc = 0.f, dha = 0.f, dhb = 0.f, dci = 0.f, sx[4] = {0.f, 0.f, 0.f, 0.f};
    if (e_ok) {
        const size_t e = (size_t)em * g.Hd + eu;
        const gcf ar = (gcf)g.gate_act + (size_t)em * K + eu;
        gi = ar[0]; gj = ar[g.Hd]; gff = ar[2 * (size_t)g.Hd]; go = ar[3 * (size_t)g.Hd];
        cp = ((gcf)g.c_prev)[e];
        cc = ((gcf)g.c)[e];
        if (g.dh_a) dha = ((gcf)g.dh_a)[e];
        if (g.dh_b) dhb = ((gcf)g.dh_b)[e];
        if (g.dc_in) dci = ((gcf)g.dc_in)[e];
        if (g.dgx_in) {
            const gcf sr = (gcf)g.dgx_in + (size_t)em * K + eu;
#pragma unroll
            for (int q = 0; q < 4; ++q) sx[q] = sr[(size_t)q * g.Hd];
        }
    }
    f32x4 acc[1][1];
    acc[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    tile16_kloop<KW, BF, true>(acc, (gcf)g.dgates_next, K, rowA, okA, g.vecA != 0, (gcf)g.w_h, K, colB, okB, g.vecB != 0, K,
                               g.M, g.Hd - 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) s_tile[wave][(4 * lg + r) * LDT + li] = acc[0][0][r];
    __syncthreads();
    if (e_ok) {
        const int off = er * LDT + ec;
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < KW; q += 4)
            v += (s_tile[q][off] + s_tile[q + 1][off]) + (s_tile[q + 2][off] + s_tile[q + 3][off]);
        // same order as the unfused pair: the product accumulates ONTO the direct dh term (beta = 1), then + dh_b
        const float dh = (v + dha) + dhb;
        const float tc = tanhf(cc);
        const float dct = dci + dh * go * (1.f - tc * tc);
        float d[4];
        d[0] = dct * gj * gi * (1.f - gi);
        d[1] = dct * gi * (1.f - gj * gj);
        d[2] = dct * cp * gff * (1.f - gff);
        d[3] = dh * tc * go * (1.f - go);
        const gf_t dr = (gf_t)g.dgates + (size_t)em * K + eu;
#pragma unroll
        for (int q = 0; q < 4; ++q) dr[(size_t)q * g.Hd] = d[q];
        ((gf_t)g.dc_prev)[(size_t)em * g.Hd + eu] = dct * gff;
        if (g.dgx_out) {
            const gf_t so = (gf_t)g.dgx_out + (size_t)em * K + eu;
#pragma unroll
            for (int q = 0; q < 4; ++q) so[(size_t)q * g.Hd] = sx[q] + d[q];
        }
    }
}

// The ENTRY of the BPTT and its first link in ONE launch (latency regime, round 5): the pointwise backward of the last step T-1
// has no product in front of it -- it was a launch of its own (air_lstm_pointwise_bwd) whose only consumer is the link of step
// T-2.  Here every workgroup of that link forms its A operand dgates_{T-1}[16 rows, 4Hd] ON THE FLY from the saved gate activations,
// cell states and the two direct dh terms of step T-1 (wave w owns the unit chunks w, w+16, ...: the four gates of a unit chunk are
// four 16-deep chunks of K, so ONE set of loads yields four A fragments), multiplies it with W_h^T and finishes step T-2's gate
// backward exactly as lstm_bwd_fused_kernel does; the (row, unit) pairs of the tile's epilogue re-form their own step T-1 values for
// dc_in and the running sum over time.  The first column of tiles stores dgates_{T-1} / dc_{T-2 <- T-1} for the weight gradients.
// One element function for both places: same bits wherever it is evaluated.
struct LstmEntryArgs {
    const float *gate_act1, *c_prev1, *c1, *dh_a1, *dh_b1;     // step T-1 (dh_a1 / dh_b1 may be NULL)
    float *dgates1, *dc_prev1;
};
__device__ __forceinline__ void lstm_pw_bwd_elem(float gi, float gj, float gf, float go, float cp, float c, float dha, float dhb,
                                                 float dci, float (&d)[4], float &dc_prev) {
#pragma clang fp contract(off)
    // tanh through one v_exp_f32 and one v_rcp_f32 (absolute error ~1e-7; exact limits +-1): every workgroup of a row tile repeats
    // this for its 16 x Hd operand elements, and libm's tanhf is two thirds of that work (AIR_LSTM_ENTRY_TANHF: libm's, for A/B builds)
#ifdef AIR_LSTM_ENTRY_TANHF
    const float tc = tanhf(c);
#else
    const float tc = 1.f - __fdividef(2.f, __expf(2.f * c) + 1.f);
#endif
    const float dhe = dha + dhb;
    const float dct = dci + dhe * go * (1.f - tc * tc);
    d[0] = dct * gj * gi * (1.f - gi);
    d[1] = dct * gi * (1.f - gj * gj);
    d[2] = dct * cp * gf * (1.f - gf);
    d[3] = dhe * tc * go * (1.f - go);
    dc_prev = dct * gf;
}
__global__ __launch_bounds__(1024) void lstm_bwd_entry_kernel(LstmBwdArgs g, LstmEntryArgs en, RmspropSlice opt) {
    constexpr int KW = 16, LDT = 20;
    __shared__ float s_tile[KW][16 * LDT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int tiles_n = g.Hd >> 4;
    {
        const int tiles = ((g.M + 15) >> 4) * tiles_n;
        if ((int)blockIdx.x >= tiles) {
            rmsprop_slice_body(opt, (int)blockIdx.x - tiles, (int)gridDim.x - tiles);
            return;
        }
    }
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * 16, n0 = tn * 16;
    const int rowA = m0 + li, colB = n0 + li;                   // Hd % 16 == 0: every column of the tile exists
    const bool okA = rowA < g.M;
    const int rowAc = okA ? rowA : g.M - 1;
    const int Hd = g.Hd, K = 4 * Hd;
    // ---- epilogue operands of thread (er, ec): step T-2's, and step T-1's for dc_in and the running sum
    const int er = threadIdx.x >> 4, ec = threadIdx.x & 15, em = m0 + er, eu = n0 + ec;
    const bool e_ok = threadIdx.x < 256 && em < g.M;
    float gi = 0.f, gj = 0.f, gff = 0.f, go = 0.f, cp = 0.f, cc = 0.f, dha = 0.f, dhb = 0.f;
    float gi1 = 0.f, gj1 = 0.f, gf1 = 0.f, go1 = 0.f, cp1 = 0.f, cc1 = 0.f, dha1 = 0.f, dhb1 = 0.f;
    if (e_ok) {
        const size_t e = (size_t)em * Hd + eu;
        const gcf ar = (gcf)g.gate_act + (size_t)em * K + eu;
        gi = ar[0]; gj = ar[Hd]; gff = ar[2 * (size_t)Hd]; go = ar[3 * (size_t)Hd];
        cp = ((gcf)g.c_prev)[e];
        cc = ((gcf)g.c)[e];
        if (g.dh_a) dha = ((gcf)g.dh_a)[e];
        if (g.dh_b) dhb = ((gcf)g.dh_b)[e];
        const gcf a1 = (gcf)en.gate_act1 + (size_t)em * K + eu;
        gi1 = a1[0]; gj1 = a1[Hd]; gf1 = a1[2 * (size_t)Hd]; go1 = a1[3 * (size_t)Hd];
        cp1 = ((gcf)en.c_prev1)[e];
        cc1 = ((gcf)en.c1)[e];
        if (en.dh_a1) dha1 = ((gcf)en.dh_a1)[e];
        if (en.dh_b1) dhb1 = ((gcf)en.dh_b1)[e];
    }
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nuc = Hd >> 4;                                     // unit chunks (16 units each)
    const gcf gW = (gcf)g.w_h + (size_t)colB * K;
#pragma nounroll
    for (int uc = wave; uc < nuc; uc += KW) {
        const int u4 = (uc << 4) + 4 * lg;
        const size_t eo = (size_t)rowAc * Hd + u4, ao = (size_t)rowAc * K + u4;
        // all loads of the group first: 4 gate vectors, 2 cell states, 2 dh terms, 4 weight fragments
        f32x4 a_g[4], fb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a_g[q] = *(gcf4)((gcf)en.gate_act1 + ao + (size_t)q * Hd);
        const f32x4 v_cp = *(gcf4)((gcf)en.c_prev1 + eo), v_c = *(gcf4)((gcf)en.c1 + eo);
        f32x4 v_da = (f32x4){0.f, 0.f, 0.f, 0.f}, v_db = v_da;
        if (en.dh_a1) v_da = *(gcf4)((gcf)en.dh_a1 + eo);
        if (en.dh_b1) v_db = *(gcf4)((gcf)en.dh_b1 + eo);
#pragma unroll
        for (int q = 0; q < 4; ++q) fb[q] = *(gcf4)(gW + (size_t)q * Hd + u4);
        f32x4 fa[4], v_dcp;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            float d[4], dcp;
            lstm_pw_bwd_elem(a_g[0][x], a_g[1][x], a_g[2][x], a_g[3][x], v_cp[x], v_c[x], v_da[x], v_db[x], 0.f, d, dcp);
            fa[0][x] = d[0]; fa[1][x] = d[1]; fa[2][x] = d[2]; fa[3][x] = d[3];
            v_dcp[x] = dcp;
        }
        if (tn == 0 && okA) {                                    // the entry's own outputs, once per row tile
#pragma unroll
            for (int q = 0; q < 4; ++q) *(f32x4 *)(en.dgates1 + ao + (size_t)q * Hd) = fa[q];
            *(f32x4 *)(en.dc_prev1 + eo) = v_dcp;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[q][j], fb[q][j], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) s_tile[wave][(4 * lg + r) * LDT + li] = acc[r];
    __syncthreads();
    if (e_ok) {
        const int off = er * LDT + ec;
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < KW; q += 4)
            v += (s_tile[q][off] + s_tile[q + 1][off]) + (s_tile[q + 2][off] + s_tile[q + 3][off]);
        float d1[4], dci;
        lstm_pw_bwd_elem(gi1, gj1, gf1, go1, cp1, cc1, dha1, dhb1, 0.f, d1, dci);     // step T-1 at this (row, unit)
        float d[4], dcp;
        lstm_pw_bwd_elem(gi, gj, gff, go, cp, cc, v + dha, dhb, dci, d, dcp);          // dh = (product + dh_a) + dh_b, as the link
        const gf_t dr = (gf_t)g.dgates + (size_t)em * K + eu;
#pragma unroll
        for (int q = 0; q < 4; ++q) dr[(size_t)q * Hd] = d[q];
        ((gf_t)g.dc_prev)[(size_t)em * Hd + eu] = dcp;
        if (g.dgx_out) {
            const gf_t so = (gf_t)g.dgx_out + (size_t)em * K + eu;
#pragma unroll
            for (int q = 0; q < 4; ++q) so[(size_t)q * Hd] = d1[q] + d[q];
        }
    }
}

// One BPTT link in the throughput regime: dh = dgates_{t+1}[M, 4Hd] . W_h[Hd, 4Hd]^T on the wide-tile scheme (both operands
// k-contiguous: 16 rows x 64 units per workgroup, 8 waves split the 4Hd-deep contraction), then -- exactly as
// lstm_bwd_fused_kernel -- the pointwise backward of step t for the (row, unit) pairs the tile owns.
template <bool BF>
__global__ __launch_bounds__(512) void lstm_bwd_wide_kernel(LstmBwdArgs g, RmspropSlice opt) {
    constexpr int KW = 8, NT = 4, LDT = 64 + 4;
    __shared__ float s_tile[KW][16 * LDT];
    const int tiles_n = g.Hd >> 6;
    {
        const int tiles = ((g.M + 15) >> 4) * tiles_n;
        if ((int)blockIdx.x >= tiles) {
            rmsprop_slice_body(opt, (int)blockIdx.x - tiles, (int)gridDim.x - tiles);
            return;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * 16, n0 = tn * 64;
    const int K = 4 * g.Hd;
    const gcf gA = (gcf)g.dgates_next, gB = (gcf)g.w_h;
    int rowA = m0 + li; if (rowA > g.M - 1) rowA = g.M - 1;
    const size_t offA = (size_t)rowA * K;
    size_t offB[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) offB[b] = (size_t)(n0 + 16 * b + li) * K;
    // epilogue operands of this thread's two (row, unit) pairs
    float gi[2], gj[2], gff[2], go[2], cp[2], cc[2], dha[2], dhb[2], dci[2], sx[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e_ = threadIdx.x + 512 * i, r = e_ >> 6, u = n0 + (e_ & 63);
        int m = m0 + r; if (m > g.M - 1) m = g.M - 1;
        const size_t e = (size_t)m * g.Hd + u;
        const gcf ar = (gcf)g.gate_act + (size_t)m * K + u;
        gi[i] = ar[0]; gj[i] = ar[g.Hd]; gff[i] = ar[2 * (size_t)g.Hd]; go[i] = ar[3 * (size_t)g.Hd];
        cp[i] = ((gcf)g.c_prev)[e];
        cc[i] = ((gcf)g.c)[e];
        dha[i] = g.dh_a ? ((gcf)g.dh_a)[e] : 0.f;
        dhb[i] = g.dh_b ? ((gcf)g.dh_b)[e] : 0.f;
        dci[i] = g.dc_in ? ((gcf)g.dc_in)[e] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) sx[i][q] = g.dgx_in ? ((gcf)g.dgx_in)[(size_t)m * K + u + (size_t)q * g.Hd] : 0.f;
    }
    f32x4 acc[1][NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[0][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int U = 4;
    const int nchunks = K >> 4;
#pragma nounroll
    for (int c = wave; c < nchunks; c += U * KW) {
        f32x4 fa[U][1], fb[U][NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int cu = c + u * KW; if (cu > nchunks - 1) cu = nchunks - 1;
            const int k = (cu << 4) + 4 * lg;
            fa[u][0] = *(gcf4)(gA + offA + k);
#pragma unroll
            for (int b = 0; b < NT; ++b) fb[u][b] = *(gcf4)(gB + offB[b] + k);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c + u * KW >= nchunks) break;
            mfma_chunk<1, NT, BF>(acc, fa[u], fb[u]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int b = 0; b < NT; ++b) s_tile[wave][(4 * lg + r) * LDT + 16 * b + li] = acc[0][b][r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e_ = threadIdx.x + 512 * i, r = e_ >> 6, uc = e_ & 63, u = n0 + uc, m = m0 + r;
        const int off = r * LDT + uc;
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < KW; q += 4) v += (s_tile[q][off] + s_tile[q + 1][off]) + (s_tile[q + 2][off] + s_tile[q + 3][off]);
        if (m >= g.M) continue;
        const float dh = (v + dha[i]) + dhb[i];                // same order as the unfused pair (beta = 1 accumulate, then + dh_b)
        const float tc = tanhf(cc[i]);
        const float dct = dci[i] + dh * go[i] * (1.f - tc * tc);
        float d[4];
        d[0] = dct * gj[i] * gi[i] * (1.f - gi[i]);
        d[1] = dct * gi[i] * (1.f - gj[i] * gj[i]);
        d[2] = dct * cp[i] * gff[i] * (1.f - gff[i]);
        d[3] = dh * tc * go[i] * (1.f - go[i]);
        const gf_t dr = (gf_t)g.dgates + (size_t)m * K + u;
#pragma unroll
        for (int q = 0; q < 4; ++q) dr[(size_t)q * g.Hd] = d[q];
        ((gf_t)g.dc_prev)[(size_t)m * g.Hd + u] = dct * gff[i];
        if (g.dgx_out) {
            const gf_t so = (gf_t)g.dgx_out + (size_t)m * K + u;
#pragma unroll
            for (int q = 0; q < 4; ++q) so[(size_t)q * g.Hd] = sx[i][q] + d[q];
        }
    }
}

// ---- the LSTM recurrence on the bf16 data path (throughput regime) -----------------------------------------------------------
// lstm_fwd_wide_kernel / lstm_bwd_wide_kernel with the operands in memory as bf16: W_h from the bf16 shadow of the parameters
// (half the bytes of the larger operand), h_prev / dgates_{t+1} from their mirrors where one exists (the previous step's launch
// wrote it; the first step reads the fp32 tiled initial state), products on v_mfma_f32_16x16x32_bf16, and the epilogue writes
// the mirrors of h / dgates / running dgx next to the fp32 values.  Same tiles, same fixed-order K split over 8 waves.
struct Lstm16 { const void *w16, *a16; void *out16, *out16_b; };
__global__ __launch_bounds__(512) void lstm_fwd_wide16_kernel(LstmFwdArgs g, Lstm16 x) {
    constexpr int KW = 8, LDT = 256 + 4;
    __shared__ float s_tile[KW][16 * LDT];                  // local column = gate * 64 + unit
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int tiles_u = g.Hd >> 6;
    const int tm = blockIdx.x / tiles_u, tu = blockIdx.x - tm * tiles_u;
    const int m0 = tm * 16, u0 = tu * 64;
    const gcf gA = (gcf)g.h_prev;
    const gch hA = (gch)x.a16, hW = (gch)x.w16;
    int rowA = m0 + li; if (rowA > g.M - 1) rowA = g.M - 1;
    const size_t offA = (size_t)rowA * g.ldh;
    const int colb = u0 + 4 * li;
    float e_gx[2][4], e_c[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = threadIdx.x + 512 * i, r = e >> 6, u = e & 63;
        int m = m0 + r; if (m > g.M - 1) m = g.M - 1;
        const gcf gx = (gcf)g.gx + (size_t)m * g.ldgx + u0 + u;
#pragma unroll
        for (int q = 0; q < 4; ++q) e_gx[i][q] = gx[(size_t)q * g.Hd];
        e_c[i] = ((gcf)g.c_prev)[(size_t)m * g.ldc + u0 + u];
    }
    f32x4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nchunks = g.Hd >> 5;                          // 32-deep chunks
#pragma nounroll
    for (int c = wave; c < nchunks; c += KW) {
        const int k = (c << 5) + 8 * lg;
        u32x4 fa;
        if (hA) fa = *(gcu4)(hA + offA + k);
        else fa = pk8(*(gcf4)(gA + offA + k), *(gcf4)(gA + offA + k + 4));
        u32x2 w[4][8];                                      // w[q][j] = W_h16[k + j, q*Hd + colb .. colb+3]
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q][j] = *(gcu2)(hW + (size_t)(k + j) * g.ldw + (size_t)q * g.Hd + colb);
        const bf16x8 ha = __builtin_bit_cast(bf16x8, fa);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[q * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, __builtin_bit_cast(bf16x8, tr16<0>(w[q])), acc[q * 4 + 0], 0, 0, 0);
            acc[q * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, __builtin_bit_cast(bf16x8, tr16<1>(w[q])), acc[q * 4 + 1], 0, 0, 0);
            acc[q * 4 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, __builtin_bit_cast(bf16x8, tr16<2>(w[q])), acc[q * 4 + 2], 0, 0, 0);
            acc[q * 4 + 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, __builtin_bit_cast(bf16x8, tr16<3>(w[q])), acc[q * 4 + 3], 0, 0, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *(f32x4 *)&s_tile[wave][(4 * lg + r) * LDT + q * 64 + 4 * li] =
                (f32x4){acc[q * 4 + 0][r], acc[q * 4 + 1][r], acc[q * 4 + 2][r], acc[q * 4 + 3][r]};
    __syncthreads();
    const gh_t h16 = (gh_t)x.out16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = threadIdx.x + 512 * i, r = e >> 6, u = e & 63;
        const int m = m0 + r;
        float pre[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int off = r * LDT + q * 64 + u;
            float v = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < KW; w4 += 4)
                v += (s_tile[w4][off] + s_tile[w4 + 1][off]) + (s_tile[w4 + 2][off] + s_tile[w4 + 3][off]);
            pre[q] = v + e_gx[i][q];
        }
        if (m < g.M) {
            const float gi = sigmoid_acc(pre[0]);
            const float gj = tanhf(pre[1]);
            const float gf = sigmoid_acc(pre[2] + g.fb);
            const float go = sigmoid_acc(pre[3]);
            const float cn = gf * e_c[i] + gi * gj;
            const size_t eo = (size_t)m * g.Hd + u0 + u;
            const float hn = tanhf(cn) * go;
            ((gf_t)g.c)[eo] = cn;
            ((gf_t)g.h)[eo] = hn;
            if (h16) h16[eo] = bf16_bits(hn);
            const gf_t ar = (gf_t)g.gate_act + (size_t)m * 4 * g.Hd + u0 + u;
            ar[0] = gi; ar[g.Hd] = gj; ar[2 * (size_t)g.Hd] = gf; ar[3 * (size_t)g.Hd] = go;
        }
    }
}

__global__ __launch_bounds__(512) void lstm_bwd_wide16_kernel(LstmBwdArgs g, Lstm16 x) {
    constexpr int KW = 8, NT = 4, LDT = 64 + 4;
    __shared__ float s_tile[KW][16 * LDT];
    const int tiles_n = g.Hd >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * 16, n0 = tn * 64;
    const int K = 4 * g.Hd;
    const gcf gA = (gcf)g.dgates_next;
    const gch hA = (gch)x.a16, hB = (gch)x.w16;
    int rowA = m0 + li; if (rowA > g.M - 1) rowA = g.M - 1;
    const size_t offA = (size_t)rowA * K;
    size_t offB[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) offB[b] = (size_t)(n0 + 16 * b + li) * K;
    float gi[2], gj[2], gff[2], go[2], cp[2], cc[2], dha[2], dhb[2], dci[2], sx[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e_ = threadIdx.x + 512 * i, r = e_ >> 6, u = n0 + (e_ & 63);
        int m = m0 + r; if (m > g.M - 1) m = g.M - 1;
        const size_t e = (size_t)m * g.Hd + u;
        const gcf ar = (gcf)g.gate_act + (size_t)m * K + u;
        gi[i] = ar[0]; gj[i] = ar[g.Hd]; gff[i] = ar[2 * (size_t)g.Hd]; go[i] = ar[3 * (size_t)g.Hd];
        cp[i] = ((gcf)g.c_prev)[e];
        cc[i] = ((gcf)g.c)[e];
        dha[i] = g.dh_a ? ((gcf)g.dh_a)[e] : 0.f;
        dhb[i] = g.dh_b ? ((gcf)g.dh_b)[e] : 0.f;
        dci[i] = g.dc_in ? ((gcf)g.dc_in)[e] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) sx[i][q] = g.dgx_in ? ((gcf)g.dgx_in)[(size_t)m * K + u + (size_t)q * g.Hd] : 0.f;
    }
    f32x4 acc[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int U = 4;
    const int nchunks = K >> 5;
#pragma nounroll
    for (int c = wave; c < nchunks; c += U * KW) {
        u32x4 fa[U], fb[U][NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int cu = c + u * KW; if (cu > nchunks - 1) cu = nchunks - 1;
            const int k = (cu << 5) + 8 * lg;
            if (hA) fa[u] = *(gcu4)(hA + offA + k);
            else fa[u] = pk8(*(gcf4)(gA + offA + k), *(gcf4)(gA + offA + k + 4));
#pragma unroll
            for (int b = 0; b < NT; ++b) fb[u][b] = *(gcu4)(hB + offB[b] + k);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c + u * KW >= nchunks) break;
#pragma unroll
            for (int b = 0; b < NT; ++b)
                acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[u]), __builtin_bit_cast(bf16x8, fb[u][b]),
                                                                acc[b], 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int b = 0; b < NT; ++b) s_tile[wave][(4 * lg + r) * LDT + 16 * b + li] = acc[b][r];
    __syncthreads();
    const gh_t d16 = (gh_t)x.out16, s16 = (gh_t)x.out16_b;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e_ = threadIdx.x + 512 * i, r = e_ >> 6, uc = e_ & 63, u = n0 + uc, m = m0 + r;
        const int off = r * LDT + uc;
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < KW; q += 4) v += (s_tile[q][off] + s_tile[q + 1][off]) + (s_tile[q + 2][off] + s_tile[q + 3][off]);
        if (m >= g.M) continue;
        const float dh = (v + dha[i]) + dhb[i];
        const float tc = tanhf(cc[i]);
        const float dct = dci[i] + dh * go[i] * (1.f - tc * tc);
        float d[4];
        d[0] = dct * gj[i] * gi[i] * (1.f - gi[i]);
        d[1] = dct * gi[i] * (1.f - gj[i] * gj[i]);
        d[2] = dct * cp[i] * gff[i] * (1.f - gff[i]);
        d[3] = dh * tc * go[i] * (1.f - go[i]);
        const gf_t dr = (gf_t)g.dgates + (size_t)m * K + u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            dr[(size_t)q * g.Hd] = d[q];
            if (d16) d16[(size_t)m * K + u + (size_t)q * g.Hd] = bf16_bits(d[q]);
        }
        ((gf_t)g.dc_prev)[(size_t)m * g.Hd + u] = dct * gff[i];
        if (g.dgx_out) {
            const gf_t so = (gf_t)g.dgx_out + (size_t)m * K + u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float sv = sx[i][q] + d[q];
                so[(size_t)q * g.Hd] = sv;
                if (s16) s16[(size_t)m * K + u + (size_t)q * g.Hd] = bf16_bits(sv);
            }
        }
    }
}

extern "C" int air_lstm_step_fwd_bf16(const float *h_prev, const void *h_prev_bf16, const float *c_prev, const void *w_h_bf16,
                                      int ldw, const float *gx, int ldgx, float *h, void *h_bf16, float *c, float *gate_act,
                                      int M, int Hd, float forget_bias, void *stream) {
    AIR_REQUIRE(h_prev && c_prev && w_h_bf16 && gx && h && c && gate_act, AIR_E_NULL);
    AIR_REQUIRE(M > 0 && Hd > 0 && Hd % 64 == 0 && ldw >= 4 * Hd && ldw % 4 == 0 && ldgx >= 4 * Hd, AIR_E_SHAPE);
    AIR_REQUIRE(air_aligned16(h_prev) && ((uintptr_t)w_h_bf16 % 8 == 0) && (!h_prev_bf16 || (uintptr_t)h_prev_bf16 % 16 == 0), AIR_E_ALIGN);
    LstmFwdArgs g;
    g.h_prev = h_prev; g.w_h = nullptr; g.gx = gx; g.c_prev = c_prev; g.h = h; g.c = c; g.gate_act = gate_act;
    g.M = M; g.Hd = Hd; g.ldw = ldw; g.ldgx = ldgx; g.vecA = 1; g.ldh = Hd; g.ldc = Hd; g.fb = forget_bias;
    g.tiles = air_cdiv(M, 16) * (Hd / 64);
    const Lstm16 x = {w_h_bf16, h_prev_bf16, h_bf16, nullptr};
    hipLaunchKernelGGL(lstm_fwd_wide16_kernel, dim3(g.tiles), dim3(512), 0, air_stream(stream), g, x);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
extern "C" int air_lstm_step_bwd_bf16(const float *dgates_next, const void *dgates_next_bf16, const void *w_h_bf16,
                                      const float *dh_a, const float *dh_b, const float *dc_in, const float *gate_act,
                                      const float *c_prev, const float *c, const float *dgx_in, float *dgates,
                                      void *dgates_bf16, float *dc_prev, float *dgx_out, void *dgx_bf16, int M, int Hd,
                                      void *stream) {
    AIR_REQUIRE(dgates_next && w_h_bf16 && gate_act && c_prev && c && dgates && dc_prev, AIR_E_NULL);
    AIR_REQUIRE(M > 0 && Hd > 0 && Hd % 64 == 0, AIR_E_SHAPE);
    AIR_REQUIRE(air_aligned16(dgates_next) && ((uintptr_t)w_h_bf16 % 16 == 0) &&
                    (!dgates_next_bf16 || (uintptr_t)dgates_next_bf16 % 16 == 0), AIR_E_ALIGN);
    LstmBwdArgs g;
    g.dgates_next = dgates_next; g.w_h = nullptr; g.dh_a = dh_a; g.dh_b = dh_b; g.dc_in = dc_in; g.gate_act = gate_act;
    g.c_prev = c_prev; g.c = c; g.dgx_in = dgx_in; g.dgates = dgates; g.dc_prev = dc_prev; g.dgx_out = dgx_out;
    g.M = M; g.Hd = Hd; g.vecA = 1; g.vecB = 1;
    const Lstm16 x = {w_h_bf16, dgates_next_bf16, dgates_bf16, dgx_bf16};
    hipLaunchKernelGGL(lstm_bwd_wide16_kernel, dim3(air_cdiv(M, 16) * (Hd / 64)), dim3(512), 0, air_stream(stream), g, x);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

template <bool BF>
static int lstm_fwd_launch(const LstmFwdArgs &g, const PrologueArgs &pro, int extra_blocks, hipStream_t st) {
    // more than 512 16x16 tiles of (batch, hidden): the wide-tile form (needs 16-byte addressable operands and Hd % 64 == 0)
    const bool wide = air_cdiv(g.M, 16) * air_cdiv(g.Hd, 16) > 512 && g.Hd % 64 == 0 && g.ldw % 4 == 0 && g.ldh % 4 == 0 &&
                      air_aligned16(g.w_h) && air_aligned16(g.h_prev);
    if (wide) {
        LstmFwdArgs gw = g;
        gw.tiles = air_cdiv(g.M, 16) * (g.Hd / 64);
        hipLaunchKernelGGL((lstm_fwd_wide_kernel<BF>), dim3(gw.tiles + extra_blocks), dim3(512), 0, st, gw, pro);
    } else {
        hipLaunchKernelGGL((lstm_fwd_fused_kernel<BF>), dim3(g.tiles + extra_blocks), dim3(256), 0, st, g, pro);
    }
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
static int lstm_fwd_fill(LstmFwdArgs &g, const float *h_prev, int ldh, const float *c_prev, int ldc, const float *w_h,
                         int ldw, const float *gx, int ldgx, float *h, float *c, float *gate_act, int M, int Hd,
                         float forget_bias, int precision) {
    AIR_REQUIRE(h_prev && c_prev && w_h && gx && h && c && gate_act, AIR_E_NULL);
    AIR_REQUIRE(M > 0 && Hd > 0 && ldw >= 4 * Hd && ldgx >= 4 * Hd, AIR_E_SHAPE);
    AIR_REQUIRE((ldh == 0 || ldh >= Hd) && (ldc == 0 || ldc >= Hd), AIR_E_SHAPE);
    AIR_REQUIRE(precision == AIR_PREC_F32 || precision == AIR_PREC_BF16, AIR_E_UNSUPPORTED);
    g.h_prev = h_prev; g.w_h = w_h; g.gx = gx; g.c_prev = c_prev; g.h = h; g.c = c; g.gate_act = gate_act;
    g.M = M; g.Hd = Hd; g.ldw = ldw; g.ldgx = ldgx; g.fb = forget_bias; g.ldh = ldh; g.ldc = ldc;
    g.vecA = ((ldh % 4) == 0 && air_aligned16(h_prev)) ? 1 : 0;
    g.tiles = air_cdiv(M, 16) * air_cdiv(Hd, 4);
    return AIR_OK;
}
extern "C" int air_lstm_step_fwd(const float *h_prev, const float *c_prev, const float *w_h, int ldw, const float *gx,
                                 int ldgx, float *h, float *c, float *gate_act, int M, int Hd, float forget_bias,
                                 int precision, void *stream) {
    LstmFwdArgs g;
    int st = lstm_fwd_fill(g, h_prev, Hd, c_prev, Hd, w_h, ldw, gx, ldgx, h, c, gate_act, M, Hd, forget_bias, precision);
    if (st) return st;
    PrologueArgs pro = {};
    return precision == AIR_PREC_BF16 ? lstm_fwd_launch<true>(g, pro, 0, air_stream(stream))
                                      : lstm_fwd_launch<false>(g, pro, 0, air_stream(stream));
}
// First LSTM step of a train step with the step prologue riding along: h0 / c0 [1,Hd] are read with a broadcast row stride
// by the step itself, while extra workgroups draw the step's noise, evaluate the annealed prior and write the tiled
// initial state (needed only by later launches: the backward reads h_tiled / c_tiled).
extern "C" int air_lstm_step_fwd_prologue(const float *h0, const float *c0, const float *w_h, int ldw, const float *gx,
                                          int ldgx, float *h, float *c, float *gate_act, int M, int Hd, float forget_bias,
                                          int precision, float *normal, size_t n_normal, float *uniform, size_t n_uniform,
                                          const uint64_t *rng_state_dev, const int64_t *global_step_dev, int anneal_type,
                                          double init, double final_value, double anneal_steps, double hold_for,
                                          double steps_div, double *prior_out_f64, int T, float *h_tiled, float *c_tiled,
                                          void *stream) {
    AIR_REQUIRE(rng_state_dev && global_step_dev && prior_out_f64 && h_tiled && c_tiled, AIR_E_NULL);
    AIR_REQUIRE((n_normal == 0 || normal) && (n_uniform == 0 || uniform), AIR_E_NULL);
    AIR_REQUIRE(T > 0 && anneal_type >= 0 && anneal_type <= 2, AIR_E_SHAPE);
    LstmFwdArgs g;
    int st = lstm_fwd_fill(g, h0, 0, c0, 0, w_h, ldw, gx, ldgx, h, c, gate_act, M, Hd, forget_bias, precision);
    if (st) return st;
    const PrologueArgs pro = make_prologue_args(normal, n_normal, uniform, n_uniform, rng_state_dev, global_step_dev,
                                                anneal_type, init, final_value, anneal_steps, hold_for, steps_div,
                                                prior_out_f64, T, h0, c0, h_tiled, c_tiled, M, Hd);
    const int extra = prologue_blocks(pro);
    return precision == AIR_PREC_BF16 ? lstm_fwd_launch<true>(g, pro, extra, air_stream(stream))
                                      : lstm_fwd_launch<false>(g, pro, extra, air_stream(stream));
}

// air_lstm_step_fwd_prologue with the hoisted input product x . W_x + b folded in (lstm_fwd_first_kernel): latency regime only --
// AIR_E_UNSUPPORTED beyond 512 tiles of (batch, hidden), where the caller keeps the gx launch and the wide-tile first step.
extern "C" int air_lstm_first_step_fwd(const float *x, int ldx, int E, const float *w_x, const float *b_gates, const float *h0,
                                       const float *c0, const float *w_h, int ldw, float *gx_out, int ldgx, float *h, float *c,
                                       float *gate_act, int M, int Hd, float forget_bias, int precision, float *normal,
                                       size_t n_normal, float *uniform, size_t n_uniform, const uint64_t *rng_state_dev,
                                       const int64_t *global_step_dev, int anneal_type, double init, double final_value,
                                       double anneal_steps, double hold_for, double steps_div, double *prior_out_f64, int T,
                                       float *h_tiled, float *c_tiled, void *stream) {
    AIR_REQUIRE(x && w_x && b_gates && gx_out, AIR_E_NULL);
    AIR_REQUIRE(rng_state_dev && global_step_dev && prior_out_f64 && h_tiled && c_tiled, AIR_E_NULL);
    AIR_REQUIRE((n_normal == 0 || normal) && (n_uniform == 0 || uniform), AIR_E_NULL);
    AIR_REQUIRE(T > 0 && anneal_type >= 0 && anneal_type <= 2 && E > 0 && ldx >= E, AIR_E_SHAPE);
    AIR_REQUIRE(air_cdiv(M, 16) * air_cdiv(Hd, 16) <= 512, AIR_E_UNSUPPORTED);
    LstmFwdArgs g;
    int st = lstm_fwd_fill(g, h0, 0, c0, 0, w_h, ldw, gx_out, ldgx, h, c, gate_act, M, Hd, forget_bias, precision);
    if (st) return st;
    LstmFirstArgs f;
    f.x = x; f.w_x = w_x; f.b = b_gates; f.gx_out = gx_out; f.E = E; f.ldx = ldx;
    f.vecX = ((ldx % 4) == 0 && air_aligned16(x)) ? 1 : 0;
    const PrologueArgs pro = make_prologue_args(normal, n_normal, uniform, n_uniform, rng_state_dev, global_step_dev,
                                                anneal_type, init, final_value, anneal_steps, hold_for, steps_div,
                                                prior_out_f64, T, h0, c0, h_tiled, c_tiled, M, Hd);
    const int extra = prologue_blocks(pro);
    if (precision == AIR_PREC_BF16)
        hipLaunchKernelGGL((lstm_fwd_first_kernel<true>), dim3(g.tiles + extra), dim3(256), 0, air_stream(stream), g, f, pro);
    else
        hipLaunchKernelGGL((lstm_fwd_first_kernel<false>), dim3(g.tiles + extra), dim3(256), 0, air_stream(stream), g, f, pro);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

template <bool BF>
static int lstm_bwd_launch(const LstmBwdArgs &g, const RmspropSlice &opt, size_t opt_nq, hipStream_t st) {
    const int tiles = air_cdiv(g.M, 16) * air_cdiv(g.Hd, 16);
    // few tiles (batch 64: 64 of them): 16 waves share the 4Hd-deep contraction of a tile; many tiles: 4 waves
    const bool wide = tiles > 512 && g.Hd % 64 == 0 && g.vecA && g.vecB;
    const int nth = tiles <= 512 ? 1024 : (wide ? 512 : 256);
    size_t extra = air_rider_blocks(opt_nq, nth, 512);                     // about two float4 per thread of the riding slice
    if (wide) hipLaunchKernelGGL((lstm_bwd_wide_kernel<BF>), dim3(air_cdiv(g.M, 16) * (g.Hd / 64) + (int)extra), dim3(512), 0, st, g, opt);
    else if (tiles <= 512) hipLaunchKernelGGL((lstm_bwd_fused_kernel<16, BF>), dim3(tiles + (int)extra), dim3(1024), 0, st, g, opt);
    else hipLaunchKernelGGL((lstm_bwd_fused_kernel<4, BF>), dim3(tiles + (int)extra), dim3(256), 0, st, g, opt);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}
extern "C" int air_lstm_step_bwd(const float *dgates_next, const float *w_h, const float *dh_a, const float *dh_b,
                                 const float *dc_in, const float *gate_act, const float *c_prev, const float *c,
                                 const float *dgx_in, float *dgates, float *dc_prev, float *dgx_out, int M, int Hd,
                                 int precision, void *stream) {
    return air_lstm_step_bwd_opt(dgates_next, w_h, dh_a, dh_b, dc_in, gate_act, c_prev, c, dgx_in, dgates, dc_prev, dgx_out, M,
                                 Hd, precision, nullptr, stream);
}
extern "C" int air_lstm_step_bwd_opt(const float *dgates_next, const float *w_h, const float *dh_a, const float *dh_b,
                                     const float *dc_in, const float *gate_act, const float *c_prev, const float *c,
                                     const float *dgx_in, float *dgates, float *dc_prev, float *dgx_out, int M, int Hd,
                                     int precision, const AirRmspropSlice *opt, void *stream) {
    RmspropSlice os; size_t onq;
    { int st_ = rmsprop_slice_from_abi(opt, os, &onq); if (st_) return st_; }
    AIR_REQUIRE(dgates_next && w_h && gate_act && c_prev && c && dgates && dc_prev, AIR_E_NULL);
    AIR_REQUIRE(M > 0 && Hd > 0, AIR_E_SHAPE);
    AIR_REQUIRE(precision == AIR_PREC_F32 || precision == AIR_PREC_BF16, AIR_E_UNSUPPORTED);
    LstmBwdArgs g;
    g.dgates_next = dgates_next; g.w_h = w_h; g.dh_a = dh_a; g.dh_b = dh_b; g.dc_in = dc_in; g.gate_act = gate_act;
    g.c_prev = c_prev; g.c = c; g.dgx_in = dgx_in; g.dgates = dgates; g.dc_prev = dc_prev; g.dgx_out = dgx_out;
    g.M = M; g.Hd = Hd;
    g.vecA = air_aligned16(dgates_next) ? 1 : 0;          // row stride 4*Hd floats is always a multiple of 16 bytes
    g.vecB = air_aligned16(w_h) ? 1 : 0;
    return precision == AIR_PREC_BF16 ? lstm_bwd_launch<true>(g, os, onq, air_stream(stream))
                                      : lstm_bwd_launch<false>(g, os, onq, air_stream(stream));
}

extern "C" int air_lstm_step_bwd_entry_fits(int M, int Hd) {
    return (M > 0 && Hd > 0 && Hd % 16 == 0 && air_cdiv(M, 16) * (Hd / 16) <= 512) ? 1 : 0;
}
extern "C" int air_lstm_step_bwd_entry(const float *gate_act1, const float *c_prev1, const float *c1, const float *dh_a1,
                                       const float *dh_b1, float *dgates1, float *dc_prev1, const float *w_h, const float *dh_a,
                                       const float *dh_b, const float *gate_act, const float *c_prev, const float *c, float *dgates,
                                       float *dc_prev, float *dgx_out, int M, int Hd, const AirRmspropSlice *opt, void *stream) {
    RmspropSlice os; size_t onq;
    { int st_ = rmsprop_slice_from_abi(opt, os, &onq); if (st_) return st_; }
    AIR_REQUIRE(gate_act1 && c_prev1 && c1 && dgates1 && dc_prev1 && w_h && gate_act && c_prev && c && dgates && dc_prev, AIR_E_NULL);
    AIR_REQUIRE(dh_a1 || dh_b1, AIR_E_NULL);
    AIR_REQUIRE(air_lstm_step_bwd_entry_fits(M, Hd) == 1, AIR_E_UNSUPPORTED);
    // (Hd % 16 == 0: every row of every operand is a multiple of 64 bytes; the bases must be 16-byte aligned)
    AIR_REQUIRE(air_aligned16(gate_act1) && air_aligned16(c_prev1) && air_aligned16(c1) && air_aligned16(dgates1) &&
                air_aligned16(dc_prev1) && air_aligned16(w_h) && (!dh_a1 || air_aligned16(dh_a1)) && (!dh_b1 || air_aligned16(dh_b1)),
                AIR_E_ALIGN);
    LstmBwdArgs g;
    g.dgates_next = dgates1; g.w_h = w_h; g.dh_a = dh_a; g.dh_b = dh_b; g.dc_in = nullptr; g.gate_act = gate_act;
    g.c_prev = c_prev; g.c = c; g.dgx_in = nullptr; g.dgates = dgates; g.dc_prev = dc_prev; g.dgx_out = dgx_out;
    g.M = M; g.Hd = Hd; g.vecA = 1; g.vecB = 1;
    LstmEntryArgs en;
    en.gate_act1 = gate_act1; en.c_prev1 = c_prev1; en.c1 = c1; en.dh_a1 = dh_a1; en.dh_b1 = dh_b1; en.dgates1 = dgates1;
    en.dc_prev1 = dc_prev1;
    const int tiles = air_cdiv(M, 16) * (Hd / 16);
    size_t extra = air_rider_blocks(onq, 1024, 512);
    hipLaunchKernelGGL(lstm_bwd_entry_kernel, dim3(tiles + (int)extra), dim3(1024), 0, air_stream(stream), g, en, os);
    AIR_LAUNCH_CHECK();
    return AIR_OK;
}

// ---- linear layer wrappers (neural.py:56-60) ------------------------------------------------------------------
__global__ __launch_bounds__(256) void mul_delu_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                       float *__restrict__ g, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float yy = y[i];
        g[i] = dy[i] * (yy > 0.f ? 1.f : yy + 1.f);
    }
}

extern "C" int air_linear_fwd(const float *x, const float *w, const float *b, float *y, int M, int K, int N, int act,
                              void *ws, size_t ws_bytes, void *stream) {
    AIR_REQUIRE(act == AIR_ACT_NONE || act == AIR_ACT_ELU, AIR_E_UNSUPPORTED);
    AIR_REQUIRE(act == AIR_ACT_NONE || b, AIR_E_NULL);
    const int epi = (act == AIR_ACT_ELU) ? AIR_EPI_BIAS_ELU : (b ? AIR_EPI_BIAS : AIR_EPI_NONE);
    return air_gemm(0, 0, M, N, K, x, K, w, N, y, N, b, epi, nullptr, 0, 0.f, nullptr, ws, ws_bytes, stream);
}

extern "C" int air_linear_bwd(const float *x, const float *w, const float *y, const float *dy, float *dx, float *dw,
                              float *db, float *gbuf, int M, int K, int N, int act, void *ws, size_t ws_bytes,
                              void *stream) {
    AIR_REQUIRE(x && w && dy && dw, AIR_E_NULL);
    AIR_REQUIRE(act == AIR_ACT_NONE || act == AIR_ACT_ELU, AIR_E_UNSUPPORTED);
    const float *g = dy;
    if (act == AIR_ACT_ELU) {
        AIR_REQUIRE(y && gbuf, AIR_E_NULL);
        const size_t n = (size_t)M * N;
        int blocks = (int)((n + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(mul_delu_kernel, dim3(blocks), dim3(256), 0, air_stream(stream), dy, y, gbuf, n);
        AIR_LAUNCH_CHECK();
        g = gbuf;
    }
    int st;
    if (dx) {   // dx[M,K] = g[M,N] . w[K,N]^T
        st = air_gemm(0, 1, M, K, N, g, N, w, N, dx, K, nullptr, AIR_EPI_NONE, nullptr, 0, 0.f, nullptr, ws, ws_bytes, stream);
        if (st) return st;
    }
    // dw[K,N] = x[M,K]^T . g[M,N]; db = colsum(g)
    st = air_gemm(1, 0, K, N, M, x, K, g, N, dw, N, nullptr, AIR_EPI_NONE, nullptr, 0, 0.f, db, ws, ws_bytes, stream);
    return st;
}
