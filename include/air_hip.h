/* air_hip.h -- C ABI of libair_hip.so: the MI355X (gfx950) kernels of the AIR hot path.
 *
 * The reference (akosiorek/attend_infer_repeat) has no FFI of its own: its only native operator boundary is the TF
 * custom-op pair behind `snt.resampler(data, warp)` (attend_infer_repeat/modules.py:109) and the TF/Eigen kernels
 * behind snt.Linear / snt.LSTM / tf.contrib.distributions.  Each entry point below replaces one of those op groups;
 * the comment on each cites the reference call site it stands in for.  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - extern "C"; returns int: 0 = ok, <0 = AIR_E_* argument error, >0 = hipError_t of the failed launch.
 *   - every pointer is a DEVICE pointer to contiguous row-major float32 unless stated (`*_f64`: double, `host`).
 *   - the caller owns every buffer, including workspaces; nothing is allocated, no host sync, no global state
 *     => re-entrant and hipGraph-capturable.  `stream` is a hipStream_t passed as void*.
 *   - `where` rows are [sx, tx, sy, ty] (modules.py:41-46, evaluation.py:23-28).
 */
#ifndef AIR_HIP_H
#define AIR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* TWO classes of entry points, marked on every declaration:
 *   AIR_API         -- the STABLE CONTRACT: the operator boundary of SURVEY 8(b) (spatial transformer read / write, linear + GEMM,
 *                      LSTM cell, Gaussian sample + KL, presence, reconstruction term, num-steps posterior, NVIL, centred RMSProp,
 *                      each with its `_bwd`), the whole-step canvas forms, the RNG / utility launches, hipGraph / event / stream
 *                      plumbing and the RCCL communicator.  What a maintainer of the reference binds.  Its
 *                      prototypes are pinned in tests/golden/abi_stable.txt; AIR_ABI_VERSION changes when one of them does.
 *   AIR_ENGINE_API  -- ENGINE PLAN ENTRIES: the fused / folded launches attend_infer_repeat_amd/engine_plan.py strings together
 *                      (riders, struct-argument epilogues, bf16 mirrors, the hipIpc data-parallel nodes).  They change whenever a
 *                      fold changes -- AIR_ENGINE_ABI_VERSION counts those changes -- and nobody outside this repository should
 *                      bind them.
 * ctypes cannot check argument lists: the loader compares both numbers and the build digest. */
#define AIR_ABI_VERSION 10
#define AIR_ENGINE_ABI_VERSION 5
#define AIR_API
#define AIR_ENGINE_API

enum {
    AIR_OK = 0,
    AIR_E_NULL = -1,      /* required pointer is NULL */
    AIR_E_SHAPE = -2,     /* non-positive / inconsistent dimension */
    AIR_E_ALIGN = -3,     /* pointer or leading dimension violates an alignment requirement */
    AIR_E_WORKSPACE = -4, /* workspace too small */
    AIR_E_UNSUPPORTED = -5
};

enum { AIR_ACT_NONE = 0, AIR_ACT_ELU = 1 };

/* GEMM epilogues (air_gemm): applied to acc = op(A).op(B) (+ beta*C) */
enum {
    AIR_EPI_NONE = 0,
    AIR_EPI_BIAS = 1,        /* + bias[n]                                    snt.Linear, neural.py:56-60          */
    AIR_EPI_BIAS_ELU = 2,    /* elu(acc + bias[n])                           Affine(transfer=elu), neural.py:58-59 */
    AIR_EPI_MUL_DELU = 3,    /* acc * elu'(aux[m,n]) with aux = saved elu OUTPUT (y>0 ? 1 : y+1): backward of ELU  */
    AIR_EPI_ADD_AUX = 4,     /* acc + aux[m,n] (+ bias[n] if given)          LSTM: x.Wx hoisted, h.Wh added        */
    AIR_EPI_ADD_AUX_ELU = 5  /* elu(acc + aux[m,n] (+ bias[n] if given))     layer whose input is a concat: parts summed */
};

AIR_API int air_abi_version(void);
AIR_ENGINE_API int air_engine_abi_version(void);
/* sha256 of the sources + headers this binary was compiled from (attend_infer_repeat_amd/build.py passes it at compile
 * time); the loader compares it with the sources it finds next to the library, so a stale binary is refused instead of
 * being called with a changed argument list. */
AIR_API const char *air_build_digest(void);
AIR_API const char *air_status_string(int status);

/* ---- spatial transformer ------------------------------------------------------------------------------------
 * Replaces snt.AffineGridWarper + snt.resampler (+ registered gradient) at modules.py:100-109.                     */

/* Glimpse read, cell.py:135.  glimpse[k] = bilinear(img[k % n_img], grid(where[k])), k < n.
 * n_img == n for one image per glimpse; n = T*n_img when T glimpses are read from each image (batched unroll).   */
AIR_API int air_st_read_fwd(const float *img, const float *where, float *glimpse,
                    int n, int n_img, int H, int W, int h, int w, void *stream);
/* dwhere[n,4] always; dimg[n_img,H,W] optional (NULL to skip; requires n_img == n).                               */
AIR_API int air_st_read_bwd(const float *img, const float *where, const float *dglimpse, float *dwhere, float *dimg,
                    int n, int n_img, int H, int W, int h, int w, void *stream);

/* Canvas write, cell.py:159-165: canvas_out[k] = canvas_in[k] + presence[k] * inverse_warp(glimpse[k], where[k]).
 * canvas_in may be NULL (zeros) or alias canvas_out; presence may be NULL (ones).                                  */
AIR_API int air_st_write_fwd(const float *glimpse, const float *where, const float *presence, const float *canvas_in,
                     float *canvas_out, int n, int H, int W, int h, int w, void *stream);
/* Gradients of the above wrt glimpse, where and (optionally, NULL to skip) presence given dcanvas[n,H,W].         */
AIR_API int air_st_write_bwd(const float *glimpse, const float *where, const float *presence, const float *dcanvas,
                     float *dglimpse, float *dwhere, float *dpresence,
                     int n, int H, int W, int h, int w, void *stream);

/* Fused T-step canvas accumulation + reconstruction term (cell.py:159-165 over dynamic_rnn model.py:83-84, then
 * model.py:92-97, 319-324).  glimpse[T,B,h,w], where[T,B,4], presence[T,B] time-major.
 *   canvas_steps[T,B,H,W] (optional): running canvas after each step, UNscaled.
 *   final_canvas[B,H,W]: canvas after T steps, UNscaled.
 *   rec_per_sample[B] (optional, needs obs): sum_pix -log N(obs | mult*canvas, std).                              */
AIR_API int air_canvas_unroll_fwd(const float *glimpse, const float *where, const float *presence, const float *obs,
                          float *canvas_steps, float *final_canvas, float *rec_per_sample,
                          int T, int B, int H, int W, int h, int w, float mult, float std, void *stream);
/* The same unroll with each image cut into `n_bands` horizontal row bands, one workgroup per (image, band): a small batch
 * then fills the chip (64 images x 4 bands = 256 workgroups).  rec_parts[n_bands, B] receives each band's share of the
 * reconstruction term; consumers add the shares in band order (air_nvil_parts, air_canvas_unroll_bwd_nvil, or
 * air_sum_leading for the plain sum).  n_bands must be what air_canvas_unroll_bands(B, H) returns (or 1).            */
AIR_ENGINE_API int air_canvas_unroll_bands(int B, int H);
AIR_ENGINE_API int air_canvas_unroll_fwd_banded(const float *glimpse, const float *where, const float *presence, const float *obs,
                                 float *canvas_steps, float *final_canvas, float *rec_parts, int n_bands,
                                 int T, int B, int H, int W, int h, int w, float mult, float std, void *stream);

/* Backward of mean_b(rec_per_sample) * loss_scale through the fused op: dcanvas is formed on the fly from
 * (final_canvas, obs).  Outputs dglimpse[T,B,h,w], dwhere[T,B,4].                                                  */
AIR_API int air_canvas_unroll_bwd(const float *glimpse, const float *where, const float *presence, const float *obs,
                          const float *final_canvas, float *dglimpse, float *dwhere,
                          int T, int B, int H, int W, int h, int w, float mult, float std, float loss_scale,
                          void *stream);
/* air_canvas_unroll_bwd that also returns dpresence[T,B] = sum_pix dL/dcanvas * (the step's write) -- what a continuous presence
 * (discrete_steps=False: cell.py:150-151, 163) receives from the canvas write.                                           */
AIR_ENGINE_API int air_canvas_unroll_bwd_dpresence(const float *glimpse, const float *where, const float *presence, const float *obs,
                                    const float *final_canvas, float *dglimpse, float *dwhere, float *dpresence,
                                    int T, int B, int H, int W, int h, int w, float mult, float std, float loss_scale,
                                    void *stream);
/* The same launch with one extra workgroup that evaluates air_nvil(imp, baseline, logp, nvil_out, dlogp, dbaseline, B)
 * (the two are independent; the step is bound by the number of dependent launches).                                  */
AIR_ENGINE_API int air_canvas_unroll_bwd_nvil(const float *glimpse, const float *where, const float *presence, const float *obs,
                               const float *final_canvas, float *dglimpse, float *dwhere,
                               int T, int B, int H, int W, int h, int w, float mult, float std, float loss_scale,
                               const float *imp_parts, int n_parts, float *imp_sum, const float *baseline,
                               const float *logp, float *nvil_out, float *dlogp, float *dbaseline, float *ema_dev, void *stream);
/* Canvas forward (banded, as air_canvas_unroll_fwd_banded) and backward (as air_canvas_unroll_bwd with final_canvas = NULL: every
 * (t, b) unit re-forms the canvas on its own footprint, bit-identically to the forward, so it reads nothing the forward writes) as
 * the two roles of ONE launch: one dependent launch less on the train step's chain at small batch.  The NVIL objective, which
 * needs the forward's reconstruction shares, then rides on air_gauss_sample_bwd_nvil.
 * n_split (1..4): workgroups per backward unit.  They own disjoint rows of the unit's dglimpse; dwhere is then written as
 * n_split slabs, dwhere[n_split][T*B][4], whose SUM (slab 0 + slab 1 + ..., in that order) is the gradient -- the consumer adds
 * them (air_attend_bwd's `dwhere_w_slabs`).  B * n_bands and B * T * n_split at most 4096, and the launch's LDS must fit:
 * air_canvas_unroll_fwd_bwd_fits(...) == 1 says so without launching (plan builders ask it and keep the two launches otherwise). */
AIR_ENGINE_API int air_canvas_unroll_fwd_bwd_fits(int n_bands, int n_split, int T, int B, int H, int W, int h, int w);
AIR_ENGINE_API int air_canvas_unroll_fwd_bwd(const float *glimpse, const float *where, const float *presence, const float *obs,
                              float *canvas_steps, float *final_canvas, float *rec_parts, int n_bands, float *dglimpse,
                              float *dwhere, int n_split, int T, int B, int H, int W, int h, int w, float mult, float std,
                              float loss_scale, void *stream);
/* ---- dense layers -------------------------------------------------------------------------------------------
 * Replaces the TF MatMul/BiasAdd/Elu nodes under snt.Linear (neural.py:42-60) and snt.LSTM (mnist_model.py:35).   */

/* C[M,N] = epi( op(A)[M,K] . op(B)[K,N] + beta*C ).  ta==0: A is [M,K] (lda); ta!=0: A is stored [K,M].
 * tb==0: B is [K,N] (ldb); tb!=0: B is stored [N,K].  fp32 MFMA (v_mfma_f32_16x16x4_f32), exact fp32 accumulate.
 * bias[N] / aux[M,N](ldaux) as required by `epilogue`.  If colsum != NULL (only with ta!=0): colsum[n] = sum_k
 * op(B)[k,n] (bias gradient fused into the dW GEMM).  ws / ws_bytes: optional split-K workspace (may be NULL).    */
AIR_API int air_gemm(int ta, int tb, int M, int N, int K, const float *A, int lda, const float *B, int ldb,
             float *C, int ldc, const float *bias, int epilogue, const float *aux, int ldaux, float beta,
             float *colsum, void *ws, size_t ws_bytes, void *stream);
/* Same contract with the operands rounded to bf16 (round-to-nearest-even) in registers and multiplied on
 * v_mfma_f32_16x16x16_bf16 (fp32 accumulate, fp32 storage everywhere): BASELINE.json configs[4], "bf16 MFMA MLP path".
 * colsum (the bias gradient) is summed from the un-rounded fp32 values.                                             */
AIR_ENGINE_API int air_gemm_bf16(int ta, int tb, int M, int N, int K, const float *A, int lda, const float *B, int ldb,
             float *C, int ldc, const float *bias, int epilogue, const float *aux, int ldaux, float beta,
             float *colsum, void *ws, size_t ws_bytes, void *stream);
size_t air_gemm_workspace_bytes(int M, int N, int K);

/* Up to 8 INDEPENDENT GEMMs in one launch (same semantics as air_gemm, no split-K).  The step is launch/latency
 * bound at batch 64, so e.g. the dW and dX products of one layer, or the transform / steps heads that share h_t,
 * are dispatched together.  `descs` is a HOST array; outputs must not alias another problem's inputs.              */
enum { AIR_PREC_F32 = 0, AIR_PREC_BF16 = 1 };
typedef struct AirGemmDesc {
    int ta, tb, M, N, K;
    const float *A; int lda;
    const float *B; int ldb;
    float *C; int ldc;
    const float *bias; int epilogue;
    const float *aux; int ldaux;
    float beta;
    float *colsum;
    int precision;           /* AIR_PREC_F32 (exact fp32 MFMA) or AIR_PREC_BF16 (operands rounded to bf16, fp32 accumulate) */
    /* Consumer-side reduction of a K-split producer (all optional, NULL / 0 = off).  A long-K product on a handful of tiles is
     * bound by what ONE CU can ingest; splitting K over more workgroups needs the partial results summed, and instead of a
     * cross-workgroup hand-off inside the producer (agent-scope fences) the CONSUMER adds them where it reads them:
     *   a[m,k] = act(A[m,k] + A2[m,k] + a_bias[k]),  act = ELU if a_elu  (ta == 0, K % 16 == 0, single-problem launch only);
     *   a_out[M, lda] receives the reduced activation (written once, by the first column of tiles) for the backward pass.
     * (A second aux slab needs no field: the partial product is written into C and picked up with beta = 1.)             */
    const float *A2;
    const float *a_bias;
    int a_elu;
    float *a_out;
    /* bf16 DATA path (precision == AIR_PREC_BF16; all optional, NULL = off).  A16 / B16: bf16 mirrors of A / B -- the same
     * values rounded to bf16 (RNE), same shape and leading dimension in ELEMENTS -- which the throughput-regime kernels read
     * instead of the fp32 buffers (half the operand bytes, v_mfma_f32_16x16x32_bf16) when every problem of the launch names
     * one for that operand; C16: the epilogue also stores bf16(C) there, the mirror a later product reads.  The fp32 buffers
     * stay authoritative: whatever is not a dense product (ELU', the losses, the optimiser) keeps reading them.              */
    const void *A16;
    const void *B16;
    void *C16;
} AirGemmDesc;
/* count <= 8; up to 24 for the deferred weight gradients of a whole step in one launch: problems the wide-tile kernel takes
 * (ta = 1, tb = 0, M, N, K and ldb multiples of 4, B 16-byte aligned) on 64x64 tiles -- at least one --, any other problem
 * (no A2) on 16x16 tiles in the same grid.                                                                                 */
AIR_ENGINE_API int air_gemm_grouped(const AirGemmDesc *descs, int count, void *stream);

/* air_gemm_grouped for the FIRST product(s) of a train step whose batch is drawn from an HBM-resident dataset (data.py:121-158's feeder
 * in HBM): the gather of air_batch_gather folded into the A-operand load.  Every problem's A lies inside obs[B, item_floats] (row 0 + a
 * column offset, lda = item_floats, not transposed, fp32) and is read from item idx_m of `dataset` instead -- idx_m drawn exactly as
 * air_batch_gather draws it --; the problems of copy_mask write the rows they read into obs (together they must cover every column),
 * idx_out (optional) receives the indices.  air_gemm_grouped_gather_fits answers, without launching, whether a launch qualifies.       */
typedef struct AirBatchGather {
    const float *dataset;
    long long n_items;
    int item_floats, shuffle, B;
    const uint64_t *seed_dev;
    const int64_t *step_dev;
    float *obs;
    int64_t *idx_out;
    unsigned copy_mask;
} AirBatchGather;
AIR_ENGINE_API int air_gemm_grouped_gather_fits(const AirGemmDesc *descs, int count, const AirBatchGather *g);
AIR_ENGINE_API int air_gemm_grouped_gather(const AirGemmDesc *descs, int count, const AirBatchGather *g, void *stream);

/* Row-slab dX chains of MLPs on the bf16 data path (csrc/mlp_chain_kernels.hip; neural.py:93-102 backward): through Linear + ELU layers
 * row r of dA_{l-1} = (dA_l . W_l^T) * elu'(out_{l-1}) needs only row r of dA_l, so ONE launch walks a whole chain of layers per slab of
 * 16 rows instead of one dependent launch per layer.  Chain c: g_in[rows, layer[0].n_in] (fp32, row pitch ld_in) is the gradient wrt
 * the last layer's pre-activation; layer l (listed from the output side): w_bf16 = the bf16 shadow of W_l[n_out, n_in] (row-major: the
 * Sonnet layout w[in, out] of the FORWARD layer, whose `in` is this n_out), aux = the saved ELU output the result is differentiated
 * through (NULL: none), out[rows, n_out] fp32 (pitch ldout) and optionally its bf16 mirror with the same pitch.  n_in % 4 == 0 or
 * n_in < 32; widths up to 1024; up to 4 chains of up to 4 layers share the launch.  fp32 accumulate; a layer consumes bf16 of the
 * previous fp32 result -- what the per-layer launches read from the mirrors.                                                        */
#define AIR_DXC_MAX_LAYERS 4
#define AIR_DXC_MAX_CHAINS 4
typedef struct AirDxLayer {
    const void *w_bf16;
    const float *aux;
    float *out;
    void *out_bf16;
    int n_in, n_out, ldaux, ldout;
} AirDxLayer;
typedef struct AirDxChain {
    const float *g_in;
    int ld_in, rows, n_layers;
    AirDxLayer layer[AIR_DXC_MAX_LAYERS];
} AirDxChain;
AIR_ENGINE_API int air_mlp_dx_chain_fits(int n_in, int n_out);
AIR_ENGINE_API int air_mlp_dx_chain_bf16(const AirDxChain *chains, int n_chains, void *stream);

/* y = act(x.w + b), neural.py:56-60.  x[M,K], w[K,N] (Sonnet layout), b[N] (may be NULL), y[M,N].                 */
AIR_API int air_linear_fwd(const float *x, const float *w, const float *b, float *y, int M, int K, int N, int act,
                   void *ws, size_t ws_bytes, void *stream);
/* Backward: g = dy * act'(y); dx = g.w^T (NULL to skip); dw = x^T.g; db = colsum(g) (NULL to skip).
 * gbuf[M,N] is required when act != AIR_ACT_NONE (holds g).                                                        */
AIR_API int air_linear_bwd(const float *x, const float *w, const float *y, const float *dy, float *dx, float *dw, float *db,
                   float *gbuf, int M, int K, int N, int act, void *ws, size_t ws_bytes, void *stream);

/* LSTM pointwise, Sonnet v1 gate order i,j,f,o (cell.py:126-127): c' = sig(f+fb)*c + sig(i)*tanh(j);
 * h' = tanh(c')*sig(o).  gates[M,4H] pre-activation; gate_act[M,4H] receives the activated gates (saved for bwd). */
AIR_API int air_lstm_pointwise_fwd(const float *gates, const float *c_prev, float *h, float *c, float *gate_act,
                           int M, int Hd, float forget_bias, void *stream);
/* dgates[M,4H], dc_prev[M,H] from dh[M,H] (+ dh2) and (optional) dc[M,H] flowing in from step t+1.              */
AIR_API int air_lstm_pointwise_bwd(const float *gate_act, const float *c_prev, const float *c, const float *dh,
                           const float *dh2 /* optional second dh term, summed */, const float *dc, float *dgates,
                           float *dc_prev, int M, int Hd, void *stream);

/* One LSTM time step with the gate math fused into the recurrent product (cell.py:126-127):
 *   gates = h_prev[M,Hd] . w_h[Hd,4Hd](ldw) + gx[M,4Hd](ldgx)      (gx = x.W_x + b, hoisted out of the time loop)
 *   then air_lstm_pointwise_fwd on `gates`; h, c [M,Hd] and gate_act [M,4Hd] are written, `gates` never exists.
 * precision: AIR_PREC_F32 / AIR_PREC_BF16 for the product.                                                          */
AIR_ENGINE_API int air_lstm_step_fwd(const float *h_prev, const float *c_prev, const float *w_h, int ldw, const float *gx, int ldgx,
                      float *h, float *c, float *gate_act, int M, int Hd, float forget_bias, int precision,
                      void *stream);
/* The first LSTM step of a train step with air_step_prologue riding along as extra workgroups: h0 / c0 [1,Hd] are read
 * with a broadcast row stride; the noise, the annealed prior and the tiled initial state (h_tiled, c_tiled [M,Hd]) are
 * written for the launches that follow.  Argument meaning as in air_lstm_step_fwd and air_step_prologue (B = M).        */
AIR_ENGINE_API int air_lstm_step_fwd_prologue(const float *h0, const float *c0, const float *w_h, int ldw, const float *gx, int ldgx,
                               float *h, float *c, float *gate_act, int M, int Hd, float forget_bias, int precision,
                               float *normal, size_t n_normal, float *uniform, size_t n_uniform,
                               const uint64_t *rng_state_dev, const int64_t *global_step_dev, int anneal_type,
                               double init, double final_value, double anneal_steps, double hold_for, double steps_div,
                               double *prior_out_f64, int T, float *h_tiled, float *c_tiled, void *stream);
/* air_lstm_step_fwd_prologue with the hoisted input product folded in (cell.py:121-127: the image never changes, so gx = x . W_x + b
 * is computed once): step 0's recurrent operand is the trainable initial state -- one row for the whole batch -- so this launch
 * accumulates x[M,E](ldx) . w_x[E,4Hd](ldw) and h0[1,Hd] . w_h[Hd,4Hd](ldw) side by side, writes gx_out[M,4Hd](ldgx) = x . w_x +
 * b_gates for the later steps and finishes step 0 on gx + h0 . w_h: the results of the gx launch + air_lstm_step_fwd_prologue it
 * replaces, bit for bit.  Latency regime only: AIR_E_UNSUPPORTED beyond 512 tiles of 16 x 16 over (M, Hd).                    */
AIR_ENGINE_API int air_lstm_first_step_fwd(const float *x, int ldx, int E, const float *w_x, const float *b_gates, const float *h0,
                            const float *c0, const float *w_h, int ldw, float *gx_out, int ldgx, float *h, float *c,
                            float *gate_act, int M, int Hd, float forget_bias, int precision, float *normal, size_t n_normal,
                            float *uniform, size_t n_uniform, const uint64_t *rng_state_dev, const int64_t *global_step_dev,
                            int anneal_type, double init, double final_value, double anneal_steps, double hold_for,
                            double steps_div, double *prior_out_f64, int T, float *h_tiled, float *c_tiled, void *stream);
/* One BPTT link: dh = dgates_next[M,4Hd] . w_h[Hd,4Hd]^T + dh_a + dh_b (either may be NULL), then
 * air_lstm_pointwise_bwd of the step that gate_act / c_prev / c belong to -> dgates[M,4Hd], dc_prev[M,Hd]; and, if
 * dgx_out != NULL, dgx_out = dgx_in + dgates (the running sum over time that the hoisted x.W_x product receives;
 * dgx_in may alias dgates_next or dgx_out, NULL = 0).                                                                */
AIR_ENGINE_API int air_lstm_step_bwd(const float *dgates_next, const float *w_h, const float *dh_a, const float *dh_b,
                      const float *dc_in, const float *gate_act, const float *c_prev, const float *c,
                      const float *dgx_in, float *dgates, float *dc_prev, float *dgx_out, int M, int Hd, int precision,
                      void *stream);
/* A slice [lo, hi) of the flat parameter / gradient / optimiser-state buffers to be updated by centred RMSProp (exactly
 * air_step_epilogue's arithmetic; elements >= n_model use lr * lr_mult_tail).  lo, hi, n_model multiples of 4, buffers 16-byte
 * aligned.  The *_opt entries below run the update as EXTRA workgroups of a backward launch that leaves most of the chip idle:
 * the caller guarantees that the slice's gradients are final and that no later launch of the backward reads its parameters.  */
typedef struct AirRmspropSlice {
    float *p; const float *g; float *ms, *mg, *mom;
    size_t lo, hi, n_model;
    const float *lr_dev;
    float lr_mult_tail, decay, momentum, eps, grad_scale;
} AirRmspropSlice;
/* air_gemm_grouped with the backward of a Gaussian head folded into the product that forms its sample gradient (latency regime):
 * problem `problem` of the group is dsample[M, D] = g . W^T (the decoder's first-layer dX for the `what` head, cell.py:154-158); the
 * thread that finishes element (m, d) writes dpre[m, d] and dpre[m, D + d] exactly as air_gauss_sample_bwd (loc_mode 0, one prior)
 * would from the stored value.  Two optional riders behind the tiles: air_nvil_parts (imp_parts == NULL: none) and the sum of the
 * head's KL shares (n_kl_parts == 0: none; kl_rows = T*B) -- together what the launch air_gauss_sample_bwd_nvil did.  Groups beyond
 * ~1000 16x16 tiles are declined (AIR_E_UNSUPPORTED).                                                                       */
typedef struct AirGaussBwdEpi {
    int problem;
    const float *pre; int ld_pre;
    const float *eps;
    float raw_offset, p_loc, p_scale;
    const float *loc, *scale, *dkl_row;
    float dkl_scale;
    float *dpre; int ld_dpre;
    int D;
    float guard_eps;
} AirGaussBwdEpi;
AIR_ENGINE_API int air_gemm_grouped_gauss_bwd(const AirGemmDesc *descs, int count, const AirGaussBwdEpi *epi, const float *imp_parts, int n_parts,
                               float *imp_sum, const float *baseline, const float *logp, float *nvil_out, float *dlogp,
                               float *dbaseline, int B, float *ema_dev, const float *kl_parts, int n_kl_parts, float *kl_row_out,
                               int kl_rows, void *stream);
/* air_gemm_grouped with the closing update of the train step folded in (single GPU, latency regime; model.py:355-367 + the weight
 * gradients of the first layers): the problems of `fold_mask` are plain weight gradients (ta = 1, beta = 0, no epilogue) whose C /
 * colsum point INTO the flat gradient buffer `g`; every element they finish is written to `g` as before and, in the same epilogue,
 * taken through centred RMSProp at the same flat offset of p / ms / mg / mom -- on one GPU a tile's gradient is final when formed.
 * Up to four further slices [range_lo, range_hi) (multiples of 4) whose gradients EARLIER launches left final are updated by rider
 * workgroups, one of which advances the device step counter and the Philox offset (as air_step_epilogue does).  The caller
 * guarantees that no problem of this launch reads a parameter it updates.  Groups the wide-tile dispatch of air_gemm_grouped would
 * take are declined with AIR_E_UNSUPPORTED.                                                                                        */
typedef struct AirOptFold {
    float *p; const float *g; float *ms, *mg, *mom;
    size_t n_model;
    const float *lr_dev;
    float lr_mult_tail, decay, momentum, eps, grad_scale;
    unsigned fold_mask;
    int n_ranges;
    size_t range_lo[4], range_hi[4];
    int64_t *global_step_dev; uint64_t *rng_state_dev; uint64_t rng_increment;
} AirOptFold;
AIR_ENGINE_API int air_gemm_grouped_opt(const AirGemmDesc *descs, int count, const AirOptFold *opt, void *stream);
/* air_lstm_step_bwd / air_lstm_pointwise_bwd with an optimiser slice riding along (opt == NULL or lo == hi: none).          */
AIR_ENGINE_API int air_lstm_step_bwd_opt(const float *dgates_next, const float *w_h, const float *dh_a, const float *dh_b,
                          const float *dc_in, const float *gate_act, const float *c_prev, const float *c,
                          const float *dgx_in, float *dgates, float *dc_prev, float *dgx_out, int M, int Hd, int precision,
                          const AirRmspropSlice *opt, void *stream);
/* The ENTRY of the BPTT and its first link in one launch (latency regime: at most 512 16x16 tiles of (batch, hidden), Hd % 16 == 0,
 * fp32 products; cell.py:126-127 backward, replaces air_lstm_pointwise_bwd(step T-1) + air_lstm_step_bwd(step T-2)): every
 * workgroup forms dgates_{T-1} -- the A operand of the link -- from the saved activations of step T-1 (gate_act1, c_prev1, c1 and
 * the direct terms dh_a1 / dh_b1, no dc flowing in), multiplies it with W_h^T and finishes step T-2's gate backward (dh_a / dh_b,
 * gate_act, c_prev, c -> dgates, dc_prev); dgates1 / dc_prev1 [M,4Hd] / [M,Hd] receive the entry's own results (the weight
 * gradient reads dgates1), dgx_out (optional) = dgates1 + dgates, the running sum over time.  All operands 16-byte aligned.
 * air_lstm_step_bwd_entry_fits(M, Hd) == 1 says whether the launch takes the shape.  opt: as air_lstm_step_bwd_opt.            */
AIR_ENGINE_API int air_lstm_step_bwd_entry_fits(int M, int Hd);
AIR_ENGINE_API int air_lstm_step_bwd_entry(const float *gate_act1, const float *c_prev1, const float *c1, const float *dh_a1,
                            const float *dh_b1, float *dgates1, float *dc_prev1, const float *w_h, const float *dh_a,
                            const float *dh_b, const float *gate_act, const float *c_prev, const float *c, float *dgates,
                            float *dc_prev, float *dgx_out, int M, int Hd, const AirRmspropSlice *opt, void *stream);
/* The recurrence on the bf16 DATA path (throughput regime: more than 512 16x16 tiles of (batch, hidden), Hd % 64 == 0): W_h is
 * read from the bf16 shadow of the parameters (w_h_bf16: same layout as w_h), h_prev / dgates_next from their bf16 mirrors when
 * given (NULL: the fp32 buffer, rounded in registers), products on v_mfma_f32_16x16x32_bf16; h_bf16 / dgates_bf16 / dgx_bf16
 * (optional) receive the mirrors of the outputs.  Otherwise exactly air_lstm_step_fwd / air_lstm_step_bwd / air_lstm_pointwise_bwd. */
AIR_ENGINE_API int air_lstm_step_fwd_bf16(const float *h_prev, const void *h_prev_bf16, const float *c_prev, const void *w_h_bf16, int ldw,
                           const float *gx, int ldgx, float *h, void *h_bf16, float *c, float *gate_act, int M, int Hd,
                           float forget_bias, void *stream);
AIR_ENGINE_API int air_lstm_step_bwd_bf16(const float *dgates_next, const void *dgates_next_bf16, const void *w_h_bf16, const float *dh_a,
                           const float *dh_b, const float *dc_in, const float *gate_act, const float *c_prev, const float *c,
                           const float *dgx_in, float *dgates, void *dgates_bf16, float *dc_prev, float *dgx_out,
                           void *dgx_bf16, int M, int Hd, void *stream);
AIR_ENGINE_API int air_lstm_pointwise_bwd_bf16(const float *gate_act, const float *c_prev, const float *c, const float *dh, const float *dh2,
                                const float *dc, float *dgates, void *dgates_bf16, float *dc_prev, int M, int Hd, void *stream);
AIR_ENGINE_API int air_lstm_pointwise_bwd_opt(const float *gate_act, const float *c_prev, const float *c, const float *dh, const float *dh2,
                               const float *dc, float *dgates, float *dc_prev, int M, int Hd, const AirRmspropSlice *opt,
                               void *stream);

/* ---- stochastic nodes ---------------------------------------------------------------------------------------*/

/* Reparameterised Gaussian + KL to a fixed Normal prior.  Replaces NormalWithSoftplusScale(...).sample() at
 * cell.py:130-133 / 154-156 (modules.py:17-24, 41-46, 58-63) and _kl(Normal, Normal) at model.py:174-209.
 *   pre[M,ld_pre]: columns [0,D) = loc pre-activation, [D,2D) = raw scale.
 *   loc_mode 0: loc = pre;  1: loc = [sigmoid,tanh,sigmoid,tanh,...] (TransformParam._transform).
 *   scale = softplus(raw + raw_offset); sample = loc + scale*eps.
 *   guard_eps (every entry point that samples a Gaussian head takes it as its last argument; 0 = off = the reference's
 *   arithmetic, inf / NaN placement included): > 0 floors the scale, scale = max(softplus(..), guard_eps), with no gradient
 *   through a floored scale, and in loc_mode 1 keeps the SAMPLED scale components (even dims) of `where` at |s| >= guard_eps
 *   (sign kept, straight-through) so that the inverse warp's 1/s (modules.py:101-102) never meets an exact zero -- the
 *   documented stability switch of SURVEY section 7 / App. B-11 (model.py:188-214, cell.py:130-133).
 *   kl_row[M] (optional) = sum_d KL(N(loc,scale) || N(p_loc[d&1], p_scale[d&1])); prior4 = {loc_even, scale_even,
 *   loc_odd, scale_odd} passed by value (where: even dims = scale prior, odd = shift prior; what: both equal).
 *   A prior location of NaN means "centred on the posterior's own mean" -- the where-shift prior given without `loc`,
 *   model.py:203-207: the (mu - p_loc)^2 term of the KL and its gradient vanish (every KL entry point honours it).  */
AIR_API int air_gauss_sample_fwd(const float *pre, int ld_pre, const float *eps, float raw_offset, int loc_mode,
                         float p_loc_even, float p_scale_even, float p_loc_odd, float p_scale_odd,
                         float *loc, float *scale, float *sample, float *kl_row, int M, int D, float guard_eps, void *stream);
/* dpre[M,ld_dpre] (both halves) from dsample[M,D] (+ dsample2[M,D]; either may be NULL) and dkl_row[M]*dkl_scale
 * (dkl_row may be NULL).                                                                                            */
AIR_API int air_gauss_sample_bwd(const float *pre, int ld_pre, const float *eps, float raw_offset, int loc_mode,
                         float p_loc_even, float p_scale_even, float p_loc_odd, float p_scale_odd,
                         const float *loc, const float *scale, const float *dsample, const float *dsample2,
                         const float *dkl_row, float dkl_scale, float *dpre, int ld_dpre, int M, int D, float guard_eps,
                         const float *kl_parts, int n_kl_parts, float *kl_row_out /* air_what_head_fwd's shares -> rows; 0: off */,
                         void *stream);
/* air_gauss_sample_bwd with air_nvil_parts riding as one extra workgroup (arguments of both, in that order; B = batch).  */
AIR_ENGINE_API int air_gauss_sample_bwd_nvil(const float *pre, int ld_pre, const float *eps, float raw_offset, int loc_mode,
                              float p_loc_even, float p_scale_even, float p_loc_odd, float p_scale_odd, const float *loc,
                              const float *scale, const float *dsample, const float *dsample2, const float *dkl_row,
                              float dkl_scale, float *dpre, int ld_dpre, int M, int D, const float *imp_parts, int n_parts,
                              float *imp_sum, const float *baseline, const float *logp, float *nvil_out, float *dlogp,
                              float *dbaseline, int B, float guard_eps, float *ema_dev, const float *kl_parts, int n_kl_parts,
                              float *kl_row_out, void *stream);

/* KL(N(loc,scale) || N(p_loc[d&1], p_scale[d&1])) summed over D per row, for given loc/scale tensors (model.py:
 * 174-209 evaluated on the cell's outputs).  kl_row[M].  Backward: dloc, dscale [M,D] from dkl_row[M].              */
AIR_API int air_normal_kl_fwd(const float *loc, const float *scale, float p_loc_even, float p_scale_even, float p_loc_odd,
                      float p_scale_odd, float *kl_row, int M, int D, void *stream);
AIR_API int air_normal_kl_bwd(const float *loc, const float *scale, float p_loc_even, float p_scale_even, float p_loc_odd,
                      float p_scale_odd, const float *dkl_row, float *dloc, float *dscale, int M, int D,
                      void *stream);

/* Presence, cell.py:137-151: p = sigmoid(logit + step_bias); if explore_eps >= 0: p = eps/2 + (1-eps)*p;
 * discrete: z = (u < p), presence[t] = presence[t-1]*z (presence[-1] = presence_in or 1); else presence = p.
 * logit/u/presence_prob/presence are [T,B] time-major (T = 1 for a single cell step).                              */
AIR_API int air_presence_fwd(const float *logit, const float *u, const float *presence_in, float step_bias,
                     float explore_eps, int discrete, float *presence_prob, float *presence, int T, int B,
                     void *stream);
/* dlogit[T,B] from dpresence_prob[T,B] (and dpresence[T,B] when !discrete; NULL otherwise).                        */
AIR_API int air_presence_bwd(const float *logit, float step_bias, float explore_eps, int discrete,
                     const float *dpresence_prob, const float *dpresence, float *dlogit, int T, int B, void *stream);

/* ---- objective ----------------------------------------------------------------------------------------------*/

/* Reconstruction term, model.py:319-324: per_sample[b] = sum_p 0.5*((obs-mult*canvas)/std)^2 + 0.5*log(2pi)+log(std) */
AIR_API int air_rec_loglik_fwd(const float *obs, const float *canvas, float mult, float std, float *per_sample,
                       int B, int P, void *stream);
/* dcanvas[b,p] = dper_sample[b] * mult*(mult*canvas-obs)/std^2  (dper_sample NULL => uniform `scale`)             */
AIR_API int air_rec_loglik_bwd(const float *obs, const float *canvas, float mult, float std, const float *dper_sample,
                       float scale, float *dcanvas, int B, int P, void *stream);

/* Number-of-steps posterior and its KL (prior.py:62-151, model.py:139-163), evaluated in float64 like the reference.
 *   presence_prob[T,B], presence[T,B] (sampled, cumulative); prior_f64[T+1] DEVICE doubles = geometric_prior(...).
 *   q[B,T+1]; kl_per_sample[B] = sum_n tabular_kl; logp[B] = log max(q[b, sum_t presence], 1e-32);
 *   step_weight[T,B] = sum_{n>t} q(n).                                                                             */
AIR_API int air_numsteps_fwd(const float *presence_prob, const float *presence, const double *prior_f64, float *q,
                     float *kl_per_sample, float *logp, float *step_weight, int T, int B, void *stream);
/* dpresence_prob[T,B] of  kl_scale*sum_b kl_per_sample[b] + sum_{t,b} dstep_weight[t,b]*step_weight[t,b]
 *                         + sum_b dlogp[b]*logp[b]   (dstep_weight / dlogp may be NULL).                           */
AIR_API int air_numsteps_bwd(const float *presence_prob, const float *presence, const double *prior_f64, float kl_scale,
                     const float *dstep_weight, const float *dlogp, float *dpresence_prob, int T, int B,
                     void *stream);

/* Engine fusions of the two above with presence (the step is launch bound at batch 64):
 *   air_presence_numsteps_fwd  = air_presence_fwd (discrete) + air_numsteps_fwd;
 *   air_numsteps_presence_bwd  = [dstep_weight = w_scale*(kl_row_a + kl_row_b)] + air_numsteps_bwd + air_presence_bwd,
 *                                producing d loss / d logit directly.
 * Continuous steps (AIRCell(discrete_steps=False), cell.py:150-151): every *_fwd entry that draws the presence takes u == NULL and
 * then writes presence = presence_prob (no Bernoulli chain); every *_bwd entry that forms dlogit takes `dpresence[T,B]` -- what the
 * canvas write's backward (air_canvas_unroll_bwd_dpresence) holds for the presence -- and adds it to d/d presence_prob (NULL: discrete). */
AIR_ENGINE_API int air_presence_numsteps_fwd(const float *logit, const float *u, float step_bias, float explore_eps,
                              const double *prior_f64, float *presence_prob, float *presence, float *q,
                              float *kl_per_sample, float *logp, float *step_weight, int T, int B, void *stream);
AIR_ENGINE_API int air_numsteps_presence_bwd(const float *presence_prob, const float *presence, const double *prior_f64,
                              float kl_scale, const float *kl_row_a, const float *kl_row_b, float w_scale,
                              const float *dlogp, const float *dpresence, const float *logit, float step_bias,
                              float explore_eps, float *dlogit, int T, int B, void *stream);

/* "Heads" launches: two independent small ops in ONE dispatch (blocks split by role).
 *   air_heads_fwd = air_gauss_sample_fwd (the where sample)  ||  air_presence_numsteps_fwd
 *   air_heads_bwd = air_gauss_sample_bwd (the where sample)  ||  air_numsteps_presence_bwd
 * Argument meaning exactly as in the four constituent entry points.                                                  */
AIR_ENGINE_API int air_heads_fwd(const float *pre, int ld_pre, const float *eps, float raw_offset, int loc_mode, float p_loc_even,
                  float p_scale_even, float p_loc_odd, float p_scale_odd, float *loc, float *scale, float *sample,
                  float *kl_row, int M, int D, const float *logit, const float *u, float step_bias, float explore_eps,
                  const double *prior_f64, float *presence_prob, float *presence, float *q, float *kl_per_sample,
                  float *logp, float *step_weight, int T, int B, float guard_eps, void *stream);
AIR_ENGINE_API int air_heads_bwd(const float *pre, int ld_pre, const float *eps, float raw_offset, int loc_mode, float p_loc_even,
                  float p_scale_even, float p_loc_odd, float p_scale_odd, const float *loc, const float *scale,
                  const float *dsample, const float *dsample2, const float *dkl_row, float dkl_scale, float *dpre,
                  int ld_dpre, int M, int D, const float *presence_prob, const float *presence,
                  const double *prior_f64, float kl_scale, const float *kl_row_a, const float *kl_row_b, float w_scale,
                  const float *dlogp, const float *dpresence, const float *logit, float step_bias, float explore_eps, float *dlogit,
                  int T, int B, float guard_eps, void *stream);

/* Annealed geometric prior over the number of steps, entirely on device (model.py:106-124,139-146; prior.py:26-32):
 *   step' = max(*global_step_dev - hold_for, 0);  anneal_type 0: s = init; 1 ("exp"): s = max(final, init *
 *   ((final/init)^(steps_div/anneal_steps))^(step'/steps_div)); 2 ("linear"): s = max(final, final + (init-final) *
 *   (1 - step'/anneal_steps)).  s is clipped to [1e-7, 1-1e-15]; prior_out_f64[n] = (1-s) s^n, n = 0..T (float64,
 *   NOT renormalised, exactly like the reference).  Reading the step counter on device keeps a captured hipGraph
 *   valid across replays; air_counter_add advances it.                                                             */
AIR_API int air_steps_prior(const int64_t *global_step_dev, int anneal_type, double init, double final_value,
                    double anneal_steps, double hold_for, double steps_div, double *prior_out_f64, int T,
                    void *stream);
AIR_API int air_counter_add(int64_t *counter_dev, int64_t increment, void *stream);

/* NVIL / REINFORCE with the reference's [B]-[B,1]->[B,B] broadcast (model.py:218-259).
 *   imp[B] (= rec_loss_per_sample), baseline[B], logp[B].
 *   out[4] = {reinforce_loss, baseline_loss, imp_weight_mean, imp_weight_var};
 *   dlogp[B] = d reinforce_loss / d logp; dbaseline[B] = d baseline_loss / d baseline.
 *   ema_dev (every entry point that evaluates NVIL takes it; NULL = decay_rate=None, the script's setting): a DEVICE block of four
 *   floats {moving_mean, moving_var, decay_rate, update} -- the EMA normalisation of model.py:232-239 / ops.py:46-64: the [B,B]
 *   weight is shifted by the moving mean and divided by max(sqrt(moving_var), 1) as the variables stand BEFORE this step, then --
 *   when update != 0 (train steps; evaluation passes read only) -- both move towards this batch's mean / variance (zero_debias
 *   off).  Kept on the device so that a captured graph carries the state from replay to replay.                        */
AIR_API int air_nvil(const float *imp, const float *baseline, const float *logp, float *out, float *dlogp,
             float *dbaseline, int B, float *ema_dev, void *stream);
/* The importance weight of a NON-analytic num-steps prior (model.py:157-163, 339-340: the step weights are the sampled presences
 * and reinforce_imp_weight += prior_loss.per_sample): rec_out[B] (optional) = sum of rec_parts[n_parts, B] in share order;
 * imp_out[b] = rec[b] + nsp_weight * kl_n[b] + sum_t step_weight[t,b] * (kl_row_a[t,b] + kl_row_b[t,b])   (kl_n / kl_row_* may be NULL).
 * dpresence_inout[T,B] (optional; continuous steps, where the step weight is the presence probability itself and carries a gradient):
 * += dkl_scale * (kl_row_a + kl_row_b).  One of imp_out / dpresence_inout may be NULL.                                       */
AIR_ENGINE_API int air_imp_weight(const float *rec_parts, int n_parts, float *rec_out, const float *kl_n, float nsp_weight, const float *kl_row_a,
                   const float *kl_row_b, const float *step_weight, int T, int B, float *imp_out, float *dpresence_inout,
                   float dkl_scale, void *stream);
/* air_nvil with the importance weight given as n_parts shares per sample (imp_parts[n_parts, B], added in share order in
 * fp32); the sum is also written to imp_sum[B] when given (the complete rec_loss_per_sample).                          */
AIR_ENGINE_API int air_nvil_parts(const float *imp_parts, int n_parts, float *imp_sum, const float *baseline, const float *logp,
                   float *out, float *dlogp, float *dbaseline, int B, float *ema_dev, void *stream);


/* Baseline input assembly, modules.py:131-139: out[B, HW + T*A + T*4 + T + S] =
 * [img | what (batch-major) | where | presence | state] from time-major what[T,B,A], where[T,B,4], presence[T,B],
 * state[B,S] = concat of up to two state parts (h, c).  HW may be 0 (img NULL): only the latent columns are packed
 * (the engine multiplies the image part of the first baseline layer straight from obs).                             */
AIR_API int air_baseline_pack(const float *img, const float *what, const float *where, const float *presence,
                      const float *state0, const float *state1, float *out, int T, int B, int P, int A, int S0,
                      int S1, void *stream);
/* air_gauss_sample_fwd for `what` (loc_mode 0, one prior) and the latent part of air_baseline_pack (HW = 0) in ONE launch:
 * pre[T*B, ld_pre] -> loc, scale, sample [T*B, D] time-major, kl_row[T*B]; pack_out[B, T*D + T*4 + T + S0 + S1] =
 * [what | where | presence | state0 | state1] batch-major (the sample is written to both places as it is drawn).      */
AIR_ENGINE_API int air_what_sample_pack(const float *pre, int ld_pre, const float *eps, float raw_offset, float p_loc, float p_scale,
                         float *loc, float *scale, float *sample, float *kl_row, int D, const float *where,
                         const float *presence, const float *state0, const float *state1, float *pack_out,
                         int T, int B, int S0, int S1, float guard_eps, void *stream);

/* The whole `what` head in ONE launch (modules.py:20-21 + cell.py:154-156 + the latent columns of modules.py:131-139): the
 * product q[T*B, 2A] = x[T*B, K](ldx) . w[K, 2A] + b, then -- in the tile that formed both halves of a (row, latent dim) pair --
 * loc, scale = softplus(raw + raw_offset), sample = loc + scale * eps (time-major [T*B, A] and batch-major into pack_out, as
 * air_what_sample_pack), while extra workgroups copy the where / presence / state columns of pack_out.  The KL row of a sample spans
 * air_what_head_parts(A) = ceil(A / 8) tiles: each writes its share to kl_parts[parts][T*B]; air_gauss_sample_bwd[_nvil] of the same
 * head adds them in tile order (kl_parts / n_kl_parts / kl_row_out).  Replaces a GEMM launch + air_what_sample_pack.            */
AIR_ENGINE_API int air_what_head_parts(int A);
AIR_ENGINE_API int air_what_head_fwd(const float *x, int ldx, int K, const float *w, const float *b, const float *eps, float raw_offset,
                      float p_loc, float p_scale, float *q, float *loc, float *scale, float *sample, float *kl_parts, int A,
                      const float *where, const float *presence, const float *state0, const float *state1, float *pack_out,
                      int T, int B, int S0, int S1, float guard_eps, int precision, void *stream);

/* ---- "attend": fused engine launches around the glimpse read (cell.py:129-151, modules.py:104-109) ------------------
 * Forward, one launch: the output layers of the transform MLP (tr_h[T*B,tr_k] . tr_w[tr_k,8] + tr_b -> pre[T*B,8]) and of the
 * steps predictor (st_h[T*B,st_k] . st_w[st_k,1] + st_b -> logit[T*B]), then everything air_heads_fwd computes from them
 * (where ~ N(loc, softplus(raw + raw_offset)), its KL rows, presence, q(n), KL, log q(n), step weights) and
 * air_st_read_fwd(img[B,H,W], where) -> glimpse[T*B,h,w] with `where` handed over inside the workgroup.
 * Needs H*W % 4 == 0, H*W <= 12288, 16-byte aligned img / tr_w (AIR_E_UNSUPPORTED otherwise: use the separate entries). */
AIR_ENGINE_API int air_attend_fwd(const float *tr_h, const float *tr_w, const float *tr_b, int tr_k, const float *st_h,
                   const float *st_w, const float *st_b, int st_k, float *pre, float *logit, const float *eps,
                   float raw_offset, float p_loc_even, float p_scale_even, float p_loc_odd, float p_scale_odd,
                   float *loc, float *scale, float *where, float *kl_row, const float *u, float step_bias,
                   float explore_eps, const double *prior_f64, float *presence_prob, float *presence, float *q,
                   float *kl_per_sample, float *logp, float *step_weight, const float *img, float *glimpse,
                   int T, int B, int H, int W, int h, int w, int precision /* of the two output-layer products */,
                   float guard_eps, void *stream);
/* Backward, one launch: air_st_read_bwd (d where through the read, one workgroup per glimpse) followed in the same
 * workgroup by the where-sampling backward of that row (dsample = dwhere_w + dwhere_r, KL term dkl_row*dkl_scale;
 * dwhere_w[dwhere_w_slabs][T*B][4]: the canvas backward's gradient as 1..4 partial slabs, added here in slab order) ->
 * dpre[T*B,8]; and, in separate workgroups, the steps-logit backward of air_heads_bwd -> dlogit[T*B].                */
AIR_ENGINE_API int air_attend_bwd(const float *img, const float *where, const float *dglimpse, float *dwhere_r, const float *pre,
                   const float *eps, float raw_offset, float p_loc_even, float p_scale_even, float p_loc_odd,
                   float p_scale_odd, const float *loc, const float *scale, const float *dwhere_w, int dwhere_w_slabs,
                   const float *dkl_row, float dkl_scale, float *dpre, const float *presence_prob,
                   const float *presence, const double *prior_f64, float kl_scale, const float *kl_row_a,
                   const float *kl_row_b, float w_scale, const float *dlogp, const float *dpresence, const float *logit, float step_bias,
                   float explore_eps, float *dlogit, int T, int B, int H, int W, int h, int w, float guard_eps, void *stream);
/* The same launch plus the dX of the two MLP OUTPUT layers (transform: [.., 8], steps: [.., 1]) whose dpre / dlogit it has just
 * formed -- the 8- and 1-deep products that otherwise need a launch of their own on the backward chain:
 *   tr_dx[k, n] = (sum_o dpre[k, o] * tr_w[n, o]) * elu'(tr_y[k, n]),  n < tr_k;   st_dx[k, n] = dlogit[k] * st_w[n] * elu'(st_y[k, n]).
 * tr_y / st_y: the layer's input activation (an ELU output) or NULL when the input is not an ELU output (no factor).
 * Their dW (and bias gradients) remain ordinary air_gemm problems over dpre / dlogit.                                        */
AIR_ENGINE_API int air_attend_bwd_dx(const float *img, const float *where, const float *dglimpse, float *dwhere_r, const float *pre,
                   const float *eps, float raw_offset, float p_loc_even, float p_scale_even, float p_loc_odd,
                   float p_scale_odd, const float *loc, const float *scale, const float *dwhere_w, int dwhere_w_slabs,
                   const float *dkl_row, float dkl_scale, float *dpre, const float *presence_prob,
                   const float *presence, const double *prior_f64, float kl_scale, const float *kl_row_a,
                   const float *kl_row_b, float w_scale, const float *dlogp, const float *dpresence, const float *logit, float step_bias,
                   float explore_eps, float *dlogit, int T, int B, int H, int W, int h, int w, const float *tr_w, const float *tr_y, float *tr_dx, int tr_k, int tr_ld,
                      const float *st_w, const float *st_y, float *st_dx, int st_k, int st_ld, int precision, float guard_eps,
                      void *stream);

/* L2 term of the objective (model.py:346-353: l2_weight * sum(w^2) / 2 over the 2-D model variables): g += l2_weight * p on up
 * to AIR_L2_MAX_RANGES slices [range_lo[k], range_hi[k]) of the flat buffers (host arrays; biases and baseline variables are not
 * in any slice).  Runs between the backward and the update.                                                            */
#define AIR_L2_MAX_RANGES 32
AIR_ENGINE_API int air_l2_grad_add(float *g, const float *p, const size_t *range_lo, const size_t *range_hi, int n_ranges, float l2_weight,
                    void *stream);

/* ---- optimiser ----------------------------------------------------------------------------------------------
 * TF centred RMSProp with momentum (model.py:265, 355-367): ms<-d*ms+(1-d)g^2; mg<-d*mg+(1-d)g;
 * mom<-m*mom + lr*g/sqrt(ms-mg^2+eps); p<-p-mom.  lr = *lr_dev * lr_mult (lr_dev: device float, graph-safe).
 * grad_scale multiplies g first (1/world_size after an all-reduce sum).                                            */
AIR_API int air_rmsprop_centered(float *p, const float *g, float *ms, float *mg, float *mom, size_t n, const float *lr_dev,
                         float lr_mult, float decay, float momentum, float eps, float grad_scale, void *stream);
/* The reference instantiates its optimiser as optimizer(learning_rate, **opt_kwargs) (model.py:265,355-363), so the keyword
 * set of tf.train.RMSPropOptimizer is part of the surface: decay, momentum, epsilon, centered.  centered = 0 drops the
 * squared mean-gradient term from the denominator (the mg slot is still maintained, as scratch).                       */
AIR_API int air_rmsprop(float *p, const float *g, float *ms, float *mg, float *mom, size_t n, const float *lr_dev, float lr_mult,
                float decay, float momentum, float eps, int centered, float grad_scale, void *stream);

/* Fused step prologue / epilogue for the launch-bound train step.
 *   prologue: air_rng_fill + air_steps_prior + tiling of the trainable LSTM initial state (h0,c0 [1,Hd] -> [B,Hd]).
 *   epilogue: centred RMSProp over the whole flat buffer (elements >= n_model use lr * lr_mult_tail: the baseline
 *             optimiser, model.py:363), then *global_step_dev += 1 and rng_state_dev[1] += rng_increment.            */
AIR_ENGINE_API int air_step_prologue(float *normal, size_t n_normal, float *uniform, size_t n_uniform, const uint64_t *rng_state_dev,
                      const int64_t *global_step_dev, int anneal_type, double init, double final_value,
                      double anneal_steps, double hold_for, double steps_div, double *prior_out_f64, int T,
                      const float *h0, const float *c0, float *h_tiled, float *c_tiled, int B, int Hd, void *stream);
/* air_step_prologue with air_f32_to_bf16(x -> x_bf16, n_x elements, n_x % 4 == 0) riding as extra workgroups (bf16 data path: the
 * observation batch's mirror is refreshed at the start of every step).                                                        */
AIR_ENGINE_API int air_step_prologue_cvt(float *normal, size_t n_normal, float *uniform, size_t n_uniform, const uint64_t *rng_state_dev,
                          const int64_t *global_step_dev, int anneal_type, double init, double final_value,
                          double anneal_steps, double hold_for, double steps_div, double *prior_out_f64, int T,
                          const float *h0, const float *c0, float *h_tiled, float *c_tiled, int B, int Hd, const float *x,
                          void *x_bf16, size_t n_x, void *stream);
/* ... and with the HBM feeder attached: row b of the batch is read from item idx_b of g->dataset (drawn as air_batch_gather draws it) and
 * written to g->obs (fp32) AND to x_bf16 -- air_batch_gather + air_step_prologue_cvt in one launch (g->copy_mask unused).          */
AIR_ENGINE_API int air_step_prologue_gather_cvt(float *normal, size_t n_normal, float *uniform, size_t n_uniform, const uint64_t *rng_state_dev,
                          const int64_t *global_step_dev, int anneal_type, double init, double final_value,
                          double anneal_steps, double hold_for, double steps_div, double *prior_out_f64, int T,
                          const float *h0, const float *c0, float *h_tiled, float *c_tiled, int B, int Hd, const AirBatchGather *g,
                          void *x_bf16, void *stream);
AIR_ENGINE_API int air_step_epilogue(float *p, const float *g, float *ms, float *mg, float *mom, size_t n_model, size_t n_total,
                      const float *lr_dev, float lr_mult_tail, float decay, float momentum, float eps, float grad_scale,
                      int64_t *global_step_dev, uint64_t *rng_state_dev, uint64_t rng_increment, void *stream);
/* air_step_epilogue + the bf16 shadow of the parameters (bf16 data path): p_bf16[i] = bf16(p[i]) for every updated element, so
 * the next step's dense products read half the weight bytes; NULL = no shadow.  air_f32_to_bf16: the same rounding as a launch of
 * its own, for buffers no kernel of this library produces (the observation batch, parameters after a load).               */
AIR_ENGINE_API int air_step_epilogue_shadow(float *p, const float *g, float *ms, float *mg, float *mom, size_t n_model, size_t n_total,
                             const float *lr_dev, float lr_mult_tail, float decay, float momentum, float eps, float grad_scale,
                             int64_t *global_step_dev, uint64_t *rng_state_dev, uint64_t rng_increment, void *p_bf16,
                             void *stream);
AIR_API int air_f32_to_bf16(const float *x, void *out_bf16, size_t n, void *stream);

/* HBM-resident batch feeder (replaces tensors_from_data's per-step tf.py_func round trip, data.py:121-158): out[b, :] =
 * dataset[idx_b, :] with idx_b drawn with replacement (shuffle != 0: Philox(seed_dev[0], stream 1, counter step*B + b), like
 * np.random.choice(n, batch_size)) or idx_b = (step*B + b) mod n_items; step = *step_dev, the DEVICE step counter that
 * air_step_epilogue advances, so the launch can be part of a captured step.  idx_out[B] (optional) receives the indices.     */
AIR_API int air_batch_gather(const float *dataset, long long n_items, int item_floats, const uint64_t *seed_dev,
                     const int64_t *step_dev, int shuffle, float *out, int B, int64_t *idx_out, void *stream);

/* ---- noise --------------------------------------------------------------------------------------------------
 * Philox4x32-10 counter RNG (replaces TF's sampler ops behind .sample(), cell.py:133,147,156).
 * normal[n_normal] ~ N(0,1), uniform[n_uniform] ~ U[0,1).  state_dev[2] = {seed, offset} DEVICE uint64; the
 * offset is advanced by air_rng_advance (separate launch, so a captured graph draws fresh noise per replay).      */
AIR_API int air_rng_fill(float *normal, size_t n_normal, float *uniform, size_t n_uniform, const uint64_t *state_dev,
                 void *stream);
AIR_API int air_rng_advance(uint64_t *state_dev, uint64_t increment, void *stream);

/* ---- small utilities ---------------------------------------------------------------------------------------*/
AIR_API int air_fill(float *p, size_t n, float v, void *stream);
/* out[m, n] = a[m, n] * row_scale[m]  (+ b[m,n] if b) : used for weighting per-row KL gradients etc.              */
AIR_API int air_axpby(const float *a, float alpha, const float *b, float beta, float *out, size_t n, void *stream);
/* broadcast rows: out[r, :] = src[0, :] for r < rows (tiling the trainable LSTM initial state, cell.py:103)       */
AIR_API int air_tile_rows(const float *src, float *out, int rows, int cols, void *stream);
AIR_API int air_colsum(const float *x, int ld, float *out, int M, int N, void *stream);   /* out[n] = sum_m x[m,n] */
/* out[i] = sum_t x[t*n + i], t < T: sums a time-major [T, n] stack over time (dGX = sum_t dgates_t in the LSTM BPTT) */
AIR_API int air_sum_leading(const float *x, float *out, int T, size_t n, void *stream);

/* ---- hipGraph capture + timing helpers (plumbing for bench / the fused train step) --------------------------*/
/* ---- data parallel: RCCL all-reduce of the flat gradient bucket, capturable into the step's hipGraph ----------------
 * The reference is single-process; SURVEY 8(e) shards the batch over one process per GPU with ONE all-reduce
 * (sum) of the flat fp32 gradient buffer per step.  RCCL is bound at run time (dlopen of the instance already in the
 * process).  air_comm_unique_id: 128 opaque bytes created on rank 0, handed to every rank's air_comm_init (a collective
 * call; the thread's current device is the rank's GPU).  air_allreduce_sum: in place on `stream`, no allocation / host
 * sync, legal inside a capture.  air_stream_wait_event = hipStreamWaitEvent (fork / join of a side stream in a capture).
 * Failures return AIR_E_UNSUPPORTED; air_comm_last_error() has the RCCL message.                                      */
AIR_API int air_comm_unique_id(void *id_out_128_bytes);
AIR_API int air_comm_init(void **comm_out, int world_size, int rank, const void *id_128_bytes);
AIR_API int air_comm_available(void);                     /* 0 if RCCL can be bound here; not collective (agree on it BEFORE air_comm_init) */
AIR_API int air_comm_count(void *comm, int *count_out);   /* ncclCommCount: the number of ranks RCCL itself sees */
AIR_API int air_comm_destroy(void *comm);
AIR_API int air_allreduce_sum(float *buf, size_t n, void *comm, void *stream);
AIR_API const char *air_comm_last_error(void);

/* Data-parallel update WITHOUT a library collective (csrc/comm_ipc.hip): the ranks of one node map each other's
 * flat gradient / parameter buffers and a block of flag words (hipIpc; the caller exchanges the handles) and every step runs
 *   air_dp_ipc_barrier(.., 0)  -- every rank's gradients final and written back --
 *   air_dp_ipc_rs_update_ag    -- rank r sums ITS 1/world shard of all ranks' gradients in rank order, scales by 1/world, runs
 *                                 centred RMSProp on that shard of its own p / ms / mg / mom (model.py:265,355-367) and writes the
 *                                 new parameters into EVERY rank's parameter buffer; the step counter / Philox offset advance --
 *   air_dp_ipc_barrier(.., 1)  -- every pushed parameter landed --
 * as kernel nodes of the step's graph.  One rank computes each element: replicas are bit-identical by construction.
 * flags[q]: rank q's block of 2 x 8 uint64 (zero-initialised); local_dev: 8 uint64 of this rank (zero-initialised; [4] / [5] =
 * bit mask of the hardware XCC ids the barrier's workgroups ran on); err_dev[0] becomes 1 when a peer did not arrive within the
 * (bounded) spin, 2 when the barrier's workgroups did not cover every XCD of the device.  A barrier is a grid of 64-thread
 * workgroups (64 of them by default, AIR_IPC_BARRIER_WGS; air_dp_ipc_barrier_wgs takes the count -- one count per rank and run:
 * instance k waits for k * count arrivals), each of which fences its own XCD's L2.                                              */
typedef struct AirIpcPeers {
    int world, rank;                   /* world <= 8 */
    const float *grads[8];
    float *params[8];
    uint64_t *flags[8];
} AirIpcPeers;
AIR_ENGINE_API int air_dp_ipc_barrier(const AirIpcPeers *peers, int which, uint64_t *local_dev, uint64_t *err_dev, void *stream);
AIR_ENGINE_API int air_dp_ipc_barrier_wgs(const AirIpcPeers *peers, int which, uint64_t *local_dev, uint64_t *err_dev, int n_wgs, void *stream);
/* The barrier's flag words in FINE-GRAINED device memory (hipExtMallocWithFlags(hipDeviceMallocFinegrained): in-kernel visibility of a
 * peer's store is only defined there), exported / opened with the HIP IPC calls; the handle is 64 bytes.  Not stream-ordered except
 * air_ipc_flags_zero.  distributed.IpcPeerBuffers falls back to ordinary allocations when any rank cannot use these.               */
AIR_ENGINE_API int air_ipc_flags_alloc(void **ptr_out, size_t bytes);
AIR_ENGINE_API int air_ipc_flags_free(void *ptr);
AIR_ENGINE_API int air_ipc_flags_zero(void *ptr, size_t bytes, void *stream);
AIR_ENGINE_API int air_ipc_handle_get(void *ptr, void *handle_64_bytes);
AIR_ENGINE_API int air_ipc_handle_open(const void *handle_64_bytes, void **ptr_out);
AIR_ENGINE_API int air_ipc_handle_close(void *ptr);
AIR_ENGINE_API int air_dp_ipc_rs_update_ag(const AirIpcPeers *peers, float *ms, float *mg, float *mom, size_t n_model, size_t n_total,
                            const float *lr_dev, float lr_mult_tail, float decay, float momentum, float eps,
                            int64_t *global_step_dev, uint64_t *rng_state_dev, uint64_t rng_increment, void *stream);
AIR_API int air_stream_wait_event(void *stream, void *event);

AIR_API int air_graph_begin_capture(void *stream);
AIR_API int air_graph_end_capture(void *stream, void **graph_exec_out);
AIR_API int air_graph_launch(void *graph_exec, void *stream);
AIR_API int air_graph_destroy(void *graph_exec);
AIR_API int air_event_create(void **event_out);
AIR_API int air_event_record(void *event, void *stream);
AIR_API int air_event_elapsed_ms(void *start, void *stop, float *ms_host_out);   /* synchronises on `stop` */
AIR_API int air_event_destroy(void *event);

#ifdef __cplusplus
}
#endif
#endif /* AIR_HIP_H */
