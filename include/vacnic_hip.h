/*
 * vacnic_hip.h — C-ABI of libvacnic_hip.so: the hand-written gfx950 (CDNA4) kernels behind the
 * VACNIC training step.
 *
 * The reference (tingyu215/VACNIC) is pure Python/PyTorch and has no FFI layer; its "plugin
 * boundary" for this path is the torch.nn operator surface of `src/models` plus three trainer
 * helpers.  Every entry point below cites the reference op sequence it replaces
 * (MFULL = src/models/modeling_mmbart_clip_inside_vis_clipcap_ent_type_final_fix_len_enc_self_face_name_ids_crossattn.py,
 *  TRAIN = train_mmbart_enc_self_face_name_ids_retrieve_crossattn_bart_guide_match.py).
 * The Python binding a maintainer would add is the ctypes stub shown in INTEGRATION.md and
 * shipped as vacnic_amd/_lib.py.
 *
 * Conventions
 *  - plain pointers + sizes, no torch types.  All pointers are DEVICE pointers owned by the caller
 *    (PyTorch-ROCm allocator) and borrowed for the stream lifetime of the call.
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*); no hidden syncs,
 *    no device allocation inside the library.
 *  - return value: 0 = ok, otherwise a vacnic_status; vacnic_last_error_string() has the text.
 *  - bf16 tensors are raw uint16 storage ("bf16"), fp32 tensors "f32"; row strides ("ld*") are in
 *    elements.  bf16 row strides must be multiples of 8 (16-byte rows); a K-contiguous GEMM operand
 *    whose K is not a multiple of 8 must have ld >= round_up(K, 8) with finite padding (zero on one side).
 */
#ifndef VACNIC_HIP_H
#define VACNIC_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  VACNIC_OK = 0,
  VACNIC_BAD_SHAPE = 1,
  VACNIC_BAD_DTYPE = 2,
  VACNIC_MISALIGNED = 3,
  VACNIC_HIP_ERROR = 4,
  VACNIC_UNSUPPORTED = 5
} vacnic_status;

const char* vacnic_last_error_string(void);
int vacnic_version(void);

/* ---- activation / epilogue codes ------------------------------------------------------------ */
enum { VACNIC_ACT_NONE = 0, VACNIC_ACT_GELU = 1, VACNIC_ACT_TANH = 2, VACNIC_ACT_QUICKGELU = 3 };

/*
 * GEMM on the MFMA matrix cores (bf16 in, fp32 accumulate).
 *   out[m][n] = epi( alpha * sum_k X(m,k) * W(n,k) + bias[n] ) (+ residual[m][n])
 * x_kstrided = 0: X stored [M][K] (X(m,k) = x[m*ldx + k]);  1: stored [K][M] (x[k*ldx + m]).
 * w_kstrided = 0: W stored [N][K] (torch nn.Linear weight); 1: stored [K][N].
 *   forward  nn.Linear (MFULL:467-483,738-741 q/k/v/out_proj, fc1/fc2 ...): x_k=0, w_k=0
 *   dgrad    dX = dY . W          : X=dY, W=weight viewed [K_red=N][N_out=K]   -> w_kstrided=1
 *   wgrad    dW = dY^T . X        : both operands reduction-strided            -> 1,1 (out_f32_atomic)
 * epilogue:
 *   act            VACNIC_ACT_* applied after bias.
 *   preact         optional bf16 [M][ldo]: receives the pre-activation (saved for backward).
 *   dact_src       optional bf16 [M][ldo]: if non-null, result is multiplied by act'(dact_src)
 *                  (fused activation backward in the dgrad of the following Linear).
 *   residual       optional bf16 [M][ldo] added after the activation.
 *   out_mode       0: bf16 store, 1: f32 store, 2: f32 atomic accumulate (out += ..; split-K ok)
 *   split_k        >=1; >1 requires out_mode 2 or a workspace.
 *   workspace      optional: with split_k > 1 the K slices meet through an ORDERED FIX-UP instead of fp32 atomics — every slice
 *                  deposits its fp32 partial tile here, the last one to arrive sums them in slice order and runs the epilogue
 *                  once — so any out_mode / activation / residual works with split_k > 1 and the result is bitwise
 *                  reproducible.  Needs vacnic_gemm_workspace_bytes(M, N, split_k) bytes (any contents) and `counters`:
 *                  vacnic_gemm_counters(M, N) uint32 words that are ZERO before the call (they are zero again after it).
 *                  Launches that may run concurrently (different streams) must not share either buffer.
 * With out_mode 0 and 256-row tiles the result is rounded to bf16 once (bias and a plain activation are applied in
 * fp32 first); a saved pre-activation, the fused activation backward and the residual are then applied to that bf16
 * value — the arithmetic of a bf16 Linear followed by a bf16 elementwise op.
 */
typedef struct {
  const void* x; const void* w; const float* bias;
  void* out; void* preact; const void* dact_src; const void* residual;
  float* xsum;                    /* optional f32 [M]: xsum[m] += sum_k X(m,k), not scaled by alpha — the bias gradient
                                     (column sums of dY) fused into the weight-gradient GEMM that stages dY anyway
                                     (replaces the separate reduction autograd runs for nn.Linear.bias, MFULL:449-452).
                                     Implemented for x_kstrided && w_kstrided (the weight-gradient layout) only; any
                                     other layout returns VACNIC_UNSUPPORTED */
  int64_t M, N, K;
  int64_t ldx, ldw, ldo;
  int32_t x_kstrided, w_kstrided;
  int32_t act, out_mode, split_k;
  float alpha;
  int32_t tile_hint;              /* 0 = library picks by problem size (M <= 8 rows: the W-streaming skinny kernel); 8 (skinny) / 64 (64x128, 4-deep ring) / 128 / 256 force a config */
  void* workspace;                /* split-K fix-up (see above); NULL = fp32 atomics */
  int64_t workspace_bytes;
  uint32_t* counters;
  int64_t counters_len;           /* words */
  /* activation dropout fused into the epilogue — nn.functional.dropout(act(fc1 x), p=activation_dropout) of the FFN blocks
     (MFULL:649,660,684,740,874) in the forward GEMM (applied after the activation) and the same mask on the gradient in the
     dgrad GEMM that carries dact_src (applied after act'): drop_p in (0, 1), quantised to 1/256; the mask is the function of
     (drop_seed ^ f(*drop_seed_dev), element index m * N + n) that vacnic_dropout_bf16 evaluates, so nothing is stored.
     Needs out_mode 0, ldo == N and N % 16 == 0.  drop_p = 0: off. */
  float drop_p;
  uint64_t drop_seed;
  const uint64_t* drop_seed_dev;
} vacnic_gemm_args;
int vacnic_gemm_bf16(const vacnic_gemm_args* a, void* stream);
/* sizes of the split-K fix-up buffers for an M x N output (upper bounds over every tile configuration the library may pick) */
int64_t vacnic_gemm_workspace_bytes(int64_t M, int64_t N, int64_t split_k);
int64_t vacnic_gemm_counters(int64_t M, int64_t N);

/*
 * Single-token decoder step: y = epi( LayerNorm(x + residual) . W^T + bias ), M <= 8 rows (beams x batch), K = d_model <= 1024.
 * The post-LN of one decoder sub-block (MFULL:836,859,880: residual add + LayerNorm, eval mode: no dropout) fused as a prologue
 * into the first projection of the next sub-block (q/k/v, cross-attention q, fc1, lm_head); ln_out (bf16 [M][K], optional)
 * receives the normalised rows — the next sub-block's residual.  Bit-identical to vacnic_add_ln_fwd followed by
 * vacnic_gemm_bf16 (skinny kernel).  x / residual rows are contiguous (stride K); act as in vacnic_gemm_bf16; out_mode 0 bf16, 1 f32.
 */
typedef struct {
  const void* x; const void* residual; const float* gamma; const float* beta; void* ln_out;
  const void* w; const float* bias; void* out;
  int64_t M, N, K, ldw, ldo;
  int32_t act, out_mode;
  float eps;
} vacnic_gemv_ln_args;
int vacnic_gemv_ln_bf16(const vacnic_gemv_ln_args* a, void* stream);

/* Grouped weight gradients — nn.Linear backward w.r.t. weight and bias (autograd of MFULL:449-452 and every other Linear of
 * the path) for SEVERAL layers in one launch:
 *     dw[N, K] += dy[M, N]^T . x[M, K]          dbias[N] += column sums of dy   (dbias may be NULL)
 * dy, x bf16 row-major (rows 16-byte aligned: ld % 8 == 0), dw / dbias fp32, accumulated in place.  No split-K and no atomics
 * on dw: every output element has one writer and a fixed summation order (bitwise reproducible).  Each 1024 x 1024 block of
 * a dw runs with its full reduction on one XCD; jobs of one call should share M (they finish together).  The torch reference
 * runs one cuBLAS GEMM + one reduction per Linear. */
typedef struct {
  const void* dy; const void* x;
  float* dw; float* dbias;
  int64_t M, N, K;
  int64_t lddy, ldx, lddw;
} vacnic_wgrad_job;
int vacnic_wgrad_group(const vacnic_wgrad_job* jobs, int64_t njobs, void* stream);

/*
 * Fused attention core (replaces bmm -> +mask -> softmax -> bmm of BartAttention.forward,
 * MFULL:509-548, and nn.MultiheadAttention inside the CLIP ViT).  head_dim must be 64.
 *   q: [B][Tq][..] bf16, row stride ldq, head h at column offset h*64; likewise k, v (rows Tk).
 *   out: [B][Tq][H*64] bf16 (row stride ldo) — heads merged, ready for out_proj.
 *   key_mask: optional uint8 [B][Tk]; 0 => additive finfo(float32).min exactly as _expand_mask
 *             (MFULL:387-398) does (an all-masked row therefore softmaxes uniformly, like torch).
 *   causal:   1 => key j masked for query i when j > i (MFULL:373-385).
 *   scale:    multiplies q.k (reference multiplies q by head_dim**-0.5, MFULL:471).
 *   lse:      f32 [B][H][Tq] log-sum-exp of the scaled+masked scores (saved for backward).
 */
typedef struct {
  const void* q; const void* k; const void* v; void* out; float* lse;
  const uint8_t* key_mask;
  int64_t B, H, Tq, Tk;
  int64_t ldq, ldk, ldv, ldo;
  int64_t bsq, bsk, bsv, bso;     /* batch strides in elements */
  int32_t causal; float scale;
  /* attention-probability dropout, nn.functional.dropout(attn_weights, p=attention_dropout) of MFULL:546 (0.1 in the bart-base / bart-large hub configs):
     p_drop in [0,1) quantised to 1/256; Philox keep bits from (seed ^ f(*seed_dev), batch, head, query, key), regenerated by
     vacnic_attn_bwd from the same seed — the [B*H, Tq, Tk] mask is never stored.  lse stays the undropped log-sum-exp. */
  float p_drop; uint64_t seed; const uint64_t* seed_dev;
} vacnic_attn_fwd_args;
int vacnic_attn_fwd(const vacnic_attn_fwd_args* a, void* stream);

typedef struct {
  const void* q; const void* k; const void* v; const void* out; const void* dout;
  const float* lse; float* delta;            /* delta: f32 [B][H][Tq] scratch: sum_k P[q][k] dP[q][k], formed by a sweep of the kernels' own
                                                P and dP for Tq <= 128 and as rowsum(dO*O) above (VACNIC_ATTN_DELTA=1 / 2 force one form) */
  void* dq; void* dk; void* dv;              /* bf16, same layouts/strides as q,k,v */
  const uint8_t* key_mask;
  int64_t B, H, Tq, Tk;
  int64_t ldq, ldk, ldv, ldo;
  int64_t bsq, bsk, bsv, bso;
  int64_t lddq, lddk, lddv, bsdq, bsdk, bsdv;
  int32_t causal; float scale;
  float p_drop; uint64_t seed; const uint64_t* seed_dev;      /* as in the forward call */
} vacnic_attn_bwd_args;
int vacnic_attn_bwd(const vacnic_attn_bwd_args* a, void* stream);

/*
 * out = LayerNorm(residual + dropout(x)) * gamma + beta   (post-LN blocks, MFULL:651-653,705-707,
 * 721-723,742-744; nn.LayerNorm eps=1e-5).  x/residual/out bf16 [R][D]; residual may be NULL
 * (plain LN: ViT ln_pre/ln_1/ln_2/ln_post, ner_map_layer_norm).  Dropout is Philox keyed on
 * (seed, row*D+col); p = 0 disables it.  mean/rstd f32 [R] are saved for backward.
 */
typedef struct {
  const void* x; const void* residual; const float* gamma; const float* beta;
  void* out; float* mean; float* rstd;
  int64_t R, D; float eps; float p_drop; uint64_t seed;
  const uint64_t* seed_dev;      /* optional device counter mixed into the seed (fresh masks under hipGraph replay) */
} vacnic_add_ln_fwd_args;
int vacnic_add_ln_fwd(const vacnic_add_ln_fwd_args* a, void* stream);
/* out[i] = x[i] * keep_i / (1 - p) on a flat bf16 array (in place allowed; n % 8 == 0): nn.functional.dropout(act(fc1 x),
 * p=activation_dropout) of the FFN blocks (MFULL:649,660,684,740,874) as a stand-alone pass — the same mask the GEMM epilogues
 * apply when vacnic_gemm_args.drop_p is set: p quantised to 1/256, element i keeps iff byte i % 16 of Philox block i / 16 of
 * (seed ^ f(*seed_dev)) is >= round(256 p).  Backward calls it on the gradient with the same seed; nothing is stored. */
int vacnic_dropout_bf16(const void* x, void* out, int64_t n, float p_drop, uint64_t seed, const uint64_t* seed_dev, void* stream);

/* backward: recomputes h = residual + dropout(x) from the saved INPUTS (x, residual) and the saved
 * mean/rstd; writes dresidual (= d h) and dx (= d h * keep/(1-p)); either may be NULL.  With
 * p_drop = 0 the two are identical, pass one.  dgamma/dbeta are accumulated with f32 atomics. */
typedef struct {
  const void* dout; const void* x; const void* residual; const float* gamma;
  const float* mean; const float* rstd;
  void* dresidual; void* dx; float* dgamma; float* dbeta;
  int64_t R, D; float p_drop; uint64_t seed; const uint64_t* seed_dev;
  float* partials; int64_t partial_rows;   /* optional caller-owned scratch f32 [partial_rows][2][D] (partial_rows >= min(1024,
                                              ceil(R/4))): dgamma/dbeta are then reduced in two stages (per-block column sums,
                                              one small fold) instead of ~1000 same-address atomics per column */
  int32_t defer_fold;                      /* 1 (with partials): leave the fold to the caller — vacnic_ln_partial_fold on a stream of
                                              its choice (the parameter gradients are needed only by AdamW / the reducer, so the fold
                                              need not sit in the backward chain) */
} vacnic_add_ln_bwd_args;
int vacnic_add_ln_bwd(const vacnic_add_ln_bwd_args* a, void* stream);
/* dgamma[D] += sum over rows of partials[row][0][:], dbeta[D] += ... [row][1][:]  (partials f32 [rows][2][D], as written by
 * vacnic_add_ln_bwd with defer_fold = 1 for rows = min(1024, ceil(R / 4))) */
int vacnic_ln_partial_fold(const float* partials, float* dgamma, float* dbeta, int64_t rows, int64_t D, void* stream);

/*
 * out = dropout(LayerNorm(embed[ids]*scale + pos[t + 2]))  — BartEncoder/BartDecoder prologue,
 * MFULL:1243-1249,1254-1260,1553-1562; BartLearnedPositionalEmbedding offset 2 (MFULL:401-418).
 * ids int64 [B][T]; embed f32 or bf16 master table [V][D] (we read the bf16 shadow); pos bf16.
 */
typedef struct {
  const int64_t* ids; const void* embed; const void* pos; const float* gamma; const float* beta;
  void* out; float* mean; float* rstd;
  int64_t B, T, D, V; int64_t pos_offset; float embed_scale; float eps; float p_drop; uint64_t seed;
  const uint64_t* seed_dev;
} vacnic_embed_ln_fwd_args;
int vacnic_embed_ln_fwd(const vacnic_embed_ln_fwd_args* a, void* stream);

typedef struct {
  const int64_t* ids; const void* embed; const void* pos; const void* dout; const float* gamma;
  const float* mean; const float* rstd;
  float* dembed; float* dpos; float* dgamma; float* dbeta;   /* f32 accumulate (atomics); any may be NULL */
  int64_t B, T, D, V; int64_t pos_offset; float embed_scale;
  int64_t padding_idx;                        /* rows with ids == padding_idx get no embedding grad (nn.Embedding) */
  float p_drop; uint64_t seed; const uint64_t* seed_dev;
} vacnic_embed_ln_bwd_args;
int vacnic_embed_ln_bwd(const vacnic_embed_ln_bwd_args* a, void* stream);

/*
 * Cross-entropy over materialised logits (CrossEntropyLoss(ignore_index=pad), TRAIN:287,816).
 * logits bf16 or f32 [R][ldl] (first V columns valid); targets int64 [R].
 * vacnic_ce_fwd writes row_lse [R] (and row_loss if non-NULL) and ACCUMULATES *loss_sum / *count
 * (f32 atomics; caller zeroes them) so that loss = loss_sum / count.
 * vacnic_ce_bwd writes dlogits bf16 [R][ldd] = (softmax - onehot) * valid * grad_scale * (*grad_out) / (*count);
 * columns [V, ldd) are zero-filled.  dlogits may alias bf16 logits when ldd == ldl.
 */
typedef struct {
  const void* logits; const int64_t* targets; float* row_lse; float* row_loss; float* loss_sum; float* count;
  void* dlogits; const float* grad_out; float grad_scale;
  int64_t R, V, ldl, ldd; int64_t ignore_index; int32_t logits_f32;
} vacnic_ce_args;
int vacnic_ce_fwd(const vacnic_ce_args* a, void* stream);
int vacnic_ce_bwd(const vacnic_ce_args* a, void* stream);
/* out4 = {total, txt, secla, colam}; total = ce_sum/count + w_secla*secla + w_colam*colam (TRAIN:363).
 * secla / colam may be NULL (treated as 0). */
int vacnic_combine_losses(const float* ce_sum, const float* count, const float* secla, const float* colam,
                          float w_secla, float w_colam, float* out4, void* stream);

/*
 * CoLaM margin loss (TRAIN:296-307,178-182,820): pool(masked mean over T with the LABEL mask,
 * nan_to_num(nan=1)) -> L2 normalise -> cos_i = <a_i, b_i> -> mean_i max(0, margin - cos_i).
 * hs (student, trainable decoder) and hg (guide) bf16 [B][T][D]; mask uint8 [B][T].
 * fwd writes loss (f32 scalar, overwritten) and saves pooled/normalised rows for bwd.
 */
typedef struct {
  const void* hs; const void* hg; const uint8_t* mask;
  float* loss; float* cos; float* pooled_s; float* pooled_g;   /* cos [B]; pooled_* f32 [B][D] (raw pooled, pre-norm) */
  int64_t B, T, D; float margin;
} vacnic_colam_fwd_args;
int vacnic_colam_fwd(const vacnic_colam_fwd_args* a, void* stream);
typedef struct {
  const float* cos; const float* pooled_s; const float* pooled_g; const uint8_t* mask;
  void* dhs;                                  /* bf16 [B][T][D], overwritten */
  int64_t B, T, D; float margin; const float* grad_out; float grad_scale; /* d loss = grad_scale * (*grad_out) */
} vacnic_colam_bwd_args;
int vacnic_colam_bwd(const vacnic_colam_bwd_args* a, void* stream);

/*
 * SECLA face-name loss (BatchSoftmax, TRAIN:631-660): faces f32/bf16 [B][F][D], names f32 [B][N][D].
 *   M1[i][j][n][f] = <name_{i,n}, face_{j,f}>; L1 = CE_i( sum_n max_f M1 / N , target i )
 *   M2[i][j][f][n] = <face_{i,f}, name_{j,n}>; L2 = CE_i( sum_f max_n M2 / F , target i )
 * loss = L1 + L2.  sim f32 [B][N][B][F] scratch holds <name_{i,n}, face_{j,f}> for bwd.
 */
typedef struct {
  const void* faces; const float* names; float* sim; float* logits1; float* logits2; float* loss;
  int64_t B, F, N, D;
} vacnic_secla_fwd_args;
int vacnic_secla_fwd(const vacnic_secla_fwd_args* a, void* stream);
typedef struct {
  const void* faces; const float* names; const float* sim; const float* logits1; const float* logits2;
  void* dfaces;                               /* bf16 [B][F][D], overwritten */
  float* wsim;                                /* f32 [B][N][B][F] scratch: d loss / d sim */
  int64_t B, F, N, D; const float* grad_out; float grad_scale;
} vacnic_secla_bwd_args;
int vacnic_secla_bwd(const vacnic_secla_bwd_args* a, void* stream);

/* Per-name mean of LN(embed_ner(ids)*scale + pos) — get_embedding_ner, TRAIN:112-133 (unmasked
 * mean over the Ln tokens, pads included).  ids int64 [B][Nn][Ln] -> out f32 [B][Nn][D]. */
typedef struct {
  const int64_t* ids; const void* embed; const void* pos; const float* gamma; const float* beta;
  float* out; int64_t B, Nn, Ln, D, V; int64_t pos_offset; float embed_scale; float eps;
} vacnic_name_embed_args;
int vacnic_name_embed_mean(const vacnic_name_embed_args* a, void* stream);

/*
 * Fused AdamW over one flat fp32 arena (torch.optim.AdamW semantics, TRAIN:91):
 *   p *= (1 - lr*wd); m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 *   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * lr, step are read from device memory (graph-capture safe): hyper = {lr, step(as float)}.
 * Also refreshes the bf16 shadow and zeroes the gradient.  grad_scale multiplies g first
 * (1/world_size for DDP averaging).  clip_coef (may be NULL): device pointer to the clip_grad_norm_
 * coefficient written by vacnic_grad_clip_coef; it multiplies g after grad_scale.
 */
typedef struct {
  float* p; float* g; float* m; float* v; void* p_bf16; const float* hyper;
  int64_t n; float beta1, beta2, eps, weight_decay, grad_scale; int32_t zero_grad;
  const float* clip_coef;
} vacnic_adamw_args;
int vacnic_adamw(const vacnic_adamw_args* a, void* stream);
/* torch.nn.utils.clip_grad_norm_(model.parameters(), clip_norm) (TRAIN:365-366) over the flat gradient arena,
 * without a host sync: out[0] <- min(1, max_norm / (||grad_scale*g||_2 + 1e-6)), out[1] <- that norm.
 * partials: device scratch of >= 1024 floats.  Deterministic (fixed grid, no float atomics).  The gradient is not
 * rewritten; pass `out` as vacnic_adamw_args.clip_coef and the scaling happens where AdamW reads g. */
int vacnic_grad_clip_coef(const float* g, int64_t n, float grad_scale, float max_norm, float* partials,
                          float* out, void* stream);
/* get_linear_schedule_with_warmup on device (TRAIN:99-107): hyper[0] <- base_lr*lambda(k), hyper[1] <- k+1
 * where k = hyper[1] on entry = optimizer steps already taken.  Also increments *rng_counter (the device-side
 * dropout counter, may be NULL).  Call once before vacnic_adamw. */
int vacnic_lr_step(float* hyper, float base_lr, float warmup_steps, float total_steps, uint64_t* rng_counter, void* stream);

/* ---- small data-movement helpers on the path ------------------------------------------------- */
/* f32 -> bf16 cast (weight shadow refresh, inputs). */
int vacnic_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream);
int vacnic_cast_bf16_f32(const void* src, float* dst, int64_t n, void* stream);
/* strided 2-D bf16 copy: dst[r*ldd + c] = src[r*lds + c], used for torch.cat along tokens
 * (MFULL:666,691) without materialising intermediates; accumulate=1 adds instead (cat backward). */
int vacnic_copy2d_bf16(const void* src, void* dst, int64_t rows, int64_t cols, int64_t lds, int64_t ldd,
                       int32_t accumulate, void* stream);
/* 3-D variant: [B][rows][cols] with batch strides. */
int vacnic_copy3d_bf16(const void* src, void* dst, int64_t B, int64_t rows, int64_t cols,
                       int64_t lds, int64_t ldd, int64_t bss, int64_t bsd, int32_t accumulate, void* stream);
/* dst[R][Cp] = src[R][C] zero-padded on the right (C not a multiple of 8: the 20-wide name-prefix FFN output,
 * MFULL:595) so the following GEMMs see 16-byte rows. */
int vacnic_pad_cols_bf16(const void* src, void* dst, int64_t R, int64_t C, int64_t Cp, int64_t lds, void* stream);
/* CLIP patch embedding im2col (conv1 with kernel = stride = patch, no bias; TRAIN:225-227):
 * img f32 [B][3][HW][HW] -> patches bf16 [B*g*g][Kp] (k = c*p*p + py*p + px, zero-padded to Kp). */
int vacnic_im2col_patches(const float* img, void* patches, int64_t B, int64_t HW, int64_t patch, int64_t Kp,
                          void* stream);
/* x[b][0] = cls + pos[0]; x[b][1+i] = patch_emb[b][i] + pos[1+i]  (TRAIN:228-229), bf16 out. */
int vacnic_vit_assemble(const void* patch_emb, const void* cls, const void* pos, void* out,
                        int64_t B, int64_t G2, int64_t W, void* stream);
/* int64 ids != pad -> uint8 mask, plus shift_tokens_right (TRAIN:196-217) in one launch. */
int vacnic_prep_ids(const int64_t* ids, uint8_t* mask, int64_t* shifted, int64_t B, int64_t T,
                    int64_t pad_id, int64_t start_id, void* stream);
/* face mask = (face_emb[:, :, -1] != 1) as uint8 (TRAIN:269; pad faces are all-ones rows, DSG:48,124). */
int vacnic_face_mask(const float* faces, uint8_t* mask, int64_t BF, int64_t D, void* stream);
/* out[B, na+nb] = cat(a[B,na], b[B,nb], dim 1) on bytes: fm = torch.cat((face_mask, name_mask), dim=1), MFULL:1262 */
int vacnic_cat2_u8(const uint8_t* a, const uint8_t* b, uint8_t* out, int64_t B, int64_t na, int64_t nb, void* stream);
/* per-row argmax over V logits (greedy decode / eval_epoch TRAIN:424): lowest index wins ties. */
int vacnic_argmax_rows(const void* logits, int64_t* out, int64_t R, int64_t V, int64_t ldl,
                       int32_t logits_f32, void* stream);
/* sum of bias gradient: dbias[n] += sum_m dy[m][n]  (bf16 dy, f32 atomics). */
int vacnic_bias_grad(const void* dy, float* dbias, int64_t M, int64_t N, int64_t ldy, void* stream);
/* out = a + b (bf16) — gradient fan-in where a tensor feeds two consumers */
int vacnic_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream);

/* ---- decode (transformers 4.18 GenerationMixin.beam_search driven at TRAIN:513-520, DDPINF:758-842) ---- */
/* Per beam row: log_softmax(logits[:V]) -> NoRepeatNGram bans (bans int32 [R][n_ban], -1 = none) -> MinLength
 * (suppress_eos) -> ForcedEOS (forced_token >= 0: that token scores 0, all others -inf) -> + beam_scores[r] ->
 * top-K (sorted, ties broken towards the lower token id).  top_val f32 [R][K], top_idx int32 [R][K]. */
int vacnic_beam_topk(const void* logits, const float* beam_scores, const int32_t* bans, int32_t n_ban, int32_t eos,
                     int32_t suppress_eos, int32_t forced_token, float* top_val, int32_t* top_idx, int64_t R, int64_t V,
                     int64_t ldl, int32_t K, int32_t logits_f32, void* stream);
/*
 * LM head + vacnic_beam_topk of the single-token decoder fused (MFULL:1997 `lm_head(outputs[0]) + final_logits_bias`, then the
 * log_softmax -> processors -> + beam_scores -> topk of beam_search above): R <= 8 rows, d_model <= 1024.  The [R][V] logits never
 * reach HBM unless `logits` (fp32 [R][ldl], optional: diagnostics) is given (matrix-core product, fp32 accumulation).
 *   h bf16 [R][d] contiguous (final hidden rows); emb bf16 [V][ldw]; bias fp32 [V] or NULL; bans / beam_scores / eos / suppress_eos /
 *   forced_token / top_val / top_idx as in vacnic_beam_topk with K = K2 (n_ban = row length of bans).
 *   workspace: vacnic_lmhead_topk_workspace(R, V, K2) floats of caller-owned scratch (per-workgroup partial results).
 *   forced_token >= 0: no logits are computed (h / emb / workspace may be NULL): the forced token scores beam_scores[r], the rest -inf.
 */
typedef struct {
  const void* h; const void* emb; const float* bias; float* logits; float* workspace;
  const float* beam_scores; const int32_t* bans; float* top_val; int32_t* top_idx;
  int64_t R, V, d, ldw, ldl, workspace_floats;
  int32_t n_ban, eos, suppress_eos, forced_token, K2;
} vacnic_lmhead_topk_args;
int64_t vacnic_lmhead_topk_workspace(int64_t R, int64_t V, int32_t K2);
int vacnic_lmhead_topk(const vacnic_lmhead_topk_args* a, void* stream);
/*
 * On-device beam-search bookkeeping: what transformers 4.18 BeamSearchScorer.process does on the host between
 * two decoder steps (TRAIN:513-520 -> GenerationMixin.beam_search), plus the NoRepeatNGram ban lists of the next position —
 * so the decode loop needs no device->host copy per token; the host reads this state once after the last position.
 * R = B * nb rows.  seq[k]: int32 [R][Lmax] token histories (position t reads seq[(t) & 1]... see beam_step), ping-pong.
 * Finished hypotheses are kept per batch item in list order: hyp_score f64 [B][nb] (= sum_logprobs / len^length_penalty),
 * hyp_len int32 [B][nb], hyp_seq int32 [B][nb][Lmax], hyp_cnt int32 [B], hyp_worst f64 [B]; done int32 [B].
 * vacnic_beam_init: histories = [start], beam scores = [0, -1e9, ...], empty lists.
 * vacnic_beam_step(cur_len): consumes beam_topk's [R][K2] candidates of the position whose histories hold cur_len tokens
 * (seq[(cur_len-1)&1]) and writes seq[cur_len&1], beam_scores (the next beam_topk's input), next_ids (last tokens, int64 [R]),
 * src_idx (int64 [R]: the row each new beam continues = KV-cache reorder index) and bans (int32 [R][Lmax], -1 padded; NULL if
 * no_repeat_ngram_size == 0).
 */
typedef struct {
  int32_t* seq[2]; float* beam_scores; int32_t* done; int32_t* hyp_cnt; double* hyp_worst; double* hyp_score; int32_t* hyp_len;
  int32_t* hyp_seq; int64_t* next_ids; int64_t* src_idx; int32_t* bans;
  int64_t B, nb, Lmax, V, eos, pad, no_repeat_ngram_size, early_stopping;
  float length_penalty;
} vacnic_beam_state;
int vacnic_beam_init(const vacnic_beam_state* st, int32_t start_token, void* stream);
int vacnic_beam_step(const vacnic_beam_state* st, const float* top_val, const int32_t* top_idx, int32_t K2, int32_t cur_len,
                     void* stream);
/*
 * Persistent single-token decoder step: every BartDecoderLayer of one position (MFULL:793-890 with
 * past_key_value, eval mode: self-attention over the KV cache, cross-attention over the encoder K/V projected once, FFN, the
 * three post-LayerNorms) in ONE launch — one workgroup per CU, grid barriers between the 8 phases of a layer, the next
 * projection's weights prefetched into registers behind each barrier.  R <= 8 rows (beams x batch), d_model <= 1024 with
 * 64-wide heads, ffn_dim <= 4096.  Same accumulation order and rounding points as the kernel-per-op chain vacnic_gemv_ln_bf16 /
 * vacnic_gemm_bf16 (skinny) / vacnic_attn_fwd (Tq = 1); results agree to the last bit up to fma-contraction choices of the compiler.
 *   layers   DEVICE array [L] of vacnic_decoder_layer: bf16 weights row-major [N][K] contiguous (w_kvq = k|v|q stacked, [3d][d]),
 *            fp32 biases and LayerNorm parameters; cross_kv bf16 [rows][S][2d] (k|v per source position) with cross_bs
 *            elements between rows (0: all rows share one source — the beams of one caption).
 *   cache    bf16 [L][R][Tmax + 1][2d]: k|v of position t are written at [l][r][t][0:2d], q is parked at [l][r][t + 1][0:d].
 *   h0       bf16 [R][d]: embedding LayerNorm output of the new token (vacnic_embed_ln_fwd).
 *   hbuf/obuf/ctx/qbuf ([R][d]) and fbuf ([R][F]): scratch.  Grid-barrier variant: on return the last layer's un-normalised
 *            block output is obuf and its residual is hbuf[L & 1]; final_layer_norm of the last layer is left to the consumer
 *            (vacnic_gemv_ln_bf16 with the LM head).  Slot variant: obuf holds the decoder's final hidden rows (final_layer_norm
 *            applied), ready for the LM-head GEMM; its projections run on the matrix cores (fp32 sums associate differently
 *            from the VALU kernels: parity to bf16 tolerance, not to the bit).
 *   sync     vacnic_decoder_step_sync_bytes() bytes, zeroed ONCE by the caller; word [sync_bytes / 4 - 64] is an error flag the
 *            kernel raises (and leaves) if a wait times out — check it when the results are read back.
 *   slots    vacnic_decoder_step_slots_bytes(L) bytes, zeroed ONCE by the caller: the phases hand their results over through
 *            tagged 16-byte units in this buffer (no grid barriers; needs max(d / 4, ffn / 16) <= 256 co-resident workgroups and
 *            d, ffn multiples of 16).  NULL selects the grid-barrier variant (kept as the A/B baseline).
 */
typedef struct {
  const void *w_kvq, *w_so, *w_cq, *w_co, *w_fc1, *w_fc2;
  const float *b_kvq, *b_so, *b_cq, *b_co, *b_fc1, *b_fc2;
  const float *ln_self_g, *ln_self_b, *ln_cross_g, *ln_cross_b, *ln_final_g, *ln_final_b;
  const void* cross_kv;
  int64_t cross_bs;
} vacnic_decoder_layer;
typedef struct {
  const vacnic_decoder_layer* layers;
  void* cache; const void* h0;
  void* hbuf[2]; void* obuf; void* ctx; void* qbuf; void* fbuf;
  const uint8_t* enc_mask;        /* uint8 [R][S], 0 = masked source position; may be NULL */
  uint32_t* sync;
  void* slots;
  int64_t L, R, d, H, F, S, t, Tmax;
  float eps, scale;
  uint64_t* trace; int64_t trace_wg;   /* profiling aid, normally NULL: 100 MHz time stamps of workgroup trace_wg, uint64 [8 L][8]
                                          ([phase][0] phase start, [1] inputs staged, [3] results stored, [4] stores acknowledged,
                                          [5] arrived at the barrier + next weights issued) */
} vacnic_decoder_step_args;
int64_t vacnic_decoder_step_sync_bytes(void);
int64_t vacnic_decoder_step_slots_bytes(int64_t L);
int vacnic_decoder_step(const vacnic_decoder_step_args* a, void* stream);
/* dst[r][:row_bytes] = src[g][:row_bytes] for rows row_stride_bytes apart (0 = row_bytes; both multiples of 16): KV-cache beam
 * reorder (_reorder_cache, MFULL:2066-2074).  period == 0: g = idx[r]; period > 0: g = (r / period) * period + idx[r % period] —
 * one beam permutation idx[period] applied to every layer's block of rows.  Copying only the filled prefix of a cache row
 * (row_bytes < row_stride_bytes) halves the traffic of the reorder on average. */
int vacnic_gather_rows(const void* src, void* dst, const int64_t* idx, int64_t rows, int64_t row_bytes, int64_t row_stride_bytes,
                       int64_t period, void* stream);

/* ---- input pipeline ---- */
/* uint8 image [B][3][H][W] -> fp32 [B][3][H][W]: torchvision ToTensor (x / 255) then Normalize ((t - mean[c]) / std[c]) of
 * TRAIN:741-764, with an optional per-image horizontal flip (flip[b] != 0; RandomHorizontalFlip, TRAIN:762 — the caller draws
 * the coin).  IEEE divisions, so the result is bit-identical to the torch formula. */
int vacnic_image_u8_normalize(const uint8_t* src, const uint8_t* flip, float* dst, int64_t B, int64_t H, int64_t W,
                              float mean0, float mean1, float mean2, float std0, float std1, float std2, void* stream);

/*
 * Fused LM head + cross entropy: logits = h . E^T (+ final_logits_bias) (MFULL:1885,1997) into
 * CrossEntropyLoss(ignore_index) (TRAIN:287,816) WITHOUT materialising the [R, V] logits.
 *   vacnic_lmhead_ce_fwd      one GEMM whose epilogue reduces each 256-column tile of a row to an online-softmax pair
 *                             {max, sum exp} (part[R][part_tiles][2], part_tiles = ceil(V/256)) and picks the target logit
 *                             (tl[R]); a combine kernel writes row_lse[R] and ACCUMULATES loss_sum / count (both zeroed by
 *                             the call) over rows whose target != ignore_index.  loss = loss_sum / count.
 *   vacnic_lmhead_ce_rowp     rowp[R][2] = {row_lse, (target != ignore) * grad_out * grad_scale / count} for the backward.
 *   vacnic_lmhead_ce_dlogits  recomputes the logits of vocabulary columns [col0, col0 + ncols) and writes
 *                             dlogits = (softmax - onehot(target)) * rowp[.][1] as bf16 [R][lddl] (columns ncols..round_up(ncols,8)
 *                             are written as zeros); the caller runs the dh / dE GEMMs on the chunk and re-uses the buffer.
 */
typedef struct {
  const void* h; const void* emb; const float* bias;      /* bf16 [R][ldh], bf16 [>=V][lde] (tied embedding), f32 [V] or NULL */
  const int64_t* targets;                                 /* [R] */
  float* part; float* tl; float* row_lse; float* loss_sum; float* count;
  int64_t R, V, D, ldh, lde, part_tiles, ignore_index;
} vacnic_lmhead_ce_args;
int vacnic_lmhead_ce_fwd(const vacnic_lmhead_ce_args* a, void* stream);
int vacnic_lmhead_ce_rowp(const float* row_lse, const int64_t* targets, const float* count, const float* grad_out,
                          float grad_scale, float* rowp, int64_t R, int64_t ignore_index, void* stream);
int vacnic_lmhead_ce_dlogits(const vacnic_lmhead_ce_args* a, int64_t col0, int64_t ncols, void* dl, int64_t lddl,
                             const float* rowp, void* stream);

/* zero `bytes` bytes at `ptr` on `stream`: the fp32 accumulators of split-K GEMMs, the arrival counters of the split-K fix-up.
 * hipMemsetAsync on an ordinary stream; a fill kernel while `stream` is being captured into a hipGraph (a memset NODE is not
 * reliably ordered before the kernel node that follows it when the graph replays). */
int vacnic_zero_bytes(void* ptr, int64_t bytes, void* stream);

/* ---- launch plans: a step's C-ABI call sequence recorded once and replayed from C++ -----------------------------------------
 * The reference issues every kernel of a training step from Python (TRAIN:253-383); the eager path here does too (~1400 calls).
 * Between vacnic_plan_begin() and vacnic_plan_end(h) every entry point of this library also freezes its arguments into plan h
 * while it runs; vacnic_plan_replay(h, first, last) re-issues commands [first, last) — same kernels, same streams, same order.
 * The caller keeps all buffers at their recorded addresses and refreshes inputs in place.  vacnic_plan_mark() (while recording)
 * = index of the next command, for replays split around host-side work.  vacnic_stream_fence(src, dst): dst waits for all work
 * enqueued on src so far — the recordable form of an event record + stream wait.  vacnic_plan_pause(1) .. vacnic_plan_pause(0)
 * (recording thread only; nests): calls in between run but are NOT recorded — the work the host repeats itself at a mark of
 * every replay (the DDP reducer's bucket casts and RCCL launches, reference role TRAIN:87).  Recording is bound to the thread
 * that called vacnic_plan_begin, and an entry point that returns an error while recording leaves no command behind.
 * Handles are small integers; -1 = error. */
int64_t vacnic_plan_begin(void);
int vacnic_plan_end(int64_t plan);
int64_t vacnic_plan_size(int64_t plan);
int64_t vacnic_plan_mark(void);
int vacnic_plan_pause(int on);
int vacnic_plan_replay(int64_t plan, int64_t first, int64_t last);
int vacnic_plan_destroy(int64_t plan);
int vacnic_stream_fence(void* src_stream, void* dst_stream);

/* ---- data-parallel communication: RCCL behind the C-ABI ------------------------------------------------------------------------
 * The reference's DDP wrapper (torch.nn.parallel.DistributedDataParallel over NCCL, TRAIN:87) all-reduces gradient buckets from
 * C++ hooks; here a bucket's SUM all-reduce is a stream-ordered launch on a stream the caller names, recordable into a launch
 * plan like any kernel.  librccl is resolved with dlopen (vacnic_comm_load; path = NULL searches the usual names), so the
 * library loads on a box without RCCL and these entry points then return VACNIC_UNSUPPORTED.
 *   vacnic_comm_unique_id   rank 0: 128 opaque bytes to hand to every rank out of band (ncclGetUniqueId)
 *   vacnic_comm_init        every rank, with its GPU current: communicator handle >= 0, or -1 (ncclCommInitRank)
 *   vacnic_allreduce_bucket in-place SUM over the communicator, dtype 0 = f32, 1 = bf16 (ncclAllReduce)
 *   vacnic_comm_broadcast   in-place broadcast from `root` (the parameter broadcast at construction, TRAIN:87)
 *   vacnic_event_record / vacnic_event_wait   named events (slots 0 .. 511): the consumer stream waits for ONE point of the
 *                           producer stream (a bucket's all-reduce), whatever that stream is given afterwards */
int vacnic_comm_load(const char* librccl_path);
int vacnic_comm_unique_id(void* out128);
int64_t vacnic_comm_init(const void* id128, int rank, int world);
int vacnic_allreduce_bucket(int64_t comm, void* buf, int64_t count, int dtype, void* stream);
int vacnic_comm_broadcast(int64_t comm, void* buf, int64_t count, int dtype, int root, void* stream);
int vacnic_comm_destroy(int64_t comm);
int vacnic_event_record(int slot, void* stream);
int vacnic_event_wait(int slot, void* stream);

/* ---- hardware probes (tests only): verify MFMA / ds_read_tr lane maps assumed by the kernels -- */
int vacnic_probe_layouts(float* out, const float* src128, int64_t n_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
