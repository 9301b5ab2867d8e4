/*
 * prodsearch_hip.h — C ABI of libprodsearch_hip.so (MI355X / gfx950 only).
 *
 * The drop-in boundary of the ProdSearch ranking-loss training step.  The
 * reference (kepingbi/ProdSearch) is pure Python/PyTorch and has NO FFI of its
 * own; each entry point below names the reference interface it replaces
 * (file:line relative to the reference repo).  A maintainer binds them with the
 * ctypes stub shown in INTEGRATION.md (prodsearch_amd/_lib.py is that stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - indices are int64 (the reference's torch.int64 batches, data/batch_data.py:17-22),
 *     parameters/activations are fp32, row-major, nn.Linear weights are [out, in];
 *   - every call is asynchronous on `stream` (a hipStream_t), allocates nothing,
 *     synchronises nothing, and may be captured into a hipGraph;
 *   - return value: 0 = ok, otherwise an error code; ps_last_error() gives text;
 *   - no CPU fallback exists: without a gfx950 device every compute call fails.
 */
#ifndef PRODSEARCH_HIP_H
#define PRODSEARCH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PS_MAX_LAYERS 8
#define PS_OK 0
#define PS_ERR_ARG 1      /* bad descriptor / unsupported shape */
#define PS_ERR_HIP 2      /* a HIP runtime call failed          */

typedef void* ps_stream_t; /* hipStream_t */

enum { PS_MODEL_TEM = 0, PS_MODEL_QEM = 1 };
enum { PS_QENC_FS = 0, PS_QENC_AVG = 1 };

/* Shape/flag descriptor of one step.  Field <- reference flag (main.py:26-137). */
typedef struct PsTemDesc {
  int32_t B;             /* batch rows                         (--batch_size)            */
  int32_t K;             /* negatives per positive             (--neg_per_pos)           */
  int32_t L;             /* padded history length              (<= --uprev_review_limit) */
  int32_t Q;             /* padded query length                                          */
  int32_t W;             /* PV window                          (--pv_window_size)        */
  int32_t C;             /* candidates per row, eval only      (--candi_batch_size)      */
  int32_t d;             /* --embedding_size                                             */
  int32_t H;             /* --heads                                                      */
  int32_t F;             /* --ff_size                                                    */
  int32_t n_layers;      /* --inter_layers                                               */
  int64_t product_size;  /* P; pad index = P, tables have P+1 rows (item_transformer.py:34,46) */
  int64_t vocab_size;    /* V; pad index = V-1                    (item_transformer.py:35)     */
  int32_t model;         /* PS_MODEL_TEM: forward_dotproduct; PS_MODEL_QEM: forward_attn/QEM */
  int32_t query_encoder; /* PS_QENC_FS / PS_QENC_AVG          (--query_encoder_name)     */
  int32_t use_pos_emb;   /* --use_pos_emb                                                */
  int32_t use_item_pos;  /* --use_item_pos : output position -1 instead of 0             */
  int32_t bias_product;  /* --sim_func bias_product                                      */
  int32_t pos_weight;    /* --pos_weight : weight K on the positive's loss term          */
  int32_t sep_prod_emb;  /* --sep_prod_emb : history rows come from hist_product_emb     */
  int32_t training;      /* nn.Module.training; dropout is drawn only if set and dropout>0 */
  float   dropout;       /* --dropout                                                    */
  uint64_t seed;         /* Philox key of the dropout / sampling streams                 */
  uint64_t step;         /* Philox counter word: distinct per training step              */
} PsTemDesc;

/* One transformer layer's tensors (state_dict keys under
 * transformer_encoder.transformer_inter.{i}. — transformer.py:37-45, neural.py:20-26,86-96). */
typedef struct PsLayerTensors {
  float *wk, *bk;        /* self_attn.linear_keys   [d,d],[d] */
  float *wv, *bv;        /* self_attn.linear_values           */
  float *wq, *bq;        /* self_attn.linear_query            */
  float *wo, *bo;        /* self_attn.final_linear            */
  float *w1, *b1;        /* feed_forward.w_1 [F,d],[F]        */
  float *w2, *b2;        /* feed_forward.w_2 [d,F],[d]        */
  float *ff_ln_g, *ff_ln_b; /* feed_forward.layer_norm        */
  float *ln_g, *ln_b;    /* layer_norm (pre-LN, used only when i != 0: transformer.py:48-51) */
} PsLayerTensors;

/* A full set of model tensors.  The same struct describes parameters, their
 * gradients, and Adam moments (pointers may be NULL where the reference has no
 * gradient: product_bias unless bias_product, layer-0 ln, hist table unless sep). */
typedef struct PsTemTensors {
  float *product_emb;       /* product_emb.weight      [P+1, d] (item_transformer.py:46) */
  float *hist_product_emb;  /* hist_product_emb.weight [P+1, d] (:48) or NULL            */
  float *word_emb;          /* word_embeddings.weight  [V, d]   (:70)                    */
  float *product_bias;      /* product_bias            [P+1]    (:56)                    */
  float *word_bias;         /* word_bias               [V]      (:57)                    */
  float *fs_w, *fs_b;       /* query_encoder.f_W       [d,d],[d] (text_encoder.py:24)    */
  float *pe;                /* transformer_encoder.pos_emb.pe [5000, d] buffer (transformer.py:10-19) */
  float *final_ln_g, *final_ln_b; /* transformer_encoder.layer_norm (transformer.py:68)  */
  PsLayerTensors layer[PS_MAX_LAYERS];
} PsTemTensors;

/* The reference's ItemPVBatch (data/batch_data.py:3-37) + the two multinomial draws. */
typedef struct PsTemBatch {
  const int64_t *query_word_idxs;   /* [B,Q]   pad V-1 */
  const int64_t *target_prod_idxs;  /* [B]             */
  const int64_t *u_item_idxs;       /* [B,L]   pad P   */
  const int64_t *pos_iword_idxs;    /* [B,W]   pad V-1 */
  const int64_t *neg_item_idxs;     /* [B,K]   draw 1: torch.multinomial(prod_dists) item_transformer.py:447 */
  const int64_t *neg_word_idxs;     /* [B,W*K] draw 2: torch.multinomial(word_dists) item_transformer.py:268 */
  const int64_t *candi_prod_idxs;   /* [B,C]   eval only, pad P (item_pv_dataloader.py:44) */
} PsTemBatch;

/* Float offsets (in units of sizeof(float)) of the intermediates inside the
 * workspace; exported so the parity tests can compare every stage to the oracle.
 * Only the LAST layer's buffers are listed (n_layers == 1 covers everything). */
typedef struct PsTemWsLayout {
  int64_t total_floats;
  int32_t R;             /* encoder replicas per batch row: K+1 if dropout is drawn, else 1 */
  int32_t S;             /* L+1 */
  int64_t qmean, query_emb, x;                 /* [B,d],[B,d],[B,S,d]                 */
  int64_t kp, vp, qp, attn, ctx;               /* last layer: [n_in*S,d] x2, [n_in,d].. ; K/V rows of MASKED key positions
                                                  are not written when the one-layer sq1 path is in use (row-list projection) */
  int64_t y1, ln1, a1, h1, y2, enc;            /* last layer replica rows              */
  int64_t item_scores, word_scores, loss_parts;/* [B,1+K],[B,W,1+K],[B,2]             */
  int64_t denc, dx;                            /* backward: [B*R,d], [B,S,d]           */
} PsTemWsLayout;

const char* ps_version(void);
const char* ps_last_error(void);
/* The arithmetic the step computes in, as text for bench.py's `dtype` field: "f32 (...)" naming the bf16x3 product form
 * when it is enabled (ps_gemm_x3_config).  No reference counterpart (the reference is fp32 ATen throughout). */
const char* ps_arith_info(void);
/* Tuning knob (no reference counterpart): the last encoder layer's per-replica backward runs as one fused kernel from
 * this many replica rows upwards (default 1024, env PS_FUSE_BWD_MIN; below it five short launches are as fast).  The
 * parity tests set it to 1 to drive the fused kernel through the small golden cases.  Returns the previous value. */
int ps_set_fuse_bwd_min(int rows);
/* Tuning knob: how the backward's side stream crosses the main stream.  Bit 1: a join is a stream write-value / wait-value
 * pair instead of an event pair; bit 0: a fork is a wait-value on the side stream whose value the NEXT kernel launched on the
 * main stream stores as its first workgroup starts (every earlier main-stream kernel has completed by then) — the main stream
 * itself executes nothing for a fork (round 2: -9 us per C2 step against write operations, which in turn were -5..-8 us
 * against events).  Default 3 (env PS_SIDE_MODE), 0 = event pairs only.  The wait is a spinning one-thread kernel: tools that
 * let only one kernel run at a time (counter-collecting profilers) are recognised from their environment and always get
 * events (PS_SIDE_EVENTS=1 forces that).  Returns the previous value. */
int ps_set_side_mode(int mode);
/* The value crossings are used only where a start-up self-test (once per process and device: the side stream parked on a
 * probe word, released by a write-value on another stream, polled with a host-side timeout and, failing that, released from
 * the host) has shown that a wait makes progress beside its producer; otherwise event pairs, which cannot hang.
 * ps_side_values_in_use(): 1 = value waits in use, 0 = event pairs or a single stream.  PS_SIDE_SELFTEST_FAIL=1 makes the
 * probe fail (tests). */
int ps_side_values_in_use(void);
/* Error-path hygiene of the side stream: every ps_*_backward that fails releases a fork of the side stream whose
 * signalling launch never happened, so the caller's next synchronize returns and the error surfaces (it could otherwise
 * wait forever on hipStreamWaitValue32).  ps_side_abort() does the same on demand; ps_debug_fail_fork(n) is the test hook
 * that makes the n-th fork from now on fail after it has parked the side stream (0 = off). */
void ps_side_abort(void);
void ps_debug_fail_fork(int nth);
/* Deterministic mode (env PS_DETERMINISTIC=1, no reference counterpart: the reference's CUDA index_add_ / embedding backward are
 * not deterministic either): the item-transformer training step becomes bitwise reproducible run to run — everything on one
 * stream, weight gradients as per-split partial matrices added up in split order by a second launch, table scatters by
 * sole-owner half-waves that walk the task lists in order.  ~2.6x the default step time (0.78 ms at C2: a
 * popular row's additions are one chain); allocates scratch buffers on first use; process-wide.  The review transformer runs
 * in this mode too.  Returns the previous value; a negative argument only queries. */
int ps_set_deterministic(int on);
/* Product form of the large linears (no reference counterpart).  mode 1: launches with enough tiles to fill the chip take
 * the bf16x3 kernel — every fp32 operand split exactly into three bf16 values, six v_mfma_f32_32x32x16_bf16 per product step,
 * fp32 accumulation: as accurate as the fp32 MFMA (1.1e-7 of sum|a b|) at a third of its cycles; mode 0: the fp32 MFMA
 * everywhere.  force_shape -1: tile chosen by the launch's size; 0 / 1 / 2: 64x64 / 128x64 / 128x128 for every launch that
 * has an instantiation (tests); 3: the direct-to-LDS form (fp32 slabs by global_load_lds, split on the fragments in registers,
 * 128x128 tiles) wherever it applies; 4: weights pre-split once per call into bf16 plane images (ps_gemm_f32_weight; the d >= 256
 * step's forward / dX products), the size rule elsewhere.  3 and 4 are measured alternatives (profiles/r04_gemm_notes.md), not defaults.  Env PS_GEMM_X3 / PS_GEMM_X3_SHAPE set the initial values.  Process-wide. */
int ps_gemm_x3_config(int mode, int force_shape);
/* C = (A . op(W) + bias) * alpha with W declared a WEIGHT — [N][K] (tb 0, nn.Linear, models/neural.py:20-21, 86-96) or [K][N]
 * (tb 1: the input-gradient product of the same linear) with dense rows: W is split once into bf16 plane images for the
 * duration of the call and the product runs on the pre-split-weight kernel (gemm_x3w_kernel) where it applies
 * (ps_gemm_x3_config(1, 4), M >= 4096, N / K multiples of 32), otherwise as ps_gemm_f32 would.  What the d >= 256 step's forward / dX products do. */
int ps_gemm_f32_weight(const float* A, int lda, const float* W, int tb, float* C, int ldc, int M, int N, int K,
                       const float* bias, float alpha, ps_stream_t stream);

/* Workspace the caller allocates once per shape (bytes) and its layout. */
int ps_tem_workspace_layout(const PsTemDesc* desc, PsTemWsLayout* out);

/* loss = model(batch)                      -- ItemTransformerRanker.forward
 *   (models/item_transformer.py:352-359 -> forward_dotproduct :440-520, or forward_attn/QEM :361-438)
 * loss3[0..2] = {ps_loss + item_loss, ps_loss, item_loss} (device floats);
 * loss_acc (optional, device float[2]) += {ps_loss, item_loss}: the running sums the reference keeps in
 * model.ps_loss / model.item_loss with two .item() syncs per step (item_transformer.py:516-517). */
int ps_tem_forward(const PsTemDesc* desc, const PsTemTensors* params, const PsTemBatch* batch,
                   float* workspace, float* loss3, float* loss_acc, ps_stream_t stream);
/* The same with the two negative draws (ps_sample_negatives) folded into the forward's first launch: batch->neg_* are
 * ignored, neg_item_out [B,K] / neg_word_out [B,W*K] receive the draws and are what the backward must be given. */
int ps_tem_forward_sampled(const PsTemDesc* desc, const PsTemTensors* params, const PsTemBatch* batch,
                           const float* alias_prob, const int32_t* alias_idx, int64_t* neg_item_out, int64_t* neg_word_out,
                           float* workspace, float* loss3, float* loss_acc, ps_stream_t stream);

/* loss.backward()                          -- trainer.py:77 (autograd of the above).
 * ACCUMULATES loss_scale * dloss/dparam into `grads` (dense, like the reference's
 * nn.Embedding(sparse=False)); the caller zeroes grads (model.zero_grad(), trainer.py:76). */
int ps_tem_backward(const PsTemDesc* desc, const PsTemTensors* params, const PsTemBatch* batch,
                    float* workspace, const PsTemTensors* grads, float loss_scale,
                    const float* loss_scale_dev /* optional device scalar multiplied in (autograd's grad_output) */,
                    ps_stream_t stream);

/* The embedding-gather+score kernel alone (the launch ps_tem_forward makes after the encoder):
 * B*(1+K) item rows . encoder output + B*W*(1+K) word rows . target item row -> workspace scores.
 * Exposed so bench.py / rocprofv3 can time exactly this launch. */
int ps_gather_score(const PsTemDesc* desc, const PsTemTensors* params, const PsTemBatch* batch,
                    float* workspace, ps_stream_t stream);

/* scores = model.test(batch) [B,C]         -- test_dotproduct (item_transformer.py:111-146)
 *                                             / test_attn QEM (:148-160,189-195) */
int ps_tem_score(const PsTemDesc* desc, const PsTemTensors* params, const PsTemBatch* batch,
                 float* workspace, float* scores, ps_stream_t stream);

/* The two torch.multinomial draws of one forward (item_transformer.py:447, :268):
 * uniform items over [0,P) and alias-table words ~ word_dists.  alias tables are
 * built on the host by ps_build_alias_host. */
int ps_sample_negatives(const PsTemDesc* desc, const float* alias_prob, const int32_t* alias_idx,
                        int64_t* neg_item_idxs, int64_t* neg_word_idxs, ps_stream_t stream);
int ps_build_alias_host(const double* dist_host, int64_t n, float* prob_host, int32_t* alias_host);

/* optim.step()                             -- Optimizer.step (models/optimizers.py:205-243):
 * clip_grad_norm_(max_grad_norm) over all listed grads, then dense Adam(eps) with
 * bias correction, optional L2 (weight_decay) and noam schedule.  The tensor
 * table lives on the device (ps_adam_plan_bytes / ps_adam_plan_write_host); its
 * tail is device scratch the two launches of a step hand the gradients' non-zero
 * mask through (the update does not re-read 16-byte groups of zeros), so a plan
 * serves ONE optimizer step at a time and must be writable.
 * state[0] (device, int64) holds the step count and is incremented by the call;
 * (float*)(state+2) is scratch for n_chunks partial sums. */
typedef struct PsAdamHyper {
  float lr;            /* --lr (original_lr when noam)                 */
  float beta1, beta2;  /* --beta1 --beta2                              */
  float eps;           /* 1e-9 (optimizers.py:186-187)                 */
  float weight_decay;  /* --l2_lambda                                  */
  float max_grad_norm; /* --max_grad_norm, 0 = no clip                 */
  int32_t noam;        /* --decay_method noam                          */
  int32_t warmup_steps;/* --warmup_steps                               */
  float grad_scale;    /* multiplies every grad first (1/world for DP) */
  int32_t zero_grads;  /* dense step: leave every gradient it consumed at 0 (the next zero_grad() is then free) */
  int32_t method;      /* --optim (optimizers.py:175-187): 0 adam, 1 sgd, 2 adagrad (eps 1e-10; the plan's `m` is the sum of
                          squares, initialised by the caller with adagrad_accum), 3 adadelta (rho 0.9, eps 1e-6; `m` = square_avg,
                          `v` = acc_delta).  Same clip, same noam schedule; sgd touches neither `m` nor `v`. */
  int32_t pad_;
} PsAdamHyper;

int64_t ps_adam_plan_bytes(int32_t n_tensors, const int64_t* numel_host);
int ps_adam_plan_write_host(int32_t n_tensors, float* const* p, float* const* g, float* const* m,
                            float* const* v, const int64_t* numel_host, void* plan_host);
int32_t ps_adam_plan_chunks_host(const void* plan_host);     /* grid size; state needs 2 int64 + n_chunks floats */
int ps_clip_adam_dense(const void* plan_dev, int32_t n_chunks, const PsAdamHyper* hyper,
                       int64_t* state_dev, float* gnorm_out_dev /* [2]: norm, lr; may be NULL */,
                       ps_stream_t stream);

/* The same step cut in two for the SHARDED data-parallel optimizer (no reference counterpart; the reference is
 * single-process, trainer.py:64-83): after a reduce-scatter every rank owns 1/world of the flat gradient, and
 * clip_grad_norm_ (optimizers.py:241-242) is a norm over ALL gradients, so the ranks' partial sums of squares meet in one
 * scalar all-reduce between the two launches.  ps_adam_sumsq: out_sumsq_dev[0] = sum over the plan of (g*grad_scale)^2
 * (fixed-order reduction) and state[0] += 1.  ps_adam_update_ext: clip + Adam exactly as ps_clip_adam_dense, with the
 * global sum of squares read from total_sumsq_dev. */
int ps_adam_sumsq(const void* plan_dev, int32_t n_chunks, const PsAdamHyper* hyper, int64_t* state_dev,
                  float* out_sumsq_dev, ps_stream_t stream);
int ps_adam_update_ext(const void* plan_dev, int32_t n_chunks, const PsAdamHyper* hyper, int64_t* state_dev,
                       const float* total_sumsq_dev, float* gnorm_out_dev, ps_stream_t stream);

/* Peer-to-peer reduce-scatter, local half (no reference counterpart: trainer.py:64-83 is single-process): recv [world][n] =
 * every rank's copy of this rank's slice of the flat gradient (after an equal-split all-to-all); out[i] = sum over ranks in
 * rank order; the same launch clears `zero_n` floats at `zero` (the flat gradient buffer the all-to-all has consumed).
 * n and zero_n multiples of 4, buffers 16-byte aligned. */
int ps_sum_slices(const float* recv, int32_t world, int64_t n, float* out, float* zero, int64_t zero_n, ps_stream_t stream);

/* model.zero_grad() helper (trainer.py:76): async memset of a float buffer. */
int ps_zero_floats(float* p, int64_t n, ps_stream_t stream);

/* ------------------------------------------------------------------ graph-replayed training step
 * ps_tem_forward_step / ps_tem_backward_step run the same launch sequences as ps_tem_forward / ps_tem_backward, but
 * capture them once per shape (second call) and replay them as HIP graphs afterwards; what varies per call — the
 * caller's index tensors, the Philox step, the loss output — enters through a staging prologue (first node) and the
 * loss node, whose arguments are refreshed before each replay.  Same kernels, same results.  It cuts the host cost of
 * a step (347 -> 285 us), which matters when the host also builds batches; it does not shorten the device timeline, so
 * it is opt-in: PS_GRAPHS=1 (otherwise ps_graph_replay_enabled() is 0 and callers use the eager entry points).
 *   forward_step : batch as for ps_tem_forward; sampler_prob/alias (ps_build_alias_host tables) non-NULL => the two
 *                  negative draws (item_transformer.py:447, :268) happen in the prologue and batch->neg_* are ignored.
 *   backward_step: reuses the staged inputs of the last forward_step on this workspace; zero_ptr/zero_floats fold
 *                  model.zero_grad() of the flat gradient buffer in as the first node (NULL/0: keep accumulating). */
int ps_graph_replay_enabled(void);
int ps_tem_staged_batch(const PsTemDesc* desc, float* workspace, PsTemBatch* out);   /* where forward_step staged the indices */
int ps_tem_forward_step(const PsTemDesc* desc, const PsTemTensors* params, const PsTemBatch* batch,
                        const float* sampler_prob, const int32_t* sampler_alias, float* workspace, float* loss3,
                        float* loss_acc, ps_stream_t stream);
int ps_tem_backward_step(const PsTemDesc* desc, const PsTemTensors* params, float* workspace, const PsTemTensors* grads,
                         float loss_scale, float* zero_ptr, int64_t zero_floats, ps_stream_t stream);

/* ------------------------------------------------------------------ full-catalogue evaluation
 * Trainer.test / validate over ALL products (trainer.py:125-226 with test_candi_size < 1): encode each (user, query)
 * row once, score it against every row of the table with one fp32 MFMA GEMM per table panel, and select the top-k and
 * the target's rank on the device.  ps_tem_encode = the eval-mode sequence representation of item_transformer.py:118-131
 * (descriptor as for ps_tem_score, C >= 1; candidates are not read). */
int ps_tem_encode(const PsTemDesc* desc, const PsTemTensors* params, const PsTemBatch* batch, float* workspace,
                  float* enc_out /* [B,d] */, ps_stream_t stream);
/* scores[b,p] = q[b]·table[p] (+ bias[p]), p in [0, n_rows).  top_idx/top_score [B,topk]: best first, ties by lower
 * row id (unfilled slots: -1 / -inf when n_rows < topk).  rank[b] (optional, needs target) = 1 + number of rows ranked
 * ahead of target[b] under the same order = position in `argsort(scores)[::-1]` (trainer.py:137,172-180); 0 if the
 * target is not a table row.  topk <= 256.  scratch: ps_rank_scratch_bytes() bytes, 256-byte aligned. */
int64_t ps_rank_scratch_bytes(int32_t B, int64_t n_rows, int32_t d, int32_t topk);
int ps_rank_all(const float* q, int32_t B, int32_t d, const float* table, int64_t n_rows, const float* bias,
                const int64_t* target, int32_t topk, int64_t* top_idx, float* top_score, int32_t* rank,
                void* scratch, int64_t scratch_bytes, ps_stream_t stream);
/* One rank's part of the same ranking over a ROW-SHARDED table (prodsearch_amd/sharded.py; the reference keeps
 * the whole table on one device, item_transformer.py:46): table row i is catalogue id i * id_mul + id_add (world, rank);
 * target[b] is a catalogue id and target_score[b] its score, computed where its row lives (+inf: not a product);
 * ahead[b] = rows of THIS shard ranked ahead of the target under (score desc, id asc), the target's own row excluded;
 * top_idx carries catalogue ids.  The caller adds the shards' counts (rank = 1 + sum) and merges their top-k lists. */
int ps_rank_shard(const float* q, int32_t B, int32_t d, const float* table, int64_t n_rows, const float* bias,
                  const int64_t* target, const float* target_score, int64_t id_mul, int64_t id_add, int32_t topk,
                  int64_t* top_idx, float* top_score, int32_t* ahead, void* scratch, int64_t scratch_bytes, ps_stream_t stream);

/* ------------------------------------------------------------------ row-sparse optimizer path
 * For tables too large to stream every step (BASELINE configs[4]: 50 M x 256).  The gradient tensors
 * stay dense, like nn.Embedding(sparse=False) gives the reference (item_transformer.py:46,70); the
 * optimizer, the zeroing and the data-parallel exchange only visit the rows a step touched.
 * Replaces: the table part of Optimizer.step (optimizers.py:241-243) and model.zero_grad (trainer.py:76).
 * Semantics of untouched rows: p, m, v unchanged (torch.optim.SparseAdam's rule), NOT dense Adam's decay. */
typedef struct PsIdxList { const int64_t* idx; int64_t n; } PsIdxList;   /* device pointer, element count */

/* Sorted unique row ids (!= pad_row) over up to 8 index lists -> rows_out[0..*count).  ws = ps_coalesce_ws_bytes
 * bytes, zeroed once by the caller and left zeroed by every call.  cap >= min(sum n, n_rows). */
int64_t ps_coalesce_ws_bytes(int64_t n_rows);
int ps_coalesce_rows(const PsIdxList* lists_host, int32_t n_lists, int64_t n_rows, int64_t pad_row, void* ws_dev,
                     int64_t* rows_out_dev, int64_t cap, int32_t* count_out_dev, ps_stream_t stream);
/* values[u,:] = table[rows[u],:] and the inverse, u < *count_dev (or < cap when count_dev is NULL); d % 4 == 0 */
int ps_gather_rows(const float* table_dev, int32_t d, const int64_t* rows_dev, const int32_t* count_dev, int64_t cap,
                   float* values_out_dev, ps_stream_t stream);
int ps_scatter_rows(float* table_dev, int32_t d, const int64_t* rows_dev, const int32_t* count_dev, int64_t cap,
                    const float* values_dev, ps_stream_t stream);
int ps_zero_rows(float* table_dev, int32_t d, const int64_t* rows_dev, const int32_t* count_dev, int64_t cap,
                 ps_stream_t stream);
/* Device address of a coalesce workspace's status word: 0 ok, 1 an index outside [0, n_rows) was dropped, 2 more
 * unique rows than `cap`.  Callers copy it back when they can afford a sync (the step itself never does). */
const int32_t* ps_coalesce_bad_flag(void* ws_dev, int64_t n_rows);

/* Data-parallel exchange of a table's touched rows with NO host synchronisation (new: the reference is single-process,
 * trainer.py:64-83).  Every rank packs a fixed-capacity message — msg_rows[cap]: its sorted touched ids, then -1;
 * msg_vals[cap,d]: their gradient rows, then zeros — the messages are all-gathered (RCCL), the union of the ids is
 * ps_coalesce_rows over the gathered ids with pad_row = -1, and ps_merge_rows writes, for every union row, the sum of the
 * ranks' rows IN RANK ORDER into the dense gradient: bitwise the same result on every rank.  world <= 32. */
int ps_pack_rows(const float* grad_dev, int32_t d, const int64_t* rows_dev, const int32_t* count_dev, int64_t cap,
                 int64_t* msg_rows_dev, float* msg_vals_dev, ps_stream_t stream);
int ps_merge_rows(const int64_t* all_rows_dev, const float* all_vals_dev, int32_t world, int64_t cap, int32_t d,
                  float* grad_dev, const int64_t* union_rows_dev, const int32_t* union_count_dev, int64_t union_cap,
                  ps_stream_t stream);

typedef struct PsRowTable {
  float* p; float* g; float* m; float* v;   /* [n_rows, d] parameter, dense gradient, Adam moments */
  const int64_t* rows;                      /* touched rows (device)                                 */
  const int32_t* count;                     /* number of valid entries of rows (device)              */
  int64_t cap;                              /* launch bound: *count <= cap                           */
  int32_t d;
  int32_t pad_;
} PsRowTable;

/* clip_grad_norm_ over (tensors of the dense plan + touched rows of the tables) then Adam on exactly those;
 * touched gradient rows are zeroed in the same pass.  plan = ps_adam_plan_* over the small tensors.
 * state_dev: 2 int64 {step, -} followed by ps_adam_rowsparse_state_floats() floats of scratch. */
int64_t ps_adam_rowsparse_state_floats(int32_t n_chunks, const PsRowTable* tables_host, int32_t n_tables);
int ps_clip_adam_rowsparse(const void* plan_dev, int32_t n_chunks, const PsRowTable* tables_host, int32_t n_tables,
                           const PsAdamHyper* hyper, int64_t* state_dev, float* gnorm_out_dev, ps_stream_t stream);

/* Lazy-EXACT dense Adam on the row-sparse machinery (args.lazy_exact_adam): last_dev[i][r] = optimizer steps row r of table i
 * has had applied; the call replays, for every listed row (tables_host[i].rows / count) or every row (all_rows != 0), the
 * steps last+1 .. state_dev[0] it missed with a ZERO gradient — the dense optimizer's arithmetic for a row without gradient
 * (optimizers.py:186-187, 241-243: the moments decay, the row keeps moving) — and sets last = state_dev[0] (+1 with `advance`:
 * the step in progress is about to update the row).  Called before a forward reads the rows, over the final touched list before
 * ps_clip_adam_rowsparse, and with all_rows before evaluation / checkpointing.  tables_host[i].g is not read. */
int ps_rowsparse_catchup(const PsRowTable* tables_host, int32_t n_tables, int32_t* const* last_dev, const int64_t* n_rows_host,
                         const PsAdamHyper* hyper, const int64_t* state_dev, int32_t advance, int32_t all_rows,
                         ps_stream_t stream);

/* ------------------------------------------------------------------ row-sharded tables (no reference
 * counterpart: item_transformer.py:46,464-469 keeps the whole table on one device).  Row i lives on rank i % world at local
 * row i / world.  Per step: ps_coalesce_rows gives the rank's sorted unique rows; ps_shard_bucket cuts that list into one
 * fixed-capacity request per owner — send_ids[o][0..capp): local rows (id / world) of the ids with id % world == o,
 * ascending, then -1 — and records the slot o * capp + position of every list entry (slot_of); the requests and then the
 * rows (ps_gather_rows accepts the -1 padding: zero rows) travel by equal-split all-to-alls; ps_shard_remap rewrites an
 * index tensor into slots (pad_in -> pad_out), so the receive buffer [world * capp + 1, d] IS the table the step reads.
 * The gradient returns the same way; ps_coalesce_rows (pad -1) + ps_merge_rows over the received requests give the owner's
 * touched rows and their rank-ordered sums.  *bad_dev: 1 = an index missing from the list, 2 = a request overflowed capp. */
int ps_shard_bucket(const int64_t* rows_dev, const int32_t* count_dev, int32_t world, int64_t capp, int64_t* send_ids_dev,
                    int32_t* slot_of_dev, int32_t* bad_dev, ps_stream_t stream);
int ps_shard_remap(const int64_t* idx_dev, int64_t n, int64_t pad_in, const int64_t* rows_dev, const int32_t* count_dev,
                   const int32_t* slot_of_dev, int64_t pad_out, int64_t* out_dev, int32_t* bad_dev, ps_stream_t stream);
/* ps_clip_adam_rowsparse cut in two for sharded tables: tables [0, n_shared) and the dense plan are replicated (their sum of
 * squares counts once), tables [n_shared, n_tables) are this rank's shards (their sums add over the ranks).  sums_dev[0] =
 * replicated part, sums_dev[1] = owned part; the caller all-reduces sums_dev[1] between the calls; the clip norm
 * (optimizers.py:241-242) is sqrt(sums[0] + sums[1]).  state as for ps_clip_adam_rowsparse. */
int ps_rowsparse_sumsq(const void* plan_dev, int32_t n_chunks, const PsRowTable* tables_host, int32_t n_tables, int32_t n_shared,
                       const PsAdamHyper* hyper, int64_t* state_dev, float* sums_dev, ps_stream_t stream);
int ps_rowsparse_update_ext(const void* plan_dev, int32_t n_chunks, const PsRowTable* tables_host, int32_t n_tables,
                            const PsAdamHyper* hyper, int64_t* state_dev, const float* sums_dev, float* gnorm_out_dev,
                            ps_stream_t stream);

/* ------------------------------------------------------------------ RTM (review_transformer)
 * ProductRanker (models/ps_model.py:53-370) with the pv (models/PV.py) / pvc (models/PVC.py) review
 * encoders.  Sequences are [query, R reviews]; K negatives per row (training) or C candidates (eval). */
enum { PS_RENC_PV = 0, PS_RENC_PVC = 1,
       PS_RENC_FS = 2,    /* review vector = tanh(f_W . dropout(masked mean of its words) + b)  (ps_model.py:148-149, 301-305;   */
       PS_RENC_AVG = 3 }; /*                 dropout(masked mean)                                text_encoder.py:19-40, 62-83)  */

typedef struct PsRtmDesc {
  int32_t B, K;          /* batch rows, --neg_per_pos                                             */
  int32_t R;             /* padded reviews per sequence (<= --uprev_review_limit + --iprev_review_limit) */
  int32_t Q;             /* padded query length                                                   */
  int32_t W;             /* PV window (--pv_window_size), words predicted per positive review     */
  int32_t WL;            /* words per review of the pvc encoder (--review_word_limit)             */
  int32_t C;             /* candidates per row, eval only                                         */
  int32_t d, H, F, n_layers;
  int64_t vocab_size;    /* V, pad = V-1                                                          */
  int64_t review_count;  /* pad review = review_count-1 (ps_model.py:70)                          */
  int32_t review_encoder;/* PS_RENC_PV / PVC / FS / AVG (--review_encoder_name)                   */
  int32_t query_encoder; /* PS_QENC_FS / PS_QENC_AVG                                              */
  int32_t use_pos_emb, use_seg_emb, pos_weight;
  int32_t train_pv;      /* forward(batch, train_pv): add the PV word-prediction loss (ps_model.py:265-280) */
  int32_t training;
  float dropout;         /* --dropout                                                             */
  float corrupt_rate;    /* --corrupt_rate: token dropout of the pvc encoder (PVC.py:46-54)       */
  uint64_t seed, step;
  int32_t use_user_emb;  /* --use_user_emb / --use_item_emb: per-position user / item embeddings added  */
  int32_t use_item_emb;  /*   to every sequence (ps_model.py:325-334, :229-234)                         */
  int64_t user_size;     /* pad user = user_size, pad item = product_size (ps_model.py:66-67)           */
  int64_t product_size;
} PsRtmDesc;

typedef struct PsRtmTensors {
  float *word_emb;       /* word_embeddings.weight [V,d] (= review_encoder.context_embeddings for pvc) */
  float *review_emb;     /* review_encoder.review_embeddings.weight [review_count,d] (pv) or NULL  */
  float *seg_emb;        /* seg_embeddings.weight [4,d]                                            */
  float *fs_w, *fs_b;    /* query_encoder.f_W                                                      */
  float *pe;             /* transformer_encoder.pos_emb.pe                                         */
  float *final_ln_g, *final_ln_b;
  float *wo_w, *wo_b;    /* transformer_encoder.wo [1,d],[1] (transformer.py:69,96)                */
  float *user_emb;       /* user_emb.weight [user_size+1,d]       (use_user_emb) or NULL           */
  float *product_emb;    /* product_emb.weight [product_size+1,d] (use_item_emb) or NULL           */
  float *rev_fs_w, *rev_fs_b; /* review_encoder.f_W [d,d],[d] (fs review encoder) or NULL          */
  PsLayerTensors layer[PS_MAX_LAYERS];
} PsRtmTensors;

/* ProdSearchTrainBatch / ProdSearchTestBatch (data/batch_data.py:137-223, :94-135) + the PV-loss draw */
typedef struct PsRtmBatch {
  const int64_t *query_word_idxs;         /* [B,Q]                                   */
  const int64_t *pos_prod_ridxs;          /* [B,R]                                   */
  const int64_t *pos_seg_idxs;            /* [B,R+1]                                 */
  const int64_t *pos_prod_rword_idxs;     /* [B,R,W] PV target words (train_pv) / [B,R,WL] review words (pvc, !train_pv) */
  const uint8_t *pos_prod_rword_masks;    /* [B,R,W] uint8                           */
  const int64_t *neg_prod_ridxs;          /* [B,K,R]                                 */
  const int64_t *neg_seg_idxs;            /* [B,K,R+1]                               */
  const int64_t *neg_prod_rword_idxs;     /* [B,K,R,WL] pvc, !train_pv               */
  const int64_t *pos_prod_rword_idxs_pvc; /* [B,R,WL]   pvc, train_pv                */
  const int64_t *neg_prod_rword_idxs_pvc; /* [B,K,R,WL] pvc, train_pv                */
  const int64_t *neg_word_idxs;           /* [B*R, W*K] torch.multinomial(word_dists) of PV.py:57 / PVC.py:81 */
  const int64_t *candi_prod_ridxs;        /* [B,C,R]   eval                          */
  const int64_t *candi_seg_idxs;          /* [B,C,R+1] eval                          */
  const float *review_embeddings;         /* [review_count,d] eval table (get_review_embeddings, ps_model.py:186-203) */
  const int64_t *pos_user_idxs;           /* [B,R+1]   use_user_emb (pad user_size)  */
  const int64_t *neg_user_idxs;           /* [B,K,R+1]                               */
  const int64_t *pos_item_idxs;           /* [B,R+1]   use_item_emb (pad product_size) */
  const int64_t *neg_item_idxs;           /* [B,K,R+1]                               */
  const int64_t *candi_seq_user_idxs;     /* [B,C,R+1] eval                          */
  const int64_t *candi_seq_item_idxs;     /* [B,C,R+1] eval                          */
  const uint8_t *neg_prod_rword_masks;    /* [B,K,R,WL] uint8: fs / avg review encoders (with pos_prod_rword_masks [B,R,WL]) */
} PsRtmBatch;

int ps_rtm_workspace_floats(const PsRtmDesc* desc, int32_t eval, int64_t* total);
/* Offsets (in floats) of the intermediates inside the workspace, for stage-by-stage parity tests.  Sequences are
 * n = b*J + j (J = 1+K training: j = 0 positive; J = C eval), S = R+1 positions [query, reviews]. */
typedef struct PsRtmWsLayout {
  int64_t total_floats;
  int32_t Bseq, S, J, pad_;
  int64_t query_emb;   /* [B,d]        FS(query)                                   (ps_model.py:257-258) */
  int64_t valid;       /* [Bseq,S]     1 = unmasked position                        (:316-318)            */
  int64_t x;           /* [Bseq,S,d]   encoder input (review vectors + seg/user/item emb, masked, + pe) (:320-334) */
  int64_t vec;         /* [B*R,d]      review vector the PV loss predicts from (train_pv)  (PV.py:53-54, PVC.py:77-78) */
  int64_t cnt;         /* [Bseq,R]     non-pad words per review (pvc)                */
  int64_t enc;         /* [Bseq,d]     final-LayerNorm output at position 0          (transformer.py:90-95) */
  int64_t scores;      /* [Bseq]       wo . enc + b                                  (transformer.py:96)    */
  int64_t weight;      /* [Bseq]       BCE weight of each sequence                   (ps_model.py:344-345)  */
  int64_t pv_scores;   /* [B*R,W,1+K]  PV-loss logits (train_pv)                     (PV.py:59-63)          */
  int64_t dx;          /* [Bseq,S,d]   gradient w.r.t. x after the backward                                 */
} PsRtmWsLayout;
int ps_rtm_workspace_layout(const PsRtmDesc* desc, int32_t eval, PsRtmWsLayout* out);
/* loss = model(batch, train_pv) -- ProductRanker.forward (ps_model.py:241-358); loss3 = {loss, ps_loss, pv_loss} */
int ps_rtm_forward(const PsRtmDesc* desc, const PsRtmTensors* params, const PsRtmBatch* batch, float* workspace,
                   float* loss3, ps_stream_t stream);
/* loss.backward() (trainer.py:77): accumulates into dense grads */
int ps_rtm_backward(const PsRtmDesc* desc, const PsRtmTensors* params, const PsRtmBatch* batch, float* workspace,
                    const PsRtmTensors* grads, float loss_scale, const float* loss_scale_dev, ps_stream_t stream);
/* scores = model.test(batch) [B,C] -- ProductRanker.test (ps_model.py:205-239) */
int ps_rtm_score(const PsRtmDesc* desc, const PsRtmTensors* params, const PsRtmBatch* batch, float* workspace,
                 float* scores, ps_stream_t stream);
/* model.get_review_embeddings() for the pvc / fs / avg encoders (ps_model.py:186-203): per review the (uncorrupted,
 * undropped) mean of its non-pad words — fs: projected, tanh(f_W . mean + b) — last row 0.
 * scratch: [review_count, d] floats, fs only (else NULL). */
int ps_rtm_review_embeddings(const PsRtmDesc* desc, const PsRtmTensors* params, const int64_t* review_words,
                             float* scratch, float* out, ps_stream_t stream);

/* Host evaluation of the dropout stream (tests pin oracle/philox.py to it): multiplier
 * (0 or 1/(1-p)) of element (row, col) of dropout site `site` at desc->seed/step/dropout. */
float ps_dropout_mult_host(const PsTemDesc* desc, uint32_t site, uint32_t row, uint32_t col);

/* Measurement hook (bench.py's roofline): while armed, the launch sites of ONE tagged kernel ("gather_score", "mlp_fwd",
 * "mlp_bwd", "rtm_embed") bracket every launch with a HIP event pair recorded on the launch stream, up to max_samples;
 * ps_ktimer_read synchronises, returns the average / minimum duration in microseconds and disarms.  tag NULL: disarm. */
int ps_ktimer_arm(const char* tag, int32_t max_samples);
int ps_ktimer_read(double* avg_us, double* min_us, int32_t* count);

/* Unit-test hook of the fp32 MFMA GEMM: C[M,N] = alpha * op(A) op(B) (+bias) (+C if accumulate).
 * ta: A stored [K,M];  tb==0: B stored [N,K] (nn.Linear), tb==1: B stored [K,N]. */
int ps_gemm_f32(const float* A, int lda, int ta, const float* Bm, int ldb, int tb,
                float* Cm, int ldc, int M, int N, int K, const float* bias, float alpha,
                int accumulate, ps_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PRODSEARCH_HIP_H */
