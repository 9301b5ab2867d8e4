// encoder.h — pieces of the TEM host orchestration (tem.hip) shared with the RTM path (rtm.hip):
// the workspace layout and the transformer-encoder layer loops (forward / backward).
#pragma once
#include "rowwise.h"

// ------------------------------------------------------------- workspace layout
struct LayerWs {
  int n_in, fan, n_out, Sq, M2;
  int64_t xn, pre_stats, kp, vp, qp, attn, ctx, y1, ff_stats, ln1, a1, h1, y2;
  int64_t amask;            // uint32 [n_in*fan][H]: keep bits of the attention dropout (attn_fwd_wf_kernel -> its backward)
};
struct Ws {
  int R, S, Mf, qpos;
  int64_t qmean, query_emb, x;
  LayerWs layer[PS_MAX_LAYERS];
  int64_t fin_stats, enc;
  int64_t item_scores, word_scores, loss_parts, item_terms, word_terms, loss_blk;
  int64_t word_blk, item_blk, ticket;   // folded scoring (ScoreArgs): word / item loss partials, arrival counter
  int64_t wsplit;           // bf16x3 planes of the last layer's wo / w1 / w2 (WSplit), 0: not allocated
  int64_t denc, dy2, do2, da1, dln1, dy1, do_, dctx, dq, dkv, dxn, dx, dqpre, dqmean;
  int64_t lnpart;           // PS_MAX_COLFOLD x [lnrows][3][d] parked LN-backward column sums
  int lnrows;               // parked rows per entry: 256 (the LayerNorm backward's row groups) or one per 32-row workgroup of the fused backward, whichever is more
  int64_t gcpart;           // [4 * row tiles][3][F] parked column sums of the FF2 dX GEMM (b1 gradient)
  int64_t abpart;           // per layer [n_in][3][d] parked attention bias gradients {bq, bk, bv} (sq1 backward)
  int64_t vrows, vcount;    // int32 [B*S] valid-row list of x and its length (EmbedArgs::vrows), TEM only
  int64_t stage;            // graph replay: step word + staged copies of the call's int64 index tensors (stage_layout)
  int64_t total;
};


#define TRY(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

int make_ws(const PsTemDesc& D, Ws& w);
// the last layer's weights as the fused kernels' bf16x3 planes inside the workspace (on = 0 when the x3 form is not taken)
WSplit make_wsplit(const PsTemDesc& D, const PsTemTensors& P, float* ws, const Ws& w);

// All encoder layers + the final LayerNorm on the consumed position: reads w.x, writes w.enc.
// Key-padding mask: `valid` [n_seq, S] floats if given, else u_item_idxs != P (TEM).
// `rows_listed`: w.vrows / w.vcount hold the list of valid (non-pad) rows of x (EmbedArgs::vrows, or the review
// transformer's rtm_rowlist_kernel); the K/V products of a one-layer encoder then run over those rows only.
// `fold_sc` (optional, TEM with replicas): item scoring + loss run in the epilogue of the last layer's fused kernel.
int enc_layers_forward(const PsTemDesc& D, const PsTemTensors& P, const int64_t* ui, const float* valid, float* ws,
                       const Ws& w, hipStream_t st, bool rows_listed = false, const ScoreArgs* fold_sc = nullptr);
bool enc_rowlist_taken(const PsTemDesc& D, const Ws& w, bool rows_listed);   // the encoder reads x through the valid-row list only
// Backward of the above: reads w.denc (grad wrt w.enc), accumulates parameter grads into G, writes w.dx.
// `fold` (optional): the LayerNorm backwards park their column sums in w.lnpart and append to this list; the caller
// must hand it to a later launch_embed_scatter (EmbedBwdArgs::fold).  nullptr: plain atomics.
// `score_on_side` (optional, TEM with replicas): the caller has NOT launched the score backward; when the last layer's
// backward is fused, d enc is derived from the scores inside that kernel and the score backward (table scatter only) is
// launched on the side stream behind it, off the dependent chain; otherwise it is launched first, as usual.
int enc_layers_backward(const PsTemDesc& D, const PsTemTensors& P, const PsTemTensors& G, const int64_t* ui,
                        const float* valid, float* ws, const Ws& w, hipStream_t st, ColFoldList* fold = nullptr,
                        const ScoreArgs* score_on_side = nullptr, bool rows_listed = false);

GemmProblem gp(const float* A, int lda, int ta, const float* Bm, int ldb, int tb, float* C, int ldc, int M, int N, int K);
int run1(const GemmProblem& p, hipStream_t st);
GemmProblem gp_wgrad(const float* dY, int lddy, const float* X, int ldx, float* dW, int n_out, int k_in, int rows);
int side_wgrads(GemmProblem* ps, int n, hipStream_t main_st);
int side_fork(hipStream_t main_st);
int side_run(GemmProblem* ps, int n, hipStream_t main_st);
int side_join(hipStream_t main_st);
void side_abort();                                   // error paths: release a fork nobody will signal (tem.hip)
void side_set_light(bool light);
hipStream_t side_stream_or(hipStream_t main_st);   // the side stream, or main_st when it is disabled
