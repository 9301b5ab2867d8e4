// rowwise.h — launchers of the row-parallel (HBM/latency-bound) kernels.
#pragma once
#include "common.h"

struct ScoreArgs {
  int B, K, W, C, R, d;          // C > 0: eval mode (B*C candidate tasks only)
  FDiv fK1, fWK1, fC;            // fast dividers of K+1, W*(K+1), C (filled by score_finish)
  int64_t P, V;
  int bias_product, pos_weight;
  const int64_t* target; const int64_t* neg_items; const int64_t* pos_words; const int64_t* neg_words;
  const int64_t* candi;
  const float* product_emb; const float* word_emb; const float* product_bias; const float* word_bias;
  const float* enc;              // [B*R, d]
  float* item_scores;            // [B,1+K]   (eval: [B,C])
  float* word_scores;            // [B,W,1+K]
  float* loss_parts;             // [B,2]
  float* item_terms;             // [B,1+K]   per-task loss terms (softplus), written by the gather+score kernel
  float* word_terms;             // [B,W,1+K]
  float* loss3;                  // {total, ps, item}
  float* loss_acc;               // optional running sums {ps, item} (item_transformer.py:516-517)
  float* loss_blk;               // [2*blocks] per-workgroup loss partials written by the gather+score kernel
  int loss_nblk;                 // filled by launch_score_fwd
  // backward
  float scale;                   // loss_scale
  const float* scale_dev;        // optional device scalar multiplied into scale
  float* denc;                   // [B*R,d]  (null: the consumer derives d enc from the scores itself, MlpBwdArgs::item_scores)
  int part;                      // replicas only: 0 = everything, 1 = d enc only (no table scatter), 2 = table scatter only
  float* g_product_emb; float* g_word_emb; float* g_product_bias; float* g_word_bias;
  // ---- folded form (TEM with replicas, the wave-specialised fused MLP forward): no gather+score / loss launches.
  //  * the word tasks (item_to_words, item_transformer.py:260-283: encoder-independent) run as extra workgroups of the
  //    embed launch, PS_WORD_TASKS_PER_WG each, leaving one loss partial per workgroup in word_blk;
  //  * the item tasks run in the epilogue of the fused MLP forward (row m = (b, j) of enc is dotted with item row
  //    idx(b, j) right where it is produced), one loss partial per workgroup in item_blk (write-through stores);
  //    idx(b, j) right where it is produced); each workgroup hands its loss partial over in fixed point with one
  //    returning 64-bit atomic (`ticket`: 8 shard words + 1 top word, zeroed by the embed launch);
  //  * the workgroup of the fused kernel that arrives last adds the word partials in a fixed order and writes the loss.
  float* word_blk; int word_nblk; float* item_blk; uint32_t* ticket;
  // negative words drawn inside the launch that consumes them (the sampling workgroups of the same launch have not
  // necessarily run yet): same Philox stream and alias table as sample_kernel, so the values equal neg_words[]
  const float* samp_prob; const int32_t* samp_alias; uint32_t samp_step, samp_k0, samp_k1; int samp_inline;
  unsigned long long* stamp;     // diagnostic build: per-workgroup s_memrealtime stamps of the gather+score launch (tools/gather_wg_times.py)
};
#define PS_WORD_TASKS_PER_WG 16
inline void score_finish(ScoreArgs& a) {
  a.fK1 = make_fdiv(a.K + 1); a.fWK1 = make_fdiv(a.W * (a.K + 1)); a.fC = make_fdiv(a.C > 0 ? a.C : 1);
}
int launch_score_fwd(ScoreArgs& a, hipStream_t st);           // gather + dot ("gather+score kernel")
int launch_loss(const ScoreArgs& a, hipStream_t st);
int launch_score_bwd(const ScoreArgs& a, hipStream_t st);

// bf16x3 weight FRAGMENT STREAMS of the fused per-replica kernels (mlp_fused.hip): every fp32 weight is split into three
// bf16 values w = hi + mid + lo (3 x 8 = 24 mantissa bits: exact) and stored in the order the kernels' waves consume them —
// one product step = 3 planes x 64 lanes x 16 bytes (a v_mfma_f32_32x32x16_bf16 A fragment per plane: lane (l31, h) holds
// 8 reduction elements of weight row phi(l31), phi(p) = 16*((p>>2)&1) + 4*(p>>3) + (p&3): an accumulator lane then owns 16
// CONSECUTIVE features).  Streams (step index -> content, d = 128):
//   fwd_wo [4 nb][8 t]            rows 32nb+phi of Wo,  k = 16t + 8h + e
//   fwd_ff [F/32 fb][16]          steps 0-7:  rows 32fb+phi of W1, k = 16t + 8h + e
//                                 steps 8-15: (t = (s-8)>>2, nb = (s-8)&3) rows 32nb+phi of W2, k = feature 32fb + 16h + 8t + e
//   bwd_ff [F/32 fb][16]          steps 0-7:  rows 32fb+phi of W2^T (features), k = output 16t + 8h + e
//                                 steps 8-15: rows 32nb+phi of W1^T (inputs), k = feature 32fb + 16h + 8t + e
//   bwd_wo [4 kb][8 t]            rows 32kb+phi of Wo^T, k = 16t + 8h + e
// Re-split at the start of every forward (extra workgroups of the embed launch): whoever changed the weights in between
// is picked up.
struct WSplit {
  const float* w[3]; int rows[3], cols[3];     // final_linear [d][d], w_1 [F][d], w_2 [d][F]  (nn.Linear [out][in])
  uint16_t* fwd_wo; uint16_t* fwd_ff; uint16_t* bwd_ff; uint16_t* bwd_wo;
  int on;
  // round 5: the K / V projection weights of a ONE-layer encoder for the fused projection + attention forward
  // (kvq_attn_fwd_kernel, attn_sq1.hip):  fwd_kv [8 nb][8 t]  rows 32(nb&3)+phi of linear_keys (nb < 4) / linear_values
  // (nb >= 4), k = 16t + 8h + e — the fwd_wo layout; null wkv[0]: not produced
  const float* wkv[2]; uint16_t* fwd_kv;
  // ... and for the K / V input gradient fused into the replica attention backward (AttnArgs::kvb_stream):
  //   bwd_kv [2 hg][4 nb][8 t]   rows 32nb+phi of [Wk^T | Wv^T] restricted to head group hg: k = 16t + 8h + e walks the group's
  //   64 linear_keys rows (t < 4: row 64hg + 16t + 8h + e) then its 64 linear_values rows
  uint16_t* bwd_kv;
};
// One thread = one 16-byte fragment chunk (8 reduction elements of one weight row) of one product step, all three planes.
// The streams form two SETS: 0 = what the forward's later launches read (fwd_wo, fwd_ff, fwd_kv), 1 = what only the backward
// reads (bwd_ff, bwd_wo, bwd_kv).  Set 0 rides in the embed launch; set 1 rides in the launch behind it when that is the fused
// projection + attention kernel (KvqArgs::split), else in the embed launch too.
__host__ __device__ inline int wsplit_chunks(const WSplit& W, int /*set: both sets have the same size*/) {
  const int d = W.cols[0], F = W.rows[1];
  return (d * d + 2 * d * F) / 8 + (W.wkv[0] ? 2 * d * d / 8 : 0);
}
#if defined(__HIPCC__)
__device__ inline void wsplit_chunk(const WSplit& W, int set, int q) {
  const int d = W.cols[0], F = W.rows[1];
  const int n_wo = d * d / 8, n_ff = 2 * d * F / 8, n_kv = W.wkv[0] ? 2 * d * d / 8 : 0;
  if (q >= n_wo + n_ff + n_kv) return;
  uint16_t* dst;
  int which;                                   // 0 fwd_wo, 1 fwd_ff, 2 bwd_ff, 3 bwd_wo, 4 fwd_kv, 5 bwd_kv
  if (q < n_wo) { which = set ? 3 : 0; dst = set ? W.bwd_wo : W.fwd_wo; }
  else if (q < n_wo + n_ff) { which = set ? 2 : 1; q -= n_wo; dst = set ? W.bwd_ff : W.fwd_ff; }
  else { which = set ? 5 : 4; q -= n_wo + n_ff; dst = set ? W.bwd_kv : W.fwd_kv; }
  const int step = q >> 6, ln = q & 63, l31 = ln & 31, hh = ln >> 5;
  const int phi = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
  const float* src; int stride;                // element e of the chunk = src[e * stride]
  if (which == 0) { const int nb = step >> 3, t = step & 7; src = W.w[0] + (size_t)(32 * nb + phi) * d + 16 * t + 8 * hh; stride = 1; }
  else if (which == 4) { const int nb = step >> 3, t = step & 7; src = W.wkv[nb >> 2] + (size_t)(32 * (nb & 3) + phi) * d + 16 * t + 8 * hh; stride = 1; }
  else if (which == 5) {                      // rows = input features (32 nb + phi), k = the head group's 64 K rows then its 64 V rows
    const int hg = step >> 5, nb = (step >> 3) & 3, t = step & 7;
    src = W.wkv[t >> 2] + (size_t)(64 * hg + 16 * (t & 3) + 8 * hh) * d + 32 * nb + phi; stride = d;
  }
  else if (which == 3) { const int kb = step >> 3, t = step & 7; src = W.w[0] + (size_t)(16 * t + 8 * hh) * d + 32 * kb + phi; stride = d; }
  else {
    const int fb = step >> 4, r = step & 15;
    if (r < 8) {
      if (which == 1) { src = W.w[1] + (size_t)(32 * fb + phi) * d + 16 * r + 8 * hh; stride = 1; }          // W1 rows
      else { src = W.w[2] + (size_t)(16 * r + 8 * hh) * F + 32 * fb + phi; stride = F; }                    // W2^T rows
    } else {
      const int t = (r - 8) >> 2, nb = (r - 8) & 3, kf = 32 * fb + 16 * hh + 8 * t;
      if (which == 1) { src = W.w[2] + (size_t)(32 * nb + phi) * F + kf; stride = 1; }                      // W2 rows, k = features
      else { src = W.w[1] + (size_t)kf * d + 32 * nb + phi; stride = d; }                                  // W1^T rows, k = features
    }
  }
  float x[8];
  if (stride == 1) {
    const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
    x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w; x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = src[(size_t)e * stride];
  }
  uint32_t wd[3][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint16_t hb[2][3];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float xv = x[2 * i + e];
      const __bf16 bh = (__bf16)xv;
      float r = xv - (float)bh;
      const __bf16 bm = (__bf16)r;
      r -= (float)bm;
      const __bf16 bl = (__bf16)r;
      hb[e][0] = __builtin_bit_cast(uint16_t, bh); hb[e][1] = __builtin_bit_cast(uint16_t, bm); hb[e][2] = __builtin_bit_cast(uint16_t, bl);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) wd[pl][i] = (uint32_t)hb[0][pl] | ((uint32_t)hb[1][pl] << 16);
  }
#pragma unroll
  for (int pl = 0; pl < 3; ++pl)
    *reinterpret_cast<uint4*>(dst + ((size_t)(step * 3 + pl) * 64 + ln) * 8) = make_uint4(wd[pl][0], wd[pl][1], wd[pl][2], wd[pl][3]);
}
#endif
struct EmbedArgs {
  int B, Q, L, S, d;
  int64_t P, V;
  int tem;                       // 1: build the [B,S,d] sequence; 0 (QEM): query only
  int fs;                        // 1: FS encoder (row 0 of x is written by the FS projection), 0: AVG
  // FS projection folded into this launch when set: query_emb = tanh(fs_w . mean + fs_b) (text_encoder.py:39) as a
  // per-row mat-vec against the L2-resident [d,d] weight (a GEMM launch of its own cost 15 us for 12.6 MFLOP at C2)
  const float* fs_w; const float* fs_b;
  // optional (tem): the list of VALID rows of x — row b*S of every sequence (the query) and rows b*S+1+l with
  // u_item_idxs[b][l] != P, ascending — and its length.  Padded positions (69 % of the rows at C2) carry exact zeros
  // through the K/V backward, so the row-list GEMMs (GemmProblem::ridx) skip them.
  int32_t* vrows; int32_t* vcount;
  int samp_wgs;                  // filled by the launcher: sampling workgroups in the grid (the list ones follow)
  int use_pos;
  const int64_t* qw; const int64_t* ui;
  const float* word_emb; const float* hist_tab; const float* pe;
  DropSpec drop_fs;
  float* qmean_d;                // [B,d] mean after FS dropout
  float* query_emb;              // [B,d] written here only for AVG
  float* x;                      // [B,S,d]
  // optional: the step's two negative draws ride in this launch as extra workgroups (nothing here depends on them; the
  // score kernel that reads them runs much later) instead of a launch of their own in front of it
  const float* samp_prob; const int32_t* samp_alias; int64_t* samp_items; int64_t* samp_words;
  int samp_nitem, samp_nword; uint32_t samp_step, samp_k0, samp_k1;
  // optional: the word tasks of the loss ride in this launch too (ScoreArgs, folded form); word_wgs / list_wgs are
  // filled by the launcher (grid = B gather + samp_wgs + list_wgs + word_wgs workgroups)
  int fold_words; ScoreArgs sc; int word_wgs, list_wgs;
  WSplit split; int split_wgs;   // optional: re-split the fused kernels' weights (WSplit); split_wgs filled by the launcher
  int split_fwd_only;            // set 1 of the streams (backward-only) is re-split by the launch behind this one (KvqArgs::split)
  uint32_t* clear_word;          // optional: FOUR words (a 64-bit ticket of a later kernel, a list counter, one spare) set to 0 by the launch
  int32_t* zero_i32; int zero_n, zero_wgs;   // optional: int32 words to clear (the review transformer's word counters); zero_wgs filled by the launcher
};
int launch_embed_fwd(const EmbedArgs& a, hipStream_t st);

struct LnFwdArgs {
  const float* x; int ldx; float* y; int ldy; float* stats;  // stats [rows,2] = mean, rstd
  const float* g; const float* b; int rows, d; float eps;
};
int launch_ln_fwd(const LnFwdArgs& a, hipStream_t st);

struct LnBwdArgs {
  const float* dy; int lddy;     // grad wrt LN output
  const float* x; int ldx;       // LN input
  const float* stats; const float* g;
  int rows, d;
  ResMap res;                    // added to dx after the LN backward
  float* dx; int lddx;
  float* out2;                   // optional: dx * dropout(site) (ld = d)
  DropSpec drop2;
  float* colsum;                 // optional: += column sums of (out2 if out2 else dx)
  float* dgamma; float* dbeta;   // += (atomics)
  float* partial;                // optional [workgroups][3][d]: column sums are parked here instead (see ColFoldList)
};
int launch_ln_bwd(const LnBwdArgs& a, hipStream_t st);
int ln_bwd_blocks(int rows);

// Column sums a kernel parked per workgroup ({dgamma, dbeta, colsum} of an LN backward) instead of sending
// workgroups x 3d atomics to the same 3d addresses (measured ~6 us per LN backward at C2).  Nothing in the backward
// reads these gradients, so the list rides to the last launch of the backward (embed_scatter), whose extra
// workgroups add each column up once.
#define PS_MAX_COLFOLD 8
struct ColFold { const float* partial; int nblk, d; float* dst[3]; };
struct ColFoldList { ColFold e[PS_MAX_COLFOLD]; int n; };

struct AttnArgs {
  int n_in, fan, H, S, Sq, d, dh, qpos;
  FDiv fS, fd, fdh, fd4, fHS, fHQS;  // fast dividers (filled by attn_finish)
  int jc;                        // replicas per LDS chunk (sq1 kernels; set by the launcher)
  int seq_div;                   // batch row = n_in_index / seq_div
  int L; int64_t P; const int64_t* ui;   // key-padding mask source (u_item_idxs != P)
  const float* valid;            // or, if set: valid[seq * S + s] != 0 (RTM: review id != pad)
  const float* kp; const float* vp; const float* qp;   // [n_in*S,d] x2, [n_in*Sq,d] (q pre-scaled)
  float* attn;                   // [n_in,H,Sq,S] softmax (pre-dropout)
  float* ctx;                    // [n_in*fan*Sq, d]
  DropSpec drop;
  // backward only
  const float* dctx;             // [n_in*fan*Sq, d]
  float* dq; int lddq;           // grad wrt the un-scaled query linear output
  float* dkv; int lddkv;         // dK at col 0.., dV at col d..
  float* dbq; float* dbk; float* dbv;   // += (atomics)
  float* bias_part;              // optional [n_in][3][d]: the sq1 backward parks {dbq, dbk, dbv} here instead (ColFoldList)
  float qscale;                  // 1/sqrt(dh)
  // optional (sq1 backward, d == 128, two head groups): the input gradient of the query projection folded into the
  // kernel's tail — each head-group workgroup multiplies its 64 dq values into Wq (rows prefetched at kernel start) and
  // writes one partial row  dxq_part[group][sequence][d];  the dX GEMM's fan-in epilogue adds both (ResMap::extra/extra2)
  const float* wq; float* dxq_part;
  // ... together with the fan-in residual of `out = dropout(context) + inputs`: the sum over the sequence's replicas of
  // fanin_src[(b*fan + j)*128 + i] (d y1), each head group adding its own 64 columns to its partial row
  const float* fanin_src;
  uint32_t* sig; uint32_t sigval;   // backward launchers: a pending side-stream fork signalled by this launch (common.h, fork_signal)
  // round 5 (replica backward, d = 128, dQ.Wq folded): the K / V input gradient  d x = dK.Wk + dV.Wv  rides in the same launch.
  // Each head-group workgroup multiplies ITS 64 dK and 64 dV columns into the matching weight rows (WSplit::bwd_kv, fragment
  // order, re-split by the forward's embed launch) and writes one PARTIAL row per valid position into dxp[group]; the row of
  // the query position also takes the group's dQ.Wq + fan-in row (dxq_part is then not written).  The consumer (embed
  // scatter) adds the two partials.  Replaces the dX GEMM launch of the step's dependent chain.
  const uint16_t* kvb_stream; float* dxp[2];
};
inline void attn_finish(AttnArgs& a) {
  a.fS = make_fdiv(a.S); a.fd = make_fdiv(a.d); a.fdh = make_fdiv(a.dh); a.fd4 = make_fdiv(a.d / 4);
  a.fHS = make_fdiv(a.H * a.S); a.fHQS = make_fdiv((a.H / 4 > 0 ? a.H / 4 : 1) * a.S);
}
int launch_attn_fwd(const AttnArgs& a, hipStream_t st);
int launch_attn_bwd(const AttnArgs& a, hipStream_t st);
// K / V / Q projections of a one-layer encoder + the replica attention of its one consumed position as ONE launch, a workgroup
// per sequence (round 5; attn_sq1.hip): `at` as for launch_attn_fwd_wf (kp / vp / qp are OUTPUTS here), x = the [n_in, S, 128]
// encoder input the embed launch wrote, kv_stream = WSplit::fwd_kv of the same embed launch.
struct KvqArgs {
  AttnArgs at;
  const float* x;
  const uint16_t* kv_stream;
  const float* bk; const float* bv; const float* wq; const float* bq;
  float* kp; float* vp; float* qp;
  uint32_t* amask;
  unsigned long long* stamp;     // diagnostic build: 8 s_memrealtime stamps per workgroup (tools/kvq_wg_times.py)
  WSplit split;                  // optional (on): the backward-only streams are re-split by extra workgroups of this launch
};
bool kvq_attn_fits(const AttnArgs& a);     // d = 128, 8 heads, <= 32 positions, 4..24 replicas, query position 0
int launch_kvq_attn_fwd(const KvqArgs& a, hipStream_t st);
// last-layer form (Sq == 1): one workgroup per sequence, all heads, replicas share K/V in LDS (attn_sq1.hip)
int launch_attn_fwd_sq1(const AttnArgs& a, hipStream_t st);
int launch_attn_bwd_sq1(const AttnArgs& a, hipStream_t st);
bool attn_sq1_fits(const AttnArgs& a);
bool attn_w1_fits(const AttnArgs& a);      // one wave per sequence (fan == 1, d 64 / 128)
int launch_attn_bwd_w1(const AttnArgs& a, bool pads_unread, hipStream_t st);
int launch_attn_fwd_w1(const AttnArgs& a, hipStream_t st);
bool attn_wf_fits(const AttnArgs& a);      // one wave per (sequence, four heads), dropout replicas inside (S <= 32)
int launch_attn_fwd_wf(const AttnArgs& a, uint32_t* amask, hipStream_t st);
int launch_attn_bwd_wf(const AttnArgs& a, const uint32_t* amask, bool pads_unread, hipStream_t st);
bool attn_bwd_wf_two_partials(const AttnArgs& a);   // ... and its backward leaves two partial dQ.Wq rows per sequence
int attn_sq1_split(const AttnArgs& a);   // head groups (workgroups) per sequence the sq1 kernels will use

struct EmbedBwdArgs {
  int B, Q, L, S, d;
  int64_t P, V;
  int tem;
  const int64_t* qw; const int64_t* ui;
  const float* dx;               // [B,S,d] (tem)
  const float* dx2;              // optional: a second partial of the same shape, added to dx wherever dx is read (AttnArgs::dxp)
  const float* dqmean_d;         // [B,d] grad wrt the post-dropout query mean
  DropSpec drop_fs;
  float* g_hist_tab; float* g_word_emb;
  // optional: weight gradient of the FS query projection folded into the same launch (extra workgroups):
  // g_fs_w[o][i] += sum_b fw_dy[b][o] * fw_x[b][i]   (text_encoder.py:38, f_W)
  const float* fw_dy; const float* fw_x; float* g_fs_w;
  // optional: the whole FS backward folded into this launch (text_encoder.py:38-39).  One workgroup per batch row takes
  // dqpre = dqe * (1 - qe^2), d mean = dqpre . f_W as a mat-vec against the L2-resident weight and scatters it to the
  // row's query words; the f_W workgroups above recompute dqpre on the fly (fw_dy is then unused) and also add up
  // g_fs_b.  Replaces a tanh-backward launch and a [B,d]x[d,d] GEMM launch on the tail of the backward.
  const float* fsb_dqe; int fsb_lddqe; const float* fsb_qe; const float* fsb_w; float* g_fs_b;
  const float* fsb_dqe2; float fsb_k2;   // filled by the launcher: the second partial of d query_emb (dx2) and its weight (1, or 0 with fsb_dqe2 = fsb_dqe)
  // optional: the row workgroups store dqpre [B,d] here and add the bias gradient themselves (atomics); the f_W weight gradient is then
  // NOT computed by this launch (the caller runs it as a GEMM over dqpre and fw_x)
  float* fsb_dqpre_out;
  ColFoldList fold;              // parked column sums to add up (n = 0: none)
  float* det_dm;                 // deterministic mode + fused FS backward: [B,d] buffer for the rows' d mean (scattered by the sole-owner pass)
  uint32_t* sig; uint32_t sigval; // a pending side-stream fork signalled by this launch (common.h, fork_signal)
};
int launch_embed_scatter(const EmbedBwdArgs& a, hipStream_t st);
// deterministic  table[keys[t]] += src[t * ld .. + d)  (keys[t] < 0: no task): one owner half-wave per table row, tasks in order
int launch_rows_scatter_det(const int32_t* keys, int ntask, const float* src, int64_t ld, int d, float* table, hipStream_t st,
                            const float* scale = nullptr, const int32_t* rowidx = nullptr);   // task t: scale[t] * row rowidx[t] of src

// dqpre = dqe * (1 - qe^2), dfb += colsum(dqpre)
// out[b][c] = sum_j src[(b * fan + j) * ld + c]: the replicas' fan-in into one row per sequence (row-list dX product, tem.hip)
int launch_fanin_sum(const float* src, int ld, int n_in, int fan, int d, float* out, hipStream_t st);
int launch_tanh_bwd(const float* dqe, int lddqe, const float* qe, float* dqpre, float* dfb, int rows, int d,
                    hipStream_t st);

// Prologue of a graph-replayed step: the only kernel whose arguments change from call to call.  Copies the caller's index
// tensors into workspace-resident buffers, draws the negatives there (if a sampler is given) and publishes the Philox
// step word; every later kernel of the step reads those fixed locations.
struct StageArgs {
  const int64_t* src[6]; int64_t* dst[6]; int n[6];   // query words, history, target, pv words, neg items, neg words
  uint32_t step; uint32_t* step_word;
  const float* prob; const int32_t* alias;            // sampler (null: negatives are among src)
  int nitem, nword; int64_t P, V; uint32_t k0, k1;
};
int launch_stage(const StageArgs& a, hipStream_t st);
const void* stage_kernel_handle();
const void* loss_kernel_handle();
int score_fwd_blocks(const ScoreArgs& a);

int launch_sample(const PsTemDesc& d, const float* alias_prob, const int32_t* alias_idx, int64_t* neg_items,
                  int64_t* neg_words, hipStream_t st);

// ---- fused per-replica tail of the last encoder layer (mlp_fused.hip; d == 128 only)
struct MlpFwdArgs {
  int M, F;                       // replica rows, hidden width (multiple of 128)
  int fan, S, qpos;               // residual source row of replica m: (m / fan) * S + qpos of xin
  const float* ctx; const float* xin;
  const float *wo, *bo, *g1, *be1, *w1, *b1, *w2, *b2, *gf, *bef;
  DropSpec drop_ctx, drop_ff1, drop_ff2;
  float *y1, *ln1, *st1, *a1, *h1, *y2, *stf, *enc;
  int fold_score; ScoreArgs sc;   // item scoring + loss in the epilogue (ScoreArgs, folded form); M = B*(K+1)
  WSplit x3;                      // bf16x3 fragment streams of wo / w1 / w2
  unsigned long long* stamp;      // diagnostics: workgroup 0's waves record s_memtime at their phase boundaries (ps_debug_set_stamp_buffer)
};
int launch_mlp_fwd_fused(const MlpFwdArgs& a, hipStream_t st);
bool mlp_fwd_can_fold_score(int M, int F, int d);   // the wave-specialised kernel will serve this shape
bool mlp_x3_enabled(int F);                          // the fused kernels serve this hidden width (F = 256, 512, 1024; they need WSplit)
bool mlp_fused_serves(int d, int F);                 // ... and this model width (d = 128)
int64_t mlp_x3_floats(int d, int F);                 // workspace floats of the fragment streams (forward + backward)
bool ps_fusion_enabled();

// ---- backward of the same tail as ONE kernel (mlp_fused.hip; d == 128, parked column sums):
//   final-LN backward -> (. W2, gelu', dropout) -> (. W1) -> FF-LN backward (+ residual) -> dropout -> (. Wo)
// replaces 2 LayerNorm-backward + 3 dX GEMM launches; everything the weight gradients read is still written once.
struct MlpBwdArgs {
  int M, F;
  const float* denc;                                   // [M,128] grad wrt enc (unused when item_scores is set)
  // optional: d enc computed from the scores instead (TEM with replicas: row m = (b, j), M = B*(K+1))
  const float* item_scores; const int64_t* target; const int64_t* neg_items; const float* product_emb;
  int B, K, pos_weight; int64_t P; float scale; const float* scale_dev;
  const float* y2; const float* stf; const float* gf;  // final LN: input, {mean, rstd}, gamma
  const float* y1; const float* st1; const float* g1;  // FF LN
  const float* a1;                                     // [M,F] pre-activation of the hidden layer
  const float *wo, *w1, *w2;
  DropSpec drop_ctx, drop_ff1, drop_ff2;
  float* do2;                                          // [M,128] d y2 after the FF2 dropout  (A of dW2)
  float* da1;                                          // [M,F]   grad wrt a1                 (A of dW1)
  float* dy1;                                          // [M,128] grad wrt y1 (fan-in residual of the layer input)
  float* dout;                                         // [M,128] dy1 after the ctx dropout (== dy1 without dropout)
  float* dctx;                                         // [M,128]
  float* part_f;                                       // [workgroups][3][128] {dgamma_f, dbeta_f, colsum -> b2}
  float* part_1;                                       // [workgroups][3][128] {dgamma_1, dbeta_1, colsum -> bo}
  float* part_b1;                                      // [mlp_bwd_b1_rows()][3][F] slot 0: colsum -> b1
  WSplit x3;                                           // bf16x3 fragment streams (the bwd_* ones are read here)
  uint32_t* sig; uint32_t sigval;                      // a pending side-stream fork signalled by this launch (common.h, fork_signal)
  unsigned long long* stamp;                           // diagnostics: as MlpFwdArgs::stamp (the backward's slots follow the forward's 128)
};
int launch_mlp_bwd_fused(const MlpBwdArgs& a, hipStream_t st);
int mlp_bwd_fused_blocks(int M);
int mlp_bwd_b1_rows(int M, int F);   // rows of MlpBwdArgs::part_b1 the kernel parks (ColFold::nblk)
