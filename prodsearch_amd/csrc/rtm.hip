// rtm.hip — the RTM (review_transformer) ranking-loss step: ProductRanker of the reference
// (models/ps_model.py:53-370) with the pv (models/PV.py) and pvc (models/PVC.py) review encoders.
//
// Sequences are n = b*J + j (J = 1+K in training: j = 0 positive, j = 1+k negative k; J = C in
// eval), S = R+1 positions: [query, R reviews].  Unlike TEM every sequence is distinct, so the
// encoder (shared layer loops of tem.hip via encoder.h) runs on B*J sequences with no replica
// fan-out; only position 0 is consumed (TransformerEncoder.forward, transformer.py:90-98), so the
// last layer is the one-query-row form.  RTM-specific kernels here:
//   rtm_embed      review vectors (pv: row gather; pvc: masked mean of <= WL word rows per review with
//                  Philox token corruption, PVC.py:46-61 — the bandwidth-heavy gather of config 4),
//                  dropout, segment embedding, key mask, positional add -> x[B*J, S, d]
//   rtm_score      wo . enc + bias (transformer.py:95-96)
//   rtm_pv_fwd     PV word-prediction logits (PV.py:59-66 / PVC.py:83-91), gather + dot per task
//   rtm_loss       weighted BCE over products (ps_model.py:341-356) + PV loss (:277-280), one block
//   *_bwd          their backward: fp32-atomic scatter-adds into the dense table gradients
#include "encoder.h"
#include "rowwise.h"
#include <string.h>

#define SITE_REV_PV 0x200u
#define SITE_REV_POS 0x201u
#define SITE_REV_NEG 0x202u
#define SITE_TOK_POS 0x203u
#define SITE_TOK_NEG 0x204u

struct RtmWs {
  int Bseq, S, J;
  int64_t qmean, query_emb, valid, vec, cnt, scores, weight, pv_scores, pv_terms, nvalid, dvec, dqe, dqpre, dqmean;
  int64_t loss_blk;         // the 64-bit loss / arrival word of rtm_score_kernel (cleared by the query-encoder launch)
  int64_t glist;            // int32 list of the review-row groups that hold at least one real review (rtm_grouplist_kernel)
  int64_t hist;             // int32 [RTM_HIST_G][V]: per-workgroup word histograms -> exclusive prefixes (rtm_hist_kernel / rtm_hist_scan_kernel)
  int64_t seqcnt;           // int32 [Bseq]: valid positions per sequence (rtm_embed_kernel -> rtm_rowlist_kernel)
  int64_t wrank;
  int64_t wcnt, woff, wcur, wl;   // pvc backward: inverted index word -> review slots (int32 arrays; wl: int2 {slot, word})
  int64_t segpart;          // [ceil(Bseq*S / 64)][3][d] parked segment-embedding gradient partials (rtm_embed_bwd_kernel)
  int64_t raw, yfs, dpre, dmean;                // fs review encoder: [Bseq*R, d] dropped means, projections, their gradients
  int64_t enc_base;         // the shared encoder workspace (Ws) starts here
  int64_t total;
};

struct RtmK {              // kernel-side view of one call
  int B, J, K, R, S, Q, W, WL, d;
  int64_t V, RC;
  int pvc, use_pos, use_seg, pos_weight, train_pv, training, eval;
  FDiv fJ, fS, fK1;
  DropSpec d_pv, d_pos, d_neg, t_pos, t_neg;
  // batch
  const int64_t *pos_r, *neg_r, *pos_seg, *neg_seg, *pos_words, *neg_words_rev, *pos_pvc, *neg_pvc, *neg_word_idxs;
  const int64_t *pos_u, *neg_u, *pos_i, *neg_i;   // per-position user / item ids (use_user_emb / use_item_emb), else null
  int64_t U, PI;                                  // their pad ids (user_size, product_size)
  const float *user_emb, *item_emb; float *g_user_emb, *g_item_emb;
  const uint8_t* pos_masks;
  // fs / avg review encoders (ps_model.py:301-305): the word-mean path of pvc with the BATCH's word masks deciding which
  // words count (text_encoder.py:6-16; pvc uses idx != pad, PVC.py:58) and no token corruption.  fs additionally
  // projects the mean: tanh(f_W . dropout(mean) + b) (text_encoder.py:32-40) — the embed kernels then leave the
  // dropped means in `raw` [B*R + B*K*R, d] (positive rows first), a GEMM writes `yfs`, rtm_fs_finish builds x.
  const uint8_t *wmask_pos, *wmask_neg;
  float *raw; const float* yfs; const float* dmean;   // dmean: backward, grad wrt raw (same layout)
  // tensors
  const float *word_emb, *table, *seg_emb, *pe, *wo_w, *wo_b;
  // workspace
  float *query_emb, *x, *valid, *vec, *cnt, *enc, *scores, *weight, *pv_scores, *pv_terms, *nvalid;
  int32_t *seqcnt, *vrows, *vcount;   // valid-row list of x (GemmProblem::ridx): per-sequence counts, rows, length
  int count_fwd;                      // rtm_embed4_kernel counts them (the counters were cleared by the query-encoder launch)
  int det;                            // deterministic mode (ps_deterministic): no fp32 atomic whose order could differ between runs
  unsigned long long* stamp;          // diagnostics (PS_RTM_STAMP=1, tools/rtm_stamps.py): phase stamps of one workgroup of rtm_embed4_kernel
  float* loss3;
  // backward
  float scale; const float* scale_dev;
  const float *dx; float *denc, *dvec, *dqe;
  float *g_word_emb, *g_table, *g_seg_emb, *g_wo_w, *g_wo_b;
  // pvc backward through an inverted index (word -> review slots) instead of one atomic row per word occurrence
  float* gs;                  // = dx, rewritten in place: row (n, s) becomes the gradient each of its words receives
  int *wcnt, *woff, *wcur;
  int2* wl;                   // the occurrence list: {slot, word}, one 8-byte store per occurrence
  int* wrank;                 // [B*R + B*K*R][WL] rank of a counted word among its word's occurrences, -1: not counted (count_fwd)
  int* hist; int hist_rows;   // LDS-histogram index (rtm_hist_kernel): [RTM_HIST_G][V]; review rows per histogram workgroup
};

// waves of rtm_embed_bwd_kernel: EB_GROUPS four-review groups per wave on each side, EB_QSEQ query slots per wave
#define EB_GROUPS 4
#define EB_QSEQ 16
static inline int rtm_eb_waves(int B, int K, int R, int* npos_w, int* nneg_w) {
  const int np = ps_cdiv((int64_t)B * R, 4 * EB_GROUPS), nn = ps_cdiv((int64_t)B * K * R, 4 * EB_GROUPS);
  if (npos_w) *npos_w = np;
  if (nneg_w) *nneg_w = nn;
  return np + nn + ps_cdiv((int64_t)B * (K + 1), EB_QSEQ);
}
#define RTM_HIST_G 256          // workgroups (= partitions of the review rows) of the LDS-histogram index
#define RTM_HIST_MAXV 38000     // vocabulary sizes whose histogram fits one workgroup's LDS (4 B per word, 160 KB)
static inline int64_t rtake(int64_t& cur, int64_t n) { int64_t o = cur; cur += (n + 3) & ~(int64_t)3; return o; }

static int rtm_check(const PsRtmDesc& D) {
  PS_REQUIRE(D.B > 0 && D.K >= 0 && D.R > 0 && D.Q > 0 && D.d > 0, "rtm desc: bad sizes");
  PS_REQUIRE(D.d % 32 == 0 && D.d <= 512 && D.d / 4 <= 64 * 2, "rtm desc: embedding_size %d", D.d);
  PS_REQUIRE(D.R + 1 <= 64, "rtm desc: %d reviews per sequence (S <= 64)", D.R);
  PS_REQUIRE(D.review_encoder >= PS_RENC_PV && D.review_encoder <= PS_RENC_AVG, "rtm desc: review encoder %d", D.review_encoder);
  PS_REQUIRE(D.review_encoder == PS_RENC_PV || D.WL > 0, "rtm desc: word-mean review encoders need WL");
  PS_REQUIRE(D.review_encoder == PS_RENC_PV || D.review_encoder == PS_RENC_PVC || !D.train_pv,
             "rtm desc: the fs / avg review encoders have no PV loss (ps_model.py:264: train_pv only applies to pv / pvc)");
  PS_REQUIRE(D.dropout >= 0.f && D.dropout < 1.f && D.corrupt_rate >= 0.f && D.corrupt_rate < 1.f, "rtm desc: rates");
  return PS_OK;
}

static PsTemDesc enc_desc(const PsRtmDesc& D, int J) {
  PsTemDesc E;
  memset(&E, 0, sizeof(E));
  E.B = D.B * J; E.K = 0; E.L = D.R; E.Q = 1; E.W = 0; E.C = 0;
  E.d = D.d; E.H = D.H; E.F = D.F; E.n_layers = D.n_layers;
  E.product_size = 1; E.vocab_size = 2;
  E.model = PS_MODEL_TEM; E.query_encoder = PS_QENC_AVG;
  E.use_pos_emb = D.use_pos_emb; E.training = D.training; E.dropout = D.dropout; E.seed = D.seed; E.step = D.step;
  return E;
}

static int rtm_make_ws(const PsRtmDesc& D, bool eval, RtmWs& r, Ws& w, PsTemDesc& E) {
  TRY(rtm_check(D));
  const int J = eval ? D.C : D.K + 1;
  PS_REQUIRE(J > 0, "rtm: no sequences per row");
  E = enc_desc(D, J);
  if (eval) E.training = 0;
  r.J = J; r.Bseq = D.B * J; r.S = D.R + 1;
  const int d = D.d;
  int64_t cur = 0;
  r.qmean = rtake(cur, (int64_t)D.B * d);
  r.query_emb = rtake(cur, (int64_t)D.B * d);
  r.valid = rtake(cur, (int64_t)r.Bseq * r.S);
  r.vec = rtake(cur, (int64_t)D.B * D.R * d);
  r.cnt = rtake(cur, (int64_t)r.Bseq * D.R);
  r.scores = rtake(cur, r.Bseq);
  r.weight = rtake(cur, r.Bseq);
  const int64_t npv = (int64_t)D.B * D.R * (D.W > 0 ? D.W : 1) * (D.K + 1);
  r.pv_scores = rtake(cur, npv);
  r.pv_terms = rtake(cur, npv);
  r.nvalid = rtake(cur, 4);
  r.seqcnt = rtake(cur, (int64_t)r.Bseq + 4);
  r.loss_blk = rtake(cur, 4);         // [0..1] the loss word, [2] the valid-group count of rtm_grouplist_kernel
  r.glist = eval ? 0 : rtake(cur, (int64_t)ps_cdiv((int64_t)D.B * D.R, 4) + ps_cdiv((int64_t)D.B * D.K * D.R, 4) + 4);
  r.dvec = rtake(cur, (int64_t)D.B * D.R * d);
  r.dqpre = rtake(cur, (int64_t)D.B * d);
  r.dqmean = rtake(cur, (int64_t)D.B * d);
  r.dqe = rtake(cur, (int64_t)D.B * d);          // dqe and wcnt are adjacent: the backward zeroes both with ONE memset
  r.wcnt = r.woff = r.wcur = r.wl = r.wrank = r.hist = 0;
  r.raw = r.yfs = r.dpre = r.dmean = 0;
  if (!eval && D.review_encoder != PS_RENC_PV) {
    r.wcnt = rtake(cur, D.vocab_size + 1);        // [V] occurrence counts + the segment allocator's running total
    r.woff = rtake(cur, D.vocab_size + 1);
    r.wcur = rtake(cur, D.vocab_size);
    r.wl = rtake(cur, 2 * (int64_t)r.Bseq * D.R * D.WL);
    r.wrank = rtake(cur, (int64_t)r.Bseq * D.R * D.WL);
    if (D.vocab_size <= RTM_HIST_MAXV) r.hist = rtake(cur, (int64_t)RTM_HIST_G * D.vocab_size);
  }
  if (!eval && D.review_encoder == PS_RENC_FS) {     // (behind the index arrays: dqe .. wcnt must stay one contiguous memset)
    const int64_t nr = (int64_t)r.Bseq * D.R * d;
    r.raw = rtake(cur, nr); r.yfs = rtake(cur, nr); r.dpre = rtake(cur, nr); r.dmean = rtake(cur, nr);
  }
  r.segpart = eval ? 0 : rtake(cur, (int64_t)ps_cdiv(rtm_eb_waves(D.B, D.K, D.R, nullptr, nullptr), 4) * 3 * d);
  r.enc_base = cur;
  TRY(make_ws(E, w));
  r.total = cur + w.total;
  return PS_OK;
}

extern "C" int ps_rtm_workspace_floats(const PsRtmDesc* desc, int32_t eval, int64_t* total) {
  PS_REQUIRE(desc && total, "rtm workspace: null argument");
  RtmWs r; Ws w; PsTemDesc E;
  TRY(rtm_make_ws(*desc, eval != 0, r, w, E));
  *total = r.total;
  return PS_OK;
}

// offsets (floats) of the intermediates the parity tests compare stage by stage
extern "C" int ps_rtm_workspace_layout(const PsRtmDesc* desc, int32_t eval, PsRtmWsLayout* out) {
  PS_REQUIRE(desc && out, "rtm workspace layout: null argument");
  RtmWs r; Ws w; PsTemDesc E;
  TRY(rtm_make_ws(*desc, eval != 0, r, w, E));
  memset(out, 0, sizeof(*out));
  out->total_floats = r.total; out->Bseq = r.Bseq; out->S = r.S; out->J = r.J;
  out->query_emb = r.query_emb; out->valid = r.valid; out->vec = r.vec; out->cnt = r.cnt; out->scores = r.scores;
  out->weight = r.weight; out->pv_scores = r.pv_scores;
  out->x = r.enc_base + w.x; out->enc = r.enc_base + w.enc; out->dx = r.enc_base + w.dx;
  return PS_OK;
}

__device__ inline int64_t rclamp(int64_t i, int64_t hi) { return i < 0 ? hi : (i > hi ? hi : i); }
// does word slot `off` (= review row * WL + slot) count in its review's mean?
__device__ inline bool word_ok(const RtmK& a, const uint8_t* wm, size_t off, int64_t wi) {
  return (wm ? wm[off] != 0 : wi != a.V - 1) && wi >= 0 && wi < a.V;
}

// ------------------------------------------------------------------ embed forward
// one wave per (sequence, position); lanes: half = lane>>5 picks every other word row, c = lane&31 the float4
// chunk of the row (d <= 128: one chunk per lane; wider rows loop).
__device__ inline void seq_decode(const RtmK& a, int n, int s, int& b, int& j, int64_t& ridx, int& revrow, int& seg,
                                  size_t* spos = nullptr) {
  b = fdiv(n, a.fJ); j = n - b * a.J;
  const int r = s - 1;
  if (a.eval || j > 0) {
    const int jj = a.eval ? j : j - 1, JN = a.eval ? a.J : a.K;
    const size_t base = ((size_t)b * JN + jj);
    ridx = s > 0 ? a.neg_r[base * a.R + r] : 0;
    seg = (int)a.neg_seg[base * a.S + s];
    revrow = (int)(base * a.R + r);
    if (spos) *spos = base * a.S + s;
  } else {
    ridx = s > 0 ? a.pos_r[(size_t)b * a.R + r] : 0;
    seg = (int)a.pos_seg[(size_t)b * a.S + s];
    revrow = b * a.R + r;
    if (spos) *spos = (size_t)b * a.S + s;
  }
}

// valid-row list of x: wave n places its sequence's valid rows behind those of all earlier sequences (their counts come
// from rtm_embed_kernel: Bseq ints, L2 hits) — one short launch that lets the three K/V GEMMs skip the padded
// positions (73 % of the rows on the synthetic C4 batches)
__global__ __launch_bounds__(256) void rtm_rowlist_kernel(const RtmK a) {
  const int lane = threadIdx.x & 63;
  const int n = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nseq = a.B * a.J;
  if (n >= nseq) return;
  int prev = strided_sum_i32<8>(a.seqcnt, n, lane, 64);      // (up to 24 elements per lane: as a plain loop, 24 serial round trips)
  prev = (int)wave_sum((float)prev);                         // < 2^24: exact in fp32
  const bool okl = lane < a.S && a.valid[(size_t)n * a.S + lane] != 0.f;
  const unsigned long long vm = __ballot(okl);
  if (okl) a.vrows[prev + __popcll(vm & ((1ull << lane) - 1ull))] = n * a.S + lane;
  if (n == nseq - 1 && lane == 0) *a.vcount = prev + __popcll(vm);
}

__global__ __launch_bounds__(256) void rtm_embed_kernel(const RtmK a) {
  const int lane = threadIdx.x & 63, half = lane >> 5, c = lane & 31;
  const int slot = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (slot >= a.B * a.J * a.S) return;
  const int n = fdiv(slot, a.fS), s = slot - n * a.S;
  const int d = a.d, nch = d >> 2;
  int b, j, revrow, seg; int64_t ridx; size_t spos;
  seq_decode(a, n, s, b, j, ridx, revrow, seg, &spos);
  const bool pos = !a.eval && j == 0;
  const int64_t rpad = a.RC - 1;
  const bool ok = s == 0 || ridx != rpad;
  if (lane == 0) a.valid[(size_t)n * a.S + s] = ok ? 1.f : 0.f;
  if (s == 0 && a.S <= 64) {        // valid positions of this sequence (lane l looks at position l) for the row list
    bool okl = lane == 0;
    if (lane > 0 && lane < a.S) {
      int b2, j2, rr2, sg2; int64_t rid2;
      seq_decode(a, n, lane, b2, j2, rid2, rr2, sg2);
      okl = rid2 != rpad;
    }
    const int cntv = __popcll(__ballot(okl));
    if (lane == 0) a.seqcnt[n] = cntv;
  }
  // per-position user / item embedding rows (ps_model.py:325-334).  The pad id addresses the table's last row,
  // which is READ like any other (nn.Embedding's padding_idx only stops its gradient) and never updated.
  int64_t uid = -1, iid = -1;
  if (a.user_emb) { uid = (pos ? a.pos_u : a.neg_u)[spos]; if (uid < 0 || uid > a.U) uid = -1; }
  if (a.item_emb) { iid = (pos ? a.pos_i : a.neg_i)[spos]; if (iid < 0 || iid > a.PI) iid = -1; }
  for (int cc0 = 0; cc0 < nch; cc0 += 32) {          // wave-uniform trip count: the shuffles below need every lane
    const int cc = cc0 + c;
    const bool act = cc < nch;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f), vcor = v;
    float cnt = 1.f;
    if (s == 0) {
      if (half == 0 && act) v = *reinterpret_cast<const float4*>(a.query_emb + (size_t)b * d + 4 * cc);
    } else if (ok) {
      const int r = s - 1;
      if (!a.pvc) {
        if (half == 0 && act) v = *reinterpret_cast<const float4*>(a.table + (size_t)rclamp(ridx, rpad) * d + 4 * cc);
        vcor = v;
      } else {
        // masked mean of the review's word rows; corrupted and (positive, train_pv) uncorrupted sums
        const int64_t* words;
        if (pos) words = (a.train_pv ? a.pos_pvc : a.pos_words) + ((size_t)b * a.R + r) * a.WL;
        else words = ((a.train_pv || a.eval) ? a.neg_pvc : a.neg_words_rev) + (size_t)revrow * a.WL;
        const uint8_t* wm = pos ? a.wmask_pos : a.wmask_neg;
        const size_t woff = (size_t)revrow * a.WL;
        const DropSpec& ts = pos ? a.t_pos : a.t_neg;
        // one Philox evaluation per (review, word slot): lane l owns slots l and l+64, the loop reads them by shuffle
        const float tm0 = drop_mult(ts, (uint32_t)revrow, (uint32_t)lane);
        const float tm1 = a.WL > 64 ? drop_mult(ts, (uint32_t)revrow, (uint32_t)(lane + 64)) : 1.f;
        // only the positive sequence under train_pv also needs the UNcorrupted sum; everywhere else a word whose
        // token mask is 0 (90 % of them at the reference's corrupt_rate) contributes nothing and is not fetched
        const bool need_unc = pos && a.train_pv;
        int nw = 0;
        if (a.WL <= 128) {
          // lane l holds word slots l and l+64 (two coalesced loads); the slots that survive the token mask are
          // compacted with ballots, so the row loop runs over ~10 % of the review at the reference's corrupt_rate
          const int64_t wa64 = lane < a.WL ? words[lane] : a.V - 1, wb64 = lane + 64 < a.WL ? words[lane + 64] : a.V - 1;
          const bool va = lane < a.WL && word_ok(a, wm, woff + lane, wa64), vb = lane + 64 < a.WL && word_ok(a, wm, woff + lane + 64, wb64);
          const int wa = va ? (int)wa64 : 0, wb = vb ? (int)wb64 : 0;
          unsigned long long ma = __ballot(va && (need_unc || tm0 != 0.f)), mb = __ballot(vb && (need_unc || tm1 != 0.f));
          // (the ballots are taken by the WHOLE wave, outside the half-wave select: inside it only lanes 0-31 would vote
          // and a review would never count more than 32 words)
          const int nvalid = __popcll(__ballot(va)) + __popcll(__ballot(vb));
          nw = half == 0 ? nvalid : 0;                                            // the halves are summed below
          while (ma | mb) {                                               // 8 word rows in flight per wave
            float4 rowv[4]; float mt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              int l0 = -1, s0 = 0, l1 = -1, s1 = 0;
              if (ma) { l0 = __ffsll((long long)ma) - 1; ma &= ma - 1; } else if (mb) { l0 = __ffsll((long long)mb) - 1; mb &= mb - 1; s0 = 1; }
              if (ma) { l1 = __ffsll((long long)ma) - 1; ma &= ma - 1; } else if (mb) { l1 = __ffsll((long long)mb) - 1; mb &= mb - 1; s1 = 1; }
              const int l = half ? l1 : l0, sl = half ? s1 : s0, src = l < 0 ? 0 : l;
              const int wia = __shfl(wa, src, 64), wib = __shfl(wb, src, 64);
              const float ta = __shfl(tm0, src, 64), tb = __shfl(tm1, src, 64);
              rowv[u] = make_float4(0.f, 0.f, 0.f, 0.f); mt[u] = 0.f;
              if (l >= 0) {
                const int wi = sl ? wib : wia;
                if (act) rowv[u] = *reinterpret_cast<const float4*>(a.word_emb + (size_t)wi * d + 4 * cc);
                mt[u] = sl ? tb : ta;
              }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              v.x += rowv[u].x; v.y += rowv[u].y; v.z += rowv[u].z; v.w += rowv[u].w;
              vcor.x += rowv[u].x * mt[u]; vcor.y += rowv[u].y * mt[u]; vcor.z += rowv[u].z * mt[u]; vcor.w += rowv[u].w * mt[u];
            }
          }
        } else
        for (int w0 = 0; w0 < a.WL; w0 += 8) {                 // 8 word rows in flight per wave
          float4 rowv[4]; float mt[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int w = w0 + 2 * u + half;
            int64_t wi = w < a.WL ? words[w] : a.V - 1;
            rowv[u] = make_float4(0.f, 0.f, 0.f, 0.f); mt[u] = 0.f;
            const float tmw = w < 64 ? __shfl(tm0, w & 63, 64) : (w < 128 ? __shfl(tm1, (w - 64) & 63, 64)
                                                                          : drop_mult(ts, (uint32_t)revrow, (uint32_t)w));
            if (w < a.WL && word_ok(a, wm, woff + w, wi)) {
              if (act && (need_unc || tmw != 0.f))
                rowv[u] = *reinterpret_cast<const float4*>(a.word_emb + (size_t)wi * d + 4 * cc);
              mt[u] = tmw;
              ++nw;
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            v.x += rowv[u].x; v.y += rowv[u].y; v.z += rowv[u].z; v.w += rowv[u].w;
            vcor.x += rowv[u].x * mt[u]; vcor.y += rowv[u].y * mt[u]; vcor.z += rowv[u].z * mt[u]; vcor.w += rowv[u].w * mt[u];
          }
        }
        nw += __shfl_xor(nw, 32, 64);
        cnt = (float)(nw > 0 ? nw : 1);
      }
    }
    // combine the two half-waves
    v.x += __shfl_xor(v.x, 32, 64); v.y += __shfl_xor(v.y, 32, 64); v.z += __shfl_xor(v.z, 32, 64); v.w += __shfl_xor(v.w, 32, 64);
    vcor.x += __shfl_xor(vcor.x, 32, 64); vcor.y += __shfl_xor(vcor.y, 32, 64);
    vcor.z += __shfl_xor(vcor.z, 32, 64); vcor.w += __shfl_xor(vcor.w, 32, 64);
    if (half != 0 || !act) continue;
    float o[4];
    if (s == 0) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    else {
      const float inv = 1.f / cnt;
      float unc[4] = {v.x * inv, v.y * inv, v.z * inv, v.w * inv};
      float cor[4] = {vcor.x * inv, vcor.y * inv, vcor.z * inv, vcor.w * inv};
      const int rr = s - 1;
      if (a.pvc && lane == 0 && cc0 == 0) a.cnt[(size_t)n * a.R + rr] = cnt;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t col = (uint32_t)(4 * cc + e);
        float val;
        if (!a.pvc) {
          val = unc[e];
          if (pos && a.train_pv) {                               // PV.forward: drop_layer(review_emb) (PV.py:54)
            val *= drop_mult(a.d_pv, (uint32_t)revrow, col);
            a.vec[(size_t)revrow * d + col] = ok ? val : 0.f;
          }
        } else if (pos && a.train_pv) {
          val = unc[e];                                          // the sequence gets the UNcorrupted mean (PVC.py:76,95)
          a.vec[(size_t)revrow * d + col] = ok ? cor[e] : 0.f;    // the PV loss the corrupted one (PVC.py:78)
        } else {
          val = cor[e];
        }
        val *= drop_mult(pos ? a.d_pos : a.d_neg, (uint32_t)revrow, col);     // dropout_layer (ps_model.py:303-304)
        o[e] = val;
      }
      if (a.raw) {                                   // fs: the projection and the rest of x follow (rtm_fs_finish_kernel)
        const size_t rg = pos ? (size_t)revrow : (size_t)a.B * a.R + revrow;
        *reinterpret_cast<float4*>(a.raw + rg * d + 4 * cc) = ok ? make_float4(o[0], o[1], o[2], o[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        continue;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int col = 4 * cc + e;
      float val = o[e];
      if (a.use_seg) val += a.seg_emb[(size_t)seg * d + col];
      if (uid >= 0) val += a.user_emb[(size_t)uid * d + col];
      if (iid >= 0) val += a.item_emb[(size_t)iid * d + col];
      val = ok ? val : 0.f;
      if (a.use_pos) val += a.pe[(size_t)s * d + col];
      o[e] = val;
    }
    *reinterpret_cast<float4*>(a.x + ((size_t)n * a.S + s) * d + 4 * cc) = make_float4(o[0], o[1], o[2], o[3]);
  }
  // padded positive reviews still own a (zero) PV vector
  if (!ok && pos && a.train_pv && s > 0 && half == 0)
    for (int cc = c; cc < nch; cc += 32)
      *reinterpret_cast<float4*>(a.vec + (size_t)revrow * d + 4 * cc) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// Which groups of four consecutive review rows hold at least one real review?  73 % of the review slots of a C4 batch are
// padding, and a launch over ALL groups spends most of its workgroup slots on waves that only find that out after a
// dependent load.  One thread per group: a group with a real review is appended to the list (wave-aggregated: one atomic
// per wave; the order of the list does not matter — a group's outputs are its own), a group of padding only has its key-mask
// flags and word counts written here and never reaches the gather launch (only used when padded rows of x are not read:
// `pads_unread`).  *gcount is cleared by the query-encoder launch (EmbedArgs::clear_word).
__global__ __launch_bounds__(256) void rtm_grouplist_kernel(const RtmK a, int npos_grp, int nneg_grp, FDiv fR, FDiv fK, int* glist,
                                                            int* gcount) {
  const int g = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  const bool in = g < npos_grp + nneg_grp;
  const bool pos = g < npos_grp;
  const int gg = pos ? g : g - npos_grp;
  const int nrev = pos ? a.B * a.R : a.B * a.K * a.R;
  const int64_t rpad = a.RC - 1;
  bool any_ok = false;
  if (in) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rr = 4 * gg + q;
      if (rr < nrev) any_ok |= (pos ? a.pos_r : a.neg_r)[rr] != rpad;
    }
    if (!any_ok) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rr = 4 * gg + q;
        if (rr >= nrev) continue;
        const int base = fdiv(rr, fR), r = rr - base * a.R;
        int n;
        if (pos) n = base * a.J;
        else { const int b = fdiv(base, fK); n = b * a.J + 1 + (base - b * a.K); }
        a.valid[(size_t)n * a.S + r + 1] = 0.f;
        a.cnt[(size_t)n * a.R + r] = 1.f;
      }
    }
  }
  const unsigned long long m = __ballot(in && any_ok);
  int base = 0;
  if (lane == 0 && m) base = atomicAdd(gcount, __popcll(m));
  base = __shfl(base, 0, 64);
  if (in && any_ok) glist[base + __popcll(m & ((1ull << lane) - 1ull))] = g;
}

// ------------------------------------------------------------------ embed forward, pvc encoder: four reviews per wave
// The per-slot kernel above spends most of its time in Philox: a wave (one review) evaluates the token masks of its word
// slots and the dropout words of its output columns, and every evaluation yields four words of which it uses ONE — the
// other three belong to the three neighbouring review rows (counter (col, row >> 2), word row & 3).  Here a wave owns
// the four review rows  4g .. 4g+3  of one side (positive: row b*R + r; negative: row (b*K + k)*R + r), so each
// evaluation serves four reviews (6x fewer Philox instructions), and the gather runs in four 16-lane groups, one review
// each (rows of d <= 256 floats in 16-byte chunks, c and c + 16, ...), four word rows in flight per group:
//   1. lane l reads word slots l and l + 64 of the four reviews (coalesced) and evaluates their token masks;
//   2. the surviving (word id, multiplier) pairs are compacted into one LDS list per review (ballot + prefix popcount);
//   3. group q walks list q, accumulating the corrupted (and, positive under train_pv, the uncorrupted) sum;
//   4. mean, dropout (the wave's 128 dropout evaluations shared through LDS), segment / user / item rows, key mask,
//      positional row -> x.   Query positions (s = 0) and the per-sequence counts ride as extra waves of the launch.
#define E4_LIST 128
template <int NCHL>
struct E4Lds {
  int wid[4][4][E4_LIST];          // [wave][review][entry]
  float tm[4][4][E4_LIST];
  uint32_t dw[4][64 * NCHL][4];    // [wave][column][review]: dropout words of the output row (d = 64 * NCHL columns)
};
template <int NCHL>     // 16-byte chunks per lane of a 16-lane group: d = 64 * NCHL
__global__ __launch_bounds__(256) void rtm_embed4_kernel(const RtmK a, int npos_grp, int nneg_grp, FDiv fR, FDiv fK, int pads_unread,
                                                         const int* __restrict__ glist, const int* __restrict__ gcount, int nq_wg, int diag_arg) {
  const int diag = PS_DIAG_ON ? diag_arg : 0;      // timing-only variants (WRONG results) exist in the diagnostic build only
  __shared__ E4Lds<NCHL> L;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#if PS_DIAG_ON      // in-kernel phase stamps: diagnostic build only
#define E4_STAMP(slot)                                                                             \
  do {                                                                                             \
    if (a.stamp && (int)blockIdx.x == nq_wg + 8 && lane == 0) {                                    \
      unsigned long long t_;                                                                       \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
      a.stamp[16 * wv + (slot)] = t_;                                                              \
    }                                                                                              \
  } while (0)
#else
#define E4_STAMP(slot) do { } while (0)
#endif
  E4_STAMP(0);
  // PS_RTM_STAMP=1 PS_RTM_DIAG=64 (tools/rtm_wg_times.py): start / end time (s_memrealtime: the 100 MHz counter all CUs share —
  // s_memtime is per CU) and XCC of EVERY workgroup's wave 0, behind the phase slots
  const bool wg_times = a.stamp && diag == 64 && threadIdx.x == 0;
  if (wg_times) {
    unsigned long long t_;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
    a.stamp[64 + 3 * blockIdx.x] = t_;
    a.stamp[64 + 3 * blockIdx.x + 2] = (unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID bits 3:0
  }
  struct WgEnd {
    unsigned long long* p; bool on;
    __device__ ~WgEnd() {
      if (on) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); *p = t_; }
    }
  } wg_end = {wg_times ? a.stamp + 64 + 3 * blockIdx.x + 1 : nullptr, wg_times};
  int g = blockIdx.x * 4 + wv;
  if (glist) {
    // with the valid-group list (rtm_grouplist_kernel) the launch is DENSE in real work: the first nq_wg workgroups are the
    // query waves, the rest take groups from the list; waves past its end leave at once (no workgroup barrier in this kernel)
    if ((int)blockIdx.x < nq_wg) g = npos_grp + nneg_grp + (int)blockIdx.x * 4 + wv;
    else {
      const int gi = ((int)blockIdx.x - nq_wg) * 4 + wv;
      const int gsp = glist[gi];                     // requested WITH the count (the list has room for every group: in bounds)
      if (gi >= *gcount) return;
      g = gsp;
    }
  }
  g = __builtin_amdgcn_readfirstlane(g);            // wave-uniform: everything derived from it lives in scalar registers
  const int d = a.d;
  const int64_t rpad = a.RC - 1;
  if (g >= npos_grp + nneg_grp) {
    // ---- query position of sequence n (one wave): x[n][0], valid[n][0], the sequence's valid-position count
    const int n = g - npos_grp - nneg_grp;
    if (n >= a.B * a.J) return;
    const int b = fdiv(n, a.fJ), j = n - b * a.J;
    const bool pos = j == 0;
    const size_t base = pos ? (size_t)b : (size_t)b * a.K + (j - 1);
    const int64_t* rid = (pos ? a.pos_r : a.neg_r) + base * a.R;
    bool okl = lane == 0;
    if (lane > 0 && lane < a.S) okl = rid[lane - 1] != rpad;
    const int cntv = __popcll(__ballot(okl));
    if (lane == 0) { a.seqcnt[n] = cntv; a.valid[(size_t)n * a.S] = 1.f; }
    const size_t spos = base * a.S;
    const int seg = (int)(pos ? a.pos_seg : a.neg_seg)[spos];
    int64_t uid = -1, iid = -1;
    if (a.user_emb) { uid = (pos ? a.pos_u : a.neg_u)[spos]; if (uid < 0 || uid > a.U) uid = -1; }
    if (a.item_emb) { iid = (pos ? a.pos_i : a.neg_i)[spos]; if (iid < 0 || iid > a.PI) iid = -1; }
    for (int col = lane; col < d; col += 64) {
      float val = a.query_emb[(size_t)b * d + col];
      if (a.use_seg) val += a.seg_emb[(size_t)seg * d + col];
      if (uid >= 0) val += a.user_emb[(size_t)uid * d + col];
      if (iid >= 0) val += a.item_emb[(size_t)iid * d + col];
      if (a.use_pos) val += a.pe[col];
      a.x[(size_t)n * a.S * d + col] = val;
    }
    return;
  }
  const bool pos = g < npos_grp;
  const int gg = pos ? g : g - npos_grp;            // = review row >> 2 on its side: the Philox row counter
  const int nrev = pos ? a.B * a.R : a.B * a.K * a.R;
  const bool need_unc = pos && a.train_pv;
  const int64_t* wsrc = pos ? (a.train_pv ? a.pos_pvc : a.pos_words) : (a.train_pv ? a.neg_pvc : a.neg_words_rev);
  const DropSpec& ts = pos ? a.t_pos : a.t_neg;
  const DropSpec& ds = pos ? a.d_pos : a.d_neg;

  // ---- 1. the four reviews: where they sit
  int nq[4], sq[4], segq[4], nwq[4], nlq[4], rrq[4];
  bool okq[4], liveq[4];
  int64_t uidq[4], iidq[4];
  // lane l reads the id and the segment of review l & 3 (two wave-instructions, one round trip; the values then go to
  // scalar registers by v_readlane).  Review after review through wave-uniform loads — each followed by its own wait — the
  // decode was eight dependent round trips: 6.1 k of a workgroup's 10.7 k cycles (profiles/r03_rtm_embed4_stamps.txt)
  int my_rid_lo, my_rid_hi, my_seg;
  {
    const int rr = 4 * gg + (lane & 3);
    const int rrc = rr < nrev ? rr : 0;
    const int base = fdiv(rrc, fR), r = rrc - base * a.R;
    const int64_t rid = (pos ? a.pos_r : a.neg_r)[rrc];
    my_seg = (int)(pos ? a.pos_seg : a.neg_seg)[(size_t)base * a.S + r + 1];
    my_rid_lo = (int)(unsigned long long)rid; my_rid_hi = (int)((unsigned long long)rid >> 32);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int rr = 4 * gg + q;
    liveq[q] = rr < nrev;
    const int rrc = liveq[q] ? rr : 0;
    rrq[q] = rrc;
    const int base = fdiv(rrc, fR), r = rrc - base * a.R;
    const int bn = fdiv(base, fK);
    const int n = pos ? base * a.J : bn * a.J + 1 + (base - bn * a.K);
    nq[q] = n; sq[q] = r + 1;
    const int64_t ridx = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(my_rid_hi, q) << 32) |
                                   (unsigned)__builtin_amdgcn_readlane(my_rid_lo, q));
    okq[q] = liveq[q] && ridx != rpad;
    const size_t spos = (size_t)base * a.S + r + 1;
    segq[q] = __builtin_amdgcn_readlane(my_seg, q);
    uidq[q] = -1; iidq[q] = -1;
    if (a.user_emb) { uidq[q] = (pos ? a.pos_u : a.neg_u)[spos]; if (uidq[q] < 0 || uidq[q] > a.U) uidq[q] = -1; }
    if (a.item_emb) { iidq[q] = (pos ? a.pos_i : a.neg_i)[spos]; if (iidq[q] < 0 || iidq[q] > a.PI) iidq[q] = -1; }
    nwq[q] = 0; nlq[q] = 0;
  }
  // The word slots of the four reviews depend on the group number only, not on what the review ids turn out to be: they are
  // requested HERE, beside the ids and segments, by unconditional loads (slot clamped to the review's last) — one round trip
  // for both.  Fetched under `ok && lane < WL` selects, each of the eight loads sat behind its own branch, the compiler
  // waited for everything in flight at every join, and the whole block came a round trip after the ids.
  const uint8_t* const wm = pos ? a.wmask_pos : a.wmask_neg;
  const int la = min(lane, a.WL - 1), lb = min(lane + 64, a.WL - 1);
  int64_t waq[4], wbq[4];
  uint8_t mka[4] = {1, 1, 1, 1}, mkb[4] = {1, 1, 1, 1};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t* words = wsrc + (size_t)rrq[q] * a.WL;
    waq[q] = words[la];
    wbq[q] = words[lb];
  }
  if (wm) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { mka[q] = wm[(size_t)rrq[q] * a.WL + la]; mkb[q] = wm[(size_t)rrq[q] * a.WL + lb]; }
  }
  // 73 % of the review slots of a C4 batch are padding: a group without a real review skips the Philox evaluations, the
  // lists and the gather (wave-uniform branch) and only writes its masked rows
  const bool any_ok = okq[0] || okq[1] || okq[2] || okq[3];
  E4_STAMP(1);
  int cwa[4] = {-1, -1, -1, -1}, cwb[4] = {-1, -1, -1, -1};     // the counted words of slots lane / lane + 64 (-1: none)
  if (any_ok) {
    // ---- the word slots and their token masks
    Philox4 t0 = {0u, 0u, 0u, 0u}, t1 = {0u, 0u, 0u, 0u};
    if (ts.thr && !(diag & 8)) {
      t0 = philox4x32_10((uint32_t)lane, (uint32_t)gg, ts.site, drop_step(ts), ts.k0, ts.k1);
      if (a.WL > 64) t1 = philox4x32_10((uint32_t)lane + 64u, (uint32_t)gg, ts.site, drop_step(ts), ts.k0, ts.k1);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // word slots lane and lane + 64 of review q
      const int64_t wa64 = (okq[q] && lane < a.WL) ? waq[q] : a.V - 1;
      const int64_t wb64 = (okq[q] && lane + 64 < a.WL) ? wbq[q] : a.V - 1;
      const bool va = okq[q] && lane < a.WL && (wm ? mka[q] != 0 : wa64 != a.V - 1) && wa64 >= 0 && wa64 < a.V;      // = word_ok
      const bool vb = okq[q] && lane + 64 < a.WL && (wm ? mkb[q] != 0 : wb64 != a.V - 1) && wb64 >= 0 && wb64 < a.V;
      const uint32_t w0 = q == 0 ? t0.x : (q == 1 ? t0.y : (q == 2 ? t0.z : t0.w));
      const uint32_t w1 = q == 0 ? t1.x : (q == 1 ? t1.y : (q == 2 ? t1.z : t1.w));
      const float tm0 = ts.thr ? drop_word(ts, w0) : 1.f, tm1 = ts.thr ? drop_word(ts, w1) : 1.f;
      // ---- 2. compaction: entries of slots < 64 first, then slots >= 64 (ascending slot order, like the per-slot kernel)
      const bool ka = va && (need_unc || tm0 != 0.f), kb = vb && (need_unc || tm1 != 0.f);
      const unsigned long long ma = __ballot(ka), mb = __ballot(kb);
      const unsigned long long lt = (1ull << lane) - 1ull;
      nwq[q] = __popcll(__ballot(va)) + __popcll(__ballot(vb));
      nlq[q] = __popcll(ma) + __popcll(mb);
      cwa[q] = va ? (int)wa64 : -1; cwb[q] = vb ? (int)wb64 : -1;
      if (ka) { const int p = __popcll(ma & lt); L.wid[wv][q][p] = (int)wa64; L.tm[wv][q][p] = tm0; }
      if (kb) { const int p = __popcll(ma) + __popcll(mb & lt); L.wid[wv][q][p] = (int)wb64; L.tm[wv][q][p] = tm1; }
    }
  }
  // dropout words of the output row: lane evaluates columns lane, lane + 64, ... for the four reviews at once
  if (ds.thr && any_ok && !(diag & 2))
    for (int col = lane; col < d; col += 64) {
      const Philox4 r = philox4x32_10((uint32_t)col, (uint32_t)gg, ds.site, drop_step(ds), ds.k0, ds.k1);
      L.dw[wv][col][0] = r.x; L.dw[wv][col][1] = r.y; L.dw[wv][col][2] = r.z; L.dw[wv][col][3] = r.w;
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the lists are read back by other lanes of this wave only
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  E4_STAMP(2);
  // ---- 3. gather: group q = lane >> 4 walks list q
  const int q = lane >> 4, c = lane & 15;
  const int myn = q == 0 ? nlq[0] : (q == 1 ? nlq[1] : (q == 2 ? nlq[2] : nlq[3]));
  const int maxn = (diag & 1) ? 0 : max(max(nlq[0], nlq[1]), max(nlq[2], nlq[3]));       // diag: timing experiments only
  float4 v[NCHL], vc[NCHL];
#pragma unroll
  for (int k = 0; k < NCHL; ++k) { v[k] = make_float4(0.f, 0.f, 0.f, 0.f); vc[k] = v[k]; }
  const int* wl = L.wid[wv][q];
  const float* tl = L.tm[wv][q];
  constexpr int E4_U = 2;          // word rows REALLY in flight per 16-lane group now that their loads are unconditional (d = 128: 95 registers, five waves per SIMD; 3 measured the same, 4 costs a wave)
  for (int i0 = 0; i0 < maxn; i0 += E4_U) {
    float4 rowv[E4_U][NCHL]; float mt[E4_U], mo[E4_U];
#pragma unroll
    for (int u = 0; u < E4_U; ++u) {
      const bool on = i0 + u < myn;
      const int wi = on ? wl[i0 + u] : 0;
      mt[u] = on ? tl[i0 + u] : 0.f;
      mo[u] = on ? 1.f : 0.f;
      const float* row = a.word_emb + (size_t)wi * d + 4 * c;
      // (unconditional: an entry past this list reads row 0 and is weighted 0 — under `on ? load : 0` every one of the
      // E4_U x NCHL loads was its own branch and the compiler waited for all loads in flight at each join)
#pragma unroll
      for (int k = 0; k < NCHL; ++k) rowv[u][k] = *reinterpret_cast<const float4*>(row + 64 * k);
    }
#pragma unroll
    for (int u = 0; u < E4_U; ++u)
#pragma unroll
      for (int k = 0; k < NCHL; ++k) {
        v[k].x += rowv[u][k].x * mo[u]; v[k].y += rowv[u][k].y * mo[u]; v[k].z += rowv[u][k].z * mo[u]; v[k].w += rowv[u][k].w * mo[u];
        vc[k].x += rowv[u][k].x * mt[u]; vc[k].y += rowv[u][k].y * mt[u];
        vc[k].z += rowv[u][k].z * mt[u]; vc[k].w += rowv[u][k].w * mt[u];
      }
  }

  E4_STAMP(3);
  // ---- first pass of the backward's inverted index (RtmK::count_fwd): every counted word takes its RANK among the
  // occurrences of that word (a returning atomic on the word's counter, in flight under the rest of the kernel (issued in front of the gather instead they cost 4 us: measured)); the fill
  // then places the occurrence at  segment start + rank  without another atomic
  int rka[4] = {-1, -1, -1, -1}, rkb[4] = {-1, -1, -1, -1};
  if (a.count_fwd && any_ok) {
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      rka[qq] = cwa[qq] >= 0 ? atomicAdd(&a.wcnt[cwa[qq]], 1) : -1;
      rkb[qq] = cwb[qq] >= 0 ? atomicAdd(&a.wcnt[cwb[qq]], 1) : -1;
    }
  }
  // ---- 4. this group's review: mean, dropout, segment / user / item rows, mask, positional row
  const bool live = q == 0 ? liveq[0] : (q == 1 ? liveq[1] : (q == 2 ? liveq[2] : liveq[3]));
  if (live) {
  const bool ok = q == 0 ? okq[0] : (q == 1 ? okq[1] : (q == 2 ? okq[2] : okq[3]));
  const int n = q == 0 ? nq[0] : (q == 1 ? nq[1] : (q == 2 ? nq[2] : nq[3]));
  const int s = q == 0 ? sq[0] : (q == 1 ? sq[1] : (q == 2 ? sq[2] : sq[3]));
  const int seg = q == 0 ? segq[0] : (q == 1 ? segq[1] : (q == 2 ? segq[2] : segq[3]));
  const int nw = q == 0 ? nwq[0] : (q == 1 ? nwq[1] : (q == 2 ? nwq[2] : nwq[3]));
  const int64_t uid = q == 0 ? uidq[0] : (q == 1 ? uidq[1] : (q == 2 ? uidq[2] : uidq[3]));
  const int64_t iid = q == 0 ? iidq[0] : (q == 1 ? iidq[1] : (q == 2 ? iidq[2] : iidq[3]));
  const int rr = 4 * gg + q;
  const float cntf = (float)(nw > 0 ? nw : 1), inv = 1.f / cntf;
  if (c == 0) { a.valid[(size_t)n * a.S + s] = ok ? 1.f : 0.f; a.cnt[(size_t)n * a.R + s - 1] = cntf; }
  const bool skip_row = (!ok && pads_unread && !need_unc) || (diag & 4);
  // segment / positional (/ user / item) rows of this position, all chunks at once (each behind its own per-element test
  // before: a round trip per element)
  float4 addv[NCHL], pev[NCHL];       // addv: segment + user + item rows (summed in that order, as the per-element form did)
#pragma unroll
  for (int k = 0; k < NCHL; ++k) { addv[k] = make_float4(0.f, 0.f, 0.f, 0.f); pev[k] = addv[k]; }
  if (!skip_row && !a.raw) {
    if (a.use_seg) {
#pragma unroll
      for (int k = 0; k < NCHL; ++k) addv[k] = *reinterpret_cast<const float4*>(a.seg_emb + (size_t)seg * d + 4 * c + 64 * k);
    }
    if (a.use_pos) {
#pragma unroll
      for (int k = 0; k < NCHL; ++k) pev[k] = *reinterpret_cast<const float4*>(a.pe + (size_t)s * d + 4 * c + 64 * k);
    }
  }
#pragma unroll
  for (int k = 0; k < NCHL; ++k) {
    if (skip_row) break;
    const int col0 = 4 * c + 64 * k;
    const float sg4[4] = {addv[k].x, addv[k].y, addv[k].z, addv[k].w}, pe4[4] = {pev[k].x, pev[k].y, pev[k].z, pev[k].w};
    const float unc[4] = {v[k].x * inv, v[k].y * inv, v[k].z * inv, v[k].w * inv};
    const float cor[4] = {vc[k].x * inv, vc[k].y * inv, vc[k].z * inv, vc[k].w * inv};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int col = col0 + e;
      float val = need_unc ? unc[e] : cor[e];        // the sequence gets the UNcorrupted mean under train_pv (PVC.py:76,95)
      if (ds.thr && ok) val *= drop_word(ds, L.dw[wv][col][q]);                   // dropout_layer (ps_model.py:303-304)
      if (a.raw) { o[e] = ok ? val : 0.f; continue; }                             // fs: rtm_fs_finish_kernel does the rest
      if (a.use_seg) val += sg4[e];
      if (uid >= 0) val += a.user_emb[(size_t)uid * d + col];       // (user / item rows: not in configs[3]; per element as before)
      if (iid >= 0) val += a.item_emb[(size_t)iid * d + col];
      val = ok ? val : 0.f;
      if (a.use_pos) val += pe4[e];
      o[e] = val;
    }
    if (a.raw) {
      const size_t rg = pos ? (size_t)rr : (size_t)a.B * a.R + rr;
      *reinterpret_cast<float4*>(a.raw + rg * d + col0) = make_float4(o[0], o[1], o[2], o[3]);
      continue;
    }
    *reinterpret_cast<float4*>(a.x + ((size_t)n * a.S + s) * d + col0) = make_float4(o[0], o[1], o[2], o[3]);
    if (need_unc)                                   // the PV loss predicts from the corrupted mean (PVC.py:78)
      *reinterpret_cast<float4*>(a.vec + (size_t)rr * d + col0) =
          ok ? make_float4(cor[0], cor[1], cor[2], cor[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  }
  if (a.count_fwd && any_ok) {
#pragma unroll
    for (int qq = 0; qq < 4; ++qq)
      if (okq[qq]) {
        int* rk = a.wrank + ((size_t)(pos ? 0 : a.B * a.R) + rrq[qq]) * a.WL;
        if (lane < a.WL) rk[lane] = rka[qq];
        if (lane + 64 < a.WL) rk[lane + 64] = rkb[qq];
      }
  }
  E4_STAMP(4);
#undef E4_STAMP
}

// ------------------------------------------------------------------ scores
// also writes the loss weight of the sequence (ps_model.py:344-345): pos_weight for the positive, and for a
// negative 1 iff it has at least one real review
// `fold_loss` (training without the PV loss): the workgroup adds its four loss terms to ONE 64-bit word with a returning
// agent-scope atomic — fixed point (2^-20 units: integer adds, so the total does not depend on the arrival order) in the
// low 48 bits, the arrival count above them — and the workgroup that sees the count complete holds the total in the
// value it got back: rtm_loss_kernel's result without its launch and without a device-scope fence (which on this part
// writes back the XCD's L2: the first form, partials + __threadfence + ticket, doubled this kernel's 6 us).
__global__ __launch_bounds__(256) void rtm_score_kernel(const RtmK a, float* out, int fold_loss, unsigned long long* ticket) {
  __shared__ float wterm[4];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int n = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  float term = 0.f;
  if (n < a.B * a.J) {
    float s = 0.f;
    for (int e = lane; e < a.d; e += 64) s += a.enc[(size_t)n * a.d + e] * a.wo_w[e];
    s = wave_sum(s) + a.wo_b[0];
    if (lane == 0) out[n] = s;
    if (!a.eval) {
      const int b = fdiv(n, a.fJ), j = n - b * a.J;
      float wgt;
      if (j == 0) wgt = a.pos_weight ? (float)a.K : 1.f;
      else {
        bool any = false;
        for (int r = lane; r < a.R; r += 64) any = any || (a.neg_r[((size_t)b * a.K + j - 1) * a.R + r] != a.RC - 1);
        wgt = __ballot(any) != 0ull ? 1.f : 0.f;
      }
      if (lane == 0) a.weight[n] = wgt;
      term = wgt * softplus_f(j == 0 ? -s : s);
    }
  }
  if (!fold_loss) return;
  if (lane == 0) wterm[wv] = term;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float mine_f = (wterm[0] + wterm[1]) + (wterm[2] + wterm[3]);           // >= 0: weights and softplus are
    // (a partial that is NaN / Inf / beyond the 48-bit field's share — 2^27 / grid per workgroup — is handed over as 0 and
    // poisons the loss through the spare word of the cleared group, set by a returning atomic the arrival add depends on:
    // garbage must not carry into the arrival count)
    uint32_t* spare = reinterpret_cast<uint32_t*>(ticket) + 3;
    long long fx = 0;
    if (mine_f >= 0.f && mine_f * (float)gridDim.x < 134217728.f) fx = (long long)llrintf(mine_f * 1048576.f);
    else fx = (long long)(__hip_atomic_fetch_or(spare, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0u);
    const unsigned long long mine = (unsigned long long)fx + (1ull << 48);
    const unsigned long long old = __hip_atomic_fetch_add(ticket, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((old >> 48) == (unsigned long long)gridDim.x - 1ull) {
      const unsigned long long tot = (old + mine) & ((1ull << 48) - 1ull);
      const bool poison = __hip_atomic_load(spare, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
      const float psl = poison ? __builtin_inff() : (float)((double)tot * (1.0 / 1048576.0)) / (float)a.B;
      a.loss3[0] = psl; a.loss3[1] = psl; a.loss3[2] = 0.f;
      a.nvalid[0] = 0.f;
    }
  }
}

// PV logits: task t = ((b*R + r)*W + w)*(1+K) + j
#define PV_U 4
__global__ __launch_bounds__(256) void rtm_pv_fwd_kernel(const RtmK a, int ntask, int lpr) {
  const int tid = threadIdx.x, c = tid % lpr;
  const int grp = blockIdx.x * (256 / lpr) + tid / lpr;
  const int K1 = a.K + 1, nch = a.d >> 2;
  const float* rows[PV_U]; const float* vecs[PV_U]; int tt[PV_U]; float sgn[PV_U];
#pragma unroll
  for (int u = 0; u < PV_U; ++u) {
    const int t = grp * PV_U + u;
    tt[u] = t; rows[u] = nullptr; vecs[u] = nullptr; sgn[u] = 1.f;
    if (t < ntask) {
      const int tw = fdiv(t, a.fK1), j = t - tw * K1;
      const int rev = tw / a.W, w = tw - rev * a.W;
      int64_t idx = j == 0 ? a.pos_words[(size_t)rev * a.W + w] : a.neg_word_idxs[(size_t)rev * a.W * a.K + (size_t)w * a.K + j - 1];
      idx = rclamp(idx, a.V - 1);
      rows[u] = a.word_emb + (size_t)idx * a.d;
      vecs[u] = a.vec + (size_t)rev * a.d;
      sgn[u] = j == 0 ? -1.f : 1.f;
    }
  }
  float4 r[PV_U], v[PV_U];
#pragma unroll
  for (int u = 0; u < PV_U; ++u) {
    r[u] = make_float4(0.f, 0.f, 0.f, 0.f); v[u] = r[u];
    if (rows[u] && c < nch) {
      r[u] = *reinterpret_cast<const float4*>(rows[u] + 4 * c);
      v[u] = *reinterpret_cast<const float4*>(vecs[u] + 4 * c);
    }
  }
#pragma unroll
  for (int u = 0; u < PV_U; ++u) {
    float s = r[u].x * v[u].x + r[u].y * v[u].y + r[u].z * v[u].z + r[u].w * v[u].w;
    if (rows[u])
      for (int cc = c + lpr; cc < nch; cc += lpr) {
        float4 rr = *reinterpret_cast<const float4*>(rows[u] + 4 * cc);
        float4 vv = *reinterpret_cast<const float4*>(vecs[u] + 4 * cc);
        s += rr.x * vv.x + rr.y * vv.y + rr.z * vv.z + rr.w * vv.w;
      }
    s = group_sum(s, lpr);
    if (c == 0 && rows[u]) {
      a.pv_scores[tt[u]] = s;
      a.pv_terms[tt[u]] = softplus_f(sgn[u] < 0.f ? -s : s);
    }
  }
}

// Loss: ps = mean_b sum_j weight * bce ; pv = sum_rev masked-mean_w(sum_j bce) / #non-pad positive reviews
__global__ __launch_bounds__(256) void rtm_loss_kernel(const RtmK a) {
  __shared__ float s1[256], s2[256], s3[256];
  const int tid = threadIdx.x, K1 = a.K + 1;
  float ps = 0.f, pv = 0.f, nv = 0.f;
  for (int b = tid; b < a.B; b += 256) {
    const float* sc = a.scores + (size_t)b * K1;
    const float* wg = a.weight + (size_t)b * K1;
    ps += wg[0] * softplus_f(-sc[0]);
    for (int k = 0; k < a.K; ++k) ps += wg[1 + k] * softplus_f(sc[1 + k]);
  }
  if (a.train_pv) {
    for (int rev = tid; rev < a.B * a.R; rev += 256) {
      nv += (a.pos_r[rev] != a.RC - 1) ? 1.f : 0.f;
      float sum = 0.f; int cnt = 0;
      for (int w = 0; w < a.W; ++w) {
        if (a.pos_masks[(size_t)rev * a.W + w]) {
          ++cnt;
          const float* t = a.pv_terms + ((size_t)rev * a.W + w) * K1;
          for (int j = 0; j < K1; ++j) sum += t[j];
        }
      }
      pv += sum / (float)(cnt > 0 ? cnt : 1);
    }
  }
  s1[tid] = ps; s2[tid] = pv; s3[tid] = nv;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { s1[tid] += s1[tid + o]; s2[tid] += s2[tid + o]; s3[tid] += s3[tid + o]; }
    __syncthreads();
  }
  if (tid == 0) {
    const float psl = s1[0] / (float)a.B;
    const float pvl = a.train_pv ? s2[0] / s3[0] : 0.f;
    a.loss3[0] = psl + pvl; a.loss3[1] = psl; a.loss3[2] = pvl;
    a.nvalid[0] = s3[0];
  }
}

// ------------------------------------------------------------------ backward kernels
__global__ __launch_bounds__(256) void rtm_score_bwd_kernel(const RtmK a, int zero_dqe, uint32_t* sig, uint32_t sigval) {
  fork_signal(sig, sigval);
  extern __shared__ float acc[];                  // [d] partial of d wo_w, + 1 for d wo_b
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), K1 = a.K + 1;
  if (zero_dqe)                                   // d query_emb collects atomics much later (rtm_embed_bwd_kernel): no memset launch
    for (int e = blockIdx.x * 256 + threadIdx.x; e < a.B * a.d; e += gridDim.x * 256) a.dqe[e] = 0.f;
  for (int e = threadIdx.x; e <= a.d; e += 256) acc[e] = 0.f;
  __syncthreads();
  const float sc = a.scale * (a.scale_dev ? *a.scale_dev : 1.f) / (float)a.B;
  const int nw = gridDim.x * 4;
  for (int n = blockIdx.x * 4 + wv; n < a.B * K1; n += nw) {
    const int b = fdiv(n, a.fK1), j = n - b * K1;
    const float wgt = a.weight[n];
    const float s = a.scores[n];
    const float ds = wgt * (sigmoid_f(s) - (j == 0 ? 1.f : 0.f)) * sc;
    for (int e = lane; e < a.d; e += 64) {
      a.denc[(size_t)n * a.d + e] = ds * a.wo_w[e];
      if (!a.det) atomicAdd(&acc[e], ds * a.enc[(size_t)n * a.d + e]);
    }
    if (lane == 0 && !a.det) atomicAdd(&acc[a.d], ds);
  }
  if (a.det) return;                              // rtm_score_wo_det_kernel adds them up in sequence order
  __syncthreads();
  for (int e = threadIdx.x; e < a.d; e += 256) atomicAdd(&a.g_wo_w[e], acc[e]);
  if (threadIdx.x == 0) atomicAdd(&a.g_wo_b[0], acc[a.d]);
}
// ---- deterministic mode (ps_deterministic): the reductions of the backward whose fp32 additions could meet in a
// different order from run to run, each as a fixed-order sum.  (The word index gets its ranks from rtm_hist_kernel<0, 64, 1>.)
// d wo_w / d wo_b: one workgroup, a thread owns its columns and walks the sequences in order
__global__ __launch_bounds__(256) void rtm_score_wo_det_kernel(const RtmK a) {
  const int K1 = a.K + 1, tid = threadIdx.x;
  const float sc = a.scale * (a.scale_dev ? *a.scale_dev : 1.f) / (float)a.B;
  float acc0 = 0.f, acc1 = 0.f, accb = 0.f;
  const int e0 = tid < a.d ? tid : a.d - 1, e1 = tid + 256 < a.d ? tid + 256 : a.d - 1;
  for (int n0 = 0; n0 < a.B * K1; n0 += 8) {
    float ds[8], x0[8], x1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = n0 + u < a.B * K1 ? n0 + u : n0;
      const int b = fdiv(n, a.fK1), j = n - b * K1;
      ds[u] = a.weight[n] * (sigmoid_f(a.scores[n]) - (j == 0 ? 1.f : 0.f)) * sc;
      x0[u] = a.enc[(size_t)n * a.d + e0];
      x1[u] = a.enc[(size_t)n * a.d + e1];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (n0 + u < a.B * K1) { acc0 += ds[u] * x0[u]; acc1 += ds[u] * x1[u]; accb += ds[u]; }
  }
  if (tid < a.d) a.g_wo_w[tid] += acc0;
  if (tid + 256 < a.d) a.g_wo_w[tid + 256] += acc1;
  if (tid == 0) a.g_wo_b[0] += accb;
}
// d query_emb[b] = the query-position gradients of b's J sequences, in sequence order
__global__ __launch_bounds__(256) void rtm_dqe_det_kernel(const RtmK a) {
  const int b = blockIdx.x;
  for (int e = threadIdx.x; e < a.d; e += 256) {
    float acc = 0.f;
    for (int j = 0; j < a.J; ++j) acc += a.dx[(size_t)(b * a.J + j) * a.S * a.d + e];
    a.dqe[(size_t)b * a.d + e] = acc;
  }
}
// column sums of x [rows, d] in two fixed-order levels (the fs projection's bias gradient)
__global__ __launch_bounds__(256) void rtm_colsum_det_kernel(const float* x, int rows, int d, float* part) {
  const int per = (rows + (int)gridDim.x - 1) / (int)gridDim.x;
  const int r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
  for (int e = threadIdx.x; e < d; e += 256) {
    float acc = 0.f;
    for (int r = r0; r < r1; ++r) acc += x[(size_t)r * d + e];
    part[(size_t)blockIdx.x * d + e] = acc;
  }
}
__global__ __launch_bounds__(256) void rtm_colsum_det_fold_kernel(const float* part, int nblk, int d, float* dst) {
  for (int e = threadIdx.x; e < d; e += 256) {
    float acc = 0.f;
    for (int b = 0; b < nblk; ++b) acc += part[(size_t)b * d + e];
    dst[e] += acc;
  }
}
// The word-gradient reduce: a word's occurrences in list order (= task order, see the ranks) by ONE wave, plain add into the
// word's row (its sole writer in this launch).  Words with more than WR_DET_LIM occurrences (Zipf heads: tens of thousands) go
// to a list — its order does not matter — and get a 16-wave workgroup each: wave k sums the k-th sixteenth of the segment, the
// sixteen partials are added in wave order.
#define WR_DET_LIM 4096
template <int NK>
__device__ inline void wr_det_sum(const RtmK& a, int off, int n, int lane, float (&acc)[NK]) {
  constexpr int U = NK <= 2 ? 16 : 4;
  const int d = a.d;
  int col[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) { acc[k] = 0.f; col[k] = lane + 64 * k < d ? lane + 64 * k : d - 1; }
  for (int i0 = 0; i0 < n; i0 += U) {
    float r[U][NK];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int sl = a.wl[off + (i0 + u < n ? i0 + u : i0)].x;
      const float* row = a.gs + (size_t)sl * d;
#pragma unroll
      for (int k = 0; k < NK; ++k) r[u][k] = row[col[k]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i0 + u < n) {
#pragma unroll
        for (int k = 0; k < NK; ++k) acc[k] += r[u][k];
      }
  }
}
template <int NK>
__global__ __launch_bounds__(256) void rtm_wreduce_det_kernel(const RtmK a, int* heavy, int* nheavy) {
  const int lane = threadIdx.x & 63;
  const int wave0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwave = (gridDim.x * blockDim.x) >> 6;
  for (int w = wave0; w < (int)a.V; w += nwave) {
    const int n = a.wcnt[w];
    if (n == 0) continue;
    if (n > WR_DET_LIM) { if (lane == 0) heavy[atomicAdd(nheavy, 1)] = w; continue; }
    float acc[NK];
    wr_det_sum<NK>(a, a.woff[w], n, lane, acc);
    float* grow = a.g_word_emb + (size_t)w * a.d;
#pragma unroll
    for (int k = 0; k < NK; ++k)
      if (lane + 64 * k < a.d) grow[lane + 64 * k] += acc[k];
  }
}
template <int NK>
__global__ __launch_bounds__(1024) void rtm_wreduce_heavy_det_kernel(const RtmK a, const int* heavy, const int* nheavy) {
  __shared__ float part[16][64 * NK];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int h = blockIdx.x; h < *nheavy; h += gridDim.x) {
    const int w = heavy[h], n = a.wcnt[w], per = (n + 15) / 16;
    const int i0 = wv * per, cnt = max(0, min(n, i0 + per) - i0);
    float acc[NK];
    wr_det_sum<NK>(a, a.woff[w] + i0, cnt, lane, acc);
#pragma unroll
    for (int k = 0; k < NK; ++k) part[wv][lane + 64 * k] = acc[k];
    __syncthreads();
    for (int e = threadIdx.x; e < a.d; e += 1024) {
      float t = 0.f;
      for (int q = 0; q < 16; ++q) t += part[q][e];
      a.g_word_emb[(size_t)w * a.d + e] += t;
    }
    __syncthreads();
  }
}

// deterministic mode: the PV loss's word-row tasks (review, window slot, 1 + K words) as scatter keys with their scalar — the
// task's row is  ds * vec[review]  (the same arithmetic as rtm_pv_bwd_kernel below)
__global__ __launch_bounds__(256) void rtm_pv_keys_kernel(const RtmK a, int32_t* keys, float* scale, int32_t* rowidx, int ntask) {
  const int t = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (t >= ntask) return;
  const int K1 = a.K + 1, per = a.W * K1;
  const int rev = t / per, rem = t - rev * per, w = rem / K1, j = rem - w * K1;
  int cnt = 0;
  for (int w2 = 0; w2 < a.W; ++w2) cnt += a.pos_masks[(size_t)rev * a.W + w2] ? 1 : 0;
  const float sc = a.scale * (a.scale_dev ? *a.scale_dev : 1.f) / ((float)(cnt > 0 ? cnt : 1) * a.nvalid[0]);
  int32_t key = -1; float ds = 0.f;
  if (a.pos_masks[(size_t)rev * a.W + w]) {
    int64_t idx = j == 0 ? a.pos_words[(size_t)rev * a.W + w] : a.neg_word_idxs[(size_t)rev * a.W * a.K + (size_t)w * a.K + j - 1];
    idx = rclamp(idx, a.V - 1);
    const float s = a.pv_scores[((size_t)rev * a.W + w) * K1 + j];
    ds = (sigmoid_f(s) - (j == 0 ? 1.f : 0.f)) * sc;
    key = idx != a.V - 1 ? (int32_t)idx : -1;
  }
  keys[t] = key; scale[t] = ds; rowidx[t] = rev;
}

// PV backward: one wave per positive review; d vec (dense) and word-row scatter-adds
__global__ __launch_bounds__(256) void rtm_pv_bwd_kernel(const RtmK a) {
  const int lane = threadIdx.x & 63, half = lane >> 5, c = lane & 31;
  const int rev = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (rev >= a.B * a.R) return;
  const int d = a.d, epl = d >> 5, K1 = a.K + 1;
  int cnt = 0;
  for (int w = 0; w < a.W; ++w) cnt += a.pos_masks[(size_t)rev * a.W + w] ? 1 : 0;
  const float sc = a.scale * (a.scale_dev ? *a.scale_dev : 1.f) / ((float)(cnt > 0 ? cnt : 1) * a.nvalid[0]);
  float dv[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) dv[k] = 0.f;
  const float* vec = a.vec + (size_t)rev * d;
  for (int t = half; t < a.W * K1; t += 2) {
    const int w = t / K1, j = t - w * K1;
    if (!a.pos_masks[(size_t)rev * a.W + w]) continue;
    int64_t idx = j == 0 ? a.pos_words[(size_t)rev * a.W + w] : a.neg_word_idxs[(size_t)rev * a.W * a.K + (size_t)w * a.K + j - 1];
    idx = rclamp(idx, a.V - 1);
    const float s = a.pv_scores[((size_t)rev * a.W + w) * K1 + j];
    const float ds = (sigmoid_f(s) - (j == 0 ? 1.f : 0.f)) * sc;
    const float* wrow = a.word_emb + (size_t)idx * d;
    float* grow = a.g_word_emb + (size_t)idx * d;
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (k < epl) {
        const int e = c + 32 * k;
        if (idx != a.V - 1 && !a.det) atomicAdd(&grow[e], ds * vec[e]);      // det: rtm_pv_keys_kernel + launch_rows_scatter_det
        dv[k] += ds * wrow[e];
      }
  }
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (k < epl) {
      float v = dv[k] + __shfl_xor(dv[k], 32, 64);
      if (half == 0) a.dvec[(size_t)rev * d + c + 32 * k] = v;
    }
}

// deterministic mode: the user / item row of every sequence position as a scatter key (-1: no gradient — a padded review
// position, the embedding's padding row, an id out of range), position-major like d x
__global__ __launch_bounds__(256) void rtm_ui_keys_kernel(const RtmK a, int32_t* ukeys, int32_t* ikeys, int32_t* rkeys, int npos) {
  const int t = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (t >= npos) return;
  const int n = t / a.S, s = t - n * a.S;
  const int b = fdiv(n, a.fJ), j = n - b * a.J;
  const bool pos = j == 0;
  const size_t base = pos ? (size_t)b : (size_t)b * a.K + (j - 1);
  bool ok = true;
  int64_t rid = -1;
  if (s > 0) { rid = (pos ? a.pos_r : a.neg_r)[base * a.R + s - 1]; ok = rid != a.RC - 1; }
  if (rkeys) rkeys[t] = (s > 0 && rid >= 0 && rid < a.RC - 1) ? (int32_t)rid : -1;      // pv encoder: the review's table row
  const size_t spos = base * a.S + s;
  if (ukeys) {
    const int64_t uid = ok ? (pos ? a.pos_u : a.neg_u)[spos] : -1;
    ukeys[t] = (uid >= 0 && uid < a.U) ? (int32_t)uid : -1;
  }
  if (ikeys) {
    const int64_t iid = ok ? (pos ? a.pos_i : a.neg_i)[spos] : -1;
    ikeys[t] = (iid >= 0 && iid < a.PI) ? (int32_t)iid : -1;
  }
}

// Backward of rtm_embed.  Like rtm_embed4_kernel it works on groups of the four review rows 4g .. 4g+3 of one side, which
// share one Philox counter row: a lane evaluates the dropout words of its columns once for the four reviews, and the
// (unconditional, all-real-address) loads of the four rows are in flight together — the wave's dependent chain is one round
// trip per group, EB_GROUPS groups per wave.  (Round 1: one wave per slot in a 38-slot grid-stride walk, one Philox
// evaluation per element: 69 us.)  Lane l owns columns l, l + 64, ...  The query slots (s = 0) ride as extra waves.
// The segment-embedding gradient (3 rows fed by EVERY slot) is summed in registers, combined over the workgroup's waves
// in LDS and PARKED as one [3][d] partial per workgroup (`seg_part`, folded by the step's last launch, ColFoldList)
// instead of 3d same-address atomics per workgroup.
// PL (the pvc encoder without the PV loss, the feature-selection layer and the user / item embeddings — configs[3]): those
// switches are compile-time off, which takes their pointers and flags out of the scalar registers (the general form reloads 34
// spilled scalars per group from VGPR lanes)
template <int NK, int PL>      // columns per lane: d <= 64 * NK
__global__ __launch_bounds__(256) void rtm_embed_bwd_kernel(const RtmK a, float* seg_part, int npos_w, int nneg_w, FDiv fR, FDiv fK) {
  __shared__ float segs[4][3][64 * NK];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wave = (int)blockIdx.x * 4 + wv;
  const int d = a.d;
  const bool pvc = PL ? true : (bool)a.pvc, has_ui = !PL && !a.det && (a.g_user_emb || a.g_item_emb);   // det: rtm_ui_keys_kernel + launch_rows_scatter_det
  const float* const dmean = PL ? nullptr : a.dmean;
  const int64_t rpad = a.RC - 1;
  DropSpec dpos = a.d_pos, dneg = a.d_neg, dpv = a.d_pv;      // the step word is read once, not per element
  dpos.step = drop_step(a.d_pos); dpos.step_ptr = nullptr;
  dneg.step = drop_step(a.d_neg); dneg.step_ptr = nullptr;
  dpv.step = drop_step(a.d_pv); dpv.step_ptr = nullptr;
  float sacc[3][NK];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int k = 0; k < NK; ++k) sacc[q][k] = 0.f;
  int colc[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) colc[k] = lane + 64 * k < d ? lane + 64 * k : d - 1;
  if (wave < npos_w + nneg_w) {
    const bool pos = wave < npos_w;
    const int w0 = pos ? wave : wave - npos_w;
    const int nrev = pos ? a.B * a.R : a.B * a.K * a.R;
    const DropSpec ds = drop_select(pos, dpos, dneg);
    const bool wantdv = !PL && pos && a.train_pv;
    // lane i < 4 * EB_GROUPS decodes review row  4 * EB_GROUPS * w0 + i  (one coalesced read of the review ids / segment ids:
    // group after group through scalar loads the decode alone took 20 us)
    int my_n = 0, my_s = 1, my_seg = 3; int64_t my_rid = rpad; bool my_ok = false;
    {
      const int rr = w0 * 4 * EB_GROUPS + lane;
      if (lane < 4 * EB_GROUPS && rr < nrev) {
        const int base = fdiv(rr, fR), r = rr - base * a.R;
        int b;
        if (pos) { b = base; my_n = b * a.J; }
        else { b = fdiv(base, fK); my_n = b * a.J + 1 + (base - b * a.K); }
        my_s = r + 1;
        my_rid = (pos ? a.pos_r : a.neg_r)[rr];
        my_ok = my_rid != rpad;
        my_seg = (int)(pos ? a.pos_seg : a.neg_seg)[(size_t)base * a.S + r + 1];
      }
    }
    const unsigned long long okm = __ballot(my_ok);
    for (int gi = 0; gi < EB_GROUPS; ++gi) {
      const int gg = w0 * EB_GROUPS + gi;                   // review rows 4gg .. 4gg+3 of this side: one Philox counter row
      if (!((okm >> (4 * gi)) & 0xfull)) continue;
      int nq[4], sq[4], segq[4], rrq[4]; bool okq[4]; int64_t ridq[4]; size_t sposq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int src_lane = 4 * gi + q;
        okq[q] = (okm >> src_lane) & 1ull;
        const int sl = okq[q] ? src_lane : 4 * gi + (__ffsll((long long)((okm >> (4 * gi)) & 0xfull)) - 1);   // a dead row repeats a live one
        rrq[q] = w0 * 4 * EB_GROUPS + sl;
        // (the source lane is the same for the whole wave: v_readlane into scalar registers, not an LDS permute)
        nq[q] = __builtin_amdgcn_readlane(my_n, sl); sq[q] = __builtin_amdgcn_readlane(my_s, sl);
        segq[q] = __builtin_amdgcn_readlane(my_seg, sl);
        ridq[q] = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)((unsigned long long)my_rid >> 32), sl) << 32) |
                            (unsigned)__builtin_amdgcn_readlane((int)(unsigned long long)my_rid, sl));
        const int base = pos ? fdiv(nq[q], a.fJ) : (fdiv(nq[q], a.fJ) * a.K + (nq[q] - fdiv(nq[q], a.fJ) * a.J - 1));
        sposq[q] = (size_t)base * a.S + sq[q];
      }
      uint32_t dw[NK][4];
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        Philox4 t = {0u, 0u, 0u, 0u};
        if (ds.thr) t = philox4x32_10((uint32_t)(lane + 64 * k), (uint32_t)gg, ds.site, ds.step, ds.k0, ds.k1);
        dw[k][0] = t.x; dw[k][1] = t.y; dw[k][2] = t.z; dw[k][3] = t.w;
      }
      // every address is a real one (a dead row repeats the group's first, columns past d clamp), so the loads of the four
      // rows are unconditional and in flight together
      float gk[4][NK], src[4][NK], dvv[4][NK], cntv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* g = a.dx + ((size_t)nq[q] * a.S + sq[q]) * d;
        // fs: the review vector reached x through tanh(f_W . raw + b); its input gradient d raw was left in a.dmean
        const float* gm = dmean ? dmean + ((pos ? (size_t)0 : (size_t)a.B * a.R) + rrq[q]) * d : g;
        const float* dvp = wantdv ? a.dvec + (size_t)rrq[q] * d : g;
#pragma unroll
        for (int k = 0; k < NK; ++k) { gk[q][k] = g[colc[k]]; src[q][k] = gm[colc[k]]; dvv[q][k] = dvp[colc[k]]; }
        cntv[q] = pvc ? a.cnt[(size_t)nq[q] * a.R + sq[q] - 1] : 1.f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (!okq[q]) continue;                             // wave-uniform
        if (has_ui) {                // user / item embedding rows of this position
          const int64_t uid = a.g_user_emb ? (pos ? a.pos_u : a.neg_u)[sposq[q]] : -1;
          const int64_t iid = a.g_item_emb ? (pos ? a.pos_i : a.neg_i)[sposq[q]] : -1;
#pragma unroll
          for (int k = 0; k < NK; ++k) {
            const int col = lane + 64 * k;
            if (col < d) {
              if (uid >= 0 && uid < a.U) atomicAdd(&a.g_user_emb[(size_t)uid * d + col], gk[q][k]);
              if (iid >= 0 && iid < a.PI) atomicAdd(&a.g_item_emb[(size_t)iid * d + col], gk[q][k]);
            }
          }
        }
        if (a.use_seg && segq[q] < 3) {                    // row 3 is the padding_idx of seg_embeddings: no gradient
#pragma unroll
          for (int k = 0; k < NK; ++k) {
            const float v = lane + 64 * k < d ? gk[q][k] : 0.f;
            if (segq[q] == 0) sacc[0][k] += v;
            else if (segq[q] == 1) sacc[1][k] += v;
            else sacc[2][k] += v;
          }
        }
        // through dropout_layer (and, pv positive with train_pv, the PV drop_layer + the PV-loss gradient)
        float t[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          float v = src[q][k] * (ds.thr ? drop_word(ds, dw[k][q]) : 1.f);
          if (wantdv) {
            v += dvv[q][k];
            if (!pvc) v *= drop_mult(dpv, (uint32_t)rrq[q], (uint32_t)(lane + 64 * k));
          }
          t[k] = v;
        }
        if (!pvc && !a.det) {
          float* grow = a.g_table + (size_t)rclamp(ridq[q], rpad) * d;
#pragma unroll
          for (int k = 0; k < NK; ++k)
            if (lane + 64 * k < d) atomicAdd(&grow[lane + 64 * k], t[k]);
        } else if (!pvc) {
          // deterministic mode: the review row's gradient parked in place of d x (like the pvc rows below); rtm_ui_keys_kernel's
          // review keys + launch_rows_scatter_det add the rows up, one owner per table row, in position order
          float* gw = a.gs + ((size_t)nq[q] * a.S + sq[q]) * d;
#pragma unroll
          for (int k = 0; k < NK; ++k)
            if (lane + 64 * k < d) gw[lane + 64 * k] = t[k];
        } else {
          // mean backward: every non-pad word row of the review gets g / cnt (token corruption bypasses autograd, PVC.py:53)
          // 1.2 M word occurrences x 512 B of fp32 atomics per step ran at 0.7 TB/s; instead the row is rewritten in
          // place as the per-word gradient (rtm_wreduce_kernel sums the rows of a word's occurrences through the index)
          const float inv = 1.f / cntv[q];
          float* gw = a.gs + ((size_t)nq[q] * a.S + sq[q]) * d;
#pragma unroll
          for (int k = 0; k < NK; ++k)
            if (lane + 64 * k < d) gw[lane + 64 * k] = t[k] * inv;
        }
      }
    }
  } else {
    // ---- query positions (s = 0) of EB_QSEQ sequences: segment / user / item rows and d query_emb
    const int n0 = (wave - npos_w - nneg_w) * EB_QSEQ, nseq = a.B * a.J;
    for (int i0 = 0; i0 < EB_QSEQ && n0 + i0 < nseq; i0 += 4) {
      float gk[4][NK]; int bq[4], segq[4]; bool on[4], posq[4]; size_t sposq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + i0 + q;
        on[q] = n < nseq;
        const int nc = on[q] ? n : n0;
        const int b = fdiv(nc, a.fJ), j = nc - b * a.J;
        bq[q] = b; posq[q] = j == 0;
        sposq[q] = (posq[q] ? (size_t)b : (size_t)b * a.K + (j - 1)) * a.S;
        segq[q] = (int)(posq[q] ? a.pos_seg : a.neg_seg)[sposq[q]];
        const float* g = a.dx + (size_t)nc * a.S * d;
#pragma unroll
        for (int k = 0; k < NK; ++k) gk[q][k] = g[colc[k]];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (!on[q]) continue;
        if (has_ui) {
          const int64_t uid = a.g_user_emb ? (posq[q] ? a.pos_u : a.neg_u)[sposq[q]] : -1;
          const int64_t iid = a.g_item_emb ? (posq[q] ? a.pos_i : a.neg_i)[sposq[q]] : -1;
#pragma unroll
          for (int k = 0; k < NK; ++k) {
            const int col = lane + 64 * k;
            if (col < d) {
              if (uid >= 0 && uid < a.U) atomicAdd(&a.g_user_emb[(size_t)uid * d + col], gk[q][k]);
              if (iid >= 0 && iid < a.PI) atomicAdd(&a.g_item_emb[(size_t)iid * d + col], gk[q][k]);
            }
          }
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          const int col = lane + 64 * k;
          if (col >= d) continue;
          if (a.use_seg && segq[q] < 3) {
            if (segq[q] == 0) sacc[0][k] += gk[q][k];
            else if (segq[q] == 1) sacc[1][k] += gk[q][k];
            else sacc[2][k] += gk[q][k];
          }
          if (!a.det) atomicAdd(&a.dqe[(size_t)bq[q] * d + col], gk[q][k]);      // det: rtm_dqe_det_kernel
        }
      }
    }
  }
  if (!a.use_seg) return;
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int k = 0; k < NK; ++k)
      if (lane + 64 * k < d) segs[wv][q][lane + 64 * k] = sacc[q][k];
  __syncthreads();
  for (int e = threadIdx.x; e < 3 * d; e += 256) {
    const int q = e / d, col = e - q * d;
    seg_part[(size_t)blockIdx.x * 3 * d + e] = (segs[0][q][col] + segs[1][q][col]) + (segs[2][q][col] + segs[3][q][col]);
  }
}

// segment of every word in the occurrence list, cursor reset.  wreduce needs a word's occurrences CONTIGUOUS, not the
// segments in word order, so instead of a scan (one workgroup, 27 us for 32k words) every wave sums its 64 counts and
// bumps one running total (`tot`, zeroed with the counts) by the sum; the total ends as the list length.
__global__ __launch_bounds__(256) void rtm_walloc_kernel(const int* cnt, int* off, int* cur, int* tot, int V) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = w < V ? cnt[w] : 0;
  int incl = c;                                            // inclusive prefix over the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  const int total = __shfl(incl, 63, 64);
  int base = 0;
  if (lane == 63 && total > 0) base = atomicAdd(tot, total);
  base = __shfl(base, 63, 64);
  if (w < V) { off[w] = base + incl - c; cur[w] = 0; }
}

// ---- the inverted index word -> review slots: count, allocate (rtm_walloc_kernel), fill.
// The index depends on the batch's indices only, not on any gradient (rtm_index_in_forward).  Round 1 walked the slots one
// wave per slot, 38 slots in a row per wave: each step of that walk is a chain of dependent reads (review id -> word ids ->
// atomic), 56 us to count and 70 us to fill.  (Aggregating a workgroup's words in an LDS hash table first was measured and
// is slower: a chunk's words are mostly distinct, the popular ones are not what costs.)  Now
//   * the count rides in rtm_embed4_kernel, which has every word id in registers anyway (RtmK::count_fwd);
//   * the fill (and the count, when it does not ride) flattens a chunk of slots: the workgroup lists its real reviews, then
//     its threads stride over (review, word slot) pairs, four independent reads / atomics in flight per lane.
#define WI_CHUNK_MAX 256
template <int FILL>     // 0: count, 1: fill (a returning atomic on the word's cursor per occurrence), 2: fill from the ranks
__global__ __launch_bounds__(256) void rtm_windex_kernel(const RtmK a, int chunk, FDiv fWL) {
  __shared__ int l_rev[WI_CHUNK_MAX], l_slot[WI_CHUNK_MAX];     // rev: review row on its side, ~row for a positive
  __shared__ int l_n;
  const int tid = threadIdx.x, lane = tid & 63;
  const int nslots = a.B * a.J * a.S;
  const int64_t rpad = a.RC - 1;
  if (tid == 0) l_n = 0;
  __syncthreads();
  {
    const int slot = (int)blockIdx.x * chunk + tid;
    bool real = false; int enc = 0;
    if (tid < chunk && slot < nslots) {
      const int n = fdiv(slot, a.fS), s = slot - n * a.S;
      if (s > 0) {
        int b, j, seg, rev; int64_t ridx;
        seq_decode(a, n, s, b, j, ridx, rev, seg);
        real = ridx != rpad;
        enc = j == 0 ? ~rev : rev;
      }
    }
    const unsigned long long m = __ballot(real);
    int base = 0;
    if (lane == 0 && m) base = atomicAdd(&l_n, __popcll(m));
    base = __shfl(base, 0, 64);
    if (real) { const int at = base + __popcll(m & ((1ull << lane) - 1ull)); l_rev[at] = enc; l_slot[at] = slot; }
  }
  __syncthreads();
  const int total = l_n * a.WL;
  for (int i0 = tid; i0 < total; i0 += 4 * 256) {
    int64_t wi[4]; int sl[4], rk[4]; bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 256 * u;
      const bool in = i < total;
      const int r = in ? fdiv(i, fWL) : 0, w = i - r * a.WL;
      const int enc = l_rev[r];
      const bool pos = enc < 0;
      const int rev = pos ? ~enc : enc;
      const int64_t* words = (pos ? (a.train_pv ? a.pos_pvc : a.pos_words) : (a.train_pv ? a.neg_pvc : a.neg_words_rev));
      const uint8_t* wm = pos ? a.wmask_pos : a.wmask_neg;
      const size_t off = (size_t)rev * a.WL + (in ? w : 0);
      wi[u] = in ? words[off] : -1;
      if (FILL >= 2) {
        const size_t grow = (size_t)(pos ? 0 : a.B * a.R) + rev;
        rk[u] = in ? a.wrank[grow * a.WL + w] : -1;
        ok[u] = rk[u] >= 0;
      } else {
        ok[u] = in && word_ok(a, wm, off, wi[u]);
      }
      sl[u] = l_slot[r];
    }
    if (FILL == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (ok[u]) atomicAdd(&a.wcnt[wi[u]], 1);
    } else {
      int at[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        at[u] = !ok[u] ? 0 : a.woff[wi[u]] + (FILL >= 2 ? rk[u] : atomicAdd(&a.wcur[wi[u]], 1));
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (ok[u]) a.wl[at[u]] = make_int2(sl[u], (int)wi[u]);
    }
  }
}

// ---- the index WITHOUT global atomics (round 3).  Timing-only variants of rtm_embed4_kernel showed its per-occurrence returning
// atomics (1.16 M scattered 4-byte atomics: the memory-side rate for one dword per lane in 64 different lines) cost 45 of its
// 97 us, and kernels of such atomics slow whatever runs beside them.  Counting sort by word with the histogram in LDS instead:
//   rtm_hist_kernel      RTM_HIST_G workgroups, each owns a contiguous range of review rows and the WHOLE vocabulary as an LDS
//                        histogram (4 B x V <= 160 KB): an occurrence's rank inside its partition is a returning LDS atomic;
//                        ranks -> wrank (-1: not counted), the partition's histogram -> hist[g][.]
//   rtm_hist_scan_kernel per word: exclusive prefix of the partitions' counts, the word's total, and (folded in: the wave-sum +
//                        one bump of the running total that rtm_walloc_kernel does) the word's segment start; hist[g][w] becomes
//                        the list position of partition g's first occurrence of w
//   rtm_hist_fill_kernel the partitions again, their row of positions in LDS: position + rank, one 8-byte store per occurrence
// All of it on the side stream at the start of the backward; the forward's gather carries no atomics at all.
// (First form, measured: 128 partitions, a one-thread-per-word scan (26 us: 128 dependent strided reads per thread) and the fill
// as rtm_windex_kernel reading hist[g][w] per occurrence (59 us: a random 4-byte read in a 16 MB table each) — the gather fell
// 97 -> 52 us but the step ROSE 0.449 -> 0.462 ms.)
__device__ inline int hist_list_rows(const RtmK& a, int base, int row1, int* l_rev, int* l_n) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t rpad = a.RC - 1;
  if (tid == 0) *l_n = 0;
  __syncthreads();
  const int grow = base + tid;
  bool real = false;
  if (grow < row1) {
    const bool pos = grow < a.B * a.R;
    real = (pos ? a.pos_r[grow] : a.neg_r[grow - a.B * a.R]) != rpad;
  }
  const unsigned long long m = __ballot(real);
  int at0 = 0;
  if (lane == 0 && m) at0 = atomicAdd(l_n, __popcll(m));     // (one wave when the order matters: DET)
  at0 = __shfl(at0, 0, 64);
  if (real) l_rev[at0 + __popcll(m & ((1ull << lane) - 1ull))] = grow;
  __syncthreads();
  return *l_n;
}
// FILL 0: histogram + ranks, 1: fill the list from the ranks and the partition's positions.
// NT threads.  DET (deterministic mode, NT = 64): ONE wave walks the partition in task order and hands out the ranks without
// atomics — per batch of 64 occurrences a lane counts the lower lanes holding its word, every lane reads the word's counter,
// the last lane of each word adds the word's batch count — so a word's occurrences sit in its segment in task order.
template <int FILL, int NT, int DET>
__global__ __launch_bounds__(NT) void rtm_hist_kernel(const RtmK a) {
  extern __shared__ int hist_lds[];                 // [V]: FILL 0 the partition's counts, FILL 1 its list positions
  __shared__ int l_rev[NT];                         // this partition's real reviews: global review row
  __shared__ int l_n;
  static_assert(!DET || NT == 64, "the deterministic ranks are one wave's");
  const int tid = threadIdx.x;
  const int g = blockIdx.x, NP = a.B * a.R, NR = NP + a.B * a.K * a.R;
  for (int i = tid; i < a.V; i += NT) hist_lds[i] = FILL ? a.hist[(size_t)g * a.V + i] : 0;
  const int row0 = g * a.hist_rows, row1 = min(row0 + a.hist_rows, NR);
  for (int base = row0; base < row1; base += NT) {             // lists of up to NT rows at a time
    const int total = hist_list_rows(a, base, row1, l_rev, &l_n) * a.WL;
    for (int ib = 0; ib < total; ib += 4 * NT) {               // (workgroup-uniform trip count: DET shuffles below)
      int64_t wi[4]; size_t ro[4]; bool ok[4]; int rk[4], sl[4]; const uint8_t* wmo[4] = {nullptr, nullptr, nullptr, nullptr};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = ib + NT * u + tid;
        const bool in = i < total;
        const int ic = in ? i : 0;                             // (a slot past the end repeats slot 0: every load below is
        const int r = ic / a.WL, w = ic - r * a.WL;            // unconditional — under `in ? load : -1` each was its own branch
        const int gr = l_rev[r];                               // and the four, or eight, round trips of a pass ran one by one)
        const bool pos = gr < NP;
        const int rev = pos ? gr : gr - NP;
        const int64_t* words = (pos ? (a.train_pv ? a.pos_pvc : a.pos_words) : (a.train_pv ? a.neg_pvc : a.neg_words_rev));
        const size_t off = (size_t)rev * a.WL + w;
        wi[u] = words[off];
        if (!in) wi[u] = -1;
        ro[u] = (size_t)gr * a.WL + w;
        if (FILL) {
          rk[u] = a.wrank[ro[u]];
          if (!in) rk[u] = -1;
          ok[u] = rk[u] >= 0;
          // the slot of review row `rev` (the inverse of seq_decode: sequence n = b*J + j, position r + 1)
          const int seq = rev / a.R, r_in = rev - seq * a.R;
          int n;
          if (pos) n = seq * a.J;
          else { const int b = seq / a.K; n = b * a.J + (seq - b * a.K) + 1; }
          sl[u] = n * a.S + r_in + 1;
        } else {
          wmo[u] = (pos ? a.wmask_pos : a.wmask_neg) + off;
          ok[u] = in;
        }
      }
      if (!FILL) {                                             // word_ok, the mask bytes (if the batch has any) in one request
        uint8_t mk[4] = {1, 1, 1, 1};
        if (a.wmask_pos && a.wmask_neg) {
#pragma unroll
          for (int u = 0; u < 4; ++u) mk[u] = *wmo[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          ok[u] = ok[u] && ((a.wmask_pos && a.wmask_neg) ? mk[u] != 0 : wi[u] != a.V - 1) && wi[u] >= 0 && wi[u] < a.V;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = ib + NT * u + tid;
        if (FILL) {
          if (ok[u]) a.wl[hist_lds[wi[u]] + rk[u]] = make_int2(sl[u], (int)wi[u]);
        } else if (DET) {
          const int w = ok[u] ? (int)wi[u] : -1;
          int before = 0, same = 0;
          for (int l = 0; l < 64; ++l) {
            const int o = __shfl(w, l, 64);
            same += o == w ? 1 : 0;
            before += (o == w && l < tid) ? 1 : 0;
          }
          const int at = ok[u] ? hist_lds[w] : 0;              // every lane reads before any lane writes (one wave, in order)
          asm volatile("" ::: "memory");
          if (ok[u] && before == same - 1) hist_lds[w] = at + same;
          asm volatile("" ::: "memory");
          if (i < total) a.wrank[ro[u]] = ok[u] ? at + before : -1;
        } else if (i < total) {
          a.wrank[ro[u]] = ok[u] ? atomicAdd(&hist_lds[wi[u]], 1) : -1;
        }
      }
    }
    __syncthreads();
  }
  if (!FILL)
    for (int i = tid; i < a.V; i += NT) a.hist[(size_t)g * a.V + i] = hist_lds[i];
}
// 64 words x 8 groups of RTM_HIST_G/8 partitions per workgroup: every lane has its group's counts in registers at once
// (independent loads), the groups meet in LDS, wave 0 allocates the 64 segments (wave prefix + ONE bump of the running total)
__global__ __launch_bounds__(512) void rtm_hist_scan_kernel(int* hist, int* wcnt, int* woff, int* tot, int V) {
  __shared__ int l_tot[8][64];
  __shared__ int l_base[64];
  constexpr int PG = RTM_HIST_G / 8;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int w = blockIdx.x * 64 + lane;
  const bool in = w < V;
  int c[PG];
#pragma unroll
  for (int g = 0; g < PG; ++g) c[g] = in ? hist[(size_t)(wv * PG + g) * V + w] : 0;
  int run = 0;
#pragma unroll
  for (int g = 0; g < PG; ++g) { const int t = c[g]; c[g] = run; run += t; }
  l_tot[wv][lane] = run;
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) { const int t = l_tot[q][lane]; before += q < wv ? t : 0; total += t; }
  if (wv == 0) {
    int incl = total;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    const int all = __shfl(incl, 63, 64);
    int base = 0;
    if (lane == 63 && all > 0) base = atomicAdd(tot, all);
    base = __shfl(base, 63, 64) + incl - total;
    l_base[lane] = base;
    if (in) { woff[w] = base; wcnt[w] = total; }
  }
  __syncthreads();
  const int start = l_base[lane] + before;
  if (in) {
#pragma unroll
    for (int g = 0; g < PG; ++g) hist[(size_t)(wv * PG + g) * V + w] = start + c[g];
  }
}

// segmented sum over the word-sorted occurrence list: a wave owns 64 consecutive entries, adds the slot rows of a
// run of equal words in registers and issues ONE atomic row per run (runs spanning waves meet in the atomics)
__global__ __launch_bounds__(256) void rtm_wreduce_kernel(const RtmK a) {
  const int lane = threadIdx.x & 63;
  const int wave0 = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), nwave = (gridDim.x * blockDim.x) >> 6;
  const int T = a.wcnt[a.V];                         // the allocator's total (rtm_walloc_kernel)
  const int d = a.d;                                 // lane l owns columns l, l+64, ... (< d <= 512)
  // (the list length is only known on the device: a capped grid strides over it instead of one workgroup per 256 POSSIBLE
  // occurrences, 30,000 launches of which 25,000 found nothing to do)
  for (int base = wave0 * 64; base < T; base += nwave * 64) {
  const int e = base + lane;
  const int2 mine = e < T ? a.wl[e] : make_int2(-1, -1);
  const int my_slot = mine.x, my_word = mine.y;
  const int n = min(64, T - base);
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  int cur = __builtin_amdgcn_readlane(my_word, 0);
  if (d <= 128) {
    // d <= 128 (two columns per lane): 8 slot rows are requested before the first is added — one by one the loop is a
    // chain of 64 dependent L2 round trips per wave (149 us for the 594 MB of a C4 step)
    const int c0 = lane < d ? lane : d - 1, c1 = lane + 64 < d ? lane + 64 : d - 1;
    constexpr int WR_U = 16;                         // rows in flight (unconditional loads: a dead entry repeats row 0)
    for (int i0 = 0; i0 < n; i0 += WR_U) {
      float r0[WR_U], r1[WR_U];
#pragma unroll
      for (int u = 0; u < WR_U; ++u) {
        const int sl = __builtin_amdgcn_readlane(my_slot, (i0 + u) & 63);
        const bool live = i0 + u < n;
        const float* row = a.gs + (size_t)(live ? sl : 0) * d;
        r0[u] = row[c0];
        r1[u] = row[c1];
      }
#pragma unroll
      for (int u = 0; u < WR_U; ++u) {
        if (i0 + u < n) {                            // wave-uniform
          const int w = __builtin_amdgcn_readlane(my_word, (i0 + u) & 63);
          if (w != cur) {
            float* grow = a.g_word_emb + (size_t)cur * d;
            if (lane < d) { atomicAdd(&grow[lane], acc[0]); acc[0] = 0.f; }
            if (lane + 64 < d) { atomicAdd(&grow[lane + 64], acc[1]); acc[1] = 0.f; }
            cur = w;
          }
          acc[0] += r0[u]; acc[1] += r1[u];
        }
      }
    }
  } else
  for (int i = 0; i < n; ++i) {
    const int w = __shfl(my_word, i, 64), sl = __shfl(my_slot, i, 64);
    if (w != cur) {
      float* grow = a.g_word_emb + (size_t)cur * d;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (lane + 64 * k < d) { atomicAdd(&grow[lane + 64 * k], acc[k]); acc[k] = 0.f; }
      cur = w;
    }
    const float* row = a.gs + (size_t)sl * d;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (lane + 64 * k < d) acc[k] += row[lane + 64 * k];
  }
  float* grow = a.g_word_emb + (size_t)cur * d;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (lane + 64 * k < d) atomicAdd(&grow[lane + 64 * k], acc[k]);
  }
}

// uncorrupted pvc review table for eval: out[i] = mean of the review's word rows, last row 0 (ps_model.py:186-203)
__global__ __launch_bounds__(256) void rtm_review_table_kernel(const float* word_emb, const int64_t* review_words, float* out,
                                                               int64_t RC, int WL, int d, int64_t V) {
  const int lane = threadIdx.x & 63;
  const int64_t rev = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (rev >= RC) return;
  for (int e = lane; e < d; e += 64) {
    float s = 0.f; int cnt = 0;
    if (rev < RC - 1)
      for (int w = 0; w < WL; ++w) {
        const int64_t wi = review_words[rev * WL + w];
        if (wi != V - 1 && wi >= 0 && wi < V) { s += word_emb[(size_t)wi * d + e]; ++cnt; }
      }
    out[rev * d + e] = s / (float)(cnt > 0 ? cnt : 1);
  }
}

// ------------------------------------------------------------------ fs review encoder (text_encoder.py:32-40)
// forward tail: x[n][s] = (valid ? tanh-projection + segment / user / item rows : 0) + pe[s]; one wave per review slot
__global__ __launch_bounds__(256) void rtm_fs_finish_kernel(const RtmK a) {
  const int lane = threadIdx.x & 63;
  const int slot = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (slot >= a.B * a.J * a.S) return;
  const int n = fdiv(slot, a.fS), s = slot - n * a.S;
  if (s == 0) return;                                   // query rows were written by the embed kernel
  int b, j, revrow, seg; int64_t ridx; size_t spos;
  seq_decode(a, n, s, b, j, ridx, revrow, seg, &spos);
  const bool pos = j == 0, ok = ridx != a.RC - 1;
  const size_t rg = pos ? (size_t)revrow : (size_t)a.B * a.R + revrow;
  int64_t uid = -1, iid = -1;
  if (a.user_emb) { uid = (pos ? a.pos_u : a.neg_u)[spos]; if (uid < 0 || uid > a.U) uid = -1; }
  if (a.item_emb) { iid = (pos ? a.pos_i : a.neg_i)[spos]; if (iid < 0 || iid > a.PI) iid = -1; }
  const int d = a.d;
  for (int col = lane; col < d; col += 64) {
    float val = 0.f;
    if (ok) {
      val = a.yfs[rg * d + col];
      if (a.use_seg) val += a.seg_emb[(size_t)seg * d + col];
      if (uid >= 0) val += a.user_emb[(size_t)uid * d + col];
      if (iid >= 0) val += a.item_emb[(size_t)iid * d + col];
    }
    if (a.use_pos) val += a.pe[(size_t)s * d + col];
    a.x[((size_t)n * a.S + s) * d + col] = val;
  }
}
// backward head: d pre[row] = valid ? dx[n][s] * (1 - y^2) : 0 for every review slot; bias gradient = its column sums
__global__ __launch_bounds__(256) void rtm_fs_bwd_kernel(const RtmK a, float* dpre, float* g_bias) {
  extern __shared__ float bsum[];                  // [d]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int d = a.d;
  for (int e = threadIdx.x; e < d; e += 256) bsum[e] = 0.f;
  __syncthreads();
  const int nslots = a.B * a.J * a.S, nw = gridDim.x * 4;
  for (int slot = blockIdx.x * 4 + wv; slot < nslots; slot += nw) {
    const int n = fdiv(slot, a.fS), s = slot - n * a.S;
    if (s == 0) continue;
    int b, j, revrow, seg; int64_t ridx;
    seq_decode(a, n, s, b, j, ridx, revrow, seg);
    const bool ok = ridx != a.RC - 1;
    const size_t rg = j == 0 ? (size_t)revrow : (size_t)a.B * a.R + revrow;
    for (int col = lane; col < d; col += 64) {
      float v = 0.f;
      if (ok) {
        const float y = a.yfs[rg * d + col];
        v = a.dx[((size_t)n * a.S + s) * d + col] * (1.f - y * y);
        if (!a.det) atomicAdd(&bsum[col], v);
      }
      dpre[rg * d + col] = v;
    }
  }
  __syncthreads();
  if (a.det) return;                               // rtm_colsum_det_kernel over dpre
  for (int e = threadIdx.x; e < d; e += 256) atomicAdd(&g_bias[e], bsum[e]);
}

// ------------------------------------------------------------------ host side
static void fill_k(const PsRtmDesc& D, const PsRtmTensors& P, const PsRtmBatch& Bt, float* ws, const RtmWs& r, const Ws& w,
                   bool eval, RtmK& k) {
  memset(&k, 0, sizeof(k));
  k.B = D.B; k.J = r.J; k.K = D.K; k.R = D.R; k.S = r.S; k.Q = D.Q; k.W = D.W > 0 ? D.W : 1; k.WL = D.WL; k.d = D.d;
  k.V = D.vocab_size; k.RC = D.review_count;
  k.det = ps_deterministic() ? 1 : 0;
  static const bool rtm_stamps = ps_diag_int("PS_RTM_STAMP", 0) != 0;
  k.stamp = rtm_stamps ? ps_debug_stamp_ptr() : nullptr;
  // `pvc` = every word-mean review encoder in training (pvc, fs, avg); only pvc corrupts tokens and ignores the word masks
  k.pvc = D.review_encoder != PS_RENC_PV && !eval; k.use_pos = D.use_pos_emb; k.use_seg = D.use_seg_emb;
  const bool masked_mean = !eval && (D.review_encoder == PS_RENC_FS || D.review_encoder == PS_RENC_AVG);
  if (masked_mean) { k.wmask_pos = Bt.pos_prod_rword_masks; k.wmask_neg = Bt.neg_prod_rword_masks; }
  if (!eval && D.review_encoder == PS_RENC_FS) { k.raw = ws + r.raw; k.yfs = ws + r.yfs; }
  k.pos_weight = D.pos_weight; k.train_pv = eval ? 0 : D.train_pv; k.training = eval ? 0 : D.training; k.eval = eval;
  k.fJ = make_fdiv(r.J); k.fS = make_fdiv(r.S); k.fK1 = make_fdiv(D.K + 1);
  PsTemDesc dd;
  memset(&dd, 0, sizeof(dd));
  dd.training = k.training; dd.dropout = D.dropout; dd.seed = D.seed; dd.step = D.step;
  k.d_pv = make_drop(dd, SITE_REV_PV); k.d_pos = make_drop(dd, SITE_REV_POS); k.d_neg = make_drop(dd, SITE_REV_NEG);
  dd.dropout = D.review_encoder == PS_RENC_PVC ? D.corrupt_rate : 0.f;
  k.t_pos = make_drop(dd, SITE_TOK_POS); k.t_neg = make_drop(dd, SITE_TOK_NEG);
  k.U = D.user_size; k.PI = D.product_size;
  if (D.use_user_emb) { k.user_emb = P.user_emb; k.pos_u = Bt.pos_user_idxs; k.neg_u = eval ? Bt.candi_seq_user_idxs : Bt.neg_user_idxs; }
  if (D.use_item_emb) { k.item_emb = P.product_emb; k.pos_i = Bt.pos_item_idxs; k.neg_i = eval ? Bt.candi_seq_item_idxs : Bt.neg_item_idxs; }
  if (eval) {
    k.neg_r = Bt.candi_prod_ridxs; k.neg_seg = Bt.candi_seg_idxs; k.table = Bt.review_embeddings;
  } else {
    k.pos_r = Bt.pos_prod_ridxs; k.neg_r = Bt.neg_prod_ridxs; k.pos_seg = Bt.pos_seg_idxs; k.neg_seg = Bt.neg_seg_idxs;
    k.pos_words = Bt.pos_prod_rword_idxs; k.neg_words_rev = Bt.neg_prod_rword_idxs;
    k.pos_pvc = Bt.pos_prod_rword_idxs_pvc; k.neg_pvc = Bt.neg_prod_rword_idxs_pvc;
    k.neg_word_idxs = Bt.neg_word_idxs; k.pos_masks = Bt.pos_prod_rword_masks;
    k.table = P.review_emb;
  }
  k.word_emb = P.word_emb; k.seg_emb = P.seg_emb; k.pe = P.pe; k.wo_w = P.wo_w; k.wo_b = P.wo_b;
  float* we = ws + r.enc_base;
  k.vrows = reinterpret_cast<int32_t*>(we + w.vrows); k.vcount = reinterpret_cast<int32_t*>(we + w.vcount);
  k.query_emb = ws + r.query_emb; k.x = we + w.x; k.valid = ws + r.valid; k.vec = ws + r.vec; k.cnt = ws + r.cnt;
  k.enc = we + w.enc; k.scores = ws + r.scores; k.weight = ws + r.weight; k.pv_scores = ws + r.pv_scores; k.pv_terms = ws + r.pv_terms;
  k.nvalid = ws + r.nvalid;
  k.seqcnt = reinterpret_cast<int32_t*>(ws + r.seqcnt);
  k.dx = we + w.dx; k.denc = we + w.denc; k.dvec = ws + r.dvec; k.dqe = ws + r.dqe;
  if (r.wcnt) {
    k.wcnt = (int*)(ws + r.wcnt); k.woff = (int*)(ws + r.woff); k.wcur = (int*)(ws + r.wcur);
    k.wl = (int2*)(ws + r.wl); k.wrank = (int*)(ws + r.wrank);
    k.hist = r.hist ? (int*)(ws + r.hist) : nullptr;
    k.hist_rows = ps_cdiv((int64_t)r.Bseq * D.R, RTM_HIST_G);
  }
}

// The inverted index word -> review slots of the pvc / fs / avg backward depends on the batch's indices only.  Its first
// pass (per-word counts, and with them every occurrence's rank inside its word) rides in the training forward when that
// embeds through rtm_embed4_kernel, which holds every word id in registers; the backward then only allocates and fills,
// on the side stream under its first kernels.
static bool rtm_embed4_taken(const PsRtmDesc& D, const RtmK& k, const RtmWs& r) {
  static const bool e4_on = ps_diag_int("PS_RTM_EMBED4", 1) != 0;
  return e4_on && k.pvc && !k.eval && D.WL <= 128 && (D.d == 64 || D.d == 128 || D.d == 256) && r.S <= 64;
}
// (Round 3, measured and dropped: the whole index built on the side stream BESIDE the forward, so that the forward carries no
// atomics.  Kernels of scattered global atomics poison whatever runs next to them: the gather launch stayed at 97 us without its
// atomics, a 3 us list kernel took 44 us, the step went 0.454 -> 0.507 ms.  Timing-only variants of rtm_embed4_kernel show the
// rank atomics cost 45 of its 97 us — the fix is fewer global atomics, not a different place for them.)
// the LDS-histogram index (rtm_hist_kernel): the backward builds the whole index on its side stream without global atomics and
// the forward carries none.  PS_RTM_HIST=0: the round-2 form (ranks by global atomics in the forward's gather).
static bool rtm_hist_index(const PsRtmDesc& D, const RtmK& k, const RtmWs& r) {
  static const bool on = ps_env_int("PS_RTM_HIST", 1) != 0;
  static const bool late = ps_diag_int("PS_RTM_LATE_INDEX", 0) != 0;
  return on && !late && k.pvc && !k.eval && r.hist != 0 && D.vocab_size <= RTM_HIST_MAXV && r.S == D.R + 1 &&
         ps_cdiv((int64_t)r.Bseq * D.R, RTM_HIST_G) <= (1 << 20);
}
static bool rtm_counts_in_forward(const PsRtmDesc& D, const RtmK& k, const RtmWs& r) {
  static const bool late = ps_diag_int("PS_RTM_LATE_INDEX", 0) != 0;
  return !late && !rtm_hist_index(D, k, r) && rtm_embed4_taken(D, k, r);
}
static int rtm_build_index_hist(const RtmK& k, const RtmWs& r, int V, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    const int lim = RTM_HIST_MAXV * (int)sizeof(int);
    PS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rtm_hist_kernel<0, 1024, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    PS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rtm_hist_kernel<0, 64, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    PS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rtm_hist_kernel<1, 1024, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    attr = true;
  }
  (void)r;
  PS_CHECK_HIP(hipMemsetAsync(k.wcnt + V, 0, sizeof(int), st));       // the allocator's running total
  if (k.det) hipLaunchKernelGGL((rtm_hist_kernel<0, 64, 1>), dim3(RTM_HIST_G), dim3(64), (size_t)V * sizeof(int), st, k);
  else hipLaunchKernelGGL((rtm_hist_kernel<0, 1024, 0>), dim3(RTM_HIST_G), dim3(1024), (size_t)V * sizeof(int), st, k);
  PS_LAUNCH_CHECK();
  hipLaunchKernelGGL(rtm_hist_scan_kernel, dim3(ps_cdiv(V, 64)), dim3(512), 0, st, k.hist, k.wcnt, k.woff, k.wcnt + V, V);
  PS_LAUNCH_CHECK();
  hipLaunchKernelGGL((rtm_hist_kernel<1, 1024, 0>), dim3(RTM_HIST_G), dim3(1024), (size_t)V * sizeof(int), st, k);
  PS_LAUNCH_CHECK();
  return PS_OK;
}
// count (unless a kernel that reads the words anyway did), allocate, fill
static int rtm_build_index(const RtmK& k, const RtmWs& r, int V, bool count, hipStream_t st) {
  static const int env_chunk = ps_diag_int("PS_RTM_IDX_CHUNK", 64);
  const int nslots = r.Bseq * r.S;
  const int chunk = env_chunk < 1 ? 1 : (env_chunk > WI_CHUNK_MAX ? WI_CHUNK_MAX : env_chunk), nwg = ps_cdiv(nslots, chunk);
  const FDiv fWL = make_fdiv(k.WL > 0 ? k.WL : 1);
  if (count) {
    hipLaunchKernelGGL(rtm_windex_kernel<0>, dim3(nwg), dim3(256), 0, st, k, chunk, fWL);
    PS_LAUNCH_CHECK();
  }
  // the allocator's running total starts at 0 on EVERY build: a second backward over the same forward (retain_graph) must
  // lay the list out from the start again, not behind the first one (the counts themselves are the forward's and stay)
  if (!count) PS_CHECK_HIP(hipMemsetAsync(k.wcnt + V, 0, sizeof(int), st));
  hipLaunchKernelGGL(rtm_walloc_kernel, dim3(ps_cdiv(V, 256)), dim3(256), 0, st, k.wcnt, k.woff, k.wcur, k.wcnt + V, V);
  PS_LAUNCH_CHECK();
  if (k.count_fwd) hipLaunchKernelGGL(rtm_windex_kernel<2>, dim3(nwg), dim3(256), 0, st, k, chunk, fWL);
  else hipLaunchKernelGGL(rtm_windex_kernel<1>, dim3(nwg), dim3(256), 0, st, k, chunk, fWL);
  PS_LAUNCH_CHECK();
  return PS_OK;
}
// workgroups of the slot-walking kernels (measured, ms/step: 128 0.850, 256 0.764, 384 0.762, 512 0.734, 1024 0.749,
// 2048 0.784, 4096 0.837)
static int rtm_slot_blocks(const RtmWs& r) {
  static const int eb_cap = ps_diag_int("PS_RTM_EB", 512);
  int eb = ps_cdiv(r.Bseq * r.S, 4);
  return eb > eb_cap ? eb_cap : eb;
}

static void to_tem_tensors(const PsRtmTensors& R, PsTemTensors& T) {
  memset(&T, 0, sizeof(T));
  T.word_emb = R.word_emb; T.fs_w = R.fs_w; T.fs_b = R.fs_b; T.pe = R.pe;
  T.final_ln_g = R.final_ln_g; T.final_ln_b = R.final_ln_b;
  for (int i = 0; i < PS_MAX_LAYERS; ++i) T.layer[i] = R.layer[i];
}

// the row list costs Bseq^2 / 2 count reads (rtm_rowlist_kernel): built up to 8k sequences, dense products beyond
static bool rtm_rows_listed(const RtmWs& r, const Ws& w) { return r.S <= 64 && w.vrows != 0 && r.Bseq <= 8192; }

static int rtm_encode(const PsRtmDesc& D, const PsRtmTensors& P, const PsRtmBatch& Bt, float* ws, const RtmWs& r,
                      const Ws& w, const PsTemDesc& E, bool eval, RtmK& k, hipStream_t st) {
  const int B = D.B, d = D.d;
  PS_REQUIRE(P.word_emb && P.seg_emb && P.wo_w && P.wo_b && Bt.query_word_idxs, "rtm: null tensors");
  fill_k(D, P, Bt, ws, r, w, eval, k);
  PS_REQUIRE(k.neg_r && k.neg_seg && (eval || (k.pos_r && k.pos_seg)), "rtm: null review index tensors");
  PS_REQUIRE(k.pvc || k.table, "rtm: null review embedding table");
  PS_REQUIRE(!D.use_user_emb || (k.user_emb && k.neg_u && (eval || k.pos_u)), "rtm: use_user_emb needs user_emb and the user index tensors");
  PS_REQUIRE(!D.use_item_emb || (k.item_emb && k.neg_i && (eval || k.pos_i)), "rtm: use_item_emb needs product_emb and the item index tensors");
  if (k.pvc) PS_REQUIRE(k.train_pv ? (k.pos_pvc && k.neg_pvc) : (k.pos_words && k.neg_words_rev), "rtm: null review word tensors");
  if (!eval && (D.review_encoder == PS_RENC_FS || D.review_encoder == PS_RENC_AVG))
    PS_REQUIRE(k.wmask_pos && k.wmask_neg, "rtm: the fs / avg review encoders need the batch's word masks");
  if (k.train_pv) PS_REQUIRE(k.pos_words && k.pos_masks && k.neg_word_idxs, "rtm: null PV-loss tensors");
  // query encoder (shared kernels): masked mean (+FS dropout) then tanh(f_W . + b)
  PsTemDesc dq;
  memset(&dq, 0, sizeof(dq));
  dq.training = k.training; dq.dropout = D.dropout; dq.seed = D.seed; dq.step = D.step;
  EmbedArgs e;
  memset(&e, 0, sizeof(e));
  e.B = B; e.Q = D.Q; e.L = 0; e.S = 1; e.d = d; e.P = 1; e.V = D.vocab_size; e.tem = 0;
  e.fs = D.query_encoder == PS_QENC_FS; e.qw = Bt.query_word_idxs; e.word_emb = P.word_emb;
  e.drop_fs = make_drop(dq, PS_SITE_FS);
  e.qmean_d = ws + r.qmean; e.query_emb = ws + r.query_emb;
  PS_REQUIRE(!e.fs || (P.fs_w && P.fs_b), "rtm: null FS encoder weights");
  const bool fs_fused = e.fs && ps_fusion_enabled() && d <= 128;
  if (fs_fused) { e.fs_w = P.fs_w; e.fs_b = P.fs_b; }
  {
    PsTemTensors Ts;
    to_tem_tensors(P, Ts);
    e.split = make_wsplit(E, Ts, ws + r.enc_base, w);      // the fused kernels' bf16x3 weight planes ride in this launch
  }
  // the backward's inverted index (rtm_counts_in_forward): its counters are cleared by this launch, filled by the next
  const bool use_e4 = rtm_embed4_taken(D, k, r);
  k.count_fwd = rtm_counts_in_forward(D, k, r);
  if (k.count_fwd) { e.zero_i32 = k.wcnt; e.zero_n = (int)D.vocab_size + 1; }
  if (!eval) e.clear_word = reinterpret_cast<uint32_t*>(ws + r.loss_blk);   // rtm_score_kernel's 64-bit loss word
  TRY(launch_embed_fwd(e, st));
  if (e.fs && !fs_fused) {
    GemmProblem p = gp(ws + r.qmean, d, 0, P.fs_w, d, 0, ws + r.query_emb, d, B, d, d);
    p.bias = P.fs_b; p.act = ACT_TANH;
    TRY(run1(p, st));
  }
  PS_REQUIRE(!D.use_pos_emb || P.pe, "rtm: null positional table");
  const int nslots = r.Bseq * r.S;
  {
    KTimeScope kt("rtm_embed", st);
    if (use_e4) {
      const int npos = ps_cdiv((int64_t)B * D.R, 4), nneg = ps_cdiv((int64_t)B * D.K * D.R, 4);
      const dim3 grid(ps_cdiv(npos + nneg + r.Bseq, 4));
      // x rows of padded positions: with the valid-row list nothing downstream reads them (the projections, the attention and
      // the backward all walk the list), so they are not written either (18 us of the launch at C4, 73 % padding)
      static const bool keep_pads = ps_diag_int("PS_RTM_WRITE_PADS", 0) != 0;
      const int pads_unread = enc_rowlist_taken(E, w, rtm_rows_listed(r, w)) && !k.raw && !keep_pads ? 1 : 0;
      const FDiv fR = make_fdiv(D.R), fK = make_fdiv(D.K > 0 ? D.K : 1);
      // the valid-group list (rtm_grouplist_kernel): only when padded rows of x are not read and the counter is cleared by
      // the query-encoder launch (training).  PS_RTM_GROUPLIST=0: every group gets a wave, as in round 2.
      static const int e4_diag = ps_diag_int("PS_RTM_DIAG", 0);      // timing experiments (wrong results)
      static const bool list_on = ps_env_int("PS_RTM_GROUPLIST", 1) != 0;
      const int* glist = nullptr; const int* gcount = nullptr;
      const int nq_wg = ps_cdiv(r.Bseq, 4);
      dim3 egrid = grid;
      if (list_on && pads_unread && !eval && r.glist && !k.train_pv) {
        int* gl = reinterpret_cast<int*>(ws + r.glist);
        int* gc = reinterpret_cast<int*>(ws + r.loss_blk) + 2;
        hipLaunchKernelGGL(rtm_grouplist_kernel, dim3(ps_cdiv(npos + nneg, 256)), dim3(256), 0, st, k, npos, nneg, fR, fK, gl, gc);
        PS_LAUNCH_CHECK();
        glist = gl; gcount = gc;
        egrid = dim3(nq_wg + ps_cdiv(npos + nneg, 4));
      }
      if (d == 64) PS_KLAUNCH(rtm_embed4_kernel<1>, egrid, dim3(256), 0, st, k, npos, nneg, fR, fK, pads_unread, glist, gcount, nq_wg, e4_diag);
      else if (d == 128) PS_KLAUNCH(rtm_embed4_kernel<2>, egrid, dim3(256), 0, st, k, npos, nneg, fR, fK, pads_unread, glist, gcount, nq_wg, e4_diag);
      else PS_KLAUNCH(rtm_embed4_kernel<4>, egrid, dim3(256), 0, st, k, npos, nneg, fR, fK, pads_unread, glist, gcount, nq_wg, e4_diag);
    } else {
      PS_KLAUNCH(rtm_embed_kernel, dim3(ps_cdiv(nslots, 4)), dim3(256), 0, st, k);
    }
  }
  PS_LAUNCH_CHECK();
  if (k.raw) {   // fs review encoder: y = tanh(f_W . raw + b) for every review slot, then the rest of x
    PS_REQUIRE(P.rev_fs_w && P.rev_fs_b, "rtm: null review-encoder f_W");
    GemmProblem p = gp(ws + r.raw, d, 0, P.rev_fs_w, d, 0, ws + r.yfs, d, r.Bseq * D.R, d, d);
    p.bias = P.rev_fs_b; p.act = ACT_TANH;
    TRY(run1(p, st));
    hipLaunchKernelGGL(rtm_fs_finish_kernel, dim3(ps_cdiv(nslots, 4)), dim3(256), 0, st, k);
    PS_LAUNCH_CHECK();
  }
  const bool listed = rtm_rows_listed(r, w);
  if (listed) {
    hipLaunchKernelGGL(rtm_rowlist_kernel, dim3(ps_cdiv(r.Bseq, 4)), dim3(256), 0, st, k);
    PS_LAUNCH_CHECK();
  }
  PsTemTensors T;
  to_tem_tensors(P, T);
  TRY(enc_layers_forward(E, T, nullptr, ws + r.valid, ws + r.enc_base, w, st, listed));
  return PS_OK;
}

extern "C" int ps_rtm_forward(const PsRtmDesc* desc, const PsRtmTensors* params, const PsRtmBatch* batch, float* ws,
                              float* loss3, ps_stream_t stream) {
  PS_REQUIRE(desc && params && batch && ws && loss3, "rtm forward: null argument");
  RtmWs r; Ws w; PsTemDesc E; RtmK k;
  TRY(rtm_make_ws(*desc, false, r, w, E));
  hipStream_t st = (hipStream_t)stream;
  TRY(rtm_encode(*desc, *params, *batch, ws, r, w, E, false, k, st));
  k.loss3 = loss3;
  const bool fold_loss = !k.train_pv;               // (the PV loss needs rtm_pv_fwd_kernel's terms: rtm_loss_kernel then)
  PS_REQUIRE(!fold_loss || ps_cdiv(r.Bseq, 4) < 65536, "rtm forward: too many sequences for the folded loss");
  hipLaunchKernelGGL(rtm_score_kernel, dim3(ps_cdiv(r.Bseq, 4)), dim3(256), 0, st, k, k.scores, fold_loss ? 1 : 0,
                     reinterpret_cast<unsigned long long*>(ws + r.loss_blk));
  PS_LAUNCH_CHECK();
  if (fold_loss) return PS_OK;
  if (k.train_pv) {
    const int ntask = k.B * k.R * k.W * (k.K + 1);
    int lpr = 1; while (lpr < k.d / 4 && lpr < 64) lpr <<= 1;
    hipLaunchKernelGGL(rtm_pv_fwd_kernel, dim3(ps_cdiv(ps_cdiv(ntask, PV_U), 256 / lpr)), dim3(256), 0, st, k, ntask, lpr);
    PS_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(rtm_loss_kernel, dim3(1), dim3(256), 0, st, k);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

extern "C" int ps_rtm_score(const PsRtmDesc* desc, const PsRtmTensors* params, const PsRtmBatch* batch, float* ws,
                            float* scores, ps_stream_t stream) {
  PS_REQUIRE(desc && params && batch && ws && scores, "rtm score: null argument");
  PS_REQUIRE(desc->C > 0 && batch->candi_prod_ridxs && batch->candi_seg_idxs && batch->review_embeddings,
             "rtm score: needs candidates and the review-embedding table");
  RtmWs r; Ws w; PsTemDesc E; RtmK k;
  TRY(rtm_make_ws(*desc, true, r, w, E));
  hipStream_t st = (hipStream_t)stream;
  TRY(rtm_encode(*desc, *params, *batch, ws, r, w, E, true, k, st));
  hipLaunchKernelGGL(rtm_score_kernel, dim3(ps_cdiv(r.Bseq, 4)), dim3(256), 0, st, k, scores, 0, nullptr);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

extern "C" int ps_rtm_review_embeddings(const PsRtmDesc* desc, const PsRtmTensors* params, const int64_t* review_words,
                                        float* scratch, float* out, ps_stream_t stream) {
  PS_REQUIRE(desc && params && review_words && out && params->word_emb && desc->WL > 0, "rtm review table: bad argument");
  const bool fs = desc->review_encoder == PS_RENC_FS;
  PS_REQUIRE(!fs || (scratch && params->rev_fs_w && params->rev_fs_b), "rtm review table: fs needs scratch and f_W");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rtm_review_table_kernel, dim3(ps_cdiv(desc->review_count, 4)), dim3(256), 0, st,
                     params->word_emb, review_words, fs ? scratch : out, desc->review_count, desc->WL, desc->d, desc->vocab_size);
  PS_LAUNCH_CHECK();
  if (fs) {
    const int d = desc->d;
    // The reference fills the table in slices of 128 reviews up to row ceil((RC-1)/128)*128 (ps_model.py:190-203): unless
    // RC-1 is a multiple of 128 that takes in the padding review, whose empty mean is projected like any other —
    // its row is tanh(bias), not 0 (its positions are masked in every sequence, so only the table itself shows it)
    const int64_t rc = desc->review_count;
    const bool pad_projected = (rc - 1) % 128 != 0;
    const int64_t n = pad_projected ? rc : rc - 1;
    PS_REQUIRE(n < ((int64_t)1 << 31), "rtm review table: too many reviews");
    if (n > 0) {
      GemmProblem p = gp(scratch, d, 0, params->rev_fs_w, d, 0, out, d, (int)n, d, d);
      p.bias = params->rev_fs_b; p.act = ACT_TANH;
      TRY(run1(p, st));
    }
    if (!pad_projected) PS_CHECK_HIP(hipMemsetAsync(out + (size_t)n * d, 0, sizeof(float) * d, st));
  }
  return PS_OK;
}

static int rtm_backward_impl(const PsRtmDesc* desc, const PsRtmTensors* params, const PsRtmBatch* batch, float* ws,
                             const PsRtmTensors* grads, float loss_scale, const float* loss_scale_dev, ps_stream_t stream);
extern "C" int ps_rtm_backward(const PsRtmDesc* desc, const PsRtmTensors* params, const PsRtmBatch* batch, float* ws,
                               const PsRtmTensors* grads, float loss_scale, const float* loss_scale_dev,
                               ps_stream_t stream) {
  const int rc = rtm_backward_impl(desc, params, batch, ws, grads, loss_scale, loss_scale_dev, stream);
  if (rc != PS_OK) side_abort();          // never leave the side stream waiting behind a failed call (tem.hip)
  return rc;
}
static int rtm_backward_impl(const PsRtmDesc* desc, const PsRtmTensors* params, const PsRtmBatch* batch, float* ws,
                             const PsRtmTensors* grads, float loss_scale, const float* loss_scale_dev, ps_stream_t stream) {
  PS_REQUIRE(desc && params && batch && ws && grads, "rtm backward: null argument");
  const PsRtmDesc& D = *desc;
  RtmWs r; Ws w; PsTemDesc E; RtmK k;
  TRY(rtm_make_ws(D, false, r, w, E));
  hipStream_t st = (hipStream_t)stream;
  fill_k(D, *params, *batch, ws, r, w, false, k);
  const PsRtmTensors& G = *grads;
  PS_REQUIRE(G.word_emb && (!D.use_seg_emb || G.seg_emb) && G.wo_w && G.wo_b && (k.pvc || G.review_emb),
             "rtm backward: null gradients");
  k.scale = loss_scale; k.scale_dev = loss_scale_dev;
  k.g_word_emb = G.word_emb; k.g_table = G.review_emb; k.g_seg_emb = G.seg_emb; k.g_wo_w = G.wo_w; k.g_wo_b = G.wo_b;
  if (D.use_user_emb) { PS_REQUIRE(G.user_emb, "rtm backward: null user_emb gradient"); k.g_user_emb = G.user_emb; }
  if (D.use_item_emb) { PS_REQUIRE(G.product_emb, "rtm backward: null product_emb gradient"); k.g_item_emb = G.product_emb; }
  const int B = D.B, d = D.d;
  const bool hist_index = rtm_hist_index(D, k, r);
  const bool fwd_index = hist_index || rtm_counts_in_forward(D, k, r);     // either way: built on the side stream, right here
  if (k.det) {
    // Deterministic mode covers every review encoder (pvc — BASELINE configs[3] —, fs, avg: the word index + ordered reduce; pv:
    // the review rows through the sole-owner row scatter), the user / item embedding rows and the PV loss's word rows.
    PS_REQUIRE(!k.pvc || hist_index, "rtm backward: deterministic mode needs the LDS-histogram word index (vocabulary <= %d, PS_RTM_HIST != 0)",
               RTM_HIST_MAXV);
  }
  k.count_fwd = fwd_index && !hist_index;
  if (fwd_index) {    // [count +] allocate + fill on the side stream (or here, without one), under the fused kernel and the attention
    hipStream_t ss = side_stream_or(st);
    if (ss != st) { side_set_light(false); TRY(side_fork(st)); }
    if (hist_index) TRY(rtm_build_index_hist(k, r, (int)D.vocab_size, ss));
    else TRY(rtm_build_index(k, r, (int)D.vocab_size, false, ss));
  }
  int blocks = ps_cdiv(r.Bseq, 4); if (blocks > 256) blocks = 256;
  uint32_t* sig = nullptr; uint32_t sigval = 0;
  static const bool sbwd_carries = ps_diag_int("PS_RTM_SBWD_SIG", 1) != 0;
  if (sbwd_carries) side_take_signal(st, &sig, &sigval);       // the index fill's fork rides on this launch
  hipLaunchKernelGGL(rtm_score_bwd_kernel, dim3(blocks), dim3(256), (size_t)(d + 1) * sizeof(float), st, k, fwd_index || !k.pvc ? 1 : 0, sig, sigval);
  PS_LAUNCH_CHECK();
  if (k.det) {
    PS_REQUIRE(d <= 512, "rtm backward: deterministic mode supports d <= 512");
    hipLaunchKernelGGL(rtm_score_wo_det_kernel, dim3(1), dim3(256), 0, st, k);
    PS_LAUNCH_CHECK();
  }
  if (k.train_pv) {
    hipLaunchKernelGGL(rtm_pv_bwd_kernel, dim3(ps_cdiv(B * k.R, 4)), dim3(256), 0, st, k);
    PS_LAUNCH_CHECK();
    if (k.det) {                                     // the PV loss's word rows: sole-owner scatter of ds * vec[review]
      const int ntask = B * k.R * k.W * (k.K + 1);
      float* scr = ps_det_scratch(1, (size_t)3 * ntask + 8, st);
      PS_REQUIRE(scr, "rtm backward: deterministic mode has no scratch (allocation failed or stream capture)");
      int32_t* pk = reinterpret_cast<int32_t*>(scr);
      float* psc = scr + ntask;
      int32_t* pri = reinterpret_cast<int32_t*>(scr + 2 * (size_t)ntask);
      hipLaunchKernelGGL(rtm_pv_keys_kernel, dim3(ps_cdiv(ntask, 256)), dim3(256), 0, st, k, pk, psc, pri, ntask);
      PS_LAUNCH_CHECK();
      TRY(launch_rows_scatter_det(pk, ntask, k.vec, d, d, k.g_word_emb, st, psc, pri));
    }
  }
  PsTemTensors T, TG;
  to_tem_tensors(*params, T);
  to_tem_tensors(G, TG);
  ColFoldList fold;
  fold.n = 0;
  // d query_emb (+, when the index is built here, the per-word counters right behind it, rtm_make_ws): one memset
  if (k.pvc && !fwd_index) {
    const int64_t zend = r.wcnt + (((int64_t)D.vocab_size + 1 + 3) & ~(int64_t)3);
    PS_CHECK_HIP(hipMemsetAsync(ws + r.dqe, 0, sizeof(float) * (size_t)(zend - r.dqe), st));
  }
  const int eb = rtm_slot_blocks(r);
  if (k.pvc || k.det) k.gs = ws + r.enc_base + w.dx;     // (det + pv encoder: the review rows' gradients are parked in place, scattered below)
  TRY(enc_layers_backward(E, T, TG, nullptr, ws + r.valid, ws + r.enc_base, w, st, &fold, nullptr, rtm_rows_listed(r, w)));
  if (D.review_encoder == PS_RENC_FS) {
    // through the review projection: d pre = dx * tanh', bias gradient, weight gradient, d raw = d pre . f_W
    PS_REQUIRE(G.rev_fs_w && G.rev_fs_b && params->rev_fs_w, "rtm backward: null review-encoder f_W gradient");
    const int NR = r.Bseq * D.R;
    k.dx = ws + r.enc_base + w.dx;
    hipLaunchKernelGGL(rtm_fs_bwd_kernel, dim3(eb), dim3(256), (size_t)d * sizeof(float), st, k, ws + r.dpre, G.rev_fs_b);
    PS_LAUNCH_CHECK();
    if (k.det) {
      float* part = ps_det_scratch(1, (size_t)256 * d, st);
      PS_REQUIRE(part, "rtm backward: deterministic mode has no scratch (allocation failed or stream capture)");
      hipLaunchKernelGGL(rtm_colsum_det_kernel, dim3(256), dim3(256), 0, st, ws + r.dpre, NR, d, part);
      PS_LAUNCH_CHECK();
      hipLaunchKernelGGL(rtm_colsum_det_fold_kernel, dim3(1), dim3(256), 0, st, part, 256, d, G.rev_fs_b);
      PS_LAUNCH_CHECK();
    }
    GemmProblem wg[1] = {gp_wgrad(ws + r.dpre, d, ws + r.raw, d, G.rev_fs_w, d, d, NR)};
    TRY(side_wgrads(wg, 1, st));
    GemmProblem px = gp(ws + r.dpre, d, 0, params->rev_fs_w, d, 1, ws + r.dmean, d, NR, d, d);
    TRY(run1(px, st));
    k.dmean = ws + r.dmean;
  }
  {
    int npw, nnw;
    const int nwg = ps_cdiv(rtm_eb_waves(B, D.K, D.R, &npw, &nnw), 4);
    const FDiv fR = make_fdiv(D.R), fK = make_fdiv(D.K > 0 ? D.K : 1);
    float* sp = ws + r.segpart;
    if (k.det && (k.g_user_emb || k.g_item_emb)) {
      // user / item embedding rows, deterministic mode: a sole-owner scatter of d x by position — BEFORE the kernel below, which
      // rewrites the review positions' rows of d x in place (pvc: RtmK::gs)
      const int npos = r.Bseq * r.S;
      const float* dxp = ws + r.enc_base + w.dx;
      int32_t* keys = reinterpret_cast<int32_t*>(ps_det_scratch(1, (size_t)2 * npos + 8, st));
      PS_REQUIRE(keys, "rtm backward: deterministic mode has no scratch (allocation failed or stream capture)");
      int32_t* uk = k.g_user_emb ? keys : nullptr;
      int32_t* ik = k.g_item_emb ? keys + npos : nullptr;
      hipLaunchKernelGGL(rtm_ui_keys_kernel, dim3(ps_cdiv(npos, 256)), dim3(256), 0, st, k, uk, ik, (int32_t*)nullptr, npos);
      PS_LAUNCH_CHECK();
      if (uk) TRY(launch_rows_scatter_det(uk, npos, dxp, d, d, k.g_user_emb, st));
      if (ik) TRY(launch_rows_scatter_det(ik, npos, dxp, d, d, k.g_item_emb, st));
    }
    const bool plain = k.pvc && !k.dmean && !k.g_user_emb && !k.g_item_emb && !k.train_pv;
#define EB_LAUNCH(NK_)                                                                                             \
  do {                                                                                                             \
    if (plain) hipLaunchKernelGGL((rtm_embed_bwd_kernel<NK_, 1>), dim3(nwg), dim3(256), 0, st, k, sp, npw, nnw, fR, fK); \
    else hipLaunchKernelGGL((rtm_embed_bwd_kernel<NK_, 0>), dim3(nwg), dim3(256), 0, st, k, sp, npw, nnw, fR, fK);       \
  } while (0)
    if (d <= 64) EB_LAUNCH(1);
    else if (d <= 128) EB_LAUNCH(2);
    else if (d <= 256) EB_LAUNCH(4);
    else EB_LAUNCH(8);
#undef EB_LAUNCH
    PS_LAUNCH_CHECK();
    if (k.det && !k.pvc) {                           // pv encoder: the parked review-row gradients -> review_emb, by review id
      const int npos = r.Bseq * r.S;
      int32_t* rk = reinterpret_cast<int32_t*>(ps_det_scratch(1, (size_t)npos + 8, st));
      PS_REQUIRE(rk, "rtm backward: deterministic mode has no scratch (allocation failed or stream capture)");
      hipLaunchKernelGGL(rtm_ui_keys_kernel, dim3(ps_cdiv(npos, 256)), dim3(256), 0, st, k, (int32_t*)nullptr, (int32_t*)nullptr, rk, npos);
      PS_LAUNCH_CHECK();
      TRY(launch_rows_scatter_det(rk, npos, k.gs, d, d, k.g_table, st));
    }
    if (k.det) {
      k.dx = ws + r.enc_base + w.dx;
      hipLaunchKernelGGL(rtm_dqe_det_kernel, dim3(B), dim3(256), 0, st, k);
      PS_LAUNCH_CHECK();
    }
    if (D.use_seg_emb) {
      PS_REQUIRE(fold.n < PS_MAX_COLFOLD, "rtm backward: too many parked column sums");
      ColFold& f = fold.e[fold.n++];
      f.partial = ws + r.segpart; f.nblk = nwg; f.d = d;
      f.dst[0] = G.seg_emb; f.dst[1] = G.seg_emb + d; f.dst[2] = G.seg_emb + 2 * (size_t)d;
    }
  }
  if (k.pvc) {
    if (!fwd_index) {
      const int V = (int)D.vocab_size;
      TRY(rtm_build_index(k, r, V, true, st));
    }
    // The word-gradient reduce needs the index (side stream, in order there) and the slot gradients the kernel above left
    // (main stream): with forks free it runs ON the side stream behind a fork — carried by the query scatter below — beside
    // that scatter, and the join that used to precede it (5 us on the main stream with nothing to wait for) is gone.
    // PS_RTM_WR_SIDE=0: joined and launched on the main stream.
    static const bool wr_side_on = ps_diag_int("PS_RTM_WR_SIDE", 1) != 0;
    hipStream_t wst = st;
    if (fwd_index) {
      hipStream_t ss = side_stream_or(st);
      if (wr_side_on && ss != st) { TRY(side_fork(st)); wst = ss; }
      else TRY(side_join(st));                      // the index (and the weight gradients queued behind it) are through
    }
    const int64_t max_occ = (int64_t)r.Bseq * D.R * D.WL;
    static const int wr_cap = ps_diag_int("PS_RTM_WR_WGS", 4096);
    int64_t wr = (max_occ + 255) / 256;
    if (wr > wr_cap) wr = wr_cap;
    // (measured and dropped: the columns split over the XCDs — workgroup i takes d/8 columns of every entry, so that an XCD's
    // L2 holds 1/8 of each gathered row and serves the 56 re-reads itself: 206 us against 66, each 4-lane entry stream keeps
    // too few bytes in flight; this form reads 594 MB from the Infinity Cache at 8.9 TB/s)
    if (k.det) {
      const int V = (int)D.vocab_size;
      int* heavy = reinterpret_cast<int*>(ps_det_scratch(1, (size_t)V + 8, st));
      PS_REQUIRE(heavy, "rtm backward: deterministic mode has no scratch (allocation failed or stream capture)");
      int* nheavy = heavy + V;
      PS_CHECK_HIP(hipMemsetAsync(nheavy, 0, sizeof(int), wst));
      const int lw = ps_cdiv(V, 4) < 2048 ? ps_cdiv(V, 4) : 2048;
#define WR_DET_LAUNCH(NK)                                                                                              \
  do {                                                                                                                \
    hipLaunchKernelGGL(rtm_wreduce_det_kernel<NK>, dim3(lw), dim3(256), 0, wst, k, heavy, nheavy);                     \
    hipLaunchKernelGGL(rtm_wreduce_heavy_det_kernel<NK>, dim3(512), dim3(1024), 0, wst, k, heavy, nheavy);             \
  } while (0)
      if (d <= 128) WR_DET_LAUNCH(2); else if (d <= 256) WR_DET_LAUNCH(4); else WR_DET_LAUNCH(8);
#undef WR_DET_LAUNCH
    } else {
      // (round 5, measured and dropped: a wave OWNING words — one atomic row per word — that walks their occurrences in passes over the
      // slot space for L2 residency, and 32 lanes x 16 bytes per slot row: 57-100 and 79 us against 55.5, profiles/r05_c4_wreduce_notes.md)
      hipLaunchKernelGGL(rtm_wreduce_kernel, dim3((unsigned)wr), dim3(256), 0, wst, k);
    }
    PS_LAUNCH_CHECK();
  }
  // query encoder backward (shared kernels) + scatter to the query word rows
  PsTemDesc dq;
  memset(&dq, 0, sizeof(dq));
  dq.training = k.training; dq.dropout = D.dropout; dq.seed = D.seed; dq.step = D.step;
  EmbedBwdArgs e;
  memset(&e, 0, sizeof(e));
  e.B = B; e.Q = D.Q; e.L = 0; e.S = 1; e.d = d; e.P = 1; e.V = D.vocab_size; e.tem = 0;
  e.qw = batch->query_word_idxs; e.drop_fs = make_drop(dq, PS_SITE_FS); e.g_word_emb = G.word_emb;
  if (D.query_encoder == PS_QENC_FS) {
    PS_REQUIRE(G.fs_w && G.fs_b, "rtm backward: null FS gradients");
    if (ps_fusion_enabled() && d <= 128) {   // whole FS backward inside the scatter launch (EmbedBwdArgs::fsb_*)
      e.fw_x = ws + r.qmean; e.g_fs_w = G.fs_w;
      e.fsb_dqe = ws + r.dqe; e.fsb_lddqe = d; e.fsb_qe = ws + r.query_emb; e.fsb_w = params->fs_w; e.g_fs_b = G.fs_b;
      e.det_dm = ws + r.dqmean;                   // (deterministic mode only: launch_embed_scatter)
    } else {
      TRY(launch_tanh_bwd(ws + r.dqe, d, ws + r.query_emb, ws + r.dqpre, G.fs_b, B, d, st));
      GemmProblem p = gp(ws + r.dqpre, d, 0, params->fs_w, d, 1, ws + r.dqmean, d, B, d, d);
      TRY(run1(p, st));
      GemmProblem wg[1] = {gp_wgrad(ws + r.dqpre, d, ws + r.qmean, d, G.fs_w, d, d, B)};
      TRY(side_wgrads(wg, 1, st));
      e.dqmean_d = ws + r.dqmean;
    }
  } else {
    e.dqmean_d = ws + r.dqe;
  }
  e.fold = fold;
  TRY(launch_embed_scatter(e, st));
  TRY(side_join(st));
  return PS_OK;
}
