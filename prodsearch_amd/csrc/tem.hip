// tem.hip — host orchestration + C ABI of the TEM / QEM ranking-loss step (gfx950).
//
// Forward  = ItemTransformerRanker.forward_dotproduct (item_transformer.py:440-520)
//            / forward_attn with model_name == 'QEM' (:361-438)
// Backward = autograd of the same (trainer.py:77)
// Score    = test_dotproduct (:111-146) / test_attn QEM (:148-195)
//
// Structure of the encoder ("replicas"): the reference encodes the SAME (query, history)
// sequence K+1 times (once for the positive, K expanded copies for the negatives).  The copies
// only differ through dropout, which first acts on the softmax output of layer 0, so:
//   * K/V/Q projections and softmax of layer 0 run once per batch row          (n_in  = B)
//   * everything after the first dropout runs per replica                       (n_out = B*R)
//     with R = K+1 when dropout is drawn (training && dropout > 0) and R = 1 otherwise,
//     where all replicas are bit-identical and one is computed;
//   * the LAST layer only produces the one output position that is consumed
//     (x[:, 0] or x[:, -1], item_transformer.py:482-492), so its query/attention/FFN rows
//     are n_out x 1 instead of n_out x S.
#include "encoder.h"
#include "graph.h"
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

// ------------------------------------------------------------------ error text
static thread_local char g_err[512] = "";
void ps_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* ps_last_error(void) { return g_err; }
extern "C" const char* ps_version(void) { return "prodsearch_hip 0.1 (gfx950, fp32 MFMA)"; }
// what the step computes in (bench.py's `dtype`): everything is fp32 in and out; products run on the fp32 MFMA or, where the
// bf16x3 form is enabled (ps_gemm_x3_config, the fused per-replica kernels), as exact three-way bf16 splits of both fp32
// operands — six bf16 MFMAs per product step, fp32 accumulation, the fp32 MFMA's accuracy
bool gemm_x3_on();
extern "C" const char* ps_arith_info(void) {
  // (the SPLIT is exact — hi + mid + lo carry all 24 mantissa bits; the PRODUCT keeps six of the nine cross terms and drops those
  // below 2^-24 of the leading one: fp32-GRADE, 1.1e-7 of sum |a b| against fp64, the fp32 MFMA's own 1.13e-7 — not "exact")
  return gemm_x3_on() ? "f32 (fp32 MFMA and VALU; wide products, the fused per-replica kernels and grouped weight gradients as fp32-grade "
                        "bf16x3 products: exact 3-way bf16 split of both fp32 operands, 6 of the 9 cross products as bf16 MFMAs per step, "
                        "fp32 accumulation)"
                      : "f32 (fp32 MFMA and VALU)";
}

// ------------------------------------------------------------------ kernel timer (common.h)
#include <vector>
static struct KTimer {
  char tag[32];
  bool armed, open;
  std::vector<hipEvent_t> e0, e1;
  int n, cap;
} g_kt = {"", false, false, {}, {}, 0, 0};
const char* ps_ktimer_tag() { return g_kt.armed ? g_kt.tag : nullptr; }
void ps_ktimer_scope(bool open) { g_kt.open = open && g_kt.armed; }
bool ps_ktimer_take(hipEvent_t* e0, hipEvent_t* e1) {
  if (!g_kt.open) return false;
  g_kt.open = false;                                              // one launch per scope
  if (g_kt.n >= g_kt.cap) return false;
  *e0 = g_kt.e0[g_kt.n]; *e1 = g_kt.e1[g_kt.n]; ++g_kt.n;
  return true;
}
extern "C" int ps_ktimer_arm(const char* tag, int32_t max_samples) {
  g_kt.armed = false; g_kt.open = false;
  g_kt.n = 0;
  if (!tag || !*tag || max_samples <= 0) return PS_OK;            // disarm
  PS_REQUIRE(strlen(tag) < sizeof(g_kt.tag), "ktimer: tag too long");
  while ((int)g_kt.e0.size() < max_samples) {
    hipEvent_t a, b;
    PS_CHECK_HIP(hipEventCreate(&a));
    PS_CHECK_HIP(hipEventCreate(&b));
    g_kt.e0.push_back(a); g_kt.e1.push_back(b);
  }
  g_kt.cap = max_samples;
  strcpy(g_kt.tag, tag);
  g_kt.armed = true;
  return PS_OK;
}
// average / min duration (us) of the launches bracketed since ps_ktimer_arm; synchronises the device; disarms
extern "C" int ps_ktimer_read(double* avg_us, double* min_us, int32_t* count) {
  PS_REQUIRE(avg_us && count, "ktimer: null argument");
  g_kt.armed = false; g_kt.open = false;
  PS_CHECK_HIP(hipDeviceSynchronize());
  double sum = 0, mn = 1e30;
  for (int i = 0; i < g_kt.n; ++i) {
    float ms = 0.f;
    PS_CHECK_HIP(hipEventElapsedTime(&ms, g_kt.e0[i], g_kt.e1[i]));
    sum += ms * 1e3; mn = ms * 1e3 < mn ? ms * 1e3 : mn;
  }
  *count = g_kt.n;
  *avg_us = g_kt.n ? sum / g_kt.n : 0.0;
  if (min_us) *min_us = g_kt.n ? mn : 0.0;
  g_kt.n = 0;
  return PS_OK;
}

static int check_desc(const PsTemDesc& D) {
  PS_REQUIRE(D.B > 0 && D.K >= 0 && D.Q > 0 && D.W >= 0 && D.d > 0, "desc: bad sizes B=%d K=%d Q=%d W=%d d=%d",
             D.B, D.K, D.Q, D.W, D.d);
  PS_REQUIRE(D.d % 32 == 0 && D.d <= 512, "desc: embedding_size %d must be a multiple of 32 and <= 512", D.d);
  PS_REQUIRE(D.model == PS_MODEL_TEM || D.model == PS_MODEL_QEM, "desc: model %d", D.model);
  if (D.model == PS_MODEL_TEM) {
    PS_REQUIRE(D.L >= 0 && D.L + 1 <= 64, "desc: history length %d (S <= 64)", D.L);
    PS_REQUIRE(D.n_layers >= 0 && D.n_layers <= PS_MAX_LAYERS, "desc: inter_layers %d", D.n_layers);
    if (D.n_layers > 0) {
      PS_REQUIRE(D.H > 0 && D.d % D.H == 0 && D.d / D.H <= 64, "desc: heads %d for d %d", D.H, D.d);
      PS_REQUIRE(D.F > 0 && D.F % 4 == 0, "desc: ff_size %d", D.F);
    }
  }
  PS_REQUIRE(D.dropout >= 0.f && D.dropout < 1.f, "desc: dropout %f", D.dropout);
  PS_REQUIRE(D.product_size > 0 && D.vocab_size > 1, "desc: table sizes");
  return PS_OK;
}

static inline int64_t take(int64_t& cur, int64_t n) {
  int64_t o = cur;
  cur += (n + 3) & ~(int64_t)3;     // keep every buffer 16-byte aligned
  return o;
}

int make_ws(const PsTemDesc& D, Ws& w) {
  int rc = check_desc(D);
  if (rc) return rc;
  memset(&w, 0, sizeof(w));
  const bool tem = D.model == PS_MODEL_TEM;
  const bool drop = D.training && D.dropout > 0.f;
  const int B = D.B, d = D.d, S = tem ? D.L + 1 : 1, NL = tem ? D.n_layers : 0;
  w.S = S;
  w.R = (tem && drop && NL > 0 && D.C == 0) ? D.K + 1 : 1;
  w.qpos = D.use_item_pos ? S - 1 : 0;
  int64_t cur = 0;
  w.qmean = take(cur, (int64_t)B * d);
  w.query_emb = take(cur, (int64_t)B * d);
  w.x = tem ? take(cur, (int64_t)B * S * d) : 0;
  int64_t maxM2 = 0, maxNS = 0;
  for (int i = 0; i < NL; ++i) {
    LayerWs& l = w.layer[i];
    l.n_in = i == 0 ? B : B * w.R;
    l.n_out = B * w.R;
    l.fan = l.n_out / l.n_in;
    l.Sq = i == NL - 1 ? 1 : S;
    l.M2 = l.n_out * l.Sq;
    const int64_t ns = (int64_t)l.n_in * S;
    l.xn = i == 0 ? w.x : take(cur, ns * d);
    l.pre_stats = i == 0 ? 0 : take(cur, ns * 2);
    l.kp = take(cur, ns * d);
    l.vp = take(cur, ns * d);
    l.qp = take(cur, (int64_t)l.n_in * l.Sq * d);
    l.attn = take(cur, (int64_t)l.n_in * D.H * l.Sq * S);
    l.amask = take(cur, (int64_t)l.n_in * l.fan * D.H);
    l.ctx = take(cur, (int64_t)l.M2 * d);
    l.y1 = take(cur, (int64_t)l.M2 * d);
    l.ff_stats = take(cur, (int64_t)l.M2 * 2);
    l.ln1 = take(cur, (int64_t)l.M2 * d);
    l.a1 = take(cur, (int64_t)l.M2 * D.F);
    l.h1 = take(cur, (int64_t)l.M2 * D.F);
    l.y2 = take(cur, (int64_t)l.M2 * d);
    maxM2 = l.M2 > maxM2 ? l.M2 : maxM2;
    maxNS = ns > maxNS ? ns : maxNS;
  }
  w.Mf = B * w.R;
  w.fin_stats = take(cur, (int64_t)w.Mf * 2);
  w.enc = tem ? take(cur, (int64_t)w.Mf * d) : w.query_emb;
  const int C = D.C > 0 ? D.C : D.K + 1;
  w.item_scores = take(cur, (int64_t)B * C);
  w.word_scores = take(cur, (int64_t)B * (D.W > 0 ? D.W : 1) * (D.K + 1));
  w.loss_parts = take(cur, (int64_t)B * 2);
  w.item_terms = take(cur, (int64_t)B * (D.K + 1));
  w.word_terms = take(cur, (int64_t)B * (D.W > 0 ? D.W : 1) * (D.K + 1));
  w.loss_blk = take(cur, 2 * ((int64_t)B * (D.K + 1) * (1 + D.W) / 4 + 2));     // >= 2 floats per score workgroup
  w.word_blk = take(cur, ps_cdiv((int64_t)B * (D.W > 0 ? D.W : 1) * (D.K + 1), PS_WORD_TASKS_PER_WG) + 4);
  w.item_blk = take(cur, ps_cdiv((int64_t)B * w.R, 32) + 4);
  w.ticket = take(cur, 20);      // 9 x 64-bit words (8 shards + top), 16-byte aligned
  w.wsplit = (tem && NL > 0 && d == 128 && mlp_x3_enabled(D.F)) ? take(cur, mlp_x3_floats(d, D.F)) : 0;
  // backward scratch (sized for the widest layer)
  w.denc = take(cur, (int64_t)w.Mf * d);
  if (tem) {
    const int64_t F = NL > 0 ? D.F : 0;
    w.dy2 = take(cur, (maxM2 > w.Mf ? maxM2 : w.Mf) * d);
    w.do2 = take(cur, maxM2 * d);
    w.da1 = take(cur, maxM2 * F);
    w.dln1 = take(cur, maxM2 * d);
    w.dy1 = take(cur, maxM2 * d);
    w.do_ = take(cur, maxM2 * d);
    w.dctx = take(cur, maxM2 * d);
    w.dq = take(cur, maxNS * d);
    w.dkv = take(cur, maxNS * 3 * d);
    w.dxn = take(cur, maxNS * d);
    w.dx = take(cur, (int64_t)B * S * d);
  }
  w.dqpre = take(cur, (int64_t)B * d);
  w.dqmean = take(cur, (int64_t)B * d);
  w.lnrows = (int)((maxM2 + 31) / 32 > 256 ? (maxM2 + 31) / 32 : 256);
  w.lnpart = take(cur, (int64_t)PS_MAX_COLFOLD * w.lnrows * 3 * d);
  w.stage = take(cur, 4 + 2 * ((int64_t)B * (D.Q + D.L + 1 + D.W + D.K + D.W * D.K) + 8));   // int64 = 2 floats
  w.gcpart = take(cur, (int64_t)4 * ((maxM2 + 31) / 32 + 1) * 3 * (tem && NL > 0 ? D.F : 0));   // >= mlp_bwd_b1_rows()
  w.abpart = take(cur, (int64_t)NL * (NL > 0 ? w.layer[NL - 1].n_in : 0) * 3 * d);
  w.vrows = tem ? take(cur, (int64_t)B * S + 4) : 0;
  w.vcount = tem ? take(cur, 4) : 0;
  w.total = cur;
  return PS_OK;
}

extern "C" int ps_tem_workspace_layout(const PsTemDesc* desc, PsTemWsLayout* out) {
  PS_REQUIRE(desc && out, "workspace_layout: null argument");
  Ws w;
  int rc = make_ws(*desc, w);
  if (rc) return rc;
  memset(out, 0, sizeof(*out));
  out->total_floats = w.total;
  out->R = w.R; out->S = w.S;
  out->qmean = w.qmean; out->query_emb = w.query_emb; out->x = w.x;
  const int NL = desc->model == PS_MODEL_TEM ? desc->n_layers : 0;
  if (NL > 0) {
    const LayerWs& l = w.layer[NL - 1];
    out->kp = l.kp; out->vp = l.vp; out->qp = l.qp; out->attn = l.attn; out->ctx = l.ctx;
    out->y1 = l.y1; out->ln1 = l.ln1; out->a1 = l.a1; out->h1 = l.h1; out->y2 = l.y2;
  }
  out->enc = w.enc;
  out->item_scores = w.item_scores; out->word_scores = w.word_scores; out->loss_parts = w.loss_parts;
  out->denc = w.denc; out->dx = w.dx;
  return PS_OK;
}

WSplit make_wsplit(const PsTemDesc& D, const PsTemTensors& P, float* ws, const Ws& w) {
  WSplit s;
  memset(&s, 0, sizeof(s));
  const int NL = D.model == PS_MODEL_TEM ? D.n_layers : 0;
  if (NL < 1 || !w.wsplit || D.d != 128 || !mlp_x3_enabled(D.F) || w.layer[NL - 1].Sq != 1) return s;
  const PsLayerTensors& L = P.layer[NL - 1];
  if (!L.wo || !L.w1 || !L.w2) return s;
  const int d = D.d, F = D.F;
  s.w[0] = L.wo; s.rows[0] = d; s.cols[0] = d;
  s.w[1] = L.w1; s.rows[1] = F; s.cols[1] = d;
  s.w[2] = L.w2; s.rows[2] = d; s.cols[2] = F;
  uint16_t* base = reinterpret_cast<uint16_t*>(ws + w.wsplit);
  const size_t n_wo = (size_t)3 * d * d, n_ff = (size_t)3 * 2 * d * F;
  s.fwd_wo = base; s.fwd_ff = base + n_wo; s.bwd_ff = base + n_wo + n_ff; s.bwd_wo = base + n_wo + 2 * n_ff;
  s.on = 1;
  if (NL == 1 && L.wk && L.wv) {      // one layer: its K / V projections can take the fused projection + attention forward
    s.wkv[0] = L.wk; s.wkv[1] = L.wv;
    s.fwd_kv = base + 2 * (n_wo + n_ff);
    s.bwd_kv = s.fwd_kv + (size_t)3 * 2 * d * d;
  }
  return s;
}

// The encoder weights a WPlaneScope (common.h) should hold for this call: the last layer's six linears when its products are
// big enough for the pre-split-weight kernel (>= 4096 replica rows) and are not taken by the fused d = 128 kernels.
int wplane_list(const PsTemDesc& D, const PsTemTensors& P, const Ws& w, const float** ws_, int* rows, int* cols) {
  const int NL = D.model == PS_MODEL_TEM ? D.n_layers : 0;
  if (NL < 1 || D.d < 256 || D.d % 32 || D.F % 32) return 0;
  int n = 0;
  for (int i = NL - 1; i >= 0 && n + 6 <= PS_WPLANES_MAX; --i) {
    const LayerWs& l = w.layer[i];
    if ((int64_t)l.M2 < 4096 && (int64_t)l.n_in * w.S < 4096) continue;
    const PsLayerTensors& L = P.layer[i];
    const float* ptr[6] = {L.wk, L.wv, L.wq, L.wo, L.w1, L.w2};
    const int r[6] = {D.d, D.d, D.d, D.d, D.F, D.d}, c[6] = {D.d, D.d, D.d, D.d, D.d, D.F};
    for (int k = 0; k < 6; ++k) { ws_[n] = ptr[k]; rows[n] = r[k]; cols[n] = c[k]; ++n; }
  }
  return n;
}

// ----------------------------------------------------------------- GEMM helpers
GemmProblem gp(const float* A, int lda, int ta, const float* Bm, int ldb, int tb, float* C, int ldc, int M,
                      int N, int K) {
  GemmProblem p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.lda = lda; p.ta = ta;
  p.Bseg[0] = Bm; p.kseg = K; p.ldb = ldb; p.tb = tb;
  p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  p.alpha = 1.f; p.ksplit = 1;
  return p;
}
int run1(const GemmProblem& p, hipStream_t st) {
  GemmGroup g;
  memset(&g, 0, sizeof(g));
  g.n = 1; g.p[0] = p;
  return ps_launch_gemm(g, st);
}
// weight gradient  dW[N_out, K_in] += dY[rows, N_out]^T . X[rows, K_in]   (atomic, split over rows)
GemmProblem gp_wgrad(const float* dY, int lddy, const float* X, int ldx, float* dW, int n_out, int k_in,
                            int rows) {
  GemmProblem p = gp(dY, lddy, 1, X, ldx, 1, dW, k_in, n_out, k_in, rows);
  p.accumulate = 2;
  return p;
}
static int pick_ksplit(int tiles, int rows) {   // ~2 workgroups per CU, but at least ~512 reduction rows per split
  // Every split adds a 64x64 tile of fp32 atomics onto the same weight-gradient addresses.  Measured: C2 (8,064 rows; step
  // time by blocks per launch: 512 0.380 ms, 256 0.375, 224 0.373, 192 0.372-0.377, 128 0.392) wants ~15 splits of ~540
  // rows; the review transformer (78k rows, 4 tiles) wants its 128 splits of ~610 rows (1.146 ms vs 1.193 with 56).
  static const int target = ps_env_int("PS_WGRAD_BLOCKS", 512);   // tuning experiments
  static const int min_rows = ps_env_int("PS_WGRAD_ROWS", 512);
  const int nt = tiles > 0 ? tiles : 1;
  const int want = target / nt;
  int ks = (rows + min_rows - 1) / min_rows;                 // >= ~512 rows per split ...
  const int fill = (128 + nt - 1) / nt, cap128 = (rows + 127) / 128;
  if (ks < fill) ks = fill < cap128 ? fill : cap128;         // ... unless that leaves fewer than ~128 workgroups (Wo: 4 tiles)
  if (ks > want) ks = want;
  return ks < 1 ? 1 : ks;
}
// ---- deterministic mode (PS_DETERMINISTIC=1 or ps_set_deterministic): see common.h / DESIGN.md 5e
static int& det_slot() {
  static int v = ps_env_int("PS_DETERMINISTIC", 0);
  return v;
}
bool ps_deterministic() { return det_slot() != 0; }
extern "C" int ps_set_deterministic(int on) {
  const int old = det_slot();
  if (on >= 0) det_slot() = on ? 1 : 0;          // negative: query only
  return old;
}
// scratch of the ordered split reduction: per device, grow-only, allocated outside any stream capture
static float* det_scratch(size_t floats, hipStream_t st) { return ps_det_scratch(0, floats, st); }
// dW[i] += sum_s part[s][i], s ascending: the second pass of a deterministic split reduction
__global__ __launch_bounds__(256) void wgrad_sum_kernel(const float* part, int ks, int64_t n, float* dW) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s0 = 0; s0 < ks; s0 += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(part + (size_t)(s0 + u < ks ? s0 + u : s0) * n + i);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (s0 + u < ks) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  float4 d = *reinterpret_cast<float4*>(dW + i);
  d.x += acc.x; d.y += acc.y; d.z += acc.z; d.w += acc.w;
  *reinterpret_cast<float4*>(dW + i) = d;
}
static int run_wgrads_det(GemmGroup& g, hipStream_t st) {
  // every member writes ks partial matrices [M][N] (plain stores), then one ordered sum per member
  size_t total = 0;
  for (int i = 0; i < g.n; ++i) {
    PS_REQUIRE(g.p[i].ldc == g.p[i].N && (g.p[i].M * (int64_t)g.p[i].N) % 4 == 0 && !g.p[i].bias,
               "deterministic weight gradient: contiguous, bias-free output expected");
    total += (size_t)g.p[i].ksplit * g.p[i].M * g.p[i].N;
  }
  float* sc = det_scratch(total, st);
  PS_REQUIRE(sc, "deterministic mode: no scratch for the split reduction (allocation failed or stream capture)");
  PS_CHECK_HIP(hipMemsetAsync(sc, 0, total * sizeof(float), st));     // splits without slabs (row lists) leave zeros
  float* dW[4]; float* part[4];
  size_t off = 0;
  for (int i = 0; i < g.n; ++i) {
    GemmProblem& p = g.p[i];
    dW[i] = p.C; part[i] = sc + off;
    p.C = part[i]; p.accumulate = 0; p.split_stride = (int64_t)p.M * p.N;
    off += (size_t)p.ksplit * p.M * p.N;
  }
  TRY(ps_launch_gemm(g, st));
  for (int i = 0; i < g.n; ++i) {
    const int64_t n = (int64_t)g.p[i].M * g.p[i].N;
    hipLaunchKernelGGL(wgrad_sum_kernel, dim3((unsigned)ps_cdiv(n / 4, 256)), dim3(256), 0, st, part[i], g.p[i].ksplit, n, dW[i]);
    PS_LAUNCH_CHECK();
  }
  return PS_OK;
}

// (measured and dropped, round 2: a two-pass split reduction — every split stores its partial tile in scratch, takes a ticket,
// the last arriver of a tile adds the partials up in split order — deterministic and free of fp32 atomics, but 122 us against
// 44 for the grouped launch at C2 and 0.395 against 0.292 ms per step: the device-scope release each of the 600 workgroups
// needs before its ticket writes back its XCD's L2, MI300-class L2s not being coherent with one another)
static int run_wgrads(GemmProblem* ps, int n, hipStream_t st) {
  GemmGroup g;
  memset(&g, 0, sizeof(g));
  g.n = n;
  bool same = true, plain = true;
  for (int i = 0; i < n; ++i) {
    same = same && ps[i].M == ps[0].M && ps[i].N == ps[0].N && ps[i].K == ps[0].K;
    plain = plain && !ps[i].ridx;
  }
  if (n > 1 && n <= 3 && !same && plain) {
    // different shapes in one launch: the flat form (GemmGroup::flat) — every problem keeps the split count it would
    // take alone, no idle workgroups for the tiles the smaller members do not have
    for (int i = 0; i < n; ++i) {
      g.p[i] = ps[i];
      g.p[i].ksplit = pick_ksplit(ps_cdiv(ps[i].M, 64) * ps_cdiv(ps[i].N, 64), ps[i].K);
    }
    g.flat = 1;
    if (ps_deterministic()) return run_wgrads_det(g, st);
    KTimeScope kt("wgrad_group", st);
    return ps_launch_gemm(g, st);
  }
  int tiles = 0, rows = 0;
  for (int i = 0; i < n; ++i) {
    tiles += ps_cdiv(ps[i].M, 64) * ps_cdiv(ps[i].N, 64);
    rows = ps[i].K > rows ? ps[i].K : rows;
  }
  int ks = pick_ksplit(tiles, rows);
  // Big weight gradients (the d = 256 step's W2 / W1: 16 tiles of 128 x 128 over 21,504 reduction rows) take the direct-to-LDS
  // bf16x3 kernel with 128x128 tiles and ~512 workgroups: 104-109 us against 129-131 for the 64x64 tiles at ANY split count
  // (MI355X, profiles/r04_gemm_wgrad_ksplit.txt) — half the operand bytes per flop through LDS, a quarter of the atomic tiles'
  // row segments.  Few tiles (Wo: 4) or few rows (C2: 8,064) cannot fill the chip that way and keep the 64x64 form.
  {
    int t128 = 0;
    for (int i = 0; i < n; ++i) t128 += ps_cdiv(ps[i].M, 128) * ps_cdiv(ps[i].N, 128);
    const int by_rows = ps_cdiv(rows, 512), by_fill = 512 / (t128 > 0 ? t128 : 1);
    const int ks3 = by_rows < by_fill ? by_rows : by_fill;
    if (plain && gemm_x3_on() && !ps_deterministic() && t128 * ks3 >= 384 && rows % 32 == 0) { ks = ks3; g.prefer_x3d = 1; }
  }
  for (int i = 0; i < n; ++i)
    if (ps[i].ridx) {   // a row-list problem maps one split's reduction rows through LDS: at most PS_GEMM_KIDX_MAX of them
      const int need = ps_cdiv(ps_cdiv(ps[i].K, 32) * 32, PS_GEMM_KIDX_MAX - 32);
      if (ks < need) ks = need;
    }
  // a split count that is a multiple of 8 lets the launch place every split's tiles on one XCD (GemmGroup::split_xcd / flat_xcd)
  static const int ks_round8 = ps_diag_int("PS_KS_ROUND8", 1);
  if (ks_round8 && ks > 8 && ks % 8 != 0 && !ps_deterministic()) ks = (ks + 7) / 8 * 8;
  for (int i = 0; i < n; ++i) { g.p[i] = ps[i]; g.p[i].ksplit = ks; }
  if (ps_deterministic() && ks > 1) return run_wgrads_det(g, st);
  return ps_launch_gemm(g, st);
}

// ---- weight-gradient GEMMs run on a side stream: they are off the dX critical path (nothing in
// the backward consumes dW), so they overlap the latency-bound main chain.  Fork = event recorded on
// the main stream after the producer; join = the main stream waits for the side stream's last event.
#include <stdlib.h>
struct SideCtx {
  hipStream_t stream;
  hipEvent_t ev[8];
  hipEvent_t join;
  int next;
  bool used;
  uint32_t* flag;            // {fork, join} sequence words for stream write / wait-value crossings (null: event pairs)
  uint32_t fork_seq, join_seq;
  bool light;                // use them for the current backward (side_set_light)
  bool sig_pending;          // a fork whose value the next main-stream kernel stores (side_take_signal)
  uint32_t sig_val; hipStream_t sig_stream;
};
// One context per device (a process drives one GPU in production; tests and tools may touch several), created under a
// mutex.  A context serves ONE host thread at a time — the single-thread contract of the step (include/prodsearch_hip.h,
// "Threading"): the loader's prefetch thread never calls into this library's device side.
#include <mutex>
#define PS_MAX_DEVICES 16
static bool env_on(const char* name) { const char* e = getenv(name); return e && *e && atoi(e) != 0; }
// Anything that lets only ONE kernel run on the device at a time deadlocks a stream wait-value (the runtime implements it
// as a one-thread kernel spinning on the word: the producer never gets to run).  Known serialisers: counter-collecting
// profilers (rocprofv3 / rocprof --pmc), AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING / CUDA_LAUNCH_BLOCKING, debuggers
// (ROCgdb sets HSA_ENABLE_DEBUG), and any tool library preloaded into the process.
static bool dispatch_may_be_serialised() {
  static const char* const truthy[] = {"AMD_SERIALIZE_KERNEL", "AMD_SERIALIZE_COPY", "HIP_LAUNCH_BLOCKING",
                                       "CUDA_LAUNCH_BLOCKING", "HSA_ENABLE_DEBUG"};
  for (const char* n : truthy) if (env_on(n)) return true;
  const char* mq = getenv("GPU_MAX_HW_QUEUES");
  if (mq && *mq && atoi(mq) == 1) return true;          // one hardware queue: both streams share it in order
  // a profiling / tracing tool is attached: value waits only in the one mode known to keep dispatches concurrent — a
  // rocprofv3 kernel trace with no counter collection, PC sampling or thread trace (so that the traced timeline is the
  // production one); every other tool, known or not, gets event pairs
  const char* pre = getenv("LD_PRELOAD");
  const bool tool = getenv("ROCP_TOOL_LIBRARIES") || getenv("HSA_TOOLS_LIB") || getenv("ROCP_METRICS") ||
                    getenv("ROCPROFILER_METRICS_PATH") ||
                    (pre && (strstr(pre, "rocprof") || strstr(pre, "roctracer") || strstr(pre, "tool")));
  if (!tool) return false;
  static const char* const counters[] = {"ROCPROF_COUNTER_COLLECTION", "ROCPROF_COUNTERS", "ROCPROF_COUNTER_GROUPS",
                                         "ROCPROF_EXTRA_COUNTERS_CONTENTS", "ROCPROFILER_PC_SAMPLING_BETA_ENABLED",
                                         "ROCPROF_PC_SAMPLING_METHOD", "ROCPROF_ADVANCED_THREAD_TRACE",
                                         "ROCPROF_ATT_LIBRARY_PATH", "ROCPROF_ATTACH_PID"};
  for (const char* n : counters) if (getenv(n)) return true;
  return !(getenv("ROCP_TOOL_LIBRARIES") && getenv("ROCPROF_KERNEL_TRACE"));
}
// Start-up self-test of the value crossings (once per device): park the side stream on a probe word, release it by a
// write-value operation on ANOTHER stream, and poll — with a host-side timeout — for the side stream to drain.  Where dispatches
// are serialised by something the environment list above does not know (a tool, a driver mode, one shared hardware queue) the
// write never executes while the wait spins: the probe is then released from the host (a copy, not a kernel) and the value
// crossings are switched off for this process (event pairs cannot hang).  If even the host cannot release it the side stream
// is unusable: the step then runs on one stream and the reason is printed once.
#include <chrono>
#include <thread>
static bool stream_drains_within(hipStream_t st, int ms) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) return true;
    if (q != hipErrorNotReady) { (void)hipGetLastError(); return false; }
    if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > ms) return false;
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}
// 1 = value waits make progress, 0 = they do not (released from the host: use events), -1 = the side stream is stuck
static int side_value_selftest(SideCtx& ctx) {
  // the probe word lives in host-coherent memory: if the release by the other stream never runs, a plain host store frees the
  // spinning wait without needing the device to execute anything
  uint32_t* probe = nullptr;
  if (hipHostMalloc((void**)&probe, sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess || !probe) {
    (void)hipGetLastError();
    return 0;
  }
  *probe = 0u;
  hipStream_t other = nullptr;
  if (hipStreamCreateWithFlags(&other, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(probe); return 0; }
  int verdict = 0;
  if (hipStreamWaitValue32(ctx.stream, probe, 1u, hipStreamWaitValueGte, 0xffffffffu) == hipSuccess &&
      hipStreamWriteValue32(other, probe, 1u, 0) == hipSuccess) {
    const char* force = getenv("PS_SIDE_SELFTEST_FAIL");          // tests: pretend the write never ran
    if (!(force && atoi(force) != 0) && stream_drains_within(ctx.stream, 500)) verdict = 1;
    else {
      __atomic_store_n(probe, 1u, __ATOMIC_SEQ_CST);
      verdict = stream_drains_within(ctx.stream, 5000) ? 0 : -1;
      fprintf(stderr, verdict == 0 ? "prodsearch_hip: stream value waits do not make progress beside their producer in this environment; "
                                     "the side stream crosses with event pairs (PS_SIDE_EVENTS=1 skips this probe)\n"
                                   : "prodsearch_hip: the side stream cannot be released (dispatches look serialised and blocked); "
                                     "the step runs on ONE stream\n");
    }
  }
  (void)hipGetLastError();
  if (verdict >= 0) {                       // (a stuck stream is left alone: destroying it would wait for it)
    (void)hipStreamSynchronize(other);
    (void)hipStreamDestroy(other);
    (void)hipHostFree(probe);
  }
  (void)hipGetLastError();
  return verdict;
}
extern "C" int ps_side_values_in_use(void);
static SideCtx* side_ctx() {
  if (ps_deterministic()) return nullptr;              // one stream: the order in which kernels add into a table is the launch order
  static SideCtx ctxs[PS_MAX_DEVICES];
  static int states[PS_MAX_DEVICES];           // 0 = uninitialised, 1 = ready, -1 = disabled
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PS_MAX_DEVICES) { (void)hipGetLastError(); return nullptr; }
  std::lock_guard<std::mutex> lock(mu);
  SideCtx& ctx = ctxs[dev];
  int& state = states[dev];
  if (state == 0) {
    state = -1;
    if (!env_on("PS_NO_SIDE")) {
      // the side stream carries filler (weight gradients, table scatters): LOWEST priority, so that when both streams have
      // workgroups ready the dependent chain of the main stream is dispatched first (PS_SIDE_PRIO=0: default priority)
      static const bool low_prio = ps_diag_int("PS_SIDE_PRIO", 1) != 0;
      int prio_lo = 0, prio_hi = 0;
      bool ok = false;
      if (low_prio && hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) == hipSuccess && prio_lo != prio_hi)
        ok = hipStreamCreateWithPriority(&ctx.stream, hipStreamNonBlocking, prio_lo) == hipSuccess;
      if (!ok) { (void)hipGetLastError(); ok = hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking) == hipSuccess; }
      for (int i = 0; ok && i < 8; ++i) ok = hipEventCreateWithFlags(&ctx.ev[i], hipEventDisableTiming) == hipSuccess;
      ok = ok && hipEventCreateWithFlags(&ctx.join, hipEventDisableTiming) == hipSuccess;
      ctx.next = 0; ctx.used = false;
      ctx.flag = nullptr; ctx.fork_seq = 0; ctx.join_seq = 0; ctx.light = false;
      ctx.sig_pending = false; ctx.sig_val = 0; ctx.sig_stream = nullptr;
      // forks / joins as stream write-value / wait-value operations on a device word instead of event pairs: the waiting
      // stream loses ~3 us per crossing instead of 6-12 when the waits are SHORT (C2: 0.353 -> 0.341 ms/step), but a
      // polled wait that lasts hundreds of microseconds wakes up late (review transformer 0.924 -> 0.942 ms, C5 1.64 ->
      // 1.72 ms), so the backward picks per step (side_set_light).  PS_SIDE_EVENTS=1 keeps the events everywhere; so
      // does a stream that is being captured into a graph (side_fork / side_join check), and so does any environment
      // in which dispatches may be serialised (dispatch_may_be_serialised: the value wait would never return).
      int can_wait = 0;
      const bool want = !env_on("PS_SIDE_EVENTS") && !dispatch_may_be_serialised();
      if (ok && want &&
          hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, dev) == hipSuccess && can_wait) {
        if (hipMalloc((void**)&ctx.flag, 2 * sizeof(uint32_t)) != hipSuccess || hipMemset(ctx.flag, 0, 2 * sizeof(uint32_t)) != hipSuccess)
          ctx.flag = nullptr;
        if (ctx.flag) {                        // the value crossings are used only where they are PROVEN to make progress
          const int v = side_value_selftest(ctx);
          if (v <= 0) { (void)hipFree(ctx.flag); ctx.flag = nullptr; }
          if (v < 0) ok = false;               // unusable side stream: one stream
        }
        // (tests of the wrap guard in side_fork: PS_SIDE_SEQ0 starts both sequences — and the words — at that value)
        const char* s0 = getenv("PS_SIDE_SEQ0");
        if (ctx.flag && s0 && *s0) {
          const uint32_t v[2] = {(uint32_t)strtoul(s0, nullptr, 0), (uint32_t)strtoul(s0, nullptr, 0)};
          if (hipMemcpy(ctx.flag, v, sizeof(v), hipMemcpyHostToDevice) == hipSuccess) { ctx.fork_seq = v[0]; ctx.join_seq = v[1]; }
        }
      }
      (void)hipGetLastError();
      if (ok) state = 1;
    }
  }
  return state == 1 ? &ctx : nullptr;
}
// 1 if this process crosses streams with value waits (the self-test passed), 0 if with event pairs / on one stream
extern "C" int ps_side_values_in_use(void) {
  SideCtx* c = side_ctx();
  return c && c->flag ? 1 : 0;
}
static int& side_mode_slot() {
  static int v = ps_env_int("PS_SIDE_MODE", 3);
  return v;
}
extern "C" int ps_set_side_mode(int mode) {
  const int old = side_mode_slot();
  side_mode_slot() = mode;
  return old;
}
static bool stream_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
  return cs != hipStreamCaptureStatusNone;
}
float* ps_det_scratch(int slot, size_t floats, hipStream_t st) {
  static float* buf[PS_MAX_DEVICES][3];
  static size_t cap[PS_MAX_DEVICES][3];
  static std::mutex mu;
  int dev = 0;
  if (slot < 0 || slot > 2) return nullptr;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PS_MAX_DEVICES) { (void)hipGetLastError(); return nullptr; }
  std::lock_guard<std::mutex> lock(mu);
  if (cap[dev][slot] < floats) {
    if (stream_capturing(st)) return nullptr;
    (void)hipStreamSynchronize(st);                      // the old buffer may still be read by a queued launch
    if (buf[dev][slot]) (void)hipFree(buf[dev][slot]);
    buf[dev][slot] = nullptr; cap[dev][slot] = 0;
    const size_t want = floats + floats / 4;
    if (hipMalloc((void**)&buf[dev][slot], want * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    cap[dev][slot] = want;
  }
  return buf[dev][slot];
}
static bool fork_by_kernel() {
  static const bool on = ps_diag_int("PS_FORK_BY_KERNEL", 1) != 0;
  return on;
}
// fork: everything enqueued on the main stream so far is visible to later side-stream work.  Each fork costs the
// main stream one event packet (~6 us before its next kernel, measured), so callers batch their weight gradients.
static int side_fork_injected_failure();
int side_fork(hipStream_t main_st) {
  SideCtx* c = side_ctx();
  if (!c) return PS_OK;
  // measured (ms/step, back-to-back launch cost on the main stream afterwards): events only 0.3527 / 4.19 us; forks as value
  // ops 0.3525 / 4.23; joins 0.3444 / 4.22; both 0.3435 / 9.5 (!) -> round 1: only the JOIN used them.  Round 2 (two forks per
  // backward, shorter kernels between them): events 0.3065, forks 0.3023, joins 0.3033, both 0.2980 -> both
  const int side_mode = side_mode_slot();   // bit 0: forks, bit 1: joins as value ops
  if (c->flag && c->light && (side_mode & 1) && !stream_capturing(main_st)) {
    if (c->fork_seq > 0xfff00000u || c->join_seq > 0xfff00000u) {
      // the sequence words are compared with >=: long before they wrap (2 per step: weeks of training) drain both streams
      // and start over from zero
      if (c->sig_pending) { PS_CHECK_HIP(hipStreamWriteValue32(c->sig_stream, c->flag, c->sig_val, 0)); c->sig_pending = false; }
      PS_CHECK_HIP(hipStreamSynchronize(c->stream));
      PS_CHECK_HIP(hipStreamSynchronize(main_st));
      PS_CHECK_HIP(hipMemset(c->flag, 0, 2 * sizeof(uint32_t)));
      c->fork_seq = 0; c->join_seq = 0;
    }
    ++c->fork_seq;
    // round 2, later: the value is stored by the NEXT kernel of the main stream as it starts (common.h, fork_signal) instead
    // of by a write operation between two dependent kernels — the timeline showed 9.5 and 10.5 us between the kernels around
    // the two forks of the C2 backward, half of it the write.  An unclaimed signal (no carrying launch follows, or one on
    // another stream) is flushed by the join.  PS_FORK_BY_KERNEL=0: the write operation.
    if (fork_by_kernel()) {
      if (c->sig_pending && c->sig_stream != main_st) PS_CHECK_HIP(hipStreamWriteValue32(c->sig_stream, c->flag, c->sig_val, 0));
      c->sig_pending = true; c->sig_val = c->fork_seq; c->sig_stream = main_st;   // (a newer value also satisfies an older wait)
    } else {
      PS_CHECK_HIP(hipStreamWriteValue32(main_st, c->flag, c->fork_seq, 0));
    }
    PS_CHECK_HIP(hipStreamWaitValue32(c->stream, c->flag, c->fork_seq, hipStreamWaitValueGte, 0xffffffffu));
    c->used = true;
    return side_fork_injected_failure();
  }
  if (c->sig_pending) {      // an event fork behind a pending value fork: release that one first
    PS_CHECK_HIP(hipStreamWriteValue32(c->sig_stream, c->flag, c->sig_val, 0));
    c->sig_pending = false;
  }
  hipEvent_t ev = c->ev[c->next];
  c->next = (c->next + 1) & 7;
  PS_CHECK_HIP(hipEventRecord(ev, main_st));
  PS_CHECK_HIP(hipStreamWaitEvent(c->stream, ev, 0));
  c->used = true;
  return side_fork_injected_failure();
}
// short steps cross streams with write / wait-value operations, long ones with events (see side_ctx)
void side_set_light(bool light) {
  static const int force = ps_diag_int("PS_SIDE_LIGHT", -1);   // tuning: 0 never, 1 always
  SideCtx* c = side_ctx();
  // (with forks signalled by the next kernel the value crossings win on the long steps too: review transformer 0.531 -> 0.527,
  // C5 shard 1.41 -> 1.38 ms per step; PS_SIDE_LIGHT=0 / PS_FORK_BY_KERNEL=0 restore the per-step choice)
  if (c) c->light = force >= 0 ? force != 0 : (light || fork_by_kernel());
}
hipStream_t side_stream_or(hipStream_t main_st) {
  SideCtx* c = side_ctx();
  return c ? c->stream : main_st;
}
// launch on the side stream (after the last fork); on the main stream when the side stream is disabled
int side_run(GemmProblem* ps, int n, hipStream_t main_st) {
  SideCtx* c = side_ctx();
  return run_wgrads(ps, n, c ? c->stream : main_st);
}
int side_wgrads(GemmProblem* ps, int n, hipStream_t main_st) {
  TRY(side_fork(main_st));
  return side_run(ps, n, main_st);
}
bool side_take_signal(hipStream_t st, uint32_t** flag, uint32_t* val) {
  SideCtx* c = side_ctx();
  if (!c || !c->sig_pending || st != c->sig_stream) return false;
  *flag = c->flag; *val = c->sig_val;
  c->sig_pending = false;
  return true;
}
void side_repend_signal(hipStream_t st, uint32_t val) {
  SideCtx* c = side_ctx();
  if (!c || !c->flag) return;
  if (!c->sig_pending || (int32_t)(val - c->sig_val) > 0) c->sig_val = val;
  c->sig_pending = true; c->sig_stream = st;
}
int side_join(hipStream_t main_st) {
  SideCtx* c = side_ctx();
  if (c && c->sig_pending) {     // nobody carried the last fork's signal: a stream write after all, or the side stream never starts
    PS_CHECK_HIP(hipStreamWriteValue32(c->sig_stream, c->flag, c->sig_val, 0));
    c->sig_pending = false;
  }
  if (!c || !c->used) return PS_OK;
  const int side_mode = side_mode_slot();
  if (c->flag && c->light && (side_mode & 2) && !stream_capturing(main_st)) {
    ++c->join_seq;
    PS_CHECK_HIP(hipStreamWriteValue32(c->stream, c->flag + 1, c->join_seq, 0));
    PS_CHECK_HIP(hipStreamWaitValue32(main_st, c->flag + 1, c->join_seq, hipStreamWaitValueGte, 0xffffffffu));
    c->used = false;
    return PS_OK;
  }
  PS_CHECK_HIP(hipEventRecord(c->join, c->stream));
  PS_CHECK_HIP(hipStreamWaitEvent(main_st, c->join, 0));
  c->used = false;
  return PS_OK;
}

// A failed entry point must not leave the side stream parked on a value nobody will store (a kernel-carried fork whose
// carrying launch never happened): release it, so that the caller's next synchronize returns and the error surfaces.
void side_abort() {
  SideCtx* c = side_ctx();
  if (!c) return;
  if (c->sig_pending && c->flag) (void)hipStreamWriteValue32(c->sig_stream, c->flag, c->sig_val, 0);
  c->sig_pending = false;
  (void)hipGetLastError();
}
extern "C" void ps_side_abort(void) { side_abort(); }
// test hook: the n-th side_fork from now on fails AFTER it has parked the side stream (0 = off)
static int g_fail_fork_in = 0;
extern "C" void ps_debug_fail_fork(int nth) { g_fail_fork_in = nth; }
static int side_fork_injected_failure() {
  if (g_fail_fork_in > 0 && --g_fail_fork_in == 0) { ps_set_error("injected failure behind a side-stream fork (ps_debug_fail_fork)"); return PS_ERR_ARG; }
  return PS_OK;
}

// does the (one-layer) encoder walk the valid-row list?  Then the rows of x at padded positions are never read, forward
// or backward (K / V products, attention and their gradients all go through the list), and need not be written.
bool enc_rowlist_taken(const PsTemDesc& D, const Ws& w, bool rows_listed) {
  static const bool rows_on = ps_env_int("PS_NO_ROWLIST", 0) == 0;
  if (!(rows_on && rows_listed && D.n_layers == 1 && w.qpos == 0 && w.vrows != 0 && w.S <= 64)) return false;
  const LayerWs& l = w.layer[0];
  if (l.Sq != 1 || l.n_in != D.B) return false;
  AttnArgs probe;
  memset(&probe, 0, sizeof(probe));
  probe.Sq = 1; probe.S = w.S; probe.d = D.d; probe.H = D.H;
  return attn_sq1_fits(probe);
}

static bool g_dx_two_partials = false;      // set by enc_layers_backward for the embed backward that follows it (single-thread contract)
static bool g_split_bwd_deferred = false;   // encode_forward -> enc_layers_forward: the fused projection + attention launch re-splits the backward's streams
// the predicate of enc_layers_forward's fused projection + attention launch, from what encode_forward knows before the embed launch
static bool kvq_fwd_will_fuse(const PsTemDesc& D, const PsTemTensors& P, float* ws, const Ws& w, bool rows_listed) {
  if (D.model != PS_MODEL_TEM || D.n_layers != 1 || !ps_fusion_enabled()) return false;
  const LayerWs& l = w.layer[0];
  const WSplit kvs = make_wsplit(D, P, ws, w);
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.n_in = l.n_in; a.fan = l.fan; a.H = D.H; a.S = w.S; a.Sq = l.Sq; a.d = D.d; a.dh = D.d / (D.H > 0 ? D.H : 1); a.qpos = w.qpos;
  a.seq_div = l.n_in / D.B; a.L = D.L; a.P = D.product_size;
  attn_finish(a);
  return kvs.on && kvs.fwd_kv && enc_rowlist_taken(D, w, rows_listed) && attn_sq1_fits(a) && a.fan > 1 && attn_wf_fits(a) &&
         kvq_attn_fits(a) && l.amask;
}
int enc_layers_forward(const PsTemDesc& D, const PsTemTensors& P, const int64_t* ui, const float* valid, float* ws,
                       const Ws& w, hipStream_t st, bool rows_listed, const ScoreArgs* fold_sc) {
  const int B = D.B, d = D.d, S = w.S, NL = D.n_layers;
  const float qscale = 1.f / sqrtf((float)(d / (D.H > 0 ? D.H : 1)));
  bool fused_final = false;
  for (int i = 0; i < NL; ++i) {
    const LayerWs& l = w.layer[i];
    const PsLayerTensors& Lp = P.layer[i];
    PS_REQUIRE(Lp.wk && Lp.wv && Lp.wq && Lp.wo && Lp.w1 && Lp.w2 && Lp.bk && Lp.bv && Lp.bq && Lp.bo && Lp.b1 &&
               Lp.b2 && Lp.ff_ln_g && Lp.ff_ln_b, "forward: layer %d has null tensors", i);
    const float* xin = i == 0 ? ws + w.x : ws + w.layer[i - 1].y2;
    const int ns = l.n_in * S;
    if (i != 0) {   // pre-LayerNorm only when iter != 0 (transformer.py:48-51)
      PS_REQUIRE(Lp.ln_g && Lp.ln_b, "forward: layer %d null pre-LN", i);
      LnFwdArgs a = {xin, d, ws + l.xn, d, ws + l.pre_stats, Lp.ln_g, Lp.ln_b, ns, d, 1e-6f};
      TRY(launch_ln_fwd(a, st));
    }
    const float* xn = ws + l.xn;
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.n_in = l.n_in; a.fan = l.fan; a.H = D.H; a.S = S; a.Sq = l.Sq; a.d = d; a.dh = d / D.H; a.qpos = w.qpos;
    a.seq_div = l.n_in / B; a.L = D.L; a.P = D.product_size; a.ui = ui; a.valid = valid;
    a.kp = ws + l.kp; a.vp = ws + l.vp; a.qp = ws + l.qp; a.attn = ws + l.attn; a.ctx = ws + l.ctx;
    a.drop = make_drop(D, PS_SITE_ATTN(i));
    a.qscale = qscale;
    attn_finish(a);
    // One layer, replicas, d = 128: projections + attention of the one consumed position in ONE launch, a workgroup per
    // sequence (kvq_attn_fwd_kernel): the K / V weight fragments were re-split by the embed launch in front (WSplit::fwd_kv)
    const WSplit kvs = (i == 0 && NL == 1) ? make_wsplit(D, P, ws, w) : WSplit{};
    const bool kvq_fused = i == 0 && NL == 1 && kvs.on && kvs.fwd_kv && ps_fusion_enabled() && enc_rowlist_taken(D, w, rows_listed) &&
                           attn_sq1_fits(a) && a.fan > 1 && attn_wf_fits(a) && kvq_attn_fits(a) && l.amask;
    PS_REQUIRE(kvq_fused || !g_split_bwd_deferred, "forward: the embed launch left the backward's weight streams to a launch that is not coming");
    if (kvq_fused) {
      KvqArgs q;
      memset(&q, 0, sizeof(q));
      if (g_split_bwd_deferred) q.split = kvs;       // the backward-only streams, left out of the embed launch (encode_forward)
      g_split_bwd_deferred = false;
      q.at = a; q.x = xn; q.kv_stream = kvs.fwd_kv;
      q.bk = Lp.bk; q.bv = Lp.bv; q.wq = Lp.wq; q.bq = Lp.bq;
      q.kp = ws + l.kp; q.vp = ws + l.vp; q.qp = ws + l.qp; q.amask = reinterpret_cast<uint32_t*>(ws + l.amask);
      TRY(launch_kvq_attn_fwd(q, st));
    } else {   // K, V, Q projections (neural.py:192-197), Q pre-divided by sqrt(dh) (:206)
      GemmGroup g;
      memset(&g, 0, sizeof(g));
      g.n = 3;
      g.p[0] = gp(xn, d, 0, Lp.wk, d, 0, ws + l.kp, d, ns, d, d); g.p[0].bias = Lp.bk;
      g.p[1] = gp(xn, d, 0, Lp.wv, d, 0, ws + l.vp, d, ns, d, d); g.p[1].bias = Lp.bv;
      if (l.Sq == S) g.p[2] = gp(xn, d, 0, Lp.wq, d, 0, ws + l.qp, d, ns, d, d);
      else g.p[2] = gp(xn + (size_t)w.qpos * d, S * d, 0, Lp.wq, d, 0, ws + l.qp, d, l.n_in, d, d);
      g.p[2].bias = Lp.bq; g.p[2].alpha = qscale;
      // valid rows only (EmbedArgs::vrows): the K / V rows of padded positions are never read (sq1_load zero-fills them)
      if (i == 0 && enc_rowlist_taken(D, w, rows_listed)) {
        const int32_t* vr = reinterpret_cast<const int32_t*>(ws + w.vrows);
        const int32_t* vc = reinterpret_cast<const int32_t*>(ws + w.vcount);
        g.p[0].ridx = vr; g.p[0].rcount = vc;
        g.p[1].ridx = vr; g.p[1].rcount = vc;
      }
      TRY(ps_launch_gemm(g, st));
    }
    if (kvq_fused) { }
    else if (attn_sq1_fits(a) && a.fan > 1 && attn_wf_fits(a)) TRY(launch_attn_fwd_wf(a, reinterpret_cast<uint32_t*>(ws + l.amask), st));
    else if (attn_sq1_fits(a) && attn_w1_fits(a)) TRY(launch_attn_fwd_w1(a, st));
    else TRY(attn_sq1_fits(a) ? launch_attn_fwd_sq1(a, st) : launch_attn_fwd(a, st));
    const bool fuse = ps_fusion_enabled() && i == NL - 1 && l.Sq == 1 && mlp_fused_serves(d, D.F) && w.wsplit &&
                      P.final_ln_g && P.final_ln_b;
    PS_REQUIRE(!fold_sc || fuse, "forward: folded scoring without the fused last layer");
    if (fuse) {   // Wo + LN + W1 + GELU + W2 + final LN of the last layer in one kernel (mlp_fused.hip)
      MlpFwdArgs m;
      memset(&m, 0, sizeof(m));
      m.M = l.M2; m.F = D.F; m.fan = l.fan; m.S = S; m.qpos = w.qpos;
      m.ctx = ws + l.ctx; m.xin = xin;
      m.wo = Lp.wo; m.bo = Lp.bo; m.g1 = Lp.ff_ln_g; m.be1 = Lp.ff_ln_b; m.w1 = Lp.w1; m.b1 = Lp.b1;
      m.w2 = Lp.w2; m.b2 = Lp.b2; m.gf = P.final_ln_g; m.bef = P.final_ln_b;
      m.drop_ctx = make_drop(D, PS_SITE_CTX(i)); m.drop_ff1 = make_drop(D, PS_SITE_FF1(i));
      m.drop_ff2 = make_drop(D, PS_SITE_FF2(i));
      m.y1 = ws + l.y1; m.ln1 = ws + l.ln1; m.st1 = ws + l.ff_stats; m.a1 = ws + l.a1; m.h1 = ws + l.h1;
      m.y2 = ws + l.y2; m.stf = ws + w.fin_stats; m.enc = ws + w.enc;
      if (fold_sc) { m.fold_score = 1; m.sc = *fold_sc; }
      m.x3 = make_wsplit(D, P, ws, w);
      TRY(launch_mlp_fwd_fused(m, st));
      fused_final = true;
      continue;
    }
    {   // final_linear + dropout + residual (neural.py:228-231, transformer.py:56)
      GemmProblem p = gp(ws + l.ctx, d, 0, Lp.wo, d, 0, ws + l.y1, d, l.M2, d, d);
      p.bias = Lp.bo; p.drop = make_drop(D, PS_SITE_CTX(i));
      p.res.mode = RES_GATHER; p.res.ptr = xin; p.res.ld = d; p.res.Sq = l.Sq; p.res.fan = l.fan; p.res.S = S;
      p.res.qpos = w.qpos; res_finish(p.res);
      TRY(run1(p, st));
    }
    {   // PositionwiseFeedForward (neural.py:30-33)
      LnFwdArgs n = {ws + l.y1, d, ws + l.ln1, d, ws + l.ff_stats, Lp.ff_ln_g, Lp.ff_ln_b, l.M2, d, 1e-6f};
      TRY(launch_ln_fwd(n, st));
      GemmProblem p1 = gp(ws + l.ln1, d, 0, Lp.w1, d, 0, ws + l.h1, D.F, l.M2, D.F, d);
      p1.bias = Lp.b1; p1.aux_out = ws + l.a1; p1.act = ACT_GELU; p1.drop = make_drop(D, PS_SITE_FF1(i));
      TRY(run1(p1, st));
      GemmProblem p2 = gp(ws + l.h1, D.F, 0, Lp.w2, D.F, 0, ws + l.y2, d, l.M2, d, D.F);
      p2.bias = Lp.b2; p2.drop = make_drop(D, PS_SITE_FF2(i));
      p2.res.mode = RES_DIRECT; p2.res.ptr = ws + l.y1; p2.res.ld = d;
      TRY(run1(p2, st));
    }
  }
  if (fused_final) return PS_OK;
  // final LayerNorm (transformer.py:86) on the consumed position only
  PS_REQUIRE(P.final_ln_g && P.final_ln_b, "forward: null final LayerNorm");
  LnFwdArgs f;
  if (NL > 0) {
    const LayerWs& l = w.layer[NL - 1];
    f = LnFwdArgs{ws + l.y2, d, ws + w.enc, d, ws + w.fin_stats, P.final_ln_g, P.final_ln_b, w.Mf, d, 1e-6f};
  } else {
    f = LnFwdArgs{ws + w.x + (size_t)w.qpos * d, S * d, ws + w.enc, d, ws + w.fin_stats, P.final_ln_g,
                  P.final_ln_b, B, d, 1e-6f};
  }
  TRY(launch_ln_fwd(f, st));
  return PS_OK;
}

// -------------------------------------------------------------- encoder forward
struct SamplerArgs { const float* prob; const int32_t* alias; int64_t* items; int64_t* words; };

// The valid-row list is built by one workgroup per sequence that counts the valid positions of ALL earlier sequences
// (B^2 L / 2 index reads in total): fine up to a few thousand sequences, dense products beyond
static bool rows_list_ok(const PsTemDesc& D) { return D.L <= 64 && (int64_t)D.B * D.B * D.L <= ((int64_t)64 << 20); }

static int encode_forward(const PsTemDesc& D, const PsTemTensors& P, const PsTemBatch& Bt, float* ws, const Ws& w,
                          hipStream_t st, const SamplerArgs* samp = nullptr, const ScoreArgs* fold_sc = nullptr) {
  const bool tem = D.model == PS_MODEL_TEM;
  const int B = D.B, d = D.d, S = w.S, NL = tem ? D.n_layers : 0;
  const float* hist = D.sep_prod_emb ? P.hist_product_emb : P.product_emb;
  PS_REQUIRE(P.word_emb && P.product_emb && hist, "forward: null embedding table");
  EmbedArgs e;
  memset(&e, 0, sizeof(e));
  e.B = B; e.Q = D.Q; e.L = D.L; e.S = S; e.d = d; e.P = D.product_size; e.V = D.vocab_size;
  e.tem = tem; e.fs = D.query_encoder == PS_QENC_FS; e.use_pos = D.use_pos_emb;
  e.qw = Bt.query_word_idxs; e.ui = Bt.u_item_idxs;
  e.word_emb = P.word_emb; e.hist_tab = hist; e.pe = P.pe;
  e.drop_fs = make_drop(D, PS_SITE_FS);
  e.qmean_d = ws + w.qmean; e.query_emb = ws + w.query_emb; e.x = ws + w.x;
  PS_REQUIRE(e.qw && (!tem || e.ui), "forward: null batch indices");
  PS_REQUIRE(!tem || !D.use_pos_emb || P.pe, "forward: null positional table");
  if (tem && rows_list_ok(D)) { e.vrows = reinterpret_cast<int32_t*>(ws + w.vrows); e.vcount = reinterpret_cast<int32_t*>(ws + w.vcount); }
  const float* wp_w[PS_WPLANES_MAX]; int wp_r[PS_WPLANES_MAX], wp_c[PS_WPLANES_MAX];
  const int wp_n = wplane_list(D, P, w, wp_w, wp_r, wp_c);
  WPlaneScope wplanes(st, wp_w, wp_r, wp_c, wp_n);             // the big linears multiply against pre-split weight planes (gemm.hip)
  if (samp) {
    e.samp_prob = samp->prob; e.samp_alias = samp->alias; e.samp_items = samp->items; e.samp_words = samp->words;
    e.samp_nitem = D.B * D.K; e.samp_nword = D.B * D.W * D.K; e.samp_step = (uint32_t)D.step;
    e.samp_k0 = (uint32_t)(D.seed & 0xffffffffu); e.samp_k1 = (uint32_t)(D.seed >> 32);
  }
  PS_REQUIRE(!e.fs || (P.fs_w && P.fs_b), "forward: null FS encoder weights");
  // FS projection as a per-row mat-vec inside the embed launch — while the [d,d] weight a workgroup streams from L2 is
  // small (64 KB at d = 128; at d = 256 the C5 step lost 60 us to it and the GEMM launch is the better deal)
  const bool fs_fused = e.fs && ps_fusion_enabled() && d <= 128;
  if (fs_fused) { e.fs_w = P.fs_w; e.fs_b = P.fs_b; }
  if (fold_sc) { e.fold_words = 1; e.sc = *fold_sc; }
  e.split = make_wsplit(D, P, ws, w);
  g_split_bwd_deferred = e.split.on && D.training && kvq_fwd_will_fuse(D, P, ws, w, rows_list_ok(D));
  e.split_fwd_only = g_split_bwd_deferred ? 1 : 0;
  TRY(launch_embed_fwd(e, st));
  if (e.fs && !fs_fused) {   // FSEncoder: tanh(f_W . mean + b)  (text_encoder.py:39); also writes row 0 of x (+pe[0])
    GemmProblem p = gp(ws + w.qmean, d, 0, P.fs_w, d, 0, ws + w.query_emb, d, B, d, d);
    p.bias = P.fs_b; p.act = ACT_TANH;
    if (tem) { p.out2 = ws + w.x; p.ld2 = S * d; p.add2 = D.use_pos_emb ? P.pe : nullptr; }
    TRY(run1(p, st));
  }
  if (!tem) return PS_OK;

  return enc_layers_forward(D, P, Bt.u_item_idxs, nullptr, ws, w, st, rows_list_ok(D), fold_sc);
}

// Folded scoring (ScoreArgs): TEM training forward with replicas whose last layer takes the wave-specialised fused form
static bool can_fold_score(const PsTemDesc& D, const PsTemTensors& P, const Ws& w, hipStream_t st) {
  if (D.model != PS_MODEL_TEM || D.n_layers < 1 || D.C != 0 || w.R != D.K + 1 || w.R < 2 || D.W < 1) return false;
  const LayerWs& l = w.layer[D.n_layers - 1];
  if (l.Sq != 1 || l.M2 != w.Mf || !P.final_ln_g || !P.final_ln_b) return false;
  if (stream_capturing(st)) return false;            // graph replay patches the loss kernel's arguments: unfolded form
  return mlp_fwd_can_fold_score(w.Mf, D.F, D.d);
}
static void fold_finish(const PsTemDesc& D, float* ws, const Ws& w, ScoreArgs& s, const SamplerArgs* samp) {
  s.word_blk = ws + w.word_blk; s.item_blk = ws + w.item_blk; s.ticket = reinterpret_cast<uint32_t*>(ws + w.ticket);
  s.word_nblk = ps_cdiv((int64_t)D.B * D.W * (D.K + 1), PS_WORD_TASKS_PER_WG);
  if (samp) {
    s.samp_inline = 1; s.samp_prob = samp->prob; s.samp_alias = samp->alias;
    s.samp_step = (uint32_t)D.step; s.samp_k0 = (uint32_t)(D.seed & 0xffffffffu); s.samp_k1 = (uint32_t)(D.seed >> 32);
  }
}

static void fill_score(const PsTemDesc& D, const PsTemTensors& P, const PsTemBatch& Bt, float* ws, const Ws& w,
                       ScoreArgs& s) {
  memset(&s, 0, sizeof(s));
  s.B = D.B; s.K = D.K; s.W = D.W; s.C = 0; s.R = w.R; s.d = D.d;
  s.P = D.product_size; s.V = D.vocab_size;
  s.bias_product = D.bias_product; s.pos_weight = D.pos_weight;
  s.target = Bt.target_prod_idxs; s.neg_items = Bt.neg_item_idxs; s.pos_words = Bt.pos_iword_idxs;
  s.neg_words = Bt.neg_word_idxs; s.candi = Bt.candi_prod_idxs;
  s.product_emb = P.product_emb; s.word_emb = P.word_emb; s.product_bias = P.product_bias; s.word_bias = P.word_bias;
  s.enc = ws + w.enc;
  s.item_scores = ws + w.item_scores; s.word_scores = ws + w.word_scores; s.loss_parts = ws + w.loss_parts;
  s.item_terms = ws + w.item_terms; s.word_terms = ws + w.word_terms; s.loss_blk = ws + w.loss_blk;
  score_finish(s);
}

extern "C" int ps_tem_forward(const PsTemDesc* desc, const PsTemTensors* params, const PsTemBatch* batch,
                              float* workspace, float* loss3, float* loss_acc, ps_stream_t stream) {
  PS_REQUIRE(desc && params && batch && workspace && loss3, "forward: null argument");
  PsTemDesc D = *desc;
  D.C = 0;
  Ws w;
  TRY(make_ws(D, w));
  hipStream_t st = (hipStream_t)stream;
  PS_REQUIRE(batch->target_prod_idxs && batch->neg_item_idxs && (D.W == 0 || (batch->pos_iword_idxs &&
             batch->neg_word_idxs)), "forward: null batch tensors");
  PS_REQUIRE(params->word_bias && (!D.bias_product || params->product_bias), "forward: null bias tensors");
  ScoreArgs s;
  fill_score(D, *params, *batch, workspace, w, s);
  s.loss3 = loss3; s.loss_acc = loss_acc;
  if (can_fold_score(D, *params, w, st)) {             // no gather+score / loss launches: see ScoreArgs, folded form
    fold_finish(D, workspace, w, s, nullptr);
    return encode_forward(D, *params, *batch, workspace, w, st, nullptr, &s);
  }
  TRY(encode_forward(D, *params, *batch, workspace, w, st));
  TRY(launch_score_fwd(s, st));
  TRY(launch_loss(s, st));
  return PS_OK;
}

// forward with the two negative draws folded into its first launch: neg_item_out / neg_word_out receive the draws
// (they are what batch->neg_* would have held) and must stay alive until the backward has run.
extern "C" int ps_tem_forward_sampled(const PsTemDesc* desc, const PsTemTensors* params, const PsTemBatch* batch,
                                      const float* alias_prob, const int32_t* alias_idx, int64_t* neg_item_out,
                                      int64_t* neg_word_out, float* workspace, float* loss3, float* loss_acc,
                                      ps_stream_t stream) {
  PS_REQUIRE(desc && params && batch && workspace && loss3 && alias_prob && alias_idx && neg_item_out && neg_word_out,
             "forward_sampled: null argument");
  PsTemDesc D = *desc;
  D.C = 0;
  Ws w;
  TRY(make_ws(D, w));
  hipStream_t st = (hipStream_t)stream;
  PS_REQUIRE(batch->target_prod_idxs && (D.W == 0 || batch->pos_iword_idxs), "forward_sampled: null batch tensors");
  PS_REQUIRE(params->word_bias && (!D.bias_product || params->product_bias), "forward_sampled: null bias tensors");
  PsTemBatch Bt = *batch;
  Bt.neg_item_idxs = neg_item_out; Bt.neg_word_idxs = neg_word_out;
  const SamplerArgs samp = {alias_prob, alias_idx, neg_item_out, neg_word_out};
  ScoreArgs s;
  fill_score(D, *params, Bt, workspace, w, s);
  s.loss3 = loss3; s.loss_acc = loss_acc;
  if (can_fold_score(D, *params, w, st)) {
    fold_finish(D, workspace, w, s, &samp);
    return encode_forward(D, *params, Bt, workspace, w, st, &samp, &s);
  }
  TRY(encode_forward(D, *params, Bt, workspace, w, st, &samp));
  TRY(launch_score_fwd(s, st));
  TRY(launch_loss(s, st));
  return PS_OK;
}

extern "C" int ps_gather_score(const PsTemDesc* desc, const PsTemTensors* params, const PsTemBatch* batch,
                               float* workspace, ps_stream_t stream) {
  PS_REQUIRE(desc && params && batch && workspace, "gather_score: null argument");
  PsTemDesc D = *desc;
  D.C = 0;
  Ws w;
  TRY(make_ws(D, w));
  ScoreArgs s;
  fill_score(D, *params, *batch, workspace, w, s);
  return launch_score_fwd(s, (hipStream_t)stream);
}

extern "C" int ps_tem_score(const PsTemDesc* desc, const PsTemTensors* params, const PsTemBatch* batch,
                            float* workspace, float* scores, ps_stream_t stream) {
  PS_REQUIRE(desc && params && batch && workspace && scores, "score: null argument");
  PsTemDesc D = *desc;
  PS_REQUIRE(D.C > 0 && batch->candi_prod_idxs, "score: needs C > 0 candidates");
  D.training = 0;
  Ws w;
  TRY(make_ws(D, w));
  hipStream_t st = (hipStream_t)stream;
  TRY(encode_forward(D, *params, *batch, workspace, w, st));
  ScoreArgs s;
  fill_score(D, *params, *batch, workspace, w, s);
  s.C = D.C; s.item_scores = scores; score_finish(s);
  TRY(launch_score_fwd(s, st));
  return PS_OK;
}

// model.eval() sequence representation the dot-product heads consume (item_transformer.py:118-131): one encode per
// (user, query) row, replicas collapse (R = 1).  Feeds ps_rank_all (full-catalogue evaluation).
extern "C" int ps_tem_encode(const PsTemDesc* desc, const PsTemTensors* params, const PsTemBatch* batch,
                             float* workspace, float* enc_out, ps_stream_t stream) {
  PS_REQUIRE(desc && params && batch && workspace && enc_out, "encode: null argument");
  PsTemDesc D = *desc;
  PS_REQUIRE(D.C > 0, "encode: build the descriptor in eval mode (C >= 1)");
  D.training = 0;
  Ws w;
  TRY(make_ws(D, w));
  hipStream_t st = (hipStream_t)stream;
  TRY(encode_forward(D, *params, *batch, workspace, w, st));
  PS_CHECK_HIP(hipMemcpyAsync(enc_out, workspace + w.enc, sizeof(float) * (size_t)D.B * D.d, hipMemcpyDeviceToDevice, st));
  return PS_OK;
}

// park the {dgamma, dbeta, colsum} column sums of one LN backward (see ColFoldList) when the caller collects them
static void park_colsums(LnBwdArgs& a, float* ws, const Ws& w, ColFoldList* fold) {
  if (!fold || fold->n >= PS_MAX_COLFOLD) return;
  ColFold& f = fold->e[fold->n];
  a.partial = ws + w.lnpart + (size_t)fold->n * w.lnrows * 3 * a.d;
  f.partial = a.partial; f.nblk = ln_bwd_blocks(a.rows); f.d = a.d;
  f.dst[0] = a.dgamma; f.dst[1] = a.dbeta; f.dst[2] = a.colsum;
  ++fold->n;
}

static int& fuse_bwd_min_slot() {
  static int v = ps_env_int("PS_FUSE_BWD_MIN", 1024);
  return v;
}
extern "C" int ps_set_fuse_bwd_min(int rows) {
  const int old = fuse_bwd_min_slot();
  fuse_bwd_min_slot() = rows;
  return old;
}

// K/V/Q weight gradients of the first layer that the caller launches on the main stream AFTER its embedding scatter
// (PS_WG3_LAST, item transformer): the scatter (atomics) then shares the machine with the side stream's W2 / W1 / Wo
// products, and these follow when those are nearly through, instead of slowing each other down product beside product
static thread_local GemmProblem g_wg3_last[4];
static thread_local int g_wg3_last_n = 0;
static thread_local bool g_wg3_defer_ok = false;          // set by a caller that will flush them
static int flush_wg3_last(hipStream_t st) {
  const int n = g_wg3_last_n;
  g_wg3_last_n = 0;
  return n ? run_wgrads(g_wg3_last, n, st) : PS_OK;
}
int enc_layers_backward(const PsTemDesc& D, const PsTemTensors& P, const PsTemTensors& G, const int64_t* ui,
                        const float* valid, float* ws, const Ws& w, hipStream_t st, ColFoldList* fold,
                        const ScoreArgs* score_on_side, bool rows_listed) {
  const bool drop = D.training && D.dropout > 0.f;
  const int B = D.B, d = D.d, S = w.S, NL = D.n_layers, F = D.F;
  PS_REQUIRE(G.final_ln_g && G.final_ln_b, "backward: null final LayerNorm gradient");
  // 2. final LayerNorm backward
  LnBwdArgs f;
  memset(&f, 0, sizeof(f));
  f.dy = ws + w.denc; f.lddy = d; f.stats = ws + w.fin_stats; f.g = P.final_ln_g; f.d = d;
  f.dgamma = G.final_ln_g; f.dbeta = G.final_ln_b;
  // The last layer's whole per-replica backward (final LN, FFN, FF LN, Wo) as one kernel (mlp_fused.hip) when the
  // forward took the fused form too; needs parked column sums (fold) and one parked row per workgroup (Ws::lnrows of them).
  side_set_light((int64_t)B * S * d <= ((int64_t)2 << 20));   // C2: 1.03 M elements of x; review transformer 10 M; C5 5.5 M
  static const bool bwd_fuse_on = ps_env_int("PS_NO_FUSE_BWD", 0) == 0;
  const int bwd_fuse_min = fuse_bwd_min_slot();
  const bool fuse_last = NL > 0 && fold && bwd_fuse_on && ps_fusion_enabled() && w.layer[NL - 1].Sq == 1 &&
                         mlp_fused_serves(d, F) && w.wsplit && w.layer[NL - 1].M2 == w.Mf && w.Mf >= bwd_fuse_min &&
                         mlp_bwd_fused_blocks(w.Mf) <= w.lnrows && fold->n + 3 <= PS_MAX_COLFOLD;
  if (score_on_side && !(fuse_last && w.R > 1)) {
    // not the fused form: d enc is needed first, so the score backward is cut in two — its d enc half leads the main
    // stream, its table scatter (the expensive half: 127 us of scattered atomics at C5) goes to the side stream
    ScoreArgs t = *score_on_side;
    t.denc = ws + w.denc;
    SideCtx* sc = side_ctx();
    if (sc && w.R > 1) {
      t.part = 1;
      TRY(launch_score_bwd(t, st));
      TRY(side_fork(st));
      t.part = 2;
      TRY(launch_score_bwd(t, sc->stream));
    } else {
      TRY(launch_score_bwd(t, st));
    }
    score_on_side = nullptr;
  }
  if (fuse_last) {
    // nothing here: launched inside the layer loop below
  } else if (NL > 0) {
    const LayerWs& l = w.layer[NL - 1];
    f.x = ws + l.y2; f.ldx = d; f.rows = w.Mf; f.dx = ws + w.dy2; f.lddx = d;
    f.colsum = G.layer[NL - 1].b2;
    if (drop) { f.out2 = ws + w.do2; f.drop2 = make_drop(D, PS_SITE_FF2(NL - 1)); }
    park_colsums(f, ws, w, fold);
    TRY(launch_ln_bwd(f, st));
  } else {
    PS_CHECK_HIP(hipMemsetAsync(ws + w.dx, 0, sizeof(float) * (size_t)B * S * d, st));
    f.x = ws + w.x + (size_t)w.qpos * d; f.ldx = S * d; f.rows = B;
    f.dx = ws + w.dx + (size_t)w.qpos * d; f.lddx = S * d;
    park_colsums(f, ws, w, fold);
    TRY(launch_ln_bwd(f, st));
  }
  // 3. layers, last to first
  for (int i = NL - 1; i >= 0; --i) {
    const LayerWs& l = w.layer[i];
    const PsLayerTensors& Lp = P.layer[i];
    const PsLayerTensors& Lg = G.layer[i];
    PS_REQUIRE(Lg.wk && Lg.wv && Lg.wq && Lg.wo && Lg.w1 && Lg.w2 && Lg.bk && Lg.bv && Lg.bq && Lg.bo && Lg.b1 &&
               Lg.b2 && Lg.ff_ln_g && Lg.ff_ln_b, "backward: layer %d has null gradients", i);
    const float* xin = i == 0 ? ws + w.x : ws + w.layer[i - 1].y2;
    const float* xn = ws + l.xn;
    const int ns = l.n_in * S, M2 = l.M2;
    const float* do2 = drop ? ws + w.do2 : ws + w.dy2;
    const bool fused = fuse_last && i == NL - 1;
    static const bool wgrad_early = ps_diag_int("PS_WGRAD_LATE", 0) == 0;
    if (fused) {
      MlpBwdArgs m;
      memset(&m, 0, sizeof(m));
      m.M = M2; m.F = F;
      m.denc = ws + w.denc; m.y2 = ws + l.y2; m.stf = ws + w.fin_stats; m.gf = P.final_ln_g;
      m.y1 = ws + l.y1; m.st1 = ws + l.ff_stats; m.g1 = Lp.ff_ln_g; m.a1 = ws + l.a1;
      m.wo = Lp.wo; m.w1 = Lp.w1; m.w2 = Lp.w2;
      m.drop_ctx = make_drop(D, PS_SITE_CTX(i)); m.drop_ff1 = make_drop(D, PS_SITE_FF1(i));
      m.drop_ff2 = make_drop(D, PS_SITE_FF2(i));
      if (score_on_side) {   // d enc from the scores (see MlpBwdArgs::item_scores)
        const ScoreArgs& sa = *score_on_side;
        m.item_scores = sa.item_scores; m.target = sa.target; m.neg_items = sa.neg_items; m.product_emb = sa.product_emb;
        m.B = sa.B; m.K = sa.K; m.pos_weight = sa.pos_weight; m.P = sa.P; m.scale = sa.scale; m.scale_dev = sa.scale_dev;
      }
      m.x3 = make_wsplit(D, P, ws, w);                 // the fragment streams the forward's embed launch left in the workspace
      m.do2 = const_cast<float*>(do2); m.da1 = ws + w.da1; m.dy1 = ws + w.dy1;
      m.dout = drop ? ws + w.do_ : ws + w.dy1; m.dctx = ws + w.dctx;
      const int nwg = mlp_bwd_fused_blocks(M2);
      {   // parked column sums: {final LN gamma, beta, b2}, {b1}, {FF LN gamma, beta, bo}
        ColFold& c0 = fold->e[fold->n];
        m.part_f = ws + w.lnpart + (size_t)fold->n * w.lnrows * 3 * d;
        c0.partial = m.part_f; c0.nblk = nwg; c0.d = d;
        c0.dst[0] = G.final_ln_g; c0.dst[1] = G.final_ln_b; c0.dst[2] = Lg.b2;
        ++fold->n;
        ColFold& c1 = fold->e[fold->n];
        m.part_1 = ws + w.lnpart + (size_t)fold->n * w.lnrows * 3 * d;
        c1.partial = m.part_1; c1.nblk = nwg; c1.d = d;
        c1.dst[0] = Lg.ff_ln_g; c1.dst[1] = Lg.ff_ln_b; c1.dst[2] = Lg.bo;
        ++fold->n;
        ColFold& c2 = fold->e[fold->n];
        m.part_b1 = ws + w.gcpart;
        c2.partial = m.part_b1; c2.nblk = mlp_bwd_b1_rows(M2, F); c2.d = F;
        c2.dst[0] = Lg.b1; c2.dst[1] = nullptr; c2.dst[2] = nullptr;
        ++fold->n;
      }
      // (measured and dropped: the table scatter of the score backward — it needs nothing of this backward — started beside
      // the fused kernel below, its fork carried by that kernel: starved by 252 workgroups that own their CUs' LDS it took
      // 74 us instead of 29 and slowed the attention backward behind it, 0.278 -> 0.282 ms/step)
      TRY(launch_mlp_bwd_fused(m, st));
      GemmProblem wg[1] = {gp_wgrad(do2, d, ws + l.h1, F, Lg.w2, d, F, M2)};
      GemmProblem wg1[1] = {gp_wgrad(ws + w.da1, F, ws + l.ln1, d, Lg.w1, F, d, M2)};
      GemmProblem wgo[1] = {gp_wgrad(m.dout, d, ws + l.ctx, d, Lg.wo, d, d, M2)};
      TRY(side_fork(st));                           // fork 1: W2, W1, Wo weight gradients under the attention backward
      if (score_on_side) {                          // ... led by the table scatter of the score backward (behind them instead: 0.284 -> 0.293 ms/step)
        ScoreArgs t = *score_on_side;
        t.denc = nullptr;
        SideCtx* sc = side_ctx();
        TRY(launch_score_bwd(t, sc ? sc->stream : st));
      }
      // (a second side stream for W1 / Wo beside W2 measured 0.389 vs 0.368 ms: slower)
      // one launch for the three (the flat group form: every member keeps its own split count; three launches of ~250
      // latency-bound workgroups one after the other took 86 us at C2; review transformer 0.563 -> 0.543 ms/step, C2 0.3156 ->
      // 0.3144).  PS_WGRAD_GROUP_ROWS=0 restores the separate launches.
      static const int wg_group_rows = ps_diag_int("PS_WGRAD_GROUP_ROWS", (1 << 30));
      if (M2 <= wg_group_rows) {
        GemmProblem all3[3] = {wg[0], wg1[0], wgo[0]};
        TRY(side_run(all3, 3, st));
      } else {
        TRY(side_run(wg, 1, st));
        TRY(side_run(wg1, 1, st));
        TRY(side_run(wgo, 1, st));
      }
    } else {
    // FFN backward
      GemmProblem p = gp(do2, d, 0, Lp.w2, F, 1, ws + w.da1, F, M2, F, d);      // d h1 = do2 . W2
      p.act = ACT_GELU_BWD; p.act_aux = ws + l.a1; p.drop = make_drop(D, PS_SITE_FF1(i)); p.colsum = Lg.b1;
      if (fold && fold->n < PS_MAX_COLFOLD && i == NL - 1) {      // park the b1 column sums (one buffer: last layer only)
        ColFold& cf = fold->e[fold->n++];
        p.colsum_part = ws + w.gcpart;
        cf.partial = p.colsum_part; cf.nblk = 4 * ps_cdiv(M2, 64); cf.d = F;
        cf.dst[0] = Lg.b1; cf.dst[1] = nullptr; cf.dst[2] = nullptr;
      }
      TRY(run1(p, st));
      // dW2 += do2^T . h1 and dW1 += da1^T . ln1 are launched further down, once the dX chain of the MLP is through
      // (beside it they slowed every link: 44 vs 33 us for the GEMM below); they then share the machine with the
      // attention backward and the big dX GEMM instead.  Measured a wash in step time (both orders 0.509 ms): the
      // backward is throughput-bound once both streams are busy.
      GemmProblem wg[1] = {gp_wgrad(do2, d, ws + l.h1, F, Lg.w2, d, F, M2)};
      GemmProblem wg1[1] = {gp_wgrad(ws + w.da1, F, ws + l.ln1, d, Lg.w1, F, d, M2)};
      GemmProblem q = gp(ws + w.da1, F, 0, Lp.w1, d, 1, ws + w.dln1, d, M2, d, F);  // d ln1 = da1 . W1
      TRY(run1(q, st));
      // fork 1: the two big weight gradients (W2, W1) start as soon as d a1 exists, under the LN backward, the Wo dX
      // GEMM and the attention backward.  (Forked one GEMM later, behind d ctx, the side stream's 112 us of weight
      // gradients ended 13 us after the main chain and the step paid a late join on top.)
      if (wgrad_early) {
        TRY(side_fork(st));
        TRY(side_run(wg, 1, st));
        TRY(side_run(wg1, 1, st));
      }
      LnBwdArgs n;
      memset(&n, 0, sizeof(n));
      n.dy = ws + w.dln1; n.lddy = d; n.x = ws + l.y1; n.ldx = d; n.stats = ws + l.ff_stats; n.g = Lp.ff_ln_g;
      n.rows = M2; n.d = d;
      n.res.mode = RES_DIRECT; n.res.ptr = ws + w.dy2; n.res.ld = d;            // residual  output + x
      n.dx = ws + w.dy1; n.lddx = d;
      if (drop) { n.out2 = ws + w.do_; n.drop2 = make_drop(D, PS_SITE_CTX(i)); }
      n.colsum = Lg.bo; n.dgamma = Lg.ff_ln_g; n.dbeta = Lg.ff_ln_b;
      park_colsums(n, ws, w, fold);
      TRY(launch_ln_bwd(n, st));
      const float* dout0 = drop ? ws + w.do_ : ws + w.dy1;
      GemmProblem pc = gp(dout0, d, 0, Lp.wo, d, 1, ws + w.dctx, d, M2, d, d);    // d ctx = do . Wo
      TRY(run1(pc, st));
      if (!wgrad_early) {
        GemmProblem wgo0[1] = {gp_wgrad(dout0, d, ws + l.ctx, d, Lg.wo, d, d, M2)};
        TRY(side_fork(st));                         // fork 1 (late form): W2, W1, Wo weight gradients under the attention backward
        TRY(side_run(wg, 1, st));
        TRY(side_run(wg1, 1, st));
        TRY(side_run(wgo0, 1, st));
      }
    }

    const float* dout = drop ? ws + w.do_ : ws + w.dy1;
    // attention backward
    {
      GemmProblem wgo[1] = {gp_wgrad(dout, d, ws + l.ctx, d, Lg.wo, d, d, M2)};
      AttnArgs a;
      memset(&a, 0, sizeof(a));
      a.n_in = l.n_in; a.fan = l.fan; a.H = D.H; a.S = S; a.Sq = l.Sq; a.d = d; a.dh = d / D.H; a.qpos = w.qpos;
      a.seq_div = l.n_in / B; a.L = D.L; a.P = D.product_size; a.ui = ui; a.valid = valid;
      a.kp = ws + l.kp; a.vp = ws + l.vp; a.qp = ws + l.qp; a.attn = ws + l.attn;
      a.drop = make_drop(D, PS_SITE_ATTN(i));
      a.dctx = ws + w.dctx;
      const bool qall = l.Sq == S;
      a.lddkv = qall ? 3 * d : 2 * d;
      a.dkv = ws + w.dkv;
      a.dq = qall ? ws + w.dkv + 2 * d : ws + w.dq;
      a.lddq = qall ? 3 * d : d;
      a.dbq = Lg.bq; a.dbk = Lg.bk; a.dbv = Lg.bv;
      a.qscale = 1.f / sqrtf((float)(d / D.H));
      attn_finish(a);
      const bool sq1 = attn_sq1_fits(a);
      if (sq1 && fold && fold->n < PS_MAX_COLFOLD) {
        // bias gradients: one parked row per sequence instead of n_in same-address atomics per column
        ColFold& cf = fold->e[fold->n++];
        a.bias_part = ws + w.abpart + (size_t)i * w.layer[D.n_layers - 1].n_in * 3 * d;
        cf.partial = a.bias_part; cf.nblk = l.n_in; cf.d = d;
        cf.dst[0] = Lg.bq; cf.dst[1] = Lg.bk; cf.dst[2] = Lg.bv;
      }
      // first layer, one query row per sequence, d == 128: dQ.Wq rides in the attention backward's tail (two partial
      // rows per sequence in the free d ln1 buffer) instead of a [n_in,128]x[128,128] GEMM launch of its own
      const bool wf = sq1 && a.fan > 1 && attn_wf_fits(a);   // one wave per (sequence, four heads), replicas inside
      const bool w1 = wf || (sq1 && attn_w1_fits(a));       // one wave per sequence (no replicas)
      const bool q_folded = sq1 && !qall && i == 0 && ps_fusion_enabled() &&
                            (w1 ? (!wf || d == 128) && (size_t)(wf ? 2 : 1) * l.n_in <= (size_t)M2
                                : d == 128 && attn_sq1_split(a) == 2 && (size_t)2 * l.n_in <= (size_t)M2);
      if (q_folded) { a.wq = Lp.wq; a.dxq_part = ws + w.dln1; a.fanin_src = ws + w.dy1; }
      // ... and, one-layer encoder with replicas: so does the K / V input gradient itself (AttnArgs::kvb_stream) — no dX GEMM launch
      // on the dependent chain; the embed scatter adds the two head groups' partial rows (EmbedBwdArgs::dx2)
      static const bool rows_on0 = ps_env_int("PS_NO_ROWLIST", 0) == 0;
      static const bool dx_fused_on = ps_env_int("PS_KVDX_FUSED", 1) != 0;
      const WSplit kvs = (i == 0 && NL == 1) ? make_wsplit(D, P, ws, w) : WSplit{};
      const bool dx_fused = dx_fused_on && wf && q_folded && i == 0 && NL == 1 && d == 128 && kvs.on && kvs.bwd_kv && !ps_deterministic() &&
                            rows_on0 && rows_listed && !qall && w.qpos == 0 && w.vrows != 0 && l.n_in == B && attn_bwd_wf_two_partials(a);
      if (dx_fused) { a.kvb_stream = kvs.bwd_kv; a.dxp[0] = ws + w.dx; a.dxp[1] = ws + w.dxn; }
      g_dx_two_partials = dx_fused;
      // valid rows only (below): the dK / dV rows of padded positions are then never read, and never written
      const bool listed0 = rows_on0 && rows_listed && sq1 && i == 0 && NL == 1 && !qall && w.qpos == 0 && w.vrows != 0 && l.n_in == B;
      // round 4: where dQ.Wq is NOT folded (d != 128: the C5 shard) the replicas' fan-in is summed by a launch of its own
      // (launch_fanin_sum, below) so that the dX product can still run over the row list: 133 -> ~50 us at C5
      const bool presum = listed0 && !q_folded && l.fan > 1 && l.Sq == 1 && !qall && i == 0 && (d % 4) == 0;
      if (wf) TRY(launch_attn_bwd_wf(a, reinterpret_cast<const uint32_t*>(ws + l.amask), listed0 && (q_folded || l.fan == 1 || presum), st));
      else if (w1) TRY(launch_attn_bwd_w1(a, listed0 && (q_folded || l.fan == 1 || presum), st));
      else TRY(sq1 ? launch_attn_bwd_sq1(a, st) : launch_attn_bwd(a, st));
      // weight gradients of Wo, Wk, Wv, Wq: one fork right behind the attention backward, off the dX chain
      GemmProblem wg3[3];
      wg3[0] = gp_wgrad(ws + w.dkv, a.lddkv, xn, d, Lg.wk, d, d, ns);
      wg3[1] = gp_wgrad(ws + w.dkv + d, a.lddkv, xn, d, Lg.wv, d, d, ns);
      if (qall) wg3[2] = gp_wgrad(ws + w.dkv + 2 * d, a.lddkv, xn, d, Lg.wq, d, d, ns);
      else wg3[2] = gp_wgrad(ws + w.dq, d, xn + (size_t)w.qpos * d, S * d, Lg.wq, d, d, l.n_in);
      // first layer, one query row per sequence: dQ.Wq is a [n_in, d] product whose rows join the big dX GEMM below
      // through its fan-in epilogue — computed here, before the weight gradients start competing for the CUs
      // (as a trailing accumulate-GEMM it took 26 us on the critical path under them)
      const bool q_via_res = !qall && i == 0;
      float* dxq = ws + w.dctx;                      // free again: the attention backward has consumed it
      if (q_via_res && !q_folded) {
        GemmProblem xq = gp(ws + w.dq, d, 0, Lp.wq, d, 1, dxq, d, l.n_in, d, d);
        xq.no_deep = 1;   // runs beside the side stream's weight gradients (at C5 the deep form waited 110 us for whole CUs)
        TRY(run1(xq, st));
      }
      // fork 2: they need the attention backward's dK / dV / dQ.  With the fused backward the side stream already
      // holds W2 / W1 / Wo (~90 us, the step's tail): the K/V/Q weight gradients then follow the dX GEMM on the MAIN
      // stream instead — one event less, and the side stream ends before the scatter does.
      // (round 2: W2 / W1 / Wo are ONE launch of ~45 us now, the side stream is free again when the attention backward
      // ends: the K / V / Q weight gradients go back to it, 0.3151 -> 0.3124 ms/step; PS_WG3_SIDE=0: main stream)
      // (later in round 2: with forks signalled by the next kernel the main stream lost its two bubbles and ENDED 30 us before
      // the side stream — score scatter 28 + W2/W1/Wo 45 + these 16 us; back on the main stream: 0.2861 -> 0.2801 ms/step)
      static const int wg3_side = ps_diag_int("PS_WG3_SIDE", -1);
      static const bool wg3_main_on = wg3_side >= 0 ? wg3_side == 0 : fork_by_kernel();
      const bool wg3_main = fused && wg3_main_on && ns <= 2 * M2;   // (review transformer: 78k K/V rows vs 1.5k replica rows -> side)
      // valid rows only: padded positions have exactly-zero dK / dV rows (their attention weights are 0), so the K/V
      // weight gradients (and the dX product below) run over the batch's row list instead of all n_in*S rows
      static const bool rows_on = ps_env_int("PS_NO_ROWLIST", 0) == 0;
      const bool listed = rows_on && rows_listed && sq1 && i == 0 && NL == 1 && !qall && w.qpos == 0 && w.vrows != 0 && l.n_in == B;
      const int32_t* vr = reinterpret_cast<const int32_t*>(ws + w.vrows);
      const int32_t* vc = reinterpret_cast<const int32_t*>(ws + w.vcount);
      if (listed) {
        wg3[0].ridx = vr; wg3[0].rcount = vc;
        wg3[1].ridx = vr; wg3[1].rcount = vc;
      }
      if (!wg3_main) {
        TRY(side_fork(st));
        TRY(side_run(wg3, 3, st));
      }
      if (wgrad_early && !fused) TRY(side_run(wgo, 1, st));
      // d xn = dK.Wk + dV.Wv (+ dQ.Wq)
      float* dxn = i == 0 ? ws + w.dx : ws + w.dxn;
      GemmProblem x = gp(ws + w.dkv, a.lddkv, 0, Lp.wk, d, 1, dxn, d, ns, d, qall ? 3 * d : 2 * d);
      x.kseg = d; x.Bseg[1] = Lp.wv; x.Bseg[2] = Lp.wq;
      if (i == 0) {   // + residual path of `out = dropout(context) + inputs`, summed over the replicas
        x.res.mode = RES_FANIN; x.res.ptr = ws + w.dy1; x.res.ld = d; x.res.Sq = l.Sq; x.res.fan = l.fan;
        x.res.S = S; x.res.qpos = w.qpos; res_finish(x.res);
        if (q_folded) {   // both partial rows already hold the replicas' fan-in sum: nothing left to walk here
          x.res.extra = ws + w.dln1; x.res.extra2 = (w1 && !(wf && attn_bwd_wf_two_partials(a))) ? nullptr : ws + w.dln1 + (size_t)l.n_in * d; x.res.extra_ld = d; x.res.ptr = nullptr;
        }
        else if (q_via_res) { x.res.extra = dxq; x.res.extra_ld = d; }
        if (presum && listed && q_via_res) {   // fan-in summed up front: one row per sequence beside the dQ.Wq row, nothing to walk
          float* fsum = ws + w.dln1;          // (free: the FF LayerNorm backward has consumed d ln1)
          TRY(launch_fanin_sum(ws + w.dy1, d, l.n_in, l.fan, d, fsum, st));
          x.res.ptr = nullptr; x.res.extra = dxq; x.res.extra2 = fsum; x.res.extra_ld = d;
        }
      }
      // (the dX product over the row list only when its fan-in residual is already folded: walking 21 replica rows per
      // query row in a third of the workgroups made it slower than the dense form — 144 vs 106 us at C5)
      if (listed && (q_folded || l.fan == 1 || (presum && q_via_res))) { x.ridx = vr; x.rcount = vc; }
      if (!dx_fused) TRY(run1(x, st));
      if (wg3_main) {
        static const bool wg3_last = ps_diag_int("PS_WG3_LAST", 1) != 0;
        if (wg3_last && g_wg3_defer_ok && i == 0) { for (int q = 0; q < 3; ++q) g_wg3_last[q] = wg3[q]; g_wg3_last_n = 3; }
        else TRY(run_wgrads(wg3, 3, st));
      }
      if (!qall && !q_via_res) {
        GemmProblem xq = gp(ws + w.dq, d, 0, Lp.wq, d, 1, dxn + (size_t)w.qpos * d, S * d, l.n_in, d, d);
        xq.accumulate = 1;
        TRY(run1(xq, st));
      }
    }
    if (i != 0) {   // pre-LayerNorm backward -> grad wrt the previous layer's output
      TRY(side_join(st));   // the next layer reuses the scratch buffers the side-stream GEMMs read
      PS_REQUIRE(Lg.ln_g && Lg.ln_b, "backward: layer %d null pre-LN gradient", i);
      LnBwdArgs n;
      memset(&n, 0, sizeof(n));
      n.dy = ws + w.dxn; n.lddy = d; n.x = xin; n.ldx = d; n.stats = ws + l.pre_stats; n.g = Lp.ln_g;
      n.rows = ns; n.d = d;
      n.res.mode = RES_FANIN; n.res.ptr = ws + w.dy1; n.res.ld = d; n.res.Sq = l.Sq; n.res.fan = l.fan;
      n.res.S = S; n.res.qpos = w.qpos; res_finish(n.res);
      n.dx = ws + w.dy2; n.lddx = d;
      if (drop) { n.out2 = ws + w.do2; n.drop2 = make_drop(D, PS_SITE_FF2(i - 1)); }
      n.colsum = G.layer[i - 1].b2; n.dgamma = Lg.ln_g; n.dbeta = Lg.ln_b;
      park_colsums(n, ws, w, fold);
      TRY(launch_ln_bwd(n, st));
    }
  }
  return PS_OK;
}

// --------------------------------------------------------------------- backward
static int tem_backward_impl(const PsTemDesc* desc, const PsTemTensors* params, const PsTemBatch* batch,
                             float* ws, const PsTemTensors* grads, float loss_scale, const float* loss_scale_dev,
                             ps_stream_t stream);
extern "C" int ps_tem_backward(const PsTemDesc* desc, const PsTemTensors* params, const PsTemBatch* batch,
                               float* ws, const PsTemTensors* grads, float loss_scale, const float* loss_scale_dev,
                               ps_stream_t stream) {
  const int rc = tem_backward_impl(desc, params, batch, ws, grads, loss_scale, loss_scale_dev, stream);
  if (rc != PS_OK) side_abort();          // never leave the side stream waiting behind a failed call
  return rc;
}
static int tem_backward_impl(const PsTemDesc* desc, const PsTemTensors* params, const PsTemBatch* batch,
                             float* ws, const PsTemTensors* grads, float loss_scale, const float* loss_scale_dev,
                             ps_stream_t stream) {
  PS_REQUIRE(desc && params && batch && ws && grads, "backward: null argument");
  PsTemDesc D = *desc;
  D.C = 0;
  Ws w;
  TRY(make_ws(D, w));
  hipStream_t st = (hipStream_t)stream;
  const PsTemTensors& P = *params;
  const PsTemTensors& G = *grads;
  const bool tem = D.model == PS_MODEL_TEM;
  const bool drop = D.training && D.dropout > 0.f;
  const int B = D.B, d = D.d, S = w.S, NL = tem ? D.n_layers : 0, F = D.F;
  const float* hist = D.sep_prod_emb ? P.hist_product_emb : P.product_emb;
  float* ghist = D.sep_prod_emb ? G.hist_product_emb : G.product_emb;
  PS_REQUIRE(G.product_emb && G.word_emb && G.word_bias && ghist && hist, "backward: null table gradient");
  PS_REQUIRE(!D.bias_product || G.product_bias, "backward: null product_bias gradient");

  // 1. loss + score backward: d enc, table-row scatter-adds
  ScoreArgs s;
  fill_score(D, P, *batch, ws, w, s);
  s.scale = loss_scale; s.scale_dev = loss_scale_dev; s.denc = ws + w.denc;
  s.g_product_emb = G.product_emb; s.g_word_emb = G.word_emb; s.g_product_bias = G.product_bias;
  s.g_word_bias = G.word_bias;
  // TEM with replicas: the encoder backward decides where the score backward runs (enc_layers_backward, score_on_side)
  static const bool score_side_on = ps_diag_int("PS_SCORE_BWD_MAIN", 0) == 0;
  const bool score_deferred = tem && NL > 0 && w.R > 1 && score_side_on;
  if (!score_deferred) TRY(launch_score_bwd(s, st));

  const float* wp_w[PS_WPLANES_MAX]; int wp_r[PS_WPLANES_MAX], wp_c[PS_WPLANES_MAX];
  const int wp_n = wplane_list(D, P, w, wp_w, wp_r, wp_c);
  WPlaneScope wplanes(st, wp_w, wp_r, wp_c, wp_n);             // the dX products read the TRANSPOSED weight planes (gemm.hip)
  ColFoldList fold;
  fold.n = 0;
  const float* dqe = ws + w.denc;   // grad wrt query_emb rows (QEM: enc IS query_emb)
  int lddqe = d;
  if (tem) {
    g_wg3_defer_ok = true; g_wg3_last_n = 0;
    const int rc_enc = enc_layers_backward(D, P, G, batch->u_item_idxs, nullptr, ws, w, st, &fold, score_deferred ? &s : nullptr,
                                           rows_list_ok(D));
    g_wg3_defer_ok = false;
    if (rc_enc != PS_OK) { g_wg3_last_n = 0; return rc_enc; }
    dqe = ws + w.dx;      // row 0 of each sequence is the query embedding
    lddqe = S * d;
  }

  // 4. query encoder backward + scatter to the word / history rows
  bool fw_by_gemm = false;
  EmbedBwdArgs e;
  memset(&e, 0, sizeof(e));
  e.B = B; e.Q = D.Q; e.L = D.L; e.S = S; e.d = d; e.P = D.product_size; e.V = D.vocab_size; e.tem = tem;
  e.qw = batch->query_word_idxs; e.ui = batch->u_item_idxs; e.dx = ws + w.dx;
  if (tem && g_dx_two_partials) e.dx2 = ws + w.dxn;      // the attention backward left d x as two partial rows per position
  g_dx_two_partials = false;
  e.drop_fs = make_drop(D, PS_SITE_FS);
  e.g_hist_tab = ghist; e.g_word_emb = G.word_emb;
  if (D.query_encoder == PS_QENC_FS) {
    PS_REQUIRE(G.fs_w && G.fs_b, "backward: null FS encoder gradient");
    e.fw_x = ws + w.qmean; e.g_fs_w = G.fs_w;   // f_W weight gradient rides in the scatter launch (extra workgroups)
    if (ps_fusion_enabled() && d <= 128) {
      // ... and so does the rest of the FS backward: tanh', d mean = dqpre . f_W (per-row mat-vec), bias gradient
      e.fsb_dqe = dqe; e.fsb_lddqe = lddqe; e.fsb_qe = ws + w.query_emb; e.fsb_w = P.fs_w; e.g_fs_b = G.fs_b;
      e.det_dm = ws + w.dqmean;                   // (deterministic mode only: launch_embed_scatter)
      // round 5: the f_W weight gradient as one more member of the weight-gradient GEMM launched behind this one, instead of 512
      // extra workgroups of the scatter launch looping over the batch (33 of its 40 us at C2: tools/scatter_parts.sh); the row
      // workgroups leave dqpre in the workspace for it and add the bias gradient themselves.  PS_FW_BY_GEMM=0: the riders.
      static const bool fw_gemm_on = ps_env_int("PS_FW_BY_GEMM", 1) != 0;
      if (fw_gemm_on && !ps_deterministic()) { e.fsb_dqpre_out = ws + w.dqpre; fw_by_gemm = true; }
    } else {
      TRY(launch_tanh_bwd(dqe, lddqe, ws + w.query_emb, ws + w.dqpre, G.fs_b, B, d, st));
      GemmProblem p = gp(ws + w.dqpre, d, 0, P.fs_w, d, 1, ws + w.dqmean, d, B, d, d);   // d mean = dqpre . f_W
      p.no_deep = 1;   // the tail of the main stream, beside the side stream's weight gradients: the 128-deep form's 133 KB of LDS per
                       // workgroup waits for whole CUs there (50 us for 0.13 GFLOP at the C5 shard, r04_c5_step_timeline.txt)
      TRY(run1(p, st));
      e.dqmean_d = ws + w.dqmean;
      e.fw_dy = ws + w.dqpre;
    }
  } else {
    // AVG encoder: query_emb == post-dropout mean; copy rows to a dense [B,d] buffer
    PS_CHECK_HIP(hipMemcpy2DAsync(ws + w.dqmean, sizeof(float) * d, dqe, sizeof(float) * lddqe, sizeof(float) * d, B,
                                  hipMemcpyDeviceToDevice, st));
    e.dqmean_d = ws + w.dqmean;
  }
  e.fold = fold;
  const int rc_sc = launch_embed_scatter(e, st);
  if (rc_sc != PS_OK) { g_wg3_last_n = 0; return rc_sc; }
  if (fw_by_gemm) {      // g_fs_w[o][i] += sum_b dqpre[b][o] * qmean[b][i]  (text_encoder.py:38)
    PS_REQUIRE(g_wg3_last_n <= 3, "backward: deferred weight-gradient group is full");
    g_wg3_last[g_wg3_last_n++] = gp_wgrad(ws + w.dqpre, d, ws + w.qmean, d, G.fs_w, d, d, B);
  }
  TRY(flush_wg3_last(st));
  TRY(side_join(st));
  return PS_OK;
}

// ------------------------------------------------------------------ graph-replayed training step (graph.h)
// Layout of the staging region: [step word | pad] then the six int64 index arrays, 16-byte aligned each.
struct StageLayout { uint32_t* step_word; int64_t* p[6]; int n[6]; };
static StageLayout stage_layout(const PsTemDesc& D, float* ws, const Ws& w) {
  StageLayout L;
  L.step_word = reinterpret_cast<uint32_t*>(ws + w.stage);
  int64_t* base = reinterpret_cast<int64_t*>(ws + w.stage + 4);
  const int n[6] = {D.B * D.Q, D.B * D.L, D.B, D.B * D.W, D.B * D.K, D.B * D.W * D.K};
  int64_t off = 0;
  for (int k = 0; k < 6; ++k) { L.p[k] = base + off; L.n[k] = n[k]; off += (n[k] + 1) & ~1; }
  return L;
}
static PsTemBatch staged_batch(const StageLayout& L) {
  PsTemBatch b;
  memset(&b, 0, sizeof(b));
  b.query_word_idxs = L.p[0]; b.u_item_idxs = L.p[1]; b.target_prod_idxs = L.p[2]; b.pos_iword_idxs = L.p[3];
  b.neg_item_idxs = L.p[4]; b.neg_word_idxs = L.p[5];
  return b;
}

extern "C" int ps_graph_replay_enabled(void) { return ps_graphs_enabled() ? 1 : 0; }

// where ps_tem_forward_step staged the call's index tensors (for an eager ps_tem_backward after a replayed forward)
extern "C" int ps_tem_staged_batch(const PsTemDesc* desc, float* workspace, PsTemBatch* out) {
  PS_REQUIRE(desc && workspace && out, "staged_batch: null argument");
  PsTemDesc D = *desc;
  D.C = 0;
  Ws w;
  TRY(make_ws(D, w));
  *out = staged_batch(stage_layout(D, workspace, w));
  return PS_OK;
}

// forward of one training step with the caller-varying inputs routed through the staging prologue, so that the launch
// sequence can be captured once per shape and replayed.  sampler_prob/alias non-null: negatives are drawn in the
// prologue (batch->neg_* ignored), else batch->neg_* are staged like the other index tensors.
static int forward_body(const PsTemDesc& D, const PsTemTensors& P, const StageArgs& sa, const PsTemBatch& Bs, float* ws,
                        const Ws& w, float* loss3, float* loss_acc, hipStream_t st) {
  TRY(launch_stage(sa, st));
  TRY(encode_forward(D, P, Bs, ws, w, st));
  ScoreArgs s;
  fill_score(D, P, Bs, ws, w, s);
  s.loss3 = loss3; s.loss_acc = loss_acc;
  TRY(launch_score_fwd(s, st));
  TRY(launch_loss(s, st));
  return PS_OK;
}

extern "C" int ps_tem_forward_step(const PsTemDesc* desc, const PsTemTensors* params, const PsTemBatch* batch,
                                   const float* sampler_prob, const int32_t* sampler_alias, float* workspace,
                                   float* loss3, float* loss_acc, ps_stream_t stream) {
  PS_REQUIRE(desc && params && batch && workspace && loss3, "forward_step: null argument");
  PsTemDesc D = *desc;
  D.C = 0;
  Ws w;
  TRY(make_ws(D, w));
  hipStream_t st = (hipStream_t)stream;
  const bool sampled = sampler_prob && sampler_alias;
  PS_REQUIRE(batch->query_word_idxs && batch->target_prod_idxs && (D.model != PS_MODEL_TEM || batch->u_item_idxs) &&
             (D.W == 0 || batch->pos_iword_idxs), "forward_step: null batch tensors");
  PS_REQUIRE(sampled || (batch->neg_item_idxs && (D.W == 0 || batch->neg_word_idxs)), "forward_step: no negatives");
  PS_REQUIRE(params->word_bias && (!D.bias_product || params->product_bias), "forward_step: null bias tensors");
  const StageLayout L = stage_layout(D, workspace, w);
  const PsTemBatch Bs = staged_batch(L);
  StageArgs sa;
  memset(&sa, 0, sizeof(sa));
  const int64_t* src[6] = {batch->query_word_idxs, D.model == PS_MODEL_TEM ? batch->u_item_idxs : nullptr,
                           batch->target_prod_idxs, D.W > 0 ? batch->pos_iword_idxs : nullptr,
                           sampled ? nullptr : batch->neg_item_idxs, (sampled || D.W == 0) ? nullptr : batch->neg_word_idxs};
  for (int k = 0; k < 6; ++k) { sa.src[k] = src[k]; sa.dst[k] = L.p[k]; sa.n[k] = L.n[k]; }
  sa.step = (uint32_t)D.step; sa.step_word = L.step_word;
  if (sampled) {
    sa.prob = sampler_prob; sa.alias = sampler_alias; sa.nitem = D.B * D.K; sa.nword = D.B * D.W * D.K;
    sa.P = D.product_size; sa.V = D.vocab_size;
    sa.k0 = (uint32_t)(D.seed & 0xffffffffu); sa.k1 = (uint32_t)(D.seed >> 32);
  }
  ps_step_ptr_slot() = L.step_word;               // every DropSpec built below reads the step from the workspace
  int rc = PS_OK;
  PsGraphEntry* e = nullptr;
  if (ps_graphs_enabled()) {
    PsTemDesc Dk = D;
    Dk.step = 0;
    uint64_t key = ps_fnv(PS_FNV0, "fwd", 3);
    key = ps_fnv(key, &Dk, sizeof(Dk)); key = ps_fnv(key, params, sizeof(*params));
    key = ps_fnv(key, &workspace, sizeof(workspace)); key = ps_fnv(key, &loss_acc, sizeof(loss_acc));
    key = ps_fnv(key, &sampler_prob, sizeof(sampler_prob)); key = ps_fnv(key, &sampler_alias, sizeof(sampler_alias));
    for (int k = 0; k < 6; ++k) { const int has = sa.src[k] != nullptr; key = ps_fnv(key, &has, sizeof(has)); }
    e = ps_graph_lookup(key);
  }
  if (e && e->state == 2) {
    ScoreArgs s;
    fill_score(D, *params, Bs, workspace, w, s);
    s.loss3 = loss3; s.loss_acc = loss_acc; s.loss_nblk = score_fwd_blocks(s);
    void* p0[1] = {(void*)&sa};
    void* p1[1] = {(void*)&s};
    rc = ps_graph_patch(e, 0, p0) || ps_graph_patch(e, 1, p1) || ps_graph_launch(e, st);
    if (rc) ps_set_error("forward_step: graph replay failed");
  } else {
    hipStream_t cap = (e && e->state == 1) ? ps_graph_begin() : nullptr;
    if (cap) {
      rc = forward_body(D, *params, sa, Bs, workspace, w, loss3, loss_acc, cap);
      const void* patch[2] = {stage_kernel_handle(), loss_kernel_handle()};
      if (ps_graph_end(cap, e, patch, 2) == PS_OK && rc == PS_OK) rc = ps_graph_launch(e, st);
      else { e->state = -1; rc = forward_body(D, *params, sa, Bs, workspace, w, loss3, loss_acc, st); }
    } else {
      if (e && e->state == 0) e->state = 1;
      rc = forward_body(D, *params, sa, Bs, workspace, w, loss3, loss_acc, st);
    }
  }
  ps_step_ptr_slot() = nullptr;
  return rc;
}

static int backward_body(const PsTemDesc& D, const PsTemTensors& P, const PsTemBatch& Bs, float* ws, const PsTemTensors& G,
                         float loss_scale, float* zero_ptr, int64_t zero_floats, hipStream_t st) {
  if (zero_ptr && zero_floats > 0) PS_CHECK_HIP(hipMemsetAsync(zero_ptr, 0, sizeof(float) * (size_t)zero_floats, st));
  return ps_tem_backward(&D, &P, &Bs, ws, &G, loss_scale, nullptr, st);
}

// backward of the step ps_tem_forward_step ran last on this workspace (its staged indices and step word are reused);
// zero_ptr/zero_floats: model.zero_grad() of the flat gradient buffer folded in as the first node (or null).
extern "C" int ps_tem_backward_step(const PsTemDesc* desc, const PsTemTensors* params, float* ws,
                                    const PsTemTensors* grads, float loss_scale, float* zero_ptr, int64_t zero_floats,
                                    ps_stream_t stream) {
  PS_REQUIRE(desc && params && ws && grads, "backward_step: null argument");
  PsTemDesc D = *desc;
  D.C = 0;
  Ws w;
  TRY(make_ws(D, w));
  hipStream_t st = (hipStream_t)stream;
  const StageLayout L = stage_layout(D, ws, w);
  const PsTemBatch Bs = staged_batch(L);
  ps_step_ptr_slot() = L.step_word;
  int rc = PS_OK;
  PsGraphEntry* e = nullptr;
  if (ps_graphs_enabled()) {
    PsTemDesc Dk = D;
    Dk.step = 0;
    uint64_t key = ps_fnv(PS_FNV0, "bwd", 3);
    key = ps_fnv(key, &Dk, sizeof(Dk)); key = ps_fnv(key, params, sizeof(*params)); key = ps_fnv(key, grads, sizeof(*grads));
    key = ps_fnv(key, &ws, sizeof(ws)); key = ps_fnv(key, &loss_scale, sizeof(loss_scale));
    key = ps_fnv(key, &zero_ptr, sizeof(zero_ptr)); key = ps_fnv(key, &zero_floats, sizeof(zero_floats));
    e = ps_graph_lookup(key);
  }
  if (e && e->state == 2) {
    rc = ps_graph_launch(e, st);
    if (rc) ps_set_error("backward_step: graph replay failed");
  } else {
    hipStream_t cap = (e && e->state == 1) ? ps_graph_begin() : nullptr;
    if (cap) {
      rc = backward_body(D, *params, Bs, ws, *grads, loss_scale, zero_ptr, zero_floats, cap);
      if (ps_graph_end(cap, e, nullptr, 0) == PS_OK && rc == PS_OK) rc = ps_graph_launch(e, st);
      else { e->state = -1; rc = backward_body(D, *params, Bs, ws, *grads, loss_scale, zero_ptr, zero_floats, st); }
    } else {
      if (e && e->state == 0) e->state = 1;
      rc = backward_body(D, *params, Bs, ws, *grads, loss_scale, zero_ptr, zero_floats, st);
    }
  }
  ps_step_ptr_slot() = nullptr;
  if (rc != PS_OK) side_abort();
  return rc;
}

extern "C" float ps_dropout_mult_host(const PsTemDesc* desc, uint32_t site, uint32_t row, uint32_t col) {
  PsTemDesc D = *desc;
  D.training = 1;
  DropSpec s = make_drop(D, site);
  if (s.thr == 0u) return 1.f;
  if (s.half) return drop_half(s, drop_call16(s, row, col >> 3, s.step), (int)(col & 7u));
  Philox4 r = philox4x32_10(col, row >> 2, s.site, s.step, s.k0, s.k1);
  uint32_t sel = row & 3u;
  uint32_t wv = sel == 0 ? r.x : (sel == 1 ? r.y : (sel == 2 ? r.z : r.w));
  return wv >= s.thr ? s.scale : 0.f;
}

// ------------------------------------------------------------ sampling / alias
extern "C" int ps_sample_negatives(const PsTemDesc* desc, const float* alias_prob, const int32_t* alias_idx,
                                   int64_t* neg_item_idxs, int64_t* neg_word_idxs, ps_stream_t stream) {
  PS_REQUIRE(desc && alias_prob && alias_idx && neg_item_idxs && neg_word_idxs, "sample: null argument");
  return launch_sample(*desc, alias_prob, alias_idx, neg_item_idxs, neg_word_idxs, (hipStream_t)stream);
}

// Vose's alias method over `n` outcomes with (unnormalised) weights dist[].
extern "C" int ps_build_alias_host(const double* dist, int64_t n, float* prob, int32_t* alias) {
  PS_REQUIRE(dist && prob && alias && n > 0 && n < (1ll << 31), "alias: bad argument");
  double sum = 0;
  for (int64_t i = 0; i < n; ++i) { PS_REQUIRE(dist[i] >= 0, "alias: negative weight"); sum += dist[i]; }
  PS_REQUIRE(sum > 0, "alias: zero total weight");
  double* q = new double[n];
  int32_t* small = new int32_t[n];
  int32_t* large = new int32_t[n];
  int64_t ns = 0, nl = 0;
  for (int64_t i = 0; i < n; ++i) {
    q[i] = dist[i] / sum * (double)n;
    alias[i] = (int32_t)i;
    if (q[i] < 1.0) small[ns++] = (int32_t)i; else large[nl++] = (int32_t)i;
  }
  while (ns > 0 && nl > 0) {
    int32_t s = small[--ns], l = large[--nl];
    prob[s] = (float)q[s];
    alias[s] = l;
    q[l] = (q[l] + q[s]) - 1.0;
    if (q[l] < 1.0) small[ns++] = l; else large[nl++] = l;
  }
  while (nl > 0) prob[large[--nl]] = 1.f;
  while (ns > 0) prob[small[--ns]] = 1.f;
  delete[] q; delete[] small; delete[] large;
  return PS_OK;
}
