// rank.hip — full-catalogue evaluation scorer with fused top-k.
//
// Reference: Trainer.test / validate / get_prod_scores / calc_metrics (trainer.py:125-226) with all products as
// candidates (test_candi_size < 1): the reference re-encodes the same (user, query) sequence for every chunk of 500
// candidates (item_transformer.py:111-146), copies every score to the host and argsorts there.  Here the sequence is
// encoded once (ps_tem_encode), scores[B, P] = enc · product_embᵀ is one fp32 MFMA GEMM per panel of the table, and
// the top-k and the target's rank come out of the same pass — nothing but [B, k] leaves the device.
//   select_kernel : one workgroup per (row, chunk of 8192 scores): MSB-first radix select of the k-th largest key in
//                   LDS histograms (4 passes of 8 bits), deterministic emit (block prefix scan, ties by lower index),
//                   rank counting against the target's score; repeated over the survivors until k remain, which a
//                   256-wide bitonic network sorts by (score desc, index asc).
#include "common.h"
#include <math.h>
#include <string.h>

extern "C" int ps_gemm_f32(const float* A, int lda, int ta, const float* Bm, int ldb, int tb, float* Cm, int ldc,
                           int M, int N, int K, const float* bias, float alpha, int accumulate, ps_stream_t stream);

#define SEL_EPT 32
#define SEL_CHUNK (256 * SEL_EPT)
#define SEL_KMAX 256

__device__ inline uint32_t f2key(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float key2f(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

struct SelArgs {
  const float* src_score;     // dense: S[b*src_ld + i]; survivors: score[b*src_ld + i]
  const int32_t* src_idx;     // null => dense (index = idx_base + i)
  int64_t src_ld;
  int n;                      // valid elements per row at this level
  int64_t idx_base;
  float* out_score; int32_t* out_idx;   // survivors: [B, out_ld], this launch writes segments chunk_off + blockIdx.x
  int64_t out_ld; int chunk_off;
  int k;
  int final;                  // sort and write top_idx/top_score instead
  int64_t* top_idx; float* top_score;
  const int64_t* target; const float* st; int32_t* rank;   // dense levels only (rank may be null)
};

__global__ __launch_bounds__(256) void select_kernel(const SelArgs a) {
  __shared__ int hist[256];
  __shared__ int scan[256];
  __shared__ uint32_t s_prefix;
  __shared__ int s_need;
  __shared__ unsigned long long srt[SEL_KMAX];
  const int tid = threadIdx.x, b = blockIdx.y, c = blockIdx.x;
  const int i0 = c * SEL_CHUNK + tid * SEL_EPT;
  uint32_t key[SEL_EPT];
  int32_t idx[SEL_EPT];
  const float* sp = a.src_score + (size_t)b * a.src_ld;
  const int32_t* ip = a.src_idx ? a.src_idx + (size_t)b * a.src_ld : nullptr;
  int cnt_gt_target = 0;
  const bool count = a.rank && !a.src_idx;
  const float st = count ? a.st[b] : 0.f;
  const int64_t tgt = count ? a.target[b] : -1;
#pragma unroll
  for (int j = 0; j < SEL_EPT; j += 4) {
    const int i = i0 + j;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i + 3 < a.n && ((((uintptr_t)(sp + i)) & 15) == 0)) v = *reinterpret_cast<const float4*>(sp + i);
    else {
      if (i < a.n) v.x = sp[i];
      if (i + 1 < a.n) v.y = sp[i + 1];
      if (i + 2 < a.n) v.z = sp[i + 2];
      if (i + 3 < a.n) v.w = sp[i + 3];
    }
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool ok = i + e < a.n;
      const int32_t gi = ok ? (ip ? ip[i + e] : (int32_t)(a.idx_base + i + e)) : -1;
      const bool live = ok && gi >= 0;
      key[j + e] = live ? f2key(vv[e]) : 0u;
      idx[j + e] = live ? gi : -1;
      if (count && live && gi != tgt) cnt_gt_target += (vv[e] > st) || (vv[e] == st && (int64_t)gi < tgt);
    }
  }
  if (count) {
    int cc = cnt_gt_target;
    for (int o = 32; o > 0; o >>= 1) cc += __shfl_down(cc, o, 64);
    if ((tid & 63) == 0 && cc) atomicAdd(&a.rank[b], cc);
  }
  // ---- radix select of the k-th largest key of this chunk
  if (tid == 0) { s_prefix = 0u; s_need = a.k; }
  uint32_t mask = 0u;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    hist[tid] = 0;
    __syncthreads();
    const uint32_t prefix = s_prefix;
#pragma unroll
    for (int j = 0; j < SEL_EPT; ++j)
      if ((key[j] & mask) == prefix) atomicAdd(&hist[(key[j] >> shift) & 255u], 1);
    __syncthreads();
    if (tid == 0) {
      int need = s_need, bin = 255;
      for (; bin > 0; --bin) {
        if (hist[bin] >= need) break;
        need -= hist[bin];
      }
      s_need = need;                       // still needed among keys sharing the extended prefix
      s_prefix = prefix | ((uint32_t)bin << shift);
    }
    mask |= 255u << shift;
    __syncthreads();
  }
  const uint32_t T = s_prefix;             // key of the k-th largest (0 when the chunk holds fewer than k live keys)
  const int need_eq = s_need;
  int ngt = 0, neq = 0;
#pragma unroll
  for (int j = 0; j < SEL_EPT; ++j) { ngt += key[j] > T; neq += key[j] == T; }
  scan[tid] = ngt | (neq << 16);
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int add = tid >= o ? scan[tid - o] : 0;
    __syncthreads();
    scan[tid] += add;
    __syncthreads();
  }
  const int incl = scan[tid], total = scan[255];
  const int tot_gt = total & 0xffff;
  int pos_gt = (incl & 0xffff) - ngt, pos_eq = (incl >> 16) - neq;
  if (a.final) {
    srt[tid] = 0ull;                       // key 0 / index pattern 0 sorts last
    __syncthreads();
  }
  float* os = a.final ? nullptr : a.out_score + (size_t)b * a.out_ld + (size_t)(a.chunk_off + c) * a.k;
  int32_t* oi = a.final ? nullptr : a.out_idx + (size_t)b * a.out_ld + (size_t)(a.chunk_off + c) * a.k;
#pragma unroll
  for (int j = 0; j < SEL_EPT; ++j) {
    int pos = -1;
    if (key[j] > T) pos = pos_gt++;
    else if (key[j] == T) { if (pos_eq < need_eq) pos = tot_gt + pos_eq; ++pos_eq; }
    if (pos >= 0 && pos < a.k) {
      if (a.final) srt[pos] = ((unsigned long long)key[j] << 32) | (uint32_t)(~(uint32_t)idx[j]);
      else { os[pos] = key2f(key[j]); oi[pos] = idx[j]; }
    }
  }
  if (!a.final) return;
  __syncthreads();
  // ---- final level: bitonic sort of SEL_KMAX composites, descending (score desc, index asc)
  for (int size = 2; size <= SEL_KMAX; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int p = tid ^ stride;
      if (p > tid) {
        const unsigned long long x = srt[tid], y = srt[p];
        const bool desc = (tid & size) == 0;
        if (desc ? (x < y) : (x > y)) { srt[tid] = y; srt[p] = x; }
      }
      __syncthreads();
    }
  }
  if (tid < a.k) {
    const unsigned long long v = srt[tid];
    const uint32_t kk = (uint32_t)(v >> 32);
    const int32_t gi = (int32_t)(~(uint32_t)(v & 0xffffffffu));
    const bool live = kk != 0u && gi >= 0;
    a.top_idx[(size_t)b * a.k + tid] = live ? (int64_t)gi : -1;
    a.top_score[(size_t)b * a.k + tid] = live ? key2f(kk) : -INFINITY;
  }
}

// gather the targets' rows (one wave per row), and afterwards pull the diagonal of q·Tᵀ (+bias) as the target score
__global__ __launch_bounds__(256) void rank_gather_kernel(const float* table, const int64_t* target, int64_t n_rows, int d,
                                                          float* out, int B) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  int64_t r = target[b];
  if (r < 0 || r >= n_rows) r = 0;
  for (int j = threadIdx.x & 63; j < d; j += 64) out[(size_t)b * d + j] = table[r * (int64_t)d + j];
}

__global__ void rank_diag_kernel(const float* tt, int B, const float* bias, const int64_t* target, int64_t n_rows,
                                 float* st, int32_t* rank) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int64_t r = target[b];
  const bool ok = r >= 0 && r < n_rows;
  st[b] = ok ? tt[(size_t)b * B + b] + (bias ? bias[r] : 0.f) : INFINITY;
  if (rank) rank[b] = ok ? 1 : 0;           // 0 = target not in the catalogue
}

static inline int64_t up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

struct RankPlan {
  int64_t panel;                 // table rows scored per GEMM
  int64_t n_panels, chunks1;     // level-1 chunks over all panels
  int64_t off_S, off_T, off_tt, off_st, off_c0s, off_c0i, off_c1s, off_c1i, total;
};

static int rank_plan(int B, int64_t n_rows, int d, int k, RankPlan* p) {
  int64_t panel = n_rows < ((int64_t)1 << 20) ? up(n_rows, 64) : ((int64_t)1 << 20);
  while ((int64_t)B * panel * 4 > ((int64_t)2 << 30) && panel > SEL_CHUNK) panel >>= 1;   // scores panel <= 2 GB
  p->panel = panel;
  p->n_panels = (n_rows + panel - 1) / panel;
  int64_t chunks1 = 0;
  for (int64_t q = 0; q < p->n_panels; ++q) {
    const int64_t w = (q + 1) * panel <= n_rows ? panel : n_rows - q * panel;
    chunks1 += (w + SEL_CHUNK - 1) / SEL_CHUNK;
  }
  p->chunks1 = chunks1;
  const int64_t cand0 = up((int64_t)B * chunks1 * k, 64);
  const int64_t chunks2 = (chunks1 * k + SEL_CHUNK - 1) / SEL_CHUNK;
  const int64_t cand1 = up((int64_t)B * chunks2 * k, 64);
  int64_t cur = 0;
  auto take = [&](int64_t bytes) { int64_t o = cur; cur += up(bytes, 256); return o; };
  p->off_S = take((int64_t)B * panel * 4);
  p->off_T = take((int64_t)B * d * 4);
  p->off_tt = take((int64_t)B * B * 4);
  p->off_st = take((int64_t)B * 4);
  p->off_c0s = take(cand0 * 4); p->off_c0i = take(cand0 * 4);
  p->off_c1s = take(cand1 * 4); p->off_c1i = take(cand1 * 4);
  p->total = cur;
  return PS_OK;
}

extern "C" int64_t ps_rank_scratch_bytes(int32_t B, int64_t n_rows, int32_t d, int32_t topk) {
  if (B < 1 || n_rows < 1 || d < 1 || topk < 1 || topk > SEL_KMAX) return -1;
  RankPlan p;
  rank_plan(B, n_rows, d, topk, &p);
  return p.total;
}

extern "C" int ps_rank_all(const float* q, int32_t B, int32_t d, const float* table, int64_t n_rows, const float* bias,
                           const int64_t* target, int32_t topk, int64_t* top_idx, float* top_score, int32_t* rank,
                           void* scratch, int64_t scratch_bytes, ps_stream_t stream) {
  PS_REQUIRE(q && table && top_idx && top_score && scratch, "rank_all: null argument");
  PS_REQUIRE(B >= 1 && d >= 1 && n_rows >= 1 && n_rows < ((int64_t)1 << 31), "rank_all: bad sizes");
  PS_REQUIRE(topk >= 1 && topk <= SEL_KMAX, "rank_all: topk must be 1..%d", SEL_KMAX);
  PS_REQUIRE(!rank || target, "rank_all: rank needs target");
  RankPlan p;
  rank_plan(B, n_rows, d, topk, &p);
  PS_REQUIRE(scratch_bytes >= p.total, "rank_all: scratch %lld < %lld bytes", (long long)scratch_bytes, (long long)p.total);
  PS_REQUIRE((((uintptr_t)scratch) & 255) == 0, "rank_all: scratch must be 256-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)scratch;
  float* S = (float*)(base + p.off_S);
  float* T = (float*)(base + p.off_T);
  float* tt = (float*)(base + p.off_tt);
  float* stv = (float*)(base + p.off_st);
  float* cs[2] = {(float*)(base + p.off_c0s), (float*)(base + p.off_c1s)};
  int32_t* ci[2] = {(int32_t*)(base + p.off_c0i), (int32_t*)(base + p.off_c1i)};
  if (target) {
    // the target's score through the SAME GEMM (bitwise equal to its entry of the score matrix)
    hipLaunchKernelGGL(rank_gather_kernel, dim3((B + 3) / 4), dim3(256), 0, st, table, target, n_rows, d, T, B);
    PS_LAUNCH_CHECK();
    int rc = ps_gemm_f32(q, d, 0, T, d, 0, tt, B, B, B, d, nullptr, 1.f, 0, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(rank_diag_kernel, dim3((B + 255) / 256), dim3(256), 0, st, tt, B, bias, target, n_rows, stv, rank);
    PS_LAUNCH_CHECK();
  }
  const bool single = p.chunks1 == 1;
  const int64_t ld0 = p.chunks1 * topk;
  int64_t chunk_off = 0;
  for (int64_t pi = 0; pi < p.n_panels; ++pi) {
    const int64_t r0 = pi * p.panel;
    const int64_t w = r0 + p.panel <= n_rows ? p.panel : n_rows - r0;
    int rc = ps_gemm_f32(q, d, 0, table + r0 * (int64_t)d, d, 0, S, (int)p.panel, B, (int)w, d, bias ? bias + r0 : nullptr,
                         1.f, 0, stream);
    if (rc) return rc;
    SelArgs a;
    memset(&a, 0, sizeof(a));
    a.src_score = S; a.src_idx = nullptr; a.src_ld = p.panel; a.n = (int)w; a.idx_base = r0;
    a.out_score = cs[0]; a.out_idx = ci[0]; a.out_ld = ld0; a.chunk_off = (int)chunk_off; a.k = topk;
    a.final = single; a.top_idx = top_idx; a.top_score = top_score;
    a.target = target; a.st = stv; a.rank = target ? rank : nullptr;
    const int nch = (int)((w + SEL_CHUNK - 1) / SEL_CHUNK);
    hipLaunchKernelGGL(select_kernel, dim3(nch, B), dim3(256), 0, st, a);
    PS_LAUNCH_CHECK();
    chunk_off += nch;
  }
  if (single) return PS_OK;
  // survivors: [B, chunks*k] -> repeat until one chunk remains, which the final pass sorts
  int cur = 0;
  int64_t n = ld0;
  while (true) {
    const int nch = (int)((n + SEL_CHUNK - 1) / SEL_CHUNK);
    SelArgs a;
    memset(&a, 0, sizeof(a));
    a.src_score = cs[cur]; a.src_idx = ci[cur]; a.src_ld = n; a.n = (int)n;
    a.out_score = cs[cur ^ 1]; a.out_idx = ci[cur ^ 1]; a.out_ld = (int64_t)nch * topk; a.chunk_off = 0; a.k = topk;
    a.final = nch == 1; a.top_idx = top_idx; a.top_score = top_score;
    hipLaunchKernelGGL(select_kernel, dim3(nch, B), dim3(256), 0, st, a);
    PS_LAUNCH_CHECK();
    if (nch == 1) break;
    n = (int64_t)nch * topk;
    cur ^= 1;
  }
  return PS_OK;
}
