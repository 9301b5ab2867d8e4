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
#include <stdlib.h>

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
  int32_t* cnt; int64_t cnt_ld;                              // [B, cnt_ld] rows ranked ahead of the target, per dense chunk
  int64_t id_mul, id_add;                                    // catalogue id of table row i = i * id_mul + id_add (a row shard: world, rank)
};

// k-th largest of the block's keys that are >= lo (N per lane): MSB-first radix select over LDS histograms.
// Returns its key in *T and how many keys equal to it are still needed in *need_eq.  All 256 lanes must call.
template <int N>
__device__ inline void block_kth_largest(const uint32_t (&key)[N], uint32_t lo, int k, int* hist, int* scan,
                                         uint32_t* s_prefix, int* s_need, uint32_t* T, int* need_eq) {
  const int tid = threadIdx.x;
  if (tid == 0) { *s_prefix = 0u; *s_need = k; }
  uint32_t mask = 0u;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    hist[tid] = 0;
    __syncthreads();
    const uint32_t prefix = *s_prefix;
#pragma unroll
    for (int j = 0; j < N; ++j)
      if (key[j] >= lo && (key[j] & mask) == prefix) atomicAdd(&hist[(key[j] >> shift) & 255u], 1);
    __syncthreads();
    // suffix sums over the 256 bins (S[b] = keys in bins >= b), all lanes: the bin holding the need-th largest key
    // is the one with S[b] >= need > S[b+1]
    const int need = *s_need;
    scan[tid] = hist[255 - tid];
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      const int add = tid >= o ? scan[tid - o] : 0;
      __syncthreads();
      scan[tid] += add;
      __syncthreads();
    }
    {
      const int b = 255 - tid;                          // this lane's bin; scan[tid] = S[b]
      const int above = tid > 0 ? scan[tid - 1] : 0;    // S[b+1]
      if ((scan[tid] >= need && above < need) || (b == 0 && scan[255] < need)) {
        *s_need = need - above;                          // still needed among keys sharing the extended prefix
        *s_prefix = prefix | ((uint32_t)b << shift);
      }
    }
    mask |= 255u << shift;
    __syncthreads();
  }
  *T = *s_prefix;
  *need_eq = *s_need;
  __syncthreads();
}

__global__ __launch_bounds__(256) void select_kernel(const SelArgs a) {
  __shared__ int hist[256];
  __shared__ int scan[256];
  __shared__ uint32_t s_prefix;
  __shared__ int s_need;
  __shared__ unsigned long long srt[SEL_KMAX];
  const int tid = threadIdx.x, b = blockIdx.y, c = blockIdx.x;
  const int i0 = c * SEL_CHUNK + 4 * tid;           // lane-interleaved 16-byte pieces: piece jj of this lane starts at
                                                    // i0 + 1024*jj, so every wave load covers 1 KB of consecutive scores
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
    const int i = i0 + 256 * j;
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
      const int64_t gid = (int64_t)gi * a.id_mul + a.id_add;
      if (count && live && gid != tgt) cnt_gt_target += (vv[e] > st) || (vv[e] == st && gid < tgt);
    }
  }
  if (count) {
    // one plain store per (row, chunk): thousands of atomics on the row's one rank word serialise at L2
    __shared__ int wsum[4];
    int cc = cnt_gt_target;
    for (int o = 32; o > 0; o >>= 1) cc += __shfl_down(cc, o, 64);
    if ((tid & 63) == 0) wsum[tid >> 6] = cc;
    __syncthreads();
    if (tid == 0) a.cnt[(size_t)b * a.cnt_ld + a.chunk_off + c] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
  }
  // ---- k-th largest key of this chunk.  First a floor: the k-th largest of the 256 per-lane maxima is <= it (k lanes
  // each hold a key at least that large), and only the ~k..2k keys above the floor then enter the LDS histograms —
  // instead of all 8192 (whose LDS atomics were most of this kernel).
  uint32_t T; int need_eq;
  {
    uint32_t mx[1] = {0u};
#pragma unroll
    for (int j = 0; j < SEL_EPT; ++j) mx[0] = key[j] > mx[0] ? key[j] : mx[0];
    uint32_t floor_key; int dummy;
    block_kth_largest<1>(mx, 0u, a.k, hist, scan, &s_prefix, &s_need, &floor_key, &dummy);
    block_kth_largest<SEL_EPT>(key, floor_key, a.k, hist, scan, &s_prefix, &s_need, &T, &need_eq);
  }
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
  // more keys equal to T than still needed: which of them survive must go by index (lower first), and lanes do not
  // hold consecutive indices — rare (exact score ties at the cut), settled by one lane after the parallel emit
  const bool tie_cut = (total >> 16) > need_eq;
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
    else if (key[j] == T && !tie_cut) { pos = tot_gt + pos_eq; ++pos_eq; }
    if (pos >= 0 && pos < a.k) {
      if (a.final) srt[pos] = ((unsigned long long)key[j] << 32) | (uint32_t)(~(uint32_t)idx[j]);
      else { os[pos] = key2f(key[j]); oi[pos] = idx[j]; }
    }
  }
  if (tie_cut && tid == 0) {
    int taken = 0;
    const int end = min(a.n, (c + 1) * SEL_CHUNK);
    for (int i = c * SEL_CHUNK; i < end && taken < need_eq; ++i) {
      const int32_t gi = ip ? ip[i] : (int32_t)(a.idx_base + i);
      if (gi < 0 || f2key(sp[i]) != T) continue;
      const int pos = tot_gt + taken++;
      if (pos < a.k) {
        if (a.final) srt[pos] = ((unsigned long long)T << 32) | (uint32_t)(~(uint32_t)gi);
        else { os[pos] = key2f(T); oi[pos] = gi; }
      }
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
    a.top_idx[(size_t)b * a.k + tid] = live ? (int64_t)gi * a.id_mul + a.id_add : -1;
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

// ---------------------------------------------------------------------------------------------------------------
// Skinny-M score GEMM for huge catalogues: S[B<=64, rows] = q . table^T streamed ONCE from HBM.  The tiled GEMM of
// gemm.hip pads M to 64 and gives every 64 table rows their own short-lived workgroup (1.4 TB/s at B = 24, 50 M rows).
// Here a persistent wave owns 32-row table tiles: coalesced 16-byte loads of the NEXT (tile, 128-deep k chunk) are in
// flight while the current chunk — transposed through a wave-private LDS slab into the [k][row+1] MFMA layout — is
// multiplied against q, which sits in LDS for the whole kernel.  Same v_mfma_f32_32x32x2_f32 chain in the same k
// order as gemm.hip, so the scores are bitwise those of ps_gemm_f32 (the target's score and the tests rely on it).
#define RS_KC 128                  // k depth of one streamed chunk
#define RS_LD 33
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MT>                  // MT = 1: B <= 32, 2: B <= 64
__global__ __launch_bounds__(256, 1) void rank_stream_kernel(const float* __restrict__ q, int B, int d,
                                                              const float* __restrict__ table, int64_t rows,
                                                              const float* __restrict__ bias, float* __restrict__ S,
                                                              int64_t ldc) {
  extern __shared__ float lds[];
  float* As = lds;                                       // [d][MT*32 + 1]   q, A-operand layout
  const int ALD = MT * 32 + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  float* Bw = lds + (size_t)d * ALD + (size_t)wave * RS_KC * RS_LD;   // this wave's [RS_KC][33] slab
  for (int i = tid; i < d * MT * 32; i += 256) {
    const int k = i / (MT * 32), m = i - k * (MT * 32);
    As[k * ALD + m] = m < B ? q[(size_t)m * d + k] : 0.f;
  }
  __syncthreads();
  const int nck = d / RS_KC;
  const int64_t ntile = (rows + 31) / 32;
  const int64_t nwave = (int64_t)gridDim.x * 4, w0 = (int64_t)blockIdx.x * 4 + wave;
  // lane -> (row within tile, k quad) of the 16 loads of one chunk: load u covers rows 2u, 2u+1
  const int lrow = lane >> 5, kq = lane & 31;
  float4 reg[16];
  auto issue = [&](int64_t tile, int ck) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int64_t r = tile * 32 + 2 * u + lrow;
      reg[u] = r < rows ? *reinterpret_cast<const float4*>(table + r * (int64_t)d + ck * RS_KC + 4 * kq)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (w0 < ntile) issue(w0, 0);
  for (int64_t tile = w0; tile < ntile; tile += nwave) {
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int ck = 0; ck < nck; ++ck) {
      // registers -> wave-private LDS slab (transpose to [k][row])
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int r = 2 * u + lrow;
        Bw[(4 * kq + 0) * RS_LD + r] = reg[u].x;
        Bw[(4 * kq + 1) * RS_LD + r] = reg[u].y;
        Bw[(4 * kq + 2) * RS_LD + r] = reg[u].z;
        Bw[(4 * kq + 3) * RS_LD + r] = reg[u].w;
      }
      // next chunk's loads go out before the multiply
      if (ck + 1 < nck) issue(tile, ck + 1);
      else if (tile + nwave < ntile) issue(tile + nwave, 0);
      __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0): the slab is written (wave-private: no barrier)
      __builtin_amdgcn_wave_barrier();
      const float* ab = As + (size_t)(ck * RS_KC + h) * ALD + l31;
      const float* bb = Bw + h * RS_LD + l31;
#pragma unroll
      for (int qd = 0; qd < RS_KC / 32; ++qd) {
        float av[MT][16], bv[16];
#pragma unroll
        for (int sft = 0; sft < 16; ++sft) {
          bv[sft] = bb[(32 * qd + 2 * sft) * RS_LD];
#pragma unroll
          for (int t = 0; t < MT; ++t) av[t][sft] = ab[(size_t)(32 * qd + 2 * sft) * ALD + 32 * t];
        }
#pragma unroll
        for (int sft = 0; sft < 16; ++sft)
#pragma unroll
          for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][sft], bv[sft], acc[t], 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();                   // every lane is done reading the slab before it is rewritten
    }
    const int64_t n = tile * 32 + l31;
    if (n < rows) {
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m < B) S[(size_t)m * ldc + n] = acc[t][r] + bv;
        }
    }
  }
}

static int launch_rank_stream(const float* q, int B, int d, const float* table, int64_t rows, const float* bias,
                              float* S, int64_t ldc, hipStream_t st) {
  const int MT = B <= 32 ? 1 : 2;
  const size_t lds = ((size_t)d * (MT * 32 + 1) + 4 * (size_t)RS_KC * RS_LD) * sizeof(float);
  static bool attr[2] = {false, false};
  const void* fn = MT == 1 ? reinterpret_cast<const void*>(rank_stream_kernel<1>)
                           : reinterpret_cast<const void*>(rank_stream_kernel<2>);
  if (!attr[MT - 1]) {
    PS_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - 512)));
    attr[MT - 1] = true;
  }
  PS_REQUIRE(lds <= 160 * 1024 - 512, "rank stream: %zu B of LDS", lds);
  int64_t tiles = (rows + 31) / 32;
  int blocks = (int)((tiles + 3) / 4 < 256 ? (tiles + 3) / 4 : 256);
  if (MT == 1) hipLaunchKernelGGL(rank_stream_kernel<1>, dim3(blocks), dim3(256), lds, st, q, B, d, table, rows, bias, S, ldc);
  else hipLaunchKernelGGL(rank_stream_kernel<2>, dim3(blocks), dim3(256), lds, st, q, B, d, table, rows, bias, S, ldc);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

// rank[b] = 1 + rows ranked ahead of the target over all dense chunks (0 stays 0: target not in the catalogue)
__global__ __launch_bounds__(256) void rank_sum_kernel(const int32_t* cnt, int64_t ld, int32_t* rank, int raw) {
  __shared__ int wsum[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  int s = 0;
  for (int64_t i = tid; i < ld; i += 256) s += cnt[(size_t)b * ld + i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((tid & 63) == 0) wsum[tid >> 6] = s;
  __syncthreads();
  if (tid == 0 && raw) rank[b] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);       // a shard's count of rows ahead
  else if (tid == 0 && rank[b] > 0) rank[b] = 1 + (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

static inline int64_t up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

struct RankPlan {
  int64_t panel;                 // table rows scored per GEMM
  int64_t n_panels, chunks1;     // level-1 chunks over all panels
  int64_t off_S, off_T, off_tt, off_st, off_c0s, off_c0i, off_c1s, off_c1i, off_cnt, total;
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
  p->off_cnt = take((int64_t)B * chunks1 * 4);
  p->total = cur;
  return PS_OK;
}

extern "C" int64_t ps_rank_scratch_bytes(int32_t B, int64_t n_rows, int32_t d, int32_t topk) {
  if (B < 1 || n_rows < 1 || d < 1 || topk < 1 || topk > SEL_KMAX) return -1;
  RankPlan p;
  rank_plan(B, n_rows, d, topk, &p);
  return p.total;
}

static int rank_all_impl(const float* q, int32_t B, int32_t d, const float* table, int64_t n_rows, const float* bias,
                         const int64_t* target, const float* ext_score, int64_t id_mul, int64_t id_add, int32_t topk,
                         int64_t* top_idx, float* top_score, int32_t* rank, void* scratch, int64_t scratch_bytes, ps_stream_t stream);
extern "C" int ps_rank_all(const float* q, int32_t B, int32_t d, const float* table, int64_t n_rows, const float* bias,
                           const int64_t* target, int32_t topk, int64_t* top_idx, float* top_score, int32_t* rank,
                           void* scratch, int64_t scratch_bytes, ps_stream_t stream) {
  return rank_all_impl(q, B, d, table, n_rows, bias, target, nullptr, 1, 0, topk, top_idx, top_score, rank, scratch, scratch_bytes,
                       stream);
}
// One rank's part of a full-catalogue ranking over a ROW-SHARDED table (sharded.py): table row i is catalogue id
// i * id_mul + id_add; target_score[b] is the target's score computed where its row lives; ahead[b] = rows of THIS shard ranked
// ahead of the target under (score desc, id asc), the target's own row excluded; top_idx carries catalogue ids.  The caller adds
// the shards' counts and merges their top-k lists.
extern "C" int ps_rank_shard(const float* q, int32_t B, int32_t d, const float* table, int64_t n_rows, const float* bias,
                             const int64_t* target, const float* target_score, int64_t id_mul, int64_t id_add, int32_t topk,
                             int64_t* top_idx, float* top_score, int32_t* ahead, void* scratch, int64_t scratch_bytes,
                             ps_stream_t stream) {
  PS_REQUIRE(target && target_score && ahead && id_mul >= 1 && id_add >= 0 && id_add < id_mul, "rank_shard: bad argument");
  return rank_all_impl(q, B, d, table, n_rows, bias, target, target_score, id_mul, id_add, topk, top_idx, top_score, ahead, scratch,
                       scratch_bytes, stream);
}
static int rank_all_impl(const float* q, int32_t B, int32_t d, const float* table, int64_t n_rows, const float* bias,
                         const int64_t* target, const float* ext_score, int64_t id_mul, int64_t id_add, int32_t topk,
                         int64_t* top_idx, float* top_score, int32_t* rank, void* scratch, int64_t scratch_bytes, ps_stream_t stream) {
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
  int32_t* cntb = (int32_t*)(base + p.off_cnt);
  if (ext_score) {
    stv = const_cast<float*>(ext_score);
  } else if (target) {
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
    int rc;
    const size_t stream_lds = ((size_t)d * ((B <= 32 ? 1 : 2) * 32 + 1) + 4 * (size_t)RS_KC * RS_LD) * sizeof(float);
    if (B <= 64 && d % RS_KC == 0 && stream_lds <= 160 * 1024 - 512 && n_rows >= ((int64_t)1 << 18))   // huge catalogue, few rows
      rc = launch_rank_stream(q, B, d, table + r0 * (int64_t)d, w, bias ? bias + r0 : nullptr, S, p.panel, st);
    else
      rc = ps_gemm_f32(q, d, 0, table + r0 * (int64_t)d, d, 0, S, (int)p.panel, B, (int)w, d, bias ? bias + r0 : nullptr,
                       1.f, 0, stream);
    if (rc) return rc;
    SelArgs a;
    memset(&a, 0, sizeof(a));
    a.src_score = S; a.src_idx = nullptr; a.src_ld = p.panel; a.n = (int)w; a.idx_base = r0;
    a.out_score = cs[0]; a.out_idx = ci[0]; a.out_ld = ld0; a.chunk_off = (int)chunk_off; a.k = topk;
    a.final = single; a.top_idx = top_idx; a.top_score = top_score;
    a.target = target; a.st = stv; a.rank = target ? rank : nullptr;
    a.cnt = cntb; a.cnt_ld = p.chunks1;
    a.id_mul = id_mul; a.id_add = id_add;
    const int nch = (int)((w + SEL_CHUNK - 1) / SEL_CHUNK);
    hipLaunchKernelGGL(select_kernel, dim3(nch, B), dim3(256), 0, st, a);
    PS_LAUNCH_CHECK();
    chunk_off += nch;
  }
  if (target && rank) {
    hipLaunchKernelGGL(rank_sum_kernel, dim3(B), dim3(256), 0, st, cntb, p.chunks1, rank, ext_score ? 1 : 0);
    PS_LAUNCH_CHECK();
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
    a.id_mul = id_mul; a.id_add = id_add;
    hipLaunchKernelGGL(select_kernel, dim3(nch, B), dim3(256), 0, st, a);
    PS_LAUNCH_CHECK();
    if (nch == 1) break;
    n = (int64_t)nch * topk;
    cur ^= 1;
  }
  return PS_OK;
}
