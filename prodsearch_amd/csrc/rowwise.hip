// rowwise.hip — the row-parallel kernels of the TEM step (gfx950): embedding
// gathers + paragraph-vector mean-pool, LayerNorm fwd/bwd, the tiny per-(sequence,
// head) attention, the gather+score kernel with its loss, their backward
// scatter-adds, and negative sampling.  All are HBM/L2-latency bound integer +
// fp32 work: rows are read with 16-byte lanes (coalesced 128 B..1 KiB per row),
// reductions are wavefront shuffles, nothing here is reshaped into a GEMM.
#include "rowwise.h"
#include <stdlib.h>

static inline int lpr_for(int d) {   // lanes per row for float4 lanes: pow2 >= d/4, <= 64
  int n = d / 4, l = 1;
  while (l < n && l < 64) l <<= 1;
  return l;
}

__device__ inline int64_t clamp_idx(int64_t i, int64_t hi) { return i < 0 ? hi : (i > hi ? hi : i); }

struct Task {
  const float* row; const float* vec; float bias; float* out;
  float* term; float tw;     // loss term = |tw| * softplus(sign(tw) * score): tw < 0 for the positive
  float lw;                  // weight of the term in the batch loss: item tasks +1; word tasks -(valid / #valid windows)
};
__device__ inline Task score_task(const ScoreArgs& a, int t) {
  Task k;
  const int K1 = a.K + 1;
  if (a.C > 0) {                                   // eval: candidates
    int b = fdiv(t, a.fC);
    int64_t idx = clamp_idx(a.candi[t], a.P);
    k.row = a.product_emb + (size_t)idx * a.d;
    k.vec = a.enc + (size_t)b * a.R * a.d;
    k.bias = a.bias_product ? a.product_bias[idx] : 0.f;
    k.out = a.item_scores + t;
    k.term = nullptr; k.tw = 0.f; k.lw = 0.f;
    return k;
  }
  const int nitem = a.B * K1;
  if (t < nitem) {
    int b = fdiv(t, a.fK1), j = t - b * K1;
    int64_t idx = clamp_idx(j == 0 ? a.target[b] : a.neg_items[(size_t)b * a.K + j - 1], a.P);
    k.row = a.product_emb + (size_t)idx * a.d;
    k.vec = a.enc + ((size_t)b * a.R + (a.R > 1 ? j : 0)) * a.d;
    k.bias = a.bias_product ? a.product_bias[idx] : 0.f;
    k.out = a.item_scores + t;
    k.term = a.item_terms + t;
    k.tw = j == 0 ? -(a.pos_weight ? (float)a.K : 1.f) : 1.f;
    k.lw = 1.f;
  } else {
    int u = t - nitem;
    int b = fdiv(u, a.fWK1), r = u - b * (a.W * K1);
    int w = fdiv(r, a.fK1), j = r - w * K1;
    int64_t widx;
    if (j == 0) widx = a.pos_words[(size_t)b * a.W + w];
    else if (a.samp_inline) {                      // the draw sample_kernel makes for this slot (same stream, same table)
      const uint32_t su = (uint32_t)(b * a.W * a.K + w * a.K + j - 1);
      Philox4 rr = philox4x32_10(su, 0u, PS_SITE_SAMPLE_WORD, a.samp_step, a.samp_k0, a.samp_k1);
      const int64_t i = (int64_t)(((uint64_t)rr.x * (uint64_t)a.V) >> 32);
      const float f = (float)(rr.y >> 8) * (1.0f / 16777216.0f);
      widx = f < a.samp_prob[i] ? i : (int64_t)a.samp_alias[i];
    } else widx = a.neg_words[(size_t)b * a.W * a.K + (size_t)w * a.K + j - 1];
    int64_t idx = clamp_idx(widx, a.V - 1);
    int64_t tb = clamp_idx(a.target[b], a.P);
    k.row = a.word_emb + (size_t)idx * a.d;
    k.vec = a.product_emb + (size_t)tb * a.d;
    k.bias = a.word_bias[idx];
    k.out = a.word_scores + u;
    k.term = a.word_terms + u;
    k.tw = j == 0 ? -1.f : 1.f;
    // masked mean over the window (get_vector_mean, item_transformer.py:281): padded slots drop out
    int cnt = 0;
    for (int ww = 0; ww < a.W; ++ww) cnt += a.pos_words[(size_t)b * a.W + ww] != a.V - 1;
    const bool valid = a.pos_words[(size_t)b * a.W + w] != a.V - 1;
    k.lw = valid ? -1.f / (float)cnt : -0.f;
  }
  return k;
}


// Word tasks of the loss as workgroups of the embed launch (ScoreArgs, folded form): 16 lanes per task (rows of d <= 512
// floats in 16-byte chunks), 16 tasks per pass, PS_WORD_TASKS_PER_WG tasks per workgroup; one {il} partial per workgroup.
__device__ inline void word_tasks_wg(const ScoreArgs& a, int wg) {
  __shared__ float wred[16];
  const int tid = threadIdx.x, grp = tid >> 4, c = tid & 15;
  const int nitem = a.B * (a.K + 1), nword = a.B * a.W * (a.K + 1);
  const int nch = a.d >> 2;
  float cil = 0.f;
  if (wg == 0 && tid < 18) a.ticket[tid] = 0u;         // the fused kernel's 9 arrival / partial-sum words, reset once per step
  for (int pass = 0; pass < PS_WORD_TASKS_PER_WG / 16; ++pass) {
    const int u = wg * PS_WORD_TASKS_PER_WG + pass * 16 + grp;
    float s = 0.f;
    Task k;
    k.out = nullptr;
    if (u < nword) {
      k = score_task(a, nitem + u);
      for (int cc = c; cc < nch; cc += 16) {
        const float4 r = *reinterpret_cast<const float4*>(k.row + 4 * cc);
        const float4 v = *reinterpret_cast<const float4*>(k.vec + 4 * cc);
        s += r.x * v.x + r.y * v.y + r.z * v.z + r.w * v.w;
      }
    }
    s = group_sum(s, 16);
    if (c == 0 && k.out) {
      const float sc = s + k.bias;
      *k.out = sc;
      const float term = fabsf(k.tw) * softplus_f(k.tw < 0.f ? -sc : sc);
      *k.term = term;
      cil -= term * k.lw;
    }
  }
  if (c == 0) wred[grp] = cil;
  __syncthreads();
  if (tid == 0) {
    float q = 0.f;
    for (int g2 = 0; g2 < 16; ++g2) q += wred[g2];
    a.word_blk[wg] = q;
  }
}

// =============================================================== embed forward
// Reference: word_embeddings(query_word_idxs) + get_vector_mean (item_transformer.py:449-450,
// text_encoder.py:6-16) + FS dropout (text_encoder.py:34-35); history gather, mask and
// positional add (item_transformer.py:452,466-471, transformer.py:77-81).
__global__ __launch_bounds__(256) void embed_fwd_kernel(const EmbedArgs a) {
  extern __shared__ float red[];   // [rpp][d]
  if ((int)blockIdx.x >= a.B && (int)blockIdx.x < a.B + a.samp_wgs) {      // sampling workgroups (see EmbedArgs::samp_*): same draws as sample_kernel
    const int t = ((int)blockIdx.x - a.B) * 256 + (int)threadIdx.x;
    if (t < a.samp_nitem) {
      Philox4 r = philox4x32_10((uint32_t)t, 0u, PS_SITE_SAMPLE_ITEM, a.samp_step, a.samp_k0, a.samp_k1);
      a.samp_items[t] = (int64_t)(((uint64_t)r.x * (uint64_t)a.P) >> 32);
    } else if (t < a.samp_nitem + a.samp_nword) {
      const int u = t - a.samp_nitem;
      Philox4 r = philox4x32_10((uint32_t)u, 0u, PS_SITE_SAMPLE_WORD, a.samp_step, a.samp_k0, a.samp_k1);
      const int64_t i = (int64_t)(((uint64_t)r.x * (uint64_t)a.V) >> 32);
      const float f = (float)(r.y >> 8) * (1.0f / 16777216.0f);
      a.samp_words[u] = f < a.samp_prob[i] ? i : (int64_t)a.samp_alias[i];
    }
    return;
  }
  if ((int)blockIdx.x >= a.B + a.samp_wgs + a.list_wgs + a.word_wgs + a.split_wgs) {   // EmbedArgs::zero_i32
    const int i = (((int)blockIdx.x - a.B - a.samp_wgs - a.list_wgs - a.word_wgs - a.split_wgs) * 256 + (int)threadIdx.x) * 4;
    if (i + 3 < a.zero_n) *reinterpret_cast<int4*>(a.zero_i32 + i) = make_int4(0, 0, 0, 0);
    else for (int j = i; j < a.zero_n; ++j) a.zero_i32[j] = 0;
    return;
  }
  if ((int)blockIdx.x >= a.B + a.samp_wgs + a.list_wgs + a.word_wgs) {   // weight re-split (EmbedArgs::split; rowwise.h, wsplit_chunk)
    const int q = ((int)blockIdx.x - a.B - a.samp_wgs - a.list_wgs - a.word_wgs) * 256 + (int)threadIdx.x;
    const int n0 = wsplit_chunks(a.split, 0);
    if (q < n0) wsplit_chunk(a.split, 0, q);
    else if (!a.split_fwd_only) wsplit_chunk(a.split, 1, q - n0);
    return;
  }
  if ((int)blockIdx.x >= a.B + a.samp_wgs + a.list_wgs) {       // word tasks of the loss (EmbedArgs::fold_words)
    word_tasks_wg(a.sc, (int)blockIdx.x - a.B - a.samp_wgs - a.list_wgs);
    return;
  }
  if ((int)blockIdx.x >= a.B + a.samp_wgs) {
    // valid-row list (EmbedArgs::vrows), one workgroup per sequence beside the gathering ones: this sequence's rows
    // start behind the valid rows of all earlier sequences, counted here (b*L coalesced int64 reads, L2 hits) — no
    // scan kernel, no host round trip, nothing added to the gather's own chain
    __shared__ int vred[4];
    const int b = (int)blockIdx.x - a.B - a.samp_wgs, tid = threadIdx.x;
    // (round 5: eight entries per lane and trip, four 16-byte loads in flight (eight measured the same) — as a loop of one 8-byte load per trip, each waited for,
    //  the last sequences' workgroups took 30 serial round trips at C2 and 80 at the C5 shard: the longest workgroups of the launch)
    int cntp = 0;
    {
      const int n = b * a.L;
      const bool al16 = (reinterpret_cast<uintptr_t>(a.ui) & 15) == 0;
      if (al16) {
        const longlong2* u2 = reinterpret_cast<const longlong2*>(a.ui);
        const int n2 = n >> 1;                                  // pairs; an odd last entry is counted below
        for (int i = tid; i < n2; i += 4 * 256) {
          longlong2 v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { const int j = i + 256 * u; v[u] = u2[j < n2 ? j : 0]; }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int j = i + 256 * u;
            cntp += (j < n2 && v[u].x != a.P) + (j < n2 && v[u].y != a.P);
          }
        }
        if ((n & 1) && tid == 0) cntp += (a.ui[n - 1] != a.P);
      } else {
        for (int i = tid; i < n; i += 256) cntp += (a.ui[i] != a.P);
      }
    }
    const bool mine = tid < a.L && a.ui[(size_t)b * a.L + tid] != a.P;       // L <= 64 (validated): wave 0 holds the row
    const unsigned long long vm = __ballot(mine);
    cntp = (int)wave_sum((float)cntp);                        // < 2^24: exact in fp32
    if ((tid & 63) == 0) vred[tid >> 6] = cntp;
    __syncthreads();
    const int off = vred[0] + vred[1] + vred[2] + vred[3] + b;   // + one query row per earlier sequence
    if (tid == 0) a.vrows[off] = b * a.S;
    if (mine) a.vrows[off + 1 + __popcll(vm & ((1ull << tid) - 1ull))] = b * a.S + 1 + tid;
    if (tid == 0 && b == a.B - 1) *a.vcount = off + 1 + __popcll(vm);
    return;
  }
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b == 0 && tid == 0 && a.clear_word) { a.clear_word[0] = 0u; a.clear_word[1] = 0u; a.clear_word[2] = 0u; a.clear_word[3] = 0u; }
  const int d = a.d, nchunk = d >> 2;
  const int rpp = 256 / nchunk;
  const int rg = tid / nchunk, c = tid - rg * nchunk;
  const bool active = rg < rpp;
  const bool fuse_fs = a.fs && a.fs_w;
  float* mean_s = red + (size_t)rpp * d;      // [d] post-dropout mean, [d] projection output (fused FS projection only)
  // d == 128: the 16 weight rows this lane group will multiply are fetched NOW (16 B per lane, row o = rg + 8u), so
  // their L2 round trips run under the query-word gather instead of behind it
  const bool fs_pre = fuse_fs && d == 128;
  float4 wpre[16];
  float4 bpre = make_float4(0.f, 0.f, 0.f, 0.f);
  if (fs_pre) {
#pragma unroll
    for (int u = 0; u < 16; ++u) wpre[u] = *reinterpret_cast<const float4*>(a.fs_w + (size_t)(rg + 8 * u) * 128 + 4 * c);
  }
  if (fuse_fs && tid < nchunk) bpre = *reinterpret_cast<const float4*>(a.fs_b + 4 * c);
  const int64_t wpad = a.V - 1;
  int cnt = 0;
  for (int q = 0; q < a.Q; ++q) cnt += (a.qw[(size_t)b * a.Q + q] != wpad);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (active) {
    for (int q = rg; q < a.Q; q += rpp) {
      int64_t idx = a.qw[(size_t)b * a.Q + q];
      if (idx != wpad && idx >= 0 && idx < a.V) {
        float4 v = *reinterpret_cast<const float4*>(a.word_emb + (size_t)idx * d + 4 * c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    *reinterpret_cast<float4*>(red + (size_t)rg * d + 4 * c) = acc;
  }
  __syncthreads();
  if (tid < nchunk) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < rpp; ++r) {
      float4 v = *reinterpret_cast<const float4*>(red + (size_t)r * d + 4 * c);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float inv = 1.f / (float)(cnt > 0 ? cnt : 1);
    float m[4] = {s.x * inv, s.y * inv, s.z * inv, s.w * inv};
#pragma unroll
    for (int e = 0; e < 4; ++e) m[e] *= drop_mult(a.drop_fs, (uint32_t)b, (uint32_t)(4 * c + e));
    float4 md = make_float4(m[0], m[1], m[2], m[3]);
    *reinterpret_cast<float4*>(a.qmean_d + (size_t)b * d + 4 * c) = md;
    if (fuse_fs) *reinterpret_cast<float4*>(mean_s + 4 * c) = md;
    if (!a.fs) {
      *reinterpret_cast<float4*>(a.query_emb + (size_t)b * d + 4 * c) = md;
      if (a.tem) {
        float4 o = md;
        if (a.use_pos) {
          float4 p = *reinterpret_cast<const float4*>(a.pe + 4 * c);
          o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
        }
        *reinterpret_cast<float4*>(a.x + (size_t)b * a.S * d + 4 * c) = o;
      }
    }
  }
  if (a.tem && active) {
    for (int l = rg; l < a.L; l += rpp) {
      int64_t idx = a.ui[(size_t)b * a.L + l];
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx != a.P && idx >= 0 && idx < a.P)
        v = *reinterpret_cast<const float4*>(a.hist_tab + (size_t)idx * d + 4 * c);
      if (a.use_pos) {
        float4 p = *reinterpret_cast<const float4*>(a.pe + (size_t)(1 + l) * d + 4 * c);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
      }
      *reinterpret_cast<float4*>(a.x + ((size_t)b * a.S + 1 + l) * d + 4 * c) = v;
    }
  }
  if (!fuse_fs) return;
  // FS projection of this row: out[o] = tanh(sum_i fs_w[o][i] * mean[i] + fs_b[o]).  A group of lpr lanes (16 B per
  // lane, one coalesced weight row per load) owns the outputs o = grp, grp + ngrp, ...; 4 weight rows in flight.
  __syncthreads();
  float* out_s = mean_s + d;
  if (fs_pre) {
    const float4 mv = *reinterpret_cast<const float4*>(mean_s + 4 * c);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const float4 wv = wpre[u];
      const float s = half_sum_last(wv.x * mv.x + wv.y * mv.y + wv.z * mv.z + wv.w * mv.w);
      if (c == 31) out_s[rg + 8 * u] = s;
    }
  } else {
    int lpr = 1;
    while (lpr < nchunk && lpr < 64) lpr <<= 1;
    const int ngrp = 256 / lpr, grp = tid / lpr, lc = tid - grp * lpr;
    for (int o0 = grp; o0 < d; o0 += 4 * ngrp) {
      float part[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int o = o0 + u * ngrp;
        float s = 0.f;
        if (o < d)
          for (int cc = lc; cc < nchunk; cc += lpr) {
            const float4 wv = *reinterpret_cast<const float4*>(a.fs_w + (size_t)o * d + 4 * cc);
            const float4 mv = *reinterpret_cast<const float4*>(mean_s + 4 * cc);
            s += wv.x * mv.x + wv.y * mv.y + wv.z * mv.z + wv.w * mv.w;
          }
        part[u] = s;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int o = o0 + u * ngrp;
        const float s = lpr == 32 ? half_sum_last(part[u]) : group_sum(part[u], lpr);
        if (lc == lpr - 1 && o < d) out_s[o] = s;
      }
    }
  }
  __syncthreads();
  if (tid < nchunk) {
    float4 y = *reinterpret_cast<const float4*>(out_s + 4 * c);
    y.x = tanh_fast(y.x + bpre.x); y.y = tanh_fast(y.y + bpre.y); y.z = tanh_fast(y.z + bpre.z); y.w = tanh_fast(y.w + bpre.w);
    *reinterpret_cast<float4*>(a.query_emb + (size_t)b * d + 4 * c) = y;
    if (a.tem) {
      if (a.use_pos) {
        float4 p = *reinterpret_cast<const float4*>(a.pe + 4 * c);
        y.x += p.x; y.y += p.y; y.z += p.z; y.w += p.w;
      }
      *reinterpret_cast<float4*>(a.x + (size_t)b * a.S * d + 4 * c) = y;
    }
  }
}

int launch_embed_fwd(const EmbedArgs& a, hipStream_t st) {
  PS_REQUIRE(a.d % 4 == 0 && a.d <= 1024, "embed: d=%d unsupported", a.d);
  int rpp = 256 / (a.d / 4);
  const int nsamp = a.samp_prob ? ps_cdiv(a.samp_nitem + a.samp_nword, 256) : 0;
  const int nlist = (a.vrows && a.tem && a.L <= 64) ? a.B : 0;
  EmbedArgs b = a;
  b.samp_wgs = nsamp;
  b.list_wgs = nlist;
  if (!nlist) b.vrows = nullptr;
  b.word_wgs = a.fold_words ? a.sc.word_nblk : 0;
  b.split_wgs = 0;
  if (a.split.on) {
    const WSplit& W = a.split;
    PS_REQUIRE(W.w[0] && W.w[1] && W.w[2] && W.fwd_wo && W.fwd_ff && W.bwd_ff && W.bwd_wo, "embed: weight split: null pointer");
    PS_REQUIRE(W.rows[0] == 128 && W.cols[0] == 128 && W.cols[1] == 128 && W.rows[2] == 128 && W.rows[1] == W.cols[2] &&
               W.rows[1] % 256 == 0, "embed: weight split: shapes [%d,%d] [%d,%d] [%d,%d]", W.rows[0], W.cols[0], W.rows[1],
               W.cols[1], W.rows[2], W.cols[2]);
    PS_REQUIRE(!W.wkv[0] || (W.wkv[1] && W.fwd_kv), "embed: weight split: K / V weights without their stream");
    PS_REQUIRE(!W.wkv[0] || W.bwd_kv, "embed: weight split: K / V weights without their backward stream");
    b.split_wgs = ps_cdiv(wsplit_chunks(W, 0) + (a.split_fwd_only ? 0 : wsplit_chunks(W, 1)), 256);   // one thread per 16-byte chunk
  }
  PS_REQUIRE(!a.fold_words || (a.sc.word_blk && a.sc.ticket && a.sc.d <= 512), "embed: folded word tasks need their buffers");
  b.zero_wgs = a.zero_i32 && a.zero_n > 0 ? ps_cdiv(a.zero_n, 1024) : 0;
  PS_REQUIRE(!b.zero_wgs || ((uintptr_t)a.zero_i32 & 15) == 0, "embed: zero_i32 must be 16-byte aligned");
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(a.B + nsamp + nlist + b.word_wgs + b.split_wgs + b.zero_wgs), dim3(256),
                     (size_t)(rpp + 2) * a.d * sizeof(float), st, b);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

// ================================================================== LayerNorm
// nn.LayerNorm(d, eps=1e-6): transformer.py:44,68,86 ; neural.py:25.  One wave per row.
#define LN_MAXI 8   // d <= 512
template <int DPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnFwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nw = (gridDim.x * blockDim.x) >> 6;
  const float invd = 1.f / (float)a.d;
  for (int row = wave; row < a.rows; row += nw) {
    float v[DPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
      int col = lane + 64 * i;
      v[i] = col < a.d ? a.x[(size_t)row * a.ldx + col] : 0.f;
      s += v[i];
    }
    const float mean = wave_sum(s) * invd;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
      int col = lane + 64 * i;
      float t = col < a.d ? v[i] - mean : 0.f;
      q += t * t;
    }
    const float rstd = 1.f / sqrtf(wave_sum(q) * invd + a.eps);
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
      int col = lane + 64 * i;
      if (col < a.d) a.y[(size_t)row * a.ldy + col] = (v[i] - mean) * rstd * a.g[col] + a.b[col];
    }
    if (lane == 0) { a.stats[2 * (size_t)row] = mean; a.stats[2 * (size_t)row + 1] = rstd; }
  }
}

int launch_ln_fwd(const LnFwdArgs& a, hipStream_t st) {
  PS_REQUIRE(a.d <= 64 * LN_MAXI, "layernorm: d=%d > %d", a.d, 64 * LN_MAXI);
  int blocks = ps_cdiv(a.rows, 4);
  if (blocks > 4096) blocks = 4096;
  const int dpl = ps_cdiv(a.d, 64);
  if (dpl <= 1) hipLaunchKernelGGL(ln_fwd_kernel<1>, dim3(blocks), dim3(256), 0, st, a);
  else if (dpl <= 2) hipLaunchKernelGGL(ln_fwd_kernel<2>, dim3(blocks), dim3(256), 0, st, a);
  else if (dpl <= 4) hipLaunchKernelGGL(ln_fwd_kernel<4>, dim3(blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(ln_fwd_kernel<8>, dim3(blocks), dim3(256), 0, st, a);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

// Backward: 16 waves per workgroup so the gamma/beta/bias column sums are combined in LDS
// before they reach the (contended) fp32 atomics: one atomic per column per workgroup.
#define LNB_WAVES 16
// EXACT (d == 64 * DPL, every hot shape): no per-column conditions at all — a row's loads (x, dy, the direct residual) are
// issued together and its stores follow one another.  With the `col < d` tests of the general form every column group was
// its own branch: the compiler waits for everything in flight at each join, so a d = 256 row cost eight dependent round
// trips (loads) plus the acknowledgement of every store before the next — 44 us per launch at the C5 shape, x 2.
template <int DPL, int EXACT>
__global__ __launch_bounds__(64 * LNB_WAVES) void ln_bwd_kernel(const LnBwdArgs a) {
  extern __shared__ float sh[];      // [3][LNB_WAVES][64*DPL]
  const int W = 64 * DPL;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nw = (gridDim.x * blockDim.x) >> 6;
  const float invd = 1.f / (float)a.d;
  float ag[DPL], ab[DPL], ac[DPL];
#pragma unroll
  for (int i = 0; i < DPL; ++i) { ag[i] = 0.f; ab[i] = 0.f; ac[i] = 0.f; }
  if (EXACT) {
    float gv[DPL];
#pragma unroll
    for (int i = 0; i < DPL; ++i) gv[i] = a.g[lane + 64 * i];
    const int rmode = a.res.mode;
    DropSpec d2 = a.drop2;                          // the step word is read once, not per element
    d2.step = drop_step(a.drop2); d2.step_ptr = nullptr;
    for (int row = wave; row < a.rows; row += nw) {
      const float mean = a.stats[2 * (size_t)row], rstd = a.stats[2 * (size_t)row + 1];
      float xv[DPL], dyv[DPL], rres[DPL];
#pragma unroll
      for (int i = 0; i < DPL; ++i) {
        xv[i] = a.x[(size_t)row * a.ldx + lane + 64 * i];
        dyv[i] = a.dy[(size_t)row * a.lddy + lane + 64 * i];
        rres[i] = 0.f;
      }
      if (rmode == RES_DIRECT) {
#pragma unroll
        for (int i = 0; i < DPL; ++i) rres[i] = a.res.ptr[(size_t)row * a.res.ld + lane + 64 * i];
      } else if (rmode != RES_NONE) {
#pragma unroll
        for (int i = 0; i < DPL; ++i) rres[i] = res_value(a.res, row, lane + 64 * i);
      }
      float xh[DPL], dxh[DPL];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < DPL; ++i) {
        xh[i] = (xv[i] - mean) * rstd;
        dxh[i] = dyv[i] * gv[i];
        s1 += dxh[i];
        s2 += dxh[i] * xh[i];
      }
      s1 = wave_sum(s1) * invd;
      s2 = wave_sum(s2) * invd;
      float dx[DPL];
#pragma unroll
      for (int i = 0; i < DPL; ++i) {
        dx[i] = rstd * (dxh[i] - s1 - xh[i] * s2);
        if (rmode != RES_NONE) dx[i] += rres[i];
        a.dx[(size_t)row * a.lddx + lane + 64 * i] = dx[i];
        ag[i] += dyv[i] * xh[i];
        ab[i] += dyv[i];
      }
      if (a.out2) {
#pragma unroll
        for (int i = 0; i < DPL; ++i) {
          const float v2 = dx[i] * drop_mult(d2, (uint32_t)row, (uint32_t)(lane + 64 * i));
          a.out2[(size_t)row * a.d + lane + 64 * i] = v2;
          ac[i] += v2;
        }
      } else {
#pragma unroll
        for (int i = 0; i < DPL; ++i) ac[i] += dx[i];
      }
    }
  } else
  for (int row = wave; row < a.rows; row += nw) {
    const float mean = a.stats[2 * (size_t)row], rstd = a.stats[2 * (size_t)row + 1];
    float xh[DPL], dxh[DPL], dyv[DPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
      int col = lane + 64 * i;
      if (col < a.d) {
        float x = a.x[(size_t)row * a.ldx + col];
        dyv[i] = a.dy[(size_t)row * a.lddy + col];
        xh[i] = (x - mean) * rstd;
        dxh[i] = dyv[i] * a.g[col];
      } else { xh[i] = 0.f; dxh[i] = 0.f; dyv[i] = 0.f; }
      s1 += dxh[i];
      s2 += dxh[i] * xh[i];
    }
    s1 = wave_sum(s1) * invd;
    s2 = wave_sum(s2) * invd;
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
      int col = lane + 64 * i;
      if (col < a.d) {
        float dx = rstd * (dxh[i] - s1 - xh[i] * s2);
        if (a.res.mode != RES_NONE) dx += res_value(a.res, row, col);
        a.dx[(size_t)row * a.lddx + col] = dx;
        float v2 = dx;
        if (a.out2) {
          v2 = dx * drop_mult(a.drop2, (uint32_t)row, (uint32_t)col);
          a.out2[(size_t)row * a.d + col] = v2;
        }
        ag[i] += dyv[i] * xh[i];
        ab[i] += dyv[i];
        ac[i] += v2;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < DPL; ++i) {
    sh[(0 * LNB_WAVES + wv) * W + lane + 64 * i] = ag[i];
    sh[(1 * LNB_WAVES + wv) * W + lane + 64 * i] = ab[i];
    sh[(2 * LNB_WAVES + wv) * W + lane + 64 * i] = ac[i];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 3 * a.d; t += blockDim.x) {
    const int which = t / a.d, col = t - which * a.d;
    float* dst = which == 0 ? a.dgamma : (which == 1 ? a.dbeta : a.colsum);
    if (!dst && !a.partial) continue;
    float s = 0.f;
    for (int w = 0; w < LNB_WAVES; ++w) s += sh[(which * LNB_WAVES + w) * W + col];
    if (a.partial) a.partial[((size_t)blockIdx.x * 3 + which) * a.d + col] = s;
    else atomicAdd(&dst[col], s);
  }
}

// d == 256, 16-byte-aligned operands, no gathered residual (the d = 256 shard step's two LayerNorm backwards over 21,504 rows: 35 us
// each in the form above — one row per wave in flight, 12 four-byte loads per lane and row, 128 VGPRs = 16 waves per CU, 48 KB in
// flight per CU against a latency-bandwidth product of ~64 KB).  Here a lane owns 4 CONSECUTIVE columns: one 16-byte load per operand
// and row, and TWO rows per wave in flight (the second one clamped, not skipped, at the end of the range).
__global__ __launch_bounds__(64 * LNB_WAVES) void ln_bwd_v4_kernel(const LnBwdArgs a) {
  extern __shared__ float sh[];      // [3][LNB_WAVES][256]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nw = (gridDim.x * blockDim.x) >> 6;
  const float invd = 1.f / 256.f;
  const float4 gv = reinterpret_cast<const float4*>(a.g)[lane];
  const bool has_res = a.res.mode == RES_DIRECT;
  DropSpec d2 = a.drop2;
  d2.step = drop_step(a.drop2); d2.step_ptr = nullptr;
  float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = ag, ac = ag;
  for (int row0 = wave; row0 < a.rows; row0 += 2 * nw) {
    const bool two = row0 + nw < a.rows;                            // wave-uniform
    const int rw[2] = {row0, two ? row0 + nw : row0};
    float4 xv[2], dyv[2], rr[2];
    float2 st[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      st[u] = reinterpret_cast<const float2*>(a.stats)[rw[u]];
      xv[u] = *reinterpret_cast<const float4*>(a.x + (size_t)rw[u] * a.ldx + 4 * lane);
      dyv[u] = *reinterpret_cast<const float4*>(a.dy + (size_t)rw[u] * a.lddy + 4 * lane);
      rr[u] = has_res ? *reinterpret_cast<const float4*>(a.res.ptr + (size_t)rw[u] * a.res.ld + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !two) break;
      const float mean = st[u].x, rstd = st[u].y;
      const float xh[4] = {(xv[u].x - mean) * rstd, (xv[u].y - mean) * rstd, (xv[u].z - mean) * rstd, (xv[u].w - mean) * rstd};
      const float dy[4] = {dyv[u].x, dyv[u].y, dyv[u].z, dyv[u].w};
      const float g4[4] = {gv.x, gv.y, gv.z, gv.w};
      const float r4[4] = {rr[u].x, rr[u].y, rr[u].z, rr[u].w};
      float dxh[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { dxh[i] = dy[i] * g4[i]; s1 += dxh[i]; s2 += dxh[i] * xh[i]; }
      s1 = wave_sum(s1) * invd;
      s2 = wave_sum(s2) * invd;
      float dx[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) dx[i] = rstd * (dxh[i] - s1 - xh[i] * s2) + r4[i];
      *reinterpret_cast<float4*>(a.dx + (size_t)rw[u] * a.lddx + 4 * lane) = make_float4(dx[0], dx[1], dx[2], dx[3]);
      ag.x += dy[0] * xh[0]; ag.y += dy[1] * xh[1]; ag.z += dy[2] * xh[2]; ag.w += dy[3] * xh[3];
      ab.x += dy[0]; ab.y += dy[1]; ab.z += dy[2]; ab.w += dy[3];
      if (a.out2) {
        float v2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v2[i] = dx[i] * drop_mult(d2, (uint32_t)rw[u], (uint32_t)(4 * lane + i));
        *reinterpret_cast<float4*>(a.out2 + (size_t)rw[u] * 256 + 4 * lane) = make_float4(v2[0], v2[1], v2[2], v2[3]);
        ac.x += v2[0]; ac.y += v2[1]; ac.z += v2[2]; ac.w += v2[3];
      } else {
        ac.x += dx[0]; ac.y += dx[1]; ac.z += dx[2]; ac.w += dx[3];
      }
    }
  }
  *reinterpret_cast<float4*>(sh + (0 * LNB_WAVES + wv) * 256 + 4 * lane) = ag;
  *reinterpret_cast<float4*>(sh + (1 * LNB_WAVES + wv) * 256 + 4 * lane) = ab;
  *reinterpret_cast<float4*>(sh + (2 * LNB_WAVES + wv) * 256 + 4 * lane) = ac;
  __syncthreads();
  for (int t = threadIdx.x; t < 3 * 256; t += blockDim.x) {
    const int which = t >> 8, col = t & 255;
    float* dst = which == 0 ? a.dgamma : (which == 1 ? a.dbeta : a.colsum);
    if (!dst && !a.partial) continue;
    float s = 0.f;
    for (int w = 0; w < LNB_WAVES; ++w) s += sh[(which * LNB_WAVES + w) * 256 + col];
    if (a.partial) a.partial[((size_t)blockIdx.x * 3 + which) * 256 + col] = s;
    else atomicAdd(&dst[col], s);
  }
}
static bool ln_bwd_v4_takes(const LnBwdArgs& a) {
  if (a.d != 256 || (a.res.mode != RES_NONE && a.res.mode != RES_DIRECT)) return false;
  uintptr_t bits = (uintptr_t)a.x | (uintptr_t)a.dy | (uintptr_t)a.dx | (uintptr_t)a.g | (uintptr_t)a.out2 | (uintptr_t)a.stats;
  int lds = a.ldx | a.lddy | a.lddx;
  if (a.res.mode == RES_DIRECT) { bits |= (uintptr_t)a.res.ptr; lds |= a.res.ld; }
  return (bits & 15) == 0 && (lds & 3) == 0;
}

int ln_bwd_blocks(int rows) {
  int blocks = ps_cdiv(rows, LNB_WAVES);
  return blocks > 256 ? 256 : blocks;   // rows are grid-strided; bounds the gamma/beta partials
}

int launch_ln_bwd(const LnBwdArgs& a, hipStream_t st) {
  PS_REQUIRE(a.d <= 64 * LN_MAXI, "layernorm bwd: d=%d > %d", a.d, 64 * LN_MAXI);
  const int blocks = ln_bwd_blocks(a.rows);
  const int dpl = ps_cdiv(a.d, 64);
  const int dp = dpl <= 1 ? 1 : (dpl <= 2 ? 2 : (dpl <= 4 ? 4 : 8));
  const size_t lds = sizeof(float) * 3 * LNB_WAVES * 64 * dp;
  const dim3 blk(64 * LNB_WAVES);
  const bool exact = a.d == 64 * dp;
  if (ln_bwd_v4_takes(a)) {
    hipLaunchKernelGGL(ln_bwd_v4_kernel, dim3(blocks), blk, sizeof(float) * 3 * LNB_WAVES * 256, st, a);
    PS_LAUNCH_CHECK();
    return PS_OK;
  }
#define LNB_LAUNCH(DP_)                                                                           \
  do {                                                                                            \
    if (exact) hipLaunchKernelGGL((ln_bwd_kernel<DP_, 1>), dim3(blocks), blk, lds, st, a);         \
    else hipLaunchKernelGGL((ln_bwd_kernel<DP_, 0>), dim3(blocks), blk, lds, st, a);               \
  } while (0)
  if (dp == 1) LNB_LAUNCH(1);
  else if (dp == 2) LNB_LAUNCH(2);
  else if (dp == 4) LNB_LAUNCH(4);
  else LNB_LAUNCH(8);
#undef LNB_LAUNCH
  PS_LAUNCH_CHECK();
  return PS_OK;
}

// ================================================================== attention
// MultiHeadedAttention.forward live branch (neural.py:142-145, 206-231) per (sequence, head):
// scores = Qs.K^T, masked_fill(key pad, -1e18), softmax, dropout, attn.V.  S <= 64 keys and
// dh <= 64, so one wave owns one (sequence, head) with everything in LDS.
struct AttnLds {
  float *Ks, *Vs, *Qs, *Ps, *Pd, *valid;
};
__device__ inline AttnLds attn_carve(float* base, int S, int Sq, int dh) {
  AttnLds l;
  l.Ks = base; base += S * dh;
  l.Vs = base; base += S * dh;
  l.Qs = base; base += Sq * dh;
  l.Ps = base; base += Sq * (S + 1);
  l.Pd = base; base += Sq * (S + 1);
  l.valid = base;
  return l;
}
static inline size_t attn_fwd_lds(int S, int Sq, int dh) {
  return sizeof(float) * ((size_t)2 * S * dh + Sq * dh + 2 * Sq * (S + 1) + S);
}

__device__ inline void attn_load_common(const AttnArgs& a, const AttnLds& l, int b, int h, int tid) {
  const int S = a.S, Sq = a.Sq, dh = a.dh, d = a.d;
  for (int i = tid; i < S * dh; i += 64) {
    int s = i / dh, c = i - s * dh;
    size_t off = ((size_t)b * S + s) * d + h * dh + c;
    l.Ks[i] = a.kp[off];
    l.Vs[i] = a.vp[off];
  }
  for (int i = tid; i < Sq * dh; i += 64) {
    int q = i / dh, c = i - q * dh;
    l.Qs[i] = a.qp[((size_t)b * Sq + q) * d + h * dh + c];
  }
  const int brow = b / a.seq_div;
  for (int s = tid; s < S; s += 64)
    l.valid[s] = a.valid ? a.valid[(size_t)brow * S + s] : ((s == 0 || a.ui[(size_t)brow * a.L + s - 1] != a.P) ? 1.f : 0.f);
}

__global__ __launch_bounds__(64) void attn_fwd_kernel(const AttnArgs a) {
  extern __shared__ float lds[];
  const int S = a.S, Sq = a.Sq, dh = a.dh, d = a.d, tid = threadIdx.x;
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  AttnLds l = attn_carve(lds, S, Sq, dh);
  attn_load_common(a, l, b, h, tid);
  __syncthreads();
  for (int idx = tid; idx < Sq * S; idx += 64) {
    int i = idx / S, s = idx - i * S;
    float acc = 0.f;
    for (int c = 0; c < dh; ++c) acc += l.Qs[i * dh + c] * l.Ks[s * dh + c];
    l.Ps[i * (S + 1) + s] = l.valid[s] != 0.f ? acc : -1e18f;
  }
  __syncthreads();
  for (int i = tid; i < Sq; i += 64) {
    float* p = l.Ps + i * (S + 1);
    float m = -INFINITY;
    for (int s = 0; s < S; ++s) m = fmaxf(m, p[s]);
    float sum = 0.f;
    for (int s = 0; s < S; ++s) { float e = expf(p[s] - m); p[s] = e; sum += e; }
    float inv = 1.f / sum;
    for (int s = 0; s < S; ++s) p[s] *= inv;
  }
  __syncthreads();
  for (int idx = tid; idx < Sq * S; idx += 64) {
    int i = idx / S, s = idx - i * S;
    a.attn[((size_t)(b * a.H + h) * Sq + i) * S + s] = l.Ps[i * (S + 1) + s];
  }
  for (int j = 0; j < a.fan; ++j) {
    const int nout = b * a.fan + j;
    const float* P = l.Ps;
    if (a.drop.thr != 0u) {
      __syncthreads();
      for (int idx = tid; idx < Sq * S; idx += 64) {
        int i = idx / S, s = idx - i * S;
        uint32_t row = (uint32_t)((nout * a.H + h) * Sq + i);
        l.Pd[i * (S + 1) + s] = l.Ps[i * (S + 1) + s] * drop_mult(a.drop, row, (uint32_t)s);
      }
      __syncthreads();
      P = l.Pd;
    }
    for (int idx = tid; idx < Sq * dh; idx += 64) {
      int i = idx / dh, c = idx - i * dh;
      float acc = 0.f;
      for (int s = 0; s < S; ++s) acc += P[i * (S + 1) + s] * l.Vs[s * dh + c];
      a.ctx[((size_t)nout * Sq + i) * d + h * dh + c] = acc;
    }
  }
}

int launch_attn_fwd(const AttnArgs& a, hipStream_t st) {
  size_t lds = attn_fwd_lds(a.S, a.Sq, a.dh);
  PS_REQUIRE(a.S <= 64 && lds <= 64 * 1024, "attention: S=%d dh=%d needs %zu B LDS", a.S, a.dh, lds);
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(a.n_in * a.H), dim3(64), lds, st, a);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

static inline size_t attn_bwd_lds(int S, int Sq, int dh) {
  return attn_fwd_lds(S, Sq, dh) + sizeof(float) * ((size_t)Sq * (S + 1) + 2 * S * dh + 2 * Sq * dh);
}

__global__ __launch_bounds__(64) void attn_bwd_kernel(const AttnArgs a) {
  extern __shared__ float lds[];
  const int S = a.S, Sq = a.Sq, dh = a.dh, d = a.d, tid = threadIdx.x;
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  AttnLds l = attn_carve(lds, S, Sq, dh);
  float* dP = l.valid + S;              // [Sq][S+1]
  float* dVs = dP + Sq * (S + 1);       // [S][dh]
  float* dKs = dVs + S * dh;            // [S][dh]
  float* dC = dKs + S * dh;             // [Sq][dh]
  float* dQs = dC + Sq * dh;            // [Sq][dh]
  attn_load_common(a, l, b, h, tid);
  for (int idx = tid; idx < Sq * S; idx += 64) {
    int i = idx / S, s = idx - i * S;
    l.Ps[i * (S + 1) + s] = a.attn[((size_t)(b * a.H + h) * Sq + i) * S + s];
    dP[i * (S + 1) + s] = 0.f;
  }
  for (int i = tid; i < S * dh; i += 64) dVs[i] = 0.f;
  __syncthreads();
  for (int j = 0; j < a.fan; ++j) {
    const int nout = b * a.fan + j;
    for (int idx = tid; idx < Sq * dh; idx += 64) {
      int i = idx / dh, c = idx - i * dh;
      dC[idx] = a.dctx[((size_t)nout * Sq + i) * d + h * dh + c];
    }
    for (int idx = tid; idx < Sq * S; idx += 64) {
      int i = idx / S, s = idx - i * S;
      uint32_t row = (uint32_t)((nout * a.H + h) * Sq + i);
      l.Pd[i * (S + 1) + s] = drop_mult(a.drop, row, (uint32_t)s);   // multiplier only
    }
    __syncthreads();
    for (int idx = tid; idx < S * dh; idx += 64) {          // dV[s][c] += sum_i P*m*dC
      int s = idx / dh, c = idx - s * dh;
      float acc = 0.f;
      for (int i = 0; i < Sq; ++i) acc += l.Ps[i * (S + 1) + s] * l.Pd[i * (S + 1) + s] * dC[i * dh + c];
      dVs[idx] += acc;
    }
    for (int idx = tid; idx < Sq * S; idx += 64) {          // dP[i][s] += m * dC[i].V[s]
      int i = idx / S, s = idx - i * S;
      float acc = 0.f;
      for (int c = 0; c < dh; ++c) acc += dC[i * dh + c] * l.Vs[s * dh + c];
      dP[i * (S + 1) + s] += l.Pd[i * (S + 1) + s] * acc;
    }
    __syncthreads();
  }
  for (int i = tid; i < Sq; i += 64) {                      // softmax backward
    float* p = l.Ps + i * (S + 1);
    float* g = dP + i * (S + 1);
    float t = 0.f;
    for (int s = 0; s < S; ++s) t += p[s] * g[s];
    for (int s = 0; s < S; ++s) g[s] = p[s] * (g[s] - t);
  }
  __syncthreads();
  for (int idx = tid; idx < Sq * dh; idx += 64) {           // dQ (un-scaled linear output)
    int i = idx / dh, c = idx - i * dh;
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc += dP[i * (S + 1) + s] * l.Ks[s * dh + c];
    acc *= a.qscale;
    dQs[idx] = acc;
    a.dq[((size_t)b * Sq + i) * a.lddq + h * dh + c] = acc;
  }
  for (int idx = tid; idx < S * dh; idx += 64) {            // dK, dV
    int s = idx / dh, c = idx - s * dh;
    float acc = 0.f;
    for (int i = 0; i < Sq; ++i) acc += dP[i * (S + 1) + s] * l.Qs[i * dh + c];
    dKs[idx] = acc;
    size_t off = ((size_t)b * S + s) * a.lddkv + h * dh + c;
    a.dkv[off] = acc;
    a.dkv[off + d] = dVs[idx];
  }
  __syncthreads();
  for (int c = tid; c < dh; c += 64) {                      // bias grads
    float sq = 0.f, sk = 0.f, sv = 0.f;
    for (int i = 0; i < Sq; ++i) sq += dQs[i * dh + c];
    for (int s = 0; s < S; ++s) { sk += dKs[s * dh + c]; sv += dVs[s * dh + c]; }
    atomicAdd(&a.dbq[h * dh + c], sq);
    atomicAdd(&a.dbk[h * dh + c], sk);
    atomicAdd(&a.dbv[h * dh + c], sv);
  }
}

int launch_attn_bwd(const AttnArgs& a, hipStream_t st) {
  size_t lds = attn_bwd_lds(a.S, a.Sq, a.dh);
  PS_REQUIRE(a.S <= 64 && lds <= 64 * 1024, "attention bwd: S=%d dh=%d needs %zu B LDS", a.S, a.dh, lds);
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(a.n_in * a.H), dim3(64), lds, st, a);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

// ========================================================== gather + score kernel
// The embedding-gather+score kernel of the metric: one task = one table row
// (target / negative item, or positive / negative word) dotted with its vector
// (the encoder output of its replica, or the target item's row):
//   pos/neg scores  item_transformer.py:464-465,485,493-499
//   item_to_words   item_transformer.py:262-275
// A row group of LPR lanes (16 B per lane) owns 4 tasks at a time: 4 index loads, then
// 4 row + 4 vector loads in flight, then shuffle reductions.
template <int SCORE_U>
__global__ __launch_bounds__(256) void score_fwd_kernel(const ScoreArgs a, int ntask, int lpr) {
  const int tid = threadIdx.x;
  const int gpb = 256 / lpr;                       // row groups per block
  const int grp = blockIdx.x * gpb + tid / lpr;
  const int c = tid % lpr;
  const int nch = a.d >> 2;
  const int t0 = grp * SCORE_U;
  Task tk[SCORE_U];
  float4 r[SCORE_U], v[SCORE_U];
#pragma unroll
  for (int u = 0; u < SCORE_U; ++u) {
    int t = t0 + u;
    if (t < ntask) tk[u] = score_task(a, t);
    else { tk[u].row = nullptr; tk[u].vec = nullptr; tk[u].bias = 0.f; tk[u].out = nullptr; tk[u].term = nullptr; tk[u].tw = 0.f; tk[u].lw = 0.f; }
  }
  float cps = 0.f, cil = 0.f;                      // this row group's share of the two loss sums
  const int wl = lpr == 32 ? 31 : 0;               // the lane of the row group that ends up holding its dot product
#pragma unroll
  for (int u = 0; u < SCORE_U; ++u) {
    r[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    v[u] = r[u];
    if (tk[u].row && c < nch) {
      const float4* rp = reinterpret_cast<const float4*>(tk[u].row + 4 * c);
      const float4* vp = reinterpret_cast<const float4*>(tk[u].vec + 4 * c);
      r[u] = *rp; v[u] = *vp;
    }
  }
#pragma unroll
  for (int u = 0; u < SCORE_U; ++u) {
    float s = r[u].x * v[u].x + r[u].y * v[u].y + r[u].z * v[u].z + r[u].w * v[u].w;
    if (tk[u].row)
      for (int cc = c + lpr; cc < nch; cc += lpr) {          // d > 256 only
        float4 rr = *reinterpret_cast<const float4*>(tk[u].row + 4 * cc);
        float4 vv = *reinterpret_cast<const float4*>(tk[u].vec + 4 * cc);
        s += rr.x * vv.x + rr.y * vv.y + rr.z * vv.z + rr.w * vv.w;
      }
    s = lpr == 32 ? half_sum_last(s) : group_sum(s, lpr);
    if (c == wl && tk[u].out) {
      const float sc = s + tk[u].bias;
      *tk[u].out = sc;
      // BCE-with-logits term of this task (item_transformer.py:510-513, :280): target 1 -> softplus(-s)
      if (tk[u].term) {
        const float term = fabsf(tk[u].tw) * softplus_f(tk[u].tw < 0.f ? -sc : sc);
        *tk[u].term = term;
        if (tk[u].lw > 0.f) cps += term; else cil -= term * tk[u].lw;
      }
    }
  }
  if (!a.loss_blk || a.C > 0) return;
  // per-workgroup loss partials in a fixed order (plain stores: the kernel boundary publishes them); loss_kernel
  // then only has 2 floats per workgroup to reduce instead of every term
  __shared__ float rps[64], ril[64];
  if (c == wl) { rps[tid / lpr] = cps; ril[tid / lpr] = cil; }
  __syncthreads();
  if (tid == 0) {
    float p = 0.f, q = 0.f;
    for (int g2 = 0; g2 < gpb; ++g2) { p += rps[g2]; q += ril[g2]; }
    a.loss_blk[2 * blockIdx.x] = p;
    a.loss_blk[2 * blockIdx.x + 1] = q;
  }
}

// ---- HBM-bound launches (>= 64 MB of rows: the C5 shape, 1 KiB rows of a 51 GB table).  Measured there
// (profiles/r01_gather_score_tuning.txt): the launch is bound by TASKS in flight, not bytes — a word task (1 KiB from
// HBM, its vector cached) costs as much as an item task (2 KiB) — because every task is two dependent round trips of
// ~2.5 us under load and a CU holds at most 32 waves.  So the row groups get NARROWER instead of the waves deeper:
// lpr = d/4/CH lanes per row, CH 16-byte chunks per lane (each wave-instruction still covers 256-B contiguous
// segments), i.e. CH times the tasks per wave at the same occupancy.  (A persistent, software-pipelined form that
// prefetches the next indices under the current rows was correct but 25 % slower: its registers halve the occupancy.)
__device__ inline float row16_sum_last(float v) {   // sum over rows of 16 lanes, valid in lane 15 of each row
#define PS_DPP_ADD(ctrl) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
  PS_DPP_ADD(0x111); PS_DPP_ADD(0x112); PS_DPP_ADD(0x114); PS_DPP_ADD(0x118);
#undef PS_DPP_ADD
  return v;
}
// (scalar registers capped at 80: 256-thread workgroups are admitted 8 per CU up to 80 SGPRs, 7 at 82-96 — MI355X_MICROARCH.md,
// residency; the kernel asked for 85 — and this launch lives on workgroups in flight: profiles/r04_gather_score_wg_times.txt)
template <int SCORE_U, int CH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void score_fwd_wide_kernel(const ScoreArgs a, int ntask, int lpr) {
  const int tid = threadIdx.x;
  const int gpb = 256 / lpr;
  const int grp = blockIdx.x * gpb + tid / lpr;
  const int c = tid % lpr;
  const int t0 = grp * SCORE_U;
#if PS_DIAG_ON      // [4 * workgroup + {0 start, 1 indices known, 2 rows arrived, 3 end}]: the 100 MHz counter all CUs share
#define GS_STAMP(slot)                                                                                    \
  do {                                                                                                    \
    if (a.stamp && threadIdx.x == 0) {                                                                    \
      unsigned long long t_;                                                                              \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                      \
      a.stamp[4 * (size_t)blockIdx.x + (slot)] = t_;                                                      \
    }                                                                                                     \
  } while (0)
#else
#define GS_STAMP(slot) do { } while (0)
#endif
  GS_STAMP(0);
  Task tk[SCORE_U];
  float4 r[SCORE_U][CH], v[SCORE_U][CH];
  bool live[SCORE_U];
  // A row group past the end repeats the LAST task (same addresses, its result dropped) instead of skipping its loads: a
  // load under a per-lane test is a branch of its own, and at the join the compiler waits for everything in flight
  //  — the vector (known from the task number alone) was then requested only after the index had come back.
#pragma unroll
  for (int u = 0; u < SCORE_U; ++u) {
    live[u] = t0 + u < ntask;
    tk[u] = score_task(a, min(t0 + u, ntask - 1));
  }
  float cps = 0.f, cil = 0.f;
  const int wl = lpr == 32 ? 31 : (lpr == 16 ? 15 : 0);
#if PS_DIAG_ON
  if (a.stamp) { asm volatile("" ::"v"(tk[0].row)); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
  GS_STAMP(1);
#pragma unroll
  for (int u = 0; u < SCORE_U; ++u)
#pragma unroll
    for (int k = 0; k < CH; ++k) v[u][k] = *reinterpret_cast<const float4*>(tk[u].vec + 4 * (c + lpr * k));
#pragma unroll
  for (int u = 0; u < SCORE_U; ++u)
#pragma unroll
    for (int k = 0; k < CH; ++k) r[u][k] = *reinterpret_cast<const float4*>(tk[u].row + 4 * (c + lpr * k));
#if PS_DIAG_ON
  if (a.stamp) { asm volatile("" ::"v"(r[0][0].x), "v"(v[0][0].x)); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
  GS_STAMP(2);
#pragma unroll
  for (int u = 0; u < SCORE_U; ++u) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k)
      s += r[u][k].x * v[u][k].x + r[u][k].y * v[u][k].y + r[u][k].z * v[u][k].z + r[u][k].w * v[u][k].w;
    s = lpr == 32 ? half_sum_last(s) : (lpr == 16 ? row16_sum_last(s) : group_sum(s, lpr));
    if (c == wl && live[u]) {
      const float sc = s + tk[u].bias;
      *tk[u].out = sc;
      if (tk[u].term) {
        const float term = fabsf(tk[u].tw) * softplus_f(tk[u].tw < 0.f ? -sc : sc);
        *tk[u].term = term;
        if (tk[u].lw > 0.f) cps += term; else cil -= term * tk[u].lw;
      }
    }
  }
  GS_STAMP(3);
  if (!a.loss_blk || a.C > 0) return;
  __shared__ float rps[64], ril[64];
  if (c == wl) { rps[tid / lpr] = cps; ril[tid / lpr] = cil; }
  __syncthreads();
  if (tid == 0) {
    float p = 0.f, q = 0.f;
    for (int g2 = 0; g2 < gpb; ++g2) { p += rps[g2]; q += ril[g2]; }
    a.loss_blk[2 * blockIdx.x] = p;
    a.loss_blk[2 * blockIdx.x + 1] = q;
  }
}

// ---- The same launch with the index hop on the SCALAR path (round 5).  The per-workgroup stamps of the kernel above
// (profiles/r04_gather_score_wg_times.txt) put 1.5 us (p90 3.6) of a 5.3 us workgroup life into "indices known": the lanes'
// index loads are vector-memory requests and queue, in the CU's in-order texture path, behind the 1-KiB row requests of every
// wave that started earlier.  Here a wave owns 64 / LPR CONSECUTIVE tasks of ONE kind (item tasks first, word tasks from a
// fresh wave), so what a task needs before its row can be requested is wave-uniform: the decode runs on the scalar ALU and
// the three indices of every task (item / word id, the row's target item, the window's positive word) arrive together by
// s_load_dwordx2 through the scalar cache — a path of its own to the L2, not behind the rows; a lane picks its row group's
// values with v_cndmask.  Item waves request their vectors (encoder rows: known from the task number) before the indices
// are back.  No load sits behind a per-lane test: tasks past the end of a kind repeat its last task.
#define PS_SLOAD_I64(dst, ptr) asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=s"(dst) : "s"(ptr) : "memory")
template <int G> struct SIdx { int64_t a[G], b[G], c[G]; };
template <int G>
__device__ inline void sidx_wait(SIdx<G>& x) {                     // ONE wait for all of the wave's scalar loads
  if constexpr (G == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(x.a[0]), "+s"(x.a[1]), "+s"(x.b[0]), "+s"(x.b[1]), "+s"(x.c[0]), "+s"(x.c[1]) :: "memory");
  else if constexpr (G == 4)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(x.a[0]), "+s"(x.a[1]), "+s"(x.a[2]), "+s"(x.a[3]), "+s"(x.b[0]), "+s"(x.b[1]), "+s"(x.b[2]), "+s"(x.b[3]),
                 "+s"(x.c[0]), "+s"(x.c[1]), "+s"(x.c[2]), "+s"(x.c[3]) :: "memory");
  else {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(x.a[0]), "+s"(x.a[1]), "+s"(x.a[2]), "+s"(x.a[3]), "+s"(x.a[4]), "+s"(x.a[5]), "+s"(x.a[6]), "+s"(x.a[7]),
                 "+s"(x.b[0]), "+s"(x.b[1]), "+s"(x.b[2]), "+s"(x.b[3]), "+s"(x.b[4]), "+s"(x.b[5]), "+s"(x.b[6]), "+s"(x.b[7]) :: "memory");
    asm volatile("" : "+s"(x.c[0]), "+s"(x.c[1]), "+s"(x.c[2]), "+s"(x.c[3]), "+s"(x.c[4]), "+s"(x.c[5]), "+s"(x.c[6]), "+s"(x.c[7]) :: "memory");
  }
}
template <int CH, int LPR>
__global__ __launch_bounds__(256) void score_fwd_sidx_kernel(const ScoreArgs a, int ntask) {
  constexpr int G = 64 / LPR;                                    // tasks per wave
  constexpr int D = 4 * CH * LPR;                                // = a.d (launch_score_fwd)
  const int tid = threadIdx.x;
  const int lane = tid & 63, myg = lane / LPR, c = lane % LPR;
  const int K1 = a.K + 1, nitem = a.B * K1, nword = ntask - nitem;
  const int niw = (nitem + G - 1) / G, nww = (nword + G - 1) / G;
  const int wv0 = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool wlive = wv0 < niw + nww;
  const int wv = wlive ? wv0 : niw + nww - 1;
  const bool item = wv < niw;                                     // the kind of all tasks of this wave (uniform)
  const int ncat = item ? nitem : nword;
  const int u0 = (item ? wv : wv - niw) * G;
#if PS_DIAG_ON
  unsigned long long stp[4] = {0, 0, 0, 0};
#define GSS_STAMP(slot) do { if (a.stamp) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stp[slot])::"memory"); } while (0)
#else
#define GSS_STAMP(slot) do { } while (0)
#endif
  GSS_STAMP(0);
  // what a lane knows of its task without memory: (q, j) = (row of the kind's index matrix, column), output slots, term sign
  const int u_l = min(u0 + myg, ncat - 1);
  const bool live = wlive && u0 + myg < ncat;
  const int q_l = fdiv(u_l, a.fK1), j_l = u_l - q_l * K1;
  const int b_l = item ? q_l : fdiv(u_l, a.fWK1);
  float* out_l = (item ? a.item_scores : a.word_scores) + u_l;
  float* term_l = (item ? a.item_terms : a.word_terms) + u_l;
  const float tw_l = j_l == 0 ? (item ? -(a.pos_weight ? (float)a.K : 1.f) : -1.f) : 1.f;
  // hop 1 (scalar): index of the task's row, the target item of its batch row, the positive word of its window
  const int64_t* pFirst = item ? a.target : a.pos_words;          // [*, 1]: column 0 of the kind's index matrix
  const int64_t* pRest = item ? a.neg_items : a.neg_words;        // [*, K]: columns 1 .. K
  SIdx<G> x;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int u = min(u0 + g, ncat - 1);
    const int q = fdiv(u, a.fK1), j = u - q * K1;
    const int bb = item ? q : fdiv(u, a.fWK1);
    const int64_t* pa = j == 0 ? pFirst + q : pRest + ((size_t)q * a.K + (j - 1));
    PS_SLOAD_I64(x.a[g], pa);
    PS_SLOAD_I64(x.b[g], a.target + bb);
    PS_SLOAD_I64(x.c[g], pFirst + q);
  }
  float4 r[CH], v[CH];
  if (item) {                                                      // encoder rows: requested under the index hop
    const float* vec = a.enc + ((size_t)b_l * a.R + (a.R > 1 ? j_l : 0)) * D;
#pragma unroll
    for (int k = 0; k < CH; ++k) v[k] = *reinterpret_cast<const float4*>(vec + 4 * (c + LPR * k));
  }
  sidx_wait<G>(x);
  int64_t ia = x.a[0], ib = x.b[0], ic = x.c[0];
#pragma unroll
  for (int g = 1; g < G; ++g) {
    ia = myg == g ? x.a[g] : ia;
    ib = myg == g ? x.b[g] : ib;
    ic = myg == g ? x.c[g] : ic;
  }
  GSS_STAMP(1);
  const int64_t idx = clamp_idx(ia, item ? a.P : a.V - 1);
  if (!item) {
    const float* vec = a.product_emb + (size_t)clamp_idx(ib, a.P) * D;
#pragma unroll
    for (int k = 0; k < CH; ++k) v[k] = *reinterpret_cast<const float4*>(vec + 4 * (c + LPR * k));
  }
  const float* row = (item ? a.product_emb : a.word_emb) + (size_t)idx * D;
#pragma unroll
  for (int k = 0; k < CH; ++k) r[k] = *reinterpret_cast<const float4*>(row + 4 * (c + LPR * k));
  // under the rows: the row's bias; the weight of a word task's term in the batch loss (masked mean over the window,
  // get_vector_mean, item_transformer.py:281: padded slots drop out)
  const float* pbias = item ? (a.bias_product ? a.product_bias : nullptr) : a.word_bias;
  const float bias_l = pbias ? pbias[idx] : 0.f;
  float lw_l = 1.f;
  if (!item) {
    int cnt = ic != a.V - 1;
    if (a.W > 1) {
      cnt = 0;
      for (int ww = 0; ww < a.W; ++ww) cnt += a.pos_words[(size_t)b_l * a.W + ww] != a.V - 1;
    }
    lw_l = ic != a.V - 1 ? -1.f / (float)cnt : -0.f;
  }
#if PS_DIAG_ON
  if (a.stamp) { asm volatile("" ::"v"(r[0].x), "v"(v[0].x)); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
  GSS_STAMP(2);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < CH; ++k) s += r[k].x * v[k].x + r[k].y * v[k].y + r[k].z * v[k].z + r[k].w * v[k].w;
  s = LPR == 32 ? half_sum_last(s) : (LPR == 16 ? row16_sum_last(s) : group_sum(s, LPR));
  constexpr int wl = LPR == 32 ? 31 : (LPR == 16 ? 15 : 0);
  float cps = 0.f, cil = 0.f;
  if (c == wl && live) {
    const float sc = s + bias_l;
    *out_l = sc;
    const float term = fabsf(tw_l) * softplus_f(tw_l < 0.f ? -sc : sc);
    *term_l = term;
    if (lw_l > 0.f) cps += term; else cil -= term * lw_l;
  }
  GSS_STAMP(3);
#if PS_DIAG_ON
  if (a.stamp && tid == 0)
    for (int q = 0; q < 4; ++q) a.stamp[4 * (size_t)blockIdx.x + q] = stp[q];
#endif
  if (!a.loss_blk) return;
  constexpr int gpb = 256 / LPR;
  __shared__ float rps[64], ril[64];
  if (tid < 64) { rps[tid] = 0.f; ril[tid] = 0.f; }
  __syncthreads();
  if (c == wl) { rps[tid / LPR] = cps; ril[tid / LPR] = cil; }
  __syncthreads();
  if (tid < 64) {                                                  // a fixed tree: bitwise reproducible
    const float p = wave_sum(rps[tid]), q = wave_sum(ril[tid]);
    if (tid == 0) { a.loss_blk[2 * blockIdx.x] = p; a.loss_blk[2 * blockIdx.x + 1] = q; }
  }
  (void)gpb;
}

// wide form: chunks per lane (0 = use the one-chunk kernels above).  Default: 16 lanes per row (d = 128: 2 chunks,
// 256: 4, 512: 8), 8 lanes per row for launches of >= 256 MB of rows; other widths keep the one-chunk kernels.
// Measured (MI355X): C5 shape B=1024 22.6 -> 13.6 us (4.99 TB/s), B=8192 167 -> 95.7 us (5.65 TB/s = 0.71 of the HBM
// peak); the latency-bound C2 launch 5.03 -> 4.23 us back to back (floor of an empty launch: 3.8 us).
static int score_wide_ch(const ScoreArgs& a, int ntask) {
  static const int env = ps_diag_int("PS_SCORE_CH", -1);   // tuning: 0 off, 2 / 4 / 8 force
  const int nch = a.d / 4;
  int ch;
  if (env >= 0) ch = env;
  else {
    // 8 lanes per row from 128 MB of rows per launch (round 3: 256 MB): with index sets that really come from HBM the B = 8192
    // launch of the C5 shape runs 97.5 us against 103.4 (0.69 against 0.65 of the HBM peak); at B = 1024 the two tie (15.2 / 15.3)
    const bool huge = (size_t)ntask * a.d * 4 >= ((size_t)128 << 20);
    ch = nch / (huge ? 8 : 16);
    if (ch > 8) ch = 8;
  }
  while (ch > 1 && (nch % ch != 0 || nch / ch < 8 || ((nch / ch) & (nch / ch - 1)) != 0 || nch / ch > 64)) ch >>= 1;
  return (ch == 2 || ch == 4 || ch == 8) && nch / ch <= 64 ? ch : 0;
}
static int score_wide_u() {
  static const int env = ps_diag_int("PS_SCORE_WIDE_U", 1);
  static const int ch = ps_diag_int("PS_SCORE_CH", 0);
  return env == 2 && ch != 8 ? 2 : 1;
}

// one-chunk fallback: one task per row group below 64 MB of rows per launch, two above
static int score_one_chunk_u(const ScoreArgs& a, int ntask) {
  static const int Uenv = ps_diag_int("PS_SCORE_U", 0);     // tuning experiments (1 or 2)
  if (Uenv == 1 || Uenv == 2) return Uenv;
  return (size_t)ntask * a.d * 4 < ((size_t)64 << 20) ? 1 : 2;
}

// training launches whose draws sit in memory take the index hop on the scalar path (score_fwd_sidx_kernel; PS_SCORE_SIDX=0: the
// per-lane form): workgroups of the launch, 0 = not that form.  A wave holds 64 / lpr tasks of ONE kind.
static int score_sidx_blocks(const ScoreArgs& a, int ntask, int ch) {
  static const bool sidx = ps_env_int("PS_SCORE_SIDX", 1) != 0;
  const int wl = a.d / 4 / ch;
  if (!sidx || a.C != 0 || a.samp_inline || score_wide_u() != 1 || (wl != 16 && wl != 8) || ntask <= 0) return 0;
  if (!((wl == 16 && (ch == 2 || ch == 4 || ch == 8)) || (wl == 8 && (ch == 4 || ch == 8)))) return 0;
  const int G = 64 / wl, nitem = a.B * (a.K + 1);
  return ps_cdiv(ps_cdiv(nitem, G) + ps_cdiv(ntask - nitem, G), 4);
}

int score_fwd_blocks(const ScoreArgs& a) {
  const int ntask = a.C > 0 ? a.B * a.C : a.B * (a.K + 1) * (1 + a.W);
  if (const int ch = score_wide_ch(a, ntask)) {
    if (const int sb = score_sidx_blocks(a, ntask, ch)) return sb;
    return ps_cdiv(ps_cdiv(ntask, score_wide_u()), 256 / (a.d / 4 / ch));
  }
  return ps_cdiv(ps_cdiv(ntask, score_one_chunk_u(a, ntask)), 256 / lpr_for(a.d));
}

int launch_score_fwd(ScoreArgs& a, hipStream_t st) {
  PS_REQUIRE(a.d % 4 == 0, "score: d %% 4");
  KTimeScope kt("gather_score", st);
#if PS_DIAG_ON
  a.stamp = ps_diag_int("PS_SCORE_STAMP", 0) ? ps_debug_stamp_ptr() : nullptr;
#else
  a.stamp = nullptr;
#endif
  int ntask = a.C > 0 ? a.B * a.C : a.B * (a.K + 1) * (1 + a.W);
  if (const int ch = score_wide_ch(a, ntask)) {
    const int wl = a.d / 4 / ch, wu = score_wide_u();
    const int wb = ps_cdiv(ps_cdiv(ntask, wu), 256 / wl);
    a.loss_nblk = wb;
    if (const int sb = score_sidx_blocks(a, ntask, ch)) {
      a.loss_nblk = sb;
      bool done = true;
      if (wl == 16 && ch == 2) PS_KLAUNCH((score_fwd_sidx_kernel<2, 16>), dim3(sb), dim3(256), 0, st, a, ntask);
      else if (wl == 16 && ch == 4) PS_KLAUNCH((score_fwd_sidx_kernel<4, 16>), dim3(sb), dim3(256), 0, st, a, ntask);
      else if (wl == 16 && ch == 8) PS_KLAUNCH((score_fwd_sidx_kernel<8, 16>), dim3(sb), dim3(256), 0, st, a, ntask);
      else if (wl == 8 && ch == 4) PS_KLAUNCH((score_fwd_sidx_kernel<4, 8>), dim3(sb), dim3(256), 0, st, a, ntask);
      else if (wl == 8 && ch == 8) PS_KLAUNCH((score_fwd_sidx_kernel<8, 8>), dim3(sb), dim3(256), 0, st, a, ntask);
      else { done = false; a.loss_nblk = wb; }
      if (done) { PS_LAUNCH_CHECK(); return PS_OK; }
    }
    if (ch == 8) PS_KLAUNCH((score_fwd_wide_kernel<1, 8>), dim3(wb), dim3(256), 0, st, a, ntask, wl);
    else if (ch == 4 && wu == 1) PS_KLAUNCH((score_fwd_wide_kernel<1, 4>), dim3(wb), dim3(256), 0, st, a, ntask, wl);
    else if (ch == 4) PS_KLAUNCH((score_fwd_wide_kernel<2, 4>), dim3(wb), dim3(256), 0, st, a, ntask, wl);
    else if (wu == 1) PS_KLAUNCH((score_fwd_wide_kernel<1, 2>), dim3(wb), dim3(256), 0, st, a, ntask, wl);
    else PS_KLAUNCH((score_fwd_wide_kernel<2, 2>), dim3(wb), dim3(256), 0, st, a, ntask, wl);
    PS_LAUNCH_CHECK();
    return PS_OK;
  }
  // widths the wide form does not cover (d/4 not 16 x {2,4,8}): one 16-byte chunk per lane, lpr_for(d) lanes per row
  const int lpr = lpr_for(a.d), U = score_one_chunk_u(a, ntask);
  const int blocks = ps_cdiv(ps_cdiv(ntask, U), 256 / lpr);
  a.loss_nblk = blocks;
  if (U == 1) PS_KLAUNCH((score_fwd_kernel<1>), dim3(blocks), dim3(256), 0, st, a, ntask, lpr);
  else PS_KLAUNCH((score_fwd_kernel<2>), dim3(blocks), dim3(256), 0, st, a, ntask, lpr);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

// Loss (item_transformer.py:500-514 weighted BCE-with-logits; :277-282 PV loss): the gather+score kernel leaves one
// {ps, il} partial per workgroup (terms already weighted by the window mask); one workgroup reduces them in a fixed
// tree => bitwise reproducible, a single memory round trip.  (Folding this last step into the score
// kernel with a last-ticket workgroup was measured 7x slower: the device-scope release each workgroup needs
// writes back its XCD's L2.)
__global__ __launch_bounds__(256) void loss_kernel(const ScoreArgs a) {
  __shared__ float sps[4], sil[4];
  const int tid = threadIdx.x;
  float aps = 0.f, ail = 0.f;
  const int n2 = a.loss_nblk >> 1;                                         // two workgroups' {ps, il} per 16 bytes
  const float4* p4 = reinterpret_cast<const float4*>(a.loss_blk);          // workspace regions are 16-byte aligned
  // (eight loads in flight per trip; an entry past the end enters with weight 0 through an fma — the same sums in the same order,
  //  common.h strided_sum_f32: as one load per trip this single block, which the whole chip waits for, took 6 serial round trips at C5)
  for (int t = tid; t < n2; t += 8 * 256) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int j = t + 256 * u; v[u] = p4[j < n2 ? j : 0]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float w = t + 256 * u < n2 ? 1.f : 0.f;
      aps = __builtin_fmaf(v[u].x + v[u].z, w, aps); ail = __builtin_fmaf(v[u].y + v[u].w, w, ail);
    }
  }
  if (tid == 0 && (a.loss_nblk & 1)) { aps += a.loss_blk[2 * (a.loss_nblk - 1)]; ail += a.loss_blk[2 * a.loss_nblk - 1]; }
  aps = wave_sum(aps); ail = wave_sum(ail);
  if ((tid & 63) == 0) { sps[tid >> 6] = aps; sil[tid >> 6] = ail; }
  __syncthreads();
  if (tid == 0) {
    const float ps = ((sps[0] + sps[1]) + (sps[2] + sps[3])) / (float)a.B;
    const float il = ((sil[0] + sil[1]) + (sil[2] + sil[3])) / (float)a.B;
    a.loss3[0] = ps + il; a.loss3[1] = ps; a.loss3[2] = il;
    if (a.loss_acc) { a.loss_acc[0] += ps; a.loss_acc[1] += il; }   // model.ps_loss / item_loss running sums
  }
}

const void* loss_kernel_handle() { return reinterpret_cast<const void*>(loss_kernel); }

int launch_loss(const ScoreArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(256), 0, st, a);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

// Backward of score + loss: per batch row, half-wave (32 lanes, 128-B segments) per task so
// that every fp32 atomic wave-instruction covers two contiguous 128-B row segments.
#define BW_MAXE 16   // d <= 512
#define SB_RG 16     // half-wave row groups per workgroup
// With replicas (R > 1) the item tasks are independent of one another (every task owns its d enc row), so they get
// workgroups of their own behind the B per-row workgroups: SB_RG tasks each, one round, instead of two dependent
// rounds inside the row's workgroup — the launch is a chain of memory round trips, not bandwidth.
// EPL = 32-column groups per row (d = 32 * EPL exactly; EPL = 16 also serves every other d <= 512, with the columns past d
// masked by VALUE).  Every task is "fetch, then add": all of a task's operand loads are issued unconditionally, then all of its
// atomics, with pad rows adding 0.f to a real address instead of branching — a load or an atomic under a per-group condition
// (the `k < epl` form this replaces) made the compiler wait for everything in flight at every join, i.e. for the previous
// group's ATOMICS to be acknowledged before the next group's loads were even issued: four dependent round trips per task.
template <int EPL>
__global__ __launch_bounds__(32 * SB_RG) void score_bwd_kernel(const ScoreArgs a) {
  extern __shared__ float red[];                 // [SB_RG][d]
  const int tid = threadIdx.x, rg = tid >> 5, c = tid & 31;
  const int d = a.d, K1 = a.K + 1;
  const bool split = a.R > 1;
  GS_STAMP(0);
  const bool item_wg = split && (int)blockIdx.x >= 2 * a.B;
  int b = blockIdx.x, j0 = rg, j1 = K1;
  int wt0 = rg, wstride = SB_RG;                 // word tasks of this workgroup: wt0, wt0 + wstride, ...
  if (item_wg) {
    const int t = ((int)blockIdx.x - 2 * a.B) * SB_RG + rg;
    if (t >= a.B * K1) return;
    b = fdiv(t, a.fK1); j0 = t - b * K1; j1 = j0 + 1;
  } else if (split) {
    // two workgroups per batch row share its W*(K+1) word tasks (21 at C2: one round each instead of two dependent
    // rounds in one workgroup); each adds its part of the target-item row gradient
    b = (int)blockIdx.x >> 1;
    wt0 = rg + ((int)blockIdx.x & 1) * SB_RG; wstride = 2 * SB_RG;
    j1 = 0;                                      // the row's workgroups only do the word tasks
  }
  const float invB = a.scale * (a.scale_dev ? *a.scale_dev : 1.f) / (float)a.B;
  const float wpos = a.pos_weight ? (float)a.K : 1.f;
  const int64_t tb = clamp_idx(a.target[b], a.P);
  int col[EPL]; bool cok[EPL];                   // this lane's columns (clamped: every address is a real one)
#pragma unroll
  for (int k = 0; k < EPL; ++k) { cok[k] = c + 32 * k < d; col[k] = cok[k] ? c + 32 * k : c; }
  float acc[EPL];
#pragma unroll
  for (int k = 0; k < EPL; ++k) acc[k] = 0.f;
  const bool scatter = a.part != 1, want_denc = a.denc && a.part != 2;
  // ---- item tasks
  for (int j = j0; j < j1; j += SB_RG) {
    const int64_t idx = clamp_idx(j == 0 ? a.target[b] : a.neg_items[(size_t)b * a.K + j - 1], a.P);
    const float s = a.item_scores[(size_t)b * K1 + j];
    const float ds = (j == 0 ? wpos * (sigmoid_f(s) - 1.f) : sigmoid_f(s)) * invB;
    const float* encr = a.enc + ((size_t)b * a.R + (a.R > 1 ? j : 0)) * d;
    const float* row = a.product_emb + (size_t)idx * d;
    float* grow = a.g_product_emb + (size_t)idx * d;
    float rv[EPL], ev[EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) { rv[k] = row[col[k]]; ev[k] = encr[col[k]]; }
    if (a.R > 1) {
      if (want_denc) {
        float* de = a.denc + ((size_t)b * a.R + j) * d;
#pragma unroll
        for (int k = 0; k < EPL; ++k)
          if (EPL < 16 || cok[k]) de[col[k]] = ds * rv[k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < EPL; ++k) acc[k] += ds * rv[k];
    }
    if (scatter) {
      const bool live = idx != a.P;              // the padding row takes +0.f (its gradient stays exactly zero)
#pragma unroll
      for (int k = 0; k < EPL; ++k)
        if (EPL < 16 || cok[k]) atomicAdd(&grow[col[k]], live ? ds * ev[k] : 0.f);
      if (a.bias_product && c == 0) atomicAdd(&a.g_product_bias[idx], ds);
    }
  }
  if (a.R == 1) {
#pragma unroll
    for (int k = 0; k < EPL; ++k)
      if (EPL < 16 || cok[k]) red[rg * d + col[k]] = acc[k];
    __syncthreads();
    for (int e = tid; e < d; e += 32 * SB_RG) {
      float s = 0.f;
      for (int r = 0; r < SB_RG; ++r) s += red[r * d + e];
      a.denc[(size_t)b * d + e] = s;
    }
    __syncthreads();
  }
  GS_STAMP(1);
  if (item_wg || a.part == 1) { GS_STAMP(3); return; }
  // ---- word tasks
#pragma unroll
  for (int k = 0; k < EPL; ++k) acc[k] = 0.f;
  int cnt = 0;
  for (int w = 0; w < a.W; ++w) cnt += (a.pos_words[(size_t)b * a.W + w] != a.V - 1);
  const float cf = invB / (float)(cnt > 0 ? cnt : 1);
  const float* prow = a.product_emb + (size_t)tb * d;
  float pv[EPL];
#pragma unroll
  for (int k = 0; k < EPL; ++k) pv[k] = prow[col[k]];
  for (int t = wt0; t < a.W * K1; t += wstride) {
    const int w = t / K1, j = t - w * K1;
    const int64_t pw = a.pos_words[(size_t)b * a.W + w];
    if (pw == a.V - 1) continue;                                  // masked window slot (get_vector_mean)
    const int64_t idx = clamp_idx(j == 0 ? pw : a.neg_words[(size_t)b * a.W * a.K + (size_t)w * a.K + j - 1], a.V - 1);
    const float s = a.word_scores[((size_t)b * a.W + w) * K1 + j];
    const float ds = (j == 0 ? sigmoid_f(s) - 1.f : sigmoid_f(s)) * cf;
    const float* wrow = a.word_emb + (size_t)idx * d;
    float* grow = a.g_word_emb + (size_t)idx * d;
    float wv[EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) wv[k] = wrow[col[k]];
    const bool live = idx != a.V - 1;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      if (EPL < 16 || cok[k]) atomicAdd(&grow[col[k]], live ? ds * pv[k] : 0.f);
      acc[k] += ds * wv[k];
    }
    if (c == 0) atomicAdd(&a.g_word_bias[idx], ds);
  }
#pragma unroll
  for (int k = 0; k < EPL; ++k)
    if (EPL < 16 || cok[k]) red[rg * d + col[k]] = acc[k];
  GS_STAMP(2);
  __syncthreads();
  if (tb != a.P)
    for (int e = tid; e < d; e += 32 * SB_RG) {
      float s = 0.f;
      for (int r = 0; r < SB_RG; ++r) s += red[r * d + e];
      atomicAdd(&a.g_product_emb[(size_t)tb * d + e], s);
    }
  GS_STAMP(3);
}
static void launch_score_bwd_kernel(const ScoreArgs& a, int blocks, hipStream_t st) {
  const size_t lds = (size_t)SB_RG * a.d * sizeof(float);
  if (a.d == 32) hipLaunchKernelGGL(score_bwd_kernel<1>, dim3(blocks), dim3(32 * SB_RG), lds, st, a);
  else if (a.d == 64) hipLaunchKernelGGL(score_bwd_kernel<2>, dim3(blocks), dim3(32 * SB_RG), lds, st, a);
  else if (a.d == 128) hipLaunchKernelGGL(score_bwd_kernel<4>, dim3(blocks), dim3(32 * SB_RG), lds, st, a);
  else if (a.d == 256) hipLaunchKernelGGL(score_bwd_kernel<8>, dim3(blocks), dim3(32 * SB_RG), lds, st, a);
  else hipLaunchKernelGGL(score_bwd_kernel<16>, dim3(blocks), dim3(32 * SB_RG), lds, st, a);
}

// Deterministic form of the table scatter above (ps_deterministic): every gradient row has ONE owner — half-wave
// o = row % owners — and an owner walks the task lists in task order, so the fp32 additions into a row happen in the same
// order in every run (the atomics remain, but no two waves ever add into one row).  Lists: item tasks (b, j) -> item row,
// the row's word-task term -> target item row, word tasks -> word row; biases ride with their tasks.
// A popular row's tasks all fall to one owner (Zipf: ~100 history slots of a C2 batch hold the top item), so the walk
// keeps that chain short: keys of 8 x 32 tasks per round trip, and an owner's matches are FETCHED four at a time (their
// operand rows in flight together) before they are COMMITTED — the atomics — in task order.
#define SBD_OWNERS_PER_WG 8
template <int EPL> struct DetItemT { float v[EPL]; float bias; int64_t row; };     // EPL = 32-column groups per row (>= d/32)      // one task's finished contribution: row values (+ bias entry)
// fetch(t, item) fills the contribution of task t; the walk adds it to table[row] (skipped for row == skip_row) and to
// bias[row] (if given).  Consecutive matches of one row are summed in registers first (in task order) — a popular row's
// chain is then one atomic per up to DW_B tasks.
template <int EPL, class FetchF>
__device__ inline void det_owner_walk(int ntask, int owner, int nown, int hl, int d, float* table, float* bias, int64_t skip_row,
                                      const int32_t* keys, FetchF fetch) {
  // A round = DW_U chunks of 32 tasks: all their keys are requested together, lane u of the half-wave keeps the match
  // mask of chunk u, and the round's matches are then taken DW_B at a time ACROSS the chunks (a popular row has a match
  // every few chunks: batching inside one chunk left every fetch alone with its round trip).
  constexpr int DW_U = 32, DW_B = 8;
  const int epl = d >> 5, c = hl, base = (int)(threadIdx.x & 32);
  for (int t0 = 0; t0 < ntask; t0 += 32 * DW_U) {
    uint32_t mymask = 0u;
    {
      int32_t kk[DW_U];                                      // (keys: row ids precomputed by a parallel pass, -1 = no task)
#pragma unroll
      for (int u = 0; u < DW_U; ++u) {
        const int t = t0 + 32 * u + hl;
        kk[u] = keys[t < ntask ? t : ntask - 1];
        if (t >= ntask) kk[u] = -1;
      }
#pragma unroll
      for (int u = 0; u < DW_U; ++u) {
        const bool mine = kk[u] >= 0 && (int)((uint32_t)kk[u] & (uint32_t)(nown - 1)) == owner;   // nown: a power of two
        const unsigned long long bm = __ballot(mine);
        if (hl == u) mymask = (uint32_t)(bm >> base);
      }
    }
    int u = 0;
    uint32_t m = (uint32_t)__shfl((int)mymask, base, 64);
    for (;;) {
      int ids[DW_B], n = 0;
#pragma unroll
      for (int q = 0; q < DW_B; ++q) {
        while (!m && u < DW_U - 1) { ++u; m = (uint32_t)__shfl((int)mymask, base + u, 64); }
        if (m) { ids[q] = t0 + 32 * u + __ffs((int)m) - 1; m &= m - 1; n = q + 1; }
      }
      if (n == 0) break;
      DetItemT<EPL> it[DW_B];
#pragma unroll
      for (int q = 0; q < DW_B; ++q)
        if (q < n) fetch(ids[q], it[q]);
      // commit, in task order; runs of one row first meet in registers
      float acc[EPL]; float bacc = 0.f; int64_t cur = -1;
#pragma unroll
      for (int k = 0; k < EPL; ++k) acc[k] = 0.f;
#pragma unroll
      for (int q = 0; q <= DW_B; ++q) {
        const bool have = q < n;
        const int64_t row = have ? it[q < DW_B ? q : 0].row : -2;
        if (cur >= 0 && row != cur) {                          // flush the finished run (half-wave uniform)
          if (cur != skip_row) {
            float* dst = table + (size_t)cur * d;
#pragma unroll
            for (int k = 0; k < EPL; ++k)
              if (k < epl) atomicAdd(&dst[c + 32 * k], acc[k]);
          }
          if (bias && c == 0) atomicAdd(&bias[cur], bacc);
#pragma unroll
          for (int k = 0; k < EPL; ++k) acc[k] = 0.f;
          bacc = 0.f;
        }
        if (have) {
          cur = row;
#pragma unroll
          for (int k = 0; k < EPL; ++k) acc[k] += it[q < DW_B ? q : 0].v[k];
          bacc += it[q < DW_B ? q : 0].bias;
        } else {
          cur = -1;
        }
      }
    }
  }
}
// Parallel pre-pass of the deterministic score backward: (1) the keys (gradient-row ids, -1 = no task) of the three task
// lists as int32 arrays — every owner reads them 2,048 times over, so they are computed once; (2) the term of the target
// item's gradient that batch row b's word tasks contribute (fixed order): term[b][d], one half-wave per row.
// keys layout: [B*K1] item tasks, [B] target rows, [B*W*K1] word tasks.
__global__ __launch_bounds__(256) void score_bwd_det_pre_kernel(const ScoreArgs a, float* term, int32_t* keys) {
  const int tid = threadIdx.x, hl = tid & 31, c = hl;
  const int d = a.d, epl = d >> 5, K1 = a.K + 1, nt = a.W * K1;
  const int nitem = a.B * K1, nword = a.B * nt;
  {
    const int g = (int)blockIdx.x * 256 + tid, gn = (int)gridDim.x * 256;
    for (int t = g; t < nitem; t += gn) {
      const int b = fdiv(t, a.fK1), j = t - b * K1;
      keys[t] = (int32_t)clamp_idx(j == 0 ? a.target[b] : a.neg_items[(size_t)b * a.K + j - 1], a.P);
    }
    for (int b = g; b < a.B; b += gn) {
      const int64_t tb = clamp_idx(a.target[b], a.P);
      keys[nitem + b] = tb == a.P ? -1 : (int32_t)tb;
    }
    for (int u = g; u < nword; u += gn) {
      const int b = fdiv(u, a.fWK1), r = u - b * nt, w = fdiv(r, a.fK1), j = r - w * K1;
      const int64_t pw = a.pos_words[(size_t)b * a.W + w];
      keys[nitem + a.B + u] = pw == a.V - 1 ? -1
          : (int32_t)clamp_idx(j == 0 ? pw : a.neg_words[(size_t)b * a.W * a.K + (size_t)w * a.K + j - 1], a.V - 1);
    }
  }
  const int b = (int)blockIdx.x * 8 + (tid >> 5);
  if (b >= a.B) return;
  const float invB = a.scale * (a.scale_dev ? *a.scale_dev : 1.f) / (float)a.B;
  int cnt = 0;
  for (int w = 0; w < a.W; ++w) cnt += (a.pos_words[(size_t)b * a.W + w] != a.V - 1);
  const float cf = invB / (float)(cnt > 0 ? cnt : 1);
  float v[BW_MAXE];
#pragma unroll
  for (int k = 0; k < BW_MAXE; ++k) v[k] = 0.f;
  for (int u0 = 0; u0 < nt; u0 += 32) {                        // lane = word task: its coefficient and word id, once
    const int u = u0 + hl;
    float ds = 0.f; int64_t idx = 0;
    if (u < nt) {
      const int w = u / K1, j = u - w * K1;
      const int64_t pw = a.pos_words[(size_t)b * a.W + w];
      if (pw != a.V - 1) {
        idx = clamp_idx(j == 0 ? pw : a.neg_words[(size_t)b * a.W * a.K + (size_t)w * a.K + j - 1], a.V - 1);
        const float s = a.word_scores[((size_t)b * a.W + w) * K1 + j];
        ds = (j == 0 ? sigmoid_f(s) - 1.f : sigmoid_f(s)) * cf;
      }
    }
    const int cntu = min(32, nt - u0), base = (threadIdx.x & 32);
    for (int i = 0; i < cntu; ++i) {                           // in task order
      const float dsi = __shfl(ds, base + i, 64);
      const int64_t idi = __shfl((long long)idx, base + i, 64);
      const float* wrow = a.word_emb + (size_t)idi * d;
#pragma unroll
      for (int k = 0; k < BW_MAXE; ++k)
        if (k < epl) v[k] = fmaf(dsi, wrow[c + 32 * k], v[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < BW_MAXE; ++k)
    if (k < epl) term[(size_t)b * d + c + 32 * k] = v[k];
}
template <int EPL>
__global__ __launch_bounds__(32 * SBD_OWNERS_PER_WG) void score_bwd_det_kernel(const ScoreArgs a, const float* term, const int32_t* keys) {
  typedef DetItemT<EPL> DetItem;
  const int tid = threadIdx.x, hl = tid & 31, c = hl;
  const int owner = (int)blockIdx.x * SBD_OWNERS_PER_WG + (tid >> 5), nown = (int)gridDim.x * SBD_OWNERS_PER_WG;
  const int d = a.d, epl = d >> 5, K1 = a.K + 1;
  const float invB = a.scale * (a.scale_dev ? *a.scale_dev : 1.f) / (float)a.B;
  const float wpos = a.pos_weight ? (float)a.K : 1.f;
  const int nitem = a.B * K1;
  const int32_t* keys_item = keys; const int32_t* keys_tgt = keys + nitem; const int32_t* keys_word = keys + nitem + a.B;
  // ---- item tasks (the pad row P: only its bias entry is touched)
  det_owner_walk<EPL>(a.B * K1, owner, nown, hl, d, a.g_product_emb, a.bias_product ? a.g_product_bias : nullptr, a.P, keys_item,
    [&](int t, DetItem& it) {
      const int b = fdiv(t, a.fK1), j = t - b * K1;
      it.row = keys_item[t];
      const float s = a.item_scores[(size_t)b * K1 + j];
      const float* encr = a.enc + ((size_t)b * a.R + (a.R > 1 ? j : 0)) * d;
      float ev[EPL];
#pragma unroll
      for (int k = 0; k < EPL; ++k) ev[k] = k < epl ? encr[c + 32 * k] : 0.f;
      const float ds = (j == 0 ? wpos * (sigmoid_f(s) - 1.f) : sigmoid_f(s)) * invB;
#pragma unroll
      for (int k = 0; k < EPL; ++k) it.v[k] = ds * ev[k];
      it.bias = ds;
    });
  // ---- the rows' word tasks: their term of the target item's gradient (computed by score_bwd_det_terms_kernel) ...
  det_owner_walk<EPL>(a.B, owner, nown, hl, d, a.g_product_emb, nullptr, a.P, keys_tgt,
    [&](int b, DetItem& it) {
      it.row = keys_tgt[b];
#pragma unroll
      for (int k = 0; k < EPL; ++k) it.v[k] = k < epl ? term[(size_t)b * d + c + 32 * k] : 0.f;
      it.bias = 0.f;
    });
  // ---- ... and the word rows
  det_owner_walk<EPL>(a.B * a.W * K1, owner, nown, hl, d, a.g_word_emb, a.g_word_bias, a.V - 1, keys_word,
    [&](int u, DetItem& it) {
      const int b = fdiv(u, a.fWK1), r = u - b * a.W * K1, w = r / K1, j = r - w * K1;
      it.row = keys_word[u];
      const float s = a.word_scores[((size_t)b * a.W + w) * K1 + j];
      const float* prow = a.product_emb + (size_t)clamp_idx(a.target[b], a.P) * d;
      float pv[EPL];
#pragma unroll
      for (int k = 0; k < EPL; ++k) pv[k] = k < epl ? prow[c + 32 * k] : 0.f;
      int cnt = 0;
      for (int w2 = 0; w2 < a.W; ++w2) cnt += (a.pos_words[(size_t)b * a.W + w2] != a.V - 1);
      const float ds = (j == 0 ? sigmoid_f(s) - 1.f : sigmoid_f(s)) * (invB / (float)(cnt > 0 ? cnt : 1));
#pragma unroll
      for (int k = 0; k < EPL; ++k) it.v[k] = ds * pv[k];
      it.bias = ds;
    });
}

int launch_score_bwd(const ScoreArgs& a, hipStream_t st) {
  PS_REQUIRE(a.d % 32 == 0 && a.d <= 32 * BW_MAXE, "score bwd: d=%d unsupported", a.d);
  if (ps_deterministic()) {
    // d enc (part 1: no table scatter) keeps its kernel — its sums are per row, in a fixed order; the table scatter
    // (part 2) goes through the sole-owner form
    if (a.denc && a.part != 2) {
      ScoreArgs e = a;
      e.part = 1;
      const int iw = a.R > 1 ? ps_cdiv(a.B * (a.K + 1), SB_RG) : 0;
      launch_score_bwd_kernel(e, (a.R > 1 ? 2 : 1) * a.B + iw, st);
      PS_LAUNCH_CHECK();
    }
    if (a.part != 1) {
      const size_t nkeys = (size_t)a.B * (a.K + 1) * (1 + a.W) + a.B;
      float* term = ps_det_scratch(1, (size_t)a.B * a.d + nkeys + 4, st);
      PS_REQUIRE(term, "score bwd: deterministic mode has no scratch (allocation failed or stream capture)");
      int32_t* keys = reinterpret_cast<int32_t*>(term + (size_t)a.B * a.d);
      hipLaunchKernelGGL(score_bwd_det_pre_kernel, dim3(ps_cdiv(a.B, 8)), dim3(256), 0, st, a, term, keys);
      PS_LAUNCH_CHECK();
      if (a.d <= 128) hipLaunchKernelGGL(score_bwd_det_kernel<4>, dim3(256), dim3(32 * SBD_OWNERS_PER_WG), 0, st, a, term, keys);
      else if (a.d <= 256) hipLaunchKernelGGL(score_bwd_det_kernel<8>, dim3(256), dim3(32 * SBD_OWNERS_PER_WG), 0, st, a, term, keys);
      else hipLaunchKernelGGL(score_bwd_det_kernel<16>, dim3(256), dim3(32 * SBD_OWNERS_PER_WG), 0, st, a, term, keys);
      PS_LAUNCH_CHECK();
    }
    return PS_OK;
  }
  const int item_wgs = a.R > 1 ? ps_cdiv(a.B * (a.K + 1), SB_RG) : 0;
#if PS_DIAG_ON
  ScoreArgs as = a;
  as.stamp = ps_diag_int("PS_SBW_STAMP", 0) ? ps_debug_stamp_ptr() : nullptr;
  launch_score_bwd_kernel(as, (a.R > 1 ? 2 : 1) * a.B + item_wgs, st);
#else
  launch_score_bwd_kernel(a, (a.R > 1 ? 2 : 1) * a.B + item_wgs, st);
#endif
  PS_LAUNCH_CHECK();
  return PS_OK;
}

// ========================================================== embedding scatter-add
template <int EPL>
__device__ __forceinline__ void row_fetch_add(float* dst, const float* src, int c, float scale) {
  float v[EPL];
#pragma unroll
  for (int k = 0; k < EPL; ++k) v[k] = src[c + 32 * k];
#pragma unroll
  for (int k = 0; k < EPL; ++k) atomicAdd(&dst[c + 32 * k], scale == 1.f ? v[k] : v[k] * scale);
}
template <int EPL>
__device__ __forceinline__ void row_fetch_add2(float* dst, const float* src, const float* src2, int c) {   // two partial rows
  float v[EPL], u[EPL];
#pragma unroll
  for (int k = 0; k < EPL; ++k) { v[k] = src[c + 32 * k]; u[k] = src2[c + 32 * k]; }
#pragma unroll
  for (int k = 0; k < EPL; ++k) atomicAdd(&dst[c + 32 * k], v[k] + u[k]);
}
// Backward of the history gather (item_transformer.py:466-469) and of the query mean
// (text_encoder.py:6-16 + FS dropout): dense grads with padding_idx rows untouched.
__global__ __launch_bounds__(256) void embed_scatter_kernel(const EmbedBwdArgs a, int ntask, int nq, int nfw, int nfold) {
  fork_signal(a.sig, a.sigval);
  extern __shared__ float fsb_s[];               // fused FS backward only: [d] dqpre, [rpp][d] partials, [d] d mean
  const int tid = threadIdx.x, c = tid & 31;
  const int d = a.d, epl = d >> 5;
  const bool fsb = a.fsb_w != nullptr;
  if ((int)blockIdx.x < nq) {
    // FS backward of batch row b, first in the grid: it is the tail of the step's dependent chain
    const int b = blockIdx.x;
    const int nchunk = d >> 2, rpp = 256 / nchunk;
    const int rg = tid / nchunk, cc = tid - rg * nchunk;
    float* dq_s = fsb_s;
    float* part_s = fsb_s + d;
    float* dm_s = part_s + (size_t)rpp * d;
    const bool pre = d == 128;                    // weight rows o = rg + 8u fetched up front, under the dqpre round trip
    float4 wq[16];
    if (pre) {
#pragma unroll
      for (int u = 0; u < 16; ++u) wq[u] = *reinterpret_cast<const float4*>(a.fsb_w + (size_t)(rg + 8 * u) * 128 + 4 * cc);
    }
    for (int e = tid; e < d; e += 256) {
      const float y = a.fsb_qe[(size_t)b * d + e];
      const float v = (a.fsb_dqe[(size_t)b * a.fsb_lddqe + e] + a.fsb_k2 * a.fsb_dqe2[(size_t)b * a.fsb_lddqe + e]) * (1.f - y * y);
      dq_s[e] = v;
      if (a.fsb_dqpre_out) {                       // (kernel-uniform) the f_W gradient is a GEMM behind this launch: leave it its operand,
        a.fsb_dqpre_out[(size_t)b * d + e] = v;    // and add this row's share of the bias gradient (text_encoder.py:38-39)
        atomicAdd(&a.g_fs_b[e], v);
      }
    }
    int cnt = 0;
    if (a.Q <= 64) {      // one load per lane and a ballot (as a loop: Q loads, each waited for — DESIGN.md 5f, loops)
      const int ql = tid & 63;
      const int64_t qm = a.qw[(size_t)b * a.Q + (ql < a.Q ? ql : 0)];
      cnt = __popcll(__ballot(ql < a.Q && qm != a.V - 1));
    } else {
      for (int q = 0; q < a.Q; ++q) cnt += (a.qw[(size_t)b * a.Q + q] != a.V - 1);
    }
    __syncthreads();
    if (pre) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float s = dq_s[rg + 8 * u];
        acc.x = fmaf(s, wq[u].x, acc.x); acc.y = fmaf(s, wq[u].y, acc.y); acc.z = fmaf(s, wq[u].z, acc.z); acc.w = fmaf(s, wq[u].w, acc.w);
      }
      *reinterpret_cast<float4*>(part_s + (size_t)rg * d + 4 * cc) = acc;
    } else if (rg < rpp) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
      for (int o = rg; o < d; o += rpp) {        // coalesced weight rows, 16 B per lane
        const float4 wv = *reinterpret_cast<const float4*>(a.fsb_w + (size_t)o * d + 4 * cc);
        const float s = dq_s[o];
        acc.x = fmaf(s, wv.x, acc.x); acc.y = fmaf(s, wv.y, acc.y); acc.z = fmaf(s, wv.z, acc.z); acc.w = fmaf(s, wv.w, acc.w);
      }
      *reinterpret_cast<float4*>(part_s + (size_t)rg * d + 4 * cc) = acc;
    }
    __syncthreads();
    const float inv = 1.f / (float)(cnt > 0 ? cnt : 1);
    for (int e = tid; e < d; e += 256) {
      float s = 0.f;
      for (int r = 0; r < rpp; ++r) s += part_s[(size_t)r * d + e];
      dm_s[e] = s * drop_mult(a.drop_fs, (uint32_t)b, (uint32_t)e) * inv;
    }
    __syncthreads();
    if (a.det_dm) {                                // deterministic mode: the sole-owner pass scatters these rows
      for (int e = tid; e < d; e += 256) a.det_dm[(size_t)b * d + e] = dm_s[e];
      return;
    }
    for (int q = tid >> 5; q < a.Q; q += 8) {
      const int64_t idx = a.qw[(size_t)b * a.Q + q];
      if (idx == a.V - 1 || idx < 0 || idx >= a.V) continue;
      float* dst = a.g_word_emb + (size_t)idx * d;
      for (int k = 0; k < epl; ++k) atomicAdd(&dst[c + 32 * k], dm_s[c + 32 * k]);
    }
    return;
  }
  if ((int)blockIdx.x < nq + nfw) {
    // f_W weight gradient (12.6 MFLOP at C2 — not worth a GEMM launch of its own on the tail).  These workgroups
    // come early in the grid so they run under the scatter: 32 outputs each, the batch sum split over 8 lane
    // groups (short dependent chains), reduced through LDS.
    __shared__ float part[8][32];
    __shared__ float bpart[8];
    const int g = ((int)blockIdx.x - nq) * 32 + c, seg = tid >> 5;
    const bool live = g < d * d;
    const int o = live ? g / d : 0, i = live ? g - o * d : 0;
    float acc = 0.f, bacc = 0.f;
    if (fsb) {
#pragma unroll 16
      for (int b = seg; b < a.B; b += 8) {
        const float y = a.fsb_qe[(size_t)b * d + o];
        const float dy = (a.fsb_dqe[(size_t)b * a.fsb_lddqe + o] + a.fsb_k2 * a.fsb_dqe2[(size_t)b * a.fsb_lddqe + o]) * (1.f - y * y);
        acc = fmaf(dy, a.fw_x[(size_t)b * d + i], acc);
        bacc += dy;
      }
    } else {
#pragma unroll 8
      for (int b = seg; b < a.B; b += 8) acc = fmaf(a.fw_dy[(size_t)b * d + o], a.fw_x[(size_t)b * d + i], acc);
    }
    part[seg][c] = acc;
    if (c == 0) bpart[seg] = bacc;
    __syncthreads();
    if (seg == 0 && live) {
      float s8 = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) s8 += part[k][c];
      a.g_fs_w[g] += s8;
      if (fsb && i == 0) {                       // column o of the bias gradient: this workgroup's lane 0 owns it
        float sb = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) sb += bpart[k];
        a.g_fs_b[o] += sb;
      }
    }
    return;
  }
  if ((int)blockIdx.x < nq + nfw + nfold) {
    // parked column sums (ColFoldList): 32 output columns per workgroup, the parked partials split over 8 lane
    // groups (short dependent chains) and reduced through LDS, like the f_W gradient above
    __shared__ float fpart[8][32];
    int blk = (int)blockIdx.x - nq - nfw;
    for (int k = 0; k < a.fold.n; ++k) {
      const ColFold& f = a.fold.e[k];
      const int span = (3 * f.d + 31) / 32;
      if (blk >= span) { blk -= span; continue; }
      const int o = blk * 32 + c, seg = tid >> 5;
      const bool live = o < 3 * f.d;
      const int which = live ? o / f.d : 0, col = live ? o - which * f.d : 0;
      float* dst = f.dst[which];
      float sum = 0.f;
      if (live && dst) {
#pragma unroll 16
        for (int b2 = seg; b2 < f.nblk; b2 += 8) sum += f.partial[((size_t)b2 * 3 + which) * f.d + col];
      }
      fpart[seg][c] = sum;
      __syncthreads();
      if (seg == 0 && live && dst) {
        float s8 = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) s8 += fpart[q][c];
        dst[col] += s8;
      }
      return;
    }
    return;
  }
  const int t = ((int)blockIdx.x - nq - nfw - nfold) * 8 + (tid >> 5);
  if (t >= ntask) return;
  const int nitem = a.tem ? a.B * a.L : 0;
  if (t < nitem) {
    int b = t / a.L, l = t - b * a.L;
    int64_t idx = a.ui[t];
    if (idx == a.P || idx < 0 || idx > a.P) return;
    const float* src = a.dx + ((size_t)b * a.S + 1 + l) * d;
    float* dst = a.g_hist_tab + (size_t)idx * d;
    if (a.dx2) {                                  // (kernel-uniform) the row arrives as two partials: AttnArgs::dxp
      const float* src2 = a.dx2 + ((size_t)b * a.S + 1 + l) * d;
      switch (epl) {
        case 1: row_fetch_add2<1>(dst, src, src2, c); break;
        case 2: row_fetch_add2<2>(dst, src, src2, c); break;
        case 4: row_fetch_add2<4>(dst, src, src2, c); break;
        case 8: row_fetch_add2<8>(dst, src, src2, c); break;
        default: for (int k = 0; k < epl; ++k) atomicAdd(&dst[c + 32 * k], src[c + 32 * k] + src2[c + 32 * k]);
      }
      return;
    }
    // fetch the whole row, then add it: in a `load, add` loop every load waits for the previous group's ATOMIC to be
    // acknowledged (one in-order counter tracks both) — epl dependent round trips per row instead of one
    switch (epl) {
      case 1: row_fetch_add<1>(dst, src, c, 1.f); break;
      case 2: row_fetch_add<2>(dst, src, c, 1.f); break;
      case 4: row_fetch_add<4>(dst, src, c, 1.f); break;
      case 8: row_fetch_add<8>(dst, src, c, 1.f); break;
      default: for (int k = 0; k < epl; ++k) atomicAdd(&dst[c + 32 * k], src[c + 32 * k]);
    }
  } else {
    int u = t - nitem;
    int b = u / a.Q;
    int64_t idx = a.qw[u];
    if (idx == a.V - 1 || idx < 0 || idx >= a.V) return;
    // non-pad words of the query: lane q of the half-wave tests word q (one round trip; a counting loop was Q of them)
    int cnt = 0;
    for (int q0 = 0; q0 < a.Q; q0 += 32) {
      const bool w = q0 + c < a.Q && a.qw[(size_t)b * a.Q + min(q0 + c, a.Q - 1)] != a.V - 1;
      cnt += __popc((uint32_t)(__ballot(w) >> (tid & 32)));
    }
    const float inv = 1.f / (float)(cnt > 0 ? cnt : 1);
    const float* src = a.dqmean_d + (size_t)b * d;
    float* dst = a.g_word_emb + (size_t)idx * d;
    if (a.drop_fs.thr == 0u && (epl == 4 || epl == 1 || epl == 2 || epl == 8)) {
      switch (epl) {
        case 1: row_fetch_add<1>(dst, src, c, inv); break;
        case 2: row_fetch_add<2>(dst, src, c, inv); break;
        case 4: row_fetch_add<4>(dst, src, c, inv); break;
        default: row_fetch_add<8>(dst, src, c, inv); break;
      }
    } else if (epl == 4) {
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = src[c + 32 * k];
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(&dst[c + 32 * k], v[k] * drop_mult(a.drop_fs, (uint32_t)b, (uint32_t)(c + 32 * k)) * inv);
    } else {
      for (int k = 0; k < epl; ++k) {
        int e = c + 32 * k;
        atomicAdd(&dst[e], src[e] * drop_mult(a.drop_fs, (uint32_t)b, (uint32_t)e) * inv);
      }
    }
  }
}

// Deterministic form of the two scatters above (see score_bwd_det_kernel): history rows, then query-word rows.
// keys of the two task lists (history slots, query-word slots): int32 row ids, -1 = no task
__global__ __launch_bounds__(256) void embed_scatter_det_keys_kernel(const EmbedBwdArgs a, int32_t* keys) {
  const int g = (int)blockIdx.x * 256 + (int)threadIdx.x, gn = (int)gridDim.x * 256;
  const int nh = a.tem ? a.B * a.L : 0;
  for (int t = g; t < nh; t += gn) { const int64_t idx = a.ui[t]; keys[t] = (idx == a.P || idx < 0 || idx > a.P) ? -1 : (int32_t)idx; }
  for (int u = g; u < a.B * a.Q; u += gn) { const int64_t idx = a.qw[u]; keys[nh + u] = (idx == a.V - 1 || idx < 0 || idx >= a.V) ? -1 : (int32_t)idx; }
}
template <int EPL>
__global__ __launch_bounds__(32 * SBD_OWNERS_PER_WG) void embed_scatter_det_kernel(const EmbedBwdArgs a, const int32_t* keys) {
  typedef DetItemT<EPL> DetItem;
  const int nh = a.tem ? a.B * a.L : 0;
  const int tid = threadIdx.x, hl = tid & 31, c = hl;
  const int owner = (int)blockIdx.x * SBD_OWNERS_PER_WG + (tid >> 5), nown = (int)gridDim.x * SBD_OWNERS_PER_WG;
  const int d = a.d, epl = d >> 5;
  if (a.tem)
    det_owner_walk<EPL>(a.B * a.L, owner, nown, hl, d, a.g_hist_tab, nullptr, -1, keys,
      [&](int t, DetItem& it) {
        const int b = t / a.L, l = t - b * a.L;
        it.row = keys[t];
        const float* src = a.dx + ((size_t)b * a.S + 1 + l) * d;
#pragma unroll
        for (int k = 0; k < EPL; ++k) it.v[k] = k < epl ? src[c + 32 * k] : 0.f;
        it.bias = 0.f;
      });
  det_owner_walk<EPL>(a.B * a.Q, owner, nown, hl, d, a.g_word_emb, nullptr, -1, keys + nh,
    [&](int u, DetItem& it) {
      const int b = u / a.Q;
      it.row = keys[nh + u];
      it.bias = 0.f;
      if (a.det_dm) {
        const float* src = a.det_dm + (size_t)b * d;
#pragma unroll
        for (int k = 0; k < EPL; ++k) it.v[k] = k < epl ? src[c + 32 * k] : 0.f;
      } else {
        int cnt = 0;
        for (int q = 0; q < a.Q; ++q) cnt += (a.qw[(size_t)b * a.Q + q] != a.V - 1);
        const float inv = 1.f / (float)(cnt > 0 ? cnt : 1);
        const float* src = a.dqmean_d + (size_t)b * d;
#pragma unroll
        for (int k = 0; k < EPL; ++k)
          it.v[k] = k < epl ? src[c + 32 * k] * drop_mult(a.drop_fs, (uint32_t)b, (uint32_t)(c + 32 * k)) * inv : 0.f;
      }
    });
}

// table[keys[t]] += [scale[t] *] src[(rowidx ? rowidx[t] : t) * ld .. + d)  for t < ntask (keys[t] < 0: no task), deterministically: every table row has one
// owner half-wave, which walks the tasks in task order (det_owner_walk) — the review transformer's user / item embedding
// gradients in deterministic mode (rtm.hip)
template <int EPL>
__global__ __launch_bounds__(32 * SBD_OWNERS_PER_WG) void rows_scatter_det_kernel(const int32_t* keys, int ntask, const float* src,
                                                                                 int64_t ld, int d, float* table, const float* scale,
                                                                                 const int32_t* rowidx) {
  typedef DetItemT<EPL> DetItem;
  const int tid = threadIdx.x, hl = tid & 31, c = hl;
  const int owner = (int)blockIdx.x * SBD_OWNERS_PER_WG + (tid >> 5), nown = (int)gridDim.x * SBD_OWNERS_PER_WG;
  const int epl = d >> 5;
  det_owner_walk<EPL>(ntask, owner, nown, hl, d, table, nullptr, -1, keys,
    [&](int t, DetItem& it) {
      it.row = keys[t];
      it.bias = 0.f;
      const float* s = src + (size_t)(rowidx ? rowidx[t] : t) * ld;
      const float sc = scale ? scale[t] : 1.f;
#pragma unroll
      for (int k = 0; k < EPL; ++k) it.v[k] = k < epl ? (scale ? sc * s[c + 32 * k] : s[c + 32 * k]) : 0.f;
    });
}
int launch_rows_scatter_det(const int32_t* keys, int ntask, const float* src, int64_t ld, int d, float* table, hipStream_t st,
                            const float* scale, const int32_t* rowidx) {
  PS_REQUIRE(d % 32 == 0 && d <= 32 * BW_MAXE, "deterministic row scatter: d=%d unsupported", d);
  if (ntask <= 0) return PS_OK;
  if (d <= 128) hipLaunchKernelGGL(rows_scatter_det_kernel<4>, dim3(256), dim3(32 * SBD_OWNERS_PER_WG), 0, st, keys, ntask, src, ld, d, table, scale, rowidx);
  else if (d <= 256) hipLaunchKernelGGL(rows_scatter_det_kernel<8>, dim3(256), dim3(32 * SBD_OWNERS_PER_WG), 0, st, keys, ntask, src, ld, d, table, scale, rowidx);
  else hipLaunchKernelGGL(rows_scatter_det_kernel<16>, dim3(256), dim3(32 * SBD_OWNERS_PER_WG), 0, st, keys, ntask, src, ld, d, table, scale, rowidx);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

int launch_embed_scatter(const EmbedBwdArgs& a, hipStream_t st) {
  PS_REQUIRE(a.d % 32 == 0, "embed scatter: d %% 32");
  const bool fsb = a.fsb_w != nullptr;
  PS_REQUIRE(!fsb || (a.fsb_dqe && a.fsb_qe && a.g_fs_b && a.g_fs_w && a.fw_x && a.d <= 1024),
             "embed scatter: fused FS backward operands missing");
  const bool det = ps_deterministic();
  PS_REQUIRE(!det || !fsb || a.det_dm, "embed scatter: deterministic mode needs the d-mean buffer");
  PS_REQUIRE(!det || a.d <= 32 * BW_MAXE, "embed scatter: deterministic mode supports d <= %d", 32 * BW_MAXE);
  int ntask = (a.tem ? a.B * a.L : 0) + (fsb ? 0 : a.B * a.Q);   // fused: the query words are scattered by the row workgroups
  if (det) ntask = 0;                                            // ... deterministic mode: by the sole-owner pass below
  static const int parts = ps_diag_int("PS_SCATTER_PARTS", 15);   // timing experiments (WRONG results): 1 FS rows, 2 f_W gradient, 4 folds, 8 scatter tasks
  if (!(parts & 8)) ntask = 0;
  const int nsb = ps_cdiv(ntask, 8);
  const int nq = fsb && (parts & 1) ? a.B : 0;
  const int nfw = a.g_fs_w && (parts & 2) && !(fsb && a.fsb_dqpre_out) ? ps_cdiv(a.d * a.d, 32) : 0;
  PS_REQUIRE(!a.g_fs_w || fsb || (a.fw_dy && a.fw_x), "embed scatter: f_W gradient operands missing");
  int nfold = 0;
  for (int k = 0; k < a.fold.n; ++k) nfold += ps_cdiv(3 * a.fold.e[k].d, 32);
  if (!(parts & 4)) nfold = 0;
  const size_t lds = fsb ? sizeof(float) * (size_t)(256 / (a.d / 4) + 2) * a.d : 0;
  EmbedBwdArgs a2 = a;
  if (!det) a2.det_dm = nullptr;
  a2.sig = nullptr; a2.sigval = 0;
  PS_REQUIRE(!a.dx2 || (!det && (!fsb || a.fsb_dqe == a.dx)), "embed scatter: a second dx partial needs the default path with d query_emb = row 0 of dx");
  if (fsb) {       // d query_emb as two partials (AttnArgs::dxp): always two reads, the second weighted 0 when there is one buffer
    a2.fsb_dqe2 = a.dx2 ? a.dx2 : a.fsb_dqe;
    a2.fsb_k2 = a.dx2 ? 1.f : 0.f;
  }
  if (nq + nsb + nfw + nfold > 0) {
    side_take_signal(st, &a2.sig, &a2.sigval);        // (every check is behind us: the launch happens)
    hipLaunchKernelGGL(embed_scatter_kernel, dim3(nq + nsb + nfw + nfold), dim3(256), lds, st, a2, ntask, nq, nfw, nfold);
    PS_LAUNCH_CHECK();
  }
  if (det) {
    const size_t nkeys = (size_t)a.B * ((a.tem ? a.L : 0) + a.Q);
    int32_t* keys = reinterpret_cast<int32_t*>(ps_det_scratch(1, nkeys + 4, st));   // (the score backward's use of the slot is over)
    PS_REQUIRE(keys, "embed scatter: deterministic mode has no scratch (allocation failed or stream capture)");
    hipLaunchKernelGGL(embed_scatter_det_keys_kernel, dim3(ps_cdiv((int64_t)nkeys, 1024)), dim3(256), 0, st, a2, keys);
    PS_LAUNCH_CHECK();
    if (a.d <= 128) hipLaunchKernelGGL(embed_scatter_det_kernel<4>, dim3(256), dim3(32 * SBD_OWNERS_PER_WG), 0, st, a2, keys);
    else if (a.d <= 256) hipLaunchKernelGGL(embed_scatter_det_kernel<8>, dim3(256), dim3(32 * SBD_OWNERS_PER_WG), 0, st, a2, keys);
    else hipLaunchKernelGGL(embed_scatter_det_kernel<16>, dim3(256), dim3(32 * SBD_OWNERS_PER_WG), 0, st, a2, keys);
    PS_LAUNCH_CHECK();
  }
  return PS_OK;
}

// dqpre = dqe * (1 - qe^2)  (tanh backward of FSEncoder, text_encoder.py:39), dfb += colsum
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* dqe, int lddqe, const float* qe, float* dqpre,
                                                       float* dfb, int rows, int d) {
  extern __shared__ float csum[];                // [d]
  const int r0 = blockIdx.x * 8, nr = min(8, rows - r0);
  for (int c = threadIdx.x; c < d; c += 256) csum[c] = 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < nr * d; i += 256) {
    const int r = r0 + i / d, col = i % d;
    const float y = qe[(size_t)r * d + col];
    const float v = dqe[(size_t)r * lddqe + col] * (1.f - y * y);
    dqpre[(size_t)r * d + col] = v;
    atomicAdd(&csum[col], v);                    // LDS atomic: 8 adders per column
  }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += 256) atomicAdd(&dfb[c], csum[c]);
}

// The residual path  out = dropout(context) + inputs  feeds every replica's d y1 back into its sequence's ONE query row:
// d x[b, qpos] += sum_j d y1[b * fan + j].  As the dX product's RES_FANIN epilogue that walk (21 rows per query row at C5) forced
// the dense form of the product over all B * S positions; summed here first (22 MB read at C5, one float4 per thread and replica,
// all of a thread's loads in flight together) the product runs over the VALID positions only, as at d = 128.
__global__ __launch_bounds__(256) void fanin_sum_kernel(const float* __restrict__ src, int ld, int n_in, int fan, int d4,
                                                        float* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_in * d4) return;
  const int b = idx / d4, c = idx - b * d4;
  const float4* p = reinterpret_cast<const float4*>(src + (size_t)b * fan * ld) + c;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int j = 0;
  for (; j + 8 <= fan; j += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(j + u) * (ld >> 2)];
#pragma unroll
    for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  if (j < fan) {      // the last fan % 8 rows: one more batch of eight (rows past the fan re-read the last one with weight 0 — fma(v, 1, s)
                      // is s + v: the same sums; as a loop of one load per trip they were up to 7 more serial round trips, 5 at fan = 21)
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(j + u < fan ? j + u : fan - 1) * (ld >> 2)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float w = j + u < fan ? 1.f : 0.f;
      s.x = __builtin_fmaf(v[u].x, w, s.x); s.y = __builtin_fmaf(v[u].y, w, s.y); s.z = __builtin_fmaf(v[u].z, w, s.z); s.w = __builtin_fmaf(v[u].w, w, s.w);
    }
  }
  reinterpret_cast<float4*>(out)[idx] = s;
}
int launch_fanin_sum(const float* src, int ld, int n_in, int fan, int d, float* out, hipStream_t st) {
  PS_REQUIRE(src && out && n_in > 0 && fan > 0 && d % 4 == 0 && ld % 4 == 0, "fanin_sum: bad argument");
  hipLaunchKernelGGL(fanin_sum_kernel, dim3(ps_cdiv((int64_t)n_in * (d / 4), 256)), dim3(256), 0, st, src, ld, n_in, fan, d / 4, out);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

int launch_tanh_bwd(const float* dqe, int lddqe, const float* qe, float* dqpre, float* dfb, int rows, int d,
                    hipStream_t st) {
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3(ps_cdiv(rows, 8)), dim3(256), (size_t)d * sizeof(float), st, dqe, lddqe,
                     qe, dqpre, dfb, rows, d);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

// ============================================================ negative sampling
// Stand-in for the two torch.multinomial draws (item_transformer.py:447 uniform items over
// [0,P), :268 words ~ word_dists via a Vose alias table); Philox streams keyed by (seed, step).
__global__ __launch_bounds__(256) void sample_kernel(int nitem, int nword, int64_t P, int64_t V, uint32_t step,
                                                     uint32_t k0, uint32_t k1, const float* prob,
                                                     const int32_t* alias, int64_t* neg_items, int64_t* neg_words) {
  int t = blockIdx.x * 256 + threadIdx.x;
  if (t < nitem) {
    Philox4 r = philox4x32_10((uint32_t)t, 0u, PS_SITE_SAMPLE_ITEM, step, k0, k1);
    neg_items[t] = (int64_t)(((uint64_t)r.x * (uint64_t)P) >> 32);
  } else if (t < nitem + nword) {
    int u = t - nitem;
    Philox4 r = philox4x32_10((uint32_t)u, 0u, PS_SITE_SAMPLE_WORD, step, k0, k1);
    int64_t i = (int64_t)(((uint64_t)r.x * (uint64_t)V) >> 32);
    float f = (float)(r.y >> 8) * (1.0f / 16777216.0f);
    neg_words[u] = f < prob[i] ? i : (int64_t)alias[i];
  }
}

__global__ __launch_bounds__(256) void tem_stage_kernel(StageArgs a) {
  int t = blockIdx.x * 256 + threadIdx.x;
  if (t == 0) *a.step_word = a.step;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int n = a.src[k] ? a.n[k] : 0;
    if (t < n) { a.dst[k][t] = a.src[k][t]; return; }
    t -= n;
  }
  if (!a.prob) return;
  if (t < a.nitem) {
    Philox4 r = philox4x32_10((uint32_t)t, 0u, PS_SITE_SAMPLE_ITEM, a.step, a.k0, a.k1);
    a.dst[4][t] = (int64_t)(((uint64_t)r.x * (uint64_t)a.P) >> 32);
  } else if (t < a.nitem + a.nword) {
    const int u = t - a.nitem;
    Philox4 r = philox4x32_10((uint32_t)u, 0u, PS_SITE_SAMPLE_WORD, a.step, a.k0, a.k1);
    const int64_t i = (int64_t)(((uint64_t)r.x * (uint64_t)a.V) >> 32);
    const float f = (float)(r.y >> 8) * (1.0f / 16777216.0f);
    a.dst[5][u] = f < a.prob[i] ? i : (int64_t)a.alias[i];
  }
}

const void* stage_kernel_handle() { return reinterpret_cast<const void*>(tem_stage_kernel); }

int launch_stage(const StageArgs& a, hipStream_t st) {
  int total = a.prob ? a.nitem + a.nword : 0;
  for (int k = 0; k < 6; ++k) total += a.src[k] ? a.n[k] : 0;
  hipLaunchKernelGGL(tem_stage_kernel, dim3(ps_cdiv(total > 0 ? total : 1, 256)), dim3(256), 0, st, a);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

int launch_sample(const PsTemDesc& d, const float* alias_prob, const int32_t* alias_idx, int64_t* neg_items,
                  int64_t* neg_words, hipStream_t st) {
  int nitem = d.B * d.K, nword = d.B * d.W * d.K;
  hipLaunchKernelGGL(sample_kernel, dim3(ps_cdiv(nitem + nword, 256)), dim3(256), 0, st, nitem, nword,
                     d.product_size, d.vocab_size, (uint32_t)d.step, (uint32_t)(d.seed & 0xffffffffu),
                     (uint32_t)(d.seed >> 32), alias_prob, alias_idx, neg_items, neg_words);
  PS_LAUNCH_CHECK();
  return PS_OK;
}
