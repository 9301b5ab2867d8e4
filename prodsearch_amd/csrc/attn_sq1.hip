// attn_sq1.hip — attention of the LAST encoder layer, where only ONE query position per
// sequence is consumed (x[:, 0] or x[:, -1], item_transformer.py:482-492): one workgroup
// (4 waves) per input sequence, all heads at once, K/V rows staged once in LDS and shared by
// every dropout replica of that sequence (MultiHeadedAttention.forward, neural.py:206-231).
//
// Forward : scores[h][s] = q_h . K_s,h ; masked softmax ; per replica j: dropout(P) . V
// Backward: per replica j: dV += (P*m_j)^T dctx_j ; dP += m_j * (dctx_j . V) ; then softmax
//           backward, dq = dS . K / sqrt(dh), dK = dS^T q, bias gradients.
// Everything is LDS-resident fp32 VALU work (S <= 64, a few KB per sequence); replicas are
// processed in chunks of JC so that the Philox masks are generated once per element.
#include "rowwise.h"
#include "x3frag.h"
#include <stdlib.h>

#define SQ1_LDS_MAX (128 * 1024)   // dynamic LDS these kernels may request (160 KB per CU on gfx950)

// replicas are processed JC at a time (all of them when the LDS budget allows: one global round trip)
static inline int sq1_pick_jc(int S, int d, int H, int fan);
#define KLD(d) ((d) + 1)     // K/V rows are padded by one float in LDS: threads that walk over keys s at a
                             // fixed column would otherwise all hit one bank (row stride d = 0 mod 32)

struct Sq1Lds {
  float *Ks, *Vs, *q, *P, *valid, *Pd, *dC, *dV, *dP;
  int* spos;      // [S] original positions of the valid keys, ascending; then 8 ints of per-sequence meta (Sq1Meta)
};
// The kernels work on the VALID key positions of their sequence only (the query column and the non-pad history /
// review positions: 31 % of S at C2, 27 % on the review transformer): K / V rows k < Sv in LDS are positions spos[k].
struct Sq1Meta { int Sv; FDiv fS, fHS, fHQS; };   // dividers of Sv, H*Sv, max(H/4,1)*Sv

__device__ inline Sq1Lds sq1_carve(float* base, int S, int d, int H, bool bwd, int JC) {
  Sq1Lds l;
  l.Ks = base; base += S * KLD(d);
  l.Vs = base; base += S * KLD(d);
  l.q = base; base += (d + 3) & ~3;
  l.P = base; base += H * (S + 1);
  l.valid = base; base += (S + 3) & ~3;
  l.spos = reinterpret_cast<int*>(base); base += ((S + 3) & ~3) + 8;
  l.Pd = base; base += JC * H * (S + 1);
  l.dC = base; l.dV = base; l.dP = base;
  if (bwd) {
    l.dC = base; base += JC * d;
    l.dV = base; base += S * d;
    l.dP = base;
  }
  return l;
}
static inline size_t sq1_lds_bytes(int S, int d, int H, bool bwd, int JC) {
  size_t n = (size_t)2 * S * KLD(d) + ((d + 3) & ~3) + H * (S + 1) + 2 * ((S + 3) & ~3) + 8 + (size_t)JC * H * (S + 1);
  if (bwd) n += (size_t)JC * d + (size_t)S * d + H * (S + 1);
  return n * sizeof(float);
}

// A workgroup owns the heads h0 .. h0+H-1 of one sequence = columns c0 .. c0+d-1 of its rows (heads are independent
// in attention); `d`, `H` below are those SUB sizes, D the row stride of the global tensors.
__device__ inline FDiv dev_fdiv(int d) {
  FDiv f;
  f.d = (uint32_t)(d > 0 ? d : 1);
  f.m = f.d == 1 ? 0u : (uint32_t)((0x100000000ull + f.d - 1) / f.d);
  return f;
}
// valid flags -> position list (wave 0, one ballot: S <= 64) -> K / V rows of the valid positions, q.  Ends with a
// barrier; returns the sequence's meta.
__device__ inline Sq1Meta sq1_load(const AttnArgs& a, const Sq1Lds& l, int b, int tid, int d, int c0, int H) {
  const int S = a.S, D = a.d;
  const int d4 = d >> 2;
  const int brow = b / a.seq_div;
  Sq1Meta* meta = reinterpret_cast<Sq1Meta*>(l.spos + ((S + 3) & ~3));
  if (tid < 64) {
    const bool v = tid < S && (a.valid ? a.valid[(size_t)brow * S + tid] != 0.f
                                       : (tid == 0 || a.ui[(size_t)brow * a.L + tid - 1] != a.P));
    const unsigned long long vm = __ballot(v);
    if (tid < S) l.valid[tid] = v ? 1.f : 0.f;
    if (v) l.spos[__popcll(vm & ((1ull << tid) - 1ull))] = tid;
    if (tid == 0) {
      const int Sv = __popcll(vm);
      meta->Sv = Sv; meta->fS = dev_fdiv(Sv); meta->fHS = dev_fdiv(H * Sv); meta->fHQS = dev_fdiv((H / 4 > 0 ? H / 4 : 1) * Sv);
    }
  }
  for (int i = tid; i < d; i += 256) l.q[i] = a.qp[(size_t)b * D + c0 + i];
  __syncthreads();
  const Sq1Meta m = *meta;
  for (int i = tid; i < m.Sv * d4; i += 256) {
    const int k = fdiv(i, a.fd4), c4 = (i - k * d4) * 4;
    const size_t row = (size_t)b * S + l.spos[k];
    const float4 kk = *reinterpret_cast<const float4*>(a.kp + row * D + c0 + c4);
    const float4 vv = *reinterpret_cast<const float4*>(a.vp + row * D + c0 + c4);
    float* lk = l.Ks + k * KLD(d) + c4;
    float* lv = l.Vs + k * KLD(d) + c4;
    lk[0] = kk.x; lk[1] = kk.y; lk[2] = kk.z; lk[3] = kk.w;
    lv[0] = vv.x; lv[1] = vv.y; lv[2] = vv.z; lv[3] = vv.w;
  }
  __syncthreads();
  return m;
}

// dropout multipliers (or P * multipliers) of replicas j0..j0+nj-1 into Pd[jj][h][s]
__device__ inline void sq1_masks(const AttnArgs& a, const Sq1Lds& l, const Sq1Meta& mt, int b, int j0, int nj, int tid,
                                 bool times_p, int H, int h0) {
  const int S = mt.Sv, LD = a.S + 1, HF = a.H, per = H * S;
  if (a.drop.thr == 0u) {
    for (int i = tid; i < nj * per; i += 256) {
      const int jj = fdiv(i, mt.fHS), r = i - jj * per, h = fdiv(r, mt.fS), k = r - h * S;
      l.Pd[(jj * H + h) * LD + k] = times_p ? l.P[h * LD + k] : 1.f;
    }
    return;
  }
  if ((H & 3) == 0) {
    // rows nout*H + 4g .. +3 share one Philox counter (col = s, row >> 2): one call serves 4 heads
    const int HQ = H >> 2, perq = HQ * S;
    for (int i = tid; i < nj * perq; i += 256) {
      const int jj = fdiv(i, mt.fHQS), r = i - jj * perq, g = fdiv(r, mt.fS), k = r - g * S;
      const uint32_t row = (uint32_t)((b * a.fan + j0 + jj) * HF + h0 + 4 * g);
      const Philox4 w = philox4x32_10((uint32_t)l.spos[k], row >> 2, a.drop.site, drop_step(a.drop), a.drop.k0, a.drop.k1);
      const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int h = 4 * g + q;
        float m = drop_word(a.drop, ws[q]);
        if (times_p) m *= l.P[h * LD + k];
        l.Pd[(jj * H + h) * LD + k] = m;
      }
    }
    return;
  }
  for (int i = tid; i < nj * per; i += 256) {
    const int jj = fdiv(i, mt.fHS), r = i - jj * per, h = fdiv(r, mt.fS), k = r - h * S;
    const uint32_t row = (uint32_t)((b * a.fan + j0 + jj) * HF + h0 + h);  // Sq == 1: row = nout*H + h
    float m = drop_mult(a.drop, row, (uint32_t)l.spos[k]);
    if (times_p) m *= l.P[h * LD + k];
    l.Pd[(jj * H + h) * LD + k] = m;
  }
}

__global__ __launch_bounds__(256) void attn_fwd_sq1_kernel(const AttnArgs a) {
  extern __shared__ float lds[];
  const int SF = a.S, LD = SF + 1, D = a.d, HF = a.H, dh = a.dh, tid = threadIdx.x, b = blockIdx.x;
  const int H = HF / (int)gridDim.y, d = D / (int)gridDim.y, h0 = (int)blockIdx.y * H, c0 = (int)blockIdx.y * d;
  const int JC = a.jc;
  Sq1Lds l = sq1_carve(lds, SF, d, H, false, JC);
  const Sq1Meta mt = sq1_load(a, l, b, tid, d, c0, H);
  const int S = mt.Sv;                                      // valid key positions of this sequence
  for (int i = tid; i < H * S; i += 256) {
    const int h = fdiv(i, mt.fS), k = i - h * S;
    float acc = 0.f;
#pragma unroll 8
    for (int c = 0; c < dh; ++c) acc += l.q[h * dh + c] * l.Ks[k * KLD(d) + h * dh + c];
    l.P[h * LD + k] = acc;                                  // masked positions (masked_fill -1e18 -> weight 0) are not listed
  }
  __syncthreads();
  for (int h = tid >> 6; h < H; h += 4) {                    // softmax: one wave per head, lane = key (S <= 64)
    float* p = l.P + h * LD;
    const int k = tid & 63;
    const float v = k < S ? p[k] : -INFINITY;
    const float m = wave_max(v);
    const float e = k < S ? expf(v - m) : 0.f;
    const float inv = 1.f / wave_sum(e);
    if (k < S) p[k] = e * inv;
  }
  __syncthreads();
  for (int i = tid; i < H * SF; i += 256) {                  // attention weights of ALL positions (0 at the masked ones)
    const int h = fdiv(i, a.fS), s = i - h * SF;
    if (l.valid[s] == 0.f) a.attn[((size_t)b * HF + h0 + h) * SF + s] = 0.f;
  }
  for (int i = tid; i < H * S; i += 256) {
    const int h = fdiv(i, mt.fS), k = i - h * S;
    a.attn[((size_t)b * HF + h0 + h) * SF + l.spos[k]] = l.P[h * LD + k];
  }
  for (int j0 = 0; j0 < a.fan; j0 += JC) {
    const int nj = min(JC, a.fan - j0);
    __syncthreads();
    sq1_masks(a, l, mt, b, j0, nj, tid, true, H, h0);
    __syncthreads();
    for (int i = tid; i < nj * d; i += 256) {
      const int jj = fdiv(i, a.fd), c = i - jj * d, h = fdiv(c, a.fdh);
      const float* pd = l.Pd + (jj * H + h) * LD;
      float acc = 0.f;
#pragma unroll 8
      for (int k = 0; k < S; ++k) acc += pd[k] * l.Vs[k * KLD(d) + c];
      a.ctx[((size_t)b * a.fan + j0 + jj) * D + c0 + c] = acc;
    }
  }
}

static inline int sq1_pick_jc(int S, int d, int H, int fan) {
  int jc = fan < 24 ? fan : 24;
  while (jc > 1 && sq1_lds_bytes(S, d, H, true, jc) > SQ1_LDS_MAX) jc = (jc + 1) / 2;
  return jc;
}
bool attn_sq1_fits(const AttnArgs& a) {
  return a.Sq == 1 && a.S <= 64 && a.d % 4 == 0 && sq1_lds_bytes(a.S, a.d, a.H, true, 1) <= SQ1_LDS_MAX;
}

// head groups per sequence (grid.y): heads are independent, so a sequence's work can be cut into workgroups that each
// walk the same latency chain over fewer columns; groups of 4 heads keep the shared Philox call of sq1_masks intact
static inline int sq1_pick_split(const AttnArgs& a) {
  static const int env = ps_diag_int("PS_ATTN_SPLIT", 0);     // tuning experiments
  int hy = env > 0 ? env : 2;
  while (hy > 1 && (a.H % hy != 0 || (a.H / hy) % 4 != 0 || (a.d / hy) % 4 != 0)) hy >>= 1;
  return hy < 1 ? 1 : hy;
}
int attn_sq1_split(const AttnArgs& a) { return sq1_pick_split(a); }
static inline void sq1_sub_dividers(AttnArgs& b, int hy) {
  const int Hs = b.H / hy, ds = b.d / hy;
  b.fd = make_fdiv(ds); b.fd4 = make_fdiv(ds / 4);
  b.fHS = make_fdiv(Hs * b.S); b.fHQS = make_fdiv((Hs / 4 > 0 ? Hs / 4 : 1) * b.S);
}

int launch_attn_fwd_sq1(const AttnArgs& a, hipStream_t st) {
  AttnArgs b = a;
  const int hy = sq1_pick_split(a);
  sq1_sub_dividers(b, hy);
  b.jc = sq1_pick_jc(a.S, a.d / hy, a.H / hy, a.fan);
  size_t lds = sq1_lds_bytes(a.S, a.d / hy, a.H / hy, false, b.jc);
  PS_REQUIRE(a.Sq == 1 && a.S <= 64 && a.d % 4 == 0 && lds <= SQ1_LDS_MAX, "attention(sq1): S=%d d=%d needs %zu B LDS",
             a.S, a.d, lds);
  static bool attr_f = false;
  if (!attr_f) {
    PS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_sq1_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, SQ1_LDS_MAX));
    attr_f = true;
  }
  hipLaunchKernelGGL(attn_fwd_sq1_kernel, dim3(a.n_in, hy), dim3(256), lds, st, b);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

__global__ __launch_bounds__(256) void attn_bwd_sq1_kernel(const AttnArgs a) {
  fork_signal(a.sig, a.sigval);
  extern __shared__ float lds[];
  const int SF = a.S, LD = SF + 1, D = a.d, HF = a.H, dh = a.dh, tid = threadIdx.x, b = blockIdx.x;
  const int H = HF / (int)gridDim.y, d = D / (int)gridDim.y, h0 = (int)blockIdx.y * H, c0 = (int)blockIdx.y * d;
  const int JC = a.jc;
  Sq1Lds l = sq1_carve(lds, SF, d, H, true, JC);
  // folded dQ.Wq (AttnArgs::wq; D == 128, d == 64): thread (half, i) owns output column i and 32 of this group's 64
  // query features; its 32 weights are requested now and used after the whole backward, ~30 us later
  const bool fold_q = a.wq != nullptr;
  float wq[32];
  if (fold_q) {
    const int i = tid & 127, half = tid >> 7;
#pragma unroll
    for (int k = 0; k < 32; ++k) wq[k] = a.wq[(size_t)(c0 + half * 32 + k) * 128 + i];
  }
  const Sq1Meta mt = sq1_load(a, l, b, tid, d, c0, H);
  const int S = mt.Sv;                                      // valid key positions; SF = all of them (global strides)
  for (int i = tid; i < H * S; i += 256) {
    const int h = fdiv(i, mt.fS), k = i - h * S;
    l.P[h * LD + k] = a.attn[((size_t)b * HF + h0 + h) * SF + l.spos[k]];
    l.dP[h * LD + k] = 0.f;
  }
  for (int i = tid; i < S * d; i += 256) l.dV[i] = 0.f;
  for (int j0 = 0; j0 < a.fan; j0 += JC) {
    const int nj = min(JC, a.fan - j0);
    __syncthreads();
    sq1_masks(a, l, mt, b, j0, nj, tid, false, H, h0);
    for (int i = tid; i < nj * d; i += 256) {
      const int jj = fdiv(i, a.fd), c = i - jj * d;
      l.dC[i] = a.dctx[((size_t)b * a.fan + j0 + jj) * D + c0 + c];
    }
    __syncthreads();
    for (int i = tid; i < S * d; i += 256) {                 // dV[s][c] += sum_j P*m_j * dctx_j[c]
      const int s = fdiv(i, a.fd), c = i - s * d, h = fdiv(c, a.fdh);
      const float p = l.P[h * LD + s];
      float acc = 0.f;
#pragma unroll 8
      for (int jj = 0; jj < nj; ++jj) acc += l.Pd[(jj * H + h) * LD + s] * l.dC[jj * d + c];
      l.dV[i] += p * acc;
    }
    for (int i = tid; i < nj * H * S; i += 256) {            // dP[h][s] += m_j * (dctx_j,h . V_s,h), all (j,h,s) in parallel
      const int jj = fdiv(i, mt.fHS), r = i - jj * H * S, h = fdiv(r, mt.fS), s = r - h * S;
      const float m = l.Pd[(jj * H + h) * LD + s];
      if (m != 0.f) {
        float dot = 0.f;
#pragma unroll 8
        for (int c = 0; c < dh; ++c) dot += l.dC[jj * d + h * dh + c] * l.Vs[s * KLD(d) + h * dh + c];
        atomicAdd(&l.dP[h * LD + s], m * dot);              // LDS atomic, <= JC adders per element
      }
    }
  }
  __syncthreads();
  for (int h = tid >> 6; h < H; h += 4) {                    // softmax backward: one wave per head, lane = key position
    const float* p = l.P + h * LD;
    float* g = l.dP + h * LD;
    const int s = tid & 63;
    const float pv = s < S ? p[s] : 0.f, gv = s < S ? g[s] : 0.f;
    const float t = wave_sum(pv * gv);
    if (s < S) g[s] = pv * (gv - t);
  }
  __syncthreads();
  // dq / dK / dV of this group's columns: the key positions are split over nsg thread groups (all 256 threads busy
  // instead of d of them walking S positions), partial column sums combined through LDS
  int nsg = 256 / d;
  if (nsg < 1 || 3 * nsg > JC) nsg = 1;                     // the partials live in the (free) dC region: [3][nsg][d]
  float* red3 = l.dC;
  __syncthreads();
  for (int idx = tid; idx < nsg * d; idx += 256) {
    const int sg = idx / d, c = idx - sg * d;
    const int h = fdiv(c, a.fdh);
    const float* g = l.dP + h * LD;
    const float qc = l.q[c];
    float dq = 0.f, sk = 0.f, sv = 0.f;
    for (int s = sg; s < S; s += nsg) {
      dq += g[s] * l.Ks[s * KLD(d) + c];
      const float dk = g[s] * qc;
      const float dv = l.dV[s * d + c];
      const size_t off = ((size_t)b * SF + l.spos[s]) * a.lddkv + c0 + c;
      a.dkv[off] = dk;
      a.dkv[off + D] = dv;
      sk += dk; sv += dv;
    }
    for (int s = sg; s < SF; s += nsg)                       // masked positions: exact zeros (dense consumers read them)
      if (l.valid[s] == 0.f) {
        const size_t off = ((size_t)b * SF + s) * a.lddkv + c0 + c;
        a.dkv[off] = 0.f;
        a.dkv[off + D] = 0.f;
      }
    red3[(0 * nsg + sg) * d + c] = dq;
    red3[(1 * nsg + sg) * d + c] = sk;
    red3[(2 * nsg + sg) * d + c] = sv;
  }
  __syncthreads();
  for (int c = tid; c < d; c += 256) {
    float dq = 0.f, sk = 0.f, sv = 0.f;
    for (int g2 = 0; g2 < nsg; ++g2) { dq += red3[(0 * nsg + g2) * d + c]; sk += red3[(1 * nsg + g2) * d + c]; sv += red3[(2 * nsg + g2) * d + c]; }
    dq *= a.qscale;
    a.dq[(size_t)b * a.lddq + c0 + c] = dq;
    if (fold_q) l.q[c] = dq;                                 // q[c] is not read any more
    if (a.bias_part) {                                       // parked: folded by the step's last launch
      float* bp = a.bias_part + (size_t)b * 3 * D + c0 + c;
      bp[0] = dq; bp[D] = sk; bp[2 * D] = sv;
    } else {
      atomicAdd(&a.dbq[c0 + c], dq);
      atomicAdd(&a.dbk[c0 + c], sk);
      atomicAdd(&a.dbv[c0 + c], sv);
    }
  }
  if (!fold_q) return;
  __syncthreads();
  {
    const int i = tid & 127, half = tid >> 7;
    float acc = 0.f;
    if (a.fanin_src && (i >> 6) == (int)blockIdx.y)          // this group's 64 columns of the replicas' fan-in sum
      for (int j = half; j < a.fan; j += 2) acc += a.fanin_src[((size_t)b * a.fan + j) * 128 + i];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc = fmaf(l.q[half * 32 + k], wq[k], acc);
    float* part = l.dV;                                      // free: dK / dV have been written out
    __syncthreads();
    if (half == 1) part[i] = acc;
    __syncthreads();
    if (half == 0) a.dxq_part[((size_t)blockIdx.y * gridDim.x + b) * 128 + i] = acc + part[i];
  }
}

int launch_attn_bwd_sq1(const AttnArgs& a, hipStream_t st) {
  AttnArgs b = a;
  const int hy = sq1_pick_split(a);
  sq1_sub_dividers(b, hy);
  b.jc = sq1_pick_jc(a.S, a.d / hy, a.H / hy, a.fan);
  size_t lds = sq1_lds_bytes(a.S, a.d / hy, a.H / hy, true, b.jc);
  PS_REQUIRE(a.Sq == 1 && a.S <= 64 && a.d % 4 == 0 && lds <= SQ1_LDS_MAX, "attention bwd(sq1): S=%d d=%d needs %zu B LDS",
             a.S, a.d, lds);
  static bool attr_b = false;
  if (!attr_b) {
    PS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_sq1_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, SQ1_LDS_MAX));
    attr_b = true;
  }
  PS_REQUIRE(!a.wq || (a.dxq_part && a.d == 128 && hy == 2), "attention bwd(sq1): folded dQ.Wq needs d == 128 and two head groups");
  b.sig = nullptr; b.sigval = 0;
  side_take_signal(st, &b.sig, &b.sigval);             // (every check is behind us: the launch happens)
  hipLaunchKernelGGL(attn_bwd_sq1_kernel, dim3(a.n_in, hy), dim3(256), lds, st, b);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

// ================================================================== one wave per sequence (fan == 1)
// Without dropout replicas (the review transformer: 1,536 sequences of 51 positions, 27 % of them real; TEM without
// dropout) a sequence's attention is ~14 keys x 128 columns of work: the workgroup form above spends its time in barriers
// and LDS staging (71 us for the C4 backward, 4 rounds of 3 workgroups per CU).  Here a WAVE owns a sequence, all heads:
// a K / V row is read as one float4 per lane (LPR = d/4 lanes per row, 64/LPR keys per step), a head's dot product is a
// reduction over its dh/4 neighbouring lanes, the per-(key, head) softmax terms wait in 4 KB of LDS between the two passes,
// and nothing synchronises until the workgroup's four sequences multiply their dq rows into Wq together (AttnArgs::wq).
#define W1_MAXH 8
__device__ inline float4 f4_ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ inline void f4_st(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ inline float f4_dot(const float4& x, const float4& y) { return (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w); }
__device__ inline void f4_fma(float4& acc, float s, const float4& v) { acc.x = fmaf(s, v.x, acc.x); acc.y = fmaf(s, v.y, acc.y); acc.z = fmaf(s, v.z, acc.z); acc.w = fmaf(s, v.w, acc.w); }
__device__ inline float4 f4_scale(float s, const float4& v) { return make_float4(s * v.x, s * v.y, s * v.z, s * v.w); }
__device__ inline float4 f4_xor_add(const float4& v, int o) {
  return make_float4(v.x + __shfl_xor(v.x, o, 64), v.y + __shfl_xor(v.y, o, 64), v.z + __shfl_xor(v.z, o, 64), v.w + __shfl_xor(v.w, o, 64));
}
// valid key positions of sequence b: lane s holds the flag; returns the mask, writes the ascending position list
__device__ inline unsigned long long w1_valid(const AttnArgs& a, int b, int lane, int* sp) {
  const int brow = b / a.seq_div;
  const bool v = lane < a.S && (a.valid ? a.valid[(size_t)brow * a.S + lane] != 0.f
                                        : (lane == 0 || a.ui[(size_t)brow * a.L + lane - 1] != a.P));
  const unsigned long long vm = __ballot(v);
  if (v) sp[__popcll(vm & ((1ull << lane) - 1ull))] = lane;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return vm;
}

// A wave of these two kernels is alone on its SIMD (1,536 sequences on 1,024 SIMDs at configs[3]): its time is the number of
// dependent memory round trips.  W1_U key steps are loaded together — 16 keys per round whatever the row width — and PIN4 makes
// every loaded row arrive BEFORE the first conditional store of the round: behind such a store's join the compiler can name a
// later row's arrival only by a count that also covers the store, i.e. it waited for each store's acknowledgement in turn.
// (Two key steps per round and no pins: 13 + 13 rounds per sequence of 51 positions, 19.7 / 44 us.)
#define PIN4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))
template <int LPR>
__global__ __launch_bounds__(256) void attn_bwd_w1_kernel(const AttnArgs a, int pads_unread) {
  fork_signal(a.sig, a.sigval);
  constexpr int D = 4 * LPR, KPS = 64 / LPR;
  __shared__ int sp[4][64];
  __shared__ float pl[4][64][W1_MAXH], dpl[4][64][W1_MAXH];
  __shared__ float dqs[4][D];
  __shared__ float part[256 / D][4][D];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = (int)blockIdx.x * 4 + wv;
  const int S = a.S, HF = a.H, lph = a.dh >> 2;
  const int sub = lane / LPR, cl = lane % LPR, c = 4 * cl, h = cl / lph;
  if (b < a.n_in) {
    const unsigned long long vm = w1_valid(a, b, lane, sp[wv]);
    const int Sv = __popcll(vm);
    const float4 dc4 = f4_ld(a.dctx + (size_t)b * D + c);
    const float4 q4 = f4_ld(a.qp + (size_t)b * D + c);
    float4 dq4 = make_float4(0.f, 0.f, 0.f, 0.f), sk4 = dq4, sv4 = dq4;
    float th = 0.f;
    const uint32_t drow = (uint32_t)(b * HF + h);
    DropSpec drop = a.drop;                                  // the step word is read once, not per key
    drop.step = drop_step(a.drop); drop.step_ptr = nullptr;
    // (an idle lane group repeats key 0 — always a real row: the query position is valid — so every load is
    // unconditional; W1_U steps of loads are in flight together)
    constexpr int W1_U = 16 / KPS;
    for (int k0 = 0; k0 < Sv; k0 += KPS * W1_U) {
      int kk[W1_U], pp[W1_U]; bool on[W1_U]; float4 v4[W1_U]; float Pv[W1_U];
#pragma unroll
      for (int u = 0; u < W1_U; ++u) {
        kk[u] = k0 + u * KPS + sub;
        on[u] = kk[u] < Sv;
        pp[u] = sp[wv][on[u] ? kk[u] : 0];
        const size_t row = (size_t)b * S + pp[u];
        v4[u] = f4_ld(a.vp + row * D + c);
        Pv[u] = a.attn[((size_t)b * HF + h) * S + pp[u]];
      }
#pragma unroll
      for (int u = 0; u < W1_U; ++u) { PIN4(v4[u]); asm volatile("" : "+v"(Pv[u])); }
#pragma unroll
      for (int u = 0; u < W1_U; ++u) {
        const size_t row = (size_t)b * S + pp[u];
        const float P = on[u] ? Pv[u] : 0.f;
        const float m = !on[u] ? 0.f : (drop.thr ? drop_mult(drop, drow, (uint32_t)pp[u]) : 1.f);
        float dot = f4_dot(dc4, v4[u]);
        for (int o = 1; o < lph; o <<= 1) dot += __shfl_xor(dot, o, 64);
        const float dP = m * dot;
        th = fmaf(P, dP, th);
        if (on[u]) {
          const float4 dv4 = f4_scale(P * m, dc4);
          f4_st(a.dkv + row * a.lddkv + D + c, dv4);
          sv4.x += dv4.x; sv4.y += dv4.y; sv4.z += dv4.z; sv4.w += dv4.w;
          if ((cl & (lph - 1)) == 0) { pl[wv][kk[u]][h] = P; dpl[wv][kk[u]][h] = dP; }
        }
      }
    }
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) th += __shfl_xor(th, o, 64);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int k0 = 0; k0 < Sv; k0 += KPS * W1_U) {
      int kk[W1_U], pp[W1_U]; bool on[W1_U]; float4 k4[W1_U];
#pragma unroll
      for (int u = 0; u < W1_U; ++u) {
        kk[u] = k0 + u * KPS + sub;
        on[u] = kk[u] < Sv;
        pp[u] = sp[wv][on[u] ? kk[u] : 0];
        k4[u] = f4_ld(a.kp + ((size_t)b * S + pp[u]) * D + c);
      }
#pragma unroll
      for (int u = 0; u < W1_U; ++u) PIN4(k4[u]);
#pragma unroll
      for (int u = 0; u < W1_U; ++u) {
        const float g = on[u] ? pl[wv][kk[u]][h] * (dpl[wv][kk[u]][h] - th) : 0.f;        // softmax backward
        f4_fma(dq4, g, k4[u]);
        if (on[u]) {
          const float4 dk4 = f4_scale(g, q4);
          f4_st(a.dkv + ((size_t)b * S + pp[u]) * a.lddkv + c, dk4);
          sk4.x += dk4.x; sk4.y += dk4.y; sk4.z += dk4.z; sk4.w += dk4.w;
        }
      }
    }
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) { dq4 = f4_xor_add(dq4, o); sk4 = f4_xor_add(sk4, o); sv4 = f4_xor_add(sv4, o); }
    if (!pads_unread) {                                      // masked positions: exact zeros (dense consumers read them)
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int s0 = 0; s0 < S; s0 += KPS) {
        const int s = s0 + sub;
        if (s < S && !((vm >> s) & 1ull)) {
          float* o = a.dkv + ((size_t)b * S + s) * a.lddkv;
          f4_st(o + c, z); f4_st(o + D + c, z);
        }
      }
    }
    dq4 = f4_scale(a.qscale, dq4);
    if (sub == 0) {
      f4_st(a.dq + (size_t)b * a.lddq + c, dq4);
      f4_st(&dqs[wv][c], dq4);
      if (a.bias_part) {                                     // parked: folded by the step's last launch
        float* bp = a.bias_part + (size_t)b * 3 * D + c;
        f4_st(bp, dq4); f4_st(bp + D, sk4); f4_st(bp + 2 * D, sv4);
      } else {
        const float dqv[4] = {dq4.x, dq4.y, dq4.z, dq4.w}, skv[4] = {sk4.x, sk4.y, sk4.z, sk4.w}, svv[4] = {sv4.x, sv4.y, sv4.z, sv4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          atomicAdd(&a.dbq[c + e], dqv[e]); atomicAdd(&a.dbk[c + e], skv[e]); atomicAdd(&a.dbv[c + e], svv[e]);
        }
      }
    }
  } else if (sub == 0) {
    f4_st(&dqs[wv][c], make_float4(0.f, 0.f, 0.f, 0.f));
  }
  if (!a.wq) return;
  // d x[query row] += dq . Wq (+ the fan-in residual row): thread (part, i) owns output column i and D/NP of the D features
  __syncthreads();
  constexpr int NP = 256 / D, KP = D / NP;
  const int i = tid % D, pt = tid / D;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const float* wcol = a.wq + (size_t)(pt * KP) * D + i;
#pragma unroll 16
  for (int k = 0; k < KP; ++k) {
    const float wv2 = wcol[(size_t)k * D];
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[s] = fmaf(dqs[s][pt * KP + k], wv2, acc[s]);
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) part[pt][s][i] = acc[s];
  __syncthreads();
  for (int e = tid; e < 4 * D; e += 256) {
    const int s = e / D, col = e - s * D, bb = (int)blockIdx.x * 4 + s;
    if (bb >= a.n_in) continue;
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < NP; ++q) v += part[q][s][col];
    if (a.fanin_src) v += a.fanin_src[(size_t)bb * D + col];
    a.dxq_part[(size_t)bb * D + col] = v;
  }
}

// forward of the same form: pass 1 scores (K rows), pass 2 exp / dropout / context (V rows), then the softmax weights of
// all S positions (0 at the masked ones) for the backward
template <int LPR>
__global__ __launch_bounds__(256) void attn_fwd_w1_kernel(const AttnArgs a) {
  constexpr int D = 4 * LPR, KPS = 64 / LPR, W1_U = 16 / KPS;       // 16 keys per round (see PIN4 above)
  __shared__ int sp[4][64];
  __shared__ float sc[4][64][W1_MAXH];
  __shared__ float ls[4][W1_MAXH], mxl[4][W1_MAXH];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = (int)blockIdx.x * 4 + wv;
  if (b >= a.n_in) return;
  const int S = a.S, HF = a.H, lph = a.dh >> 2;
  const int sub = lane / LPR, cl = lane % LPR, c = 4 * cl, h = cl / lph;
  const unsigned long long vm = w1_valid(a, b, lane, sp[wv]);
  const int Sv = __popcll(vm);
  const float4 q4 = f4_ld(a.qp + (size_t)b * D + c);
  DropSpec drop = a.drop;
  drop.step = drop_step(a.drop); drop.step_ptr = nullptr;
  const uint32_t drow = (uint32_t)(b * HF + h);
  float mx = -INFINITY;
  for (int k0 = 0; k0 < Sv; k0 += KPS * W1_U) {
    int kk[W1_U]; bool on[W1_U]; float4 k4[W1_U];
#pragma unroll
    for (int u = 0; u < W1_U; ++u) {
      kk[u] = k0 + u * KPS + sub;
      on[u] = kk[u] < Sv;
      k4[u] = f4_ld(a.kp + ((size_t)b * S + sp[wv][on[u] ? kk[u] : 0]) * D + c);
    }
#pragma unroll
    for (int u = 0; u < W1_U; ++u) {
      float dot = f4_dot(q4, k4[u]);
      for (int o = 1; o < lph; o <<= 1) dot += __shfl_xor(dot, o, 64);
      if (on[u]) {
        mx = fmaxf(mx, dot);
        if ((cl & (lph - 1)) == 0) sc[wv][kk[u]][h] = dot;
      }
    }
  }
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float lsum = 0.f;
  float4 ctx4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k0 = 0; k0 < Sv; k0 += KPS * W1_U) {
    int kk[W1_U], pp[W1_U]; bool on[W1_U]; float4 v4[W1_U];
#pragma unroll
    for (int u = 0; u < W1_U; ++u) {
      kk[u] = k0 + u * KPS + sub;
      on[u] = kk[u] < Sv;
      pp[u] = sp[wv][on[u] ? kk[u] : 0];
      v4[u] = f4_ld(a.vp + ((size_t)b * S + pp[u]) * D + c);
    }
#pragma unroll
    for (int u = 0; u < W1_U; ++u) {
      if (!on[u]) continue;
      const float e = expf(sc[wv][kk[u]][h] - mx);
      lsum += e;
      const float m = drop.thr ? drop_mult(drop, drow, (uint32_t)pp[u]) : 1.f;
      f4_fma(ctx4, e * m, v4[u]);
    }
  }
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) { lsum += __shfl_xor(lsum, o, 64); ctx4 = f4_xor_add(ctx4, o); }
  const float inv = 1.f / lsum;
  if (sub == 0) {
    f4_st(a.ctx + (size_t)b * D + c, f4_scale(inv, ctx4));
    if ((cl & (lph - 1)) == 0) { ls[wv][h] = inv; mxl[wv][h] = mx; }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane < S) {                                            // lane = position: its softmax weights, 0 when masked
    const bool v = (vm >> lane) & 1ull;
    const int k = __popcll(vm & ((1ull << lane) - 1ull));
    for (int hh = 0; hh < HF; ++hh)
      a.attn[((size_t)b * HF + hh) * S + lane] = v ? expf(sc[wv][k][hh] - mxl[wv][hh]) * ls[wv][hh] : 0.f;
  }
}
int launch_attn_fwd_w1(const AttnArgs& a, hipStream_t st) {
  const dim3 grid(ps_cdiv(a.n_in, 4));
  if (a.d == 128) hipLaunchKernelGGL(attn_fwd_w1_kernel<32>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(attn_fwd_w1_kernel<16>, grid, dim3(256), 0, st, a);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

// ================================================================== one wave per (sequence, four heads), replicas inside
// The dropout replicas of a sequence (fan = K + 1 at C2) share its K / V rows and softmax weights; only the dropout mask
// of the weights and the context / d context rows differ.  A wave owns four heads (one Philox counter row: the group's
// keep bits of a replica are FOUR ballots, lane = key) and keeps its keys' V rows, weights and d V / d P sums in
// registers across the replica loop — no LDS staging, no barriers (the workgroup form: 14.5 / 29.6 us at C2 for ~0.1
// GFLOP).  The forward leaves the keep bits in `amask` ([n_in*fan][H] words, bit = key rank) for the backward.
// Lanes: LPR = dh lanes cover the group's 4*dh columns of a row (float4 each), 64/LPR keys per step, up to MAXK steps.
template <int DH, int MAXK>
__global__ __launch_bounds__(256) void attn_fwd_wf_kernel(const AttnArgs a, uint32_t* amask, int nchunk) {
  constexpr int LPR = DH, KPS = 64 / LPR, HC = 4 * DH, LPH = DH / 4;
  __shared__ int sp[4][64];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int NHG = a.H >> 2, D = a.d, S = a.S, HF = a.H;
  // wave -> (sequence, head group, replica chunk): the replicas are independent in the forward, so `nchunk` waves share a
  // (sequence, head group) — each repeats the short softmax, each takes fan / nchunk replicas (768 waves of 21 serial
  // replicas left the chip idle: 15 us)
  const int gw = (int)blockIdx.x * 4 + wv, ch = gw % nchunk, gq = gw / nchunk, b = gq / NHG, hg = gq - b * NHG;
  if (b >= a.n_in) return;
  const int jper = (a.fan + nchunk - 1) / nchunk, jbeg = ch * jper, jend = min(a.fan, jbeg + jper);
  const int sub = lane / LPR, cl = lane % LPR, c = hg * HC + 4 * cl, hl = cl / LPH, h = 4 * hg + hl;
  const unsigned long long vm = w1_valid(a, b, lane, sp[wv]);
  const int Sv = __popcll(vm);
  const float4 q4 = f4_ld(a.qp + (size_t)b * D + c);
  // scores of my keys, softmax over all keys of the head
  float4 v4[MAXK]; float P[MAXK]; int pk[MAXK];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < MAXK; ++i) {
    const int k = i * KPS + sub;
    pk[i] = sp[wv][k < Sv ? k : 0];
    const size_t row = (size_t)b * S + pk[i];
    const float4 k4 = f4_ld(a.kp + row * D + c);
    v4[i] = f4_ld(a.vp + row * D + c);
    float dot = f4_dot(q4, k4);
#pragma unroll
    for (int o = 1; o < LPH; o <<= 1) dot += __shfl_xor(dot, o, 64);
    P[i] = k < Sv ? dot : -INFINITY;
    mx = fmaxf(mx, P[i]);
  }
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float lsum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXK; ++i) { P[i] = i * KPS + sub < Sv ? expf(P[i] - mx) : 0.f; lsum += P[i]; }
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) lsum += __shfl_xor(lsum, o, 64);
  const float inv = 1.f / lsum;
#pragma unroll
  for (int i = 0; i < MAXK; ++i) {
    P[i] *= inv;
    if (ch == 0 && i * KPS + sub < Sv && (cl & (LPH - 1)) == 0) a.attn[((size_t)b * HF + h) * S + pk[i]] = P[i];
  }
  if (ch == 0 && lane < S && !((vm >> lane) & 1ull))         // masked positions: weight 0 (lane = position)
    for (int hh = 0; hh < 4; ++hh) a.attn[((size_t)b * HF + 4 * hg + hh) * S + lane] = 0.f;
  // replicas
  DropSpec drop = a.drop;
  drop.step = drop_step(a.drop); drop.step_ptr = nullptr;
  const int mykey = sp[wv][lane < Sv ? lane : 0];
  for (int j = jbeg; j < jend; ++j) {
    const size_t rrow = (size_t)b * a.fan + j;
    uint32_t keep[4] = {~0u, ~0u, ~0u, ~0u};
    if (drop.thr) {
      const Philox4 r = philox4x32_10((uint32_t)mykey, (uint32_t)(rrow * HF + 4 * hg) >> 2, drop.site, drop.step, drop.k0, drop.k1);
      const bool live = lane < Sv;
      keep[0] = (uint32_t)__ballot(live && r.x >= drop.thr); keep[1] = (uint32_t)__ballot(live && r.y >= drop.thr);
      keep[2] = (uint32_t)__ballot(live && r.z >= drop.thr); keep[3] = (uint32_t)__ballot(live && r.w >= drop.thr);
    }
    if (amask && lane < 4) amask[rrow * HF + 4 * hg + lane] = lane == 0 ? keep[0] : (lane == 1 ? keep[1] : (lane == 2 ? keep[2] : keep[3]));
    const uint32_t kh = hl == 0 ? keep[0] : (hl == 1 ? keep[1] : (hl == 2 ? keep[2] : keep[3]));
    float4 ctx4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < MAXK; ++i) {
      const int k = i * KPS + sub;
      const float m = (k < Sv && ((kh >> k) & 1u)) ? drop.scale : 0.f;
      f4_fma(ctx4, P[i] * (drop.thr ? m : (k < Sv ? 1.f : 0.f)), v4[i]);
    }
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) ctx4 = f4_xor_add(ctx4, o);
    if (sub == 0) f4_st(a.ctx + rrow * D + c, ctx4);
  }
}

// ================================================================== K / V / Q projections + replica attention, one launch
// Round 5: the C2 forward's front end was three latency-bound launches over the same 384 x 21 rows — embed
// (11 us), the K / V / Q GEMM (13 us: 0.5 GFLOP) and the replica attention above (10 us).  Here a workgroup of 8 waves owns ONE
// sequence for the last two:
//   * K / V:  D^T = W . X^T as in the fused per-replica kernels (x3frag.h): the sequence's <= 32 positions are the MFMA's N
//     dimension (padded positions are zero rows), wave w streams the 8 fragments of feature block w (0-3: linear_keys, 4-7:
//     linear_values; WSplit::fwd_kv, re-split by the embed launch in front) global -> registers at kernel start and runs 48
//     bf16 MFMAs (fp32-grade bf16x3).  The tiles go to LDS for the attention and, valid positions only, to kp / vp for the
//     backward (a lane's 16 consecutive features = one 64-byte run);
//   * Q (one row): 16 lane groups of 32 lanes dot the query row with 8 coalesced rows of linear_query each (exact fp32);
//   * attention: attn_fwd_wf_kernel's body with wave = (head group, replica chunk), K / V / q read from LDS.
// Outputs are exactly those of the three launches it replaces (kp, vp, qp, attn, amask, ctx): the backward is unchanged.
struct KvqLds {
  uint16_t Xa[3][X3_ROWS * X3_K];    // x planes [position][k]
  float Ks[X3_ROWS][132];            // K / V tiles [position][feature], +16 B per row
  float Vs[X3_ROWS][132];
  float qs[128];
  int sp[8][64];
};
#if PS_DIAG_ON      // [8 * workgroup + slot]: the 100 MHz counter all CUs share, wave 0's view
#define KVQ_STAMP(slot)                                                                                  \
  do {                                                                                                   \
    if (g.stamp && threadIdx.x == 0) {                                                                   \
      unsigned long long t_;                                                                             \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
      kvq_st[slot] = t_;                                                                                 \
    }                                                                                                    \
  } while (0)
#else
#define KVQ_STAMP(slot) do { } while (0)
#endif
template <int MAXK>
__global__ __launch_bounds__(512, 2) void kvq_attn_fwd_kernel(const KvqArgs g) {      // <= 128 registers: two workgroups per CU, every sequence resident
  constexpr int DH = 16, LPR = DH, KPS = 64 / LPR, HC = 4 * DH, LPH = DH / 4, D = 128;
  extern __shared__ float kvq_lds_raw[];
  KvqLds& L = *reinterpret_cast<KvqLds*>(kvq_lds_raw);
  const AttnArgs& a = g.at;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h = lane >> 5;
  if ((int)blockIdx.x >= a.n_in) {     // the backward-only weight streams are re-split here, under the sequences' workgroups (KvqArgs::split)
    wsplit_chunk(g.split, 1, ((int)blockIdx.x - a.n_in) * 512 + tid);
    return;
  }
  const int b = blockIdx.x, S = a.S, HF = a.H;
#if PS_DIAG_ON
  unsigned long long kvq_st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  KVQ_STAMP(0);
  DropSpec drop = a.drop;                                              // (the step word lives in device memory: read it with the rest)
  drop.step = drop_step(a.drop); drop.step_ptr = nullptr;
  // ---- everything with a global round trip, requested now
  // (1) the wave's weight fragments: feature block nb = wave & 3 of K (wave < 4) or V
  const uint16_t* stream = g.kv_stream + (size_t)(wave * 8) * 1536;
  uint4 wf[4][3];                                                      // a ring of four: steps 4..7 are requested as 0..3 retire
#pragma unroll
  for (int t = 0; t < 4; ++t) load_frag(wf[t], stream, t, lane);
  // (2) x rows -> planes: thread = (position, one 8-element chunk); positions past S are zero rows
  {
    const int row = tid >> 4, kc = tid & 15;
    const float* src = g.x + ((size_t)b * S + (row < S ? row : 0)) * D + 8 * kc;
    const float4 v0 = f4_ld(src), v1 = f4_ld(src + 4);
    const bool on = row < S;
    const float v[8] = {on ? v0.x : 0.f, on ? v0.y : 0.f, on ? v0.z : 0.f, on ? v0.w : 0.f,
                        on ? v1.x : 0.f, on ? v1.y : 0.f, on ? v1.z : 0.f, on ? v1.w : 0.f};
    put8(L.Xa, row, 8 * kc, v);
  }
  // (3) Q: lane group gq (32 lanes) owns outputs gq + 16 u; lane cq holds 4 elements of the query row and of each weight row
  const int gq = tid >> 5, cq = tid & 31;
  const float4 x0 = f4_ld(g.x + (size_t)b * S * D + 4 * cq);
  float4 wq[8];
  float bqv[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) { wq[u] = f4_ld(g.wq + (size_t)(gq + 16 * u) * D + 4 * cq); bqv[u] = g.bq[gq + 16 * u]; }
  // (4) the sequence's valid positions (every wave keeps its own copy of the list)
  const unsigned long long vm = w1_valid(a, b, lane, L.sp[wave]);
  const int Sv = __popcll(vm);
  const int nb = wave & 3, f0 = 32 * nb + 16 * h;
  // Q while the planes settle
  KVQ_STAMP(1);                                                        // valid list known (the first full wait)
  // (no global load or store under the lane test: each would be a round trip of its own behind a full wait)
  float qv[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) qv[u] = (half_sum_last(f4_dot(wq[u], x0)) + bqv[u]) * a.qscale;
  if (cq == 31) {
#pragma unroll
    for (int u = 0; u < 8; ++u) L.qs[gq + 16 * u] = qv[u];
  }
  KVQ_STAMP(2);
  __syncthreads();                                                     // planes + q in LDS
  KVQ_STAMP(3);
  if (tid < D) g.qp[(size_t)b * D + tid] = L.qs[tid];
  // ---- K / V tile of this wave: [32 features x 32 positions] over k = 128
  {
    // this lane's 16 bias values (requested under the products)
    const float* bsrc = (wave < 4 ? g.bk : g.bv) + f0;
    float4 bias4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bias4[q] = f4_ld(bsrc + 4 * q);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      uint4 bq[3];
      read_b(bq, L.Xa, l31, 16 * t + 8 * h);
      x3_mma(acc, wf[t & 3], bq);
      if (t < 4) load_frag(wf[t], stream, t + 4, lane);
      __builtin_amdgcn_sched_barrier(0);       // keep the refill where it is issued (x3frag.h, load_frag)
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { acc[4 * q] += bias4[q].x; acc[4 * q + 1] += bias4[q].y; acc[4 * q + 2] += bias4[q].z; acc[4 * q + 3] += bias4[q].w; }
    float* tile = (wave < 4 ? &L.Ks[0][0] : &L.Vs[0][0]) + l31 * 132 + f0;
#pragma unroll
    for (int q = 0; q < 4; ++q) f4_st(tile + 4 * q, make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]));
  }
  KVQ_STAMP(4);
  __syncthreads();                                                     // K / V tiles in LDS
  KVQ_STAMP(5);
  // K / V rows leave from the tiles, a whole 512-byte row per half wave (from the accumulators a lane holds 64 bytes of ONE row:
  // 64 partial-line pieces per store instruction, profiles/r05_mlp_notes.md); valid positions only: nothing reads the others
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = wave + 8 * it;                                     // (wave-uniform)
    if (row < S && ((vm >> row) & 1ull)) {
      const int c4 = 4 * (lane & 31);
      const float4 v = f4_ld(lane < 32 ? &L.Ks[row][c4] : &L.Vs[row][c4]);
      f4_st((lane < 32 ? g.kp : g.vp) + ((size_t)b * S + row) * D + c4, v);
    }
  }
  // ---- attention of position 0: wave = (head group hg, replica chunk ch)
  const int hg = wave >> 2, ch = wave & 3;
  const int jper = (a.fan + 3) >> 2, jbeg = ch * jper, jend = min(a.fan, jbeg + jper);
  const int sub = lane / LPR, cl = lane % LPR, c = hg * HC + 4 * cl, hl = cl / LPH, hd = 4 * hg + hl;
  const int* sp = L.sp[wave];
  const float4 q4 = f4_ld(&L.qs[c]);
  float4 v4[MAXK]; float P[MAXK]; int pk[MAXK];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < MAXK; ++i) {
    const int k = i * KPS + sub;
    pk[i] = sp[k < Sv ? k : 0];
    const float4 k4 = f4_ld(&L.Ks[pk[i]][c]);
    v4[i] = f4_ld(&L.Vs[pk[i]][c]);
    float dot = f4_dot(q4, k4);
#pragma unroll
    for (int o = 1; o < LPH; o <<= 1) dot += __shfl_xor(dot, o, 64);
    P[i] = k < Sv ? dot : -INFINITY;
    mx = fmaxf(mx, P[i]);
  }
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float lsum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXK; ++i) { P[i] = i * KPS + sub < Sv ? expf(P[i] - mx) : 0.f; lsum += P[i]; }
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) lsum += __shfl_xor(lsum, o, 64);
  const float inv = 1.f / lsum;
#pragma unroll
  for (int i = 0; i < MAXK; ++i) {
    P[i] *= inv;
    if (ch == 0 && i * KPS + sub < Sv && (cl & (LPH - 1)) == 0) a.attn[((size_t)b * HF + hd) * S + pk[i]] = P[i];
  }
  if (ch == 0 && lane < S && !((vm >> lane) & 1ull))         // masked positions: weight 0 (lane = position)
    for (int hh = 0; hh < 4; ++hh) a.attn[((size_t)b * HF + 4 * hg + hh) * S + lane] = 0.f;
  const int mykey = sp[lane < Sv ? lane : 0];
  KVQ_STAMP(6);
  for (int j = jbeg; j < jend; ++j) {
    const size_t rrow = (size_t)b * a.fan + j;
    uint32_t keep[4] = {~0u, ~0u, ~0u, ~0u};
    if (drop.thr) {
      const Philox4 r = philox4x32_10((uint32_t)mykey, (uint32_t)(rrow * HF + 4 * hg) >> 2, drop.site, drop.step, drop.k0, drop.k1);
      const bool live = lane < Sv;
      keep[0] = (uint32_t)__ballot(live && r.x >= drop.thr); keep[1] = (uint32_t)__ballot(live && r.y >= drop.thr);
      keep[2] = (uint32_t)__ballot(live && r.z >= drop.thr); keep[3] = (uint32_t)__ballot(live && r.w >= drop.thr);
    }
    if (lane < 4) g.amask[rrow * HF + 4 * hg + lane] = lane == 0 ? keep[0] : (lane == 1 ? keep[1] : (lane == 2 ? keep[2] : keep[3]));
    const uint32_t kh = hl == 0 ? keep[0] : (hl == 1 ? keep[1] : (hl == 2 ? keep[2] : keep[3]));
    float4 ctx4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < MAXK; ++i) {
      const int k = i * KPS + sub;
      const float m = (k < Sv && ((kh >> k) & 1u)) ? drop.scale : 0.f;
      f4_fma(ctx4, P[i] * (drop.thr ? m : (k < Sv ? 1.f : 0.f)), v4[i]);
    }
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) ctx4 = f4_xor_add(ctx4, o);
    if (sub == 0) f4_st(a.ctx + rrow * D + c, ctx4);
  }
  KVQ_STAMP(7);
#if PS_DIAG_ON
  if (g.stamp && tid == 0)
    for (int q = 0; q < 8; ++q) g.stamp[8 * (size_t)blockIdx.x + q] = kvq_st[q];
#endif
}
bool kvq_attn_fits(const AttnArgs& a) {
  static const bool on = ps_env_int("PS_KVQ_FUSED", 1) != 0;
  return on && a.Sq == 1 && a.qpos == 0 && a.H == 8 && a.d == 128 && a.dh == 16 && a.S <= 32 && a.fan >= 4 && a.fan <= 24 &&
         a.seq_div == 1 && !a.valid;
}
int launch_kvq_attn_fwd(const KvqArgs& g, hipStream_t st) {
  PS_REQUIRE(kvq_attn_fits(g.at), "fused K/V/Q + attention: unsupported shape");
  PS_REQUIRE(g.x && g.kv_stream && g.bk && g.bv && g.wq && g.bq && g.kp && g.vp && g.qp && g.amask && g.at.attn && g.at.ctx && g.at.ui,
             "fused K/V/Q + attention: null pointer");
  static bool attr24 = false, attr32 = false;
  const int riders = g.split.on ? ps_cdiv(wsplit_chunks(g.split, 1), 512) : 0;
#if PS_DIAG_ON
  KvqArgs gd = g;
  gd.stamp = ps_diag_int("PS_KVQ_STAMP", 0) ? ps_debug_stamp_ptr() : nullptr;
  const KvqArgs& gl = gd;
#else
  const KvqArgs& gl = g;
#endif
  if (g.at.S <= 24) {
    if (!attr24) { PS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kvq_attn_fwd_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(KvqLds))); attr24 = true; }
    hipLaunchKernelGGL((kvq_attn_fwd_kernel<6>), dim3(g.at.n_in + riders), dim3(512), sizeof(KvqLds), st, gl);
  } else {
    if (!attr32) { PS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kvq_attn_fwd_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(KvqLds))); attr32 = true; }
    hipLaunchKernelGGL((kvq_attn_fwd_kernel<8>), dim3(g.at.n_in + riders), dim3(512), sizeof(KvqLds), st, gl);
  }
  PS_LAUNCH_CHECK();
  return PS_OK;
}

// Backward with the replicas of a (sequence, head group) split over the FOUR waves of a workgroup (d = 128, H = 8: two
// workgroups per sequence).  Each wave repeats the short set-up and sums d V / d P over its 5-6 replicas (their d context
// rows all requested up front); the partial sums meet in LDS, wave 0 finishes (softmax backward, d K / d V rows, dq) and the
// workgroup multiplies its 64 dq values into Wq — 32 weights per thread, requested before the barriers — into one of the
// sequence's two partial rows (AttnArgs::dxq_part, like the LDS form).  Registers <= 170 keep three workgroups on a CU: all
// 768 resident at once.  (One wave per (sequence, head group): 37 us, a serial chain of 21 replicas on a mostly idle chip;
// eight waves per sequence in one workgroup: 18 us alone but two rounds of 256-register workgroups, 40 us in the step; the
// LDS workgroup form: 29 us.)
template <int DH, int MAXK, int NH>   // DH = 16: d = 128 (C2); DH = 32: d = 256 (C5 shard; its dQ.Wq is a GEMM of its own: no tail)
__global__ __launch_bounds__(256, 3) void attn_bwd_wf4_kernel(const AttnArgs a, const uint32_t* amask, int pads_unread, unsigned long long* stamp) {
  fork_signal(a.sig, a.sigval);
#if PS_DIAG_ON      // [8 * workgroup + slot]: the 100 MHz counter all CUs share, wave 0's view (tools/attn_bwd_wg_times.py)
  unsigned long long abw_st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define ABW_STAMP(slot)                                                                                  \
  do {                                                                                                   \
    if (stamp && threadIdx.x == 0) {                                                                     \
      unsigned long long t_;                                                                             \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
      abw_st[slot] = t_;                                                                                 \
    }                                                                                                    \
  } while (0)
#else
#define ABW_STAMP(slot) do { } while (0)
#endif
  ABW_STAMP(0);
  // keys of a lane group: i = hf * MAXK + ii, hf < NH (the key range is walked in NH parts so that only MAXK V rows and
  // d V sums are in registers at a time: d = 256 needs 11 keys per group for 21 positions)
  constexpr int LPR = DH, KPS = 64 / LPR, HC = 4 * DH, LPH = DH / 4, D = 8 * DH, NK = NH * MAXK, NV = 5 * NK, JB = 6;
  // fused d x (DH == 16 only): the head group's dK | dV rows as bf16x3 B-operand planes [position][128 k].  Only the rows of
  // valid positions are ever written, the product reads 32: a column of the MFMA's B operand only reaches the same column of
  // its result, and the columns of other positions are dropped — so the planes have XR = MAXK * KPS rows and reads past them
  // land in what follows inside this struct.
  constexpr int XR = DH == 16 ? MAXK * KPS : 1;
  struct Lds {
    int sp[4][64];
    uint16_t Xb[3][XR * X3_K];
    float red[4][NV][64];                                     // [wave][value][lane]
    float dqs[HC];
    float part[128];
    float part0[128];
  };
  static_assert(DH != 16 || sizeof(Lds) >= sizeof(int) * 256 + 3 * 32 * X3_K * 2, "planes: over-reads must stay inside the struct");
  __shared__ Lds Ls;
  auto& sp = Ls.sp; auto& red = Ls.red; auto& dqs = Ls.dqs; auto& part = Ls.part;
  const int tid = threadIdx.x, lane = tid & 63, ch = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = a.S, HF = a.H;
  const int b = (int)blockIdx.x >> 1, hg = (int)blockIdx.x & 1;
  const int sub = lane / LPR, cl = lane % LPR, c = hg * HC + 4 * cl, hl = cl / LPH, h = 4 * hg + hl;
  const bool fold_q = a.wq != nullptr && DH == 16;
  const bool fuse_dx = DH == 16 && fold_q && a.kvb_stream != nullptr;   // (kernel-uniform)
  const unsigned long long vm = w1_valid(a, b, lane, sp[ch]);
  const int Sv = __popcll(vm);
  const int jper = (a.fan + 3) >> 2, jbeg = ch * jper, jend = min(a.fan, jbeg + jper), nj = max(0, jend - jbeg);   // nj <= JB (fits)
  const bool masked = a.drop.thr != 0u;
  uint32_t mw = ~0u;                                          // lane l: keep word of (replica jbeg + l/4, head l&3)
  if (masked && lane < nj * 4) mw = amask[((size_t)b * a.fan + jbeg + (lane >> 2)) * HF + 4 * hg + (lane & 3)];
  const float scale = masked ? a.drop.scale : 1.f;
  float4 dc[JB];                                              // this wave's d context rows, all requested up front
#pragma unroll
  for (int u = 0; u < JB; ++u) dc[u] = f4_ld(a.dctx + ((size_t)b * a.fan + (u < nj ? jbeg + u : 0)) * D + c);
  float P[NK];
#pragma unroll
  for (int hf = 0; hf < NH; ++hf) {
    float4 v4[MAXK], dv4[MAXK]; float dP[MAXK];
#pragma unroll
    for (int ii = 0; ii < MAXK; ++ii) {
      const int k = (hf * MAXK + ii) * KPS + sub;
      const int p = sp[ch][k < Sv ? k : 0];
      v4[ii] = f4_ld(a.vp + ((size_t)b * S + p) * D + c);
      P[hf * MAXK + ii] = k < Sv ? a.attn[((size_t)b * HF + h) * S + p] : 0.f;
      dv4[ii] = make_float4(0.f, 0.f, 0.f, 0.f); dP[ii] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < JB; ++u) {
      if (u >= nj) break;
      const uint32_t kh = __shfl(mw, u * 4 + hl, 64);
#pragma unroll
      for (int ii = 0; ii < MAXK; ++ii) {
        const int k = (hf * MAXK + ii) * KPS + sub;
        const float m = ((kh >> k) & 1u) ? scale : 0.f;
        float dot = f4_dot(dc[u], v4[ii]);
#pragma unroll
        for (int o = 1; o < LPH; o <<= 1) dot += __shfl_xor(dot, o, 64);
        dP[ii] = fmaf(m, dot, dP[ii]);
        f4_fma(dv4[ii], P[hf * MAXK + ii] * m, dc[u]);
      }
    }
#pragma unroll
    for (int ii = 0; ii < MAXK; ++ii) {
      const int i = hf * MAXK + ii;
      red[ch][5 * i + 0][lane] = dv4[ii].x; red[ch][5 * i + 1][lane] = dv4[ii].y; red[ch][5 * i + 2][lane] = dv4[ii].z;
      red[ch][5 * i + 3][lane] = dv4[ii].w; red[ch][5 * i + 4][lane] = dP[ii];
    }
  }
  ABW_STAMP(1);
  // the tail's fan-in rows (this group's 64 columns of the replicas' d y1 sum): requested now (the d context rows are consumed), all
  // at once, and in flight ACROSS the two barriers below (raw s_barrier behind an LDS wait: `__syncthreads` would drain them) — as a
  // loop with a runtime trip count in the tail each of its ~11 loads was a round trip of its own (3.4 of the workgroup's 14.8 us,
  // tools/attn_bwd_wg_times.py).  Same addition order as before: bitwise the same sums.
  constexpr int FIN = 12;                                    // fan <= 24: at most 12 replicas per half
  float fin[FIN];
  if (fold_q) {                                              // (kernel-uniform)
    const int i = tid & 127, half = tid >> 7;
    const bool mine = a.fanin_src && (i >> 6) == hg;
    const float* src = a.fanin_src ? a.fanin_src : a.dctx;   // (any mapped address: the value is dropped)
#pragma unroll
    for (int u = 0; u < FIN; ++u) {
      const int j = half + 2 * u;
      fin[u] = src[((size_t)b * a.fan + (j < a.fan ? j : 0)) * D + i];
    }
    (void)mine;                                              // (masked where they are used: see the tail)
  }
  // fused d x: wave ch owns the 32 input features 32 ch ..; its first four weight fragments (of eight) are requested now
  uint4 wfr[4][3];
  const uint16_t* kvb = nullptr;
  if constexpr (DH == 16) {
    if (fuse_dx) {
      kvb = a.kvb_stream + (size_t)((hg * 4 + ch) * 8) * 1536;
#pragma unroll
      for (int t = 0; t < 4; ++t) load_frag(wfr[t], kvb, t, lane);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // (what crosses it is in LDS)
  ABW_STAMP(2);
  if (ch == 0) {
    const float4 q4 = f4_ld(a.qp + (size_t)b * D + c);
    float dPs[NK];
    float th = 0.f;
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      dPs[i] = (red[0][5 * i + 4][lane] + red[1][5 * i + 4][lane]) + (red[2][5 * i + 4][lane] + red[3][5 * i + 4][lane]);
      th = fmaf(P[i], dPs[i], th);
    }
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) th += __shfl_xor(th, o, 64);
    float4 dq4 = make_float4(0.f, 0.f, 0.f, 0.f), sk4 = dq4, sv4 = dq4;
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int k = i * KPS + sub;
      const size_t row = (size_t)b * S + sp[0][k < Sv ? k : 0];
      const float4 k4 = f4_ld(a.kp + row * D + c);
      const float g = P[i] * (dPs[i] - th);                    // softmax backward (0 for a dead key: P = 0)
      f4_fma(dq4, g, k4);
      if (k < Sv) {
        float4 dvs;
        dvs.x = (red[0][5 * i + 0][lane] + red[1][5 * i + 0][lane]) + (red[2][5 * i + 0][lane] + red[3][5 * i + 0][lane]);
        dvs.y = (red[0][5 * i + 1][lane] + red[1][5 * i + 1][lane]) + (red[2][5 * i + 1][lane] + red[3][5 * i + 1][lane]);
        dvs.z = (red[0][5 * i + 2][lane] + red[1][5 * i + 2][lane]) + (red[2][5 * i + 2][lane] + red[3][5 * i + 2][lane]);
        dvs.w = (red[0][5 * i + 3][lane] + red[1][5 * i + 3][lane]) + (red[2][5 * i + 3][lane] + red[3][5 * i + 3][lane]);
        const float4 dk4 = f4_scale(g, q4);
        f4_st(a.dkv + row * a.lddkv + c, dk4);
        f4_st(a.dkv + row * a.lddkv + D + c, dvs);
        if constexpr (DH == 16) {
          if (fuse_dx) {     // k index inside the group's product: its 64 dK columns, then its 64 dV columns
            const int pos = sp[0][k];
            put4<XR>(Ls.Xb, pos, 4 * cl, dk4);
            put4<XR>(Ls.Xb, pos, 64 + 4 * cl, dvs);
          }
        }
        sk4.x += dk4.x; sk4.y += dk4.y; sk4.z += dk4.z; sk4.w += dk4.w;
        sv4.x += dvs.x; sv4.y += dvs.y; sv4.z += dvs.z; sv4.w += dvs.w;
      }
    }
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) { dq4 = f4_xor_add(dq4, o); sk4 = f4_xor_add(sk4, o); sv4 = f4_xor_add(sv4, o); }
    dq4 = f4_scale(a.qscale, dq4);
    if (sub == 0) {
      f4_st(a.dq + (size_t)b * a.lddq + c, dq4);
      f4_st(&dqs[4 * cl], dq4);
      if (a.bias_part) {                                     // parked: folded by the step's last launch
        float* bp = a.bias_part + (size_t)b * 3 * D + c;
        f4_st(bp, dq4); f4_st(bp + D, sk4); f4_st(bp + 2 * D, sv4);
      } else {
        const float dqv[4] = {dq4.x, dq4.y, dq4.z, dq4.w}, skv[4] = {sk4.x, sk4.y, sk4.z, sk4.w}, svv[4] = {sv4.x, sv4.y, sv4.z, sv4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          atomicAdd(&a.dbq[c + e], dqv[e]); atomicAdd(&a.dbk[c + e], skv[e]); atomicAdd(&a.dbv[c + e], svv[e]);
        }
      }
    }
  } else if (ch == 1 && !pads_unread) {                      // masked positions: exact zeros (dense consumers read them)
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = 0; s0 < S; s0 += KPS) {
      const int s = s0 + sub;
      if (s < S && !((vm >> s) & 1ull)) {
        float* o = a.dkv + ((size_t)b * S + s) * a.lddkv;
        f4_st(o + c, z); f4_st(o + D + c, z);
      }
    }
  }
  ABW_STAMP(3);
  if (!fold_q) return;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // (dK | dV planes and d q: LDS)
  ABW_STAMP(4);
  // the tail's weights: thread (half, i) owns output column i and 32 of this group's 64 query features; requested behind the d x
  // product's last fragment refill (so that no product step waits for them), used in the tail
  float wq[32];
  if (!fuse_dx) {                                            // (no product here: requested now)
    const int i = tid & 127, half = tid >> 7;
#pragma unroll
    for (int k = 0; k < 32; ++k) wq[k] = a.wq[(size_t)(hg * HC + half * 32 + k) * D + i];
  }
  if constexpr (DH == 16) {
    if (fuse_dx) {
      // d x^T block = [Wk^T | Wv^T](32 features x 128 k) . (dK | dV)^T (128 k x positions): 48 bf16 MFMAs per wave
      const int l31 = lane & 31, hh = lane >> 5;
      f32x16 accx;
#pragma unroll
      for (int r = 0; r < 16; ++r) accx[r] = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        uint4 bq[3];
        read_b_rows<XR>(bq, Ls.Xb, l31, 16 * t + 8 * hh);
        x3_mma(accx, wfr[t & 3], bq);
        if (t < 4) load_frag(wfr[t], kvb, t + 4, lane);
        __builtin_amdgcn_sched_barrier(0);
        if (t == 3) {
          const int i = tid & 127, half = tid >> 7;
#pragma unroll
          for (int k = 0; k < 32; ++k) wq[k] = a.wq[(size_t)(hg * HC + half * 32 + k) * D + i];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // lane (position l31, half hh) holds input features 32 ch + 16 hh + r: 64 bytes of ONE position's partial row — stored from
      // there an instruction is 64 partial-line pieces (profiles/r05_mlp_notes.md).  Through the wave's own slice of `red` (dead
      // since the barrier above; LDS runs a wave's instructions in order) an instruction covers 8 positions x one 128-byte line.
      float* tile = &Ls.red[ch][0][0];
      static_assert(NV * 64 >= 32 * 36, "dX tile: the wave's slice of red holds 32 rows of 36 floats");
      {
        float* p = tile + l31 * 36 + 16 * hh;
#pragma unroll
        for (int q = 0; q < 4; ++q) f4_st(p + 4 * q, make_float4(accx[4 * q], accx[4 * q + 1], accx[4 * q + 2], accx[4 * q + 3]));
      }
      const int rr = lane >> 3, cc = 4 * (lane & 7);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + rr;
        const float4 t = f4_ld(tile + row * 36 + cc);
        if (row == 0) f4_st(&Ls.part0[32 * ch + cc], t);                         // the query position's row leaves with the tail
        else if (row < S && ((vm >> row) & 1ull)) f4_st(a.dxp[hg] + ((size_t)b * S + row) * D + 32 * ch + cc, t);
      }
    }
  }
  ABW_STAMP(5);
  {
    const int i = tid & 127, half = tid >> 7;
    float acc = 0.f;
    {   // this group's 64 columns of the replicas' fan-in sum.  Masked by a 0 / 1 WEIGHT in an fma, not by a test: under a test of
        // its only use hipcc sinks each load beneath it — a branch, a load and a full wait per row.  fma(v, 1, acc) is
        // acc + v rounded once, fma(v, 0, acc) is acc: bitwise the sums of the guarded loop.
      const bool mine = a.fanin_src && (i >> 6) == hg;
#pragma unroll
      for (int u = 0; u < FIN; ++u) acc = __builtin_fmaf(fin[u], mine && half + 2 * u < a.fan ? 1.f : 0.f, acc);
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) acc = fmaf(dqs[half * 32 + k], wq[k], acc);
    if (half == 1) part[i] = acc;
    __syncthreads();
    if (half == 0) {
      if (fuse_dx) a.dxp[hg][(size_t)b * S * D + i] = (acc + part[i]) + Ls.part0[i];
      else a.dxq_part[((size_t)hg * a.n_in + b) * D + i] = acc + part[i];
    }
  }
  ABW_STAMP(6);
#if PS_DIAG_ON
  if (stamp && tid == 0)
    for (int q = 0; q < 8; ++q) stamp[8 * (size_t)blockIdx.x + q] = abw_st[q];
#endif
}
// ---------------------------------------------------------------------------------------------------------------------------------
// The same backward with the KEYS split across the workgroup's four waves instead of the replicas (round 5, d = 256 / DH = 32, the
// C5 shard).  attn_bwd_wf4_kernel gives every wave a quarter of the replicas and ALL keys: the waves' d V / d P partials (5 values x 12
// keys x 64 lanes each) then meet in 61 KB of LDS and ONE wave finishes the softmax backward.  At d = 256 that footprint lets two
// workgroups onto a CU alone and none beside the two 64 KB weight-gradient workgroups of the side stream: 47 us stand-alone, 140-150 in
// the C5 step, on its dependent chain (profiles/r05_c5_step_timeline.txt).  Here wave ch owns keys (4 ii + ch) KPS + sub, ii < NKW, for
// ALL replicas: its d V / d P sums are complete in registers, the softmax backward and the d K / d V rows of its keys are its own, and
// what crosses waves is one float per lane (sum of P dP of the lane's head) and three float4 (d q, column sums of d K and d V): 4 KB.
// Every d context row is read by all four waves (L1 / L2 hits); rows past the fan re-read replica 0 with an all-zero keep word, so
// no load sits under a test.  Sums run over replicas 0 .. fan-1 in order, then over waves 0 .. 3: deterministic, not the bits of the
// replica-split form (which adds its four waves' partial sums pairwise).
// sum over the 8 lanes of a head (DH = 32: 8 lanes x float4), returned to every one of them: three DPP adds (quad_perm xor 1, xor 2,
// then row_half_mirror — within a quad the values are already equal, so the mirror is the xor-4 exchange) instead of three
// ds_bpermute round trips per dot product (216 per wave here)
__device__ __forceinline__ float head8_sum(float v) {
#define PS_DPP_ADD8(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
  PS_DPP_ADD8(0xB1); PS_DPP_ADD8(0x4E); PS_DPP_ADD8(0x141);
#undef PS_DPP_ADD8
  return v;
}
template <int DH, int NKW>
__global__ __launch_bounds__(256, 3) void attn_bwd_wk_kernel(const AttnArgs a, const uint32_t* amask, int pads_unread) {
  fork_signal(a.sig, a.sigval);
  constexpr int LPR = DH, KPS = 64 / LPR, HC = 4 * DH, LPH = DH / 4, D = 8 * DH, JMAX = 24;
  static_assert(4 * NKW * KPS >= JMAX, "key slots cover 24 positions");
  __shared__ int sp[4][64];
  __shared__ float ths[4][LPR];
  __shared__ float4 sums[4][3][LPR];
  const int tid = threadIdx.x, lane = tid & 63, ch = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = a.S, HF = a.H;
  const int b = (int)blockIdx.x >> 1, hg = (int)blockIdx.x & 1;
  const int sub = lane / LPR, cl = lane % LPR, c = hg * HC + 4 * cl, hl = cl / LPH, h = 4 * hg + hl;
  // ---- requests that do not need the valid list: the first batch of the replicas' d context chunks (four rows; the next batch is
  // requested when one starts to be consumed, in a ROLLED loop: unrolled, hipcc hoists every batch to the top and the kernel needs
  // 240 registers — at <= 152 a workgroup fits beside two of the side stream's 180-register weight-gradient workgroups on every SIMD),
  // the keep words, the query row
  constexpr int JB = 4;
  float4 cur[JB];
  auto dc_load = [&](float4 (&dst)[JB], const int j0) {
#pragma unroll
    for (int u = 0; u < JB; ++u) dst[u] = f4_ld(a.dctx + ((size_t)b * a.fan + (j0 + u < a.fan ? j0 + u : 0)) * D + c);
  };
  dc_load(cur, 0);
  const bool masked = a.drop.thr != 0u;
  // lane l: keep word of (replica l / 4, head l & 3); a second register for replicas 16 .. 23
  uint32_t mw0, mw1;
  {
    const int j0 = lane >> 2, j1 = 16 + (lane >> 2);
    mw0 = amask[((size_t)b * a.fan + (j0 < a.fan ? j0 : 0)) * HF + 4 * hg + (lane & 3)];
    mw1 = amask[((size_t)b * a.fan + (j1 < a.fan ? j1 : 0)) * HF + 4 * hg + (lane & 3)];
    if (!masked) { mw0 = ~0u; mw1 = ~0u; }
  }
  const float4 q4 = f4_ld(a.qp + (size_t)b * D + c);
  const float scale = masked ? a.drop.scale : 1.f;
  const unsigned long long vm = w1_valid(a, b, lane, sp[ch]);
  const int Sv = __popcll(vm);
  // ---- this wave's keys
  int kk[NKW], pk[NKW]; bool on[NKW];
  float4 v4[NKW], dv4[NKW]; float P[NKW], dP[NKW];
#pragma unroll
  for (int ii = 0; ii < NKW; ++ii) {
    kk[ii] = (ii * 4 + ch) * KPS + sub;
    on[ii] = kk[ii] < Sv;
    const int p = sp[ch][on[ii] ? kk[ii] : 0];
    pk[ii] = p;
    v4[ii] = f4_ld(a.vp + ((size_t)b * S + p) * D + c);
    const float pr = a.attn[((size_t)b * HF + h) * S + p];
    P[ii] = on[ii] ? pr : 0.f;
    dv4[ii] = make_float4(0.f, 0.f, 0.f, 0.f); dP[ii] = 0.f;
  }
  // ---- all replicas (those past the fan: keep word 0), four at a time, two batches per trip of the rolled loop (ping-pong: a batch
  // is requested one batch of arithmetic before it is used, and nothing is copied)
  auto dc_batch = [&](const float4 (&dc)[JB], const int j0) {
#pragma unroll
    for (int u = 0; u < JB; ++u) {
      const int j = j0 + u;
      uint32_t kh = __shfl(j < 16 ? mw0 : mw1, (j & 15) * 4 + hl, 64);
      kh = j < a.fan ? kh : 0u;
#pragma unroll
      for (int ii = 0; ii < NKW; ++ii) {
        const float m = ((kh >> kk[ii]) & 1u) ? scale : 0.f;
        static_assert(LPH == 8, "head8_sum");
        const float dot = head8_sum(f4_dot(dc[u], v4[ii]));
        dP[ii] = fmaf(m, dot, dP[ii]);
        f4_fma(dv4[ii], P[ii] * m, dc[u]);
      }
    }
  };
  static_assert(JMAX % (2 * JB) == 0, "two batches per trip");
#pragma unroll 1
  for (int j0 = 0; j0 < JMAX; j0 += 2 * JB) {
    float4 oth[JB];
    dc_load(oth, j0 + JB);
    __builtin_amdgcn_sched_barrier(0);
    dc_batch(cur, j0);
    __builtin_amdgcn_sched_barrier(0);
    dc_load(cur, j0 + 2 * JB < JMAX ? j0 + 2 * JB : 0);      // (the last trip re-reads batch 0: no load under a test)
    __builtin_amdgcn_sched_barrier(0);
    dc_batch(oth, j0 + JB);
  }
  // the K rows of this wave's keys (their round trip runs under the exchange below)
  float4 k4[NKW];
#pragma unroll
  for (int ii = 0; ii < NKW; ++ii) k4[ii] = f4_ld(a.kp + ((size_t)b * S + pk[ii]) * D + c);
  // ---- softmax backward: th = sum over ALL keys of P dP (per head): this wave's keys, its lane groups, then the four waves
  float thp = 0.f;
#pragma unroll
  for (int ii = 0; ii < NKW; ++ii) thp = fmaf(P[ii], dP[ii], thp);
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) thp += __shfl_xor(thp, o, 64);
  if (sub == 0) ths[ch][cl] = thp;
  __syncthreads();
  const float th = (ths[0][cl] + ths[1][cl]) + (ths[2][cl] + ths[3][cl]);
  float4 dq4 = make_float4(0.f, 0.f, 0.f, 0.f), sk4 = dq4, sv4 = dq4;
#pragma unroll
  for (int ii = 0; ii < NKW; ++ii) {
    const float g = P[ii] * (dP[ii] - th);                   // (0 for a dead key: P = 0)
    f4_fma(dq4, g, k4[ii]);
    const float4 dk4 = f4_scale(g, q4);
    if (on[ii]) {
      float* o = a.dkv + ((size_t)b * S + pk[ii]) * a.lddkv;
      f4_st(o + c, dk4);
      f4_st(o + D + c, dv4[ii]);
      sk4.x += dk4.x; sk4.y += dk4.y; sk4.z += dk4.z; sk4.w += dk4.w;
      sv4.x += dv4[ii].x; sv4.y += dv4[ii].y; sv4.z += dv4[ii].z; sv4.w += dv4[ii].w;
    }
  }
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) { dq4 = f4_xor_add(dq4, o); sk4 = f4_xor_add(sk4, o); sv4 = f4_xor_add(sv4, o); }
  if (sub == 0) { sums[ch][0][cl] = dq4; sums[ch][1][cl] = sk4; sums[ch][2][cl] = sv4; }
  __syncthreads();
  if (ch == 0 && sub == 0) {
    float4 t[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const float4 s0 = sums[0][q][cl], s1 = sums[1][q][cl], s2 = sums[2][q][cl], s3 = sums[3][q][cl];
      t[q] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w));
    }
    dq4 = f4_scale(a.qscale, t[0]); sk4 = t[1]; sv4 = t[2];
    f4_st(a.dq + (size_t)b * a.lddq + c, dq4);
    if (a.bias_part) {                                       // parked: folded by the step's last launch
      float* bp = a.bias_part + (size_t)b * 3 * D + c;
      f4_st(bp, dq4); f4_st(bp + D, sk4); f4_st(bp + 2 * D, sv4);
    } else {
      const float dqv[4] = {dq4.x, dq4.y, dq4.z, dq4.w}, skv[4] = {sk4.x, sk4.y, sk4.z, sk4.w}, svv[4] = {sv4.x, sv4.y, sv4.z, sv4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        atomicAdd(&a.dbq[c + e], dqv[e]); atomicAdd(&a.dbk[c + e], skv[e]); atomicAdd(&a.dbv[c + e], svv[e]);
      }
    }
  } else if (ch == 1 && !pads_unread) {                      // masked positions: exact zeros (dense consumers read them)
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = 0; s0 < S; s0 += KPS) {
      const int sidx = s0 + sub;
      if (sidx < S && !((vm >> sidx) & 1ull)) {
        float* o = a.dkv + ((size_t)b * S + sidx) * a.lddkv;
        f4_st(o + c, z); f4_st(o + D + c, z);
      }
    }
  }
}

// shapes the replica form is built for (both directions): 8 heads — two four-head groups per sequence — of 16 (d = 128)
// or 32 (d = 256, <= 24 positions) columns, keys
// within the register budget and their keep bits within one word, the replicas of a chunk within the backward's row batch
static bool attn_wf4_fits(const AttnArgs& a) {
  static const bool on = ps_env_int("PS_ATTN_WF", 1) != 0;
  return on && a.Sq == 1 && a.H == 8 && ((a.d == 128 && a.dh == 16 && a.S <= 32) || (a.d == 256 && a.dh == 32 && a.S <= 24)) &&
         a.fan >= 4 && a.fan <= 24;
}
bool attn_wf_fits(const AttnArgs& a) { return attn_wf4_fits(a); }
// its backward writes TWO partial dQ.Wq rows per sequence (one per head group), like the LDS form
bool attn_bwd_wf_two_partials(const AttnArgs& a) { return attn_wf4_fits(a); }
int launch_attn_fwd_wf(const AttnArgs& a, uint32_t* amask, hipStream_t st) {
  PS_REQUIRE(attn_wf_fits(a) && amask, "attention(wf): unsupported shape");
  static const int env_ch = ps_diag_int("PS_ATTN_WF_CHUNKS", 0);
  int nch = env_ch > 0 ? env_ch : (a.fan >= 12 ? 4 : 2);
  if (nch > a.fan) nch = a.fan;
  const dim3 grid(ps_cdiv(a.n_in * (a.H / 4) * nch, 4));
  if (a.dh == 32) hipLaunchKernelGGL((attn_fwd_wf_kernel<32, 12>), grid, dim3(256), 0, st, a, amask, nch);
  else if (a.S <= 24) hipLaunchKernelGGL((attn_fwd_wf_kernel<16, 6>), grid, dim3(256), 0, st, a, amask, nch);
  else hipLaunchKernelGGL((attn_fwd_wf_kernel<16, 8>), grid, dim3(256), 0, st, a, amask, nch);
  PS_LAUNCH_CHECK();
  return PS_OK;
}
int launch_attn_bwd_wf(const AttnArgs& a, const uint32_t* amask, bool pads_unread, hipStream_t st) {
  PS_REQUIRE(attn_wf_fits(a) && amask, "attention bwd(wf): unsupported shape");
  PS_REQUIRE(!a.wq || (a.dxq_part && a.d == 128), "attention bwd(wf): folded dQ.Wq needs d == 128 and its output row buffer");
  PS_REQUIRE(!a.kvb_stream || (a.wq && a.dxp[0] && a.dxp[1] && a.d == 128), "attention bwd(wf): fused d x needs the folded dQ.Wq form and both partial buffers");
  AttnArgs b = a;
  b.sig = nullptr; b.sigval = 0;
  side_take_signal(st, &b.sig, &b.sigval);             // (every check is behind us: the launch happens)
  unsigned long long* stamp = nullptr;
#if PS_DIAG_ON
  if (ps_diag_int("PS_ABW_STAMP", 0)) stamp = ps_debug_stamp_ptr();
#endif
  static const bool wk_on = ps_env_int("PS_ATTN_WK", 1) != 0;       // d = 256: keys, not replicas, across the waves (0: the replica-split form)
  if (a.dh == 32 && wk_on && !a.wq) hipLaunchKernelGGL((attn_bwd_wk_kernel<32, 3>), dim3(a.n_in * 2), dim3(256), 0, st, b, amask, pads_unread ? 1 : 0);
  else if (a.dh == 32) hipLaunchKernelGGL((attn_bwd_wf4_kernel<32, 6, 2>), dim3(a.n_in * 2), dim3(256), 0, st, b, amask, pads_unread ? 1 : 0, stamp);
  else if (a.S <= 24) hipLaunchKernelGGL((attn_bwd_wf4_kernel<16, 6, 1>), dim3(a.n_in * 2), dim3(256), 0, st, b, amask, pads_unread ? 1 : 0, stamp);
  else hipLaunchKernelGGL((attn_bwd_wf4_kernel<16, 8, 1>), dim3(a.n_in * 2), dim3(256), 0, st, b, amask, pads_unread ? 1 : 0, stamp);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

bool attn_w1_fits(const AttnArgs& a) {
  static const bool on = ps_env_int("PS_ATTN_W1", 1) != 0;
  const int lph = a.dh / 4;
  return on && a.Sq == 1 && a.fan == 1 && a.S <= 64 && (a.d == 128 || a.d == 64) && a.H <= W1_MAXH && a.dh % 4 == 0 &&
         lph >= 1 && (lph & (lph - 1)) == 0 && a.dh * a.H == a.d;
}
int launch_attn_bwd_w1(const AttnArgs& a, bool pads_unread, hipStream_t st) {
  PS_REQUIRE(attn_w1_fits(a), "attention bwd(w1): unsupported shape");
  PS_REQUIRE(!a.wq || a.dxq_part, "attention bwd(w1): folded dQ.Wq needs its output row buffer");
  const dim3 grid(ps_cdiv(a.n_in, 4));
  AttnArgs b = a;
  b.sig = nullptr; b.sigval = 0;
  side_take_signal(st, &b.sig, &b.sigval);             // (every check is behind us: the launch happens)
  if (a.d == 128) hipLaunchKernelGGL(attn_bwd_w1_kernel<32>, grid, dim3(256), 0, st, b, pads_unread ? 1 : 0);
  else hipLaunchKernelGGL(attn_bwd_w1_kernel<16>, grid, dim3(256), 0, st, b, pads_unread ? 1 : 0);
  PS_LAUNCH_CHECK();
  return PS_OK;
}
