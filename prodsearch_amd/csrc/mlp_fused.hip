// mlp_fused.hip — the per-replica tail of the last encoder layer as ONE kernel (gfx950, d = 128).
//
// After attention, every encoder replica row m (B*R of them, R = K+1 when dropout is drawn) runs
//   y1  = dropout(ctx . Wo^T + bo) + x[b, qpos]                 (neural.py:228-231, transformer.py:56)
//   ln1 = LayerNorm_ff(y1) ; a1 = ln1 . W1^T + b1 ; h1 = dropout(gelu(a1))
//   y2  = dropout(h1 . W2^T + b2) + y1                           (neural.py:30-33)
//   enc = LayerNorm_final(y2)                                    (transformer.py:86)
// Unfused this is 3 GEMM + 2 LayerNorm launches whose 64x64 tiles each live for one short,
// latency-bound round trip.  Here a workgroup (4 waves) owns 32 replica rows for the whole chain:
// activations stay in LDS/registers, and the 0.6 MB of weights stream through a double-buffered
// LDS slab ring (64-deep slabs of 128 output columns, prefetched global->registers while the
// previous slab is multiplied), so the MFMA pipe sees 18 slabs x 32 v_mfma_f32_32x32x2_f32 per
// wave back to back instead of 5 cold starts.  W1/W2 are walked in 128-unit chunks of the hidden
// layer so h1 never exists as a whole tile: a1 chunk -> gelu/dropout -> LDS -> accumulate into y2.
// Everything the backward needs (y1, ln1, a1, h1, y2, LN statistics) is still written once.
// Measured alternatives (MI355X, 8,064 rows): 32-deep slabs 83 us (prefetch latency exposed), 64-deep 73 us (this
// file); an 8-wave variant on v_mfma_f32_16x16x4_f32 (two waves per SIMD, k-quad-interleaved LDS operands) was correct
// but no faster (76 us): 16x16 tiles need 3x the LDS operand bandwidth of 32x32 tiles for the same flops.
#include "rowwise.h"
#include <stdlib.h>

bool ps_fusion_enabled() {
  static const bool on = !(getenv("PS_NO_FUSE") && atoi(getenv("PS_NO_FUSE")) != 0);
  return on;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Launder a value through an empty asm: stops LLVM from hoisting the (many) per-row store addresses of the
// stage epilogues out of the slab loop, which would pin >100 VGPRs for the whole kernel.
__device__ inline int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

#define MD 128            // model width this kernel is specialised for
#define MBM 32            // replica rows per workgroup
#ifndef MBK
#define MBK 64            // reduction depth of one weight slab (64: slab MFMA time ~ the L2 latency of the prefetch)
#endif
#define SPP (128 / MBK)   // slabs per 128-deep product
#define WRN (MBK / 8)     // float4 registers per thread holding one prefetched slab
#define XLD 33            // k-major activation tiles: [k][row + 1]
#define WLD 129           // weight slabs: [k][col + 1]
#define YLD 129           // row-major staging tile for the LayerNorms: [row][col + 1]

struct MlpLds {
  float Xs[MD * XLD];           // A operand of Wo / W1 products: ctx, then ln1      (k-major)
  float Hs[MD * XLD];           // A operand of the W2 product: h1 chunk (k-major); aliased as Y staging
  float Ws[2][MBK * WLD];       // weight slab ring
};
static_assert(MBM * YLD <= MD * XLD, "Y staging must fit the h1 tile");

// global -> registers: slab of 128 output rows x 32 k from a [n][k] (k contiguous) matrix
__device__ inline void slab_load(const float* __restrict__ W, int ldw, int n0, int k0, float4 (&r)[WRN], int tid) {
  constexpr int KQ = MBK / 4, NPP = 256 / KQ;      // float4 columns per slab row, slab rows per pass
#pragma unroll
  for (int u = 0; u < WRN; ++u) {
    const int n = tid / KQ + NPP * u, kq = tid % KQ;
    r[u] = *reinterpret_cast<const float4*>(W + (size_t)(n0 + n) * ldw + k0 + 4 * kq);
  }
}
__device__ inline void slab_store(float* Wsb, const float4 (&r)[WRN], int tid) {
  constexpr int KQ = MBK / 4, NPP = 256 / KQ;
#pragma unroll
  for (int u = 0; u < WRN; ++u) {
    const int n = tid / KQ + NPP * u, kk = 4 * (tid % KQ);
    Wsb[(kk + 0) * WLD + n] = r[u].x;
    Wsb[(kk + 1) * WLD + n] = r[u].y;
    Wsb[(kk + 2) * WLD + n] = r[u].z;
    Wsb[(kk + 3) * WLD + n] = r[u].w;
  }
}

// LayerNorm of the 32 x 128 tile staged row-major in Y: wave w normalises rows 8w..8w+7, two columns per lane.
// Writes the normalised tile k-major into Xk (if given), row-major to `out` (global, ld = MD), stats to `stats`.
__device__ inline void tile_layernorm(const float* Y, const float* __restrict__ g, const float* __restrict__ bta,
                                      float* Xk, float* out, float* stats, int m0, int M, int wave, int lane) {
  const float g0 = g[lane], g1 = g[lane + 64], b0 = bta[lane], b1 = bta[lane + 64];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = wave * 8 + i;
    const float v0 = Y[row * YLD + lane], v1 = Y[row * YLD + lane + 64];
    const float mean = wave_sum(v0 + v1) * (1.f / MD);
    const float d0 = v0 - mean, d1 = v1 - mean;
    const float rstd = 1.f / sqrtf(wave_sum(d0 * d0 + d1 * d1) * (1.f / MD) + 1e-6f);
    const float o0 = d0 * rstd * g0 + b0, o1 = d1 * rstd * g1 + b1;
    if (Xk) { Xk[lane * XLD + row] = o0; Xk[(lane + 64) * XLD + row] = o1; }
    const int m = m0 + row;
    if (m < M) {
      out[(size_t)m * MD + lane] = o0;
      out[(size_t)m * MD + lane + 64] = o1;
      if (lane == 0) { stats[2 * (size_t)m] = mean; stats[2 * (size_t)m + 1] = rstd; }
    }
  }
}

__global__ __launch_bounds__(256, 1) void mlp_fwd_fused_kernel(const MlpFwdArgs a) {
  extern __shared__ float lds_raw[];
  MlpLds& L = *reinterpret_cast<MlpLds*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * MBM, M = a.M;
  const int col = wave * 32 + l31;                 // this lane's output column in every 128-wide product
  const int nchunk = a.F / 128;
  const int NS = SPP * (1 + 2 * nchunk);           // weight slabs: Wo, then per chunk W1c + W2c (SPP slabs each)

  auto slab_src = [&](int s, const float*& W, int& ldw, int& n0, int& k0) {
    if (s < SPP) { W = a.wo; ldw = MD; n0 = 0; k0 = MBK * s; return; }
    const int t = s - SPP, c = t / (2 * SPP), r = t % (2 * SPP);
    if (r < SPP) { W = a.w1; ldw = MD; n0 = 128 * c; k0 = MBK * r; }
    else { W = a.w2; ldw = a.F; n0 = 0; k0 = 128 * c + MBK * (r - SPP); }
  };

  // ---- prologue: first weight slab, the ctx tile (k-major), the residual rows
  // two weight slabs in flight in registers (wr0 / wr1 alternate: the loop body below is instantiated once per slot so
  // the slots stay static): one slab of MFMAs (~0.85 us) does not cover the round trip of the next slab's loads
  float4 wr0[WRN], wr1[WRN];
  {
    const float* W; int ldw, n0, k0;
    slab_src(0, W, ldw, n0, k0);
    slab_load(W, ldw, n0, k0, wr0, tid);
    slab_src(1, W, ldw, n0, k0);
    slab_load(W, ldw, n0, k0, wr1, tid);
  }
  {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f = tid + 256 * u, row = f >> 5, kq = f & 31;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m0 + row < M) v = *reinterpret_cast<const float4*>(a.ctx + (size_t)(m0 + row) * MD + 4 * kq);
      L.Xs[(4 * kq + 0) * XLD + row] = v.x;
      L.Xs[(4 * kq + 1) * XLD + row] = v.y;
      L.Xs[(4 * kq + 2) * XLD + row] = v.z;
      L.Xs[(4 * kq + 3) * XLD + row] = v.w;
    }
  }
  float y1v[16];                                    // starts as the residual x[b, qpos][col] of this lane's 16 rows
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
    y1v[r] = m < M ? a.xin[((size_t)(m / a.fan) * a.S + a.qpos) * MD + col] : 0.f;
  }
  slab_store(L.Ws[0], wr0, tid);
  __syncthreads();

  f32x16 acc, acc_o;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc_o[r] = 0.f; }
  int buf = 0;

  // one slab: refill `wfree` (its slab already sits in LDS) two slabs ahead, multiply slab s, run the stage boundary,
  // publish `wnext` (slab s + 1) to the other LDS buffer
  auto slab_step = [&](const int s, float4 (&wfree)[WRN], const float4 (&wnext)[WRN]) {
    {   // unconditional (past the end it re-reads the last slab, a cache hit that is dropped): under a branch the
        // registers become a phi and the compiler copies them — waiting for the loads — before the MFMA block
      const float* W; int ldw, n0, k0;
      slab_src(s + 2 < NS ? s + 2 : NS - 1, W, ldw, n0, k0);
      slab_load(W, ldw, n0, k0, wfree, tid);
    }
    // which product is this slab part of?
    const int t = s - SPP, r8 = t % (2 * SPP);
    const bool is_w2 = s >= SPP && r8 >= SPP;
    const int ka = s < SPP ? MBK * s : (is_w2 ? MBK * (r8 - SPP) : MBK * r8);     // k offset inside the A tile
    const float* A = is_w2 ? L.Hs : L.Xs;
#pragma unroll
    for (int half = 0; half < MBK / 32; ++half) {                               // 16 MFMAs (32 k) at a time
      const float* ab = A + (ka + 32 * half + h) * XLD + l31;
      const float* bb = L.Ws[buf] + (32 * half + h) * WLD + col;
      float av[16], bv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) { av[i] = ab[2 * i * XLD]; bv[i] = bb[2 * i * WLD]; }
      if (is_w2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc_o = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc_o, 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc, 0, 0, 0);
      }
    }

    // ---- stage boundaries
    if (s == SPP - 1) {
      // y1 = dropout(ctx.Wo^T + bo) + residual ; LayerNorm_ff -> ln1 (k-major in Xs)
      const float bias = a.bo[col];
      float* Y = L.Hs;
      const int mm0 = opaque(m0);
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int rb = mm0 + 8 * gq + 4 * h;
        Philox4 rnd = {0u, 0u, 0u, 0u};
        if (a.drop_ctx.thr) rnd = philox4x32_10((uint32_t)col, (uint32_t)rb >> 2, a.drop_ctx.site, drop_step(a.drop_ctx),
                                               a.drop_ctx.k0, a.drop_ctx.k1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 4 * gq + q;
          float v = acc[r] + bias;
          if (a.drop_ctx.thr) v *= drop_word(a.drop_ctx, q == 0 ? rnd.x : (q == 1 ? rnd.y : (q == 2 ? rnd.z : rnd.w)));
          v += y1v[r];
          y1v[r] = v;
          const int lrow = 8 * gq + 4 * h + q;
          Y[lrow * YLD + col] = v;
          if (rb + q < M) a.y1[(size_t)(rb + q) * MD + col] = v;
          acc[r] = 0.f;
        }
      }
      __syncthreads();                              // Y complete; every wave is done reading ctx from Xs
      tile_layernorm(Y, a.g1, a.be1, L.Xs, a.ln1, a.st1, opaque(m0), M, wave, lane);
      // the end-of-iteration barrier below publishes ln1 before the W1 slabs read it
    } else if (s >= SPP && r8 == SPP - 1) {
      // a1 chunk -> gelu -> dropout -> h1 chunk (k-major in Hs)
      const int c = t / (2 * SPP), f = 128 * c + col;
      const float bias = a.b1[f];
      const int mm0 = opaque(m0);
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int rb = mm0 + 8 * gq + 4 * h;
        Philox4 rnd = {0u, 0u, 0u, 0u};
        if (a.drop_ff1.thr) rnd = philox4x32_10((uint32_t)f, (uint32_t)rb >> 2, a.drop_ff1.site, drop_step(a.drop_ff1),
                                               a.drop_ff1.k0, a.drop_ff1.k1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 4 * gq + q;
          const float pre = acc[r] + bias;
          float hv = gelu_tanh_f(pre);
          if (a.drop_ff1.thr) hv *= drop_word(a.drop_ff1, q == 0 ? rnd.x : (q == 1 ? rnd.y : (q == 2 ? rnd.z : rnd.w)));
          const int lrow = 8 * gq + 4 * h + q;
          L.Hs[col * XLD + lrow] = hv;
          if (rb + q < M) {
            a.a1[(size_t)(rb + q) * a.F + f] = pre;
            a.h1[(size_t)(rb + q) * a.F + f] = hv;
          }
          acc[r] = 0.f;
        }
      }
    } else if (s == NS - 1) {
      // y2 = dropout(h1.W2^T + b2) + y1 ; final LayerNorm -> enc
      const float bias = a.b2[col];
      float* Y = L.Hs;
      __syncthreads();                              // every wave is done reading the last h1 chunk from Hs
      const int mm0 = opaque(m0);
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int rb = mm0 + 8 * gq + 4 * h;
        Philox4 rnd = {0u, 0u, 0u, 0u};
        if (a.drop_ff2.thr) rnd = philox4x32_10((uint32_t)col, (uint32_t)rb >> 2, a.drop_ff2.site, drop_step(a.drop_ff2),
                                               a.drop_ff2.k0, a.drop_ff2.k1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 4 * gq + q;
          float v = acc_o[r] + bias;
          if (a.drop_ff2.thr) v *= drop_word(a.drop_ff2, q == 0 ? rnd.x : (q == 1 ? rnd.y : (q == 2 ? rnd.z : rnd.w)));
          v += y1v[r];
          const int lrow = 8 * gq + 4 * h + q;
          Y[lrow * YLD + col] = v;
          if (rb + q < M) a.y2[(size_t)(rb + q) * MD + col] = v;
        }
      }
      __syncthreads();
      tile_layernorm(Y, a.gf, a.bef, nullptr, a.enc, a.stf, opaque(m0), M, wave, lane);
    }

    if (s + 1 < NS) slab_store(L.Ws[buf ^ 1], wnext, tid);
    __syncthreads();
    buf ^= 1;
  };
  for (int s = 0; s < NS; s += 2) {                 // NS = SPP * (1 + 2 * nchunk) is even
    slab_step(s, wr0, wr1);
    slab_step(s + 1, wr1, wr0);
  }
}

// ====================================================================== forward, wave-specialised form
// PMC profile of the kernel above (profiles/r02_mlp_counters.md): per wave 36.9k cycles of MFMA, ~33k cycles of OTHER
// vector instructions (Philox, GELU, LayerNorm, addresses) and ~33k cycles of waiting — one wave per SIMD does them one
// after the other, so the matrix pipe is busy 27 % of the time.  The matrix and vector pipes of a SIMD run concurrently
// when the instructions come from DIFFERENT waves, hence this form: 8 waves per workgroup, two per SIMD,
//   waves 0-3 (matrix waves)  only read operands from LDS and issue MFMAs — 32 per 64-deep weight slab;
//   waves 4-7 (helper waves)  stream the weight slabs (global -> registers -> LDS ring) and run the GELU/dropout
//                             epilogue of W1 chunk c out of an LDS copy of its accumulators WHILE the matrix waves
//                             multiply the next product (W1 chunk c+1 or W2 chunk c-1): the products are ordered
//                             Wo, W1c0, W1c1, W2c0, W1c2, W2c1, ..., W2c(n-1) so nobody waits for that epilogue.
// Only the two LayerNorm stages (after Wo, after the last W2 chunk) are on the critical path; all 8 waves share them,
// 4 rows each, keeping their y1 elements in registers from the first to the second.  Same k order of every
// accumulation and the same element-wise arithmetic as the kernel above: bitwise the same outputs.
#define WS_THREADS 512
struct MlpWsLds {
  float Xs[MD * XLD];           // A operand of Wo / W1: ctx, then ln1                      (k-major)
  float Hs[2][MD * XLD];        // A operand of W2: h1 chunk c in Hs[c & 1]                 (k-major)
  float Ws[2][MBK * WLD];       // weight slab ring
  float Cs[2][MBM * YLD];       // accumulator tiles handed to the epilogues (row-major): Wo / final in [0], W1 chunk c in [c & 1]
  float red[16];                // folded score: per-wave loss partials, the `last arriver` flag
};
static_assert(sizeof(MlpWsLds) <= 160 * 1024, "wave-specialised MLP: LDS budget");
static_assert(SPP == 2, "the helper waves split a chunk epilogue over the two slabs of the following product");

// STAMP: diagnostic build only (tools/dbg/ws_stamps.py) — workgroup 0's wave 0 (matrix) and wave 4 (helper) record
// s_memtime when they start a slab's work and when they reach its closing barrier
static unsigned long long* g_ws_stamp = nullptr;
extern "C" void ps_debug_set_stamp_buffer(void* p) { g_ws_stamp = (unsigned long long*)p; }
#define WS_STAMP(slot)                                                                                         \
  do {                                                                                                         \
    if (STAMP && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4)) {                                   \
      unsigned long long t_;                                                                                   \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                               \
      stamp[((wave >> 2) * 128 + (slot)) ] = t_;                                                                \
    }                                                                                                          \
  } while (0)
template <bool STAMP>
__global__ __launch_bounds__(WS_THREADS, 2) void mlp_fwd_ws_kernel(const MlpFwdArgs a, unsigned long long* stamp) {
  extern __shared__ float lds_raw[];
  MlpWsLds& L = *reinterpret_cast<MlpWsLds*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const bool is_m = wave < 4;
  const int htid = tid & 255, hw = wave & 3;          // helper-wave local thread / wave ids
  const int m0 = blockIdx.x * MBM, M = a.M;
  const int mcol = wave * 32 + l31;                   // matrix waves: this lane's output column of every 128-wide product
  const int nchunk = a.F / 128;
  const int NP = 1 + 2 * nchunk, NS = SPP * NP;

  // product p: kind 0 = Wo, 1 = W1 chunk c, 2 = W2 chunk c
  auto prod = [&](int p, int& kind, int& c) {
    if (p == 0) { kind = 0; c = 0; }
    else if (p == 1) { kind = 1; c = 0; }
    else if (p == 2 * nchunk) { kind = 2; c = nchunk - 1; }
    else if (p & 1) { kind = 2; c = (p - 3) >> 1; }
    else { kind = 1; c = p >> 1; }
  };
  auto slab_src = [&](int s, const float*& W, int& ldw, int& n0, int& k0) {
    int kind, c;
    prod(s / SPP, kind, c);
    const int r = s % SPP;
    if (kind == 0) { W = a.wo; ldw = MD; n0 = 0; k0 = MBK * r; }
    else if (kind == 1) { W = a.w1; ldw = MD; n0 = 128 * c; k0 = MBK * r; }
    else { W = a.w2; ldw = a.F; n0 = 0; k0 = 128 * c + MBK * r; }
  };

  // the Philox step of the three dropout sites, read once (graph replay keeps it in device memory: DropSpec::step_ptr)
  const uint32_t step_ctx = drop_step(a.drop_ctx), step_ff1 = drop_step(a.drop_ff1), step_ff2 = drop_step(a.drop_ff2);

  // the six small vectors of the two LayerNorm stages, requested now: at the stages themselves their L2 round trip would
  // sit on the critical path of all 8 waves
  const float pv_bo[2] = {a.bo[lane], a.bo[lane + 64]}, pv_g1[2] = {a.g1[lane], a.g1[lane + 64]},
              pv_be1[2] = {a.be1[lane], a.be1[lane + 64]}, pv_b2[2] = {a.b2[lane], a.b2[lane + 64]},
              pv_gf[2] = {a.gf[lane], a.gf[lane + 64]}, pv_bef[2] = {a.bef[lane], a.bef[lane + 64]};

  // ---- prologue
  float4 wr0[WRN], wr1[WRN];
  if (!is_m) {
    const float* W; int ldw, n0, k0;
    slab_src(0, W, ldw, n0, k0);
    slab_load(W, ldw, n0, k0, wr0, htid);
    slab_src(1, W, ldw, n0, k0);
    slab_load(W, ldw, n0, k0, wr1, htid);
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {                       // ctx tile -> Xs (k-major)
    const int f = tid + WS_THREADS * u, row = f >> 5, kq = f & 31;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m0 + row < M) v = *reinterpret_cast<const float4*>(a.ctx + (size_t)(m0 + row) * MD + 4 * kq);
    L.Xs[(4 * kq + 0) * XLD + row] = v.x;
    L.Xs[(4 * kq + 1) * XLD + row] = v.y;
    L.Xs[(4 * kq + 2) * XLD + row] = v.z;
    L.Xs[(4 * kq + 3) * XLD + row] = v.w;
  }
  // rows 4*wave .. 4*wave+3, columns lane and lane + 64: this lane's elements in both LayerNorm stages
  float y1r[4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int m = m0 + 4 * wave + q;
    const float* src = a.xin + ((size_t)((m < M ? m : 0) / a.fan) * a.S + a.qpos) * MD;
    y1r[q][0] = m < M ? src[lane] : 0.f;
    y1r[q][1] = m < M ? src[lane + 64] : 0.f;
  }
  // folded scoring (MlpFwdArgs::fold_score): the item row each of this lane's 4 enc rows will be dotted with — requested
  // now, consumed after the last LayerNorm
  float itr[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
  float ibias[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.fold_score) {
    const ScoreArgs& S = a.sc;
    const int K1 = S.K + 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = m0 + 4 * wave + q;
      if (m < M) {
        const int b = fdiv(m, S.fK1), j = m - b * K1;
        int64_t idx = j == 0 ? S.target[b] : S.neg_items[(size_t)b * S.K + j - 1];
        idx = idx < 0 ? S.P : (idx > S.P ? S.P : idx);
        const float* row = S.product_emb + (size_t)idx * MD;
        itr[q][0] = row[lane]; itr[q][1] = row[lane + 64];
        if (S.bias_product) ibias[q] = S.product_bias[idx];
      }
    }
  }
  if (!is_m) {
    slab_store(L.Ws[0], wr0, htid);
    const float* W; int ldw, n0, k0;
    slab_src(2 < NS ? 2 : NS - 1, W, ldw, n0, k0);
    slab_load(W, ldw, n0, k0, wr0, htid);
  }
  __syncthreads();

  f32x16 acc, acc_o;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc_o[r] = 0.f; }

  // one LayerNorm stage, all 8 waves: v = dropout(C + bias) + residual -> (out_pre) ; LayerNorm -> out_ln, stats, Xk
  // (the global stores are a separate step, ln_store: the last stage scores and takes its ticket first, so that the
  // signalling lane has no store of this stage to wait for)
  auto ln_stage = [&](const float* Cst, const float (&bias)[2], const DropSpec& drop, const uint32_t dstep, const float (&g)[2],
                      const float (&bta)[2], float* Xk, float (&res)[4][2], float (&o)[4][2], float (&mr)[4][2]) {
    const int rb = opaque(m0) + 4 * wave;
    const float b0 = bias[0], b1 = bias[1];
    const float g0 = g[0], g1 = g[1], e0 = bta[0], e1 = bta[1];
    Philox4 r0 = {0u, 0u, 0u, 0u}, r1 = {0u, 0u, 0u, 0u};
    if (drop.thr) {
      r0 = philox4x32_10((uint32_t)lane, (uint32_t)rb >> 2, drop.site, dstep, drop.k0, drop.k1);
      r1 = philox4x32_10((uint32_t)lane + 64u, (uint32_t)rb >> 2, drop.site, dstep, drop.k0, drop.k1);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 4 * wave + q, m = rb + q;
      float v0 = Cst[row * YLD + lane] + b0, v1 = Cst[row * YLD + lane + 64] + b1;
      if (drop.thr) {
        v0 *= drop_word(drop, q == 0 ? r0.x : (q == 1 ? r0.y : (q == 2 ? r0.z : r0.w)));
        v1 *= drop_word(drop, q == 0 ? r1.x : (q == 1 ? r1.y : (q == 2 ? r1.z : r1.w)));
      }
      v0 += res[q][0]; v1 += res[q][1];
      res[q][0] = v0; res[q][1] = v1;
      const float mean = wave_sum(v0 + v1) * (1.f / MD);
      const float d0 = v0 - mean, d1 = v1 - mean;
      const float rstd = 1.f / sqrtf(wave_sum(d0 * d0 + d1 * d1) * (1.f / MD) + 1e-6f);
      const float o0 = d0 * rstd * g0 + e0, o1 = d1 * rstd * g1 + e1;
      o[q][0] = o0; o[q][1] = o1;
      mr[q][0] = mean; mr[q][1] = rstd;
      if (Xk) { Xk[lane * XLD + row] = o0; Xk[(lane + 64) * XLD + row] = o1; }
      (void)m;
    }
  };
  auto ln_store = [&](float* out_pre, float* out_ln, float* stats, const float (&res)[4][2], const float (&o)[4][2],
                      const float (&mr)[4][2]) {
    const int rb = opaque(m0) + 4 * wave;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = rb + q;
      if (m < M) {
        out_pre[(size_t)m * MD + lane] = res[q][0]; out_pre[(size_t)m * MD + lane + 64] = res[q][1];
        out_ln[(size_t)m * MD + lane] = o[q][0]; out_ln[(size_t)m * MD + lane + 64] = o[q][1];
        if (lane == 0) { stats[2 * (size_t)m] = mr[q][0]; stats[2 * (size_t)m + 1] = mr[q][1]; }
      }
    }
  };

  // folded scoring, all 8 waves, after the final LayerNorm: o = this lane's enc elements.  score, loss term, one loss
  // partial per workgroup, handed over by agent-scope atomics alone (below); the workgroup that arrives last adds the
  // word tasks' partials (left by the embed launch, an earlier kernel) and writes the loss.
  auto score_stage = [&](const float (&o)[4][2], auto&& stores) {
    const ScoreArgs& S = a.sc;
    const int K1 = S.K + 1;
    WS_STAMP(40);
    // lane q (< 4) ends up with row q's score and computes its loss term: one softplus per wave instead of four
    float scq = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float sdot = wave_sum(o[q][0] * itr[q][0] + o[q][1] * itr[q][1]) + ibias[q];
      scq = lane == q ? sdot : scq;
    }
    const int mq = m0 + 4 * wave + (lane & 3);
    const bool rowq = lane < 4 && mq < M;
    const int bq = fdiv(rowq ? mq : 0, S.fK1), jq = (rowq ? mq : 0) - bq * K1;
    const float twq = jq == 0 ? -(S.pos_weight ? (float)S.K : 1.f) : 1.f;
    const float termq = fabsf(twq) * softplus_f(twq < 0.f ? -scq : scq);
    const float cps = wave_sum(rowq ? termq : 0.f);
    if (lane == 0) L.red[wave] = cps;
    WS_STAMP(41);
    __syncthreads();
    WS_STAMP(42);
    unsigned long long mine = 0ull, old = 0ull;
    const unsigned long long one = 1ull << 48, mask = one - 1ull;
    if (tid == 0) {
      // one returning 64-bit atomic per workgroup carries BOTH its partial (fixed point, 2^-20 units: integer adds
      // commute, so the total is exact and independent of the arrival order) and its arrival (bits 48+): nothing to
      // store, drain or re-read.  (Eight shard words + a top word, so that at most 32 + 8 workgroups meet on one address,
      // measured no faster: the workgroup that ends the kernel then pays two round trips.)
      const float p = ((L.red[0] + L.red[1]) + (L.red[2] + L.red[3])) + ((L.red[4] + L.red[5]) + (L.red[6] + L.red[7]));
      unsigned long long* tk = reinterpret_cast<unsigned long long*>(S.ticket);
      mine = (unsigned long long)(long long)__float2ll_rn(p * 1048576.f) | one;
      old = __hip_atomic_fetch_add(tk, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the stage's stores are issued under the atomic's round trip (2-2.5 us while the chip streams this kernel's writes)
    if (rowq) { S.item_scores[mq] = scq; S.item_terms[mq] = termq; }
    stores();
    if (tid == 0) {
      float last = 0.f;
      if ((old >> 48) == gridDim.x - 1u) {
        last = 1.f;
        L.red[9] = (float)((double)((old & mask) + (mine & mask)) * (1.0 / 1048576.0));
      }
      L.red[8] = last;
    }
    WS_STAMP(43);
    __syncthreads();
    WS_STAMP(44);
    if (L.red[8] == 0.f) return;
    // last arriver: add the word tasks' partials (left by the embed launch) in a fixed order => bitwise reproducible
    if (wave == 0) {
      float il = 0.f;
      for (int i = lane; i < S.word_nblk; i += 64) il += S.word_blk[i];
      il = wave_sum(il);
      if (lane == 0) {
        const float ps = L.red[9] / (float)S.B;
        il /= (float)S.B;
        S.loss3[0] = ps + il; S.loss3[1] = ps; S.loss3[2] = il;
        if (S.loss_acc) { S.loss_acc[0] += ps; S.loss_acc[1] += il; }
      }
    }
  };

  // matrix waves: accumulator tile -> Cs (row-major)
  auto dump = [&](float* Cst, f32x16& v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      Cst[((r & 3) + 8 * (r >> 2) + 4 * h) * YLD + mcol] = v[r];
      v[r] = 0.f;
    }
  };

  // ---- the two roles run their own loops (every barrier below is reached by all 8 waves: one per slab, one more in
  // front of each LayerNorm stage).  Separate loops rather than one loop with a role branch inside: registers loaded
  // under a branch become phis at the join and the compiler then waits for the prefetch right where it was issued.
  if (is_m) {
    for (int s = 0; s < NS; ++s) {
      int kind, c;
      prod(s / SPP, kind, c);
      const int r = s % SPP;
      const float* A = kind == 2 ? L.Hs[c & 1] : L.Xs;
      const float* Wb = L.Ws[s & 1];
      WS_STAMP(2 * s);
#pragma unroll
      for (int half = 0; half < MBK / 32; ++half) {                               // 16 MFMAs (32 k) at a time
        const float* ab = A + (MBK * r + 32 * half + h) * XLD + l31;
        const float* bb = Wb + (32 * half + h) * WLD + mcol;
        float av[16], bv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { av[i] = ab[2 * i * XLD]; bv[i] = bb[2 * i * WLD]; }
        if (kind == 2) {
#pragma unroll
          for (int i = 0; i < 16; ++i) acc_o = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc_o, 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc, 0, 0, 0);
        }
      }
      if (r == SPP - 1) {
        if (kind == 1) dump(L.Cs[c & 1], acc);
        else if (kind == 0) {
          dump(L.Cs[0], acc);
          __syncthreads();                            // Wo accumulators in Cs; every matrix wave is done reading ctx from Xs
          float o[4][2], mr[4][2];
          ln_stage(L.Cs[0], pv_bo, a.drop_ctx, step_ctx, pv_g1, pv_be1, L.Xs, y1r, o, mr);
          ln_store(a.y1, a.ln1, a.st1, y1r, o, mr);
        } else if (s == NS - 1) {
          dump(L.Cs[0], acc_o);
          __syncthreads();
          float o[4][2], mr[4][2];
          ln_stage(L.Cs[0], pv_b2, a.drop_ff2, step_ff2, pv_gf, pv_bef, nullptr, y1r, o, mr);
          if (a.fold_score) score_stage(o, [&]() { ln_store(a.y2, a.enc, a.stf, y1r, o, mr); });
          else ln_store(a.y2, a.enc, a.stf, y1r, o, mr);
        }
      }
      WS_STAMP(2 * s + 1);
      __syncthreads();
    }
    WS_STAMP(2 * NS);
    return;
  }

  // helper waves.  wnext holds slab s + 1 on entry of step s (loaded two steps earlier)
  auto helper_step = [&](const int s, float4 (&wnext)[WRN]) __attribute__((always_inline)) {
    int kind, c;
    prod(s / SPP, kind, c);
    const int r = s % SPP;
    // epilogue of the W1 chunk whose product ended just before this one: rows 8*hw + 4*r .. +3, columns 2*lane, 2*lane+1
    int pk = -1, pc = 0;
    if (s >= SPP) prod(s / SPP - 1, pk, pc);
    const bool epi = pk == 1;
    const int f0 = 128 * pc + 2 * lane;
    WS_STAMP(2 * s);
    // its two bias values are requested FIRST: behind the slab loads below they would make the compiler wait for
    // every load in flight (vmcnt(0)) — the whole L2 round trip of the prefetch, once per slab
    float2 bia = make_float2(0.f, 0.f);
    if (epi) bia = *reinterpret_cast<const float2*>(a.b1 + f0);
    // publish slab s + 1 to the other ring buffer (the matrix waves finished reading it a barrier ago), refill its
    // registers two slabs ahead (unconditional: past the end a cache hit that is dropped)
    if (s + 1 < NS) slab_store(L.Ws[(s + 1) & 1], wnext, htid);
    {
      const float* W; int ldw, n0, k0;
      slab_src(s + 3 < NS ? s + 3 : NS - 1, W, ldw, n0, k0);
      slab_load(W, ldw, n0, k0, wnext, htid);
    }
    if (epi) {
      const float* Cst = L.Cs[pc & 1];
      float* Hk = L.Hs[pc & 1];
      const int rl0 = 8 * hw + 4 * r, rb = opaque(m0) + rl0;
      Philox4 r0 = {0u, 0u, 0u, 0u}, r1 = {0u, 0u, 0u, 0u};
      if (a.drop_ff1.thr) {
        r0 = philox4x32_10((uint32_t)f0, (uint32_t)rb >> 2, a.drop_ff1.site, step_ff1, a.drop_ff1.k0, a.drop_ff1.k1);
        r1 = philox4x32_10((uint32_t)f0 + 1u, (uint32_t)rb >> 2, a.drop_ff1.site, step_ff1, a.drop_ff1.k0, a.drop_ff1.k1);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rl = rl0 + q, m = rb + q;
        const float p0 = Cst[rl * YLD + 2 * lane] + bia.x, p1 = Cst[rl * YLD + 2 * lane + 1] + bia.y;
        float h0 = gelu_tanh_f(p0), h1v = gelu_tanh_f(p1);
        if (a.drop_ff1.thr) {
          h0 *= drop_word(a.drop_ff1, q == 0 ? r0.x : (q == 1 ? r0.y : (q == 2 ? r0.z : r0.w)));
          h1v *= drop_word(a.drop_ff1, q == 0 ? r1.x : (q == 1 ? r1.y : (q == 2 ? r1.z : r1.w)));
        }
        Hk[(2 * lane) * XLD + rl] = h0; Hk[(2 * lane + 1) * XLD + rl] = h1v;
        if (m < M) {
          *reinterpret_cast<float2*>(a.a1 + (size_t)m * a.F + f0) = make_float2(p0, p1);
          *reinterpret_cast<float2*>(a.h1 + (size_t)m * a.F + f0) = make_float2(h0, h1v);
        }
      }
    }
    if (r == SPP - 1) {
      if (kind == 0) {
        __syncthreads();
        float o[4][2], mr[4][2];
        ln_stage(L.Cs[0], pv_bo, a.drop_ctx, step_ctx, pv_g1, pv_be1, L.Xs, y1r, o, mr);
        ln_store(a.y1, a.ln1, a.st1, y1r, o, mr);
      } else if (s == NS - 1) {
        __syncthreads();
        float o[4][2], mr[4][2];
        ln_stage(L.Cs[0], pv_b2, a.drop_ff2, step_ff2, pv_gf, pv_bef, nullptr, y1r, o, mr);
        if (a.fold_score) score_stage(o, [&]() { ln_store(a.y2, a.enc, a.stf, y1r, o, mr); });
        else ln_store(a.y2, a.enc, a.stf, y1r, o, mr);
      }
    }
    WS_STAMP(2 * s + 1);
    __syncthreads();
  };
  for (int s = 0; s < NS; s += 2) {                   // NS = 2 * NP is even; wr1 holds slab s + 1, wr0 slab s + 2
    helper_step(s, wr1);
    helper_step(s + 1, wr0);
  }
  WS_STAMP(2 * NS);
}

// ====================================================================== forward, bf16x3 form
// Measured (tools/micro/coexec.hip, bf16x3.hip; profiles/r02_mfma_notes.md): v_mfma_f32_32x32x2_f32 runs on the SIMD's
// vector ALUs — it does NOT overlap the other wave's VALU work (120 us + 277 us -> 385 us) and costs 4,445 cycles per
// 32x32x128 product.  The same product as SIX v_mfma_f32_32x32x16_bf16 over a three-way bf16 split of both operands
// (x = hi + mid + lo, 3 x 8 = 24 mantissa bits, exact;  hh + hm + mh + mm + hl + lh, fp32 accumulation) costs 1,457
// cycles and is as accurate (max error / sum|a b|: 1.10e-7 against 1.13e-7 for the fp32 MFMA, fp64 reference).
// This kernel is mlp_fwd_ws_kernel with that product: same roles, same product order, same epilogue arithmetic; the
// A operands (ctx, ln1, h1 chunks) are split where they are produced, the weights arrive pre-split (WSplit).
// LDS images are [row][k] bf16 with their 16-byte chunks XOR-swizzled so that ds_read_b128 of 16 rows is conflict-free
// without padding (155 KB in all): A tiles [32][128]: chunk ^ (row & 15);  weight slabs [128 n][32 k]: chunk ^ ((n >> 2) & 3).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define X3_BK 64          // reduction depth of one weight slab
#define X3_SPP 2          // slabs per 128-deep product
#define X3_CH 6           // 16-byte chunks of a slab per thread: 3 planes x 128 n x (X3_BK / 8) chunks / 512 threads
struct MlpX3Lds {         // exactly the 160 KiB a workgroup may own
  uint16_t Xa[3][MBM * MD];          // A planes of Wo / W1: ctx, then ln1
  uint16_t Ha[3][MBM * MD];          // A planes of W2: the current h1 chunk
  uint16_t Wb[2][3][MD * X3_BK];     // weight slab ring (its first floats double as the last stage's scratch `red`)
  float Cs[MBM * MD];                // accumulator tile handed to the element-wise stages (row-major, unpadded)
};
static_assert(sizeof(MlpX3Lds) <= 160 * 1024, "bf16x3 MLP: LDS budget");
__device__ inline int x3_a_off(int row, int k) { return row * MD + ((((k >> 3) ^ (row & 15)) << 3) | (k & 7)); }
// weight slab rows are X3_BK * 2 = 128 bytes: 8 chunks; 16 rows read the same logical chunk -> XOR with (n >> 1) & 7
__device__ inline int x3_b_off(int n, int k) { return n * X3_BK + ((((k >> 3) ^ ((n >> 1) & 7)) << 3) | (k & 7)); }
__device__ inline void x3_split(float x, uint16_t& h, uint16_t& m, uint16_t& l) {
  const __bf16 bh = (__bf16)x;
  float r = x - (float)bh;
  const __bf16 bm = (__bf16)r;
  r -= (float)bm;
  const __bf16 bl = (__bf16)r;
  h = __builtin_bit_cast(uint16_t, bh); m = __builtin_bit_cast(uint16_t, bm); l = __builtin_bit_cast(uint16_t, bl);
}
// two adjacent k of one row -> the three planes (4-byte stores)
__device__ inline void x3_put2(uint16_t (*planes)[MBM * MD], int row, int k, float v0, float v1) {
  uint16_t h0, m0, l0, h1, m1, l1;
  x3_split(v0, h0, m0, l0); x3_split(v1, h1, m1, l1);
  const int off = x3_a_off(row, k);
  *reinterpret_cast<uint32_t*>(&planes[0][off]) = (uint32_t)h0 | ((uint32_t)h1 << 16);
  *reinterpret_cast<uint32_t*>(&planes[1][off]) = (uint32_t)m0 | ((uint32_t)m1 << 16);
  *reinterpret_cast<uint32_t*>(&planes[2][off]) = (uint32_t)l0 | ((uint32_t)l1 << 16);
}
// six bf16 MFMAs = one exact-fp32-grade 32x32x16 product step (small terms first)
__device__ inline void x3_mma(f32x16& acc, const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

// Structure (per-slab stamps of the role-split form showed the helper waves, not the matrix pipe, on the critical path
// once the products were bf16): all 8 waves stream the weight slabs (6 chunks of 16 bytes per thread and slab, prefetched
// two slabs ahead); waves 0-3 multiply (24 MFMAs per slab); every element-wise stage — the two LayerNorms and the
// GELU / dropout between W1 chunk c and W2 chunk c — is shared by all 8 waves, 4 rows each, behind one barrier.
// Products run in their natural order  Wo, (W1c, W2c) for c = 0 .. n-1.
template <bool STAMP>
__global__ __launch_bounds__(WS_THREADS, 2) void mlp_fwd_x3_kernel(const MlpFwdArgs a, unsigned long long* stamp) {
  extern __shared__ float lds_raw[];
  MlpX3Lds& L = *reinterpret_cast<MlpX3Lds*>(lds_raw);
  float* red = reinterpret_cast<float*>(&L.Wb[0][0][0]);          // free once the last slab has been multiplied
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const bool is_m = wave < 4;
  const int m0 = blockIdx.x * MBM, M = a.M;
  const int mcol = (wave & 3) * 32 + l31;
  const int nchunk = a.F / 128;
  const int NP = 1 + 2 * nchunk, NS = X3_SPP * NP;
  const uint32_t step_ctx = drop_step(a.drop_ctx), step_ff1 = drop_step(a.drop_ff1), step_ff2 = drop_step(a.drop_ff2);

  // product p: 0 = Wo; 1 + 2c = W1 chunk c; 2 + 2c = W2 chunk c
  auto slab_src = [&](int s, const uint16_t*& base, size_t& psz, int& ld, int& n0, int& k0) {
    const int p = s / X3_SPP, r = s % X3_SPP;
    if (p == 0) { base = a.x3.nat[0]; psz = (size_t)MD * MD; ld = MD; n0 = 0; k0 = X3_BK * r; }
    else if (p & 1) { base = a.x3.nat[1]; psz = (size_t)a.F * MD; ld = MD; n0 = 128 * ((p - 1) >> 1); k0 = X3_BK * r; }
    else { base = a.x3.nat[2]; psz = (size_t)MD * a.F; ld = a.F; n0 = 0; k0 = 128 * ((p - 2) >> 1) + X3_BK * r; }
  };
  // chunk q = tid + 512 u of a slab: plane q / 1024, row (q % 1024) / 8, chunk q % 8
  auto slab_load = [&](int s, uint4 (&r)[X3_CH]) {
    const uint16_t* base; size_t psz; int ld, n0, k0;
    slab_src(s, base, psz, ld, n0, k0);
#pragma unroll
    for (int u = 0; u < X3_CH; ++u) {
      const int q = tid + WS_THREADS * u, pl = q >> 10, rem = q & 1023, n = rem >> 3, j = rem & 7;
      r[u] = *reinterpret_cast<const uint4*>(base + pl * psz + (size_t)(n0 + n) * ld + k0 + 8 * j);
    }
  };
  auto slab_store = [&](uint16_t (*Wbuf)[MD * X3_BK], const uint4 (&r)[X3_CH]) {
#pragma unroll
    for (int u = 0; u < X3_CH; ++u) {
      const int q = tid + WS_THREADS * u, pl = q >> 10, rem = q & 1023, n = rem >> 3, j = rem & 7;
      *reinterpret_cast<uint4*>(&Wbuf[pl][x3_b_off(n, 8 * j)]) = make_uint4(r[u].x, r[u].y, r[u].z, r[u].w);
    }
  };

  // ---- prologue
  uint4 wr0[X3_CH], wr1[X3_CH];
  slab_load(0, wr0);
  slab_load(1, wr1);
  {                                                   // ctx tile -> Xa planes: thread = (row, one 8-element chunk)
    const int row = tid >> 4, kc = tid & 15;
    float v[8];
    if (m0 + row < M) {
      const float4 v0 = *reinterpret_cast<const float4*>(a.ctx + (size_t)(m0 + row) * MD + 8 * kc);
      const float4 v1 = *reinterpret_cast<const float4*>(a.ctx + (size_t)(m0 + row) * MD + 8 * kc + 4);
      v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; e += 2) x3_put2(L.Xa, row, 8 * kc + e, v[e], v[e + 1]);
  }
  // rows 4*wave .. 4*wave+3, columns 2*lane and 2*lane + 1: this lane's elements in every element-wise stage
  float y1r[4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int m = m0 + 4 * wave + q;
    const float* src = a.xin + ((size_t)((m < M ? m : 0) / a.fan) * a.S + a.qpos) * MD;
    const float2 v = m < M ? *reinterpret_cast<const float2*>(src + 2 * lane) : make_float2(0.f, 0.f);
    y1r[q][0] = v.x; y1r[q][1] = v.y;
  }
  float itr[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
  float ibias[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.fold_score) {
    const ScoreArgs& S = a.sc;
    const int K1 = S.K + 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = m0 + 4 * wave + q;
      if (m < M) {
        const int b = fdiv(m, S.fK1), j = m - b * K1;
        int64_t idx = j == 0 ? S.target[b] : S.neg_items[(size_t)b * S.K + j - 1];
        idx = idx < 0 ? S.P : (idx > S.P ? S.P : idx);
        const float2 v = *reinterpret_cast<const float2*>(S.product_emb + (size_t)idx * MD + 2 * lane);
        itr[q][0] = v.x; itr[q][1] = v.y;
        if (S.bias_product) ibias[q] = S.product_bias[idx];
      }
    }
  }
  slab_store(L.Wb[0], wr0);
  slab_load(2 < NS ? 2 : NS - 1, wr0);
  __syncthreads();

  f32x16 acc, acc_o;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc_o[r] = 0.f; }

  // one LayerNorm stage, all 8 waves (see mlp_fwd_ws_kernel); the normalised rows go to Xa as bf16x3 planes
  auto ln_stage = [&](const float* __restrict__ bias, const DropSpec& drop, const uint32_t dstep,
                      const float* __restrict__ g, const float* __restrict__ bta, bool to_xa, float (&res)[4][2],
                      float (&o)[4][2], float (&mr)[4][2]) {
    const int rb = opaque(m0) + 4 * wave;
    const float2 bb = *reinterpret_cast<const float2*>(bias + 2 * lane);
    const float2 gg = *reinterpret_cast<const float2*>(g + 2 * lane), ee = *reinterpret_cast<const float2*>(bta + 2 * lane);
    Philox4 r0 = {0u, 0u, 0u, 0u}, r1 = {0u, 0u, 0u, 0u};
    if (drop.thr) {
      r0 = philox4x32_10((uint32_t)(2 * lane), (uint32_t)rb >> 2, drop.site, dstep, drop.k0, drop.k1);
      r1 = philox4x32_10((uint32_t)(2 * lane + 1), (uint32_t)rb >> 2, drop.site, dstep, drop.k0, drop.k1);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 4 * wave + q;
      const float2 cv = *reinterpret_cast<const float2*>(L.Cs + row * MD + 2 * lane);
      float v0 = cv.x + bb.x, v1 = cv.y + bb.y;
      if (drop.thr) {
        v0 *= drop_word(drop, q == 0 ? r0.x : (q == 1 ? r0.y : (q == 2 ? r0.z : r0.w)));
        v1 *= drop_word(drop, q == 0 ? r1.x : (q == 1 ? r1.y : (q == 2 ? r1.z : r1.w)));
      }
      v0 += res[q][0]; v1 += res[q][1];
      res[q][0] = v0; res[q][1] = v1;
      const float mean = wave_sum(v0 + v1) * (1.f / MD);
      const float d0 = v0 - mean, d1 = v1 - mean;
      const float rstd = 1.f / sqrtf(wave_sum(d0 * d0 + d1 * d1) * (1.f / MD) + 1e-6f);
      const float o0 = d0 * rstd * gg.x + ee.x, o1 = d1 * rstd * gg.y + ee.y;
      o[q][0] = o0; o[q][1] = o1;
      mr[q][0] = mean; mr[q][1] = rstd;
      if (to_xa) x3_put2(L.Xa, row, 2 * lane, o0, o1);
    }
  };
  auto ln_store = [&](float* out_pre, float* out_ln, float* stats, const float (&res)[4][2], const float (&o)[4][2],
                      const float (&mr)[4][2]) {
    const int rb = opaque(m0) + 4 * wave;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = rb + q;
      if (m < M) {
        *reinterpret_cast<float2*>(out_pre + (size_t)m * MD + 2 * lane) = make_float2(res[q][0], res[q][1]);
        *reinterpret_cast<float2*>(out_ln + (size_t)m * MD + 2 * lane) = make_float2(o[q][0], o[q][1]);
        if (lane == 0) { stats[2 * (size_t)m] = mr[q][0]; stats[2 * (size_t)m + 1] = mr[q][1]; }
      }
    }
  };
  // GELU / dropout stage between W1 chunk c and W2 chunk c, all 8 waves: a1 chunk in Cs -> h1 chunk planes in Ha
  auto gelu_stage = [&](const int c) {
    const int f0 = 128 * c + 2 * lane, rb = opaque(m0) + 4 * wave;
    const float2 bia = *reinterpret_cast<const float2*>(a.b1 + f0);
    Philox4 r0 = {0u, 0u, 0u, 0u}, r1 = {0u, 0u, 0u, 0u};
    if (a.drop_ff1.thr) {
      r0 = philox4x32_10((uint32_t)f0, (uint32_t)rb >> 2, a.drop_ff1.site, step_ff1, a.drop_ff1.k0, a.drop_ff1.k1);
      r1 = philox4x32_10((uint32_t)f0 + 1u, (uint32_t)rb >> 2, a.drop_ff1.site, step_ff1, a.drop_ff1.k0, a.drop_ff1.k1);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rl = 4 * wave + q, m = rb + q;
      const float2 cv = *reinterpret_cast<const float2*>(L.Cs + rl * MD + 2 * lane);
      const float p0 = cv.x + bia.x, p1 = cv.y + bia.y;
      float h0 = gelu_tanh_f(p0), h1v = gelu_tanh_f(p1);
      if (a.drop_ff1.thr) {
        h0 *= drop_word(a.drop_ff1, q == 0 ? r0.x : (q == 1 ? r0.y : (q == 2 ? r0.z : r0.w)));
        h1v *= drop_word(a.drop_ff1, q == 0 ? r1.x : (q == 1 ? r1.y : (q == 2 ? r1.z : r1.w)));
      }
      x3_put2(L.Ha, rl, 2 * lane, h0, h1v);
      if (m < M) {
        *reinterpret_cast<float2*>(a.a1 + (size_t)m * a.F + f0) = make_float2(p0, p1);
        *reinterpret_cast<float2*>(a.h1 + (size_t)m * a.F + f0) = make_float2(h0, h1v);
      }
    }
  };
  // folded scoring (see mlp_fwd_ws_kernel)
  auto score_stage = [&](const float (&o)[4][2], auto&& stores) {
    const ScoreArgs& S = a.sc;
    const int K1 = S.K + 1;
    float scq = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float sdot = wave_sum(o[q][0] * itr[q][0] + o[q][1] * itr[q][1]) + ibias[q];
      scq = lane == q ? sdot : scq;
    }
    const int mq = m0 + 4 * wave + (lane & 3);
    const bool rowq = lane < 4 && mq < M;
    const int bq = fdiv(rowq ? mq : 0, S.fK1), jq = (rowq ? mq : 0) - bq * K1;
    const float twq = jq == 0 ? -(S.pos_weight ? (float)S.K : 1.f) : 1.f;
    const float termq = fabsf(twq) * softplus_f(twq < 0.f ? -scq : scq);
    const float cps = wave_sum(rowq ? termq : 0.f);
    if (lane == 0) red[wave] = cps;
    __syncthreads();
    unsigned long long mine = 0ull, old = 0ull;
    const unsigned long long one = 1ull << 48, mask = one - 1ull;
    if (tid == 0) {
      const float p = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
      unsigned long long* tk = reinterpret_cast<unsigned long long*>(S.ticket);
      mine = (unsigned long long)(long long)__float2ll_rn(p * 1048576.f) | one;
      old = __hip_atomic_fetch_add(tk, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (rowq) { S.item_scores[mq] = scq; S.item_terms[mq] = termq; }
    stores();
    if (tid == 0) {
      float last = 0.f;
      if ((old >> 48) == gridDim.x - 1u) {
        last = 1.f;
        red[9] = (float)((double)((old & mask) + (mine & mask)) * (1.0 / 1048576.0));
      }
      red[8] = last;
    }
    __syncthreads();
    if (red[8] == 0.f) return;
    if (wave == 0) {
      float il = 0.f;
      for (int i = lane; i < S.word_nblk; i += 64) il += S.word_blk[i];
      il = wave_sum(il);
      if (lane == 0) {
        const float ps = red[9] / (float)S.B;
        il /= (float)S.B;
        S.loss3[0] = ps + il; S.loss3[1] = ps; S.loss3[2] = il;
        if (S.loss_acc) { S.loss_acc[0] += ps; S.loss_acc[1] += il; }
      }
    }
  };
  auto dump = [&](f32x16& v) {                         // matrix waves: accumulator tile -> Cs
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      L.Cs[((r & 3) + 8 * (r >> 2) + 4 * h) * MD + mcol] = v[r];
      v[r] = 0.f;
    }
  };

  auto slab_step = [&](const int s, uint4 (&wnext)[X3_CH]) __attribute__((always_inline)) {
    const int p = s / X3_SPP, r = s % X3_SPP;
    const bool is_w2 = p >= 2 && (p & 1) == 0;
    WS_STAMP(2 * s);
    if (is_m) {
      const uint16_t (*A)[MBM * MD] = is_w2 ? L.Ha : L.Xa;
      const uint16_t (*Wb)[MD * X3_BK] = L.Wb[s & 1];
#pragma unroll
      for (int ks = 0; ks < X3_BK / 16; ++ks) {
        bf16x8 av[3], bv[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          av[pl] = *reinterpret_cast<const bf16x8*>(&A[pl][x3_a_off(l31, X3_BK * r + 16 * ks + 8 * h)]);
          bv[pl] = *reinterpret_cast<const bf16x8*>(&Wb[pl][x3_b_off(mcol, 16 * ks + 8 * h)]);
        }
        if (is_w2) x3_mma(acc_o, av, bv); else x3_mma(acc, av, bv);
      }
    }
    // every wave: publish slab s + 1 (its ring buffer was multiplied a barrier ago), refill the registers two slabs ahead
    if (s + 1 < NS) slab_store(L.Wb[(s + 1) & 1], wnext);
    slab_load(s + 3 < NS ? s + 3 : NS - 1, wnext);
    if (r == X3_SPP - 1) {
      if (p == 0) {                                   // Wo done: y1 = dropout(. + bo) + x ; LayerNorm_ff -> ln1
        if (is_m) dump(acc);
        __syncthreads();
        float o[4][2], mr[4][2];
        ln_stage(a.bo, a.drop_ctx, step_ctx, a.g1, a.be1, true, y1r, o, mr);
        ln_store(a.y1, a.ln1, a.st1, y1r, o, mr);
      } else if (p & 1) {                             // W1 chunk done: GELU / dropout -> h1 chunk
        if (is_m) dump(acc);
        __syncthreads();
        gelu_stage((p - 1) >> 1);
      } else if (s == NS - 1) {                       // last W2 chunk: y2 = dropout(. + b2) + y1 ; final LayerNorm -> enc
        if (is_m) dump(acc_o);
        __syncthreads();
        float o[4][2], mr[4][2];
        ln_stage(a.b2, a.drop_ff2, step_ff2, a.gf, a.bef, false, y1r, o, mr);
        if (a.fold_score) score_stage(o, [&]() { ln_store(a.y2, a.enc, a.stf, y1r, o, mr); });
        else ln_store(a.y2, a.enc, a.stf, y1r, o, mr);
      }
    }
    WS_STAMP(2 * s + 1);
    __syncthreads();
  };
  for (int s = 0; s < NS; s += 2) {                   // NS = 2 * NP is even; wr1 holds slab s + 1, wr0 slab s + 2
    slab_step(s, wr1);
    slab_step(s + 1, wr0);
  }
  WS_STAMP(2 * NS);
}

// Opt-in (PS_MLP_X3=1): validated by the whole parity suite, but at C2 it does not pay — the fused forward goes 59.4 ->
// 54.7 us while the embed launch grows by the re-split, 0.3186 against 0.3178 ms/step (profiles/r02_mlp_notes.md): once the
// products are cheap the kernel is bound by its element-wise VALU work (Philox, GELU, LayerNorm, the splits) and barriers.
bool mlp_x3_enabled(int F) {
  static const bool on = getenv("PS_MLP_X3") && atoi(getenv("PS_MLP_X3")) != 0;
  return on && ps_fusion_enabled() && F % 128 == 0 && F >= 256;
}
int64_t mlp_x3_floats(int d, int F) {      // 2 layouts x 3 planes x (d*d + 2*d*F) bf16
  return ((int64_t)2 * 3 * ((int64_t)d * d + 2 * (int64_t)d * F) * 2 + 3) / 4 + 16;
}

static bool mlp_ws_enabled() {
  static const bool on = !(getenv("PS_MLP_WS") && atoi(getenv("PS_MLP_WS")) == 0);
  return on;
}
bool mlp_fwd_can_fold_score(int M, int F, int d) {
  static const bool fold_on = !(getenv("PS_NO_FOLD_SCORE") && atoi(getenv("PS_NO_FOLD_SCORE")) != 0);
  return fold_on && ps_fusion_enabled() && (mlp_ws_enabled() || mlp_x3_enabled(F)) && d == MD && F % 128 == 0 && F >= 256 && M > 0 &&
         ps_cdiv(M, MBM) <= 256;               // one workgroup per CU, all resident: the ticket hand-off's measured regime
}

int launch_mlp_fwd_fused(const MlpFwdArgs& a, hipStream_t st) {
  PS_REQUIRE(a.F % 128 == 0 && a.M > 0, "fused mlp: F=%d M=%d", a.F, a.M);
  static bool attr_set = false;
  const size_t lds = sizeof(MlpLds);
  if (!attr_set) {
    PS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fwd_fused_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  KTimeScope kt("mlp_fwd", st);
  PS_REQUIRE(!a.fold_score || mlp_fwd_can_fold_score(a.M, a.F, MD), "fused mlp: folded scoring needs the wave-specialised form");
  PS_REQUIRE(!a.fold_score || a.x3.on || mlp_ws_enabled(), "fused mlp: folded scoring needs an 8-wave form");
  if (a.x3.on) {
    static bool x3_attr = false;
    if (!x3_attr) {
      PS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fwd_x3_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MlpX3Lds)));
      PS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fwd_x3_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MlpX3Lds)));
      x3_attr = true;
    }
    if (g_ws_stamp)
      hipLaunchKernelGGL(mlp_fwd_x3_kernel<true>, dim3(ps_cdiv(a.M, MBM)), dim3(WS_THREADS), sizeof(MlpX3Lds), st, a, g_ws_stamp);
    else
      hipLaunchKernelGGL(mlp_fwd_x3_kernel<false>, dim3(ps_cdiv(a.M, MBM)), dim3(WS_THREADS), sizeof(MlpX3Lds), st, a,
                         (unsigned long long*)nullptr);
    PS_LAUNCH_CHECK();
    return PS_OK;
  }
  if (mlp_ws_enabled() && a.F >= 256) {
    static bool ws_attr = false;
    if (!ws_attr) {
      PS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fwd_ws_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MlpWsLds)));
      PS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fwd_ws_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MlpWsLds)));
      ws_attr = true;
    }
    if (g_ws_stamp)
      hipLaunchKernelGGL(mlp_fwd_ws_kernel<true>, dim3(ps_cdiv(a.M, MBM)), dim3(WS_THREADS), sizeof(MlpWsLds), st, a, g_ws_stamp);
    else
      hipLaunchKernelGGL(mlp_fwd_ws_kernel<false>, dim3(ps_cdiv(a.M, MBM)), dim3(WS_THREADS), sizeof(MlpWsLds), st, a,
                         (unsigned long long*)nullptr);
    PS_LAUNCH_CHECK();
    return PS_OK;
  }
  hipLaunchKernelGGL(mlp_fwd_fused_kernel, dim3(ps_cdiv(a.M, MBM)), dim3(256), lds, st, a);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

// ====================================================================== backward
// Same ownership as the forward: a workgroup keeps 32 replica rows for the whole chain, weights stream through the slab
// ring.  Every product here is  grad[m][n] = sum_k g[m][k] * W[k][n]  with W read in its stored order (W2 [d][F],
// W1 [F][d], Wo [d][d] are all [k][n] for the input-gradient), so slabs are loaded AND stored as 16-byte rows.
#define WLB 132           // backward weight slabs: [k][col], row stride 528 B (16-byte aligned ds_write_b128)

struct MlpBwdLds {
  float Xs[MD * XLD];           // A operand of the W2 / Wo products: do2, then dout   (k-major)
  float Hs[MD * XLD];           // A operand of the W1 product: d a1 chunk (k-major); aliased as Y staging
  float Ws[2][MBK * WLB];       // weight slab ring
  float Cs[3][4][MD];           // column sums of the two LayerNorm backwards, per wave
};

// slab of MBK k-rows x 128 columns from a [k][n] (n contiguous) matrix
__device__ inline void slab_load_kn(const float* __restrict__ W, int ldw, int k0, int n0, float4 (&r)[WRN], int tid) {
#pragma unroll
  for (int u = 0; u < WRN; ++u) r[u] = *reinterpret_cast<const float4*>(W + (size_t)(k0 + (tid >> 5) + 8 * u) * ldw + n0 + 4 * (tid & 31));
}
__device__ inline void slab_store_kn(float* Wsb, const float4 (&r)[WRN], int tid) {
#pragma unroll
  for (int u = 0; u < WRN; ++u) {   // component-wise: copying r[u] whole keeps the slab registers in a scratch alloca (SROA gives up)
    const float4 v = make_float4(r[u].x, r[u].y, r[u].z, r[u].w);
    *reinterpret_cast<float4*>(Wsb + ((tid >> 5) + 8 * u) * WLB + 4 * (tid & 31)) = v;
  }
}

// LayerNorm backward of the 32 x 128 tile: wave w owns rows 8w..8w+7, lane the columns {lane, lane+64}.
//   dyv(i, a, b): grad wrt the LN output at row 8w+i, columns lane (a) and lane + 64 (b); called for rows < M only
//   x, stats, g: LN input rows (global, ld = MD), {mean, rstd} per row, gamma
//   res        : added to dx (or null);  drop: site of the dropout applied to the result for `dropped`
// Results: dxr[i][j] = dx (+res), dropped value written k-major to Xk and row-major to `out_drop` (global), dx to
// `out_dx` when given; column sums {dy*xhat, dy, dropped} accumulated into Cs[.][wave][.].
template <bool HAS_RES, class DyF>
__device__ __forceinline__ void tile_ln_bwd(DyF dyv, const float* __restrict__ x, const float* __restrict__ stats,
                                   const float* __restrict__ g, const float (&res)[8][2], const DropSpec& drop,
                                   float (&dxr)[8][2], float* Xk, float* out_dx, float* out_drop, float (*Cs)[4][MD],
                                   int m0, int M, int wave, int lane) {
  const float g0 = g[lane], g1 = g[lane + 64];
  float ag[2] = {0.f, 0.f}, ab[2] = {0.f, 0.f}, ac[2] = {0.f, 0.f};
#pragma unroll
  for (int g4 = 0; g4 < 2; ++g4) {
    const int rb = m0 + wave * 8 + 4 * g4;            // multiple of 4: the four rows share their Philox calls
    Philox4 r0 = {0u, 0u, 0u, 0u}, r1 = {0u, 0u, 0u, 0u};
    if (drop.thr) {
      r0 = philox4x32_10((uint32_t)lane, (uint32_t)rb >> 2, drop.site, drop_step(drop), drop.k0, drop.k1);
      r1 = philox4x32_10((uint32_t)lane + 64u, (uint32_t)rb >> 2, drop.site, drop_step(drop), drop.k0, drop.k1);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = 4 * g4 + q, row = wave * 8 + i, m = rb + q;
      const bool ok = m < M;
      const float mean = ok ? stats[2 * (size_t)m] : 0.f, rstd = ok ? stats[2 * (size_t)m + 1] : 0.f;
      const float x0 = ok ? x[(size_t)m * MD + lane] : 0.f, x1 = ok ? x[(size_t)m * MD + lane + 64] : 0.f;
      float dy0 = 0.f, dy1 = 0.f;
      if (ok) dyv(i, dy0, dy1);
      const float xh0 = (x0 - mean) * rstd, xh1 = (x1 - mean) * rstd;
      const float dh0 = dy0 * g0, dh1 = dy1 * g1;
      const float s1 = wave_sum(dh0 + dh1) * (1.f / MD);
      const float s2 = wave_sum(dh0 * xh0 + dh1 * xh1) * (1.f / MD);
      float d0 = rstd * (dh0 - s1 - xh0 * s2), d1 = rstd * (dh1 - s1 - xh1 * s2);
      if (HAS_RES) { d0 += res[i][0]; d1 += res[i][1]; }
      dxr[i][0] = d0; dxr[i][1] = d1;
      float v0 = d0, v1 = d1;
      if (drop.thr) {
        v0 *= drop_word(drop, q == 0 ? r0.x : (q == 1 ? r0.y : (q == 2 ? r0.z : r0.w)));
        v1 *= drop_word(drop, q == 0 ? r1.x : (q == 1 ? r1.y : (q == 2 ? r1.z : r1.w)));
      }
      Xk[lane * XLD + row] = v0; Xk[(lane + 64) * XLD + row] = v1;
      if (ok) {
        if (out_dx) { out_dx[(size_t)m * MD + lane] = d0; out_dx[(size_t)m * MD + lane + 64] = d1; }
        if (out_drop) { out_drop[(size_t)m * MD + lane] = v0; out_drop[(size_t)m * MD + lane + 64] = v1; }
      }
      ag[0] += dy0 * xh0; ag[1] += dy1 * xh1;
      ab[0] += dy0; ab[1] += dy1;
      ac[0] += v0; ac[1] += v1;
    }
  }
  Cs[0][wave][lane] = ag[0]; Cs[0][wave][lane + 64] = ag[1];
  Cs[1][wave][lane] = ab[0]; Cs[1][wave][lane + 64] = ab[1];
  Cs[2][wave][lane] = ac[0]; Cs[2][wave][lane + 64] = ac[1];
}
// after a barrier: park the workgroup's three column sums  part[blk][3][128]
__device__ inline void park_cs(const float (*Cs)[4][MD], float* part, int tid) {
  for (int t = tid; t < 3 * MD; t += 256) {
    const int which = t >> 7, colx = t & 127;
    part[((size_t)blockIdx.x * 3 + which) * MD + colx] =
        (Cs[which][0][colx] + Cs[which][1][colx]) + (Cs[which][2][colx] + Cs[which][3][colx]);
  }
}

__global__ __launch_bounds__(256, 1) void mlp_bwd_fused_kernel(const MlpBwdArgs a) {
  fork_signal(a.sig, a.sigval);
  extern __shared__ float lds_raw[];
  MlpBwdLds& L = *reinterpret_cast<MlpBwdLds*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * MBM, M = a.M;
  const int col = wave * 32 + l31;
  const int nchunk = a.F / 128;
  const int NC = 2 * SPP * nchunk;                  // slabs of the chunk loop; then SPP slabs of Wo
  const int NS = NC + SPP;

  auto slab_src = [&](int s, const float*& W, int& ldw, int& k0, int& n0) {
    if (s >= NC) { W = a.wo; ldw = MD; k0 = MBK * (s - NC); n0 = 0; return; }
    const int c = s / (2 * SPP), r = s % (2 * SPP);
    if (r < SPP) { W = a.w2; ldw = a.F; k0 = MBK * r; n0 = 128 * c; }          // d h1[:, chunk] = do2 . W2[:, chunk]
    else { W = a.w1; ldw = MD; k0 = 128 * c + MBK * (r - SPP); n0 = 0; }       // d ln1 += d a1[:, chunk] . W1[chunk, :]
  };

  float4 wr0[WRN], wr1[WRN];
  {
    const float* W; int ldw, k0, n0;
    slab_src(0, W, ldw, k0, n0);
    slab_load_kn(W, ldw, k0, n0, wr0, tid);
    slab_src(1, W, ldw, k0, n0);
    slab_load_kn(W, ldw, k0, n0, wr1, tid);
  }
  // ---- final LayerNorm backward (transformer.py:86) -> d y2 (kept: residual of the FF LayerNorm), do2 -> Xs
  float dy2r[8][2];
  {
    if (a.item_scores) {
      // d enc taken straight from the score: row m = (b, j) of enc was dotted with item row idx(b, j), so
      // d enc[m] = loss'(score[m]) * product_emb[idx]  (item_transformer.py:485,493-494,500-514) — the score backward
      // launch then only scatters into the tables and leaves the step's dependent chain (it runs on the side stream)
      const float invB = a.scale * (a.scale_dev ? *a.scale_dev : 1.f) / (float)a.B;
      const float wpos = a.pos_weight ? (float)a.K : 1.f;
      const int K1 = a.K + 1;
      auto dyv = [&](int i, float& d0, float& d1) {
        const int m = m0 + wave * 8 + i;
        const int b = m / K1, j = m - b * K1;
        int64_t idx = j == 0 ? a.target[b] : a.neg_items[(size_t)b * a.K + j - 1];
        idx = idx < 0 ? a.P : (idx > a.P ? a.P : idx);
        const float sc = a.item_scores[m];
        const float ds = (j == 0 ? wpos * (sigmoid_f(sc) - 1.f) : sigmoid_f(sc)) * invB;
        const float* row = a.product_emb + (size_t)idx * MD;
        d0 = ds * row[lane]; d1 = ds * row[lane + 64];
      };
      tile_ln_bwd<false>(dyv, a.y2, a.stf, a.gf, dy2r, a.drop_ff2, dy2r, L.Xs, nullptr, a.do2, L.Cs, m0, M, wave, lane);
    } else {
      const float* de = a.denc;
      auto dyv = [&](int i, float& d0, float& d1) {
        const float* p = de + (size_t)(m0 + wave * 8 + i) * MD;
        d0 = p[lane]; d1 = p[lane + 64];
      };
      tile_ln_bwd<false>(dyv, a.y2, a.stf, a.gf, dy2r, a.drop_ff2, dy2r, L.Xs, nullptr, a.do2, L.Cs, m0, M, wave, lane);
    }
  }
  slab_store_kn(L.Ws[0], wr0, tid);
  __syncthreads();
  park_cs(L.Cs, a.part_f, tid);

  f32x16 acc, acc_o;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc_o[r] = 0.f; }
  int buf = 0;

  auto slab_step = [&](const int s, float4 (&wfree)[WRN], const float4 (&wnext)[WRN]) __attribute__((always_inline)) {
    {
      const float* W; int ldw, k0, n0;
      slab_src(s + 2 < NS ? s + 2 : NS - 1, W, ldw, k0, n0);
      slab_load_kn(W, ldw, k0, n0, wfree, tid);
    }
    const int r8 = s % (2 * SPP);
    const bool is_w1 = s < NC && r8 >= SPP;
    const int ka = s >= NC ? MBK * (s - NC) : (is_w1 ? MBK * (r8 - SPP) : MBK * r8);
    const float* A = is_w1 ? L.Hs : L.Xs;
    const bool gelu_stage = s < NC && r8 == SPP - 1;
    // pre-activations of this chunk's epilogue: requested before the MFMA block, consumed after it
    float a1v[16];
    if (gelu_stage) {
      const int f = 128 * (s / (2 * SPP)) + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        a1v[r] = m < M ? a.a1[(size_t)m * a.F + f] : 0.f;
      }
    }
#pragma unroll
    for (int half = 0; half < MBK / 32; ++half) {
      const float* ab = A + (ka + 32 * half + h) * XLD + l31;
      const float* bb = L.Ws[buf] + (32 * half + h) * WLB + col;
      float av[16], bv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) { av[i] = ab[2 * i * XLD]; bv[i] = bb[2 * i * WLB]; }
      if (is_w1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc_o = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc_o, 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc, 0, 0, 0);
      }
    }

    if (gelu_stage) {
      // d a1 = (do2 . W2) * gelu'(a1) * dropout  (neural.py:30-33 backwards) -> Hs (k-major) + global; b1 column sums
      const int f = 128 * (s / (2 * SPP)) + col;
      const int mm0 = opaque(m0);
      float cs = 0.f;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int rb = mm0 + 8 * gq + 4 * h;
        Philox4 rnd = {0u, 0u, 0u, 0u};
        if (a.drop_ff1.thr) rnd = philox4x32_10((uint32_t)f, (uint32_t)rb >> 2, a.drop_ff1.site, drop_step(a.drop_ff1),
                                               a.drop_ff1.k0, a.drop_ff1.k1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 4 * gq + q;
          float v = acc[r] * gelu_tanh_grad(a1v[r]);
          if (a.drop_ff1.thr) v *= drop_word(a.drop_ff1, q == 0 ? rnd.x : (q == 1 ? rnd.y : (q == 2 ? rnd.z : rnd.w)));
          const int lrow = 8 * gq + 4 * h + q;
          L.Hs[col * XLD + lrow] = v;
          if (rb + q < M) a.da1[(size_t)(rb + q) * a.F + f] = v;
          cs += v;
          acc[r] = 0.f;
        }
      }
      cs += __shfl_xor(cs, 32, 64);                 // the two half-waves hold the other 16 rows of the same column
      if (h == 0) a.part_b1[(size_t)blockIdx.x * 3 * a.F + f] = cs;
    } else if (s == NC - 1) {
      // d ln1 complete -> FF LayerNorm backward (+ d y2 residual) -> dy1, dout = dropout(dy1) -> Xs
      float* Y = L.Hs;
      __syncthreads();                              // every wave is done reading the last d a1 chunk from Hs
#pragma unroll
      for (int r = 0; r < 16; ++r) Y[((r & 3) + 8 * (r >> 2) + 4 * h) * YLD + col] = acc_o[r];
      __syncthreads();
      float dy1r[8][2];
      auto dyv = [&](int i, float& d0, float& d1) { d0 = Y[(wave * 8 + i) * YLD + lane]; d1 = Y[(wave * 8 + i) * YLD + lane + 64]; };
      const bool same = a.dout == a.dy1;
      tile_ln_bwd<true>(dyv, a.y1, a.st1, a.g1, dy2r, a.drop_ctx, dy1r, L.Xs, a.dy1, same ? nullptr : a.dout, L.Cs,
                  opaque(m0), M, wave, lane);
      __syncthreads();
      park_cs(L.Cs, a.part_1, tid);
    } else if (s == NS - 1) {
      // d ctx = dout . Wo  (neural.py:228-231 backwards)
      const int mm0 = opaque(m0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mm0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < M) a.dctx[(size_t)m * MD + col] = acc[r];
      }
    }

    if (s + 1 < NS) slab_store_kn(L.Ws[buf ^ 1], wnext, tid);
    __syncthreads();
    buf ^= 1;
  };
  for (int s = 0; s < NS; s += 2) {
    slab_step(s, wr0, wr1);
    slab_step(s + 1, wr1, wr0);
  }
}

// ====================================================================== backward, wave-specialised form
// Same roles as mlp_fwd_ws_kernel: waves 0-3 multiply, waves 4-7 stream the weight slabs and run the GELU' / dropout
// epilogue of W2 chunk c (its pre-activations requested a step ahead) while the matrix waves multiply the next product.
// Product order  W2c0, W2c1, W1c0, W2c2, W1c1, ..., W2c(n-1), W1c(n-2), W1c(n-1), Wo ;  the two LayerNorm backwards
// (before the first product, before Wo) are shared by all 8 waves, 4 rows each.
struct MlpBwdWsLds {
  float Xs[MD * XLD];           // A operand of the W2 / Wo products: do2, then dout       (k-major)
  float Hs[2][MD * XLD];        // A operand of the W1 product: d a1 chunk c in Hs[c & 1]  (k-major)
  float Ws[2][MBK * WLB];       // weight slab ring
  float Cs[2][MBM * YLD];       // accumulator tiles handed to the epilogues (row-major); [1] doubles as the LayerNorm
                                // stages' column-sum scratch [3][8 waves][128]
  float b1red[2][4][MD];        // b1 column sums of the four helper waves, chunk parity: summed before they are parked
};
static_assert(sizeof(MlpBwdWsLds) <= 160 * 1024, "wave-specialised MLP backward: LDS budget");
static_assert(3 * 8 * MD <= MBM * YLD, "column-sum scratch must fit an accumulator tile");

__global__ __launch_bounds__(WS_THREADS, 2) void mlp_bwd_ws_kernel(const MlpBwdArgs a) {
  fork_signal(a.sig, a.sigval);
  extern __shared__ float lds_raw[];
  MlpBwdWsLds& L = *reinterpret_cast<MlpBwdWsLds*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const bool is_m = wave < 4;
  const int htid = tid & 255, hw = wave & 3;
  const int m0 = blockIdx.x * MBM, M = a.M;
  const int mcol = wave * 32 + l31;
  const int nchunk = a.F / 128;
  const int NP = 2 * nchunk + 1, NS = SPP * NP;
  float (*Csum)[8][MD] = reinterpret_cast<float (*)[8][MD]>(L.Cs[1]);
  const uint32_t step_ctx = drop_step(a.drop_ctx), step_ff1 = drop_step(a.drop_ff1), step_ff2 = drop_step(a.drop_ff2);

  // product p: kind 0 = W2 chunk c (d h1), 1 = W1 chunk c (d ln1 +=), 2 = Wo (d ctx)
  auto prod = [&](int p, int& kind, int& c) {
    if (p == 2 * nchunk) { kind = 2; c = 0; }
    else if (p < 2) { kind = 0; c = p; }
    else if (p == 2 * nchunk - 1) { kind = 1; c = nchunk - 1; }
    else if (p & 1) { kind = 0; c = (p + 1) >> 1; }
    else { kind = 1; c = (p >> 1) - 1; }
  };
  auto slab_src = [&](int s, const float*& W, int& ldw, int& k0, int& n0) {
    int kind, c;
    prod(s / SPP, kind, c);
    const int r = s % SPP;
    if (kind == 0) { W = a.w2; ldw = a.F; k0 = MBK * r; n0 = 128 * c; }            // d h1[:, chunk] = do2 . W2[:, chunk]
    else if (kind == 1) { W = a.w1; ldw = MD; k0 = 128 * c + MBK * r; n0 = 0; }    // d ln1 += d a1[:, chunk] . W1[chunk, :]
    else { W = a.wo; ldw = MD; k0 = MBK * r; n0 = 0; }                              // d ctx = dout . Wo
  };

  float4 wr0[WRN], wr1[WRN];
  if (!is_m) {
    const float* W; int ldw, k0, n0;
    slab_src(0, W, ldw, k0, n0);
    slab_load_kn(W, ldw, k0, n0, wr0, htid);
    slab_src(1, W, ldw, k0, n0);
    slab_load_kn(W, ldw, k0, n0, wr1, htid);
  }

  // LayerNorm backward of rows 4*wave .. 4*wave+3 (columns lane, lane + 64), all 8 waves:
  //   dy(q, d0, d1): grad wrt the LN output; x / stats / g: LN input rows, {mean, rstd}, gamma;  res: added to dx
  // dx (+res) -> dxr (and out_dx); dropout(dx) -> Xs (k-major) and out_drop; column sums {dy*xhat, dy, dropped} -> Csum
  auto ln_bwd_stage = [&](auto&& dy, const float* __restrict__ x, const float* __restrict__ stats,
                          const float* __restrict__ g, const bool has_res, const float (&res)[4][2], const DropSpec& drop,
                          const uint32_t dstep, float (&dxr)[4][2], float* out_dx, float* out_drop) {
    const int rb = opaque(m0) + 4 * wave;
    const float g0 = g[lane], g1 = g[lane + 64];
    float ag[2] = {0.f, 0.f}, ab[2] = {0.f, 0.f}, ac[2] = {0.f, 0.f};
    Philox4 r0 = {0u, 0u, 0u, 0u}, r1 = {0u, 0u, 0u, 0u};
    if (drop.thr) {
      r0 = philox4x32_10((uint32_t)lane, (uint32_t)rb >> 2, drop.site, dstep, drop.k0, drop.k1);
      r1 = philox4x32_10((uint32_t)lane + 64u, (uint32_t)rb >> 2, drop.site, dstep, drop.k0, drop.k1);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 4 * wave + q, m = rb + q;
      const bool ok = m < M;
      const float mean = ok ? stats[2 * (size_t)m] : 0.f, rstd = ok ? stats[2 * (size_t)m + 1] : 0.f;
      const float x0 = ok ? x[(size_t)m * MD + lane] : 0.f, x1 = ok ? x[(size_t)m * MD + lane + 64] : 0.f;
      float dy0 = 0.f, dy1 = 0.f;
      if (ok) dy(q, dy0, dy1);
      const float xh0 = (x0 - mean) * rstd, xh1 = (x1 - mean) * rstd;
      const float dh0 = dy0 * g0, dh1 = dy1 * g1;
      const float s1 = wave_sum(dh0 + dh1) * (1.f / MD);
      const float s2 = wave_sum(dh0 * xh0 + dh1 * xh1) * (1.f / MD);
      float d0 = rstd * (dh0 - s1 - xh0 * s2), d1 = rstd * (dh1 - s1 - xh1 * s2);
      if (has_res) { d0 += res[q][0]; d1 += res[q][1]; }
      dxr[q][0] = d0; dxr[q][1] = d1;
      float v0 = d0, v1 = d1;
      if (drop.thr) {
        v0 *= drop_word(drop, q == 0 ? r0.x : (q == 1 ? r0.y : (q == 2 ? r0.z : r0.w)));
        v1 *= drop_word(drop, q == 0 ? r1.x : (q == 1 ? r1.y : (q == 2 ? r1.z : r1.w)));
      }
      L.Xs[lane * XLD + row] = v0; L.Xs[(lane + 64) * XLD + row] = v1;
      if (ok) {
        if (out_dx) { out_dx[(size_t)m * MD + lane] = d0; out_dx[(size_t)m * MD + lane + 64] = d1; }
        if (out_drop) { out_drop[(size_t)m * MD + lane] = v0; out_drop[(size_t)m * MD + lane + 64] = v1; }
      }
      ag[0] += dy0 * xh0; ag[1] += dy1 * xh1;
      ab[0] += dy0; ab[1] += dy1;
      ac[0] += v0; ac[1] += v1;
    }
    Csum[0][wave][lane] = ag[0]; Csum[0][wave][lane + 64] = ag[1];
    Csum[1][wave][lane] = ab[0]; Csum[1][wave][lane + 64] = ab[1];
    Csum[2][wave][lane] = ac[0]; Csum[2][wave][lane + 64] = ac[1];
  };
  // after a barrier: park the workgroup's three column sums  part[blk][3][128]  (fixed order over the 8 waves)
  auto park = [&](float* part) {
    for (int t = tid; t < 3 * MD; t += WS_THREADS) {
      const int which = t >> 7, colx = t & 127;
      part[((size_t)blockIdx.x * 3 + which) * MD + colx] =
          ((Csum[which][0][colx] + Csum[which][1][colx]) + (Csum[which][2][colx] + Csum[which][3][colx])) +
          ((Csum[which][4][colx] + Csum[which][5][colx]) + (Csum[which][6][colx] + Csum[which][7][colx]));
    }
  };

  // ---- final LayerNorm backward (transformer.py:86) -> d y2 (kept: residual of the FF LayerNorm), do2 -> Xs
  float dy2r[4][2];
  {
    const float zero[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    if (a.item_scores) {
      // d enc straight from the score (MlpBwdArgs::item_scores): d enc[m] = loss'(score[m]) * product_emb[idx(b, j)]
      const float invB = a.scale * (a.scale_dev ? *a.scale_dev : 1.f) / (float)a.B;
      const float wpos = a.pos_weight ? (float)a.K : 1.f;
      const int K1 = a.K + 1;
      auto dyv = [&](int q, float& d0, float& d1) {
        const int m = m0 + 4 * wave + q;
        const int b = m / K1, j = m - b * K1;
        int64_t idx = j == 0 ? a.target[b] : a.neg_items[(size_t)b * a.K + j - 1];
        idx = idx < 0 ? a.P : (idx > a.P ? a.P : idx);
        const float sc = a.item_scores[m];
        const float ds = (j == 0 ? wpos * (sigmoid_f(sc) - 1.f) : sigmoid_f(sc)) * invB;
        const float* row = a.product_emb + (size_t)idx * MD;
        d0 = ds * row[lane]; d1 = ds * row[lane + 64];
      };
      ln_bwd_stage(dyv, a.y2, a.stf, a.gf, false, zero, a.drop_ff2, step_ff2, dy2r, nullptr, a.do2);
    } else {
      const float* de = a.denc;
      auto dyv = [&](int q, float& d0, float& d1) {
        const float* pq = de + (size_t)(m0 + 4 * wave + q) * MD;
        d0 = pq[lane]; d1 = pq[lane + 64];
      };
      ln_bwd_stage(dyv, a.y2, a.stf, a.gf, false, zero, a.drop_ff2, step_ff2, dy2r, nullptr, a.do2);
    }
  }
  if (!is_m) {
    slab_store_kn(L.Ws[0], wr0, htid);
    const float* W; int ldw, k0, n0;
    slab_src(2 < NS ? 2 : NS - 1, W, ldw, k0, n0);
    slab_load_kn(W, ldw, k0, n0, wr0, htid);
  }
  __syncthreads();
  park(a.part_f);
  __syncthreads();                                    // Csum (= Cs[1]) is free again before W2 chunk 1 is dumped into it

  f32x16 acc, acc_o;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc_o[r] = 0.f; }
  auto dump = [&](float* Cst, f32x16& v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      Cst[((r & 3) + 8 * (r >> 2) + 4 * h) * YLD + mcol] = v[r];
      v[r] = 0.f;
    }
  };
  // FF LayerNorm backward, all 8 waves: d ln1 tile in Cs[0] -> dy1 (+ d y2 residual), dout = dropout(dy1) -> Xs
  auto ff_ln_stage = [&]() {
    const float* Y = L.Cs[0];
    float dy1r[4][2];
    auto dyv = [&](int q, float& d0, float& d1) { d0 = Y[(4 * wave + q) * YLD + lane]; d1 = Y[(4 * wave + q) * YLD + lane + 64]; };
    const bool same = a.dout == a.dy1;
    ln_bwd_stage(dyv, a.y1, a.st1, a.g1, true, dy2r, a.drop_ctx, step_ctx, dy1r, a.dy1, same ? nullptr : a.dout);
    __syncthreads();
    park(a.part_1);
  };

  if (is_m) {
    for (int s = 0; s < NS; ++s) {
      int kind, c;
      prod(s / SPP, kind, c);
      const int r = s % SPP;
      const float* A = kind == 1 ? L.Hs[c & 1] : L.Xs;
      const float* Wb = L.Ws[s & 1];
#pragma unroll
      for (int half = 0; half < MBK / 32; ++half) {
        const float* ab = A + (MBK * r + 32 * half + h) * XLD + l31;
        const float* bb = Wb + (32 * half + h) * WLB + mcol;
        float av[16], bv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { av[i] = ab[2 * i * XLD]; bv[i] = bb[2 * i * WLB]; }
        if (kind == 1) {
#pragma unroll
          for (int i = 0; i < 16; ++i) acc_o = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc_o, 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc, 0, 0, 0);
        }
      }
      if (r == SPP - 1) {
        if (kind == 0) dump(L.Cs[c & 1], acc);
        else if (kind == 1 && c == nchunk - 1) {
          dump(L.Cs[0], acc_o);
          __syncthreads();                            // d ln1 in Cs[0]; every d a1 chunk has been consumed
          ff_ln_stage();
        } else if (kind == 2) {
          const int mm0 = opaque(m0);
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) {
            const int m = mm0 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
            if (m < M) a.dctx[(size_t)m * MD + mcol] = acc[rr];
          }
        }
      }
      __syncthreads();
    }
    return;
  }

  float csb[2] = {0.f, 0.f};                          // b1 column sums of this wave's 8 rows (two half steps)
  auto helper_step = [&](const int s, float4 (&wnext)[WRN]) __attribute__((always_inline)) {
    int kind, c;
    prod(s / SPP, kind, c);
    const int r = s % SPP;
    // epilogue of the W2 chunk whose product ended just before this one: rows 8*hw + 4*r .. +3, columns 2*lane, 2*lane+1
    int pk = -1, pc = 0;
    if (s >= SPP) prod(s / SPP - 1, pk, pc);
    const bool epi = pk == 0;
    const int f0 = 128 * pc + 2 * lane;
    const int rl0 = 8 * hw + 4 * r, rb = opaque(m0) + rl0;
    // its pre-activations are requested FIRST (see mlp_fwd_ws_kernel)
    float2 a1v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a1v[q] = make_float2(0.f, 0.f);
      if (epi && rb + q < M) a1v[q] = *reinterpret_cast<const float2*>(a.a1 + (size_t)(rb + q) * a.F + f0);
    }
    if (s + 1 < NS) slab_store_kn(L.Ws[(s + 1) & 1], wnext, htid);
    {
      const float* W; int ldw, k0, n0;
      slab_src(s + 3 < NS ? s + 3 : NS - 1, W, ldw, k0, n0);
      slab_load_kn(W, ldw, k0, n0, wnext, htid);
    }
    if (epi) {
      // d a1 = (do2 . W2) * gelu'(a1) * dropout  (neural.py:30-33 backwards) -> Hs (k-major) + global; b1 column sums
      const float* Cst = L.Cs[pc & 1];
      float* Hk = L.Hs[pc & 1];
      Philox4 r0 = {0u, 0u, 0u, 0u}, r1 = {0u, 0u, 0u, 0u};
      if (a.drop_ff1.thr) {
        r0 = philox4x32_10((uint32_t)f0, (uint32_t)rb >> 2, a.drop_ff1.site, step_ff1, a.drop_ff1.k0, a.drop_ff1.k1);
        r1 = philox4x32_10((uint32_t)f0 + 1u, (uint32_t)rb >> 2, a.drop_ff1.site, step_ff1, a.drop_ff1.k0, a.drop_ff1.k1);
      }
      if (r == 0) { csb[0] = 0.f; csb[1] = 0.f; }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rl = rl0 + q, m = rb + q;
        float v0 = Cst[rl * YLD + 2 * lane] * gelu_tanh_grad(a1v[q].x);
        float v1 = Cst[rl * YLD + 2 * lane + 1] * gelu_tanh_grad(a1v[q].y);
        if (a.drop_ff1.thr) {
          v0 *= drop_word(a.drop_ff1, q == 0 ? r0.x : (q == 1 ? r0.y : (q == 2 ? r0.z : r0.w)));
          v1 *= drop_word(a.drop_ff1, q == 0 ? r1.x : (q == 1 ? r1.y : (q == 2 ? r1.z : r1.w)));
        }
        Hk[(2 * lane) * XLD + rl] = v0; Hk[(2 * lane + 1) * XLD + rl] = v1;
        if (m < M) *reinterpret_cast<float2*>(a.da1 + (size_t)m * a.F + f0) = make_float2(v0, v1);
        csb[0] += v0; csb[1] += v1;
      }
      // the four helper waves' sums meet in LDS (chunk parity) and are parked as ONE row per workgroup behind this step's
      // barrier: part_b1[wg][slot 0][F] (a row per helper wave made the fold of the step's last launch walk 1,008 rows)
      if (r == SPP - 1) { L.b1red[pc & 1][hw][2 * lane] = csb[0]; L.b1red[pc & 1][hw][2 * lane + 1] = csb[1]; }
    }
    if (r == SPP - 1 && kind == 1 && c == nchunk - 1) {
      __syncthreads();
      ff_ln_stage();
    }
    __syncthreads();
    if (epi && r == SPP - 1 && hw == 0) {
      const float* q = &L.b1red[pc & 1][0][0];
      const float s0 = (q[2 * lane] + q[MD + 2 * lane]) + (q[2 * MD + 2 * lane] + q[3 * MD + 2 * lane]);
      const float s1 = (q[2 * lane + 1] + q[MD + 2 * lane + 1]) + (q[2 * MD + 2 * lane + 1] + q[3 * MD + 2 * lane + 1]);
      *reinterpret_cast<float2*>(a.part_b1 + ((size_t)blockIdx.x * 3) * a.F + f0) = make_float2(s0, s1);
    }
  };
  for (int s = 0; s < NS; s += 2) {
    helper_step(s, wr1);
    if (s + 1 < NS) helper_step(s + 1, wr0);
  }
}

bool mlp_bwd_ws_enabled() {
  static const bool on = !(getenv("PS_MLP_BWD_WS") && atoi(getenv("PS_MLP_BWD_WS")) == 0) && mlp_ws_enabled();
  return on;
}
int mlp_bwd_fused_blocks(int M) { return ps_cdiv(M, MBM); }
// parked rows of the b1 column sums: one per workgroup
int mlp_bwd_b1_rows(int M, int F) { (void)F; return ps_cdiv(M, MBM); }

int launch_mlp_bwd_fused(const MlpBwdArgs& a, hipStream_t st) {
  PS_REQUIRE(a.F % 128 == 0 && a.M > 0, "fused mlp backward: F=%d M=%d", a.F, a.M);
  PS_REQUIRE(a.part_f && a.part_1 && a.part_b1, "fused mlp backward: column sums must be parked");
  static bool attr_set = false;
  const size_t lds = sizeof(MlpBwdLds);
  if (!attr_set) {
    PS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_fused_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  KTimeScope kt("mlp_bwd", st);
  MlpBwdArgs b = a;
  b.sig = nullptr; b.sigval = 0;
  if (mlp_bwd_ws_enabled() && a.F >= 256) {
    static bool ws_attr = false;
    if (!ws_attr) {
      PS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_ws_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MlpBwdWsLds)));
      ws_attr = true;
    }
    side_take_signal(st, &b.sig, &b.sigval);           // (every check is behind us: the launch happens)
    hipLaunchKernelGGL(mlp_bwd_ws_kernel, dim3(ps_cdiv(a.M, MBM)), dim3(WS_THREADS), sizeof(MlpBwdWsLds), st, b);
    PS_LAUNCH_CHECK();
    return PS_OK;
  }
  side_take_signal(st, &b.sig, &b.sigval);
  hipLaunchKernelGGL(mlp_bwd_fused_kernel, dim3(ps_cdiv(a.M, MBM)), dim3(256), lds, st, b);
  PS_LAUNCH_CHECK();
  return PS_OK;
}
