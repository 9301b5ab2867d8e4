// mlp_fused.hip — the per-replica tail of the last encoder layer as ONE kernel, forward and backward (gfx950, d = 128).
//
// After attention, every encoder replica row m (B*R of them, R = K+1 when dropout is drawn) runs
//   y1  = dropout(ctx . Wo^T + bo) + x[b, qpos]                 (neural.py:228-231, transformer.py:56)
//   ln1 = LayerNorm_ff(y1) ; a1 = ln1 . W1^T + b1 ; h1 = dropout(gelu(a1))
//   y2  = dropout(h1 . W2^T + b2) + y1                           (neural.py:30-33)
//   enc = LayerNorm_final(y2)                                    (transformer.py:86)
// A workgroup (8 waves, one per CU) owns 32 replica rows for the whole chain.
//
// Round 3 form — TRANSPOSED, REGISTER-CHAINED bf16x3 products.  Every product is computed as  D^T = W . X^T : the WEIGHTS are
// the MFMA's A operand (32 output features per v_mfma_f32_32x32x16_bf16), the 32 replica rows its B operand / N dimension.
// Consequences (what the round-2 kernels, which staged 64-deep weight slabs through an LDS ring behind one barrier per slab,
// spent their time on):
//   * a weight fragment is consumed by exactly ONE wave, so it never needs LDS: the forward's embed launch re-splits the
//     fp32 weights into three bf16 planes IN FRAGMENT ORDER (WSplit), every wave streams its own contiguous run of 1-KiB
//     fragments global -> registers, fully coalesced, prefetched three product steps ahead.  No slab ring, no slab barriers;
//   * an accumulator lane holds 16 output FEATURES of ONE replica row — exactly the shape of the next product's B fragment
//     (k = those features), so  a1 -> gelu -> dropout -> h1  stays in registers and feeds  h1 . W2^T  directly (the feature
//     order inside a 32-block is a fixed permutation both sides of the chain agree on): no accumulator dumps, no LDS round
//     trip, and the 16 features of a lane are CONSECUTIVE (one 64-byte run of a1 / h1 / da1 per lane, one Philox call per 8);
//   * wave w owns feature blocks {NBW*w .. NBW*w+NBW-1} of the hidden layer end to end (W1 rows AND the matching W2 columns):
//     between the two LayerNorm stages the 8 waves run without any barrier and drift apart, so one wave's GELU / Philox /
//     split instructions overlap the other wave's MFMAs on the same SIMD (bf16 MFMAs, unlike the fp32 ones, co-issue);
//   * only the two LayerNorm stages need all 128 features of a row: the waves' partial [128 x 32] tiles meet in LDS there.
// Products are exact-fp32-grade bf16x3: x = hi + mid + lo (3 x 8 = 24 mantissa bits), six bf16 MFMAs per 16-deep step
// (hl, lh, mm, hm, mh, hh; fp32 accumulation): max error / sum|a b| 1.1e-7 against fp64, the fp32 MFMA's own 1.13e-7.
// Dropout masks of the three sites are the 16-bit column-shared Philox form (common.h): a lane's 8 consecutive features /
// columns share one call.  Everything the backward needs (y1, ln1, a1, h1, y2, LN statistics) is still written once.
#include "rowwise.h"
#include "x3frag.h"
#include <stdlib.h>
#ifndef TRY
#define TRY(x) do { int _rc = (x); if (_rc) return _rc; } while (0)
#endif

bool ps_fusion_enabled() {
  static const bool on = ps_env_int("PS_NO_FUSE", 0) == 0;
  return on;
}

#define MD 128            // model width this kernel is specialised for
#define MBM 32            // replica rows per workgroup
#define MT_THREADS 512
#define MLP_PF_DEFAULT 3   // weight fragments in flight per wave (round 5: 4 / 5 / 6 measured, no change: profiles/r05_mlp_notes.md)
#define PLD 132           // row stride (floats) of the partial tiles in LDS: [m][n], 16-byte aligned rows, +16 B per row

// Launder a value through an empty asm: stops LLVM from hoisting per-row store addresses out of unrolled code.
__device__ inline int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

struct MlpTLds {
  uint16_t Xa[3][MBM * MD];     // B-operand planes [m][k] (16-byte chunks XOR-swizzled by row): ctx then ln1 / do2 then dout
  float Ps[8][MBM * PLD];       // partial tiles [slot][m][n]: the two k-halves of the Wo product, later the 8 waves' partials
  float vec[6][MD];             // bias / LayerNorm vectors of the two LayerNorm stages
  float red[16];
};
static_assert(sizeof(MlpTLds) <= 160 * 1024, "fused MLP: LDS budget");

// sum over the 16 lanes of a DPP row, returned to every lane of the row (rotations: 4 adds)
__device__ __forceinline__ float row16_sum(float v) {
#define PS_ROR_ADD(n) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (n), 0xf, 0xf, false))
  PS_ROR_ADD(8); PS_ROR_ADD(4); PS_ROR_ADD(2); PS_ROR_ADD(1);
#undef PS_ROR_ADD
  return v;
}
// accumulator tile -> partial slot: lane (m = l31, h) holds features 32 nb + 16 h + r of row m
__device__ __forceinline__ void dump_acc(float* slot, const f32x16& v, int l31, int h, int nb) {
  float* p = slot + l31 * PLD + 32 * nb + 16 * h;
#pragma unroll
  for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(p + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// Row-per-lane register tiles <-> global memory through a wave-private LDS tile.  In the products a lane owns 16 consecutive
// features (64 B) of ONE replica row, so a `global_store_dwordx4` straight from those registers writes 64 separate 16-byte
// pieces in 32 rows, 2 KB apart — 64 partial-line requests per instruction, and the chain's a1 / h1 stores were a sixth of the
// forward kernel (tools/mlp_stamps.py, PS_MLP_DIAG=3: 57.0k -> 47.4k cycles without them).  Through the tile an instruction
// covers 8 rows x one full 128-byte line (8 lanes x 16 B per row).  The tile is the wave's own partial slot, idle during the
// chain; LDS executes a wave's instructions in order, so no barrier is needed.  Row stride 36 floats: the row-per-lane side is
// conflict-free, the line-per-8-lanes side pays one extra cycle per instruction.
#define TLD 36
__device__ __forceinline__ void tile_put(float* tile, const float (&v)[16], int l31, int h) {
  float* p = tile + l31 * TLD + 16 * h;
#pragma unroll
  for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(p + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}
__device__ __forceinline__ void tile_get(const float* tile, float (&v)[16], int l31, int h) {
  const float* p = tile + l31 * TLD + 16 * h;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 t = *reinterpret_cast<const float4*>(p + 4 * q);
    v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
  }
}
// tile -> rows m0 .. m0+31 of g (row stride ldg floats, the block's 32 features at g): four coalesced instructions
__device__ __forceinline__ void tile_store(const float* tile, float* g, int ldg, int lane, int rows_left) {
  const int rr = lane >> 3, cc = 4 * (lane & 7);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * i + rr;
    const float4 t = *reinterpret_cast<const float4*>(tile + row * TLD + cc);
    if (row < rows_left) *reinterpret_cast<float4*>(g + (size_t)row * ldg + cc) = t;
  }
}
// the same rows of g, requested coalesced into registers (tile_load_put places them once they have arrived)
__device__ __forceinline__ void tile_load(float4 (&t)[4], const float* g, int ldg, int lane, int rows_left) {
  const int rr = lane >> 3, cc = 4 * (lane & 7);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * i + rr;
    t[i] = *reinterpret_cast<const float4*>(g + (size_t)(row < rows_left ? row : 0) * ldg + cc);
    if (row >= rows_left) t[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__device__ __forceinline__ void tile_load_put(float* tile, const float4 (&t)[4], int lane) {
  const int rr = lane >> 3, cc = 4 * (lane & 7);
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(tile + (8 * i + rr) * TLD + cc) = t[i];
}

// diagnostic timeline (tools/mlp_stamps.py): wave w of workgroup 0 stores s_memtime into stamp[16 * w + slot]
static unsigned long long* g_mlp_stamp = nullptr;
extern "C" void ps_debug_set_stamp_buffer(void* p) { g_mlp_stamp = (unsigned long long*)p; }
unsigned long long* ps_debug_stamp_ptr() { return g_mlp_stamp; }
#if PS_DIAG_ON      // in-kernel phase stamps: diagnostic build only
#define MLP_STAMP(slot)                                                                              \
  do {                                                                                               \
    if (a.stamp && blockIdx.x == 0 && lane == 0) {                                                   \
      unsigned long long t_;                                                                         \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
      a.stamp[16 * wave + (slot)] = t_;                                                              \
    }                                                                                                \
  } while (0)
#else
#define MLP_STAMP(slot) do { } while (0)
#endif

// ====================================================================== forward
// Lane roles.  Products: lane (l31, h) = replica row l31 of the workgroup, half h; accumulator register r = logical feature
// 16 h + r of the 32-feature block.  LayerNorm stages: lane = (row 4*wave + (lane >> 4), columns 8c .. 8c+7 with c = lane & 15).
// DIAG (timing experiments only, results are wrong): 1 = the low weight plane is not loaded, 2 = no weight loads in the main
// phase, 3 = a1 / h1 are not stored
template <int NBW, int PF, int DIAG = 0>       // hidden-layer feature blocks per wave: F = 256 * NBW; PF = weight fragments in flight per wave
__global__ __launch_bounds__(MT_THREADS, 2) void mlp_fwd_t_kernel(const MlpFwdArgs a) {
  extern __shared__ float lds_raw[];
  MlpTLds& L = *reinterpret_cast<MlpTLds*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * MBM, M = a.M, F = a.F;
  const int mrow = 4 * wave + (lane >> 4), c8 = 8 * (lane & 15);       // LayerNorm-stage role
  const int mg = m0 + mrow;
  const uint32_t step_ctx = drop_step(a.drop_ctx), step_ff1 = drop_step(a.drop_ff1), step_ff2 = drop_step(a.drop_ff2);
  MLP_STAMP(0);

  // ---- prologue: everything with a global round trip is requested now
  // (1) the wave's four Wo fragments: output block nb = wave & 3, k half kh = wave >> 2
  const int nbo = wave & 3, kh = wave >> 2;
  const uint16_t* wo_stream = a.x3.fwd_wo + (size_t)(nbo * 8 + 4 * kh) * 1536;
  uint4 wof[4][3];
#pragma unroll
  for (int t = 0; t < 4; ++t) load_frag(wof[t], wo_stream, t, lane);
  // (2) the six small vectors of the LayerNorm stages -> LDS
  if (tid < 192) {
    const int which = tid >> 5, q = tid & 31;
    const float* src = which == 0 ? a.bo : (which == 1 ? a.g1 : (which == 2 ? a.be1 : (which == 3 ? a.b2 : (which == 4 ? a.gf : a.bef))));
    *reinterpret_cast<float4*>(&L.vec[which][4 * q]) = *reinterpret_cast<const float4*>(src + 4 * q);
  }
  // (3) ctx tile -> Xa planes: thread = (row, one 8-element chunk)
  {
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
    put8(L.Xa, row, 8 * kc, v);
  }
  // (4) the residual x[b, qpos] of this lane's LayerNorm elements (kept through both stages as y1)
  float y1r[8];
  {
    const float* src = a.xin + ((size_t)((mg < M ? mg : 0) / a.fan) * a.S + a.qpos) * MD + c8;
    const float4 v0 = mg < M ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 v1 = mg < M ? *reinterpret_cast<const float4*>(src + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    y1r[0] = v0.x; y1r[1] = v0.y; y1r[2] = v0.z; y1r[3] = v0.w; y1r[4] = v1.x; y1r[5] = v1.y; y1r[6] = v1.z; y1r[7] = v1.w;
  }
  // (5) the wave's first FF fragments (its stream: per feature block 8 W1 steps then 8 W2 steps)
  const uint16_t* ff_stream = a.x3.fwd_ff + (size_t)(wave * NBW) * 16 * 1536;
  uint4 ring[PF][3];
#pragma unroll
  for (int s = 0; s < PF; ++s) load_frag(ring[s], ff_stream, s, lane);
  __syncthreads();                                                     // P: ctx planes + vectors in LDS
  MLP_STAMP(1);

  // ---- Wo: this wave's [32 features x 32 rows] partial over its k half
  {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uint4 b[3];
      read_b(b, L.Xa, l31, 16 * (4 * kh + t) + 8 * h);
      x3_mma(acc, wof[t], b);
    }
    dump_acc(L.Ps[kh], acc, l31, h, nbo);
  }
  MLP_STAMP(2);
  __syncthreads();                                                     // A: Wo partials in Ps[0..1]; ctx planes consumed
  MLP_STAMP(3);

  // one LayerNorm stage, all 8 waves: v = dropout(sum of partial tiles + bias) + residual ; LayerNorm
  auto ln_stage = [&](const int nslots, const int vb, const DropSpec& drop, const uint32_t dstep, const int vg, const int ve,
                      const bool to_xa, float (&res)[8], float (&o)[8], float& mean, float& rstd) {
    float v[8];
    {
      const float4 b0 = *reinterpret_cast<const float4*>(&L.vec[vb][c8]), b1 = *reinterpret_cast<const float4*>(&L.vec[vb][c8 + 4]);
      v[0] = b0.x; v[1] = b0.y; v[2] = b0.z; v[3] = b0.w; v[4] = b1.x; v[5] = b1.y; v[6] = b1.z; v[7] = b1.w;
    }
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nslots; ++s) {                                 // fixed slot order
      const float* p = &L.Ps[s][mrow * PLD + c8];
      const float4 p0 = *reinterpret_cast<const float4*>(p), p1 = *reinterpret_cast<const float4*>(p + 4);
      s8[0] += p0.x; s8[1] += p0.y; s8[2] += p0.z; s8[3] += p0.w; s8[4] += p1.x; s8[5] += p1.y; s8[6] += p1.z; s8[7] += p1.w;
    }
    Philox4 rnd = {0u, 0u, 0u, 0u};
    if (drop.thr) rnd = drop_call16(drop, (uint32_t)mg, (uint32_t)(lane & 15), dstep);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x = s8[j] + v[j];
      if (drop.thr) x *= drop_half(drop, rnd, j);
      x += res[j];
      res[j] = x;
      sum += x;
    }
    mean = row16_sum(sum) * (1.f / MD);
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float dlt = res[j] - mean; sq += dlt * dlt; }
    rstd = 1.f / sqrtf(row16_sum(sq) * (1.f / MD) + 1e-6f);
    const float4 g0 = *reinterpret_cast<const float4*>(&L.vec[vg][c8]), g1 = *reinterpret_cast<const float4*>(&L.vec[vg][c8 + 4]);
    const float4 e0 = *reinterpret_cast<const float4*>(&L.vec[ve][c8]), e1 = *reinterpret_cast<const float4*>(&L.vec[ve][c8 + 4]);
    const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, ee[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (res[j] - mean) * rstd * gg[j] + ee[j];
    if (to_xa) put8(L.Xa, mrow, c8, o);
  };
  auto ln_store = [&](float* out_pre, float* out_ln, float* stats, const float (&res)[8], const float (&o)[8], float mean, float rstd) {
    if (mg < M) {
      float* p = out_pre + (size_t)mg * MD + c8;
      float* q = out_ln + (size_t)mg * MD + c8;
      *reinterpret_cast<float4*>(p) = make_float4(res[0], res[1], res[2], res[3]);
      *reinterpret_cast<float4*>(p + 4) = make_float4(res[4], res[5], res[6], res[7]);
      *reinterpret_cast<float4*>(q) = make_float4(o[0], o[1], o[2], o[3]);
      *reinterpret_cast<float4*>(q + 4) = make_float4(o[4], o[5], o[6], o[7]);
      if ((lane & 15) == 0) *reinterpret_cast<float2*>(stats + 2 * (size_t)mg) = make_float2(mean, rstd);
    }
  };

  {   // y1 = dropout(ctx.Wo^T + bo) + x ; ln1 = LayerNorm_ff(y1) -> Xa planes
    float o[8], mean, rstd;
    ln_stage(2, 0, a.drop_ctx, step_ctx, 1, 2, true, y1r, o, mean, rstd);
    ln_store(a.y1, a.ln1, a.st1, y1r, o, mean, rstd);
  }
  MLP_STAMP(4);
  __syncthreads();                                                     // B: ln1 planes in Xa
  MLP_STAMP(5);

  // ---- the feed-forward chain of this wave's feature blocks: no barrier until every wave is through
  f32x16 acc2[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[nb][r] = 0.f;
  const int mrow_p = m0 + l31;                                         // this lane's replica row in the products
  constexpr int NS = 16 * NBW;
#pragma unroll
  for (int bi = 0; bi < NBW; ++bi) {
    const int fb = wave * NBW + bi, f0 = 32 * fb + 16 * h;             // this lane's 16 consecutive hidden features
    float bias[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 bq = *reinterpret_cast<const float4*>(a.b1 + f0 + 4 * q);
      bias[4 * q] = bq.x; bias[4 * q + 1] = bq.y; bias[4 * q + 2] = bq.z; bias[4 * q + 3] = bq.w;
    }
    f32x16 acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
    uint4 bq[2][3];                                                    // the B operand of step t + 1 is read under step t's MFMAs
    read_b(bq[0], L.Xa, l31, 8 * h);
    // product steps one wave-priority level above epilogues: the SIMD's other wave is usually in the other kind of phase, and an MFMA
    // that waits behind its vector instructions leaves the pipe idle (round 5: 29.0 -> 28.5 us forward, the same backward; four
    // alternating pairs).  And one more level for waves 4-7 in the first feature block, for waves 0-3 in the second: at equal priority
    // the older wave of a SIMD wins every tie and waves 0-3 reached the chain's end 7-10 k cycles before their partners
    // (profiles/r05_mlp_stamps.txt); another 0.3 us.
    { if ((wave >= 4) == (bi == 0)) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
#pragma unroll
    for (int t = 0; t < 8; ++t) {                                      // a1^T block = W1[block rows] . ln1^T
      const int s = 16 * bi + t;
      if (t + 1 < 8) read_b(bq[(t + 1) & 1], L.Xa, l31, 16 * (t + 1) + 8 * h);
      __builtin_amdgcn_sched_barrier(0);       // (the reads of step t + 1 go out BEFORE this step's MFMAs, not behind them)
      x3_mma(acc1, ring[s % PF], bq[t & 1]);
      if (DIAG == 1) load_frag2(ring[s % PF], ff_stream, s + PF < NS ? s + PF : NS - 1, lane);
      else if (DIAG != 2) load_frag(ring[s % PF], ff_stream, s + PF < NS ? s + PF : NS - 1, lane);
      __builtin_amdgcn_sched_barrier(0);       // keep the step's order: operand reads, six MFMAs, the refill PF steps ahead (see load_frag)
    }
    { if ((wave >= 4) == (bi == 0)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    MLP_STAMP(6 + 3 * bi);
    // epilogue in registers: bias, GELU, dropout; a1 / h1 leave as one 64-byte run per lane; h1 becomes the next B operand
    float hv[16];
    float* tile = &L.Ps[wave][0];                                      // the wave's own partial slot: idle until the dumps below
    {
      float av[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc1[r] += bias[r]; av[r] = acc1[r]; }
      if (DIAG != 3) tile_put(tile, av, l31, h);
    }
    Philox4 r0 = {0u, 0u, 0u, 0u}, r1 = {0u, 0u, 0u, 0u};
    if (a.drop_ff1.thr) {
      r0 = drop_call16(a.drop_ff1, (uint32_t)mrow_p, (uint32_t)(f0 >> 3), step_ff1);
      r1 = drop_call16(a.drop_ff1, (uint32_t)mrow_p, (uint32_t)(f0 >> 3) + 1u, step_ff1);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float g = gelu_tanh_f(acc1[r]);
      if (a.drop_ff1.thr) g *= drop_half(a.drop_ff1, r < 8 ? r0 : r1, r & 7);
      hv[r] = g;
    }
    if (DIAG != 3) {
      tile_put(tile + MBM * TLD, hv, l31, h);
      tile_store(tile, a.a1 + (size_t)m0 * F + 32 * fb, F, lane, M - m0);
      tile_store(tile + MBM * TLD, a.h1 + (size_t)m0 * F + 32 * fb, F, lane, M - m0);
    }
    uint4 hf[2][3];
    {
      const float lo8[8] = {hv[0], hv[1], hv[2], hv[3], hv[4], hv[5], hv[6], hv[7]};
      const float hi8[8] = {hv[8], hv[9], hv[10], hv[11], hv[12], hv[13], hv[14], hv[15]};
      split8(lo8, hf[0]);
      split8(hi8, hf[1]);
    }
    MLP_STAMP(7 + 3 * bi);
    { if ((wave >= 4) == (bi == 0)) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
#pragma unroll
    for (int u = 0; u < 8; ++u) {                                      // y2^T += W2[:, block] . h1^T block   (k step u >> 2, rows 32 (u & 3) ..)
      const int s = 16 * bi + 8 + u;
      x3_mma(acc2[u & 3], ring[s % PF], hf[u >> 2]);
      if (DIAG == 1) load_frag2(ring[s % PF], ff_stream, s + PF < NS ? s + PF : NS - 1, lane);
      else if (DIAG != 2) load_frag(ring[s % PF], ff_stream, s + PF < NS ? s + PF : NS - 1, lane);
      __builtin_amdgcn_sched_barrier(0);       // keep the step's order: operand reads, six MFMAs, the refill PF steps ahead (see load_frag)
    }
  }
  __builtin_amdgcn_s_setprio(0);
  MLP_STAMP(12);
  // folded scoring: the item row this lane's enc elements will be dotted with (requested under the dumps and the barrier)
  float itr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float ibias = 0.f;
  if (a.fold_score && mg < M) {
    const ScoreArgs& S = a.sc;
    const int K1 = S.K + 1;
    const int b = fdiv(mg, S.fK1), j = mg - b * K1;
    int64_t idx = j == 0 ? S.target[b] : S.neg_items[(size_t)b * S.K + j - 1];
    idx = idx < 0 ? S.P : (idx > S.P ? S.P : idx);
    const float* row = S.product_emb + (size_t)idx * MD + c8;
    const float4 v0 = *reinterpret_cast<const float4*>(row), v1 = *reinterpret_cast<const float4*>(row + 4);
    itr[0] = v0.x; itr[1] = v0.y; itr[2] = v0.z; itr[3] = v0.w; itr[4] = v1.x; itr[5] = v1.y; itr[6] = v1.z; itr[7] = v1.w;
    if (S.bias_product) ibias = S.product_bias[idx];
  }
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) dump_acc(L.Ps[wave], acc2[nb], l31, h, nb);
  __syncthreads();                                                     // C: the 8 waves' y2 partials in Ps[0..7]
  MLP_STAMP(13);

  // y2 = dropout(h1.W2^T + b2) + y1 ; enc = LayerNorm_final(y2)
  float o[8], mean, rstd;
  ln_stage(8, 3, a.drop_ff2, step_ff2, 4, 5, false, y1r, o, mean, rstd);
  MLP_STAMP(14);
  if (!a.fold_score) { ln_store(a.y2, a.enc, a.stf, y1r, o, mean, rstd); return; }

  // ---- folded scoring: score, loss term, one loss partial per workgroup handed over by ONE returning 64-bit atomic (fixed
  // point: integer adds commute, so the total is exact and order-independent; the arrival count rides in bits 48+); the
  // workgroup that arrives last adds the word tasks' partials (left by the embed launch) and writes the loss.
  const ScoreArgs& S = a.sc;
  const int K1 = S.K + 1;
  float dot = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) dot += o[j] * itr[j];
  const float sc = row16_sum(dot) + ibias;
  const bool rowq = (lane & 15) == 0 && mg < M;
  const int bq = fdiv(rowq ? mg : 0, S.fK1), jq = (rowq ? mg : 0) - bq * K1;
  const float twq = jq == 0 ? -(S.pos_weight ? (float)S.K : 1.f) : 1.f;
  const float termq = fabsf(twq) * softplus_f(twq < 0.f ? -sc : sc);
  const float cps = wave_sum(rowq ? termq : 0.f);
  if (lane == 0) L.red[wave] = cps;
  __syncthreads();
  unsigned long long mine = 0ull, old = 0ull;
  const unsigned long long one = 1ull << 48, mask = one - 1ull;
  // Fixed-point scale: the sum of ALL partials must stay below bit 47 (a carry into bit 48 would corrupt the arrival count: the
  // last-arriver test would fire early or never).  2^20 up to 256 workgroups, one bit coarser per doubling of the
  // grid beyond that: a workgroup's partial has 2^19 = 524k of headroom at any grid (16k per replica row — a diverged run).
  // A partial beyond that, negative, NaN or Inf is handed over as 0 and POISONS the step's loss: a flag in the ticket's
  // second word, set by a RETURNING atomic whose value the arrival add depends on, so the flag is in place before the
  // arrival can be counted; the last arriver reads it and reports +inf.
  int shift = 20;
  for (unsigned g = 256u; g < gridDim.x && shift > 8; g <<= 1) --shift;
  const float fscale = (float)(1u << shift);
  if (tid == 0) {
    const float p = ((L.red[0] + L.red[1]) + (L.red[2] + L.red[3])) + ((L.red[4] + L.red[5]) + (L.red[6] + L.red[7]));
    unsigned long long* tk = reinterpret_cast<unsigned long long*>(S.ticket);
    long long fx = 0;
    if (p >= 0.f && p < 524288.f) fx = __float2ll_rn(p * fscale);
    else fx = (long long)(__hip_atomic_fetch_or(tk + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0ull);
    mine = (unsigned long long)fx | one;
    old = __hip_atomic_fetch_add(tk, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // the stage's stores are issued under the atomic's round trip
  if (rowq) { S.item_scores[mg] = sc; S.item_terms[mg] = termq; }
  ln_store(a.y2, a.enc, a.stf, y1r, o, mean, rstd);
  if (tid == 0) {
    float last = 0.f;
    if ((old >> 48) == gridDim.x - 1u) {
      last = 1.f;
      const unsigned long long poison =
          __hip_atomic_load(reinterpret_cast<unsigned long long*>(S.ticket) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      L.red[9] = poison ? __builtin_inff() : (float)((double)((old & mask) + (mine & mask)) / (double)fscale);
    }
    L.red[8] = last;
  }
  __syncthreads();
  MLP_STAMP(15);
  if (L.red[8] == 0.f) return;
  if (wave == 0) {      // last arriver: add the word tasks' partials in a fixed order => bitwise reproducible
    float il = 0.f;
    il = strided_sum_f32<8>(S.word_blk, S.word_nblk, lane, 64);      // (eight loads in flight: this wave ends the launch)
    il = wave_sum(il);
    if (lane == 0) {
      const float ps = L.red[9] / (float)S.B;
      il /= (float)S.B;
      S.loss3[0] = ps + il; S.loss3[1] = ps; S.loss3[2] = il;
      if (S.loss_acc) { S.loss_acc[0] += ps; S.loss_acc[1] += il; }
    }
  }
}

// the fused kernels take F = 256 * {1, 2, 4}; the planes live in the workspace (WSplit), re-split by the embed launch
bool mlp_x3_enabled(int F) { return ps_fusion_enabled() && (F == 256 || F == 512 || F == 1024); }
int64_t mlp_x3_floats(int d, int F) {      // forward + backward fragment streams: 2 x 3 planes x (d*d + 2*d*F) bf16,
  return ((int64_t)2 * 3 * ((int64_t)d * d + 2 * (int64_t)d * F) * 2 + (int64_t)2 * 3 * 2 * d * d * 2 + 3) / 4 + 16;   // + the K / V streams (WSplit::fwd_kv, bwd_kv)
}
bool mlp_fwd_can_fold_score(int M, int F, int d) {
  static const bool fold_on = ps_env_int("PS_NO_FOLD_SCORE", 0) == 0;
  // (round 3 stopped at 256 workgroups — all resident — and B >= 391 at K = 20 silently fell off the folded path.  Nothing in
  // the hand-off needs residency: every workgroup ADDS its fixed-point partial and its arrival into one 64-bit word and
  // leaves; whichever add returns count == grid - 1 finishes — no workgroup waits for another.  The arrival count has 16 bits.)
  return fold_on && d == MD && mlp_x3_enabled(F) && M > 0 && ps_cdiv(M, MBM) <= 65535;
}
bool mlp_fused_serves(int d, int F) { return d == MD && mlp_x3_enabled(F); }

template <class K>
static int set_lds_attr(K kernel, bool& done) {
  if (!done) {
    PS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MlpTLds)));
    done = true;
  }
  return PS_OK;
}

int launch_mlp_fwd_fused(const MlpFwdArgs& a, hipStream_t st) {
  PS_REQUIRE(mlp_x3_enabled(a.F) && a.M > 0, "fused mlp: F=%d M=%d (F must be 256, 512 or 1024)", a.F, a.M);
  PS_REQUIRE(a.x3.on && a.x3.fwd_wo && a.x3.fwd_ff, "fused mlp: the weight fragment streams are missing (WSplit)");
  PS_REQUIRE(!a.fold_score || mlp_fwd_can_fold_score(a.M, a.F, MD), "fused mlp: folded scoring needs <= 65535 workgroups");
  KTimeScope kt("mlp_fwd", st);
  MlpFwdArgs as = a;
  as.stamp = g_mlp_stamp;
  const dim3 grid(ps_cdiv(a.M, MBM)), block(MT_THREADS);
  static bool a1 = false, a2 = false, a4 = false;
#ifdef PS_DIAG                                                     // timing experiments (WRONG results): diagnostic build only
  static bool ad1 = false, ad2 = false, ad3 = false;
  static const int diag = ps_diag_int("PS_MLP_DIAG", 0);
  if (a.F == 512 && diag == 1) { TRY(set_lds_attr(mlp_fwd_t_kernel<2, 3, 1>, ad1)); PS_KLAUNCH((mlp_fwd_t_kernel<2, 3, 1>), grid, block, sizeof(MlpTLds), st, as); }
  else if (a.F == 512 && diag == 2) { TRY(set_lds_attr(mlp_fwd_t_kernel<2, 3, 2>, ad2)); PS_KLAUNCH((mlp_fwd_t_kernel<2, 3, 2>), grid, block, sizeof(MlpTLds), st, as); }
  else if (a.F == 512 && diag == 3) { TRY(set_lds_attr(mlp_fwd_t_kernel<2, 3, 3>, ad3)); PS_KLAUNCH((mlp_fwd_t_kernel<2, 3, 3>), grid, block, sizeof(MlpTLds), st, as); }
  else
#endif
  if (a.F == 256) { TRY(set_lds_attr(mlp_fwd_t_kernel<1, 3>, a1)); PS_KLAUNCH((mlp_fwd_t_kernel<1, 3>), grid, block, sizeof(MlpTLds), st, as); }
  else if (a.F == 512) { TRY(set_lds_attr(mlp_fwd_t_kernel<2, 3>, a2)); PS_KLAUNCH((mlp_fwd_t_kernel<2, 3>), grid, block, sizeof(MlpTLds), st, as); }
  else { TRY(set_lds_attr(mlp_fwd_t_kernel<4, 3>, a4)); PS_KLAUNCH((mlp_fwd_t_kernel<4, 3>), grid, block, sizeof(MlpTLds), st, as); }
  PS_LAUNCH_CHECK();
  return PS_OK;
}

// ====================================================================== backward
// The same structure backwards, every product again  grad^T = W^T-fragment . g^T  with the weights as the A operand:
//   final-LN backward (all waves) -> do2 planes
//   per feature block of a wave:  d h1^T = W2^T[block] . do2^T  (8 steps) -> * gelu'(a1) * dropout = d a1^T (registers; the
//       block's b1 column sums by a DPP reduction over the 32 rows) -> d ln1^T partial += W1^T[:, block] . d a1^T (8 steps)
//   the 8 waves' d ln1 partials meet in LDS -> FF-LN backward (+ d y2 residual) -> dy1, dout = dropout(dy1) planes
//   d ctx^T = Wo^T . dout^T  (4 output blocks x 2 reduction halves over the 8 waves, halves met in LDS)
// Writes only what the weight gradients and the fan-in residual read (do2, da1, dy1, dout, dctx) and parks the seven
// bias / gamma / beta column sums per workgroup (fixed-order sums: bitwise reproducible).
template <int NBW, int PF>
__global__ __launch_bounds__(MT_THREADS, 2) void mlp_bwd_t_kernel(const MlpBwdArgs a) {
  fork_signal(a.sig, a.sigval);
  extern __shared__ float lds_raw[];
  MlpTLds& L = *reinterpret_cast<MlpTLds*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * MBM, M = a.M, F = a.F;
  const int mrow = 4 * wave + (lane >> 4), c8 = 8 * (lane & 15);
  const int mg = m0 + mrow;
  const bool ok = mg < M;
  const uint32_t step_ctx = drop_step(a.drop_ctx), step_ff1 = drop_step(a.drop_ff1), step_ff2 = drop_step(a.drop_ff2);
  MLP_STAMP(0);
  float (*Csum)[MBM][MD] = reinterpret_cast<float (*)[MBM][MD]>(&L.Ps[2][0]);   // [3][32 row groups][128] column-sum scratch (slots 2-4)
  static_assert(3 * MBM * MD <= 3 * MBM * PLD, "column-sum scratch must fit three partial slots");

  // ---- prologue requests
  const uint16_t* ff_stream = a.x3.bwd_ff + (size_t)(wave * NBW) * 16 * 1536;
  uint4 ring[PF][3];
#pragma unroll
  for (int s = 0; s < PF; ++s) load_frag(ring[s], ff_stream, s, lane);
  if (tid < 64) {
    const int which = tid >> 5, q = tid & 31;
    *reinterpret_cast<float4*>(&L.vec[which][4 * q]) = *reinterpret_cast<const float4*>((which == 0 ? a.gf : a.g1) + 4 * q);
  }
  // LayerNorm inputs of this lane's elements: y2 (final LN) now, y1 (FF LN) too — both are needed across the main phase
  float x2[8], st2[2] = {0.f, 0.f};
  {
    const float* src = a.y2 + (size_t)(ok ? mg : 0) * MD + c8;
    const float4 v0 = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 v1 = ok ? *reinterpret_cast<const float4*>(src + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    x2[0] = v0.x; x2[1] = v0.y; x2[2] = v0.z; x2[3] = v0.w; x2[4] = v1.x; x2[5] = v1.y; x2[6] = v1.z; x2[7] = v1.w;
    if (ok) { const float2 s2 = *reinterpret_cast<const float2*>(a.stf + 2 * (size_t)mg); st2[0] = s2.x; st2[1] = s2.y; }
  }
  // d enc of this lane's elements
  float dy[8];
  if (a.item_scores) {
    // straight from the score: row m = (b, j) of enc was dotted with item row idx(b, j), so d enc[m] = loss'(score[m]) *
    // product_emb[idx] (item_transformer.py:485,493-494,500-514); the score backward then only scatters into the tables
    const float invB = a.scale * (a.scale_dev ? *a.scale_dev : 1.f) / (float)a.B;
    const float wpos = a.pos_weight ? (float)a.K : 1.f;
    const int K1 = a.K + 1;
    const int mm = ok ? mg : 0;
    const int b = mm / K1, j = mm - b * K1;
    int64_t idx = j == 0 ? a.target[b] : a.neg_items[(size_t)b * a.K + j - 1];
    idx = idx < 0 ? a.P : (idx > a.P ? a.P : idx);
    const float sc = a.item_scores[mm];
    const float ds = ok ? (j == 0 ? wpos * (sigmoid_f(sc) - 1.f) : sigmoid_f(sc)) * invB : 0.f;
    const float* row = a.product_emb + (size_t)idx * MD + c8;
    const float4 v0 = *reinterpret_cast<const float4*>(row), v1 = *reinterpret_cast<const float4*>(row + 4);
    dy[0] = ds * v0.x; dy[1] = ds * v0.y; dy[2] = ds * v0.z; dy[3] = ds * v0.w;
    dy[4] = ds * v1.x; dy[5] = ds * v1.y; dy[6] = ds * v1.z; dy[7] = ds * v1.w;
  } else {
    const float* src = a.denc + (size_t)(ok ? mg : 0) * MD + c8;
    const float4 v0 = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 v1 = ok ? *reinterpret_cast<const float4*>(src + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    dy[0] = v0.x; dy[1] = v0.y; dy[2] = v0.z; dy[3] = v0.w; dy[4] = v1.x; dy[5] = v1.y; dy[6] = v1.z; dy[7] = v1.w;
  }
  __syncthreads();                                                     // P: gamma vectors in LDS
  MLP_STAMP(1);

  // LayerNorm backward of this lane's elements:  dx = rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat)) (+ res);
  // dropped = dx * mask -> Xa planes (+ global out_drop); column sums {dy*xhat, dy, dropped} -> Csum[.][row group][col]
  auto ln_bwd_stage = [&](const float (&dyv)[8], const float (&x)[8], const float mean, const float rstd, const int vg,
                          const bool has_res, const float (&res)[8], const DropSpec& drop, const uint32_t dstep,
                          float (&dxr)[8], float* out_dx, float* out_drop) {
    const float4 g0 = *reinterpret_cast<const float4*>(&L.vec[vg][c8]), g1 = *reinterpret_cast<const float4*>(&L.vec[vg][c8 + 4]);
    const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    float xh[8], dh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      xh[j] = (x[j] - mean) * rstd;
      dh[j] = dyv[j] * gg[j];
      s1 += dh[j]; s2 += dh[j] * xh[j];
    }
    s1 = row16_sum(s1) * (1.f / MD);
    s2 = row16_sum(s2) * (1.f / MD);
    Philox4 rnd = {0u, 0u, 0u, 0u};
    if (drop.thr) rnd = drop_call16(drop, (uint32_t)mg, (uint32_t)(lane & 15), dstep);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float dlt = rstd * (dh[j] - s1 - xh[j] * s2);
      if (has_res) dlt += res[j];
      if (!ok) dlt = 0.f;
      dxr[j] = dlt;
      v[j] = drop.thr ? dlt * drop_half(drop, rnd, j) : dlt;
    }
    put8(L.Xa, mrow, c8, v);
    if (ok) {
      if (out_dx) {
        float* p = out_dx + (size_t)mg * MD + c8;
        *reinterpret_cast<float4*>(p) = make_float4(dxr[0], dxr[1], dxr[2], dxr[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(dxr[4], dxr[5], dxr[6], dxr[7]);
      }
      if (out_drop) {
        float* p = out_drop + (size_t)mg * MD + c8;
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
    float* c0 = &Csum[0][mrow][c8];
    float* c1 = &Csum[1][mrow][c8];
    float* c2 = &Csum[2][mrow][c8];
    *reinterpret_cast<float4*>(c0) = make_float4(dyv[0] * xh[0], dyv[1] * xh[1], dyv[2] * xh[2], dyv[3] * xh[3]);
    *reinterpret_cast<float4*>(c0 + 4) = make_float4(dyv[4] * xh[4], dyv[5] * xh[5], dyv[6] * xh[6], dyv[7] * xh[7]);
    *reinterpret_cast<float4*>(c1) = make_float4(dyv[0], dyv[1], dyv[2], dyv[3]);
    *reinterpret_cast<float4*>(c1 + 4) = make_float4(dyv[4], dyv[5], dyv[6], dyv[7]);
    *reinterpret_cast<float4*>(c2) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(c2 + 4) = make_float4(v[4], v[5], v[6], v[7]);
  };
  // after a barrier: park the workgroup's three column sums part[blk][3][128], rows added in a fixed order
  auto park = [&](float* part) {
    if (tid < 3 * MD) {
      const int which = tid >> 7, colx = tid & 127;
      float s = 0.f;
#pragma unroll 8
      for (int r = 0; r < MBM; ++r) s += Csum[which][r][colx];
      part[((size_t)blockIdx.x * 3 + which) * MD + colx] = s;
    }
  };

  // ---- final LayerNorm backward (transformer.py:86) -> d y2 (kept: residual of the FF LayerNorm), do2 planes
  float dy2r[8];
  {
    const float zero[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dyz[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dyz[j] = ok ? dy[j] : 0.f;
    ln_bwd_stage(dyz, x2, st2[0], st2[1], 0, false, zero, a.drop_ff2, step_ff2, dy2r, nullptr, a.do2);
  }
  __syncthreads();                                                     // 1: do2 planes + column sums
  MLP_STAMP(2);
  park(a.part_f);
  __syncthreads();                                                     // 2: column-sum scratch (partial slots 2-4) free again
  MLP_STAMP(3);

  // ---- per feature block: d h1 -> d a1 (registers) -> d ln1 partial
  f32x16 acc2[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[nb][r] = 0.f;
  const int mrow_p = m0 + l31;
  const bool row_ok = mrow_p < M;
  constexpr int NS = 16 * NBW;
#pragma unroll
  for (int bi = 0; bi < NBW; ++bi) {
    const int fb = wave * NBW + bi, f0 = 32 * fb + 16 * h;
    float* tile = &L.Ps[wave][0];                                      // the wave's own partial slot (tile_put above): idle in the chain
    float4 a1g[4];                                                     // a1 of the block, requested coalesced; placed behind the steps
    tile_load(a1g, a.a1 + (size_t)m0 * F + 32 * fb, F, lane, M - m0);
    f32x16 acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
    uint4 bq[2][3];                                                    // the B operand of step t + 1 is read under step t's MFMAs
    read_b(bq[0], L.Xa, l31, 8 * h);
    { if ((wave >= 4) == (bi == 0)) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
#pragma unroll
    for (int t = 0; t < 8; ++t) {                                      // d h1^T block = W2^T[block rows] . do2^T
      const int s = 16 * bi + t;
      if (t + 1 < 8) read_b(bq[(t + 1) & 1], L.Xa, l31, 16 * (t + 1) + 8 * h);
      __builtin_amdgcn_sched_barrier(0);       // (the reads of step t + 1 go out BEFORE this step's MFMAs, not behind them)
      x3_mma(acc1, ring[s % PF], bq[t & 1]);
      load_frag(ring[s % PF], ff_stream, s + PF < NS ? s + PF : NS - 1, lane);
      __builtin_amdgcn_sched_barrier(0);       // keep the step's order: operand reads, six MFMAs, the refill PF steps ahead (see load_frag)
    }
    { if ((wave >= 4) == (bi == 0)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    MLP_STAMP(4 + 2 * bi);
    Philox4 r0 = {0u, 0u, 0u, 0u}, r1 = {0u, 0u, 0u, 0u};
    if (a.drop_ff1.thr) {
      r0 = drop_call16(a.drop_ff1, (uint32_t)mrow_p, (uint32_t)(f0 >> 3), step_ff1);
      r1 = drop_call16(a.drop_ff1, (uint32_t)mrow_p, (uint32_t)(f0 >> 3) + 1u, step_ff1);
    }
    float a1v[16];
    tile_load_put(tile, a1g, lane);
    tile_get(tile, a1v, l31, h);
    float dv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float g = acc1[r] * gelu_tanh_grad(a1v[r]);
      if (a.drop_ff1.thr) g *= drop_half(a.drop_ff1, r < 8 ? r0 : r1, r & 7);
      dv[r] = row_ok ? g : 0.f;
    }
    tile_put(tile + MBM * TLD, dv, l31, h);
    tile_store(tile + MBM * TLD, a.da1 + (size_t)m0 * F + 32 * fb, F, lane, M - m0);
    // b1 column sums of this block over the workgroup's 32 rows: DPP scan over the 32 lanes of each half; lanes 31 and 63
    // hold the totals of features f0 .. f0+15 of their half and park them (one row per workgroup: part_b1[wg][slot 0][F])
    {
      float cs[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) cs[r] = half_sum_last(dv[r]);
      if (l31 == 31) {
        float* pb = a.part_b1 + ((size_t)blockIdx.x * 3) * F + f0;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(pb + 4 * q) = make_float4(cs[4 * q], cs[4 * q + 1], cs[4 * q + 2], cs[4 * q + 3]);
      }
    }
    MLP_STAMP(5 + 2 * bi);
    { if ((wave >= 4) == (bi == 0)) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
    uint4 df[2][3];
    {
      const float lo8[8] = {dv[0], dv[1], dv[2], dv[3], dv[4], dv[5], dv[6], dv[7]};
      const float hi8[8] = {dv[8], dv[9], dv[10], dv[11], dv[12], dv[13], dv[14], dv[15]};
      split8(lo8, df[0]);
      split8(hi8, df[1]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {                                      // d ln1^T += W1^T[:, block] . d a1^T block
      const int s = 16 * bi + 8 + u;
      x3_mma(acc2[u & 3], ring[s % PF], df[u >> 2]);
      load_frag(ring[s % PF], ff_stream, s + PF < NS ? s + PF : NS - 1, lane);
      __builtin_amdgcn_sched_barrier(0);       // keep the step's order: operand reads, six MFMAs, the refill PF steps ahead (see load_frag)
    }
  }
  // FF LayerNorm inputs and the Wo^T fragments, requested under the dumps and the barrier
  __builtin_amdgcn_s_setprio(0);
  MLP_STAMP(8);
  float x1[8], st1v[2] = {0.f, 0.f};
  {
    const float* src = a.y1 + (size_t)(ok ? mg : 0) * MD + c8;
    const float4 v0 = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 v1 = ok ? *reinterpret_cast<const float4*>(src + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    x1[0] = v0.x; x1[1] = v0.y; x1[2] = v0.z; x1[3] = v0.w; x1[4] = v1.x; x1[5] = v1.y; x1[6] = v1.z; x1[7] = v1.w;
    if (ok) { const float2 s2 = *reinterpret_cast<const float2*>(a.st1 + 2 * (size_t)mg); st1v[0] = s2.x; st1v[1] = s2.y; }
  }
  const int kbo = wave & 3, nh = wave >> 2;                            // d ctx: output block kbo, reduction half nh
  const uint16_t* wo_stream = a.x3.bwd_wo + (size_t)(kbo * 8 + 4 * nh) * 1536;
  uint4 wof[4][3];
#pragma unroll
  for (int t = 0; t < 4; ++t) load_frag(wof[t], wo_stream, t, lane);
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) dump_acc(L.Ps[wave], acc2[nb], l31, h, nb);
  __syncthreads();                                                     // 3: the 8 waves' d ln1 partials in Ps[0..7]
  MLP_STAMP(9);

  // ---- FF LayerNorm backward (+ d y2 residual) -> dy1 ; dout = dropout(dy1) -> planes
  float dl[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < 8; ++s) {                                        // fixed slot order
    const float* p = &L.Ps[s][mrow * PLD + c8];
    const float4 p0 = *reinterpret_cast<const float4*>(p), p1 = *reinterpret_cast<const float4*>(p + 4);
    dl[0] += p0.x; dl[1] += p0.y; dl[2] += p0.z; dl[3] += p0.w; dl[4] += p1.x; dl[5] += p1.y; dl[6] += p1.z; dl[7] += p1.w;
  }
  __syncthreads();                                                     // 4: partial slots consumed (the column sums reuse slots 2-4)
  MLP_STAMP(10);
  {
    float dy1r[8];
    const bool same = a.dout == a.dy1;
    ln_bwd_stage(dl, x1, st1v[0], st1v[1], 1, true, dy2r, a.drop_ctx, step_ctx, dy1r, a.dy1, same ? nullptr : a.dout);
  }
  __syncthreads();                                                     // 5: dout planes + column sums
  MLP_STAMP(11);
  park(a.part_1);
  // ---- d ctx^T = Wo^T . dout^T : this wave's output block over its reduction half; halves meet in slots 0 / 1
  {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uint4 b[3];
      read_b(b, L.Xa, l31, 16 * (4 * nh + t) + 8 * h);
      x3_mma(acc, wof[t], b);
    }
    dump_acc(L.Ps[nh], acc, l31, h, kbo);
  }
  __syncthreads();                                                     // 6: both halves in Ps[0..1]
  MLP_STAMP(12);
  if (ok) {
    const float* p0 = &L.Ps[0][mrow * PLD + c8];
    const float* p1 = &L.Ps[1][mrow * PLD + c8];
    const float4 u0 = *reinterpret_cast<const float4*>(p0), u1 = *reinterpret_cast<const float4*>(p0 + 4);
    const float4 w0 = *reinterpret_cast<const float4*>(p1), w1 = *reinterpret_cast<const float4*>(p1 + 4);
    float* q = a.dctx + (size_t)mg * MD + c8;
    *reinterpret_cast<float4*>(q) = make_float4(u0.x + w0.x, u0.y + w0.y, u0.z + w0.z, u0.w + w0.w);
    *reinterpret_cast<float4*>(q + 4) = make_float4(u1.x + w1.x, u1.y + w1.y, u1.z + w1.z, u1.w + w1.w);
  }
  MLP_STAMP(13);
}

int mlp_bwd_fused_blocks(int M) { return ps_cdiv(M, MBM); }
// parked rows of the b1 column sums: one per workgroup
int mlp_bwd_b1_rows(int M, int F) { (void)F; return ps_cdiv(M, MBM); }

int launch_mlp_bwd_fused(const MlpBwdArgs& a, hipStream_t st) {
  PS_REQUIRE(mlp_x3_enabled(a.F) && a.M > 0, "fused mlp backward: F=%d M=%d (F must be 256, 512 or 1024)", a.F, a.M);
  PS_REQUIRE(a.part_f && a.part_1 && a.part_b1, "fused mlp backward: column sums must be parked");
  PS_REQUIRE(a.x3.on && a.x3.bwd_wo && a.x3.bwd_ff, "fused mlp backward: the weight fragment streams are missing (WSplit)");
  KTimeScope kt("mlp_bwd", st);
  MlpBwdArgs b = a;
  b.sig = nullptr; b.sigval = 0;
  b.stamp = g_mlp_stamp ? g_mlp_stamp + 128 : nullptr;
  const dim3 grid(ps_cdiv(a.M, MBM)), block(MT_THREADS);
  static bool a1 = false, a2 = false, a4 = false;
  if (a.F == 256) TRY(set_lds_attr(mlp_bwd_t_kernel<1, 3>, a1));
  else if (a.F == 512) TRY(set_lds_attr(mlp_bwd_t_kernel<2, 3>, a2));
  else TRY(set_lds_attr(mlp_bwd_t_kernel<4, 3>, a4));
  side_take_signal(st, &b.sig, &b.sigval);             // (every check is behind us: the launch happens)
  if (a.F == 256) hipLaunchKernelGGL((mlp_bwd_t_kernel<1, 3>), grid, block, sizeof(MlpTLds), st, b);
  else if (a.F == 512) hipLaunchKernelGGL((mlp_bwd_t_kernel<2, 3>), grid, block, sizeof(MlpTLds), st, b);
  else hipLaunchKernelGGL((mlp_bwd_t_kernel<4, 3>), grid, block, sizeof(MlpTLds), st, b);
  PS_LAUNCH_CHECK();
  return PS_OK;
}
