// gemm.hip — exact-fp32 MFMA GEMM with fused epilogues (gfx950).
//
// The only dense contractions of the hot path are the transformer's linears
// (reference models/neural.py:86-96 K/V/Q/final_linear, :20-21 w_1/w_2, and
// text_encoder.py:24 f_W) and their two backward products.  They are computed
// with v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate = a k-ordered fmaf chain,
// so results are exact fp32 like the reference's mm), tiled for 64-wide waves:
//
//   workgroup = 4 waves, tile 64x64 (each wave one 32x32 accumulator = 16 AGPRs);
//   the reduction advances in slabs of 32 (128 for launches of few workgroups: the hot path's K
//   is d = 128, so a whole projection is then ONE global->register->LDS round trip followed by
//   64 back-to-back MFMAs per wave); the next slabs are prefetched into registers while the
//   current one is multiplied — which only holds if nothing touches the loaded registers before
//   the LDS store (see tile_load: a zero-fill select, a scratch-resident segment table or a load
//   under a branch each made the compiler drain vmcnt BEFORE the MFMA block; fixing the three
//   took the step from 0.51 to 0.46 ms); operands sit in LDS as [k][row+1] so every MFMA operand
//   fetch is a conflict-free ds_read_b32 per lane.
//
// One kernel serves forward (A[m][k] . W[n][k]), input-grad (dY[m][n] . W[n][k'])
// and weight-grad (dY[m][n]^T . X[m][k'], split over the row reduction with fp32
// atomics) through (ta, tb) layouts.  Two epilogue instantiations keep the code in
// the instruction cache: PLAIN (bias, scale, store / += / atomic) and FULL (adds
// GELU/tanh or their derivatives, Philox dropout, residual direct / gathered from
// the layer input / fan-in summed over replicas, bias-grad column sums, 2nd output),
// whose body is a rolled loop over an LDS-staged tile so it exists once in the code.
//
// Measured alternatives (MI355X, tools/gemm_bench.py): a 128x128 tile with 2x2 accumulators per wave (half the LDS and
// L2 bytes per flop) is SLOWER on every hot shape — K is only 128-512 deep, so a workgroup's life is 4-16 slabs and the
// larger tile just has fewer workgroups to hide its fill/drain behind (78k x 256 x 128: 92 vs 80 us; 78k x 128 x 128:
// 66 vs 44 us; weight gradients over 78k rows: 65 vs 49 us).  128-deep slabs for every launch cost the review
// transformer 15 % (one workgroup per CU).
#include "common.h"
#include <stdlib.h>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 64
#define BN 64
#define LDT (BM + 1)

// TRANS==0: src[row][k] (k contiguous): one instruction = 8 rows x 128 B
// TRANS==1: src[k][row] (row contiguous): one instruction = 4 k-rows x 256 B

// BK = reduction depth of one slab (32 or 128); NLD = float4 loads per thread per operand per slab.
// The loaded registers must not be touched before the slab is stored to LDS, or the compiler waits for the loads
// BEFORE the MFMAs of the current slab and the prefetch overlaps nothing (it did: a v_cndmask zero-fill per load put
// s_waitcnt vmcnt(0) ahead of the MFMA block).  So rows are clamped instead of masked — rows past the edge are
// duplicates whose products land in output rows/columns the epilogue never stores — and only a ragged reduction
// tail (kmask: K not a multiple of the slab, block-uniform) takes the zero-filling path.
// pr0 / pr1 (TRANS == 0, row list; -1: none): the two physical rows this thread loads from, fetched once per workgroup.
// kmap (TRANS == 1, row list): LDS copy of the physical reduction rows of this workgroup's range, kmap[k - kbase].
template <int TRANS, int BK>
__device__ inline void tile_load(const float* __restrict__ src, int ld, int row0, int nrows, int k0, int kend,
                                 float4 (&reg)[BK / 16], int tid, bool kmask, const float* __restrict__ seg1 = nullptr,
                                 const float* __restrict__ seg2 = nullptr, int kseg = 0, int pr0 = -1, int pr1 = -1,
                                 const int* kmap = nullptr, int kbase = 0) {
  constexpr int NLD = BK / 16;
#pragma unroll
  for (int u = 0; u < NLD; ++u) {
    int r, k;
    if (TRANS == 0) {
      r = pr0 >= 0 ? ((u & 1) ? pr1 : pr0) : min(row0 + (tid >> 3) + 32 * (u & 1), nrows - 1);   // u is compile-time
      k = k0 + 32 * (u >> 1) + 4 * (tid & 7);
    } else {
      r = min(row0 + 4 * (tid & 15), nrows - 4);              // nrows % 4 == 0 (validated)
      k = k0 + (tid >> 4) + 16 * u;
    }
    const bool ok = !kmask || k < kend;
    int kl = ok ? k : k0;
    if (TRANS == 1 && kmap) kl = kmap[kl - kbase];
    const float* base = src;
    if (kseg > 0) {                            // K-concatenated operand (kseg > 0): pick the segment of this k.  Plain
      const int sg = (kl >= kseg) + (kl >= 2 * kseg);   // scalars, no table: a table indexed per lane lives in scratch,
      base = sg == 0 ? src : (sg == 1 ? seg1 : seg2);   // whose loads share vmcnt with the tile loads and serialise them
      kl -= sg * kseg;
    }
    const size_t off = TRANS == 0 ? (size_t)r * ld + kl : (size_t)kl * ld + r;
    reg[u] = *reinterpret_cast<const float4*>(base + off);    // raw: tile_store zero-fills a ragged tail
  }
}

template <int TRANS, int BK>
__device__ inline void tile_store(float (*T)[LDT], const float4 (&reg)[BK / 16], int tid, bool kmask, int k0, int kend) {
  constexpr int NLD = BK / 16;
#pragma unroll
  for (int u = 0; u < NLD; ++u) {
    float4 v = reg[u];
    if (TRANS == 0) {
      const int i = (tid >> 3) + 32 * (u & 1), kk = 32 * (u >> 1) + 4 * (tid & 7);
      if (kmask && k0 + kk >= kend) v = make_float4(0.f, 0.f, 0.f, 0.f);
      T[kk + 0][i] = v.x;
      T[kk + 1][i] = v.y;
      T[kk + 2][i] = v.z;
      T[kk + 3][i] = v.w;
    } else {
      const int kk = (tid >> 4) + 16 * u, i = 4 * (tid & 15);
      if (kmask && k0 + kk >= kend) v = make_float4(0.f, 0.f, 0.f, 0.f);
      T[kk][i + 0] = v.x;
      T[kk][i + 1] = v.y;
      T[kk][i + 2] = v.z;
      T[kk][i + 3] = v.w;
    }
  }
}

// ---------------------------------------------------------------------- epilogues (shared by both product forms)
// One 64x64 tile whose four 32x32 accumulators sit in the workgroup's four waves (wave (wm, wn), MFMA 32x32 C layout).
template <int TA, int IDX>
__device__ __forceinline__ void epi_plain(const GemmProblem& P, const f32x16& acc, int m0, int n0, int split, int M, int N,
                                          bool listed, int wm, int wn, int l31, int h, const int* rows_s = nullptr) {
  const int col = n0 + wn * 32 + l31;
  const bool col_ok = col < N;
  const float bias = (P.bias && col_ok && split == 0) ? P.bias[col] : 0.f;
  const float alpha = P.alpha;
  float* const C = P.C + (size_t)split * P.split_stride;
  const int ldc = P.ldc, accumulate = P.accumulate;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (col_ok && row < M) {
      const float v = (acc[r] + bias) * alpha;
      const size_t off = (size_t)((listed && TA == 0) ? (rows_s ? rows_s[row - m0] : P.ridx[row]) : row) * ldc + col;
      if (accumulate == 0) C[off] = v;
      else if (accumulate == 1) C[off] += v;
      else atomicAdd(&C[off], v);
    }
  }
}

__device__ __forceinline__ void epi_stage(float (*Ct)[LDT], const f32x16& acc, int wm, int wn, int l31, int h) {
#pragma unroll
  for (int r = 0; r < 16; ++r) Ct[wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h][wn * 32 + l31] = acc[r];
}

template <int TA, int IDX>
__device__ __forceinline__ void epi_full(const GemmProblem& P, const float (*Ct)[LDT], int m0, int n0, int tm, int split,
                                         int M, int N, bool listed, int tid, const int* rows_s = nullptr) {
  float* const C = P.C + (size_t)split * P.split_stride;
  const float alpha = P.alpha;
  const int ldc = P.ldc, accumulate = P.accumulate;
  const int ccol = tid & 63, gcol = n0 + ccol;
  if (gcol >= N) return;
  const float cbias = (P.bias && split == 0) ? P.bias[gcol] : 0.f;
  const int act = P.act;
  const DropSpec drop = P.drop;
  const bool dropping = drop.thr != 0u, has_res = P.res.mode != RES_NONE;
  const float add2 = (P.out2 && P.add2) ? P.add2[gcol] : 0.f;
  float csum = 0.f;
  // global operands of the epilogue (activation aux, residual) of ALL 16 rows of this thread, fetched up front by loads that
  // are unconditional inside their (kernel-uniform) switch: a load under a per-row condition makes the compiler wait for
  // everything in flight at the join, and the rolled loop this used to be — one row group fetched ahead under `ok ? load : 0`
  // selects — was sixteen dependent round trips (11 us of the 24 us dX product over the step's row list,
  // tools/gemm_f32_stamps.py).  Rows past M repeat row M - 1, so every address is a real one; their values are never used.
  const bool need_aux = act == ACT_GELU_BWD || act == ACT_TANH_BWD;
  float paux[4][4], pres[4][4];
  int prow[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int lrow = min(m0 + 4 * ((tid >> 6) + 4 * i) + q, M - 1);
      // physical row of the operands (rows_s: the tile's 64 list entries, put in LDS by the prologue)
      prow[i][q] = (listed && TA == 0) ? (rows_s ? rows_s[lrow - m0] : P.ridx[lrow]) : lrow;
      paux[i][q] = 0.f; pres[i][q] = 0.f;
    }
  if (need_aux) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) paux[i][q] = P.act_aux[(size_t)prow[i][q] * ldc + gcol];
  }
  if (has_res) {
    const ResMap& R = P.res;
    if (R.mode == RES_DIRECT) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) pres[i][q] = R.ptr[(size_t)prow[i][q] * R.ld + gcol];
    } else if (R.mode == RES_FANIN && !R.ptr && R.extra) {   // the fan-in sums already sit in extra (+ extra2): one row per sequence
      const float* const e2 = R.extra2 ? R.extra2 : R.extra;
      const float w2 = R.extra2 ? 1.f : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int nin = fdiv(prow[i][q], R.dS), pos = prow[i][q] - nin * R.S;
          const float v1 = R.extra[(size_t)nin * R.extra_ld + gcol], v2 = e2[(size_t)nin * R.extra_ld + gcol];
          pres[i][q] = (R.Sq == R.S || pos == R.qpos) ? v1 + w2 * v2 : 0.f;
        }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) pres[i][q] = res_value(R, prow[i][q], gcol);
    }
  }
  // lean form (bias / scale / residual only — the step's dX products): straight-line, nothing re-read from the argument
  // segment per row (the general body below is so large that the compiler keeps none of P's fields in scalar registers: it
  // fetched three of them again for EVERY row, each a scalar-cache round trip in front of the row's store)
  if (!dropping && act == ACT_NONE && !P.aux_out && !P.out2 && !P.colsum && !P.colsum_part && accumulate == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // (no branch per row: a row past M repeats row M - 1 — same address, same value — because a store under a per-row
        // condition made the compiler wait for the previous row's store to be acknowledged before the next LDS read)
        const int lrow = min(4 * ((tid >> 6) + 4 * i) + q, M - 1 - m0);
        C[(size_t)((IDX && TA == 0) ? prow[i][q] : m0 + lrow) * ldc + gcol] = (Ct[lrow][ccol] + cbias) * alpha + pres[i][q];
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int lrow = 4 * ((tid >> 6) + 4 * i);
    const int rbase = m0 + lrow;
    if (rbase >= M) break;
    Philox4 rnd = {0u, 0u, 0u, 0u};
    float hm[4] = {1.f, 1.f, 1.f, 1.f};
    if (dropping && drop.half) {
      // half form (common.h): the 8 lanes that hold columns 8g .. 8g+7 share ONE call per row; lane j of the group draws
      // the call of row rbase + (j & 3) and the group exchanges the words (tiles start at multiples of 64 columns, a lane's
      // column is n0 + (tid & 63): groups are aligned)
      const int j = tid & 7, e = gcol & 7, base = (tid & 63) & ~7;
      if ((N & 7) == 0) {                                  // whole groups are active (lanes past N have left)
        const Philox4 mine = drop_call16(drop, (uint32_t)(rbase + (j & 3)), (uint32_t)gcol >> 3, drop_step(drop));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          Philox4 rq;
          rq.x = __shfl(mine.x, base + q, 64); rq.y = __shfl(mine.y, base + q, 64);
          rq.z = __shfl(mine.z, base + q, 64); rq.w = __shfl(mine.w, base + q, 64);
          hm[q] = drop_half(drop, rq, e);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) hm[q] = drop_half(drop, drop_call16(drop, (uint32_t)(rbase + q), (uint32_t)gcol >> 3, drop_step(drop)), e);
      }
    } else if (dropping) rnd = philox4x32_10((uint32_t)gcol, (uint32_t)rbase >> 2, drop.site, drop_step(drop), drop.k0, drop.k1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = rbase + q;
      if (row >= M) break;
      float v = (Ct[lrow + q][ccol] + cbias) * alpha;
      const size_t off = (size_t)((IDX && TA == 0) ? prow[i][q] : row) * ldc + gcol;
      if (P.aux_out) P.aux_out[off] = v;
      if (act == ACT_GELU) v = gelu_tanh_f(v);
      else if (act == ACT_TANH) v = tanh_fast(v);
      else if (act == ACT_GELU_BWD) v *= gelu_tanh_grad(paux[i][q]);
      else if (act == ACT_TANH_BWD) v *= (1.f - paux[i][q] * paux[i][q]);
      if (dropping) {
        const uint32_t wv = q == 0 ? rnd.x : (q == 1 ? rnd.y : (q == 2 ? rnd.z : rnd.w));
        v *= drop.half ? hm[q] : drop_word(drop, wv);
      }
      v += pres[i][q];
      csum += v;
      if (P.out2) P.out2[(size_t)((IDX && TA == 0) ? prow[i][q] : row) * P.ld2 + gcol] = v + add2;
      if (accumulate == 0) C[off] = v;
      else if (accumulate == 1) C[off] += v;
      else atomicAdd(&C[off], v);
    }
  }
  if (P.colsum_part) P.colsum_part[((size_t)(4 * tm + (tid >> 6)) * 3) * N + gcol] = csum;   // one parked row per (tile, row group)
  else if (P.colsum) atomicAdd(&P.colsum[gcol], csum);
}

#define KIDX_MAX PS_GEMM_KIDX_MAX
// diagnostics (PS_GEMM_STAMP=2 / 3, tools/gemm_f32_stamps.py): s_memtime of the four waves of workgroup (0, 8, 0) at the phase
// boundaries of the fp32 kernel — the step's own dX product over the row list (2) or its K/V projection (3)
#if PS_DIAG_ON      // in-kernel phase stamps: diagnostic build only
#define F32_STAMP(slot)                                                                                              \
  do {                                                                                                               \
    if (g.stamp && blockIdx.x == 0 && blockIdx.y == 8 && blockIdx.z == 0 && (threadIdx.x & 63) == 0) {               \
      unsigned long long t_;                                                                                         \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                     \
      if ((slot) < 32) g.stamp[32 * (threadIdx.x >> 6) + (slot)] = t_;                                               \
    }                                                                                                                \
  } while (0)
#else
#define F32_STAMP(slot) do { } while (0)
#endif
template <int TA, int TB, int FULL, int BK, int PF, int IDX = 0>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmGroup g) {
  fork_signal(g.sig, g.sigval);
  F32_STAMP(0);
  constexpr int NLD = BK / 16;
  __shared__ float As[2][BK][LDT];
  __shared__ float Bs[2][BK][LDT];
  __shared__ int kidx[IDX && TA == 1 ? KIDX_MAX : 1];
  __shared__ int rows_s[IDX && TA == 0 ? BM : 1];

  int prob, split, ftile = 0;
  if (g.flat) {
    const int L = blockIdx.x;
    prob = L >= g.flat0[2] ? 2 : (L >= g.flat0[1] ? 1 : 0);
    const int t = L - g.flat0[prob];
    if (g.flat_xcd) {
      const int s8 = t >> 3, grp = s8 / g.flat_tiles[prob];
      split = (t & 7) + 8 * grp;
      ftile = s8 - grp * g.flat_tiles[prob];
    } else {
      split = t / g.flat_tiles[prob];
      ftile = t - split * g.flat_tiles[prob];
    }
  } else if (g.split_xcd) {       // 3-D grid of a split reduction: all tiles of a split on one XCD, as GemmGroup::flat_xcd does it
    const int T = gridDim.x * gridDim.y, ks = g.p[0].ksplit;
    const int Lz = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;       // dispatch order
    prob = Lz / (T * ks);
    const int t = Lz - prob * T * ks, s8 = t >> 3, grp = s8 / T;
    split = (t & 7) + 8 * grp;
    ftile = s8 - grp * T;
  } else {
    prob = blockIdx.z / g.p[0].ksplit;
    split = blockIdx.z - prob * g.p[0].ksplit;
  }
  const GemmProblem& P = g.p[prob];
  const bool listed = IDX && P.ridx != nullptr;                 // block-uniform
  // XCD-aware tile mapping: workgroups are dealt round-robin over the 8 XCDs (each with its own L2),
  // so give every XCD whole ROW tiles: all column tiles that re-read one A row-tile share an L2.
  int tm = blockIdx.y, tn = blockIdx.x;
  if (g.flat) {
    tm = ftile % g.flat_tm[prob];
    tn = ftile / g.flat_tm[prob];
  } else if (g.split_xcd) {
    tm = ftile % (int)gridDim.y;
    tn = ftile / (int)gridDim.y;
  } else {
    const int nx = gridDim.x, ny = gridDim.y;
    const int L = blockIdx.y * nx + blockIdx.x;       // dispatch order (x fastest)
    const int grp = L / (8 * nx), r = L - grp * 8 * nx;
    const int rows_here = min(8, ny - grp * 8);
    tm = grp * 8 + r % rows_here;
    tn = r / rows_here;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  // row list, ta == 0: the tile's 64 list entries are fetched NOW, beside the list length (the list has room for P.M
  // entries, common.h; what lies past the length is never used), and kept in LDS for the operand loads and the epilogue
  // — after the length they were a second dependent round trip here and a third in front of every epilogue row group
  int my_row = 0;
  if (IDX && TA == 0 && listed && threadIdx.x < BM) my_row = P.ridx[min(m0 + (int)threadIdx.x, P.M - 1)];
  const int nlist = listed ? *P.rcount : 0;
  const int M = (listed && TA == 0) ? nlist : P.M, N = P.N, K = (listed && TA == 1) ? nlist : P.K;
  const int nslab = (K + BK - 1) / BK;
  const int per = (nslab + P.ksplit - 1) / P.ksplit;
  const int kbeg = split * per * BK;
  const int kend = min(K, kbeg + per * BK);
  if (m0 >= M || n0 >= N || kbeg >= kend) return;   // block-uniform
  F32_STAMP(1);
  const bool kmask = (kend - kbeg) % BK != 0;       // ragged reduction tail (never on the hot shapes)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;

  int pr0 = -1, pr1 = -1;
  const int* kmap = nullptr;
  if (IDX && TA == 0) {                                         // the two A rows this thread loads, once (plain
    pr0 = min(m0 + (tid >> 3), M - 1);                          // problems of a row-list launch: the natural rows)
    pr1 = min(m0 + (tid >> 3) + 32, M - 1);
    if (listed) {
      if (tid < BM) rows_s[tid] = my_row;
      __syncthreads();
      pr0 = rows_s[pr0 - m0]; pr1 = rows_s[pr1 - m0];
    }
  }
  if (listed && TA == 1) {                                      // physical reduction rows of this split -> LDS
    for (int i = tid; i < kend - kbeg; i += 256) kidx[i] = P.ridx[kbeg + i];
    __syncthreads();
    kmap = kidx;
  }
  const int* const rows_l = (IDX && TA == 0 && listed) ? rows_s : nullptr;
  F32_STAMP(2);

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // PF slabs are in flight in registers (slot u holds slab  u  mod PF of the current group): with one workgroup per
  // CU — the 8k-row shapes of the step — a single slab of prefetch (1024 MFMA cycles) does not cover a global round trip
  float4 ra[PF][NLD], rb[PF][NLD];
  const int ldb = P.ldb;
  const float* const bs0 = P.Bseg[0];
  const float* const bs1 = P.Bseg[1];
  const float* const bs2 = P.Bseg[2];
  const int bkseg = K > P.kseg ? P.kseg : 0;          // 0: B is one segment

  // every slot load is UNCONDITIONAL (past the end it re-reads the first slab, a cache hit whose result is dropped): a
  // load under a branch makes the number of loads in flight unknown at the join, and the compiler then waits for all
  // of them (vmcnt(0)) where vmcnt(4*(PF-1)) is enough
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const int kl = kbeg + u * BK < kend ? kbeg + u * BK : kbeg;
    tile_load<TA, BK>(P.A, P.lda, m0, M, kl, kend, ra[u], tid, kmask, nullptr, nullptr, 0, pr0, pr1, kmap, kbeg);
    tile_load<TB, BK>(bs0, ldb, n0, N, kl, kend, rb[u], tid, kmask, bs1, bs2, bkseg, -1, -1, TA == 1 ? kmap : nullptr, kbeg);
  }
  F32_STAMP(3);
  tile_store<TA, BK>(As[0], ra[0], tid, kmask, kbeg, kend);
  tile_store<TB, BK>(Bs[0], rb[0], tid, kmask, kbeg, kend);
  __syncthreads();
  F32_STAMP(4);

  int buf = 0;
  for (int kg = kbeg; kg < kend; kg += PF * BK) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int k0 = kg + u * BK;
      if (k0 >= kend) break;                         // block-uniform
      const int kp = k0 + PF * BK < kend ? k0 + PF * BK : kbeg;   // slot u is free (its slab sits in LDS): refill it
      tile_load<TA, BK>(P.A, P.lda, m0, M, kp, kend, ra[u], tid, kmask, nullptr, nullptr, 0, pr0, pr1, kmap, kbeg);            // PF slabs ahead
      tile_load<TB, BK>(bs0, ldb, n0, N, kp, kend, rb[u], tid, kmask, bs1, bs2, bkseg, -1, -1, TA == 1 ? kmap : nullptr, kbeg);
      const float* a_base = &As[buf][h][wm * 32 + l31];
      const float* b_base = &Bs[buf][h][wn * 32 + l31];
#pragma unroll
      for (int q = 0; q < BK / 32; ++q) {
        float av[16], bv[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          av[s] = a_base[(32 * q + 2 * s) * LDT];
          bv[s] = b_base[(32 * q + 2 * s) * LDT];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0);
      }
      if (k0 + BK < kend) {
        const int nx = (u + 1) % PF;                 // compile-time after unrolling
        tile_store<TA, BK>(As[buf ^ 1], ra[nx], tid, kmask, k0 + BK, kend);
        tile_store<TB, BK>(Bs[buf ^ 1], rb[nx], tid, kmask, k0 + BK, kend);
      }
      __syncthreads();
      F32_STAMP(5 + (k0 - kbeg) / BK);
      buf ^= 1;
    }
  }

  // ------------------------------------------------------------------ epilogue
  if (!FULL) {
    epi_plain<TA, IDX>(P, acc, m0, n0, split, M, N, listed, wm, wn, l31, h, rows_l);
    F32_STAMP(31);
    return;
  }
  // FULL: stage the 64x64 tile through LDS so that the (large) epilogue body exists ONCE in the
  // instruction stream: thread t owns column t&63 and rows 4*((t>>6)+4i)+q  (i,q < 4); the four q-rows
  // of one i share a Philox call (counter (col, row>>2), word row&3); stores are 256-B coalesced.
  float (*Ct)[LDT] = reinterpret_cast<float (*)[LDT]>(&As[0][0][0]);   // 64x65 floats fit As for BK >= 32
  epi_stage(Ct, acc, wm, wn, l31, h);
  __syncthreads();
  F32_STAMP(30);
  epi_full<TA, IDX>(P, Ct, m0, n0, tm, split, M, N, listed, tid, rows_l);
  F32_STAMP(31);
}

// ====================================================================== bf16x3 product form
// v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate (4,445 cycles per 32x32x128 product); the same product as SIX
// v_mfma_f32_32x32x16_bf16 over a three-way bf16 split of both operands (x = hi + mid + lo: 3 x 8 = 24 mantissa bits, an
// exact decomposition;  hl + lh + mm + hm + mh + hh, fp32 accumulation) costs 1,457 and is as accurate (max error /
// sum|a b| against fp64: 1.10e-7, the fp32 MFMA's own 1.13e-7 — tools/micro/bf16x3.hip, profiles/r02_mlp_notes.md).
// This kernel is gemm_f32_kernel with that product for the large launches (the d = 256 step's linears over 21,504 rows,
// the review transformer's over 78k): tile (64 MT) x (64 NT), each of the 4 waves owns the 32x32 accumulator (wm, wn) of
// every 64x64 quadrant — so the epilogues above serve unchanged, quadrant by quadrant — 32-deep slabs; fp32 operands
// are fetched one slab ahead into registers, split on their way into LDS (three planes per operand, [row][32 k] bf16,
// 16-byte chunks XOR-swizzled by (row >> 2) & 3 so that ds_read_b128 by its lane groups and ds_write_b128 by 8-lane
// groups are conflict-free), and read back as whole MFMA operands (one ds_read_b128 per plane and 32 rows x 16 k).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define X3K 32

__device__ __forceinline__ int x3g_off(int row, int chunk) { return row * X3K + ((chunk ^ ((row >> 2) & 3)) << 3); }

// two adjacent reduction elements -> one packed pair per plane.  Non-finite operands: hi carries the Inf / NaN and the residual
// Inf - Inf = NaN follows it, so every output the operand reaches is NON-FINITE, as with the fp32 MFMA — but its class is NaN
// where the fp32 kernel gives +-Inf (the form's own terms Inf * b_hi and Inf * b_mid have opposite signs: no split of b can
// preserve the class); tests/test_gpu_gemm_x3.py pins exactly this
__device__ __forceinline__ void x3g_split2(float x0, float x1, uint32_t& hi, uint32_t& mi, uint32_t& lo) {
  f32x2 v = {x0, x1};
  bf16x2 b = __builtin_convertvector(v, bf16x2);                         // v_cvt_pk_bf16_f32 (round to nearest even)
  hi = __builtin_bit_cast(uint32_t, b);
  v -= f32x2{__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};   // v_pk_add_f32: exact
  b = __builtin_convertvector(v, bf16x2);
  mi = __builtin_bit_cast(uint32_t, b);
  v -= f32x2{__uint_as_float(mi << 16), __uint_as_float(mi & 0xffff0000u)};
  b = __builtin_convertvector(v, bf16x2);
  lo = __builtin_bit_cast(uint32_t, b);
}
// eight consecutive reduction elements of one tile row -> its 16-byte chunk in each plane
template <int R>
__device__ __forceinline__ void x3g_put8(uint16_t* planes, int row, int chunk, const float (&v)[8]) {
  uint32_t H[4], Mi[4], Lo[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) x3g_split2(v[2 * e], v[2 * e + 1], H[e], Mi[e], Lo[e]);
  uint16_t* dst = planes + x3g_off(row, chunk);
  *reinterpret_cast<uint4*>(dst) = make_uint4(H[0], H[1], H[2], H[3]);
  *reinterpret_cast<uint4*>(dst + R * X3K) = make_uint4(Mi[0], Mi[1], Mi[2], Mi[3]);
  *reinterpret_cast<uint4*>(dst + 2 * R * X3K) = make_uint4(Lo[0], Lo[1], Lo[2], Lo[3]);
}
// One operand slab (R tile rows x 32 reduction elements) global -> registers, R / 8 floats per thread.
//   TRANS == 0 (src[row][k]): thread = (row, 8-element chunk): q = tid + 256 u, row q >> 2, chunk q & 3: two float4
//   TRANS == 1 (src[k][row]): wave w takes reduction elements 8 w .. 8 w + 7, lane l the R / 64 adjacent rows at (R / 64) l
//                             (their 16-byte LDS stores are 2-way conflicts: 16 LDS-array cycles under a 13-cycle store)
//                             (measured alternative for weight gradients: waves 0-1 fetch A, waves 2-3 B, four float4 loads of
//                             4 rows x 4 reduction elements per thread and 8-byte LDS stores — 1024x256x21504: 120 -> 115 us,
//                             256x256x21504: 55 -> 63 us, 128x384x78336: 164 -> 197 us, C2 step unchanged: not kept)
// Rows are clamped (duplicates land in outputs the epilogue never stores), reduction indices past the end too (x3g_store
// zero-fills them); the loads are unconditional and untouched until the store (see tile_load).
template <int TRANS, int R>
__device__ __forceinline__ void x3g_load(const float* __restrict__ src, int ld, int row0, int nrows, int k0, int kend,
                                         bool kmask, float (&reg)[R / 8], int tid, const float* __restrict__ seg1,
                                         const float* __restrict__ seg2, int kseg, int pr0, int pr1, const int* kmap,
                                         int kbase) {
  if (TRANS == 0) {
#pragma unroll
    for (int u = 0; u < R / 64; ++u) {
      const int q = tid + 256 * u, row = q >> 2, c = q & 3;
      const int r = pr0 >= 0 ? (u ? pr1 : pr0) : min(row0 + row, nrows - 1);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int kl = min(k0 + 8 * c + 4 * j, kend - 4);              // (kend % 4 == 0: validated)
        const float* base = src;
        if (kseg > 0) {
          const int sg = (kl >= kseg) + (kl >= 2 * kseg);
          base = sg == 0 ? src : (sg == 1 ? seg1 : seg2);
          kl -= sg * kseg;
        }
        const float4 t = *reinterpret_cast<const float4*>(base + (size_t)r * ld + kl);
        reg[8 * u + 4 * j + 0] = t.x; reg[8 * u + 4 * j + 1] = t.y; reg[8 * u + 4 * j + 2] = t.z; reg[8 * u + 4 * j + 3] = t.w;
      }
    }
  } else {
    constexpr int RPT = R / 64;
    const int w = tid >> 6;
    const int r = min(row0 + RPT * (tid & 63), nrows - RPT);      // nrows % 4 == 0 (validated)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int kl = min(k0 + 8 * w + j, kend - 1);
      if (kmap) kl = kmap[kl - kbase];
      const float* base = src;
      if (kseg > 0) {
        const int sg = (kl >= kseg) + (kl >= 2 * kseg);
        base = sg == 0 ? src : (sg == 1 ? seg1 : seg2);
        kl -= sg * kseg;
      }
      if (RPT == 2) {
        const float2 t = *reinterpret_cast<const float2*>(base + (size_t)kl * ld + r);
        reg[2 * j] = t.x; reg[2 * j + 1] = t.y;
      } else {
        reg[j] = base[(size_t)kl * ld + r];
      }
    }
  }
}
template <int TRANS, int R, bool KMASK>
__device__ __forceinline__ void x3g_store_impl(uint16_t* planes, const float (&reg)[R / 8], int tid, int k0, int kend) {
  if (TRANS == 0) {
#pragma unroll
    for (int u = 0; u < R / 64; ++u) {
      const int q = tid + 256 * u, row = q >> 2, c = q & 3;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (KMASK && k0 + 8 * c + e >= kend) ? 0.f : reg[8 * u + e];
      x3g_put8<R>(planes, row, c, v);
    }
  } else {
    constexpr int RPT = R / 64;
    const int w = tid >> 6;
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (KMASK && k0 + 8 * w + j >= kend) ? 0.f : reg[RPT * j + u];
      x3g_put8<R>(planes, RPT * (tid & 63) + u, w, v);
    }
  }
}
// (the zero-fill of a ragged last slab as selects in the common path cost ~100 vector instructions per slab: a block-uniform branch)
template <int TRANS, int R>
__device__ __forceinline__ void x3g_store(uint16_t* planes, const float (&reg)[R / 8], int tid, bool kmask, int k0, int kend) {
  if (kmask && k0 + X3K > kend) x3g_store_impl<TRANS, R, true>(planes, reg, tid, k0, kend);
  else x3g_store_impl<TRANS, R, false>(planes, reg, tid, k0, kend);
}

#if PS_DIAG_ON      // in-kernel phase stamps: diagnostic build only
#define GEMM_STAMP(slot)                                                                                             \
  do {                                                                                                               \
    if (g.stamp && blockIdx.x == 0 && blockIdx.y == gridDim.y / 2 && blockIdx.z == 0 && (threadIdx.x & 63) == 0) {   \
      unsigned long long t_;                                                                                         \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                     \
      if ((slot) < 32) g.stamp[32 * (threadIdx.x >> 6) + (slot)] = t_;                                               \
    }                                                                                                                \
  } while (0)
#else
#define GEMM_STAMP(slot) do { } while (0)
#endif
template <int TA, int TB, int FULL, int MT, int NT, int IDX, int PF = 1>
__global__ __launch_bounds__(256, 2) void gemm_x3_kernel(const GemmGroup g) {
  fork_signal(g.sig, g.sigval);
  GEMM_STAMP(0);
  constexpr int RA = 64 * MT, RB = 64 * NT;
  constexpr int MAIN_BYTES = 3 * (RA + RB) * X3K * 2;
  constexpr int EPI_BYTES = FULL ? MT * NT * 64 * LDT * 4 : 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES];
  __shared__ int kidx[IDX && TA == 1 ? KIDX_MAX : 1];
  uint16_t* const As = reinterpret_cast<uint16_t*>(smem);
  uint16_t* const Bs = As + 3 * RA * X3K;

  int prob, split, ftile = 0;
  if (g.flat) {
    const int L = blockIdx.x;
    prob = L >= g.flat0[2] ? 2 : (L >= g.flat0[1] ? 1 : 0);
    const int t = L - g.flat0[prob];
    if (g.flat_xcd) {
      const int s8 = t >> 3, grp = s8 / g.flat_tiles[prob];
      split = (t & 7) + 8 * grp;
      ftile = s8 - grp * g.flat_tiles[prob];
    } else {
      split = t / g.flat_tiles[prob];
      ftile = t - split * g.flat_tiles[prob];
    }
  } else if (g.split_xcd) {       // 3-D grid of a split reduction: all tiles of a split on one XCD, as GemmGroup::flat_xcd does it
    const int T = gridDim.x * gridDim.y, ks = g.p[0].ksplit;
    const int Lz = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;       // dispatch order
    prob = Lz / (T * ks);
    const int t = Lz - prob * T * ks, s8 = t >> 3, grp = s8 / T;
    split = (t & 7) + 8 * grp;
    ftile = s8 - grp * T;
  } else {
    prob = blockIdx.z / g.p[0].ksplit;
    split = blockIdx.z - prob * g.p[0].ksplit;
  }
  const GemmProblem& P = g.p[prob];
  const bool listed = IDX && P.ridx != nullptr;                 // block-uniform
  const int nlist = listed ? *P.rcount : 0;
  const int M = (listed && TA == 0) ? nlist : P.M, N = P.N, K = (listed && TA == 1) ? nlist : P.K;
  int tm, tn;                                                    // XCD-aware tile mapping as in gemm_f32_kernel
  if (g.flat) {
    tm = ftile % g.flat_tm[prob];
    tn = ftile / g.flat_tm[prob];
  } else if (g.split_xcd) {
    tm = ftile % (int)gridDim.y;
    tn = ftile / (int)gridDim.y;
  } else {
    const int nx = gridDim.x, ny = gridDim.y;
    const int L = blockIdx.y * nx + blockIdx.x;
    const int grp = L / (8 * nx), r = L - grp * 8 * nx;
    const int rows_here = min(8, ny - grp * 8);
    tm = grp * 8 + r % rows_here;
    tn = r / rows_here;
  }
  const int m0 = tm * RA, n0 = tn * RB;
  const int nslab = (K + X3K - 1) / X3K;
  const int per = (nslab + P.ksplit - 1) / P.ksplit;
  const int kbeg = split * per * X3K;
  const int kend = min(K, kbeg + per * X3K);
  if (m0 >= M || n0 >= N || kbeg >= kend) return;   // block-uniform
  const bool kmask = (kend - kbeg) % X3K != 0;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;

  int pr0 = -1, pr1 = -1;
  const int* kmap = nullptr;
  if (IDX && TA == 0) {                                         // the A rows this thread loads, once
    pr0 = min(m0 + (tid >> 2), M - 1);
    pr1 = min(m0 + (tid >> 2) + 64, M - 1);
    if (listed) { pr0 = P.ridx[pr0]; pr1 = P.ridx[pr1]; }
  }
  if (listed && TA == 1) {                                      // physical reduction rows of this split -> LDS
    for (int i = tid; i < kend - kbeg; i += 256) kidx[i] = P.ridx[kbeg + i];
    __syncthreads();
    kmap = kidx;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // PF slabs in flight in registers (slot u holds slab u mod PF of the current group), every load unconditional — see gemm_f32_kernel
  float ra[PF][RA / 8], rb[PF][RB / 8];
  const int ldb = P.ldb;
  const float* const bs0 = P.Bseg[0];
  const float* const bs1 = P.Bseg[1];
  const float* const bs2 = P.Bseg[2];
  const int bkseg = K > P.kseg ? P.kseg : 0;          // 0: B is one segment
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const int kl = kbeg + u * X3K < kend ? kbeg + u * X3K : kbeg;
    x3g_load<TA, RA>(P.A, P.lda, m0, M, kl, kend, kmask, ra[u], tid, nullptr, nullptr, 0, pr0, pr1, kmap, kbeg);
    x3g_load<TB, RB>(bs0, ldb, n0, N, kl, kend, kmask, rb[u], tid, bs1, bs2, bkseg, -1, -1, TA == 1 ? kmap : nullptr, kbeg);
  }

  GEMM_STAMP(1);
  // one slab: publish slot `ra_u / rb_u` (slab k0), refill it two slabs ahead, multiply
  auto slab = [&](float (&ra_u)[RA / 8], float (&rb_u)[RB / 8], const int k0) __attribute__((always_inline)) {
    const int si = (k0 - kbeg) / X3K;
    x3g_store<TA, RA>(As, ra_u, tid, kmask, k0, kend);
    x3g_store<TB, RB>(Bs, rb_u, tid, kmask, k0, kend);
    GEMM_STAMP(2 + 4 * si);
    __syncthreads();
    GEMM_STAMP(3 + 4 * si);
    const int kp = k0 + PF * X3K < kend ? k0 + PF * X3K : kbeg;
    x3g_load<TA, RA>(P.A, P.lda, m0, M, kp, kend, kmask, ra_u, tid, nullptr, nullptr, 0, pr0, pr1, kmap, kbeg);
    x3g_load<TB, RB>(bs0, ldb, n0, N, kp, kend, kmask, rb_u, tid, bs1, bs2, bkseg, -1, -1, TA == 1 ? kmap : nullptr, kbeg);
#pragma unroll
    for (int s = 0; s < X3K / 16; ++s) {
      bf16x8 a[MT][3], b[NT][3];
      const int cl = 2 * s + h;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
          a[mi][p] = *reinterpret_cast<const bf16x8*>(As + p * RA * X3K + x3g_off(64 * mi + 32 * wm + l31, cl));
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          b[ni][p] = *reinterpret_cast<const bf16x8*>(Bs + p * RB * X3K + x3g_off(64 * ni + 32 * wn + l31, cl));
      }
      // small terms first; the MT x NT accumulators of one term are independent MFMAs
#define X3G_TERM(pa, pb)                                                                                          \
  _Pragma("unroll") for (int mi = 0; mi < MT; ++mi) _Pragma("unroll") for (int ni = 0; ni < NT; ++ni)              \
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][pa], b[ni][pb], acc[mi][ni], 0, 0, 0);
      X3G_TERM(0, 2) X3G_TERM(2, 0) X3G_TERM(1, 1) X3G_TERM(0, 1) X3G_TERM(1, 0) X3G_TERM(0, 0)
#undef X3G_TERM
    }
    GEMM_STAMP(4 + 4 * si);
    __syncthreads();
    GEMM_STAMP(5 + 4 * si);
  };
  // slabs go in pairs with no exit between them — a `break` inside the pair joins paths with different loads in flight
  // and the compiler then waits for ALL of them (vmcnt(0)) at the next store; an odd last slab runs after the loop
  static_assert(PF == 1 || PF == 2, "one or two register slots");
  const int nsl = (kend - kbeg + X3K - 1) / X3K;
  int k0 = kbeg;
  if (PF == 1) {
    for (int it = 0; it < nsl; ++it, k0 += X3K) slab(ra[0], rb[0], k0);
  } else {
    for (int it = 0; it < (nsl >> 1); ++it) {
      slab(ra[0], rb[0], k0);
      slab(ra[PF - 1], rb[PF - 1], k0 + X3K);
      k0 += 2 * X3K;
    }
    if (nsl & 1) slab(ra[0], rb[0], k0);
  }

  GEMM_STAMP(30);
  // ------------------------------------------------------------------ epilogue, one 64x64 quadrant at a time
  if (!FULL) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
        if (m0 + 64 * mi < M && n0 + 64 * ni < N)
          epi_plain<TA, IDX>(P, acc[mi][ni], m0 + 64 * mi, n0 + 64 * ni, split, M, N, listed, wm, wn, l31, h);
    return;
  }
  float (*Ct)[64][LDT] = reinterpret_cast<float (*)[64][LDT]>(smem);       // all quadrants staged, ONE rolled body
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) epi_stage(Ct[mi * NT + ni], acc[mi][ni], wm, wn, l31, h);
  __syncthreads();
#pragma unroll 1
  for (int q = 0; q < MT * NT; ++q) {
    const int mq = m0 + 64 * (q / NT), nq = n0 + 64 * (q % NT);
    if (mq < M && nq < N) epi_full<TA, IDX>(P, Ct[q], mq, nq, MT * tm + q / NT, split, M, N, listed, tid);
  }
  GEMM_STAMP(31);
}

// ====================================================================== bf16x3 product form, operands straight into LDS (round 4)
// gemm_x3_kernel above spends as long moving a slab VGPR -> split -> ds_write_b128 as multiplying it, and the two phases ADD
// (profiles/r03_gemm_notes.md: fixed 31 + products 36 + loads / split / LDS stores 41 of the FF1 product's 100 us).  This form has
// no store phase at all:
//   * the fp32 operand slabs go global -> LDS by global_load_lds_dwordx4 (LDS-DMA: no VGPR staging, no ds_write); the LDS image
//     is the DMA's lane-linear one, the bank swizzle sits on the per-lane SOURCE address (k contiguous: 128-byte rows, 16-byte
//     chunk c of row r is stored at chunk c ^ ((r >> 1) & 7), so the ds_read_b128 lane groups of a fragment read hit 16
//     different 16-byte slots; row contiguous: [k][128 rows], a fragment is eight conflict-free ds_read_b32);
//   * a wave reads the fp32 fragments of a WHOLE 32-deep slab into registers (64 VGPRs), then the workgroup's loads of the
//     next slab are issued into the other LDS stage and fly under the slab's products: one barrier per slab, and hipcc's
//     forced vmcnt(0) in front of a ds_read that follows an LDS-DMA (SIInsertWaitcnts cannot tell the stages apart) falls
//     where the loop has to wait anyway;
//   * the three-way split happens on the fragments in registers, between the MFMAs (a wave pays it for the operands of its own
//     64x64 accumulator only): tile 128x128, 4 waves, each wave the (wm, wn) 32x32 block of every 64x64 quadrant as above — so
//     the epilogues serve unchanged — = 0.5 ds_read_b128 per MFMA; 64 KB of LDS per workgroup, two workgroups per CU, the other
//     workgroup's wave on the SIMD multiplies while this one splits.
// Takes what the kernel above takes except a ragged reduction tail and K-concatenated B operands (the caller falls back).
#define X3D_BK 32
#define X3D_T 128
#define X3D_STAGE (2 * X3D_T * X3D_BK * 4)          // bytes of one stage: A slab + B slab

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// the four 16-byte pieces this thread fetches of one operand's slab (wave w issues instructions 4 w .. 4 w + 3, each 1 KB)
template <int TRANS>
__device__ __forceinline__ void x3d_issue(const float* const (&p)[4], size_t step, unsigned char* stage, int wave) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(p[j] + step), (lds_ptr_t)(stage + (4 * wave + j) * 1024), 16, 0, 0);
}

// fragment (8 reduction elements of one tile row per lane) of k-step s from a stage image, fp32
template <int TRANS>
__device__ __forceinline__ void x3d_frag(const unsigned char* img, int row, int s, int h, int swz, float (&f)[8]) {
  if (TRANS == 0) {
    const int c = 4 * s + 2 * h;
    const float4 u = *reinterpret_cast<const float4*>(img + row * 128 + ((c ^ swz) << 4));
    const float4 v = *reinterpret_cast<const float4*>(img + row * 128 + (((c + 1) ^ swz) << 4));
    f[0] = u.x; f[1] = u.y; f[2] = u.z; f[3] = u.w; f[4] = v.x; f[5] = v.y; f[6] = v.z; f[7] = v.w;
  } else {
    const float* t = reinterpret_cast<const float*>(img) + (16 * s + 8 * h) * X3D_T + row;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = t[j * X3D_T];
  }
}
// The same split for the kernels that split FRAGMENTS between their MFMAs (gemm_x3d_kernel, gemm_x3w_kernel): the residuals by two
// plain v_sub_f32 instead of one v_pk_add_f32 (inline asm: -O3 re-packs adjacent scalar subtractions) — packed fp32 VALU is the one
// vector instruction class that does not co-issue with bf16 MFMAs (MI355X_MICROARCH.md, cycle constants: +13 cycles each beside an
// MFMA).  Measured both ways in every kernel (profiles/r04_gemm_notes.md): here 2-3 % faster (4096^3 783 -> 765 us, the 21,504-row
// weight gradients 289 -> 279), in gemm_x3_kernel — whose split runs in a phase of its own, away from the MFMAs — 9 % SLOWER
// (193 -> 211 us), so that kernel keeps the packed form.  Same values bit for bit (both subtractions are exact).
__device__ __forceinline__ float x3d_sub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ void x3d_split2(float x0, float x1, uint32_t& hi, uint32_t& mi, uint32_t& lo) {
  hi = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2{x0, x1}), bf16x2));
  float r0 = x3d_sub(x0, __uint_as_float(hi << 16)), r1 = x3d_sub(x1, __uint_as_float(hi & 0xffff0000u));
  mi = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2{r0, r1}), bf16x2));
  r0 = x3d_sub(r0, __uint_as_float(mi << 16)); r1 = x3d_sub(r1, __uint_as_float(mi & 0xffff0000u));
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2{r0, r1}), bf16x2));
}
__device__ __forceinline__ void x3d_split8(const float (&f)[8], bf16x8 (&pl)[3]) {
  uint32_t H[4], Mi[4], Lo[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) x3d_split2(f[2 * e], f[2 * e + 1], H[e], Mi[e], Lo[e]);
  pl[0] = __builtin_bit_cast(bf16x8, make_uint4(H[0], H[1], H[2], H[3]));
  pl[1] = __builtin_bit_cast(bf16x8, make_uint4(Mi[0], Mi[1], Mi[2], Mi[3]));
  pl[2] = __builtin_bit_cast(bf16x8, make_uint4(Lo[0], Lo[1], Lo[2], Lo[3]));
}

// DIAG (diagnostic build only, PS_X3D_DIAG; WRONG results): 1 no split (raw bits as planes), 2 no MFMAs, 3 no slab loads in the loop
template <int TA, int TB, int FULL, int IDX, int DIAG = 0>
__global__ __launch_bounds__(256, 2) void gemm_x3d_kernel(const GemmGroup g) {
  fork_signal(g.sig, g.sigval);
  constexpr int MT = 2, NT = 2;
  constexpr int MAIN_BYTES = 2 * X3D_STAGE;
  constexpr int EPI_BYTES = FULL ? MT * NT * 64 * LDT * 4 : 16;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES];
  __shared__ int kidx[IDX && TA == 1 ? KIDX_MAX : 1];

  int prob, split, ftile = 0;
  if (g.flat) {
    const int L = blockIdx.x;
    prob = L >= g.flat0[2] ? 2 : (L >= g.flat0[1] ? 1 : 0);
    const int t = L - g.flat0[prob];
    if (g.flat_xcd) {
      const int s8 = t >> 3, grp = s8 / g.flat_tiles[prob];
      split = (t & 7) + 8 * grp;
      ftile = s8 - grp * g.flat_tiles[prob];
    } else {
      split = t / g.flat_tiles[prob];
      ftile = t - split * g.flat_tiles[prob];
    }
  } else if (g.split_xcd) {       // 3-D grid of a split reduction: all tiles of a split on one XCD, as GemmGroup::flat_xcd does it
    const int T = gridDim.x * gridDim.y, ks = g.p[0].ksplit;
    const int Lz = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;       // dispatch order
    prob = Lz / (T * ks);
    const int t = Lz - prob * T * ks, s8 = t >> 3, grp = s8 / T;
    split = (t & 7) + 8 * grp;
    ftile = s8 - grp * T;
  } else {
    prob = blockIdx.z / g.p[0].ksplit;
    split = blockIdx.z - prob * g.p[0].ksplit;
  }
  const GemmProblem& P = g.p[prob];
  const bool listed = IDX && P.ridx != nullptr;                 // block-uniform
  const int nlist = listed ? *P.rcount : 0;
  const int M = (listed && TA == 0) ? nlist : P.M, N = P.N, K = (listed && TA == 1) ? nlist : P.K;
  int tm, tn;                                                    // XCD-aware tile mapping as in gemm_f32_kernel
  if (g.flat) {
    tm = ftile % g.flat_tm[prob];
    tn = ftile / g.flat_tm[prob];
  } else if (g.split_xcd) {
    tm = ftile % (int)gridDim.y;
    tn = ftile / (int)gridDim.y;
  } else {
    const int nx = gridDim.x, ny = gridDim.y;
    const int L = blockIdx.y * nx + blockIdx.x;
    const int grp = L / (8 * nx), r = L - grp * 8 * nx;
    const int rows_here = min(8, ny - grp * 8);
    tm = grp * 8 + r % rows_here;
    tn = r / rows_here;
  }
  const int m0 = tm * X3D_T, n0 = tn * X3D_T;
  const int nslab = (K + X3D_BK - 1) / X3D_BK;
  const int per = (nslab + P.ksplit - 1) / P.ksplit;
  const int kbeg = split * per * X3D_BK;
  const int kend = min(K, kbeg + per * X3D_BK);
  if (m0 >= M || n0 >= N || kbeg >= kend) return;   // block-uniform
  const int nsl = (kend - kbeg + X3D_BK - 1) / X3D_BK;      // listed weight gradients: a ragged last slab repeats the list's last row ...
  const int ktail = kend - kbeg - (nsl - 1) * X3D_BK;       // ... and the fragments of the repeated rows are zeroed (below)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;

  // ---- per-thread sources of the slab loads
  const float* pa[4];
  const float* pb[4];
  size_t stepa, stepb;                                           // floats per slab
  if (listed && TA == 1) {                                       // physical reduction rows of this split -> LDS (padded by the last one)
    for (int i = tid; i < nsl * X3D_BK; i += 256) kidx[i] = P.ridx[min(kbeg + i, kend - 1)];
    __syncthreads();
  }
  if (TA == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 32 * wave + 8 * j + (lane >> 3);
      int row = min(m0 + r, M - 1);
      if (listed) row = P.ridx[row];
      pa[j] = P.A + (size_t)row * P.lda + kbeg + 4 * ((lane & 7) ^ ((r >> 1) & 7));
    }
    stepa = X3D_BK;
  } else {
    const int col = min(m0 + 4 * (lane & 31), M - 4);          // M % 4 == 0 (validated)
#pragma unroll
    for (int j = 0; j < 4; ++j) pa[j] = P.A + (size_t)(listed ? 0 : kbeg + 8 * wave + 2 * j + (lane >> 5)) * P.lda + col;
    stepa = (size_t)X3D_BK * P.lda;
  }
  if (TB == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 32 * wave + 8 * j + (lane >> 3);
      pb[j] = P.Bseg[0] + (size_t)min(n0 + r, N - 1) * P.ldb + kbeg + 4 * ((lane & 7) ^ ((r >> 1) & 7));
    }
    stepb = X3D_BK;
  } else {
    const int col = min(n0 + 4 * (lane & 31), N - 4);          // N % 4 == 0 (validated)
#pragma unroll
    for (int j = 0; j < 4; ++j) pb[j] = P.Bseg[0] + (size_t)((listed && TA == 1) ? 0 : kbeg + 8 * wave + 2 * j + (lane >> 5)) * P.ldb + col;
    stepb = (size_t)X3D_BK * P.ldb;
  }
  auto issue = [&](int it) __attribute__((always_inline)) {
    unsigned char* const stage = smem + (it & 1) * X3D_STAGE;
    if (listed && TA == 1) {                                     // reduction rows through the list
      const float* qa[4];
      const float* qb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const size_t kr = (size_t)kidx[it * X3D_BK + 8 * wave + 2 * j + (lane >> 5)];
        qa[j] = pa[j] + kr * P.lda;
        qb[j] = pb[j] + kr * P.ldb;
      }
      x3d_issue<TA>(qa, 0, stage, wave);
      x3d_issue<TB>(qb, 0, stage + X3D_STAGE / 2, wave);
    } else {
      x3d_issue<TA>(pa, it * stepa, stage, wave);
      x3d_issue<TB>(pb, it * stepb, stage + X3D_STAGE / 2, wave);
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int swz = (l31 >> 1) & 7;
  issue(0);
  for (int it = 0; it < nsl; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this thread's pieces of slab `it` have landed
    __syncthreads();                                             // everybody's; and everybody is done with the other stage
    const unsigned char* const sa = smem + (it & 1) * X3D_STAGE;
    const unsigned char* const sb = sa + X3D_STAGE / 2;
    float fa[2][MT][8], fb[2][NT][8];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) x3d_frag<TA>(sa, 64 * mi + 32 * wm + l31, s, h, swz, fa[s][mi]);
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) x3d_frag<TB>(sb, 64 * ni + 32 * wn + l31, s, h, swz, fb[s][ni]);
    }
    if (it + 1 < nsl && DIAG != 3) issue(it + 1);                // block-uniform; flies under the products below
    if (IDX && TA == 1 && it == nsl - 1 && ktail < X3D_BK) {     // ragged tail of a listed reduction: zero A's repeated rows
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (16 * s + 8 * h + j >= ktail) fa[s][mi][j] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[MT][3], b[NT][3];
      if (DIAG == 1) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          a[mi][0] = a[mi][2] = __builtin_bit_cast(bf16x8, make_float4(fa[s][mi][0], fa[s][mi][1], fa[s][mi][2], fa[s][mi][3]));
          a[mi][1] = __builtin_bit_cast(bf16x8, make_float4(fa[s][mi][4], fa[s][mi][5], fa[s][mi][6], fa[s][mi][7]));
        }
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
          b[ni][0] = b[ni][2] = __builtin_bit_cast(bf16x8, make_float4(fb[s][ni][0], fb[s][ni][1], fb[s][ni][2], fb[s][ni][3]));
          b[ni][1] = __builtin_bit_cast(bf16x8, make_float4(fb[s][ni][4], fb[s][ni][5], fb[s][ni][6], fb[s][ni][7]));
        }
      } else {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) x3d_split8(fa[s][mi], a[mi]);
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) x3d_split8(fb[s][ni], b[ni]);
      }
      if (DIAG == 2) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int p = 0; p < 3; ++p) { asm volatile("" ::"v"(a[mi][p])); asm volatile("" ::"v"(b[mi][p])); }
        continue;
      }
#define X3D_TERM(pa_, pb_)                                                                                        \
  _Pragma("unroll") for (int mi = 0; mi < MT; ++mi) _Pragma("unroll") for (int ni = 0; ni < NT; ++ni)              \
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][pa_], b[ni][pb_], acc[mi][ni], 0, 0, 0);
      X3D_TERM(0, 2) X3D_TERM(2, 0) X3D_TERM(1, 1) X3D_TERM(0, 1) X3D_TERM(1, 0) X3D_TERM(0, 0)
#undef X3D_TERM
    }
  }

  // ------------------------------------------------------------------ epilogue, one 64x64 quadrant at a time (as gemm_x3_kernel)
  if (!FULL) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
        if (m0 + 64 * mi < M && n0 + 64 * ni < N)
          epi_plain<TA, IDX>(P, acc[mi][ni], m0 + 64 * mi, n0 + 64 * ni, split, M, N, listed, wm, wn, l31, h);
    return;
  }
  __syncthreads();                                               // the last slab's fragments have left LDS in every wave
  float (*Ct)[64][LDT] = reinterpret_cast<float (*)[64][LDT]>(smem);
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) epi_stage(Ct[mi * NT + ni], acc[mi][ni], wm, wn, l31, h);
  __syncthreads();
#pragma unroll 1
  for (int q = 0; q < MT * NT; ++q) {
    const int mq = m0 + 64 * (q / NT), nq = n0 + 64 * (q % NT);
    if (mq < M && nq < N) epi_full<TA, IDX>(P, Ct[q], mq, nq, MT * tm + q / NT, split, M, N, listed, tid);
  }
}

// ====================================================================== bf16x3 products against PRE-SPLIT weights (round 4)
// In the products of a layer's forward and input gradients the B operand is a WEIGHT: a few hundred KB that every one of the
// launch's hundreds of workgroups used to split again, slab by slab (and, for the dX products, fetch through the row-contiguous
// loader with its strided 4 / 8-byte accesses).  WPlaneScope splits each weight once per entry-point call into bf16 plane
// matrices [3][N][K] in BOTH orientations; this kernel then
//   * brings B's planes global -> LDS by global_load_lds_dwordx4 ([row][32 k] bf16 images, 16-byte chunks XOR-swizzled by
//     (row >> 2) & 3 on the source address: the read pattern of gemm_x3_kernel, SQ_LDS_BANK_CONFLICT 0) and feeds them to the
//     MFMAs as they are: no VALU, no LDS store;
//   * brings A (the activation, fp32) in as gemm_x3d_kernel does and splits its fragments in registers — but the four waves
//     are stacked 4 x 1 over the 128-row tile (wave tile 32 x 128), so every A fragment is split by exactly ONE wave: a quarter
//     of gemm_x3d_kernel's split work per MFMA, an eighth of gemm_x3_kernel's per element (which re-split A for every 64-wide
//     column tile);
//   * tile 128 x 128, 32-deep slabs, two LDS stages of 16 KB (A) + 24 KB (B planes), one barrier per slab, two workgroups per CU.
// K-concatenated B operands (the K/V dX product: [dK | dV] against [Wk ; Wv]) pick the segment per slab (kseg % 32 == 0).
#define X3W_STAGE (X3D_T * X3D_BK * 4 + 3 * X3D_T * X3D_BK * 2)      // 16 KB + 24 KB
static bool x3_on();

template <int FULL, int IDX>
__global__ __launch_bounds__(256, 2) void gemm_x3w_kernel(const GemmGroup g) {
  fork_signal(g.sig, g.sigval);
  constexpr int MAIN_BYTES = 2 * X3W_STAGE;
  constexpr int EPI_BYTES = FULL ? 4 * 64 * LDT * 4 : 16;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES];

  const int prob = blockIdx.z;
  const GemmProblem& P = g.p[prob];
  const bool listed = IDX && P.ridx != nullptr;                 // block-uniform
  const int nlist = listed ? *P.rcount : 0;
  const int M = listed ? nlist : P.M, N = P.N, K = P.K;
  int tm, tn;                                                    // XCD-aware tile mapping as in gemm_f32_kernel
  {
    const int nx = gridDim.x, ny = gridDim.y;
    const int L = blockIdx.y * nx + blockIdx.x;
    const int grp = L / (8 * nx), r = L - grp * 8 * nx;
    const int rows_here = min(8, ny - grp * 8);
    tm = grp * 8 + r % rows_here;
    tn = r / rows_here;
  }
  const int m0 = tm * X3D_T, n0 = tn * X3D_T;
  if (m0 >= M || n0 >= N) return;                               // block-uniform
  const int nsl = K / X3D_BK;                                    // (K % 32 == 0: the launcher checked)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;

  // ---- per-thread sources of the slab loads.  A: 16 instructions of 8 rows x 128 B, wave w issues 4 w .. 4 w + 3.
  const float* pa[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = 32 * wave + 8 * j + (lane >> 3);
    int row = min(m0 + r, M - 1);
    if (listed) row = P.ridx[row];
    pa[j] = P.A + (size_t)row * P.lda + 4 * ((lane & 7) ^ ((r >> 1) & 7));
  }
  // B planes: 24 instructions of 16 rows x 64 B (3 planes x 8 row blocks), wave w issues 6 w .. 6 w + 5
  const int kseg = P.kseg;
  size_t pboff[6];                                               // element offset inside a segment's plane set, slab 0
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int q = 6 * wave + j, pl = q >> 3, rb = q & 7;
    const int r = 16 * rb + (lane >> 2);
    const int row = min(n0 + r, N - 1);
    pboff[j] = ((size_t)pl * N + row) * kseg + 8 * ((lane & 3) ^ ((r >> 2) & 3));
  }
  auto issue = [&](int it) __attribute__((always_inline)) {
    unsigned char* const stage = smem + (it & 1) * X3W_STAGE;
    const int k0 = it * X3D_BK;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(pa[j] + k0), (lds_ptr_t)(stage + (4 * wave + j) * 1024), 16, 0, 0);
    const int sg = k0 >= kseg ? (k0 >= 2 * kseg ? 2 : 1) : 0;   // block-uniform
    const uint16_t* const bp = (sg == 0 ? P.bpl[0] : (sg == 1 ? P.bpl[1] : P.bpl[2])) + (k0 - sg * kseg);
#pragma unroll
    for (int j = 0; j < 6; ++j)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(bp + pboff[j]), (lds_ptr_t)(stage + X3D_T * X3D_BK * 4 + (6 * wave + j) * 1024), 16, 0, 0);
  };

  f32x16 acc[4];                                                 // column block j = 2 ni + wn of the tile, rows 32 wave ..
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int swz = (l31 >> 1) & 7, bsw = (l31 >> 2) & 3;
  issue(0);
  for (int it = 0; it < nsl; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned char* const sa = smem + (it & 1) * X3W_STAGE;
    const unsigned char* const sb = sa + X3D_T * X3D_BK * 4;
    float fa[2][8];
    bf16x8 b[2][4][3];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      x3d_frag<0>(sa, 32 * wave + l31, s, h, swz, fa[s]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          b[s][j][p] = *reinterpret_cast<const bf16x8*>(sb + p * (X3D_T * 64) + (32 * j + l31) * 64 + (((2 * s + h) ^ bsw) << 4));
    }
    if (it + 1 < nsl) issue(it + 1);                             // block-uniform; flies under the products below
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[3];
      x3d_split8(fa[s], a);
#define X3W_TERM(pa_, pb_)                                                                                        \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                    \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa_], b[s][j][pb_], acc[j], 0, 0, 0);
      X3W_TERM(0, 2) X3W_TERM(2, 0) X3W_TERM(1, 1) X3W_TERM(0, 1) X3W_TERM(1, 0) X3W_TERM(0, 0)
#undef X3W_TERM
    }
  }

  // ------------------------------------------------------------------ epilogue: this wave holds block (wm, wn) = (wave & 1, j & 1) of
  // the 64x64 quadrants (mi, ni) = (wave >> 1, j >> 1) — the layout the shared epilogues expect
  const int mi = wave >> 1, wm = wave & 1;
  if (!FULL) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (m0 + 64 * mi < M && n0 + 64 * (j >> 1) < N)
        epi_plain<0, IDX>(P, acc[j], m0 + 64 * mi, n0 + 64 * (j >> 1), 0, M, N, listed, wm, j & 1, l31, h);
    return;
  }
  __syncthreads();                                               // the last slab's fragments have left LDS in every wave
  float (*Ct)[64][LDT] = reinterpret_cast<float (*)[64][LDT]>(smem);
#pragma unroll
  for (int j = 0; j < 4; ++j) epi_stage(Ct[mi * 2 + (j >> 1)], acc[j], wm, j & 1, l31, h);
  __syncthreads();
#pragma unroll 1
  for (int q = 0; q < 4; ++q) {
    const int mq = m0 + 64 * (q >> 1), nq = n0 + 64 * (q & 1);
    if (mq < M && nq < N) epi_full<0, IDX>(P, Ct[q], mq, nq, 2 * tm + (q >> 1), 0, M, N, listed, tid);
  }
}

// ---- the weight planes
struct WPlaneJob { const float* w; int rows, cols; uint16_t* nrm; uint16_t* trn; int first_block; };
struct WPlaneJobs { WPlaneJob j[PS_WPLANES_MAX]; int n; };
// one thread per 8-element chunk of an output plane row: first the rows x cols / 8 chunks of the normal orientation, then the
// cols x rows / 8 chunks of the transposed one (strided reads of a matrix that lives in L2)
__global__ __launch_bounds__(256) void wplanes_kernel(const WPlaneJobs J) {
  int ji = 0;
#pragma unroll
  for (int i = 1; i < PS_WPLANES_MAX; ++i)
    if (i < J.n && (int)blockIdx.x >= J.j[i].first_block) ji = i;
  const WPlaneJob& jb = J.j[ji];
  const int rows = jb.rows, cols = jb.cols;
  const int64_t c = (int64_t)(blockIdx.x - jb.first_block) * 256 + threadIdx.x;
  const int64_t n_nrm = (int64_t)rows * (cols >> 3), n_trn = (int64_t)cols * (rows >> 3);
  if (c >= n_nrm + n_trn) return;
  float v[8];
  uint16_t* dst;
  size_t pstride = (size_t)rows * cols;
  if (c < n_nrm) {
    const int r = (int)(c / (cols >> 3)), kc = (int)(c - (int64_t)r * (cols >> 3));
    const float4 v0 = *reinterpret_cast<const float4*>(jb.w + (size_t)r * cols + 8 * kc);
    const float4 v1 = *reinterpret_cast<const float4*>(jb.w + (size_t)r * cols + 8 * kc + 4);
    v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
    dst = jb.nrm + (size_t)r * cols + 8 * kc;
  } else {
    const int64_t t = c - n_nrm;
    const int cc = (int)(t / (rows >> 3)), rc = (int)(t - (int64_t)cc * (rows >> 3));
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = jb.w[(size_t)(8 * rc + j) * cols + cc];
    dst = jb.trn + (size_t)cc * rows + 8 * rc;
  }
  bf16x8 pl[3];
  x3d_split8(v, pl);
#pragma unroll
  for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(dst + p * pstride) = pl[p];
}

struct WPlaneEntry { const float* w; int rows, cols; const uint16_t* nrm; const uint16_t* trn; };
static thread_local WPlaneEntry g_wplanes[PS_WPLANES_MAX];
static thread_local int g_wplanes_n = 0;

WPlaneScope::WPlaneScope(hipStream_t st, const float* const* w, const int* rows, const int* cols, int n) : on(false) {
  g_wplanes_n = 0;
  if (!gemm_x3w_on() || n <= 0) return;
  WPlaneJobs J;
  memset(&J, 0, sizeof(J));
  size_t elems = 0;
  int blocks = 0, k = 0;
  for (int i = 0; i < n && k < PS_WPLANES_MAX; ++i) {
    if (!w[i] || rows[i] % 32 || cols[i] % 32 || ((uintptr_t)w[i] & 15)) continue;
    bool dup = false;
    for (int q = 0; q < k; ++q) dup = dup || J.j[q].w == w[i];
    if (dup) continue;
    J.j[k].w = w[i]; J.j[k].rows = rows[i]; J.j[k].cols = cols[i]; J.j[k].first_block = blocks;
    blocks += ps_cdiv((int64_t)2 * rows[i] * (cols[i] >> 3), 256);
    elems += (size_t)rows[i] * cols[i];
    ++k;
  }
  if (!k) return;
  uint16_t* base = reinterpret_cast<uint16_t*>(ps_det_scratch(2, (elems * 2 * 3 * 2 + 3) / 4 + 64, st));
  if (!base) return;                                             // stream capture or out of memory: the products split B themselves
  size_t off = 0;
  for (int i = 0; i < k; ++i) {
    const size_t e = (size_t)J.j[i].rows * J.j[i].cols;
    J.j[i].nrm = base + off; off += 3 * e;
    J.j[i].trn = base + off; off += 3 * e;
    g_wplanes[i] = WPlaneEntry{J.j[i].w, J.j[i].rows, J.j[i].cols, J.j[i].nrm, J.j[i].trn};
  }
  J.n = k;
  hipLaunchKernelGGL(wplanes_kernel, dim3(blocks), dim3(256), 0, st, J);
  if (hipGetLastError() != hipSuccess) return;
  g_wplanes_n = k;
  on = true;
}
WPlaneScope::~WPlaneScope() { g_wplanes_n = 0; }

// B segments of `p` as plane matrices [3][N][kseg]; false: not (all) registered, or a shape the kernel does not take
static bool wplanes_for(GemmProblem& p) {
  if (g_wplanes_n == 0 || p.ta || p.K % X3D_BK || p.kseg % X3D_BK || ((uintptr_t)p.A & 15) || p.lda % 4) return false;
  const int nseg = ps_cdiv(p.K, p.kseg), ks = nseg > 1 ? p.kseg : p.K;
  for (int sg = 0; sg < nseg; ++sg) {
    const uint16_t* found = nullptr;
    for (int i = 0; i < g_wplanes_n; ++i) {
      const WPlaneEntry& e = g_wplanes[i];
      if (e.w != p.Bseg[sg]) continue;
      if (!p.tb && e.rows == p.N && e.cols == ks && p.ldb == e.cols) found = e.nrm;       // B[n][k] = W[n][k]
      if (p.tb && e.cols == p.N && e.rows == ks && p.ldb == e.cols) found = e.trn;        // B[k][n] = W[k][n]: planes of W^T
    }
    if (!found) return false;
    p.bpl[sg] = found;
  }
  return true;
}

static bool needs_full(const GemmProblem& p) {
  return p.act != ACT_NONE || p.drop.thr != 0u || p.res.mode != RES_NONE || p.aux_out || p.out2 || p.colsum;
}

static int validate(const GemmProblem& p) {
  PS_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem %d %d %d", p.M, p.N, p.K);
  PS_REQUIRE(p.lda % 4 == 0 && p.ldb % 4 == 0, "gemm: lda/ldb must be multiples of 4 (%d,%d)", p.lda, p.ldb);
  PS_REQUIRE(((uintptr_t)p.A & 15) == 0, "gemm: A not 16-byte aligned");
  if (p.ta) PS_REQUIRE(p.M % 4 == 0, "gemm: ta needs M %% 4 == 0 (%d)", p.M);
  else PS_REQUIRE(p.K % 4 == 0, "gemm: K %% 4 != 0 (%d)", p.K);
  if (p.tb) PS_REQUIRE(p.N % 4 == 0, "gemm: tb needs N %% 4 == 0 (%d)", p.N);
  else PS_REQUIRE(p.K % 4 == 0, "gemm: K %% 4 != 0 (%d)", p.K);
  int nseg = (p.K + p.kseg - 1) / p.kseg;
  PS_REQUIRE(p.kseg > 0 && nseg <= 3, "gemm: bad segments kseg=%d K=%d", p.kseg, p.K);
  if (nseg > 1) PS_REQUIRE(p.kseg % 4 == 0, "gemm: kseg %% 4 != 0 (%d)", p.kseg);
  for (int s = 0; s < nseg; ++s)
    PS_REQUIRE(p.Bseg[s] && ((uintptr_t)p.Bseg[s] & 15) == 0, "gemm: B segment %d null/unaligned", s);
  PS_REQUIRE(p.ksplit >= 1, "gemm: ksplit");
  if (p.ksplit > 1) PS_REQUIRE((p.accumulate == 2 || (p.accumulate == 0 && p.split_stride > 0)) && !needs_full(p),
                               "gemm: split reduction needs a plain atomic epilogue (or per-split outputs)");
  return PS_OK;
}

template <int FULL, int BK, int PF>
static void launch_pf(int ta, int tb, dim3 grid, hipStream_t stream, const GemmGroup& g) {
  if (ta == 0 && tb == 0) hipLaunchKernelGGL((gemm_f32_kernel<0, 0, FULL, BK, PF>), grid, dim3(256), 0, stream, g);
  else if (ta == 0 && tb == 1) hipLaunchKernelGGL((gemm_f32_kernel<0, 1, FULL, BK, PF>), grid, dim3(256), 0, stream, g);
  else if (ta == 1 && tb == 1) hipLaunchKernelGGL((gemm_f32_kernel<1, 1, FULL, BK, PF>), grid, dim3(256), 0, stream, g);
  else hipLaunchKernelGGL((gemm_f32_kernel<1, 0, FULL, BK, PF>), grid, dim3(256), 0, stream, g);
}
template <int FULL, int BK>
static void launch(int ta, int tb, dim3 grid, hipStream_t stream, const GemmGroup& g) {
  // two 32-deep slabs in flight (measured on MI355X: 1 -> 2 takes the 8064x128x512 product from 24.4 to 20.5 us and
  // 4096^3 from 104 to 112 TFLOP/s; 3 and 4 give nothing more); a 128-deep slab is a whole reduction already
  static const int wg_pf = ps_diag_int("PS_WGRAD_PF", 2);   // tuning experiments
  if (BK == 32 && !FULL && ta == 1 && tb == 1 && wg_pf == 4) {
    hipLaunchKernelGGL((gemm_f32_kernel<1, 1, 0, 32, 4>), grid, dim3(256), 0, stream, g);
    return;
  }
  if (BK == 32 && !FULL && ta == 1 && tb == 1 && wg_pf == 3) {
    hipLaunchKernelGGL((gemm_f32_kernel<1, 1, 0, 32, 3>), grid, dim3(256), 0, stream, g);
    return;
  }
  launch_pf<FULL, BK, (BK == 32 ? 2 : 1)>(ta, tb, grid, stream, g);
}

// ---- bf16x3 form: which launches take it, and with which tile
// PS_GEMM_X3: 0 = never, 1 (default) = wide products with enough tiles to fill the chip.  shape 2 = 128x128, 1 = 128x64,
// 0 = 64x64 tiles; -1 = the fp32 kernel (small / latency-bound launches, where the deep-slab forms above matter more).
// Measured at the d = 256 shard step (21,504 rows): 128x64 wins on every forward / dX product (128x128: 2 workgroups per
// CU, 1.53 vs 1.41 ms per step), the weight gradients take 64x64 with the split counts tem.hip picks.
static int g_x3_mode = -2, g_x3_force = -1;           // -2: not read yet
extern "C" int ps_gemm_x3_config(int mode, int force_shape) {   // tests / experiments: mode 0|1, force_shape -1 (rule) | 0 | 1 | 2
  PS_REQUIRE((mode == 0 || mode == 1) && force_shape >= -1 && force_shape <= 4, "gemm x3 config %d %d", mode, force_shape);
  g_x3_mode = mode; g_x3_force = force_shape;
  return PS_OK;
}
static bool x3_on() {
  if (g_x3_mode == -2) { g_x3_mode = ps_env_int("PS_GEMM_X3", 1) ? 1 : 0; g_x3_force = ps_env_int("PS_GEMM_X3_SHAPE", -1); }
  return g_x3_mode != 0;
}
bool gemm_x3_on() { return x3_on(); }
static int x3_shape(const GemmGroup& g, int maxM, int maxN) {
  x3_on();
  static const int t22 = ps_diag_int("PS_GEMM_X3_T22", 4096), t21 = ps_diag_int("PS_GEMM_X3_T21", 384), t11 = ps_diag_int("PS_GEMM_X3_T11", 512);
  if (!g_x3_mode) return -1;
  if (g_x3_force >= 0 && g_x3_force != 4) return g_x3_force > 3 ? 3 : g_x3_force;   // (4: pre-split weights where registered — the caller's business —, the rule elsewhere)
  const int z = g.n * g.p[0].ksplit;
  int kmin = g.p[0].K;
  for (int i = 1; i < g.n; ++i) kmin = g.p[i].K < kmin ? g.p[i].K : kmin;
  // Products of a wide model (every weight dimension >= 256: the d = 256 configuration) — at d = 128 a reduction is 4 slabs
  // deep and the fp32 kernel's deep-slab forms win at C2's 8k rows (0.289 -> 0.293 ms per step with this form forced); the
  // review transformer's 78k-row launches lose with 128x64 tiles and with their weight gradients included (0.540 -> 0.550) ...
  int nmin = g.p[0].N, mmin = g.p[0].M;
  for (int i = 1; i < g.n; ++i) { nmin = g.p[i].N < nmin ? g.p[i].N : nmin; mmin = g.p[i].M < mmin ? g.p[i].M : mmin; }
  // ... and the forward / dX products over very many rows (the review transformer's 78k sequence positions), with 64x64 tiles:
  // 0.544 -> 0.534 ms per step there (its weight gradients gain nothing: PS_GEMM_X3_TALL=2 adds them)
  static const int tall = ps_diag_int("PS_GEMM_X3_TALL", 1);
  if (tall >= 1 && !g.p[0].ta && maxM >= 32768) return 0;
  if (tall >= 2 && g.p[0].ta && kmin >= 32768) return 0;
  if (kmin < 256 || nmin < 256 || kmin / g.p[0].ksplit < 256 || (g.p[0].ta && mmin < 256)) return -1;
  if ((long)ps_cdiv(maxM, 128) * ps_cdiv(maxN, 128) * z >= t22) return 2;
  if ((long)ps_cdiv(maxM, 128) * ps_cdiv(maxN, 64) * z >= t21) return 1;
  if ((long)ps_cdiv(maxM, 64) * ps_cdiv(maxN, 64) * z >= t11) return 0;
  return -1;
}
// (one 32-deep slab of operands in flight in registers: two — the PF = 2 instantiation, 36 more registers — measured the same
// in the d = 256 step, 1.41-1.43 ms either way, and 5 % slower alone: the workgroups' own count hides the latency)
template <int TA, int TB, int FULL, int IDX>
static void launch_x3(int shape, dim3 grid, hipStream_t stream, const GemmGroup& g) {
  if (shape == 2) hipLaunchKernelGGL((gemm_x3_kernel<TA, TB, FULL, 2, 2, IDX, 1>), grid, dim3(256), 0, stream, g);
  else if (shape == 1) hipLaunchKernelGGL((gemm_x3_kernel<TA, TB, FULL, 2, 1, IDX, 1>), grid, dim3(256), 0, stream, g);
  else hipLaunchKernelGGL((gemm_x3_kernel<TA, TB, FULL, 1, 1, IDX, 1>), grid, dim3(256), 0, stream, g);
}
// flat weight-gradient groups: which tile (false: 64x64 of gemm_x3_kernel, true: 128x128 of gemm_x3d_kernel)
static bool x3_flat_d(const GemmGroup& g) {
  x3_on();
  (void)g;
  static const int diag = ps_diag_int("PS_X3_FLAT_D", 0);
  return g_x3_force == 3 || diag != 0;
}
// the direct-to-LDS form (gemm_x3d_kernel) takes single-segment operands and whole 32-deep slabs (a listed weight gradient pads
// its reduction list itself)
static bool x3d_takes(const GemmGroup& g) {
  for (int i = 0; i < g.n; ++i) {
    const GemmProblem& p = g.p[i];
    if (p.K > p.kseg) return false;
    if (p.K % X3D_BK != 0 && !(p.ridx && p.ta)) return false;
    if (((uintptr_t)p.Bseg[0] & 15) || ((uintptr_t)p.A & 15)) return false;
  }
  return true;
}
template <int TA, int TB, int FULL, int IDX>
static void launch_x3d(dim3 grid, hipStream_t stream, const GemmGroup& g) {
  hipLaunchKernelGGL((gemm_x3d_kernel<TA, TB, FULL, IDX>), grid, dim3(256), 0, stream, g);
}
static bool try_x3d(int ta, int tb, bool full, bool listed, int maxM, int maxN, int z, hipStream_t stream, const GemmGroup& g) {
  if (!x3d_takes(g)) return false;
  const dim3 grid(ps_cdiv(maxN, X3D_T), ps_cdiv(maxM, X3D_T), z);
  if (grid.y > 65535) return false;
#ifdef PS_DIAG
  static const int diag = ps_diag_int("PS_X3D_DIAG", 0), pad = ps_diag_int("PS_X3D_LDSPAD", 0);   // pad: dynamic LDS bytes (96 KB total = one workgroup per CU)
  if (!listed && ta == 0 && tb == 0 && !full && (diag || pad)) {
    if (diag == 1) hipLaunchKernelGGL((gemm_x3d_kernel<0, 0, 0, 0, 1>), grid, dim3(256), pad, stream, g);
    else if (diag == 2) hipLaunchKernelGGL((gemm_x3d_kernel<0, 0, 0, 0, 2>), grid, dim3(256), pad, stream, g);
    else if (diag == 3) hipLaunchKernelGGL((gemm_x3d_kernel<0, 0, 0, 0, 3>), grid, dim3(256), pad, stream, g);
    else hipLaunchKernelGGL((gemm_x3d_kernel<0, 0, 0, 0, 0>), grid, dim3(256), pad, stream, g);
    return true;
  }
#endif
  if (!listed) {
    if (ta == 0 && tb == 0 && full) launch_x3d<0, 0, 1, 0>(grid, stream, g);
    else if (ta == 0 && tb == 0) launch_x3d<0, 0, 0, 0>(grid, stream, g);
    else if (ta == 0 && tb == 1 && full) launch_x3d<0, 1, 1, 0>(grid, stream, g);
    else if (ta == 0 && tb == 1) launch_x3d<0, 1, 0, 0>(grid, stream, g);
    else if (ta == 1 && tb == 1 && !full) launch_x3d<1, 1, 0, 0>(grid, stream, g);
    else return false;
  } else {
    if (ta == 0 && tb == 1 && full) launch_x3d<0, 1, 1, 1>(grid, stream, g);
    else if (ta == 1 && tb == 1 && !full) launch_x3d<1, 1, 0, 1>(grid, stream, g);
    else if (ta == 0 && tb == 0 && !full) launch_x3d<0, 0, 0, 1>(grid, stream, g);
    else return false;
  }
  return true;
}
// false: no instantiation for this combination (the caller falls back to the fp32 kernel)
static bool try_x3(int ta, int tb, bool full, bool listed, int shape, int maxM, int maxN, int z, hipStream_t stream, const GemmGroup& g) {
  if (shape == 3) {
    if (try_x3d(ta, tb, full, listed, maxM, maxN, z, stream, g)) return true;
    shape = 1;                                                   // what the rule picked before this form existed
  }
  const dim3 grid(ps_cdiv(maxN, shape == 2 ? 128 : 64), ps_cdiv(maxM, shape >= 1 ? 128 : 64), z);
  if (grid.y > 65535) return false;
  if (!listed) {
    if (ta == 0 && tb == 0 && full) launch_x3<0, 0, 1, 0>(shape, grid, stream, g);
    else if (ta == 0 && tb == 0) launch_x3<0, 0, 0, 0>(shape, grid, stream, g);
    else if (ta == 0 && tb == 1 && full) launch_x3<0, 1, 1, 0>(shape, grid, stream, g);
    else if (ta == 0 && tb == 1) launch_x3<0, 1, 0, 0>(shape, grid, stream, g);
    else if (ta == 1 && tb == 1 && !full) launch_x3<1, 1, 0, 0>(shape, grid, stream, g);
    else return false;
  } else {
    if (ta == 0 && tb == 1 && full) launch_x3<0, 1, 1, 1>(shape, grid, stream, g);
    else if (ta == 1 && tb == 1 && !full) launch_x3<1, 1, 0, 1>(shape, grid, stream, g);
    else if (ta == 0 && tb == 0 && !full) launch_x3<0, 0, 0, 1>(shape, grid, stream, g);
    else return false;
  }
  return true;
}

// the pre-split-weight form is OFF unless asked for (ps_gemm_x3_config(1, 4) / PS_GEMM_X3_SHAPE=4): measured on MI355X it ties
// the 128x64 kernel on the d = 256 step's products (profiles/r04_gemm_notes.md) — kept as a tested alternative, not the default
bool gemm_x3w_on() { x3_on(); return g_x3_mode != 0 && g_x3_force == 4; }
static int x3w_min_rows() { static const int v = ps_diag_int("PS_GEMM_X3W_MIN_ROWS", 4096); return v; }
static int launch_gemm_impl(const GemmGroup& g, hipStream_t stream);
int ps_launch_gemm(const GemmGroup& g0, hipStream_t stream) {
  GemmGroup g = g0;
  g.sig = nullptr; g.sigval = 0;
  {   // split reductions on the 3-D grid (row-list weight gradients keep it): splits placed by XCD when their count allows
    static const int split_xcd = ps_diag_int("PS_SPLIT_XCD", 1);
    const int ks = g.p[0].ksplit;
    g.split_xcd = split_xcd && !g.flat && g.p[0].ta == 1 && ks >= 8 && ks % 8 == 0;
  }
  static const int stamps = ps_diag_int("PS_GEMM_STAMP", 0);
  const bool stamp_this = stamps == 1 || (stamps == 2 && g0.p[0].ridx && g0.p[0].res.mode == RES_FANIN) ||
                          (stamps == 3 && g0.p[0].ridx && !g0.p[0].ta && g0.p[0].res.mode == RES_NONE) ||
                          (stamps == 4 && g0.n == 3 && g0.p[0].ta && !g0.p[0].ridx);      // 4: the grouped FF / Wo weight gradients
  g.stamp = stamp_this ? ps_debug_stamp_ptr() : nullptr;
  const bool took = side_take_signal(stream, &g.sig, &g.sigval);      // a pending fork of the side stream rides on this launch
  const int rc = launch_gemm_impl(g, stream);
  if (took && rc != PS_OK) side_repend_signal(stream, g.sigval);
  return rc;
}
static int launch_gemm_impl(const GemmGroup& g, hipStream_t stream) {
  PS_REQUIRE(g.n >= 1 && g.n <= (g.flat ? 3 : 4), "gemm: group size %d", g.n);
  if (g.flat) {
    for (int i = 0; i < g.n; ++i) {
      int rc = validate(g.p[i]);
      if (rc) return rc;
      PS_REQUIRE(g.p[i].ta == 1 && g.p[i].tb == 1 && !needs_full(g.p[i]) && !g.p[i].ridx && g.p[i].ksplit >= 1 &&
                 (g.p[i].accumulate == 2 || (g.p[i].accumulate == 0 && g.p[i].split_stride > 0)),
                 "gemm: the flat group form takes plain weight-gradient problems");
    }
    // the bf16x3 forms (the same flat tables): C2's W2 / W1 / Wo weight gradients 0.2905 -> 0.2845 ms per step with 64x64 tiles
    // (round 2); 128x128 tiles of the direct-to-LDS form where x3_flat_shape() says so
    static const int flat_x3 = ps_diag_int("PS_GEMM_X3_FLAT", 1);
    const bool x3 = flat_x3 && x3_on();
    const bool x3d = x3 && (x3_flat_d(g) || g.prefer_x3d) && x3d_takes(g);
    static const int flat_shape = ps_diag_int("PS_X3_FLAT_SHAPE", 0);      // 1: 128x64, 2: 128x128 tiles of gemm_x3_kernel
    const int fs = (x3 && !x3d) ? flat_shape : 0;
    const int TM = x3d ? X3D_T : (fs >= 1 ? 128 : BM), TN = x3d ? X3D_T : (fs == 2 ? 128 : BM);
    GemmGroup f = g;
    int total = 0;
    for (int i = 0; i < 3; ++i) {
      f.flat0[i] = total;
      if (i >= g.n) { f.flat_tm[i] = 1; f.flat_tiles[i] = 1; continue; }
      f.flat_tm[i] = ps_cdiv(g.p[i].M, TM);
      f.flat_tiles[i] = f.flat_tm[i] * ps_cdiv(g.p[i].N, TN);
      total += f.flat_tiles[i] * g.p[i].ksplit;
    }
    f.flat0[3] = total;
    for (int i = g.n; i < 3; ++i) f.flat0[i] = total + 1;      // never selected
    static const int flat_xcd = ps_diag_int("PS_FLAT_XCD", 1);
    f.flat_xcd = flat_xcd;
    for (int i = 0; i < g.n; ++i)
      if (g.p[i].ksplit % 8 != 0 || f.flat0[i] % 8 != 0) f.flat_xcd = 0;
    if (x3d) PS_KLAUNCH((gemm_x3d_kernel<1, 1, 0, 0>), dim3(total, 1, 1), dim3(256), 0, stream, f);
    else if (x3 && fs == 2) PS_KLAUNCH((gemm_x3_kernel<1, 1, 0, 2, 2, 0, 1>), dim3(total, 1, 1), dim3(256), 0, stream, f);
    else if (x3 && fs == 1) PS_KLAUNCH((gemm_x3_kernel<1, 1, 0, 2, 1, 0, 1>), dim3(total, 1, 1), dim3(256), 0, stream, f);
    else if (x3) PS_KLAUNCH((gemm_x3_kernel<1, 1, 0, 1, 1, 0, 1>), dim3(total, 1, 1), dim3(256), 0, stream, f);
    else launch<0, 32>(1, 1, dim3(total, 1, 1), stream, f);
    PS_LAUNCH_CHECK();
    return PS_OK;
  }
  int maxM = 0, maxN = 0;
  bool full = false;
  for (int i = 0; i < g.n; ++i) {
    int rc = validate(g.p[i]);
    if (rc) return rc;
    PS_REQUIRE(g.p[i].ta == g.p[0].ta && g.p[i].tb == g.p[0].tb && g.p[i].ksplit == g.p[0].ksplit,
               "gemm: group members must share layout and ksplit");
    maxM = g.p[i].M > maxM ? g.p[i].M : maxM;
    maxN = g.p[i].N > maxN ? g.p[i].N : maxN;
    full = full || needs_full(g.p[i]);
  }
  dim3 grid(ps_cdiv(maxN, BN), ps_cdiv(maxM, BM), g.n * g.p[0].ksplit);
  PS_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "gemm: grid too large");
  bool listed = false;
  for (int i = 0; i < g.n; ++i) {
    const GemmProblem& p = g.p[i];
    if (!p.ridx) continue;
    listed = true;
    PS_REQUIRE(p.rcount && p.drop.thr == 0u && !p.colsum && !p.aux_out && p.kseg >= p.K * (p.ta ? 1 : 0),
               "gemm: row-list problems take no dropout / column sums / pre-activation output");
    if (p.ta) {
      const int nslab = ps_cdiv(p.K, 32), per = ps_cdiv(nslab, p.ksplit);
      PS_REQUIRE(p.tb == 1 && per * 32 <= KIDX_MAX, "gemm: row-list weight gradient: %d rows per split > %d", per * 32, KIDX_MAX);
    }
  }
  // Plain split reductions (weight gradients) whose form is a bf16x3 kernel that also has a FLAT (1-D grid) launch go through it: the
  // flat decode puts all tiles of a reduction split on one XCD (GemmGroup::flat_xcd), where this 3-D grid spreads them over all
  // eight — the C5 shard's W2 / W1 gradients fetched 2 x 129 = 258 MB per launch for 110 MB of operands, its Wo gradient 2 x 68 = 135
  // MB for 44 (PMC, profiles/r04_c5_pmc_traffic.txt)
  {
    static const int flat_all = ps_diag_int("PS_WGRAD_FLAT_ALL", 1);
    const int ks = g.p[0].ksplit;
    bool plain = flat_all && !listed && !full && g.p[0].ta == 1 && g.p[0].tb == 1 && ks >= 8 && ks % 8 == 0 && x3_on();
    for (int i = 0; i < g.n && plain; ++i)
      plain = g.p[i].accumulate == 2 || (g.p[i].accumulate == 0 && g.p[i].split_stride > 0);
    if (plain) {
      const bool d = g.prefer_x3d && (g_x3_force < 0 || g_x3_force == 4) && x3d_takes(g);
      if (d || x3_shape(g, maxM, maxN) == 0) {
        GemmGroup f = g;
        f.flat = 1;
        f.prefer_x3d = d ? 1 : 0;
        return launch_gemm_impl(f, stream);
      }
    }
  }
  // products against pre-split weights (WPlaneScope): every member's B found, enough rows to be worth a 128-row tile
  if (gemm_x3w_on() && !g.p[0].ta && g.p[0].ksplit == 1 && maxM >= x3w_min_rows()) {
    GemmGroup gw = g;
    bool all = true;
    for (int i = 0; i < gw.n; ++i) all = all && wplanes_for(gw.p[i]);
    const dim3 wgrid(ps_cdiv(maxN, X3D_T), ps_cdiv(maxM, X3D_T), gw.n);
    if (all && wgrid.y <= 65535) {
      for (int i = 0; i < gw.n; ++i)
        if (ps_cdiv(gw.p[i].K, gw.p[i].kseg) == 1) gw.p[i].kseg = gw.p[i].K;      // one segment: its planes are [3][N][K]
      if (full && listed) hipLaunchKernelGGL((gemm_x3w_kernel<1, 1>), wgrid, dim3(256), 0, stream, gw);
      else if (full) hipLaunchKernelGGL((gemm_x3w_kernel<1, 0>), wgrid, dim3(256), 0, stream, gw);
      else if (listed) hipLaunchKernelGGL((gemm_x3w_kernel<0, 1>), wgrid, dim3(256), 0, stream, gw);
      else hipLaunchKernelGGL((gemm_x3w_kernel<0, 0>), wgrid, dim3(256), 0, stream, gw);
      PS_LAUNCH_CHECK();
      return PS_OK;
    }
  }
  if (g.prefer_x3d && x3_on() && (g_x3_force < 0 || g_x3_force == 4) &&
      try_x3d(g.p[0].ta, g.p[0].tb, full, listed, maxM, maxN, g.n * g.p[0].ksplit, stream, g)) {
    PS_LAUNCH_CHECK();
    return PS_OK;
  }
  const int x3 = x3_shape(g, maxM, maxN);
  if (x3 >= 0 && try_x3(g.p[0].ta, g.p[0].tb, full, listed, x3, maxM, maxN, g.n * g.p[0].ksplit, stream, g)) {
    PS_LAUNCH_CHECK();
    return PS_OK;
  }
  if (listed) {   // row-list instantiations exist for the three shapes of the step that use them (32-deep slabs)
    const int ta = g.p[0].ta, tb = g.p[0].tb;
    // (the K/V dX product as 128-deep slabs measured slower, 0.363 vs 0.355 ms/step: its 133 KB of LDS per workgroup
    // crowds out the weight gradients running beside it)
    if (ta == 0 && tb == 1 && full) hipLaunchKernelGGL((gemm_f32_kernel<0, 1, 1, 32, 2, 1>), grid, dim3(256), 0, stream, g);
    else if (ta == 1 && tb == 1 && !full) hipLaunchKernelGGL((gemm_f32_kernel<1, 1, 0, 32, 2, 1>), grid, dim3(256), 0, stream, g);
    else if (ta == 0 && tb == 0 && !full) {
      // small launches whose whole reduction is one 128-deep slab (the K/V projection at C2: ~240 live workgroups):
      // one global round trip instead of four shallow slabs (0.3615 -> 0.3564 ms/step); big grids keep 4 workgroups per CU
      static const int deep_kv = ps_diag_int("PS_KV_DEEP", 1);   // tuning experiment
      if (deep_kv && g.p[0].K <= 128 && (size_t)grid.x * grid.y * grid.z <= 1024)
        hipLaunchKernelGGL((gemm_f32_kernel<0, 0, 0, 128, 1, 1>), grid, dim3(256), 0, stream, g);
      else hipLaunchKernelGGL((gemm_f32_kernel<0, 0, 0, 32, 2, 1>), grid, dim3(256), 0, stream, g);
    }
    else PS_REQUIRE(false, "gemm: no row-list instantiation for ta=%d tb=%d full=%d", ta, tb, (int)full);
    PS_LAUNCH_CHECK();
    return PS_OK;
  }
  static const int repeat = ps_diag_int("PS_DEBUG_REPEAT", 1);   // timing experiments only
  for (int r = 0; r < repeat; ++r) {
    // few workgroups (latency-bound chain): one deep slab per round trip; many: shallow slabs, 4 wgs per CU
    static const int deep_max = ps_diag_int("PS_GEMM_DEEP_MAX", 64);         // tuning experiments
    const bool deep = (size_t)grid.x * grid.y * grid.z <= (size_t)deep_max && !g.p[0].no_deep;
    if (full && deep) launch<1, 128>(g.p[0].ta, g.p[0].tb, grid, stream, g);
    else if (full) launch<1, 32>(g.p[0].ta, g.p[0].tb, grid, stream, g);
    else if (deep) launch<0, 128>(g.p[0].ta, g.p[0].tb, grid, stream, g);
    else launch<0, 32>(g.p[0].ta, g.p[0].tb, grid, stream, g);
  }
  PS_LAUNCH_CHECK();
  return PS_OK;
}

static int gemm_f32_impl(const float* A, int lda, int ta, const float* Bm, int ldb, int tb, float* Cm, int ldc,
                         int M, int N, int K, const float* bias, float alpha, int accumulate, ps_stream_t stream);
extern "C" int ps_gemm_f32(const float* A, int lda, int ta, const float* Bm, int ldb, int tb, float* Cm, int ldc,
                           int M, int N, int K, const float* bias, float alpha, int accumulate,
                           ps_stream_t stream) {
  return gemm_f32_impl(A, lda, ta, Bm, ldb, tb, Cm, ldc, M, N, K, bias, alpha, accumulate, stream);
}
// the same product with B declared a WEIGHT ([N][K] for tb == 0, [K][N] for tb == 1, dense rows): split once into bf16 planes
// for the duration of the call (WPlaneScope), multiplied by gemm_x3w_kernel where that kernel applies — the path the
// training step's forward and dX products of a d >= 256 model take; tests and tools/gemm_x3_bench.py
extern "C" int ps_gemm_f32_weight(const float* A, int lda, const float* W, int tb, float* Cm, int ldc, int M, int N, int K,
                                  const float* bias, float alpha, ps_stream_t stream) {
  const float* ws[1] = {W};
  const int rows[1] = {tb ? K : N}, cols[1] = {tb ? N : K};
  WPlaneScope scope((hipStream_t)stream, ws, rows, cols, 1);
  return gemm_f32_impl(A, lda, 0, W, tb ? N : K, tb, Cm, ldc, M, N, K, bias, alpha, 0, stream);
}
static int gemm_f32_impl(const float* A, int lda, int ta, const float* Bm, int ldb, int tb, float* Cm, int ldc,
                         int M, int N, int K, const float* bias, float alpha, int accumulate, ps_stream_t stream) {
  GemmGroup g = {};
  g.n = 1;
  GemmProblem& p = g.p[0];
  p.A = A; p.lda = lda; p.ta = ta;
  p.Bseg[0] = Bm; p.kseg = K; p.ldb = ldb; p.tb = tb;
  p.C = Cm; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  p.bias = bias; p.alpha = alpha; p.act = ACT_NONE;
  p.accumulate = accumulate == 2 ? 2 : accumulate;
  static const int ks = ps_diag_int("PS_GEMM_KSPLIT", 4);   // timing experiments
  p.ksplit = accumulate == 2 ? ks : 1;
  return ps_launch_gemm(g, (hipStream_t)stream);
}
