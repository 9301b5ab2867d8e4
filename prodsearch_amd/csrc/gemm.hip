// gemm.hip — exact-fp32 MFMA GEMM with fused epilogues (gfx950).
//
// The only dense contractions of the hot path are the transformer's linears
// (reference models/neural.py:86-96 K/V/Q/final_linear, :20-21 w_1/w_2, and
// text_encoder.py:24 f_W) and their two backward products.  They are computed
// with v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate = a k-ordered fmaf chain,
// so results are exact fp32 like the reference's mm), tiled for 64-wide waves:
//
//   workgroup = 4 waves, tile 64x64 (each wave one 32x32 accumulator = 16 VGPRs),
//   BK = 32 reduction slab, both operands staged global -> registers -> LDS
//   (double-buffered, one barrier per slab) in a [k][row+1] image so that every
//   MFMA operand fetch is one conflict-free ds_read_b32 per lane.
//
// One kernel serves forward (A[m][k] . W[n][k]), input-grad (dY[m][n] . W[n][k'])
// and weight-grad (dY[m][n]^T . X[m][k'], split over the row reduction with
// fp32 atomics) through (ta, tb) layouts; the epilogue fuses bias, scale,
// GELU/tanh (or their derivatives), Philox dropout, residual (direct / gathered
// from the layer input / fan-in summed over replicas), bias-grad column sums.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 64
#define BN 64
#define BK 32
#define LDT (BM + 1)

template <int TRANS>  // TRANS==0: src[row][k] (k contiguous) ; TRANS==1: src[k][row] (row contiguous)
__device__ inline void tile_load(const float* __restrict__ src, int ld, int row0, int nrows, int k0, int kend,
                                 float4 (&reg)[2], int tid) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    int f = tid + 256 * u;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (TRANS == 0) {
      int i = f >> 3, kq = f & 7;
      int r = row0 + i, k = k0 + kq * 4;
      if (r < nrows && k < kend) v = *reinterpret_cast<const float4*>(src + (size_t)r * ld + k);
    } else {
      int kk = f >> 4, iq = f & 15;
      int r = row0 + iq * 4, k = k0 + kk;
      if (r < nrows && k < kend) v = *reinterpret_cast<const float4*>(src + (size_t)k * ld + r);
    }
    reg[u] = v;
  }
}

template <int TRANS>
__device__ inline void tile_store(float (*T)[LDT], const float4 (&reg)[2], int tid) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    int f = tid + 256 * u;
    if (TRANS == 0) {
      int i = f >> 3, kq = f & 7;
      T[kq * 4 + 0][i] = reg[u].x;
      T[kq * 4 + 1][i] = reg[u].y;
      T[kq * 4 + 2][i] = reg[u].z;
      T[kq * 4 + 3][i] = reg[u].w;
    } else {
      int kk = f >> 4, iq = f & 15;
      T[kk][iq * 4 + 0] = reg[u].x;
      T[kk][iq * 4 + 1] = reg[u].y;
      T[kk][iq * 4 + 2] = reg[u].z;
      T[kk][iq * 4 + 3] = reg[u].w;
    }
  }
}

template <int TA, int TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmGroup g) {
  __shared__ float As[2][BK][LDT];
  __shared__ float Bs[2][BK][LDT];

  const int prob = blockIdx.z / g.p[0].ksplit;
  const int split = blockIdx.z - prob * g.p[0].ksplit;
  const GemmProblem& P = g.p[prob];
  const int M = P.M, N = P.N, K = P.K;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int nslab = (K + BK - 1) / BK;
  const int per = (nslab + P.ksplit - 1) / P.ksplit;
  const int kbeg = split * per * BK;
  const int kend = min(K, kbeg + per * BK);
  if (m0 >= M || n0 >= N || kbeg >= kend) return;   // block-uniform

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  float4 ra[2], rb[2];
  auto bptr = [&](int k) -> const float* {
    int s = k / P.kseg;
    const float* b = P.Bseg[s];
    // segment-local k offset
    return TB == 0 ? b - (size_t)s * P.kseg : b - (size_t)s * P.kseg * P.ldb;
  };

  tile_load<TA>(P.A, P.lda, m0, M, kbeg, kend, ra, tid);
  tile_load<TB>(bptr(kbeg), P.ldb, n0, N, kbeg, kend, rb, tid);
  tile_store<TA>(As[0], ra, tid);
  tile_store<TB>(Bs[0], rb, tid);
  __syncthreads();

  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const int kn = k0 + BK;
    const bool more = kn < kend;
    if (more) {
      tile_load<TA>(P.A, P.lda, m0, M, kn, kend, ra, tid);
      tile_load<TB>(bptr(kn), P.ldb, n0, N, kn, kend, rb, tid);
    }
    const float* a_base = &As[buf][h][wm * 32 + l31];
    const float* b_base = &Bs[buf][h][wn * 32 + l31];
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      float a = a_base[2 * s * LDT];
      float b = b_base[2 * s * LDT];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (more) {
      tile_store<TA>(As[buf ^ 1], ra, tid);
      tile_store<TB>(Bs[buf ^ 1], rb, tid);
    }
    __syncthreads();
    buf ^= 1;
  }

  // ------------------------------------------------------------------ epilogue
  const int col = n0 + wn * 32 + l31;
  const bool col_ok = col < N;
  const float bias = (P.bias && col_ok && split == 0) ? P.bias[col] : 0.f;
  float csum = 0.f;
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const int rbase = m0 + wm * 32 + 8 * gq + 4 * h;     // rows rbase..rbase+3 <-> regs 4gq..4gq+3
    Philox4 rnd;
    if (P.drop.thr != 0u)
      rnd = philox4x32_10((uint32_t)col, (uint32_t)rbase >> 2, P.drop.site, P.drop.step, P.drop.k0, P.drop.k1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = rbase + q;
      if (!(col_ok && row < M)) continue;
      float v = (acc[4 * gq + q] + bias) * P.alpha;
      const size_t off = (size_t)row * P.ldc + col;
      if (P.aux_out) P.aux_out[off] = v;
      if (P.act == ACT_GELU) v = gelu_tanh_f(v);
      else if (P.act == ACT_TANH) v = tanhf(v);
      else if (P.act == ACT_GELU_BWD) v *= gelu_tanh_grad(P.act_aux[off]);
      else if (P.act == ACT_TANH_BWD) { float y = P.act_aux[off]; v *= (1.f - y * y); }
      if (P.drop.thr != 0u) {
        uint32_t wv = q == 0 ? rnd.x : (q == 1 ? rnd.y : (q == 2 ? rnd.z : rnd.w));
        v *= drop_word(P.drop, wv);
      }
      if (P.res.mode != RES_NONE) v += res_value(P.res, row, col);
      csum += v;
      if (P.out2) P.out2[(size_t)row * P.ld2 + col] = v + (P.add2 ? P.add2[col] : 0.f);
      if (P.accumulate == 0) P.C[off] = v;
      else if (P.accumulate == 1) P.C[off] += v;
      else atomicAdd(&P.C[off], v);
    }
  }
  if (P.colsum) {
    csum += __shfl_xor(csum, 32, 64);
    if (h == 0 && col_ok) atomicAdd(&P.colsum[col], csum);
  }
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
  if (nseg > 1) PS_REQUIRE(p.kseg % BK == 0, "gemm: kseg %% %d != 0 (%d)", BK, p.kseg);
  for (int s = 0; s < nseg; ++s)
    PS_REQUIRE(p.Bseg[s] && ((uintptr_t)p.Bseg[s] & 15) == 0, "gemm: B segment %d null/unaligned", s);
  PS_REQUIRE(p.ksplit >= 1, "gemm: ksplit");
  if (p.ksplit > 1) PS_REQUIRE(p.accumulate == 2, "gemm: split reduction needs atomic accumulate");
  return PS_OK;
}

int ps_launch_gemm(const GemmGroup& g, hipStream_t stream) {
  PS_REQUIRE(g.n >= 1 && g.n <= 3, "gemm: group size %d", g.n);
  int maxM = 0, maxN = 0;
  for (int i = 0; i < g.n; ++i) {
    int rc = validate(g.p[i]);
    if (rc) return rc;
    PS_REQUIRE(g.p[i].ta == g.p[0].ta && g.p[i].tb == g.p[0].tb && g.p[i].ksplit == g.p[0].ksplit,
               "gemm: group members must share layout and ksplit");
    maxM = g.p[i].M > maxM ? g.p[i].M : maxM;
    maxN = g.p[i].N > maxN ? g.p[i].N : maxN;
  }
  dim3 grid(ps_cdiv(maxN, BN), ps_cdiv(maxM, BM), g.n * g.p[0].ksplit);
  PS_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "gemm: grid too large");
  const int ta = g.p[0].ta, tb = g.p[0].tb;
  if (ta == 0 && tb == 0) hipLaunchKernelGGL((gemm_f32_kernel<0, 0>), grid, dim3(256), 0, stream, g);
  else if (ta == 0 && tb == 1) hipLaunchKernelGGL((gemm_f32_kernel<0, 1>), grid, dim3(256), 0, stream, g);
  else if (ta == 1 && tb == 1) hipLaunchKernelGGL((gemm_f32_kernel<1, 1>), grid, dim3(256), 0, stream, g);
  else if (ta == 1 && tb == 0) hipLaunchKernelGGL((gemm_f32_kernel<1, 0>), grid, dim3(256), 0, stream, g);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

extern "C" int ps_gemm_f32(const float* A, int lda, int ta, const float* Bm, int ldb, int tb, float* Cm, int ldc,
                           int M, int N, int K, const float* bias, float alpha, int accumulate,
                           ps_stream_t stream) {
  GemmGroup g = {};
  g.n = 1;
  GemmProblem& p = g.p[0];
  p.A = A; p.lda = lda; p.ta = ta;
  p.Bseg[0] = Bm; p.kseg = K; p.ldb = ldb; p.tb = tb;
  p.C = Cm; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  p.bias = bias; p.alpha = alpha; p.act = ACT_NONE;
  p.accumulate = accumulate == 2 ? 2 : accumulate;
  p.ksplit = accumulate == 2 ? 4 : 1;
  return ps_launch_gemm(g, (hipStream_t)stream);
}
