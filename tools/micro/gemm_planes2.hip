// gemm_planes2.hip — prototype 2 (round 5): the bf16x3 product on BLOCKED plane operands with a 4-stage LDS-DMA pipeline.
//   C[M][N] = A[M][K] . B[N][K]^T ; every operand is three bf16 planes in the layout  P[plane][K/16][rows][16]  — the 16 k-values of
//   a row for one k16 block are one 32-byte piece, the pieces of consecutive rows are contiguous: a 256-row x 16-k tile of a plane is
//   ONE contiguous 8 KB run in memory (full-line LDS-DMA, no 64-byte row segments), and a producer epilogue whose lane holds 16
//   consecutive features of a row (mlp_fused.hip, gemm epilogues) writes exactly one piece per plane.
// Loop: tile 256 x 128, 8 waves (4 x 2, 64 x 64 each), K-step 16 (one 32x32x16 product step = 6 bf16 MFMAs per tile pair), FOUR 36 KB
// stages; the DMA of step k + 3 is issued right behind the barrier that retires step k - 1's reads, a counted vmcnt leaves two steps
// in flight across every barrier (never 0 inside the loop), fragments by inline-asm ds_read_b128 (hipcc would drain the DMA in front
// of a ds_read it can see).  gemm_planes.hip (two 72 KB stages, vmcnt(0) per 32-deep step) is the form this one is measured against.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm_planes2 gemm_planes2.hip && ./gemm_planes2 [variant]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

constexpr int BM = 256, BN = 128, NTHR = 512, NST = 4;
constexpr int A_PLANE = BM * 32, B_PLANE = BN * 32;                 // bytes per plane image of one k16 block: 8 KB, 4 KB
constexpr int STAGE = 3 * A_PLANE + 3 * B_PLANE;                    // 36 KB

// fp32 [rows][K] -> blocked planes P[plane][K/16][rows][16]
__global__ void split_blocked_kernel(const float* x, uint16_t* planes, int rows, int K) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)rows * K;
  if (i >= n) return;
  const int r = (int)(i / K), k = (int)(i - (size_t)r * K);
  const float v = x[i];
  const __bf16 h = (__bf16)v;
  float q = v - (float)h;
  const __bf16 m = (__bf16)q;
  q -= (float)m;
  const __bf16 l = (__bf16)q;
  const size_t o = ((size_t)(k >> 4) * rows + r) * 16 + (k & 15);
  planes[o] = __builtin_bit_cast(uint16_t, h);
  planes[n + o] = __builtin_bit_cast(uint16_t, m);
  planes[2 * n + o] = __builtin_bit_cast(uint16_t, l);
}

#define LDS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define BF(x) __builtin_bit_cast(bf16x8, x)

struct Src { const uint16_t* a; const uint16_t* b; size_t a_plane, b_plane; int M, N; };
// the DMA of one K-step: 2304 16-byte chunks, lane-linear in the stage image [A p0 | A p1 | A p2 | B p0 | B p1 | B p2];
// position (row, pos) of an image holds chunk pos ^ ((row >> 3) & 1) of the row's 32-byte piece (bank swizzle on the SOURCE side)
__device__ __forceinline__ void stage_issue(const Src& s, int m0, int n0, int kb, unsigned char* stage, int tid, int wave) {
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    if (r == 4 && wave >= 4) break;                                  // the ninth half round: waves 0-3 only (wave-uniform)
    const int q = r * NTHR + tid;
    const uint16_t* src;
    if (r < 3) {                                                     // A plane r: 512 chunks
      const int row = tid >> 1, pos = tid & 1, c = pos ^ ((row >> 3) & 1);
      src = s.a + r * s.a_plane + ((size_t)kb * s.M + m0 + row) * 16 + 8 * c;
    } else {                                                         // B planes: 256 chunks each
      const int w = q - 1536, plane = w >> 8, within = w & 255;
      const int row = within >> 1, pos = within & 1, c = pos ^ ((row >> 3) & 1);
      src = s.b + plane * s.b_plane + ((size_t)kb * s.N + n0 + row) * 16 + 8 * c;
    }
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(stage + (r * NTHR + wave * 64) * 16), 16, 0, 0);
  }
}

template <int DIAG>   // timing-only variants (WRONG results): 1 no DMA inside the loop, 2 no MFMAs
__global__ __launch_bounds__(NTHR) void gemm_planes2_kernel(const uint16_t* Ap, const uint16_t* Bp, float* C, int M, int N, int K, int ntn) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int L = blockIdx.x, nt = gridDim.x;
  int tile = L;
  if (nt % 8 == 0) tile = (L & 7) * (nt >> 3) + (L >> 3);            // XCD L % 8 walks a contiguous run of tiles (row-block major)
  const int tm = tile / ntn, tn = tile - tm * ntn;
  const int m0 = tm * BM, n0 = tn * BN;
  const Src s = {Ap, Bp, (size_t)M * K, (size_t)N * K, M, N};
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
  const uint32_t base = (uint32_t)(l31 * 32 + ((h ^ ((l31 >> 3) & 1)) << 4));
  const uint32_t aA0 = lds0 + base + wm * 2048, aB0 = lds0 + base + 3 * A_PLANE + wn * 2048;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int KT = K / 16;
  // prologue: three steps in flight
#pragma unroll
  for (int p = 0; p < 3; ++p)
    if (p < KT) stage_issue(s, m0, n0, p, smem + p * STAGE, tid, wave);
  for (int kt = 0; kt < KT; ++kt) {
    // step kt has landed when at most the two younger steps' pieces (5 or 4 each for this wave) are still in flight
    if (kt + 2 < KT) { if (wave < 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    else if (kt + 1 < KT) { if (wave < 4) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                    // everybody's pieces of step kt; everybody is done reading step kt - 1
    if (kt + 3 < KT && DIAG != 1) stage_issue(s, m0, n0, kt + 3, smem + ((kt + 3) & 3) * STAGE, tid, wave);   // into the stage step kt - 1 used
    const uint32_t so = (uint32_t)((kt & 3) * STAGE);
    const uint32_t aA = aA0 + so, aB = aB0 + so;
    i32x4 fa[2][3], fb[2][3];
    LDS_READ(fa[0][0], aA, 0);    LDS_READ(fa[0][1], aA, 8192);  LDS_READ(fa[0][2], aA, 16384);
    LDS_READ(fb[0][0], aB, 0);    LDS_READ(fb[0][1], aB, 4096);  LDS_READ(fb[0][2], aB, 8192);
    LDS_READ(fa[1][0], aA, 1024); LDS_READ(fa[1][1], aA, 9216);  LDS_READ(fa[1][2], aA, 17408);
    LDS_READ(fb[1][0], aB, 1024); LDS_READ(fb[1][1], aB, 5120);  LDS_READ(fb[1][2], aB, 9216);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (DIAG != 2) {
      __builtin_amdgcn_s_setprio(1);
      // six product terms, small ones first; consecutive MFMAs go to different accumulators
#define TERM(pa, pb)                                                                                                \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(fa[0][pa]), BF(fb[0][pb]), acc[0][0], 0, 0, 0);          \
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(fa[0][pa]), BF(fb[1][pb]), acc[0][1], 0, 0, 0);          \
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(fa[1][pa]), BF(fb[0][pb]), acc[1][0], 0, 0, 0);          \
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(fa[1][pa]), BF(fb[1][pb]), acc[1][1], 0, 0, 0);
      TERM(0, 2) TERM(2, 0) TERM(1, 1) TERM(0, 1) TERM(1, 0) TERM(0, 0)
#undef TERM
      __builtin_amdgcn_s_setprio(0);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) { asm volatile("" :: "v"(fa[i][p])); asm volatile("" :: "v"(fb[i][p])); }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int col = n0 + 64 * wn + 32 * j + l31;
        C[(size_t)row * N + col] = acc[i][j][r];
      }
}

int main(int argc, char** argv) {
  struct Shape { int M, N, K; const char* what; };
  const Shape shapes[] = {{21504, 1024, 256, "FF1 of the C5 shard"}, {21504, 256, 1024, "FF2"}, {4096, 4096, 4096, "4096^3"}, {8192, 8192, 1024, "8192 x 8192 x 1024"}};
  const int diag = argc > 1 ? atoi(argv[1]) : 0;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_planes2_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_planes2_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_planes2_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE));
  if (diag) printf("timing-only variant %d (results are wrong on purpose)\n", diag);
  for (const Shape& sh : shapes) {
    const int M = sh.M, N = sh.N, K = sh.K;
    if (M % BM || N % BN || K % 16) { printf("%s: shape not tileable\n", sh.what); continue; }
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hB) v = rnd();
    float *dA, *dB, *dC; uint16_t *pA, *pB;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMalloc(&pA, hA.size() * 6)); CK(hipMalloc(&pB, hB.size() * 6));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(split_blocked_kernel, dim3((hA.size() + 255) / 256), dim3(256), 0, 0, dA, pA, M, K);
    hipLaunchKernelGGL(split_blocked_kernel, dim3((hB.size() + 255) / 256), dim3(256), 0, 0, dB, pB, N, K);
    const int ntm = M / BM, ntn = N / BN;
    auto launch = [&]() {
      if (diag == 1) hipLaunchKernelGGL(gemm_planes2_kernel<1>, dim3(ntm * ntn), dim3(NTHR), NST * STAGE, 0, pA, pB, dC, M, N, K, ntn);
      else if (diag == 2) hipLaunchKernelGGL(gemm_planes2_kernel<2>, dim3(ntm * ntn), dim3(NTHR), NST * STAGE, 0, pA, pB, dC, M, N, K, ntn);
      else hipLaunchKernelGGL(gemm_planes2_kernel<0>, dim3(ntm * ntn), dim3(NTHR), NST * STAGE, 0, pA, pB, dC, M, N, K, ntn);
    };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters;
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 256; ++t) {
      st = st * 1664525u + 1013904223u; const int m = (st >> 4) % M;
      st = st * 1664525u + 1013904223u; const int n = (st >> 4) % N;
      double ref = 0, mag = 0;
      for (int k = 0; k < K; ++k) { const double p = (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k]; ref += p; mag += fabs(p); }
      const double err = fabs(ref - hC[(size_t)m * N + n]) / mag;
      if (err > worst) worst = err;
    }
    const double flop = 2.0 * M * N * K;
    printf("%-24s M=%5d N=%5d K=%5d  %8.1f us  %6.1f TFLOP/s fp32-equivalent (%.2f of the 417 bf16x3 peak)  max err / sum|ab| %.2e  grid %d\n",
           sh.what, M, N, K, us, flop / us * 1e-6, flop / us * 1e-6 / 417.0, worst, ntm * ntn);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(pA)); CK(hipFree(pB));
  }
  return 0;
}
