// gemm_planes.hip — prototype (round 5): a bf16x3 product  C[M][N] = A[M][K] . B[N][K]^T  whose operands are ALREADY three bf16
// planes in memory (hi / mid / lo of an exact 3-way split of the fp32 values, [3][rows][K] bf16), so that the loop has no split
// VALU at all: operands go global -> LDS by LDS-DMA, fragments LDS -> registers by ds_read_b128, six bf16 MFMAs per 32x32x16
// product step (fp32-grade, as gemm_x3_kernel), one raw barrier per 32-deep K-step, the next stage's DMA in flight under the
// MFMAs ("minimum 2-phase" loop of cdna_hip_programming.md 5).  Tile 256 x 128, 8 waves (4 x 2, 64 x 64 each), two 72 KB stages.
// Standalone: measures what the plane format would buy the d >= 256 chain before any producer is changed to write planes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm_planes gemm_planes.hip && ./gemm_planes
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

constexpr int BM = 256, BN = 128, BK = 32, NTHR = 512;
constexpr int A_PLANE = BM * BK * 2, B_PLANE = BN * BK * 2;        // bytes per plane image: 16 KB, 8 KB
constexpr int STAGE = 3 * A_PLANE + 3 * B_PLANE;                    // 72 KB
constexpr int ROUNDS = STAGE / 16 / NTHR;                           // 9 DMA instructions per thread and stage

// ---- split fp32 -> three bf16 planes (round-to-nearest pieces, exact sum)
__global__ void split_kernel(const float* x, uint16_t* planes, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  const __bf16 h = (__bf16)v;
  float r = v - (float)h;
  const __bf16 m = (__bf16)r;
  r -= (float)m;
  const __bf16 l = (__bf16)r;
  planes[i] = __builtin_bit_cast(uint16_t, h);
  planes[n + i] = __builtin_bit_cast(uint16_t, m);
  planes[2 * n + i] = __builtin_bit_cast(uint16_t, l);
}

#define LDS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))

// one stage's DMA: chunk q of the stage image (16 B each, lane-linear) <- the source chunk its swizzled position holds
struct Src { const uint16_t* a; const uint16_t* b; size_t a_plane, b_plane; int K; };
__device__ __forceinline__ void stage_issue(const Src& s, int m0, int n0, int k0, unsigned char* stage, int tid, int wave) {
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int q = r * NTHR + tid;                                   // chunk index inside the stage
    const bool isA = r < 6;                                          // rounds 0-5: A planes (1024 chunks each), 6-8: B planes (512 each)
    const int plane = isA ? r >> 1 : r - 6;
    const int within = isA ? q - plane * 1024 : q - 3072 - plane * 512;
    const int row = within >> 2, pos = within & 3;
    const int c = pos ^ ((row >> 2) & 3);                            // source chunk held at this position
    const uint16_t* src = isA ? s.a + plane * s.a_plane + (size_t)(m0 + row) * s.K + k0 + 8 * c
                              : s.b + plane * s.b_plane + (size_t)(n0 + row) * s.K + k0 + 8 * c;
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(stage + (r * NTHR + wave * 64) * 16), 16, 0, 0);
  }
}

#define BF(x) __builtin_bit_cast(bf16x8, x)
__device__ __forceinline__ void x3(f32x16& acc, const i32x4 (&a)[3], const i32x4 (&b)[3]) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[0]), BF(b[2]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[2]), BF(b[0]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[1]), BF(b[1]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[0]), BF(b[1]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[1]), BF(b[0]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[0]), BF(b[0]), acc, 0, 0, 0);
}

// fragments of one K-step out of the stage at LDS byte address `sb`: A rows 64 wm + 32 i + l31, B rows 64 wn + 32 j + l31
template <int T>                                                     // k16 sub-step
__device__ __forceinline__ void read_frags(i32x4 (&fa)[2][3], i32x4 (&fb)[2][3], uint32_t aA, uint32_t aB) {
  // immediate offsets: plane p * plane bytes + tile * 2048
  LDS_READ(fa[0][0], aA, 0);     LDS_READ(fa[0][1], aA, 16384); LDS_READ(fa[0][2], aA, 32768);
  LDS_READ(fa[1][0], aA, 2048);  LDS_READ(fa[1][1], aA, 18432); LDS_READ(fa[1][2], aA, 34816);
  LDS_READ(fb[0][0], aB, 0);     LDS_READ(fb[0][1], aB, 8192);  LDS_READ(fb[0][2], aB, 16384);
  LDS_READ(fb[1][0], aB, 2048);  LDS_READ(fb[1][1], aB, 10240); LDS_READ(fb[1][2], aB, 18432);
  (void)T;
}

template <int DIAG>   // timing-only variants (WRONG results): 1 no DMA inside the loop, 2 no MFMAs, 3 no fragment reads
__global__ __launch_bounds__(NTHR) void gemm_planes_kernel(const uint16_t* Ap, const uint16_t* Bp, float* C, int M, int N, int K, int ntn) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile walk: workgroup L runs on XCD L % 8; every XCD gets a contiguous run of tiles in row-block-major order, so the
  // column tiles that share an A row block share an L2
  const int L = blockIdx.x, nt = gridDim.x;
  int tile = L;
  if (nt % 8 == 0) tile = (L & 7) * (nt >> 3) + (L >> 3);
  const int tm = tile / ntn, tn = tile - tm * ntn;
  const int m0 = tm * BM, n0 = tn * BN;
  const Src s = {Ap, Bp, (size_t)M * K, (size_t)N * K, K};
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
  // per-lane fragment addresses (stage 0); chunk (2t + h) of a row sits at position (2t + h) ^ ((row >> 2) & 3)
  const int swz = (l31 >> 2) & 3;
  uint32_t aA[2], aB[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const uint32_t off = (uint32_t)(l31 * 64 + (((2 * t + h) ^ swz) << 4));
    aA[t] = lds0 + off + wm * 4096;
    aB[t] = lds0 + off + 3 * A_PLANE + wn * 4096;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int KT = K / BK;
  stage_issue(s, m0, n0, 0, smem, tid, wave);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    const uint32_t so = cur ? STAGE : 0;
    if (kt + 1 < KT && DIAG != 1) stage_issue(s, m0, n0, (kt + 1) * BK, smem + (cur ? 0 : STAGE), tid, wave);   // flies under this step's products
    i32x4 fa0[2][3], fb0[2][3], fa1[2][3], fb1[2][3];
    if (DIAG != 3) {
      read_frags<0>(fa0, fb0, aA[0] + so, aB[0] + so);
      read_frags<1>(fa1, fb1, aA[1] + so, aB[1] + so);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) { asm volatile("" : "=v"(fa0[i][p])); asm volatile("" : "=v"(fb0[i][p])); asm volatile("" : "=v"(fa1[i][p])); asm volatile("" : "=v"(fb1[i][p])); }
    }
    asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if (DIAG != 2) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) x3(acc[i][j], fa0[i], fb0[j]);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) { asm volatile("" :: "v"(fa0[i][p])); asm volatile("" :: "v"(fb0[i][p])); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (DIAG != 2) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) x3(acc[i][j], fa1[i], fb1[j]);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) { asm volatile("" :: "v"(fa1[i][p])); asm volatile("" :: "v"(fb1[i][p])); }
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the next stage has landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();                                    // ... everybody's, and everybody is done reading this one
  }
  // epilogue: lane (column l31, half h) of tile (i, j): register r = row (r & 3) + 8 (r >> 2) + 4 h
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

static double now_us(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms * 1e3; }

int main(int argc, char** argv) {
  struct Shape { int M, N, K; const char* what; };
  const Shape shapes[] = {{21504, 1024, 256, "FF1 of the C5 shard"}, {21504, 256, 1024, "FF2"}, {4096, 4096, 4096, "4096^3"}, {8192, 8192, 1024, "8192 x 8192 x 1024"}};
  const int diag = argc > 1 ? atoi(argv[1]) : 0;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_planes_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_planes_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_planes_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_planes_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
  if (diag) printf("timing-only variant %d (results are wrong on purpose)\n", diag);
  for (const Shape& sh : shapes) {
    const int M = sh.M, N = sh.N, K = sh.K;
    if (M % BM || N % BN || K % BK) { printf("%s: shape not tileable\n", sh.what); continue; }
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hB) v = rnd();
    float *dA, *dB, *dC; uint16_t *pA, *pB;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMalloc(&pA, hA.size() * 6)); CK(hipMalloc(&pB, hB.size() * 6));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(split_kernel, dim3((hA.size() + 255) / 256), dim3(256), 0, 0, dA, pA, hA.size());
    hipLaunchKernelGGL(split_kernel, dim3((hB.size() + 255) / 256), dim3(256), 0, 0, dB, pB, hB.size());
    const int ntm = M / BM, ntn = N / BN;
    auto launch = [&]() {
      if (diag == 1) hipLaunchKernelGGL(gemm_planes_kernel<1>, dim3(ntm * ntn), dim3(NTHR), 2 * STAGE, 0, pA, pB, dC, M, N, K, ntn);
      else if (diag == 2) hipLaunchKernelGGL(gemm_planes_kernel<2>, dim3(ntm * ntn), dim3(NTHR), 2 * STAGE, 0, pA, pB, dC, M, N, K, ntn);
      else if (diag == 3) hipLaunchKernelGGL(gemm_planes_kernel<3>, dim3(ntm * ntn), dim3(NTHR), 2 * STAGE, 0, pA, pB, dC, M, N, K, ntn);
      else hipLaunchKernelGGL(gemm_planes_kernel<0>, dim3(ntm * ntn), dim3(NTHR), 2 * STAGE, 0, pA, pB, dC, M, N, K, ntn);
    };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    const double us = now_us(e0, e1) / iters;
    // check 64 random entries against fp64
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 64; ++t) {
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
