// fp32 GEMM tile through bf16 MFMAs: x = hi + mid + lo (three bf16, 24 mantissa bits), products hh + hm + mh + mm + hl + lh.
// Checks accuracy against fp64 and the fp32 MFMA, times both forms, and whether bf16 MFMAs overlap the other wave's VALU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define KD 128
#define AST (KD + 8)     // padded row stride in bf16 elements (272 B)

__device__ inline void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x; float r = x - (float)h; m = (__bf16)r; r -= (float)m; l = (__bf16)r;
}
// one wave: C[32x32] = A[32xKD] . B[32xKD]^T ; A, B given as fp32 in global
__global__ __launch_bounds__(64) void tile_bf16x3(const float* A, const float* B, float* C, int reps) {
  __shared__ __bf16 As[3][32 * AST], Bs[3][32 * AST];
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  for (int i = lane; i < 32 * KD; i += 64) {
    const int row = i / KD, k = i % KD;
    split3(A[i], As[0][row * AST + k], As[1][row * AST + k], As[2][row * AST + k]);
    split3(B[i], Bs[0][row * AST + k], Bs[1][row * AST + k], Bs[2][row * AST + k]);
  }
  __syncthreads();
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int rep = 0; rep < reps; ++rep) {
    if (rep) for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int k0 = 0; k0 < KD; k0 += 16) {
      bf16x8 a[3], b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        a[p] = *reinterpret_cast<const bf16x8*>(&As[p][r * AST + k0 + 8 * h]);
        b[p] = *reinterpret_cast<const bf16x8*>(&Bs[p][r * AST + k0 + 8 * h]);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);   // small terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
  }
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}
__global__ __launch_bounds__(64) void tile_f32(const float* A, const float* B, float* C, int reps) {
  __shared__ float As[KD][33], Bs[KD][33];
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  for (int i = lane; i < 32 * KD; i += 64) { As[i % KD][i / KD] = A[i]; Bs[i % KD][i / KD] = B[i]; }
  __syncthreads();
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int rep = 0; rep < reps; ++rep) {
    if (rep) for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 16
    for (int k = 0; k < KD; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[k + h][r], Bs[k + h][r], acc, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}
// co-execution: waves 0-3 bf16 MFMA chain, waves 4-7 v_fma chain
__global__ __launch_bounds__(512, 2) void coexec(float* out, int nm, int nv) {
  const int wave = threadIdx.x >> 6;
  float rr = 0.f;
  if (wave < 4) {
    f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f + i); b[i] = (__bf16)1.0f; }
    for (int i = 0; i < nm; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) rr += acc[i];
  } else {
    float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f;
    for (int i = 0; i < nv; i += 4) {
      x0 = __builtin_fmaf(x0, 1.0001f, 0.5f); x1 = __builtin_fmaf(x1, 1.0001f, 0.5f);
      x2 = __builtin_fmaf(x2, 1.0001f, 0.5f); x3 = __builtin_fmaf(x3, 1.0001f, 0.5f);
    }
    rr = x0 + x1 + x2 + x3;
  }
  if (rr == 12345.678f) out[threadIdx.x] = rr;
}
template <class F> static float timeit(F f, int n) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipEventRecord(e0, 0);
  for (int i = 0; i < n; ++i) f();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1000.f / n;
}
int main() {
  const int n = 32 * KD;
  float *hA = (float*)malloc(n * 4), *hB = (float*)malloc(n * 4), hC[1024], hD[1024];
  srand(1);
  for (int i = 0; i < n; ++i) { hA[i] = (rand() / (float)RAND_MAX - 0.5f) * 4.f; hB[i] = (rand() / (float)RAND_MAX - 0.5f) * 0.3f; }
  for (int i = 0; i < 64; ++i) hA[i] *= 1e-6f;            // a row of tiny values (gradient-like magnitudes)
  float *dA, *dB, *dC, *dD;
  hipMalloc(&dA, n * 4); hipMalloc(&dB, n * 4); hipMalloc(&dC, 4096); hipMalloc(&dD, 4096);
  hipMemcpy(dA, hA, n * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(tile_bf16x3, dim3(1), dim3(64), 0, 0, dA, dB, dC, 1);
  hipLaunchKernelGGL(tile_f32, dim3(1), dim3(64), 0, 0, dA, dB, dD, 1);
  hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost); hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
  double e3 = 0, e1 = 0, mx = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double ref = 0, mag = 0;
    for (int k = 0; k < KD; ++k) { ref += (double)hA[i * KD + k] * hB[j * KD + k]; mag += fabs((double)hA[i * KD + k] * hB[j * KD + k]); }
    e3 = fmax(e3, fabs(hC[i * 32 + j] - ref) / mag); e1 = fmax(e1, fabs(hD[i * 32 + j] - ref) / mag); mx = fmax(mx, fabs(ref));
  }
  printf("max |err| / sum|a b|:  bf16x3 (6 MFMA) %.3e   fp32 MFMA %.3e\n", e3, e1);
  const int reps = 2000;
  float t3 = timeit([&] { hipLaunchKernelGGL(tile_bf16x3, dim3(256), dim3(64), 0, 0, dA, dB, dC, reps); }, 5);
  float t1 = timeit([&] { hipLaunchKernelGGL(tile_f32, dim3(256), dim3(64), 0, 0, dA, dB, dD, reps); }, 5);
  printf("one wave per CU, %d x (32x32x%d): bf16x3 %.1f us (%.0f cycles/product @2.1GHz)   fp32 %.1f us (%.0f cycles/product)\n",
         reps, KD, t3, t3 * 2100.0 / reps, t1, t1 * 2100.0 / reps);
  float tm = timeit([&] { hipLaunchKernelGGL(coexec, dim3(256), dim3(512), 0, 0, dC, 8000, 0); }, 5);
  float tv = timeit([&] { hipLaunchKernelGGL(coexec, dim3(256), dim3(512), 0, 0, dC, 0, 64000); }, 5);
  float tb = timeit([&] { hipLaunchKernelGGL(coexec, dim3(256), dim3(512), 0, 0, dC, 8000, 64000); }, 5);
  printf("bf16 mfma only %.1f us, valu only %.1f us, both (other wave) %.1f us\n", tm, tv, tb);
  return 0;
}
