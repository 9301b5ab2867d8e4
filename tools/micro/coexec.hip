// Do f32 MFMAs of one wave overlap with VALU / LDS work of the OTHER wave on the same SIMD?  (gfx950)
// 512-thread workgroups, one per CU: waves 0-3 run `nm` dependent v_mfma_f32_32x32x2_f32, waves 4-7 run `nv` v_fma_f32
// (mode 1), ds_write_b32 (mode 2) or MFMAs too (mode 3); either side can be switched off.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512, 2) void k(float* out, int nm, int nv, int mode, int swap) {
  __shared__ float lds[8192];
  const int wave = threadIdx.x >> 6;
  const bool first = swap ? wave >= 4 : wave < 4;
  float r = 0.f;
  if (first) {
    f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < nm; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) r += acc[i];
  } else if (mode == 1) {
    float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f;
    for (int i = 0; i < nv; i += 4) {
      x0 = __builtin_fmaf(x0, 1.0001f, 0.5f); x1 = __builtin_fmaf(x1, 1.0001f, 0.5f);
      x2 = __builtin_fmaf(x2, 1.0001f, 0.5f); x3 = __builtin_fmaf(x3, 1.0001f, 0.5f);
    }
    r = x0 + x1 + x2 + x3;
  } else if (mode == 2) {
    for (int i = 0; i < nv; ++i) lds[(threadIdx.x + 64 * (i & 15)) & 8191] = (float)i;
    __syncthreads_count(0);
    r = lds[threadIdx.x];
  } else if (mode == 3) {
    f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < nv; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) r += acc[i];
  }
  if (r == 12345.678f) out[threadIdx.x] = r;
}
static float run(float* d, int nm, int nv, int mode, int swap) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, nm, nv, mode, swap);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, nm, nv, mode, swap);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 100.f;   // us per launch
}
int main() {
  float* d; hipMalloc(&d, 4096);
  const int NM = 4000;                       // 4000 MFMAs = 256k cycles
  printf("mfma only (%d):            %.1f us\n", NM, run(d, NM, 0, 1, 0));
  const int NV = 64000;                      // 64000 fma = 256k cycles at 4 cycles each
  printf("valu only (%d fma):       %.1f us\n", NV, run(d, 0, NV, 1, 0));
  printf("mfma + valu (other wave):    %.1f us\n", run(d, NM, NV, 1, 0));
  printf("mfma + valu (roles swapped): %.1f us\n", run(d, NM, NV, 1, 1));
  const int NL = 32000;
  printf("lds writes only (%d):     %.1f us\n", NL, run(d, 0, NL, 2, 0));
  printf("mfma + lds writes:           %.1f us\n", run(d, NM, NL, 2, 0));
  printf("mfma + mfma (both waves):    %.1f us\n", run(d, NM, NM, 3, 0));
  return 0;
}
