// Does a HIP event pair BOUND to a kernel launch (hipExtLaunchKernelGGL start / stop events) read the kernel's own dispatch
// duration — what rocprofv3's kernel trace reports — where a hipEventRecord pair around the launch also reads the pair's packets?
//   hipcc --offload-arch=gfx950 -O2 -o extlaunch extlaunch.hip && ./extlaunch
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void spin_kernel(float* p, int iters, long long* stamps) {
  const long long t0 = __builtin_readcyclecounter();
  float v = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[blockIdx.x * blockDim.x + threadIdx.x] = v;
  (void)t0; (void)stamps;
}

int main() {
  float* p;
  CK(hipMalloc(&p, 2048 * 256 * 4));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const int N = 50;
  hipEvent_t a[N], b[N], c[N], d[N];
  for (int i = 0; i < N; ++i) { CK(hipEventCreate(&a[i])); CK(hipEventCreate(&b[i])); CK(hipEventCreate(&c[i])); CK(hipEventCreate(&d[i])); }
  for (int iters : {100, 2000, 20000}) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(spin_kernel, dim3(2048), dim3(256), 0, st, p, iters, nullptr);
    CK(hipStreamSynchronize(st));
    // (1) event pair bound to the launch
    for (int i = 0; i < N; ++i) hipExtLaunchKernelGGL(spin_kernel, dim3(2048), dim3(256), 0, st, a[i], b[i], 0, p, iters, nullptr);
    CK(hipStreamSynchronize(st));
    double s1 = 0;
    for (int i = 0; i < N; ++i) { float ms; CK(hipEventElapsedTime(&ms, a[i], b[i])); s1 += ms * 1e3; }
    // (2) hipEventRecord pair around a plain launch
    for (int i = 0; i < N; ++i) {
      CK(hipEventRecord(c[i], st));
      hipLaunchKernelGGL(spin_kernel, dim3(2048), dim3(256), 0, st, p, iters, nullptr);
      CK(hipEventRecord(d[i], st));
    }
    CK(hipStreamSynchronize(st));
    double s2 = 0;
    for (int i = 0; i < N; ++i) { float ms; CK(hipEventElapsedTime(&ms, c[i], d[i])); s2 += ms * 1e3; }
    // (3) back-to-back loop
    CK(hipEventRecord(c[0], st));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin_kernel, dim3(2048), dim3(256), 0, st, p, iters, nullptr);
    CK(hipEventRecord(d[0], st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, c[0], d[0]));
    printf("iters %6d: bound pair %.2f us   record pair %.2f us   back-to-back %.2f us per launch\n", iters, s1 / N, s2 / N, ms * 1e3 / N);
  }
  return 0;
}
