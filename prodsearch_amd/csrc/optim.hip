// optim.hip — fused clip-by-global-norm + dense Adam over a table of tensors (gfx950).
//
// Reference: Optimizer.step (models/optimizers.py:205-243) = clip_grad_norm_(params,
// max_grad_norm) then torch.optim.Adam(lr, betas, eps=1e-9, weight_decay) (:186-187).
// Two launches regardless of the number of tensors:
//   1. sumsq : per-chunk sum of (g*grad_scale)^2 -> partial[chunk]; block 0 bumps the step.
//   2. update: every block re-reduces partial[] in a fixed order (bitwise reproducible
//      norm), derives clip coefficient + bias corrections, then streams p,g,m,v once
//      (16-byte lanes when the four pointers are aligned).  Pure HBM streaming: 7 dwords
//      of traffic per parameter (read p,g,m,v; write p,m,v).
#include "common.h"
#include <math.h>
#include <string.h>

#define ADAM_CHUNK 8192     // elements per block

struct AdamPlanHeader {
  int32_t n_tensors;
  int32_t n_chunks;
  int64_t off_p, off_g, off_m, off_v, off_numel, off_chunk0;   // byte offsets inside the plan
};

static inline int64_t align16(int64_t x) { return (x + 15) & ~(int64_t)15; }

extern "C" int64_t ps_adam_plan_bytes(int32_t n, const int64_t* numel) {
  (void)numel;
  return align16(sizeof(AdamPlanHeader)) + 4 * align16(8 * (int64_t)n) + align16(8 * (int64_t)n) +
         align16(4 * (int64_t)(n + 1));
}

extern "C" int ps_adam_plan_write_host(int32_t n, float* const* p, float* const* g, float* const* m,
                                       float* const* v, const int64_t* numel, void* plan_host) {
  PS_REQUIRE(n > 0, "adam plan: no tensors");
  char* base = (char*)plan_host;
  AdamPlanHeader h;
  int64_t off = align16(sizeof(AdamPlanHeader));
  h.off_p = off; off += align16(8 * (int64_t)n);
  h.off_g = off; off += align16(8 * (int64_t)n);
  h.off_m = off; off += align16(8 * (int64_t)n);
  h.off_v = off; off += align16(8 * (int64_t)n);
  h.off_numel = off; off += align16(8 * (int64_t)n);
  h.off_chunk0 = off;
  int32_t* chunk0 = (int32_t*)(base + h.off_chunk0);
  int64_t chunks = 0;
  for (int i = 0; i < n; ++i) {
    PS_REQUIRE(p[i] && g[i] && m[i] && v[i] && numel[i] > 0, "adam plan: tensor %d null/empty", i);
    ((float**)(base + h.off_p))[i] = p[i];
    ((float**)(base + h.off_g))[i] = g[i];
    ((float**)(base + h.off_m))[i] = m[i];
    ((float**)(base + h.off_v))[i] = v[i];
    ((int64_t*)(base + h.off_numel))[i] = numel[i];
    chunk0[i] = (int32_t)chunks;
    chunks += (numel[i] + ADAM_CHUNK - 1) / ADAM_CHUNK;
  }
  chunk0[n] = (int32_t)chunks;
  PS_REQUIRE(chunks < (1 << 30), "adam plan: too many chunks");
  h.n_tensors = n; h.n_chunks = (int32_t)chunks;
  memcpy(base, &h, sizeof(h));
  return PS_OK;
}

__device__ inline int find_tensor(const int32_t* chunk0, int n, int chunk) {
  int lo = 0, hi = n;                 // chunk0[lo] <= chunk < chunk0[hi]
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (chunk0[mid] <= chunk) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ inline float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void adam_sumsq_kernel(const char* plan, float grad_scale, int64_t* state,
                                                         float* partial) {
  __shared__ float sh[4];
  const AdamPlanHeader* h = (const AdamPlanHeader*)plan;
  const int chunk = blockIdx.x;
  const int32_t* chunk0 = (const int32_t*)(plan + h->off_chunk0);
  const int t = find_tensor(chunk0, h->n_tensors, chunk);
  const float* g = ((float* const*)(plan + h->off_g))[t];
  const int64_t n = ((const int64_t*)(plan + h->off_numel))[t];
  const int64_t beg = (int64_t)(chunk - chunk0[t]) * ADAM_CHUNK;
  const int64_t end = beg + ADAM_CHUNK < n ? beg + ADAM_CHUNK : n;
  float s = 0.f;
  if ((((uintptr_t)g) & 15) == 0 && end - beg == ADAM_CHUNK) {
    const float4* g4 = (const float4*)(g + beg);
#pragma unroll
    for (int i = 0; i < ADAM_CHUNK / 4 / 256; ++i) {
      float4 x = g4[threadIdx.x + 256 * i];
      x.x *= grad_scale; x.y *= grad_scale; x.z *= grad_scale; x.w *= grad_scale;
      s += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    }
  } else {
    for (int64_t i = beg + threadIdx.x; i < end; i += 256) { float x = g[i] * grad_scale; s += x * x; }
  }
  s = block_sum_256(s, sh);
  if (threadIdx.x == 0) {
    partial[chunk] = s;
    if (chunk == 0) state[0] += 1;
  }
}

__global__ __launch_bounds__(256) void adam_update_kernel(const char* plan, const PsAdamHyper hp, const int64_t* state,
                                                          const float* partial, float* gnorm_out) {
  __shared__ float sh[4];
  __shared__ float scal[4];   // coef, step_size, inv_sqrt_bc2, lr
  const AdamPlanHeader* h = (const AdamPlanHeader*)plan;
  const int chunk = blockIdx.x;
  // fixed-order reduction of the partial sums: identical in every block
  float s = 0.f;
  for (int i = threadIdx.x; i < h->n_chunks; i += 256) s += partial[i];
  const float total = block_sum_256(s, sh);
  if (threadIdx.x == 0) {
    const float norm = sqrtf(total);
    float coef = 1.f;
    if (hp.max_grad_norm > 0.f) coef = fminf(hp.max_grad_norm / (norm + 1e-6f), 1.f);
    const double t = (double)state[0];
    double lr = (double)hp.lr;
    if (hp.noam) lr = (double)hp.lr * fmin(pow(t, -0.5), t * pow((double)hp.warmup_steps, -1.5));
    const double bc1 = 1.0 - pow((double)hp.beta1, t);
    const double bc2 = 1.0 - pow((double)hp.beta2, t);
    scal[0] = coef * hp.grad_scale;
    scal[1] = (float)(lr / bc1);
    scal[2] = (float)(1.0 / sqrt(bc2));
    if (chunk == 0 && gnorm_out) { gnorm_out[0] = norm; gnorm_out[1] = (float)lr; }
  }
  __syncthreads();
  const float gmul = scal[0], step_size = scal[1], inv_sbc2 = scal[2];
  const float b1 = hp.beta1, b2 = hp.beta2, eps = hp.eps, wd = hp.weight_decay;
  const float omb1 = 1.f - b1, omb2 = 1.f - b2;

  const int32_t* chunk0 = (const int32_t*)(plan + h->off_chunk0);
  const int t = find_tensor(chunk0, h->n_tensors, chunk);
  float* p = ((float* const*)(plan + h->off_p))[t];
  const float* g = ((float* const*)(plan + h->off_g))[t];
  float* m = ((float* const*)(plan + h->off_m))[t];
  float* v = ((float* const*)(plan + h->off_v))[t];
  const int64_t n = ((const int64_t*)(plan + h->off_numel))[t];
  const int64_t beg = (int64_t)(chunk - chunk0[t]) * ADAM_CHUNK;
  const int64_t end = beg + ADAM_CHUNK < n ? beg + ADAM_CHUNK : n;

  auto upd = [&](float& pp, float gg, float& mm, float& vv) {
    gg *= gmul;
    if (wd != 0.f) gg += wd * pp;
    mm = mm + (gg - mm) * omb1;                 // exp_avg.lerp_(grad, 1-beta1)
    vv = vv * b2 + (omb2 * gg) * gg;            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
    const float denom = sqrtf(vv) * inv_sbc2 + eps;
    pp = pp - step_size * (mm / denom);         // param.addcdiv_(exp_avg, denom, -step_size)
  };
  const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0 &&
                   end - beg == ADAM_CHUNK;
  if (vec) {
    float4* p4 = (float4*)(p + beg); const float4* g4 = (const float4*)(g + beg);
    float4* m4 = (float4*)(m + beg); float4* v4 = (float4*)(v + beg);
#pragma unroll 2
    for (int i = 0; i < ADAM_CHUNK / 4 / 256; ++i) {
      const int k = threadIdx.x + 256 * i;
      float4 pp = p4[k], gg = g4[k], mm = m4[k], vv = v4[k];
      upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y);
      upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
      p4[k] = pp; m4[k] = mm; v4[k] = vv;
    }
  } else {
    for (int64_t i = beg + threadIdx.x; i < end; i += 256) {
      float pp = p[i], mm = m[i], vv = v[i];
      upd(pp, g[i], mm, vv);
      p[i] = pp; m[i] = mm; v[i] = vv;
    }
  }
}

// state_dev: int64[2] {step, unused} followed by float scratch partial[n_chunks] at state_dev+2.
extern "C" int ps_clip_adam_dense(const void* plan_dev, int32_t n_chunks, const PsAdamHyper* hyper,
                                    int64_t* state_dev, float* gnorm_out_dev, ps_stream_t stream) {
  PS_REQUIRE(plan_dev && hyper && state_dev && n_chunks > 0, "clip_adam: bad argument");
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)(state_dev + 2);
  const float gs = hyper->grad_scale == 0.f ? 1.f : hyper->grad_scale;
  PsAdamHyper hp = *hyper;
  hp.grad_scale = gs;
  hipLaunchKernelGGL(adam_sumsq_kernel, dim3(n_chunks), dim3(256), 0, st, (const char*)plan_dev, gs, state_dev,
                     partial);
  PS_LAUNCH_CHECK();
  hipLaunchKernelGGL(adam_update_kernel, dim3(n_chunks), dim3(256), 0, st, (const char*)plan_dev, hp, state_dev,
                     partial, gnorm_out_dev);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

extern "C" int32_t ps_adam_plan_chunks_host(const void* plan_host) {
  return ((const AdamPlanHeader*)plan_host)->n_chunks;
}

extern "C" int ps_zero_floats(float* p, int64_t n, ps_stream_t stream) {
  PS_REQUIRE(p && n >= 0, "zero: bad argument");
  PS_CHECK_HIP(hipMemsetAsync(p, 0, (size_t)n * sizeof(float), (hipStream_t)stream));
  return PS_OK;
}
