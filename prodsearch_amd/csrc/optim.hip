// optim.hip — fused clip-by-global-norm + dense Adam over a table of tensors (gfx950).
//
// Reference: Optimizer.step (models/optimizers.py:205-243) = clip_grad_norm_(params,
// max_grad_norm) then torch.optim.Adam(lr, betas, eps=1e-9, weight_decay) (:186-187).
// Two launches regardless of the number of tensors:
//   1. sumsq : per-chunk sum of (g*grad_scale)^2 -> partial[chunk]; block 0 bumps the step.
//   2. update: every block re-reduces partial[] in a fixed order (bitwise reproducible
//      norm), derives clip coefficient + bias corrections, then streams p,g,m,v once
//      (16-byte lanes when the four pointers are aligned).  Pure HBM streaming: 7 dwords
//      of traffic per parameter (read p,g,m,v; write p,m,v) — 6 where the sum-of-squares pass saw
//      only zeros in a lane's 16 bytes of gradient (AdamPlanHeader::off_mask): untouched table rows.
#include "optim_core.h"
#include <string.h>

static inline int64_t align16(int64_t x) { return (x + 15) & ~(int64_t)15; }

extern "C" int64_t ps_adam_plan_bytes(int32_t n, const int64_t* numel) {
  int64_t chunks = 0;
  for (int i = 0; i < n; ++i) chunks += (numel[i] + ADAM_CHUNK - 1) / ADAM_CHUNK;
  return align16(sizeof(AdamPlanHeader)) + align16((int64_t)sizeof(AdamChunkRec) * chunks) + 4 * align16(8 * (int64_t)n) +
         align16(8 * (int64_t)n) + align16(4 * (int64_t)(n + 1)) + align16(4 * chunks) + 8 * (int64_t)ADAM_MASK_WORDS * chunks;
}

extern "C" int ps_adam_plan_write_host(int32_t n, float* const* p, float* const* g, float* const* m,
                                       float* const* v, const int64_t* numel, void* plan_host) {
  PS_REQUIRE(n > 0, "adam plan: no tensors");
  char* base = (char*)plan_host;
  AdamPlanHeader h;
  int64_t total_chunks = 0;
  for (int i = 0; i < n; ++i) total_chunks += (numel[i] + ADAM_CHUNK - 1) / ADAM_CHUNK;
  int64_t off = ADAM_REC_OFF + align16((int64_t)sizeof(AdamChunkRec) * total_chunks);
  AdamChunkRec* recs = (AdamChunkRec*)(base + ADAM_REC_OFF);
  h.off_p = off; off += align16(8 * (int64_t)n);
  h.off_g = off; off += align16(8 * (int64_t)n);
  h.off_m = off; off += align16(8 * (int64_t)n);
  h.off_v = off; off += align16(8 * (int64_t)n);
  h.off_numel = off; off += align16(8 * (int64_t)n);
  h.off_chunk0 = off; off += align16(4 * (int64_t)(n + 1));
  h.off_chunk_tensor = off; off += align16(4 * total_chunks);
  h.off_mask = off;                   // device scratch: never read before the sum-of-squares pass of a step has written it
  int32_t* chunk0 = (int32_t*)(base + h.off_chunk0);
  int32_t* chunk_tensor = (int32_t*)(base + h.off_chunk_tensor);
  int64_t chunks = 0;
  for (int i = 0; i < n; ++i) {
    PS_REQUIRE(p[i] && g[i] && m[i] && v[i] && numel[i] > 0, "adam plan: tensor %d null/empty", i);
    ((float**)(base + h.off_p))[i] = p[i];
    ((float**)(base + h.off_g))[i] = g[i];
    ((float**)(base + h.off_m))[i] = m[i];
    ((float**)(base + h.off_v))[i] = v[i];
    ((int64_t*)(base + h.off_numel))[i] = numel[i];
    chunk0[i] = (int32_t)chunks;
    const int64_t nc = (numel[i] + ADAM_CHUNK - 1) / ADAM_CHUNK;
    PS_REQUIRE(chunks + nc < (1 << 30), "adam plan: too many chunks");
    for (int64_t c = 0; c < nc; ++c) {
      chunk_tensor[chunks + c] = i;
      const int64_t beg = c * ADAM_CHUNK, left = numel[i] - beg;
      AdamChunkRec& r = recs[chunks + c];
      r.p = p[i] + beg; r.g = g[i] + beg; r.m = m[i] + beg; r.v = v[i] + beg;
      r.n = (int32_t)(left < ADAM_CHUNK ? left : ADAM_CHUNK); r.pad_ = 0;
    }
    chunks += nc;
  }
  chunk0[n] = (int32_t)chunks;
  PS_REQUIRE(chunks < (1 << 30), "adam plan: too many chunks");
  h.n_tensors = n; h.n_chunks = (int32_t)chunks;
  memcpy(base, &h, sizeof(h));
  return PS_OK;
}

__global__ __launch_bounds__(256) void adam_sumsq_kernel(const char* plan, float grad_scale, int64_t* state,
                                                         float* partial) {
  __shared__ float sh[4];
  const int chunk = blockIdx.x;
  const float s = adam_sumsq_chunk(plan, chunk, grad_scale, sh);
  if (threadIdx.x == 0) {
    partial[chunk] = s;
    if (chunk == 0) state[0] += 1;
  }
}

__global__ __launch_bounds__(256) void adam_update_kernel(const char* plan, const PsAdamHyper hp, const int64_t* state,
                                                          const float* partial, float* gnorm_out) {
  __shared__ float sh[4];
  __shared__ float scal[4];   // coef*grad_scale, step_size, inv_sqrt_bc2, lr
  const AdamPlanHeader* h = (const AdamPlanHeader*)plan;
  const int chunk = blockIdx.x;
  // fixed-order reduction of the partial sums: identical in every block
  const float s = strided_sum_f32<8>(partial, h->n_chunks, threadIdx.x, 256);    // (in front of every block's stream: eight loads in flight)
  const float total = block_sum_256(s, sh);
  if (threadIdx.x == 0) {
    float norm;
    adam_scalars(hp, total, state[0], scal, &norm);
    if (chunk == 0 && gnorm_out) { gnorm_out[0] = norm; gnorm_out[1] = scal[3]; }
  }
  __syncthreads();
  const AdamScal a = {scal[0], scal[1], scal[2], hp.beta1, hp.beta2, hp.eps, hp.weight_decay, hp.zero_grads, scal[3], hp.method};
  adam_update_chunk(plan, chunk, a);
}

// state_dev: int64[2] {step, unused} followed by float scratch partial[n_chunks] at state_dev+2.
extern "C" int ps_clip_adam_dense(const void* plan_dev, int32_t n_chunks, const PsAdamHyper* hyper,
                                    int64_t* state_dev, float* gnorm_out_dev, ps_stream_t stream) {
  PS_REQUIRE(plan_dev && hyper && state_dev && n_chunks > 0, "clip_adam: bad argument");
  PS_REQUIRE(hyper->method >= 0 && hyper->method <= 3, "clip_adam: unknown method %d", hyper->method);
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

// ---- sharded form (data parallel, prodsearch_amd/dist.py ShardedAdamExchange): every rank owns 1/world of the flat
// gradient after a reduce-scatter, so the global clip norm (optimizers.py:241-242 is a norm over ALL gradients) needs the
// ranks' partial sums added by one scalar all-reduce BETWEEN the two launches.  ps_adam_sumsq leaves this rank's sum of
// squares (fixed-order reduction of the per-chunk partials) in out_sumsq_dev and bumps the step; after the all-reduce
// ps_adam_update_ext applies clip + Adam with that external total.
__global__ __launch_bounds__(256) void adam_total_kernel(const float* partial, int n_chunks, float* out) {
  __shared__ float sh[4];
  const float s = strided_sum_f32<8>(partial, n_chunks, threadIdx.x, 256);
  const float total = block_sum_256(s, sh);
  if (threadIdx.x == 0) out[0] = total;
}
__global__ __launch_bounds__(256) void adam_update_ext_kernel(const char* plan, const PsAdamHyper hp, const int64_t* state,
                                                              const float* total_sumsq, float* gnorm_out) {
  __shared__ float scal[4];
  const int chunk = blockIdx.x;
  if (threadIdx.x == 0) {
    float norm;
    adam_scalars(hp, *total_sumsq, state[0], scal, &norm);
    if (chunk == 0 && gnorm_out) { gnorm_out[0] = norm; gnorm_out[1] = scal[3]; }
  }
  __syncthreads();
  const AdamScal a = {scal[0], scal[1], scal[2], hp.beta1, hp.beta2, hp.eps, hp.weight_decay, hp.zero_grads, scal[3], hp.method};
  adam_update_chunk(plan, chunk, a);
}
extern "C" int ps_adam_sumsq(const void* plan_dev, int32_t n_chunks, const PsAdamHyper* hyper, int64_t* state_dev,
                             float* out_sumsq_dev, ps_stream_t stream) {
  PS_REQUIRE(plan_dev && hyper && state_dev && out_sumsq_dev && n_chunks > 0, "adam_sumsq: bad argument");
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)(state_dev + 2);
  const float gs = hyper->grad_scale == 0.f ? 1.f : hyper->grad_scale;
  hipLaunchKernelGGL(adam_sumsq_kernel, dim3(n_chunks), dim3(256), 0, st, (const char*)plan_dev, gs, state_dev, partial);
  PS_LAUNCH_CHECK();
  hipLaunchKernelGGL(adam_total_kernel, dim3(1), dim3(256), 0, st, partial, n_chunks, out_sumsq_dev);
  PS_LAUNCH_CHECK();
  return PS_OK;
}
extern "C" int ps_adam_update_ext(const void* plan_dev, int32_t n_chunks, const PsAdamHyper* hyper, int64_t* state_dev,
                                  const float* total_sumsq_dev, float* gnorm_out_dev, ps_stream_t stream) {
  PS_REQUIRE(plan_dev && hyper && state_dev && total_sumsq_dev && n_chunks > 0, "adam_update_ext: bad argument");
  PS_REQUIRE(hyper->method >= 0 && hyper->method <= 3, "adam_update_ext: unknown method %d", hyper->method);
  PsAdamHyper hp = *hyper;
  if (hp.grad_scale == 0.f) hp.grad_scale = 1.f;
  hipLaunchKernelGGL(adam_update_ext_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, (const char*)plan_dev, hp,
                     state_dev, total_sumsq_dev, gnorm_out_dev);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

extern "C" int32_t ps_adam_plan_chunks_host(const void* plan_host) {
  return ((const AdamPlanHeader*)plan_host)->n_chunks;
}

// Data-parallel reduce-scatter in its peer-to-peer form (dist.ShardedAdamExchange, rs 'a2a'): after the equal-split all-to-all
// recv[r][0..n) is rank r's copy of THIS rank's slice of the flat gradient; out[i] = sum_r recv[r][i] added in rank order (the
// same bits on every run), and — the all-to-all has consumed it — the flat gradient buffer is cleared by the same launch, so
// the step path has no torch kernel and no separate memset.  float4 grid-stride; n % 4 == 0 (slices are cut that way).
__global__ __launch_bounds__(256) void sum_slices_kernel(const float4* __restrict__ recv, int W, int64_t n4, float4* __restrict__ out,
                                                         float4* __restrict__ zero, int64_t zero4) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 s = recv[i];
    for (int r = 1; r < W; ++r) {
      const float4 v = recv[(int64_t)r * n4 + i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    out[i] = s;
  }
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < zero4; i += stride) zero[i] = z;
}
extern "C" int ps_sum_slices(const float* recv, int32_t world, int64_t n, float* out, float* zero, int64_t zero_n,
                             ps_stream_t stream) {
  PS_REQUIRE(recv && out && world >= 1 && n >= 0 && n % 4 == 0 && zero_n >= 0 && zero_n % 4 == 0 && (zero || zero_n == 0),
             "sum_slices: bad argument (n %lld, zero_n %lld)", (long long)n, (long long)zero_n);
  PS_REQUIRE((((uintptr_t)recv | (uintptr_t)out | (uintptr_t)zero) & 15) == 0, "sum_slices: buffers must be 16-byte aligned");
  const int64_t work = (n > zero_n ? n : zero_n) / 4;
  if (work == 0) return PS_OK;
  const int grid = (int)(work / 256 + 1 < 2048 ? work / 256 + 1 : 2048);
  hipLaunchKernelGGL(sum_slices_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float4*)recv, world, n / 4, (float4*)out,
                     (float4*)zero, zero_n / 4);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

extern "C" int ps_zero_floats(float* p, int64_t n, ps_stream_t stream) {
  PS_REQUIRE(p && n >= 0, "zero: bad argument");
  PS_CHECK_HIP(hipMemsetAsync(p, 0, (size_t)n * sizeof(float), (hipStream_t)stream));
  return PS_OK;
}
