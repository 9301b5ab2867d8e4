// optim_core.h — pieces shared by the dense (optim.hip) and row-sparse (optim_rows.hip) clip+Adam.
#pragma once
#include "common.h"
#include <math.h>

#define ADAM_CHUNK 4096     // elements per block (8192: 0.356 ms/step, 4096: 0.353, 2048: 0.355)

struct AdamPlanHeader {
  int32_t n_tensors;
  int32_t n_chunks;
  int64_t off_p, off_g, off_m, off_v, off_numel, off_chunk0;   // byte offsets inside the plan
  int64_t off_chunk_tensor;           // int32[n_chunks]: the tensor each chunk belongs to
  int64_t off_mask;                   // uint64[n_chunks][ADAM_MASK_WORDS]: which 16-byte groups of the chunk hold a non-zero gradient
};
// Written by the sum-of-squares pass (which reads every gradient anyway), read by the update pass of the SAME step: a lane whose
// four gradients were all zero does not load them again (70 % of the table rows of a C2 step: a quarter of the update's reads).
// Word [wave * 4 + i] = the ballot of wave `wave` in iteration i of the chunk loop.  DEVICE scratch inside the plan buffer.
#define ADAM_MASK_WORDS (ADAM_CHUNK / 4 / 64)
// One record per chunk, at a FIXED offset behind the header: a block reads its four pointers (already advanced to the chunk)
// and its element count with ONE dependent load — header -> tensor index -> pointer / numel / first chunk -> data was a
// chain of four, and with every block of the launch resident at once that chain is most of the sum-of-squares kernel.
struct AdamChunkRec { float* p; float* g; float* m; float* v; int32_t n; int32_t pad_; };
#define ADAM_REC_OFF ((int64_t)((sizeof(AdamPlanHeader) + 15) & ~(size_t)15))
__host__ __device__ inline const AdamChunkRec* adam_chunk_recs(const char* plan) {
  return reinterpret_cast<const AdamChunkRec*>(plan + ADAM_REC_OFF);
}

__device__ inline int find_tensor(const char* plan, const AdamPlanHeader* h, int chunk) {
  return ((const int32_t*)(plan + h->off_chunk_tensor))[chunk];
}

__device__ inline float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return r;
}


__device__ inline unsigned long long* adam_chunk_mask(const char* plan, int chunk) {
  return (unsigned long long*)(const_cast<char*>(plan) + ((const AdamPlanHeader*)plan)->off_mask) + (int64_t)chunk * ADAM_MASK_WORDS;
}
// Sum of (g*grad_scale)^2 over one ADAM_CHUNK of the plan (block-wide result); leaves the chunk's non-zero mask for the update.
__device__ inline float adam_sumsq_chunk(const char* plan, int chunk, float grad_scale, float* sh) {
  const AdamChunkRec rec = adam_chunk_recs(plan)[chunk];
  const float* g = rec.g;                              // (advanced to the chunk)
  const int64_t beg = 0, end = rec.n;
  float s = 0.f;
  if ((((uintptr_t)g) & 15) == 0 && end - beg == ADAM_CHUNK) {
    const float4* g4 = (const float4*)(g + beg);
    unsigned long long* mask = adam_chunk_mask(plan, chunk) + (threadIdx.x >> 6) * (ADAM_CHUNK / 4 / 256);
#pragma unroll
    for (int i = 0; i < ADAM_CHUNK / 4 / 256; ++i) {
      float4 x = g4[threadIdx.x + 256 * i];
      const unsigned long long nz = __ballot(x.x != 0.f || x.y != 0.f || x.z != 0.f || x.w != 0.f);   // NaN counts as non-zero
      if ((threadIdx.x & 63) == 0) mask[i] = nz;
      x.x *= grad_scale; x.y *= grad_scale; x.z *= grad_scale; x.w *= grad_scale;
      s += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    }
  } else {
    for (int64_t i = beg + threadIdx.x; i < end; i += 256) { float x = g[i] * grad_scale; s += x * x; }
  }
  return block_sum_256(s, sh);
}

// clip coefficient, bias-corrected step size, 1/sqrt(bias_correction2), lr -> out[0..3]; one thread.
// the scalars of step `step` that do not depend on the gradients: bias-corrected step size, 1/sqrt(bias_correction2), lr.
// ONE definition: the lazy catch-up of the row-sparse optimizer (optim_rows.hip) replays past steps with these.
__device__ inline void adam_step_scalars(const PsAdamHyper& hp, int64_t step, float* step_size, float* inv_sbc2, float* lr_out) {
  const double t = (double)step;
  double lr = (double)hp.lr;
  if (hp.noam) lr = (double)hp.lr * fmin(pow(t, -0.5), t * pow((double)hp.warmup_steps, -1.5));
  const double bc1 = 1.0 - pow((double)hp.beta1, t);
  const double bc2 = 1.0 - pow((double)hp.beta2, t);
  *step_size = (float)(lr / bc1);
  *inv_sbc2 = (float)(1.0 / sqrt(bc2));
  *lr_out = (float)lr;
}
__device__ inline void adam_scalars(const PsAdamHyper& hp, float total_sumsq, int64_t step, float* out, float* norm_out) {
  const float norm = sqrtf(total_sumsq);
  float coef = 1.f;
  if (hp.max_grad_norm > 0.f) coef = fminf(hp.max_grad_norm / (norm + 1e-6f), 1.f);
  out[0] = coef * hp.grad_scale;
  adam_step_scalars(hp, step, &out[1], &out[2], &out[3]);
  *norm_out = norm;
}

struct AdamScal { float gmul, step_size, inv_sbc2, b1, b2, eps, wd; int zero_g; float lr; int method; };
#define PS_OPT_ADAM 0
#define PS_OPT_SGD 1
#define PS_OPT_ADAGRAD 2
#define PS_OPT_ADADELTA 3

// The rounding sequence is PINNED (no contraction left to the compiler, explicit fused multiply-adds where torch's kernels fuse):
// this function is inlined into the dense chunk loop, the row-sparse row loop and the lazy replay (optim_rows.hip), and the
// three must agree to the last bit (tests/test_gpu_lazy_exact.py) — left to the optimiser, `v*b2 + x` became an fma in one
// context and a multiply + add in another.
__device__ inline void adam_elem(const AdamScal& a, float& pp, float gg, float& mm, float& vv) {
#pragma clang fp contract(off)
  gg *= a.gmul;
  if (a.wd != 0.f) gg = __builtin_fmaf(a.wd, pp, gg);
  mm = __builtin_fmaf(gg - mm, 1.f - a.b1, mm);  // exp_avg.lerp_(grad, 1-beta1): a + w (b - a), fused
  vv = vv * a.b2 + ((1.f - a.b2) * gg) * gg;     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2): two roundings
  const float denom = sqrtf(vv) * a.inv_sbc2 + a.eps;
  pp = pp - a.step_size * (mm / denom);          // param.addcdiv_(exp_avg, denom, -step_size)
}

// The reference's other `--optim` methods (optimizers.py:175-183), torch's update rules with its defaults, after the same clip:
//   sgd       p -= lr (g + wd p)                                                              (torch.optim.SGD, no momentum)
//   adagrad   s1 += g g;  p -= lr g / (sqrt(s1) + 1e-10)                                      (lr_decay 0; s1 starts at adagrad_accum)
//   adadelta  s1 = 0.9 s1 + 0.1 g g;  dlt = sqrt(s2 + 1e-6) / sqrt(s1 + 1e-6) g;  s2 = 0.9 s2 + 0.1 dlt dlt;  p -= lr dlt
// lr = the noam rate of the step when --decay_method noam, else --lr (optimizers.py:231-236 sets the rate only under noam).
__device__ inline void other_elem(const AdamScal& a, float& pp, float gg, float& s1, float& s2) {
#pragma clang fp contract(off)
  gg *= a.gmul;
  if (a.wd != 0.f) gg = __builtin_fmaf(a.wd, pp, gg);
  if (a.method == PS_OPT_SGD) {
    pp = pp - a.lr * gg;
  } else if (a.method == PS_OPT_ADAGRAD) {
    s1 = __builtin_fmaf(gg, gg, s1);
    pp = pp - a.lr * (gg / (sqrtf(s1) + 1e-10f));
  } else {
    s1 = s1 * 0.9f + (0.1f * gg) * gg;
    const float dlt = (sqrtf(s2 + 1e-6f) / sqrtf(s1 + 1e-6f)) * gg;
    s2 = s2 * 0.9f + (0.1f * dlt) * dlt;
    pp = pp - a.lr * dlt;
  }
}
__device__ inline void other_update_chunk(const AdamChunkRec& rec, const AdamScal& a) {
  float* p = rec.p; float* g = rec.g; float* m = rec.m; float* v = rec.v;
  const bool s1 = a.method != PS_OPT_SGD, s2 = a.method == PS_OPT_ADADELTA;      // which state tensors exist
  for (int i = threadIdx.x; i < rec.n; i += 256) {
    float pp = p[i], gg = g[i], a1 = s1 ? m[i] : 0.f, a2 = s2 ? v[i] : 0.f;
    other_elem(a, pp, gg, a1, a2);
    p[i] = pp;
    if (s1) m[i] = a1;
    if (s2) v[i] = a2;
    if (a.zero_g && gg != 0.f) g[i] = 0.f;
  }
}

// clip + Adam over one ADAM_CHUNK of the plan.
__device__ inline void adam_update_chunk(const char* plan, int chunk, const AdamScal& a) {
  const AdamChunkRec rec = adam_chunk_recs(plan)[chunk];
  if (a.method != PS_OPT_ADAM) { other_update_chunk(rec, a); return; }       // block-uniform
  float* p = rec.p; float* g = rec.g; float* m = rec.m; float* v = rec.v;    // (advanced to the chunk)
  const int64_t beg = 0, end = rec.n;
  const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0 &&
                   end - beg == ADAM_CHUNK;
  if (vec) {
    float4* p4 = (float4*)(p + beg); float4* g4 = (float4*)(g + beg);
    float4* m4 = (float4*)(m + beg); float4* v4 = (float4*)(v + beg);
    const unsigned long long* mask = adam_chunk_mask(plan, chunk) + (threadIdx.x >> 6) * (ADAM_CHUNK / 4 / 256);
    const int lane = threadIdx.x & 63;
#pragma unroll 2
    for (int i = 0; i < ADAM_CHUNK / 4 / 256; ++i) {
      const int k = threadIdx.x + 256 * i;
      const bool nz = (mask[i] >> lane) & 1;        // what the sum-of-squares pass of this step saw in these 16 bytes
      float4 pp = p4[k], mm = m4[k], vv = v4[k];
      float4 gg = make_float4(0.f, 0.f, 0.f, 0.f);
      if (nz) gg = g4[k];
      // PsAdamHyper::zero_grads: the step's memset rides here — and only where there is something to clear (70 % of the
      // table rows of a C2 step hold no gradient)
      if (a.zero_g && nz) g4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      adam_elem(a, pp.x, gg.x, mm.x, vv.x); adam_elem(a, pp.y, gg.y, mm.y, vv.y);
      adam_elem(a, pp.z, gg.z, mm.z, vv.z); adam_elem(a, pp.w, gg.w, mm.w, vv.w);
      p4[k] = pp; m4[k] = mm; v4[k] = vv;
    }
  } else {
    for (int64_t i = beg + threadIdx.x; i < end; i += 256) {
      float pp = p[i], mm = m[i], vv = v[i];
      adam_elem(a, pp, g[i], mm, vv);
      p[i] = pp; m[i] = mm; v[i] = vv;
      if (a.zero_g && g[i] != 0.f) g[i] = 0.f;
    }
  }
}
