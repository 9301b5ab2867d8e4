// common.h — device helpers shared by the gfx950 kernels of libprodsearch_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "../../include/prodsearch_hip.h"

#define PS_WAVE 64

// ----------------------------------------------------------------- error plumbing
void ps_set_error(const char* fmt, ...);
#define PS_CHECK_HIP(expr)                                                        \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess) {                                                       \
      ps_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return PS_ERR_HIP;                                                          \
    }                                                                             \
  } while (0)
#define PS_REQUIRE(cond, ...)                                                     \
  do {                                                                            \
    if (!(cond)) {                                                                \
      ps_set_error(__VA_ARGS__);                                                  \
      return PS_ERR_ARG;                                                          \
    }                                                                             \
  } while (0)
#define PS_LAUNCH_CHECK() PS_CHECK_HIP(hipGetLastError())

static inline int ps_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------ environment switches
// ps_env_int: the SUPPORTED switches ("Switches"): alternative code paths with results of their own, every one
// covered by tools/env_matrix.sh or a test.  ps_diag_int: launch-shape tuning and timing-experiment knobs (some of the latter
// compute WRONG results on purpose: PS_MLP_DIAG, PS_RTM_DIAG, PS_X3D_DIAG) — they exist only in the diagnostic build
// (hipcc -DPS_DIAG: `python -m prodsearch_amd.build --diag` -> lib/libprodsearch_hip_diag.so, which prodsearch_amd._lib
// loads when PS_DIAG_LIB=1); in the shipped library the call folds to its default and a stray variable in a training
// environment changes nothing.
#include <stdlib.h>
static inline int ps_env_int(const char* name, int dflt) { const char* e = getenv(name); return (e && *e) ? atoi(e) : dflt; }
#ifdef PS_DIAG
static inline int ps_diag_int(const char* name, int dflt) { return ps_env_int(name, dflt); }
#define PS_DIAG_ON 1
#else
static inline int ps_diag_int(const char*, int dflt) { return dflt; }
#define PS_DIAG_ON 0
#endif

// ------------------------------------------------------------------ kernel timer (measurement only)
// ps_ktimer_arm("tag", n) makes the launch site of ONE tagged kernel attach a HIP event pair to that launch
// (hipExtLaunchKernelGGL start / stop events: the pair reads the DISPATCH's own begin / end timestamps — the duration
// rocprofv3's kernel trace reports for it, tools/micro/extlaunch.hip: 5.59 us against 5.75 — where a hipEventRecord pair
// around the launch also reads the two barrier packets, +2.5 us); bench.py's roofline: the kernel's in-step duration, measured
// live on the stream it is launched on.  ps_ktimer_read averages the pairs.  Unarmed (the default) a scope is one pointer
// compare and PS_KLAUNCH one flag test.  A scope times the FIRST PS_KLAUNCH inside it, nothing else it brackets.
const char* ps_ktimer_tag();                       // armed tag or nullptr
void ps_ktimer_scope(bool open);
bool ps_ktimer_take(hipEvent_t* e0, hipEvent_t* e1);   // inside an open scope, once: the launch's event pair
struct KTimeScope {
  bool on;
  KTimeScope(const char* t, hipStream_t) : on(false) {
    const char* armed = ps_ktimer_tag();
    if (armed && __builtin_strcmp(armed, t) == 0) { on = true; ps_ktimer_scope(true); }
  }
  ~KTimeScope() { if (on) ps_ktimer_scope(false); }
};
#define PS_KLAUNCH(kernel, grid, block, lds, st, ...)                                                        \
  do {                                                                                                       \
    hipEvent_t kt_e0_, kt_e1_;                                                                               \
    if (ps_ktimer_take(&kt_e0_, &kt_e1_)) hipExtLaunchKernelGGL(kernel, grid, block, lds, st, kt_e0_, kt_e1_, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                       \
  } while (0)

// ------------------------------------------------------------------- Philox4x32-10
// Counter-based RNG (Salmon et al., SC'11), the generator torch/curand use; 10 rounds.
// Dropout element (row, col) of site s at step t: see DropSpec below (two forms).  The oracle restates the same function
// (oracle/philox.py).
struct Philox4 {
  uint32_t x, y, z, w;
};

__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * c0;
    uint64_t p1 = (uint64_t)M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  Philox4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3;
  return o;
}

struct DropSpec {      // one dropout site
  uint32_t thr;        // keep iff u32 >= thr ; thr = floor(p * 2^32); 0 => disabled
  float scale;         // 1/(1-p)
  uint32_t site, step, k0, k1;
  const uint32_t* step_ptr;   // if set the step lives in device memory (graph replay: kernel arguments stay constant)
  uint32_t half;       // 1: the 16-bit column-shared form below (sites ctx / ff1 / ff2 of every layer)
  uint32_t thr16;      // keep iff u16 >= thr16 ; thr16 = floor(p * 2^16)
};
// `c ? a : b` on two specs by VALUE, field by field: a reference to one of two structs makes the compiler keep both in scratch
// memory and read the fields back through a pointer (rtm_embed_bwd_kernel: 56 B of scratch, loads inside its loop)
__host__ __device__ inline DropSpec drop_select(bool c, const DropSpec& a, const DropSpec& b) {
  DropSpec r;
  r.thr = c ? a.thr : b.thr; r.scale = c ? a.scale : b.scale; r.site = c ? a.site : b.site; r.step = c ? a.step : b.step;
  r.k0 = c ? a.k0 : b.k0; r.k1 = c ? a.k1 : b.k1; r.step_ptr = c ? a.step_ptr : b.step_ptr; r.half = c ? a.half : b.half;
  r.thr16 = c ? a.thr16 : b.thr16;
  return r;
}
// Two forms of the stream (both restated in oracle/philox.py, pinned through ps_dropout_mult_host):
//   classic (fs, attention, review and token sites): element (row, col) = word (row & 3) of Philox(col, row >> 2, site, step)
//     — four consecutive ROWS of a column share a call, one 32-bit word per decision;
//   half    (ctx, ff1, ff2: the [rows, d] / [rows, F] activations of the feed-forward tail, where the Philox work was
//     ~12 % of the fused kernels): element (row, col) = 16-bit half (col & 7) of Philox(col >> 3, row, site, step) — EIGHT
//     consecutive COLUMNS of a row share a call (a lane of the fused kernels owns one replica row and runs of consecutive
//     features), one 16-bit half per decision: keep probability 1 - floor(p * 65536) / 65536 (p = 0.1: 0.90001).
__host__ __device__ inline bool drop_site_is_half(uint32_t site) {
  const uint32_t k = (site - 1u) & 7u;
  return site >= 1u && site < 0x100u && k >= 1u && k <= 3u;
}
// the step a kernel keys its Philox counters with
__device__ inline uint32_t drop_step(const DropSpec& s) { return s.step_ptr ? *s.step_ptr : s.step; }
// set by an entry point for the duration of its argument building (see graph.h): where the step word lives, or null
inline const uint32_t*& ps_step_ptr_slot() { static thread_local const uint32_t* p = nullptr; return p; }

__host__ inline DropSpec make_drop(const PsTemDesc& d, uint32_t site) {
  DropSpec s;
  bool on = d.training && d.dropout > 0.f;
  double p = on ? (double)d.dropout : 0.0;
  s.thr = on ? (uint32_t)(p * 4294967296.0) : 0u;
  s.scale = on ? (float)(1.0 / (1.0 - p)) : 1.f;
  s.site = site; s.step = (uint32_t)d.step; s.step_ptr = ps_step_ptr_slot();
  s.k0 = (uint32_t)(d.seed & 0xffffffffu); s.k1 = (uint32_t)(d.seed >> 32);
  s.half = drop_site_is_half(site) ? 1u : 0u;
  s.thr16 = on ? (uint32_t)(p * 65536.0) : 0u;
  return s;
}

__device__ inline float drop_word(const DropSpec& s, uint32_t word) {
  return word >= s.thr ? s.scale : 0.f;
}
// half form: the call shared by columns 8*colgrp .. 8*colgrp+7 of `row`, and decision e (0..7) of it
__host__ __device__ inline Philox4 drop_call16(const DropSpec& s, uint32_t row, uint32_t colgrp, uint32_t step) {
  return philox4x32_10(colgrp, row, s.site, step, s.k0, s.k1);
}
__host__ __device__ inline float drop_half(const DropSpec& s, const Philox4& r, int e) {
  const uint32_t w = e < 2 ? r.x : (e < 4 ? r.y : (e < 6 ? r.z : r.w));
  const uint32_t hv = (e & 1) ? (w >> 16) : (w & 0xffffu);
  return hv >= s.thr16 ? s.scale : 0.f;
}
// multiplier (0 or 1/(1-p)) of element (row, col), either form
__device__ inline float drop_mult(const DropSpec& s, uint32_t row, uint32_t col) {
  if (s.thr == 0u) return 1.f;
  if (s.half) return drop_half(s, drop_call16(s, row, col >> 3, drop_step(s)), (int)(col & 7u));
  Philox4 r = philox4x32_10(col, row >> 2, s.site, drop_step(s), s.k0, s.k1);
  uint32_t sel = row & 3u;
  uint32_t wv = sel == 0 ? r.x : (sel == 1 ? r.y : (sel == 2 ? r.z : r.w));
  return drop_word(s, wv);
}

// site ids
#define PS_SITE_FS 0u
#define PS_SITE_ATTN(l) (1u + 8u * (uint32_t)(l) + 0u)
#define PS_SITE_CTX(l)  (1u + 8u * (uint32_t)(l) + 1u)
#define PS_SITE_FF1(l)  (1u + 8u * (uint32_t)(l) + 2u)
#define PS_SITE_FF2(l)  (1u + 8u * (uint32_t)(l) + 3u)
#define PS_SITE_SAMPLE_ITEM 0x40000000u
#define PS_SITE_SAMPLE_WORD 0x40000001u

// ------------------------------------------------------- division by a launch-invariant
// A runtime integer division costs ~40 VALU instructions on gfx950 and the row-wise kernels do
// several per element; q = mulhi(n, ceil(2^32/d)) is exact while n*d < 2^32 (all uses here).
struct FDiv {
  uint32_t d, m;
};
__host__ inline FDiv make_fdiv(int d) {
  FDiv f;
  f.d = (uint32_t)(d > 0 ? d : 1);
  f.m = f.d == 1 ? 0u : (uint32_t)((0x100000000ull + f.d - 1) / f.d);
  return f;
}
__device__ inline int fdiv(int n, const FDiv& f) { return f.d == 1 ? n : (int)__umulhi((uint32_t)n, f.m); }

// ------------------------------------------------------------------ strided sums of a global array
// p[first] + p[first + stride] + ... (index < n), added in that order: what `for (i = first; i < n; i += stride) s += p[i]` computes —
// but that loop is one load and one full wait per trip (hipcc neither unrolls a runtime trip count nor overlaps the trips' loads), i.e. a
// round trip of its own for every element a lane adds (round 5: 30 serial round trips in the embed launch's list workgroups, 24 in
// rtm_rowlist_kernel, 7 at the head of every workgroup of the dense Adam update, 8 in the fused forward's last arriver).  Here U
// loads are in flight per trip, unconditional from a clamped index; an element past the end enters with weight 0 through an fma
// (fma(v, 1, s) is s + v rounded once: bitwise the same sums; under `if (j < n) s += v` hipcc sinks the load beneath the test, 5f).
template <int U>
__device__ __forceinline__ float strided_sum_f32(const float* __restrict__ p, int n, int first, int stride) {
  float s = 0.f;
  for (int i = first; i < n; i += U * stride) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int j = i + u * stride; v[u] = p[j < n ? j : 0]; }
#pragma unroll
    for (int u = 0; u < U; ++u) s = __builtin_fmaf(v[u], i + u * stride < n ? 1.f : 0.f, s);
  }
  return s;
}
template <int U>
__device__ __forceinline__ int strided_sum_i32(const int* __restrict__ p, int n, int first, int stride) {
  int s = 0;
  for (int i = first; i < n; i += U * stride) {
    int v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int j = i + u * stride; v[u] = p[j < n ? j : 0]; }
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u] * (i + u * stride < n ? 1 : 0);
  }
  return s;
}

// ------------------------------------------------------------------ wave helpers
// Sum over the 64 lanes, returned to every lane.  DPP adds, not ds_bpermute shuffles: an inclusive scan inside each
// row of 16 lanes (row_shr 1, 2, 4, 8; lanes shifted in from outside the row contribute 0), then row_bcast:15 /
// row_bcast:31 carry the row totals into lane 63, which is broadcast through an SGPR.  Six ~8-cycle ALU ops instead of
// six LDS round trips in a dependent chain (a LayerNorm row needs two such sums back to back).
__device__ inline float wave_sum(float v) {
#define PS_DPP_ADD(ctrl, rmask) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
  PS_DPP_ADD(0x111, 0xf);      // row_shr:1
  PS_DPP_ADD(0x112, 0xf);      // row_shr:2
  PS_DPP_ADD(0x114, 0xf);      // row_shr:4
  PS_DPP_ADD(0x118, 0xf);      // row_shr:8   -> lane 15 of each row holds the row total
  PS_DPP_ADD(0x142, 0xa);      // row_bcast:15 into rows 1 and 3
  PS_DPP_ADD(0x143, 0xc);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave total
#undef PS_DPP_ADD
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// sum over groups of 32 consecutive lanes, valid in the LAST lane of each group (lanes 31 and 63): DPP scan as above
__device__ inline float half_sum_last(float v) {
#define PS_DPP_ADD(ctrl, rmask) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
  PS_DPP_ADD(0x111, 0xf);
  PS_DPP_ADD(0x112, 0xf);
  PS_DPP_ADD(0x114, 0xf);
  PS_DPP_ADD(0x118, 0xf);
  PS_DPP_ADD(0x142, 0xa);
#undef PS_DPP_ADD
  return v;
}
// sum over groups of `width` consecutive lanes (width power of two <= 64)
__device__ inline float group_sum(float v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// 1 / x as ONE v_rcp_f32 (1 ulp).  `__frcp_rn(x)` and `1.f / x` are the correctly rounded division on this compiler: v_div_scale x2 +
// v_rcp + 6 fma / mul + v_div_fmas + v_div_fixup = 11 vector instructions (round 5: 36 of them per wave of the fused per-replica
// kernels, one per GELU, a seventh of the wave's vector instructions).  Every caller below divides 1 by a value in [1, inf].
__device__ inline float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
// tanh via one v_exp_f32 + one v_rcp_f32: 1 - 2/(e^{2u}+1); saturates correctly at +-inf.
// |error| <= ~2e-7 absolute, far inside the 1e-4 budget and ~8x fewer instructions than tanhf.
__device__ inline float tanh_fast(float u) {
  const float e = __expf(2.f * u);
  return 1.f - 2.f * rcp_fast(e + 1.f);
}
// gelu (models/neural.py:7-8): 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3).  Computed in its logistic form —
// 0.5 (1 + tanh u) = 1 / (1 + exp(-2u)) — with the constants folded into the exponent of ONE v_exp_f32: 7 vector instructions
// per element instead of 12 (the fused per-replica kernels are vector-issue-bound and spend a fifth of their epilogue here,
// DESIGN.md 5d); the same function to a few ulp, saturating the same way (x -> -inf: x * 0; x -> +inf: x * 1).
#define PS_GELU_A 2.3022082f          /* 2 sqrt(2/pi) log2(e) */
#define PS_GELU_B 0.10294324f         /* PS_GELU_A * 0.044715 */
__device__ inline float gelu_sigmoid2u(float x, float x2) {      // 1 / (1 + exp(-2u)) = 0.5 (1 + tanh u)
  const float e = __builtin_amdgcn_exp2f(-x * fmaf(PS_GELU_B, x2, PS_GELU_A));
  return rcp_fast(1.f + e);
}
__device__ inline float gelu_tanh_f(float x) {
  return x * gelu_sigmoid2u(x, x * x);
}
// d gelu / dx = s + x s (1 - s) (2 du/dx) with s = 0.5 (1 + tanh u):  0.5 (1 + t) = s,  0.5 x (1 - t^2) du = 2 x s (1 - s) du
__device__ inline float gelu_tanh_grad(float x) {
  const float x2 = x * x;
  const float s = gelu_sigmoid2u(x, x2);
  const float du2 = fmaf(2.f * 0.7978845608028654f * 3.f * 0.044715f, x2, 2.f * 0.7978845608028654f);   // 2 du/dx
  return fmaf(s, x * (1.f - s) * du2, s);
}
__device__ inline float softplus_f(float x) {      // log(1+exp(x)), stable
  return fmaxf(x, 0.f) + log1pf(__expf(-fabsf(x)));
}
__device__ inline float sigmoid_f(float x) {
  const float e = __expf(-fabsf(x));
  const float r = rcp_fast(1.f + e);
  return x >= 0.f ? r : e * r;
}

// ------------------------------------------------------------------- GEMM launcher
enum { RES_NONE = 0, RES_DIRECT = 1, RES_GATHER = 2, RES_FANIN = 3 };
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_TANH = 2, ACT_GELU_BWD = 3, ACT_TANH_BWD = 4 };

struct ResMap {           // row map between replica rows and layer-input rows
  int mode;               // RES_*
  const float* ptr;       // residual source
  int ld;                 // row stride of ptr (floats)
  FDiv dSq, dfan, dS;     // fast dividers of Sq, fan, S (filled by res_finish)
  int Sq, fan, S, qpos;   // GATHER: out row m=(n_out,i) -> src row (n_out/fan)*S + (Sq==S ? i : qpos)
                          // FANIN : out row m=(n_in,pos) -> sum_j src row ((n_in*fan+j)*Sq + i), only q rows
  const float* extra;     // FANIN with Sq == 1: one more row per sequence added at the query position
  int extra_ld;           //   (the dQ.Wq term of the attention backward), [n_in, extra_ld]
  const float* extra2;    // second partial of the same term (the attention backward's two head groups), or null
};

__device__ inline float res_value(const ResMap& R, int row, int col) {
  if (R.mode == RES_DIRECT) return R.ptr[(size_t)row * R.ld + col];
  if (R.mode == RES_GATHER) {
    int nout = fdiv(row, R.dSq), i = row - nout * R.Sq;
    int src = fdiv(nout, R.dfan) * R.S + (R.Sq == R.S ? i : R.qpos);
    return R.ptr[(size_t)src * R.ld + col];
  }
  if (R.mode == RES_FANIN) {
    int nin = fdiv(row, R.dS), pos = row - nin * R.S;
    int i;
    if (R.Sq == R.S) i = pos;
    else if (pos == R.qpos) i = 0;
    else return 0.f;
    float s = 0.f;
    if (R.ptr)                                   // null: the fan-in sum already sits in extra / extra2
      for (int j = 0; j < R.fan; ++j) s += R.ptr[(size_t)((nin * R.fan + j) * R.Sq + i) * R.ld + col];
    if (R.extra) s += R.extra[(size_t)nin * R.extra_ld + col];
    if (R.extra2) s += R.extra2[(size_t)nin * R.extra_ld + col];
    return s;
  }
  return 0.f;
}

__host__ inline void res_finish(ResMap& R) {   // call after Sq/fan/S are set
  R.dSq = make_fdiv(R.Sq); R.dfan = make_fdiv(R.fan); R.dS = make_fdiv(R.S);
}

#define PS_WPLANES_MAX 8         // weights one WPlaneScope can hold
#define PS_GEMM_KIDX_MAX 2048   // reduction rows of ONE split a row-list weight gradient can map (LDS ints)

struct GemmProblem {
  const float* A; int lda; int ta;      // ta: A stored [K][M] (reduction index is the slow one)
  const float* Bseg[3]; int kseg;       // up to 3 B segments of kseg reduction rows each (nseg = ceil(K/kseg))
  int ldb; int tb;                      // tb==0: B[n][k] (nn.Linear), tb==1: B[k][n]
  float* C; int ldc;
  int M, N, K;
  const float* bias;                    // [N] or null
  float alpha;                          // applied after bias:  v = (acc + bias) * alpha
  int act;                              // ACT_*
  const float* act_aux;                 // ACT_*_BWD: pre-activation / activation, same shape+ld as C
  float* aux_out;                       // if set: pre-activation (acc+bias)*alpha stored here (ld = ldc)
  DropSpec drop;                        // dropout applied after act
  ResMap res;                           // residual added after dropout
  float* out2; int ld2; const float* add2;  // if set: out2[row*ld2+col] = v + add2[col] (second copy of the result)
  float* colsum;                        // if set: atomicAdd column sums of the final value into colsum[n]
  float* colsum_part;                   // if set (with colsum): park them instead, [4*row_tiles][3][N] slot 0 (ColFoldList layout)
  int accumulate;                       // 0: C = v ; 1: C += v (plain) ; 2: atomicAdd(C, v) (split reduction)
  int ksplit;                           // number of reduction splits (grid.z multiplier), >=1
  // optional row list (IDX instantiations of the kernel): only the *rcount rows ridx[0..] of the operands exist as
  // far as the product is concerned — the valid (non-pad) positions of the batch's sequences.
  //   ta == 0:            logical row i of A / C / residual is physical row ridx[i]; M is the list length on the device
  //                       (the host's M bounds it, and ridx has room for that many entries: the kernel may read past the length)
  //   ta == 1 && tb == 1: logical reduction row k of A and B is physical row ridx[k]; K is the list length
  const int32_t* ridx; const int32_t* rcount;
  int64_t split_stride;                 // split s of a split reduction writes at C + s * split_stride (deterministic mode: every split
                                        // stores its own partial matrix, a second launch adds them up in split order)
  int no_deep;                          // never pick the 128-deep-slab form (133 KB of LDS per workgroup): for small products that
                                        // run BESIDE other launches, where that footprint starves them of CUs
  // filled by the launcher (gemm.hip, WPlaneScope): the B segments as pre-split bf16 plane matrices [3][N][kseg] (hi / mid / lo,
  // reduction index contiguous whatever tb says), or null — gemm_x3w_kernel reads B from them
  const uint16_t* bpl[3];
};

struct GemmGroup {
  GemmProblem p[4];      // (the flat form below holds at most three)
  int n;
  // a fork of the side stream signalled by THIS launch: its first workgroup stores sigval to *sig as it starts (side_take_signal)
  uint32_t* sig; uint32_t sigval;
  // flat form (weight gradients of different shapes in one launch): grid.x walks  sum_i tiles_i * ksplit_i  workgroups,
  // problem i owns [flat0[i], flat0[i+1]): split-major, then row tile, then column tile; each problem keeps its own ksplit
  int flat;
  int flat0[4], flat_tm[3], flat_tiles[3];
  // flat_xcd: workgroup L runs on XCD L % 8 (round-robin dispatch); with every flat0[] and ksplit a multiple of 8 the decode puts ALL
  // tiles of a reduction split on one XCD (split % 8 == L % 8), so the split's operand rows cross the fabric once per XCD instead of
  // once per tile pair (C2's grouped W2 / W1 / Wo gradients: FETCH_SIZE x2 = 143 MB per launch for 49.5 MB of operands, PMC)
  int flat_xcd;
  int split_xcd;      // the same placement on the 3-D grid (set by ps_launch_gemm: ta == 1, ksplit a multiple of 8)
  // weight-gradient launches whose caller sized ksplit for 128x128 tiles: take gemm_x3d_kernel (tem.hip, run_wgrads)
  int prefer_x3d;
  // diagnostics (PS_GEMM_STAMP=1 + ps_debug_set_stamp_buffer, tools/gemm_stamps.py): the waves of one mid-grid workgroup of the
  // bf16x3 kernel record s_memtime at their phase boundaries, 32 slots per wave
  unsigned long long* stamp;
};
unsigned long long* ps_debug_stamp_ptr();

int ps_launch_gemm(const GemmGroup& g, hipStream_t stream);
// Fork of the side stream without a stream operation on the main stream (tem.hip, side_fork): the side stream waits for a
// sequence value, and the NEXT kernel launched on the main stream stores it as its first workgroup starts — every earlier
// main-stream kernel has completed by then (in-order stream), which is all a fork promises.  A launcher that can carry the
// signal asks for it right before its launch; side_repend_signal hands it back when the launch did not happen (the join
// flushes an unclaimed signal with a stream write).  The write-value operation this replaces cost the main stream ~5 us
// between two dependent kernels, twice per backward.
bool side_take_signal(hipStream_t st, uint32_t** flag, uint32_t* val);
void side_repend_signal(hipStream_t st, uint32_t val);
__device__ __forceinline__ void fork_signal(uint32_t* sig, uint32_t val) {
  if (sig && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
    __hip_atomic_store(sig, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// PS_DETERMINISTIC=1 / ps_set_deterministic(1): bitwise run-to-run reproducible TEM training steps — one stream,
// weight gradients through per-split partials + an ordered sum, table scatters by sole-owner waves walking the tasks in order.
bool ps_deterministic();
// library-owned scratch: per device and slot (0: deterministic split reductions, 1: deterministic score backward, 2: the weight
// planes of WPlaneScope), grow-only
float* ps_det_scratch(int slot, size_t floats, hipStream_t st);
// For its lifetime the listed fp32 weight matrices ([rows][cols], nn.Linear layout, row stride = cols) exist as bf16x3 plane
// images in both orientations — split ONCE per entry-point call by one small launch on `st` — and ps_launch_gemm routes
// products whose B operand is one of them (ta == 0, whole 32-deep slabs) to gemm_x3w_kernel, which no longer splits B at all
// (and reads the transposed orientation for tb == 1: the dX products stop paying the row-contiguous loader).  Not nestable;
// thread-local; nothing outlives the scope, so a weight the optimizer has moved is never multiplied through stale planes.
bool gemm_x3w_on();
struct WPlaneScope {
  WPlaneScope(hipStream_t st, const float* const* w, const int* rows, const int* cols, int n);
  ~WPlaneScope();
  bool on;
};
