// x3frag.h — the transposed bf16x3 product step shared by the fused per-replica kernels (mlp_fused.hip) and the fused K / V
// projection + attention forward (attn_sq1.hip): D^T = W . X^T with the WEIGHTS as the A operand of v_mfma_f32_32x32x16_bf16,
// streamed global -> registers in fragment order (WSplit, rowwise.h), and 32 activation rows as the B operand, held in LDS as
// three bf16 planes [row][k] whose 16-byte chunks are XOR-swizzled by the row (conflict-free ds_read_b128 / ds_write_b128).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define X3_ROWS 32        // activation rows of a product (the MFMA's N dimension)
#define X3_K 128          // reduction depth of the planes (= the model width these kernels are specialised for)

__device__ __forceinline__ int xa_off(int row, int k) { return row * X3_K + ((((k >> 3) ^ (row & 15)) << 3) | (k & 7)); }

// exact three-way bf16 split of 8 fp32 values -> one 16-byte chunk per plane
__device__ __forceinline__ void split8(const float (&v)[8], uint4 (&pl)[3]) {
  uint32_t w[3][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint16_t hb[2][3];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float x = v[2 * i + e];
      const __bf16 bh = (__bf16)x;
      float r = x - (float)bh;
      const __bf16 bm = (__bf16)r;
      r -= (float)bm;
      const __bf16 bl = (__bf16)r;
      hb[e][0] = __builtin_bit_cast(uint16_t, bh); hb[e][1] = __builtin_bit_cast(uint16_t, bm); hb[e][2] = __builtin_bit_cast(uint16_t, bl);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) w[p][i] = (uint32_t)hb[0][p] | ((uint32_t)hb[1][p] << 16);
  }
#pragma unroll
  for (int p = 0; p < 3; ++p) pl[p] = make_uint4(w[p][0], w[p][1], w[p][2], w[p][3]);
}
__device__ __forceinline__ void put8(uint16_t (*planes)[X3_ROWS * X3_K], int row, int k0, const float (&v)[8]) {
  uint4 pl[3];
  split8(v, pl);
  const int off = xa_off(row, k0);
#pragma unroll
  for (int p = 0; p < 3; ++p) *reinterpret_cast<uint4*>(&planes[p][off]) = pl[p];
}
// six bf16 MFMAs = one exact-fp32-grade 32x32x16 product step (small terms first); a = weight planes, b = activation planes
__device__ __forceinline__ void x3_mma(f32x16& acc, const uint4 (&a)[3], const uint4 (&b)[3]) {
#define BF(x) __builtin_bit_cast(bf16x8, x)
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[0]), BF(b[2]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[2]), BF(b[0]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[1]), BF(b[1]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[0]), BF(b[1]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[1]), BF(b[0]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[0]), BF(b[0]), acc, 0, 0, 0);
#undef BF
}
// one product step's weight fragment: three planes of 64 lanes x 16 bytes, contiguous (WSplit, fragment order).
// The main phase is one fully unrolled basic block; left alone, hipcc's scheduler sinks these refills next to the MFMAs that
// consume them and waits vmcnt(0..1) in front of every product step — an effective prefetch distance of ONE step whatever PF
// says (round-3 ISA listing).  Every step therefore ends in a sched_barrier: the refill stays where it is issued, PF steps
// ahead of its use, and the waits become the counted vmcnt(3 (PF - 1)) they should be.
__device__ __forceinline__ void load_frag(uint4 (&f)[3], const uint16_t* __restrict__ stream, int step, int lane) {
  const uint4* q = reinterpret_cast<const uint4*>(stream) + (size_t)step * 192 + lane;
  f[0] = q[0]; f[1] = q[64]; f[2] = q[128];
}
__device__ __forceinline__ void load_frag2(uint4 (&f)[3], const uint16_t* __restrict__ stream, int step, int lane) {   // DIAG
  const uint4* q = reinterpret_cast<const uint4*>(stream) + (size_t)step * 192 + lane;
  f[0] = q[0]; f[1] = q[64];
}
__device__ __forceinline__ void read_b(uint4 (&b)[3], const uint16_t (*planes)[X3_ROWS * X3_K], int l31, int k0) {
  const int off = xa_off(l31, k0);
#pragma unroll
  for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const uint4*>(&planes[p][off]);
}

// planes with fewer than 32 rows (reads past the last row land in whatever follows them: see attn_bwd_wf4_kernel)
template <int ROWS>
__device__ __forceinline__ void read_b_rows(uint4 (&b)[3], const uint16_t (*planes)[ROWS * X3_K], int l31, int k0) {
  const int off = xa_off(l31, k0);
#pragma unroll
  for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const uint4*>(&planes[p][off]);
}
// exact three-way bf16 split of 4 fp32 values -> one 8-byte half chunk per plane (k0 a multiple of 4)
template <int ROWS>
__device__ __forceinline__ void put4(uint16_t (*planes)[ROWS * X3_K], int row, int k0, const float4& v4) {
  const float v[4] = {v4.x, v4.y, v4.z, v4.w};
  uint32_t w[3][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    uint16_t hb[2][3];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float x = v[2 * i + e];
      const __bf16 bh = (__bf16)x;
      float r = x - (float)bh;
      const __bf16 bm = (__bf16)r;
      r -= (float)bm;
      const __bf16 bl = (__bf16)r;
      hb[e][0] = __builtin_bit_cast(uint16_t, bh); hb[e][1] = __builtin_bit_cast(uint16_t, bm); hb[e][2] = __builtin_bit_cast(uint16_t, bl);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) w[p][i] = (uint32_t)hb[0][p] | ((uint32_t)hb[1][p] << 16);
  }
  const int off = xa_off(row, k0);
#pragma unroll
  for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(&planes[p][off]) = make_uint2(w[p][0], w[p][1]);
}
