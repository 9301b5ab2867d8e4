// optim_rows.hip — row-sparse path of the optimizer for tables too large to stream densely
// (SURVEY.md §8b ps_coalesce_rows / ps_adam_rowsparse, §8d config 5: 50 M × 256 table).
//
// The embedding gradients stay DENSE tensors, as nn.Embedding(sparse=False) gives the reference
// (models/item_transformer.py:46,70); what becomes sparse is who touches them:
//   ps_coalesce_rows      sorted unique non-pad row ids of a step's index lists.  Bitmap of one bit per
//                         table row (atomicOr), popcount per 256-word segment, ordered emit that also
//                         clears the bitmap.  Deterministic, no sort, 3 launches, 6 MB of bitmap at 50 M rows.
//   ps_gather_rows        values[u,:] = grad[rows[u],:]  (payload of the sparse gradient exchange)
//   ps_scatter_rows       grad[rows[u],:] = values[u,:]  (merged gradient back into the dense tensor)
//   ps_clip_adam_rowsparse  global-norm clip over (dense small tensors + touched rows) and Adam on exactly
//                         those; touched gradient rows are re-zeroed in the same pass, so zero_grad never
//                         streams the table.  Untouched rows keep p, m, v (the lazy "SparseAdam" rule;
//                         dense Adam would keep decaying their moments — DESIGN.md §5b).
#include "optim_core.h"

#define CO_WORDS_PER_THREAD 1
#define CO_WORDS_PER_BLOCK (256 * CO_WORDS_PER_THREAD)
#define PS_MAX_IDX_LISTS 8

struct IdxLists { PsIdxList l[PS_MAX_IDX_LISTS]; int32_t n; int64_t total; };

__global__ __launch_bounds__(256) void co_mark_kernel(IdxLists L, int64_t n_rows, int64_t pad_row,
                                                      unsigned long long* bitmap, int32_t* bad) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= L.total) return;
  int k = 0;
  while (k < L.n - 1 && i >= L.l[k].n) { i -= L.l[k].n; ++k; }
  const int64_t r = L.l[k].idx[i];
  if (r == pad_row) return;
  if (r < 0 || r >= n_rows) { *bad = 1; return; }
  const unsigned long long bit = 1ull << (r & 63);
  // popular rows (Zipf words) repeat thousands of times: skip the atomic once the bit is visible
  if (__builtin_nontemporal_load(&bitmap[r >> 6]) & bit) return;
  atomicOr(&bitmap[r >> 6], bit);
}

__global__ __launch_bounds__(256) void co_count_kernel(const unsigned long long* bitmap, int64_t n_words,
                                                       int32_t* blocksum) {
  __shared__ float shf[4];
  (void)shf;
  __shared__ int sh[4];
  const int64_t w0 = (int64_t)blockIdx.x * CO_WORDS_PER_BLOCK + (int64_t)threadIdx.x * CO_WORDS_PER_THREAD;
  int c = 0;
#pragma unroll
  for (int j = 0; j < CO_WORDS_PER_THREAD; ++j)
    if (w0 + j < n_words) c += __popcll(bitmap[w0 + j]);
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) blocksum[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void co_emit_kernel(unsigned long long* bitmap, int64_t n_words,
                                                      const int32_t* blocksum, int64_t* rows, int64_t cap,
                                                      int32_t* count_out, int32_t* bad) {
  __shared__ int sh[256];
  __shared__ int base_sh;
  // rows emitted by earlier blocks
  int part = 0;
  part = strided_sum_i32<4>(blocksum, (int)blockIdx.x, threadIdx.x, 256);
  sh[threadIdx.x] = part;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) base_sh = sh[0];
  __syncthreads();
  const int base = base_sh;
  __syncthreads();
  const int64_t w0 = (int64_t)blockIdx.x * CO_WORDS_PER_BLOCK + (int64_t)threadIdx.x * CO_WORDS_PER_THREAD;
  unsigned long long w[CO_WORDS_PER_THREAD];
  int c = 0;
#pragma unroll
  for (int j = 0; j < CO_WORDS_PER_THREAD; ++j) {
    w[j] = (w0 + j < n_words) ? bitmap[w0 + j] : 0ull;
    c += __popcll(w[j]);
  }
  // exclusive scan of c over the block (Hillis-Steele in LDS)
  sh[threadIdx.x] = c;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    int add = ((int)threadIdx.x >= o) ? sh[threadIdx.x - o] : 0;
    __syncthreads();
    sh[threadIdx.x] += add;
    __syncthreads();
  }
  int64_t pos = (int64_t)base + sh[threadIdx.x] - c;
  if (c) {
#pragma unroll
    for (int j = 0; j < CO_WORDS_PER_THREAD; ++j) {
      unsigned long long x = w[j];
      while (x) {
        const int b = __ffsll((long long)x) - 1;
        x &= x - 1;
        if (pos < cap) rows[pos] = (w0 + j) * 64 + b; else *bad = 2;
        ++pos;
      }
      if (w[j]) bitmap[w0 + j] = 0ull;          // leave the bitmap clean for the next step
    }
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) *count_out = base + sh[255];
}

static inline int64_t co_words(int64_t n_rows) { return (n_rows + 63) / 64; }
static inline int64_t co_blocks(int64_t n_rows) { return (co_words(n_rows) + CO_WORDS_PER_BLOCK - 1) / CO_WORDS_PER_BLOCK; }

// workspace: bitmap words | blocksum int32[blocks] | bad flag int32 — zero it ONCE; calls leave it zeroed.
extern "C" int64_t ps_coalesce_ws_bytes(int64_t n_rows) {
  if (n_rows <= 0) return 0;
  return 8 * co_words(n_rows) + 4 * co_blocks(n_rows) + 16;
}

extern "C" int ps_coalesce_rows(const PsIdxList* lists_host, int32_t n_lists, int64_t n_rows, int64_t pad_row,
                                void* ws_dev, int64_t* rows_out_dev, int64_t cap, int32_t* count_out_dev,
                                ps_stream_t stream) {
  PS_REQUIRE(lists_host && n_lists > 0 && n_lists <= PS_MAX_IDX_LISTS, "coalesce: 1..%d index lists", PS_MAX_IDX_LISTS);
  PS_REQUIRE(n_rows > 0 && n_rows < ((int64_t)1 << 37) && ws_dev && rows_out_dev && count_out_dev && cap > 0,
             "coalesce: bad argument");
  IdxLists L;
  L.n = n_lists; L.total = 0;
  for (int i = 0; i < n_lists; ++i) {
    PS_REQUIRE(lists_host[i].n >= 0 && (lists_host[i].n == 0 || lists_host[i].idx), "coalesce: list %d null", i);
    L.l[i] = lists_host[i];
    L.total += lists_host[i].n;
  }
  const int64_t need = L.total < n_rows ? L.total : n_rows;
  PS_REQUIRE(cap >= need, "coalesce: rows_out capacity %lld < %lld", (long long)cap, (long long)need);
  PS_REQUIRE(L.total < ((int64_t)1 << 31), "coalesce: too many indices");
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* bitmap = (unsigned long long*)ws_dev;
  const int64_t nw = co_words(n_rows), nb = co_blocks(n_rows);
  int32_t* blocksum = (int32_t*)(bitmap + nw);
  int32_t* bad = blocksum + nb;
  if (L.total > 0) {
    hipLaunchKernelGGL(co_mark_kernel, dim3((unsigned)((L.total + 255) / 256)), dim3(256), 0, st, L, n_rows, pad_row,
                       bitmap, bad);
    PS_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(co_count_kernel, dim3((unsigned)nb), dim3(256), 0, st, bitmap, nw, blocksum);
  PS_LAUNCH_CHECK();
  hipLaunchKernelGGL(co_emit_kernel, dim3((unsigned)nb), dim3(256), 0, st, bitmap, nw, blocksum, rows_out_dev, cap,
                     count_out_dev, bad);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

// ---------------------------------------------------------------- gather / scatter of touched rows
// 32 lanes per row, 16 B per lane; 8 rows per 256-thread block.
template <int MODE>   // 0: vals = tab[rows]   1: tab[rows] = vals
__global__ __launch_bounds__(256) void rows_copy_kernel(float* tab, const int64_t* rows, const int32_t* count,
                                                        int64_t fixed_count, float* vals, int d) {
  const int64_t n = count ? (int64_t)*count : fixed_count;
  const int64_t u = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (u >= n) return;
  const int lane = threadIdx.x & 31;
  const int64_t r = rows[u];
  float4* v4 = (float4*)(vals + u * (int64_t)d);
  if (r < 0) {                      // padding entry of a fixed-capacity list (row-sharded tables): zeros out, nothing in
    if (MODE == 0) for (int j = lane; j < d / 4; j += 32) v4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  float4* t4 = (float4*)(tab + r * (int64_t)d);
  for (int j = lane; j < d / 4; j += 32) {
    if (MODE == 0) v4[j] = t4[j]; else t4[j] = v4[j];
  }
}

static int rows_copy(int mode, float* tab, const int64_t* rows, const int32_t* count, int64_t cap, float* vals,
                     int32_t d, ps_stream_t stream) {
  PS_REQUIRE(tab && rows && vals && cap >= 0 && d > 0 && d % 4 == 0, "rows copy: bad argument");
  PS_REQUIRE(((((uintptr_t)tab) | ((uintptr_t)vals)) & 15) == 0, "rows copy: 16-byte alignment");
  if (cap == 0) return PS_OK;
  const unsigned grid = (unsigned)((cap + 7) / 8);
  if (mode == 0)
    hipLaunchKernelGGL(rows_copy_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, tab, rows, count, cap, vals, d);
  else
    hipLaunchKernelGGL(rows_copy_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, tab, rows, count, cap, vals, d);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

extern "C" int ps_gather_rows(const float* table_dev, int32_t d, const int64_t* rows_dev, const int32_t* count_dev,
                              int64_t cap, float* values_out_dev, ps_stream_t stream) {
  return rows_copy(0, (float*)table_dev, rows_dev, count_dev, cap, values_out_dev, d, stream);
}

extern "C" int ps_scatter_rows(float* table_dev, int32_t d, const int64_t* rows_dev, const int32_t* count_dev,
                               int64_t cap, const float* values_dev, ps_stream_t stream) {
  return rows_copy(1, table_dev, rows_dev, count_dev, cap, (float*)values_dev, d, stream);
}

// the `bad` word of a coalesce workspace: 0 ok, 1 an index outside [0, n_rows), 2 more unique rows than `cap`
extern "C" const int32_t* ps_coalesce_bad_flag(void* ws_dev, int64_t n_rows) {
  if (!ws_dev || n_rows <= 0) return nullptr;
  unsigned long long* bitmap = (unsigned long long*)ws_dev;
  return (const int32_t*)(bitmap + co_words(n_rows)) + co_blocks(n_rows);
}

// ---------------------------------------------------------------- data-parallel exchange of touched rows
// One rank's wire format for one table: msg_rows[u] = row id (u < count, ascending) or -1 (u >= count), and
// msg_vals[u, :] = its gradient row (zeros past count, so the bytes on the wire are a function of the gradient only).
// Fixed capacity `cap` (the step's index count, a function of the batch SHAPE): no size ever crosses to the host.
__global__ __launch_bounds__(256) void rows_pack_kernel(const float* grad, const int64_t* rows, const int32_t* count,
                                                        int64_t cap, int64_t* msg_rows, float* msg_vals, int d) {
  const int64_t n = (int64_t)*count;
  const int64_t u = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (u >= cap) return;
  const int lane = threadIdx.x & 31;
  const bool live = u < n;
  const int64_t r = live ? rows[u] : -1;
  if (lane == 0) msg_rows[u] = r;
  const float4* g4 = (const float4*)(grad + (live ? r : 0) * (int64_t)d);
  float4* v4 = (float4*)(msg_vals + u * (int64_t)d);
  for (int j = lane; j < d / 4; j += 32) v4[j] = live ? g4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
}

extern "C" int ps_pack_rows(const float* grad_dev, int32_t d, const int64_t* rows_dev, const int32_t* count_dev,
                            int64_t cap, int64_t* msg_rows_dev, float* msg_vals_dev, ps_stream_t stream) {
  PS_REQUIRE(grad_dev && rows_dev && count_dev && msg_rows_dev && msg_vals_dev && cap > 0 && d > 0 && d % 4 == 0,
             "pack_rows: bad argument");
  PS_REQUIRE(((((uintptr_t)grad_dev) | ((uintptr_t)msg_vals_dev)) & 15) == 0, "pack_rows: 16-byte alignment");
  hipLaunchKernelGGL(rows_pack_kernel, dim3((unsigned)((cap + 7) / 8)), dim3(256), 0, (hipStream_t)stream, grad_dev,
                     rows_dev, count_dev, cap, msg_rows_dev, msg_vals_dev, d);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

// grad[u_rows[u], :] = sum over ranks r = 0 .. world-1, IN RANK ORDER, of the row rank r sent for that id (if any):
// every rank computes bit-identical sums from the same all-gathered messages, so the replicas stay in lock step.
// 32 lanes per union row: lane r < world binary-searches rank r's sorted id list, then all lanes add the found rows.
#define PS_MERGE_MAX_WORLD 32
__global__ __launch_bounds__(256) void rows_merge_kernel(const int64_t* all_rows, const float* all_vals, int world,
                                                         int64_t cap, int d, float* grad, const int64_t* u_rows,
                                                         const int32_t* u_count) {
  const int64_t u = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  const int lane = threadIdx.x & 31;
  const bool live = u < (int64_t)*u_count;          // uniform per 32-lane group
  int64_t found = -1;
  const int64_t row = live ? u_rows[u] : -1;
  if (live && lane < world) {
    const int64_t* lst = all_rows + (size_t)lane * cap;   // ascending ids, then -1 padding (= +infinity here)
    int64_t lo = 0, hi = cap;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      const int64_t v = lst[mid];
      if (v >= 0 && v < row) lo = mid + 1; else hi = mid;
    }
    if (lo < cap && lst[lo] == row) found = lo;
  }
  if (!live) return;
  float4* g4 = (float4*)(grad + row * (int64_t)d);
  for (int j = lane; j < d / 4; j += 32) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < world; ++r) {
      const int64_t pos = __shfl(found, (threadIdx.x & 32) + r, 64);
      if (pos >= 0) {
        const float4 x = ((const float4*)(all_vals + ((size_t)r * cap + pos) * d))[j];
        acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
      }
    }
    g4[j] = acc;
  }
}

extern "C" int ps_merge_rows(const int64_t* all_rows_dev, const float* all_vals_dev, int32_t world, int64_t cap,
                             int32_t d, float* grad_dev, const int64_t* union_rows_dev, const int32_t* union_count_dev,
                             int64_t union_cap, ps_stream_t stream) {
  PS_REQUIRE(all_rows_dev && all_vals_dev && grad_dev && union_rows_dev && union_count_dev, "merge_rows: null argument");
  PS_REQUIRE(world > 0 && world <= PS_MERGE_MAX_WORLD && cap > 0 && union_cap > 0 && d > 0 && d % 4 == 0,
             "merge_rows: world %d (<= %d), cap %lld, d %d", world, PS_MERGE_MAX_WORLD, (long long)cap, d);
  PS_REQUIRE(((((uintptr_t)grad_dev) | ((uintptr_t)all_vals_dev)) & 15) == 0, "merge_rows: 16-byte alignment");
  hipLaunchKernelGGL(rows_merge_kernel, dim3((unsigned)((union_cap + 7) / 8)), dim3(256), 0, (hipStream_t)stream,
                     all_rows_dev, all_vals_dev, world, cap, d, grad_dev, union_rows_dev, union_count_dev);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

// ---------------------------------------------------------------- row-sparse clip + Adam
#define RS_MAX_TABLES 4
#define RS_ROWS_PER_BLOCK 8
struct RowTables { PsRowTable t[RS_MAX_TABLES]; int32_t blk0[RS_MAX_TABLES + 1]; int32_t n; };

__device__ inline int rs_find(const RowTables& T, int blk) {
  int k = 0;
  while (k < T.n - 1 && blk >= T.blk0[k + 1]) ++k;
  return k;
}

// grid = n_chunks dense chunks, then the row blocks of every table.  partial[block] = its sum of squares.
__global__ __launch_bounds__(256) void rs_sumsq_kernel(const char* plan, int n_chunks, RowTables T, float grad_scale,
                                                       int64_t* state, float* partial) {
  __shared__ float sh[4];
  const int blk = blockIdx.x;
  float s;
  if (blk < n_chunks) {
    s = adam_sumsq_chunk(plan, blk, grad_scale, sh);
  } else {
    const int k = rs_find(T, blk - n_chunks);
    const PsRowTable& tb = T.t[k];
    const int64_t u = (int64_t)(blk - n_chunks - T.blk0[k]) * RS_ROWS_PER_BLOCK + (threadIdx.x >> 5);
    const int64_t n = *tb.count;
    float acc = 0.f;
    if (u < n) {
      const float4* g4 = (const float4*)(tb.g + tb.rows[u] * (int64_t)tb.d);
      for (int j = threadIdx.x & 31; j < tb.d / 4; j += 32) {
        float4 x = g4[j];
        x.x *= grad_scale; x.y *= grad_scale; x.z *= grad_scale; x.w *= grad_scale;
        acc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
      }
    }
    s = block_sum_256(acc, sh);
  }
  if (threadIdx.x == 0) {
    partial[blk] = s;
    if (blk == 0) state[0] += 1;
  }
}

// one block: fixed-order reduction of every partial -> clip coefficient and step scalars.
__global__ __launch_bounds__(1024) void rs_finalize_kernel(const PsAdamHyper hp, const int64_t* state,
                                                           const float* partial, int n_partial, float* scal,
                                                           float* gnorm_out) {
  __shared__ float sh[16];
  float s = strided_sum_f32<8>(partial, n_partial, threadIdx.x, 1024);       // (one block ends the clip norm: eight loads in flight per trip)
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float total = 0.f;
    for (int i = 0; i < 16; ++i) total += sh[i];
    float norm;
    adam_scalars(hp, total, state[0], scal, &norm);
    if (gnorm_out) { gnorm_out[0] = norm; gnorm_out[1] = scal[3]; }
  }
}

__global__ __launch_bounds__(256) void rs_update_kernel(const char* plan, int n_chunks, RowTables T,
                                                        const PsAdamHyper hp, const float* scal) {
  const AdamScal a = {scal[0], scal[1], scal[2], hp.beta1, hp.beta2, hp.eps, hp.weight_decay, 0, scal[3], hp.method};
  const int blk = blockIdx.x;
  if (blk < n_chunks) { adam_update_chunk(plan, blk, a); return; }
  const int k = rs_find(T, blk - n_chunks);
  const PsRowTable& tb = T.t[k];
  const int64_t u = (int64_t)(blk - n_chunks - T.blk0[k]) * RS_ROWS_PER_BLOCK + (threadIdx.x >> 5);
  if (u >= (int64_t)*tb.count) return;
  const int64_t off = tb.rows[u] * (int64_t)tb.d;
  float4* p4 = (float4*)(tb.p + off); float4* g4 = (float4*)(tb.g + off);
  float4* m4 = (float4*)(tb.m + off); float4* v4 = (float4*)(tb.v + off);
  for (int j = threadIdx.x & 31; j < tb.d / 4; j += 32) {
    float4 pp = p4[j], gg = g4[j], mm = m4[j], vv = v4[j];
    adam_elem(a, pp.x, gg.x, mm.x, vv.x); adam_elem(a, pp.y, gg.y, mm.y, vv.y);
    adam_elem(a, pp.z, gg.z, mm.z, vv.z); adam_elem(a, pp.w, gg.w, mm.w, vv.w);
    p4[j] = pp; m4[j] = mm; v4[j] = vv;
    g4[j] = make_float4(0.f, 0.f, 0.f, 0.f);     // zero_grad of the touched row, fused
  }
}

static int rs_pack(const PsRowTable* tabs, int32_t n_tables, RowTables* T) {
  PS_REQUIRE(n_tables >= 0 && n_tables <= RS_MAX_TABLES && (n_tables == 0 || tabs), "rowsparse: 0..%d tables", RS_MAX_TABLES);
  T->n = n_tables;
  int64_t b = 0;
  for (int i = 0; i < n_tables; ++i) {
    const PsRowTable& t = tabs[i];
    PS_REQUIRE(t.p && t.g && t.m && t.v && t.rows && t.count && t.d > 0 && t.d % 4 == 0 && t.cap >= 0,
               "rowsparse: table %d bad field", i);
    PS_REQUIRE(((((uintptr_t)t.p) | ((uintptr_t)t.g) | ((uintptr_t)t.m) | ((uintptr_t)t.v)) & 15) == 0,
               "rowsparse: table %d not 16-byte aligned", i);
    T->t[i] = t;
    T->blk0[i] = (int32_t)b;
    b += (t.cap + RS_ROWS_PER_BLOCK - 1) / RS_ROWS_PER_BLOCK;
    PS_REQUIRE(b < (1 << 30), "rowsparse: too many rows");
  }
  for (int i = n_tables; i <= RS_MAX_TABLES; ++i) T->blk0[i] = (int32_t)b;
  return PS_OK;
}

// floats of scratch after the 2 int64 of state: 4 scalars + one partial per block.
extern "C" int64_t ps_adam_rowsparse_state_floats(int32_t n_chunks, const PsRowTable* tables_host, int32_t n_tables) {
  RowTables T;
  if (rs_pack(tables_host, n_tables, &T) != PS_OK) return -1;
  return 4 + (int64_t)n_chunks + T.blk0[RS_MAX_TABLES];
}

extern "C" int ps_clip_adam_rowsparse(const void* plan_dev, int32_t n_chunks, const PsRowTable* tables_host,
                                      int32_t n_tables, const PsAdamHyper* hyper, int64_t* state_dev,
                                      float* gnorm_out_dev, ps_stream_t stream) {
  PS_REQUIRE(hyper && hyper->method == 0, "ps_clip_adam_rowsparse: the row-sparse optimizer is Adam only (method %d)", hyper ? hyper->method : -1);
  PS_REQUIRE(hyper && state_dev && n_chunks >= 0 && (n_chunks == 0 || plan_dev), "clip_adam_rowsparse: bad argument");
  RowTables T;
  int rc = rs_pack(tables_host, n_tables, &T);
  if (rc != PS_OK) return rc;
  const int n_blocks = n_chunks + T.blk0[RS_MAX_TABLES];
  PS_REQUIRE(n_blocks > 0, "clip_adam_rowsparse: nothing to update");
  hipStream_t st = (hipStream_t)stream;
  float* scal = (float*)(state_dev + 2);
  float* partial = scal + 4;
  PsAdamHyper hp = *hyper;
  if (hp.grad_scale == 0.f) hp.grad_scale = 1.f;
  hipLaunchKernelGGL(rs_sumsq_kernel, dim3(n_blocks), dim3(256), 0, st, (const char*)plan_dev, n_chunks, T,
                     hp.grad_scale, state_dev, partial);
  PS_LAUNCH_CHECK();
  hipLaunchKernelGGL(rs_finalize_kernel, dim3(1), dim3(1024), 0, st, hp, state_dev, partial, n_blocks, scal,
                     gnorm_out_dev);
  PS_LAUNCH_CHECK();
  hipLaunchKernelGGL(rs_update_kernel, dim3(n_blocks), dim3(256), 0, st, (const char*)plan_dev, n_chunks, T, hp, scal);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

// ---------------------------------------------------------------- lazy-EXACT dense Adam on a row-sparse table
// The reference's optimizer is dense (optimizers.py:186-187, 241-243): a row whose gradient is zero this step still moves —
// its first moment decays (m <- m + (0 - m)(1 - b1)), its second moment decays, and p <- p - step_size * m / (sqrt(v)/sqrt(bc2) + eps)
// — so the row-sparse rule above ("rows no step addressed stand still") is an extension, not the reference's arithmetic.
// This pass makes the row-sparse machinery reproduce the dense result EXACTLY: last[r] = number of optimizer steps row r has
// had applied; before a row is READ by a forward (and again, over the final touched list, before the step updates it) the
// steps it missed are replayed one by one with a zero gradient through the same adam_elem and the same per-step scalars
// (adam_step_scalars) as the dense kernel, so touched rows are bitwise what dense Adam would hold; everything else is stale
// until `all_rows` flushes the table (before evaluation / state_dict).  The replay stops early once a step changes no element
// of the row any more (m and v decay to values the multiplication maps to themselves, the update then underflows against p):
// later steps cannot change it either — except for an element so small that a larger later step size could reach it, which
// keeps the loop going (|p| < 1e-20 with m != 0).  Cost: the replay is sequential per row, up to ~1e5 steps for a row that
// was last touched long ago — an exactness mode for pinning the optimizer, not the fast path.
struct CatchTables { PsRowTable t[RS_MAX_TABLES]; int32_t* last[RS_MAX_TABLES]; int64_t n_rows[RS_MAX_TABLES]; int64_t wave0[RS_MAX_TABLES + 1]; int32_t n; };
template <int NK>
__device__ inline void catchup_row(const PsRowTable& tb, int32_t* last, int64_t row, const PsAdamHyper& hp, int64_t T, int advance,
                                   float gzero, int lane) {
  const int l0 = last[row];
  if ((int64_t)l0 >= T) {                                // nothing missed (or already advanced for the step in progress)
    if (advance && (int64_t)l0 == T && lane == 0) last[row] = (int32_t)(T + 1);
    return;
  }
  const int d = tb.d;
  const int64_t off = row * (int64_t)d;
  float pp[NK], mm[NK], vv[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int e = lane + 64 * k;
    pp[k] = e < d ? tb.p[off + e] : 1.f; mm[k] = e < d ? tb.m[off + e] : 0.f; vv[k] = e < d ? tb.v[off + e] : 0.f;
  }
  bool done = false;
  for (int64_t s0 = (int64_t)l0 + 1; s0 <= T && !done; s0 += 64) {
    float my_ss = 0.f, my_is = 0.f, my_lr;                 // lane i: the scalars of step s0 + i
    if (s0 + lane <= T) adam_step_scalars(hp, s0 + lane, &my_ss, &my_is, &my_lr);
    const int nstep = (int)((T - s0 + 1) < 64 ? (T - s0 + 1) : 64);
    for (int i = 0; i < nstep; ++i) {
      AdamScal a = {1.f, __shfl(my_ss, i, 64), __shfl(my_is, i, 64), hp.beta1, hp.beta2, hp.eps, hp.weight_decay, 0, 0.f, PS_OPT_ADAM};
      bool moving = false;
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const float p0 = pp[k], m0 = mm[k], v0 = vv[k];
        adam_elem(a, pp[k], gzero, mm[k], vv[k]);
        moving |= (pp[k] != p0) || (mm[k] != m0) || (vv[k] != v0) || (mm[k] != 0.f && fabsf(pp[k]) < 1e-20f);
      }
      if (!__any(moving)) { done = true; break; }
    }
  }
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int e = lane + 64 * k;
    if (e < d) { tb.p[off + e] = pp[k]; tb.m[off + e] = mm[k]; tb.v[off + e] = vv[k]; }
  }
  if (lane == 0) last[row] = (int32_t)(advance ? T + 1 : T);
}
__global__ __launch_bounds__(256) void rs_catchup_kernel(CatchTables C, const PsAdamHyper hp, const int64_t* state, int advance,
                                                         int all_rows, float gzero) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  int k = 0;
  while (k < C.n - 1 && wave >= C.wave0[k + 1]) ++k;
  const PsRowTable& tb = C.t[k];
  const int64_t u = wave - C.wave0[k];
  int64_t row;
  if (all_rows) { if (u >= C.n_rows[k]) return; row = u; }
  else { if (u >= (int64_t)*tb.count) return; row = tb.rows[u]; }
  const int64_t T = state[0];
  if (tb.d <= 128) catchup_row<2>(tb, C.last[k], row, hp, T, advance, gzero, lane);
  else if (tb.d <= 256) catchup_row<4>(tb, C.last[k], row, hp, T, advance, gzero, lane);
  else catchup_row<8>(tb, C.last[k], row, hp, T, advance, gzero, lane);
}
extern "C" int ps_rowsparse_catchup(const PsRowTable* tables_host, int32_t n_tables, int32_t* const* last_dev,
                                    const int64_t* n_rows_host, const PsAdamHyper* hyper, const int64_t* state_dev,
                                    int32_t advance, int32_t all_rows, ps_stream_t stream) {
  PS_REQUIRE(tables_host && last_dev && n_rows_host && hyper && state_dev && n_tables >= 1 && n_tables <= RS_MAX_TABLES,
             "rowsparse_catchup: bad argument");
  CatchTables C = CatchTables();
  C.n = n_tables;
  int64_t w = 0;
  for (int i = 0; i < n_tables; ++i) {
    const PsRowTable& t = tables_host[i];
    PS_REQUIRE(t.p && t.m && t.v && last_dev[i] && t.d > 0 && t.d <= 512 && n_rows_host[i] > 0 && n_rows_host[i] < ((int64_t)1 << 31) &&
               (all_rows || (t.rows && t.count && t.cap >= 0)), "rowsparse_catchup: table %d bad field", i);
    C.t[i] = t; C.last[i] = last_dev[i]; C.n_rows[i] = n_rows_host[i];
    C.wave0[i] = w;
    w += all_rows ? n_rows_host[i] : t.cap;
  }
  for (int i = n_tables; i <= RS_MAX_TABLES; ++i) C.wave0[i] = w;
  if (w == 0) return PS_OK;
  PS_REQUIRE((w + 3) / 4 < ((int64_t)1 << 31), "rowsparse_catchup: too many rows");
  hipLaunchKernelGGL(rs_catchup_kernel, dim3((unsigned)((w + 3) / 4)), dim3(256), 0, (hipStream_t)stream, C, *hyper, state_dev,
                     advance ? 1 : 0, all_rows ? 1 : 0, 0.f);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

// zero_grad() of rows a backward touched but no optimizer step consumed.
__global__ __launch_bounds__(256) void rows_zero_kernel(float* tab, const int64_t* rows, const int32_t* count, int d) {
  const int64_t u = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (u >= (int64_t)*count) return;
  float4* t4 = (float4*)(tab + rows[u] * (int64_t)d);
  for (int j = threadIdx.x & 31; j < d / 4; j += 32) t4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
}

extern "C" int ps_zero_rows(float* table_dev, int32_t d, const int64_t* rows_dev, const int32_t* count_dev,
                            int64_t cap, ps_stream_t stream) {
  PS_REQUIRE(table_dev && rows_dev && count_dev && d > 0 && d % 4 == 0 && cap >= 0, "zero_rows: bad argument");
  if (cap == 0) return PS_OK;
  hipLaunchKernelGGL(rows_zero_kernel, dim3((unsigned)((cap + 7) / 8)), dim3(256), 0, (hipStream_t)stream, table_dev,
                     rows_dev, count_dev, d);
  PS_LAUNCH_CHECK();
  return PS_OK;
}


// ================================================================== row-sharded tables (prodsearch_amd/sharded.py)
// Row i of a sharded table lives on rank i % world at local row i / world.  Per step a rank needs the rows its batch
// addresses: its sorted unique list (ps_coalesce_rows) is cut into one FIXED-capacity request per owner (no size ever
// crosses to the host; capacity = share of the step's index count with headroom, overflow sets the status word), the
// requests and the rows travel by equal-split all-to-alls, and the batch's indices are remapped to the slot each row
// arrives in, so the receive buffer itself is the table the step's kernels read.
//   ps_shard_bucket: one workgroup per owner walks the list; request o = local rows (id / world) of the ids with
//                    id % world == o, ascending, then -1; slot_of[u] = o * capp + position (world * capp = the pad slot)
//   ps_shard_remap : index tensor -> slots (binary search of the id in the sorted list)
__global__ __launch_bounds__(256) void shard_bucket_kernel(const int64_t* rows, const int32_t* count, int world, int64_t capp,
                                                           int64_t* send_ids, int32_t* slot_of, int32_t* bad) {
  __shared__ int wsum[4];
  const int o = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = *count;
  int base = 0;
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + tid;
    const int64_t r = i < n ? rows[i] : -1;
    const bool mine = i < n && (int)(r % world) == o;
    const unsigned long long m = __ballot(mine);
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    int before = 0;
    for (int k = 0; k < wv; ++k) before += wsum[k];
    const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (mine) {
      const int pos = base + before + __popcll(m & ((1ull << lane) - 1ull));
      if (pos < capp) { send_ids[(int64_t)o * capp + pos] = r / world; slot_of[i] = (int32_t)((int64_t)o * capp + pos); }
      else { *bad = 2; slot_of[i] = (int32_t)((int64_t)world * capp); }
    }
    base += total;
    __syncthreads();
  }
  for (int64_t p = (base < capp ? base : capp) + tid; p < capp; p += 256) send_ids[(int64_t)o * capp + p] = -1;
}
__global__ __launch_bounds__(256) void shard_remap_kernel(const int64_t* idx, int64_t n, int64_t pad_in, const int64_t* rows,
                                                          const int32_t* count, const int32_t* slot_of, int64_t pad_out,
                                                          int64_t* out, int32_t* bad) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t v = idx[i];
  if (v == pad_in) { out[i] = pad_out; return; }
  int lo = 0, hi = *count;                    // first position with rows[pos] >= v
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (rows[mid] < v) lo = mid + 1; else hi = mid; }
  if (lo < *count && rows[lo] == v) out[i] = slot_of[lo];
  else { out[i] = pad_out; *bad = 1; }
}
extern "C" int ps_shard_bucket(const int64_t* rows_dev, const int32_t* count_dev, int32_t world, int64_t capp,
                               int64_t* send_ids_dev, int32_t* slot_of_dev, int32_t* bad_dev, ps_stream_t stream) {
  PS_REQUIRE(rows_dev && count_dev && send_ids_dev && slot_of_dev && bad_dev && world > 0 && world <= 1024 && capp > 0 &&
             (int64_t)world * capp < ((int64_t)1 << 31), "shard_bucket: bad argument");
  hipLaunchKernelGGL(shard_bucket_kernel, dim3(world), dim3(256), 0, (hipStream_t)stream, rows_dev, count_dev, world, capp,
                     send_ids_dev, slot_of_dev, bad_dev);
  PS_LAUNCH_CHECK();
  return PS_OK;
}
extern "C" int ps_shard_remap(const int64_t* idx_dev, int64_t n, int64_t pad_in, const int64_t* rows_dev, const int32_t* count_dev,
                              const int32_t* slot_of_dev, int64_t pad_out, int64_t* out_dev, int32_t* bad_dev, ps_stream_t stream) {
  PS_REQUIRE(idx_dev && rows_dev && count_dev && slot_of_dev && out_dev && bad_dev && n >= 0, "shard_remap: bad argument");
  if (n == 0) return PS_OK;
  hipLaunchKernelGGL(shard_remap_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx_dev, n,
                     pad_in, rows_dev, count_dev, slot_of_dev, pad_out, out_dev, bad_dev);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

// ---- the row-sparse clip + Adam cut in two for sharded tables: the first n_shared tables (and the dense plan) are
// REPLICATED on every rank — their sum of squares is the same everywhere and counts once — while the remaining tables are
// this rank's SHARDS, whose sums of squares add up over the ranks (one scalar all-reduce between the two calls).
//   ps_rowsparse_sumsq     : sums[0] = dense plan + shared tables, sums[1] = owned shards (fixed-order reductions); step += 1
//   ps_rowsparse_update_ext: clip coefficient from sums[0] + sums[1] (the caller has all-reduced sums[1]), then Adam on the
//                            dense plan and the touched rows of every table; touched gradient rows come back zeroed
__global__ __launch_bounds__(1024) void rs_two_sums_kernel(const float* partial, int n_common, int n_all, float* sums) {
  __shared__ float sh[2][16];
  float s0 = strided_sum_f32<8>(partial, n_common, threadIdx.x, 1024);
  float s1 = strided_sum_f32<8>(partial + n_common, n_all - n_common, threadIdx.x, 1024);
  s0 = wave_sum(s0); s1 = wave_sum(s1);
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s0; sh[1][threadIdx.x >> 6] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t0 = 0.f, t1 = 0.f;
    for (int i = 0; i < 16; ++i) { t0 += sh[0][i]; t1 += sh[1][i]; }
    sums[0] = t0; sums[1] = t1;
  }
}
__global__ void rs_scalars_ext_kernel(const PsAdamHyper hp, const int64_t* state, const float* sums, float* scal, float* gnorm_out) {
  float norm;
  adam_scalars(hp, sums[0] + sums[1], state[0], scal, &norm);
  if (gnorm_out) { gnorm_out[0] = norm; gnorm_out[1] = scal[3]; }
}
extern "C" int ps_rowsparse_sumsq(const void* plan_dev, int32_t n_chunks, const PsRowTable* tables_host, int32_t n_tables,
                                  int32_t n_shared, const PsAdamHyper* hyper, int64_t* state_dev, float* sums_dev,
                                  ps_stream_t stream) {
  PS_REQUIRE(hyper && hyper->method == 0, "ps_rowsparse_sumsq: the row-sparse optimizer is Adam only (method %d)", hyper ? hyper->method : -1);
  PS_REQUIRE(hyper && state_dev && sums_dev && n_chunks >= 0 && (n_chunks == 0 || plan_dev) && n_shared >= 0 && n_shared <= n_tables,
             "rowsparse_sumsq: bad argument");
  RowTables T;
  int rc = rs_pack(tables_host, n_tables, &T);
  if (rc != PS_OK) return rc;
  const int n_blocks = n_chunks + T.blk0[RS_MAX_TABLES];
  PS_REQUIRE(n_blocks > 0, "rowsparse_sumsq: nothing to update");
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)(state_dev + 2) + 4;
  const float gs = hyper->grad_scale == 0.f ? 1.f : hyper->grad_scale;
  hipLaunchKernelGGL(rs_sumsq_kernel, dim3(n_blocks), dim3(256), 0, st, (const char*)plan_dev, n_chunks, T, gs, state_dev, partial);
  PS_LAUNCH_CHECK();
  hipLaunchKernelGGL(rs_two_sums_kernel, dim3(1), dim3(1024), 0, st, partial, n_chunks + T.blk0[n_shared], n_blocks, sums_dev);
  PS_LAUNCH_CHECK();
  return PS_OK;
}
extern "C" int ps_rowsparse_update_ext(const void* plan_dev, int32_t n_chunks, const PsRowTable* tables_host, int32_t n_tables,
                                       const PsAdamHyper* hyper, int64_t* state_dev, const float* sums_dev,
                                       float* gnorm_out_dev, ps_stream_t stream) {
  PS_REQUIRE(hyper && hyper->method == 0, "ps_rowsparse_update_ext: the row-sparse optimizer is Adam only (method %d)", hyper ? hyper->method : -1);
  PS_REQUIRE(hyper && state_dev && sums_dev && n_chunks >= 0 && (n_chunks == 0 || plan_dev), "rowsparse_update_ext: bad argument");
  RowTables T;
  int rc = rs_pack(tables_host, n_tables, &T);
  if (rc != PS_OK) return rc;
  const int n_blocks = n_chunks + T.blk0[RS_MAX_TABLES];
  PS_REQUIRE(n_blocks > 0, "rowsparse_update_ext: nothing to update");
  hipStream_t st = (hipStream_t)stream;
  float* scal = (float*)(state_dev + 2);
  PsAdamHyper hp = *hyper;
  if (hp.grad_scale == 0.f) hp.grad_scale = 1.f;
  hipLaunchKernelGGL(rs_scalars_ext_kernel, dim3(1), dim3(1), 0, st, hp, state_dev, sums_dev, scal, gnorm_out_dev);
  PS_LAUNCH_CHECK();
  hipLaunchKernelGGL(rs_update_kernel, dim3(n_blocks), dim3(256), 0, st, (const char*)plan_dev, n_chunks, T, hp, scal);
  PS_LAUNCH_CHECK();
  return PS_OK;
}
