// collate.cpp — host-side batch builder behind include/prodsearch_data.h.
// Plain C++ (no GPU): CSR walks + a CPython-compatible Mersenne Twister, so that seeded batches are
// bit-identical to data/item_pv_dataloader.py:121-143 driven by Python's `random`.
#include "../../include/prodsearch_data.h"
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <vector>

static thread_local char g_err[512] = "";
static int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return 1;
}
extern "C" const char* ps_data_last_error(void) { return g_err; }

// ----------------------------------------------------------------------------- MT19937 as CPython drives it
struct PsRng {
  uint32_t mt[624];
  int idx;
  void init_genrand(uint32_t s) {
    mt[0] = s;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
  }
  void init_by_array(const uint32_t* key, int len) {          // _randommodule.c init_by_array
    init_genrand(19650218u);
    int i = 1, j = 0;
    for (int k = (624 > len ? 624 : len); k; --k) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
      if (++i >= 624) { mt[0] = mt[623]; i = 1; }
      if (++j >= len) j = 0;
    }
    for (int k = 623; k; --k) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
      if (++i >= 624) { mt[0] = mt[623]; i = 1; }
    }
    mt[0] = 0x80000000u;
  }
  uint32_t next() {                                             // genrand_uint32
    if (idx >= 624) {
      for (int k = 0; k < 624; ++k) {
        uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
  }
  uint32_t randbelow(uint32_t n) {                              // Random._randbelow_with_getrandbits
    int k = 0;
    for (uint32_t t = n; t; t >>= 1) ++k;                       // n.bit_length()
    uint32_t r = next() >> (32 - k);                            // getrandbits(k), 0 < k <= 32
    while (r >= n) r = next() >> (32 - k);
    return r;
  }
};

extern "C" void ps_rng_seed(void* rng, uint64_t seed) {         // random.seed(int): key = 32-bit digits of |seed|
  uint32_t key[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
  ((PsRng*)rng)->init_by_array(key, key[1] ? 2 : 1);
}
extern "C" void* ps_rng_create(uint64_t seed) {
  PsRng* r = new PsRng;
  ps_rng_seed(r, seed);
  return r;
}
extern "C" void ps_rng_destroy(void* rng) { delete (PsRng*)rng; }
extern "C" uint32_t ps_rng_randbelow(void* rng, uint32_t n) { return n ? ((PsRng*)rng)->randbelow(n) : 0; }
extern "C" double ps_rng_random(void* rng) {                    // random_random: 53-bit
  PsRng* r = (PsRng*)rng;
  uint32_t a = r->next() >> 5, b = r->next() >> 6;
  return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

// random.sample(population of size n, k) -> the selected POSITIONS, in CPython's selection order
static void py_sample_positions(PsRng* rng, int n, int k, std::vector<int>& out) {
  static thread_local std::vector<int> pool;
  static thread_local std::vector<uint8_t> sel;
  out.resize(k);
  int setsize = 21;
  if (k > 5) setsize += (int)pow(4.0, ceil(log((double)k * 3.0) / log(4.0)));
  if (n <= setsize) {
    pool.resize(n);
    for (int i = 0; i < n; ++i) pool[i] = i;
    for (int i = 0; i < k; ++i) {
      int j = (int)rng->randbelow((uint32_t)(n - i));
      out[i] = pool[j];
      pool[j] = pool[n - i - 1];
    }
  } else {
    sel.assign(n, 0);
    for (int i = 0; i < k; ++i) {
      int j = (int)rng->randbelow((uint32_t)n);
      while (sel[j]) j = (int)rng->randbelow((uint32_t)n);
      sel[j] = 1;
      out[i] = j;
    }
  }
}

static int check_corpus(const PsCorpusView* c, const PsCollateArgs* a) {
  if (!c || !a) return fail("collate: null corpus/args");
  if (!c->review_u_p || !c->u_seq_ptr || !c->u_seq || !c->train_review || !c->query_words)
    return fail("collate: corpus array missing");
  if (a->uprev_review_limit < 1) return fail("collate: uprev_review_limit must be >= 1");
  if (a->do_seq && !c->review_loc) return fail("collate: do_seq needs review_loc");
  if (c->Q < 1) return fail("collate: Q < 1");
  return 0;
}

// get_user_review_idxs (item_pv_dataloader.py:85-102) followed by review -> product (:136, :44): writes the
// history ITEMS of (user, review) into dst[0..limit) and returns their count, or -1 on a bad id.
static int history_items(const PsCorpusView* c, const PsCollateArgs* a, PsRng* rng, int64_t user, int64_t review,
                         bool fix, int64_t* dst, std::vector<int64_t>& tmp, std::vector<int>& pos) {
  const int limit = a->uprev_review_limit;
  const int64_t beg = c->u_seq_ptr[user], end = c->u_seq_ptr[user + 1];
  tmp.clear();
  if (a->do_seq) {
    int64_t loc = c->review_loc[review];
    if (loc < 0 || loc > end - beg) return -1;
    int64_t from = loc > limit ? loc - limit : 0;                // [:loc][-limit:]
    for (int64_t i = from; i < loc; ++i) tmp.push_back(c->u_seq[beg + i]);
  } else {
    for (int64_t i = beg; i < end; ++i) {
      const int64_t r = c->u_seq[i];
      if (r < 0 || r >= c->n_reviews) return -1;
      if (c->train_review[r] && r != review) tmp.push_back(r);
    }
    const int n = (int)tmp.size();
    if (n > limit) {
      if (fix) {
        tmp.erase(tmp.begin(), tmp.end() - limit);               // [-limit:]
      } else {
        py_sample_positions(rng, n, limit, pos);                 // random.sample -> set -> order-preserving filter
        static thread_local std::vector<uint8_t> keep;
        keep.assign(n, 0);
        for (int p : pos) keep[p] = 1;
        int w = 0;
        for (int i = 0; i < n; ++i) if (keep[i]) tmp[w++] = tmp[i];
        tmp.resize(w);
      }
    }
  }
  int n = (int)tmp.size();
  for (int i = 0; i < n; ++i) {
    const int64_t r = tmp[i];
    if (r < 0 || r >= c->n_reviews) return -1;
    dst[i] = c->review_u_p[2 * r + 1];
  }
  for (int i = n; i < limit; ++i) dst[i] = a->prod_pad;
  return n;
}

extern "C" int ps_collate_train(const PsCorpusView* c, const PsCollateArgs* a, void* rng_,
                                const int64_t* sample_words, const int64_t* sample_review, int64_t n_samples, int32_t W,
                                const int64_t* batch_ids, int32_t B, int64_t* out_qw, int64_t* out_target,
                                int64_t* out_u_items, int64_t* out_pos_words, int64_t* out_query_idx,
                                int64_t* out_user_idx, int32_t* out_hist_len, int32_t* out_lmax) {
  if (check_corpus(c, a)) return 1;
  PsRng* rng = (PsRng*)rng_;
  if (!rng || !c->pq_ptr || !c->pq_idx) return fail("collate_train: rng / product-query CSR missing");
  if (!sample_words || !sample_review || !batch_ids || B < 1 || W < 1) return fail("collate_train: bad sample arguments");
  if (!out_qw || !out_target || !out_u_items || !out_pos_words || !out_hist_len || !out_lmax)
    return fail("collate_train: null output");
  std::vector<int64_t> tmp;
  std::vector<int> pos;
  const int limit = a->uprev_review_limit, Q = c->Q;
  int lmax = 0;
  for (int b = 0; b < B; ++b) {                                   // sample order = RNG order (:126-139)
    const int64_t s = batch_ids[b];
    if (s < 0 || s >= n_samples) return fail("collate_train: sample id %lld out of range", (long long)s);
    const int64_t review = sample_review[s];
    if (review < 0 || review >= c->n_reviews) return fail("collate_train: review id %lld out of range", (long long)review);
    const int64_t user = c->review_u_p[2 * review], prod = c->review_u_p[2 * review + 1];
    if (user < 0 || user >= c->n_users || prod < 0 || prod >= c->n_products)
      return fail("collate_train: review %lld maps to user %lld / product %lld", (long long)review, (long long)user, (long long)prod);
    const int64_t nq = c->pq_ptr[prod + 1] - c->pq_ptr[prod];
    if (nq < 1) return fail("collate_train: product %lld has no query", (long long)prod);
    const int64_t q = c->pq_idx[c->pq_ptr[prod] + rng->randbelow((uint32_t)nq)];      // random.choice (:130)
    if (q < 0 || q >= c->n_queries) return fail("collate_train: query id %lld out of range", (long long)q);
    memcpy(out_qw + (size_t)b * Q, c->query_words + (size_t)q * Q, sizeof(int64_t) * Q);
    memcpy(out_pos_words + (size_t)b * W, sample_words + (size_t)s * W, sizeof(int64_t) * W);
    out_target[b] = prod;
    if (out_query_idx) out_query_idx[b] = q;
    if (out_user_idx) out_user_idx[b] = user;
    int n = history_items(c, a, rng, user, review, a->fix != 0, out_u_items + (size_t)b * limit, tmp, pos);
    if (n < 0) return fail("collate_train: corrupt history of user %lld", (long long)user);
    out_hist_len[b] = n;
    if (n > lmax) lmax = n;
  }
  *out_lmax = lmax;
  return 0;
}

extern "C" int ps_collate_test(const PsCorpusView* c, const PsCollateArgs* a, const int64_t* quad, int32_t B,
                               const int64_t* candi_ptr, const int64_t* candi_items, int32_t candi_width,
                               int64_t* out_qw, int64_t* out_target, int64_t* out_u_items, int64_t* out_candi,
                               int32_t* out_hist_len, int32_t* out_lmax) {
  if (check_corpus(c, a)) return 1;
  if (!quad || B < 1 || !candi_ptr || candi_width < 0 || (candi_width > 0 && !candi_items))
    return fail("collate_test: bad arguments");
  if (!out_qw || !out_target || !out_u_items || (candi_width > 0 && !out_candi) || !out_hist_len || !out_lmax)
    return fail("collate_test: null output");
  std::vector<int64_t> tmp;
  std::vector<int> pos;
  const int limit = a->uprev_review_limit, Q = c->Q;
  int lmax = 0;
  for (int b = 0; b < B; ++b) {
    const int64_t q = quad[4 * b], user = quad[4 * b + 1], prod = quad[4 * b + 2], review = quad[4 * b + 3];
    if (q < 0 || q >= c->n_queries || user < 0 || user >= c->n_users || review < 0 || review >= c->n_reviews)
      return fail("collate_test: entry %d out of range", b);
    memcpy(out_qw + (size_t)b * Q, c->query_words + (size_t)q * Q, sizeof(int64_t) * Q);
    out_target[b] = prod;
    int n = history_items(c, a, nullptr, user, review, true, out_u_items + (size_t)b * limit, tmp, pos);   // fix=True (:42)
    if (n < 0) return fail("collate_test: corrupt history of user %lld", (long long)user);
    out_hist_len[b] = n;
    if (n > lmax) lmax = n;
    const int64_t cb = candi_ptr[b], ce = candi_ptr[b + 1];
    if (ce - cb > candi_width || ce < cb) return fail("collate_test: candidate list %d longer than width", b);
    if (candi_width == 0) continue;                                                     // full-catalogue entries
    int64_t* row = out_candi + (size_t)b * candi_width;
    memcpy(row, candi_items + cb, sizeof(int64_t) * (size_t)(ce - cb));
    for (int64_t i = ce - cb; i < candi_width; ++i) row[i] = a->prod_pad;               // util.pad (:46)
  }
  *out_lmax = lmax;
  return 0;
}

// ----------------------------------------------------------------------------- epoch sample collection
extern "C" void ps_rng_shuffle(void* rng_, int64_t* x, int64_t n) {          // Random.shuffle, CPython 3.10
  PsRng* rng = (PsRng*)rng_;
  for (int64_t i = n - 1; i >= 1; --i) {
    const int64_t j = (int64_t)rng->randbelow((uint32_t)(i + 1));
    const int64_t t = x[i]; x[i] = x[j]; x[j] = t;
  }
}

extern "C" int ps_collect_train_samples(const int64_t* rw_ptr, int64_t* rw_words, int64_t n_reviews,
                                        const int64_t* train_reviews, int64_t n_train, const double* rnd, int64_t n_rand,
                                        const double* sub_rate, int64_t vocab_size, int32_t W, int64_t word_pad, void* rng,
                                        int64_t* out_words, int64_t* out_review, int64_t cap, int64_t* out_n) {
  if (!rw_ptr || !rw_words || !train_reviews || !rnd || !sub_rate || !rng || !out_words || !out_review || !out_n || W < 1)
    return fail("collect_train_samples: bad argument");
  int64_t entry = 0, n_out = 0, last_review = -1;
  int fill = 0;
  for (int64_t t = 0; t < n_train; ++t) {
    const int64_t r = train_reviews[t];
    if (r < 0 || r >= n_reviews) return fail("collect_train_samples: review id %lld out of range", (long long)r);
    last_review = r;
    int64_t* w = rw_words + rw_ptr[r];
    const int64_t len = rw_ptr[r + 1] - rw_ptr[r];
    ps_rng_shuffle(rng, w, len);                                              // :81
    for (int64_t i = 0; i < len; ++i) {
      const int64_t word = w[i];
      if (word < 0 || word >= vocab_size) return fail("collect_train_samples: word id %lld out of range", (long long)word);
      if (entry >= n_rand) return fail("collect_train_samples: rand stream exhausted");
      if (rnd[entry] > sub_rate[word]) continue;                             // entry NOT advanced (:84-85)
      if (n_out >= cap) return fail("collect_train_samples: output capacity %lld too small", (long long)cap);
      out_words[n_out * W + fill] = word;
      if (++fill == W) { out_review[n_out++] = r; fill = 0; }
      ++entry;
    }
  }
  if (fill > 0) {                                                             // :91-92
    if (n_out >= cap) return fail("collect_train_samples: output capacity %lld too small", (long long)cap);
    for (int i = fill; i < W; ++i) out_words[n_out * W + i] = word_pad;
    out_review[n_out++] = last_review;
  }
  *out_n = n_out;
  return 0;
}

// ============================================================================= review-transformer batches
// numpy's legacy draws on the same MT19937 (numpy/random/src/distributions/distributions.c: random_interval;
// src/mt19937/mt19937.h: mt19937_next_double) — the caller loads np.random's state with ps_rng_set_state.
static uint64_t np_interval(PsRng* g, uint64_t max) {
  if (max == 0) return 0;
  uint64_t mask = max, value;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
  if (max <= 0xffffffffull) {
    while ((value = ((uint64_t)g->next() & mask)) > max) {}
  } else {
    while ((value = ((((uint64_t)g->next() << 32) | g->next()) & mask)) > max) {}
  }
  return value;
}
static inline double np_double(PsRng* g) {
  const uint32_t a = g->next() >> 5, b = g->next() >> 6;
  return (a * 67108864.0 + b) / 9007199254740992.0;
}
extern "C" uint64_t ps_rng_np_interval(void* rng, uint64_t max) { return np_interval((PsRng*)rng, max); }
extern "C" void ps_rng_get_state(void* rng, uint32_t key[624], int32_t* pos) {
  PsRng* g = (PsRng*)rng;
  memcpy(key, g->mt, sizeof(g->mt));
  *pos = g->idx;
}
extern "C" void ps_rng_set_state(void* rng, const uint32_t key[624], int32_t pos) {
  PsRng* g = (PsRng*)rng;
  memcpy(g->mt, key, sizeof(g->mt));
  g->idx = pos < 0 ? 0 : (pos > 624 ? 624 : pos);
}

static int check_rtm(const PsRtmCorpusView* c, const PsRtmCollateArgs* a) {
  if (!c || !a) return fail("rtm collate: null corpus/args");
  if (!c->review_u_p || !c->u_seq_ptr || !c->u_seq || !c->i_seq_ptr || !c->i_seq || !c->query_words)
    return fail("rtm collate: corpus array missing");
  if (a->do_seq ? !c->loc_time : (!c->ut_seq_ptr || !c->ut_seq || !c->it_seq_ptr || !c->it_seq))
    return fail("rtm collate: %s missing", a->do_seq ? "review_loc_time" : "train-review sequences");
  if (a->uprev_review_limit < 1 || a->iprev_review_limit < 1) return fail("rtm collate: review limits must be >= 1");
  if (c->Q < 1) return fail("rtm collate: Q < 1");
  return 0;
}

// get_user_review_idxs / get_item_review_idxs (prod_search_dataloader.py:135-194).  do_seq: the `limit` reviews before
// position `loc` of the full sequence; else the TRAIN reviews of the owner other than `review`, cut to `limit` by the
// last ones (fix) or by random.sample with the sequence order kept.
static void prev_reviews(const int64_t* fptr, const int64_t* fseq, const int64_t* tptr, const int64_t* tseq, int64_t id,
                         int64_t review, bool do_seq, int64_t loc, int limit, bool fix, PsRng* rng,
                         std::vector<int64_t>& out, std::vector<int>& pos) {
  out.clear();
  if (do_seq) {
    const int64_t beg = fptr[id], n = fptr[id + 1] - beg;
    if (loc > n) loc = n;                                         // a Python slice clamps
    if (loc <= 0) return;
    for (int64_t i = loc > limit ? loc - limit : 0; i < loc; ++i) out.push_back(fseq[beg + i]);      // [:loc][-limit:]
    return;
  }
  for (int64_t i = tptr[id]; i < tptr[id + 1]; ++i)
    if (tseq[i] != review) out.push_back(tseq[i]);
  const int n = (int)out.size();
  if (n <= limit) return;
  if (fix) { out.erase(out.begin(), out.end() - limit); return; }
  py_sample_positions(rng, n, limit, pos);
  static thread_local std::vector<uint8_t> keep;
  keep.assign(n, 0);
  for (int p : pos) keep[p] = 1;
  int w = 0;
  for (int i = 0; i < n; ++i) if (keep[i]) out[w++] = out[i];
  out.resize(w);
}

// ProdSearchDataset.bisect_right (prod_search_dataset.py:133-151) on the time column
static int64_t bisect_time(const PsRtmCorpusView* c, int64_t prod, int64_t ts) {
  const int64_t* arr = c->i_seq + c->i_seq_ptr[prod];
  int64_t lo = 0, hi = c->i_seq_ptr[prod + 1] - c->i_seq_ptr[prod];
  while (lo < hi) {
    const int64_t mid = (lo + hi) / 2;
    if (ts < c->loc_time[3 * arr[mid] + 2]) hi = mid; else lo = mid + 1;
  }
  return lo;
}

struct RtmSeq { size_t off; int nu, ni; int64_t user, item; };     // [user's reviews | item's reviews] in `flat`

// one padded row of the four per-position tensors (:212-219, :232-239)
static void write_seq(const PsRtmCorpusView* c, const PsRtmCollateArgs* a, const std::vector<int64_t>& flat,
                      const RtmSeq* s, int R, int64_t* ridx, int64_t* seg, int64_t* usr, int64_t* itm) {
  const int nu = s ? s->nu : 0, ni = s ? s->ni : 0;
  seg[0] = s ? 0 : 3; usr[0] = a->user_pad; itm[0] = a->prod_pad;
  for (int r = 0; r < R; ++r) {
    if (r < nu + ni) {
      const int64_t x = flat[s->off + r];
      ridx[r] = x;
      seg[r + 1] = r < nu ? 1 : 2;
      usr[r + 1] = r < nu ? s->user : c->review_u_p[2 * x];
      itm[r + 1] = r < nu ? c->review_u_p[2 * x + 1] : s->item;
    } else {
      ridx[r] = a->review_pad; seg[r + 1] = 3; usr[r + 1] = a->user_pad; itm[r + 1] = a->prod_pad;
    }
  }
}

static int check_reviews(const PsRtmCorpusView* c, const std::vector<int64_t>& v) {
  for (int64_t x : v) if (x < 0 || x >= c->n_reviews) return 1;
  return 0;
}

extern "C" int ps_rtm_collate_train(const PsRtmCorpusView* c, const PsRtmCollateArgs* a, void* rng_,
                                    const int64_t* rows, int32_t B, const int64_t* neg_products, int64_t n_lines,
                                    int64_t* out_qw, int64_t* out_kept, int64_t* out_pos_ridxs, int64_t* out_pos_seg,
                                    int64_t* out_pos_user, int64_t* out_pos_item, int64_t* out_neg_ridxs,
                                    int64_t* out_neg_seg, int64_t* out_neg_user, int64_t* out_neg_item, int32_t dims[4]) {
  if (check_rtm(c, a)) return 1;
  PsRng* rng = (PsRng*)rng_;
  if (!rng || !c->pq_ptr || !c->pq_idx) return fail("rtm collate_train: rng / product-query CSR missing");
  if (!rows || B < 1 || !neg_products || a->neg_per_pos < 1) return fail("rtm collate_train: bad arguments");
  if (!out_qw || !out_kept || !out_pos_ridxs || !out_pos_seg || !out_pos_user || !out_pos_item || !out_neg_ridxs ||
      !out_neg_seg || !out_neg_user || !out_neg_item || !dims)
    return fail("rtm collate_train: null output");
  static thread_local std::vector<int64_t> flat, up, ip, np_;
  static thread_local std::vector<RtmSeq> posq, negq;
  static thread_local std::vector<int> negcnt, pos;
  static thread_local std::vector<int64_t> qsel;
  flat.clear(); posq.clear(); negq.clear(); negcnt.clear(); qsel.clear();
  const int K = a->neg_per_pos, Q = c->Q;
  const bool seq = a->do_seq != 0;
  int Bk = 0, Rp = 0, Kk = 0, Rn = 0;
  for (int b = 0; b < B; ++b) {
    const int64_t line = rows[4 * b], user = rows[4 * b + 1], prod = rows[4 * b + 2], review = rows[4 * b + 3];
    if (line < 0 || line >= n_lines || user < 0 || user >= c->n_users || prod < 0 || prod >= c->n_products ||
        review < 0 || review >= c->n_reviews)
      return fail("rtm collate_train: row %d out of range", b);
    const int64_t nq = c->pq_ptr[prod + 1] - c->pq_ptr[prod];
    if (nq < 1) return fail("rtm collate_train: product %lld has no query", (long long)prod);
    const int64_t q = c->pq_idx[c->pq_ptr[prod] + rng->randbelow((uint32_t)nq)];             // random.choice (:201)
    if (q < 0 || q >= c->n_queries) return fail("rtm collate_train: query id %lld out of range", (long long)q);
    prev_reviews(c->u_seq_ptr, c->u_seq, c->ut_seq_ptr, c->ut_seq, user, review, seq,
                 seq ? c->loc_time[3 * review] : 0, a->uprev_review_limit, false, rng, up, pos);          // :203
    prev_reviews(c->i_seq_ptr, c->i_seq, c->it_seq_ptr, c->it_seq, prod, review, seq,
                 seq ? c->loc_time[3 * review + 1] : 0, a->iprev_review_limit, false, rng, ip, pos);      // :204
    if (ip.empty()) continue;                                                                 // :208-209
    if (check_reviews(c, up) || check_reviews(c, ip)) return fail("rtm collate_train: corrupt review sequence");
    const int64_t ts = seq ? c->loc_time[3 * review + 2] : 0;
    const size_t flat0 = flat.size(), neg0 = negq.size();
    RtmSeq ps = {flat.size(), (int)up.size(), (int)ip.size(), user, prod};
    flat.insert(flat.end(), up.begin(), up.end());
    flat.insert(flat.end(), ip.begin(), ip.end());
    int cnt = 0;
    for (int k = 0; k < K; ++k) {                                                             // :226-242
      const int64_t neg = neg_products[line * K + k];
      if (neg < 0 || neg >= c->n_products) return fail("rtm collate_train: negative product %lld out of range", (long long)neg);
      prev_reviews(c->i_seq_ptr, c->i_seq, c->it_seq_ptr, c->it_seq, neg, -1, seq, seq ? bisect_time(c, neg, ts) : 0,
                   a->iprev_review_limit, false, rng, np_, pos);
      if (np_.empty()) continue;
      if (check_reviews(c, np_)) return fail("rtm collate_train: corrupt review sequence");
      RtmSeq ns = {flat.size(), (int)up.size(), (int)np_.size(), user, neg};
      flat.insert(flat.end(), up.begin(), up.end());
      flat.insert(flat.end(), np_.begin(), np_.end());
      negq.push_back(ns);
      if (ns.nu + ns.ni > Rn) Rn = ns.nu + ns.ni;
      ++cnt;
    }
    if (cnt == 0) { flat.resize(flat0); negq.resize(neg0); continue; }                        // :243-245
    posq.push_back(ps);
    negcnt.push_back(cnt);
    qsel.push_back(q);
    out_kept[Bk++] = b;
    if (ps.nu + ps.ni > Rp) Rp = ps.nu + ps.ni;
    if (cnt > Kk) Kk = cnt;
  }
  dims[0] = Bk; dims[1] = Rp; dims[2] = Kk; dims[3] = Rn;
  size_t nq_ = 0;
  for (int b = 0; b < Bk; ++b) {
    memcpy(out_qw + (size_t)b * Q, c->query_words + (size_t)qsel[b] * Q, sizeof(int64_t) * Q);
    write_seq(c, a, flat, &posq[b], Rp, out_pos_ridxs + (size_t)b * Rp, out_pos_seg + (size_t)b * (Rp + 1),
              out_pos_user + (size_t)b * (Rp + 1), out_pos_item + (size_t)b * (Rp + 1));
    for (int k = 0; k < Kk; ++k) {
      const size_t row = (size_t)b * Kk + k;
      write_seq(c, a, flat, k < negcnt[b] ? &negq[nq_ + k] : nullptr, Rn, out_neg_ridxs + row * Rn,
                out_neg_seg + row * (Rn + 1), out_neg_user + row * (Rn + 1), out_neg_item + row * (Rn + 1));
    }
    nq_ += negcnt[b];
  }
  return 0;
}

extern "C" int ps_rtm_collate_test(const PsRtmCorpusView* c, const PsRtmCollateArgs* a, const int64_t* quad, int32_t B,
                                   const int64_t* candi_ptr, const int64_t* candi_items, int64_t* out_qw,
                                   int64_t* out_candi, int64_t* out_ridxs, int64_t* out_seg, int64_t* out_user,
                                   int64_t* out_item, int32_t dims[2]) {
  if (check_rtm(c, a)) return 1;
  if (!quad || B < 1 || !candi_ptr || !candi_items) return fail("rtm collate_test: bad arguments");
  if (!out_qw || !out_candi || !out_ridxs || !out_seg || !out_user || !out_item || !dims)
    return fail("rtm collate_test: null output");
  static thread_local std::vector<int64_t> flat, up, ip;
  static thread_local std::vector<RtmSeq> seqs;
  static thread_local std::vector<int> pos;
  flat.clear(); seqs.clear();
  const bool seq = a->do_seq != 0;
  const int Q = c->Q;
  int C = 0, Rc = 0;
  for (int b = 0; b < B; ++b) {
    const int64_t q = quad[4 * b], user = quad[4 * b + 1], review = quad[4 * b + 3];
    if (q < 0 || q >= c->n_queries || user < 0 || user >= c->n_users || review < 0 || review >= c->n_reviews)
      return fail("rtm collate_test: entry %d out of range", b);
    const int64_t n = candi_ptr[b + 1] - candi_ptr[b];
    if (n < 0) return fail("rtm collate_test: candidate CSR not monotone");
    if (n > C) C = (int)n;
    prev_reviews(c->u_seq_ptr, c->u_seq, c->ut_seq_ptr, c->ut_seq, user, review, seq,
                 seq ? c->loc_time[3 * review] : 0, a->uprev_review_limit, true, nullptr, up, pos);       // fix=True (:64)
    if (check_reviews(c, up)) return fail("rtm collate_test: corrupt review sequence");
    const int64_t ts = seq ? c->loc_time[3 * review + 2] : 0;
    for (int64_t i = candi_ptr[b]; i < candi_ptr[b + 1]; ++i) {                               // :73-92
      const int64_t cand = candi_items[i];
      if (cand < 0 || cand >= c->n_products) return fail("rtm collate_test: candidate %lld out of range", (long long)cand);
      prev_reviews(c->i_seq_ptr, c->i_seq, c->it_seq_ptr, c->it_seq, cand, -1, seq, seq ? bisect_time(c, cand, ts) : 0,
                   a->iprev_review_limit, true, nullptr, ip, pos);
      if (check_reviews(c, ip)) return fail("rtm collate_test: corrupt review sequence");
      RtmSeq s = {flat.size(), (int)up.size(), (int)ip.size(), user, cand};
      flat.insert(flat.end(), up.begin(), up.end());
      flat.insert(flat.end(), ip.begin(), ip.end());
      seqs.push_back(s);
      if (s.nu + s.ni > Rc) Rc = s.nu + s.ni;
    }
  }
  dims[0] = C; dims[1] = Rc;
  size_t si = 0;
  for (int b = 0; b < B; ++b) {
    memcpy(out_qw + (size_t)b * Q, c->query_words + (size_t)quad[4 * b] * Q, sizeof(int64_t) * Q);
    const int n = (int)(candi_ptr[b + 1] - candi_ptr[b]);
    for (int k = 0; k < C; ++k) {
      const size_t row = (size_t)b * C + k;
      out_candi[row] = k < n ? candi_items[candi_ptr[b] + k] : -1;                            // util.pad(.., -1) (:95)
      write_seq(c, a, flat, k < n ? &seqs[si + k] : nullptr, Rc, out_ridxs + row * Rc, out_seg + row * (Rc + 1),
                out_user + row * (Rc + 1), out_item + row * (Rc + 1));
    }
    si += n;
  }
  return 0;
}

extern "C" int ps_rtm_word_masks(void* np_rng, const int64_t* words, int64_t n, int64_t word_pad, const double* sub_rate,
                                 int64_t vocab_size, uint8_t* masks) {
  if (!words || !masks || n < 0) return fail("rtm word_masks: bad arguments");
  PsRng* g = (PsRng*)np_rng;
  if (sub_rate && !g) return fail("rtm word_masks: sub-sampling needs the numpy generator state");
  for (int64_t i = 0; i < n; ++i) {
    const int64_t w = words[i];
    if (sub_rate) {
      if (w < 0 || w >= vocab_size) return fail("rtm word_masks: word id %lld out of range", (long long)w);
      const double r = np_double(g);                              // np.random.random(shape), C order (:91)
      masks[i] = (w != word_pad) && (r < sub_rate[w]);
    } else {
      masks[i] = w != word_pad;
    }
  }
  return 0;
}

extern "C" int ps_rtm_pv_windows(void* np_rng, int64_t* words, uint8_t* masks, int32_t Bk, int32_t Rp, int32_t WL,
                                 int32_t W, int64_t word_pad, const double* sub_rate, int64_t vocab_size,
                                 int32_t shuffle_rows, int32_t permute, int64_t* slide_words, uint8_t* slide_masks,
                                 int64_t* batch_index) {
  PsRng* g = (PsRng*)np_rng;
  if (!words || !masks || !slide_words || !slide_masks || !batch_index || Bk < 1 || Rp < 0 || WL < 1 || W < 1)
    return fail("rtm pv_windows: bad arguments");
  if ((sub_rate || shuffle_rows || permute) && !g) return fail("rtm pv_windows: the numpy generator state is required");
  const size_t row = (size_t)Rp * WL;
  if (ps_rtm_word_masks(np_rng, words, (int64_t)Bk * row, word_pad, sub_rate, vocab_size, masks)) return 1;   // :289-291
  if (shuffle_rows && Rp > 1) {                                   // shuffle_words_in_reviews (:307-308): np.random.shuffle of
    std::vector<int64_t> buf(WL);                                 // a [Rp,WL] slice swaps whole review rows
    for (int b = 0; b < Bk; ++b) {
      int64_t* x = words + (size_t)b * row;
      for (int i = Rp - 1; i >= 1; --i) {
        const int j = (int)np_interval(g, (uint64_t)i);
        if (i == j) continue;
        memcpy(buf.data(), x + (size_t)j * WL, sizeof(int64_t) * WL);
        memcpy(x + (size_t)j * WL, x + (size_t)i * WL, sizeof(int64_t) * WL);
        memcpy(x + (size_t)i * WL, buf.data(), sizeof(int64_t) * WL);
      }
    }
  }
  const int seg = (WL + W - 1) / W;                               // slide_padded_matrices_for_pv (:121-131)
  const int64_t n = (int64_t)seg * Bk;
  std::vector<int64_t> perm((size_t)n);
  for (int64_t i = 0; i < n; ++i) perm[i] = i;
  if (permute)                                                    // np.random.permutation(n) = shuffle(arange(n)) (:325)
    for (int64_t i = n - 1; i >= 1; --i) {
      const int64_t j = (int64_t)np_interval(g, (uint64_t)i);
      const int64_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }
  for (int64_t i = 0; i < n; ++i) {
    const int64_t s = perm[i] / Bk, b = perm[i] % Bk;
    batch_index[i] = b;
    for (int r = 0; r < Rp; ++r) {
      const int64_t* wsrc = words + ((size_t)b * Rp + r) * WL;
      const uint8_t* msrc = masks + ((size_t)b * Rp + r) * WL;
      int64_t* wd = slide_words + ((size_t)i * Rp + r) * W;
      uint8_t* md = slide_masks + ((size_t)i * Rp + r) * W;
      for (int w = 0; w < W; ++w) {
        const int64_t col = s * W + w;
        wd[w] = col < WL ? wsrc[col] : word_pad;
        md[w] = col < WL ? msrc[col] : 0;
      }
    }
  }
  return 0;
}

// ----------------------------------------------------------------------------- epoch producer (native prefetch thread)
// `for batch in dataloader` (trainer.py:64-66) with the collate OFF the consumer's thread: one std::thread builds the epoch's
// train batches one after another — so the generator is consumed in exactly the sequential order and seeded runs stay the
// reference's — into a ring of caller-owned (pinned) slots, while the consumer ships and trains on earlier ones.  No Python in
// the producer: a Python producer thread shares the interpreter lock with the step's launch code and made the fed step SLOWER
// (0.338 against 0.320 ms; the GPU step is 0.236).
#include <condition_variable>
#include <mutex>
#include <thread>

struct PsEpoch {
  const PsCorpusView* c; PsCollateArgs a; void* rng;
  const int64_t* sample_words; const int64_t* sample_review; int64_t n_samples; int32_t W;
  const int64_t* order; int64_t n_ids; int32_t B; int64_t n_batches;
  std::vector<PsTrainSlot> slots;
  std::vector<int> state;                 // 0 free, 1 ready, 2 held by the consumer
  std::vector<int32_t> slot_B, slot_lmax;
  int64_t produced = 0, consumed = 0;     // batch counters: batch k lives in slot k % depth
  bool stop = false, failed = false;
  char err[512] = "";
  std::mutex mu;
  std::condition_variable cv;
  std::thread th;
};

static void epoch_work(PsEpoch* e) {
  const int depth = (int)e->slots.size();
  for (int64_t k = 0; k < e->n_batches; ++k) {
    const int s = (int)(k % depth);
    {
      std::unique_lock<std::mutex> lk(e->mu);
      e->cv.wait(lk, [&] { return e->stop || e->state[s] == 0; });
      if (e->stop) return;
    }
    const int64_t beg = k * e->B;
    const int32_t nb = (int32_t)((e->n_ids - beg) < e->B ? (e->n_ids - beg) : e->B);
    const PsTrainSlot& o = e->slots[s];
    int32_t lmax = 0;
    const int rc = ps_collate_train(e->c, &e->a, e->rng, e->sample_words, e->sample_review, e->n_samples, e->W, e->order + beg, nb,
                                    o.query_words, o.target, o.u_items, o.pos_words, o.query_idx, o.user_idx, o.hist_len, &lmax);
    std::lock_guard<std::mutex> lk(e->mu);
    if (rc) {
      e->failed = true;
      snprintf(e->err, sizeof(e->err), "%s", g_err);           // this thread's message -> the handle
      e->cv.notify_all();
      return;
    }
    e->slot_B[s] = nb; e->slot_lmax[s] = lmax;
    e->state[s] = 1;
    e->produced = k + 1;
    e->cv.notify_all();
  }
}

extern "C" void* ps_epoch_start(const PsCorpusView* c, const PsCollateArgs* a, void* rng, const int64_t* sample_words,
                                const int64_t* sample_review, int64_t n_samples, int32_t W, const int64_t* order, int64_t n_ids,
                                int32_t B, int32_t drop_last, const PsTrainSlot* slots, int32_t depth) {
  if (!c || !a || !rng || !sample_words || !sample_review || !order || !slots || B < 1 || W < 1 || n_ids < 0 || depth < 2 || depth > 64) {
    fail("epoch_start: bad argument (B %d, depth %d, n_ids %lld)", B, depth, (long long)n_ids);
    return nullptr;
  }
  for (int i = 0; i < depth; ++i)
    if (!slots[i].query_words || !slots[i].target || !slots[i].u_items || !slots[i].pos_words || !slots[i].hist_len) {
      fail("epoch_start: slot %d has a null buffer", i);
      return nullptr;
    }
  PsEpoch* e = new PsEpoch();
  e->c = c; e->a = *a; e->rng = rng;
  e->sample_words = sample_words; e->sample_review = sample_review; e->n_samples = n_samples; e->W = W;
  e->order = order; e->n_ids = n_ids; e->B = B;
  e->n_batches = drop_last ? n_ids / B : (n_ids + B - 1) / B;
  e->slots.assign(slots, slots + depth);
  e->state.assign(depth, 0); e->slot_B.assign(depth, 0); e->slot_lmax.assign(depth, 0);
  e->th = std::thread(epoch_work, e);
  return e;
}

extern "C" int ps_epoch_next(void* h, int32_t* out_B, int32_t* out_lmax) {
  PsEpoch* e = (PsEpoch*)h;
  if (!e || !out_B || !out_lmax) { fail("epoch_next: null argument"); return -2; }
  std::unique_lock<std::mutex> lk(e->mu);
  if (e->consumed >= e->n_batches) return -1;
  const int depth = (int)e->slots.size(), s = (int)(e->consumed % depth);
  e->cv.wait(lk, [&] { return e->failed || e->state[s] == 1; });
  if (e->state[s] != 1) {                                       // the producer stopped on an error before this batch
    snprintf(g_err, sizeof(g_err), "%s", e->err);
    return -2;
  }
  e->state[s] = 2;
  e->consumed += 1;
  *out_B = e->slot_B[s]; *out_lmax = e->slot_lmax[s];
  return s;
}

extern "C" int ps_epoch_release(void* h, int32_t slot) {
  PsEpoch* e = (PsEpoch*)h;
  if (!e || slot < 0 || slot >= (int)e->slots.size()) return fail("epoch_release: bad slot %d", slot);
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->state[slot] != 2) return fail("epoch_release: slot %d is not held by the consumer", slot);
  e->state[slot] = 0;
  e->cv.notify_all();
  return 0;
}

extern "C" void ps_epoch_stop(void* h) {
  PsEpoch* e = (PsEpoch*)h;
  if (!e) return;
  {
    std::lock_guard<std::mutex> lk(e->mu);
    e->stop = true;
    e->cv.notify_all();
  }
  if (e->th.joinable()) e->th.join();
  delete e;
}
