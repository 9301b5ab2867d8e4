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
