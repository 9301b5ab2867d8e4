/* prodsearch_data.h — C ABI of the host-side batch builder.
 *
 * Replaces the Python collate of the TEM loaders, which becomes the end-to-end bottleneck once the
 * training step takes ~0.6 ms on an MI355X:
 *   ItemPVDataloader.get_train_batch      data/item_pv_dataloader.py:121-143
 *   ItemPVDataloader.get_test_batch       data/item_pv_dataloader.py:32-50
 *   ItemPVDataloader.get_user_review_idxs data/item_pv_dataloader.py:85-102
 *   others.util.pad                       others/util.py:36-40
 * The corpus is handed over once as flat arrays (CSR for the ragged members); every call fills caller-owned
 * (pinned) host buffers with the int64 tensors of data/batch_data.py:ItemPVBatch, ready for one async H2D copy.
 *
 * Randomness: the reference draws the query of a sample with random.choice and, when fix_train_review is off,
 * the history subset with random.sample — CPython's `random` module (third-party to the reference, absent from
 * /root/reference; pinned here to CPython 3.10: Lib/random.py choice/_randbelow_with_getrandbits/sample,
 * Modules/_randommodule.c MT19937 init_by_array/genrand_uint32/getrandbits).  PsRng restates exactly that
 * generator, so with the same seed the batches are bit-identical to the reference's with num_workers=0.
 * All pointers are HOST pointers; nothing here touches the GPU. */
#ifndef PRODSEARCH_DATA_H
#define PRODSEARCH_DATA_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct PsCorpusView {
  int64_t n_reviews, n_users, n_products, n_queries;
  const int64_t* review_u_p;    /* [n_reviews,2] (user, product): global_data.review_u_p                    */
  const int64_t* u_seq_ptr;     /* [n_users+1] CSR offsets of global_data.u_r_seq                           */
  const int64_t* u_seq;         /* review ids of every user in time order                                   */
  const uint8_t* train_review;  /* [n_reviews] 1 iff the review is in prod_data.u_reviews[its user]         */
  const int64_t* review_loc;    /* [n_reviews] global_data.review_loc_time[r][0] (position in the user seq) */
  const int64_t* pq_ptr;        /* [n_products+1] CSR offsets of prod_data.product_query_idx                */
  const int64_t* pq_idx;        /* query ids per product                                                    */
  const int64_t* query_words;   /* [n_queries,Q] global_data.query_words (already padded with V-1)          */
  int32_t Q;
  int32_t pad_;
} PsCorpusView;

typedef struct PsCollateArgs {
  int32_t uprev_review_limit;   /* --uprev_review_limit (>= 1)                                              */
  int32_t do_seq;               /* --do_seq_review_train / (--do_seq_review_test and not train_review_only) */
  int32_t fix;                  /* --fix_train_review; the test collate always passes 1                     */
  int32_t pad_;
  int64_t prod_pad;             /* product_size (item_pv_dataset.py:23)                                     */
} PsCollateArgs;

/* CPython-compatible Mersenne Twister (random.seed(int) for 0 <= seed < 2**64). */
void* ps_rng_create(uint64_t seed);
void ps_rng_destroy(void* rng);
void ps_rng_seed(void* rng, uint64_t seed);            /* re-seed in place: random.seed(seed) */
uint32_t ps_rng_randbelow(void* rng, uint32_t n);      /* random._randbelow(n), n >= 1 (test hook)  */
double ps_rng_random(void* rng);                       /* random.random() (test hook)               */

/* get_train_batch: samples are rows of (word ids [W], review id) — ItemPVDataset._data; batch_ids picks B of them.
 * out_u_items has row stride uprev_review_limit and is padded with prod_pad; *out_lmax = longest history of the
 * batch (util.pad pads to that width).  Returns 0, or an error code with ps_data_last_error() text. */
int ps_collate_train(const PsCorpusView* corpus, const PsCollateArgs* args, void* rng,
                     const int64_t* sample_words, const int64_t* sample_review, int64_t n_samples, int32_t W,
                     const int64_t* batch_ids, int32_t B,
                     int64_t* out_query_words /* [B,Q] */, int64_t* out_target /* [B] */,
                     int64_t* out_u_items /* [B,limit] */, int64_t* out_pos_words /* [B,W] */,
                     int64_t* out_query_idx /* [B] */, int64_t* out_user_idx /* [B] */,
                     int32_t* out_hist_len /* [B] */, int32_t* out_lmax);

/* get_test_batch: entries (query, user, product, review) + ragged candidate lists (CSR over the batch);
 * out_candi [B,candi_width] padded with prod_pad, histories as above with fix = 1. */
int ps_collate_test(const PsCorpusView* corpus, const PsCollateArgs* args,
                    const int64_t* entry_quad /* [B,4] query,user,product,review */, int32_t B,
                    const int64_t* candi_ptr /* [B+1] */, const int64_t* candi_items, int32_t candi_width,
                    int64_t* out_query_words, int64_t* out_target, int64_t* out_u_items, int64_t* out_candi,
                    int32_t* out_hist_len, int32_t* out_lmax);

/* ItemPVDataset.collect_train_samples (data/item_pv_dataset.py:73-93): for every train review, in review_info order,
 * random.shuffle its words IN PLACE (the shuffled order persists into later epochs, as there), keep word w when
 * rand[entry] <= sub_rate[w] (entry advances only on kept words, :84-88), and cut the kept stream — which runs across
 * review boundaries — into windows of W; a window is labelled with the review that completed it, the tail is padded with
 * word_pad.  rand = np.random.random(sum(review_length)) drawn by the caller.  Outputs: out_words [cap,W], out_review [cap];
 * *out_n windows written (cap >= total_words/W + 1). */
void ps_rng_shuffle(void* rng, int64_t* x, int64_t n);          /* random.shuffle(x) */
int ps_collect_train_samples(const int64_t* rw_ptr, int64_t* rw_words, int64_t n_reviews,
                             const int64_t* train_reviews, int64_t n_train, const double* rand, int64_t n_rand,
                             const double* sub_rate, int64_t vocab_size, int32_t W, int64_t word_pad, void* rng,
                             int64_t* out_words, int64_t* out_review, int64_t cap, int64_t* out_n);

const char* ps_data_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
