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

/* The epoch loop `for batch_data in dataloader` (trainer.py:64-66) with the collate on a NATIVE producer thread: the batches
 * of `order` (n_ids dataset rows in sampler order, cut into batches of B; the last one short unless drop_last) are built one
 * after another by ps_collate_train — the generator is consumed in exactly the sequential order, so seeded runs stay the
 * reference's — into a ring of `depth` caller-owned slots (pinned host buffers; u_items with row stride uprev_review_limit),
 * at most `depth` batches ahead of the consumer.  corpus, args' values, rng, samples, order and the slots' buffers must stay
 * alive until ps_epoch_stop.  A consumer that stops early leaves the generator up to `depth` batches further than the
 * sequential loop would have.
 *   ps_epoch_next    blocks until the next batch is ready; returns its slot (>= 0) with *out_B rows and *out_lmax = the
 *                    longest history (util.pad's width); -1 after the last batch; -2 on a collate error (ps_data_last_error).
 *   ps_epoch_release the consumer has finished reading the slot (its H2D copies have completed): it may be overwritten.
 *   ps_epoch_stop    stops and joins the thread, frees the handle (any time). */
typedef struct PsTrainSlot {
  int64_t* query_words;  /* [B,Q]     */
  int64_t* target;       /* [B]       */
  int64_t* u_items;      /* [B,limit] */
  int64_t* pos_words;    /* [B,W]     */
  int64_t* query_idx;    /* [B] or NULL */
  int64_t* user_idx;     /* [B] or NULL */
  int32_t* hist_len;     /* [B]       */
} PsTrainSlot;
void* ps_epoch_start(const PsCorpusView* corpus, const PsCollateArgs* args, void* rng,
                     const int64_t* sample_words, const int64_t* sample_review, int64_t n_samples, int32_t W,
                     const int64_t* order, int64_t n_ids, int32_t B, int32_t drop_last,
                     const PsTrainSlot* slots, int32_t depth /* 2..64 */);
int ps_epoch_next(void* epoch, int32_t* out_B, int32_t* out_lmax);
int ps_epoch_release(void* epoch, int32_t slot);
void ps_epoch_stop(void* epoch);

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

/* ------------------------------------------------------------------------------------------------------------
 * Review-transformer (RTM) batch builder: replaces the Python collate of
 *   ProdSearchDataLoader.prepare_train_batch / get_train_batch   data/prod_search_dataloader.py:196-262, :275-358
 *   ProdSearchDataLoader.get_test_batch                           data/prod_search_dataloader.py:44-109
 *   get_user_review_idxs / get_item_review_idxs                   data/prod_search_dataloader.py:135-194
 *   ProdSearchDataset.bisect_right, slide_padded_matrices_for_pv  data/prod_search_dataset.py:103-158
 * Sequences of REVIEW ids: [query | the user's previous reviews | the item's previous reviews], with the segment,
 * user and item id of every position.  `random.choice` / `random.sample` are drawn from the PsRng above; the
 * paragraph-vector branch also consumes numpy's legacy global generator (np.random.shuffle / permutation / random:
 * numpy/random/mtrand.pyx + distributions.c random_interval, MT19937 — third-party, pinned to numpy 2.2), whose raw
 * state the caller round-trips through ps_rng_get_state / ps_rng_set_state. */
typedef struct PsRtmCorpusView {
  int64_t n_reviews, n_users, n_products, n_queries;
  const int64_t* review_u_p;    /* [n_reviews,2] (user, product)                                             */
  const int64_t* u_seq_ptr;     /* [n_users+1]    CSR of global_data.u_r_seq (all reviews, time order)       */
  const int64_t* u_seq;
  const int64_t* i_seq_ptr;     /* [n_products+1] CSR of global_data.i_r_seq                                 */
  const int64_t* i_seq;
  const int64_t* ut_seq_ptr;    /* the same two sequences restricted to TRAIN reviews (`x in u_reviews[u]`,   */
  const int64_t* ut_seq;        /* `x in p_reviews[p]`), order kept — built once by the caller                */
  const int64_t* it_seq_ptr;
  const int64_t* it_seq;
  const int64_t* loc_time;      /* [n_reviews,3] review_loc_time: position in user seq, in item seq, time     */
  const int64_t* pq_ptr;        /* [n_products+1] CSR of prod_data.product_query_idx (train collate only)     */
  const int64_t* pq_idx;
  const int64_t* query_words;   /* [n_queries,Q]                                                              */
  int32_t Q;
  int32_t pad_;
} PsRtmCorpusView;

typedef struct PsRtmCollateArgs {
  int32_t uprev_review_limit;   /* --uprev_review_limit                                                       */
  int32_t iprev_review_limit;   /* --iprev_review_limit                                                       */
  int32_t do_seq;               /* train: --do_seq_review_train; test: --do_seq_review_test and not
                                   --train_review_only                                                       */
  int32_t neg_per_pos;          /* columns of neg_sample_products                                             */
  int64_t user_pad, prod_pad, review_pad;   /* user_size, product_size, review_count-1; the segment pad is 3  */
} PsRtmCollateArgs;

/* prepare_train_batch + the padding of get_train_batch.  rows = dataset entries (line_id, user, product, review);
 * neg_products = prod_data.neg_sample_products [n_lines, neg_per_pos].  Entries whose item has no usable review, or
 * whose negatives all have none, are dropped (:208-209, :243-245) AFTER consuming their random draws.  Outputs are
 * written COMPACTLY with the batch's own widths, returned in dims = {Bk, Rp, Kk, Rn}:
 *   query_words [Bk,Q]   pos_ridxs [Bk,Rp]       pos_seg / pos_user / pos_item [Bk,Rp+1]
 *   kept [Bk] (index into rows)  neg_ridxs [Bk,Kk,Rn]    neg_seg / neg_user / neg_item [Bk,Kk,Rn+1]
 * Buffers must hold the worst case: B rows, neg_per_pos negatives, uprev+iprev reviews. */
int ps_rtm_collate_train(const PsRtmCorpusView* corpus, const PsRtmCollateArgs* args, void* rng,
                         const int64_t* rows /* [B,4] */, int32_t B,
                         const int64_t* neg_products, int64_t n_lines,
                         int64_t* out_query_words, int64_t* out_kept,
                         int64_t* out_pos_ridxs, int64_t* out_pos_seg, int64_t* out_pos_user, int64_t* out_pos_item,
                         int64_t* out_neg_ridxs, int64_t* out_neg_seg, int64_t* out_neg_user, int64_t* out_neg_item,
                         int32_t dims[4]);

/* get_test_batch: entries (query, user, product, review) + ragged candidate lists (CSR over the batch).
 * dims = {C, Rc}: C = longest candidate list, Rc = longest review sequence.  Compact outputs:
 *   query_words [B,Q]  candi [B,C] (pad -1, :95)  candi_ridxs [B,C,Rc]  candi_seg / candi_user / candi_item [B,C,Rc+1]
 * Buffers must hold B * C * (uprev+iprev+1) elements. */
int ps_rtm_collate_test(const PsRtmCorpusView* corpus, const PsRtmCollateArgs* args,
                        const int64_t* entry_quad /* [B,4] */, int32_t B,
                        const int64_t* candi_ptr /* [B+1] */, const int64_t* candi_items,
                        int64_t* out_query_words, int64_t* out_candi,
                        int64_t* out_ridxs, int64_t* out_seg, int64_t* out_user, int64_t* out_item, int32_t dims[2]);

/* The paragraph-vector branch of get_train_batch (:301-345), for N = Bk*Rp review slots of WL words:
 *   words [Bk,Rp,WL] (review_words[pos_ridxs], gathered by the caller) is shuffled IN PLACE exactly as
 *   shuffle_words_in_reviews does it — np.random.shuffle on each [Rp,WL] slice permutes that entry's REVIEW ROWS;
 *   masks [Bk,Rp,WL] = (word != word_pad) [& np.random.random < sub_rate[word] when sub_rate != NULL], taken BEFORE the
 *   shuffle (:289-291), so they stay in the unshuffled order like the reference's;
 *   both are cut into windows of W words (right-padded to a multiple of W), giving seg = ceil(WL/W) sub-batches whose
 *   rows are mixed by np.random.permutation(Bk*seg) when `permute` (the DataLoader's shuffle flag).
 * Outputs: slide_words / slide_masks [seg,Bk,Rp,W], batch_index [seg,Bk] (row of the collated batch each sub-batch
 * row comes from).  np_rng holds numpy's global MT19937 state (ps_rng_set_state) and is advanced. */
int ps_rtm_pv_windows(void* np_rng, int64_t* words, uint8_t* masks, int32_t Bk, int32_t Rp, int32_t WL, int32_t W,
                      int64_t word_pad, const double* sub_rate, int64_t vocab_size, int32_t shuffle_rows, int32_t permute,
                      int64_t* slide_words, uint8_t* slide_masks, int64_t* batch_index);

/* get_pv_word_masks of the non-PV branch (:347-349) when sub_rate != NULL: masks = (word != pad) & (rand < rate[word]) */
int ps_rtm_word_masks(void* np_rng, const int64_t* words, int64_t n, int64_t word_pad, const double* sub_rate,
                      int64_t vocab_size, uint8_t* masks);

/* Raw MT19937 state (624 words + position) — `random.getstate()[1]` / `np.random.get_state()[1:3]` layout. */
void ps_rng_get_state(void* rng, uint32_t key[624], int32_t* pos);
void ps_rng_set_state(void* rng, const uint32_t key[624], int32_t pos);
/* numpy legacy draws (test hooks): random_interval(max) and random_sample() */
uint64_t ps_rng_np_interval(void* rng, uint64_t max);

const char* ps_data_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
