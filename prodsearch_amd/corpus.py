"""Corpus readers and the TEM dataset.

Mirrors, attribute for attribute, of the reference's
  ``GlobalProdSearchData``  data/data_util.py:166-287   (gz text corpus: ids, vocab, queries, review text, sequences)
  ``ProdSearchData``        data/data_util.py:10-164    (per split: vocab distribution, sub-sampling rates, negative
                                                          sampling distributions, queries per product, review lists)
  ``ItemPVDataset``         data/item_pv_dataset.py     (train samples per epoch, (user, query) test entries)
so that ``main.py``'s wiring (``main.py:179-186``, ``trainer.py:44-60``) works unchanged on top of them.  The per-epoch
sample collection — a Python loop over every word of every training review in the reference — runs in C++
(``ps_collect_train_samples``) on a CSR copy of the review text and consumes the same random streams
(``np.random.random`` for sub-sampling, the CPython-compatible generator of ``pyrandom`` for the shuffles).
"""
import gzip
import os

import numpy as np

from . import _lib, pyrandom


def _text_lines(path):
    with gzip.open(path, 'rt') as f:
        return [ln.strip() for ln in f]


def _int_rows(path):
    with gzip.open(path, 'rt') as f:
        return [[int(t) for t in ln.split()] for ln in f]


class GlobalProdSearchData(object):
    def __init__(self, args, data_path, input_train_dir):
        j = os.path.join
        self.product_ids = _text_lines(j(data_path, 'product.txt.gz'))
        self.product_asin2ids = {a: i for i, a in enumerate(self.product_ids)}
        self.product_size = len(self.product_ids)
        self.user_ids = _text_lines(j(data_path, 'users.txt.gz'))
        self.user_size = len(self.user_ids)
        self.words = _text_lines(j(data_path, 'vocab.txt.gz'))
        self.vocab_size = len(self.words) + 1
        self.word_pad_idx = self.vocab_size - 1
        q = _int_rows(j(input_train_dir, 'query.txt.gz'))
        width = max(len(x) for x in q)
        self.query_words = [x + [self.word_pad_idx] * (width - len(x)) for x in q]
        self.review_words = _int_rows(j(data_path, 'review_text.txt.gz'))
        self.review_length = [len(x) for x in self.review_words]
        self.review_count = len(self.review_words) + 1
        if args.model_name == 'review_transformer':
            self.review_words.append([self.word_pad_idx])
            if args.do_subsample_mask:
                lim = args.review_word_limit
                self.review_words = [x[:lim] + [self.word_pad_idx] * (lim - len(x)) for x in self.review_words]
        self.u_r_seq = _int_rows(j(data_path, 'u_r_seq.txt.gz'))
        self.i_r_seq = _int_rows(j(data_path, 'p_r_seq.txt.gz'))
        self.review_loc_time = _int_rows(j(data_path, 'review_uloc_ploc_and_time.txt.gz'))
        self.line_review_id_map = self.read_review_id_line_map(j(data_path, 'review_id.txt.gz'))
        self.train_review_info, self.train_query_idxs = self.read_review_id(
            j(input_train_dir, 'train_id.txt.gz'), self.line_review_id_map)
        self.review_u_p = _int_rows(j(data_path, 'review_u_p.txt.gz'))
        self.padded_review_words = None

    def set_padded_review_words(self, review_words):
        self.padded_review_words = review_words

    @staticmethod
    def read_review_id_line_map(fname):
        return {int(ln.rsplit('_', 1)[-1]): i for i, ln in enumerate(_text_lines(fname))}

    @staticmethod
    def read_review_id(fname, line_review_id_map):
        info, query_ids = [], []
        for no, ln in enumerate(_text_lines(fname)):
            cols = ln.split('\t')
            info.append((no, int(cols[0]), int(cols[1]), line_review_id_map[int(cols[2].rsplit('_', 1)[-1])]))
            if cols[-1].isdigit():
                query_ids.append(int(cols[-1]))
        return info, query_ids

    read_arr_from_lines = staticmethod(_int_rows)
    read_lines = staticmethod(_text_lines)


class ProdSearchData(object):
    def __init__(self, args, input_train_dir, set_name, global_data):
        gd = global_data
        j = os.path.join
        self.args = args
        self.neg_per_pos = args.neg_per_pos
        self.set_name = set_name
        self.global_data = gd
        self.product_size, self.user_size, self.vocab_size = gd.product_size, gd.user_size, gd.vocab_size
        self.sub_sampling_rate = None
        self.neg_sample_products = None
        self.word_dists = None
        self.uq_pids = None
        self.subsampling_rate = 0 if args.fix_emb else args.subsampling_rate
        if set_name == 'train':
            counts = np.zeros(self.vocab_size)
            with gzip.open(j(input_train_dir, 'train.txt.gz'), 'rt') as f:
                for ln in f:
                    ids = np.asarray(ln.strip().split('\t')[2].split(' '), dtype=np.int64)
                    np.add.at(counts, ids, 1)
            self.vocab_distribute = counts.tolist()
            self.sub_sampling(self.subsampling_rate)
            self.word_dists = self.neg_distributes(self.vocab_distribute)
            self.product_query_idx = _int_rows(j(input_train_dir, 'train_query_idx.txt.gz'))
            self.review_info = gd.train_review_info
            self.review_query_idx = gd.train_query_idxs
        else:
            read_name = set_name if args.has_valid else 'test'
            self.product_query_idx = _int_rows(j(input_train_dir, 'test_query_idx.txt.gz'))
            self.review_info, self.review_query_idx = GlobalProdSearchData.read_review_id(
                j(input_train_dir, '%s_id.txt.gz' % read_name), gd.line_review_id_map)
            ranklist = j(input_train_dir, '%s.bias_product.ranklist' % read_name)
            if args.test_candi_size > 0 and os.path.exists(ranklist):
                self.uq_pids = self.read_ranklist(ranklist, gd.product_asin2ids)
        self.u_reviews = [set() for _ in range(self.user_size)]
        self.p_reviews = [set() for _ in range(self.product_size)]
        for _, u, p, r in gd.train_review_info:
            self.u_reviews[u].add(r)
            self.p_reviews[p].add(r)
        if args.prod_freq_neg_sample:
            self.product_distribute = np.zeros(self.product_size)
            for _, _, p, _ in gd.train_review_info:
                self.product_distribute[p] += 1
        else:
            self.product_distribute = np.ones(self.product_size)
        self.product_dists = self.neg_distributes(self.product_distribute)
        self.set_review_size = len(self.review_info)

    @staticmethod
    def read_ranklist(fname, product_asin2ids):
        from collections import defaultdict
        out = defaultdict(list)
        with open(fname) as f:
            for ln in f:
                cols = ln.strip().split(' ')
                uid, qid = cols[0].split('_')
                out[(uid, int(qid))].append(product_asin2ids[cols[2]])
        return out

    def sub_sampling(self, subsample_threshold):
        """word2vec-style keep probability per word (data_util.py:139-155); 0 for words absent from training."""
        c = np.asarray(self.vocab_distribute, dtype=np.float64)
        rate = np.ones(self.vocab_size)
        if subsample_threshold != 0.0:
            thr = sum(self.vocab_distribute) * subsample_threshold
            for i in range(self.vocab_size):              # scalar arithmetic in the reference's order of operations
                if c[i] == 0:
                    rate[i] = 0
                else:
                    rate[i] = min(1.0, (np.sqrt(float(c[i]) / thr) + 1) * thr / float(c[i]))
            self.sample_count = sum(rate[i] * self.vocab_distribute[i] for i in range(self.vocab_size))
        self.sub_sampling_rate = rate

    @staticmethod
    def neg_distributes(weights, distortion=0.75):
        w = np.asarray(weights)
        wf = np.power(w / w.sum(), distortion)
        return wf / wf.sum()

    def initialize_epoch(self):
        """data_util.py:92-117.  TEM: nothing.  RTM: the epoch's negative products and, unless ``do_subsample_mask``,
        the sub-sampled review texts cut/padded to ``review_word_limit`` (``global_data.padded_review_words``, here a
        ``[review_count, limit]`` int64 array).  The reference never advances ``entry_id`` in its filter loop
        (:104-112), so EVERY word is tested against the first of the ``sum(review_length)`` numbers it draws — kept
        as is, it decides what the reference trains on."""
        if self.args.model_name == 'item_transformer':
            return
        self.neg_sample_products = np.random.choice(self.product_size, size=(self.set_review_size, self.neg_per_pos),
                                                    replace=True, p=self.product_dists)
        if self.args.do_subsample_mask:
            return
        gd = self.global_data
        rand_numbers = np.random.random(sum(gd.review_length))
        ptr, flat = _rtm_review_csr(gd)
        keep = ~(rand_numbers[0] > np.asarray(self.sub_sampling_rate)[flat])
        n_rev, limit, pad = len(ptr) - 1, int(self.args.review_word_limit), gd.word_pad_idx
        out = np.full((n_rev + 1, limit), pad, dtype=np.int64)
        rid = np.repeat(np.arange(n_rev), np.diff(ptr))[keep]
        kept = np.bincount(rid, minlength=n_rev)
        start = np.concatenate([[0], np.cumsum(kept)[:-1]])
        col = np.arange(rid.size) - start[rid]
        m = col < limit
        out[rid[m], col[m]] = flat[keep][m]
        gd.set_padded_review_words(out)


def _rtm_review_csr(global_data):
    """CSR of the review texts without the appended pad review (review_transformer corpus, un-padded lists)."""
    csr = getattr(global_data, '_rtm_csr', None)
    if csr is None:
        rw = global_data.review_words[:len(global_data.review_length)]
        ptr = np.zeros(len(rw) + 1, dtype=np.int64)
        ptr[1:] = np.cumsum([len(x) for x in rw])
        flat = np.fromiter((w for x in rw for w in x), dtype=np.int64, count=int(ptr[-1]))
        csr = global_data._rtm_csr = (ptr, flat)
    return csr


class ItemPVDataset(object):
    _always_materialize = False

    def __init__(self, args, global_data, prod_data):
        self.args = args
        self.valid_candi_size = args.valid_candi_size
        self.prod_pad_idx = global_data.product_size
        self.word_pad_idx = global_data.vocab_size - 1
        self.seg_pad_idx = 3
        self.pv_window_size = args.pv_window_size
        self.train_review_only = args.train_review_only
        self.uprev_review_limit = args.uprev_review_limit
        self.global_data = global_data
        self.prod_data = prod_data
        if prod_data.set_name == 'train':
            self.sample_words, self.sample_review = self.collect_train_samples(global_data, prod_data)
            self._data = _TrainSamples(self.sample_words, self.sample_review)
        else:
            self._data = self.collect_test_samples(global_data, prod_data, args.candi_batch_size)

    @staticmethod
    def _review_csr(global_data):
        """CSR copy of ``global_data.review_words`` kept on the object: the epoch shuffles persist in it, as the
        in-place ``random.shuffle`` of the reference persists in its lists (item_pv_dataset.py:81)."""
        csr = getattr(global_data, '_review_csr', None)
        if csr is None:
            rw = global_data.review_words[:len(global_data.review_length)]
            ptr = np.zeros(len(rw) + 1, dtype=np.int64)
            ptr[1:] = np.cumsum([len(x) for x in rw])
            flat = np.fromiter((w for x in rw for w in x), dtype=np.int64, count=int(ptr[-1]))
            csr = global_data._review_csr = (ptr, flat)
        return csr

    def collect_train_samples(self, global_data, prod_data):
        lib = _lib.load_data()
        ptr, flat = self._review_csr(global_data)
        rand = np.random.random(sum(global_data.review_length))            # item_pv_dataset.py:77
        reviews = np.asarray([r for _, _, _, r in prod_data.review_info], dtype=np.int64)
        rate = np.ascontiguousarray(prod_data.sub_sampling_rate, dtype=np.float64)
        W = int(self.pv_window_size)
        cap = int(ptr[-1]) // W + 2
        words = np.empty((cap, W), dtype=np.int64)
        rev = np.empty(cap, dtype=np.int64)
        n = np.zeros(1, dtype=np.int64)
        _lib.check_data(lib.ps_collect_train_samples(
            ptr.ctypes.data, flat.ctypes.data, len(ptr) - 1, reviews.ctypes.data, len(reviews), rand.ctypes.data, len(rand),
            rate.ctypes.data, len(rate), W, self.word_pad_idx, pyrandom.handle(), words.ctypes.data, rev.ctypes.data, cap,
            n.ctypes.data), 'ps_collect_train_samples')
        return np.ascontiguousarray(words[:int(n[0])]), np.ascontiguousarray(rev[:int(n[0])])

    def collect_test_samples(self, global_data, prod_data, candi_batch_size=1000):
        """(query, user, product, review, candidates) per distinct (user, query) (item_pv_dataset.py:36-70).  With all
        products as candidates the list is left as ``None`` — ``evaluate.rank_all`` ranks the catalogue on the device —
        unless ``args.materialize_candidates`` asks for the reference's chunked lists."""
        out, seen = [], set()
        full = prod_data.uq_pids is None and not (prod_data.set_name == 'valid' and self.valid_candi_size > 1)
        for _, user_idx, prod_idx, review_idx in prod_data.review_info:
            for query_idx in prod_data.product_query_idx[prod_idx]:
                if (user_idx, query_idx) in seen:
                    continue
                seen.add((user_idx, query_idx))
                if full and not (self._always_materialize or getattr(self.args, 'materialize_candidates', False)):
                    out.append([query_idx, user_idx, prod_idx, review_idx, None])
                    continue
                if prod_data.uq_pids is None:
                    if not full:
                        cands = np.random.choice(global_data.product_size, size=self.valid_candi_size - 1,
                                                 replace=False, p=prod_data.product_dists).tolist()
                        cands.append(prod_idx)
                        pyrandom.shuffle(cands)
                    else:
                        cands = list(range(global_data.product_size))
                else:
                    cands = prod_data.uq_pids[(global_data.user_ids[user_idx], query_idx)]
                    pyrandom.shuffle(cands)
                for s in range(0, len(cands), candi_batch_size):
                    out.append([query_idx, user_idx, prod_idx, review_idx, cands[s:s + candi_batch_size]])
        return out

    def __len__(self):
        return len(self._data)

    def __getitem__(self, index):
        return self._data[index]


class _TrainSamples(object):
    """List-like view ``[[word ids], review id]`` over the sample arrays (what ``ItemPVDataset._data`` holds)."""

    def __init__(self, words, review):
        self.words, self.review = words, review

    def __len__(self):
        return len(self.review)

    def __getitem__(self, i):
        return [self.words[i].tolist(), int(self.review[i])]

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


class ProdSearchDataset(ItemPVDataset):
    """``data/prod_search_dataset.py:ProdSearchDataset``: train entries are ``prod_data.review_info`` rows
    ``(line, user, product, review)`` (:86-89); evaluation entries are the (user, query) pairs with their candidate
    chunks (:43-83, the same walk as the TEM dataset — always materialised: every candidate is its own sequence)."""
    _always_materialize = True

    def __init__(self, args, global_data, prod_data):
        self.args = args
        self.valid_candi_size = args.valid_candi_size
        self.user_pad_idx = global_data.user_size
        self.prod_pad_idx = global_data.product_size
        self.word_pad_idx = global_data.vocab_size - 1
        self.review_pad_idx = global_data.review_count - 1
        self.seg_pad_idx = 3
        self.shuffle_review_words = args.shuffle_review_words
        self.review_encoder_name = args.review_encoder_name
        self.pv_window_size = args.pv_window_size
        self.corrupt_rate = args.corrupt_rate
        self.train_review_only = args.train_review_only
        self.uprev_review_limit = args.uprev_review_limit
        self.iprev_review_limit = args.iprev_review_limit
        self.total_review_limit = self.uprev_review_limit + self.iprev_review_limit
        self.global_data = global_data
        self.prod_data = prod_data
        if prod_data.set_name == 'train':
            self._data = prod_data.review_info
        else:
            self._data = self.collect_test_samples(global_data, prod_data, args.candi_batch_size)
