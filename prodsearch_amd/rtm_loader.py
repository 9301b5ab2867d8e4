"""ProdSearchDataLoader — the reference's review-transformer loader with the collate moved into C++.

Mirror of ``data/prod_search_dataloader.py:ProdSearchDataLoader`` (constructor ``(args, dataset, prepare_pv, batch_size,
shuffle)``; iteration yields ``None`` for a batch with no usable entry, one ``ProdSearchTrainBatch``, or — in the
paragraph-vector epochs — the sequence of window sub-batches the reference returns as a list; evaluation yields
``ProdSearchTestBatch``).  MI355X-first split of the work:

* host, C++ (``ps_rtm_collate_train`` / ``ps_rtm_collate_test``, include/prodsearch_data.h): the review-id sequences
  ``[query | user's previous reviews | item's previous reviews]`` with their segment / user / item ids — CSR walks and
  the reference's ``random.choice`` / ``random.sample`` draws on the CPython-compatible generator (``pyrandom``);
* host, C++ (``ps_rtm_pv_windows``): the paragraph-vector windows with numpy's legacy generator, state round-tripped
  through ``np.random.get_state`` / ``set_state`` so ``np.random.seed`` reproduces the reference's stream;
* device: the review-word tensors (``review_words[ridxs]``, up to 51 MB per batch at the reference's sizes) are gathered
  from a table resident in HBM instead of being built on the host and copied.

Batch order: ``DataLoader(shuffle=True, num_workers=0)``'s (``dataloader.sampler_batches``), so ``torch.manual_seed``,
``random.seed`` -> ``pyrandom.seed`` and ``np.random.seed`` give batches bit-identical to the reference's.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, pyrandom
from .dataloader import _csr, prefetch_iter, sampler_batches
from .rtm_data import ProdSearchTestBatch, ProdSearchTrainBatch

_np_handle = None


def _numpy_rng():
    """PsRng loaded with np.random's current MT19937 state; ``_numpy_rng_store`` writes the advanced state back."""
    global _np_handle
    lib = _lib.load_data()
    if _np_handle is None:
        _np_handle = lib.ps_rng_create(0)
    st = np.random.get_state()
    key = np.ascontiguousarray(st[1], dtype=np.uint32)
    lib.ps_rng_set_state(_np_handle, key.ctypes.data, int(st[2]))
    return _np_handle, st


def _numpy_rng_store(st):
    lib = _lib.load_data()
    key = np.empty(624, dtype=np.uint32)
    pos = C.c_int32(0)
    lib.ps_rng_get_state(_np_handle, key.ctypes.data, C.byref(pos))
    np.random.set_state((st[0], key, pos.value, st[3], st[4]))


class RtmCorpus(object):
    """Flat arrays of ``global_data`` / ``prod_data`` for the C collate (host memory, borrowed per call)."""

    def __init__(self, global_data, prod_data):
        gd, pd = global_data, prod_data
        self.review_u_p = np.ascontiguousarray(np.asarray(gd.review_u_p, dtype=np.int64).reshape(-1, 2))
        n_rev = self.review_u_p.shape[0]
        self.u_seq_ptr, self.u_seq = _csr(gd.u_r_seq)
        self.i_seq_ptr, self.i_seq = _csr(gd.i_r_seq)
        self.n_users, self.n_products = len(gd.u_r_seq), len(gd.i_r_seq)
        # ``x in prod_data.u_reviews[u]`` / ``x in prod_data.p_reviews[p]`` for x in that owner's sequence: both sets are
        # filled from global_data.train_review_info (data_util.py:45-50), so membership = "train review of this owner"
        tu = np.full(n_rev, -1, dtype=np.int64)
        tp = np.full(n_rev, -1, dtype=np.int64)
        info = np.asarray(gd.train_review_info, dtype=np.int64).reshape(-1, 4)
        tu[info[:, 3]] = info[:, 1]
        tp[info[:, 3]] = info[:, 2]
        self.ut_seq_ptr, self.ut_seq = self._restrict(self.u_seq_ptr, self.u_seq, tu)
        self.it_seq_ptr, self.it_seq = self._restrict(self.i_seq_ptr, self.i_seq, tp)
        self.loc_time = np.ascontiguousarray(np.asarray(gd.review_loc_time, dtype=np.int64).reshape(-1, 3))
        self.pq_ptr, self.pq_idx = _csr(pd.product_query_idx)
        self.query_words = np.ascontiguousarray(np.asarray(gd.query_words, dtype=np.int64))
        v = self.view = _lib.PsRtmCorpusView()
        v.n_reviews, v.n_users, v.n_products, v.n_queries = n_rev, self.n_users, self.n_products, self.query_words.shape[0]
        for name in ('review_u_p', 'u_seq_ptr', 'u_seq', 'i_seq_ptr', 'i_seq', 'ut_seq_ptr', 'ut_seq', 'it_seq_ptr',
                     'it_seq', 'loc_time', 'pq_ptr', 'pq_idx', 'query_words'):
            setattr(v, name, getattr(self, name).ctypes.data)
        v.Q = self.query_words.shape[1]

    @staticmethod
    def _restrict(ptr, seq, owner_of_train_review):
        owner = np.repeat(np.arange(len(ptr) - 1, dtype=np.int64), np.diff(ptr))
        keep = owner_of_train_review[seq] == owner
        cnt = np.bincount(owner[keep], minlength=len(ptr) - 1)
        out = np.zeros(len(ptr), dtype=np.int64)
        out[1:] = np.cumsum(cnt)
        return out, np.ascontiguousarray(seq[keep])


class _SubBatches(object):
    """The paragraph-vector sub-batches of one collate (the reference returns them as a list, :337-345), assembled on
    the device one at a time: materialising all ``ceil(review_word_limit / pv_window_size)`` of them would hold that
    many copies of the negatives' review words."""

    def __init__(self, base, slide_words, slide_masks, batch_index, pos_words, review_table):
        self.base, self.sw, self.sm, self.bi = base, slide_words, slide_masks, batch_index
        self.pos_words, self.table = pos_words, review_table

    def __len__(self):
        return self.sw.shape[0]

    def __getitem__(self, i):
        if i < 0 or i >= len(self):
            raise IndexError(i)
        b, ix = self.base, self.bi[i]
        neg_r = b['neg_r'][ix]
        return ProdSearchTrainBatch(b['qw'][ix], b['pos_r'][ix], b['pos_seg'][ix], self.sw[i], self.sm[i], neg_r,
                                    b['neg_seg'][ix], b['pos_u'][ix], b['neg_u'][ix], b['pos_i'][ix], b['neg_i'][ix],
                                    pos_prod_rword_idxs_pvc=self.pos_words[ix], neg_prod_rword_idxs_pvc=self.table[neg_r],
                                    to_tensor=False)

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


class ProdSearchDataLoader(object):
    def __init__(self, args, dataset, prepare_pv=True, batch_size=1, shuffle=False, drop_last=False, device=None,
                 pin_memory=True, prefetch=0, **_ignored):
        self.args = args
        self.dataset = dataset
        self.prepare_pv = prepare_pv
        self.batch_size = batch_size
        self.shuffle, self.drop_last = bool(shuffle), bool(drop_last)
        self.prod_pad_idx, self.user_pad_idx = dataset.prod_pad_idx, dataset.user_pad_idx
        self.review_pad_idx, self.word_pad_idx = dataset.review_pad_idx, dataset.word_pad_idx
        self.seg_pad_idx = dataset.seg_pad_idx
        self.global_data, self.prod_data = dataset.global_data, dataset.prod_data
        self.shuffle_review_words = dataset.shuffle_review_words
        self.total_review_limit = args.uprev_review_limit + args.iprev_review_limit
        self.device = None if device in (None, 'cpu') else torch.device(device)
        self._pin = bool(pin_memory) and self.device is not None and torch.cuda.is_available()
        self._lib = _lib.load_data()
        self.prefetch = int(prefetch)
        self._train = self.prod_data.set_name == 'train'
        self.corpus = RtmCorpus(self.global_data, self.prod_data)
        if args.do_subsample_mask:                               # :31-36
            table, self.sub_sampling_rate = self.global_data.review_words, self.prod_data.sub_sampling_rate
        else:
            table, self.sub_sampling_rate = self.global_data.padded_review_words, None
        if self._train:
            if table is None:
                raise RuntimeError("padded_review_words is not set: call prod_data.initialize_epoch() first "
                                   "(trainer.py:56)")
            self.review_words = np.ascontiguousarray(np.asarray(table, dtype=np.int64))
            if self.review_words.ndim != 2:
                raise ValueError("the review-word table must be padded to [review_count, review_word_limit]")
            self._table = torch.from_numpy(self.review_words)
            if self.device is not None:
                self._table = self._table.to(self.device)
            self._rate = None if self.sub_sampling_rate is None else \
                np.ascontiguousarray(self.sub_sampling_rate, dtype=np.float64)
            self.rows = np.ascontiguousarray(np.asarray(dataset._data, dtype=np.int64).reshape(-1, 4))
            self.neg_products = np.ascontiguousarray(self.prod_data.neg_sample_products, dtype=np.int64)
        else:
            data = dataset._data
            self.entry_quad = np.ascontiguousarray(np.asarray([e[:4] for e in data], dtype=np.int64).reshape(len(data), 4))
            self.candi_ptr, self.candi_items = _csr([e[4] for e in data])

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    # ------------------------------------------------------------------ plumbing
    def _cargs(self, do_seq):
        a = _lib.PsRtmCollateArgs()
        a.uprev_review_limit, a.iprev_review_limit = int(self.args.uprev_review_limit), int(self.args.iprev_review_limit)
        a.do_seq, a.neg_per_pos = int(bool(do_seq)), int(self.args.neg_per_pos)
        a.user_pad, a.prod_pad, a.review_pad = self.user_pad_idx, self.prod_pad_idx, self.review_pad_idx
        return a

    def _host(self, *shape, dtype=torch.int64):
        return torch.empty(*shape, dtype=dtype, pin_memory=self._pin)

    def _ship(self, t):
        return t if self.device is None else t.to(self.device, non_blocking=True)

    # ------------------------------------------------------------------ train (prod_search_dataloader.py:196-358)
    def train_batch_from_ids(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        rows = np.ascontiguousarray(self.rows[ids])
        B, K, T, Q = len(ids), int(self.args.neg_per_pos), self.total_review_limit, self.corpus.view.Q
        qw, kept = self._host(B, Q), torch.empty(B, dtype=torch.int64)
        pos_r, pos_x = self._host(B * T), [self._host(B * (T + 1)) for _ in range(3)]
        neg_r, neg_x = self._host(B * K * T), [self._host(B * K * (T + 1)) for _ in range(3)]
        dims = (C.c_int32 * 4)()
        a = self._cargs(self.args.do_seq_review_train)
        _lib.check_data(self._lib.ps_rtm_collate_train(
            C.byref(self.corpus.view), C.byref(a), pyrandom.handle(), rows.ctypes.data, B,
            self.neg_products.ctypes.data, self.neg_products.shape[0], qw.data_ptr(), kept.data_ptr(),
            pos_r.data_ptr(), pos_x[0].data_ptr(), pos_x[1].data_ptr(), pos_x[2].data_ptr(),
            neg_r.data_ptr(), neg_x[0].data_ptr(), neg_x[1].data_ptr(), neg_x[2].data_ptr(), dims), 'ps_rtm_collate_train')
        Bk, Rp, Kk, Rn = (int(x) for x in dims)
        if Bk == 0:
            print("0 available instance in the batch")            # :279-281
            return None
        host = dict(qw=qw[:Bk], pos_r=pos_r[:Bk * Rp].view(Bk, Rp), neg_r=neg_r[:Bk * Kk * Rn].view(Bk, Kk, Rn))
        for name, t in zip(('pos_seg', 'pos_u', 'pos_i'), pos_x):
            host[name] = t[:Bk * (Rp + 1)].view(Bk, Rp + 1)
        for name, t in zip(('neg_seg', 'neg_u', 'neg_i'), neg_x):
            host[name] = t[:Bk * Kk * (Rn + 1)].view(Bk, Kk, Rn + 1)
        pv = 'pv' in self.dataset.review_encoder_name and self.prepare_pv
        if pv:
            return self._pv_batches(host, Bk, Rp)
        dev = {k: self._ship(v) for k, v in host.items()}
        pad = self.word_pad_idx
        if self._rate is None:                                   # masks = words != pad, on the device
            pos_w, neg_w = self._table[dev['pos_r']], self._table[dev['neg_r']]
            pos_m, neg_m = (pos_w != pad).to(torch.uint8), (neg_w != pad).to(torch.uint8)
        else:                                                    # get_pv_word_masks draws np.random.random (:90-94)
            pos_hw = np.ascontiguousarray(self.review_words[host['pos_r'].numpy()])
            neg_hw = np.ascontiguousarray(self.review_words[host['neg_r'].numpy()])
            pos_hm, neg_hm = self._host(*pos_hw.shape, dtype=torch.uint8), self._host(*neg_hw.shape, dtype=torch.uint8)
            h, st = _numpy_rng()
            for w, m in ((pos_hw, pos_hm), (neg_hw, neg_hm)):    # positives first (:289), negatives second (:347)
                _lib.check_data(self._lib.ps_rtm_word_masks(h, w.ctypes.data, w.size, pad, self._rate.ctypes.data,
                                                            len(self._rate), m.data_ptr()), 'ps_rtm_word_masks')
            _numpy_rng_store(st)
            pos_w, neg_w = self._table[dev['pos_r']], self._table[dev['neg_r']]
            pos_m, neg_m = self._ship(pos_hm), self._ship(neg_hm)
        return ProdSearchTrainBatch(dev['qw'], dev['pos_r'], dev['pos_seg'], pos_w, pos_m, dev['neg_r'], dev['neg_seg'],
                                    dev['pos_u'], dev['neg_u'], dev['pos_i'], dev['neg_i'], neg_prod_rword_idxs=neg_w,
                                    neg_prod_rword_masks=neg_m, to_tensor=False)

    def _pv_batches(self, host, Bk, Rp):
        WL, W, pad = self.review_words.shape[1], int(self.dataset.pv_window_size), self.word_pad_idx
        seg = (WL + W - 1) // W
        words = self._host(Bk, Rp, WL)
        np.take(self.review_words, host['pos_r'].numpy().reshape(-1), axis=0,
                out=words.numpy().reshape(Bk * Rp, WL))
        masks = torch.empty(Bk, Rp, WL, dtype=torch.uint8)
        sw, sm = self._host(seg, Bk, Rp, W), self._host(seg, Bk, Rp, W, dtype=torch.uint8)
        bi = self._host(seg, Bk)
        h, st = _numpy_rng()
        _lib.check_data(self._lib.ps_rtm_pv_windows(
            h, words.data_ptr(), masks.data_ptr(), Bk, Rp, WL, W, pad,
            None if self._rate is None else self._rate.ctypes.data, 0 if self._rate is None else len(self._rate),
            int(bool(self.shuffle_review_words)), int(self.shuffle), sw.data_ptr(), sm.data_ptr(), bi.data_ptr()),
            'ps_rtm_pv_windows')
        _numpy_rng_store(st)
        dev = {k: self._ship(v) for k, v in host.items()}
        return _SubBatches(dev, self._ship(sw), self._ship(sm), self._ship(bi), self._ship(words), self._table)

    # ------------------------------------------------------------------ evaluation (:44-109)
    def test_batch_from_ids(self, ids):
        ids = np.asarray(ids, dtype=np.int64)
        quad = np.ascontiguousarray(self.entry_quad[ids])
        lens = self.candi_ptr[ids + 1] - self.candi_ptr[ids]
        cptr = np.zeros(len(ids) + 1, dtype=np.int64)
        cptr[1:] = np.cumsum(lens)
        items = np.concatenate([self.candi_items[self.candi_ptr[i]:self.candi_ptr[i + 1]] for i in ids]) \
            if len(ids) else np.zeros(0, dtype=np.int64)
        items = np.ascontiguousarray(items, dtype=np.int64)
        B, Cw, T, Q = len(ids), int(lens.max()), self.total_review_limit, self.corpus.view.Q
        qw, candi = self._host(B, Q), torch.empty(B, Cw, dtype=torch.int64)
        ridx = self._host(B * Cw * T)
        xs = [self._host(B * Cw * (T + 1)) for _ in range(3)]
        dims = (C.c_int32 * 2)()
        do_seq = self.args.do_seq_review_test and not self.args.train_review_only      # :63
        a = self._cargs(do_seq)
        _lib.check_data(self._lib.ps_rtm_collate_test(
            C.byref(self.corpus.view), C.byref(a), quad.ctypes.data, B, cptr.ctypes.data, items.ctypes.data,
            qw.data_ptr(), candi.data_ptr(), ridx.data_ptr(), xs[0].data_ptr(), xs[1].data_ptr(), xs[2].data_ptr(), dims),
            'ps_rtm_collate_test')
        Cw, Rc = int(dims[0]), int(dims[1])
        seg, usr, itm = (self._ship(t[:B * Cw * (Rc + 1)].view(B, Cw, Rc + 1)) for t in xs)
        return ProdSearchTestBatch(quad[:, 0].tolist(), quad[:, 1].tolist(), quad[:, 2].tolist(), candi,
                                   self._ship(qw), self._ship(ridx[:B * Cw * Rc].view(B, Cw, Rc)), seg, usr, itm,
                                   to_tensor=False)

    # ------------------------------------------------------------------ the reference's collate_fn entry points
    def get_train_batch(self, batch):
        rows = np.asarray(batch, dtype=np.int64).reshape(-1, 4)
        saved, self.rows = self.rows, rows
        try:
            return self.train_batch_from_ids(np.arange(len(rows)))
        finally:
            self.rows = saved

    def get_test_batch(self, batch):
        if getattr(self, '_index_of', None) is None:
            self._index_of = {id(e): i for i, e in enumerate(self.dataset._data)}
        return self.test_batch_from_ids([self._index_of[id(e)] for e in batch])

    def _batches(self):
        for ids in sampler_batches(len(self.dataset), self.batch_size, self.shuffle, self.drop_last):
            yield self.train_batch_from_ids(ids) if self._train else self.test_batch_from_ids(ids)

    def __iter__(self):
        """``prefetch`` > 0: the batches are built that many ahead by one producer thread on its own stream (the
        draw order stays sequential).  The paragraph-vector sub-batches are assembled lazily by the consumer."""
        if self.prefetch <= 0:
            return self._batches()
        return prefetch_iter(self._batches, self.prefetch, self.device)
