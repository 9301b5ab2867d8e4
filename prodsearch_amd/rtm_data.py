"""Batch containers and synthetic batches of the RTM (review_transformer) hot path.

``ProdSearchTrainBatch`` / ``ProdSearchTestBatch`` mirror the reference's field bags
(``data/batch_data.py:137-223`` and ``:94-135``): int64 index tensors (+ uint8 word masks),
row-major contiguous, ``.to(device)`` returning a new object.  The reference's own batch
objects are accepted unchanged by the model (duck typing).

    query_word_idxs          [B,Q]        pad V-1
    pos_prod_ridxs           [B,R]        pad review_count-1   (user reviews then item reviews)
    pos_seg_idxs             [B,R+1]      0 query, 1 user review, 2 item review, 3 pad
    pos_prod_rword_idxs      [B,R,W]      window words of the PV loss (train_pv) / review words [B,R,WL] otherwise
    pos_prod_rword_masks     [B,R,W]      uint8, 1 = valid target word
    neg_prod_ridxs           [B,K,R]      neg_seg_idxs [B,K,R+1]
    pos/neg_prod_rword_idxs_pvc  [B,R,WL] / [B,K,R,WL]   review words of the pvc encoder (train_pv)
    neg_prod_rword_idxs      [B,K,R,WL]   review words of the negatives when not train_pv (pvc)
"""
import numpy as np
import torch

from .synth import make_word_dists, rng_for

_TRAIN_FIELDS = ('query_word_idxs', 'pos_prod_ridxs', 'pos_seg_idxs', 'pos_prod_rword_idxs',
                 'pos_prod_rword_masks', 'neg_prod_ridxs', 'neg_seg_idxs', 'pos_user_idxs', 'neg_user_idxs',
                 'pos_item_idxs', 'neg_item_idxs', 'neg_prod_rword_idxs', 'neg_prod_rword_masks',
                 'pos_prod_rword_idxs_pvc', 'neg_prod_rword_idxs_pvc')


def _t(x, dtype=torch.int64):
    if x is None or torch.is_tensor(x):
        return x
    return torch.as_tensor(np.asarray(x), dtype=dtype)


class ProdSearchTrainBatch(object):
    def __init__(self, query_word_idxs, pos_prod_ridxs, pos_seg_idxs, pos_prod_rword_idxs, pos_prod_rword_masks,
                 neg_prod_ridxs, neg_seg_idxs, pos_user_idxs=None, neg_user_idxs=None, pos_item_idxs=None,
                 neg_item_idxs=None, neg_prod_rword_idxs=None, neg_prod_rword_masks=None,
                 pos_prod_rword_idxs_pvc=None, neg_prod_rword_idxs_pvc=None, to_tensor=True):
        vals = (query_word_idxs, pos_prod_ridxs, pos_seg_idxs, pos_prod_rword_idxs, pos_prod_rword_masks,
                neg_prod_ridxs, neg_seg_idxs, pos_user_idxs, neg_user_idxs, pos_item_idxs, neg_item_idxs,
                neg_prod_rword_idxs, neg_prod_rword_masks, pos_prod_rword_idxs_pvc, neg_prod_rword_idxs_pvc)
        for k, v in zip(_TRAIN_FIELDS, vals):
            if to_tensor:
                v = _t(v, torch.uint8 if k.endswith('_masks') else torch.int64)
            setattr(self, k, v)

    def to(self, device):
        if device == "cpu":
            return self
        mv = lambda x: None if x is None else x.to(device, non_blocking=True)
        return self.__class__(*[mv(getattr(self, k)) for k in _TRAIN_FIELDS], to_tensor=False)


class ProdSearchTestBatch(object):
    def __init__(self, query_idxs, user_idxs, target_prod_idxs, candi_prod_idxs, query_word_idxs,
                 candi_prod_ridxs, candi_seg_idxs, candi_seq_user_idxs=None, candi_seq_item_idxs=None,
                 to_tensor=True):
        self.query_idxs, self.user_idxs = query_idxs, user_idxs
        self.target_prod_idxs, self.candi_prod_idxs = target_prod_idxs, candi_prod_idxs
        conv = _t if to_tensor else (lambda x: x)
        self.query_word_idxs = conv(query_word_idxs)
        self.candi_prod_ridxs = conv(candi_prod_ridxs)
        self.candi_seg_idxs = conv(candi_seg_idxs)
        self.candi_seq_user_idxs = conv(candi_seq_user_idxs)
        self.candi_seq_item_idxs = conv(candi_seq_item_idxs)

    def to(self, device):
        if device == "cpu":
            return self
        mv = lambda x: None if x is None else x.to(device, non_blocking=True)
        return self.__class__(self.query_idxs, self.user_idxs, self.target_prod_idxs, self.candi_prod_idxs,
                              mv(self.query_word_idxs), mv(self.candi_prod_ridxs), mv(self.candi_seg_idxs),
                              mv(self.candi_seq_user_idxs), mv(self.candi_seq_item_idxs), to_tensor=False)


def make_review_words(seed, review_count, vocab_size, word_limit, word_dists=None):
    """[review_count, WL] padded review texts; the last review is the pad review (all pad words),
    as ``ProdSearchData.sub_sampling`` + ``pad`` build it (data/data_util.py:138-153)."""
    rng = rng_for(seed)
    wd = make_word_dists(vocab_size) if word_dists is None else word_dists
    rw = np.full((review_count, word_limit), vocab_size - 1, dtype=np.int64)
    lens = rng.integers(max(2, word_limit // 4), word_limit + 1, size=review_count - 1)
    words = rng.choice(vocab_size, size=(review_count - 1, word_limit), p=wd)
    for i in range(review_count - 1):
        rw[i, :lens[i]] = words[i, :lens[i]]
    return torch.from_numpy(rw)


def _seq(rng, n_seq, R, u_lim, i_lim, review_count, empty_frac=0.0):
    """review ids [n_seq,R] (user reviews, then item reviews, then pad) and seg ids [n_seq,R+1]."""
    pad = review_count - 1
    rid = np.full((n_seq, R), pad, dtype=np.int64)
    seg = np.full((n_seq, R + 1), 3, dtype=np.int64)
    seg[:, 0] = 0
    for n in range(n_seq):
        nu = int(min(u_lim, rng.geometric(0.2)))
        ni = int(min(i_lim, rng.geometric(0.12)))
        if rng.random() < empty_frac:
            nu = ni = 0
        nu = min(nu, R)
        ni = min(ni, R - nu)
        rid[n, :nu + ni] = rng.integers(0, pad, size=nu + ni)
        seg[n, 1:1 + nu] = 1
        seg[n, 1 + nu:1 + nu + ni] = 2
    return rid, seg


def _seq_ids(seed, ridxs, pad_rev, size):
    """Per-position ids [.., R+1] for a review-index tensor [.., R]: position 0 (the query) and padded reviews
    carry the pad id ``size`` (prod_search_dataloader.py:212-215); a private stream, so adding them leaves the
    other draws of a seed unchanged."""
    rng = rng_for(seed)
    ids = rng.integers(0, size, size=ridxs.shape)
    ids = np.where(ridxs == pad_rev, size, ids)
    ids = np.where(rng.random(ridxs.shape) < 0.03, size, ids)        # a pad id on a live position is legal too
    lead = np.full(ridxs.shape[:-1] + (1,), size, dtype=np.int64)
    return np.concatenate([lead, ids.astype(np.int64)], axis=-1)


def make_rtm_batch(seed, B, K, review_count, vocab_size, review_words, Q=8, u_lim=4, i_lim=6, W=1, train_pv=True,
                   encoder='pv', word_dists=None, user_size=None, product_size=None):
    """One RTM training batch (CPU tensors).  R = u_lim + i_lim; a few negatives have NO reviews
    (their loss weight is 0, ps_model.py:344-345)."""
    rng = rng_for(seed)
    V, R = vocab_size, u_lim + i_lim
    wd = make_word_dists(V) if word_dists is None else word_dists
    qlen = rng.integers(2, min(6, Q) + 1, size=B)
    qw = np.full((B, Q), V - 1, dtype=np.int64)
    words = rng.choice(V, size=(B, Q), p=wd)
    for b in range(B):
        qw[b, :qlen[b]] = words[b, :qlen[b]]
    pos_r, pos_seg = _seq(rng, B, R, u_lim, i_lim, review_count)
    # every positive sequence has at least one review (ps_model.py:278 comment)
    for b in range(B):
        if (pos_r[b] == review_count - 1).all():
            pos_r[b, 0] = rng.integers(0, review_count - 1)
            pos_seg[b, 1] = 2
    neg_r, neg_seg = _seq(rng, B * K, R, u_lim, i_lim, review_count, empty_frac=0.1)
    neg_r, neg_seg = neg_r.reshape(B, K, R), neg_seg.reshape(B, K, R + 1)
    rw = review_words.numpy()
    WL = rw.shape[1]
    pad_rev = review_count - 1
    kw = {}
    if train_pv:
        tw = np.full((B, R, W), V - 1, dtype=np.int64)
        tm = np.zeros((B, R, W), dtype=np.uint8)
        for b in range(B):
            for r in range(R):
                if pos_r[b, r] != pad_rev:
                    toks = rw[pos_r[b, r]]
                    toks = toks[toks != V - 1]
                    n = min(W, len(toks))
                    tw[b, r, :n] = rng.choice(toks, size=n, replace=False) if n else []
                    tm[b, r, :n] = 1
        pos_words, pos_masks = tw, tm
        if encoder == 'pvc':
            kw['pos_prod_rword_idxs_pvc'] = rw[pos_r]
            kw['neg_prod_rword_idxs_pvc'] = rw[neg_r]
    else:
        pos_words = rw[pos_r] if encoder in ('pvc', 'fs', 'avg') else np.full((B, R, W), V - 1, dtype=np.int64)
        pos_masks = (pos_words != V - 1).astype(np.uint8)
        if encoder in ('pvc', 'fs', 'avg'):
            kw['neg_prod_rword_idxs'] = rw[neg_r]
            kw['neg_prod_rword_masks'] = (rw[neg_r] != V - 1).astype(np.uint8)
    pos_u = neg_u = pos_i = neg_i = None
    if user_size is None:
        pos_u, neg_u = np.zeros_like(pos_seg), np.zeros_like(neg_seg)
    else:
        pos_u, neg_u = _seq_ids(seed + 71, pos_r, pad_rev, user_size), _seq_ids(seed + 72, neg_r, pad_rev, user_size)
    if product_size is None:
        pos_i, neg_i = np.zeros_like(pos_seg), np.zeros_like(neg_seg)
    else:
        pos_i, neg_i = (_seq_ids(seed + 73, pos_r, pad_rev, product_size),
                        _seq_ids(seed + 74, neg_r, pad_rev, product_size))
    return ProdSearchTrainBatch(qw, pos_r, pos_seg, pos_words, pos_masks, neg_r, neg_seg,
                                pos_u, neg_u, pos_i, neg_i, **kw)


def make_rtm_test_batch(seed, B, C, review_count, vocab_size, Q=8, u_lim=4, i_lim=6, word_dists=None,
                        user_size=None, product_size=None):
    rng = rng_for(seed)
    V, R = vocab_size, u_lim + i_lim
    wd = make_word_dists(V) if word_dists is None else word_dists
    qw = np.full((B, Q), V - 1, dtype=np.int64)
    qlen = rng.integers(2, min(6, Q) + 1, size=B)
    words = rng.choice(V, size=(B, Q), p=wd)
    for b in range(B):
        qw[b, :qlen[b]] = words[b, :qlen[b]]
    cr, cs = _seq(rng, B * C, R, u_lim, i_lim, review_count, empty_frac=0.05)
    cr, cs = cr.reshape(B, C, R), cs.reshape(B, C, R + 1)
    z = np.zeros_like(cs)
    candi = rng.integers(0, 1000, size=(B, C))
    su = z if user_size is None else _seq_ids(seed + 75, cr, review_count - 1, user_size)
    si = z if product_size is None else _seq_ids(seed + 76, cr, review_count - 1, product_size)
    return ProdSearchTestBatch(list(range(B)), list(range(B)), candi[:, 0].copy(), candi, qw, cr, cs, su, si)
