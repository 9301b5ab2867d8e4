#!/usr/bin/env python3
"""Golden batches of the REFERENCE's review-transformer data path on a synthetic gz corpus.

Run here (needs /root/reference):  python tests/golden/make_golden_rtm_loader.py
``prodsearch_amd.synth.write_corpus(seed)`` writes the files (data only; the tests regenerate them from the seed); the
reference's ``GlobalProdSearchData`` / ``ProdSearchData.initialize_epoch`` (data/data_util.py), ``ProdSearchDataset``
(data/prod_search_dataset.py) and ``ProdSearchDataLoader`` (data/prod_search_dataloader.py) run under
``random.seed`` / ``np.random.seed`` / ``torch.manual_seed``; every tensor of the first batches of two epochs, the
epoch's negative products / sub-sampled review table and a few evaluation batches go to tests/golden/rtmload_*.npz."""
import os
import random
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')

import numpy as np  # noqa: E402
import torch  # noqa: E402

from data.data_util import GlobalProdSearchData as RefGlobal, ProdSearchData as RefProd  # noqa: E402
from data.prod_search_dataset import ProdSearchDataset as RefDataset  # noqa: E402
from data.prod_search_dataloader import ProdSearchDataLoader as RefLoader  # noqa: E402
import others.util as ref_util  # noqa: E402  (reference)
from prodsearch_amd import default_args, synth  # noqa: E402
from prodsearch_amd.rtm_data import _TRAIN_FIELDS  # noqa: E402



class _IndexableList(list):
    """The paragraph-vector branch of the reference indexes four padded Python lists with an index array
    (prod_search_dataloader.py:342-343: ``pos_user_idxs[batch_indices[i]]`` ...) and raises TypeError as shipped — the
    user/item id lists were never converted like their neighbours (:335-336).  The harness hands the branch lists that
    accept array indices (the evident intent), nothing else changes, so the rest of it can be recorded."""

    def __getitem__(self, idx):
        if isinstance(idx, np.ndarray):
            return np.asarray(self)[idx]
        return list.__getitem__(self, idx)


_pad, _pad_3d = ref_util.pad, ref_util.pad_3d
ref_util.pad = lambda *a, **k: _IndexableList(_pad(*a, **k))
ref_util.pad_3d = lambda *a, **k: _IndexableList(_pad_3d(*a, **k))

BASE = dict(model_name='review_transformer', has_valid=True)
CASES = {
    # defaults of the reference: sub-sample then cut, no PV epochs, reviews by training-set membership
    'rtmload_a': (dict(seed=21), dict(review_encoder_name='pvc', uprev_review_limit=3, iprev_review_limit=4, neg_per_pos=3,
                                      review_word_limit=12, subsampling_rate=1e-3, valid_candi_size=7, candi_batch_size=4,
                                      test_candi_size=-1), dict(batch_size=8, pv_epochs=0, n_batches=4)),
    # paragraph-vector epochs: shuffled review rows, windows of 3 over 10 words (padded to 12), permuted sub-batches
    'rtmload_b': (dict(seed=22, n_users=25, n_products=30, n_words=120),
                  dict(review_encoder_name='pv', uprev_review_limit=2, iprev_review_limit=3, neg_per_pos=2,
                       review_word_limit=10, pv_window_size=3, subsampling_rate=1e-2, valid_candi_size=-1,
                       candi_batch_size=9, test_candi_size=-1, prod_freq_neg_sample=True),
                  dict(batch_size=6, pv_epochs=2, n_batches=2)),
    # sub-sampling MASKS (np.random.random per word), time-ordered review windows in training and evaluation
    'rtmload_c': (dict(seed=23), dict(review_encoder_name='pvc', uprev_review_limit=4, iprev_review_limit=5, neg_per_pos=4,
                                      review_word_limit=8, do_subsample_mask=True, subsampling_rate=5e-3,
                                      do_seq_review_train=True, do_seq_review_test=True, train_review_only=False,
                                      valid_candi_size=5, candi_batch_size=5, test_candi_size=-1, pv_window_size=2,
                                      shuffle_review_words=False),
                  dict(batch_size=10, pv_epochs=1, n_batches=2)),
}
TEST_FIELDS = ('query_word_idxs', 'candi_prod_ridxs', 'candi_seg_idxs', 'candi_seq_user_idxs', 'candi_seq_item_idxs')


def put_train(out, key, b):
    for f in _TRAIN_FIELDS:
        v = getattr(b, f, None)
        if v is not None:
            out['%s_%s' % (key, f)] = np.asarray(v)


def main():
    for name, (ckw, over, run) in CASES.items():
        ckw = dict(ckw)
        seed = ckw.pop('seed')
        args = default_args(**dict(BASE, **over))
        out = dict(corpus_seed=seed, corpus_kw=np.array(repr(ckw)), args_over=np.array(repr(dict(BASE, **over))),
                   run=np.array(repr(run)))
        with tempfile.TemporaryDirectory() as tmp:
            data_path, inp = synth.write_corpus(tmp, seed, **ckw)
            gd = RefGlobal(args, data_path, inp)
            train_pd = RefProd(args, inp, 'train', gd)
            valid_pd = RefProd(args, inp, 'valid', gd)
            test_pd = RefProd(args, inp, 'test', gd)
        random.seed(700 + seed)
        np.random.seed(800 + seed)
        torch.manual_seed(900 + seed)
        for ep in range(2):
            train_pd.initialize_epoch()
            out['ep%d_neg_sample_products' % ep] = np.asarray(train_pd.neg_sample_products)
            if not args.do_subsample_mask:
                out['ep%d_padded_review_words' % ep] = np.asarray(gd.padded_review_words)
            ds = RefDataset(args, gd, train_pd)
            prepare_pv = ep < run['pv_epochs']
            dl = RefLoader(args, ds, prepare_pv=prepare_pv, batch_size=run['batch_size'], shuffle=True, num_workers=0)
            n_seen = 0
            for i, b in enumerate(dl):
                if i == run['n_batches']:
                    break
                n_seen += 1
                key = 'ep%d_b%d' % (ep, i)
                if b is None:
                    out[key + '_none'] = np.int64(1)
                elif isinstance(b, list):
                    out[key + '_n'] = np.int64(len(b))
                    for j in sorted({0, 1, len(b) // 2, len(b) - 1}):
                        put_train(out, '%s_s%d' % (key, j), b[j])
                else:
                    put_train(out, key, b)
            out['ep%d_batches' % ep] = np.int64(n_seen)
        vds = RefDataset(args, gd, valid_pd)
        tds = RefDataset(args, gd, test_pd)
        for key, d in (('valid', vds), ('test', tds)):
            out[key + '_quad'] = np.asarray([e[:4] for e in d._data], dtype=np.int64)
            lens = [len(e[4]) for e in d._data]
            out[key + '_candi_ptr'] = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            out[key + '_candi_flat'] = np.asarray([v for e in d._data for v in e[4]], dtype=np.int64)
            dl = RefLoader(args, d, batch_size=5, shuffle=False, num_workers=0)
            for i, b in enumerate(dl):
                if i == 2:
                    break
                for f in TEST_FIELDS:
                    out['%s_b%d_%s' % (key, i, f)] = getattr(b, f).numpy()
                out['%s_b%d_candi_prod_idxs' % (key, i)] = np.asarray(b.candi_prod_idxs)
                out['%s_b%d_target_prod_idxs' % (key, i)] = np.asarray(b.target_prod_idxs)
                out['%s_b%d_query_idxs' % (key, i)] = np.asarray(b.query_idxs)
                out['%s_b%d_user_idxs' % (key, i)] = np.asarray(b.user_idxs)
        np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
        print(name, 'reviews', gd.review_count - 1, 'train rows', len(train_pd.review_info), 'valid entries', len(vds),
              'test entries', len(tds), '%.0f KB' % (os.path.getsize(os.path.join(HERE, name + '.npz')) / 1024))


if __name__ == '__main__':
    main()
