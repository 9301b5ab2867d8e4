#!/usr/bin/env python3
"""Golden batches of the REFERENCE's TEM dataloader (data/item_pv_dataloader.py) on a synthetic corpus.

Run here (needs /root/reference):  python tests/golden/make_golden_collate.py
Writes tests/golden/collate_*.npz: for each case the batches the reference's ``ItemPVDataloader`` yields
(``DataLoader(shuffle=..., num_workers=0)`` iteration, so sampler order AND the ``random.choice`` /
``random.sample`` draws are the reference's) plus the seeds that regenerate the corpus with
``prodsearch_amd.synth.make_corpus``.  Data only — no reference source is stored."""
import os
import random
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')

import numpy as np  # noqa: E402
import torch  # noqa: E402

from data.item_pv_dataloader import ItemPVDataloader as RefLoader  # noqa: E402  (the reference)
from prodsearch_amd import default_args, synth  # noqa: E402

CASES = {
    # name: corpus kwargs, arg overrides, batch size, shuffle, batches kept
    'collate_fix':    (dict(seed=1), dict(uprev_review_limit=20, fix_train_review=True), 48, True, 4),
    'collate_rand20': (dict(seed=2), dict(uprev_review_limit=20, fix_train_review=False), 64, True, 4),   # pool AND set sample()
    'collate_rand5':  (dict(seed=3), dict(uprev_review_limit=5, fix_train_review=False), 32, True, 4),    # k <= 5: setsize 21
    'collate_seq':    (dict(seed=4), dict(uprev_review_limit=10, do_seq_review_train=True), 40, False, 3),
    'collate_short':  (dict(seed=6, max_reviews_per_user=7, n_users=150), dict(uprev_review_limit=20, fix_train_review=False), 16, True, 5),
    'collate_w3':     (dict(seed=5, W=3, Q=9), dict(uprev_review_limit=7, fix_train_review=False, pv_window_size=3), 25, True, 3),
}


def main():
    for name, (ckw, over, B, shuffle, keep) in CASES.items():
        ckw = dict(ckw)
        seed = ckw.pop('seed')
        train_ds, test_ds = synth.make_corpus(seed, **ckw)
        args = default_args(**over)
        out = dict(corpus_seed=seed, batch_size=B, shuffle=int(shuffle), py_seed=1000 + seed, torch_seed=2000 + seed)
        out['corpus_kw'] = np.array(repr(ckw))
        out['args_over'] = np.array(repr(over))
        random.seed(1000 + seed)
        torch.manual_seed(2000 + seed)
        dl = RefLoader(args, train_ds, batch_size=B, shuffle=shuffle, num_workers=0)
        n = 0
        for b in dl:
            out['train%d_query_word_idxs' % n] = b.query_word_idxs.numpy()
            out['train%d_target_prod_idxs' % n] = b.target_prod_idxs.numpy()
            out['train%d_u_item_idxs' % n] = b.u_item_idxs.numpy().astype(np.int64).reshape(len(b.target_prod_idxs), -1)
            out['train%d_pos_iword_idxs' % n] = b.pos_iword_idxs.numpy()
            n += 1
            if n == keep:
                break
        out['n_train'] = n
        for do_seq in (0, 1):
            targs = default_args(**dict(over, do_seq_review_test=bool(do_seq), train_review_only=not do_seq))
            tl = RefLoader(targs, test_ds, batch_size=9, shuffle=False, num_workers=0)
            for i, b in enumerate(tl):
                if i == 2:
                    break
                pre = 'test%d_seq%d_' % (i, do_seq)
                out[pre + 'query_word_idxs'] = b.query_word_idxs.numpy()
                out[pre + 'target_prod_idxs'] = b.target_prod_idxs.numpy()
                out[pre + 'u_item_idxs'] = b.u_item_idxs.numpy().astype(np.int64).reshape(len(b.target_prod_idxs), -1)
                out[pre + 'candi_prod_idxs'] = b.candi_prod_idxs.numpy()
                out[pre + 'query_idxs'] = np.asarray(b.query_idxs)
                out[pre + 'user_idxs'] = np.asarray(b.user_idxs)
        np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
        print(name, 'train batches', n, 'L widths', [out['train%d_u_item_idxs' % i].shape[1] for i in range(n)])


if __name__ == '__main__':
    main()
