"""Batch builder: oracle and C++ product against batches produced by the REFERENCE's
ItemPVDataloader (tests/golden/collate_*.npz, made by tests/golden/make_golden_collate.py).  Index work: bit-exact.
Host-only code, so everything here runs without a GPU."""
import ast
import glob
import os
import random
import re

import numpy as np
import pytest
import torch
from torch.utils.data import BatchSampler, RandomSampler, SequentialSampler

from oracle import collate as ocollate
from prodsearch_amd import _lib, default_args, synth
from prodsearch_amd.dataloader import ItemPVDataloader

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLD, 'collate_*.npz')))


def _case(name):
    z = np.load(os.path.join(GOLD, name + '.npz'))
    ckw = ast.literal_eval(str(z['corpus_kw']))
    over = ast.literal_eval(str(z['args_over']))
    train_ds, test_ds = synth.make_corpus(int(z['corpus_seed']), **ckw)
    return z, over, train_ds, test_ds


def test_cases_present():
    assert len(CASES) >= 6


def test_header_symbols_exported():
    lib = _lib.load_data()
    hdr = open(os.path.join(os.path.dirname(GOLD), '..', 'include', 'prodsearch_data.h')).read()
    declared = set(re.findall(r'\b(ps_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_lib.DATA_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)


def test_mersenne_twister_is_cpythons():
    lib = _lib.load_data()
    for seed in (0, 1, 666, 2 ** 31, 2 ** 32 + 5, 123456789012345):
        r = lib.ps_rng_create(seed)
        random.seed(seed)
        for n in (1, 2, 3, 5, 21, 85, 86, 1000, 2 ** 31 - 1, 2 ** 31, 2 ** 32 - 1):
            assert lib.ps_rng_randbelow(r, n) == random._inst._randbelow(n)
        assert lib.ps_rng_random(r) == random.random()
        lib.ps_rng_destroy(r)


def _sampler(ds, B, shuffle):
    torch.empty((), dtype=torch.int64).random_()          # DataLoader iterator's base seed comes first
    return BatchSampler(RandomSampler(ds) if shuffle else SequentialSampler(ds), B, False)


@pytest.mark.parametrize('case', CASES)
def test_oracle_matches_reference_batches(case):
    z, over, train_ds, test_ds = _case(case)
    args = default_args(**over)
    random.seed(int(z['py_seed']))
    torch.manual_seed(int(z['torch_seed']))
    for n, ids in zip(range(int(z['n_train'])), _sampler(train_ds, int(z['batch_size']), bool(z['shuffle']))):
        got = ocollate.train_batch(train_ds, args, [train_ds[i] for i in ids])
        for k in ('query_word_idxs', 'target_prod_idxs', 'u_item_idxs', 'pos_iword_idxs'):
            want = z['train%d_%s' % (n, k)]
            assert np.array_equal(np.asarray(got[k], dtype=np.int64).reshape(want.shape), want), (n, k)
    for do_seq in (0, 1):
        targs = default_args(**dict(over, do_seq_review_test=bool(do_seq), train_review_only=not do_seq))
        for i, ids in zip(range(2), BatchSampler(SequentialSampler(test_ds), 9, False)):
            got = ocollate.test_batch(test_ds, targs, [test_ds[j] for j in ids])
            for k in ('query_word_idxs', 'target_prod_idxs', 'u_item_idxs', 'candi_prod_idxs', 'query_idxs', 'user_idxs'):
                want = z['test%d_seq%d_%s' % (i, do_seq, k)]
                assert np.array_equal(np.asarray(got[k], dtype=np.int64).reshape(want.shape), want), (i, do_seq, k)


@pytest.mark.parametrize('case', CASES)
def test_native_loader_matches_reference_batches(case):
    """The C++ collate behind the ItemPVDataloader mirror, iterated like the reference's DataLoader."""
    z, over, train_ds, test_ds = _case(case)
    args = default_args(**over)
    dl = ItemPVDataloader(args, train_ds, batch_size=int(z['batch_size']), shuffle=bool(z['shuffle']),
                          seed=int(z['py_seed']))
    torch.manual_seed(int(z['torch_seed']))
    n = 0
    for b in dl:
        for k in ('query_word_idxs', 'target_prod_idxs', 'u_item_idxs', 'pos_iword_idxs'):
            want = z['train%d_%s' % (n, k)]
            got = getattr(b, k)
            assert got.dtype == torch.int64 and got.is_contiguous()
            assert np.array_equal(got.numpy().reshape(want.shape), want), (n, k)
        n += 1
        if n == int(z['n_train']):
            break
    assert n == int(z['n_train'])
    for do_seq in (0, 1):
        targs = default_args(**dict(over, do_seq_review_test=bool(do_seq), train_review_only=not do_seq))
        tl = ItemPVDataloader(targs, test_ds, batch_size=9, shuffle=False)
        for i, b in zip(range(2), tl):
            for k in ('query_word_idxs', 'target_prod_idxs', 'u_item_idxs', 'candi_prod_idxs'):
                want = z['test%d_seq%d_%s' % (i, do_seq, k)]
                assert np.array_equal(getattr(b, k).numpy().reshape(want.shape), want), (i, do_seq, k)
            assert b.query_idxs == z['test%d_seq%d_query_idxs' % (i, do_seq)].tolist()
            assert b.user_idxs == z['test%d_seq%d_user_idxs' % (i, do_seq)].tolist()


def test_collate_entry_lists_like_the_reference_collate_fn():
    """get_train_batch(batch) takes the dataset entries themselves, as DataLoader hands them to collate_fn."""
    z, over, train_ds, _ = _case('collate_rand20')
    args = default_args(**over)
    dl = ItemPVDataloader(args, train_ds, batch_size=8, seed=5)
    entries = [train_ds[i] for i in (3, 1, 4, 1, 5)]
    random.seed(5)
    want = ocollate.train_batch(train_ds, args, entries)
    got = dl.get_train_batch(entries)
    assert np.array_equal(got.u_item_idxs.numpy(), np.asarray(want['u_item_idxs']))
    assert np.array_equal(got.query_word_idxs.numpy(), np.asarray(want['query_word_idxs']))


def test_bad_ids_raise_instead_of_faulting():
    _, over, train_ds, _ = _case('collate_fix')
    dl = ItemPVDataloader(default_args(**over), train_ds, batch_size=4, seed=1)
    with pytest.raises(RuntimeError, match='out of range'):
        dl.train_batch_from_ids([0, len(train_ds) + 7])
    dl.sample_review[2] = 10 ** 9
    with pytest.raises(RuntimeError, match='review id'):
        dl.train_batch_from_ids([2])


@pytest.mark.parametrize('seed', range(24))
def test_native_collate_equals_oracle_on_random_corpora(seed):
    """Randomised corpora and flags (history limits 1..25, fixed / random-subset / sequential histories, pv windows,
    users with no training review at all): the C++ collate must equal the Python restatement bit for bit, with both
    consuming the same Mersenne-Twister stream."""
    rng = np.random.default_rng(1000 + seed)
    limit = int(rng.integers(1, 26))
    over = dict(uprev_review_limit=limit, fix_train_review=bool(rng.integers(0, 2)),
                do_seq_review_train=bool(rng.integers(0, 4) == 0), pv_window_size=int(rng.integers(1, 4)))
    train_ds, test_ds = synth.make_corpus(500 + seed, n_users=int(rng.integers(5, 80)), n_products=int(rng.integers(3, 60)),
                                          n_queries=int(rng.integers(1, 30)), vocab_size=int(rng.integers(20, 400)),
                                          Q=int(rng.integers(1, 9)), W=over['pv_window_size'],
                                          max_reviews_per_user=int(rng.integers(2, 130)),
                                          train_frac=float(rng.choice([0.3, 0.8, 1.0])))
    args = default_args(**over)
    B = int(rng.integers(1, 70))
    dl = ItemPVDataloader(args, train_ds, batch_size=B, shuffle=True, seed=77 + seed)
    random.seed(77 + seed)
    torch.manual_seed(seed)
    ids_per_batch = []
    for n, b in enumerate(dl):
        ids_per_batch.append(b)
        if n == 3:
            break
    torch.manual_seed(seed)
    for n, ids in zip(range(len(ids_per_batch)), _sampler(train_ds, B, True)):
        want = ocollate.train_batch(train_ds, args, [train_ds[i] for i in ids])
        got = ids_per_batch[n]
        for k in ('query_word_idxs', 'target_prod_idxs', 'u_item_idxs', 'pos_iword_idxs'):
            w = np.asarray(want[k], dtype=np.int64).reshape(len(ids), -1)
            g = getattr(got, k).numpy().reshape(len(ids), -1)
            if k == 'u_item_idxs' and w.shape[1] == 0:
                assert g.shape[1] == 0 or (g == train_ds.prod_pad_idx).all()
                continue
            assert np.array_equal(g, w), (seed, n, k)
    # evaluation entries
    targs = default_args(**dict(over, do_seq_review_test=bool(seed % 2), train_review_only=not bool(seed % 2)))
    if len(test_ds) > 0:
        tl = ItemPVDataloader(targs, test_ds, batch_size=5, shuffle=False)
        for i, (b, ids) in enumerate(zip(tl, BatchSampler(SequentialSampler(test_ds), 5, False))):
            want = ocollate.test_batch(test_ds, targs, [test_ds[j] for j in ids])
            for k in ('query_word_idxs', 'target_prod_idxs', 'candi_prod_idxs'):
                assert np.array_equal(getattr(b, k).numpy(), np.asarray(want[k])), (seed, i, k)
            w = np.asarray(want['u_item_idxs'], dtype=np.int64).reshape(len(ids), -1)
            if w.shape[1]:
                assert np.array_equal(b.u_item_idxs.numpy(), w), (seed, i)
            if i == 2:
                break


@pytest.mark.parametrize('case', CASES)
def test_native_producer_thread_matches_reference_batches(case):
    """prefetch > 0: the epoch's batches come from the native producer thread (ps_epoch_start) — the reference's batches still."""
    z, over, train_ds, _ = _case(case)
    dl = ItemPVDataloader(default_args(**over), train_ds, batch_size=int(z['batch_size']), shuffle=bool(z['shuffle']),
                          seed=int(z['py_seed']), prefetch=2)
    torch.manual_seed(int(z['torch_seed']))
    n = 0
    for b in dl:
        for k in ('query_word_idxs', 'target_prod_idxs', 'u_item_idxs', 'pos_iword_idxs'):
            want = z['train%d_%s' % (n, k)]
            got = getattr(b, k)
            assert got.dtype == torch.int64 and got.is_contiguous()
            assert np.array_equal(got.numpy().reshape(want.shape), want), (n, k)
        n += 1
        if n == int(z['n_train']):
            break
    assert n == int(z['n_train'])


@pytest.mark.parametrize('prefetch,drop_last,B', [(1, False, 9), (3, False, 16), (2, True, 16), (5, False, 1000)])
def test_native_producer_thread_equals_the_sequential_loop_over_whole_epochs(prefetch, drop_last, B):
    """Two epochs, ragged last batch / drop_last / one batch larger than the dataset: every tensor, the per-row ids and the
    generator state afterwards equal the sequential loader's; batches handed out earlier are not overwritten by later ones."""
    train_ds, _ = synth.make_corpus(31, n_users=40, n_products=30, n_queries=12, vocab_size=90, Q=4, W=2, max_reviews_per_user=60)
    args = default_args(uprev_review_limit=6, fix_train_review=False, pv_window_size=2)
    seq = ItemPVDataloader(args, train_ds, batch_size=B, shuffle=True, seed=9, drop_last=drop_last)
    thr = ItemPVDataloader(args, train_ds, batch_size=B, shuffle=True, seed=9, drop_last=drop_last, prefetch=prefetch)
    assert len(train_ds) % B != 0
    for epoch in range(2):
        torch.manual_seed(100 + epoch)
        want = list(seq)
        torch.manual_seed(100 + epoch)
        got = list(thr)                                  # ALL batches kept alive: a slot reused too early would show here
        assert len(got) == len(want) == (len(train_ds) // B if drop_last else -(-len(train_ds) // B))
        for w, g in zip(want, got):
            for k in ('query_word_idxs', 'target_prod_idxs', 'u_item_idxs', 'pos_iword_idxs'):
                assert torch.equal(getattr(w, k), getattr(g, k)), k
            assert list(w.query_idxs) == list(g.query_idxs) and list(w.user_idxs) == list(g.user_idxs)
    lib = _lib.load_data()
    assert lib.ps_rng_randbelow(seq._rng, 1 << 30) == lib.ps_rng_randbelow(thr._rng, 1 << 30)


def test_native_producer_thread_surfaces_collate_errors_and_stops_early():
    train_ds, _ = synth.make_corpus(32, n_users=30, n_products=20, n_queries=8, vocab_size=70, Q=3, W=1, max_reviews_per_user=20)
    args = default_args(uprev_review_limit=5, fix_train_review=True)
    dl = ItemPVDataloader(args, train_ds, batch_size=8, shuffle=False, seed=1, prefetch=2)
    it = iter(dl)
    next(it)
    it.close()                                           # consumer leaves early: the thread is stopped and joined
    dl.sample_review[20] = 10 ** 9                       # third batch is corrupt
    it = iter(dl)
    next(it), next(it)
    with pytest.raises(RuntimeError, match='review id'):
        next(it)
