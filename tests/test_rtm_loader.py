"""Review-transformer data path (ProdSearchData.initialize_epoch, ProdSearchDataset, ProdSearchDataLoader over the
native collate) against batches the REFERENCE's own loader produced from the same synthetic gz corpus under the same
three seeds (tests/golden/rtmload_*.npz, make_golden_rtm_loader.py).  All index / mask work: bit-exact."""
import ast
import ctypes as C
import glob
import os

import numpy as np
import pytest
import torch

from prodsearch_amd import _lib, default_args, pyrandom, synth
from prodsearch_amd.corpus import GlobalProdSearchData, ProdSearchData, ProdSearchDataset
from prodsearch_amd.rtm_data import _TRAIN_FIELDS
from prodsearch_amd.rtm_loader import ProdSearchDataLoader

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLD, 'rtmload_*.npz')))
TEST_FIELDS = ('query_word_idxs', 'candi_prod_ridxs', 'candi_seg_idxs', 'candi_seq_user_idxs', 'candi_seq_item_idxs')


def _same_train(z, key, b):
    for f in _TRAIN_FIELDS:
        k = '%s_%s' % (key, f)
        got = getattr(b, f, None)
        if k not in z.files:
            assert got is None, (key, f)
            continue
        assert got is not None, (key, f)
        got = got.numpy()
        assert got.shape == z[k].shape, (key, f, got.shape, z[k].shape)
        assert np.array_equal(got, z[k]), (key, f)


def test_cases_present():
    assert len(CASES) >= 3


@pytest.mark.parametrize('case', CASES)
def test_loader_matches_reference(case, tmp_path):
    z = np.load(os.path.join(GOLD, case + '.npz'))
    ckw, over, run = (ast.literal_eval(str(z[k])) for k in ('corpus_kw', 'args_over', 'run'))
    seed = int(z['corpus_seed'])
    args = default_args(**over)
    data_path, inp = synth.write_corpus(str(tmp_path), seed, **ckw)
    gd = GlobalProdSearchData(args, data_path, inp)
    pds = {s: ProdSearchData(args, inp, s, gd) for s in ('train', 'valid', 'test')}
    pyrandom.seed(700 + seed)
    np.random.seed(800 + seed)
    torch.manual_seed(900 + seed)
    for ep in range(2):
        pds['train'].initialize_epoch()
        assert np.array_equal(pds['train'].neg_sample_products, z['ep%d_neg_sample_products' % ep])
        if not args.do_subsample_mask:
            assert np.array_equal(gd.padded_review_words, z['ep%d_padded_review_words' % ep])
        ds = ProdSearchDataset(args, gd, pds['train'])
        dl = ProdSearchDataLoader(args, ds, prepare_pv=ep < run['pv_epochs'], batch_size=run['batch_size'], shuffle=True)
        seen = 0
        for i, b in enumerate(dl):
            if i == run['n_batches']:
                break
            seen += 1
            key = 'ep%d_b%d' % (ep, i)
            if key + '_none' in z.files:
                assert b is None
            elif key + '_n' in z.files:
                assert len(b) == int(z[key + '_n'])
                for j in sorted({0, 1, len(b) // 2, len(b) - 1}):
                    _same_train(z, '%s_s%d' % (key, j), b[j])
            else:
                _same_train(z, key, b)
        assert seen == int(z['ep%d_batches' % ep])
    for key in ('valid', 'test'):
        d = ProdSearchDataset(args, gd, pds[key])
        assert np.array_equal(np.asarray([e[:4] for e in d._data], dtype=np.int64), z[key + '_quad'])
        assert np.array_equal(np.asarray([v for e in d._data for v in e[4]], dtype=np.int64), z[key + '_candi_flat'])
        dl = ProdSearchDataLoader(args, d, batch_size=5, shuffle=False)
        for i, b in enumerate(dl):
            if i == 2:
                break
            for f in TEST_FIELDS:
                ref = z['%s_b%d_%s' % (key, i, f)]
                got = getattr(b, f).numpy()
                assert got.shape == ref.shape and np.array_equal(got, ref), (key, i, f)
            assert np.array_equal(np.asarray(b.candi_prod_idxs), z['%s_b%d_candi_prod_idxs' % (key, i)])
            assert np.array_equal(np.asarray(b.target_prod_idxs), z['%s_b%d_target_prod_idxs' % (key, i)])
            assert np.array_equal(np.asarray(b.query_idxs), z['%s_b%d_query_idxs' % (key, i)])
            assert np.array_equal(np.asarray(b.user_idxs), z['%s_b%d_user_idxs' % (key, i)])


def test_numpy_generator_clone():
    """The native MT19937 follows numpy's legacy global generator draw for draw (shuffle / permutation / random)."""
    lib = _lib.load_data()
    h = lib.ps_rng_create(0)
    for seed in (0, 5, 12345):
        np.random.seed(seed)
        st = np.random.get_state()
        lib.ps_rng_set_state(h, np.ascontiguousarray(st[1], dtype=np.uint32).ctypes.data, int(st[2]))
        for n in (1, 2, 3, 7, 100, 1000, 70000):
            ref = np.random.permutation(n)
            mine = np.arange(n)
            for i in range(n - 1, 0, -1):
                j = lib.ps_rng_np_interval(h, i)
                mine[i], mine[j] = mine[j], mine[i]
            assert np.array_equal(ref, mine), (seed, n)
        ref = np.random.random(1000)
        assert np.array_equal(ref, [lib.ps_rng_random(h) for _ in range(1000)])
        key, pos = np.empty(624, dtype=np.uint32), C.c_int32(0)
        lib.ps_rng_get_state(h, key.ctypes.data, C.byref(pos))
        now = np.random.get_state()
        assert np.array_equal(key, now[1]) and pos.value == now[2]
    lib.ps_rng_destroy(h)


def test_bad_arguments_are_reported():
    lib = _lib.load_data()
    v, a = _lib.PsRtmCorpusView(), _lib.PsRtmCollateArgs()
    dims = (C.c_int32 * 4)()
    assert lib.ps_rtm_collate_train(C.byref(v), C.byref(a), None, None, 1, None, 0, *([None] * 10), dims) != 0
    assert b'rtm collate' in lib.ps_data_last_error()
