"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the
header declares, the nn.Module boundary has the reference's parameter names/shapes, the
product refuses to run without a GPU (no CPU fallback), and host-side helpers (alias table,
workspace layout, the dropout stream) behave.  No GPU compute is called here."""
import os
import re

import numpy as np
import pytest
import torch

from prodsearch_amd import ItemTransformerRanker, Optimizer, build_optim, default_args, synth, _lib
from prodsearch_amd import build as pbuild

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    pbuild.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(REPO, 'include', 'prodsearch_hip.h')).read()
    declared = set(re.findall(r'\b(ps_[a-z0-9_]+)\s*\(', hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert b'gfx950' in lib.ps_version()


def test_struct_layout_matches_header_sizes(lib):
    import ctypes as C
    assert C.sizeof(_lib.PsTemDesc) == 112
    assert C.sizeof(_lib.PsLayerTensors) == 16 * 8
    assert C.sizeof(_lib.PsTemTensors) == 10 * 8 + _lib.PS_MAX_LAYERS * 16 * 8
    assert C.sizeof(_lib.PsTemBatch) == 7 * 8
    assert C.sizeof(_lib.PsAdamHyper) == 48


@pytest.mark.parametrize('over', [dict(), dict(inter_layers=2, sep_prod_emb=True),
                                  dict(model_name='QEM'), dict(query_encoder_name='avg')])
def test_module_has_reference_state_dict(over):
    a = default_args(**dict(dict(model_name='item_transformer', inter_layers=1, embedding_size=32,
                                 ff_size=64, heads=4), **over))
    m = ItemTransformerRanker(a, 'cpu', 300, 200, None)
    want = synth.tem_param_shapes(a, 300, 200)      # asserted equal to the reference's in make_golden.py
    got = {k: tuple(v.shape) for k, v in m.state_dict().items() if not k.endswith('pos_emb.pe')}
    assert list(got) == list(want) and got == want
    if a.model_name == 'item_transformer':
        assert tuple(m.state_dict()['transformer_encoder.pos_emb.pe'].shape) == (1, 5000, 32)
    assert all(v.dtype == torch.float32 for v in m.state_dict().values())
    # reference init (row I of SURVEY §8a): product pad row zero, word pad row overwritten
    assert float(m.product_emb.weight[200].abs().max()) == 0.0
    assert float(m.word_embeddings.weight[299].abs().max()) > 0.0


def test_positional_table_matches_oracle():
    from oracle.tem import positional_encoding
    a = default_args(model_name='item_transformer', inter_layers=1, embedding_size=32, ff_size=64, heads=4)
    m = ItemTransformerRanker(a, 'cpu', 300, 200, None)
    assert torch.equal(m.transformer_encoder.pos_emb.pe[0], positional_encoding(5000, 32))


def test_no_cpu_fallback():
    a = default_args(model_name='item_transformer', inter_layers=1, embedding_size=32, ff_size=64, heads=4)
    m = ItemTransformerRanker(a, 'cpu', 300, 200, None, word_dists=synth.make_word_dists(300))
    b = synth.make_tem_batch(1, 4, 200, 300, Q=4, L=5)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(b)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m.test(b)
    opt = build_optim(a, m, None)
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        opt.step()


def test_unsupported_heads_raise():
    for over in (dict(model_name='ZAM'), dict(model_name='AEM'), dict(model_name='review_transformer'),
                 dict(model_name='item_transformer', use_dot_prod=False)):
        with pytest.raises(NotImplementedError):
            ItemTransformerRanker(default_args(**over), 'cpu', 300, 200, None)
    with pytest.raises(NotImplementedError):
        Optimizer('sparseadam', 0.1, 5.0)
    with pytest.raises(RuntimeError, match='Invalid optim method'):      # optimizers.py:194-195
        Optimizer('rmsprop', 0.1, 5.0)
    assert Optimizer('sgd', 0.1, 5.0).method == 'sgd'                     # the reference's other methods: tests/test_gpu_optim_methods.py


def test_product_never_imports_oracle():
    import prodsearch_amd
    pkg = os.path.dirname(prodsearch_amd.__file__)
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, re.M), f


def test_alias_table_host(lib):
    wd = synth.make_word_dists(500)
    prob = np.zeros(500, dtype=np.float32)
    alias = np.zeros(500, dtype=np.int32)
    _lib.check(lib.ps_build_alias_host(wd.ctypes.data, 500, prob.ctypes.data, alias.ctypes.data), 'alias')
    # exact reconstruction of the distribution from (prob, alias)
    rec = prob.astype(np.float64) / 500
    np.add.at(rec, alias, (1.0 - prob.astype(np.float64)) / 500)
    assert np.abs(rec - wd).max() < 1e-7
    assert rec[-1] < 1e-9            # the pad word is never drawn


def test_workspace_layout_host(lib):
    d = _lib.PsTemDesc()
    d.B, d.K, d.L, d.Q, d.W, d.C = 384, 20, 20, 8, 1, 0
    d.d, d.H, d.F, d.n_layers = 128, 8, 512, 1
    d.product_size, d.vocab_size = 18357, 32387
    d.use_pos_emb, d.training, d.dropout = 1, 1, 0.1
    lay = _lib.PsTemWsLayout()
    _lib.check(lib.ps_tem_workspace_layout(d, lay), 'layout')
    assert lay.R == 21 and lay.S == 21 and lay.total_floats > 0
    d.dropout = 0.0
    _lib.check(lib.ps_tem_workspace_layout(d, lay), 'layout')
    assert lay.R == 1
    d.d = 100                         # unsupported shape -> error code + message, never a crash
    assert lib.ps_tem_workspace_layout(d, lay) != 0
    assert b'embedding_size' in lib.ps_last_error()


def test_dropout_stream_matches_oracle_philox(lib):
    from oracle import philox
    d = _lib.PsTemDesc()
    d.dropout, d.seed, d.step = 0.1, 666 + (5 << 32), 3
    rows = np.arange(0, 41)[:, None]
    cols = np.arange(0, 37)[None, :]
    for site in (0, 1, 2, 3, 4, 12, 0x203):      # fs, attn, ctx / ff1 / ff2 (16-bit column-shared form), layer-1 ff2, a review site
        want = philox.drop_mult(rows, cols, site, 3, 666 + (5 << 32), 0.1)
        got = np.array([[lib.ps_dropout_mult_host(d, site, int(r), int(c)) for c in cols[0]] for r in rows[:, 0]],
                       dtype=np.float32)
        assert (want == got).all()
    assert philox.site_is_half(2) and philox.site_is_half(3) and philox.site_is_half(4) and philox.site_is_half(12)
    assert not philox.site_is_half(0) and not philox.site_is_half(1) and not philox.site_is_half(9) and not philox.site_is_half(0x203)
    for site in (5, 3):                              # classic and half form: keep rate and scale
        keep = philox.drop_mult(np.arange(4000)[:, None], np.arange(64)[None, :], site, 1, 42, 0.1)
        assert abs((keep > 0).mean() - 0.9) < 0.005
        assert set(np.unique(keep).tolist()) == {0.0, float(np.float32(1.0 / (1.0 - float(np.float32(0.1)))))}
