"""Deterministic mode (``ps_set_deterministic`` / PS_DETERMINISTIC=1): the item-transformer training step is bitwise
reproducible run to run — one stream, weight gradients as per-split partials added up in split order, table scatters by
sole-owner half-waves walking the task lists in order.  The reference's own embedding backward (CUDA index_add_) is not
deterministic; this mode exists so that a run can be replayed bit for bit (trainer.py:64-83 call order)."""
import numpy as np
import pytest
import torch

from golden_util import rel_err

pytestmark = pytest.mark.gpu


def _train(det, steps=3, dropout=0.1, B=96):
    from prodsearch_amd import ItemTransformerRanker, _lib, build_optim, readme_tem_args, synth
    lib = _lib.load()
    old = lib.ps_set_deterministic(1 if det else 0)
    try:
        P_, V, K, L, Q, W = 18357, 32387, 20, 20, 8, 1
        a = readme_tem_args(dropout=dropout, lr=0.002)
        wd = synth.make_word_dists(V)
        torch.manual_seed(77)
        m = ItemTransformerRanker(a, 'cuda', V, P_, None, word_dists=wd)
        optim = build_optim(a, m, None)
        m.train()
        losses = []
        for s in range(steps):
            batch = synth.make_tem_batch(500 + s, B, P_, V, Q=Q, L=L, W=W, word_dists=wd).to('cuda')
            ni, nw = synth.sample_negatives(600 + s, B, K, W, P_, wd)
            loss = m(batch, neg_item_idxs=ni.cuda(), neg_word_idxs=nw.cuda())
            m.zero_grad()
            loss.backward()
            optim.step()
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        return losses, {n: p.detach().clone() for n, p in m.named_parameters()}
    finally:
        lib.ps_set_deterministic(old)


@pytest.mark.parametrize('dropout', [0.1, 0.0])
def test_deterministic_mode_repeats_bit_for_bit_and_agrees_with_the_default_path(dropout):
    l1, p1 = _train(True, dropout=dropout)
    l2, p2 = _train(True, dropout=dropout)
    assert l1 == l2
    for n in p1:
        assert torch.equal(p1[n], p2[n]), n                                    # bitwise, every parameter, after 3 Adam steps
    l0, p0 = _train(False, dropout=dropout)
    assert np.allclose(l0, l1, rtol=1e-5)
    for n in p1:
        if float(p0[n].abs().max()) == 0.0 or n.endswith('linear_keys.bias'):   # (exactly-zero true gradient: rounding noise)
            continue
        # Adam normalises every element's step: where a gradient is at rounding level the two summation orders may even
        # disagree in sign, so the paths are compared by the size of the parameter change, not element by element
        assert float((p1[n] - p0[n]).abs().max()) <= 2 * 0.002 * 3 + 1e-6, n
        assert rel_err(p1[n], p0[n]) < 5e-2, n
