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


# ---- the weight-gradient split count changes nothing but fp32 rounding: two deterministic runs whose
# split reductions are cut differently (PS_WGRAD_ROWS is read once per process: one child per setting) are each bitwise
# repeatable, their gradients agree to rounding, and after Adam the only elements that differ by more than a sliver of lr are
# those whose gradient is itself at rounding level — Adam's g / (|g| + eps) turns any sign change there into a full +-lr step.
_SPLIT_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %(root)r)
from prodsearch_amd import ItemTransformerRanker, _lib, build_optim, readme_tem_args, synth
assert _lib.load().ps_set_deterministic(1) in (0, 1)
P_, V, K, L, Q, W, B = 18357, 32387, 20, 20, 8, 1, 384
a = readme_tem_args(dropout=0.1, lr=0.002)
wd = synth.make_word_dists(V)
torch.manual_seed(77)
m = ItemTransformerRanker(a, 'cuda', V, P_, None, word_dists=wd)
sd = synth.make_state_dict(synth.tem_param_shapes(a, V, P_), 31, {'product_emb.weight': P_})
m.load_state_dict(sd, strict=False)
optim = build_optim(a, m, None)
m.train()
batch = synth.make_tem_batch(500, B, P_, V, Q=Q, L=L, W=W, word_dists=wd).to('cuda')
ni, nw = synth.sample_negatives(600, B, K, W, P_, wd)
loss = m(batch, neg_item_idxs=ni.cuda(), neg_word_idxs=nw.cuda())
m.zero_grad(); loss.backward()
out = {}
for n, p in m.named_parameters():
    if p.grad is not None:
        out['g:' + n] = p.grad.detach().cpu().numpy().copy()
optim.step()
torch.cuda.synchronize()
for n, p in m.named_parameters():
    out['p:' + n] = p.detach().cpu().numpy().copy()
out['loss'] = np.float64(float(loss.detach()))
np.savez(sys.argv[1], **out)
'''


def _split_run(tmp_path, tag, rows):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop('PS_WGRAD_BLOCKS', None)
    env['PS_WGRAD_ROWS'] = str(rows)
    out = str(tmp_path / ('split_%s.npz' % tag))
    proc = subprocess.run([sys.executable, '-c', _SPLIT_CHILD % {'root': root}, out], env=env, capture_output=True, text=True,
                          timeout=300)
    assert proc.returncode == 0, proc.stderr[-3000:]
    return dict(np.load(out))


def test_two_weight_gradient_split_counts_differ_by_rounding_only(tmp_path):
    lr = 0.002
    a1 = _split_run(tmp_path, 'a1', 512)          # the default: ~16 splits of the 8,064 replica rows
    a2 = _split_run(tmp_path, 'a2', 512)
    b = _split_run(tmp_path, 'b', 192)            # ~42 splits
    assert sorted(a1) == sorted(a2) == sorted(b)
    for k in a1:
        assert np.array_equal(a1[k], a2[k]), k                                   # one setting: bit for bit
    differs = [k for k in a1 if k.startswith('g:') and not np.array_equal(a1[k], b[k])]
    assert differs, "the split count did not change any sum: PS_WGRAD_ROWS no longer reaches the deterministic weight gradients"
    assert abs(a1['loss'] - b['loss']) == 0.0                                     # ... and nothing in the forward
    worst = 0.0
    for k in a1:
        if not k.startswith('g:'):
            continue
        g1, g2 = a1[k].astype(np.float64), b[k].astype(np.float64)
        gmax = float(np.abs(g1).max())
        if gmax == 0.0:
            assert float(np.abs(g2).max()) == 0.0, k
            continue
        assert float(np.abs(g1 - g2).max()) <= 1e-5 * gmax, (k, float(np.abs(g1 - g2).max()), gmax)     # rounding of fp32 sums
        p1, p2 = a1['p:' + k[2:]], b['p:' + k[2:]]
        moved = np.abs(p1.astype(np.float64) - p2) > 1e-3 * lr
        if moved.any():
            # only where the gradient is at the noise floor of its tensor (the clip factor is common to all of them)
            worst = max(worst, float(np.abs(g1[moved]).max()) / gmax)
            assert float(np.abs(g1[moved]).max()) <= 5e-5 * gmax, (k, float(np.abs(g1[moved]).max()), gmax)
            assert float(np.abs(p1.astype(np.float64) - p2).max()) <= 2 * lr * 1.0001 + 1e-7, k
    print("largest |g| / max|g| among elements whose Adam step differed: %.2e" % worst)
