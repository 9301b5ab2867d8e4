"""The oracle against every golden vector the reference produced (CPU only).

This is what pins the oracle: ``tests/golden/*.npz`` hold the reference's own
loss / intermediates / gradients / post-Adam parameters / eval scores on fixed
inputs (made by tests/golden/make_golden.py in the build container)."""
import pytest
import torch

from golden_util import CASES, Golden, rel_err
from oracle import optim as ooptim
from oracle import tem as otem

FP_TOL = 2e-6      # oracle vs reference, fp32 CPU both: same ops, tiny reassociation only


def _fwd(g, P, step, replicate, keep=None):
    ni, nw = g.negs(step)
    fn = otem.qem_forward if g.args.model_name == 'QEM' else otem.tem_forward
    kw = {} if g.args.model_name == 'QEM' else dict(replicate=replicate)
    if g.args.dropout > 0:
        # dropout drawn: the reference ran with the product's Philox masks (make_golden.py hook 3);
        # only the replicated structure is meaningful (the K+1 copies diverge)
        kw['replicate'] = True
        kw['drop'] = g.dropout(step)
    return fn(P, g.args, g.batch(), ni, nw, g.V, g.P, training=True, keep=keep, **kw)


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('replicate', [False, True])
def test_forward_loss_and_intermediates(case, replicate):
    g = Golden(case)
    if g.args.model_name == 'QEM' and replicate:
        pytest.skip('QEM has no replicated encoder')
    if g.args.dropout > 0 and not replicate:
        pytest.skip('dropout drawn: replicas differ, dedup structure does not apply')
    P = g.params()
    keep = {}
    with torch.no_grad():
        loss, ps, il = _fwd(g, P, 0, replicate, keep)
    assert rel_err(loss, g.tensor('loss_0')) < FP_TOL
    assert rel_err(ps, g.tensor('ps_loss_0')) < FP_TOL
    assert rel_err(il, g.tensor('item_loss_0')) < FP_TOL
    assert rel_err(keep['query_emb'], g.tensor('query_emb')) < FP_TOL
    scores = torch.cat([keep['pos_scores'].unsqueeze(-1), keep['neg_scores']], -1)
    assert rel_err(scores, g.tensor('prod_scores')) < 1e-5
    assert rel_err(keep['word_scores'], g.tensor('word_scores')) < 1e-5
    if g.has('enc_full'):
        pos = -1 if g.args.use_item_pos else 0
        assert rel_err(keep['enc'], g.tensor('enc_full')[:, pos]) < 1e-5


@pytest.mark.parametrize('case', CASES)
def test_gradients(case):
    g = Golden(case)
    P = {k: v.clone().requires_grad_(True) for k, v in g.params().items()}
    loss, _, _ = _fwd(g, P, 0, False)
    grads = otem.grads_of(loss, P, otem.tem_pad_rows(g.args, g.V, g.P))
    none = sorted(n for n, v in grads.items() if v is None)
    assert none == sorted(g.meta['none_grads'])          # same unreachable parameters
    for n, v in grads.items():
        if v is None:
            continue
        ref = g.tensor('grad_' + n)
        assert v.shape == ref.shape
        if n.endswith('linear_keys.bias'):
            # exactly 0 in real arithmetic (softmax shift invariance): both sides are rounding noise
            scale = float(g.tensor('grad_' + n.replace('.bias', '.weight')).abs().max())
            assert float(v.abs().max()) < 1e-5 * scale and float(ref.abs().max()) < 1e-5 * scale, n
            continue
        assert rel_err(v, ref) < 2e-5, n
        # bit-exact index work: the set of touched rows of a table grad
        if v.dim() == 2 and v.shape[0] > 256:
            assert torch.equal(v.ne(0).any(1), ref.ne(0).any(1)), n


@pytest.mark.parametrize('case', CASES)
def test_multi_step_clip_adam(case):
    g = Golden(case)
    a = g.args
    P = {k: v.clone().requires_grad_(True) for k, v in g.params().items()}
    init = {k: v.detach().clone() for k, v in P.items()}
    opt = ooptim.ClipAdam(a.lr, a.max_grad_norm, a.beta1, a.beta2, 1e-9, a.l2_lambda,
                          a.decay_method, a.warmup_steps)
    pad = otem.tem_pad_rows(a, g.V, g.P)
    for step in range(g.steps):
        loss, ps, il = _fwd(g, P, step, False)
        assert rel_err(loss, g.tensor('loss_%d' % step)) < 5e-6, step
        grads = otem.grads_of(loss, P, pad)
        with torch.no_grad():
            opt.step(P, grads)
        assert abs(opt.learning_rate - float(g.z['lr_%d' % step])) < 1e-12
        if step in (0, g.steps - 1):
            for n in P:
                ref = g.tensor('param%d_%s' % (step, n), base=init[n])
                if n.endswith('linear_keys.bias'):
                    # d loss / d key-bias is exactly 0 in real arithmetic (softmax is invariant
                    # to a per-query constant), so its fp32 gradient is rounding noise and
                    # Adam(eps 1e-9) turns noise into +-lr steps: only |delta| <= lr*steps is pinned.
                    assert float((P[n].detach() - ref).abs().max()) <= 2.01 * a.lr * (step + 1), (step, n)
                    continue
                # Adam(eps 1e-9) normalises every element's step: where a gradient element is itself of the order of eps
                # (|g| ~ 1e-8 after the clip) the two sides' rounding noise becomes a different fraction of lr.  Such
                # elements are isolated (<= 2 per 10^4) and bounded by lr per step; everything else is pinned at 5e-6.
                diff = (P[n].detach() - ref).abs()
                bad = diff > 5e-6 * float(ref.abs().max())
                assert float(bad.float().mean()) <= 2e-4 and (int(bad.sum()) == 0 or
                                                             float(diff[bad].max()) <= 2.01 * a.lr * (step + 1)), (step, n)


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('replicate', [False, True])
def test_eval_scores_and_ranklist(case, replicate):
    g = Golden(case)
    P = g.params()
    b = g.batch()
    with torch.no_grad():
        if g.args.model_name == 'QEM':
            s = otem.qem_test(P, g.args, b, g.V, g.P)
        else:
            s = otem.tem_test(P, g.args, b, g.V, g.P, replicate=replicate)
    ref = g.tensor('test_scores')
    assert rel_err(s, ref) < 1e-5
    order, mrr, p1 = otem.rank_metrics(ref, b.candi_prod_idxs, b.target_prod_idxs)
    assert (order == g.z['test_ranklist']).all()             # bit-exact index work
    assert abs(mrr - float(g.z['test_mrr'])) < 1e-12 and abs(p1 - float(g.z['test_p1'])) < 1e-12
    # the oracle's own scores give the same ranklist wherever the reference's top gaps exceed fp noise
    o2, mrr2, _ = otem.rank_metrics(s, b.candi_prod_idxs, b.target_prod_idxs)
    assert abs(mrr2 - mrr) < 1e-9


def test_row_sparse_clip_adam_equals_dense_on_touched_rows_and_freezes_the_rest():
    """oracle/optim.py ``touched=``: first step identical to dense Adam on the touched rows (zero moments
    everywhere), later steps leave untouched rows alone where dense Adam keeps moving them."""
    torch.manual_seed(3)
    P1 = {'t': torch.randn(50, 8), 'w': torch.randn(4, 4)}
    P2 = {k: v.clone() for k, v in P1.items()}
    o1, o2 = ooptim.ClipAdam(0.01), ooptim.ClipAdam(0.01)
    rows_a, rows_b = torch.tensor([1, 7, 20]), torch.tensor([7, 30])
    for it, rows in enumerate([rows_a, rows_b]):
        g = {'t': torch.zeros(50, 8), 'w': torch.randn(4, 4)}
        g['t'][rows] = torch.randn(len(rows), 8)
        o1.step(P1, {k: v.clone() for k, v in g.items()})
        o2.step(P2, {k: v.clone() for k, v in g.items()}, touched={'t': rows})
        assert torch.equal(P1['w'], P2['w'])
        if it == 0:
            assert torch.equal(P1['t'], P2['t'])
    frozen = torch.tensor([1, 20])                  # touched at step 1 only
    assert not torch.equal(P1['t'][frozen], P2['t'][frozen])        # dense Adam kept applying momentum
    untouched = torch.ones(50, dtype=torch.bool)
    untouched[torch.cat([rows_a, rows_b])] = False
    assert torch.equal(P1['t'][untouched], P2['t'][untouched])      # never-touched rows: identical (no motion)
