"""The RTM oracle (oracle/rtm.py) against every golden vector the reference's ProductRanker produced."""
import pytest
import torch

from golden_util import rel_err
from golden_util_rtm import RTM_CASES, RtmGolden
from oracle import optim as ooptim
from oracle import rtm as ortm


def _fwd(g, P, step, keep=None):
    gen = g.dropout(step)
    drop = gen if (gen is not None and g.args.dropout > 0) else None
    tok = gen.tok if (gen is not None and gen.corrupt_rate > 0) else None
    return ortm.rtm_forward(P, g.args, g.batch(), g.neg_words(step), g.V, g.RC, training=True,
                            train_pv=g.train_pv, drop=drop, tok_drop=tok, keep=keep)


@pytest.mark.parametrize('case', RTM_CASES)
def test_rtm_forward(case):
    g = RtmGolden(case)
    P = g.params()
    keep = {}
    with torch.no_grad():
        loss, ps, pv = _fwd(g, P, 0, keep)
    assert rel_err(loss, g.tensor('loss_0')) < 5e-6
    assert rel_err(keep['scores'], g.tensor('prod_scores')) < 2e-5
    if g.train_pv:
        assert rel_err(keep['pv_scores'], g.tensor('pv_scores')) < 2e-5


@pytest.mark.parametrize('case', RTM_CASES)
def test_rtm_gradients_and_adam(case):
    g = RtmGolden(case)
    a = g.args
    P = {k: v.clone().requires_grad_(True) for k, v in g.params().items()}
    init = {k: v.detach().clone() for k, v in P.items()}
    opt = ooptim.ClipAdam(a.lr, a.max_grad_norm, a.beta1, a.beta2, 1e-9, a.l2_lambda, a.decay_method, a.warmup_steps)
    pad = {'word_embeddings.weight': g.V - 1, 'seg_embeddings.weight': 3,
           'review_encoder.review_embeddings.weight': g.RC - 1,
           'user_emb.weight': -1, 'product_emb.weight': -1}          # padding_idx rows: no gradient
    for step in range(g.steps):
        loss, _, _ = _fwd(g, P, step)
        assert rel_err(loss, g.tensor('loss_%d' % step)) < 1e-5
        names = list(P)
        gs = torch.autograd.grad(loss, [P[n] for n in names], allow_unused=True)
        grads = {}
        for n, gr in zip(names, gs):
            if gr is not None and n in pad:
                gr = gr.clone(); gr[pad[n]] = 0
            grads[n] = gr
        if step == 0:
            assert sorted(n for n, v in grads.items() if v is None) == sorted(g.meta['none_grads'])
            for n, v in grads.items():
                if v is None:
                    continue
                ref = g.tensor('grad_' + n)
                if n.endswith('linear_keys.bias'):
                    continue
                assert rel_err(v, ref) < 5e-5, n
        with torch.no_grad():
            opt.step(P, grads)
    last = g.steps - 1
    for n in P:
        if n.endswith('linear_keys.bias'):
            continue
        ref = g.tensor('param%d_%s' % (last, n), base=init[n])
        assert rel_err(P[n].detach(), ref) < 2e-5, n


@pytest.mark.parametrize('case', RTM_CASES)
def test_rtm_eval_scores(case):
    g = RtmGolden(case)
    P = g.params()
    with torch.no_grad():
        rev = ortm.rtm_review_embeddings(P, g.args, g.review_words, g.V)
        assert abs(float(rev.double().sum()) - float(g.z['test_review_embeddings_sum'])) < 1e-3
        s = ortm.rtm_test(P, g.args, g.test_batch(), rev, g.V, g.RC)
    assert rel_err(s, g.tensor('test_scores')) < 2e-5
