"""GPU parity of the RTM path (ProductRanker through the ps_rtm_* C ABI) against the golden vectors
the reference's ProductRanker produced (tests/golden/rtm_*.npz) — loss, logits, gradients,
post-Adam parameters, eval scores; dropout / token-corruption cases use the shared Philox masks."""
import pytest
import torch

from golden_util import rel_err
from golden_util_rtm import RTM_CASES, RtmGolden

pytestmark = pytest.mark.gpu

LOSS_TOL, GRAD_TOL = 1e-4, 5e-4


def _model(g, training=True):
    from prodsearch_amd.ps_model import ProductRanker
    torch.manual_seed(0)
    m = ProductRanker(g.args, 'cuda', g.V, g.RC, 50, 40, g.review_words, None, word_dists=g.word_dists)
    sd = g.params()
    missing = m.load_state_dict(sd, strict=False)
    real_missing = [k for k in missing.missing_keys
                    if not (k.endswith('pos_emb.pe') or k.startswith('review_encoder.word_embeddings')
                            or k.startswith('review_encoder.context_embeddings'))]
    assert real_missing == [], real_missing
    m.train(training)
    return m


def _neg(g, step):
    nw = g.neg_words(step)
    return None if nw is None else nw.cuda()


@pytest.mark.parametrize('case', RTM_CASES)
def test_rtm_state_dict_keys(case):
    g = RtmGolden(case)
    m = _model(g)
    assert list(m.state_dict().keys()) == g.meta['state_dict_keys']
    assert [n for n, _ in m.named_parameters()] == g.meta['param_names']


@pytest.mark.parametrize('case', RTM_CASES)
def test_rtm_forward_matches_reference(case):
    from oracle import rtm as ortm
    g = RtmGolden(case)
    m = _model(g)
    with torch.no_grad():
        loss = m(g.batch().to('cuda'), train_pv=g.train_pv, neg_word_idxs=_neg(g, 0))
    assert rel_err(loss.cpu(), g.tensor('loss_0')) < LOSS_TOL
    # the HIP path's own logits against the REFERENCE's (golden), straight from the workspace
    plan = next(iter(m._plans.values()))
    B, K, R, d = g.B, g.K, g.R, g.args.embedding_size
    S, J = R + 1, K + 1
    torch.cuda.synchronize()
    scores = m.workspace_view(plan, 'scores', (B, J)).cpu()
    assert rel_err(scores, g.tensor('prod_scores')) < LOSS_TOL
    real = g.batch().pos_prod_ridxs.reshape(-1).ne(g.RC - 1)
    if g.train_pv:
        pv = m.workspace_view(plan, 'pv_scores', (B * R, g.W, J)).cpu()
        assert rel_err(pv[real], g.tensor('pv_scores')[real]) < LOSS_TOL          # padded reviews carry no PV loss
    # every stage against the oracle run with the same Philox masks
    gen = g.dropout(0)
    drop = gen if (gen is not None and g.args.dropout > 0) else None
    tok = gen.tok if (gen is not None and gen.corrupt_rate > 0) else None
    keep = {}
    with torch.no_grad():
        ortm.rtm_forward(g.params(), g.args, g.batch(), g.neg_words(0), g.V, g.RC, training=True,
                         train_pv=g.train_pv, drop=drop, tok_drop=tok, keep=keep)
    assert rel_err(keep['scores'], g.tensor('prod_scores')) < 1e-4                # the oracle itself is pinned
    assert rel_err(m.workspace_view(plan, 'query_emb', (B, d)).cpu(), keep['query_emb']) < 2e-4
    mask = torch.cat([keep['pos_mask'].unsqueeze(1), keep['neg_mask']], dim=1)                    # [B,J,S]
    assert torch.equal(m.workspace_view(plan, 'valid', (B, J, S)).cpu().ne(0), mask)               # bit-exact
    seq = torch.cat([keep['pos_seq'].unsqueeze(1), keep['neg_seq']], dim=1) * mask.unsqueeze(-1).float()
    if g.args.use_pos_emb:
        seq = seq + ortm.positional_encoding(5000, d)[:S]
    # (rows of x at padded positions are never read — every consumer walks the valid-row list — and not written either)
    mk = mask.unsqueeze(-1).float()
    assert rel_err(torch.nan_to_num(m.workspace_view(plan, 'x', (B, J, S, d)).cpu()) * mk, seq * mk) < 2e-4
    enc = torch.cat([keep['enc_pos'].unsqueeze(1), keep['enc_neg']], dim=1)
    assert rel_err(m.workspace_view(plan, 'enc', (B, J, d)).cpu(), enc) < 2e-4
    assert torch.equal(m.workspace_view(plan, 'weight', (B, J)).cpu(), keep['weight'])


@pytest.mark.parametrize('case', RTM_CASES)
def test_rtm_gradients_and_steps(case):
    from prodsearch_amd import build_optim
    g = RtmGolden(case)
    m = _model(g)
    init = {k: v.clone() for k, v in g.params().items()}
    optim = build_optim(g.args, m, None)
    b = g.batch().to('cuda')
    for step in range(g.steps):
        loss = m(b, train_pv=g.train_pv, neg_word_idxs=_neg(g, step))
        m.zero_grad()
        loss.backward()
        if step == 0:
            torch.cuda.synchronize()
            none = sorted(n for n, p in m.named_parameters() if p.grad is None)
            assert none == sorted(g.meta['none_grads'])
            for n, p in m.named_parameters():
                if p.grad is None or n.endswith('linear_keys.bias'):
                    continue
                assert rel_err(p.grad.cpu(), g.tensor('grad_' + n)) < GRAD_TOL, n
        optim.step()
        assert rel_err(loss.detach().cpu(), g.tensor('loss_%d' % step)) < 2 * LOSS_TOL, step
    last = g.steps - 1
    sd = m.state_dict()
    for n, _ in m.named_parameters():
        if n.endswith('linear_keys.bias'):
            continue
        ref = g.tensor('param%d_%s' % (last, n), base=init[n])
        got = sd[n].cpu()
        assert float((got - ref).abs().max()) < 2e-3 * float(ref.abs().max()) + 0.02 * g.args.lr * g.steps, n


@pytest.mark.parametrize('case', ['rtm_pvc_drop', 'rtm_pv'])
def test_rtm_gradients_and_steps_fused_backward(case):
    """Same check with the fused per-replica backward kernel forced on (it normally starts at 1024 replica rows)."""
    from prodsearch_amd import _lib
    old = _lib.load().ps_set_fuse_bwd_min(1)
    try:
        test_rtm_gradients_and_steps(case)
    finally:
        _lib.load().ps_set_fuse_bwd_min(old)


@pytest.mark.parametrize('case', RTM_CASES)
def test_rtm_eval_scores(case):
    g = RtmGolden(case)
    m = _model(g, training=False)
    with torch.no_grad():
        m.get_review_embeddings()
        assert abs(float(m.review_embeddings.double().sum().cpu()) - float(g.z['test_review_embeddings_sum'])) < 1e-2
        s = m.test(g.test_batch().to('cuda')).cpu()
    m.clear_review_embbeddings()
    assert rel_err(s, g.tensor('test_scores')) < LOSS_TOL


@pytest.mark.parametrize('case', ['rtm_pvc', 'rtm_pv_ui'])
def test_rtm_unequal_sequence_widths(case):
    """The reference's loader pads positives and negatives to their own longest sequence; widening one side with
    pad positions (as ProductRanker._same_width does for the narrower one) must not move the loss or any gradient."""
    import torch.nn.functional as F
    g = RtmGolden(case)
    m = _model(g)
    b = g.batch().to('cuda')
    loss = m(b, train_pv=g.train_pv, neg_word_idxs=_neg(g, 0))
    m.zero_grad(); loss.backward()
    torch.cuda.synchronize()
    ref = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    assert rel_err(loss.detach().cpu(), g.tensor('loss_0')) < LOSS_TOL
    pads = dict(neg_prod_ridxs=g.RC - 1, neg_seg_idxs=3, neg_user_idxs=40, neg_item_idxs=50,
                neg_prod_rword_idxs=g.V - 1, neg_prod_rword_masks=0, neg_prod_rword_idxs_pvc=g.V - 1)
    for name, val in pads.items():
        t = getattr(b, name, None)
        if t is None:
            continue
        spec = [0, 0] * (t.dim() - 3) + [0, 2]
        setattr(b, name, F.pad(t, spec, value=val))
    assert b.neg_prod_ridxs.shape[2] == b.pos_prod_ridxs.shape[1] + 2
    if g.train_pv:       # the PV-loss draws are laid out per positive review slot: recompute them at the new width
        R2 = b.neg_prod_ridxs.shape[2]
        nw = _neg(g, 0).view(g.B, g.R, -1)
        nw = F.pad(nw, (0, 0, 0, R2 - g.R), value=0).reshape(g.B * R2, -1)
    else:
        nw = None
    loss2 = m(b, train_pv=g.train_pv, neg_word_idxs=nw)
    m.zero_grad(); loss2.backward()
    torch.cuda.synchronize()
    assert rel_err(loss2.detach().cpu(), loss.detach().cpu()) < 1e-5
    for n, p in m.named_parameters():
        if p.grad is not None and not n.endswith('linear_keys.bias'):
            assert rel_err(p.grad.cpu(), ref[n].cpu()) < 2e-4, n
