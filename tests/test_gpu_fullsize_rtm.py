"""Parity of the review transformer at BASELINE.json's full configs[3] size (B=256, K=5, R=20+30, WL=100, d=128, pvc and
pv review encoders) — the paths that only exist because of that size: the fused per-replica backward (>= 1024 rows), the
side-stream inverted index of the pvc backward, the 512-workgroup slot walkers, the ballot compaction of word slots l and
l+64, the valid-row list.  Reference semantics: ``ProductRanker.forward`` (models/ps_model.py:241-358), ``PVC.py:46-61``.

One full-size CPU forward of the oracle is affordable (seconds), so loss / logits / encoder outputs are compared with it
directly; index work is bit-exact against numpy."""
import numpy as np
import pytest
import torch

from golden_util import rel_err

pytestmark = pytest.mark.gpu

V, RC, B, K, WL, U_LIM, I_LIM, D = 32387, 60000, 256, 5, 100, 20, 30, 128
R = U_LIM + I_LIM


def _setup(encoder, dropout=0.0, corrupt=0.0, train_pv=False, seed=11):
    from prodsearch_amd import ProductRanker, default_args, synth, rtm_data
    a = default_args(model_name='review_transformer', review_encoder_name=encoder, embedding_size=D, heads=8,
                     ff_size=512, inter_layers=1, neg_per_pos=K, dropout=dropout, corrupt_rate=corrupt, lr=0.0005,
                     review_word_limit=WL, uprev_review_limit=U_LIM, iprev_review_limit=I_LIM)
    wd = synth.make_word_dists(V)
    rng = synth.rng_for(seed)
    rw = torch.from_numpy(rng.integers(0, V - 1, size=(RC, WL)))
    lens = torch.from_numpy(rng.integers(WL // 4, WL + 1, size=RC))
    rw[torch.arange(WL)[None, :] >= lens[:, None]] = V - 1
    rw[-1] = V - 1
    torch.manual_seed(0)
    m = ProductRanker(a, 'cuda', V, RC, 1000, 1000, rw, None, word_dists=wd)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    batch = rtm_data.make_rtm_batch(100 + seed, B, K, RC, V, rw, Q=8, u_lim=U_LIM, i_lim=I_LIM, W=1, train_pv=train_pv,
                                    encoder=encoder, word_dists=wd)
    m.train()
    return a, sd, m, batch


def _stages(m):
    plan = next(iter(m._plans.values()))
    torch.cuda.synchronize()
    J, S = K + 1, R + 1
    return dict(scores=m.workspace_view(plan, 'scores', (B, J)).cpu(), enc=m.workspace_view(plan, 'enc', (B, J, D)).cpu(),
                valid=m.workspace_view(plan, 'valid', (B, J, S)).cpu(), x=m.workspace_view(plan, 'x', (B, J, S, D)).cpu(),
                query_emb=m.workspace_view(plan, 'query_emb', (B, D)).cpu())


def _check_forward(a, sd, m, batch, loss, keep, oloss):
    from oracle import rtm as ortm
    st = _stages(m)
    assert rel_err(loss.detach().cpu(), oloss) < 1e-4
    assert rel_err(st['scores'], keep['scores']) < 1e-4                           # fp32 logits, north_star tolerance
    mask = torch.cat([keep['pos_mask'].unsqueeze(1), keep['neg_mask']], dim=1)
    assert torch.equal(st['valid'].ne(0), mask)                                   # index work: bit-exact
    seq = torch.cat([keep['pos_seq'].unsqueeze(1), keep['neg_seq']], dim=1) * mask.unsqueeze(-1).float()
    seq = seq + ortm.positional_encoding(5000, D)[:R + 1]
    # (rows of x at padded positions are never read — every consumer walks the valid-row list — and not written either)
    assert rel_err(torch.nan_to_num(st["x"]) * mask.unsqueeze(-1), seq * mask.unsqueeze(-1)) < 1e-4
    assert rel_err(st['query_emb'], keep['query_emb']) < 1e-4
    enc = torch.cat([keep['enc_pos'].unsqueeze(1), keep['enc_neg']], dim=1)
    assert rel_err(st['enc'], enc) < 2e-4


@pytest.mark.parametrize('encoder', ['pvc', 'pv'])
def test_rtm_full_size_matches_the_oracle(encoder):
    """dropout 0, corrupt 0: loss, logits, every forward stage, the gradients of every small tensor (oracle autograd with
    the word table held constant, so nothing table-sized is saved), the touched word / review rows bit-exactly."""
    from oracle import rtm as ortm
    a, sd, m, batch = _setup(encoder)
    loss = m(batch.to('cuda'), train_pv=False)
    m.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    big = ('word_embeddings.weight', 'review_encoder.review_embeddings.weight', 'review_encoder.word_embeddings.weight',
           'review_encoder.context_embeddings.weight')
    P = {k: (v.clone().requires_grad_(True) if (k not in big and v.dtype.is_floating_point and not k.endswith('pos_emb.pe'))
             else v) for k, v in sd.items()}
    keep = {}
    oloss, _, _ = ortm.rtm_forward(P, a, batch, None, V, RC, training=True, train_pv=False, keep=keep)
    with torch.no_grad():
        _check_forward(a, sd, m, batch, loss, {k: (v.detach() if torch.is_tensor(v) else v) for k, v in keep.items()},
                       oloss.detach())
    names = [k for k, v in P.items() if torch.is_tensor(v) and v.requires_grad]
    grads = torch.autograd.grad(oloss, [P[k] for k in names], allow_unused=True)
    got = dict(m.named_parameters())
    checked = 0
    for k, g in zip(names, grads):
        if g is None or k.endswith('linear_keys.bias'):       # exactly-zero true gradient (softmax shift invariance)
            continue
        assert got[k].grad is not None, k
        assert rel_err(got[k].grad.cpu(), g) < 5e-4, k
        checked += 1
    assert checked >= 18
    # bit-exact index work: rows of the tables that received a gradient = the rows the batch addresses
    qw = batch.query_word_idxs.numpy().ravel()
    pos_ok = batch.pos_prod_ridxs.numpy() != RC - 1
    neg_ok = batch.neg_prod_ridxs.numpy() != RC - 1
    if encoder == 'pvc':
        words = np.concatenate([qw, batch.pos_prod_rword_idxs.numpy()[pos_ok].ravel(),
                                batch.neg_prod_rword_idxs.numpy()[neg_ok].ravel()])
    else:
        words = qw
        revs = np.unique(np.concatenate([batch.pos_prod_ridxs.numpy()[pos_ok], batch.neg_prod_ridxs.numpy()[neg_ok]]))
        got_r = torch.nonzero(m.review_encoder.review_embeddings.weight.grad.ne(0).any(1)).flatten().cpu().numpy()
        assert np.array_equal(got_r, revs)
    words = np.setdiff1d(np.unique(words), [V - 1])
    got_w = torch.nonzero(m.word_embeddings.weight.grad.ne(0).any(1)).flatten().cpu().numpy()
    assert np.array_equal(got_w, words)


def test_rtm_full_size_token_corruption_follows_the_philox_stream():
    """corrupt_rate 0.9 (the reference default, PVC.py:46-54), dropout 0: the product's token masks are the ones
    oracle/philox.py restates — the oracle run with them reproduces every review vector, logit and the loss; the
    forward is bitwise deterministic and the masks move with the step counter."""
    from oracle import rtm as ortm
    from oracle.philox import RtmPhiloxDropout
    a, sd, m, batch = _setup('pvc', corrupt=0.9)
    b = batch.to('cuda')
    with torch.no_grad():
        loss = m(b, train_pv=False)
        st1 = _stages(m)
        gen = RtmPhiloxDropout(0.0, getattr(a, 'seed', 666), m._fwd_step, B, K, a.heads, R + 1, 1, 0.9)
        keep = {}
        oloss, _, _ = ortm.rtm_forward(sd, a, batch, None, V, RC, training=True, train_pv=False, drop=None,
                                       tok_drop=gen.tok, keep=keep)
        _check_forward(a, sd, m, batch, loss, keep, oloss)
        loss2 = m(b, train_pv=False)                    # next step: other masks
        assert float(loss2) != float(loss)
    a2, sd2, m2, batch2 = _setup('pvc', corrupt=0.9)
    with torch.no_grad():
        loss_again = m2(batch2.to('cuda'), train_pv=False)
        st2 = _stages(m2)
    assert float(loss_again) == float(loss)             # same seed and step: identical bits
    vm = st1['valid'].ne(0).unsqueeze(-1)
    assert torch.equal(st1['valid'], st2['valid'])
    assert torch.equal(torch.nan_to_num(st1['x']) * vm, torch.nan_to_num(st2['x']) * vm) and torch.equal(st1['scores'], st2['scores'])


def test_rtm_full_size_training_step_with_reference_defaults_is_finite_and_learns():
    """dropout 0.1 + corrupt 0.9 (reference defaults), 12 steps of the trainer's call order: finite losses, parameters
    move, and the loss on the fixed batch goes down (the fused backward, side-stream index and 512-workgroup walkers are
    all on this path)."""
    from prodsearch_amd import build_optim
    a, sd, m, batch = _setup('pvc', dropout=0.1, corrupt=0.9)
    optim = build_optim(a, m, None)              # lr 0.0005 (README.md:13-25)
    b = batch.to('cuda')
    losses = []
    for _ in range(12):
        loss = m(b, train_pv=False)
        m.zero_grad()
        loss.backward()
        optim.step()
        losses.append(float(loss.detach()))
    print("losses", losses)
    assert all(np.isfinite(losses))
    assert np.mean(losses[-3:]) < np.mean(losses[:3])          # (every step draws new dropout / corruption masks)
    moved = (m.transformer_encoder.wo.weight.detach().cpu() - sd['transformer_encoder.wo.weight']).abs().max()
    assert float(moved) > 0
