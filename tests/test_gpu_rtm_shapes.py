"""Review transformer at shapes the golden cases (d = 32 / 64 / 128) and the full-size test (d = 128) do not reach: the
wide instantiations of the embed forward / backward kernels (d = 256: four 16-byte chunks per lane and four columns per lane;
d = 512: the per-slot embed kernel, eight columns per lane, the index built behind the embed backward), ragged review-row
counts (B*R not a multiple of the 16 rows a backward wave owns).  Loss and EVERY gradient against the oracle's autograd
(``ProductRanker.forward``, models/ps_model.py:241-358; ``PVC.py:46-61``), dropout 0, corrupt 0."""
import pytest
import torch

from golden_util import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('d,heads,B,K,u_lim,i_lim,WL,encoder', [
    (256, 8, 6, 2, 5, 6, 40, 'pvc'),
    (512, 8, 3, 2, 4, 3, 70, 'pvc'),
    (256, 8, 5, 3, 3, 4, 30, 'avg'),
    (64, 4, 7, 1, 7, 6, 100, 'pvc'),
    (128, 8, 9, 4, 9, 4, 128, 'pvc'),
])
def test_rtm_wide_and_ragged_shapes_match_the_oracle(d, heads, B, K, u_lim, i_lim, WL, encoder):
    from oracle import rtm as ortm
    from prodsearch_amd import ProductRanker, default_args, synth, rtm_data
    V, RC = 1500, 900
    a = default_args(model_name='review_transformer', review_encoder_name=encoder, embedding_size=d, heads=heads,
                     ff_size=2 * d, inter_layers=1, neg_per_pos=K, dropout=0.0, corrupt_rate=0.0, lr=0.0005,
                     review_word_limit=WL, uprev_review_limit=u_lim, iprev_review_limit=i_lim)
    wd = synth.make_word_dists(V)
    rng = synth.rng_for(3)
    rw = torch.from_numpy(rng.integers(0, V - 1, size=(RC, WL)))
    lens = torch.from_numpy(rng.integers(1, WL + 1, size=RC))
    rw[torch.arange(WL)[None, :] >= lens[:, None]] = V - 1
    rw[-1] = V - 1
    torch.manual_seed(0)
    m = ProductRanker(a, 'cuda', V, RC, 50, 60, rw, None, word_dists=wd)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    batch = rtm_data.make_rtm_batch(21, B, K, RC, V, rw, Q=5, u_lim=u_lim, i_lim=i_lim, W=1, train_pv=False,
                                    encoder=encoder, word_dists=wd)
    m.train()
    loss = m(batch.to('cuda'), train_pv=False)
    m.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    P = {k: (v.clone().requires_grad_(True) if (v.dtype.is_floating_point and not k.endswith('pos_emb.pe')) else v)
         for k, v in sd.items()}
    oloss, _, _ = ortm.rtm_forward(P, a, batch, None, V, RC, training=True, train_pv=False)
    assert rel_err(loss.detach().cpu(), oloss.detach()) < 1e-4
    names = [k for k, v in P.items() if torch.is_tensor(v) and v.requires_grad]
    grads = torch.autograd.grad(oloss, [P[k] for k in names], allow_unused=True)
    got = dict(m.named_parameters())
    checked = 0
    for k, g in zip(names, grads):
        if g is None or k.endswith('linear_keys.bias') or k not in got or float(g.abs().max()) == 0.0:
            continue
        assert got[k].grad is not None, k
        assert rel_err(got[k].grad.cpu(), g) < 5e-4, k
        checked += 1
    assert checked >= 16
    m.check_index_errors() if hasattr(m, 'check_index_errors') else None


def test_second_backward_over_the_same_forward_accumulates_the_same_gradient():
    """``loss.backward(retain_graph=True); loss.backward()``: the second backward rebuilds the inverted index of the word
    gradient from the forward's counts (its allocator must start from 0 again) and adds the same gradient once more."""
    from prodsearch_amd import ProductRanker, default_args, synth, rtm_data
    V, RC, B, K, WL = 1500, 900, 6, 2, 40
    a = default_args(model_name='review_transformer', review_encoder_name='pvc', embedding_size=128, heads=8, ff_size=256,
                     inter_layers=1, neg_per_pos=K, dropout=0.0, corrupt_rate=0.0, lr=0.0005, review_word_limit=WL,
                     uprev_review_limit=5, iprev_review_limit=6)
    wd = synth.make_word_dists(V)
    rng = synth.rng_for(5)
    rw = torch.from_numpy(rng.integers(0, V - 1, size=(RC, WL)))
    rw[-1] = V - 1
    torch.manual_seed(0)
    m = ProductRanker(a, 'cuda', V, RC, 50, 60, rw, None, word_dists=wd)
    batch = rtm_data.make_rtm_batch(22, B, K, RC, V, rw, Q=5, u_lim=5, i_lim=6, W=1, train_pv=False, encoder='pvc',
                                    word_dists=wd)
    m.train()
    loss = m(batch.to('cuda'), train_pv=False)
    m.zero_grad()
    loss.backward(retain_graph=True)
    g1 = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    loss.backward()
    torch.cuda.synchronize()
    for n, p in m.named_parameters():
        if p.grad is None or float(g1[n].abs().max()) == 0.0:
            continue
        assert rel_err(p.grad.cpu(), (2 * g1[n]).cpu()) < 1e-5, n
