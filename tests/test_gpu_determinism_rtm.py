"""Deterministic mode of the review transformer (``ps_set_deterministic`` / PS_DETERMINISTIC=1; csrc/rtm.hip): the training step
of ``ProductRanker`` with a word-mean review encoder (pvc = BASELINE configs[3], fs, avg) is bitwise reproducible — the word
index hands out its ranks in task order (one wave per partition, no atomics), a word's occurrences are summed in list order by
one wave (sixteen fixed slices for the Zipf heads), d wo / d query_emb / the fs bias as fixed-order sums, weight gradients as
per-split partials added in split order, one stream.  The default path differs from it only by the order of fp32 additions.
The user / item embedding rows, the pv encoder's review rows and the PV loss's word rows go through the sole-owner row scatter of
the item transformer (one owner half-wave per table row, tasks in order).
Reference semantics: models/ps_model.py:241-358, models/PVC.py:46-61 (its own CUDA embedding backward is not deterministic)."""
import numpy as np
import pytest
import torch

from golden_util import rel_err

pytestmark = pytest.mark.gpu

V, RC, K, WL, U_LIM, I_LIM, D = 32387, 60000, 5, 100, 20, 30, 128


def _train(det, encoder, B, steps=2, dropout=0.1, corrupt=0.9, ui=False, train_pv=False):
    from prodsearch_amd import ProductRanker, _lib, build_optim, default_args, synth, rtm_data
    lib = _lib.load()
    old = lib.ps_set_deterministic(1 if det else 0)
    try:
        a = default_args(model_name='review_transformer', review_encoder_name=encoder, embedding_size=D, heads=8,
                         ff_size=512, inter_layers=1, neg_per_pos=K, dropout=dropout, corrupt_rate=corrupt, lr=0.0005,
                         review_word_limit=WL, uprev_review_limit=U_LIM, iprev_review_limit=I_LIM,
                         use_user_emb=ui, use_item_emb=ui)
        wd = synth.make_word_dists(V)
        rng = synth.rng_for(3)
        # Zipf-distributed review words: the head words collect far more than WR_DET_LIM occurrences at B = 256
        rw = torch.from_numpy(rng.choice(V - 1, size=(RC, WL), p=_zipf(V - 1)))
        lens = torch.from_numpy(rng.integers(WL // 4, WL + 1, size=RC))
        rw[torch.arange(WL)[None, :] >= lens[:, None]] = V - 1
        rw[-1] = V - 1
        torch.manual_seed(0)
        m = ProductRanker(a, 'cuda', V, RC, 1000, 1000, rw, None, word_dists=wd)
        optim = build_optim(a, m, None)
        m.train()
        losses = []
        for s in range(steps):
            batch = rtm_data.make_rtm_batch(300 + s, B, K, RC, V, rw, Q=8, u_lim=U_LIM, i_lim=I_LIM, W=1, train_pv=train_pv,
                                            encoder=encoder, word_dists=wd, user_size=1000 if ui else None,
                                            product_size=1000 if ui else None)
            loss = m(batch.to('cuda'), train_pv=train_pv)
            m.zero_grad()
            loss.backward()
            if s == 0:
                g0 = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
            optim.step()
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        return losses, g0, {n: p.detach().clone() for n, p in m.named_parameters()}
    finally:
        lib.ps_set_deterministic(old)


def _zipf(n):
    p = 1.0 / np.arange(1, n + 1) ** 1.05
    return p / p.sum()


@pytest.mark.parametrize('encoder,B,corrupt,ui,tp', [
    ('pvc', 256, 0.9, False, False), ('pvc', 64, 0.0, False, False), ('fs', 32, 0.0, False, False), ('avg', 32, 0.0, False, False),
    ('pvc', 64, 0.9, True, False), ('fs', 32, 0.0, True, False),            # + user / item embedding rows
    ('pv', 48, 0.0, False, False), ('pv', 32, 0.0, True, False),            # the pv encoder's review rows
    ('pv', 32, 0.0, False, True), ('pvc', 32, 0.9, True, True)])            # + the PV loss (train_pv): its word rows
def test_rtm_deterministic_mode_repeats_bit_for_bit_and_agrees_with_the_default_path(encoder, B, corrupt, ui, tp):
    l1, g1, p1 = _train(True, encoder, B, corrupt=corrupt, ui=ui, train_pv=tp)
    l2, g2, p2 = _train(True, encoder, B, corrupt=corrupt, ui=ui, train_pv=tp)
    assert l1 == l2
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n                                     # every gradient of the first step, bitwise
    for n in p1:
        assert torch.equal(p1[n], p2[n]), n                                     # every parameter after two clip+Adam steps
    l0, g0, p0 = _train(False, encoder, B, corrupt=corrupt, ui=ui, train_pv=tp)
    if ui:
        assert 'user_emb.weight' in g1 and 'product_emb.weight' in g1 and float(g1['user_emb.weight'].abs().max()) > 0
    assert np.allclose(l0, l1, rtol=1e-5)
    assert sorted(g0) == sorted(g1)
    for n in g1:                                                                # the same sums in another order
        assert rel_err(g1[n], g0[n]) < 2e-5, (n, rel_err(g1[n], g0[n]))


def test_rtm_deterministic_mode_refuses_what_it_does_not_cover():
    """Nothing of the review transformer is refused any more (round 3: the pv encoder, the PV loss and the user / item rows go
    through the sole-owner row scatter); what IS still refused is a configuration whose word index cannot be the LDS histogram."""
    import os
    if os.environ.get('PS_RTM_HIST', '1') != '0':
        pytest.skip("the LDS-histogram index is on: every configuration is covered")
    with pytest.raises(RuntimeError, match='deterministic mode'):
        _train(True, 'pvc', 16, steps=1)
