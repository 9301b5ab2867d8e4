"""Default initialisation reproduces the reference's distributions, quirks included:
``ItemTransformerRanker.initialize_parameters`` (models/item_transformer.py:576-586), ``TransformerEncoder`` /
``FSEncoder.initialize_parameters`` (models/transformer.py:100-118, models/text_encoder.py:42-60), ``ProductRanker``
(models/ps_model.py:360-370):
  * word / seg embeddings N(0,1) — which OVERWRITES their zeroed padding rows;
  * product_emb keeps nn.Embedding's default N(0,1) with a ZERO padding row;
  * >=2-D 'weight' tensors Xavier-normal, every 'bias' 0;
  * 1-D LayerNorm gains are neither '>1-D weight' nor 'bias', so they fall through to normal_(): N(0,1), not ones;
  * product_bias / word_bias zeros."""
import math

import torch

from prodsearch_amd import ItemTransformerRanker, ProductRanker, default_args, readme_tem_args


def _moments(t):
    t = t.detach().float()
    return float(t.mean()), float(t.std())


def _check_encoder(te, d, ff):
    for name, p in te.named_parameters():
        if name.endswith('layer_norm.weight'):                       # N(0,1) gains (the quirk)
            assert abs(float(p.detach().std()) - 1.0) < 0.25 and abs(float(p.detach().mean())) < 0.3, name
            assert float((p.detach() - 1).abs().max()) > 0.5, name  # certainly not the nn.LayerNorm default of ones
        elif name.endswith('bias'):
            assert float(p.detach().abs().max()) == 0.0, name
        elif p.dim() > 1:
            fan_out, fan_in = p.shape
            want = math.sqrt(2.0 / (fan_in + fan_out))
            m, s = _moments(p)
            tol = 0.35 if p.numel() < 1000 else 0.08                 # wo is [1, d]
            assert abs(s - want) < tol * want and abs(m) < 4 * want / math.sqrt(p.numel()) + 1e-3, (name, s, want)


def test_tem_default_init_matches_the_reference_distributions():
    torch.manual_seed(3)
    a = readme_tem_args()
    P_, V = 3000, 5000
    m = ItemTransformerRanker(a, 'cpu', V, P_, None, word_dists=None)
    we, pe, se = m.word_embeddings.weight.detach(), m.product_emb.weight.detach(), m.seg_embeddings.weight.detach()
    for t in (we, pe[:P_]):
        mu, sd = _moments(t)
        assert abs(mu) < 0.01 and abs(sd - 1.0) < 0.01
    assert float(we[V - 1].abs().max()) > 0.0                        # pad row overwritten by normal_ (item_transformer.py:580)
    assert float(se[3].abs().max()) > 0.0                            # same for the seg pad row (:581)
    assert float(pe[P_].abs().max()) == 0.0                          # product_emb keeps its zero pad row
    assert float(m.product_bias.detach().abs().max()) == 0.0 and float(m.word_bias.detach().abs().max()) == 0.0
    _check_encoder(m.transformer_encoder, a.embedding_size, a.ff_size)
    fw = m.query_encoder.f_W
    assert abs(_moments(fw.weight)[1] - math.sqrt(2.0 / 256)) < 0.08 * math.sqrt(2.0 / 256)
    assert float(fw.bias.detach().abs().max()) == 0.0
    # the positional table is the reference's sin / cos buffer, not a parameter
    pos = m.transformer_encoder.pos_emb.pe
    assert pos.shape == (1, 5000, a.embedding_size) and not pos.requires_grad
    assert abs(float(pos[0, 1, 0]) - math.sin(1.0)) < 1e-6 and abs(float(pos[0, 1, 1]) - math.cos(1.0)) < 1e-6


def test_rtm_default_init_matches_the_reference_distributions():
    torch.manual_seed(4)
    a = default_args(model_name='review_transformer', review_encoder_name='pv', embedding_size=64, heads=4, ff_size=128,
                     inter_layers=2, use_user_emb=True, use_item_emb=True)
    V, RC = 3000, 2000
    rw = torch.full((RC, 5), V - 1, dtype=torch.int64)
    m = ProductRanker(a, 'cpu', V, RC, 50, 40, rw, None, word_dists=None)
    for t in (m.word_embeddings.weight, m.review_encoder.review_embeddings.weight):
        mu, sd = _moments(t)
        assert abs(mu) < 0.02 and abs(sd - 1.0) < 0.02
    assert float(m.word_embeddings.weight.detach()[V - 1].abs().max()) > 0.0
    assert float(m.review_encoder.review_embeddings.weight.detach()[RC - 1].abs().max()) > 0.0     # PV.py:82-90 normal_
    assert float(m.user_emb.weight.detach()[40].abs().max()) == 0.0                                  # nn.Embedding defaults
    assert float(m.product_emb.weight.detach()[50].abs().max()) == 0.0
    _check_encoder(m.transformer_encoder, 64, 128)
