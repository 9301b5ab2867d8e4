"""GPU parity at shapes no golden fixture covers (oracle-driven, synthetic): d=256 / ff=1024 — the shape of
BASELINE configs[4] — and d=512, both through the module API.  Tolerances as test_gpu_parity.py."""
import pytest
import torch

from golden_util import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('d,F,B,K,dropout', [(256, 1024, 24, 6, 0.0), (512, 1024, 9, 4, 0.0), (256, 1024, 16, 5, 0.1),
                                             (256, 1024, 70, 20, 0.1),   # 1,470 replica rows at d = 256: the replicas' fan-in summed by its own launch, the K/V dX product over the row list (round 4)
                                             (512, 1024, 300, 3, 0.0)])   # 6,300 K/V rows, few splits: row-list weight gradients remap their split count
def test_wide_embeddings_match_oracle(d, F, B, K, dropout):
    from oracle import tem as otem, philox
    from prodsearch_amd import ItemTransformerRanker, default_args, synth
    P_, V = 3000, 2000
    a = default_args(model_name='item_transformer', embedding_size=d, ff_size=F, heads=8, inter_layers=1,
                     neg_per_pos=K, dropout=dropout, uprev_review_limit=20)
    wd = synth.make_word_dists(V)
    sd = synth.make_state_dict(synth.tem_param_shapes(a, V, P_), 5, {'product_emb.weight': P_})
    m = ItemTransformerRanker(a, 'cuda', V, P_, None, word_dists=wd)
    m.load_state_dict(sd, strict=False)
    m.train()
    batch = synth.make_tem_batch(3, B, P_, V, Q=8, L=20, W=1, word_dists=wd)
    ni, nw = synth.sample_negatives(4, B, K, 1, P_, wd)
    loss = m(batch.to('cuda'), neg_item_idxs=ni.cuda(), neg_word_idxs=nw.cuda())
    m.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    Pm = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    kw = {}
    if dropout > 0:
        kw = dict(replicate=True, drop=philox.PhiloxDropout(dropout, m._seed, m._fwd_step, B, K, 8, 21, 1,
                                                            20 if a.use_item_pos else 0))
    oloss, _, _ = otem.tem_forward(Pm, a, batch, ni, nw, V, P_, training=True, **kw)
    assert rel_err(loss.detach().cpu(), oloss.detach()) < 1e-4
    grads = otem.grads_of(oloss, Pm, otem.tem_pad_rows(a, V, P_))
    for n, p in m.named_parameters():
        ref = grads.get(n)
        assert (p.grad is None) == (ref is None), n
        if ref is None:
            continue
        got = p.grad.cpu()
        if n.endswith('linear_keys.bias'):
            continue
        assert rel_err(got, ref) < 5e-4, n
        if ref.dim() == 2 and ref.shape[0] > 256:
            assert torch.equal(got.ne(0).any(1), ref.ne(0).any(1)), n


@pytest.mark.parametrize('B,K,L,Q,W,d,H,zero_hist,dropout', [
    (1, 1, 1, 1, 1, 32, 1, 0.0, 0.0),        # smallest legal everything
    (3, 2, 20, 8, 1, 32, 4, 1.0, 0.0),       # every user without history: only the query column is unmasked
    (5, 33, 7, 3, 2, 64, 8, 0.3, 0.0),       # K+1 above the 32 row groups of the score kernels, pv window 2
    (130, 3, 12, 5, 1, 96, 8, 0.1, 0.1),     # ragged batch (not a multiple of any tile), d not a power of two, dropout
    (2, 5, 20, 8, 3, 128, 8, 0.0, 0.1),
    (50, 20, 9, 4, 1, 128, 8, 0.2, 0.1),     # 1,050 replica rows: the fused per-replica kernels with a partial last tile
    (400, 20, 9, 4, 1, 128, 8, 0.2, 0.1),    # 8,400 replica rows = 263 workgroups: past the 256 round 3 stopped the fused / folded forms at
    # the attention forms (attn_sq1.hip): replicas inside a wave (d = 128, 8 heads, 4..24 replicas, <= 32 positions) ...
    (6, 23, 29, 4, 1, 128, 8, 0.2, 0.1),     # 30 positions (8 key steps), 24 replicas (chunks of 6)
    (9, 3, 31, 4, 1, 128, 8, 0.1, 0.1),      # 32 positions, 4 replicas (one per wave)
    (7, 2, 12, 4, 1, 128, 8, 0.1, 0.1),      # ... 3 replicas: the LDS workgroup form
    (6, 4, 40, 4, 1, 128, 8, 0.1, 0.1),      # ... 41 positions with dropout: the LDS workgroup form
    (6, 4, 40, 4, 1, 128, 8, 0.1, 0.0),      # ... no replicas: one wave per sequence, 32 lanes per row
    (6, 4, 40, 4, 1, 64, 4, 0.1, 0.0),       # ... and 16 lanes per row (d = 64)
])
def test_edge_shapes_match_oracle(B, K, L, Q, W, d, H, zero_hist, dropout):
    """Edge cases of the batch layout: tiny and ragged batches, zero-history users,
    padded pv windows, wide negative fans — loss, gradients, touched rows and eval scores against the oracle."""
    from oracle import tem as otem, philox
    from prodsearch_amd import ItemTransformerRanker, default_args, synth
    P_, V = 700, 900
    a = default_args(model_name='item_transformer', embedding_size=d, ff_size=2 * d, heads=H, inter_layers=1,
                     neg_per_pos=K, dropout=dropout, uprev_review_limit=L, pv_window_size=W)
    wd = synth.make_word_dists(V)
    sd = synth.make_state_dict(synth.tem_param_shapes(a, V, P_), 7, {'product_emb.weight': P_})
    m = ItemTransformerRanker(a, 'cuda', V, P_, None, word_dists=wd)
    m.load_state_dict(sd, strict=False)
    m.train()
    batch = synth.make_tem_batch(11, B, P_, V, Q=Q, L=L, W=W, C=9, word_dists=wd, zero_hist_frac=zero_hist)
    if zero_hist >= 1.0:
        assert int((batch.u_item_idxs != P_).sum()) == 0
    ni, nw = synth.sample_negatives(12, B, K, W, P_, wd)
    loss = m(batch.to('cuda'), neg_item_idxs=ni.cuda(), neg_word_idxs=nw.cuda())
    m.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    Pm = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    kw = {}
    if dropout > 0:
        kw = dict(replicate=True, drop=philox.PhiloxDropout(dropout, m._seed, m._fwd_step, B, K, H, L + 1, 1,
                                                            L if a.use_item_pos else 0))
    oloss, _, _ = otem.tem_forward(Pm, a, batch, ni, nw, V, P_, training=True, **kw)
    assert rel_err(loss.detach().cpu(), oloss.detach()) < 1e-4
    grads = otem.grads_of(oloss, Pm, otem.tem_pad_rows(a, V, P_))
    for n, p in m.named_parameters():
        ref = grads.get(n)
        assert (p.grad is None) == (ref is None), n
        if ref is None or n.endswith('linear_keys.bias'):
            continue
        got = p.grad.cpu()
        scale = float(ref.abs().max())
        if scale == 0.0:                       # e.g. history-only parameters when nobody has a history
            assert float(got.abs().max()) < 1e-6, n
            continue
        assert rel_err(got, ref) < 5e-4, n
        if ref.dim() == 2 and ref.shape[0] > 256:
            assert torch.equal(got.ne(0).any(1), ref.ne(0).any(1)), n
    m.eval()
    with torch.no_grad():
        s = m.test(batch.to('cuda')).cpu()
        ref_s = otem.tem_test(sd, a, batch, V, P_)
    assert rel_err(s, ref_s) < 1e-4


@pytest.mark.parametrize('B,K,L,Q,W,zero_hist', [(7, 4, 20, 8, 1, 0.0), (2, 5, 20, 8, 3, 0.0), (33, 1, 5, 3, 1, 0.5)])
def test_fused_backward_partial_tiles(B, K, L, Q, W, zero_hist):
    """The fused per-replica backward (32-row tiles) forced onto tiny batches: 35, 12 and 66 replica rows."""
    from prodsearch_amd import _lib
    old = _lib.load().ps_set_fuse_bwd_min(1)
    try:
        test_edge_shapes_match_oracle(B, K, L, Q, W, 128, 8, zero_hist, 0.1)
    finally:
        _lib.load().ps_set_fuse_bwd_min(old)



def test_pre_split_weight_products_inside_the_step():
    """``ps_gemm_x3_config(1, 4)``: the d = 256 step's forward / input-gradient products multiply against weight planes split once
    per entry-point call (WPlaneScope + gemm_x3w_kernel, csrc/gemm.hip) — exercised where the kernel really runs (>= 4,096 replica
    rows: B = 200, K = 20) against the replicated oracle: loss and every gradient, K-concatenated [dK | dV] product and the
    transposed planes of the dX products included (models/neural.py:30-33, 86-96, 192-231)."""
    from prodsearch_amd import _lib
    lib = _lib.load()
    lib.ps_gemm_x3_config(1, 4)
    try:
        test_wide_embeddings_match_oracle(256, 1024, 200, 20, 0.1)
    finally:
        lib.ps_gemm_x3_config(0 if __import__('os').environ.get('PS_GEMM_X3') == '0' else 1,
                              int(__import__('os').environ.get('PS_GEMM_X3_SHAPE', '-1')))
