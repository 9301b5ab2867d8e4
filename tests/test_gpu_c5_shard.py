"""One GPU's shard of BASELINE.json configs[4] as a parity test: item_transformer d=256, ff=1024, bs=1024, K=20,
row-sparse Adam (``args.row_sparse_adam``), and an item table PAST 2^31 bytes (8,000,001 rows x 1 KiB = 8.2 GB), so every
row offset of the gather / scatter / optimizer kernels needs its 64-bit form.  Reference semantics: ``forward_dotproduct``
(models/item_transformer.py:440-520), ``trainer.py:74-79``.

The full table never goes to the host: the oracle runs on the COMPACTED table (the rows the batch addresses, remapped),
which is the same arithmetic — an embedding lookup only ever sees the rows it indexes."""
import numpy as np
import pytest
import torch

from golden_util import rel_err

pytestmark = pytest.mark.gpu

P_, V, B, K, L, Q, W, D, FF = 8_000_000, 32387, 1024, 20, 20, 8, 1, 256, 1024


def _setup(dropout, P_=P_):
    from prodsearch_amd import ItemTransformerRanker, build_optim, readme_tem_args, synth
    a = readme_tem_args(dropout=dropout, embedding_size=D, ff_size=FF, row_sparse_adam=True, lr=0.002)
    wd = synth.make_word_dists(V)
    torch.manual_seed(5)
    m = ItemTransformerRanker(a, 'cuda', V, P_, None, word_dists=wd)
    assert m.product_emb.weight.numel() * 4 > 2 ** 31
    optim = build_optim(a, m, None)
    batch = synth.make_tem_batch(77, B, P_, V, Q=Q, L=L, W=W, word_dists=wd)
    # make sure rows at both ends of the table (offsets beyond 2^31 bytes) are addressed
    batch.target_prod_idxs[:3] = torch.tensor([P_ - 1, P_ - 2, 0])
    batch.u_item_idxs[5, 0] = P_ - 3
    ni, nw = synth.sample_negatives(78, B, K, W, P_, wd)
    ni[0, 0] = P_ - 4
    m.train()
    return a, wd, m, optim, batch, ni, nw


def _compact(m, batch, ni, P_=P_):
    """(compact state dict on the host, remapped batch / negatives, sorted unique item rows)."""
    import copy
    items = torch.unique(torch.cat([batch.target_prod_idxs.reshape(-1), ni.reshape(-1), batch.u_item_idxs.reshape(-1)]))
    items = items[items != P_]
    U = items.numel()
    remap = lambda t: torch.where(t == P_, torch.full_like(t, U), torch.searchsorted(items, t.clamp(max=P_ - 1)))
    sd = {}
    for k, v in m.state_dict().items():
        if k == 'product_emb.weight':
            sd[k] = torch.cat([v[items.cuda()].cpu(), torch.zeros(1, D)], 0)
        elif k == 'product_bias':
            sd[k] = torch.cat([v[items.cuda()].cpu(), torch.zeros(1)], 0)
        else:
            sd[k] = v.detach().cpu().clone()
    b2 = copy.copy(batch)
    b2.target_prod_idxs = remap(batch.target_prod_idxs)
    b2.u_item_idxs = remap(batch.u_item_idxs)
    return sd, b2, remap(ni), items


@pytest.mark.parametrize('P_', [8_000_000, 50_000_000])
def test_c5_shard_step_matches_the_oracle_on_the_gathered_rows(P_):
    """8 M rows: offsets past 2^31 bytes.  50 M rows: configs[4]'s stated table — 51 GB of parameters, 205 GB resident with
    the dense gradient and both Adam moments, a 6.25 MB coalesce bitmap, row offsets past 2^35 bytes."""
    from oracle import tem as otem
    if P_ > 10_000_000 and torch.cuda.get_device_properties(0).total_memory < 240e9:
        pytest.skip("needs the 288 GB of an MI355X")
    torch.cuda.empty_cache()
    a, wd, m, optim, batch, ni, nw = _setup(0.0, P_)
    sd, b2, ni2, items = _compact(m, batch, ni, P_)
    U = items.numel()
    before_rows = m.product_emb.weight.detach()[items.cuda()].clone()
    probe = torch.tensor([1, 12345, P_ // 2, P_ - 5], device='cuda')          # rows no index of the step addresses
    probe = probe[~torch.isin(probe, items.cuda())]
    before_probe = m.product_emb.weight.detach()[probe].clone()
    loss = m(batch.to('cuda'), neg_item_idxs=ni.cuda(), neg_word_idxs=nw.cuda())
    m.zero_grad()
    loss.backward()
    # oracle on the compact table (dropout 0: one encode per row is exact)
    Pm = {k: v.clone().requires_grad_(v.dtype.is_floating_point and not k.endswith('pos_emb.pe')) for k, v in sd.items()}
    oloss, ops, oil = otem.tem_forward(Pm, a, b2, ni2, nw, V, U, training=True)
    assert rel_err(loss.detach().cpu(), oloss.detach()) < 1e-4
    grads = otem.grads_of(oloss, Pm, otem.tem_pad_rows(a, V, U))
    # bit-exact index work: the touched lists are the batch's index sets
    touched = m.touched_rows()
    assert torch.equal(touched['product_emb.weight'].cpu(), items)
    words = np.setdiff1d(np.unique(np.concatenate([batch.query_word_idxs.numpy().ravel(), nw.numpy().ravel(),
                                                   batch.pos_iword_idxs.numpy().ravel()])), [V - 1])
    assert np.array_equal(touched['word_embeddings.weight'].cpu().numpy(), words)
    # gradients: table rows on the touched list vs the oracle's compact-table gradient; dense tensors whole
    g_items = m.product_emb.weight.grad[items.cuda()].cpu()
    assert rel_err(g_items, grads['product_emb.weight'][:U]) < 5e-4
    assert rel_err(m.word_embeddings.weight.grad.cpu(), grads['word_embeddings.weight']) < 5e-4
    for n, p in m.named_parameters():
        if p.grad is None or n in ('product_emb.weight', 'word_embeddings.weight') or n.endswith('linear_keys.bias'):
            continue
        assert rel_err(p.grad.cpu(), grads[n]) < 5e-4, n
    # nothing outside the touched rows received a gradient (checked on the device: the table is 8 GB)
    nz = m.product_emb.weight.grad.ne(0).any(1)
    assert int(nz.sum()) == int(nz[items.cuda()].sum())
    had_grad = nz[items.cuda()].clone()        # (a history row whose attention weights underflow has an all-zero gradient)
    assert float(had_grad.float().mean()) > 0.99
    optim.step()
    torch.cuda.synchronize()
    after_rows = m.product_emb.weight.detach()[items.cuda()]
    moved = (after_rows != before_rows).any(1)
    assert bool((moved <= had_grad).all()) and float(moved.float().mean()) > 0.9    # rows with a (non-vanishing) gradient moved ...
    assert torch.equal(m.product_emb.weight.detach()[probe], before_probe)       # ... untouched rows did not
    assert float(m.product_emb.weight.grad[items.cuda()].abs().max()) == 0       # touched gradient rows come back zeroed
    m.check_index_errors()
    if P_ > 10_000_000:
        # full-catalogue ranking at this size (rank_stream_kernel: the table streamed once): top-k and the target's rank
        # against a chunked torch product over the same rows
        from prodsearch_amd import evaluate
        import copy
        eb = copy.copy(batch)
        for k in ('query_word_idxs', 'target_prod_idxs', 'u_item_idxs', 'pos_iword_idxs'):
            setattr(eb, k, getattr(batch, k)[:24].contiguous().cuda())
        top_idx, top_score, rank = evaluate.rank_all(m, eb, topk=100)
        enc = m.encode(eb)
        best_s = torch.full((24, 100), -float('inf'), device='cuda')
        best_i = torch.zeros(24, 100, dtype=torch.int64, device='cuda')
        ahead = torch.zeros(24, dtype=torch.int64, device='cuda')
        tgt = eb.target_prod_idxs
        w = m.product_emb.weight.detach()
        tscore = (enc * w[tgt]).sum(-1)
        for lo in range(0, P_, 1 << 22):
            sc = enc @ w[lo:min(lo + (1 << 22), P_)].t()
            ahead += (sc > tscore[:, None]).sum(1)
            cs, ci = torch.cat([best_s, sc], 1).topk(100, dim=1)
            best_i = torch.gather(torch.cat([best_i, torch.arange(lo, lo + sc.shape[1], device='cuda')[None].expand(24, -1)], 1), 1, ci)
            best_s = cs
        assert rel_err(top_score.cpu(), best_s.cpu()) < 2e-4
        agree = (top_idx == best_i).float().mean()
        assert float(agree) > 0.98                                   # (scores closer than the fp32 tolerance may swap places)
        # the target sits among ~10^6 products per unit of score: a reference score that differs in its last bits (torch's
        # summation order against the MFMA chain's) moves the count by its density times that difference
        assert int((rank.long() - (ahead + 1)).abs().max()) <= max(2, P_ // 200_000)
    del m, optim
    torch.cuda.empty_cache()


def test_c5_shard_dropout_step_is_deterministic_and_moves_only_touched_rows():
    """reference default dropout 0.1: the K+1 replicas are really computed (R = 21, 21,504 replica rows)."""
    res = []
    for rep in range(2):
        a, wd, m, optim, batch, ni, nw = _setup(0.1)
        before = m.product_emb.weight.detach().clone()
        losses = []
        for _ in range(2):
            loss = m(batch.to('cuda'), neg_item_idxs=ni.cuda(), neg_word_idxs=nw.cuda())
            m.zero_grad()
            loss.backward()
            optim.step()
            losses.append(float(loss.detach()))
        assert all(np.isfinite(losses))
        moved = torch.nonzero((m.product_emb.weight.detach() != before).any(1)).flatten().cpu()
        items = torch.unique(torch.cat([batch.target_prod_idxs.reshape(-1), ni.reshape(-1), batch.u_item_idxs.reshape(-1)]))
        items = items[items != P_]
        assert bool(torch.isin(moved, items).all())                              # nothing outside the touched rows moved
        assert moved.numel() > 0.99 * items.numel()                              # (all-zero gradient rows stay put)
        res.append(losses)
        del m, optim, before
        torch.cuda.empty_cache()
    assert res[0][0] == res[1][0]          # forward of step 1: same seed, same Philox step -> identical bits


def test_c5_shard_dropout_step_matches_the_replicated_oracle():
    """The shard step bench.py's `also` line times (BASELINE configs[4] with the reference's default dropout 0.1: B = 1024,
    K = 20, d = 256, ff = 1024, R = 21 replicas = 21,504 replica rows) against the oracle in the reference's own replicated
    structure (item_transformer.py:471-494: the encoder on B and on B*K expanded copies) on the compacted table, with the
    product's Philox masks injected: loss / ps / item loss and the [1024, 21] logits <= 1e-4, the gradient of every tensor
    <= 5e-4 of its max, touched rows of both tables bit-exact.  Only at this size does the step select the 128x128
    direct-to-LDS weight gradients (gemm_x3d_kernel, tem.hip `t128 * ks3 >= 384`), fanin_sum_kernel and the row-list K/V dX
    product; the oracle side is ~1 minute and ~20 GB of host memory."""
    from oracle import tem as otem, philox
    a, wd, m, optim, batch, ni, nw = _setup(0.1)
    sd, b2, ni2, items = _compact(m, batch, ni)
    U = items.numel()
    loss = m(batch.to('cuda'), neg_item_idxs=ni.cuda(), neg_word_idxs=nw.cuda())
    m.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    plan = next(iter(m._plans.values()))
    assert plan.layout.R == K + 1                                   # the replicas really are computed
    scores = m.workspace_view(plan, 'item_scores', (B, K + 1)).cpu()
    Pm = {k: v.clone().requires_grad_(v.dtype.is_floating_point and not k.endswith('pos_emb.pe')) for k, v in sd.items()}
    keep = {}
    drop = philox.PhiloxDropout(0.1, m._seed, m._fwd_step, B, K, a.heads, L + 1, 1, L if a.use_item_pos else 0)
    oloss, ops, oil = otem.tem_forward(Pm, a, b2, ni2, nw, V, U, training=True, replicate=True, drop=drop, keep=keep)
    assert rel_err(loss.detach().cpu(), oloss.detach()) < 1e-4
    assert abs(m.ps_loss - float(ops)) < 1e-4 * abs(float(ops)) and abs(m.item_loss - float(oil)) < 1e-4 * abs(float(oil))
    ref_scores = torch.cat([keep['pos_scores'].detach()[:, None], keep['neg_scores'].detach()], 1)
    assert rel_err(scores, ref_scores) < 1e-4
    grads = otem.grads_of(oloss, Pm, otem.tem_pad_rows(a, V, U))
    del keep, oloss
    # the item table: the touched list is the batch's index set, its gradient rows against the compact table's
    assert torch.equal(m.touched_rows()['product_emb.weight'].cpu(), items)
    g_items = m.product_emb.weight.grad[items.cuda()].cpu()
    ref_items = grads['product_emb.weight'][:U]
    assert rel_err(g_items, ref_items) < 5e-4, rel_err(g_items, ref_items)
    assert torch.equal(g_items.ne(0).any(1), ref_items.ne(0).any(1))
    nz = m.product_emb.weight.grad.ne(0).any(1)
    assert int(nz.sum()) == int(nz[items.cuda()].sum())             # nothing outside the touched rows received a gradient
    for n, p in m.named_parameters():
        ref = grads.get(n)
        if n == 'product_emb.weight':
            continue
        assert (p.grad is None) == (ref is None), n
        if ref is None or n.endswith('linear_keys.bias'):          # (exactly-zero true gradient: rounding noise on both sides)
            continue
        got = p.grad.cpu()
        assert rel_err(got, ref) < 5e-4, (n, rel_err(got, ref))
        if ref.dim() == 2 and ref.shape[0] > 1024:
            assert torch.equal(got.ne(0).any(1), ref.ne(0).any(1)), n
    m.check_index_errors()
    del m, optim
    torch.cuda.empty_cache()


def test_gather_score_launch_in_its_8_lane_form_matches_a_host_dot_product():
    """B = 8192: 344k tasks x 1 KiB = 352 MB of rows per launch selects the 8-lanes-per-row form of
    score_fwd_wide_kernel (rowwise.hip: `huge`); logits against a host fp32 dot product of the same rows."""
    from prodsearch_amd import ItemTransformerRanker, _lib, readme_tem_args, synth
    Bh, Ph = 8192, 3_000_000                                                     # 3.07 GB table: past 2^31 bytes too
    a = readme_tem_args(dropout=0.0, embedding_size=D, ff_size=FF)
    wd = synth.make_word_dists(V)
    torch.manual_seed(9)
    m = ItemTransformerRanker(a, 'cuda', V, Ph, None, word_dists=wd)
    m.train()
    batch = synth.make_tem_batch(3, Bh, Ph, V, Q=Q, L=L, W=W, word_dists=wd)
    batch.target_prod_idxs[0] = Ph - 1
    ni, nw = synth.sample_negatives(4, Bh, K, W, Ph, wd)
    with torch.no_grad():
        m(batch.to('cuda'), neg_item_idxs=ni.cuda(), neg_word_idxs=nw.cuda())
    plan = next(iter(m._plans.values()))
    lay = plan.layout
    enc = torch.randn(Bh, D, device='cuda')
    m.workspace_view(plan, 'enc', (Bh, D)).copy_(enc)
    ps, _ = m._structs()
    _lib.check(_lib.load().ps_gather_score(plan.desc, ps, plan.batch, plan.ws.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream), 'ps_gather_score')
    torch.cuda.synchronize()
    got = m.workspace_view(plan, 'item_scores', (Bh, K + 1)).cpu()
    idx = torch.cat([batch.target_prod_idxs[:, None], ni], 1)
    rows = m.product_emb.weight.detach()[idx.cuda()].cpu()                      # [Bh, K+1, D]
    want = (rows * enc.cpu()[:, None, :]).sum(-1)
    assert rel_err(got, want) < 1e-4
    gw = m.workspace_view(plan, 'word_scores', (Bh, W, K + 1)).cpu()
    widx = torch.cat([batch.pos_iword_idxs[:, :, None], nw.view(Bh, W, K)], 2)
    tgt_rows = m.product_emb.weight.detach()[batch.target_prod_idxs.cuda()].cpu()
    wrows = m.word_embeddings.weight.detach().cpu()[widx]
    wb = m.word_bias.detach().cpu()[widx]
    wantw = (wrows * tgt_rows[:, None, None, :]).sum(-1) + wb
    assert rel_err(gw, wantw) < 1e-4
