"""Row-sharded item table (prodsearch_amd/sharded.py) through the HIP step: the compact table + remapped
indices that ``ShardedTable.lookup`` hands to the kernels give the same loss and, routed back by ``push_grads``, the same
table gradient as the replicated table (item_transformer.py:464-469 only ever reads the rows a batch indexes)."""
import copy

import pytest
import torch

from golden_util import rel_err

pytestmark = pytest.mark.gpu


def test_tem_step_on_a_sharded_item_table_equals_the_replicated_step():
    from prodsearch_amd import ItemTransformerRanker, readme_tem_args, synth
    from prodsearch_amd.sharded import ShardedTable
    P_, V, B, K, L = 20000, 5000, 96, 20, 20
    a = readme_tem_args(dropout=0.0)
    wd = synth.make_word_dists(V)
    torch.manual_seed(2)
    full = ItemTransformerRanker(a, 'cuda', V, P_, None, word_dists=wd)
    full.train()
    batch = synth.make_tem_batch(7, B, P_, V, Q=8, L=L, W=1, word_dists=wd)
    ni, nw = synth.sample_negatives(8, B, K, 1, P_, wd)
    loss = full(batch.to('cuda'), neg_item_idxs=ni.cuda(), neg_word_idxs=nw.cuda())
    full.zero_grad(); loss.backward()
    # the same step through a sharded table (one process = one shard holding every row) and a compact-capacity model
    tab = ShardedTable(P_, a.embedding_size, P_, device='cuda')
    tab.load_full(full.product_emb.weight.detach())
    cap = B * (1 + K + L)
    small = ItemTransformerRanker(a, 'cuda', V, cap, None, word_dists=wd)
    sd = {k: v for k, v in full.state_dict().items() if not k.startswith('product_')}
    small.load_state_dict(sd, strict=False)
    small.train()
    b = batch.to('cuda')
    compact, (tgt, hist, neg), ctx = tab.lookup([b.target_prod_idxs, b.u_item_idxs, ni.cuda()])
    U = ctx['U']
    # lookup's pad id is U (the compact table's last row); the model built for `cap` rows pads with `cap`
    fix = lambda t: torch.where(t == U, torch.full_like(t, cap), t)
    with torch.no_grad():
        small.product_emb.weight.zero_()
        small.product_emb.weight[:U].copy_(compact[:U])
    b2 = copy.copy(b)
    b2.target_prod_idxs, b2.u_item_idxs = fix(tgt), fix(hist)
    loss2 = small(b2, neg_item_idxs=fix(neg), neg_word_idxs=nw.cuda())
    small.zero_grad(); loss2.backward()
    torch.cuda.synchronize()
    assert rel_err(loss2.detach().cpu(), loss.detach().cpu()) < 1e-6
    g = torch.cat([small.product_emb.weight.grad[:U], torch.zeros(1, a.embedding_size, device='cuda')], 0)
    touched = tab.push_grads(ctx, g)
    ref = full.product_emb.weight.grad[:P_]
    assert rel_err(tab.grad.cpu(), ref.cpu()) < 5e-4                       # (fp32 atomics reassociate)
    assert torch.equal(touched.cpu(), torch.nonzero(ref.ne(0).any(1)).flatten().cpu()) or \
        set(touched.tolist()) >= set(torch.nonzero(ref.ne(0).any(1)).flatten().tolist())
    for n, p in small.named_parameters():
        if p.grad is not None and not n.startswith('product_') and not n.endswith('linear_keys.bias'):
            assert rel_err(p.grad.cpu(), dict(full.named_parameters())[n].grad.cpu()) < 5e-4, n
