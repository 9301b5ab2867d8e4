"""Row-sharded item table END TO END (prodsearch_amd/sharded.py; ``args.shard_tables``): lookup (fixed-
capacity device-side all-to-all) -> the unchanged HIP step on the receive buffer -> push_grads -> the owners' row-sparse
clip + Adam on their shards -> the next step reads the updated rows.  Checked against the ORACLE (oracle/tem.py forward and
autograd, oracle/optim.py with its ``touched=`` rule) and against the single-process replicated row-sparse run, in one
process (world 1) and as two ranks (gloo, both on cuda:0): an embedding lookup only ever sees the rows it indexes
(item_transformer.py:464-469), so the parameters after three steps must agree."""
import os
import sys

import numpy as np
import pytest
import torch

from golden_util import rel_err

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_sharded_step_matches_the_oracle_and_the_replicated_row_sparse_step():
    from prodsearch_amd import ItemTransformerRanker, build_optim, readme_tem_args, synth
    from oracle import tem as otem, optim as ooptim
    P_, V, B, K, L, steps = 6000, 3000, 64, 20, 20, 3
    wd = synth.make_word_dists(V)
    runs = {}
    a0 = readme_tem_args(dropout=0.0, lr=0.002, row_sparse_adam=True, batch_size=B)
    sd0 = synth.make_state_dict(synth.tem_param_shapes(a0, V, P_), 123, {'product_emb.weight': P_})
    batches = [synth.make_tem_batch(40 + s, B, P_, V, Q=8, L=L, W=1, word_dists=wd) for s in range(steps)]
    negs = [synth.sample_negatives(80 + s, B, K, 1, P_, wd) for s in range(steps)]
    for mode in ('replicated', 'sharded'):
        a = readme_tem_args(dropout=0.0, lr=0.002, row_sparse_adam=True, shard_tables=(mode == 'sharded'), batch_size=B)
        m = ItemTransformerRanker(a, 'cuda', V, P_, None, word_dists=wd)
        m.load_state_dict(sd0, strict=False)
        optim = build_optim(a, m, None)
        m.train()
        losses = []
        for s in range(steps):
            ni, nw = negs[s]
            loss = m(batches[s].to('cuda'), neg_item_idxs=ni.cuda(), neg_word_idxs=nw.cuda())
            m.zero_grad()
            loss.backward()
            optim.step()
            losses.append(float(loss.detach()))
        m.check_index_errors()
        runs[mode] = (losses, {k: v.detach().cpu() for k, v in m.state_dict().items()})
        if mode == 'sharded':
            assert m.product_emb.weight.shape[0] == B * (1 + K + L) + 1          # the receive buffer, not the catalogue
            assert m._shard.weight.shape[0] == P_
    # oracle: the same three steps on the host (row-sparse rule for the two tables)
    Pm = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    opt = ooptim.ClipAdam(a0.lr, a0.max_grad_norm, a0.beta1, a0.beta2, 1e-9, a0.l2_lambda)
    pad = otem.tem_pad_rows(a0, V, P_)
    olosses = []
    for s in range(steps):
        ni, nw = negs[s]
        loss, _, _ = otem.tem_forward(Pm, a0, batches[s], ni, nw, V, P_, training=True, replicate=False)
        grads = otem.grads_of(loss, Pm, pad)
        touched = {n: torch.nonzero(grads[n].ne(0).any(1)).flatten() for n in ('product_emb.weight', 'word_embeddings.weight')}
        with torch.no_grad():
            opt.step(Pm, grads, touched=touched)
        olosses.append(float(loss.detach()))
    ls, ps = runs['sharded']
    lr_, pr = runs['replicated']
    assert np.allclose(ls, olosses, rtol=2e-4) and np.allclose(ls, lr_, rtol=1e-5)
    checked = 0
    for n, ref in Pm.items():
        if n.endswith('linear_keys.bias') or n not in ps or float(ref.detach().abs().max()) == 0.0:
            continue
        ref = ref.detach()
        for got in (ps[n], pr[n]):
            tol = 2e-3 * float(ref.abs().max()) + 0.02 * a0.lr * steps
            assert got.shape == ref.shape and float((got - ref).abs().max()) < tol, (n, float((got - ref).abs().max()), tol)
        checked += 1
    assert checked >= 20
    # rows no step addressed did not move (the SparseAdam rule) — in the shards too
    moved = (ps['product_emb.weight'][:P_] != sd0['product_emb.weight'][:P_]).any(1)
    want = torch.zeros(P_, dtype=torch.bool)
    for s in range(steps):
        for t in (batches[s].target_prod_idxs, batches[s].u_item_idxs, negs[s][0]):
            v = t.reshape(-1)
            want[v[v != P_]] = True
    assert torch.equal(moved, want)


def test_two_ranks_with_a_sharded_item_table_equal_one_replicated_rank(tmp_path):
    sys.path.insert(0, os.path.join(HERE, 'helpers'))
    import dp_worker
    from test_gpu_dp import _run_ranks

    class _NoExchange(object):
        def __call__(self):
            return None

    lr, steps = 0.002, 3
    single = dp_worker.run('sparse', 384, steps, 0, 1, lambda m, o: _NoExchange())
    r0, r1 = _run_ranks('sharded', str(tmp_path / 'dp_sharded'), 2, steps)
    for k in r0:
        if not k.startswith('__'):
            assert np.array_equal(r0[k], r1[k]), "replicas diverged: " + k       # small tensors / word table: bitwise lock step
    checked = 0
    for k, ref in single.items():
        if k.startswith('__') or k.endswith('__sum') or k.endswith('linear_keys.bias'):
            continue
        got = r0[k]
        assert got.shape == ref.shape, k
        tol = 2e-3 * float(np.abs(ref).max()) + 0.02 * lr * steps
        assert float(np.abs(got - ref).max()) < tol, (k, float(np.abs(got - ref).max()), tol)
        checked += 1
    assert checked >= 20
    for k in ('product_emb.weight__sum', 'word_embeddings.weight__sum'):           # nothing moved outside the compared rows
        assert abs(r0[k] - single[k]) < 1e-3 * max(1.0, abs(single[k])) + 1.0, k
    assert abs(0.5 * (r0['__loss'] + r1['__loss']) - single['__loss']) < 2e-3 * abs(single['__loss'])
