"""GPU parity of the row-sparse optimizer path (ps_coalesce_rows / ps_clip_adam_rowsparse, SURVEY.md §8b/§8d
config 5) against the oracle's restatement (oracle/optim.py ``touched=``) and numpy for the index work."""
import numpy as np
import pytest
import torch

from golden_util import Golden, rel_err

pytestmark = pytest.mark.gpu


def _coalesce(lists, n_rows, pad, ws=None):
    from prodsearch_amd import _lib
    lib = _lib.load()
    dev = 'cuda'
    ts = [torch.as_tensor(x, dtype=torch.int64, device=dev) for x in lists]
    total = sum(t.numel() for t in ts)
    cap = max(1, min(total, n_rows))
    if ws is None:
        ws = torch.zeros(lib.ps_coalesce_ws_bytes(n_rows), dtype=torch.uint8, device=dev)
    rows = torch.full((cap,), -7, dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    arr = (_lib.PsIdxList * len(ts))()
    for i, t in enumerate(ts):
        arr[i].idx, arr[i].n = t.data_ptr(), t.numel()
    _lib.check(lib.ps_coalesce_rows(arr, len(ts), n_rows, pad, ws.data_ptr(), rows.data_ptr(), cap,
                                    count.data_ptr(), torch.cuda.current_stream().cuda_stream), 'ps_coalesce_rows')
    torch.cuda.synchronize()
    return rows[:int(count[0])].cpu().numpy(), ws


@pytest.mark.parametrize('n_rows,sizes', [(1001, (64, 320, 1280)), (18358, (384, 7680, 7680)), (70, (500,)),
                                          (5_000_001, (1024, 20480, 20480)), (131072, (0, 17))])
def test_coalesce_rows_is_sorted_unique(n_rows, sizes):
    rng = np.random.default_rng(n_rows)
    pad = n_rows - 1
    lists = []
    for s in sizes:
        x = rng.integers(0, n_rows, size=s)
        x[rng.random(s) < 0.2] = pad                       # padding entries are not rows
        if s > 8:
            x[:4] = [0, n_rows - 2, 63, 64]                # word boundaries, last real row
        lists.append(x)
    want = np.unique(np.concatenate(lists)) if lists else np.zeros(0, np.int64)
    want = want[want != pad]
    got, ws = _coalesce(lists, n_rows, pad)
    assert np.array_equal(got, want)                        # bit-exact index work
    assert int(ws[:8 * ((n_rows + 63) // 64)].to(torch.int32).sum()) == 0      # bitmap left clean
    got2, _ = _coalesce(lists[::-1], n_rows, pad, ws)       # reuse: same answer, any list order
    assert np.array_equal(got2, want)


def test_coalesce_rejects_out_of_range_rows_without_fault():
    got, ws = _coalesce([np.array([3, 5, 10_000_000, -2, 5])], 100, 99)
    assert list(got) == [3, 5]
    nwords = (100 + 63) // 64
    flag = ws.view(torch.int32)[2 * nwords + 1]             # bad-index flag after bitmap + 1 block sum
    assert int(flag) == 1


def test_gather_scatter_zero_rows_roundtrip():
    from prodsearch_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    tab = torch.randn(5000, 96, device='cuda')
    rows = torch.tensor(sorted(np.random.default_rng(1).choice(5000, 777, replace=False)), device='cuda')
    cnt = torch.tensor([700], dtype=torch.int32, device='cuda')          # only the first 700 are valid
    vals = torch.zeros(777, 96, device='cuda')
    _lib.check(lib.ps_gather_rows(tab.data_ptr(), 96, rows.data_ptr(), cnt.data_ptr(), 777, vals.data_ptr(), st), 'g')
    assert torch.equal(vals[:700], tab[rows[:700]]) and float(vals[700:].abs().max()) == 0
    tab2 = torch.zeros_like(tab)
    _lib.check(lib.ps_scatter_rows(tab2.data_ptr(), 96, rows.data_ptr(), cnt.data_ptr(), 777, vals.data_ptr(), st), 's')
    ref = torch.zeros_like(tab)
    ref[rows[:700]] = tab[rows[:700]]
    assert torch.equal(tab2, ref)
    _lib.check(lib.ps_zero_rows(tab2.data_ptr(), 96, rows.data_ptr(), cnt.data_ptr(), 777, st), 'z')
    assert float(tab2.abs().max()) == 0


def _touched_np(g, step, args):
    b = g.batch()
    ni, nw = g.negs(step)
    P, V = g.P, g.V
    prod = [b.target_prod_idxs, ni]
    hist = []
    if args.model_name == 'item_transformer':
        (hist if args.sep_prod_emb else prod).append(b.u_item_idxs)
    word = [b.query_word_idxs, b.pos_iword_idxs, nw]
    uniq = lambda ls, pad: np.setdiff1d(np.unique(np.concatenate([np.asarray(x).ravel() for x in ls])), [pad])
    out = {'product_emb.weight': uniq(prod, P), 'word_embeddings.weight': uniq(word, V - 1)}
    if args.sep_prod_emb:
        out['hist_product_emb.weight'] = uniq(hist, P) if hist else np.zeros(0, np.int64)
    return out


@pytest.mark.parametrize('case', ['tem_c1', 'tem_c2s', 'tem_opts', 'qem_c1', 'tem_l2'])
def test_rowsparse_training_matches_oracle(case):
    """trainer.py:74-79 order with args.row_sparse_adam: touched lists bit-exact, parameters after every
    step equal to the oracle's row-sparse ClipAdam driven by the oracle's own gradients."""
    import copy
    from oracle import tem as otem, optim as ooptim
    from prodsearch_amd import ItemTransformerRanker, build_optim
    g = Golden(case)
    if g.args.dropout > 0:
        pytest.skip('dropout-free cases only')
    a = copy.copy(g.args)
    a.row_sparse_adam = True
    torch.manual_seed(0)
    m = ItemTransformerRanker(a, 'cuda', g.V, g.P, None, word_dists=g.word_dists)
    m.load_state_dict(g.params(), strict=False)
    m.train()
    optim = build_optim(a, m, None)
    assert optim.row_sparse
    P = {k: v.clone().requires_grad_(True) for k, v in g.params().items()}
    init = {k: v.detach().clone() for k, v in P.items()}
    opt = ooptim.ClipAdam(a.lr, a.max_grad_norm, a.beta1, a.beta2, 1e-9, a.l2_lambda, a.decay_method, a.warmup_steps)
    pad = otem.tem_pad_rows(a, g.V, g.P)
    fn = otem.qem_forward if a.model_name == 'QEM' else otem.tem_forward
    b = g.batch().to('cuda')
    steps = max(3, g.steps)
    for step in range(steps):
        s = step % g.steps
        ni, nw = g.negs(s)
        loss = m(b, neg_item_idxs=ni.cuda(), neg_word_idxs=nw.cuda())
        m.zero_grad()
        loss.backward()
        touched = _touched_np(g, s, a)
        got_t = m.touched_rows()
        assert sorted(got_t) == sorted(touched)
        for n in touched:
            assert np.array_equal(got_t[n].cpu().numpy(), touched[n]), (step, n)      # bit-exact index work
        optim.step()
        for n, p in m.named_parameters():                   # touched gradient rows come back zeroed
            if n in touched:
                assert float(p.grad.abs().max()) == 0, (step, n)
        oloss, _, _ = fn(P, a, g.batch(), ni, nw, g.V, g.P, training=True)
        assert rel_err(loss.detach().cpu(), oloss.detach()) < 2e-4, step
        grads = otem.grads_of(oloss, P, pad)
        with torch.no_grad():
            opt.step(P, grads, touched=touched)
        assert abs(optim.last_grad_norm - opt.last_total_norm) < 1e-3 * opt.last_total_norm
        sd = m.state_dict()
        for n in P:
            got, ref = sd[n].cpu(), P[n].detach()
            if n.endswith('linear_keys.bias'):
                assert float((got - ref).abs().max()) <= 2.01 * a.lr * (step + 1), (step, n)
                continue
            if n in touched:                                # rows the optimizer moved = touched rows, exactly
                assert torch.equal((got != init[n]).any(1), (ref != init[n]).any(1)), (step, n)
            assert float((got - ref).abs().max()) < 2e-3 * float(ref.abs().max()) + 0.02 * a.lr * (step + 1), (step, n)
            # a gradient element of the order of Adam's eps (1e-9) turns fp32 rounding noise into a few % of lr;
            # a wrong sign or a missing term gives >= 1 (same bound as tests/test_gpu_parity.py)
            assert rel_err(got - init[n], ref - init[n]) < 1e-1, (step, n)


def test_rowsparse_zero_grad_without_step_cleans_touched_rows():
    import copy
    from prodsearch_amd import ItemTransformerRanker
    g = Golden('tem_c1')
    a = copy.copy(g.args)
    a.row_sparse_adam = True
    m = ItemTransformerRanker(a, 'cuda', g.V, g.P, None, word_dists=g.word_dists)
    m.load_state_dict(g.params(), strict=False)
    m.train()
    b = g.batch().to('cuda')
    ni, nw = g.negs(0)
    grads = []
    for _ in range(2):                                      # two backwards, zero_grad between, no optimizer
        loss = m(b, neg_item_idxs=ni.cuda(), neg_word_idxs=nw.cuda())
        m.zero_grad()
        loss.backward()
        grads.append({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    for n in grads[0]:
        assert rel_err(grads[1][n], grads[0][n]) < 1e-5, n  # not accumulated
    loss = m(b, neg_item_idxs=ni.cuda(), neg_word_idxs=nw.cuda())
    with pytest.raises(NotImplementedError):
        loss.backward()                                     # accumulation over dirty rows is refused


@pytest.mark.parametrize('world,d', [(2, 128), (3, 96), (8, 256)])
def test_pack_and_merge_rows_sum_the_ranks_in_rank_order(world, d):
    """Wire format + merge of the data-parallel row exchange (dist.SparseGradExchange) with the ranks simulated in one
    process: ps_pack_rows per rank, union by ps_coalesce_rows(pad -1), ps_merge_rows -> bitwise the rank-ordered sum."""
    from prodsearch_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    n_rows, cap = 70_000, 500
    rng = np.random.default_rng(world * 1000 + d)
    all_rows = torch.empty(world, cap, dtype=torch.int64, device='cuda')
    all_vals = torch.empty(world, cap, d, device='cuda')
    ref = torch.zeros(n_rows, d)
    union = set()
    for r in range(world):
        n = [0, 1, cap, 137, 333][r % 5] if r else 321
        rows = np.sort(rng.choice(n_rows, n, replace=False))
        rows[: min(n, 3)] = np.sort(np.array([0, 64, n_rows - 1])[: min(n, 3)])      # shared by every non-empty rank
        rows = np.unique(rows)
        n = len(rows)
        grad = torch.zeros(n_rows, d, device='cuda')
        vals = torch.randn(n, d)
        grad[torch.from_numpy(rows).cuda()] = vals.cuda()
        rows_dev = torch.full((cap,), 12345, dtype=torch.int64, device='cuda')       # garbage past count must not leak
        rows_dev[:n] = torch.from_numpy(rows).cuda()
        cnt = torch.tensor([n], dtype=torch.int32, device='cuda')
        _lib.check(lib.ps_pack_rows(grad.data_ptr(), d, rows_dev.data_ptr(), cnt.data_ptr(), cap,
                                    all_rows[r].data_ptr(), all_vals[r].data_ptr(), st), 'ps_pack_rows')
        ref.index_add_(0, torch.from_numpy(rows), vals)                              # (0 + r0) + r1 + ...: rank order
        union |= set(rows.tolist())
    torch.cuda.synchronize()
    assert all(int((all_rows[r] >= 0).sum()) == int((all_rows[r] != -1).sum()) for r in range(world))
    ucap = min(world * cap, n_rows)
    ws = torch.zeros(lib.ps_coalesce_ws_bytes(n_rows), dtype=torch.uint8, device='cuda')
    urows = torch.empty(ucap, dtype=torch.int64, device='cuda')
    ucount = torch.zeros(1, dtype=torch.int32, device='cuda')
    lst = (_lib.PsIdxList * 1)()
    lst[0].idx, lst[0].n = all_rows.data_ptr(), world * cap
    _lib.check(lib.ps_coalesce_rows(lst, 1, n_rows, -1, ws.data_ptr(), urows.data_ptr(), ucap, ucount.data_ptr(), st), 'co')
    grad = torch.full((n_rows, d), 7.0, device='cuda')                # merged rows are OVERWRITTEN, others untouched
    _lib.check(lib.ps_merge_rows(all_rows.data_ptr(), all_vals.data_ptr(), world, cap, d, grad.data_ptr(),
                                 urows.data_ptr(), ucount.data_ptr(), ucap, st), 'ps_merge_rows')
    torch.cuda.synchronize()
    want_union = np.array(sorted(union))
    assert np.array_equal(urows[:int(ucount[0])].cpu().numpy(), want_union)
    got = grad.cpu()
    assert torch.equal(got[torch.from_numpy(want_union)], ref[torch.from_numpy(want_union)])      # bitwise
    mask = torch.ones(n_rows, dtype=torch.bool)
    mask[torch.from_numpy(want_union)] = False
    assert bool((got[mask] == 7.0).all())
    addr = lib.ps_coalesce_bad_flag(ws.data_ptr(), n_rows)
    off = addr - ws.data_ptr()
    assert int(ws[off:off + 4].view(torch.int32)[0]) == 0


def test_index_errors_are_reported_when_asked():
    import copy
    from prodsearch_amd import ItemTransformerRanker
    g = Golden('tem_c1')
    a = copy.copy(g.args)
    a.row_sparse_adam = True
    m = ItemTransformerRanker(a, 'cuda', g.V, g.P, None, word_dists=g.word_dists)
    m.load_state_dict(g.params(), strict=False)
    m.train()
    b = g.batch().to('cuda')
    ni, nw = g.negs(0)
    loss = m(b, neg_item_idxs=ni.cuda(), neg_word_idxs=nw.cuda())
    m.zero_grad()
    loss.backward()
    m.check_index_errors()                                   # clean step: silent
    info = m.product_emb.weight._ps_rows
    lib = __import__('prodsearch_amd._lib', fromlist=['x']).load()
    off = lib.ps_coalesce_bad_flag(info['ws'].data_ptr(), m.product_emb.weight.shape[0]) - info['ws'].data_ptr()
    info['ws'][off:off + 4].view(torch.int32)[0] = 1         # what co_mark_kernel sets for an out-of-range index
    with pytest.raises(RuntimeError, match='outside'):
        m.check_index_errors()
    m.check_index_errors()                                   # the flag is cleared once reported
