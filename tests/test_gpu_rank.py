"""GPU parity of the full-catalogue evaluation: ps_tem_encode + ps_rank_all against the reference's
own Trainer.test numbers (tests/golden/rank_*.npz) and, for the multi-chunk / multi-panel selection, against numpy on
the very scores the device GEMM produced (index work: bit-exact)."""
import glob
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN_DIR, Golden, rel_err

pytestmark = pytest.mark.gpu
RANK_CASES = sorted(os.path.basename(f)[5:-4] for f in glob.glob(os.path.join(GOLDEN_DIR, 'rank_*.npz')))


def test_rank_cases_present():
    assert len(RANK_CASES) >= 5


@pytest.mark.parametrize('case', RANK_CASES)
def test_rank_all_matches_reference_trainer(case):
    from oracle import rank as orank
    from prodsearch_amd import ItemTransformerRanker, evaluate
    g = Golden(case)
    z = np.load(os.path.join(GOLDEN_DIR, 'rank_%s.npz' % case))
    m = ItemTransformerRanker(g.args, 'cuda', g.V, g.P, None, word_dists=g.word_dists)
    m.load_state_dict(g.params(), strict=False)
    m.eval()
    b = g.batch().to('cuda')
    top_idx, top_score, rank = evaluate.rank_all(m, b, topk=100)
    top_idx, top_score, rank = top_idx.cpu().numpy(), top_score.cpu().numpy(), rank.cpu().numpy()
    ref = z['scores']
    tol = 2e-4 * np.abs(ref).max()
    # scores of the returned ids are the reference's scores of those ids
    got_ref = np.take_along_axis(ref, top_idx, axis=1)
    assert np.abs(top_score - got_ref).max() < tol
    assert (np.diff(top_score, axis=1) <= 0).all()                       # sorted, best first
    # ranklist identical wherever the reference's neighbouring scores are separated by more than the fp tolerance
    rs = z['top_score']
    sep = np.abs(np.diff(rs, axis=1)) > 2 * tol
    same = top_idx == z['top_idx']
    assert (same[:, :-1] | ~sep)[:, 1:].all() or (same[:, 1:-1] | ~(sep[:, :-1] & sep[:, 1:])).all()
    # rank of the target: exact unless another score lies within the tolerance of the target's
    tgt = g.batch().target_prod_idxs.numpy()
    st = ref[np.arange(len(tgt)), tgt]
    close = (np.abs(ref - st[:, None]) < 2 * tol).sum(1) - 1
    assert (np.abs(rank - z['rank']) <= close).all()
    if (close == 0).all():
        mrr, p1 = evaluate.calc_metrics(rank, 100)
        assert abs(mrr - float(z['mrr'])) < 1e-12 and abs(p1 - float(z['p1'])) < 1e-12
    # the oracle agrees with the reference on the reference's scores (pins oracle/rank.py)
    otop, _, orank_ = orank.rank_scores(ref, tgt, 100)
    assert np.array_equal(orank_, z['rank'])
    assert np.array_equal(otop, z['top_idx']) or (np.take_along_axis(ref, otop, 1) == z['top_score']).all()


def _device_scores(q, table, bias):
    from prodsearch_amd import _lib
    lib = _lib.load()
    B, d = q.shape
    N = table.shape[0]
    S = torch.empty(B, N, device='cuda')
    lib.ps_gemm_x3_config(0, -1)       # the ranking kernel multiplies with the fp32 MFMA: bitwise comparisons need the same form
    try:
        _lib.check(lib.ps_gemm_f32(q.data_ptr(), d, 0, table.data_ptr(), d, 0, S.data_ptr(), N, B, N, d,
                                   _lib.ptr(bias), 1.0, 0, torch.cuda.current_stream().cuda_stream), 'gemm')
    finally:
        lib.ps_gemm_x3_config(0 if os.environ.get('PS_GEMM_X3') == '0' else 1, int(os.environ.get('PS_GEMM_X3_SHAPE', '-1')))
    return S


@pytest.mark.parametrize('B,N,d,k,use_bias', [(7, 300, 32, 100, False), (5, 20000, 64, 100, True), (3, 70001, 32, 10, False),
                                              (2, 1200007, 32, 100, True), (4, 50, 128, 100, False), (6, 9000, 128, 256, False),
                                              (5, 300001, 128, 100, True), (40, 270011, 256, 10, False)])   # streamed skinny-M path
def test_rank_all_selection_is_exact(B, N, d, k, use_bias):
    """Multi-chunk (N > 8192), multi-panel (N > 2**20), k not a multiple of 4, N < k, ties: selection, order and
    target rank must equal numpy's on the device's own scores."""
    from oracle import rank as orank
    from prodsearch_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device='cpu').manual_seed(N)
    q = torch.randn(B, d, generator=g).cuda()
    table = torch.randn(N, d, generator=g)
    table[N // 3] = table[N // 2]                                         # exact duplicates -> exact score ties
    table[5] = table[N // 2]
    table = table.cuda()
    bias = (torch.randn(N, generator=g) * 0.1).cuda() if use_bias else None
    if bias is not None:
        bias[N // 3] = bias[N // 2] = bias[5] = 0.25
    target = torch.tensor(([N // 2, 5, N - 1, 0, N // 3, 17, 3] * (B // 7 + 1))[:B], dtype=torch.int64).cuda()
    nbytes = lib.ps_rank_scratch_bytes(B, N, d, k)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    top_idx = torch.empty(B, k, dtype=torch.int64, device='cuda')
    top_score = torch.empty(B, k, device='cuda')
    rank = torch.empty(B, dtype=torch.int32, device='cuda')
    _lib.check(lib.ps_rank_all(q.data_ptr(), B, d, table.data_ptr(), N, _lib.ptr(bias), target.data_ptr(), k,
                               top_idx.data_ptr(), top_score.data_ptr(), rank.data_ptr(), scratch.data_ptr(), nbytes,
                               torch.cuda.current_stream().cuda_stream), 'ps_rank_all')
    S = _device_scores(q, table, bias).cpu().numpy()
    want_idx, want_score, want_rank = orank.rank_scores(S, target.cpu().numpy(), k)
    kk = min(k, N)
    assert np.array_equal(top_idx.cpu().numpy()[:, :kk], want_idx)
    assert np.array_equal(top_score.cpu().numpy()[:, :kk], want_score)
    assert (top_idx.cpu().numpy()[:, kk:] == -1).all()
    assert np.array_equal(rank.cpu().numpy(), want_rank)


@pytest.mark.parametrize('N,k', [(20000, 100), (300, 256), (70000, 7)])
def test_rank_all_exact_ties_go_by_lower_index(N, k):
    """Every product identical => every score equal: the ranklist must be ids 0..k-1 and the target's rank its id + 1
    (the cut through a run of equal scores is the one place where lanes cannot decide locally)."""
    from prodsearch_amd import _lib
    lib = _lib.load()
    B, d = 3, 32
    q = torch.randn(B, d, device='cuda')
    table = torch.randn(1, d, device='cuda').expand(N, d).contiguous()
    target = torch.tensor([0, N - 1, N // 2], dtype=torch.int64, device='cuda')
    nbytes = lib.ps_rank_scratch_bytes(B, N, d, k)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    ti = torch.empty(B, k, dtype=torch.int64, device='cuda')
    ts = torch.empty(B, k, device='cuda')
    rk = torch.empty(B, dtype=torch.int32, device='cuda')
    _lib.check(lib.ps_rank_all(q.data_ptr(), B, d, table.data_ptr(), N, None, target.data_ptr(), k, ti.data_ptr(),
                               ts.data_ptr(), rk.data_ptr(), scratch.data_ptr(), nbytes,
                               torch.cuda.current_stream().cuda_stream), 'ps_rank_all')
    kk = min(k, N)
    assert torch.equal(ti[:, :kk].cpu(), torch.arange(kk).expand(B, kk))
    assert rk.tolist() == [1, N, N // 2 + 1]
    assert float((ts[:, :kk] - ts[:, :1]).abs().max()) == 0.0


def test_rank_all_target_outside_catalogue_and_bad_topk():
    from prodsearch_amd import _lib
    lib = _lib.load()
    q = torch.randn(2, 32, device='cuda')
    table = torch.randn(100, 32, device='cuda')
    target = torch.tensor([100, -1], dtype=torch.int64, device='cuda')
    nbytes = lib.ps_rank_scratch_bytes(2, 100, 32, 10)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    ti = torch.empty(2, 10, dtype=torch.int64, device='cuda')
    ts = torch.empty(2, 10, device='cuda')
    rk = torch.empty(2, dtype=torch.int32, device='cuda')
    _lib.check(lib.ps_rank_all(q.data_ptr(), 2, 32, table.data_ptr(), 100, None, target.data_ptr(), 10, ti.data_ptr(),
                               ts.data_ptr(), rk.data_ptr(), scratch.data_ptr(), nbytes,
                               torch.cuda.current_stream().cuda_stream), 'ps_rank_all')
    assert rk.tolist() == [0, 0]
    assert lib.ps_rank_scratch_bytes(2, 100, 32, 257) == -1
    assert lib.ps_rank_all(q.data_ptr(), 2, 32, table.data_ptr(), 100, None, target.data_ptr(), 300, ti.data_ptr(),
                           ts.data_ptr(), rk.data_ptr(), scratch.data_ptr(), nbytes, None) != 0


def test_ranklist_lines_format():
    from prodsearch_amd import evaluate
    lines = list(evaluate.ranklist_lines(['A1', 'B2'], [3, 4], ['p%d' % i for i in range(5)],
                                         torch.tensor([[2, 0], [4, -1]]), torch.tensor([[1.5, 0.25], [2.0, float('-inf')]])))
    assert lines == ["A1_3 Q0 p2 1 1.500000 ReviewTransformer\n", "A1_3 Q0 p0 2 0.250000 ReviewTransformer\n",
                     "B2_4 Q0 p4 1 2.000000 ReviewTransformer\n"]
