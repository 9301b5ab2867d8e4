"""Adversarial operands for the bf16x3 product form (csrc/gemm.hip, gemm_x3_kernel; the fused per-replica kernels use the
same split): every fp32 operand is split exactly into three bf16 values and a product step is six bf16 MFMAs with fp32
accumulation.  The claim "the fp32 MFMA's accuracy" (include/prodsearch_hip.h, ps_gemm_x3_config) is checked where it is
hardest: a 21,504-deep reduction whose terms span 2^+-40 and cancel; non-finite operands (they must poison exactly the
outputs the fp32 kernel poisons — the CLASS may differ: Inf * b becomes Inf * b_hi + Inf * b_mid + ..., whose terms have
opposite signs, so the form yields NaN where the fp32 MFMA yields +-Inf; no split of b can avoid that); and operands so
small that the low plane of the split is a bf16 denormal.  Reference: models/transformer.py:47-57, neural.py:30-33 run these
products in fp32 on ATen."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def x3_restore():
    from prodsearch_amd import _lib
    lib = _lib.load()
    yield lib
    lib.ps_gemm_x3_config(1, -1)


def _run(lib, A, Bm, ta, tb, mode, shape=-1, acc=0):
    """C = A @ Bm through ps_gemm_f32 in the requested storage layout; mode 1 = bf16x3 form, 0 = fp32 MFMA."""
    from prodsearch_amd import _lib
    M, K = A.shape
    N = Bm.shape[1]
    Ad = (A.t().contiguous() if ta else A.contiguous()).cuda()
    Bd = (Bm.contiguous() if tb else Bm.t().contiguous()).cuda()
    lib.ps_gemm_x3_config(mode, shape if mode else -1)
    C = torch.zeros(M, N, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.ps_gemm_f32(Ad.data_ptr(), M if ta else K, ta, Bd.data_ptr(), N if tb else K, tb, C.data_ptr(), N,
                               M, N, K, None, 1.0, acc, st), 'ps_gemm_f32')
    torch.cuda.synchronize()
    return C.cpu()


@pytest.mark.parametrize('layout', [(0, 0, 1), (0, 1, 1), (1, 1, 0), (0, 0, 3), (0, 1, 3), (1, 1, 3)])      # forward, dX, weight-gradient storage; forced tile (3: direct-to-LDS form)
def test_deep_cancelling_reduction_with_mixed_magnitudes(x3_restore, layout):
    """K = 21,504 terms per output whose magnitudes span 2^-40 .. 2^+40 and whose big terms cancel in pairs: the error is
    bounded by a few fp32 ulps of sum |a b| for BOTH forms (a product form that dropped the low plane would be off by
    2^-16 of the largest term)."""
    lib = x3_restore
    ta, tb, shape = layout
    gen = torch.Generator().manual_seed(11)
    M, N, K = 256, 256, 21504
    ea = torch.randint(-40, 41, (M, K), generator=gen).float()
    A = torch.randn(M, K, generator=gen) * torch.exp2(ea)
    Bm = torch.randn(K, N, generator=gen) * torch.exp2(-ea[0:1].t().expand(K, N) * 0.5)
    # cancellation: the second half of the reduction repeats the first with the opposite sign, plus a small signal
    h = K // 2
    A[:, h:] = -A[:, :h]
    Bm[h:] = Bm[:h] * (1.0 + 2.0 ** -12 * torch.randn(h, N, generator=gen))
    ref = A.double() @ Bm.double()
    mag = A.double().abs() @ Bm.double().abs()
    e = {}
    for mode in (1, 0):
        C = _run(lib, A, Bm, ta, tb, mode, shape, acc=2 if ta else 0)
        assert torch.isfinite(C).all()
        e[mode] = float(((C.double() - ref).abs() / mag).max())
    # fp32 accumulation over 21,504 terms: ~1e-6 of sum |a b| for EITHER form (measured 8.7e-7 bf16x3, 9.5e-7 fp32 MFMA); a form
    # that dropped a plane would sit at 2^-16 = 1.5e-5
    assert e[1] < 2e-6 and e[0] < 2e-6 and e[1] < 1.5 * e[0] + 1e-7, e
    assert float((ref.abs() / mag).median()) < 1e-2          # the case really cancels


@pytest.mark.parametrize('shape', [1, 3])
def test_non_finite_operands_poison_exactly_the_outputs_the_fp32_kernel_poisons(x3_restore, shape):
    """An Inf / NaN operand makes exactly the outputs non-finite that the fp32 MFMA makes non-finite (never a silently finite
    value, never a poisoned neighbour); NaN operands give NaN in both forms; an Inf operand gives +-Inf in the fp32 form and
    may give NaN in the bf16x3 form (see the module docstring)."""
    lib = x3_restore
    gen = torch.Generator().manual_seed(12)
    M, N, K = 21504, 256, 256
    A = torch.randn(M, K, generator=gen)
    Bm = torch.randn(K, N, generator=gen) * 0.1
    A[3, 7] = float('inf')            # row 3: +-Inf by the sign of B[7, :]
    A[64, 0] = float('-inf')
    A[64, 1] = float('inf')           # row 64: Inf - Inf -> NaN wherever both hit non-zero weights
    A[100, 5] = float('nan')          # row 100: NaN
    Bm[9, :] = 0.0
    A[200, 9] = float('inf')          # row 200: Inf * 0 -> NaN
    Bm[11, 17] = float('inf')         # column 17: an Inf weight
    outs = {mode: _run(lib, A, Bm, 0, 0, mode, shape) for mode in (1, 0)}
    c3, c1 = outs[1], outs[0]
    assert torch.equal(torch.isfinite(c3), torch.isfinite(c1))
    assert bool((torch.isnan(c3) | ~torch.isnan(c1)).all())                      # NaN in the fp32 form => NaN in the bf16x3 form
    fin = torch.isfinite(c1)
    assert int((~fin).sum()) >= 4 * N and float((c3[fin] - c1[fin]).abs().max()) < 1e-4
    assert torch.isinf(c1[3]).all() and torch.isnan(c1[64]).any() and torch.isnan(c1[100]).all() and torch.isnan(c1[200]).all()


@pytest.mark.parametrize('shape', [1, 3])
def test_operands_whose_low_plane_is_a_bf16_denormal(x3_restore, shape):
    """|a| ~ 2^-112: hi / mid are normal bf16 values, the low plane falls below 2^-126.  Whatever the matrix core does with
    a denormal bf16 input, the result must stay within the product form's bound against fp64 — if denormal inputs were
    flushed the low plane would be lost and the error 2^-17 of a term (1e-5 relative), 20x the bound."""
    lib = x3_restore
    gen = torch.Generator().manual_seed(13)
    M, N, K = 21504, 256, 256
    A = torch.randn(M, K, generator=gen) * 2.0 ** -112
    Bm = torch.randn(K, N, generator=gen) * 2.0 ** 90
    ref = A.double() @ Bm.double()
    mag = A.double().abs() @ Bm.double().abs()
    e = {mode: float(((_run(lib, A, Bm, 0, 0, mode, shape).double() - ref).abs() / mag).max()) for mode in (1, 0)}
    print("denormal low plane: error / sum|ab|  bf16x3 %.3g  fp32 MFMA %.3g" % (e[1], e[0]))
    assert e[0] < 6e-7
    assert e[1] < 6e-7, e


@pytest.mark.parametrize('M,N,K,tb', [(21504, 256, 256, 0), (21504, 1024, 256, 0), (21504, 256, 1024, 0), (21504, 256, 1024, 1),
                                      (21504, 1024, 256, 1), (8064, 128, 512, 1), (4100, 160, 96, 0), (5001, 96, 224, 1)])
def test_products_against_pre_split_weights_match_fp64(x3_restore, M, N, K, tb):
    """gemm_x3w_kernel (csrc/gemm.hip): the weight operand split ONCE into bf16 planes (both orientations), the activation split
    on its fragments in registers — the forward / input-gradient products of the d = 256 step (models/neural.py:30-33, 86-96).
    Same bound against fp64 as the fp32 MFMA, ragged row and column tails included."""
    from prodsearch_amd import _lib
    lib = x3_restore
    lib.ps_gemm_x3_config(1, 4)
    gen = torch.Generator().manual_seed(M + 3 * N + 7 * K + tb)
    A = torch.randn(M, K, generator=gen)
    W = torch.randn(K, N, generator=gen) * 0.1           # logical [K][N]
    bias = torch.randn(N, generator=gen)
    Wd = (W.contiguous() if tb else W.t().contiguous()).cuda()
    Ad, bd = A.cuda(), bias.cuda()
    C = torch.zeros(M, N, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.ps_gemm_f32_weight(Ad.data_ptr(), K, Wd.data_ptr(), tb, C.data_ptr(), N, M, N, K, bd.data_ptr(), 0.5, st),
               'ps_gemm_f32_weight')
    lib.ps_gemm_x3_config(1, -1)
    C1 = torch.zeros(M, N, device='cuda')
    _lib.check(lib.ps_gemm_f32(Ad.data_ptr(), K, 0, Wd.data_ptr(), N if tb else K, tb, C1.data_ptr(), N, M, N, K,
                               bd.data_ptr(), 0.5, 0, st), 'ps_gemm_f32')
    torch.cuda.synchronize()
    rows = slice(0, M, 7)
    ref = (A[rows].double() @ W.double() + bias.double()) * 0.5
    mag = A[rows].double().abs() @ W.double().abs() + bias.double().abs()
    e = float(((C.cpu()[rows].double() - ref).abs() / mag).max())
    e1 = float(((C1.cpu()[rows].double() - ref).abs() / mag).max())
    assert e < 6e-7 and e1 < 6e-7, (e, e1)
    # (the two kernels issue the same six MFMAs per step in the same order over the same k sequence: their results may be — and at
    # these shapes are — bitwise equal; what differs is who splits the operands, and when)
    assert float((C - C1).abs().max()) < 1e-4 * float(C1.abs().max())
