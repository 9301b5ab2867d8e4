"""Time ps_gemm_f32 on the hot path's shapes: python tools/gemm_sweep.py  (env PS_GEMM_DEEP_MAX to force the deep-K variant)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prodsearch_amd import _lib
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
shapes = [(8064, 512, 128, 0, 0), (8064, 128, 512, 0, 0), (8064, 128, 128, 0, 0), (8064, 512, 128, 0, 1), (8064, 128, 512, 0, 1),
          (8064, 128, 128, 0, 1), (384, 128, 128, 0, 0), (8064, 384, 128, 0, 0)]
for M, N, K, ta, tb in shapes:
    A = torch.randn(M, K, device='cuda')
    Bm = torch.randn(N, K, device='cuda') if tb == 0 else torch.randn(K, N, device='cuda')
    C = torch.zeros(M, N, device='cuda')
    ldb = K if tb == 0 else N
    for _ in range(20):
        rc = lib.ps_gemm_f32(A.data_ptr(), K, ta, Bm.data_ptr(), ldb, tb, C.data_ptr(), N, M, N, K, None, 1.0, 0, st)
        assert rc == 0, lib.ps_last_error()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(200):
        lib.ps_gemm_f32(A.data_ptr(), K, ta, Bm.data_ptr(), ldb, tb, C.data_ptr(), N, M, N, K, None, 1.0, 0, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    print("M=%d N=%d K=%d ta=%d tb=%d: %.1f us  %.1f TFLOP/s" % (M, N, K, ta, tb, us, 2.0 * M * N * K / us / 1e6))
