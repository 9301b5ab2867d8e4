"""Micro-benchmark of the fp32 MFMA GEMM entry (ps_gemm_f32): time per launch and TFLOP/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prodsearch_amd import _lib

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream


def run(M, N, K, ta=0, tb=0, acc=0, iters=200):
    A = torch.randn(K, M, device='cuda') if ta else torch.randn(M, K, device='cuda')
    Bm = torch.randn(K, N, device='cuda') if tb else torch.randn(N, K, device='cuda')
    C = torch.zeros(M, N, device='cuda')
    args = (A.data_ptr(), M if ta else K, ta, Bm.data_ptr(), N if tb else K, tb, C.data_ptr(), N, M, N, K, None, 1.0, acc, st)
    lib.ps_gemm_f32(*args)
    ref = (A.t() if ta else A) @ (Bm if tb else Bm.t())
    err = float((C - ref).abs().max() / ref.abs().max())
    for _ in range(10):
        lib.ps_gemm_f32(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        lib.ps_gemm_f32(*args)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / iters
    print("M=%6d N=%4d K=%5d ta=%d tb=%d acc=%d : %8.2f us  %7.2f TFLOP/s  err %.1e" % (M, N, K, ta, tb, acc, t * 1e6, 2.0 * M * N * K / t / 1e12, err))


if __name__ == '__main__':
    x = torch.zeros(1, device='cuda')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(1000):
        lib.ps_zero_floats(x.data_ptr(), 1, st)
    e1.record(); torch.cuda.synchronize()
    print("memset launch: %.2f us" % (e0.elapsed_time(e1)))
    for shp in [(384, 128, 128), (8064, 128, 128), (8064, 512, 128), (8064, 128, 512), (4096, 4096, 4096)]:
        run(*shp)
    run(8064, 512, 128, 0, 1)
    run(512, 128, 8064, 1, 1, 2)
    run(128, 128, 8064, 1, 1, 2)
    # review-transformer shapes (78k sequence positions)
    run(78336, 256, 128)
    run(78336, 128, 128)
    run(128, 128, 78336, 1, 1, 2)
    run(128, 256, 78336, 1, 1, 2)
