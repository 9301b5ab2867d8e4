"""Does kernel time depend on the instruction cache being warm?  Same tiny GEMM, (a) repeated,
(b) interleaved with other GEMM code variants (different ta/tb/epilogue = different code)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prodsearch_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
M, N, K = 384, 128, 128
A = torch.randn(M, K, device='cuda'); Bm = torch.randn(N, K, device='cuda'); C = torch.zeros(M, N, device='cuda')
bias = torch.randn(N, device='cuda')
def call(ta, tb, acc=0):
    lib.ps_gemm_f32(A.data_ptr(), K, ta, Bm.data_ptr(), K, tb, C.data_ptr(), N, M if not ta else 128, N, K if not ta else 384, None, 1.0, acc, st)
def timeit(seq, n=300):
    for f in seq: f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        for f in seq: f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n / len(seq)
same = [lambda: call(0, 0)]
mix = [lambda: call(0, 0), lambda: call(0, 1), lambda: call(1, 1), lambda: call(1, 0), lambda: call(0, 0, 2), lambda: call(0, 1, 2)]
print("same kernel repeated : %.2f us/launch" % timeit(same))
print("6 code variants mixed: %.2f us/launch" % timeit(mix))
x = torch.randn(1 << 22, device='cuda')
mix2 = [lambda: call(0, 0), lambda: x.mul_(1.0001), lambda: torch.tanh(x[:1024]), lambda: x[:4096].sum()]
print("gemm + 3 torch kernels: %.2f us/launch" % timeit(mix2))
