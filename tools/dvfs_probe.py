"""Does the clock drop under the step's load?  (disproved as the cause of slow tiny kernels: round-1 investigation)"""
import sys, os, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prodsearch_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
def mk(M, N, K):
    A = torch.randn(M, K, device='cuda'); Bm = torch.randn(N, K, device='cuda'); C = torch.zeros(M, N, device='cuda')
    return (A.data_ptr(), K, 0, Bm.data_ptr(), K, 0, C.data_ptr(), N, M, N, K, None, 1.0, 0, st), (A, Bm, C)
tiny, k1 = mk(384, 128, 128)
big, k2 = mk(4096, 4096, 4096)
def timeit(args, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): lib.ps_gemm_f32(*args)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
print("cold tiny: %.2f us" % timeit(tiny, 100))
for r in range(5):
    print("tiny x2000 round %d: %.2f us" % (r, timeit(tiny, 2000)))
print(subprocess.run("rocm-smi --showclocks 2>/dev/null | grep -E 'sclk|mclk|fclk' | head -4", shell=True, capture_output=True, text=True).stdout)
print("big x50: %.2f us" % timeit(big, 50))
print("tiny right after big: %.2f us" % timeit(tiny, 200))
# interleave: one big then 20 tiny, time only the tiny ones
e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
tot = 0
for r in range(10):
    lib.ps_gemm_f32(*big)
    e[0].record()
    for _ in range(20): lib.ps_gemm_f32(*tiny)
    e[1].record(); torch.cuda.synchronize(); tot += e[0].elapsed_time(e[1])
print("tiny interleaved after big: %.2f us" % (tot * 1e3 / 200))
x = torch.zeros(1 << 20, device='cuda')
e[0].record()
for _ in range(1000): lib.ps_zero_floats(x.data_ptr(), 4, st)
e[1].record(); torch.cuda.synchronize(); print("memset(16B) back-to-back: %.2f us" % (e[0].elapsed_time(e[1])))
