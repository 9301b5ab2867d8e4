"""A few launches of one product in one form (for counter passes): gemm_x3_one.py M N K ta tb acc mode shape [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prodsearch_amd import _lib

M, N, K, ta, tb, acc, mode, shape = [int(x) for x in sys.argv[1:9]]
iters = int(sys.argv[9]) if len(sys.argv) > 9 else 10
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
lib.ps_gemm_x3_config(mode, shape)
A = torch.randn((K, M) if ta else (M, K), device='cuda')
Bm = torch.randn((K, N) if tb else (N, K), device='cuda') * 0.1
C = torch.zeros(M, N, device='cuda')
for _ in range(iters):
    if shape == 4 and not ta:      # the pre-split-weight form (gemm_x3w_kernel)
        _lib.check(lib.ps_gemm_f32_weight(A.data_ptr(), K, Bm.data_ptr(), tb, C.data_ptr(), N, M, N, K, None, 1.0, st), 'gemm_w')
        continue
    _lib.check(lib.ps_gemm_f32(A.data_ptr(), M if ta else K, ta, Bm.data_ptr(), N if tb else K, tb, C.data_ptr(), N, M, N, K,
                               None, 1.0, acc, st), 'gemm')
torch.cuda.synchronize()
