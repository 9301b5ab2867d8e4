"""Time one ps_gemm_f32 shape: python tools/gemm_one.py M N K"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prodsearch_amd import _lib
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
M, N, K = [int(x) for x in sys.argv[1:4]]
A = torch.randn(M, K, device='cuda'); Bm = torch.randn(N, K, device='cuda'); C = torch.zeros(M, N, device='cuda')
for _ in range(30):
    lib.ps_gemm_f32(A.data_ptr(), K, 0, Bm.data_ptr(), K, 0, C.data_ptr(), N, M, N, K, None, 1.0, 0, st)
torch.cuda.synchronize()
