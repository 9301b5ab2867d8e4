"""Timing-only variants of the direct-to-LDS bf16x3 GEMM (gemm_x3d_kernel) — diagnostic library only:
  PS_DIAG_LIB=1 [PS_X3D_DIAG=1|2|3] [PS_X3D_LDSPAD=bytes] python tools/x3d_diag.py
one process per variant (the knobs are read once); prints the time per launch of a few shapes in form 3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prodsearch_amd import _lib

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
tag = "diag=%s pad=%s" % (os.environ.get('PS_X3D_DIAG', '0'), os.environ.get('PS_X3D_LDSPAD', '0'))
shape = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for (M, N, K) in [(21504, 1024, 256), (21504, 256, 1024), (21504, 256, 256), (4096, 4096, 4096)]:
    lib.ps_gemm_x3_config(1, shape)
    A = torch.randn(M, K, device='cuda')
    Bm = torch.randn(N, K, device='cuda') * 0.1
    C = torch.zeros(M, N, device='cuda')
    args = (A.data_ptr(), K, 0, Bm.data_ptr(), K, 0, C.data_ptr(), N, M, N, K, None, 1.0, 0, st)
    for _ in range(5):
        _lib.check(lib.ps_gemm_f32(*args), 'gemm')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(30):
        lib.ps_gemm_f32(*args)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e3 / 30
    print("%-16s shape %d M=%6d N=%5d K=%5d  %8.1f us  %6.1f TF" % (tag, shape, M, N, K, t, 2.0 * M * N * K / t / 1e6), flush=True)
