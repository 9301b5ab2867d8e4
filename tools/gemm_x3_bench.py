"""bf16x3 GEMM form against the fp32 MFMA form: accuracy (vs fp64) and time per launch on the step's large shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prodsearch_amd import _lib

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream


def one(M, N, K, ta, tb, acc, mode, shape, iters=50, check=True):
    lib.ps_gemm_x3_config(mode, shape)
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    A = torch.randn((K, M) if ta else (M, K), device='cuda', generator=g)
    Bm = torch.randn((K, N) if tb else (N, K), device='cuda', generator=g) * 0.1
    C = torch.zeros(M, N, device='cuda')
    args = (A.data_ptr(), M if ta else K, ta, Bm.data_ptr(), N if tb else K, tb, C.data_ptr(), N, M, N, K, None, 1.0, acc, st)
    _lib.check(lib.ps_gemm_f32(*args), 'gemm')
    err = -1.0
    if check:
        rows = slice(0, min(M, 2048))
        Ad = (A.t() if ta else A)[rows].double()
        ref = Ad @ (Bm if tb else Bm.t()).double()
        mag = Ad.abs() @ (Bm if tb else Bm.t()).double().abs()
        err = float(((C[rows].double() - ref).abs() / mag).max())
    for _ in range(5):
        lib.ps_gemm_f32(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        lib.ps_gemm_f32(*args)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters, err


if __name__ == '__main__':
    shapes = [
        (21504, 256, 256, 0, 0, 0), (21504, 1024, 256, 0, 0, 0), (21504, 256, 1024, 0, 0, 0),      # C5 forward linears
        (21504, 256, 1024, 0, 1, 0), (21504, 1024, 256, 0, 1, 0), (21504, 256, 256, 0, 1, 0),      # their dX products
        (1024, 256, 21504, 1, 1, 2), (256, 1024, 21504, 1, 1, 2), (256, 256, 21504, 1, 1, 2),      # weight gradients
        (78336, 384, 128, 0, 0, 0), (78336, 128, 128, 0, 0, 0), (78336, 128, 384, 0, 1, 0),        # review transformer
        (128, 384, 78336, 1, 1, 2),
        (8064, 512, 128, 0, 0, 0), (8064, 128, 512, 0, 0, 0), (4096, 4096, 4096, 0, 0, 0),
    ]
    if len(sys.argv) > 1 and sys.argv[1] == 'wgrad':
        shapes = [x for x in shapes if x[3] == 1]
        print('ksplit', os.environ.get('PS_GEMM_KSPLIT', '4'))
    for (M, N, K, ta, tb, acc) in shapes:
        t0, e0 = one(M, N, K, ta, tb, acc, 0, -1)
        line = "M=%6d N=%5d K=%6d ta=%d tb=%d acc=%d | fp32 %7.1f us %6.1f TF err %.1e |" % (
            M, N, K, ta, tb, acc, t0, 2.0 * M * N * K / t0 / 1e6, e0)
        for shp in (0, 1, 2, 3):
            t, e = one(M, N, K, ta, tb, acc, 1, shp)
            line += " x3[%d] %7.1f us %6.1f TF err %.1e |" % (shp, t, 2.0 * M * N * K / t / 1e6, e)
        if not ta and N % 32 == 0 and K % 32 == 0:      # the same product against pre-split weight planes (gemm_x3w_kernel)
            lib.ps_gemm_x3_config(1, 4)
            A = torch.randn(M, K, device='cuda')
            W = torch.randn((K, N) if tb else (N, K), device='cuda') * 0.1
            C = torch.zeros(M, N, device='cuda')
            wargs = (A.data_ptr(), K, W.data_ptr(), tb, C.data_ptr(), N, M, N, K, None, 1.0, st)
            for _ in range(5):
                _lib.check(lib.ps_gemm_f32_weight(*wargs), 'gemm_w')
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(50):
                lib.ps_gemm_f32_weight(*wargs)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e3 / 50
            line += " x3w %7.1f us %6.1f TF (incl. the plane split launch) |" % (t, 2.0 * M * N * K / t / 1e6)
        print(line, flush=True)
