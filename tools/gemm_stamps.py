"""Diagnostic: phase timeline of one mid-grid workgroup of the bf16x3 GEMM kernel (s_memtime per wave; PS_GEMM_STAMP=1).
    PS_GEMM_STAMP=1 python tools/gemm_stamps.py M N K ta tb shape        (on the GPU box)"""
import ctypes, os, sys
os.environ['PS_DIAG_LIB'] = '1'      # stamps exist in the diagnostic build only (python -m prodsearch_amd.build --diag)
os.environ['PS_GEMM_STAMP'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prodsearch_amd import _lib

M, N, K, ta, tb, shape = [int(x) for x in sys.argv[1:7]] if len(sys.argv) > 6 else (21504, 1024, 256, 0, 0, 1)
lib = _lib.load()
raw = ctypes.CDLL(_lib.lib_path())
st = torch.cuda.current_stream().cuda_stream
lib.ps_gemm_x3_config(1, shape)
A = torch.randn((K, M) if ta else (M, K), device='cuda')
Bm = torch.randn((K, N) if tb else (N, K), device='cuda') * 0.1
C = torch.zeros(M, N, device='cuda')


def run():
    _lib.check(lib.ps_gemm_f32(A.data_ptr(), M if ta else K, ta, Bm.data_ptr(), N if tb else K, tb, C.data_ptr(), N, M, N, K,
                               None, 1.0, 0, st), 'gemm')


for _ in range(5):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
print("M=%d N=%d K=%d ta=%d tb=%d shape=%d: %.1f us per launch" % (M, N, K, ta, tb, shape, 1e3 * e0.elapsed_time(e1) / 20))
buf = torch.zeros(4 * 32, dtype=torch.int64, device='cuda')
raw.ps_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
run()
torch.cuda.synchronize()
raw.ps_debug_set_stamp_buffer(ctypes.c_void_p(0))
t = buf.cpu().view(4, 32)
t0 = int(t[:, 0].min())
names = {0: 'start', 1: 'prologue loads issued', 30: 'main loop done', 31: 'end'}
for s in range(7):
    names[2 + 4 * s] = 'slab %d stored' % s
    names[3 + 4 * s] = 'slab %d barrier' % s
    names[4 + 4 * s] = 'slab %d products' % s
    names[5 + 4 * s] = 'slab %d barrier 2' % s
print('%-24s' % 'phase' + ''.join('   wave%d' % w for w in range(4)) + '   (s_memtime ticks since the first wave started; 100 MHz: 1 tick = 10 ns)')
for i in range(32):
    if i in names and any(int(t[w, i]) for w in range(4)):
        print('%-24s' % names[i] + ''.join('%8d' % (int(t[w, i]) - t0 if int(t[w, i]) else -1) for w in range(4)))
