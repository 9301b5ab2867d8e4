"""Is the step host-bound?  Host time to ENQUEUE vs time to drain, for single launches and a whole step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prodsearch_amd import _lib, readme_tem_args, synth, ItemTransformerRanker, build_optim

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
A = torch.randn(384, 128, device='cuda'); Bm = torch.randn(128, 128, device='cuda'); C = torch.zeros(384, 128, device='cuda')
args = (A.data_ptr(), 128, 0, Bm.data_ptr(), 128, 0, C.data_ptr(), 128, 384, 128, 128, None, 1.0, 0, st)
for n in (1, 1000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        lib.ps_gemm_f32(*args)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("gemm tiny x%d: enqueue %.2f us/call, drain %.2f us" % (n, (t1 - t0) / n * 1e6, (t2 - t1) * 1e6))

ns = readme_tem_args(dropout=float(os.environ.get('DROPOUT', '0.1')))
P_, V_ = 18357, 32387
wd = synth.make_word_dists(V_)
m = ItemTransformerRanker(ns, 'cuda', V_, P_, None, word_dists=wd); opt = build_optim(ns, m, None); m.train()
b = synth.make_tem_batch(1, 384, P_, V_, Q=8, L=20, W=1, word_dists=wd).to('cuda')
def step():
    loss = m(b); m.zero_grad(); loss.backward(); opt.step(); return loss
for _ in range(20): step()
torch.cuda.synchronize()
for phase in ('fwd', 'bwd', 'opt', 'all'):
    N = 100
    torch.cuda.synchronize(); tf = tb = to = 0.0; t00 = time.perf_counter()
    for _ in range(N):
        t0 = time.perf_counter(); loss = m(b); t1 = time.perf_counter(); m.zero_grad(); loss.backward(); t2 = time.perf_counter(); opt.step(); t3 = time.perf_counter()
        tf += t1 - t0; tb += t2 - t1; to += t3 - t2
        if phase != 'all': torch.cuda.synchronize()
    t1_ = time.perf_counter(); torch.cuda.synchronize(); t2_ = time.perf_counter()
    print("%s: host enqueue fwd %.1f bwd %.1f opt %.1f us/step; loop %.1f us/step; final drain %.1f us" % (phase, tf/N*1e6, tb/N*1e6, to/N*1e6, (t1_-t00)/N*1e6, (t2_-t1_)*1e6))
    break
# GPU-only time of each phase: enqueue then sync, minus enqueue
for name, fn in (('fwd', lambda: m(b)),):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); print(name, "with sync at end: %.1f us/iter" % ((time.perf_counter() - t0) / 50 * 1e6))
