"""Diagnostic: phase timeline of one workgroup of the fp32 GEMM kernel INSIDE the C2 training step (s_memtime per wave).
    python tools/gemm_f32_stamps.py 2      the backward's dX product over the valid-row list (dK.Wk + dV.Wv + fan-in)
    python tools/gemm_f32_stamps.py 3      the forward K/V projection over the valid-row list
    python tools/gemm_f32_stamps.py 4      the grouped FF / Wo weight gradients (bf16x3 kernel: use tools/gemm_stamps.py's phase names)   (on the GPU box)"""
import argparse, ctypes, os, sys
which = sys.argv[1] if len(sys.argv) > 1 else '2'
os.environ['PS_DIAG_LIB'] = '1'      # stamps exist in the diagnostic build only (python -m prodsearch_amd.build --diag)
os.environ['PS_GEMM_STAMP'] = which
if len(sys.argv) > 2:
    os.environ['PS_NO_SIDE'] = sys.argv[2]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from prodsearch_amd import _lib

a = argparse.Namespace(workload='c2', encoder='pvc', dropout=0.1, row_sparse=False, sharded=False)
wl = bench.TemWorkload(a, 'c2', 0, torch.device('cuda', 0))
wl.model.train()
raw = ctypes.CDLL(_lib.lib_path())


def step(i):
    loss = wl.forward(i)
    wl.model.zero_grad()
    loss.backward()
    wl.optim.step()


for i in range(8):
    step(i)
torch.cuda.synchronize()
buf = torch.zeros(4 * 32, dtype=torch.int64, device='cuda')
raw.ps_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
step(9)
torch.cuda.synchronize()
raw.ps_debug_set_stamp_buffer(ctypes.c_void_p(0))
t = buf.cpu().view(4, 32)
live = [w for w in range(4) if int(t[w, 0])]
t0 = min(int(t[w, 0]) for w in live) if live else 0
names = {0: 'start', 1: 'list length known', 2: 'row indices in LDS', 3: 'first slab in registers', 4: 'first slab in LDS',
         30: 'tile staged for the epilogue', 31: 'end'}
for s in range(16):
    names[5 + s] = 'slab %d done' % s
if which == '4':
    names = {0: 'start', 1: 'prologue loads issued', 30: 'main loop done', 31: 'end'}
    for sl in range(7):
        names.update({2 + 4 * sl: 'slab %d stored' % sl, 3 + 4 * sl: 'slab %d barrier' % sl, 4 + 4 * sl: 'slab %d products' % sl, 5 + 4 * sl: 'slab %d barrier 2' % sl})
print('%-30s' % 'phase' + ''.join('   wave%d' % w for w in range(4)) + '   (s_memtime ticks since the first wave started; 1 tick = 10 ns)')
for i in range(32):
    if i in names and any(int(t[w, i]) for w in range(4)):
        print('%-30s' % names[i] + ''.join('%8d' % (int(t[w, i]) - t0 if int(t[w, i]) else -1) for w in range(4)))
