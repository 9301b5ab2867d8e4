"""Diagnostic: phase timeline of the fused per-replica forward (mlp_fwd_t_kernel), workgroup 0, all 8 waves (s_memtime).
    python tools/mlp_stamps.py        (on the GPU box)"""
import ctypes, os, sys
os.environ['PS_DIAG_LIB'] = '1'      # stamps exist in the diagnostic build only (python -m prodsearch_amd.build --diag)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prodsearch_amd import ItemTransformerRanker, readme_tem_args, synth, _lib
P_, V, B = 18357, 32387, 384
a = readme_tem_args(dropout=0.1)
wd = synth.make_word_dists(V)
m = ItemTransformerRanker(a, 'cuda', V, P_, None, word_dists=wd)
m.train()
b = synth.make_tem_batch(1, B, P_, V, word_dists=wd).to('cuda')
lib = ctypes.CDLL(_lib.lib_path())
for _ in range(5):
    m(b)
for _ in range(3):
    loss = m(b); m.zero_grad(); loss.backward()
buf = torch.zeros(256, dtype=torch.int64, device='cuda')
lib.ps_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
loss = m(b); m.zero_grad(); loss.backward()          # forward stamps in words 0-127, the fused backward's in 128-255
torch.cuda.synchronize()
lib.ps_debug_set_stamp_buffer(ctypes.c_void_p(0))
t = buf.cpu()[:128].view(8, 16)
tb = buf.cpu()[128:].view(8, 16)
names = ['start', 'P barrier', 'Wo done', 'A barrier', 'LN1 done', 'B barrier', 'blk0 W1', 'blk0 epilogue', '-', 'blk1 W1',
         'blk1 epilogue', '-', 'chain done', 'C barrier', 'LN2 done', 'ticket']
t0 = int(t[:, 0].min())
print('%-14s' % 'phase' + ''.join('  wave%d' % w for w in range(8)) + '   (cycles since the first wave started)')
for i, n in enumerate(names):
    if n == '-':
        continue
    print('%-14s' % n + ''.join('%7d' % (int(t[w, i]) - t0 if int(t[w, i]) else -1) for w in range(8)))

namesb = ['start', 'P barrier', 'LN2-bwd, barrier 1', 'park, barrier 2', 'blk0 W2^T', 'blk0 epilogue', 'blk1 W2^T', 'blk1 epilogue', 'chain done',
          'barrier 3', 'barrier 4', 'LN1-bwd, barrier 5', 'park+Wo^T, barrier 6', 'end']
if int(tb.max()):
    t0 = int(tb[:, 0].min())
    print('fused backward (mlp_bwd_t_kernel), workgroup 0')
    print('%-22s' % 'phase' + ''.join('  wave%d' % w for w in range(8)))
    for i, n in enumerate(namesb):
        print('%-22s' % n + ''.join('%7d' % (int(tb[w, i]) - t0 if int(tb[w, i]) else -1) for w in range(8)))
