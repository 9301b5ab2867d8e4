"""Diagnostic: the phases of every workgroup of ONE replica attention backward launch (attn_bwd_wf4_kernel, C2 shape) on the 100 MHz counter
all CUs share (s_memrealtime, 10 ns ticks; wave 0 of each workgroup).
    python tools/attn_bwd_wg_times.py        (GPU box; diagnostic library)"""
import ctypes, os, sys
os.environ['PS_DIAG_LIB'] = '1'
os.environ['PS_ABW_STAMP'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from prodsearch_amd import ItemTransformerRanker, readme_tem_args, synth, _lib
P_, V, B = 18357, 32387, 384
a = readme_tem_args(dropout=0.1)
wd = synth.make_word_dists(V)
m = ItemTransformerRanker(a, 'cuda', V, P_, None, word_dists=wd)
m.train()
b = synth.make_tem_batch(1, B, P_, V, word_dists=wd).to('cuda')
lib = ctypes.CDLL(_lib.lib_path())
for _ in range(4):
    loss = m(b); m.zero_grad(); loss.backward()
buf = torch.zeros(8 * 4096, dtype=torch.int64, device='cuda')
loss = m(b); m.zero_grad()
torch.cuda.synchronize()
lib.ps_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
loss.backward()
torch.cuda.synchronize()
lib.ps_debug_set_stamp_buffer(ctypes.c_void_p(0))
t = buf.cpu().numpy().reshape(4096, 8)[32:2 * B]          # (the fused MLP kernels' own stamps share the first 256 words of the buffer)
t = t[t[:, 0] != 0]
t0 = t[:, 0].min()
names = ['start', 'replica sums parked', 'barrier 1', 'wave 0: softmax bwd, dK | dV', 'barrier 2', 'd x product + stores', 'end']
print("%d workgroups stamped; 1 tick = 10 ns; span first start -> last end: %d ticks" % (len(t), t[:, 6].max() - t0))
print("%-30s %6s %6s %6s %6s %6s   (since the launch's first start)" % ('phase reached', 'min', 'p10', 'median', 'p90', 'max'))
for i, n in enumerate(names):
    c = t[:, i] - t0
    print("%-30s %6d %6d %6d %6d %6d" % (n, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
print("%-30s %6s %6s %6s %6s %6s   (per workgroup, since the previous phase)" % ('phase length', 'min', 'p10', 'median', 'p90', 'max'))
for i in range(1, 7):
    c = t[:, i] - t[:, i - 1]
    print("%-30s %6d %6d %6d %6d %6d" % (names[i], c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
