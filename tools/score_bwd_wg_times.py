"""Diagnostic: every workgroup of ONE score-backward launch (score_bwd_kernel, C2 shape, side stream of the step) on the 100 MHz counter all
CUs share (s_memrealtime, 10 ns ticks).    python tools/score_bwd_wg_times.py        (GPU box; diagnostic library)"""
import ctypes, os, sys
os.environ['PS_DIAG_LIB'] = '1'
os.environ['PS_SBW_STAMP'] = '1'
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
n = 2 * B + (B * 21 + 15) // 16
t = buf.cpu().numpy().reshape(8192, 4)[64:n]          # (the fused MLP kernels' own stamps share the first 256 words of the buffer)
t = t[t[:, 0] != 0]
t0 = t[:, 0].min()
rows, items = t[: 2 * B - 64], t[2 * B - 64:]
print("%d workgroups stamped (%d batch-row, %d item); 1 tick = 10 ns; span first start -> last end: %d ticks" % (len(t), len(rows), len(items), t[:, 3].max() - t0))
for name, x in (('batch-row workgroups (word tasks)', rows), ('item workgroups', items)):
    print(name)
    print("  %-26s %6s %6s %6s %6s %6s" % ('reached (since first start)', 'min', 'p10', 'median', 'p90', 'max'))
    for i, nme in enumerate(['start', 'item tasks issued', 'word tasks issued', 'end']):
        c = x[:, i][x[:, i] != 0] - t0
        if len(c): print("  %-26s %6d %6d %6d %6d %6d" % (nme, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
    life = x[:, 3] - x[:, 0]
    print("  %-26s %6d %6d %6d %6d %6d" % ('life', life.min(), np.percentile(life, 10), np.median(life), np.percentile(life, 90), life.max()))
