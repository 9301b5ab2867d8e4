"""Diagnostic: per-slab timeline of the wave-specialised fused MLP forward (workgroup 0: matrix wave 0, helper wave 4)."""
import ctypes, os, sys
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
buf = torch.zeros(256, dtype=torch.int64, device='cuda')
lib.ps_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
with torch.no_grad():
    m(b)
torch.cuda.synchronize()
lib.ps_debug_set_stamp_buffer(ctypes.c_void_p(0))
t = buf.cpu().tolist()
M, H = t[:128], t[128:]
t0 = min(M[0], H[0])
NS = 18
print("slab  M:start  M:work  M:wait | H:start  H:work  H:wait   (cycles; work = start->barrier arrival, wait = arrival->next start)")
for s in range(NS):
    ms, me, mn = M[2 * s], M[2 * s + 1], M[2 * s + 2]
    hs, he, hn = H[2 * s], H[2 * s + 1], H[2 * s + 2]
    print("%3d  %8d %7d %7d | %8d %7d %7d" % (s, ms - t0, me - ms, mn - me, hs - t0, he - hs, hn - he))
print("total M %d  H %d cycles" % (M[2 * NS] - t0, H[2 * NS] - t0))

