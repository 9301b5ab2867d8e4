"""Diagnostic: the life of every workgroup of ONE gather+score launch at the C5 shape (8 M-row table, an index set the caches have
not seen): start / indices known / rows arrived / end on the 100 MHz counter all CUs share (s_memrealtime, 10 ns ticks).
    PS_SCORE_CH=4|8 python tools/gather_wg_times.py [--batch 1024]        (GPU box; diagnostic library)"""
import argparse, ctypes, os, sys
os.environ['PS_DIAG_LIB'] = '1'
os.environ['PS_SCORE_STAMP'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from prodsearch_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument('--rows', type=int, default=8_000_000)
ap.add_argument('--batch', type=int, default=1024)
a = ap.parse_args()
lib = _lib.load()
raw = ctypes.CDLL(_lib.lib_path())
d, B, K, W, P, V = 256, a.batch, 20, 1, a.rows, 2_000_000
dev = 'cuda'
gen = torch.Generator(device=dev).manual_seed(1)
table = torch.empty(P + 1, d, device=dev)
for i in range(0, P + 1, 1 << 22):
    table[i:i + (1 << 22)].normal_(generator=gen)
words = torch.randn(V, d, device=dev, generator=gen)
wbias = torch.zeros(V, device=dev)
desc = _lib.PsTemDesc()
desc.B, desc.K, desc.L, desc.Q, desc.W, desc.C = B, K, 20, 8, W, 0
desc.d, desc.H, desc.F, desc.n_layers = d, 8, 1024, 1
desc.product_size, desc.vocab_size = P, V
desc.use_pos_emb, desc.training, desc.dropout = 1, 1, 0.1
lay = _lib.PsTemWsLayout()
_lib.check(lib.ps_tem_workspace_layout(desc, lay), 'layout')
ws = torch.randn(lay.total_floats, device=dev)
params = _lib.PsTemTensors()
params.product_emb, params.word_emb, params.word_bias = table.data_ptr(), words.data_ptr(), wbias.data_ptr()
mk = lambda hi, *shape: torch.randint(0, hi, shape, device=dev, dtype=torch.int64, generator=gen)
sets = []
for _ in range(9):
    idx = (mk(P, B), mk(P, B, K), mk(V - 1, B, W), mk(V - 1, B, W * K))
    bt = _lib.PsTemBatch()
    bt.target_prod_idxs, bt.neg_item_idxs = idx[0].data_ptr(), idx[1].data_ptr()
    bt.pos_iword_idxs, bt.neg_word_idxs = idx[2].data_ptr(), idx[3].data_ptr()
    sets.append((idx, bt))
st = torch.cuda.current_stream()
for i in range(8):
    _lib.check(lib.ps_gather_score(desc, params, sets[i][1], ws.data_ptr(), st.cuda_stream), 'gs')
torch.cuda.synchronize()
NWG = 65536
buf = torch.zeros(4 * NWG, dtype=torch.int64, device=dev)
raw.ps_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
_lib.check(lib.ps_gather_score(desc, params, sets[8][1], ws.data_ptr(), st.cuda_stream), 'gs')      # a set no launch has touched
torch.cuda.synchronize()
raw.ps_debug_set_stamp_buffer(ctypes.c_void_p(0))
t = buf.cpu().numpy().reshape(NWG, 4)
live = t[:, 0] != 0
t = t[live]
t0 = t[:, 0].min()
print("PS_SCORE_CH=%s PS_SCORE_SIDX=%s B=%d: %d workgroups stamped; 1 tick = 10 ns" % (os.environ.get("PS_SCORE_CH", "default"), os.environ.get("PS_SCORE_SIDX", "1"), B, live.sum()))
print("span first start -> last end: %d ticks" % (t[:, 3].max() - t0))
for name, col in (("start", t[:, 0] - t0), ("idx known", t[:, 1] - t[:, 0]), ("rows arrived", t[:, 2] - t[:, 1]), ("tail", t[:, 3] - t[:, 2]),
                  ("life", t[:, 3] - t[:, 0])):
    print("%-13s min %5d  p10 %5d  median %5d  p90 %5d  max %5d" % (name, col.min(), np.percentile(col, 10), np.median(col), np.percentile(col, 90), col.max()))
T = t[:, 3].max() - t0
pts = [int(T * k / 16) for k in range(17)]
print("started by k/16 of the span: " + " ".join("%5d" % int((t[:, 0] - t0 <= p).sum()) for p in pts))
print("ended   by k/16 of the span: " + " ".join("%5d" % int((t[:, 3] - t0 <= p).sum()) for p in pts))
