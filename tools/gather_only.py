"""Run only the embedding-gather+score launch (ps_gather_score) in a loop: target for rocprofv3 --pmc passes.
    python tools/gather_only.py [c2|c5]   (c5: d=256, an 8 M-row table = 8.2 GB, B=1024: the HBM-bound shape)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prodsearch_amd import ItemTransformerRanker, _lib, readme_tem_args, synth
shape = sys.argv[1] if len(sys.argv) > 1 else 'c2'
B, P_, V, D, FF = (384, 18357, 32387, 128, 512) if shape == 'c2' else (1024, 8_000_000, 32387, 256, 1024)
ns = readme_tem_args(dropout=0.1, embedding_size=D, ff_size=FF)
wd = synth.make_word_dists(V)
model = ItemTransformerRanker(ns, 'cuda', V, P_, None, word_dists=wd)
model.train()
b = synth.make_tem_batch(1000, B, P_, V, Q=8, L=20, W=1, word_dists=wd).to('cuda')
with torch.no_grad():
    model(b)
plan = next(iter(model._plans.values()))
lib = _lib.load()
ps, _ = model._structs()
st = torch.cuda.current_stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(20):
    _lib.check(lib.ps_gather_score(plan.desc, ps, plan.batch, plan.ws.data_ptr(), st.cuda_stream), 'gather_score')
torch.cuda.synchronize()
e0.record(st)
for _ in range(200):
    lib.ps_gather_score(plan.desc, ps, plan.batch, plan.ws.data_ptr(), st.cuda_stream)
e1.record(st)
torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e-3 / 200
R, K, W = plan.layout.R, 20, 1
rows = B * (1 + K) * (1 + W)
nbytes = rows * (4 * D + 8) + (B * R + B) * 4 * D + rows * 4
print("gather+score %s: %.2f us/launch back to back, %d B algorithmic, %.0f GB/s (R=%d)" % (shape, t * 1e6, nbytes, nbytes / t / 1e9, R))
