"""Run only the embedding-gather+score launch (ps_gather_score) in a loop: target for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from prodsearch_amd import _lib, readme_tem_args, synth
drop = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
ns = readme_tem_args(dropout=drop)
model, optim, wd = bench.make_model(ns, 'cuda', 1234)
model.train()
b = synth.make_tem_batch(1000, bench.B, bench.P_ITEMS, bench.V_WORDS, Q=bench.Q, L=bench.L, W=bench.W, word_dists=wd).to('cuda')
with torch.no_grad():
    model(b)
plan = next(iter(model._plans.values()))
t = bench.time_gather_score(model, plan, 200)
print("gather+score: %.2f us/launch, %.0f GB/s algorithmic (R=%d)" % (t * 1e6, bench.gather_score_bytes(plan.layout.R) / t / 1e9, plan.layout.R))
