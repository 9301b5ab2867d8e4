"""Host enqueue time vs GPU time of the training step: python tools/host_overhead.py [--dropout 0.1]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from prodsearch_amd import readme_tem_args, synth
ap = argparse.ArgumentParser(); ap.add_argument('--dropout', type=float, default=0.1); a = ap.parse_args()
ns = readme_tem_args(dropout=a.dropout)
model, optim, wd = bench.make_model(ns, 'cuda', 1234)
model.train()
batches = [synth.make_tem_batch(1000 + i, 384, bench.P_ITEMS, bench.V_WORDS, Q=8, L=20, W=1, word_dists=wd).to('cuda') for i in range(8)]
def step(i):
    loss = model(batches[i % 8]); model.zero_grad(); loss.backward(); optim.step()
for i in range(30): step(i)
torch.cuda.synchronize()
for n in (20, 200):
    t0 = time.perf_counter()
    for i in range(n): step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("n=%d host enqueue %.1f us/step, total %.1f us/step" % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(200): step(i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
