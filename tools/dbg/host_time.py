"""Host time per step (enqueue only) against the device step: is the bench loop host-bound?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
sys.argv = ['bench.py'] + sys.argv[1:]
a = bench.parse()
dev = torch.device('cuda', 0)
wl = bench.TemWorkload(a, 0, dev)
model, optim = wl.model, wl.optim
model.train()
def step(i):
    loss = wl.forward(i); model.zero_grad(); loss.backward(); optim.step(); return loss
for i in range(30): step(i)
torch.cuda.synchronize()
for N in (300, 300):
    t0 = time.perf_counter()
    for i in range(N): step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("steps %d: enqueue %.1f us/step, total %.1f us/step, drain after loop %.1f us" % (N, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6, (t2 - t1) * 1e6))
# host-only cost: same loop with a sync after each step -> per-step host+device serial
parts = dict(fwd=0.0, zero=0.0, bwd=0.0, opt=0.0)
for i in range(200):
    t = time.perf_counter(); loss = wl.forward(i); parts['fwd'] += time.perf_counter() - t
    t = time.perf_counter(); model.zero_grad(); parts['zero'] += time.perf_counter() - t
    t = time.perf_counter(); loss.backward(); parts['bwd'] += time.perf_counter() - t
    t = time.perf_counter(); optim.step(); parts['opt'] += time.perf_counter() - t
    if i % 8 == 7: torch.cuda.synchronize()
print("host us per call:", {k: round(v / 200 * 1e6, 1) for k, v in parts.items()})
