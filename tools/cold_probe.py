"""Is the step slowed by cold caches after the dense-Adam stream?  fwd+bwd GPU time with / without optim.step()."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prodsearch_amd import readme_tem_args, synth, ItemTransformerRanker, build_optim
ns = readme_tem_args(dropout=float(os.environ.get('DROPOUT', '0.0')))
P_, V_ = 18357, 32387
wd = synth.make_word_dists(V_)
m = ItemTransformerRanker(ns, 'cuda', V_, P_, None, word_dists=wd); opt = build_optim(ns, m, None); m.train()
b = synth.make_tem_batch(1, 384, P_, V_, Q=8, L=20, W=1, word_dists=wd).to('cuda')
big = torch.empty(64 << 20, device='cuda')   # 256 MB
def run(mode, n=200):
    for _ in range(20):
        loss = m(b); m.zero_grad(); loss.backward(); opt.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        loss = m(b); m.zero_grad(); loss.backward()
        if mode == 'opt': opt.step()
        elif mode == 'thrash': big.add_(1.0)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for mode in ('none', 'opt', 'thrash', 'none'):
    print("fwd+bwd + %-6s: %.1f us/step" % (mode, run(mode)))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): opt.step()
torch.cuda.synchronize(); print("opt.step alone: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
t0 = time.perf_counter()
for _ in range(200): big.add_(1.0)
torch.cuda.synchronize(); print("thrash alone (512 MB traffic): %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
