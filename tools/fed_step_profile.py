"""Host-side time of the loader-fed training step, by part (MI355X): python tools/fed_step_profile.py [steps] [prefetch]
Every part is host wall time between perf_counter() reads — the GPU runs behind asynchronously; the last line is the whole loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prodsearch_amd import readme_tem_args, synth, ItemTransformerRanker, build_optim
from prodsearch_amd.dataloader import ItemPVDataloader

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
prefetch = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B, P, V = 384, 18357, 32387
args = readme_tem_args(fix_train_review=False)
train_ds, _ = synth.make_corpus(7, n_users=20000, n_products=P, n_queries=2000, vocab_size=V, Q=8, W=1, max_reviews_per_user=400)
wd = synth.make_word_dists(V)
torch.manual_seed(0)
model = ItemTransformerRanker(args, 'cuda', V, P, None, word_dists=wd)
optim = build_optim(args, model, None)
model.train()
dl = ItemPVDataloader(args, train_ds, batch_size=B, shuffle=True, seed=1, device='cuda', drop_last=True, prefetch=prefetch)
it = iter(dl)
for _ in range(50):
    b = next(it); loss = model(b); model.zero_grad(); loss.backward(); optim.step()
torch.cuda.synchronize()
T = dict(next=0.0, fwd=0.0, zero=0.0, bwd=0.0, opt=0.0)
pc = time.perf_counter
t_all = pc()
for _ in range(steps):
    t0 = pc(); b = next(it)
    t1 = pc(); loss = model(b)
    t2 = pc(); model.zero_grad()
    t3 = pc(); loss.backward()
    t4 = pc(); optim.step()
    t5 = pc()
    T['next'] += t1 - t0; T['fwd'] += t2 - t1; T['zero'] += t3 - t2; T['bwd'] += t4 - t3; T['opt'] += t5 - t4
torch.cuda.synchronize()
t_all = pc() - t_all
print("prefetch %d: " % prefetch + "  ".join("%s %.1f us" % (k, v / steps * 1e6) for k, v in T.items()) + "  | loop %.1f us/step" % (t_all / steps * 1e6))
