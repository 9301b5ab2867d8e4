import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, numpy as np
import test_gpu_fullsize_rtm as T
from oracle import rtm as ortm

def run(B, K, U, I, WL):
    T.B, T.K, T.U_LIM, T.I_LIM, T.WL, T.R = B, K, U, I, WL, U + I
    a, sd, m, batch = T._setup('pvc', corrupt=0.0)
    R = U + I
    with torch.no_grad():
        loss = m(batch.to('cuda'), train_pv=False)
        st = T._stages(m)
        keep = {}
        ortm.rtm_forward(sd, a, batch, None, T.V, T.RC, training=True, train_pv=False, keep=keep)
    pe = ortm.positional_encoding(5000, T.D)[:R + 1]
    seg = sd['seg_embeddings.weight']
    we = sd['word_embeddings.weight']
    print("WL", WL)
    for r in range(R):
        words = batch.pos_prod_rword_idxs[0, r]
        valid = words != T.V - 1
        n = int(valid.sum())
        if batch.pos_prod_ridxs[0, r] == T.RC - 1:
            print(r, 'pad review'); continue
        mean = we[words[valid]].sum(0) / max(n, 1)
        got = st['x'][0, 0, r + 1] - pe[r + 1] - seg[batch.pos_seg_idxs[0, r + 1]]
        orc = keep['pos_rev'][0, r]
        # which count would explain got?  got*k = sum => k estimate by least squares against sum
        ssum = we[words[valid]].sum(0)
        k_est = float((ssum * ssum).sum() / (ssum * got).sum())
        first32 = we[words[:32][valid[:32]]].sum(0)
        print(r, 'nvalid', n, 'err_vs_torch %.3e' % float((got - mean).abs().max()), 'oracle_vs_torch %.3e' % float((orc - mean).abs().max()),
              'k_est %.2f' % k_est, 'err if only first 32 slots summed/ n: %.3e' % float((got - first32 / n).abs().max()))

for cfg in [(8, 2, 3, 4, 20), (8, 2, 3, 4, 65)]:
    run(*cfg)
