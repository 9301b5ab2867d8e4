import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch, numpy as np
import test_gpu_fullsize_rtm as T
from golden_util import rel_err
from oracle import rtm as ortm
from oracle.philox import RtmPhiloxDropout

def run(B, K, U, I, WL, corrupt, enc='pvc'):
    T.B, T.K, T.U_LIM, T.I_LIM, T.WL, T.R = B, K, U, I, WL, U + I
    a, sd, m, batch = T._setup(enc, corrupt=corrupt)
    R = U + I
    with torch.no_grad():
        loss = m(batch.to('cuda'), train_pv=False)
        st = T._stages(m)
        gen = RtmPhiloxDropout(0.0, 666, m._fwd_step, B, K, a.heads, R + 1, 1, corrupt)
        keep = {}
        ol, _, _ = ortm.rtm_forward(sd, a, batch, None, T.V, T.RC, training=True, train_pv=False, drop=None,
                                    tok_drop=gen.tok if corrupt > 0 else None, keep=keep)
    mask = torch.cat([keep['pos_mask'].unsqueeze(1), keep['neg_mask']], dim=1)
    seq = torch.cat([keep['pos_seq'].unsqueeze(1), keep['neg_seq']], dim=1) * mask.unsqueeze(-1).float()
    seq = seq + ortm.positional_encoding(5000, T.D)[:R + 1]
    enc_o = torch.cat([keep['enc_pos'].unsqueeze(1), keep['enc_neg']], dim=1)
    ex = (st['x'] - seq).abs()
    worst = ex.amax(-1)
    n_bad = int((worst > 1e-3).sum())
    idx = torch.nonzero(worst > 1e-3)[:5].tolist()
    print("B%d K%d R%d WL%d corrupt %.1f: loss %.2e qe %.2e x %.2e (bad slots %d, first %s) enc %.2e scores %.2e"
          % (B, K, R, WL, corrupt, rel_err(loss.cpu(), ol), rel_err(st['query_emb'], keep['query_emb']),
             rel_err(st['x'], seq), n_bad, idx, rel_err(st['enc'], enc_o), rel_err(st['scores'], keep['scores'])), flush=True)

for cfg in [(8, 2, 3, 4, 20, 0.0), (8, 2, 3, 4, 100, 0.0), (8, 2, 3, 4, 64, 0.0), (8, 2, 3, 4, 65, 0.0), (64, 5, 20, 30, 100, 0.0),
            (256, 5, 20, 30, 100, 0.0), (256, 5, 20, 30, 100, 0.9), (256, 5, 20, 30, 60, 0.9), (8, 2, 3, 4, 100, 0.9)]:
    run(*cfg)
