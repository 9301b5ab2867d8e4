#!/usr/bin/env python3
"""BASELINE configs[3]: review_transformer (RTM) d=128, review-sequence encoder path, bs=256, one MI355X.

Synthetic Amazon-shaped inputs: 296k reviews, V=32,387, R = 20 user + 30 item reviews,
100 words per review, K=5 negatives (main.py:125), pvc review encoder with corrupt_rate 0.9 and dropout
0.1 (reference defaults), train_pv False (main.py: train_pv_epoch 0).  One step = trainer.py:74-78.

    python tools/bench_rtm.py [--steps N] [--encoder pvc|pv] [--layers 1]
Prints one JSON line: sequences/s, (q,u,i,neg) tuples/s, ms/step and the algorithmic gather bytes of the
review-vector kernel (rtm_embed_kernel).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from prodsearch_amd import ProductRanker, build_optim, default_args, synth, rtm_data  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--encoder', default='pvc')
    ap.add_argument('--layers', type=int, default=1)
    ap.add_argument('--batch', type=int, default=256)
    a = ap.parse_args()
    V, RC, B, K, WL, u, i = 32387, 296000, a.batch, 5, 100, 20, 30
    ns = default_args(model_name='review_transformer', review_encoder_name=a.encoder, embedding_size=128, heads=8,
                      ff_size=512, inter_layers=a.layers, neg_per_pos=K, dropout=0.1, corrupt_rate=0.9, lr=0.0005,
                      review_word_limit=WL, uprev_review_limit=u, iprev_review_limit=i)
    wd = synth.make_word_dists(V)
    rng = synth.rng_for(5)
    rw = torch.from_numpy(rng.integers(0, V - 1, size=(RC, WL)))
    lens = torch.from_numpy(rng.integers(WL // 4, WL + 1, size=RC))
    rw[torch.arange(WL)[None, :] >= lens[:, None]] = V - 1
    rw[-1] = V - 1
    torch.manual_seed(0)
    model = ProductRanker(ns, 'cuda', V, RC, 1000, 1000, rw, None, word_dists=wd)
    optim = build_optim(ns, model, None)
    model.train()
    batches = [rtm_data.make_rtm_batch(100 + s, B, K, RC, V, rw, Q=8, u_lim=u, i_lim=i, W=1, train_pv=False,
                                       encoder=a.encoder, word_dists=wd).to('cuda') for s in range(4)]

    def step(s):
        loss = model(batches[s % 4], train_pv=False)
        model.zero_grad()
        loss.backward()
        optim.step()
        return loss

    for s in range(a.warmup):
        step(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(a.steps):
        loss = step(s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    b0 = batches[0]
    n_rev = int((b0.pos_prod_ridxs != RC - 1).sum() + (b0.neg_prod_ridxs != RC - 1).sum())
    if a.encoder == 'pvc':
        words = int((b0.pos_prod_rword_idxs != V - 1).sum() + (b0.neg_prod_rword_idxs != V - 1).sum())
    else:
        words = n_rev
    print(json.dumps({"workload": "review_transformer d=128 %d layer(s) bs=%d K=%d R=50 WL=100 %s (BASELINE configs[3])"
                      % (a.layers, B, K, a.encoder), "ms_per_step": dt * 1e3,
                      "sequences_per_s": B * (K + 1) / dt, "tuples_per_s": B * K / dt,
                      "review_vector_rows_per_step": words, "review_vector_gather_MB": words * 512 / 1e6,
                      "final_loss": float(loss.detach())}))


if __name__ == '__main__':
    main()
