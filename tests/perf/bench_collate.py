#!/usr/bin/env python3
"""Batch-builder throughput: the C++ collate behind prodsearch_amd.dataloader.ItemPVDataloader
against the Python collate of the reference (restated in oracle/collate.py) on a synthetic Amazon-shaped corpus,
and — on an MI355X — the training step fed by the native loader against pre-built device batches.

    python tests/perf/bench_collate.py [--users 40000] [--batches 200] [--gpu-steps 300]
"""
import argparse
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from oracle import collate as ocollate  # noqa: E402  (CPU baseline of this tool only)
from prodsearch_amd import readme_tem_args, synth  # noqa: E402
from prodsearch_amd.dataloader import ItemPVDataloader  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--users', type=int, default=40000)
    ap.add_argument('--batches', type=int, default=200)
    ap.add_argument('--py-batches', type=int, default=20)
    ap.add_argument('--gpu-steps', type=int, default=300)
    a = ap.parse_args()
    B, P, V = 384, 18357, 32387
    args = readme_tem_args(fix_train_review=False)
    t0 = time.perf_counter()
    train_ds, _ = synth.make_corpus(7, n_users=a.users, n_products=P, n_queries=2000, vocab_size=V, Q=8, W=1,
                                    max_reviews_per_user=400)
    t_corpus = time.perf_counter() - t0
    t0 = time.perf_counter()
    dl = ItemPVDataloader(args, train_ds, batch_size=B, shuffle=True, seed=1)
    t_flat = time.perf_counter() - t0
    it = iter(dl)
    next(it)
    t0 = time.perf_counter()
    for _ in range(a.batches):
        next(it)
    t_native = (time.perf_counter() - t0) / a.batches
    random.seed(1)
    ids = list(range(B * a.py_batches))
    random.shuffle(ids)
    t0 = time.perf_counter()
    for k in range(a.py_batches):
        batch = [train_ds[i] for i in ids[k * B:(k + 1) * B]]
        d = ocollate.train_batch(train_ds, args, batch)
        for key in ('query_word_idxs', 'target_prod_idxs', 'u_item_idxs', 'pos_iword_idxs'):
            torch.tensor(d[key])                      # ItemPVBatch.to_tensor (batch_data.py:17-22)
    t_py = (time.perf_counter() - t0) / a.py_batches
    out = {"workload": "get_train_batch B=%d uprev=20 random history subset, %d users / %d reviews / %d train samples"
                       % (B, a.users, len(train_ds.global_data.review_u_p), len(train_ds)),
           "native_us_per_batch": t_native * 1e6, "native_tuples_per_s": B * 20 / t_native,
           "python_us_per_batch": t_py * 1e6, "python_tuples_per_s": B * 20 / t_py, "speedup": t_py / t_native,
           "corpus_flatten_s": t_flat, "cores": 1}
    if torch.cuda.is_available() and a.gpu_steps > 0:
        from prodsearch_amd import ItemTransformerRanker, build_optim
        wd = synth.make_word_dists(V)
        torch.manual_seed(0)
        model = ItemTransformerRanker(args, 'cuda', V, P, None, word_dists=wd)
        optim = build_optim(args, model, None)
        model.train()
        gdl = ItemPVDataloader(args, train_ds, batch_size=B, shuffle=True, seed=1, device='cuda', drop_last=True)
        pdl = ItemPVDataloader(args, train_ds, batch_size=B, shuffle=True, seed=1, device='cuda', drop_last=True, prefetch=3)

        def run(batches):
            n = 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for b in batches:
                loss = model(b)
                model.zero_grad()
                loss.backward()
                optim.step()
                n += 1
                if n == a.gpu_steps:
                    break
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n

        pre = []
        for b in gdl:
            pre.append(b)
            if len(pre) == 16:
                break
        run(pre * 3)                                   # warm-up
        t_pre = run(pre * (a.gpu_steps // 16 + 1))
        t_live = run(iter(gdl))
        t_pref = run(iter(pdl))
        out.update({"step_ms_prebuilt_batches": t_pre * 1e3, "step_ms_native_loader": t_live * 1e3,
                    "step_ms_native_loader_prefetch3": t_pref * 1e3,
                    "tuples_per_s_native_loader": B * 20 / min(t_live, t_pref)})
    print(json.dumps(out))


if __name__ == '__main__':
    main()
