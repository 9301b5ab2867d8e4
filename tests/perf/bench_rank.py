#!/usr/bin/env python3
"""Full-catalogue evaluation throughput: evaluate.rank_all on one MI355X vs the reference's
procedure (model.test over chunks of 500 candidates with the sequence re-encoded per candidate, host argsort) restated
by the oracle on the host cores.

    python tests/perf/bench_rank.py [--items 18357] [--batch 24] [--d 128] [--cpu-batches 1]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from prodsearch_amd import ItemTransformerRanker, evaluate, readme_tem_args, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--items', type=int, default=18357)
    ap.add_argument('--batch', type=int, default=24)          # --valid_batch_size default (main.py)
    ap.add_argument('--d', type=int, default=128)
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--cpu-batches', type=int, default=1)
    a = ap.parse_args()
    V, P, B = 32387, a.items, a.batch
    args = readme_tem_args(embedding_size=a.d, ff_size=4 * a.d)
    wd = synth.make_word_dists(V)
    torch.manual_seed(0)
    model = ItemTransformerRanker(args, 'cuda', V, P, None, word_dists=wd)
    model.eval()
    batches = [synth.make_tem_batch(50 + i, B, P, V, Q=8, L=20, W=1, word_dists=wd).to('cuda') for i in range(4)]
    for i in range(5):
        evaluate.rank_all(model, batches[i % 4], 100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.iters):
        top_idx, top_score, rank = evaluate.rank_all(model, batches[i % 4], 100)
    torch.cuda.synchronize()
    t_gpu = (time.perf_counter() - t0) / a.iters
    out = {"workload": "rank all %d products for %d (user,query) rows, d=%d, top-100 + target rank" % (P, B, a.d),
           "gpu_ms_per_batch": t_gpu * 1e3, "gpu_rows_per_s": B / t_gpu, "gpu_scores_per_s": B * P / t_gpu,
           "gemm_tflops": 2.0 * B * P * a.d / t_gpu / 1e12}
    if a.cpu_batches > 0 and P <= 100000:
        from oracle import tem as otem, rank as orank
        Pm = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        b = synth.make_tem_batch(50, B, P, V, Q=8, L=20, W=1, word_dists=wd)
        t0 = time.perf_counter()
        for _ in range(a.cpu_batches):
            cols = []
            with torch.no_grad():
                for s in range(0, P, 500):                    # candi_batch_size 500, trainer.py:198-216
                    ids = torch.arange(s, min(P, s + 500))
                    b.candi_prod_idxs = ids.unsqueeze(0).expand(B, -1).contiguous()
                    cols.append(otem.tem_test(Pm, args, b, V, P, replicate=True).numpy())
            sc = np.concatenate(cols, axis=1)
            order = sc.argsort(axis=-1)[:, ::-1]
            tgt = b.target_prod_idxs.numpy()
            [int(np.where(order[i] == tgt[i])[0][0]) for i in range(B)]
        t_cpu = (time.perf_counter() - t0) / a.cpu_batches
        # same rows on the device for a spot check of the ranks
        _, _, r = evaluate.rank_all(model, b.to('cuda'), 100)
        _, _, r_cpu = orank.rank_scores(sc, tgt, 100)
        out.update({"cpu_ms_per_batch": t_cpu * 1e3, "cpu_rows_per_s": B / t_cpu, "cpu_threads": torch.get_num_threads(),
                    "speedup": t_cpu / t_gpu, "rank_mismatches": int((r.cpu().numpy() != r_cpu).sum())})
    print(json.dumps(out))


if __name__ == '__main__':
    main()
