#!/usr/bin/env python3
"""Review-transformer batch builder: the native loader (prodsearch_amd.rtm_loader, C++ collate + device-side review-word
gather) against the reference's own ProdSearchDataLoader (imported from /root/reference when it is there — this
container only) on one synthetic gz corpus at the reference's default shapes (bs 256 here as in BASELINE configs[3],
5 negatives, 20 + 30 reviews, 100 words), and — on an MI355X — the ProductRanker step fed by the loader.

    python tests/perf/bench_rtm_collate.py [--users 20000] [--batches 30] [--gpu-steps 100]
"""
import argparse
import json
import os
import random
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from prodsearch_amd import default_args, pyrandom, synth  # noqa: E402
from prodsearch_amd.corpus import GlobalProdSearchData, ProdSearchData, ProdSearchDataset  # noqa: E402
from prodsearch_amd.rtm_loader import ProdSearchDataLoader  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--users', type=int, default=20000)
    ap.add_argument('--products', type=int, default=2000)
    ap.add_argument('--batches', type=int, default=30)
    ap.add_argument('--ref-batches', type=int, default=3)
    ap.add_argument('--gpu-steps', type=int, default=100)
    a = ap.parse_args()
    B = 256
    args = default_args(model_name='review_transformer', review_encoder_name='pvc', batch_size=B, neg_per_pos=5,
                        uprev_review_limit=20, iprev_review_limit=30, review_word_limit=100, embedding_size=128,
                        ff_size=512, heads=8, inter_layers=1, has_valid=True, valid_candi_size=500, candi_batch_size=500,
                        valid_batch_size=24)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()
        data_path, inp = synth.write_corpus(tmp, 5, n_users=a.users, n_products=a.products, n_words=8000, n_queries=500,
                                            max_reviews=60)
        gd = GlobalProdSearchData(args, data_path, inp)
        train_pd = ProdSearchData(args, inp, 'train', gd)
        valid_pd = ProdSearchData(args, inp, 'valid', gd)
        out['corpus_s'] = time.perf_counter() - t0
        ref = None
        if os.path.isdir('/root/reference'):
            sys.path.insert(0, '/root/reference')
            from data.data_util import GlobalProdSearchData as RG, ProdSearchData as RP
            from data.prod_search_dataset import ProdSearchDataset as RD
            from data.prod_search_dataloader import ProdSearchDataLoader as RL
            rgd = RG(args, data_path, inp)
            ref = (rgd, RP(args, inp, 'train', rgd), RP(args, inp, 'valid', rgd), RD, RL)
    pyrandom.seed(1); np.random.seed(1); torch.manual_seed(1)
    t0 = time.perf_counter()
    train_pd.initialize_epoch()
    out['initialize_epoch_s'] = time.perf_counter() - t0
    ds = ProdSearchDataset(args, gd, train_pd)
    dev = 'cuda' if torch.cuda.is_available() else None
    t0 = time.perf_counter()
    dl = ProdSearchDataLoader(args, ds, prepare_pv=False, batch_size=B, shuffle=True, device=dev)
    out['loader_setup_s'] = time.perf_counter() - t0
    it = iter(dl)
    next(it)
    t0 = time.perf_counter()
    shapes = []
    for _ in range(a.batches):
        b = next(it)
        shapes.append(tuple(b.neg_prod_ridxs.shape))
    if dev:
        torch.cuda.synchronize()
    t_native = (time.perf_counter() - t0) / a.batches
    out.update(workload="get_train_batch B=%d K=5 uprev=20 iprev=30 WL=100, %d users / %d products / %d reviews / %d train rows"
                        % (B, gd.user_size, gd.product_size, gd.review_count - 1, len(ds)),
               native_ms_per_batch=t_native * 1e3, neg_shape_example=list(shapes[-1]), cores=1)
    vds = ProdSearchDataset(args, gd, valid_pd)
    vdl = ProdSearchDataLoader(args, vds, batch_size=24, shuffle=False, device=dev)
    vit = iter(vdl)
    next(vit)
    t0 = time.perf_counter()
    for _ in range(5):
        next(vit)
    if dev:
        torch.cuda.synchronize()
    out['native_test_ms_per_batch_24x500'] = (time.perf_counter() - t0) / 5 * 1e3
    if ref is not None:
        rgd, rtrain, rvalid, RD, RL = ref
        random.seed(1); np.random.seed(1); torch.manual_seed(1)
        t0 = time.perf_counter()
        rtrain.initialize_epoch()
        out['reference_initialize_epoch_s'] = time.perf_counter() - t0
        rdl = RL(args, RD(args, rgd, rtrain), prepare_pv=False, batch_size=B, shuffle=True, num_workers=0)
        rit = iter(rdl)
        next(rit)
        t0 = time.perf_counter()
        for _ in range(a.ref_batches):
            next(rit)
        t_ref = (time.perf_counter() - t0) / a.ref_batches
        out.update(reference_ms_per_batch=t_ref * 1e3, speedup=t_ref / t_native)
        rv = iter(RL(args, RD(args, rgd, rvalid), batch_size=24, shuffle=False, num_workers=0))
        next(rv)
        t0 = time.perf_counter()
        next(rv)
        out['reference_test_ms_per_batch_24x500'] = (time.perf_counter() - t0) * 1e3
    if dev and a.gpu_steps > 0:
        from prodsearch_amd import ProductRanker, build_optim
        args.device = 'cuda'
        torch.manual_seed(0)
        model = ProductRanker(args, 'cuda', gd.vocab_size, gd.review_count, gd.product_size, gd.user_size, gd.review_words,
                              gd.words, word_dists=train_pd.word_dists)
        optim = build_optim(args, model, None)
        model.train()

        def run(batches, n_steps):
            n = 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for b in batches:
                if b is None:
                    continue
                loss = model(b, train_pv=False)
                model.zero_grad()
                loss.backward()
                optim.step()
                n += 1
                if n == n_steps:
                    break
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n

        pre = [next(it) for _ in range(8)]
        run(pre, 8)
        out['step_ms_prebuilt_batches'] = run(pre * (a.gpu_steps // 8 + 1), a.gpu_steps) * 1e3
        out['step_ms_native_loader'] = run(iter(dl), a.gpu_steps) * 1e3
        pdl = ProdSearchDataLoader(args, ds, prepare_pv=False, batch_size=B, shuffle=True, device=dev, prefetch=3)
        out['step_ms_native_loader_prefetch3'] = run(iter(pdl), a.gpu_steps) * 1e3
    print(json.dumps(out))


if __name__ == '__main__':
    main()
