"""One rank of the data-parallel equivalence test (tests/test_gpu_dp.py): started as a FRESH process per rank (nothing
touches the GPU before the rendezvous), gloo over 127.0.0.1, every rank on cuda:0.  Runs `steps` training steps of the
trainer's call order (trainer.py:74-78) on its slice of the global batch with injected negatives and dropout 0, then
writes its parameters (small tensors whole, tables as the rows the global batch touches) to <out>.rank<r>.npz."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def global_problem(mode, Bg, seed=31):
    """Model arguments + the GLOBAL batch and negative draws, identical in every process."""
    from prodsearch_amd import readme_tem_args, synth
    P_, V, K, L, Q, W = 18357, 32387, 20, 20, 8, 1
    a = readme_tem_args(dropout=0.0, lr=0.002, row_sparse_adam=(mode in ('sparse', 'sharded')),      # 'dense' / 'allreduce': dense Adam
                        shard_tables=(mode == 'sharded'), lazy_exact_adam=(mode == 'lazy'))          # 'lazy': row-sparse machinery, dense results
    wd = synth.make_word_dists(V)
    batch = synth.make_tem_batch(seed, Bg, P_, V, Q=Q, L=L, W=W, word_dists=wd)
    ni, nw = synth.sample_negatives(seed + 1, Bg, K, W, P_, wd)
    return a, wd, P_, V, batch, ni, nw


def slice_batch(batch, ni, nw, lo, hi):
    import copy
    b = copy.copy(batch)
    for k in ('query_word_idxs', 'target_prod_idxs', 'u_item_idxs', 'pos_iword_idxs'):
        setattr(b, k, getattr(batch, k)[lo:hi].contiguous())
    b.query_idxs, b.user_idxs = batch.query_idxs[lo:hi], batch.user_idxs[lo:hi]
    return b, ni[lo:hi].contiguous(), nw[lo:hi].contiguous()


def run(mode, Bg, steps, rank, world, exchange_factory, ragged=0):
    import numpy as np
    import torch
    from prodsearch_amd import ItemTransformerRanker, build_optim
    a, wd, P_, V, batch, ni, nw = global_problem(mode, Bg)
    from prodsearch_amd import synth
    per = Bg // world
    a.batch_size = per                                   # (sizes the sharded table's per-step capacity)
    torch.manual_seed(1234)
    m = ItemTransformerRanker(a, 'cuda', V, P_, None, word_dists=wd)
    # identical weights in every mode and process (the sharded model draws a different number of random values at construction)
    sd0 = synth.make_state_dict(synth.tem_param_shapes(a, V, P_), 123, {'product_emb.weight': P_})
    m.load_state_dict(sd0, strict=False)
    optim = build_optim(a, m, None)
    exchange = exchange_factory(m, optim)
    # --ragged n: the LAST rank's slice is n rows short (unequal per-rank batches, e.g. a loader with drop_last=False)
    b, bi, bw = slice_batch(batch, ni, nw, rank * per, (rank + 1) * per - (ragged if rank == world - 1 else 0))
    b, bi, bw = b.to('cuda'), bi.cuda(), bw.cuda()
    m.train()
    times = []
    for s in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = m(b, neg_item_idxs=bi, neg_word_idxs=bw)
        m.zero_grad()
        loss.backward()
        exchange()
        optim.step()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    m.check_index_errors()
    items = np.setdiff1d(np.unique(np.concatenate([batch.target_prod_idxs.numpy().ravel(), ni.numpy().ravel(),
                                                   batch.u_item_idxs.numpy().ravel()])), [P_])
    words = np.setdiff1d(np.unique(np.concatenate([batch.query_word_idxs.numpy().ravel(), nw.numpy().ravel(),
                                                   batch.pos_iword_idxs.numpy().ravel()])), [V - 1])
    out = {'__loss': np.float32(float(loss.detach())), '__ms': np.float64(1e3 * min(times))}
    sd_out = m.state_dict()                              # (sharded item table: gathered from the ranks' shards)
    for n, p in m.named_parameters():
        t = sd_out[n].detach()
        if n == 'product_emb.weight':
            out[n] = t[torch.from_numpy(items).cuda()].cpu().numpy()
            out[n + '__sum'] = np.float64(float(t.double().sum()))
        elif n == 'word_embeddings.weight':
            out[n] = t[torch.from_numpy(words).cuda()].cpu().numpy()
            out[n + '__sum'] = np.float64(float(t.double().sum()))
        else:
            out[n] = t.cpu().numpy()
    return out


def run_rtm(Bg, steps, rank, world, exchange_factory):
    """The same for the review transformer (pvc encoder, no PV loss: the product-score loss is a batch mean,
    ps_model.py:344-358; dropout and token corruption 0)."""
    import copy
    import numpy as np
    import torch
    from prodsearch_amd import ProductRanker, build_optim, default_args, synth, rtm_data
    V, RC, K, WL, U, I = 3000, 2500, 3, 40, 6, 8
    a = default_args(model_name='review_transformer', review_encoder_name='pvc', embedding_size=128, heads=8, ff_size=512,
                     inter_layers=1, neg_per_pos=K, dropout=0.0, corrupt_rate=0.0, lr=0.002, review_word_limit=WL,
                     uprev_review_limit=U, iprev_review_limit=I)
    wd = synth.make_word_dists(V)
    rng = synth.rng_for(17)
    rw = torch.from_numpy(rng.integers(0, V - 1, size=(RC, WL)))
    lens = torch.from_numpy(rng.integers(1, WL + 1, size=RC))
    rw[torch.arange(WL)[None, :] >= lens[:, None]] = V - 1
    rw[-1] = V - 1
    torch.manual_seed(4321)
    m = ProductRanker(a, 'cuda', V, RC, 50, 60, rw, None, word_dists=wd)
    optim = build_optim(a, m, None)
    exchange = exchange_factory(m, optim)
    batch = rtm_data.make_rtm_batch(91, Bg, K, RC, V, rw, Q=6, u_lim=U, i_lim=I, W=1, train_pv=False, encoder='pvc',
                                    word_dists=wd)
    per = Bg // world
    b = copy.copy(batch)
    for k, v in vars(batch).items():
        if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == Bg:
            setattr(b, k, v[rank * per:(rank + 1) * per].contiguous())
    b = b.to('cuda')
    m.train()
    for s in range(steps):
        loss = m(b, train_pv=False)
        m.zero_grad()
        loss.backward()
        exchange()
        optim.step()
    torch.cuda.synchronize()
    out = {'__loss': np.float32(float(loss.detach())), '__ms': np.float64(0.0)}
    for n, p in m.named_parameters():
        out[n] = p.detach().cpu().numpy()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='dense')
    ap.add_argument('--model', default='tem')
    ap.add_argument('--global-batch', type=int, default=384)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--out', required=True)
    ap.add_argument('--ragged', type=int, default=0)
    a = ap.parse_args()
    import numpy as np
    import torch
    from prodsearch_amd import dist as pdist
    rank, local, world = pdist.init_from_env(backend='gloo')
    if a.model == 'rtm':
        out = run_rtm(a.global_batch, a.steps, rank, world, lambda m, o: pdist.make_exchange(m, o))
    else:
        xmode = 'allreduce' if a.mode == 'allreduce' else None
        out = run(a.mode, a.global_batch, a.steps, rank, world, lambda m, o: pdist.make_exchange(m, o, mode=xmode), ragged=a.ragged)
    np.savez(a.out + '.rank%d.npz' % rank, **out)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
