"""One rank of the sharded full-catalogue evaluation test (tests/test_gpu_sharded_eval.py): a model with a row-sharded item table
and an unsharded replica of the same parameters rank the same (rank-specific) eval batch; both results are saved for the test."""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

P_, V, B, L, C_ = 6001, 3000, 24, 12, 100


def run(rank, world, out=None, topk=50):
    from prodsearch_amd import ItemTransformerRanker, evaluate, readme_tem_args, synth
    wd = synth.make_word_dists(V)
    res = {}
    sd0 = None
    for mode in ('plain', 'sharded'):
        a = readme_tem_args(dropout=0.0, batch_size=B, row_sparse_adam=True, shard_tables=(mode == 'sharded'))
        m = ItemTransformerRanker(a, 'cuda', V, P_, None, word_dists=wd)
        if sd0 is None:
            sd0 = synth.make_state_dict(synth.tem_param_shapes(a, V, P_), 77, {'product_emb.weight': P_})
        m.load_state_dict(sd0, strict=False)
        batch = synth.make_tem_batch(4000 + rank, B, P_, V, Q=6, L=L, W=1, word_dists=wd).to('cuda')
        if rank == 0:
            batch.target_prod_idxs[3] = P_          # a row whose target is not a product: rank 0, never ahead of anything
        ti, ts, rk = evaluate.rank_all(m, batch, topk)
        # test() with explicit candidate lists (trainer.py:125-160): C = 100 columns do not fit the exchange's 984 indices beside
        # the 288 history ids, so the sharded model scores them in column chunks of 29
        g = torch.Generator().manual_seed(900 + rank)
        batch.candi_prod_idxs = torch.randint(0, P_, (B, C_), generator=g).cuda()
        batch.candi_prod_idxs[:, -3:] = P_           # padded tails, as the loader's ragged candidate lists have
        batch.candi_prod_idxs[5, 40:] = P_
        sc = m.test(batch)
        torch.cuda.synchronize()
        res[mode] = (ti.cpu().numpy(), ts.cpu().numpy(), rk.cpu().numpy(), sc.cpu().numpy())
        if mode == 'sharded':
            m.check_index_errors()
    if out:
        np.savez(out + '.rank%d.npz' % rank, pi=res['plain'][0], ps=res['plain'][1], pr=res['plain'][2],
                 si=res['sharded'][0], ss=res['sharded'][1], sr=res['sharded'][2],
                 pc=res['plain'][3], sc=res['sharded'][3])
    return res


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    a = ap.parse_args()
    from prodsearch_amd import dist as pdist
    pdist.init_from_env(backend='gloo')
    import torch.distributed as dist
    run(dist.get_rank(), dist.get_world_size(), a.out)
    dist.barrier()
    dist.destroy_process_group()
