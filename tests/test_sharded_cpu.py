"""Row-sharded tables (prodsearch_amd/sharded.py) on CPU: world_size-2 gloo processes check the
all-to-all index / row exchange against a replicated table — the compact table a rank receives holds exactly the rows its
batch addresses, the remapped indices address them, and the gradients routed back accumulate, in every owner's shard, to
the gradient a replicated table would have received from both ranks."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from prodsearch_amd import dist as pdist
from prodsearch_amd.sharded import ShardedTable, sharded_grad_sumsq


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    pdist.init_from_env(backend='gloo')
    n_rows, d, pad = 1001, 8, 1001
    full = torch.randn(n_rows, d, generator=torch.Generator().manual_seed(5))            # identical on every rank
    tab = ShardedTable(n_rows, d, pad)
    tab.load_full(full)
    ok = [bool(torch.equal(tab.gather_full(), full))]                                     # shard <-> full round trip
    gen = torch.Generator().manual_seed(100 + rank)
    ref_grad = torch.zeros(n_rows, d)
    for step in range(3):
        # ragged per-rank batches with shared rows, repeats and padding entries; rank 1's last step addresses nothing
        n = (37, 90)[rank] if not (rank == 1 and step == 2) else 0
        tgt = torch.randint(0, n_rows, (n,), generator=gen)
        hist = torch.randint(0, n_rows, (n, 5), generator=gen)
        hist[torch.rand(n, 5, generator=gen) < 0.3] = pad
        if n:
            tgt[:3] = torch.tensor([0, 1, n_rows - 1])                                    # rows both ranks address
        compact, (tgt2, hist2), ctx = tab.lookup([tgt, hist])
        # the compact table + remapped indices reproduce the replicated lookup (pad -> zero row)
        padded = torch.cat([full, torch.zeros(1, d)], 0)
        ok.append(bool(torch.equal(compact[tgt2], padded[tgt]) and torch.equal(compact[hist2], padded[hist])))
        ok.append(compact.shape[0] == ctx['U'] + 1 and float(compact[-1].abs().max()) == 0.0)
        # a made-up gradient of the compact table, routed back to the owners
        g = torch.randn(compact.shape[0], d, generator=gen)
        g[-1] = 0                                                                          # padding_idx row: no gradient
        tab.grad.zero_()
        touched = tab.push_grads(ctx, g)
        mine = torch.zeros(n_rows, d)
        mine[ctx['uniq']] = g[:ctx['U']]
        dist.all_reduce(mine)                                                              # what a replicated table would hold
        rows = torch.arange(rank, n_rows, world)
        ok.append(bool(torch.allclose(tab.grad[:rows.numel()], mine[rows], atol=1e-6)))
        want_touched = torch.nonzero(mine[rows].abs().sum(1) > 0).flatten()
        ok.append(bool(torch.equal(touched, want_touched)))
        ss = sharded_grad_sumsq([tab], [touched])
        ok.append(abs(float(ss) - float((mine.double() ** 2).sum())) < 1e-6 * max(1.0, float(ss)))
        ref_grad += mine
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_sharded_table_exchange():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok in res:
        assert all(ok), (rank, ok)


def test_single_process_sharded_table_is_the_table():
    full = torch.randn(50, 4)
    tab = ShardedTable(50, 4, 50)
    tab.load_full(full)
    idx = torch.tensor([3, 50, 7, 3])
    compact, (r,), ctx = tab.lookup([idx])
    assert torch.equal(compact[r], torch.cat([full, torch.zeros(1, 4)])[idx])
    g = torch.ones(compact.shape[0], 4)
    touched = tab.push_grads(ctx, g)
    assert touched.tolist() == [3, 7] and float(tab.grad.sum()) == 8.0
