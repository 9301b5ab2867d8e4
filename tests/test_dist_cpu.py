"""N>1 path on CPU: world_size-2 gloo processes exercise the gradient exchange the GPU ranks use
(prodsearch_amd/dist.py): flat-buffer all-reduce + 1/world scale hand-off, parameter broadcast,
max-over-ranks timing, per-rank Philox keys."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from prodsearch_amd import dist as pdist


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeOptim(object):
    grad_scale = 1.0


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    r, l, w = pdist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)
    flat = torch.randn(1000)
    mine = flat.clone()
    opt = _FakeOptim()
    ex = pdist.GradExchange(lambda: flat, opt)
    ex()
    # every rank must now hold the SAME summed buffer, and the optimizer the mean scale
    gathered = [torch.zeros(1000) for _ in range(world)]
    dist.all_gather(gathered, mine)
    want = sum(gathered)
    ok_sum = torch.allclose(flat, want, atol=1e-6) and opt.grad_scale == 1.0 / world
    # replicas start identical after broadcast
    lin = torch.nn.Linear(4, 3)
    pdist.broadcast_parameters(lin, src=0)
    ws = [torch.zeros_like(lin.weight) for _ in range(world)]
    dist.all_gather(ws, lin.weight.data)
    ok_bc = all(torch.equal(ws[0], x) for x in ws)
    mx = pdist.max_over_ranks(float(rank + 1), torch.device('cpu'))
    q.put((rank, bool(ok_sum), bool(ok_bc), mx, pdist.rank_seed(666, rank)))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_gradient_exchange():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert all(r[1] and r[2] for r in res)
    assert all(r[3] == float(world) for r in res)          # max over ranks
    assert res[0][4] != res[1][4]                           # per-rank Philox key
    assert res[0][4] == 666                                 # rank 0 keeps the base seed


def _sparse_worker(rank, world, port, q):
    """The wire protocol of dist.SparseGradExchange on CPU tensors: fixed-capacity messages (sorted ids then -1; rows
    then zeros) through dist._all_gather_flat, then the union / rank-ordered merge the HIP kernels perform
    (ps_coalesce_rows + ps_merge_rows; restated here in numpy as the checker) against a dense all-reduce."""
    import numpy as np
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    pdist.init_from_env(backend='gloo')
    n_rows, d, cap = 500, 16, 96
    gen = torch.Generator().manual_seed(7 + rank)
    out = []
    # ragged: rank 0 touches 37 rows, rank 1 touches 90 (some shared), plus an empty-list round
    for n_touch in ((37, 90)[rank], 0 if rank == 0 else 5):
        rows = torch.sort(torch.randperm(n_rows, generator=gen)[:n_touch]).values
        vals = torch.randn(n_touch, d, generator=gen)
        dense = torch.zeros(n_rows, d)
        dense[rows] = vals
        msg_rows = torch.full((cap,), -1, dtype=torch.int64)
        msg_vals = torch.zeros(cap, d)
        msg_rows[:n_touch] = rows
        msg_vals[:n_touch] = vals
        all_rows = torch.empty(world, cap, dtype=torch.int64)
        all_vals = torch.empty(world, cap, d)
        pdist._all_gather_flat(all_rows, msg_rows, world, None)
        pdist._all_gather_flat(all_vals, msg_vals, world, None)
        ar, av = all_rows.numpy(), all_vals.numpy()
        union = np.unique(ar[ar >= 0])
        merged = np.zeros((n_rows, d), np.float32)
        for r in range(world):                                 # rank order: identical sums on every rank
            live = ar[r] >= 0
            merged[ar[r][live]] += av[r][live]
        dist.all_reduce(dense)                                 # what a dense exchange would have produced
        rows_all = [None] * world
        dist.all_gather_object(rows_all, rows.tolist())
        want_union = sorted(set(sum(rows_all, [])))
        mt = torch.from_numpy(merged)
        gathered = [torch.zeros_like(mt) for _ in range(world)]
        dist.all_gather(gathered, mt)
        out.append((union.tolist() == want_union, bool(torch.allclose(mt, dense, atol=1e-6)),
                    all(torch.equal(gathered[0], x) for x in gathered)))      # bitwise identical replicas
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_sparse_row_exchange():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_sparse_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, out in res:
        assert all(all(t) for t in out), out


def test_single_process_is_a_no_op():
    flat = torch.ones(8)
    opt = _FakeOptim()
    ex = pdist.GradExchange(lambda: flat, opt)
    assert ex() is None and opt.grad_scale == 1.0 and float(flat.sum()) == 8.0
    assert pdist.max_over_ranks(3.5, torch.device('cpu')) == 3.5
