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
    rank_round_flat = [True]
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
        flat = pdist._flat_gather_supported(None) and rank_round_flat[0]
        pdist._all_gather_flat(all_rows, msg_rows, world, None, flat)
        pdist._all_gather_flat(all_vals, msg_vals, world, None, flat)
        rank_round_flat[0] = False                             # second round: the list form
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


# ------------------------------------------------------------------ sharded optimizer (reduce-scatter / owner Adam / all-gather)
class _StubModel(torch.nn.Module):
    """The attributes dist.flatten_parameters / ShardedAdamExchange use of the product models, on CPU tensors."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.table = torch.nn.Parameter(torch.randn(37, 8, generator=g))
        self.w = torch.nn.Parameter(torch.randn(6, 5, generator=g))
        self.b = torch.nn.Parameter(torch.randn(7, generator=g))
        self._params_struct = self._grads_struct = None
        self._grad_flat = self._grad_views = None
        self.fields = {}

    def _named_hot_params(self):
        return [(('table',), self.table), (('w',), self.w), (('b',), self.b)]

    def _set_field(self, struct, path, value):
        struct[path] = value

    def _structs(self):
        if self._params_struct is not None:
            return self._params_struct, self._grads_struct
        graded = sorted(self._named_hot_params(), key=lambda t: t[1].numel())
        offs, cur = [], 0
        for _, p in graded:
            offs.append(cur)
            cur += (p.numel() + 3) // 4 * 4
        pad = int(self.__dict__.get('_flat_pad_to', 4))
        cur = (cur + pad - 1) // pad * pad
        self._grad_flat = torch.zeros(cur)
        self._grad_views = [(p, self._grad_flat[o:o + p.numel()].view_as(p)) for (_, p), o in zip(graded, offs)]
        self._params_struct, self._grads_struct = {}, {}
        return self._params_struct, self._grads_struct


class _Hyper(object):
    lr, beta1, beta2, eps, weight_decay, max_grad_norm, grad_scale, zero_grads = 0.01, 0.9, 0.999, 1e-9, 0.0, 5.0, 1.0, 0


def _cpu_sharded_class():
    import math

    class CpuSharded(pdist.ShardedAdamExchange):
        def _load_lib(self):
            return None

        def _build_plan(self, dev):
            self.n_chunks, self.plan, self.state = 1, None, torch.zeros(2, dtype=torch.int64)

        def _k_zero(self, flat):
            flat.zero_()

        def _k_sum_slices(self, recv, out, zero):          # ps_sum_slices restated: rank-ordered sum + clear
            acc = recv[0].clone()
            for r in range(1, recv.shape[0]):
                acc += recv[r]
            out.copy_(acc)
            if zero is not None:
                zero.zero_()

        def _k_sumsq(self, hp):
            self.state[0] += 1
            self.sumsq[0] = float(((self.g_shard * hp.grad_scale) ** 2).sum())

        def _k_update(self, hp):
            t = int(self.state[0])
            norm = math.sqrt(float(self.sumsq[0]))
            coef = min(hp.max_grad_norm / (norm + 1e-6), 1.0)
            g = self.g_shard * (coef * hp.grad_scale)
            self.m_shard.mul_(hp.beta1).add_(g, alpha=1 - hp.beta1)
            self.v_shard.mul_(hp.beta2).addcmul_(g, g, value=1 - hp.beta2)
            denom = self.v_shard.sqrt() / math.sqrt(1 - hp.beta2 ** t) + hp.eps
            self.p_shard.addcdiv_(self.m_shard, denom, value=-hp.lr / (1 - hp.beta1 ** t))
            self.gnorm[0] = norm

    return CpuSharded


def _sharded_worker(rank, world, port, q, forms='rccl'):
    """ShardedAdamExchange's protocol over gloo with a CPU restatement of its three kernels (oracle/optim.py's clip+Adam
    arithmetic): after every step all ranks hold bitwise identical parameters, and they equal a single-process
    clip_grad_norm_ + Adam(eps=1e-9) on the MEAN of the ranks' gradients (optimizers.py:241-243)."""
    import math
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), PS_DP_RS=forms, PS_DP_AG=forms)
    pdist.init_from_env(backend='gloo')

    CpuSharded = _cpu_sharded_class()

    class Opt(object):
        _step, grad_scale, _sharded, _plan = 0, 1.0, None, None

    torch.manual_seed(3)                       # identical start on every rank
    model, opt = _StubModel(), Opt()
    ref = {n: p.detach().clone() for n, p in model.named_parameters()}
    ex = CpuSharded(model, opt)
    assert ex.rs_mode == forms and ex.ag_mode == forms
    ok_layout = model._param_flat.numel() % (4 * world) == 0 and all(
        p.data_ptr() == model._param_flat.data_ptr() + 4 * v.storage_offset() for p, v in model._grad_views)
    # reference: torch's own clip + Adam on the mean gradient
    refp = [torch.nn.Parameter(ref[n].clone()) for n, _ in model.named_parameters()]
    ropt = torch.optim.Adam(refp, lr=_Hyper.lr, betas=(0.9, 0.999), eps=1e-9)
    ok = True
    for step in range(3):
        model._structs()
        grads_all = []
        for r in range(world):                 # every rank can restate every rank's gradient (seeded by rank and step)
            g = torch.Generator().manual_seed(1000 * step + r)
            grads_all.append([torch.randn(p.shape, generator=g) * 3.0 for _, p in model.named_parameters()])
        for (p, v), gr in zip([(p, dict((id(q), w) for q, w in model._grad_views)[id(p)]) for _, p in model.named_parameters()],
                              grads_all[rank]):
            v.copy_(gr)
        ex()
        assert float(model._grad_flat.abs().sum()) == 0.0 and model._grad_clean
        hp = _Hyper()
        hp.grad_scale = opt.grad_scale
        ex.step(hp)
        for i, q_ in enumerate(refp):
            q_.grad = sum(ga[i] for ga in grads_all) / world
        torch.nn.utils.clip_grad_norm_(refp, 5.0)
        ropt.step()
        for (n, p), q_ in zip(model.named_parameters(), refp):
            ok = ok and torch.allclose(p.detach(), q_.detach(), rtol=2e-5, atol=2e-6)
    flats = [torch.zeros_like(model._param_flat) for _ in range(world)]
    dist.all_gather(flats, model._param_flat)
    same = all(torch.equal(flats[0], f) for f in flats)
    mf, vf = ex.full_moments()
    ok_m = all(torch.allclose(mf[v.storage_offset():v.storage_offset() + p.numel()].view_as(p),
                              ropt.state[q_]['exp_avg'], rtol=2e-5, atol=1e-6)
               for (p, v), q_ in zip([(p, dict((id(a), b) for a, b in model._grad_views)[id(p)]) for _, p in model.named_parameters()], refp))
    q.put((rank, bool(ok_layout), bool(ok), bool(same), bool(ok_m), opt.grad_scale))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,forms', [(2, 'rccl'), (2, 'a2a'), (8, 'a2a')])      # the library collectives / slices sent peer to
def test_gloo_sharded_adam_protocol(world, forms):                                   # peer (all-to-all + batched send / recv); 8 = a node's rank count
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q, forms)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert r[1] and r[2] and r[3] and r[4] and r[5] == 1.0 / world, r


def _subgroup_worker(rank, world, port, q):
    """the peer-to-peer all-gather names its peers by GLOBAL rank.  World 4, the exchange on the strict sub-group
    {1, 3} (group rank 0 = global rank 1, group rank 1 = global rank 3): a group-local peer number would send rank 1's slice to
    global rank 1 (itself) / receive from 0 — a hang or slices in the wrong replicas.  The members must end with bitwise equal
    parameters that match clip + Adam on the mean of THEIR two gradients; ranks 0 and 2 take no part."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), PS_DP_RS='a2a', PS_DP_AG='a2a')
    pdist.init_from_env(backend='gloo')
    members = [1, 3]
    grp = dist.new_group(ranks=members)                     # collective over the whole world
    if rank in members:
        CpuSharded = _cpu_sharded_class()

        class Opt(object):
            _step, grad_scale, _sharded, _plan = 0, 1.0, None, None

        torch.manual_seed(3)
        model, opt = _StubModel(), Opt()
        ref = {n: p.detach().clone() for n, p in model.named_parameters()}
        ex = CpuSharded(model, opt, group=grp)
        assert ex.world == 2 and ex.rank == members.index(rank) and ex.ag_mode == 'a2a'
        refp = [torch.nn.Parameter(ref[n].clone()) for n, _ in model.named_parameters()]
        ropt = torch.optim.Adam(refp, lr=_Hyper.lr, betas=(0.9, 0.999), eps=1e-9)
        ok = True
        for step in range(2):
            model._structs()
            grads_all = []
            for r in members:
                g = torch.Generator().manual_seed(1000 * step + r)
                grads_all.append([torch.randn(p.shape, generator=g) * 3.0 for _, p in model.named_parameters()])
            views = dict((id(a), b) for a, b in model._grad_views)
            for (_, p), gr in zip(model.named_parameters(), grads_all[members.index(rank)]):
                views[id(p)].copy_(gr)
            ex()
            hp = _Hyper()
            hp.grad_scale = opt.grad_scale
            ex.step(hp)
            for i, q_ in enumerate(refp):
                q_.grad = sum(ga[i] for ga in grads_all) / len(members)
            torch.nn.utils.clip_grad_norm_(refp, 5.0)
            ropt.step()
            for (n, p), q_ in zip(model.named_parameters(), refp):
                ok = ok and torch.allclose(p.detach(), q_.detach(), rtol=2e-5, atol=2e-6)
        flats = [torch.zeros_like(model._param_flat) for _ in members]
        dist.all_gather(flats, model._param_flat, group=grp)
        q.put((rank, bool(ok), bool(torch.equal(flats[0], flats[1]))))
    else:
        q.put((rank, True, True))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_sharded_adam_on_a_strict_subgroup_of_the_world():
    world = 4
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_subgroup_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert r[1] and r[2], r


def _resume_worker(rank, world, port, q):
    """Save / resume under the sharded optimizer: three steps, then the moments and the step count go through
    ``Optimizer.state_dict()`` / ``load_state_dict()`` semantics — ``full_moments()`` -> per-parameter tensors ->
    ``load_moments()`` of a FRESH exchange over a fresh copy of the model — and step four of the resumed pair must equal step
    four of the uninterrupted pair bit for bit (it used to restart from zero moments with the old step count)."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), PS_DP_RS='rccl', PS_DP_AG='rccl')
    pdist.init_from_env(backend='gloo')
    CpuSharded = _cpu_sharded_class()

    class Opt(object):
        _step, grad_scale, _sharded, _plan, _state_tensors = 0, 1.0, None, None, None

    def grads_into(model, step):
        model._structs()
        g = torch.Generator().manual_seed(1000 * step + rank)
        views = dict((id(a), b) for a, b in model._grad_views)
        for _, p in model.named_parameters():
            views[id(p)].copy_(torch.randn(p.shape, generator=g) * 3.0)

    def run_step(model, opt, ex, step):
        grads_into(model, step)
        ex()
        opt._step += 1
        hp = _Hyper()
        hp.grad_scale = opt.grad_scale
        ex.step(hp)

    torch.manual_seed(3)
    model, opt = _StubModel(), Opt()
    ex = CpuSharded(model, opt)
    for step in range(3):
        run_step(model, opt, ex, step)
    # "checkpoint": parameters, per-parameter moments (what Optimizer.state_dict() builds from full_moments()), the step
    mf, vf = ex.full_moments()
    ck_params = {n: p.detach().clone() for n, p in model.named_parameters()}
    ck_m = {n: (mf[v.storage_offset():v.storage_offset() + p.numel()].view_as(p).clone(),
                vf[v.storage_offset():v.storage_offset() + p.numel()].view_as(p).clone())
            for (n, p), v in zip(model.named_parameters(),
                                 [dict((id(a), b) for a, b in model._grad_views)[id(p)] for _, p in model.named_parameters()])}
    run_step(model, opt, ex, 3)                        # the uninterrupted fourth step
    # resume: a fresh model and optimizer whose state was loaded BEFORE the exchange is built (build_optim(train_from)) ...
    torch.manual_seed(3)
    model2, opt2 = _StubModel(), Opt()
    with torch.no_grad():
        for n, p in model2.named_parameters():
            p.copy_(ck_params[n])
    opt2._step = 3
    opt2._state_tensors = {id(p): ck_m[n] for n, p in model2.named_parameters()}
    ex2 = CpuSharded(model2, opt2)
    ok_loaded = int(ex2.state[0]) == 3 and float(ex2.m_shard.abs().sum()) > 0
    run_step(model2, opt2, ex2, 3)
    same = all(torch.equal(p.detach(), q_.detach()) for (_, p), (_, q_) in zip(model.named_parameters(), model2.named_parameters()))
    # ... and load_moments() called on a LIVE exchange (Optimizer.load_state_dict after make_exchange) does the same
    torch.manual_seed(3)
    model3, opt3 = _StubModel(), Opt()
    with torch.no_grad():
        for n, p in model3.named_parameters():
            p.copy_(ck_params[n])
    ex3 = CpuSharded(model3, opt3)
    ex3.load_moments({id(p): ck_m[n] for n, p in model3.named_parameters()}, 3)
    opt3._step = 3
    run_step(model3, opt3, ex3, 3)
    same3 = all(torch.equal(p.detach(), q_.detach()) for (_, p), (_, q_) in zip(model.named_parameters(), model3.named_parameters()))
    q.put((rank, bool(ok_loaded), bool(same), bool(same3)))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_sharded_adam_resume_keeps_moments_and_step():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_resume_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert r[1] and r[2] and r[3], r
