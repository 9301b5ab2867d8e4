"""Data-parallel exchange of the step's gradients: one process per GPU, RCCL over xGMI.

The reference has no distributed code; this is new design.  The path shards
by batch rows: every rank runs the same step on its own ``B`` rows (weak scaling), all tables
and weights are replicated, and because both loss terms are batch means
(item_transformer.py:282,514) the data-parallel gradient is the mean of the per-rank
gradients.  The module keeps all gradients in ONE flat fp32 buffer (small tensors first,
tables last), so the exchange is a single collective and the fused clip+Adam kernel applies
the ``1/world`` scale (``Optimizer.grad_scale``) — the global clip norm is then computed on the
reduced gradient, identically on every rank, with no extra collective.

``backend='nccl'`` IS RCCL on ROCm; ``gloo`` is used by the CPU tests of this logic.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Rendezvous from torchrun's environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = os.environ.get('PS_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            # one process per GPU; the modulo only matters for single-GPU dry runs of the N>1 path (gloo)
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def rank_seed(base_seed, rank):
    """Per-rank Philox key: negatives and dropout masks are independent across ranks."""
    return (int(base_seed) & 0xFFFFFFFF) | ((int(rank) + 1) << 32) if rank else int(base_seed)


class GradExchange(object):
    """Sum the flat gradient buffer over ranks; the optimizer divides by the world size.

    ``flat_getter`` returns the flat fp32 gradient tensor (``model._grad_flat`` after a
    backward).  Works on any device/backend, which is what the gloo CPU tests exercise."""

    def __init__(self, flat_getter, optim=None, group=None):
        self.flat_getter = flat_getter
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if optim is not None:
            optim.grad_scale = 1.0 / self.world

    def __call__(self):
        if self.world == 1:
            return None
        flat = self.flat_getter()
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)


def exchange_rows(rows, vals, group=None):
    """Sparse all-reduce of one table's gradient: ``rows`` [u] sorted unique int64 row ids and their
    gradient rows ``vals`` [u, d] on this rank -> (union of the ranks' rows, sorted; summed values).

    all-gather of counts, then of (rows, vals) padded to the largest count — xGMI is point-to-point, so an
    all-gather keeps all 7 links busy where a ring all-reduce of the dense table would move table-sized
    buffers.  The merge adds the ranks' contributions in rank order (each rank's rows are unique, so every
    index_add is collision-free): bitwise identical on every rank, which keeps the replicas in lock step."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return rows, vals
    dev, d = rows.device, vals.shape[1]
    cnt = torch.tensor([rows.numel()], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    cnts = [int(c) for c in cnts]
    mx = max(max(cnts), 1)
    prow = torch.full((mx,), -1, dtype=torch.int64, device=dev)
    pval = torch.zeros(mx, d, dtype=vals.dtype, device=dev)
    prow[:rows.numel()] = rows
    pval[:rows.numel()] = vals
    all_rows = [torch.empty_like(prow) for _ in range(world)]
    all_vals = [torch.empty_like(pval) for _ in range(world)]
    dist.all_gather(all_rows, prow, group=group)
    dist.all_gather(all_vals, pval, group=group)
    union = torch.unique(torch.cat([r[:c] for r, c in zip(all_rows, cnts)]))     # sorted
    acc = torch.zeros(union.numel(), d, dtype=vals.dtype, device=dev)
    for r, v, c in zip(all_rows, all_vals, cnts):
        if c:
            acc.index_add_(0, torch.searchsorted(union, r[:c]), v[:c])
    return union, acc


class SparseGradExchange(object):
    """Data-parallel exchange in row-sparse mode (``args.row_sparse_adam``): one all-reduce of the dense
    head of the flat gradient buffer (small tensors, 0.86 MB at C2) + ``exchange_rows`` per table.
    The merged rows are written back into the dense gradient tensors and become the touched list the
    optimizer consumes, so clip norm and Adam are computed on identical data on every rank."""

    def __init__(self, model, optim=None, group=None):
        self.model = model
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if optim is not None:
            optim.grad_scale = 1.0 / self.world

    def __call__(self):
        if self.world == 1:
            return None
        m = self.model
        dist.all_reduce(m._grad_flat[:m._n_dense_grad], op=dist.ReduceOp.SUM, group=self.group)
        for _, p, gview in m._sparse_tabs:
            info = p._ps_rows
            rows = info['rows'][:int(info['count'][0])]
            union, acc = exchange_rows(rows, gview[rows], self.group)
            gview[union] = acc
            info['rows'] = union
            info['count'] = torch.tensor([union.numel()], dtype=torch.int32, device=union.device)
            info['cap'] = max(1, union.numel())
        return None


def make_exchange(model, optim=None, group=None):
    """The exchange matching the model's optimizer mode."""
    if getattr(model, '_row_sparse', lambda: False)():
        return SparseGradExchange(model, optim, group)
    return GradExchange(lambda: model._grad_flat, optim, group)


def broadcast_parameters(model, src=0, group=None):
    """Replicas must start identical (tables and weights replicated)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def max_over_ranks(value, device):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])
