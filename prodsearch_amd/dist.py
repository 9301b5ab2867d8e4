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
