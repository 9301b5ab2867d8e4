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


def _all_gather_flat(out, inp, world, group):
    """``out`` [world, ...] <- every rank's ``inp`` [...].  One collective; backends without the flat form
    (older gloo) get the list form over views of the same buffer."""
    try:
        dist.all_gather_into_tensor(out.view(-1), inp.view(-1), group=group)
    except (RuntimeError, NotImplementedError):
        dist.all_gather([out[r].view(-1) for r in range(world)], inp.view(-1), group=group)


class SparseGradExchange(object):
    """Data-parallel exchange in row-sparse mode (``args.row_sparse_adam``) with NO host synchronisation.

    Per step: one all-reduce of the dense head of the flat gradient buffer (small tensors, 0.86 MB at C2), and per
    table two all-gathers of fixed-capacity messages — the rank's sorted touched row ids (then -1) and their gradient
    rows — whose capacity is the step's index count, a function of the batch SHAPE only.  ``ps_coalesce_rows`` over
    the gathered ids (pad -1) gives the union, ``ps_merge_rows`` writes the rank-ordered sums into the dense gradient
    (bitwise identical on every rank), and the union becomes the touched list the optimizer walks, so clip norm and
    Adam see identical data on every rank without another collective.  xGMI is point-to-point: an all-gather keeps
    all 7 links busy where a ring all-reduce of a table-sized buffer would be bound by one."""

    def __init__(self, model, optim=None, group=None):
        self.model = model
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if optim is not None:
            optim.grad_scale = 1.0 / self.world

    def _buffers(self, info, p, cap):
        x = info.get('xchg')
        if x is None or x['cap'] != cap or x['world'] != self.world:
            from . import _lib
            lib = _lib.load()
            dev, d, W = p.device, p.shape[1], self.world
            ucap = max(1, min(W * cap, p.shape[0]))
            x = dict(cap=cap, world=W, ucap=ucap,
                     msg_rows=torch.empty(cap, device=dev, dtype=torch.int64),
                     msg_vals=torch.empty(cap, d, device=dev, dtype=torch.float32),
                     all_rows=torch.empty(W, cap, device=dev, dtype=torch.int64),
                     all_vals=torch.empty(W, cap, d, device=dev, dtype=torch.float32),
                     urows=torch.empty(ucap, device=dev, dtype=torch.int64),
                     ucount=torch.zeros(1, device=dev, dtype=torch.int32),
                     ws=torch.zeros(lib.ps_coalesce_ws_bytes(p.shape[0]), device=dev, dtype=torch.uint8))
            info['xchg'] = x
        return x

    def __call__(self):
        if self.world == 1:
            return None
        from . import _lib
        lib = _lib.load()
        m = self.model
        dist.all_reduce(m._grad_flat[:m._n_dense_grad], op=dist.ReduceOp.SUM, group=self.group)
        for _, p, gview in m._sparse_tabs:
            info = p._ps_rows
            cap, d = int(info['cap']), p.shape[1]
            x = self._buffers(info, p, cap)
            st = torch.cuda.current_stream(p.device).cuda_stream
            _lib.check(lib.ps_pack_rows(gview.data_ptr(), d, info['rows'].data_ptr(), info['count'].data_ptr(), cap,
                                        x['msg_rows'].data_ptr(), x['msg_vals'].data_ptr(), st), 'ps_pack_rows')
            _all_gather_flat(x['all_rows'], x['msg_rows'], self.world, self.group)
            _all_gather_flat(x['all_vals'], x['msg_vals'], self.world, self.group)
            lst = (_lib.PsIdxList * 1)()
            lst[0].idx, lst[0].n = x['all_rows'].data_ptr(), self.world * cap
            _lib.check(lib.ps_coalesce_rows(lst, 1, p.shape[0], -1, x['ws'].data_ptr(), x['urows'].data_ptr(),
                                            x['ucap'], x['ucount'].data_ptr(), st), 'ps_coalesce_rows')
            _lib.check(lib.ps_merge_rows(x['all_rows'].data_ptr(), x['all_vals'].data_ptr(), self.world, cap, d,
                                         gview.data_ptr(), x['urows'].data_ptr(), x['ucount'].data_ptr(), x['ucap'],
                                         st), 'ps_merge_rows')
            info['active'] = (x['urows'], x['ucount'], x['ucap'])     # what the optimizer / zero_grad walk this step
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
