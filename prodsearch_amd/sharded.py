"""Row-sharded embedding tables with an all-to-all row exchange (the reference has no counterpart —
it is single-process, trainer.py:64-83, with every table on one device).

Why: BASELINE configs[4] replicates a 50 M x 256 item table on each of 8 GPUs — 51 GB of parameters plus the same again
for the dense gradient and each Adam moment (205 GB per GPU).  Sharded by row over the ranks the same table costs 1/world
of that, and the per-step exchange shrinks from "every rank receives every other rank's touched rows" (all-gather,
``dist.SparseGradExchange``) to "every row travels to its one owner" (all-to-all): 1/world of the bytes per link.

Layout: row ``i`` lives on rank ``i % world`` at local row ``i // world`` (interleaved, so Zipf-popular low ids spread
over the ranks); each rank holds ``[ceil(n_rows / world), d]`` parameters + gradient + moments.

One training step on every rank (``lookup`` -> the unchanged HIP step on a compact table -> ``push_grads``):

  1. ``lookup(index tensors)``: sorted unique non-pad row ids this rank's batch addresses (U of them), bucketed by owner;
     all-to-all of the counts, of the requested ids, then of the rows themselves (owners gather locally).  The result is a
     COMPACT table ``[U + 1, d]`` (last row = the zero padding row) and the batch's index tensors REMAPPED into it — the
     step's kernels run on that pair exactly as they run on a replicated table: an embedding lookup only ever sees the
     rows it indexes (tests/test_gpu_c5_shard.py checks this equivalence against the oracle).
  2. forward / backward as usual; the dense gradient of the compact table is small (U rows).
  3. ``push_grads(compact gradient)``: the reverse all-to-all; an owner receives, per requesting rank, (local row, gradient
     row) pairs and sums rows requested by several ranks IN RANK ORDER (deterministic), which gives the touched rows and
     their summed gradients of its shard — the input of the row-sparse clip + Adam (``ps_clip_adam_rowsparse``) on the shard.
     The global clip norm needs one scalar all-reduce of the shards' sums of squares (each row is owned once).

The exchange below is backend-agnostic torch.distributed code (RCCL on the GPUs, gloo in the CPU tests); sizes of the
all-to-all messages are data dependent, so the counts cross to the host once per step and table (the fixed-capacity form
of ``dist.SparseGradExchange`` does not apply: a rank's request to ONE owner has no useful static bound below U).
"""
import torch
import torch.distributed as dist


def _world(group):
    return (dist.get_world_size(group), dist.get_rank(group)) if dist.is_initialized() else (1, 0)


def _all_to_all_rows(send, send_counts, recv_counts, group):
    """Variable-size all-to-all of the rows of ``send`` ([sum(send_counts), ...]) -> [sum(recv_counts), ...]."""
    world, _ = _world(group)
    out = send.new_empty((int(sum(recv_counts)),) + tuple(send.shape[1:]))
    if world == 1:
        out.copy_(send)
        return out
    try:
        dist.all_to_all_single(out, send.contiguous(), output_split_sizes=list(recv_counts),
                               input_split_sizes=list(send_counts), group=group)
    except (RuntimeError, NotImplementedError):          # gloo: no all_to_all_single -> pairwise form over views
        outs = list(out.split(list(recv_counts)))
        ins = list(send.contiguous().split(list(send_counts)))
        dist.all_to_all(outs, ins, group=group)
    return out


class ShardedTable(object):
    """One table ``[n_rows, d]`` (+ a virtual zero padding row ``pad_row``) sharded by ``row % world``."""

    def __init__(self, n_rows, d, pad_row, device='cpu', group=None, dtype=torch.float32):
        self.n_rows, self.d, self.pad_row, self.group = int(n_rows), int(d), int(pad_row), group
        self.world, self.rank = _world(group)
        self.local_rows = (self.n_rows + self.world - 1) // self.world
        self.weight = torch.zeros(self.local_rows, d, device=device, dtype=dtype)
        self.grad = torch.zeros_like(self.weight)

    # ------------------------------------------------------------------ construction helpers
    def load_full(self, full):
        """Take this rank's rows of a full ``[n_rows(+pad), d]`` table (tests / checkpoint import)."""
        rows = torch.arange(self.rank, self.n_rows, self.world, device=full.device)
        self.weight[:rows.numel()].copy_(full[rows])

    def gather_full(self):
        """The full table on every rank (tests / checkpoint export)."""
        parts = [torch.zeros_like(self.weight) for _ in range(self.world)]
        if self.world > 1:
            dist.all_gather(parts, self.weight, group=self.group)
        else:
            parts = [self.weight]
        full = self.weight.new_zeros(self.n_rows, self.d)
        for r, p in enumerate(parts):
            rows = torch.arange(r, self.n_rows, self.world, device=full.device)
            full[rows] = p[:rows.numel()]
        return full

    # ------------------------------------------------------------------ step
    def lookup(self, index_tensors):
        """-> (compact table [U+1, d], remapped index tensors, ctx).  ``ctx`` feeds ``push_grads``."""
        dev = self.weight.device
        flat = torch.cat([t.reshape(-1) for t in index_tensors]) if index_tensors else torch.zeros(0, dtype=torch.int64, device=dev)
        uniq = torch.unique(flat[flat != self.pad_row])                        # sorted
        if uniq.numel() and (int(uniq[0]) < 0 or int(uniq[-1]) >= self.n_rows):
            raise RuntimeError("ShardedTable.lookup: row id outside [0, %d)" % self.n_rows)
        owner = uniq % self.world
        order = torch.argsort(owner, stable=True)                              # requests grouped by owner, ids ascending inside
        req = uniq[order]
        send_counts = torch.bincount(owner, minlength=self.world).tolist()
        recv_counts = self._exchange_counts(send_counts)
        asked = _all_to_all_rows(req // self.world, send_counts, recv_counts, self.group)      # local rows others want from me
        rows = _all_to_all_rows(self.weight[asked], recv_counts, send_counts, self.group)       # ... and mine, back from owners
        compact = self.weight.new_zeros(uniq.numel() + 1, self.d)
        compact[order] = rows                                                   # row u of compact = table row uniq[u]
        U = uniq.numel()
        remapped = []
        for t in index_tensors:
            pos = torch.searchsorted(uniq, t.clamp(0, max(self.n_rows - 1, 0)))
            remapped.append(torch.where(t == self.pad_row, torch.full_like(t, U), pos))
        ctx = dict(order=order, send_counts=send_counts, recv_counts=recv_counts, asked=asked, U=U, uniq=uniq)
        return compact, remapped, ctx

    def push_grads(self, ctx, compact_grad):
        """Route the compact table's gradient rows to their owners and accumulate them into ``self.grad``.
        -> (touched local rows of this shard, sorted unique).  Sums over requesting ranks run in rank order."""
        send = compact_grad[:ctx['U']][ctx['order']]
        got = _all_to_all_rows(send, ctx['send_counts'], ctx['recv_counts'], self.group)         # aligned with ctx['asked']
        asked = ctx['asked']
        touched = torch.unique(asked)
        # deterministic: one index_add per requesting rank (ids are unique inside a rank's request), in rank order
        lo = 0
        for c in ctx['recv_counts']:
            if c:
                self.grad.index_add_(0, asked[lo:lo + c], got[lo:lo + c])
            lo += c
        return touched

    def _exchange_counts(self, send_counts):
        if self.world == 1:
            return list(send_counts)
        dev = self.weight.device
        sc = torch.tensor(send_counts, dtype=torch.int64, device=dev)
        rc = torch.empty_like(sc)
        try:
            dist.all_to_all_single(rc, sc, group=self.group)
        except (RuntimeError, NotImplementedError):
            outs = list(rc.split(1))
            dist.all_to_all(outs, list(sc.split(1)), group=self.group)
        return rc.tolist()                                                      # the step's one host sync per table


def sharded_grad_sumsq(tables, touched, group=None):
    """Sum of squares of the sharded tables' gradients over all ranks (each row is owned once): the tables' share of the
    global clip norm (optimizers.py:241-242)."""
    s = None
    for t, rows in zip(tables, touched):
        v = (t.grad[rows].double() ** 2).sum()
        s = v if s is None else s + v
    if s is None:
        s = torch.zeros((), dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(s, group=group)
    return s
