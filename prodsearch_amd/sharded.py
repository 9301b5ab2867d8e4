"""Row-sharded item table with a device-side all-to-all row exchange (the reference has no
counterpart — it is single-process, trainer.py:64-83, with the whole table on one device, item_transformer.py:46,464-469).

Why: BASELINE configs[4] replicates a 50 M x 256 item table on each of 8 GPUs — 51 GB of parameters plus the same again for
the dense gradient and each Adam moment (205 GB per GPU).  Sharded by row over the ranks the same table costs 1/world of
that, and the per-step exchange shrinks from "every rank receives every other rank's touched rows" (all-gather,
``dist.SparseGradExchange``) to "every row travels to its one owner" (all-to-all): 1/world of the bytes per link.

Layout: row ``i`` lives on rank ``i % world`` at local row ``i // world`` (interleaved, so Zipf-popular low ids spread over
the ranks); each rank holds ``[ceil(n_rows / world), d]`` parameters + gradient + both Adam moments.

One training step (``args.shard_tables``; ``ItemTransformerRanker`` drives it, no host synchronisation anywhere):

  1. ``lookup(index tensors)`` — ``ps_coalesce_rows``: the sorted unique rows this rank's batch addresses; ``ps_shard_bucket``:
     one FIXED-capacity request per owner (capacity = the step's index count / world with 25 % headroom: ownership is
     interleaved, so a rank's requests spread evenly; an overflow sets the status word read by ``check_index_errors``);
     two equal-split all-to-alls (the requested local rows, then the rows themselves, gathered by their owners with
     ``ps_gather_rows``); ``ps_shard_remap`` rewrites the batch's index tensors into SLOTS of the receive buffer, which is
     the table the step's kernels read: ``[world * capp + 1, d]``, last row = the zero padding row.  An embedding lookup
     only ever sees the rows it indexes (tests/test_gpu_c5_shard.py checks this equivalence against the oracle).
  2. forward / backward as usual on that buffer; its dense gradient has the same (small) shape.
  3. ``push_grads()`` — the reverse all-to-all of the gradient buffer; the owner turns the received requests into its touched
     local rows (``ps_coalesce_rows``, pad -1) and sums rows requested by several ranks IN RANK ORDER (``ps_merge_rows``:
     bitwise reproducible) into its shard's dense gradient.
  4. the row-sparse clip + Adam on the shard (``Optimizer._step_rows``): ``ps_rowsparse_sumsq`` -> ONE scalar all-reduce of the
     shards' sums of squares (each row is owned once; the replicated tensors count once) -> ``ps_rowsparse_update_ext``.
     The next ``lookup`` reads the updated rows.
"""
import torch
import torch.distributed as dist

from . import _lib


def _world(group):
    return (dist.get_world_size(group), dist.get_rank(group)) if dist.is_initialized() else (1, 0)


class ShardedItemTable(object):
    """One table ``[n_rows, d]`` (+ a virtual zero padding row ``pad_row``) sharded by ``row % world``; ``cap`` = the
    largest number of indices one step of this rank addresses (a function of the batch SHAPE)."""

    def __init__(self, n_rows, d, pad_row, cap, device, group=None, headroom=1.25):
        self.n_rows, self.d, self.pad_row, self.group = int(n_rows), int(d), int(pad_row), group
        self.world, self.rank = _world(group)
        self.cap = max(1, min(int(cap), self.n_rows))
        W = self.world
        self.capp = self.cap if W == 1 else min(self.cap, int(self.cap / W * headroom) + 64)     # per (requester, owner) pair
        self.slots = W * self.capp                                                              # rows of the receive buffer (+ 1 pad row)
        self.local_rows = (self.n_rows + W - 1) // W
        dev = self.device = torch.device(device)
        f32, i64, i32 = torch.float32, torch.int64, torch.int32
        self.weight = torch.zeros(self.local_rows, d, device=dev, dtype=f32)       # this rank's shard: parameters,
        self.grad = torch.zeros(self.local_rows, d, device=dev, dtype=f32)         # dense gradient (touched rows non-zero between
        self.m = torch.zeros(self.local_rows, d, device=dev, dtype=f32)            # push_grads and the optimizer step), Adam moments
        self.v = torch.zeros(self.local_rows, d, device=dev, dtype=f32)
        lib = _lib.load()
        self.rows = torch.empty(self.cap, device=dev, dtype=i64)                   # the step's sorted unique rows
        self.count = torch.zeros(1, device=dev, dtype=i32)
        self.co_ws = torch.zeros(lib.ps_coalesce_ws_bytes(self.n_rows), device=dev, dtype=torch.uint8)
        self.slot_of = torch.empty(self.cap, device=dev, dtype=i32)
        self.send_ids = torch.empty(W, self.capp, device=dev, dtype=i64)
        self.asked = torch.empty(W, self.capp, device=dev, dtype=i64)              # local rows each rank wants from me (-1 padded)
        self.send_rows = torch.empty(self.slots, d, device=dev, dtype=f32)
        self.table_buf = torch.zeros(self.slots + 1, d, device=dev, dtype=f32)     # what the kernels read (re-pointed by attach())
        self.ggot = torch.empty(self.slots, d, device=dev, dtype=f32)
        ucap = max(1, min(self.slots, self.local_rows))
        self.urows = torch.empty(ucap, device=dev, dtype=i64)                      # owner side: touched local rows of the step
        self.ucount = torch.zeros(1, device=dev, dtype=i32)
        self.ucap = ucap
        self.co_ws2 = torch.zeros(lib.ps_coalesce_ws_bytes(self.local_rows), device=dev, dtype=torch.uint8)
        self.bad = torch.zeros(1, device=dev, dtype=i32)
        self._remap_out = {}
        self.pending = False             # a backward's gradient buffer waits to be pushed to the owners
        # the overflow / dropped-index status word reaches the host ONE lookup late without a synchronisation: every lookup
        # queues an asynchronous 4-byte copy into pinned memory, the next lookup reads what the previous one left (its copy
        # finished a whole step ago) — round 3 only looked where the trainer read the loss, up to steps_per_checkpoint steps
        # of silently wrong embeddings later
        self._bad_host = torch.zeros(1, dtype=torch.int32).pin_memory() if dev.type == 'cuda' else None
        self._bad_evt = None

    def attach(self, param):
        """Use ``param`` (``[slots + 1, d]``, the model's ``product_emb.weight``) as the receive buffer."""
        assert tuple(param.shape) == (self.slots + 1, self.d) and param.is_contiguous()
        self.table_buf = param.data
        self.table_buf.zero_()

    def _st(self):
        return torch.cuda.current_stream(self.device).cuda_stream if self.device.type == 'cuda' else None

    # ------------------------------------------------------------------ construction helpers
    def init_normal(self, seed):
        """N(0,1) rows like nn.Embedding's default; every rank draws its own shard."""
        g = torch.Generator(device=self.device).manual_seed(int(seed) + 7919 * self.rank)
        for i in range(0, self.local_rows, 1 << 22):
            self.weight[i:i + (1 << 22)].normal_(generator=g)

    def load_full(self, full, which='weight'):
        """Take this rank's rows of a full ``[n_rows(+pad), d]`` tensor (tests / checkpoint import); ``which``: ``weight``, or
        ``m`` / ``v`` for the Adam moments of an optimizer checkpoint."""
        rows = torch.arange(self.rank, self.n_rows, self.world, device=full.device)
        getattr(self, which)[:rows.numel()].copy_(full[rows].to(self.device))

    def gather_full(self, which='weight'):
        """The full ``[n_rows, d]`` tensor on every rank (tests / checkpoint export of small tables; a collective)."""
        src = getattr(self, which)
        parts = [torch.zeros_like(src) for _ in range(self.world)]
        if self.world > 1:
            dist.all_gather(parts, src, group=self.group)
        else:
            parts = [src]
        full = src.new_zeros(self.n_rows, self.d)
        for r, p in enumerate(parts):
            rows = torch.arange(r, self.n_rows, self.world, device=full.device)
            full[rows] = p[:rows.numel()]
        return full

    # ------------------------------------------------------------------ device operations (overridable: tests/test_sharded_cpu.py
    # drives the exchange protocol over gloo with a torch restatement of these five kernels)
    def _k_coalesce(self, tensors, n_rows, pad, ws, rows, cap, count):
        lib = _lib.load()
        lists = (_lib.PsIdxList * len(tensors))()
        for i, t in enumerate(tensors):
            lists[i].idx, lists[i].n = t.data_ptr(), t.numel()
        _lib.check(lib.ps_coalesce_rows(lists, len(tensors), n_rows, pad, ws.data_ptr(), rows.data_ptr(), cap, count.data_ptr(),
                                        self._st()), 'ps_coalesce_rows')

    def _k_bucket(self):
        _lib.check(_lib.load().ps_shard_bucket(self.rows.data_ptr(), self.count.data_ptr(), self.world, self.capp,
                                               self.send_ids.data_ptr(), self.slot_of.data_ptr(), self.bad.data_ptr(), self._st()),
                   'ps_shard_bucket')

    def _k_gather(self, out):
        _lib.check(_lib.load().ps_gather_rows(self.weight.data_ptr(), self.d, self.asked.data_ptr(), None, self.slots,
                                              out.data_ptr(), self._st()), 'ps_gather_rows')

    def _k_remap(self, t, out):
        _lib.check(_lib.load().ps_shard_remap(t.data_ptr(), t.numel(), self.pad_row, self.rows.data_ptr(), self.count.data_ptr(),
                                              self.slot_of.data_ptr(), self.slots, out.data_ptr(), self.bad.data_ptr(), self._st()),
                   'ps_shard_remap')

    def _k_merge(self, got):
        _lib.check(_lib.load().ps_merge_rows(self.asked.data_ptr(), got.data_ptr(), self.world, self.capp, self.d,
                                             self.grad.data_ptr(), self.urows.data_ptr(), self.ucount.data_ptr(), self.ucap,
                                             self._st()), 'ps_merge_rows')

    # ------------------------------------------------------------------ step
    def lookup(self, index_tensors):
        """Fetch the rows the index tensors address into the receive buffer; -> the tensors remapped into its slots."""
        W = self.world
        if self.pending:
            raise RuntimeError("ShardedItemTable.lookup: the last backward's gradient still sits in the receive buffer (slots are "
                               "per lookup): call optimizer.step() before the next forward / encode / test")
        self._poll_flag()
        total = sum(t.numel() for t in index_tensors)
        if total > self.cap and self.cap < self.n_rows:
            raise RuntimeError("ShardedItemTable.lookup: %d indices but the table was built for %d per step "
                               "(args.batch_size / neg_per_pos / uprev_review_limit)" % (total, self.cap))
        self._k_coalesce(index_tensors, self.n_rows, self.pad_row, self.co_ws, self.rows, self.cap, self.count)
        self._k_bucket()
        if W > 1:
            from .dist import _comm
            with _comm('all_to_all(row requests)'):
                dist.all_to_all_single(self.asked.view(-1), self.send_ids.view(-1), group=self.group)
            self._k_gather(self.send_rows)                  # owners gather the requested rows (-1 entries give zero rows)
            with _comm('all_to_all(rows)'):
                dist.all_to_all_single(self.table_buf[:self.slots].view(-1), self.send_rows.view(-1), group=self.group)
        else:                                               # nobody to exchange with: straight into the receive buffer
            self.asked.copy_(self.send_ids)
            self._k_gather(self.table_buf)
        out = []
        for k, t in enumerate(index_tensors):
            o = self._remap_out.get((k, t.numel()))
            if o is None:
                o = self._remap_out[(k, t.numel())] = torch.empty(t.numel(), device=self.device, dtype=torch.int64)
            self._k_remap(t, o)
            out.append(o.view(t.shape))
        self._post_flag()
        return out

    def _post_flag(self):
        if self._bad_host is not None:
            src = self.bad
            if self.world > 1:
                # the status is per rank but every rank must fail in the SAME lookup (a rank that raised alone would leave the
                # others blocked in the next all_to_all until the collective's timeout): max over the ranks, on the stream, no sync
                if getattr(self, '_bad_all', None) is None:
                    self._bad_all = torch.zeros_like(self.bad)
                self._bad_all.copy_(self.bad)
                dist.all_reduce(self._bad_all, op=dist.ReduceOp.MAX, group=self.group)
                src = self._bad_all
            self._bad_host.copy_(src, non_blocking=True)
            self._bad_evt = torch.cuda.Event()
            self._bad_evt.record(torch.cuda.current_stream(self.device))

    def _poll_flag(self):
        """Raise if the PREVIOUS lookup overflowed a request or dropped an index (no stall: its status copy is a step old)."""
        if self._bad_evt is None:
            return
        self._bad_evt.synchronize()
        self._bad_evt = None
        flag = int(self._bad_host[0])
        if flag:
            self.bad.zero_()
            raise RuntimeError("sharded item table (previous lookup, on this or another rank — every rank raises here together; that step's "
                               "forward, backward and update ran with zero rows in place of the missing ones: restore the last checkpoint): "
                               + ("an index was not in the step's row list" if flag == 1 else
                                  "a request to one owner overflowed its capacity of %d rows (skewed ids: raise the headroom)" % self.capp))

    def push_grads(self, grad_buf):
        """Route the receive buffer's gradient rows (``[slots + 1, d]``) to their owners and merge them into ``self.grad``;
        ``urows[:ucount]`` = the shard's touched local rows (sorted)."""
        if self.world > 1:
            from .dist import _comm
            with _comm('all_to_all(gradient rows)'):
                dist.all_to_all_single(self.ggot.view(-1), grad_buf[:self.slots].reshape(-1), group=self.group)
            got = self.ggot
        else:
            got = grad_buf[:self.slots]
        self._k_coalesce([self.asked], self.local_rows, -1, self.co_ws2, self.urows, self.ucap, self.ucount)
        self._k_merge(got)
        self.pending = False

    def row_table(self):
        """The owned shard as the optimizer's ``PsRowTable``."""
        t = _lib.PsRowTable()
        t.p, t.g, t.m, t.v = self.weight.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr()
        t.rows, t.count, t.cap, t.d = self.urows.data_ptr(), self.ucount.data_ptr(), self.ucap, self.d
        return t

    def check_errors(self):
        """Raise if a step dropped an index or overflowed a request (host sync: call where the loss is read)."""
        flag = int(self.bad[0])
        if flag:
            self.bad.zero_()
            raise RuntimeError("sharded item table: " + ("an index was not in the step's row list" if flag == 1 else
                               "a request to one owner overflowed its capacity of %d rows (skewed ids: raise the headroom)" % self.capp))
        for ws, n in ((self.co_ws, self.n_rows), (self.co_ws2, self.local_rows)):
            lib = _lib.load()
            off = lib.ps_coalesce_bad_flag(ws.data_ptr(), n) - ws.data_ptr()
            f = int(ws[off:off + 4].view(torch.int32)[0])
            if f:
                ws[off:off + 4].zero_()
                raise RuntimeError("sharded item table: " + ("an index outside [0, %d)" % n if f == 1 else "row list overflow"))
