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


# ------------------------------------------------------------------ communication clock (measurement only)
# bench.py --gpus N: a HIP event pair around every collective of the step, recorded on the stream the collective is issued
# from (torch's process group makes that stream wait for the collective's own stream when the call returns), so that the
# step's communication time can be reported beside its compute time.  Off (the default) the context manager is one
# attribute test.
class _CommClock(object):
    enabled = False
    spans = []


def comm_timing(on):
    _CommClock.enabled = bool(on) and torch.cuda.is_available()
    _CommClock.spans = []


class _comm(object):
    __slots__ = ('name', 'e0')

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _CommClock.enabled:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if _CommClock.enabled:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _CommClock.spans.append((self.name, self.e0, e1))
        return False


def comm_timing_read():
    """{collective name: total ms since comm_timing(True)} (synchronises); clears the spans."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    out = {}
    for name, e0, e1 in _CommClock.spans:
        out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
    _CommClock.spans = []
    return out


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
        with _comm('all_reduce(flat gradient)'):
            return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)


def _flat_gather_supported(group=None):
    """Chosen ONCE from the backend, never by catching an exception around a live collective (a rank that fails inside a
    collective and then enters a different one desynchronises the group and hides the real error).  nccl (= RCCL) and
    current gloo have ``all_gather_into_tensor``; PS_DIST_LIST_GATHER=1 forces the list form."""
    if os.environ.get('PS_DIST_LIST_GATHER', '0') not in ('', '0'):
        return False
    return hasattr(dist, 'all_gather_into_tensor') and dist.get_backend(group) in ('nccl', 'gloo')


def _all_gather_flat(out, inp, world, group, flat=True):
    """``out`` [world, ...] <- every rank's ``inp`` [...].  One collective: the flat form where the backend has it
    (``flat``, decided at construction), else the list form over views of the same buffer."""
    if flat:
        dist.all_gather_into_tensor(out.view(-1), inp.view(-1), group=group)
    else:
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
        self._flat = _flat_gather_supported(group) if self.world > 1 else True
        self._capacity = {}          # table -> message capacity every rank agreed on (rows)
        if optim is not None:
            optim.grad_scale = 1.0 / self.world

    def _agree_capacity(self, p, bound):
        """all_gather_into_tensor needs the SAME message size on every rank, but a rank's index count depends on its
        batch (ragged last batch with drop_last=False, history width padded to the batch's own maximum).  The capacity is
        therefore agreed ONCE per table, the first time the exchange runs (every rank is in that step): MAX over ranks of
        the largest index count the rank's batch size allows under the model's flags (``model._index_cap_bound``), one
        host sync.  Later steps never re-negotiate — a collective only some ranks enter would hang — so a step whose
        local count exceeds the agreement raises before any collective is entered."""
        t = torch.tensor([int(bound)], dtype=torch.int64, device=p.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        cap = max(1, min(int(t[0]), p.shape[0]))
        self._capacity[id(p)] = cap
        return cap

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
        # capacities first (first step only), and the local check, BEFORE any collective of this step is entered
        caps = []
        for path, p, gview in m._sparse_tabs:
            local = int(p._ps_rows['cap'])
            cap = self._capacity.get(id(p))
            if cap is None:
                cap = self._agree_capacity(p, max(local, m._index_cap_bound(path)))
            if local > cap:
                raise RuntimeError("row-sparse exchange: this rank's step addresses up to %d rows of a table but the ranks "
                                   "agreed on messages of %d rows at the first step; build the exchange after setting "
                                   "args.batch_size / uprev_review_limit to the largest batch any rank will see" % (local, cap))
            caps.append(cap)
        with _comm('all_reduce(small tensors)'):
            dist.all_reduce(m._grad_flat[:getattr(m, '_n_allreduce_grad', m._n_dense_grad)], op=dist.ReduceOp.SUM, group=self.group)
        for (_, p, gview), cap in zip(m._sparse_tabs, caps):
            info = p._ps_rows
            d = p.shape[1]
            x = self._buffers(info, p, cap)
            st = torch.cuda.current_stream(p.device).cuda_stream
            # the rank's list holds at most info['cap'] <= cap rows: the message is padded with -1 / zeros up to cap
            _lib.check(lib.ps_pack_rows(gview.data_ptr(), d, info['rows'].data_ptr(), info['count'].data_ptr(), cap,
                                        x['msg_rows'].data_ptr(), x['msg_vals'].data_ptr(), st), 'ps_pack_rows')
            with _comm('all_gather(row ids + rows)'):
                _all_gather_flat(x['all_rows'], x['msg_rows'], self.world, self.group, self._flat)
                _all_gather_flat(x['all_vals'], x['msg_vals'], self.world, self.group, self._flat)
            lst = (_lib.PsIdxList * 1)()
            lst[0].idx, lst[0].n = x['all_rows'].data_ptr(), self.world * cap
            _lib.check(lib.ps_coalesce_rows(lst, 1, p.shape[0], -1, x['ws'].data_ptr(), x['urows'].data_ptr(),
                                            x['ucap'], x['ucount'].data_ptr(), st), 'ps_coalesce_rows')
            _lib.check(lib.ps_merge_rows(x['all_rows'].data_ptr(), x['all_vals'].data_ptr(), self.world, cap, d,
                                         gview.data_ptr(), x['urows'].data_ptr(), x['ucount'].data_ptr(), x['ucap'],
                                         st), 'ps_merge_rows')
            info['active'] = (x['urows'], x['ucount'], x['ucap'])     # what the optimizer / zero_grad walk this step
        return None


def flatten_parameters(model, multiple=4):
    """Re-home every graded hot-path parameter of ``model`` into ONE flat fp32 buffer laid out exactly like the model's
    flat gradient buffer (small tensors first, tables last, 16-byte aligned slices, total length a multiple of
    ``multiple``).  The parameters stay ordinary ``nn.Parameter``s with the reference's names — only their storage
    moves — and the C-ABI tensor table is refreshed.  Returns the flat parameter buffer."""
    model.__dict__['_flat_pad_to'] = int(multiple)
    for p in model.parameters():
        p.grad = None
    model._params_struct = None                 # (re)build the flat gradient buffer with the padding
    model._grads_struct = None
    model.__dict__['_grad_clean'] = False
    ps, _ = model._structs()
    gflat = model._grad_flat
    assert gflat.numel() % multiple == 0
    pflat = torch.zeros_like(gflat)
    with torch.no_grad():
        for p, v in model._grad_views:
            o, n = v.storage_offset(), p.numel()
            dst = pflat[o:o + n].view_as(p)
            dst.copy_(p.data)
            p.data = dst
    for path, p in model._named_hot_params():    # pointers moved
        model._set_field(ps, path, p.data_ptr())
    model.__dict__['_param_flat'] = pflat
    return pflat


class ShardedAdamExchange(object):
    """Dense data-parallel step with the optimizer SHARDED over the ranks (ZeRO-1 form; new design, the reference is
    single-process: trainer.py:74-79, optimizers.py:241-243).

        reduce-scatter(sum) of the flat gradient  ->  rank r owns slice r of the reduced gradient
        ps_adam_sumsq on the slice  ->  ONE scalar all-reduce  =  the global clip norm over ALL gradients
        ps_adam_update_ext: clip + dense Adam on the slice only (moments exist only for the slice: 1/world of the state)
        all-gather of the updated parameter slices into every rank's flat parameter buffer

    Same arithmetic as the replicated optimizer (every element sees the identical reduced gradient, global norm, step
    count), so the reference's dense-Adam semantics hold exactly, but each rank streams 1/world of the optimizer's
    8 dwords per parameter (C2, N = 8: 42 us -> ~6 us of clip+Adam per step) and the two collectives move what ONE
    all-reduce moves (an all-reduce IS a reduce-scatter followed by an all-gather).  Replicas stay bitwise identical:
    every parameter value is computed once, by its owner.

    ``exchange()`` (between ``loss.backward()`` and ``optim.step()``) runs the reduce-scatter; ``optim.step()`` runs the
    rest (the exchange installs itself as ``optim._sharded``), so trainer.py's call order is unchanged."""

    def __init__(self, model, optim, group=None):
        self.model, self.optim, self.group = model, optim, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.lib = self._load_lib()
        W = self.world
        self.pflat = flatten_parameters(model, multiple=4 * W)
        n = self.pflat.numel()
        self.shard = n // W
        self.lo, self.hi = self.rank * self.shard, (self.rank + 1) * self.shard
        dev = self.pflat.device
        self.g_shard = torch.zeros(self.shard, device=dev, dtype=torch.float32)
        self.m_shard = torch.zeros(self.shard, device=dev, dtype=torch.float32)
        self.v_shard = torch.zeros(self.shard, device=dev, dtype=torch.float32)
        self.p_shard = self.pflat[self.lo:self.hi]
        self._build_plan(dev)
        self.state[0] = optim._step
        if getattr(optim, '_state_tensors', None):       # build_optim(train_from) loaded a checkpoint before the exchange existed
            self.load_moments(optim._state_tensors, optim._step)
        self.sumsq = torch.zeros(1, device=dev, dtype=torch.float32)
        self.gnorm = torch.zeros(2, device=dev, dtype=torch.float32)
        self._reduced = False
        # how the reduce-scatter is issued: 'rccl' = reduce_scatter_tensor (RCCL picks the algorithm; rings are bound by ONE
        # xGMI link); 'a2a' = equal-split all_to_all_single of the W slices + a local sum in rank order — on a fully connected
        # node every slice travels over its own direct link (7 links busy), one hop, and the sum order is fixed
        # PS_DP_RS / PS_DP_AG unset on the RCCL backend: 'auto' = both forms of each collective are TIMED on this node's links
        # inside the first exchange (a warm-up step) and the faster one is kept (`_tune`; every rank takes the max-over-ranks
        # times, so all ranks choose alike).  The all-gather has the same two forms ('a2a': every rank sends its slice to each
        # peer over that peer's direct link).
        auto = 'auto' if (W > 1 and dist.is_initialized() and dist.get_backend(group) == 'nccl') else 'rccl'
        if auto == 'auto' and self.lib.ps_set_deterministic(-1) != 0:
            auto = 'a2a'         # PS_DETERMINISTIC: no timing-dependent choice; the a2a form sums the slices in rank order
        self.rs_mode = os.environ.get('PS_DP_RS', auto)
        self.ag_mode = os.environ.get('PS_DP_AG', auto)
        self.tuned = None
        self._a2a_recv = None
        self._set_modes(self.rs_mode, self.ag_mode)
        self._tune_scratch = None
        if 'auto' in (self.rs_mode, self.ag_mode):       # _tune's operands, allocated HERE (an out-of-memory error surfaces at
            self._tune_scratch = (torch.zeros_like(self.pflat), torch.empty_like(self.g_shard), torch.empty_like(self.pflat))   # construction, on every rank alike, not between live collectives)
        optim.grad_scale = 1.0 / W
        optim._sharded = self
        optim._plan = None

    def _load_lib(self):
        from . import _lib
        return _lib.load()

    def _build_plan(self, dev):
        from . import _lib
        numel = torch.tensor([self.shard], dtype=torch.int64)
        addr = lambda t: torch.tensor([t.data_ptr()], dtype=torch.int64)
        pa, ga, ma, va = addr(self.p_shard), addr(self.g_shard), addr(self.m_shard), addr(self.v_shard)
        nbytes = self.lib.ps_adam_plan_bytes(1, numel.data_ptr())
        host = torch.zeros(nbytes, dtype=torch.uint8)
        _lib.check(self.lib.ps_adam_plan_write_host(1, pa.data_ptr(), ga.data_ptr(), ma.data_ptr(), va.data_ptr(),
                                                    numel.data_ptr(), host.data_ptr()), 'ps_adam_plan_write_host')
        self.n_chunks = self.lib.ps_adam_plan_chunks_host(host.data_ptr())
        self.plan = host.to(dev)
        self.state = torch.zeros(2 + (self.n_chunks + 1) // 2, device=dev, dtype=torch.int64)

    # the three device operations, overridable so that tests/test_dist_cpu.py can drive the collective protocol with a
    # CPU restatement of the kernels over gloo (the product path below has no CPU form: it calls the HIP library)
    def _k_zero(self, flat):
        from . import _lib
        st = torch.cuda.current_stream(flat.device).cuda_stream
        _lib.check(self.lib.ps_zero_floats(flat.data_ptr(), flat.numel(), st), 'ps_zero_floats')

    def _k_sumsq(self, hp):
        from . import _lib
        st = torch.cuda.current_stream(self.pflat.device).cuda_stream
        _lib.check(self.lib.ps_adam_sumsq(self.plan.data_ptr(), self.n_chunks, hp, self.state.data_ptr(),
                                          self.sumsq.data_ptr(), st), 'ps_adam_sumsq')

    def _k_update(self, hp):
        from . import _lib
        st = torch.cuda.current_stream(self.pflat.device).cuda_stream
        _lib.check(self.lib.ps_adam_update_ext(self.plan.data_ptr(), self.n_chunks, hp, self.state.data_ptr(),
                                               self.sumsq.data_ptr(), self.gnorm.data_ptr(), st), 'ps_adam_update_ext')

    def _k_sum_slices(self, recv, out, zero):
        """out = sum over ranks of recv[r] in rank order; clears ``zero`` (or nothing) in the same launch (ps_sum_slices)."""
        from . import _lib
        st = torch.cuda.current_stream(out.device).cuda_stream
        _lib.check(self.lib.ps_sum_slices(recv.data_ptr(), self.world, out.numel(), out.data_ptr(),
                                          zero.data_ptr() if zero is not None else None,
                                          zero.numel() if zero is not None else 0, st), 'ps_sum_slices')

    def _set_modes(self, rs, ag):
        W, dev = self.world, self.pflat.device
        self.rs_mode, self.ag_mode = rs, ag
        if W > 1 and rs in ('a2a', 'auto') and self._a2a_recv is None:
            self._a2a_recv = torch.empty(W, self.shard, device=dev, dtype=torch.float32)

    def _reduce_scatter(self, out, flat, mode, zero=None):
        """``out`` <- this rank's slice of the sum of ``flat`` over the ranks.  ``zero``: a buffer the caller wants cleared once
        the collective has consumed ``flat`` (the flat gradient itself on the step path): the a2a form clears it inside its
        summing launch and returns True, otherwise the caller clears it."""
        if mode == 'a2a':
            dist.all_to_all_single(self._a2a_recv.view(-1), flat, group=self.group)
            self._k_sum_slices(self._a2a_recv, out, zero)      # rank order: the same sum on every run
            return zero is not None
        dist.reduce_scatter_tensor(out, flat, op=dist.ReduceOp.SUM, group=self.group)
        return False

    def _all_gather(self, full, shard, mode):
        """``full`` [world * shard] <- every rank's ``shard``.  'a2a': one send of the slice to each peer and one receive from
        each, batched (ncclSend / ncclRecv pairs: on a fully connected xGMI node every pair has its own direct link) straight
        from the slice into its place — no staging copy (round 3 expanded the slice W times for all_to_all_single: a 27 MB
        torch copy per step at C2).  When ``shard`` IS the rank's slice of ``full`` (the step path) nothing is copied at all."""
        if mode == 'a2a' and shard.is_cuda and dist.get_backend(self.group) == 'gloo':
            # gloo rehearsals of the N-rank path on GPU tensors (tests/test_gpu_dp.py): its send / recv read device memory from
            # the host WITHOUT waiting for the stream (observed: replicas diverged at world 4), its collectives do wait — so
            # the slice goes through all_to_all_single from a W-fold staging copy there, as in round 3
            if getattr(self, '_a2a_send', None) is None:
                self._a2a_send = torch.empty(self.world, shard.numel(), device=shard.device, dtype=torch.float32)
            self._a2a_send.copy_(shard.unsqueeze(0).expand_as(self._a2a_send))
            dist.all_to_all_single(full, self._a2a_send.view(-1), group=self.group)
        elif mode == 'a2a':
            W, r, n = self.world, self.rank, shard.numel()
            mine = full[r * n:(r + 1) * n]
            if mine.data_ptr() != shard.data_ptr():
                mine.copy_(shard)
            ops = []
            # P2POp's peer is a GLOBAL rank; self.rank is the rank inside self.group (they differ on a strict sub-group)
            peer = (lambda g: g) if self.group is None else (lambda g: dist.get_global_rank(self.group, g))
            for k in range(1, W):                # ring-shifted order: in every round each rank sends to a different peer
                dst, src = (r + k) % W, (r - k) % W
                ops.append(dist.P2POp(dist.isend, shard, peer(dst), group=self.group))
                ops.append(dist.P2POp(dist.irecv, full[src * n:(src + 1) * n], peer(src), group=self.group))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        else:
            dist.all_gather_into_tensor(full, shard, group=self.group)

    def _tune(self, flat, iters=5):
        """Time 'rccl' and 'a2a' for each collective still on 'auto' (scratch operands of the step's own sizes) and keep the
        faster.  Runs once, inside the first exchange; GPU backends only (events)."""
        dev = flat.device
        src, out, full = self._tune_scratch
        ops = {'rs': lambda mode: self._reduce_scatter(out, src, mode), 'ag': lambda mode: self._all_gather(full, out, mode)}
        times = {}
        for name, cur in (('rs', self.rs_mode), ('ag', self.ag_mode)):
            if cur != 'auto':
                continue
            t = []
            for mode in ('rccl', 'a2a'):
                for _ in range(2):
                    ops[name](mode)
                torch.cuda.synchronize(dev)
                dist.barrier(group=self.group)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    ops[name](mode)
                e1.record()
                torch.cuda.synchronize(dev)
                t.append(e0.elapsed_time(e1) / iters)
            tt = torch.tensor(t, device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=self.group)
            times[name] = {'rccl_ms': float(tt[0]), 'a2a_ms': float(tt[1])}
        rs = self.rs_mode if self.rs_mode != 'auto' else ('a2a' if times['rs']['a2a_ms'] < times['rs']['rccl_ms'] else 'rccl')
        ag = self.ag_mode if self.ag_mode != 'auto' else ('a2a' if times['ag']['a2a_ms'] < times['ag']['rccl_ms'] else 'rccl')
        self._set_modes(rs, ag)
        if rs != 'a2a':
            self._a2a_recv = None
        self.tuned = dict(times, reduce_scatter=rs, all_gather=ag)
        self._tune_scratch = None
        if self.rank == 0:
            import sys
            print("ShardedAdamExchange: %s" % self.tuned, file=sys.stderr)

    def __call__(self):
        """reduce-scatter of the step's flat gradient; leaves the flat buffer zeroed for the next backward."""
        m = self.model
        flat = m._grad_flat
        if self.world > 1 and 'auto' in (self.rs_mode, self.ag_mode):
            # no try / except here: _tune issues live collectives, and a rank that caught a local failure between two of them
            # would enter the step's reduce-scatter while its peers still sit in the tuning all-to-all.  Its scratch operands
            # were allocated at construction; a collective that fails, fails the step on this rank (and the peers' timeout).
            self._tune(flat)
        zeroed = False
        if self.world > 1:
            with _comm('reduce_scatter(flat gradient)'):
                zeroed = self._reduce_scatter(self.g_shard, flat, self.rs_mode, zero=flat)
        else:
            self.g_shard.copy_(flat[self.lo:self.hi])
        if not zeroed:
            self._k_zero(flat)
        m.__dict__['_grad_clean'] = True
        self._reduced = True
        return None

    def step(self, hp):
        """The optimizer half (called by ``Optimizer.step``)."""
        if not self._reduced:
            raise RuntimeError("sharded optimizer: call the exchange between loss.backward() and optim.step()")
        self._reduced = False
        hp.zero_grads = 0
        self._k_sumsq(hp)
        if self.world > 1:
            with _comm('all_reduce(clip norm scalar)'):
                dist.all_reduce(self.sumsq, op=dist.ReduceOp.SUM, group=self.group)
        self._k_update(hp)
        if self.world > 1:
            with _comm('all_gather(parameter slices)'):
                self._all_gather(self.pflat, self.p_shard, self.ag_mode)

    def load_moments(self, state_tensors, step):
        """Adam moments of a checkpoint — ``{id(parameter): (exp_avg, exp_avg_sq)}`` as ``Optimizer.load_state_dict`` keeps them —
        laid out like the flat buffers; this rank keeps its slice.  Also sets the step the bias corrections continue from.
        Resuming is then the reference's Adam resume (optimizers.py:186-187 + ps_model.py:39-51) under data parallelism too."""
        mfull, vfull = torch.zeros_like(self.pflat), torch.zeros_like(self.pflat)
        with torch.no_grad():
            for p, view in self.model._grad_views:
                st = state_tensors.get(id(p))
                if st is None:
                    continue
                o, n = view.storage_offset(), p.numel()
                if tuple(st[0].shape) != tuple(p.shape) or tuple(st[1].shape) != tuple(p.shape):
                    raise RuntimeError("sharded optimizer: a loaded moment has shape %s, its parameter %s"
                                       % (tuple(st[0].shape), tuple(p.shape)))
                mfull[o:o + n].copy_(st[0].reshape(-1))
                vfull[o:o + n].copy_(st[1].reshape(-1))
            self.m_shard.copy_(mfull[self.lo:self.hi])
            self.v_shard.copy_(vfull[self.lo:self.hi])
            self.state[0] = int(step)

    def full_moments(self):
        """(exp_avg, exp_avg_sq) flat buffers gathered from every rank's shard (checkpointing: a collective)."""
        out = []
        for t in (self.m_shard, self.v_shard):
            full = torch.empty_like(self.pflat)
            if self.world > 1:
                dist.all_gather_into_tensor(full, t, group=self.group)
            else:
                full.copy_(t)
            out.append(full)
        return out


def make_exchange(model, optim=None, group=None, mode=None):
    """The exchange matching the model's optimizer mode.  Dense mode: ``sharded`` (reduce-scatter -> owner clip+Adam ->
    all-gather, the default when there is more than one rank and an optimizer to shard) or ``allreduce`` (one flat
    all-reduce, every rank runs the whole optimizer; PS_DP_EXCHANGE=allreduce or ``mode=``)."""
    if getattr(model, '_row_sparse', lambda: False)():
        return SparseGradExchange(model, optim, group)
    mode = mode or os.environ.get('PS_DP_EXCHANGE') or 'sharded'
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if mode == 'sharded' and optim is not None and world > 1 and getattr(optim, 'method', 'adam') == 'adam':
        return ShardedAdamExchange(model, optim, group)          # (the other --optim methods: flat all-reduce, replicated update)
    return GradExchange(lambda: model._grad_flat, optim, group)


def broadcast_parameters(model, src=0, group=None):
    """Replicas must start identical (tables and weights replicated).  ``src`` is a rank INSIDE ``group`` (torch's broadcast
    wants the global one: translated here, so a strict sub-group that does not hold global rank 0 works)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    if group is not None:
        src = dist.get_global_rank(group, src)
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def max_over_ranks(value, device):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])
