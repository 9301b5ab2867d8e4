"""Optimizer — the reference's controller class over the fused HIP clip + Adam.

Mirrors ``models/optimizers.py:111-243`` (``Optimizer(method, learning_rate,
max_grad_norm, beta1, beta2, decay_method, warmup_steps, weight_decay)``,
``set_parameters(named_params)``, ``step()``, ``learning_rate``, ``_step``) and
``build_optim`` (``models/ps_model.py:20-51``).  ``step()`` is two kernel launches
(global-norm partials, then clip + update over every tensor) through
``ps_clip_adam_dense``; dense semantics identical to ``torch.optim.Adam(eps=1e-9)``
after ``clip_grad_norm_`` — parameters without a gradient are skipped, as there.
``method`` 'adam' is the hot path (``main.py:62`` default); 'sgd', 'adagrad' and 'adadelta'
(``optimizers.py:175-183``: torch's rules with its defaults) run through the same two
launches (``PsAdamHyper.method``); the row-sparse / row-sharded / lazy-exact extensions are
Adam only.  'sparseadam' (``optimizers.py:188-193``, undocumented in ``main.py``) is not built.
"""
import os
import weakref

import torch

from . import _lib


class Optimizer(object):
    METHODS = {'adam': 0, 'sgd': 1, 'adagrad': 2, 'adadelta': 3}                      # PsAdamHyper.method
    # state tensor names per parameter in torch.optim's state_dict() (first -> the plan's `m`, second -> its `v`)
    STATE_KEYS = {'adam': ('exp_avg', 'exp_avg_sq'), 'sgd': (), 'adagrad': ('sum',), 'adadelta': ('square_avg', 'acc_delta')}

    def __init__(self, method, learning_rate, max_grad_norm,
                 lr_decay=1, start_decay_steps=None, decay_steps=None,
                 beta1=0.9, beta2=0.999, adagrad_accum=0.0,
                 decay_method=None, warmup_steps=4000, weight_decay=0., row_sparse=False):
        if method not in self.METHODS:
            if method == 'sparseadam':
                raise NotImplementedError("method 'sparseadam' (optimizers.py:188-193) is not built; args.row_sparse_adam is the "
                                          "touched-rows-only Adam of this package")
            raise RuntimeError("Invalid optim method: " + str(method))               # optimizers.py:194-195
        if method != 'adam' and row_sparse:
            raise NotImplementedError("row_sparse_adam / lazy_exact_adam / shard_tables need method='adam'")
        self.adagrad_accum = adagrad_accum
        self.last_ppl = None
        self.learning_rate = learning_rate
        self.original_lr = learning_rate
        self.max_grad_norm = max_grad_norm
        self.method = method
        self.lr_decay = lr_decay
        self.start_decay_steps = start_decay_steps
        self.decay_steps = decay_steps
        self.start_decay = False
        self._step = 0
        self.betas = [beta1, beta2]
        self.decay_method = decay_method
        self.warmup_steps = warmup_steps
        self.weight_decay = weight_decay
        self.eps = 1e-9                      # optimizers.py:186-187
        self.grad_scale = 1.0                # 1/world_size under data parallelism
        self.zero_grads_owner = None         # weakref to the model whose flat gradient buffer the dense step zeroes (build_optim)
        self.row_sparse = bool(row_sparse)   # extension: tables updated by touched rows only
        self.params = []
        self._plan = None
        self._sharded = None                 # dist.ShardedAdamExchange: this rank updates 1/world of the flat buffers
        self.shard_owner = None              # weakref to a model whose item table is row-sharded (args.shard_tables)
        self.lazy_exact = False              # args.lazy_exact_adam: row-sparse machinery, the DENSE optimizer's results (catch_up_rows)
        self._lazy_last = {}                 # id(table parameter) -> int32 [n_rows]: optimizer steps applied to each row
        self._lazy_base = 0                  # value a new entry of _lazy_last starts from (the step of a loaded checkpoint)
        self._lazy_awaiting_step = False     # a training forward has advanced its rows to the coming step (a second one is refused)

    def set_parameters(self, params):
        """``set_parameters`` (optimizers.py:165-187): every parameter that requires grad."""
        # (a sharded item table's receive buffer, ``_ps_shard_view``, is never a parameter of the optimizer: the owners
        # update their shards — sharded.ShardedItemTable — through ``_step_rows``)
        self.params = [p for _, p in params if p.requires_grad and not getattr(p, '_ps_shard_view', False)]
        self._names = [k for k, p in params if p.requires_grad and not getattr(p, '_ps_shard_view', False)]
        # checkpoint indices follow the REFERENCE's parameter order (optimizers.py:165-187 enumerates named_parameters): a
        # row-sharded table keeps its slot there although its owners, not this list, update it
        self._ref_params = [p for _, p in params if p.requires_grad]
        self._plan = None

    # ------------------------------------------------------------------ plan
    def _build_plan(self):
        lib = _lib.load()
        live = [p for p in self.params if p.grad is not None]
        if not live:
            raise RuntimeError("Optimizer.step(): no parameter has a gradient")
        dev = live[0].device
        if not live[0].is_cuda:
            raise RuntimeError("Optimizer.step() needs parameters on a gfx950 device (no CPU fallback)")
        all_live = live
        rows_live = [p for p in live if self.row_sparse and getattr(p, '_ps_rows', None) is not None]
        live = [p for p in live if not any(p is q for q in rows_live)]
        if not live:
            raise RuntimeError("Optimizer.step(): row-sparse mode needs at least one dense tensor")
        n = len(live)
        numel = torch.tensor([p.numel() for p in live], dtype=torch.int64)
        sizes = [(p.numel() + 3) // 4 * 4 for p in live]
        total = sum(sizes)
        old = getattr(self, '_state_tensors', None)
        # (sgd keeps no state and its kernels touch neither buffer; adagrad's sum starts at adagrad_accum, optimizers.py:178-181)
        self._m_flat = torch.full((total,), float(self.adagrad_accum) if self.method == 'adagrad' else 0.0, device=dev,
                                  dtype=torch.float32) if self.method != 'sgd' else torch.zeros(total, device=dev)
        self._v_flat = torch.zeros(total, device=dev, dtype=torch.float32)
        ms, vs, o = [], [], 0
        for p, s in zip(live, sizes):
            ms.append(self._m_flat[o:o + p.numel()].view_as(p))
            vs.append(self._v_flat[o:o + p.numel()].view_as(p))
            o += s
        for p in rows_live:          # full-size moments; only touched rows are ever read or written
            live.append(p)
            if old is not None and id(p) in old:
                ms.append(old[id(p)][0]); vs.append(old[id(p)][1])
            else:
                ms.append(torch.zeros_like(p)); vs.append(torch.zeros_like(p))
        n_dense = n
        if old is not None:          # keep moments of parameters that were already being updated
            for p, m, v in zip(live[:n_dense], ms, vs):
                if id(p) in old:
                    m.copy_(old[id(p)][0]); v.copy_(old[id(p)][1])
        self._state_tensors = {id(p): (m, v) for p, m, v in zip(live, ms, vs)}
        addr = lambda ts: torch.tensor([t.data_ptr() for t in ts], dtype=torch.int64)
        dl = live[:n_dense]
        pa, ga, ma, va = addr(dl), addr([p.grad for p in dl]), addr(ms[:n_dense]), addr(vs[:n_dense])
        nbytes = lib.ps_adam_plan_bytes(n, numel.data_ptr())
        host = torch.zeros(nbytes, dtype=torch.uint8)
        _lib.check(lib.ps_adam_plan_write_host(n, pa.data_ptr(), ga.data_ptr(), ma.data_ptr(), va.data_ptr(),
                                               numel.data_ptr(), host.data_ptr()), 'ps_adam_plan_write_host')
        n_chunks = lib.ps_adam_plan_chunks_host(host.data_ptr())
        plan = dict(dev=host.to(dev), n_chunks=n_chunks, live=all_live,
                    rows=[(p, self._state_tensors[id(p)]) for p in rows_live],
                    sig=tuple((p.data_ptr(), p.grad.data_ptr()) for p in all_live),
                    state=torch.zeros(2 + (n_chunks + 1) // 2, device=dev, dtype=torch.int64),
                    gnorm=torch.zeros(2, device=dev, dtype=torch.float32))
        plan['state'][0] = self._step
        self._plan = plan
        return plan

    def _plan_ok(self):
        pl = self._plan
        if pl is None:
            return False
        # cheap identity check first: the same Parameter objects carrying the same .grad tensor OBJECTS as last step
        # (the model re-attaches its persistent flat-buffer views after every zero_grad)
        fast = pl.get('fast')
        if fast is not None and all(p.grad is g for p, g in fast) and \
                all(p.grad is None for p in pl['dead']):
            return True
        live = [p for p in self.params if p.grad is not None]
        if len(live) != len(pl['live']):
            return False
        ok = all(a is b and (a.data_ptr(), a.grad.data_ptr()) == s for a, b, s in zip(live, pl['live'], pl['sig']))
        if ok:
            pl['fast'] = [(p, p.grad) for p in live]
            pl['dead'] = [p for p in self.params if p.grad is None]
        return ok

    def _hyper(self):
        hp = _lib.PsAdamHyper()
        hp.lr = self.original_lr if self.decay_method == "noam" else self.learning_rate
        hp.beta1, hp.beta2, hp.eps = self.betas[0], self.betas[1], self.eps
        hp.weight_decay = self.weight_decay
        hp.max_grad_norm = self.max_grad_norm if self.max_grad_norm else 0.0
        hp.noam = int(self.decay_method == "noam")
        hp.warmup_steps = self.warmup_steps
        hp.grad_scale = self.grad_scale
        hp.method = self.METHODS[self.method]
        return hp

    # ------------------------------------------------------------------ lazy-exact dense Adam (args.lazy_exact_adam)
    def _catch_up(self, advance, all_rows, use_active):
        """``ps_rowsparse_catchup`` over the row-sparse tables of the current plan: replay, with a zero gradient, the optimizer
        steps a row missed (the dense optimizer moves every row every step, optimizers.py:241-243).  Before the first step there
        is no state and nothing to replay."""
        plan = self._plan
        if not self.lazy_exact or plan is None or not plan['rows']:
            return
        import ctypes as C
        lib = _lib.load()
        n = len(plan['rows'])
        tabs = (_lib.PsRowTable * n)()
        last = (C.c_void_p * n)()
        nrows = (C.c_int64 * n)()
        for i, (p, (m, v)) in enumerate(plan['rows']):
            info = p._ps_rows
            t = tabs[i]
            t.p, t.m, t.v = p.data_ptr(), m.data_ptr(), v.data_ptr()
            t.g = p.grad.data_ptr() if p.grad is not None else 0
            rows, count, cap = ((info.get('active') if use_active else None) or (info['rows'], info['count'], info['cap']))
            t.rows, t.count, t.cap, t.d = rows.data_ptr(), count.data_ptr(), cap, p.shape[1]
            lt = self._lazy_last.get(id(p))
            if lt is None or lt.numel() != p.shape[0] or lt.device != p.device:
                # fresh moments are zero: replaying from step 0 changes nothing and ends after one iteration; moments loaded
                # from a checkpoint are current as of its step (load_state_dict sets _lazy_base)
                lt = self._lazy_last[id(p)] = torch.full((p.shape[0],), int(self._lazy_base), device=p.device, dtype=torch.int32)
            last[i], nrows[i] = lt.data_ptr(), p.shape[0]
        st = torch.cuda.current_stream(plan['live'][0].device).cuda_stream
        _lib.check(lib.ps_rowsparse_catchup(tabs, n, last, nrows, self._hyper(), plan['state'].data_ptr(), int(advance),
                                            int(all_rows), st), 'ps_rowsparse_catchup')

    def catch_up_rows(self):
        """Before a forward READS the step's rows (the model calls this after coalescing them)."""
        self._catch_up(True, False, False)

    def flush_rows(self):
        """Bring EVERY row up to the current step (before evaluation, ``state_dict()``, or leaving the mode)."""
        self._catch_up(False, True, False)

    # ------------------------------------------------------------------ step
    def step(self):
        """``Optimizer.step`` (optimizers.py:205-243)."""
        lib = _lib.load()
        plan = None if self._sharded is not None else (self._plan if self._plan_ok() else self._build_plan())
        self._step += 1
        self._lazy_awaiting_step = False
        if self.decay_method == "noam":      # host mirror of the in-kernel schedule (optimizers.py:214-219)
            self.learning_rate = self.original_lr * min(self._step ** (-0.5),
                                                        self._step * self.warmup_steps ** (-1.5))
        hp = self._hyper()
        if self._sharded is not None:
            return self._sharded.step(hp)
        dev = plan['live'][0].device
        st = torch.cuda.current_stream(dev).cuda_stream
        if plan['rows'] or self.shard_owner is not None:
            return self._step_rows(lib, plan, hp, dev, st)
        owner = self.zero_grads_owner() if self.zero_grads_owner is not None else None
        # only when the plan covers the model's whole flat gradient buffer (every view is the attached .grad)
        zero = owner is not None and all(p.grad is v for p, v in owner._grad_views)
        hp.zero_grads = int(zero)
        _lib.check(lib.ps_clip_adam_dense(plan['dev'].data_ptr(), plan['n_chunks'], hp, plan['state'].data_ptr(),
                                          plan['gnorm'].data_ptr(), st), 'ps_clip_adam_dense')
        if zero:
            owner.__dict__['_grad_clean'] = True

    def _step_rows(self, lib, plan, hp, dev, st):
        """Row-sparse step: dense plan for the small tensors + touched rows of every table, one global
        norm, three launches (``ps_clip_adam_rowsparse``); touched gradient rows come back zeroed."""
        tabs = (_lib.PsRowTable * len(plan['rows']))()
        for i, (p, (m, v)) in enumerate(plan['rows']):
            info = p._ps_rows
            t = tabs[i]
            t.p, t.g, t.m, t.v = p.data_ptr(), p.grad.data_ptr(), m.data_ptr(), v.data_ptr()
            rows, count, cap = info.get('active') or (info['rows'], info['count'], info['cap'])
            t.rows, t.count, t.cap, t.d = rows.data_ptr(), count.data_ptr(), cap, p.shape[1]
        owner = self.shard_owner() if self.shard_owner is not None else None
        shard = getattr(owner, '_shard', None) if owner is not None else None
        if shard is not None:
            # row-sharded item table: route the step's gradient rows to their owners, then the owned shard joins the plan as
            # one more table whose sum of squares is added over the ranks (every row is owned once)
            if shard.pending:
                shard.push_grads(owner.product_emb.weight.grad)
            n_shared = len(plan['rows'])
            tabs2 = (_lib.PsRowTable * (n_shared + 1))()
            for i in range(n_shared):
                for f, _t in _lib.PsRowTable._fields_:
                    setattr(tabs2[i], f, getattr(tabs[i], f))
            own = shard.row_table()
            for f, _t in _lib.PsRowTable._fields_:
                setattr(tabs2[n_shared], f, getattr(own, f))
            tabs = tabs2
        need = lib.ps_adam_rowsparse_state_floats(plan['n_chunks'], tabs, len(tabs))
        if need < 0:
            _lib.check(1, 'ps_adam_rowsparse_state_floats')
        words = 2 + (need + 1) // 2
        if plan['state'].numel() < words:
            plan['state'] = torch.zeros(words, device=dev, dtype=torch.int64)
            plan['state'][0] = self._step - 1
        if shard is not None:
            import torch.distributed as tdist
            sums = plan.get('sums')
            if sums is None:
                sums = plan['sums'] = torch.zeros(2, device=dev, dtype=torch.float32)
            _lib.check(lib.ps_rowsparse_sumsq(plan['dev'].data_ptr(), plan['n_chunks'], tabs, len(tabs), len(tabs) - 1, hp,
                                              plan['state'].data_ptr(), sums.data_ptr(), st), 'ps_rowsparse_sumsq')
            if shard.world > 1:
                tdist.all_reduce(sums[1:2], op=tdist.ReduceOp.SUM, group=shard.group)
            _lib.check(lib.ps_rowsparse_update_ext(plan['dev'].data_ptr(), plan['n_chunks'], tabs, len(tabs), hp,
                                                   plan['state'].data_ptr(), sums.data_ptr(), plan['gnorm'].data_ptr(), st),
                       'ps_rowsparse_update_ext')
        else:
            if self.lazy_exact:          # over the FINAL touched lists (a data-parallel exchange may have widened them)
                self._catch_up(True, False, True)
            _lib.check(lib.ps_clip_adam_rowsparse(plan['dev'].data_ptr(), plan['n_chunks'], tabs, len(tabs), hp,
                                                  plan['state'].data_ptr(), plan['gnorm'].data_ptr(), st),
                       'ps_clip_adam_rowsparse')
        for p, _ in plan['rows']:
            p._ps_rows['dirty'] = False

    @property
    def last_grad_norm(self):
        """Pre-clip global gradient norm of the last step (host sync when read)."""
        if self._sharded is not None:
            return float(self._sharded.gnorm[0])
        return float(self._plan['gnorm'][0]) if self._plan else None

    # ------------------------------------------------------------ checkpointing
    def _shard(self):
        owner = self.shard_owner() if self.shard_owner is not None else None
        return getattr(owner, '_shard', None) if owner is not None else None

    def state_dict(self):
        """Adam state in ``torch.optim.Adam.state_dict()`` form (exp_avg / exp_avg_sq / step per parameter index, indices
        in the reference's ``named_parameters`` order) so a reference checkpoint's ``optim.optimizer.state_dict()`` maps
        1:1.  A row-sharded item table (``args.shard_tables``) exports its moments gathered from the owners in the full
        table's shape, like ``model.state_dict()`` does for the weight; under the sharded optimizer the moment slices are
        gathered from the ranks.  Both are collectives: every rank calls this."""
        state = {}
        self.flush_rows()                    # (lazy_exact_adam: every row's moments current as of this step)
        st = getattr(self, '_state_tensors', {})
        if self._sharded is not None:        # moments live sharded over the ranks: gather them (a collective)
            mf, vf = self._sharded.full_moments()
            st = {}
            for p, v in self._sharded.model._grad_views:
                o, n = v.storage_offset(), p.numel()
                st[id(p)] = (mf[o:o + n].view_as(p), vf[o:o + n].view_as(p))
        shard = self._shard()
        ref = getattr(self, '_ref_params', self.params)
        for i, p in enumerate(ref):
            if getattr(p, '_ps_shard_view', False):
                if shard is None or self._step == 0:
                    continue
                pad = torch.zeros(1, shard.d, device=shard.device)      # the reference's table has its padding row last
                state[i] = {'step': torch.tensor(float(self._step)),
                            'exp_avg': torch.cat([shard.gather_full('m'), pad]),
                            'exp_avg_sq': torch.cat([shard.gather_full('v'), pad])}
            elif id(p) in st and self.method != 'sgd':
                keys = self.STATE_KEYS[self.method]
                state[i] = {'step': torch.tensor(float(self._step))}
                for key, t in zip(keys, st[id(p)]):
                    state[i][key] = t.clone()
        group = {'lr': self.learning_rate, 'weight_decay': self.weight_decay, 'params': list(range(len(ref)))}
        if self.method == 'adam':
            group.update(betas=tuple(self.betas), eps=self.eps)
        return {'state': state, 'param_groups': [group], '_step': self._step}

    def load_state_dict(self, sd):
        """Restore ``state_dict()``'s (or the reference optimizer's) Adam state.  Indices are positions in the reference's
        parameter order; every moment must have its parameter's shape (a dense checkpoint loaded into a run with another
        table layout raises instead of assigning moments to the wrong tensors)."""
        self._step = int(sd.get('_step', 0))
        loaded = {}
        shard = self._shard()
        ref = getattr(self, '_ref_params', self.params)
        for i, s in sd['state'].items():
            p = ref[int(i)]
            self._step = max(self._step, int(float(s['step'])))
            if getattr(p, '_ps_shard_view', False):
                if shard is None:
                    raise RuntimeError("optimizer state for a row-sharded table but the model has no shard")
                want = (shard.n_rows, shard.d)
                for which, key in (('m', 'exp_avg'), ('v', 'exp_avg_sq')):
                    t = s[key]
                    if tuple(t.shape) not in (want, (want[0] + 1, want[1])):
                        raise RuntimeError("optimizer state %d (%s): shape %s, the sharded table is %s (+ padding row)"
                                           % (int(i), key, tuple(t.shape), want))
                    shard.load_full(t, which)
                continue
            keys = self.STATE_KEYS[self.method]
            for key in keys:
                if key not in s:
                    raise RuntimeError("optimizer state %d has no '%s': the checkpoint was written by another --optim method"
                                       % (int(i), key))
                if tuple(s[key].shape) != tuple(p.shape):
                    raise RuntimeError("optimizer state %d (%s): shape %s does not match its parameter's %s — the checkpoint "
                                       "was written for a different parameter list" % (int(i), key, tuple(s[key].shape), tuple(p.shape)))
            if keys:                             # (the plan's second buffer is unused by adagrad: zeros)
                first = s[keys[0]].to(p.device)
                loaded[id(p)] = (first, s[keys[1]].to(p.device) if len(keys) > 1 else torch.zeros_like(first))
        self._state_tensors = loaded
        self._plan = None
        self._lazy_last, self._lazy_base = {}, self._step      # lazy_exact_adam: a checkpoint is written flushed (state_dict)
        if self._sharded is not None:        # sharded optimizer: the moments live in its slices, not in _state_tensors
            self._sharded.load_moments(loaded, self._step)


def build_optim(args, model, checkpoint):
    """``build_optim`` (models/ps_model.py:20-51)."""
    optim = Optimizer(args.optim, args.lr, args.max_grad_norm,
                      beta1=args.beta1, beta2=args.beta2, adagrad_accum=getattr(args, 'adagrad_accum', 0.0),
                      decay_method=args.decay_method,
                      warmup_steps=args.warmup_steps,
                      weight_decay=args.l2_lambda,
                      row_sparse=getattr(args, 'row_sparse_adam', False) or getattr(args, 'lazy_exact_adam', False))
    optim.set_parameters(list(model.named_parameters()))
    if getattr(args, 'lazy_exact_adam', False):
        # the row-sparse machinery producing the DENSE optimizer's parameters (optimizers.py:241-243) — see catch_up_rows
        if getattr(model, '_shard', None) is not None:
            raise NotImplementedError("lazy_exact_adam with shard_tables")
        optim.lazy_exact = True
        model.__dict__['_lazy_optim'] = weakref.ref(optim)
    # The dense step leaves every gradient it consumed at zero (PsAdamHyper.zero_grads), so the memset of the next
    # zero_grad() / backward (trainer.py:76-77) costs nothing.  Visible difference: ``p.grad`` reads 0 after ``optim.step()``
    # where the reference still holds the step's gradient (nothing in it reads that).  ``args.keep_grads_after_step`` /
    # ``PS_KEEP_GRADS=1`` restore the memset.
    if hasattr(model, '_grad_flat') and not getattr(args, 'keep_grads_after_step', False) \
            and os.environ.get('PS_KEEP_GRADS', '0') in ('', '0'):
        optim.zero_grads_owner = weakref.ref(model)
    if getattr(model, '_shard', None) is not None:
        if args.optim != 'adam':      # (the constructor's own check only sees its row_sparse ARGUMENT; this path sets the flag afterwards)
            raise NotImplementedError("shard_tables: the owner-side row-sparse optimizer is Adam only (--optim %s)" % args.optim)
        optim.row_sparse = True
        optim.shard_owner = weakref.ref(model)
    if getattr(args, 'train_from', '') != '' and checkpoint is not None:
        optim.load_state_dict(checkpoint['optim'])
    return optim
