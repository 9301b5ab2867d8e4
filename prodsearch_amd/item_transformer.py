"""ItemTransformerRanker — drop-in boundary of the TEM / QEM ranking-loss step.

Mirrors the reference's ``nn.Module`` contract (``models/item_transformer.py:22-100,
352-359``; call sites ``main.py:148-149``, ``trainer.py:74-79,190,201``):

    model = ItemTransformerRanker(args, device, vocab_size, product_size, vocab_words, word_dists)
    loss = model(batch, train_pv)      # 0-dim fp32 tensor with a grad_fn
    model.zero_grad(); loss.backward(); optim.step(); loss.item()
    scores = model.test(batch)         # [B, candi_k]

Same parameter names / shapes / ``state_dict`` keys, so reference checkpoints load
(``load_cp``).  The torch modules below are PARAMETER HOLDERS only: no torch math
runs on the hot path.  ``forward`` / ``backward`` / ``test`` are one C-ABI call
each into ``libprodsearch_hip.so`` (hand-written gfx950 kernels); PyTorch owns
the memory and lends raw pointers for the duration of the call.  There is no
CPU fallback — on a CPU tensor the model raises.

Supported: ``model_name`` in {item_transformer (with use_dot_prod), QEM}.  The other
scoring heads of the reference (forward_trans, ZAM/AEM, forward_seq) are outside
the hot path and raise NotImplementedError.
"""
import math

import torch
import torch.nn as nn

from . import _lib


_DET_SET_BY_ARGS = False     # a model's args.deterministic turned the process-wide mode on


# ------------------------------------------------------------ parameter holders
class _Holder(nn.Module):
    def forward(self, *a, **k):     # pragma: no cover - never on the hot path
        raise RuntimeError("parameter holder: the hot path runs in libprodsearch_hip.so")


class _PositionalEncoding(_Holder):
    """``PositionalEncoding`` buffer (transformer.py:10-19): pe [1, 5000, d]."""
    def __init__(self, dim, max_len=5000):
        super().__init__()
        pe = torch.zeros(max_len, dim)
        position = torch.arange(0, max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, dim, 2, dtype=torch.float) * -(math.log(10000.0) / dim))
        pe[:, 0::2] = torch.sin(position.float() * div_term)
        pe[:, 1::2] = torch.cos(position.float() * div_term)
        self.register_buffer('pe', pe.unsqueeze(0))


class _MultiHeadedAttention(_Holder):
    """neural.py:86-96 parameter names."""
    def __init__(self, d):
        super().__init__()
        self.linear_keys = nn.Linear(d, d)
        self.linear_values = nn.Linear(d, d)
        self.linear_query = nn.Linear(d, d)
        self.final_linear = nn.Linear(d, d)


class _PositionwiseFeedForward(_Holder):
    """neural.py:20-26 parameter names."""
    def __init__(self, d, ff):
        super().__init__()
        self.w_1 = nn.Linear(d, ff)
        self.w_2 = nn.Linear(ff, d)
        self.layer_norm = nn.LayerNorm(d, eps=1e-6)


class _TransformerEncoderLayer(_Holder):
    """transformer.py:37-45."""
    def __init__(self, d, ff):
        super().__init__()
        self.self_attn = _MultiHeadedAttention(d)
        self.feed_forward = _PositionwiseFeedForward(d, ff)
        self.layer_norm = nn.LayerNorm(d, eps=1e-6)


class _TransformerEncoder(_Holder):
    """transformer.py:59-69."""
    def __init__(self, d, ff, n_layers):
        super().__init__()
        self.pos_emb = _PositionalEncoding(d)
        self.transformer_inter = nn.ModuleList([_TransformerEncoderLayer(d, ff) for _ in range(n_layers)])
        self.layer_norm = nn.LayerNorm(d, eps=1e-6)
        self.wo = nn.Linear(d, 1, bias=True)


class _FSEncoder(_Holder):
    """text_encoder.py:19-26."""
    def __init__(self, d):
        super().__init__()
        self.f_W = nn.Linear(d, d)


def _init_like_reference(module):
    """``initialize_parameters`` (transformer.py:100-118, text_encoder.py:42-60): >=2-D
    'weight' Xavier-normal, 'bias' 0, everything else (1-D LayerNorm gains!) N(0,1)."""
    for name, p in module.named_parameters():
        if 'weight' in name and p.dim() > 1:
            nn.init.xavier_normal_(p)
        elif 'bias' in name:
            nn.init.constant_(p, 0)
        else:
            nn.init.normal_(p)


# -------------------------------------------------------------------- autograd
def _check_same_forward(model, step):
    """Each batch shape owns ONE workspace (activations, sampled negatives, batch pointers) that the next forward
    overwrites, so only the most recent forward can be differentiated — unlike autograd, which would keep both graphs
    alive.  Anything else must fail loudly rather than return the other forward's gradients."""
    if model._fwd_step != step:
        raise RuntimeError("backward() of a loss whose forward is no longer the model's latest one (forward #%d, now #%d): "
                           "the HIP workspace holds one forward at a time; call backward() before the next forward()"
                           % (step, model._fwd_step))


class _RankLossFn(torch.autograd.Function):
    """One node: forward launched the HIP forward; backward launches the HIP backward,
    which writes the dense ``.grad`` of every reachable parameter directly."""

    @staticmethod
    def forward(ctx, anchor, model, plan, loss3):
        ctx.model, ctx.plan, ctx.step = model, plan, model._fwd_step
        return loss3[0]

    @staticmethod
    def backward(ctx, grad_out):
        _check_same_forward(ctx.model, ctx.step)
        ctx.model._run_backward(ctx.plan, grad_out)
        return None, None, None, None


class _LossTensor(torch.Tensor):
    """The 0-dim loss ``forward`` returns.  It is an ordinary autograd tensor (``grad_fn`` = the node above), but the
    trainer's plain ``loss.backward()`` (trainer.py:77) — no ``gradient``, no ``inputs``, no ``create_graph`` — is
    d loss/d loss = 1 through a single node, so it calls the HIP backward directly: no autograd-engine round trip on
    the host and no ``ones_like`` fill kernel on the device.  Anything else (scaled losses, retained graphs) takes the
    normal autograd path."""

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        fast = self.__dict__.pop('_ps_fast', None)
        if fast is not None and gradient is None and not create_graph and inputs is None and not retain_graph:
            model, plan, step = fast
            _check_same_forward(model, step)         # the workspace must still hold this forward's activations
            model._run_backward(plan, None)
            return None
        return super().backward(gradient, retain_graph, create_graph, inputs)


class _Plan(object):
    """Per-shape cached call state: descriptor, batch struct, workspace."""
    __slots__ = ('desc', 'batch', 'ws', 'layout', 'key', 'neg_items', 'neg_words', 'keep', 'staged', 'coalesced_at')


# ------------------------------------------------------------------------ model
class ItemTransformerRanker(nn.Module):
    def __init__(self, args, device, vocab_size, product_size, vocab_words, word_dists=None):
        super(ItemTransformerRanker, self).__init__()
        # args.deterministic: bitwise reproducible steps.  The switch is PROCESS-WIDE (ps_set_deterministic); a model built
        # WITHOUT the flag after one that set it through its args turns the mode off again, so it does not silently inherit
        # the ~2.6x slower single-stream step.  Explicit ps_set_deterministic() calls / PS_DETERMINISTIC=1 are left alone.
        global _DET_SET_BY_ARGS
        if getattr(args, 'deterministic', False):
            _lib.load().ps_set_deterministic(1)
            _DET_SET_BY_ARGS = True
        elif _DET_SET_BY_ARGS:
            _lib.load().ps_set_deterministic(0)
            _DET_SET_BY_ARGS = False
        if args.model_name not in ('item_transformer', 'QEM'):
            raise NotImplementedError("model_name %r is outside the hot path (item_transformer/QEM only)"
                                      % args.model_name)
        if args.model_name == 'item_transformer' and not args.use_dot_prod:
            raise NotImplementedError("forward_trans (use_dot_prod=False) is outside the hot path")
        if getattr(args, 'pretrain_emb_dir', '') or getattr(args, 'pretrain_up_emb_dir', ''):
            raise NotImplementedError("pretrained-embedding text loaders are out of scope; load a state_dict")
        self.args = args
        self.device = device
        self.train_review_only = args.train_review_only
        self.embedding_size = args.embedding_size
        self.vocab_words = vocab_words
        self.vocab_size = vocab_size
        self.product_size = product_size
        self.word_dists = None
        if word_dists is not None:
            self.word_dists = torch.as_tensor(word_dists, dtype=torch.float64)
        self.prod_pad_idx = product_size
        self.word_pad_idx = vocab_size - 1
        self.seg_pad_idx = 3
        self.emb_dropout = args.dropout
        d = self.embedding_size

        # args.shard_tables (extension; prodsearch_amd/sharded.py): the item table is sharded by row over
        # the ranks; ``product_emb.weight`` is then the per-step RECEIVE buffer [slots + 1, d] the kernels read, the batch's
        # item indices are remapped into its slots, and the owners update their shards with the row-sparse optimizer.
        self._shard = None
        if getattr(args, 'shard_tables', False):
            if args.sep_prod_emb or args.sim_func == 'bias_product':
                raise NotImplementedError("shard_tables: sep_prod_emb / bias_product are not supported with a sharded item table")
            from .sharded import ShardedItemTable
            tem_l = int(getattr(args, 'uprev_review_limit', 20)) if args.model_name == 'item_transformer' else 0
            cap = int(args.batch_size) * (1 + int(args.neg_per_pos) + tem_l)
            self._shard = ShardedItemTable(product_size, d, product_size, cap, device)
        # same registration order as the reference => same state_dict key order
        # tables above 0.5 GB (config 5: 51 GB) are created on the device, never staged through host memory
        emb_dev = device if (product_size + 1) * d > (1 << 27) else None
        if self._shard is not None:
            self.product_emb = nn.Embedding(self._shard.slots + 1, d, padding_idx=self._shard.slots, device=device)
            self.product_emb.weight._ps_shard_view = True       # never optimised: its gradient is routed to the owners
        else:
            self.product_emb = nn.Embedding(product_size + 1, d, padding_idx=self.prod_pad_idx, device=emb_dev)
        if args.sep_prod_emb:
            self.hist_product_emb = nn.Embedding(product_size + 1, d, padding_idx=self.prod_pad_idx, device=emb_dev)
        self.product_bias = nn.Parameter(torch.zeros(product_size + 1), requires_grad=True)
        self.word_bias = nn.Parameter(torch.zeros(vocab_size), requires_grad=True)
        self.word_embeddings = nn.Embedding(vocab_size, d, padding_idx=self.word_pad_idx)
        if args.model_name == 'item_transformer':
            self.transformer_encoder = _TransformerEncoder(d, args.ff_size, args.inter_layers)
        else:
            self.attention_encoder = _MultiHeadedAttention(d)
        if args.query_encoder_name == 'fs':
            self.query_encoder = _FSEncoder(d)
        else:
            self.query_encoder = _Holder()
        self.seg_embeddings = nn.Embedding(4, d, padding_idx=self.seg_pad_idx)
        self.initialize_parameters()
        self.to(device)
        if self._shard is not None:
            self._shard.attach(self.product_emb.weight)
            self._shard.init_normal(int(getattr(args, 'seed', 666)))

        self._plans = {}
        self._params_struct = None
        self._grads_struct = None
        self._grad_flat = None
        self._grad_views = None
        self._loss_acc = None
        self._alias = None
        self._fwd_step = 0
        self._seed = int(getattr(args, 'seed', 666))

    # ---------------------------------------------------------------- reference API
    def initialize_parameters(self, logger=None):
        """item_transformer.py:576-586."""
        nn.init.normal_(self.word_embeddings.weight)
        nn.init.normal_(self.seg_embeddings.weight)
        if self.args.query_encoder_name == 'fs':
            _init_like_reference(self.query_encoder)
        if self.args.model_name == 'item_transformer':
            _init_like_reference(self.transformer_encoder)

    def clear_loss(self):
        if self._loss_acc is not None:
            self._loss_acc.zero_()

    @property
    def ps_loss(self):
        """Accumulated on the device; reading it is the only host sync (the reference
        syncs twice per step with .item(), item_transformer.py:516-517)."""
        return 0.0 if self._loss_acc is None else float(self._loss_acc[0])

    @property
    def item_loss(self):
        return 0.0 if self._loss_acc is None else float(self._loss_acc[1])

    def load_cp(self, pt, strict=True):
        self.load_state_dict(pt['model'], strict=strict)

    def state_dict(self, *a, **k):
        """With a sharded item table ``product_emb.weight`` is exported as the FULL [P+1, d] table, gathered from the
        ranks' shards (a collective; for catalogue-sized tables export ``model._shard.weight`` per rank instead)."""
        self._lazy_flush()
        sd = super().state_dict(*a, **k)
        if self.__dict__.get('_shard') is not None:
            key = [q for q in sd if q.endswith('product_emb.weight')][0]
            full = self._shard.gather_full()
            sd[key] = torch.cat([full, full.new_zeros(1, full.shape[1])], 0)
        return sd

    def load_state_dict(self, sd, strict=True, **k):
        if self.__dict__.get('_shard') is not None and 'product_emb.weight' in sd:
            sd = dict(sd)
            self._shard.load_full(sd.pop('product_emb.weight'))
            strict = False
        return super().load_state_dict(sd, strict=strict, **k)

    def forward(self, batch_data, train_pv=False, neg_item_idxs=None, neg_word_idxs=None):
        plan, loss3 = self._run_forward(batch_data, neg_item_idxs, neg_word_idxs)
        if not torch.is_grad_enabled():
            return loss3[0]
        out = _RankLossFn.apply(self._anchor(), self, plan, loss3).as_subclass(_LossTensor)
        out._ps_fast = (self, plan, self._fwd_step)
        return out

    def test(self, batch_data):
        if self.__dict__.get('_shard') is not None:
            return self._run_score_sharded(batch_data)
        return self._run_score(batch_data)

    def _run_score_sharded(self, batch):
        """``test()`` over a row-sharded item table: the candidates' (and the history's) rows come from their owners first, like
        the training forward's lookup (a COLLECTIVE: every rank calls test() the same number of times with batches of the same
        shape, each with its own candidates), then the ordinary score launch reads the receive buffer through remapped ids.
        The buffer holds ``shard.cap`` indices (a function of the training batch shape), so a wide candidate list goes through in
        column chunks; scores are per (row, candidate), so the chunks concatenate (item_transformer.py:87-131)."""
        import copy
        sh = self._shard
        tem = self.args.model_name == 'item_transformer'
        if getattr(batch, 'candi_prod_idxs', None) is None:
            raise RuntimeError("test(): batch.candi_prod_idxs [B,C] is required")
        cand = self._check_idx(batch.candi_prod_idxs, 'candi_prod_idxs')
        hist = self._check_idx(batch.u_item_idxs, 'u_item_idxs') if tem else None
        Bn, C = int(cand.shape[0]), int(cand.shape[1])
        room = sh.cap - (hist.numel() if tem else 0)
        per = min(C, room // max(Bn, 1))
        if per < 1:
            raise RuntimeError("test(): %d rows x (history + 1 candidate) do not fit the shard exchange's %d indices per step "
                               "(sized from args.batch_size x (1 + neg_per_pos + history)); use a smaller test batch" % (Bn, sh.cap))
        outs = []
        for c0 in range(0, C, per):
            cc = cand[:, c0:c0 + per].contiguous()
            rem = sh.lookup(([hist] if tem else []) + [cc])
            b2 = copy.copy(batch)
            if tem:
                b2.u_item_idxs = rem[0]
            b2.candi_prod_idxs = rem[-1].view(Bn, -1)
            if getattr(b2, 'target_prod_idxs', None) is not None:      # not read by the score launch; keep it inside the buffer
                b2.target_prod_idxs = torch.full_like(b2.target_prod_idxs, sh.slots)
            outs.append(self._run_score(b2))
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)

    # -------------------------------------------------------------------- plumbing
    def _dev(self):
        p = self.word_embeddings.weight
        if not p.is_cuda:
            raise RuntimeError("ItemTransformerRanker needs its parameters on a gfx950 device "
                               "(no CPU fallback): model.to('cuda')")
        return p.device

    def _anchor(self):
        a = getattr(self, '_anchor_t', None)
        if a is None or a.device != self._dev():
            a = torch.zeros((), device=self._dev(), requires_grad=True)
            self._anchor_t = a
        return a

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        # storage may have moved: drop cached pointers
        self._plans = {}
        self._params_struct = None
        self._grads_struct = None
        self._grad_flat = None
        self._grad_views = None
        self._loss_acc = None
        self._alias = None
        self.__dict__.pop('_zg_params', None)
        self.__dict__.pop('_param_flat', None)
        if self.__dict__.get('_shard') is not None and self.product_emb.weight.device == self._shard.device:
            self._shard.attach(self.product_emb.weight)          # the receive buffer moved with the parameter
        return r

    def _named_hot_params(self):
        """(C-ABI field path, parameter) of every tensor the kernels read."""
        a = self.args
        out = [(('product_emb',), self.product_emb.weight),
               (('word_emb',), self.word_embeddings.weight),
               (('product_bias',), self.product_bias),
               (('word_bias',), self.word_bias)]
        if a.sep_prod_emb:
            out.append((('hist_product_emb',), self.hist_product_emb.weight))
        if a.query_encoder_name == 'fs':
            out += [(('fs_w',), self.query_encoder.f_W.weight), (('fs_b',), self.query_encoder.f_W.bias)]
        if a.model_name == 'item_transformer':
            te = self.transformer_encoder
            out += [(('final_ln_g',), te.layer_norm.weight), (('final_ln_b',), te.layer_norm.bias)]
            for i, l in enumerate(te.transformer_inter):
                sa, ff = l.self_attn, l.feed_forward
                out += [(('layer', i, 'wk'), sa.linear_keys.weight), (('layer', i, 'bk'), sa.linear_keys.bias),
                        (('layer', i, 'wv'), sa.linear_values.weight), (('layer', i, 'bv'), sa.linear_values.bias),
                        (('layer', i, 'wq'), sa.linear_query.weight), (('layer', i, 'bq'), sa.linear_query.bias),
                        (('layer', i, 'wo'), sa.final_linear.weight), (('layer', i, 'bo'), sa.final_linear.bias),
                        (('layer', i, 'w1'), ff.w_1.weight), (('layer', i, 'b1'), ff.w_1.bias),
                        (('layer', i, 'w2'), ff.w_2.weight), (('layer', i, 'b2'), ff.w_2.bias),
                        (('layer', i, 'ff_ln_g'), ff.layer_norm.weight), (('layer', i, 'ff_ln_b'), ff.layer_norm.bias),
                        (('layer', i, 'ln_g'), l.layer_norm.weight), (('layer', i, 'ln_b'), l.layer_norm.bias)]
        return out

    def _has_grad(self, path):
        """Parameters the reference's autograd reaches in this configuration (others keep
        grad None exactly like the reference: product_bias unless bias_product, layer-0
        pre-LN, seg_embeddings, wo)."""
        if path == ('product_bias',):
            return self.args.sim_func == 'bias_product'
        if path[0] == 'layer' and path[2] in ('ln_g', 'ln_b'):
            return path[1] != 0
        return True

    @staticmethod
    def _set_field(struct, path, value):
        if path[0] == 'layer':
            setattr(struct.layer[path[1]], path[2], value)
        else:
            setattr(struct, path[0], value)

    def _structs(self):
        if self._params_struct is not None:
            return self._params_struct, self._grads_struct
        dev = self._dev()
        hot = self._named_hot_params()
        for _, p in hot:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("parameters must be contiguous fp32")
        ps, gs = _lib.PsTemTensors(), _lib.PsTemTensors()
        for path, p in hot:
            self._set_field(ps, path, p.data_ptr())
        if self.args.model_name == 'item_transformer':
            ps.pe = self.transformer_encoder.pos_emb.pe.data_ptr()
        # one flat gradient buffer: small tensors first, tables last; 16-byte aligned slices
        graded = [(path, p) for path, p in hot if self._has_grad(path)]
        sparse = self._sparse_paths()
        # order: small dense tensors, [the sharded table's receive buffer], row-sparse tables
        graded.sort(key=lambda t: (t[0] in sparse, self._shard is not None and t[0] == ('product_emb',), t[1].numel()))
        offs, cur = [], 0
        for _, p in graded:
            offs.append(cur)
            cur += (p.numel() + 3) // 4 * 4
        pad = int(self.__dict__.get('_flat_pad_to', 4))       # dist.flatten_parameters: a multiple of 4 * world
        cur = (cur + pad - 1) // pad * pad
        self._grad_flat = torch.zeros(cur, device=dev, dtype=torch.float32)
        self._grad_views = []
        for (path, p), o in zip(graded, offs):
            v = self._grad_flat[o:o + p.numel()].view_as(p)
            self._grad_views.append((p, v))
            self._set_field(gs, path, v.data_ptr())
        self._n_small = sum((p.numel() + 3) // 4 * 4 for _, p in graded if p.numel() < (1 << 20))
        # row-sparse mode: the flat buffer is [dense tensors | row-sparse tables]; only the first part is
        # ever memset, table rows are re-zeroed by the optimizer (or zero_grad) through their touched list
        self._n_dense_grad = sum((p.numel() + 3) // 4 * 4 for path, p in graded if path not in sparse)
        # what a data-parallel exchange all-reduces: the small tensors only — with a sharded item table the receive buffer's
        # gradient (the LARGEST of the non-sparse tensors, hence the last of them) travels to the owners instead
        self._n_allreduce_grad = self._n_dense_grad - ((self.product_emb.weight.numel() + 3) // 4 * 4 if self._shard is not None else 0)
        self._sparse_tabs = [(path, p, v) for (path, p), (_, v) in zip(graded, self._grad_views) if path in sparse]
        self._params_struct, self._grads_struct = ps, gs
        self._loss_acc = torch.zeros(2, device=dev, dtype=torch.float32)
        return ps, gs

    def _check_idx(self, t, name, shape_tail=None):
        if not torch.is_tensor(t) or t.dtype != torch.int64 or not t.is_cuda:
            raise RuntimeError("batch.%s must be an int64 tensor on the model's device" % name)
        return t if t.is_contiguous() else t.contiguous()

    def _plan_for(self, batch, eval_mode):
        a = self.args
        qw = self._check_idx(batch.query_word_idxs, 'query_word_idxs')
        ui = self._check_idx(batch.u_item_idxs, 'u_item_idxs')
        B, Q = qw.shape
        L = ui.shape[1]
        C = 0
        if eval_mode:
            ca = getattr(batch, 'candi_prod_idxs', None)
            C = ca.shape[1] if (torch.is_tensor(ca) and ca.dim() == 2 and ca.shape[1] > 0) else 1   # 1: encode-only use
            W = max(1, getattr(a, 'pv_window_size', 1))
        else:
            W = batch.pos_iword_idxs.shape[1]
        training = bool(self.training) and not eval_mode
        key = (B, Q, L, W, C, training)
        plan = self._plans.get(key)
        if plan is None:
            lib = _lib.load()
            plan = _Plan()
            plan.key = key
            d = _lib.PsTemDesc()
            d.B, d.K, d.L, d.Q, d.W, d.C = B, a.neg_per_pos, L, Q, W, C
            d.d, d.H, d.F = a.embedding_size, a.heads, a.ff_size
            d.n_layers = a.inter_layers if a.model_name == 'item_transformer' else 0
            d.product_size, d.vocab_size = (self._shard.slots if self._shard is not None else self.product_size), self.vocab_size
            d.model = _lib.PS_MODEL_TEM if a.model_name == 'item_transformer' else _lib.PS_MODEL_QEM
            d.query_encoder = _lib.PS_QENC_FS if a.query_encoder_name == 'fs' else _lib.PS_QENC_AVG
            d.use_pos_emb, d.use_item_pos = int(a.use_pos_emb), int(a.use_item_pos)
            d.bias_product = int(a.sim_func == 'bias_product')
            d.pos_weight, d.sep_prod_emb = int(a.pos_weight), int(a.sep_prod_emb)
            d.training, d.dropout = int(training), float(a.dropout)
            d.seed, d.step = self._seed, 0
            lay = _lib.PsTemWsLayout()
            _lib.check(lib.ps_tem_workspace_layout(d, lay), 'ps_tem_workspace_layout')
            plan.desc, plan.layout = d, lay
            plan.ws = torch.empty(lay.total_floats, device=self._dev(), dtype=torch.float32)
            plan.batch = _lib.PsTemBatch()
            plan.neg_items = plan.neg_words = None
            plan.keep = None
            self._plans[key] = plan
        return plan

    def _fill_batch(self, plan, batch, eval_mode, neg_items=None, neg_words=None, need_negs=True):
        b = plan.batch
        qw = self._check_idx(batch.query_word_idxs, 'query_word_idxs')
        ui = self._check_idx(batch.u_item_idxs, 'u_item_idxs')
        keep = [qw, ui]
        b.query_word_idxs, b.u_item_idxs = qw.data_ptr(), ui.data_ptr()
        if eval_mode:
            ca = getattr(batch, 'candi_prod_idxs', None)
            if torch.is_tensor(ca) and ca.dim() == 2 and ca.shape[1] > 0:
                ca = self._check_idx(ca, 'candi_prod_idxs')
                b.candi_prod_idxs = ca.data_ptr()
                keep.append(ca)
            else:
                b.candi_prod_idxs = None
        else:
            tg = self._check_idx(batch.target_prod_idxs, 'target_prod_idxs')
            pw = self._check_idx(batch.pos_iword_idxs, 'pos_iword_idxs')
            b.target_prod_idxs, b.pos_iword_idxs = tg.data_ptr(), pw.data_ptr()
            if need_negs:
                ni = self._check_idx(neg_items, 'neg_item_idxs')
                nw = self._check_idx(neg_words, 'neg_word_idxs')
                d = plan.desc
                if ni.numel() != d.B * d.K or nw.numel() != d.B * d.W * d.K:
                    raise RuntimeError("negative sample shapes: want [%d,%d] and [%d,%d]" % (d.B, d.K, d.B, d.W * d.K))
                b.neg_item_idxs, b.neg_word_idxs = ni.data_ptr(), nw.data_ptr()
                keep += [tg, pw, ni, nw]
            else:                          # drawn on the device inside the step's prologue
                b.neg_item_idxs = b.neg_word_idxs = None
                keep += [tg, pw]
        plan.keep = keep       # keep the index tensors alive until backward has run

    def _stream(self):
        return torch.cuda.current_stream(self._dev()).cuda_stream

    def _alias_tables(self):
        if self._alias is None:
            if self.word_dists is None:
                raise RuntimeError("word_dists is required to sample negative words "
                                   "(or pass neg_word_idxs= explicitly)")
            lib = _lib.load()
            wd = self.word_dists.contiguous()
            n = wd.numel()
            prob = torch.empty(n, dtype=torch.float32)
            alias = torch.empty(n, dtype=torch.int32)
            _lib.check(lib.ps_build_alias_host(wd.data_ptr(), n, prob.data_ptr(), alias.data_ptr()),
                       'ps_build_alias_host')
            self._alias = (prob.to(self._dev()), alias.to(self._dev()))
        return self._alias

    def sample_negatives(self, plan):
        """The two ``torch.multinomial`` draws (item_transformer.py:447, :268) on the device."""
        lib = _lib.load()
        d = plan.desc
        if plan.neg_items is None:
            plan.neg_items = torch.empty(d.B, d.K, device=self._dev(), dtype=torch.int64)
            plan.neg_words = torch.empty(d.B, d.W * d.K, device=self._dev(), dtype=torch.int64)
        prob, alias = self._alias_tables()
        if self._shard is not None:                 # the plan's descriptor counts SLOTS; negatives are drawn over the catalogue
            d = _lib.PsTemDesc.from_buffer_copy(d)
            d.product_size = self.product_size
        _lib.check(lib.ps_sample_negatives(d, prob.data_ptr(), alias.data_ptr(), plan.neg_items.data_ptr(),
                                           plan.neg_words.data_ptr(), self._stream()), 'ps_sample_negatives')
        return plan.neg_items, plan.neg_words

    def _use_step_api(self):
        """Graph-replayed step entry points (ps_tem_forward_step / ps_tem_backward_step), opt-in with PS_GRAPHS=1 (they
        cut host time per step, not device time); the row-sparse mode keeps the eager ones (its touched-row lists read
        the caller's index tensors)."""
        return bool(_lib.load().ps_graph_replay_enabled()) and not self._row_sparse()

    def _run_forward(self, batch, neg_items=None, neg_words=None):
        lib = _lib.load()
        if self._lazy_exact():
            # catch_up_rows() (below) advances the touched rows' "steps applied" to T + 1 on the promise that optim.step() T + 1
            # follows: a second training forward before that step would leave the first one's rows one replay short of the dense
            # optimizer — refused here, before anything of this forward has happened (the first forward stays backward-able)
            opt = self.__dict__.get('_lazy_optim')
            opt = opt() if opt is not None else None
            if opt is not None and opt._lazy_awaiting_step:
                raise RuntimeError("lazy_exact_adam: a second training forward without an optimizer step in between (the rows of "
                                   "the first were already advanced to the coming step); call optim.step() after every "
                                   "training forward, or model.eval() for forwards that do not train")
        ps, _ = self._structs()
        plan = self._plan_for(batch, eval_mode=False)
        self.__dict__['_last_plan'] = plan
        self._fwd_step += 1
        plan.desc.step = self._fwd_step
        loss3 = torch.empty(3, device=self._dev(), dtype=torch.float32)
        if self._shard is not None:
            # sharded item table: every item id of the step must be known BEFORE the forward (the rows are fetched from their
            # owners), so the two draws are a launch of their own; then lookup -> remapped indices -> the unchanged step
            import copy
            if neg_items is None or neg_words is None:
                d = plan.desc
                if plan.neg_items is None:
                    plan.neg_items = torch.empty(d.B, d.K, device=self._dev(), dtype=torch.int64)
                    plan.neg_words = torch.empty(d.B, d.W * d.K, device=self._dev(), dtype=torch.int64)
                neg_items, neg_words = self.sample_negatives(plan)
            tem = self.args.model_name == 'item_transformer'
            tg = self._check_idx(batch.target_prod_idxs, 'target_prod_idxs')
            ni = self._check_idx(neg_items, 'neg_item_idxs')
            lists = [tg, ni] + ([self._check_idx(batch.u_item_idxs, 'u_item_idxs')] if tem else [])
            rem = self._shard.lookup(lists)
            b2 = copy.copy(batch)
            b2.target_prod_idxs = rem[0]
            if tem:
                b2.u_item_idxs = rem[2]
            self._fill_batch(plan, b2, False, rem[1], neg_words)
            plan.staged = False
            _lib.check(lib.ps_tem_forward(plan.desc, ps, plan.batch, plan.ws.data_ptr(), loss3.data_ptr(),
                                          self._loss_acc.data_ptr(), self._stream()), 'ps_tem_forward')
            return plan, loss3
        if self._lazy_exact():
            # every row the step reads must be current BEFORE the forward: the draws as a launch of their own, the touched lists
            # now (not under the backward), the optimizer's replay of the steps those rows missed, then the unchanged step
            opt = self.__dict__.get('_lazy_optim')
            opt = opt() if opt is not None else None
            if opt is None:
                raise RuntimeError("lazy_exact_adam: build_optim(args, model, ...) must run before the first training forward")
            if neg_items is None or neg_words is None:
                d = plan.desc
                if plan.neg_items is None:
                    plan.neg_items = torch.empty(d.B, d.K, device=self._dev(), dtype=torch.int64)
                    plan.neg_words = torch.empty(d.B, d.W * d.K, device=self._dev(), dtype=torch.int64)
                neg_items, neg_words = self.sample_negatives(plan)
            self._fill_batch(plan, batch, False, neg_items, neg_words)
            self._structs()
            for path, p, gview in self._sparse_tabs:          # a backward no optimizer step consumed: its rows, before the lists change
                info = getattr(p, '_ps_rows', None)
                if info is not None and info['dirty'] and info.get('has_grad'):
                    rows, count, cap = self._rows_view(info)
                    _lib.check(lib.ps_zero_rows(gview.data_ptr(), p.shape[1], rows.data_ptr(), count.data_ptr(), cap,
                                                self._stream()), 'ps_zero_rows')
                    info['has_grad'] = False
            self._coalesce_touched(plan)
            plan.coalesced_at = self._fwd_step
            opt.catch_up_rows()
            opt._lazy_awaiting_step = True
            plan.staged = False
            _lib.check(lib.ps_tem_forward(plan.desc, ps, plan.batch, plan.ws.data_ptr(), loss3.data_ptr(),
                                          self._loss_acc.data_ptr(), self._stream()), 'ps_tem_forward')
            return plan, loss3
        if self._use_step_api():
            inject = neg_items is not None and neg_words is not None
            self._fill_batch(plan, batch, False, neg_items if inject else None, neg_words if inject else None,
                             need_negs=inject)
            prob, alias = (None, None) if inject else self._alias_tables()
            _lib.check(lib.ps_tem_forward_step(plan.desc, ps, plan.batch, _lib.ptr(prob), _lib.ptr(alias),
                                               plan.ws.data_ptr(), loss3.data_ptr(), self._loss_acc.data_ptr(),
                                               self._stream()), 'ps_tem_forward_step')
            plan.staged = True
            return plan, loss3
        plan.staged = False
        if neg_items is None or neg_words is None:
            # the two draws ride in the forward's first launch; plan.neg_* receive them for the backward
            d = plan.desc
            if plan.neg_items is None:
                plan.neg_items = torch.empty(d.B, d.K, device=self._dev(), dtype=torch.int64)
                plan.neg_words = torch.empty(d.B, d.W * d.K, device=self._dev(), dtype=torch.int64)
            self._fill_batch(plan, batch, False, plan.neg_items, plan.neg_words)
            prob, alias = self._alias_tables()
            _lib.check(lib.ps_tem_forward_sampled(plan.desc, ps, plan.batch, prob.data_ptr(), alias.data_ptr(),
                                                  plan.neg_items.data_ptr(), plan.neg_words.data_ptr(),
                                                  plan.ws.data_ptr(), loss3.data_ptr(), self._loss_acc.data_ptr(),
                                                  self._stream()), 'ps_tem_forward_sampled')
            return plan, loss3
        self._fill_batch(plan, batch, False, neg_items, neg_words)
        _lib.check(lib.ps_tem_forward(plan.desc, ps, plan.batch, plan.ws.data_ptr(), loss3.data_ptr(),
                                      self._loss_acc.data_ptr(), self._stream()), 'ps_tem_forward')
        return plan, loss3

    # ------------------------------------------------------------ row-sparse optimizer support
    _SPARSE_PATHS = (('product_emb',), ('word_emb',), ('hist_product_emb',))

    def _row_sparse(self):
        """``args.row_sparse_adam`` (extension, default False; implied by ``args.shard_tables``): table gradients stay dense
        tensors but zeroing / clip / Adam / exchange only visit the rows a step touched (BASELINE configs[4])."""
        return bool(getattr(self.args, 'row_sparse_adam', False)) or self._lazy_exact() or self.__dict__.get('_shard') is not None

    def _lazy_exact(self):
        """``args.lazy_exact_adam``: the row-sparse machinery with the DENSE optimizer's results (optimizers.py:241-243 moves
        every row every step): the rows a step addresses are first brought up to date by replaying the steps they missed
        (``Optimizer.catch_up_rows`` -> ``ps_rowsparse_catchup``), all other rows are stale until ``_lazy_flush``."""
        return bool(getattr(self.args, 'lazy_exact_adam', False))

    def _lazy_flush(self):
        if self._lazy_exact():
            opt = self.__dict__.get('_lazy_optim')
            opt = opt() if opt is not None else None
            if opt is not None:
                opt.flush_rows()

    def _sparse_paths(self):
        """Tables handled by their touched-row lists on THIS rank (a sharded item table is not: its receive buffer is a
        small dense tensor whose gradient goes to the owners)."""
        if not self._row_sparse():
            return ()
        return tuple(p for p in self._SPARSE_PATHS if not (self._shard is not None and p == ('product_emb',)))

    def _index_lists(self, path, plan):
        """Index tensors of the step that address ``path``'s rows, and that table's pad row."""
        tg, pw, ni, nw = plan.keep[2:6]
        qw, ui = plan.keep[0], plan.keep[1]
        tem = self.args.model_name == 'item_transformer'
        if path == ('product_emb',):
            return [tg, ni] + ([ui] if tem and not self.args.sep_prod_emb else []), self.prod_pad_idx
        if path == ('hist_product_emb',):
            return ([ui] if tem else []), self.prod_pad_idx
        return [qw, pw, nw], self.word_pad_idx

    def _index_cap_bound(self, path):
        """Largest index count one step of this rank can address in ``path``'s table under the model's flags, for the
        batch size of the latest forward: the history width is padded per batch (<= uprev_review_limit,
        item_pv_dataloader.py:121-143) while Q is the corpus-wide padded query length.  The data-parallel row exchange
        sizes its fixed-capacity messages with the maximum of this over the ranks (dist.SparseGradExchange)."""
        plan = self.__dict__.get('_last_plan')
        if plan is None:
            return 1
        d, a = plan.desc, self.args
        tem = a.model_name == 'item_transformer'
        Lmax = max(int(d.L), int(getattr(a, 'uprev_review_limit', d.L)))
        if path == ('product_emb',):
            return d.B * (1 + d.K + (Lmax if tem and not a.sep_prod_emb else 0))
        if path == ('hist_product_emb',):
            return d.B * Lmax if tem else 1
        return d.B * (d.Q + d.W + d.W * d.K)

    def _coalesce_touched(self, plan):
        lib = _lib.load()
        dev, st = self._dev(), self._stream()
        for path, p, gview in self._sparse_tabs:
            lists, pad = self._index_lists(path, plan)
            info = getattr(p, '_ps_rows', None)
            total = sum(t.numel() for t in lists)
            cap = max(1, min(total, p.shape[0]))
            if info is None or info['rows'].numel() < cap or info['grad_ptr'] != gview.data_ptr():
                info = dict(rows=torch.empty(cap, device=dev, dtype=torch.int64),
                            count=torch.zeros(1, device=dev, dtype=torch.int32),
                            ws=torch.zeros(lib.ps_coalesce_ws_bytes(p.shape[0]), device=dev, dtype=torch.uint8),
                            grad_ptr=gview.data_ptr(), dirty=False)
                p._ps_rows = info
            info['cap'] = cap
            info['active'] = None          # a data-parallel exchange installs the ranks' union here (dist.SparseGradExchange)
            if not lists:
                info['count'].zero_()
                continue
            arr = (_lib.PsIdxList * len(lists))()
            for i, t in enumerate(lists):
                arr[i].idx, arr[i].n = t.data_ptr(), t.numel()
            _lib.check(lib.ps_coalesce_rows(arr, len(lists), p.shape[0], pad, info['ws'].data_ptr(),
                                            info['rows'].data_ptr(), info['rows'].numel(),
                                            info['count'].data_ptr(), st), 'ps_coalesce_rows')
            info['dirty'] = True

    @staticmethod
    def _rows_view(info):
        """(rows, count, cap) the optimizer and zero_grad walk: the ranks' union after an exchange, else this rank's."""
        return info.get('active') or (info['rows'], info['count'], info['cap'])

    def touched_rows(self):
        """{state_dict key: sorted unique row ids} of the last backward (row-sparse mode; host sync)."""
        self.check_index_errors()
        out = {}
        for name, p in self.named_parameters():
            info = getattr(p, '_ps_rows', None)
            if info is not None:
                rows, count, _ = self._rows_view(info)
                out[name] = rows[:int(count[0])].clone()
        return out

    def check_index_errors(self):
        """Row-sparse mode: raise if a step's index lists held a row outside its table or overflowed a touched list
        (the kernels drop such rows and set a status word; reading it is a host sync, so the step itself never does —
        the trainer calls this where it reads the loss)."""
        import ctypes
        lib = _lib.load()
        if self.__dict__.get('_shard') is not None:
            self._shard.check_errors()
        for name, p in self.named_parameters():
            info = getattr(p, '_ps_rows', None)
            if info is None:
                continue
            for ws in (info['ws'], (info.get('xchg') or {}).get('ws')):
                if ws is None:
                    continue
                addr = lib.ps_coalesce_bad_flag(ws.data_ptr(), p.shape[0])
                off = addr - ws.data_ptr()
                flag = int(ws[off:off + 4].view(torch.int32)[0])
                if flag:
                    ws[off:off + 4].zero_()
                    raise RuntimeError("row-sparse step: %s" % ("an index outside [0, %d) addressed %s" % (p.shape[0], name)
                                       if flag == 1 else "touched-row list of %s overflowed its capacity" % name))

    def _zero_for_backward(self):
        lib = _lib.load()
        st = self._stream()
        if not self._row_sparse():
            _lib.check(lib.ps_zero_floats(self._grad_flat.data_ptr(), self._grad_flat.numel(), st), 'ps_zero_floats')
            return
        _lib.check(lib.ps_zero_floats(self._grad_flat.data_ptr(), self._n_dense_grad, st), 'ps_zero_floats')
        for path, p, gview in self._sparse_tabs:
            info = getattr(p, '_ps_rows', None)
            if info is not None and info['dirty']:       # touched by a backward no optimizer step consumed
                rows, count, cap = self._rows_view(info)
                _lib.check(lib.ps_zero_rows(gview.data_ptr(), p.shape[1], rows.data_ptr(), count.data_ptr(), cap, st),
                           'ps_zero_rows')
                info['dirty'] = False

    def zero_grad(self, set_to_none=True):
        """``model.zero_grad()`` (trainer.py:76).  nn.Module's version walks the module tree on every call (≈120 us of
        host time per step here); the parameters are fixed, so their list is cached.  The flat gradient buffer itself
        is zeroed by the next backward (``_assign_grads`` sees the ``None`` grads)."""
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        plist = self.__dict__.get('_zg_params')
        if plist is None:
            plist = self.__dict__['_zg_params'] = list(self.parameters())
        for p in plist:
            p.grad = None

    def _assign_grads(self):
        """Give every reachable parameter its dense ``.grad`` view; returns True if the flat
        buffer must be zeroed first (i.e. zero_grad() ran, trainer.py:76)."""
        fresh = self._grad_views[0][0].grad is None
        for p, v in self._grad_views:
            if p.grad is None:
                p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():
                raise RuntimeError("a foreign .grad tensor is attached to a hot-path parameter; "
                                   "call model.zero_grad() before backward")
        return fresh

    def _run_backward(self, plan, grad_out):
        lib = _lib.load()
        ps, gs = self._structs()
        st = self._stream()
        if grad_out is None and getattr(plan, 'staged', False) and self._use_step_api():
            fresh = self._assign_grads() and not self.__dict__.get('_grad_clean', False)   # zero_grad() folded into the replay
            self.__dict__['_grad_clean'] = False
            _lib.check(lib.ps_tem_backward_step(plan.desc, ps, plan.ws.data_ptr(), gs, 1.0,
                                                self._grad_flat.data_ptr() if fresh else None,
                                                self._grad_flat.numel() if fresh else 0, st), 'ps_tem_backward_step')
            return
        batch_struct = plan.batch
        if getattr(plan, 'staged', False):           # explicit upstream gradient: eager backward over the staged inputs
            batch_struct = _lib.PsTemBatch()
            _lib.check(lib.ps_tem_staged_batch(plan.desc, plan.ws.data_ptr(), batch_struct), 'ps_tem_staged_batch')
        if self._assign_grads():
            if not self.__dict__.get('_grad_clean', False):     # (clean: the last optimizer step left the buffer at 0)
                self._zero_for_backward()
        elif self._row_sparse() and any(getattr(p, '_ps_rows', {}).get('dirty') for _, p, _ in self._sparse_tabs):
            raise NotImplementedError("row_sparse_adam: gradient accumulation over several backwards is not "
                                      "supported; call model.zero_grad() (trainer.py:76) or optim.step() first")
        pre = self._row_sparse() and getattr(plan, 'coalesced_at', None) == self._fwd_step     # lazy_exact_adam: done before the forward
        if self._row_sparse() and not pre:
            # the touched lists only need the step's indices: built on a side stream under the backward
            main = torch.cuda.current_stream(self._dev())
            side = getattr(self, '_side_stream', None)
            if side is None:
                side = self._side_stream = torch.cuda.Stream(self._dev())
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self._coalesce_touched(plan)
        if self._row_sparse():
            for _, p, _ in self._sparse_tabs:
                if getattr(p, '_ps_rows', None) is not None:
                    p._ps_rows['has_grad'] = True
        self.__dict__['_grad_clean'] = False
        go = None if grad_out is None else grad_out.contiguous().float()      # None: d loss / d loss = 1
        _lib.check(lib.ps_tem_backward(plan.desc, ps, batch_struct, plan.ws.data_ptr(), gs, 1.0,
                                       _lib.ptr(go), st), 'ps_tem_backward')
        if self._row_sparse() and not pre:
            main.wait_stream(side)
        if self._shard is not None:
            self._shard.pending = True           # the receive buffer's gradient waits for push_grads (Optimizer.step)

    def encode(self, batch):
        """Eval-mode sequence representation [B,d] that the dot-product head scores items with
        (item_transformer.py:118-131): one encode per (user, query) row.  Used by ``evaluate.rank_all``."""
        self._lazy_flush()
        lib = _lib.load()
        ps, _ = self._structs()
        plan = self._plan_for(batch, eval_mode=True)
        if self._shard is not None:
            # row-sharded item table: the history rows (and the targets', for evaluate.rank_all) come from their owners first —
            # a collective, like the training forward's lookup; the encode then reads the receive buffer through remapped ids
            import copy
            tem = self.args.model_name == 'item_transformer'
            tg = self._check_idx(batch.target_prod_idxs, 'target_prod_idxs')
            lists = [tg] + ([self._check_idx(batch.u_item_idxs, 'u_item_idxs')] if tem else [])
            rem = self._shard.lookup(lists)
            batch = copy.copy(batch)
            batch.target_prod_idxs = rem[0]
            if tem:
                batch.u_item_idxs = rem[1]
            self.__dict__['_shard_eval_target_slots'] = rem[0]
        self._fill_batch(plan, batch, True)
        enc = torch.empty(plan.desc.B, plan.desc.d, device=self._dev(), dtype=torch.float32)
        _lib.check(lib.ps_tem_encode(plan.desc, ps, plan.batch, plan.ws.data_ptr(), enc.data_ptr(), self._stream()),
                   'ps_tem_encode')
        return enc

    def _run_score(self, batch):
        self._lazy_flush()
        lib = _lib.load()
        ps, _ = self._structs()
        plan = self._plan_for(batch, eval_mode=True)
        self._fill_batch(plan, batch, True)
        if not plan.batch.candi_prod_idxs:
            raise RuntimeError("test(): batch.candi_prod_idxs [B,C] is required")
        d = plan.desc
        scores = torch.empty(d.B, d.C, device=self._dev(), dtype=torch.float32)
        _lib.check(lib.ps_tem_score(d, ps, plan.batch, plan.ws.data_ptr(), scores.data_ptr(), self._stream()),
                   'ps_tem_score')
        return scores

    # --------------------------------------------------------------- test support
    def workspace_view(self, plan, name, shape):
        """View of one intermediate inside the workspace (parity tests compare every stage)."""
        off = getattr(plan.layout, name)
        n = 1
        for s in shape:
            n *= s
        return plan.ws[off:off + n].view(*shape)
