"""ProductRanker — drop-in boundary of the RTM (review_transformer) ranking-loss step.

Mirrors the reference's ``nn.Module`` contract (``models/ps_model.py:53-370``; call sites
``main.py:144-146``, ``trainer.py:74-79,190-201,225``):

    model = ProductRanker(args, device, vocab_size, review_count, product_size, user_size,
                          review_words, vocab_words, word_dists)
    loss = model(batch, train_pv)           # 0-dim fp32 tensor with a grad_fn
    model.zero_grad(); loss.backward(); optim.step()
    model.get_review_embeddings(); scores = model.test(batch); model.clear_review_embbeddings()

Same ``state_dict`` keys as the reference (incl. the aliases ``review_encoder.word_embeddings.weight``
and, for pvc, ``review_encoder.context_embeddings.weight``).  Supported review encoders: ``pv``, ``pvc`` (the reference
default), ``fs`` and ``avg`` (``ps_model.py:148-151``, ``:301-305``; ``models/text_encoder.py``), with or without the
per-position user / item embeddings (``use_user_emb`` / ``use_item_emb``); ``fix_emb`` is outside the built path and
raises ``NotImplementedError``.  All numerics run in
``libprodsearch_hip.so`` (``ps_rtm_*`` entry points); the torch modules are parameter holders.
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .item_transformer import _FSEncoder, _Holder, _TransformerEncoder, _init_like_reference


class _ReviewEncoder(_Holder):
    """Holder with the reference's attribute names (PV.py:18-34, PVC.py:18-30)."""
    def __init__(self, word_embeddings, name, review_count, d):
        super().__init__()
        self.word_embeddings = word_embeddings
        if name == 'pv':
            self.review_embeddings = nn.Embedding(review_count, d, padding_idx=review_count - 1)
        else:
            self.context_embeddings = word_embeddings


class _RtmLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, plan, loss3):
        ctx.model, ctx.plan, ctx.step = model, plan, model._fwd_step
        return loss3[0]

    @staticmethod
    def backward(ctx, grad_out):
        from .item_transformer import _check_same_forward
        _check_same_forward(ctx.model, ctx.step)
        ctx.model._run_backward(ctx.plan, grad_out)
        return None, None, None, None


class ProductRanker(nn.Module):
    def __init__(self, args, device, vocab_size, review_count, product_size, user_size,
                 review_words, vocab_words, word_dists=None):
        super(ProductRanker, self).__init__()
        if args.review_encoder_name not in ('pv', 'pvc', 'fs', 'avg'):
            raise NotImplementedError("review_encoder_name %r: pv / pvc / fs / avg are built" % args.review_encoder_name)
        if getattr(args, 'fix_emb', False):
            raise NotImplementedError("fix_emb is outside the built path")
        if getattr(args, 'pretrain_emb_dir', '') or getattr(args, 'pretrain_up_emb_dir', ''):
            raise NotImplementedError("pretrained-embedding text loaders are out of scope; load a state_dict")
        self.args = args
        self.device = device
        self.train_review_only = args.train_review_only
        self.embedding_size = d = args.embedding_size
        self.vocab_words = vocab_words
        self.vocab_size, self.review_count = vocab_size, review_count
        self.word_dists = None if word_dists is None else torch.as_tensor(word_dists, dtype=torch.float64)
        self.prod_pad_idx, self.user_pad_idx = product_size, user_size
        self.word_pad_idx = vocab_size - 1
        self.seg_pad_idx = 3
        self.review_pad_idx = review_count - 1
        self.review_encoder_name = args.review_encoder_name
        if not torch.is_tensor(review_words) and not getattr(args, 'do_subsample_mask', False) and \
                len({len(x) for x in review_words}) > 1:
            # ps_model.py:75-78: ragged review texts are cut / padded to review_word_limit (others/util.py:pad)
            lim = int(args.review_word_limit)
            review_words = [list(x[:lim]) + [self.word_pad_idx] * (lim - len(x)) for x in review_words]
        rw = torch.as_tensor(np.asarray(review_words), dtype=torch.int64) if not torch.is_tensor(review_words) \
            else review_words.to(torch.int64)
        if rw.dim() != 2:
            raise ValueError("review_words must be a padded [review_count, review_word_limit] table "
                             "(the reference pads it with others/util.py:pad)")
        self.review_words = rw.to(device)                       # plain attribute, as in the reference (:79)

        # registration order follows ps_model.py:88-126 so state_dicts line up key for key
        self.use_user_emb, self.use_item_emb = bool(args.use_user_emb), bool(args.use_item_emb)
        if self.use_user_emb:
            self.user_emb = nn.Embedding(user_size + 1, d, padding_idx=self.user_pad_idx)
        if self.use_item_emb:
            self.product_emb = nn.Embedding(product_size + 1, d, padding_idx=self.prod_pad_idx)
        self.word_embeddings = nn.Embedding(vocab_size, d, padding_idx=self.word_pad_idx)
        self.transformer_encoder = _TransformerEncoder(d, args.ff_size, args.inter_layers)
        if self.review_encoder_name == 'fs':
            self.review_encoder = _FSEncoder(d)              # review_encoder.f_W (ps_model.py:148-149)
        elif self.review_encoder_name == 'avg':
            self.review_encoder = _Holder()                  # AVGEncoder has no parameters (:150-151)
        else:
            self.review_encoder = _ReviewEncoder(self.word_embeddings, self.review_encoder_name, review_count, d)
        self.query_encoder = _FSEncoder(d) if args.query_encoder_name == 'fs' else _Holder()
        self.seg_embeddings = nn.Embedding(4, d, padding_idx=self.seg_pad_idx)
        self.review_embeddings = None
        self.initialize_parameters()
        self.to(device)
        self._reset_cache()
        self._fwd_step = 0
        self._seed = int(getattr(args, 'seed', 666))

    # ---------------------------------------------------------------- reference API
    def initialize_parameters(self, logger=None):
        """ps_model.py:360-370 (+ PV.py:82-90)."""
        nn.init.normal_(self.word_embeddings.weight)
        nn.init.normal_(self.seg_embeddings.weight)
        if self.review_encoder_name == 'pv':
            nn.init.normal_(self.review_encoder.review_embeddings.weight)
        elif self.review_encoder_name == 'fs':
            _init_like_reference(self.review_encoder)
        if self.args.query_encoder_name == 'fs':
            _init_like_reference(self.query_encoder)
        _init_like_reference(self.transformer_encoder)

    def load_cp(self, pt, strict=True):
        self.load_state_dict(pt['model'], strict=strict)

    def clear_review_embbeddings(self):           # (sic) reference spelling, ps_model.py:177
        self.review_embeddings = None

    def get_review_embeddings(self, batch_size=128):
        """ps_model.py:186-203: the table ``test`` indexes (pv: the parameter itself; pvc: one launch
        computing the uncorrupted mean of every review's words, last row 0)."""
        if self.review_embeddings is not None:
            return
        if self.review_encoder_name == 'pv':
            self.review_embeddings = self.review_encoder.review_embeddings.weight
            return
        lib = _lib.load()
        ps, _ = self._structs()
        d = self._desc(1, 0, 1, eval_mode=True, C=1)
        rw = self.review_words if self.review_words.is_cuda else self.review_words.to(self._dev())
        self.review_words = rw.contiguous()
        d.WL = rw.shape[1]
        out = torch.empty(self.review_count, self.embedding_size, device=self._dev(), dtype=torch.float32)
        scratch = torch.empty_like(out) if self.review_encoder_name == 'fs' else None
        _lib.check(lib.ps_rtm_review_embeddings(d, ps, self.review_words.data_ptr(), _lib.ptr(scratch), out.data_ptr(),
                                                self._stream()), 'ps_rtm_review_embeddings')
        self.review_embeddings = out

    def forward(self, batch_data, train_pv=True, neg_word_idxs=None):
        plan, loss3 = self._run_forward(batch_data, bool(train_pv), neg_word_idxs)
        if not torch.is_grad_enabled():
            return loss3[0]
        # same direct path as ItemTransformerRanker: a plain ``loss.backward()`` calls the HIP backward without the
        # autograd engine's round trip and its ``ones_like`` fill kernel (item_transformer._LossTensor)
        from .item_transformer import _LossTensor
        out = _RtmLossFn.apply(self._anchor(), self, plan, loss3).as_subclass(_LossTensor)
        out._ps_fast = (self, plan, self._fwd_step)
        return out

    def test(self, batch_data):
        return self._run_score(batch_data)

    # -------------------------------------------------------------------- plumbing
    def _reset_cache(self):
        self._plans = {}
        self._params_struct = self._grads_struct = None
        self._grad_flat = self._grad_views = None
        self._alias = None

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._reset_cache()
        return r

    def _dev(self):
        p = self.word_embeddings.weight
        if not p.is_cuda:
            raise RuntimeError("ProductRanker needs its parameters on a gfx950 device (no CPU fallback): "
                               "model.to('cuda')")
        return p.device

    def _stream(self):
        return torch.cuda.current_stream(self._dev()).cuda_stream

    def _anchor(self):
        a = getattr(self, '_anchor_t', None)
        if a is None or a.device != self._dev():
            a = torch.zeros((), device=self._dev(), requires_grad=True)
            self._anchor_t = a
        return a

    def _named_hot_params(self):
        a, te = self.args, self.transformer_encoder
        out = [(('word_emb',), self.word_embeddings.weight), (('seg_emb',), self.seg_embeddings.weight),
               (('final_ln_g',), te.layer_norm.weight), (('final_ln_b',), te.layer_norm.bias),
               (('wo_w',), te.wo.weight), (('wo_b',), te.wo.bias)]
        if self.review_encoder_name == 'pv':
            out.append((('review_emb',), self.review_encoder.review_embeddings.weight))
        if self.review_encoder_name == 'fs':
            out += [(('rev_fs_w',), self.review_encoder.f_W.weight), (('rev_fs_b',), self.review_encoder.f_W.bias)]
        if self.use_user_emb:
            out.append((('user_emb',), self.user_emb.weight))
        if self.use_item_emb:
            out.append((('product_emb',), self.product_emb.weight))
        if a.query_encoder_name == 'fs':
            out += [(('fs_w',), self.query_encoder.f_W.weight), (('fs_b',), self.query_encoder.f_W.bias)]
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
        if path == ('seg_emb',):
            return bool(self.args.use_seg_emb)
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
        ps, gs = _lib.PsRtmTensors(), _lib.PsRtmTensors()
        for path, p in hot:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("parameters must be contiguous fp32")
            self._set_field(ps, path, p.data_ptr())
        ps.pe = self.transformer_encoder.pos_emb.pe.data_ptr()
        graded = [(path, p) for path, p in hot if self._has_grad(path)]
        graded.sort(key=lambda t: t[1].numel())
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
        self._params_struct, self._grads_struct = ps, gs
        return ps, gs

    def _desc(self, B, K, R, eval_mode, C=0, Q=1, W=1, WL=0, train_pv=False):
        a = self.args
        d = _lib.PsRtmDesc()
        d.B, d.K, d.R, d.Q, d.W, d.WL, d.C = B, K, R, Q, W, WL, C
        d.d, d.H, d.F, d.n_layers = a.embedding_size, a.heads, a.ff_size, a.inter_layers
        d.vocab_size, d.review_count = self.vocab_size, self.review_count
        d.review_encoder = dict(pv=_lib.PS_RENC_PV, pvc=_lib.PS_RENC_PVC, fs=_lib.PS_RENC_FS, avg=_lib.PS_RENC_AVG)[self.review_encoder_name]
        d.query_encoder = _lib.PS_QENC_FS if a.query_encoder_name == 'fs' else _lib.PS_QENC_AVG
        d.use_pos_emb, d.use_seg_emb, d.pos_weight = int(a.use_pos_emb), int(a.use_seg_emb), int(a.pos_weight)
        d.train_pv = int(train_pv)
        d.training = int(bool(self.training) and not eval_mode)
        d.dropout, d.corrupt_rate = float(a.dropout), float(a.corrupt_rate)
        d.seed, d.step = self._seed, 0
        d.use_user_emb, d.use_item_emb = int(self.use_user_emb), int(self.use_item_emb)
        d.user_size, d.product_size = self.user_pad_idx, self.prod_pad_idx
        return d

    def _seq_ids(self, bt, batch, names, shape, keep):
        """Per-position user / item ids (ps_model.py:325-334, :233-238): [.., R+1] int64, position 0 = pad."""
        want = []
        if self.use_user_emb:
            want += [n for n in names if 'user' in n]
        if self.use_item_emb:
            want += [n for n in names if 'item' in n]
        for n in names:
            setattr(bt, n, None)
        for n in want:
            t = self._idx(getattr(batch, n, None), n)
            if t is None:
                raise RuntimeError("use_user_emb / use_item_emb: the batch lacks %s" % n)
            exp = shape[n]
            if tuple(t.shape) != exp:
                raise RuntimeError("batch.%s has shape %s, expected %s" % (n, tuple(t.shape), exp))
            setattr(bt, n, t.data_ptr())
            keep.append(t)

    @staticmethod
    def _idx(t, name, dtype=torch.int64):
        if t is None:
            return None
        if not torch.is_tensor(t) or t.dtype != dtype or not t.is_cuda:
            raise RuntimeError("batch.%s must be a %s tensor on the model's device" % (name, dtype))
        return t if t.is_contiguous() else t.contiguous()

    def _plan(self, key, desc, eval_mode):
        plan = self._plans.get(key)
        if plan is None:
            lib = _lib.load()
            import ctypes as C
            tot = C.c_int64(0)
            _lib.check(lib.ps_rtm_workspace_floats(desc, int(eval_mode), C.byref(tot)), 'ps_rtm_workspace_floats')
            lay = _lib.PsRtmWsLayout()
            _lib.check(lib.ps_rtm_workspace_layout(desc, int(eval_mode), lay), 'ps_rtm_workspace_layout')
            plan = dict(desc=desc, batch=_lib.PsRtmBatch(), ws=torch.empty(tot.value, device=self._dev(), dtype=torch.float32),
                        neg_words=None, keep=None, layout=lay)
            self._plans[key] = plan
        return plan

    def _alias_tables(self):
        if self._alias is None:
            if self.word_dists is None:
                raise RuntimeError("word_dists is required to sample the PV-loss words (or pass neg_word_idxs=)")
            lib = _lib.load()
            wd = self.word_dists.contiguous()
            n = wd.numel()
            prob, alias = torch.empty(n, dtype=torch.float32), torch.empty(n, dtype=torch.int32)
            _lib.check(lib.ps_build_alias_host(wd.data_ptr(), n, prob.data_ptr(), alias.data_ptr()), 'ps_build_alias_host')
            self._alias = (prob.to(self._dev()), alias.to(self._dev()))
        return self._alias

    def _sample_pv_words(self, plan, n_rev, W, K):
        """``torch.multinomial(word_dists, B*R*W*K)`` (PV.py:57 / PVC.py:81) on the device."""
        lib = _lib.load()
        if plan['neg_words'] is None:
            plan['neg_words'] = torch.empty(n_rev, W * K, device=self._dev(), dtype=torch.int64)
            plan['dummy_items'] = torch.empty(1, device=self._dev(), dtype=torch.int64)
        sd = _lib.PsTemDesc()
        sd.B, sd.K, sd.W = n_rev, K, W
        sd.product_size, sd.vocab_size = 1, self.vocab_size
        sd.seed, sd.step = self._seed, self._fwd_step
        prob, alias = self._alias_tables()
        # item draws are not needed here: ask for zero of them by sampling only the word stream (B*W*K words)
        items = torch.empty(n_rev * K, device=self._dev(), dtype=torch.int64)
        _lib.check(lib.ps_sample_negatives(sd, prob.data_ptr(), alias.data_ptr(), items.data_ptr(),
                                           plan['neg_words'].data_ptr(), self._stream()), 'ps_sample_negatives')
        return plan['neg_words']

    _POS_SEQ = (('pos_prod_ridxs', 1, 'review'), ('pos_seg_idxs', 1, 'seg'), ('pos_user_idxs', 1, 'user'),
                ('pos_item_idxs', 1, 'item'), ('pos_prod_rword_idxs', 1, 'word3'), ('pos_prod_rword_masks', 1, 'mask3'),
                ('pos_prod_rword_idxs_pvc', 1, 'word3'))
    _NEG_SEQ = (('neg_prod_ridxs', 2, 'review'), ('neg_seg_idxs', 2, 'seg'), ('neg_user_idxs', 2, 'user'),
                ('neg_item_idxs', 2, 'item'), ('neg_prod_rword_idxs', 2, 'word3'), ('neg_prod_rword_masks', 2, 'mask3'),
                ('neg_prod_rword_idxs_pvc', 2, 'word3'))

    def _same_width(self, batch):
        """The reference pads the positive and the negative sequences of a batch separately (prod_search_dataloader.py:
        283-300), so their review counts can differ; the kernels run every sequence of a step at one width.  The
        shorter side is padded here with the pad ids — masked positions, no effect on the loss or any gradient."""
        rp, rn = batch.pos_prod_ridxs.shape[1], batch.neg_prod_ridxs.shape[2]
        if rp == rn:
            return batch
        pads = dict(review=self.review_pad_idx, seg=self.seg_pad_idx, user=self.user_pad_idx, item=self.prod_pad_idx,
                    word3=self.word_pad_idx, mask3=0)
        grow, by = (self._POS_SEQ, rn - rp) if rp < rn else (self._NEG_SEQ, rp - rn)

        class _Padded(object):
            pass
        out = _Padded()
        out.__dict__.update(batch.__dict__)
        for name, dim, kind in grow:
            t = getattr(batch, name, None)
            if t is None:
                continue
            spec = [0, 0] * (t.dim() - 1 - dim) + [0, by]      # F.pad lists the last dimension first
            setattr(out, name, torch.nn.functional.pad(t, spec, value=pads[kind]))
        return out

    def _run_forward(self, batch, train_pv, neg_word_idxs=None):
        lib = _lib.load()
        ps, _ = self._structs()
        b = self._same_width(batch)
        qw = self._idx(b.query_word_idxs, 'query_word_idxs')
        pr = self._idx(b.pos_prod_ridxs, 'pos_prod_ridxs')
        nr = self._idx(b.neg_prod_ridxs, 'neg_prod_ridxs')
        B, Q = qw.shape
        R, K = pr.shape[1], nr.shape[1]
        if tuple(nr.shape) != (B, K, R):
            raise RuntimeError("neg_prod_ridxs has shape %s, expected %s" % (tuple(nr.shape), (B, K, R)))
        pvc = self.review_encoder_name in ('pvc', 'fs', 'avg')      # the word-mean encoders: review vectors from review words
        if self.review_encoder_name in ('fs', 'avg'):
            train_pv = False                                         # "pv" not in the name: no PV loss (ps_model.py:264)
        pw = self._idx(b.pos_prod_rword_idxs, 'pos_prod_rword_idxs')
        W = pw.shape[2] if train_pv else max(1, self.args.pv_window_size)
        WL = 0
        if pvc:
            src = b.pos_prod_rword_idxs_pvc if train_pv else b.pos_prod_rword_idxs
            if src is None:
                raise RuntimeError("pvc encoder: the batch lacks the review word indices")
            WL = src.shape[2]
        key = (B, K, R, Q, W, WL, bool(train_pv), bool(self.training))
        desc = self._desc(B, K, R, False, 0, Q, W, WL, train_pv)
        plan = self._plan(key, desc, False)
        self._fwd_step += 1
        plan['desc'].step = self._fwd_step
        bt = plan['batch']
        keep = [qw, pr, nr, pw]
        bt.query_word_idxs, bt.pos_prod_ridxs, bt.neg_prod_ridxs = qw.data_ptr(), pr.data_ptr(), nr.data_ptr()
        bt.pos_prod_rword_idxs = pw.data_ptr()
        for name, dtype in (('pos_seg_idxs', torch.int64), ('neg_seg_idxs', torch.int64),
                            ('pos_prod_rword_masks', torch.uint8), ('neg_prod_rword_masks', torch.uint8),
                            ('neg_prod_rword_idxs', torch.int64),
                            ('pos_prod_rword_idxs_pvc', torch.int64), ('neg_prod_rword_idxs_pvc', torch.int64)):
            t = self._idx(getattr(b, name, None), name, dtype)
            lead = {'pos_seg_idxs': (B, R + 1), 'neg_seg_idxs': (B, K, R + 1)}.get(
                name, (B, K, R) if name.startswith('neg') else (B, R))
            if t is not None and tuple(t.shape[:len(lead)]) != lead:
                raise RuntimeError("batch.%s has shape %s, expected %s + word axis" % (name, tuple(t.shape), lead))
            setattr(bt, name, None if t is None else t.data_ptr())
            keep.append(t)
        if train_pv:
            if neg_word_idxs is None:
                neg_word_idxs = self._sample_pv_words(plan, B * R, W, K)
            nw = self._idx(neg_word_idxs, 'neg_word_idxs')
            if nw.numel() != B * R * W * K:
                raise RuntimeError("neg_word_idxs must hold B*R*W*K = %d draws" % (B * R * W * K))
            bt.neg_word_idxs = nw.data_ptr()
            keep.append(nw)
        self._seq_ids(bt, b, ('pos_user_idxs', 'neg_user_idxs', 'pos_item_idxs', 'neg_item_idxs'),
                      dict(pos_user_idxs=(B, R + 1), pos_item_idxs=(B, R + 1),
                           neg_user_idxs=(B, K, R + 1), neg_item_idxs=(B, K, R + 1)), keep)
        plan['keep'] = keep
        loss3 = torch.empty(3, device=self._dev(), dtype=torch.float32)
        _lib.check(lib.ps_rtm_forward(plan['desc'], ps, bt, plan['ws'].data_ptr(), loss3.data_ptr(), self._stream()),
                   'ps_rtm_forward')
        return plan, loss3

    def _run_backward(self, plan, grad_out):
        lib = _lib.load()
        ps, gs = self._structs()
        st = self._stream()
        fresh = self._grad_views[0][0].grad is None
        for p, v in self._grad_views:
            if p.grad is None:
                p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():
                raise RuntimeError("a foreign .grad tensor is attached; call model.zero_grad() before backward")
        if fresh and not self.__dict__.get('_grad_clean', False):     # (clean: the last optimizer step left the buffer at 0)
            _lib.check(lib.ps_zero_floats(self._grad_flat.data_ptr(), self._grad_flat.numel(), st), 'ps_zero_floats')
        self.__dict__['_grad_clean'] = False
        go = None if grad_out is None else grad_out.contiguous().float()          # None: d loss / d loss = 1
        _lib.check(lib.ps_rtm_backward(plan['desc'], ps, plan['batch'], plan['ws'].data_ptr(), gs, 1.0,
                                       None if go is None else go.data_ptr(), st), 'ps_rtm_backward')

    def _run_score(self, batch):
        lib = _lib.load()
        ps, _ = self._structs()
        if self.review_embeddings is None:
            self.get_review_embeddings()
        qw = self._idx(batch.query_word_idxs, 'query_word_idxs')
        cr = self._idx(batch.candi_prod_ridxs, 'candi_prod_ridxs')
        cs = self._idx(batch.candi_seg_idxs, 'candi_seg_idxs')
        B, C, R = cr.shape
        key = ('eval', B, C, R, qw.shape[1])
        desc = self._desc(B, 0, R, True, C, qw.shape[1], 1, int(self.review_words.shape[1]))
        plan = self._plan(key, desc, True)
        bt = plan['batch']
        tab = self.review_embeddings.detach().contiguous()
        bt.query_word_idxs, bt.candi_prod_ridxs, bt.candi_seg_idxs = qw.data_ptr(), cr.data_ptr(), cs.data_ptr()
        bt.review_embeddings = tab.data_ptr()
        keep = [qw, cr, cs, tab]
        self._seq_ids(bt, batch, ('candi_seq_user_idxs', 'candi_seq_item_idxs'),
                      dict(candi_seq_user_idxs=(B, C, R + 1), candi_seq_item_idxs=(B, C, R + 1)), keep)
        plan['keep'] = keep
        scores = torch.empty(B, C, device=self._dev(), dtype=torch.float32)
        _lib.check(lib.ps_rtm_score(plan['desc'], ps, bt, plan['ws'].data_ptr(), scores.data_ptr(), self._stream()),
                   'ps_rtm_score')
        return scores

    # --------------------------------------------------------------- test support
    def workspace_view(self, plan, name, shape):
        """View of one intermediate inside the workspace (parity tests compare every stage; PsRtmWsLayout)."""
        off = getattr(plan['layout'], name)
        n = 1
        for s in shape:
            n *= s
        return plan['ws'][off:off + n].view(*shape)
