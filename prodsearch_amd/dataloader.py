"""ItemPVDataloader — the reference's TEM loader with the collate moved into C++.

Mirror of ``data/item_pv_dataloader.py:ItemPVDataloader`` (constructor ``(args, dataset, batch_size, shuffle)``,
``get_train_batch`` / ``get_test_batch`` collates, iteration yielding ``ItemPVBatch``) over a dataset object shaped
like ``data/item_pv_dataset.py:ItemPVDataset`` (``global_data``, ``prod_data``, ``_data``, pad ids).  The Python
structures are flattened ONCE into arrays (``Corpus``); each batch is then built by ``ps_collate_train`` /
``ps_collate_test`` (include/prodsearch_data.h) into pinned host buffers and shipped with one asynchronous copy
per tensor.  Sampling order: torch's own ``RandomSampler`` / ``BatchSampler`` (what ``DataLoader(shuffle=True)``
uses), including the base-seed draw a ``DataLoader`` iterator makes first, so ``torch.manual_seed`` reproduces the
reference's batch composition; query choice / history subsampling: CPython-compatible Mersenne Twister seeded
like ``random.seed`` (``seed=`` argument), bit-identical to the reference with ``num_workers=0``.
"""
import ctypes as C

import numpy as np
import torch
from torch.utils.data import BatchSampler, RandomSampler, SequentialSampler

from . import _lib, pyrandom
from .batch import ItemPVBatch


def _csr(lists):
    ptr = np.zeros(len(lists) + 1, dtype=np.int64)
    ptr[1:] = np.cumsum([len(x) for x in lists])
    flat = np.fromiter((v for x in lists for v in x), dtype=np.int64, count=int(ptr[-1]))
    return ptr, flat


def sampler_order(n, shuffle):
    """Row ids of one epoch in the order ``DataLoader(shuffle=..., num_workers=0)`` visits them: a DataLoader
    iterator draws its base seed first, then RandomSampler seeds a private generator from the default one and
    takes ONE ``torch.randperm`` (torch/utils/data/sampler.py) — reproduced here without the per-index Python
    generator chain."""
    torch.empty((), dtype=torch.int64).random_()
    if shuffle:
        g = torch.Generator()
        g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
        return torch.randperm(n, generator=g).numpy()
    return np.arange(n, dtype=np.int64)


def sampler_batches(n, batch_size, shuffle, drop_last=False):
    """``sampler_order`` cut into batches (``BatchSampler``)."""
    order = sampler_order(n, shuffle)
    stop = n - n % batch_size if drop_last else n
    for i in range(0, stop, batch_size):
        yield order[i:i + batch_size]


def _device_tensors(obj, depth=0):
    if torch.is_tensor(obj):
        if obj.is_cuda:
            yield obj
    elif depth < 2:
        vals = obj.values() if isinstance(obj, dict) else (vars(obj).values() if hasattr(obj, '__dict__') else ())
        for v in vals:
            for t in _device_tensors(v, depth + 1):
                yield t


def prefetch_iter(make_batches, depth, device):
    """Run ``make_batches()`` in ONE producer thread (the RNG order stays the sequential one; the C collate runs
    without the GIL) on its own HIP stream, ``depth`` batches ahead.  The consumer's stream waits for the batch's
    copies and every device tensor of the batch is recorded on it, so the caching allocator cannot hand the memory
    back to the producer while kernels already queued on the consumer's stream still read it."""
    import queue
    import threading
    q = queue.Queue(maxsize=depth)
    dev = torch.cuda.current_device() if (device is not None and torch.cuda.is_available()) else None
    stream = torch.cuda.Stream() if dev is not None else None

    def work():
        try:
            if stream is not None:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(stream):
                    for b in make_batches():
                        ev = torch.cuda.Event()
                        ev.record(stream)
                        q.put((b, ev))
            else:
                for b in make_batches():
                    q.put((b, None))
            q.put(None)
        except BaseException as e:       # surface producer errors in the consumer
            q.put(e)

    threading.Thread(target=work, daemon=True).start()
    while True:
        item = q.get()
        if item is None:
            return
        if isinstance(item, BaseException):
            raise item
        b, ev = item
        if ev is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            for t in _device_tensors(b):
                t.record_stream(cur)
        yield b


class Corpus(object):
    """Flat-array view of ``global_data`` / ``prod_data`` (host memory, borrowed by the C calls)."""

    def __init__(self, global_data, prod_data):
        gd, pd = global_data, prod_data
        self.review_u_p = np.ascontiguousarray(np.asarray(gd.review_u_p, dtype=np.int64).reshape(-1, 2))
        n_rev = self.review_u_p.shape[0]
        self.u_seq_ptr, self.u_seq = _csr(gd.u_r_seq)
        self.n_users = len(gd.u_r_seq)
        self.train_review = np.zeros(n_rev, dtype=np.uint8)
        u_reviews = pd.u_reviews
        it = u_reviews.items() if hasattr(u_reviews, 'items') else enumerate(u_reviews)
        for u, revs in it:
            for r in revs:
                if self.review_u_p[r, 0] == u:       # ``x in u_train_review_set`` for x in that user's sequence
                    self.train_review[r] = 1
        self.review_loc = np.asarray([t[0] for t in gd.review_loc_time], dtype=np.int64)
        self.pq_ptr, self.pq_idx = _csr(pd.product_query_idx)
        self.query_words = np.ascontiguousarray(np.asarray(gd.query_words, dtype=np.int64))
        self.n_products = int(gd.product_size)
        self.view = _lib.PsCorpusView()
        v = self.view
        v.n_reviews, v.n_users, v.n_products, v.n_queries = n_rev, self.n_users, self.n_products, self.query_words.shape[0]
        for name in ('review_u_p', 'u_seq_ptr', 'u_seq', 'train_review', 'review_loc', 'pq_ptr', 'pq_idx', 'query_words'):
            setattr(v, name, getattr(self, name).ctypes.data)
        v.Q = self.query_words.shape[1]


class ItemPVDataloader(object):
    def __init__(self, args, dataset, prepare_pv=True, batch_size=1, shuffle=False, drop_last=False,
                 seed=None, device=None, pin_memory=True, prefetch=0, **_ignored):
        self.args = args
        self.dataset = dataset
        self.batch_size = batch_size
        self.prod_pad_idx = dataset.prod_pad_idx
        self.word_pad_idx = dataset.word_pad_idx
        self.seg_pad_idx = dataset.seg_pad_idx
        self.global_data = dataset.global_data
        self.prod_data = dataset.prod_data
        self.device = device
        self._lib = _lib.load_data()
        self.corpus = Corpus(self.global_data, self.prod_data)
        self._train = self.prod_data.set_name == 'train'
        self.shuffle, self.drop_last = bool(shuffle), bool(drop_last)
        self.prefetch = int(prefetch)
        sampler = RandomSampler(dataset) if shuffle else SequentialSampler(dataset)
        self.batch_sampler = BatchSampler(sampler, batch_size, drop_last)     # len() and the reference's semantics
        # seed=None: the process-wide generator (pyrandom.seed(s) == the reference's random.seed(s)), shared with
        # the dataset's epoch shuffles; seed=int: a private stream
        self._own_rng = self._lib.ps_rng_create(int(seed)) if seed is not None else None
        self._pin = bool(pin_memory) and torch.cuda.is_available()
        # staging buffers rotate over 3 slots; a slot is reused only after the copies issued from it have finished
        self._slots = [dict() for _ in range(3)]
        self._events = [None] * 3
        self._slot = 0
        self._flatten_dataset()

    def __del__(self):
        try:
            if self._own_rng is not None:
                self._lib.ps_rng_destroy(self._own_rng)
        except Exception:
            pass

    @property
    def _rng(self):
        return self._own_rng if self._own_rng is not None else pyrandom.handle()

    def reseed(self, seed):
        """``random.seed(seed)`` for the query / history draws."""
        if self._own_rng is None:
            pyrandom.seed(seed)
        else:
            self._lib.ps_rng_seed(self._own_rng, int(seed))

    def __len__(self):
        return len(self.batch_sampler)

    # ------------------------------------------------------------------ dataset -> arrays (once per epoch)
    def _flatten_dataset(self):
        data = self.dataset._data
        if self._train and hasattr(self.dataset, 'sample_words'):       # prodsearch_amd.corpus.ItemPVDataset: already flat
            self.sample_words, self.sample_review = self.dataset.sample_words, self.dataset.sample_review
            self._index_of = None
        elif self._train:
            self.sample_words = np.ascontiguousarray(np.asarray([e[0] for e in data], dtype=np.int64).reshape(len(data), -1))
            self.sample_review = np.asarray([e[1] for e in data], dtype=np.int64)
            self._index_of = None
        else:
            self.entry_quad = np.ascontiguousarray(np.asarray([e[:4] for e in data], dtype=np.int64).reshape(len(data), 4))
            self.candi_ptr, self.candi_items = _csr([e[4] if e[4] is not None else () for e in data])

    def _next_slot(self):
        self._slot = (self._slot + 1) % len(self._slots)
        ev = self._events[self._slot]
        if ev is not None:
            ev.synchronize()
            self._events[self._slot] = None

    def _shipped(self):
        if self.device is not None and torch.cuda.is_available():
            ev = torch.cuda.Event()
            ev.record()
            self._events[self._slot] = ev

    def _buf(self, name, shape):
        if self.device is None:                  # host consumer: a fresh tensor per batch, written by the C call
            return torch.empty(shape, dtype=torch.int64)
        bufs = self._slots[self._slot]
        hit = bufs.get(name)
        if hit is not None and hit[0] == shape:
            return hit[1]
        n = 1
        for x in shape:
            n *= x
        t = torch.empty(max(n, 1), dtype=torch.int64, pin_memory=self._pin)[:n].view(*shape)
        bufs[name] = (shape, t)
        return t

    def _args(self, do_seq, fix):
        a = _lib.PsCollateArgs()
        a.uprev_review_limit, a.do_seq, a.fix = int(self.args.uprev_review_limit), int(bool(do_seq)), int(bool(fix))
        a.prod_pad = self.prod_pad_idx
        return a

    def _ship(self, t):
        # the staging buffers are reused by the next batch: clone on the host, or copy to the device on the current
        # stream (pinned => asynchronous; ordered before the buffers' next overwrite by the caller's sync cadence)
        if self.device is None:
            return t
        return t.to(self.device, non_blocking=True)

    # ------------------------------------------------------------------ collates
    def train_batch_from_ids(self, ids, sample_words=None, sample_review=None):
        """``get_train_batch`` (item_pv_dataloader.py:121-143) for dataset rows ``ids``."""
        ids = np.ascontiguousarray(np.asarray(ids, dtype=np.int64))
        self._next_slot()
        sample_words = self.sample_words if sample_words is None else sample_words
        sample_review = self.sample_review if sample_review is None else sample_review
        B, W, Q = len(ids), sample_words.shape[1], self.corpus.view.Q
        lim = int(self.args.uprev_review_limit)
        qw, tg = self._buf('qw', (B, Q)), self._buf('tg', (B,))
        ui, pw = self._buf('ui', (B, lim)), self._buf('pw', (B, W))
        qi, us = self._buf('qi', (B,)), self._buf('us', (B,))
        hl = np.zeros(B, dtype=np.int32)
        lmax = C.c_int32(0)
        a = self._args(self.args.do_seq_review_train, self.args.fix_train_review)
        _lib.check_data(self._lib.ps_collate_train(
            self.corpus.view, a, self._rng, sample_words.ctypes.data, sample_review.ctypes.data,
            len(sample_review), W, ids.ctypes.data, B, qw.data_ptr(), tg.data_ptr(), ui.data_ptr(), pw.data_ptr(),
            qi.data_ptr(), us.data_ptr(), hl.ctypes.data, C.addressof(lmax)), 'ps_collate_train')
        L = lmax.value                                   # util.pad: width of the longest history in the batch
        out = ItemPVBatch(self._ship(qw), self._ship(tg), self._ship(ui[:, :L]).contiguous(), self._ship(pw),
                          query_idxs=qi.numpy().copy(), user_idxs=us.numpy().copy(), to_tensor=False)
        self._shipped()
        return out

    def test_batch_from_ids(self, ids):
        """``get_test_batch`` (item_pv_dataloader.py:32-50)."""
        ids = np.asarray(ids, dtype=np.int64)
        self._next_slot()
        B, Q = len(ids), self.corpus.view.Q
        lim = int(self.args.uprev_review_limit)
        quad = np.ascontiguousarray(self.entry_quad[ids])
        lens = self.candi_ptr[ids + 1] - self.candi_ptr[ids]
        cptr = np.zeros(B + 1, dtype=np.int64)
        cptr[1:] = np.cumsum(lens)
        citems = np.concatenate([self.candi_items[self.candi_ptr[i]:self.candi_ptr[i + 1]] for i in ids]) \
            if B else np.zeros(0, np.int64)
        citems = np.ascontiguousarray(citems)
        width = int(lens.max())                          # 0: full-catalogue entries (evaluate.rank_all needs no list)
        qw, tg = self._buf('qw', (B, Q)), self._buf('tg', (B,))
        ui, ca = self._buf('ui', (B, lim)), self._buf('ca', (B, max(width, 1)))
        hl = np.zeros(B, dtype=np.int32)
        lmax = C.c_int32(0)
        do_seq = getattr(self.args, 'do_seq_review_test', False) and not self.args.train_review_only
        a = self._args(do_seq, True)
        _lib.check_data(self._lib.ps_collate_test(
            self.corpus.view, a, quad.ctypes.data, B, cptr.ctypes.data, citems.ctypes.data, width,
            qw.data_ptr(), tg.data_ptr(), ui.data_ptr(), ca.data_ptr(), hl.ctypes.data, C.addressof(lmax)),
            'ps_collate_test')
        L = lmax.value
        out = ItemPVBatch(self._ship(qw), self._ship(tg), self._ship(ui[:, :L]).contiguous(),
                          torch.zeros(0, dtype=torch.int64), query_idxs=quad[:, 0].tolist(),
                          user_idxs=quad[:, 1].tolist(),
                          candi_prod_idxs=self._ship(ca if width else ca[:, :0].contiguous()), to_tensor=False)
        self._shipped()
        return out

    def _ids_of(self, batch):
        """Entries as the reference's collate receives them -> dataset row ids (identity lookup)."""
        if self._index_of is None:
            self._index_of = {id(e): i for i, e in enumerate(self.dataset._data)}
        return [self._index_of[id(e)] for e in batch]

    def get_train_batch(self, batch):
        """The reference's collate_fn signature: ``batch`` = dataset entries ``[word ids, review id]``."""
        sw = np.ascontiguousarray(np.asarray([e[0] for e in batch], dtype=np.int64).reshape(len(batch), -1))
        sr = np.asarray([e[1] for e in batch], dtype=np.int64)
        return self.train_batch_from_ids(np.arange(len(batch)), sw, sr)

    def get_test_batch(self, batch):
        if getattr(self, '_index_of', None) is None:
            self._index_of = {id(e): i for i, e in enumerate(self.dataset._data)}
        return self.test_batch_from_ids([self._index_of[id(e)] for e in batch])

    def _batch_ids(self):
        return sampler_batches(len(self.dataset), self.batch_size, self.shuffle, self.drop_last)

    def _batches(self):
        for ids in self._batch_ids():
            yield self.train_batch_from_ids(ids) if self._train else self.test_batch_from_ids(ids)

    def __iter__(self):
        if self.prefetch <= 0:
            return self._batches()
        return self._prefetched()

    def _prefetched(self):
        if self._train:
            return self._native_epoch()
        return prefetch_iter(self._batches, self.prefetch, self.device)

    def _native_epoch(self):
        """The epoch's train batches from the NATIVE producer thread (``ps_epoch_start``, include/prodsearch_data.h): the collate
        of batch k + 1 .. k + prefetch runs on its own core, without the interpreter lock, while the consumer ships batch k and
        launches its step.  (A Python producer thread made the fed step slower than no prefetch at all: 0.338 against 0.320 ms
        on a 0.236 ms GPU step — its tensor bookkeeping takes the lock the step's launch code needs.)  Same batches, same
        generator order as the sequential loop; a consumer that stops early leaves the generator up to ``prefetch + 1`` batches
        further than the sequential loop would."""
        import collections
        import weakref
        lib = self._lib
        # The producer thread draws from self._rng — with seed=None the process-wide generator the dataset's epoch shuffles use
        # too — and that generator has no lock: ONE live epoch per loader, and nothing else may draw from the generator while it
        # runs (a private fork would change the batches: they are the sequential loop's, draw for draw).  ADVICE r4.
        box = self.__dict__.setdefault('_live_epoch', [None])
        if box[0] is not None:
            raise RuntimeError("ItemPVDataloader: an epoch iterator of this loader is still alive (its producer thread owns the "
                               "random generator): exhaust or close() it before starting another")
        depth = min(max(self.prefetch + 2, 4), 64)
        order = np.ascontiguousarray(sampler_order(len(self.dataset), self.shuffle), dtype=np.int64)
        B, W, Q = int(self.batch_size), self.sample_words.shape[1], self.corpus.view.Q
        lim = int(self.args.uprev_review_limit)
        on_dev = self.device is not None and torch.cuda.is_available()
        pin = self._pin and on_dev
        # one slot = ONE pinned buffer [qw | tg | ui | pw] (a single H2D copy per batch) + the host-side per-row ids
        o_qw, o_tg, o_ui, o_pw, n_flat = 0, B * Q, B * Q + B, B * Q + B + B * lim, B * Q + B + B * lim + B * W
        ring = [dict(flat=torch.empty(n_flat, dtype=torch.int64, pin_memory=pin), qi=torch.empty(B, dtype=torch.int64),
                     us=torch.empty(B, dtype=torch.int64), hl=torch.empty(B, dtype=torch.int32)) for _ in range(depth)]
        slots = (_lib.PsTrainSlot * depth)()
        for sl, r in zip(slots, ring):
            base = r['flat'].data_ptr()
            sl.query_words, sl.target, sl.u_items, sl.pos_words = base + 8 * o_qw, base + 8 * o_tg, base + 8 * o_ui, base + 8 * o_pw
            sl.query_idx, sl.user_idx, sl.hist_len = r['qi'].data_ptr(), r['us'].data_ptr(), r['hl'].data_ptr()
        a = self._args(self.args.do_seq_review_train, self.args.fix_train_review)
        sw, sr = self.sample_words, self.sample_review          # kept alive by this frame, like `order`, `ring`, `a`
        h = lib.ps_epoch_start(self.corpus.view, a, self._rng, sw.ctypes.data, sr.ctypes.data, len(sr), W, order.ctypes.data,
                               len(order), B, int(self.drop_last), slots, depth)
        if not h:
            raise RuntimeError("ps_epoch_start failed: %s" % lib.ps_data_last_error().decode('utf-8', 'replace'))
        box[0] = h

        def _stop(box=box, lib=lib, keep=(ring, order, sw, sr, slots, a)):      # the thread writes into `ring`: stop it before
            if box[0] is not None:                                               # anything it touches can be freed
                lib.ps_epoch_stop(box[0])
                box[0] = None
        fin = weakref.finalize(self, _stop)      # also runs at interpreter exit, where a generator's `finally` may not

        def views(flat, n, L):
            ui = flat[o_ui:o_ui + B * lim].view(B, lim)[:n]
            if L < lim:
                ui = ui[:, :L].contiguous()                      # util.pad: the width of the batch's longest history
            return (flat[o_qw:o_qw + B * Q].view(B, Q)[:n], flat[o_tg:o_tg + B][:n], ui, flat[o_pw:o_pw + B * W].view(B, W)[:n])

        # The copies go on their OWN stream, one batch ahead of the step that uses them: issued on the step's stream the four
        # copies of a batch queued behind the previous step's kernels and added ~40 us to every step (0.236 -> 0.28 ms).
        copy_stream = torch.cuda.Stream() if on_dev else None
        held = collections.deque()                              # (slot, event of its copy) in shipping order
        shipped = collections.deque()                           # batches whose copy is under way, oldest first
        nb, lmax = C.c_int32(0), C.c_int32(0)
        done = False
        try:
            while True:
                while not done and len(shipped) < (2 if on_dev else 1):
                    while held and (len(held) >= depth - 2 or held[0][1].query()):
                        s, ev = held.popleft()
                        ev.synchronize()
                        _lib.check_data(lib.ps_epoch_release(h, s), 'ps_epoch_release')
                    s = lib.ps_epoch_next(h, C.byref(nb), C.byref(lmax))
                    if s == -1:
                        done = True
                        break
                    if s < 0:
                        raise RuntimeError("ps_epoch_next failed: %s" % lib.ps_data_last_error().decode('utf-8', 'replace'))
                    r, n, L = ring[s], nb.value, lmax.value
                    ids = (r['qi'][:n].numpy().copy(), r['us'][:n].numpy().copy())
                    if on_dev:
                        with torch.cuda.stream(copy_stream):
                            flat = r['flat'].to(self.device, non_blocking=True)
                            ev = torch.cuda.Event()
                            ev.record(copy_stream)
                        held.append((s, ev))
                        shipped.append((flat, ev, n, L, ids))
                    else:
                        shipped.append((r['flat'].clone(), None, n, L, ids))
                        _lib.check_data(lib.ps_epoch_release(h, s), 'ps_epoch_release')
                if not shipped:
                    return
                flat, ev, n, L, ids = shipped.popleft()
                if ev is not None:
                    cur = torch.cuda.current_stream()
                    cur.wait_event(ev)
                    flat.record_stream(cur)                      # allocated on the copy stream, read by the step's kernels
                qw, tg, ui, pw = views(flat, n, L)
                yield ItemPVBatch(qw, tg, ui, pw, query_idxs=ids[0], user_idxs=ids[1], to_tensor=False)
        finally:
            for s, ev in held:
                ev.synchronize()
            fin()                                 # ps_epoch_stop, once
