"""Batch container of the TEM hot path.

Field-for-field the reference's ``ItemPVBatch`` (``data/batch_data.py:3-37``): a
bag of int64 index tensors, row-major contiguous, with ``.to(device)`` returning
a new object.  The reference's own batch objects are accepted unchanged by the
model (duck typing); this class exists so the hot path can be driven without
the reference's data loaders (out of scope).

    query_word_idxs  [B, Q]  pad = vocab_size-1
    target_prod_idxs [B]
    u_item_idxs      [B, L]  pad = product_size
    pos_iword_idxs   [B, W]  pad = vocab_size-1
    candi_prod_idxs  [B, C]  (eval only) pad = product_size
"""
import torch


class ItemPVBatch(object):
    def __init__(self, query_word_idxs, target_prod_idxs, u_item_idxs,
                 pos_iword_idxs=(), query_idxs=(), user_idxs=(),
                 candi_prod_idxs=(), to_tensor=True):
        self.query_word_idxs = query_word_idxs
        self.target_prod_idxs = target_prod_idxs
        self.u_item_idxs = u_item_idxs
        self.pos_iword_idxs = pos_iword_idxs
        self.query_idxs = query_idxs
        self.user_idxs = user_idxs
        self.candi_prod_idxs = candi_prod_idxs
        if to_tensor:
            self.to_tensor()

    def to_tensor(self):
        def t(x):
            return x if torch.is_tensor(x) else torch.as_tensor(x, dtype=torch.int64)
        self.query_word_idxs = t(self.query_word_idxs)
        self.target_prod_idxs = t(self.target_prod_idxs)
        self.candi_prod_idxs = t(self.candi_prod_idxs)
        self.u_item_idxs = t(self.u_item_idxs)
        self.pos_iword_idxs = t(self.pos_iword_idxs)

    def to(self, device):
        if device == "cpu":
            return self
        mv = lambda x: x.to(device, non_blocking=True)
        return self.__class__(
            mv(self.query_word_idxs), mv(self.target_prod_idxs), mv(self.u_item_idxs),
            mv(self.pos_iword_idxs), self.query_idxs, self.user_idxs,
            mv(self.candi_prod_idxs), to_tensor=False)

    def pin(self):
        """Pinned host copies so ``.to('cuda')`` is an async DMA (the reference copies
        from pageable memory, trainer.py:70)."""
        pn = lambda x: x.pin_memory() if x.numel() else x
        return self.__class__(
            pn(self.query_word_idxs), pn(self.target_prod_idxs), pn(self.u_item_idxs),
            pn(self.pos_iword_idxs), self.query_idxs, self.user_idxs,
            pn(self.candi_prod_idxs), to_tensor=False)
