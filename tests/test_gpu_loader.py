"""Loader-fed training: the native producer thread + copy stream hand the device the same batches as the
sequential host loader, in the same order, with the step's stream reading them while later batches are being built and copied."""
import pytest
import torch

from prodsearch_amd import default_args, synth
from prodsearch_amd.dataloader import ItemPVDataloader

pytestmark = pytest.mark.gpu


def test_prefetched_device_batches_equal_the_sequential_host_batches():
    train_ds, _ = synth.make_corpus(41, n_users=300, n_products=200, n_queries=40, vocab_size=500, Q=5, W=2, max_reviews_per_user=80)
    args = default_args(uprev_review_limit=12, fix_train_review=False, pv_window_size=2)
    B = 64
    assert len(train_ds) % B != 0 and len(train_ds) // B >= 20
    host = ItemPVDataloader(args, train_ds, batch_size=B, shuffle=True, seed=3)
    dev = ItemPVDataloader(args, train_ds, batch_size=B, shuffle=True, seed=3, device='cuda', prefetch=3)
    torch.manual_seed(5)
    want = list(host)
    torch.manual_seed(5)
    busy = torch.zeros(1 << 22, device='cuda')
    got = []
    for b in dev:
        busy.add_(1.0)                                   # the consumer's stream has work queued in front of every batch
        got.append((b, b.u_item_idxs.sum()))             # ... and reads the batch on that stream
    assert len(got) == len(want)
    torch.cuda.synchronize()
    for w, (g, s) in zip(want, got):
        for k in ('query_word_idxs', 'target_prod_idxs', 'u_item_idxs', 'pos_iword_idxs'):
            t = getattr(g, k)
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.int64
            assert torch.equal(t.cpu(), getattr(w, k)), k
        assert int(s) == int(w.u_item_idxs.sum())
        assert list(w.query_idxs) == list(g.query_idxs) and list(w.user_idxs) == list(g.user_idxs)
