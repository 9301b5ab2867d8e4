"""Full-catalogue evaluation on the device: what ``Trainer.test`` / ``validate`` compute with
``test_candi_size < 1`` (trainer.py:125-226) — every product scored for every (user, query) row, top-``cutoff``
ranklist, MRR and P@1 — without re-encoding the sequence per candidate chunk and without moving the score matrix to the
host.  ``rank_all`` = ``ps_tem_encode`` (one encode per row) + ``ps_rank_all`` (fp32 MFMA GEMM per table panel, radix-select
top-k, rank of the target)."""
import numpy as np
import torch

from . import _lib


def _rank_all_sharded(model, batch, topk):
    """``rank_all`` over a row-sharded item table (``args.shard_tables``, sharded.py): a COLLECTIVE — every rank calls it with a
    batch of the same size.  Each rank encodes its own rows (the history rows come from their owners), the encodings, targets and
    target scores are all-gathered, every rank ranks ALL rows against its shard (``ps_rank_shard``: top-k with catalogue ids and
    the number of its rows ahead of each target), the counts are summed and the W top-k lists merged by (score desc, id asc) —
    the order ``ps_rank_all`` uses on one table."""
    import torch.distributed as dist
    lib = _lib.load()
    sh = model._shard
    W, r = sh.world, sh.rank
    if model.args.sim_func == 'bias_product':
        raise NotImplementedError("rank_all over a sharded table with sim_func='bias_product'")
    was_training = model.training
    model.eval()
    try:
        enc = model.encode(batch)
    finally:
        model.train(was_training)
    B, d = enc.shape
    dev = enc.device
    st = torch.cuda.current_stream(dev).cuda_stream
    P = model.product_size
    target = batch.target_prod_idxs.contiguous()
    # the target's score through the SAME product kernel that scores the shards: q . T^T over the fetched target rows, diagonal
    trows = sh.table_buf[model.__dict__['_shard_eval_target_slots']].contiguous()
    tt = torch.empty(B, B, device=dev, dtype=torch.float32)
    _lib.check(lib.ps_gemm_f32(enc.data_ptr(), d, 0, trows.data_ptr(), d, 0, tt.data_ptr(), B, B, B, d, None, 1.0, 0, st),
               'ps_gemm_f32')
    ok = (target >= 0) & (target < P)
    tscore = torch.where(ok, tt.diagonal(), torch.full_like(tt.diagonal(), float('inf'))).contiguous()
    if W > 1:
        encs = torch.empty(W * B, d, device=dev, dtype=torch.float32)
        tgts = torch.empty(W * B, device=dev, dtype=torch.int64)
        tscs = torch.empty(W * B, device=dev, dtype=torch.float32)
        dist.all_gather_into_tensor(encs, enc.contiguous(), group=sh.group)
        dist.all_gather_into_tensor(tgts, target, group=sh.group)
        dist.all_gather_into_tensor(tscs, tscore, group=sh.group)
    else:
        encs, tgts, tscs = enc.contiguous(), target, tscore
    n_mine = (P - r + W - 1) // W                              # catalogue rows i < P with i % W == r
    k = int(topk)
    NB = W * B
    nbytes = lib.ps_rank_scratch_bytes(NB, n_mine, d, k)
    if nbytes < 0:
        raise RuntimeError("rank_all: unsupported sizes (topk must be 1..256)")
    scratch = model.__dict__.get('_rank_scratch')
    if scratch is None or scratch.numel() < nbytes:
        scratch = model.__dict__['_rank_scratch'] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    top_idx = torch.empty(NB, k, dtype=torch.int64, device=dev)
    top_score = torch.empty(NB, k, dtype=torch.float32, device=dev)
    ahead = torch.zeros(NB, dtype=torch.int32, device=dev)
    _lib.check(lib.ps_rank_shard(encs.data_ptr(), NB, d, sh.weight.data_ptr(), n_mine, None, tgts.data_ptr(), tscs.data_ptr(),
                                 W, r, k, top_idx.data_ptr(), top_score.data_ptr(), ahead.data_ptr(), scratch.data_ptr(),
                                 scratch.numel(), st), 'ps_rank_shard')
    if W > 1:
        all_idx = torch.empty(W, NB, k, dtype=torch.int64, device=dev)
        all_sc = torch.empty(W, NB, k, dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(all_idx.view(W * NB, k), top_idx, group=sh.group)
        dist.all_gather_into_tensor(all_sc.view(W * NB, k), top_score, group=sh.group)
        dist.all_reduce(ahead, group=sh.group)
        mine = slice(r * B, (r + 1) * B)
        ci = all_idx[:, mine].permute(1, 0, 2).reshape(B, W * k)
        cs = all_sc[:, mine].permute(1, 0, 2).reshape(B, W * k)
        key = torch.where(ci >= 0, ci, torch.full_like(ci, P + 1))          # unfilled slots last among equal (-inf) scores
        o1 = torch.argsort(key, dim=1, stable=True)
        ci, cs = torch.gather(ci, 1, o1), torch.gather(cs, 1, o1)
        o2 = torch.argsort(cs, dim=1, descending=True, stable=True)
        top_idx, top_score = torch.gather(ci, 1, o2)[:, :k].contiguous(), torch.gather(cs, 1, o2)[:, :k].contiguous()
        ahead = ahead[mine]
    rank = torch.where(ok, 1 + ahead, torch.zeros_like(ahead)).to(torch.int32)
    return top_idx, top_score, rank


def rank_all(model, batch, topk=100):
    """-> (top_idx [B,k] int64, top_score [B,k] fp32, rank [B] int32; 1-based, 0 = target not a product), on device."""
    if getattr(model, '_shard', None) is not None:
        return _rank_all_sharded(model, batch, topk)
    lib = _lib.load()
    was_training = model.training
    model.eval()
    try:
        enc = model.encode(batch)
    finally:
        model.train(was_training)
    B, d = enc.shape
    P = model.product_size
    table = model.product_emb.weight                       # rows 0..P-1; the pad row P is not a candidate
    bias = model.product_bias if model.args.sim_func == 'bias_product' else None
    target = batch.target_prod_idxs
    if not (torch.is_tensor(target) and target.is_cuda and target.dtype == torch.int64):
        raise RuntimeError("batch.target_prod_idxs must be an int64 tensor on the model's device")
    k = int(topk)
    nbytes = lib.ps_rank_scratch_bytes(B, P, d, k)
    if nbytes < 0:
        raise RuntimeError("rank_all: unsupported sizes (topk must be 1..256)")
    scratch = getattr(model, '_rank_scratch', None)
    if scratch is None or scratch.numel() < nbytes:
        scratch = model.__dict__['_rank_scratch'] = torch.empty(nbytes, dtype=torch.uint8, device=enc.device)
    top_idx = torch.empty(B, k, dtype=torch.int64, device=enc.device)
    top_score = torch.empty(B, k, dtype=torch.float32, device=enc.device)
    rank = torch.empty(B, dtype=torch.int32, device=enc.device)
    _lib.check(lib.ps_rank_all(enc.data_ptr(), B, d, table.data_ptr(), P, _lib.ptr(bias), target.contiguous().data_ptr(), k,
                               top_idx.data_ptr(), top_score.data_ptr(), rank.data_ptr(), scratch.data_ptr(),
                               scratch.numel(), torch.cuda.current_stream(enc.device).cuda_stream), 'ps_rank_all')
    return top_idx, top_score, rank


def calc_metrics(rank, cutoff=100):
    """``Trainer.calc_metrics`` (trainer.py:172-187) from the 1-based ranks of ``rank_all``."""
    r = torch.as_tensor(rank).detach().cpu().numpy().astype(np.int64)
    hit = (r > 0) & ((cutoff < 0) | (r <= cutoff))
    mrr = float(np.where(hit, 1.0 / np.maximum(r, 1), 0.0).sum() / len(r))
    return mrr, float((r == 1).sum() / len(r))


def ranklist_lines(user_ids, query_idxs, product_ids, top_idx, top_score, tag="ReviewTransformer"):
    """Lines of the TREC-style ranklist ``Trainer.test`` writes (trainer.py:160-170)."""
    ti = torch.as_tensor(top_idx).cpu().numpy()
    ts = torch.as_tensor(top_score).cpu().numpy()
    for i in range(ti.shape[0]):
        for r in range(ti.shape[1]):
            if ti[i, r] < 0:
                break
            yield "%s_%d Q0 %s %d %f %s\n" % (user_ids[i], query_idxs[i], product_ids[int(ti[i, r])], r + 1,
                                              float(ts[i, r]), tag)


def evaluate(model, batches, topk=100, cutoff=100):
    """MRR / P@1 over an iterable of eval batches (``Trainer.test`` without the host-side argsort)."""
    ranks = [rank_all(model, b, topk)[2] for b in batches]
    return calc_metrics(torch.cat(ranks), cutoff)
