"""Oracle (TEST INFRASTRUCTURE): full-catalogue evaluation restated (trainer.py:125-226 with test_candi_size < 1).

``get_prod_scores`` (:189-226) concatenates ``model.test`` over candidate chunks; ``test``/``validate`` take
``argsort(axis=-1)[:, ::-1]`` (:137, :155) and ``calc_metrics`` (:172-187) finds the target's position.  numpy's
argsort leaves the order of exactly equal scores unspecified; the product (and this restatement) break such ties by
the lower product id.  Pinned against tests/golden/rank_*.npz (the reference itself, make_golden_rank.py)."""
import numpy as np


def rank_scores(scores, target, topk=100):
    """scores [B,P] (column p = product p) -> top ids [B,k], top scores [B,k], rank of target [B] (1-based)."""
    scores = np.asarray(scores, dtype=np.float32)
    B, P = scores.shape
    ids = np.arange(P)
    k = min(topk, P)
    top = np.zeros((B, k), dtype=np.int64)
    rank = np.zeros(B, dtype=np.int32)
    for b in range(B):
        order = np.lexsort((ids, -scores[b].astype(np.float64)))      # score desc, id asc
        top[b] = order[:k]
        rank[b] = int(np.where(order == target[b])[0][0]) + 1 if 0 <= target[b] < P else 0
    return top, np.take_along_axis(scores, top, axis=1), rank


def calc_metrics(rank, cutoff=100):
    """``Trainer.calc_metrics`` (trainer.py:172-187) from 1-based ranks (0 = target absent)."""
    rank = np.asarray(rank)
    found = rank > 0
    rr = np.where(found & ((cutoff < 0) | (rank <= cutoff)), 1.0 / np.maximum(rank, 1), 0.0)
    return float(rr.sum() / len(rank)), float((rank == 1).sum() / len(rank))
