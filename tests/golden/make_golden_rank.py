#!/usr/bin/env python3
"""Golden full-catalogue ranking of the REFERENCE (Trainer.test semantics, trainer.py:125-226, test_candi_size < 1).

Run here (needs /root/reference):  python tests/golden/make_golden_rank.py
For a few TEM/QEM golden cases (same weights and batch as tests/golden/<case>.npz) the reference model scores ALL
products in chunks of candi_batch_size like ``get_prod_scores`` does (``model.test`` per chunk, pad P in the ragged
tail), then ``argsort(axis=-1)[:, ::-1]`` and ``Trainer.calc_metrics`` — stored: full score matrix, top-100 ids and
scores, the target's rank, MRR and P@1.  Data only."""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
sys.path.insert(0, '/root/reference')

import numpy as np  # noqa: E402
import torch  # noqa: E402

_orig_mf = torch.Tensor.masked_fill
torch.Tensor.masked_fill = lambda self, mask, value: _orig_mf(self, mask.bool() if mask.dtype == torch.uint8 else mask, value)

from golden_util import Golden  # noqa: E402
from models.item_transformer import ItemTransformerRanker  # noqa: E402  (the reference)
from data.batch_data import ItemPVBatch as RefBatch  # noqa: E402
from trainer import Trainer  # noqa: E402

CASES = ['tem_c1', 'tem_c2s', 'qem_c1', 'tem_opts', 'tem_l2']
CHUNK = 500          # --candi_batch_size default (main.py)


def main():
    for case in CASES:
        g = Golden(case)
        model = ItemTransformerRanker(g.args, 'cpu', g.V, g.P, None, word_dists=g.word_dists)
        model.load_state_dict(g.params(), strict=False)
        model.eval()
        b = g.batch()
        B, P = g.B, g.P
        seg = (P - 1) // CHUNK + 1
        scores, idxs = [], []
        with torch.no_grad():
            for s in range(seg):                                   # item_pv_dataset.py:62-65 + util.pad with P
                ids = list(range(s * CHUNK, min(P, (s + 1) * CHUNK)))
                ids = ids + [P] * (CHUNK - len(ids))
                cand = torch.tensor([ids] * B)
                rb = RefBatch(b.query_word_idxs, b.target_prod_idxs, b.u_item_idxs, [], b.query_idxs, b.user_idxs,
                              cand, to_tensor=False)
                scores.append(model.test(rb).numpy())
                idxs.append(cand.numpy())
        all_scores = np.concatenate(scores, axis=1)[:, :P]          # trainer.py:215-216
        all_idxs = np.concatenate(idxs, axis=1)[:, :P]
        sorted_idx = all_scores.argsort(axis=-1)[:, ::-1]           # trainer.py:137
        target = b.target_prod_idxs.numpy()
        mrr, prec = Trainer.calc_metrics(None, all_idxs, sorted_idx, target, P, cutoff=100)
        top = np.take_along_axis(all_idxs, sorted_idx[:, :100], axis=1)
        top_s = np.take_along_axis(all_scores, sorted_idx[:, :100], axis=1)
        rank = np.array([int(np.where(all_idxs[i][sorted_idx[i]] == target[i])[0][0]) + 1 for i in range(B)])
        np.savez_compressed(os.path.join(HERE, 'rank_%s.npz' % case), scores=all_scores.astype(np.float32),
                            top_idx=top.astype(np.int64), top_score=top_s.astype(np.float32), rank=rank.astype(np.int32),
                            mrr=np.float64(mrr), p1=np.float64(prec))
        print(case, 'P', P, 'mrr', mrr, 'p@1', prec, 'rank range', rank.min(), rank.max())


if __name__ == '__main__':
    main()
