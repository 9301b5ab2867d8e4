"""Trainer — the reference's training / validation / test driver, TEM and review-transformer.

Mirror of ``trainer.py:17-227`` and ``main.py:141-191`` (``create_model``, ``train``) with the same call sequence per
step — ``loss = model(batch); model.zero_grad(); loss.backward(); optim.step()`` (``trainer.py:74-78``) — the same
checkpoint dictionary (``epoch / model / opt / optim``, ``trainer.py:112-122``), the same best-model selection by
validation MRR and the same TREC-style ranklist (``trainer.py:160-170``).  MI355X-first differences:

* batches come from the native loader (``prodsearch_amd.dataloader``) already on the device, prefetched by a thread;
* the loss is accumulated on the device and read every ``steps_per_checkpoint`` steps, not synchronised every step
  (the reference calls ``.item()`` three times per step);
* evaluation over all products (``*_candi_size < 1``) is ``evaluate.rank_all`` (one encode per row, scores and top-k
  on the device); sampled-candidate validation goes through ``model.test`` like the reference.
"""
import logging
import os
import shutil
import time

import numpy as np
import torch

from . import corpus, evaluate, pyrandom
from .dataloader import ItemPVDataloader
from .item_transformer import ItemTransformerRanker
from .optimizers import build_optim
from .ps_model import ProductRanker
from .rtm_loader import ProdSearchDataLoader

logger = logging.getLogger('prodsearch_amd')


def create_model(args, global_data, prod_data, load_path=''):
    """``create_model`` (main.py:141-165)."""
    if args.model_name == 'review_transformer':
        model = ProductRanker(args, args.device, global_data.vocab_size, global_data.review_count,
                              global_data.product_size, global_data.user_size, global_data.review_words,
                              global_data.words, word_dists=prod_data.word_dists)
    elif args.model_name in ('item_transformer', 'QEM'):
        model = ItemTransformerRanker(args, args.device, global_data.vocab_size, global_data.product_size,
                                      global_data.words, word_dists=prod_data.word_dists)
    else:
        raise NotImplementedError("trainer: model_name %r is not built (SURVEY.md §8)" % args.model_name)
    if load_path and os.path.exists(load_path):
        logger.info('Loading checkpoint from %s', load_path)
        ckpt = torch.load(load_path, map_location='cpu', weights_only=False)
        args.start_epoch = ckpt['epoch']
        model.load_cp(ckpt)
        optim = build_optim(args, model, ckpt if getattr(args, 'train_from', '') else None)
    else:
        logger.info('No available model to load. Build new model.')
        optim = build_optim(args, model, None)
    return model, optim


class Trainer(object):
    def __init__(self, args, model, optim):
        self.args = args
        self.model = model
        self.optim = optim
        if model is not None:
            logger.info('* number of parameters: %d', sum(p.nelement() for p in model.parameters()))
        self.rtm = args.model_name == 'review_transformer'       # trainer.py:31-36
        self.ExpDataset = corpus.ProdSearchDataset if self.rtm else corpus.ItemPVDataset
        self.ExpDataloader = ProdSearchDataLoader if self.rtm else ItemPVDataloader

    # ------------------------------------------------------------------ training (trainer.py:37-110)
    def train(self, args, global_data, train_prod_data, valid_prod_data):
        valid_dataset = self.ExpDataset(args, global_data, valid_prod_data)
        best_mrr, best_path = 0., ''
        if not self.rtm:
            self.model.clear_loss()
        acc = None                                               # RTM: running loss sum on the device
        step = 0
        t_log = time.time()
        for epoch in range(args.start_epoch + 1, args.max_train_epoch + 1):
            self.model.train()
            train_prod_data.initialize_epoch()
            dataset = self.ExpDataset(args, global_data, train_prod_data)
            prepare_pv = epoch < args.train_pv_epoch + 1
            loader = self.ExpDataloader(args, dataset, prepare_pv=prepare_pv, batch_size=args.batch_size, shuffle=True,
                                        device=args.device, prefetch=getattr(args, 'prefetch', 2))
            for batch in loader:
                if batch is None:                                # no usable entry in this batch (trainer.py:66-67)
                    continue
                # the paragraph-vector epochs of the review transformer yield a sequence of window sub-batches (:68-71)
                for b in (batch if hasattr(batch, '__len__') and not hasattr(batch, 'query_word_idxs') else (batch,)):
                    loss = self.model(b, train_pv=prepare_pv)
                    self.model.zero_grad()
                    loss.backward()
                    self.optim.step()
                    step += 1
                    if self.rtm:
                        acc = loss.detach().clone() if acc is None else acc.add_(loss.detach())
                    if step % args.steps_per_checkpoint == 0:    # the only host sync of the loop
                        n = args.steps_per_checkpoint
                        # row-sparse mode: the kernels drop out-of-range indices / list overflows and set a status word;
                        # it is read HERE, where the loop synchronises anyway, so such a step cannot pass silently
                        chk = getattr(self.model, 'check_index_errors', None)
                        if chk is not None:
                            chk()
                        if self.rtm:
                            ps, iw, tot = 0., 0., float(acc) / n
                            acc = None
                        else:
                            ps, iw = self.model.ps_loss / n, self.model.item_loss / n
                            tot = ps + iw
                            self.model.clear_loss()
                        logger.info("Epoch %d lr = %5.6f loss = %6.2f ps_loss: %3.2f iw_loss: %3.2f time %.2f",
                                    epoch, self.optim.learning_rate, tot, ps, iw, time.time() - t_log)
                        t_log = time.time()
            path = os.path.join(args.save_dir, 'model_epoch_%d.ckpt' % epoch)
            self._save(epoch, path)
            mrr, prec = self.validate(args, global_data, valid_dataset)
            logger.info("Epoch %d: MRR:%s P@1:%s", epoch, mrr, prec)
            if mrr > best_mrr:
                best_mrr = mrr
                best_path = os.path.join(args.save_dir, 'model_best.ckpt')
                shutil.copyfile(path, best_path)
        return best_path

    def _save(self, epoch, checkpoint_path):
        """``trainer.py:112-122``; ``optim`` holds the Adam state in ``torch.optim.Adam.state_dict()`` form."""
        torch.save({'epoch': epoch, 'model': self.model.state_dict(), 'opt': self.args,
                    'optim': self.optim.state_dict() if self.optim is not None else None}, checkpoint_path)

    # ------------------------------------------------------------------ evaluation (trainer.py:124-226)
    def _scores_all(self, args, dataset, topk):
        loader = self.ExpDataloader(args, dataset, batch_size=args.valid_batch_size, shuffle=False, device=args.device)
        tops, scores, ranks, users, queries = [], [], [], [], []
        self.model.eval()
        with torch.no_grad():
            for b in loader:
                ti, ts, rk = evaluate.rank_all(self.model, b, topk)
                tops.append(ti); scores.append(ts); ranks.append(rk)
                users += list(b.user_idxs); queries += list(b.query_idxs)
        return torch.cat(tops), torch.cat(scores), torch.cat(ranks), users, queries

    def _scores_candidates(self, args, dataset, candidate_size, topk=0):
        """Candidate-list evaluation as ``get_prod_scores`` does it (trainer.py:189-226), ranks on the device.
        Returns the rank of the target per (user, query) entry (0 = not among the candidates) and, with ``topk``,
        the ids / scores of the best ``topk`` candidates plus the entries' user and query ids."""
        loader = self.ExpDataloader(args, dataset, batch_size=args.valid_batch_size, shuffle=False, device=args.device)
        seg = (candidate_size - 1) // args.candi_batch_size + 1
        sc, ids, tg, users, queries = [], [], [], [], []
        self.model.eval()
        with torch.no_grad():
            if self.rtm:
                self.model.get_review_embeddings()               # trainer.py:192-193
            for b in loader:
                s = self.model.test(b)
                cand = torch.as_tensor(b.candi_prod_idxs).to(s.device)
                pad = args.candi_batch_size - s.shape[1]
                if pad:                                          # ragged last chunk of this batch
                    s = torch.nn.functional.pad(s, (0, pad), value=float('-inf'))
                    cand = torch.nn.functional.pad(cand, (0, pad), value=-1)
                sc.append(s); ids.append(cand); tg.append(torch.as_tensor(b.target_prod_idxs).to(s.device))
                users += list(b.user_idxs); queries += list(b.query_idxs)
            if self.rtm:
                self.model.clear_review_embbeddings()
        sc = torch.cat(sc).reshape(-1, seg * args.candi_batch_size)[:, :candidate_size]
        ids = torch.cat(ids).reshape(-1, seg * args.candi_batch_size)[:, :candidate_size]
        tg = torch.cat(tg).reshape(-1, seg)[:, 0]
        sc = sc.masked_fill(ids < 0, float('-inf'))              # padding of ragged candidate lists (util.pad(.., -1))
        hit = ids == tg[:, None]
        found = hit.any(1)
        pos = hit.float().argmax(1)                              # first occurrence, like np.where(...)[0][0] after sorting ties
        st = sc.gather(1, pos[:, None])
        col = torch.arange(sc.shape[1], device=sc.device)[None, :]
        ahead = (sc > st) | ((sc == st) & (col < pos[:, None]))
        rank = torch.where(found, ahead.sum(1) + 1, torch.zeros_like(pos))
        if not topk:
            return rank
        ts, ti = sc.topk(min(topk, sc.shape[1]), dim=1)
        return rank, ids.gather(1, ti), ts, users[::seg], queries[::seg]

    def validate(self, args, global_data, valid_dataset):
        if self.rtm:
            size = args.valid_candi_size if args.valid_candi_size >= 1 else global_data.product_size   # trainer.py:127-129
            return evaluate.calc_metrics(self._scores_candidates(args, valid_dataset, size), 100)
        if args.valid_candi_size < 1 or getattr(valid_dataset.prod_data, 'uq_pids', None) is None and \
                all(e[4] is None for e in valid_dataset._data[:1]):
            _, _, rank, _, _ = self._scores_all(args, valid_dataset, 100)
        else:
            rank = self._scores_candidates(args, valid_dataset, args.valid_candi_size)
        return evaluate.calc_metrics(rank, 100)

    def test(self, args, global_data, test_prod_data, rankfname="test.best_model.ranklist", cutoff=100):
        dataset = self.ExpDataset(args, global_data, test_prod_data)
        if self.rtm:
            size = args.test_candi_size if args.test_candi_size >= 1 else global_data.product_size     # trainer.py:141-143
            rank, top_idx, top_score, users, queries = self._scores_candidates(args, dataset, size, topk=min(cutoff, size))
        elif args.test_candi_size >= 1 and test_prod_data.uq_pids is not None:
            rank = self._scores_candidates(args, dataset, args.test_candi_size)
            mrr, prec = evaluate.calc_metrics(rank, cutoff)
            logger.info("Test: MRR:%s P@1:%s", mrr, prec)
            return mrr, prec
        else:
            k = min(cutoff, global_data.product_size, 256)
            top_idx, top_score, rank, users, queries = self._scores_all(args, dataset, k)
        mrr, prec = evaluate.calc_metrics(rank, cutoff)
        logger.info("Test: MRR:%s P@1:%s", mrr, prec)
        with open(os.path.join(args.save_dir, rankfname), 'w') as f:
            f.writelines(evaluate.ranklist_lines([global_data.user_ids[u] for u in users], queries,
                                                 global_data.product_ids, top_idx, top_score))
        return mrr, prec


def train(args):
    """``main.py:train`` (:167-191): seed, read the corpus, train, test the best checkpoint.  Returns (mrr, p@1)."""
    args.start_epoch = 0
    torch.manual_seed(args.seed)
    pyrandom.seed(args.seed)                                     # random.seed(args.seed)
    os.makedirs(args.save_dir, exist_ok=True)
    gd = corpus.GlobalProdSearchData(args, args.data_dir, args.input_train_dir)
    train_pd = corpus.ProdSearchData(args, args.input_train_dir, 'train', gd)
    model, optim = create_model(args, gd, train_pd, args.train_from)
    trainer = Trainer(args, model, optim)
    valid_pd = corpus.ProdSearchData(args, args.input_train_dir, 'valid', gd)
    best = trainer.train(args, gd, train_pd, valid_pd)
    test_pd = corpus.ProdSearchData(args, args.input_train_dir, 'test', gd)
    best_model, _ = create_model(args, gd, train_pd, best)
    return Trainer(args, best_model, None).test(args, gd, test_pd, args.rankfname)
