"""Oracle (TEST INFRASTRUCTURE): CPU restatement of the RTM training/eval step.

Plain PyTorch fp32 CPU ops restating kepingbi/ProdSearch's ``ProductRanker``
(models/ps_model.py:53-370) with the ``pv`` (models/PV.py), ``pvc``
(models/PVC.py), ``fs`` and ``avg`` (models/text_encoder.py) review encoders; parameters are a dict keyed by the reference's
``state_dict`` names.  Pinned by tests/golden/rtm_*.npz (outputs of the reference
itself); see oracle/__init__.py for who may import this.

Sequences are indexed n = b*(1+K) + j with j = 0 the positive product and j = 1+k
negative k — the order the HIP kernels use; the reference's two encoder calls
(B positives, then B*K negatives; ps_model.py:336-339) are the same rows split in two.
"""
import torch
import torch.nn.functional as F

from .tem import (bce_rank_loss, encoder_encode, no_dropout, positional_encoding,  # noqa: F401
                  query_encode, vector_mean)


def pv_loss_terms(word_emb_pos, word_emb_neg, review_vec, word_mask):
    """Word-prediction loss shared by ``ParagraphVector.forward`` (PV.py:59-66) and
    ``ParagraphVectorCorruption.forward`` (PVC.py:83-91): per review, BCE(1/0) of
    w.v over the window word and its K sampled words, summed over 1+K, masked mean
    over the window.  word_emb_pos [N,W,d], word_emb_neg [N,W*K,d], review_vec [N,d]."""
    N, W, _ = word_emb_pos.shape
    out_pos = torch.bmm(word_emb_pos, review_vec.unsqueeze(2))
    out_neg = torch.bmm(word_emb_neg, review_vec.unsqueeze(2)).view(N, W, -1)
    scores = torch.cat((out_pos, out_neg), dim=-1)
    target = torch.cat((torch.ones_like(out_pos), torch.zeros_like(out_neg)), dim=-1)
    loss = F.binary_cross_entropy_with_logits(scores, target, reduction='none').sum(-1)
    return vector_mean(loss.unsqueeze(-1), word_mask), scores


def pvc_para_vector(P, idxs, word_pad, tok_mult):
    """``ParagraphVectorCorruption.get_para_vector`` (PVC.py:56-61): gather context
    embeddings (alias of word_embeddings, PVC.py:30), token dropout applied to the DATA
    (PVC.py:46-54: autograd never sees the mask or the 1/(1-p) scale), masked mean.
    ``tok_mult`` [N,WL] multiplier (0 or 1/(1-p)) or None."""
    out = []
    for lo in range(0, idxs.shape[0], 4096):          # per-review op: chunks only bound the [N,WL,d] temporaries
        ix = idxs[lo:lo + 4096]
        emb = P['word_embeddings.weight'][ix]
        if tok_mult is not None:
            # value uses the corrupted data, gradient flows as if uncorrupted
            emb = emb + (emb.detach() * tok_mult[lo:lo + 4096].unsqueeze(-1) - emb.detach())
        out.append(vector_mean(emb, ix.ne(word_pad)))
    return torch.cat(out, dim=0)


def rtm_forward(P, args, batch, neg_word_idxs, vocab_size, review_count, training=True, train_pv=True,
                drop=None, tok_drop=None, keep=None):
    """``ProductRanker.forward`` (ps_model.py:241-358) for review_encoder_name in {pv, pvc}.
    ``neg_word_idxs`` [B*R, W*K] is the multinomial draw of the PV loss (PV.py:57 / PVC.py:81),
    only used when ``train_pv``.  ``drop(x, kind, call)`` as in oracle.tem; ``tok_drop(shape, which)``
    returns the token-dropout multipliers of the pvc encoder ('pos' / 'neg').  Returns (loss, ps, pv)."""
    drop = drop if (drop is not None and training) else no_dropout
    word_pad, rev_pad = vocab_size - 1, review_count - 1
    enc_name = args.review_encoder_name
    qw = batch.query_word_idxs
    pos_r, neg_r = batch.pos_prod_ridxs, batch.neg_prod_ridxs
    B, R = pos_r.shape
    K = neg_r.shape[1]
    d = args.embedding_size
    pe = positional_encoding(5000, d)
    query_emb, _, _ = query_encode(P, args, qw, word_pad, drop)                       # :257-258
    pv_loss = None
    if enc_name == 'pv':
        table = P['review_encoder.review_embeddings.weight']
        if train_pv:                                                                 # :267-271
            rw = batch.pos_prod_rword_idxs.view(B * R, -1)
            wpos = P['word_embeddings.weight'][rw]                                   # :261
            rv = drop(table[pos_r.view(-1)], 'rev_pv', 0)                            # PV.py:53-54
            wneg = P['word_embeddings.weight'][neg_word_idxs.view(B * R, -1)]
            per_rev, pv_scores = pv_loss_terms(wpos, wneg, rv, batch.pos_prod_rword_masks.view(B * R, -1).bool())
            pos_rev = rv
        else:
            pos_rev = table[pos_r.view(-1)]                                          # :288-289
        neg_rev = table[neg_r]                                                       # :296-297
    elif enc_name == 'pvc':
        tm = (lambda shape, which: tok_drop(shape, which)) if (tok_drop is not None and training) else (lambda s, w: None)
        if train_pv:                                                                 # :272-276
            rw = batch.pos_prod_rword_idxs.view(B * R, -1)
            wpos = P['word_embeddings.weight'][rw]
            pvc_idx = batch.pos_prod_rword_idxs_pvc.view(B * R, -1)
            pos_rev = pvc_para_vector(P, pvc_idx, word_pad, None)                    # PVC.py:76 (uncorrupted)
            corr = pvc_para_vector(P, pvc_idx, word_pad, tm(tuple(pvc_idx.shape), 'pos'))   # PVC.py:77-78
            wneg = P['word_embeddings.weight'][neg_word_idxs.view(B * R, -1)]
            per_rev, pv_scores = pv_loss_terms(wpos, wneg, corr, batch.pos_prod_rword_masks.view(B * R, -1).bool())
            neg_idx = batch.neg_prod_rword_idxs_pvc.view(B * K * R, -1)
        else:                                                                        # :290-293, :299-300
            pidx = batch.pos_prod_rword_idxs.view(B * R, -1)
            pos_rev = pvc_para_vector(P, pidx, word_pad, tm(tuple(pidx.shape), 'pos'))
            neg_idx = batch.neg_prod_rword_idxs.view(B * K * R, -1)
        neg_rev = pvc_para_vector(P, neg_idx, word_pad, tm(tuple(neg_idx.shape), 'neg')).view(B, K, R, d)
    elif enc_name in ('fs', 'avg'):
        # FSEncoder / AVGEncoder over the review words with the BATCH's word masks (ps_model.py:301-305,
        # text_encoder.py:32-40 / :76-83): masked mean -> dropout (-> tanh(f_W . + b)); no further dropout_layer, no PV loss
        def enc(idx, mask, which):
            n = idx.shape[0] * idx.shape[1] if idx.dim() == 3 else idx.shape[0]
            idx2, m2 = idx.reshape(-1, idx.shape[-1]), mask.reshape(-1, mask.shape[-1]).bool()
            out = []
            for lo in range(0, idx2.shape[0], 4096):
                out.append(vector_mean(P['word_embeddings.weight'][idx2[lo:lo + 4096]], m2[lo:lo + 4096]))
            mean = drop(torch.cat(out, 0), which, 0)
            if enc_name == 'fs':
                mean = torch.tanh(F.linear(mean, P['review_encoder.f_W.weight'], P['review_encoder.f_W.bias']))
            return mean
        pos_rev = enc(batch.pos_prod_rword_idxs, batch.pos_prod_rword_masks, 'rev_pos').view(B * R, d)
        neg_rev = enc(batch.neg_prod_rword_idxs.view(B * K * R, -1), batch.neg_prod_rword_masks.view(B * K * R, -1),
                      'rev_neg').view(B, K, R, d)
        train_pv = False
    else:
        raise NotImplementedError(enc_name)
    if train_pv:
        sample_count = pos_r.ne(rev_pad).float().sum(-1)                             # :277
        pv_loss = per_rev.sum() / sample_count.sum()                                 # :280
    if enc_name in ('pv', 'pvc'):
        pos_rev = drop(pos_rev, 'rev_pos', 0)                                        # :303
        neg_rev = drop(neg_rev.reshape(B, K, R, d), 'rev_neg', 0)                    # :304
    pos_rev = pos_rev.view(B, R, d)
    neg_rev = neg_rev.reshape(B, K, R, d)

    pos_mask = torch.cat([torch.ones(B, 1, dtype=torch.bool), pos_r.ne(rev_pad)], dim=1)        # :316
    neg_ridx_mask = neg_r.ne(rev_pad)
    neg_mask = torch.cat([torch.ones(B, K, 1, dtype=torch.bool), neg_ridx_mask], dim=2)         # :317-318
    pos_seq = torch.cat((query_emb.unsqueeze(1), pos_rev), dim=1)                                # :320
    neg_seq = torch.cat((query_emb.unsqueeze(1).expand(-1, K, -1).unsqueeze(2), neg_rev), dim=2)  # :322-323
    if args.use_seg_emb:                                                                          # :326-328
        pos_seq = pos_seq + P['seg_embeddings.weight'][batch.pos_seg_idxs]
        neg_seq = neg_seq + P['seg_embeddings.weight'][batch.neg_seg_idxs]
    if getattr(args, 'use_item_emb', False):                                                      # :329-333
        pos_seq = pos_seq + P['product_emb.weight'][batch.pos_item_idxs]
        neg_seq = neg_seq + P['product_emb.weight'][batch.neg_item_idxs]
    if getattr(args, 'use_user_emb', False):                                                      # :334-338
        pos_seq = pos_seq + P['user_emb.weight'][batch.pos_user_idxs]
        neg_seq = neg_seq + P['user_emb.weight'][batch.neg_user_idxs]
    k0 = {} if keep is not None else None
    te = 'transformer_encoder.'
    top_p = encoder_encode(P, args, pos_seq, pos_mask, pe, drop, 0, k0)                           # :336
    top_n = encoder_encode(P, args, neg_seq.reshape(B * K, R + 1, d), neg_mask.reshape(B * K, R + 1), pe, drop, 1)
    pos_scores = F.linear(top_p[:, 0, :], P[te + 'wo.weight'], P[te + 'wo.bias']).squeeze(-1)   # transformer.py:95-96
    neg_scores = F.linear(top_n[:, 0, :], P[te + 'wo.weight'], P[te + 'wo.bias']).squeeze(-1).view(B, K)
    pos_weight = K if args.pos_weight else 1
    weight = torch.cat([torch.ones(B, 1) * float(pos_weight), neg_ridx_mask.sum(-1).ne(0).float()], dim=-1)   # :344-345
    scores = torch.cat([pos_scores.unsqueeze(-1), neg_scores], dim=-1)
    target = torch.cat([torch.ones(B, 1), torch.zeros(B, K)], dim=-1)
    ps_loss = F.binary_cross_entropy_with_logits(scores, target, weight=weight, reduction='none').sum(-1).mean()
    loss = ps_loss + pv_loss if pv_loss is not None else ps_loss                                  # :357
    if keep is not None:
        keep.update(k0)
        keep.update(query_emb=query_emb, pos_rev=pos_rev, neg_rev=neg_rev, scores=scores,
                    enc_pos=top_p[:, 0, :], enc_neg=top_n[:, 0, :].view(B, K, d),
                    pos_seq=pos_seq, neg_seq=neg_seq, pos_mask=pos_mask, neg_mask=neg_mask, weight=weight)
        if train_pv:
            keep.update(pv_scores=pv_scores, per_rev=per_rev)
    return loss, ps_loss, pv_loss


def rtm_review_embeddings(P, args, review_words, vocab_size):
    """``get_review_embeddings`` (ps_model.py:186-203): the per-review vectors ``test`` indexes —
    the pv table itself, or the UNcorrupted pvc mean of each review's words with the last row 0."""
    if args.review_encoder_name == 'pv':
        return P['review_encoder.review_embeddings.weight']
    emb = pvc_para_vector(P, review_words[:-1], vocab_size - 1, None)       # masked mean of the non-pad words
    last = torch.zeros(1, emb.shape[1])
    if args.review_encoder_name == 'fs':                                     # :198-200: review_encoder(words, words != pad), eval mode
        w, bias = P['review_encoder.f_W.weight'], P['review_encoder.f_W.bias']
        emb = torch.tanh(F.linear(emb, w, bias))
        # the reference walks review_words in slices of 128 up to row ceil((RC-1)/128)*128: unless RC-1 is a multiple of
        # 128 that includes the padding review, whose (empty) mean goes through the projection too: tanh(bias), not 0
        if (review_words.shape[0] - 1) % 128 != 0:
            last = torch.tanh(bias).unsqueeze(0)
    return torch.cat([emb, last], dim=0)


def rtm_test(P, args, batch, review_embeddings, vocab_size, review_count):
    """``ProductRanker.test`` (ps_model.py:205-239): eval scores [B, candi_k]."""
    word_pad, rev_pad = vocab_size - 1, review_count - 1
    qw, cr = batch.query_word_idxs, batch.candi_prod_ridxs
    B, C, R = cr.shape
    d = args.embedding_size
    pe = positional_encoding(5000, d)
    query_emb, _, _ = query_encode(P, args, qw, word_pad, no_dropout)
    rev = review_embeddings[cr]
    mask = torch.cat([torch.ones(B, C, 1, dtype=torch.bool), cr.ne(rev_pad)], dim=2)
    seq = torch.cat((query_emb.unsqueeze(1).expand(-1, C, -1).unsqueeze(2), rev), dim=2)
    if args.use_seg_emb:
        seq = seq + P['seg_embeddings.weight'][batch.candi_seg_idxs]
    if getattr(args, 'use_user_emb', False):                                                      # :233-235
        seq = seq + P['user_emb.weight'][batch.candi_seq_user_idxs]
    if getattr(args, 'use_item_emb', False):                                                      # :236-238
        seq = seq + P['product_emb.weight'][batch.candi_seq_item_idxs]
    top = encoder_encode(P, args, seq.reshape(B * C, R + 1, d), mask.reshape(B * C, R + 1), pe, no_dropout, 0)
    te = 'transformer_encoder.'
    return F.linear(top[:, 0, :], P[te + 'wo.weight'], P[te + 'wo.bias']).squeeze(-1).view(B, C)
