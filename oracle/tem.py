"""Oracle (TEST INFRASTRUCTURE): CPU restatement of the TEM training/eval step.

Plain PyTorch fp32 CPU ops restating kepingbi/ProdSearch's
``ItemTransformerRanker`` dot-product path; parameters are a dict keyed by the
reference's ``state_dict`` names.  Each function cites the reference lines it
follows.  Pinned by ``tests/golden/*.npz`` (outputs of the reference itself);
see ``oracle/__init__.py`` for who may import this.

Two structures of the same arithmetic are offered:

* ``replicate=True``  — op-for-op the reference: the encoder runs on B sequences
  for the positive and on B*K *expanded copies* for the negatives
  (``item_transformer.py:471-491``).  This is what the CPU baseline times.
* ``replicate=False`` — one encode per batch row, reused for all K+1 scores.
  Identical results whenever no dropout is drawn (dropout 0 or eval), because
  the K copies are equal inputs through a deterministic function.
"""
import math
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- ops
def positional_encoding(max_len, dim):
    """``PositionalEncoding.__init__`` (models/transformer.py:10-17): sin on even,
    cos on odd channels, frequency exp(-2i*ln(1e4)/dim)."""
    pe = torch.zeros(max_len, dim)
    position = torch.arange(0, max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, dim, 2, dtype=torch.float) * -(math.log(10000.0) / dim))
    pe[:, 0::2] = torch.sin(position.float() * div_term)
    pe[:, 1::2] = torch.cos(position.float() * div_term)
    return pe


def gelu_tanh(x):
    """``gelu`` (models/neural.py:7-8)."""
    return 0.5 * x * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * torch.pow(x, 3))))


def vector_mean(inputs, mask):
    """``get_vector_mean`` (models/text_encoder.py:6-16): masked sum over dim 1
    divided by max(count, 1)."""
    s = (inputs * mask.float().unsqueeze(-1)).sum(1)
    cnt = mask.sum(-1)
    cnt = cnt.masked_fill(cnt.eq(0), 1).unsqueeze(-1)
    return s / cnt.float()


def no_dropout(x, site, call):
    return x


class TorchDropout(object):
    """torch's own RNG dropout (what the timed CPU baseline uses, p>0, training)."""
    def __init__(self, p):
        self.p = p

    def __call__(self, x, site, call):
        return F.dropout(x, self.p, True)


def query_encode(P, args, query_word_idxs, word_pad_idx, drop):
    """``word_embeddings`` gather + ``FSEncoder.forward`` (text_encoder.py:32-40) or
    ``AVGEncoder.forward`` (:76-83); call site item_transformer.py:449-450."""
    emb = P['word_embeddings.weight'][query_word_idxs]
    mean = vector_mean(emb, query_word_idxs.ne(word_pad_idx))
    mean_d = drop(mean, 'fs', 0)
    if args.query_encoder_name == 'fs':
        q = torch.tanh(F.linear(mean_d, P['query_encoder.f_W.weight'], P['query_encoder.f_W.bias']))
    else:
        q = mean_d
    return q, mean, mean_d


def mha(P, pre, x_kv, x_q, key_pad, heads, drop, call, keep=None):
    """``MultiHeadedAttention.forward`` live branch (models/neural.py:192-199,
    206-231): K/V/Q linears, reshape to heads, Q/sqrt(dh), QK^T,
    masked_fill(key_pad, -1e18), softmax, dropout, attn.V, final_linear.
    ``key_pad`` [N,S] bool, True = padded key."""
    N, S, d = x_kv.shape
    dh = d // heads
    K = F.linear(x_kv, P[pre + 'linear_keys.weight'], P[pre + 'linear_keys.bias'])
    V = F.linear(x_kv, P[pre + 'linear_values.weight'], P[pre + 'linear_values.bias'])
    Q = F.linear(x_q, P[pre + 'linear_query.weight'], P[pre + 'linear_query.bias'])
    Sq = x_q.shape[1]
    Kh = K.view(N, S, heads, dh).transpose(1, 2)
    Vh = V.view(N, S, heads, dh).transpose(1, 2)
    Qh = Q.view(N, Sq, heads, dh).transpose(1, 2) / math.sqrt(dh)
    scores = torch.matmul(Qh, Kh.transpose(2, 3))
    scores = scores.masked_fill(key_pad[:, None, None, :].expand_as(scores), -1e18)
    attn = torch.softmax(scores, dim=-1)
    attn_d = drop(attn, 'attn', call)
    ctx = torch.matmul(attn_d, Vh).transpose(1, 2).contiguous().view(N, Sq, d)
    out = F.linear(ctx, P[pre + 'final_linear.weight'], P[pre + 'final_linear.bias'])
    if keep is not None:
        keep.update(K=K, V=V, Qs=Qh.transpose(1, 2).reshape(N, Sq, d), attn=attn, ctx=ctx)
    return out


def ffn(P, pre, x, drop, call, keep=None):
    """``PositionwiseFeedForward.forward`` (models/neural.py:30-33): LN(eps 1e-6) ->
    w_1 -> tanh-GELU -> dropout_1 -> w_2 -> dropout_2 -> + x."""
    d = x.shape[-1]
    ln = F.layer_norm(x, (d,), P[pre + 'layer_norm.weight'], P[pre + 'layer_norm.bias'], 1e-6)
    a1 = F.linear(ln, P[pre + 'w_1.weight'], P[pre + 'w_1.bias'])
    h1 = drop(gelu_tanh(a1), 'ff1', call)
    o2 = drop(F.linear(h1, P[pre + 'w_2.weight'], P[pre + 'w_2.bias']), 'ff2', call)
    if keep is not None:
        keep.update(ln1=ln, a1=a1, h1=h1)
    return o2 + x


def encoder_encode(P, args, seq_emb, seq_mask, pe, drop, call, keep=None):
    """``TransformerEncoder.encode`` (models/transformer.py:71-88) with
    ``TransformerEncoderLayer.forward`` (:47-57): x = in*mask (+ pe[:S], no sqrt(d)
    scaling, no dropout); layer i applies its pre-LayerNorm only when i != 0;
    key-padding mask = 1-mask; residuals; final LayerNorm(eps 1e-6)."""
    te = 'transformer_encoder.'
    d = seq_emb.shape[-1]
    S = seq_emb.shape[1]
    x = seq_emb * seq_mask[:, :, None].float()
    if args.use_pos_emb:
        x = x + pe[None, :S]
    if keep is not None:
        keep['x'] = x
    key_pad = ~seq_mask.bool()
    for i in range(args.inter_layers):
        lp = te + 'transformer_inter.%d.' % i
        inp = x
        if i != 0:
            inp_n = F.layer_norm(inp, (d,), P[lp + 'layer_norm.weight'], P[lp + 'layer_norm.bias'], 1e-6)
        else:
            inp_n = inp
        lk = {} if keep is not None else None
        ctx_out = mha(P, lp + 'self_attn.', inp_n, inp_n, key_pad, args.heads, drop, (call, i), lk)
        y1 = drop(ctx_out, 'ctx', (call, i)) + inp
        x = ffn(P, lp + 'feed_forward.', y1, drop, (call, i), lk)
        if keep is not None:
            lk.update(y1=y1, y2=x)
            keep['layer%d' % i] = lk
    out = F.layer_norm(x, (d,), P[te + 'layer_norm.weight'], P[te + 'layer_norm.bias'], 1e-6)
    return out


def bce_rank_loss(pos_scores, neg_scores, pos_weight):
    """Ranking loss (item_transformer.py:500-514): BCE-with-logits over
    cat[pos, neg] with targets [1,0..0] and weights [pos_weight,1..1], summed
    over the 1+K columns, mean over the batch."""
    B, K = neg_scores.shape
    scores = torch.cat([pos_scores.unsqueeze(-1), neg_scores], dim=-1)
    target = torch.cat([torch.ones(B, 1), torch.zeros(B, K)], dim=-1)
    weight = torch.cat([torch.ones(B, 1) * float(pos_weight), torch.ones(B, K)], dim=-1)
    loss = F.binary_cross_entropy_with_logits(scores, target, weight=weight, reduction='none')
    return loss.sum(-1).mean()


def item_to_words(P, target_prod_idxs, pos_iword_idxs, neg_word_idxs, word_pad_idx, keep=None):
    """``item_to_words`` (item_transformer.py:260-283): PV word-prediction loss of the
    target item: pos word and K sampled words per window slot, score = w.p +
    word_bias[w], BCE(1/0) summed over 1+K, masked mean over the window
    (``get_vector_mean`` with mask idx != pad), mean over the batch.
    ``neg_word_idxs`` is the recorded multinomial draw, [B, W*K]."""
    B, W = pos_iword_idxs.shape
    prod = P['product_emb.weight'][target_prod_idxs]
    wpos = P['word_embeddings.weight'][pos_iword_idxs]
    wneg = P['word_embeddings.weight'][neg_word_idxs.view(B, -1)]
    out_pos = torch.bmm(wpos, prod.unsqueeze(2))
    out_neg = torch.bmm(wneg, prod.unsqueeze(2)).view(B, W, -1)
    out_pos = out_pos + P['word_bias'][pos_iword_idxs.view(-1)].view(B, W, 1)
    out_neg = out_neg + P['word_bias'][neg_word_idxs.reshape(-1)].view(B, W, -1)
    scores = torch.cat((out_pos, out_neg), dim=-1)
    target = torch.cat((torch.ones_like(out_pos), torch.zeros_like(out_neg)), dim=-1)
    loss = F.binary_cross_entropy_with_logits(scores, target, reduction='none').sum(-1)
    loss = vector_mean(loss.unsqueeze(-1), pos_iword_idxs.ne(word_pad_idx))
    if keep is not None:
        keep['word_scores'] = scores
    return loss.mean()


# ------------------------------------------------------------------ TEM train step
def tem_forward(P, args, batch, neg_item_idxs, neg_word_idxs, vocab_size, product_size,
                training=True, replicate=False, drop=None, keep=None):
    """``forward_dotproduct`` (item_transformer.py:440-520).  ``neg_item_idxs`` [B,K]
    and ``neg_word_idxs`` [B,W*K] are the two multinomial draws (items first,
    :447; words second, :268), injected so index work is bit-exact.
    Returns (loss, ps_loss, item_loss)."""
    drop = drop if (drop is not None and training) else no_dropout
    word_pad, prod_pad = vocab_size - 1, product_size
    qw, tgt, ui = batch.query_word_idxs, batch.target_prod_idxs, batch.u_item_idxs
    B, L = ui.shape
    K = neg_item_idxs.shape[1]
    d = args.embedding_size
    pe = positional_encoding(5000, d)
    query_emb, qmean, _ = query_encode(P, args, qw, word_pad, drop)
    u_mask = ui.ne(prod_pad)
    seq_mask = torch.cat([torch.ones(B, 1, dtype=torch.bool), u_mask], dim=1)      # :451-454
    hist_tab = P['hist_product_emb.weight'] if args.sep_prod_emb else P['product_emb.weight']
    target_emb = P['product_emb.weight'][tgt]                                       # :464
    neg_emb = P['product_emb.weight'][neg_item_idxs]                                # :465
    u_emb = hist_tab[ui]                                                            # :466-469
    pos_seq = torch.cat([query_emb.unsqueeze(1), u_emb], dim=1)                     # :471
    out_pos = -1 if args.use_item_pos else 0                                        # :482
    k0 = {} if keep is not None else None
    top = encoder_encode(P, args, pos_seq, seq_mask, pe, drop, 0, k0)               # :483
    pos_out = top[:, out_pos, :]
    pos_scores = torch.bmm(pos_out.unsqueeze(1), target_emb.unsqueeze(2)).view(B)   # :485
    if replicate:
        neg_seq = torch.cat([query_emb.unsqueeze(1).expand(-1, K, -1).unsqueeze(2),
                             u_emb.unsqueeze(1).expand(-1, K, -1, -1)], dim=2)      # :473-476
        neg_mask = seq_mask.unsqueeze(1).expand(-1, K, -1)                          # :458-460
        top_n = encoder_encode(P, args, neg_seq.reshape(B * K, L + 1, d),
                               neg_mask.reshape(B * K, L + 1), pe, drop, 1)         # :486-491
        neg_out = top_n[:, out_pos, :].view(B, K, d)
    else:
        neg_out = pos_out.unsqueeze(1).expand(-1, K, -1)
    neg_scores = (neg_out * neg_emb).sum(-1)                                        # :493-494
    if args.sim_func == 'bias_product':                                             # :495-499
        pos_scores = pos_scores + P['product_bias'][tgt]
        neg_scores = neg_scores + P['product_bias'][neg_item_idxs]
    ps_loss = bce_rank_loss(pos_scores, neg_scores, K if args.pos_weight else 1)     # :500-514
    item_loss = item_to_words(P, tgt, batch.pos_iword_idxs, neg_word_idxs, word_pad, keep)  # :515
    if keep is not None:
        keep.update(k0)
        keep.update(query_mean=qmean, query_emb=query_emb, seq_mask=seq_mask, enc=pos_out,
                    pos_scores=pos_scores, neg_scores=neg_scores)
    return ps_loss + item_loss, ps_loss, item_loss


def tem_test(P, args, batch, vocab_size, product_size, replicate=False):
    """``test_dotproduct`` (item_transformer.py:111-146): eval-mode scores
    [B, candi_k] of every candidate against the encoded (query, history)."""
    word_pad, prod_pad = vocab_size - 1, product_size
    qw, ui, candi = batch.query_word_idxs, batch.u_item_idxs, batch.candi_prod_idxs
    B, L = ui.shape
    C = candi.shape[1]
    d = args.embedding_size
    pe = positional_encoding(5000, d)
    query_emb, _, _ = query_encode(P, args, qw, word_pad, no_dropout)
    seq_mask = torch.cat([torch.ones(B, 1, dtype=torch.bool), ui.ne(prod_pad)], dim=1)
    hist_tab = P['hist_product_emb.weight'] if args.sep_prod_emb else P['product_emb.weight']
    candi_emb = P['product_emb.weight'][candi]
    seq = torch.cat([query_emb.unsqueeze(1), hist_tab[ui]], dim=1)
    out_pos = -1 if args.use_item_pos else 0
    if replicate:
        seq_r = seq.unsqueeze(1).expand(-1, C, -1, -1).reshape(B * C, L + 1, d)
        mask_r = seq_mask.unsqueeze(1).expand(-1, C, -1).reshape(B * C, L + 1)
        top = encoder_encode(P, args, seq_r, mask_r, pe, no_dropout, 0)
        out = top[:, out_pos, :].view(B, C, d)
    else:
        top = encoder_encode(P, args, seq, seq_mask, pe, no_dropout, 0)
        out = top[:, out_pos, :].unsqueeze(1).expand(-1, C, -1)
    scores = (out * candi_emb).sum(-1)
    if args.sim_func == 'bias_product':
        scores = scores + P['product_bias'][candi]
    return scores


# -------------------------------------------------------------- QEM ("HEM-like") step
def qem_forward(P, args, batch, neg_item_idxs, neg_word_idxs, vocab_size, product_size,
                training=True, drop=None, keep=None):
    """``forward_attn`` with ``model_name == 'QEM'`` (item_transformer.py:361-377,
    410-438): score = FS(query).item, no attention/transformer; + item_to_words."""
    drop = drop if (drop is not None and training) else no_dropout
    word_pad = vocab_size - 1
    tgt = batch.target_prod_idxs
    K = neg_item_idxs.shape[1]
    query_emb, qmean, _ = query_encode(P, args, batch.query_word_idxs, word_pad, drop)
    target_emb = P['product_emb.weight'][tgt]
    neg_emb = P['product_emb.weight'][neg_item_idxs]
    pos_scores = (query_emb * target_emb).sum(-1)
    neg_scores = (query_emb.unsqueeze(1) * neg_emb).sum(-1)
    if args.sim_func == 'bias_product':
        pos_scores = pos_scores + P['product_bias'][tgt]
        neg_scores = neg_scores + P['product_bias'][neg_item_idxs]
    ps_loss = bce_rank_loss(pos_scores, neg_scores, K if args.pos_weight else 1)
    item_loss = item_to_words(P, tgt, batch.pos_iword_idxs, neg_word_idxs, word_pad, keep)
    if keep is not None:
        keep.update(query_mean=qmean, query_emb=query_emb, enc=query_emb,
                    pos_scores=pos_scores, neg_scores=neg_scores)
    return ps_loss + item_loss, ps_loss, item_loss


def qem_test(P, args, batch, vocab_size, product_size):
    """``test_attn`` with ``model_name == 'QEM'`` (item_transformer.py:148-160,189-195)."""
    query_emb, _, _ = query_encode(P, args, batch.query_word_idxs, vocab_size - 1, no_dropout)
    candi = batch.candi_prod_idxs
    scores = (query_emb.unsqueeze(1) * P['product_emb.weight'][candi]).sum(-1)
    if args.sim_func == 'bias_product':
        scores = scores + P['product_bias'][candi]
    return scores


# ------------------------------------------------------------------------ backward
def grads_of(loss, P, pad_rows):
    """``loss.backward()`` (trainer.py:77) restated with autograd on the oracle's own
    forward.  Dense grads; ``padding_idx`` rows of the embedding tables are zeroed
    (nn.Embedding(padding_idx=...), item_transformer.py:46,48,70-71); parameters
    the loss does not reach get ``None`` exactly as in the reference."""
    names = [n for n, p in P.items() if p.requires_grad]
    gs = torch.autograd.grad(loss, [P[n] for n in names], allow_unused=True)
    out = {}
    for n, g in zip(names, gs):
        if g is not None and n in pad_rows:
            g = g.clone()
            g[pad_rows[n]] = 0
        out[n] = g
    return out


def tem_pad_rows(args, vocab_size, product_size):
    rows = {'product_emb.weight': product_size, 'word_embeddings.weight': vocab_size - 1,
            'seg_embeddings.weight': 3}
    if args.sep_prod_emb:
        rows['hist_product_emb.weight'] = product_size
    return rows


# ----------------------------------------------------------------------- metrics
def rank_metrics(scores, candi_prod_idxs, target_prod_idxs, cutoff=100):
    """``argsort(axis=-1)[:, ::-1]`` + ``calc_metrics`` (trainer.py:136, 171-187):
    descending ranklist, MRR with cutoff, P@1."""
    import numpy as np
    s = scores.detach().cpu().numpy()
    order = s.argsort(axis=-1)[:, ::-1]
    cand = candi_prod_idxs.cpu().numpy()
    tgt = target_prod_idxs.cpu().numpy()
    mrr = prec = 0.0
    for i in range(s.shape[0]):
        hit = np.where(cand[i][order[i]] == tgt[i])[0]
        if len(hit):
            rank = hit[0] + 1
            if cutoff < 0 or rank <= cutoff:
                mrr += 1.0 / rank
            if rank == 1:
                prec += 1
    return order.copy(), mrr / s.shape[0], prec / s.shape[0]
