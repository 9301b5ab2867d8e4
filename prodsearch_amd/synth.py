"""Synthetic "Amazon-5core-shaped" inputs for the TEM hot path.

The reference ships no data (its ``data/`` directory is code only), so shapes and
distributions are this build's choice, fixed here so tests, fixtures and the
bench all draw the same batches:

* query length ~U{2..6}, padded to Q with ``vocab_size-1``;
* history length ``min(L, Geometric(0.15))`` with >=5 % zero-history users, padded
  with ``product_size`` (reference pads: ``data/item_pv_dataloader.py:135-139``);
* history/target item ids Zipf(1.05) over a fixed permutation, negatives uniform
  over ``[0, P)`` (reference: ``prod_dists = ones(P)``, ``item_transformer.py:33``);
* word ids from ``word_dists`` = ``(freq/sum)^0.75`` renormalised with Zipf(1.0)
  freq and pad weight 0 (reference: ``data/data_util.py:155-162``).

Everything is numpy ``PCG64`` seeded; nothing here touches the GPU.
"""
import hashlib
import numpy as np
import torch

from .batch import ItemPVBatch


def rng_for(seed):
    return np.random.Generator(np.random.PCG64(int(seed)))


def make_word_dists(vocab_size, seed=7):
    """float64 [V] negative-word distribution; last entry (pad) has weight 0."""
    rng = rng_for(seed)
    n = vocab_size - 1
    freq = 1.0 / np.arange(1, n + 1, dtype=np.float64)
    freq = freq[rng.permutation(n)]
    wf = freq / freq.sum()
    wf = np.power(wf, 0.75)
    wf = wf / wf.sum()
    return np.concatenate([wf, [0.0]])


def _zipf_ids(rng, n_ids, size, a=1.05, perm_seed=11):
    if n_ids > (1 << 21):
        # huge catalogues (config 5: 50 M items): no n_ids-sized weight / permutation arrays — bounded Zipf
        # ranks by rejection, scattered over the id space by an odd multiplier coprime to n_ids
        n = int(np.prod(size))
        out = np.empty(0, dtype=np.int64)
        while out.size < n:
            r = rng.zipf(a, size=4 * n + 64)
            out = np.concatenate([out, r[r <= n_ids].astype(np.int64)])
        mult = 2654435761
        while np.gcd(mult, n_ids) != 1:
            mult += 2
        return (((out[:n] - 1) * mult) % n_ids).reshape(size)
    w = 1.0 / np.power(np.arange(1, n_ids + 1, dtype=np.float64), a)
    w /= w.sum()
    perm = rng_for(perm_seed).permutation(n_ids)
    return perm[rng.choice(n_ids, size=size, p=w)]


def make_tem_batch(seed, B, product_size, vocab_size, Q=8, L=20, W=1, C=0,
                   word_dists=None, zero_hist_frac=0.05, full_history=False):
    """One TEM batch (``ItemPVBatch``), CPU int64 tensors."""
    rng = rng_for(seed)
    P, V = product_size, vocab_size
    if word_dists is None:
        word_dists = make_word_dists(V)
    # queries
    qlen = rng.integers(min(2, Q), min(6, Q) + 1, size=B)
    qw = np.full((B, Q), V - 1, dtype=np.int64)
    words = rng.choice(V, size=(B, Q), p=word_dists)
    for b in range(B):
        qw[b, :qlen[b]] = words[b, :qlen[b]]
    # history
    if full_history:
        hlen = np.full(B, L, dtype=np.int64)
    else:
        hlen = np.minimum(L, rng.geometric(0.15, size=B))
        hlen[rng.random(B) < zero_hist_frac] = 0
    ui = np.full((B, L), P, dtype=np.int64)
    items = _zipf_ids(rng, P, (B, L))
    for b in range(B):
        ui[b, :hlen[b]] = items[b, :hlen[b]]
    target = _zipf_ids(rng, P, B)
    # positive item words (pv window); a few pads when W > 1
    pw = rng.choice(V, size=(B, W), p=word_dists).astype(np.int64)
    if W > 1:
        pw[rng.random((B, W)) < 0.15] = V - 1
        pw[:, 0] = np.where(pw[:, 0] == V - 1, 0, pw[:, 0])
    candi = []
    if C > 0:
        candi = rng.integers(0, P, size=(B, C)).astype(np.int64)
        candi[:, 0] = target            # target among candidates (as validation does)
        candi[-1, C - max(1, C // 8):] = P   # ragged tail padded with P (item_pv_dataloader.py:44)
    return ItemPVBatch(qw, target.astype(np.int64), ui, pw,
                       query_idxs=list(range(B)), user_idxs=list(range(B)),
                       candi_prod_idxs=candi)


def sample_negatives(seed, B, K, W, product_size, word_dists):
    """Host-side stand-in for the two ``torch.multinomial`` draws of one forward
    (items first, ``item_transformer.py:447``; words second, ``:268``)."""
    rng = rng_for(seed)
    neg_items = rng.integers(0, product_size, size=(B, K)).astype(np.int64)
    neg_words = rng.choice(len(word_dists), size=(B, W * K), p=word_dists).astype(np.int64)
    return torch.from_numpy(neg_items), torch.from_numpy(neg_words)


def tem_param_shapes(args, vocab_size, product_size):
    """state_dict name -> shape of the reference's ItemTransformerRanker
    (item_transformer.py:46-84, transformer.py:59-69, neural.py:20-26,86-96)."""
    d, F = args.embedding_size, args.ff_size
    P1, V = product_size + 1, vocab_size
    shapes = {
        'product_bias': (P1,),
        'word_bias': (V,),
        'product_emb.weight': (P1, d),
    }
    if args.sep_prod_emb:
        shapes['hist_product_emb.weight'] = (P1, d)
    shapes['word_embeddings.weight'] = (V, d)
    if args.model_name == 'item_transformer':
        te = 'transformer_encoder.'
        for i in range(args.inter_layers):
            p = te + 'transformer_inter.%d.' % i
            for lin in ('linear_keys', 'linear_values', 'linear_query', 'final_linear'):
                shapes[p + 'self_attn.%s.weight' % lin] = (d, d)
                shapes[p + 'self_attn.%s.bias' % lin] = (d,)
            shapes[p + 'feed_forward.w_1.weight'] = (F, d)
            shapes[p + 'feed_forward.w_1.bias'] = (F,)
            shapes[p + 'feed_forward.w_2.weight'] = (d, F)
            shapes[p + 'feed_forward.w_2.bias'] = (d,)
            shapes[p + 'feed_forward.layer_norm.weight'] = (d,)
            shapes[p + 'feed_forward.layer_norm.bias'] = (d,)
            shapes[p + 'layer_norm.weight'] = (d,)
            shapes[p + 'layer_norm.bias'] = (d,)
        shapes[te + 'layer_norm.weight'] = (d,)
        shapes[te + 'layer_norm.bias'] = (d,)
        shapes[te + 'wo.weight'] = (1, d)
        shapes[te + 'wo.bias'] = (1,)
    else:
        p = 'attention_encoder.'
        for lin in ('linear_keys', 'linear_values', 'linear_query', 'final_linear'):
            shapes[p + '%s.weight' % lin] = (d, d)
            shapes[p + '%s.bias' % lin] = (d,)
    if args.query_encoder_name == 'fs':
        shapes['query_encoder.f_W.weight'] = (d, d)
        shapes['query_encoder.f_W.bias'] = (d,)
    shapes['seg_embeddings.weight'] = (4, d)
    return shapes


def make_state_dict(shapes, seed, pad_rows=None):
    """Deterministic fp32 weights for parity tests: N(0, s) with s chosen per
    tensor so activations stay O(1) (tables 0.5, matrices 1/sqrt(fan_in),
    LayerNorm gains 1 + 0.1 N, biases 0.05 N).  ``pad_rows`` = {name: row} rows
    zeroed (the reference's ``padding_idx`` rows of product_emb)."""
    rng = rng_for(seed)
    sd = {}
    for name, shp in shapes.items():
        x = rng.standard_normal(shp).astype(np.float32)
        if name.endswith('layer_norm.weight'):
            x = 1.0 + 0.1 * x
        elif len(shp) == 1:
            x = 0.05 * x
        elif 'emb' in name:
            x = 0.5 * x
        else:
            x = x / np.sqrt(shp[-1])
        sd[name] = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    for name, row in (pad_rows or {}).items():
        if name in sd:
            sd[name][row] = 0.0
    return sd


def checksum(t):
    """sha256 of the raw little-endian bytes of a tensor/array (fixture pinning)."""
    a = t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


# ------------------------------------------------------------------ synthetic corpus (batch builder)
class _NS(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


def make_corpus(seed, n_users=60, n_products=200, n_queries=40, vocab_size=500, Q=6, W=1,
                max_reviews_per_user=120, train_frac=0.8):
    """Python structures shaped like the reference's ``global_data`` / ``prod_data`` (data/data_util.py) and an
    ``ItemPVDataset``-like object over them: lists of lists, sets, (user, product) tuples.  User activity is
    heavy-tailed so that histories shorter than, around and far above ``uprev_review_limit`` all occur."""
    rng = rng_for(seed)
    V = vocab_size
    review_u_p, u_r_seq, review_loc_time = [], [], []
    n_rev_u = np.minimum(max_reviews_per_user, rng.geometric(0.06, size=n_users))
    n_rev_u[0] = max_reviews_per_user                   # one very active user, one with a single review
    n_rev_u[1] = 1
    for u in range(n_users):
        seq = []
        for loc in range(int(n_rev_u[u])):
            r = len(review_u_p)
            review_u_p.append((u, int(rng.integers(0, n_products))))
            review_loc_time.append((loc, 0, r))
            seq.append(r)
        u_r_seq.append(seq)
    n_reviews = len(review_u_p)
    is_train = rng.random(n_reviews) < train_frac
    u_reviews = [set(r for r in seq if is_train[r]) for seq in u_r_seq]
    query_words = []
    for _ in range(n_queries):
        n = int(rng.integers(1, Q + 1))
        query_words.append([int(x) for x in rng.integers(0, V - 1, size=n)] + [V - 1] * (Q - n))
    product_query_idx = [[int(x) for x in rng.integers(0, n_queries, size=int(rng.integers(1, 4)))]
                         for _ in range(n_products)]
    gd = _NS(product_size=n_products, vocab_size=V, review_u_p=review_u_p, u_r_seq=u_r_seq,
             review_loc_time=review_loc_time, query_words=query_words, user_ids=['u%d' % u for u in range(n_users)])
    train_pd = _NS(set_name='train', u_reviews=u_reviews, product_query_idx=product_query_idx)
    test_pd = _NS(set_name='test', u_reviews=u_reviews, product_query_idx=product_query_idx)
    train_data = []
    for r in np.flatnonzero(is_train):
        for _ in range(int(rng.integers(1, 3))):
            words = [int(x) for x in rng.integers(0, V - 1, size=W)]
            train_data.append([words, int(r)])
    test_data = []
    for r in np.flatnonzero(~is_train)[:200]:
        u, p = review_u_p[r]
        q = product_query_idx[p][0]
        n_c = int(rng.integers(3, 12))
        cands = [int(x) for x in rng.integers(0, n_products, size=n_c - 1)] + [p]
        test_data.append([q, u, p, int(r), cands])

    class _Dataset(object):
        def __init__(self, pd, data):
            self.prod_pad_idx, self.word_pad_idx, self.seg_pad_idx = n_products, V - 1, 3
            self.global_data, self.prod_data, self._data = gd, pd, data

        def __len__(self):
            return len(self._data)

        def __getitem__(self, i):
            return self._data[i]

    return _Dataset(train_pd, train_data), _Dataset(test_pd, test_data)


# ------------------------------------------------------------------ synthetic corpus on disk (reference file formats)
def write_corpus(root, seed, n_users=40, n_products=60, n_words=300, n_queries=25, max_reviews=14, split_dir='split'):
    """Write a small Amazon-shaped corpus in the gz text formats ``data/data_util.py:166-287`` reads
    (product/users/vocab/review_text/u_r_seq/p_r_seq/review_uloc_ploc_and_time/review_id/review_u_p in ``root``;
    query/train/train_id/valid_id/test_id/train_query_idx/test_query_idx in ``root/split``).  Returns
    (data_path, input_train_dir)."""
    import gzip
    import os
    rng = rng_for(seed)
    inp = os.path.join(root, split_dir)
    os.makedirs(inp, exist_ok=True)

    def put(path, lines):
        with gzip.open(path, 'wt') as f:
            for ln in lines:
                f.write(ln + '\n')

    put(os.path.join(root, 'product.txt.gz'), ['B%06d' % i for i in range(n_products)])
    put(os.path.join(root, 'users.txt.gz'), ['A%05d' % i for i in range(n_users)])
    put(os.path.join(root, 'vocab.txt.gz'), ['w%d' % i for i in range(n_words)])
    queries = [[int(x) for x in rng.integers(0, n_words, size=int(rng.integers(1, 6)))] for _ in range(n_queries)]
    put(os.path.join(inp, 'query.txt.gz'), [' '.join(map(str, q)) for q in queries])
    pq = [[int(x) for x in rng.integers(0, n_queries, size=int(rng.integers(1, 4)))] for _ in range(n_products)]
    put(os.path.join(inp, 'train_query_idx.txt.gz'), [' '.join(map(str, q)) for q in pq])
    put(os.path.join(inp, 'test_query_idx.txt.gz'), [' '.join(map(str, q)) for q in pq])
    # reviews: (time, user, product, text); the review index is the line number in review_text
    events = []
    for u in range(n_users):
        for _ in range(int(min(max_reviews, 2 + rng.geometric(0.25)))):
            events.append((float(rng.random()), u, int(rng.integers(0, n_products))))
    order = rng.permutation(len(events))                      # file order is unrelated to time order
    events = [events[i] for i in order]
    n_rev = len(events)
    text = [[int(x) for x in rng.integers(0, n_words, size=int(rng.integers(3, 25)))] for _ in range(n_rev)]
    by_time = sorted(range(n_rev), key=lambda r: events[r][0])
    u_seq = [[] for _ in range(n_users)]
    p_seq = [[] for _ in range(n_products)]
    loc = [None] * n_rev
    for rank, r in enumerate(by_time):
        _, u, p = events[r]
        loc[r] = (len(u_seq[u]), len(p_seq[p]), rank)
        u_seq[u].append(r)
        p_seq[p].append(r)
    orig = rng.permutation(n_rev * 3)[:n_rev]                  # original line ids: sparse and shuffled
    put(os.path.join(root, 'review_text.txt.gz'), [' '.join(map(str, t)) for t in text])
    put(os.path.join(root, 'u_r_seq.txt.gz'), [' '.join(map(str, s)) for s in u_seq])
    put(os.path.join(root, 'p_r_seq.txt.gz'), [' '.join(map(str, s)) for s in p_seq])
    put(os.path.join(root, 'review_uloc_ploc_and_time.txt.gz'), ['%d %d %d' % loc[r] for r in range(n_rev)])
    put(os.path.join(root, 'review_id.txt.gz'), ['line_%d' % orig[r] for r in range(n_rev)])
    put(os.path.join(root, 'review_u_p.txt.gz'), ['%d %d' % (events[r][1], events[r][2]) for r in range(n_rev)])
    train, valid, test = [], [], []
    for u in range(n_users):
        s = u_seq[u]
        test.append(s[-1])
        if len(s) >= 3:
            valid.append(s[-2])
            train += s[:-2]
        else:
            train += s[:-1]
    ident = lambda r: '%d\t%d\tline_%d' % (events[r][1], events[r][2], orig[r])
    put(os.path.join(inp, 'train_id.txt.gz'), [ident(r) for r in train])
    put(os.path.join(inp, 'train.txt.gz'), ['%d\t%d\t%s' % (events[r][1], events[r][2], ' '.join(map(str, text[r])))
                                            for r in train])
    for name, rows in (('valid', valid), ('test', test)):
        put(os.path.join(inp, '%s_id.txt.gz' % name),
            [ident(r) + '\t%d' % pq[events[r][2]][int(rng.integers(0, len(pq[events[r][2]])))] for r in rows])
    return root, inp
