#!/usr/bin/env python3
"""The embedding-gather+score launch at the HBM-bound shape of BASELINE configs[4] (C5): d=256 (1 KiB rows),
a 50 M-row item table (51 GB fp32, far beyond the 256 MB Infinity Cache), B=1024, K=20, W=1.
At C2 the same launch moves only 12.8 MB (1.6 us at HBM peak) and is bounded by two dependent memory round
trips; this is the size at which the HBM roofline is the binding limit.

    python tools/gather_c5.py [--rows 50000000] [--iters 50]
Prints one JSON line with algorithmic GB/s and the fraction of the 8 TB/s HBM peak.
"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from prodsearch_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=50_000_000)
    ap.add_argument('--vocab', type=int, default=2_000_000)
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--batch', type=int, default=1024)
    ap.add_argument('--w', type=int, default=1, help='pv_window_size (0: item tasks only)')
    ap.add_argument('--sets', type=int, default=8, help='index sets the launches rotate through (1: the same rows every launch = Infinity-Cache hits)')
    a = ap.parse_args()
    lib = _lib.load()
    d, B, K, W, P, V = 256, a.batch, 20, a.w, a.rows, a.vocab
    dev = 'cuda'
    gen = torch.Generator(device=dev).manual_seed(1)
    table = torch.empty(P + 1, d, device=dev)
    for i in range(0, P + 1, 1 << 22):                      # fill in slices: no 51 GB temporary
        table[i:i + (1 << 22)].normal_(generator=gen)
    words = torch.randn(V, d, device=dev, generator=gen)
    wbias = torch.zeros(V, device=dev)
    desc = _lib.PsTemDesc()
    desc.B, desc.K, desc.L, desc.Q, desc.W, desc.C = B, K, 20, 8, W, 0
    desc.d, desc.H, desc.F, desc.n_layers = d, 8, 1024, 1
    desc.product_size, desc.vocab_size = P, V
    desc.use_pos_emb, desc.training, desc.dropout = 1, 1, 0.1
    lay = _lib.PsTemWsLayout()
    _lib.check(lib.ps_tem_workspace_layout(desc, lay), 'layout')
    ws = torch.randn(lay.total_floats, device=dev)
    params = _lib.PsTemTensors()
    params.product_emb, params.word_emb, params.word_bias = table.data_ptr(), words.data_ptr(), wbias.data_ptr()
    mk = lambda hi, *shape: torch.randint(0, hi, shape, device=dev, dtype=torch.int64, generator=gen)
    sets = []
    for _ in range(max(1, a.sets)):
        idx = (mk(P, B), mk(P, B, K), mk(V - 1, B, max(W, 1)), mk(V - 1, B, max(W, 1) * K))
        bt = _lib.PsTemBatch()
        bt.target_prod_idxs, bt.neg_item_idxs = idx[0].data_ptr(), idx[1].data_ptr()
        bt.pos_iword_idxs, bt.neg_word_idxs = idx[2].data_ptr(), idx[3].data_ptr()
        sets.append((idx, bt))
    st = torch.cuda.current_stream()
    call = lambda i=0: lib.ps_gather_score(desc, params, sets[i % len(sets)][1], ws.data_ptr(), st.cuda_stream)
    for i in range(8):
        _lib.check(call(i), 'ps_gather_score')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(st)
    for i in range(a.iters):
        call(i)
    e1.record(st)
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / a.iters
    R = lay.R
    rows = B * (1 + K) * (1 + W)
    nbytes = rows * (4 * d + 8) + (B * R + B) * 4 * d + rows * 4
    print(json.dumps({"workload": "gather+score launch, C5 shape: d=256, %d-row item table (%.1f GB), B=%d, K=%d, R=%d"
                      % (P, (P + 1) * d * 4 / 1e9, B, K, R), "index_sets": len(sets), "env": {k: v for k, v in os.environ.items() if k.startswith('PS_SCORE')}, "us_per_launch": t * 1e6, "bytes_per_launch": nbytes,
                      "achieved_GBps": nbytes / t / 1e9, "frac_of_8TBps": nbytes / t / 8e12}))


if __name__ == '__main__':
    main()
