#!/usr/bin/env python3
"""bench.py — train (u,q,i,neg) tuples/sec of the ranking-loss step on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c2|c4|c5]

N > 1: either the driver's form (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ..., one rank
per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or plain `python bench.py --gpus N`: with no
WORLD_SIZE in the environment the script starts the N ranks itself as child processes BEFORE anything touches the GPU
and relays rank 0's line; it exits non-zero — it never prints an `n_gpus: 1` line — if N ranks cannot be started.

Workloads (BASELINE.json `configs`):
  c2 (default, configs[1], the config the metric is quoted on): item_transformer, d=128, 1 layer, 8 heads, ff 512,
     uprev 20, B=384 per GPU, 20 negatives, Q=8, W=1, P=18,357 items, V=32,387 words, README training
     flags (README.md:13-25) incl. the reference's default dropout 0.1 (main.py:60) => the K+1 encoder replicas are
     really computed.
  c4 (configs[3]): review_transformer (RTM) d=128, bs=256, K=5, 20+30 reviews of 100 words, pvc review encoder,
     dropout 0.1, corrupt_rate 0.9 (reference defaults).
  c5 (one GPU's shard of configs[4]): item_transformer d=256, ff 1024, bs=1024, 50 M-item table, row-sparse Adam.
One step = trainer.py:74-78: loss = model(batch); model.zero_grad(); loss.backward(); optim.step() — sampling,
forward, backward, gradient exchange (N>1), clip+Adam.  Inputs are synthetic and already resident in HBM.

Timing: `value` / `ms_per_step` = EXACTLY --steps steps between barrier + synchronize on both sides (max over ranks).
Then, outside that region (rank 0 keeps the numbers): `median_ms_per_step` over --reps steps timed one by one with HIP
events, and
  roofline     — the workload's dominant kernel: algorithmic flops / bytes per launch ÷ its average IN-STEP duration,
                 measured live with one HIP event pair around every launch of it, on the stream it is launched on, over a
                 second pass of --steps steps (ps_ktimer_arm / ps_ktimer_read, include/prodsearch_hip.h)
  roofline_longest_kernel — c2: the same for the step's LONGEST kernel by rocprofv3 duration, the grouped W2 / W1 / Wo weight
                 gradients on the side stream (the fused forward of `roofline` is the longest on the critical path)
  cpu_baseline — the oracle (op-for-op CPU restatement; for c2 incl. the B*(K+1) replicated encoder) timed on this box's
                 host cores on a bounded sample: pools of 8, 32 and all hardware threads (rank 0, N=1)
and, in the default c2 run at N=1 (bounded to about a minute; --no-also skips them):
  roofline_hbm — the embedding-gather+score launch ALONE at the HBM-bound shape of configs[4] (d=256, 8 M-row item
                 table = 8.2 GB, B=1024 and B=8192; index sets rotate so that no row is re-read out of the 256 MB
                 Infinity Cache), event-timed per launch: the kernel the north star's 70 % target names
  also         — the contract lines of c4 and of the c5 shard (the stated 50 M-row table: 205 GB resident), each with its own
                 roofline object.
  fed_by_loader — the c2 step with its batches built by the data loader (native collate thread + copy stream) instead of resident
                 in HBM: wall clock per step, data loading included (informational; `value` is the resident-input figure).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')      # kernel arguments in device memory (prodsearch_amd/__init__.py)
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 MFMA (v_mfma_f32_32x32x2_f32), = the fp32 vector peak
MFMA_BF16_PEAK_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense bf16 MFMA; a bf16x3 product = 6 bf16 MFMAs per fp32-grade step
V_WORDS = 32387
# TEM workloads: c2 = BASELINE configs[1]; c5 = one GPU's shard of configs[4] (50 M-item table, d=256, bs=1024/GPU,
# SURVEY.md §8d C5): 51 GB table + dense gradient + Adam moments = 205 GB of HBM, row-sparse optimizer
TEM_CFG = {
    'c2': dict(P=18357, B=384, K=20, L=20, Q=8, W=1, D=128, FF=512, row_sparse=False, config_index=1),
    'c5': dict(P=50_000_000, B=1024, K=20, L=20, Q=8, W=1, D=256, FF=1024, row_sparse=True, config_index=4),
}
C4 = dict(RC=296000, B=256, K=5, WL=100, U=20, I=30)      # BASELINE configs[3]
# HBM traffic of ONE gather+score launch at B=1024, d=256 from the committed counter passes (PMC cannot run inside this process)
GATHER_TRAFFIC = 70.4e6
GATHER_TRAFFIC_SOURCE = ("committed PMC passes profiles/r05_gather_score_c5_pmc.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over "
                         "tools/gather_c5.py --rows 8000000 --batch 1024, 8 rotating index sets): FETCH_SIZE 35.0 MB x2 (gfx950 16-B/lane streaming-read "
                         "correction, MI355X_MICROARCH.md HBM) + WRITE_SIZE 0.42 MB = 70.4 MB per launch against 67.6 MB algorithmic (1.04x: nothing is "
                         "re-read; rounds 2-3 with ONE index set: 62.8 MB, part of the rows out of the Infinity Cache)")
# memory-side bytes per launch of the two C2 roofline kernels from the committed counter passes (profiles/r05_c2_pmc_traffic.txt:
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 20 --warmup 5 --no-extras`; FETCH_SIZE x2 — the
# gfx950 16-B/lane streaming-read correction, MI355X_MICROARCH.md HBM; both counters sit at the L2's fabric side and include
# Infinity-Cache hits: at C2 the working set lives in that cache, so these are L2-miss bytes)
C2_TRAFFIC = {'mlp_fwd': 2 * 7.83e6 + 48.8e6, 'wgrad_group': 2 * 25.9e6 + 12.3e6}
C2_TRAFFIC_SOURCE = ("committed PMC passes profiles/r05_c2_pmc_traffic.txt (FETCH_SIZE x2 + WRITE_SIZE per launch, L2 fabric side, "
                     "Infinity-Cache hits included): mlp_fwd 2 x 7.83 + 48.8 MB (a1 + h1 + y1 / ln1 / y2 / enc written once, nothing "
                     "re-read); grouped weight gradients 2 x 25.9 + 12.3 MB for 49.5 MB of operands + 10.5 MB of fp32 atomics "
                     "(143 MB before the XCD-aware split placement)")
PREWARM_TOTAL = 100                                        # untimed steps in front of the timed region, the W warm-up steps included
ALSO_C5_ITEMS = 50_000_000                                 # the c5 line inside the default run: the STATED table of configs[4] (205 GB resident)
GATHER_LEG_ITEMS = 8_000_000                               # the stand-alone gather+score leg: an 8.2 GB table is 32x the Infinity Cache already


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--reps', type=int, default=200, help='steps timed one by one with HIP events for the median (0 = skip)')
    ap.add_argument('--dropout', type=float, default=0.1)
    ap.add_argument('--cpu-steps', type=int, default=3, help='CPU-baseline steps per thread pool (0 = skip)')
    ap.add_argument('--no-extras', action='store_true', help='skip median / roofline / cpu_baseline legs')
    ap.add_argument('--no-also', action='store_true', help='default c2 run: skip the roofline_hbm / also legs')
    ap.add_argument('--workload', default='c2', choices=['c2', 'c4', 'c5'],
                    help='c2 = BASELINE configs[1] (the metric); c4 = configs[3] (RTM); c5 = per-GPU shard of configs[4]')
    ap.add_argument('--encoder', default='pvc', choices=['pvc', 'pv'], help='c4: review encoder')
    ap.add_argument('--items', type=int, default=0, help='override the catalogue size (c5 dry runs)')
    ap.add_argument('--sharded', action='store_true',
                    help='item table sharded by row over the ranks (args.shard_tables; implies --row-sparse): the N4 path')
    ap.add_argument('--row-sparse', action='store_true',
                    help='touched-rows-only zero/clip/Adam/exchange (args.row_sparse_adam); always on for c5')
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------ workloads
class TemWorkload(object):
    """item_transformer (c2 / c5)."""

    def __init__(self, a, name, rank, dev, items=0):
        from prodsearch_amd import ItemTransformerRanker, build_optim, readme_tem_args, synth
        self.a, self.name = a, name
        c = self.c = dict(TEM_CFG[name])
        if items:
            c['P'] = items
        c['row_sparse'] = c['row_sparse'] or a.row_sparse
        self.ns = readme_tem_args(dropout=a.dropout, embedding_size=c['D'], ff_size=c['FF'], batch_size=c['B'],
                                  row_sparse_adam=c['row_sparse'] or a.sharded, shard_tables=a.sharded)
        self.wd = synth.make_word_dists(V_WORDS)
        torch.manual_seed(1234)                     # identical init on every rank
        self.model = ItemTransformerRanker(self.ns, 'cuda', V_WORDS, c['P'], None, word_dists=self.wd)
        self.optim = build_optim(self.ns, self.model, None)
        self.batches = [synth.make_tem_batch(1000 + 97 * rank + i, c['B'], c['P'], V_WORDS, Q=c['Q'], L=c['L'], W=c['W'],
                                             word_dists=self.wd).to(dev) for i in range(8)]
        self.B, self.K = c['B'], c['K']

    def forward(self, i):
        return self.model(self.batches[i % len(self.batches)])      # trainer.py:74 (negatives sampled on device)

    def describe(self):
        c = self.c
        R = next(iter(self.model._plans.values())).layout.R
        return ("item_transformer d=%d 1 layer 8 heads ff=%d uprev=20 bs=%d/GPU 20 neg Q=8 W=1 P=%d V=32387 dropout=%.2f%s "
                "(BASELINE configs[%d])" % (c['D'], c['FF'], c['B'], c['P'], self.a.dropout,
                                            (" row-sharded item table + row-sparse Adam" if self.a.sharded else
                                             " row-sparse Adam" if c['row_sparse'] else ""), c['config_index'])), {"replicas_per_row": R}

    def metric(self):
        return "train (u,q,i,neg) tuples/sec at bs=%d, 20 neg, d=%d" % (self.c['B'], self.c['D'])

    def dtype(self):
        from prodsearch_amd import _lib
        return _lib.load().ps_arith_info().decode()

    def roofline_spec(self):
        """The step's dominant kernel.  With replicas at d = 128 (c2) that is the fused per-replica forward (Wo -> LN -> W1
        -> GELU -> W2 -> LN for every one of the B*(K+1) replica rows, with the item gather + score + loss in its
        epilogue): MFMA-bound, 2*d*d + 4*d*F flops per row.  Otherwise (c5: d = 256) the stand-alone gather+score launch:
        HBM-bound, every table row once (4d B) + its int64 index, every distinct vector it is dotted with, every score
        written."""
        c = self.c
        B, K, W, D, FF = c['B'], c['K'], c['W'], c['D'], c['FF']
        R = next(iter(self.model._plans.values())).layout.R
        rows = B * (1 + K) * (1 + W)
        vecs = B * R + B                            # encoder outputs + target-item rows
        gather_bytes = rows * (4 * D + 8) + vecs * 4 * D + rows * 4
        if D == 128 and R > 1 and (B * R + 31) // 32 <= 256:
            flops = B * R * (2 * D * D + 4 * D * FF)
            return dict(tag='mlp_fwd', bound='mfma', work=flops, peak=MFMA_F32_PEAK_TFLOPS, unit='TFLOP/s', scale=1e12,
                        kernel="fused per-replica encoder tail (mlp_fwd), %d rows x (2*%d*%d + 4*%d*%d) flop; item gather + "
                               "score + loss folded into its epilogue: %d B of gathered rows and scores"
                               % (B * R, D, D, D, FF, B * (1 + K) * (4 * D + 8 + 4)),
                        extra={"gather_score_bytes_all_tasks": gather_bytes,
                               "peak_note": "peak = dense fp32 MFMA; products issued as fp32-grade bf16x3 (exact 3-way bf16 split of "
                                            "both operands, 6 of the 9 cross products as bf16 MFMAs per step) have an equivalent "
                                            "peak of %.0f TFLOP/s" % (MFMA_BF16_PEAK_TFLOPS / 6.0),
                               "frac_of_bf16x3_equivalent_peak": None})
        return dict(tag='gather_score', bound='hbm', work=gather_bytes, peak=HBM_PEAK_GBS, unit='GB/s', scale=1e9,
                    kernel="score_fwd_sidx_kernel<%d,16> (embedding gather + score; 16 lanes per %d-B row, indices by scalar loads)" % (D // 64, 4 * D),
                    extra={"traffic": GATHER_TRAFFIC if (B, D) == (1024, 256) else None,
                           "traffic_source": GATHER_TRAFFIC_SOURCE if (B, D) == (1024, 256) else None})

    def roofline_longest_spec(self):
        """c2 only: by rocprofv3's durations the step's LONGEST kernel is not the fused forward but the grouped W2 / W1 / Wo
        weight gradients (one flat-group launch of gemm_x3_kernel<1,1,0,1,1,0,1>, 640 workgroups of 64x64 tiles, fp32 atomics
        into 0.59 MB of dW): the same 2*d*d + 4*d*F flops per replica row.  It runs on the side stream underneath the score
        scatter, attention backward, K/V dX product and history scatter of the main stream, so its in-step duration is that
        of a kernel sharing the chip four ways; reported beside `roofline` so that the line does not flatter the step."""
        c = self.c
        R = next(iter(self.model._plans.values())).layout.R
        if not (c['D'] == 128 and R > 1):
            return None
        rows = c['B'] * R
        return dict(tag='wgrad_group', work=rows * (2 * c['D'] * c['D'] + 4 * c['D'] * c['FF']),
                    kernel="grouped W2 / W1 / Wo weight gradients (gemm_x3_kernel<1,1,0,1,1,0,1>, flat group): %d reduction rows, "
                           "dW[%d x %d] + dW[%d x %d] + dW[%d x %d]; side stream, concurrent with the main stream's backward tail"
                           % (rows, c['D'], c['FF'], c['FF'], c['D'], c['D'], c['D']))

    def cpu_baseline(self, n_steps):
        return cpu_baseline_tem(self.ns, self.c, n_steps)


class RtmWorkload(object):
    """review_transformer (c4)."""

    def __init__(self, a, name, rank, dev, items=0):
        from prodsearch_amd import ProductRanker, build_optim, default_args, synth, rtm_data
        self.a, self.name = a, name
        c = C4
        self.ns = default_args(model_name='review_transformer', review_encoder_name=a.encoder, embedding_size=128,
                               heads=8, ff_size=512, inter_layers=1, neg_per_pos=c['K'], dropout=a.dropout,
                               corrupt_rate=0.9, lr=0.0005, review_word_limit=c['WL'], uprev_review_limit=c['U'],
                               iprev_review_limit=c['I'])
        self.wd = synth.make_word_dists(V_WORDS)
        rng = synth.rng_for(5)
        rw = torch.from_numpy(rng.integers(0, V_WORDS - 1, size=(c['RC'], c['WL'])))
        lens = torch.from_numpy(rng.integers(c['WL'] // 4, c['WL'] + 1, size=c['RC']))
        rw[torch.arange(c['WL'])[None, :] >= lens[:, None]] = V_WORDS - 1
        rw[-1] = V_WORDS - 1
        self.rw = rw
        torch.manual_seed(1234)
        self.model = ProductRanker(self.ns, 'cuda', V_WORDS, c['RC'], 1000, 1000, rw, None, word_dists=self.wd)
        self.optim = build_optim(self.ns, self.model, None)
        self.cpu_batches = [rtm_data.make_rtm_batch(100 + 97 * rank + s, c['B'], c['K'], c['RC'], V_WORDS, rw, Q=8,
                                                    u_lim=c['U'], i_lim=c['I'], W=1, train_pv=False, encoder=a.encoder,
                                                    word_dists=self.wd) for s in range(4)]
        self.batches = [b.to(dev) for b in self.cpu_batches]
        self.B, self.K = c['B'], c['K']

    def forward(self, i):
        return self.model(self.batches[i % len(self.batches)], train_pv=False)

    def describe(self):
        c = C4
        return ("review_transformer (RTM) d=128 1 layer 8 heads ff=512 bs=%d/GPU K=%d R=%d+%d WL=%d %s review encoder "
                "dropout=%.2f corrupt_rate=0.90 %dk reviews V=32387 (BASELINE configs[3])"
                % (c['B'], c['K'], c['U'], c['I'], c['WL'], self.a.encoder, self.a.dropout, c['RC'] // 1000)), {}

    def metric(self):
        return "train (u,q,i,neg) tuples/sec at bs=%d, %d neg, d=128 (review_transformer)" % (C4['B'], C4['K'])

    def dtype(self):
        return TemWorkload.dtype(self)

    def roofline_spec(self):
        """The review-vector gather (PVC.py:46-61): per valid review slot its word ids, per word that is neither padding nor
        dropped by the token corruption one 4d-byte row (expected count: the Philox masks are drawn on the device), the
        rows of the encoder input it writes and what it leaves for the backward's inverted index."""
        c, d = C4, 128
        b0 = self.cpu_batches[0]
        pad_r = c['RC'] - 1
        slots = int((b0.pos_prod_ridxs != pad_r).sum() + (b0.neg_prod_ridxs != pad_r).sum())
        nseq = c['B'] * (c['K'] + 1)
        out_bytes = (slots + nseq) * d * 4              # x: the real positions + one query row per sequence (padded
        if self.a.encoder == 'pvc':                     # positions are neither read nor written: DESIGN.md 7c)
            words = int((b0.pos_prod_rword_idxs != V_WORDS - 1).sum() + (b0.neg_prod_rword_idxs != V_WORDS - 1).sum())
            rows = words * (1.0 - 0.9)
            # PS_RTM_HIST=0 (the round-2 form): the first pass of the backward's inverted index rides in this kernel and writes a
            # rank per word slot; the default builds the whole index in the backward (rtm_hist_kernel), the gather writes none
            legacy = os.environ.get('PS_RTM_HIST', '1') == '0'
            rank_bytes = slots * c['WL'] * 4 if legacy else 0
            nbytes = slots * c['WL'] * 8 + rows * 4 * d + out_bytes + rank_bytes
            note = "%d review slots x %d ids + %.0f surviving word rows (%.0f non-pad words x 0.1) + %d B of x%s" % (
                slots, c['WL'], rows, words, out_bytes, (" + %d B of word ranks" % rank_bytes) if legacy else "")
        else:
            nbytes = slots * (8 + 4 * d) + out_bytes
            note = "%d review rows + %d B of x" % (slots, out_bytes)
        return dict(tag='rtm_embed', bound='hbm', work=int(nbytes), peak=HBM_PEAK_GBS, unit='GB/s', scale=1e9,
                    kernel="rtm_embed4_kernel (review-vector gather + mean-pool; %s)" % note,
                    extra={"traffic": 79.3e6 if self.a.encoder == 'pvc' else None,
                           "traffic_source": "committed PMC passes profiles/r05_c4_pmc_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate "
                                             "passes over bench.py --workload c4): FETCH_SIZE 34.4 MB x2 + WRITE_SIZE 10.6 MB per launch of "
                                             "rtm_embed4_kernel (round 2, with the rank atomics in this kernel: 123.3 MB)" if self.a.encoder == 'pvc' else None,
                           "bound_note": "the 59 MB of word rows come from a 16.6 MB table that lives in the L2s / Infinity Cache; "
                                         "what the kernel takes from HBM is the ids and what it writes"})

    def cpu_baseline(self, n_steps):
        return cpu_baseline_rtm(self, n_steps)


# ------------------------------------------------------------------------------------------ CPU baselines
def _pools(one_step, ncpu, n_steps, cands=(8, 32, None)):
    """torch's intra-op pool does not scale to every hardware thread of a 2-socket host on these small ops, so the same
    step is timed on a few pool sizes — 8 (the survey probe's), 32 and ALL hardware threads — `n_steps` steps each after one
    warm-up; the fastest pool is the reported baseline and every pool's rate is listed beside it."""
    cands = sorted({min(ncpu, c if c else ncpu) for c in cands})
    before = torch.get_num_threads()
    torch.set_num_threads(cands[0])
    one_step()                                   # warm-up (allocator, first-touch)
    per = {}
    for c in cands:
        torch.set_num_threads(c)
        # (the all-threads pool oversubscribes these small ops ~15x on a 256-thread host: one step there bounds the run)
        ts = [one_step() for _ in range(n_steps if c <= 64 else 1)]
        per[c] = sum(ts) / len(ts)
    best = min(per, key=per.get)
    torch.set_num_threads(before)                # a 256-thread pool left behind slows the HOST side of the GPU steps that follow
    return best, per


def cpu_baseline_tem(args_ns, c, n_steps):
    """The oracle as the CPU path: same shapes/flags, reference structure (replicated encoder,
    torch RNG dropout), fwd + bwd + clip/Adam on the host cores."""
    from oracle import tem as otem, optim as ooptim
    from prodsearch_amd import synth
    ncpu = os.cpu_count() or 1
    P, B, K, L, Q, W, D = c['P'], c['B'], c['K'], c['L'], c['Q'], c['W'], c['D']
    wd = synth.make_word_dists(V_WORDS)
    shapes = synth.tem_param_shapes(args_ns, V_WORDS, P)
    Pm = {k: v.requires_grad_(True) for k, v in synth.make_state_dict(shapes, 1, {'product_emb.weight': P}).items()}
    opt = ooptim.ClipAdam(args_ns.lr, args_ns.max_grad_norm, args_ns.beta1, args_ns.beta2, 1e-9, args_ns.l2_lambda)
    pad = otem.tem_pad_rows(args_ns, V_WORDS, P)
    drop = otem.TorchDropout(args_ns.dropout) if args_ns.dropout > 0 else None
    counter = [0]

    def one_step():
        s = counter[0]
        counter[0] += 1
        batch = synth.make_tem_batch(100 + s, B, P, V_WORDS, Q=Q, L=L, W=W, word_dists=wd)
        ni, nw = synth.sample_negatives(200 + s, B, K, W, P, wd)
        t0 = time.perf_counter()
        loss, _, _ = otem.tem_forward(Pm, args_ns, batch, ni, nw, V_WORDS, P, training=True,
                                      replicate=True, drop=drop)
        grads = otem.grads_of(loss, Pm, pad)
        with torch.no_grad():
            opt.step(Pm, grads)
        return time.perf_counter() - t0

    best, per = _pools(one_step, ncpu, n_steps)
    t = per[best]
    return {"value": B * K / t, "unit": "tuples/s", "cores": best, "kind": "port",
            "pools_tuples_per_s": {str(k): B * K / v for k, v in sorted(per.items())},
            "sample": "%d steps per pool of the same B=%d,K=%d,d=%d step (oracle/tem.py: fwd+bwd+clip/Adam, replicated encoder, "
                      "dropout %.2f) on pools of %s of %d host threads after 1 warm-up step; fastest: %.2f s/step on %d threads"
                      % (n_steps, B, K, D, args_ns.dropout, "/".join(str(k) for k in sorted(per)), ncpu, t, best)}


def cpu_baseline_rtm(wl, n_steps):
    """oracle/rtm.py as the CPU path of configs[3]: the same batches and flags, torch RNG dropout and token corruption,
    fwd + autograd bwd + clip/Adam on the host cores."""
    from oracle import rtm as ortm, tem as otem, optim as ooptim
    ncpu = os.cpu_count() or 1
    ns, c = wl.ns, C4
    Pm = {}
    for k, v in wl.model.state_dict().items():
        if k.endswith('pos_emb.pe') or k.startswith('review_encoder.'):      # aliases of word_embeddings / buffers
            if k != 'review_encoder.review_embeddings.weight':
                continue
        Pm[k] = v.detach().cpu().clone().requires_grad_(True)
    opt = ooptim.ClipAdam(ns.lr, ns.max_grad_norm, ns.beta1, ns.beta2, 1e-9, ns.l2_lambda)
    drop = otem.TorchDropout(ns.dropout) if ns.dropout > 0 else None
    p = float(ns.corrupt_rate)
    tok = (lambda shape, which: (torch.rand(shape) >= p).float() / (1.0 - p)) if wl.a.encoder == 'pvc' else None
    counter = [0]

    def one_step():
        s = counter[0]
        counter[0] += 1
        batch = wl.cpu_batches[s % len(wl.cpu_batches)]
        t0 = time.perf_counter()
        loss, _, _ = ortm.rtm_forward(Pm, ns, batch, None, V_WORDS, c['RC'], training=True, train_pv=False, drop=drop,
                                      tok_drop=tok)
        names = [n for n in Pm]
        gs = torch.autograd.grad(loss, [Pm[n] for n in names], allow_unused=True)
        with torch.no_grad():
            opt.step(Pm, dict(zip(names, gs)))
        return time.perf_counter() - t0

    best, per = _pools(one_step, ncpu, n_steps, cands=(8, 32))
    t = per[best]
    return {"value": c['B'] * c['K'] / t, "unit": "tuples/s", "cores": best, "kind": "port",
            "pools_tuples_per_s": {str(k): c['B'] * c['K'] / v for k, v in sorted(per.items())},
            "sample": "%d steps per pool of the same B=%d,K=%d RTM step (oracle/rtm.py: fwd + autograd bwd + clip/Adam, dropout "
                      "%.2f, token corruption 0.9) on pools of %s of %d host threads after 1 warm-up step; fastest: %.2f s/step on "
                      "%d threads" % (n_steps, c['B'], c['K'], ns.dropout, "/".join(str(k) for k in sorted(per)), ncpu, t, best)}


# ------------------------------------------------------------------------------------------ one workload
def measure(a, name, rank, world, dev, steps, warmup, reps, extras, cpu_steps, items=0):
    """Build the workload, time EXACTLY `steps` steps (barrier + synchronize on both sides, max over ranks), then the
    extra legs; returns the contract dict."""
    from prodsearch_amd import _lib, dist as pdist
    wl = (RtmWorkload if name == 'c4' else TemWorkload)(a, name, rank, dev, items)
    model, optim = wl.model, wl.optim
    model._seed = pdist.rank_seed(wl.ns.seed, rank)
    pdist.broadcast_parameters(model)
    exchange = pdist.make_exchange(model, optim)
    model.train()

    def step(i):
        loss = wl.forward(i)                             # trainer.py:74
        model.zero_grad()                                # :76
        loss.backward()                                  # :77
        exchange()                                       # RCCL exchange of the flat gradient (N>1)
        optim.step()                                     # :78
        return loss

    # Untimed, BEFORE the W warm-up steps the contract asks for: with a small W (the driver runs --warmup 5 --steps 20) the timed
    # region would otherwise start on a GPU that has run for ~2 ms — plans built a moment ago, clocks not yet raised (round 3:
    # 0.245 ms/step in the driver's 20 steps against 0.2305 over 300).  Reported as `prewarm_steps`; the timed region is still
    # EXACTLY `steps` steps after `warmup` steps.
    prewarm = max(0, PREWARM_TOTAL - warmup)
    for i in range(prewarm):
        step(i)
    for i in range(warmup):
        step(i)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed = pdist.max_over_ranks(time.perf_counter() - t0, dev)
    last_loss = float(loss.detach())

    desc, extra_cfg = wl.describe()
    Bw, Kw = wl.B, wl.K
    xname = type(exchange).__name__
    xdesc = {"GradExchange": "flat all-reduce", "ShardedAdamExchange": "reduce-scatter + owner clip/Adam + all-gather",
             "SparseGradExchange": "row-sparse all-gather exchange"}.get(xname, xname)
    if getattr(a, 'sharded', False):
        xdesc += " + all-to-all of the sharded item table's rows and gradients"
    out = {
        "metric": wl.metric(),
        "value": world * Bw * Kw * steps / elapsed, "unit": "tuples/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "prewarm_steps": prewarm,
        "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": wl.dtype(), "data": "synthetic",
        "config": dict({"workload": desc, "global_batch": world * Bw, "parallelism": "dp%d" % world,
                        "step": "sample+fwd+bwd+%sclip/Adam via nn.Module API (trainer.py:74-78)"
                                % ((xdesc + "+") if world > 1 else "")}, **extra_cfg),
        "samples_per_s": world * Bw * steps / elapsed, "final_loss": last_loss,
        # N > 1, sharded optimizer: the forms of its two collectives it timed on this node and kept (dist.ShardedAdamExchange._tune)
        "exchange_tuning": getattr(exchange, 'tuned', None),
    }
    if world > 1:
        # comm / compute split (every rank runs it; rank 0 keeps the numbers): a HIP event pair around every collective of the
        # step on the stream it is issued from (dist.comm_timing) and around each whole step, over a further `steps` steps
        pdist.comm_timing(True)
        stc = torch.cuda.current_stream()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for i, (e0, e1) in enumerate(evs):
            e0.record(stc)
            step(i)
            e1.record(stc)
        spans = pdist.comm_timing_read()
        pdist.comm_timing(False)
        tot = sum(e0.elapsed_time(e1) for e0, e1 in evs) / steps
        comm = {k: v / steps for k, v in spans.items()}
        out["step_ms_event_timed"] = tot
        out["comm_ms"] = sum(comm.values())
        out["compute_ms"] = tot - out["comm_ms"]
        out["comm_breakdown_ms"] = comm
        out["comm_note"] = ("per step, rank 0, HIP events on the step's stream around every collective call (the stream waits for the "
                            "collective when the call returns); compute_ms = event-timed step - comm_ms (includes pack / merge kernels)")
    if extras:
        # Every rank runs these extra steps (they contain the gradient exchange: a collective only rank 0 entered would
        # never return); rank 0 keeps the numbers.
        lib = _lib.load()
        if reps > 0:     # (1) median of single steps, each between two HIP events on the step's stream
            st = torch.cuda.current_stream()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for i, (e0, e1) in enumerate(ev):
                e0.record(st)
                step(i)
                e1.record(st)
            torch.cuda.synchronize()
            ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
            out["median_ms_per_step"] = ts[len(ts) // 2]
            out["reps"] = reps
            out["p10_p90_ms_per_step"] = [ts[len(ts) // 10], ts[(9 * len(ts)) // 10]]
        # (2) roofline: in-step duration of the workload's dominant kernel, one event pair per launch on its own stream
        spec = wl.roofline_spec()
        _lib.check(lib.ps_ktimer_arm(spec['tag'].encode(), steps), 'ps_ktimer_arm')
        for i in range(steps):
            step(i)
        avg, mn, cnt = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int32(0)
        _lib.check(lib.ps_ktimer_read(ctypes.byref(avg), ctypes.byref(mn), ctypes.byref(cnt)), 'ps_ktimer_read')
        t_k = avg.value * 1e-6
        ach = spec['work'] / t_k / spec['scale'] if t_k > 0 else 0.0
        extra = dict(spec['extra'])
        if 'frac_of_bf16x3_equivalent_peak' in extra:
            extra['frac_of_bf16x3_equivalent_peak'] = ach / (MFMA_BF16_PEAK_TFLOPS / 6.0)
        c2_shape = name == 'c2' and (wl.c['B'], wl.c['D'], wl.c['FF']) == (384, 128, 512)
        out["roofline"] = dict({"bound": spec['bound'], "achieved": ach, "peak": spec['peak'], "unit": spec['unit'],
                                "frac": ach / spec['peak'],
                                "traffic": C2_TRAFFIC.get(spec['tag']) if c2_shape else None,
                                "traffic_source": C2_TRAFFIC_SOURCE if c2_shape and spec['tag'] in C2_TRAFFIC else None,
                                "kernel": spec['kernel'],
                                ("flops_per_launch" if spec['bound'] == 'mfma' else "bytes_per_launch"): spec['work'],
                                "us_per_launch": avg.value, "us_per_launch_min": mn.value, "launches_timed": cnt.value,
                                "timing": "HIP event pair BOUND to every in-step launch of the kernel (hipExtLaunchKernelGGL start / stop "
                                          "events: the dispatch's own begin / end, what rocprofv3's kernel trace reports), on "
                                          "the launch stream, over a second pass of %d steps (ps_ktimer)" % steps}, **extra)
        spec2 = wl.roofline_longest_spec() if hasattr(wl, 'roofline_longest_spec') else None
        if spec2:        # (2b) the longest kernel of the step when that is not the one the critical path is made of
            _lib.check(lib.ps_ktimer_arm(spec2['tag'].encode(), steps), 'ps_ktimer_arm')
            for i in range(steps):
                step(i)
            _lib.check(lib.ps_ktimer_read(ctypes.byref(avg), ctypes.byref(mn), ctypes.byref(cnt)), 'ps_ktimer_read')
            if cnt.value > 0:
                ach2 = spec2['work'] / (avg.value * 1e-6) / 1e12
                out["roofline_longest_kernel"] = {
                    "bound": "mfma", "achieved": ach2, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach2 / MFMA_F32_PEAK_TFLOPS, "frac_of_bf16x3_equivalent_peak": ach2 / (MFMA_BF16_PEAK_TFLOPS / 6.0),
                    "traffic": C2_TRAFFIC.get(spec2['tag']) if c2_shape else None,
                    "traffic_source": C2_TRAFFIC_SOURCE if c2_shape else None,
                    "kernel": spec2['kernel'], "flops_per_launch": spec2['work'], "us_per_launch": avg.value,
                    "us_per_launch_min": mn.value, "launches_timed": cnt.value,
                    "timing": "HIP event pair bound to every in-step launch (the dispatch's own begin / end), on the SIDE stream it "
                              "runs on (ps_ktimer)"}
        if world == 1 and cpu_steps > 0 and name == 'c4':
            out["cpu_baseline"] = wl.cpu_baseline(cpu_steps)
        elif world == 1 and cpu_steps > 0 and name == 'c2':
            # deferred by the caller to the very end of the run (after the GPU legs of `also`): only ns / shapes are needed
            ns, cfg = wl.ns, dict(wl.c)
            out["_cpu_baseline_later"] = lambda: cpu_baseline_tem(ns, cfg, cpu_steps)
    del wl, model, optim, exchange
    torch.cuda.empty_cache()
    return out


def gather_score_hbm_leg(dev, rows=GATHER_LEG_ITEMS, iters=40):
    """The embedding-gather+score launch alone at the C5 shape (d=256, 1-KiB rows, `rows`-row item table far beyond the
    Infinity Cache), B=1024 and B=8192.  Every launch carries a HIP event pair bound to its dispatch (ps_ktimer); the
    launches walk 8 different index sets in turn (8 x 67.6 MB at B=1024 > the 256 MB Infinity Cache), so a row read by
    one launch is not served from on-die cache to the next."""
    from prodsearch_amd import _lib
    lib = _lib.load()
    d, K, W, P, V = 256, 20, 1, rows, 2_000_000
    gen = torch.Generator(device=dev).manual_seed(1)
    table = torch.empty(P + 1, d, device=dev)
    for i in range(0, P + 1, 1 << 22):                      # fill in slices: no table-sized temporary
        table[i:i + (1 << 22)].normal_(generator=gen)
    words = torch.randn(V, d, device=dev, generator=gen)
    wbias = torch.zeros(V, device=dev)
    res = []
    for B in (1024, 8192):
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
        for _ in range(8):
            idx = (mk(P, B), mk(P, B, K), mk(V - 1, B, W), mk(V - 1, B, W * K))
            bt = _lib.PsTemBatch()
            bt.target_prod_idxs, bt.neg_item_idxs = idx[0].data_ptr(), idx[1].data_ptr()
            bt.pos_iword_idxs, bt.neg_word_idxs = idx[2].data_ptr(), idx[3].data_ptr()
            sets.append((idx, bt))
        st = torch.cuda.current_stream()
        for i in range(8):
            _lib.check(lib.ps_gather_score(desc, params, sets[i][1], ws.data_ptr(), st.cuda_stream), 'ps_gather_score')
        torch.cuda.synchronize()
        _lib.check(lib.ps_ktimer_arm(b'gather_score', iters), 'ps_ktimer_arm')
        for i in range(iters):
            _lib.check(lib.ps_gather_score(desc, params, sets[i % 8][1], ws.data_ptr(), st.cuda_stream), 'ps_gather_score')
        avg, mn, cnt = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int32(0)
        _lib.check(lib.ps_ktimer_read(ctypes.byref(avg), ctypes.byref(mn), ctypes.byref(cnt)), 'ps_ktimer_read')
        # the same launches back to back between ONE event pair: launch THROUGHPUT (consecutive launches overlap head and tail),
        # reported beside the per-launch duration, never as it
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(st)
        for i in range(iters):
            lib.ps_gather_score(desc, params, sets[i % 8][1], ws.data_ptr(), st.cuda_stream)
        e1.record(st)
        torch.cuda.synchronize()
        loop_us = e0.elapsed_time(e1) * 1e3 / iters
        R = lay.R
        nrows = B * (1 + K) * (1 + W)
        nbytes = nrows * (4 * d + 8) + (B * R + B) * 4 * d + nrows * 4
        ach = nbytes / (avg.value * 1e-6) / 1e9
        res.append({"B": B, "bytes_per_launch": nbytes, "us_per_launch": avg.value, "us_per_launch_min": mn.value,
                    "launches_timed": cnt.value, "achieved": ach, "frac": ach / HBM_PEAK_GBS,
                    "back_to_back_us_per_launch": loop_us, "back_to_back_throughput_frac": nbytes / (loop_us * 1e-6) / 1e9 / HBM_PEAK_GBS})
        del ws, sets
    del table, words
    torch.cuda.empty_cache()
    return {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": res[0]["achieved"], "frac": res[0]["frac"],
            "kernel": "score_fwd_sidx_kernel<4,16> alone (embedding gather + score, index hop on the scalar path), C5 shape: d=256, %d-row item table "
                      "(%.1f GB), K=20, W=1, R=21; 8 rotating index sets" % (rows, (rows + 1) * d * 4 / 1e9),
            "by_batch": res, "traffic": GATHER_TRAFFIC, "traffic_source": GATHER_TRAFFIC_SOURCE,
            "timing": "%d launches per batch size; us_per_launch / achieved / frac: one HIP event pair BOUND to every launch "
                      "(hipExtLaunchKernelGGL start / stop events = the dispatch's own begin / end, what rocprofv3's kernel trace reports: "
                      "profiles/r05_gather_score_kernel_stats.csv); back_to_back_*: the same launches between ONE event pair — launch "
                      "throughput, consecutive launches overlapping head and tail, not a kernel duration" % iters}


# ------------------------------------------------------------------------------------------ launching N ranks
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(a, backend, extra_env):
    """One set of child ranks -> (rank exit codes, rank 0's JSON lines, tail of rank 0's stderr).  A rank that dies leaves its peers
    inside a collective that never returns: as soon as one rank has exited non-zero the others are killed."""
    import tempfile
    port = _free_port()
    procs = []
    out0, err0 = tempfile.TemporaryFile(mode='w+'), tempfile.TemporaryFile(mode='w+')
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0', PS_BENCH_CHILD='1', **extra_env)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL,
                                      stderr=err0 if r == 0 else None, text=True))
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs):
            break
        if any(rc not in (None, 0) for rc in rcs):
            time.sleep(2.0)                       # (let the failing rank's peers print what they have)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            rcs = [p.wait() for p in procs]
            break
        time.sleep(0.2)
    out0.seek(0)
    err0.seek(0)
    out_text, err_tail = out0.read(), err0.read()[-2000:]
    out0.close()
    err0.close()
    sys.stderr.write(err_tail)
    lines = [ln for ln in out_text.splitlines() if ln.startswith('{')]
    return rcs, lines, err_tail


def self_launch(a):
    """`python bench.py --gpus N` with no torchrun environment: start the N ranks as child processes of this one — nothing
    here has touched the GPU (device_count() does not initialise it) — relay rank 0's JSON line and fail loudly otherwise.

    The sharded optimizer's peer-to-peer forms (batched send / recv all-gather, all-to-all reduce-scatter, the `_tune` pass that
    times them: prodsearch_amd/dist.py) cannot be exercised on the one-GPU boxes this code is developed on (RCCL refuses two
    ranks on one device), so the first multi-GPU run is also their first run on RCCL.  If the first set of ranks fails, ONE fresh
    set of child processes is started with PS_DP_RS=rccl PS_DP_AG=rccl — the library collectives only (reduce_scatter_tensor /
    all_gather_into_tensor) — and the line says so (`exchange_fallback`, with rank 0's stderr of the failed attempt).  Nothing
    is retried inside a process that has initialised the GPU."""
    backend = os.environ.get('PS_DIST_BACKEND') or 'nccl'
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py --gpus %d: no GPU visible (no CPU fallback)" % a.gpus)
    if ndev < a.gpus and backend != 'gloo':
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible; one rank per GPU over RCCL needs %d "
                         "(PS_DIST_BACKEND=gloo rehearses the N-rank path on fewer devices)" % (a.gpus, ndev, a.gpus))
    rcs, lines, err_tail = _run_ranks(a, backend, {})
    fallback = None
    forced = os.environ.get('PS_DP_RS') == 'rccl' and os.environ.get('PS_DP_AG') == 'rccl'
    if (any(rcs) or len(lines) != 1) and not forced:
        sys.stderr.write("bench.py --gpus %d: first attempt failed (rank exit codes %s, %d JSON line(s)); starting a fresh set of ranks "
                         "with PS_DP_RS=rccl PS_DP_AG=rccl (library collectives only)\n" % (a.gpus, rcs, len(lines)))
        fallback = {"first_attempt_exit_codes": rcs, "first_attempt_rank0_stderr_tail": err_tail[-800:],
                    "env": {"PS_DP_RS": "rccl", "PS_DP_AG": "rccl"}}
        rcs, lines, err_tail = _run_ranks(a, backend, {'PS_DP_RS': 'rccl', 'PS_DP_AG': 'rccl'})
    if any(rcs) or len(lines) != 1:
        sys.stderr.write("bench.py --gpus %d: rank exit codes %s, %d JSON line(s) from rank 0\n" % (a.gpus, rcs, len(lines)))
        raise SystemExit(1)
    d = json.loads(lines[0])
    if d.get('n_gpus') != a.gpus:
        sys.stderr.write("bench.py --gpus %d: rank 0 reported n_gpus=%r\n" % (a.gpus, d.get('n_gpus')))
        raise SystemExit(1)
    if fallback is not None:
        d["exchange_fallback"] = True
        d["exchange_fallback_detail"] = fallback
        print(json.dumps(d))
    else:
        print(lines[0])


def fed_step_leg(a, dev, steps=400, warmup=60):
    """c2 with the batches coming from the data loader instead of sitting in HBM (not the contract's `value`,
    which wants resident inputs): a synthetic Amazon-shaped corpus (20k users, 336k reviews, the catalogue and vocabulary of
    configs[1]), `ItemPVDataloader(device, prefetch=3)` — collate on the native producer thread, one H2D copy per batch on a copy
    stream — and the same module-API step.  The whole loop between two synchronisations, data loading included."""
    from prodsearch_amd import ItemTransformerRanker, build_optim, readme_tem_args, synth
    from prodsearch_amd.dataloader import ItemPVDataloader
    c = TEM_CFG['c2']
    ns = readme_tem_args(dropout=a.dropout, embedding_size=c['D'], ff_size=c['FF'], batch_size=c['B'], fix_train_review=False)
    train_ds, _ = synth.make_corpus(7, n_users=20000, n_products=c['P'], n_queries=2000, vocab_size=V_WORDS, Q=c['Q'], W=c['W'],
                                    max_reviews_per_user=400)
    torch.manual_seed(1234)
    model = ItemTransformerRanker(ns, 'cuda', V_WORDS, c['P'], None, word_dists=synth.make_word_dists(V_WORDS))
    optim = build_optim(ns, model, None)
    model.train()
    it = iter(ItemPVDataloader(ns, train_ds, batch_size=c['B'], shuffle=True, seed=1, device=dev, drop_last=True, prefetch=3))

    def run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            loss = model(next(it))
            model.zero_grad()
            loss.backward()
            optim.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n
    run(warmup)
    t = run(steps)
    it.close()
    del model, optim
    torch.cuda.empty_cache()
    return {"ms_per_step": t * 1e3, "tuples_per_s": c['B'] * c['K'] / t, "steps": steps, "warmup": warmup,
            "what": "the c2 step fed by prodsearch_amd.dataloader.ItemPVDataloader(prefetch=3) from a synthetic corpus of %d train "
                    "samples (random history subsets of <= %d of up to 400 reviews per user): collate + H2D + step, wall clock"
                    % (len(train_ds), c['L'])}


# ------------------------------------------------------------------------------------------ main
def main():
    a = parse()
    if os.environ.get('PS_BENCH_WATCHDOG'):      # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['PS_BENCH_WATCHDOG']), exit=True)
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(a)
    from prodsearch_amd import dist as pdist
    if os.environ.get('PS_BENCH_TEST_FAIL_P2P') and os.environ.get('PS_BENCH_CHILD') and os.environ.get('RANK') == '1' \
            and not (os.environ.get('PS_DP_RS') == 'rccl' and os.environ.get('PS_DP_AG') == 'rccl'):
        # test hook (tests/test_gpu_dp.py): a rank of the FIRST self-launched set dies before it has touched the GPU, as a rank
        # whose peer-to-peer collectives failed would; the library-collectives set that self_launch starts next must deliver
        raise SystemExit(3)
    rank, local, world = pdist.init_from_env()
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE %d: refusing to report a line for a different rank count" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    try:
        out = measure(a, a.workload, rank, world, dev, a.steps, a.warmup, a.reps, not a.no_extras, a.cpu_steps, a.items)
    except Exception:
        if world > 1 and not (os.environ.get('PS_DP_RS') == 'rccl' and os.environ.get('PS_DP_AG') == 'rccl'):
            # (this process has initialised the GPU: nothing is retried here.  `python bench.py --gpus N` without torchrun does the
            # retry itself, in fresh child processes: self_launch)
            sys.stderr.write("bench.py rank %d of %d failed.  If the traceback below ends inside the sharded optimizer's peer-to-peer "
                             "collectives (prodsearch_amd/dist.py: batch_isend_irecv / all_to_all_single / _tune), rerun with "
                             "PS_DP_RS=rccl PS_DP_AG=rccl in the environment: reduce_scatter_tensor / all_gather_into_tensor "
                             "only.\n" % (rank, world))
        raise
    later = out.pop("_cpu_baseline_later", None)
    if a.workload == 'c2' and world == 1 and not a.no_extras and not a.no_also:
        out["roofline_hbm"] = gather_score_hbm_leg(dev)
        also = []
        for name, items in (('c5', ALSO_C5_ITEMS), ('c4', 0)):
            also.append(measure(a, name, rank, world, dev, 200, 30, 0, True, 1 if name == 'c4' else 0, items))
        out["also"] = also
        out["fed_by_loader"] = fed_step_leg(a, dev)
    if later is not None:
        out["cpu_baseline"] = later()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == '__main__':
    main()
