#!/usr/bin/env python3
"""bench.py — train (u,q,i,neg) tuples/sec of the ranking-loss step on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c2|c4|c5]
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

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
Then, outside that region (rank 0): `median_ms_per_step` over --reps steps timed one by one with HIP events, and
  roofline     — the workload's dominant gather kernel: algorithmic bytes per launch / its average IN-STEP duration,
                 measured live with one HIP event pair around every launch of it, on the stream it is launched on, over a
                 second pass of --steps steps (ps_ktimer_arm / ps_ktimer_read, include/prodsearch_hip.h)
  cpu_baseline — the oracle (op-for-op CPU restatement; for c2 incl. the B*(K+1) replicated encoder) timed on this box's
                 host cores on a bounded sample (rank 0, N=1)
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 MFMA (v_mfma_f32_32x32x2_f32), = the fp32 vector peak
P_ITEMS, V_WORDS = 18357, 32387
B, K, L, Q, W, D = 384, 20, 20, 8, 1, 128
FF = 512
# --workload c5: one GPU's shard of BASELINE configs[4] (50 M-item table, d=256, bs=1024/GPU);
# 51 GB table + dense gradient + Adam moments = 205 GB of HBM, row-sparse optimizer (dense Adam would stream 1.4 TB)
C5 = dict(P_ITEMS=50_000_000, B=1024, D=256, FF=1024)
# --workload c4: BASELINE configs[3]
C4 = dict(RC=296000, B=256, K=5, WL=100, U=20, I=30)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--reps', type=int, default=200, help='steps timed one by one with HIP events for the median (0 = skip)')
    ap.add_argument('--dropout', type=float, default=0.1)
    ap.add_argument('--cpu-steps', type=int, default=3, help='CPU-baseline sample (0 = skip)')
    ap.add_argument('--no-extras', action='store_true', help='skip median / roofline / cpu_baseline legs')
    ap.add_argument('--workload', default='c2', choices=['c2', 'c4', 'c5'],
                    help='c2 = BASELINE configs[1] (the metric); c4 = configs[3] (RTM); c5 = per-GPU shard of configs[4]')
    ap.add_argument('--encoder', default='pvc', choices=['pvc', 'pv'], help='c4: review encoder')
    ap.add_argument('--items', type=int, default=0, help='override the catalogue size (c5 dry runs)')
    ap.add_argument('--row-sparse', action='store_true',
                    help='touched-rows-only zero/clip/Adam/exchange (args.row_sparse_adam); always on for c5')
    return ap.parse_args()


# ------------------------------------------------------------------------------------------ workloads
class TemWorkload(object):
    """item_transformer (c2 / c5)."""

    def __init__(self, a, rank, dev):
        from prodsearch_amd import ItemTransformerRanker, build_optim, readme_tem_args, synth
        self.a = a
        self.ns = readme_tem_args(dropout=a.dropout, embedding_size=D, ff_size=FF, row_sparse_adam=a.row_sparse)
        self.wd = synth.make_word_dists(V_WORDS)
        torch.manual_seed(1234)                     # identical init on every rank
        self.model = ItemTransformerRanker(self.ns, 'cuda', V_WORDS, P_ITEMS, None, word_dists=self.wd)
        self.optim = build_optim(self.ns, self.model, None)
        self.batches = [synth.make_tem_batch(1000 + 97 * rank + i, B, P_ITEMS, V_WORDS, Q=Q, L=L, W=W,
                                             word_dists=self.wd).to(dev) for i in range(8)]
        self.B, self.K = B, K
        self.ktag = 'gather_score'

    def forward(self, i):
        return self.model(self.batches[i % len(self.batches)])      # trainer.py:74 (negatives sampled on device)

    def describe(self):
        R = next(iter(self.model._plans.values())).layout.R
        return ("item_transformer d=%d 1 layer 8 heads ff=%d uprev=20 bs=%d/GPU 20 neg Q=8 W=1 P=%d V=32387 dropout=%.2f%s "
                "(BASELINE configs[%d])" % (D, FF, B, P_ITEMS, self.a.dropout, " row-sparse Adam" if self.a.row_sparse else "",
                                            4 if self.a.workload == 'c5' else 1)), {"replicas_per_row": R}

    def metric(self):
        return "train (u,q,i,neg) tuples/sec at bs=%d, 20 neg, d=%d" % (B, D)

    def roofline_spec(self):
        """The step's dominant kernel.  With replicas at d = 128 (c2) that is the fused per-replica forward
        (mlp_fwd_ws_kernel: Wo -> LN -> W1 -> GELU -> W2 -> LN for every one of the B*(K+1) replica rows, with the item
        gather + score + loss in its epilogue): MFMA-bound, 2*d*d + 4*d*F flops per row.  Otherwise (c5: d = 256) the
        stand-alone gather+score launch: HBM-bound, every table row once (4d B) + its int64 index, every distinct vector
        it is dotted with, every score written."""
        R = next(iter(self.model._plans.values())).layout.R
        rows = B * (1 + K) * (1 + W)
        vecs = B * R + B                            # encoder outputs + target-item rows
        gather_bytes = rows * (4 * D + 8) + vecs * 4 * D + rows * 4
        if D == 128 and R > 1 and (B * R + 31) // 32 <= 256:
            flops = B * R * (2 * D * D + 4 * D * FF)
            return dict(tag='mlp_fwd', bound='mfma', work=flops, peak=MFMA_F32_PEAK_TFLOPS, unit='TFLOP/s', scale=1e12,
                        kernel="mlp_fwd_ws_kernel (fused per-replica encoder tail, %d rows x (2*%d*%d + 4*%d*%d) flop, fp32 MFMA; "
                               "item gather + score + loss folded into its epilogue: %d B of gathered rows and scores)"
                               % (B * R, D, D, D, FF, B * (1 + K) * (4 * D + 8 + 4)),
                        traffic_key=None, extra={"gather_score_bytes_all_tasks": gather_bytes})
        return dict(tag='gather_score', bound='hbm', work=gather_bytes, peak=HBM_PEAK_GBS, unit='GB/s', scale=1e9,
                    kernel="score_fwd_wide_kernel<1,%d> (embedding gather + score; 16 lanes per %d-B row)" % (D // 64, 4 * D),
                    traffic_key=None, extra={})

    def cpu_baseline(self, n_steps):
        return cpu_baseline_tem(self.ns, n_steps)


class RtmWorkload(object):
    """review_transformer (c4)."""

    def __init__(self, a, rank, dev):
        from prodsearch_amd import ProductRanker, build_optim, default_args, synth, rtm_data
        self.a = a
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
        self.ktag = 'rtm_embed'

    def forward(self, i):
        return self.model(self.batches[i % len(self.batches)], train_pv=False)

    def describe(self):
        c = C4
        return ("review_transformer (RTM) d=128 1 layer 8 heads ff=512 bs=%d/GPU K=%d R=%d+%d WL=%d %s review encoder "
                "dropout=%.2f corrupt_rate=0.90 %dk reviews V=32387 (BASELINE configs[3])"
                % (c['B'], c['K'], c['U'], c['I'], c['WL'], self.a.encoder, self.a.dropout, c['RC'] // 1000)), {}

    def metric(self):
        return "train (u,q,i,neg) tuples/sec at bs=%d, %d neg, d=128 (review_transformer)" % (C4['B'], C4['K'])

    def roofline_spec(self):
        """rtm_embed4_kernel (review vectors, PVC.py:46-61): per valid review slot its WL int64 word ids, per word that is
        neither padding nor dropped by the token corruption one 4d-byte row (expected count: the Philox masks are drawn
        on the device), the rows of the encoder input it writes and the per-word ranks it leaves for the backward."""
        c, d = C4, 128
        b0 = self.cpu_batches[0]
        pad_r = c['RC'] - 1
        slots = int((b0.pos_prod_ridxs != pad_r).sum() + (b0.neg_prod_ridxs != pad_r).sum())
        nseq = c['B'] * (c['K'] + 1)
        out_bytes = (slots + nseq) * d * 4              # x: the real positions + one query row per sequence (padded
        if self.a.encoder == 'pvc':                     # positions are neither read nor written: DESIGN.md 7c)
            words = int((b0.pos_prod_rword_idxs != V_WORDS - 1).sum() + (b0.neg_prod_rword_idxs != V_WORDS - 1).sum())
            rows = words * (1.0 - 0.9)
            rank_bytes = slots * c['WL'] * 4            # first pass of the backward's inverted index rides in this kernel
            nbytes = slots * c['WL'] * 8 + rows * 4 * d + out_bytes + rank_bytes
            note = "%d review slots x %d ids + %.0f surviving word rows (%.0f non-pad words x 0.1) + %d B of x + %d B of word ranks" % (
                slots, c['WL'], rows, words, out_bytes, rank_bytes)
        else:
            nbytes = slots * (8 + 4 * d) + out_bytes
            note = "%d review rows + %d B of x" % (slots, out_bytes)
        return dict(tag='rtm_embed', bound='hbm', work=int(nbytes), peak=HBM_PEAK_GBS, unit='GB/s', scale=1e9,
                    kernel="rtm_embed4_kernel (review-vector gather + mean-pool + word ranks; %s)" % note, traffic_key=None, extra={})

    def cpu_baseline(self, n_steps):
        return cpu_baseline_rtm(self, n_steps)


# ------------------------------------------------------------------------------------------ CPU baselines
def _best_pool(one_step, ncpu, cands=(8, 16, 32, 64)):
    """torch's intra-op pool does not scale to every hardware thread of a 2-socket host on these small ops: a short
    calibration picks the fastest of a few pool sizes; `cores` reports the pool that was timed."""
    cands = sorted({min(ncpu, c) for c in cands})
    torch.set_num_threads(cands[0])
    one_step()                                   # warm-up (allocator, first-touch)
    trial = {}
    for c in cands:
        torch.set_num_threads(c)
        trial[c] = one_step()
    best = min(trial, key=trial.get)
    torch.set_num_threads(best)
    return best, trial, cands


def cpu_baseline_tem(args_ns, n_steps):
    """The oracle as the CPU path: same shapes/flags, reference structure (replicated encoder,
    torch RNG dropout), fwd + bwd + clip/Adam on the host cores."""
    from oracle import tem as otem, optim as ooptim
    from prodsearch_amd import synth
    ncpu = os.cpu_count() or 1
    wd = synth.make_word_dists(V_WORDS)
    shapes = synth.tem_param_shapes(args_ns, V_WORDS, P_ITEMS)
    Pm = {k: v.requires_grad_(True) for k, v in synth.make_state_dict(shapes, 1, {'product_emb.weight': P_ITEMS}).items()}
    opt = ooptim.ClipAdam(args_ns.lr, args_ns.max_grad_norm, args_ns.beta1, args_ns.beta2, 1e-9, args_ns.l2_lambda)
    pad = otem.tem_pad_rows(args_ns, V_WORDS, P_ITEMS)
    drop = otem.TorchDropout(args_ns.dropout) if args_ns.dropout > 0 else None
    counter = [0]

    def one_step():
        s = counter[0]
        counter[0] += 1
        batch = synth.make_tem_batch(100 + s, B, P_ITEMS, V_WORDS, Q=Q, L=L, W=W, word_dists=wd)
        ni, nw = synth.sample_negatives(200 + s, B, K, W, P_ITEMS, wd)
        t0 = time.perf_counter()
        loss, _, _ = otem.tem_forward(Pm, args_ns, batch, ni, nw, V_WORDS, P_ITEMS, training=True,
                                      replicate=True, drop=drop)
        grads = otem.grads_of(loss, Pm, pad)
        with torch.no_grad():
            opt.step(Pm, grads)
        return time.perf_counter() - t0

    best, trial, cands = _best_pool(one_step, ncpu)
    times = [one_step() for _ in range(n_steps)]
    t = sum(times) / len(times)
    return {"value": B * K / t, "unit": "tuples/s", "cores": best, "kind": "port",
            "sample": "%d steps of the same B=%d,K=%d,d=%d step (fwd+bwd+clip/Adam, replicated encoder, dropout %.2f), "
                      "%.2f s/step on %d of %d host threads (fastest of pools %s after 1 warm-up step)"
                      % (len(times), B, K, D, args_ns.dropout, t, best, ncpu,
                         ", ".join("%d: %.1fs" % (c, trial[c]) for c in cands))}


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

    best, trial, cands = _best_pool(one_step, ncpu, cands=(8, 16, 32))
    times = [one_step() for _ in range(n_steps)]
    t = sum(times) / len(times)
    return {"value": c['B'] * c['K'] / t, "unit": "tuples/s", "cores": best, "kind": "port",
            "sample": "%d steps of the same B=%d,K=%d RTM step (oracle/rtm.py: fwd + autograd bwd + clip/Adam, dropout %.2f, "
                      "token corruption 0.9), %.2f s/step on %d of %d host threads (fastest of pools %s after 1 warm-up step)"
                      % (len(times), c['B'], c['K'], ns.dropout, t, best, ncpu,
                         ", ".join("%d: %.1fs" % (cc, trial[cc]) for cc in cands))}


# ------------------------------------------------------------------------------------------ main
def main():
    a = parse()
    if os.environ.get('PS_BENCH_WATCHDOG'):      # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['PS_BENCH_WATCHDOG']), exit=True)
    from prodsearch_amd import _lib, dist as pdist
    rank, local, world = pdist.init_from_env()
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE %d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if a.workload == 'c5':
        globals().update(C5)
        a.row_sparse = True
    if a.items:
        globals().update(P_ITEMS=a.items)
    wl = RtmWorkload(a, rank, dev) if a.workload == 'c4' else TemWorkload(a, rank, dev)
    model, optim = wl.model, wl.optim
    model._seed = pdist.rank_seed(wl.ns.seed, rank)
    pdist.broadcast_parameters(model)
    exchange = pdist.make_exchange(model, optim)
    model.train()

    def step(i):
        loss = wl.forward(i)                             # trainer.py:74
        model.zero_grad()                                # :76
        loss.backward()                                  # :77
        exchange()                                       # RCCL all-reduce of the flat gradient (N>1)
        optim.step()                                     # :78
        return loss

    for i in range(a.warmup):
        step(i)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed = pdist.max_over_ranks(time.perf_counter() - t0, dev)
    last_loss = float(loss.detach())

    desc, extra_cfg = wl.describe()
    Bw, Kw = wl.B, wl.K
    out = {
        "metric": wl.metric(),
        "value": world * Bw * Kw * a.steps / elapsed, "unit": "tuples/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": dict({"workload": desc, "global_batch": world * Bw, "parallelism": "dp%d" % world,
                        "step": "sample+fwd+bwd+%sclip/Adam via nn.Module API (trainer.py:74-78)"
                                % ("allreduce+" if world > 1 else "")}, **extra_cfg),
        "samples_per_s": world * Bw * a.steps / elapsed, "final_loss": last_loss,
    }
    if not a.no_extras:
        # Every rank runs these extra steps (they contain the gradient exchange: a collective only rank 0 entered would
        # never return); rank 0 keeps the numbers.
        lib = _lib.load()
        # (1) median of single steps, each between two HIP events on the step's stream
        if a.reps > 0:
            st = torch.cuda.current_stream()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
            for i, (e0, e1) in enumerate(ev):
                e0.record(st)
                step(i)
                e1.record(st)
            torch.cuda.synchronize()
            ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
            out["median_ms_per_step"] = ts[len(ts) // 2]
            out["reps"] = a.reps
            out["p10_p90_ms_per_step"] = [ts[len(ts) // 10], ts[(9 * len(ts)) // 10]]
        # (2) roofline: in-step duration of the workload's gather kernel, one event pair per launch on its own stream
        spec = wl.roofline_spec()
        _lib.check(lib.ps_ktimer_arm(spec['tag'].encode(), a.steps), 'ps_ktimer_arm')
        for i in range(a.steps):
            step(i)
        avg, mn, cnt = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int32(0)
        _lib.check(lib.ps_ktimer_read(ctypes.byref(avg), ctypes.byref(mn), ctypes.byref(cnt)), 'ps_ktimer_read')
        traffic = traffic_src = None                # PMC passes cannot run inside this process
        if spec['traffic_key']:
            try:
                tj = json.load(open(os.path.join(ROOT, 'profiles', 'gather_score_traffic.json')))
                traffic, traffic_src = tj.get(spec['traffic_key']), "committed profile: " + tj.get('source', '')
            except Exception:
                pass
        t_k = avg.value * 1e-6
        ach = spec['work'] / t_k / spec['scale'] if t_k > 0 else 0.0
        out["roofline"] = dict({"bound": spec['bound'], "achieved": ach, "peak": spec['peak'], "unit": spec['unit'],
                                "frac": ach / spec['peak'], "traffic": traffic, "traffic_source": traffic_src,
                                "kernel": spec['kernel'],
                                ("flops_per_launch" if spec['bound'] == 'mfma' else "bytes_per_launch"): spec['work'],
                                "us_per_launch": avg.value, "us_per_launch_min": mn.value, "launches_timed": cnt.value,
                                "timing": "HIP event pair around every in-step launch, on the launch stream, over a second "
                                          "pass of %d steps (ps_ktimer)" % a.steps}, **spec['extra'])
        if world == 1 and a.cpu_steps > 0 and a.workload in ('c2', 'c4'):
            out["cpu_baseline"] = wl.cpu_baseline(a.cpu_steps)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == '__main__':
    main()
