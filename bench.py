#!/usr/bin/env python3
"""bench.py — train (u,q,i,neg) tuples/sec of the TEM ranking-loss step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1], the config the metric is quoted on): item_transformer,
d=128, 1 layer, 8 heads, ff 512, uprev 20, B=384 per GPU, 20 negatives, Q=8, W=1,
P=18,357 items, V=32,387 words, README training flags (README.md:13-25)
incl. the reference's default dropout 0.1 (main.py:60) => the K+1 encoder replicas are
really computed.  One step = trainer.py:74-78: loss = model(batch); model.zero_grad();
loss.backward(); optim.step()  — sampling, forward, backward, grad exchange (N>1), clip+Adam.
Inputs are synthetic and already resident in HBM.

The JSON line also carries
  roofline     — the embedding-gather+score kernel: algorithmic bytes / HIP-event time
  cpu_baseline — the oracle (op-for-op CPU restatement incl. the B*(K+1) replicated
                 encoder) timed on this box's host cores on a bounded sample (rank 0, N=1)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
P_ITEMS, V_WORDS = 18357, 32387
B, K, L, Q, W, D = 384, 20, 20, 8, 1, 128
FF = 512
# --workload c5: one GPU's shard of BASELINE configs[4] (50 M-item table, d=256, bs=1024/GPU);
# 51 GB table + dense gradient + Adam moments = 205 GB of HBM, row-sparse optimizer (dense Adam would stream 1.4 TB)
C5 = dict(P_ITEMS=50_000_000, B=1024, D=256, FF=1024)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--dropout', type=float, default=0.1)
    ap.add_argument('--cpu-steps', type=int, default=3, help='CPU-baseline sample (0 = skip)')
    ap.add_argument('--kernel-iters', type=int, default=300)
    ap.add_argument('--no-extras', action='store_true', help='skip roofline / cpu_baseline legs')
    ap.add_argument('--workload', default='c2', choices=['c2', 'c5'],
                    help='c2 = BASELINE configs[1] (the metric); c5 = per-GPU shard of configs[4]')
    ap.add_argument('--items', type=int, default=0, help='override the catalogue size (c5 dry runs)')
    ap.add_argument('--row-sparse', action='store_true',
                    help='touched-rows-only zero/clip/Adam/exchange (args.row_sparse_adam); always on for c5')
    return ap.parse_args()


def make_model(args_ns, device, seed):
    from prodsearch_amd import ItemTransformerRanker, build_optim, synth
    wd = synth.make_word_dists(V_WORDS)
    torch.manual_seed(seed)                     # identical init on every rank
    model = ItemTransformerRanker(args_ns, device, V_WORDS, P_ITEMS, None, word_dists=wd)
    optim = build_optim(args_ns, model, None)
    return model, optim, wd


def gather_score_bytes(R):
    """Algorithmic bytes of ONE gather+score launch: every table row once
    (4d B) + its int64 index, every distinct vector it is dotted with, every score written."""
    rows = B * (1 + K) * (1 + W)
    vecs = B * R + B                            # encoder outputs + target-item rows
    return rows * (4 * D + 8) + vecs * 4 * D + rows * 4


def time_gather_score(model, plan, iters):
    """Average duration of the gather+score launch, HIP events on the launch stream."""
    from prodsearch_amd import _lib
    lib = _lib.load()
    ps, _ = model._structs()
    st = torch.cuda.current_stream()
    for _ in range(20):
        _lib.check(lib.ps_gather_score(plan.desc, ps, plan.batch, plan.ws.data_ptr(), st.cuda_stream), 'gather_score')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(st)
    for _ in range(iters):
        lib.ps_gather_score(plan.desc, ps, plan.batch, plan.ws.data_ptr(), st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def cpu_baseline(args_ns, n_steps):
    """The oracle as the CPU path: same shapes/flags, reference structure (replicated encoder,
    torch RNG dropout), fwd + bwd + clip/Adam on the host cores.  torch's intra-op pool does not
    scale to every hardware thread of a 2-socket host on these small ops, so a short calibration
    picks the fastest of a few pool sizes first; ``cores`` reports the pool that was timed."""
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

    cands = sorted({min(ncpu, c) for c in (8, 16, 32, 64)})
    torch.set_num_threads(cands[0])
    one_step()                                   # warm-up (allocator, first-touch)
    trial = {}
    for c in cands:
        torch.set_num_threads(c)
        trial[c] = one_step()
    best = min(trial, key=trial.get)
    torch.set_num_threads(best)
    times = [one_step() for _ in range(n_steps)]
    t = sum(times) / len(times)
    return {"value": B * K / t, "unit": "tuples/s", "cores": best, "kind": "port",
            "sample": "%d steps of the same B=%d,K=%d,d=%d step (fwd+bwd+clip/Adam, replicated encoder, dropout %.2f), "
                      "%.2f s/step on %d of %d host threads (fastest of pools %s after 1 warm-up step)"
                      % (len(times), B, K, D, args_ns.dropout, t, best, ncpu,
                         ", ".join("%d: %.1fs" % (c, trial[c]) for c in cands))}


def main():
    a = parse()
    if os.environ.get('PS_BENCH_WATCHDOG'):      # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['PS_BENCH_WATCHDOG']), exit=True)
    from prodsearch_amd import dist as pdist, readme_tem_args, synth
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
    ns = readme_tem_args(dropout=a.dropout, embedding_size=D, ff_size=FF, row_sparse_adam=a.row_sparse)
    model, optim, wd = make_model(ns, 'cuda', seed=1234)
    model._seed = pdist.rank_seed(ns.seed, rank)
    pdist.broadcast_parameters(model)
    exchange = pdist.make_exchange(model, optim)
    model.train()
    batches = [synth.make_tem_batch(1000 + 97 * rank + i, B, P_ITEMS, V_WORDS, Q=Q, L=L, W=W, word_dists=wd).to(dev)
               for i in range(8)]

    def step(i):
        loss = model(batches[i % len(batches)])          # trainer.py:74 (negatives sampled on device)
        model.zero_grad()                                # :76
        loss.backward()                                  # :77
        exchange()                                       # RCCL all-reduce of the flat gradient (N>1)
        optim.step()                                     # :78
        return loss

    # Warm-up, then the roofline leg (rank 0), then the timed steps.  The step's stream join is a write-value / wait-value
    # pair (ps_set_side_mode, include/prodsearch_hip.h); back-to-back launch timings on a stream that has carried one are
    # noisy (4.2 us in 7 of 12 runs, 4.4-7.3 us in the others), so the warm-up steps cross streams with events and the
    # gather+score launch is timed on the warm, still pristine training stream (always 4.19-4.27 us there)
    from prodsearch_amd import _lib
    crossing = _lib.load().ps_set_side_mode(0)
    for i in range(max(a.warmup - 1, 0)):
        step(i)
    roof = None
    if rank == 0 and not a.no_extras:
        if not model._plans:                         # --warmup 0 / 1: one forward builds the launch plan
            with torch.no_grad():
                model(batches[0])
        plan0 = next(iter(model._plans.values()))
        roof = (time_gather_score(model, plan0, a.kernel_iters), plan0.layout.R)
    _lib.load().ps_set_side_mode(crossing)
    if a.warmup > 0:
        step(a.warmup - 1)                           # the last warm-up step runs in the timed configuration
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

    out = {
        "metric": "train (u,q,i,neg) tuples/sec at bs=%d, 20 neg, d=%d" % (B, D),
        "value": world * B * K * a.steps / elapsed, "unit": "tuples/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "item_transformer d=%d 1 layer 8 heads ff=%d uprev=20 bs=%d/GPU 20 neg "
                               "Q=8 W=1 P=%d V=32387 dropout=%.2f%s (BASELINE configs[%d])"
                               % (D, FF, B, P_ITEMS, a.dropout, " row-sparse Adam" if a.row_sparse else "",
                                  4 if a.workload == 'c5' else 1),
                   "global_batch": world * B, "parallelism": "dp%d" % world,
                   "step": "sample+fwd+bwd+%sclip/Adam via nn.Module API (trainer.py:74-78)"
                           % ("allreduce+" if world > 1 else ""),
                   "replicas_per_row": next(iter(model._plans.values())).layout.R},
        "samples_per_s": world * B * a.steps / elapsed, "final_loss": last_loss,
    }
    if rank == 0 and not a.no_extras:
        t_k, R_k = roof
        nbytes = gather_score_bytes(R_k)
        traffic = None          # PMC passes cannot run inside this process: taken from the committed profile
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', 'gather_score_traffic.json')))
            traffic = tj.get('R%d_bytes_per_launch' % R_k) if a.workload == 'c2' else None
        except Exception:
            pass
        out["roofline"] = {"bound": "hbm", "achieved": nbytes / t_k / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": nbytes / t_k / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                           "kernel": "score_fwd_wide_kernel<1,%d> (embedding gather + score; 16 lanes per %d-B row)" % (D // 64, 4 * D),
                           "bytes_per_launch": nbytes, "us_per_launch": t_k * 1e6}
        if world == 1 and a.cpu_steps > 0 and a.workload == 'c2':
            out["cpu_baseline"] = cpu_baseline(ns, a.cpu_steps)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == '__main__':
    main()
