"""profiles/r05_summary.md from the installed r05 profiles (after tools/refresh_profiles_r05.sh + tools/install_profiles_r02.py r05)."""
import csv, json, os
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles') + '/'


def table(f, n=18, only=None):
    rows = list(csv.DictReader(open(root + f)))
    if only:
        rows = [r for r in rows if only in r['Name']]
    out = ["| kernel | calls | avg µs | share |", "|---|---|---|---|"]
    for r in rows[:n]:
        out.append("| `%s` | %s | %.1f | %s%% |" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1000, r['Percentage']))
    return "\n".join(out)


def last(f):
    return json.loads(open(root + f).read().strip().splitlines()[-1])


def span(f):
    return [l for l in open(root + f).read().splitlines() if 'launches' in l][-1].strip()


def kavg(f, name):
    for r in csv.DictReader(open(root + f)):
        if name in r['Name']:
            return float(r['AverageNs']) / 1000
    return float('nan')


b, r, c = last('r05_bench.json'), last('r05_rtm_bench.json'), last('r05_c5_bench.json')
hb = b['roofline_hbm']['by_batch']
also = {('c5' if 'd=256' in a['config']['workload'] else 'c4'): a for a in b.get('also', [])}
g1 = kavg('r05_gather_score_kernel_stats.csv', 'score_fwd_sidx_kernel')
g8 = kavg('r05_gather_score_b8192_kernel_stats.csv', 'score_fwd_sidx_kernel')
mlp = kavg('r05_bench_kernel_stats.csv', 'mlp_fwd_t_kernel')
s = '''# Round 05 — summary of the committed measurements (MI355X, one GPU)

Produced by `bash tools/refresh_profiles_r05.sh` on the GPU box, then `python tools/install_profiles_r02.py r05` and
`python tools/make_profile_summary_r05.py`.  Kernel statistics are `rocprofv3 --kernel-trace --stats --output-format csv -- python3
...` summaries; counter passes (`--pmc`) ran alone, as the guide prescribes.  Kernel durations inside `bench.py` are now read from a HIP
event pair BOUND to the launch (`hipExtLaunchKernelGGL` start / stop events = the dispatch's own begin / end, the quantity rocprofv3's
kernel trace reports; `tools/micro/extlaunch.hip`: 5.59 us against 5.75 us, where a `hipEventRecord` pair around the launch read 8.22).
The round's other notes: `r05_gemm_notes.md` (the GEMM, closed), `r05_gather_sidx.txt` + `r05_gather_score_wg_times.txt` (the index hop
on the scalar path), `r05_kvq_wg_times.txt` (the fused projection + attention forward), `r05_front_end_ab.txt` (the two C2 fusions, same
box, alternating), `r05_mlp_notes.md` + `r05_mlp_stamps_before.txt` / `r05_mlp_stamps.txt` + `r05_c2_pmc_l2.txt` (the fused per-replica
kernels: what their stores cost, why their chain is L2-bound, five experiments that did not move it), `r05_attn_bwd_wg_times.txt` /
`r05_score_bwd_wg_times.txt` (per-workgroup phases of the replica attention backward and of the score backward in the step),
`r05_c4_wreduce_notes.md`, `r05_env_matrix.txt`.

## C2 — `python bench.py` (BASELINE configs[1]: item_transformer d=128, bs 384, 20 negatives, dropout 0.1)

bench line (`r05_bench.json`): **%.0f tuples/s, %.4f ms/step**, median of 200 single steps %.4f ms (p10-p90 %.4f-%.4f), %d untimed
pre-warm steps in front of the %d warm-up steps; round 4: 33.4-34.3 M tuples/s, 0.224-0.230 ms; round 3: 0.2305-0.245; round 2: 0.276; round 1: 0.344.
Boxes of this pool differ by up to 8 %% on identical code this round (0.229 / 0.247 ms for the round-4 path on two boxes): compare
`r05_front_end_ab.txt`, one box, alternating runs.  Timeline `r05_step_timeline.txt`: %s.
Roofline object: `mlp_fwd_t_kernel<2,3>`, bound `mfma`, %.1f TFLOP/s of 157.3 = **%.3f** on the in-step dispatch duration %.1f us
(fastest launch %.1f us); rocprofv3's average for the kernel %.1f us = %.3f.  `roofline_longest_kernel` (grouped W2 / W1 / Wo weight
gradients, side stream): %.1f us in the step = **%.3f**; rocprofv3 %.1f us.
`roofline_hbm` — the stand-alone gather+score launch at the C5 shape (8 M-row table, 8 rotating index sets), 40 launches, a bound event
pair per launch: B = 1024: %.0f GB/s = **%.3f** of 8 TB/s (%.2f us; back to back %.2f us per launch = %.3f as launch throughput),
B = 8192: %.0f GB/s = **%.3f** (%.1f us; back to back %.1f).  rocprofv3 kernel trace of `tools/gather_c5.py`'s loop
(`r05_gather_score_kernel_stats.csv`, `r05_gather_score_b8192_kernel_stats.csv`; the averages include the 8 cold warm-up launches of 14-27 us):
%.1f us and %.1f us per launch = %.3f / %.3f.  PMC traffic `r05_gather_score_c5_pmc.txt`.
`also`: the c5 line at the STATED 50 M-row table (%.3f ms/step, in-step gather+score %.1f us = %.3f) and the c4 line (%.4f ms/step).
Kernel statistics (`r05_bench_kernel_stats.csv`, the timed steps plus the roofline passes):

%s

## C4 — `python bench.py --workload c4` (BASELINE configs[3]: review_transformer, bs 256, K 5, R 20+30, WL 100, pvc)

bench line (`r05_rtm_bench.json`): **%.0f tuples/s, %.4f ms/step** (median %.4f); round 4: 0.382-0.387 ms; round 3: 0.388-0.398; round 2: 0.502; round 1: 0.733.
Roofline object: `rtm_embed4_kernel` ALONE (the event pair is bound to its dispatch: round 4's scope also bracketed the 7 us group-list
kernel), bound `hbm`, %.0f GB/s of 8000 = **%.3f** on %.1f MB of algorithmic bytes (in-step %.1f us; rocprofv3 %.1f us).
Timeline `r05_rtm_step_timeline.txt` (%s); kernel statistics (`r05_rtm_kernel_stats.csv`):

%s

## C5 shard — `python bench.py --workload c5 --items 8000000` (one GPU's share of BASELINE configs[4]: d=256, bs 1024, row-sparse Adam)

bench line (`r05_c5_bench.json`): **%.0f tuples/s, %.3f ms/step** (round 4: 1.25-1.30; the 50 M-row table in the default run's `also`:
%.3f); roofline object: the gather+score launch inside the step (`score_fwd_sidx_kernel<4,16>`), %.0f GB/s = %.3f of peak (round 4: 0.41
between event pairs, 0.47 by rocprofv3).  The step is GEMM time (`r05_gemm_notes.md`); what moved it this round is the key-split replica
attention backward (`attn_bwd_wk_kernel<32,3>`: 30 us alone for 46.5, 114 in the step for 147; 1.247 -> 1.226 ms on one box).  Timeline
`r05_c5_step_timeline.txt` (%s); kernel statistics (`r05_c5_kernel_stats.csv`):

%s
''' % (b['value'], b['ms_per_step'], b['median_ms_per_step'], b['p10_p90_ms_per_step'][0], b['p10_p90_ms_per_step'][1],
       b.get('prewarm_steps', 0), b['warmup'], span('r05_step_timeline.txt'),
       b['roofline']['achieved'], b['roofline']['frac'], b['roofline']['us_per_launch'], b['roofline']['us_per_launch_min'],
       mlp, b['roofline']['flops_per_launch'] / (mlp * 1e-6) / 1e12 / 157.3,
       b['roofline_longest_kernel']['us_per_launch'], b['roofline_longest_kernel']['frac'], kavg('r05_bench_kernel_stats.csv', 'gemm_x3_kernel<1, 1, 0, 1, 1, 0, 1>'),
       hb[0]['achieved'], hb[0]['frac'], hb[0]['us_per_launch'], hb[0]['back_to_back_us_per_launch'], hb[0]['back_to_back_throughput_frac'],
       hb[1]['achieved'], hb[1]['frac'], hb[1]['us_per_launch'], hb[1]['back_to_back_us_per_launch'],
       g1, g8, hb[0]['bytes_per_launch'] / (g1 * 1e-6) / 8e12, hb[1]['bytes_per_launch'] / (g8 * 1e-6) / 8e12,
       also['c5']['ms_per_step'], also['c5']['roofline']['us_per_launch'], also['c5']['roofline']['frac'], also['c4']['ms_per_step'],
       table('r05_bench_kernel_stats.csv', 17),
       r['value'], r['ms_per_step'], r['median_ms_per_step'], r['roofline']['achieved'], r['roofline']['frac'],
       r['roofline']['bytes_per_launch'] / 1e6, r['roofline']['us_per_launch'], kavg('r05_rtm_kernel_stats.csv', 'rtm_embed4_kernel'), span('r05_rtm_step_timeline.txt'),
       table('r05_rtm_kernel_stats.csv', 22),
       c['value'], c['ms_per_step'], also['c5']['ms_per_step'], c['roofline']['achieved'], c['roofline']['frac'],
       span('r05_c5_step_timeline.txt'), table('r05_c5_kernel_stats.csv', 24))
open(root + 'r05_summary.md', 'w').write(s)
print(s[:1200])
