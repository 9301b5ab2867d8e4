"""profiles/r03_summary.md from the installed r03 profiles (after tools/refresh_profiles_r03.sh + tools/install_profiles_r02.py r03)."""
import csv, json, os
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles') + '/'


def table(f, n=18):
    rows = list(csv.DictReader(open(root + f)))
    out = ["| kernel | calls | avg µs | share |", "|---|---|---|---|"]
    for r in rows[:n]:
        out.append("| `%s` | %s | %.1f | %s%% |" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1000, r['Percentage']))
    return "\n".join(out)


def last(f):
    return json.loads(open(root + f).read().strip().splitlines()[-1])


def span(f):
    return [l for l in open(root + f).read().splitlines() if 'launches' in l][-1].strip()


b, r, c = last('r03_bench.json'), last('r03_rtm_bench.json'), last('r03_c5_bench.json')
hb = b['roofline_hbm']['by_batch']
also = {('c5' if 'd=256' in a['config']['workload'] else 'c4'): a for a in b.get('also', [])}
s = '''# Round 03 — summary of the committed measurements (MI355X, one GPU)

Produced by `bash tools/refresh_profiles_r03.sh` on the GPU box, then `python tools/install_profiles_r02.py r03` and
`python tools/make_profile_summary_r03.py`.  Kernel statistics are `rocprofv3 --kernel-trace --stats --output-format csv -- python3
bench.py ...` summaries; counter passes (`--pmc`) ran alone, as the guide prescribes.

## C2 — `python bench.py` (BASELINE configs[1]: item_transformer d=128, bs 384, 20 negatives, dropout 0.1)

bench line (`r03_bench.json`): **%.0f tuples/s, %.4f ms/step**, median of 200 single steps %.4f ms (p10-p90 %.4f-%.4f); round 2:
27.8 M tuples/s, 0.276 ms; round 1: 22.3 M, 0.344 ms.  Timeline `r03_step_timeline.txt`: %s.
Roofline object: `mlp_fwd_t_kernel<2,3>` (the transposed, register-chained fused forward: DESIGN.md 5), bound `mfma`, %.1f TFLOP/s of
157.3 = **%.3f** (in-step HIP-event duration %.1f µs, fastest launch %.1f µs; the rocprof average is in the table); round 2: 0.262 at
57.6 µs.  Matrix-pipe utilisation `r03_mfma_utilisation.md`, instruction mix `r03_inst_counters.txt`, wait / busy cycles
`r03_sq_counters.txt`, in-kernel `s_memtime` stamps `r03_mlp_stamps.txt`, what bounds it now `r03_mlp_notes.md`.
The same line carries `roofline_hbm` — the stand-alone gather+score launch at the C5 shape (8 M-row table, 8 rotating index sets so
that no launch finds its rows cached): B = 1024: %.0f GB/s = **%.3f** of 8 TB/s (%.1f µs), B = 8192: %.0f GB/s = **%.3f** (%.1f µs)
— and `also`: the c5-shard line (%.3f ms/step) and the c4 line (%.4f ms/step), each with its own roofline and, for c4, a CPU baseline.
Kernel statistics (`r03_bench_kernel_stats.csv`, the timed steps plus the roofline pass):

%s

## C4 — `python bench.py --workload c4` (BASELINE configs[3]: review_transformer, bs 256, K 5, R 20+30, WL 100, pvc)

bench line (`r03_rtm_bench.json`): **%.0f tuples/s, %.4f ms/step** (median %.4f); round 2: 0.502 ms; round 1: 0.733 ms.
Roofline object: `rtm_embed4_kernel`, bound `hbm`, %.0f GB/s of 8000 = **%.3f** on %.1f MB of algorithmic bytes (in-step %.1f µs; round
2: 100.9 µs with the word-rank atomics in it).  PMC `r03_rtm_embed_pmc.txt`: FETCH 34.5 MB x2 + WRITE 10.6 MB = 79.5 MB per launch
against 84.5 MB algorithmic (round 2: 123.3 MB against 91.9): nothing is re-read; the 59 MB of word rows come out of the L2s / Infinity
Cache (a 16.6 MB table), and the kernel's loads are no longer one round trip per element (DESIGN.md 5f: 55.5 -> 37.4 µs; per-workgroup lives `r03_rtm_embed4_wg_times.txt`).  Timeline `r03_rtm_step_timeline.txt`
(%s); kernel statistics (`r03_rtm_kernel_stats.csv`):

%s

## C5 shard — `python bench.py --workload c5 --items 8000000` (one GPU's share of BASELINE configs[4]: d=256, bs 1024, row-sparse Adam)

bench line (`r03_c5_bench.json`): **%.0f tuples/s, %.3f ms/step**; roofline object: the gather+score launch inside the step, %.0f GB/s =
%.3f of peak (it shares the machine there; alone: `roofline_hbm` above and `r03_gather_c5_shape.jsonl`); PMC traffic
`r03_gather_score_c5_pmc.txt` (62.8 MB against 67.6 MB algorithmic).  `r03_c5_bench_fp32_products.json`: the same run with
`PS_GEMM_X3=0`.  Timeline `r03_c5_step_timeline.txt` (%s), kernel statistics `r03_c5_kernel_stats.csv`.  Unchanged this round:
80 %% of its kernel time is the bf16x3 GEMM kernel at 100-120 TFLOP/s fp32-equivalent; why a d = 256 fused kernel would not beat
that, and what would (pre-split operands, a wider wave tile): DESIGN.md 9.

## Deterministic mode (`PS_DETERMINISTIC=1`)

`r03_det_step_timeline.txt` (C2: %s) and, new this round, `r03_det_rtm_step_timeline.txt` (review transformer: %s).
''' % (b['value'], b['ms_per_step'], b['median_ms_per_step'], b['p10_p90_ms_per_step'][0], b['p10_p90_ms_per_step'][1],
       span('r03_step_timeline.txt'),
       b['roofline']['achieved'], b['roofline']['frac'], b['roofline']['us_per_launch'], b['roofline']['us_per_launch_min'],
       hb[0]['achieved'], hb[0]['frac'], hb[0]['us_per_launch'], hb[1]['achieved'], hb[1]['frac'], hb[1]['us_per_launch'],
       also['c5']['ms_per_step'], also['c4']['ms_per_step'],
       table('r03_bench_kernel_stats.csv', 17),
       r['value'], r['ms_per_step'], r['median_ms_per_step'], r['roofline']['achieved'], r['roofline']['frac'],
       r['roofline']['bytes_per_launch'] / 1e6, r['roofline']['us_per_launch'], span('r03_rtm_step_timeline.txt'),
       table('r03_rtm_kernel_stats.csv', 22),
       c['value'], c['ms_per_step'], c['roofline']['achieved'], c['roofline']['frac'], span('r03_c5_step_timeline.txt'),
       span('r03_det_step_timeline.txt'), span('r03_det_rtm_step_timeline.txt'))
open(root + 'r03_summary.md', 'w').write(s)
print(s[:1500])
