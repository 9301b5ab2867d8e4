"""Copy what tools/refresh_profiles_r02.sh / refresh_profiles_r03.sh left under gpurun_out/refresh/ into profiles/ (tracked).
    python tools/install_profiles_r02.py [r02]"""
import os, shutil, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else 'r02'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, 'gpurun_out', 'refresh'), os.path.join(root, 'profiles')
last = lambda p: open(os.path.join(src, p)).read().strip().splitlines()[-1] + '\n'
for a, b in (('bench.json', '_bench.json'), ('rtm_bench.json', '_rtm_bench.json'), ('c5_bench.json', '_c5_bench.json'),
             ('c5_bench_fp32_products.json', '_c5_bench_fp32_products.json')):
    if os.path.exists(os.path.join(src, a)):
        open(os.path.join(dst, rnd + b), 'w').write(last(a))
for a, b in (('bench_kernel_stats.csv', '_bench_kernel_stats.csv'), ('rtm_kernel_stats.csv', '_rtm_kernel_stats.csv'),
             ('step_timeline.txt', '_step_timeline.txt'), ('rtm_step_timeline.txt', '_rtm_step_timeline.txt'), ('mfma_utilisation.md', '_mfma_utilisation.md'),
             ('sq_counters.txt', '_sq_counters.txt'), ('inst_counters.txt', '_inst_counters.txt'),
             ('rtm_embed_pmc.txt', '_rtm_embed_pmc.txt'), ('gather_score_c5_pmc.txt', '_gather_score_c5_pmc.txt'),
             ('gather_c5.log', '_gather_c5_shape.jsonl'), ('c5_step_timeline.txt', '_c5_step_timeline.txt'),
             ('c5_kernel_stats.csv', '_c5_kernel_stats.csv'), ('det_step_timeline.txt', '_det_step_timeline.txt'),
             ('det_rtm_step_timeline.txt', '_det_rtm_step_timeline.txt'), ('mlp_stamps.txt', '_mlp_stamps.txt'),
             ('gather_score_kernel_stats.csv', '_gather_score_kernel_stats.csv'),
             ('gather_score_b8192_kernel_stats.csv', '_gather_score_b8192_kernel_stats.csv'),
             ('gather_score_wg_times.txt', '_gather_score_wg_times.txt'), ('kvq_wg_times.txt', '_kvq_wg_times.txt'),
             ('attn_bwd_wg_times.txt', '_attn_bwd_wg_times.txt'), ('score_bwd_wg_times.txt', '_score_bwd_wg_times.txt')):
    if os.path.exists(os.path.join(src, a)):
        shutil.copyfile(os.path.join(src, a), os.path.join(dst, rnd + b))
print(sorted(f for f in os.listdir(dst) if f.startswith(rnd)))
