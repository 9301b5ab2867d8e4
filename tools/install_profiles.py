"""Copy what tools/refresh_profiles.sh left under gpurun_out/refresh/ into profiles/ (tracked) and derive the summaries.
    python tools/install_profiles.py r01"""
import glob, json, os, re, shutil, subprocess, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else 'r01'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, 'gpurun_out', 'refresh'), os.path.join(root, 'profiles')
cp = lambda a, b: shutil.copyfile(os.path.join(src, a), os.path.join(dst, b))
line = open(os.path.join(src, 'bench.json')).read().strip().splitlines()[-1]
open(os.path.join(dst, rnd + '_bench.json'), 'w').write(line + '\n')
stats = max(glob.glob(src + '/prof/**/*kernel_stats.csv', recursive=True), key=os.path.getmtime)   # newest run
shutil.copyfile(stats, os.path.join(dst, rnd + '_bench_kernel_stats.csv'))
cp('step_timeline.txt', rnd + '_step_timeline.txt')
cp('gather_score_pmc.txt', rnd + '_gather_score_pmc.txt')
cp('mfma_utilisation.md', rnd + '_mfma_utilisation.md')
cp('gather_c5.log', rnd + '_gather_c5_shape.jsonl')
open(os.path.join(dst, rnd + '_rtm_bench.json'), 'w').write(open(os.path.join(src, 'rtm.log')).read().strip().splitlines()[-1] + '\n')
# HBM traffic of the gather+score launch from the two PMC passes (KB per launch; gfx950: FETCH_SIZE x2 for 16-B/lane reads)
txt = open(os.path.join(src, 'gather_score_pmc.txt')).read()
vals = [float(m.group(1)) for m in re.finditer(r'score_fwd\w*kernel.*?\s(\d+)\s*$', txt, re.M)]
fetch_kb, write_kb = vals[0], vals[1]
bj = json.loads(line)
R = bj['config']['replicas_per_row']
rd = 2 * fetch_kb * 1024
note = ("FETCH_SIZE %.0f KB -> x2 (gfx950 16-B/lane streaming-read correction, MI355X_MICROARCH.md §HBM) = %.2f MB read, "
        "WRITE_SIZE %.0f KB; algorithmic bytes %.2f MB -> no re-reads." % (fetch_kb, rd / 1e6, write_kb, bj['roofline']['bytes_per_launch'] / 1e6))
json.dump({"source": "profiles/%s_gather_score_pmc.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, MI355X)" % rnd,
           "R%d_bytes_per_launch" % R: int(rd + write_kb * 1024), "note": note},
          open(os.path.join(dst, 'gather_score_traffic.json'), 'w'))
subprocess.check_call([sys.executable, os.path.join(root, 'tools', 'make_profile_summary.py'), os.path.join(src, 'prof'),
                       os.path.join(src, 'prof_bench.json'), os.path.join(src, 'bench.json'), rnd])
print(note)
