# Runs on the MI355X box (gpurun): everything profiles/r05_* is built from, under gpurun_out/refresh/.
#   bash tools/refresh_profiles_r05.sh            then here:  python tools/install_profiles_r02.py r05 && python tools/make_profile_summary_r05.py
# Counter passes (--pmc) run alone, never with a trace domain; the program comes directly after `--`.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/refresh; rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 100 --warmup 10 --cpu-steps 0 --no-also > $O/prof_bench.json 2> $O/prof.err
python tools/trace_step.py $O/prof > $O/step_timeline.txt
cp $(ls -t $(find $O/prof -name '*kernel_stats.csv') | head -1) $O/bench_kernel_stats.csv
echo "trace done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-extras > $O/pmc_mfma.log 2>&1
python tools/mfma_util.py $O/pmc_mfma r05 > $O/mfma_utilisation.md
echo "mfma counters done"
# review transformer (configs[3])
python bench.py --workload c4 --steps 200 --warmup 20 > $O/rtm_bench.json 2> $O/rtm_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rtm_prof -- python3 bench.py --workload c4 --steps 60 --warmup 10 --cpu-steps 0 --no-extras > $O/rtm_prof.json 2> $O/rtm_prof.err
cp $(ls -t $(find $O/rtm_prof -name '*kernel_stats.csv') | head -1) $O/rtm_kernel_stats.csv
python tools/trace_step.py $O/rtm_prof 30 rtm_embed4 > $O/rtm_step_timeline.txt
echo "rtm done"
# gather+score at the HBM-bound C5 shape (8 M-row table, rotating index sets): kernel trace of the very loop bench.py times, and PMC traffic
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gs_prof -- python3 tools/gather_c5.py --rows 8000000 --batch 1024 --iters 48 > $O/gs_prof.json 2> $O/gs_prof.err
cp $(ls -t $(find $O/gs_prof -name '*kernel_stats.csv') | head -1) $O/gather_score_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gs_prof8 -- python3 tools/gather_c5.py --rows 8000000 --batch 8192 --iters 48 > $O/gs_prof8.json 2> $O/gs_prof8.err
cp $(ls -t $(find $O/gs_prof8 -name '*kernel_stats.csv') | head -1) $O/gather_score_b8192_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_gf -- python3 tools/gather_c5.py --rows 8000000 --batch 1024 --iters 24 > $O/pmc_gf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_gw -- python3 tools/gather_c5.py --rows 8000000 --batch 1024 --iters 24 > $O/pmc_gw.log 2>&1
( echo "rocprofv3 --pmc FETCH_SIZE -- python3 tools/gather_c5.py --rows 8000000 --batch 1024 --iters 24   (8 rotating index sets; per-launch average, KB; gfx950: x2 for 16-B/lane streaming reads)"; python tools/pmc_summary.py $O/pmc_gf | grep -E "kernel|score_fwd"; echo; echo "rocprofv3 --pmc WRITE_SIZE -- same command   (per-launch average, KB)"; python tools/pmc_summary.py $O/pmc_gw | grep -E "kernel|score_fwd"; echo; tail -1 $O/pmc_gf.log ) > $O/gather_score_c5_pmc.txt
for B in 1024 8192; do python tools/gather_c5.py --rows 8000000 --batch $B --iters 48 2>/dev/null | tail -1; done > $O/gather_c5.log
python tools/gather_c5.py --rows 8000000 --batch 1024 --iters 48 --sets 1 2>/dev/null | tail -1 >> $O/gather_c5.log
python tools/gather_wg_times.py --batch 1024 > $O/gather_score_wg_times.txt 2>&1 || true
PS_SCORE_SIDX=0 python tools/gather_wg_times.py --batch 1024 >> $O/gather_score_wg_times.txt 2>&1 || true
python tools/kvq_wg_times.py > $O/kvq_wg_times.txt 2>&1 || true
python tools/attn_bwd_wg_times.py > $O/attn_bwd_wg_times.txt 2>&1 || true
python tools/score_bwd_wg_times.py > $O/score_bwd_wg_times.txt 2>&1 || true
python tools/mlp_stamps.py > $O/mlp_stamps.txt 2>&1 || true
echo "gather done"
python bench.py --workload c5 --items 8000000 --steps 100 --warmup 10 --cpu-steps 0 > $O/c5_bench.json 2> $O/c5_bench.err || true
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_prof -- python3 bench.py --workload c5 --items 8000000 --steps 30 --warmup 5 --cpu-steps 0 --no-extras > $O/c5_prof.json 2> $O/c5_prof.err
python tools/trace_step.py $O/c5_prof > $O/c5_step_timeline.txt
cp $(ls -t $(find $O/c5_prof -name '*kernel_stats.csv') | head -1) $O/c5_kernel_stats.csv
echo "c5 done"
rm -rf $O/pmc_mfma $O/pmc_gf $O/pmc_gw
find $O -name '*kernel_trace.csv' -delete
find $O -name '*agent_info.csv' -delete
tail -c 600 $O/bench.json; echo; tail -c 300 $O/rtm_bench.json
