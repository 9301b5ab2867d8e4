# Runs on the MI355X box (gpurun): everything profiles/ is built from, under gpurun_out/refresh/.
set -e
export PS_SIDE_EVENTS_NOTE="counter passes run with event-pair stream crossings (auto-detected from ROCPROF_COUNTER_COLLECTION)"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/refresh; rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 100 --warmup 10 --cpu-steps 0 > $O/prof_bench.json 2> $O/prof.err
python tools/trace_step.py $O/prof > $O/step_timeline.txt
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/gather_only.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/gather_only.py > $O/pmc_write.log 2>&1
( echo "rocprofv3 --pmc FETCH_SIZE -- python3 tools/gather_only.py   (per-launch average, KB)"; python tools/pmc_summary.py $O/pmc_fetch; echo; echo "rocprofv3 --pmc WRITE_SIZE -- python3 tools/gather_only.py"; python tools/pmc_summary.py $O/pmc_write ) > $O/gather_score_pmc.txt
echo "pmc done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-extras > $O/pmc_mfma.log 2>&1
python tools/mfma_util.py $O/pmc_mfma r01 > $O/mfma_utilisation.md
echo "mfma done"
python tools/bench_rtm.py > $O/rtm.log 2>&1 || true
tail -2 $O/rtm.log
for B in 1024 8192; do python tools/gather_c5.py --rows 8000000 --batch $B --iters 50 2>/dev/null | tail -1; done > $O/gather_c5.log
cat $O/gather_c5.log
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_mfma
find $O/prof -name '*kernel_trace.csv' -delete
