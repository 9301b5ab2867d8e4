set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/det_tl; rm -rf $O; mkdir -p $O
PS_DETERMINISTIC=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-extras > $O/prof.json 2> $O/prof.err
python tools/trace_step.py $O/prof > $O/step_timeline.txt
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
