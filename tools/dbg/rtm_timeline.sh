set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/rtm_tl; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --workload c4 --steps 60 --warmup 10 --cpu-steps 0 --no-extras > $O/prof.json 2> $O/prof.err
python tools/trace_step.py $O/prof > $O/step_timeline.txt
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
