set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for wf in 1 0; do
O=gpurun_out/c2_alone_$wf; rm -rf $O; mkdir -p $O
PS_ATTN_WF=$wf PS_NO_SIDE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 60 --warmup 10 --cpu-steps 0 --no-extras > $O/prof.json 2> $O/prof.err
python tools/trace_step.py $O/prof > $O/step_timeline.txt
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
done
