set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PS_GEMM_X3=1
O=gpurun_out/c5_tl_x3; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --workload c5 --items 8000000 --steps 30 --warmup 5 --cpu-steps 0 --no-extras > $O/prof.json 2> $O/prof.err
python tools/trace_step.py $O/prof > $O/step_timeline.txt
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
