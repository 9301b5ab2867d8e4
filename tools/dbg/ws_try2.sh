cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ws; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_shapes.py tests/test_gpu_parity_rtm.py tests/test_gpu_fullsize_rtm.py -q -x > $O/tests.log 2>&1; tail -5 $O/tests.log
python bench.py --steps 200 --warmup 20 --reps 0 --cpu-steps 0 > $O/c2.json 2> $O/c2.err; python -c "
import json;d=json.load(open('$O/c2.json'));print('ws  ', d['ms_per_step'])"
PS_MLP_BWD_WS=0 python bench.py --steps 200 --warmup 20 --reps 0 --cpu-steps 0 > $O/c2o.json 2> $O/c2o.err; python -c "
import json;d=json.load(open('$O/c2o.json'));print('bwd old ', d['ms_per_step'])"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 50 --warmup 10 --cpu-steps 0 --no-extras > $O/prof.json 2> $O/prof.err
python tools/trace_step.py $O/prof > $O/timeline.txt; cat $O/timeline.txt
find $O/prof -name '*kernel_trace.csv' -delete
