cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/x3; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_shapes.py tests/test_gpu_parity_rtm.py tests/test_gpu_fullsize_rtm.py -q -x > $O/tests.log 2>&1; tail -5 $O/tests.log
python bench.py --steps 200 --warmup 20 --reps 0 --cpu-steps 0 > $O/c2.json 2> $O/c2.err; python -c "
import json;d=json.load(open('$O/c2.json'));print('x3  ', d['ms_per_step'], d['roofline']['us_per_launch'])"
PS_MLP_X3=0 python bench.py --steps 200 --warmup 20 --reps 0 --cpu-steps 0 > $O/c2o.json 2> $O/c2o.err; python -c "
import json;d=json.load(open('$O/c2o.json'));print('f32 ', d['ms_per_step'], d['roofline']['us_per_launch'])"
