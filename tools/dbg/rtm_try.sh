cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/rtmt; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_parity_rtm.py tests/test_gpu_fullsize_rtm.py tests/test_gpu_trainer.py -q -x > $O/tests.log 2>&1; tail -5 $O/tests.log
python bench.py --workload c4 --steps 100 --warmup 10 --reps 0 --cpu-steps 0 > $O/c4.json 2> $O/c4.err; python -c "
import json;d=json.load(open('$O/c4.json'));print('e4  ', d['ms_per_step'], d['roofline']['us_per_launch'], d['roofline']['frac'])"
PS_RTM_EMBED4=0 python bench.py --workload c4 --steps 100 --warmup 10 --reps 0 --cpu-steps 0 > $O/c4o.json 2> $O/c4o.err; python -c "
import json;d=json.load(open('$O/c4o.json'));print('old ', d['ms_per_step'], d['roofline']['us_per_launch'], d['roofline']['frac'])"
