cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/bt; rm -rf $O; mkdir -p $O
python bench.py --steps 100 --warmup 20 --reps 100 --cpu-steps 1 > $O/c2.json 2> $O/c2.err; tail -c 2500 $O/c2.json; tail -3 $O/c2.err
python bench.py --workload c4 --steps 100 --warmup 10 --reps 100 --cpu-steps 1 > $O/c4.json 2> $O/c4.err; tail -c 2500 $O/c4.json; tail -3 $O/c4.err
