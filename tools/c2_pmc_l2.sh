# gpurun -- bash tools/c2_pmc_l2.sh : what the C2 step's kernels ask of the L2s (requests per launch; a pass of its own, no trace domain)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c2_l2; rm -rf $O; mkdir -p $O
rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/a -- python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-extras > $O/a.log 2>&1 || { tail -5 $O/a.log; exit 1; }
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum --output-format csv -d $O/b -- python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-extras > $O/b.log 2>&1 || { tail -5 $O/b.log; }
( echo "rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum -- python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-extras   (per-launch averages; a TCC request is one 128-byte line or part of it)"
  python tools/pmc_summary.py $O/a
  echo; echo "rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum -- same command"
  python tools/pmc_summary.py $O/b 2>/dev/null ) > gpurun_out/r05_c2_pmc_l2.txt
rm -rf $O/a $O/b
cut -c1-150 gpurun_out/r05_c2_pmc_l2.txt
