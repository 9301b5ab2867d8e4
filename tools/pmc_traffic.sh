# gpurun -- bash tools/pmc_traffic.sh <tag> <bench args...> : FETCH_SIZE / WRITE_SIZE per launch of every kernel of a bench workload
# (separate passes; gfx950: FETCH_SIZE x2 for 16-B/lane streaming reads; L2 fabric side, Infinity-Cache hits included)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
O=gpurun_out/pmc_$tag; rm -rf $O; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 bench.py "$@" --steps 12 --warmup 4 --cpu-steps 0 --no-extras > $O/f.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 bench.py "$@" --steps 12 --warmup 4 --cpu-steps 0 --no-extras > $O/w.log 2>&1 || exit 1
( echo "rocprofv3 --pmc FETCH_SIZE -- python3 bench.py $* --steps 12 --warmup 4 --cpu-steps 0 --no-extras   (per-launch average, KB as reported; x2 for 16-B/lane reads)"
  python tools/pmc_summary.py $O/f
  echo; echo "rocprofv3 --pmc WRITE_SIZE -- same command   (per-launch average, KB)"
  python tools/pmc_summary.py $O/w ) > gpurun_out/r05_${tag}_pmc_traffic.txt
rm -rf $O/f $O/w
