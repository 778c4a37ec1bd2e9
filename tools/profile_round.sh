set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
for E in 1 8; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_e$E -o runc --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-second --spec-eps $E > $O/bench_e${E}_under_rocprof.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_e$E -o runc --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-second --spec-eps $E > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_e$E -o runc --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-second --spec-eps $E > /dev/null 2>&1
done
find $O -name "*.csv" | head -30
tail -1 $O/bench_e1_under_rocprof.log | cut -c1-300
