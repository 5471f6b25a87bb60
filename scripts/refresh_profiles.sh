#!/bin/bash
# Regenerates the evidence under profiles/ on the GPU box (run through gpurun; outputs land in
# gpurun_out/r1_profiles/, copy them to profiles/ afterwards):
#   bench JSON lines for cfg4 / cfg2 / cfg3, rocprofv3 kernel stats of the cfg4 command, and the
#   FETCH_SIZE / WRITE_SIZE passes the roofline `traffic` figures come from.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r1_profiles
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $O/r1_bench_cfg4.json 2> $O/bench_cfg4.err
python $R/bench.py --workload cfg2 --steps 20 --warmup 5 > $O/r1_bench_cfg2.json 2> $O/bench_cfg2.err
python $R/bench.py --workload cfg3 --steps 50 --warmup 10 > $O/r1_bench_cfg3.json 2> $O/bench_cfg3.err
rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cp /tmp/ks/ks_kernel_stats.csv $O/r1_bench_cfg4_kernel_stats_raw.csv
python $R/scripts/kernel_stats_summary.py /tmp/ks/ks_kernel_stats.csv > $O/r1_bench_cfg4_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d /tmp/pm_$c -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python $R/scripts/pmc_summary.py /tmp/pm_$c > $O/pmc_$c.txt
done
ls -la $O
