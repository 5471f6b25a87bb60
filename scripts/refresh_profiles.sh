#!/bin/bash
# Regenerates the evidence under profiles/ on the GPU box (run through gpurun; outputs land in
# gpurun_out/r2_profiles/, copy them to profiles/ afterwards):
#   bench JSON lines for cfg4 / cfg2 / cfg3 (with in-run roofline.traffic), the rocprofv3 kernel
#   stats of the cfg4 command, SQ counters of the three big kernels and their FETCH / WRITE sizes.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r3}
O=$R/gpurun_out/${TAG}_profiles
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 --breakdown --out $O/${TAG}_bench_cfg4_full.json > $O/${TAG}_bench_cfg4.json 2> $O/bench_cfg4.err
python $R/bench.py --workload cfg2 --steps 20 --warmup 5 > $O/${TAG}_bench_cfg2.json 2> $O/bench_cfg2.err
python $R/bench.py --workload cfg3 --steps 50 --warmup 10 > $O/${TAG}_bench_cfg3.json 2> $O/bench_cfg3.err
rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > /dev/null 2>&1
cp /tmp/ks/ks_kernel_stats.csv $O/${TAG}_bench_cfg4_kernel_stats_raw.csv
python $R/scripts/kernel_stats_summary.py /tmp/ks/ks_kernel_stats.csv > $O/${TAG}_bench_cfg4_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c
  rocprofv3 --pmc $c -d /tmp/pm_$c -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic > /dev/null 2>&1
  python $R/scripts/pmc_summary.py /tmp/pm_$c > $O/pmc_$c.txt
done
ls -la $O
