# rocprofv3 kernel stats of one bench.py workload -> gpurun_out/prof_<w>/stats.txt
w=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$w
rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o p --output-format csv -- python $R/bench.py --workload $w --steps 20 --warmup 3 --no-traffic --no-cpu-baseline > /tmp/prof_$w.json 2>/tmp/prof_$w.err
mkdir -p $R/gpurun_out/prof_$w
python $R/scripts/kernel_stats_summary.py /tmp/prof_$w/p_kernel_stats.csv > $R/gpurun_out/prof_$w/stats.txt
head -30 $R/gpurun_out/prof_$w/stats.txt
