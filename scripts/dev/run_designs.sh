# the reference's own benchmark designs (benchmark/generate_matrices.py) + cfg1 through bench.py -> gpurun_out/r6_designs/
mkdir -p gpurun_out/r6_designs
for w in dense one_cat two_cat dense_cat dense_smallcat sparse sparse_narrow sparse_wide cfg1; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-traffic > gpurun_out/r6_designs/r6_bench_$w.json 2> gpurun_out/r6_designs/$w.err || echo "FAILED $w"
  python -c "
import json
d=json.load(open('gpurun_out/r6_designs/r6_bench_$w.json')); print('$w', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], (d.get('matvec') or {}).get('ms'), (d.get('transpose_matvec') or {}).get('ms'))"
done
