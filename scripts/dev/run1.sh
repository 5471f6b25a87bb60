set -x
mkdir -p gpurun_out
python -m pytest tests/test_lg_twin.py -x -q -m gpu 2>&1 | tail -8
python scripts/dev/time_k3_lg.py 10000000 f64 2>&1 | tail -8
python scripts/dev/time_k3_lg.py 10000000 f32 2>&1 | tail -8
