for dbg in 0 1 2 3 4 5 6 7; do echo "dbg=$dbg (1: no slab copy, 2: no stream loads, 4: no compute)"; TM_LG_DBG=$dbg python scripts/dev/time_k3_lg.py 10000000 f64 2>&1 | grep "lg unc"; done
