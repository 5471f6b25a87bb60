#!/bin/bash
# K3 lane-group kernel: 16 waves x 16 columns (NG=1) vs 8 waves x 32 columns (NG=2)
for ng in 1 2; do echo "== NG=$ng"; TABMAT_AMD_LG_NG=$ng timeout 300 python scripts/dev/time_k3_lg.py 10000000 f64 2>&1 | tail -3; done
