#!/bin/bash
for w in 0.6 0.8 0.9 1.0 1.1 1.25; do echo "== diag weight $w"; TABMAT_AMD_PT_W=$w timeout 200 python scripts/dev/time_k2.py 2>&1 | grep -E "v4"; done
