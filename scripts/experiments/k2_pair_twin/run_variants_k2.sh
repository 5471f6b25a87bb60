#!/bin/bash
# A/B of K2 experiments: the default library and every tabmat_amd/_abl/*.so
timeout 200 python scripts/dev/time_k2.py "$@" 2>&1 | grep -E "v3|v4"
for so in tabmat_amd/_abl/*.so; do echo "== $so"; TABMAT_AMD_LIB=$PWD/$so timeout 200 python scripts/dev/time_k2.py "$@" 2>&1 | grep -E "v4"; done
