#!/bin/bash
# A/B of kernel experiments: every tabmat_amd/_abl/*.so and the default library, same box
timeout 200 python scripts/dev/time_k3_variants.py "$@" 2>&1 | tail -1
for so in tabmat_amd/_abl/*.so; do TABMAT_AMD_LIB=$PWD/$so timeout 200 python scripts/dev/time_k3_variants.py "$@" 2>&1 | tail -1; done
timeout 200 python scripts/dev/time_k3_variants.py "$@" 2>&1 | tail -1
