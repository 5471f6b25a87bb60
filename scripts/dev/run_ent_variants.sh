#!/bin/bash
# A/B of the entry-list K3 kernel: every tabmat_amd/_abl/libtabmat_ent_*.so and the default library, same box
timeout 200 python scripts/dev/time_k3_ent_only.py "$@" 2>&1 | tail -1
for so in tabmat_amd/_abl/libtabmat_ent_*.so; do TABMAT_AMD_LIB=$PWD/$so timeout 200 python scripts/dev/time_k3_ent_only.py "$@" 2>&1 | tail -1; done
timeout 200 python scripts/dev/time_k3_ent_only.py "$@" 2>&1 | tail -1
