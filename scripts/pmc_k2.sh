#!/bin/bash
# PMC counters of the K2 kernel (scripts/time_k2.py), one rocprofv3 pass per counter group.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d /tmp/pk$i -o k2 --output-format csv -- python $R/scripts/${1:-time_k2.py} > $R/gpurun_out/pmc_k2_$i.log 2>&1
  python $R/scripts/pmc_summary.py /tmp/pk$i 2>&1 | grep -A6 "${2:-sparse_sandwich_chunked}"
done
