#!/bin/bash
# PMC counters of one kernel: scripts/pmc_run.sh "<python script + args>" <kernel-name filter> [tag]
# One rocprofv3 pass per counter group (--pmc only, no trace domains).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
tag=${3:-pmc}
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rm -rf /tmp/pk$i
  rocprofv3 --pmc $grp -d /tmp/pk$i -o k --output-format csv -- python $R/$1 > $R/gpurun_out/${tag}_$i.log 2>&1
  python $R/scripts/pmc_summary.py /tmp/pk$i 2>&1 | grep -A7 "$2"
done
