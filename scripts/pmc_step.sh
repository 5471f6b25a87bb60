#!/bin/bash
# SQ counters of the big kernels of one cfg4 step (bench.py, 2 steps): one rocprofv3 pass per counter group
# (--pmc only).  usage (through gpurun): bash scripts/pmc_step.sh > gpurun_out/r3_sq_counters_raw.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/pk$i
  rocprofv3 --pmc $grp -d /tmp/pk$i -o k --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic > /tmp/pk$i.log 2>&1
  echo "## group $i: $grp"
  python $R/scripts/pmc_summary.py /tmp/pk$i 2>&1 | grep -A7 "csr_dense_ent_kernel\|csr_dense_lg_kernel\|sparse_sandwich_blocks_kernel\|syrk_i8_kernel<\|multi_cat_dense_wide_kernel\|multi_cat_sparse_ent_kernel"
done
