#!/bin/bash
# SQ / TA / TCP counters of the (B) accumulate kernels at the configs[3] image shape (3 images instead of 23)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/tools/bench_c4.py --images ${IMAGES:-3}"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU -d /tmp/pr_a -o p -- $CMD > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pr_a/p_results.db $R/gpurun_out/r2_reg_pmc_sq_a.txt k_reg_pass > /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM -d /tmp/pr_b -o p -- $CMD > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pr_b/p_results.db $R/gpurun_out/r2_reg_pmc_sq_b.txt k_reg_pass > /dev/null
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE -d /tmp/pr_c -o p -- $CMD > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pr_c/p_results.db $R/gpurun_out/r2_reg_pmc_ta.txt k_reg_pass > /dev/null
for f in a b; do grep -v "^#" $R/gpurun_out/r2_reg_pmc_sq_$f.txt | awk -F', ' 'NF==4{print substr($1,1,34), $2, $3, $4}'; done
grep -v "^#" $R/gpurun_out/r2_reg_pmc_ta.txt | awk -F', ' 'NF==4{print substr($1,1,34), $2, $3, $4}'
grep "k_reg_pass" $R/gpurun_out/r2_reg_pmc_sq_a.txt | head -4 | awk -F', ' 'NF>4{print substr($1,1,34), $(NF-4), $(NF-2)}'
