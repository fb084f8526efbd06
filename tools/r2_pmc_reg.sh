#!/bin/bash
# SQ counters of the (B) accumulate kernels at the configs[3] image shape (3 images instead of 23, accumulate passes only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
CMD="python $R/tools/bench_c4.py --images ${IMAGES:-3} --accumulate-only"
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU -d /tmp/pr_a -o p -- $CMD > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pr_a/p_results.db $R/gpurun_out/r2_reg_pmc_sq_a.txt k_reg_pass > /dev/null
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM -d /tmp/pr_b -o p -- $CMD > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pr_b/p_results.db $R/gpurun_out/r2_reg_pmc_sq_b.txt k_reg_pass > /dev/null
for f in a b; do grep -v "^#" $R/gpurun_out/r2_reg_pmc_sq_$f.txt | awk -F', ' 'NF==4{print substr($1,1,34), $2, $3, $4}'; done
grep "k_reg_pass" $R/gpurun_out/r2_reg_pmc_sq_a.txt | head -4 | awk -F', ' 'NF>4{print substr($1,1,34), $(NF-4), $(NF-2)}'
