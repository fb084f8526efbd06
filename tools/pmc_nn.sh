cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-20000000}
for mode in ${2:-2 3}; do
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU -d /tmp/pmc_a_m$mode -o p -- python $R/tools/prof_nn.py $N $mode 2 2>&1 | grep -E "^mode|rror"
python $R/tools/rocpd_summary.py /tmp/pmc_a_m$mode/p_results.db $R/gpurun_out/pmc_a_m$mode.txt k_nn > /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA -d /tmp/pmc_b_m$mode -o p -- python $R/tools/prof_nn.py $N $mode 2 2>&1 | grep -E "^mode|rror"
python $R/tools/rocpd_summary.py /tmp/pmc_b_m$mode/p_results.db $R/gpurun_out/pmc_b_m$mode.txt k_nn > /dev/null
done
cat $R/gpurun_out/pmc_*.txt
