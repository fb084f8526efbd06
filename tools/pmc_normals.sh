# SQ counters of k_knn_normals (k = 32, 20 M points), two passes (8 counters each); summaries to gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU -d /tmp/pn_a -o p -- python $R/tools/bench_normals.py --k 32 --no-cpu --repeat 1 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pn_a/p_results.db $R/gpurun_out/normals_pmc_sq_a.txt k_knn > /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM -d /tmp/pn_b -o p -- python $R/tools/bench_normals.py --k 32 --no-cpu --repeat 1 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pn_b/p_results.db $R/gpurun_out/normals_pmc_sq_b.txt k_knn > /dev/null
cat $R/gpurun_out/normals_pmc_sq_a.txt $R/gpurun_out/normals_pmc_sq_b.txt | cut -c1-40,150-260
