cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_reg_$c -o p -- python $R/tools/bench_reg.py 4000000 3840 2160 > /dev/null 2>&1
  python $R/tools/rocpd_summary.py /tmp/pmc_reg_$c/p_results.db $R/gpurun_out/reg_pmc_$c.txt k_reg > /dev/null
done
head -12 $R/gpurun_out/reg_pmc_FETCH_SIZE.txt | cut -c1-200
head -12 $R/gpurun_out/reg_pmc_WRITE_SIZE.txt | cut -c1-200
