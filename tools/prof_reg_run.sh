cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/prof_reg_run.py 0 4
rocprofv3 --kernel-trace --stats -d /tmp/pr -o r -- python $R/tools/prof_reg_run.py 0 4 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pr/r_results.db $R/gpurun_out/reg_run_kernel_stats.txt e3d > /dev/null
cut -d, -f1-4 $R/gpurun_out/reg_run_kernel_stats.txt | sed 's/(.*)//' | head -24
