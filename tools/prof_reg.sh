cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/bench_reg.py 4000000 3840 2160
rocprofv3 --kernel-trace --stats -d /tmp/pr -o r -- python $R/tools/bench_reg.py 4000000 3840 2160 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pr/r_results.db $R/gpurun_out/reg_kernel_stats.txt e3d > /dev/null
cut -c1-70,150-240 $R/gpurun_out/reg_kernel_stats.txt | head -20
