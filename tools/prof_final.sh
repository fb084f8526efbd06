# Final-round evidence: default bench (with cpu_baseline), kernel-trace stats, and HBM traffic counters (separate passes).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/bench.py > $R/gpurun_out/bench_default.json 2> $R/gpurun_out/bench_default.err
tail -c 3000 $R/gpurun_out/bench_default.json
rocprofv3 --kernel-trace --stats -d /tmp/pf_trace -o b -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/bench_traced.json 2>/dev/null
python $R/tools/rocpd_summary.py /tmp/pf_trace/b_results.db $R/gpurun_out/kernel_stats.txt e3d > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf_fetch -o b -- python $R/bench.py --no-cpu-baseline --steps 2 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_fetch/b_results.db $R/gpurun_out/pmc_fetch.txt e3d > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pf_write -o b -- python $R/bench.py --no-cpu-baseline --steps 2 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_write/b_results.db $R/gpurun_out/pmc_write.txt e3d > /dev/null
grep -E "k_nn_rows|k_lm_pass|k_lm_cost|k_compact|k_transform_bbox" $R/gpurun_out/kernel_stats.txt | cut -c1-40,150-260
grep -E "FETCH|WRITE" $R/gpurun_out/pmc_fetch.txt $R/gpurun_out/pmc_write.txt | grep -E "k_nn_rows|k_lm_pass|k_lm_cost|k_compact|k_transform_bbox" | cut -c1-80,120-260
